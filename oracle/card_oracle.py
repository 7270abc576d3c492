"""CPU oracle for CARD's reverse-diffusion sampling path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The product path (`card_amd/`) never imports, links or falls back to anything
under `oracle/`; it fails loudly when the HIP library is missing.

It is an independent restatement (own code, same aten-op order) of the
reference's algorithm for the hot path, so that on one machine it is bit-equal
to the reference for a given injected noise tensor:

  * coefficient table     <- regression/diffusion_utils.py:71-80,:88 (p_sample),
                              :97-98 (p_sample_t_1to0)
  * eps_theta (guided MLP) <- regression/model.py:14-18 (ConditionalLinear),
                              :37-51 (ConditionalGuidedModel.forward)
  * eps_theta (cls head)   <- classification/model.py:79-94 (ConditionalModel.forward,
                              eval-mode BatchNorm1d)
  * eps_theta (ImageNet)   <- classification/imagenet/card/builder.py:54-92 (linear head, no BN/softplus)
  * p_sample               <- regression/diffusion_utils.py:57-90,
                              classification/diffusion_utils.py:54-87
  * p_sample_t_1to0        <- regression/diffusion_utils.py:94-104
  * p_sample_loop          <- regression/diffusion_utils.py:107-119,
                              classification/diffusion_utils.py:119-144
  * q_sample               <- regression/diffusion_utils.py:42-53
  * f_phi (guidance net)   <- regression/model.py:55-82 (eval mode), OLS guidance regression/card_regression.py:115-117

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by fixtures produced by importing
the reference itself in the build container (`oracle/gen_golden.py` ->
`tests/golden/*.npz`) and checked in `tests/test_oracle_golden.py`.

Noise: the reference draws `torch.randn_like` once for y_T and once per p_sample
call (T draws in total).  `noise[d]`, d = 0..T-1, is draw number d: d = 0 makes
y_T, and the p_sample call for timestep index t consumes draw d = T - t.
`philox_normal` restates the in-kernel counter-based generator (Philox4x32-10 +
Box-Muller) so RNG-mode kernel output can be checked against the oracle too.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# columns of the per-timestep coefficient table (fp32); row index = timestep index t
C_INV_SQRT_AB = 0      # 1 / sqrt(alpha_bar_t)
C_ONE_M_SQRT_AB = 1    # 1 - sqrt(alpha_bar_t)
C_SOMAB = 2            # sqrt(1 - alpha_bar_t)  (the input table itself)
C_G0 = 3               # gamma_0
C_G1 = 4               # gamma_1
C_G2 = 5               # gamma_2
C_SIGMA = 6            # sqrt(beta_t_hat)
C_SQRT_AB = 7          # sqrt(alpha_bar_t) (kept for q_sample-style uses / debugging)
COEF_COLS = 8


def make_coef_table(alphas: torch.Tensor, one_minus_alphas_bar_sqrt: torch.Tensor) -> torch.Tensor:
    """(T, 8) fp32 table of everything p_sample derives from `t` alone.

    Same fp32 op order as regression/diffusion_utils.py:71-80 and :88; row 0 holds only
    the p_sample_t_1to0 entries (:97-98), its gamma/sigma columns are zero."""
    a = alphas.detach().to(torch.float32).cpu().reshape(-1)
    s = one_minus_alphas_bar_sqrt.detach().to(torch.float32).cpu().reshape(-1)
    T = a.shape[0]
    assert s.shape[0] == T
    tab = torch.zeros(T, COEF_COLS, dtype=torch.float32)
    sqrt_ab = (1 - s.square()).sqrt()                       # :74
    tab[:, C_INV_SQRT_AB] = 1 / sqrt_ab                     # :83 "1 / sqrt_alpha_bar_t"
    tab[:, C_ONE_M_SQRT_AB] = 1 - sqrt_ab                   # :84
    tab[:, C_SOMAB] = s
    tab[:, C_SQRT_AB] = sqrt_ab
    if T > 1:
        alpha_t = a[1:]
        s_t = s[1:]
        s_tm1 = s[:-1]
        sab_t = sqrt_ab[1:]
        sab_tm1 = sqrt_ab[:-1]                              # :75
        s_t_sq = s_t.square()
        tab[1:, C_G0] = (1 - alpha_t) * sab_tm1 / s_t_sq    # :77
        tab[1:, C_G1] = s_tm1.square() * alpha_t.sqrt() / s_t_sq   # :78
        tab[1:, C_G2] = 1 + (sab_t - 1) * (alpha_t.sqrt() + sab_tm1) / s_t_sq  # :79-80
        beta_hat = s_tm1.square() / s_t_sq * (1 - alpha_t)  # :88
        tab[1:, C_SIGMA] = beta_hat.sqrt()                  # :89
    return tab


# --------------------------------------------------------------------------------------
# eps_theta networks
# --------------------------------------------------------------------------------------
@dataclass
class GuidedWeights:
    """Weights of regression/model.py:21 ConditionalGuidedModel (state_dict layout)."""
    W: List[torch.Tensor]        # 3 x (H, in)
    b: List[torch.Tensor]        # 3 x (H,)
    gamma: List[torch.Tensor]    # 3 x (T+1, H)   nn.Embedding tables
    W4: torch.Tensor             # (y_dim, H)
    b4: torch.Tensor             # (y_dim,)
    cat_x: bool = True
    cat_y_pred: bool = True

    @staticmethod
    def from_state_dict(sd: Dict[str, torch.Tensor], cat_x=True, cat_y_pred=True, dtype=torch.float32):
        g = lambda k: sd[k].detach().cpu().to(dtype)
        return GuidedWeights(
            W=[g(f"lin{i}.lin.weight") for i in (1, 2, 3)],
            b=[g(f"lin{i}.lin.bias") for i in (1, 2, 3)],
            gamma=[g(f"lin{i}.embed.weight") for i in (1, 2, 3)],
            W4=g("lin4.weight"), b4=g("lin4.bias"), cat_x=cat_x, cat_y_pred=cat_y_pred)

    def to(self, dtype):
        c = lambda v: v.to(dtype)
        return GuidedWeights([c(v) for v in self.W], [c(v) for v in self.b], [c(v) for v in self.gamma],
                             c(self.W4), c(self.b4), self.cat_x, self.cat_y_pred)


def _cond_linear(h, W, b, gamma_tab, t: int):
    # ConditionalLinear.forward, regression/model.py:14-18: gamma[t].view(-1,H) * lin(x)
    out = F.linear(h, W, b)
    g = gamma_tab[t].view(-1, W.shape[0])
    return g * out


def eps_theta_guided(w: GuidedWeights, x, y_t, y_0_hat, t: int):
    """ConditionalGuidedModel.forward, regression/model.py:37-51."""
    if w.cat_x:
        if w.cat_y_pred:
            h = torch.cat((y_t, y_0_hat, x), dim=1)
        else:
            h = torch.cat((y_t, x), dim=1)
    else:
        if w.cat_y_pred:
            h = torch.cat((y_t, y_0_hat), dim=1)
        else:
            h = y_t
    for l in range(3):
        h = F.softplus(_cond_linear(h, w.W[l], w.b[l], w.gamma[l], t))
    return F.linear(h, w.W4, w.b4)


@dataclass
class ClsWeights:
    """theta part of classification/model.py:22 ConditionalModel (encoder_x/norm hoisted:
    `x_feat = norm(encoder_x(x))` is step-invariant in eval mode and is an input here)."""
    W: List[torch.Tensor]        # (F, 2C or C), (F, F), (F, F)
    b: List[torch.Tensor]
    gamma: List[torch.Tensor]    # 3 x (T+1, F)
    bn_scale: List[torch.Tensor]  # 3 x (F,)  w / sqrt(var + eps)
    bn_shift: List[torch.Tensor]  # 3 x (F,)  b - mean * scale
    bn_raw: List[Dict[str, torch.Tensor]]  # weight, bias, running_mean, running_var, eps per layer
    W4: torch.Tensor             # (C, F)
    b4: torch.Tensor
    guidance: bool = True

    @staticmethod
    def from_state_dict(sd: Dict[str, torch.Tensor], guidance=True, eps=1e-5, dtype=torch.float32):
        g = lambda k: sd[k].detach().cpu().to(dtype)
        raw, sc, sh = [], [], []
        for i in (1, 2, 3):
            r = dict(weight=g(f"unetnorm{i}.weight"), bias=g(f"unetnorm{i}.bias"),
                     running_mean=g(f"unetnorm{i}.running_mean"), running_var=g(f"unetnorm{i}.running_var"),
                     eps=eps)
            s = r["weight"] / torch.sqrt(r["running_var"] + eps)
            raw.append(r)
            sc.append(s)
            sh.append(r["bias"] - r["running_mean"] * s)
        return ClsWeights(
            W=[g(f"lin{i}.lin.weight") for i in (1, 2, 3)],
            b=[g(f"lin{i}.lin.bias") for i in (1, 2, 3)],
            gamma=[g(f"lin{i}.embed.weight") for i in (1, 2, 3)],
            bn_scale=sc, bn_shift=sh, bn_raw=raw,
            W4=g("lin4.weight"), b4=g("lin4.bias"), guidance=guidance)


def _bn_eval(h, r):
    return F.batch_norm(h, r["running_mean"], r["running_var"], r["weight"], r["bias"], False, 0.0, r["eps"])


def eps_theta_cls(w: ClsWeights, x_feat, y, y_0_hat, t: int):
    """ConditionalModel.forward after the hoisted encoder, classification/model.py:82-94."""
    h = torch.cat([y, y_0_hat], dim=-1) if w.guidance else y
    h = F.softplus(_bn_eval(_cond_linear(h, w.W[0], w.b[0], w.gamma[0], t), w.bn_raw[0]))
    h = x_feat * h
    h = F.softplus(_bn_eval(_cond_linear(h, w.W[1], w.b[1], w.gamma[1], t), w.bn_raw[1]))
    h = F.softplus(_bn_eval(_cond_linear(h, w.W[2], w.b[2], w.gamma[2], t), w.bn_raw[2]))
    return F.linear(h, w.W4, w.b4)


@dataclass
class LinearHeadWeights:
    """classification/imagenet/card/builder.py:69-92 ConditionalModel (ImageNet head): lin1, lin2 conditional,
    lin3 plain, no activation / BatchNorm; `x_feat` = linx(z) is step-invariant and an input here."""
    W: List[torch.Tensor]        # lin1 (F, 2C or C), lin2 (F, F)
    b: List[torch.Tensor]
    gamma: List[torch.Tensor]    # 2 x (n_steps, F)
    W3: torch.Tensor             # (C, F)
    b3: torch.Tensor
    guidance: bool = True

    @staticmethod
    def from_state_dict(sd, guidance=True, dtype=torch.float32):
        g = lambda k: sd[k].detach().cpu().to(dtype)
        return LinearHeadWeights(W=[g("lin1.lin.weight"), g("lin2.lin.weight")], b=[g("lin1.lin.bias"), g("lin2.lin.bias")],
                                 gamma=[g("lin1.embed.weight"), g("lin2.embed.weight")], W3=g("lin3.weight"),
                                 b3=g("lin3.bias"), guidance=guidance)


def eps_theta_linear_head(w: LinearHeadWeights, x_feat, y, y_0_hat, t: int):
    """ConditionalModel.forward, classification/imagenet/card/builder.py:86-92, with linx(z) hoisted."""
    h = torch.cat([y, y_0_hat], dim=-1) if w.guidance else y
    h = _cond_linear(h, w.W[0], w.b[0], w.gamma[0], t)
    h = x_feat * h
    h = _cond_linear(h, w.W[1], w.b[1], w.gamma[1], t)
    return F.linear(h, w.W3, w.b3)


def eps_theta(w, x, y_t, y_0_hat, t: int):
    if isinstance(w, GuidedWeights):
        return eps_theta_guided(w, x, y_t, y_0_hat, t)
    if isinstance(w, LinearHeadWeights):
        return eps_theta_linear_head(w, x, y_t, y_0_hat, t)
    return eps_theta_cls(w, x, y_t, y_0_hat, t)


# --------------------------------------------------------------------------------------
# sampler
# --------------------------------------------------------------------------------------
def p_sample_update(coef_row, y, y_T_mean, eps, z):
    """The elementwise part of p_sample, regression/diffusion_utils.py:83-89, given eps_theta."""
    c = coef_row
    y_0_reparam = c[C_INV_SQRT_AB] * (y - c[C_ONE_M_SQRT_AB] * y_T_mean - eps * c[C_SOMAB])
    y_hat = c[C_G0] * y_0_reparam + c[C_G1] * y + c[C_G2] * y_T_mean
    return y_hat + c[C_SIGMA] * z


def p_sample(w, x, y, y_0_hat, y_T_mean, t: int, coef, z):
    """One reverse step y_t -> y_{t-1} (regression/diffusion_utils.py:57-90) with injected z."""
    eps = eps_theta(w, x, y, y_0_hat, t)
    return p_sample_update(coef[t].to(y.dtype), y, y_T_mean, eps, z)


def p_sample_t_1to0(w, x, y, y_0_hat, y_T_mean, coef):
    """Last step, timestep index 0, no noise (regression/diffusion_utils.py:94-104)."""
    eps = eps_theta(w, x, y, y_0_hat, 0)
    c = coef[0].to(y.dtype)
    return c[C_INV_SQRT_AB] * (y - c[C_ONE_M_SQRT_AB] * y_T_mean - eps * c[C_SOMAB])


def p_sample_loop(w, x, y_0_hat, y_T_mean, n_steps: int, alphas, one_minus_alphas_bar_sqrt,
                  noise: torch.Tensor, only_last_sample: bool = False, coef: Optional[torch.Tensor] = None):
    """regression/diffusion_utils.py:107-119 / classification/diffusion_utils.py:119-144.

    noise: (n_steps, N, y_dim); noise[0] makes y_T, noise[n_steps - t] feeds timestep index t.
    Returns the list of n_steps+1 tensors (index k <-> timestep n_steps-k), or y_0 alone."""
    if coef is None:
        coef = make_coef_table(alphas, one_minus_alphas_bar_sqrt)
    coef = coef.to(y_T_mean.dtype)
    cur_y = noise[0] + y_T_mean
    seq = None if only_last_sample else [cur_y]
    for t in reversed(range(1, n_steps)):
        cur_y = p_sample(w, x, cur_y, y_0_hat, y_T_mean, t, coef, noise[n_steps - t])
        if seq is not None:
            seq.append(cur_y)
    y_0 = p_sample_t_1to0(w, x, cur_y, y_0_hat, y_T_mean, coef)
    if only_last_sample:
        return y_0
    seq.append(y_0)
    return seq


def q_sample(y, y_0_hat, alphas_bar_sqrt, one_minus_alphas_bar_sqrt, t, noise):
    """regression/diffusion_utils.py:42-53 with injected noise; t is an int64 tensor (N,)."""
    shape = [t.shape[0]] + [1] * (y.dim() - 1)
    sab = torch.gather(alphas_bar_sqrt, 0, t).reshape(*shape)
    somab = torch.gather(one_minus_alphas_bar_sqrt, 0, t).reshape(*shape)
    return sab * y + (1 - sab) * y_0_hat + somab * noise


def fphi_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, negative_slope: float = 0.01, bn_eps: float = 1e-5):
    """DeterministicFeedForwardNeuralNetwork.forward in eval mode (regression/model.py:70-82): per hidden layer
    Linear -> [BatchNorm1d with running statistics] -> LeakyReLU -> Dropout (identity), then a plain Linear.
    sd: the module's state_dict ("network.<i>.weight" ...; a 1-D "weight" belongs to a BatchNorm1d)."""
    idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith("network.")})
    lin = [i for i in idx if sd[f"network.{i}.weight"].dim() == 2]
    h = x
    for n, i in enumerate(lin):
        h = torch.addmm(sd[f"network.{i}.bias"], h, sd[f"network.{i}.weight"].t())
        if n == len(lin) - 1:
            return h
        j = i + 1
        if f"network.{j}.running_mean" in sd:
            h = (h - sd[f"network.{j}.running_mean"]) / torch.sqrt(sd[f"network.{j}.running_var"] + bn_eps) \
                * sd[f"network.{j}.weight"] + sd[f"network.{j}.bias"]
        h = torch.nn.functional.leaky_relu(h, negative_slope)
    return h


def ols_forward(beta: torch.Tensor, x: torch.Tensor):
    """compute_guiding_prediction(method="OLS"), regression/card_regression.py:115-117."""
    return torch.cat((torch.ones(x.shape[0], 1, dtype=x.dtype), x), dim=1).mm(beta)


def make_beta_schedule(schedule="linear", num_timesteps=1000, start=1e-5, end=1e-2):
    """regression/diffusion_utils.py:6-31 (all schedules)."""
    if schedule == "linear":
        betas = torch.linspace(start, end, num_timesteps)
    elif schedule == "const":
        betas = end * torch.ones(num_timesteps)
    elif schedule == "quad":
        betas = torch.linspace(start ** 0.5, end ** 0.5, num_timesteps) ** 2
    elif schedule == "jsd":
        betas = 1.0 / torch.linspace(num_timesteps, 1, num_timesteps)
    elif schedule == "sigmoid":
        betas = torch.sigmoid(torch.linspace(-6, 6, num_timesteps)) * (end - start) + start
    elif schedule in ("cosine", "cosine_reverse"):
        max_beta, s = 0.999, 0.008
        f = lambda i: math.cos((i / num_timesteps + s) / (1 + s) * math.pi / 2) ** 2
        betas = torch.tensor([min(1 - f(i + 1) / f(i), max_beta) for i in range(num_timesteps)])
        if schedule == "cosine_reverse":
            betas = betas.flip(0)
    elif schedule == "cosine_anneal":
        betas = torch.tensor([start + 0.5 * (end - start) * (1 - math.cos(t / (num_timesteps - 1) * math.pi))
                              for t in range(num_timesteps)])
    else:
        raise ValueError(schedule)
    return betas


def make_schedule(schedule="linear", T=1000, start=1e-4, end=0.02):
    """Diffusion.__init__, regression/card_regression.py:39-50 -> (alphas, alphas_bar_sqrt, omabs)."""
    betas = make_beta_schedule(schedule, T, start, end).float()
    alphas = 1.0 - betas
    alphas_cumprod = alphas.cumprod(dim=0)
    alphas_bar_sqrt = torch.sqrt(alphas_cumprod)
    omabs = torch.sqrt(1 - alphas_cumprod)
    if schedule == "cosine":
        omabs = omabs * 0.9999
    return alphas, alphas_bar_sqrt, omabs


# --------------------------------------------------------------------------------------
# counter-based RNG restatement (what the kernel does when noise == NULL)
# --------------------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(ctr: np.ndarray, key: Sequence[int]) -> np.ndarray:
    """Philox4x32-10 (Salmon et al., SC'11).  ctr: (..., 4) uint32, key: 2 x uint32 -> (..., 4) uint32."""
    c = [ctr[..., i].astype(np.uint32) for i in range(4)]
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c[0].astype(np.uint64)
            p1 = _PHILOX_M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return np.stack(c, axis=-1)


def philox_normal(seed: int, elem_index: np.ndarray, draw: int) -> np.ndarray:
    """Standard normal number `draw` (0..T-1) of element `elem_index` (global flat index into
    the (N, y_dim) output, uint64) as the kernel generates it: counter = (elem_lo, elem_hi,
    draw >> 2, 0), key = (seed_lo, seed_hi); the four outputs feed two Box-Muller pairs and
    draw & 3 selects one of the four normals."""
    e = np.asarray(elem_index, dtype=np.uint64)
    ctr = np.stack([(e & np.uint64(0xFFFFFFFF)).astype(np.uint32), (e >> np.uint64(32)).astype(np.uint32),
                    np.full(e.shape, draw >> 2, dtype=np.uint32), np.zeros(e.shape, dtype=np.uint32)], axis=-1)
    r = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    pair, which = (draw & 3) >> 1, draw & 1
    # 23 random bits + 0.5 is exact in fp32 and lies strictly inside (0, 1)
    u1 = ((r[..., 2 * pair] >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    u2 = ((r[..., 2 * pair + 1] >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)
    rad = np.sqrt(-2.0 * np.log(u1.astype(np.float64))).astype(np.float32)
    ang = 2.0 * np.pi * u2.astype(np.float64)
    trig = np.sin(ang) if which else np.cos(ang)
    return (rad * trig.astype(np.float32)).astype(np.float32)


def philox_noise(seed: int, n_steps: int, N: int, y_dim: int, elem_offset: int = 0) -> torch.Tensor:
    """(n_steps, N, y_dim) noise tensor equal (up to transcendental rounding) to what the kernel
    draws in RNG mode with the same seed and element offset."""
    idx = np.arange(N * y_dim, dtype=np.uint64) + np.uint64(elem_offset)
    out = np.stack([philox_normal(seed, idx, d) for d in range(n_steps)], axis=0)
    return torch.from_numpy(out.reshape(n_steps, N, y_dim))
