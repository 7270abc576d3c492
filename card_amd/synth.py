"""Seeded random-init weights with the reference modules' state_dict layout (no network, no checkpoints: bench.py
and the oracle-only parity tests use these).  nn.Linear default init, gamma tables U(0,1) (regression/model.py:12),
the output layer scaled x0.1 so free-running trajectories stay O(1)."""
import math

import torch

from .diffusion_utils import make_beta_schedule


def _linear(sd, gen, name, i, o, scale=1.0):
    b = 1.0 / math.sqrt(i)
    sd[name + ".weight"] = (torch.rand(o, i, generator=gen) * 2 - 1) * b * scale
    sd[name + ".bias"] = (torch.rand(o, generator=gen) * 2 - 1) * b * scale


def guided_state_dict(x_dim, T, seed=1234, *, y_dim=1, hidden=128, cat_x=True, cat_y_pred=True):
    """regression/model.py:21 ConditionalGuidedModel."""
    gen = torch.Generator().manual_seed(seed)
    d_in = y_dim + (y_dim if cat_y_pred else 0) + (x_dim if cat_x else 0)
    sd = {}
    for k, i in (("lin1", d_in), ("lin2", hidden), ("lin3", hidden)):
        _linear(sd, gen, k + ".lin", i, hidden)
        sd[k + ".embed.weight"] = torch.rand(T + 1, hidden, generator=gen)
    _linear(sd, gen, "lin4", hidden, y_dim, 0.1)
    return sd


def cls_state_dict(F, C, T, seed, guidance=True):
    """classification/model.py:40 ConditionalModel theta part (lin1..lin4, unetnorm1..3 with non-trivial statistics)."""
    gen = torch.Generator().manual_seed(seed)
    sd = {}
    for k, i in (("lin1", 2 * C if guidance else C), ("lin2", F), ("lin3", F)):
        _linear(sd, gen, k + ".lin", i, F)
        sd[k + ".embed.weight"] = torch.rand(T + 1, F, generator=gen)
    _linear(sd, gen, "lin4", F, C, 0.1)
    for i in (1, 2, 3):
        sd[f"unetnorm{i}.weight"] = torch.rand(F, generator=gen) * 0.6 + 0.7
        sd[f"unetnorm{i}.bias"] = torch.randn(F, generator=gen) * 0.2
        sd[f"unetnorm{i}.running_mean"] = torch.randn(F, generator=gen) * 0.3
        sd[f"unetnorm{i}.running_var"] = torch.rand(F, generator=gen) + 0.5
        sd[f"unetnorm{i}.num_batches_tracked"] = torch.tensor(0)
    return sd


def make_schedule(schedule="linear", T=1000, start=1e-4, end=0.02):
    """(alphas, alphas_bar_sqrt, one_minus_alphas_bar_sqrt) the way the reference's runner derives them from the beta
    schedule (Diffusion.__init__, regression/card_regression.py:39-50), on the CPU."""
    betas = make_beta_schedule(schedule, T, start, end).float()
    alphas = 1.0 - betas
    alphas_cumprod = alphas.cumprod(dim=0)
    omabs = torch.sqrt(1 - alphas_cumprod)
    if schedule == "cosine":
        omabs = omabs * 0.9999
    return alphas, torch.sqrt(alphas_cumprod), omabs
