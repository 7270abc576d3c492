"""Host-side owner of a packed eps_theta network (card_sampler handle) and the run entry points."""
import ctypes as C
import weakref

import torch

from . import _lib
from ._lib import CardError


def _dev_f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(device):
    if device.type != "cuda":
        raise CardError(
            f"card_amd runs on an AMD GPU only (model is on '{device}'); there is no CPU fallback. "
            "Move the model and inputs to cuda.")


def make_coef_table(alphas, one_minus_alphas_bar_sqrt, T=None):
    """(T, 8) fp32 CPU tensor: card_make_coef_table (regression/diffusion_utils.py:71-80, :88)."""
    a = alphas.detach().to("cpu", torch.float32).contiguous().reshape(-1)
    s = one_minus_alphas_bar_sqrt.detach().to("cpu", torch.float32).contiguous().reshape(-1)
    T = int(a.shape[0] if T is None else T)
    if a.shape[0] < T or s.shape[0] < T:
        raise CardError(f"alphas / one_minus_alphas_bar_sqrt have fewer than n_steps={T} entries")
    out = torch.empty(T, _lib.COEF_COLS, dtype=torch.float32)
    fp = C.POINTER(C.c_float)
    _lib.check(_lib.lib().card_make_coef_table(C.cast(a.data_ptr(), fp), C.cast(s.data_ptr(), fp), T,
                                               C.cast(out.data_ptr(), fp)), "card_make_coef_table")
    return out


_coef_cache = {}


def coef_on_device(alphas, omabs, T, device):
    """The (T, 8) coefficient table of a schedule on `device`, cached per schedule TENSOR OBJECT (the runner keeps
    `self.alphas` / `self.one_minus_alphas_bar_sqrt` for its lifetime, regression/card_regression.py:42-48).  An entry
    holds the two tensors it was computed from, so their ids / addresses cannot be recycled while it lives; a fresh
    tensor (even with equal values at a reused address) is a miss and gets its own table."""
    key = (id(alphas), id(omabs), int(T), str(device))
    hit = _coef_cache.get(key)
    if hit is not None and hit[0] is alphas and hit[1] is omabs and hit[2] == (alphas._version, omabs._version):
        return hit[3]
    if len(_coef_cache) > 16:
        _coef_cache.clear()
    tab = make_coef_table(alphas, omabs, T).to(device)
    _coef_cache[key] = (alphas, omabs, (alphas._version, omabs._version), tab)
    return tab


class Sampler:
    """A packed eps_theta network on one GPU.  Immutable after construction."""

    def __init__(self, *, variant, x_dim, y_dim, hidden, n_embed, cat_x, cat_y_pred, W, b, gamma, W4, b4,
                 bn_scale=None, bn_shift=None, device):
        device = torch.device(device)
        _require_cuda(device)
        L = _lib.lib()
        self.device, self.variant = device, variant
        self.x_dim, self.y_dim, self.hidden, self.n_embed = int(x_dim), int(y_dim), int(hidden), int(n_embed)
        self.cat_x, self.cat_y_pred = bool(cat_x), bool(cat_y_pred)
        keep = []

        def dev(t):
            t = _dev_f32(t, device)
            keep.append(t)
            return t

        d = _lib.ModelDesc()
        d.struct_bytes = C.sizeof(_lib.ModelDesc)
        d.variant, d.x_dim, d.y_dim, d.hidden, d.n_embed = variant, self.x_dim, self.y_dim, self.hidden, self.n_embed
        d.cat_x, d.cat_y_pred = int(self.cat_x), int(self.cat_y_pred)
        for i in range(len(W)):
            d.W[i] = dev(W[i]).data_ptr()
            d.b[i] = dev(b[i]).data_ptr()
            d.gamma[i] = dev(gamma[i]).data_ptr()
            if bn_scale is not None:
                d.bn_scale[i] = dev(bn_scale[i]).data_ptr()
                d.bn_shift[i] = dev(bn_shift[i]).data_ptr()
        d.W4, d.b4 = dev(W4).data_ptr(), dev(b4).data_ptr()
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(L.card_sampler_create(C.byref(d), _stream(device), C.byref(h)), "card_sampler_create")
            # the pack kernels read `keep` asynchronously on the current stream; the caching allocator is
            # stream-ordered, so releasing them after this point is safe
        self._h = h
        self.workspace_limit = None      # bytes; set to bound the GEMM back end's workspace (runs then go in chunks)
        self._ws = None                  # (tensor, aligned ptr, bytes): grown on demand, reused by later calls
        self._slot_cache = {}            # CARD_OUT_SELECT state -> slot tables on the device
        self._destroy = L.card_sampler_destroy
        weakref.finalize(self, L.card_sampler_destroy, h)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, sd, *, cat_x=True, cat_y_pred=True, device):
        """Regression ConditionalGuidedModel weights (regression/model.py:21-35) from a state_dict."""
        W1 = sd["lin1.lin.weight"]
        y_dim = int(sd["lin4.weight"].shape[0])
        d_in = int(W1.shape[1])
        x_dim = d_in - y_dim - (y_dim if cat_y_pred else 0) if cat_x else 0
        return cls(variant=_lib.VARIANT_GUIDED, x_dim=x_dim, y_dim=y_dim, hidden=int(W1.shape[0]),
                   n_embed=int(sd["lin1.embed.weight"].shape[0]), cat_x=cat_x, cat_y_pred=cat_y_pred,
                   W=[sd[f"lin{i}.lin.weight"] for i in (1, 2, 3)], b=[sd[f"lin{i}.lin.bias"] for i in (1, 2, 3)],
                   gamma=[sd[f"lin{i}.embed.weight"] for i in (1, 2, 3)], W4=sd["lin4.weight"], b4=sd["lin4.bias"],
                   device=device)

    @classmethod
    def from_cls_state_dict(cls, sd, *, guidance=True, bn_eps=1e-5, device):
        """Classification ConditionalModel theta weights (classification/model.py:67-77): lin1..lin4 plus the
        eval-mode BatchNorm1d layers unetnorm1..3 folded to per-feature (scale, shift).  `x` of run() is then
        x_feat = norm(encoder_x(x)) (classification/model.py:80-81, step-invariant in eval mode)."""
        W1 = sd["lin1.lin.weight"]
        C, F = int(sd["lin4.weight"].shape[0]), int(W1.shape[0])
        scale, shift = [], []
        for i in (1, 2, 3):
            w, b = sd[f"unetnorm{i}.weight"].float(), sd[f"unetnorm{i}.bias"].float()
            mean, var = sd[f"unetnorm{i}.running_mean"].float(), sd[f"unetnorm{i}.running_var"].float()
            sc = w / torch.sqrt(var + bn_eps)
            scale.append(sc)
            shift.append(b - mean * sc)
        return cls(variant=_lib.VARIANT_BN_HADAMARD, x_dim=F, y_dim=C, hidden=F,
                   n_embed=int(sd["lin1.embed.weight"].shape[0]), cat_x=False, cat_y_pred=guidance,
                   W=[sd[f"lin{i}.lin.weight"] for i in (1, 2, 3)], b=[sd[f"lin{i}.lin.bias"] for i in (1, 2, 3)],
                   gamma=[sd[f"lin{i}.embed.weight"] for i in (1, 2, 3)], W4=sd["lin4.weight"], b4=sd["lin4.bias"],
                   bn_scale=scale, bn_shift=shift, device=device)

    @classmethod
    def from_imagenet_head_state_dict(cls, sd, *, guidance=True, device):
        """ImageNet theta head (classification/imagenet/card/builder.py:69-92): lin1, lin2 conditional, lin3 plain.
        `x` of run() is then linx(z) (N, F), the step-invariant feature projection."""
        W1 = sd["lin1.lin.weight"]
        C, F = int(sd["lin3.weight"].shape[0]), int(W1.shape[0])
        return cls(variant=_lib.VARIANT_LINEAR_HADAMARD, x_dim=F, y_dim=C, hidden=F,
                   n_embed=int(sd["lin1.embed.weight"].shape[0]), cat_x=False, cat_y_pred=guidance,
                   W=[sd["lin1.lin.weight"], sd["lin2.lin.weight"]], b=[sd["lin1.lin.bias"], sd["lin2.lin.bias"]],
                   gamma=[sd["lin1.embed.weight"], sd["lin2.embed.weight"]], W4=sd["lin3.weight"], b4=sd["lin3.bias"],
                   device=device)

    @classmethod
    def from_description(cls, d, *, device=None):
        """Pack what `describe(model)` found (the recognition is device-free; this is where the weights reach the GPU)."""
        t = d.tensors
        nl = 2 if d.variant == _lib.VARIANT_LINEAR_HADAMARD else 3
        scale = shift = None
        if d.variant == _lib.VARIANT_BN_HADAMARD:
            scale, shift = [], []
            for i in (1, 2, 3):          # eval-mode BatchNorm1d as a per-feature affine map (classification/model.py:85, :89, :92)
                sc = t[f"bn{i}.weight"].detach().float() / torch.sqrt(t[f"bn{i}.running_var"].detach().float() + d.bn_eps)
                scale.append(sc)
                shift.append(t[f"bn{i}.bias"].detach().float() - t[f"bn{i}.running_mean"].detach().float() * sc)
        device = device if device is not None else t["W1"].device
        return cls(variant=d.variant, x_dim=d.x_dim, y_dim=d.y_dim, hidden=d.hidden, n_embed=d.n_embed, cat_x=d.cat_x,
                   cat_y_pred=d.cat_y_pred, W=[t[f"W{i}"] for i in range(1, nl + 1)], b=[t[f"b{i}"] for i in range(1, nl + 1)],
                   gamma=[t[f"gamma{i}"] for i in range(1, nl + 1)], W4=t["W4"], b4=t["b4"], bn_scale=scale, bn_shift=shift,
                   device=device)

    def _workspace(self, N):
        """(tensor, ptr, bytes) scratch for N chains: card_sampler_workspace_bytes (0 bytes on the persistent kernel;
        bounded on the GEMM back end, which walks larger batches in chunks)."""
        nb = C.c_uint64()
        _lib.check(_lib.lib().card_sampler_workspace_bytes(self._h, N, C.byref(nb)), "card_sampler_workspace_bytes")
        if nb.value == 0:
            return None, C.c_void_p(0), 0
        nbytes = nb.value if self.workspace_limit is None else min(nb.value, int(self.workspace_limit))
        # one buffer per sampler, kept between calls (calls on one sampler must not overlap anyway, card_hip.h) and
        # replaced only when a call needs more: the reference's per-timestep p_sample calls then allocate nothing
        if self._ws is None or self._ws[2] < nbytes or (self.workspace_limit is not None and self._ws[2] > nbytes):
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._ws = (ws, C.c_void_p((ws.data_ptr() + 255) & ~255), nbytes)
        return self._ws[0], self._ws[1], nbytes

    # ------------------------------------------------------------------------------------------
    def _prep(self, t, cols, name):
        if t is None:
            return None
        if not (t.dtype is torch.float32 and t.device == self.device and t.is_contiguous() and not t.requires_grad):
            t = _dev_f32(t, self.device)
        if t.dim() != 2 or t.shape[1] != cols:
            raise CardError(f"{name}: expected shape (N, {cols}), got {tuple(t.shape)}")
        return t

    def run(self, x, y_0_hat, y_T_mean, coef, T, *, t_hi=None, t_lo=0, y_init=None, noise=None, seed=0,
            elem_offset=0, out_mode=_lib.OUT_LAST, states=None, n_z=1, static_scratch=False, fphi=None, fphi_out=None,
            arith=_lib.ARITH_F32_MFMA, plan=False):
        """card_sampler_run.  Returns (N, y_dim) for OUT_LAST, (n_states, N, y_dim) otherwise.
        plan=True launches nothing and returns what card_sampler_plan reports for exactly this call: a dict with the
        kernel that carries the flops, its launches, all launches, algorithmic and executed flops per sample-step.
        n_z > 1: x / y_0_hat / y_T_mean are per test ROW and every row is expanded to n_z chains inside the
        kernel (N = rows * n_z, chain order (row, z) like regression/card_regression.py:1027-1035).
        static_scratch=True goes through card_sampler_run_static: the kernels' scratch comes from the library's per-GPU
        static array, no workspace is passed (the output tensor is still torch's).
        fphi = a card_amd.guidance.GuidanceNet: y_0_hat = f_phi(x) is computed inside the call (first kernel) into
        `fphi_out` ((rows, y_dim), allocated when None; kept as self.last_y_0_hat); y_0_hat is ignored and y_T_mean=None
        means y_T_mean = y_0_hat (regression/card_regression.py:1021-1035).
        arith = _lib.ARITH_SPLIT_BF16: the hidden layers on the bf16 matrix pipe at fp32 accuracy (card_arith in
        include/card_hip.h; regression samplers at hidden 128 only)."""
        T = int(T)
        t_hi = T - 1 if t_hi is None else int(t_hi)
        t_lo = int(t_lo)
        n_z = max(1, int(n_z))
        if fphi is not None:
            if self.variant != _lib.VARIANT_GUIDED:
                raise CardError("fphi is for regression (GUIDED) samplers")
            xx = self._prep(x, fphi.dim_in, "x")
            rows = int(xx.shape[0])
            if fphi_out is None:
                fphi_out = torch.empty(rows, self.y_dim, device=self.device, dtype=torch.float32)
            elif not (fphi_out.dtype is torch.float32 and fphi_out.device == self.device and fphi_out.is_contiguous()
                      and tuple(fphi_out.shape) == (rows, self.y_dim)):
                raise CardError(f"fphi_out must be a contiguous fp32 ({rows}, {self.y_dim}) tensor on {self.device}")
            y0 = self.last_y_0_hat = fphi_out
            mu = self._prep(y_T_mean, self.y_dim, "y_T_mean") if y_T_mean is not None else None
        else:
            y0 = self._prep(y_0_hat, self.y_dim, "y_0_hat")
            mu = self._prep(y_T_mean, self.y_dim, "y_T_mean")
            rows = int(y0.shape[0])
            need_x = self.variant != _lib.VARIANT_GUIDED or (self.cat_x and self.x_dim > 0)
            xx = self._prep(x, self.x_dim, "x") if need_x else None
        N = rows * n_z
        yi = self._prep(y_init, self.y_dim, "y") if y_init is not None else None
        for name, t, want in (("x", xx, rows), ("y_T_mean", mu, rows), ("y", yi, N)):
            if t is not None and t.shape[0] != want:
                raise CardError(f"{name} has {t.shape[0]} rows, expected {want}")
        n_steps = t_hi - t_lo + 1
        n_states = n_steps + (0 if yi is not None else 1)
        n_draws = (0 if yi is not None else 1) + (t_hi - max(t_lo, 1) + 1 if t_hi >= 1 else 0)
        nz = None
        if noise is not None:
            nz = _dev_f32(noise, self.device)
            if nz.numel() != n_draws * N * self.y_dim:
                raise CardError(f"noise: expected {n_draws} x {N} x {self.y_dim} values, got {tuple(nz.shape)}")
        slot = None
        if out_mode == _lib.OUT_LAST:
            out = torch.empty(N, self.y_dim, device=self.device, dtype=torch.float32)
        elif out_mode == _lib.OUT_ALL:
            out = torch.empty(n_states, N, self.y_dim, device=self.device, dtype=torch.float32)
        else:
            states = tuple(int(s) for s in states)
            # states this call produces: s = T - t for the processed timesteps, plus y_T (s = 0) when the loop starts here
            first = T - t_hi - (0 if yi is not None else 1)
            slot = self._slot_cache.get((T, states))
            if slot is None:
                if len(set(states)) != len(states):
                    raise CardError(f"states {list(states)} lists a state twice (each output slot is written once)")
                table = torch.full((T + 1,), -1, dtype=torch.int32)
                for i, s in enumerate(states):
                    if not 0 <= s <= T:
                        raise CardError(f"state index {s} outside [0, {T}]")
                    table[s] = i
                if len(self._slot_cache) > 64:
                    self._slot_cache.clear()
                slot = self._slot_cache[(T, states)] = table.to(self.device)
            bad = [s for s in states if not first <= s <= T - t_lo]
            if bad:
                raise CardError(f"states {bad} are not produced by timesteps {t_hi}..{t_lo} of T={T} "
                                f"(this call produces states {first}..{T - t_lo}); their slots would stay unwritten")
            out = torch.empty(len(states), N, self.y_dim, device=self.device, dtype=torch.float32)
        a = _lib.RunArgs()
        a.struct_bytes = C.sizeof(_lib.RunArgs)
        a.T, a.t_hi, a.t_lo, a.N = T, t_hi, t_lo, N
        a.x, a.y_0_hat, a.y_T_mean, a.y_init = _ptr(xx), _ptr(y0), _ptr(mu), _ptr(yi)
        a.coef, a.noise = _ptr(coef), _ptr(nz)
        a.seed, a.elem_offset = int(seed) & (2 ** 64 - 1), int(elem_offset)
        a.out, a.out_mode, a.state_slot, a.n_z = _ptr(out), out_mode, _ptr(slot), n_z
        a.arith = int(arith)
        if fphi is not None:
            a.fphi, a.fphi_out = fphi._h, _ptr(fphi_out)
        L = _lib.lib()
        if static_scratch:
            fn, what = L.card_sampler_run_static, "card_sampler_run_static"
        else:
            fn, what = L.card_sampler_run, "card_sampler_run"
            ws, a.workspace, a.workspace_bytes = self._workspace(N)
        if plan:
            pl = _lib.Plan()
            pl.struct_bytes = C.sizeof(_lib.Plan)
            _lib.check(L.card_sampler_plan(self._h, C.byref(a), C.byref(pl)), "card_sampler_plan")
            return {"kernel": pl.kernel.decode(), "launches": pl.dominant_launches, "launches_all_kernels": pl.launches,
                    "backend": pl.backend, "flops_per_sample_step": pl.flops_per_sample_step,
                    "flops_executed_per_sample_step": pl.flops_executed_per_sample_step}
        if torch.cuda.current_device() == self.device.index:
            _lib.check(fn(self._h, C.byref(a), _stream(self.device)), what)
        else:
            with torch.cuda.device(self.device):
                _lib.check(fn(self._h, C.byref(a), _stream(self.device)), what)
        return out

    def eps_theta(self, x, y_t, y_0_hat, t):
        """model(x, y_t, y_0_hat, t) of the packed network: card_eps_theta."""
        y0 = self._prep(y_0_hat, self.y_dim, "y_0_hat")
        yt = self._prep(y_t, self.y_dim, "y_t")
        need_x = self.variant != _lib.VARIANT_GUIDED or (self.cat_x and self.x_dim > 0)
        xx = self._prep(x, self.x_dim, "x") if need_x else None
        N = int(y0.shape[0])
        out = torch.empty(N, self.y_dim, device=self.device, dtype=torch.float32)
        ws, wptr, wbytes = self._workspace(N)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().card_eps_theta(self._h, _ptr(xx), _ptr(yt), _ptr(y0), N, int(t), _ptr(out),
                                                 wptr, wbytes, _stream(self.device)), "card_eps_theta")
        return out


# ------------------------------------------------------------------------------------------------
# model -> Sampler, cached per model object.  A cached sampler is reused only while the model's weights are what it
# was packed from.  Two checks:
#   * identity key: (id, data_ptr, _version) of every parameter and buffer -- catches optimizer steps, load_state_dict,
#     in-place ops on the parameters, replaced parameter objects.  Cheap (no device work), evaluated on every call.
#   * content hash: card_content_hash over the same tensors (one launch + one 8-byte read-back) -- catches what PyTorch's
#     version counters do not see, notably the reference's own EMA copy `param.data.copy_(shadow)` (regression/ema.py:21,
#     applied before testing at regression/card_regression.py:934-940).  Evaluated at every whole-loop entry
#     (p_sample_loop, p_sample_loop_with_eval) and, for the per-timestep entry points, on the first step of a loop
#     (t = T-1, where `for t in reversed(range(1, T))` starts) -- a model edited through `.data` BETWEEN two p_sample
#     calls of one loop is the one case left to `invalidate(model)`.
# ------------------------------------------------------------------------------------------------
_cache = weakref.WeakKeyDictionary()          # keyed on the CORE module (the one whose tensors are packed), not on a wrapper


class ModelDescription:
    """What `describe(model)` found: which eps_theta network of the reference `model` is, its dimensions and flags, and
    the tensors to pack, in a fixed order.  Nothing here touches a device, so recognition and weight extraction can be
    tested against the reference's real classes in a container without a GPU (tests/test_reference_classes_cpu.py).

    kind      "guided" (regression/model.py:21), "bn_hadamard" (classification/model.py:22) or "linear_hadamard"
              (classification/imagenet/card/builder.py:69)
    core      the module whose parameters and buffers the sampler depends on: `model` itself with DistributedDataParallel /
              DataParallel holders removed, or the `model_theta` of the ImageNet CARD wrapper (builder.py:8; its frozen
              model_phi backbone is not part of the sampling path and is neither packed nor hashed)
    tensors   ordered name -> tensor (the module's own Parameter / buffer objects): W1 b1 gamma1 ... W4 b4
              [+ bn{i}.weight / .bias / .running_mean / .running_var]; for the ImageNet head W4 / b4 are its lin3
    features  how the kernels' `x` is obtained from the caller's x: None (x itself), "norm(encoder_x(x))" or "linx(z)"
    """
    __slots__ = ("kind", "variant", "core", "x_dim", "y_dim", "hidden", "n_embed", "cat_x", "cat_y_pred", "bn_eps",
                 "tensors", "features", "wrapped")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw[k])

    @property
    def d_in(self):
        """columns of lin1: y_t, then y_0_hat when cat_y_pred, then x when cat_x (regression/model.py:37-47)"""
        return int(self.tensors["W1"].shape[1])

    def __repr__(self):
        return (f"ModelDescription({self.kind}, x_dim={self.x_dim}, y_dim={self.y_dim}, hidden={self.hidden}, "
                f"n_embed={self.n_embed}, cat_x={self.cat_x}, cat_y_pred={self.cat_y_pred}, features={self.features!r}, "
                f"wrapped={self.wrapped})")


def unwrap(model):
    """`model` without its DistributedDataParallel / DataParallel holders: classification/imagenet/main.py:222-237 wraps the
    CARD model before handing it to p_sample_loop (:482, :557) and itself reaches in with `model.module` (:357).  Any
    other holder that only carries a `.module` nn.Module (apex DDP, a bare container) is unwrapped the same way when the
    holder itself has none of the eps_theta attributes."""
    import torch.nn as nn
    for _ in range(4):
        inner = getattr(model, "module", None)
        if not isinstance(inner, nn.Module):
            break
        if isinstance(model, (nn.parallel.DistributedDataParallel, nn.DataParallel)) or not hasattr(model, "lin1"):
            model = inner
        else:
            break
    return model


def _tensors(core):
    import itertools
    return list(itertools.chain(core.parameters(), core.buffers()))


def _fingerprint(core):
    return tuple((id(p), p.data_ptr(), p._version) for p in _tensors(core))


def _content_hash(core):
    """card_content_hash of every parameter and buffer (device-side, one launch; reads back 8 bytes)."""
    ts = [p.detach() for p in _tensors(core) if p.numel() > 0]
    ts = [t if t.is_contiguous() else t.contiguous() for t in ts]
    if not ts or ts[0].device.type != "cuda":
        return 0
    dev = ts[0].device
    n = len(ts)
    L = _lib.lib()
    out = torch.empty(1, dtype=torch.int64, device=dev)
    total = 0
    with torch.cuda.device(dev):
        for lo in range(0, n, 32):                      # the entry point takes at most 32 arrays per call
            part = ts[lo:lo + 32]
            ptrs = (C.c_void_p * len(part))(*[t.data_ptr() for t in part])
            words = (C.c_int64 * len(part))(*[t.numel() * t.element_size() // 4 for t in part])
            _lib.check(L.card_content_hash(ptrs, words, len(part), _ptr(out), _stream(dev)), "card_content_hash")
            total = (total * 1000003 + int(out.item())) & (2 ** 64 - 1)
    return total


def _is_cond_linear(m):
    # ConditionalLinear (regression/model.py:6, classification/model.py:7, imagenet/card/builder.py:54): .lin + .embed
    return hasattr(m, "lin") and hasattr(m, "embed") and hasattr(m.lin, "weight") and hasattr(m.embed, "weight")


def _is_guided(model):
    # regression/model.py:21 ConditionalGuidedModel, recognised by structure (duck-typed `model` contract)
    return all(hasattr(model, n) for n in ("lin1", "lin2", "lin3", "lin4", "cat_x", "cat_y_pred")) and \
        _is_cond_linear(model.lin1) and not hasattr(model, "unetnorm1")


def _is_cls(model):
    # classification/model.py:22 ConditionalModel: lin1..lin4 + unetnorm1..3 + encoder_x + norm
    return all(hasattr(model, n) for n in ("lin1", "lin2", "lin3", "lin4", "unetnorm1", "unetnorm2", "unetnorm3",
                                           "encoder_x", "norm")) and _is_cond_linear(model.lin1)


def _imagenet_theta(model):
    # classification/imagenet/card/builder.py: the CARD wrapper (:8, model_theta attribute) or the bare head (:69)
    theta = getattr(model, "model_theta", model)
    ok = all(hasattr(theta, n) for n in ("linx", "lin1", "lin2", "lin3")) and not hasattr(theta, "lin4") and \
        _is_cond_linear(theta.lin1) and _is_cond_linear(theta.lin2) and not _is_cond_linear(theta.lin3)
    return theta if ok else None


def _shape_check(what, t, want):
    if tuple(t.shape) != tuple(want):
        raise CardError(f"{what}: shape {tuple(t.shape)}, expected {tuple(want)}")


def describe(model):
    """Recognise `model` (device-free) and list what the sampler packs from it: a ModelDescription.  Raises CardError
    for anything that is not one of the reference's three eps_theta networks (there is no eager fallback)."""
    inner = unwrap(model)
    wrapped = inner is not model
    t = {}

    def cond(i, m):
        t[f"W{i}"], t[f"b{i}"], t[f"gamma{i}"] = m.lin.weight, m.lin.bias, m.embed.weight

    if _is_guided(inner):
        core, kind, variant, features, bn_eps = inner, "guided", _lib.VARIANT_GUIDED, None, None
        for i in (1, 2, 3):
            cond(i, getattr(core, f"lin{i}"))
        t["W4"], t["b4"] = core.lin4.weight, core.lin4.bias
        cat_x, cat_y_pred = bool(core.cat_x), bool(core.cat_y_pred)
        y_dim, hidden, d_in = int(t["W4"].shape[0]), int(t["W1"].shape[0]), int(t["W1"].shape[1])
        x_dim = d_in - y_dim - (y_dim if cat_y_pred else 0) if cat_x else 0
        if x_dim < 0 or (not cat_x and d_in != y_dim * (2 if cat_y_pred else 1)):
            raise CardError(f"lin1 takes {d_in} inputs, which does not fit y_dim={y_dim}, cat_x={cat_x}, cat_y_pred={cat_y_pred} "
                            "(regression/model.py:27-32)")
    elif _imagenet_theta(inner) is not None:
        core, kind, variant, features, bn_eps = _imagenet_theta(inner), "linear_hadamard", _lib.VARIANT_LINEAR_HADAMARD, "linx(z)", None
        wrapped = wrapped or core is not inner
        cond(1, core.lin1)
        cond(2, core.lin2)
        t["W4"], t["b4"] = core.lin3.weight, core.lin3.bias
        cat_x, cat_y_pred = False, bool(getattr(core, "guidance", True))
        y_dim, hidden = int(t["W4"].shape[0]), int(t["W1"].shape[0])
        x_dim = hidden
        _shape_check("linx.weight", core.linx.weight, (hidden, core.linx.weight.shape[1]))
    elif _is_cls(inner):
        core, kind, variant, features = inner, "bn_hadamard", _lib.VARIANT_BN_HADAMARD, "norm(encoder_x(x))"
        for i in (1, 2, 3):
            cond(i, getattr(core, f"lin{i}"))
            bn = getattr(core, f"unetnorm{i}")
            if getattr(bn, "running_mean", None) is None or getattr(bn, "weight", None) is None:
                raise CardError(f"unetnorm{i} has no running statistics / affine parameters to fold (classification/model.py:72-76 "
                                "builds nn.BatchNorm1d with both)")
            t[f"bn{i}.weight"], t[f"bn{i}.bias"] = bn.weight, bn.bias
            t[f"bn{i}.running_mean"], t[f"bn{i}.running_var"] = bn.running_mean, bn.running_var
        t["W4"], t["b4"] = core.lin4.weight, core.lin4.bias
        eps = {float(getattr(core, f"unetnorm{i}").eps) for i in (1, 2, 3)}
        if len(eps) != 1:
            raise CardError(f"unetnorm1..3 use different eps values {sorted(eps)}")
        bn_eps = eps.pop()
        cat_x, cat_y_pred = False, bool(getattr(core, "guidance", True))
        y_dim, hidden = int(t["W4"].shape[0]), int(t["W1"].shape[0])
        x_dim = hidden
    else:
        raise CardError(f"card_amd does not recognise {type(inner).__name__} as a CARD eps_theta network "
                        "(expected regression ConditionalGuidedModel, classification ConditionalModel or the ImageNet head, "
                        "bare or inside DistributedDataParallel / DataParallel); there is no eager fallback")
    n_embed = int(t["gamma1"].shape[0])
    nl = 2 if variant == _lib.VARIANT_LINEAR_HADAMARD else 3
    if variant != _lib.VARIANT_GUIDED:
        _shape_check("lin1.lin.weight", t["W1"], (hidden, y_dim * (2 if cat_y_pred else 1)))
    for i in range(1, nl + 1):
        if i > 1:
            _shape_check(f"lin{i}.lin.weight", t[f"W{i}"], (hidden, hidden))
        _shape_check(f"lin{i}.lin.bias", t[f"b{i}"], (hidden,))
        _shape_check(f"lin{i}.embed.weight", t[f"gamma{i}"], (n_embed, hidden))
        if variant == _lib.VARIANT_BN_HADAMARD:
            for k in ("weight", "bias", "running_mean", "running_var"):
                _shape_check(f"unetnorm{i}.{k}", t[f"bn{i}.{k}"], (hidden,))
    _shape_check("output layer weight", t["W4"], (y_dim, hidden))
    _shape_check("output layer bias", t["b4"], (y_dim,))
    order = [f"{p}{i}" for i in range(1, nl + 1) for p in ("W", "b", "gamma")] + ["W4", "b4"] + \
        [f"bn{i}.{k}" for i in (1, 2, 3) for k in ("weight", "bias", "running_mean", "running_var") if f"bn{i}.{k}" in t]
    return ModelDescription(kind=kind, variant=variant, core=core, x_dim=x_dim, y_dim=y_dim, hidden=hidden, n_embed=n_embed,
                            cat_x=cat_x, cat_y_pred=cat_y_pred, bn_eps=bn_eps, tensors={k: t[k] for k in order},
                            features=features, wrapped=wrapped)


def core_of(model):
    """The module the caches are keyed on (see ModelDescription.core); cheap enough for every call."""
    inner = unwrap(model)
    theta = getattr(inner, "model_theta", None)
    if theta is not None and hasattr(theta, "linx"):
        return theta
    return inner


def invalidate(model):
    """Forget the packed sampler (and hoisted features) of `model`: the next call re-packs from the current weights."""
    core = core_of(model)
    _cache.pop(core, None)
    _xfeat_cache.pop(core, None)


_xfeat_cache = weakref.WeakKeyDictionary()


def x_features(model, x):
    """x_feat = norm(encoder_x(x)) (classification/model.py:80-81) or linx(z) (imagenet/card/builder.py:90), computed
    ONCE per x tensor object instead of once per timestep: in eval mode it does not depend on t or y.  The cache entry
    is tied to the x OBJECT through a weak reference (an address alone can be recycled by the caching allocator for the
    next batch) and dies with it, so the features are not pinned beyond the caller's own x.  `model` may be wrapped
    (DistributedDataParallel / DataParallel / the ImageNet CARD wrapper): the inner module is used, like the reference's
    own `model.module.model_theta` (imagenet/main.py:357).
    Stock PyTorch-ROCm (out of scope to fuse)."""
    core = core_of(model)
    hit = _xfeat_cache.get(core)
    if hit is not None and hit[0]() is x and hit[1] == (x._version, _cache.get(core, (None,))[0]):
        return hit[2]
    with torch.no_grad():
        xf = (core.linx(x) if hasattr(core, "linx") else core.norm(core.encoder_x(x))).float().contiguous()
    if xf is x:                  # identity encoder: nothing was computed, and caching x under its own weak key would pin it
        return xf
    mref = weakref.ref(core)

    def _drop(_, mref=mref):
        m = mref()
        if m is not None:
            _xfeat_cache.pop(m, None)

    _xfeat_cache[core] = (weakref.ref(x, _drop), (x._version, _cache.get(core, (None,))[0]), xf)
    return xf


def sampler_for(model, check_content=True):
    """The cached Sampler of an eps_theta `nn.Module` (raises for architectures the kernels do not cover).
    check_content=False skips the content hash (per-timestep callers inside one loop)."""
    core = core_of(model)
    if getattr(core, "training", False) and hasattr(core, "unetnorm1"):
        raise CardError("card_amd samples a BatchNorm model in eval mode only: in train mode BatchNorm normalises with "
                        "batch statistics and the folded (scale, shift) would be wrong (call model.eval(); the "
                        "reference tests under model.eval(), classification/card_classification.py:873)")
    fp = _fingerprint(core)
    hit = _cache.get(core)
    if hit is not None and hit[0] == fp:
        if not check_content:
            return hit[1]
        h = _content_hash(core)
        if h == hit[2]:
            return hit[1]
    else:
        h = None
    d = describe(model)
    assert d.core is core
    # no train/eval-dependent layer in regression/model.py:21-51: the reference samples from it in train mode too
    # (validation inside train(), regression/card_regression.py:365 -> :457, :533)
    device = d.tensors["W1"].device
    _require_cuda(device)
    s = Sampler.from_description(d, device=device)
    _cache[core] = (fp, s, _content_hash(core) if h is None else h)
    _xfeat_cache.pop(core, None)
    return s
