"""GPU tests of the streaming-GEMM back end beyond the classification fixtures: regression-style
ConditionalGuidedModel networks at shapes the persistent kernel is not built for (hidden 256 of BASELINE's
128/256 sweep, y_dim > 1) against the CPU oracle, and chunked runs (bounded workspace) against unchunked ones."""
import pytest
import torch

from card_amd import _lib
from card_amd._lib import CardError
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict
from oracle import card_oracle as orc
from tests.helpers import load_golden, norm_rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def cuda(t):
    return t.to(DEV)


@pytest.mark.parametrize("H,x_dim,y_dim,cat_x,cat_y_pred", [
    (256, 13, 1, True, True),      # BASELINE configs[4]: hidden 256 at the Boston input width
    (256, 90, 1, True, False),     # Year input width, no y_0_hat column
    (384, 5, 1, False, True),      # cat(y_t, y_0_hat) only
    (128, 7, 3, True, True),       # multi-output regression at the reference width
])
def test_guided_shapes_on_gemm_backend(H, x_dim, y_dim, cat_x, cat_y_pred):
    T, N = 12, 200
    sd = guided_state_dict(x_dim, T, 21, y_dim=y_dim, hidden=H, cat_x=cat_x, cat_y_pred=cat_y_pred)
    w = orc.GuidedWeights.from_state_dict(sd, cat_x=cat_x, cat_y_pred=cat_y_pred)
    gen = torch.Generator().manual_seed(22)
    x = torch.randn(N, x_dim, generator=gen)
    y0 = torch.randn(N, y_dim, generator=gen)
    mu = y0 + 0.1 * torch.randn(N, y_dim, generator=gen)
    noise = torch.randn(T, N, y_dim, generator=gen)
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    want = torch.stack(orc.p_sample_loop(w, x, y0, mu, T, alphas, omabs, noise), 0)
    s = Sampler.from_state_dict(sd, cat_x=cat_x, cat_y_pred=cat_y_pred, device=DEV)
    coef = cuda(make_coef_table(alphas, omabs))
    got = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, noise=cuda(noise), out_mode=_lib.OUT_ALL).cpu()
    worst = max(norm_rel_err(got[k], want[k]) for k in range(T + 1))
    print(f"H={H} d=({x_dim},{y_dim}) cat=({cat_x},{cat_y_pred}): worst per-state norm-rel err {worst:.3e}")
    assert worst <= TOL
    eps = s.eps_theta(cuda(x), cuda(want[3]), cuda(y0), 5).cpu()
    assert norm_rel_err(eps, orc.eps_theta(w, x, want[3], y0, 5)) <= TOL
    # in-kernel row tiling and RNG mode: equal to the same run on pre-tiled rows
    from card_amd.distributed import tile_rows
    a = s.run(cuda(x[:9]), cuda(y0[:9]), cuda(mu[:9]), coef, T, seed=3, n_z=20)
    b = s.run(cuda(tile_rows(x[:9], 20)), cuda(tile_rows(y0[:9], 20)), cuda(tile_rows(mu[:9], 20)), coef, T, seed=3)
    assert torch.equal(a, b)


@pytest.mark.parametrize("n_z", [1, 10, 96])
def test_chunked_run_equals_unchunked(n_z, monkeypatch):
    monkeypatch.setenv("CARD_AMD_CLSCOOP", "0")            # this F = 128 head on the GEMM back end
    g = load_golden("cls_mnist_T50")
    s = Sampler.from_cls_state_dict(g["sd"], device=DEV)
    T, C = g["T"], g["y0hat"].shape[1]
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    rows = 60
    idx = torch.arange(rows) % g["x_feat"].shape[0]
    xf, y0 = cuda(g["x_feat"][idx]), cuda(g["y0hat"][idx])
    N = rows * n_z
    full = s.run(xf, y0, y0, coef, T, seed=9, elem_offset=77, n_z=n_z, out_mode=_lib.OUT_ALL)
    z = torch.randn(T, N, C, generator=torch.Generator().manual_seed(1)).to(DEV)
    full_inj = s.run(xf, y0, y0, coef, T, noise=z, n_z=n_z, states=[0, 3, T], out_mode=_lib.OUT_SELECT)
    import ctypes
    nb = ctypes.c_uint64()
    _lib.check(_lib.lib().card_sampler_workspace_bytes(s._h, 1, ctypes.byref(nb)), "ws")
    per_chain = nb.value                                    # >= bytes per chain (256-B alignment of each buffer)
    s.workspace_limit = per_chain * max(3 * n_z, 300) // 2  # a few chunks, ragged last one
    try:
        part = s.run(xf, y0, y0, coef, T, seed=9, elem_offset=77, n_z=n_z, out_mode=_lib.OUT_ALL)
        part_inj = s.run(xf, y0, y0, coef, T, noise=z, n_z=n_z, states=[0, 3, T], out_mode=_lib.OUT_SELECT)
        last = s.run(xf, y0, y0, coef, T, seed=9, elem_offset=77, n_z=n_z)
        s.workspace_limit = 2048                            # less than one input row of chains
        with pytest.raises(CardError):
            s.run(xf, y0, y0, coef, T, seed=9, n_z=max(n_z, 2))
    finally:
        s.workspace_limit = None
    assert torch.equal(part, full) and torch.equal(part_inj, full_inj) and torch.equal(last, full[-1])


@pytest.mark.parametrize("F,C,rows,n_z", [(128, 1, 77, 1), (128, 3, 50, 7), (384, 8, 33, 3), (256, 9, 129, 2), (1024, 16, 40, 5),
                                          (256, 17, 31, 9), (128, 32, 65, 2)])
def test_fused_head_update_equals_gemm_head(F, C, rows, n_z, monkeypatch):
    """wide_head_update_kernel (lin4 + posterior update in one VALU launch, classes <= 32) against the head as a GEMM launch
    + wide_update_kernel (CARD_AMD_WIDE=gemmhead, read when a sampler is created): same states to rounding (the dot products are associated differently),
    y_T bit-equal, eps_theta to rounding; class counts at the edges of the kernel's 8 / 16 / 32 paddings, a feature width
    that is an odd multiple of 128, ragged row groups."""
    from card_amd.synth import cls_state_dict
    from card_amd.synth import make_schedule
    monkeypatch.setenv("CARD_AMD_CLSCOOP", "off")
    T = 12
    g = torch.Generator().manual_seed(C * 1000 + F)
    xf = cuda(torch.randn(rows, F, generator=g))
    y0 = cuda(torch.softmax(torch.randn(rows, C, generator=g) * 2, 1))
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    outs = {}
    for head in ("fused", "gemm"):
        monkeypatch.setenv("CARD_AMD_WIDE", "" if head == "fused" else "gemmhead")
        s = Sampler.from_cls_state_dict(cls_state_dict(F, C, T, 5), device=DEV)
        outs[head] = (s.run(xf, y0, y0, coef, T, seed=4, elem_offset=11, n_z=n_z, out_mode=_lib.OUT_ALL),
                      s.eps_theta(xf, y0, y0, 3))
    a, b = outs["fused"], outs["gemm"]
    assert torch.equal(a[0][0], b[0][0])
    assert torch.isfinite(a[0]).all()
    worst = max(norm_rel_err(a[0][k].cpu(), b[0][k].cpu()) for k in range(T + 1))
    print(f"F={F} C={C}: fused vs GEMM head, worst per-state norm-rel difference {worst:.2e}")
    assert worst <= 2e-5
    assert norm_rel_err(a[1].cpu(), b[1].cpu()) <= 2e-6


@pytest.mark.parametrize("F,rows,n_z", [(256, 300, 7), (1024, 1000, 3), (128, 4000, 1)])
def test_tail_tiles_change_no_bits(F, rows, n_z, monkeypatch):
    """The last, partly filled round of a GEMM launch runs as 64 x 64 tiles (wide_gemm_tile<.., 64, 64>): every output
    element is the same fp32 FMA chain over k as in a 128 x 128 tile, so CARD_AMD_WIDE=notail gives the same bits."""
    from card_amd.synth import cls_state_dict
    from card_amd.synth import make_schedule
    monkeypatch.setenv("CARD_AMD_CLSCOOP", "off")
    T, C = 10, 10
    g = torch.Generator().manual_seed(F)
    xf = cuda(torch.randn(rows, F, generator=g))
    y0 = cuda(torch.softmax(torch.randn(rows, C, generator=g) * 2, 1))
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    outs = []
    for tail in ("1", "0"):
        monkeypatch.setenv("CARD_AMD_WIDE", "" if tail == "1" else "notail")
        s = Sampler.from_cls_state_dict(cls_state_dict(F, C, T, 6), device=DEV)
        outs.append(s.run(xf, y0, y0, coef, T, seed=8, n_z=n_z, out_mode=_lib.OUT_ALL))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
