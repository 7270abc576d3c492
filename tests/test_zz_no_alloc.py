"""card_sampler_run_static (include/card_hip.h): the entry point that allocates nothing, against card_sampler_run.

No counterpart in the reference; for callers that cannot allocate.  The kernels are card_sampler_run's own; what differs
is where their scratch lives (the library's per-GPU static array instead of the sampler's wave slots / the caller's
workspace), so the bar is bit-equality with card_sampler_run on the same inputs -- on two calls in a row, the second
smaller (the array keeps the first call's contents), for both back ends that use scratch."""
import pytest
import torch

from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import cls_state_dict, make_schedule
from tests.helpers import load_golden, norm_rel_err
from oracle import card_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cuda(t):
    return t.to(DEV)


def test_static_scratch_persistent_slot_kernel(monkeypatch):
    """Regression network, pre-tiled rows (n_z = 1) with the cooperative kernel off: the scratch-slot kernel <8, 1>,
    whose per-wave layer-1 slots then sit in the static array."""
    g = load_golden("reg_year_T50")                     # d_in = 92: 12 k-groups of hoisted layer-1 columns
    monkeypatch.setenv("CARD_AMD_COOP", "0")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    T, N0 = g["T"], g["x"].shape[0]
    for N in (9000, 1300):                              # second call smaller: stale slots of the first stay behind
        idx = torch.arange(N) % N0
        x, y0 = cuda(g["x"][idx] + 0.01 * torch.arange(N).reshape(-1, 1) / N), cuda(g["y0hat"][idx])
        kw = dict(seed=5, elem_offset=77, out_mode=_lib.OUT_SELECT, states=[0, T // 2, T])
        want = s.run(x, y0, y0, coef, T, **kw)
        got = s.run(x, y0, y0, coef, T, static_scratch=True, **kw)
        assert torch.equal(got, want), N
    # and against the oracle with injected noise, through the new entry point alone
    got = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), static_scratch=True)
    assert norm_rel_err(got.cpu(), g["traj"][-1]) <= 1e-4


def test_static_scratch_gemm_back_end():
    """Classification head at F = 256 (GEMM back end): activations in the static array, walked in chunks of 9 600 chains
    (what 32 MiB holds, cut to whole rows and GEMM tiles), against one-chunk runs on a caller workspace."""
    T, F, C, n_z = 6, 256, 10, 100
    sd = cls_state_dict(F, C, T, 3)
    s = Sampler.from_cls_state_dict(sd, device=DEV)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    gen = torch.Generator().manual_seed(4)
    for rows in (230, 41):                              # 23 000 chains = three chunks, then 4 100 = one
        xf = cuda(torch.randn(rows, F, generator=gen))
        y0 = cuda(torch.softmax(torch.randn(rows, C, generator=gen) * 2, 1))
        kw = dict(seed=9, elem_offset=1234, n_z=n_z, out_mode=_lib.OUT_ALL)
        want = s.run(xf, y0, y0, coef, T, **kw)
        got = s.run(xf, y0, y0, coef, T, static_scratch=True, **kw)
        assert got.shape == (T + 1, rows * n_z, C) and torch.equal(got, want), rows


def test_static_scratch_limit_is_an_error_not_a_fault():
    """One input row's n_z chains must fit the array: F = 4096 holds 1 004 chains, n_z = 2000 does not."""
    T, F, C = 4, 4096, 10
    s = Sampler.from_cls_state_dict(cls_state_dict(F, C, T, 3), device=DEV)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    xf, y0 = torch.randn(2, F, device=DEV), torch.softmax(torch.randn(2, C, device=DEV), 1)
    with pytest.raises(_lib.CardError, match="static scratch"):
        s.run(xf, y0, y0, coef, T, seed=1, n_z=2000, static_scratch=True)
    ok = s.run(xf, y0, y0, coef, T, seed=1, n_z=500, static_scratch=True)
    assert torch.equal(ok, s.run(xf, y0, y0, coef, T, seed=1, n_z=500))
