"""The time-sliced paired cooperative kernel (card_sample_coop2_kernel, card_guided.hip): between one and two 32-chain
tiles per CU (the reference's Year minibatch, 64 rows x 200 z = 12 800 chains, regression/configs/uci_year.yml:62,67)
the loop is cut into slices in which every CU is busy, with chain states handed over in global memory between the
launches.  Bar: bit-identical to the un-sliced kernels (CARD_AMD_COOP=no2, read when a sampler is created) in every output mode, and within 1e-4 of the
oracle under injected noise on slices of the batch."""
import pytest
import torch

from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from oracle import card_oracle as orc
from tests.helpers import load_golden, norm_rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def _tiles_per_cu(N):
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    return (N + 31) // 32 / cu


def _inputs(g, rows, seed):
    gen = torch.Generator().manual_seed(seed)
    x_dim = g["x"].shape[1]
    x = torch.randn(rows, x_dim, generator=gen)
    y0 = torch.randn(rows, 1, generator=gen)
    mu = y0 + 0.1 * torch.randn(rows, 1, generator=gen)
    return x.to(DEV), y0.to(DEV), mu.to(DEV)


@pytest.mark.parametrize("N", [12800, 9001, 8224, 14336])
def test_sliced_equals_unsliced_bit_for_bit(N, monkeypatch):
    assert 1.0 < _tiles_per_cu(N) < 2.0, "sized for a 256-CU MI355X"
    g = load_golden("reg_boston_T1000")
    T = 200                                   # the first 200 timesteps' schedule rows of the T = 1000 model
    coef = make_coef_table(g["alphas"][:T], g["omabs"][:T]).to(DEV)
    x, y0, mu = _inputs(g, N, 3)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CARD_AMD_COOP", "no16" if mode == "1" else "no16,no2")       # keep these sizes on the 32-chain kernels
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        last = s.run(x, y0, mu, coef, T, seed=21, elem_offset=7)
        sel = s.run(x, y0, mu, coef, T, seed=21, elem_offset=7, out_mode=_lib.OUT_SELECT, states=[0, 1, 57, 58, 120, T - 1, T])
        # a segment that starts in the middle of the loop from a given state (p_sample_loop_with_eval's segments)
        seg = s.run(x, y0, mu, coef, T, seed=21, elem_offset=7, t_hi=150, t_lo=20, y_init=sel[2])
        outs[mode] = (last, sel, seg)
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    assert torch.equal(outs["1"][0], outs["1"][1][-1])
    # the entry point that allocates nothing takes the hand-over buffer from the library's static array
    monkeypatch.setenv("CARD_AMD_COOP", "no16")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    assert torch.equal(s.run(x, y0, mu, coef, T, seed=21, elem_offset=7, static_scratch=True), outs["0"][0])


def test_sliced_full_trajectory_and_row_tiling_vs_oracle(monkeypatch):
    """OUT_ALL with injected noise, rows tiled in-kernel (64 rows x 200 z like uci_year.yml): every state equals the
    un-sliced run bit for bit, and the chains of the first and last rows match the oracle."""
    g = load_golden("reg_year_T1000")
    T, rows, n_z = 80, 64, 200
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.25)
    coef = make_coef_table(alphas, omabs).to(DEV)
    x, y0, mu = _inputs(g, rows, 5)
    N = rows * n_z
    noise = torch.randn(T, N, 1, generator=torch.Generator().manual_seed(9))
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("CARD_AMD_COOP", "" if mode == "1" else "no2")
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        outs.append(s.run(x, y0, mu, coef, T, noise=noise.to(DEV), out_mode=_lib.OUT_ALL, n_z=n_z))
    assert torch.equal(outs[0], outs[1])
    w = g["weights"]
    for r in (0, rows - 1):
        sl = slice(r * n_z, r * n_z + 40)
        tile = lambda t: t[r:r + 1].cpu().repeat(40, 1)
        want = torch.stack(orc.p_sample_loop(w, tile(x), tile(y0), tile(mu), T, alphas, omabs, noise[:, sl]), 0)
        err = max(norm_rel_err(outs[0][k, sl].cpu(), want[k]) for k in range(T + 1))
        print(f"sliced kernel, row {r}: worst per-state norm-rel err vs oracle {err:.2e}")
        assert err <= TOL


@pytest.mark.parametrize("N", [1, 15, 16, 33, 1000, 4096, 5100, 6400, 8192])
def test_16_chain_tile_kernel_is_bit_identical_to_the_32_chain_kernels(N, monkeypatch):
    """card_sample_coop16_kernel (batches up to 2 x CUs x 16 chains; BASELINE configs[0] is 5 100): v_mfma_f32_16x16x4
    fed in the 32-chain kernels' k order gives the same bits -- against the 32-chain cooperative kernel and the persistent
    kernel, RNG and injected noise, every output mode, mid-loop segments, in-kernel row tiling, the static entry point."""
    g = load_golden("reg_boston_T1000")
    T = 160
    coef = make_coef_table(g["alphas"][:T], g["omabs"][:T]).to(DEV)
    x, y0, mu = _inputs(g, N, 11)
    outs = {}
    for name, env in (("coop16", {}), ("coop32", {"CARD_AMD_COOP": "no16,no2"}), ("persistent", {"CARD_AMD_COOP": "off"})):
        monkeypatch.delenv("CARD_AMD_COOP", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        last = s.run(x, y0, mu, coef, T, seed=31, elem_offset=5)
        sel = s.run(x, y0, mu, coef, T, seed=31, elem_offset=5, out_mode=_lib.OUT_SELECT, states=[0, 1, 80, T])
        seg = s.run(x, y0, mu, coef, T, seed=31, elem_offset=5, t_hi=100, t_lo=3, y_init=sel[2])
        eps = s.eps_theta(x, sel[1], y0, 7)
        outs[name] = (last, sel, seg, eps)
        if name == "coop16":
            assert torch.equal(s.run(x, y0, mu, coef, T, seed=31, elem_offset=5, static_scratch=True), last)
    for name in ("coop32", "persistent"):
        for a, b in zip(outs["coop16"], outs[name]):
            assert torch.equal(a, b) and torch.isfinite(a).all(), name


def test_16_chain_tile_kernel_year_width_and_row_tiling(monkeypatch):
    """d_in = 92 (twelve hoisted k-groups), rows tiled in-kernel, injected noise, full trajectory: coop16 == coop32 bit
    for bit and within 1e-4 of the oracle."""
    g = load_golden("reg_year_T1000")
    T, rows, n_z = 70, 51, 100                       # 5 100 chains, the Boston plumbing batch at Year's input width
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.25)
    coef = make_coef_table(alphas, omabs).to(DEV)
    x, y0, mu = _inputs(g, rows, 13)
    noise = torch.randn(T, rows * n_z, 1, generator=torch.Generator().manual_seed(14))
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("CARD_AMD_COOP", "" if mode == "1" else "no16,no2")
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        outs.append(s.run(x, y0, mu, coef, T, noise=noise.to(DEV), out_mode=_lib.OUT_ALL, n_z=n_z))
    assert torch.equal(outs[0], outs[1])
    w = g["weights"]
    for r in (0, rows - 1):
        sl = slice(r * n_z, r * n_z + 24)
        tile = lambda t: t[r:r + 1].cpu().repeat(24, 1)
        want = torch.stack(orc.p_sample_loop(w, tile(x), tile(y0), tile(mu), T, alphas, omabs, noise[:, sl]), 0)
        err = max(norm_rel_err(outs[0][k, sl].cpu(), want[k]) for k in range(T + 1))
        print(f"16-chain kernel, row {r}: worst per-state norm-rel err vs oracle {err:.2e}")
        assert err <= TOL


@pytest.mark.parametrize("N", [18000, 70000, 100000])
def test_a_call_cut_into_rounds_of_different_kernels_is_bit_identical(N, monkeypatch):
    """Above a cooperative round (16 384 chains on 256 CUs) a call is cut into whole rounds of the persistent kernel, whole
    rounds of the cooperative kernel and a rest on the small-batch forms (card_guided.hip guided_run): same bits as the
    persistent kernel alone (CARD_AMD_COOP=off), in RNG mode with in-kernel row tiling, and the plan shows the cut."""
    g = load_golden("reg_boston_T1000")
    T, n_z = 64, 8
    coef = make_coef_table(g["alphas"][:T], g["omabs"][:T]).to(DEV)
    x, y0, mu = _inputs(g, N // n_z, 5)
    outs, plans = {}, {}
    for mode in ("", "off"):
        monkeypatch.setenv("CARD_AMD_COOP", mode)
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        outs[mode] = (s.run(x, y0, mu, coef, T, seed=9, elem_offset=3, n_z=n_z),
                      s.run(x, y0, mu, coef, T, seed=9, elem_offset=3, n_z=n_z, out_mode=_lib.OUT_SELECT, states=[0, 13, T]),
                      s.run(x, y0, mu, coef, T, seed=9, elem_offset=3, n_z=n_z, static_scratch=True))
        plans[mode] = s.run(x, y0, mu, coef, T, seed=9, n_z=n_z, plan=True)
    for a, b in zip(outs[""], outs["off"]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    assert plans["off"]["kernel"].startswith("card_sample_guided_kernel") and plans["off"]["launches"] == 1
    assert plans[""]["launches"] >= 2
    assert plans[""]["kernel"].startswith("card_sample_guided_kernel" if N >= 65536 else "card_sample_coop_kernel")


@pytest.mark.parametrize("N", [16383, 16385, 32769, 57345, 65535, 65537, 81921, 131073])
def test_round_boundaries_of_the_cut_are_bit_identical(N, monkeypatch):
    """One chain either side of a round of the cooperative (16 384) and of the persistent kernel (65 536), short calls
    (no time slicing) and pre-tiled rows: the cut call equals the persistent kernel alone."""
    g = load_golden("reg_boston_T1000")
    T = 6
    coef = make_coef_table(g["alphas"][:T], g["omabs"][:T]).to(DEV)
    x, y0, mu = _inputs(g, N, 11)
    outs = []
    for mode in ("", "off"):
        monkeypatch.setenv("CARD_AMD_COOP", mode)
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        outs.append(s.run(x, y0, mu, coef, T, seed=4, out_mode=_lib.OUT_ALL))
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()
