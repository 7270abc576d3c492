"""GPU parity of the streamed-weights persistent kernel (card_stream.hip: ConditionalGuidedModel with hidden = 256,
BASELINE.json configs[4] width-256 leg) against the CPU oracle, on seeded random weights (the reference fixes
hidden = 128, regression/model.py:32-35, so no reference-generated fixture exists for this width: parity here is
pinned by the oracle only), and against the GEMM back end running the same network."""
import os

import pytest
import torch

from card_amd import _lib
from card_amd.distributed import tile_rows
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict
from oracle import card_oracle as orc
from tests.helpers import norm_rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def cuda(t):
    return t.to(DEV)


@pytest.fixture(params=["small4", "full8"], autouse=True)
def wave_form(request, monkeypatch):
    """Batches of at most num_cu * 64 chains run with four waves per workgroup (one per SIMD) by default;
    CARD_AMD_STREAM=big (read when a sampler is created) keeps the 8-wave form.  Every test runs against both."""
    monkeypatch.setenv("CARD_AMD_STREAM", "" if request.param == "small4" else "big")
    return request.param


def _case(x_dim, T, N, seed, cat_x=True, cat_y_pred=True):
    sd = guided_state_dict(x_dim, T, seed, hidden=256, cat_x=cat_x, cat_y_pred=cat_y_pred)
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(N, x_dim, generator=gen)
    y0 = torch.randn(N, 1, generator=gen)
    mu = y0 + 0.1 * torch.randn(N, 1, generator=gen)
    noise = torch.randn(T, N, 1, generator=gen)
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    return sd, x, y0, mu, noise, alphas, omabs


@pytest.mark.parametrize("x_dim,cat_x,cat_y_pred", [
    (1, True, True),      # d_in 3  -> 1 k-step
    (4, True, True),      # d_in 6  -> 2
    (13, True, True),     # d_in 15 -> 4 (Boston)
    (28, True, True),     # d_in 30 -> 8
    (50, True, False),    # d_in 51 -> 16, no y_0_hat column
    (90, True, True),     # d_in 92 -> 24 (Year), three layer-1 slabs
    (7, False, True),     # cat(y_t, y_0_hat)
    (7, False, False),    # y_t alone
])
def test_stream_kernel_matches_oracle(x_dim, cat_x, cat_y_pred):
    T, N = 14, 150                                   # 150 chains: ragged last tile and idle waves
    sd, x, y0, mu, noise, alphas, omabs = _case(x_dim, T, N, 31, cat_x, cat_y_pred)
    w = orc.GuidedWeights.from_state_dict(sd, cat_x=cat_x, cat_y_pred=cat_y_pred)
    want = torch.stack(orc.p_sample_loop(w, x, y0, mu, T, alphas, omabs, noise), 0)
    s = Sampler.from_state_dict(sd, cat_x=cat_x, cat_y_pred=cat_y_pred, device=DEV)
    coef = cuda(make_coef_table(alphas, omabs))
    got = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, noise=cuda(noise), out_mode=_lib.OUT_ALL).cpu()
    worst = max(norm_rel_err(got[k], want[k]) for k in range(T + 1))
    print(f"stream H=256 x_dim={x_dim} cat=({cat_x},{cat_y_pred}): worst per-state norm-rel err {worst:.3e}")
    assert worst <= TOL
    # teacher-forced single steps and eps_theta
    for t in (T - 1, 5, 1, 0):
        nz = cuda(noise[T - t:T - t + 1]) if t >= 1 else None
        one = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, t_hi=t, t_lo=t, y_init=cuda(want[T - 1 - t]), noise=nz).cpu()
        assert norm_rel_err(one, want[T - t]) <= TOL
    eps = s.eps_theta(cuda(x), cuda(want[3]), cuda(y0), 5).cpu()
    assert norm_rel_err(eps, orc.eps_theta(w, x, want[3], y0, 5)) <= TOL
    sel = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, noise=cuda(noise), out_mode=_lib.OUT_SELECT, states=[T, 0, 4]).cpu()
    assert torch.equal(sel, got[[T, 0, 4]])


@pytest.mark.parametrize("N", [1, 16, 17, 129, 1000])
def test_stream_kernel_rng_and_ragged(N):
    T = 9
    sd, x, y0, mu, _, alphas, omabs = _case(13, T, N, 41)
    s = Sampler.from_state_dict(sd, device=DEV)
    coef = cuda(make_coef_table(alphas, omabs))
    z = torch.empty(T, N, 1, device=DEV)
    _lib.check(_lib.lib().card_philox_normal(5, 1000, N, 0, T, z.data_ptr(), None), "philox")
    a = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=5, elem_offset=1000, out_mode=_lib.OUT_ALL)
    b = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, noise=z, out_mode=_lib.OUT_ALL)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    last = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=5, elem_offset=1000)
    assert torch.equal(last, a[-1])


def test_stream_kernel_row_tiling_and_many_groups():
    """n_z tiling equals pre-tiled rows; 40 000 chains = 313 groups of 128 on 256 workgroups (two trips for some,
    idle clamped trips for others) agree with the GEMM back end running the same network."""
    T = 6
    rows, n_z = 400, 100
    sd, x, y0, mu, _, alphas, omabs = _case(13, T, rows, 51)
    s = Sampler.from_state_dict(sd, device=DEV)
    coef = cuda(make_coef_table(alphas, omabs))
    a = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=8, n_z=n_z)
    b = s.run(cuda(tile_rows(x, n_z)), cuda(tile_rows(y0, n_z)), cuda(tile_rows(mu, n_z)), coef, T, seed=8)
    assert a.shape == (rows * n_z, 1) and torch.equal(a, b)
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("CARD_AMD_STREAM", "off")
        sg = Sampler.from_state_dict(sd, device=DEV)      # same weights on the GEMM back end
    c = sg.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=8, n_z=n_z)
    err = norm_rel_err(a.cpu(), c.cpu())
    print(f"stream vs GEMM back end, 40k chains: {err:.3e}")
    assert err <= TOL


def test_stream_kernel_several_launches():
    """300 000 chains = two launches of the streamed kernel (262 144 chains each at most): agree with the GEMM back end
    and are finite; the second launch's chains use their own rows / RNG elements (n_begin offset)."""
    T, rows, n_z = 3, 3000, 100
    sd, x, y0, mu, _, alphas, omabs = _case(13, T, rows, 61)
    s = Sampler.from_state_dict(sd, device=DEV)
    coef = cuda(make_coef_table(alphas, omabs))
    a = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=4, n_z=n_z)
    with pytest.MonkeyPatch.context() as mp:
        mp.setenv("CARD_AMD_STREAM", "off")
        sg = Sampler.from_state_dict(sd, device=DEV)
    b = sg.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=4, n_z=n_z)
    assert a.shape == (rows * n_z, 1) and torch.isfinite(a).all()
    assert norm_rel_err(a.cpu(), b.cpu()) <= TOL


@pytest.mark.parametrize("N", [64, 1000, 16384, 20000, 40000])
def test_four_wave_and_eight_wave_forms_agree_bit_for_bit(N, monkeypatch):
    """(20 000 and 40 000 chains: calls the default dispatch cuts into rounds of different forms, card_stream.hip stream_plan)"""
    T = 7
    sd, x, y0, mu, _, alphas, omabs = _case(13, T, N, 77)
    coef = cuda(make_coef_table(alphas, omabs))
    outs = []
    for small in ("1", "0"):
        monkeypatch.setenv("CARD_AMD_STREAM", "" if small == "1" else "big")
        s = Sampler.from_state_dict(sd, device=DEV)
        outs.append(s.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=3, out_mode=_lib.OUT_ALL))
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()


@pytest.mark.parametrize("N", [16, 700, 4096, 4097, 8192])
def test_tile_split_over_waves_agrees_bit_for_bit(N, monkeypatch, wave_form):
    """Up to num_cu * 32 chains a 16-chain tile is shared by the 4 waves of a workgroup (each computes a share of every
    layer's output features and the activations are traded through LDS, card_stream.hip SPLIT; above num_cu * 16 chains two
    such workgroups per CU): every output element is the same FMA chain over k, so CARD_AMD_STREAM=nosplit (one wave per
    tile) gives the same bits -- RNG mode, every state, in-kernel row tiling, eps_theta -- and the plan names the form."""
    if wave_form != "small4":
        pytest.skip("the split forms belong to the four-wave form")
    T, n_z = 9, 1
    rows = N
    sd, x, y0, mu, _, alphas, omabs = _case(13, T, rows, 40 + N % 7)
    coef = cuda(make_coef_table(alphas, omabs))
    outs, eps = [], []
    for env in ("", "nosplit"):
        monkeypatch.setenv("CARD_AMD_STREAM", env)
        s = Sampler.from_state_dict(sd, device=DEV)
        kw = dict(seed=3, elem_offset=17, out_mode=_lib.OUT_ALL)
        outs.append(s.run(cuda(x), cuda(y0), cuda(mu), coef, T, **kw))
        eps.append(s.eps_theta(cuda(x), cuda(y0), cuda(y0), 4))
        form = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, plan=True, **kw)["kernel"]
        want = "(4 waves)" if env or N > 8192 else "(4 waves, tile over 4 waves)"      # 256 CUs
        assert want in form, form
    assert torch.equal(outs[0], outs[1]) and torch.equal(eps[0], eps[1]) and torch.isfinite(outs[0]).all()
    if N >= 700:        # rows tiled in-kernel: 7 chains per row, ragged against the 16-chain tiles
        r = N // 7
        monkeypatch.setenv("CARD_AMD_STREAM", "")
        s = Sampler.from_state_dict(sd, device=DEV)
        a = s.run(cuda(x[:r]), cuda(y0[:r]), cuda(mu[:r]), coef, T, seed=5, n_z=7)
        monkeypatch.setenv("CARD_AMD_STREAM", "nosplit")
        s1 = Sampler.from_state_dict(sd, device=DEV)
        assert torch.equal(a, s1.run(cuda(x[:r]), cuda(y0[:r]), cuda(mu[:r]), coef, T, seed=5, n_z=7))


@pytest.mark.parametrize("N", [4095, 4097, 8191, 8193, 16385, 32767, 32769, 36865, 70001])
def test_round_boundaries_of_the_cut_are_bit_identical(N, monkeypatch):
    """One chain either side of a round of every form (4 096 / 8 192 / 16 384 / 32 768 chains on 256 CUs): the call cut into
    rounds (card_stream.hip stream_plan) equals the eight-wave ring form alone, every state."""
    T = 5
    sd, x, y0, mu, _, alphas, omabs = _case(13, T, N, 90 + N % 5)
    coef = cuda(make_coef_table(alphas, omabs))
    outs = []
    for env in ("", "big"):
        monkeypatch.setenv("CARD_AMD_STREAM", env)
        s = Sampler.from_state_dict(sd, device=DEV)
        outs.append(s.run(cuda(x), cuda(y0), cuda(mu), coef, T, seed=6, elem_offset=5, out_mode=_lib.OUT_ALL))
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()


@pytest.mark.parametrize("N", [4112, 5000, 6144, 7150])
def test_time_sliced_paired_form_is_bit_identical(N, monkeypatch):
    """Between one and two shared tiles per CU the call is cut into K launches of the paired form (two tiles per 8-wave
    workgroup, each tile alone on its CU in one of the launches; chain states cross launches in a buffer): same bits as
    the un-sliced forms (CARD_AMD_STREAM=noslice) for the last sample, selected states, a segment that starts from a
    given state, injected noise, in-kernel row tiling and the entry point that allocates nothing."""
    T = 150
    sd, x, y0, mu, _, _, _ = _case(13, T, N, 17)
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.02)        # (_case's schedule is made for a dozen timesteps)
    coef = cuda(make_coef_table(alphas, omabs))
    gen = torch.Generator().manual_seed(5)
    noise = cuda(torch.randn(T, N, 1, generator=gen))
    outs = {}
    for env in ("", "noslice"):
        monkeypatch.setenv("CARD_AMD_STREAM", env)
        s = Sampler.from_state_dict(sd, device=DEV)
        kw = dict(seed=31, elem_offset=9)
        last = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, **kw)
        sel = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, out_mode=_lib.OUT_SELECT, states=[0, 1, 40, 41, 99, T - 1, T], **kw)
        seg = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, t_hi=120, t_lo=10, y_init=sel[2], **kw)
        inj = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, noise=noise, out_mode=_lib.OUT_SELECT, states=[T, 77, 3])
        r7 = N // 7
        tiled = s.run(cuda(x[:r7]), cuda(y0[:r7]), cuda(mu[:r7]), coef, T, n_z=7, **kw) if r7 * 7 > 4096 else last
        stat = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, static_scratch=True, **kw)
        outs[env] = (last, sel, seg, inj, tiled, stat)
        form = s.run(cuda(x), cuda(y0), cuda(mu), coef, T, plan=True, **kw)
        assert ("time-sliced" in form["kernel"]) == (env == ""), form
        assert form["launches"] == (1 if env else (N + 15) // 16 and -(-((N + 15) // 16) // (512 - (N + 15) // 16)))
    for a, b in zip(outs[""], outs["noslice"]):
        assert torch.equal(a, b) and torch.isfinite(a).all()
    assert torch.equal(outs[""][0], outs[""][1][-1]) and torch.equal(outs[""][0], outs[""][5])
