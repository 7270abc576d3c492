"""GPU parity of the kernel configuration bench.py times, at bench scale.

bench.py's default line runs `card_sample_guided_kernel<8, 2>` (hoisted layer 1 held as two LDS columns, in-kernel row
tiling with n_z = 1000) as several launches with n_begin > 0 (card_guided.hip: guided_run's chunking).  These tests run
exactly that configuration -- the toy network (reference fixture reg_toy_T50, d_in = 3), n_z in {1000, 37}, more than
two full launches plus a ragged last one -- and check it

  (a) in RNG mode, bit for bit, against the pre-tiled form of the same chains (the reference's host-side tiling,
      regression/card_regression.py:1027-1035, which takes the scratch-slot kernel <8, 1> or the cooperative kernel) on
      slices that straddle every launch boundary, the ragged end, and test-row boundaries inside a tile;
  (b) with injected noise against the CPU oracle (pinned to regression/diffusion_utils.py:107-119 by the fixtures)
      on the same slices, to the 1e-4 bar.

The same is done for the streamed hidden-256 kernel (card_stream.hip) with in-kernel tiling over > 2 launches."""
import pytest
import torch

from card_amd import _lib
from card_amd.distributed import tile_rows
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict
from oracle import card_oracle as orc
from tests.helpers import load_golden, elem_errs, fmt_errs

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"


def cuda(t):
    return t.to(DEV)


def _chunk_guided():
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    return cu * 8 * 16 * 32            # guided_run: num_cu x 8 waves x 16 tiles x 32 chains per launch


def _chunk_stream():
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    return cu * 8 * 128                # stream_run: num_cu x 8 groups x 128 chains per launch


def _slices(N, chunk, n_z):
    """[a, b) chain ranges: around each launch boundary (unaligned starts, so the pre-tiled run puts the same chains in
    other lanes and tiles), the start, the ragged end, and one range that begins exactly on a test-row boundary."""
    out = [(0, 2 * n_z + 45)]
    for k in range(1, N // chunk + 1):
        b = k * chunk
        if b < N:
            out.append((max(0, b - 2 * n_z - 17), min(N, b + n_z + 1029)))
    out.append((max(0, N - n_z - 333), N))
    r = (N // 2) // n_z
    out.append((r * n_z, min(N, (r + 3) * n_z + 7)))
    return out


def _rows_for_slice(a, b, n_z):
    chains = torch.arange(a, b)
    return chains // n_z


@pytest.mark.parametrize("n_z", [1000, 37])
def test_bench_kernel_configuration_multi_launch(n_z, monkeypatch):
    g = load_golden("reg_toy_T50")
    T = g["T"]
    chunk = _chunk_guided()
    rows = (2 * chunk) // n_z + 3                       # > 2 full launches + a ragged third
    N = rows * n_z
    assert N > 2 * chunk
    gen = torch.Generator().manual_seed(5)
    idx = torch.randint(0, g["x"].shape[0], (rows,), generator=gen)
    x = g["x"][idx] + 0.05 * torch.randn(rows, g["x"].shape[1], generator=gen)
    y0 = g["y0hat"][idx] + 0.05 * torch.randn(rows, 1, generator=gen)
    mu = y0 + 0.01 * torch.randn(rows, 1, generator=gen)
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    states = [0, T // 2, T]
    seed, off = 0xBEEF_1234, 7_000_001

    monkeypatch.setenv("CARD_AMD_COOP", "1")
    s = Sampler.from_state_dict(g["sd"], device=DEV)    # big batch -> persistent kernel, n_z >= 32 -> <8, 2>
    monkeypatch.setenv("CARD_AMD_COOP", "0")
    s_slot = Sampler.from_state_dict(g["sd"], device=DEV)   # small pre-tiled batches on the scratch-slot kernel <8, 1>
    xd, y0d, mud = cuda(x), cuda(y0), cuda(mu)

    # (a) RNG mode
    big = s.run(xd, y0d, mud, coef, T, seed=seed, elem_offset=off, n_z=n_z, out_mode=_lib.OUT_SELECT, states=states)
    assert big.shape == (3, N, 1) and torch.isfinite(big).all()
    last = s.run(xd, y0d, mud, coef, T, seed=seed, elem_offset=off, n_z=n_z)          # bench.py's own call (OUT_LAST)
    assert torch.equal(last, big[2])
    for a, b in _slices(N, chunk, n_z):
        r = _rows_for_slice(a, b, n_z)
        args = (cuda(x[r]), cuda(y0[r]), cuda(mu[r]), coef, T)
        kw = dict(seed=seed, elem_offset=off + a, out_mode=_lib.OUT_SELECT, states=states)
        for name, smp in (("coop", s), ("slot", s_slot)):
            small = smp.run(*args, **kw)
            assert torch.equal(small, big[:, a:b]), (name, a, b)

    # (b) injected noise against the oracle
    zgen = torch.Generator(device=DEV).manual_seed(11)
    noise = torch.randn(T, N, 1, device=DEV, generator=zgen)
    bigz = s.run(xd, y0d, mud, coef, T, noise=noise, n_z=n_z, out_mode=_lib.OUT_SELECT, states=states)
    worst = (0.0, 0.0, 0.0)
    for a, b in _slices(N, chunk, n_z):
        r = _rows_for_slice(a, b, n_z)
        nz = noise[:, a:b].contiguous()
        small = s_slot.run(cuda(x[r]), cuda(y0[r]), cuda(mu[r]), coef, T, noise=nz, out_mode=_lib.OUT_SELECT, states=states)
        assert torch.equal(small, bigz[:, a:b]), (a, b)
        want = torch.stack(orc.p_sample_loop(g["weights"], x[r], y0[r], mu[r], T, g["alphas"], g["omabs"], nz.cpu()), 0)
        got = bigz[:, a:b].cpu()
        for i, k in enumerate(states):
            e = elem_errs(got[i], want[k])
            worst = tuple(max(u, v) for u, v in zip(worst, e))
            assert e[0] <= TOL, (a, b, k, e)
    print(f"bench configuration n_z={n_z}, {N} chains in {-(-N // chunk)} launches vs oracle: {fmt_errs(worst)}")


@pytest.mark.parametrize("n_z", [100, 37])
def test_stream_kernel_multi_launch_with_row_tiling(n_z):
    T, x_dim = 8, 13
    chunk = _chunk_stream()
    rows = (2 * chunk) // n_z + 5
    N = rows * n_z
    sd = guided_state_dict(x_dim, T, 71, hidden=256)
    w = orc.GuidedWeights.from_state_dict(sd)
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(rows, x_dim, generator=gen)
    y0 = torch.randn(rows, 1, generator=gen)
    mu = y0 + 0.1 * torch.randn(rows, 1, generator=gen)
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    s = Sampler.from_state_dict(sd, device=DEV)
    states = [0, 3, T]
    xd, y0d, mud = cuda(x), cuda(y0), cuda(mu)
    big = s.run(xd, y0d, mud, coef, T, seed=9, elem_offset=12345, n_z=n_z, out_mode=_lib.OUT_SELECT, states=states)
    zgen = torch.Generator(device=DEV).manual_seed(12)
    noise = torch.randn(T, N, 1, device=DEV, generator=zgen)
    bigz = s.run(xd, y0d, mud, coef, T, noise=noise, n_z=n_z, out_mode=_lib.OUT_SELECT, states=states)
    worst = (0.0, 0.0, 0.0)
    for a, b in _slices(N, chunk, n_z):
        r = _rows_for_slice(a, b, n_z)
        args = (cuda(x[r]), cuda(y0[r]), cuda(mu[r]), coef, T)
        small = s.run(*args, seed=9, elem_offset=12345 + a, out_mode=_lib.OUT_SELECT, states=states)
        assert torch.equal(small, big[:, a:b]), (a, b)
        nz = noise[:, a:b].contiguous()
        want = torch.stack(orc.p_sample_loop(w, x[r], y0[r], mu[r], T, alphas, omabs, nz.cpu()), 0)
        got = bigz[:, a:b].cpu()
        for i, k in enumerate(states):
            e = elem_errs(got[i], want[k])
            worst = tuple(max(u, v) for u, v in zip(worst, e))
            assert e[0] <= TOL, (a, b, k, e)
    print(f"stream kernel n_z={n_z}, {N} chains in {-(-N // chunk)} launches vs oracle: {fmt_errs(worst)}")
