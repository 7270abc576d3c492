"""GPU parity: the HIP sampler (through the C ABI) against the oracle and the reference's golden fixtures.

Bar (BASELINE.json north_star): fp32, every stored y_t of the trajectory within 1e-4
norm-relative (max|d| / max(1, max|ref|)) under injected noise, teacher-forced single steps likewise;
q_sample and integer/RNG-bit work bit-exact."""
import argparse

import numpy as np
import pytest
import torch

import card_amd
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from oracle import card_oracle as orc
from tests.helpers import golden_names, load_golden, norm_rel_err, elem_errs, fmt_errs, record_parity, worst_errs

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"

REG = [n for n in golden_names("reg")]
SUPPORTED_REG = REG


class RefShapedGuided(torch.nn.Module):
    """An nn.Module with the attribute layout of regression/model.py:21 ConditionalGuidedModel (lin1..lin3 with
    .lin/.embed, lin4, cat_x, cat_y_pred): what the shim's duck-typed `model` contract recognises."""

    class CL(torch.nn.Module):
        def __init__(self, i, o, n):
            super().__init__()
            self.lin = torch.nn.Linear(i, o)
            self.embed = torch.nn.Embedding(n, o)

    def __init__(self, sd, cat_x, cat_y_pred):
        super().__init__()
        d_in, n = sd["lin1.lin.weight"].shape[1], sd["lin1.embed.weight"].shape[0]
        self.cat_x, self.cat_y_pred = cat_x, cat_y_pred
        self.lin1, self.lin2, self.lin3 = self.CL(d_in, 128, n), self.CL(128, 128, n), self.CL(128, 128, n)
        self.lin4 = torch.nn.Linear(128, 1)
        self.load_state_dict(sd)


def model_of(g):
    return RefShapedGuided(g["sd"], g["cat_x"], g["cat_y_pred"]).to(DEV).eval()


def cuda(t):
    return t.to(DEV)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"


@pytest.fixture(params=["coop", "persistent"], autouse=True)
def small_batch_kernel(request, monkeypatch):
    """Every test runs twice: with the cooperative small-batch kernel (the default below ~24 000 chains) and with it
    switched off (CARD_AMD_COOP=off, read when a sampler is created), so the fixtures pin both kernels."""
    monkeypatch.setenv("CARD_AMD_COOP", "" if request.param == "coop" else "off")
    return request.param


@pytest.mark.parametrize("name", SUPPORTED_REG)
def test_free_running_loop_matches_reference(name, small_batch_kernel):
    g = load_golden(name)
    m = model_of(g)
    T = g["T"]
    seq = card_amd.p_sample_loop(m, cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(g["alphas"]),
                                 cuda(g["omabs"]), noise=cuda(g["noise"]))
    assert isinstance(seq, list) and len(seq) == T + 1 and seq[0].shape == g["y0hat"].shape
    traj = torch.stack(seq, 0).cpu()
    assert torch.equal(traj[0], g["noise"][0] + g["yTmean"])
    errs = [elem_errs(traj[k], g["traj"][i]) for i, k in enumerate(g["traj_idx"])]
    worst = max(e[0] for e in errs)
    print(f"{name}: worst per-state {fmt_errs(worst_errs(errs))}")
    record_parity(name, "card_sample_coop16_kernel" if small_batch_kernel == "coop" else "card_sample_guided_kernel", worst_errs(errs))
    assert worst <= TOL
    # only_last_sample (classification/diffusion_utils.py:136-139) and a state selection
    y0 = card_amd.p_sample_loop(m, cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(g["alphas"]),
                                cuda(g["omabs"]), True, noise=cuda(g["noise"]))
    assert torch.equal(y0.cpu(), traj[T])
    sel = [0, T // 3, T - 1, T]
    got = card_amd.p_sample_loop(m, cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(g["alphas"]),
                                 cuda(g["omabs"]), noise=cuda(g["noise"]), states=sel)
    assert torch.equal(got.cpu(), traj[sel])


@pytest.mark.parametrize("name", SUPPORTED_REG)
def test_teacher_forced_steps_and_eps(name):
    g = load_golden(name)
    m, d, T = model_of(g), g["raw"], g["T"]
    full = dict(zip(g["traj_idx"], g["traj"]))
    s = card_amd.sampler_for(m)
    args = (cuda(g["x"]),)
    for t in g["probe_t"]:
        if (T - t) not in full:
            continue
        y_t = cuda(full[T - t])
        eps = s.eps_theta(cuda(g["x"]), y_t, cuda(g["y0hat"]), t).cpu()
        assert norm_rel_err(eps, d[f"eps.t{t}"]) <= TOL, t
        out = card_amd.p_sample(m, cuda(g["x"]), y_t, cuda(g["y0hat"]), cuda(g["yTmean"]), t, cuda(g["alphas"]),
                                cuda(g["omabs"]), noise=cuda(torch.from_numpy(d[f"stepz.t{t}"]))).cpu()
        assert norm_rel_err(out, d[f"step.t{t}"]) <= TOL, t
    last = card_amd.p_sample_t_1to0(m, cuda(g["x"]), cuda(full[T - 1]), cuda(g["y0hat"]), cuda(g["yTmean"]),
                                    cuda(g["omabs"])).cpu()
    assert norm_rel_err(last, d["last"]) <= TOL
    eps0 = s.eps_theta(cuda(g["x"]), cuda(full[T - 1]), cuda(g["y0hat"]), 0).cpu()
    assert norm_rel_err(eps0, d["eps.t0"]) <= TOL


@pytest.mark.parametrize("name", ["reg_boston_T50", "reg_toy_T50"])
def test_every_step_teacher_forced_against_oracle(name, small_batch_kernel):
    """Feed the reference y_t at EVERY timestep, compare y_{t-1} (SURVEY.md section 8c teacher-forced metric)."""
    g = load_golden(name)
    m, T = model_of(g), g["T"]
    assert g["traj_idx"] == list(range(T + 1))
    errs = []
    for t in range(T - 1, 0, -1):
        # timestep t maps list position T-1-t to position T-t and consumes noise draw T-t
        out = card_amd.p_sample(m, cuda(g["x"]), cuda(g["traj"][T - 1 - t]), cuda(g["y0hat"]), cuda(g["yTmean"]), t,
                                cuda(g["alphas"]), cuda(g["omabs"]), noise=cuda(g["noise"][T - t])).cpu()
        errs.append(elem_errs(out, g["traj"][T - t]))
    worst = max(e[0] for e in errs)
    print(f"{name}: worst teacher-forced {fmt_errs(worst_errs(errs))}")
    record_parity(name, "card_sample_coop16_kernel" if small_batch_kernel == "coop" else "card_sample_guided_kernel", worst_errs(errs),
                  mode="teacher-forced, every timestep")
    assert worst <= TOL


@pytest.mark.parametrize("name", SUPPORTED_REG[:3])
def test_q_sample_bit_exact(name):
    g = load_golden(name)
    d, T = g["raw"], g["T"]
    full = dict(zip(g["traj_idx"], g["traj"]))
    out = card_amd.q_sample(cuda(full[T]), cuda(g["y0hat"]), cuda(g["alphas_bar_sqrt"]), cuda(g["omabs"]),
                            cuda(torch.from_numpy(d["q.t"])), noise=cuda(torch.from_numpy(d["q.noise"]))).cpu()
    assert torch.equal(out, torch.from_numpy(d["q.out"]))


def test_hoisted_layer1_slots_are_reused_across_tiles():
    """The persistent kernel's hoisted layer-1 tile (static part computed once per tile into a per-wave scratch slot) on
    the cat variants: parity with the reference, and many tiles per wave give every copy of a chain the same bits."""
    for name in ("reg_boston_T50", "reg_toy_T50", "reg_yonly_T50", "reg_nocaty_T50"):
        g = load_golden(name)
        kw = dict(cat_x=g["cat_x"], cat_y_pred=g["cat_y_pred"], device=DEV)
        b = Sampler.from_state_dict(g["sd"], **kw)
        rb = _raw_run(b, g, noise=cuda(g["noise"]), out_mode=_lib.OUT_ALL).cpu()
        ref = g["traj"]
        assert max(norm_rel_err(rb[k], ref[i]) for i, k in enumerate(g["traj_idx"])) <= TOL
        # many tiles per wave: slot reuse across tiles
        reps = 700
        big = _raw_run(b, g, N_rep=reps, noise=cuda(g["noise"].repeat(1, reps, 1))).cpu().reshape(reps, -1, 1)
        assert torch.equal(big, big[:1].expand_as(big)) and torch.equal(big[0], rb[-1])


def _raw_run(s, g, N_rep=1, **kw):
    rep = lambda t: cuda(t.repeat(N_rep, 1))
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    x = rep(g["x"]) if g["x"].shape[1] else None
    return s.run(x, rep(g["y0hat"]), rep(g["yTmean"]), coef, g["T"], **kw)


def test_rng_mode_equals_injected_mode_fed_with_its_own_draws():
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    T, N = g["T"], g["x"].shape[0]
    seed, off = 0x1234_5678_9ABC_DEF0, 1_000_003
    z = torch.empty(T, N, 1, device=DEV)
    _lib.check(_lib.lib().card_philox_normal(seed, off, N, 0, T, z.data_ptr(), None), "philox")
    torch.cuda.synchronize()
    a = _raw_run(s, g, seed=seed, elem_offset=off, out_mode=_lib.OUT_ALL)
    b = _raw_run(s, g, noise=z, out_mode=_lib.OUT_ALL)
    assert torch.equal(a, b)
    # and the draws are the oracle's Philox4x32-10 + Box-Muller restatement (transcendental rounding only)
    want = orc.philox_noise(seed, T, N, 1, elem_offset=off)
    assert float((z.cpu() - want).abs().max()) < 2e-6
    zc = z.cpu().reshape(-1)
    assert torch.isfinite(zc).all()


def test_generator_against_the_float64_restatement_on_a_large_sample():
    """2.4 million normals (the Box-Muller half runs on the hardware log2 / sqrt / sin / cos, card_common.h): every one
    within 2e-6 of oracle.philox_normal's float64 value -- tools/ubench/boxmuller.hip bounds the error at 1.5e-6 over all
    2^23 uniforms the generator can produce -- and the sample's moments are a standard normal's."""
    seed, off, N, D = 0xC0FFEE_0000_0042, (1 << 33) + 12345, 200_000, 12       # the element index crosses 2^32
    z = torch.empty(D, N, 1, device=DEV)
    _lib.check(_lib.lib().card_philox_normal(seed, off, N, 0, D, z.data_ptr(), None), "philox")
    torch.cuda.synchronize()
    want = orc.philox_noise(seed, D, N, 1, elem_offset=off)
    got = z.cpu()
    assert torch.isfinite(got).all()
    assert float((got - want).abs().max()) < 2e-6
    v = got.double().reshape(-1)
    assert abs(float(v.mean())) < 3e-3 and abs(float(v.var()) - 1.0) < 5e-3
    assert abs(float((v ** 3).mean())) < 1e-2 and abs(float((v ** 4).mean()) - 3.0) < 3e-2
    assert float(v.abs().max()) > 4.5                                           # the tail is there


def test_rng_sharding_invariance_and_seed_sensitivity():
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    N = g["x"].shape[0]
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    whole = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"], seed=7, elem_offset=100)
    k = 13
    lo = s.run(cuda(g["x"][:k]), cuda(g["y0hat"][:k]), cuda(g["yTmean"][:k]), coef, g["T"], seed=7, elem_offset=100)
    hi = s.run(cuda(g["x"][k:]), cuda(g["y0hat"][k:]), cuda(g["yTmean"][k:]), coef, g["T"], seed=7,
               elem_offset=100 + k)
    assert torch.equal(whole, torch.cat([lo, hi], 0))
    other = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"], seed=8, elem_offset=100)
    assert not torch.equal(whole, other)
    assert N == whole.shape[0]


@pytest.mark.parametrize("n", [1, 31, 32, 33, 64, 257])
def test_ragged_row_counts(n):
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    N0 = g["x"].shape[0]
    idx = torch.arange(n) % N0
    out = s.run(cuda(g["x"][idx]), cuda(g["y0hat"][idx]), cuda(g["yTmean"][idx]), coef, g["T"],
                noise=cuda(g["noise"][:, idx]), out_mode=_lib.OUT_ALL).cpu()
    base = s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"], noise=cuda(g["noise"]),
                 out_mode=_lib.OUT_ALL).cpu()
    assert torch.equal(out, base[:, idx])            # a chain's result does not depend on its lane / tile
    assert norm_rel_err(out[-1], g["traj"][-1][idx]) <= TOL


def test_empty_batch_and_argument_errors():
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    e = torch.empty(0, 1)
    out = s.run(torch.empty(0, 13), e, e, coef, g["T"])
    assert out.shape == (0, 1)
    with pytest.raises(_lib.CardError):   # t_hi must be T-1 when starting a loop
        s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"], t_hi=3)
    with pytest.raises(_lib.CardError):   # wrong x width
        s.run(cuda(g["x"][:, :5]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"])
    with pytest.raises(_lib.CardError):   # T beyond the gamma tables
        s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"] + 5)
    with pytest.raises(_lib.CardError, match="twice"):
        s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"], out_mode=_lib.OUT_SELECT, states=[3, 3])
    with pytest.raises(_lib.CardError, match="not produced"):       # a single step (state 11 -> 12) cannot fill slot of state 5
        s.run(cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), coef, g["T"], t_hi=g["T"] - 12, t_lo=g["T"] - 12,
              y_init=cuda(g["y0hat"]), out_mode=_lib.OUT_SELECT, states=[5, 12])


def test_guided_model_samples_in_train_mode_like_the_reference():
    """ConditionalGuidedModel has no train/eval-dependent layer and the reference samples from it right after model.train()
    (validation inside Diffusion.train: regression/card_regression.py:365 -> :457, :533): same result as in eval mode."""
    g = load_golden("reg_boston_T50")
    m = model_of(g)
    args = (cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), g["T"], cuda(g["alphas"]), cuda(g["omabs"]))
    a = card_amd.p_sample_loop(m, *args, True, noise=cuda(g["noise"]))
    b = card_amd.p_sample_loop(m.train(), *args, True, noise=cuda(g["noise"]))
    assert torch.equal(a, b) and norm_rel_err(b.cpu(), g["traj"][-1]) <= TOL


def test_data_copy_weight_update_is_noticed():
    """The reference's EMA swap is `param.data.copy_(shadow)` (regression/ema.py:21), which leaves Parameter._version
    untouched: the content hash taken at every loop entry must catch it."""
    g, g2 = load_golden("reg_boston_T50"), load_golden("reg_boston_rand_T50")
    m = model_of(g)
    args = (cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), g["T"], cuda(g["alphas"]), cuda(g["omabs"]))
    a = card_amd.p_sample_loop(m, *args, True, noise=cuda(g["noise"]))
    s_before = card_amd.sampler_for(m)
    versions = [p._version for p in m.parameters()]
    for name, p in m.named_parameters():                      # ema.ema(): param.data.copy_(shadow[name].data)
        p.data.copy_(g2["sd"][name].to(DEV))
    assert [p._version for p in m.parameters()] == versions   # the blind spot of the identity key
    b = card_amd.p_sample_loop(m, *args, True, noise=cuda(g["noise"]))
    assert card_amd.sampler_for(m) is not s_before
    want = orc.p_sample_loop(g2["weights"], g["x"], g["y0hat"], g["yTmean"], g["T"], g["alphas"], g["omabs"], g["noise"],
                             only_last_sample=True)
    assert not torch.equal(a, b) and norm_rel_err(b.cpu(), want) <= TOL
    card_amd.invalidate(m)
    assert torch.equal(card_amd.p_sample_loop(m, *args, True, noise=cuda(g["noise"])), b)


def test_schedule_temporaries_of_equal_length_do_not_share_a_table():
    """Two schedules with the same T passed as temporaries (freed after each call, so the allocator hands the same
    address to the next one): each call must use its own coefficients."""
    g = load_golden("reg_boston_T50")
    m, T = model_of(g), g["T"]
    for end in (0.4, 0.05, 0.4, 0.2):
        alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, end)
        got = card_amd.p_sample_loop(m, cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(alphas), cuda(omabs),
                                     True, noise=cuda(g["noise"]))
        want = orc.p_sample_loop(g["weights"], g["x"], g["y0hat"], g["yTmean"], T, alphas, omabs, g["noise"],
                                 only_last_sample=True)
        assert norm_rel_err(got.cpu(), want) <= TOL, end


def test_q_sample_shared_timestep_and_range_check():
    """card_regression.py:512-514 passes a 1-element t for the whole batch (the reference's extract() broadcasts it)."""
    g = load_golden("reg_boston_T50")
    T, N = g["T"], g["x"].shape[0]
    y, z = cuda(g["traj"][-1]), cuda(g["noise"][3])
    sab, som = cuda(g["alphas_bar_sqrt"]), cuda(g["omabs"])
    one = card_amd.q_sample(y, cuda(g["y0hat"]), sab, som, torch.tensor([T - 1]).to(DEV), noise=z)
    per_row = card_amd.q_sample(y, cuda(g["y0hat"]), sab, som, torch.full((N,), T - 1).to(DEV), noise=z)
    assert torch.equal(one, per_row)
    assert torch.equal(one.cpu(), orc.q_sample(g["traj"][-1], g["y0hat"], g["alphas_bar_sqrt"], g["omabs"],
                                               torch.full((N,), T - 1), g["noise"][3]))
    with pytest.raises(_lib.CardError, match="outside"):
        card_amd.q_sample(y, cuda(g["y0hat"]), sab, som, torch.full((N,), T).to(DEV), noise=z)


def test_weight_update_invalidates_cached_sampler():
    g = load_golden("reg_boston_T50")
    m = model_of(g)
    a = card_amd.sampler_for(m)
    assert card_amd.sampler_for(m) is a
    with torch.no_grad():
        m.lin4.bias.add_(1.0)           # e.g. EMA copy, regression/card_regression.py:934-940
    b = card_amd.sampler_for(m)
    assert b is not a


def test_full_chip_many_tiles_property():
    """Whole-GPU launch (more tiles than waves, several launches): every replica of a fixture row, fed the
    same noise, must reproduce that row's trajectory end bit-for-bit, and the first replica matches the golden."""
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    N0 = g["x"].shape[0]
    cu = torch.cuda.get_device_properties(0).multi_processor_count
    reps = (cu * 8 * 32 * 17) // N0 + 3          # > one launch chunk (16 tiles per wave)
    out = _raw_run(s, g, N_rep=reps, noise=cuda(g["noise"].repeat(1, reps, 1))).cpu()
    out = out.reshape(reps, N0, 1)
    assert torch.equal(out, out[:1].expand_as(out))
    assert norm_rel_err(out[0], g["traj"][-1]) <= TOL


def test_end_to_end_rng_statistics_match_reference_sampling():
    """In-kernel Philox vs the oracle driven by torch.randn: per-row predictive mean / std and the pooled
    PICP-style coverage agree within sampling error (north_star: metric-level agreement, not bitwise)."""
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    rows, nz, T = 16, 1500, g["T"]
    tile = lambda t: t[:rows].repeat_interleave(nz, 0)        # (row, z) flattening, card_regression.py:1027-1035
    x, y0 = tile(g["x"]), tile(g["y0hat"])
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    hip = s.run(cuda(x), cuda(y0), cuda(y0), coef, T, seed=2024).cpu().reshape(rows, nz)
    torch.manual_seed(99)
    ref = orc.p_sample_loop(g["weights"], x, y0, y0, T, g["alphas"], g["omabs"], torch.randn(T, rows * nz, 1),
                            only_last_sample=True).reshape(rows, nz)
    sd = ref.std(1)
    assert ((hip.mean(1) - ref.mean(1)).abs() < 5 * sd / nz ** 0.5 * 2 ** 0.5).all()
    assert ((hip.std(1) / sd - 1).abs() < 0.12).all()
    lo, hi = np.percentile(ref.numpy(), [2.5, 97.5], axis=1)
    cover = ((hip.numpy() >= lo[:, None]) & (hip.numpy() <= hi[:, None])).mean()
    assert abs(cover - 0.95) < 0.01


def test_in_kernel_row_tiling_equals_host_tiling():
    """n_z > 1 (rows expanded to chains inside the kernel) == the reference's host-side repeat tiling
    (regression/card_regression.py:1027-1035), bit for bit, in RNG and injected-noise mode."""
    from card_amd.distributed import tile_rows
    for name in ("reg_boston_T50", "reg_year_T50"):
        g = load_golden(name)
        s = Sampler.from_state_dict(g["sd"], device=DEV)
        coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
        rows, T = 9, g["T"]
        x, y0 = g["x"][:rows], g["y0hat"][:rows]
        # n_z >= 32 takes the LDS-resident two-column form of the hoisted layer 1, n_z < 32 the scratch-slot form
        for n_z in (5, 32, 37, 1000):
            a = s.run(cuda(x), cuda(y0), cuda(y0), coef, T, seed=3, n_z=n_z, out_mode=_lib.OUT_ALL)
            b = s.run(cuda(tile_rows(x, n_z)), cuda(tile_rows(y0, n_z)), cuda(tile_rows(y0, n_z)), coef, T, seed=3,
                      out_mode=_lib.OUT_ALL)
            assert a.shape == (T + 1, rows * n_z, 1) and torch.equal(a, b), n_z
        n_z = 37
        z = torch.randn(T, rows * n_z, 1)
        a = s.run(cuda(x), cuda(y0), cuda(y0), coef, T, noise=cuda(z), n_z=n_z)
        want = orc.p_sample_loop(g["weights"], tile_rows(x, n_z), tile_rows(y0, n_z), tile_rows(y0, n_z), T, g["alphas"],
                                 g["omabs"], z, only_last_sample=True)
        assert norm_rel_err(a.cpu(), want) <= TOL
    with pytest.raises(_lib.CardError):
        s.run(cuda(x), cuda(y0), cuda(y0[:-1]), coef, T, n_z=n_z)


def test_cooperative_kernel_many_tiles_per_workgroup(monkeypatch):
    """40 000 chains = 1250 tiles on at most 512 cooperative workgroups (each walks two or three tiles) against the
    persistent kernel on the same batch: the two kernels are bit-identical by construction."""
    g = load_golden("reg_boston_T50")
    reps = 1000
    outs = []
    for mode in ("force", "off"):                         # cooperative forced / switched off
        monkeypatch.setenv("CARD_AMD_COOP", mode)
        s = Sampler.from_state_dict(g["sd"], cat_x=g["cat_x"], cat_y_pred=g["cat_y_pred"], device=DEV)
        outs.append(_raw_run(s, g, N_rep=reps, seed=17, out_mode=_lib.OUT_SELECT, states=[0, 10, g["T"]]))
    assert outs[0].shape[1] == 40 * reps and torch.equal(outs[0], outs[1])
