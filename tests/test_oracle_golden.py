"""The CPU oracle against fixtures produced by the reference itself (oracle/gen_golden.py).

Pins the oracle: same aten ops in the same order as the reference, so on the build container
(where the fixtures were made) the match is bit-exact; elsewhere (other CPU / BLAS kernel
selection) it must stay inside 2e-6 norm-relative, far below the 1e-4 parity bar."""
import numpy as np
import pytest
import torch

from oracle import card_oracle as orc
from tests.helpers import golden_names, load_golden, norm_rel_err, GOLDEN

TOL = 2e-6


@pytest.mark.parametrize("name", golden_names())
def test_loop_matches_reference(name):
    g = load_golden(name)
    torch.set_num_threads(8 if g["T"] * g["x_in"].shape[1] > 1_000_000 else 4)
    seq = orc.p_sample_loop(g["weights"], g["x_in"], g["y0hat"], g["yTmean"], g["T"], g["alphas"], g["omabs"],
                            g["noise"])
    assert len(seq) == g["T"] + 1
    traj = torch.stack(seq, 0)[g["traj_idx"]]
    assert norm_rel_err(traj, g["traj"]) <= TOL
    if g["T"] * g["x_in"].shape[1] > 1_000_000:
        return                    # clsw_cifar10_T1000: one pass of the oracle is ~30 s of CPU; the flag is covered by every other fixture
    y0 = orc.p_sample_loop(g["weights"], g["x_in"], g["y0hat"], g["yTmean"], g["T"], g["alphas"], g["omabs"],
                           g["noise"], only_last_sample=True)
    assert torch.equal(y0, seq[-1])


@pytest.mark.parametrize("name", golden_names())
def test_eps_theta_probes(name):
    g = load_golden(name)
    T = g["T"]
    full = dict(zip(g["traj_idx"], g["traj"]))
    for t in g["probe_t"]:
        if (T - t) not in full:
            continue
        eps = orc.eps_theta(g["weights"], g["x_in"], full[T - t], g["y0hat"], t)
        assert norm_rel_err(eps, g["raw"][f"eps.t{t}"]) <= TOL


@pytest.mark.parametrize("name", golden_names("reg"))
def test_single_steps_and_q_sample(name):
    g = load_golden(name)
    T, d = g["T"], g["raw"]
    coef = orc.make_coef_table(g["alphas"], g["omabs"])
    full = dict(zip(g["traj_idx"], g["traj"]))
    for t in g["probe_t"]:
        if (T - t) not in full:
            continue
        out = orc.p_sample(g["weights"], g["x"], full[T - t], g["y0hat"], g["yTmean"], t, coef,
                           torch.from_numpy(d[f"stepz.t{t}"]))
        assert norm_rel_err(out, d[f"step.t{t}"]) <= TOL
    last = orc.p_sample_t_1to0(g["weights"], g["x"], full[T - 1], g["y0hat"], g["yTmean"], coef)
    assert norm_rel_err(last, d["last"]) <= TOL
    q = orc.q_sample(full[T], g["y0hat"], g["alphas_bar_sqrt"], g["omabs"], torch.from_numpy(d["q.t"]),
                     torch.from_numpy(d["q.noise"]))
    assert torch.equal(q, torch.from_numpy(d["q.out"]))


def test_schedules_bit_exact():
    z = np.load(f"{GOLDEN}/schedules.npz")
    keys = sorted({k.rsplit(".", 1)[0] for k in z.files})
    args = {"linear_1000": (1e-4, 0.02), "linear_50": (1e-4, 0.4)}
    for key in keys:
        name, T = key.rsplit("_", 1)
        b0, b1 = args.get(key, (1e-4, 0.02))
        alphas, abs_, omabs = orc.make_schedule(name, int(T), b0, b1)
        assert np.array_equal(alphas.numpy(), z[key + ".alphas"]), key
        assert np.array_equal(abs_.numpy(), z[key + ".abs"]), key
        assert np.array_equal(omabs.numpy(), z[key + ".omabs"]), key


def test_coef_table_against_scalar_formulas():
    """Every row equals the (1,1)-tensor arithmetic of reference p_sample lines 71-80, 88."""
    alphas, _, omabs = orc.make_schedule("linear", 1000, 1e-4, 0.02)
    tab = orc.make_coef_table(alphas, omabs)
    for t in (1, 2, 17, 500, 999):
        a = alphas[t:t + 1].reshape(1, 1)
        s = omabs[t:t + 1].reshape(1, 1)
        sm = omabs[t - 1:t].reshape(1, 1)
        sab, sabm = (1 - s.square()).sqrt(), (1 - sm.square()).sqrt()
        g0 = (1 - a) * sabm / s.square()
        g1 = sm.square() * a.sqrt() / s.square()
        g2 = 1 + (sab - 1) * (a.sqrt() + sabm) / s.square()
        sig = (sm.square() / s.square() * (1 - a)).sqrt()
        want = [1 / sab, 1 - sab, s, g0, g1, g2, sig, sab]
        for c, w in enumerate(want):
            assert tab[t, c].item() == w.item(), (t, c)


def test_philox_known_answer():
    """Random123 known-answer vectors for Philox4x32-10 (kat_vectors: zero, all-ones, pi digits)."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = orc.philox4x32_10(np.asarray(ctr, dtype=np.uint32), key)
        assert tuple(int(v) for v in got) == want


def test_philox_normal_moments():
    z = orc.philox_noise(1234, 8, 50000, 1).numpy().reshape(-1)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.08


@pytest.mark.parametrize("name", ["fphi_boston", "fphi_year", "fphi_bn_wide"])
def test_fphi_matches_reference(name):
    """The oracle's f_phi (regression/model.py:55-82, eval mode) against the reference module's own output."""
    from tests.helpers import load_raw
    d = load_raw(name)
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    got = orc.fphi_forward(sd, torch.from_numpy(d["x"]), float(d["negative_slope"]))
    assert norm_rel_err(got, d["y"]) <= TOL


def test_classification_metric_fixtures_are_self_consistent():
    """tests/golden/clsmetrics_*.npz (the reference's compute_and_store_cls_metrics on seeded samples, oracle/gen_golden.py):
    vote proportions are counts over n_z that sum to one, the stored majority vote is an arg-max of them, instance accuracy
    is the true label's proportion, interval bounds are ordered, reject flags follow the stored p-values."""
    import glob
    import os
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clsmetrics_*.npz")))
    assert len(paths) >= 4
    for p in paths:
        d = np.load(p)
        n_z, lp = int(d["n_z"]), d["label_probs"]
        assert d["gen"].shape == (int(d["rows"]), n_z, int(d["C"]))
        assert np.allclose(lp.sum(1), 1.0, atol=1e-6) and np.allclose(lp * n_z, np.round(lp * n_z), atol=1e-4)
        assert np.array_equal(lp[np.arange(len(lp)), d["majority_vote"]], lp.max(1))
        assert np.allclose(d["instance_accuracy"], lp[np.arange(len(lp)), d["labels"]])
        assert (d["ci"][:, 0] <= d["ci"][:, 1]).all()
        ok = ~np.isnan(d["ttest_pvalue"])
        assert np.array_equal(d["ttest_reject"][ok], (d["ttest_pvalue"][ok] < float(d["alpha"])).astype(np.int32))
