"""GPU parity of the classification head (CARD_VARIANT_BN_HADAMARD, streaming-GEMM back end) against the
reference's golden fixtures (F = 128: gaussmix toy C = 3, MNIST-linear C = 10) and, at the CIFAR-10 width
(F = 4096, classification/configs/cifar10.yml:15-16), against the CPU oracle on seeded random weights."""
import pytest
import torch

import card_amd
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from oracle import card_oracle as orc
from card_amd.synth import cls_state_dict
from tests.helpers import golden_names, load_golden, norm_rel_err, elem_errs, fmt_errs, record_parity, worst_errs

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"
CLS = golden_names("cls")


class RefShapedCls(torch.nn.Module):
    """Attribute layout of classification/model.py:22 ConditionalModel; the encoder is the identity here, so
    `x` passed to the shim is the fixture's x_feat (the fixtures store norm(encoder_x(x)), not the encoder)."""

    class CL(torch.nn.Module):
        def __init__(self, i, o, n):
            super().__init__()
            self.lin = torch.nn.Linear(i, o)
            self.embed = torch.nn.Embedding(n, o)

    def __init__(self, sd, guidance=True):
        super().__init__()
        F, d1 = sd["lin1.lin.weight"].shape
        C, n = sd["lin4.weight"].shape[0], sd["lin1.embed.weight"].shape[0]
        self.guidance = guidance
        self.encoder_x, self.norm = torch.nn.Identity(), torch.nn.Identity()
        self.lin1, self.lin2, self.lin3 = self.CL(d1, F, n), self.CL(F, F, n), self.CL(F, F, n)
        self.unetnorm1, self.unetnorm2, self.unetnorm3 = (torch.nn.BatchNorm1d(F) for _ in range(3))
        self.lin4 = torch.nn.Linear(F, C)
        self.load_state_dict(sd)


def cuda(t):
    return t.to(DEV)


@pytest.fixture(params=["coop", "gemm"], autouse=True)
def f128_kernel(request, monkeypatch):
    """F = 128 heads run on the cooperative persistent kernel by default (card_clscoop.hip); CARD_AMD_CLSCOOP=off (read
    when a sampler is created) keeps them on the GEMM back end.  Every test runs against both."""
    monkeypatch.setenv("CARD_AMD_CLSCOOP", "" if request.param == "coop" else "off")
    return request.param


@pytest.mark.parametrize("name", CLS)
def test_cls_loop_matches_reference(name, f128_kernel):
    g = load_golden(name)
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    T = g["T"]
    args = (m, cuda(g["x_feat"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(g["alphas"]), cuda(g["omabs"]))
    seq = card_amd.p_sample_loop(*args, noise=cuda(g["noise"]))
    assert len(seq) == T + 1 and seq[0].shape == g["y0hat"].shape
    traj = torch.stack(seq, 0).cpu()
    errs = [elem_errs(traj[k], g["traj"][k]) for k in range(T + 1)]
    worst = max(e[0] for e in errs)
    print(f"{name}: worst per-state {fmt_errs(worst_errs(errs))}")
    record_parity(name, "card_sample_cls_coop_kernel" if f128_kernel == "coop" else "wide_gemm_kernel (GEMM back end)", worst_errs(errs))
    assert worst <= TOL
    y0 = card_amd.p_sample_loop(*args, True, noise=cuda(g["noise"]))
    assert torch.equal(y0.cpu(), traj[T])
    sel = [0, 7, T - 1, T]
    got = card_amd.p_sample_loop(*args, noise=cuda(g["noise"]), states=sel)
    assert torch.equal(got.cpu(), traj[sel])


@pytest.mark.parametrize("name", CLS)
def test_cls_eps_single_step_and_reparam(name):
    g = load_golden(name)
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    T, d = g["T"], g["raw"]
    s = card_amd.sampler_for(m)
    xf = cuda(g["x_feat"])
    for t in g["probe_t"]:
        eps = s.eps_theta(xf, cuda(g["traj"][T - t]), cuda(g["y0hat"]), t).cpu()
        assert norm_rel_err(eps, d[f"eps.t{t}"]) <= TOL
    # classification/card_classification.py:645-653 style local loop: one p_sample call per timestep
    y = cuda(g["traj"][0])
    for t in range(T - 1, T - 6, -1):
        y = card_amd.p_sample(m, xf, y, cuda(g["y0hat"]), cuda(g["yTmean"]), t, cuda(g["alphas"]), cuda(g["omabs"]),
                              noise=cuda(g["noise"][T - t]))
        assert norm_rel_err(y.cpu(), g["traj"][T - t]) <= TOL
    last = card_amd.p_sample_t_1to0(m, xf, cuda(g["traj"][T - 1]), cuda(g["y0hat"]), cuda(g["yTmean"]), cuda(g["omabs"]))
    assert norm_rel_err(last.cpu(), g["traj"][T]) <= TOL
    t = T // 2
    yr = card_amd.y_0_reparam(m, xf, cuda(g["traj"][T - t]), cuda(g["y0hat"]), cuda(g["yTmean"]), torch.tensor([t]),
                              cuda(g["omabs"])).cpu()
    assert norm_rel_err(yr, d["y0reparam"]) <= TOL


def test_cls_rng_mode_and_ragged_rows():
    g = load_golden("cls_mnist_T50")
    s = Sampler.from_cls_state_dict(g["sd"], device=DEV)
    T, N0, C = g["T"], g["x_feat"].shape[0], g["y0hat"].shape[1]
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    for n in (1, 129, 300):                 # GEMM row-tile edges
        idx = torch.arange(n) % N0
        xf, y0 = cuda(g["x_feat"][idx]), cuda(g["y0hat"][idx])
        z = torch.empty(T, n, C, device=DEV)
        _lib.check(_lib.lib().card_philox_normal(5, 1000, n * C, 0, T, z.data_ptr(), None), "philox")
        a = s.run(xf, y0, y0, coef, T, seed=5, elem_offset=1000, out_mode=_lib.OUT_ALL)
        b = s.run(xf, y0, y0, coef, T, noise=z, out_mode=_lib.OUT_ALL)
        assert torch.equal(a, b)
        ref = s.run(xf, y0, y0, coef, T, noise=cuda(g["noise"][:, idx]), out_mode=_lib.OUT_ALL).cpu()
        assert norm_rel_err(ref[-1], g["traj"][-1][idx]) <= TOL
        assert torch.isfinite(a).all()


@pytest.mark.parametrize("name", golden_names("clsw"))
def test_cls_wide_widths_match_reference(name, f128_kernel):
    """CIFAR-10 width (F = 4096, C = 10: classification/configs/cifar10.yml:15-19) and a 100-class head against fixtures
    made by the reference's own ConditionalModel + p_sample_loop carrying card_amd.synth's seeded weights: the fixture
    holds the seed, the inputs and the reference's trajectory; the weights are regenerated here from the seed.
    clsw_cifar10_T1000 is that head at the config's REAL horizon (T = 1000, beta 1e-4 .. 0.02; every 50th state kept), with
    the small tensors trained so that |y| stays O(1) (oracle/gen_golden.py: gen_classification_synth_trained): 1000 steps
    of K = 4096 fp32 dot products on the GEMM back end against classification/diffusion_utils.py:119-144."""
    if f128_kernel == "gemm":
        pytest.skip("F > 128 runs on the GEMM back end either way: one pass is enough")
    g = load_golden(name)
    T, d = g["T"], g["raw"]
    s = Sampler.from_cls_state_dict(g["sd"], device=DEV)
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    xf, y0 = cuda(g["x_feat"]), cuda(g["y0hat"])
    idx = g["traj_idx"]
    got = s.run(xf, y0, cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), out_mode=_lib.OUT_SELECT, states=idx).cpu()
    errs = [elem_errs(got[i], g["traj"][i]) for i in range(len(idx))]
    print(f"{name}: worst per-state {fmt_errs(worst_errs(errs))}")
    record_parity(name, "wide_gemm_kernel (GEMM back end)", worst_errs(errs))
    assert max(e[0] for e in errs) <= TOL
    if idx == list(range(T + 1)):
        assert torch.equal(got, s.run(xf, y0, cuda(g["yTmean"]), coef, T, noise=cuda(g["noise"]), out_mode=_lib.OUT_ALL).cpu())
    full = dict(zip(idx, g["traj"]))
    for t in g["probe_t"]:
        if T - t not in full:      # a fixture that keeps every 50th state holds no y_t for the probe at t = T - 1
            continue
        eps = s.eps_theta(xf, cuda(full[T - t]), y0, t).cpu()
        assert norm_rel_err(eps, d[f"eps.t{t}"]) <= TOL, t
    if "last" in d:          # p_sample_t_1to0 from the reference's y_1 (teacher-forced last step)
        one = s.run(xf, y0, cuda(g["yTmean"]), coef, T, t_hi=0, t_lo=0, y_init=cuda(full[T - 1])).cpu()
        assert norm_rel_err(one, d["last"]) <= TOL


@pytest.mark.parametrize("F,C", [(4096, 10), (256, 100)])
def test_cls_wide_widths_against_oracle(F, C):
    """CIFAR-10 width (F = 4096, C = 10) and a 100-class head: HIP vs the CPU oracle, seeded random weights, more rows
    than the reference fixtures above hold (72: a ragged GEMM row tile)."""
    T, N = 6, 72
    sd = cls_state_dict(F, C, T, 3)
    w = orc.ClsWeights.from_state_dict(sd)
    gen = torch.Generator().manual_seed(4)
    xf = torch.randn(N, F, generator=gen)
    y0 = torch.softmax(torch.randn(N, C, generator=gen) * 2, 1)
    noise = torch.randn(T, N, C, generator=gen)
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    torch.set_num_threads(16)
    want = torch.stack(orc.p_sample_loop(w, xf, y0, y0, T, alphas, omabs, noise), 0)
    s = Sampler.from_cls_state_dict(sd, device=DEV)
    got = s.run(cuda(xf), cuda(y0), cuda(y0), cuda(make_coef_table(alphas, omabs)), T, noise=cuda(noise),
                out_mode=_lib.OUT_ALL).cpu()
    worst = max(norm_rel_err(got[k], want[k]) for k in range(T + 1))
    print(f"F={F} C={C}: worst per-state norm-rel err {worst:.3e}")
    assert worst <= TOL
    eps = s.eps_theta(cuda(xf), cuda(want[2]), cuda(y0), 3).cpu()
    assert norm_rel_err(eps, orc.eps_theta(w, xf, want[2], y0, 3)) <= TOL


def test_cls_in_kernel_row_tiling():
    from card_amd.distributed import tile_rows
    g = load_golden("cls_mnist_T50")
    s = Sampler.from_cls_state_dict(g["sd"], device=DEV)
    coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
    rows, n_z, T = 5, 40, g["T"]
    xf, y0 = g["x_feat"][:rows], g["y0hat"][:rows]
    a = s.run(cuda(xf), cuda(y0), cuda(y0), coef, T, seed=11, n_z=n_z, out_mode=_lib.OUT_ALL)
    b = s.run(cuda(tile_rows(xf, n_z)), cuda(tile_rows(y0, n_z)), cuda(tile_rows(y0, n_z)), coef, T, seed=11,
              out_mode=_lib.OUT_ALL)
    assert a.shape == (T + 1, rows * n_z, y0.shape[1]) and torch.equal(a, b)


def test_imagenet_linear_head_matches_reference():
    """classification/imagenet/card/builder.py head (no BN / softplus, T = 10, beta 0.01..0.95) through the shim,
    against the fixture generated by the ImageNet tree's own p_sample_loop."""
    g = load_golden("inet_head_T10")
    sd, T = g["sd"], g["T"]
    F, d1 = sd["lin1.lin.weight"].shape
    C, n = sd["lin3.weight"].shape[0], sd["lin1.embed.weight"].shape[0]

    class Head(torch.nn.Module):           # attribute layout of builder.py:69-84
        def __init__(self):
            super().__init__()
            self.guidance = True
            self.linx = torch.nn.Linear(F, F)
            self.lin1, self.lin2 = RefShapedCls.CL(d1, F, n), RefShapedCls.CL(F, F, n)
            self.lin3 = torch.nn.Linear(F, C)
            self.load_state_dict(sd)

    class Card(torch.nn.Module):           # builder.py:8 wrapper: model(x, y, t, yhat, mode='theta')
        def __init__(self):
            super().__init__()
            self.model_theta = Head()

    m = Card().to(DEV).eval()
    args = (m, cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(g["alphas"]), cuda(g["omabs"]))
    seq = card_amd.p_sample_loop(*args, noise=cuda(g["noise"]))
    traj = torch.stack(seq, 0).cpu()
    worst = max(norm_rel_err(traj[k], g["traj"][k]) for k in range(T + 1))
    print(f"inet head: worst per-state norm-rel err {worst:.3e}")
    assert worst <= TOL
    y0 = card_amd.p_sample_loop(*args, True, noise=cuda(g["noise"]))     # imagenet/main.py:482 only_last_sample
    assert torch.equal(y0.cpu(), traj[T])
    s = card_amd.sampler_for(m)
    eps = s.eps_theta(cuda(g["x_feat"]), cuda(g["traj"][T - 3]), cuda(g["y0hat"]), 3).cpu()
    assert norm_rel_err(eps, g["raw"]["eps.t3"]) <= TOL
    # save_prediction's repeated sampling (imagenet/main.py:555-566) as one call: (B, n_repeats, C), repeat r of image b
    # equals a separate p_sample_loop call fed the noise of chain (b, r)
    B, R = g["y0hat"].shape[0], 5
    nz = torch.randn(T, B * R, C, generator=torch.Generator().manual_seed(2))
    rep = card_amd.p_sample_loop_repeated(*args, R, noise=cuda(nz))
    assert rep.shape == (B, R, C)
    for r in range(R):
        one = card_amd.p_sample_loop(*args, True, noise=cuda(nz[:, r::R].contiguous()))
        assert torch.equal(rep[:, r], one), r
    again = card_amd.p_sample_loop_repeated(*args, R, seed=3)
    assert torch.equal(again, card_amd.p_sample_loop_repeated(*args, R, seed=3)) and again.shape == (B, R, C)


@pytest.mark.parametrize("num_ts", [None, [1, 8, 9, 30, 51], [51], [17]])
def test_loop_with_eval_hook_matches_full_trajectory(num_ts):
    """classification/card_classification.py:628-654: states handed to the callback on the fly are bit-equal to
    the same states of one full run with the same seed, wherever the segments are cut."""
    g = load_golden("cls_mnist_T50")
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    T = g["T"]
    args = (m, cuda(g["x_feat"][:6]), cuda(g["y0hat"][:6]), cuda(g["yTmean"][:6]), T, cuda(g["alphas"]), cuda(g["omabs"]))
    full = torch.stack(card_amd.p_sample_loop(*args, seed=42, n_z=7), 0)
    seen = {}
    y0 = card_amd.p_sample_loop_with_eval(*args, lambda y, num_t: seen.__setitem__(num_t, y.clone()), num_ts=num_ts,
                                          seed=42, n_z=7)
    want = list(range(1, T + 2)) if num_ts is None else num_ts
    assert sorted(seen) == sorted(want)
    for num_t, y in seen.items():
        assert torch.equal(y, full[num_t - 1]), num_t
    assert torch.equal(y0, full[T])


def test_f128_cooperative_kernel_equals_gemm_backend_in_rng_mode(monkeypatch):
    """Same seed, same (chain, class) Philox elements: the two back ends agree to rounding on a ragged multi-tile batch
    with in-kernel row tiling, for C = 10 and C = 3, with and without guidance columns."""
    for name, guidance in (("cls_mnist_T50", True), ("cls_gaussmix_T50", True)):
        g = load_golden(name)
        T = g["T"]
        coef = cuda(make_coef_table(g["alphas"], g["omabs"]))
        rows, n_z = 37, 50                                  # 1850 chains = 58 tiles, last one ragged
        idx = torch.arange(rows) % g["x_feat"].shape[0]
        xf, y0 = cuda(g["x_feat"][idx]), cuda(g["y0hat"][idx])
        outs = []
        for mode in ("1", "0"):
            monkeypatch.setenv("CARD_AMD_CLSCOOP", "" if mode == "1" else "off")
            s = Sampler.from_cls_state_dict(g["sd"], guidance=guidance, device=DEV)
            assert (s._workspace(64)[2] == 0) == (mode == "1")      # the cooperative kernel needs no workspace
            outs.append(s.run(xf, y0, y0, coef, T, seed=31, elem_offset=12345, n_z=n_z, out_mode=_lib.OUT_SELECT,
                              states=[0, 1, T // 2, T]).cpu())
        assert torch.equal(outs[0][0], outs[1][0])          # y_T = draw 0 + mean: identical
        err = max(norm_rel_err(outs[0][k], outs[1][k]) for k in range(4))
        print(f"{name}: cooperative vs GEMM back end {err:.3e}")
        assert err <= TOL


@pytest.mark.parametrize("F,C", [(128, 10), (128, 32), (128, 5), (256, 10), (128, 1), (128, 8), (128, 9), (128, 16), (128, 17)])
def test_cls_without_guidance_and_class_count_edges(F, C):
    """ConditionalModel(guidance=False) (classification/model.py:31: lin1 sees y_t alone) and class counts at the
    edges of the cooperative kernel's 8-class groups and of its two lin4 forms (four-block 16 x 16 x 1 MFMA up to 16
    classes, 32 x 32 x 2 above), against the oracle on seeded weights."""
    T, N = 8, 70
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    for guidance in (False, True):
        sd = cls_state_dict(F, C, T, 9, guidance=guidance)
        w = orc.ClsWeights.from_state_dict(sd, guidance=guidance)
        gen = torch.Generator().manual_seed(10)
        xf = torch.randn(N, F, generator=gen)
        y0 = torch.softmax(torch.randn(N, C, generator=gen) * 2, 1)
        noise = torch.randn(T, N, C, generator=gen)
        want = torch.stack(orc.p_sample_loop(w, xf, y0, y0, T, alphas, omabs, noise), 0)
        s = Sampler.from_cls_state_dict(sd, guidance=guidance, device=DEV)
        got = s.run(cuda(xf), cuda(y0), cuda(y0), coef, T, noise=cuda(noise), out_mode=_lib.OUT_ALL).cpu()
        worst = max(norm_rel_err(got[k], want[k]) for k in range(T + 1))
        assert worst <= TOL, (guidance, worst)
        eps = s.eps_theta(cuda(xf), cuda(want[2]), cuda(y0), 3).cpu()
        assert norm_rel_err(eps, orc.eps_theta(w, xf, want[2], y0, 3)) <= TOL


def test_hoisted_features_follow_the_x_object_not_its_address():
    """x_features caches norm(encoder_x(x)) per x OBJECT: a per-batch `x = xb.to(dev)` that is freed and whose address the
    allocator hands to the next batch must get its own features."""
    g = load_golden("cls_mnist_T50")
    sd = dict(g["sd"])
    m = RefShapedCls(sd).to(DEV).eval()
    m.encoder_x = torch.nn.Linear(128, 128).to(DEV)            # a real (step-invariant) encoder instead of the identity
    T = g["T"]
    al, om = cuda(g["alphas"]), cuda(g["omabs"])
    outs, ptrs = [], []
    for k in range(3):
        xb = g["x_feat"] * (1.0 + k)                           # three "batches" of the same shape
        x = xb.to(DEV)
        ptrs.append(x.data_ptr())
        got = card_amd.p_sample_loop(m, x, cuda(g["y0hat"]), cuda(g["yTmean"]), T, al, om, True, noise=cuda(g["noise"]))
        with torch.no_grad():
            xf = m.encoder_x(x)
        s = card_amd.sampler_for(m)
        want = s.run(xf, cuda(g["y0hat"]), cuda(g["yTmean"]), cuda(make_coef_table(g["alphas"], g["omabs"])), T,
                     noise=cuda(g["noise"]))
        assert torch.equal(got, want), k
        outs.append(got)
        del x, xf
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    print("x addresses:", ptrs)


def test_bn_model_in_train_mode_is_refused():
    g = load_golden("cls_mnist_T50")
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    args = (cuda(g["x_feat"]), cuda(g["y0hat"]), cuda(g["yTmean"]), g["T"], cuda(g["alphas"]), cuda(g["omabs"]))
    card_amd.p_sample_loop(m, *args, True, seed=1)              # cached while in eval mode ...
    with pytest.raises(_lib.CardError, match="eval mode"):      # ... and still refused once switched to train mode
        card_amd.p_sample_loop(m.train(), *args, True, seed=1)


@pytest.mark.parametrize("rows,n_z,C", [(250, 100, 10), (173, 97, 3), (40, 100, 10), (120, 100, 20)])
def test_f128_time_sliced_and_plain_forms_are_bit_identical(rows, n_z, C, monkeypatch, f128_kernel):
    """card_clscoop.hip: the round-robin time slicing over two one-tile workgroups per CU (782 tiles = the MNIST test
    minibatch of 250 images x 100 z, classification/configs/mnist.yml) against whole loops in one launch
    (CARD_AMD_CLSCOOP=noslice, read when a sampler is created: the kernel then walks the tiles in rounds), and the compact
    PAIR form (in-kernel row tiling, up to 16 classes) against the plain PAIR form (the same chains pre-tiled by the
    caller, or more than 16 classes): same bits in every output mode, also through the entry point that allocates
    nothing."""
    if f128_kernel != "coop":
        pytest.skip("cooperative kernel only")
    T = 130
    sd = cls_state_dict(128, C, T, 5)
    gen = torch.Generator().manual_seed(6)
    xf = cuda(torch.randn(rows, 128, generator=gen))
    y0 = cuda(torch.softmax(torch.randn(rows, C, generator=gen) * 2, 1))
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.1)
    coef = cuda(make_coef_table(alphas, omabs))
    tiled = lambda t: t.repeat_interleave(n_z, 0).contiguous()          # chain order (row, z), as the reference tiles
    outs = {}
    for name, env, pre in (("sliced", "", False), ("plain", "noslice", False), ("sliced_pretiled", "", True),
                           ("plain_pretiled", "noslice", True)):
        monkeypatch.setenv("CARD_AMD_CLSCOOP", env)
        s = Sampler.from_cls_state_dict(sd, device=DEV)
        a, b, nz = (tiled(xf), tiled(y0), 1) if pre else (xf, y0, n_z)
        kw = dict(seed=12, elem_offset=3, n_z=nz)
        last = s.run(a, b, b, coef, T, **kw)
        sel = s.run(a, b, b, coef, T, out_mode=_lib.OUT_SELECT, states=[0, 1, 64, 65, T], **kw)
        seg = s.run(a, b, b, coef, T, t_hi=100, t_lo=7, y_init=sel[2], **kw)
        outs[name] = (last, sel, seg)
        if name == "sliced":
            form = s.run(a, b, b, coef, T, plan=True, **kw)["kernel"]
            want = "ONE" if rows * n_z <= 256 * 32 else ("PAIR compact" if C <= 16 else "<PAIR>")      # 256 CUs
            assert want in form, form
            assert torch.equal(s.run(a, b, b, coef, T, static_scratch=True, **kw), last)
    for name in ("plain", "sliced_pretiled", "plain_pretiled"):
        for a, b in zip(outs["sliced"], outs[name]):
            assert torch.equal(a, b) and torch.isfinite(a).all(), name
    assert torch.equal(outs["sliced"][0], outs["sliced"][1][-1])
    # and against the oracle under injected noise on the first and the last input row's chains
    w = orc.ClsWeights.from_state_dict(sd, guidance=True)
    monkeypatch.setenv("CARD_AMD_CLSCOOP", "")
    s = Sampler.from_cls_state_dict(sd, device=DEV)
    N = rows * n_z
    noise = torch.randn(T, N, C, generator=gen)
    got = s.run(xf, y0, y0, coef, T, noise=cuda(noise), n_z=n_z).cpu()
    for r in (0, rows - 1):
        sl_ = slice(r * n_z, r * n_z + 16)
        tile = lambda t: t[r:r + 1].cpu().repeat(16, 1)
        want = orc.p_sample_loop(w, tile(xf), tile(y0), tile(y0), T, alphas, omabs, noise[:, sl_], only_last_sample=True)
        err = norm_rel_err(got[sl_], want)
        print(f"sliced cls kernel rows={rows} n_z={n_z} C={C}, row {r}: norm-rel err vs oracle {err:.2e}")
        assert err <= TOL


def test_per_timestep_shim_loop_equals_fused_loop_and_pins_nothing():
    """The reference's p_sample_loop_only_y_0 shape (classification/card_classification.py:656-671: T - 1 calls of
    p_sample, one of p_sample_t_1to0) through the shim's cached call context: with the same seed it reproduces the fused
    loop bit for bit (noise is keyed by (seed, element, draw)), picks up an in-place weight edit between two loops, and
    does not keep the caller's tensors alive."""
    import gc
    import weakref
    from card_amd import diffusion_utils as du
    g = load_golden("cls_mnist_T50")
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    T = g["T"]
    al, om = cuda(g["alphas"]), cuda(g["omabs"])
    xf, y0 = cuda(g["x_feat"]), cuda(g["y0hat"])

    def stepwise(seed):
        cur = card_amd.p_sample_loop(m, xf, y0, y0, T, al, om, seed=seed, states=[0])[0]
        for t in reversed(range(1, T)):
            cur = card_amd.p_sample(m, xf, cur, y0, y0, t, al, om, seed=seed)
        return card_amd.p_sample_t_1to0(m, xf, cur, y0, y0, om)

    want = card_amd.p_sample_loop(m, xf, y0, y0, T, al, om, True, seed=77)
    assert torch.equal(stepwise(77), want)
    assert du._tls.ctx is not None and du._tls.ctx.matches((m, xf, y0, y0, al, om))
    # an EMA-style copy behind the version counters between two loops: the first step of the next loop re-packs
    with torch.no_grad():
        m.lin4.weight.data.copy_(m.lin4.weight.data * 1.25)
    got = stepwise(77)
    assert not torch.equal(got, want) and torch.equal(got, card_amd.p_sample_loop(m, xf, y0, y0, T, al, om, True, seed=77))
    # the context holds the caller's tensors weakly: dropping x frees it and clears the context
    ref = weakref.ref(xf)
    del xf
    gc.collect()
    assert ref() is None and du._tls.ctx.refs is None and du._tls.ctx.keep is None          # released: owns nothing, can never match


def test_per_timestep_loop_runs_ahead_and_survives_pattern_breaks(monkeypatch):
    """The look-ahead of the per-timestep entry points (card_amd/diffusion_utils.py: LOOKAHEAD_MAX_STEPS): the reference's
    T-call loop (classification/card_classification.py:656-671) issues O(log T) launches instead of T, every state it
    hands out is bit-equal to the one-launch-per-call result, and a caller that edits y in place, re-feeds an older
    state, or passes its own noise falls back to its own launch with the same bits."""
    from card_amd import diffusion_utils as du
    g = load_golden("cls_mnist_T50")
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    T = g["T"]
    al, om = cuda(g["alphas"]), cuda(g["omabs"])
    xf, y0 = cuda(g["x_feat"]), cuda(g["y0hat"])
    yT = card_amd.p_sample_loop(m, xf, y0, y0, T, al, om, seed=5, states=[0])[0]

    def loop(edit_at=None):
        du._drop_step_ctx()
        states, cur = [], yT
        for t in reversed(range(1, T)):
            cur = card_amd.p_sample(m, xf, cur, y0, y0, t, al, om, seed=5)
            if t == edit_at:
                cur.mul_(0.5)                     # in place: the version counter moves, what was computed ahead is stale
            states.append(cur)
        states.append(card_amd.p_sample_t_1to0(m, xf, cur, y0, y0, om))
        return states

    calls = []
    real = _lib.lib().card_sampler_run

    def counting(h, a, st):
        calls.append(1)
        return real(h, a, st)

    monkeypatch.setattr(du, "LOOKAHEAD_MAX_STEPS", 0)
    plain = loop()
    plain_edit = loop(edit_at=T // 2)
    monkeypatch.setattr(du, "LOOKAHEAD_MAX_STEPS", 16)
    monkeypatch.setattr(_lib.lib(), "card_sampler_run", counting, raising=False)
    ahead = loop()
    n_launches = len(calls)
    assert n_launches <= 8, n_launches            # 1 + 1 (2) + 1 (4) + 1 (8) + 2 (16, 16) + 1 (the rest) for T = 50
    assert all(torch.equal(a, b) for a, b in zip(ahead, plain))
    assert torch.equal(ahead[-1], card_amd.p_sample_loop(m, xf, y0, y0, T, al, om, True, seed=5))
    calls.clear()
    ahead_edit = loop(edit_at=T // 2)
    assert all(torch.equal(a, b) for a, b in zip(ahead_edit, plain_edit))
    assert n_launches < len(calls) < T            # the edit cost one fresh start of the doubling, not the look-ahead
    # re-feeding an older state, and explicit noise, run on their own
    du._drop_step_ctx()
    a1 = card_amd.p_sample(m, xf, yT, y0, y0, T - 1, al, om, seed=5)
    a2 = card_amd.p_sample(m, xf, a1, y0, y0, T - 2, al, om, seed=5)         # runs T-2 and T-3
    again = card_amd.p_sample(m, xf, a1, y0, y0, T - 2, al, om, seed=5)      # not a continuation of a2
    assert torch.equal(again, a2) and torch.equal(a2, plain[1])
    z = torch.randn_like(a2)
    nz = card_amd.p_sample(m, xf, again, y0, y0, T - 3, al, om, noise=z)
    du._drop_step_ctx()
    monkeypatch.setattr(du, "LOOKAHEAD_MAX_STEPS", 0)
    assert torch.equal(nz, card_amd.p_sample(m, xf, again, y0, y0, T - 3, al, om, noise=z))


def test_wrapped_models_module_holders_sample_like_the_bare_model():
    """classification/imagenet/main.py:222-237 hands p_sample_loop a DistributedDataParallel / DataParallel wrapper around
    the CARD model (:482, :557); round 2 raised 'does not recognise DistributedDataParallel' there.  The ImageNet head
    inside CARD inside DataParallel, and the regression / classification models inside DataParallel, must give the bare
    model's bits, share its packed sampler (the cache is keyed on the inner module) and see its weight edits."""
    g = load_golden("inet_head_T10")
    sd, T = g["sd"], g["T"]
    F, d1 = sd["lin1.lin.weight"].shape
    C, n = sd["lin3.weight"].shape[0], sd["lin1.embed.weight"].shape[0]

    class Head(torch.nn.Module):           # attribute layout of builder.py:69-84
        def __init__(self):
            super().__init__()
            self.guidance = True
            self.linx = torch.nn.Linear(F, F)
            self.lin1, self.lin2 = RefShapedCls.CL(d1, F, n), RefShapedCls.CL(F, F, n)
            self.lin3 = torch.nn.Linear(F, C)
            self.load_state_dict(sd)

    class Card(torch.nn.Module):           # builder.py:8: model_theta next to a (frozen) backbone
        def __init__(self):
            super().__init__()
            self.model_phi = torch.nn.Linear(F, F)
            self.model_theta = Head()

    card = Card().to(DEV).eval()
    dp = torch.nn.DataParallel(card)       # main.py:237
    rest = (cuda(g["x"]), cuda(g["y0hat"]), cuda(g["yTmean"]), T, cuda(g["alphas"]), cuda(g["omabs"]))
    seq = card_amd.p_sample_loop(dp, *rest, noise=cuda(g["noise"]))
    traj = torch.stack(seq, 0).cpu()
    worst = max(norm_rel_err(traj[k], g["traj"][k]) for k in range(T + 1))
    print(f"inet head inside CARD inside DataParallel: worst per-state norm-rel err {worst:.3e}")
    assert worst <= TOL
    assert card_amd.sampler_for(dp) is card_amd.sampler_for(card) is card_amd.sampler_for(card.model_theta)
    y0 = card_amd.p_sample_loop(dp, *rest, only_last_sample=True, noise=cuda(g["noise"]))      # main.py:482
    assert torch.equal(y0.cpu(), traj[T])
    rep = card_amd.p_sample_loop_repeated(dp, *rest, 5, seed=3)                                 # main.py:555-566
    assert torch.equal(rep, card_amd.p_sample_loop_repeated(card, *rest, 5, seed=3)) and rep.shape == (g["y0hat"].shape[0], 5, C)
    # an edit of the backbone leaves the packed head alone; an edit of the head (behind the version counters) re-packs
    before = card_amd.sampler_for(dp)
    with torch.no_grad():
        card.model_phi.weight.data.mul_(2.0)
    assert card_amd.sampler_for(dp) is before
    with torch.no_grad():
        card.model_theta.lin3.weight.data.mul_(1.5)
    assert card_amd.sampler_for(dp) is not before
    assert not torch.equal(card_amd.p_sample_loop(dp, *rest, only_last_sample=True, noise=cuda(g["noise"])).cpu(), traj[T])
    card_amd.invalidate(dp)
    # the classification model and the per-timestep entry points through a wrapper
    gc_ = load_golden("cls_mnist_T50")
    m = RefShapedCls(gc_["sd"]).to(DEV).eval()
    mdp = torch.nn.DataParallel(m)
    Tc = gc_["T"]
    al, om = cuda(gc_["alphas"]), cuda(gc_["omabs"])
    xf, yh = cuda(gc_["x_feat"]), cuda(gc_["y0hat"])
    want = card_amd.p_sample_loop(m, xf, yh, yh, Tc, al, om, True, seed=5)
    assert torch.equal(card_amd.p_sample_loop(mdp, xf, yh, yh, Tc, al, om, True, seed=5), want)
    cur = card_amd.p_sample_loop(mdp, xf, yh, yh, Tc, al, om, seed=5, states=[0])[0]
    for t in reversed(range(1, Tc)):
        cur = card_amd.p_sample(mdp, xf, cur, yh, yh, t, al, om, seed=5)
    assert torch.equal(card_amd.p_sample_t_1to0(mdp, xf, cur, yh, yh, om), want)
    # a regression model inside a one-rank DistributedDataParallel on RCCL
    import torch.distributed as dist
    from tests.test_gpu_parity import RefShapedGuided
    gr = load_golden("reg_boston_T50")
    r = RefShapedGuided(gr["sd"], gr["cat_x"], gr["cat_y_pred"]).to(DEV)
    made = not dist.is_initialized()
    if made:
        import tempfile
        dist.init_process_group("nccl", store=dist.FileStore(tempfile.mktemp(prefix="card_ddp_"), 1), rank=0, world_size=1,
                                device_id=torch.device(DEV))
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(r, device_ids=[torch.device(DEV).index], find_unused_parameters=True)
        a = (cuda(gr["x"]), cuda(gr["y0hat"]), cuda(gr["yTmean"]), gr["T"], cuda(gr["alphas"]), cuda(gr["omabs"]))
        got = torch.stack(card_amd.p_sample_loop(ddp, *a, noise=cuda(gr["noise"])), 0).cpu()
        ref_ = torch.stack(card_amd.p_sample_loop(r, *a, noise=cuda(gr["noise"])), 0).cpu()
        assert torch.equal(got, ref_)
        assert max(norm_rel_err(got[k], gr["traj"][i]) for i, k in enumerate(gr["traj_idx"])) <= TOL
    finally:
        if made:
            dist.destroy_process_group()


@pytest.mark.parametrize("F,C", [(2048, 100), (2048, 1000)])
def test_imagenet_head_at_its_real_dimensions_against_oracle(F, C):
    """imagenet/card/builder.py:69-84 at the sizes the ImageNet tree runs (feature_dim 2048 = ResNet-50 features; 100
    classes is its default, 1000 the full label set): layer 1 with K = 2C = 200 / 2000 columns of state, the head as a
    GEMM launch (more than 32 classes), T = 10 with the tree's beta range (main.py:83-88).  Seeded random weights against
    the CPU oracle (the fixture inet_head_T10 pins the same code path at 20 classes against the reference itself)."""
    T, N = 10, 24
    gen = torch.Generator().manual_seed(F + C)
    lin = lambda o, i, s=1.0: (torch.randn(o, i, generator=gen) * (s / i ** 0.5), torch.randn(o, generator=gen) * 0.1)
    sd = {}
    for name, (o, i) in {"lin1.lin": (F, 2 * C), "lin2.lin": (F, F), "lin3": (C, F)}.items():
        w, b = lin(o, i, 0.3 if name == "lin3" else 1.0)
        sd[name + ".weight"], sd[name + ".bias"] = w, b
    sd["lin1.embed.weight"] = torch.rand(T, F, generator=gen)
    sd["lin2.embed.weight"] = torch.rand(T, F, generator=gen)
    w = orc.LinearHeadWeights.from_state_dict(sd)
    xf = torch.randn(N, F, generator=gen)
    y0 = torch.softmax(torch.randn(N, C, generator=gen) * 2, 1)
    noise = torch.randn(T, N, C, generator=gen)
    alphas, _, omabs = orc.make_schedule("linear", T, 0.01, 0.95)
    want = torch.stack(orc.p_sample_loop(w, xf, y0, y0, T, alphas, omabs, noise), 0)
    s = Sampler.from_imagenet_head_state_dict(sd, device=DEV)
    coef = cuda(make_coef_table(alphas, omabs))
    got = s.run(cuda(xf), cuda(y0), cuda(y0), coef, T, noise=cuda(noise), out_mode=_lib.OUT_ALL).cpu()
    worst = max(norm_rel_err(got[k], want[k]) for k in range(T + 1))
    print(f"inet head F={F} C={C}: worst per-state norm-rel err {worst:.3e}, max|y| {want.abs().max():.2f}")
    assert worst <= TOL
    # in-kernel repeats (save_prediction, main.py:555-566) == pre-tiled rows
    a = s.run(cuda(xf[:3]), cuda(y0[:3]), cuda(y0[:3]), coef, T, seed=7, n_z=9)
    b = s.run(cuda(xf[:3].repeat_interleave(9, 0)), cuda(y0[:3].repeat_interleave(9, 0)), cuda(y0[:3].repeat_interleave(9, 0)), coef, T, seed=7)
    assert torch.equal(a, b) and torch.isfinite(a).all()


@pytest.mark.parametrize("C", [1, 9, 16, 17, 32])
def test_f128_two_workgroups_per_cu_form_over_class_counts(C, f128_kernel):
    """The default form above one tile per CU (two one-tile workgroups per CU at 256 registers per wave, lin1 / lin4
    fragments re-read from L2) at class counts on both sides of the 8-class groups and of the two lin4 forms: same bits
    as the one-workgroup-per-CU form (the same chains in two calls of at most one tile per CU), in RNG mode and for eps_theta, and <= 1e-4 from the
    oracle under injected noise on the chains of the first and the last input row."""
    if f128_kernel != "coop":
        pytest.skip("cooperative kernel only")
    T, rows, n_z = 12, 90, 100                               # 9 000 chains = 282 tiles > 256 CUs
    sd = cls_state_dict(128, C, T, 11)
    gen = torch.Generator().manual_seed(C)
    xf = cuda(torch.randn(rows, 128, generator=gen))
    y0 = cuda(torch.softmax(torch.randn(rows, C, generator=gen) * 2, 1))
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    outs = {}
    # the whole batch (282 tiles: PAIR forms, two workgroups per CU) against the same chains as two calls of 141 tiles
    # each (at most one tile per CU: the ONE form, every fragment in registers), element offsets continuing
    s = Sampler.from_cls_state_dict(sd, device=DEV)
    rep = lambda t: t.repeat_interleave(n_z, 0)
    kw = dict(seed=5, n_z=n_z, out_mode=_lib.OUT_ALL)
    whole = (s.run(xf, y0, y0, coef, T, elem_offset=7, **kw), s.eps_theta(rep(xf), rep(y0), rep(y0), 4))
    h = rows // 2
    assert "PAIR" in s.run(xf, y0, y0, coef, T, elem_offset=7, plan=True, **kw)["kernel"]
    assert "ONE" in s.run(xf[:h], y0[:h], y0[:h], coef, T, elem_offset=7, plan=True, **kw)["kernel"]
    halves = (torch.cat([s.run(xf[:h], y0[:h], y0[:h], coef, T, elem_offset=7, **kw),
                         s.run(xf[h:], y0[h:], y0[h:], coef, T, elem_offset=7 + h * n_z * C, **kw)], 1),
              torch.cat([s.eps_theta(rep(xf[:h]), rep(y0[:h]), rep(y0[:h]), 4), s.eps_theta(rep(xf[h:]), rep(y0[h:]), rep(y0[h:]), 4)], 0))
    assert torch.equal(whole[0], halves[0]) and torch.equal(whole[1], halves[1])
    assert torch.isfinite(whole[0]).all()
    w = orc.ClsWeights.from_state_dict(sd, guidance=True)
    N = rows * n_z
    noise = torch.randn(T, N, C, generator=gen)
    got = s.run(xf, y0, y0, coef, T, noise=cuda(noise), n_z=n_z, out_mode=_lib.OUT_ALL).cpu()
    for r in (0, rows - 1):
        sl_ = slice(r * n_z, r * n_z + 20)
        tile = lambda t: t[r:r + 1].cpu().repeat(20, 1)
        want = torch.stack(orc.p_sample_loop(w, tile(xf), tile(y0), tile(y0), T, alphas, omabs, noise[:, sl_]), 0)
        worst = max(norm_rel_err(got[k][sl_], want[k]) for k in range(T + 1))
        assert worst <= TOL, (C, r, worst)


@pytest.mark.parametrize("n_z", [31, 32, 33, 47])
def test_f128_compact_form_at_the_edge_of_two_rows_per_tile(n_z, f128_kernel):
    """The compact two-workgroup form keeps the hoisted layer-1 tile and x_feat as the TWO columns of the input rows a
    32-chain tile spans, which holds from n_z = 32 on (the host selects the plain form below that): same bits as the
    one-workgroup-per-CU form on both sides of that edge, every input row different, ragged last tile."""
    if f128_kernel != "coop":
        pytest.skip("cooperative kernel only")
    T, rows, C = 9, 301, 10                                   # > 8 192 chains, rows * n_z not a multiple of 32
    sd = cls_state_dict(128, C, T, 13)
    gen = torch.Generator().manual_seed(n_z)
    xf = cuda(torch.randn(rows, 128, generator=gen))
    y0 = cuda(torch.softmax(torch.randn(rows, C, generator=gen) * 2, 1))
    mu = cuda(torch.randn(rows, C, generator=gen))
    alphas, _, omabs = orc.make_schedule("linear", T, 1e-4, 0.3)
    coef = cuda(make_coef_table(alphas, omabs))
    s = Sampler.from_cls_state_dict(sd, device=DEV)
    kw = dict(seed=3, n_z=n_z, out_mode=_lib.OUT_ALL)
    per = max(1, 8192 // n_z)                                 # rows per call that stay within one tile per CU: the ONE form
    parts = [s.run(xf[r:r + per], y0[r:r + per], mu[r:r + per], coef, T, elem_offset=r * n_z * C, **kw) for r in range(0, rows, per)]
    outs = [s.run(xf, y0, mu, coef, T, **kw), torch.cat(parts, 1)]
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
