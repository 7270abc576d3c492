"""GPU parity of the guidance network f_phi inside the sampling call.

Reference: regression/model.py:55-82 (DeterministicFeedForwardNeuralNetwork), regression/card_regression.py:111-124
(compute_guiding_prediction), :1021-1044 (f_phi, tiling, p_sample_loop in the test loop).  Bar: the f_phi output within
1e-4 norm-relative of the reference module's own output (fixtures fphi_*), and the fused call bit-equal to the two-call
form (f_phi forward, then the sampler on its result) and within 1e-4 of the oracle."""
import pytest
import torch

import card_amd
from card_amd import _lib
from card_amd.guidance import GuidanceNet, guidance_for
from card_amd.sampler import Sampler, make_coef_table
from oracle import card_oracle as orc
from tests.helpers import load_golden, load_raw, norm_rel_err
from tests.test_gpu_parity import RefShapedGuided

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def ref_shaped_fphi(d):
    """An nn.Module with the layout of regression/model.py:55-82 (`.network` Sequential) carrying a fixture's weights."""
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    idx = sorted({int(k.split(".")[1]) for k in sd})
    n_mod = max(idx) + 1
    layers = [torch.nn.Identity() for _ in range(n_mod)]
    for i in idx:
        w = sd[f"network.{i}.weight"]
        layers[i] = torch.nn.Linear(w.shape[1], w.shape[0]) if w.dim() == 2 else torch.nn.BatchNorm1d(w.shape[0])
    for i, m in enumerate(layers):       # the activation / dropout slots between the parametrised layers
        if isinstance(m, torch.nn.Identity):
            prev = layers[i - 1]
            layers[i] = torch.nn.LeakyReLU(float(d["negative_slope"])) if not isinstance(prev, torch.nn.LeakyReLU) \
                else torch.nn.Dropout(0.2)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.network = torch.nn.Sequential(*layers)

        def forward(self, x):
            return self.network(x)

    net = Net()
    net.load_state_dict(sd)
    return net.eval(), sd


@pytest.mark.parametrize("name", ["fphi_boston", "fphi_year", "fphi_bn_wide"])
def test_fphi_forward_matches_reference_fixture(name):
    d = load_raw(name)
    net, sd = ref_shaped_fphi(d)
    with torch.no_grad():      # the stand-in module reproduces the reference module's output: the layout is right
        assert norm_rel_err(net(torch.from_numpy(d["x"])), d["y"]) <= 2e-6
    g = GuidanceNet.from_module(net.to(DEV))
    got = g.forward(torch.from_numpy(d["x"]).to(DEV)).cpu()
    err = norm_rel_err(got, d["y"])
    print(f"{name}: f_phi norm-rel err vs the reference module {err:.2e}")
    assert got.shape == d["y"].shape and err <= TOL
    assert norm_rel_err(got, orc.fphi_forward(sd, torch.from_numpy(d["x"]), float(d["negative_slope"]))) <= TOL
    # ragged row counts (the kernel walks 8 rows per workgroup pass) and an empty batch
    for rows in (1, 7, 9):
        part = g.forward(torch.from_numpy(d["x"][:rows]).to(DEV)).cpu()
        assert torch.equal(part, got[:rows])
    assert g.forward(torch.zeros(0, g.dim_in, device=DEV)).shape == (0, g.dim_out)


def test_fphi_train_mode_is_rejected():
    d = load_raw("fphi_bn_wide")
    net, _ = ref_shaped_fphi(d)
    net.to(DEV).train()
    with pytest.raises(_lib.CardError):
        GuidanceNet.from_module(net)


def test_ols_guidance():
    torch.manual_seed(3)
    beta = torch.randn(14, 1)
    x = torch.randn(51, 13)
    got = GuidanceNet.from_ols(beta.to(DEV)).forward(x.to(DEV)).cpu()
    assert norm_rel_err(got, orc.ols_forward(beta, x)) <= 1e-6


@pytest.mark.parametrize("n_z", [1, 5, 100])
@pytest.mark.parametrize("coop", ["1", "0"])
def test_fused_call_equals_two_calls_and_oracle(n_z, coop, monkeypatch):
    """card_sampler_run with fphi set == f_phi forward + card_sampler_run on its output, bit for bit; and the whole
    thing within 1e-4 of the oracle fed with the REFERENCE module's y_0_hat (fixture)."""
    monkeypatch.setenv("CARD_AMD_COOP", coop)
    g = load_golden("reg_boston_T50")
    d = load_raw("fphi_boston")
    net, _ = ref_shaped_fphi(d)
    gn = GuidanceNet.from_module(net.to(DEV))
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    T = g["T"]
    coef = make_coef_table(g["alphas"], g["omabs"]).to(DEV)
    x = torch.from_numpy(d["x"]).to(DEV)
    rows = x.shape[0]
    N = rows * n_z
    noise = torch.randn(T, N, 1, generator=torch.Generator().manual_seed(5))
    y0 = gn.forward(x)
    two = s.run(x, y0, y0, coef, T, noise=noise.to(DEV), out_mode=_lib.OUT_ALL, n_z=n_z)
    buf = torch.full((rows, 1), float("nan"), device=DEV)
    one = s.run(x, None, None, coef, T, noise=noise.to(DEV), out_mode=_lib.OUT_ALL, n_z=n_z, fphi=gn, fphi_out=buf)
    assert torch.equal(buf, y0) and torch.equal(one, two)
    # RNG mode too, and the entry point that allocates nothing
    a = s.run(x, None, None, coef, T, seed=9, n_z=n_z, fphi=gn)
    b = s.run(x, y0, y0, coef, T, seed=9, n_z=n_z)
    c = s.run(x, None, None, coef, T, seed=9, n_z=n_z, fphi=gn, static_scratch=True)
    assert torch.equal(a, b) and torch.equal(a, c)
    # oracle on the reference module's own prediction, rows tiled like regression/card_regression.py:1027-1035
    y0_ref = torch.from_numpy(d["y"])
    tile = lambda t: t.repeat(n_z, 1, 1).transpose(0, 1).flatten(0, 1)
    want = torch.stack(orc.p_sample_loop(g["weights"], tile(torch.from_numpy(d["x"])), tile(y0_ref), tile(y0_ref), T,
                                         g["alphas"], g["omabs"], noise), 0)
    err = max(norm_rel_err(one[k].cpu(), want[k]) for k in range(T + 1))
    print(f"fused f_phi + sampler, n_z={n_z}, coop={coop}: worst per-state norm-rel err vs oracle {err:.2e}")
    assert err <= TOL
    # noise_prior (regression/card_regression.py:1032-1033): zeros as y_T_mean, f_phi(x) still the y_0_hat input
    z = torch.zeros(rows, 1, device=DEV)
    assert torch.equal(s.run(x, None, z, coef, T, seed=4, n_z=n_z, fphi=gn), s.run(x, y0, z, coef, T, seed=4, n_z=n_z))


def test_shim_p_sample_loop_from_x():
    """card_amd.p_sample_loop_from_x(model, f_phi module, x rows, ...) == the reference's sequence f_phi -> tile ->
    p_sample_loop, with both networks given as nn.Modules (cached, re-packed when their weights change)."""
    g = load_golden("reg_boston_T50")
    d = load_raw("fphi_boston")
    net, _ = ref_shaped_fphi(d)
    net = net.to(DEV)
    model = RefShapedGuided(g["sd"], True, True).to(DEV).eval()
    x = torch.from_numpy(d["x"]).to(DEV)
    T, n_z = g["T"], 7
    al, om = g["alphas"].to(DEV), g["omabs"].to(DEV)
    seq, y0hat = card_amd.p_sample_loop_from_x(model, net, x, T, al, om, n_z=n_z, seed=11)
    assert len(seq) == T + 1 and seq[0].shape == (x.shape[0] * n_z, 1)
    with torch.no_grad():
        assert norm_rel_err(y0hat.cpu(), net(x).cpu()) <= 1e-5
    ref = card_amd.p_sample_loop(model, x, y0hat, y0hat, T, al, om, n_z=n_z, seed=11)
    assert all(torch.equal(a, b) for a, b in zip(seq, ref))
    y_last, _ = card_amd.p_sample_loop_from_x(model, net, x, T, al, om, n_z=n_z, seed=11, only_last_sample=True)
    assert torch.equal(y_last, seq[-1])
    assert guidance_for(net) is guidance_for(net)
    # an in-place edit behind the version counters (param.data.copy_) re-packs f_phi
    first = guidance_for(net)
    with torch.no_grad():
        p = next(net.parameters())
        p.data.copy_(p.data * 1.5)
    assert guidance_for(net) is not first
    seq2, y0hat2 = card_amd.p_sample_loop_from_x(model, net, x, T, al, om, n_z=n_z, seed=11, only_last_sample=True)
    assert not torch.equal(y0hat2, y0hat)


def test_fused_regression_test_batch_helper():
    """card_amd.p_sample_loop_with_reg_metrics (f_phi + tiling + sampling + per-row metrics of the requested states, one
    call per test minibatch: regression/card_regression.py:1014-1074) == p_sample_loop_from_x followed by
    regression_row_metrics on the same states, bit for bit."""
    from card_amd.metrics import regression_row_metrics
    g = load_golden("reg_boston_T50")
    d = load_raw("fphi_boston")
    net, _ = ref_shaped_fphi(d)
    net = net.to(DEV)
    model = RefShapedGuided(g["sd"], True, True).to(DEV).eval()
    x = torch.from_numpy(d["x"]).to(DEV)
    rows, T, n_z = x.shape[0], g["T"], 50
    y_true = torch.randn(rows, generator=torch.Generator().manual_seed(2)).to(DEV)
    al, om = g["alphas"].to(DEV), g["omabs"].to(DEV)
    kw = dict(y_scale=9.2, y_shift=22.5, tau=None, trimmed_mean_range=(0.0, 100.0))
    m, y0hat = card_amd.p_sample_loop_with_reg_metrics(model, net, x, y_true, T, al, om, n_z=n_z, states=[0, T // 2, T], seed=3, **kw)
    assert m["se"].shape == (3, rows) and y0hat.shape == (rows, 1)
    ys, y0b = card_amd.p_sample_loop_from_x(model, net, x, T, al, om, n_z=n_z, seed=3, states=[0, T // 2, T])
    ref = regression_row_metrics(ys.reshape(3, rows, n_z), y_true, **kw)
    assert torch.equal(y0hat, y0b) and all(torch.equal(m[k], ref[k]) for k in ref)
    last, _ = card_amd.p_sample_loop_with_reg_metrics(model, net, x, y_true, T, al, om, n_z=n_z, seed=3, **kw)
    assert all(torch.equal(last[k][0], m[k][2]) for k in last) and torch.isfinite(last["nll"]).all()


def test_fphi_argument_errors():
    g = load_golden("reg_boston_T50")
    s = Sampler.from_state_dict(g["sd"], device=DEV)
    coef = make_coef_table(g["alphas"], g["omabs"]).to(DEV)
    year, _ = ref_shaped_fphi(load_raw("fphi_year"))
    gn = GuidanceNet.from_module(year.to(DEV))
    with pytest.raises(_lib.CardError):      # f_phi reads 90 features, the sampler's x has 13
        s.run(torch.zeros(4, 90, device=DEV), None, None, coef, g["T"], fphi=gn)
