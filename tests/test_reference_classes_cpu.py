"""The drop-in shim against the reference's REAL classes (build container only: /root/reference does not travel).

`card_amd.describe(model)` is the device-free half of `sampler_for`: recognition, dimensions, flags and the ordered list of
tensors to pack.  Here it is fed the actual `ConditionalGuidedModel` (regression/model.py:21), `ConditionalModel`
(classification/model.py:22, archs linear / simple and the toy encoder), the ImageNet `CARD` wrapper and its
`ConditionalModel` head (classification/imagenet/card/builder.py:8, :69) -- bare, inside `DataParallel`, inside a real
`DistributedDataParallel` (gloo, one rank) as classification/imagenet/main.py:222-237 wraps it -- and the guidance network
`DeterministicFeedForwardNeuralNetwork` (regression/model.py:55), and must return the module's own Parameter objects.
The packed weights are then checked through the CPU oracle: the oracle driven by the tensors `describe` returned must
reproduce the real class's forward."""
import argparse
import os
import tempfile

import pytest
import torch

import card_amd
from card_amd import _lib
from card_amd._lib import CardError
from card_amd.guidance import describe_guidance
from card_amd.sampler import core_of, describe, unwrap

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
NS = argparse.Namespace


@pytest.fixture(scope="module")
def ref():
    from oracle import gen_golden as gg
    reg_du, reg_model, cls_du, cls_model = gg.load_reference()
    inet_du, inet_builder = gg.load_imagenet_reference()
    return NS(reg_du=reg_du, reg_model=reg_model, cls_du=cls_du, cls_model=cls_model, inet_du=inet_du, inet_builder=inet_builder)


def _guided(ref, x_dim, cat_x, cat_y_pred, T=20):
    cfg = NS(diffusion=NS(timesteps=T), model=NS(cat_x=cat_x, cat_y_pred=cat_y_pred, x_dim=x_dim, y_dim=1))
    return ref.reg_model.ConditionalGuidedModel(cfg)


@pytest.mark.parametrize("cat_x,cat_y_pred", [(True, True), (True, False), (False, True), (False, False)])
def test_real_conditional_guided_model_all_cat_variants(ref, cat_x, cat_y_pred):
    T, x_dim = 20, 7
    m = _guided(ref, x_dim, cat_x, cat_y_pred, T)
    d = describe(m)
    assert (d.kind, d.variant, d.core, d.wrapped, d.features) == ("guided", _lib.VARIANT_GUIDED, m, False, None)
    assert (d.y_dim, d.hidden, d.n_embed, d.cat_x, d.cat_y_pred) == (1, 128, T + 1, cat_x, cat_y_pred)
    assert d.x_dim == (x_dim if cat_x else 0) and d.d_in == 1 + (1 if cat_y_pred else 0) + (x_dim if cat_x else 0)
    assert list(d.tensors) == ["W1", "b1", "gamma1", "W2", "b2", "gamma2", "W3", "b3", "gamma3", "W4", "b4"]
    for i in (1, 2, 3):
        cl = getattr(m, f"lin{i}")
        assert d.tensors[f"W{i}"] is cl.lin.weight and d.tensors[f"b{i}"] is cl.lin.bias and d.tensors[f"gamma{i}"] is cl.embed.weight
    assert d.tensors["W4"] is m.lin4.weight and d.tensors["b4"] is m.lin4.bias
    # the described tensors, fed to the oracle, reproduce the real module's forward
    from oracle import card_oracle as orc
    w = orc.GuidedWeights(*[[d.tensors[f"{p}{i}"].detach() for i in (1, 2, 3)] for p in ("W", "b", "gamma")],
                          d.tensors["W4"].detach(), d.tensors["b4"].detach(), d.cat_x, d.cat_y_pred)
    x, y, y0 = torch.randn(5, x_dim), torch.randn(5, 1), torch.randn(5, 1)
    with torch.no_grad():
        want = m(x, y, y0, torch.tensor([3]))
    got = orc.eps_theta(w, x, y, y0, 3)
    assert torch.allclose(got, want, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("dataset,arch,data_dim,F,C", [("MNIST", "linear", 784, 128, 10), ("CIFAR10", "linear", 96, 256, 10),
                                                        ("FashionMNIST", "simple", 784, 128, 10), ("toy", "linear", 2, 128, 3)])
@pytest.mark.parametrize("guidance", [True, False])
def test_real_classification_conditional_model(ref, dataset, arch, data_dim, F, C, guidance):
    T = 15
    cfg = NS(diffusion=NS(timesteps=T), model=NS(data_dim=data_dim, arch=arch, feature_dim=F, hidden_dim=F),
             data=NS(num_classes=C, dataset=dataset))
    m = ref.cls_model.ConditionalModel(cfg, guidance=guidance).eval()
    d = describe(m)
    assert (d.kind, d.variant, d.core, d.features) == ("bn_hadamard", _lib.VARIANT_BN_HADAMARD, m, "norm(encoder_x(x))")
    assert (d.x_dim, d.y_dim, d.hidden, d.n_embed, d.cat_x, d.cat_y_pred) == (F, C, F, T + 1, False, guidance)
    assert d.d_in == (2 * C if guidance else C) and d.bn_eps == m.unetnorm1.eps == 1e-5
    for i in (1, 2, 3):
        bn = getattr(m, f"unetnorm{i}")
        assert d.tensors[f"bn{i}.weight"] is bn.weight and d.tensors[f"bn{i}.running_var"] is bn.running_var
        assert d.tensors[f"gamma{i}"] is getattr(m, f"lin{i}").embed.weight
    assert d.tensors["W4"] is m.lin4.weight
    # hoisted features from the REAL encoder (Sequential archs, or the toy dataset's single Linear) + oracle theta == forward
    from card_amd.sampler import x_features
    from oracle import card_oracle as orc
    with torch.no_grad():
        for bn in [m.norm] + [getattr(m, f"unetnorm{i}") for i in (1, 2, 3)]:
            bn.running_mean.normal_(0, 0.3)
            bn.running_var.uniform_(0.5, 1.5)
    x, y = torch.randn(6, data_dim), torch.randn(6, C)
    y0 = torch.softmax(torch.randn(6, C), 1)
    xf = x_features(m, x)
    assert xf.shape == (6, F) and x_features(m, x) is xf          # cached per x object
    w = orc.ClsWeights.from_state_dict(m.state_dict(), guidance=guidance)
    with torch.no_grad():
        want = m(x, y, torch.tensor([4]), y0)
    assert torch.allclose(orc.eps_theta(w, xf, y, y0, 4), want, atol=2e-5, rtol=1e-5)
    m.train()
    with pytest.raises(CardError, match="eval mode"):
        card_amd.sampler_for(m)


class _Phi(torch.nn.Module):
    """stand-in for the frozen ResNet-50 backbone: builder.py:21-29 only needs `.fc` (an nn.Linear) on it"""

    def __init__(self, num_classes=100, feat=64):
        super().__init__()
        self.body = torch.nn.Linear(12, feat)
        self.fc = torch.nn.Linear(feat, num_classes)

    def forward(self, x):
        return self.fc(self.body(x))


def _assert_inet(d, theta, C, F, T):
    assert (d.kind, d.variant, d.core, d.features) == ("linear_hadamard", _lib.VARIANT_LINEAR_HADAMARD, theta, "linx(z)")
    assert (d.x_dim, d.y_dim, d.hidden, d.n_embed, d.cat_x, d.cat_y_pred) == (F, C, F, T, False, True)
    assert list(d.tensors) == ["W1", "b1", "gamma1", "W2", "b2", "gamma2", "W4", "b4"]
    assert d.tensors["W1"] is theta.lin1.lin.weight and d.tensors["gamma2"] is theta.lin2.embed.weight
    assert d.tensors["W4"] is theta.lin3.weight and d.tensors["b4"] is theta.lin3.bias


def test_real_imagenet_card_wrapper_bare_dataparallel_and_ddp(ref):
    """imagenet/main.py:222-237 wraps CARD in DistributedDataParallel / DataParallel and passes THAT to p_sample_loop
    (:482, :557); the tree itself reaches through with model.module.model_theta (:357)."""
    C, F, T = 20, 64, 10
    card = ref.inet_builder.CARD(lambda num_classes: _Phi(num_classes, F), num_classes=C, n_steps=T)
    theta = card.model_theta
    assert isinstance(theta, ref.inet_builder.ConditionalModel)
    _assert_inet(describe(theta), theta, C, F, T)                     # the bare head
    d = describe(card)                                                # the CARD wrapper
    _assert_inet(d, theta, C, F, T)
    assert d.wrapped and core_of(card) is theta
    dp = torch.nn.DataParallel(card)                                  # main.py:237
    d = describe(dp)
    _assert_inet(d, theta, C, F, T)
    assert d.wrapped and unwrap(dp) is card and core_of(dp) is theta
    import torch.distributed as dist
    made = False
    if not dist.is_initialized():
        store = dist.FileStore(tempfile.mktemp(prefix="card_ddp_store_"), 1)
        dist.init_process_group("gloo", store=store, rank=0, world_size=1)
        made = True
    try:
        ddp = torch.nn.parallel.DistributedDataParallel(card, find_unused_parameters=True)     # main.py:222 / :227
        d = describe(ddp)
        _assert_inet(d, theta, C, F, T)
        assert d.wrapped and unwrap(ddp) is card and core_of(ddp) is theta
        # the hoisted features come from the inner head too (the wrapper's forward is never called by the shim)
        from card_amd.sampler import x_features
        z = torch.randn(4, F)
        with torch.no_grad():
            assert torch.equal(x_features(ddp, z), theta.linx(z))
        # the frozen backbone is not part of the sampling path: neither packed nor hashed
        from card_amd.sampler import _tensors
        ids = {id(t) for t in _tensors(core_of(ddp))}
        assert id(card.model_phi.body.weight) not in ids and id(theta.linx.weight) in ids
    finally:
        if made:
            dist.destroy_process_group()
    # oracle on the described tensors == the real wrapper's forward in mode 'theta'
    from oracle import card_oracle as orc
    w = orc.LinearHeadWeights.from_state_dict(theta.state_dict())
    z, y = torch.randn(5, F), torch.randn(5, C)
    y0 = torch.softmax(torch.randn(5, C), 1)
    with torch.no_grad():
        want = card(z, y, torch.tensor([2]), y0, mode="theta")
        xf = theta.linx(z)
    assert torch.allclose(orc.eps_theta(w, xf, y, y0, 2), want, atol=2e-5, rtol=1e-5)


def test_module_holder_around_regression_and_classification_models(ref):
    m = _guided(ref, 5, True, True)
    for holder in (torch.nn.DataParallel(m), torch.nn.DataParallel(torch.nn.DataParallel(m))):
        d = describe(holder)
        assert d.core is m and d.wrapped and d.tensors["W1"] is m.lin1.lin.weight

    class Holder(torch.nn.Module):            # e.g. apex's DistributedDataParallel: only a `.module`
        def __init__(self, inner):
            super().__init__()
            self.module = inner

    assert describe(Holder(m)).core is m
    with pytest.raises(CardError, match="does not recognise Linear"):
        describe(torch.nn.DataParallel(torch.nn.Linear(3, 3)))
    with pytest.raises(CardError, match="does not recognise"):
        describe(torch.nn.Sequential(torch.nn.Linear(3, 3)))


def test_describe_rejects_inconsistent_shapes(ref):
    m = _guided(ref, 5, True, True)
    m.lin2.lin = torch.nn.Linear(64, 128)
    with pytest.raises(CardError, match=r"lin2.lin.weight: shape \(128, 64\)"):
        describe(m)
    m = _guided(ref, 5, True, True)
    m.lin3.embed = torch.nn.Embedding(7, 128)
    with pytest.raises(CardError, match="lin3.embed.weight"):
        describe(m)


@pytest.mark.parametrize("bn", [False, True])
def test_real_guidance_network(ref, bn):
    net = ref.reg_model.DeterministicFeedForwardNeuralNetwork(dim_in=13, dim_out=1, hid_layers=[100, 50], use_batchnorm=bn,
                                                              negative_slope=0.01, dropout_rate=0.2)
    with pytest.raises(CardError, match="eval mode"):
        describe_guidance(net)                       # train mode: Dropout (and BatchNorm) are batch-dependent
    net.eval()
    d = describe_guidance(net)
    lins = [m for m in net.network if isinstance(m, torch.nn.Linear)]
    bns = [m for m in net.network if isinstance(m, torch.nn.BatchNorm1d)]
    assert d["dims"] == [13, 100, 50, 1] and d["negative_slope"] == 0.01
    assert all(a is b.weight for a, b in zip(d["weights"], lins)) and all(a is b.bias for a, b in zip(d["biases"], lins))
    assert d["batchnorm"] == ((bns + [None]) if bn else [None, None, None])
    # device-free restatement of what the kernel computes from the described tensors == the module's forward
    from card_amd.guidance import _bn_fold
    x = torch.randn(9, 13)
    with torch.no_grad():
        for m in bns:
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
        h = x
        for l, (W, b) in enumerate(zip(d["weights"], d["biases"])):
            h = h @ W.t() + b
            if d["batchnorm"][l] is not None:
                sc, sh = _bn_fold(d["batchnorm"][l])
                h = h * sc + sh
            if l < len(d["weights"]) - 1:
                h = torch.nn.functional.leaky_relu(h, d["negative_slope"])
        assert torch.allclose(h, net(x), atol=1e-5, rtol=1e-5)


def test_shim_entry_points_reach_the_device_check_with_real_wrapped_models(ref):
    """No GPU here: every entry point must get THROUGH recognition of the real (wrapped) classes and stop at the device
    check ('runs on an AMD GPU only'), not at 'does not recognise DistributedDataParallel' (round-2 defect)."""
    C, F, T = 20, 64, 10
    card = ref.inet_builder.CARD(lambda num_classes: _Phi(num_classes, F), num_classes=C, n_steps=T)
    dp = torch.nn.DataParallel(card)
    from card_amd.synth import make_schedule
    al, _, om = make_schedule("linear", T, 0.01, 0.95)
    z, y0 = torch.randn(3, F), torch.softmax(torch.randn(3, C), 1)
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(CardError, match="AMD GPU only"):
        card_amd.p_sample_loop(dp, z, y0, y0, T, al, om, only_last_sample=True)
    with pytest.raises(CardError, match="AMD GPU only"):
        card_amd.p_sample_loop_repeated(dp, z, y0, y0, T, al, om, 7)
    with pytest.raises(CardError, match="AMD GPU only"):
        card_amd.p_sample(torch.nn.DataParallel(_guided(ref, 5, True, True, T - 1)), torch.randn(3, 5), torch.randn(3, 1),
                          torch.randn(3, 1), torch.randn(3, 1), 3, al, om)
