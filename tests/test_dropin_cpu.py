"""CPU-side checks of the drop-in module: it offers every name the reference's runners take from their star import of
`diffusion_utils` (regression/card_regression.py:16, classification/card_classification.py:19), the host-side helpers
behave like the reference's, and the caches cannot hand out another tensor's result."""
import ast
import importlib
import inspect
import os
import sys
import warnings

import numpy as np
import pytest
import torch

import card_amd
from card_amd import diffusion_utils as du
from card_amd._lib import CardError
from card_amd.sampler import coef_on_device, make_coef_table
from card_amd.synth import make_schedule

REF = "/root/reference"
# what the two runners use out of `from diffusion_utils import *` (read off the reference; re-derived below when present)
USED_BY_RUNNERS = {"make_beta_schedule", "q_sample", "p_sample", "p_sample_t_1to0", "p_sample_loop", "kld", "y_0_reparam"}
REFERENCE_PUBLIC = USED_BY_RUNNERS | {"extract"}


def test_shim_exports_every_name_the_runners_use():
    assert REFERENCE_PUBLIC <= set(du.__all__)
    for n in du.__all__:
        assert callable(getattr(du, n)), n
    # the documented drop-in: the module under the reference's own name, star-imported
    sys.modules["diffusion_utils"] = du
    try:
        ns = {}
        exec("from diffusion_utils import *", ns)
        assert REFERENCE_PUBLIC <= set(ns)
    finally:
        del sys.modules["diffusion_utils"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_names_rederived_from_the_reference_sources():
    """Every function the reference's diffusion_utils modules define and its runners mention is exported here."""
    warnings.simplefilter("ignore")          # the reference sources hold invalid escape sequences
    for tree, runner in (("regression", "card_regression.py"), ("classification", "card_classification.py")):
        defined = {n.name for n in ast.parse(open(f"{REF}/{tree}/diffusion_utils.py").read()).body
                   if isinstance(n, ast.FunctionDef)}
        used = {n.id for n in ast.walk(ast.parse(open(f"{REF}/{tree}/{runner}").read())) if isinstance(n, ast.Name)}
        need = defined & used
        assert need and need <= set(du.__all__), (tree, need - set(du.__all__))
        assert need <= USED_BY_RUNNERS | {"extract"}


def test_signatures_follow_the_reference_argument_order():
    sig = lambda f: [p.name for p in inspect.signature(f).parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD]
    assert sig(du.p_sample_loop) == ["model", "x", "y_0_hat", "y_T_mean", "n_steps", "alphas", "one_minus_alphas_bar_sqrt",
                                     "only_last_sample"]                      # reg :107 / cls :119-120
    assert sig(du.p_sample) == ["model", "x", "y", "y_0_hat", "y_T_mean", "t", "alphas", "one_minus_alphas_bar_sqrt"]
    assert sig(du.p_sample_t_1to0) == ["model", "x", "y", "y_0_hat", "y_T_mean", "one_minus_alphas_bar_sqrt"]
    assert sig(du.q_sample) == ["y", "y_0_hat", "alphas_bar_sqrt", "one_minus_alphas_bar_sqrt", "t", "noise"]
    assert sig(du.y_0_reparam) == ["model", "x", "y", "y_0_hat", "y_T_mean", "t", "one_minus_alphas_bar_sqrt"]
    assert sig(du.kld) == ["y1", "y2", "grid", "num_grid"]
    # BASELINE.json north_star's 8-argument form, as an adapter
    assert sig(du.p_sample_loop_with_y) == ["model", "x", "y", "y_0_hat", "y_T_mean", "T", "alphas",
                                            "one_minus_alphas_bar_sqrt"]


def test_eight_argument_adapter_validates_y_then_takes_the_reference_path():
    m = torch.nn.Linear(2, 2)          # not a CARD network: the reference path must be reached and refuse it
    x, y0 = torch.zeros(4, 3), torch.zeros(4, 1)
    a, _, s = make_schedule("linear", 10, 1e-4, 0.02)
    with pytest.raises(CardError, match="does not match"):
        du.p_sample_loop_with_y(m, x, torch.zeros(5, 1), y0, y0, 10, a, s)
    with pytest.raises(CardError, match="cuda|recognise"):
        du.p_sample_loop_with_y(m, x, torch.zeros(4, 1), y0, y0, 10, a, s)
    with pytest.raises(CardError, match="cuda|recognise"):
        du.p_sample_loop_with_y(m, x, None, y0, y0, 10, a, s)


def test_kld_matches_the_reference_formula():
    g = torch.Generator().manual_seed(3)
    y1, y2 = torch.randn(4000, 1, generator=g) * 2, torch.randn(4000, 1, generator=g) * 2.5 + 0.3
    got = du.kld(y1, y2)
    p1, _ = np.histogram(y1.numpy().flatten(), bins=400, range=[-20, 20], density=True)
    p2, _ = np.histogram(y2.numpy().flatten(), bins=400, range=[-20, 20], density=True)
    p1 += 1e-7
    p2 += 1e-7
    assert got == (p1 * np.log(p1 / p2)).sum() and got > 0
    assert du.kld(y1, y1) == 0.0
    if os.path.isdir(REF):
        spec = importlib.util.spec_from_file_location("ref_reg_du_kld", f"{REF}/regression/diffusion_utils.py")
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        assert ref.kld(y1, y2) == got and ref.kld(y2, y1, grid=(-10, 10), num_grid=100) == du.kld(y2, y1, (-10, 10), 100)


def test_coefficient_cache_never_serves_another_schedule():
    """Two schedules of equal T passed as short-lived temporaries: a cache keyed on addresses alone could hand the second
    one the first one's table once the allocator recycles the address.  Entries hold their tensors."""
    T = 64
    for rep in range(3):
        for end in (0.02, 0.4, 0.1):
            alphas, _, omabs = make_schedule("linear", T, 1e-4, end)
            got = coef_on_device(alphas.clone(), omabs.clone(), T, torch.device("cpu"))
            assert torch.equal(got, make_coef_table(alphas, omabs)), (rep, end)
    a, _, s = make_schedule("linear", T, 1e-4, 0.02)
    first = coef_on_device(a, s, T, torch.device("cpu"))
    assert coef_on_device(a, s, T, torch.device("cpu")) is first          # same objects: a hit
    s.mul_(0.999)                                                          # in-place edit bumps the version: a miss
    assert not torch.equal(coef_on_device(a, s, T, torch.device("cpu")), first)


def test_package_reexports():
    for n in ("p_sample_loop", "p_sample", "p_sample_t_1to0", "q_sample", "kld", "p_sample_loop_with_y", "invalidate",
              "sampler_for", "Sampler"):
        assert hasattr(card_amd, n), n
