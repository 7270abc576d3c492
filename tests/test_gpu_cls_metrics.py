"""GPU test of card_classification_row_metrics (card_metrics.hip) against fixtures produced by the reference's own
closures compute_val_before_softmax / compute_and_store_cls_metrics (classification/card_classification.py:529-603,
pulled out of Diffusion.test_image_task with `ast` by oracle/gen_golden.py) and scipy.stats.ttest_rel on seeded samples.

Bars: vote proportions, majority votes and instance accuracies are integer counts over n_z -> exact; probabilities and
credible-interval bounds within 2e-6 (float32 softmax / lerp, summation order of the mean); t-test p-values within 1e-4
relative (float32 probabilities that differ in their last bit move a tail p-value of 1e-20 by that much; measured <= 2.3e-5)
on the images whose two most voted classes are unambiguous (torch.argsort leaves the order
of equal counts unspecified; the kernel takes the lower class index), reject flags equal there."""
import glob
import os

import numpy as np
import pytest
import torch

from card_amd.metrics import classification_metric_sums, classification_row_metrics

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "clsmetrics_*.npz")))


def test_fixtures_present():
    assert len(NAMES) >= 4


@pytest.mark.parametrize("name", NAMES)
def test_matches_reference_closures(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    rows, n_z, C = int(d["rows"]), int(d["n_z"]), int(d["C"])
    gen = torch.from_numpy(d["gen"]).to(DEV)
    labels = torch.from_numpy(d["labels"]).to(DEV)
    kw = dict(scale_T=float(d["scale_T"]), ci_range=tuple(float(v) for v in d["ci_range"]), ttest_alpha=float(d["alpha"]))
    # the sampler's flat (rows * n_z, C) layout, as the reference's loop hands it over (:1020-1023)
    m = classification_row_metrics(gen.reshape(rows * n_z, C), labels, n_z=n_z, return_sample_probs=True, **kw)
    m = {k: v[0].cpu().numpy() for k, v in m.items()}
    assert np.array_equal(m["label_probs"], d["label_probs"])
    assert np.array_equal(m["instance_accuracy"], d["instance_accuracy"])
    uniq = d["unique_top2"]
    top_unique = np.sort(d["label_probs"], 1)[:, -1] > (np.sort(d["label_probs"], 1)[:, -2] if C > 1 else -1)
    assert np.array_equal(m["majority_vote"][top_unique], d["majority_vote"][top_unique])
    assert np.array_equal(m["majority_vote"], d["label_probs"].argmax(1))            # ties: lower index
    e_mean = np.abs(m["mean_probs"] - d["mean_probs"]).max()
    e_prob = np.abs(m["sample_probs"][:8] - d["probs_head"]).max()
    e_ci = np.abs(m["ci"] - d["ci"]).max() / max(1.0, np.abs(d["ci"]).max())
    pv, pr = m["ttest_pvalue"][uniq], d["ttest_pvalue"][uniq]
    e_p = np.max(np.abs(pv - pr) / np.maximum(np.abs(pr), 1e-300)) if uniq.any() else 0.0
    print(f"{name}: mean-prob err {e_mean:.2e}, sample-prob err {e_prob:.2e}, CI err {e_ci:.2e}, p-value rel err {e_p:.2e} "
          f"on {int(uniq.sum())}/{rows} images with an unambiguous top two")
    assert e_mean <= 2e-6 and e_prob <= 2e-6 and e_ci <= 2e-6
    assert e_p <= 1e-4
    # reject flags: equal wherever the pair is unambiguous and the p-value is not within rounding of alpha
    clear = uniq & (np.abs(d["ttest_pvalue"] - float(d["alpha"])) > 1e-9)
    assert np.array_equal(m["ttest_reject"][clear], d["ttest_reject"][clear])
    # several states at once + the additive sums
    two = torch.stack([gen, gen.flip(1)], 0)
    m2 = classification_row_metrics(two, labels, **kw)
    assert torch.equal(m2["label_probs"][0], m2["label_probs"][1]) and torch.equal(m2["majority_vote"][0].cpu(), torch.from_numpy(m["majority_vote"]))
    sums = classification_metric_sums(gen, labels, **kw)[0].cpu().numpy()
    assert sums[0] == rows and sums[1] == (m["majority_vote"] == d["labels"]).sum()
    assert abs(sums[2] - d["instance_accuracy"].astype(np.float64).sum()) < 1e-9


def test_degenerate_rows_and_limits():
    """One sample per image (no variance: p-value NaN, no rejection), one class, all votes for one class (runner-up by index),
    and the LDS limit is reported as an error."""
    from card_amd._lib import CardError
    y = torch.tensor([[[0.9, 0.1, 0.0]], [[0.0, 0.2, 1.1]]], device=DEV)             # (2 rows, n_z = 1, C = 3)
    m = classification_row_metrics(y, torch.tensor([0, 1], device=DEV))
    assert m["majority_vote"][0].tolist() == [0, 2] and m["instance_accuracy"][0].tolist() == [1.0, 0.0]
    assert torch.isnan(m["ttest_pvalue"]).all() and m["ttest_reject"].sum() == 0
    same = torch.zeros(1, 50, 4, device=DEV); same[..., 2] = 1.0                      # every sample predicts class 2, identical
    m = classification_row_metrics(same, torch.tensor([2], device=DEV))
    assert m["majority_vote"].item() == 2 and m["label_probs"][0, 0].tolist() == [0.0, 0.0, 1.0, 0.0]
    assert m["ttest_pvalue"].item() == 0.0 or torch.isnan(m["ttest_pvalue"]).item()   # constant non-zero difference: t = inf
    one = classification_row_metrics(torch.randn(3, 10, 1, device=DEV), torch.zeros(3, device=DEV))
    assert (one["majority_vote"] == 0).all() and torch.allclose(one["mean_probs"], torch.ones_like(one["mean_probs"]))
    with pytest.raises(CardError, match="LDS"):
        classification_row_metrics(torch.zeros(1, 1000, 100, device=DEV), torch.zeros(1, device=DEV))


def test_fused_test_batch_helper_equals_sampling_then_metrics():
    """p_sample_loop_with_cls_metrics (one classification test minibatch on the device: in-kernel tiling, sampling, metrics
    of the requested states) == p_sample_loop on the same seed followed by classification_row_metrics."""
    import card_amd
    from tests.helpers import load_golden
    from tests.test_gpu_cls import RefShapedCls
    g = load_golden("cls_mnist_T50")
    m = RefShapedCls(g["sd"]).to(DEV).eval()
    T, rows, n_s = g["T"], 9, 33
    xf, y0 = g["x_feat"][:rows].to(DEV), g["y0hat"][:rows].to(DEV)
    labels = y0.argmax(1)
    al, om = g["alphas"].to(DEV), g["omabs"].to(DEV)
    got = card_amd.p_sample_loop_with_cls_metrics(m, xf, y0, labels, T, al, om, n_samples=n_s, num_ts=[1, 20, T + 1], seed=5)
    assert sorted(got) == [1, 20, T + 1]
    full = card_amd.p_sample_loop(m, xf, y0, y0, T, al, om, seed=5, n_z=n_s)
    for num_t, mm in got.items():
        ref = classification_row_metrics(full[num_t - 1], labels, n_z=n_s)
        for k, v in mm.items():
            assert torch.equal(v, ref[k][0]) or (k == "ttest_pvalue" and torch.equal(torch.isnan(v), torch.isnan(ref[k][0]))), (num_t, k)
    default = card_amd.p_sample_loop_with_cls_metrics(m, xf, y0, labels, T, al, om, n_samples=n_s, seed=5)
    assert list(default) == [T + 1] and torch.equal(default[T + 1]["majority_vote"], got[T + 1]["majority_vote"])
