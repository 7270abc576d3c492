"""Test-set level classification metrics (card_amd.metrics.classification_test_sums / finalize_classification) against the
reference's own code: compute_quantile_metrics, compute_ttest_metrics and the PAvPU / NLL / ECE statements of
Diffusion.test_image_task (classification/card_classification.py:673-822, :1075-1123), run by oracle/gen_golden.py on the
per-image outputs of the reference's compute_and_store_cls_metrics and stored in tests/golden/clsmetrics_*.npz (ds_*).
Pure host logic on torch tensors: runs without a GPU.  Also: the sums are additive over shards."""
import glob
import os

import numpy as np
import pytest
import torch

from card_amd.metrics import classification_test_sums, finalize_classification

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "clsmetrics_*.npz")))
PAIRS = [("accuracy_by_class", "ds_acc_by_class"), ("ci_width_correct_by_class", "ds_ciw_cor_by_class"),
         ("ci_width_incorrect_by_class", "ds_ciw_inc_by_class"), ("reject_rate_correct_by_class", "ds_rej_cor_by_class"),
         ("reject_rate_incorrect_by_class", "ds_rej_inc_by_class"), ("reject_rate_by_class", "ds_rej_by_class"),
         ("accuracy_rejected_by_class", "ds_acc_rej_by_class"), ("accuracy_not_rejected_by_class", "ds_acc_nrej_by_class"),
         ("pavpu", "ds_pavpu"), ("nll", "ds_nll"), ("ece", "ds_ece")]


def per_image(d, sl=slice(None)):
    return {k: torch.from_numpy(d[k][sl]) for k in ("majority_vote", "ttest_reject", "ci", "mean_probs")}


@pytest.mark.parametrize("name", NAMES)
def test_matches_reference_test_set_metrics(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    labels = torch.from_numpy(d["labels"])
    f = finalize_classification(classification_test_sums(per_image(d), labels))
    for mine, ref in PAIRS:
        got, want = f[mine].numpy(), d[ref]
        assert np.array_equal(np.isnan(got), np.isnan(want)), (mine, got, want)      # empty groups: NaN on both sides
        assert np.allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=2e-6, atol=2e-7), (mine, got, want)
    assert abs(float(f["accuracy"]) - np.mean(d["majority_vote"] == d["labels"])) < 1e-12


def test_sums_are_additive_over_shards():
    d = np.load(os.path.join(GOLD, "clsmetrics_c10_nz100.npz"))
    labels = torch.from_numpy(d["labels"])
    whole = classification_test_sums(per_image(d), labels)
    cut = 23
    a = classification_test_sums(per_image(d, slice(0, cut)), labels[:cut])
    b = classification_test_sums(per_image(d, slice(cut, None)), labels[cut:])
    for k in whole:
        assert torch.allclose(a[k] + b[k], whole[k], rtol=0, atol=1e-9), k
