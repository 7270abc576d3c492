"""On-device regression test metrics (RMSE / QICE / PICP / NLL pieces) for generated samples.

Host-side mirror of the metric helpers inside `Diffusion.test` (regression/card_regression.py:763-893): the
per-row statistics come from `card_regression_row_metrics` (one workgroup sorts a row's n_z samples in LDS);
the reductions over rows are tiny torch ops and, across GPUs, one all-reduce (card_amd.distributed)."""
import ctypes as C

import torch

from . import _lib
from ._lib import CardError
from .sampler import _dev_f32, _ptr, _stream, _require_cuda


def regression_row_metrics(samples, y_true, *, n_bins=10, picp_range=(2.5, 97.5), trimmed_mean_range=(0.0, 100.0),
                           y_scale=1.0, y_shift=0.0, tau=None):
    """samples: (rows, n_z) / (rows, n_z, 1) or (n_states, rows, n_z[, 1]) generated y (normalised, chain order
    (row, z)); y_true: (rows,) / (rows, 1) normalised targets.  Returns dict of (n_states, rows) tensors:
    se, nll (float32), qice_bin, picp_in (int32)."""
    dev = samples.device
    _require_cuda(dev)
    s = _dev_f32(samples, dev)
    if s.dim() >= 3 and s.shape[-1] == 1:
        s = s.squeeze(-1)
    if s.dim() == 2:
        s = s.unsqueeze(0)
    if s.dim() != 3:
        raise CardError(f"samples: expected (rows, n_z) or (n_states, rows, n_z), got {tuple(samples.shape)}")
    n_states, rows, n_z = s.shape
    yt = _dev_f32(y_true, dev).reshape(-1)
    if yt.numel() != rows:
        raise CardError(f"y_true has {yt.numel()} entries, samples have {rows} rows")
    out = dict(se=torch.empty(n_states, rows, device=dev), nll=torch.empty(n_states, rows, device=dev),
               qice_bin=torch.empty(n_states, rows, device=dev, dtype=torch.int32),
               picp_in=torch.empty(n_states, rows, device=dev, dtype=torch.int32))
    a = _lib.MetricArgs()
    a.struct_bytes = C.sizeof(_lib.MetricArgs)
    a.n_z, a.rows, a.n_states, a.n_bins = n_z, rows, n_states, int(n_bins)
    a.samples, a.y_true = _ptr(s), _ptr(yt)
    a.y_scale, a.y_shift = float(y_scale), float(y_shift)
    a.trim_lo, a.trim_hi = float(trimmed_mean_range[0]), float(trimmed_mean_range[1])
    a.picp_lo, a.picp_hi = float(picp_range[0]), float(picp_range[1])
    a.tau = float(tau) if tau is not None else -1.0
    a.se, a.nll, a.qice_bin, a.picp_in = (_ptr(out[k]) for k in ("se", "nll", "qice_bin", "picp_in"))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().card_regression_row_metrics(C.byref(a), _stream(dev)), "card_regression_row_metrics")
    return out


def regression_metric_sums(samples, y_true, *, n_bins=10, **kw):
    """Additive sums per state, layout [n_rows, sum SE, PICP count, sum NLL, QICE counts (n_bins + 2)] as float64
    (n_states, 4 + n_bins + 2): all-reduce these over ranks, then `finalize`."""
    m = regression_row_metrics(samples, y_true, n_bins=n_bins, **kw)
    n_states, rows = m["se"].shape
    out = torch.zeros(n_states, 4 + n_bins + 2, dtype=torch.float64, device=m["se"].device)
    out[:, 0] = rows
    out[:, 1] = m["se"].double().sum(1)
    out[:, 2] = m["picp_in"].double().sum(1)
    out[:, 3] = m["nll"].double().sum(1)
    onehot = torch.nn.functional.one_hot(m["qice_bin"].long(), n_bins + 2)
    out[:, 4:] = onehot.sum(1).double()
    return out


def finalize(sums, n_bins=10):
    """RMSE, PICP, NLL, QICE per state from (all-reduced) sums (regression/card_regression.py:803-810, :1109-1124)."""
    n = sums[:, 0]
    counts = sums[:, 4:].clone()
    counts[:, 1] += counts[:, 0]
    counts[:, -2] += counts[:, -1]
    ratio = counts[:, 1:-1] / n[:, None]
    return {"rmse": (sums[:, 1] / n).sqrt(), "picp": sums[:, 2] / n, "nll": sums[:, 3] / n,
            "qice": (1.0 / n_bins - ratio).abs().mean(1)}


def classification_row_metrics(samples, labels, *, n_z=None, scale_T=1.0, ci_range=(2.5, 97.5), ttest_alpha=0.05,
                               return_sample_probs=False):
    """Per-image statistics of generated classification samples: `card_classification_row_metrics`, the device side of
    compute_and_store_cls_metrics (classification/card_classification.py:540-603).
    samples: (rows, n_z, C), (n_states, rows, n_z, C), or the sampler's flat (rows * n_z, C) with n_z given (chain order
    (image, z)); labels: (rows,) true classes.  Returns a dict of tensors with a leading n_states axis: label_probs,
    mean_probs (rows, C), ci (rows, 2, C), majority_vote, ttest_reject (int32), instance_accuracy (float32),
    ttest_pvalue (float64) and, on request, sample_probs (rows, n_z, C).  Vote ties go to the lower class index."""
    dev = samples.device
    _require_cuda(dev)
    s = _dev_f32(samples, dev)
    if s.dim() == 2:
        if not n_z or s.shape[0] % int(n_z):
            raise CardError(f"samples of shape {tuple(s.shape)} need n_z (a divisor of {s.shape[0]})")
        s = s.reshape(s.shape[0] // int(n_z), int(n_z), s.shape[1])
    if s.dim() == 3:
        s = s.unsqueeze(0)
    if s.dim() != 4:
        raise CardError(f"samples: expected (rows, n_z, C) or (n_states, rows, n_z, C), got {tuple(samples.shape)}")
    n_states, rows, nz, ncls = s.shape
    lab = labels.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
    if lab.numel() != rows:
        raise CardError(f"labels has {lab.numel()} entries, samples have {rows} rows")
    f32 = dict(device=dev, dtype=torch.float32)
    out = dict(label_probs=torch.empty(n_states, rows, ncls, **f32), mean_probs=torch.empty(n_states, rows, ncls, **f32),
               ci=torch.empty(n_states, rows, 2, ncls, **f32),
               majority_vote=torch.empty(n_states, rows, device=dev, dtype=torch.int32),
               instance_accuracy=torch.empty(n_states, rows, **f32),
               ttest_pvalue=torch.empty(n_states, rows, device=dev, dtype=torch.float64),
               ttest_reject=torch.empty(n_states, rows, device=dev, dtype=torch.int32))
    if return_sample_probs:
        out["sample_probs"] = torch.empty(n_states, rows, nz, ncls, **f32)
    a = _lib.ClsMetricArgs()
    a.struct_bytes = C.sizeof(_lib.ClsMetricArgs)
    a.n_z, a.n_classes, a.n_states, a.rows = nz, ncls, n_states, rows
    a.samples, a.labels = _ptr(s), lab.data_ptr()
    a.scale_T, a.ci_lo, a.ci_hi, a.ttest_alpha = float(scale_T), float(ci_range[0]), float(ci_range[1]), float(ttest_alpha)
    for k in ("label_probs", "mean_probs", "ci", "majority_vote", "instance_accuracy", "ttest_pvalue", "ttest_reject"):
        setattr(a, k, out[k].data_ptr())
    a.sample_probs = out["sample_probs"].data_ptr() if return_sample_probs else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().card_classification_row_metrics(C.byref(a), _stream(dev)), "card_classification_row_metrics")
    return out


def classification_metric_sums(samples, labels, **kw):
    """Additive sums per state as float64 (n_states, 4): [n images, majority votes that are correct, sum of instance
    accuracies, t-test rejections] -- all-reduce over ranks, divide by column 0 (card_classification.py:1033-1060)."""
    m = classification_row_metrics(samples, labels, **kw)
    lab = labels.to(device=m["majority_vote"].device, dtype=torch.int32).reshape(1, -1)
    n_states, rows = m["majority_vote"].shape
    out = torch.zeros(n_states, 4, dtype=torch.float64, device=lab.device)
    out[:, 0] = rows
    out[:, 1] = (m["majority_vote"] == lab).double().sum(1)
    out[:, 2] = m["instance_accuracy"].double().sum(1)
    out[:, 3] = m["ttest_reject"].double().sum(1)
    return out


def classification_test_sums(m, labels, *, n_bins=10, state=0):
    """Additive test-set sums from per-image statistics (`classification_row_metrics` output or the same arrays from
    anywhere; torch tensors on any device).  Everything Diffusion.test_image_task reports after its loop is a function
    of these sums, so shards / batches / GPUs just add them (all-reduce) and call `finalize_classification`:
    compute_quantile_metrics (classification/card_classification.py:673-741), compute_ttest_metrics (:743-822), PAvPU
    (:1076-1081), NLL (:1092-1094) and ECE (:1096-1125).
    Returns float64 tensors: per_class (C, 6) = [n, correct, rejected, correct & rejected, sum CI width of the true class
    over correct predictions, ... over incorrect ones]; ece_bins (n_bins + 1, 3) = [count, correct, sum confidence] by the
    reference's bin membership; hist (n_bins,) = torch.histogram counts of the confidences; nll_sum (1,)."""
    pick = lambda k: m[k][state] if m[k].dim() > (3 if k == "ci" else (2 if k in ("mean_probs", "label_probs") else 1)) else m[k]
    mv, rej = pick("majority_vote").long().reshape(-1), pick("ttest_reject").bool().reshape(-1)
    ci, mp = pick("ci").float(), pick("mean_probs").float()
    dev = mp.device
    lab = labels.to(dev).long().reshape(-1)
    C = mp.shape[-1]
    cor = mv == lab
    ciw_true = (ci[:, 1] - ci[:, 0]).gather(1, lab[:, None])[:, 0].double()
    cols = torch.stack([torch.ones_like(ciw_true), cor.double(), rej.double(), (cor & rej).double(),
                        ciw_true * cor.double(), ciw_true * (~cor).double()], 1)
    per_class = torch.zeros(C, 6, dtype=torch.float64, device=dev).index_add_(0, lab, cols)
    conf = mp.gather(1, mv[:, None])[:, 0]                                    # confidence: mean probability of the voted class
    nll_sum = -torch.log(mp.gather(1, lab[:, None])[:, 0]).double().sum().reshape(1)
    edges = torch.linspace(0, 1, n_bins + 1, device=dev)[1:]
    memb = ((conf[:, None] - edges) >= 0).sum(1)                              # 0 .. n_bins (:1104)
    ece_cols = torch.stack([torch.ones_like(conf, dtype=torch.float64), cor.double(), conf.double()], 1)
    ece_bins = torch.zeros(n_bins + 1, 3, dtype=torch.float64, device=dev).index_add_(0, memb, ece_cols)
    hb = torch.clamp((conf.double() * n_bins).floor().long(), 0, n_bins - 1)  # torch.histogram(bins, range=(0, 1)): last bin closed
    hist = torch.zeros(n_bins, dtype=torch.float64, device=dev).index_add_(0, hb, torch.ones_like(conf, dtype=torch.float64))
    return {"per_class": per_class, "ece_bins": ece_bins, "hist": hist, "nll_sum": nll_sum}


def finalize_classification(sums):
    """The numbers test_image_task logs, from (summed) `classification_test_sums`.  Means over empty groups are NaN, as the
    reference's `.mean()` of an empty selection."""
    pc, eb, hist = sums["per_class"].double(), sums["ece_bins"].double(), sums["hist"].double()
    n, cor, rej, cr, w_cor, w_inc = (pc[:, i] for i in range(6))
    N, n_cor, n_rej, n_cr = n.sum(), cor.sum(), rej.sum(), cr.sum()
    n_bins = hist.numel()
    acc_b, conf_b = eb[:n_bins, 1] / eb[:n_bins, 0], eb[:n_bins, 2] / eb[:n_bins, 0]
    cal = torch.where(eb[:n_bins, 0] > 0, (acc_b - conf_b).abs(), torch.zeros_like(acc_b))
    return {
        "accuracy": n_cor / N,
        "accuracy_by_class": cor / n,                                          # maj_vote_acc_by_class
        "ci_width_correct": w_cor.sum() / n_cor, "ci_width_incorrect": w_inc.sum() / (N - n_cor),
        "ci_width_correct_by_class": w_cor / cor, "ci_width_incorrect_by_class": w_inc / (n - cor),
        "reject_rate_correct": n_cr / n_cor, "reject_rate_incorrect": (n_rej - n_cr) / (N - n_cor),
        "reject_rate_correct_by_class": cr / cor, "reject_rate_incorrect_by_class": (rej - cr) / (n - cor),
        "reject_rate_by_class": rej / n,
        "accuracy_rejected": n_cr / n_rej, "accuracy_not_rejected": (n_cor - n_cr) / (N - n_rej),
        "accuracy_rejected_by_class": cr / rej, "accuracy_not_rejected_by_class": (cor - cr) / (n - rej),
        "pavpu": (n_cr + (N - n_cor - (n_rej - n_cr))) / N,                    # (n_ac + n_iu) / n
        "nll": sums["nll_sum"].double().sum() / N,
        "ece": ((hist / N) * cal).sum(),
    }
