"""Shared helpers for the test-suite: fixture loading and the parity metric."""
import glob
import os

import numpy as np
import torch

from oracle import card_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(kind=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    names = [n for n in names if n.split("_")[0] in ("reg", "cls", "clsw", "inet")]     # sampler fixtures
    if kind:
        names = [n for n in names if n.split("_")[0] == kind]
    return names


def load_raw(name):
    """A fixture as a plain dict of numpy arrays (metrics_* and fphi_* fixtures have no sampler fields)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def load_golden(name):
    d = load_raw(name)
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    if str(d["kind"]) == "classification_synth":
        # wide heads: the fixture holds the seed, inputs and the reference's outputs; the theta weights are
        # card_amd.synth.cls_state_dict(F, C, T, seed), which the reference model carried when the fixture was made
        from card_amd.synth import cls_state_dict
        over = sd
        sd = cls_state_dict(int(d["F"]), int(d["C"]), int(d["T"]), int(d["seed"]))
        sd.update(over)                   # trained small tensors stored next to the seed (clsw_cifar10_T1000)
    t = lambda k: torch.from_numpy(d[k])
    out = dict(raw=d, sd=sd, T=int(d["T"]), kind=str(d["kind"]),
               x=t("x") if "x" in d else (t("z") if "z" in d else t("x_feat")), y0hat=t("y0hat"), yTmean=t("yTmean"), alphas=t("alphas"), omabs=t("omabs"),
               alphas_bar_sqrt=t("alphas_bar_sqrt"), noise=t("noise"), traj=t("traj"),
               traj_idx=[int(i) for i in d["traj_idx"]], probe_t=[int(i) for i in d["probe_t"]])
    if out["kind"] == "regression":
        out["cat_x"], out["cat_y_pred"] = bool(d["cat_x"]), bool(d["cat_y_pred"])
        out["weights"] = orc.GuidedWeights.from_state_dict(sd, out["cat_x"], out["cat_y_pred"])
        out["x_in"] = out["x"]
    elif out["kind"] == "classification_synth":
        out["weights"] = orc.ClsWeights.from_state_dict(sd, guidance=True)
        out["x_feat"] = t("x_feat")
        out["x_in"] = out["x_feat"]
    elif out["kind"] == "imagenet_head":
        out["weights"] = orc.LinearHeadWeights.from_state_dict(sd, guidance=True)
        out["x_feat"] = t("x_feat")
        out["x_in"] = out["x_feat"]
    else:
        out["weights"] = orc.ClsWeights.from_state_dict(sd, guidance=True)
        out["x_feat"] = t("x_feat")
        out["x_in"] = out["x_feat"]
    return out


def norm_rel_err(got, ref):
    """Parity metric of SURVEY.md section 8c: max|got - ref| / max(1, max|ref|)."""
    got = torch.as_tensor(got, dtype=torch.float64)
    ref = torch.as_tensor(ref, dtype=torch.float64)
    return float((got - ref).abs().max() / max(1.0, float(ref.abs().max())))


def elem_errs(got, ref):
    """(norm-relative, max abs, max element-wise relative) error: the parity bar is the first;
    the other two are printed next to it so that a large |ref| cannot hide an absolute error.  The element-wise figure
    uses |ref| floored at 1e-3 of max|ref| (a state crossing zero has no meaningful relative error at the crossing)."""
    got = torch.as_tensor(got, dtype=torch.float64)
    ref = torch.as_tensor(ref, dtype=torch.float64)
    d = (got - ref).abs()
    floor = max(1e-30, 1e-3 * float(ref.abs().max()))
    return norm_rel_err(got, ref), float(d.max()), float((d / ref.abs().clamp_min(floor)).max())


PARITY_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_table.jsonl")


def record_parity(fixture, kernel, errs, mode="free-running"):
    """Append one measured parity line {fixture, kernel / form, mode, norm_rel, max_abs, max_elem_rel} to
    gpurun_out/parity_table.jsonl (scratch; merged back from the GPU box).  tools/collect_parity.py folds the file into
    profiles/r03_parity_table.json -- `pytest -q` drops what the tests print, this keeps it."""
    import json
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        with open(PARITY_LOG, "a") as f:
            f.write(json.dumps({"fixture": fixture, "kernel": kernel, "mode": mode, "norm_rel": errs[0], "max_abs": errs[1],
                                "max_elem_rel": errs[2]}) + "\n")
    except OSError:
        pass                      # a read-only checkout: the table is evidence, not part of the assertion


def worst_errs(errs):
    return tuple(max(e[j] for e in errs) for j in range(3))


def fmt_errs(e):
    return f"norm-rel {e[0]:.2e}  max-abs {e[1]:.2e}  max-elem-rel {e[2]:.2e}"


def cpu_metric_sums(samples, y_true, *, n_bins=10, picp_range=(2.5, 97.5), tau=1.0):
    """CPU stand-in for card_amd.metrics.regression_metric_sums in the gloo tests (same layout: (1, 4 + n_bins + 2)
    float64 = [n, sum SE, PICP count, sum NLL, QICE bin counts]), restating regression/card_regression.py:776-821 and
    :866-893 with numpy; trimmed range (0, 100), global precision tau, no y scaler."""
    from scipy.special import logsumexp
    g = np.asarray(samples, dtype=np.float32).reshape(samples.shape[0], -1)
    y = np.asarray(y_true, dtype=np.float32).reshape(-1, 1)
    rows, n_z = g.shape
    out = np.zeros((1, 4 + n_bins + 2))
    if rows:
        out[0, 0] = rows
        out[0, 1] = ((g.mean(axis=1, keepdims=True) - y) ** 2).sum()
        ci = np.percentile(g, q=list(picp_range), axis=1)
        out[0, 2] = ((y.T >= ci[0]) & (y.T <= ci[1])).sum()
        nll = -(logsumexp(-0.5 * tau * (y.T - g.T) ** 2., 0) - np.log(n_z) - 0.5 * np.log(2 * np.pi) + 0.5 * np.log(tau))
        out[0, 3] = nll.sum()
        quant = np.percentile(g, q=np.arange(n_bins + 1) * (100 / n_bins), axis=1)
        member = ((y.T - quant) > 0).astype(int).sum(axis=0)
        out[0, 4:] = np.bincount(member, minlength=n_bins + 2)
    return torch.from_numpy(out)
