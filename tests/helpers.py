"""Shared helpers for the test-suite: fixture loading and the parity metric."""
import glob
import os

import numpy as np
import torch

from oracle import card_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names(kind=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    names = [n for n in names if n != "schedules"]
    if kind:
        names = [n for n in names if n.startswith(kind)]
    return names


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    sd = {k[3:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("sd.")}
    t = lambda k: torch.from_numpy(d[k])
    out = dict(raw=d, sd=sd, T=int(d["T"]), kind=str(d["kind"]),
               x=t("x") if "x" in d else t("z"), y0hat=t("y0hat"), yTmean=t("yTmean"), alphas=t("alphas"), omabs=t("omabs"),
               alphas_bar_sqrt=t("alphas_bar_sqrt"), noise=t("noise"), traj=t("traj"),
               traj_idx=[int(i) for i in d["traj_idx"]], probe_t=[int(i) for i in d["probe_t"]])
    if out["kind"] == "regression":
        out["cat_x"], out["cat_y_pred"] = bool(d["cat_x"]), bool(d["cat_y_pred"])
        out["weights"] = orc.GuidedWeights.from_state_dict(sd, out["cat_x"], out["cat_y_pred"])
        out["x_in"] = out["x"]
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
