#!/usr/bin/env python3
"""The reference's own call, unmodified: p_sample_loop(model, x_tile, y_0_hat_tile, y_T_mean_tile, T, alphas, omabs) on
pre-tiled tensors of one test minibatch (regression/card_regression.py:1027-1043: 64 rows x 1000 z), returning the list of
T + 1 states.  Times the shim end to end (Python included)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import card_amd
from card_amd.synth import guided_state_dict, make_schedule

dev, T, x_dim = "cuda:0", 1000, 13


class CL(torch.nn.Module):
    def __init__(self, i, o, n):
        super().__init__()
        self.lin = torch.nn.Linear(i, o)
        self.embed = torch.nn.Embedding(n, o)


class Guided(torch.nn.Module):            # attribute layout of regression/model.py:21
    def __init__(self, sd):
        super().__init__()
        self.cat_x, self.cat_y_pred = True, True
        self.lin1, self.lin2, self.lin3 = CL(x_dim + 2, 128, T + 1), CL(128, 128, T + 1), CL(128, 128, T + 1)
        self.lin4 = torch.nn.Linear(128, 1)
        self.load_state_dict(sd)


m = Guided(guided_state_dict(x_dim, T)).to(dev).eval()
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
alphas, omabs = alphas.to(dev), omabs.to(dev)
ONLY = os.environ.get("DROPIN_ONLY", "")          # "cls128": only the F = 128 classification loops (for rocprofv3 runs)
for rows, nz in (() if ONLY else ((64, 1000), (64, 200), (256, 1000))):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, x_dim, generator=g).to(dev)
    y0 = torch.randn(rows, 1, generator=g).to(dev)
    # the reference's tiling (:1027-1035)
    x_tile = x.repeat(nz, 1, 1).transpose(0, 1).flatten(0, 1)
    y0_tile = y0.repeat(nz, 1, 1).transpose(0, 1).flatten(0, 1)
    for only_last in (False, True):
        args = (m, x_tile, y0_tile, y0_tile, T, alphas, omabs)
        kw = {"states": [T]} if only_last else {}
        card_amd.p_sample_loop(*args, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            seq = card_amd.p_sample_loop(*args, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(json.dumps({"rows": rows, "n_z": nz, "chains": rows * nz, "returns": "y_0 only" if only_last else "list of T+1 states",
                          "seconds_per_minibatch": round(dt, 4), "sample_steps_per_s": rows * nz * T / dt}), flush=True)


# ------------------------------------------------------------------------------------------------
# The classification test loop: classification/card_classification.py:656-671 (p_sample_loop_only_y_0) calls p_sample
# T - 1 times and p_sample_t_1to0 once, each on the whole tiled batch (250 images x 100 z, mnist.yml / cifar10.yml).
# Per call: host time of the shim (Python + one launch) next to the kernel time of that step.
# ------------------------------------------------------------------------------------------------
from card_amd.synth import cls_state_dict


class Cls(torch.nn.Module):               # attribute layout of classification/model.py:22 (identity encoder: x is x_feat)
    def __init__(self, sd):
        super().__init__()
        F, d1 = sd["lin1.lin.weight"].shape
        C, n = sd["lin4.weight"].shape[0], sd["lin1.embed.weight"].shape[0]
        self.guidance = True
        self.encoder_x, self.norm = torch.nn.Identity(), torch.nn.Identity()
        self.lin1, self.lin2, self.lin3 = CL(d1, F, n), CL(F, F, n), CL(F, F, n)
        self.unetnorm1, self.unetnorm2, self.unetnorm3 = (torch.nn.BatchNorm1d(F) for _ in range(3))
        self.lin4 = torch.nn.Linear(F, C)
        self.load_state_dict(sd)


def only_y_0(model, x, y_0_hat, y_T_mean, n_steps, alphas, omabs):
    cur_y = torch.randn_like(y_T_mean) + y_T_mean
    for i in reversed(range(1, n_steps)):
        cur_y = card_amd.p_sample(model, x, cur_y, y_0_hat, y_T_mean, i, alphas, omabs)
    return card_amd.p_sample_t_1to0(model, x, cur_y, y_0_hat, y_T_mean, omabs)


CASES = ((128, 10, 250, 100, 1000), (128, 10, 1, 32, 1000), (4096, 10, 250, 100, 20))
for F, C, rows, nz, Tc in (CASES[:1] if ONLY == "cls128" else CASES):
    mc = Cls(cls_state_dict(F, C, Tc, 3)).to(dev).eval()
    al, _, om = make_schedule("linear", Tc, 1e-4, 0.02)
    al, om = al.to(dev), om.to(dev)
    g = torch.Generator().manual_seed(1)
    N = rows * nz
    xf = torch.randn(N, F, generator=g).to(dev)
    y0 = torch.softmax(torch.randn(N, C, generator=g) * 2, 1).to(dev)
    only_y_0(mc, xf, y0, y0, Tc, al, om)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    only_y_0(mc, xf, y0, y0, Tc, al, om)
    t_issue = time.perf_counter() - t0            # host side: all T calls issued (the stream may still be running)
    torch.cuda.synchronize()
    t_loop = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    card_amd.p_sample_loop(mc, xf, y0, y0, Tc, al, om, True)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({"loop": "p_sample_loop_only_y_0 (T calls through the shim)", "F": F, "C": C, "chains": N, "T": Tc,
                      "host_us_per_call": round(t_issue / Tc * 1e6, 1), "loop_us_per_step": round(t_loop / Tc * 1e6, 1),
                      "fused_loop_us_per_step": round(e0.elapsed_time(e1) / Tc * 1e3, 1),
                      "sample_steps_per_s": N * Tc / t_loop}), flush=True)
