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
for rows, nz in ((64, 1000), (64, 200), (256, 1000)):
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
