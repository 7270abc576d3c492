#!/usr/bin/env python3
"""Throughput of the classification head back end at the CIFAR-10 test shape (BASELINE configs[3]):
N = 250 images x 100 z-samples = 25 000 chains per reference minibatch (--rows / --nz), C = 10, F = 4096 (cifar10.yml:15-16, :66, :71), synthetic x_feat /
softmax y_0_hat, random-init weights.  Prints per-timestep time, sample-steps/s and algorithmic TFLOP/s
(2 F (2C + 2F + C) flops per sample-step)."""
import argparse, json, math, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import cls_state_dict, make_schedule

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=250); ap.add_argument("--nz", type=int, default=100)
ap.add_argument("--F", type=int, default=4096)
ap.add_argument("--C", type=int, default=10); ap.add_argument("--T", type=int, default=40)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
a.N = a.rows * a.nz
dev = "cuda:0"
sd = cls_state_dict(a.F, a.C, a.T, 3)
s = Sampler.from_cls_state_dict(sd, device=dev)
g = torch.Generator().manual_seed(1)
xf = torch.randn(a.rows, a.F, generator=g).to(dev)          # x_feat per test image; chains tiled in-kernel (n_z)
y0 = torch.softmax(torch.randn(a.rows, a.C, generator=g) * 2, 1).to(dev)
alphas, _, omabs = make_schedule("linear", a.T, 1e-4, min(0.3, 12.0 / a.T))   # keep alpha_bar_T representable in fp32
coef = make_coef_table(alphas, omabs).to(dev)
s.run(xf, y0, y0, coef, a.T, seed=1, t_hi=a.T - 1, t_lo=a.T - 2, y_init=y0.repeat_interleave(a.nz, 0), n_z=a.nz)   # warm-up: 2 timesteps
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.reps):
    y = s.run(xf, y0, y0, coef, a.T, seed=2 + i, n_z=a.nz)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
flops = 2 * a.F * (2 * a.C + 2 * a.F + a.C)
rate = a.N * a.T / (ms * 1e-3)
print(json.dumps({"rows": a.rows, "n_z": a.nz, "N": a.N, "F": a.F, "C": a.C, "T": a.T, "ms_per_timestep": ms / a.T, "sample_steps_per_s": rate,
                  "algorithmic_TFLOPs": rate * flops / 1e12, "frac_of_157.3": rate * flops / 157.3e12,
                  "finite": bool(torch.isfinite(y).all())}))
