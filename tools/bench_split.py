#!/usr/bin/env python3
"""CARD_ARITH_SPLIT_BF16 against the default FP32-MFMA kernels: one p_sample_loop call (T = 1000, hidden 128, in-kernel
Philox noise, last-sample output, in-kernel row tiling) over a range of batch sizes; milliseconds and sample-steps/s."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule

dev, T = "cuda:0", 1000
x_dim = int(os.environ.get("X_DIM", "13"))
n_z = int(os.environ.get("N_Z", "100"))
sd = guided_state_dict(x_dim, T)
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
coef = make_coef_table(alphas, omabs).to(dev)
s = Sampler.from_state_dict(sd, device=dev)
sizes = [int(v) for v in sys.argv[1:]] or [3200, 8192, 12800, 24576, 49152, 98304, 262144, 1048576]
g = torch.Generator().manual_seed(0)
for N in sizes:
    rows = max(1, N // n_z)
    x, y0 = torch.randn(rows, x_dim, generator=g).to(dev), torch.randn(rows, 1, generator=g).to(dev)
    row = {"chains": rows * n_z, "x_dim": x_dim}
    for name, arith in (("f32_mfma", _lib.ARITH_F32_MFMA), ("split_bf16", _lib.ARITH_SPLIT_BF16)):
        s.run(x, y0, y0, coef, T, seed=1, n_z=n_z, arith=arith)
        torch.cuda.synchronize()
        reps = 3 if N <= 300000 else 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            y = s.run(x, y0, y0, coef, T, seed=2 + i, n_z=n_z, arith=arith)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        row[name + "_ms"] = round(ms, 3)
        row[name + "_steps_per_s"] = rows * n_z * T / (ms * 1e-3)
    row["speedup"] = round(row["f32_mfma_ms"] / row["split_bf16_ms"], 3)
    print(json.dumps(row), flush=True)
