#!/usr/bin/env python3
"""Small-batch latency of one p_sample_loop call (T = 1000, hidden 128, Boston input width): cooperative kernel vs
the persistent kernel alone, over the batch sizes the reference's per-minibatch test loop produces
(regression/card_regression.py:1043: 64 test rows x n_z chains)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule

dev, T, x_dim = "cuda:0", 1000, 13
sd = guided_state_dict(x_dim, T)
alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
coef = make_coef_table(alphas, omabs).to(dev)
samplers = {}
for mode in (os.environ.get("COOP_MODE", "1"), "0"):
    os.environ["CARD_AMD_COOP"] = mode
    samplers[mode] = Sampler.from_state_dict(sd, device=dev)
sizes = [int(v) for v in sys.argv[1:]] or [32, 1024, 5100, 6400, 8192, 12800, 16384, 24576, 32768, 49152, 64000, 131072]
g = torch.Generator().manual_seed(0)
for N in sizes:
    x, y0 = torch.randn(N, x_dim, generator=g).to(dev), torch.randn(N, 1, generator=g).to(dev)
    row = {"chains": N}
    for mode, s in samplers.items():
        s.run(x, y0, y0, coef, T, seed=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(3):
            y = s.run(x, y0, y0, coef, T, seed=2 + i)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        row["coop_ms" if mode != "0" else "persistent_ms"] = round(ms, 3)
    row["sample_steps_per_s_default"] = N * T / (row["coop_ms"] * 1e-3)
    print(json.dumps(row), flush=True)
