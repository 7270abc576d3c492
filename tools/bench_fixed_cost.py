#!/usr/bin/env python3
"""Fixed cost of a launch of the cooperative regression kernels: ms per p_sample_loop call at T = 250 / 500 / 1000 for a few
batch sizes, and the (slope, intercept) of the line through them -- what a time slice pays on top of its timesteps."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule

dev, x_dim = "cuda:0", int(os.environ.get("XDIM", 90))
g = torch.Generator().manual_seed(0)
for N in [int(v) for v in sys.argv[1:]] or [4096, 8192, 12800, 16384]:
    x, y0 = torch.randn(N, x_dim, generator=g).to(dev), torch.randn(N, 1, generator=g).to(dev)
    pts = []
    for T in (250, 500, 1000):
        s = Sampler.from_state_dict(guided_state_dict(x_dim, T), device=dev)
        alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
        coef = make_coef_table(alphas, omabs).to(dev)
        s.run(x, y0, y0, coef, T, seed=1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(5):
            s.run(x, y0, y0, coef, T, seed=2 + i)
        e1.record(); torch.cuda.synchronize()
        pts.append((T, e0.elapsed_time(e1) / 5))
    slope = (pts[2][1] - pts[0][1]) / (pts[2][0] - pts[0][0])
    print(json.dumps({"chains": N, "x_dim": x_dim, "ms_at_T": {str(t): round(m, 4) for t, m in pts}, "us_per_timestep": round(slope * 1e3, 3),
                      "fixed_us": round((pts[2][1] - slope * pts[2][0]) * 1e3, 1), "plan": s.run(x, y0, y0, coef, 1000, seed=1, plan=True)}, default=str), flush=True)
