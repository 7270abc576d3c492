"""Randomised cross-check of the two dispatch planners (card_guided.hip guided_run, card_stream.hip stream_plan): random
batch sizes, horizons, widths, row tilings and output modes; the default dispatch (calls cut into rounds of several kernel
forms, time slices) must equal the single-kernel form bit for bit.  usage: stress_planners.py [seed] [cases per width]"""
import os, random, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import _lib
from card_amd.sampler import Sampler, make_coef_table
from card_amd.synth import guided_state_dict, make_schedule
dev = "cuda:0"
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for hidden, env, alt in ((128, "CARD_AMD_COOP", "off"), (256, "CARD_AMD_STREAM", "big")):
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
        T = rnd.choice([1, 3, 7, 64, 65, 100, 130])
        x_dim = rnd.choice([1, 5, 13, 30])
        n_z = rnd.choice([1, 1, 3, 40])
        rows = max(1, rnd.choice([rnd.randint(1, 300), rnd.randint(300, 9000), rnd.randint(9000, 140000)]) // n_z)
        N = rows * n_z
        mode = rnd.choice(["last", "all" if N * T < 4e7 else "last", "select", "seg"])
        sd = guided_state_dict(x_dim, T, hidden=hidden, seed=case)
        alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
        coef = make_coef_table(alphas, omabs).to(dev)
        g = torch.Generator().manual_seed(case)
        x, y0 = torch.randn(rows, x_dim, generator=g).to(dev), torch.randn(rows, 1, generator=g).to(dev)
        outs = []
        for e in ("", alt):
            os.environ[env] = e
            s = Sampler.from_state_dict(sd, device=dev)
            kw = dict(seed=case, elem_offset=case * 7, n_z=n_z)
            if mode == "last": o = s.run(x, y0, y0, coef, T, **kw)
            elif mode == "all": o = s.run(x, y0, y0, coef, T, out_mode=_lib.OUT_ALL, **kw)
            elif mode == "select": o = s.run(x, y0, y0, coef, T, out_mode=_lib.OUT_SELECT, states=sorted({0, T // 2, T}), **kw)
            else:
                t_hi = max(0, T - 2)
                yi = torch.randn(N, 1, generator=g).to(dev) if not outs else yi
                o = s.run(x, y0, y0, coef, T, t_hi=t_hi, t_lo=min(t_hi, 1), y_init=yi, **kw)
            outs.append(o)
            if e == "": plan = s.run(x, y0, y0, coef, T, plan=True, **kw)
        ok = torch.equal(outs[0], outs[1]) and bool(torch.isfinite(outs[0]).all())
        bad += not ok
        print(hidden, "N", N, "T", T, "x_dim", x_dim, "n_z", n_z, mode, plan["kernel"][12:], plan["launches"], "OK" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
