#!/usr/bin/env python3
"""bench.py - reverse-diffusion sample-steps/sec of the fused p_sample_loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one whole p_sample_loop (all T timesteps, in-kernel Philox noise, last-only output) over one
batch of synthetic (row, z) chains already resident in HBM.  Default workload = BASELINE.json configs[1]:
toy 8-Gaussians shape, 10 000 test rows x 1000 z-samples, x_dim=1 (d_in=3), H=128, T=1000 -> 1e10 sample-steps
per GPU per step.  With N > 1 (torch.distributed.run, one rank per GPU, RCCL) every rank samples its own
shard of rows (weak scaling, no data-path collective); the evaluation sums are all-reduced once per step.

Rank 0 prints ONE JSON line with the driver contract's fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = 128
PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
TILE_M, WAVES_PER_CU, TILES_PER_WAVE_PER_LAUNCH = 32, 8, 16   # guided_run() chunking in card_amd/csrc/card_guided.hip


# shape presets: name -> (rows per GPU, n_z, x_dim, description)
WORKLOADS = {
    "toy": (10_000, 1000, 1, "toy 8-Gaussians shape (BASELINE configs[1]: 10k test x, 1000 z-samples, toy_8gauss.yml)"),
    "boston": (51, 100, 13, "UCI Boston plumbing shape (BASELINE configs[0]: 51 test rows x 100 z-samples, x_dim=13)"),
    "year": (64_418, 200, 90, "UCI YearPredictionMSD shape, one GPU's 1/8 shard (BASELINE configs[2]: 515 345 rows / 8, "
                              "200 z-samples, 90 features)"),
    "sweep": (1_000_000, 100, 13, "synthetic 1M rows x 100 z-samples, x_dim=13, H=128 (BASELINE configs[4])"),
}


def synth_state_dict(x_dim, T, seed=1234):
    """Random-init weights of regression/model.py:21 ConditionalGuidedModel (nn.Linear default init, gamma
    tables U(0,1) as regression/model.py:12, lin4 scaled x0.1 per SURVEY.md section 8d)."""
    g = torch.Generator().manual_seed(seed)
    d_in = x_dim + 2
    sd = {}

    def linear(name, i, o):
        b = 1.0 / math.sqrt(i)
        sd[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * b
        sd[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * b

    for k, i in (("lin1", d_in), ("lin2", H), ("lin3", H)):
        linear(k + ".lin", i, H)
        sd[k + ".embed.weight"] = torch.rand(T + 1, H, generator=g)
    linear("lin4", H, 1)
    sd["lin4.weight"] *= 0.1
    sd["lin4.bias"] *= 0.1
    return sd


def cpu_baseline(sd, x_dim, T, rows, steps, alphas, omabs):
    """The oracle (torch-CPU restatement, bit-equal to the reference here) timed on this host: `steps`
    timesteps of one reference-sized minibatch, noise drawn per step like the reference's randn_like."""
    from oracle import card_oracle as orc
    w = orc.GuidedWeights.from_state_dict(sd)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(rows, x_dim, generator=g)
    y0 = torch.randn(rows, 1, generator=g)
    coef = orc.make_coef_table(alphas, omabs)
    y = torch.randn(rows, 1) + y0
    with torch.no_grad():
        y = orc.p_sample(w, x, y, y0, y0, T - 1, coef, torch.randn_like(y))   # warm-up
        t0 = time.perf_counter()
        for t in range(T - 2, T - 2 - steps, -1):
            y = orc.p_sample(w, x, y, y0, y0, t, coef, torch.randn_like(y))
        dt = time.perf_counter() - t0
    return rows * steps / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="toy", choices=sorted(WORKLOADS),
                    help="shape preset (BASELINE.json configs); toy = configs[1], the headline line")
    ap.add_argument("--rows", type=int, default=None, help="test rows per GPU (default: preset)")
    ap.add_argument("--nz", type=int, default=None, help="z-samples per row (default: preset)")
    ap.add_argument("--x-dim", type=int, default=None)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    # stdout carries exactly one JSON line: route everything libraries print there (the RCCL banner ...) to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with one rank)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from card_amd import _lib
    from card_amd.sampler import Sampler, make_coef_table
    from oracle.card_oracle import make_schedule   # schedule constructor only (host-side setup, not measured)

    preset = WORKLOADS[a.workload]
    a.rows = preset[0] if a.rows is None else a.rows
    a.nz = preset[1] if a.nz is None else a.nz
    a.x_dim = preset[2] if a.x_dim is None else a.x_dim
    T, x_dim = a.T, a.x_dim
    N = a.rows * a.nz
    d_in = x_dim + 2
    flops_per_unit = 2 * H * (d_in + 2 * H + 1)      # algorithmic flops per sample-step
    sd = synth_state_dict(x_dim, T)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
    sampler = Sampler.from_state_dict(sd, device=dev)
    coef = make_coef_table(alphas, omabs).to(dev)

    # synthetic test rows resident in HBM; the (row, z) tiling of regression/card_regression.py:1027-1035 is done
    # inside the kernel (card_run_args.n_z)
    g = torch.Generator().manual_seed(1234 + rank)
    x_rows = torch.randn(a.rows, x_dim, generator=g)
    y0_rows = torch.randn(a.rows, 1, generator=g)
    x, y0 = x_rows.to(dev), y0_rows.to(dev)
    elem_offset = rank * N
    y_true = (y0_rows + 0.3 * torch.randn(a.rows, 1, generator=g)).to(dev)
    from card_amd.distributed import regression_metric_sums, all_reduce_sums

    def one_step(i):
        y = sampler.run(x, y0, y0, coef, T, seed=1234 + i, elem_offset=elem_offset, out_mode=_lib.OUT_LAST, n_z=a.nz)
        if use_dist:
            # evaluation sums (n, squared error, PICP count, QICE bin counts: card_regression.py:776-821) of this
            # rank's rows, then ONE 15-element RCCL all-reduce per loop -- the only collective on the path
            sums = regression_metric_sums(y.reshape(a.rows, a.nz, 1), y_true)
            all_reduce_sums(sums)
        return y

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(i)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(a.steps):
        ts = time.perf_counter()
        y = one_step(a.warmup + i)
        if rank == 0 and a.steps <= 4:
            torch.cuda.synchronize()
            print("Minibatch {} sampling took {:.4f} seconds.".format(i, time.perf_counter() - ts), file=sys.stderr)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(y).all()

    if rank == 0:
        units = float(N) * T * a.steps * world
        value = units / elapsed
        # dominant kernel = card_sample_guided_kernel; launches per step from guided_run()'s chunking
        cu = torch.cuda.get_device_properties(dev).multi_processor_count
        chunk = cu * WAVES_PER_CU * TILES_PER_WAVE_PER_LAUNCH * TILE_M
        launches = math.ceil(N / chunk) * a.steps
        ev_ms = ev0.elapsed_time(ev1)                      # HIP events on the launch stream, rank 0
        avg_launch_ms = ev_ms / launches
        achieved = (float(N) * T * a.steps * flops_per_unit) / (ev_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            # measured for ONE full-size launch of the same shape (tools/profile_pmc.sh); valid when this run's
            # launches are that size too
            if tj.get("x_dim") == x_dim and tj.get("T") == T and tj.get("rows_per_launch") == chunk and N >= chunk:
                traffic = tj.get("hbm_bytes_per_launch")
        out = {
            "metric": "reverse-diffusion sample-steps/sec (p_sample_loop)",
            "value": value, "unit": "sample-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{preset[3]}: {a.rows} test rows x {a.nz} z-samples per GPU, x_dim={x_dim} "
                                   f"(d_in={d_in}), H={H}, T={T}, in-kernel Philox noise, last-sample output, "
                                   "random-init weights",
                       "rows_per_gpu": a.rows, "n_z": a.nz, "T": T, "sharding": f"rows x{world}, no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                         "kernel": "card_sample_guided_kernel", "flops_per_sample_step": flops_per_unit,
                         "launches": launches, "avg_launch_ms": avg_launch_ms},
        }
        if world == 1 and not a.no_cpu_baseline:
            threads = torch.get_num_threads()
            rows_cpu, steps_cpu = 256 * 1000, 150     # one reference minibatch of toy_8gauss.yml (:67, :72)
            if a.workload != "toy":
                rows_cpu = min(N, 64 * a.nz)          # one reference minibatch: testing.batch_size 64 x n_z
            v, dt = cpu_baseline(sd, x_dim, T, rows_cpu, steps_cpu, alphas, omabs)
            out["cpu_baseline"] = {"value": v, "unit": "sample-steps/s", "cores": threads, "kind": "port",
                                   "sample": f"{steps_cpu} of the {T} timesteps of one reference minibatch "
                                             f"({rows_cpu} chains), torch CPU fp32, {dt:.1f} s"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
