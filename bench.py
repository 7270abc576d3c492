#!/usr/bin/env python3
"""bench.py - reverse-diffusion sample-steps/sec of the fused p_sample_loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one whole p_sample_loop (all T timesteps, in-kernel Philox noise, last-only output) over one
batch of synthetic (row, z) chains already resident in HBM.  Default workload = BASELINE.json configs[1]:
toy 8-Gaussians shape, 10 000 test rows x 1000 z-samples, x_dim=1 (d_in=3), H=128, T=1000 -> 1e10 sample-steps
per GPU per step.  With N > 1 (torch.distributed.run, one rank per GPU, RCCL) every rank samples its own
shard of rows (weak scaling, no data-path collective); the evaluation sums are all-reduced once per step.

Other shapes of BASELINE.json's configs are presets (--workload boston | year | sweep | sweep256 | mnist | cifar10); networks
other than the reference's H=128 / y_dim=1 regression MLP run on the streaming-GEMM back end.

Rank 0 prints ONE JSON line with the driver contract's fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3       # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MATRIX_TFLOPS = 2500.0      # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense); priced against only by --arith split
TILE_M, WAVES_PER_CU, TILES_PER_WAVE_PER_LAUNCH = 32, 8, 16   # guided_run() chunking in card_amd/csrc/card_guided.hip


# shape presets: name -> (kind, rows per GPU, n_z, x_dim, hidden, y_dim, description)
WORKLOADS = {
    "toy": ("guided", 10_000, 1000, 1, 128, 1,
            "toy 8-Gaussians shape (BASELINE configs[1]: 10k test x, 1000 z-samples, toy_8gauss.yml)"),
    "boston": ("guided", 51, 100, 13, 128, 1,
               "UCI Boston plumbing shape (BASELINE configs[0]: 51 test rows x 100 z-samples, x_dim=13)"),
    "year": ("guided", 64_418, 200, 90, 128, 1,
             "UCI YearPredictionMSD shape, one GPU's 1/8 shard (BASELINE configs[2]: 515 345 rows / 8, 200 z-samples, "
             "90 features)"),
    "sweep": ("guided", 1_000_000, 100, 13, 128, 1,
              "synthetic 1M rows x 100 z-samples, x_dim=13, hidden 128 (BASELINE configs[4])"),
    "sweep256": ("guided", 100_000, 100, 13, 256, 1,
                 "synthetic rows x 100 z-samples, x_dim=13, hidden 256 (BASELINE configs[4], width-256 leg; 100k of the 1M "
                 "rows per GPU by default, --rows 1000000 for all)"),
    "mnist": ("cls", 250, 100, 128, 128, 10,
              "MNIST head shape (classification/configs/mnist.yml: F=128, C=10, 100 z-samples, one reference test batch of 250 "
              "images per GPU; x_feat and y_0_hat synthetic)"),
    "cifar10": ("cls", 250, 100, 4096, 4096, 10,
                "CIFAR-10 head shape (BASELINE configs[3]: F=4096, C=10, 100 z-samples; one reference test batch of 250 "
                "images per GPU by default, cifar10.yml:66,:71; x_feat and y_0_hat synthetic)"),
}


def cpu_baseline(w, x, y0, T, steps, alphas, omabs):
    """The oracle (torch-CPU restatement, bit-equal to the reference here) timed on this host: `steps`
    timesteps of one reference-sized minibatch, noise drawn per step like the reference's randn_like."""
    from oracle import card_oracle as orc
    coef = orc.make_coef_table(alphas, omabs)
    y = torch.randn_like(y0) + y0
    with torch.no_grad():
        y = orc.p_sample(w, x, y, y0, y0, T - 1, coef, torch.randn_like(y))   # warm-up
        t0 = time.perf_counter()
        for t in range(T - 2, T - 2 - steps, -1):
            y = orc.p_sample(w, x, y, y0, y0, t, coef, torch.randn_like(y))
        dt = time.perf_counter() - t0
    return y0.shape[0] * steps / dt, dt


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
    ap.add_argument("--hidden", type=int, default=None, help="hidden width (128: persistent kernel; other multiples of 128: GEMM back end)")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--arith", default="f32", choices=["f32", "split"],
                    help="f32 = FP32 MFMA (default, the headline line); split = CARD_ARITH_SPLIT_BF16: the hidden layers as six "
                         "bf16 MFMAs per k-step on exact three-way bf16 cuts of the fp32 operands (hidden-128 regression only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend under torch.distributed.run: nccl (= RCCL, the default and what the driver "
                         "uses) or gloo (a rehearsal of the multi-rank path on a box with fewer GPUs than ranks: barrier, the "
                         "evaluation sums and the max-over-ranks time go through the host)")
    ap.add_argument("--share-device", action="store_true",
                    help="every rank uses cuda:0 instead of cuda:LOCAL_RANK (with --backend gloo: N ranks rehearsed on one GPU; "
                         "the ranks then share the GPU's time, so `value` says nothing about scaling)")
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
    if a.share_device:
        if a.backend != "gloo" and world > 1:
            sys.exit("--share-device puts every rank on cuda:0; RCCL needs one GPU per rank (use --backend gloo)")
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with one rank)
    if use_dist:
        import torch.distributed as dist
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from card_amd import _lib
    from card_amd.sampler import Sampler, make_coef_table
    from card_amd.synth import guided_state_dict, cls_state_dict, make_schedule

    kind, p_rows, p_nz, p_xdim, p_hidden, y_dim, desc = WORKLOADS[a.workload]
    a.rows = p_rows if a.rows is None else a.rows
    a.nz = p_nz if a.nz is None else a.nz
    a.x_dim = p_xdim if a.x_dim is None else a.x_dim
    H = p_hidden if a.hidden is None else a.hidden
    T, x_dim = a.T, a.x_dim
    N = a.rows * a.nz
    g = torch.Generator().manual_seed(1234 + rank)
    alphas, _, omabs = make_schedule("linear", T, 1e-4, 0.02)
    coef = make_coef_table(alphas, omabs).to(dev)
    if kind == "guided":
        d_in = x_dim + 2
        flops_per_unit = 2 * H * (d_in + 2 * H + 1)      # algorithmic flops per sample-step
        sd = guided_state_dict(x_dim, T, hidden=H)
        sampler = Sampler.from_state_dict(sd, device=dev)
        # synthetic test rows resident in HBM; the (row, z) tiling of regression/card_regression.py:1027-1035 is done
        # inside the kernel (card_run_args.n_z)
        x_rows = torch.randn(a.rows, x_dim, generator=g)
        y0_rows = torch.randn(a.rows, 1, generator=g)
        y_true = (y0_rows + 0.3 * torch.randn(a.rows, 1, generator=g)).to(dev)
    else:
        H = x_dim                                        # F: x is x_feat = norm(encoder_x(x)), hoisted out of the loop
        d_in = 2 * y_dim
        flops_per_unit = 2 * H * (d_in + 2 * H + y_dim)  # theta only, encoder hoisted
        sd = cls_state_dict(H, y_dim, T, 3)
        sampler = Sampler.from_cls_state_dict(sd, device=dev)
        x_rows = torch.randn(a.rows, H, generator=g)
        y0_rows = torch.softmax(torch.randn(a.rows, y_dim, generator=g) * 2, 1)
        y_true = None
        cls_labels = y0_rows.argmax(1).to(dev)
    arith = _lib.ARITH_SPLIT_BF16 if a.arith == "split" else _lib.ARITH_F32_MFMA
    if arith == _lib.ARITH_SPLIT_BF16 and not (kind == "guided" and H == 128 and y_dim == 1):
        sys.exit("--arith split is built for the hidden-128 regression sampler")
    x, y0 = x_rows.to(dev), y0_rows.to(dev)
    elem_offset = rank * N * y_dim
    from card_amd.distributed import all_reduce_sums
    from card_amd.metrics import regression_metric_sums, classification_metric_sums

    def one_step(i):
        y = sampler.run(x, y0, y0, coef, T, seed=1234 + i, elem_offset=elem_offset, out_mode=_lib.OUT_LAST, n_z=a.nz,
                        arith=arith)
        if use_dist:
            # evaluation sums of this rank's rows, then ONE small RCCL all-reduce per loop -- the only collective
            if kind == "guided":
                # n, sum SE, PICP count, sum NLL, QICE bin counts (card_regression.py:776-821, :866-893): 16 numbers from
                # the HIP metrics kernel (one workgroup sorts a row's n_z samples in LDS)
                sums = regression_metric_sums(y.reshape(a.rows, a.nz, 1), y_true, tau=1.0)
            else:
                # images, correct majority votes, summed instance accuracies, t-test rejections (card_classification.py:
                # 562-596, :1033-1060) from the HIP metrics kernel (one workgroup per image: softmax, votes, CI, t-test)
                sums = classification_metric_sums(y, cls_labels, n_z=a.nz)
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
        if rank == 0 and a.steps <= 4 and float(N) * T >= 1e9:      # (a per-step sync would dominate a 4 ms step)
            torch.cuda.synchronize()
            print("Minibatch {} sampling took {:.4f} seconds.".format(i, time.perf_counter() - ts), file=sys.stderr)
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(y).all()

    if rank == 0:
        units = float(N) * T * a.steps * world
        value = units / elapsed
        ev_ms = ev0.elapsed_time(ev1)                      # HIP events on the launch stream, rank 0
        # kernel name, launches and executed flops come from the library itself: card_sampler_plan is a dry run of the
        # dispatch code this call went through (no Python copy of the thresholds / slicing / chunking to drift)
        pl = sampler.run(x, y0, y0, coef, T, seed=1234, elem_offset=elem_offset, out_mode=_lib.OUT_LAST, n_z=a.nz, arith=arith,
                         plan=True)
        kernel, launches = pl["kernel"], pl["launches"] * a.steps
        if pl["backend"] == 1:
            kernel += " (+ layer-1 GEMM and head/update launches, all inside the timed region)"
        assert abs(pl["flops_per_sample_step"] - flops_per_unit) < 0.5, (pl, flops_per_unit)
        chunk = None
        if pl["kernel"].startswith("card_sample_guided_kernel"):
            # chains of one full launch of the persistent kernel (guided_run: CUs x 8 waves x 16 tiles x 32 chains), the key
            # of the measured-traffic table below
            cu = torch.cuda.get_device_properties(dev).multi_processor_count
            chunk = cu * WAVES_PER_CU * TILES_PER_WAVE_PER_LAUNCH * TILE_M
        avg_launch_ms = ev_ms / launches
        achieved = (float(N) * T * a.steps * flops_per_unit) / (ev_ms * 1e-3) / 1e12
        # flops the kernels actually issue per sample-step (`achieved` prices the ALGORITHMIC count of SURVEY.md 8d, which
        # the hoisted layer 1 undercuts -- on the Year shape by 26 % -- and K / class padding overshoots): from the plan
        executed = pl["flops_executed_per_sample_step"]
        if arith == _lib.ARITH_SPLIT_BF16:
            executed = 2 * H * (2 * H) + 2 * H * 2 + 2 * H        # (the fp32-equivalent count; the bf16 flops are priced below)
        achieved_exec = achieved * executed / flops_per_unit
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if chunk is not None and os.path.exists(tpath):
            # measured for ONE full-size launch of the same shape (tools/profile_pmc.sh); valid when this run's
            # launches are that size too
            for tj in json.load(open(tpath)).get("entries", []):
                if (tj.get("x_dim"), tj.get("hidden"), tj.get("T"), tj.get("rows_per_launch")) == (x_dim, H, T, chunk) and N >= chunk:
                    traffic = tj.get("hbm_bytes_per_launch")
        out = {
            "metric": "reverse-diffusion sample-steps/sec (p_sample_loop)",
            "value": value, "unit": "sample-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{desc}: {a.rows} test rows x {a.nz} z-samples per GPU, x_dim={x_dim} "
                                   f"(d_in={d_in}), H={H}, y_dim={y_dim}, T={T}, in-kernel Philox noise, last-sample "
                                   "output, random-init weights",
                       "rows_per_gpu": a.rows, "n_z": a.nz, "T": T, "sharding": f"rows x{world}, no data-path collective",
                       "backend": (a.backend if use_dist else None), "ranks_share_one_gpu": bool(a.share_device and world > 1)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                         "kernel": kernel, "flops_per_sample_step": flops_per_unit,
                         "flops_executed_per_sample_step": executed, "achieved_executed": achieved_exec,
                         "frac_executed": achieved_exec / PEAK_FP32_MATRIX_TFLOPS,
                         "launches": launches, "avg_launch_ms": avg_launch_ms},
        }
        if arith == _lib.ARITH_SPLIT_BF16:
            # opt-in arithmetic, not the headline: the two hidden layers run as 6 bf16 MFMAs per 16-deep k-step, so the
            # matrix pipe that bounds this kernel is the bf16 one; `achieved` = the bf16 flops it issues
            ex_bf16 = pl["flops_executed_per_sample_step"]
            ach = (float(N) * T * a.steps * ex_bf16) / (ev_ms * 1e-3) / 1e12
            out["arith"] = "split_bf16 (CARD_ARITH_SPLIT_BF16: fp32 operands cut exactly into 3 bf16 parts, fp32 accumulation)"
            out["roofline"].update({"achieved": ach, "peak": PEAK_BF16_MATRIX_TFLOPS, "frac": ach / PEAK_BF16_MATRIX_TFLOPS,
                                    "flops_executed_per_sample_step": ex_bf16, "achieved_executed": ach,
                                    "frac_executed": ach / PEAK_BF16_MATRIX_TFLOPS, "traffic": None,
                                    "algorithmic_fp32_tflops": achieved,
                                    "algorithmic_over_fp32_matrix_peak": achieved / PEAK_FP32_MATRIX_TFLOPS})
        if (world == 1 and not a.no_cpu_baseline and arith == _lib.ARITH_F32_MFMA and kind == "guided" and H == 128
                and y_dim == 1 and N > 7 * 256 * 32):
            # beside the headline, never instead of it: ONE more pass of the same workload through the opt-in split-bf16
            # arithmetic (CARD_ARITH_SPLIT_BF16: fp32 operands cut exactly into three bf16 parts, six bf16 MFMAs per
            # k-step, fp32 accumulation; same parity bar).  Not part of `value` / `ms_per_step`.
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sampler.run(x[:256], y0[:256], y0[:256], coef, T, seed=1, n_z=a.nz, arith=_lib.ARITH_SPLIT_BF16)   # warm-up
            s0.record()
            ys = sampler.run(x, y0, y0, coef, T, seed=4321, elem_offset=elem_offset, n_z=a.nz, arith=_lib.ARITH_SPLIT_BF16)
            s1.record()
            torch.cuda.synchronize()
            ms = s0.elapsed_time(s1)
            out["opt_in_split_bf16"] = {"value": float(N) * T / (ms * 1e-3), "unit": "sample-steps/s", "ms_per_step": ms,
                                        "finite": bool(torch.isfinite(ys).all()), "kernel": "card_sample_split_kernel",
                                        "note": "opt-in arithmetic (card_run_args.arith = CARD_ARITH_SPLIT_BF16), fp32 accuracy on "
                                                "the bf16 matrix pipe; the headline above is FP32 MFMA as north_star specifies"}
        if world == 1 and not a.no_cpu_baseline:
            threads = torch.get_num_threads()
            from oracle import card_oracle as orc
            gc = torch.Generator().manual_seed(7)
            if kind == "guided":
                rows_cpu, steps_cpu = 256 * 1000, 150     # one reference minibatch of toy_8gauss.yml (:67, :72)
                if a.workload != "toy":
                    rows_cpu = min(N, 64 * a.nz)          # one reference minibatch: testing.batch_size 64 x n_z
                if H != 128:
                    steps_cpu = 60
                w = orc.GuidedWeights.from_state_dict(sd)
                xc, yc = torch.randn(rows_cpu, x_dim, generator=gc), torch.randn(rows_cpu, 1, generator=gc)
            else:
                rows_cpu, steps_cpu = 2500, (4 if H >= 2048 else 200)   # a tenth of one reference test batch (250 images x 100 z)
                w = orc.ClsWeights.from_state_dict(sd)
                xc = torch.randn(rows_cpu, H, generator=gc)
                yc = torch.softmax(torch.randn(rows_cpu, y_dim, generator=gc) * 2, 1)
            v, dt = cpu_baseline(w, xc, yc, T, steps_cpu, alphas, omabs)
            out["cpu_baseline"] = {"value": v, "unit": "sample-steps/s", "cores": threads, "kind": "port",
                                   "sample": f"{steps_cpu} of the {T} timesteps of one reference minibatch "
                                             f"({rows_cpu} chains), torch CPU fp32, {dt:.1f} s"}
            if threads > 4:
                # the reference's own default, torch.set_num_threads(args.thread = 4) (regression/main.py:25, :259-261)
                torch.set_num_threads(4)
                v4, _ = cpu_baseline(w, xc, yc, T, max(2, steps_cpu // 4), alphas, omabs)
                torch.set_num_threads(threads)
                out["cpu_baseline"]["at_reference_default_threads"] = {"cores": 4, "value": v4}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
