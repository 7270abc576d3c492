#!/bin/bash
# One bench line per BASELINE.json shape preset (1 GPU).  Results -> gpurun_out/workloads.jsonl
set -u
out=gpurun_out/workloads.jsonl; : > $out
timeout -k 10 300 python3 bench.py --workload toy --steps 2 --warmup 1 2>/dev/null >> $out || exit 1
timeout -k 10 300 python3 bench.py --workload boston --steps 20 --warmup 3 2>/dev/null >> $out || exit 1
timeout -k 10 300 python3 bench.py --workload year --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null >> $out || exit 1
timeout -k 10 400 python3 bench.py --workload sweep --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null >> $out || exit 1
timeout -k 10 400 python3 bench.py --workload sweep256 --rows 16384 --nz 128 --steps 1 --warmup 0 2>/dev/null >> $out || exit 1
timeout -k 10 400 python3 bench.py --workload mnist --steps 10 --warmup 2 2>/dev/null >> $out || exit 1
timeout -k 10 400 python3 bench.py --workload cifar10 --steps 1 --warmup 0 --T 200 2>/dev/null >> $out || exit 1
timeout -k 10 400 python3 bench.py --workload cifar10 --rows 2560 --steps 1 --warmup 0 --T 20 --no-cpu-baseline 2>/dev/null >> $out || exit 1
python3 - <<'PY'
import json
for l in open("gpurun_out/workloads.jsonl"):
    d = json.loads(l)
    cb = d.get("cpu_baseline", {})
    print("%-16s %.4e sample-steps/s  %.1f ms/step  %.1f TFLOP/s  frac %.3f  cpu %.3e (%s cores)" % (
        d["config"]["workload"][:8] + " H=%s" % d["config"]["workload"].split("H=")[1].split(",")[0], d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"],
        cb.get("value", float("nan")), cb.get("cores", "-")))
PY
