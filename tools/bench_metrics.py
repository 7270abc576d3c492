#!/usr/bin/env python3
"""Time of the on-device metric kernels next to the sampling they follow (one MI355X):
card_regression_row_metrics on the bench shape's samples (10 000 rows x 1000 z) and the Year / Boston minibatch shapes,
card_classification_row_metrics on the CIFAR-10 / MNIST test minibatch (250 images x 100 z x 10 classes) and a CIFAR-100 one.
Algorithmic bytes = the samples read once + the per-row results written; prints one JSON line per shape."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from card_amd import metrics

dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for rows, n_z in [(10000, 1000), (64, 200), (51, 100), (64418, 200)]:
    s = torch.randn(rows, n_z, generator=g).to(dev)
    y = torch.randn(rows, generator=g).to(dev)
    ms = timed(lambda: metrics.regression_row_metrics(s, y))
    by = rows * n_z * 4 + rows * 20
    print(json.dumps({"kernel": "row_metrics_kernel", "rows": rows, "n_z": n_z, "ms": round(ms, 4), "algorithmic_bytes": by,
                      "GB_per_s": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 4),
                      "rows_per_s": round(rows / ms * 1e3)}))
for rows, n_z, C in [(250, 100, 10), (10000, 100, 10), (250, 100, 100), (250, 1000, 10)]:
    s = torch.randn(rows * n_z, C, generator=g).to(dev)
    lab = torch.randint(0, C, (rows,), generator=g).to(dev)
    ms = timed(lambda: metrics.classification_row_metrics(s, lab, n_z=n_z))
    by = rows * n_z * C * 4 + rows * (C * 12 + 40)
    print(json.dumps({"kernel": "cls_row_metrics_kernel", "images": rows, "n_z": n_z, "C": C, "ms": round(ms, 4), "algorithmic_bytes": by,
                      "GB_per_s": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / 8000, 4),
                      "images_per_s": round(rows / ms * 1e3)}))
