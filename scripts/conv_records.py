"""Per-launch records of the convolution kernels in one eager training step (variant, duration net of the event bracket,
algorithmic MB, GFLOP, GB/s), grouped by (variant, bytes, flops):  python scripts/conv_records.py [model] [batch] [json] [f32|bf16]
With a third argument the per-variant sums (launches, algorithmic bytes and flops per step) are also written as JSON."""
import ctypes as C
import importlib
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    model_name = sys.argv[1] if len(sys.argv) > 1 else "enet-naive-upsampling"
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    trainers = bench.pkg("trainers")
    cfg = {"input_shape": [None, None, 3], "optimizer": {"name": "adam", "learning_rate": 5e-4}, "ignore_classes": [0],
           "random_seed": 14874, "num_labeled_per_batch": batch}
    if len(sys.argv) > 4 and sys.argv[4] == "bf16":
        cfg["activation_dtype"] = "bf16"
    tr = trainers.SegmentationTrainer("segmentation_supervised", model_name, "bench", config=cfg, num_classes=24)
    model = tr.model
    images, labels, weights = bench.synthetic_batch(batch, 512, 512, 24, 14874, torch.device(tr.device))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        model.set_batch([images, labels, weights, np.zeros((batch, 1), dtype=np.int32)])
    L = importlib.import_module("semantic-segmentation_amd._lib").lib()
    model.use_hip_graph = False
    with torch.cuda.stream(stream):
        model.train_step_async()
        stream.synchronize()
        L.profile_begin()
        for _ in range(3):
            model.train_step_async()
        stream.synchronize()
        cnt = C.c_int32()
        L.profile_end(C.byref(cnt))
    ov = float(L.profile_bracket_overhead_ms())
    name = C.create_string_buffer(128)
    ms, by, fl = C.c_float(), C.c_double(), C.c_double()
    agg = defaultdict(lambda: [0.0, 0])
    for i in range(cnt.value):
        L.profile_get(i, name, 128, C.byref(ms), C.byref(by), C.byref(fl))
        k = (name.value.decode(), round(by.value / 1e6, 2), round(fl.value / 1e9, 3))
        agg[k][0] += max(ms.value - ov, 0.0)
        agg[k][1] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    print("bracket overhead %.2f us" % (ov * 1e3))
    for (k, mb, gf), (t, n) in rows[:40]:
        us = t / n * 1e3
        print("%-52s n/step %3d  %7.1f us  %7.2f MB  %7.3f GFLOP  %7.1f GB/s  %6.2f TF  total %6.3f ms/step" % (
            k[:52], n // 3, us, mb, gf, mb / us * 1e3, gf / us * 1e3, t / 3))
    if len(sys.argv) > 3:
        import json
        per = defaultdict(lambda: {"launches_per_step": 0, "bytes_per_step": 0.0, "flops_per_step": 0.0, "ms_per_step_eager": 0.0})
        for (k, mb, gf), (t, n) in agg.items():
            e = per[k]
            e["launches_per_step"] += n // 3
            e["bytes_per_step"] += mb * 1e6 * (n // 3)
            e["flops_per_step"] += gf * 1e9 * (n // 3)
            e["ms_per_step_eager"] += t / 3
        json.dump(per, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
