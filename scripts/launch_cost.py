"""CPU-side cost of launching one graph-replayed training step vs the GPU time of the step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
tr = bench.pkg("trainers").SegmentationTrainer("segmentation_supervised", "enet-naive-upsampling", "x",
    config={"input_shape": [None, None, 3], "optimizer": {"name": "adam", "learning_rate": 5e-4}, "ignore_classes": [0],
            "random_seed": 14874, "num_labeled_per_batch": 16}, num_classes=24)
m = tr.model
im, lab, w = bench.synthetic_batch(16, 512, 512, 24, 1, torch.device("cuda"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m.set_batch([im, lab, w, np.zeros((16, 1), dtype=np.int32)])
    for _ in range(5): m.train_step_async()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): m.train_step_async()
    t1 = time.perf_counter()
    s.synchronize()
    t2 = time.perf_counter()
print("cpu launch per step %.3f ms, wall per step %.3f ms" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
