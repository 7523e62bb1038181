"""N training steps of one model in ONE activation dtype and nothing else (a clean target for rocprofv3: no second dtype in
the same process, no roofline brackets, no CPU baseline): step_only.py [model] [f32|bf16] [batch] [steps]
SSMI_NO_OVERLAP=1: weight gradients on the main stream (one kernel on the chip at a time); SSMI_EAGER=1: direct launches
instead of graph replay (counter passes)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
model_name = sys.argv[1] if len(sys.argv) > 1 else "enet-naive-upsampling"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
cfg = {"input_shape": [None, None, 3], "optimizer": {"name": "adam", "learning_rate": 5e-4}, "ignore_classes": [0],
       "random_seed": 14874, "num_labeled_per_batch": batch}
if dtype == "bf16":
    cfg["activation_dtype"] = "bf16"
tr = bench.pkg("trainers").SegmentationTrainer("segmentation_supervised", model_name, "x", config=cfg, num_classes=24)
m = tr.model
if os.environ.get("SSMI_NO_OVERLAP"):
    m.graph.overlap_weight_gradients = False
if os.environ.get("SSMI_EAGER"):
    m.use_hip_graph = False
im, lab, w = bench.synthetic_batch(batch, 512, 512, 24, 1, torch.device("cuda"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m.set_batch([im, lab, w, np.zeros((batch, 1), dtype=np.int32)])
    for _ in range(steps):
        m.train_step_async()
    s.synchronize()
print("done", model_name, dtype, batch, steps)
