"""How long a replayed training step runs after its main chain has ended (the weight-gradient side chain finishing alone):
a device-timestamp kernel (ssmi_device_timestamp) is captured in front of the join, two more bracket the graph launch.
tail_probe.py [model] [f32|bf16] [batch]

CAUTION: the captured timestamp node is one more dependent of the main chain's last kernel, and that alone changes how the HIP
graph executor schedules the step: with the probe the ENet fp32 step takes 11.2 ms instead of 8.89 ms - the main chain ends
after 7.96 ms (its stand-alone time) and ALL weight gradients run behind it.  The numbers describe the probed graph, not the
production one; the script is kept because it shows how sensitive the executor's schedule is to the graph's shape."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
model_name = sys.argv[1] if len(sys.argv) > 1 else "enet-naive-upsampling"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
eng = bench.pkg("engine")
L = bench.pkg("_lib").lib()
stamps = torch.zeros(4, dtype=torch.int64, device="cuda")
orig_join = eng.RunCfg.join


def join(self):
    self.flush()
    self._launch_deferred(0)
    if self.side is not None and self.training:
        L.device_timestamp(stamps.data_ptr() + 8, self.main.cuda_stream)
    orig_join(self)


eng.RunCfg.join = join
cfg = {"input_shape": [None, None, 3], "optimizer": {"name": "adam", "learning_rate": 5e-4}, "ignore_classes": [0],
       "random_seed": 14874, "num_labeled_per_batch": batch}
if dtype == "bf16":
    cfg["activation_dtype"] = "bf16"
tr = bench.pkg("trainers").SegmentationTrainer("segmentation_supervised", model_name, "x", config=cfg, num_classes=24)
m = tr.model
im, lab, w = bench.synthetic_batch(batch, 512, 512, 24, 1, torch.device("cuda"))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    m.set_batch([im, lab, w, np.zeros((batch, 1), dtype=np.int32)])
    for _ in range(5):
        m.train_step_async()
    s.synchronize()
    rows = []
    khz = float(L.device_timestamp_khz())
    for _ in range(10):
        L.device_timestamp(stamps.data_ptr(), s.cuda_stream)
        m.train_step_async()
        L.device_timestamp(stamps.data_ptr() + 16, s.cuda_stream)
        s.synchronize()
        t = stamps.cpu().numpy()
        rows.append(((t[1] - t[0]) / khz, (t[2] - t[1]) / khz))
    # the same steps back to back, as a training loop issues them
    L.device_timestamp(stamps.data_ptr(), s.cuda_stream)
    for _ in range(10):
        m.train_step_async()
    L.device_timestamp(stamps.data_ptr() + 16, s.cuda_stream)
    s.synchronize()
    t = stamps.cpu().numpy()
    print("10 steps back to back: %.3f ms per step; the last main chain ended %.3f ms before the end" %
          ((t[2] - t[0]) / khz / 10, (t[2] - t[1]) / khz))
r = np.median(np.array(rows), axis=0)
print("%s %s: main chain ends %.3f ms after the launch, the step %.3f ms later" % (model_name, dtype, r[0], r[1]))
