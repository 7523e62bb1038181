"""Probe: gloo all-reduce of CUDA tensors from two ranks sharing one GPU (rehearsal of the data-parallel path)."""
import os, sys, time, faulthandler
import torch, torch.distributed as dist
faulthandler.dump_traceback_later(60, exit=True)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
mode = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
g = torch.full((N,), float(rank + 1), device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.mul_(2.0)
    t0 = time.time()
    if mode == "sync":
        dist.all_reduce(g[100:])
        dist.all_reduce(g[:100])
    elif mode == "async1":
        w = dist.all_reduce(g[100:], async_op=True); w.wait()
        w = dist.all_reduce(g[:100], async_op=True); w.wait()
    elif mode == "async2":
        w1 = dist.all_reduce(g[100:], async_op=True)
        w2 = dist.all_reduce(g[:100], async_op=True)
        w1.wait(); w2.wait()
    s.synchronize()
print(mode, rank, g[0].item(), g[-1].item(), "%.3f s" % (time.time() - t0), flush=True)
dist.destroy_process_group()
