"""World-size-2 data-parallel checks on CPU (gloo): the bucketed reducer + 1/world scaling reproduce the global-batch
gradient of equally sized shards, the int64 confusion-matrix sum is exact, and sharding covers the batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import pkg, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from conftest import pkg as _pkg
    from oracle import enet_oracle as E
    trainers = _pkg("trainers")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    red = trainers.GradientBucketReducer()
    name, nc, GB, H, W = "enet-naive-upsampling", 6, 4, 32, 32
    w0 = E.init_weights(name, nc, seed=1, perturb=True)
    images, labels, pw = E.synthetic_batch(GB, H, W, nc, seed=2, block=8)
    lo, hi = trainers.shard_range(GB, world, rank)
    ref = E.train_step(name, w0, images[lo:hi], labels[lo:hi], pw[lo:hi], None)
    keys = E.trainable_keys(name, nc)
    flat = torch.cat([torch.from_numpy(ref["grads"][k].reshape(-1)) for k in keys])
    n = flat.numel()
    cut = n // 3
    # two buckets in backward order, then the optimizer-side 1/world scaling
    red.start(flat, (cut, n))
    red.start(flat, (0, cut))
    red.finish()
    flat /= red.world_size
    cm = torch.from_numpy(ref["cm"].copy())
    red.sum_int64_(cm)
    # sharded evaluation: matrix and mean loss over all shards
    ecm, eloss, eimg = trainers.all_reduce_evaluation(torch.from_numpy(ref["cm"].copy()), ref["loss"] * (hi - lo), hi - lo)
    t = torch.full((3,), float(rank))
    red.broadcast_([t], src=0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "r0.npz"), flat=flat.numpy(), cm=cm.numpy(), bytes=red.bytes_reduced, t=t.numpy(),
                 ecm=ecm, eloss=eloss, eimg=eimg)
    else:
        np.savez(os.path.join(out_dir, "r1.npz"), flat=flat.numpy(), cm=cm.numpy(), t=t.numpy(), ecm=ecm, eloss=eloss, eimg=eimg)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_average_and_cm_sum(tmp_path):
    from oracle import enet_oracle as E
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["cm"], r1["cm"])
    assert np.array_equal(r1["t"], np.zeros(3))
    name, nc, GB, H, W = "enet-naive-upsampling", 6, 4, 32, 32
    w0 = E.init_weights(name, nc, seed=1, perturb=True)
    images, labels, pw = E.synthetic_batch(GB, H, W, nc, seed=2, block=8)
    keys = E.trainable_keys(name, nc)
    # per-replica BatchNorm: the reference quantity is the mean of the two shard gradients
    g = [E.train_step(name, w0, images[a:b], labels[a:b], pw[a:b], None) for a, b in ((0, 2), (2, 4))]
    exp = torch.cat([torch.from_numpy(0.5 * (g[0]["grads"][k] + g[1]["grads"][k]).reshape(-1)) for k in keys]).numpy()
    assert np.allclose(r0["flat"], exp, rtol=1e-12, atol=1e-15)
    assert np.array_equal(r0["cm"], g[0]["cm"] + g[1]["cm"])
    assert int(r0["bytes"]) == exp.size * 8
    # evaluation reduce: identical on both ranks, exact matrix, image-weighted mean loss
    assert np.array_equal(r0["ecm"], r1["ecm"]) and np.array_equal(r0["ecm"], g[0]["cm"] + g[1]["cm"])
    assert int(r0["eimg"]) == 4 and float(r0["eloss"]) == float(r1["eloss"])
    assert abs(float(r0["eloss"]) - 0.5 * (g[0]["loss"] + g[1]["loss"])) < 1e-12


def test_shard_range():
    trainers = pkg("trainers")
    cover = []
    for r in range(8):
        lo, hi = trainers.shard_range(128, 8, r)
        assert hi - lo == 16
        cover.extend(range(lo, hi))
    assert cover == list(range(128))
    with pytest.raises(ValueError):
        trainers.shard_range(10, 4, 0)


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_data_parallel_step():
    """bench.py with SSMI_FORCE_DP=1: a process group of one rank on the nccl (RCCL) backend drives the bucketed
    all-reduce between the captured graph segments.  Summing over one rank is the identity, so the loss after the same
    number of steps must equal the single-process run bit for bit, and stdout must hold exactly one JSON line (RCCL's
    banner must not reach it)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "2", "--size", "64",
           "--no-cpu-baseline", "--no-roofline"]
    outs = []
    for extra in ({}, {"SSMI_FORCE_DP": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547"}):
        env = dict(os.environ, **extra)
        env.pop("WORLD_SIZE", None)
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        outs.append(json.loads(lines[0]))
    assert outs[0]["config"]["final_loss"] == outs[1]["config"]["final_loss"]
    assert outs[1]["n_gpus"] == 1 and outs[1]["value"] > 0


@pytest.mark.gpu
def test_two_gloo_ranks_share_the_gpu_and_finish_the_data_parallel_step():
    """Two ranks (gloo, both on GPU 0) run bench.py's data-parallel step: graph segments, a gradient bucket after each, the
    optimizer behind the last one.  The buckets travel through host memory here (GradientBucketReducer's gloo path), so this
    checks the segment / bucket logic with two real ranks, not RCCL.  Both ranks start from rank 0's weights and see
    all-reduced gradients, so the step must finish, report two ranks and two buckets covering the whole gradient arena."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (SSMI_DP_CUTS=1: the two-segment ENet step - the default is one bucket - so that the cut, its join and the second graph run)
    env = dict(os.environ, SSMI_DIST_BACKEND="gloo", SSMI_BENCH_WATCHDOG="100", SSMI_DP_CUTS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "2", "--size", "64"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=140, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    dp = d["data_parallel"]
    assert d["n_gpus"] == 2 and dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and dp["buckets_per_step"] == 2
    assert dp["bytes_reduced_per_step"] == sum(dp["bucket_bytes"]) and len(dp["segments_ms"]) == 2
    assert np.isfinite(d["config"]["final_loss"]) and d["value"] > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how a plain command line invokes it): the script itself starts two
    fresh rank processes under torch.distributed.run and relays rank 0's single line and the exit code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSMI_DIST_BACKEND="gloo", SSMI_BENCH_WATCHDOG="100")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "2",
           "--size", "64"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=140, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["data_parallel"]["rccl_ranks"] == 2 and d["value"] > 0


def test_bench_launcher_relays_failure_of_the_ranks(tmp_path):
    """The self-launch path on CPU: with no GPU the ranks fail; the parent must return their non-zero exit code and print no
    result line (instead of hanging or reporting success)."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSMI_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--batch", "1",
                        "--size", "32", "--no-cpu-baseline", "--no-roofline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=200, text=True)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_c_abi_collectives_one_rank():
    """ssmi_comm_* (RCCL behind the C ABI) with a communicator of one rank: sums and the broadcast are the identity, the
    calls are stream-ordered with the kernels around them, the handle reports rank and size.  Run in a child process: the
    communicator belongs to a process of its own, not to the test runner."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch
from conftest import pkg
L = pkg("_lib").lib()
uid = C.create_string_buffer(128)
L.comm_unique_id(uid)
comm = C.c_void_p()
L.comm_init(uid, 0, 1, C.byref(comm))
assert L.comm_size(comm) == 1 and L.comm_rank(comm) == 0
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.arange(100000, dtype=torch.float32, device="cuda") * 0.5
    ref = g.clone()
    g.mul_(3.0)
    L.comm_allreduce_sum_f32(comm, g.data_ptr(), g.numel(), s.cuda_stream)
    g.add_(1.0)
    cm = torch.arange(24 * 24, dtype=torch.int64, device="cuda")
    L.comm_allreduce_sum_i64(comm, cm.data_ptr(), cm.numel(), s.cuda_stream)
    L.comm_broadcast_f32(comm, g.data_ptr(), g.numel(), 0, s.cuda_stream)
s.synchronize()
assert torch.equal(g, ref * 3.0 + 1.0) and torch.equal(cm.cpu(), torch.arange(24 * 24))
L.comm_destroy(comm)
print("comm ok")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, text=True)
    assert r.returncode == 0 and "comm ok" in r.stdout, r.stderr[-3000:]
