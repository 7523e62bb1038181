"""Headline benchmark: ENet (enet-naive-upsampling, 24 classes) training images/s at 512x512, per-GPU batch 16, fp32,
synthetic data resident in HBM.  One step = dropout tables + forward + fused loss/gradient/confusion matrix + backward
+ (N>1: bucketed RCCL all-reduce) + Adam, replayed from HIP graphs.

    python bench.py --gpus N --steps K --warmup W
    (N>1: either under a launcher - python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... -
     or plain: with WORLD_SIZE unset the script starts the N ranks itself as child processes, launch_ranks())

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     : algorithmic bytes / measured duration of the dominant kernel family against the HBM peak
  cpu_baseline : the CPU oracle (PyTorch-CPU restatement of the Keras/TF graph, TF itself is unavailable offline)
                 timed on this host on a bounded sample (config 1: batch 4 at 512x512)
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def pkg(sub=None):
    return importlib.import_module("semantic-segmentation_amd" + ("." + sub if sub else ""))


def synthetic_batch(batch, h, w, nc, seed, device):
    """images ~ N(0,1); labels constant over 32x32 tiles; weights = w[labels] with w = [0,1,...,1]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    images = torch.randn(batch, h, w, 3, generator=g, dtype=torch.float32)
    coarse = torch.randint(0, nc, (batch, (h + 31) // 32, (w + 31) // 32), generator=g, dtype=torch.int32)
    labels = coarse.repeat_interleave(32, 1).repeat_interleave(32, 2)[:, :h, :w].contiguous()
    cw = torch.ones(nc)
    cw[0] = 0.0
    weights = cw[labels.long()].contiguous()
    return images.to(device), labels.to(device), weights.to(device)


def cpu_baseline(seconds_budget=20.0):
    from oracle import enet_oracle as E
    # threads = this job's CPU share: the GPU box exports it as OMP_NUM_THREADS (16 per GPU) while os.cpu_count() reports
    # the whole machine; without that variable every host core is used
    host_cores = os.cpu_count() or 1
    threads = min(host_cores, int(os.environ.get("OMP_NUM_THREADS", str(host_cores))))
    torch.set_num_threads(threads)
    name = "enet-naive-upsampling"
    w = E.init_weights(name, 24, dtype=np.float32)
    B = 4
    im, lab, pw = E.synthetic_batch(B, 512, 512)
    masks = E.make_dropout_masks(name, B, 1, dtype=np.float32)
    state = {"iterations": 0, "m": {}, "v": {}}
    times = []
    t_start = time.time()
    for i in range(2 + 12):
        t0 = time.time()
        E.train_step(name, w, im, lab, pw, masks, dtype=torch.float32, adam=dict(lr=5e-4), adam_state=state)
        dt = time.time() - t0
        if i >= 2:
            times.append(dt)
        if i >= 2 + 4 and time.time() - t_start > seconds_budget:
            break
    med = float(np.median(times))
    return {"value": B / med, "unit": "images/s", "cores": threads, "cpu_count": host_cores,
            "torch_threads": torch.get_num_threads(), "kind": "port",
            "sample": "{} timed fwd+bwd+Adam steps (after 2 warm-ups) of enet-naive-upsampling, batch 4 at 512x512, fp32, "
                      "PyTorch-CPU restatement of the Keras/TF graph (TensorFlow unavailable offline); median step {:.3f} s".format(
                          len(times), med)}


F32_MFMA_PEAK_TFLOPS = 157.3   # dense fp32 matrix peak (MI355X_MICROARCH.md, MFMA table)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 matrix peak (same table; the headline figures with 2:1 sparsity are twice that)


def launch_ranks(n):
    """Runs this script under `python -m torch.distributed.run --nproc-per-node n` (one rank per GPU, rendezvous on
    127.0.0.1 at a free port) as a child process, passes the ranks' stderr through, prints the one JSON line rank 0 wrote
    and returns the launcher's exit code.  Everything else the children print to stdout goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if line is None and out.startswith("{") and '"metric"' in out:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the ranks finished without a result line", file=sys.stderr)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--model", default="enet-naive-upsampling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "eval"],
                    help="eval: inference-phase forward + loss + confusion matrix (e.g. --size 1024 --batch 10), the matrices "
                         "of all ranks summed at the end")
    ap.add_argument("--dtype", default=None, choices=["f32", "bf16", "both"],
                    help="bf16 / both: also time the step with bfloat16 activation storage and report it in the \"bf16\" field "
                         "(value / dtype stay the float32 measurement).  Default: both on one GPU, f32 on several")
    ap.add_argument("--no-overlap-wgrad", action="store_true", help="keep the weight gradients on the main stream")
    ap.add_argument("--eager", action="store_true", help="launch kernels directly instead of replaying HIP graphs")
    ap.add_argument("--host-inputs", nargs="?", const="f32", default=None, choices=["f32", "u8"],
                    help="side measurement (the headline value keeps its inputs resident in HBM).  f32: every timed step "
                         "first stages the float32 batch from pageable numpy arrays, as train_on_batch does for a "
                         "generator's output.  u8: uint8 crops and masks go through pinned double buffers and the "
                         "device pre-pass (generators.PinnedBatchFeeder), the next batch in flight during the step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process starts the N ranks itself, as fresh children, before anything
        # here has touched a GPU, and relays rank 0's line and the launcher's exit code
        raise SystemExit(launch_ranks(args.gpus))
    import torch.distributed as dist
    if os.environ.get("SSMI_BENCH_WATCHDOG"):      # dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SSMI_BENCH_WATCHDOG"]), exit=True)
    trainers = pkg("trainers")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus {} does not match WORLD_SIZE={} of the launcher".format(args.gpus, world))
    cfg = {"input_shape": [None, None, 3], "optimizer": {"name": "adam", "learning_rate": 5e-4}, "ignore_classes": [0],
           "random_seed": 14874, "num_labeled_per_batch": args.batch}
    eval_bf16 = args.mode == "eval" and args.dtype == "bf16"      # evaluation is one measurement: --dtype picks its storage type
    if eval_bf16:
        cfg["activation_dtype"] = "bf16"
    tr = trainers.SegmentationTrainer("segmentation_supervised", args.model, "bench", config=cfg, num_classes=24)
    rank, world = tr.rank, tr.world_size
    dev = torch.device(tr.device)
    t_begin = time.time()

    def note(msg):      # progress on stderr (a silent multi-rank run is indistinguishable from a hung one)
        if world > 1 or os.environ.get("SSMI_BENCH_VERBOSE"):
            print("[bench rank {} +{:.1f}s] {}".format(rank, time.time() - t_begin, msg), file=sys.stderr, flush=True)
    note("trainer ready on {}".format(dev))
    images, labels, weights = synthetic_batch(args.batch, args.size, args.size, 24, 14874 + rank, dev)
    nu = np.zeros((args.batch, 1), dtype=np.int32)
    stream = torch.cuda.Stream(device=dev)

    def timed_run(tr, tag):
        """W untimed steps, then exactly K timed ones between barrier + synchronize on both sides; max over ranks."""
        model = tr.model
        if args.eager:
            model.use_hip_graph = False
        if args.no_overlap_wgrad:
            model.graph.overlap_weight_gradients = False
        ev = None
        with torch.cuda.stream(stream):
            model.set_batch([images, labels, weights, nu])
            step = model.train_step_async if args.mode == "train" else model.eval_step_async
            for i in range(max(args.warmup, 1)):
                step()
                if i == 0:
                    stream.synchronize()
                    note("{}: first step done".format(tag))
            stream.synchronize()
            note("{}: warm-up done".format(tag))
            model.reset_metrics()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            host = feeder = None
            if args.host_inputs == "f32":
                host = [t.cpu().numpy() for t in (images, labels, weights)] + [nu]
            elif args.host_inputs == "u8":
                gen = pkg("generators")
                hrng = np.random.RandomState(4242 + rank)
                u8_images = hrng.randint(0, 256, size=(args.batch, args.size, args.size, 3)).astype(np.uint8)
                u8_masks = hrng.randint(0, 24, size=(args.batch, args.size, args.size)).astype(np.uint8)
                enc = gen.DeviceBatchEncoder({c: c for c in range(24)}, [0.0] + [1.0] * 23, [0.48, 0.45, 0.40], [0.23, 0.22, 0.23],
                                             device=dev)
                feeder = gen.PinnedBatchFeeder(enc, args.batch, args.size, args.size)
                feeder.submit(u8_images, u8_masks)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                if host is not None:
                    model.set_batch(host)
                elif feeder is not None:
                    feeder.submit(u8_images, u8_masks)      # the next batch travels while this one trains
                    feeder.feed(model)
                step()
            if args.mode == "eval":
                ev = tr.reduce_evaluation(0.0, args.batch * args.steps)   # sums the int64 matrices over the ranks
            stream.synchronize()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, float(model._loss_buf.item()), ev

    model = tr.model
    elapsed, loss, ev = timed_run(tr, "f32")
    ms = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed

    # Data-parallel runs: where a step's time goes, from HIP events on the step's stream over a few extra (untimed) steps -
    # each backward segment, the time the stream then waits for each gradient bucket, the optimizer - maximum over ranks.
    dp = None
    if tr.reducer is not None and args.mode == "train":
        red = tr.reducer
        b0, n0 = red.bytes_reduced, red.buckets_started
        k = min(args.steps, 10)
        with torch.cuda.stream(stream):
            model.segment_timing(True)
            for _ in range(k):
                model.train_step_async()
            times = model.segment_times_ms()
            model.segment_timing(False)
        if times is not None:       # eager launches (--eager) carry no marks
            flat = torch.tensor(times["segments"] + times["bucket_waits"] + [times["optimizer"]], dtype=torch.float64, device=dev)
            dist.all_reduce(flat, op=dist.ReduceOp.MAX)
            flat = [round(v, 3) for v in flat.tolist()]
            ns = len(times["segments"])
            times = {"segments_ms": flat[:ns], "bucket_wait_ms": flat[ns:-1], "optimizer_ms": flat[-1]}
        dp = dict({"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(),
                   "buckets_per_step": (red.buckets_started - n0) // k, "bytes_reduced_per_step": (red.bytes_reduced - b0) // k,
                   "bucket_bytes": [(hi - lo) * 4 for lo, hi in model.bucket_ranges()],
                   "note": "segments_ms[0] = dropout tables + forward + loss + backward of the blocks behind the first cut; "
                           "bucket_wait_ms[j] = time the stream waited for bucket j after the last backward segment "
                           "(0 when the all-reduce was hidden behind backward work); maximum over ranks, {} steps".format(k)},
                  **(times or {}))

    if args.mode == "eval":
        if rank == 0:
            cm = ev["confusion_matrix"]
            print(json.dumps({"metric": "eval images/sec @{0}x{0} ENet".format(args.size), "value": round(value, 2),
                              "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "bf16 activation storage" if eval_bf16 else "f32", "data": "synthetic",
                              "config": {"workload": "{} 24 classes {}x{} per-GPU batch {} inference + loss + confusion matrix, "
                                                     "matrices summed over ranks".format(args.model, args.size, args.size, args.batch),
                                         "parallelism": "dp{}".format(world), "pixels_counted": int(cm.sum()),
                                         "miou": round(ev["miou"], 6), "accuracy": round(ev["accuracy"], 6)}}), flush=True)
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            dist.destroy_process_group()
        return

    roof = None
    if rank == 0 and not args.no_roofline:
        roof = measure_roofline(model, stream, args.model)

    # Second measurement, same step with bfloat16 activation storage (float32 arithmetic, parameters and optimizer state).
    # The headline value above stays the float32 one.
    bf16 = None
    want_bf16 = args.dtype in ("bf16", "both") if args.dtype else world == 1
    if want_bf16:
        tr.model.close()
        del model
        tr16 = trainers.SegmentationTrainer("segmentation_supervised", args.model, "bench", num_classes=24,
                                            config=dict(cfg, activation_dtype="bf16"))
        el16, loss16, _ = timed_run(tr16, "bf16")
        bf16 = {"value": round(args.batch * world * args.steps / el16, 2), "unit": "images/s",
                "ms_per_step": round(el16 / args.steps * 1e3, 3), "final_loss": round(loss16, 5),
                "dtype": "bf16 activation and activation-gradient storage; layers with >= 16 channels multiply bf16 operands on the bf16 "
                         "matrix instructions (f32 accumulation), the others in f32; statistics, parameters, Adam f32",
                "roofline": measure_roofline(tr16.model, stream, args.model) if rank == 0 and not args.no_roofline else None}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        family = "ENet" if args.model.startswith("enet") else args.model
        out = {"metric": "train images/sec @{0}x{0} {1}".format(args.size, family), "value": round(value, 2), "unit": "images/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "{} 24 classes {}x{} per-GPU batch {} fp32 train step (fwd+loss+bwd+Adam), global batch {}".format(
                   args.model, args.size, args.size, args.batch, args.batch * world),
                   "parallelism": "dp{}".format(world), "final_loss": round(loss, 5),
                   "inputs": {None: "resident in HBM", "f32": "float32 host arrays staged every step (PCIe-inclusive)",
                              "u8": "uint8 host arrays through pinned double buffers + device pre-pass every step "
                                    "(PCIe-inclusive, overlapped)"}[args.host_inputs]},
               "roofline": roof, "cpu_baseline": cpu, "bf16": bf16, "data_parallel": dp}
        print(json.dumps(out), flush=True)
    if world > 1:
        # rank 0 measured the roofline alone; the others wait here so that no rank tears the group down under it
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()


def measure_roofline(model, stream, model_name, steps=3):
    """Eager instrumented steps (same buffers and kernels as the graph replay).  The library brackets every convolution
    kernel launch (forward and data gradient) with HIP events on the launch stream and records the kernel variant with
    the algorithmic bytes / flops of that launch (ssmi_profile_*).  The dominant kernel is the variant with the largest
    summed duration per step; its roofline entry is  sum(algorithmic work of its launches) / sum(their durations),
    i.e. average work per launch over average launch duration (net of the bracket's own cost, measured around an empty
    kernel in the same run, so that it is comparable with a kernel trace's per-kernel duration), against the HBM peak when its arithmetic intensity is
    below the fp32-matrix ridge (157.3 TFLOP/s / 8 TB/s = 19.7 FLOP/B) and against the fp32 MFMA peak otherwise."""
    import ctypes as C
    L = pkg("_lib").lib()
    model.use_hip_graph = False
    saved_sync, model.grad_sync = model.grad_sync, None     # rank 0 measures alone: no collectives in these steps
    with torch.cuda.stream(stream):
        model.train_step_async()
        stream.synchronize()
        L.profile_begin()
        for _ in range(steps):
            model.train_step_async()
        stream.synchronize()
        cnt = C.c_int32()
        L.profile_end(C.byref(cnt))
    model.use_hip_graph = True
    model.grad_sync = saved_sync
    overhead_ms = float(L.profile_bracket_overhead_ms())      # what the event bracket adds around an empty kernel
    records = []
    name = C.create_string_buffer(128)
    ms, by, fl = C.c_float(), C.c_double(), C.c_double()
    for i in range(cnt.value):
        L.profile_get(i, name, 128, C.byref(ms), C.byref(by), C.byref(fl))
        records.append((name.value.decode(), max(ms.value - overhead_ms, 1e-4), by.value, fl.value))
    agg = {}
    for kernel, ms_, by, fl in records:
        r = agg.setdefault(kernel, [0.0, 0.0, 0.0, 0])
        r[0] += ms_
        r[1] += by
        r[2] += fl
        r[3] += 1
    # the bf16 patch weight gradient is recorded too, but it runs on a side stream beside the main chain: its bracket also
    # times the kernels it shares the chip with, so it is listed (wgrad_patch_ms_per_step_overlapped) but not priced here
    main = {k: v for k, v in agg.items() if "wgrad" not in k}
    kernel, (ms, by, fl, n) = max(main.items(), key=lambda kv: kv[1][0])
    conv_ms = sum(v[0] for k, v in agg.items() if "wgrad" not in k) / steps      # (the bf16 patch weight gradient is recorded too)
    ai = fl / by
    gbs = by / (ms * 1e-3) / 1e9
    tfs = fl / (ms * 1e-3) / 1e12
    # HBM bytes per launch from the committed counter passes (scripts/clean_evidence.sh: FETCH_SIZE / WRITE_SIZE with one kernel
    # on the chip at a time); the trace names carry the storage type as first template argument, the records do not
    traffic = traffic_source = None
    import re
    want_bf16 = getattr(model.graph.store, "act_dtype", torch.float32) == torch.bfloat16
    base = kernel.replace(" bf16 mfma", "")
    tag = "" if model_name == "enet-naive-upsampling" else "_" + model_name
    if want_bf16:
        files = ("r03_traffic{}_bf16.json".format(tag),) + (("r02_traffic_bf16.json",) if not tag else ())
    else:
        files = ("r03_traffic{}.json".format(tag),) + (("r02_traffic.json", "r01_traffic.json") if not tag else ())
    for fname in files:
        tpath = os.path.join(ROOT, "profiles", fname)
        if traffic is None and os.path.exists(tpath):
            with open(tpath) as f:
                raw = json.load(f)
            # trace names: storage type first, then the record's arguments, then (newer builds) the normalise-on-load flag;
            # kernels that exist for one storage type only carry no type and are recorded under the trace's own name
            for k, v in raw.items():
                m = re.match(r"^(\w+)<(float|__bf16|bf16), (.*?)(, false|, true|, 0)?>$", k)
                if k == base or (m and (m.group(2) != "float") == want_bf16 and
                                 base in ("{}<{}>".format(m.group(1), m.group(3)), "{}<{}{}>".format(m.group(1), m.group(3), m.group(4) or ""))):
                    traffic = v.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/" + fname
    common = {"kernel": "ssmi::" + kernel, "launches_per_step": n // steps, "avg_launch_us": round(ms / n * 1e3, 2),
              "algorithmic_bytes_per_launch": round(by / n), "algorithmic_flops_per_launch": round(fl / n),
              "arithmetic_intensity_flop_per_byte": round(ai, 2), "achieved_gbs": round(gbs, 1), "achieved_tflops": round(tfs, 2),
              "kernel_ms_per_step": round(ms / steps, 3), "all_conv_fwd_bwd_data_ms_per_step": round(conv_ms, 3),
              "event_bracket_overhead_us": round(overhead_ms * 1e3, 2),
              "wgrad_patch_ms_per_step_overlapped": round(sum(v[0] for k, v in agg.items() if "wgrad" in k) / steps, 3), "traffic": traffic,
              # HBM bytes per launch come from committed rocprofv3 counter passes of this kernel, not from this run
              "traffic_source": traffic_source}
    # a variant that multiplies on the bf16 matrix instructions is priced against THEIR dense peak (ridge 312 FLOP/B)
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if (kernel.endswith("bf16 mfma") or "_bf16" in kernel) else F32_MFMA_PEAK_TFLOPS
    if ai < mfma_peak * 1e3 / HBM_PEAK_GBS:
        return dict({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4)}, **common)
    return dict({"bound": "mfma", "achieved": round(tfs, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                 "frac": round(tfs / mfma_peak, 4)}, **common)


if __name__ == "__main__":
    main()
