"""Properties that hold at the benchmark's full sizes (ENet, 24 classes, 512x512 batch 16 training; 1024x1024 inference),
where the CPU oracle would take minutes per step.  All of them are exact (bit for bit / integer equality):

  * replay determinism: the step launched eagerly and the same step replayed from captured HIP graphs (weight gradients
    overlapped on the side stream) give identical losses, gradients and updated weights;
  * linearity in the pixel weights: doubling every weight doubles the loss and every gradient exactly (a power-of-two
    scale commutes with every rounding in the backward pass) and leaves the BN statistics and the matrix untouched;
  * the confusion matrix accounts for every pixel whose label is not ignored, rows of ignored labels stay empty;
  * pooling indices (src/layers/pooling.py:34-50) equal the first maximum of their 2x2 window in scan order (recomputed with
    torch on the device from the block's own input), address the pooled value, and unpooling (both entry points) puts
    every pooled value back at exactly that position with zeros elsewhere;
  * inference is independent of the batch composition: an image predicted inside a batch of 4 and alone gives identical logits.
"""
import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E

pytestmark = pytest.mark.gpu
NC = 24


def _build(name, seed=3, dtype="f32"):
    models, losses, metrics, opt = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    mb = models.get_model(name, (None, None, 3), NC, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY, dtype=dtype)
    m = mb.model
    m.graph.store.init_values(seed)
    m.compile(optimizer=opt.Adam(lr=5e-4), loss={'loss': losses.dummy_loss, 'logits': lambda t, p: 0 * p},
              loss_weights={'loss': 1.0, 'logits': 0.0},
              metrics={'logits': [metrics.segmentation_accuracy(0, [0]), metrics.segmentation_mean_iou(NC, 0, [0]),
                                  metrics.segmentation_mean_per_class_accuracy(NC, 0, [0]),
                                  metrics.segmentation_confusion_matrix(NC, 0, [0])]})
    return m


def _snapshot(m):
    st = m.graph.store
    return st.theta.clone(), st.state.clone(), st.grad.clone()


def _restore(m, snap, opt_snap):
    st = m.graph.store
    st.theta.copy_(snap[0])
    st.state.copy_(snap[1])
    for k, t in m.optimizer.slot_tensors().items():
        t.copy_(opt_snap[k])
    m.optimizer.set_iterations(0)
    m.reset_metrics()


@pytest.mark.parametrize("name,B,dtype", [("enet-naive-upsampling", 16, "f32"), ("enet-max-unpooling", 16, "f32"), ("segnet", 4, "f32"),
                                          ("unet", 4, "f32"), ("enet-naive-upsampling", 16, "bf16"),
                                          ("enet-max-unpooling", 16, "bf16"), ("segnet", 4, "bf16")])
def test_full_size_training_step_properties(name, B, dtype):
    """bf16 storage: the same exact properties hold - rounding to bfloat16 is deterministic and commutes with a power-of-two
    scale, and the pooling indices are first maxima of the ROUNDED tensor the kernel pooled."""
    H, W = 512, 512
    m = _build(name, dtype=dtype)
    cw = np.ones(NC, dtype=np.float32)
    cw[0] = 0
    cw[1:] = 0.5 + np.arange(1, NC) / 23.0
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=77, block=32, class_weights=cw)
    x = [torch.from_numpy(images).cuda(), torch.from_numpy(labels).cuda(), torch.from_numpy(pw).cuda(), np.zeros((B, 1), np.int32)]
    snap = _snapshot(m)
    opt_snap = {k: t.clone() for k, t in m.optimizer.slot_tensors().items()}
    stream = torch.cuda.Stream()
    runs = {}
    for mode in ("eager", "graph", "graph2", "double"):
        _restore(m, snap, opt_snap)
        m.use_hip_graph = mode != "eager"
        xx = list(x)
        if mode == "double":
            xx[2] = x[2] * 2.0
        with torch.cuda.stream(stream):
            outs = m.train_on_batch(xx, None)
        torch.cuda.synchronize()
        st = m.graph.store
        runs[mode] = dict(loss=outs[0], cm=outs[6].copy(), grad=st.grad.clone(), theta=st.theta.clone(), state=st.state.clone(),
                          dlogits=m.graph.dlogits.clone())
    a = runs["eager"]
    assert np.isfinite(a["loss"]) and a["loss"] > 0
    for other in ("graph", "graph2"):
        b = runs[other]
        assert a["loss"] == b["loss"], "loss differs between eager launch and graph replay"
        assert torch.equal(a["grad"], b["grad"]) and torch.equal(a["theta"], b["theta"]) and torch.equal(a["state"], b["state"])
        assert np.array_equal(a["cm"], b["cm"])
    d = runs["double"]
    assert d["loss"] == np.float32(a["loss"]) * np.float32(2.0)
    assert torch.equal(d["dlogits"], a["dlogits"] * 2.0)
    assert torch.equal(d["grad"], a["grad"] * 2.0), "gradients are not linear in the pixel weights"
    assert torch.equal(d["state"], a["state"]) and np.array_equal(d["cm"], a["cm"])
    # confusion matrix bookkeeping
    cm = a["cm"]
    assert cm.dtype == np.int64 and int(cm.sum()) == int((labels != 0).sum()) and int(cm[0].sum()) == 0
    assert np.array_equal(cm.sum(axis=1), np.bincount(labels.reshape(-1), minlength=NC) * (np.arange(NC) != 0))
    if name == "enet-max-unpooling":
        g = m.graph
        L = pkg("_lib").lib()
        for bname in ("en_bn_1.0", "en_bn_2.0"):
            blk = g.by_name[bname]
            xin, pooled = blk.x, blk.pooled          # [B, 2h, 2w, C] block input, [B, h, w, C] pooled skip branch
            idx = blk.idx.long()                     # per-image flat (y*W + x)*C + c into the block's input
            n, h2, w2, c = idx.shape
            hin, win = 2 * h2, 2 * w2
            assert tuple(xin.shape) == (n, hin, win, c) and int(idx.max()) < hin * win * c and int(idx.min()) >= 0
            # the index addresses the pooled value ...
            assert torch.equal(torch.gather(xin.reshape(n, -1), 1, idx.reshape(n, -1)).reshape(idx.shape), pooled)
            # ... which is the window's maximum, and no earlier window element (scan order h then w) equals it
            c0, c1, c2, c3 = xin[:, 0::2, 0::2], xin[:, 0::2, 1::2], xin[:, 1::2, 0::2], xin[:, 1::2, 1::2]
            mx = torch.maximum(torch.maximum(c0, c1), torch.maximum(c2, c3))
            assert torch.equal(mx, pooled)
            k = torch.where(c0 == mx, 0, torch.where(c1 == mx, 1, torch.where(c2 == mx, 2, 3)))
            ar_c = torch.arange(c, device="cuda").view(1, 1, 1, c)
            ar_y = torch.arange(h2, device="cuda").view(1, h2, 1, 1)
            ar_x = torch.arange(w2, device="cuda").view(1, 1, w2, 1)
            expect = ((2 * ar_y + k // 2) * win + 2 * ar_x + k % 2) * c + ar_c
            assert torch.equal(idx, expect), "argmax indices of {} are not the first maximum of their window".format(bname)
            # unpooling puts every pooled value back at exactly that position and zeros everywhere else
            # (the stand-alone unpooling entry points are float32; a bf16 network's pooled values are widened for them)
            pooled = pooled.float().contiguous()
            up = torch.full((n, hin, win, c), 7.0, device="cuda")
            L.maxunpool_local_fwd(pooled.data_ptr(), blk.idx.data_ptr(), n, h2, w2, c, up.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
            up2 = torch.full((n, hin, win, c), 7.0, device="cuda")
            L.maxunpool_fwd(pooled.data_ptr(), blk.idx.data_ptr(), n, h2, w2, c, up2.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(up, up2)
            assert torch.equal(torch.gather(up.reshape(n, -1), 1, idx.reshape(n, -1)).reshape(idx.shape), pooled)
            assert int((up != 0).sum()) == int((pooled != 0).sum())
            del up, up2


def test_bf16_storage_against_float32_at_the_benchmark_size():
    """The same network (identical weights, batch, dropout draws) with float32 and with bfloat16 activation storage at
    512x512, batch 16.  The loss must agree within 2e-2 (the tolerance the bf16 path is held to; measured 5e-5).  Whole-
    network TENSORS cannot: this randomly initialised, training-phase network amplifies any perturbation several hundred
    times on the way to the logits - rounding nothing but the IMAGES to bfloat16 and running the float32 kernels moves the
    logits by 25 % (relative L2), one float32 ulp of input noise by 4e-5 (scripts/sensitivity_probe.py).  So the bound on
    the logits is set against that yardstick, measured in the same test: bfloat16 storage of every tensor on the path may
    move the logits at most three times as far as bfloat16 rounding of the input alone does in the float32 network
    (measured 1.8x).  Printed: both distances, the fraction of labeled pixels classified alike."""
    name, B, H, W = "enet-naive-upsampling", 16, 512, 512
    cw = np.ones(NC, dtype=np.float32)
    cw[0] = 0
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=78, block=32, class_weights=cw)
    xi = torch.from_numpy(images).cuda()
    res = {}
    for dtype in ("f32", "bf16"):
        m = _build(name, seed=5, dtype=dtype)
        m.keep_logits = True

        def fwd(inp):
            lg = m.graph.forward(inp.contiguous(), training=True, update_moving=False, use_dropout=False)
            torch.cuda.synchronize()
            return lg.float().clone()
        r = dict(fwd=fwd(xi))
        if dtype == "f32":
            r["fwd_rounded_input"] = fwd(xi.bfloat16().float())
        x = [xi, torch.from_numpy(labels).cuda(), torch.from_numpy(pw).cuda(), np.zeros((B, 1), np.int32)]
        with torch.cuda.stream(torch.cuda.Stream()):
            outs = m.train_on_batch(x, None)
        torch.cuda.synchronize()
        r.update(loss=outs[0], cm=outs[6].copy())
        res[dtype] = r
        m.close()
        del m
    a, b = res["f32"], res["bf16"]

    def rel(u, v):
        return float((u - v).double().norm() / u.double().norm())
    rel_loss = abs(a["loss"] - b["loss"]) / abs(a["loss"])
    d_input, d_storage = rel(a["fwd"], a["fwd_rounded_input"]), rel(a["fwd"], b["fwd"])
    keep = torch.from_numpy(labels != 0).cuda()
    agree = float((a["fwd"].argmax(-1) == b["fwd"].argmax(-1))[keep].float().mean())
    agree_in = float((a["fwd"].argmax(-1) == a["fwd_rounded_input"].argmax(-1))[keep].float().mean())
    print("bf16 vs f32 storage at 512x512 batch 16: loss {:.6f} vs {:.6f} (rel {:.2e}); logits rel L2: bf16 storage {:.3e}, "
          "float32 kernels on bf16-rounded images {:.3e} (ratio {:.2f}); same class on {:.2%} / {:.2%} of labeled pixels".format(
              a["loss"], b["loss"], rel_loss, d_storage, d_input, d_storage / d_input, agree, agree_in))
    assert rel_loss <= 2e-2
    assert d_storage <= 3.0 * d_input
    assert int(a["cm"].sum()) == int(b["cm"].sum()) == int((labels != 0).sum())


@pytest.mark.parametrize("B", [4, 10])
def test_full_size_inference_is_batch_independent(B):
    """B = 10 is the validation batch of the reference's configuration (configs/segmentation/config-segmentation-supervised.json:40-42)."""
    m = _build("enet-naive-upsampling", seed=9)
    H, W = 1024, 1024
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=5, block=64)
    xb = torch.from_numpy(images).cuda()
    full = m.predict_on_batch_device(xb).clone()
    torch.cuda.synchronize()
    assert tuple(full.shape) == (B, H, W, NC) and bool(torch.isfinite(full).all())
    for i in (0, B - 1):
        one = m.predict_on_batch_device(xb[i:i + 1]).clone()
        assert torch.equal(one[0], full[i]), "image {} differs between batch sizes".format(i)
    # evaluation: the matrix over the batch equals the matrix recomputed from the logits, and the loss is the mean over pixels.
    # The evaluation computes the last layer inside the loss kernel, whose sums run in another order than the stored-logits
    # route's: the two agree to an ulp (random weights at this size give logits of 6e4, an ulp of 0.008), and a pixel whose two
    # best classes are that close may go either way - so the matrix is checked against the evaluation's OWN logits, and those
    # against the prediction's.
    m.reset_metrics()
    m.keep_logits = True
    outs = m.test_on_batch([xb, labels, pw, np.zeros((B, 1), np.int32)], None)
    torch.cuda.synchronize()
    ev = m.graph.logits.float()
    assert float((ev - full).abs().max()) <= 4e-7 * float(full.abs().max())
    full = ev.clone()
    pred = full.argmax(dim=-1).cpu().numpy()
    keep = labels != 0
    cm = np.zeros((NC, NC), np.int64)
    np.add.at(cm, (labels[keep], pred[keep]), 1)
    # ties in the argmax are resolved towards the lower class on both sides
    assert np.array_equal(outs[6], cm)
    lse = torch.logsumexp(full.double(), dim=-1).cpu().numpy()
    picked = torch.gather(full.double(), 3, torch.from_numpy(labels).long().cuda().unsqueeze(-1)).squeeze(-1).cpu().numpy()
    ref = float((pw.astype(np.float64) * (lse - picked)).mean())
    assert abs(outs[0] - ref) <= 1e-5 * abs(ref)
