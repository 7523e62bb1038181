"""bfloat16 activation storage (the benchmark configuration names "ENet 512x512 batch 16, fp32 and bf16 training").

Activations and their gradients live in HBM as bfloat16; kernels, statistics, optimizer state and every multiply-add are
float32.  Three layers of checks:

  * operator level, bit-exact: a bfloat16 kernel must return round-to-nearest-even(float32 kernel result) when both read
    the same bfloat16-representable inputs - the arithmetic between load and store is the same code.  Weight gradients
    (float32 outputs) must be identical.  That ties every bfloat16 kernel to its float32 twin, which tests/test_ops_gpu.py
    ties to the fp64 oracle at rtol 1e-3;
  * step level against the fp64 CPU oracle, BLOCK BY BLOCK on the kernels' own tensors: every block's output, input
    gradient and parameter gradients within 2e-2 (relative L2 norm; 2e-2 is the tolerance this project states for bf16) of the oracle's
    evaluation of that block on the block input / output gradient the device stored, the oracle rounding to bfloat16 where
    the kernels store (oracle/enet_oracle.py block_step, storage="bf16") and following the PReLU / pooling branches the
    kernels took.  Whole-network tensors cannot carry that tolerance with this storage format, whoever computes them: a
    rounding that falls the other way (one bfloat16 ulp = 2^-8 relative) is a fresh perturbation at every one of the ~100
    stored tensors on the path, and the randomly initialised network amplifies perturbations about 40-fold from input to
    logits - the fp64 oracle with bfloat16 storage sits 14 % (relative L2) from the oracle with exact storage, and two
    evaluations with bfloat16 storage that differ in the last float32 bit sit as far from each other.  So the whole-network
    checks are: the loss (a mean over all pixels) within 2e-2 of the exact oracle; the kernels' logits no farther from the
    exact oracle than 1.5x the bfloat16-storage oracle's; pooling indices bit-exact against the oracle's pooling of the
    kernel's own (rounded) input, confusion matrix bit-exact against the argmax of the kernel's own logits;
  * training: the loss falls over 30 steps, HIP-graph replay equals eager launches bit for bit."""
import types

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import enet_oracle as E
from oracle import tf_keras_ops as K

pytestmark = pytest.mark.gpu
DEV = "cuda"
NC = 24


def _bf(a):
    """float64/32 numpy -> values representable in bfloat16 (as float32 numpy)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def _dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(dtype)


CONV_CASES = [
    # kh kw cin cout stride dil padding bias  N  H  W
    (3, 3, 3, 13, 2, 1, "same", True, 2, 16, 20),      # initial block (scalar path, 2-byte loads)
    (2, 2, 16, 16, 2, 1, "valid", False, 2, 12, 8),    # downsampling projection / pixel-shuffle data gradient
    (2, 2, 64, 32, 2, 1, "valid", False, 1, 8, 8),
    (1, 1, 64, 16, 1, 1, "valid", False, 2, 9, 7),     # LDS-weight kernel, 16x16 tiles, ragged pixels
    (1, 1, 16, 64, 1, 1, "valid", False, 1, 8, 8),
    (1, 1, 32, 128, 1, 1, "valid", False, 1, 8, 8),
    (1, 1, 128, 32, 1, 1, "valid", False, 1, 6, 6),
    (3, 3, 16, 16, 1, 1, "same", True, 2, 8, 10),
    (3, 3, 32, 32, 1, 1, "same", True, 1, 9, 9),
    (3, 3, 32, 32, 1, 16, "same", True, 1, 8, 8),
    (1, 5, 32, 32, 1, 1, "same", False, 1, 6, 11),
    (5, 1, 32, 32, 1, 1, "same", True, 1, 11, 6),
    (3, 3, 4, 4, 1, 1, "same", True, 2, 8, 8),
    (1, 1, 64, 4, 1, 1, "valid", False, 1, 8, 8),
    (3, 3, 5, 7, 2, 1, "same", True, 1, 7, 9),         # odd everything
    (3, 3, 64, 64, 1, 1, "same", True, 2, 9, 11),      # wide layers: tiled weight gradient
    (3, 3, 32, 128, 1, 1, "same", False, 1, 8, 8),
]


def _same_sums(a, b, what):
    """Equal up to the rounding of float32 sums taken in a different order (1e-5 of the tensor's largest entry)."""
    scale = float(a.abs().max())
    err = float((a - b).abs().max())
    assert err <= 1e-5 * scale + 1e-30, "{}: {} apart at scale {}".format(what, err, scale)


def _conv_pair(case, transpose=False):
    kh, kw, cin, cout, s, d, pad, bias, N, H, W = case
    eng = pkg("engine")
    rng = np.random.RandomState(sum((i + 1) * int(v) for i, v in enumerate(case) if not isinstance(v, str)))
    w = rng.standard_normal((kh, kw, cout, cin) if transpose else (kh, kw, cin, cout)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        store = eng.ParamStore(dt)
        cv = eng.Conv(store, "c", kh, kw, cin, cout, stride=s, dil=d, padding=pad, bias=bias, transpose=transpose)
        store.materialise(torch.device(DEV))
        store.views["c/kernel"].copy_(torch.from_numpy(w))
        if bias:
            store.views["c/bias"].copy_(torch.from_numpy(b))
        ho, wo = cv.plan(N, H, W)
        out[dt] = (store, cv, ho, wo)
    return out, rng


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}".format(*c[:7]))
def test_conv2d_bf16_equals_rounded_float32_kernel(case):
    kh, kw, cin, cout, s, d, pad, bias, N, H, W = case
    pair, rng = _conv_pair(case)
    ho, wo = pair[torch.float32][2:]
    x = _bf(rng.standard_normal((N, H, W, cin)))
    dy = _bf(rng.standard_normal((N, ho, wo, cout)))
    add = _bf(rng.standard_normal((N, H, W, cin)))
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for dt, (store, cv, _, _) in pair.items():
        xd, dyd, addd = _dev(x, dt), _dev(dy, dt), _dev(add, dt)
        y = torch.full((N, ho, wo, cout), 7.0, device=DEV, dtype=dt)
        cv.fwd(xd, y, st)
        dx = torch.full((N, H, W, cin), 7.0, device=DEV, dtype=dt)
        cv.bwd_data(dyd, dx, st)
        dx2 = torch.full((N, H, W, cin), 7.0, device=DEV, dtype=dt)
        cv.bwd_data(dyd, dx2, st, add=addd)
        ws = torch.empty(cv.ws_bytes // 4 + 8, device=DEV)
        cv.bwd_weight(xd, dyd, ws, st)
        torch.cuda.synchronize()
        res[dt] = dict(y=y, dx=dx, dx2=dx2, dw=store.gviews["c/kernel"].clone(),
                       db=store.gviews["c/bias"].clone() if bias else None)
    a, b = res[torch.float32], res[torch.bfloat16]
    for k in ("y", "dx", "dx2"):
        assert torch.equal(a[k].to(torch.bfloat16), b[k]), "{}: bf16 kernel != rounded float32 kernel".format(k)
    # The weight gradient sees the same bf16 operands in both runs: every product is exact in float32 and only the order
    # of the float32 additions differs (bf16 storage sums 16 pixels per matrix instruction, float32 storage 2 or 4).
    _same_sums(a["dw"], b["dw"], "weight gradient")
    if bias:
        _same_sums(a["db"], b["db"], "bias gradient")


@pytest.mark.parametrize("k,cin,cout,N,H,W", [(3, 16, 16, 2, 6, 5), (3, 4, 4, 1, 8, 8), (2, 16, 24, 2, 5, 7), (2, 16, 24, 1, 3, 40)])
def test_conv2d_transpose_bf16_equals_rounded_float32_kernel(k, cin, cout, N, H, W):
    case = (k, k, cin, cout, 2, 1, "same", True, N, H, W)
    pair, rng = _conv_pair(case, transpose=True)
    ho, wo = pair[torch.float32][2:]
    x = _bf(rng.standard_normal((N, H, W, cin)))
    dy = _bf(rng.standard_normal((N, ho, wo, cout)))
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for dt, (store, cv, _, _) in pair.items():
        y = torch.empty(N, ho, wo, cout, device=DEV, dtype=dt)
        cv.fwd(_dev(x, dt), y, st)
        dx = torch.empty(N, H, W, cin, device=DEV, dtype=dt)
        cv.bwd_data(_dev(dy, dt), dx, st)
        ws = torch.empty(cv.ws_bytes // 4 + 8, device=DEV)
        cv.bwd_weight(_dev(x, dt), _dev(dy, dt), ws, st)
        torch.cuda.synchronize()
        res[dt] = (y, dx, store.gviews["c/kernel"].clone(), store.gviews["c/bias"].clone())
    a, b = res[torch.float32], res[torch.bfloat16]
    assert torch.equal(a[0].to(torch.bfloat16), b[0]) and torch.equal(a[1].to(torch.bfloat16), b[1])
    _same_sums(a[2], b[2], "weight gradient")
    _same_sums(a[3], b[3], "bias gradient")


@pytest.mark.parametrize("c,npix,act", [(16, 1000, 1), (4, 333, 1), (128, 260, 1), (64, 77, 0), (24, 50, 2)])
def test_bn_act_bf16_equals_rounded_float32_kernel(c, npix, act):
    L = pkg("_lib").lib()
    rng = np.random.RandomState(c + npix)
    x = _bf(rng.standard_normal((npix, c)) * 2 + 0.5)
    dy = _bf(rng.standard_normal((npix, c)))
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    gamma, beta, alpha = f(rng.uniform(0.5, 1.5, c)), f(rng.uniform(-0.5, 0.5, c)), f(rng.uniform(0.05, 0.4, c))
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for dt, sfx in ((torch.float32, ""), (torch.bfloat16, "_bf16")):
        xd, dyd = _dev(x, dt), _dev(dy, dt)
        mean, invstd = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        ws = torch.empty(L.bn_workspace_bytes(npix, c) // 4 + 8, device=DEV)
        getattr(L, "bn_train_stats" + sfx)(xd.data_ptr(), npix, c, c, 1e-3, 0.1, 0, None, None, mean.data_ptr(), invstd.data_ptr(),
                                           ws.data_ptr(), ws.numel() * 4, st)
        y = torch.empty(npix, c, device=DEV, dtype=dt)
        getattr(L, "bn_act_fwd" + sfx)(xd.data_ptr(), c, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), act,
                                       alpha.data_ptr(), 0.1, y.data_ptr(), npix, c, st)
        dx = torch.empty(npix, c, device=DEV, dtype=dt)
        dg, db, da = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        getattr(L, "bn_act_bwd" + sfx)(dyd.data_ptr(), xd.data_ptr(), c, mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                                       beta.data_ptr(), act, alpha.data_ptr(), 0.1, dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                       da.data_ptr(), npix, c, ws.data_ptr(), ws.numel() * 4, st)
        torch.cuda.synchronize()
        res[dt] = (mean, invstd, y, dx, dg, db, da)
    a, b = res[torch.float32], res[torch.bfloat16]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2].to(torch.bfloat16), b[2]) and torch.equal(a[3].to(torch.bfloat16), b[3])
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]) and torch.equal(a[6], b[6])


@pytest.mark.parametrize("mode,c,c_other,h,w", [(0, 64, 64, 6, 10), (1, 64, 16, 8, 6), (2, 16, 16, 8, 12), (3, 64, 64, 4, 6)])
def test_merge_bf16_equals_rounded_float32_kernel(mode, c, c_other, h, w):
    lib_mod = pkg("_lib")
    L = lib_mod.lib()
    import ctypes as C
    n = 2
    rng = np.random.RandomState(mode * 31 + c)
    f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)
    main = _bf(rng.standard_normal((n, h, w, c)))
    dout = _bf(rng.standard_normal((n, h, w, c)))
    oshape = (n, h, w, c) if mode == 0 else ((n, h, w, c_other) if mode == 1 else (n, h // 2, w // 2, c))
    other = _bf(rng.standard_normal(oshape))
    p3 = [f(rng.standard_normal(c) * 0.1), f(rng.uniform(0.5, 2, c)), f(rng.uniform(0.5, 1.5, c)), f(rng.uniform(-0.5, 0.5, c))]
    po = [f(rng.standard_normal(c) * 0.1), f(rng.uniform(0.5, 2, c)), f(rng.uniform(0.5, 1.5, c)), f(rng.uniform(-0.5, 0.5, c))]
    alpha = f(rng.uniform(-0.2, 0.5, c))
    drop = f((rng.uniform(size=(n, c)) > 0.2) / 0.8)
    idx = None
    if mode == 3:      # indices a 2x2 pooling of random data would produce
        _, idx_np = K.max_pool_with_argmax_np(rng.standard_normal((n, h, w, c)))
        idx = torch.from_numpy(idx_np.astype(np.int32)).to(DEV)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for dt, code in ((torch.float32, lib_mod.F32), (torch.bfloat16, lib_mod.BF16)):
        md = lib_mod.MergeDesc()
        md.n, md.h, md.w, md.c, md.other_mode, md.c_other, md.dtype = n, h, w, c, mode, c_other, code
        bn3 = lib_mod.BnParams(*[t.data_ptr() for t in p3])
        bno = lib_mod.BnParams(*[t.data_ptr() for t in po])
        mainD, otherD, doutD = _dev(main, dt), _dev(other, dt), _dev(dout, dt)
        out = torch.empty(n, h, w, c, device=DEV, dtype=dt)
        L.merge_fwd(C.byref(md), mainD.data_ptr(), C.byref(bn3), drop.data_ptr(), otherD.data_ptr(), C.byref(bno),
                    None if idx is None else idx.data_ptr(), alpha.data_ptr(), out.data_ptr(), st)
        dmain, dsum = torch.empty_like(out), torch.empty_like(out)
        doth = torch.empty(oshape, device=DEV, dtype=dt) if mode >= 2 else None
        dg, db, da = torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV)
        ws = torch.empty(L.merge_workspace_bytes(C.byref(md)) // 4 + 8, device=DEV)
        L.merge_bwd(C.byref(md), doutD.data_ptr(), mainD.data_ptr(), C.byref(bn3), drop.data_ptr(), otherD.data_ptr(), C.byref(bno),
                    None if idx is None else idx.data_ptr(), alpha.data_ptr(), dmain.data_ptr(), dsum.data_ptr(),
                    None if doth is None else doth.data_ptr(), dg.data_ptr(), db.data_ptr(), da.data_ptr(), ws.data_ptr(),
                    ws.numel() * 4, st)
        torch.cuda.synchronize()
        res[dt] = (out, dsum, dg, db, da, dmain, doth)
    a, b = res[torch.float32], res[torch.bfloat16]
    assert torch.equal(a[0].to(torch.bfloat16), b[0]), "merge output"
    assert torch.equal(a[1].to(torch.bfloat16), b[1]), "dsum"
    assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]), "parameter gradients"
    # dmain / dother re-read the ROUNDED dsum in the bfloat16 build: one rounding apart, not bit-equal
    for k in (5, 6):
        if a[k] is not None:
            ref, got = a[k].double(), b[k].double()
            assert float((ref - got).abs().max()) <= 2.0 ** -7 * float(ref.abs().max()), "dmain / dother"


def test_pool_and_loss_bf16():
    L = pkg("_lib").lib()
    rng = np.random.RandomState(5)
    st = torch.cuda.current_stream().cuda_stream
    for (n, h, w, c) in ((2, 8, 12, 16), (1, 7, 9, 3), (2, 6, 6, 64)):
        x = _bf(rng.standard_normal((n, h, w, c)))
        x[0, :2, :2, 0] = 0.5      # a tie: the first element of the window wins
        ho, wo = (h + 1) // 2, (w + 1) // 2
        xd = _dev(x, torch.bfloat16)
        y = torch.empty(n, ho, wo, c, device=DEV, dtype=torch.bfloat16)
        idx = torch.empty(n, ho, wo, c, device=DEV, dtype=torch.int32)
        L.maxpool_argmax_fwd_bf16(xd.data_ptr(), n, h, w, c, y.data_ptr(), c, idx.data_ptr(), None, st)
        ref_y, ref_i = K.max_pool_with_argmax_np(x)
        assert np.array_equal(idx.cpu().numpy().astype(np.int64), ref_i) and np.array_equal(y.float().cpu().numpy(), ref_y)
        dy = _bf(rng.standard_normal((n, ho, wo, c)))
        dx = torch.empty(n, h, w, c, device=DEV, dtype=torch.bfloat16)
        L.maxpool_argmax_bwd_bf16(_dev(dy, torch.bfloat16).data_ptr(), c, idx.data_ptr(), n, h, w, c, dx.data_ptr(), 0, st)
        ref_dx = np.zeros((n, h * w * c), np.float32)
        np.add.at(ref_dx, (np.arange(n)[:, None], ref_i.reshape(n, -1)), dy.reshape(n, -1))
        assert np.array_equal(dx.float().cpu().numpy(), ref_dx.reshape(n, h, w, c))
    # loss: bfloat16 logits in, float32 arithmetic, rounded gradient out; confusion matrix exact
    for C_, npix in ((24, 5000), (7, 333)):
        logits = _bf(rng.standard_normal((npix, C_)) * 3)
        labels = rng.randint(0, C_, size=npix).astype(np.int32)
        wts = rng.uniform(0.2, 2.0, size=npix).astype(np.float32)
        out = {}
        lab_d, wts_d = torch.from_numpy(labels).to(DEV), torch.from_numpy(wts).to(DEV)
        for dt, sfx in ((torch.float32, ""), (torch.bfloat16, "_bf16")):
            lg = _dev(logits, dt)
            dl = torch.empty(npix, C_, device=DEV, dtype=dt)
            loss = torch.zeros(1, device=DEV)
            cm = torch.zeros(C_, C_, dtype=torch.int64, device=DEV)
            ws = torch.empty(L.softmax_xent_workspace_bytes(npix) // 4 + 8, device=DEV)
            getattr(L, "softmax_xent_fwd_bwd" + sfx)(lg.data_ptr(), lab_d.data_ptr(), wts_d.data_ptr(), npix, C_, 1.0, dl.data_ptr(),
                                                     loss.data_ptr(), cm.data_ptr(), npix, 1, None, None, ws.data_ptr(),
                                                     ws.numel() * 4, st)
            torch.cuda.synchronize()
            out[dt] = (loss.clone(), dl, cm)
        a, b = out[torch.float32], out[torch.bfloat16]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1].to(torch.bfloat16), b[1]) and torch.equal(a[2], b[2])
        assert np.array_equal(b[2].cpu().numpy(), K.confusion_matrix_np(labels, logits, C_, [0]))


PACKED_CASES = [  # kh, kw, cin, cout, stride, dil, padding, N, H, W
    (1, 1, 128, 32, 1, 1, "valid", 2, 12, 20), (3, 3, 32, 32, 1, 1, "same", 2, 9, 13), (3, 3, 32, 32, 1, 4, "same", 1, 16, 16),
    (5, 1, 32, 32, 1, 1, "same", 2, 8, 8), (1, 5, 32, 32, 1, 1, "same", 1, 7, 24), (1, 1, 32, 128, 1, 1, "valid", 2, 10, 10),
    (2, 2, 64, 32, 2, 1, "valid", 2, 12, 16), (3, 3, 64, 64, 1, 1, "same", 1, 12, 12), (1, 1, 48, 80, 1, 1, "valid", 1, 6, 10),
    # wide layers: the kernel rows of a workgroup's 128 output channels staged through LDS (ragged pixel and channel tails)
    (3, 3, 64, 128, 1, 1, "same", 1, 12, 12), (3, 3, 128, 128, 1, 1, "same", 2, 9, 11), (3, 3, 64, 160, 1, 2, "same", 1, 10, 13),
    (1, 1, 256, 256, 1, 1, "valid", 1, 7, 9),
    # narrow outputs: the register-weight kernel on v_mfma_f32_16x16x16_bf16 (forward of the first four, data gradient of all
    # but the fourth; ragged pixel groups)
    (3, 3, 16, 16, 1, 1, "same", 2, 11, 13), (1, 1, 64, 16, 1, 1, "valid", 2, 9, 7), (1, 1, 128, 16, 1, 1, "valid", 1, 10, 10),
    (1, 1, 64, 4, 1, 1, "valid", 1, 12, 9), (1, 1, 16, 64, 1, 1, "valid", 3, 5, 7), (1, 1, 16, 16, 1, 1, "valid", 1, 6, 11),
]


@pytest.mark.parametrize("case", PACKED_CASES, ids=lambda c: "k{}x{}_{}to{}_s{}_d{}_{}".format(*c[:7]))
def test_conv2d_bf16_matrix_instruction_path(case):
    """Forward and data gradient from bfloat16 kernel packs (v_mfma_f32_32x32x16_bf16) against the float32-MFMA kernels run
    on the same bfloat16 activations with the kernel ROUNDED to bfloat16: the products are the same numbers, only the
    summation order differs (float32 accumulation both ways), so after the bfloat16 store all but a few results are
    identical and the rest differ by one bfloat16 ulp.  Fused BatchNormalization statistics: rtol 1e-5."""
    eng, L0 = pkg("engine"), pkg("_lib")
    kh, kw, cin, cout, s, d, pad, N, H, W = case
    rng = np.random.RandomState(kh * 100 + cin + cout + d)
    x = _bf(rng.standard_normal((N, H, W, cin)))
    w = rng.standard_normal((kh, kw, cin, cout)).astype(np.float32) * np.sqrt(2.0 / (kh * kw * cin))
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for packed in (False, True):
        store = eng.ParamStore(torch.bfloat16)
        cv = eng.Conv(store, "c", kh, kw, cin, cout, stride=s, dil=d, padding=pad, bias=True)
        bn = eng.BatchNorm(store, "bn", cout)
        store.materialise(torch.device(DEV))
        store.init_values(1)
        ho, wo = cv.plan(N, H, W)
        bn.plan(torch.device(DEV))
        store.views["c/kernel"].copy_(torch.from_numpy(w if packed else _bf(w)))
        store.views["c/bias"].copy_(torch.from_numpy(b))
        usage = int(store.L.conv2d_pack_usage(cv._dref))
        keep = []
        if packed:
            assert usage & 1, "the library does not take this layer to the bfloat16 matrix instructions"
            sp = store.specs["c/kernel"]
            fp = torch.zeros(2 * store.theta.numel(), dtype=torch.bfloat16, device=DEV)
            rp = torch.zeros_like(fp)
            tab = np.zeros(1, dtype=np.dtype([("offset", "<i8"), ("kh", "<i4"), ("kw", "<i4"), ("cin", "<i4"), ("cout", "<i4")]))
            tab[0] = (sp.offset, kh, kw, cin, cout)
            tabd = torch.from_numpy(tab.view(np.uint8).copy()).to(DEV)
            L0.lib().conv_pack_weights_bf16(tabd.data_ptr(), 1, store.theta.data_ptr(), fp.data_ptr(), rp.data_ptr(), st)
            cv.desc.w_fwd_pack = fp.data_ptr() + 4 * sp.offset
            if usage & 2:
                cv.desc.w_dgrad_pack = rp.data_ptr() + 4 * sp.offset
            keep = [fp, rp, tabd]
        xd = _dev(x, torch.bfloat16)
        y = torch.empty(N, ho, wo, cout, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(cv.ws_bytes // 4 + 64, device=DEV)
        cv.fwd_bn(xd, y, bn, N * ho * wo, True, False, False, ws, st)
        dy = _dev(_bf(np.random.RandomState(7).standard_normal((N, ho, wo, cout))), torch.bfloat16)
        dx = torch.empty(N, H, W, cin, dtype=torch.bfloat16, device=DEV)
        cv.bwd_data(dy, dx, st)
        torch.cuda.synchronize()
        outs[packed] = (y.float().cpu().numpy(), bn.mean.cpu().numpy().copy(), bn.invstd.cpu().numpy().copy(),
                        dx.float().cpu().numpy(), usage)
        del keep
    ref, got = outs[False], outs[True]
    for what, a, r_ in (("forward", got[0], ref[0]),) + ((("data gradient", got[3], ref[3]),) if got[4] & 2 else ()):
        diff = np.abs(a - r_)
        ulp = np.maximum(np.abs(r_), 1e-30) * 2.0 ** -7          # one bfloat16 step is at most 2^-7 of the value
        scale = np.abs(r_).max()
        assert np.all(diff <= ulp + 2.0 ** -8 * 1e-2 * scale), "{}: max deviation {:.3e} on scale {:.3e}".format(what, diff.max(), scale)
        assert (diff > 0).mean() <= 2e-2, "{}: {:.2%} of the results differ".format(what, (diff > 0).mean())
    np.testing.assert_allclose(got[1], ref[1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got[2], ref[2], rtol=1e-5)


TILED_CASES = [  # cin, cout, N, H, W: 3x3 / stride 1 / SAME layers the LDS-tiled bfloat16 kernel takes
    (64, 64, 2, 16, 32), (64, 128, 1, 32, 16), (128, 256, 1, 16, 16), (256, 64, 1, 16, 32), (32, 192, 2, 16, 16), (96, 128, 1, 48, 16),
    (192, 64, 3, 48, 32), (128, 128, 5, 16, 48),
    # large enough for 16-row tiles (the cases above run on 8-row tiles: fewer than 512 workgroups otherwise)
    (64, 64, 2, 256, 256), (128, 128, 8, 128, 128),
    # map sizes that are not multiples of the tile: masked border tiles (120 x 160 is the third level of a 480 x 640 input)
    (64, 64, 2, 24, 40), (128, 128, 1, 30, 50), (64, 128, 3, 9, 17), (64, 64, 1, 120, 160), (64, 64, 2, 250, 270),
]


def _packed_conv(eng, L0, cin, cout, w, b, packed, N, H, W, st, x_ld=None, y_ld=None):
    store = eng.ParamStore(torch.bfloat16)
    cv = eng.Conv(store, "c", 3, 3, cin, cout, stride=1, dil=1, padding="same", bias=b is not None)
    bn_out = eng.BatchNorm(store, "bn", cout)
    bn_in = eng.BatchNorm(store, "bn_in", cin)
    bn_in.add_prelu(store, "pr")
    store.materialise(torch.device(DEV))
    store.init_values(1)
    cv.plan(N, H, W, x_ld=x_ld, y_ld=y_ld)
    bn_out.plan(torch.device(DEV))
    bn_in.plan(torch.device(DEV))
    store.views["c/kernel"].copy_(torch.from_numpy(w if packed else _bf(w)))
    if b is not None:
        store.views["c/bias"].copy_(torch.from_numpy(b))
    keep = []
    if packed:
        usage = int(store.L.conv2d_pack_usage(cv._dref))
        assert usage & 3 == 3
        sp = store.specs["c/kernel"]
        fp = torch.zeros(2 * store.theta.numel(), dtype=torch.bfloat16, device=DEV)
        rp = torch.zeros_like(fp)
        tab = np.zeros(1, dtype=np.dtype([("offset", "<i8"), ("kh", "<i4"), ("kw", "<i4"), ("cin", "<i4"), ("cout", "<i4")]))
        tab[0] = (sp.offset, 3, 3, cin, cout)
        tabd = torch.from_numpy(tab.view(np.uint8).copy()).to(DEV)
        L0.lib().conv_pack_weights_bf16(tabd.data_ptr(), 1, store.theta.data_ptr(), fp.data_ptr(), rp.data_ptr(), st)
        cv.desc.w_fwd_pack = fp.data_ptr() + 4 * sp.offset
        cv.desc.w_dgrad_pack = rp.data_ptr() + 4 * sp.offset
        keep = [fp, rp, tabd]
    return store, cv, bn_out, bn_in, keep


@pytest.mark.parametrize("case", TILED_CASES, ids=lambda c: "{}to{}_n{}_{}x{}".format(*c))
def test_conv3x3_tiled_bf16_kernel(case):
    """The LDS-tiled bfloat16 kernel (forward with fused statistics, plain forward, data gradient with a skip addend, data
    gradient with the fused BatchNormalization backward sums incl. PReLU and dropout) against the float32-MFMA kernels on the
    same bfloat16 tensors and bfloat16-rounded kernel: same products, float32 accumulation in a different order - stored
    results equal or one bfloat16 step apart, reductions to rtol 2e-4 of their scale."""
    eng, L0 = pkg("engine"), pkg("_lib")
    cin, cout, N, H, W = case
    rng = np.random.RandomState(cin + 3 * cout + H)
    x = _bf(rng.standard_normal((N, H, W, cin)))
    w = rng.standard_normal((3, 3, cin, cout)).astype(np.float32) * np.sqrt(2.0 / (9 * cin))
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    dy = _bf(rng.standard_normal((N, H, W, cout)))
    add = _bf(rng.standard_normal((N, H, W, cin)))
    bnx = _bf(rng.standard_normal((N, H, W, cin)) * 1.5 + 0.3)
    drop = (rng.uniform(size=(N, cin)) > 0.3).astype(np.float32) / 0.7
    st = torch.cuda.current_stream().cuda_stream
    npix = N * H * W
    outs = {}
    for packed in (False, True):
        store, cv, bn_out, bn_in, keep = _packed_conv(eng, L0, cin, cout, w, b, packed, N, H, W, st)
        names = []
        xd, dyd, addd, bnxd = (_dev(t, torch.bfloat16) for t in (x, dy, add, bnx))
        dropd = torch.from_numpy(drop).to(DEV)
        ws = torch.empty(max(cv.ws_bytes, 1 << 20) // 4 + 64, device=DEV)
        y = torch.empty(N, H, W, cout, dtype=torch.bfloat16, device=DEV)
        y2 = torch.empty_like(y)
        cv.fwd_bn(xd, y, bn_out, npix, True, False, False, ws, st)
        names.append(store.L.last_kernel())
        cv.fwd(xd, y2, st)
        names.append(store.L.last_kernel())
        y3 = torch.empty_like(y)      # inference phase: BatchNormalization (the statistics just taken) + LeakyReLU in the epilogue
        cv.fwd_bn_act_infer(xd, y3, bn_out, st, act=eng.ACT_LEAKY, slope=0.1)
        infer_name = store.L.last_kernel()
        dx_plain = torch.empty(N, H, W, cin, dtype=torch.bfloat16, device=DEV)
        dx_add = torch.empty_like(dx_plain)
        cv.bwd_data(dyd, dx_plain, st)
        names.append(store.L.last_kernel())
        cv.bwd_data(dyd, dx_add, st, add=addd)
        names.append(store.L.last_kernel())
        # the BatchNormalization + PReLU in front of the convolution: its statistics from its raw input, then the fused backward
        bn_in.stats(bnxd, npix, cin, True, False, False, ws, st)
        prng = np.random.RandomState(99)      # the same parameters in both passes
        store.views["bn_in/gamma"].copy_(torch.from_numpy(prng.uniform(0.5, 1.5, cin).astype(np.float32)))
        store.views["bn_in/beta"].copy_(torch.from_numpy(prng.uniform(-0.5, 0.5, cin).astype(np.float32)))
        store.views["pr/alpha"].copy_(torch.from_numpy(prng.uniform(0.05, 0.4, (1, 1, cin)).astype(np.float32)))
        da = torch.empty_like(dx_plain)
        dxb = torch.empty_like(dx_plain)
        store.grad.zero_()
        cv.bwd_data_bn(dyd, da, bn_in, bnxd, cin, dxb, npix, ws, st, drop=dropd, hw=H * W)
        names.append(store.L.last_kernel())
        store.grad.zero_()
        cv.bwd_weight(xd, dyd, ws, st)
        assert (b"wgrad_patch" in (store.L.last_kernel() or b"")) == (cin % 64 == 0 and cout % 64 == 0)
        torch.cuda.synchronize()
        dw, db = cv.dw.cpu().numpy().astype(np.float64), cv.db.cpu().numpy().astype(np.float64)
        cv.bwd_data_bn(dyd, da, bn_in, bnxd, cin, dxb, npix, ws, st, drop=dropd, hw=H * W)      # (parameter gradients again)
        torch.cuda.synchronize()
        f = lambda t: t.float().cpu().numpy()
        outs[packed] = dict(dw=dw, db=db, y3=f(y3), infer_name=infer_name, y=f(y), y2=f(y2), mean=f(bn_out.mean), invstd=f(bn_out.invstd), dx_plain=f(dx_plain), dx_add=f(dx_add),
                            da=f(da), dxb=f(dxb), dgamma=f(bn_in.dgamma), dbeta=f(bn_in.dbeta), dalpha=f(bn_in.dalpha), names=names)
        del keep
    ref, got = outs[False], outs[True]
    want = [True, True] + [cin % 64 == 0] * 3      # the data gradient is a forward pass over cout -> cin channels
    assert [b"tiled" in (n or b"") for n in got["names"]] == want, got["names"]
    assert not any(b"tiled" in (n or b"") for n in ref["names"]), ref["names"]
    assert b"tiled" in got["infer_name"] and b"tiled" not in ref["infer_name"], (got["infer_name"], ref["infer_name"])
    for k in ("y", "y2", "y3", "dx_plain", "dx_add", "da"):
        a, r_ = got[k], ref[k]
        diff = np.abs(a - r_)
        scale = np.abs(r_).max()
        assert np.all(diff <= np.maximum(np.abs(r_), 1e-30) * 2.0 ** -7 + 2.0 ** -8 * 1e-2 * scale), "{}: max deviation {:.3e} on scale {:.3e}".format(k, diff.max(), scale)
        assert (diff > 0).mean() <= 2e-2, "{}: {:.2%} of the results differ".format(k, (diff > 0).mean())
    assert np.array_equal(got["y"], got["y2"])
    np.testing.assert_allclose(got["mean"], ref["mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["invstd"], ref["invstd"], rtol=1e-5)
    for k in ("dgamma", "dbeta", "dalpha"):
        np.testing.assert_allclose(got[k], ref[k], rtol=2e-4, atol=2e-4 * np.abs(ref[k]).max())
    # weight gradient (staged-patch kernel when input and output channels come in whole 64-blocks): float32 sums of exact
    # products against float64
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    ref_dw = np.stack([np.stack([np.einsum("nyxi,nyxo->io", xp[:, ky:ky + H, kx:kx + W], dy.astype(np.float64)) for kx in range(3)])
                       for ky in range(3)])
    for k in (False, True):
        assert np.abs(outs[k]["dw"] - ref_dw).max() <= 2e-5 * np.abs(ref_dw).max(), np.abs(outs[k]["dw"] - ref_dw).max()
        np.testing.assert_allclose(outs[k]["db"], dy.astype(np.float64).sum((0, 1, 2)), rtol=1e-5, atol=1e-4)
    # the apply pass runs on da (equal up to single bfloat16 steps) with coefficients equal to 2e-4
    d = np.abs(got["dxb"] - ref["dxb"])
    assert d.max() <= 2e-2 * np.abs(ref["dxb"]).max() and (d > 0).mean() <= 5e-2, (d.max(), (d > 0).mean())


def test_conv3x3_tiled_bf16_kernels_on_channel_slices():
    """The tiled kernels on channel slices of wider buffers (pixel strides larger than the channel counts, as the descriptor
    allows): forward with statistics, data gradient and weight gradient read and write only their slice."""
    eng, L0 = pkg("engine"), pkg("_lib")
    cin, cout, N, H, W, xl, yl, x0, y0 = 64, 128, 2, 32, 32, 96, 192, 16, 64
    rng = np.random.RandomState(5)
    xb = _bf(rng.standard_normal((N, H, W, xl)))
    w = rng.standard_normal((3, 3, cin, cout)).astype(np.float32) * np.sqrt(2.0 / (9 * cin))
    b = rng.standard_normal(cout).astype(np.float32) * 0.1
    dyb = _bf(rng.standard_normal((N, H, W, yl)))
    st = torch.cuda.current_stream().cuda_stream
    outs = {}
    for packed in (False, True):
        store, cv, bn_out, _, keep = _packed_conv(eng, L0, cin, cout, w, b, packed, N, H, W, st, x_ld=xl, y_ld=yl)
        xd, dyd = _dev(xb, torch.bfloat16), _dev(dyb, torch.bfloat16)
        y = torch.full((N, H, W, yl), 7.0, dtype=torch.bfloat16, device=DEV)
        dx = torch.full((N, H, W, xl), 7.0, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(max(cv.ws_bytes, 1 << 20) // 4 + 64, device=DEV)
        xs, ys, dys, dxs = xd[..., x0:], y[..., y0:], dyd[..., y0:], dx[..., x0:]      # slices: only their first element's address is used
        cv.fwd_bn(xs, ys, bn_out, N * H * W, True, False, False, ws, st)
        n1 = store.L.last_kernel()
        cv.bwd_data(dys, dxs, st)
        n2 = store.L.last_kernel()
        store.grad.zero_()
        cv.bwd_weight(xs, dys, ws, st)
        n3 = store.L.last_kernel()
        torch.cuda.synchronize()
        f = lambda t: t.float().cpu().numpy()
        outs[packed] = dict(y=f(y), dx=f(dx), mean=f(bn_out.mean), dw=cv.dw.cpu().numpy().astype(np.float64), names=(n1, n2, n3))
        del keep
    ref, got = outs[False], outs[True]
    assert all(b"tiled" in n or b"wgrad_patch" in n for n in got["names"]), got["names"]
    for k, lo, hi in (("y", y0, y0 + cout), ("dx", x0, x0 + cin)):
        a, r_ = got[k], ref[k]
        assert np.all(a[..., :lo] == 7.0) and np.all(a[..., hi:] == 7.0), k + ": wrote outside its channel slice"
        d = np.abs(a[..., lo:hi] - r_[..., lo:hi])
        assert np.all(d <= np.abs(r_[..., lo:hi]) * 2.0 ** -7 + 2.0 ** -8 * 1e-2 * np.abs(r_).max()) and (d > 0).mean() <= 2e-2, k
    np.testing.assert_allclose(got["mean"], ref["mean"], rtol=1e-5, atol=1e-6)
    xp = np.pad(xb[..., x0:x0 + cin].astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    dyv = dyb[..., y0:y0 + cout].astype(np.float64)
    ref_dw = np.stack([np.stack([np.einsum("nyxi,nyxo->io", xp[:, ky:ky + H, kx:kx + W], dyv) for kx in range(3)]) for ky in range(3)])
    for k in (False, True):
        assert np.abs(outs[k]["dw"] - ref_dw).max() <= 2e-5 * np.abs(ref_dw).max()


def _compiled(name, w0, lr=5e-4, dtype="bf16"):
    models, losses, metrics, xo = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    m = models.get_model(name, (None, None, 3), NC, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY,
                         dtype=dtype).model
    m.set_weights([w0[k] for k in m.weight_names()])
    m.compile(optimizer=xo.Adam(lr=lr), metrics={'logits': [metrics.segmentation_accuracy(0, [0]),
                                                            metrics.segmentation_confusion_matrix(NC, 0, [0])]})
    m.keep_logits = True      # the tests below read the logits the step's loss kernel computed
    return m


def _np(t):
    return t.float().cpu().numpy().astype(np.float64)


def _rel_l2(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.linalg.norm(np.asarray(got, np.float64) - ref) / max(np.linalg.norm(ref), 1e-30))


@pytest.mark.parametrize("name,B,H,W,seed,expand", [("enet-naive-upsampling", 2, 64, 64, 3, False), ("enet-max-unpooling", 2, 64, 96, 5, False),
                                                    ("enet-naive-upsampling-enhanced", 2, 48, 80, 9, False),
                                                    ("enet-naive-upsampling", 2, 64, 64, 3, True)])
def test_bf16_training_step_against_fp64_oracle(name, B, H, W, seed, expand, monkeypatch):
    """expand: the regular bottlenecks recompute their expansion instead of storing it (ssmi_expand_*, the default); False:
    the stored route (SSMI_NO_EXPAND=1)."""
    from test_parity_step_gpu import _hip_branches
    monkeypatch.setenv("SSMI_KEEP_Y3", "1")      # the branch replay reads y3
    if not expand:
        monkeypatch.setenv("SSMI_NO_EXPAND", "1")
    w0 = E.init_weights(name, NC, seed=seed, perturb=True)
    m = _compiled(name, w0)
    cw = np.ones(NC, dtype=np.float32)
    cw[0] = 0
    cw[1:] = 0.5 + np.arange(1, NC) / 23.0
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=seed + 100, block=8, class_weights=cw)
    masks = E.make_dropout_masks(name, B, seed)
    m._ensure_buffers(B, H, W)
    g = m.graph
    assert g.store.act_dtype == torch.bfloat16
    g.drop_groups = []
    for k, mk in masks.items():
        g.drop_views[k].copy_(torch.from_numpy(mk.astype(np.float32)))
    m.use_hip_graph = False
    with torch.cuda.stream(torch.cuda.Stream()):
        outs = m.train_on_batch([images, labels, pw, np.zeros((B, 1), np.int32)])
    torch.cuda.synchronize()
    assert g.logits.dtype == torch.bfloat16 and g.by_name["en_bn_1.0"].y1.dtype == torch.bfloat16
    assert any(getattr(b, "_expanded", False) for b in g.blocks) == expand
    forced = _hip_branches(m, w0)
    hip_logits = g.logits.float().cpu().numpy()
    # --- whole network, quantities that survive compounding --------------------------------------------------------------
    ref = E.train_step(name, w0, images, labels, pw, masks)                       # exact storage, the oracle's own branches
    # layers that run on the bfloat16 matrix instructions multiply bfloat16-rounded kernels: the oracle does the same
    bfw = {k for k, u in g.packed_layers.items() if u & 1}
    assert bfw, "no layer took the bfloat16 matrix-instruction path"
    refq = E.train_step(name, w0, images, labels, pw, masks, storage="bf16", bf16_weights=bfw)   # same arithmetic and storage
    # the loss is a mean over all pixels: storage rounding averages out of it even where single logits moved
    assert abs(outs[0] - ref["loss"]) <= 2e-2 * abs(ref["loss"]), (outs[0], ref["loss"])
    # the kernels' distance to exact storage is the distance ANY evaluation with bfloat16 storage has (here: the fp64
    # oracle with the storage roundings inserted), not more
    e_hip, e_emul = _rel_l2(hip_logits, ref["logits"]), _rel_l2(refq["logits"], ref["logits"])
    assert e_hip <= 1.5 * e_emul + 1e-3, (e_hip, e_emul)
    # bit-exact integer artefacts: pooling indices for the kernel's own (rounded) input, matrix for the kernel's own logits
    for bname in ("en_bn_1.0", "en_bn_2.0"):
        blk = g.by_name[bname]
        pooled, idx = K.max_pool_with_argmax_np(blk.x.float().cpu().numpy())
        assert np.array_equal(blk.idx.cpu().numpy().astype(np.int64), idx), "argmax indices of " + bname
        assert np.array_equal(blk.pooled.float().cpu().numpy(), pooled)
    assert np.array_equal(outs[4], K.confusion_matrix_np(labels, hip_logits, NC, [0]))
    # --- block by block on the kernels' own tensors, rtol 2e-2 -----------------------------------------------------------
    blocks = g.blocks
    # a block whose merge backward formed its dout inside the kernel (chained route) left the next block's input gradient
    # unformed: form it now from the operands the step left behind, so that the replay below has every block's dout - the
    # chained kernels themselves are then held to the oracle through their blocks' gradients
    chained = [i for i, blk in enumerate(blocks) if getattr(blk, "_chained", False)]
    assert bool(chained) == expand
    for i in chained:
        pkg("engine").DeferredInputGradient(blocks[i + 1]).materialise(
            types.SimpleNamespace(st=torch.cuda.current_stream().cuda_stream), weight_gradient=False)
    torch.cuda.synchronize()
    worst = {"out": (0.0, ""), "dx": (0.0, ""), "grad": (0.0, "")}

    def note(kind, e, what):
        if e > worst[kind][0]:
            worst[kind] = (e, what)
        assert e <= 2e-2, "{} of {}: relative L2 error {:.3e}".format(kind, what, e)

    for i, blk in enumerate(blocks):
        last = i == len(blocks) - 1
        bname = "initial_block" if i == 0 else ("head" if last else blk.name)
        x_in = g._images_act if i == 0 else blocks[i - 1].out
        dout = g.dlogits if last else blocks[i + 1].dx
        ui = getattr(blk, "unpool_idx", None)
        r = E.block_step(name, w0, bname, _np(x_in), _np(dout), masks, forced, "bf16",
                         None if ui is None else ui.cpu().numpy().astype(np.int64), bf16_weights=bfw)
        note("out", _rel_l2(_np(blk.out), r["out"]), bname)
        if r["dx"] is not None:
            note("dx", _rel_l2(_np(blk.dx), r["dx"]), bname)
        for k, gr in r["grads"].items():
            got = g.store.gviews[k].cpu().numpy().astype(np.float64)
            if k.endswith("/bias") and k not in ("logits/bias",) + (("initial_block_conv2d/bias",) * (name == "enet-max-unpooling")):
                # every other bias sits in front of a BatchNormalization: its gradient is the sum of a tensor that sums to
                # zero in exact arithmetic, what is left is the rounding residue of that tensor's storage on both sides
                lim = 2e-2 * np.abs(r["grads"][k[:-5] + "/kernel"]).max()
                assert np.abs(got).max() <= lim and np.abs(gr).max() <= lim, k
                continue
            note("grad", _rel_l2(got, gr), k)
    print("{} bf16, block by block against the fp64 oracle with bf16 storage: worst block output {:.2e} ({}), input gradient "
          "{:.2e} ({}), parameter gradient {:.2e} ({}); whole network against exact storage: logits {:.3f} (the oracle with "
          "bf16 storage: {:.3f}), loss {:.5f} vs {:.5f}".format(name, worst["out"][0], worst["out"][1], worst["dx"][0],
                                                               worst["dx"][1], worst["grad"][0], worst["grad"][1], e_hip, e_emul,
                                                               outs[0], ref["loss"]))


def test_bf16_training_loss_falls_and_graph_replay_is_exact():
    name, B, H, W, lr = "enet-naive-upsampling", 2, 64, 64, 2e-3
    w0 = E.init_weights(name, NC, seed=21, perturb=True)
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=33, block=16)
    x = [images, labels, pw, np.zeros((B, 1), np.int32)]
    runs = {}
    for mode in ("graph", "eager"):
        m = _compiled(name, w0, lr=lr)
        m.use_hip_graph = mode == "graph"
        losses_ = []
        with torch.cuda.stream(torch.cuda.Stream()):
            for step in range(30 if mode == "graph" else 4):
                losses_.append(m.train_on_batch(x)[0])
                if step == 3:
                    torch.cuda.synchronize()
                    runs[mode] = (list(losses_), m.graph.store.theta.clone())
        torch.cuda.synchronize()
        if mode == "graph":
            assert losses_[-1] < 0.9 * losses_[0] and np.mean(losses_[-5:]) < np.mean(losses_[:5]), losses_
        m.close()
    assert runs["graph"][0] == runs["eager"][0] and torch.equal(runs["graph"][1], runs["eager"][1])


@pytest.mark.parametrize("name", ["segnet", "unet"])
def test_bf16_wide_models_follow_the_float32_engine(name):
    """SegNet / U-Net with bf16 storage (bf16 matrix instructions in forward, data gradient and the tiled weight gradient)
    against the same engine with float32 storage on the same weights and batch: first-step loss within 2e-2, and the loss
    falls under training.  The parameter gradients of the whole network are only bounded loosely: rounding every stored
    activation to bf16 moves them by 20-25 % in norm on these randomly initialised networks whichever kernels multiply
    (0.249 / 0.207 with the bf16 matrix instructions, the same with the float32 ones in the weight gradients, 0.226 /
    0.201 with float32 instructions everywhere); what the kernels compute is pinned per layer above."""
    B, H, W, lr = 2, 64, 64, 1e-3
    w0 = E.init_weights(name, NC, seed=17, perturb=True)
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=41, block=16)
    x = [images, labels, pw, np.zeros((B, 1), np.int32)]
    first, grads, curve = {}, {}, None
    for dtype in ("f32", "bf16"):
        m = _compiled(name, w0, lr=lr, dtype=dtype)
        m.use_hip_graph = False
        with torch.cuda.stream(torch.cuda.Stream()):
            first[dtype] = m.train_on_batch(x)[0]
            torch.cuda.synchronize()
            grads[dtype] = m.graph.store.grad.double().cpu().numpy().copy()
            if dtype == "bf16":
                curve = [first[dtype]] + [m.train_on_batch(x)[0] for _ in range(24)]
        torch.cuda.synchronize()
        m.close()
    assert abs(first["bf16"] - first["f32"]) <= 2e-2 * abs(first["f32"]), first
    err = _rel_l2(grads["bf16"], grads["f32"])
    print("{}: first-step loss {:.5f} (bf16) / {:.5f} (f32), parameter gradients {:.3e} apart in norm, loss after 25 steps "
          "{:.4f}".format(name, first["bf16"], first["f32"], err, curve[-1]))
    assert err <= 0.5, err
    assert curve[-1] < 0.9 * curve[0] and np.mean(curve[-5:]) < np.mean(curve[:5]), curve


def test_bf16_rejects_unknown_storage_types():
    models = pkg("models")
    with pytest.raises(ValueError):
        models.get_model("enet-naive-upsampling", (None, None, 3), NC, dtype="fp8")


@pytest.mark.parametrize("with_mt", [True, False])
def test_bf16_semi_supervised_student_follows_the_float32_student(with_mt):
    """The mean-teacher / superpixel students with bf16 storage (the loss kernels see float32 copies of the bf16 logits, the
    summed logits gradient is rounded once): first-step loss and both unlabeled-cost terms within 2e-2 of the float32
    student on the same weights and batch, logits gradient within 5e-2 in norm, replayed graphs == eager launches."""
    models, losses, metrics, xo = pkg("models"), pkg("losses"), pkg("metrics"), pkg("keras_extensions.extended_optimizers")
    name, B, H, W, seed = "enet-naive-upsampling", 3, 32, 32, 13
    w0 = E.init_weights(name, NC, seed=seed, perturb=True)
    T = losses.ModelLambdaLossType
    kind = T.SEGMENTATION_SEMI_SUPERVISED_MEAN_TEACHER_SUPERPIXEL if with_mt else T.SEGMENTATION_SEMI_SUPERVISED_SUPERPIXEL
    cw = np.ones(NC, dtype=np.float32)
    cw[0] = 0
    images, labels, pw = E.synthetic_batch(B, H, W, NC, seed=seed + 100, block=8, class_weights=cw)
    rng = np.random.RandomState(seed)
    labels[1:] = np.repeat(np.repeat(rng.randint(0, 3, size=(2, H // 4, W // 4)), 4, axis=1), 4, axis=2)
    mt_logits = rng.standard_normal((B, H, W, NC)).astype(np.float32) * 2
    x = [images, labels, pw, np.full((B, 1), 2, np.int32)]
    if with_mt:
        x += [mt_logits, np.full((B, 1), 1.5, np.float32)]
    x += [np.full((B, 1), 4.0, np.float32)]
    masks = E.make_dropout_masks(name, B, seed)
    res = {}
    for dtype in ("f32", "bf16"):
        for mode in (("eager", "graph") if dtype == "bf16" else ("eager",)):
            m = models.get_model(name, (None, None, 3), NC, kind, dtype=dtype).model
            m.set_weights([w0[k] for k in m.weight_names()])
            m.compile(optimizer=xo.Adam(lr=5e-4), metrics={'logits': [metrics.segmentation_confusion_matrix(NC, 0, [0])]})
            m._ensure_buffers(B, H, W)
            g = m.graph
            g.drop_groups = []
            for k, mk in masks.items():
                g.drop_views[k].copy_(torch.from_numpy(mk.astype(np.float32)))
            m.use_hip_graph = mode == "graph"
            with torch.cuda.stream(torch.cuda.Stream()):
                outs = m.train_on_batch(x, None)
            torch.cuda.synchronize()
            res[dtype, mode] = dict(loss=outs[0], sp=m.superpixel_cost(), mt=m.consistency_cost() if with_mt else 0.0,
                                    dl=g.dlogits.float().cpu().numpy().astype(np.float64), theta=g.store.theta.clone(),
                                    lg=g.logits.float().cpu().numpy().astype(np.float64))
            m.close()
    a, b, c = res["f32", "eager"], res["bf16", "eager"], res["bf16", "graph"]
    for k in ("loss", "sp", "mt"):
        assert abs(b[k] - a[k]) <= 2e-2 * abs(a[k]) + 1e-7, (k, b[k], a[k])
    assert b["loss"] == c["loss"] and torch.equal(b["theta"], c["theta"])
    # the loss side on the model's OWN (bf16) logits against the fp64 oracle: loss to 1e-4, gradient equal but for its one
    # rounding to bf16 (the whole-network distance to the float32 student is storage noise and is only printed)
    lg = torch.tensor(b["lg"], dtype=torch.float64, requires_grad=True)
    tl, tw = torch.tensor(labels.astype(np.int64)), torch.tensor(pw.astype(np.float64))
    if with_mt:
        ref = K.segmentation_mean_teacher_superpixel_lambda_loss(lg, tl, tw, 2, torch.tensor(mt_logits.astype(np.float64)), 1.5, 4.0)
    else:
        ref = K.segmentation_superpixel_lambda_loss(lg, tl, tw, 2, 4.0)
    (dref,) = torch.autograd.grad(ref, [lg])
    dref = dref.numpy()
    assert abs(b["loss"] - ref.item()) <= 1e-4 * abs(ref.item()), (b["loss"], ref.item())
    assert np.all(np.abs(b["dl"] - dref) <= np.abs(dref) * 2.0 ** -7 + 1e-4 * np.abs(dref).max()), np.abs(b["dl"] - dref).max()
    print("bf16 student vs float32 student: logits gradient {:.3f} apart in norm".format(_rel_l2(b["dl"], a["dl"])))


@pytest.mark.parametrize("name", ["enet-naive-upsampling-encoder-only", "segnet-encoder-only"])
def test_bf16_encoder_only_classification_step_follows_float32(name):
    """Encoder-only classification models with bf16 storage (global average pooling and the loss see float32 copies): loss
    within 2e-2 of the float32 model, logits within 5e-2 in norm, replayed graph == eager launches, prediction shape."""
    models, losses, xo = pkg("models"), pkg("losses"), pkg("keras_extensions.extended_optimizers")
    B, H, W, seed = 2, 64, 64, 17
    w0 = E.init_weights(name, NC, seed=seed, perturb=True)
    rng = np.random.RandomState(seed)
    images = rng.standard_normal((B, H, W, 3)).astype(np.float32)
    y = (rng.uniform(size=(B, NC)) < 0.2).astype(np.float32)
    cw = np.broadcast_to(rng.uniform(0.5, 1.5, size=NC).astype(np.float32), (B, NC)).copy()
    masks = E.make_dropout_masks(name, B, seed)
    res = {}
    for dtype in ("f32", "bf16"):
        for mode in (("eager", "graph") if dtype == "bf16" else ("eager",)):
            m = models.get_model(name, (H, W, 3), NC, losses.ModelLambdaLossType.CLASSIFICATION_CATEGORICAL_CROSS_ENTROPY, dtype=dtype).model
            m.set_weights([w0[k] for k in m.weight_names()])
            m.compile(optimizer=xo.Adam(lr=5e-4))
            m._ensure_buffers(B, H, W)
            g = m.graph
            g.drop_groups = []
            for k, mk in masks.items():
                g.drop_views[k].copy_(torch.from_numpy(mk.astype(np.float32)))
            m.use_hip_graph = mode == "graph"
            with torch.cuda.stream(torch.cuda.Stream()):
                outs = m.train_on_batch([images, y, cw, np.zeros((B, 1), np.int32)], None)
            torch.cuda.synchronize()
            res[dtype, mode] = dict(loss=outs[0], logits=g.logits.float().cpu().numpy().astype(np.float64), theta=g.store.theta.clone())
            if mode == "graph":
                assert m.predict_on_batch(images).shape == (B, NC)
            m.close()
    a, b, c = res["f32", "eager"], res["bf16", "eager"], res["bf16", "graph"]
    assert abs(b["loss"] - a["loss"]) <= 2e-2 * abs(a["loss"]), (b["loss"], a["loss"])
    assert _rel_l2(b["logits"], a["logits"]) <= 5e-2, _rel_l2(b["logits"], a["logits"])
    assert b["loss"] == c["loss"] and torch.equal(b["theta"], c["theta"])
