"""The expansion entry points (ssmi_expand_*, include/ssmi.h) called through the C ABI on single blocks, against a float64
restatement of what they replace (src/models.py:1142-1172: Conv2D 1x1 -> BatchNormalization -> SpatialDropout2D -> Add -> PReLU,
with the BatchNormalization + PReLU in front of the convolution applied to its operand; 1090-1100 for the next block's 1x1
reduction the *_through / *_chained forms carry).  Sizes are RAGGED on purpose - pixel counts that are no multiple of the
kernels' 32- / 64-pixel tiles and that leave workgroups with different tile counts - because the whole-network tests
(test_prebn_gpu.py, test_parity_step_gpu.py) only reach these kernels with whole tiles.
float32: 1e-4 of each tensor's scale (float32 sums in another order); bfloat16 storage: 2e-2 (operand, kernel matrices and the
recomputed expansion are rounded to bfloat16 as the stored route's tensors are; the restatement rounds the inputs only).
The backward restatement takes the branch of the last PReLU from the kernels' own forward output (its slopes are positive here,
so the sign of the output is the sign of the pre-activation): an element whose pre-activation the bfloat16 roundings move across
zero would otherwise differ by a whole gradient value - the branch-replay rule of the whole-network oracle tests."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPS, MOM = 1e-3, 0.1

# (the last two: more tiles than workgroups, so some workgroups walk two tiles and some one)
CASES = [(32, 128, 1, 5, 3), (32, 128, 3, 7, 9), (16, 64, 1, 5, 6), (16, 64, 2, 9, 11), (32, 128, 2, 40, 33), (32, 128, 4, 70, 67),
         (16, 64, 4, 100, 91)]


def _prelu(z, a):
    return torch.where(z > 0, z, a * z)


class _Block:
    """Random operands of one bottleneck tail (and of the reduction of a block behind it), float64 masters + device copies."""

    def __init__(self, c4, c, n, h, w, dtype, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        self.c4, self.c, self.n, self.h, self.w = c4, c, n, h, w
        self.adt = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.npix = n * h * w

        def rnd(*shape, scale=1.0, lo=None, hi=None):
            t = torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo if lo is not None else \
                torch.randn(*shape, generator=g, dtype=torch.float64) * scale
            return t

        def act(t):      # activation tensors live in the storage type: the restatement sees the rounded values
            return t.to(self.adt).to(DEV)

        def par(t):
            return t.to(torch.float32).to(DEV)

        self.y2 = act(rnd(n, h, w, c4))
        self.skip = act(rnd(n, h, w, c))
        self.dout = act(rnd(n, h, w, c))
        self.dy_next = act(rnd(n, h, w, c4, scale=0.5))
        self.dsum_next = act(rnd(n, h, w, c, scale=0.5))
        self.w3 = par(rnd(c4, c, scale=0.3))
        self.w1n = par(rnd(c, c4, scale=0.2))
        self.g2, self.b2 = par(rnd(c4, lo=0.5, hi=1.5)), par(rnd(c4, lo=-0.5, hi=0.5))
        self.a2 = par(rnd(c4, lo=-0.2, hi=0.4))
        self.g3, self.b3 = par(rnd(c, lo=0.5, hi=1.5)), par(rnd(c, lo=-0.5, hi=0.5))
        self.a3 = par(rnd(c, lo=0.05, hi=0.4))
        self.drop = par((torch.rand(n, c, generator=g) > 0.2).double() / 0.8)
        # BN2's statistics are the batch statistics of y2 (its backward assumes so)
        y2d = self.y2.double().reshape(-1, c4)
        self.m2 = par(y2d.mean(0))
        self.is2 = par(1.0 / torch.sqrt(y2d.var(0, unbiased=False) + EPS))
        self.lib = pkg("_lib")
        self.L = self.lib.lib()
        d = self.lib.ExpandDesc()
        d.n, d.h, d.w, d.cin, d.cout, d.dtype = n, h, w, c4, c, (self.lib.BF16 if dtype == "bf16" else self.lib.F32)
        self.d = d
        assert self.L.expand_supported(C.byref(d))
        self.ws = torch.empty(self.L.expand_workspace_bytes(C.byref(d)) // 4 + 16, dtype=torch.float32, device=DEV)
        self.bn2 = self._bn(self.m2, self.is2, self.g2, self.b2)

    def _bn(self, m, i, g, b):
        p = self.lib.BnParams()
        p.mean, p.invstd, p.gamma, p.beta = m.data_ptr(), i.data_ptr(), g.data_ptr(), b.data_ptr()
        return p

    # ---- float64 restatement ------------------------------------------------------------------------------------
    def reference(self, dout=None, mask3=None):
        """Forward and (with dout) backward of the tail in float64 on the stored operands; autograd through training-phase
        BatchNormalizations gives exactly what the merge backward / BN backward kernels compute."""
        y2 = self.y2.double().requires_grad_(True)
        g2, b2, a2 = (t.double().requires_grad_(True) for t in (self.g2, self.b2, self.a2))
        g3, b3, a3 = (t.double().requires_grad_(True) for t in (self.g3, self.b3, self.a3))
        w3 = self.w3.double()
        if self.adt == torch.bfloat16:
            w3 = w3.to(torch.bfloat16).double()
        skip = self.skip.double().requires_grad_(True)
        m2 = y2.reshape(-1, self.c4).mean(0)
        v2 = y2.reshape(-1, self.c4).var(0, unbiased=False)
        x2 = (y2 - m2) / torch.sqrt(v2 + EPS)
        act2 = _prelu(g2 * x2 + b2, a2)
        y3 = act2 @ w3
        y3.retain_grad()
        m3 = y3.reshape(-1, self.c).mean(0)
        v3 = y3.reshape(-1, self.c).var(0, unbiased=False)
        z3 = g3 * (y3 - m3) / torch.sqrt(v3 + EPS) + b3
        s = z3 * self.drop.double()[:, None, None, :] + skip
        out = _prelu(s, a3) if mask3 is None else torch.where(mask3, s, a3 * s)
        r = dict(y3=y3.detach(), mean3=m3.detach(), invstd3=(1 / torch.sqrt(v3 + EPS)).detach(), out=out.detach())
        if dout is not None:
            out.backward(dout)
            r.update(dsum=skip.grad, dmain=y3.grad, dgamma3=g3.grad, dbeta3=b3.grad, dalpha3=a3.grad, dy2=y2.grad, dgamma2=g2.grad,
                     dbeta2=b2.grad, dalpha2=a2.grad)
        return r


def _close(got, ref, tol, what):
    ref = ref.double()
    err = float((got.double() - ref).abs().max())
    scale = float(ref.abs().max())
    assert err <= tol * scale + 1e-7, "{}: max error {:.3e} against scale {:.3e}".format(what, err, scale)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("c4,c,n,h,w", CASES)
def test_expansion_entry_points_on_ragged_blocks(c4, c, n, h, w, dtype):
    B = _Block(c4, c, n, h, w, dtype, seed=c + h * w)
    L, d, st = B.L, C.byref(B.d), torch.cuda.current_stream().cuda_stream
    tol = 2e-2 if dtype == "bf16" else 1e-4
    ref = B.reference()
    wsb = B.ws.numel() * 4
    # ---- statistics, merge forward ----------------------------------------------------------------------------------
    m3 = torch.zeros(c, device=DEV)
    is3 = torch.zeros(c, device=DEV)
    mm, mv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    L.expand_bn_stats(d, B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), EPS, MOM, 0, mm.data_ptr(), mv.data_ptr(),
                      m3.data_ptr(), is3.data_ptr(), B.ws.data_ptr(), wsb, st)
    torch.cuda.synchronize()
    _close(m3, ref["mean3"], tol, "BN3 mean")
    _close(is3, ref["invstd3"], tol, "BN3 invstd")
    _close(mm, (1 - MOM) * ref["mean3"], tol, "moving mean")
    bn3 = B._bn(m3, is3, B.g3, B.b3)
    out = torch.full((n, h, w, c), 7.0, dtype=B.adt, device=DEV)
    y3 = torch.full_like(out, 7.0)
    L.expand_merge_fwd(d, B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(),
                       B.skip.data_ptr(), c, B.a3.data_ptr(), out.data_ptr(), y3.data_ptr(), st)
    torch.cuda.synchronize()
    _close(y3, ref["y3"], tol, "y3")
    _close(out, ref["out"], tol, "out")
    # ---- the same carried through the next block's reduction ----------------------------------------------------------
    out2 = torch.full_like(out, 7.0)
    y1n = torch.full((n, h, w, c4), 7.0, dtype=B.adt, device=DEV)
    m1, is1 = torch.zeros(c4, device=DEV), torch.zeros(c4, device=DEV)
    L.expand_merge_fwd_through(d, B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(),
                               B.skip.data_ptr(), c, B.a3.data_ptr(), out2.data_ptr(), None, B.w1n.data_ptr(), y1n.data_ptr(), EPS, MOM, 0,
                               None, None, m1.data_ptr(), is1.data_ptr(), B.ws.data_ptr(), wsb, st)
    torch.cuda.synchronize()
    assert torch.equal(out2, out), "the carried-through forward changes the block output"
    w1 = B.w1n.double()
    if dtype == "bf16":
        w1 = w1.to(torch.bfloat16).double()
    y1r = out.double() @ w1      # on the stored (rounded) output, as the separate launch would read it
    _close(y1n, y1r, tol, "y1 of the next block")
    _close(m1, y1r.reshape(-1, c4).mean(0), tol, "BN1' mean")
    _close(is1, 1 / torch.sqrt(y1r.reshape(-1, c4).var(0, unbiased=False) + EPS), tol, "BN1' invstd")
    # ---- inference-phase form: BN1' (fixed statistics) + PReLU1' applied to the carried-through product ----------------
    gen = torch.Generator(device="cpu").manual_seed(5)
    g1 = (torch.rand(c4, generator=gen) + 0.5).to(DEV)
    b1 = (torch.rand(c4, generator=gen) - 0.5).to(DEV)
    al1 = (torch.rand(c4, generator=gen) * 0.6 - 0.2).to(DEV)
    bn1 = B._bn(m1, is1, g1, b1)
    out3 = torch.full_like(out, 7.0)
    a1n = torch.full((n, h, w, c4), 7.0, dtype=B.adt, device=DEV)
    L.expand_merge_fwd_through_infer(d, B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(),
                                     B.skip.data_ptr(), c, B.a3.data_ptr(), out3.data_ptr(), B.w1n.data_ptr(), C.byref(bn1),
                                     al1.data_ptr(), a1n.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(out3, out)
    z1 = g1.double() * (y1r - m1.double()) * is1.double() + b1.double()
    _close(a1n, _prelu(z1, al1.double()), tol, "activated input of the next block's second convolution")
    # ---- merge backward: plain, chained, carried through --------------------------------------------------------------
    mask3 = out.double() > 0
    for form in ("plain", "chained", "through", "chained+through"):
        chained, through = "chained" in form, "through" in form
        dout = (B.dy_next.double() @ w1.t() + B.dsum_next.double()) if chained else B.dout.double()
        if chained and dtype == "bf16":
            dout = dout.to(torch.bfloat16).double()      # the tensor the two-launch route would have stored
        r = B.reference(dout, mask3)
        dmain = torch.full((n, h, w, c), 7.0, dtype=B.adt, device=DEV)
        dsum = torch.full_like(dmain, 7.0)
        dg3, db3, da3 = (torch.full((c,), 7.0, device=DEV) for _ in range(3))
        dy2 = torch.full((n, h, w, c4), 7.0, dtype=B.adt, device=DEV)
        dg2, db2, da2 = (torch.full((c4,), 7.0, device=DEV) for _ in range(3))
        tail = (B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(), B.skip.data_ptr(), c,
                B.a3.data_ptr(), dmain.data_ptr(), dsum.data_ptr(), dg3.data_ptr(), db3.data_ptr(), da3.data_ptr())
        nxt = (B.dy_next.data_ptr(), B.w1n.data_ptr(), B.dsum_next.data_ptr())
        if through:
            L.expand_merge_bwd_through(d, *((None,) + nxt if chained else (B.dout.data_ptr(), None, None, None)), *tail, dy2.data_ptr(),
                                       dg2.data_ptr(), db2.data_ptr(), da2.data_ptr(), None, None, None, B.ws.data_ptr(), wsb, st)
        elif chained:
            L.expand_merge_bwd_chained(d, *nxt, *tail, B.ws.data_ptr(), wsb, st)
        else:
            L.expand_merge_bwd(d, B.dout.data_ptr(), *tail, B.ws.data_ptr(), wsb, st)
        torch.cuda.synchronize()
        for k, t in (("dsum", dsum), ("dmain", dmain), ("dgamma3", dg3), ("dbeta3", db3), ("dalpha3", da3)):
            _close(t, r[k], tol, form + " " + k)
        if through:
            # the BN2 backward works on the (rounded) dmain the kernel holds; its sums cancel, so the bound is on the scale of
            # the gradient in front of the BatchNormalization
            for k, t in (("dy2", dy2), ("dgamma2", dg2), ("dbeta2", db2), ("dalpha2", da2)):
                _close(t, r[k], 5 * tol, form + " " + k)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_fused_weight_gradients_and_deferred_apply(dtype):
    """wpart3 / wpart1_next + ssmi_expand_weight_reduce against the products they stand for, and next_apply against
    the apply pass of ssmi_bn_act_bwd's arithmetic, on a ragged block."""
    c4, c, n, h, w = 32, 128, 2, 9, 7
    B = _Block(c4, c, n, h, w, dtype, seed=3)
    L, d, st = B.L, C.byref(B.d), torch.cuda.current_stream().cuda_stream
    lib = B.lib
    tol = 2e-2 if dtype == "bf16" else 1e-4
    wsb = B.ws.numel() * 4
    ref0 = B.reference()
    m3, is3 = ref0["mean3"].float().contiguous(), ref0["invstd3"].float().contiguous()
    bn3 = B._bn(m3, is3, B.g3, B.b3)
    # the next block's BN1 + PReLU1 backward, stopped after its reduction: da1 = gradient behind them, y1 raw, coef = k1 | m1 | m2
    g = torch.Generator(device="cpu").manual_seed(11)
    y1 = torch.randn(n, h, w, c4, generator=g, dtype=torch.float64).to(B.adt).to(DEV)
    m1 = y1.double().reshape(-1, c4).mean(0).float()
    is1 = (1 / torch.sqrt(y1.double().reshape(-1, c4).var(0, unbiased=False) + EPS)).float()
    g1 = (torch.rand(c4, generator=g) + 0.5).to(DEV)
    b1 = (torch.rand(c4, generator=g) - 0.5).to(DEV)
    a1 = (torch.rand(c4, generator=g) * 0.6 - 0.2).to(DEV)
    coef = (torch.rand(3 * c4, generator=g) - 0.3).to(DEV)
    bn1 = B._bn(m1, is1, g1, b1)
    da1 = B.dy_next.clone()
    xh = (y1.double() - m1.double()) * is1.double()
    z = g1.double() * xh + b1.double()
    gr = torch.where(z > 0, torch.ones_like(z), torch.where(z < 0, a1.double().expand_as(z), torch.zeros_like(z)))
    k1, q1, q2 = coef[:c4].double(), coef[c4:2 * c4].double(), coef[2 * c4:].double()
    dy1 = k1 * (da1.double() * gr - q1 - xh * q2)
    if dtype == "bf16":
        dy1 = dy1.to(torch.bfloat16).double()
    w1 = B.w1n.double()
    if dtype == "bf16":
        w1 = w1.to(torch.bfloat16).double()
    dout = dy1 @ w1.t() + B.dsum_next.double()
    if dtype == "bf16":
        dout = dout.to(torch.bfloat16).double()
    out_k = torch.empty((n, h, w, c), dtype=B.adt, device=DEV)      # the kernels' own forward output: its sign is the PReLU branch
    L.expand_merge_fwd(d, B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(),
                       B.skip.data_ptr(), c, B.a3.data_ptr(), out_k.data_ptr(), None, st)
    torch.cuda.synchronize()
    r = B.reference(dout, out_k.double() > 0)
    na = lib.ExpandNextApply()
    na.y1, na.bn1, na.alpha, na.coef = y1.data_ptr(), C.pointer(bn1), a1.data_ptr(), coef.data_ptr()
    nfl = L.expand_weight_partial_bytes(d) // 4
    wp3 = torch.full((nfl,), 7.0, device=DEV)
    wp1 = torch.full((nfl,), 7.0, device=DEV)
    dsum = torch.full((n, h, w, c), 7.0, dtype=B.adt, device=DEV)
    dg3, db3, da3 = (torch.full((c,), 7.0, device=DEV) for _ in range(3))
    dy2 = torch.full((n, h, w, c4), 7.0, dtype=B.adt, device=DEV)
    dg2, db2, da2 = (torch.full((c4,), 7.0, device=DEV) for _ in range(3))
    # (float32 storage: the fused weight gradient and the deferred apply pass do not go together - registers; the apply pass is
    # then checked on its own and the weight gradients take the applied gradient)
    use_na = dtype == "bf16"
    if not use_na:
        chk = da1.clone()
        dsum0 = torch.empty((n, h, w, c), dtype=B.adt, device=DEV)
        dm0 = torch.empty_like(dsum0)
        L.expand_merge_bwd_through(d, None, chk.data_ptr(), B.w1n.data_ptr(), B.dsum_next.data_ptr(), B.y2.data_ptr(), C.byref(B.bn2),
                                   B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(), B.skip.data_ptr(), c, B.a3.data_ptr(),
                                   dm0.data_ptr(), dsum0.data_ptr(), dg3.data_ptr(), db3.data_ptr(), da3.data_ptr(), None, None, None, None,
                                   None, None, C.byref(na), B.ws.data_ptr(), wsb, st)
        torch.cuda.synchronize()
        _close(chk, dy1, tol, "dy1 written back by the apply pass")
        _close(dsum0, r["dsum"], tol, "dsum behind the apply pass")
        with pytest.raises(lib.SsmiError):
            L.expand_merge_bwd_through(d, None, chk.data_ptr(), B.w1n.data_ptr(), B.dsum_next.data_ptr(), B.y2.data_ptr(), C.byref(B.bn2),
                                       B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(), B.skip.data_ptr(), c,
                                       B.a3.data_ptr(), dm0.data_ptr(), dsum0.data_ptr(), dg3.data_ptr(), db3.data_ptr(), da3.data_ptr(), None,
                                       None, None, None, None, wp1.data_ptr(), C.byref(na), B.ws.data_ptr(), wsb, st)
        da1 = dy1.to(B.adt).contiguous()
    L.expand_merge_bwd_through(d, None, da1.data_ptr(), B.w1n.data_ptr(), B.dsum_next.data_ptr(), B.y2.data_ptr(), C.byref(B.bn2),
                               B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), B.drop.data_ptr(), B.skip.data_ptr(), c, B.a3.data_ptr(),
                               None, dsum.data_ptr(), dg3.data_ptr(), db3.data_ptr(), da3.data_ptr(), dy2.data_ptr(), dg2.data_ptr(),
                               db2.data_ptr(), da2.data_ptr(), wp3.data_ptr(), wp1.data_ptr(), C.byref(na) if use_na else None,
                               B.ws.data_ptr(), wsb, st)
    dw3 = torch.full((c4, c), 7.0, device=DEV)
    dw1 = torch.full((c, c4), 7.0, device=DEV)
    L.expand_weight_reduce(d, wp3.data_ptr(), dw3.data_ptr(), st)
    L.expand_weight_reduce(d, wp1.data_ptr(), dw1.data_ptr(), st)
    torch.cuda.synchronize()
    _close(da1, dy1, tol, "dy1 written back by the apply pass")
    _close(dsum, r["dsum"], tol, "dsum")
    _close(dy2, r["dy2"], 5 * tol, "dy2")
    # dW3 = a2^T dmain, dW1' = out^T dy1' (operands as the separate launches would read them: stored, rounded tensors)
    y2 = B.y2.double()
    act2 = _prelu(B.g2.double() * (y2 - B.m2.double()) * B.is2.double() + B.b2.double(), B.a2.double())
    dmain, out = r["dmain"], r["out"]
    if dtype == "bf16":
        act2, dmain, out = (t.to(torch.bfloat16).double() for t in (act2, dmain, out))
    _close(dw3, act2.reshape(-1, c4).t() @ dmain.reshape(-1, c), 2 * tol, "dW3")
    _close(dw1, out.reshape(-1, c).t() @ dy1.reshape(-1, c4), 2 * tol, "dW1'")


def test_expansion_entry_points_reject_what_they_cannot_do():
    lib = pkg("_lib")
    L = lib.lib()
    d = lib.ExpandDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.dtype = 1, 4, 4, 8, 32, lib.F32
    assert not L.expand_supported(C.byref(d)) and L.expand_workspace_bytes(C.byref(d)) == 0
    with pytest.raises(lib.SsmiError):
        L.expand_bn_stats(C.byref(d), None, None, None, None, EPS, MOM, 0, None, None, None, None, None, 0, 0)
    d.cin, d.cout = 32, 128
    B = _Block(32, 128, 1, 4, 4, "f32", seed=1)
    st = torch.cuda.current_stream().cuda_stream
    m3 = torch.zeros(128, device=DEV)
    with pytest.raises(lib.SsmiError):      # workspace too small
        L.expand_bn_stats(C.byref(B.d), B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), EPS, MOM, 0, None, None,
                          m3.data_ptr(), m3.data_ptr(), B.ws.data_ptr(), 16, st)
    bn3 = B._bn(m3, m3, B.g3, B.b3)
    with pytest.raises(lib.SsmiError):      # neither dout nor the chained operands
        L.expand_merge_bwd_through(C.byref(B.d), None, None, None, None, B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(),
                                   C.byref(bn3), None, B.skip.data_ptr(), 128, B.a3.data_ptr(), B.dout.data_ptr(), B.dout.data_ptr(),
                                   m3.data_ptr(), m3.data_ptr(), m3.data_ptr(), B.y2.data_ptr(), m3.data_ptr(), m3.data_ptr(), m3.data_ptr(),
                                   None, None, None, B.ws.data_ptr(), B.ws.numel() * 4, st)
    with pytest.raises(lib.SsmiError):      # skip channels beyond the output's
        L.expand_merge_fwd(C.byref(B.d), B.y2.data_ptr(), C.byref(B.bn2), B.a2.data_ptr(), B.w3.data_ptr(), C.byref(bn3), None,
                           B.skip.data_ptr(), 132, B.a3.data_ptr(), B.dout.data_ptr(), None, st)
