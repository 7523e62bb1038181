"""Per-layer timing of the convolution entry points on the unique conv shapes of ENet @512x512, batch 16."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import pkg
eng = pkg("engine"); L = pkg("_lib").lib()

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
# (label, kh, kw, cin, cout, stride, dil, padding, bias, transpose, H_in)
SHAPES = [
    ("init 3x3s2 3->13", 3, 3, 3, 13, 2, 1, "same", True, False, 512),
    ("2x2s2 16->16", 2, 2, 16, 16, 2, 1, "valid", False, False, 256),
    ("3x3 16->16 @128", 3, 3, 16, 16, 1, 1, "same", True, False, 128),
    ("1x1 16->64 @128", 1, 1, 16, 64, 1, 1, "valid", False, False, 128),
    ("1x1 64->16 @128", 1, 1, 64, 16, 1, 1, "valid", False, False, 128),
    ("2x2s2 64->32", 2, 2, 64, 32, 2, 1, "valid", False, False, 128),
    ("3x3 32->32 @64", 3, 3, 32, 32, 1, 1, "same", True, False, 64),
    ("3x3 d16 32->32 @64", 3, 3, 32, 32, 1, 16, "same", True, False, 64),
    ("1x5 32->32 @64", 1, 5, 32, 32, 1, 1, "same", False, False, 64),
    ("5x1 32->32 @64", 5, 1, 32, 32, 1, 1, "same", True, False, 64),
    ("1x1 32->128 @64", 1, 1, 32, 128, 1, 1, "valid", False, False, 64),
    ("1x1 128->32 @64", 1, 1, 128, 32, 1, 1, "valid", False, False, 64),
    ("1x1 128->64 @64", 1, 1, 128, 64, 1, 1, "valid", False, False, 64),
    ("1x1 128->16 @64", 1, 1, 128, 16, 1, 1, "valid", False, False, 64),
    ("t3x3s2 16->16 64->128", 3, 3, 16, 16, 2, 1, "same", True, True, 64),
    ("1x1 64->4 @128", 1, 1, 64, 4, 1, 1, "valid", False, False, 128),
    ("t3x3s2 4->4 128->256", 3, 3, 4, 4, 2, 1, "same", True, True, 128),
    ("1x1 4->16 @256", 1, 1, 4, 16, 1, 1, "valid", False, False, 256),
    ("1x1 16->4 @256", 1, 1, 16, 4, 1, 1, "valid", False, False, 256),
    ("3x3 4->4 @256", 3, 3, 4, 4, 1, 1, "same", True, False, 256),
    ("logits t2x2s2 16->24", 2, 2, 16, 24, 2, 1, "same", True, True, 256),
]
if os.environ.get("SSMI_WIDE"):        # the 3x3 layers of SegNet / U-Net at 512x512 (run with batch 4)
    SHAPES = [("3x3 64->64 @512", 3, 3, 64, 64, 1, 1, "same", True, False, 512),
              ("3x3 128->128 @256", 3, 3, 128, 128, 1, 1, "same", True, False, 256),
              ("3x3 256->256 @128", 3, 3, 256, 256, 1, 1, "same", True, False, 128),
              ("3x3 512->512 @64", 3, 3, 512, 512, 1, 1, "same", True, False, 64),
              ("3x3 1024->512 @64", 3, 3, 1024, 512, 1, 1, "same", False, False, 64)]
if os.environ.get("SSMI_SHAPES"):      # comma-separated indices into SHAPES
    SHAPES = [SHAPES[int(i)] for i in os.environ["SSMI_SHAPES"].split(",")]
ADT = torch.bfloat16 if os.environ.get("SSMI_BENCH_DTYPE") == "bf16" else torch.float32      # activation storage type
dev = torch.device("cuda")
stream = torch.cuda.Stream()
print("%-26s %10s %10s %10s   %8s %8s %8s   GB/s(fwd) TF(fwd) TF(wg)" % ("layer", "fwd us", "bwdD us", "bwdW us", "MB in", "MB out", "GFLOP"))
tot = [0, 0, 0]
with torch.cuda.stream(stream):
    st = stream.cuda_stream
    for (label, kh, kw, cin, cout, s, d, pad, bias, tr, H) in SHAPES:
        store = eng.ParamStore(ADT)
        cv = eng.Conv(store, "c", kh, kw, cin, cout, stride=s, dil=d, padding=pad, bias=bias, transpose=tr)
        store.materialise(dev)
        store.init_values(1)
        ho, wo = cv.plan(B, H, H)
        x = torch.randn(B, H, H, cin, device=dev).to(ADT)
        y = torch.empty(B, ho, wo, cout, device=dev, dtype=ADT)
        dy = torch.randn(B, ho, wo, cout, device=dev).to(ADT)
        dx = torch.empty_like(x)
        ws = torch.empty(cv.ws_bytes // 4 + 16, device=dev)
        def t(fn, reps=20):
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        tf = t(lambda: cv.fwd(x, y, st))
        tb = t(lambda: cv.bwd_data(dy, dx, st))
        tw = t(lambda: cv.bwd_weight(x, dy, ws, st))
        mb_in, mb_out = x.numel() * x.element_size() / 1e6, y.numel() * y.element_size() / 1e6
        macs = (y.numel() if not tr else x.numel() * cout) * kh * kw * cin
        gflop = 2 * macs / 1e9
        print("%-26s %10.1f %10.1f %10.1f   %8.1f %8.1f %8.2f   %8.0f %7.1f %7.1f" % (label, tf, tb, tw, mb_in, mb_out, gflop, (mb_in + mb_out) / tf * 1e3, gflop / tf * 1e3, gflop / tw * 1e3))
