import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import pkg
eng = pkg("engine")
for cin, cout in ((32, 128), (128, 32), (32, 32), (16, 128), (32, 64), (32,16)):
    store = eng.ParamStore()
    cv = eng.Conv(store, "c", 1, 1, cin, cout, padding="valid", bias=False)
    store.materialise(torch.device("cuda"))
    N, H, W = 1, 8, 8
    cv.plan(N, H, W)
    rng = np.random.RandomState(1)
    x = rng.standard_normal((N, H, W, cin)); dy = rng.standard_normal((N, H, W, cout))
    ref = x.reshape(-1, cin).T @ dy.reshape(-1, cout)
    ws = torch.empty(cv.ws_bytes // 4 + 8, device="cuda")
    cv.bwd_weight(torch.tensor(x, dtype=torch.float32).cuda(), torch.tensor(dy, dtype=torch.float32).cuda(), ws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = store.gviews["c/kernel"].cpu().numpy().reshape(cin, cout).astype(np.float64)
    bad = np.abs(got - ref) > 1e-3 * np.abs(ref).max()
    print(cin, cout, "bad", int(bad.sum()), "rows with errors", sorted(set(np.where(bad)[0]))[:40], "cols", sorted(set(np.where(bad)[1]))[:40])
    if bad.any():
        r, c = np.where(bad)
        print("  sample", r[0], c[0], got[r[0], c[0]], ref[r[0], c[0]])
        # is got a permutation? find where got[r0,c0] value appears in ref
        d = np.abs(ref - got[r[0], c[0]]); print("  nearest ref at", np.unravel_index(d.argmin(), d.shape), d.min())
