"""Debug helper: per-tensor error profile of one HIP training step vs the fp64 and fp32 CPU oracle."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import pkg
from oracle import enet_oracle as E

model_name = sys.argv[1] if len(sys.argv) > 1 else "enet-naive-upsampling"
B, H, W = 2, int(sys.argv[2]) if len(sys.argv) > 2 else 64, int(sys.argv[3]) if len(sys.argv) > 3 else 64
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 3
nc = 24
models, losses, opt = pkg("models"), pkg("losses"), pkg("keras_extensions.extended_optimizers")
w0 = E.init_weights(model_name, nc, seed=seed, perturb=True)
mb = models.get_model(model_name, (None, None, 3), nc, losses.ModelLambdaLossType.SEGMENTATION_CATEGORICAL_CROSS_ENTROPY)
model = mb.model
model.set_weights([w0[k] for k in model.weight_names()])
model.compile(optimizer=opt.Adam(lr=5e-4))
cw = np.ones(nc, dtype=np.float32); cw[0] = 0; cw[1:] = 0.5 + np.arange(1, nc) / 23.0
images, labels, pw = E.synthetic_batch(B, H, W, nc, seed=seed + 100, block=8, class_weights=cw)
masks = E.make_dropout_masks(model_name, B, seed)
model._ensure_buffers(B, H, W)
g = model.graph
g.drop_groups = []
for name, m in masks.items():
    g.drop_views[name].copy_(torch.from_numpy(m.astype(np.float32)))
ref64 = E.train_step(model_name, w0, images, labels, pw, masks, collect=True)
ref32 = E.train_step(model_name, w0, images, labels, pw, masks, collect=True, dtype=torch.float32)
model.use_hip_graph = False
model.set_batch([images, labels, pw, np.zeros((B, 1), dtype=np.int32)])
# forward+loss+backward only (no optimizer): run segments a and b
segs = model._segments(True)
st = torch.cuda.current_stream().cuda_stream
segs[0](st); segs[1](st)
torch.cuda.synchronize()
print("loss hip %.7f ref64 %.7f ref32 %.7f" % (model._loss_buf.item(), ref64["loss"], ref32["loss"]))
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
print("logits  hip-vs-64 %.2e   32-vs-64 %.2e" % (rel(g.logits.cpu().numpy(), ref64["logits"]), rel(ref32["logits"], ref64["logits"])))
# activations
for b in g.blocks[:-1]:
    nm = b.name if b.name != "initial_block" else "initial_block"
    key = nm + "_prelu_3" if nm != "initial_block" else "initial_block"
    if key in ref64["acts"]:
        print("act %-28s hip-vs-64 %.2e   32-vs-64 %.2e" % (key, rel(b.out.cpu().numpy(), ref64["acts"][key]), rel(ref32["acts"][key], ref64["acts"][key])))
gv = g.store.gviews
keys = list(ref64["grads"].keys())
for k in reversed(keys):
    e1 = rel(gv[k].cpu().numpy(), ref64["grads"][k]); e2 = rel(ref32["grads"][k], ref64["grads"][k])
    flag = " <<<" if e1 > 10 * max(e2, 1e-6) and e1 > 1e-4 else ""
    print("grad %-40s hip-vs-64 %.2e   32-vs-64 %.2e  scale %.2e%s" % (k, e1, e2, np.abs(ref64["grads"][k]).max(), flag))
