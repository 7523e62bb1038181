"""Writes tests/golden/ref_metrics.npz: outputs of the REFERENCE's own pure-numpy definitions on seeded inputs.

Run in the build container only (it reads /root/reference, which never travels):
    python tests/golden/make_reference_metric_vectors.py

The reference tree is Python 2 (print statements), so its modules cannot be imported or even parsed whole; the four
functions used here are plain numpy and valid Python 3.  Their source text is cut out of the reference files at run time
(def line to the next top-level definition), compiled and called; nothing of that text is written anywhere.  Functions:
    get_accuracy / get_mpca / get_miou   src/scripts/calculate_metrics_from_cfms.py:12-53
    np_softmax                           src/predict.py:37-79
    _calculate_mfb_class_weights / _calculate_enet_class_weights   src/utils/dataset_utils.py:498-529 (methods; self unused)
The fixture holds inputs and the functions' outputs only.
"""
import os
import re
import textwrap

import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_metrics.npz")


def _function(path, name):
    """Compiles the function `name` from the text of `path` and returns it."""
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"\s*def {}\(".format(re.escape(name)), l))
    indent = len(lines[start]) - len(lines[start].lstrip())
    end = start + 1
    while end < len(lines) and (not lines[end].strip() or len(lines[end]) - len(lines[end].lstrip()) > indent):
        end += 1
    src = textwrap.dedent("\n".join(lines[start:end]))
    ns = {"np": np}
    exec(compile(src, "{}:{}".format(path, name), "exec"), ns)
    return ns[name]


def main():
    cfms = os.path.join(REF, "scripts", "calculate_metrics_from_cfms.py")
    get_accuracy, get_mpca, get_miou = (_function(cfms, n) for n in ("get_accuracy", "get_mpca", "get_miou"))
    np_softmax = _function(os.path.join(REF, "predict.py"), "np_softmax")
    du = os.path.join(REF, "utils", "dataset_utils.py")
    mfb, enet = _function(du, "_calculate_mfb_class_weights"), _function(du, "_calculate_enet_class_weights")

    rng = np.random.RandomState(20240611)
    out = {}
    cms, acc, mpca, miou, mpca_all, miou_all = [], [], [], [], [], []
    for trial in range(8):
        n = 24
        cm = rng.randint(0, 5000, size=(n, n)).astype(np.int64)
        cm[0, :] = 0                        # the ignored background class never appears as a label
        if trial == 1:
            cm[7, :] = 0
            cm[:, 7] = 0                    # a class that is absent altogether
        if trial == 2:
            cm *= 30000                     # counts beyond 2**24 (the float32 numerators of the script round)
        if trial == 3:
            cm = np.diag(np.diag(cm))       # perfect predictions
        if trial == 4:
            cm[:, 3] = 0                    # a class that is never predicted
        cms.append(cm)
        # the functions overwrite zero denominators in arrays derived from their argument: pass copies
        acc.append(get_accuracy(cm.copy()))
        mpca.append(get_mpca(cm.copy(), [0]))
        miou.append(get_miou(cm.copy(), [0]))
        mpca_all.append(get_mpca(cm.copy(), None))
        miou_all.append(get_miou(cm.copy(), None))
    out.update(cm=np.stack(cms), accuracy=np.array(acc, np.float64), mpca_ignore0=np.array(mpca, np.float64),
               miou_ignore0=np.array(miou, np.float64), mpca_all=np.array(mpca_all, np.float64),
               miou_all=np.array(miou_all, np.float64))
    logits = (rng.standard_normal((3, 5, 7, 24)) * 4).astype(np.float32)
    out["softmax_in"] = logits
    out["softmax_out"] = np_softmax(logits, axis=-1)
    out["softmax_theta_out"] = np_softmax(logits.astype(np.float64), theta=0.5, axis=-1)
    probs = rng.uniform(0.0, 1.0, size=24)
    probs[0] = 0.0
    probs[11] = 0.0
    probs /= probs.sum()
    out["class_probabilities"] = probs
    out["mfb_class_weights"] = np.array(mfb(None, probs), np.float64)
    out["enet_class_weights"] = np.array(enet(None, probs), np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
