"""Known-answer tests that pin the CPU oracle (the reference ships no test vectors for this path, so these
hand-computable cases plus the reference's own numpy definitions are what anchors the restated TF/Keras semantics)."""
import numpy as np
import pytest
import torch

from oracle import tf_keras_ops as K
from oracle import enet_oracle as E


def test_same_pad_rule():
    # 3x3 stride 2 on an even input pads bottom/right only; stride 1 pads symmetrically; dilation widens the pad
    assert K.same_pad(512, 3, 2) == (256, 0, 1)
    assert K.same_pad(64, 3, 1) == (64, 1, 1)
    assert K.same_pad(64, 3, 1, 16) == (64, 16, 16)
    assert K.same_pad(64, 5, 1) == (64, 2, 2)
    assert K.same_pad(7, 3, 2) == (4, 1, 1)
    assert K.same_pad(64, 2, 2) == (32, 0, 0)


def test_conv_same_stride2_alignment():
    # a single 1 at (0,0) and a 3x3 kernel of tap ids: with pad_before 0 output (0,0) sees tap (0,0) only
    x = torch.zeros(1, 4, 4, 1, dtype=torch.float64)
    x[0, 0, 0, 0] = 1
    w = torch.arange(9, dtype=torch.float64).reshape(3, 3, 1, 1)
    y = K.conv2d(x, w, None, stride=2, padding="same")
    assert y.shape == (1, 2, 2, 1)
    assert y[0, 0, 0, 0] == 0 and y[0, 0, 1, 0] == 0 and y[0, 1, 1, 0] == 0
    # a 1 at (2,2): output (1,1) sees tap (0,0); output (0,0) sees tap (2,2)=8; (0,1) tap (2,0)=6; (1,0) tap (0,2)=2
    x.zero_()
    x[0, 2, 2, 0] = 1
    y = K.conv2d(x, w, None, stride=2, padding="same")
    assert y[0, :, :, 0].tolist() == [[8, 6], [2, 0]]


def test_conv_transpose_same_semantics():
    # y[s*i + k] += x[i] * W[k]; 3x3/s2 'same' crops the full result to the first 2N rows/cols
    x = torch.zeros(1, 2, 2, 1, dtype=torch.float64)
    x[0, 0, 0, 0] = 1
    w = torch.arange(1, 10, dtype=torch.float64).reshape(3, 3, 1, 1)
    y = K.conv2d_transpose(x, w, None, stride=2)
    assert y.shape == (1, 4, 4, 1)
    assert y[0, :3, :3, 0].tolist() == [[1, 2, 3], [4, 5, 6], [7, 8, 9]]
    x.zero_()
    x[0, 1, 1, 0] = 1
    y = K.conv2d_transpose(x, w, None, stride=2)
    assert y[0, 2:, 2:, 0].tolist() == [[1, 2], [4, 5]]      # cropped at 2N
    # it is the adjoint of the SAME forward conv: <conv(a), b> == <a, tconv(b)>
    g = torch.Generator().manual_seed(0)
    a = torch.randn(2, 8, 6, 3, generator=g, dtype=torch.float64)
    b = torch.randn(2, 4, 3, 5, generator=g, dtype=torch.float64)
    w2 = torch.randn(3, 3, 3, 5, generator=g, dtype=torch.float64)   # HWIO of the forward conv = (kh,kw,out_t,in_t)
    lhs = (K.conv2d(a, w2, None, stride=2, padding="same") * b).sum()
    rhs = (a * K.conv2d_transpose(b, w2, None, stride=2)).sum()
    assert abs(float(lhs - rhs)) < 1e-9
    # 2x2/s2: exact tiling
    w3 = torch.randn(2, 2, 4, 5, generator=g, dtype=torch.float64)
    a3 = torch.randn(1, 4, 4, 4, generator=g, dtype=torch.float64)
    b3 = torch.randn(1, 2, 2, 5, generator=g, dtype=torch.float64)
    assert abs(float((K.conv2d(a3, w3, None, stride=2, padding="same") * b3).sum() -
                     (a3 * K.conv2d_transpose(b3, w3, None, stride=2)).sum())) < 1e-9


def test_maxpool_argmax_ties_and_index_formula():
    # first maximum in (h then w) scan order wins; index = (h*W + w)*C + c inside the image
    x = np.zeros((2, 4, 4, 2), dtype=np.float32)
    x[0, 0, 1, 0] = 5
    x[0, 1, 0, 0] = 5          # tie inside window (0,0) channel 0 -> (0,1) wins
    x[1, 3, 3, 1] = 7
    y, idx = K.max_pool_with_argmax_np(x)
    assert y.shape == (2, 2, 2, 2)
    assert idx[0, 0, 0, 0] == (0 * 4 + 1) * 2 + 0 and y[0, 0, 0, 0] == 5
    assert idx[0, 0, 0, 1] == 1                      # all-equal window -> first element (0,0), channel 1
    assert idx[1, 1, 1, 1] == (3 * 4 + 3) * 2 + 1    # batch is NOT part of the index
    assert idx[0, 1, 1, 0] == (2 * 4 + 2) * 2 + 0
    # odd size: windows are clipped, the padding never wins even for negative data
    xo = -np.ones((1, 3, 3, 1), dtype=np.float32)
    yo, io = K.max_pool_with_argmax_np(xo)
    assert yo.shape == (1, 2, 2, 1) and (yo == -1).all()
    assert io[0, :, :, 0].tolist() == [[0, 2], [6, 8]]
    # unpool is the exact inverse placement
    up = K.max_unpool_np(y, idx)
    assert up.shape == x.shape and up[0, 0, 1, 0] == 5 and up[0, 1, 0, 0] == 0 and up[1, 3, 3, 1] == 7
    assert up.sum() == y.sum()


def test_maxpool_argmax_matches_torch_pool():
    rng = np.random.RandomState(0)
    x = rng.standard_normal((2, 6, 8, 3)).astype(np.float32)
    y, idx = K.max_pool_with_argmax_np(x)
    yt, it = torch.nn.functional.max_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), 2, 2, return_indices=True)
    assert np.array_equal(y, yt.permute(0, 2, 3, 1).numpy())
    # torch plane index p <-> TF index p*C + c
    tf_idx = it.permute(0, 2, 3, 1).numpy() * 3 + np.arange(3)[None, None, None, :]
    assert np.array_equal(idx, tf_idx)


def test_weighted_xent_single_pixel():
    z = torch.tensor([[[[1.0, 2.0, 3.0]]]], dtype=torch.float64)
    lab = torch.tensor([[[2]]])
    w = torch.tensor([[[0.5]]], dtype=torch.float64)
    lse = np.log(np.exp(1) + np.exp(2) + np.exp(3))
    assert abs(float(K.sparse_weighted_pixelwise_xent(z, lab, w)) - 0.5 * (lse - 3.0)) < 1e-12
    # the mean runs over ALL pixels, ignored (weight 0) ones included
    z2 = torch.cat([z, z], dim=2)
    lab2 = torch.tensor([[[2, 0]]])
    w2 = torch.tensor([[[0.5, 0.0]]], dtype=torch.float64)
    assert abs(float(K.sparse_weighted_pixelwise_xent(z2, lab2, w2)) - 0.25 * (lse - 3.0)) < 1e-12
    assert np.allclose(K.np_softmax(np.array([1.0, 2.0, 3.0])).sum(), 1.0)


def test_confusion_matrix_orientation_and_metrics():
    labels = np.array([1, 1, 2, 0, 2])
    logits = np.eye(3)[[1, 2, 2, 1, 2]] * 1.0        # predictions 1,2,2,1,2
    cm = K.confusion_matrix_np(labels, logits, 3, ignore_classes=[0])
    assert cm.tolist() == [[0, 0, 0], [0, 1, 1], [0, 0, 2]]      # rows = truth, ignored label row empty
    assert abs(K.cfm_accuracy(cm) - 3 / 4) < 1e-12
    assert abs(K.cfm_mpca(cm, [0]) - (0.5 + 1.0) / 2) < 1e-12
    assert abs(K.cfm_miou(cm, [0]) - (1 / 2 + 2 / 3) / 2) < 1e-12
    assert abs(K.tf_mean_per_class_accuracy(cm) - (0 + 0.5 + 1.0) / 3) < 1e-12
    assert abs(K.tf_mean_iou(cm) - (1 / 2 + 2 / 3) / 2) < 1e-12
    assert abs(K.tf_mean_iou(cm, only_present=False) - (0 + 1 / 2 + 2 / 3) / 3) < 1e-12
    # argmax ties -> lowest index
    assert K.confusion_matrix_np(np.array([1]), np.array([[0.0, 3.0, 3.0]]), 3).tolist() == [[0, 0, 0], [0, 1, 0], [0, 0, 0]]


def test_bn_momentum_is_weight_of_old_value():
    assert abs(float(K.bn_moving_update(torch.tensor(1.0, dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64))) - 2.8) < 1e-15
    x = torch.tensor([[[[1.0], [3.0]]]], dtype=torch.float64)
    y, mean, var = K.batch_norm_train(x, torch.ones(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64))
    assert float(mean) == 2.0 and float(var) == 1.0          # biased variance
    assert np.allclose(y.numpy().ravel(), np.array([-1, 1]) / np.sqrt(1 + 1e-3))


def test_adam_first_step_matches_closed_form():
    p = [torch.tensor([1.0, -2.0], dtype=torch.float64)]
    g = [torch.tensor([0.5, -0.25], dtype=torch.float64)]
    m = [torch.zeros(2, dtype=torch.float64)]
    v = [torch.zeros(2, dtype=torch.float64)]
    K.adam_step(p, g, m, v, 0, lr=1e-3)
    # t = 1: m = 0.1 g, v = 0.001 g^2, lr_t = lr*sqrt(0.001)/0.1 -> step = lr * g/(|g| + eps*sqrt(1000)...)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp = np.array([1.0, -2.0]) - lr_t * (0.1 * np.array([0.5, -0.25])) / (np.sqrt(0.001 * np.array([0.25, 0.0625])) + 1e-8)
    assert np.allclose(p[0].numpy(), exp, rtol=0, atol=1e-15)
    # lr_scalers multiply the step of the indexed weight only
    p2 = [torch.ones(1, dtype=torch.float64), torch.ones(1, dtype=torch.float64)]
    g2 = [torch.ones(1, dtype=torch.float64), torch.ones(1, dtype=torch.float64)]
    K.adam_step(p2, g2, [torch.zeros(1, dtype=torch.float64)] * 2, [torch.zeros(1, dtype=torch.float64)] * 2, 0, lr=1e-3,
                lr_scalers={1: 0.5})
    assert abs((1 - float(p2[1])) / (1 - float(p2[0])) - 0.5) < 1e-12


def test_sgd_momentum_nesterov():
    p = [torch.tensor([1.0], dtype=torch.float64)]
    mom = [torch.tensor([0.5], dtype=torch.float64)]
    K.sgd_step(p, [torch.tensor([2.0], dtype=torch.float64)], mom, 0, lr=0.1, momentum=0.9)
    assert abs(float(mom[0]) - (0.9 * 0.5 - 0.2)) < 1e-15 and abs(float(p[0]) - (1 + 0.25)) < 1e-15
    p = [torch.tensor([1.0], dtype=torch.float64)]
    mom = [torch.tensor([0.5], dtype=torch.float64)]
    K.sgd_step(p, [torch.tensor([2.0], dtype=torch.float64)], mom, 0, lr=0.1, momentum=0.9, nesterov=True)
    assert abs(float(p[0]) - (1 + 0.9 * 0.25 - 0.2)) < 1e-15


def test_enet_parameter_counts_match_the_reference_graph():
    # trainable parameter counts derived from src/models.py
    for name, n in (("enet-naive-upsampling", 364972), ("enet-max-unpooling", 364924)):
        w = E.init_weights(name, 24)
        assert sum(v.size for k, v in w.items() if k.endswith(E.TRAINABLE_SUFFIXES)) == n
    spec = [k for k, _, _, _ in E.weight_spec("enet-naive-upsampling", 24)]
    assert spec[0] == "initial_block_conv2d/kernel" and spec[-1] == "logits/bias"
    assert len(set(spec)) == len(spec)


def test_enet_forward_shapes_and_inference_mode():
    name = "enet-max-unpooling"
    w = {k: torch.tensor(v) for k, v in E.init_weights(name, 5, perturb=True).items()}
    x = torch.randn(1, 32, 48, 3, dtype=torch.float64)
    logits, ctx = E.enet_forward(name, w, x, training=False)
    assert logits.shape == (1, 32, 48, 5) and not ctx.bn_stats
    logits2, ctx2 = E.enet_forward(name, w, x, training=True)
    assert len(ctx2.bn_stats) > 60 and not torch.allclose(logits, logits2)


def test_input_contract_known_answers():
    """src/generators.py:711-770, 1787-1825 restated: byte 0 -> -1, 255 -> +1; labeled rows map the red colour to the
    class id and take the class weight, unlabeled rows keep the raw value and weight 1."""
    from oracle import input_oracle as O
    img = np.array([[[[0, 255, 51]]]], dtype=np.uint8)
    x = O.normalize_image_batch(img)
    assert x.dtype == np.float32 and np.allclose(x[0, 0, 0], [-1.0, 1.0, -0.6], atol=1e-6)
    x2 = O.normalize_image_batch(img, [0.0, 0.5, 0.0], [0.5, 0.5, 1.0])
    assert np.allclose(x2[0, 0, 0], [-2.0, 1.0, -0.6], atol=1e-6)
    x3 = O.normalize_image_batch(img, [127.5, 127.5, 127.5], None)          # mean given in the [0, 255] range = 0 normalised
    assert np.allclose(x3, x, atol=1e-6)
    masks = np.array([[[10, 20]], [[7, 20]]], dtype=np.uint8)
    y = O.index_encode_batch_masks(masks, {10: 1, 20: 2}, num_labeled=1)
    assert np.array_equal(y, [[[1, 2]], [[7, 20]]])
    w = O.create_batch_weights(y, np.array([0.0, 0.25, 4.0], dtype=np.float32), num_labeled=1)
    assert np.array_equal(w, [[[0.25, 4.0]], [[1.0, 1.0]]])


def test_mean_teacher_consistency_and_loss_known_answers():
    """src/losses.py:195-215, 377-436 on values worked by hand: student softmax (1/2, 1/2), teacher (3/4, 1/4)."""
    y = torch.zeros(2, 1, 1, 2, dtype=torch.float64)
    t_ = torch.tensor([np.log(3.0), 0.0], dtype=torch.float64).expand(2, 1, 1, 2)
    c = K.mean_teacher_consistency_cost(y, t_, 2.0)
    assert abs(float(c) - 2.0 * 0.0625) < 1e-15
    # one labeled row (label 1, weight 3: CE = 3 ln 2, averaged over that single pixel), one unlabeled row
    lab = torch.tensor([[[1]], [[0]]])
    w = torch.tensor([[[3.0]], [[7.0]]], dtype=torch.float64)
    loss = K.segmentation_mean_teacher_lambda_loss(y, lab, w, 1, t_, 2.0)
    assert abs(float(loss) - (3.0 * np.log(2.0) + 0.125)) < 1e-14
    # gradient of the consistency cost w.r.t. the student's logits: 2*coef/(C*N) * s_j * (d_j - sum_c d_c s_c)
    yy = y.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(K.mean_teacher_consistency_cost(yy, t_, 2.0), [yy])
    d = np.array([-0.25, 0.25])
    exp = 2 * 2.0 / (2 * 2) * 0.5 * (d - (d * 0.5).sum())
    np.testing.assert_allclose(g[0, 0, 0].numpy(), exp, rtol=1e-14, atol=1e-16)
    # no gradient reaches the teacher's logits
    tt = t_.clone().requires_grad_(True)
    assert not K.mean_teacher_consistency_cost(y, tt, 2.0).requires_grad


def test_ema_update_known_answer():
    """src/trainers.py:1704 keeps float32: a*t + (1-a)*s with a Python float."""
    out = K.ema_update_np([np.array([1.0, -2.0], np.float32)], [np.array([2.0, 4.0], np.float32)], 0.75)
    assert out[0].dtype == np.float32 and np.array_equal(out[0], np.array([1.25, -0.5], np.float32))


def test_superpixel_cost_known_answers():
    """src/losses.py:108-192 on a 3x3 image with a uniform softmax (1/2, 1/2): interior responses vanish and are clipped to
    epsilon, the zero padding makes the border respond; bottom-right corner: Gx = Gy = 1.5 for both classes, so
    sqrt(2 * (1.5^2 + 1.5^2)) = 3; bottom-left corner: Gx = 1.5, Gy = -1.5 -> epsilon."""
    y = torch.zeros(1, 3, 3, 2, dtype=torch.float64)
    only = lambda yy, xx: torch.tensor(np.eye(9)[yy * 3 + xx].reshape(1, 3, 3).astype(np.int64))
    eps = K.SUPERPIXEL_EPSILON
    assert eps == 1e-6
    assert abs(float(K.unlabeled_superpixel_cost(only(2, 2), y, 9.0)) - 3.0) < 1e-12
    assert abs(float(K.unlabeled_superpixel_cost(only(2, 0), y, 9.0)) - np.sqrt(2 * (2.25 + eps ** 2))) < 1e-12
    assert abs(float(K.unlabeled_superpixel_cost(only(1, 1), y, 9.0)) - np.sqrt(4 * eps ** 2)) < 1e-15
    assert abs(float(K.unlabeled_superpixel_cost(only(0, 0), y, 9.0)) - np.sqrt(4 * eps ** 2)) < 1e-15
    # superpixel ids above 1 count as interior (clip to [0, 1]); the coefficient scales linearly
    assert abs(float(K.unlabeled_superpixel_cost(only(2, 2) * 7, y, 18.0)) - 6.0) < 1e-12
    # clipped responses pass no gradient: with only the top-left pixel unmasked every logit gradient is zero
    yy = y.clone().requires_grad_(True)
    (g,) = torch.autograd.grad(K.unlabeled_superpixel_cost(only(0, 0), yy, 9.0), [yy])
    assert float(g.abs().max()) == 0.0
    # a batch without unlabeled rows has no superpixel term (src/losses.py:484)
    lab = torch.zeros(1, 3, 3, dtype=torch.int64)
    w = torch.ones(1, 3, 3, dtype=torch.float64)
    assert abs(float(K.segmentation_superpixel_lambda_loss(y, lab, w, 0, 5.0)) - np.log(2.0)) < 1e-14


def test_classification_loss_known_answers():
    """src/losses.py:579-588: two samples, two classes.  Sample 0: softmax (1/2, 1/2), one-hot label on class 0 with weight 3
    -> 3 ln 2.  Sample 1: softmax of class 1 is ~1e-13, below the clip at 1e-6: the term is -ln(1e-6) * 2 and passes no
    gradient.  The loss is the mean over the two samples."""
    z = torch.tensor([[0.0, 0.0], [0.0, -30.0]], dtype=torch.float64, requires_grad=True)
    y = torch.tensor([[1.0, 0.0], [0.0, 1.0]], dtype=torch.float64)
    w = torch.tensor([[3.0, 5.0], [7.0, 2.0]], dtype=torch.float64)
    loss = K.classification_weighted_categorical_crossentropy(y, z, w)
    assert abs(float(loss.detach()) - 0.5 * (3.0 * np.log(2.0) + 2.0 * -np.log(1e-6))) < 1e-12
    (g,) = torch.autograd.grad(loss, [z])
    np.testing.assert_allclose(g[0].numpy(), [0.5 * 3.0 * (0.5 - 1.0), 0.5 * 3.0 * 0.5], rtol=1e-12)
    assert float(g[1].abs().max()) == 0.0


def test_encoder_only_oracle_shapes():
    for name, cin in (("enet-naive-upsampling-encoder-only", 128), ("segnet-encoder-only", 512)):
        w0 = E.init_weights(name, 7, seed=1)
        assert w0["logits/kernel"].shape == (cin, 7)
        lim = np.sqrt(6.0 / (cin + 7))
        assert np.abs(w0["logits/kernel"]).max() <= lim
        params = {k: torch.tensor(v) for k, v in w0.items()}
        with torch.no_grad():
            logits, _ = E.enet_forward(name, params, torch.zeros(2, 32, 32, 3, dtype=torch.float64), training=False)
        assert tuple(logits.shape) == (2, 7)


@pytest.mark.parametrize("name", ["segnet", "unet"])
def test_forcing_the_oracles_own_branches_changes_nothing(name):
    """U-Net / SegNet oracle with LeakyReLU signs and max-pool selections forced to the ones it takes by itself: loss and
    every gradient identical (the forcing hooks used by the GPU step-parity tests select, they do not alter)."""
    w0 = E.init_weights(name, 24, seed=15, perturb=True)
    im, lab, pw = E.synthetic_batch(2, 32, 32, 24, seed=1, block=8)
    masks = E.make_dropout_masks(name, 2, 3)
    r = E.train_step(name, w0, im, lab, pw, masks, collect=True)
    forced = {k: np.sign(v) for k, v in r["acts"].items() if k.endswith("_activation")}
    from oracle import unet_segnet_oracle as V
    for prefix, n, _ in V.topology(name)[0]:
        x = np.asarray(r["acts"]["{}_conv{}_dropout".format(prefix, n)])
        forced[prefix + "_pool/argmax"] = K.max_pool_with_argmax_np(x)[1]
    r2 = E.train_step(name, w0, im, lab, pw, masks, forced=forced)
    assert r2["loss"] == r["loss"]
    for k, g in r["grads"].items():
        assert np.array_equal(r2["grads"][k], g), k
    # a flipped sign does change the result (the hook is live)
    k0 = sorted(forced)[0]
    flipped = dict(forced)
    flipped[k0] = -forced[k0]
    assert E.train_step(name, w0, im, lab, pw, masks, forced=flipped)["loss"] != r["loss"]
