"""Shared parity cases (test infrastructure).

Each ``case_*`` function builds seeded inputs with ``oracle.gen``, runs ONE
implementation of the path (passed in as ``impl``: the imported reference when
``tools/gen_fixtures.py`` writes the goldens, the oracle restatement in the CPU
tests, the HIP modules in the GPU tests) and returns a dict of numpy arrays.
Comparing two such dicts is the parity test.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gen

SEED = 20240607


class Impl:
    """Bundle of constructors/functions with the reference's call shapes."""

    def __init__(self, Branch, ASPP, Decoder, ce_loss, Evaluator, DeepLab=None, BatchNorm=nn.BatchNorm2d,
                 interpolate=None, lr_groups=None, device="cpu", name="impl", PAM=None, CAM=None, DAMM=None,
                 DeepLabDAMM=None, focal_loss=None, class_weights=None, optimizer=None, scheduler=None,
                 dropout_masks=None):
        self.Branch = Branch          # (inplanes, planes, k, padding, dilation, BatchNorm)
        self.ASPP = ASPP              # (backbone, output_stride, BatchNorm)
        self.Decoder = Decoder        # (num_classes, backbone, BatchNorm)
        self.ce_loss = ce_loss        # (logit, target, weight) -> scalar tensor
        self.Evaluator = Evaluator    # (num_class)
        self.DeepLab = DeepLab        # () -> module (R101 / OS16 / K=21)
        self.BatchNorm = BatchNorm
        self.interpolate = interpolate or (
            lambda x, size: F.interpolate(x, size=size, mode="bilinear", align_corners=True))
        self.lr_groups = lr_groups    # (model) -> (1x params, 10x params)
        self.device = device
        self.name = name
        self.PAM = PAM                # (in_dim)
        self.CAM = CAM                # (in_dim)
        self.DAMM = DAMM              # (backbone, BatchNorm)
        self.DeepLabDAMM = DeepLabDAMM  # () -> module (R101 / OS16 / K=8)
        self.focal_loss = focal_loss  # (logit, target, weight) -> scalar tensor
        self.class_weights = class_weights  # (dataset name, iterable of {'label': tensor}, num_classes) -> [K] float64
        # (param groups, momentum=, weight_decay=, nesterov=) -> optimizer; None: torch.optim.SGD
        self.optimizer = optimizer
        # (mode, base_lr, num_epochs, iters_per_epoch) -> callable(optimizer, i, epoch, best_pred); None: the poly
        # formula written out in case_e2e
        self.scheduler = scheduler
        # (list of NCHW keep-masks) -> context manager under which the implementation's training-mode dropouts take
        # these masks in evaluation order; None: the implementation draws its own (the reference: torch's generator)
        self.dropout_masks = dropout_masks


def _t(a, device, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if grad:
        t.requires_grad_(True)
    return t


def _np(t):
    return t.detach().float().cpu().numpy()


def _load(module, seed, tag, device):
    sd = gen.fill_state_dict(module.state_dict(), seed, prefix=tag + "/")
    module.load_state_dict(sd)
    return module.to(device)


def _zero_dropout(module):
    for m in module.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0


def _gain(n, c=1, tag="gain"):
    """Per-image, per-channel scale in [0.5, 1.5) so that images differ the way real ones do.  Without it
    the image-pooling branch of ASPP sees near-identical pooled vectors, its batch norm divides by
    sqrt(var + eps) with var ~ eps, and the rounding noise of any implementation is amplified a thousandfold;
    with a per-image scale only, pooled vectors are collinear and that branch's weight gradient cancels to
    noise.  Both are ill-conditioned tests, not kernel properties.  (For the same reason the cases use >= 3
    images wherever that branch is differentiated: batch norm over two samples maps them to +-1 whatever
    their values, so its input gradient is pure cancellation.)"""
    if c == 1:
        return np.linspace(0.5, 1.5, n, dtype=np.float32).reshape(n, 1, 1, 1)
    return gen.uniform(SEED, tag, (n, c, 1, 1), 0.5, 1.5)


def _sample(a, count=4096):
    flat = np.asarray(a).reshape(-1)
    step = max(1, flat.size // count)
    return flat[::step][:count].copy()


# ----------------------------------------------------------- (1) branches ----
BRANCH_CFGS = [(1, 1), (3, 1), (3, 6), (3, 12), (3, 18), (3, 24), (3, 36)]


def case_branch(impl, k, dil, cin=24, cout=16, shape=(2, 17, 23)):
    n, h, w = shape
    tag = "branch_k%d_d%d" % (k, dil)
    pad = 0 if k == 1 else dil
    out = {}
    for mode in ("train", "eval"):
        m = _load(impl.Branch(cin, cout, k, pad, dil, impl.BatchNorm), SEED, tag, impl.device)
        m.train(mode == "train")
        x = _t(gen.normal(SEED, tag + "/x", (n, cin, h, w)), impl.device, grad=True)
        y = m(x)
        gy = _t(gen.normal(SEED, tag + "/gy", (n, cout, h, w)), impl.device)
        y.backward(gy)
        out[mode + "_y"] = _np(y)
        out[mode + "_dx"] = _np(x.grad)
        out[mode + "_dw"] = _np(m.atrous_conv.weight.grad)
        out[mode + "_dgamma"] = _np(m.bn.weight.grad)
        out[mode + "_dbeta"] = _np(m.bn.bias.grad)
        if mode == "train":
            out["train_running_mean"] = _np(m.bn.running_mean)
            out["train_running_var"] = _np(m.bn.running_var)
    return out


# ---------------------------------------------------------------- (2) ASPP ----
def case_aspp(impl, backbone="mobilenet", output_stride=16, shape=(4, 9, 11), inplanes=320):
    n, h, w = shape
    tag = "aspp_%s_os%d" % (backbone, output_stride)
    out = {}
    for mode in ("train", "eval"):
        m = _load(impl.ASPP(backbone, output_stride, impl.BatchNorm), SEED, tag, impl.device)
        _zero_dropout(m)
        m.train(mode == "train")
        x = _t(gen.relu_normal(SEED, tag + "/x", (n, inplanes, h, w)) * _gain(n, inplanes), impl.device, grad=True)
        y = m(x)
        gy = _t(gen.normal(SEED, tag + "/gy", tuple(y.shape)), impl.device)
        y.backward(gy)
        out[mode + "_y"] = _np(y)
        out[mode + "_dx"] = _sample(_np(x.grad), 16384)
        for name, p in m.named_parameters():
            g = _np(p.grad)
            out[mode + "_d/" + name] = g if g.size <= 4096 else _sample(g)
    return out


# ------------------------------------------------------------- (3) Decoder ----
def case_decoder(impl, num_classes=21, shape_x=(2, 5, 7), shape_low=(2, 17, 25)):
    tag = "decoder_k%d" % num_classes
    out = {}
    for mode in ("train", "eval"):
        m = _load(impl.Decoder(num_classes, "resnet", impl.BatchNorm), SEED, tag, impl.device)
        _zero_dropout(m)
        m.train(mode == "train")
        x = _t(gen.relu_normal(SEED, tag + "/x", (shape_x[0], 256) + shape_x[1:]) * _gain(shape_x[0], 256), impl.device,
               grad=True)
        low = _t(gen.relu_normal(SEED, tag + "/low", (shape_low[0], 256) + shape_low[1:]) * _gain(shape_low[0], 256, "gainlow"),
                 impl.device, grad=True)
        y = m(x, low)
        gy = _t(gen.normal(SEED, tag + "/gy", tuple(y.shape)), impl.device)
        y.backward(gy)
        out[mode + "_y"] = _np(y)
        out[mode + "_dx"] = _np(x.grad)
        out[mode + "_dlow"] = _sample(_np(low.grad), 16384)
        for name, p in m.named_parameters():
            g = _np(p.grad)
            out[mode + "_d/" + name] = g if g.size <= 4096 else _sample(g)
    return out


# ----------------------------------------------------- (3b) dropout switched on ----
DROPOUT_KINDS = ["aspp", "decoder"]
DROPOUT_SEED = 4711


def unpack_masks(golden):
    """The keep-masks a dropout fixture recorded, as bool tensors in evaluation order."""
    masks = []
    i = 0
    while "mask%d" % i in golden:
        shape = tuple(int(v) for v in golden["mask%d_shape" % i])
        bits = np.unpackbits(golden["mask%d" % i])[:int(np.prod(shape))]
        masks.append(torch.from_numpy(bits.reshape(shape).astype(np.bool_)))
        i += 1
    return masks


def case_dropout(impl, kind, masks=None):
    """ASPP (dropout 0.5 behind its projection, /root/reference/modeling/aspp.py:147) or the decoder (0.5 and 0.1
    inside last_conv, /root/reference/modeling/decoder.py:146,151) in training mode with dropout ON.
    masks=None: the implementation draws its own masks from torch's generator, seeded here (the reference, when the
    fixture is written); the draws are repeated afterwards on tensors of ones and recorded bit-packed in the result.
    Otherwise the implementation runs on the given masks (impl.dropout_masks)."""
    tag = "dropout_" + kind
    if kind == "aspp":
        n, h, w, inplanes = 4, 9, 11, 320
        m = _load(impl.ASPP("mobilenet", 16, impl.BatchNorm), SEED, tag, impl.device)
        inputs = [_t(gen.relu_normal(SEED, tag + "/x", (n, inplanes, h, w)) * _gain(n, inplanes), impl.device, grad=True)]
        drops = [((n, 256, h, w), m.dropout.p)]
        names = ["dx"]
    else:
        sx, sl = (2, 5, 7), (2, 17, 25)
        m = _load(impl.Decoder(21, "resnet", impl.BatchNorm), SEED, tag, impl.device)
        inputs = [_t(gen.relu_normal(SEED, tag + "/x", (sx[0], 256) + sx[1:]) * _gain(sx[0], 256), impl.device, grad=True),
                  _t(gen.relu_normal(SEED, tag + "/low", (sl[0], 256) + sl[1:]) * _gain(sl[0], 256, "gainlow"),
                     impl.device, grad=True)]
        drops = [((sl[0], 256) + sl[1:], m.last_conv[3].p), ((sl[0], 256) + sl[1:], m.last_conv[7].p)]
        names = ["dx", "dlow"]
    m.train()
    out = {}
    if masks is None:
        torch.manual_seed(DROPOUT_SEED)
        y = m(*inputs)
        # the same draws again: F.dropout of a tensor of ones is the keep-mask over (1 - p)
        torch.manual_seed(DROPOUT_SEED)
        for i, (shape, p) in enumerate(drops):
            keep = (F.dropout(torch.ones(shape), p, True) != 0).numpy()
            out["mask%d" % i] = np.packbits(keep.reshape(-1))
            out["mask%d_shape" % i] = np.array(shape, dtype=np.int64)
    else:
        with impl.dropout_masks(masks):
            y = m(*inputs)
    gy = _t(gen.normal(SEED, tag + "/gy", tuple(y.shape)), impl.device)
    y.backward(gy)
    out["y"] = _np(y)
    for nm, t in zip(names, inputs):
        out[nm] = _sample(_np(t.grad), 16384)
    for name, p in m.named_parameters():
        g = _np(p.grad)
        out["d/" + name] = g if g.size <= 4096 else _sample(g)
    return out


# --------------------------------------------------------- (4) interpolate ----
INTERP_CFGS = [((2, 3, 1, 1), (5, 7)), ((1, 2, 9, 11), (33, 41)), ((1, 2, 33, 33), (129, 129)),
               ((1, 1, 97, 97), (193, 193)), ((2, 3, 7, 5), (20, 13)), ((1, 2, 6, 6), (6, 6)),
               ((1, 2, 8, 8), (1, 1))]


def case_interp(impl, in_shape, size):
    tag = "interp_%s_%s" % ("x".join(map(str, in_shape)), "x".join(map(str, size)))
    x = _t(gen.normal(SEED, tag + "/x", in_shape), impl.device, grad=True)
    y = impl.interpolate(x, size)
    gy = _t(gen.normal(SEED, tag + "/gy", tuple(y.shape)), impl.device)
    y.backward(gy)
    return {"y": _np(y), "dx": _np(x.grad)}


# ---------------------------------------------------------------- (5) loss ----
def ce_inputs(name, K=21, n=2, h=13, w=17):
    tag = "ce_" + name
    logit = gen.normal(SEED, tag + "/logit", (n, K, h, w), std=3.0)
    target = gen.labels(SEED, tag + "/t", (n, h, w), K, border=1, ignore_frac=0.1)
    weight = None
    if name == "all_ignored":
        target[:] = 255
    elif name == "one_image_ignored":
        target[0] = 255
    elif name == "weighted":
        weight = gen.uniform(SEED, tag + "/w", (K,), 0.2, 2.0)
    elif name == "k81":
        pass
    return logit, target, weight


CE_CFGS = [("plain", 21), ("all_ignored", 21), ("one_image_ignored", 21), ("weighted", 21), ("k81", 81),
           ("k8", 8)]


def case_ce(impl, name, K):
    logit_np, target_np, weight_np = ce_inputs(name, K=K)
    logit = _t(logit_np, impl.device, grad=True)
    target = _t(target_np, impl.device)
    weight = None if weight_np is None else _t(weight_np, impl.device)
    loss = impl.ce_loss(logit, target, weight)
    loss.backward()
    return {"loss": _np(loss).reshape(1), "dlogit": _np(logit.grad)}


FOCAL_CFGS = [("plain", 21), ("one_image_ignored", 21), ("weighted", 21), ("k8", 8)]


def case_focal(impl, name, K):
    """SegmentationLosses(...).build_loss('focal') (/root/reference/utils/loss.py:37-53) on the cross-entropy
    inputs: value and gradient with respect to the logits."""
    logit_np, target_np, weight_np = ce_inputs(name, K=K)
    logit = _t(logit_np, impl.device, grad=True)
    target = _t(target_np, impl.device)
    weight = None if weight_np is None else _t(weight_np, impl.device)
    loss = impl.focal_loss(logit, target, weight)
    loss.backward()
    return {"loss": _np(loss).reshape(1), "dlogit": _np(logit.grad)}


def class_weight_batches(K, batches=3, n=2, h=40, w=33):
    """Label batches the way a loader hands them to calculate_weigths_labels
    (/root/reference/utils/calculate_weights.py:12-19): float maps with 255, negative and too-large labels, and one
    class that never occurs."""
    out = []
    for b in range(batches):
        tag = "cw_k%d/%d" % (K, b)
        y = gen.labels(SEED, tag, (n, h, w), K, border=1, ignore_frac=0.15)
        odd = gen.uniform(SEED, tag + "/odd", (n, h, w))
        y[odd < 0.02] = -1.0
        y[(odd >= 0.02) & (odd < 0.04)] = K + 3
        y[y == 2] = 5.0  # class 2 never occurs: weight 1 / ln(1.02)
        out.append(y)
    return out


CW_KS = [8, 21]


def case_class_weights(impl, K):
    loader = [{"label": torch.from_numpy(y)} for y in class_weight_batches(K)]
    return {"weights": np.asarray(impl.class_weights("synthetic", loader, K), dtype=np.float64)}


# ----------------------------------------------------------- (6) evaluator ----
EVAL_KS = [8, 19, 21, 81]


def eval_inputs(K, n=3, h=37, w=41):
    tag = "eval_k%d" % K
    gt = gen.labels(SEED, tag + "/gt", (n, h, w), K, border=2, ignore_frac=0.05)
    # out-of-range and negative labels must fall out through the mask
    odd = gen.uniform(SEED, tag + "/odd", (n, h, w))
    gt[odd < 0.01] = K
    gt[(odd >= 0.01) & (odd < 0.02)] = -1.0
    gt[(odd >= 0.02) & (odd < 0.03)] = K + 7
    logits = gen.normal(SEED, tag + "/logits", (n, K, h, w))
    # force exact ties: quantise a third of the pixels coarsely
    q = gen.uniform(SEED, tag + "/tie", (n, 1, h, w)) < 0.33
    logits = np.where(q, np.round(logits), logits).astype(np.float32)
    if K >= 19:
        logits[:, K - 2] = -5.0  # a class that is never predicted -> zero column
        gt[gt == 3] = 4          # a class that never occurs -> NaN row in the per-class ratios
    return gt, logits


def case_eval(impl, K):
    gt, logits = eval_inputs(K)
    ev = impl.Evaluator(K)
    pred = np.argmax(logits, axis=1)
    ev.add_batch(gt[:2], pred[:2])
    ev.add_batch(gt[2:], pred[2:])
    with np.errstate(all="ignore"):
        return {"cm": np.asarray(ev.confusion_matrix, dtype=np.float64), "pred": pred.astype(np.int64),
                "PA": np.array([ev.Pixel_Accuracy()]), "mPA": np.array([ev.Pixel_Accuracy_Class()]),
                "mIoU": np.array([ev.Mean_Intersection_over_Union()]),
                "FWIoU": np.array([ev.Frequency_Weighted_Intersection_over_Union()])}


# --------------------------------------------------------- (8) end to end ----
def case_e2e(impl, steps=2, n=4, size=65, K=21, base_lr=0.007, total_iters=100):
    """Two SGD steps of the whole model the way the trainers drive it
    (/root/reference/tain_origin.py:124-147; groups and hyper-parameters
    /root/reference/train.py:56-61): poly LR, zero_grad, forward, CE, backward, step."""
    tag = "e2e"
    model = _load(impl.DeepLab(), SEED, tag, impl.device)
    _zero_dropout(model)
    model.train()
    g1, g10 = impl.lr_groups(model)
    make_opt = impl.optimizer or torch.optim.SGD
    opt = make_opt([{"params": g1, "lr": base_lr}, {"params": g10, "lr": base_lr * 10}],
                   momentum=0.9, weight_decay=5e-4, nesterov=False)
    sched = impl.scheduler("poly", base_lr, 1, total_iters) if impl.scheduler else None
    names = ("aspp.aspp3.atrous_conv.weight", "aspp.conv1.weight", "decoder.last_conv.0.weight",
             "decoder.last_conv.8.bias", "aspp.bn1.weight", "aspp.bn1.running_var",
             "decoder.bn1.running_mean", "backbone.layer4.2.conv2.weight", "backbone.bn1.running_mean")
    out = {}
    losses = []
    for it in range(steps):
        image = _t(gen.normal(SEED, tag + "/img%d" % it, (n, 3, size, size)) * _gain(n), impl.device)
        target = _t(gen.labels(SEED, tag + "/lab%d" % it, (n, size, size), K, border=2), impl.device)
        if sched is not None:
            sched(opt, it, 0, 0.0)
        else:
            lr = base_lr * pow((1 - 1.0 * it / total_iters), 0.9)
            opt.param_groups[0]["lr"] = lr
            opt.param_groups[1]["lr"] = lr * 10
        opt.zero_grad()
        output = model(image)
        loss = impl.ce_loss(output, target, None)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach().cpu()))
        if it == 0:
            # the first step is a pure function of the inputs: its predictions and the weights it produces are
            # comparable tightly (later steps inherit the chaotic growth of last-bit differences)
            ev = impl.Evaluator(K)
            ev.add_batch(target.cpu().numpy(), np.argmax(_np(output), axis=1))
            out["cm0"] = np.asarray(ev.confusion_matrix, dtype=np.float64)
            out["logits0_sample"] = _sample(_np(output), 8192)
            sd = model.state_dict()
            for name in names:
                out["w0/" + name] = _sample(_np(sd[name]), 2048)
    out["losses"] = np.array(losses, dtype=np.float64)
    out["logits_sample"] = _sample(_np(output), 8192)
    ev = impl.Evaluator(K)
    ev.add_batch(target.cpu().numpy(), np.argmax(_np(output), axis=1))
    out["cm"] = np.asarray(ev.confusion_matrix, dtype=np.float64)
    sd = model.state_dict()
    for name in names:
        out["w/" + name] = _sample(_np(sd[name]), 2048)
    return out


# ------------------------------------------------- (9) full-size summaries ----
def case_fullsize(impl, config="A", n=3, backward=True, dropout=False):
    """Head only, at BASELINE shapes.  A: OS16 513^2 K=21 (x 33^2, low 129^2);
    D: OS8 769^2 K=19 (x 97^2, low 193^2).  dropout: the three dropouts stay at their probabilities (0.5, 0.5, 0.1)
    and take keep-masks from the portable generator (needs impl.dropout_masks)."""
    if config == "A":
        os_, h, l, H, K = 16, 33, 129, 513, 21
    else:
        os_, h, l, H, K = 8, 97, 193, 769, 19
    tag = "full" + config
    aspp = _load(impl.ASPP("resnet", os_, impl.BatchNorm), SEED, tag + "/aspp", impl.device)
    dec = _load(impl.Decoder(K, "resnet", impl.BatchNorm), SEED, tag + "/dec", impl.device)
    for m in (aspp, dec):
        if not dropout:
            _zero_dropout(m)
        m.train()
    x = _t(gen.relu_normal(SEED, tag + "/x", (n, 2048, h, h)) * _gain(n, 2048), impl.device, grad=backward)
    low = _t(gen.relu_normal(SEED, tag + "/low", (n, 256, l, l)) * _gain(n, 256, "gainlow"), impl.device, grad=backward)
    target = _t(gen.labels(SEED, tag + "/lab", (n, H, H), K, border=10), impl.device)
    if dropout:
        ps = (aspp.dropout.p, dec.last_conv[3].p, dec.last_conv[7].p)
        shapes = ((n, 256, h, h), (n, 256, l, l), (n, 256, l, l))
        masks = [torch.from_numpy(gen.uniform(SEED, tag + "/drop%d" % i, shp) >= p) for i, (shp, p) in enumerate(zip(shapes, ps))]
        with impl.dropout_masks(masks):
            a = aspp(x)
            d = dec(a, low)
    else:
        a = aspp(x)
        d = dec(a, low)
    logits = impl.interpolate(d, (H, H))
    loss = impl.ce_loss(logits, target, None)
    out = {"loss": _np(loss).reshape(1), "aspp_chan_sum": _np(a.sum(dim=(0, 2, 3))),
           "aspp_chan_sq": _np((a.float() ** 2).sum(dim=(0, 2, 3))), "aspp_sample": _sample(_np(a), 8192),
           "dec_chan_sum": _np(d.sum(dim=(0, 2, 3))), "dec_sample": _sample(_np(d), 8192),
           "logits_sample": _sample(_np(logits), 8192)}
    if backward:
        loss.backward()
        out["dx_sample"] = _sample(_np(x.grad), 8192)
        out["dlow_sample"] = _sample(_np(low.grad), 8192)
        out["d/aspp4.w"] = _sample(_np(aspp.aspp4.atrous_conv.weight.grad), 4096)
        out["d/aspp.conv1.w"] = _sample(_np(aspp.conv1.weight.grad), 4096)
        out["d/dec.last0.w"] = _sample(_np(dec.last_conv[0].weight.grad), 4096)
        out["d/dec.cls.b"] = _np(dec.last_conv[8].bias.grad)
        out["d/aspp.bn1.gamma"] = _np(aspp.bn1.weight.grad)
    return out


# ------------------------------------------------------ (10) dual attention ----
ATT_KINDS = ["pam", "cam"]


def _param_grads(m, out, prefix):
    for name, p in m.named_parameters():
        if p.grad is None:
            continue
        g = _np(p.grad)
        out[prefix + "d/" + name] = g if g.size <= 4096 else _sample(g)


def case_attention(impl, kind, C=64, shape=(3, 7, 9)):
    """One attention module alone (/root/reference/modeling/damm_cam_pam.py:22-48, :72-94) with gamma != 0
    (the reference initialises it to 0, which would make the module the identity)."""
    n, h, w = shape
    tag = "att_" + kind
    m = _load(impl.PAM(C) if kind == "pam" else impl.CAM(C), SEED, tag, impl.device)
    m.train()
    x = _t(0.5 * gen.normal(SEED, tag + "/x", (n, C, h, w)) * _gain(n, C), impl.device, grad=True)
    y = m(x)
    gy = _t(gen.normal(SEED, tag + "/gy", tuple(y.shape)), impl.device)
    y.backward(gy)
    out = {"y": _np(y), "dx": _np(x.grad)}
    _param_grads(m, out, "")
    return out


def case_damm(impl, backbone="ghostnet", shape=(3, 7, 9), inplanes=160, modes=("train", "eval")):
    """Both attention modules, their sum and the 1x1 + BN + ReLU behind it (damm_cam_pam.py:130-136)."""
    n, h, w = shape
    tag = "damm_" + backbone
    out = {}
    for mode in modes:
        m = _load(impl.DAMM(backbone, impl.BatchNorm), SEED, tag, impl.device)
        m.train(mode == "train")
        x = _t(0.5 * gen.relu_normal(SEED, tag + "/x", (n, inplanes, h, w)) * _gain(n, inplanes), impl.device, grad=True)
        y = m(x)
        gy = _t(gen.normal(SEED, tag + "/gy", tuple(y.shape)), impl.device)
        y.backward(gy)
        yn = _np(y)
        out[mode + "_y"] = yn if yn.size <= 65536 else _sample(yn, 16384)
        out[mode + "_dx"] = _sample(_np(x.grad), 16384)
        _param_grads(m, out, mode + "_")
        if mode == "train":
            out["train_running_var"] = _np(m.conv1[1].running_var)
    return out


def case_e2e_damm(impl, n=3, size=65, K=8):
    """Forward, loss and backward of the whole dual-attention DeepLab (/root/reference/modeling/deeplab_damm.py:39-51)."""
    tag = "e2e_damm"
    model = _load(impl.DeepLabDAMM(), SEED, tag, impl.device)
    _zero_dropout(model)
    model.train()
    image = _t(gen.normal(SEED, tag + "/img", (n, 3, size, size)) * _gain(n), impl.device)
    target = _t(gen.labels(SEED, tag + "/lab", (n, size, size), K, border=2), impl.device)
    output = model(image)
    loss = impl.ce_loss(output, target, None)
    loss.backward()
    out = {"loss": _np(loss).reshape(1), "logits_sample": _sample(_np(output), 8192)}
    params = dict(model.named_parameters())
    for name in ("damm.damm1.gamma", "damm.damm2.gamma", "damm.damm1.query_conv.weight", "damm.damm1.value_conv.bias",
                 "damm.conv1.0.weight", "damm_conv1.0.weight", "damm_conv1.1.weight", "aspp.conv1.weight",
                 "decoder.last_conv.8.bias", "backbone.layer4.2.conv2.weight", "backbone.conv1.weight"):
        out["d/" + name] = _sample(_np(params[name].grad), 2048)
    return out
