"""autograd.Function wrappers that chain the C-ABI kernels into the head's training step.

Data flow (all intermediate activations pixel-major [pixels, channels], dtype = config.compute_dtype()):

  ASPPFunction     x[N,Cin,h,w] -> 4 conv branches + pooled branch written straight into the slices of one
                   [M,5P] buffer -> 1x1 projection -> BN -> ReLU -> dropout                (modeling/aspp.py:131-147)
  DecoderFunction  low-level 1x1 + BN + ReLU and the upsampled ASPP output written into one [M,304(+16 zero)]
                   buffer -> 3x3 -> 3x3 -> classifier                                       (modeling/decoder.py:171-180)
  UpsampleFunction pixel-major logits -> NCHW fp32 at input size                            (modeling/deeplab.py:252)
  CrossEntropyFunction fused log-softmax + NLL with ignore index                            (utils/loss.py:23-35)
  SyncBNFunction   stand-alone batch norm for the feeder's layers                           (sync_batchnorm/batchnorm.py:48-78)

Every conv->BN->ReLU(->dropout) unit is: implicit-GEMM conv that also emits per-tile sum / sum-of-squares,
a tiny reduce (+ all-reduce across ranks), coefficient kernel, one fused apply pass.  Backward mirrors it:
reduce pass (sum g, sum g*xhat) (+ all-reduce), apply pass, weight-gradient GEMM, data-gradient GEMM.
"""
import torch

from . import config, ops, parallel
from ._lib import NCHW, PM

_BN_TYPES = (torch.nn.modules.batchnorm._BatchNorm,)

# Test hook: when set to a list, every fused conv->BN->ReLU unit appends (pixel-major output view, (N,H,W)) and
# every stand-alone batch norm with a fused ReLU appends (NCHW-shaped output, None), so a test can read the
# activation pattern the kernels produced.  None in normal operation.
CAPTURE = None

# Test hook: when set to a list of keep-masks (NCHW-shaped, non-zero = keep), every training-mode dropout with
# 0 < p < 1 takes the next one instead of drawing from torch's generator, in evaluation order (ASPP's, then the
# decoder's two), so that another implementation can be run on the same masks.  None in normal operation.
DROPOUT_MASKS = None


# ------------------------------------------------------------------ views ----
def pm_view_of(t):
    """2-D pixel-major view [N*H*W, C] of a 4-D tensor whose memory already is pixel-major
    (channels_last tensors, or views handed out by this module); None otherwise."""
    if t.dim() != 4:
        return None
    N, C, H, W = t.shape
    s = t.stride()
    if C > 1 and s[1] != 1:
        return None
    if W > 1:
        ld = s[3]
    elif H > 1:
        ld = s[2]
    elif N > 1:
        ld = s[0]
    else:
        ld = C
    if ld < C:
        return None
    if (W > 1 and s[3] != ld) or (H > 1 and s[2] != W * ld) or (N > 1 and s[0] != H * W * ld):
        return None
    return torch.as_strided(t, (N * H * W, C), (ld, 1), t.storage_offset())


def nchw_view_of(v, N, H, W):
    """NCHW-shaped strided view of a pixel-major 2-D view (no copy)."""
    ld, C = v.stride(0), v.shape[1]
    return torch.as_strided(v, (N, C, H, W), (H * W * ld, 1, W * ld, ld), v.storage_offset())


def to_pm(t, dtype):
    """Pixel-major [M, C] tensor of `dtype` holding the 4-D tensor t (view when possible, else one
    transposing kernel)."""
    v = pm_view_of(t)
    if v is not None and v.dtype == dtype:
        return v
    N, C, H, W = t.shape
    if v is not None:
        out = torch.empty((N * H * W, C), dtype=dtype, device=t.device)
        out.copy_(v)
        return out
    src = t
    if src.dtype != torch.float32 or not src.is_contiguous():
        src = src.float().contiguous()
    out = torch.empty((N * H * W, C), dtype=dtype, device=t.device)
    return ops.nchw_to_pm(src, out)


def grad_dtype_for(input_tensor):
    """dtype in which to produce the pixel-major gradient of an input: the input's own when its memory is
    pixel-major (the gradient is then returned as a view, no extra pass), else fp32 for the NCHW transpose."""
    if input_tensor.dtype in (torch.float32, torch.bfloat16) and pm_view_of(input_tensor) is not None:
        return input_tensor.dtype
    return torch.float32


def grad_like(input_tensor, g_pm, N, H, W):
    """Gradient for `input_tensor` from a pixel-major gradient: a strided NCHW view when the input was
    pixel-major memory of the same dtype, otherwise a contiguous NCHW tensor of the input's dtype."""
    if pm_view_of(input_tensor) is not None and g_pm.dtype == input_tensor.dtype:
        return nchw_view_of(g_pm, N, H, W)
    out = ops.pm_to_nchw(g_pm, N, H, W)
    return out if input_tensor.dtype == torch.float32 else out.to(input_tensor.dtype)


# --------------------------------------------------------------- BN helper ----
class _BNSpec:
    """What a batch-norm module contributes to a fused unit."""

    def __init__(self, bn):
        if not isinstance(bn, _BN_TYPES):
            raise TypeError("expected a batch-norm module, got %r" % type(bn))
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average batch norm (momentum=None) is not on this path")
        self.eps = float(bn.eps)
        self.momentum = float(bn.momentum)
        self.training = bool(bn.training) or bn.running_mean is None
        self.running_mean = bn.running_mean
        self.running_var = bn.running_var
        from .modeling.sync_batchnorm.batchnorm import _SynchronizedBatchNorm
        self.sync = isinstance(bn, _SynchronizedBatchNorm) and parallel.is_distributed()


def _global_sums(local, sync):
    """All-reduce (SUM) the [2C+1] statistics vector in place across ranks; identity for unsynchronised layers."""
    if sync:
        parallel.all_reduce_sum_(local)
        if config.debug_sync_stats():
            parallel.assert_same_on_all_ranks(local, "batch-norm statistics")
    return local


def _train_coeffs(partials, count, spec, gamma, beta):
    """Partial statistic rows of this rank -> coefficient rows [mean, inv_std, scale, shift]; running statistics are
    updated.  Synchronised layers exchange the [2C+1] vector in between, others finish in one launch."""
    _check_count(count * (parallel.world_size() if spec.sync else 1))
    if not spec.sync:
        return ops.bn_reduce_coeffs(partials, count, gamma, beta, spec.eps, spec.momentum, spec.running_mean,
                                    spec.running_var)
    sums = _global_sums(ops.bn_reduce(partials, count), True)
    return ops.bn_coeffs(sums, gamma, beta, spec.eps, spec.momentum, spec.running_mean, spec.running_var,
                         legacy_clamp=config.legacy_clamp_eps())


def _check_count(sums_host_count):
    if sums_host_count <= 1:
        raise ValueError("Expected more than 1 value per channel when training (batch-norm statistics)")


class _UnitRecord:
    __slots__ = ("x", "raw", "coef", "mask", "mask_scale", "relu", "training", "sync", "geom", "R", "dil", "Cout",
                 "Cin", "weight")


def _dropout_mask(shape, p, device, geom=None):
    """(uint8 keep-mask of `shape` = [pixels, channels] or None, scale of the kept elements)."""
    if p <= 0.0:
        return None, 1.0
    if p >= 1.0:
        return torch.zeros(shape, dtype=torch.uint8, device=device), 0.0
    if DROPOUT_MASKS is not None:
        given = DROPOUT_MASKS.pop(0)
        N, H, W = geom
        if tuple(given.shape) != (N, shape[1], H, W):
            raise ValueError("injected dropout mask has shape %s, expected %s" % (tuple(given.shape), (N, shape[1], H, W)))
        keep = (given.to(device) != 0).permute(0, 2, 3, 1).reshape(shape)
        return keep.to(torch.uint8).contiguous(), 1.0 / (1.0 - p)
    if torch.device(device).type == "cuda":
        return ops.dropout_mask(shape, p, device), 1.0 / (1.0 - p)
    keep = torch.rand(shape, device=device) >= p
    return keep.to(torch.uint8), 1.0 / (1.0 - p)


def _raw_dtype(x, M):
    """dtype of a head unit's pre-normalisation output: bf16 for bf16 operands (as the feeder's units and PyTorch's
    autocast keep it; the statistics still come from the fp32 accumulators), fp32 for fp32 operands (the image-pooling
    branch has them in either mode) and under config.set_head_raw_fp32."""
    if x.dtype == torch.bfloat16 and not config.head_raw_fp32():
        return torch.bfloat16
    return torch.float32


def unit_forward(x, weight, R, dil, geom, spec, gamma, beta, out, relu=True, drop_p=0.0, drop_training=True):
    """conv(x) -> BN -> ReLU -> dropout into `out` (a pixel-major view).  x: [M, >=Ck] view whose padding
    channels are zero.  Returns the record the backward needs."""
    N, H, W = geom
    M = N * H * W
    Cout, Cin = weight.shape[0], weight.shape[1]
    wp = ops.pack_weight(weight.detach(), x.dtype)
    # (see _raw_dtype: with config.set_head_raw_fp32 the pre-normalisation output stays in the accumulator's precision,
    # statistics and normalised values then refer to the same numbers and bf16 mode rounds once per layer, not twice)
    raw = torch.empty((M, Cout), dtype=_raw_dtype(x, M), device=x.device)
    stats = ops.conv_fwd([(x, wp, R, dil)], N, H, W, Cout, raw, want_stats=spec.training, true_ck=[Cin])
    return unit_finish(x, raw, stats, weight, R, dil, geom, spec, gamma, beta, out, relu, drop_p, drop_training)


def unit_finish(x, raw, stats, weight, R, dil, geom, spec, gamma, beta, out, relu=True, drop_p=0.0,
                drop_training=True):
    """Second half of a unit: batch-norm coefficients from the conv epilogue's statistic partials, then one fused
    apply pass (BN, ReLU, dropout) from the fp32 conv output `raw` into `out`."""
    N, H, W = geom
    M = N * H * W
    Cout, Cin = weight.shape[0], weight.shape[1]
    if spec.training:
        coef = _train_coeffs(stats, float(M), spec, gamma, beta)
    else:
        coef = ops.bn_coeffs_eval(spec.running_mean, spec.running_var, gamma, beta, spec.eps)
    mask, mask_scale = (None, 1.0)
    if drop_training and drop_p > 0.0:
        mask, mask_scale = _dropout_mask((M, Cout), drop_p, x.device, geom)
    ops.bn_apply(raw, out, coef, PM, relu, mask, mask_scale)
    if CAPTURE is not None:
        CAPTURE.append((out, geom))
    rec = _UnitRecord()
    rec.x, rec.raw, rec.coef, rec.mask, rec.mask_scale = x, raw, coef, mask, mask_scale
    rec.relu, rec.training, rec.sync, rec.geom = relu, spec.training, spec.sync, geom
    rec.R, rec.dil, rec.Cout, rec.Cin, rec.weight = R, dil, Cout, Cin, weight
    return rec


# ---- weight gradients off the data-gradient chain (config.set_overlap_wgrad) -----------------------------------
_SIDE_STREAMS = {}   # device index -> the second stream (config.set_wgrad_second_stream)
_PENDING = []        # (weight, dW, main stream, side stream, tensors the side stream reads, backward pass id)
_GROUPS = {}         # shape key -> _Group: gradients of one shape waiting for a launch together (config.group_wgrad)
_CLOCK = [0, None]   # [weight_grad calls on the deferred path, backward pass id of everything queued]
GROUP_IDLE = 8       # a group is launched once this many deferred gradients have passed without one of its shape


class _Group:
    __slots__ = ("items", "last", "args")

    def __init__(self, args):
        self.items, self.last, self.args = [], 0, args


def _backward_pass_id():
    """Identifies the running backward pass (autograd's graph task), -1 outside of one."""
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def _launch_deferred(entries, args, pass_id):
    """Start the weight gradients of `entries` = [(weight, x, dy, like, main, side)] (one shape, one device) on
    `side` (the launching stream itself, or the second stream), as one launch when there are several, and queue them
    for the join."""
    N, H, W, R, dil, Cout, Cin = args
    main, side = entries[0][4], entries[0][5]
    if side is not main:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        if len(entries) == 1:
            weight, x, dy, like = entries[0][:4]
            dws = [ops.conv_wgrad(x, dy, N, H, W, R, dil, Cout, Cin, like=like)]
        else:
            dws = ops.conv_wgrad_group([e[1] for e in entries], [e[2] for e in entries], N, H, W, R, dil, Cout, Cin,
                                       like=entries[0][3])
        for i, (weight, x, dy, like, _, _) in enumerate(entries):
            if like is None and (dws[i].dtype != weight.dtype or dws[i].stride() != weight.stride()):
                dws[i] = torch.empty_like(weight).copy_(dws[i])   # what autograd would have done on accumulation
    # x and dy stay referenced until the join, so the allocator cannot hand their memory to later launches of the
    # main stream while the second stream still reads them
    for (weight, x, dy, like, _, _), dw in zip(entries, dws):
        _PENDING.append((weight, dw, main, side, (x, dy), pass_id))


def _flush_groups(only_idle):
    for key in list(_GROUPS):
        g = _GROUPS[key]
        if only_idle and _CLOCK[0] - g.last < GROUP_IDLE and len(g.items) < ops.WGRAD_GROUP_MAX:
            continue
        del _GROUPS[key]
        _launch_deferred(g.items, g.args, _CLOCK[1])


def _join_weight_grads():
    """End-of-backward callback: the groups still waiting are launched, the launching stream waits for the second one,
    then every deferred gradient is stored where autograd's accumulation node would have put it."""
    _flush_groups(only_idle=False)
    pending = list(_PENDING)
    del _PENDING[:]
    joined = set()
    for weight, dw, main, side, _, _ in pending:
        if side is not main and (main, side) not in joined:
            main.wait_stream(side)
            joined.add((main, side))
    for weight, dw, main, side, _, _ in pending:
        with torch.cuda.stream(main), torch.no_grad():
            if weight.grad is None:
                weight.grad = dw
            else:
                weight.grad.add_(dw)


def weight_grad(weight, x, dy, N, H, W, R, dil, Cout, Cin, like=None):
    """dW of one convolution.  Returns the gradient, or None when it was started (or queued for a launch together with
    others of its shape) on the second stream and will be stored in weight.grad at the end of the running backward
    pass (config.overlap_wgrad, config.group_wgrad)."""
    deferred = (config.overlap_wgrad() and not parallel.grad_hooks_active() and isinstance(weight, torch.Tensor)
                and weight.is_leaf and weight.requires_grad and x.is_cuda)
    if not deferred:
        return ops.conv_wgrad(x, dy, N, H, W, R, dil, Cout, Cin, like=like)
    main = torch.cuda.current_stream(x.device)
    side = main
    if config.wgrad_second_stream():
        side = _SIDE_STREAMS.get(x.device.index)
        if side is None:
            side = _SIDE_STREAMS[x.device.index] = torch.cuda.Stream(device=x.device)
    # (gradients left over from a backward pass that raised before its end-of-pass callback belong to no parameter
    # any more: they are dropped, not added to this pass's)
    pass_id = _backward_pass_id()
    if _CLOCK[1] != pass_id:
        del _PENDING[:]
        _GROUPS.clear()
        _CLOCK[1] = pass_id
    _CLOCK[0] += 1
    # one callback per deferred gradient: the first to run joins and stores them all, the others find nothing left
    # (a callback queued only "when the list was empty" would be lost for good after a backward pass that raised)
    torch.autograd.Variable._execution_engine.queue_callback(_join_weight_grads)
    args = (N, H, W, R, dil, Cout, Cin)
    entry = (weight, x, dy, like, main, side)
    if config.group_wgrad() and x.dim() == 2 and dy.dim() == 2 and x.stride(1) == 1 and dy.stride(1) == 1 \
            and ops.wgrad_group_pays(N * H * W, x.shape[1], dy.shape[1], R, x.dtype) and dy.dtype == x.dtype:
        layout = (weight.dtype, weight.stride()) if like is None else (like.dtype, like.stride(), "like")
        key = (x.device.index, main.cuda_stream, args, x.shape[1], x.stride(0), dy.shape[1], dy.stride(0), layout)
        g = _GROUPS.get(key)
        if g is None:
            g = _GROUPS[key] = _Group(args)
        g.items.append(entry)
        g.last = _CLOCK[0]
    else:
        _launch_deferred([entry], args, pass_id)
    _flush_groups(only_idle=True)
    return None


def unit_backward(rec, dz, need_dw=True, dw_like=None, partials=None):
    """dz: pixel-major gradient w.r.t. the unit's output.  Returns (dy [M, Cout rounded up to 32] with zero
    padding = gradient w.r.t. the conv output, dW, dgamma, dbeta).  dw_like: a weight whose dtype and strides dW
    takes (default: contiguous fp32).  partials: the backward statistic rows of dz when the launch that produced dz
    has emitted them already (StatLink)."""
    N, H, W = rec.geom
    M = N * H * W
    C = rec.Cout
    cnt = float(M)
    if partials is None:
        partials, cnt = ops.bn_bwd_stat_partials(dz, rec.raw, rec.coef, PM, rec.relu, rec.mask, rec.mask_scale)
    local, dgb = ops.bn_reduce(partials, cnt, want_f32=True)
    dbeta, dgamma = dgb[0], dgb[1]
    gs = _global_sums(local, rec.sync and rec.training)
    cpad = ops.round_up(C, 32)
    alloc = torch.zeros if cpad != C else torch.empty
    dy = alloc((M, cpad), dtype=dz.dtype, device=dz.device)
    ops.bn_bwd_apply(dz, rec.raw, dy[:, :C] if cpad != C else dy, rec.coef, gs, PM, rec.relu, rec.mask, rec.mask_scale,
                     eval_mode=not rec.training)
    dw = None
    if need_dw:
        c16 = ops.round_up(C, 16)
        dw = weight_grad(rec.weight, rec.x, dy[:, :c16] if c16 != cpad else dy, N, H, W, rec.R, rec.dil, C, rec.Cin,
                         like=dw_like)
    return dy, dw, dgamma, dbeta


def dgrad(items, geom, Cin, out_dtype, img_bias=None, addend=None, addend_bits=None, stat_link=None):
    """Data gradient of one or several convolutions that share their input: items = [(dy, weight, R, dil)].
    Returns a pixel-major [M, Cin rounded up to 4] tensor of out_dtype.  addend (a pixel-major view of out_dtype, or
    None): another gradient of the same input, added in the convolution's epilogue when the kernels can
    (ops.conv_addend_ok) and by a separate addition otherwise - the same bits either way.  addend_bits (uint8
    [M, Cin/8]): only the elements of the addend whose bit is set are added.  stat_link (StatLink): the result is the
    output gradient of the batch norm that offered the link; the launch emits that layer's backward statistics when it
    can and leaves them in the link."""
    N, H, W = geom
    segs = []
    for dy, weight, R, dil in items:
        segs.append((dy, ops.pack_weight(weight.detach(), dy.dtype, transpose=True), R, dil))
    cpad = ops.round_up(Cin, 4)
    out = torch.empty((N * H * W, cpad), dtype=out_dtype, device=items[0][0].device)
    fused = addend is not None and img_bias is None and ops.conv_addend_ok(segs, Cin, out_dtype, addend)
    req = None
    if (stat_link is not None and stat_link.offered() and img_bias is None and cpad == Cin and addend is None
            and config.fuse_bwd_stats()):
        req = stat_link.request()
    ops.conv_fwd(segs, N, H, W, Cin, out, img_bias=img_bias, true_ck=[it[1].shape[0] for it in items],
                 addend=addend if fused else None, addend_bits=addend_bits if fused else None, bwd_stats=req)
    if req is not None and req.rows is not None:
        stat_link.fulfil(req.rows, out)
    res = out[:, :Cin] if cpad != Cin else out
    if addend is not None and not fused:
        extra = addend[:, :Cin]
        if addend_bits is not None:
            extra = extra * ops.unpack_bits(addend_bits, Cin).to(extra.dtype)
        res = res + extra
    return res


def _padded_input(v, mult=32):
    """Pixel-major tensor whose row length is a multiple of `mult` with zero padding (copy only if needed)."""
    M, C = v.shape
    if C % mult == 0:
        return v
    buf = torch.zeros((M, ops.round_up(C, mult)), dtype=v.dtype, device=v.device)
    buf[:, :C].copy_(v)
    return buf


# ------------------------------------------------------------ ASPP branch ----
class BranchFunction(torch.autograd.Function):
    """One `_ASPPModule`: conv -> BN -> ReLU (modeling/aspp.py:17-21)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mod):
        cd = config.compute_dtype()
        N, _, H, W = x.shape
        conv = mod.atrous_conv
        R, dil = conv.kernel_size[0], conv.dilation[0]
        xp = _padded_input(to_pm(x, cd))
        out = torch.empty((N * H * W, weight.shape[0]), dtype=cd, device=x.device)
        rec = unit_forward(xp, weight, R, dil, (N, H, W), _BNSpec(mod.bn), gamma, beta, out, relu=True)
        ctx.rec, ctx.x = rec, x
        return nchw_view_of(out, N, H, W)

    @staticmethod
    def backward(ctx, g):
        rec, x = ctx.rec, ctx.x
        N, H, W = rec.geom
        dy, dw, dgamma, dbeta = unit_backward(rec, to_pm(g, rec.x.dtype))
        dxp = dgrad([(dy, rec.weight, rec.R, rec.dil)], rec.geom, rec.Cin, grad_dtype_for(x))
        return grad_like(x, dxp, N, H, W), dw, dgamma, dbeta, None


# ------------------------------------------------------------------- ASPP ----
class ASPPFunction(torch.autograd.Function):
    """Parameters arrive flat: x, then for each of aspp1..4, pool, proj: (conv weight, gamma, beta)."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        cd = config.compute_dtype()
        N, Cin, H, W = x.shape
        M = N * H * W
        geom = (N, H, W)
        branches = [mod.aspp1, mod.aspp2, mod.aspp3, mod.aspp4]
        P = mod.conv1.out_channels
        xp = _padded_input(to_pm(x, cd))
        cat = torch.empty((M, 5 * P), dtype=cd, device=x.device)
        recs = []
        convs = [br.atrous_conv for br in branches]
        dils = [c.dilation[0] for c in convs]
        ladder = (all(c.kernel_size[0] == 3 and c.out_channels == P for c in convs[1:])
                  and dils[2] - dils[1] == dils[3] - dils[2] and all(br.bn.training == branches[1].bn.training
                                                                     for br in branches[1:]))
        for b, br in enumerate(branches):
            if ladder and b >= 1:
                break
            w, g, be = params[3 * b:3 * b + 3]
            conv = br.atrous_conv
            recs.append(unit_forward(xp, w, conv.kernel_size[0], conv.dilation[0], geom, _BNSpec(br.bn), g, be,
                                     cat[:, b * P:(b + 1) * P], relu=True))
        if ladder:
            # the three dilated 3x3 branches read the same input: one launch (three times the workgroups, which
            # 274 tiles per branch need to fill 256 CUs), outputs side by side in one fp32 buffer
            ws = [params[3 * b] for b in (1, 2, 3)]
            Ck = ops.round_up(Cin, 32)
            # the three operands side by side in a buffer the module keeps: weights registered with
            # ops.cache_weight_packs (segk.optim.SGD does that) are then packed once and renewed by the optimizer's
            # launch, like every other kept operand
            wp3 = getattr(mod, "_segk_ladder_pack", None)
            if wp3 is None or tuple(wp3.shape) != (3, 9, P, Ck) or wp3.dtype != cd or wp3.device != x.device:
                if torch.cuda.is_current_stream_capturing():
                    wp3 = torch.empty((3, 9, P, Ck), dtype=cd, device=x.device)
                else:
                    wp3 = mod._segk_ladder_pack = torch.empty((3, 9, P, Ck), dtype=cd, device=x.device)
            keep = wp3 is getattr(mod, "_segk_ladder_pack", None)
            for i, w in enumerate(ws):
                ops.pack_weight(w.detach(), cd, out=wp3[i], keep=keep)
            raw3 = torch.empty((M, 3 * P), dtype=_raw_dtype(xp, M), device=x.device)
            training = branches[1].bn.training or branches[1].bn.running_mean is None
            stats3 = ops.conv_fwd([(xp, wp3, 3, dils[1])], N, H, W, P, raw3[:, :P], want_stats=training,
                                  true_ck=[Cin], batch=3, dil_step=dils[2] - dils[1], out_batch=P)
            for i, b in enumerate((1, 2, 3)):
                w, g, be = params[3 * b:3 * b + 3]
                recs.append(unit_finish(xp, raw3[:, i * P:(i + 1) * P], stats3[i] if training else None, w, 3, dils[b],
                                        geom, _BNSpec(branches[b].bn), g, be, cat[:, b * P:(b + 1) * P], relu=True))
        # image-pooling branch: mean over pixels -> 1x1 conv -> BN -> ReLU -> broadcast
        w, g, be = params[12:15]
        # (kept in fp32 whatever the compute dtype: N rows only, and its batch norm sees just N samples per
        # channel, which amplifies operand rounding)
        pooled = _padded_input(ops.pool_sum(xp, N, H * W, 1.0 / (H * W)))
        p5 = torch.empty((N, P), dtype=torch.float32, device=x.device)
        rec5 = unit_forward(pooled, w, 1, 1, (N, 1, 1), _BNSpec(mod.global_avg_pool[2]), g, be, p5, relu=True)
        ops.upsample_fwd(p5, N, P, 1, 1, cat[:, 4 * P:5 * P], H, W, PM)
        # projection
        w, g, be = params[15:18]
        out = torch.empty((M, P), dtype=cd, device=x.device)
        recp = unit_forward(cat, w, 1, 1, geom, _BNSpec(mod.bn1), g, be, out, relu=True, drop_p=mod.dropout.p,
                            drop_training=mod.dropout.training)
        ctx.recs, ctx.rec5, ctx.recp, ctx.x, ctx.P, ctx.geom = recs, rec5, recp, x, P, geom
        return nchw_view_of(out, N, H, W)

    @staticmethod
    def backward(ctx, gout):
        recs, rec5, recp, x, P = ctx.recs, ctx.rec5, ctx.recp, ctx.x, ctx.P
        N, H, W = ctx.geom
        cd = recp.x.dtype
        grads = [None] * 18
        dyp, dwp, dgp, dbp = unit_backward(recp, to_pm(gout, cd))
        grads[15:18] = [dwp, dgp, dbp]
        dcat = dgrad([(dyp, recp.weight, 1, 1)], ctx.geom, 5 * P, cd)
        # pooled branch: adjoint of the broadcast is a sum over pixels
        d5 = ops.pool_sum(dcat[:, 4 * P:5 * P], N, H * W, 1.0)
        dy5, dw5, dg5, db5 = unit_backward(rec5, d5)
        grads[12:15] = [dw5, dg5, db5]
        dpooled = dgrad([(dy5, rec5.weight, 1, 1)], (N, 1, 1), rec5.Cin, torch.float32)
        img_bias = (dpooled * (1.0 / (H * W))).contiguous()
        items = []
        for b, rec in enumerate(recs):
            dy, dw, dg, db = unit_backward(rec, dcat[:, b * P:(b + 1) * P])
            grads[3 * b:3 * b + 3] = [dw, dg, db]
            items.append((dy, rec.weight, rec.R, rec.dil))
        if ctx.needs_input_grad[1]:
            dxp = dgrad(items, ctx.geom, recs[0].Cin, grad_dtype_for(x), img_bias=img_bias)
            dx = grad_like(x, dxp, N, H, W)
        else:
            dx = None
        return (None, dx) + tuple(grads)


# ---------------------------------------------------------------- Decoder ----
class DecoderFunction(torch.autograd.Function):
    """params: conv1.w, bn1.g, bn1.b, last0.w, last1.g, last1.b, last4.w, last5.g, last5.b, last8.w, last8.b"""

    @staticmethod
    def forward(ctx, mod, x, low, *params):
        cd = config.compute_dtype()
        N, Cx, h, w = x.shape
        _, Cl, H, W = low.shape
        M = N * H * W
        geom = (N, H, W)
        lc = mod.last_conv
        C48 = mod.conv1.out_channels
        Ccat = Cx + C48
        Ck = ops.round_up(Ccat, 32)
        cat = torch.empty((M, Ck), dtype=cd, device=x.device)
        if Ck != Ccat:
            cat[:, Ccat:].zero_()
        xp = to_pm(x, cd)
        ops.upsample_fwd(xp, N, Cx, h, w, cat[:, :Cx], H, W, PM)
        lowp = _padded_input(to_pm(low, cd))
        rec1 = unit_forward(lowp, params[0], 1, 1, geom, _BNSpec(mod.bn1), params[1], params[2], cat[:, Cx:Ccat],
                            relu=True)
        P = lc[0].out_channels
        a1 = torch.empty((M, P), dtype=cd, device=x.device)
        recA = unit_forward(cat, params[3], 3, lc[0].dilation[0], geom, _BNSpec(lc[1]), params[4], params[5], a1,
                            relu=True, drop_p=lc[3].p, drop_training=lc[3].training)
        a2 = torch.empty((M, lc[4].out_channels), dtype=cd, device=x.device)
        recB = unit_forward(a1, params[6], 3, lc[4].dilation[0], geom, _BNSpec(lc[5]), params[7], params[8], a2,
                            relu=True, drop_p=lc[7].p, drop_training=lc[7].training)
        K = lc[8].out_channels
        Kpad = ops.round_up(K, 32)
        logits = torch.zeros((M, Kpad), dtype=torch.float32, device=x.device)  # small: keep the scores fp32
        bias = params[10]
        a2p = _padded_input(a2)
        ops.conv_fwd([(a2p, ops.pack_weight(params[9].detach(), cd), 1, 1)], N, H, W, K,
                     logits[:, :ops.round_up(K, 4)], bias=None if bias is None else bias.detach().float().contiguous(),
                     true_ck=[a2.shape[1]])
        ctx.rec1, ctx.recA, ctx.recB, ctx.a2p = rec1, recA, recB, a2p
        ctx.x, ctx.low, ctx.geom, ctx.hw, ctx.K, ctx.Cx, ctx.wcls = x, low, geom, (h, w), K, Cx, params[9]
        ctx.has_bias = bias is not None
        return nchw_view_of(logits[:, :K], N, H, W)

    @staticmethod
    def backward(ctx, gout):
        rec1, recA, recB = ctx.rec1, ctx.recA, ctx.recB
        N, H, W = ctx.geom
        h, w = ctx.hw
        K, Cx = ctx.K, ctx.Cx
        cd = recA.x.dtype
        M = N * H * W
        Kpad = ops.round_up(K, 32)
        # own zero-padded copy of the incoming gradient (its padding is not ours to trust)
        gv = pm_view_of(gout)
        dyc = torch.zeros((M, Kpad), dtype=cd, device=gout.device)
        if gv is not None:
            dyc[:, :K].copy_(gv)
        else:
            ops.nchw_to_pm(gout.float().contiguous(), dyc[:, :K])
        Pb = recB.Cout
        dw8 = weight_grad(ctx.wcls, ctx.a2p, dyc, N, H, W, 1, 1, K, Pb)
        db8 = ops.bn_local_sums(dyc, PM)[:K].float() if ctx.has_bias else None
        da2 = dgrad([(dyc, ctx.wcls, 1, 1)], ctx.geom, Pb, cd)
        dyB, dwB, dgB, dbB = unit_backward(recB, da2)
        da1 = dgrad([(dyB, recB.weight, recB.R, recB.dil)], ctx.geom, recB.Cin, cd)
        dyA, dwA, dgA, dbA = unit_backward(recA, da1)
        Ccat = recA.Cin
        dcat = dgrad([(dyA, recA.weight, recA.R, recA.dil)], ctx.geom, Ccat, cd)
        dy1, dw1, dg1, db1 = unit_backward(rec1, dcat[:, Cx:Ccat])
        dlow = dx = None
        if ctx.needs_input_grad[2]:
            dlowp = dgrad([(dy1, rec1.weight, 1, 1)], ctx.geom, rec1.Cin, grad_dtype_for(ctx.low))
            dlow = grad_like(ctx.low, dlowp, N, H, W)
        if ctx.needs_input_grad[1]:
            dxp = torch.empty((N * h * w, Cx), dtype=cd, device=gout.device)
            ops.upsample_bwd(dcat[:, :Cx], PM, N, Cx, H, W, dxp, Cx, h, w)
            if pm_view_of(ctx.x) is not None and ctx.x.dtype == cd:
                dx = nchw_view_of(dxp, N, h, w)
            else:
                dx = ops.pm_to_nchw(dxp, N, h, w).to(ctx.x.dtype)
        return (None, dx, dlow, dw1, dg1, db1, dwA, dgA, dbA, dwB, dgB, dbB, dw8, db8)


# --------------------------------------------------------------- upsample ----
class UpsampleFunction(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=True) -> contiguous NCHW fp32."""

    @staticmethod
    def forward(ctx, x, size):
        N, C, Hi, Wi = x.shape
        Ho, Wo = int(size[0]), int(size[1])
        v = pm_view_of(x)
        if v is None or v.stride(0) % 4 != 0 or v.stride(0) < ops.round_up(C, 4) or v.data_ptr() % 16 != 0:
            buf = torch.zeros((N * Hi * Wi, ops.round_up(C, 4)), dtype=x.dtype if x.dtype == torch.bfloat16 else torch.float32,
                              device=x.device)
            src = v if v is not None else to_pm(x, buf.dtype)
            buf[:, :C].copy_(src)
            v = buf[:, :C]
        out = torch.empty((N, C, Ho, Wo), dtype=torch.float32, device=x.device)
        ops.upsample_fwd(v, N, C, Hi, Wi, out, Ho, Wo, NCHW)
        ctx.x, ctx.dims, ctx.gdtype = x, (N, C, Hi, Wi, Ho, Wo), v.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, Hi, Wi, Ho, Wo = ctx.dims
        cpad = ops.round_up(C, 32)
        din = torch.empty((N * Hi * Wi, cpad), dtype=ctx.gdtype, device=g.device)
        ops.upsample_bwd(g.float().contiguous(), NCHW, N, C, Ho, Wo, din, cpad, Hi, Wi)
        x = ctx.x
        if pm_view_of(x) is not None and x.dtype == din.dtype:
            return nchw_view_of(din[:, :C], N, Hi, Wi), None
        return ops.pm_to_nchw(din[:, :C], N, Hi, Wi).to(x.dtype), None


def interpolate_bilinear_ac(x, size):
    return UpsampleFunction.apply(x, size)


class MaxPool3x3S2Function(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on a channels_last tensor (the ResNet stem,
    newmodeling/backbone/resnet101_2.py:79,149): one byte of window position per output element instead of a 64-bit
    index, and a gathering backward pass."""

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y, idx = ops.maxpool3x3s2(pm_view_of(x), N, H, W)
        ctx.idx, ctx.dims = idx, (N, C, H, W)
        return nchw_view_of(y, N, (H - 1) // 2 + 1, (W - 1) // 2 + 1)

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.dims
        gv = pm_view_of(g)
        if not ops.maxpool3x3s2_ok(gv):
            gv = pm_view_of(g.contiguous(memory_format=torch.channels_last))
        return nchw_view_of(ops.maxpool3x3s2_bwd(gv, ctx.idx, N, C, H, W), N, H, W)


def maxpool3x3s2(x):
    """The HIP form when x is a channels_last fp32 / bf16 tensor the kernels take, None otherwise."""
    if x.dim() != 4 or not ops.maxpool3x3s2_ok(pm_view_of(x)):
        return None
    return MaxPool3x3S2Function.apply(x)


# ------------------------------------------------------------------- loss ----
def _check_labels(target, K, ignore_index):
    """config.debug_labels: labels that are neither the ignore index nor a class index raise (the reference fails
    inside ATen for them, /root/reference/utils/loss.py:25-30; the kernels treat them as ignored)."""
    if not config.debug_labels() or torch.cuda.is_current_stream_capturing():
        return
    t = target.long()
    bad = int(((t != ignore_index) & ((t < 0) | (t >= K))).sum())
    if bad:
        raise ValueError("cross entropy: %d labels are neither the ignore index %d nor in [0, %d)" % (bad, ignore_index, K))



def _loss_denominator(pack, size_average, batch_average):
    """What the summed loss pack[0] is divided by (utils/loss.py:25-35: CrossEntropyLoss(size_average) then `/ n`):
    the global weight sum pack[1] and / or the global batch pack[2], in as few launches as it takes."""
    if size_average and batch_average:
        return pack[1] * pack[2]
    if size_average:
        return pack[1]
    if batch_average:
        return pack[2]
    return torch.ones((), dtype=torch.float32, device=pack.device)


class CrossEntropyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, target, weight, ignore_index, size_average, batch_average):
        lg = logit if (logit.dtype == torch.float32 and logit.is_contiguous()) else logit.float().contiguous()
        tg = target if (target.dtype == torch.float32 and target.is_contiguous()) else target.float().contiguous()
        n = lg.shape[0]
        _check_labels(tg, lg.shape[1], ignore_index)
        sums, lse = ops.ce_fwd(lg, tg, weight, ignore_index)
        pack = torch.cat([sums, torch.full((1,), float(n), dtype=torch.float32, device=lg.device)])
        parallel.all_reduce_sum_(pack)  # global {sum w*nll, sum w, batch}
        denom = _loss_denominator(pack, size_average, batch_average)
        ctx.saved = (lg, tg, weight, lse, denom, ignore_index, logit.dtype)
        return pack[0] / denom

    @staticmethod
    def backward(ctx, g):
        lg, tg, weight, lse, denom, ignore_index, dt = ctx.saved
        coef = (g.float() / denom).reshape(1).contiguous()
        dl = ops.ce_bwd(lg, tg, weight, lse, ignore_index, coef)
        return (dl if dt == torch.float32 else dl.to(dt)), None, None, None, None, None


def cross_entropy(logit, target, weight=None, ignore_index=255, size_average=True, batch_average=True):
    return CrossEntropyFunction.apply(logit, target, weight, ignore_index, size_average, batch_average)


class UpsampleCrossEntropyFunction(torch.autograd.Function):
    """cross_entropy(interpolate_bilinear_ac(x, target.shape[1:]), target, ...) in two kernels each way, without
    the full-resolution logits (/root/reference/modeling/deeplab.py:252 followed by utils/loss.py:23-35)."""

    @staticmethod
    def forward(ctx, x, target, weight, ignore_index, size_average, batch_average):
        N, K, Hi, Wi = x.shape
        Ho, Wo = int(target.shape[-2]), int(target.shape[-1])
        v = _low_logits_pm(x)
        tg = target if (target.dtype == torch.float32 and target.is_contiguous()) else target.float().contiguous()
        _check_labels(tg, K, ignore_index)
        sums, lse = ops.upsample_ce_fwd(v, N, Hi, Wi, tg, weight, Ho, Wo, ignore_index)
        pack = torch.cat([sums, torch.full((1,), float(N), dtype=torch.float32, device=x.device)])
        parallel.all_reduce_sum_(pack)  # global {sum w*nll, sum w, batch}
        denom = _loss_denominator(pack, size_average, batch_average)
        ctx.saved = (x, v, tg, weight, lse, denom, ignore_index, (N, K, Hi, Wi, Ho, Wo))
        return pack[0] / denom

    @staticmethod
    def backward(ctx, g):
        x, v, tg, weight, lse, denom, ignore_index, (N, K, Hi, Wi, Ho, Wo) = ctx.saved
        coef = (g.float() / denom).reshape(1).contiguous()
        dlow = ops.upsample_ce_bwd(v, N, Hi, Wi, tg, weight, lse, Ho, Wo, ignore_index, coef, ops.round_up(K, 32))
        if pm_view_of(x) is not None and x.dtype == torch.float32:
            return nchw_view_of(dlow[:, :K], N, Hi, Wi), None, None, None, None, None
        return ops.pm_to_nchw(dlow[:, :K], N, Hi, Wi).to(x.dtype), None, None, None, None, None


def _low_logits_pm(x):
    """float32 pixel-major view [N*Hi*Wi, K] of low-resolution logits the fused kernels can read (copy if needed)."""
    N, K, Hi, Wi = x.shape
    v = pm_view_of(x)
    if v is None or v.dtype != torch.float32 or v.stride(0) % 4 != 0 or v.stride(0) < ops.round_up(K, 4) \
            or v.data_ptr() % 16 != 0:
        buf = torch.zeros((N * Hi * Wi, ops.round_up(K, 4)), dtype=torch.float32, device=x.device)
        buf[:, :K].copy_(v if v is not None else to_pm(x, torch.float32))
        v = buf[:, :K]
    return v


def upsample_argmax(x, size):
    """np.argmax(interpolate_bilinear_ac(x, size), axis=1) on the device without the full-size logits."""
    N, K, Hi, Wi = x.shape
    return ops.upsample_argmax(_low_logits_pm(x), N, Hi, Wi, int(size[0]), int(size[1]))


def upsample_cross_entropy(x, target, weight=None, ignore_index=255, size_average=True, batch_average=True):
    """Loss of the bilinearly (align_corners) upsampled logits x against a target of the full size."""
    return UpsampleCrossEntropyFunction.apply(x, target, weight, ignore_index, size_average, batch_average)


# ----------------------------------------------------------------- SyncBN ----
def _pm_aligned8(v):
    """Whether a pixel-major view allows the kernels' 8-channel vector access."""
    return v.shape[1] % 8 == 0 and v.stride(0) % 8 == 0 and v.data_ptr() % 16 == 0


class SyncBNFunction(torch.autograd.Function):
    """Stand-alone batch norm on NCHW (or channels_last) tensors: the feeder's layers.  Optionally with the ReLU
    that follows it, and with the residual that a bottleneck block adds in between, in the same passes."""

    @staticmethod
    def forward(ctx, x, gamma, beta, mod, relu=False, residual=None, grad_slot=None):
        """grad_slot (GradSlot or None): where the residual's gradient goes instead of back through autograd."""
        ctx.grad_slot = grad_slot
        spec = _BNSpec(mod)
        layout, xv, shape = NCHW, x, x.shape
        v = pm_view_of(x) if x.dim() == 4 else None
        if v is not None and not x.is_contiguous() and v.shape[1] % 4 == 0:
            layout, xv = PM, v
        elif not x.is_contiguous():
            xv = x.contiguous()
        if xv.dtype not in (torch.float32, torch.bfloat16):
            xv = xv.float()
        rv = None
        if residual is not None:
            rv = pm_view_of(residual) if layout == PM and residual.shape == x.shape else None
            if rv is None or rv.dtype != x.dtype or xv.dtype != x.dtype:
                raise ValueError("the fused residual form needs channels_last input and residual of one dtype")
        if spec.training:
            partials, cnt = ops.bn_stat_partials(xv, layout)
            coef = _train_coeffs(partials, cnt, spec, gamma, beta)
        else:
            coef = ops.bn_coeffs_eval(spec.running_mean, spec.running_var, gamma, beta, spec.eps)
        bits = None
        if layout == PM:
            y = torch.empty_like(x, memory_format=torch.channels_last)
            yv = pm_view_of(y)
            if rv is not None and relu:
                # with a residual the backward cannot recompute the ReLU pattern from x; it gets one bit per element
                # (1/16 of re-reading the bf16 output) when the vector kernels apply, else the output itself
                bits = ops.relu_bits_like(yv) if (_pm_aligned8(xv) and _pm_aligned8(rv)) else None
            ops.bn_apply(xv, yv, coef, PM, relu, res=rv, relu_bits=bits)
        else:
            y = torch.empty_like(xv)
            ops.bn_apply(xv, y, coef, NCHW, relu)
        if CAPTURE is not None and relu:
            CAPTURE.append((y, None))   # already NCHW-shaped
        # without a residual the ReLU pattern is recomputed from x*scale + shift; with one it is read from y
        act = (bits if bits is not None else pm_view_of(y)) if (rv is not None and relu) else None
        ctx.saved = (xv, coef, layout, spec.training, spec.sync, x.dtype, shape, relu, act, rv is not None)
        ctx.affine = gamma is not None
        return y if y.dtype == x.dtype else y.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        xv, coef, layout, training, sync, in_dtype, shape, relu, act, has_res = ctx.saved
        C = shape[1]
        dres = None
        if layout == PM:
            gv = pm_view_of(g)
            if gv is None or gv.dtype != xv.dtype:
                gc = g.to(xv.dtype).contiguous(memory_format=torch.channels_last)
                gv = pm_view_of(gc)
            if act is not None and act.dtype == torch.uint8:
                gv = _vector_aligned(gv)   # the bit pattern is read by the 8-channel vector kernels only
            dx = torch.empty(shape, dtype=xv.dtype, device=g.device, memory_format=torch.channels_last)
            dxv = pm_view_of(dx)
            # the consumer of the residual's gradient can mask the incoming gradient with the stored ReLU bits itself
            lazy = (has_res and ctx.grad_slot is not None and ctx.needs_input_grad[5] and act is not None
                    and act.dtype == torch.uint8 and relu)
            if has_res and not lazy:
                dres = torch.empty(shape, dtype=xv.dtype, device=g.device, memory_format=torch.channels_last)
        else:
            lazy = False
            gv = g.to(xv.dtype).contiguous()
            dx = torch.empty_like(xv)
            dxv = dx
        partials, cnt = ops.bn_bwd_stat_partials(gv, xv, coef, layout, relu, act=act)
        local, dgb = ops.bn_reduce(partials, cnt, want_f32=True)
        dbeta, dgamma = dgb[0], dgb[1]
        gs = _global_sums(local, sync and training)
        ops.bn_bwd_apply(gv, xv, dxv, coef, gs, layout, relu, eval_mode=not training, act=act,
                         dres=pm_view_of(dres) if dres is not None else None)
        if dx.dtype != in_dtype:
            dx = dx.to(in_dtype)
        if not ctx.affine:
            dgamma = dbeta = None
        if lazy:
            ctx.grad_slot.put(gv, bits=act)
        elif ctx.grad_slot is not None and dres is not None and ctx.needs_input_grad[5]:
            ctx.grad_slot.put(dres)
            dres = None
        return dx, dgamma, dbeta, None, None, dres, None


def _feeder_compute_dtype(x):
    """Operand dtype of a feeder unit: bf16 for bf16 activations and under bf16 autocast, else fp32 (an fp16
    autocast region is not this path's: its fp32 result is what autocast's own cast policy then handles)."""
    if x.dtype == torch.bfloat16:
        return torch.bfloat16
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    return torch.float32


def _vector_aligned(v):
    """The pixel-major view itself when it allows 8-channel vector access, else a contiguous copy (which does,
    for channel counts that are multiples of 8)."""
    return v if _pm_aligned8(v) else v.contiguous()


class GradSlot:
    """Hand-over of one gradient between the two backward functions of a bottleneck block whose input x feeds both
    conv1 and the shortcut: the function that produces the shortcut's gradient (it runs first in backward) leaves it
    here instead of returning it to autograd, and conv1's function adds it inside its data-gradient convolution, so
    x receives ONE gradient and autograd's separate addition of two 36 - 136 MB tensors per block disappears."""
    __slots__ = ("grad", "bits", "taken")

    def __init__(self):
        self.grad = self.bits = None
        self.taken = False

    def put(self, g, bits=None):
        """g: the gradient (an NCHW-shaped channels_last tensor or its pixel-major view).  bits (uint8 [M, C/8], the
        ReLU pattern of the block's last unit): the gradient is g where the bit is set and zero elsewhere - the
        consumer applies the pattern, so the producer need not write a masked copy of g."""
        self.grad, self.bits = g, bits

    def take(self):
        """-> (gradient or None, bit pattern or None)"""
        g, b = self.grad, self.bits
        self.grad = self.bits = None
        self.taken = True
        return g, b


class GradToSlotFunction(torch.autograd.Function):
    """Identity in forward.  In backward the gradient goes into a GradSlot instead of back to autograd: for a block
    whose input x feeds conv1's unit and a projection shortcut (newmodeling/backbone/resnet101_2.py:46-47,
    `residual = self.downsample(x)`), the shortcut's backward runs first (its nodes were recorded after conv1's and
    conv2's), and conv1's data-gradient convolution adds what it finds in the slot - x receives one gradient instead
    of two that autograd adds in a pass of its own.  Should the consumer already have run (another execution order, a
    second backward pass over a kept graph), the gradient is returned to autograd as usual."""

    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.slot.taken:
            return g, None
        ctx.slot.put(g)
        return None, None


def grad_to_slot(x, slot):
    return GradToSlotFunction.apply(x, slot)


class StatLink:
    """Hand-over between a unit (convolution -> batch norm -> ReLU) and the ONE unit that consumes its output.  In
    forward the producer offers what its batch norm's backward statistics need (its pre-normalisation output, the
    coefficient rows, the ReLU pattern if stored).  In backward the consumer runs first: its data-gradient convolution
    produces the producer's output gradient, and its epilogue emits the partial sums of g and g * xhat from the tile in
    registers (ops.conv_fwd(bwd_stats=...)); the producer then skips the pass that would read the gradient and its
    own pre-normalisation output again.  The rows are only used when the gradient the producer receives IS the tensor
    that launch wrote, unchanged (same memory, same version counter): any other route - a second consumer whose
    gradient autograd adds, a hook - falls back to the stand-alone pass."""
    __slots__ = ("raw", "coef", "relu", "bits", "rows", "out", "version")

    def __init__(self):
        self.raw = self.coef = self.bits = self.rows = self.out = None
        self.relu, self.version = True, -1

    def offer(self, raw, coef, relu=True, bits=None):
        self.raw, self.coef, self.relu, self.bits = raw, coef, relu, bits
        self.rows = self.out = None

    def offered(self):
        return self.raw is not None and self.coef is not None

    def request(self):
        return ops.BwdStatRequest(self.raw, self.coef, self.relu, self.bits)

    def fulfil(self, rows, out):
        self.rows, self.out, self.version = rows, out, out._version

    def rows_for(self, gv):
        """The emitted rows when gv (the pixel-major gradient the producer received) is the tensor they were computed
        from, else None; the link is spent either way."""
        rows, out, version = self.rows, self.out, self.version
        self.raw = self.coef = self.bits = self.rows = self.out = None
        if rows is None or out is None or gv is None:
            return None
        if (gv.data_ptr() != out.data_ptr() or gv.shape != out.shape or gv.stride() != out.stride()
                or gv.dtype != out.dtype or out._version != version):
            return None
        return rows


# ------------------------------------------------- feeder conv + BN + ReLU ----
class ConvBNReLUFunction(torch.autograd.Function):
    """relu(bn(conv(x))) of a feeder bottleneck block (newmodeling/backbone/resnet101_2.py:19-21 and
    Bottleneck.forward, conv2 -> bn2 -> relu: 3x3, stride 1, padding = dilation; conv1 -> bn1 -> relu: 1x1) as one
    unit on the implicit-GEMM
    kernels: the convolution's epilogue emits the batch-norm statistics, one pass applies BN + ReLU; backward is the
    BN reduce / apply pair, the weight-gradient GEMM and the data-gradient convolution.  Operands take the
    activation's dtype (bf16 under autocast); the convolution output is kept in that dtype as well, like the
    PyTorch path it replaces.  The weight may be fp32 or bf16, contiguous or channels_last; its gradient comes back
    in the weight's own dtype and layout."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, conv, bn, grad_slot=None, stats_for=None, stats_to=None):
        """grad_slot (GradSlot or None): a second gradient of x that another function of the block leaves there.
        stats_for (StatLink or None): the link the unit that produced x offered - x reaches this unit only (and the
        shortcut whose gradient arrives through grad_slot); stats_to (StatLink or None): offered to this unit's only
        consumer."""
        N, Cin, H, W = x.shape
        R, dil = conv.kernel_size[0], conv.dilation[0]
        cd = _feeder_compute_dtype(x)
        xv = _padded_input(to_pm(x, cd))
        Cout = weight.shape[0]
        M = N * H * W
        spec = _BNSpec(bn)
        raw = torch.empty((M, Cout), dtype=cd, device=x.device)
        w_op = weight.detach()
        if w_op.dtype != cd or not ops.weight_usable_in_place(w_op, R, Cout):   # a channels_last bf16 weight is read as it is
            w_op = ops.pack_weight(w_op, cd)
        with ops.scoped("@feeder"):
            stats = ops.conv_fwd([(xv, w_op, R, dil)], N, H, W, Cout, raw, want_stats=spec.training, true_ck=[Cin])
        y = torch.empty((N, Cout, H, W), dtype=cd, device=x.device, memory_format=torch.channels_last)
        rec = unit_finish(xv, raw, stats, weight, R, dil, (N, H, W), spec, gamma, beta, pm_view_of(y), relu=True)
        ctx.rec, ctx.x, ctx.grad_slot = rec, x, grad_slot
        ctx.stats_for, ctx.stats_to = stats_for, None
        nchw_caller = pm_view_of(x) is None or x.is_contiguous()
        if stats_to is not None and not nchw_caller and (y.dtype == x.dtype or torch.is_autocast_enabled()):
            stats_to.offer(raw, rec.coef, True)
            ctx.stats_to = stats_to
        if nchw_caller:   # hand an NCHW caller an NCHW tensor back
            y = y.contiguous()
        return y if y.dtype == x.dtype or torch.is_autocast_enabled() else y.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        rec, x = ctx.rec, ctx.x
        N, H, W = rec.geom
        with ops.scoped("@feeder"):
            gv = to_pm(g, rec.x.dtype)
            rows = ctx.stats_to.rows_for(gv) if ctx.stats_to is not None else None
            dy, dw, dgamma, dbeta = unit_backward(rec, gv, need_dw=ctx.needs_input_grad[1], dw_like=rec.weight,
                                                  partials=rows)
            dx = None
            other, other_bits = ctx.grad_slot.take() if ctx.grad_slot is not None else (None, None)
            if ctx.needs_input_grad[0]:
                gd = grad_dtype_for(x)
                ov = None
                if other is not None:
                    ov = other if other.dim() == 2 else pm_view_of(other)
                if other is not None and (ov is None or ov.dtype != gd):
                    # not a pixel-major tensor of the gradient's dtype: add afterwards
                    late = other if other.dim() == 4 else grad_like(x, other, N, H, W)
                    if other_bits is not None:
                        keep = ops.unpack_bits(other_bits, rec.Cin).reshape(N, H, W, rec.Cin).permute(0, 3, 1, 2)
                        late = late * keep.to(late.dtype)
                    ov = other_bits = None
                else:
                    late = None
                dxp = dgrad([(dy, rec.weight, rec.R, rec.dil)], rec.geom, rec.Cin, gd, addend=ov, addend_bits=other_bits,
                            stat_link=ctx.stats_for if late is None else None)
                dx = grad_like(x, dxp, N, H, W)
                if late is not None:
                    dx = dx + late.to(dx.dtype)
            elif other is not None:
                raise RuntimeError("a gradient was left for an input that needs none")
        return dx, dw, dgamma, dbeta, None, None, None, None, None


class ConvBNAddReLUFunction(torch.autograd.Function):
    """relu(bn(conv1x1(x)) + residual): the tail of a feeder bottleneck block (newmodeling/backbone/resnet101_2.py
    Bottleneck.forward, conv3 -> bn3 -> += residual -> relu) as one unit: statistics from the convolution's epilogue,
    one pass applies BN, adds the residual, applies the ReLU and stores its pattern as one bit per element; backward
    is the BN reduce / apply pair (which also emits the residual's gradient), the weight-gradient GEMM and the
    data-gradient convolution.  Channels_last operands of one dtype."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv, bn, grad_slot=None, stats_for=None, stats_to=None):
        """grad_slot (GradSlot or None): where the residual's gradient goes instead of back through autograd.
        stats_for / stats_to (StatLink or None): as in ConvBNReLUFunction.forward."""
        ctx.grad_slot = grad_slot
        ctx.stats_for, ctx.stats_to = stats_for, None
        N, Cin, H, W = x.shape
        R, dil = conv.kernel_size[0], conv.dilation[0]
        cd = _feeder_compute_dtype(x)
        xv = _padded_input(to_pm(x, cd))
        rv = pm_view_of(residual)
        Cout = weight.shape[0]
        M = N * H * W
        if rv is None or rv.dtype != cd or tuple(residual.shape) != (N, Cout, H, W):
            raise ValueError("the fused residual form needs a channels_last residual of the compute dtype")
        spec = _BNSpec(bn)
        raw = torch.empty((M, Cout), dtype=cd, device=x.device)
        w_op = weight.detach()
        if w_op.dtype != cd or not ops.weight_usable_in_place(w_op, R, Cout):
            w_op = ops.pack_weight(w_op, cd)
        with ops.scoped("@feeder"):
            stats = ops.conv_fwd([(xv, w_op, R, dil)], N, H, W, Cout, raw, want_stats=spec.training, true_ck=[Cin])
        if spec.training:
            coef = _train_coeffs(stats, float(M), spec, gamma, beta)
        else:
            coef = ops.bn_coeffs_eval(spec.running_mean, spec.running_var, gamma, beta, spec.eps)
        y = torch.empty((N, Cout, H, W), dtype=cd, device=x.device, memory_format=torch.channels_last)
        yv = pm_view_of(y)
        bits = ops.relu_bits_like(yv) if (_pm_aligned8(raw) and _pm_aligned8(rv)) else None
        ops.bn_apply(raw, yv, coef, PM, True, res=rv, relu_bits=bits)
        if CAPTURE is not None:
            CAPTURE.append((y, None))
        ctx.saved = (x, xv, raw, coef, bits if bits is not None else yv, weight, (N, H, W), R, dil, spec.training,
                     spec.sync)
        if stats_to is not None and bits is not None:
            stats_to.offer(raw, coef, True, bits)
            ctx.stats_to = stats_to
        return y

    @staticmethod
    def backward(ctx, g):
        x, xv, raw, coef, act, weight, geom, R, dil, training, sync = ctx.saved
        N, H, W = geom
        Cout, Cin = weight.shape[0], weight.shape[1]
        gv = to_pm(g, raw.dtype)
        if act.dtype == torch.uint8:
            gv = _vector_aligned(gv)   # the bit pattern is read by the 8-channel vector kernels only
        partials = ctx.stats_to.rows_for(gv) if ctx.stats_to is not None else None
        cnt = float(gv.shape[0])
        if partials is None:
            partials, cnt = ops.bn_bwd_stat_partials(gv, raw, coef, PM, True, act=act)
        local, dgb = ops.bn_reduce(partials, cnt, want_f32=True)
        dbeta, dgamma = dgb[0], dgb[1]
        gs = _global_sums(local, sync and training)
        dy = torch.empty_like(raw)
        # the consumer of the residual's gradient can mask the incoming gradient with the stored ReLU bits itself
        lazy = ctx.grad_slot is not None and ctx.needs_input_grad[4] and act.dtype == torch.uint8
        dres = None if lazy else torch.empty((N, Cout, H, W), dtype=raw.dtype, device=g.device,
                                             memory_format=torch.channels_last)
        ops.bn_bwd_apply(gv, raw, dy, coef, gs, PM, True, eval_mode=not training, act=act,
                         dres=pm_view_of(dres) if dres is not None else None)
        with ops.scoped("@feeder"):
            dw = None
            if ctx.needs_input_grad[1]:
                dw = weight_grad(weight, xv, dy, N, H, W, R, dil, Cout, Cin, like=weight)
            dx = None
            if ctx.needs_input_grad[0]:
                dxp = dgrad([(dy, weight, R, dil)], geom, Cin, grad_dtype_for(x), stat_link=ctx.stats_for)
                dx = grad_like(x, dxp, N, H, W)
        if lazy:
            ctx.grad_slot.put(gv, bits=act)
        elif ctx.grad_slot is not None and ctx.needs_input_grad[4]:
            ctx.grad_slot.put(dres)
            dres = None
        return dx, dw, dgamma, dbeta, (dres if ctx.needs_input_grad[4] else None), None, None, None, None, None


# ------------------------------------------------------------ plain conv ----
class ConvFunction(torch.autograd.Function):
    """nn.Conv2d with stride 1, a square kernel and padding = dilation*(k-1)/2 (output size = input size),
    optional bias: the 1x1 projections of modeling/damm_cam_pam.py:14-16 and the size-preserving part of
    modeling/deeplab_damm.py:31-32."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil):
        cd = config.compute_dtype()
        N, Cin, H, W = x.shape
        Cout, R = weight.shape[0], weight.shape[2]
        xp = _padded_input(to_pm(x, cd))
        out = torch.empty((N * H * W, ops.round_up(Cout, 4)), dtype=cd, device=x.device)
        ops.conv_fwd([(xp, ops.pack_weight(weight.detach(), cd), R, dil)], N, H, W, Cout, out,
                     bias=None if bias is None else bias.detach().float().contiguous(), true_ck=[Cin])
        ctx.saved = (x, xp, weight, R, dil, (N, H, W), bias is not None)
        return nchw_view_of(out[:, :Cout] if out.shape[1] != Cout else out, N, H, W)

    @staticmethod
    def backward(ctx, g):
        x, xp, weight, R, dil, geom, has_bias = ctx.saved
        N, H, W = geom
        Cout, Cin = weight.shape[0], weight.shape[1]
        gp = _padded_input(to_pm(g, xp.dtype))
        dw = ops.conv_wgrad(xp, gp, N, H, W, R, dil, Cout, Cin) if ctx.needs_input_grad[1] else None
        db = ops.pool_sum(gp, 1, N * H * W, 1.0).reshape(-1)[:Cout].contiguous() if has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            dxp = dgrad([(gp, weight, R, dil)], geom, Cin, grad_dtype_for(x))
            dx = grad_like(x, dxp, N, H, W)
        return dx, dw, db, None


def conv2d_same(x, weight, bias=None, dilation=1):
    return ConvFunction.apply(x, weight, bias, dilation)


# ------------------------------------------------------- position attention ----
def _b3(v, N, P):
    """[N*P, C] pixel-major view -> [N, P, C] view (no copy)."""
    return v.unflatten(0, (N, P))


class PAMFunction(torch.autograd.Function):
    """_PAM_Module.forward (modeling/damm_cam_pam.py:22-48): q/k/v 1x1 projections, energy = q^T k per image,
    row softmax, out = v att^T, gamma*out + x.  Query/key projections and energies stay fp32 whatever the compute
    dtype (the softmax amplifies operand rounding); the value path runs in the compute dtype."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, gamma):
        cd = config.compute_dtype()
        N, C, H, W = x.shape
        P, M = H * W, N * H * W
        if C % 32 != 0:
            raise ValueError("attention modules need a channel count that is a multiple of 32, got %d" % C)
        Pp = ops.round_up(P, 32)
        Cq = wq.shape[0]
        Cqp = ops.round_up(Cq, 32)
        xp = to_pm(x, cd)
        f32 = torch.float32
        alloc = torch.zeros if Cqp != ops.round_up(Cq, 4) else torch.empty
        qk = alloc((M, 2 * Cqp), dtype=f32, device=x.device)
        for i, (w, b) in enumerate(((wq, bq), (wk, bk))):
            ops.conv_fwd([(xp, ops.pack_weight(w.detach(), cd), 1, 1)], N, H, W, Cq, qk[:, i * Cqp:(i + 1) * Cqp],
                         bias=b.detach().float().contiguous(), true_ck=[C])
        v = torch.empty((M, C), dtype=cd, device=x.device)
        ops.conv_fwd([(xp, ops.pack_weight(wv.detach(), cd), 1, 1)], N, H, W, C, v,
                     bias=bv.detach().float().contiguous(), true_ck=[C])
        e = torch.empty((N, P, Pp), dtype=f32, device=x.device)
        ops.gemm_nt([(_b3(qk[:, :Cqp], N, P), _b3(qk[:, Cqp:], N, P))], e, true_k=[Cq])
        att = torch.empty((N, P, Pp), dtype=cd, device=x.device)
        ops.softmax_rows(e, P, att)
        del e
        vt = torch.empty((N, C, Pp), dtype=cd, device=x.device)
        ops.transpose(_b3(v, N, P), vt)
        o = torch.empty((M, C), dtype=cd, device=x.device)
        ops.gemm_nt([(att, vt)], _b3(o, N, P), true_k=[P])
        del vt
        out = torch.empty((M, C), dtype=cd, device=x.device)
        gdev = gamma.detach().float().reshape(1)
        ops.lincomb(out, [(o, 1.0, gdev), (xp, 1.0, None)])
        ctx.saved = (x, xp, qk, v, att, o, wq, wk, wv, gdev, (N, H, W), Cq, Cqp, Pp)
        return nchw_view_of(out, N, H, W)

    @staticmethod
    def backward(ctx, g):
        x, xp, qk, v, att, o, wq, wk, wv, gdev, geom, Cq, Cqp, Pp = ctx.saved
        N, H, W = geom
        P, M, C = H * W, N * H * W, xp.shape[1]
        cd, f32, dev = xp.dtype, torch.float32, xp.device
        gp = to_pm(g, cd)
        dgamma = ops.dot(gp, o)
        do = torch.empty((M, C), dtype=cd, device=dev)
        ops.lincomb(do, [(gp, 1.0, gdev)])
        do3 = _b3(do, N, P)
        datt = torch.empty((N, P, Pp), dtype=f32, device=dev)
        ops.gemm_nt([(do3, _b3(v, N, P))], datt, true_k=[C])
        # gradient buffer of the three projections: [dq | dk | dv], padded slices zero
        dqkv = torch.zeros((M, 2 * Cqp + C), dtype=cd, device=dev)
        ops.gemm_tn(att, do3, _b3(dqkv[:, 2 * Cqp:], N, P), rows=P, cols=C, true_k=P)
        de = torch.empty((N, P, Pp), dtype=f32, device=dev)
        ops.softmax_rows_bwd(att, datt, P, de, 1.0)
        del datt
        q3, k3 = _b3(qk[:, :Cqp], N, P), _b3(qk[:, Cqp:], N, P)
        kt = torch.empty((N, Cqp, Pp), dtype=f32, device=dev)
        ops.transpose(k3, kt)
        ops.gemm_nt([(de, kt)], _b3(dqkv[:, :Cqp], N, P), true_k=[P])
        ops.gemm_tn(de, q3, _b3(dqkv[:, Cqp:2 * Cqp], N, P), rows=P, cols=Cqp, true_k=P)
        del de, kt
        sl = (dqkv[:, :Cqp], dqkv[:, Cqp:2 * Cqp], dqkv[:, 2 * Cqp:])
        grads = []
        for s, w in zip(sl, (wq, wk, wv)):
            co = w.shape[0]
            grads.append(ops.conv_wgrad(xp, s, N, H, W, 1, 1, co, C))
            grads.append(ops.pool_sum(s, 1, M, 1.0).reshape(-1)[:co].contiguous())
        dx = None
        if ctx.needs_input_grad[0]:
            dxa = dgrad([(sl[0], wq, 1, 1), (sl[1], wk, 1, 1), (sl[2], wv, 1, 1)], geom, C, cd)
            dxp = torch.empty((M, C), dtype=grad_dtype_for(x), device=dev)
            ops.lincomb(dxp, [(dxa, 1.0, None), (gp, 1.0, None)])
            dx = grad_like(x, dxp, N, H, W)
        return (dx,) + tuple(grads) + (dgamma,)


# -------------------------------------------------------- channel attention ----
class CAMFunction(torch.autograd.Function):
    """_CAM_Module.forward (modeling/damm_cam_pam.py:72-94): energy = x x^T over pixels per image ([C,C]),
    softmax of (row max - energy), out = att x, gamma*out + x.  Energies are accumulated and normalised in fp32."""

    @staticmethod
    def forward(ctx, x, gamma):
        cd = config.compute_dtype()
        N, C, H, W = x.shape
        P, M = H * W, N * H * W
        if C % 32 != 0:
            raise ValueError("attention modules need a channel count that is a multiple of 32, got %d" % C)
        Pp = ops.round_up(P, 32)
        xp = to_pm(x, cd)
        x3 = _b3(xp, N, P)
        xt = torch.empty((N, C, Pp), dtype=cd, device=x.device)
        ops.transpose(x3, xt)
        e = torch.empty((N, C, C), dtype=torch.float32, device=x.device)
        ops.gemm_nt([(xt, xt)], e, true_k=[P])
        del xt
        att = torch.empty((N, C, C), dtype=cd, device=x.device)
        ops.softmax_rows(e, C, att, reversed_=True)
        del e
        o = torch.empty((M, C), dtype=cd, device=x.device)
        ops.gemm_nt([(x3, att)], _b3(o, N, P), true_k=[C])
        out = torch.empty((M, C), dtype=cd, device=x.device)
        gdev = gamma.detach().float().reshape(1)
        ops.lincomb(out, [(o, 1.0, gdev), (xp, 1.0, None)])
        ctx.saved = (x, xp, att, o, gdev, (N, H, W))
        return nchw_view_of(out, N, H, W)

    @staticmethod
    def backward(ctx, g):
        x, xp, att, o, gdev, geom = ctx.saved
        N, H, W = geom
        P, M, C = H * W, N * H * W, xp.shape[1]
        cd, dev = xp.dtype, xp.device
        gp = to_pm(g, cd)
        dgamma = ops.dot(gp, o)
        dx = None
        if ctx.needs_input_grad[0]:
            do = torch.empty((M, C), dtype=cd, device=dev)
            ops.lincomb(do, [(gp, 1.0, gdev)])
            do3, x3 = _b3(do, N, P), _b3(xp, N, P)
            datt = torch.empty((N, C, C), dtype=torch.float32, device=dev)
            ops.gemm_tn(do3, x3, datt, true_k=P)
            de = torch.empty((N, C, C), dtype=cd, device=dev)
            ops.softmax_rows_bwd(att, datt, C, de, -1.0)
            del datt
            att_t = torch.empty((N, C, C), dtype=cd, device=dev)
            ops.transpose(att, att_t)
            de_t = torch.empty((N, C, C), dtype=cd, device=dev)
            ops.transpose(de, de_t)
            dxa = torch.empty((M, C), dtype=cd, device=dev)
            ops.gemm_nt([(do3, att_t), (x3, de), (x3, de_t)], _b3(dxa, N, P), true_k=[C, C, C])
            del att_t, de_t, de
            dxp = torch.empty((M, C), dtype=grad_dtype_for(x), device=dev)
            ops.lincomb(dxp, [(dxa, 1.0, None), (gp, 1.0, None)])
            dx = grad_like(x, dxp, N, H, W)
        return dx, dgamma
