"""Kernel-level parity on a real MI355X: each C-ABI entry point against the same op in PyTorch
(fp32 reference on the GPU, or numpy for the integer work)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import gen, ops_ref  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from segk import ops as o
    return o


DEV = "cuda"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def to_pm(x, dtype, ld=None):
    """NCHW torch tensor -> pixel-major view [M, C] inside a zero buffer with leading dimension ld."""
    N, C, H, W = x.shape
    ld = ld or C
    buf = torch.zeros((N * H * W, ld), dtype=dtype, device=x.device)
    buf[:, :C] = x.permute(0, 2, 3, 1).reshape(-1, C).to(dtype)
    return buf[:, :C] if ld != C else buf, buf


def from_pm(v, N, H, W):
    return v.float().reshape(N, H, W, -1).permute(0, 3, 1, 2).contiguous()


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------ layout ----
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layout_roundtrip(ops, dtype):
    x = T(gen.normal(1, "lay", (3, 37, 5, 13)))
    buf = torch.zeros((3 * 5 * 13, 64), dtype=dtype, device=DEV)
    view = buf[:, 8:45]
    ops.nchw_to_pm(x, view)
    want = x.permute(0, 2, 3, 1).reshape(-1, 37).to(dtype)
    assert torch.equal(view, want)
    assert float(buf[:, :8].abs().sum()) == 0 and float(buf[:, 45:].abs().sum()) == 0
    back = ops.pm_to_nchw(view, 3, 5, 13)
    assert torch.equal(back, want.float().reshape(3, 5, 13, 37).permute(0, 3, 1, 2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pack_weight(ops, dtype):
    w = T(gen.normal(1, "pw", (21, 40, 3, 3)))
    p = ops.pack_weight(w, dtype, transpose=False)
    assert tuple(p.shape) == (9, 21, 64)
    want = torch.zeros((9, 21, 64), dtype=dtype, device=DEV)
    want[:, :, :40] = w.permute(2, 3, 0, 1).reshape(9, 21, 40).to(dtype)
    assert torch.equal(p, want)
    pt = ops.pack_weight(w, dtype, transpose=True)
    assert tuple(pt.shape) == (9, 40, 32)
    want = torch.zeros((9, 40, 32), dtype=dtype, device=DEV)
    want[:, :, :21] = w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, 40, 21).to(dtype)
    assert torch.equal(pt, want)


# -------------------------------------------------------------------- conv ----
CONV_CASES = [
    # N, H, W, Cin, Cout, R, dil
    (2, 9, 7, 64, 48, 1, 1),
    (1, 17, 23, 96, 256, 3, 1),
    (2, 17, 23, 128, 130, 3, 2),
    (2, 33, 33, 64, 64, 3, 6),
    (1, 33, 33, 64, 32, 3, 12),
    (1, 20, 33, 32, 21, 3, 18),
    (2, 13, 11, 64, 16, 3, 36),
]


# shapes that select the 256x256-tile kernel (>= 512 tiles, Cout a multiple of 256, 128-byte K rows): the first two
# leave 2 tiles beyond whole rounds of 256 workgroups (main launch + small-tile tail launch), the third 195 (one launch)
BIG_TILE_CASES = [(1, 256, 257, 64, 512, 3, 2), (2, 181, 182, 128, 256, 1, 1), (1, 340, 339, 64, 256, 3, 1)]


def _conv_inputs(case, integer=False):
    N, H, W, Cin, Cout, R, dil = case
    tag = "conv%s" % (case,)
    if integer:
        x = gen.integers(3, tag + "x", (N, Cin, H, W), -3, 4).astype(np.float32)
        w = gen.integers(3, tag + "w", (Cout, Cin, R, R), -2, 3).astype(np.float32)
    else:
        x = gen.normal(3, tag + "x", (N, Cin, H, W))
        w = gen.normal(3, tag + "w", (Cout, Cin, R, R), std=(2.0 / (Cin * R * R)) ** 0.5)
    return T(x), T(w)


@pytest.mark.parametrize("case", CONV_CASES + BIG_TILE_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_fwd_exact_integers(ops, case, dtype):
    """Small-integer operands make every product and sum exact in bf16 and fp32: any fragment-layout
    or tap-indexing mistake shows up as a hard mismatch."""
    N, H, W, Cin, Cout, R, dil = case
    x, w = _conv_inputs(case, integer=True)
    xv, _ = to_pm(x, dtype, ld=Cin + 32)
    wp = ops.pack_weight(w, dtype)
    out = torch.zeros((N * H * W, ops.round_up(Cout, 4) + 4), dtype=torch.float32, device=DEV)
    ov = out[:, :ops.round_up(Cout, 4)]
    stats = ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, ov, want_stats=True)
    ref = F.conv2d(x, w, padding=dil * (R // 2), dilation=dil)
    got = from_pm(ov[:, :Cout], N, H, W)
    assert torch.equal(got, ref)
    assert float(out[:, ops.round_up(Cout, 4):].abs().sum()) == 0
    s = stats.sum(0)
    assert torch.equal(s[0], ref.double().sum((0, 2, 3))) and torch.equal(s[1], (ref.double() ** 2).sum((0, 2, 3)))


@pytest.mark.parametrize("case", CONV_CASES + BIG_TILE_CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_conv_fwd_random(ops, case, dtype, tol):
    N, H, W, Cin, Cout, R, dil = case
    x, w = _conv_inputs(case)
    bias = T(gen.normal(3, "convb%s" % (case,), (Cout,)))
    img_bias = T(gen.normal(3, "convib%s" % (case,), (N, Cout)))
    xv, _ = to_pm(x, dtype)
    wp = ops.pack_weight(w, dtype)
    out = torch.zeros((N * H * W, ops.round_up(Cout, 4)), dtype=dtype, device=DEV)
    ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, out, bias=bias, img_bias=img_bias)
    ref = F.conv2d(x, w, bias, padding=dil * (R // 2), dilation=dil) + img_bias[:, :, None, None]
    assert relerr(from_pm(out[:, :Cout], N, H, W), ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_conv_multi_segment_is_sum(ops, dtype, tol):
    """The ASPP data gradient: four branches (1x1, 3x3 at three rates) summed into one output."""
    N, H, W, C, Cout = 2, 21, 19, 64, 96
    segs, ref = [], 0
    for i, (R, dil) in enumerate([(1, 1), (3, 6), (3, 12), (3, 18)]):
        x = T(gen.normal(4, "ms_x%d" % i, (N, C, H, W)))
        w = T(gen.normal(4, "ms_w%d" % i, (Cout, C, R, R), std=0.05))
        segs.append((to_pm(x, dtype)[0], ops.pack_weight(w, dtype), R, dil))
        ref = ref + F.conv2d(x, w, padding=dil * (R // 2), dilation=dil)
    out = torch.empty((N * H * W, Cout), dtype=torch.float32, device=DEV)
    ops.conv_fwd(segs, N, H, W, Cout, out)
    assert relerr(from_pm(out, N, H, W), ref) < tol


# N, H, W, Cin, Cout, R, dil: 128 output channels = one narrow ring tile, 256 = the wide one, 512 = two of them
ADDEND_CASES = [(2, 17, 15, 128, 256, 3, 2), (1, 33, 33, 256, 128, 1, 1), (2, 9, 11, 64, 512, 1, 1)]


@pytest.mark.parametrize("case", ADDEND_CASES)
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shift", [0, 4])
def test_conv_addend_and_bit_pattern(ops, case, out_dtype, shift):
    """segk_conv_args.addend / addend_bits: the epilogue adds a second tensor to the ROUNDED convolution result (the
    bits of a separate addition), all of it or only the elements whose bit is set in a [M, Cout/8] byte pattern
    (the ReLU pattern of a bottleneck's last unit masking the shortcut's gradient).  shift = 4: output and addend
    start four channels into wider buffers, which takes the epilogue's 4-channel store path."""
    N, H, W, Cin, Cout, R, dil = case
    M = N * H * W
    x, w = _conv_inputs(case)
    xv = to_pm(x, torch.bfloat16)[0]
    wp = ops.pack_weight(w, torch.bfloat16)
    plain = torch.empty((M, Cout), dtype=out_dtype, device=DEV)
    ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, plain)
    add_buf = T(gen.normal(13, "ad%s" % (case,), (M, Cout + 8)), out_dtype)
    addend = add_buf[:, shift:shift + Cout]
    bits = torch.from_numpy(gen.integers(13, "ab%s" % (case,), (M, Cout // 8), 0, 256).astype(np.uint8)).to(DEV)
    out_buf = torch.full((M, Cout + 8), float("nan"), dtype=out_dtype, device=DEV)
    out = out_buf[:, shift:shift + Cout]
    assert ops.conv_addend_ok([(xv, wp, R, dil)], Cout, out_dtype, addend)
    ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, out, addend=addend)
    assert torch.equal(out, plain + addend)
    out_buf.fill_(float("nan"))
    ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, out, addend=addend, addend_bits=bits)
    keep = ops.unpack_bits(bits, Cout)
    assert torch.equal(out, plain + torch.where(keep, addend, torch.zeros_like(addend)))
    assert bool(torch.isnan(out_buf[:, :shift]).all()) and bool(torch.isnan(out_buf[:, shift + Cout:]).all())
    assert 0.3 < float(keep.float().mean()) < 0.7
    with pytest.raises(ValueError):
        ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, out, addend_bits=bits)


BSTAT_CASES = [(2, 17, 15, 128, 256, 3, 2), (1, 33, 33, 256, 128, 1, 1), (16, 33, 33, 1024, 256, 1, 1),
               (4, 33, 33, 64, 512, 1, 1)]


@pytest.mark.parametrize("case", BSTAT_CASES)
@pytest.mark.parametrize("mode", ["recompute", "bits", "linear"])
def test_conv_emits_bn_backward_statistics(ops, case, mode):
    """segk_conv_args.bst_*: a data-gradient launch whose result is the output gradient of a batch norm leaves that
    layer's backward statistics (partial sums of g and g * xhat, g = the stored result through the layer's ReLU) from
    its epilogue.  Against segk_bn_bwd_stats on the tensor the same launch wrote, and against float64 PyTorch; the
    pixel counts include tiles with a ragged end."""
    from segk._lib import PM
    N, H, W, Cin, Cout, R, dil = case
    M = N * H * W
    x, w = _conv_inputs(case)
    xv = to_pm(x, torch.bfloat16)[0]
    wp = ops.pack_weight(w, torch.bfloat16)
    raw = T(gen.normal(15, "bst_raw%s" % (case,), (M, Cout)), torch.bfloat16)
    coef = torch.stack([T(gen.normal(15, "bst_mu%d" % Cout, (Cout,), std=0.3)),
                        T(np.abs(gen.normal(15, "bst_is%d" % Cout, (Cout,))) + 0.5),
                        T(gen.normal(15, "bst_sc%d" % Cout, (Cout,))),
                        T(gen.normal(15, "bst_sh%d" % Cout, (Cout,), std=0.3))]).contiguous()
    bits = None
    if mode == "bits":
        bits = torch.from_numpy(gen.integers(15, "bst_b%s" % (case,), (M, Cout // 8), 0, 256).astype(np.uint8)).to(DEV)
    relu = mode != "linear"
    plain = torch.empty((M, Cout), dtype=torch.bfloat16, device=DEV)
    ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, plain)
    out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    req = ops.BwdStatRequest(raw, coef, relu, bits)
    ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, out, bwd_stats=req)
    assert req.rows is not None and req.rows.shape[1:] == (2, Cout)
    assert torch.equal(out, plain)                       # the output itself is untouched by the request
    got = ops.bn_reduce(req.rows, float(M))
    part, cnt = ops.bn_bwd_stat_partials(out, raw, coef, PM, relu, act=bits)
    ref = ops.bn_reduce(part, cnt)
    # float64 PyTorch on the same stored values
    g = out.double()
    xf = raw.float()
    if mode == "bits":
        g = g * ops.unpack_bits(bits, Cout).double()
    elif mode == "recompute":
        g = g * (torch.addcmul(coef[3], xf, coef[2]) > 0).double()
    xhat = ((xf - coef[0]) * coef[1]).double()
    want = torch.cat([g.sum(0), (g * xhat).sum(0), torch.tensor([float(M)], dtype=torch.float64, device=DEV)])
    scale = torch.cat([g.abs().sum(0), (g * xhat).abs().sum(0), torch.ones(1, dtype=torch.float64, device=DEV)])
    for name, other in (("segk_bn_bwd_stats", ref), ("float64", want)):
        err = float(((got - other).abs() / (scale + 1e-30)).max())
        assert err < 4e-6, (name, err)
    assert float(got[2 * Cout]) == float(M)


def test_conv_backward_statistics_declined_where_the_tiles_cannot(ops):
    """fp32 output, a channel count that leaves part of a tile empty, a misaligned pre-normalisation tensor, an addend,
    or a problem that takes the 256-channel tiles (they have no registers to spare for it): the request comes back without
    rows (the caller runs the stand-alone pass) and the convolution itself is unchanged."""
    R, dil = 1, 1
    for N, H, W, Cin, Cout, out_dtype, shift in ((1, 9, 9, 64, 128, torch.float32, 0), (1, 9, 9, 64, 192, torch.bfloat16, 0),
                                                 (1, 9, 9, 64, 128, torch.bfloat16, 4),
                                                 (16, 33, 33, 64, 1024, torch.bfloat16, 0)):
        M = N * H * W
        x = T(gen.normal(15, "bsd_x%d" % N, (N, Cin, H, W)))
        xv = to_pm(x, torch.bfloat16)[0]
        w = T(gen.normal(15, "bsd_w%d" % Cout, (Cout, Cin, 1, 1), std=0.05))
        wp = ops.pack_weight(w, torch.bfloat16)
        raw_buf = torch.zeros((M, Cout + 8), dtype=torch.bfloat16, device=DEV)
        raw = raw_buf[:, shift:shift + Cout]
        coef = torch.ones((4, Cout), dtype=torch.float32, device=DEV)
        out = torch.empty((M, Cout), dtype=out_dtype, device=DEV)
        plain = torch.empty_like(out)
        ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, plain)
        req = ops.BwdStatRequest(raw, coef, True, None)
        ops.conv_fwd([(xv, wp, R, dil)], N, H, W, Cout, out, bwd_stats=req)
        assert req.rows is None
        assert torch.equal(out, plain)
    # ... or beside an addend
    Cout = 128
    raw = torch.zeros((M, Cout), dtype=torch.bfloat16, device=DEV)
    req = ops.BwdStatRequest(raw, torch.ones((4, Cout), dtype=torch.float32, device=DEV), True, None)
    w = T(gen.normal(15, "bsd_wa", (Cout, Cin, 1, 1), std=0.05))
    out = torch.empty((M, Cout), dtype=torch.bfloat16, device=DEV)
    ops.conv_fwd([(xv, ops.pack_weight(w, torch.bfloat16), R, dil)], N, H, W, Cout, out, addend=torch.ones_like(out),
                 bwd_stats=req)
    assert req.rows is None


def test_conv_addend_refused_off_the_ring_tile(ops):
    N, H, W, Cin, Cout = 1, 9, 9, 64, 64      # 64 output channels: not a multiple of 128
    x = T(gen.normal(13, "adr_x", (N, Cin, H, W)))
    w = T(gen.normal(13, "adr_w", (Cout, Cin, 1, 1), std=0.05))
    xv = to_pm(x, torch.bfloat16)[0]
    wp = ops.pack_weight(w, torch.bfloat16)
    out = torch.empty((N * H * W, Cout), dtype=torch.bfloat16, device=DEV)
    assert not ops.conv_addend_ok([(xv, wp, 1, 1)], Cout, torch.bfloat16, out)
    with pytest.raises(RuntimeError, match="addend"):
        ops.conv_fwd([(xv, wp, 1, 1)], N, H, W, Cout, out, addend=torch.zeros_like(out))


@pytest.mark.parametrize("rows", [(1, 1, 1), (16, 1, 1), (4, 2, 4), (3, 1, 9)])
@pytest.mark.parametrize("cin,cout", [(2048, 256), (256, 2048), (96, 21)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_conv_1x1_over_a_handful_of_rows(ops, rows, cin, cout, out_dtype):
    """The ASPP image-pooling branch (modeling/aspp.py:121-123): a 1 x 1 convolution whose "pixels" are the N pooled
    vectors.  fp32 problems of at most 32 rows take a kernel of their own (four output channels per workgroup, K split
    over its threads): values, bias and the statistic partials (one row) against plain PyTorch."""
    N, H, W = rows
    M = N * H * W
    x = T(gen.normal(14, "sr_x%s%d" % (rows, cin), (N, cin, H, W)))
    w = T(gen.normal(14, "sr_w%d_%d" % (cin, cout), (cout, cin, 1, 1), std=(1.0 / cin) ** 0.5))
    bias = T(gen.normal(14, "sr_b%d" % cout, (cout,)))
    xv = to_pm(x, torch.float32, ld=ops.round_up(cin, 32))[1]
    wp = ops.pack_weight(w, torch.float32)
    cpad = ops.round_up(cout, 4)
    out = torch.full((M, cpad + 4), float("nan"), dtype=out_dtype, device=DEV)
    stats = ops.conv_fwd([(xv, wp, 1, 1)], N, H, W, cout, out[:, :cpad], want_stats=True, true_ck=[cin])
    ref = F.conv2d(x.double(), w.double())
    refpm = ref.permute(0, 2, 3, 1).reshape(M, cout)
    tol = 2e-5 if out_dtype == torch.float32 else 1e-2
    assert relerr(out[:, :cout], refpm) < tol
    assert bool((out[:, cout:cpad] == 0).all()) and bool(torch.isnan(out[:, cpad:]).all())
    assert tuple(stats.shape) == (1, 2, cout)
    assert relerr(stats[0, 0], refpm.sum(0)) < 1e-5 and relerr(stats[0, 1], (refpm * refpm).sum(0)) < 1e-5
    out2 = torch.empty((M, cpad), dtype=out_dtype, device=DEV)
    ops.conv_fwd([(xv, wp, 1, 1)], N, H, W, cout, out2, bias=bias, true_ck=[cin])
    assert relerr(out2[:, :cout], refpm + bias.double()) < tol


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_conv_dgrad(ops, case, dtype, tol):
    N, H, W, Cin, Cout, R, dil = case
    x, w = _conv_inputs(case)
    x.requires_grad_(True)
    y = F.conv2d(x, w, padding=dil * (R // 2), dilation=dil)
    gy = T(gen.normal(5, "dg%s" % (case,), tuple(y.shape)))
    (gx,) = torch.autograd.grad(y, x, gy)
    cpad = ops.round_up(Cout, 32)
    gv, _ = to_pm(gy, dtype, ld=cpad)
    gfull = gv if cpad == Cout else _[:, :cpad]
    wt = ops.pack_weight(w, dtype, transpose=True)
    out = torch.empty((N * H * W, Cin), dtype=torch.float32, device=DEV)
    ops.conv_fwd([(gfull, wt, R, dil)], N, H, W, Cin, out)
    assert relerr(from_pm(out, N, H, W), gx) < tol


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("splits", [1, 3])
def test_conv_wgrad_exact_integers(ops, case, dtype, splits):
    N, H, W, Cin, Cout, R, dil = case
    x, w = _conv_inputs(case, integer=True)
    w.requires_grad_(True)
    y = F.conv2d(x, w, padding=dil * (R // 2), dilation=dil)
    gy = T(gen.integers(5, "wg%s" % (case,), tuple(y.shape), -2, 3).astype(np.float32))
    (gw,) = torch.autograd.grad(y, w, gy)
    cpad = ops.round_up(Cout, 16)
    _, xfull = to_pm(x, dtype)
    _, gfull = to_pm(gy, dtype, ld=cpad)
    got = ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin, splits=splits)
    assert torch.equal(got, gw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("splits", [32, 53])
def test_conv_wgrad_many_parts_small_gradient(ops, dtype, splits):
    """A small gradient (1x1 onto 21 channels, as the classifier's) cut into many parts: the slabs are summed by the
    tree form of segk_wgrad_finish (eight thread rows, every eighth part each); integers make any order exact.
    53 parts: thread rows with different numbers of parts; 40 input channels: a last ci group that is half empty."""
    N, H, W, Cin, Cout, R, dil = 2, 47, 41, 40, 21, 1, 1
    x = T(gen.integers(8, "wgm_x", (N, Cin, H, W), -3, 4).astype(np.float32))
    w = torch.zeros((Cout, Cin, 1, 1), device=DEV, requires_grad=True)
    y = F.conv2d(x, w)
    gy = T(gen.integers(8, "wgm_g", tuple(y.shape), -2, 3).astype(np.float32))
    (gw,) = torch.autograd.grad(y, w, gy)
    _, xfull = to_pm(x, dtype, ld=ops.round_up(Cin, 32))
    _, gfull = to_pm(gy, dtype, ld=ops.round_up(Cout, 16))
    got = ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin, splits=splits)
    assert torch.equal(got, gw)
    like = torch.empty((Cout, Cin, 1, 1), dtype=torch.bfloat16, device=DEV).to(memory_format=torch.channels_last)
    got16 = ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin, splits=splits, like=like)
    assert got16.dtype == torch.bfloat16 and torch.equal(got16.float(), gw.to(torch.bfloat16).float())


# N, H, W, Cin, Cout, R, dil, parts: the ring tile (256 x 128 of one tap per workgroup, 64-pixel chunks, chunks the tap
# does not reach are skipped).  160 / 32 input channels: a last ci tile that is a quarter full (one multiplying
# wavefront column idle); 512 output channels: two co tiles; dilation 12 on a 17 x 15 map: most chunks of the outer
# taps are skipped, rows beyond the image in the others; 5 x 5 images: several images per chunk; 7 parts of 2 x 11 x 9
# pixels: parts 4 to 6 are empty and must still write zero slabs; 1 part over 1200 pixels: 19 chunks, the last partial.
RING_WGRAD_CASES = [(2, 17, 15, 160, 256, 3, 12, 2), (1, 23, 19, 32, 512, 3, 1, 3), (9, 5, 5, 128, 256, 3, 2, 1),
                    (2, 11, 9, 64, 256, 1, 1, 7), (3, 20, 20, 256, 256, 3, 18, 1), (1, 33, 33, 96, 256, 3, 6, 4)]


@pytest.mark.parametrize("case", RING_WGRAD_CASES)
def test_conv_wgrad_ring_exact_integers(ops, case):
    N, H, W, Cin, Cout, R, dil, parts = case
    x = T(gen.integers(9, "rw_x%s" % (case,), (N, Cin, H, W), -3, 4).astype(np.float32))
    w = torch.zeros((Cout, Cin, R, R), device=DEV, requires_grad=True)
    y = F.conv2d(x, w, padding=dil * (R // 2), dilation=dil)
    gy = T(gen.integers(9, "rw_g%s" % (case,), tuple(y.shape), -2, 3).astype(np.float32))
    (gw,) = torch.autograd.grad(y, w, gy)
    _, xfull = to_pm(x, torch.bfloat16)
    _, gfull = to_pm(gy, torch.bfloat16)
    got = ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin, splits=parts, tile=384)
    assert torch.equal(got, gw)
    # the library's own choice for these shapes is the same tile; the planned part count
    assert ops.wgrad_plan(N * H * W, Cin, Cout, R, torch.bfloat16)[0] == 384
    assert torch.equal(ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin), gw)
    # and the older tile agrees
    assert torch.equal(ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin, splits=parts, tile=128), gw)


@pytest.mark.parametrize("case", [(2, 47, 45, 512, 128), (1, 65, 65, 256, 32), (2, 50, 50, 256, 96)])
@pytest.mark.parametrize("layout", ["contiguous", "channels_last bf16"])
def test_conv_wgrad_1x1_transposed_on_the_ring_tile(ops, case, layout):
    """1 x 1 gradients whose OUTPUT channels do not fill the ring tile's 256-channel side but whose input channels do
    (ResNet layer 2's 512 -> 128, the classifier's 256 -> 21 padded to 32): the operands change places and the sums are
    written back through swapped strides.  Exact on integers, in both gradient layouts; the older tile agrees."""
    N, H, W, Cin, Cout = case
    x = T(gen.integers(9, "rt_x%s" % (case,), (N, Cin, H, W), -3, 4).astype(np.float32))
    w = torch.zeros((Cout, Cin, 1, 1), device=DEV, requires_grad=True)
    y = F.conv2d(x, w)
    gy = T(gen.integers(9, "rt_g%s" % (case,), tuple(y.shape), -2, 3).astype(np.float32))
    (gw,) = torch.autograd.grad(y, w, gy)
    _, xfull = to_pm(x, torch.bfloat16)
    _, gfull = to_pm(gy, torch.bfloat16)
    like = None
    if layout != "contiguous":
        like = torch.empty((Cout, Cin, 1, 1), dtype=torch.bfloat16, device=DEV).contiguous(memory_format=torch.channels_last)
    seen = []
    real = ops.call

    def spy(name, *a):
        seen.append(name)
        return real(name, *a)
    from unittest import mock
    with mock.patch.object(ops, "call", spy):
        got = ops.conv_wgrad(xfull, gfull, N, H, W, 1, 1, Cout, Cin, like=like)
    assert "segk_conv_wgrad" in seen
    assert got.shape == gw.shape and (like is None or (got.dtype == like.dtype and got.stride() == like.stride()))
    want = gw.to(got.dtype)     # (a bf16 gradient rounds the exact integer sums once)
    assert torch.equal(got, want)
    older = ops.conv_wgrad(xfull, gfull, N, H, W, 1, 1, Cout, Cin, splits=4, tile=128, like=like)
    assert torch.equal(older, want)


@pytest.mark.parametrize("case", RING_WGRAD_CASES)
@pytest.mark.parametrize("groups", [1, 3, 8])
def test_conv_wgrad_group_exact_integers(ops, case, groups):
    """Several convolutions of one shape in one launch (segk_conv_wgrad_group): every member gets its own gradient,
    exact on integers for any number of parts; random data: bit for bit the single launch with the same parts."""
    N, H, W, Cin, Cout, R, dil, parts = case
    xs, gs, want = [], [], []
    for k in range(groups):
        x = T(gen.integers(10, "gw_x%s%d" % (case, k), (N, Cin, H, W), -3, 4).astype(np.float32))
        w = torch.zeros((Cout, Cin, R, R), device=DEV, requires_grad=True)
        y = F.conv2d(x, w, padding=dil * (R // 2), dilation=dil)
        gy = T(gen.integers(10, "gw_g%s%d" % (case, k), tuple(y.shape), -2, 3).astype(np.float32))
        want.append(torch.autograd.grad(y, w, gy)[0])
        xs.append(to_pm(x, torch.bfloat16)[1])
        gs.append(to_pm(gy, torch.bfloat16)[1])
    for sp in (parts, None):
        got = ops.conv_wgrad_group(xs, gs, N, H, W, R, dil, Cout, Cin, splits=sp)
        assert len(got) == groups
        for k in range(groups):
            assert torch.equal(got[k], want[k]), "member %d" % k
    # real-valued operands, a trimmed gradient in the weight's own bf16 channels_last layout
    rx = [T(gen.normal(10, "gw_rx%s%d" % (case, k), (N, Cin, H, W))) for k in range(groups)]
    rg = [T(gen.normal(10, "gw_rg%s%d" % (case, k), (N, Cout, H, W))) for k in range(groups)]
    rxs = [to_pm(t, torch.bfloat16)[1] for t in rx]
    rgs = [to_pm(t, torch.bfloat16)[1] for t in rg]
    like = torch.empty((Cout - 3, Cin - 5, R, R), dtype=torch.bfloat16, device=DEV).to(memory_format=torch.channels_last)
    got = ops.conv_wgrad_group(rxs, rgs, N, H, W, R, dil, Cout - 3, Cin - 5, like=like, splits=parts)
    for k in range(groups):
        one = ops.conv_wgrad(rxs[k], rgs[k], N, H, W, R, dil, Cout - 3, Cin - 5, like=like, splits=parts, tile=384)
        assert got[k].stride() == like.stride() and got[k].dtype == torch.bfloat16
        assert torch.equal(got[k], one), "member %d" % k


def test_conv_wgrad_group_refuses_other_tiles(ops):
    x = torch.zeros((64, 64), dtype=torch.bfloat16, device=DEV)
    g = torch.zeros((64, 128), dtype=torch.bfloat16, device=DEV)   # 128 output channels: not the ring tile
    with pytest.raises(RuntimeError, match="ring tile"):
        ops.conv_wgrad_group([x, x], [g, g], 1, 8, 8, 1, 1, 128, 64, splits=1)
    with pytest.raises(ValueError):
        ops.conv_wgrad_group([x] * 9, [g] * 9, 1, 8, 8, 1, 1, 128, 64, splits=1)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_conv_wgrad_random_padded_input(ops, dtype, tol):
    """Decoder shape in miniature: 304 real input channels inside a 320-wide buffer."""
    N, H, W, Cin, Cout, R, dil = 2, 19, 17, 304, 64, 3, 1
    x = T(gen.normal(6, "wgp_x", (N, Cin, H, W)))
    w = T(gen.normal(6, "wgp_w", (Cout, Cin, R, R), std=0.02)).requires_grad_(True)
    y = F.conv2d(x, w, padding=1)
    gy = T(gen.normal(6, "wgp_g", tuple(y.shape)))
    (gw,) = torch.autograd.grad(y, w, gy)
    _, xfull = to_pm(x, dtype, ld=320)
    _, gfull = to_pm(gy, dtype)
    got = ops.conv_wgrad(xfull, gfull, N, H, W, R, dil, Cout, Cin)
    assert relerr(got, gw) < tol
    # forward over the padded buffer as well
    out = torch.empty((N * H * W, Cout), dtype=torch.float32, device=DEV)
    ops.conv_fwd([(xfull, ops.pack_weight(w.detach(), dtype), R, dil)], N, H, W, Cout, out)
    assert relerr(from_pm(out, N, H, W), y.detach()) < tol


# ---------------------------------------------------------------------- BN ----
def _bn_reference(x, g, b, relu, mask, mask_scale, gy, eps=1e-5):
    x = x.clone().requires_grad_(True)
    g = g.clone().requires_grad_(True)
    b = b.clone().requires_grad_(True)
    rm = torch.zeros_like(g)
    rv = torch.ones_like(g)
    y = F.batch_norm(x, rm, rv, g, b, True, 0.1, eps)
    if relu:
        y = F.relu(y)
    if mask is not None:
        y = y * mask.float() * mask_scale
    y.backward(gy)
    return y.detach(), x.grad, g.grad, b.grad, rm, rv


@pytest.mark.parametrize("layout", ["pm", "nchw"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("C,shape", [(48, (2, 9, 11)), (256, (3, 7, 5)), (5, (4, 6, 3)), (2048, (2, 5, 3)), (1300, (2, 3, 3))])
def test_bn_train(ops, layout, dtype, tol, C, shape):
    from segk._lib import NCHW, PM
    if layout == "pm" and C % 4 and dtype == torch.bfloat16:
        pytest.skip("odd channel counts only occur in the fp32 NCHW feeder")
    N, H, W = shape
    tag = "bn%d%s" % (C, layout)
    x = T(gen.normal(7, tag + "x", (N, C, H, W), std=2.0, mean=0.7)).to(dtype).float()
    g = T(gen.uniform(7, tag + "g", (C,), 0.5, 1.5))
    b = T(gen.normal(7, tag + "b", (C,)))
    gy = T(gen.normal(7, tag + "gy", (N, C, H, W))).to(dtype).float()
    keep = T((gen.uniform(7, tag + "m", (N, C, H, W)) > 0.3).astype(np.uint8))
    y_ref, dx_ref, dg_ref, db_ref, rm_ref, rv_ref = _bn_reference(x, g, b, True, keep, 1 / 0.7, gy)
    rm = torch.zeros(C, device=DEV)
    rv = torch.ones(C, device=DEV)
    if layout == "pm":
        L = PM
        xv, _ = to_pm(x, dtype, ld=C + 8)
        gv, _ = to_pm(gy, dtype, ld=C + 4)
        mv = keep.permute(0, 2, 3, 1).reshape(-1, C).contiguous()
        yv = torch.zeros((N * H * W, C + 12), dtype=dtype, device=DEV)[:, 4:4 + C]
        dxv = torch.zeros((N * H * W, C), dtype=dtype, device=DEV)
    else:
        L = NCHW
        xv, gv, mv = x.to(dtype).contiguous(), gy.to(dtype).contiguous(), keep.contiguous()
        yv, dxv = torch.empty_like(xv), torch.empty_like(xv)
    sums = ops.bn_local_sums(xv, L)
    assert abs(float(sums[-1]) - N * H * W) < 0.5
    coef = ops.bn_coeffs(sums, g, b, 1e-5, 0.1, rm, rv)
    ops.bn_apply(xv, yv, coef, L, True, mv, 1 / 0.7)
    gs = ops.bn_bwd_local_sums(gv, xv, coef, L, True, mv, 1 / 0.7)
    ops.bn_bwd_apply(gv, xv, dxv, coef, gs, L, True, mv, 1 / 0.7)
    y = from_pm(yv, N, H, W) if layout == "pm" else yv.float()
    dx = from_pm(dxv, N, H, W) if layout == "pm" else dxv.float()
    assert relerr(y, y_ref) < tol
    assert relerr(dx, dx_ref) < tol * 5
    assert relerr(gs[:C], db_ref) < tol * 5 and relerr(gs[C:2 * C], dg_ref) < tol * 5
    assert relerr(rm, rm_ref) < 1e-5 and relerr(rv, rv_ref) < 1e-4


def test_bn_clamp_and_eval(ops):
    """legacy clamp(var, eps)^-1/2 against the fixture of the reference's _compute_mean_std, and eval mode."""
    from helpers import load_golden
    from segk._lib import NCHW
    want = load_golden("syncbn")
    C = 12
    sums = torch.cat([T(want["sum"]), T(want["ssum"]), torch.tensor([6 * 5 * 7.0], device=DEV)]).double()
    rm = T(gen.normal(20240607, "syncbn/rm", (C,), std=0.1))
    rv = T(gen.uniform(20240607, "syncbn/rv", (C,), 0.5, 1.5))
    coef = ops.bn_coeffs(sums, None, None, 1e-5, 0.1, rm, rv, legacy_clamp=True)
    assert relerr(coef[0], T(want["mean"])) < 1e-6
    assert relerr(coef[1], T(want["inv_std"])) < 1e-5
    assert relerr(rm, T(want["running_mean"])) < 1e-6 and relerr(rv, T(want["running_var"])) < 1e-5
    x = T(gen.normal(8, "bneval", (2, C, 4, 5)))
    g, b = T(gen.uniform(8, "g", (C,), 0.5, 1.5)), T(gen.normal(8, "b", (C,)))
    ce = ops.bn_coeffs_eval(rm, rv, g, b, 1e-5)
    y = ops.bn_apply(x, torch.empty_like(x), ce, NCHW, False)
    assert relerr(y, F.batch_norm(x, rm, rv, g, b, False, 0.1, 1e-5)) < 1e-5


# ---------------------------------------------------------------- resample ----
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_pool_sum(ops, dtype, tol):
    x = T(gen.normal(9, "pool", (3, 320, 9, 11))).to(dtype).float()
    xv, _ = to_pm(x, dtype, ld=352)
    got = ops.pool_sum(xv, 3, 99, 1.0 / 99)
    assert relerr(got, x.mean((2, 3))) < tol


UP_CASES = [((2, 8, 1, 1), (5, 7)), ((1, 16, 9, 11), (33, 41)), ((2, 256, 33, 33), (129, 129)),
            ((1, 8, 97, 97), (193, 193)), ((2, 12, 7, 5), (20, 13)), ((1, 4, 6, 6), (6, 6)), ((1, 4, 8, 8), (1, 1))]


@pytest.mark.parametrize("shape,size", UP_CASES)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_upsample_pm(ops, shape, size, dtype, tol):
    from segk._lib import PM
    N, C, Hi, Wi = shape
    Ho, Wo = size
    x = T(gen.normal(10, "up%s" % (shape,), shape)).to(dtype).float().requires_grad_(True)
    y_ref = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    gy = T(gen.normal(10, "upg%s" % (shape,), tuple(y_ref.shape))).to(dtype).float()
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    xv, _ = to_pm(x.detach(), dtype, ld=C + 4)
    yv = torch.zeros((N * Ho * Wo, C + 8), dtype=dtype, device=DEV)[:, 4:4 + C]
    ops.upsample_fwd(xv, N, C, Hi, Wi, yv, Ho, Wo, PM)
    assert relerr(from_pm(yv, N, Ho, Wo), y_ref.detach()) < tol
    gv, _ = to_pm(gy, dtype, ld=C + 4)
    din = torch.full((N * Hi * Wi, C + 8), 7.0, dtype=dtype, device=DEV)
    ops.upsample_bwd(gv, PM, N, C, Ho, Wo, din[:, :C + 4], C + 4, Hi, Wi)
    assert relerr(from_pm(din[:, :C], N, Hi, Wi), gx_ref) < tol * 2
    assert float(din[:, C:C + 4].abs().sum()) == 0 and float((din[:, C + 4:] - 7).abs().sum()) == 0


@pytest.mark.parametrize("K", [21, 19, 81, 8])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_upsample_to_nchw_logits(ops, K, dtype, tol):
    from segk._lib import NCHW
    N, Hi, Wi, Ho, Wo = 2, 17, 13, 65, 49
    x = T(gen.normal(11, "upl%d" % K, (N, K, Hi, Wi))).to(dtype).float().requires_grad_(True)
    y_ref = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True)
    gy = T(gen.normal(11, "uplg%d" % K, tuple(y_ref.shape)))
    (gx_ref,) = torch.autograd.grad(y_ref, x, gy)
    ld = ops.round_up(K, 32)
    xv, _ = to_pm(x.detach(), dtype, ld=ld)
    y = torch.empty((N, K, Ho, Wo), dtype=torch.float32, device=DEV)
    ops.upsample_fwd(xv, N, K, Hi, Wi, y, Ho, Wo, NCHW)
    assert relerr(y, y_ref.detach()) < tol
    din = torch.full((N * Hi * Wi, ld), 3.0, dtype=dtype, device=DEV)
    ops.upsample_bwd(gy, NCHW, N, K, Ho, Wo, din, ld, Hi, Wi)
    assert relerr(from_pm(din[:, :K], N, Hi, Wi), gx_ref) < tol * 2
    assert float(din[:, K:].abs().sum()) == 0


# -------------------------------------------------------------------- loss ----
@pytest.mark.parametrize("name,K", [("plain", 21), ("one_image_ignored", 21), ("weighted", 21), ("k81", 81), ("k8", 8)])
def test_ce_kernels(ops, name, K):
    from oracle import cases
    logit_np, target_np, weight_np = cases.ce_inputs(name, K=K)
    logit, target = T(logit_np), T(target_np)
    weight = None if weight_np is None else T(weight_np)
    sums, lse = ops.ce_fwd(logit, target, weight, 255)
    lref = logit.clone().requires_grad_(True)
    ref = F.cross_entropy(lref, target.long(), weight=weight, ignore_index=255, reduction="sum")
    assert relerr(sums[0:1], ref.detach().reshape(1)) < 1e-5
    assert relerr(lse, torch.logsumexp(logit, dim=1)) < 1e-6
    ref.backward()
    coef = torch.ones(1, device=DEV)
    dl = ops.ce_bwd(logit, target, weight, lse, 255, coef)
    assert relerr(dl, lref.grad) < 1e-5


@pytest.mark.parametrize("K", [8, 19, 21, 81, 200])
def test_argmax_and_confmat_bit_exact(ops, K):
    from oracle import cases
    gt_np, logits_np = cases.eval_inputs(min(K, 81))
    if K > 81:
        logits_np = gen.normal(12, "bigk", (3, K, 37, 41))
        gt_np = gen.labels(12, "bigk/gt", (3, 37, 41), K, border=1)
    logits_np[0, 0, 0, :5] = np.nan
    logits_np[0, 2, 0, 2:7] = np.nan
    pred_np = np.argmax(logits_np, axis=1)
    pred = ops.argmax(T(logits_np))
    assert np.array_equal(pred.cpu().numpy(), pred_np)
    cm = torch.zeros(K * K + 1, dtype=torch.int64, device=DEV)
    ops.confmat_accumulate(T(gt_np), pred, K, cm)
    ops.confmat_accumulate(T(gt_np), pred, K, cm)
    want = 2 * ops_ref.confusion_matrix(gt_np, pred_np, K)
    assert np.array_equal(cm[:-1].cpu().numpy().reshape(K, K), want)
    assert int(cm[-1]) == 0
    # int64 labels and an out-of-range prediction
    cm.zero_()
    bad = pred.clone()
    bad.view(-1)[7] = K + 3
    gt_i = T(np.where((gt_np >= 0) & (gt_np < K), gt_np, -1).astype(np.int64))
    ops.confmat_accumulate(gt_i, bad, K, cm)
    valid7 = 0 <= gt_np.reshape(-1)[7] < K
    assert int(cm[-1]) == (1 if valid7 else 0)


# ------------------------------------------------- batched matrix products ----
GEMM_CASES = [  # batch, M, N, K (true), Kpad
    (3, 63, 63, 8, 32),        # position-attention energies of a small map
    (2, 200, 72, 100, 128),    # ragged everything
    (2, 961, 256, 992, 992),   # config-E sized rows, full 128-byte K rows
    (1, 300, 512, 64, 64),
]


def _int_operand(tag, shape, kpad, dtype):
    """Small integers (exact in bf16 and in any fp32 summation order) in a zero-padded [.., kpad] buffer."""
    v = gen.integers(3, tag, shape, -3, 4).astype(np.float32)
    buf = torch.zeros(shape[:-1] + (kpad,), dtype=dtype, device=DEV)
    buf[..., :shape[-1]] = T(v, dtype)
    return buf, T(v)


@pytest.mark.parametrize("case", GEMM_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_nt_exact_integers(ops, case, dtype):
    b, M, N, K, Kp = case
    A, a = _int_operand("gA%d" % M, (b, M, K), Kp, dtype)
    B, bm = _int_operand("gB%d" % N, (b, N, K), Kp, dtype)
    ldc = (N + 3) // 4 * 4 + 8
    out = torch.full((b, M, ldc), 7.0, dtype=torch.float32, device=DEV)
    ops.gemm_nt([(A, B)], out[:, :, :(N + 3) // 4 * 4])
    assert torch.equal(out[:, :, :N], torch.bmm(a, bm.transpose(1, 2)))
    assert float(out[:, :, (N + 3) // 4 * 4:].min()) == 7.0  # nothing written beyond the rounded width


def test_gemm_nt_terms_shared_operand_and_strided_views(ops):
    b, M, N, K = 3, 70, 40, 64
    big = T(gen.integers(3, "gbig", (b * M, 3 * K), -2, 3).astype(np.float32))
    A1, A2 = big[:, :K].unflatten(0, (b, M)), big[:, 2 * K:].unflatten(0, (b, M))  # column slices, batched view
    B1 = T(gen.integers(3, "gb1", (b, N, K), -2, 3).astype(np.float32))
    B2 = T(gen.integers(3, "gb2", (1, N, K), -2, 3).astype(np.float32))            # shared by the batch
    out = torch.empty((b, M, N), dtype=torch.float32, device=DEV)
    ops.gemm_nt([(A1, B1), (A2, B2)], out)
    want = torch.bmm(A1, B1.transpose(1, 2)) + torch.matmul(A2, B2[0].t())
    assert torch.equal(out, want)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1e-2)])
def test_gemm_nt_random(ops, dtype, tol):
    b, M, N, K = 2, 130, 96, 160
    A = T(gen.normal(4, "rA", (b, M, K)), dtype)
    B = T(gen.normal(4, "rB", (b, N, K)), dtype)
    out = torch.empty((b, M, N), dtype=dtype, device=DEV)
    ops.gemm_nt([(A, B)], out)
    assert relerr(out, torch.bmm(A.float(), B.float().transpose(1, 2))) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("splits", [1, 3])
def test_gemm_tn_exact_integers(ops, dtype, splits):
    b, K, M, N = 3, 63, 64, 48          # padded widths; true rows/cols below
    rows, cols = 63, 40
    A = torch.zeros((b, K, M), dtype=dtype, device=DEV)
    B = torch.zeros((b, K, N), dtype=dtype, device=DEV)
    a = T(gen.integers(5, "tA", (b, K, rows), -3, 4).astype(np.float32))
    bm = T(gen.integers(5, "tB", (b, K, cols), -3, 4).astype(np.float32))
    A[:, :, :rows], B[:, :, :cols] = a.to(dtype), bm.to(dtype)
    wide = torch.full((b, rows, cols + 8), 7.0, dtype=dtype, device=DEV)
    ops.gemm_tn(A, B, wide[:, :, :cols], rows=rows, cols=cols, splits=splits)
    assert torch.equal(wide[:, :, :cols].float(), torch.bmm(a.transpose(1, 2), bm))
    assert float(wide[:, :, cols:].float().min()) == 7.0
    if splits == 1:
        dense = torch.empty((b, M, N), dtype=torch.float32, device=DEV)   # written by the kernel itself
        ops.gemm_tn(A, B, dense)
        assert torch.equal(dense[:, :rows, :cols], torch.bmm(a.transpose(1, 2), bm))
        assert float(dense[:, rows:].abs().sum()) == 0 and float(dense[:, :, cols:].abs().sum()) == 0


@pytest.mark.parametrize("reversed_", [False, True])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
@pytest.mark.parametrize("rows,cols,ld", [(7, 63, 64), (5, 961, 992), (3, 2048, 2048), (4, 300, 320)])
def test_softmax_rows_and_gradient(ops, reversed_, dtype, tol, rows, cols, ld):
    e = torch.full((rows, ld), float("nan"), dtype=torch.float32, device=DEV)   # padding must not be read
    ev = T(3.0 * gen.normal(6, "sm%d" % cols, (rows, cols)))
    e[:, :cols] = ev
    att = torch.full((rows, ld), 9.0, dtype=dtype, device=DEV)
    ops.softmax_rows(e, cols, att, reversed_=reversed_)
    z = (ev.max(-1, keepdim=True)[0] - ev) if reversed_ else ev
    want = torch.softmax(z.double(), -1)
    assert relerr(att[:, :cols], want) < tol
    assert float(att[:, cols:].float().abs().sum()) == 0
    # gradient w.r.t. e for upstream gradient g on att (the stored, possibly rounded, att is what backward uses)
    g = torch.zeros((rows, ld), dtype=torch.float32, device=DEV)
    gv = T(gen.normal(6, "smg%d" % cols, (rows, cols)))
    g[:, :cols] = gv
    de = torch.full((rows, ld), 9.0, dtype=dtype, device=DEV)
    ops.softmax_rows_bwd(att, g, cols, de, -1.0 if reversed_ else 1.0)
    a = att[:, :cols].double()
    want_de = a * (gv.double() - (a * gv.double()).sum(-1, keepdim=True)) * (-1.0 if reversed_ else 1.0)
    assert relerr(de[:, :cols], want_de) < max(tol, 1e-5)
    assert float(de[:, cols:].float().abs().sum()) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_transpose_batched_padded(ops, dtype):
    b, R, C = 3, 63, 200
    wide = T(gen.normal(7, "tr", (b * R, C + 24)), dtype)
    src = wide[:, 8:8 + C].unflatten(0, (b, R))
    out = torch.full((b, C, 96), 5.0, dtype=dtype, device=DEV)
    ops.transpose(src, out)
    assert torch.equal(out[:, :, :R], src.transpose(1, 2))
    assert float(out[:, :, R:].float().abs().sum()) == 0


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-6), (torch.bfloat16, 1e-2)])
def test_lincomb_and_dot(ops, dtype, tol):
    M, C = 500, 160
    x = T(gen.normal(8, "lx", (M, C + 32)), dtype)[:, 16:16 + C]
    y = T(gen.normal(8, "ly", (M, C)), dtype)
    z = T(gen.normal(8, "lz", (M, C)), dtype)
    s = T(np.array([0.7], dtype=np.float32))
    out = torch.empty((M, C), dtype=torch.float32, device=DEV)
    ops.lincomb(out, [(x, 1.0, s), (y, 2.0, None), (z, -1.0, s)])
    want = 0.7 * x.double() + 2.0 * y.double() - 0.7 * z.double()
    assert relerr(out, want) < 1e-6
    out2 = torch.empty((M, C), dtype=dtype, device=DEV)
    ops.lincomb(out2, [(x, 1.0, s)])
    assert relerr(out2, 0.7 * x.double()) < tol
    d = ops.dot(x, y)
    assert abs(float(d) - float((x.double() * y.double()).sum())) < 1e-5 * float((x.double() * y.double()).abs().sum())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_bn_residual_relu_fused(ops, dtype, tol):
    """relu(bn(x) + res) and its three gradients (x, res, gamma/beta sums) against autograd in fp64."""
    from segk._lib import PM
    M, C = 3 * 7 * 5, 48
    x = T(gen.normal(9, "brx", (M, C), std=2.0, mean=0.3), dtype)
    res = T(gen.normal(9, "brr", (M, C)), dtype)
    g = T(gen.normal(9, "brg", (M, C)), dtype)
    gamma = T(gen.uniform(9, "brga", (C,), 0.5, 1.5))
    beta = T(gen.normal(9, "brb", (C,), std=0.3))
    coef = ops.bn_coeffs(ops.bn_local_sums(x, PM), gamma, beta, 1e-5, 0.1, None, None)
    y = torch.empty_like(x)
    ops.bn_apply(x, y, coef, PM, True, res=res)
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5) + rr)
    assert relerr(y, yr.detach()) < tol
    # backward on the kernel's own activation pattern
    yr2 = (F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5) + rr) * (y > 0).double()
    yr2.backward(g.double())
    sums = ops.bn_bwd_local_sums(g, x, coef, PM, True, act=y)
    dx, dres = torch.empty_like(x), torch.empty_like(x)
    ops.bn_bwd_apply(g, x, dx, coef, sums, PM, True, act=y, dres=dres)
    assert relerr(dx, xr.grad) < tol
    assert relerr(dres, rr.grad) < tol
    assert relerr(sums[:C], br.grad) < tol
    xhat_scale = relerr(sums[C:2 * C] * coef[1].double(), gr.grad * coef[1].double())  # sum g*xhat = dgamma
    assert xhat_scale < tol
    # the stored ReLU bit pattern (one bit per element beside the output) in place of the output itself: the same
    # output, the bits of y > 0, and bit-identical backward results
    bits = ops.relu_bits_like(y)
    assert bits is not None and tuple(bits.shape) == (M, C // 8)
    y2 = torch.empty_like(x)
    ops.bn_apply(x, y2, coef, PM, True, res=res, relu_bits=bits)
    assert torch.equal(y2, y)
    want = ((y > 0).reshape(M, C // 8, 8).to(torch.int32) << torch.arange(8, device=y.device, dtype=torch.int32)).sum(-1)
    assert torch.equal(bits.to(torch.int32), want)
    sums2 = ops.bn_bwd_local_sums(g, x, coef, PM, True, act=bits)
    dx2, dres2 = torch.empty_like(x), torch.empty_like(x)
    ops.bn_bwd_apply(g, x, dx2, coef, sums2, PM, True, act=bits, dres=dres2)
    assert torch.equal(sums2, sums) and torch.equal(dx2, dx) and torch.equal(dres2, dres)


@pytest.mark.parametrize("p", [0.5, 0.1])
def test_dropout_mask_statistics_and_seeding(ops, p):
    """The one-launch keep-mask of nn.Dropout (modeling/decoder.py:146,151): bytes 0 / 1, kept with probability 1 - p
    (five standard deviations), neighbours and far-apart elements independent, fixed by torch.manual_seed, new on
    every call.  1 000 003 elements: the last group of sixteen is partial."""
    n = 1000003
    torch.manual_seed(1234)
    a = ops.dropout_mask((n,), p, DEV)
    b = ops.dropout_mask((n,), p, DEV)
    torch.manual_seed(1234)
    a2 = ops.dropout_mask((n,), p, DEV)
    assert a.dtype == torch.uint8 and int(a.max()) == 1 and int(a.min()) == 0
    assert torch.equal(a, a2) and not torch.equal(a, b)
    q = 1.0 - p
    sd = (q * p / n) ** 0.5
    for m in (a, b):
        assert abs(float(m.float().mean()) - q) < 5 * sd
    af, bf = a.float(), b.float()
    for lag in (1, 4, 16, 4096):
        joint = float((af[:-lag] * af[lag:]).mean())
        assert abs(joint - q * q) < 5 * (q * q * (1 - q * q) / (n - lag)) ** 0.5, (lag, joint)
    assert abs(float((af * bf).mean()) - q * q) < 5 * (q * q * (1 - q * q) / n) ** 0.5
    # a two-dimensional shape is the same stream laid out row after row
    torch.manual_seed(1234)
    c = ops.dropout_mask((1000, 1000), p, DEV)
    assert torch.equal(c.reshape(-1), a[:1000000])


def test_debug_labels_raises_on_out_of_range_labels():
    """config.set_debug_labels: a label that is neither the ignore index nor a class raises, like the reference's
    nn.CrossEntropyLoss does (/root/reference/utils/loss.py:25-30); without the switch the kernels ignore it."""
    from segk import config
    from segk.utils.loss import SegmentationLosses
    K = 5
    logit = torch.randn(2, K, 6, 7, device="cuda")
    target = torch.randint(0, K, (2, 6, 7), device="cuda").float()
    target[0, 0, 0] = 255.0
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    clean = float(crit(logit, target))
    bad = target.clone()
    bad[1, 2, 3] = float(K + 1)
    assert np.isfinite(float(crit(logit, bad)))      # ignored silently by default
    config.set_debug_labels(True)
    try:
        assert float(crit(logit, target)) == clean
        with pytest.raises(ValueError, match="neither the ignore index"):
            crit(logit, bad)
    finally:
        config.set_debug_labels(False)
