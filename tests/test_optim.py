"""Device SGD (segk.optim.SGD) against torch.optim.SGD and the oracle restatement."""
import numpy as np
import pytest
import torch

from oracle import ops_ref


def test_oracle_sgd_matches_torch_cpu():
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal(1000).astype(np.float32)
    p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
    opt = torch.optim.SGD([p], lr=0.05, momentum=0.9, weight_decay=5e-4)
    w, buf = w0.copy(), np.zeros_like(w0)
    for it in range(4):
        g = rng.standard_normal(1000).astype(np.float32)
        opt.param_groups[0]["lr"] = ops_ref.poly_lr(0.05, it, 10)
        p.grad = torch.from_numpy(g.copy())
        opt.step()
        w, buf = ops_ref.sgd_step(w, g, buf, ops_ref.poly_lr(0.05, it, 10))
        np.testing.assert_allclose(w, p.detach().numpy(), rtol=2e-6, atol=1e-7)


SHAPES = [(7,), (1,), (65536 + 3,), (256, 48, 3, 3), (21, 256, 1, 1), (3, 70001)]


def _grads(shapes, seed, dev, dtype=torch.float32):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn(s, generator=g).to(dev).to(dtype) for s in shapes]


@pytest.mark.gpu
def test_sgd_fp32_matches_torch():
    from segk.optim import SGD
    dev = "cuda"
    init = _grads(SHAPES, 0, dev)
    mine = [torch.nn.Parameter(t.clone()) for t in init]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    # one convolution weight in channels_last, as the feeder holds them
    mine[3].data = mine[3].data.contiguous(memory_format=torch.channels_last)
    skip = 4  # this parameter never receives a gradient
    o1 = SGD([{"params": mine[:3], "lr": 0.007}, {"params": mine[3:], "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
    o2 = torch.optim.SGD([{"params": ref[:3], "lr": 0.007}, {"params": ref[3:], "lr": 0.07}], momentum=0.9,
                         weight_decay=5e-4)
    for it in range(4):
        lr = ops_ref.poly_lr(0.007, it, 10)
        for o in (o1, o2):
            o.param_groups[0]["lr"] = lr
            o.param_groups[1]["lr"] = 10 * lr
            o.zero_grad(set_to_none=True)
        gs = _grads(SHAPES, 10 + it, dev)
        for k, (a, b, g) in enumerate(zip(mine, ref, gs)):
            if k == skip:
                continue
            b.grad = g.clone()
            a.grad = g.clone().contiguous(memory_format=torch.channels_last) if k == 3 else g.clone()
        o1.step()
        o2.step()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(mine, ref)):
        torch.testing.assert_close(a.detach().contiguous(), b.detach(), rtol=2e-6, atol=1e-7)
    assert torch.equal(mine[skip].detach(), init[skip])


@pytest.mark.gpu
def test_sgd_bf16_parameters_keep_fp32_master():
    from segk.optim import SGD
    dev = "cuda"
    init = _grads(SHAPES, 1, dev)
    mine = [torch.nn.Parameter(t.clone()) for t in init]
    ref = [torch.nn.Parameter(t.clone()) for t in init]
    o1 = SGD(mine, lr=0.05, momentum=0.9, weight_decay=5e-4, low_precision=mine)
    o2 = torch.optim.SGD(ref, lr=0.05, momentum=0.9, weight_decay=5e-4)
    assert all(p.dtype == torch.bfloat16 for p in mine)
    for it in range(3):
        gs = _grads(SHAPES, 20 + it, dev, torch.bfloat16)
        for a, b, g in zip(mine, ref, gs):
            a.grad = g.clone()
            b.grad = g.float()
        o1.step()
        o2.step()
    torch.cuda.synchronize()
    for a, b in zip(mine, ref):
        master = o1.state[a]["master"].reshape(a.shape)
        torch.testing.assert_close(master, b.detach(), rtol=2e-6, atol=1e-7)
        assert torch.equal(a.detach(), master.to(torch.bfloat16))   # the bf16 copy is the rounded master

    lin = torch.nn.Linear(4, 4, bias=False).to(dev)
    o3 = SGD(lin.parameters(), lr=0.1, low_precision=[lin.weight])
    sd = o3.full_precision_state_dict(lin)
    assert sd["weight"].dtype == torch.float32 and lin.weight.dtype == torch.bfloat16


@pytest.mark.gpu
@pytest.mark.parametrize("low", [False, True])
def test_sgd_checkpoint_round_trip_and_interchange_with_torch(low):
    """step, save, load into FRESH parameters and optimizers, step again (the resume of /root/reference/train.py:
    112-131): the resumed segk.optim.SGD follows an uninterrupted torch.optim.SGD, its checkpoint loads into
    torch.optim.SGD and torch's loads into it."""
    import io
    from segk.optim import SGD
    dev = "cuda"
    init = _grads(SHAPES, 5, dev)
    gdt = torch.bfloat16 if low else torch.float32

    def fresh(values, cls, **kw):
        ps = [torch.nn.Parameter(t.clone()) for t in values]
        if cls is SGD:
            ps[3].data = ps[3].data.contiguous(memory_format=torch.channels_last)
            if low:
                kw["low_precision"] = ps
        return ps, cls([{"params": ps[:3], "lr": 0.007}, {"params": ps[3:], "lr": 0.07}], momentum=0.9,
                       weight_decay=5e-4, **kw)

    def run(ps, opt, its):
        for it in its:
            for k, (p, g) in enumerate(zip(ps, _grads(SHAPES, 40 + it, dev, gdt))):
                g = g.to(p.dtype) if p.dtype == torch.bfloat16 else g.float()
                p.grad = g.contiguous(memory_format=torch.channels_last) if (k == 3 and p.dim() == 4 and
                                                                             not p.is_contiguous()) else g
            opt.step()

    def values(ps, opt):
        if isinstance(opt, SGD) and low:
            return [opt.state[p]["master"].reshape(-1).clone() if p.is_contiguous() else
                    SGD._shaped(opt.state[p]["master"], p).contiguous().reshape(-1) for p in ps]
        return [p.detach().contiguous().reshape(-1).clone() for p in ps]

    ref, oref = fresh(init, torch.optim.SGD)
    run(ref, oref, range(4))                                     # uninterrupted torch run: 4 steps

    a, oa = fresh(init, SGD)
    run(a, oa, range(2))
    buf = io.BytesIO()
    torch.save({"w": values(a, oa), "optimizer": oa.state_dict()}, buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=True)
    sd = ck["optimizer"]
    assert all(v["momentum_buffer"].shape == p.shape and v["momentum_buffer"].dtype == torch.float32
               for v, p in zip(sd["state"].values(), a))
    assert all(("master" in v) == low for v in sd["state"].values())

    # (1) resume segk from its own checkpoint, in fresh objects
    b, ob = fresh([w.reshape(s) for w, s in zip(ck["w"], SHAPES)], SGD)
    ob.load_state_dict(sd)
    run(b, ob, range(2, 4))
    torch.cuda.synchronize()
    for x, y in zip(values(b, ob), values(ref, oref)):
        torch.testing.assert_close(x, y, rtol=4e-6, atol=2e-7)
    if low:
        for p in b:
            assert p.dtype == torch.bfloat16
            assert torch.equal(p.detach(), SGD._shaped(ob.state[p]["master"], p).to(torch.bfloat16))

    # (2) the same checkpoint resumes torch.optim.SGD
    c, oc = fresh([w.reshape(s) for w, s in zip(ck["w"], SHAPES)], torch.optim.SGD)
    oc.load_state_dict(sd)
    run(c, oc, range(2, 4))
    for x, y in zip(values(c, oc), values(ref, oref)):
        torch.testing.assert_close(x, y, rtol=4e-6, atol=2e-7)

    # (3) a torch.optim.SGD checkpoint resumes segk, also after segk has already stepped (tables are rebuilt)
    d, od = fresh(init, torch.optim.SGD)
    run(d, od, range(2))
    e, oe = fresh(init, SGD)
    run(e, oe, range(1))
    for p, q in zip(e, d):
        if low:
            SGD._shaped(oe.state[p]["master"], p).copy_(q.detach())
        p.data.copy_(q.detach())
    tsd = od.state_dict()
    oe.load_state_dict(tsd)
    run(e, oe, range(2, 4))
    torch.cuda.synchronize()
    for x, y in zip(values(e, oe), values(ref, oref)):
        torch.testing.assert_close(x, y, rtol=(1e-2 if low else 4e-6), atol=(1e-2 if low else 2e-7))


@pytest.mark.gpu
def test_sgd_reads_flat_gradient_buckets():
    """reduce_flat() + step(flat_grads=...) on one rank equals the plain step (mixed fp32 / bf16 parameters, one
    channels_last, one without a gradient)."""
    from segk.optim import SGD
    from segk.parallel import GradientBuckets
    dev = "cuda"
    init = _grads(SHAPES, 2, dev)

    def make():
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        ps[3].data = ps[3].data.contiguous(memory_format=torch.channels_last)
        return ps, SGD(ps, lr=0.05, momentum=0.9, weight_decay=5e-4, low_precision=[ps[3], ps[5]])

    a, oa = make()
    b, ob = make()
    buckets = GradientBuckets(b, bucket_mb=0.5)
    assert len(buckets.buckets) > 1
    for it in range(3):
        gs = _grads(SHAPES, 30 + it, dev)
        for ps in (a, b):
            for k, (p, g) in enumerate(zip(ps, gs)):
                if k == 1:
                    p.grad = None
                    continue
                g = g.to(p.dtype)
                p.grad = g.contiguous(memory_format=torch.channels_last) if k == 3 else g.clone()
        oa.step()
        ob.step(flat_grads=buckets.reduce_flat())
    torch.cuda.synchronize()
    for k, (p, q) in enumerate(zip(a, b)):
        if k == 1:
            # no gradient: the plain step skips the parameter, the flat step sees a zero gradient and decays it
            assert torch.equal(p.detach(), init[1])
            continue
        assert torch.equal(p.detach(), q.detach()), k
        assert torch.equal(oa.state[p]["master"], ob.state[q]["master"]), k


@pytest.mark.gpu
def test_c_abi_communicator_single_rank():
    """segk_comm_*: RCCL opened on demand; on one rank an in-place SUM all-reduce leaves the buffers unchanged."""
    import ctypes
    from segk import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    cid = ctypes.create_string_buffer(128)
    _lib.call("segk_comm_unique_id", cid)
    _lib.call("segk_comm_init", cid, 0, 1)
    try:
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        f = torch.arange(1025, dtype=torch.float32, device="cuda")
        d = torch.arange(513, dtype=torch.float64, device="cuda") / 7
        i = torch.arange(21 * 21 + 1, dtype=torch.int64, device="cuda") * (1 << 40)
        f0, d0, i0 = f.clone(), d.clone(), i.clone()
        _lib.call("segk_allreduce_f32", ctypes.c_void_p(f.data_ptr()), f.numel(), st)
        _lib.call("segk_allreduce_f64", ctypes.c_void_p(d.data_ptr()), d.numel(), st)
        _lib.call("segk_allreduce_i64", ctypes.c_void_p(i.data_ptr()), i.numel(), st)
        torch.cuda.synchronize()
        assert torch.equal(f, f0) and torch.equal(d, d0) and torch.equal(i, i0)
        with pytest.raises(_lib.SegkError):
            _lib.call("segk_comm_init", cid, 0, 1)      # one communicator per process
    finally:
        _lib.call("segk_comm_destroy")
    with pytest.raises(_lib.SegkError):
        _lib.call("segk_allreduce_f32", ctypes.c_void_p(f.data_ptr()), f.numel(), st)
    assert lib is not None


@pytest.mark.gpu
def test_kept_weight_packs_follow_the_optimizer():
    """Convolution weights of a segk.optim.SGD keep their packed operands (ops.pack_weight returns the kept tensor
    while the weight's version counter stands); the optimizer's step, which writes the weights through raw pointers,
    renews all of them in one launch (segk_pack_weights), bit-identical to packing each weight on its own: fp32 OIHW
    3x3 and 1x1 (the head), bf16 channels_last 3x3 and 1x1 (the feeder), forward and transposed operands, channel
    counts that are no multiples of 32."""
    from segk import ops
    from segk.optim import SGD
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(11)
    shapes = [(48, 304, 3, 3), (21, 256, 1, 1), (256, 128, 3, 3), (128, 1024, 1, 1), (40, 24, 3, 3)]
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]
    for k in (2, 3):
        ps[k].data = ps[k].data.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    opt = SGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-4)
    uses = [(p, dt, tr) for p in ps for dt in ((torch.bfloat16,) if p.dtype == torch.bfloat16 else (torch.float32, torch.bfloat16))
            for tr in (False, True)]
    kept = [ops.pack_weight(p.detach(), dt, transpose=tr) for p, dt, tr in uses]
    for (p, dt, tr), k in zip(uses, kept):
        assert ops.pack_weight(p.detach(), dt, transpose=tr) is k          # unchanged weight: the kept operand
        assert torch.equal(k, ops.pack_weight(p.detach().clone(), dt, transpose=tr))
    for step in range(2):
        for p in ps:
            p.grad = torch.randn(p.shape, generator=g).to(dev).to(p.dtype).contiguous(
                memory_format=torch.channels_last if p.dtype == torch.bfloat16 else torch.contiguous_format)
        opt.step()
        for (p, dt, tr), k in zip(uses, kept):
            got = ops.pack_weight(p.detach(), dt, transpose=tr)
            assert got is k and got.data_ptr() == k.data_ptr()
            assert torch.equal(got, ops.pack_weight(p.detach().clone(), dt, transpose=tr)), (tuple(p.shape), dt, tr, step)
    # a torch operation moves the version counter: packed again on use, into the same tensor
    with torch.no_grad():
        ps[0].mul_(0.5)
    got = ops.pack_weight(ps[0].detach(), torch.bfloat16)
    assert torch.equal(got, ops.pack_weight(ps[0].detach().clone(), torch.bfloat16))
    # in-place changes through .data move no counter: the caller has to say so
    ps[1].data.add_(1.0)
    ops.invalidate_weight_packs()
    got = ops.pack_weight(ps[1].detach(), torch.float32, transpose=True)
    assert torch.equal(got, ops.pack_weight(ps[1].detach().clone(), torch.float32, transpose=True))


@pytest.mark.gpu
def test_debug_packs_catches_a_weight_changed_through_data():
    """config.set_debug_packs: a registered weight written through `.data` (no version counter moves) is caught at
    the next use of its kept operand instead of multiplying with stale values."""
    from segk import config, ops
    from segk.optim import SGD
    dev = "cuda"
    p = torch.nn.Parameter(torch.randn(32, 64, 3, 3, device=dev))
    opt = SGD([p], lr=0.1)
    kept = ops.pack_weight(p.detach(), torch.bfloat16)
    assert ops.pack_weight(p.detach(), torch.bfloat16) is kept
    p.data.mul_(2.0)
    config.set_debug_packs(True)
    try:
        with pytest.raises(RuntimeError, match="no longer matches"):
            ops.pack_weight(p.detach(), torch.bfloat16)
        ops.invalidate_weight_packs()
        fresh = ops.pack_weight(p.detach(), torch.bfloat16)
        assert torch.equal(fresh, ops.pack_weight(p.detach().clone(), torch.bfloat16))
        assert ops.pack_weight(p.detach(), torch.bfloat16) is fresh     # consistent again: no complaint
    finally:
        config.set_debug_packs(False)
    del opt


@pytest.mark.gpu
def test_torch_sgd_checkpoint_keeps_exact_masters_restored_before():
    """Resume order of tests/test_checkpoint.py with a torch.optim.SGD-written optimizer dict (no `master` entries):
    load_full_precision_state_dict restores the exact fp32 masters first; load_state_dict must then leave them alone
    (the bf16 copies still are their roundings) instead of overwriting them with the rounded values.  When the
    weights were loaded into the parameters only, the masters follow the parameters."""
    from segk.optim import SGD
    dev = "cuda"
    torch.manual_seed(4)
    lin = torch.nn.Conv2d(16, 32, 3, bias=False).to(dev)
    exact = torch.randn_like(lin.weight) * 1.2345
    topt = torch.optim.SGD([torch.nn.Parameter(exact.clone())], lr=0.1, momentum=0.9)
    topt.param_groups[0]["params"][0].grad = torch.ones_like(exact)
    topt.step()
    tsd = topt.state_dict()
    opt = SGD(lin.parameters(), lr=0.1, momentum=0.9, low_precision=[lin.weight])
    assert lin.weight.dtype == torch.bfloat16
    opt.load_full_precision_state_dict(lin, {"weight": exact.clone()})
    master = SGD._shaped(opt.state[lin.weight]["master"], lin.weight)
    assert torch.equal(master, exact)
    seen = []
    opt.register_load_state_dict_pre_hook(lambda o, sd: seen.append("pre"))
    opt.register_load_state_dict_post_hook(lambda o: seen.append("post"))
    opt.load_state_dict(tsd)
    assert seen == ["pre", "post"]
    assert torch.equal(master, exact)                                     # not replaced by its bf16 rounding
    assert torch.equal(lin.weight.detach(), exact.to(torch.bfloat16))
    mom = SGD._shaped(opt.state[lin.weight]["momentum_buffer"], lin.weight)
    assert torch.equal(mom, tsd["state"][0]["momentum_buffer"])
    # weights loaded into the bf16 parameter only: the master is stale and is re-derived from the parameter
    other = (torch.randn_like(exact) * 0.5).to(torch.bfloat16)
    lin.weight.data.copy_(other)
    opt.load_state_dict(tsd)
    assert torch.equal(master, other.float())
