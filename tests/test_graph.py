"""segk.graph.GraphedTrainStep: the training iteration of train.py:139-146 recorded once as a HIP graph and replayed.

What must hold: a replayed iteration does what the same iteration issued launch by launch does (losses, weights,
batch-norm running statistics), the poly schedule of utils/lr_scheduler.py:50-73 keeps reaching the recorded
optimizer launch (segk_sgd_step_dev reads lr / momentum / weight decay from device memory), and a new batch copied
into the fixed buffers is what the replay trains on."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def test_recorded_sgd_follows_the_schedule():
    """segk.optim.SGD.step() recorded in a graph, replayed with three different learning rates and a changed momentum,
    against torch.optim.SGD stepping eagerly on the same gradients (fp32 and bf16-held parameters)."""
    from segk.optim import SGD
    dev = "cuda:0"
    torch.manual_seed(3)
    shapes = [(70000,), (33, 7, 3, 3), (5,)]
    ps = [nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    low = nn.Parameter(torch.randn(4096 + 3, device=dev))
    ref = [nn.Parameter(p.detach().clone()) for p in ps + [low]]
    grads = [torch.randn_like(p) for p in ref]
    opt = SGD([{"params": ps[:2]}, {"params": ps[2:] + [low], "lr": 0.07}], lr=0.007, momentum=0.9, weight_decay=5e-4,
              low_precision=[low])
    topt = torch.optim.SGD([{"params": ref[:2]}, {"params": ref[2:], "lr": 0.07}], lr=0.007, momentum=0.9,
                           weight_decay=5e-4)
    static = [torch.zeros_like(p) for p in ps + [low]]   # gradients at fixed addresses (bf16 for the bf16-held one)
    for p, g in zip(ps + [low], static):
        p.grad = g
    for s, g, r in zip(static, grads, ref):    # one eager step first: the optimizer builds its tables in it
        s.copy_(g * 0.5)
        r.grad = s.detach().float().clone()
    opt.step()
    topt.step()
    opt.sync_hyper()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        opt.step()
    for it, (lr0, lr1, mom) in enumerate([(0.007, 0.07, 0.9), (0.003, 0.03, 0.9), (0.001, 0.01, 0.5)]):
        for g, tg, sg in zip(opt.param_groups, topt.param_groups, (lr0, lr1)):
            g["lr"] = tg["lr"] = sg
            g["momentum"] = tg["momentum"] = mom
        for s, g, r in zip(static, grads, ref):
            gi = g * (it + 1)
            s.copy_(gi)                              # rounds to bf16 for the bf16-held parameter
            r.grad = s.detach().float().clone()      # the reference steps on exactly the gradient the kernel reads
        opt.sync_hyper()
        graph.replay()
        topt.step()
    torch.cuda.synchronize()
    for p, r in zip(ps, ref[:3]):
        assert torch.allclose(p, r, rtol=2e-6, atol=2e-7), float((p - r).abs().max())
    master = opt.state[low]["master"]
    assert torch.allclose(master, ref[3].detach(), rtol=2e-6, atol=2e-7)
    assert torch.equal(low.detach(), master.to(torch.bfloat16))


def _small_model(seed):
    from segk.modeling.deeplab import DeepLab
    torch.manual_seed(seed)
    model = DeepLab(backbone="resnet", output_stride=16, num_classes=6, sync_bn=True).to("cuda:0")
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0   # eager and replayed iterations draw different random numbers; the comparison needs none
    return model.train()


def _trainer(model):
    from segk.optim import SGD
    from segk.utils.loss import SegmentationLosses
    from segk.utils.lr_scheduler import LR_Scheduler
    groups = [{"params": list(model.get_1x_lr_params()), "lr": 0.01},
              {"params": list(model.get_10x_lr_params()), "lr": 0.1}]
    opt = SGD(groups, momentum=0.9, weight_decay=5e-4)
    crit = SegmentationLosses(weight=None, cuda=True).build_loss("ce")
    sched = LR_Scheduler("poly", 0.01, 1, 6, quiet=True)   # 6 iterations: the rate falls from 0.01 to 0.002
    return opt, crit, sched


@pytest.fixture
def deterministic_feeder():
    """MIOpen's default fp32 convolution gradients differ from run to run (two eager runs of the same tiny training
    part by 1 % in the loss after one step); its deterministic choices make the feeder repeatable.  The HIP kernels
    sum in a fixed order and need no such switch."""
    torch.use_deterministic_algorithms(True, warn_only=True)
    yield
    torch.use_deterministic_algorithms(False)


@pytest.mark.parametrize("overlap", [False, True, "second stream"])
def test_replayed_iterations_equal_eager_iterations(deterministic_feeder, overlap):
    """overlap: weight gradients deferred inside the recorded iteration (what bench.py runs), on the launching stream
    or on a second one."""
    from segk import config
    from segk.graph import GraphedTrainStep
    dev = "cuda:0"
    config.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(5)
    images = [torch.randn(2, 3, 65, 65, generator=g).to(dev) for _ in range(6)]
    targets = [torch.randint(0, 6, (2, 65, 65), generator=g).float().to(dev) for _ in range(6)]
    for t in targets:
        t[:, :3] = 255.0

    with config.feeder_conv_as("torch"):
        eager = _small_model(11)
        replayed = copy.deepcopy(eager)
        opt, crit, sched = _trainer(eager)
        sched(opt, 0, 0, 0.0)
        want = []
        for i in range(6):
            if i >= 2:
                sched(opt, i, 0, 0.0)
            # the two warm-up iterations of the recorded trainer run on its example batch at the initial rate
            k = 0 if i < 2 else i
            opt.zero_grad(set_to_none=True)
            loss = crit(eager(images[k]), targets[k])
            loss.backward()
            opt.step()
            want.append(float(loss.detach()))

        opt2, crit2, sched2 = _trainer(replayed)
        sched2(opt2, 0, 0, 0.0)
        with config.overlap_wgrad_as(bool(overlap)), config.wgrad_second_stream_as(overlap == "second stream"):
            stepper = GraphedTrainStep(replayed, crit2, opt2, images[0], targets[0], warmup=2)
        got = []
        for i in range(2, 6):
            sched2(opt2, i, 0, 0.0)
            got.append(float(stepper(images[i], targets[i])))
    print(got, want[2:])
    assert np.allclose(got, want[2:], rtol=1e-5), (got, want[2:])
    # a frozen learning rate or a batch that never reached the fixed buffers would show here
    assert abs(got[-1] - want[-1]) < 0.25 * abs(want[2] - want[-1]) + 2e-3 * abs(want[-1])
    sd_e, sd_r = eager.state_dict(), replayed.state_dict()
    for k in ("decoder.last_conv.8.weight", "aspp.conv1.weight", "backbone.layer4.2.conv3.weight",
              "decoder.bn1.running_var", "backbone.layer3.5.bn2.running_mean"):
        a, b = sd_e[k].float(), sd_r[k].float()
        err = float((a - b).norm() / a.norm())
        assert err < 5e-3, (k, err)


def test_refuses_other_shapes_and_frozen_rates():
    from segk.graph import GraphedTrainStep
    dev = "cuda:0"
    lin = nn.Sequential(nn.Conv2d(3, 4, 1)).to(dev)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    x, y = torch.randn(2, 3, 8, 8, device=dev), torch.randn(2, 4, 8, 8, device=dev)
    stepper = GraphedTrainStep(lin, nn.MSELoss(), opt, x, y, warmup=1)
    l0 = float(stepper(x, y))
    l1 = float(stepper(x, y))
    assert l1 < l0
    with pytest.raises(ValueError, match="recorded for"):
        stepper(x[:1], y[:1])
    opt.param_groups[0]["lr"] = 0.05
    with pytest.raises(RuntimeError, match="learning rate changed"):
        stepper(x, y)


def test_deferred_weight_gradients_equal_inline(deterministic_feeder):
    """config.set_overlap_wgrad: the head's weight gradients are launched off the chain (on the launching stream, or on
    a second stream with config.set_wgrad_second_stream) and reach weight.grad through the end-of-backward callback.
    Same kernels, same operands: every gradient equals the inline path bit for bit, also when a second backward pass
    accumulates into existing gradients."""
    from segk import config
    from segk.utils.loss import SegmentationLosses
    dev = "cuda:0"
    config.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(9)
    image = torch.randn(2, 3, 65, 65, generator=g).to(dev)
    target = torch.randint(0, 6, (2, 65, 65), generator=g).float().to(dev)
    crit = SegmentationLosses(weight=None, cuda=True).build_loss("ce")

    def grads(overlap):
        model = _small_model(21)
        with config.feeder_conv_as("torch"), config.overlap_wgrad_as(bool(overlap)), \
                config.wgrad_second_stream_as(overlap == "second stream"):
            for _ in range(2):   # the second pass adds to the first one's gradients
                crit(model(image), target).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    inline = grads(False)
    for mode in (True, "second stream"):
        deferred = grads(mode)
        assert inline.keys() == deferred.keys()
        for n in inline:
            assert torch.equal(inline[n], deferred[n]), (mode, n)


def test_recorded_step_survives_another_optimizer(deterministic_feeder):
    """A recorded iteration holds raw device addresses of the optimizer's chunk tables and of the table that renews
    the kept weight operands.  A second model with its own segk.optim.SGD, created, stepped and destroyed between
    replays - which changes the process-wide set of kept operands - and a load_state_dict on the recorded optimizer
    must not disturb the replays: losses equal the eagerly issued twin's."""
    import gc

    from segk import config, ops
    from segk.graph import GraphedTrainStep
    dev = "cuda:0"
    config.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(5)
    images = [torch.randn(2, 3, 65, 65, generator=g).to(dev) for _ in range(5)]
    targets = [torch.randint(0, 6, (2, 65, 65), generator=g).float().to(dev) for _ in range(5)]

    with config.feeder_conv_as("torch"):
        eager = _small_model(13)
        replayed = copy.deepcopy(eager)
        opt, crit, _ = _trainer(eager)
        want = []
        for i in range(5):
            k = 0 if i < 2 else i
            opt.zero_grad(set_to_none=True)
            loss = crit(eager(images[k]), targets[k])
            loss.backward()
            opt.step()
            want.append(float(loss.detach()))

        opt2, crit2, _ = _trainer(replayed)
        stepper = GraphedTrainStep(replayed, crit2, opt2, images[0], targets[0], warmup=2)
        got = [float(stepper(images[2], targets[2]))]
        # another model + optimizer: registers weights, creates kept operands, steps eagerly, dies
        other = _small_model(17)
        opt3, crit3, _ = _trainer(other)
        for _ in range(2):
            opt3.zero_grad(set_to_none=True)
            crit3(other(images[1]), targets[1]).backward()
            opt3.step()
        del other, opt3, crit3
        gc.collect()
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(64)]   # reuse whatever was freed
        got.append(float(stepper(images[3], targets[3])))
        # the recorded optimizer reloads its own state: its tables are rebuilt for eager use, the graph keeps the old ones
        opt2.load_state_dict(opt2.state_dict())
        got.append(float(stepper(images[4], targets[4])))
        del junk
    assert np.allclose(got, want[2:], rtol=1e-5), (got, want[2:])
    assert len(ops._RECORDED_KEEP) >= 1
