"""Worker process of the multi-process tests: `python dist_worker.py MODE RANK WORLD PORT OUTDIR`.

MODE cpu : gloo on CPU tensors - the three exchanges of segk.parallel (BN statistics vector, loss normaliser,
           gradient buckets) with the local arithmetic done by the oracle; each rank owns a shard of the batch.
MODE gpu : gloo on CUDA tensors, every rank on cuda:0 (one-GPU box) - the real HIP modules with
           SynchronizedBatchNorm2d in feeder and head, head-only model to keep it quick.
MODE nccl: the same with rank r on cuda:r over RCCL (needs WORLD devices).
MODE mbx : every rank on cuda:0 again, gloo only for the set-up: all exchanges through the peer-mapped mailboxes
           (segk_mailbox_allreduce), and SGD iterations replayed from one HIP graph per rank.
Each rank writes OUTDIR/rank{R}.npz; the parent compares with a single process that ran the whole batch.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import cases, gen, ops_ref  # noqa: E402

SEED = 4242
N_TOTAL, C, H, W, K = 6, 8, 5, 7, 5


def cpu_inputs():
    x = torch.from_numpy(gen.normal(SEED, "dw/x", (N_TOTAL, C, H, W), std=2.0, mean=0.3))
    gy = torch.from_numpy(gen.normal(SEED, "dw/gy", (N_TOTAL, C, H, W)))
    gamma = torch.from_numpy(gen.uniform(SEED, "dw/g", (C,), 0.5, 1.5))
    beta = torch.from_numpy(gen.normal(SEED, "dw/b", (C,)))
    logits = torch.from_numpy(gen.normal(SEED, "dw/l", (N_TOTAL, K, H, W), std=2.0))
    target = torch.from_numpy(gen.labels(SEED, "dw/t", (N_TOTAL, H, W), K, border=0, ignore_frac=0.3))
    target[0] = 255  # one rank sees a fully ignored image: ranks hold different numbers of valid pixels
    return x, gy, gamma, beta, logits, target


def shard(t, rank, world):
    per = t.shape[0] // world
    return t[rank * per:(rank + 1) * per]


def run_cpu(rank, world, outdir):
    from segk import parallel
    x, gy, gamma, beta, logits, target = cpu_inputs()
    xs, gs = shard(x, rank, world), shard(gy, rank, world)
    eps = 1e-5
    # ---- exchange 1: BN statistics as one [2C+1] fp64 vector ------------------------------------
    v = xs.double().reshape(xs.shape[0], C, -1)
    vec = torch.cat([v.sum((0, 2)), (v * v).sum((0, 2)), torch.tensor([float(v.shape[0] * v.shape[2])], dtype=torch.float64)])
    parallel.all_reduce_sum_(vec)
    cnt = vec[-1]
    mean = vec[:C] / cnt
    var = vec[C:2 * C] / cnt - mean * mean
    inv_std = 1.0 / torch.sqrt(var + eps)
    xhat = (xs.double() - mean[None, :, None, None]) * inv_std[None, :, None, None]
    y = xhat * gamma[None, :, None, None] + beta[None, :, None, None]
    # backward exchange: [sum g, sum g*xhat, count]
    g = gs.double()
    bvec = torch.cat([g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3)), torch.tensor([float(cnt)], dtype=torch.float64)])
    dgamma_local, dbeta_local = bvec[C:2 * C].clone(), bvec[:C].clone()
    bvec[-1] = 0.0
    parallel.all_reduce_sum_(bvec)
    dx = (gamma * inv_std)[None, :, None, None] * (g - (bvec[:C] / cnt)[None, :, None, None]
                                                  - xhat * (bvec[C:2 * C] / cnt)[None, :, None, None])
    # ---- exchange 2: loss normaliser ----------------------------------------------------------------
    ls, ts = shard(logits, rank, world).clone().requires_grad_(True), shard(target, rank, world)
    nll_sum = F.cross_entropy(ls, ts.long(), ignore_index=255, reduction="sum")
    valid = (ts != 255).sum().double()
    pack = torch.tensor([float(nll_sum), float(valid), float(ls.shape[0])], dtype=torch.float64)
    parallel.all_reduce_sum_(pack)
    # every rank reports the loss of the gathered batch; its own logits get d(global loss)/d(own logits)
    loss = pack[0] / pack[1] / pack[2]
    (nll_sum / pack[1] / pack[2]).backward()
    # ---- exchange 3: gradient buckets (SUM) on a toy parameter set --------------------------------
    torch.manual_seed(0)
    params = [nn.Parameter(torch.zeros(3, 4)), nn.Parameter(torch.zeros(5)), nn.Parameter(torch.zeros(2, 2, 2))]
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    params[1].grad = None if rank == 1 else params[1].grad
    buckets = parallel.GradientBuckets(params, bucket_mb=1e-5)  # tiny buckets: several all-reduces
    buckets.reduce()
    # weight grads dgamma/dbeta are local sums that the gradient all-reduce adds up
    wg = torch.cat([dgamma_local, dbeta_local])
    parallel.all_reduce_sum_(wg)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), y=y.numpy(), dx=dx.numpy(), dgamma=wg[:C].numpy(),
             dbeta=wg[C:].numpy(), loss=np.array([float(loss)]), dlogits=ls.grad.numpy(),
             pg0=params[0].grad.numpy(), pg1=params[1].grad.numpy(), pg2=params[2].grad.numpy(),
             nbuckets=np.array([len(buckets.buckets)]))


# ------------------------------------------------------------------------------------------------
GPU_N, GPU_K = 4, 7


def make_net(device):
    """Head + a small SyncBN'd stem standing in for the feeder (a strided convolution, then a dilated 3x3
    convolution -> batch norm -> ReLU unit on the implicit-GEMM kernels, whose statistics come out of the
    convolution's epilogue and are all-reduced like any other)."""
    from segk.modeling.aspp import build_aspp
    from segk.modeling.decoder import build_decoder
    from segk.modeling.sync_batchnorm import SynchronizedBatchNorm2d
    from segk.utils.loss import SegmentationLosses
    import segk.functional as SF

    BN = SynchronizedBatchNorm2d

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Conv2d(3, 320, 3, stride=4, padding=1, bias=False)
            self.stem_bn = BN(320)
            self.mid = nn.Conv2d(320, 320, 3, padding=2, dilation=2, bias=False)
            self.mid_bn = BN(320)
            self.low = nn.Conv2d(3, 24, 3, stride=2, padding=1, bias=False)
            self.low_bn = BN(24)
            self.aspp = build_aspp("mobilenet", 16, BN)
            self.decoder = build_decoder(GPU_K, "mobilenet", BN)

        def forward(self, img):
            x = F.relu(self.stem_bn(self.stem(img))).contiguous(memory_format=torch.channels_last)
            if not self.mid_bn.can_fuse_conv(self.mid, x):
                raise RuntimeError("the 3x3 unit of this case must run on the implicit-GEMM kernels")
            x = self.mid_bn.forward_conv_relu(self.mid, x)
            low = F.relu(self.low_bn(self.low(img)))
            y = self.decoder(self.aspp(x), low)
            return SF.interpolate_bilinear_ac(y, img.shape[2:])

    net = cases._load(Net(), SEED, "gpudist", device)
    cases._zero_dropout(net)
    return net.train()


def shard_inputs(device, images, tag="gd"):
    lo, hi = images
    img = torch.from_numpy(gen.normal(SEED, tag + "/img", (GPU_N, 3, 41, 37)) * cases._gain(GPU_N))[lo:hi].to(device)
    tgt = torch.from_numpy(gen.labels(SEED, tag + "/lab", (GPU_N, 41, 37), GPU_K, border=1, ignore_frac=0.2))
    tgt[0, :30] = 255
    return img, tgt[lo:hi].to(device)


def gpu_case(device, images, sync, overlap_buckets=None, backwards=1):
    """One SGD-free step of make_net on a shard of the batch; returns net, loss, output.
    overlap_buckets (MB): attach GradientBuckets(overlap=True) before backward and finish them (net.overlapped)."""
    from segk.utils.loss import SegmentationLosses
    net = make_net(device)
    img, tgt = shard_inputs(device, images)
    if overlap_buckets is not None:
        from segk import parallel
        net.overlapped = parallel.GradientBuckets(list(net.parameters()), bucket_mb=overlap_buckets, overlap=True)
    out = net(img)
    loss = SegmentationLosses(cuda=True).build_loss("ce")(out, tgt)
    for i in range(backwards):   # gradient accumulation: the parts add up to the gradient of the loss
        (loss / backwards).backward(retain_graph=i + 1 < backwards)
    if overlap_buckets is not None:
        net.overlapped.reduce_flat()
    return net, loss, out


def run_gpu(rank, world, outdir, device="cuda:0"):
    from segk import parallel
    per = GPU_N // world
    net, loss, out = gpu_case(device, (rank * per, (rank + 1) * per), True)
    # a second pass with hooks that start each bucket during backward: its sums must equal the ones started after
    net2, loss2, _ = gpu_case(device, (rank * per, (rank + 1) * per), True, overlap_buckets=0.5)
    # and a third whose loss is back-propagated in two halves: buckets started during the first half are reduced again
    net3, _, _ = gpu_case(device, (rank * per, (rank + 1) * per), True, overlap_buckets=0.5, backwards=2)
    buckets = parallel.GradientBuckets(list(net.parameters()), bucket_mb=0.5)
    # the flat form first (it leaves p.grad alone), then the copy-back form: both must hold the same sums
    buckets.reduce_flat()
    flat = {n: buckets.view_of(p).detach().reshape(p.shape).cpu().numpy() for n, p in net.named_parameters()}
    parallel.GradientBuckets(list(net.parameters()), bucket_mb=0.5).reduce()
    res = {"loss": np.array([float(loss.detach())]), "out": out.detach().cpu().numpy()}
    for n, p in net.named_parameters():
        res["d/" + n] = p.grad.detach().cpu().numpy()
        res["f/" + n] = flat[n]
    for n, p in net2.named_parameters():
        res["o/" + n] = net2.overlapped.view_of(p).detach().reshape(p.shape).cpu().numpy()
    for n, p in net3.named_parameters():
        res["a/" + n] = net3.overlapped.view_of(p).detach().reshape(p.shape).cpu().numpy()
    for n, b in net.named_buffers():
        if b.is_floating_point():
            res["b/" + n] = b.detach().cpu().numpy()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)


def train_steps(device, images, steps, graphed):
    """`steps` SGD iterations of make_net on a shard (segk.optim.SGD reading the all-reduced flat buckets), issued
    launch by launch or - after two eager warm-up iterations - replayed from one HIP graph per rank.  Returns the
    losses and the final weights."""
    from segk import parallel
    from segk.graph import GraphedTrainStep
    from segk.optim import SGD
    from segk.utils.loss import SegmentationLosses
    net = make_net(device)
    opt = SGD(net.parameters(), lr=0.02, momentum=0.9, weight_decay=5e-4)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    buckets = parallel.GradientBuckets(list(net.parameters()), bucket_mb=0.5)
    batches = [shard_inputs(device, images, "gd%d" % (i % 2)) for i in range(steps)]
    losses = []
    if graphed:
        stepper = GraphedTrainStep(net, crit, opt, batches[0][0], batches[0][1], warmup=2, buckets=buckets)
        assert stepper.split == (not parallel.step_is_capturable())
        for img, tgt in batches[2:]:
            losses.append(float(stepper(img, tgt)))
    else:
        for i, (img, tgt) in enumerate(batches):
            k = 0 if i < 2 else i   # the recorded trainer warms up twice on its example batch
            opt.zero_grad(set_to_none=True)
            loss = crit(net(batches[k][0]), batches[k][1])
            loss.backward()
            opt.step(flat_grads=buckets.reduce_flat())
            if i >= 2:
                losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return losses, {n: p.detach().float().cpu().numpy() for n, p in net.named_parameters()}


def run_mailbox(rank, world, outdir, gradients=True):
    """Every exchange through the peer-mapped mailboxes: the case of run_gpu (one step, statistics + loss sums +
    gradient buckets), then five SGD iterations issued eagerly and the same five with the last three replayed from
    a HIP graph, with the cross-rank comparison of the statistics switched on for the eager ones.
    gradients=False: the buckets stay on torch.distributed (gloo here, RCCL in production); the replayed iteration
    is then two graphs with the buckets' all-reduce between them, and its weight gradients are deferred and grouped."""
    from segk import config, parallel
    if world > 1:
        assert parallel.connect_mailboxes(timeout_ms=20000, gradients=gradients), parallel.exchange_description()
        assert parallel.step_is_capturable() == gradients and parallel.mailbox_active()
    run_gpu(rank, world, outdir)
    res = dict(np.load(os.path.join(outdir, "rank%d.npz" % rank)))
    per = GPU_N // world
    shard_ = (rank * per, (rank + 1) * per)
    # (MIOpen's default convolution gradients of the two stand-in feeder layers differ from run to run; its
    # deterministic choices make the eager and the replayed run comparable)
    torch.use_deterministic_algorithms(True, warn_only=True)
    config.set_debug_sync_stats(True)
    eager_losses, eager_w = train_steps("cuda:0", shard_, 5, graphed=False)
    config.set_debug_sync_stats(False)
    with config.overlap_wgrad_as(not gradients):
        graph_losses, graph_w = train_steps("cuda:0", shard_, 5, graphed=True)
    torch.use_deterministic_algorithms(False)
    res["eager_losses"], res["graph_losses"] = np.array(eager_losses), np.array(graph_losses)
    for n, v in eager_w.items():
        res["we/" + n] = v
    for n, v in graph_w.items():
        res["wg/" + n] = v
    res["exchanges"] = np.array([parallel.check_exchange()])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **res)


def main():
    mode, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import faulthandler
    faulthandler.dump_traceback_later(180 if mode in ("mbx", "split") else 60, exit=True)  # a stuck collective names its call site instead of hanging the test
    if mode == "nccl":   # one rank per device over RCCL, the production transport
        torch.cuda.set_device(rank)
    if world > 1:
        dist.init_process_group("nccl" if mode == "nccl" else "gloo", init_method="tcp://127.0.0.1:%s" % port,
                                rank=rank, world_size=world)
    torch.set_num_threads(2)
    if mode == "cpu":
        run_cpu(rank, world, outdir)
    elif mode == "nccl":
        run_gpu(rank, world, outdir, device="cuda:%d" % rank)
    elif mode in ("mbx", "split"):
        torch.cuda.set_device(0)
        run_mailbox(rank, world, outdir, gradients=mode == "mbx")
    else:
        torch.cuda.set_device(0)
        run_gpu(rank, world, outdir)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
