#!/usr/bin/env python3
"""Headline benchmark: train images/sec of DeepLabv3+ (ResNet-101 feeder, OS16, 21 classes, 513x513,
batch 16 per GPU) with the segmentation head on the HIP kernels.

  python bench.py --gpus N --steps K --warmup W          (N > 1: starts N rank processes itself, one per device)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" is what the reference's trainer does per batch (/root/reference/tain_origin.py:124-147): poly-LR
update, zero_grad, forward, cross-entropy (ignore 255), backward, [gradient all-reduce], SGD(momentum 0.9,
wd 5e-4, 1x/10x groups).  Synthetic inputs are resident in HBM before the timed region.  Rank 0 prints
ONE JSON line.  The `roofline` object is measured live with HIP events around every launch of the
dominant kernel (the implicit-GEMM convolution) inside the timed steps; `cpu_baseline` is the oracle's
plain-PyTorch restatement of the same step timed on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "semantic-segmentation_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X bf16 MFMA, dense
FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X fp32 MFMA
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E (about 6300 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["A", "D", "E"], default="A",
                    help="A (default, the headline workload): OS16 513x513 K=21 batch 16; D: OS8 769x769 K=19 batch 4; "
                         "E: dual-attention DeepLab, 513x513 K=81 batch 32.  D and E are parity-test shapes; their "
                         "bench lines are informative only.  --batch/--size/--classes/--out-stride override.")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--classes", type=int, default=None)
    ap.add_argument("--out-stride", type=int, default=None)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--feeder-fp32", action="store_true", help="run the PyTorch feeder in fp32 even in bf16 mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32 leg of the default workload")
    ap.add_argument("--fp32-steps", type=int, default=4)
    ap.add_argument("--no-torch-baseline", action="store_true",
                    help="skip timing the oracle modules on PyTorch-ROCm (the reference path on this GPU)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-miopen-benchmark", action="store_true",
                    help="leave torch.backends.cudnn.benchmark off (MIOpen then picks the feeder's conv algorithms "
                         "heuristically instead of timing them in the first step)")
    ap.add_argument("--feeder-nchw", action="store_true", help="keep the feeder in NCHW memory format")
    ap.add_argument("--fused-loss", action="store_true",
                    help="model.forward_loss: final upsample fused into the cross entropy (opt-in API; the default "
                         "step materialises the full-size logits like the reference's trainer)")
    ap.add_argument("--optimizer", choices=["hip", "torch"], default="hip",
                    help="hip: segk.optim.SGD (one launch per parameter group; the bf16 feeder's convolution weights "
                         "are held in bf16 with fp32 masters in the optimizer); torch: torch.optim.SGD on fp32 weights")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="record the training iteration once as a HIP graph and replay it (auto: with one rank and the "
                         "HIP optimizer; ranks that exchange statistics and gradients issue their steps eagerly)")
    ap.add_argument("--overlap-wgrad", choices=["auto", "on", "off"], default="auto",
                    help="segk.config.set_overlap_wgrad: weight gradients taken off the backward chain and launched in "
                         "groups (auto: with --graph); SEGK_WGRAD_SECOND_STREAM=1 puts them on a second HIP stream")
    ap.add_argument("--feeder-conv", choices=["auto", "hip", "torch"], default=None,
                    help="segk.config.set_feeder_conv: which kernels run the feeder's stride-1 3x3 convolutions of layers 2-4 "
                         "(default: the package default)")
    ap.add_argument("--sync-bn", choices=["auto", "on", "off"], default="on",
                    help="SynchronizedBatchNorm2d (HIP) in every layer, the reference's --sync-bn (train.py:321-322); "
                         "auto = only with more than one rank, its default when the flag is not given "
                         "(train.py:381-385); off = torch's nn.BatchNorm2d (MIOpen) in the feeder")
    args = ap.parse_args()
    preset = {"A": (16, 513, 21, 16), "D": (4, 769, 19, 8), "E": (32, 513, 81, 16)}[args.config]
    for name, v in zip(("batch", "size", "classes", "out_stride"), preset):
        if getattr(args, name) is None:
            setattr(args, name, v)
    return args


_T0 = time.perf_counter()


def progress(msg):
    """Phase marker on stderr (the JSON line on stdout stays the only stdout output)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def heartbeat(period=60.0):
    """A line on stderr every minute from a daemon thread, so that a long MIOpen algorithm search in the first step
    does not look like a hang to whoever watches the run."""
    import threading

    def beat():
        while True:
            time.sleep(period)
            progress("... still running")
    threading.Thread(target=beat, daemon=True).start()


def synthetic_batch(batch, size, classes, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    image = torch.randn((batch, 3, size, size), generator=g)
    target = torch.randint(0, classes, (batch, size, size), generator=g).float()
    target[torch.rand((batch, size, size), generator=g) < 0.02] = 255.0
    for sl in (np.s_[:, :10, :], np.s_[:, -10:, :], np.s_[:, :, :10], np.s_[:, :, -10:]):
        target[sl] = 255.0
    return image.to(device), target.to(device)


def cpu_baseline(args):
    """The oracle's restatement of the same training step on the host cores: exactly the work the GPU `value`
    times (scheduler, zero_grad, forward, loss, backward, SGD); the Evaluator is timed on its own beside it."""
    import torch.nn as nn
    from oracle import head_ref, ops_ref
    torch.manual_seed(1)
    threads = torch.get_num_threads()
    model = head_ref.RefDeepLab("resnet", args.out_stride, args.classes, nn.BatchNorm2d)
    model.train()
    g1, g10 = model.lr_groups()
    opt = torch.optim.SGD([{"params": g1, "lr": 0.007}, {"params": g10, "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
    image, target = synthetic_batch(args.cpu_batch, args.size, args.classes, "cpu", 1)
    ev = ops_ref.RefEvaluator(args.classes)
    keep = {}

    def step(i):
        lr = ops_ref.poly_lr(0.007, i, 50000)
        opt.param_groups[0]["lr"], opt.param_groups[1]["lr"] = lr, 10 * lr
        opt.zero_grad()
        out = model(image)
        loss = ops_ref.ce_loss(out, target)
        loss.backward()
        opt.step()
        keep["out"] = out.detach()

    progress("CPU baseline: warm-up step")
    step(0)
    times = []
    for i in range(2):
        t0 = time.perf_counter()
        step(1 + i)
        times.append(time.perf_counter() - t0)
        progress("CPU baseline: timed step %d: %.2f s" % (i, times[-1]))
    sec = float(np.median(times))
    t0 = time.perf_counter()
    ev.add_batch(target.numpy(), ops_ref.predict(keep["out"].numpy()))   # train.py:215-219 on the host
    ev_sec = time.perf_counter() - t0
    return {"value": args.cpu_batch / sec, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "oracle (plain PyTorch fp32) DeepLabv3+ R101 OS%d K=%d batch %d %dx%d: 1 warm-up + 2 timed "
                      "train steps (the work `value` times), median %.2f s/step"
                      % (args.out_stride, args.classes, args.cpu_batch, args.size, args.size, sec),
            "evaluator_images_per_sec": args.cpu_batch / ev_sec,
            "evaluator_sample": "host argmax + confusion matrix of one batch of %d: %.3f s" % (args.cpu_batch, ev_sec)}


def torch_rocm_baseline(args, dev):
    """The oracle modules (the reference's graph in plain PyTorch) on this GPU through PyTorch-ROCm / MIOpen: the
    'reference path on ROCm' the HIP path replaces.  fp32 as the reference runs it, and bf16 autocast with
    channels_last tensors, the fastest stock configuration."""
    import torch.nn as nn
    from oracle import head_ref, ops_ref
    out = {}
    image, target = synthetic_batch(args.batch, args.size, args.classes, dev, 1)
    for tag, amp in (("fp32", False), ("bf16_autocast_channels_last", True)):
        torch.manual_seed(1)
        model = head_ref.RefDeepLab("resnet", args.out_stride, args.classes, nn.BatchNorm2d).to(dev)
        if amp:
            model = model.to(memory_format=torch.channels_last)
        model.train()
        g1, g10 = model.lr_groups()
        opt = torch.optim.SGD([{"params": g1, "lr": 0.007}, {"params": g10, "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
        img = image.contiguous(memory_format=torch.channels_last) if amp else image

        def step(i):
            lr = ops_ref.poly_lr(0.007, i, 50000)
            opt.param_groups[0]["lr"], opt.param_groups[1]["lr"] = lr, 10 * lr
            opt.zero_grad(set_to_none=True)
            with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=amp):
                logits = model(img)
            loss = ops_ref.ce_loss(logits.float(), target)
            loss.backward()
            opt.step()

        progress("PyTorch-ROCm baseline %s: warm-up" % tag)
        for i in range(3):
            step(i)
            torch.cuda.synchronize()
            progress("PyTorch-ROCm baseline %s: warm-up step %d done" % (tag, i))
        t0 = time.perf_counter()
        n = 5
        for i in range(n):
            step(3 + i)
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / n
        out[tag] = {"value": args.batch / sec, "unit": "images/sec", "ms_per_step": 1000.0 * sec}
        del model, opt
        torch.cuda.empty_cache()
    out["sample"] = ("oracle.head_ref.RefDeepLab + torch.optim.SGD on PyTorch-ROCm (MIOpen, algorithm search %s), batch %d, "
                     "3 warm-up + 5 timed steps" % ("on" if torch.backends.cudnn.benchmark else "off", args.batch))
    return out


def calibrated_hbm_peak(dev, gib=1.0, reps=5):
    """What a plain device-to-device copy reaches on this GPU right now: GB/s of bytes read + bytes written by
    `dst.copy_(src)` over `gib` GiB (median of `reps`, after one warm-up), timed with HIP events.  The memory-bound
    kernels are reported against the nominal 8 TB/s AND against this figure."""
    n = int(gib * (1 << 30))
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.fill_(1)
    dst.copy_(src)
    times = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        dst.copy_(src)
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    del src, dst
    torch.cuda.empty_cache()
    return 2.0 * n / (float(np.median(times)) * 1e-3) / 1e9


def _file_sha1(path):
    import hashlib
    with open(path, "rb") as f:
        return hashlib.sha1(f.read()).hexdigest()


def measured_traffic(args, families, calls_per_step):
    """HBM bytes per library call of the kernel `families` from the committed rocprofv3 counter passes
    (tools/pmc_traffic.sh: one FETCH_SIZE pass and one WRITE_SIZE pass over this same bench command, FETCH_SIZE
    doubled as gfx950 requires).  A call may be more than one launch (main + tail), so the launches' bytes are
    summed and divided by the calls of the profiled steps.  Counters cannot be collected from inside a timed run,
    so the figure is read from profiles/; it applies to the default workload only."""
    default = (args.batch, args.size, args.classes, args.out_stride, args.dtype) == (16, 513, 21, 16, "bf16")
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not default:
        return None, "counter passes exist for the default workload only"
    if not os.path.exists(path) or calls_per_step <= 0:
        return None, "no counter passes committed"
    try:
        with open(path) as f:
            doc = json.load(f)
        # the passes are only as fresh as the kernels they profiled: the file records the hash of the convolution
        # source it was collected with (tools/pmc_traffic.sh), and a different source today voids the figure
        now = _file_sha1(os.path.join(ROOT, "semantic-segmentation_amd", "csrc", "conv.hip"))
        if doc.get("conv_hip_sha1") != now:
            return None, ("stale: profiles/pmc_traffic.json was collected with conv.hip %s, this tree has %s; "
                          "run tools/pmc_traffic.sh again" % (str(doc.get("conv_hip_sha1"))[:12], now[:12]))
        total = sum(float(doc["kernels"][k]["launches"]) * float(doc["kernels"][k]["hbm_bytes_per_launch"])
                    for k in families if k in doc["kernels"])
        if total <= 0:
            return None, "the counter passes hold no launch of these kernels"
        return total / (float(doc.get("bench_steps", 3)) * calls_per_step), None
    except (KeyError, ValueError, OSError) as e:
        return None, "unreadable counter passes: %s" % e


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this same script (one per device,
    RCCL through torch.distributed), before this process has made any GPU call, and return their worst exit code.
    Rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    have = torch.cuda.device_count()   # counts devices without initialising the GPU runtime
    if have < args.gpus and os.environ.get("SEGK_DIST_BACKEND") != "gloo":   # gloo: rehearsal, ranks share devices
        print("bench.py: --gpus %d but only %d device(s) visible" % (args.gpus, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return max(abs(c) for c in codes)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    from segk import config, ops, parallel
    from segk.modeling.deeplab import DeepLab
    from segk.modeling.deeplab_damm import DeepLab as DeepLabDAMM
    from segk.utils.loss import SegmentationLosses
    from segk.utils.lr_scheduler import LR_Scheduler

    rank, world = parallel.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    backend = dist.get_backend() if world > 1 else None
    if backend == "nccl" and world > torch.cuda.device_count():
        if rank == 0:
            print("bench.py: %d ranks but %d device(s) visible (RCCL needs one device per rank)"
                  % (world, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.feeder_conv is not None:
        config.set_feeder_conv(args.feeder_conv)
    search = not args.no_miopen_benchmark
    if rank == 0:
        heartbeat()
    sync_bn = (world > 1) if args.sync_bn == "auto" else (args.sync_bn == "on")

    def measure(dtype, steps, warmup, detailed):
        """Build the model in `dtype`, run warmup + steps training steps, return the timings.  detailed: HIP events
        around the convolution calls inside the timed steps (about 1 % of the step), one extra untimed step with
        events around the memory-bound launches as well, and an untimed evaluation leg."""
        cd = torch.bfloat16 if dtype == "bf16" else torch.float32
        config.set_compute_dtype(cd)
        # MIOpen's timed algorithm search (about 100 s of the first bf16 step) only for the headline leg: the fp32
        # search runs for minutes
        torch.backends.cudnn.benchmark = search and dtype == args.dtype
        feeder_amp = dtype == "bf16" and not args.feeder_fp32
        torch.manual_seed(1)  # the reference's --seed default; identical initial weights on every rank
        ctor = DeepLabDAMM if args.config == "E" else DeepLab
        model = ctor(backbone="resnet", output_stride=args.out_stride, num_classes=args.classes, sync_bn=sync_bn).to(dev)
        if args.config == "E":
            # the reference initialises both attention gammas to 0 (identity branch); give them the value a trained
            # model would have so the attention products are not multiplied by zero
            for m in (model.damm.damm1, model.damm.damm2):
                m.gamma.data.fill_(0.5)
        model.train()
        if feeder_amp:
            fmt = torch.contiguous_format if args.feeder_nchw else torch.channels_last
            model.backbone = model.backbone.to(memory_format=fmt)
            feeder_fwd = model.backbone.forward

            def amp_forward(x):
                with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                    return feeder_fwd(x.contiguous(memory_format=fmt))
            model.backbone.forward = amp_forward
        groups = [{"params": list(model.get_1x_lr_params()), "lr": 0.007},
                  {"params": list(model.get_10x_lr_params()), "lr": 0.07}]
        if args.optimizer == "hip":
            from segk.optim import SGD
            low = [m.weight for m in model.backbone.modules() if isinstance(m, torch.nn.Conv2d)] if feeder_amp else []
            opt = SGD(groups, momentum=0.9, weight_decay=5e-4, nesterov=False, low_precision=low)
        else:
            opt = torch.optim.SGD(groups, momentum=0.9, weight_decay=5e-4, nesterov=False)
        criterion = SegmentationLosses(weight=None, cuda=True).build_loss("ce")
        scheduler = LR_Scheduler("poly", 0.007, 50, 1000, quiet=True)
        image, target = synthetic_batch(args.batch, args.size, args.classes, dev, 1 + rank)

        # ranks > 1: recorded when the batch-norm statistics and loss sums travel through the mailboxes (launches of
        # this library); the gradient buckets then either go through them too (SEGK_GRAD_EXCHANGE=mailbox, a rehearsal
        # path: one graph) or through RCCL between two graphs (segk.graph).  Without mailboxes the step is issued
        # eagerly and the buckets' all-reduces start from parameter hooks, under the backward pass.
        use_graph = args.optimizer == "hip" and args.graph != "off" and (
            world == 1 or parallel.step_is_capturable() or parallel.mailbox_active())
        # recorded iterations exchange the buckets between two graphs, not under the backward pass: one large
        # all-reduce then beats eight of 32 MB (2 (N - 1) ring steps of latency each)
        buckets = parallel.GradientBuckets([p for g in opt.param_groups for p in g["params"]],
                                           bucket_mb=512.0 if use_graph else 32.0,
                                           overlap=world > 1 and args.optimizer == "hip" and not use_graph)
        config.set_overlap_wgrad(args.overlap_wgrad == "on" or (args.overlap_wgrad == "auto" and use_graph))

        def step(i):
            scheduler(opt, i, 0, 0.0)
            return body()

        def body():
            opt.zero_grad(set_to_none=True)
            if args.fused_loss:
                loss = model.forward_loss(image, target)
            else:
                out = model(image)
                loss = criterion(out, target)
            loss.backward()
            if world > 1 and args.optimizer == "hip":
                opt.step(flat_grads=buckets.reduce_flat())   # the optimizer reads the all-reduced buckets in place
            else:
                buckets.reduce()
                opt.step()
            return loss

        progress("%s: model built, %d warm-up steps%s" % (dtype, warmup, ", then recorded as one HIP graph" if use_graph else ""))
        run = step
        if use_graph:
            # one process, no collectives: the iteration is recorded once and replayed (segk.graph); the schedule
            # stays on the host and reaches the recorded optimizer launch through device memory
            from segk.graph import GraphedTrainStep
            stepper = GraphedTrainStep(model, criterion, opt, image, target, warmup=warmup,
                                       loss_fn=model.forward_loss if args.fused_loss else None,
                                       buckets=buckets if world > 1 else None)
            progress("%s: warm-up done (MIOpen algorithm search included), step recorded" % dtype)

            def run(i):
                scheduler(opt, i, 0, 0.0)
                return stepper(image, target)
        else:
            for i in range(warmup):
                step(i)
                if i == 0:
                    torch.cuda.synchronize()
                    progress("%s: first step done (MIOpen algorithm search included)" % dtype)
        timer = None
        if detailed and not use_graph:
            timer = ops.KernelTimer()
            ops.TIMER = timer
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = run(warmup + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        progress("%s: %d timed steps, %.2f ms/step" % (dtype, steps, 1000.0 * elapsed / steps))
        if world > 1:
            parallel.check_exchange()   # raises if an in-kernel wait of the mailbox path timed out
        res = {"final_loss": float(loss.detach()), "feeder_amp": feeder_amp, "conv": None, "hbm": None,
               "graph": use_graph, "graph_split": bool(use_graph and stepper.split)}
        if detailed and use_graph:
            # the launches of a replayed graph cannot be bracketed one by one: the same number of steps again,
            # issued eagerly with HIP events around every convolution call, right behind the timed region
            timer = ops.KernelTimer()
            ops.TIMER = timer
            config.set_overlap_wgrad(False)   # one launch at a time on the chip: durations of overlapped launches mean nothing
            for i in range(steps):
                step(warmup + steps + i)
            torch.cuda.synchronize()
            progress("%s: %d eager steps with per-launch events" % (dtype, steps))
        if timer is not None:
            torch.cuda.synchronize()
            res["conv"] = timer.summary()   # the convolution figures
            # one extra, untimed step with events around the ~450 memory-bound launches as well (their events
            # would cost the timed region about 1.5 %), then an untimed evaluation leg (train.py:215-219 on the
            # device): class argmax + confusion matrix of the step's logits, and the fused upsample + argmax form
            timer.records = {}
            timer.hbm = True
            step(warmup + steps)
            torch.cuda.synchronize()
            res["hbm"] = timer.summary()
            timer.records = {}
            from segk.utils.metrics import Evaluator
            with torch.no_grad():
                x, low = model.backbone(image)
                low_logits = model.decoder(model.aspp(x), low)
                import segk.functional as SF
                logits = SF.interpolate_bilinear_ac(low_logits, image.shape[2:])
                ev = Evaluator(args.classes)
                ev.add_logits(target, logits)
                ev.add_low_logits(target, low_logits)
            torch.cuda.synchronize()
            res["eval"] = timer.summary()
        ops.TIMER = None
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        res["elapsed"] = elapsed
        del model, opt, buckets
        torch.cuda.empty_cache()
        return res

    def conv_rooflines(res, dtype, steps, elapsed):
        """roofline objects of the two MFMA kernel families from the HIP-event records of the timed steps."""
        out = {}
        peak = BF16_DENSE_PEAK_TFLOPS if dtype == "bf16" else FP32_MATRIX_PEAK_TFLOPS
        dom = res["conv"].get("conv_fwd_kernel")
        if dom:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                               "traffic": None,
                               # nominal FLOPs count every tap, also those that read the zero padding; this is the
                               # share of them whose tap lies inside the image (dilations 6 / 12 / 18 on 33 x 33 maps
                               # leave 0.77 / 0.57 / 0.40 of a 3 x 3) - the kernels skip a tap only where it misses
                               # the image for a whole tile, so they execute at least this share
                               "inbounds_flops_frac": dom.get("flops_inbounds", dom["flops"]) / dom["flops"],
                               "achieved_inbounds": ach * dom.get("flops_inbounds", dom["flops"]) / dom["flops"],
                               "kernel": "conv_fwd_ring_kernel (bf16) / conv_fwd_dma_kernel / conv_fwd_kernel (implicit-GEMM forward + data "
                               "gradient, all %d segk_conv_fwd calls of the head in the timed steps, each bracketed by two HIP events "
                               "on the launch stream; a call is one launch or a main + tail pair)" % dom["launches"],
                               "avg_launch_ms": dom["avg_ms"], "share_of_step": dom["ms"] / (1000.0 * elapsed)}
        wg = res["conv"].get("conv_wgrad_kernel")
        if wg:
            ach = wg["flops"] / (wg["ms"] * 1e-3) / 1e12
            out["roofline_wgrad"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                                     "inbounds_flops_frac": wg.get("flops_inbounds", wg["flops"]) / wg["flops"],
                                     "avg_launch_ms": wg["avg_ms"],
                                     "kernel": "conv_wgrad_ring_kernel (bf16, 256 output channels) / conv_wgrad_kernel / conv_wgrad_dma_kernel + wgrad_finish_kernel (%d "
                                               "weight gradients of the head: the split-slab launch and the fixed-order "
                                               "slab sum inside one bracket)" % wg["launches"],
                                     "share_of_step": wg["ms"] / (1000.0 * elapsed)}
        # the feeder's 3x3 bottleneck convolutions, when they run on the same kernels (segk.config.feeder_conv):
        # 33 x 33 maps, 274 or 548 workgroups per launch - a population of its own, kept apart from the head's
        fd, fw = res["conv"].get("conv_fwd_kernel@feeder"), res["conv"].get("conv_wgrad_kernel@feeder")
        if fd and fw:
            a1, a2 = fd["flops"] / (fd["ms"] * 1e-3) / 1e12, fw["flops"] / (fw["ms"] * 1e-3) / 1e12
            out["roofline_feeder_convs"] = {
                "bound": "mfma", "peak": peak, "unit": "TFLOP/s",
                "fwd_dgrad": {"achieved": a1, "frac": a1 / peak, "launches": fd["launches"], "avg_launch_ms": fd["avg_ms"],
                              "share_of_step": fd["ms"] / (1000.0 * elapsed)},
                "wgrad": {"achieved": a2, "frac": a2 / peak, "launches": fw["launches"], "avg_launch_ms": fw["avg_ms"],
                          "share_of_step": fw["ms"] / (1000.0 * elapsed)},
                "kernel": "the same kernels on the 28 stride-1 3x3 bottleneck convolutions of ResNet layers 2 to 4 "
                          "(65 x 65 and 33 x 33 maps, 128 to 512 channels): small launches, a third of their time is fixed cost"}
        return out

    main_res = measure(args.dtype, args.steps, args.warmup, not args.no_kernel_timing)
    elapsed = main_res["elapsed"]
    feeder_amp = main_res["feeder_amp"]

    if rank == 0:
        images = world * args.batch * args.steps
        line = {
            "metric": "train images/sec @%dx%d DeepLabv3+ OS%d" % (args.size, args.size, args.out_stride),
            "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "DeepLabv3+ %sResNet-101 OS%d, %d classes, %dx%d, batch %d/GPU: full train step "
                                   "(feeder convolutions on PyTorch-ROCm %s, batch norm %s, head on HIP kernels %s, "
                                   "CE ignore 255, %s)"
                                   % ("+ dual attention (DAMM) " if args.config == "E" else "", args.out_stride,
                                      args.classes, args.size, args.size, args.batch,
                                      "bf16 autocast channels_last" if feeder_amp else "fp32",
                                      "HIP SynchronizedBatchNorm2d in all layers" if sync_bn else "nn.BatchNorm2d (MIOpen)",
                                      args.dtype,
                                      "HIP SGD, fp32 master weights" if args.optimizer == "hip" else "torch SGD"),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "world_size": dist.get_world_size() if world > 1 else 1,
                       "collectives": ("RCCL (torch.distributed nccl backend)" if backend == "nccl" else backend) if world > 1 else None,
                       "small_exchanges": parallel.exchange_description(),
                       "launch": (("two HIP graphs per training iteration (segk.graph.GraphedTrainStep: forward + backward "
                                   "+ bucket packing, then the optimizer) with the buckets' all-reduce issued between their "
                                   "replays" if main_res.get("graph_split") else
                                   "one HIP graph per training iteration (segk.graph.GraphedTrainStep)") +
                                  ": recorded after the warm-up, replayed in the timed region; learning-rate schedule on the host"
                                  if main_res["graph"] else "eager: one host call per kernel launch"),
                       "feeder_3x3_convs": config.feeder_conv(),
                       "kernel_timing": ("HIP events around each convolution call in %d eagerly issued steps right after "
                                         "the timed region (a replayed graph cannot be bracketed per launch)" % args.steps
                                         if main_res["graph"] else "HIP events around each convolution call inside the "
                                         "timed region (about 1 %% of the step)") if main_res["conv"] is not None else None,
                       "final_loss": main_res["final_loss"]},
        }
        if main_res["conv"] is not None:
            line.update(conv_rooflines(main_res, args.dtype, args.steps, elapsed))
            if "roofline" in line:
                calls = main_res["conv"]["conv_fwd_kernel"]["launches"] / float(args.steps)
                traffic, why = measured_traffic(args, ("conv_fwd_ring_kernel", "conv_fwd_dma_kernel", "conv_fwd_kernel"), calls)
                line["roofline"]["traffic"] = traffic
                if traffic is None:
                    line["roofline"]["traffic_note"] = why
                line["roofline"]["traffic_unit"] = ("HBM bytes per segk_conv_fwd call (rocprofv3 FETCH_SIZE x2 + "
                                                    "WRITE_SIZE, profiles/pmc_traffic.json: counter passes with the feeder on MIOpen, "
                                                    "so that only the head's launches are counted)")
            summ = main_res["hbm"]
            step_ms = 1000.0 * elapsed / args.steps
            hbm_cal = calibrated_hbm_peak(dev)
            progress("device copy of 1 GiB: %.0f GB/s (read + written)" % hbm_cal)
            bn = [v for k, v in summ.items() if k.startswith("bn_")]
            if bn:
                nbytes, ms = sum(v["flops"] for v in bn), sum(v["ms"] for v in bn)
                ach = nbytes / (ms * 1e-3) / 1e9
                line["roofline_hbm"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": ach / HBM_PEAK_GBS,
                                        "hbm_peak_calibrated": hbm_cal, "frac_of_calibrated": ach / hbm_cal,
                                        "calibration": "1 GiB device-to-device copy in this run, bytes read + written",
                                        "kernel": "bn_stats / bn_apply / bn_bwd_stats / bn_bwd_apply (pixel-major batch "
                                                  "norm, %d launches of one extra step after the timed region; algorithmic bytes = every "
                                                  "operand tensor once)" % sum(v["launches"] for v in bn),
                                        "share_of_step": ms / step_ms}
            # the other memory-bound kernels of the path, each against the same 8 TB/s: algorithmic bytes of the
            # kernel's real operand types / HIP-event time, from the extra step and the evaluation leg
            others = {}
            for name in ("upsample_fwd", "upsample_bwd", "ce_fwd", "ce_bwd", "upsample_ce_fwd", "upsample_ce_bwd",
                         "argmax", "confmat", "upsample_argmax", "pack_weight"):
                in_step = name not in ("argmax", "confmat", "upsample_argmax")
                v = (summ if in_step else main_res["eval"]).get(name)
                if v and v["ms"] > 0:
                    ach = v["flops"] / (v["ms"] * 1e-3) / 1e9
                    others[name] = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "frac_of_calibrated": ach / hbm_cal,
                                    "launches": v["launches"],
                                    "avg_launch_ms": v["avg_ms"], "bytes_per_launch": v["flops"] / v["launches"],
                                    "in_step": in_step}
            if others:
                line["roofline_hbm_kernels"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "hbm_peak_calibrated": hbm_cal,
                                                "unit": "GB/s", "kernels": others,
                                                "note": "argmax / confmat / upsample_argmax: evaluation leg after the "
                                                        "timed region (Evaluator.add_logits, add_low_logits), not part of `value`"}
        default_workload = world == 1 and args.config == "A"
        if default_workload and not args.no_fp32 and args.dtype == "bf16":
            # the fp32 half of the same configuration (exact-fp32 MFMA in the head, fp32 feeder), a short loop
            torch.backends.cudnn.benchmark = False
            r32 = measure("fp32", args.fp32_steps, 2, not args.no_kernel_timing)
            e32 = r32["elapsed"]
            line["fp32"] = {"value": args.batch * args.fp32_steps / e32, "unit": "images/sec", "steps": args.fp32_steps,
                            "warmup": 2, "ms_per_step": 1000.0 * e32 / args.fp32_steps, "final_loss": r32["final_loss"],
                            "workload": "the same step with fp32 operands everywhere (exact-fp32 MFMA in the head, "
                                        "fp32 feeder convolutions with MIOpen's default algorithm choice, no timed search)"}
            if r32["conv"] is not None:
                line["fp32"].update(conv_rooflines(r32, "fp32", args.fp32_steps, e32))
        if default_workload and not args.no_torch_baseline:
            torch.backends.cudnn.benchmark = False
            line["torch_rocm_baseline"] = torch_rocm_baseline(args, dev)
        if default_workload and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
