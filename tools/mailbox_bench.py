#!/usr/bin/env python3
"""Latency of one small all-reduce between two processes that share this GPU: the peer-mapped mailbox path
(segk_mailbox_allreduce_f64) issued launch by launch and replayed from a HIP graph, against torch.distributed's
gloo all-reduce of the same vector staged through the host (what segk.parallel falls back to on one card; RCCL needs
one device per rank).  usage: tools/mailbox_bench.py [--n 513] [--iters 400]      (starts its two ranks itself)

On one device the mailboxes live in the same HBM, so this prices the kernel side of an exchange (launch, stores,
flag hand-off, spin, sum), not the xGMI hop of a multi-GPU node."""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))


def worker(rank, world, port, n, iters):
    import torch
    import torch.distributed as dist
    from segk import parallel
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    v = torch.ones(n, dtype=torch.float64, device=dev)

    def timed(fn, reps):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    # gloo, host-staged (segk.parallel's fallback on a shared card)
    gloo_us = timed(lambda: parallel.all_reduce_sum_(v.fill_(1.0)), max(iters // 8, 20))
    assert parallel.connect_mailboxes(timeout_ms=20000), parallel.exchange_description()
    for _ in range(20):
        parallel.all_reduce_sum_(v.fill_(1.0))
    eager_us = timed(lambda: parallel.all_reduce_sum_(v), iters)
    # the same launches recorded once: what a replayed training step pays per exchange
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    per_graph = 50
    with torch.cuda.graph(graph, stream=side):
        for _ in range(per_graph):
            parallel.all_reduce_sum_(v)
    graph.replay()
    graph_us = timed(graph.replay, max(iters // per_graph, 4)) / per_graph
    done = parallel.check_exchange()
    if rank == 0:
        print("all-reduce of %d fp64 values between 2 processes on one MI355X, microseconds per exchange:" % n)
        print("  gloo, staged through the host        %9.1f" % gloo_us)
        print("  mailboxes, one launch per exchange   %9.1f" % eager_us)
        print("  mailboxes, replayed from a HIP graph %9.1f   (%d exchanges per graph)" % (graph_us, per_graph))
        print("  exchanges completed on rank 0: %d" % done)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=513)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--port", default="")
    a = ap.parse_args()
    if a.rank >= 0:
        worker(a.rank, 2, a.port, a.n, a.iters)
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank", str(r), "--port", port, "--n", str(a.n),
                               "--iters", str(a.iters)], env=env) for r in range(2)]
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait(timeout=300)))
    sys.exit(rc)


if __name__ == "__main__":
    main()
