import sys, copy, os
sys.path.insert(0, "/root/repo/semantic-segmentation_amd"); sys.path.insert(0, "/root/repo")
import torch, numpy as np
sys.path.insert(0, "/root/repo/tests")
import test_graph as T
from segk import config
from segk.graph import GraphedTrainStep
from segk.utils.lr_scheduler import LR_Scheduler
dev = "cuda:0"
config.set_compute_dtype(torch.float32)
g = torch.Generator().manual_seed(5)
images = [torch.randn(2, 3, 65, 65, generator=g).to(dev) for _ in range(6)]
targets = [torch.randint(0, 6, (2, 65, 65), generator=g).float().to(dev) for _ in range(6)]
def eager_run(model, base):
    opt, crit, _ = T._trainer(model)
    sched = LR_Scheduler("poly", base, 1, 6, quiet=True)
    sched(opt, 0, 0, 0.0)
    out = []
    for i in range(6):
        if i >= 2: sched(opt, i, 0, 0.0)
        k = 0 if i < 2 else i
        opt.zero_grad(set_to_none=True)
        loss = crit(model(images[k]), targets[k]); loss.backward(); opt.step()
        out.append(float(loss.detach()))
    return out
def graph_run(model, base):
    opt, crit, _ = T._trainer(model)
    sched = LR_Scheduler("poly", base, 1, 6, quiet=True)
    sched(opt, 0, 0, 0.0)
    st = GraphedTrainStep(model, crit, opt, images[0], targets[0], warmup=2)
    out = []
    for i in range(2, 6):
        sched(opt, i, 0, 0.0)
        out.append(float(st(images[i], targets[i])))
    return out
with config.feeder_conv_as("torch"):
    for base in (0.01, 0.001):
        m = T._small_model(11)
        a = eager_run(copy.deepcopy(m), base); b = eager_run(copy.deepcopy(m), base); c = graph_run(copy.deepcopy(m), base)
        print(base, "eager A", a); print(base, "eager B", b); print(base, "graph  ", c, flush=True)
