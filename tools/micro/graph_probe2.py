import sys, copy
sys.path.insert(0, "/root/repo/semantic-segmentation_amd"); sys.path.insert(0, "/root/repo")
import torch, torch.nn as nn
from oracle import head_ref
dev = "cuda:0"
g = torch.Generator().manual_seed(5)
images = [torch.randn(2, 3, 65, 65, generator=g).to(dev) for _ in range(6)]
targets = [torch.randint(0, 6, (2, 65, 65), generator=g).to(dev) for _ in range(6)]
torch.manual_seed(11)
m0 = head_ref.RefDeepLab("resnet", 16, 6, nn.BatchNorm2d).to(dev).train()
for m in m0.modules():
    if isinstance(m, nn.Dropout): m.p = 0.0
def run(model, lr):
    opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=0.9, weight_decay=5e-4)
    out = []
    for i in range(4):
        opt.zero_grad(set_to_none=True)
        loss = nn.functional.cross_entropy(model(images[i]), targets[i]); loss.backward(); opt.step()
        out.append(float(loss.detach()))
    return out
for lr in (0.01, 0.001):
    print(lr, "torch A", run(copy.deepcopy(m0), lr)); print(lr, "torch B", run(copy.deepcopy(m0), lr), flush=True)
for det in (True,):
    torch.use_deterministic_algorithms(True, warn_only=True)
    print("deterministic A", run(copy.deepcopy(m0), 0.001)); print("deterministic B", run(copy.deepcopy(m0), 0.001))
