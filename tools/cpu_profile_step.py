"""cProfile of the host side of a few training steps (where does the launch overhead go?)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "semantic-segmentation_amd"))
import torch  # noqa: E402

from segk import config  # noqa: E402
from segk.modeling.deeplab import DeepLab  # noqa: E402
from segk.utils.loss import SegmentationLosses  # noqa: E402
import bench  # noqa: E402

sync = len(sys.argv) > 1 and sys.argv[1] == "sync"
torch.backends.cudnn.benchmark = True
config.set_compute_dtype(torch.bfloat16)
dev = torch.device("cuda", 0)
model = DeepLab("resnet", 16, 21, sync_bn=sync).to(dev).train()
model.backbone = model.backbone.to(memory_format=torch.channels_last)
fwd = model.backbone.forward


def amp(x):
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        return fwd(x.contiguous(memory_format=torch.channels_last))


model.backbone.forward = amp
opt = torch.optim.SGD(model.parameters(), lr=0.007, momentum=0.9, weight_decay=5e-4)
crit = SegmentationLosses(cuda=True).build_loss("ce")
image, target = bench.synthetic_batch(16, 513, 21, dev, 1)


def step():
    opt.zero_grad(set_to_none=True)
    loss = crit(model(image), target)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
