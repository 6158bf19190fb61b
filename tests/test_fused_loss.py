"""Final upsample fused with the cross entropy (DeepLab.forward_loss) against the unfused HIP path and the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ops_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(N, K, hi, wi, ho, wo, seed, ignore_frac=0.1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, K, hi, wi), generator=g) * 3.0
    t = torch.randint(0, K, (N, ho, wo), generator=g).float()
    t[torch.rand((N, ho, wo), generator=g) < ignore_frac] = 255.0
    t[:, :2, :] = 255.0
    w = torch.rand((K,), generator=g) + 0.5
    return x, t, w


CASES = [(2, 21, 9, 9, 33, 33), (3, 8, 5, 7, 17, 26), (2, 81, 10, 10, 37, 37), (1, 19, 13, 13, 25, 25),
         (2, 4, 1, 1, 6, 5), (2, 21, 33, 33, 33, 33)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("weighted", [False, True])
def test_fused_matches_oracle_and_unfused(case, weighted):
    from segk import functional as SF
    N, K, hi, wi, ho, wo = case
    x, t, w = _case(*case, seed=hash(case) % 1000)
    w = w if weighted else None
    # oracle: the reference's two steps on the CPU
    xr = x.clone().requires_grad_(True)
    up = F.interpolate(xr, size=(ho, wo), mode="bilinear", align_corners=True)
    lr = ops_ref.ce_loss(up, t, weight=w)
    lr.backward()
    # fused HIP
    xf = x.to(DEV).requires_grad_(True)
    lf = SF.upsample_cross_entropy(xf, t.to(DEV), None if w is None else w.to(DEV))
    lf.backward()
    # unfused HIP
    xu = x.to(DEV).requires_grad_(True)
    lu = SF.cross_entropy(SF.interpolate_bilinear_ac(xu, (ho, wo)), t.to(DEV), None if w is None else w.to(DEV))
    lu.backward()
    assert abs(float(lf.detach()) - float(lr.detach())) <= 1e-4 * abs(float(lr.detach()))       # fp32 forward bar
    assert abs(float(lf.detach()) - float(lu.detach())) <= 1e-5 * abs(float(lu.detach()))
    gr = xr.grad
    scale = float(gr.abs().max())
    assert float((xf.grad.cpu() - gr).abs().max()) <= 1e-3 * scale    # fp32 gradient bar
    assert float((xf.grad - xu.grad).abs().max()) <= 1e-5 * scale


def test_fused_all_ignored_is_nan_with_zero_gradient():
    from segk import functional as SF
    x = torch.randn((2, 5, 4, 4), device=DEV, requires_grad=True)
    t = torch.full((2, 13, 13), 255.0, device=DEV)
    loss = SF.upsample_cross_entropy(x, t)
    loss.backward()
    assert torch.isnan(loss)
    assert torch.isfinite(x.grad).all() or torch.isnan(x.grad).all()


def test_forward_loss_equals_criterion_of_forward():
    from segk import config
    from segk.modeling.deeplab import DeepLab
    from segk.utils.loss import SegmentationLosses
    config.set_compute_dtype(torch.float32)
    torch.manual_seed(3)
    model = DeepLab("resnet", 16, 7, sync_bn=True).to(DEV).train()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    g = torch.Generator().manual_seed(5)
    image = torch.randn((2, 3, 65, 65), generator=g).to(DEV)
    target = torch.randint(0, 7, (2, 65, 65), generator=g).float().to(DEV)
    target[:, :3] = 255.0
    grads = []
    losses = []
    for fused in (False, True):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(11)   # same dropout masks
        loss = model.forward_loss(image, target) if fused else crit(model(image), target)
        loss.backward()
        losses.append(float(loss.detach()))
        grads.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    # The feeder's MIOpen convolutions are not reproducible run to run (two unfused evaluations differ by 1.5e-4 in
    # the decoder output and by 2e-2 in the ASPP weight gradients at this tiny size), so this test only shows that
    # the API computes the same quantity; exact agreement is test_fused_matches_oracle_and_unfused's job.
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[0])
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        if n.startswith("backbone."):
            continue
        a, b = grads[0][n].double(), grads[1][n].double()
        assert float((a - b).norm()) <= 1e-1 * float(a.norm()) + 1e-9, n


@pytest.mark.parametrize("case", CASES + [(1, 21, 17, 23, 131, 203), (2, 21, 129, 129, 513, 513), (1, 5, 4, 3, 9, 7)])
def test_upsample_argmax_equals_argmax_of_upsample(case):
    """Bit-exact: the fused kernel blends with the same rounded operations as the kernel that writes the logits."""
    from segk import functional as SF, ops
    from segk.utils.metrics import Evaluator
    N, K, hi, wi, ho, wo = case
    x, t, _ = _case(*case, seed=7)
    x[0, :, 0, 0] = 0.5          # exact ties: the first class must win
    if K > 2:
        x[-1, 2, -1, -1] = float("nan")
    xd = x.to(DEV)
    full = SF.interpolate_bilinear_ac(xd, (ho, wo))
    ref = ops.argmax(full)
    got = SF.upsample_argmax(xd, (ho, wo))
    assert torch.equal(ref, got)
    assert np.array_equal(got.cpu().numpy(), ops_ref.predict(full.cpu().numpy()))
    e1, e2 = Evaluator(K), Evaluator(K)
    e1.add_logits(t.to(DEV), full)
    e2.add_low_logits(t.to(DEV), xd)
    assert np.array_equal(e1.confusion_matrix, e2.confusion_matrix)
    assert np.array_equal(e1.confusion_matrix, ops_ref.confusion_matrix(t.numpy(), ref.cpu().numpy(), K))


def test_class_balanced_weights_match_oracle():
    from segk.utils.calculate_weights import calculate_weigths_labels
    g = torch.Generator().manual_seed(9)
    K = 21
    batches = []
    for _ in range(3):
        y = torch.randint(0, K, (2, 40, 33), generator=g).float()
        y[torch.rand(y.shape, generator=g) < 0.2] = 255.0
        y[0, 0, :5] = -1.0
        batches.append({"label": y})
    got = calculate_weigths_labels("synthetic", batches, K)
    ref = ops_ref.class_weights([b["label"].numpy() for b in batches], K)
    assert got.dtype == np.float64 and np.array_equal(got, ref)
