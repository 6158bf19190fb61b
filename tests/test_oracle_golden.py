"""The oracle restatement against fixtures produced by the imported reference.

On the machine that generated the fixtures the floating-point results are bit-identical;
another CPU may pick different oneDNN kernels, so floats are compared to 2e-5 relative
and everything integer (confusion matrix, predictions) bit-exactly.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from helpers import assert_dicts_close, load_golden
from oracle import cases, damm_ref, gen, head_ref, ops_ref

TOL = 2e-5


def oracle_impl():
    def ce(logit, target, weight):
        return ops_ref.ce_loss(logit, target, weight)

    return cases.Impl(Branch=head_ref.RefBranch, ASPP=head_ref.RefASPP, Decoder=head_ref.RefDecoder, ce_loss=ce,
                      Evaluator=ops_ref.RefEvaluator,
                      DeepLab=lambda: head_ref.RefDeepLab("resnet", 16, 21, nn.BatchNorm2d),
                      lr_groups=lambda m: m.lr_groups(), name="oracle", PAM=damm_ref.RefPAM, CAM=damm_ref.RefCAM,
                      DAMM=damm_ref.RefDAMM, focal_loss=lambda lo, t, w: ops_ref.focal_loss(lo, t, w),
                      class_weights=lambda ds, loader, K: ops_ref.class_weights([b["label"].numpy() for b in loader], K),
                      DeepLabDAMM=lambda: damm_ref.RefDeepLabDAMM("resnet", 16, 8, nn.BatchNorm2d),
                      dropout_masks=head_ref.use_dropout_masks)


@pytest.fixture(scope="module")
def impl():
    torch.manual_seed(0)
    return oracle_impl()


@pytest.mark.parametrize("k,d", cases.BRANCH_CFGS)
def test_branch(impl, k, d):
    assert_dicts_close(cases.case_branch(impl, k, d), load_golden("branch_k%d_d%d" % (k, d)), TOL)


def test_aspp_mobilenet(impl):
    assert_dicts_close(cases.case_aspp(impl, "mobilenet", 16), load_golden("aspp_mobilenet_os16"), TOL)


def test_aspp_drn_os8(impl):
    got = cases.case_aspp(impl, "drn", 8, shape=(4, 13, 9), inplanes=512)
    assert_dicts_close(got, load_golden("aspp_drn_os8"), TOL)


@pytest.mark.parametrize("K", [21, 19])
def test_decoder(impl, K):
    assert_dicts_close(cases.case_decoder(impl, K), load_golden("decoder_k%d" % K), TOL)


@pytest.mark.parametrize("kind", cases.DROPOUT_KINDS)
def test_dropout_on(impl, kind):
    """Dropout switched ON: the fixture is the reference running its own nn.Dropout from a seeded generator and holds
    the masks it drew; the oracle's masked twin (x * keep / (1 - p)) on those masks reproduces outputs and gradients."""
    gold = load_golden("dropout_" + kind)
    masks = cases.unpack_masks(gold)
    assert len(masks) == (1 if kind == "aspp" else 2)
    for m in masks:   # p = 0.5 (and 0.1): the recorded draws are dropout masks, not all-ones
        assert 0.3 < float(m.float().mean()) < 0.95
    want = {k: v for k, v in gold.items() if not k.startswith("mask")}
    assert_dicts_close(cases.case_dropout(impl, kind, masks), want, TOL)


def test_dropout_masks_all_consumed(impl):
    masks = cases.unpack_masks(load_golden("dropout_aspp"))
    with pytest.raises(AssertionError):
        cases.case_dropout(impl, "aspp", masks + masks)


@pytest.mark.parametrize("i", range(len(cases.INTERP_CFGS)))
def test_interp(impl, i):
    shp, size = cases.INTERP_CFGS[i]
    assert_dicts_close(cases.case_interp(impl, shp, size), load_golden("interp_%d" % i), TOL)


@pytest.mark.parametrize("name,K", cases.CE_CFGS)
def test_ce(impl, name, K):
    got = cases.case_ce(impl, name, K)
    want = load_golden("ce_" + name)
    if name == "all_ignored":
        assert np.isnan(want["loss"]).all() and np.isnan(got["loss"]).all()
    assert_dicts_close(got, want, TOL)


@pytest.mark.parametrize("name,K", cases.FOCAL_CFGS)
def test_focal(impl, name, K):
    assert_dicts_close(cases.case_focal(impl, name, K), load_golden("focal_" + name), TOL)


@pytest.mark.parametrize("K", cases.CW_KS)
def test_class_weights(impl, K):
    got = cases.case_class_weights(impl, K)
    want = load_golden("class_weights_k%d" % K)
    assert np.array_equal(got["weights"], want["weights"])   # float64 from integer counts: exact
    assert got["weights"][2] == 1 / np.log(1.02)              # the class that never occurs


@pytest.mark.parametrize("K", cases.EVAL_KS)
def test_evaluator(impl, K):
    got = cases.case_eval(impl, K)
    want = load_golden("eval_k%d" % K)
    assert_dicts_close(got, want, 1e-12, exact=("cm", "pred"))


def test_syncbn_clamp_formula():
    want = load_golden("syncbn")
    s, q = torch.from_numpy(want["sum"]), torch.from_numpy(want["ssum"])
    mean, inv_std, unb = ops_ref.syncbn_mean_invstd_clamp(s, q, 6 * 5 * 7, 1e-5)
    rm0 = torch.from_numpy(gen.normal(cases.SEED, "syncbn/rm", (12,), std=0.1))
    rv0 = torch.from_numpy(gen.uniform(cases.SEED, "syncbn/rv", (12,), 0.5, 1.5))
    rm, rv = ops_ref.syncbn_running_update(rm0, rv0, mean, unb, 0.1)
    got = {"mean": mean.numpy(), "inv_std": inv_std.numpy(), "running_mean": rm.numpy(), "running_var": rv.numpy()}
    assert_dicts_close(got, {k: want[k] for k in got}, 1e-6)
    # the constant channel hits the clamp: inv_std == eps^-1/2
    assert abs(got["inv_std"][3] - 1e-5 ** -0.5) / 1e-5 ** -0.5 < 1e-5


def test_syncbn_replicas_equal_single_device():
    """N replicas with global statistics == one device on the concatenated batch."""
    x = torch.from_numpy(gen.normal(cases.SEED, "syncbn/eq", (8, 6, 5, 4)))
    g = torch.from_numpy(gen.uniform(cases.SEED, "syncbn/g", (6,), 0.5, 1.5))
    b = torch.from_numpy(gen.normal(cases.SEED, "syncbn/b", (6,)))
    outs, mean, inv_std, _ = ops_ref.syncbn_forward_replicated(list(x.chunk(4)), g, b, 1e-5, legacy_clamp=False)
    ref = torch.nn.functional.batch_norm(x, None, None, g, b, True, 0.1, 1e-5)
    assert torch.allclose(torch.cat(outs), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.slow
def test_e2e_two_sgd_steps(impl):
    got = cases.case_e2e(impl)
    want = load_golden("e2e")
    assert_dicts_close(got, want, 1e-4, exact=())
    assert np.array_equal(got["cm"], want["cm"])


@pytest.mark.parametrize("kind", cases.ATT_KINDS)
def test_attention(impl, kind):
    assert_dicts_close(cases.case_attention(impl, kind), load_golden("att_" + kind), TOL)


def test_damm(impl):
    assert_dicts_close(cases.case_damm(impl), load_golden("damm_ghostnet"), TOL)


def test_e2e_damm(impl):
    assert_dicts_close(cases.case_e2e_damm(impl), load_golden("e2e_damm"), 1e-4)


@pytest.mark.slow
def test_fullsize_A(impl):
    assert_dicts_close(cases.case_fullsize(impl, "A", n=3, backward=True), load_golden("fullA"), 1e-4)


@pytest.mark.slow
def test_fullsize_D(impl):
    assert_dicts_close(cases.case_fullsize(impl, "D", n=3, backward=True), load_golden("fullD"), 1e-4)
