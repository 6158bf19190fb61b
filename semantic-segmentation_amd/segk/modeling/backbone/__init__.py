"""Feeder factory (same call shape as /root/reference/newmodeling/backbone/__init__.py:3-25)."""
from . import resnet


def build_backbone(backbone, output_stride, BatchNorm, pretrained=False):
    if backbone == "resnet":
        return resnet.ResNet101(output_stride, BatchNorm, pretrained=pretrained)
    if backbone == "resnet50":
        return resnet.ResNet50(output_stride, BatchNorm, pretrained=pretrained)
    raise NotImplementedError("only the ResNet feeders are restated here; got %r" % (backbone,))
