"""SynchronizedBatchNorm{1,2,3}d on HIP kernels, one process per GPU.

Replaces /root/reference/modeling/sync_batchnorm/batchnorm.py:38-126: constructor signature, class
hierarchy (subclass of torch's _BatchNorm, not of nn.BatchNorm2d), parameters and buffers are the
same, so state_dicts interchange.  Statistics are global over all ranks of the default process group:
each rank reduces sum / sum-of-squares of its shard with a HIP kernel, one all-reduce of a [2C+1] fp64
vector replaces the master/slave queue hand-off (comm.py:97-125) plus ReduceAddCoalesced/Broadcast
(batchnorm.py:101-105), and every rank derives mean, inv_std and the running statistics itself.  With a
single rank, or in eval mode, the arithmetic is F.batch_norm's (batchnorm.py:50-53).
"""
from torch.nn.modules.batchnorm import _BatchNorm

from ... import functional as SF

__all__ = ["SynchronizedBatchNorm1d", "SynchronizedBatchNorm2d", "SynchronizedBatchNorm3d"]


def conv_is_hip_eligible(conv):
    """The static part of can_fuse_conv: the stride-1 bottleneck 3x3 convolutions of ResNet layers 2 to 4 (128
    channels and up; measured GPU time per unit, forward + statistics + both gradients, bf16, batch 16 at output
    stride 16, HIP against PyTorch-ROCm: 110 / 170 us at 128 channels, 91 / 181 at 256, 253 / 448 at 512) and the 1x1
    convolutions that reduce 256 or more channels (conv1 of the bottlenecks: 160 / 172 us for 256 -> 64 at 129 x 129,
    99 / 114 for 512 -> 128 at 65 x 65, 70 / 105 for 1024 -> 256, 154 / 231 for 2048 -> 512).  The 64-channel 3x3
    convolutions of layer 1 stay on the vendor kernels."""
    if not (conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None and conv.padding_mode == "zeros"
            and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0):
        return False
    if conv.kernel_size == (3, 3):
        return (conv.padding == conv.dilation and conv.dilation[0] == conv.dilation[1] and conv.in_channels >= 128
                and conv.out_channels >= 128)
    if conv.kernel_size == (1, 1):
        return conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.in_channels >= 256 and conv.out_channels >= 64
    return False


class _SynchronizedBatchNorm(_BatchNorm):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)

    def forward(self, input):
        self._check_input_dim(input)
        if input.shape[1] != self.num_features:
            raise ValueError("expected %d channels, got %d" % (self.num_features, input.shape[1]))
        return SF.SyncBNFunction.apply(input, self.weight, self.bias, self)

    def forward_relu(self, input):
        """relu(self(input)) in one pass (and one pass fewer in backward): for callers that apply a ReLU right
        behind this layer, as every feeder block does (modeling/backbone/resnet101.py Bottleneck.forward)."""
        self._check_input_dim(input)
        if input.shape[1] != self.num_features:
            raise ValueError("expected %d channels, got %d" % (self.num_features, input.shape[1]))
        return SF.SyncBNFunction.apply(input, self.weight, self.bias, self, True)

    def forward_add_relu(self, input, residual, grad_slot=None):
        """relu(self(input) + residual) in one pass: the tail of a bottleneck block.  Needs channels_last tensors of
        one dtype; callers fall back to the separate operations otherwise.  grad_slot: see new_grad_slot."""
        self._check_input_dim(input)
        if input.shape[1] != self.num_features:
            raise ValueError("expected %d channels, got %d" % (self.num_features, input.shape[1]))
        return SF.SyncBNFunction.apply(input, self.weight, self.bias, self, True, residual, grad_slot)

    @staticmethod
    def new_grad_slot():
        """For a block whose input x feeds both a forward_conv_relu unit and the shortcut of a forward_add_relu /
        forward_conv_add_relu unit: pass the same slot to both and the shortcut's gradient is added inside the first
        unit's data-gradient convolution instead of by a separate addition (segk.functional.GradSlot)."""
        return SF.GradSlot()

    @staticmethod
    def can_fuse_residual(input, residual):
        return (input.dim() == 4 and input.shape == residual.shape and input.dtype == residual.dtype
                and input.dtype in (SF.torch.float32, SF.torch.bfloat16) and not input.is_contiguous()
                and SF.pm_view_of(input) is not None and SF.pm_view_of(residual) is not None
                and input.shape[1] % 4 == 0)

    def can_fuse_conv(self, conv, input):
        """Whether forward_conv_relu(conv, input) applies: a stride-1 3x3 convolution with padding = dilation and 128
        or more channels on both sides, or a 1x1 convolution from 256 or more channels onto 64 or more (no bias,
        channel counts multiples of 32), a GPU tensor, and the mode set with segk.config.set_feeder_conv."""
        from ... import config
        mode = config.feeder_conv()
        if mode == "torch" or not (isinstance(input, SF.torch.Tensor) and input.is_cuda and input.dim() == 4):
            return False
        if not conv_is_hip_eligible(conv) or conv.out_channels != self.num_features:
            return False
        if input.dtype not in (SF.torch.float32, SF.torch.bfloat16):
            return False
        if mode == "auto":
            return SF.pm_view_of(input) is not None and not input.is_contiguous()
        return True

    @staticmethod
    def gradient_to_slot(x, slot):
        """x for a branch whose gradient is to arrive through `slot` (segk.functional.GradToSlotFunction): the
        projection shortcut of a block whose conv1 unit holds the same slot."""
        return SF.grad_to_slot(x, slot)

    @staticmethod
    def new_stat_link():
        """For two units of which the second is the only consumer of the first one's output: pass the link as
        `stats_to` to the first and as `stats_for` to the second, and the second unit's data-gradient convolution emits
        the backward statistics of the first unit's batch norm (segk.functional.StatLink)."""
        return SF.StatLink()

    def forward_conv_relu(self, conv, input, grad_slot=None, stats_for=None, stats_to=None):
        """relu(self(conv(input))) as one unit on the implicit-GEMM kernels (statistics from the convolution's
        epilogue): a feeder bottleneck's conv2 -> bn2 -> relu, or its conv1 -> bn1 -> relu."""
        return SF.ConvBNReLUFunction.apply(input, conv.weight, self.weight, self.bias, conv, self, grad_slot, stats_for,
                                           stats_to)

    def can_fuse_conv_residual(self, conv, input, residual):
        """Whether forward_conv_add_relu(conv, input, residual) applies: a 1x1 convolution from 128 or more channels
        onto 512 or more (conv3 of the bottlenecks of ResNet layers 2 to 4; measured per unit as in
        conv_is_hip_eligible: 94 against 117 us for 128 -> 512 at 65 x 65, 74 / 104 for 256 -> 1024, 159 / 252 for
        512 -> 2048; 64 -> 256 at 129 x 129 is 166 / 157 and stays on the vendor kernels), channels_last GPU tensors
        of one dtype, and the mode set with segk.config.set_feeder_conv."""
        from ... import config
        T = SF.torch
        if config.feeder_conv() == "torch" or not (isinstance(input, T.Tensor) and input.is_cuda and input.dim() == 4):
            return False
        if not (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
                and conv.dilation == (1, 1) and conv.bias is None and conv.in_channels >= 128
                and conv.in_channels % 32 == 0 and conv.out_channels >= 512 and conv.out_channels % 32 == 0
                and conv.out_channels == self.num_features):
            return False
        cd = SF._feeder_compute_dtype(input)
        return (input.dtype in (T.float32, T.bfloat16) and isinstance(residual, T.Tensor) and residual.dtype == cd
                and tuple(residual.shape) == (input.shape[0], conv.out_channels, input.shape[2], input.shape[3])
                and SF.pm_view_of(input) is not None and not input.is_contiguous()
                and SF.pm_view_of(residual) is not None and not residual.is_contiguous())

    def forward_conv_add_relu(self, conv, input, residual, grad_slot=None, stats_for=None, stats_to=None):
        """relu(self(conv(input)) + residual) as one unit on the implicit-GEMM kernels: a feeder bottleneck's
        conv3 -> bn3 -> += residual -> relu."""
        return SF.ConvBNAddReLUFunction.apply(input, conv.weight, self.weight, self.bias, residual, conv, self, grad_slot,
                                              stats_for, stats_to)

    def __data_parallel_replicate__(self, ctx, copy_id):
        raise RuntimeError("this SynchronizedBatchNorm runs one process per GPU (torch.distributed); "
                           "nn.DataParallel replication is not supported")


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 2 and input.dim() != 3:
            raise ValueError("expected 2D or 3D input (got {}D input)".format(input.dim()))


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 4:
            raise ValueError("expected 4D input (got {}D input)".format(input.dim()))


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    def _check_input_dim(self, input):
        if input.dim() != 5:
            raise ValueError("expected 5D input (got {}D input)".format(input.dim()))
