"""segk — MI355X-native DeepLabv3+ segmentation-head training path.

Drop-in surface (same names and call shapes as the reference tree):
  segk.modeling.aspp.build_aspp, segk.modeling.decoder.build_decoder,
  segk.modeling.sync_batchnorm.SynchronizedBatchNorm2d / patch_replication_callback / DataParallelWithCallback,
  segk.modeling.deeplab.DeepLab, segk.utils.loss.SegmentationLosses, segk.utils.metrics.Evaluator,
  segk.utils.lr_scheduler.LR_Scheduler.
All arithmetic of those runs in hand-written HIP kernels behind the C ABI of libsegk.so
(include/segk.h); there is no host fallback.
"""
__version__ = "0.1.0"
