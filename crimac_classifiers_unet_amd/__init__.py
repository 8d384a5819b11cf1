"""MI355X-native U-Net hot path of CRIMAC-classifiers-unet (drop-in for the reference's
``UNet_Baseline`` / ``SegPipe`` surface; compute = hand-written HIP kernels for gfx950)."""
from .unet import UNet_Baseline, UNet_LateMetInject  # noqa: F401
from .pipeline import SegPipe, SegPipeUNet, get_in_channels  # noqa: F401
from .train_ops import WeightedCrossEntropy, SGDMomentum, ExponentialLR  # noqa: F401

__all__ = ["UNet_Baseline", "UNet_LateMetInject", "SegPipe", "SegPipeUNet", "get_in_channels", "WeightedCrossEntropy",
           "SGDMomentum", "ExponentialLR"]
