"""rnntransducer_amd — MI355X-native RNN-Transducer training hot path (hand-written gfx950 HIP kernels behind
a C ABI, see include/rnnt_hip.h), exposed through the reference's own module surface."""
from .data import AudioDataLoader, collate_batch
from .frontend import LogMelFrontend, spec_augment
from .loss import RNNTLoss
from .model import RNNTransducer
from .networks import AudioTransNet, JointNet, TextPredNet

__all__ = ["RNNTransducer", "JointNet", "AudioTransNet", "TextPredNet", "RNNTLoss", "LogMelFrontend", "spec_augment",
           "AudioDataLoader", "collate_batch"]
