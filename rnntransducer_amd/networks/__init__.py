from .encoder import AudioTransNet
from .decoder import TextPredNet
from .transducer import JointNet

__all__ = ["AudioTransNet", "TextPredNet", "JointNet"]
