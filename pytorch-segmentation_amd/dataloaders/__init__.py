from .synth import Synth, SynthImages  # noqa: F401
from .voc import VOC  # noqa: F401
