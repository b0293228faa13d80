from .synth import Synth  # noqa: F401
