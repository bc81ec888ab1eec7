"""waveome_amd - MI355X-native GP fitting core behind waveome's GPSearch / PSVGP surface.

Only the per-output Gaussian marginal-likelihood optimisation path of omicsEye/waveome is
implemented; everything numerical runs in libwaveome_amd.so on gfx950."""
from . import kernels  # noqa: F401
from .spec import ModelSpec  # noqa: F401

__version__ = "0.1.0"
