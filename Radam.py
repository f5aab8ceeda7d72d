"""Drop-in shim with the reference's module name (Train.py:17 `from Radam import RAdam`)."""
from glow_tts_amd.optim import RAdam  # noqa: F401
