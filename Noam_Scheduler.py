"""Drop-in shim with the reference's module name (Train.py:16 `from Noam_Scheduler import Modified_Noam_Scheduler`)."""
from glow_tts_amd.optim import Modified_Noam_Scheduler, Noam_Scheduler  # noqa: F401
