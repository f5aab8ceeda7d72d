"""Drop-in for the reference's top-level `Modules.py`: `from Modules import GlowTTS, MLE_Loss` (Train.py:14, Inference.py:11)
resolves to the MI355X implementation."""
from glow_tts_amd.modules import GlowTTS, MLE_Loss  # noqa: F401
from glow_tts_amd.hparams import get_hp as _get_hp

hp = _get_hp()
