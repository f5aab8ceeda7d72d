"""glow_tts_amd - MI355X-native (gfx950) Glow-TTS hot path behind the reference's Python interface.
See DESIGN.md for the path / boundary and include/glowtts_hip.h for the C ABI."""
__version__ = "0.1.0"
