"""Reference entry point preserved (Train.py:592-598): `python Train.py -s <steps>`; the implementation is glow_tts_amd/trainer.py."""
from glow_tts_amd.trainer import Trainer, main  # noqa: F401

if __name__ == "__main__":
    main()
