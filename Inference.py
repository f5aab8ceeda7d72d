"""Reference entry point preserved (Inference.py:285-313): `python Inference.py -c <checkpoint>`; the implementation is glow_tts_amd/inferencer.py."""
from glow_tts_amd.inferencer import Inferencer, main  # noqa: F401

if __name__ == "__main__":
    main()
