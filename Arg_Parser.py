"""Reference module preserved (Arg_Parser.py:3-12): `Recursive_Parse(dict) -> nested argparse.Namespace`."""
from glow_tts_amd.hparams import Recursive_Parse  # noqa: F401
