"""Drop-in for the reference's `monotonic_align` package (monotonic_align/__init__.py:6-21): GPU-resident MAS."""
from glow_tts_amd.monotonic_align import maximum_path  # noqa: F401
