#!/bin/sh
# Experiment build of the conv kernel alone: one instantiation (bf16 GATE k=5), optional in-kernel timeline stamps.
#   tools/build_tl.sh [extra hipcc flags]   ->  tools/_build/libconv_tl.so   (used by tools/conv_timeline.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGLOWTTS_TOOLS -DGLOWTTS_TOOLS_MIN -DGLOWTTS_TIMELINE "$@" \
    glow_tts_amd/csrc/gemm_cl.hip glow_tts_amd/csrc/common.hip -o tools/_build/libconv_tl.so
