// Experiment switches of the kernels' launch heuristics (tile shapes, waves per workgroup, LDS stages, fusion on / off).
// The PRODUCT library (make -C glow_tts_amd/csrc) compiles every one of them to its measured default: what libglowtts_hip.so
// computes, and through which kernel, never depends on the caller's environment.  Tools builds (-DGLOWTTS_TOOLS: `make tools` ->
// tools/_build/libglowtts_hip_tools.so, used by tools/ab.sh for A/B measurements; tools/build_tl.sh) read them from the environment.
#pragma once
#include <stdlib.h>

#ifdef GLOWTTS_TOOLS
#define GLOWTTS_TUNABLE(name, dflt) ([]() -> int { static const int v_ = []() -> int { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }(); return v_; }())
#else
#define GLOWTTS_TUNABLE(name, dflt) (dflt)
#endif
