#!/bin/bash
# usage (GPU box): tools/ab_conv.sh "ENV1=.. ENV2=.." "ENV=.." ...   -> kernel times of tools/bench_hot.py for every environment, on the TOOLS build
REPO=${GRAFT_REPO_ROOT:-/root/repo}
make -C $REPO/glow_tts_amd/csrc tools > /tmp/tools_build.log 2>&1 || { tail -20 /tmp/tools_build.log; exit 1; }
for e in "$@"; do
  echo "[$e] $(env $e GLOWTTS_LIB_PATH=$REPO/tools/_build/libglowtts_hip_tools.so timeout 200 python $REPO/tools/bench_hot.py 2>&1 | tail -1)"
done
