#!/bin/bash
# VGPR / SGPR / scratch of the kernels whose mangled name contains $1 (default: wn_), read from the built library's code-object notes.
PAT=${1:-wn_}
D=$(mktemp -d /tmp/kmeta.XXXX)
cp /root/repo/glow_tts_amd/libglowtts_hip.so $D/lib.so
(cd $D && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so >/dev/null && for f in *gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f; done) | python3 -c "
import re,sys
notes=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', notes, re.S):
    if '$PAT' in m.group(1): print(m.group(1)[:90], 'scratch', m.group(2), 'sgpr', m.group(3), 'vgpr', m.group(4))
"
