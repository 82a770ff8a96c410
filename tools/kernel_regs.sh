#!/bin/bash
# VGPRs / spilled VGPRs / scratch bytes of every kernel of the given translation units (default: all):
#   bash tools/kernel_regs.sh [file.hip ...]
# (the persistent igemm kernels sit at 236-256 VGPRs: check after any change to igemm_pers_tile.h; and compare the instruction
#  stream of an untouched instantiation before / after with `hipcc -S --cuda-device-only` + diff, as r03 did)
cd "$(dirname "$0")/../diff-mining_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result"
for f in ${@:-*.hip}; do
    /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o /tmp/kernel_regs.s "$f" 2>/dev/null || { echo "$f: compile failed"; continue; }
    python3 - "$f" <<'PY'
import re, sys
t = open('/tmp/kernel_regs.s').read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)', t, re.S):
    print(f"{sys.argv[1]:24s} vgpr {m.group(3):>3s} spill {m.group(4):>3s} scratch {m.group(2):>4s}  {m.group(1)}")
PY
done
