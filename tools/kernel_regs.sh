#!/bin/bash
# VGPR / spill / scratch / LDS of every kernel of the given translation units (default: all):  bash tools/kernel_regs.sh [file.hip ...]
cd "$(dirname "$0")/../diff-mining_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wno-unused-result"
for f in ${@:-*.hip}; do
    /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o - "$f" 2>/dev/null | \
        awk -v f="$f" '/^ +\.name:/ {n=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.agpr_count:/ {a=$2} /\.symbol:/ {printf "%-22s vgpr %3s agpr %3s spill %3s scratch %4s  %s\n", f, v, a, s, p, n}'
done
