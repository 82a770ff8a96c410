#!/bin/bash
# debug build of the engine with the attention phase timers (tools/attn_timing.py)
set -e
cd "$(dirname "$0")/../diff-mining_amd/csrc"
python ../build.py > /dev/null
OBJS=""; for f in $(ls build/*.o | grep -v "attention_pipe\|timing"); do OBJS="$OBJS $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DDM_ATTN_TIMING -c attention_pipe.hip -o build/attention_pipe_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdm_timing.so $OBJS build/attention_pipe_timing.o
