#!/bin/bash
# debug build of the engine with the phase timers of the persistent igemm tile (tools/igemm_timing.py) and of the pipelined
# attention kernel (tools/attn_timing.py): diff-mining_amd/lib/libdm_timing.so
set -e
cd "$(dirname "$0")/../diff-mining_amd/csrc"
python ../build.py > /dev/null
OBJS=""; for f in $(ls build/*.o | grep -v "attention_pipe\|build/igemm_pers.o\|timing\|_plain"); do OBJS="$OBJS $f"; done
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc $FLAGS -DDM_ATTN_TIMING -c attention_pipe.hip -o build/attention_pipe_timing.o
/opt/rocm/bin/hipcc $FLAGS -DDM_ATTN_TIMING -c attention_pipe80.hip -o build/attention_pipe80_timing.o
/opt/rocm/bin/hipcc $FLAGS -DDM_IGEMM_TIMING -c igemm_pers.hip -o build/igemm_pers_timing.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libdm_timing.so $OBJS build/attention_pipe_timing.o build/attention_pipe80_timing.o build/igemm_pers_timing.o
