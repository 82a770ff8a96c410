#!/bin/bash
# PMC passes over the head_dim-40 attention variants (tools/ab_attn_pp.py): per kernel (template arguments tell the variants apart)
# wave cycles, issue / wait split, matrix-pipe and VALU busy, LDS activity.   bash tools/attn_pmc.sh <out dir> <variants...>
OUT=$1; shift
mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_IDX_ACTIVE"; do
    DM_BENCH_ITERS=1 rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmc_$i -o pmc -- python tools/ab_attn_pp.py "$@" > $OUT/run_$i.txt 2> $OUT/err_$i.txt
    python tools/pmc_summary.py $OUT/pmc_$i > $OUT/summary_$i.txt 2>&1
    rm -rf $OUT/pmc_$i
    i=$((i+1))
done
