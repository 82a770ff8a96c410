#!/bin/bash
# SQ counters of the 3x3-convolution launches at the bench batch (r06): where do the wave cycles of igemm_pers_tr_kernel go?
#     bash tools/conv_pmc.sh <out dir>        -> <out dir>/conv_sq.txt
OUT=${1:-gpurun_out/conv_pmc}; mkdir -p $OUT; export TMPDIR=/tmp
export DM_PMC_FILTER="conv 3x3"
i=0; DIRS=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT"; do
    rocprofv3 --pmc $set --kernel-trace -f csv -d $OUT/pmc_$i -o pmc -- python tools/pmc_shapes.py run > $OUT/run_$i.txt 2> $OUT/err_$i.txt
    DIRS="$DIRS $OUT/pmc_$i"; i=$((i+1))
done
python tools/pmc_shapes.py parse_sq $DIRS > $OUT/conv_sq.txt 2> $OUT/parse.err
rm -rf $OUT/pmc_0 $OUT/pmc_1
cat $OUT/conv_sq.txt
