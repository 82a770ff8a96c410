#!/bin/bash
# Same-box alternating A/B of two builds of libdm_engine.so on the graded bench step (box-to-box spread is +-2-3 %: only
# interleaved runs on one box separate changes of ~1 %):
#     bash tools/ab_lib.sh <libA.so> <libB.so> [pairs=3] [steps=8] [shape filter for the per-shape table]
# Prints ms/step, images/s, igemm TF/s per run, then the per-shape tables (DM_PROF_DUMP) of one run of each library.
A=$1; B=$2; PAIRS=${3:-3}; STEPS=${4:-8}; FILTER=${5:-geglu}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.3f ms/step %8.4f img/s  igemm %7.2f TF/s (%6.2f ms)  attn %6.2f TF/s  checksum %r' % (d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_total']/d['steps'], d['roofline']['attention_tflops'], d['scores_checksum']))"; }
for i in $(seq $PAIRS); do
    echo -n "A $(basename $A): "; DM_ENGINE_LIB=$A python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | line
    echo -n "B $(basename $B): "; DM_ENGINE_LIB=$B python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | line
done
for L in $A $B; do
    rm -f /tmp/ab_shapes.txt
    DM_ENGINE_LIB=$L DM_PROF_DUMP=/tmp/ab_shapes.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side --no-parity > /dev/null 2>&1
    echo "--- per shape, $(basename $L)"; python tools/prof_shapes.py /tmp/ab_shapes.txt 3 | grep -i -E "$FILTER|^total"
done
