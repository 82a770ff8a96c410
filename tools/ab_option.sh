#!/bin/bash
# Same-box alternating A/B of one engine option (environment form DM_<NAME>) on the graded bench step:
#     bash tools/ab_option.sh DM_GN_EPI 0 1 [pairs=3] [steps=8]
OPT=$1; VA=$2; VB=$3; PAIRS=${4:-3}; STEPS=${5:-8}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.3f ms/step %8.4f img/s  igemm %7.2f TF/s (%6.2f ms)  attn %6.2f TF/s  checksum %r' % (d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_total']/d['steps'], d['roofline']['attention_tflops'], d['scores_checksum']))"; }
for i in $(seq $PAIRS); do
    echo -n "$OPT=$VA: "; env $OPT=$VA python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | line
    echo -n "$OPT=$VB: "; env $OPT=$VB python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | line
done
