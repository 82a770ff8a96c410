#!/bin/bash
# Same-box A/B of two builds of libdm_engine.so on the graded bench step in ABBA order (see tools/ab_option.sh for why):
#     bash tools/ab_libs_abba.sh <libA.so> <libB.so> [quads=2] [steps=8]
A=$1; B=$2; QUADS=${3:-2}; STEPS=${4:-8}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.3f ms/step %8.4f img/s  igemm %7.2f TF/s (%6.2f ms)  attn %6.2f TF/s  checksum %r' % (d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_total']/d['steps'], d['roofline']['attention_tflops'], d['scores_checksum']))"; }
run() { echo -n "$1 $(basename $2): "; DM_ENGINE_LIB=$2 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | line; }
for i in $(seq $QUADS); do run A $A; run B $B; run B $B; run A $A; done | tee /tmp/ab_libs.$$
python - <<PY
import re
a, b = [], []
for l in open('/tmp/ab_libs.$$'):
    m = re.match(r'([AB]) \S+:\s+([\d.]+) ms/step', l)
    if m: (a if m.group(1) == 'A' else b).append(float(m.group(2)))
print('mean A %.3f ms/step (n=%d)   B %.3f ms/step (n=%d)   B - A %+.3f ms' % (sum(a) / len(a), len(a), sum(b) / len(b), len(b), sum(b) / len(b) - sum(a) / len(a)))
PY
rm -f /tmp/ab_libs.$$
