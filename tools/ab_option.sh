#!/bin/bash
# Same-box A/B of one engine option (environment form DM_<NAME>) on the graded bench step, in ABBA order (A B B A A B B A ...): a box
# drifts by a few 0.1 % over minutes and whichever arm runs second in a fixed order inherits the drift (seen in r05: rocprofv3 traces
# of the two arms of `gn_epi` differed by +3.2 ms or -5.5 ms over 8 steps depending on which ran first):
#     bash tools/ab_option.sh DM_GN_EPI 0 1 [quads=2] [steps=8]
OPT=$1; VA=$2; VB=$3; QUADS=${4:-2}; STEPS=${5:-8}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.3f ms/step %8.4f img/s  igemm %7.2f TF/s (%6.2f ms)  attn %6.2f TF/s  checksum %r' % (d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['kernel_ms_total']/d['steps'], d['roofline']['attention_tflops'], d['scores_checksum']))"; }
run() { echo -n "$OPT=$1: "; env $OPT=$1 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | line; }
for i in $(seq $QUADS); do run $VA; run $VB; run $VB; run $VA; done | tee /tmp/ab_option.$$
python - <<PY
import re
a, b = [], []
for l in open('/tmp/ab_option.$$'):
    m = re.match(r'$OPT=(\S+):\s+([\d.]+) ms/step', l)
    if m: (a if m.group(1) == '$VA' else b).append(float(m.group(2)))
print('mean $OPT=$VA %.3f ms/step (n=%d)   $OPT=$VB %.3f ms/step (n=%d)   difference %+.3f ms' % (sum(a) / len(a), len(a), sum(b) / len(b), len(b), sum(b) / len(b) - sum(a) / len(a)))
PY
rm -f /tmp/ab_option.$$
