#!/bin/bash
# Same-box round-robin of several builds of libdm_engine.so on the graded bench step:  bash tools/ab_libs.sh <rounds> <steps> <lib...>
R=$1; S=$2; shift 2
for i in $(seq $R); do for L in "$@"; do
  echo -n "$(basename $L): "; DM_ENGINE_LIB=$L python bench.py --steps $S --warmup 2 --no-cpu-baseline --no-side --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.3f ms/step  igemm %7.2f TF/s (%6.2f ms)  checksum %r' % (d['ms_per_step'], d['roofline']['achieved'], d['roofline']['kernel_ms_total']/d['steps'], d['scores_checksum']))"
done; done
