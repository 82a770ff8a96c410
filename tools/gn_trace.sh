#!/bin/bash
# per-launch durations of the GroupNorm kernels on the U-Net's shapes: rocprofv3 kernel trace of tools/bench_ops.py norm
set -u
OUT=$PWD/gpurun_out/gn_trace; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d $OUT/t -o t -- python tools/bench_ops.py norm > $OUT/bench.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/gn_trace/t/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gn_' not in n: continue
    key = (n.split('(')[0][-24:], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'])
    agg.setdefault(key, []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0]:24s} grid {k[1]:>8s} x {k[2]:>4s} wg {k[3]:>4s}  n={len(v):3d}  median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}")
PY
