#!/bin/bash
# per-kernel time with an engine option at 0 / 1 (rocprofv3 kernel trace of 6 bench steps each): every kernel whose total differs by > 0.1 ms
#     bash tools/option_kernel_diff.sh DM_CONV_OUT_ROWS
OPT=${1:-DM_GN_EPI}; ORDER=${2:-0 1}      # second argument: the order the two arms run in ("1 0": the thermal drift of the box falls on the other arm)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in $ORDER; do
    rm -rf /tmp/gnp$v
    env $OPT=$v rocprofv3 --kernel-trace --stats -f csv -d /tmp/gnp$v -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-side --no-parity > /dev/null 2>&1
done
python - <<PY
import csv,glob
def load(v):
    f=glob.glob('/tmp/gnp%d/**/*kernel_stats.csv'%v,recursive=True)[0]
    return {r['Name']:(int(r['Calls']),float(r['TotalDurationNs'])/1e6) for r in csv.DictReader(open(f))}
a,b=load(0),load(1)
print("kernel | calls 0 -> 1 | total ms 0 -> 1 | delta   (8 steps incl. warm-up and set-up)")
tot=0
for n in sorted(set(a)|set(b), key=lambda n:-abs(b.get(n,(0,0))[1]-a.get(n,(0,0))[1])):
    ca,ta=a.get(n,(0,0)); cb,tb=b.get(n,(0,0))
    tot+=tb-ta
    if abs(tb-ta)>0.1: print(f"{n[:110]:110s} {ca:5d} -> {cb:5d}  {ta:9.3f} -> {tb:9.3f}  {tb-ta:+8.3f}")
print("sum of all deltas", round(tot,3), "ms; totals", round(sum(t for _,t in a.values()),2), round(sum(t for _,t in b.values()),2))
PY
