#!/usr/bin/env python
"""Per-shape table of the igemm / attention launches of one bench run.

    DM_PROF_DUMP=/tmp/shapes.txt python bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/prof_shapes.py /tmp/shapes.txt 3

Columns: launches/step, ms/step, TFLOP/s, and for igemm the read-once/write-once HBM time at 5 TB/s
(the practical HBM rate) next to the MFMA time at 2.5 PF/s: the larger of the two is the shape's roofline."""
import sys
from collections import defaultdict

path, steps = sys.argv[1], int(sys.argv[2])
PEAK = float(sys.argv[3]) if len(sys.argv) > 3 else 2.5e15      # 157.3e12 for the fp32 net (bench.py --workload dift)
EB = 4.0 if PEAK < 1e15 else 2.0                                 # bytes per element
acc = defaultdict(lambda: [0, 0.0, 0.0])
for line in open(path):
    kind, M, N, K, mode, flops, ms = line.split()
    k = (int(kind), int(M), int(N), int(K), int(mode))
    acc[k][0] += 1
    acc[k][1] += float(ms)
    acc[k][2] += float(flops)
rows = []
for (kind, M, N, K, mode), (n, ms, fl) in acc.items():
    n_s, ms_s = n / steps, ms / steps
    tf = fl / ms / 1e9
    if kind == 0:
        epi = mode // 10
        taps = 1 if mode % 10 == 0 else 4 if mode % 10 == 5 else 9      # mode 5: Upsample2D + conv as four 2x2 convolutions (M = 4 parity classes x source rows)
        nout = N // 2 if epi else N
        byts = EB * (M * (K // taps) + N * K + M * nout)
        hbm_ms = byts / 5e12 * 1e3
        mfma_ms = 2.0 * M * N * K / PEAK * 1e3
        name = f"igemm mode{mode % 10}{' geglu' if epi else ''} M={M} N={N} K={K}"
        bound = max(hbm_ms, mfma_ms) * n_s
        rows.append((ms_s, name, n_s, tf, bound, "hbm" if hbm_ms > mfma_ms else "mfma"))
    else:
        name = f"attn {'self' if mode == 100 else 'cross'} rows={M} Tk={N} D={K}"
        mfma_ms = fl / n / PEAK * 1e3
        rows.append((ms_s, name, n_s, tf, mfma_ms * n_s, "mfma"))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{'shape':52s} {'n/step':>6s} {'ms/step':>8s} {'TF/s':>7s} {'roof ms':>8s} {'bound':>5s} {'gap ms':>7s}")
for ms_s, name, n_s, tf, bound, b in rows:
    print(f"{name:52s} {n_s:6.1f} {ms_s:8.3f} {tf:7.1f} {bound:8.3f} {b:>5s} {ms_s - bound:7.3f}")
print(f"total {tot:.2f} ms/step")
