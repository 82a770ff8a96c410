#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max / %.
Usage: python tools/rocpd_stats.py <results.db> [--by-grid]   (writes CSV-ish text to stdout)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    by_grid = "--by-grid" in sys.argv
    name_col = "name" if "name" in cols else "kernel_name"
    key = f"{name_col}, grid_x, workgroup_x" if by_grid and "grid_x" in cols else name_col
    q = (f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
         f"from kernels group by {key} order by 3 desc")
    rows = list(cur.execute(q))
    total = sum(r[-4] for r in rows)
    print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
    for r in rows:
        name = " ".join(str(x) for x in r[:-5])
        calls, tot, avg, mn, mx = r[-5:]
        print(f"\"{name[:110]}\",{calls},{tot/1e6:.3f},{avg/1e3:.1f},{mn/1e3:.1f},{mx/1e3:.1f},{100*tot/total:.2f}")
    print(f"TOTAL,,{total/1e6:.3f}")


if __name__ == "__main__":
    main()
