#!/usr/bin/env python
"""Fold rocprofv3 --pmc passes (counter_collection.csv under <dir>/pmc_*/) into one JSON summary.

Usage: python tools/pmc_to_json.py <dir> <steps_in_pass>  > profiles/<tag>_pmc.json
Kernels are grouped by family; FETCH_SIZE/WRITE_SIZE are KiB, FETCH_SIZE is doubled (gfx950 correction,
MI355X_MICROARCH.md HBM section).  Every pass is a separate process run of the same bench command.
"""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
what = sys.argv[3] if len(sys.argv) > 3 else "bench.py"          # e.g. "bench.py --workload dift" (the fp32 net)


def family(name):
    if "gemm32" in name:
        return "gemm32"
    if "attn32" in name:
        return "attention32"
    if "gn32" in name or "ln32" in name:
        return "norms32"
    if "igemm" in name:
        return "igemm_family"
    if "attn_" in name:
        return "attention"
    if "gn_" in name:
        return "groupnorm"
    if "layernorm" in name:
        return "layernorm"
    return "other"


sums = defaultdict(lambda: defaultdict(float))
counts = defaultdict(lambda: defaultdict(int))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "dm::" not in name and "_ZN2dm" not in name and "dm32::" not in name and "_ZN4dm32" not in name:
            continue
        fam = family(name)
        sums[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        counts[fam][r["Counter_Name"]] += 1

out = {
    "command": f"rocprofv3 --pmc <set> --kernel-trace -f csv -- python {what} --steps 1 --warmup 1 --no-cpu-baseline "
               "(separate process per counter set: FETCH_SIZE | WRITE_SIZE | SQ set A | SQ set B; each pass = "
               f"{steps} steps: 1 warm-up + 1 timed)",
    "steps_in_pass": steps,
    "units": "FETCH_SIZE/WRITE_SIZE in KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, "
             "MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected; SQ_* as reported",
    "kernels": {},
}
for fam in sums:
    s, c = sums[fam], counts[fam]
    k = {}
    n = max(c.values())
    k["launches_in_pass"] = n
    if "FETCH_SIZE" in s:
        k["fetch_bytes_per_launch"] = s["FETCH_SIZE"] * 1024 * 2 / c["FETCH_SIZE"]
        k["fetch_GB_per_step"] = s["FETCH_SIZE"] * 1024 * 2 / steps / 1e9
    if "WRITE_SIZE" in s:
        k["write_bytes_per_launch"] = s["WRITE_SIZE"] * 1024 / c["WRITE_SIZE"]
        k["write_GB_per_step"] = s["WRITE_SIZE"] * 1024 / steps / 1e9
    if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
        k["hbm_bytes_per_launch"] = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
    sq = {n_: v for n_, v in s.items() if n_.startswith("SQ_")}
    if sq:
        k["sq"] = sq
        wc = sq.get("SQ_WAVE_CYCLES")
        if wc:
            k["sq_fractions_of_wave_cycles"] = {
                a: sq[b] / wc for a, b in (("active", "SQ_ACTIVE_INST_ANY"), ("wait_any", "SQ_WAIT_ANY"),
                                           ("wait_inst_any", "SQ_WAIT_INST_ANY")) if b in sq}
        if sq.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_bank_conflict_frac"] = sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][fam] = k
print(json.dumps(out, indent=1, sort_keys=True))
