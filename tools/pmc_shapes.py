#!/usr/bin/env python
"""Per-shape HBM traffic of the igemm family from PMC counters (VERDICT r02 next #2c: name the source of the 2.5x
counter traffic).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $O/pmc_fetch -o pmc -- python tools/pmc_shapes.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $O/pmc_write -o pmc -- python tools/pmc_shapes.py run
    python tools/pmc_shapes.py parse $O/pmc_fetch $O/pmc_write > profiles/<tag>_pmc_shapes.txt

`run` launches every shape REPS times through the operator-level C ABI at the bench batch, a marker kernel (torch fill of
a tagged size) between shapes; `parse` folds the per-dispatch counters by shape (FETCH_SIZE in KiB, doubled per the gfx950
note of MI355X_MICROARCH.md; WRITE_SIZE in KiB) next to the algorithmic bytes (activations once, weights once, output
once, residual once)."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B, REPS = 160, 3
# (name, mode, H, W, C1, C2, Cout, epi, extra)
SHAPES = [
    ("ff1 geglu 320->2560 @64 (ln)", 0, 64, 64, 320, 0, 2560, 1, "ln"),
    ("proj 320->320 @64 + res", 0, 64, 64, 320, 0, 320, 0, "res"),
    ("ff1 geglu 640->5120 @32 (ln)", 0, 32, 32, 640, 0, 5120, 1, "ln"),
    ("conv 3x3 640->640 @32 + temb", 1, 32, 32, 640, 0, 640, 0, "temb"),
    ("conv 3x3 1280->1280 @16 + res", 1, 16, 16, 1280, 0, 1280, 0, "res"),
    ("conv 3x3 320->320 @64 + temb", 1, 64, 64, 320, 0, 320, 0, "temb"),
    ("ff1 geglu 1280->10240 @16 (ln)", 0, 16, 16, 1280, 0, 10240, 1, "ln"),
    ("proj 640->640 @32 + res", 0, 32, 32, 640, 0, 640, 0, "res"),
    ("proj 1280->1280 @16 + res", 0, 16, 16, 1280, 0, 1280, 0, "res"),
    ("ff2 1280->320 @64 + res", 0, 64, 64, 1280, 0, 320, 0, "res"),
    ("qkv 320->960 @64 (ln)", 0, 64, 64, 320, 0, 960, 0, "ln"),
    ("conv 3x3 cat 640+320->320 @64 + temb", 1, 64, 64, 640, 320, 320, 0, "temb"),
    ("up 3x3 640->640 @32->64 (up_fold 0)", 3, 32, 32, 640, 0, 640, 0, ""),
    ("up folded 4 x 2x2 640->640 @32->64", 5, 32, 32, 640, 0, 640, 0, "up4"),       # what ships (option up_fold = 1, igemm_pers_up.hip)
]


def algo_bytes(mode, H, W, C1, C2, Cout, epi, extra):
    taps = 1 if mode == 0 else (16 if mode == 5 else 9)            # folded up-sampler: four classes x four taps of pre-summed weights
    OH, OW = (2 * H, 2 * W) if mode in (3, 5) else (H, W)
    M = B * OH * OW
    rd = B * H * W * (C1 + C2) * 2 + Cout * taps * (C1 + C2) * 2
    if extra == "res":
        rd += M * Cout * 2
    wr = M * (Cout // 2 if epi else Cout) * 2
    return rd, wr


def run():
    import torch
    from tests import gpu_util as U
    lib = U.E.load_library()
    d = U.dev()
    g = torch.Generator(device="cuda").manual_seed(1)
    for si, (name, mode, H, W, C1, C2, Cout, epi, extra) in enumerate(SHAPES):
        if os.environ.get("DM_PMC_FILTER") and os.environ["DM_PMC_FILTER"] not in name:
            continue
        taps = 1 if mode == 0 else 9
        Cin = C1 + C2
        OH, OW = (2 * H, 2 * W) if mode in (3, 5) else (H, W)
        M = B * OH * OW
        x = (torch.randn(B, H, W, C1, device=d, generator=g) * 0.5).half()
        x2 = (torch.randn(B, H, W, C2, device=d, generator=g) * 0.5).half() if C2 else None
        w = (torch.randn(Cout, taps * Cin, device=d, generator=g) * (taps * Cin) ** -0.5).half()
        bias = torch.zeros(Cout, device=d).half()
        temb = torch.randn(B, Cout, device=d, generator=g).half() if extra == "temb" else None
        res = torch.randn(B, OH, OW, Cout, device=d, generator=g).half() if extra == "res" else None
        y = torch.empty(B, OH, OW, Cout // 2 if epi else Cout, device=d, dtype=torch.float16)
        st = U.stream()
        ln_s, ln_t = w.float().sum(1).contiguous(), torch.zeros(Cout, device=d)
        big = torch.empty(1 << 28, dtype=torch.float16, device=d)          # 512 MiB: evicts L2 + MALL between launches
        torch.cuda.synchronize()
        torch.zeros(4096 + si, device=d)                                   # marker: a fill kernel of a tagged size
        w4 = (torch.randn(4, Cout, 4 * Cin, device=d, generator=g) * (4 * Cin) ** -0.5).half() if extra == "up4" else None
        torch.cuda.synchronize()
        for _ in range(REPS):
            big.fill_(1.0)
            if extra == "up4":
                assert lib.dm_op_upconv_folded(st, U.ptr(x), U.ptr(w4), U.ptr(bias), U.ptr(y), B, H, W, Cin, Cout) == 0
            elif extra == "ln":
                assert lib.dm_op_igemm_ln(st, U.ptr(x), U.ptr(w), U.ptr(ln_s), U.ptr(ln_t), None, U.ptr(y), M, Cin, Cout, epi) == 0
            else:
                assert lib.dm_op_igemm(st, U.ptr(x), U.ptr(x2), U.ptr(w), U.ptr(bias), U.ptr(temb), U.ptr(res), U.ptr(y),
                                       B, H, W, C1, C2, Cout, OH, OW, mode, epi, Cout if temb is not None else 0) == 0
        torch.cuda.synchronize()
        del x, x2, w, y, res, big


def load(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def parse(dfetch, dwrite):
    # robust attribution: count igemm dispatches per shape from the kernel order — a new shape starts after every REPS-th launch;
    # a launch is 1 dispatch, or 2 with the head / tail row split: detect by grouping consecutive igemm dispatches between fills
    def groups(rows, counter):
        gs, cur = [], []
        for r in rows:
            if r["Counter_Name"] != counter:
                continue
            if "igemm" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]:
                cur.append(float(r["Counter_Value"]))
            elif cur:
                gs.append(cur)
                cur = []
        if cur:
            gs.append(cur)
        return gs
    gf, gw = groups(load(dfetch), "FETCH_SIZE"), groups(load(dwrite), "WRITE_SIZE")
    # every launch is preceded by big.fill_ -> one group per launch; REPS groups per shape
    assert len(gf) == len(gw) == REPS * len(SHAPES), (len(gf), len(gw))
    print(f"# batch {B}; per launch, L2 + MALL flushed before every launch (512 MiB fill); FETCH_SIZE x2 (gfx950 note); bytes in MB")
    print(f"{'shape':40s} {'disp':>4s} {'fetch':>8s} {'algo rd':>8s} {'ratio':>6s} {'write':>8s} {'algo wr':>8s} {'ratio':>6s}")
    for i, sh in enumerate(SHAPES):
        f = sorted(sum(g) for g in gf[i * REPS:(i + 1) * REPS])[REPS // 2] * 1024 * 2
        w = sorted(sum(g) for g in gw[i * REPS:(i + 1) * REPS])[REPS // 2] * 1024
        rd, wr = algo_bytes(*sh[1:])
        print(f"{sh[0]:40s} {len(gf[i * REPS]):4d} {f / 1e6:8.1f} {rd / 1e6:8.1f} {f / rd:6.2f} {w / 1e6:8.1f} {wr / 1e6:8.1f} {w / wr:6.2f}")


def parse_sq(dirs):
    """SQ counters per shape (r06, VERDICT r05 #1: which unit saturates): `DM_PMC_FILTER=<substr> rocprofv3 --pmc <SQ set> ... pmc_shapes.py run`
    once per counter set, then `pmc_shapes.py parse_sq <dir>...`.  One launch group per fill marker, REPS groups per shape; counters are
    summed over a launch's dispatches and the median launch is printed, with the ratios that say where the wave cycles go."""
    flt = os.environ.get("DM_PMC_FILTER", "")
    shapes = [sh for sh in SHAPES if flt in sh[0]]
    vals = {}
    for d in dirs:
        rows = load(d)
        names = sorted({r["Counter_Name"] for r in rows})
        for c in names:
            gs, cur = [], []
            for r in rows:
                if r["Counter_Name"] != c:
                    continue
                if "igemm" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]:
                    cur.append(float(r["Counter_Value"]))
                elif cur:
                    gs.append(cur)
                    cur = []
            if cur:
                gs.append(cur)
            assert len(gs) == REPS * len(shapes), (c, len(gs), len(shapes))
            vals[c] = [sorted(sum(g) for g in gs[i * REPS:(i + 1) * REPS])[REPS // 2] for i in range(len(shapes))]
    for i, sh in enumerate(shapes):
        v = {c: vals[c][i] for c in vals}
        print(sh[0])
        for c in sorted(v):
            print(f"    {c:30s} {v[c]:.4g}")
        wc = v.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM_RD"):
                if c in v:
                    print(f"    {c + ' / SQ_WAVE_CYCLES':42s} {v[c] / wc:.3f}")
        if "SQ_BUSY_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            # SQ_BUSY_CYCLES is summed over the SEs / XCDs that report it; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD with an MFMA in
            # flight (MI355X_MICROARCH.md: = 16 x N_mfma for 16x16x32).  Their ratio per SIMD is the matrix-pipe duty.
            print(f"    {'SQ_VALU_MFMA_BUSY_CYCLES / (4 x 256 SIMDs)':42s} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024:.4g}   per-SIMD cycles with an MFMA executing")
            print(f"    {'SQ_BUSY_CYCLES':42s} {v['SQ_BUSY_CYCLES']:.4g}")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    elif sys.argv[1] == "parse_sq":
        parse_sq(sys.argv[2:])
    else:
        parse(sys.argv[2], sys.argv[3])
