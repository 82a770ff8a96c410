"""Builds libdm_engine.so (hand-written gfx950 HIP kernels + C-ABI runtime) in-tree with hipcc.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the CPU-only build
container; the resulting .so is git-ignored but travels to the GPU box with the repo snapshot.

A translation unit is recompiled when the sha256 of (its source, every header, the flags, the hipcc
version) differs from the stamp written next to its object file — content, not mtimes (a snapshot copy
resets mtimes).  `build()` prints `compiled N/<TUs> TUs` so a build check can see what it exercised;
`force=True` (or DM_BUILD_FORCE=1) recompiles everything.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdm_engine.so")
SOURCES = ["igemm.hip", "igemm_pers.hip", "igemm_pers_ln.hip", "igemm_pers_part.hip", "igemm_pers_ws.hip", "igemm_ws.hip", "igemm_pers_sc.hip", "igemm_sc.hip", "igemm_pers_tr.hip", "igemm_pers_up.hip", "igemm_ko.hip", "igemm_ln.hip", "igemm64.hip", "igemm_splitk.hip", "attention.hip", "attention_pipe.hip", "attention_qk32.hip", "attention_d160.hip", "attention_pp.hip", "attention_pipe80.hip", "attention_cross.hip", "norm.hip", "misc.hip", "conv_out.hip", "vae.hip", "clip.hip", "f32_gemm.hip", "f32_ops.hip", "unet_f32.hip", "probe_peak.hip", "engine.hip"]
HEADERS = [os.path.join(CSRC, "dm_kernels.h"), os.path.join(CSRC, "f32_kernels.h"), os.path.join(CSRC, "arena.h"), os.path.join(CSRC, "igemm_tile.h"), os.path.join(CSRC, "igemm_pers_tile.h"), os.path.join(os.path.dirname(HERE), "include", "dm_engine.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


_hipcc_id_cache = {}


def _hipcc_id(hipcc: str) -> bytes:
    if hipcc not in _hipcc_id_cache:
        r = subprocess.run([hipcc, "--version"], capture_output=True, text=True)
        _hipcc_id_cache[hipcc] = (r.stdout + r.stderr).encode()
    return _hipcc_id_cache[hipcc]


def tu_digest(src: str, hipcc: str) -> str:
    """Content hash of everything one object file depends on."""
    h = hashlib.sha256()
    h.update(_read(os.path.join(CSRC, src)))
    for hd in HEADERS:
        h.update(hd.encode())
        h.update(_read(hd))
    h.update(" ".join(FLAGS).encode())
    h.update(_hipcc_id(hipcc))
    return h.hexdigest()


def _stamp(obj: str) -> str:
    return obj + ".sha256"


def _tu_stale(src: str, obj: str, hipcc: str) -> bool:
    if not os.path.exists(obj) or not os.path.exists(_stamp(obj)):
        return True
    return _read(_stamp(obj)).decode().strip() != tu_digest(src, hipcc)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    force = force or os.environ.get("DM_BUILD_FORCE", "0") not in ("", "0")
    t0 = time.time()
    jobs = []
    for src in SOURCES:
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _tu_stale(src, o, hipcc):
            jobs.append((src, o, [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", o]))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    def compile_one(job):
        src, o, cmd = job
        if os.path.exists(_stamp(o)):
            os.remove(_stamp(o))
        run(cmd)
        with open(_stamp(o), "w") as f:
            f.write(tu_digest(src, hipcc) + "\n")
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    link_digest = hashlib.sha256("".join(_read(_stamp(o)).decode() for o in objs).encode()).hexdigest()
    linked = False
    if force or jobs or not os.path.exists(LIB) or not os.path.exists(_stamp(LIB)) or _read(_stamp(LIB)).decode().strip() != link_digest:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
        with open(_stamp(LIB), "w") as f:
            f.write(link_digest + "\n")
        linked = True
    print(f"dm_engine build: compiled {len(jobs)}/{len(SOURCES)} TUs"
          f"{' (' + ', '.join(j[0] for j in jobs) + ')' if jobs and len(jobs) < len(SOURCES) else ''}, "
          f"{'linked' if linked else 'link up to date'} {os.path.relpath(LIB, os.path.dirname(HERE))} "
          f"[{link_digest[:12]}] in {time.time() - t0:.1f} s{' (forced)' if force else ''}", file=sys.stderr, flush=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
