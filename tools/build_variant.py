#!/usr/bin/env python
"""A variant build of libdm_engine.so for same-box A/B runs (tools/ab_lib.sh, tools/ab_libs_abba.sh):
    python tools/build_variant.py <name> -DDM_PERS_AHEAD=0 [more hipcc flags]
compiles every translation unit with the extra flags into diff-mining_amd/csrc/build/variant_<name>/ and links
diff-mining_amd/lib/libdm_engine_<name>.so (git-ignored; travels to the GPU box like the product library)."""
import importlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("diff-mining_amd.build")
name, extra = sys.argv[1], sys.argv[2:]
obj = os.path.join(b.OBJ, "variant_" + name)
os.makedirs(obj, exist_ok=True)


def cc(src):
    o = os.path.join(obj, src.replace(".hip", ".o"))
    subprocess.run([b._hipcc()] + b.FLAGS + extra + ["-c", os.path.join(b.CSRC, src), "-o", o], check=True, capture_output=True)
    return o


with ThreadPoolExecutor(max_workers=8) as ex:
    objs = list(ex.map(cc, b.SOURCES))
out = os.path.join(b.LIBDIR, f"libdm_engine_{name}.so")
subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
