"""CPU tier: the C-ABI library builds, loads and exports every symbol include/dm_engine.h declares;
host-only entry points agree with the oracle; the product path fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from diff_mining_amd import engine as E
from oracle import unet_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(E.LIB_PATH):
        from diff_mining_amd import build
        build.build()
    return E.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dm_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(E.SYMBOLS), declared ^ set(E.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), f"{s} declared in include/dm_engine.h but not exported"
    assert b"gfx950" in lib.dm_version()


def test_scheduler_table_matches_oracle(lib):
    a = E.scheduler_alphas_cumprod()
    b = R.alphas_cumprod().numpy()
    assert np.abs(a - b).max() < 1e-7
    assert np.array_equal(a.astype(np.float16), b.astype(np.float16))     # identical after the fp16 cast (R3)
    for t, v in {0: 0.99914998, 161: 0.81210744, 261: 0.65566903, 500: 0.27633247, 999: 0.00466010}.items():
        assert abs(float(a[t]) - v) < 2e-7


def test_sinusoid_matches_oracle(lib):
    worst = 0.0
    for t in (0, 1, 161, 261, 500, 999):
        s = E.timestep_sinusoid(t)
        r = R.timestep_sinusoid(torch.tensor([t]))[0].numpy()
        worst = max(worst, float(np.abs(s - r).max()))
    assert worst < 1e-4      # fp32 argument rounding at t*f ~ 1e3; fp16 ulp near 1 is 4.9e-4
    np.testing.assert_allclose(E.timestep_sinusoid(161)[:2], [-0.71177477, 0.36481935], atol=2e-5)


def test_dift_shape(lib):
    assert E.dift_shape(64, 64, 1) == (1280, 32, 32)       # DIFT-161 tap: up_blocks[1] incl. its upsampler
    assert E.dift_shape(64, 64, 0) == (1280, 16, 16)
    assert E.dift_shape(64, 64, 2) == (640, 64, 64)
    assert E.dift_shape(64, 64, 3) == (320, 64, 64)
    assert E.dift_shape(32, 42, 1) == (1280, 16, 21)       # odd sizes follow the skip tensor (upsample_size)


def test_engine_fails_loudly_without_gpu(lib):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError):
        E.UNetEngine(0)
    h = E.C.c_void_p()
    assert lib.dm_engine_create(0, E.C.byref(h)) != 0
    assert b"no HIP device" in lib.dm_last_error(None) or b"fallback" in lib.dm_last_error(None)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(E.EngineError):
        E.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "diff-mining_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("the oracle", "").replace("CPU oracle", ""), f


def test_tap_reuse_kernels_do_not_spill():
    """igemm_pers_tr.hip picks, per instantiation, the unrolled or the run-time-dx form of its k loop by whether the register
    allocator handles it without spilling (a spilled accumulator is reloaded inside the k loop, behind the LDS-DMA on vmcnt).
    That list is a property of the compiler, so it is checked against the compiled ISA: every instantiation of the kernel that
    the library ships must have no spilled VGPR."""
    import importlib
    import re
    import subprocess
    import tempfile
    b = importlib.import_module("diff-mining_amd.build")
    src = os.path.join(os.path.dirname(b.__file__), "csrc", "igemm_pers_tr.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "tr.s")
        subprocess.run([b._hipcc()] + b.FLAGS + ["-S", "--cuda-device-only", "-o", out, src], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    found = re.findall(r"\.name:\s+(\S*igemm_pers_tr_kernel\S*).*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    assert len(found) == 13, found
    spilled = [(n, int(c)) for n, c in found if int(c) != 0]
    assert not spilled, f"tap-reuse kernels with spilled registers (move them to the run-time-dx loop: TrUnroll): {spilled}"
    # the folded up-sampler (igemm_pers_up.hip, template parameter UP4 of the persistent tile) carries a parity class and a scattering
    # epilogue on top of the plain kernel's state: same check
    src = os.path.join(os.path.dirname(b.__file__), "csrc", "igemm_pers_up.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "up.s")
        subprocess.run([b._hipcc()] + b.FLAGS + ["-S", "--cuda-device-only", "-o", out, src], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = open(out).read()
    found = re.findall(r"\.name:\s+(\S*igemm_pers_kernelILi0ELb0ELi0ELb0ELb0ELb1E\S*).*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    assert len(found) == 1 and int(found[0][1]) == 0, found
