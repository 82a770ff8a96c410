import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sd15_weights_f16():
    """Full-size synthetic SDv1.5 U-Net weights (fp16 ndarray per diffusers name), ~28 s to build."""
    from diff_mining_amd import synth
    return synth.synth_state_dict(seed=0, dtype=np.float16)


@pytest.fixture(scope="session")
def sd15_weights_torch(sd15_weights_f16):
    import torch
    return {k: torch.from_numpy(v).float() for k, v in sd15_weights_f16.items()}
