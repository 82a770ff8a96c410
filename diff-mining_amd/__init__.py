"""diff-mining typicality hot path for MI355X (gfx950).

Host-side mirror of the reference's scoring surface (`diffmining/typicality/compute.py:95-160`,
`diffmining/typicality/dift.py:173-232`) over a C-ABI HIP library (`include/dm_engine.h`).
The arithmetic lives in `csrc/*.hip`; nothing here falls back to PyTorch ops.
"""
from .unet_spec import SD15, UNetConfig, unet_tensor_spec, param_count  # noqa: F401


def __getattr__(name):   # lazy: importing torch / loading the .so only when the engine is used
    if name in ("UNetEngine", "EngineError", "load_library"):
        from . import engine
        return getattr(engine, name)
    if name in ("TypicalityScorer", "UNetCallable", "shard_indices", "gather_scores"):
        from . import typicality
        return getattr(typicality, name)
    if name == "SDFeaturizer":
        from .dift import SDFeaturizer
        return SDFeaturizer
    raise AttributeError(name)
