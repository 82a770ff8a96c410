"""diff-mining typicality hot path for MI355X (gfx950).

Host-side mirror of the reference's scoring surface (`diffmining/typicality/compute.py:95-160`,
`diffmining/typicality/dift.py:173-232`) over a C-ABI HIP library (`include/dm_engine.h`).
"""
from .unet_spec import SD15, UNetConfig, unet_tensor_spec, param_count  # noqa: F401
