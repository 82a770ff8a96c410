"""Import alias: the product package lives in `diff-mining_amd/` (the name the build contract
fixes), which is not a valid Python identifier.  `import diff_mining_amd` loads that directory
as a regular package under this importable name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "diff-mining_amd")
_spec = _ilu.spec_from_file_location("diff_mining_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["diff_mining_amd"] = _mod
_spec.loader.exec_module(_mod)
