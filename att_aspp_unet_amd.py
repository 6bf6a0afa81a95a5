"""Import shim: the package directory is ``att-aspp-unet_amd/`` (a hyphen is not a
valid identifier), so ``import att_aspp_unet_amd`` loads that directory as a
regular package under this name."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "att-aspp-unet_amd")
_name = "att_aspp_unet_amd" if __name__ == "__main__" else __name__
_spec = _ilu.spec_from_file_location(_name, _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[_name] = _mod
_spec.loader.exec_module(_mod)

if __name__ == "__main__":
    # ``python -m att_aspp_unet_amd train|predict|calibrate ...``: the command line of the reference script
    # (attention_aspp_unet_pipeline_stage.py:538-556)
    _mod.pipeline.main()
