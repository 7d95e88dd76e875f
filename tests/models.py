"""Shared model builders: re-exported from the package (nerf-cuda_amd/models.py), where bench.py and the scripts take them from."""
import importlib.util
import sys
from pathlib import Path

_p = Path(__file__).resolve().parent.parent / "nerf-cuda_amd" / "models.py"
_spec = importlib.util.spec_from_file_location("nrf_models_impl", _p)
_m = importlib.util.module_from_spec(_spec)
if str(_p.parent) not in sys.path:
    sys.path.insert(0, str(_p.parent))
_spec.loader.exec_module(_m)
build_model, psnr = _m.build_model, _m.psnr
