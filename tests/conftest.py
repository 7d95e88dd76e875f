"""pytest configuration: registers the `gpu` marker and puts the package dir
(`nerf-cuda_amd/`, not importable by name because of the hyphen) on sys.path."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "nerf-cuda_amd", ROOT / "tests", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
