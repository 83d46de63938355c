"""pytest configuration: `gpu` marker, seeded synthetic model files (session-scoped)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools import ggml_io  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model_dir(tmp_path_factory):
    return tmp_path_factory.mktemp("models")


@pytest.fixture(scope="session")
def model_file(model_dir):
    """model_file(preset, seed=1234, ftype=None, **overrides) -> path of a cached synthetic ggml file."""
    cache = {}

    def get(preset="micro.en", seed=1234, ftype=None, quiet_special=False, **ov):
        key = (preset, seed, ftype, quiet_special, tuple(sorted(ov.items())))
        if key not in cache:
            name = "ggml-%s-s%d-f%s-q%d-%d.bin" % (preset, seed, ftype, int(quiet_special), len(cache))
            path = str(model_dir / name)
            ggml_io.make_model_file(path, preset, seed, ftype, quiet_special=quiet_special, **ov)
            cache[key] = path
        return cache[key]

    return get


def load_package():
    """Import obs-localvocal_amd/ (hyphenated directory) as module `obs_localvocal_amd`."""
    import importlib.util
    if "obs_localvocal_amd" in sys.modules:
        return sys.modules["obs_localvocal_amd"]
    path = os.path.join(ROOT, "obs-localvocal_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("obs_localvocal_amd", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["obs_localvocal_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def pkg():
    return load_package()
