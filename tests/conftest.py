import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import the hyphen-named product package ugproj-detector_amd/ as `ugproj_detector_amd`."""
    name = "ugproj_detector_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "ugproj-detector_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_synth():
    load_package()
    import importlib
    return importlib.import_module("ugproj_detector_amd.synth")


@pytest.fixture(scope="session")
def pkg():
    if not os.path.exists(os.path.join(ROOT, "ugproj-detector_amd", "libhip_cascade.so")):
        import __graft_entry__
        __graft_entry__.build()
    return load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    return load_synth()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
