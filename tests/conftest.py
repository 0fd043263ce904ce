import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The in-tree liballset_hip.so must exist for every test session (hipcc cross-compiles on CPU)."""
    from allset_amd.build import build_library
    build_library()


# ---- opt-in audit (tools/kernel_audit.py): ALLSET_ABI_TRACE=<file.json> records, for every C-ABI entry point, how often the session's
# tests called it, under which dense arithmetic mode, and the first tests that did -- "which -m gpu test pins this entry".  Off by
# default: the wrappers cost a Python call per launch.
_ABI_TRACE = os.environ.get("ALLSET_ABI_TRACE")
_abi_calls = {}
_abi_current = [""]


def pytest_runtest_setup(item):
    _abi_current[0] = item.nodeid


def pytest_sessionstart(session):
    if not _ABI_TRACE:
        return
    from allset_amd import _lib, dense
    lib = _lib.load()

    def wrap(name, fn):
        def traced(*a):
            rec = _abi_calls.setdefault(name, {"calls": 0, "modes": {}, "tests": []})
            rec["calls"] += 1
            mode = dense.get_arithmetic()
            rec["modes"][mode] = rec["modes"].get(mode, 0) + 1
            t = _abi_current[0].split("[")[0]
            if t and t not in rec["tests"] and len(rec["tests"]) < 6:
                rec["tests"].append(t)
            return fn(*a)
        return traced
    for name in list(_lib.SIGNATURES) + ["allset_last_error", "allset_version"]:
        setattr(lib, name, wrap(name, getattr(lib, name)))


def pytest_sessionfinish(session, exitstatus):
    if _ABI_TRACE and _abi_calls:
        import json
        with open(_ABI_TRACE, "w") as f:
            json.dump(_abi_calls, f, indent=1, sort_keys=True)
