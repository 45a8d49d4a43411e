import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "gr-air-modes_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"),
          os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
HIP_LIB = os.path.join(ROOT, "gr-air-modes_amd", "csrc", "libairmodes_hip.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libairmodes_emu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a HIP device skips the GPU tests; asked for explicitly
    (`-m gpu`) they fail loudly instead -- there is no CPU fallback to hide behind."""
    asked = "gpu" in (config.getoption("-m") or "")
    if asked:
        return
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device (run with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def emu_lib():
    """The product sources compiled against tests/emu (CPU fibers): kernel logic on a box
    without a GPU.  Test-only; never loaded by the product package."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu")])
    from air_modes import _capi
    return _capi.Library(EMU_LIB)


@pytest.fixture(scope="session")
def emu_lib_rare():
    """Second emulated build (tests/emu/Makefile): 4-slot block heads and ticket-ordered chained scans, so that
    the branches a normal capture hardly ever takes in the greedy chain run on every hop.  Test-only."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libairmodes_emu_rare.so"])
    from air_modes import _capi
    return _capi.Library(os.path.join(ROOT, "tests", "emu", "libairmodes_emu_rare.so"))


@pytest.fixture(scope="session")
def hip_lib():
    """The real library on a real GPU.  Fails loudly (no fallback) when it is missing."""
    if not os.path.exists(HIP_LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gr-air-modes_amd", "csrc")])
    from air_modes import _capi
    return _capi.Library(HIP_LIB)


@pytest.fixture(scope="session")
def hip_knobs_lib():
    """TEST-ONLY build of the product configuration with the environment knobs compiled in (-DAM_TEST_KNOBS): the tests
    that steer which kernels run (AIRMODES_FE / AIRMODES_GENERIC) or NaN-fill work arrays (AIRMODES_POISON) load this one;
    the product library reads nothing from the environment."""
    path = os.path.join(ROOT, "tests", "gpu_variants", "libairmodes_hip_knobs.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gr-air-modes_amd", "csrc"), "knobs"])
    from air_modes import _capi
    return _capi.Library(path)
