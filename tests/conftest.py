import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

if os.environ.get("ETX_TESTS_PRELOAD_TORCH"):
    # Reproduction switch for the round-5 crash (tools/gpu_calls/gpu_r6c.sh): up to round 5 collecting tests/ imported torch - and with it
    # the ROCm 7.0.2 runtime bundled in the wheel - before libetx_hip.so was loaded. Never set in a normal run.
    import torch  # noqa: F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (gfx950) device; run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def etx():
    import etx_tracer_amd
    return etx_tracer_amd


@pytest.fixture(scope="session")
def kat_reference():
    import json
    with open(os.path.join(GOLDEN, "kat_reference.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat_library():
    """oracle/liboracle_kat.so: the plain-C restatement (checker only)."""
    import ctypes
    import subprocess
    path = os.path.join(ROOT, "oracle", "liboracle_kat.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(path)


@pytest.fixture(scope="session")
def gpu_context(etx):
    from etx_tracer_amd import api
    ctx = api.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def bluenoise_64spp():
    """The reference's sample_blue_noise outputs for the 64-spp class, uint8 [128,128,256,8] (oracle/gen_golden.py)."""
    from tools import bluenoise_tables
    return bluenoise_tables.load(os.path.join(GOLDEN, "bluenoise_64spp.npz"))


@pytest.fixture(scope="session")
def bluenoise_256spp():
    """... for the 256-spp class (set 8: scenes of 129 samples and more), oracle/_ref/etx_oracle --dump-bluenoise 256."""
    from tools import bluenoise_tables
    return bluenoise_tables.load(os.path.join(GOLDEN, "bluenoise_256spp.npz"))


@pytest.fixture(scope="session")
def rgb_response():
    """(rgb float32 [391, 3], first wavelength): the rows of the reference's rgb_response table (oracle/gen_golden.py)."""
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "rgb_response.npz"))
    return z["rgb"], float(z["first_wavelength"])


@pytest.fixture(scope="session")
def cie_observer():
    """(xyz float32 [441, 3], first wavelength): spectrum::spectral_xyz of the reference (oracle/gen_golden.py)."""
    import numpy as np
    z = np.load(os.path.join(GOLDEN, "cie_observer.npz"))
    return z["xyz"], float(z["first_wavelength"])
