"""GPU (-m gpu): the process runs on ONE ROCm runtime - the one libetx_hip.so was built and linked against.

Up to round 5 collecting tests/ imported torch, whose wheel bundles ROCm 7.0.2 under the sonames of /opt/rocm's 7.2 libraries: the dynamic
loader then bound libetx_hip.so to the older copies. Since round 6 nothing a `-m gpu` run collects imports torch
(tests/lazy_torch.py), the library binds all its symbols when it is loaded (RTLD_NOW, -z now), bench.py needs no torch at all (its collectives are
RCCL inside the library), and etx_hip_create refuses a runtime older than the one it was compiled against."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_one_rocm_runtime_in_the_process(etx, gpu_context):
    assert "torch" not in sys.modules, "something a -m gpu run collects imported torch (and with it the wheel's own ROCm runtime)"
    info = etx.api.runtime_info()
    assert info["hip_runtime"] // 100000 >= info["hip_built_against"] // 100000, info
    assert info["rccl_runtime"] >= info["rccl_built_against"], info
    for name in ("libamdhip64", "libhsa-runtime64", "librccl"):
        copies = [m for m in info["mapped"] if name in m]
        assert len(copies) == 1, "%s is mapped %d times: %s" % (name, len(copies), copies)
        assert "/torch/" not in copies[0], copies


@pytest.mark.gpu
def test_an_older_runtime_loaded_first_is_refused():
    """The round-5 configuration, on purpose, in a child process: torch first, then the library - the loader binds libetx_hip.so to the ROCm runtime bundled
    with the wheel. That is the configuration in which the round-5 suite died (heap corruption in one of six processes). etx_hip_create names
    the runtime it found and refuses; ETX_HIP_ALLOW_OLDER_RUNTIME=1 is the documented way to run on it anyway. (The round-5 library has neither the
    check nor etx_hip_runtime_info: this test fails on it.)"""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "import etx_tracer_amd as etx\n"
        "info = etx.api.runtime_info()\n"
        "print('INFO', info['hip_runtime'], info['hip_built_against'], ' '.join(info['mapped']))\n"
        "try:\n"
        "    etx.api.Context(0).close(); print('CREATED')\n"
        "except etx.EtxHipError as e:\n"
        "    print('REFUSED', e.code, str(e))\n" % ROOT)
    env = dict(os.environ)
    env.pop("ETX_HIP_ALLOW_OLDER_RUNTIME", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env).stdout
    info = [l for l in out.splitlines() if l.startswith("INFO ")]
    assert info, out
    runtime, built = int(info[0].split()[1]), int(info[0].split()[2])
    assert "/torch/lib/libamdhip64" in info[0], info[0]  # torch first: its copy is the one the library was bound to
    if runtime // 100000 >= built // 100000:
        assert "CREATED" in out, out
        pytest.skip("the torch wheel of this image bundles a runtime at least as new as the library's")
    assert "REFUSED -3" in out and "older than" in out and "torch" in out, out
    env["ETX_HIP_ALLOW_OLDER_RUNTIME"] = "1"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env).stdout
    assert "CREATED" in out, out


def test_collecting_the_gpu_tests_does_not_import_torch():
    """CPU: what `python -m pytest tests/ -m gpu` imports while it collects (the driver's command) - no torch, so no second ROCm runtime."""
    code = (
        "import sys, pytest\n"
        "class Probe:\n"
        "    def pytest_collection_finish(self, session):\n"
        "        mapped = [l for l in open('/proc/self/maps') if '/torch/lib/' in l]\n"
        "        open(sys.argv[1], 'w').write('%d %d' % ('torch' in sys.modules, len(mapped)))\n"
        "sys.exit(pytest.main(['tests/', '-q', '-m', 'gpu', '--collect-only', '-p', 'no:cacheprovider'], plugins=[Probe()]))\n")
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".txt") as result:
        subprocess.run([sys.executable, "-c", code, result.name], cwd=ROOT, capture_output=True, text=True, timeout=600)
        imported, mapped = open(result.name).read().split()
    assert (imported, mapped) == ("0", "0")
