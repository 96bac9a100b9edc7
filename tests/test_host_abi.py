"""CPU: the C-ABI shared library loads, exports every symbol include/etx_hip.h declares, fails loudly without a GPU;
host-side logic (snapshot relocation, BVH build, option mapping) works without a device."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "etx_hip.h")) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(etx_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(etx):
    lib = etx.Library.get()
    symbols = declared_symbols()
    assert len(symbols) >= 18
    missing = [s for s in symbols if not lib.has_symbol(s)]
    assert missing == []
    assert set(etx.api.EXPORTED_SYMBOLS) == set(symbols)
    assert lib.lib.etx_hip_abi_version() == 4  # 2: etx_hip_stats_t grew (pool_grows), etx_hip_set_pool_policy; 3: reference_seeding, asynchronous film reduce; 4: runtime info, small collectives, timed trace


def test_missing_library_is_an_error(etx, tmp_path):
    with pytest.raises(FileNotFoundError):
        etx.Library(str(tmp_path / "libetx_hip.so"))


def test_abi_struct_sizes(etx):
    import ctypes
    assert ctypes.sizeof(etx.VCMOptions) == 32       # etx::VCMOptions, SURVEY.md 8
    assert etx.scene_snapshot.SCENE_SIZE == 528 and etx.scene_snapshot.CAMERA_SIZE == 176
    o = etx.VCMOptions.default_values()
    assert (o.options, o.radius_decay, o.kernel, o.initial_radius) == (0x7F, 256, 1, 0.0)  # vcm_shared.cxx:6-13


@pytest.mark.parametrize("name,triangles,size", [("cornell_classic_128", 32, (128, 128)), ("cornell_full_128", 44, (128, 128)),
                                                 ("cornell_classic_1080p", 32, (1920, 1080)), ("cornell_full_512", 44, (512, 512))])
def test_snapshot_relocation(etx, golden_dir, name, triangles, size):
    snap = etx.SceneSnapshot(os.path.join(golden_dir, name + ".etxscene"))
    assert snap.film_size == size
    assert snap.triangle_count == triangles
    assert snap.max_path_length == 1023
    assert abs(snap.bounding_sphere_radius - 3.0 ** 0.5) < 1e-5
    v = snap.vertices()
    t = snap.triangles()
    assert v.shape[1] == 14 and t.shape == (triangles, 8)
    assert t[:, 0:3].max() < v.shape[0]
    # unit normals and the box extents of scenes/make_scenes.py
    np.testing.assert_allclose(np.linalg.norm(v[:, 3:6], axis=1), 1.0, atol=1e-5)
    assert v[:, 0].min() == -1.0 and v[:, 0].max() == 1.0 and v[:, 1].min() == 0.0 and v[:, 1].max() == 2.0


def test_host_bvh_invariants(etx, golden_dir):
    from etx_tracer_amd import api
    for name in ("cornell_classic_128", "cornell_full_128", "cornell_gems_128"):
        snap = etx.SceneSnapshot(os.path.join(golden_dir, name + ".etxscene"))
        rc, info = api.host_check_bvh(snap)  # BVH2 and the BVH4 the device traverses: every triangle in one leaf, boxes nested
        assert rc == 0
        assert info["triangles"] == snap.triangle_count
        assert 1 <= info["nodes"] < snap.triangle_count / 2  # four-wide nodes over leaves of up to four triangles
        assert info["depth"] <= 10                          # three stack entries per level must fit the 32-entry stack
        assert info["bytes"] == info["nodes"] * 128 + info["triangles"] * 48
        if name == "cornell_gems_128":
            assert info["nodes"] <= 1024, info  # a few hundred nodes: mostly inside the LDS-staged top of the tree


def test_options_mapping_follows_vcmoptions_load(etx):
    from etx_tracer_amd import api, integrator
    o = integrator.vcm_options_from_dict({})
    assert o.options == api.VCM_FULL_OPTIONS and o.blue_noise == 1
    o = integrator.vcm_options_from_dict({"vcm-merging": False, "vcm-blue_noise": False, "vcm-radius_decay": 64})
    assert (o.options & api.VCM_ENABLE_MERGING) == 0 and o.blue_noise == 0 and o.radius_decay == 64
    o = integrator.vcm_options_from_dict({"vcm-connect_vertices": False, "vcm-mis": False})
    assert (o.options & api.VCM_CONNECT_VERTICES) == 0 and (o.options & api.VCM_ENABLE_MIS) == 0
    assert o.options & api.VCM_CONNECT_TO_LIGHT


def test_create_without_gpu_fails_loudly(etx):
    if os.path.exists("/dev/kfd"):  # (not torch.cuda.is_available(): importing the torch wheel maps its own, older ROCm runtime into the process)
        pytest.skip("a GPU is present")
    from etx_tracer_amd import api
    with pytest.raises(api.EtxHipError) as e:
        api.Context(0)
    assert e.value.code == -2  # ETX_HIP_ERROR_NO_DEVICE: no CPU fallback exists


def test_cpp_binding_fails_loudly_without_a_device():
    """integration/etx_hip_integrators.hxx compiled into the reference-based driver: without a gfx950 device etx_hip_create
    fails, HIPVCM::run() logs the error and stays Stopped (the reference's convention), the driver exits with code 6 - it
    never falls back to the CPU integrator."""
    import subprocess
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present: the binding is exercised by tests/test_gpu_binding.py")
    oracle = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
    if not os.path.exists(oracle):
        pytest.skip("oracle/_ref/etx_oracle is not built (needs /root/reference)")
    result = subprocess.run([oracle, "--load-snapshot", os.path.join(ROOT, "tests", "golden", "cornell_classic_128.etxscene"), "--integrator", "hip-vcm", "--spp", "2"],
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert result.returncode == 6, result.stdout[-1000:]
    assert "no HIP device available" in result.stdout


def test_headless_driver_of_the_backend_links_no_cpu_integrator():
    """oracle/_ref/etx_hip_render = the driver built with -DETX_DRIVER_HIP_ONLY (oracle/build_ref.sh): the headless host of the HIP backend
     holds the scene loader, the Film and the binding - none of the reference's CPU integrators, and it refuses their names."""
    import subprocess
    binary = os.path.join(ROOT, "oracle", "_ref", "etx_hip_render")
    if not os.path.exists(binary):
        pytest.skip("oracle/_ref/etx_hip_render is not built (needs /root/reference)")
    symbols = subprocess.run(["nm", "-C", binary], stdout=subprocess.PIPE, text=True).stdout
    assert ("HIPVCM" in symbols) and ("HIPBidirectional" in symbols)
    for name in ("CPUVCM", "CPUPathTracing", "CPUBidirectional"):
        assert name not in symbols, name
    snapshot = os.path.join(ROOT, "tests", "golden", "cornell_classic_128.etxscene")
    result = subprocess.run([binary, "--load-snapshot", snapshot, "--integrator", "vcm"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert result.returncode == 1 and "hip-vcm" in result.stdout  # usage


def test_film_merge_iteration_patch_applies_and_compiles(tmp_path):
    """integration/film_merge_iteration.patch (the additive bulk film interface) against a scratch copy of the
    two reference files: it applies cleanly, the patched film.cxx compiles, and the binding compiles against the patched
    header with ETX_FILM_HAS_MERGE_ITERATION (its bulk publish path)."""
    import shutil
    import subprocess
    ref = "/root/reference"
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not (os.path.isdir(os.path.join(ref, "sources", "etx")) and os.path.exists(cxx)):
        pytest.skip("needs /root/reference (build container)")
    host = tmp_path / "sources" / "etx" / "render" / "host"
    host.mkdir(parents=True)
    for name in ("film.hxx", "film.cxx"):
        shutil.copy(os.path.join(ref, "sources", "etx", "render", "host", name), host / name)
    patch = os.path.join(ROOT, "integration", "film_merge_iteration.patch")
    subprocess.check_call(["patch", "-p1", "-s", "-i", patch], cwd=tmp_path)
    t = os.path.join(ref, "thirdparty")
    flags = ["-std=c++23", "-include", "atomic", "-include", "map", "-DNDEBUG", "-D_stricmp=strcasecmp", "-DETX_LIBRARY=1", "-w", "-fsyntax-only", "-I%s" % (tmp_path / "sources"), "-I%s/sources" % ref, "-I%s" % t,
             "-I%s/enkits" % t, "-I%s/bluenoise" % t, "-I%s/json" % t, "-I%s" % os.path.join(ROOT, "include"), "-I%s" % os.path.join(ROOT, "integration")]
    subprocess.check_call([cxx] + flags + [str(host / "film.cxx")])
    tu = tmp_path / "binding.cxx"
    tu.write_text("#define ETX_FILM_HAS_MERGE_ITERATION 1\n#include <etx/rt/rt.hxx>\n#include <etx_hip_integrators.hxx>\nint main() { return 0; }\n")
    subprocess.check_call([cxx] + flags + [str(tu)])


def test_public_headers_are_plain_c(tmp_path):
    """The drop-in boundary is a C ABI (cgo / JNI / ctypes-style bindings include these headers): both compile as C99, pedantic."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    source = tmp_path / "c_abi.c"
    source.write_text('#include "include/etx_scene_abi.h"\n#include "include/etx_hip.h"\nint main(void) { return (int)sizeof(etx_hip_stats_t) * 0; }\n')
    result = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + root, "-fsyntax-only", str(source)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert result.returncode == 0, result.stdout
