#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ from the reference-based oracle (oracle/_ref/etx_oracle).
Run in the build container (needs /root/reference): python3 oracle/gen_golden.py

  scene snapshots  cornell_{classic,full}_{128,512,1080p}.etxscene   reference loader -> byte-exact etx::Scene
  golden films     cornell_{classic,full}_128_vcm.npz                reference CPUVCM, 256 / 64 spp, vcm-blue_noise=false
                   cornell_full_128_vcm_bluenoise.npz                reference CPUVCM, 64 spp, VCMOptions defaults (blue noise on)
                   cornell_classic_128_pt.npz                        reference CPUPathTracing, 256 spp, bn=false (+ normal / albedo AOVs)
                   cornell_full_128_pt_bluenoise.npz                 reference CPUPathTracing, 64 spp, PTOptions defaults
                   cornell_{rough,glass}_128_{vcm,pt}.npz            all BSDF classes: reference CPUVCM 64 spp / CPUPathTracing 256 spp
                   cornell_{spectral,diamond,gems}_128_{vcm,pt}.npz  spectral mode: classic box / dispersive diamond + thinfilm / 2 892-triangle gems, VCM 64 spp, PT 256 spp
                   cornell_cloud_128_{vcm,pt}.npz                    heterogeneous medium (procedural 32^3 density in the fog box)
                   cornell_sss_128_{vcm,pt}.npz                      random-walk subsurface scattering, CPUVCM 64 spp / CPUPathTracing 256 spp
                   cornell_{textured,envmap,lens,equirect,spectex,ssscb}_128_{vcm,pt}.npz  textures + alpha cut-out + normal map / image environment map /
                                                                     thin lens + aperture image / equirectangular camera, VCM 256 spp, PT 1024 spp
  spectral         cie_observer.npz                                  spectrum::spectral_xyz of the reference (etx_hip_upload_cie_table)
  blue noise       bluenoise_64spp.npz                               the reference's sample_blue_noise for the 64-spp class,
                                                                     factorised by tools/bluenoise_tables.py (258 KiB instead of 32 MiB)
  KAT vectors      kat_reference.json                                reference header functions
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import film_io  # noqa: E402

ORACLE = os.path.join(ROOT, "oracle", "_ref", "etx_oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes", "cornell")
GOLDEN_SPP = {"classic": 256, "full": 64}  # the fog scene needs ~10x more wavefront rounds per iteration


def run(*args):
    print("+", " ".join(args))
    subprocess.check_call([ORACLE] + list(args), stdout=subprocess.DEVNULL)


def bluenoise_golden():
    from tools import bluenoise_tables
    raw_path = "/tmp/bluenoise_64.raw"
    run("--dump-bluenoise", "64", raw_path)
    raw = np.fromfile(raw_path, dtype=np.uint8)
    base, index_xor, value_xor = bluenoise_tables.factor(raw)
    assert np.array_equal(bluenoise_tables.expand(base, index_xor, value_xor).reshape(-1), raw)
    np.savez_compressed(os.path.join(GOLDEN, "bluenoise_64spp.npz"), base=base, index_xor=index_xor, value_xor=value_xor)
    film_path = "/tmp/golden_full_bluenoise.raw"
    run("--load-snapshot", os.path.join(GOLDEN, "cornell_full_128.etxscene"), "--integrator", "vcm", "--spp", "64", "--out", film_path)
    film = film_io.read_film(film_path)
    np.savez_compressed(os.path.join(GOLDEN, "cornell_full_128_vcm_bluenoise.npz"), camera=film["camera"][..., :3], light=film["light"][..., :3],
                        spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]), threads=np.int32(film["threads"]))


def pt_golden():
    # CPUPathTracing (configs[0] family): classic box, 256 spp, bn=false; fog box, 64 spp, PTOptions defaults (blue noise)
    for name, snapshot, spp, extra in (("cornell_classic_128_pt", "cornell_classic_128", 256, ["--opt", "bn=false"]),
                                       ("cornell_full_128_pt_bluenoise", "cornell_full_128", 64, [])):
        film_path = "/tmp/golden_%s.raw" % name
        run("--load-snapshot", os.path.join(GOLDEN, snapshot + ".etxscene"), "--integrator", "pt", "--spp", str(spp), "--out", film_path, *extra)
        film = film_io.read_film(film_path)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), camera=film["camera"][..., :3], normal=film["normal"][..., :3], albedo=film["albedo"][..., :3],
                            spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]), threads=np.int32(film["threads"]))


def materials_golden():
    # every BSDF class once: "rough" (diffuse variations 1 / 2, principled, velvet, plastic, rough dielectric, rough gold
    # with a thin film) and "glass" (delta dielectric, thinfilm class); scenes/make_scenes.py
    for flavour in ("rough", "glass"):
        snapshot = os.path.join(GOLDEN, "cornell_%s_128.etxscene" % flavour)
        run("--scene", os.path.join(SCENES, "%s_test_128.json" % flavour), "--integrator", "none", "--snapshot", snapshot)
        for integrator, spp, extra in (("vcm", 64, ["--opt", "vcm-blue_noise=false"]), ("pt", 256, ["--opt", "bn=false"])):
            film_path = "/tmp/golden_%s_%s.raw" % (flavour, integrator)
            run("--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path, *extra)
            film = film_io.read_film(film_path)
            np.savez_compressed(os.path.join(GOLDEN, "cornell_%s_128_%s.npz" % (flavour, integrator)), camera=film["camera"][..., :3], light=film["light"][..., :3],
                                normal=film["normal"][..., :3], albedo=film["albedo"][..., :3], spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]),
                                threads=np.int32(film["threads"]))


def spectral_golden():
    # Scene::spectral(): the classic box and the box with a dispersive diamond + thinfilm class (configs[2] family);
    # the CIE observer of the reference for etx_hip_upload_cie_table
    raw_path = "/tmp/cie.raw"
    run("--dump-cie", raw_path)
    raw = np.fromfile(raw_path, dtype=np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "cie_observer.npz"), first_wavelength=np.float32(raw[0]), xyz=raw[2:].reshape(int(raw[1]), 3))
    # rgb_response of the reference (RGB textures in spectral mode) for etx_hip_upload_rgb_response
    run("--dump-rgb-response", raw_path)
    raw = np.fromfile(raw_path, dtype=np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "rgb_response.npz"), first_wavelength=np.float32(raw[0]), rgb=raw[2:].reshape(int(raw[1]), 3))
    for flavour in ("spectral", "diamond", "gems"):  # gems: 2 892 triangles -> BVH traversal
        snapshot = os.path.join(GOLDEN, "cornell_%s_128.etxscene" % flavour)
        run("--scene", os.path.join(SCENES, "%s_test_128.json" % flavour), "--integrator", "none", "--snapshot", snapshot)
        for integrator, spp, extra in (("vcm", 64, ["--opt", "vcm-blue_noise=false"]), ("pt", 256, ["--opt", "bn=false"])):
            film_path = "/tmp/golden_%s_%s.raw" % (flavour, integrator)
            run("--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path, *extra)
            film = film_io.read_film(film_path)
            np.savez_compressed(os.path.join(GOLDEN, "cornell_%s_128_%s.npz" % (flavour, integrator)), camera=film["camera"][..., :3], light=film["light"][..., :3],
                                normal=film["normal"][..., :3], albedo=film["albedo"][..., :3], spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]),
                                threads=np.int32(film["threads"]))


def cloud_golden():
    # heterogeneous medium (delta tracking / ratio tracking, scene_medium.hxx:191-239, 284-349): the fog box with a
    # procedural 32^3 density grid injected by the oracle driver (the loader only reads .nvdb files and the tree has none)
    snapshot = os.path.join(GOLDEN, "cornell_cloud_128.etxscene")
    run("--scene", os.path.join(SCENES, "full_test_128.json"), "--inject-density", "32", "--integrator", "none", "--snapshot", snapshot)
    for integrator, spp, extra in (("vcm", 64, ["--opt", "vcm-blue_noise=false"]), ("pt", 256, ["--opt", "bn=false"])):
        film_path = "/tmp/golden_cloud_%s.raw" % integrator
        run("--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path, *extra)
        film = film_io.read_film(film_path)
        np.savez_compressed(os.path.join(GOLDEN, "cornell_cloud_128_%s.npz" % integrator), camera=film["camera"][..., :3], light=film["light"][..., :3],
                            normal=film["normal"][..., :3], albedo=film["albedo"][..., :3], spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]),
                            threads=np.int32(film["threads"]))


def sss_golden():
    # random-walk subsurface scattering in the path tracer (path_tracing_shared.hxx:64-159, 391-439)
    snapshot = os.path.join(GOLDEN, "cornell_sss_128.etxscene")
    run("--scene", os.path.join(SCENES, "sss_test_128.json"), "--integrator", "none", "--snapshot", snapshot)
    for integrator, spp, extra in (("vcm", 64, ["--opt", "vcm-blue_noise=false"]), ("pt", 256, ["--opt", "bn=false"])):
        film_path = "/tmp/golden_sss_%s.raw" % integrator
        run("--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path, *extra)
        film = film_io.read_film(film_path)
        np.savez_compressed(os.path.join(GOLDEN, "cornell_sss_128_%s.npz" % integrator), camera=film["camera"][..., :3], light=film["light"][..., :3],
                            normal=film["normal"][..., :3], albedo=film["albedo"][..., :3], spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]),
                            threads=np.int32(film["threads"]))


def sssmesh_snapshot():
    # the two subsurface materials on sphere meshes (21 772 triangles: BVH traversal instead of the flat sweep); the films of this
    # scene are the high-sample-count ones of oracle/gen_golden_hi.py (... --integrators pt --spp 1024 sssmesh, vcm,rekeyed / bdpt at 256)
    run("--scene", os.path.join(SCENES, "sssmesh_test_128.json"), "--integrator", "none", "--snapshot", os.path.join(GOLDEN, "cornell_sssmesh_128.etxscene"))


def features_golden():
    # branches no other scene reaches: albedo texture + alpha cut-out (stochastic alpha test inside traversal) + normal map;
    # image environment map (2-D sampling tables) as the only light; thin lens with an aperture image; equirectangular camera
    # spectex: the textured box in spectral mode (apply_rgb); ssscb: the subsurface box with the Christensen-Burley class
    for flavour in ("textured", "envmap", "lens", "equirect", "spectex", "ssscb"):
        snapshot = os.path.join(GOLDEN, "cornell_%s_128.etxscene" % flavour)
        run("--scene", os.path.join(SCENES, "%s_test_128.json" % flavour), "--integrator", "none", "--snapshot", snapshot)
        for integrator, spp, extra in (("vcm", 256, ["--opt", "vcm-blue_noise=false"]), ("pt", 1024, ["--opt", "bn=false", "--noise-threshold", "0"])):
            film_path = "/tmp/golden_%s_%s.raw" % (flavour, integrator)
            run("--load-snapshot", snapshot, "--integrator", integrator, "--spp", str(spp), "--out", film_path, *extra)
            film = film_io.read_film(film_path)
            np.savez_compressed(os.path.join(GOLDEN, "cornell_%s_128_%s.npz" % (flavour, integrator)), camera=film["camera"][..., :3], light=film["light"][..., :3],
                                normal=film["normal"][..., :3], albedo=film["albedo"][..., :3], spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]),
                                threads=np.int32(film["threads"]))


def main():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scenes", "make_scenes.py")])
    for flavour in ("classic", "full"):
        for json_name, tag in (("test_128", "128"), ("c1_512", "512"), ("c2_1080p", "1080p")):
            run("--scene", os.path.join(SCENES, "%s_%s.json" % (flavour, json_name)), "--integrator", "none", "--snapshot",
                os.path.join(GOLDEN, "cornell_%s_%s.etxscene" % (flavour, tag)))
        snapshot = os.path.join(GOLDEN, "cornell_%s_128.etxscene" % flavour)
        film_path = "/tmp/golden_%s.raw" % flavour
        run("--load-snapshot", snapshot, "--integrator", "vcm", "--spp", str(GOLDEN_SPP[flavour]), "--opt", "vcm-blue_noise=false", "--out", film_path)
        film = film_io.read_film(film_path)
        np.savez_compressed(os.path.join(GOLDEN, "cornell_%s_128_vcm.npz" % flavour), camera=film["camera"][..., :3], light=film["light"][..., :3],
                            spp=np.int32(film["spp"]), seconds=np.float64(film["seconds"]), threads=np.int32(film["threads"]))
    bluenoise_golden()
    pt_golden()
    materials_golden()
    spectral_golden()
    cloud_golden()
    sss_golden()
    sssmesh_snapshot()
    features_golden()
    with open(os.path.join(GOLDEN, "kat_reference.json"), "w") as f:
        subprocess.check_call([ORACLE, "--kat"], stdout=f)


if __name__ == "__main__":
    if (len(sys.argv) > 1) and (sys.argv[1] == "bluenoise"):
        bluenoise_golden()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "pt"):
        pt_golden()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "materials"):
        materials_golden()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "spectral"):
        spectral_golden()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "cloud"):
        cloud_golden()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "sss"):
        sss_golden()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "sssmesh"):
        sssmesh_snapshot()
    elif (len(sys.argv) > 1) and (sys.argv[1] == "features"):
        features_golden()
    else:
        main()
