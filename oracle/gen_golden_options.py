#!/usr/bin/env python3
"""Reference films for every NON-DEFAULT option branch of CPUVCM and CPUPathTracing, from the
reference-based oracle (oracle/_ref/etx_oracle), for tests/test_gpu_options.py.

    python3 oracle/gen_golden_options.py [--spp 1024] [--cores 0-5] [--only vcm|pt] [set ...]

Option keys: VCMOptions::load (sources/etx/rt/integrators/vcm_shared.cxx:15-29) and CPUPathTracingImpl::start
(sources/etx/rt/integrators/path_tracing.cxx:36-40). Every key is flipped away from its default in at least one set:

  vcm  nomis      vcm-mis=false
       tophat     vcm-kernel=0                                     (VCMOptions::TopHat; the default is Epanechnikov)
       connonly   vcm-merging=false                                (vm_weight = 0, vcm_cpu.cxx:110)
       mergeonly  vcm-connect_vertices=false vcm-connect_to_light=false
       nodirect   vcm-direct_hit=false vcm-connect_to_camera=false (the light image stays empty)
       radius     vcm-initial_radius=0.05 vcm-radius_decay=16
       nomergev   vcm-merge_vertices=false                         (weights still count the merges that are not done)
  pt   nonee      nee=false
       nomis      mis=false
       nodirect   direct=false
  bdpt nomis      bdpt-conn_mis=false                               (all: bdpt-mode=3, BDPTFull)
       nodirect   bdpt-conn_direct_hit=false bdpt-conn_connect_to_camera=false
       noconnect  bdpt-conn_connect_to_light=false bdpt-conn_connect_vertices=false

Films (tests/golden/opt/, 128 x 128, blue noise off, every iteration of `--spp`):
  cornell_<classic|full>_128_vcm_<spp>_<set>_rekeyed.npz       ETX_ORACLE_DECORRELATE=2: independent light / camera streams = the device's default estimator
  cornell_<classic|full>_128_vcm_<spp>_<set>_opaque_none.npz   ETX_ORACLE_BVH_DRAWS=opaque_none: the UNMODIFIED integrator (shared seeds), pinned
       (its film no longer depends on the traversal order; the device matches it with hip-reference_seeding = true)
  cornell_<classic|full>_128_pt_<spp>_<set>.npz       CPUPathTracing, --noise-threshold 0
  cornell_<classic|full>_128_bdpt3_<spp>_<set>_opaque_none.npz   CPUBidirectional (BDPTFull), pinned like the VCM films

~2 min per VCM film on 8 cores at 1024 spp.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden_hi  # noqa: E402  (render(): one oracle run -> one .npz)

OPT = os.path.join(gen_golden_hi.GOLDEN, "opt")

VCM_SETS = {
    "nomis": {"vcm-mis": "false"},
    "tophat": {"vcm-kernel": "0"},
    "connonly": {"vcm-merging": "false"},
    "mergeonly": {"vcm-connect_vertices": "false", "vcm-connect_to_light": "false"},
    "nodirect": {"vcm-direct_hit": "false", "vcm-connect_to_camera": "false"},
    "radius": {"vcm-initial_radius": "0.05", "vcm-radius_decay": "16"},
    "nomergev": {"vcm-merge_vertices": "false"},
}
PT_SETS = {
    "nonee": {"nee": "false"},
    "nomis": {"mis": "false"},
    "nodirect": {"direct": "false"},
}
BDPT_SETS = {  # CPUBidirectionalImpl::start, bidirectional.cxx:1469-1478; mode BDPTFull
    "nomis": {"bdpt-conn_mis": "false"},
    "nodirect": {"bdpt-conn_direct_hit": "false", "bdpt-conn_connect_to_camera": "false"},
    "noconnect": {"bdpt-conn_connect_to_light": "false", "bdpt-conn_connect_vertices": "false"},
}


def opt_args(options):
    out = []
    for key, value in options.items():
        out += ["--opt", "%s=%s" % (key, value)]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--cores", default="")
    ap.add_argument("--only", default="vcm,pt,bdpt")
    ap.add_argument("sets", nargs="*")
    args = ap.parse_args()
    gen_golden_hi.HI = OPT  # render() creates its output directory from this
    os.makedirs(OPT, exist_ok=True)
    only = args.only.split(",")
    for flavour in ("full", "classic"):
        snapshot = os.path.join(gen_golden_hi.GOLDEN, "cornell_%s_128.etxscene" % flavour)
        if "pt" in only:
            for name, options in PT_SETS.items():
                if args.sets and (name not in args.sets):
                    continue
                gen_golden_hi.render(snapshot, "pt", args.spp, os.path.join(OPT, "cornell_%s_128_pt_%d_%s.npz" % (flavour, args.spp, name)), args.cores,
                                     extra=["--opt", "bn=false", "--noise-threshold", "0"] + opt_args(options))
        if "bdpt" in only:
            for name, options in BDPT_SETS.items():
                if args.sets and (name not in args.sets):
                    continue
                gen_golden_hi.render(snapshot, "bdpt", args.spp, os.path.join(OPT, "cornell_%s_128_bdpt3_%d_%s_opaque_none.npz" % (flavour, args.spp, name)), args.cores,
                                     env_extra={"ETX_ORACLE_BVH_DRAWS": "opaque_none"}, extra=["--opt", "bdpt-blue_noise=false", "--opt", "bdpt-mode=3"] + opt_args(options))
        if "vcm" in only:
            for name, options in VCM_SETS.items():
                if args.sets and (name not in args.sets):
                    continue
                extra = ["--opt", "vcm-blue_noise=false"] + opt_args(options)
                gen_golden_hi.render(snapshot, "vcm", args.spp, os.path.join(OPT, "cornell_%s_128_vcm_%d_%s_opaque_none.npz" % (flavour, args.spp, name)), args.cores,
                                     env_extra={"ETX_ORACLE_BVH_DRAWS": "opaque_none"}, extra=extra)
                if True:  # both boxes: on the classic box, too, the pinned film (streams aligned draw for draw) differs from independent streams by up to 0.7 % of the mean
                    gen_golden_hi.render(snapshot, "vcm", args.spp, os.path.join(OPT, "cornell_%s_128_vcm_%d_%s_rekeyed.npz" % (flavour, args.spp, name)), args.cores,
                                         env_extra={"ETX_ORACLE_DECORRELATE": "2"}, extra=extra)


if __name__ == "__main__":
    main()
