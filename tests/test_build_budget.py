"""CPU: register budget of the gfx950 kernels, read from the assembly hipcc emits (tools/kernel_resources.py).

Round 1 shipped general-material kernels at the register allocator's limit (512 VGPRs, 700+ spilled VGPRs, 2 600+ spilled SGPRs,
33 min of compile time) whose results depended on the build. The guard: the kernels of the bench path (simple shading group, traversal,
pair connections, merge) stay within 256 registers with no scratch at all; the general-material kernels keep their architectural
VGPRs at 256, spill to AGPRs and at most a handful of VGPRs to scratch (today: 0 in k_merge_eval_generic, 17 in the subsurface camera kernel - its Isect carries the geometric normal since round 4 -, 0 elsewhere); a translation unit compiles in minutes, not tens of minutes."""
import concurrent.futures
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import kernel_resources  # noqa: E402


def test_kernel_register_budget():
    sources = ["kernels_vcm.hip", "kernels_connect.hip", "kernels_trace.hip", "kernels_shade_camera_general.hip", "kernels_bdpt.hip"]
    t0 = time.time()
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        rows = [r for rs in pool.map(kernel_resources.analyse, sources) for r in rs]
    elapsed = time.time() - t0
    kernels = {r["name"]: r for r in rows if r["kernel"]}
    assert len(kernels) > 30
    for name in ("void etxd::k_light_shade<0u, false>", "void etxd::k_camera_shade<0u, false>", "void etxd::k_connect_pairs<true>", "etxd::k_merge_diffuse", "void etxd::k_expand_pairs<true>",
                 "void etxd::k_trace_closest<true, true, false>", "void etxd::k_trace_closest<true, true, true>", "void etxd::k_trace_shadow<true, false, false>", "void etxd::k_trace_closest_bvh<true, 16u, 64u, false>", "void etxd::k_trace_closest_bvh<true, 16u, 64u, true>"):
        k = kernels[name]
        assert k["total_vgprs"] <= 256 and k.get("agprs", 0) == 0 and k["vgpr_spills"] == 0, (name, k)
    assert kernels["void etxd::k_camera_shade<0u, false>"]["total_vgprs"] <= 224  # two waves per SIMD (<= 256) with room; 213 today
    # shadow segments of tree scenes without Class::Boundary materials and density grids: one any-hit traversal and one exp - compiled for
    # seven wavefronts per SIMD (a handful of spilled registers) where the general kernel has three
    for name in ("void etxd::k_trace_shadow<false, false, true>", "void etxd::k_trace_shadow<false, true, true>"):
        k = kernels[name]
        assert k["total_vgprs"] <= 72 and k["scratch"] <= 32 and k["vgpr_spills"] <= 6, (name, k)
    # the phased tree kernel (round 5): node phase / leaf phase with one postponed leaf per lane, 68-70 registers
    assert kernels["void etxd::k_trace_closest_bvh<true, 32u, 64u, false>"]["total_vgprs"] <= 80
    # bidirectional (round 3): the walk-event kernels carry no BSDF code and fit four wavefronts per SIMD; the inline-BSDF instantiations
    # need no AGPRs and (almost) no scratch, three wavefronts per SIMD
    for name in ("etxd::k_bdpt_walk_light", "etxd::k_bdpt_walk_camera"):
        k = kernels[name]
        assert k["total_vgprs"] <= 128 and k["scratch"] == 0 and k["vgpr_spills"] == 0, (name, k)
    # (<true, 0u>: scenes of simple classes, every item; <true, 1u>: the inline part of a mixed scene, which hands the items of general classes to a list - round 6)
    for name in ("void etxd::k_bdpt_light_shade<true, 0u>", "void etxd::k_bdpt_camera_shade<true, 0u>", "void etxd::k_bdpt_light_shade<true, 1u>", "void etxd::k_bdpt_camera_shade<true, 1u>",
                 "void etxd::k_bdpt_connect_pairs<false>", "void etxd::k_bdpt_connect_light<true, 0u>", "void etxd::k_bdpt_connect_camera<true, 0u>",
                 "void etxd::k_bdpt_connect_light<true, 1u>", "void etxd::k_bdpt_connect_camera<true, 1u>", "void etxd::k_bdpt_walk_exit_light<true>", "void etxd::k_bdpt_walk_exit_camera<true>"):
        k = kernels[name]
        assert k["total_vgprs"] <= 184 and k.get("agprs", 0) == 0 and k["vgpr_spills"] == 0 and k["scratch"] <= 32, (name, k)
    general = [k for name, k in kernels.items() if ("<1u" in name) or ("<2u" in name) or name.endswith("k_merge_eval_generic") or ("k_connect_endpoints" in name) or
               name.endswith("etxd::k_connect_pairs<false>") or name.endswith("k_bdpt_connect_pairs<true>")]  # (k_bdpt_connect_pairs<kGeneral>: <true> = the dense list of general pairs, round 6)
    k = kernels["etxd::k_bdpt_expand_pairs"]
    assert k["total_vgprs"] <= 48 and k["scratch"] == 0, k
    assert len(general) >= 6
    # round 6: the filter half of the generic merge carries no BSDF code (it runs at full occupancy); the matrix-core sweep keeps its results in VGPRs
    k = kernels["etxd::k_merge_filter_generic"]
    assert k["total_vgprs"] <= 48 and k["scratch"] == 0 and k["vgpr_spills"] == 0, k
    k = kernels["void etxd::k_trace_closest_mfma<true, false>"]
    assert k["total_vgprs"] <= 128 and k.get("agprs", 0) == 0 and k["scratch"] == 0, k
    for k in general:
        assert k["vgprs"] <= 256 and k["vgpr_spills"] <= 24, (k["name"], k["vgprs"], k["vgpr_spills"])
    assert elapsed < 600.0, "the five translation units took %.0f s to compile" % elapsed
