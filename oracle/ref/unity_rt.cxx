// ORACLE BUILD RECIPE - reference etx-rt integrator TUs by path (sources/etx/CMakeLists.txt: create_library(rt)),
// minus rt.cxx (Embree; restated in oracle/shims/raytracing_bvh.cxx) and debug.cxx (not Monte-Carlo transport).
#include <atomic>
#include <map>
#include <mutex>
#include <bluenoise.hxx>  // the pack build gets it through debug.cxx, which precedes path_tracing.cxx
#include <etx/core/core.hxx>
#include <etx/render/host/film.hxx>
#include <etx/rt/integrators/integrator.cxx>
#include <etx/rt/integrators/path_tracing.cxx>
#include <etx/rt/integrators/vcm_cpu.cxx>
#include <etx/rt/integrators/vcm_shared.cxx>
#include <etx/rt/integrators/bidirectional.cxx>
