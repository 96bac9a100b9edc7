// BUILD RECIPE of etx_hip_render (oracle/build_ref.sh): the part of the reference's etx-rt library the HIP binding needs on the host side -
// the Integrator base class and the VCM option block - WITHOUT its CPU integrators (path_tracing.cxx, vcm_cpu.cxx, bidirectional.cxx).
#include <atomic>
#include <map>
#include <mutex>
#include <bluenoise.hxx>
#include <etx/core/core.hxx>
#include <etx/render/host/film.hxx>
#include <etx/rt/integrators/integrator.cxx>
#include <etx/rt/integrators/vcm_shared.cxx>
