// ORACLE BUILD RECIPE - reference etx-render host TUs by path (sources/etx/CMakeLists.txt: create_library(render)),
// minus denoiser.cxx (OIDN; stub in oracle/shims) and tasks.cxx (enkiTS pimpl too small on libstdc++; oracle/shims).
#include <atomic>
#include <map>
#include <etx/render/host/film.cxx>
#include <etx/render/host/gltf_accessor.cxx>
#include <etx/render/host/image_pool.cxx>
#include <etx/render/host/medium_pool.cxx>
#include <etx/render/host/scattering.cxx>
#include <etx/render/host/scene_representation.cxx>
#include <etx/render/host/spectrum.cxx>
