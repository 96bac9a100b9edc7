// ORACLE BUILD RECIPE - pulls reference translation units in by path (nothing is copied into this repo).
// Mirrors the reference's own "pack" build (sources/etx/CMakeLists.txt:1-30) for etx-core + etx-util,
// minus core/platform.cxx (no Linux branch; replaced by oracle/shims/platform_linux.cxx).
#include <atomic>
#include <map>
#include <etx/core/core.cxx>
#include <etx/core/log.cxx>
#include <etx/core/environment.cxx>
#include <etx/core/profiler.cxx>
#include <etx/util/options.cxx>
