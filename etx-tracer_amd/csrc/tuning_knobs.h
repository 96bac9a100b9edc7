// tuning_knobs.h - experiment knobs of the launch code.
// The product library runs with the defaults compiled in: `tuning_knob` returns `fallback`. A build with -DETX_HIP_DEBUG
// (ETX_HIP_EXTRA_FLAGS=-DETX_HIP_DEBUG etx-tracer_amd/csrc/build.sh, tools/build_variant.sh) reads the ETX_HIP_* environment variable of
// that name instead - what the A/B and cost-attribution tools (tools/ab_bench.sh, tools/cost_probe.sh, tools/option_cost.py) switch.
// Documented run-time configuration is NOT a knob and is always read: ETX_HIP_LANES,
// ETX_HIP_BVH_BUILD_THREADS, ETX_HIP_VERBOSE (include/etx_hip.h).
#pragma once

#include <cstdint>
#include <cstdlib>

namespace etxh {

inline bool tuning_knob_present(const char* name) {
#if defined(ETX_HIP_DEBUG)
  return getenv(name) != nullptr;
#else
  (void)name;
  return false;
#endif
}

inline uint32_t tuning_knob(const char* name, uint32_t fallback) {
#if defined(ETX_HIP_DEBUG)
  if (const char* e = getenv(name))
    return uint32_t(strtoul(e, nullptr, 0));
#else
  (void)name;
#endif
  return fallback;
}

inline float tuning_knob_f(const char* name, float fallback) {
#if defined(ETX_HIP_DEBUG)
  if (const char* e = getenv(name))
    return float(atof(e));
#else
  (void)name;
#endif
  return fallback;
}

}  // namespace etxh
