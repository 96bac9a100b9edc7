// k_light_shade<false>: vcm_light_step over all BSDF classes (vcm_shared.hxx:1090-1260). Own translation unit: see kernels_shade.inl.
#include "kernels_shade.inl"
namespace etxd {
void launch_light_shade_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid) {
  hipLaunchKernelGGL(k_light_shade<false>, grid, dim3(kBlockSize), 0, stream, p, it, in_set);
}
}  // namespace etxd
