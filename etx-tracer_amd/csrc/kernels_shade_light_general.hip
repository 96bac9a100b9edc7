// k_light_shade of the general and subsurface shading groups: vcm_light_step over all BSDF classes
// (vcm_shared.hxx:1090-1260). Own translation unit: see kernels_shade.inl.
#include "kernels_shade.inl"
namespace etxd {
void launch_light_shade_group(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid, uint32_t group) {
  if (group == kShadeGroupGeneral)
    hipLaunchKernelGGL((k_light_shade<kShadeGroupGeneral, false>), grid, dim3(kBlockSize), 0, stream, p, it, in_set);
  else
    hipLaunchKernelGGL((k_light_shade<kShadeGroupSubsurface, false>), grid, dim3(kBlockSize), 0, stream, p, it, in_set);
}
}  // namespace etxd
