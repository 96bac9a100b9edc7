// k_path_tail<camera> of the general and subsurface shading groups. Own translation unit: see kernels_shade.inl.
#include "kernels_shade.inl"
namespace etxd {
void launch_camera_tail_group(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t blocks, uint32_t group) {
  if (group == kShadeGroupGeneral)
    hipLaunchKernelGGL((k_path_tail<true, kShadeGroupGeneral>), dim3(blocks), dim3(kBlockSize), 0, stream, p, it, in_set);
  else
    hipLaunchKernelGGL((k_path_tail<true, kShadeGroupSubsurface>), dim3(blocks), dim3(kBlockSize), 0, stream, p, it, in_set);
}
}  // namespace etxd
