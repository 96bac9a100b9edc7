// k_path_tail<light, general materials>. Own translation unit: see kernels_shade.inl.
#include "kernels_shade.inl"
namespace etxd {
void launch_light_tail_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, uint32_t blocks) {
  hipLaunchKernelGGL((k_path_tail<false, false>), dim3(blocks), dim3(kBlockSize), 0, stream, p, it, in_set);
}
}  // namespace etxd
