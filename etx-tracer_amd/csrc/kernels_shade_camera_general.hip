// k_camera_shade<false>: vcm_camera_step over all BSDF classes (vcm_shared.hxx:927-1079). Own translation unit: see kernels_shade.inl.
#include "kernels_shade.inl"
namespace etxd {
void launch_camera_shade_general(hipStream_t stream, const Pipeline& p, const VcmParams& it, uint32_t in_set, const dim3& grid) {
  hipLaunchKernelGGL(k_camera_shade<false>, grid, dim3(kBlockSize), 0, stream, p, it, in_set);
}
}  // namespace etxd
