// ORACLE / TEST INFRASTRUCTURE ONLY - never linked into libetx_hip.so.
//
// No-op etx::Denoiser. The reference wraps OpenImageDenoise here (sources/etx/render/host/denoiser.cxx:60-143);
// OIDN headers are not available in this image and denoising is a post-process outside the Monte-Carlo loop
// (out of scope). Film only needs the symbol to exist (film.cxx allocate/denoise).
#include <etx/render/host/denoiser.hxx>

namespace etx {

struct DenoiserImpl {
  uint32_t unused = 0;
};

Denoiser::Denoiser() {
  ETX_PIMPL_INIT(Denoiser);
}

Denoiser::~Denoiser() {
  ETX_PIMPL_CLEANUP(Denoiser);
}

void Denoiser::init() {
}

void Denoiser::shutdown() {
}

void Denoiser::allocate_buffers(float3*, float3*, const uint2&) {
}

void Denoiser::denoise(float4*, float3*) {
}

}  // namespace etx
