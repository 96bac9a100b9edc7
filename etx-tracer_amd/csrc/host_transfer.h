// host_transfer.h - every byte that crosses PCIe travels through PINNED memory the library owns.
//
// The C ABI hands over plain host pointers (scene tables of the host application, film destinations, checkpoints: include/etx_hip.h). Passing such
// a pointer to hipMemcpy* makes the HIP runtime pin the CALLER's pages itself for every transfer above 4 KB (ROCclr KernelBlitManager::readBuffer /
// writeBuffer: pinHostMemory + a blit kernel) and keep up to eight of those pins cached per stream, keyed by address and size, "for a later
// release". The caller's memory is the caller's: a numpy film buffer or a scene snapshot is freed and its address range reused (by the next film
// buffer, by an allocator arena) while a cached pin of the old range still exists. Round 5's GPU suite died of exactly that family of defects
// (the interpreter's heap was overwritten after two contexts had been rendered, read back and destroyed; only with the ROCm 7.0.2
// runtime of the torch wheel in the process). The library therefore never shows HIP a host pointer it did not allocate with hipHostMalloc:
//   to_device: memcpy into a pinned slot, hipMemcpyAsync from the slot;   to_host: hipMemcpyAsync into a pinned slot, memcpy out of it.
// Two slots, so the CPU copy of one chunk overlaps the DMA of the other. Both calls are SYNCHRONOUS for the caller (on return the source may be
// freed / the destination holds the data) and ordered on `stream` like the hipMemcpyAsync they replace. Not on the hot path: scene upload, film
// read-back, checkpoints, test entry points.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>
#include <string>

namespace etxh {

class HostTransfer {
 public:
  HostTransfer() = default;
  HostTransfer(const HostTransfer&) = delete;
  HostTransfer& operator=(const HostTransfer&) = delete;
  ~HostTransfer();

  // 0 or ETX_HIP_ERROR_HIP with `error` set. stream == nullptr: the transfer's own stream.
  int to_device(void* dst_device, const void* src_host, size_t bytes, hipStream_t stream, std::string& error);
  int to_host(void* dst_host, const void* src_device, size_t bytes, hipStream_t stream, std::string& error);
  void release();  // frees the pinned slots (the device must be current); the object can be used again afterwards

 private:
  int prepare(std::string& error);
  static constexpr size_t kSlotBytes = size_t(4) << 20;
  std::mutex mutex_;
  unsigned char* slots_[2] = {nullptr, nullptr};
  hipEvent_t slot_free_[2] = {nullptr, nullptr};  // recorded behind the newest DMA that reads / writes the slot
  hipStream_t own_stream_ = nullptr;
};

}  // namespace etxh
