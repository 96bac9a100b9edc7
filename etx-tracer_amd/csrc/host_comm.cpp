// host_comm.cpp - multi-GPU film reduction over RCCL (xGMI).
// The reference has no communication layer at all; iterations are sharded over ranks
// (etx_hip_begin first/stride) and the ONLY exchange is one sum-reduce of the two float4 film accumulators plus the
// iteration counter. Payload at 1080p: 2 x 33 MB fp32 - one ring all-reduce, per-link bound.
#include "../../include/etx_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

static_assert(sizeof(ncclUniqueId) == ETX_HIP_UNIQUE_ID_BYTES, "ncclUniqueId size");

hipStream_t etx_hip_internal_stream(etx_hip_context* c);
void** etx_hip_internal_comm(etx_hip_context* c);
void etx_hip_internal_film(etx_hip_context* c, float** camera, float** light, size_t* floats);
void etx_hip_internal_set_error(etx_hip_context* c, const std::string& e);
void etx_hip_internal_rank(etx_hip_context* c, int** rank, int** world);
void etx_hip_internal_iterations(etx_hip_context* c, uint32_t** local, uint64_t** global, bool** reduced);
uint32_t etx_hip_internal_counted_iterations(etx_hip_context* c);
int etx_hip_internal_device(etx_hip_context* c);
void** etx_hip_internal_comm_scratch(etx_hip_context* c);  // device words {iterations of this rank, 1 if this rank failed}, allocated with the communicator

extern "C" {

void etx_hip_comm_destroy_internal(etx_hip_context* context) {
  void** comm = etx_hip_internal_comm(context);
  if (*comm) {
    (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(*comm));
    *comm = nullptr;
  }
  void** scratch = etx_hip_internal_comm_scratch(context);
  if (*scratch) {
    (void)hipFree(*scratch);
    *scratch = nullptr;
  }
}

int etx_hip_comm_unique_id(void* out_id_128_bytes) {
  if (out_id_128_bytes == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess)
    return ETX_HIP_ERROR_COMM;
  memcpy(out_id_128_bytes, &id, sizeof(id));
  return ETX_HIP_OK;
}

int etx_hip_comm_init(etx_hip_context* context, int rank, int world_size, const void* id_128_bytes) {
  if ((context == nullptr) || (id_128_bytes == nullptr) || (world_size < 1) || (rank < 0) || (rank >= world_size))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  etx_hip_comm_destroy_internal(context);
  if (hipSetDevice(etx_hip_internal_device(context)) != hipSuccess) {
    etx_hip_internal_set_error(context, "hipSetDevice failed");
    return ETX_HIP_ERROR_HIP;
  }
  ncclUniqueId id;
  memcpy(&id, id_128_bytes, sizeof(id));
  ncclComm_t comm = nullptr;
  ncclResult_t r = ncclCommInitRank(&comm, world_size, id, rank);
  if (r != ncclSuccess) {
    etx_hip_internal_set_error(context, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    return ETX_HIP_ERROR_COMM;
  }
  // everything the reduce needs is allocated here: a rank must not find out inside etx_hip_reduce_film that it cannot join
  void* scratch = nullptr;
  if (hipMalloc(&scratch, 2 * sizeof(unsigned long long)) != hipSuccess) {
    (void)ncclCommDestroy(comm);
    etx_hip_internal_set_error(context, "hipMalloc failed (film reduce counters)");
    return ETX_HIP_ERROR_HIP;
  }
  *etx_hip_internal_comm(context) = comm;
  *etx_hip_internal_comm_scratch(context) = scratch;
  int *prank = nullptr, *pworld = nullptr;
  etx_hip_internal_rank(context, &prank, &pworld);
  *prank = rank;
  *pworld = world_size;
  return ETX_HIP_OK;
}

int etx_hip_reduce_film(etx_hip_context* context) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  // Every iteration handed to a lane has reached the film. A failed iteration on THIS rank (for example
  // ETX_HIP_ERROR_OVERFLOW, a data-dependent condition of the fixed pools) must not keep the rank out of the collective:
  // the other ranks are already inside ncclAllReduce and would wait for the RCCL timeout. Every rank therefore always
  // takes part and the error travels with the iteration counter; afterwards ALL ranks return an error.
  const int sync_rc = etx_hip_sync(context);
  uint32_t* local = nullptr;
  uint64_t* global = nullptr;
  bool* reduced = nullptr;
  etx_hip_internal_iterations(context, &local, &global, &reduced);
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(*etx_hip_internal_comm(context));
  if (comm == nullptr) {
    // single rank: the reduce is the identity
    if (sync_rc)
      return sync_rc;
    *global = *local;
    *reduced = true;
    return ETX_HIP_OK;
  }
  std::string local_error = sync_rc ? std::string(etx_hip_last_error(context)) : std::string();
  // No early return between here and the collective: a local failure (even of hipSetDevice) is carried INTO the all-reduce as this
  // rank's failed flag; whether the collective itself can run is for RCCL to say.
  int local_rc = sync_rc;
  if (hipSetDevice(etx_hip_internal_device(context)) != hipSuccess) {
    if (local_rc == 0)
      local_rc = ETX_HIP_ERROR_HIP, local_error = "hipSetDevice failed before the film reduce";
  }
  hipStream_t stream = etx_hip_internal_stream(context);
  float *camera = nullptr, *light = nullptr;
  size_t floats = 0;
  etx_hip_internal_film(context, &camera, &light, &floats);
  unsigned long long* d_counters = reinterpret_cast<unsigned long long*>(*etx_hip_internal_comm_scratch(context));  // allocated by etx_hip_comm_init
  unsigned long long h_counters[2] = {etx_hip_internal_counted_iterations(context), local_rc ? 1ull : 0ull};
  (void)hipMemcpyAsync(d_counters, h_counters, sizeof(h_counters), hipMemcpyHostToDevice, stream);
  ncclResult_t r = ncclGroupStart();
  if (r == ncclSuccess)
    r = ncclAllReduce(camera, camera, floats, ncclFloat, ncclSum, comm, stream);  // all film layers, one buffer
  if (r == ncclSuccess)
    r = ncclAllReduce(d_counters, d_counters, 2, ncclUint64, ncclSum, comm, stream);
  ncclResult_t r2 = ncclGroupEnd();
  if (r == ncclSuccess)
    r = r2;
  if (r != ncclSuccess) {
    etx_hip_internal_set_error(context, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
    return ETX_HIP_ERROR_COMM;
  }
  (void)hipMemcpyAsync(h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream);
  if (hipStreamSynchronize(stream) != hipSuccess) {
    etx_hip_internal_set_error(context, "stream synchronize failed after all-reduce");
    return ETX_HIP_ERROR_HIP;
  }
  *global = h_counters[0];
  *reduced = true;
  if (local_rc) {
    etx_hip_internal_set_error(context, local_error);
    return local_rc;
  }
  if (h_counters[1] != 0ull) {
    etx_hip_internal_set_error(context, std::to_string(h_counters[1]) + " other rank(s) reported a failed iteration before the film reduce (their films are incomplete)");
    return ETX_HIP_ERROR_COMM;
  }
  return ETX_HIP_OK;
}

}  // extern "C"
