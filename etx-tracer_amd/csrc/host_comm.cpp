// host_comm.cpp - multi-GPU film reduction over RCCL (xGMI).
// The reference has no communication layer at all; iterations (or pixels, etx_hip_begin_ex) are sharded over ranks and the
// ONLY exchange is a sum-reduce of the float4 film layers the armed integrator writes plus two counter words. north_star asks for
// it "at the end of each iteration": the reduce here is asynchronous, out of place and NOT terminal - it runs on a communication stream of its
// own from a snapshot of the film while the lanes keep rendering, and rendering continues afterwards (host_reduce.h has the whole scheme; the
// reference's film is consumed progressively too: Film::commit_light_iteration per iteration, film.cxx:332-343; GUI pump app.cxx:150-155).
// Payload at 1080p: VCM 2 x 33 MB fp32 per reduce - one ring all-reduce over xGMI, per-link bound (DESIGN.md 6 has the model).
#include "../../include/etx_hip.h"

#include "host_reduce.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

static_assert(sizeof(ncclUniqueId) == ETX_HIP_UNIQUE_ID_BYTES, "ncclUniqueId size");

void** etx_hip_internal_comm(etx_hip_context* c);
EtxReduceState* etx_hip_internal_reduce(etx_hip_context* c);
void etx_hip_internal_set_error(etx_hip_context* c, const std::string& e);
void etx_hip_internal_rank(etx_hip_context* c, int** rank, int** world);
uint64_t* etx_hip_internal_global_iterations(etx_hip_context* c);
int etx_hip_internal_device(etx_hip_context* c);
int etx_hip_internal_reduce_allocate(etx_hip_context* c);
void etx_hip_internal_reduce_release(etx_hip_context* c);
int etx_hip_internal_reduce_prepare(etx_hip_context* c, int local_rc, float4** out_snapshot, float4** out_reduced, size_t* out_pixels, uint32_t* out_layer_mask);
int etx_hip_internal_reduce_prepare_failed(etx_hip_context* c, float4** out_snapshot, float4** out_reduced, size_t* out_pixels, uint32_t* out_layer_mask);
int etx_hip_internal_reduce_finish(etx_hip_context* c);

namespace {

// One reduce, enqueued: snapshot, the collectives, the counter read-back. `local_rc`: this rank's own failure so far (travels INTO the collective
// as its failed flag; reported by reduce_end after the collective, so no rank ever stays out of an all-reduce the others are already in).
int reduce_begin(etx_hip_context* context, int local_rc, std::string local_error) {
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(*etx_hip_internal_comm(context));
  EtxReduceState* r = etx_hip_internal_reduce(context);
  if (comm == nullptr)
    return local_rc;  // single rank: the film IS the job's film, nothing to exchange
  float4 *snapshot = nullptr, *reduced = nullptr;
  size_t pixels = 0;
  uint32_t layer_mask = 0;
  if (int rc = etx_hip_internal_reduce_prepare(context, local_rc, &snapshot, &reduced, &pixels, &layer_mask)) {
    // the other ranks are entering the collective: this rank joins it with a zero snapshot and its failed flag, and reports its error from
    // etx_hip_reduce_film_end like any other local failure - unless it has nothing to join with (no scene, no buffers: etx_hip.h)
    const std::string why = etx_hip_last_error(context);
    if (etx_hip_internal_reduce_prepare_failed(context, &snapshot, &reduced, &pixels, &layer_mask) != ETX_HIP_OK) {
      etx_hip_internal_set_error(context, why);
      return rc;
    }
    if (local_rc == 0)
      local_rc = rc, local_error = why;
  }
  // contiguous runs of layers become one all-reduce each (VCM: [camera, light]; path tracer: [camera], [normal, albedo]; bidirectional: all four)
  ncclResult_t res = ncclGroupStart();
  for (uint32_t layer = 0; (res == ncclSuccess) && (layer < 4u);) {
    if (((layer_mask >> layer) & 1u) == 0u) {
      ++layer;
      continue;
    }
    uint32_t end = layer;
    while ((end < 4u) && ((layer_mask >> end) & 1u))
      ++end;
    res = ncclAllReduce(snapshot + size_t(layer) * pixels, reduced + size_t(layer) * pixels, size_t(end - layer) * pixels * 4u, ncclFloat, ncclSum, comm, r->stream);
    layer = end;
  }
  if (res == ncclSuccess)
    res = ncclAllReduce(r->d_counters, r->d_counters + 2, 2, ncclUint64, ncclSum, comm, r->stream);
  const ncclResult_t res_end = ncclGroupEnd();
  if (res == ncclSuccess)
    res = res_end;
  if (res != ncclSuccess) {
    etx_hip_internal_set_error(context, std::string("ncclAllReduce: ") + ncclGetErrorString(res));
    return ETX_HIP_ERROR_COMM;
  }
  if (int rc = etx_hip_internal_reduce_finish(context))
    return rc;
  r->pending += 1u;
  if (local_rc && (r->pending_local_rc == 0)) {
    r->pending_local_rc = local_rc;
    r->pending_local_error = local_error;
  }
  return ETX_HIP_OK;
}

}  // namespace

extern "C" {

void etx_hip_internal_rccl_versions(int* mapped, int* built) {
  *built = NCCL_VERSION_CODE;
  *mapped = 0;
  (void)ncclGetVersion(mapped);
}

void etx_hip_comm_destroy_internal(etx_hip_context* context) {
  EtxReduceState* r = etx_hip_internal_reduce(context);
  if (r->stream)
    (void)hipStreamSynchronize(r->stream);  // a reduce still in flight
  void** comm = etx_hip_internal_comm(context);
  if (*comm) {
    (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(*comm));
    *comm = nullptr;
  }
  etx_hip_internal_reduce_release(context);
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
  // everything the reduce needs is allocated here (the film-sized buffers as soon as a scene is uploaded: here, or at the next etx_hip_begin):
  // a rank must not find out inside a reduce that it cannot join
  if (int rc = etx_hip_internal_reduce_allocate(context)) {
    (void)ncclCommDestroy(comm);
    etx_hip_internal_reduce_release(context);
    return rc;
  }
  *etx_hip_internal_comm(context) = comm;
  EtxReduceState* state = etx_hip_internal_reduce(context);
  state->reduces = 0, state->last_device_ms = state->total_device_ms = 0.0;
  int *prank = nullptr, *pworld = nullptr;
  etx_hip_internal_rank(context, &prank, &pworld);
  *prank = rank;
  *pworld = world_size;
  return ETX_HIP_OK;
}

int etx_hip_reduce_film_begin(etx_hip_context* context) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  return reduce_begin(context, 0, std::string());
}

int etx_hip_reduce_film_end(etx_hip_context* context, int wait) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  EtxReduceState* r = etx_hip_internal_reduce(context);
  if (r->pending == 0u)
    return 1;  // nothing in flight (also: no communicator)
  if (hipSetDevice(etx_hip_internal_device(context)) != hipSuccess) {
    etx_hip_internal_set_error(context, "hipSetDevice failed");
    return ETX_HIP_ERROR_HIP;
  }
  if (wait) {
    if (hipEventSynchronize(r->done) != hipSuccess) {
      etx_hip_internal_set_error(context, "event synchronize failed after the film all-reduce");
      return ETX_HIP_ERROR_HIP;
    }
  } else {
    const hipError_t q = hipEventQuery(r->done);
    if (q == hipErrorNotReady)
      return 0;
    if (q != hipSuccess) {
      etx_hip_internal_set_error(context, std::string("film all-reduce: ") + hipGetErrorString(q));
      return ETX_HIP_ERROR_HIP;
    }
  }
  // `done` is the NEWEST reduce's: behind it on the communication stream every earlier one has finished too
  const uint32_t finished = r->pending;
  r->pending = 0u;
  r->valid = true;
  float ms = 0.0f;
  if (hipEventElapsedTime(&ms, r->time_begin, r->time_end) == hipSuccess) {  // of the newest one (the events are re-recorded by every reduce)
    r->last_device_ms = double(ms);
    r->total_device_ms += double(ms) * double(finished);
  }
  r->reduces += finished;
  *etx_hip_internal_global_iterations(context) = r->h_counters[2];
  if (r->pending_local_rc) {
    const int rc = r->pending_local_rc;
    etx_hip_internal_set_error(context, r->pending_local_error);
    r->pending_local_rc = 0, r->pending_local_error.clear();
    return rc;
  }
  if (r->h_counters[3] != 0ull) {
    etx_hip_internal_set_error(context, std::to_string(r->h_counters[3]) + " rank(s) reported a failed iteration before the film reduce (their films are incomplete)");
    return ETX_HIP_ERROR_COMM;
  }
  return 1;
}

int etx_hip_reduce_film(etx_hip_context* context) {
  if (context == nullptr)
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  // Every iteration handed to a lane has reached the film. A failed iteration on THIS rank (for example ETX_HIP_ERROR_OVERFLOW beyond the
  // pool limit) must not keep the rank out of the collective: the other ranks are already inside ncclAllReduce and would wait for the RCCL
  // timeout. Every rank therefore always takes part and the error travels with the counters; afterwards ALL ranks return an error.
  const int sync_rc = etx_hip_sync(context);
  const std::string local_error = sync_rc ? std::string(etx_hip_last_error(context)) : std::string();
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(*etx_hip_internal_comm(context));
  if (comm == nullptr)
    return sync_rc;  // single rank: the reduce is the identity
  const int begun = reduce_begin(context, sync_rc, local_error);
  if (begun != ETX_HIP_OK)
    return begun;
  const int ended = etx_hip_reduce_film_end(context, 1);
  return (ended < 0) ? ended : ETX_HIP_OK;
}

// A handful of doubles all-reduced over the communicator (bench.py's barrier and max-over-ranks): pinned words -> device -> ncclAllReduce -> pinned, on
// the communication stream, behind whatever reduces are in flight there.
int etx_hip_comm_all_reduce_f64(etx_hip_context* context, double* values, uint32_t count, int op) {
  if ((context == nullptr) || (values == nullptr) || (count == 0u) || (count > 16u) || (op < ETX_HIP_REDUCE_SUM) || (op > ETX_HIP_REDUCE_MIN))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(*etx_hip_internal_comm(context));
  if (comm == nullptr)
    return ETX_HIP_OK;  // one rank: the values are the job's
  EtxReduceState* r = etx_hip_internal_reduce(context);
  if (hipSetDevice(etx_hip_internal_device(context)) != hipSuccess) {
    etx_hip_internal_set_error(context, "hipSetDevice failed");
    return ETX_HIP_ERROR_HIP;
  }
  if (r->h_words == nullptr) {
    if ((hipHostMalloc(reinterpret_cast<void**>(&r->h_words), 32 * sizeof(double), hipHostMallocDefault) != hipSuccess) ||
        (hipMalloc(reinterpret_cast<void**>(&r->d_words), 32 * sizeof(double)) != hipSuccess)) {
      etx_hip_internal_set_error(context, "allocation of the collective's words failed");
      return ETX_HIP_ERROR_HIP;
    }
  }
  memcpy(r->h_words, values, count * sizeof(double));
  const ncclRedOp_t red = (op == ETX_HIP_REDUCE_SUM) ? ncclSum : ((op == ETX_HIP_REDUCE_MAX) ? ncclMax : ncclMin);
  if (hipMemcpyAsync(r->d_words, r->h_words, count * sizeof(double), hipMemcpyHostToDevice, r->stream) != hipSuccess) {
    etx_hip_internal_set_error(context, "hipMemcpyAsync failed (collective words)");
    return ETX_HIP_ERROR_HIP;
  }
  const ncclResult_t res = ncclAllReduce(r->d_words, r->d_words + 16, count, ncclDouble, red, comm, r->stream);
  if (res != ncclSuccess) {
    etx_hip_internal_set_error(context, std::string("ncclAllReduce: ") + ncclGetErrorString(res));
    return ETX_HIP_ERROR_COMM;
  }
  if ((hipMemcpyAsync(r->h_words + 16, r->d_words + 16, count * sizeof(double), hipMemcpyDeviceToHost, r->stream) != hipSuccess) || (hipStreamSynchronize(r->stream) != hipSuccess)) {
    etx_hip_internal_set_error(context, std::string("all-reduce of the collective's words failed: ") + hipGetErrorString(hipGetLastError()));
    return ETX_HIP_ERROR_HIP;
  }
  memcpy(values, r->h_words + 16, count * sizeof(double));
  return ETX_HIP_OK;
}

int etx_hip_comm_barrier(etx_hip_context* context) {
  double word = 1.0;
  return etx_hip_comm_all_reduce_f64(context, &word, 1u, ETX_HIP_REDUCE_SUM);
}

int etx_hip_reduce_info(etx_hip_context* context, etx_hip_reduce_info_t* out_info, size_t info_size) {
  if ((context == nullptr) || (out_info == nullptr) || (info_size != sizeof(etx_hip_reduce_info_t)))
    return ETX_HIP_ERROR_INVALID_ARGUMENT;
  const EtxReduceState* r = etx_hip_internal_reduce(context);
  etx_hip_reduce_info_t info = {};
  info.reduces = r->reduces;
  info.payload_bytes = r->payload_bytes;
  info.global_iterations = *etx_hip_internal_global_iterations(context);
  info.last_device_ms = r->last_device_ms;
  info.total_device_ms = r->total_device_ms;
  info.pending = r->pending;
  info.layer_mask = r->layer_mask;
  *out_info = info;
  return ETX_HIP_OK;
}

}  // extern "C"
