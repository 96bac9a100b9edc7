// host_reduce.h - state of the multi-GPU film reduce, shared by host_api.cpp (the context, the lanes' commits) and host_comm.cpp (RCCL).
//
// north_star: "a single RCCL reduce of the float framebuffer over xGMI at the end of each iteration"; SURVEY.md 8e row 2: "after each round
// (or every k rounds for display)". The reference consumes its film progressively - Film::commit_light_iteration once per iteration
// (sources/etx/render/host/film.cxx:332-343), the GUI pump reads it every frame (sources/raytracer/app.cxx:150-155) - so the reduce must not end
// the render. Shape:
//   * a communication stream of its own; the lanes keep rendering while a reduce runs, and etx_hip_reduce_film_begin never waits - not for the lanes and not
//     for an earlier reduce (the first version finished the previous reduce before it began the next: a snapshot waits for the commits of the iterations in
//     flight, ~one iteration's latency, so a reduce per iteration stalled the thread that hands out iterations - 93.8 instead of 100.8 Msamples/s on one GPU)
//   * SNAPSHOT: one kernel copies the layers the armed integrator writes (VCM camera + light, 2 x 16 B per pixel = 66 MB at 1080p; path tracer
//     camera + normal + albedo; bidirectional all four) from the film sums into `snapshot`. Commits and snapshots exclude each other through
//     events (a lane's commit kernel waits for the newest snapshot, a snapshot waits for every lane's newest commit; both are enqueued under
//     `mutex`, so the order is well defined and neither side ever blocks the host): a snapshot holds WHOLE iterations
//   * OUT OF PLACE: ncclAllReduce(snapshot -> reduced, sum). The film sums are never touched, so rendering continues after a reduce and a
//     checkpoint taken later still holds this rank's own iterations
//   * the per-pixel sample counts ride in the camera layer's w (k_vcm_commit / k_pt_commit count there): the reduced image is normalised pixel by
//     pixel by the reduced count, whatever each rank had finished when its snapshot was taken
//   * etx_hip_read_film* on a context with a communicator returns the reduced copy of the most recent finished reduce.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>

struct EtxReduceState {
  std::mutex mutex;                    // orders "enqueue a commit" (lane threads) against "enqueue a snapshot" (the caller's thread)
  hipStream_t stream = nullptr;        // snapshot, collectives, counter copies, reads of the reduced copy
  hipEvent_t snapshot_done = nullptr;  // recorded behind the snapshot kernel: later commits wait for it (device side)
  hipEvent_t done = nullptr;           // recorded behind the collectives and the counter read-back of the newest reduce
  hipEvent_t time_begin = nullptr, time_end = nullptr;  // device time of the newest reduce (snapshot + collectives)
  bool snapshot_recorded = false;
  float4* snapshot = nullptr;          // kFilmLayers x pixels: send buffer
  float4* reduced = nullptr;           // kFilmLayers x pixels: whole-job sums as of the newest reduce
  size_t pixels = 0;
  unsigned long long* d_counters = nullptr;  // device: [0..1] send {iterations counted by this rank, 1 if this rank failed} (written by a kernel: no host staging a later
                                             // reduce could overwrite), [2..3] receive
  unsigned long long* h_counters = nullptr;  // pinned: [2..3] the received sums of the newest reduce
  double* h_words = nullptr;           // pinned [32] / device [32]: etx_hip_comm_all_reduce_f64 (send [0..16), receive [16..32))
  double* d_words = nullptr;
  uint32_t pending = 0;                // reduces enqueued and not yet collected by etx_hip_reduce_film_end. Any number may be in flight: the communication stream
                                       // orders them (snapshot N+1 overwrites the send buffer behind all-reduce N), _begin never waits
  bool valid = false;                  // `reduced` holds a finished reduce of the current run (etx_hip_begin invalidates)
  int pending_local_rc = 0;            // this rank's own failure at the time of _begin: reported by _end, after the collective
  std::string pending_local_error;
  uint32_t layer_mask = 0;             // layers of the newest reduce (bit = layer index in the film allocation)
  uint64_t reduces = 0;                // finished reduces since etx_hip_comm_init
  uint64_t payload_bytes = 0;          // film bytes of one reduce (per rank, what the all-reduce sums)
  double last_device_ms = 0.0, total_device_ms = 0.0;
};
