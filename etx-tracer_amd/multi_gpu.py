"""Multi-GPU: iterations are sharded over ranks, the only exchange is the sum-reduce of the film.

The reference has no communication layer. One VCM iteration is self-contained (its own light paths, photon grid,
radius and MIS weights depend only on the iteration index and the pixel count, vcm_cpu.cxx:100-113), so rank r of R
renders iterations r, r+R, r+2R, ... over the FULL frame with a scene replica, accumulates float4 SUMS (not the
reference's running mean, film.cxx:200-206, so that the reduce is associative) and the film all-reduce yields the
whole-job image; the host divides by the total iteration count.

Everything that crosses ranks goes through RCCL inside libetx_hip.so (etx_hip_comm_init, etx_hip_reduce_film*,
etx_hip_comm_all_reduce_f64 / _barrier). The one thing RCCL cannot do for itself is hand the 128-byte ncclUniqueId of
rank 0 to the other ranks; one process per GPU on ONE node (the launch contract of bench.py) shares a file system, so the
default exchange is a file (FileRendezvous). A host that already has a process group can pass its own broadcast instead
(torch_broadcast) - but must then import torch AFTER libetx_hip.so is loaded, or the loader binds the library to the older
ROCm runtime bundled with the wheel (etx_hip_create refuses that, include/etx_hip.h etx_hip_runtime_info). This module never
imports torch by itself.
"""
import os
import tempfile
import time


def shard_iterations(total_iterations, rank, world_size):
    """-> (first_iteration, stride, count) of the iterations rank `rank` renders out of 0..total-1."""
    if rank >= total_iterations:
        return rank, world_size, 0
    return rank, world_size, (total_iterations - rank + world_size - 1) // world_size


class FileRendezvous:
    """Hands rank 0's bytes to the other ranks of the same launch through a file in a directory every rank of the node sees.

    The file name is unique per launch: MASTER_ADDR / MASTER_PORT (what `python -m torch.distributed.run` exports to every worker), its run id,
    and the launcher's pid (every worker's parent) - a file left behind by a launch that died cannot be mistaken for this one's. Rank 0 writes to
    a temporary name and renames (readers never see a partial file) and removes the file once every rank has joined the communicator."""

    def __init__(self, rank, world_size, directory=None, key=None, timeout=300.0):
        self.rank, self.world_size, self.timeout = int(rank), int(world_size), float(timeout)
        directory = directory or os.environ.get("ETX_HIP_RENDEZVOUS_DIR") or tempfile.gettempdir()
        if key is None:
            key = "%s_%s_%s_%d" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid())
        self.path = os.path.join(directory, "etx_hip_rendezvous_%s_%d" % ("".join(c if c.isalnum() else "_" for c in key), self.world_size))

    def broadcast(self, payload):
        """rank 0: `payload` (bytes) -> every rank returns it."""
        if self.rank == 0:
            temporary = "%s.tmp%d" % (self.path, os.getpid())
            with open(temporary, "wb") as f:
                f.write(payload)
            os.replace(temporary, self.path)
            return bytes(payload)
        deadline = time.monotonic() + self.timeout
        while True:
            try:
                with open(self.path, "rb") as f:
                    data = f.read()
                if data:
                    return data
            except OSError:
                pass
            if time.monotonic() > deadline:
                raise TimeoutError("rank %d: rank 0 did not publish %s within %.0f s" % (self.rank, self.path, self.timeout))
            time.sleep(0.005)

    def finish(self):
        """after a collective that proves every rank has read the file (etx_hip_comm_init is one)"""
        if self.rank == 0:
            try:
                os.remove(self.path)
            except OSError:
                pass


def torch_broadcast(payload, rank):
    """The same exchange through a torch.distributed process group the HOST has already initialised (any backend)."""
    import torch.distributed as dist
    box = [payload if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def init_context_comm(context, rank, world_size, make_id=None, broadcast=None):
    """Creates the RCCL communicator inside libetx_hip.so: rank 0 makes the ncclUniqueId (etx_hip_comm_unique_id), every rank receives its 128
    bytes and calls etx_hip_comm_init (a collective: it returns once all ranks have joined).
    `broadcast(payload_or_None) -> bytes`: the exchange (default: FileRendezvous on this node); `make_id`: replaces etx_hip_comm_unique_id where
    no GPU exists (the CPU tests)."""
    from . import api
    make_id = make_id or getattr(context, "make_unique_id", None) or (lambda: api.comm_unique_id(context.library))
    rendezvous = None
    if broadcast is None:
        rendezvous = FileRendezvous(rank, world_size)
        broadcast = rendezvous.broadcast
    unique_id = broadcast(make_id() if rank == 0 else None)
    context.comm_init(rank, world_size, unique_id)
    if rendezvous is not None:
        barrier = getattr(context, "comm_barrier", None)
        if barrier is not None:
            barrier()  # every rank is past its read (a stub context's comm_init is not a collective)
        rendezvous.finish()
    return unique_id


def max_over_ranks(context, seconds):
    """bench.py: the timed region of the job is the slowest rank's (etx_hip_comm_all_reduce_f64, max; the identity on one rank)."""
    from . import api
    return float(context.comm_all_reduce([seconds], api.REDUCE_MAX)[0])
