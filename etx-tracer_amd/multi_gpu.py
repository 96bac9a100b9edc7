"""Multi-GPU: iterations are sharded over ranks, the only exchange is one sum-reduce of the film.

The reference has no communication layer. One VCM iteration is self-contained (its own light paths, photon grid,
radius and MIS weights depend only on the iteration index and the pixel count, vcm_cpu.cxx:100-113), so rank r of R
renders iterations r, r+R, r+2R, ... over the FULL frame with a scene replica, accumulates float4 SUMS (not the
reference's running mean, film.cxx:200-206, so that the reduce is associative) and a single all-reduce at the end
yields the whole-job image; the host divides by the total iteration count.

Production path: RCCL inside libetx_hip.so (etx_hip_comm_init / etx_hip_reduce_film); the 128-byte ncclUniqueId is
broadcast here through torch.distributed (init_context_comm). The CPU tests (gloo, world size 2) run this plumbing with a
stub context and check the sharding / sum / normalisation arithmetic with a stand-in film (tests/film_accumulator.py).
"""
# torch is imported where a process group is used, never at module level: a process that only renders (tests -m gpu, a host application) keeps ONE ROCm
# runtime - the one libetx_hip.so was built and linked against - instead of the older copy bundled with the torch wheel


def shard_iterations(total_iterations, rank, world_size):
    """-> (first_iteration, stride, count) of the iterations rank `rank` renders out of 0..total-1."""
    if rank >= total_iterations:
        return rank, world_size, 0
    return rank, world_size, (total_iterations - rank + world_size - 1) // world_size


def init_context_comm(context, rank, world_size, make_id=None):
    """Creates the RCCL communicator inside libetx_hip.so: rank 0 makes the ncclUniqueId (etx_hip_comm_unique_id), everyone
    receives its 128 bytes through the torch.distributed process group (any backend) and calls etx_hip_comm_init.
    `make_id`: replaces etx_hip_comm_unique_id where no GPU exists (the gloo tests)."""
    import torch.distributed as dist
    from . import api
    make_id = make_id or getattr(context, "make_unique_id", None) or (lambda: api.comm_unique_id(context.library))
    payload = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(payload, src=0)
    context.comm_init(rank, world_size, payload[0])
    return payload[0]


def max_over_ranks(seconds, device=None):
    """bench.py: the timed region of the job is the slowest rank's."""
    import torch
    import torch.distributed as dist
    if (dist.is_initialized() is False) or (dist.get_world_size() == 1):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
