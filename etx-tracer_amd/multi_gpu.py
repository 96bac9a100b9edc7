"""Multi-GPU: iterations are sharded over ranks, the only exchange is one sum-reduce of the film.

The reference has no communication layer. One VCM iteration is self-contained (its own light paths, photon grid,
radius and MIS weights depend only on the iteration index and the pixel count, vcm_cpu.cxx:100-113), so rank r of R
renders iterations r, r+R, r+2R, ... over the FULL frame with a scene replica, accumulates float4 SUMS (not the
reference's running mean, film.cxx:200-206, so that the reduce is associative) and a single all-reduce at the end
yields the whole-job image; the host divides by the total iteration count.

Production path: RCCL inside libetx_hip.so (etx_hip_comm_init / etx_hip_reduce_film); the 128-byte ncclUniqueId is
broadcast here through torch.distributed. `FilmAccumulator` is the same arithmetic on torch tensors (any backend)
and is what the gloo CPU tests exercise.
"""
import torch
import torch.distributed as dist


def shard_iterations(total_iterations, rank, world_size):
    """-> (first_iteration, stride, count) of the iterations rank `rank` renders out of 0..total-1."""
    if rank >= total_iterations:
        return rank, world_size, 0
    return rank, world_size, (total_iterations - rank + world_size - 1) // world_size


class FilmAccumulator:
    """Sum-accumulated film (camera + light) with an iteration counter; reduce() = all-reduce(sum) of all three."""

    def __init__(self, height, width, device="cpu"):
        self.camera_sum = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
        self.light_sum = torch.zeros((height, width, 4), dtype=torch.float32, device=device)
        self.iterations = torch.zeros((1,), dtype=torch.int64, device=device)

    def add_iteration(self, camera, light):
        self.camera_sum += camera
        self.light_sum += light
        self.iterations += 1

    def reduce(self, group=None):
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.camera_sum, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(self.light_sum, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(self.iterations, op=dist.ReduceOp.SUM, group=group)
        return self

    def result(self):
        """Film::Result = max(0, camera + light) of the means (film.cxx:401-409)"""
        n = max(int(self.iterations.item()), 1)
        out = torch.clamp((self.camera_sum + self.light_sum) / n, min=0.0)
        out[..., 3] = 1.0
        return out


def init_context_comm(context, rank, world_size):
    """Creates the RCCL communicator inside libetx_hip.so: rank 0 makes the ncclUniqueId, everyone receives it
    through the torch.distributed process group (any backend) and calls etx_hip_comm_init."""
    from . import api
    payload = [api.comm_unique_id(context.library) if rank == 0 else None]
    dist.broadcast_object_list(payload, src=0)
    context.comm_init(rank, world_size, payload[0])
