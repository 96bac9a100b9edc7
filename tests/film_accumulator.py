"""TEST DOUBLE (not product code): the arithmetic of the multi-GPU film reduce on torch tensors, any backend.
etx_hip_reduce_film does the same with RCCL on the device film: per-rank SUMS of the camera / light layers and the
iteration counter are all-reduced, the image is the sum divided by the total iteration count."""
from tests.lazy_torch import torch, dist


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
