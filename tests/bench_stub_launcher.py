"""Launched by tests/test_multi_gpu_gloo.py::test_bench_under_the_real_launcher through `python -m torch.distributed.run` exactly as the driver launches
bench.py for N > 1 - same environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT, TORCHELASTIC_RUN_ID, one parent for all workers) -, with
tests/stub_context.StubContext in place of the device context (this container has no GPU) and gloo behind the stub's collectives. What it proves: the
file rendezvous of etx_tracer_amd/multi_gpu.py finds its peers under the real launcher, every rank runs the same sequence, rank 0 prints ONE JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir = sys.argv[1]
    import torch.distributed as dist
    dist.init_process_group("gloo")  # env:// - the stub context's collectives; bench.py itself never touches torch
    import bench
    from tests.stub_context import StubContext
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    line = bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--repeats", "2", "--workload", "classic", "--no-kernel-table"], context_factory=StubContext)
    ctx = StubContext.instances[-1]
    ids = [c for c in ctx.calls if c[0] == "comm_init"]
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump({"rank": rank, "world": world, "comm_init": [[c[1], c[2], c[3].hex()] for c in ids], "line": line,
                   "sequence": [c[0] for c in ctx.calls if c[0] in ("begin_vcm", "reduce_film_begin", "reduce_film", "comm_barrier")]}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
