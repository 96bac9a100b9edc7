"""CPU, world_size 2, gloo: iteration sharding + film sum-reduce give the same image as one rank rendering all
iterations (the N>1 path of bench.py without GPUs)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def fake_iteration(iteration, h, w):
    """deterministic stand-in for one VCM iteration: camera and light images that depend on the iteration index"""
    g = torch.Generator().manual_seed(1234 + iteration)
    camera = torch.rand((h, w, 4), generator=g) * (1.0 + 0.01 * iteration)
    light = torch.rand((h, w, 4), generator=g) * 0.1
    return camera, light


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, total, h, w, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from etx_tracer_amd import multi_gpu
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, stride, count = multi_gpu.shard_iterations(total, rank, world)
    from tests.film_accumulator import FilmAccumulator
    film = FilmAccumulator(h, w)
    for k in range(count):
        film.add_iteration(*fake_iteration(first + k * stride, h, w))
    film.reduce()
    # the id-broadcast plumbing of bench.py / a multi-GPU host: rank 0 makes the 128-byte id, everyone calls comm_init with it
    calls = []

    class StubContext:
        library = None

        def comm_init(self, r, w, unique_id):
            calls.append((r, w, bytes(unique_id)))

    received = multi_gpu.init_context_comm(StubContext(), rank, world, make_id=lambda: bytes(range(128)))
    assert calls == [(rank, world, bytes(range(128)))] and bytes(received) == bytes(range(128))
    assert multi_gpu.max_over_ranks(1.0 + rank) == float(world)  # the slowest rank's time
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), film.result().numpy())
    np.save(os.path.join(out_dir, "count%d.npy" % rank), film.iterations.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_iterations_cover_everything():
    from etx_tracer_amd import multi_gpu
    for total in (1, 2, 7, 64):
        for world in (1, 2, 4, 8):
            seen = []
            for rank in range(world):
                first, stride, count = multi_gpu.shard_iterations(total, rank, world)
                seen += [first + k * stride for k in range(count)]
            assert sorted(seen) == list(range(total))


def test_two_ranks_equal_one_rank(tmp_path):
    from etx_tracer_amd import multi_gpu
    total, h, w = 7, 12, 16  # odd count: ranks hold different numbers of iterations (SURVEY.md 8e "if G does not divide spp")
    mp.spawn(worker, args=(2, free_port(), total, h, w, str(tmp_path)), nprocs=2, join=True)
    from tests.film_accumulator import FilmAccumulator
    single = FilmAccumulator(h, w)
    for it in range(total):
        single.add_iteration(*fake_iteration(it, h, w))
    expected = single.result().numpy()
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert int(np.load(tmp_path / "count0.npy")[0]) == total
    np.testing.assert_array_equal(r0, r1)                    # every rank holds the whole-job image
    np.testing.assert_allclose(r0, expected, rtol=1e-6, atol=1e-6)  # fp32 sums in a different order
