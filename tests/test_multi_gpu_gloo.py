"""CPU, world_size 2, gloo: iteration sharding + film sum-reduce give the same image as one rank rendering all
iterations (the N>1 path of bench.py without GPUs)."""
import os
import socket

import numpy as np

from tests.lazy_torch import torch, dist, mp  # imported on first use: a -m gpu run collects this module and must not load torch's ROCm runtime


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

        def comm_barrier(self):
            dist.barrier()

        def comm_all_reduce(self, values, op):
            t = torch.tensor(values, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return [float(v) for v in t]

    # through the host's own process group ...
    received = multi_gpu.init_context_comm(StubContext(), rank, world, make_id=lambda: bytes(range(128)), broadcast=lambda payload: multi_gpu.torch_broadcast(payload, rank))
    assert calls == [(rank, world, bytes(range(128)))] and bytes(received) == bytes(range(128))
    # ... and through the default exchange, a file on the node (what bench.py uses: no torch in a rendering process)
    os.environ["ETX_HIP_RENDEZVOUS_DIR"] = out_dir
    received = multi_gpu.init_context_comm(StubContext(), rank, world, make_id=lambda: bytes(range(64, 192)))
    assert calls[1] == (rank, world, bytes(range(64, 192))) and bytes(received) == bytes(range(64, 192))
    dist.barrier()
    assert [f for f in os.listdir(out_dir) if f.startswith("etx_hip_rendezvous")] == []  # rank 0 removed it after the barrier
    assert multi_gpu.max_over_ranks(StubContext(), 1.0 + rank) == float(world)  # the slowest rank's time
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


def bench_worker(rank, world, port, out_dir):
    """bench.py's main() as the driver launches it for N = 2 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), with gloo
    in place of nccl and tests/stub_context.StubContext in place of the device context: every line of the N > 1 branch runs."""
    import contextlib
    import io
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ETX_HIP_RENDEZVOUS_DIR": out_dir})
    dist.init_process_group("gloo", rank=rank, world_size=world)  # the stub context's collectives; bench.py itself never touches torch
    import bench
    from tests.stub_context import StubContext
    stdout = io.StringIO()
    with contextlib.redirect_stdout(stdout):
        line = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "classic", "--repeats", "3"], context_factory=StubContext)
    dist.barrier()
    dist.destroy_process_group()
    ctx = StubContext.instances[-1]
    printed = [l for l in stdout.getvalue().splitlines() if l.startswith("{")]
    assert (len(printed) == 1) == (rank == 0)  # ONE JSON line, from rank 0
    if rank == 0:
        assert json.loads(printed[0]) == json.loads(json.dumps(line))
        with open(os.path.join(out_dir, "line.json"), "w") as f:
            f.write(printed[0])
    assert ("comm_init", rank, world, bytes(range(128))) in ctx.calls
    assert ("begin_vcm", rank + 1 * world, world) in ctx.calls and ctx.calls[-1] == ("close",)
    # north_star's cadence (--reduce-every 1, the default): inside every region of 3 steps a reduce is begun after the first and the second
    # iteration and the blocking one closes the region; the warm-up run of 1 step has only its closing reduce. The same sequence on every rank.
    sequence = [c[0] for c in ctx.calls if c[0] in ("begin_vcm", "reduce_film_begin", "reduce_film")]
    region = ["begin_vcm", "reduce_film_begin", "reduce_film_begin", "reduce_film"]
    expected_sequence = ["begin_vcm", "reduce_film"] + region * 3 + (["begin_vcm"] if rank == 0 else [])  # rank 0: the per-kernel pass afterwards, no reduce
    assert sequence == expected_sequence, sequence
    np.save(os.path.join(out_dir, "bench_film%d.npy" % rank), ctx.reduced_result)
    np.save(os.path.join(out_dir, "bench_iterations%d.npy" % rank), np.array(ctx.reduced_iterations))


def test_bench_two_gpu_control_flow_under_gloo(tmp_path):
    import json
    from tests.stub_context import fake_iteration as stub_iteration
    world, steps, warmup, repeats = 2, 3, 1, 3
    mp.spawn(bench_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    line = json.loads(open(tmp_path / "line.json").read())
    assert line["n_gpus"] == 2 and line["steps"] == steps and line["warmup"] == warmup and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["repeats"]["count"] == repeats and line["repeats"]["min"] <= line["value"] == line["repeats"]["median"] <= line["repeats"]["max"]
    assert line["reduce"]["every"] == 1 and line["reduce"]["count"] == steps  # one reduce per iteration of the (median) region, the last one blocking
    assert line["config"]["workload"] == "cornell_classic_vcm_1920x1080" and line["unit"] == "Msamples/s"
    # value = the samples ALL ranks rendered in the timed region / the slowest rank's time
    assert abs(line["value"] - 1920 * 1080 * steps * world / (line["ms_per_step"] * 1.0e-3 * steps) / 1.0e6) < 1.0e-3 * line["value"]
    # the last timed region: rank r rendered iterations r + (offset + k) * world, and after its closing reduce every rank holds their mean
    offset = warmup + (repeats - 1) * steps
    rendered = sorted(int(i) for r in range(world) for i in np.load(tmp_path / ("bench_iterations%d.npy" % r)))
    assert rendered == list(range(offset * world, (offset + steps) * world))
    camera = sum(stub_iteration(i)[0] for i in rendered)
    light = sum(stub_iteration(i)[1] for i in rendered)
    expected = torch.clamp((camera + light) / len(rendered), min=0.0)
    expected[..., 3] = 1.0
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / ("bench_film%d.npy" % r)), expected.numpy(), rtol=1e-6, atol=1e-6)


def pixel_bench_worker(rank, world, port, out_dir):
    """bench.py --shard pixels for N = 2 under gloo: every rank renders the SAME iterations for its own pixels (etx_hip_begin_ex)."""
    import contextlib
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ETX_HIP_RENDEZVOUS_DIR": out_dir})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from tests.stub_context import StubContext
    stdout = io.StringIO()
    with contextlib.redirect_stdout(stdout):
        bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "cloud_bdpt", "--shard", "pixels", "--no-cpu-baseline", "--no-kernel-table", "--reduce-every", "2",
                    "--repeats", "2"],
                   context_factory=StubContext)
    dist.barrier()
    dist.destroy_process_group()
    ctx = StubContext.instances[-1]
    if rank == 0:
        with open(os.path.join(out_dir, "pixel_line.json"), "w") as f:
            f.write([l for l in stdout.getvalue().splitlines() if l.startswith("{")][0])
    assert ("begin_bdpt", 1, 1, rank, world) in ctx.calls  # first timed region: iterations warmup .. warmup + steps - 1 on EVERY rank, pixels rank, rank + world, ...
    # --reduce-every 2 over 3 steps: one asynchronous reduce (after the second iteration) and the closing one, in every region
    sequence = [c[0] for c in ctx.calls if c[0] in ("begin_bdpt", "reduce_film_begin", "reduce_film")]
    assert sequence == ["begin_bdpt", "reduce_film"] + ["begin_bdpt", "reduce_film_begin", "reduce_film"] * 2, sequence
    np.save(os.path.join(out_dir, "pixel_film%d.npy" % rank), ctx.reduced_result)
    np.save(os.path.join(out_dir, "pixel_iterations%d.npy" % rank), np.array(ctx.reduced_iterations))


def test_bench_pixel_sharding_control_flow_under_gloo(tmp_path):
    """camera images tile-local, light images full-frame, ONE sum-reduce of zero-padded sums; the iteration count comes from
    pixel shard 0 alone. The reduced film equals the unsharded mean of the same iterations."""
    import json
    from tests.stub_context import fake_iteration as stub_iteration
    world, steps, warmup, repeats = 2, 3, 1, 2
    mp.spawn(pixel_bench_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    line = json.loads(open(tmp_path / "pixel_line.json").read())
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["parallelism"].startswith("pixel-sharded x2")
    # the job is `steps` iterations of the frame, whatever the rank count
    assert abs(line["value"] - 2048 * 2048 * steps / (line["ms_per_step"] * 1.0e-3 * steps) / 1.0e6) < 1.0e-3 * line["value"]
    assert line["reduce"]["every"] == 2 and line["reduce"]["count"] == 2 and line["repeats"]["count"] == repeats
    first = warmup + (repeats - 1) * steps  # the last region's iterations
    for r in range(world):
        assert [int(i) for i in np.load(tmp_path / ("pixel_iterations%d.npy" % r))] == list(range(first, first + steps))
    camera = sum(stub_iteration(i)[0] for i in range(first, first + steps))
    light = sum(stub_iteration(i)[1] for i in range(first, first + steps))
    expected = torch.clamp((camera + light) / steps, min=0.0)
    expected[..., 3] = 1.0
    for r in range(world):
        np.testing.assert_allclose(np.load(tmp_path / ("pixel_film%d.npy" % r)), expected.numpy(), rtol=1e-6, atol=1e-6)


def cadence_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.stub_context import StubContext

    class Snapshot:
        film_size = (16, 12)

    ctx = StubContext(rank)
    ctx.upload_scene(Snapshot())
    ctx.begin_vcm(None, first_iteration=rank, iteration_stride=world)
    films = []
    # the ranks progress UNEVENLY between reduces (rank 0 two iterations, rank 1 one): every reduce still yields the mean of exactly the
    # iterations the ranks had committed, because the count travels with the sums
    for _ in range(3):
        for _ in range(2 - rank):
            ctx.render_iteration()
        ctx.reduce_film_begin()
        assert ctx.reduce_film_end(True)
        films.append((ctx.result().copy(), int(ctx.reduced_count.item())))
    ctx.render_iteration()  # rendering goes on after a reduce
    ctx.reduce_film()
    films.append((ctx.result().copy(), int(ctx.reduced_count.item())))
    np.save(os.path.join(out_dir, "cadence%d.npy" % rank), np.stack([f[0] for f in films]))
    np.save(os.path.join(out_dir, "cadence_counts%d.npy" % rank), np.array([f[1] for f in films]))
    np.save(os.path.join(out_dir, "cadence_own%d.npy" % rank), np.array(ctx.iterations_rendered))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_cadence_progressive_and_not_terminal(tmp_path):
    """SURVEY.md 8e row 2 / north_star: the film is reduced while the render runs - every reduce returns the job's film so far, the ranks'
    own sums stay their own, and the film after the last reduce equals the one-reduce-at-the-end film of the same iterations."""
    from tests.stub_context import fake_iteration as stub_iteration
    world = 2
    mp.spawn(cadence_worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    films = [np.load(tmp_path / ("cadence%d.npy" % r)) for r in range(world)]
    counts = [np.load(tmp_path / ("cadence_counts%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(films[0], films[1])
    assert list(counts[0]) == list(counts[1]) == [3, 6, 9, 11]
    own = [[int(i) for i in np.load(tmp_path / ("cadence_own%d.npy" % r))] for r in range(world)]
    assert own[0] == [0, 2, 4, 6, 8, 10, 12] and own[1] == [1, 3, 5, 7]
    for k, (n0, n1) in enumerate([(2, 1), (4, 2), (6, 3), (7, 4)]):
        rendered = own[0][:n0] + own[1][:n1]
        expected = torch.clamp(sum(stub_iteration(i)[0] + stub_iteration(i)[1] for i in rendered) / len(rendered), min=0.0)
        expected[..., 3] = 1.0
        np.testing.assert_allclose(films[0][k], expected.numpy(), rtol=1e-6, atol=1e-6)


def test_bench_under_the_real_launcher(tmp_path):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P <script>`: the driver's own command line for
    N > 1 (tests/bench_stub_launcher.py stands in for bench.py only in that it passes the stub context). The 128-byte id travels through the file
    rendezvous keyed by what the launcher exports; both ranks receive rank 0's id, run the same sequence of collectives and exit cleanly."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("ETX_HIP_RENDEZVOUS_DIR", None)  # the default directory (the system's temporary directory), as on the GPU node
    result = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
                             os.path.join(root, "tests", "bench_stub_launcher.py"), str(tmp_path)], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert result.returncode == 0, result.stdout[-3000:]
    printed = [l for l in result.stdout.splitlines() if l.startswith("{")]
    assert len(printed) == 1, result.stdout[-2000:]  # ONE JSON line, from rank 0
    line = json.loads(printed[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["repeats"]["count"] == 2
    ranks = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    ids = {tuple(map(tuple, r["comm_init"])) for r in ranks}
    assert [r["comm_init"][0][:2] for r in ranks] == [[0, 2], [1, 2]]
    assert ranks[0]["comm_init"][0][2] == ranks[1]["comm_init"][0][2] == bytes(range(128)).hex()  # the stub's id, made by rank 0
    assert [c for c in ranks[0]["sequence"] if c != "begin_vcm"] == [c for c in ranks[1]["sequence"] if c != "begin_vcm"]  # the same collectives in the same order
    import glob
    import tempfile
    assert glob.glob(os.path.join(tempfile.gettempdir(), "etx_hip_rendezvous_127_0_0_1_*")) == []  # removed after the barrier
