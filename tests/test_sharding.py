"""Multi-GPU path on CPU: parcel-range sharding and the moment-sum all-reduce over gloo, world_size 2."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly(cloudy):
    for n in (0, 1, 7, 10_000_000, 10_000_003):
        for w in (1, 2, 3, 4, 8):
            r = [cloudy.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_mode_sums(cloudy):
    s = np.arange(1.0, 9.0)  # modes (3, 2, 3)
    assert np.array_equal(cloudy.mode_sums(s, (3, 2, 3)), [1 + 4 + 6, 2 + 5 + 7, 3 + 8])


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import bench
    n = 10007
    mom = bench.synth_moments(2, n, seed=3)          # every rank can regenerate the global batch
    lo, hi = pkg.shard_range(n, rank, world)
    local = mom[:, lo:hi].sum(axis=1)                # stands in for cloudy_moment_sums on the rank's GPU
    glob = pkg.allreduce_sums(local)
    ref = sum(mom[:, a:b].sum(axis=1) for a, b in (pkg.shard_range(n, r, world) for r in range(world)))
    assert np.allclose(glob, ref, rtol=1e-15, atol=0), (glob, ref)
    assert np.allclose(glob, mom.sum(axis=1), rtol=1e-12)
    tot = pkg.mode_sums(glob, (3, 3))
    assert tot.shape == (3,)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_allreduce_of_moment_sums_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("ok") == 2


def test_allreduce_of_moment_sums_gloo_world8(tmp_path):
    """the same exchange at the world size of the driver's scaling sweep (8 ranks, gloo, CPU): contiguous parcel ranges,
    the all-reduced sums equal to the sums over the whole batch"""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
           "127.0.0.1", "--master-port", "29531", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("ok") == 8


FAIL_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import __graft_entry__ as ge
    pkg = ge.load_package()
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    # no GPU on this box: cloudy_comm_unique_id fails on rank 0.  Every rank must hear of it through the SAME broadcast and
    # raise -- then the next collective (bench._comm_setup's all-reduced flag) matches on all ranks.
    try:
        pkg.Communicator.from_torch_distributed(0)
        print("rank", rank, "unexpectedly formed a communicator")
    except pkg.CloudyError as e:
        assert "rank 0 could not create the RCCL unique id" in str(e), str(e)
        import torch
        flag = torch.tensor([0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        print("rank", rank, "raised consistently")
    dist.barrier()
    dist.destroy_process_group()
""")


def test_unique_id_failure_on_rank0_reaches_every_rank(tmp_path, cloudy):
    """rank 0 used to raise BEFORE the broadcast of the unique id and leave the other ranks blocked in it."""
    if cloudy.device_count() > 0:
        pytest.skip("needs a box without a HIP device (rank 0 must fail to create the id)")
    script = tmp_path / "fail_worker.py"
    script.write_text(FAIL_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", str(script)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.count("raised consistently") == 2


def test_bench_gpus_flag_rejects_mismatched_launch():
    """`--gpus N` under a launcher that started a different number of ranks is an error, not a silent 1-rank run."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_bench_gpus_2_spawns_two_ranks(gpu_cloudy):
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks (before touching the GPU), rank 0's
    JSON line reports n_gpus = 2, a per-rank roofline entry for each rank and the all-reduced mass residual.  On the
    1-GPU test box both ranks share GPU 0 and rendezvous over gloo (CLOUDY_BENCH_BACKEND); the driver's multi-GPU runs
    use nccl (= RCCL) with one GPU per rank."""
    import json

    env = dict(os.environ, CLOUDY_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--parcels", "400000", "--steps", "5",
                        "--warmup", "2", "--no-variants", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_parcels"] == 800000 and out["scaling"] == "weak"
    assert len(line) < 4096 and len(out["roofline"]["per_rank_kernel_ms"]) == 2
    assert all(ms > 0 for ms in out["roofline"]["per_rank_kernel_ms"])
    assert out["mass_rate_residual"] is not None and out["value"] > 0
    # both ranks share the one GPU of the test box: no RCCL communicator can be formed, and the line says so
    assert out["collective_fallback"] is True and "share one GPU" in out["collective"]


@pytest.mark.gpu
def test_bench_gpus_8_on_one_gpu_with_an_empty_jit_cache(gpu_cloudy, tmp_path):
    """the shape of the driver's 8-GPU run on the one test GPU -- 8 ranks (gloo as the courier, all
    sharing GPU 0) that hiprtc-compile the SAME plans at the same moment into one EMPTY cache directory (the atomic-rename
    path of jit.hpp under 8-way contention).  One JSON line, 8 per-rank entries, the all-reduced residual."""
    import json

    cache = tmp_path / "jit_cache"
    cache.mkdir()
    env = dict(os.environ, CLOUDY_BENCH_BACKEND="gloo", CLOUDY_HIP_CACHE_DIR=str(cache))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--parcels", "200000", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = p.stdout.splitlines()[-1]
    assert len(line) < 4096, len(line)
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["config"]["global_parcels"] == 1_600_000 and out["scaling"] == "weak"
    assert len(out["roofline"]["per_rank_kernel_ms"]) == 8 and all(ms > 0 for ms in out["roofline"]["per_rank_kernel_ms"])
    assert out["mass_rate_residual"] is not None and out["value"] > 0
    files = [f for f in os.listdir(cache) if not f.endswith(".tmp")]
    assert files and not [f for f in os.listdir(cache) if f.endswith(".tmp")], os.listdir(cache)


@pytest.mark.gpu
def test_bench_one_process_mode(gpu_cloudy):
    """`bench.py --one-process --gpus N` -- one process, N GPUs, cloudy_comm_create_all and the group
    bracket around the per-device all-reduce (the single-Julia-process pattern).  The test box has one GPU: N = 1 (the
    ncclCommInitAll path with one device); more devices than visible is an error, not a silent smaller run."""
    import json

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--one-process", "--gpus", "1", "--parcels", "300000",
                        "--steps", "5", "--warmup", "2"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = p.stdout.splitlines()[-1]
    assert len(line) < 4096
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["config"]["global_parcels"] == 300000 and "ONE process" in out["config"]["launch"]
    assert out["value"] > 0 and out["mass_rate_residual"] < 1e-9 and "ncclAllReduce" in out["collective"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--one-process", "--gpus", "64", "--parcels", "1000",
                        "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "HIP device(s) visible" in (p.stdout + p.stderr)


def test_comm_entry_points_without_a_gpu(cloudy):
    """The RCCL-backed entry points validate their arguments and report status codes without a device (no compute, no
    communicator is formed on the CPU box); with a GPU this test only checks the argument errors."""
    import ctypes as C

    L, E = cloudy.lib(), cloudy._lib
    h = C.c_void_p()
    buf = C.create_string_buffer(E.COMM_ID_BYTES)
    assert L.cloudy_comm_create(2, 5, buf, -1, C.byref(h)) == E.EINVAL and b"rank" in L.cloudy_last_error()
    assert L.cloudy_comm_create(1, 0, None, -1, C.byref(h)) == E.EINVAL
    assert L.cloudy_comm_unique_id(None) == E.EINVAL
    assert L.cloudy_allreduce_sum_f64(None, None, None, 3, None) == E.EINVAL
    assert L.cloudy_moment_sums_allreduce(None, None, 0, 0, 1, None, None, None) == E.EINVAL
    if cloudy.device_count() == 0:
        assert L.cloudy_comm_unique_id(buf) == E.ENODEVICE
        assert L.cloudy_comm_create(1, 0, buf, -1, C.byref(h)) == E.ENODEVICE and not h.value
        assert L.cloudy_comm_create_all(1, None, C.byref(h)) == E.ENODEVICE
    L.cloudy_comm_destroy(None)   # a no-op


@pytest.mark.gpu
def test_rccl_communicator_behind_the_c_abi_one_rank(gpu_cloudy):
    """cloudy_comm_create forms a REAL RCCL communicator (ncclCommInitRank, 1 rank) on the test GPU from inside
    libcloudy_hip.so; cloudy_moment_sums_allreduce = local plane sums + ncclAllReduce(sum, f64) on the caller's stream
    must reproduce cloudy_moment_sums bit for bit (the sum over one rank is the rank's own sum)."""
    import ctypes as C

    import bench

    cloudy = gpu_cloudy
    L = cloudy.lib()
    assert L.cloudy_comm_rccl_version() >= 20000
    wl = bench.make_workload("cfg3a", 300_001, seed=9)
    plan = wl["coal_data"].plan(wl["dist_types"])
    m = cloudy.DeviceArray.from_numpy(wl["mom"])
    dm = cloudy.DeviceArray.zeros(*wl["mom"].shape)
    cloudy.make_box_model_rhs(cloudy.AnalyticalCoalStyle())(dm, m, wl["par"], 0.0)
    local = cloudy.moment_sums(plan, dm)
    comm = cloudy.Communicator(1, 0, cloudy.Communicator.unique_id())
    assert L.cloudy_comm_rank(comm.handle) == 0 and L.cloudy_comm_world_size(comm.handle) == 1
    assert L.cloudy_comm_device(comm.handle) == L.cloudy_plan_device(plan.handle)
    glob = comm.allreduce_moment_sums(plan, dm)
    assert np.array_equal(glob, local) and np.all(np.isfinite(glob) | np.isnan(local))
    # out of place, explicit stream-ordered use: recv = sum over ranks of send
    send = cloudy.DeviceArray.from_numpy(np.arange(1.0, 7.0).reshape(6, 1))
    recv = cloudy.DeviceArray.zeros(6, 1)
    cloudy._lib.check(L.cloudy_allreduce_sum_f64(comm.handle, send.ptr, recv.ptr, 6, None))
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    assert np.array_equal(recv.to_numpy().ravel(), np.arange(1.0, 7.0))
    comm.close()
    # the single-process form (ncclCommInitAll over the visible GPUs; one here) with the group bracket
    hs = (C.c_void_p * 1)()
    cloudy._lib.check(L.cloudy_comm_create_all(1, None, hs))
    out = cloudy.DeviceArray.zeros(6, 1)
    cloudy._lib.check(L.cloudy_comm_group_start())
    cloudy._lib.check(L.cloudy_moment_sums_allreduce(plan.handle, hs[0], 300_001, 300_001, 6, dm.ptr, out.ptr, None))
    cloudy._lib.check(L.cloudy_comm_group_end())
    cloudy._lib.check(L.cloudy_stream_synchronize(None))
    assert np.array_equal(out.to_numpy().ravel(), local)
    L.cloudy_comm_destroy(hs[0])


@pytest.mark.gpu
def test_bench_single_rank_line_uses_the_c_abi_collective(gpu_cloudy):
    """`python bench.py` (N = 1) forms the 1-rank RCCL communicator through the C ABI and says so in its JSON line"""
    import json

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--parcels", "400000", "--steps", "5", "--warmup", "2",
                        "--no-variants", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads(p.stdout.splitlines()[-1])
    assert "ncclAllReduce" in out["collective"] and "libcloudy_hip.so" in out["collective"], out["collective"]
    assert out["mass_rate_residual"] is not None and out["collective_fallback"] is False
