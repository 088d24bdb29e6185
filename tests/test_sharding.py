"""Multi-GPU path on CPU: parcel-range sharding and the moment-sum all-reduce over gloo, world_size 2."""
import os
import subprocess
import sys
import textwrap
import time

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


def _bench_dry(tmp_path, n, fault, extra_env=None, launcher="spawn", port=29641):
    """`bench.py --gpus n` with the control plane only (no GPU here) and one injected failure; -> (CompletedProcess, seconds)"""
    env = dict(os.environ, CLOUDY_BENCH_DRY_CONTROL_PLANE="1", CLOUDY_BENCH_LOG_DIR=str(tmp_path), OMP_NUM_THREADS="1",
               CLOUDY_BENCH_FAULT=fault, **(extra_env or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    bench = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"]
    if launcher == "spawn":
        cmd = [sys.executable] + bench
    else:   # the driver's launch line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + bench
    t0 = time.monotonic()
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=170)
    return p, time.monotonic() - t0


def test_launcher_control_plane_runs_clean_without_faults(tmp_path):
    """the dry control plane itself (rendezvous, barriers, max over ranks, gather of the per-rank records, rank 0's line) at
    world size 3 through the parent's own spawner: exit 0, one strict-JSON line that says it is NOT a measurement, a log
    pair per rank with the end-of-run mark"""
    import json

    p, dt = _bench_dry(tmp_path, 3, "")
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.splitlines()[-1])
    assert out["n_gpus"] == 3 and out["value"] is None and "dry-run" in out["data"] and len(out["devices"]) == 3
    for r in range(3):
        assert "bench.py rank done" in (tmp_path / f"rank{r}.err").read_text()


@pytest.mark.parametrize("fault, expect", [
    ("raise@after_init:1", "rank 1 exited with code 5"),          # a rank that raises before the first barrier
    ("raise@before_barrier:2", "rank 2 exited with code 5"),
    ("exit0@before_barrier:1", "rank 1 exited with code 0 before the end of the run"),   # a rank that exits 0 early
])
def test_launcher_ends_the_run_when_a_rank_dies(tmp_path, fault, expect):
    """`bench.py --gpus 3` (the parent's own spawner) with one rank that dies while its siblings
    sit in a collective that will never complete: the parent notices within its polling interval, terminates the siblings
    and exits NON-ZERO with one line naming the rank and the reason -- not after torch's 30-minute collective timeout."""
    p, dt = _bench_dry(tmp_path, 3, fault)
    assert p.returncode != 0 and dt < 60, (p.returncode, dt, p.stderr[-2000:])
    last = [l for l in p.stderr.splitlines() if "FAILED" in l][-1]
    assert expect in last and "rank<k>.err" in last, last
    if fault.startswith("raise"):
        assert "injected failure" in last
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]     # no line from a failed run
    # nobody is left behind: every child of the parent has been reaped (its process group is gone)
    assert (tmp_path / "rank0.err").exists() and (tmp_path / "rank2.err").exists()


def test_launcher_wall_clock_limit_ends_a_hung_rank(tmp_path):
    """a rank that neither dies nor arrives: the parent's wall-clock limit (CLOUDY_BENCH_LAUNCH_TIMEOUT) ends the run"""
    p, dt = _bench_dry(tmp_path, 2, "hang@before_barrier:1", {"CLOUDY_BENCH_LAUNCH_TIMEOUT": "8"})
    assert p.returncode == 6 and dt < 60, (p.returncode, dt, p.stderr[-2000:])
    assert "still running after 8 s" in p.stderr


@pytest.mark.parametrize("fault", ["raise@before_barrier:1", "exit0@before_barrier:1", "hang@before_barrier:1"])
def test_driver_launch_line_survives_a_dead_or_hung_rank(tmp_path, fault):
    """the same three failures under the DRIVER's launcher (`python -m torch.distributed.run ... bench.py --gpus 2`): the
    surviving rank's control-plane collective fails -- at once when the peer's sockets close, after
    CLOUDY_BENCH_DIST_TIMEOUT (default 300 s, 8 s here) when the peer hangs -- with a one-line reason, non-zero exit"""
    p, dt = _bench_dry(tmp_path, 2, fault, {"CLOUDY_BENCH_DIST_TIMEOUT": "8"}, launcher="torchrun",
                       port=29643 + len(fault) % 7)
    assert p.returncode != 0 and dt < 60, (p.returncode, dt, p.stderr[-2000:])
    assert "bench.py rank" in p.stderr and "FAILED" in p.stderr or "injected failure" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_fewer_devices_than_ranks_is_an_error(tmp_path, cloudy):
    """ranks are never mapped onto a shared GPU silently.  Here: no device at all for 2 ranks."""
    if cloudy.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 HIP devices")
    env = dict(os.environ, CLOUDY_BENCH_LOG_DIR=str(tmp_path), OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CLOUDY_BENCH_ALLOW_SHARED_GPU"):
        env.pop(k, None)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=170)
    assert p.returncode == 4 and time.monotonic() - t0 < 60, (p.returncode, p.stderr[-2000:])
    assert "HIP device(s) visible for 2 local rank(s)" in p.stderr and "FAILED" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_gpus_2_spawns_two_ranks(gpu_cloudy, tmp_path):
    """`python bench.py --gpus 2` without a launcher: the parent starts two ranks (before touching the GPU), rank 0's
    JSON line reports n_gpus = 2, a per-rank roofline entry for each rank and the all-reduced mass residual.  On the
    1-GPU test box both ranks share GPU 0 and rendezvous over gloo (CLOUDY_BENCH_BACKEND); the driver's multi-GPU runs
    use nccl (= RCCL) with one GPU per rank."""
    import json

    env = dict(os.environ, CLOUDY_BENCH_BACKEND="gloo", CLOUDY_BENCH_ALLOW_SHARED_GPU="1", CLOUDY_BENCH_LOG_DIR=str(tmp_path))
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
    # which device every rank ran on (one PCI bus id here, shared; N distinct ones on a real node)
    assert out["ranks_share_a_gpu"] is True and out["distinct_devices"] == 1
    assert len(out["devices"]) == 2 and out["devices"][0] == out["devices"][1] and ":" in out["devices"][0]
    assert os.path.exists(tmp_path / "rank1.err") and "bench.py rank done" in (tmp_path / "rank1.err").read_text()
    # ... and without the opt-in, ranks that would share a GPU are an ERROR (exit code 4), not a 1x "scaling curve"
    env.pop("CLOUDY_BENCH_ALLOW_SHARED_GPU")
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--parcels", "1000", "--steps", "1",
                        "--warmup", "1", "--no-variants", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode == 4 and "1 HIP device(s) visible for 2 local rank(s)" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")] and time.monotonic() - t0 < 60


@pytest.mark.gpu
def test_bench_gpus_8_on_one_gpu_with_an_empty_jit_cache(gpu_cloudy, tmp_path):
    """the shape of the driver's 8-GPU run on the one test GPU -- 8 ranks (gloo as the courier, all
    sharing GPU 0) that hiprtc-compile the SAME plans at the same moment into one EMPTY cache directory (the atomic-rename
    path of jit.hpp under 8-way contention).  One JSON line, 8 per-rank entries, the all-reduced residual."""
    import json

    cache = tmp_path / "jit_cache"
    cache.mkdir()
    env = dict(os.environ, CLOUDY_BENCH_BACKEND="gloo", CLOUDY_HIP_CACHE_DIR=str(cache), CLOUDY_BENCH_ALLOW_SHARED_GPU="1",
               CLOUDY_BENCH_LOG_DIR=str(tmp_path / "logs"))
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
