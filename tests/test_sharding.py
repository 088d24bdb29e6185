"""Multi-GPU path on CPU: parcel-range sharding and the moment-sum all-reduce over gloo, world_size 2."""
import os
import subprocess
import sys
import textwrap

import numpy as np

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


def test_bench_gpus_flag_rejects_mismatched_launch():
    """`--gpus N` under a launcher that started a different number of ranks is an error, not a silent 1-rank run."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stderr + p.stdout)


import pytest  # noqa: E402


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
    assert [r["rank"] for r in out["roofline"]["per_rank"]] == [0, 1]
    assert all(r["kernel_ms"] > 0 for r in out["roofline"]["per_rank"])
    assert out["mass_rate_residual"] is not None and out["value"] > 0
