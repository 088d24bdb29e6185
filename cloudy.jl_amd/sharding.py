"""Multi-GPU: parcels are independent, so ranks own contiguous parcel ranges and exchange nothing per RHS.
The only collective is the all-reduce of the nmom moment sums of the conservation diagnostic
(reference analogue: moments_sum, test/examples/utils/plotting_helpers.jl:240-252)."""
import ctypes as C

import numpy as np

from . import _lib
from .device import DeviceArray, as_device


def shard_range(n_parcels, rank, world_size):
    """[lo, hi) of the contiguous parcel range owned by `rank`: sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(int(n_parcels), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def moment_sums(plan, arr, stream=None):
    """Local per-plane sums of a (planes, n) device array -> numpy (planes,) (fp64 tree sum on the GPU)."""
    ptr, planes, n, ld = as_device(arr)
    out = DeviceArray(planes, 1)
    _lib.check(_lib.lib().cloudy_moment_sums(plan.handle, n, ld, planes, ptr, out.ptr, stream))
    _lib.check(_lib.lib().cloudy_stream_synchronize(stream))
    return out.to_numpy().reshape(-1)


def allreduce_sums(local_sums, group=None):
    """Sum the per-rank moment sums over all ranks with torch.distributed (RCCL on GPUs, gloo on CPU tests)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return np.asarray(local_sums, dtype=np.float64)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    t = torch.as_tensor(np.asarray(local_sums, dtype=np.float64), device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def mode_sums(global_sums, NProgMoms):
    """Sum moment j over the modes (the quantity conservation is stated for): M_j^total."""
    nmax = max(NProgMoms)
    out = np.zeros(nmax)
    off = 0
    for npm in NProgMoms:
        for j in range(npm):
            out[j] += global_sums[off + j]
        off += npm
    return out
