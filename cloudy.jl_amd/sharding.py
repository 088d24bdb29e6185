"""Multi-GPU: parcels are independent, so ranks own contiguous parcel ranges and exchange nothing per RHS.
The only collective is the all-reduce of the nmom moment sums of the conservation diagnostic
(reference analogue: moments_sum, test/examples/utils/plotting_helpers.jl:240-252): ncclAllReduce inside
libcloudy_hip.so (cloudy_moment_sums_allreduce), reached here through `Communicator`."""
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


class Communicator:
    """An RCCL communicator behind the C ABI (cloudy_comm_create: ncclCommInitRank on this rank's GPU).  The library
    does no rendezvous: `unique_id()` (rank 0) yields the 128 bytes every rank must pass to the constructor --
    `from_torch_distributed` moves them over an already initialised process group of ANY backend (gloo is enough: the
    group is only the courier, the collective itself is RCCL inside libcloudy_hip.so)."""

    def __init__(self, world_size, rank, uid, device=-1):
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(uid), _lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().cloudy_comm_create(int(world_size), int(rank), buf, int(device), C.byref(h)))
        self.handle, self.world_size, self.rank = h, int(world_size), int(rank)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(_lib.lib().cloudy_comm_unique_id(buf))
        return bytes(buf.raw)

    @classmethod
    def from_torch_distributed(cls, device=-1, group=None):
        rank, world, uid = cls.exchange_unique_id(group)
        return cls(world, rank, uid, device)

    @classmethod
    def exchange_unique_id(cls, group=None):
        """-> (rank, world, the 128 bytes of rank 0's unique id) over an initialised torch.distributed group: the
        rendezvous half of from_torch_distributed (the caller may then build the communicator under its own watchdog)."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # Every rank takes part in the broadcast whatever happens on rank 0: a failure there (RCCL not loadable, no
        # device) travels as ("error", text) and is raised on ALL ranks, so that the callers' next collective is the same
        # one everywhere (a rank 0 that raised before the broadcast left the others blocked in it).
        box = [None]
        if rank == 0:
            try:
                box[0] = ("id", cls.unique_id())
            except Exception as e:  # noqa: BLE001 -- whatever it is, the other ranks must hear of it
                box[0] = ("error", f"{type(e).__name__}: {e}")
        dist.broadcast_object_list(box, src=0, group=group)
        kind, payload = box[0]
        if kind != "id":
            raise _lib.CloudyError(_lib.ECOMM, "rank 0 could not create the RCCL unique id: " + payload)
        return rank, world, payload

    def allreduce_moment_sums(self, plan, arr, stream=None):
        """cloudy_moment_sums_allreduce: plane sums of this rank's (planes, n) device array summed over all ranks."""
        ptr, planes, n, ld = as_device(arr)
        out = DeviceArray(planes, 1)
        _lib.check(_lib.lib().cloudy_moment_sums_allreduce(plan.handle, self.handle, n, ld, planes, ptr, out.ptr, stream))
        _lib.check(_lib.lib().cloudy_stream_synchronize(stream))
        return out.to_numpy().reshape(-1)

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().cloudy_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def allreduce_sums(local_sums, group=None):
    """Sum per-rank moment sums that are already on the HOST over a torch.distributed group (CPU tests over gloo, and the
    fallback of bench.py when no RCCL communicator could be formed).  The device path is
    Communicator.allreduce_moment_sums (RCCL behind the C ABI, no torch tensor involved)."""
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
