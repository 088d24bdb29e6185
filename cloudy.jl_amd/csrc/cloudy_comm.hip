// cloudy_comm.hip -- the one collective of the path, behind the C ABI: the all-reduce of the nmom plane sums of the
// mass-conservation diagnostic (reference analogue: moments_sum over the modes, test/examples/utils/plotting_helpers.jl:
// 240-252) over RCCL.  Parcels shard with no data-path exchange; this is the only place ranks talk.
//
// RCCL is bound at FIRST USE (dlopen of librccl.so.1, six entry points) rather than at load time: the library is ~0.5 GB,
// a single-GPU host never needs it, and a host process that already carries an RCCL (same soname) keeps exactly one copy.
// No torch, no MPI: the rendezvous is the 128-byte unique id, which the host moves between its ranks by whatever means it
// has (Julia: MPI.bcast, Distributed.jl, a file); a single process driving several GPUs uses cloudy_comm_create_all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/cloudy_hip.h"

namespace cloudy_comm_detail {

// the slice of rccl.h this unit binds (ABI-stable since NCCL 2.0)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat64 = 8 };  // ncclDataType_t: ncclDouble
enum { ncclSum = 0 };      // ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    int (*GetVersion)(int *) = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    char why[256] = "";
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            std::snprintf(r.why, sizeof r.why, "librccl.so.1 not found: %s", dlerror());
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) {
                ok = false;
                std::snprintf(r.why, sizeof r.why, "librccl: missing symbol %s", name);
            }
            return p;
        };
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r;
}

}  // namespace cloudy_comm_detail

using namespace cloudy_comm_detail;

struct cloudy_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

// error text goes through the library's thread-local buffer (cloudy_hip.hip)
extern "C" void cloudy_set_last_error_(const char *msg);

namespace {

int cfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    cloudy_set_last_error_(buf);
    return code;
}

int need_rccl() {
    Rccl &r = rccl();
    if (!r.handle) return cfail(CLOUDY_EUNSUPPORTED, "RCCL is not available: %s", r.why);
    return CLOUDY_OK;
}

int nccl_fail(int rc, const char *what) {
    Rccl &r = rccl();
    return cfail(CLOUDY_ECOMM, "%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
}

struct DevGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DevGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            switched = err == hipSuccess;
        }
    }
    ~DevGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

extern "C" {

int cloudy_comm_rccl_version(void) {
    if (need_rccl() != CLOUDY_OK) return 0;
    int v = 0;
    if (rccl().GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int cloudy_comm_unique_id(void *id_out) {
    if (!id_out) return cfail(CLOUDY_EINVAL, "id_out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return cfail(CLOUDY_ENODEVICE, "no HIP device");
    int rc = need_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    const int n = rccl().GetUniqueId(&id);
    if (n != ncclSuccess) return nccl_fail(n, "ncclGetUniqueId");
    static_assert(sizeof(id) == CLOUDY_COMM_ID_BYTES, "unique id size");
    std::memcpy(id_out, &id, sizeof id);
    return CLOUDY_OK;
}

int cloudy_comm_create(int world_size, int rank, const void *id, int device, cloudy_comm **out) {
    if (!out) return cfail(CLOUDY_EINVAL, "out is NULL");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size) return cfail(CLOUDY_EINVAL, "need 0 <= rank < world_size");
    if (!id) return cfail(CLOUDY_EINVAL, "id is NULL (rank 0 calls cloudy_comm_unique_id and the host distributes it)");
    // (the device check comes first, as in cloudy_comm_unique_id: a host without a GPU answers ENODEVICE whether or not
    // librccl is installed, and never loads the 0.5 GB library)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return cfail(CLOUDY_ENODEVICE, "no HIP device");
    int rc = need_rccl();
    if (rc) return rc;
    if (device < 0) (void)hipGetDevice(&device);
    if (device >= ndev) return cfail(CLOUDY_EINVAL, "device %d out of range (%d devices)", device, ndev);
    cloudy_comm *c = new (std::nothrow) cloudy_comm();
    if (!c) return cfail(CLOUDY_ENOMEM, "out of host memory");
    c->rank = rank;
    c->world = world_size;
    c->device = device;
    DevGuard g(device);  // the communicator binds to the current device
    if (g.err != hipSuccess) {
        delete c;
        return cfail(CLOUDY_EHIP, "hipSetDevice(%d): %s", device, hipGetErrorString(g.err));
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    const int n = rccl().CommInitRank(&c->comm, world_size, uid, rank);
    if (n != ncclSuccess) {
        delete c;
        return nccl_fail(n, "ncclCommInitRank");
    }
    *out = c;
    return CLOUDY_OK;
}

int cloudy_comm_create_all(int n_devices, const int *devices, cloudy_comm **comms_out) {
    if (!comms_out || n_devices < 1) return cfail(CLOUDY_EINVAL, "need n_devices >= 1 and comms_out");
    for (int i = 0; i < n_devices; ++i) comms_out[i] = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return cfail(CLOUDY_ENODEVICE, "no HIP device");
    int rc = need_rccl();
    if (rc) return rc;
    if (n_devices > 64) return cfail(CLOUDY_EUNSUPPORTED, "n_devices > 64");
    int devs[64];
    for (int i = 0; i < n_devices; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) return cfail(CLOUDY_EINVAL, "device %d out of range (%d devices)", devs[i], ndev);
    }
    ncclComm_t raw[64];
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int n = rccl().CommInitAll(raw, n_devices, devs);
    (void)hipSetDevice(prev);
    if (n != ncclSuccess) return nccl_fail(n, "ncclCommInitAll");
    for (int i = 0; i < n_devices; ++i) {
        cloudy_comm *c = new (std::nothrow) cloudy_comm();
        if (!c) {
            for (int j = 0; j < n_devices; ++j) {
                if (j < i) delete comms_out[j];
                (void)rccl().CommDestroy(raw[j]);
                comms_out[j] = nullptr;
            }
            return cfail(CLOUDY_ENOMEM, "out of host memory");
        }
        c->comm = raw[i];
        c->rank = i;
        c->world = n_devices;
        c->device = devs[i];
        comms_out[i] = c;
    }
    return CLOUDY_OK;
}

void cloudy_comm_destroy(cloudy_comm *comm) {
    if (!comm) return;
    if (comm->comm && rccl().handle) {
        DevGuard g(comm->device);
        (void)rccl().CommDestroy(comm->comm);
    }
    delete comm;
}

int cloudy_comm_rank(const cloudy_comm *comm) { return comm ? comm->rank : cfail(CLOUDY_EINVAL, "comm is NULL"); }
int cloudy_comm_world_size(const cloudy_comm *comm) { return comm ? comm->world : cfail(CLOUDY_EINVAL, "comm is NULL"); }
int cloudy_comm_device(const cloudy_comm *comm) { return comm ? comm->device : cfail(CLOUDY_EINVAL, "comm is NULL"); }

int cloudy_comm_group_start(void) {
    int rc = need_rccl();
    if (rc) return rc;
    const int n = rccl().GroupStart();
    return n == ncclSuccess ? CLOUDY_OK : nccl_fail(n, "ncclGroupStart");
}

int cloudy_comm_group_end(void) {
    int rc = need_rccl();
    if (rc) return rc;
    const int n = rccl().GroupEnd();
    return n == ncclSuccess ? CLOUDY_OK : nccl_fail(n, "ncclGroupEnd");
}

int cloudy_allreduce_sum_f64(cloudy_comm *comm, const double *send_dev, double *recv_dev, size_t count, void *stream) {
    if (!comm) return cfail(CLOUDY_EINVAL, "comm is NULL");
    if (count == 0) return CLOUDY_OK;
    if (!send_dev || !recv_dev) return cfail(CLOUDY_EINVAL, "device buffer is NULL");
    DevGuard g(comm->device);
    if (g.err != hipSuccess) return cfail(CLOUDY_EHIP, "selecting the communicator's device: %s", hipGetErrorString(g.err));
    const int n = rccl().AllReduce(send_dev, recv_dev, count, ncclFloat64, ncclSum, comm->comm, (hipStream_t)stream);
    return n == ncclSuccess ? CLOUDY_OK : nccl_fail(n, "ncclAllReduce");
}

int cloudy_moment_sums_allreduce(const cloudy_plan *plan, cloudy_comm *comm, size_t n_parcels, size_t ld, int planes,
                                 const void *arr_dev, double *sums_dev, void *stream) {
    if (!comm) return cfail(CLOUDY_EINVAL, "comm is NULL");
    if (!plan) return cfail(CLOUDY_EINVAL, "plan is NULL");
    if (cloudy_plan_device(plan) != comm->device)
        return cfail(CLOUDY_EINVAL, "plan lives on device %d, communicator on device %d", cloudy_plan_device(plan), comm->device);
    // local two-pass plane sums (deterministic order), then the all-reduce in place, both on the caller's stream
    int rc = cloudy_moment_sums(plan, n_parcels, ld, planes, arr_dev, sums_dev, stream);
    if (rc) return rc;
    return cloudy_allreduce_sum_f64(comm, sums_dev, sums_dev, (size_t)planes, stream);
}

}  // extern "C"
