// helm_comm.cpp — the RCCL communicator behind include/helm_comm.h.
//
// The sharded unit is the level of reference src/circuit.rs:531 (gates mode), :1057 (LUT mode) and
// :1321 (arithmetic mode); the reference itself has no multi-GPU path.  One process per GPU, keys and
// wire tables replicated, the output LWE rows of a launch all-gathered over RCCL / xGMI.
//
// RCCL is bound with dlopen at first use: a process that already holds a librccl.so.1 (PyTorch ships
// its own) must not get a second, different copy mapped over it, and a machine without RCCL must still
// load libhelm_hip.so (single-GPU use).  Only entry points of the stable NCCL 2 ABI are used.
#include "../../include/helm_comm.h"
#include "../../include/helm_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <link.h>
#include <limits.h>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include <algorithm>

int helm_hip_fail_(int code, const std::string &msg); // helm_hip.hip: sets helm_hip_last_error()

// ------------------------------------------------------------------------------------------------------------------
// One HIP runtime per process.  A hipStream_t, an event or a device pointer means something to the libamdhip64 that made it
// and to no other copy: a process that maps two (PyTorch ships its own next to libtorch_hip.so, this library's RUNPATH names
// the ROCm installation's) and hands a handle of one to the other does not get an error, it aborts inside the runtime
// (std::bad_variant_access, round 5's diag.log).  helm_amd/_native.py makes the second copy impossible to get by import
// order; the guard below is for every other way in (a host that links its own HIP, dlopen by a plug-in): every entry point
// where a foreign handle crosses the ABI calls it and fails with both paths in the message instead.
// ------------------------------------------------------------------------------------------------------------------
namespace {
int collect_runtime(struct dl_phdr_info *info, size_t, void *data)
{
    const char *name = info->dlpi_name;
    if (!name || !*name) return 0;
    const char *base = strrchr(name, '/');
    base = base ? base + 1 : name;
    if (strncmp(base, "libamdhip64.so", 14) != 0) return 0;
    char real[PATH_MAX];
    std::string path = realpath(name, real) ? real : name;
    auto &paths = *static_cast<std::vector<std::string> *>(data);
    if (std::find(paths.begin(), paths.end(), path) == paths.end()) paths.push_back(path);
    return 0;
}
std::vector<std::string> mapped_runtimes()
{
    std::vector<std::string> paths;
    dl_iterate_phdr(collect_runtime, &paths);
    return paths;
}
} // namespace

int helm_hip_runtime_guard_(const char *where)
{
    const std::vector<std::string> paths = mapped_runtimes();
    if (paths.size() <= 1) return 0;
    std::string msg = std::string(where) + ": this process maps " + std::to_string(paths.size()) + " HIP runtimes (";
    for (size_t i = 0; i < paths.size(); i++) msg += (i ? ", " : "") + paths[i];
    msg += "); a stream or device pointer of one is not valid in the other - load libhelm_hip.so after the host's HIP runtime "
           "(helm_amd/_native.py does; INTEGRATION.md \"One HIP runtime per process\")";
    return helm_hip_fail_(HELM_ERR_STATE, msg);
}

extern "C" int helm_hip_runtime_copies(char *paths, size_t cap)
{
    const std::vector<std::string> found = mapped_runtimes();
    if (paths && cap) {
        std::string joined;
        for (size_t i = 0; i < found.size(); i++) joined += (i ? "\n" : "") + found[i];
        const size_t m = std::min(cap - 1, joined.size());
        memcpy(paths, joined.data(), m);
        paths[m] = 0;
    }
    return (int)found.size();
}

namespace {

struct Rccl {
    void *handle = nullptr;
    std::string origin, error;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;
std::once_flag g_once;

template <typename T> bool sym(T &fn, const char *name)
{
    fn = reinterpret_cast<T>(dlsym(g_rccl.handle, name));
    if (!fn) g_rccl.error = std::string("librccl: missing symbol ") + name;
    return fn != nullptr;
}

void bind_rccl()
{
    // HELM_RCCL_LIB names a library explicitly; otherwise: the copy this process already holds, the loader's
    // search path, the ROCm installation
    struct Try {
        const char *path;
        int flags;
        const char *what;
    };
    const char *forced = getenv("HELM_RCCL_LIB");
    // the RCCL of the distribution the process's HIP runtime comes from (PyTorch's wheel ships both side by side): the copy a
    // later `import torch` would map anyway, built against that runtime
    std::string beside, beside1;
    const std::vector<std::string> rt = mapped_runtimes();
    if (rt.size() == 1 && rt[0].find('/') != std::string::npos) {
        const std::string dir = rt[0].substr(0, rt[0].rfind('/') + 1);
        beside1 = dir + "librccl.so.1";
        beside = dir + "librccl.so";
    }
    const Try tries[] = {
        {forced, RTLD_NOW | RTLD_GLOBAL, "HELM_RCCL_LIB"},
        {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD, "already loaded by the process"},
        {"librccl.so", RTLD_NOW | RTLD_NOLOAD, "already loaded by the process"},
        {beside1.c_str(), RTLD_NOW | RTLD_GLOBAL, "next to the process's HIP runtime"},
        {beside.c_str(), RTLD_NOW | RTLD_GLOBAL, "next to the process's HIP runtime"},
        {"librccl.so.1", RTLD_NOW | RTLD_GLOBAL, "loader search path"},
        {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL, "/opt/rocm/lib"},
    };
    for (const Try &t : tries) {
        if (!t.path || !*t.path) continue;
        g_rccl.handle = dlopen(t.path, t.flags);
        if (g_rccl.handle) {
            g_rccl.origin = std::string(t.path) + " (" + t.what + ")";
            break;
        }
    }
    if (!g_rccl.handle) {
        g_rccl.error = "no RCCL library could be loaded (librccl.so.1; set HELM_RCCL_LIB)";
        return;
    }
    const bool ok = sym(g_rccl.GetVersion, "ncclGetVersion") && sym(g_rccl.GetUniqueId, "ncclGetUniqueId") &&
                    sym(g_rccl.CommInitRank, "ncclCommInitRank") && sym(g_rccl.CommDestroy, "ncclCommDestroy") &&
                    sym(g_rccl.CommCount, "ncclCommCount") && sym(g_rccl.CommCuDevice, "ncclCommCuDevice") &&
                    sym(g_rccl.CommUserRank, "ncclCommUserRank") && sym(g_rccl.AllGather, "ncclAllGather") &&
                    sym(g_rccl.AllReduce, "ncclAllReduce") && sym(g_rccl.GetErrorString, "ncclGetErrorString");
    if (!ok) g_rccl.handle = nullptr; // the mapping stays (it may be the process's own copy); we just do not use it
}

int need_rccl()
{
    std::call_once(g_once, bind_rccl);
    if (!g_rccl.handle) return helm_hip_fail_(HELM_ERR_STATE, g_rccl.error);
    return 0;
}

#define NCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        ncclResult_t r__ = (expr);                                                                              \
        if (r__ != ncclSuccess)                                                                                 \
            return helm_hip_fail_(HELM_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r__));        \
    } while (0)
#define HIPC_TRY(expr)                                                                                          \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) return helm_hip_fail_(HELM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

} // namespace

// helm_comm_create_in_process: the ranks are threads of this process.  A reusable barrier that can be broken (a rank that
// fails must not leave the others waiting) and the ranks' send pointers of the collective in flight.
struct InProcessGroup {
    int world = 0;
    double timeout_s = 600.0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    std::vector<const void *> sends;
    // 0 on success, 1 when the barrier is broken or the wait timed out (which breaks it for everybody)
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return 1;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return 0;
        }
        const bool ok = cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || broken; });
        if (!ok || broken) {
            broken = true;
            cv.notify_all();
            return 1;
        }
        return 0;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(m);
        broken = true;
        cv.notify_all();
    }
};

struct helm_comm {
    std::shared_ptr<InProcessGroup> group; // helm_comm_create_in_process
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t side = nullptr; // the host-side helpers' own stream
    double *scratch = nullptr;  // one double on the device for them
    int64_t collectives = 0, bytes_sent = 0;
    helm_comm_all_gather_fn transport = nullptr; // helm_comm_create_with_transport: the host's all-gather instead of RCCL
    void *transport_user = nullptr;
};

extern "C" {

int helm_comm_available(void)
{
    std::call_once(g_once, bind_rccl);
    return g_rccl.handle ? 1 : 0;
}

int helm_comm_precheck(int device_id)
{
    if (int rc = need_rccl()) return rc;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev)
        return helm_hip_fail_(HELM_ERR_NO_DEVICE, "helm_comm_precheck: no such device (" + std::to_string(device_id) + " of " +
                                                      std::to_string(n_dev) + ")");
    HIPC_TRY(hipSetDevice(device_id));
    return 0;
}

int helm_comm_get_unique_id(uint8_t id[HELM_COMM_ID_BYTES])
{
    if (!id) return helm_hip_fail_(HELM_ERR_INVALID, "null id");
    if (int rc = need_rccl()) return rc;
    static_assert(sizeof(ncclUniqueId) == HELM_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return 0;
}

int helm_comm_create(int device_id, const uint8_t id[HELM_COMM_ID_BYTES], int rank, int world, helm_comm **out)
{
    if (!id || !out) return helm_hip_fail_(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return helm_hip_fail_(HELM_ERR_INVALID, "helm_comm_create: bad rank / world");
    if (int rc = need_rccl()) return rc;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev)
        return helm_hip_fail_(HELM_ERR_NO_DEVICE, "helm_comm_create: no such device");
    HIPC_TRY(hipSetDevice(device_id));
    helm_comm *c = new (std::nothrow) helm_comm();
    if (!c) return helm_hip_fail_(HELM_ERR_OOM, "communicator");
    c->device = device_id;
    c->rank = rank;
    c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return helm_hip_fail_(HELM_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r) + " [" + g_rccl.origin + "]");
    }
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess || hipMalloc(&c->scratch, sizeof(double)) != hipSuccess) {
        helm_comm_destroy(c);
        return helm_hip_fail_(HELM_ERR_HIP, "helm_comm_create: side stream / scratch");
    }
    *out = c;
    return 0;
}

int helm_comm_create_with_transport(int device_id, int rank, int world, helm_comm_all_gather_fn all_gather, void *user,
                                    helm_comm **out)
{
    if (!all_gather || !out) return helm_hip_fail_(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return helm_hip_fail_(HELM_ERR_INVALID, "helm_comm_create_with_transport: bad rank / world");
    // the host's all-gather is handed this library's device pointers and stream
    if (int rc = helm_hip_runtime_guard_("helm_comm_create_with_transport")) return rc;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev)
        return helm_hip_fail_(HELM_ERR_NO_DEVICE, "helm_comm_create_with_transport: no such device");
    HIPC_TRY(hipSetDevice(device_id));
    helm_comm *c = new (std::nothrow) helm_comm();
    if (!c) return helm_hip_fail_(HELM_ERR_OOM, "communicator");
    c->device = device_id;
    c->rank = rank;
    c->world = world;
    c->transport = all_gather;
    c->transport_user = user;
    // the host-side helpers gather one double per rank
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&c->scratch, sizeof(double) * (size_t)world) != hipSuccess) {
        helm_comm_destroy(c);
        return helm_hip_fail_(HELM_ERR_HIP, "helm_comm_create_with_transport: side stream / scratch");
    }
    *out = c;
    return 0;
}

int helm_comm_create_in_process(const int *device_ids, int world, double timeout_s, helm_comm **out)
{
    if (!device_ids || !out || world < 1) return helm_hip_fail_(HELM_ERR_INVALID, "helm_comm_create_in_process: bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) return helm_hip_fail_(HELM_ERR_NO_DEVICE, "helm_comm_create_in_process: no device");
    for (int r = 0; r < world; r++) {
        out[r] = nullptr;
        if (device_ids[r] < 0 || device_ids[r] >= n_dev)
            return helm_hip_fail_(HELM_ERR_NO_DEVICE, "helm_comm_create_in_process: no such device");
    }
    // ranks on different devices copy from each other's buffers: peer access where the topology offers it (without it the
    // runtime stages the copies; an error here - no peer path, already enabled - is not one for the group)
    for (int r = 0; r < world; r++)
        for (int q = 0; q < world; q++)
            if (device_ids[r] != device_ids[q]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, device_ids[r], device_ids[q]) == hipSuccess && can &&
                    hipSetDevice(device_ids[r]) == hipSuccess)
                    (void)hipDeviceEnablePeerAccess(device_ids[q], 0);
            }
    (void)hipGetLastError(); // (hipErrorPeerAccessAlreadyEnabled is sticky until read)
    auto group = std::make_shared<InProcessGroup>();
    group->world = world;
    if (timeout_s > 0) group->timeout_s = timeout_s;
    group->sends.assign((size_t)world, nullptr);
    for (int r = 0; r < world; r++) {
        helm_comm *c = new (std::nothrow) helm_comm();
        bool ok = c != nullptr;
        if (ok) {
            c->device = device_ids[r];
            c->rank = r;
            c->world = world;
            c->group = group;
            ok = hipSetDevice(c->device) == hipSuccess && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
                 hipMalloc(&c->scratch, sizeof(double) * (size_t)world) == hipSuccess;
        }
        if (!ok) {
            if (c) helm_comm_destroy(c);
            for (int q = 0; q < r; q++) {
                helm_comm_destroy(out[q]);
                out[q] = nullptr;
            }
            return helm_hip_fail_(HELM_ERR_HIP, "helm_comm_create_in_process: side stream / scratch");
        }
        out[r] = c;
    }
    return 0;
}

int helm_comm_abort_group(helm_comm *c)
{
    if (!c) return helm_hip_fail_(HELM_ERR_INVALID, "null communicator");
    if (c->group) c->group->abort();
    return 0;
}

// the in-process all-gather: every rank thread calls it for the same collective; device-to-device copies between the
// ranks' buffers (peer copies when the ranks sit on different devices), a barrier on either side
static int in_process_all_gather(helm_comm *c, const void *send_dev, void *recv_dev, size_t bytes, hipStream_t stream)
{
    InProcessGroup &g = *c->group;
    auto broken = [&]() { return helm_hip_fail_(HELM_ERR_STATE, "in-process all-gather: another rank failed or did not arrive in time"); };
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamSynchronize(stream); // my chunk is complete
    if (e != hipSuccess) {
        g.abort();
        return helm_hip_fail_(HELM_ERR_HIP, std::string("in-process all-gather: ") + hipGetErrorString(e));
    }
    {
        std::lock_guard<std::mutex> lk(g.m);
        g.sends[(size_t)c->rank] = send_dev;
    }
    if (g.wait()) return broken(); // ... and so is everybody's
    for (int p = 0; p < c->world && e == hipSuccess; p++) {
        const void *src;
        {
            std::lock_guard<std::mutex> lk(g.m);
            src = g.sends[(size_t)p];
        }
        char *dst = static_cast<char *>(recv_dev) + (size_t)p * bytes;
        if (src != dst) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream); // (in place: nothing to move for my own slot)
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) {
        g.abort();
        return helm_hip_fail_(HELM_ERR_HIP, std::string("in-process all-gather: ") + hipGetErrorString(e));
    }
    if (g.wait()) return broken(); // nobody reuses its buffer before everybody has pulled
    return 0;
}

int helm_comm_destroy(helm_comm *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    (void)hipFree(c->scratch);
    ncclResult_t r = ncclSuccess;
    if (c->comm && g_rccl.handle) r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) return helm_hip_fail_(HELM_ERR_HIP, std::string("ncclCommDestroy: ") + g_rccl.GetErrorString(r));
    return 0;
}

int helm_comm_info(const helm_comm *c, int *rank, int *world, int *device, int *rccl_version)
{
    if (!c) return helm_hip_fail_(HELM_ERR_INVALID, "null communicator");
    int v = 0;
    if (c->transport || c->group) {
        if (rank) *rank = c->rank;
        if (world) *world = c->world;
        if (device) *device = c->device;
        if (rccl_version) *rccl_version = 0;
        return 0;
    }
    if (rank) NCCL_TRY(g_rccl.CommUserRank(c->comm, rank));
    if (world) NCCL_TRY(g_rccl.CommCount(c->comm, world));
    if (device) NCCL_TRY(g_rccl.CommCuDevice(c->comm, device));
    if (rccl_version) {
        NCCL_TRY(g_rccl.GetVersion(&v));
        *rccl_version = v;
    }
    return 0;
}

int helm_comm_stats(const helm_comm *c, int64_t *collectives, int64_t *bytes_sent)
{
    if (!c) return helm_hip_fail_(HELM_ERR_INVALID, "null communicator");
    if (collectives) *collectives = c->collectives;
    if (bytes_sent) *bytes_sent = c->bytes_sent;
    return 0;
}

int helm_comm_all_gather(helm_comm *c, const void *send_dev, void *recv_dev, size_t bytes_per_rank, void *hip_stream)
{
    if (!c || !send_dev || !recv_dev) return helm_hip_fail_(HELM_ERR_INVALID, "helm_comm_all_gather: null argument");
    if (bytes_per_rank == 0) return 0;
    if (c->group) {
        if (int rc = in_process_all_gather(c, send_dev, recv_dev, bytes_per_rank, static_cast<hipStream_t>(hip_stream))) return rc;
        c->collectives++;
        c->bytes_sent += (int64_t)bytes_per_rank;
        return 0;
    }
    if (c->transport) {
        if (int rc = c->transport(c->transport_user, send_dev, recv_dev, bytes_per_rank, hip_stream))
            return helm_hip_fail_(HELM_ERR_STATE, "helm_comm_all_gather: the host's transport failed (" + std::to_string(rc) + ")");
        c->collectives++;
        c->bytes_sent += (int64_t)bytes_per_rank;
        return 0;
    }
    HIPC_TRY(hipSetDevice(c->device)); // a host thread that never selected a device must not enqueue on device 0
    // words where the size allows it (rows of u32 / u64 always do), bytes otherwise
    if (bytes_per_rank % 4 == 0)
        NCCL_TRY(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank / 4, ncclUint32, c->comm, static_cast<hipStream_t>(hip_stream)));
    else
        NCCL_TRY(g_rccl.AllGather(send_dev, recv_dev, bytes_per_rank, ncclUint8, c->comm, static_cast<hipStream_t>(hip_stream)));
    c->collectives++;
    c->bytes_sent += (int64_t)bytes_per_rank;
    return 0;
}

int helm_comm_all_reduce_f64(helm_comm *c, double *value, int op)
{
    if (!c || !value || (op != 0 && op != 1)) return helm_hip_fail_(HELM_ERR_INVALID, "helm_comm_all_reduce_f64: bad argument");
    HIPC_TRY(hipSetDevice(c->device));
    if (c->transport || c->group) { // one double per rank through the host's (or the in-process) all-gather, reduced here
        HIPC_TRY(hipMemcpyAsync(c->scratch + c->rank, value, sizeof(double), hipMemcpyHostToDevice, c->side));
        if (c->group) {
            if (int rc = in_process_all_gather(c, c->scratch + c->rank, c->scratch, sizeof(double), c->side)) return rc;
        } else if (int rc = c->transport(c->transport_user, c->scratch + c->rank, c->scratch, sizeof(double), c->side))
            return helm_hip_fail_(HELM_ERR_STATE, "helm_comm_all_reduce_f64: the host's transport failed (" + std::to_string(rc) + ")");
        std::vector<double> all((size_t)c->world);
        HIPC_TRY(hipMemcpyAsync(all.data(), c->scratch, sizeof(double) * all.size(), hipMemcpyDeviceToHost, c->side));
        HIPC_TRY(hipStreamSynchronize(c->side));
        double r = all[0];
        for (size_t i = 1; i < all.size(); i++) r = op == 0 ? r + all[i] : std::max(r, all[i]);
        *value = r;
        c->collectives++;
        c->bytes_sent += (int64_t)sizeof(double);
        return 0;
    }
    HIPC_TRY(hipMemcpyAsync(c->scratch, value, sizeof(double), hipMemcpyHostToDevice, c->side));
    NCCL_TRY(g_rccl.AllReduce(c->scratch, c->scratch, 1, ncclFloat64, op == 0 ? ncclSum : ncclMax, c->comm, c->side));
    HIPC_TRY(hipMemcpyAsync(value, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->side));
    HIPC_TRY(hipStreamSynchronize(c->side));
    c->collectives++;
    c->bytes_sent += (int64_t)sizeof(double);
    return 0;
}

int helm_comm_barrier(helm_comm *c)
{
    double one = 1.0;
    return helm_comm_all_reduce_f64(c, &one, 0);
}

} // extern "C"
