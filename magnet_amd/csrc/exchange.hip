// Halo exchange of a row-partitioned run, posted from native code: packing launches, one grouped RCCL
// send/receive to every peer on the exchange's own stream, and the stream ordering on both sides of it.
//
// The reference runs on one device (src/sparse_Magnet.py:23-24) and has no counterpart.  The Python host
// (magnet_amd/dist.py: RowPartition) can post the same transfers through torch.distributed; this file is the
// variant without the interpreter between the packing kernels and the communication library: one call posts an
// order's whole exchange, one call makes the compute stream wait for it.
//
// RCCL is looked up at run time (dlopen), not linked: a process that never creates an exchange never loads it,
// and inside a PyTorch process the copy PyTorch already loaded is the one used (same SONAME), so the process
// holds one communication library, not two.  MAGNET_RCCL_LIBRARY names another library to load instead.
#include <dlfcn.h>

#include <cstdlib>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    char why[256] = {0};
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void* open_rccl() {
    if (const char* path = std::getenv("MAGNET_RCCL_LIBRARY")) {
        if (*path) return dlopen(path, RTLD_NOW | RTLD_LOCAL);
    }
    // a copy that is already in the process first (PyTorch brings its own), then the system's
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)) return h;
    for (const char* n : names)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) return h;
    return dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
}

void load_rccl() {
    void* h = open_rccl();
    if (!h) {
        const char* e = dlerror();
        snprintf(g_rccl.why, sizeof(g_rccl.why), "the RCCL library could not be loaded: %s", e ? e : "unknown reason");
        return;
    }
    bool all = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) {
            all = false;
            snprintf(g_rccl.why, sizeof(g_rccl.why), "the RCCL library lacks %s", name);
        }
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    g_rccl.ok = all;
}

int need_rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) return mh::fail(MH_ERR_COMM, "%s", g_rccl.why);
    return MH_OK;
}

int comm_fail(ncclResult_t r, const char* what) {
    return mh::fail(MH_ERR_COMM, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

#define MH_RCCL(call)                                        \
    do {                                                     \
        ncclResult_t r__ = (call);                           \
        if (r__ != ncclSuccess) return comm_fail(r__, #call); \
    } while (0)

}  // namespace

struct mh_exchange {
    ncclComm_t comm = nullptr;
    int32_t rank = 0, world = 0;
    int device = 0;
    int64_t n_local = 0;
    std::vector<const int32_t*> give_rows;   // device pointers, one per peer
    std::vector<int64_t> give_count, recv_ptr;
    hipStream_t stream = nullptr;            // the exchange's own stream: transfers overlap the compute stream
    hipEvent_t ready = nullptr, done = nullptr;
    float* stage = nullptr;                  // packed rows of the peers that take a subset of the block
    size_t stage_floats = 0;
    bool in_flight = false;
};

extern "C" int mh_exchange_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int mh_exchange_new_id(void* id, size_t id_bytes) {
    MH_REQUIRE(id != nullptr && id_bytes >= NCCL_UNIQUE_ID_BYTES, "mh_exchange_new_id: id must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
    if (int rc = need_rccl()) return rc;
    ncclUniqueId u;
    MH_RCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
    return MH_OK;
}

extern "C" int mh_exchange_create(const void* id, size_t id_bytes, int32_t rank, int32_t world,
                                  const int32_t* const* give_rows, const int64_t* give_count,
                                  const int64_t* recv_ptr, int64_t n_local, mh_exchange** out) {
    MH_REQUIRE(out != nullptr, "mh_exchange_create: out is null");
    *out = nullptr;
    MH_REQUIRE(id != nullptr && id_bytes >= NCCL_UNIQUE_ID_BYTES, "mh_exchange_create: id must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
    MH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "mh_exchange_create: rank %d of %d", rank, world);
    MH_REQUIRE(give_rows && give_count && recv_ptr && n_local >= 0, "mh_exchange_create: null plan");
    MH_REQUIRE(recv_ptr[0] == 0, "mh_exchange_create: recv_ptr must start at 0");
    for (int p = 0; p < world; ++p) {
        MH_REQUIRE(give_count[p] >= 0 && give_count[p] <= n_local, "mh_exchange_create: give_count[%d] = %lld outside [0, n_local]",
                   p, (long long)give_count[p]);
        MH_REQUIRE(give_count[p] == 0 || give_rows[p] != nullptr, "mh_exchange_create: give_rows[%d] is null", p);
        MH_REQUIRE(recv_ptr[p + 1] >= recv_ptr[p], "mh_exchange_create: recv_ptr must ascend");
    }
    if (int rc = need_rccl()) return rc;
    mh_exchange* x = new mh_exchange();
    x->rank = rank; x->world = world; x->n_local = n_local;
    x->give_rows.assign(give_rows, give_rows + world);
    x->give_count.assign(give_count, give_count + world);
    x->recv_ptr.assign(recv_ptr, recv_ptr + world + 1);
    auto cleanup = [&](int rc) { mh_exchange_destroy(x); return rc; };
    hipError_t e;
    if ((e = hipGetDevice(&x->device)) != hipSuccess) return cleanup(mh::hip_fail(e, "hipGetDevice"));
    if ((e = hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking)) != hipSuccess) return cleanup(mh::hip_fail(e, "hipStreamCreate"));
    if ((e = hipEventCreateWithFlags(&x->ready, hipEventDisableTiming)) != hipSuccess) return cleanup(mh::hip_fail(e, "hipEventCreate"));
    if ((e = hipEventCreateWithFlags(&x->done, hipEventDisableTiming)) != hipSuccess) return cleanup(mh::hip_fail(e, "hipEventCreate"));
    ncclUniqueId u;
    memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = g_rccl.CommInitRank(&x->comm, world, u, rank);
    if (r != ncclSuccess) {
        x->comm = nullptr;
        return cleanup(comm_fail(r, "ncclCommInitRank"));
    }
    *out = x;
    return MH_OK;
}

extern "C" int mh_exchange_destroy(mh_exchange* x) {
    if (!x) return MH_OK;
    int dev = -1;
    (void)hipGetDevice(&dev);
    (void)hipSetDevice(x->device);
    if (x->stream) (void)hipStreamSynchronize(x->stream);
    if (x->comm) g_rccl.CommDestroy(x->comm);
    if (x->stage) (void)hipFree(x->stage);
    if (x->ready) (void)hipEventDestroy(x->ready);
    if (x->done) (void)hipEventDestroy(x->done);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    if (dev >= 0) (void)hipSetDevice(dev);
    delete x;
    return MH_OK;
}

extern "C" int mh_exchange_begin(mh_exchange* x, const float* own, int64_t ld, int64_t im, float* halo, int32_t F,
                                 void* compute_stream) {
    MH_REQUIRE(x != nullptr && x->comm != nullptr, "mh_exchange_begin: no exchange");
    MH_REQUIRE(F > 0, "mh_exchange_begin: F = %d", F);
    MH_REQUIRE(own != nullptr || x->n_local == 0, "mh_exchange_begin: own is null");
    MH_REQUIRE(halo != nullptr || x->recv_ptr[x->world] == 0, "mh_exchange_begin: halo is null");
    hipStream_t cs = static_cast<hipStream_t>(compute_stream);
    const int64_t row = 2 * int64_t(F);
    const bool contiguous = (im == F && ld == row);
    // which peers are served from a packed copy, and how much staging that takes
    size_t need = 0;
    for (int p = 0; p < x->world; ++p) {
        const bool in_place = contiguous && x->give_count[p] == x->n_local;
        if (x->give_count[p] > 0 && !in_place) need += size_t(x->give_count[p]) * size_t(row);
    }
    // the staging area and the events belong to one exchange at a time: a new one starts after the last has
    // been consumed by the compute stream (normally it has been: mh_exchange_end)
    if (x->in_flight) MH_HIP(hipStreamWaitEvent(cs, x->done, 0));
    if (need > x->stage_floats) {
        MH_HIP(hipStreamSynchronize(x->stream));
        if (x->stage) MH_HIP(hipFree(x->stage));
        x->stage = nullptr; x->stage_floats = 0;
        MH_HIP(hipMalloc(reinterpret_cast<void**>(&x->stage), need * sizeof(float)));
        x->stage_floats = need;
    }
    // packing runs on the compute stream, ahead of whatever the caller launches next: it is the head of the exchange
    std::vector<const float*> src(x->world, nullptr);
    size_t off = 0;
    for (int p = 0; p < x->world; ++p) {
        const int64_t cnt = x->give_count[p];
        if (cnt == 0) continue;
        if (contiguous && cnt == x->n_local) {
            src[p] = own;                   // ascending and complete: the block itself
        } else {
            float* dst = x->stage + off;
            if (int rc = mh_gather_rows(own, ld, im, x->give_rows[p], cnt, dst, F, cs)) return rc;
            src[p] = dst;
            off += size_t(cnt) * size_t(row);
        }
    }
    MH_HIP(hipEventRecord(x->ready, cs));
    MH_HIP(hipStreamWaitEvent(x->stream, x->ready, 0));
    MH_RCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int step = 0; step < x->world && r == ncclSuccess; ++step) {
        // every rank walks its peers starting from its own position, so that the pairs are spread over the links
        const int to = (x->rank + step) % x->world, from = (x->rank - step + x->world) % x->world;
        if (x->give_count[to] > 0)
            r = g_rccl.Send(src[to], size_t(x->give_count[to]) * size_t(row), ncclFloat, to, x->comm, x->stream);
        const int64_t n_from = x->recv_ptr[from + 1] - x->recv_ptr[from];
        if (r == ncclSuccess && n_from > 0)
            r = g_rccl.Recv(halo + x->recv_ptr[from] * row, size_t(n_from) * size_t(row), ncclFloat, from, x->comm, x->stream);
    }
    ncclResult_t r_end = g_rccl.GroupEnd();
    if (r != ncclSuccess) return comm_fail(r, "ncclSend / ncclRecv");
    if (r_end != ncclSuccess) return comm_fail(r_end, "ncclGroupEnd");
    MH_HIP(hipEventRecord(x->done, x->stream));
    x->in_flight = true;
    return MH_OK;
}

extern "C" int mh_exchange_end(mh_exchange* x, void* compute_stream) {
    MH_REQUIRE(x != nullptr, "mh_exchange_end: no exchange");
    if (!x->in_flight) return MH_OK;
    MH_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(compute_stream), x->done, 0));
    x->in_flight = false;
    return MH_OK;
}

extern "C" int mh_exchange_all_gather(mh_exchange* x, const float* mine, float* all, int64_t floats_per_rank,
                                      void* compute_stream) {
    MH_REQUIRE(x != nullptr && x->comm != nullptr, "mh_exchange_all_gather: no exchange");
    MH_REQUIRE(floats_per_rank >= 0 && (floats_per_rank == 0 || (mine && all)), "mh_exchange_all_gather: null buffer");
    hipStream_t cs = static_cast<hipStream_t>(compute_stream);
    if (x->in_flight) MH_HIP(hipStreamWaitEvent(cs, x->done, 0));
    MH_HIP(hipEventRecord(x->ready, cs));
    MH_HIP(hipStreamWaitEvent(x->stream, x->ready, 0));
    if (floats_per_rank > 0)
        MH_RCCL(g_rccl.AllGather(mine, all, size_t(floats_per_rank), ncclFloat, x->comm, x->stream));
    MH_HIP(hipEventRecord(x->done, x->stream));
    x->in_flight = true;
    return MH_OK;
}
