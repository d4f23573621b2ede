// Test double for the communication library: the handful of RCCL entry points the halo exchange of
// libmagnet_hip.so calls, carried by POSIX shared memory between processes of ONE host.  RCCL itself refuses
// two ranks on the same GPU, so the multi-rank logic of mh_exchange_* (who sends which rows to whom, where
// they land, the stream ordering around the transfers) could not otherwise be run on a one-GPU machine; with
// MAGNET_RCCL_LIBRARY pointing here it runs with 2-4 real processes sharing the device.
//
// Semantics: sends and receives queue up between ncclGroupStart and ncclGroupEnd; ncclGroupEnd waits for the
// stream they were posted on, then performs all sends (device -> mailbox) and then all receives (mailbox ->
// device), blocking the host.  One mailbox per ordered pair of ranks, one message at a time, fixed capacity
// (MOCK_RCCL_SLOT_BYTES, default 8 MiB): test sizes only.  Results equal the real library's; timing does not.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

namespace {

struct Slot {
    std::atomic<uint64_t> written;   // messages put in
    std::atomic<uint64_t> taken;     // messages taken out
    uint64_t bytes;
    char pad[40];
};

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> left;
};

struct Pending {
    bool send;
    void* ptr;
    size_t bytes;
    int peer;
    hipStream_t stream;
};

size_t slot_bytes() {
    const char* e = std::getenv("MOCK_RCCL_SLOT_BYTES");
    return e && *e ? size_t(std::atoll(e)) : (size_t(8) << 20);
}

constexpr double kTimeoutSeconds = 60.0;

template <class Cond>
bool wait_for(Cond c) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!c()) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) return false;
    }
    return true;
}

thread_local int t_group_depth = 0;
thread_local std::vector<std::pair<struct ncclComm*, Pending>> t_queue;

}  // namespace

struct ncclComm {
    int rank = 0, world = 0;
    size_t cap = 0, map_bytes = 0;
    char* base = nullptr;
    Header* header() const { return reinterpret_cast<Header*>(base); }
    Slot* slot(int src, int dst) const { return reinterpret_cast<Slot*>(base + 64) + (src * world + dst); }
    char* payload(int src, int dst) const { return base + 64 + sizeof(Slot) * world * world + cap * (size_t(src) * world + dst); }
};

namespace {

ncclResult_t run(ncclComm* c, const Pending& p) {
    if (p.bytes > c->cap) {
        fprintf(stderr, "mock_rccl: a message of %zu bytes exceeds the mailbox (%zu); raise MOCK_RCCL_SLOT_BYTES\n", p.bytes, c->cap);
        return ncclInvalidArgument;
    }
    if (p.send) {
        Slot* s = c->slot(c->rank, p.peer);
        if (!wait_for([&] { return s->written.load() == s->taken.load(); })) return ncclSystemError;
        if (hipMemcpy(c->payload(c->rank, p.peer), p.ptr, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        s->bytes = p.bytes;
        s->written.fetch_add(1);
    } else {
        Slot* s = c->slot(p.peer, c->rank);
        if (!wait_for([&] { return s->written.load() > s->taken.load(); })) return ncclSystemError;
        if (s->bytes != p.bytes) {
            fprintf(stderr, "mock_rccl: rank %d expects %zu bytes from rank %d, which sent %llu\n", c->rank, p.bytes, p.peer,
                    (unsigned long long)s->bytes);
            return ncclInvalidArgument;
        }
        if (hipMemcpy(p.ptr, c->payload(p.peer, c->rank), p.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        s->taken.fetch_add(1);
    }
    return ncclSuccess;
}

ncclResult_t flush() {
    // MOCK_RCCL_STALL: never complete (a test of the caller's own time limit)
    if (const char* e = std::getenv("MOCK_RCCL_STALL"))
        if (*e) std::this_thread::sleep_for(std::chrono::hours(1));
    std::vector<std::pair<ncclComm*, Pending>> q;
    q.swap(t_queue);
    for (auto& e : q)
        if (hipStreamSynchronize(e.second.stream) != hipSuccess) return ncclUnhandledCudaError;
    for (int pass = 0; pass < 2; ++pass)            // all sends, then all receives
        for (auto& e : q)
            if (e.second.send == (pass == 0)) {
                ncclResult_t r = run(e.first, e.second);
                if (r != ncclSuccess) return r;
            }
    return ncclSuccess;
}

ncclResult_t post(ncclComm* c, Pending p) {
    if (!c || p.peer < 0 || p.peer >= c->world) return ncclInvalidArgument;
    t_queue.emplace_back(c, p);
    return t_group_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<int> counter{0};
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/magnet_mock_rccl_%d_%d", int(getpid()), counter.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = world; c->cap = slot_bytes();
    c->map_bytes = 64 + sizeof(Slot) * world * world + c->cap * size_t(world) * world;
    int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, off_t(c->map_bytes)) != 0) { delete c; return ncclSystemError; }
    void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->base = static_cast<char*>(m);                 // a new shared-memory object reads as zeros: counters start at 0
    c->header()->arrived.fetch_add(1);
    if (!wait_for([&] { return c->header()->arrived.load() >= world; })) return ncclSystemError;
    if (c->header()->left.fetch_add(1) + 1 == world) shm_unlink(id.internal);   // everyone has mapped it
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    ncclComm* c = reinterpret_cast<ncclComm*>(comm);
    if (c) {
        if (c->base) munmap(c->base, c->map_bytes);
        delete c;
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++t_group_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (t_group_depth <= 0) return ncclInvalidUsage;
    return --t_group_depth == 0 ? flush() : ncclSuccess;
}

static size_t width(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 4;
    }
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(reinterpret_cast<ncclComm*>(comm), Pending{true, const_cast<void*>(buf), count * width(type), peer, stream});
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(reinterpret_cast<ncclComm*>(comm), Pending{false, buf, count * width(type), peer, stream});
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    ncclComm* c = reinterpret_cast<ncclComm*>(comm);
    if (!c) return ncclInvalidArgument;
    const size_t bytes = count * width(type);
    ncclGroupStart();
    for (int p = 0; p < c->world; ++p) {
        post(c, Pending{true, const_cast<void*>(send), bytes, p, stream});
        post(c, Pending{false, static_cast<char*>(recv) + bytes * p, bytes, p, stream});
    }
    return ncclGroupEnd();
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclInvalidArgument: return "invalid argument (mock RCCL)";
        case ncclInvalidUsage: return "invalid usage (mock RCCL)";
        case ncclSystemError: return "system error or time-out (mock RCCL)";
        default: return "error (mock RCCL)";
    }
}

}  // extern "C"
