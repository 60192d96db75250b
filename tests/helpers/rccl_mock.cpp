// TEST INFRASTRUCTURE — a stand-in for librccl that lets SEVERAL PROCESSES ON ONE GPU run the library's
// one-process-per-GPU path (murbhip_create_rank: rank > 0, world > 1), which real RCCL refuses ("duplicate
// GPU").  Same entry points and argument meaning as the ncclXxx functions libmurbhip binds at run time
// (csrc/murb_rccl.h); the collectives go through a POSIX shared-memory segment and a process-shared
// barrier, synchronously: each call drains the stream it is given, stages through host memory, and returns
// when the result is in place.  Loaded only when MURBHIP_RCCL_LIBRARY points at it (tests/test_rank_mode_mock.py).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
constexpr size_t kCapacity = 256ul << 20;   // bytes of staging area
struct Segment {
    pthread_barrier_t barrier;
    std::atomic<int> ready;
    int nranks;
    char data[1];
};
struct LocalGroup;   // one process driving several "devices" (ncclCommInitAll)
struct Comm {
    Segment* seg = nullptr;
    int rank = 0, nranks = 1;
    std::string name;
    LocalGroup* local = nullptr;
    int device = 0;
};
struct PendingOp {
    int kind;   // 0 all-gather, 1 reduce-scatter
    const void* send; void* recv; size_t count; Comm* comm; hipStream_t stream;
};
struct LocalGroup {
    int nranks = 0, alive = 0;
    std::vector<PendingOp> pending;
};
int group_depth = 0;
std::vector<LocalGroup*> touched;   // groups with pending work inside the current ncclGroupStart/End
struct Id { char bytes[128]; };
std::string segment_name(const Id& id)
{
    char buf[64];
    unsigned long long h = 0;
    std::memcpy(&h, id.bytes + 8, sizeof h);
    std::snprintf(buf, sizeof buf, "/murbmock_%016llx", h);
    return buf;
}
constexpr int kOk = 0, kInvalid = 4, kSystem = 2;
}  // namespace

extern "C" {

int ncclGetUniqueId(Id* id)
{
    std::memset(id, 0, sizeof *id);
    std::memcpy(id->bytes, "MOCKRCCL", 8);
    FILE* f = std::fopen("/dev/urandom", "rb");
    if (!f || std::fread(id->bytes + 8, 1, 16, f) != 16) { if (f) std::fclose(f); return kSystem; }
    std::fclose(f);
    return kOk;
}

int ncclCommInitRank(void** out, int nranks, Id id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks || std::memcmp(id.bytes, "MOCKRCCL", 8) != 0) return kInvalid;
    Comm* c = new Comm;
    c->rank = rank; c->nranks = nranks; c->name = segment_name(id);
    const size_t bytes = sizeof(Segment) + kCapacity;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { delete c; return kSystem; }
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) {   // up to ~20 s for rank 0 to create it
            fd = shm_open(c->name.c_str(), O_RDWR, 0600);
            if (fd < 0) usleep(1000);
        }
        if (fd < 0) { delete c; return kSystem; }
        struct stat st;
        for (int tries = 0; tries < 20000; ++tries) {   // rank 0 may not have sized it yet
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
            usleep(1000);
        }
        if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { close(fd); delete c; return kSystem; }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return kSystem; }
    c->seg = static_cast<Segment*>(p);
    if (rank == 0) {
        pthread_barrierattr_t attr;
        pthread_barrierattr_init(&attr);
        pthread_barrierattr_setpshared(&attr, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->seg->barrier, &attr, (unsigned)nranks);
        c->seg->nranks = nranks;
        c->seg->ready.store(1);
    } else {
        for (int tries = 0; tries < 20000 && c->seg->ready.load() != 1; ++tries) usleep(1000);
        if (c->seg->ready.load() != 1 || c->seg->nranks != nranks) { delete c; return kSystem; }
    }
    pthread_barrier_wait(&c->seg->barrier);
    *out = c;
    return kOk;
}

// one process, several "devices" (the same ordinal may repeat): the collectives are executed at ncclGroupEnd,
// when every rank's call has been recorded
int ncclCommInitAll(void** out, int ndev, const int* devices)
{
    if (!out || ndev < 1 || !devices) return kInvalid;
    LocalGroup* g = new LocalGroup;
    g->nranks = g->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        Comm* c = new Comm;
        c->rank = r; c->nranks = ndev; c->local = g; c->device = devices[r];
        out[r] = c;
    }
    return kOk;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return kOk;
    if (c->local) {
        if (--c->local->alive == 0) delete c->local;
        delete c;
        return kOk;
    }
    pthread_barrier_wait(&c->seg->barrier);
    munmap(c->seg, sizeof(Segment) + kCapacity);
    if (c->rank == 0) shm_unlink(c->name.c_str());
    delete c;
    return kOk;
}

// every rank contributes `count` floats; recv gets nranks * count, rank r's block at r * count
int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    const size_t bytes = count * 4;
    if (c && c->local) {
        if (dtype != 7 || group_depth == 0) return kInvalid;
        c->local->pending.push_back(PendingOp{0, send, recv, count, c, stream});
        touched.push_back(c->local);
        return kOk;
    }
    if (!c || dtype != 7 || bytes * c->nranks > kCapacity) return kInvalid;
    if (hipStreamSynchronize(stream) != hipSuccess) return kSystem;
    if (hipMemcpy(c->seg->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kSystem;
    pthread_barrier_wait(&c->seg->barrier);
    if (hipMemcpy(recv, c->seg->data, bytes * c->nranks, hipMemcpyHostToDevice) != hipSuccess) return kSystem;
    pthread_barrier_wait(&c->seg->barrier);
    return kOk;
}

// every rank contributes nranks * recvcount floats; rank r receives the sum over ranks of their block r
int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (c && c->local) {
        if (dtype != 7 || op != 0 || group_depth == 0) return kInvalid;
        c->local->pending.push_back(PendingOp{1, send, recv, recvcount, c, stream});
        touched.push_back(c->local);
        return kOk;
    }
    const size_t block = recvcount * 4, mine = block * c->nranks;
    if (!c || dtype != 7 || op != 0 || mine * c->nranks > kCapacity) return kInvalid;
    if (hipStreamSynchronize(stream) != hipSuccess) return kSystem;
    if (hipMemcpy(c->seg->data + (size_t)c->rank * mine, send, mine, hipMemcpyDeviceToHost) != hipSuccess) return kSystem;
    pthread_barrier_wait(&c->seg->barrier);
    std::vector<float> sum(recvcount, 0.f);
    for (int r = 0; r < c->nranks; ++r) {
        const float* src = reinterpret_cast<const float*>(c->seg->data + (size_t)r * mine + (size_t)c->rank * block);
        for (size_t k = 0; k < recvcount; ++k) sum[k] += src[k];
    }
    if (hipMemcpy(recv, sum.data(), block, hipMemcpyHostToDevice) != hipSuccess) return kSystem;
    pthread_barrier_wait(&c->seg->barrier);
    return kOk;
}

static int run_local(LocalGroup* g)
{
    if (g->pending.empty()) return kOk;
    if ((int)g->pending.size() != g->nranks) return kInvalid;   // every rank must have made the same single call
    const int kind = g->pending[0].kind;
    const size_t count = g->pending[0].count, n = (size_t)g->nranks;
    std::vector<std::vector<float>> host(n);
    for (const PendingOp& op : g->pending) {
        if (op.kind != kind || op.count != count) return kInvalid;
        const size_t floats = kind == 0 ? count : count * n;
        host[op.comm->rank].resize(floats);
        if (hipSetDevice(op.comm->device) != hipSuccess || hipStreamSynchronize(op.stream) != hipSuccess ||
            hipMemcpy(host[op.comm->rank].data(), op.send, floats * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return kSystem;
    }
    for (const PendingOp& op : g->pending) {
        std::vector<float> out(kind == 0 ? count * n : count, 0.f);
        if (kind == 0)
            for (size_t r = 0; r < n; ++r) std::copy(host[r].begin(), host[r].end(), out.begin() + r * count);
        else
            for (size_t r = 0; r < n; ++r)
                for (size_t k = 0; k < count; ++k) out[k] += host[r][(size_t)op.comm->rank * count + k];
        if (hipSetDevice(op.comm->device) != hipSuccess ||
            hipMemcpy(op.recv, out.data(), out.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            return kSystem;
    }
    g->pending.clear();
    return kOk;
}

int ncclGroupStart() { ++group_depth; return kOk; }
int ncclGroupEnd()
{
    if (group_depth > 0 && --group_depth == 0) {
        int rc = kOk;
        for (LocalGroup* g : touched) { const int r = run_local(g); if (r != kOk) rc = r; }
        touched.clear();
        return rc;
    }
    return kOk;
}
const char* ncclGetErrorString(int code) { return code == kOk ? "no error" : code == kInvalid ? "mock: invalid argument" : "mock: system error"; }

}  // extern "C"
