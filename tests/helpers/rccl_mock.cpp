// TEST INFRASTRUCTURE — a stand-in for librccl that lets SEVERAL PROCESSES ON ONE GPU run the library's
// one-process-per-GPU path (murbhip_create_rank: rank > 0, world > 1), which real RCCL refuses ("duplicate
// GPU").  Same entry points and argument meaning as the ncclXxx functions libmurbhip binds at run time
// (csrc/murb_rccl.h); the data moves through host memory (a POSIX shared-memory segment between processes).
//
// Two modes (environment variable MURB_MOCK_MODE):
//   async (default)  like real RCCL, a call only ENQUEUES work on the stream it is given and returns: a
//                    device->host copy of the caller's contribution, a host function that meets the other
//                    ranks at a process-shared barrier (and, for the reduce-scatter, adds the blocks up), a
//                    host->device copy of the result, a second barrier (the staging area is free again).
//                    Nothing is drained: a consumer that forgets to wait for the collective's stream reads
//                    stale data and the comparison with the single-GPU run fails — the property a synchronous
//                    stand-in cannot check.
//   solo             (MURB_MOCK_SOLO=1, tools/solo_rank.py) ONE rank of a W-rank communicator runs without its peers: nobody
//                    is waited for, the all-gather leaves the caller's own slice where it is and the reduce-scatter
//                    copies the caller's own block of its own contribution device-to-device.  Results are meaningless;
//                    what it gives is the step time of one real rank process (its own three streams) without the wire.
//   sync             each call drains its stream, stages through host memory and returns when the result
//                    is in place (the behaviour of round 1).
// Loaded only when MURBHIP_RCCL_LIBRARY points at it (tests/test_rank_mode_mock.py).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
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
    bool pinned = false;
};
struct PendingOp {
    int kind;   // 0 all-gather, 1 reduce-scatter
    const void* send; void* recv; size_t count; Comm* comm; hipStream_t stream;
};
// one process, several ranks: a ring of pinned staging buffers, each guarded by the events of its last use
struct Staging {
    float* host = nullptr;
    size_t floats = 0;
    std::vector<hipEvent_t> done;   // one per rank: recorded after that rank's host->device copy
    bool used = false;
};
struct LocalGroup {
    int nranks = 0, alive = 0;
    std::vector<PendingOp> pending;
    Staging ring[4];
    int next = 0;
    std::vector<hipEvent_t> copied;   // per rank: its contribution has reached the staging buffer
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

bool solo_mode()
{
    static const bool a = [] { const char* m = std::getenv("MURB_MOCK_SOLO"); return m && *m && *m != '0'; }();
    return a;
}

bool async_mode()
{
    static const bool a = [] {
        const char* m = std::getenv("MURB_MOCK_MODE");
        return !(m && std::string(m) == "sync");
    }();
    return a;
}

// ---- host functions (hipLaunchHostFunc): no HIP call inside
void host_barrier(void* p) { pthread_barrier_wait(&static_cast<Comm*>(p)->seg->barrier); }

struct ReduceJob { Comm* c; size_t recvcount; };
// result area of rank r: behind the nranks contributions
inline float* reduce_result(Comm* c, size_t recvcount)
{
    const size_t block = recvcount * 4, mine = block * c->nranks;
    return reinterpret_cast<float*>(c->seg->data + mine * c->nranks + (size_t)c->rank * block);
}
void host_reduce(void* p)
{
    ReduceJob* j = static_cast<ReduceJob*>(p);
    Comm* c = j->c;
    const size_t block = j->recvcount * 4, mine = block * c->nranks;
    float* out = reduce_result(c, j->recvcount);
    for (size_t k = 0; k < j->recvcount; ++k) out[k] = 0.f;
    for (int r = 0; r < c->nranks; ++r) {   // rank order: reproducible
        const float* src = reinterpret_cast<const float*>(c->seg->data + (size_t)r * mine + (size_t)c->rank * block);
        for (size_t k = 0; k < j->recvcount; ++k) out[k] += src[k];
    }
    delete j;
}

struct LocalReduceJob { const float* staged; float* out; size_t count; int nranks, rank; };
void host_local_reduce(void* p)
{
    LocalReduceJob* j = static_cast<LocalReduceJob*>(p);
    for (size_t k = 0; k < j->count; ++k) j->out[k] = 0.f;
    for (int r = 0; r < j->nranks; ++r) {
        const float* src = j->staged + ((size_t)r * j->nranks + (size_t)j->rank) * j->count;
        for (size_t k = 0; k < j->count; ++k) j->out[k] += src[k];
    }
    delete j;
}
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
    if (solo_mode()) { *out = c; return kOk; }   // no peers, no shared segment
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
    // pinned: the copies of the async mode are then truly asynchronous (unpinned they still run in stream order)
    if (async_mode()) c->pinned = hipHostRegister(c->seg->data, kCapacity, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
    pthread_barrier_wait(&c->seg->barrier);
    *out = c;
    return kOk;
}

// one process, several "devices" (the same ordinal may repeat): the collectives are issued at ncclGroupEnd,
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
        if (--c->local->alive == 0) {
            (void)hipDeviceSynchronize();
            for (Staging& s : c->local->ring) {
                if (s.host) (void)hipHostFree(s.host);
                for (hipEvent_t e : s.done) (void)hipEventDestroy(e);
            }
            for (hipEvent_t e : c->local->copied) (void)hipEventDestroy(e);
            delete c->local;
        }
        delete c;
        return kOk;
    }
    (void)hipDeviceSynchronize();   // async mode: host functions of this communicator may still be queued
    if (!c->seg) { delete c; return kOk; }   // solo mode
    pthread_barrier_wait(&c->seg->barrier);
    if (c->pinned) (void)hipHostUnregister(c->seg->data);
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
    if (solo_mode()) return kOk;   // in place: the caller's slice is where it belongs, the others never arrive
    if (async_mode()) {
        if (hipMemcpyAsync(c->seg->data + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return kSystem;
        if (hipLaunchHostFunc(stream, host_barrier, c) != hipSuccess) return kSystem;
        if (hipMemcpyAsync(recv, c->seg->data, bytes * c->nranks, hipMemcpyHostToDevice, stream) != hipSuccess) return kSystem;
        if (hipLaunchHostFunc(stream, host_barrier, c) != hipSuccess) return kSystem;
        return kOk;
    }
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
    if (!c) return kInvalid;
    const size_t block = recvcount * 4, mine = block * c->nranks;
    if (dtype != 7 || op != 0 || mine * c->nranks + block * c->nranks > kCapacity) return kInvalid;
    if (solo_mode())
        return hipMemcpyAsync(recv, static_cast<const char*>(send) + (size_t)c->rank * block, block, hipMemcpyDeviceToDevice, stream) == hipSuccess
                   ? kOk : kSystem;
    if (async_mode()) {
        if (hipMemcpyAsync(c->seg->data + (size_t)c->rank * mine, send, mine, hipMemcpyDeviceToHost, stream) != hipSuccess) return kSystem;
        if (hipLaunchHostFunc(stream, host_barrier, c) != hipSuccess) return kSystem;
        if (hipLaunchHostFunc(stream, host_reduce, new ReduceJob{c, recvcount}) != hipSuccess) return kSystem;
        if (hipMemcpyAsync(recv, reduce_result(c, recvcount), block, hipMemcpyHostToDevice, stream) != hipSuccess) return kSystem;
        if (hipLaunchHostFunc(stream, host_barrier, c) != hipSuccess) return kSystem;
        return kOk;
    }
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

// one process, several ranks, synchronous form
static int run_local_sync(LocalGroup* g)
{
    const int kind = g->pending[0].kind;
    const size_t count = g->pending[0].count, n = (size_t)g->nranks;
    std::vector<std::vector<float>> host(n);
    for (const PendingOp& op : g->pending) {
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
    return kOk;
}

// ... asynchronous form: copies and host functions on the ranks' own streams, ordered by events.  Staging layout:
// all-gather [rank][count]; reduce-scatter [rank][nranks][count] followed by the results [rank][count].
static int run_local_async(LocalGroup* g)
{
    const int kind = g->pending[0].kind;
    const size_t count = g->pending[0].count, n = (size_t)g->nranks;
    const size_t per_rank = kind == 0 ? count : count * n;
    const size_t need = per_rank * n + (kind == 1 ? count * n : 0);
    Staging& st = g->ring[g->next];
    g->next = (g->next + 1) % 4;
    if (st.used)   // its previous collective (four calls ago) must have left the buffer
        for (hipEvent_t e : st.done)
            if (hipEventSynchronize(e) != hipSuccess) return kSystem;
    if (st.floats < need) {
        if (st.host) (void)hipHostFree(st.host);
        if (hipHostMalloc((void**)&st.host, need * 4, hipHostMallocDefault) != hipSuccess) return kSystem;
        st.floats = need;
    }
    if (st.done.empty()) {
        st.done.resize(n);
        for (hipEvent_t& e : st.done)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return kSystem;
    }
    if (g->copied.empty()) {
        g->copied.resize(n);
        for (hipEvent_t& e : g->copied)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return kSystem;
    }
    for (const PendingOp& op : g->pending) {
        const int r = op.comm->rank;
        if (hipSetDevice(op.comm->device) != hipSuccess ||
            hipMemcpyAsync(st.host + (size_t)r * per_rank, op.send, per_rank * 4, hipMemcpyDeviceToHost, op.stream) != hipSuccess ||
            hipEventRecord(g->copied[r], op.stream) != hipSuccess)
            return kSystem;
    }
    for (const PendingOp& op : g->pending) {
        const int r = op.comm->rank;
        if (hipSetDevice(op.comm->device) != hipSuccess) return kSystem;
        for (size_t k = 0; k < n; ++k)
            if (hipStreamWaitEvent(op.stream, g->copied[k], 0) != hipSuccess) return kSystem;
        const float* src = st.host;
        size_t floats = count * n;
        if (kind == 1) {
            float* out = st.host + per_rank * n + (size_t)r * count;
            if (hipLaunchHostFunc(op.stream, host_local_reduce, new LocalReduceJob{st.host, out, count, (int)n, r}) != hipSuccess)
                return kSystem;
            src = out;
            floats = count;
        }
        if (hipMemcpyAsync(op.recv, src, floats * 4, hipMemcpyHostToDevice, op.stream) != hipSuccess ||
            hipEventRecord(st.done[r], op.stream) != hipSuccess)
            return kSystem;
    }
    st.used = true;
    return kOk;
}

static int run_local(LocalGroup* g)
{
    if (g->pending.empty()) return kOk;
    if ((int)g->pending.size() != g->nranks) return kInvalid;   // every rank must have made the same single call
    for (const PendingOp& op : g->pending)
        if (op.kind != g->pending[0].kind || op.count != g->pending[0].count) return kInvalid;
    const int rc = async_mode() ? run_local_async(g) : run_local_sync(g);
    g->pending.clear();
    return rc;
}

int ncclGroupStart() { ++group_depth; return kOk; }
int ncclGroupEnd()
{
    if (group_depth > 0 && --group_depth == 0) {
        int rc = kOk;
        std::vector<LocalGroup*> groups;
        for (LocalGroup* g : touched)
            if (std::find(groups.begin(), groups.end(), g) == groups.end()) groups.push_back(g);
        touched.clear();
        for (LocalGroup* g : groups) { const int r = run_local(g); if (r != kOk) rc = r; }
        return rc;
    }
    return kOk;
}
const char* ncclGetErrorString(int code) { return code == kOk ? "no error" : code == kInvalid ? "mock: invalid argument" : "mock: system error"; }

}  // extern "C"
