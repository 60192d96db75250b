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
// Point-to-point: ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd (the library's "exchange_p2p"), between processes
// through the shared segment and between the shard threads of one process through the staging ring; same modes.
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
// one process, several ranks, each driven by ITS OWN THREAD (the library's ShardCrew): a ring of pinned staging
// buffers, each guarded by the events of its last use; the threads meet at a barrier inside every call
struct Staging {
    float* host = nullptr;
    size_t floats = 0;
    std::vector<hipEvent_t> copied;   // per rank: its contribution has reached the staging buffer
    std::vector<hipEvent_t> done;     // per rank: recorded after that rank's host->device copy
    std::atomic<bool> used{false};
};
struct LocalGroup {
    int nranks = 0;
    std::atomic<int> alive{0};
    pthread_barrier_t barrier;        // among the ranks' threads
    Staging ring[4];
    std::vector<unsigned long> calls; // per rank: collectives issued so far (picks the staging buffer)
    std::atomic<int> failed{0};
};
// point-to-point operations collect between ncclGroupStart and ncclGroupEnd (per calling thread: in the one-process mode every
// communicator has a thread of its own) and run at the end of the group
struct P2POp { bool send; void* buf; size_t count; int peer; Comm* comm; hipStream_t stream; };
thread_local std::vector<P2POp> p2p_pending;
thread_local int group_depth = 0;
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

// One collective of one rank of a one-process group, called by that rank's own thread; the threads of all ranks make the
// same calls in the same order and meet inside.  Asynchronous like RCCL towards the GPU: copies and host functions on the
// caller's stream, ordered by events — nothing is drained.  Staging layout: all-gather [rank][count]; reduce-scatter
// [rank][nranks][count] followed by the results [rank][count].  (MURB_MOCK_MODE=sync: every step drained instead.)
int run_local(Comm* c, int kind, const void* send, void* recv, size_t count, hipStream_t stream)
{
    LocalGroup* g = c->local;
    const int r = c->rank;
    const size_t n = (size_t)g->nranks;
    const size_t per_rank = kind == 0 ? count : count * n;
    const size_t need = per_rank * n + (kind == 1 ? count * n : 0);
    Staging& st = g->ring[g->calls[(size_t)r]++ % 4];
    const bool sync = !async_mode();
    int rc = kOk;
    if (st.used.load())   // its previous collective (four calls ago) must have left the buffer on every rank
        for (hipEvent_t e : st.done)
            if (hipEventSynchronize(e) != hipSuccess) rc = kSystem;
    if (st.floats < need || st.copied.empty()) {   // same verdict on every thread: st changes only between two barriers
        pthread_barrier_wait(&g->barrier);
        if (r == 0) {
            if (st.floats < need) {
                if (st.host) (void)hipHostFree(st.host);
                if (hipHostMalloc((void**)&st.host, need * 4, hipHostMallocDefault) != hipSuccess) { g->failed.store(1); st.host = nullptr; }
                st.floats = st.host ? need : 0;
            }
            if (st.copied.empty()) {
                st.copied.resize(n); st.done.resize(n);
                for (size_t k = 0; k < n; ++k)
                    if (hipEventCreateWithFlags(&st.copied[k], hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&st.done[k], hipEventDisableTiming) != hipSuccess) g->failed.store(1);
            }
        }
        pthread_barrier_wait(&g->barrier);
    }
    if (g->failed.load()) rc = kSystem;
    if (rc == kOk && (hipMemcpyAsync(st.host + (size_t)r * per_rank, send, per_rank * 4, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                      hipEventRecord(st.copied[(size_t)r], stream) != hipSuccess))
        rc = kSystem;
    if (rc == kOk && sync && hipStreamSynchronize(stream) != hipSuccess) rc = kSystem;
    if (rc != kOk) g->failed.store(1);
    pthread_barrier_wait(&g->barrier);   // every rank's `copied` event of this call is recorded
    if (g->failed.load()) return kSystem;
    for (size_t k = 0; k < n; ++k)
        if (hipStreamWaitEvent(stream, st.copied[k], 0) != hipSuccess) return kSystem;
    const float* src = st.host;
    size_t floats = count * n;
    if (kind == 1) {
        float* out = st.host + per_rank * n + (size_t)r * count;
        if (hipLaunchHostFunc(stream, host_local_reduce, new LocalReduceJob{st.host, out, count, (int)n, r}) != hipSuccess) return kSystem;
        src = out;
        floats = count;
    }
    if (hipMemcpyAsync(recv, src, floats * 4, hipMemcpyHostToDevice, stream) != hipSuccess ||
        hipEventRecord(st.done[(size_t)r], stream) != hipSuccess)
        return kSystem;
    if (sync && hipStreamSynchronize(stream) != hipSuccess) return kSystem;
    st.used.store(true);
    return kOk;
}

// One group of sends and receives of one rank (all of `count` floats, all on one stream).  Staging slot [src][dst].  Like the
// collectives above: asynchronous towards the GPU, every rank (thread) of the communicator must issue its group, and they meet
// inside (a real library only pairs senders with receivers; the library under test issues the same groups on every rank).
int run_p2p(std::vector<P2POp>& ops)
{
    if (ops.empty()) return kOk;
    Comm* c = ops[0].comm;
    hipStream_t stream = ops[0].stream;
    const size_t count = ops[0].count, n = (size_t)c->nranks;
    for (const P2POp& op : ops)
        if (op.comm != c || op.stream != stream || op.count != count || op.peer < 0 || op.peer >= c->nranks || op.peer == c->rank) return kInvalid;
    const size_t need = n * n * count;
    if (solo_mode()) return kOk;   // nobody to talk to: receive buffers keep what they hold
    const bool sync = !async_mode();
    const int r = c->rank;
    if (!c->local) {               // ranks are processes: through the shared segment
        if (need * 4 > kCapacity) return kInvalid;
        float* stage = reinterpret_cast<float*>(c->seg->data);
        if (sync && hipStreamSynchronize(stream) != hipSuccess) return kSystem;
        for (const P2POp& op : ops)
            if (op.send && hipMemcpyAsync(stage + ((size_t)r * n + (size_t)op.peer) * count, op.buf, count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess)
                return kSystem;
        if (sync) { if (hipStreamSynchronize(stream) != hipSuccess) return kSystem; pthread_barrier_wait(&c->seg->barrier); }
        else if (hipLaunchHostFunc(stream, host_barrier, c) != hipSuccess) return kSystem;
        for (const P2POp& op : ops)
            if (!op.send && hipMemcpyAsync(op.buf, stage + ((size_t)op.peer * n + (size_t)r) * count, count * 4, hipMemcpyHostToDevice, stream) != hipSuccess)
                return kSystem;
        if (sync) { if (hipStreamSynchronize(stream) != hipSuccess) return kSystem; pthread_barrier_wait(&c->seg->barrier); }
        else if (hipLaunchHostFunc(stream, host_barrier, c) != hipSuccess) return kSystem;
        return kOk;
    }
    // ranks are threads of this process: the staging ring of run_local
    LocalGroup* g = c->local;
    Staging& st = g->ring[g->calls[(size_t)r]++ % 4];
    int rc = kOk;
    if (st.used.load())
        for (hipEvent_t e : st.done)
            if (hipEventSynchronize(e) != hipSuccess) rc = kSystem;
    if (st.floats < need || st.copied.empty()) {
        pthread_barrier_wait(&g->barrier);
        if (r == 0) {
            if (st.floats < need) {
                if (st.host) (void)hipHostFree(st.host);
                if (hipHostMalloc((void**)&st.host, need * 4, hipHostMallocDefault) != hipSuccess) { g->failed.store(1); st.host = nullptr; }
                st.floats = st.host ? need : 0;
            }
            if (st.copied.empty()) {
                st.copied.resize(n); st.done.resize(n);
                for (size_t k = 0; k < n; ++k)
                    if (hipEventCreateWithFlags(&st.copied[k], hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&st.done[k], hipEventDisableTiming) != hipSuccess) g->failed.store(1);
            }
        }
        pthread_barrier_wait(&g->barrier);
    }
    if (g->failed.load()) rc = kSystem;
    for (const P2POp& op : ops)
        if (rc == kOk && op.send &&
            hipMemcpyAsync(st.host + ((size_t)r * n + (size_t)op.peer) * count, op.buf, count * 4, hipMemcpyDeviceToHost, stream) != hipSuccess)
            rc = kSystem;
    if (rc == kOk && hipEventRecord(st.copied[(size_t)r], stream) != hipSuccess) rc = kSystem;
    if (rc == kOk && sync && hipStreamSynchronize(stream) != hipSuccess) rc = kSystem;
    if (rc != kOk) g->failed.store(1);
    pthread_barrier_wait(&g->barrier);   // every rank's sends are enqueued and their `copied` events recorded
    if (g->failed.load()) return kSystem;
    for (const P2POp& op : ops)
        if (!op.send && (hipStreamWaitEvent(stream, st.copied[(size_t)op.peer], 0) != hipSuccess ||
                         hipMemcpyAsync(op.buf, st.host + ((size_t)op.peer * n + (size_t)r) * count, count * 4, hipMemcpyHostToDevice, stream) != hipSuccess))
            return kSystem;
    if (hipEventRecord(st.done[(size_t)r], stream) != hipSuccess) return kSystem;
    if (sync && hipStreamSynchronize(stream) != hipSuccess) return kSystem;
    st.used.store(true);
    return kOk;
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

// one process, several "devices" (the same ordinal may repeat); every communicator is then used by a thread of its own
int ncclCommInitAll(void** out, int ndev, const int* devices)
{
    if (!out || ndev < 1 || !devices) return kInvalid;
    LocalGroup* g = new LocalGroup;
    g->nranks = ndev;
    g->alive.store(ndev);
    g->calls.assign((size_t)ndev, 0ul);
    pthread_barrier_init(&g->barrier, nullptr, (unsigned)ndev);
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
                for (hipEvent_t e : s.copied) (void)hipEventDestroy(e);
            }
            pthread_barrier_destroy(&c->local->barrier);
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
    if (c && c->local) return dtype == 7 ? run_local(c, 0, send, recv, count, stream) : kInvalid;
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
    if (c && c->local) return (dtype == 7 && op == 0) ? run_local(c, 1, send, recv, recvcount, stream) : kInvalid;
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

// groups only matter for the point-to-point calls (the library issues its collectives ungrouped, one thread per communicator)
int ncclGroupStart() { ++group_depth; return kOk; }
int ncclGroupEnd()
{
    if (group_depth <= 0 || --group_depth > 0) return kOk;
    const int rc = run_p2p(p2p_pending);
    p2p_pending.clear();
    return rc;
}
int ncclSend(const void* send, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    if (!comm || dtype != 7 || group_depth == 0) return kInvalid;   // the library always groups its sends and receives
    p2p_pending.push_back(P2POp{true, const_cast<void*>(send), count, peer, static_cast<Comm*>(comm), stream});
    return kOk;
}
int ncclRecv(void* recv, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    if (!comm || dtype != 7 || group_depth == 0) return kInvalid;
    p2p_pending.push_back(P2POp{false, recv, count, peer, static_cast<Comm*>(comm), stream});
    return kOk;
}
const char* ncclGetErrorString(int code) { return code == kOk ? "no error" : code == kInvalid ? "mock: invalid argument" : "mock: system error"; }

}  // extern "C"
