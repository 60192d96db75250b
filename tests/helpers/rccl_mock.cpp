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
struct Comm {
    Segment* seg = nullptr;
    int rank = 0, nranks = 1;
    std::string name;
};
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

int ncclCommInitAll(void**, int, const int*) { return kInvalid; }   // one process, several devices: not mocked

int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return kOk;
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

int ncclGroupStart() { return kOk; }
int ncclGroupEnd() { return kOk; }
const char* ncclGetErrorString(int code) { return code == kOk ? "no error" : code == kInvalid ? "mock: invalid argument" : "mock: system error"; }

}  // extern "C"
