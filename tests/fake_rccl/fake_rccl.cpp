// tests/fake_rccl/fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl.so with the eight entry points libairband_hip.so binds
// (csrc/airband_hip.cpp, rccl()): ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll, ncclAllReduce, ncclCommDestroy, ncclGroupStart,
// ncclGroupEnd, ncclGetErrorString.  It exists so that the SHIPPING exchange code -- airband_hip_comm_init_all / _init_rank / _group_begin /
// allreduce_mixers / _group_end, and the reference-side shim's use of them -- runs with MORE THAN ONE RANK on a box with one GPU (RCCL proper
// refuses two ranks on one GPU) and, for the host logic, on a box with none.  Selected with AIRBAND_HIP_RCCL_LIB=<this library>.
//
// What it does: an all-reduce through a POSIX shared-memory segment named by the unique id.  Ranks may be threads of one process (the shim),
// several communicators driven by ONE thread inside a group (ncclCommInitAll + ncclGroupStart/End, the shim's form), or separate processes
// (bench.py's form).  Per collective: every rank waits for its stream, copies its send buffer into its slot of the segment, all ranks meet,
// every rank reduces the slots IN RANK ORDER (so the result is deterministic and the same on every rank) and copies it into its receive buffer,
// all ranks meet again.  Collectives of one communicator match by order, as in NCCL.
// What it is not: asynchronous.  The work happens on the host when the outermost group ends (or at once outside a group), after a
// hipStreamSynchronize -- stricter than RCCL's stream-ordered enqueue, so it cannot show a missing stream dependency.  A rank that never
// shows up is a 120 s timeout and ncclSystemError, not a hang.
// FAKE_RCCL_HOST=1: buffers are host memory (the CPU-only tests); FAKE_RCCL_LOG=<file>: one line per collective per rank.
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

namespace {

constexpr size_t SLOT_BYTES = 4u << 20;  // per rank: 64 mixers x 2000 floats = 512 KB is the largest thing the library sends
constexpr int MAX_RANKS = 64;
constexpr uint64_t MAGIC = 0x66616b6572636c31ull;  // "fakercl1"

struct Header {
    std::atomic<uint64_t> ready;    // MAGIC once the creator has initialised the rest
    std::atomic<uint64_t> attached; // ranks that have mapped the segment
    std::atomic<uint64_t> arrived;  // monotonically increasing: every rank adds 1 per phase (two phases per collective)
    uint64_t nranks;
};

struct Comm {
    Header* hdr = nullptr;
    unsigned char* slots = nullptr;  // nranks x SLOT_BYTES behind the header
    size_t map_bytes = 0;
    int nranks = 0, rank = 0, device = 0;
    uint64_t phases = 0;  // phases this rank has completed
    uint64_t collectives = 0;
};

struct Op {
    Comm* comm;
    const void* send;
    void* recv;
    size_t count;
    ncclDataType_t dtype;
    ncclRedOp_t op;
    hipStream_t stream;
};

thread_local int t_depth = 0;
thread_local std::vector<Op> t_pending;

bool host_mode() {
    const char* e = getenv("FAKE_RCCL_HOST");
    return e && *e == '1';
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

size_t elem_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

void shm_name(const ncclUniqueId& id, char* out, size_t n) {
    uint64_t v[2];
    memcpy(v, id.internal + 8, sizeof(v));
    snprintf(out, n, "/fake_rccl_%016llx%016llx", (unsigned long long)v[0], (unsigned long long)v[1]);
}

bool wait_until(const std::atomic<uint64_t>& a, uint64_t target) {
    const double t0 = now_s();
    for (unsigned spin = 0; a.load(std::memory_order_acquire) < target; spin++) {
        if ((spin & 255) == 255) {
            if (now_s() - t0 > 120.0) return false;
            usleep(50);
        } else {
            sched_yield();
        }
    }
    return true;
}

ncclResult_t attach(Comm* c, const ncclUniqueId& id, int nranks, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    uint64_t magic;
    memcpy(&magic, id.internal, sizeof(magic));
    if (magic != MAGIC) return ncclInvalidArgument;  // an id that did not come from this library's ncclGetUniqueId
    char name[96];
    shm_name(id, name, sizeof(name));
    const size_t bytes = sizeof(Header) + (size_t)nranks * SLOT_BYTES;
    bool creator = true;
    int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0) {
        creator = false;
        const double t0 = now_s();
        while ((fd = shm_open(name, O_RDWR, 0600)) < 0) {
            if (now_s() - t0 > 120.0) return ncclSystemError;
            usleep(100);
        }
    }
    if (creator && ftruncate(fd, (off_t)bytes) != 0) {
        close(fd);
        shm_unlink(name);
        return ncclSystemError;
    }
    if (!creator) {  // the creator sizes the segment before anybody maps it
        const double t0 = now_s();
        struct stat st;
        while (fstat(fd, &st) == 0 && (size_t)st.st_size < bytes) {
            if (now_s() - t0 > 120.0) {
                close(fd);
                return ncclSystemError;
            }
            usleep(100);
        }
    }
    void* p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->hdr = (Header*)p;
    c->slots = (unsigned char*)p + sizeof(Header);
    c->map_bytes = bytes;
    c->nranks = nranks;
    c->rank = rank;
    if (creator) {  // a fresh segment is zero-filled: the counters start at 0
        c->hdr->nranks = (uint64_t)nranks;
        c->hdr->ready.store(MAGIC, std::memory_order_release);
    } else {
        const bool up = wait_until(c->hdr->ready, MAGIC);
        const bool same = up && c->hdr->nranks == (uint64_t)nranks;
        if (!same) {
            munmap(p, bytes);
            c->hdr = nullptr;
            return up ? ncclInvalidArgument : ncclSystemError;
        }
    }
    c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
    return ncclSuccess;
}

// ncclCommInitRank returns once every rank has joined, as RCCL's does; the last one to see that removes the name
ncclResult_t finish_attach(Comm* c, const ncclUniqueId& id) {
    if (!wait_until(c->hdr->attached, (uint64_t)c->nranks)) return ncclSystemError;
    if (c->rank == 0) {
        char name[96];
        shm_name(id, name, sizeof(name));
        shm_unlink(name);
    }
    return ncclSuccess;
}

template <class T>
void reduce_into(T* acc, const T* in, size_t n, ncclRedOp_t op) {
    switch (op) {
        case ncclSum: for (size_t i = 0; i < n; i++) acc[i] = (T)(acc[i] + in[i]); break;
        case ncclProd: for (size_t i = 0; i < n; i++) acc[i] = (T)(acc[i] * in[i]); break;
        case ncclMax: for (size_t i = 0; i < n; i++) acc[i] = std::max(acc[i], in[i]); break;
        case ncclMin: for (size_t i = 0; i < n; i++) acc[i] = std::min(acc[i], in[i]); break;
        default: break;
    }
}

void reduce_any(void* acc, const void* in, size_t n, ncclDataType_t t, ncclRedOp_t op) {
    switch (t) {
        case ncclInt8: reduce_into((int8_t*)acc, (const int8_t*)in, n, op); break;
        case ncclUint8: reduce_into((uint8_t*)acc, (const uint8_t*)in, n, op); break;
        case ncclInt32: reduce_into((int32_t*)acc, (const int32_t*)in, n, op); break;
        case ncclUint32: reduce_into((uint32_t*)acc, (const uint32_t*)in, n, op); break;
        case ncclInt64: reduce_into((int64_t*)acc, (const int64_t*)in, n, op); break;
        case ncclUint64: reduce_into((uint64_t*)acc, (const uint64_t*)in, n, op); break;
        case ncclFloat32: reduce_into((float*)acc, (const float*)in, n, op); break;
        case ncclFloat64: reduce_into((double*)acc, (const double*)in, n, op); break;
        default: break;
    }
}

void log_line(const Op& o) {
    const char* path = getenv("FAKE_RCCL_LOG");
    if (!path || !*path) return;
    char line[160];
    const int n = snprintf(line, sizeof(line), "allreduce rank=%d nranks=%d count=%zu dtype=%d op=%d pid=%d\n", o.comm->rank, o.comm->nranks, o.count, (int)o.dtype, (int)o.op, (int)getpid());
    const int fd = open(path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        if (write(fd, line, (size_t)n) < 0) {}  // one write: lines of different ranks do not interleave
        close(fd);
    }
}

// One round: at most one collective per communicator.  The ranks this thread drives go through each phase together -- that is what lets ONE
// thread stand for several ranks of a clique inside a group.
ncclResult_t run_round(std::vector<Op>& ops) {
    const bool host = host_mode();
    int dev0 = 0;
    if (!host) (void)hipGetDevice(&dev0);
    ncclResult_t rc = ncclSuccess;
    for (Op& o : ops) {  // phase 1: my contribution into my slot
        const size_t bytes = o.count * elem_size(o.dtype);
        if (elem_size(o.dtype) == 0 || bytes > SLOT_BYTES) return ncclInvalidArgument;
        unsigned char* mine = o.comm->slots + (size_t)o.comm->rank * SLOT_BYTES;
        if (host) {
            memcpy(mine, o.send, bytes);
        } else {
            if (hipSetDevice(o.comm->device) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess || hipMemcpy(mine, o.send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                rc = ncclUnhandledCudaError;
        }
        o.comm->hdr->arrived.fetch_add(1, std::memory_order_acq_rel);
        o.comm->phases++;
    }
    for (Op& o : ops)
        if (!wait_until(o.comm->hdr->arrived, o.comm->phases * (uint64_t)o.comm->nranks)) return ncclSystemError;
    std::vector<unsigned char> acc;
    for (Op& o : ops) {  // phase 2: everybody's slots, in rank order
        const size_t bytes = o.count * elem_size(o.dtype);
        acc.assign(o.comm->slots, o.comm->slots + bytes);
        for (int r = 1; r < o.comm->nranks; r++) reduce_any(acc.data(), o.comm->slots + (size_t)r * SLOT_BYTES, o.count, o.dtype, o.op);
        if (host) {
            memcpy(o.recv, acc.data(), bytes);
        } else {
            if (hipSetDevice(o.comm->device) != hipSuccess || hipMemcpy(o.recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        }
        o.comm->hdr->arrived.fetch_add(1, std::memory_order_acq_rel);
        o.comm->phases++;
        o.comm->collectives++;
        log_line(o);
    }
    for (Op& o : ops)  // nobody's slot is overwritten by the next collective before every rank has read it
        if (!wait_until(o.comm->hdr->arrived, o.comm->phases * (uint64_t)o.comm->nranks)) return ncclSystemError;
    if (!host) (void)hipSetDevice(dev0);
    return rc;
}

ncclResult_t flush() {
    std::vector<Op> pending;
    pending.swap(t_pending);
    std::vector<Comm*> order;  // communicators in the order they first appear in the group
    std::map<Comm*, std::vector<Op>> queue;
    for (const Op& o : pending) {
        if (!queue.count(o.comm)) order.push_back(o.comm);
        queue[o.comm].push_back(o);
    }
    for (size_t r = 0;; r++) {  // round r = the r-th collective of every communicator in the group
        std::vector<Op> round;
        for (Comm* c : order)
            if (queue[c].size() > r) round.push_back(queue[c][r]);
        if (round.empty()) break;
        const ncclResult_t rc = run_round(round);
        if (rc != ncclSuccess) return rc;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

/* capability libairband_hip.so looks for (csrc/airband_hip.cpp, rccl()): ranks of one communicator may share a GPU here -- the duplicate-GPU check of
 * airband_hip_comm_init_all, which protects callers from RCCL's late and generic refusal, is skipped for this library only */
int airband_rccl_allows_shared_gpu(void) { return 1; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    memcpy(id->internal, &MAGIC, sizeof(MAGIC));
    uint64_t v[2] = {(uint64_t)getpid(), (uint64_t)(now_s() * 1e9)};
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) {
        if (read(fd, v, sizeof(v)) < 0) {}
        close(fd);
    }
    static std::atomic<uint64_t> counter{0};
    v[1] ^= counter.fetch_add(1) << 48;
    memcpy(id->internal + 8, v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm) return ncclInvalidArgument;
    Comm* c = new Comm;
    if (!host_mode()) (void)hipGetDevice(&c->device);
    ncclResult_t rc = attach(c, id, nranks, rank);
    if (rc == ncclSuccess) rc = finish_attach(c, id);
    if (rc != ncclSuccess) {
        delete c;
        return rc;
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    ncclUniqueId id;
    ncclGetUniqueId(&id);
    std::vector<Comm*> cs;
    for (int i = 0; i < ndev; i++) {
        Comm* c = new Comm;
        c->device = devlist ? devlist[i] : i;
        const ncclResult_t rc = attach(c, id, ndev, i);
        if (rc != ncclSuccess) {
            delete c;
            return rc;
        }
        cs.push_back(c);
    }
    for (int i = 0; i < ndev; i++) {
        const ncclResult_t rc = finish_attach(cs[i], id);
        if (rc != ncclSuccess) return rc;
        comms[i] = (ncclComm_t)cs[i];
    }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    t_pending.push_back(Op{(Comm*)comm, sendbuff, recvbuff, count, datatype, op, stream});
    return t_depth > 0 ? ncclSuccess : flush();
}

ncclResult_t ncclGroupStart() {
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    return --t_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = (Comm*)comm;
    if (!c) return ncclSuccess;
    if (c->hdr) munmap((void*)c->hdr, c->map_bytes);
    delete c;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake_rccl: HIP call failed";
        case ncclSystemError: return "fake_rccl: a rank did not show up within 120 s (or shared memory failed)";
        case ncclInvalidArgument: return "fake_rccl: invalid argument (id not from this library, rank out of range, buffer above 4 MiB)";
        case ncclInvalidUsage: return "fake_rccl: invalid usage";
        default: return "fake_rccl: error";
    }
}

/* lets a test ask what it loaded */
int fake_rccl_marker(void) { return 1; }

}  // extern "C"
