// comm.hip — the one multi-GPU exchange of the batched path (BASELINE config C4): problem instances are
// sharded block-contiguously over the ranks of one node and never interact (distinct `Solver`s are independent in the reference;
// its only cross-problem loop is differentiate.jl:29-58).  After a batch round the per-problem status rows are all-gathered and the
// step counters all-reduced — RCCL over xGMI, one process per GPU.  There is no collective on the data path.
//
// RCCL is loaded at run time (dlopen) the first time a communicator is made, so single-GPU users of libcalipso_hip.so do not load
// it; the entry points mirror ncclGetUniqueId / ncclCommInitRank so that a Julia (or C) launcher can distribute the 128-byte id
// by whatever means it has (a file, MPI, a socket).
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "internal.hpp"

struct calipso_hip_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl& rccl() {
    static Rccl r;
    if (r.lib || !r.err.empty()) return r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("cannot load librccl: ") + dlerror(); return r; }
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name)); if (!r.field) { r.err = std::string("librccl lacks ") + name; return r; }
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(CommCount, "ncclCommCount"); SYM(CommUserRank, "ncclCommUserRank");
    SYM(AllGather, "ncclAllGather"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return r;
}
thread_local std::string g_comm_err;
int fail(calipso_hip_comm* c, const std::string& m) { if (c) c->err = m; g_comm_err = m; return CALIPSO_ERR_HIP; }
}  // namespace

#define NC(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) return fail(c, std::string(#call) + ": " + R.GetErrorString(r__)); } while (0)
#define HC(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(c, std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)

extern "C" {

const char* calipso_hip_comm_last_error(calipso_hip_comm* c) { return c ? c->err.c_str() : g_comm_err.c_str(); }

// ncclGetUniqueId: called by ONE rank; the 128 bytes are handed to every rank's calipso_hip_comm_init
int32_t calipso_hip_comm_unique_id(uint8_t id[128]) {
    calipso_hip_comm* c = nullptr;
    if (!id) return CALIPSO_ERR_ARGUMENT;
    Rccl& R = rccl();
    if (!R.err.empty()) return fail(c, R.err);
    ncclUniqueId u;
    NC(R.GetUniqueId(&u));
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id, &u, 128);
    return CALIPSO_OK;
}

// ncclCommInitRank on `device` (one process per GPU).  Collective: every rank of the job calls it with the same id.
int32_t calipso_hip_comm_init(int32_t rank, int32_t nranks, const uint8_t id[128], int32_t device, calipso_hip_comm** out) {
    calipso_hip_comm* c = nullptr;
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return CALIPSO_ERR_ARGUMENT;
    *out = nullptr;
    Rccl& R = rccl();
    if (!R.err.empty()) return fail(c, R.err);
    c = new calipso_hip_comm();
    c->rank = rank; c->nranks = nranks; c->device = device;
    *out = c;
    HC(hipSetDevice(device));
    HC(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    ncclUniqueId u;
    std::memcpy(&u, id, 128);
    NC(R.CommInitRank(&c->comm, nranks, u, rank));
    return CALIPSO_OK;
}

// what the COMMUNICATOR reports (ncclCommCount / ncclCommUserRank), not what the caller passed to calipso_hip_comm_init: a job whose ranks did not all join shows here
int32_t calipso_hip_comm_size(calipso_hip_comm* c, int32_t out[2]) {
    if (!c || !out || !c->comm) return CALIPSO_ERR_ARGUMENT;
    Rccl& R = rccl();
    int n = 0, r = -1;
    NC(R.CommCount(c->comm, &n));
    NC(R.CommUserRank(c->comm, &r));
    out[0] = n; out[1] = r;
    return CALIPSO_OK;
}

int32_t calipso_hip_comm_destroy(calipso_hip_comm* c) {
    if (!c) return CALIPSO_OK;
    Rccl& R = rccl();
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && R.CommDestroy) (void)R.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CALIPSO_OK;
}

// All-gather of the per-problem status rows (int32 x 4 each: e.g. [converged, iterations, outer, factorisations]).  Ranks may hold
// different numbers of rows (uneven shards): `counts_out` (nranks entries, may be NULL) receives every rank's row count and
// `all_rows` (capacity cap_rows rows) the rows of all problems in global problem-id order (rank order = block-contiguous shards).
// Returns the total number of rows, or a negative status.
int64_t calipso_hip_comm_gather_status(calipso_hip_comm* c, const int32_t* rows, int64_t n_rows, int32_t* all_rows, int64_t cap_rows, int64_t* counts_out) {
    if (!c || n_rows < 0 || (n_rows > 0 && !rows) || !all_rows) return CALIPSO_ERR_ARGUMENT;
    Rccl& R = rccl();
    HC(hipSetDevice(c->device));
    const int W = c->nranks;
    struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } } cntb, rowb;     // freed on every exit
    HC(hipMalloc(&cntb.p, sizeof(long long) * (size_t)(W + 1)));
    long long* d_cnt = static_cast<long long*>(cntb.p);
    long long mine = n_rows;
    HC(hipMemcpyAsync(d_cnt + W, &mine, sizeof(long long), hipMemcpyHostToDevice, c->stream));
    NC(R.AllGather(d_cnt + W, d_cnt, 1, ncclInt64, c->comm, c->stream));
    std::vector<long long> cnt((size_t)W);
    HC(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(long long) * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    long long kmax = 0, total = 0;
    for (long long k : cnt) { kmax = k > kmax ? k : kmax; total += k; }
    if (counts_out) for (int r = 0; r < W; ++r) counts_out[r] = cnt[(size_t)r];
    // A rank whose all_rows is too small must still take part in the second all-gather (capacities are per rank: leaving here would
    // hang the others inside the collective); it reports the error afterwards.
    const bool fits = total <= cap_rows;
    const size_t slot = (size_t)(kmax > 0 ? kmax : 1) * 4;
    HC(hipMalloc(&rowb.p, sizeof(int32_t) * slot * (size_t)(W + 1)));
    int32_t* d_rows = static_cast<int32_t*>(rowb.p);
    HC(hipMemsetAsync(d_rows + slot * (size_t)W, 0xff, sizeof(int32_t) * slot, c->stream));     // padding rows = -1
    if (n_rows) HC(hipMemcpyAsync(d_rows + slot * (size_t)W, rows, sizeof(int32_t) * 4 * (size_t)n_rows, hipMemcpyHostToDevice, c->stream));
    NC(R.AllGather(d_rows + slot * (size_t)W, d_rows, slot, ncclInt32, c->comm, c->stream));
    std::vector<int32_t> h(slot * (size_t)W);
    HC(hipMemcpyAsync(h.data(), d_rows, sizeof(int32_t) * slot * (size_t)W, hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    if (!fits) return fail(c, "calipso_hip_comm_gather_status: all_rows too small");
    size_t o = 0;
    for (int r = 0; r < W; ++r) { std::memcpy(all_rows + 4 * o, h.data() + slot * (size_t)r, sizeof(int32_t) * 4 * (size_t)cnt[(size_t)r]); o += (size_t)cnt[(size_t)r]; }
    return total;
}

// in-place sum over ranks of `count` doubles (step / problem counters for the throughput figure)
int32_t calipso_hip_comm_allreduce_sum(calipso_hip_comm* c, double* values, int64_t count) {
    if (!c || count < 0 || (count > 0 && !values)) return CALIPSO_ERR_ARGUMENT;
    if (count == 0) return CALIPSO_OK;
    Rccl& R = rccl();
    HC(hipSetDevice(c->device));
    struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) (void)hipFree(p); } } buf;
    HC(hipMalloc(&buf.p, sizeof(double) * (size_t)count));
    double* d = static_cast<double*>(buf.p);
    HC(hipMemcpyAsync(d, values, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    NC(R.AllReduce(d, d, (size_t)count, ncclDouble, ncclSum, c->comm, c->stream));
    HC(hipMemcpyAsync(values, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HC(hipStreamSynchronize(c->stream));
    return CALIPSO_OK;
}

}  // extern "C"
