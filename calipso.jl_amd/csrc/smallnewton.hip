// smallnewton.hip — solve! for a BATCH of small conic QPs: one workgroup per problem instance, the whole Newton iteration resident in the compute unit's LDS.
//
// What it is: the inner loop body of solve! (src/solver/solve.jl:98-353) and its outer updates (:356-368) — evaluate!, cone!, merit / merit gradient, residual!,
// optimality_error, inertia_correction! (inertia.jl:30-80), the condensed assemble + LDL^T + search_direction_symmetric! (search_direction.jl:25-104),
// iterative_refinement! (iterative_refinement.jl:1-52), the fraction-to-boundary cone search (:190-221), the filter line search (:224-302, line_search.jl, filter.jl)
// and the accept (:309-333) — for problems so small that the general path (one launch per kernel of the step, ~130 launches, five host read-backs) is nothing but
// latency: the MPC-sized problems of the reference's auto-tuning loop (examples/autotuning/cartpole.jl:179-227: n = 89) by the thousand.  EVERY decision the
// reference's host code takes (exit tests, regularisation loop, refinement loop, step-size searches, filter) is taken on the device; ONE launch carries whole solve!s
// (or `count` Newton steps) of all instances.
//
// Scope: the device-resident QP evaluator of qp.hip (f = c x'Px + q'x, g = Ax - b, cone constraint h - Gx) with nonnegative cones (second-order cones and other
// evaluators: the general path); residual_norm = constraint_norm = 1 (the defaults of options.jl).  When iterative refinement fails, the reference falls back to
// `H \ residual` (search_direction.jl:22): such an instance stops with status CALIPSO_WARN_REFINEMENT and is left to the general path.
//
// Arithmetic: the condensed system in the constraint-first order [z | y | x] of DESIGN.md 4 — closed-form pivots for the y and z blocks, S = Lsym + ep I +
// [A; -G]' Omega [A; -G] factored without pivoting in LDS, inertia = signs of the closed-form pivots + signs of D(S) (negative = #(d <= 0) as compute_inertia!).
// Iterates agree with the oracle's per accepted step to 1e-8 (tests/test_gpu_smallnewton.py); not bit for bit with the general path (other summation orders).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "internal.hpp"
#include "device_utils.hpp"

struct calipso_hip_smallnewton {
    int nx = 0, ne = 0, nc = 0, batch = 0, device = 0;
    int nq = 0; std::vector<int> soc_start, soc_dim, soc_woff; int wsz = 0, maxd = 0;      // cone layout: nq nonnegative entries, then the second-order cones (contiguous)
    int *d_soc = nullptr;                                                                   // device: [start | dim | woff], nsoc each
    calipso::Options opt;
    double objective_scale = 0.5;
    bool shared_qp = false, have_qp = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *P = nullptr, *q = nullptr, *Z = nullptr, *bh = nullptr;      // Lxx = 2 c P (nx x nx), q, Z = [A; -G] (m x nx, ld m), bh = [-b; h]: per instance or shared
    double *w = nullptr, *lam = nullptr, *sc = nullptr, *filt = nullptr, *info = nullptr, *trace = nullptr, *prof = nullptr;
    double *rtheta = nullptr, *sens = nullptr, *stf = nullptr; size_t cap_diff = 0; bool diff_shared = false;      // differentiate!: batch x N x p each
    long long* cnt = nullptr; int* status = nullptr;
    int trace_rows = 0;
    size_t lds_bytes = 0;
    int threads = 0;                 // options.threads: 0 = by the LDS footprint (sn_threads), 64 / 128 / 256 forced
    double last_ms = 0.0;
    std::string err;
};

namespace {
using calipso::Options;
typedef calipso_hip_smallnewton SN;

#ifndef SN_JB
#define SN_JB 8          // columns per panel of the LDL^T (bench/small_newton_phases.sh builds other values)
#endif
enum { SC_KAPPA = 0, SC_TAU, SC_RHO, SC_EP, SC_EPLAST, SC_ED, SC_EQV, SC_CPV, SC_F, SC_COUNT = 16 };
enum { CN_TOTAL = 0, CN_OUTER, CN_INNER, CN_FACT, CN_RFAIL, CN_RMAX, CN_RLAST, CN_STEPS, CN_FILTER, CN_TRACE, CN_COUNT = 16 };
enum { IN_STEP = 0, IN_STEP_T, IN_ROUNDS, IN_NFACT, IN_MH, IN_THETAH, IN_EXIT, IN_OPT, IN_COUNT = 8 };
enum { MODE_SOLVE = 0, MODE_STEPS = 1, MODE_DIFF = 2 };

struct Dm {
    int nx, ne, nc, m, n, N, ldz, q, nsoc, wsz, maxd;      // q nonnegative entries first, then nsoc second-order cones (contiguous ranges); wsz = sum of dim^2; ldz: leading dimension of Z in LDS (odd: conflict-free column walks)
    __host__ __device__ int orr() const { return nx; }
    __host__ __device__ int os() const { return nx + ne; }
    __host__ __device__ int oy() const { return nx + ne + nc; }
    __host__ __device__ int oz() const { return nx + ne + nc + ne; }
    __host__ __device__ int ot() const { return nx + ne + nc + ne + nc; }
};

// LDS carve-up (offsets in doubles): the same function sizes the launch on the host and places the pointers on the device
struct Lay { int Z, S, q, bh, lam, sol, cand, step, res, rerr, corr, rsym, fx, gzx, gh, ghc, cprod, bgrad, wz, wsoc, bsoc, vsoc, D, Dinv, xb, t1, t2, ycol, red, total; };
__host__ __device__ inline Lay layout(const Dm& d) {
    Lay L; int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
    L.Z = take(d.ldz * d.nx); L.S = take(d.nx * (d.nx + 1) / 2);
    L.q = take(d.nx); L.bh = take(d.m); L.lam = take(d.ne);
    L.sol = take(d.N); L.cand = take(d.N); L.step = take(d.N); L.res = take(d.N); L.rerr = take(d.N); L.corr = take(d.N);
    L.rsym = take(d.n);
    L.fx = take(d.nx); L.gzx = take(d.nx); L.gh = take(d.m); L.ghc = take(d.m);
    L.cprod = take(d.nc); L.bgrad = take(d.nc); L.wz = take(d.nc); L.wsoc = take(d.wsz); L.bsoc = take(d.wsz); L.vsoc = take(4 * d.maxd * d.nsoc);
    L.D = take(d.nx); L.Dinv = take(d.nx); L.xb = take(d.nx); L.t1 = take(d.m); L.t2 = take(d.m);
    L.ycol = take(SN_JB * d.nx);
    L.red = take(64);
    L.total = o;
    return L;
}

struct Args {
    Dm d; Options o;
    const double *P, *q, *Z, *bh; long long sP, sq, sZ, sbh;      // element strides per instance (0: one problem shared by all)
    double *w, *lam, *sc, *filt, *info, *trace, *prof; long long* cnt; int* status;
    const int *soc_start, *soc_dim, *soc_woff;      // per second-order cone: first cone-local index, dimension, offset of its dim x dim blocks
    int batch, mode, count, advance, trace_rows;
    double* stf;                                    // batch x 2 nc: s and t at the last search direction (smallnewton_device.hpp: quirk B-12)
    const double* rtheta; double* sens; long long srtheta;             // differentiate!: dR/dtheta and the sensitivities, per instance N x count, column-major
};

// the device code, once per workgroup size (launch() picks: sn_threads())
#define SN_THREADS 64
namespace t64 {
#include "smallnewton_device.hpp"
}
#undef SN_THREADS
#define SN_THREADS 128
namespace t128 {
#include "smallnewton_device.hpp"
}
#undef SN_THREADS
#define SN_THREADS 256
namespace t256 {
#include "smallnewton_device.hpp"
}
#undef SN_THREADS

int fail(SN* s, int code, const std::string& msg) { s->err = msg; return code; }
thread_local std::string g_sn_err;

#define SK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(s, CALIPSO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)

Dm dims_of(const SN* s) {
    Dm d; d.nx = s->nx; d.ne = s->ne; d.nc = s->nc; d.m = s->ne + s->nc; d.n = s->nx + d.m; d.N = s->nx + 2 * s->ne + 3 * s->nc;
    d.ldz = (d.m > 0 ? d.m : 1) | 1;
    d.q = s->nq; d.nsoc = (int)s->soc_dim.size(); d.wsz = s->wsz; d.maxd = s->maxd;
    return d;
}

// threads per instance.  The kernel is a chain of dependent phases: a compute unit's throughput is (resident instances) / (latency of one), the latency grows slowly as
// wavefronts are taken away (C5 shape at equal residency: 92.7 / 99.7 / 121.4 us a step with 4 / 2 / 1 wavefronts), and with 256 registers per thread a compute unit
// holds 8 wavefronts.  So: as many instances as the LDS footprint allows, and the most wavefronts each that still fit — one wavefront from 6 instances per compute unit,
// two from 3, four otherwise (profiles/r06_small_newton_threads.txt; options.threads forces a count)
int sn_threads(const SN* s) {
    if (s->threads == 64 || s->threads == 128 || s->threads == 256) return s->threads;
    const size_t per = s->lds_bytes + 1280, lds = 160 * 1024;
    return 6 * per <= lds ? 64 : 3 * per <= lds ? 128 : 256;
}
int grant_lds(SN* s) {
    if (s->lds_bytes <= 64 * 1024) return CALIPSO_OK;
    (void)calipso::lds_attribute((const void*)t64::k_smallnewton<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)t64::k_smallnewton<true>, 160 * 1024);
    (void)calipso::lds_attribute((const void*)t128::k_smallnewton<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)t128::k_smallnewton<true>, 160 * 1024);
    (void)calipso::lds_attribute((const void*)t256::k_smallnewton<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)t256::k_smallnewton<true>, 160 * 1024);
    (void)calipso::lds_attribute((const void*)t64::k_smallnewton_diff<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)t64::k_smallnewton_diff<true>, 160 * 1024);
    (void)calipso::lds_attribute((const void*)t128::k_smallnewton_diff<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)t128::k_smallnewton_diff<true>, 160 * 1024);
    (void)calipso::lds_attribute((const void*)t256::k_smallnewton_diff<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)t256::k_smallnewton_diff<true>, 160 * 1024);
    return CALIPSO_OK;
}

int launch(SN* s, int mode, int count, int advance) {
    if (!s->have_qp) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton: no problem data (calipso_hip_smallnewton_set_qp)");
    SK(hipSetDevice(s->device));
    Args a;
    a.d = dims_of(s); a.o = s->opt;
    const size_t nx = s->nx, m = a.d.m;
    a.P = s->P; a.q = s->q; a.Z = s->Z; a.bh = s->bh;
    a.sP = s->shared_qp ? 0 : (long long)(nx * nx); a.sq = s->shared_qp ? 0 : (long long)nx; a.sZ = s->shared_qp ? 0 : (long long)(std::max<size_t>(m, 1) * nx);
    a.sbh = s->shared_qp ? 0 : (long long)std::max<size_t>(m, 1);
    a.w = s->w; a.lam = s->lam; a.sc = s->sc; a.filt = s->filt; a.info = s->info; a.trace = s->trace; a.prof = s->prof; a.cnt = s->cnt; a.status = s->status;
    { const int ns = (int)s->soc_dim.size(); a.soc_start = s->d_soc; a.soc_dim = s->d_soc ? s->d_soc + ns : nullptr; a.soc_woff = s->d_soc ? s->d_soc + 2 * ns : nullptr; }
    a.batch = s->batch; a.mode = mode; a.count = count; a.advance = advance; a.trace_rows = s->trace_rows;
    a.rtheta = s->rtheta; a.sens = s->sens; a.stf = s->stf; a.srtheta = s->diff_shared ? 0 : (long long)a.d.N * (long long)count;
    static_assert(sizeof(Args) <= 3800, "kernel arguments");
    SK(hipEventRecord(s->ev0, s->stream));
    const bool soc = !s->soc_dim.empty();
    if (mode == MODE_DIFF) {
        if (sn_threads(s) == 64) {
            if (soc) hipLaunchKernelGGL(t64::k_smallnewton_diff<true>, dim3((unsigned)s->batch), dim3(64), s->lds_bytes, s->stream, a);
            else hipLaunchKernelGGL(t64::k_smallnewton_diff<false>, dim3((unsigned)s->batch), dim3(64), s->lds_bytes, s->stream, a);
        } else if (sn_threads(s) == 128) {
            if (soc) hipLaunchKernelGGL(t128::k_smallnewton_diff<true>, dim3((unsigned)s->batch), dim3(128), s->lds_bytes, s->stream, a);
            else hipLaunchKernelGGL(t128::k_smallnewton_diff<false>, dim3((unsigned)s->batch), dim3(128), s->lds_bytes, s->stream, a);
        } else {
            if (soc) hipLaunchKernelGGL(t256::k_smallnewton_diff<true>, dim3((unsigned)s->batch), dim3(256), s->lds_bytes, s->stream, a);
            else hipLaunchKernelGGL(t256::k_smallnewton_diff<false>, dim3((unsigned)s->batch), dim3(256), s->lds_bytes, s->stream, a);
        }
    } else if (sn_threads(s) == 64) {
        if (soc) hipLaunchKernelGGL(t64::k_smallnewton<true>, dim3((unsigned)s->batch), dim3(64), s->lds_bytes, s->stream, a);
        else hipLaunchKernelGGL(t64::k_smallnewton<false>, dim3((unsigned)s->batch), dim3(64), s->lds_bytes, s->stream, a);
    } else if (sn_threads(s) == 128) {
        if (soc) hipLaunchKernelGGL(t128::k_smallnewton<true>, dim3((unsigned)s->batch), dim3(128), s->lds_bytes, s->stream, a);
        else hipLaunchKernelGGL(t128::k_smallnewton<false>, dim3((unsigned)s->batch), dim3(128), s->lds_bytes, s->stream, a);
    } else {
        if (soc) hipLaunchKernelGGL(t256::k_smallnewton<true>, dim3((unsigned)s->batch), dim3(256), s->lds_bytes, s->stream, a);
        else hipLaunchKernelGGL(t256::k_smallnewton<false>, dim3((unsigned)s->batch), dim3(256), s->lds_bytes, s->stream, a);
    }
    SK(hipGetLastError());
    SK(hipEventRecord(s->ev1, s->stream));
    SK(hipStreamSynchronize(s->stream));
    float ms = 0.f;
    SK(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    s->last_ms = ms;
    return CALIPSO_OK;
}

}  // namespace

extern "C" {

const char* calipso_hip_smallnewton_last_error(calipso_hip_smallnewton* s) { return s ? s->err.c_str() : g_sn_err.c_str(); }

int32_t calipso_hip_smallnewton_create(int64_t nx, int64_t ne, int64_t nc, int64_t batch, int32_t device, calipso_hip_smallnewton** out) {
    if (!out) return CALIPSO_ERR_ARGUMENT;
    *out = nullptr;
    if (nx < 1 || ne < 0 || nc < 0 || batch < 1 || nx > 128 || batch > (1 << 22)) { g_sn_err = "calipso_hip_smallnewton_create: 1 <= nx <= 128, ne, nc >= 0, batch >= 1"; return CALIPSO_ERR_ARGUMENT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) { g_sn_err = "no HIP device available (libcalipso_hip has no CPU path)"; return CALIPSO_ERR_HIP; }
    SN* s = new SN();
    s->nx = (int)nx; s->ne = (int)ne; s->nc = (int)nc; s->batch = (int)batch; s->device = device;
    s->nq = (int)nc;                                  // all cone entries nonnegative until calipso_hip_smallnewton_set_cones says otherwise
    *out = s;
    const Dm d = dims_of(s);
    s->lds_bytes = sizeof(double) * (size_t)layout(d).total;
    if (s->lds_bytes > 160 * 1024) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton_create: the problem does not fit the 160 KB of LDS of a compute unit (" + std::to_string(s->lds_bytes) + " bytes): the general path takes it");
    SK(hipSetDevice(device));
    grant_lds(s);
    SK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    SK(hipEventCreate(&s->ev0)); SK(hipEventCreate(&s->ev1));
    const size_t B = (size_t)batch, N = (size_t)d.N;
    auto alloc = [&](double** p, size_t n) { if (hipMalloc((void**)p, sizeof(double) * std::max<size_t>(n, 1)) != hipSuccess) return false; return hipMemsetAsync(*p, 0, sizeof(double) * std::max<size_t>(n, 1), s->stream) == hipSuccess; };
    if (!alloc(&s->w, B * N) || !alloc(&s->lam, B * std::max(1, d.ne)) || !alloc(&s->sc, B * SC_COUNT) || !alloc(&s->filt, B * 6 * (size_t)s->opt.max_filter) || !alloc(&s->info, B * IN_COUNT) || !alloc(&s->prof, 16) || !alloc(&s->stf, B * 2 * (size_t)std::max(1, d.nc)))
        return fail(s, CALIPSO_ERR_HIP, "calipso_hip_smallnewton_create: device allocation failed");
    SK(hipMalloc((void**)&s->cnt, sizeof(long long) * B * CN_COUNT)); SK(hipMemsetAsync(s->cnt, 0, sizeof(long long) * B * CN_COUNT, s->stream));
    SK(hipMalloc((void**)&s->status, sizeof(int) * B)); SK(hipMemsetAsync(s->status, 0, sizeof(int) * B, s->stream));
    {   // solver.jl:81-85 defaults of the scalars
        std::vector<double> sc(B * SC_COUNT, 0.0);
        for (size_t k = 0; k < B; ++k) { sc[k * SC_COUNT + SC_KAPPA] = 0.1; sc[k * SC_COUNT + SC_TAU] = 0.99; sc[k * SC_COUNT + SC_RHO] = 10.0; }
        SK(hipMemcpyAsync(s->sc, sc.data(), sizeof(double) * sc.size(), hipMemcpyHostToDevice, s->stream));
        SK(hipStreamSynchronize(s->stream));
    }
    return CALIPSO_OK;
}

int32_t calipso_hip_smallnewton_destroy(calipso_hip_smallnewton* s) {
    if (!s) return CALIPSO_OK;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (double* p : {s->P, s->q, s->Z, s->bh, s->w, s->lam, s->sc, s->filt, s->info, s->trace, s->prof, s->rtheta, s->sens, s->stf}) if (p) (void)hipFree(p);
    if (s->cnt) (void)hipFree(s->cnt);
    if (s->d_soc) (void)hipFree(s->d_soc);
    if (s->status) (void)hipFree(s->status);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return CALIPSO_OK;
}

// The cone layout (indices.jl:45-63 in the only arrangement the reference is self-consistent for): the first n_nonnegative cone entries are nonnegative,
// the rest are n_soc second-order cones of the given dimensions, one after the other.  Dimensions 2 .. 16 (one thread per cone forms its d x d block: wider cones belong to
// the general path); n_nonnegative + sum(dims) must be nc.
int32_t calipso_hip_smallnewton_set_cones(calipso_hip_smallnewton* s, int64_t n_nonnegative, int64_t n_soc, const int64_t* dims) {
    if (!s || n_nonnegative < 0 || n_soc < 0 || (n_soc > 0 && !dims)) return CALIPSO_ERR_ARGUMENT;
    long long total = n_nonnegative;
    for (int64_t j = 0; j < n_soc; ++j) { if (dims[j] < 2 || dims[j] > 16) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton_set_cones: second-order cones of dimension 2 .. 16"); total += dims[j]; }
    if (total != s->nc) return fail(s, CALIPSO_ERR_LAYOUT, "calipso_hip_smallnewton_set_cones: n_nonnegative + sum(dims) must equal nc");
    SK(hipSetDevice(s->device));
    s->nq = (int)n_nonnegative;
    s->soc_start.clear(); s->soc_dim.clear(); s->soc_woff.clear(); s->wsz = 0; s->maxd = 0;
    int at = s->nq;
    for (int64_t j = 0; j < n_soc; ++j) { s->soc_start.push_back(at); s->soc_dim.push_back((int)dims[j]); s->soc_woff.push_back(s->wsz); at += (int)dims[j]; s->wsz += (int)(dims[j] * dims[j]); s->maxd = std::max(s->maxd, (int)dims[j]); }
    if (s->d_soc) { (void)hipFree(s->d_soc); s->d_soc = nullptr; }
    if (n_soc > 0) {
        std::vector<int> h;
        h.insert(h.end(), s->soc_start.begin(), s->soc_start.end()); h.insert(h.end(), s->soc_dim.begin(), s->soc_dim.end()); h.insert(h.end(), s->soc_woff.begin(), s->soc_woff.end());
        SK(hipMalloc((void**)&s->d_soc, sizeof(int) * h.size()));
        SK(hipMemcpy(s->d_soc, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
    }
    s->lds_bytes = sizeof(double) * (size_t)layout(dims_of(s)).total;
    if (s->lds_bytes > 160 * 1024) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton_set_cones: the problem no longer fits the 160 KB of LDS of a compute unit");
    grant_lds(s);
    return CALIPSO_OK;
}

// options.jl:6-59 by name (the hot-path subset)
int32_t calipso_hip_smallnewton_set_option(calipso_hip_smallnewton* s, const char* name, double value) {
    if (!s || !name) return CALIPSO_ERR_ARGUMENT;
    Options& o = s->opt;
    const std::string n = name;
#define OD(f) if (n == #f) { o.f = value; return CALIPSO_OK; }
#define OI(f) if (n == #f) { o.f = (calipso::i64)value; return CALIPSO_OK; }
    OD(scaling_line_search) OD(iterative_refinement_tolerance) OD(central_path_initial) OD(central_path_update_tolerance) OD(central_path_scaling) OD(central_path_exponent)
    OD(penalty_initial) OD(penalty_scaling) OD(dual_initial) OD(residual_tolerance) OD(optimality_tolerance) OD(slack_tolerance) OD(equality_tolerance)
    OD(complementarity_tolerance) OD(min_regularization) OD(primal_regularization_initial) OD(dual_regularization_initial) OD(max_regularization) OD(dual_regularization)
    OD(dual_regularization_exponent) OD(scaling_regularization_initial) OD(scaling_regularization) OD(scaling_regularization_last) OD(max_penalty) OD(violation_tolerance)
    OD(violation_exponent) OD(merit_tolerance) OD(merit_exponent) OD(armijo_tolerance) OD(machine_tolerance) OD(warmstart)
    OI(max_outer_iterations) OI(max_residual_iterations) OI(max_residual_line_search) OI(max_cone_line_search) OI(iterative_refinement) OI(max_iterative_refinement)
    OI(min_iterative_refinement)
#undef OD
#undef OI
    if (n == "residual_norm" || n == "constraint_norm") { if (value == 1.0) return CALIPSO_OK; return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton: only the 1-norm (the default) for " + n); }
    if (n == "threads") {         // (not an option of the reference) threads per instance: 0 = chosen by the LDS footprint, 64, 128 or 256
        if (value != 0.0 && value != 64.0 && value != 128.0 && value != 256.0) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton: threads is 0 (automatic), 64, 128 or 256");
        s->threads = (int)value; return CALIPSO_OK;
    }
    if (n == "max_filter") {      // filter.jl:7-13: the instances' filter pairs live in global memory (6 x max_filter doubles each): re-sized here, emptied
        if (value < 1.0 || value > 1.0e6) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton: 1 <= max_filter <= 1e6");
        if ((calipso::i64)value == (calipso::i64)o.max_filter) return CALIPSO_OK;
        SK(hipSetDevice(s->device));
        SK(hipStreamSynchronize(s->stream));
        double* nf = nullptr;
        const size_t cnt = (size_t)s->batch * 6 * (size_t)value;
        SK(hipMalloc((void**)&nf, sizeof(double) * cnt));
        SK(hipMemset(nf, 0, sizeof(double) * cnt));
        if (s->filt) (void)hipFree(s->filt);
        s->filt = nf;
        o.max_filter = (double)(calipso::i64)value;
        std::vector<long long> cn((size_t)s->batch * CN_COUNT);
        SK(hipMemcpy(cn.data(), s->cnt, sizeof(long long) * cn.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k < s->batch; ++k) cn[(size_t)k * CN_COUNT + CN_FILTER] = 0;
        SK(hipMemcpy(s->cnt, cn.data(), sizeof(long long) * cn.size(), hipMemcpyHostToDevice));
        return CALIPSO_OK;
    }
    return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton_set_option: unknown option " + n);
}

// min c x'Px + q'x  s.t.  Ax = b, h - Gx >= 0  (qp.hip's conventions; column-major host arrays).  shared != 0: ONE problem for all instances (the arrays hold one
// problem), else batch-major arrays (instance k at offset k * size).
int32_t calipso_hip_smallnewton_set_qp(calipso_hip_smallnewton* s, const double* P, const double* q, const double* A, const double* b, const double* G, const double* h,
                                       double objective_scale, int32_t shared) {
    if (!s || !P || !q || (s->ne && (!A || !b)) || (s->nc && (!G || !h))) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    const size_t nx = s->nx, ne = s->ne, nc = s->nc, m = ne + nc, K = shared ? 1 : (size_t)s->batch;
    for (double** p : {&s->P, &s->q, &s->Z, &s->bh}) if (*p) { (void)hipFree(*p); *p = nullptr; }
    std::vector<double> Lxx(K * nx * nx), Z(K * std::max<size_t>(m, 1) * nx, 0.0), bh(K * std::max<size_t>(m, 1), 0.0);
    for (size_t k = 0; k < K; ++k) {
        for (size_t e = 0; e < nx * nx; ++e) Lxx[k * nx * nx + e] = 2.0 * objective_scale * P[k * nx * nx + e];
        double* Zk = Z.data() + k * std::max<size_t>(m, 1) * nx;
        for (size_t c = 0; c < nx; ++c) {
            for (size_t r = 0; r < ne; ++r) Zk[r + c * m] = A[k * ne * nx + r + c * ne];
            for (size_t r = 0; r < nc; ++r) Zk[ne + r + c * m] = -G[k * nc * nx + r + c * nc];
        }
        for (size_t r = 0; r < ne; ++r) bh[k * std::max<size_t>(m, 1) + r] = -b[k * ne + r];
        for (size_t r = 0; r < nc; ++r) bh[k * std::max<size_t>(m, 1) + ne + r] = h[k * nc + r];
    }
    SK(hipMalloc((void**)&s->P, sizeof(double) * Lxx.size())); SK(hipMalloc((void**)&s->q, sizeof(double) * K * nx));
    SK(hipMalloc((void**)&s->Z, sizeof(double) * Z.size())); SK(hipMalloc((void**)&s->bh, sizeof(double) * bh.size()));
    SK(hipMemcpyAsync(s->P, Lxx.data(), sizeof(double) * Lxx.size(), hipMemcpyHostToDevice, s->stream));
    SK(hipMemcpyAsync(s->q, q, sizeof(double) * K * nx, hipMemcpyHostToDevice, s->stream));
    SK(hipMemcpyAsync(s->Z, Z.data(), sizeof(double) * Z.size(), hipMemcpyHostToDevice, s->stream));
    SK(hipMemcpyAsync(s->bh, bh.data(), sizeof(double) * bh.size(), hipMemcpyHostToDevice, s->stream));
    SK(hipStreamSynchronize(s->stream));
    s->shared_qp = shared != 0; s->have_qp = true; s->objective_scale = objective_scale;
    return CALIPSO_OK;
}

// the points (batch x N, the layout of point.jl:13-22), the multiplier estimates lambda (batch x ne) and per instance [central_path, fraction_to_boundary, penalty]
// (batch x 3); NULL leaves what is resident.  initialize!(solver, guess) = set_state with x in the first nx entries of every point, then solve (cold start).
int32_t calipso_hip_smallnewton_set_state(calipso_hip_smallnewton* s, const double* w, const double* lambda, const double* scalars) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    const Dm d = dims_of(s);
    const size_t B = s->batch;
    if (w) SK(hipMemcpyAsync(s->w, w, sizeof(double) * B * d.N, hipMemcpyHostToDevice, s->stream));
    if (lambda && d.ne) SK(hipMemcpyAsync(s->lam, lambda, sizeof(double) * B * d.ne, hipMemcpyHostToDevice, s->stream));
    if (scalars) {
        std::vector<double> sc(B * SC_COUNT);
        SK(hipMemcpyAsync(sc.data(), s->sc, sizeof(double) * sc.size(), hipMemcpyDeviceToHost, s->stream));
        SK(hipStreamSynchronize(s->stream));
        for (size_t k = 0; k < B; ++k) { sc[k * SC_COUNT + SC_KAPPA] = scalars[3 * k]; sc[k * SC_COUNT + SC_TAU] = scalars[3 * k + 1]; sc[k * SC_COUNT + SC_RHO] = scalars[3 * k + 2]; }
        SK(hipMemcpyAsync(s->sc, sc.data(), sizeof(double) * sc.size(), hipMemcpyHostToDevice, s->stream));
    }
    SK(hipStreamSynchronize(s->stream));
    return CALIPSO_OK;
}

// points, lambda, scalars [central_path, fraction_to_boundary, penalty, primal_regularization, primal_regularization_last, dual_regularization] (batch x 6),
// counters [total_iterations, outer, factorizations, refinement_failures, max_refinement_rounds, last_refinement_rounds, newton_steps, accepted iterates] (batch x 8); NULLs skipped
int32_t calipso_hip_smallnewton_get_state(calipso_hip_smallnewton* s, double* w, double* lambda, double* scalars, int64_t* counters) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    const Dm d = dims_of(s);
    const size_t B = s->batch;
    if (w) SK(hipMemcpyAsync(w, s->w, sizeof(double) * B * d.N, hipMemcpyDeviceToHost, s->stream));
    if (lambda && d.ne) SK(hipMemcpyAsync(lambda, s->lam, sizeof(double) * B * d.ne, hipMemcpyDeviceToHost, s->stream));
    std::vector<double> sc(B * SC_COUNT); std::vector<long long> cn(B * CN_COUNT);
    SK(hipMemcpyAsync(sc.data(), s->sc, sizeof(double) * sc.size(), hipMemcpyDeviceToHost, s->stream));
    SK(hipMemcpyAsync(cn.data(), s->cnt, sizeof(long long) * cn.size(), hipMemcpyDeviceToHost, s->stream));
    SK(hipStreamSynchronize(s->stream));
    for (size_t k = 0; k < B; ++k) {
        if (scalars) { const double* r = sc.data() + k * SC_COUNT; double* o = scalars + 6 * k; o[0] = r[SC_KAPPA]; o[1] = r[SC_TAU]; o[2] = r[SC_RHO]; o[3] = r[SC_EP]; o[4] = r[SC_EPLAST]; o[5] = r[SC_ED]; }
        if (counters) { const long long* r = cn.data() + k * CN_COUNT; int64_t* o = counters + 8 * k; o[0] = r[CN_TOTAL]; o[1] = r[CN_OUTER]; o[2] = r[CN_FACT]; o[3] = r[CN_RFAIL]; o[4] = r[CN_RMAX]; o[5] = r[CN_RLAST]; o[6] = r[CN_STEPS]; o[7] = r[CN_TRACE]; }
    }
    return CALIPSO_OK;
}

// keep the first `rows` accepted iterates of every instance (solution.all after each accepted inner iteration, solve.jl:309-326): what tests compare with the oracle's trace
int32_t calipso_hip_smallnewton_trace(calipso_hip_smallnewton* s, int32_t rows, double* out) {
    if (!s || rows < 0) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    const Dm d = dims_of(s);
    if (out) {      // read the rows recorded so far
        if (!s->trace || rows > s->trace_rows) return fail(s, CALIPSO_ERR_ARGUMENT, "calipso_hip_smallnewton_trace: no trace of that many rows was requested");
        std::vector<double> all((size_t)s->batch * s->trace_rows * d.N);
        SK(hipMemcpy(all.data(), s->trace, sizeof(double) * all.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k < s->batch; ++k) std::memcpy(out + (size_t)k * rows * d.N, all.data() + (size_t)k * s->trace_rows * d.N, sizeof(double) * (size_t)rows * d.N);
        return CALIPSO_OK;
    }
    if (s->trace) { (void)hipFree(s->trace); s->trace = nullptr; }
    s->trace_rows = rows;
    if (rows > 0) { SK(hipMalloc((void**)&s->trace, sizeof(double) * (size_t)s->batch * rows * d.N)); SK(hipMemset(s->trace, 0, sizeof(double) * (size_t)s->batch * rows * d.N)); }
    return CALIPSO_OK;
}

// solve!(solver) for every instance in ONE launch (cold start unless opt.warmstart: x from the resident points).  result[k] = 1 converged, 0 iteration caps reached,
// CALIPSO_ERR_INERTIA / CALIPSO_ERR_CONE_SEARCH as the reference's error()s, -100 - CALIPSO_WARN_REFINEMENT where the reference would fall back to H \ residual.
int32_t calipso_hip_smallnewton_solve(calipso_hip_smallnewton* s, int32_t* result, double* ms) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    const int rc = launch(s, MODE_SOLVE, 0, 1);
    if (rc < 0) return rc;
    if (result) SK(hipMemcpy(result, s->status, sizeof(int) * (size_t)s->batch, hipMemcpyDeviceToHost));
    if (ms) *ms = s->last_ms;
    return CALIPSO_OK;
}

// `count` Newton steps (the inner loop body of solve!) of every instance in ONE launch from the resident state; advance = 0: every step starts from the same state
// (the benchmark step of calipso_hip_newton_step).  info: batch x 8 [step_size, step_size_t, refinement rounds, factorisations, merit and violation of the accepted
// candidate, exit kind (2: inner-loop exit of solve.jl:165, no step), optimality error] of the LAST step; status as calipso_hip_smallnewton_solve (0: stepped).
int32_t calipso_hip_smallnewton_steps(calipso_hip_smallnewton* s, int32_t count, int32_t advance, double* info, int32_t* status, double* ms) {
    if (!s || count < 0) return CALIPSO_ERR_ARGUMENT;
    const int rc = launch(s, MODE_STEPS, count, advance);
    if (rc < 0) return rc;
    if (info) SK(hipMemcpy(info, s->info, sizeof(double) * (size_t)s->batch * IN_COUNT, hipMemcpyDeviceToHost));
    if (status) SK(hipMemcpy(status, s->status, sizeof(int) * (size_t)s->batch, hipMemcpyDeviceToHost));
    if (ms) *ms = s->last_ms;
    return CALIPSO_OK;
}

// differentiate!(solver) for every instance in ONE launch (differentiate.jl:1-61) at the resident points (after calipso_hip_smallnewton_solve): one factorisation of
// the condensed matrix with the regularisation solve! left, then search_direction_symmetric! per column of dR/dtheta and sensitivity = -1.0 * the result.
// jacobian_parameters: batch x (N x p), column-major per instance (host), or ONE N x p matrix for all instances (shared != 0: the model of an MPC loop is the same
// for every problem); sensitivity: batch x (N x p).  status[k]: 0, or 1 when the factorisation's inertia is not (nx, ne + nc, 0).
int32_t calipso_hip_smallnewton_differentiate(calipso_hip_smallnewton* s, int64_t p, int32_t shared, const double* jacobian_parameters, double* sensitivity, int32_t* status, double* ms) {
    if (!s || p < 1 || p > (1 << 20) || !jacobian_parameters || !sensitivity) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    const Dm d = dims_of(s);
    const size_t need = (size_t)s->batch * (size_t)d.N * (size_t)p;
    if (need > s->cap_diff) {
        if (s->rtheta) (void)hipFree(s->rtheta);
        if (s->sens) (void)hipFree(s->sens);
        s->rtheta = s->sens = nullptr; s->cap_diff = 0;
        SK(hipMalloc((void**)&s->rtheta, sizeof(double) * need));
        SK(hipMalloc((void**)&s->sens, sizeof(double) * need));
        s->cap_diff = need;
    }
    SK(hipMemcpyAsync(s->rtheta, jacobian_parameters, sizeof(double) * (shared ? (size_t)d.N * (size_t)p : need), hipMemcpyHostToDevice, s->stream));
    s->diff_shared = shared != 0;
    const int rc = launch(s, MODE_DIFF, (int)p, 0);
    if (rc < 0) return rc;
    SK(hipMemcpy(sensitivity, s->sens, sizeof(double) * need, hipMemcpyDeviceToHost));
    if (status) SK(hipMemcpy(status, s->status, sizeof(int) * (size_t)s->batch, hipMemcpyDeviceToHost));
    if (ms) *ms = s->last_ms;
    return CALIPSO_OK;
}

// phase clocks of instance 0 in the last launch, microseconds (a build with -DSN_TRACE; zeros otherwise): [0] evaluation + residual + norms, [1] inertia logic, [2] cone
// weights + assembly of S, [3] LDL^T, [4] first condensed solve, [5] refinement, [6] cone search + candidate, [7] candidate merit + line search, [8] accept, [9] of the LDL^T: the panels (one wavefront), [3] then holds its trailing updates, [11] between steps
// out = {threads per instance, LDS bytes per instance, instances a compute unit holds (the runtime's occupancy query for the kernel launch() would pick), compute units}
int32_t calipso_hip_debug_smallnewton_describe(calipso_hip_smallnewton* s, double out[4]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    const int nt = sn_threads(s); const bool soc = !s->soc_dim.empty();
    const void* f = nt == 64 ? (soc ? (const void*)t64::k_smallnewton<true> : (const void*)t64::k_smallnewton<false>)
                  : nt == 128 ? (soc ? (const void*)t128::k_smallnewton<true> : (const void*)t128::k_smallnewton<false>)
                              : (soc ? (const void*)t256::k_smallnewton<true> : (const void*)t256::k_smallnewton<false>);
    int per = 0, cus = 0;
    SK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, f, nt, s->lds_bytes));
    SK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
    out[0] = nt; out[1] = (double)s->lds_bytes; out[2] = per; out[3] = cus;
    return CALIPSO_OK;
}

int32_t calipso_hip_debug_smallnewton_profile(calipso_hip_smallnewton* s, double out[12]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    SK(hipMemcpy(out, s->prof, sizeof(double) * 12, hipMemcpyDeviceToHost));
    return CALIPSO_OK;
}

}  // extern "C"
