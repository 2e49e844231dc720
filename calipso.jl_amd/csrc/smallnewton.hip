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
    long long* cnt = nullptr; int* status = nullptr;
    int trace_rows = 0;
    size_t lds_bytes = 0;
    double last_ms = 0.0;
    std::string err;
};

namespace {
using calipso::Options;
typedef calipso_hip_smallnewton SN;

#ifndef SN_NT
#define SN_NT 256        // threads per workgroup = per instance (bench/small_newton_phases.sh builds other values)
#endif
constexpr int NT = SN_NT, NW = NT / 64;
static_assert(NT == 64 || NT == 128 || NT == 256, "one, two or four wavefronts per instance");
#ifndef SN_JB
#define SN_JB 8          // columns per panel of the LDL^T (bench/small_newton_phases.sh builds other values)
#endif
enum { SC_KAPPA = 0, SC_TAU, SC_RHO, SC_EP, SC_EPLAST, SC_ED, SC_EQV, SC_CPV, SC_F, SC_COUNT = 16 };
enum { CN_TOTAL = 0, CN_OUTER, CN_INNER, CN_FACT, CN_RFAIL, CN_RMAX, CN_RLAST, CN_STEPS, CN_FILTER, CN_TRACE, CN_COUNT = 16 };
enum { IN_STEP = 0, IN_STEP_T, IN_ROUNDS, IN_NFACT, IN_MH, IN_THETAH, IN_EXIT, IN_OPT, IN_COUNT = 8 };
enum { MODE_SOLVE = 0, MODE_STEPS = 1 };

struct Dm {
    int nx, ne, nc, m, n, N, ldz, lds, q, nsoc, wsz, maxd;      // q nonnegative entries first, then nsoc second-order cones (contiguous ranges); wsz = sum of dim^2; ldz: leading dimension of Z in LDS (odd: conflict-free column walks), lds: of S / Lxx
    __host__ __device__ int orr() const { return nx; }
    __host__ __device__ int os() const { return nx + ne; }
    __host__ __device__ int oy() const { return nx + ne + nc; }
    __host__ __device__ int oz() const { return nx + ne + nc + ne; }
    __host__ __device__ int ot() const { return nx + ne + nc + ne + nc; }
};

// LDS carve-up (offsets in doubles): the same function sizes the launch on the host and places the pointers on the device
struct Lay { int Lxx, Z, S, q, bh, lam, sol, cand, step, res, rerr, corr, rsym, fx, gzx, gh, ghc, cprod, bgrad, wz, wsoc, bsoc, vsoc, D, Dinv, xb, t1, t2, ycol, red, total; };
__host__ __device__ inline Lay layout(const Dm& d) {
    Lay L; int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 1) & ~1; return at; };
    L.Lxx = take(d.lds * d.nx); L.Z = take(d.ldz * d.nx); L.S = take(d.lds * d.nx);
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
};

// ---- workgroup-wide reductions (every thread calls; all get the result) ----------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
template <int K> __device__ __forceinline__ void block_sum(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { const double s = wave_sum(v[k]); if (lane == 0) red[k * 4 + wave] = s; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { double t = red[k * 4]; for (int w = 1; w < NW; ++w) t += red[k * 4 + w]; v[k] = t; }
}
template <int K> __device__ __forceinline__ void block_max(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { const double s = wave_max(v[k]); if (lane == 0) red[k * 4 + wave] = s; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { double t = red[k * 4]; for (int w = 1; w < NW; ++w) t = fmax(t, red[k * 4 + w]); v[k] = t; }
}
// KS sums and KM maxima with ONE pair of barriers
template <int KS, int KM> __device__ __forceinline__ void block_sum_max(double (&sv)[KS], double (&mv)[KM], double* red) {
    static_assert((KS + KM) * 4 <= 64, "reduction scratch");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KS; ++k) { const double s = wave_sum(sv[k]); if (lane == 0) red[k * 4 + wave] = s; }
#pragma unroll
    for (int k = 0; k < KM; ++k) { const double s = wave_max(mv[k]); if (lane == 0) red[(KS + k) * 4 + wave] = s; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KS; ++k) { double t = red[k * 4]; for (int w = 1; w < NW; ++w) t += red[k * 4 + w]; sv[k] = t; }
#pragma unroll
    for (int k = 0; k < KM; ++k) { double t = red[(KS + k) * 4]; for (int w = 1; w < NW; ++w) t = fmax(t, red[(KS + k) * 4 + w]); mv[k] = t; }
}
// |v| with NaN -> +inf: a NaN in a residual must FAIL the refinement's `norm <= tolerance` test (Julia's norm is NaN there), not slip through fmax
__device__ __forceinline__ double nabs(double v) { return v != v ? __longlong_as_double(0x7ff0000000000000LL) : fabs(v); }

// y[r] = sum_c M[r + c ld] x[c] (+ add[r]), r < rows: a thread per row (consecutive rows in consecutive lanes: conflict-free), x broadcast
__device__ __forceinline__ void mv_n(const double* M, int ld, int rows, int cols, const double* x, double* y, const double* add) {
    // (one thread walks a whole row: with a single accumulator every multiply-add waits for its own LDS round trip — 49 columns were 3.3 us; four accumulators over
    // batches of eight columns keep eight loads in flight)
    for (int r = threadIdx.x; r < rows; r += NT) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int c = 0;
        for (; c + 8 <= cols; c += 8) {
            double mv[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { mv[q] = M[r + (c + q) * ld]; xv[q] = x[c + q]; }
            a0 += mv[0] * xv[0]; a1 += mv[1] * xv[1]; a2 += mv[2] * xv[2]; a3 += mv[3] * xv[3];
            a0 += mv[4] * xv[4]; a1 += mv[5] * xv[5]; a2 += mv[6] * xv[6]; a3 += mv[7] * xv[7];
        }
        for (; c < cols; ++c) a0 += M[r + c * ld] * x[c];
        const double a = (a0 + a1) + (a2 + a3);
        y[r] = add ? a + add[r] : a;
    }
}
// y[c] = sum_r M[r + c ld] x[r] (+ add[c]), c < cols: a thread per column (ld odd: conflict-free)
__device__ __forceinline__ void mv_t(const double* M, int ld, int rows, int cols, const double* x, double* y, const double* add) {
    for (int c = threadIdx.x; c < cols; c += NT) {
        const double* col = M + c * ld;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = 0;
        for (; r + 8 <= rows; r += 8) {
            double mv[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { mv[q] = col[r + q]; xv[q] = x[r + q]; }
            a0 += mv[0] * xv[0]; a1 += mv[1] * xv[1]; a2 += mv[2] * xv[2]; a3 += mv[3] * xv[3];
            a0 += mv[4] * xv[4]; a1 += mv[5] * xv[5]; a2 += mv[6] * xv[6]; a3 += mv[7] * xv[7];
        }
        for (; r < rows; ++r) a0 += col[r] * x[r];
        const double a = (a0 + a1) + (a2 + a3);
        y[c] = add ? a + add[c] : a;
    }
}

// second_order_vector_inverse(u, x) (cones/second_order.jl:50-60): arrow(u)^-1 x, the reference's operations in its order
__device__ __forceinline__ void arrow_inverse(int n, const double* u, const double* x, double* out) {
    double uu = 0.0;
    for (int i = 1; i < n; ++i) uu += u[i] * u[i];
    const double alpha = -1.0 / (u[0] * u[0]) * uu;
    const double beta = 1.0 / (1.0 + alpha);
    double d0 = 0.0;
    for (int i = 1; i < n; ++i) d0 += (u[i] / u[0]) * x[i];
    const double x0_1 = x[0] - d0;
    double d1 = 0.0;
    for (int i = 1; i < n; ++i) { const double o = x[i] - beta * ((u[i] / u[0]) * x0_1); out[i] = o; d1 += (u[i] / u[0]) * o; }
    const double x2_1 = x[0] - d1;
    out[0] = 1.0 / u[0] * x2_1;
    for (int i = 1; i < n; ++i) out[i] = 1.0 / u[0] * out[i];
}

// v_readlane of a double: the value lane `src` (wave-uniform) holds
__device__ __forceinline__ double rl(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

template <int R> __device__ __forceinline__ double pick(const double (&x)[R], int cu) {
    if constexpr (R == 1) return x[0];
    else return cu == 0 ? x[0] : x[1];
}

// 1 / d on the pivot chain: v_rcp_f64 and two Newton steps (an IEEE division is ~25 dependent instructions; the pivots are exact zeros only for singular matrices,
// which the inertia test reports: d = 0 gives inf here as the division does)
__device__ __forceinline__ double recip(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    double r = r0;
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return d == 0.0 ? r0 : r;
}

template <bool SOC> struct CtxT {
    Dm d; const Options* o;
    double *Lxx, *Z, *S, *q, *bh, *lam, *sol, *cand, *step, *res, *rerr, *corr, *rsym, *fx, *gzx, *gh, *ghc, *cprod, *bgrad, *wz, *wsoc, *bsoc, *vsoc, *D, *Dinv, *xb, *t1, *t2, *ycol, *red;
    const int *soc_start, *soc_dim, *soc_woff;
    double* filt;                                   // global: [pairs theta | pairs merit | cache theta | cache merit | saved theta | saved merit], max_filter each
    // uniform scalars (every thread holds the same values)
    double kappa, tau, rho, ep, ep_last, ed, fcur, fcand, eqv, cpv, omega_y, kyy;
    long long filter_index, nfact_total, rfail, rmax, rlast, nsteps;
    int tid, mf;      // mf = options.max_filter
#ifdef SN_TRACE
    long long tph[12]; long long tlast;
    __device__ __forceinline__ void stamp(int k) { const long long t = wall_clock64(); tph[k] += t - tlast; tlast = t; }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif

    // ---- evaluate! of the QP (qp.hip): which = the point (sol / cand) ------------------------------------------------------------------------
    __device__ __forceinline__ double eval_objective(const double* p) {        // f = 1/2 x'Lxx x + q'x   (uses xb as scratch)
        mv_n(Lxx, d.lds, d.nx, d.nx, p, xb, nullptr);
        __syncthreads();
        double v[2] = {0.0, 0.0};
        for (int i = tid; i < d.nx; i += NT) { v[0] += p[i] * xb[i]; v[1] += q[i] * p[i]; }
        block_sum(v, red);
        return 0.5 * v[0] + v[1];
    }
    __device__ __forceinline__ void eval_constraints(const double* p, double* out) {      // [g; h] = [A; -G] x + [-b; hvec]
        mv_n(Z, d.ldz, d.m, d.nx, p, out, bh);
        __syncthreads();
    }
    __device__ __forceinline__ void eval_gradients(const double* p) {                     // fx = Lxx x + q ; gzx = A'y + (-G)'z
        mv_n(Lxx, d.lds, d.nx, d.nx, p, fx, q);
        mv_t(Z, d.ldz, d.m, d.nx, p + d.oy(), gzx, nullptr);
        __syncthreads();
    }

    // cone_target (cone.jl:55-59): 1 for nonnegative entries and for the first entry of a second-order cone, 0 for its other entries
    __device__ __forceinline__ double target(int i) const {
        if (i < d.q) return 1.0;
        for (int j = 0; SOC && j < d.nsoc; ++j) if (i == soc_start[j]) return 1.0;
        return 0.0;
    }

    // ---- cone!(product): s o t ----------------------------------------------------------------------------------------
    __device__ __forceinline__ void cone_product(const double* p) {
        for (int i = tid; i < d.q; i += NT) cprod[i] = p[d.os() + i] * p[d.ot() + i];
        for (int j = tid; SOC && j < d.nsoc; j += NT) {                 // second_order_product (second_order.jl:17)
            const int st = soc_start[j], dm = soc_dim[j];
            const double* a = p + d.os() + st; const double* b = p + d.ot() + st;
            double dot = 0.0;
            for (int e = 0; e < dm; ++e) dot += a[e] * b[e];
            cprod[st] = dot;
            for (int e = 1; e < dm; ++e) cprod[st + e] = a[0] * b[e] + b[0] * a[e];
        }
        __syncthreads();
    }

    // ---- H v  (residual_jacobian_variables.jl:1-108, block form; regularisation included) -> out ----------------------------------------------
    __device__ __forceinline__ void Hmul(const double* v, double* out) {
        // x rows: (Lxx + ep) vx + Z'[vy; vz]   — two passes (t-products need all of vy, vz; n-products all of vx)
        mv_n(Lxx, d.lds, d.nx, d.nx, v, xb, nullptr);
        mv_n(Z, d.ldz, d.m, d.nx, v, t2, nullptr);                      // [A; -G] vx
        __syncthreads();
        for (int c = tid; c < d.nx; c += NT) {
            const double* col = Z + c * d.ldz; const double* vy = v + d.oy();
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int r = 0;
            for (; r + 8 <= d.m; r += 8) {
                double mv[8], xv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { mv[q] = col[r + q]; xv[q] = vy[r + q]; }
                a0 += mv[0] * xv[0]; a1 += mv[1] * xv[1]; a2 += mv[2] * xv[2]; a3 += mv[3] * xv[3];
                a0 += mv[4] * xv[4]; a1 += mv[5] * xv[5]; a2 += mv[6] * xv[6]; a3 += mv[7] * xv[7];
            }
            for (; r < d.m; ++r) a0 += col[r] * vy[r];
            out[c] = (xb[c] + ep * v[c]) + ((a0 + a1) + (a2 + a3));
        }
        for (int i = tid; i < d.ne; i += NT) {
            out[d.orr() + i] = (rho + ep) * v[d.orr() + i] - v[d.oy() + i];
            out[d.oy() + i] = t2[i] - v[d.orr() + i] + (0.0 - ed) * v[d.oy() + i];
        }
        for (int i = tid; i < d.nc; i += NT) {
            out[d.os() + i] = (0.0 + ep) * v[d.os() + i] - v[d.oz() + i] - v[d.ot() + i];
            out[d.oz() + i] = t2[d.ne + i] - v[d.os() + i] + (0.0 - ed) * v[d.oz() + i];
            if (i < d.q) { const double sl = sol[d.os() + i], t = sol[d.ot() + i]; out[d.ot() + i] = t * v[d.os() + i] + (sl - ed) * v[d.ot() + i]; }
        }
        for (int j = tid; SOC && j < d.nsoc; j += NT) {                 // arrow(t) v_s + (arrow(s) - ed I) v_t
            const int st = soc_start[j], dm = soc_dim[j];
            const double* sl = sol + d.os() + st; const double* t = sol + d.ot() + st;
            const double* vs = v + d.os() + st; const double* vt = v + d.ot() + st;
            double acc = t[0] * vs[0] + (sl[0] - ed) * vt[0];
            for (int e = 1; e < dm; ++e) acc += t[e] * vs[e] + sl[e] * vt[e];
            out[d.ot() + st] = acc;
            for (int e = 1; e < dm; ++e) out[d.ot() + st + e] = (t[e] * vs[0] + sl[e] * vt[0]) + (t[0] * vs[e] + (sl[0] - ed) * vt[e]);
        }
        __syncthreads();
    }

    template <int RP> __device__ __forceinline__ void panel_(int j0, int jb, int lane, double* pan) {
        constexpr int JB = SN_JB;
                            double pr[RP][JB];
        #pragma unroll
                            for (int r = 0; r < RP; ++r) {
                                const int i = j0 + lane + 64 * r;
        #pragma unroll
                                for (int c = 0; c < JB; ++c) pr[r][c] = (i < d.nx && c < jb && j0 + c <= i) ? S[i + (j0 + c) * d.lds] : 0.0;
                            }
        #pragma unroll
                            for (int u = 0; u < JB; ++u) {
                                if (u < jb) {                                           // (uniform)
                                    const double dj = rl(pr[0][u], u);                  // row j0 + u sits in lane u, chunk 0
                                    const double rinv = recip(dj);
                                    if (lane == 0) { D[j0 + u] = dj; Dinv[j0 + u] = rinv; }
                                    double yk[JB];
        #pragma unroll
                                    for (int c = u + 1; c < JB; ++c) yk[c] = rl(pr[0][u], c);      // raw entries of the pivot column in the panel's own rows
        #pragma unroll
                                    for (int r = 0; r < RP; ++r) {
                                        const int i = j0 + lane + 64 * r;
                                        const double y = pr[r][u];
                                        if (i > j0 + u && i < d.nx) pan[u * d.nx + i] = y;
                                        const double li = y * rinv;
        #pragma unroll
                                        for (int c = u + 1; c < JB; ++c) pr[r][c] -= li * yk[c];     // (entries above the diagonal take garbage: never read)
                                        if (i > j0 + u) pr[r][u] = li;
                                    }
                                }
                            }
        #pragma unroll
                            for (int r = 0; r < RP; ++r) {
                                const int i = j0 + lane + 64 * r;
        #pragma unroll
                                for (int c = 0; c < JB; ++c) if (i < d.nx && c < jb && j0 + c < i) S[i + (j0 + c) * d.lds] = pr[r][c];
                            }
    }

    // ---- factorize! + compute_inertia! of the condensed matrix for the current (ep, ed): returns true when the inertia is (nx, ne + nc, 0) ---------
    __device__ __forceinline__ bool factorize(int& zero_pivots) {
        stamp(1);
        kyy = -1.0 / (rho + ep) + (0.0 - ed);
        omega_y = -1.0 / kyy;
        int pos = 0, nonpos = 0, zero = 0;
        if (d.ne > 0) { if (kyy > 0.0) pos += d.ne; else nonpos += d.ne; if (kyy == 0.0) zero += d.ne; }
        // nonnegative entries: K_zz = -Sb / (T + Sb P) + D with Sb = s - ed, T = t, P = ep, D = -ed   (residual_jacobian_variables.jl:139-143)
        double cnt[3] = {0.0, 0.0, 0.0};
        for (int i = tid; i < d.q; i += NT) {
            const double Sb = sol[d.os() + i] - ed, T = sol[d.ot() + i];
            const double kz = -1.0 * Sb / (T + Sb * ep) + (0.0 - ed);
            wz[i] = -1.0 / kz;
            if (kz > 0.0) cnt[0] += 1.0; else cnt[1] += 1.0;
            if (kz == 0.0) cnt[2] += 1.0;
        }
        // second-order cones (residual_jacobian_variables.jl:145-164): the block  B = -(Cs + Cbar_t P)^-1 Cbar_t + D  column by column through the closed-form arrow
        // inverse (quirk: second_order_matrix_inverse uses only the FIRST ROW of its matrix, second_order.jl:63-65), then what a factorisation of triu(K) sees — the upper
        // triangle mirrored —, its LDL^T in the natural order (the pivots count towards the inertia) and Omega = -B_sym^-1.  One thread per cone.
        for (int j = tid; SOC && j < d.nsoc; j += NT) {
            const int st = soc_start[j], dm = soc_dim[j];
            double* B = bsoc + soc_woff[j]; double* W = wsoc + soc_woff[j];
            double* u = vsoc + 4 * d.maxd * j; double* col = u + d.maxd; double* o = col + d.maxd; double* dg = o + d.maxd;
            const double* sl = sol + d.os() + st; const double* t = sol + d.ot() + st;
            for (int b = 0; b < dm; ++b) u[b] = t[b] + (sl[b] - (b == 0 ? ed : 0.0)) * ep;
            for (int i = 0; i < dm; ++i) {
                for (int a = 0; a < dm; ++a) col[a] = (a == i ? sl[0] - ed : 0.0) + ((i == 0 && a > 0) ? sl[a] : 0.0) + ((a == 0 && i > 0) ? sl[i] : 0.0);      // column i of arrow(s) - ed I
                arrow_inverse(dm, u, col, o);
                for (int a = 0; a < dm; ++a) B[a + i * dm] = -o[a] + (a == i ? (0.0 - ed) : 0.0);
            }
            for (int a = 0; a < dm; ++a) for (int b = 0; b < a; ++b) B[a + b * dm] = B[b + a * dm];      // triu mirrored
            // LDL^T of B_sym in place (unit lower in the strict lower triangle, pivots in dg)
            for (int k = 0; k < dm; ++k) {
                double dk = B[k + k * dm];
                for (int p2 = 0; p2 < k; ++p2) dk -= B[k + p2 * dm] * B[k + p2 * dm] * dg[p2];
                dg[k] = dk;
                if (dk > 0.0) cnt[0] += 1.0; else cnt[1] += 1.0;
                if (dk == 0.0) cnt[2] += 1.0;
                for (int i = k + 1; i < dm; ++i) {
                    double v = B[i + k * dm];
                    for (int p2 = 0; p2 < k; ++p2) v -= B[i + p2 * dm] * B[k + p2 * dm] * dg[p2];
                    B[i + k * dm] = v / dk;
                }
            }
            // Omega = -(L D L')^-1, column by column
            for (int c0 = 0; c0 < dm; ++c0) {
                for (int a = 0; a < dm; ++a) col[a] = a == c0 ? 1.0 : 0.0;
                for (int k = 0; k < dm; ++k) for (int i = k + 1; i < dm; ++i) col[i] -= B[i + k * dm] * col[k];
                for (int k = 0; k < dm; ++k) col[k] /= dg[k];
                for (int k = dm - 1; k >= 0; --k) for (int i = 0; i < k; ++i) col[i] -= B[k + i * dm] * col[k];
                for (int a = 0; a < dm; ++a) W[a + c0 * dm] = -col[a];
            }
        }
        __syncthreads();             // (wz, wsoc are read by every thread below; the pivot-sign counts of this part join those of D in ONE reduction at the end)
        // S(i, j), i >= j: what triu(K) holds of the Hessian (Lxx[j, i]) + ep on the diagonal + sum_k Z[k, i] Omega_k Z[k, j]
        const int ntri = d.nx * (d.nx + 1) / 2;
        for (int e = tid; e < ntri; e += NT) {
            // e -> (i, j) of the lower triangle, row-major
            int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= e) ++i;
            while (i * (i + 1) / 2 > e) --i;
            const int j = e - i * (i + 1) / 2;
            double a = 0.0;
            const double* zi = Z + i * d.ldz; const double* zj = Z + j * d.ldz;
            {
                double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
                int k = 0;
                for (; k + 4 <= d.ne; k += 4) { e0 += zi[k] * zj[k]; e1 += zi[k + 1] * zj[k + 1]; e2 += zi[k + 2] * zj[k + 2]; e3 += zi[k + 3] * zj[k + 3]; }
                for (; k < d.ne; ++k) e0 += zi[k] * zj[k];
                a += omega_y * ((e0 + e1) + (e2 + e3));
                const double* ci = zi + d.ne; const double* cj = zj + d.ne;
                e0 = e1 = e2 = e3 = 0.0;
                for (k = 0; k + 4 <= d.q; k += 4) { e0 += ci[k] * wz[k] * cj[k]; e1 += ci[k + 1] * wz[k + 1] * cj[k + 1]; e2 += ci[k + 2] * wz[k + 2] * cj[k + 2]; e3 += ci[k + 3] * wz[k + 3] * cj[k + 3]; }
                for (; k < d.q; ++k) e0 += ci[k] * wz[k] * cj[k];
                a += (e0 + e1) + (e2 + e3);
            }
            for (int c0 = 0; SOC && c0 < d.nsoc; ++c0) {
                const int st = d.ne + soc_start[c0], dm = soc_dim[c0];
                const double* W = wsoc + soc_woff[c0];
                for (int b = 0; b < dm; ++b) {
                    double wv = 0.0;
                    for (int a2 = 0; a2 < dm; ++a2) wv += zi[st + a2] * W[a2 + b * dm];
                    a += wv * zj[st + b];
                }
            }
            double v = Lxx[j + i * d.lds] + a;
            if (i == j) v += ep;
            S[i + j * d.lds] = v;
        }
        __syncthreads();
        stamp(2);
        // Blocked right-looking LDL^T in place (unit lower L below the diagonal), panels of 8 columns.  A panel is factored by ONE wavefront in registers: lane l holds
        // the panel entries of the rows j0 + l (+ 64, 128, 192), the pivot row's entries travel by v_readlane — no barrier and no LDS round trip between the 8 pivots —
        // and leaves the raw (unscaled) pivot columns in `ycol` (8 x nx); then all threads apply the panel to the trailing matrix, entry by entry in pivot order
        // (S(i, k) -= l_i y_k for the panel's pivots in turn: the arithmetic of the column-by-column algorithm), two barriers per panel instead of one per pivot.
        {
            constexpr int JB = SN_JB;
            const int lane = tid & 63, wave = tid >> 6;
            const int ti = tid >> 4, tk = tid & 15;
            double* pan = ycol;
            for (int j0 = 0; j0 < d.nx; j0 += JB) {
                const int jb = d.nx - j0 < JB ? d.nx - j0 : JB;
                stamp(3);
                if (wave == 0) { if (d.nx - j0 <= 64) panel_<1>(j0, jb, lane, pan); else panel_<2>(j0, jb, lane, pan); }
                __syncthreads();
                stamp(9);
                const int base = j0 + jb;
                for (int i = base + ti; i < d.nx; i += NT / 16) {
                    double li[JB];
#pragma unroll
                    for (int u = 0; u < JB; ++u) li[u] = u < jb ? S[i + (j0 + u) * d.lds] : 0.0;
                    // three entries of the row at a time: their chains of 8 dependent multiply-adds interleave (one entry alone is ~230 cycles of latency)
                    constexpr int KU = 3;
                    for (int k0 = base + tk; k0 <= i; k0 += 16 * KU) {
                        double v[KU];
#pragma unroll
                        for (int q = 0; q < KU; ++q) { const int k = k0 + 16 * q; v[q] = k <= i ? S[i + k * d.lds] : 0.0; }
#pragma unroll
                        for (int u = 0; u < JB; ++u) {
#pragma unroll
                            for (int q = 0; q < KU; ++q) { const int k = k0 + 16 * q; if (u < jb && k <= i) v[q] -= li[u] * pan[u * d.nx + k]; }
                        }
#pragma unroll
                        for (int q = 0; q < KU; ++q) { const int k = k0 + 16 * q; if (k <= i) S[i + k * d.lds] = v[q]; }
                    }
                }
                __syncthreads();
            }
        }
        double c2[3] = {cnt[0], cnt[1], cnt[2]};
        for (int i = tid; i < d.nx; i += NT) { const double dv = D[i]; if (dv > 0.0) c2[0] += 1.0; else c2[1] += 1.0; if (dv == 0.0) c2[2] += 1.0; }
        block_sum(c2, red);
        pos += (int)c2[0]; nonpos += (int)c2[1]; zero += (int)c2[2];
        nfact_total += 1;
        stamp(3);
        zero_pivots = zero;
        return zero == 0 && pos == d.nx && nonpos == d.ne + d.nc;
    }

    // ---- xb <- S^-1 xb with the factors in S / Dinv: one wavefront, lane-owned rows in registers, the pivot entry by v_readlane (no barrier inside) ----------
    __device__ __forceinline__ void solve_S() { if (d.nx <= 64) solve_S_<1>(); else solve_S_<2>(); }
    template <int RPL> __device__ __forceinline__ void solve_S_() {
        if (tid < 64) {
            constexpr int PF = 8;              // rows per lane (nx <= 256); pivots whose column entries are fetched together (one LDS latency per PF pivots)
            double x[RPL];
#pragma unroll
            for (int u = 0; u < RPL; ++u) { const int i = tid + 64 * u; x[u] = i < d.nx ? xb[i] : 0.0; }
            const int nchunk = (d.nx + 63) >> 6;        // (uniform) chunks of 64 rows in use
            for (int k0 = 0; k0 < d.nx; k0 += PF) {     // L u = b, PF columns at a time
                double l[RPL][PF];
#pragma unroll
                for (int u = 0; u < RPL; ++u) {
                    const int i = tid + 64 * u;
#pragma unroll
                    for (int q = 0; q < PF; ++q) l[u][q] = (u < nchunk && i < d.nx && k0 + q < d.nx && i > k0 + q) ? S[i + (k0 + q) * d.lds] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const int k = k0 + q;
                    if (k < d.nx) {                      // (uniform)
                        const int cu = k >> 6, src = k & 63;
                        const double xs = pick<RPL>(x, cu);
                        const double xk = rl(xs, src);
#pragma unroll
                        for (int u = 0; u < RPL; ++u) x[u] -= l[u][q] * xk;      // (zero multipliers for the rows at or above the pivot)
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RPL; ++u) { const int i = tid + 64 * u; if (i < d.nx) x[u] *= Dinv[i]; }
            for (int k1 = d.nx; k1 > 0; k1 -= PF) {      // L' v = u, from the last column
                double l[RPL][PF];
#pragma unroll
                for (int u = 0; u < RPL; ++u) {
                    const int i = tid + 64 * u;
#pragma unroll
                    for (int q = 0; q < PF; ++q) { const int k = k1 - 1 - q; l[u][q] = (u < nchunk && k >= 0 && i < k) ? S[k + i * d.lds] : 0.0; }
                }
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const int k = k1 - 1 - q;
                    if (k >= 0) {
                        const int cu = k >> 6, src = k & 63;
                        const double xs = pick<RPL>(x, cu);
                        const double xk = rl(xs, src);
#pragma unroll
                        for (int u = 0; u < RPL; ++u) x[u] -= l[u][q] * xk;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RPL; ++u) { const int i = tid + 64 * u; if (i < d.nx) xb[i] = x[u]; }
        }
        __syncthreads();
    }

    // ---- search_direction_symmetric!(out, r): condensed right-hand side, solve, back-substitution, recovery (search_direction.jl:25-104) -----------------
    __device__ __forceinline__ void search_direction_symmetric(const double* r, double* out) {
        const double hrr = rho + ep;
        for (int i = tid; i < d.nx; i += NT) rsym[i] = r[i];
        for (int i = tid; i < d.ne; i += NT) { const double v = r[d.oy() + i] + r[d.orr() + i] / hrr; rsym[d.nx + i] = v; t1[i] = omega_y * v; }
        for (int i = tid; i < d.q; i += NT) {
            const double Sb = sol[d.os() + i] - ed, T = sol[d.ot() + i];
            const double v = r[d.oz() + i] + (r[d.ot() + i] + Sb * r[d.os() + i]) / (T + Sb * ep);
            rsym[d.nx + d.ne + i] = v; t1[d.ne + i] = wz[i] * v;
        }
        for (int j = tid; SOC && j < d.nsoc; j += NT) {                 // residual.jl:84-99: b_z += (Cs + Cbar_t P)^-1 (r_t + Cbar_t r_s), then Omega b_z of the cone
            const int st = soc_start[j], dm = soc_dim[j];
            double* u = vsoc + 4 * d.maxd * j; double* v = u + d.maxd; double* o = v + d.maxd;
            const double* sl = sol + d.os() + st; const double* t = sol + d.ot() + st;
            const double* rs = r + d.os() + st; const double* rt = r + d.ot() + st;
            for (int b = 0; b < dm; ++b) u[b] = t[b] + (sl[b] - (b == 0 ? ed : 0.0)) * ep;
            double acc = (sl[0] - ed) * rs[0];
            for (int e = 1; e < dm; ++e) acc += sl[e] * rs[e];
            v[0] = acc + rt[0];
            for (int e = 1; e < dm; ++e) v[e] = (sl[e] * rs[0] + (sl[0] - ed) * rs[e]) + rt[e];
            arrow_inverse(dm, u, v, o);
            double* bz = rsym + d.nx + d.ne + st;
            for (int e = 0; e < dm; ++e) bz[e] = r[d.oz() + st + e] + o[e];
            const double* W = wsoc + soc_woff[j];
            for (int a2 = 0; a2 < dm; ++a2) { double wv = 0.0; for (int b = 0; b < dm; ++b) wv += W[a2 + b * dm] * bz[b]; t1[d.ne + st + a2] = wv; }
        }
        __syncthreads();
        mv_t(Z, d.ldz, d.m, d.nx, t1, xb, rsym);                 // b_x + [A; -G]' (Omega b_m)
        __syncthreads();
        solve_S();                                              // dx
        mv_n(Z, d.ldz, d.m, d.nx, xb, t2, nullptr);              // [A; -G] dx
        __syncthreads();
        for (int i = tid; i < d.nx; i += NT) out[i] = xb[i];
        for (int i = tid; i < d.ne; i += NT) {
            const double dy = -omega_y * (rsym[d.nx + i] - t2[i]);
            out[d.oy() + i] = dy;
            out[d.orr() + i] = (r[d.orr() + i] + dy) / hrr;
        }
        for (int i = tid; i < d.q; i += NT) {
            const double dz = -wz[i] * (rsym[d.nx + d.ne + i] - t2[d.ne + i]);
            const double Sb = sol[d.os() + i] - ed, T = sol[d.ot() + i];
            const double ds = (r[d.ot() + i] + Sb * (r[d.os() + i] + dz)) / (T + Sb * ep);
            out[d.oz() + i] = dz;
            out[d.os() + i] = ds;
            out[d.ot() + i] = (r[d.ot() + i] - T * ds) / Sb;
        }
        for (int j = tid; SOC && j < d.nsoc; j += NT) {                 // search_direction.jl:80-101 for a second-order cone
            const int st = soc_start[j], dm = soc_dim[j];
            double* u = vsoc + 4 * d.maxd * j; double* v = u + d.maxd; double* o = v + d.maxd; double* ct = o + d.maxd;
            const double* sl = sol + d.os() + st; const double* t = sol + d.ot() + st;
            const double* rs = r + d.os() + st; const double* rt = r + d.ot() + st;
            const double* W = wsoc + soc_woff[j];
            const double* bz = rsym + d.nx + d.ne + st;
            double* dz = out + d.oz() + st; double* ds = out + d.os() + st; double* dt = out + d.ot() + st;
            for (int a2 = 0; a2 < dm; ++a2) { double wv = 0.0; for (int b = 0; b < dm; ++b) wv += W[a2 + b * dm] * (bz[b] - t2[d.ne + st + b]); dz[a2] = -wv; }
            for (int b = 0; b < dm; ++b) u[b] = t[b] + (sl[b] - (b == 0 ? ed : 0.0)) * ep;
            double acc = (sl[0] - ed) * (rs[0] + dz[0]);
            for (int e = 1; e < dm; ++e) acc += sl[e] * (rs[e] + dz[e]);
            v[0] = rt[0] + acc;
            for (int e = 1; e < dm; ++e) v[e] = rt[e] + (sl[e] * (rs[0] + dz[0]) + (sl[0] - ed) * (rs[e] + dz[e]));
            arrow_inverse(dm, u, v, o);
            for (int e = 0; e < dm; ++e) ds[e] = o[e];
            for (int b = 0; b < dm; ++b) ct[b] = sl[b] - (b == 0 ? ed : 0.0);          // first row of Cbar_t = arrow(s) - ed I
            double a0 = t[0] * ds[0];
            for (int e = 1; e < dm; ++e) a0 += t[e] * ds[e];
            v[0] = rt[0] - a0;
            for (int e = 1; e < dm; ++e) v[e] = rt[e] - (t[e] * ds[0] + t[0] * ds[e]);
            arrow_inverse(dm, ct, v, o);
            for (int e = 0; e < dm; ++e) dt[e] = o[e];
        }
        __syncthreads();
    }

    // residual_error = residual - H step; returns its inf-norm
    __device__ __forceinline__ double residual_error() {
        Hmul(step, rerr);                          // (H step lands in residual_error itself and is turned into residual - H step in place: no N-vector of scratch)
        double v[1] = {0.0};
        for (int i = tid; i < d.N; i += NT) { const double e = res[i] - rerr[i]; rerr[i] = e; v[0] = fmax(v[0], nabs(e)); }
        block_max(v, red);
        return v[0];
    }

    // ---- filter (filter.jl), thread 0 on the instance's global arrays, result through LDS -------------------------------------------------------
    __device__ __forceinline__ bool check_filter(double theta, double merit) {
        __syncthreads();
        if (tid == 0) {
            const double* ft = filt; const double* fm = filt + mf;
            bool ok = true;
            for (long long i = 0; i < filter_index; ++i) if (!(theta < ft[i] || merit < fm[i])) { ok = false; break; }      // (entries beyond the index are (1e8, 1e8))
            if (ok && filter_index < mf && !(theta < 1.0e8 || merit < 1.0e8)) ok = false;
            red[0] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        return red[0] != 0.0;
    }
    __device__ __forceinline__ void augment_filter(double theta, double merit) {
        const bool ok = filter_index == 0 ? true : check_filter(theta, merit);
        __syncthreads();
        if (tid == 0) {
            double* ft = filt; double* fm = filt + mf; double* ct = filt + 2 * mf; double* cm = filt + 3 * mf;
            long long idx = filter_index;
            if (idx == 0) { ft[0] = theta; fm[0] = merit; idx = 1; }
            else if (ok) {
                const long long nold = idx;
                for (long long i = 0; i < nold; ++i) { ct[i] = ft[i]; cm[i] = fm[i]; }
                idx = 0;
                ft[idx] = theta; fm[idx] = merit; ++idx;
                for (long long i = 0; i < nold; ++i) if (!(ct[i] >= theta && cm[i] >= merit) && idx < mf) { ft[idx] = ct[i]; fm[idx] = cm[i]; ++idx; }
            }
            red[1] = (double)idx;
        }
        __syncthreads();
        filter_index = (long long)red[1];
        __syncthreads();
    }
    __device__ __forceinline__ void filter_reset() { filter_index = 0; }      // (only the first filter_index pairs are ever read)
};

// line_search.jl:2-18 on scalars (the reference's dot(merit_gradient, step.primals) is passed in)
__device__ __forceinline__ bool switching_condition(double step_size, double dd, double merit_exponent, double violation, double violation_exponent, double reg) {
    return dd < 0.0 && step_size * pow(-dd, merit_exponent) > reg * pow(violation, violation_exponent);
}
__device__ __forceinline__ bool sufficient_progress(double v, double vc, double m, double mc, double vt, double mt, double mach) {
    return vc - 10.0 * mach * fabs(v) <= (1.0 - vt) * v || mc - 10.0 * mach * fabs(m) <= m - mt * v;
}
__device__ __forceinline__ bool armijo(double m, double mc, double dd, double step_size, double at, double mach) {
    return mc - m - 10.0 * mach * fabs(m) <= at * step_size * dd;
}

struct StepOut { int exit_kind = 0; int rc = 0; double step_size = 1.0, step_size_t = 1.0, Mh = 0.0, thetah = 0.0, optimality = 0.0; int rounds = 0; int nfact = 0; };

// one pass of the inner loop body of solve! (solve.jl:98-353); equality_violation / cone_product_violation as the caller holds them (:85-86, :332-333)
template <bool SOC> __device__ __forceinline__ StepOut inner_iteration(CtxT<SOC>& c, bool may_converge) {
    const Dm& d = c.d; const Options& o = *c.o; const int tid = c.tid;
    StepOut out;
    double* sol = c.sol; double* cand = c.cand; double* step = c.step; double* res = c.res;
    c.stamp(11);
    // :100-104 gradients, :106-109 barrier + barrier gradient
    c.eval_gradients(sol);
    double s4[4] = {0.0, 0.0, 0.0, 0.0};      // Phi, lambda'r, r'r, -
    for (int i = tid; i < d.q; i += NT) { const double sl = sol[d.os() + i]; s4[0] += log(sl); c.bgrad[i] = 1.0 / sl; }
    for (int j = tid; SOC && j < d.nsoc; j += NT) {                     // second_order.jl:13-14
        const int st = c.soc_start[j], dm = c.soc_dim[j];
        const double* sl = sol + d.os() + st;
        double dd2 = 0.0;
        for (int e = 1; e < dm; ++e) dd2 += sl[e] * sl[e];
        const double det = sl[0] * sl[0] - dd2;
        s4[0] += 0.5 * log(det);
        const double sc = 1.0 / det;
        c.bgrad[st] = sc * sl[0];
        for (int e = 1; e < dm; ++e) c.bgrad[st + e] = sc * (-sl[e]);
    }
    for (int i = tid; i < d.ne; i += NT) { const double r = sol[d.orr() + i]; s4[1] += c.lam[i] * r; s4[2] += r * r; }
    const double* lam = c.lam;
    // :118-124 merit_gradient = [fx; lambda + rho r; -kappa barrier_gradient]: not stored — its only use is the directional derivative below, formed from the parts
    // (the point does not move in between)
    // :127 residual!
    for (int i = tid; i < d.nx; i += NT) res[i] = c.fx[i] + c.gzx[i];
    for (int i = tid; i < d.ne; i += NT) {
        res[d.orr() + i] = lam[i] + c.rho * sol[d.orr() + i] - sol[d.oy() + i];
        res[d.oy() + i] = c.gh[i] - sol[d.orr() + i];
    }
    for (int i = tid; i < d.nc; i += NT) {
        res[d.os() + i] = -sol[d.oz() + i] - sol[d.ot() + i];
        res[d.oz() + i] = c.gh[d.ne + i] - sol[d.os() + i];
        res[d.ot() + i] = c.cprod[i] - c.kappa * c.target(i);
    }
    __syncthreads();
    // :130-135, :170-172 norms
    double n4[4] = {0.0, 0.0, 0.0, 0.0};      // ||res||_1, ||y||_1 + ||z||_1, ||t||_1, theta numerator
    double m4[4] = {0.0, 0.0, 0.0, 0.0};      // ||res[primals]||inf, ||res_y||inf, ||res_z||inf, ||res_t||inf
    for (int i = tid; i < d.N; i += NT) {
        const double a = fabs(res[i]);
        n4[0] += a;
        if (i < d.n) m4[0] = fmax(m4[0], a);
        else if (i < d.oz()) m4[1] = fmax(m4[1], a);
        else if (i < d.ot()) m4[2] = fmax(m4[2], a);
        else m4[3] = fmax(m4[3], a);
        if (i >= d.oy() && i < d.ot()) n4[1] += fabs(sol[i]);
        if (i >= d.ot()) n4[2] += fabs(sol[i]);
        if (i >= d.oy() && i < d.ot()) n4[3] += a;      // res_y = g - r, res_z = h - s: the entries of constraint_violation.jl:1-13
    }
    {   // the merit's three sums, the four norm sums and the four maxima: one reduction
        double sv[7] = {s4[0], s4[1], s4[2], n4[0], n4[1], n4[2], n4[3]};
        block_sum_max(sv, m4, c.red);
        s4[0] = sv[0]; s4[1] = sv[1]; s4[2] = sv[2]; n4[0] = sv[3]; n4[1] = sv[4]; n4[2] = sv[5]; n4[3] = sv[6];
    }
    const double M = c.fcur + (s4[1] + 0.5 * c.rho * s4[2]) - c.kappa * s4[0];                                               // :112-116 merit.jl:2-15
    const double residual_violation = n4[0] / (double)d.N;
    const double sd = (d.ne + d.nc > 0) ? fmax(100.0, n4[1] / (double)(d.ne + d.nc)) / 100.0 : 1.0;      // optimality_error.jl:8
    const double scn = (d.nc > 0) ? fmax(100.0, n4[2] / (double)d.nc) / 100.0 : 1.0;                     // :9
    const double optimality = fmax(fmax(m4[0] / sd, m4[1]), fmax(m4[2], m4[3] / scn));
    const double slack_violation = fmax(m4[1], m4[2]);
    const double theta = (d.ne + d.nc > 0) ? n4[3] / (double)(d.ne + d.nc) : 0.0;
    out.optimality = optimality;
    if (may_converge && residual_violation < o.residual_tolerance && slack_violation < o.slack_tolerance && c.eqv <= o.equality_tolerance &&
        c.cpv <= o.complementarity_tolerance) { out.exit_kind = 1; return out; }                           // :138-143
    if (optimality <= fmax(o.central_path_update_tolerance * c.kappa, o.optimality_tolerance)) { out.exit_kind = 2; return out; }      // :165
    c.stamp(0);
    // :175-185: the Hessian and the Jacobians of a QP are constant; the cone Jacobians are functions of (s, t) formed where they are used
    // ---- :187 search_direction!: inertia_correction! (inertia.jl:30-80, quirk B-1: IC-3 always takes max(min_regularization, scaling_regularization_last * eps_last))
    {   // (one loop, ONE instance of the factorisation's code: IC-1, then IC-4 as often as the inertia test fails)
        int zero = 0, count = 0;
        c.ep = o.primal_regularization_initial; c.ed = o.dual_regularization_initial;
        for (;;) {
            const bool ok = c.factorize(zero); ++count;                                                      // IC-1 / IC-4
            if (ok) { if (count > 1) c.ep_last = c.ep; break; }
            if (count == 1) {
                if (zero != 0) c.ed = o.dual_regularization * pow(c.kappa, o.dual_regularization_exponent);  // IC-2
                c.ep = fmax(o.min_regularization, o.scaling_regularization_last * c.ep_last);               // IC-3
            } else {
                if (c.ep_last == 0.0) c.ep = o.scaling_regularization_initial * c.ep;                        // IC-5
                else c.ep = o.scaling_regularization * c.ep;
                if (c.ep > o.max_regularization) { out.rc = CALIPSO_ERR_INERTIA; out.nfact = count; return out; }      // IC-6
            }
        }
        out.nfact = count;
    }
    c.stamp(1);
    {   // search_direction_symmetric!(step, residual), then iterative_refinement! (iterative_refinement.jl:1-52) — one loop, ONE instance of the solve's and the residual's code:
        // the first pass is the solve for the step itself, every further pass a correction round
        int it = 0;
        bool first = true, good = false;
        double norm = 0.0, norm0 = 0.0;
        for (;;) {
            c.search_direction_symmetric(first ? res : c.rerr, first ? step : c.corr);
            if (first) c.stamp(4);
            if (!o.iterative_refinement) { good = true; break; }
            if (!first) { for (int i = tid; i < d.N; i += NT) step[i] += c.corr[i]; __syncthreads(); it += 1; }
            norm = c.residual_error();
            if (first) { norm0 = norm; first = false; }
            if (it > o.max_iterative_refinement) break;                                                      // `while iteration <= max_iterative_refinement`
            if (norm <= o.iterative_refinement_tolerance && it >= o.min_iterative_refinement) { good = true; break; }
        }
        out.rounds = it;
        if (o.iterative_refinement) { c.rlast = it; if (it > c.rmax) c.rmax = it; }
        if (!good && !(norm <= norm0)) { c.rfail += 1; out.rc = CALIPSO_WARN_REFINEMENT; return out; }      // (the reference would take H \ residual: left to the general path)
    }
    c.stamp(5);
    // ---- :190-221 cone search: separate step sizes for s and t -----------------------------------------------------------------------------------------
    double a_s = 1.0, a_t = 1.0;
    if (d.nc > 0) {
        const double omt = 1.0 - c.tau;
        for (int which = 0; which < 2; ++which) {
            const int off = which == 0 ? d.os() : d.ot();
            double a = 1.0;
            int it = 0;
            for (;;) {
                double v[1] = {0.0};
                for (int i = tid; i < d.q; i += NT) if (sol[off + i] - a * step[off + i] <= omt * sol[off + i]) v[0] = 1.0;      // nonnegative.jl:29-34
                for (int j = tid; SOC && j < d.nsoc; j += NT) {                                                                              // second_order.jl:45-47
                    const int st = c.soc_start[j], dm = c.soc_dim[j];
                    const double* x = sol + off + st; const double* dx = step + off + st;
                    double nrm = 0.0;
                    for (int e = 1; e < dm; ++e) { const double df = (x[e] - a * dx[e]) - omt * x[e]; nrm += df * df; }
                    if ((x[0] - a * dx[0]) - omt * x[0] <= sqrt(nrm)) v[0] = 1.0;
                }
                block_max(v, c.red);
                if (v[0] == 0.0) break;
                a = o.scaling_line_search * a;
                it += 1;
                if (it > o.max_cone_line_search) { out.rc = CALIPSO_ERR_CONE_SEARCH; return out; }            // solve.jl:210,220
            }
            if (which == 0) a_s = a; else a_t = a;
        }
    }
    out.step_size_t = a_t;
    double step_size = a_s;
    // candidate (:206-218, :224-229) and the directional derivative of the merit function
    double dd1[1] = {0.0};
    for (int i = tid; i < d.nx; i += NT) dd1[0] += c.fx[i] * step[i];
    for (int i = tid; i < d.ne; i += NT) dd1[0] += (lam[i] + c.rho * sol[d.orr() + i]) * step[d.orr() + i];
    for (int i = tid; i < d.nc; i += NT) dd1[0] += (-1.0 * c.kappa * c.bgrad[i]) * step[d.os() + i];
    block_sum(dd1, c.red);
    const double dd = dd1[0];
    for (int i = tid; i < d.n; i += NT) cand[i] = sol[i] - step_size * step[i];
    for (int i = tid; i < d.nc; i += NT) cand[d.ot() + i] = sol[d.ot() + i] - a_t * step[d.ot() + i];
    __syncthreads();
    auto candidate_merit = [&](double& Mh, double& thetah) {                                                // :231-250 / :278-297: evaluate!(objective, equality, cone), cone!(barrier), merit, violation
        mv_n(c.Lxx, d.lds, d.nx, d.nx, cand, c.xb, nullptr);
        mv_n(c.Z, d.ldz, d.m, d.nx, cand, c.ghc, c.bh);
        __syncthreads();
        double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // Phi, lambda'r, r'r, theta numerator, x'Lxx x, q'x
        for (int i = tid; i < d.nx; i += NT) { v[4] += cand[i] * c.xb[i]; v[5] += c.q[i] * cand[i]; }
        for (int i = tid; i < d.nc; i += NT) { const double sl = cand[d.os() + i]; if (i < d.q) v[0] += log(sl); v[3] += fabs(c.ghc[d.ne + i] - sl); }
        for (int j = tid; SOC && j < d.nsoc; j += NT) {
            const int st = c.soc_start[j], dm = c.soc_dim[j];
            const double* sl = cand + d.os() + st;
            double dd2 = 0.0;
            for (int e = 1; e < dm; ++e) dd2 += sl[e] * sl[e];
            v[0] += 0.5 * log(sl[0] * sl[0] - dd2);
        }
        for (int i = tid; i < d.ne; i += NT) { const double r = cand[d.orr() + i]; v[1] += lam[i] * r; v[2] += r * r; v[3] += fabs(c.ghc[i] - r); }
        block_sum(v, c.red);
        c.fcand = 0.5 * v[4] + v[5];
        Mh = c.fcand + (v[1] + 0.5 * c.rho * v[2]) - c.kappa * v[0];
        thetah = (d.ne + d.nc > 0) ? v[3] / (double)(d.ne + d.nc) : 0.0;
    };
    c.stamp(6);
    double Mh, thetah;
    int residual_iteration = 0;
    for (;;) {                                                                                              // :231-250, then :254-302
        candidate_merit(Mh, thetah);
        if (!(residual_iteration < o.max_residual_line_search)) break;
        if (c.check_filter(thetah, Mh)) {
            if (theta <= o.slack_tolerance && switching_condition(step_size, dd, o.merit_exponent, theta, o.violation_exponent, 1.0) &&
                armijo(M, Mh, dd, step_size, o.armijo_tolerance, o.machine_tolerance)) break;
            else if (sufficient_progress(theta, thetah, M, Mh, o.violation_tolerance, o.merit_tolerance, o.machine_tolerance)) break;
        }
        step_size = o.scaling_line_search * step_size;
        for (int i = tid; i < d.n; i += NT) cand[i] = sol[i] - step_size * step[i];                          // :268-276 (x, r, s; t keeps its own step size)
        __syncthreads();
        residual_iteration += 1;
    }
    if (residual_iteration >= o.max_residual_line_search) out.rc = CALIPSO_WARN_LINE_SEARCH;
    if (!switching_condition(step_size, dd, o.merit_exponent, theta, o.violation_exponent, 1.0) || !armijo(M, Mh, dd, step_size, o.armijo_tolerance, o.machine_tolerance))
        c.augment_filter((1.0 - o.violation_tolerance) * theta, M - o.merit_tolerance * theta);              // filter.jl:81-89
    c.stamp(7);
    // :309-326 accept
    for (int i = tid; i < d.n; i += NT) sol[i] = cand[i];
    for (int i = tid; i < d.m; i += NT) sol[d.oy() + i] = sol[d.oy() + i] - step_size * step[d.oy() + i];
    for (int i = tid; i < d.nc; i += NT) sol[d.ot() + i] = cand[d.ot() + i];
    for (int i = tid; i < d.m; i += NT) c.gh[i] = c.ghc[i];
    c.fcur = c.fcand;
    __syncthreads();
    c.cone_product(sol);                                                                                    // :328-330
    double v2[2] = {0.0, 0.0};
    for (int i = tid; i < d.ne; i += NT) v2[0] = fmax(v2[0], fabs(c.gh[i]));                                 // :332
    for (int i = tid; i < d.nc; i += NT) v2[1] = fmax(v2[1], fabs(c.cprod[i]));                              // :333
    block_max(v2, c.red);
    c.eqv = v2[0]; c.cpv = v2[1];
    c.nsteps += 1;
    c.stamp(8);
    out.step_size = step_size; out.Mh = Mh; out.thetah = thetah;
    return out;
}

template <bool SOC> __global__ __launch_bounds__(NT, 2) void k_smallnewton(Args a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (inst >= a.batch) return;
    const Dm d = a.d;
    const Lay L = layout(d);
    CtxT<SOC> c;
    c.d = d; c.o = &a.o; c.tid = tid;
    c.Lxx = sm + L.Lxx; c.Z = sm + L.Z; c.S = sm + L.S; c.q = sm + L.q; c.bh = sm + L.bh; c.lam = sm + L.lam; c.sol = sm + L.sol; c.cand = sm + L.cand; c.step = sm + L.step;
    c.res = sm + L.res; c.rerr = sm + L.rerr; c.corr = sm + L.corr; c.rsym = sm + L.rsym;
    c.fx = sm + L.fx; c.gzx = sm + L.gzx; c.gh = sm + L.gh; c.ghc = sm + L.ghc; c.cprod = sm + L.cprod; c.bgrad = sm + L.bgrad; c.wz = sm + L.wz; c.wsoc = sm + L.wsoc; c.bsoc = sm + L.bsoc; c.vsoc = sm + L.vsoc;
    c.soc_start = a.soc_start; c.soc_dim = a.soc_dim; c.soc_woff = a.soc_woff;
    c.D = sm + L.D; c.Dinv = sm + L.Dinv; c.xb = sm + L.xb; c.t1 = sm + L.t1; c.t2 = sm + L.t2; c.ycol = sm + L.ycol; c.red = sm + L.red;
    c.mf = (int)a.o.max_filter;
    const long long mf_ = c.mf;
    c.filt = a.filt + (size_t)inst * 6 * (size_t)c.mf;
    const Options& o = a.o;
    // ---- problem and state into LDS ---------------------------------------------------------------------------------------------------------------
    {
        const double* P = a.P + (size_t)inst * a.sP; const double* q = a.q + (size_t)inst * a.sq;
        const double* Zg = a.Z + (size_t)inst * a.sZ; const double* bh = a.bh + (size_t)inst * a.sbh;
        for (int e = tid; e < d.nx * d.nx; e += NT) c.Lxx[(e % d.nx) + (e / d.nx) * d.lds] = P[e];
        for (int e = tid; e < d.m * d.nx; e += NT) c.Z[(e % d.m) + (e / d.m) * d.ldz] = Zg[e];
        for (int i = tid; i < d.nx; i += NT) c.q[i] = q[i];
        for (int i = tid; i < d.m; i += NT) c.bh[i] = bh[i];
        const double* w = a.w + (size_t)inst * d.N;
        for (int i = tid; i < d.N; i += NT) c.sol[i] = w[i];
    }
    double* lam = c.lam;
    double* gsc = a.sc + (size_t)inst * SC_COUNT;
    long long* cnt = a.cnt + (size_t)inst * CN_COUNT;
    c.kappa = gsc[SC_KAPPA]; c.tau = gsc[SC_TAU]; c.rho = gsc[SC_RHO]; c.ep = gsc[SC_EP]; c.ep_last = gsc[SC_EPLAST]; c.ed = gsc[SC_ED];
    c.eqv = gsc[SC_EQV]; c.cpv = gsc[SC_CPV]; c.fcur = gsc[SC_F]; c.fcand = 0.0; c.omega_y = 0.0; c.kyy = 0.0;
    c.filter_index = cnt[CN_FILTER]; c.nfact_total = cnt[CN_FACT]; c.rfail = cnt[CN_RFAIL]; c.rmax = cnt[CN_RMAX]; c.rlast = cnt[CN_RLAST]; c.nsteps = cnt[CN_STEPS];
    long long total_iterations = cnt[CN_TOTAL], outer = cnt[CN_OUTER], trace_row = cnt[CN_TRACE];
    __syncthreads();
#ifdef SN_TRACE
    for (int k = 0; k < 12; ++k) c.tph[k] = 0;
    c.tlast = wall_clock64();
#endif
    int status = 0;
    StepOut last;
    auto load_lambda = [&] { const double* lg = a.lam + (size_t)inst * (d.ne > 0 ? d.ne : 1); for (int i = tid; i < d.ne; i += NT) lam[i] = lg[i]; __syncthreads(); };
    auto record_trace = [&] {
        if (a.trace && trace_row < a.trace_rows) { double* tr = a.trace + ((size_t)inst * a.trace_rows + (size_t)trace_row) * d.N; for (int i = tid; i < d.N; i += NT) tr[i] = c.sol[i]; }
        trace_row += 1;
    };
    const bool solving = a.mode == MODE_SOLVE;
    if (solving) {
        // ---- solve!(solver)  solve.jl:8-96: initialisation ------------------------------------------------------------------------------------------
        c.nfact_total = 0; c.rfail = 0; c.rmax = 0; c.rlast = 0; c.nsteps = 0; trace_row = 0;
        if (o.warmstart == 0.0) {                                                                            // initialize_slacks! / initialize_duals!  initialize.jl:15-36
            c.eval_constraints(c.sol, c.gh);
            for (int i = tid; i < d.ne; i += NT) { c.sol[d.orr() + i] = c.gh[i]; c.sol[d.oy() + i] = 0.0; }
            for (int i = tid; i < d.nc; i += NT) { const double v0 = c.target(i) != 0.0 ? 1.0 : 0.1; c.sol[d.os() + i] = v0; c.sol[d.oz() + i] = 0.0; c.sol[d.ot() + i] = v0; }      // nonnegative.jl:2-8, second_order.jl:2-10
            __syncthreads();
        }
        c.kappa = o.central_path_initial; c.tau = fmax(0.99, 1.0 - c.kappa);                                 // initialize.jl:38-42
        c.rho = o.penalty_initial;                                                                           // :44-48
        for (int i = tid; i < d.ne; i += NT) lam[i] = o.dual_initial;
        __syncthreads();
        total_iterations = 1;
        c.filter_reset();                                                                                    // :95
    } else load_lambda();
    // :78-83 (solve!) / the values at the resident point (steps: a resident state does not carry them)
    c.fcur = c.eval_objective(c.sol);
    c.eval_constraints(c.sol, c.gh);
    if (solving) {
        double v[1] = {0.0};
        for (int i = tid; i < d.ne; i += NT) v[0] = fmax(v[0], fabs(c.gh[i]));
        block_max(v, c.red);
        c.eqv = v[0];                                                                                        // :85
        c.cpv = 0.0;                                                                                         // :86 reads cone_product BEFORE cone!(product): zeros on a fresh solver (quirk B-6)
    }
    c.cone_product(c.sol);                                                                                   // :88-91 (the target of a nonnegative cone is 1)
    // ---- the loops of solve.jl:97-372 (solving) or `count` passes of the inner loop body from the resident state (calipso_hip_newton_steps: never "converged",
    // exit kind 2 leaves the point as it is) — ONE loop, one instance of the iteration's code
    long long jo = 1, ii = 1;
    int kdone = 0;
    if (solving) outer = 1;
    for (;;) {
        if (solving ? jo > o.max_outer_iterations : kdone >= a.count) break;
        const double kap = c.kappa, tau = c.tau, rho = c.rho, epl = c.ep_last, fc = c.fcur;
        const long long fidx = c.filter_index;
        const bool restore = !solving && !a.advance;
        if (restore && tid == 0) for (long long i = 0; i < fidx; ++i) { c.filt[4 * mf_ + i] = c.filt[i]; c.filt[5 * mf_ + i] = c.filt[mf_ + i]; }
        last = inner_iteration(c, solving);
        if (last.rc < 0 || last.rc == CALIPSO_WARN_REFINEMENT) { status = last.rc < 0 ? last.rc : -100 - last.rc; break; }
        if (solving) {
            if (last.exit_kind == 1) { status = 1; break; }                                                  // :138-160
            bool inner_done = last.exit_kind == 2;                                                           // :165
            if (!inner_done) { total_iterations += 1; record_trace(); ii += 1; if (ii > o.max_residual_iterations) inner_done = true; }
            if (inner_done) {
                c.kappa = fmax(o.residual_tolerance / 10.0, fmin(o.central_path_scaling * c.kappa, pow(c.kappa, o.central_path_exponent)));      // :356
                c.tau = fmax(0.99, 1.0 - c.kappa);                                                           // :359
                for (int i = tid; i < d.ne; i += NT) lam[i] = lam[i] + c.rho * c.sol[d.orr() + i];           // :362-364
                __syncthreads();
                c.rho = fmin(fmax(o.penalty_scaling * c.rho, 1.0 / c.kappa), o.max_penalty);                 // :365
                c.filter_reset();                                                                            // :368
                jo += 1; ii = 1;
                if (jo <= o.max_outer_iterations) outer = jo;
            }
        } else {
            if (last.exit_kind == 0) { total_iterations += 1; record_trace(); }
            if (restore) {
                // benchmark mode: the point from the instance's global copy (untouched until the write-back), scalars from registers, the filter's pairs from their
                // saved copy (entries beyond the index are never read)
                const double* w = a.w + (size_t)inst * d.N;
                for (int i = tid; i < d.N; i += NT) c.sol[i] = w[i];
                if (tid == 0) for (long long i = 0; i < fidx; ++i) { c.filt[i] = c.filt[4 * mf_ + i]; c.filt[mf_ + i] = c.filt[5 * mf_ + i]; }
                __syncthreads();
                c.kappa = kap; c.tau = tau; c.rho = rho; c.ep_last = epl; c.fcur = fc; c.filter_index = fidx;
                c.eval_constraints(c.sol, c.gh);
                c.cone_product(c.sol);
            }
            kdone += 1;
        }
    }
    // ---- write the state back -----------------------------------------------------------------------------------------------------------------------
    __syncthreads();
    if (a.mode == MODE_SOLVE || a.advance) {
        double* w = a.w + (size_t)inst * d.N;
        for (int i = tid; i < d.N; i += NT) w[i] = c.sol[i];
        double* lg = a.lam + (size_t)inst * (d.ne > 0 ? d.ne : 1);
        for (int i = tid; i < d.ne; i += NT) lg[i] = lam[i];
    }
    if (tid == 0) {
        if (a.mode == MODE_SOLVE || a.advance) {
            gsc[SC_KAPPA] = c.kappa; gsc[SC_TAU] = c.tau; gsc[SC_RHO] = c.rho; gsc[SC_EPLAST] = c.ep_last; gsc[SC_EQV] = c.eqv; gsc[SC_CPV] = c.cpv; gsc[SC_F] = c.fcur;
            cnt[CN_FILTER] = c.filter_index;
        }
        gsc[SC_EP] = c.ep; gsc[SC_ED] = c.ed;
        cnt[CN_TOTAL] = total_iterations; cnt[CN_OUTER] = outer; cnt[CN_FACT] = c.nfact_total; cnt[CN_RFAIL] = c.rfail; cnt[CN_RMAX] = c.rmax; cnt[CN_RLAST] = c.rlast;
        cnt[CN_STEPS] = c.nsteps; cnt[CN_TRACE] = trace_row;
        a.status[inst] = status;
#ifdef SN_TRACE
        if (a.prof && inst == 0) for (int k = 0; k < 12; ++k) a.prof[k] = (double)c.tph[k] * 0.01;      // microseconds (100 MHz)
#endif
        double* inf = a.info + (size_t)inst * IN_COUNT;
        inf[IN_STEP] = last.step_size; inf[IN_STEP_T] = last.step_size_t; inf[IN_ROUNDS] = last.rounds; inf[IN_NFACT] = last.nfact; inf[IN_MH] = last.Mh; inf[IN_THETAH] = last.thetah;
        inf[IN_EXIT] = last.exit_kind; inf[IN_OPT] = last.optimality;
    }
}

int fail(SN* s, int code, const std::string& msg) { s->err = msg; return code; }
thread_local std::string g_sn_err;

#define SK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(s, CALIPSO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)

Dm dims_of(const SN* s) {
    Dm d; d.nx = s->nx; d.ne = s->ne; d.nc = s->nc; d.m = s->ne + s->nc; d.n = s->nx + d.m; d.N = s->nx + 2 * s->ne + 3 * s->nc;
    d.ldz = (d.m > 0 ? d.m : 1) | 1; d.lds = s->nx | 1;
    d.q = s->nq; d.nsoc = (int)s->soc_dim.size(); d.wsz = s->wsz; d.maxd = s->maxd;
    return d;
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
    static_assert(sizeof(Args) <= 3800, "kernel arguments");
    SK(hipEventRecord(s->ev0, s->stream));
    if (s->soc_dim.empty()) hipLaunchKernelGGL(k_smallnewton<false>, dim3((unsigned)s->batch), dim3(NT), s->lds_bytes, s->stream, a);
    else hipLaunchKernelGGL(k_smallnewton<true>, dim3((unsigned)s->batch), dim3(NT), s->lds_bytes, s->stream, a);
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
    if (s->lds_bytes > 64 * 1024) { (void)calipso::lds_attribute((const void*)k_smallnewton<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)k_smallnewton<true>, 160 * 1024); }
    SK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    SK(hipEventCreate(&s->ev0)); SK(hipEventCreate(&s->ev1));
    const size_t B = (size_t)batch, N = (size_t)d.N;
    auto alloc = [&](double** p, size_t n) { if (hipMalloc((void**)p, sizeof(double) * std::max<size_t>(n, 1)) != hipSuccess) return false; return hipMemsetAsync(*p, 0, sizeof(double) * std::max<size_t>(n, 1), s->stream) == hipSuccess; };
    if (!alloc(&s->w, B * N) || !alloc(&s->lam, B * std::max(1, d.ne)) || !alloc(&s->sc, B * SC_COUNT) || !alloc(&s->filt, B * 6 * (size_t)s->opt.max_filter) || !alloc(&s->info, B * IN_COUNT) || !alloc(&s->prof, 16))
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
    for (double* p : {s->P, s->q, s->Z, s->bh, s->w, s->lam, s->sc, s->filt, s->info, s->trace, s->prof}) if (p) (void)hipFree(p);
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
    if (s->lds_bytes > 64 * 1024) { (void)calipso::lds_attribute((const void*)k_smallnewton<false>, 160 * 1024); (void)calipso::lds_attribute((const void*)k_smallnewton<true>, 160 * 1024); }
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

// phase clocks of instance 0 in the last launch, microseconds (a build with -DSN_TRACE; zeros otherwise): [0] evaluation + residual + norms, [1] inertia logic, [2] cone
// weights + assembly of S, [3] LDL^T, [4] first condensed solve, [5] refinement, [6] cone search + candidate, [7] candidate merit + line search, [8] accept, [9] of the LDL^T: the panels (one wavefront), [3] then holds its trailing updates, [11] between steps
int32_t calipso_hip_debug_smallnewton_profile(calipso_hip_smallnewton* s, double out[12]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    SK(hipMemcpy(out, s->prof, sizeof(double) * 12, hipMemcpyDeviceToHost));
    return CALIPSO_OK;
}

}  // extern "C"
