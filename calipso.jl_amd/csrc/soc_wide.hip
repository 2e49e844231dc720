// soc_wide.hip — second-order cones of dimension > 4: ONE WAVEFRONT PER CONE, lane l holds elements l, l + 64, ... of every cone vector (E elements per
// lane: dimension <= 64 E; E = 1 for cones up to 64, 2 / 4 / 8 / 16 for the wider ones up to 1024 (MAX_SOC_DIM; at E = 16 the cone's vectors no longer fit the register
// file and the compiler keeps part of them in scratch: slower, same arithmetic) — the reference has no limit, cones/second_order.jl:1-69), so
// nothing lives in private arrays indexed at run time.
//   cones/second_order.jl:50-65            second_order_vector_inverse (the closed-form arrow inverse)       -> arrow_inverse_wave
//   residual_jacobian_variables.jl:151-164  K_zz block of a cone, column by column                            -> k_cone_weights_wide
//   residual.jl:78-99                       condensed right-hand side of the cone rows                        -> k_residual_symmetric_wide
//   search_direction.jl:83-101              recovery of (ds, dt) of a cone                                    -> k_recover_wide
//   iterative_refinement.jl:9,39 (cone rows of residual - H step) + the next condensed right-hand side       -> k_refine_local_wide
// The kernels of schur.hip / vectors.hip keep the register path for cones of dimension <= 4 (friction cones, SOC2 / SOC3: BASELINE's sizes)
// and skip the wide ones; these kernels run right behind them, only on handles that HAVE wide cones (portfolio: dimension 12), one 64-thread
// workgroup per wide cone.  Sums run over the elements in index order with the operand broadcast by v_readlane, i.e. in the reference's own
// operation order (the reductions are NOT tree-shaped): what a lane computes is what the sequential loop computed for that index.  The d x d block of a cone
// (its triu-symmetrised LDL^T) sits in LDS up to dimension 128 and in the handle's cone scratch beyond; its factorisation is O(d^3) on one wavefront — a wide
// cone is a slow path (as it is in the reference: dense d x d inverses per column), not an error.
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

__device__ __forceinline__ double bc(double v, int lane) {     // element `lane` of a cone vector (lane: wave-uniform)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// element i (wave-uniform, any slot) of a cone vector held E elements per lane
template <int E> __device__ __forceinline__ double bce(const double (&v)[E], int i) {
    double r = bc(v[0], i & 63);
#pragma unroll
    for (int e = 1; e < E; ++e) { const double c = bc(v[e], i & 63); r = (i >> 6) == e ? c : r; }
    return r;
}
// f(e, l) for the elements 64 e + l = i0 .. n - 1 in ascending order (i0 = 0 or 1); e is a compile-time constant inside f after unrolling
template <int E, typename F> __device__ __forceinline__ void for_elems(int i0, int n, F f) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int hi = n - 64 * e < 64 ? n - 64 * e : 64;
        for (int l = e == 0 ? i0 : 0; l < hi; ++l) f(e, l);
    }
}
template <int E, typename F> __device__ __forceinline__ void for_elems_down(int n, int ilast, F f) {     // elements n - 1 down to ilast
#pragma unroll
    for (int e = E - 1; e >= 0; --e) {
        const int hi = n - 64 * e < 64 ? n - 64 * e : 64;
        for (int l = hi - 1; l >= (e == 0 ? ilast : 0); --l) f(e, l);
    }
}

// out = arrow(u)^-1 x, lane l holds u[l + 64 e], x[l + 64 e] and receives out[l + 64 e] (device_utils.hpp: arrow_inverse, same operations in the same order)
template <int E>
__device__ __forceinline__ void arrow_inverse_wave(int n, int lane, const double (&u)[E], const double (&x)[E], double (&out)[E]) {
    const double u0 = bc(u[0], 0), x0 = bc(x[0], 0);
    double uu = 0.0;
    for_elems<E>(1, n, [&](int e, int l) { const double ui = bc(u[e], l); uu += ui * ui; });
    const double alpha = -1.0 / (u0 * u0) * uu;
    const double beta = 1.0 / (1.0 + alpha);
    double d0 = 0.0;
    for_elems<E>(1, n, [&](int e, int l) { d0 += (bc(u[e], l) / u0) * bc(x[e], l); });
    const double x0_1 = x0 - d0;
    double v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = x[e] - beta * ((u[e] / u0) * x0_1);
    double d1 = 0.0;
    for_elems<E>(1, n, [&](int e, int l) { d1 += (bc(u[e], l) / u0) * bc(v[e], l); });
    const double x2_1 = x0 - d1;
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = (lane + 64 * e == 0) ? 1.0 / u0 * x2_1 : 1.0 / u0 * v[e];
}

// ---- K_zz block, its triu-symmetrised LDL^T (pivot signs -> inertia) and W = -(B_sym)^-1 ----------------------------------------------------
// M: dim x (dim + 1) doubles — dynamic LDS (dimension <= 128) or the cone scratch of the handle (Mg != nullptr)
template <int E>
__global__ __launch_bounds__(64) void k_cone_weights_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, double* __restrict__ Bsoc,
                                                           double* __restrict__ Wsoc, int* __restrict__ icount, double* __restrict__ Mg) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    inst_shift(bt.b, w, Bsoc, Wsoc);
    if (Mg) inst_shift(bt.b, Mg);
    inst_shift_i(bt.b, icount);
    const Scalars sc = bt.scal(blockIdx.z);
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j], off = cd.soc_woff[j];
    const int LD = dim + 1;
    double* ycol = lds;                                   // [64 E]
    double* M = Mg ? Mg + 2 * (size_t)off : lds + 64 * E;  // the symmetrised block, then its L (below) and D (diagonal); (the scratch holds 2 d^2 doubles per cone)
    const int lane = threadIdx.x;
    double* B = Bsoc + off;
    double* W = Wsoc + off;
    const double Hss = 0.0 + sc.ep;
    double sl[E], t[E], u[E];
    bool in[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; in[e] = a < dim; sl[e] = in[e] ? w[d.os() + st + a] : 0.0; t[e] = in[e] ? w[d.ot() + st + a] : 0.0; }
    const double sb1 = bc(sl[0], 0) - sc.ed;
#pragma unroll
    for (int e = 0; e < E; ++e) u[e] = (lane + 64 * e == 0) ? t[e] + sb1 * Hss : t[e] + sl[e] * Hss;
    // B = -(Cs + Cbar_t P)^-1 Cbar_t + D, column by column; only triu(B) enters the factorisation (linear_solver.jl:23)
    for (int col = 0; col < dim; ++col) {
        const double slc = bce<E>(sl, col);
        double c[E], o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; c[e] = (a == col) ? sb1 : (col == 0 ? sl[e] : (a == 0 ? slc : 0.0)); }
        arrow_inverse_wave<E>(dim, lane, u, c, o);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int a = lane + 64 * e;
            double bv = 0.0 - o[e];
            if (a == col) bv += (0.0 - sc.ed);
            if (in[e]) {
                B[a + col * dim] = bv;
                if (a <= col) { M[a * LD + col] = bv; M[col * LD + a] = bv; }
            }
        }
    }
    __syncthreads();
    int pos = 0, nonpos = 0, zero = 0;
    for (int jj = 0; jj < dim; ++jj) {
        const double dj = M[jj * LD + jj];
        pos += dj > 0.0; nonpos += dj <= 0.0; zero += dj == 0.0;
        double yij[E];
#pragma unroll
        for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; yij[e] = (in[e] && a > jj) ? M[a * LD + jj] : 0.0; if (in[e] && a > jj) ycol[a] = yij[e]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int a = lane + 64 * e;
            if (in[e] && a > jj) {
                const double l = yij[e] / dj;
                for (int k = jj + 1; k <= a; ++k) M[a * LD + k] -= l * (k == a ? yij[e] : ycol[k]);    // y_k = unscaled column entry
                M[a * LD + jj] = l;
            }
        }
        __syncthreads();
    }
    if (lane == 0) { atomicAdd(&icount[0], pos); atomicAdd(&icount[1], nonpos); atomicAdd(&icount[2], zero); }
    // W = -(B_sym)^-1: L D L' x = e_col, forward / scale / backward with the finished entries broadcast
    double dinv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; dinv[e] = in[e] ? M[a * LD + a] : 1.0; }
    for (int col = 0; col < dim; ++col) {
        double o[E];
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = (lane + 64 * e == col) ? 1.0 : 0.0;
        for_elems<E>(0, dim - 1, [&](int ek, int lk) {
            const int k = 64 * ek + lk;
            const double ok = bc(o[ek], lk);
#pragma unroll
            for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; if (in[e] && a > k) o[e] -= M[a * LD + k] * ok; }
        });
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] /= dinv[e];
        for_elems_down<E>(dim, 1, [&](int ek, int lk) {
            const int k = 64 * ek + lk;
            const double ok = bc(o[ek], lk);
#pragma unroll
            for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; if (a < k) o[e] -= M[k * LD + a] * ok; }
        });
#pragma unroll
        for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; if (in[e]) W[a + col * dim] = -o[e]; }
    }
}

// t1 rows of a cone: s_a = sum_b W[a + b dim] o_b in index order
template <int E>
__device__ __forceinline__ void w_times(int dim, int lane, const bool (&in)[E], const double* __restrict__ W, const double (&o)[E], double (&s)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.0;
    for_elems<E>(0, dim, [&](int eb, int lb) {
        const int b = 64 * eb + lb;
        const double ob = bc(o[eb], lb);
#pragma unroll
        for (int e = 0; e < E; ++e) { const int a = lane + 64 * e; if (in[e]) s[e] += W[a + b * dim] * ob; }
    });
}

// ---- condensed right-hand side of the rows of one cone: b_z = r_z + U^-1 (Cbar_t r_s + r_t), t1 = W b_z -------------------------------------
template <int E>
__global__ __launch_bounds__(64) void k_residual_symmetric_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ res_,
                                                                 const double* __restrict__ Wsoc, double* __restrict__ rsym_, double* __restrict__ t1_) {
    inst_shift(bt.b, w, res_, Wsoc, rsym_, t1_);
    const Scalars sc = bt.scal(blockIdx.z);
    const double* res = res_ + (size_t)blockIdx.y * d.N;
    double* rsym = rsym_ + (size_t)blockIdx.y * d.n;
    double* t1 = t1_ + (size_t)blockIdx.y * d.m;
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j];
    const int lane = threadIdx.x;
    const double Hss = 0.0 + sc.ep;
    double sl[E], t[E], rs[E], rt[E], rz[E], u[E], v[E], o[E], s[E];
    bool in[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e; in[e] = a < dim;
        sl[e] = in[e] ? w[d.os() + st + a] : 0.0; t[e] = in[e] ? w[d.ot() + st + a] : 0.0;
        rs[e] = in[e] ? res[d.os() + st + a] : 0.0; rt[e] = in[e] ? res[d.ot() + st + a] : 0.0; rz[e] = in[e] ? res[d.oz() + st + a] : 0.0;
    }
    const double sb1 = bc(sl[0], 0) - sc.ed, rs0 = bc(rs[0], 0);
    double acc = sb1 * rs0;
    for_elems<E>(1, dim, [&](int e, int l) { acc += bc(sl[e], l) * bc(rs[e], l); });
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e;
        u[e] = a == 0 ? t[e] + sb1 * Hss : t[e] + sl[e] * Hss;
        v[e] = a == 0 ? acc + rt[e] : (sl[e] * rs0 + sb1 * rs[e]) + rt[e];
    }
    arrow_inverse_wave<E>(dim, lane, u, v, o);
#pragma unroll
    for (int e = 0; e < E; ++e) { o[e] = rz[e] + o[e]; if (in[e]) rsym[d.nx + d.ne + st + lane + 64 * e] = o[e]; }
    w_times<E>(dim, lane, in, Wsoc + cd.soc_woff[j], o, s);
#pragma unroll
    for (int e = 0; e < E; ++e) if (in[e]) t1[d.ne + st + lane + 64 * e] = s[e];
}

// ---- dz back-substitution + (ds, dt) recovery of one cone (k_recover's second-order branch) --------------------------------------------------
template <int E>
__global__ __launch_bounds__(64) void k_recover_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ res_,
                                                      const double* __restrict__ b_, const double* __restrict__ t2_, const double* __restrict__ Wsoc,
                                                      double* __restrict__ dsym_, double* __restrict__ step_, double* __restrict__ accum, double* __restrict__ zsx, int zsx_mode) {
    inst_shift(bt.b, w, res_, b_, t2_, Wsoc, dsym_, step_);
    if (accum) inst_shift(bt.b, accum);
    if (zsx_mode) inst_shift(bt.b, zsx);
    const Scalars sc = bt.scal(blockIdx.z);
    const double* res = res_ + (size_t)blockIdx.y * d.N;
    const double* b = b_ + (size_t)blockIdx.y * d.n;
    const double* t2 = t2_ + (size_t)blockIdx.y * d.m;
    double* dsym = dsym_ + (size_t)blockIdx.y * d.n;
    double* step = step_ + (size_t)blockIdx.y * d.N;
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j];
    const int lane = threadIdx.x;
    const double Hss = 0.0 + sc.ep;
    double sl[E], t[E], rs[E], rt[E], o0[E], s[E], dz[E], u[E], v[E], ds[E], dt[E];
    bool in[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e; in[e] = a < dim;
        sl[e] = in[e] ? w[d.os() + st + a] : 0.0; t[e] = in[e] ? w[d.ot() + st + a] : 0.0;
        rs[e] = in[e] ? res[d.os() + st + a] : 0.0; rt[e] = in[e] ? res[d.ot() + st + a] : 0.0;
        const double tt = in[e] ? t2[d.ne + st + a] : 0.0;
        o0[e] = in[e] ? b[d.nx + d.ne + st + a] - tt : 0.0;
        if (zsx_mode && in[e]) zsx[d.ne + st + a] = zsx_mode == 1 ? tt : zsx[d.ne + st + a] + tt;
    }
    w_times<E>(dim, lane, in, Wsoc + cd.soc_woff[j], o0, s);
#pragma unroll
    for (int e = 0; e < E; ++e) { dz[e] = -1.0 * s[e]; if (in[e]) dsym[d.nx + d.ne + st + lane + 64 * e] = dz[e]; }
    const double sb1 = bc(sl[0], 0) - sc.ed;
    // ds = U^-1 (r_t + Cbar_t (r_s + dz))
    const double rs0 = bc(rs[0], 0), dz0 = bc(dz[0], 0), t0 = bc(t[0], 0);
    double acc = sb1 * (rs0 + dz0);
    for_elems<E>(1, dim, [&](int e, int l) { acc += bc(sl[e], l) * (bc(rs[e], l) + bc(dz[e], l)); });
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e;
        u[e] = a == 0 ? t[e] + sb1 * Hss : t[e] + sl[e] * Hss;
        v[e] = a == 0 ? rt[e] + acc : rt[e] + (sl[e] * (rs0 + dz0) + sb1 * (rs[e] + dz[e]));
    }
    arrow_inverse_wave<E>(dim, lane, u, v, ds);
    // dt = Cbar_t^-1 (r_t - Cs ds),  Cs = arrow(t)
    const double ds0 = bc(ds[0], 0);
    acc = t0 * ds0;
    for_elems<E>(1, dim, [&](int e, int l) { acc += bc(t[e], l) * bc(ds[e], l); });
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e;
        v[e] = a == 0 ? rt[e] - acc : rt[e] - (t[e] * ds0 + t0 * ds[e]);
        u[e] = a == 0 ? sb1 : sl[e];
    }
    arrow_inverse_wave<E>(dim, lane, u, v, dt);
#pragma unroll
    for (int e = 0; e < E; ++e) if (in[e]) {
        const int a = lane + 64 * e;
        step[d.oz() + st + a] = dz[e]; step[d.os() + st + a] = ds[e]; step[d.ot() + st + a] = dt[e];
        if (accum) { accum[d.oz() + st + a] += dz[e]; accum[d.os() + st + a] += ds[e]; accum[d.ot() + st + a] += dt[e]; }
    }
}

// ---- cone rows of residual_error = residual - H step, their norm, and the next condensed right-hand side (k_refine_local's second-order branch) ----
template <int E>
__global__ __launch_bounds__(64) void k_refine_local_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ v, const double* __restrict__ res,
                                                           const double* __restrict__ zsx, const double* __restrict__ Wsoc, double* __restrict__ e_, double* __restrict__ rsym,
                                                           double* __restrict__ t1, double* __restrict__ part, int part0) {
    inst_shift(bt.b, w, v, res, zsx, Wsoc, e_, rsym, t1, part);
    const Scalars sc = bt.scal(blockIdx.z);
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j];
    const int lane = threadIdx.x;
    const double Hss = 0.0 + sc.ep;
    double sl[E], t[E], vs[E], vz[E], vt[E], rs[E], rz[E], rt[E], u[E], vv[E], o[E], ss[E];
    bool in[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e, k = st + a; in[e] = a < dim;
        sl[e] = in[e] ? w[d.os() + k] : 0.0; t[e] = in[e] ? w[d.ot() + k] : 0.0;
        vs[e] = in[e] ? v[d.os() + k] : 0.0; vz[e] = in[e] ? v[d.oz() + k] : 0.0; vt[e] = in[e] ? v[d.ot() + k] : 0.0;
        const double hs = (0.0 + sc.ep) * vs[e] - vz[e] - vt[e];
        rs[e] = in[e] ? res[d.os() + k] - hs : 0.0;
        const double hz = (in[e] ? zsx[d.ne + k] : 0.0) + (-vs[e] + (0.0 - sc.ed) * vz[e]);
        rz[e] = in[e] ? res[d.oz() + k] - hz : 0.0;
    }
    const double t0 = bc(t[0], 0), sl0 = bc(sl[0], 0), vs0 = bc(vs[0], 0), vt0 = bc(vt[0], 0);
    double ht0 = t0 * vs0 + (sl0 - sc.ed) * vt0;
    for_elems<E>(1, dim, [&](int e, int l) { ht0 += bc(t[e], l) * bc(vs[e], l) + bc(sl[e], l) * bc(vt[e], l); });
    double m = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e, k = st + a;
        double ht = t[e] * vs0 + sl[e] * vt0;
        ht += t0 * vs[e] + (sl0 - sc.ed) * vt[e];
        if (a == 0) ht = ht0;
        rt[e] = in[e] ? res[d.ot() + k] - ht : 0.0;
        if (in[e]) {
            e_[d.os() + k] = rs[e]; e_[d.oz() + k] = rz[e]; e_[d.ot() + k] = rt[e];
            m = fmax(m, fmax(fmax(fabs(rs[e]), fabs(rz[e])), fabs(rt[e])));
        }
    }
    const double sb1 = sl0 - sc.ed, rs0 = bc(rs[0], 0);
    double acc = sb1 * rs0;
    for_elems<E>(1, dim, [&](int e, int l) { acc += bc(sl[e], l) * bc(rs[e], l); });
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int a = lane + 64 * e;
        u[e] = a == 0 ? t[e] + sb1 * Hss : t[e] + sl[e] * Hss;
        vv[e] = a == 0 ? acc + rt[e] : (sl[e] * rs0 + sb1 * rs[e]) + rt[e];
    }
    arrow_inverse_wave<E>(dim, lane, u, vv, o);
#pragma unroll
    for (int e = 0; e < E; ++e) { o[e] = rz[e] + o[e]; if (in[e]) rsym[d.nx + d.ne + st + lane + 64 * e] = o[e]; }
    w_times<E>(dim, lane, in, Wsoc + cd.soc_woff[j], o, ss);
#pragma unroll
    for (int e = 0; e < E; ++e) if (in[e]) t1[d.ne + st + lane + 64 * e] = ss[e];
    m = wave_max(m);
    if (lane == 0) part[part0 + blockIdx.x] = m;
}

// ---- launchers (called by the launchers of schur.hip / vectors.hip right after their own kernel, only when the handle has wide cones) ----------
// E = elements per lane for the widest cone of the handle
static int wide_E(const calipso_hip_solver* s) { const int m = s->d.max_dim; return m <= 64 ? 1 : (m <= 128 ? 2 : (m <= 256 ? 4 : (m <= 512 ? 8 : 16))); }
#define WIDE_DISPATCH(E_, CALL) do { switch (E_) { case 1: { constexpr int E = 1; CALL; } break; case 2: { constexpr int E = 2; CALL; } break; case 4: { constexpr int E = 4; CALL; } break; case 8: { constexpr int E = 8; CALL; } break; default: { constexpr int E = 16; CALL; } } } while (0)

void launch_cone_weights_wide(calipso_hip_solver* s) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    const int dim = s->d.max_dim, E_ = wide_E(s);
    // the d x (d + 1) block of the widest cone in LDS when it fits (dimension <= 128), in the cone scratch of the handle (2 d^2 doubles per cone) otherwise
    const bool in_lds = dim <= 128;
    const size_t lds = sizeof(double) * (64 * (size_t)E_ + (in_lds ? (size_t)dim * (dim + 1) : 0));
    WIDE_DISPATCH(E_, {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_cone_weights_wide<E>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_cone_weights_wide<E>, dim3(s->d.n_wide, 1, B.b.n), dim3(64), lds, s->stream, B, s->d, s->cone, s->solution, s->Bsoc, s->Wsoc, s->icount,
                           in_lds ? (double*)nullptr : s->socwork);
    });
}
void launch_residual_symmetric_wide(calipso_hip_solver* s, const double* res, int p, double* rsym, double* t1) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    WIDE_DISPATCH(wide_E(s), hipLaunchKernelGGL(k_residual_symmetric_wide<E>, dim3(s->d.n_wide, p, p > 1 ? 1 : B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, res, s->Wsoc, rsym, t1));
}
void launch_recover_wide(calipso_hip_solver* s, const double* res, int p, const double* rsym, const double* t2, double* dsym, double* step, double* accumulate, int zsx_mode) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    WIDE_DISPATCH(wide_E(s), hipLaunchKernelGGL(k_recover_wide<E>, dim3(s->d.n_wide, p, p > 1 ? 1 : B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, res, rsym, t2, s->Wsoc, dsym, step,
                                               accumulate, zsx_mode ? s->zsx : (double*)nullptr, zsx_mode));
}
void launch_refine_local_wide(calipso_hip_solver* s, int part0) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    WIDE_DISPATCH(wide_E(s), hipLaunchKernelGGL(k_refine_local_wide<E>, dim3(s->d.n_wide, 1, B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, s->step, s->residual, s->zsx, s->Wsoc,
                                               s->residual_error, s->residual_symmetric, s->t1, s->refpart, part0));
}

}  // namespace calipso
