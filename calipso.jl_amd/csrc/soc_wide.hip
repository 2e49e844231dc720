// soc_wide.hip — second-order cones of dimension > 4: ONE WAVEFRONT PER CONE, lane a holds element a of every cone vector (dimension <= 64 = one
// wave64), so nothing lives in private arrays (round 2: one thread per cone with 64-double arrays in scratch memory).
//   cones/second_order.jl:50-65            second_order_vector_inverse (the closed-form arrow inverse)       -> arrow_inverse_wave
//   residual_jacobian_variables.jl:151-164  K_zz block of a cone, column by column                            -> k_cone_weights_wide
//   residual.jl:78-99                       condensed right-hand side of the cone rows                        -> k_residual_symmetric_wide
//   search_direction.jl:83-101              recovery of (ds, dt) of a cone                                    -> k_recover_wide
//   iterative_refinement.jl:9,39 (cone rows of residual - H step) + the next condensed right-hand side       -> k_refine_local_wide
// The kernels of schur.hip / vectors.hip keep the register path for cones of dimension <= 4 (friction cones, SOC2 / SOC3: BASELINE's sizes)
// and skip the wide ones; these kernels run right behind them, only on handles that HAVE wide cones (portfolio: dimension 12), one 64-thread
// workgroup per wide cone.  Sums run over the elements in index order with the operand broadcast by v_readlane, i.e. in the reference's own
// operation order (the reductions are NOT tree-shaped): what a lane computes is what the sequential loop computed for that index.
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

__device__ __forceinline__ double bc(double v, int lane) {     // element `lane` of a cone vector (lane: wave-uniform)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// out = arrow(u)^-1 x, lane a holds u[a], x[a] and receives out[a] (device_utils.hpp: arrow_inverse, same operations in the same order)
__device__ __forceinline__ double arrow_inverse_wave(int n, int a, double u, double x) {
    const double u0 = bc(u, 0), x0 = bc(x, 0);
    double uu = 0.0;
    for (int i = 1; i < n; ++i) { const double ui = bc(u, i); uu += ui * ui; }
    const double alpha = -1.0 / (u0 * u0) * uu;
    const double beta = 1.0 / (1.0 + alpha);
    double d0 = 0.0;
    for (int i = 1; i < n; ++i) d0 += (bc(u, i) / u0) * bc(x, i);
    const double x0_1 = x0 - d0;
    const double v = x - beta * ((u / u0) * x0_1);
    double d1 = 0.0;
    for (int i = 1; i < n; ++i) d1 += (bc(u, i) / u0) * bc(v, i);
    const double x2_1 = x0 - d1;
    return a == 0 ? 1.0 / u0 * x2_1 : 1.0 / u0 * v;
}

// ---- K_zz block, its triu-symmetrised LDL^T (pivot signs -> inertia) and W = -(B_sym)^-1 ----------------------------------------------------
constexpr int LDM = MAX_SOC_DIM + 1;
__global__ __launch_bounds__(64) void k_cone_weights_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, double* __restrict__ Bsoc,
                                                           double* __restrict__ Wsoc, int* __restrict__ icount) {
    __shared__ double M[MAX_SOC_DIM * LDM];       // the symmetrised block, then its L (below) and D (diagonal)
    __shared__ double ycol[MAX_SOC_DIM];
    inst_shift(bt.b, w, Bsoc, Wsoc);
    inst_shift_i(bt.b, icount);
    const Scalars sc = bt.scal(blockIdx.z);
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j], off = cd.soc_woff[j];
    const int a = threadIdx.x;
    const bool in = a < dim;
    double* B = Bsoc + off;
    double* W = Wsoc + off;
    const double Hss = 0.0 + sc.ep;
    const double sl = in ? w[d.os() + st + a] : 0.0, t = in ? w[d.ot() + st + a] : 0.0;
    const double sb1 = bc(sl, 0) - sc.ed;
    const double u = a == 0 ? t + sb1 * Hss : t + sl * Hss;
    // B = -(Cs + Cbar_t P)^-1 Cbar_t + D, column by column; only triu(B) enters the factorisation (linear_solver.jl:23)
    for (int col = 0; col < dim; ++col) {
        const double slc = bc(sl, col);
        const double c = (a == col) ? sb1 : (col == 0 ? sl : (a == 0 ? slc : 0.0));
        const double o = arrow_inverse_wave(dim, a, u, c);
        double bv = 0.0 - o;
        if (a == col) bv += (0.0 - sc.ed);
        if (in) {
            B[a + col * dim] = bv;
            if (a <= col) { M[a * LDM + col] = bv; M[col * LDM + a] = bv; }
        }
    }
    __syncthreads();
    int pos = 0, nonpos = 0, zero = 0;
    for (int jj = 0; jj < dim; ++jj) {
        const double dj = M[jj * LDM + jj];
        pos += dj > 0.0; nonpos += dj <= 0.0; zero += dj == 0.0;
        const double yij = (in && a > jj) ? M[a * LDM + jj] : 0.0;
        if (in && a > jj) ycol[a] = yij;
        __syncthreads();
        if (in && a > jj) {
            const double l = yij / dj;
            for (int k = jj + 1; k <= a; ++k) M[a * LDM + k] -= l * (k == a ? yij : ycol[k]);    // y_k = unscaled column entry
            M[a * LDM + jj] = l;
        }
        __syncthreads();
    }
    if (a == 0) { atomicAdd(&icount[0], pos); atomicAdd(&icount[1], nonpos); atomicAdd(&icount[2], zero); }
    // W = -(B_sym)^-1: L D L' x = e_col, forward / scale / backward with the finished entries broadcast
    const double dinv_a = in ? M[a * LDM + a] : 1.0;
    for (int col = 0; col < dim; ++col) {
        double o = (a == col) ? 1.0 : 0.0;
        for (int k = 0; k + 1 < dim; ++k) { const double ok = bc(o, k); if (in && a > k) o -= M[a * LDM + k] * ok; }
        o /= dinv_a;
        for (int k = dim - 1; k >= 1; --k) { const double ok = bc(o, k); if (a < k) o -= M[k * LDM + a] * ok; }
        if (in) W[a + col * dim] = -o;
    }
}

// ---- condensed right-hand side of the rows of one cone: b_z = r_z + U^-1 (Cbar_t r_s + r_t), t1 = W b_z -------------------------------------
__global__ __launch_bounds__(64) void k_residual_symmetric_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ res_,
                                                                 const double* __restrict__ Wsoc, double* __restrict__ rsym_, double* __restrict__ t1_) {
    inst_shift(bt.b, w, res_, Wsoc, rsym_, t1_);
    const Scalars sc = bt.scal(blockIdx.z);
    const double* res = res_ + (size_t)blockIdx.y * d.N;
    double* rsym = rsym_ + (size_t)blockIdx.y * d.n;
    double* t1 = t1_ + (size_t)blockIdx.y * d.m;
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j];
    const int a = threadIdx.x;
    const bool in = a < dim;
    const double Hss = 0.0 + sc.ep;
    const double sl = in ? w[d.os() + st + a] : 0.0, t = in ? w[d.ot() + st + a] : 0.0;
    const double rs = in ? res[d.os() + st + a] : 0.0, rt = in ? res[d.ot() + st + a] : 0.0, rz = in ? res[d.oz() + st + a] : 0.0;
    const double sb1 = bc(sl, 0) - sc.ed, rs0 = bc(rs, 0);
    const double u = a == 0 ? t + sb1 * Hss : t + sl * Hss;
    double acc = sb1 * rs0;
    for (int k = 1; k < dim; ++k) acc += bc(sl, k) * bc(rs, k);
    const double v = a == 0 ? acc + rt : (sl * rs0 + sb1 * rs) + rt;
    double o = arrow_inverse_wave(dim, a, u, v);
    o = rz + o;
    if (in) rsym[d.nx + d.ne + st + a] = o;
    const double* W = Wsoc + cd.soc_woff[j];
    double s = 0.0;
    for (int b = 0; b < dim; ++b) { const double ob = bc(o, b); if (in) s += W[a + b * dim] * ob; }
    if (in) t1[d.ne + st + a] = s;
}

// ---- dz back-substitution + (ds, dt) recovery of one cone (k_recover's second-order branch) --------------------------------------------------
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
    const int a = threadIdx.x;
    const bool in = a < dim;
    const double Hss = 0.0 + sc.ep;
    const double sl = in ? w[d.os() + st + a] : 0.0, t = in ? w[d.ot() + st + a] : 0.0;
    const double rs = in ? res[d.os() + st + a] : 0.0, rt = in ? res[d.ot() + st + a] : 0.0;
    const double tt = in ? t2[d.ne + st + a] : 0.0;
    const double o0 = in ? b[d.nx + d.ne + st + a] - tt : 0.0;
    if (zsx_mode && in) zsx[d.ne + st + a] = zsx_mode == 1 ? tt : zsx[d.ne + st + a] + tt;
    const double* W = Wsoc + cd.soc_woff[j];
    double s = 0.0;
    for (int c = 0; c < dim; ++c) { const double oc = bc(o0, c); if (in) s += W[a + c * dim] * oc; }
    const double dz = -1.0 * s;
    if (in) dsym[d.nx + d.ne + st + a] = dz;
    const double sb1 = bc(sl, 0) - sc.ed;
    double u = a == 0 ? t + sb1 * Hss : t + sl * Hss;
    // ds = U^-1 (r_t + Cbar_t (r_s + dz))
    const double rs0 = bc(rs, 0), dz0 = bc(dz, 0), t0 = bc(t, 0);
    double acc = sb1 * (rs0 + dz0);
    for (int k = 1; k < dim; ++k) acc += bc(sl, k) * (bc(rs, k) + bc(dz, k));
    double v = a == 0 ? rt + acc : rt + (sl * (rs0 + dz0) + sb1 * (rs + dz));
    const double ds = arrow_inverse_wave(dim, a, u, v);
    // dt = Cbar_t^-1 (r_t - Cs ds),  Cs = arrow(t)
    const double ds0 = bc(ds, 0);
    acc = t0 * ds0;
    for (int k = 1; k < dim; ++k) acc += bc(t, k) * bc(ds, k);
    v = a == 0 ? rt - acc : rt - (t * ds0 + t0 * ds);
    u = a == 0 ? sb1 : sl;
    const double dt = arrow_inverse_wave(dim, a, u, v);
    if (in) {
        step[d.oz() + st + a] = dz; step[d.os() + st + a] = ds; step[d.ot() + st + a] = dt;
        if (accum) { accum[d.oz() + st + a] += dz; accum[d.os() + st + a] += ds; accum[d.ot() + st + a] += dt; }
    }
}

// ---- cone rows of residual_error = residual - H step, their norm, and the next condensed right-hand side (k_refine_local's second-order branch) ----
__global__ __launch_bounds__(64) void k_refine_local_wide(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ v, const double* __restrict__ res,
                                                           const double* __restrict__ zsx, const double* __restrict__ Wsoc, double* __restrict__ e, double* __restrict__ rsym,
                                                           double* __restrict__ t1, double* __restrict__ part, int part0) {
    inst_shift(bt.b, w, v, res, zsx, Wsoc, e, rsym, t1, part);
    const Scalars sc = bt.scal(blockIdx.z);
    const int j = cd.wide[blockIdx.x];
    const int st = cd.soc_start[j], dim = cd.soc_dim[j];
    const int a = threadIdx.x;
    const bool in = a < dim;
    const double Hss = 0.0 + sc.ep;
    const int k = st + a;
    const double sl = in ? w[d.os() + k] : 0.0, t = in ? w[d.ot() + k] : 0.0;
    const double vs = in ? v[d.os() + k] : 0.0, vz = in ? v[d.oz() + k] : 0.0, vt = in ? v[d.ot() + k] : 0.0;
    const double hs = (0.0 + sc.ep) * vs - vz - vt;
    const double rs = in ? res[d.os() + k] - hs : 0.0;
    const double hz = (in ? zsx[d.ne + k] : 0.0) + (-vs + (0.0 - sc.ed) * vz);
    const double rz = in ? res[d.oz() + k] - hz : 0.0;
    const double t0 = bc(t, 0), sl0 = bc(sl, 0), vs0 = bc(vs, 0), vt0 = bc(vt, 0);
    double ht0 = t0 * vs0 + (sl0 - sc.ed) * vt0;
    for (int q = 1; q < dim; ++q) ht0 += bc(t, q) * bc(vs, q) + bc(sl, q) * bc(vt, q);
    double ht = t * vs0 + sl * vt0;
    ht += t0 * vs + (sl0 - sc.ed) * vt;
    if (a == 0) ht = ht0;
    const double rt = in ? res[d.ot() + k] - ht : 0.0;
    double m = 0.0;
    if (in) {
        e[d.os() + k] = rs; e[d.oz() + k] = rz; e[d.ot() + k] = rt;
        m = fmax(fmax(fabs(rs), fabs(rz)), fabs(rt));
    }
    const double sb1 = sl0 - sc.ed, rs0 = bc(rs, 0);
    const double u = a == 0 ? t + sb1 * Hss : t + sl * Hss;
    double acc = sb1 * rs0;
    for (int q = 1; q < dim; ++q) acc += bc(sl, q) * bc(rs, q);
    const double vv = a == 0 ? acc + rt : (sl * rs0 + sb1 * rs) + rt;
    double o = arrow_inverse_wave(dim, a, u, vv);
    o = rz + o;
    if (in) rsym[d.nx + d.ne + k] = o;
    const double* W = Wsoc + cd.soc_woff[j];
    double ss = 0.0;
    for (int b2 = 0; b2 < dim; ++b2) { const double ob = bc(o, b2); if (in) ss += W[a + b2 * dim] * ob; }
    if (in) t1[d.ne + k] = ss;
    m = wave_max(m);
    if (a == 0) part[part0 + blockIdx.x] = m;
}

// ---- launchers (called by the launchers of schur.hip / vectors.hip right after their own kernel, only when the handle has wide cones) ----------
void launch_cone_weights_wide(calipso_hip_solver* s) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_cone_weights_wide, dim3(s->d.n_wide, 1, B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, s->Bsoc, s->Wsoc, s->icount);
}
void launch_residual_symmetric_wide(calipso_hip_solver* s, const double* res, int p, double* rsym, double* t1) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_residual_symmetric_wide, dim3(s->d.n_wide, p, p > 1 ? 1 : B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, res, s->Wsoc, rsym, t1);
}
void launch_recover_wide(calipso_hip_solver* s, const double* res, int p, const double* rsym, const double* t2, double* dsym, double* step, double* accumulate, int zsx_mode) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_recover_wide, dim3(s->d.n_wide, p, p > 1 ? 1 : B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, res, rsym, t2, s->Wsoc, dsym, step,
                       accumulate, zsx_mode ? s->zsx : (double*)nullptr, zsx_mode);
}
void launch_refine_local_wide(calipso_hip_solver* s, int part0) {
    if (!s->d.n_wide) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_refine_local_wide, dim3(s->d.n_wide, 1, B.b.n), dim3(64), 0, s->stream, B, s->d, s->cone, s->solution, s->step, s->residual, s->zsx, s->Wsoc,
                       s->residual_error, s->residual_symmetric, s->t1, s->refpart, part0);
}

}  // namespace calipso
