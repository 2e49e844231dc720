// vectors.hip — O(N) kernels of the hot path: residual, condensed right-hand side, step recovery, merit, violation,
// candidate/accept updates, the vector part of the matrix-free H*v, dense K materialisation.
// N is ~10^4: elementwise kernels are a handful of workgroups; every reduction is ONE workgroup with a fixed
// summation order (deterministic), its result written to the device scalar block `dscal`.
#include <algorithm>
#include <mutex>

#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

constexpr int RT = 1024;  // reduction workgroup size

// residual!(data, problem, idx, solution, kappa, rho, lambda)   residual.jl:1-51
__global__ void k_residual(BatchSc bt, Dims d, const double* __restrict__ w, const double* __restrict__ lam,
                           const double* __restrict__ fx, const double* __restrict__ gyx, const double* __restrict__ hzx,
                           const double* __restrict__ g, const double* __restrict__ hc, const double* __restrict__ prod,
                           const double* __restrict__ targ, double* __restrict__ res) {
    inst_shift(bt.b, w, lam, fx, gyx, hzx, g, hc, prod, targ, res);
    const Scalars sc = bt.scal(blockIdx.z);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    double v;
    if (i < d.orr()) {
        v = fx[i];
        v += gyx[i];
        v += hzx[i];
    } else if (i < d.os()) {
        const int k = i - d.orr();
        v = lam[k] + sc.rho * w[d.orr() + k] - w[d.oy() + k];
    } else if (i < d.oy()) {
        const int k = i - d.os();
        v = -w[d.oz() + k] - w[d.ot() + k];
    } else if (i < d.oz()) {
        const int k = i - d.oy();
        v = g[k];
        v -= w[d.orr() + k];
    } else if (i < d.ot()) {
        const int k = i - d.oz();
        v = hc[k];
        v -= w[d.os() + k];
    } else {
        const int k = i - d.ot();
        v = prod[k] - sc.kappa * targ[k];
    }
    res[i] = v;
}

void launch_residual(calipso_hip_solver* s) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_residual, dim3((s->d.N + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B, s->d, s->solution, s->lambda, s->fx,
                       s->gyx, s->hzx, s->g, s->hc, s->cone_product, s->cone_target, s->residual);
}

__device__ __forceinline__ double pnorm_term(double v, int ptype) { return ptype == 2 ? v * v : fabs(v); }
// |v| as a term of the infinity norm of a refinement residual, with NaN -> +inf: fmax / block_max / the atomic maximum on bit patterns drop a NaN, and a norm of 0 would
// read as "converged" (iterative_refinement.jl:14-16 sees norm(., Inf) = NaN, NaN <= tol false, and goes on to the failure path: so does +inf)
__device__ __forceinline__ double rabs(double v) { return v != v ? __longlong_as_double(0x7ff0000000000000ll) : fabs(v); }

// A read-back without a launch of its own: the LAST kernel in front of a scalar read-back (one workgroup, a single handle) copies dscal[first .. first +
// count) — its own results and those of the kernels before it on the stream — to the handle's mapped host mirror and then stores the sequence number the
// host spins on (api.hip: wait_published), as k_publish_words would.  hpub = nullptr: nothing (groups gather by their own kernels).
__device__ __forceinline__ void publish_tail(const double* dscal, int first, int count, double* hpub, unsigned long long* hseq, unsigned long long seq) {
    if (!hpub) return;
    __syncthreads();                                             // this kernel's own stores to dscal are done (they are re-read past the L1 below)
    if ((int)threadIdx.x < count) hpub[first + threadIdx.x] = __hip_atomic_load(dscal + first + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the reductions of solve.jl:130-135,332-333 and optimality_error.jl:1-27 -> dscal[8..17]
__device__ __forceinline__ void violations_body(const Dims& d, int ptype, const double* __restrict__ res, const double* __restrict__ w,
                                                const double* __restrict__ g, const double* __restrict__ prod, double* __restrict__ dscal) {
    const int tid = threadIdx.x;
    double rp = 0.0, rprim = 0.0, ry = 0.0, rz = 0.0, rt = 0.0, y1 = 0.0, z1 = 0.0, t1 = 0.0, ginf = 0.0, pinf = 0.0;
    // one workgroup walks the whole vectors: eight of a thread's entries are fetched before the first is used (the loops were one memory round trip per
    // entry); the order in which a thread combines its entries is unchanged
    for (int i0 = tid; i0 < d.N; i0 += 8 * RT) {
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * RT; vv[u] = i < d.N ? res[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * RT;
            if (i < d.N) {
                const double v = vv[u];
                if (ptype == 0) rp = fmax(rp, fabs(v)); else rp += pnorm_term(v, ptype);
                const double a = fabs(v);
                if (i < d.n) rprim = fmax(rprim, a);
                else if (i < d.oz()) ry = fmax(ry, a);
                else if (i < d.ot()) rz = fmax(rz, a);
                else rt = fmax(rt, a);
            }
        }
    }
    for (int i0 = tid; i0 < d.ne; i0 += 4 * RT) {
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * RT; const bool in = i < d.ne; a[u] = in ? w[d.oy() + i] : 0.0; b[u] = in ? g[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * RT < d.ne) { y1 += fabs(a[u]); ginf = fmax(ginf, fabs(b[u])); }
    }
    for (int i0 = tid; i0 < d.nc; i0 += 4 * RT) {
        double a[4], b[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * RT; const bool in = i < d.nc; a[u] = in ? w[d.oz() + i] : 0.0; b[u] = in ? w[d.ot() + i] : 0.0; c[u] = in ? prod[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * RT < d.nc) { z1 += fabs(a[u]); t1 += fabs(b[u]); pinf = fmax(pinf, fabs(c[u])); }
    }
    // the ten reductions share ONE barrier round: every wavefront reduces its ten values by shuffles, lane 0 parks them in LDS, and after the barrier
    // thread q combines the per-wave values of quantity q in wave order — the operations and their order are those of ten block_sum / block_max calls
    __shared__ double red[10][RT / 64];
    const int lane = tid & 63, wv = tid >> 6;
    const double w0 = (ptype == 0) ? wave_max(rp) : wave_sum(rp);
    const double w1 = wave_max(rprim), w2 = wave_max(ry), w3 = wave_max(rz), w4 = wave_max(rt);
    const double w5 = wave_sum(y1), w6 = wave_sum(z1), w7 = wave_sum(t1), w8 = wave_max(ginf), w9 = wave_max(pinf);
    if (lane == 0) {
        red[0][wv] = w0; red[1][wv] = w1; red[2][wv] = w2; red[3][wv] = w3; red[4][wv] = w4;
        red[5][wv] = w5; red[6][wv] = w6; red[7][wv] = w7; red[8][wv] = w8; red[9][wv] = w9;
    }
    __syncthreads();
    if (tid < 10) {
        const bool is_max = tid == 0 ? ptype == 0 : (tid <= 4 || tid >= 8);
        double r = 0.0;
        for (int i = 0; i < RT / 64; ++i) r = is_max ? fmax(r, red[tid][i]) : r + red[tid][i];
        dscal[8 + tid] = (tid == 0 && ptype == 2) ? sqrt(r) : r;
    }
}
__global__ __launch_bounds__(RT) void k_violations(Batch bt, Dims d, int ptype, const double* __restrict__ res, const double* __restrict__ w,
                                                    const double* __restrict__ g, const double* __restrict__ prod,
                                                    double* __restrict__ dscal, int pub_first, int pub_count, double* __restrict__ hpub,
                                                    unsigned long long* __restrict__ hseq, unsigned long long seq) {
    inst_shift(bt, res, w, g, prod, dscal);
    violations_body(d, ptype, res, w, g, prod, dscal);
    publish_tail(dscal, pub_first, pub_count, hpub, hseq, seq);
}

static int norm_type(double p) { return p == 1.0 ? 1 : (p == 2.0 ? 2 : 0); }

void launch_violations(calipso_hip_solver* s, int pub_first, int pub_count) {
    const BatchSc B = batch_of(s);
    const bool pub = pub_count > 0 && !s->cur;               // a single handle: the kernel publishes the scalars of the read-back that follows it
    hipLaunchKernelGGL(k_violations, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B.b, s->d, norm_type(s->opt.residual_norm), s->residual, s->solution,
                       s->g, s->cone_product, s->dscal, pub_first, pub_count, pub ? s->hscal_dev : (double*)nullptr,
                       pub ? s->hseq_dev : (unsigned long long*)nullptr, pub ? ++s->pub_seq : 0ULL);
}

// residual_symmetric!   residual.jl:53-101 (condensed right-hand side b).  The same kernel also emits the first operands of
// the condensed solve: xbuf = [b_x; 0 padding] and t1 = Omega b_m (omega_y b_y ; Omega_z b_z), see solvek.hip.
__global__ void k_residual_symmetric(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ res_,
                                     const double* __restrict__ wz, const double* __restrict__ Wsoc, double* __restrict__ rsym_,
                                     double* __restrict__ xbuf_, double* __restrict__ t1_) {
    inst_shift(bt.b, w, res_, wz, Wsoc, rsym_, xbuf_, t1_);
    const Scalars sc = bt.scal(blockIdx.z);
    // blockIdx.y = right-hand-side column (differentiate!: one column per parameter); columns are N / n / NP / m apart
    const double* res = res_ + (size_t)blockIdx.y * d.N;
    double* rsym = rsym_ + (size_t)blockIdx.y * d.n;
    double* xbuf = xbuf_ + (size_t)blockIdx.y * d.NP;
    double* t1 = t1_ + (size_t)blockIdx.y * d.m;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double Hrr = sc.rho + sc.ep;   // H[r,r] = rho then += eps_p  (residual_jacobian_variables.jl:60,87)
    const double Hss = 0.0 + sc.ep;      // H[s,s]
    if (i < d.NP) {
        const double v = i < d.nx ? res[i] : 0.0;
        if (i < d.nx) rsym[i] = v;
        xbuf[i] = v;
        return;
    }
    const int e = i - d.NP;              // index into the constraint part
    if (e < d.ne) {
        double v = res[d.oy() + e];
        v += res[d.orr() + e] / Hrr;
        rsym[d.nx + e] = v;
        const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
        t1[e] = omega_y * v;
    } else if (e < d.ne + d.q) {
        const int k = e - d.ne;
        const double Sb = w[d.os() + k] - sc.ed, Ti = w[d.ot() + k], Pi = Hss;
        double v = res[d.oz() + k];
        v += (res[d.ot() + k] + Sb * res[d.os() + k]) / (Ti + Sb * Pi);
        rsym[d.nx + d.ne + k] = v;
        t1[d.ne + k] = wz[k] * v;
    } else if (e < d.ne + d.q + d.n_soc) {
        // one lane per second-order cone of dimension <= 4:  b_z[soc] = r_z + U^-1 (Cbar_t r_s + r_t),  U = Cs + Cbar_t P  (registers, loops unrolled
        // to constant indices); wider cones: k_residual_symmetric_wide (soc_wide.hip), one wavefront per cone
        const int j = e - d.ne - d.q;
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        if (dim > 4) return;
        constexpr int MD = 4;
        double sl[MD], t[MD], rs[MD], rt[MD], rz[MD], u[MD], v[MD], o[MD], W[MD * MD];
        const int woff = cd.soc_woff[j];
#pragma unroll
        for (int a = 0; a < MD; ++a) {
            const bool in = a < dim;
            sl[a] = in ? w[d.os() + st + a] : 0.0; t[a] = in ? w[d.ot() + st + a] : 0.0;
            rs[a] = in ? res[d.os() + st + a] : 0.0; rt[a] = in ? res[d.ot() + st + a] : 0.0; rz[a] = in ? res[d.oz() + st + a] : 0.0;
            u[a] = 0.0; v[a] = 0.0; o[a] = 0.0;
        }
#pragma unroll
        for (int e2 = 0; e2 < MD * MD; ++e2) W[e2] = 0.0;
#pragma unroll
        for (int c = 0; c < MD; ++c)
#pragma unroll
            for (int a = 0; a < MD; ++a) if (a < dim && c < dim) W[a + c * MD] = Wsoc[woff + a + c * dim];
        const double sb1 = sl[0] - sc.ed;
        u[0] = t[0] + sb1 * Hss;
#pragma unroll
        for (int k = 1; k < MD; ++k) if (k < dim) u[k] = t[k] + sl[k] * Hss;
        double acc = sb1 * rs[0];
#pragma unroll
        for (int k = 1; k < MD; ++k) if (k < dim) acc += sl[k] * rs[k];
        v[0] = acc + rt[0];
#pragma unroll
        for (int k = 1; k < MD; ++k) if (k < dim) v[k] = (sl[k] * rs[0] + sb1 * rs[k]) + rt[k];
        arrow_inverse_small<MD>(dim, u, v, o);
#pragma unroll
        for (int k = 0; k < MD; ++k) if (k < dim) { o[k] = rz[k] + o[k]; rsym[d.nx + d.ne + st + k] = o[k]; }
#pragma unroll
        for (int a = 0; a < MD; ++a) if (a < dim) {
            double ss = 0.0;
#pragma unroll
            for (int b2 = 0; b2 < MD; ++b2) if (b2 < dim) ss += W[a + b2 * MD] * o[b2];
            t1[d.ne + st + a] = ss;
        }
    }
}

void launch_residual_symmetric(calipso_hip_solver* s, const double* res) {
    const int work = s->d.NP + s->d.ne + s->d.q + s->d.n_soc;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_residual_symmetric, dim3((work + 127) / 128, 1, B.b.n), dim3(128), 0, s->stream, B, s->d, s->cone, s->solution, res,
                       s->wz, s->Wsoc, s->residual_symmetric, s->xbuf, s->t1);
    launch_residual_symmetric_wide(s, res, 1, s->residual_symmetric, s->t1);
}
void launch_residual_symmetric_multi(calipso_hip_solver* s, const double* res, int p, double* rsym, double* xbuf, double* t1) {
    const int work = s->d.NP + s->d.ne + s->d.q + s->d.n_soc;
    const BatchSc B = batch_of(s);   // (the multi-column path is single-instance)
    hipLaunchKernelGGL(k_residual_symmetric, dim3((work + 127) / 128, p, 1), dim3(128), 0, s->stream, B, s->d, s->cone, s->solution, res,
                       s->wz, s->Wsoc, rsym, xbuf, t1);
    launch_residual_symmetric_wide(s, res, p, rsym, t1);
}

// Tail of the condensed solve + search_direction_symmetric! (search_direction.jl:38-101) in one kernel:
//   [dy; dz] = -Omega (b_m - [gx; hx] dx)      (back-substitution through the constraint pivots; t2 = [gx; hx] dx)
//   scatter (dx, dy, dz); recover dr, ds, dt (diagonal for nonnegative entries, arrow inverses for second-order cones)
//   optionally accumulate += step  (iterative_refinement.jl:34: step .+= step_correction)
__global__ void k_recover(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ res_,
                          const double* __restrict__ b_, const double* __restrict__ dx_, const double* __restrict__ t2_,
                          const double* __restrict__ wz, const double* __restrict__ Wsoc, double* __restrict__ dsym_,
                          double* __restrict__ step_, double* __restrict__ accum, double* __restrict__ zsx, int zsx_mode) {
    inst_shift(bt.b, w, res_, b_, dx_, t2_, wz, Wsoc, dsym_, step_);
    if (accum) inst_shift(bt.b, accum);
    if (zsx_mode) inst_shift(bt.b, zsx);
    const Scalars sc = bt.scal(blockIdx.z);
    // blockIdx.y = right-hand-side column (see k_residual_symmetric); dsym_ may be null for the multi-column use
    const double* res = res_ + (size_t)blockIdx.y * d.N;
    const double* b = b_ + (size_t)blockIdx.y * d.n;
    const double* dx = dx_ + (size_t)blockIdx.y * d.NP;
    const double* t2 = t2_ + (size_t)blockIdx.y * d.m;
    double* dsym = dsym_ + (size_t)blockIdx.y * d.n;
    double* step = step_ + (size_t)blockIdx.y * d.N;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const double Hrr = sc.rho + sc.ep, Hss = 0.0 + sc.ep;
    if (i < d.nx) {
        const double v = dx[i];
        dsym[i] = v; step[i] = v;
        if (accum) accum[i] += v;
    } else if (i < d.nx + d.ne) {
        const int k = i - d.nx;
        const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
        const double dy = -1.0 * omega_y * (b[d.nx + k] - t2[k]);
        dsym[d.nx + k] = dy;
        if (zsx_mode) zsx[k] = zsx_mode == 1 ? t2[k] : zsx[k] + t2[k];         // [gx; hx] step_x follows the step
        const double dr = (res[d.orr() + k] + dy) / Hrr;
        step[d.oy() + k] = dy;
        step[d.orr() + k] = dr;
        if (accum) { accum[d.oy() + k] += dy; accum[d.orr() + k] += dr; }
    } else if (i < d.nx + d.ne + d.q) {
        const int k = i - d.nx - d.ne;
        const double dz = -1.0 * wz[k] * (b[d.nx + d.ne + k] - t2[d.ne + k]);
        dsym[d.nx + d.ne + k] = dz;
        if (zsx_mode) zsx[d.ne + k] = zsx_mode == 1 ? t2[d.ne + k] : zsx[d.ne + k] + t2[d.ne + k];
        const double Sb = w[d.os() + k] - sc.ed, Ti = w[d.ot() + k], Pi = Hss;
        const double rt = res[d.ot() + k], rs = res[d.os() + k];
        const double ds = (rt + Sb * (rs + dz)) / (Ti + Sb * Pi);
        const double dt = (rt - Ti * ds) / Sb;
        step[d.oz() + k] = dz; step[d.os() + k] = ds; step[d.ot() + k] = dt;
        if (accum) { accum[d.oz() + k] += dz; accum[d.os() + k] += ds; accum[d.ot() + k] += dt; }
    } else if (i < d.nx + d.ne + d.q + d.n_soc) {
        const int j = i - d.nx - d.ne - d.q;
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        if (dim <= 4) {
            // small cones (friction cones, SOC2 / SOC3): everything the cone needs is fetched in ONE round of independent loads into registers and the
            // loops are unrolled to constant indices; the operations and their order are those of the sequential formulation (search_direction.jl:83-101)
            constexpr int MD = 4;
            double sl[MD], t[MD], rs[MD], rt[MD], bb[MD], tt[MD], zz[MD], W[MD * MD];
            const int woff = cd.soc_woff[j];
#pragma unroll
            for (int a = 0; a < MD; ++a) {
                const bool in = a < dim;
                sl[a] = in ? w[d.os() + st + a] : 0.0; t[a] = in ? w[d.ot() + st + a] : 0.0;
                rs[a] = in ? res[d.os() + st + a] : 0.0; rt[a] = in ? res[d.ot() + st + a] : 0.0;
                bb[a] = in ? b[d.nx + d.ne + st + a] : 0.0; tt[a] = in ? t2[d.ne + st + a] : 0.0;
                zz[a] = (in && zsx_mode == 2) ? zsx[d.ne + st + a] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < MD * MD; ++e) W[e] = 0.0;
#pragma unroll
            for (int c = 0; c < MD; ++c)
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim && c < dim) W[a + c * MD] = Wsoc[woff + a + c * dim];
            double u[MD], v[MD], ds[MD], o[MD], dz[MD];
#pragma unroll
            for (int a = 0; a < MD; ++a) { o[a] = bb[a] - tt[a]; u[a] = 0.0; v[a] = 0.0; ds[a] = 0.0; dz[a] = 0.0; }
            if (zsx_mode) {
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim) zsx[d.ne + st + a] = zsx_mode == 1 ? tt[a] : zz[a] + tt[a];
            }
#pragma unroll
            for (int a = 0; a < MD; ++a) if (a < dim) {
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < MD; ++c) if (c < dim) s += W[a + c * MD] * o[c];
                dz[a] = -1.0 * s;
                dsym[d.nx + d.ne + st + a] = dz[a];
            }
            const double sb1 = sl[0] - sc.ed;
            u[0] = t[0] + sb1 * Hss;
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) u[k] = t[k] + sl[k] * Hss;
            double acc = sb1 * (rs[0] + dz[0]);
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) acc += sl[k] * (rs[k] + dz[k]);
            v[0] = rt[0] + acc;
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) v[k] = rt[k] + (sl[k] * (rs[0] + dz[0]) + sb1 * (rs[k] + dz[k]));
            arrow_inverse_small<MD>(dim, u, v, ds);
            acc = t[0] * ds[0];
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) acc += t[k] * ds[k];
            v[0] = rt[0] - acc;
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) v[k] = rt[k] - (t[k] * ds[0] + t[0] * ds[k]);
            u[0] = sb1;
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) u[k] = sl[k];
            arrow_inverse_small<MD>(dim, u, v, o);
#pragma unroll
            for (int k = 0; k < MD; ++k) if (k < dim) {
                step[d.oz() + st + k] = dz[k]; step[d.os() + st + k] = ds[k]; step[d.ot() + st + k] = o[k];
                if (accum) { accum[d.oz() + st + k] += dz[k]; accum[d.os() + st + k] += ds[k]; accum[d.ot() + st + k] += o[k]; }
            }
            return;
        }
        // (dimension > 4: k_recover_wide, soc_wide.hip)
    }
}

void launch_recover(calipso_hip_solver* s, double* step, const double* res, double* accumulate, int zsx_mode) {
    const int work = s->d.nx + s->d.ne + s->d.q + s->d.n_soc;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_recover, dim3((work + 127) / 128, 1, B.b.n), dim3(128), 0, s->stream, B, s->d, s->cone, s->solution, res,
                       s->residual_symmetric, s->xbuf, s->t2, s->wz, s->Wsoc, s->step_symmetric, step, accumulate, s->zsx, zsx_mode);
    launch_recover_wide(s, res, 1, s->residual_symmetric, s->t2, s->step_symmetric, step, accumulate, zsx_mode);
}
__global__ void k_scale_inplace(size_t n, double* __restrict__ x, double a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = a * x[i];
}
// all p columns at once; rsym is reused as the dsym scratch (each lane reads b before the same lane writes dsym)
void launch_recover_multi(calipso_hip_solver* s, const double* res, int p, const double* rsym, const double* xbuf, const double* t2, double* step, double scale) {
    const int work = s->d.nx + s->d.ne + s->d.q + s->d.n_soc;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_recover, dim3((work + 127) / 128, p, 1), dim3(128), 0, s->stream, B, s->d, s->cone, s->solution, res,
                       rsym, xbuf, t2, s->wz, s->Wsoc, s->dsym_multi, step, (double*)nullptr, (double*)nullptr, 0);
    launch_recover_wide(s, res, p, rsym, t2, s->dsym_multi, step, nullptr, 0);
    if (scale != 1.0) {
        const size_t n = (size_t)s->d.N * p;
        hipLaunchKernelGGL(k_scale_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, n, step, scale);
    }
}

// candidate x, r (, s) = solution - step_size * step    solve.jl:224-229, 268-276
struct PerInst { double v[MAX_BATCH]; };
__global__ void k_axpy_points(Batch bt, Dims d, const double* __restrict__ sol, const double* __restrict__ step, double* __restrict__ cand,
                              PerInst av, int with_s) {
    inst_shift(bt, sol, step, cand);
    const double a = av.v[blockIdx.z];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lim = with_s ? d.n : d.nx + d.ne;
    if (i < lim) cand[i] = sol[i] - a * step[i];
}
void launch_axpy_points_batch(calipso_hip_solver* s, const double* step_size, int with_s) {   // one step size per covered instance
    const BatchSc B = batch_of(s);
    PerInst av;
    for (int k = 0; k < B.b.n; ++k) av.v[k] = step_size[k];
    hipLaunchKernelGGL(k_axpy_points, dim3((s->d.n + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B.b, s->d, s->solution, s->step, s->candidate,
                       av, with_s);
}
void launch_axpy_points(calipso_hip_solver* s, double step_size, int with_s) { launch_axpy_points_batch(s, &step_size, with_s); }

// accept: x,r,s <- candidate; y,z -= a*step; t <- candidate t    solve.jl:309-326
__global__ void k_accept(Batch bt, Dims d, double* __restrict__ sol, const double* __restrict__ cand, const double* __restrict__ step, PerInst av) {
    inst_shift(bt, sol, cand, step);
    const double a = av.v[blockIdx.z];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    if (i < d.n) sol[i] = cand[i];
    else if (i < d.ot()) sol[i] = sol[i] - a * step[i];
    else sol[i] = cand[i];
}
void launch_accept_batch(calipso_hip_solver* s, const double* step_size) {
    const BatchSc B = batch_of(s);
    PerInst av;
    for (int k = 0; k < B.b.n; ++k) av.v[k] = step_size[k];
    hipLaunchKernelGGL(k_accept, dim3((s->d.N + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B.b, s->d, s->solution, s->candidate, s->step, av);
}
void launch_accept(calipso_hip_solver* s, double step_size) { launch_accept_batch(s, &step_size); }

// merit(f, r, Phi, kappa, lambda, rho)   merit.jl:2-15  -> dscal[4]   (one workgroup of RT threads per instance; `sm` = RT / 64 doubles of LDS)
__device__ __forceinline__ void merit_body(const Scalars sc, const Dims& d, const double* __restrict__ point, const double* __restrict__ lam,
                                           double* __restrict__ dscal, double* sm) {
    const double* r = point + d.orr();
    double lr = 0.0, rr = 0.0;
    for (int i = threadIdx.x; i < d.ne; i += RT) { lr += lam[i] * r[i]; rr += r[i] * r[i]; }
    const double a = block_sum(lr, sm);
    const double b = block_sum(rr, sm);
    if (threadIdx.x == 0) {
        double M = 0.0;
        M += dscal[0];
        M += a + 0.5 * sc.rho * b;
        M -= sc.kappa * dscal[1];
        dscal[4] = M;
    }
}
__global__ __launch_bounds__(RT) void k_merit(BatchSc bt, Dims d, const double* __restrict__ point, const double* __restrict__ lam,
                                               double* __restrict__ dscal) {
    __shared__ double sm[RT / 64];
    inst_shift(bt.b, point, lam, dscal);
    merit_body(bt.scal(blockIdx.z), d, point, lam, dscal, sm);
}
void launch_merit(calipso_hip_solver* s, const double* point) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_merit, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B, s->d, point, s->lambda, s->dscal);
}

// merit_gradient!   merit.jl:17-31   (entry i of the instance the pointers were shifted to)
__device__ __forceinline__ void merit_gradient_entry(const Scalars sc, const Dims& d, int i, const double* __restrict__ w, const double* __restrict__ lam,
                                                     const double* __restrict__ fx, const double* __restrict__ bgrad, double* __restrict__ grad) {
    if (i >= d.n) return;
    if (i < d.nx) grad[i] = fx[i];
    else if (i < d.nx + d.ne) grad[i] = lam[i - d.nx] + sc.rho * w[d.orr() + i - d.nx];
    else grad[i] = -1.0 * sc.kappa * bgrad[i - d.nx - d.ne];
}
__global__ void k_merit_gradient(BatchSc bt, Dims d, const double* __restrict__ w, const double* __restrict__ lam,
                                 const double* __restrict__ fx, const double* __restrict__ bgrad, double* __restrict__ grad) {
    inst_shift(bt.b, w, lam, fx, bgrad, grad);
    merit_gradient_entry(bt.scal(blockIdx.z), d, blockIdx.x * blockDim.x + threadIdx.x, w, lam, fx, bgrad, grad);
}
void launch_merit_gradient(calipso_hip_solver* s) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_merit_gradient, dim3((s->d.n + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B, s->d, s->solution, s->lambda, s->fx,
                       s->barrier_gradient, s->merit_gradient);
}
// merit at the current point AND its gradient in one launch (solve.jl:112-124: the two are always wanted together there): workgroup 0 of an instance is
// k_merit, the others are k_merit_gradient with RT-thread workgroups; operation for operation the two kernels above
__global__ __launch_bounds__(RT) void k_merit_and_gradient(BatchSc bt, Dims d, const double* __restrict__ w, const double* __restrict__ lam, const double* __restrict__ fx,
                                                            const double* __restrict__ bgrad, double* __restrict__ grad, double* __restrict__ dscal) {
    __shared__ double sm[RT / 64];
    inst_shift(bt.b, w, lam, fx, bgrad, grad, dscal);
    const Scalars sc = bt.scal(blockIdx.z);
    if (blockIdx.x == 0) merit_body(sc, d, w, lam, dscal, sm);
    else merit_gradient_entry(sc, d, (blockIdx.x - 1) * RT + threadIdx.x, w, lam, fx, bgrad, grad);
}
void launch_merit_and_gradient(calipso_hip_solver* s) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_merit_and_gradient, dim3(1 + (s->d.n + RT - 1) / RT, 1, B.b.n), dim3(RT), 0, s->stream, B, s->d, s->solution, s->lambda, s->fx,
                       s->barrier_gradient, s->merit_gradient, s->dscal);
}

// constraint_violation!   constraint_violation.jl:1-13 -> dscal[5]
__device__ __forceinline__ void constraint_violation_body(const Dims& d, int ptype, const double* __restrict__ point, const double* __restrict__ g,
                                                          const double* __restrict__ hc, double* __restrict__ dscal) {
    __shared__ double sm[RT / 64];
    double acc = 0.0;
    for (int i0 = threadIdx.x; i0 < d.ne + d.nc; i0 += 4 * RT) {      // four entries of a thread in flight together, combined in the same order
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * RT; const bool in = i < d.ne + d.nc;
            a[u] = !in ? 0.0 : (i < d.ne ? g[i] : hc[i - d.ne]);
            b[u] = !in ? 0.0 : (i < d.ne ? point[d.orr() + i] : point[d.os() + i - d.ne]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * RT < d.ne + d.nc) {
            const double c = a[u] - b[u];
            if (ptype == 0) acc = fmax(acc, fabs(c)); else acc += pnorm_term(c, ptype);
        }
    }
    double r = (ptype == 0) ? block_max(acc, sm) : block_sum(acc, sm);
    if (threadIdx.x == 0) {
        if (ptype == 2) r = sqrt(r);
        dscal[5] = r / (double)(d.ne + d.nc);
    }
}
__global__ __launch_bounds__(RT) void k_constraint_violation(Batch bt, Dims d, int ptype, const double* __restrict__ point,
                                                              const double* __restrict__ g, const double* __restrict__ hc,
                                                              double* __restrict__ dscal, int pub_first, int pub_count, double* __restrict__ hpub,
                                                              unsigned long long* __restrict__ hseq, unsigned long long seq) {
    inst_shift(bt, point, g, hc, dscal);
    constraint_violation_body(d, ptype, point, g, hc, dscal);
    publish_tail(dscal, pub_first, pub_count, hpub, hseq, seq);
}
// k_violations and k_constraint_violation at the current point back to back in one workgroup (the head of the inner loop wants both: solve.jl:130-135, 170-172)
__global__ __launch_bounds__(RT) void k_violations_and_constraint(Batch bt, Dims d, int ptype, int ctype, const double* __restrict__ res, const double* __restrict__ w,
                                                                   const double* __restrict__ g, const double* __restrict__ prod, const double* __restrict__ hc,
                                                                   double* __restrict__ dscal, int pub_first, int pub_count, double* __restrict__ hpub,
                                                                   unsigned long long* __restrict__ hseq, unsigned long long seq) {
    inst_shift(bt, res, w, g, prod, hc, dscal);
    violations_body(d, ptype, res, w, g, prod, dscal);
    constraint_violation_body(d, ctype, w, g, hc, dscal);
    publish_tail(dscal, pub_first, pub_count, hpub, hseq, seq);
}
void launch_constraint_violation(calipso_hip_solver* s, const double* point, int pub_first, int pub_count) {
    const BatchSc B = batch_of(s);
    const bool pub = pub_count > 0 && !s->cur;
    hipLaunchKernelGGL(k_constraint_violation, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B.b, s->d, norm_type(s->opt.constraint_norm), point, s->g,
                       s->hc, s->dscal, pub_first, pub_count, pub ? s->hscal_dev : (double*)nullptr, pub ? s->hseq_dev : (unsigned long long*)nullptr,
                       pub ? ++s->pub_seq : 0ULL);
}

// k_merit and k_constraint_violation of a candidate back to back in one workgroup (solve.jl:242-250: the line search wants both, then the host decides): the two bodies
// as they are, one launch less on the path between the refinement and the accepted step
__global__ __launch_bounds__(RT) void k_merit_and_constraint(BatchSc bt, Dims d, int ptype, const double* __restrict__ point, const double* __restrict__ lam,
                                                              const double* __restrict__ g, const double* __restrict__ hc, double* __restrict__ dscal, int pub_first,
                                                              int pub_count, double* __restrict__ hpub, unsigned long long* __restrict__ hseq, unsigned long long seq) {
    __shared__ double sm[RT / 64];
    inst_shift(bt.b, point, lam, g, hc, dscal);
    merit_body(bt.scal(blockIdx.z), d, point, lam, dscal, sm);
    __syncthreads();
    constraint_violation_body(d, ptype, point, g, hc, dscal);
    publish_tail(dscal, pub_first, pub_count, hpub, hseq, seq);
}
void launch_merit_and_constraint(calipso_hip_solver* s, const double* point, int pub_first, int pub_count) {
    const BatchSc B = batch_of(s);
    const bool pub = pub_count > 0 && !s->cur;
    hipLaunchKernelGGL(k_merit_and_constraint, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B, s->d, norm_type(s->opt.constraint_norm), point, s->lambda, s->g, s->hc, s->dscal,
                       pub_first, pub_count, pub ? s->hscal_dev : (double*)nullptr, pub ? s->hseq_dev : (unsigned long long*)nullptr, pub ? ++s->pub_seq : 0ULL);
}

void launch_violations_and_constraint(calipso_hip_solver* s, int pub_first, int pub_count) {
    const BatchSc B = batch_of(s);
    const bool pub = pub_count > 0 && !s->cur;
    hipLaunchKernelGGL(k_violations_and_constraint, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B.b, s->d, norm_type(s->opt.residual_norm), norm_type(s->opt.constraint_norm),
                       s->residual, s->solution, s->g, s->cone_product, s->hc, s->dscal, pub_first, pub_count, pub ? s->hscal_dev : (double*)nullptr,
                       pub ? s->hseq_dev : (unsigned long long*)nullptr, pub ? ++s->pub_seq : 0ULL);
}

// d = dot(merit_gradient, step.primals)   line_search.jl:3,16 -> dscal[6]   (one workgroup of RT threads per instance)
__device__ __forceinline__ void dot_body(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, double* sm) {
    double acc = 0.0;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * RT) {             // four entries of a thread in flight together, same order of its sum
        double av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * RT; av[u] = i < n ? a[i] : 0.0; bv[u] = i < n ? b[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * RT < n) acc += av[u] * bv[u];
    }
    const double r = block_sum(acc, sm);
    if (threadIdx.x == 0) *out = r;
}
__global__ __launch_bounds__(RT) void k_dot(Batch bt, int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
    __shared__ double sm[RT / 64];
    inst_shift(bt, a, b, out);
    dot_body(n, a, b, out, sm);
}
void launch_dot_merit(calipso_hip_solver* s) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_dot, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B.b, s->d.n, s->merit_gradient, s->step, s->dscal + 6);
}
// The first candidate of the line search in ONE launch (solve.jl:206-208, 216-218, 224-229 and line_search.jl:3): x, r, s <- solution - a_s step (the residual
// search starts at the cone step size), t <- solution - a_t step, and the directional derivative dot(merit_gradient, step) -> dscal[6] by the last
// workgroup of the instance.  Entry for entry the expressions of k_cone_candidate / k_axpy_points / k_dot.
__global__ __launch_bounds__(RT) void k_first_candidate(Batch bt, Dims d, const double* __restrict__ sol, const double* __restrict__ step, double* __restrict__ cand,
                                                         PerInst as, PerInst at, const double* __restrict__ mgrad, double* __restrict__ dscal, int nb) {
    __shared__ double sm[RT / 64];
    inst_shift(bt, sol, step, cand, mgrad, dscal);
    if ((int)blockIdx.x == nb) { dot_body(d.n, mgrad, step, dscal + 6, sm); return; }
    const double a_s = as.v[blockIdx.z], a_t = at.v[blockIdx.z];
    const int i = blockIdx.x * RT + threadIdx.x;
    if (i < d.n) cand[i] = sol[i] - a_s * step[i];
    else if (i < d.n + d.nc) { const int k = d.ot() + (i - d.n); cand[k] = sol[k] - a_t * step[k]; }
}
void launch_first_candidate_batch(calipso_hip_solver* s, const double* a_s, const double* a_t) {   // one (a_s, a_t) per covered instance
    const BatchSc B = batch_of(s);
    PerInst as, at;
    for (int k = 0; k < B.b.n; ++k) { as.v[k] = a_s[k]; at.v[k] = a_t[k]; }
    const int nb = (s->d.n + s->d.nc + RT - 1) / RT;
    hipLaunchKernelGGL(k_first_candidate, dim3(nb + 1, 1, B.b.n), dim3(RT), 0, s->stream, B.b, s->d, s->solution, s->step, s->candidate, as, at, s->merit_gradient,
                       s->dscal, nb);
}
void launch_first_candidate(calipso_hip_solver* s, double a_s, double a_t) { launch_first_candidate_batch(s, &a_s, &a_t); }

// The same launch with the step sizes taken from the cone-search masks ON THE DEVICE (a single handle: api.hip queues the first candidate behind the cone search without
// a host round trip in between).  Every workgroup repeats what the host does with the published masks (host_logic.hpp: first_feasible_trial, then api.hip's repeated
// multiplication by scaling_line_search — the same IEEE operations, so the host's step sizes are these to the bit); no feasible trial: NaN (the host sees the same masks
// and raises "cone search failure").
__global__ __launch_bounds__(RT) void k_first_candidate_masks(Batch bt, Dims d, const double* __restrict__ sol, const double* __restrict__ step, double* __restrict__ cand,
                                                               const int* __restrict__ icount, double sls, int nk, const double* __restrict__ mgrad, double* __restrict__ dscal,
                                                               int nb) {
    __shared__ double sm[RT / 64];
    __shared__ double sa[2];
    inst_shift(bt, sol, step, cand, mgrad, dscal);
    inst_shift_i(bt, icount);
    if ((int)blockIdx.x == nb) { dot_body(d.n, mgrad, step, dscal + 6, sm); return; }
    if (threadIdx.x < 2) {
        const int* mask = icount + (threadIdx.x == 0 ? 6 : 32);
        int kk = -1;
        for (int k = 0; k < nk; ++k) if (!(mask[k >> 5] & (1 << (k & 31)))) { kk = k; break; }
        double a = 1.0;
        for (int k = 0; k < kk; ++k) a = sls * a;
        sa[threadIdx.x] = kk < 0 ? __longlong_as_double(0x7ff8000000000000LL) : a;
    }
    __syncthreads();
    const double a_s = sa[0], a_t = sa[1];
    const int i = blockIdx.x * RT + threadIdx.x;
    if (i < d.n) cand[i] = sol[i] - a_s * step[i];
    else if (i < d.n + d.nc) { const int k = d.ot() + (i - d.n); cand[k] = sol[k] - a_t * step[k]; }
}
void launch_first_candidate_from_masks(calipso_hip_solver* s) {      // nc > 0, a single handle; behind launch_cone_search on the same stream
    const BatchSc B = batch_of(s);
    const int nb = (s->d.n + s->d.nc + RT - 1) / RT;
    const int nk = (int)std::min<i64>(s->opt.max_cone_line_search + 1, CONE_MASK_TRIALS);
    hipLaunchKernelGGL(k_first_candidate_masks, dim3(nb + 1, 1, B.b.n), dim3(RT), 0, s->stream, B.b, s->d, s->solution, s->step, s->candidate, s->icount,
                       s->opt.scaling_line_search, nk, s->merit_gradient, s->dscal, nb);
}

// vector part of out = H v (block rows of residual_jacobian_variables.jl:1-108); the mat-vec parts were accumulated
// into out_x, out_y, out_z beforehand.
__global__ void k_Hmul_vec(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ v,
                           double* __restrict__ out) {
    inst_shift(bt.b, w, v, out);
    const Scalars sc = bt.scal(blockIdx.z);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    if (i < d.nx) {
        out[i] += sc.ep * v[i];
    } else if (i < d.os()) {
        const int k = i - d.orr();
        out[i] = (sc.rho + sc.ep) * v[i] - v[d.oy() + k];
    } else if (i < d.oy()) {
        const int k = i - d.os();
        out[i] = (0.0 + sc.ep) * v[i] - v[d.oz() + k] - v[d.ot() + k];
    } else if (i < d.oz()) {
        const int k = i - d.oy();
        out[i] += -v[d.orr() + k] + (0.0 - sc.ed) * v[i];
    } else if (i < d.ot()) {
        const int k = i - d.oz();
        out[i] += -v[d.os() + k] + (0.0 - sc.ed) * v[i];
    } else {
        const int k = i - d.ot();
        const double* sl = w + d.os(); const double* t = w + d.ot();
        const double* vs = v + d.os(); const double* vt = v + d.ot();
        const int j = cd.entry_soc[k];
        if (j < 0) {
            out[i] = t[k] * vs[k] + (sl[k] - sc.ed) * vt[k];
        } else {
            const int st = cd.soc_start[j], dim = cd.soc_dim[j];
            double acc;
            if (k == st) {   // first row of arrow(t), arrow(s) - ed*I
                acc = t[st] * vs[st] + (sl[st] - sc.ed) * vt[st];
                for (int e = 1; e < dim; ++e) acc += t[st + e] * vs[st + e] + sl[st + e] * vt[st + e];
            } else {
                acc = t[k] * vs[st] + sl[k] * vt[st];
                acc += t[st] * vs[k] + (sl[st] - sc.ed) * vt[k];
            }
            out[i] = acc;
        }
    }
}

static void hmul_matvecs(calipso_hip_solver* s, const double* v, double* out) {
    const Dims& d = s->d;
    if (s->compact) {       // structured handle: the same three products on the packed blocks
        gemv_n(s, d.nx, d.nx, s->Lxx, d.nx, v, out, 1.0, 0.0, SP_LXX);
        if (d.m) { gemv_t(s, d.m, d.nx, s->Z, d.m, v + d.oy(), out, 1.0, 1.0, SP_Z); gemv_n(s, d.m, d.nx, s->Z, d.m, v, out + d.oy(), 1.0, 0.0, SP_Z); }
        return;
    }
    gemv_n(s, d.nx, d.nx, s->Lxx, d.nx, v, out, 1.0, 0.0, SP_LXX);
    // out_x += gx'v_y + hx'v_z (y and z are adjacent in a Point) and out_yz = [gx; hx] v_x with one pass over the stacked Jacobian
    if (d.m) gemv_both(s, d.m, d.nx, s->Z, d.m, v, v + d.oy(), out + d.oy(), out, 1.0, SP_Z);
}

void launch_Hmul(calipso_hip_solver* s, const double* v, double* out) {
    hmul_matvecs(s, v, out);
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_Hmul_vec, dim3((s->d.N + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B, s->d, s->cone, s->solution, v, out);
}

// residual_error = residual - H*step ; dscal[7] = ||residual_error||_inf   (iterative_refinement.jl:8-12,38-41).
// One workgroup: the vector part of H*step, the subtraction and the norm in a single pass over the N entries.
__global__ __launch_bounds__(RT) void k_Hmul_err(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ v,
                                                  const double* __restrict__ res, double* __restrict__ e, double* __restrict__ out) {
    __shared__ double sm[RT / 64];
    inst_shift(bt.b, w, v, res, e, out);
    const Scalars sc = bt.scal(blockIdx.z);
    const double* sl = w + d.os(); const double* t = w + d.ot();
    const double* vs = v + d.os(); const double* vt = v + d.ot();
    double m = 0.0;
    for (int i = threadIdx.x; i < d.N; i += RT) {
        double hv;
        if (i < d.nx) hv = e[i] + sc.ep * v[i];
        else if (i < d.os()) hv = (sc.rho + sc.ep) * v[i] - v[d.oy() + i - d.orr()];
        else if (i < d.oy()) { const int k = i - d.os(); hv = (0.0 + sc.ep) * v[i] - v[d.oz() + k] - v[d.ot() + k]; }
        else if (i < d.oz()) hv = e[i] + (-v[d.orr() + i - d.oy()] + (0.0 - sc.ed) * v[i]);
        else if (i < d.ot()) hv = e[i] + (-v[d.os() + i - d.oz()] + (0.0 - sc.ed) * v[i]);
        else {
            const int k = i - d.ot();
            const int j = cd.entry_soc[k];
            if (j < 0) hv = t[k] * vs[k] + (sl[k] - sc.ed) * vt[k];
            else {
                const int st = cd.soc_start[j], dim = cd.soc_dim[j];
                if (k == st) {
                    hv = t[st] * vs[st] + (sl[st] - sc.ed) * vt[st];
                    for (int q = 1; q < dim; ++q) hv += t[st + q] * vs[st + q] + sl[st + q] * vt[st + q];
                } else {
                    hv = t[k] * vs[st] + sl[k] * vt[st];
                    hv += t[st] * vs[k] + (sl[st] - sc.ed) * vt[k];
                }
            }
        }
        const double r = res[i] - hv;
        e[i] = r;
        m = fmax(m, rabs(r));
    }
    const double r = block_max(m, sm);
    if (threadIdx.x == 0) *out = r;
}

void launch_residual_error(calipso_hip_solver* s, const double* step) {
    hmul_matvecs(s, step, s->residual_error);
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_Hmul_err, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B, s->d, s->cone, s->solution, step, s->residual, s->residual_error, s->dscal + 7);
}

// ---- one refinement residual, split around its mat-vecs (iterative_refinement.jl:8-12,38-41 + residual.jl:53-101) ------------------------------
// residual_error = residual - H step needs [gx; hx] step_x (rows y, z), [gx; hx]'(step_y; step_z) and Lxx step_x (rows x).  zsx = [gx; hx] step_x
// is kept up to date by k_recover (step += correction  =>  zsx += [gx; hx] dx, which the condensed solve has just computed as t2), so the rows
// r, s, y, z, t of the residual need no mat-vec at all — and with them the whole condensed right-hand side b_m of the NEXT solve and its first
// operand t1 = Omega b_m.  k_refine_local does that; then ONE pass over [gx; hx] yields both transposed products (gemv_t2), one pass over Lxx
// the Hessian product, and k_refine_x finishes the x rows, the norm and the solve operand xbuf = b_x + [gx; hx]' Omega b_m.
// A round thus reads [gx; hx] twice (here and for t2 = [gx; hx] dx in the solve) instead of three times.
// One work item per CONSTRAINT (equality row, nonnegative entry, second-order cone): the item forms the rows of residual_error that belong to it
// (r, y for an equality; s, z, t for a cone) and, from them, its entries of the condensed right-hand side and of t1 — nothing crosses items, so the
// kernel runs on as many workgroups as the constraints fill; every workgroup leaves its part of the infinity norm in part[blockIdx.x] (k_refine_x
// combines them: a maximum, exact in any order).
constexpr int RL_THREADS = 256;
__global__ __launch_bounds__(RL_THREADS) void k_refine_local(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w, const double* __restrict__ v,
                                                              const double* __restrict__ res, const double* __restrict__ zsx, const double* __restrict__ wz,
                                                              const double* __restrict__ Wsoc, double* __restrict__ e, double* __restrict__ rsym,
                                                              double* __restrict__ t1, double* __restrict__ part) {
    __shared__ double sm[RL_THREADS / 64];
    inst_shift(bt.b, w, v, res, zsx, wz, Wsoc, e, rsym, t1, part);
    const Scalars sc = bt.scal(blockIdx.z);
    const double* sl = w + d.os(); const double* t = w + d.ot();
    const double* vs = v + d.os(); const double* vt = v + d.ot();
    const double Hrr = sc.rho + sc.ep, Hss = 0.0 + sc.ep;
    double m = 0.0;
    const int ee = blockIdx.x * RL_THREADS + threadIdx.x;
    if (ee < d.ne) {
        const int ir = d.orr() + ee, iy = d.oy() + ee;
        const double hr = (sc.rho + sc.ep) * v[ir] - v[d.oy() + ir - d.orr()];
        const double er = res[ir] - hr;
        const double hy = zsx[iy - d.oy()] + (-v[d.orr() + iy - d.oy()] + (0.0 - sc.ed) * v[iy]);
        const double ey = res[iy] - hy;
        e[ir] = er; e[iy] = ey;
        m = fmax(rabs(er), rabs(ey));
        double b = ey;
        b += er / Hrr;
        rsym[d.nx + ee] = b;
        const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
        t1[ee] = omega_y * b;
    } else if (ee < d.ne + d.q) {
        const int k = ee - d.ne;
        const int is = d.os() + k, iz = d.oz() + k, it = d.ot() + k;
        const double hs = (0.0 + sc.ep) * v[is] - v[d.oz() + k] - v[d.ot() + k];
        const double es = res[is] - hs;
        const double hz = zsx[d.ne + iz - d.oz()] + (-v[d.os() + iz - d.oz()] + (0.0 - sc.ed) * v[iz]);
        const double ez = res[iz] - hz;
        const double ht = t[k] * vs[k] + (sl[k] - sc.ed) * vt[k];
        const double et = res[it] - ht;
        e[is] = es; e[iz] = ez; e[it] = et;
        m = fmax(fmax(rabs(es), rabs(ez)), rabs(et));
        const double Sb = w[d.os() + k] - sc.ed, Ti = w[d.ot() + k], Pi = Hss;
        double b = ez;
        b += (et + Sb * es) / (Ti + Sb * Pi);
        rsym[d.nx + d.ne + k] = b;
        t1[d.ne + k] = wz[k] * b;
    } else if (ee < d.ne + d.q + d.n_soc) {
        const int j = ee - d.ne - d.q;
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        if (dim <= 4) {
            // small cones: one round of independent loads into registers, loops unrolled to constant indices, operations and order of the general branch
            constexpr int MD = 4;
            double lsl[MD], lt[MD], lvs[MD], lvt[MD], lvz[MD], lres_s[MD], lres_z[MD], lres_t[MD], lzsx[MD], W[MD * MD];
            const int woff = cd.soc_woff[j];
#pragma unroll
            for (int a = 0; a < MD; ++a) {
                const bool in = a < dim;
                const int k = st + a;
                lsl[a] = in ? sl[k] : 0.0; lt[a] = in ? t[k] : 0.0; lvs[a] = in ? vs[k] : 0.0; lvt[a] = in ? vt[k] : 0.0;
                lvz[a] = in ? v[d.oz() + k] : 0.0;
                lres_s[a] = in ? res[d.os() + k] : 0.0; lres_z[a] = in ? res[d.oz() + k] : 0.0; lres_t[a] = in ? res[d.ot() + k] : 0.0;
                lzsx[a] = in ? zsx[d.ne + k] : 0.0;
            }
#pragma unroll
            for (int e2 = 0; e2 < MD * MD; ++e2) W[e2] = 0.0;
#pragma unroll
            for (int c = 0; c < MD; ++c)
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim && c < dim) W[a + c * MD] = Wsoc[woff + a + c * dim];
            double rs[MD], rt[MD], rz[MD];
#pragma unroll
            for (int a = 0; a < MD; ++a) { rs[a] = 0.0; rt[a] = 0.0; rz[a] = 0.0; }
#pragma unroll
            for (int a = 0; a < MD; ++a) if (a < dim) {
                const int k = st + a;
                const double hs = (0.0 + sc.ep) * lvs[a] - lvz[a] - lvt[a];
                rs[a] = lres_s[a] - hs;
                const double hz = lzsx[a] + (-lvs[a] + (0.0 - sc.ed) * lvz[a]);
                rz[a] = lres_z[a] - hz;
                double ht;
                if (a == 0) {
                    ht = lt[0] * lvs[0] + (lsl[0] - sc.ed) * lvt[0];
#pragma unroll
                    for (int q = 1; q < MD; ++q) if (q < dim) ht += lt[q] * lvs[q] + lsl[q] * lvt[q];
                } else {
                    ht = lt[a] * lvs[0] + lsl[a] * lvt[0];
                    ht += lt[0] * lvs[a] + (lsl[0] - sc.ed) * lvt[a];
                }
                rt[a] = lres_t[a] - ht;
                e[d.os() + k] = rs[a]; e[d.oz() + k] = rz[a]; e[d.ot() + k] = rt[a];
                m = fmax(m, fmax(fmax(rabs(rs[a]), rabs(rz[a])), rabs(rt[a])));
            }
            double u[MD], vv[MD], o[MD];
#pragma unroll
            for (int a = 0; a < MD; ++a) { u[a] = 0.0; vv[a] = 0.0; o[a] = 0.0; }
            const double sb1 = lsl[0] - sc.ed;
            u[0] = lt[0] + sb1 * Hss;
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) u[k] = lt[k] + lsl[k] * Hss;
            double acc = sb1 * rs[0];
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) acc += lsl[k] * rs[k];
            vv[0] = acc + rt[0];
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) vv[k] = (lsl[k] * rs[0] + sb1 * rs[k]) + rt[k];
            arrow_inverse_small<MD>(dim, u, vv, o);
#pragma unroll
            for (int k = 0; k < MD; ++k) if (k < dim) { o[k] = rz[k] + o[k]; rsym[d.nx + d.ne + st + k] = o[k]; }
#pragma unroll
            for (int a = 0; a < MD; ++a) if (a < dim) {
                double ss = 0.0;
#pragma unroll
                for (int b2 = 0; b2 < MD; ++b2) if (b2 < dim) ss += W[a + b2 * MD] * o[b2];
                t1[d.ne + st + a] = ss;
            }
        }   // (dimension > 4: k_refine_local_wide, soc_wide.hip)
    }
    const double mr = block_max(m, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = mr;
}

// x rows: residual_error_x = residual_x - ((Lxx step_x + [gx; hx]' step_yz) + ep step_x); dscal[7] = ||residual_error||_inf (with the partial norm of
// k_refine_local); condensed b_x = residual_error_x; xbuf = [b_x + [gx; hx]' Omega b_m; 0]
__global__ __launch_bounds__(RT) void k_refine_x(BatchSc bt, Dims d, int have_m, const double* __restrict__ v, const double* __restrict__ res,
                                                  const double* __restrict__ lxv, const double* __restrict__ w1, const double* __restrict__ w2,
                                                  double* __restrict__ e, double* __restrict__ rsym, double* __restrict__ xbuf, double* __restrict__ dscal, const double* __restrict__ part, int nparts,
                                                  double* __restrict__ hpub, unsigned long long* __restrict__ hseq, unsigned long long seq) {
    __shared__ double sm[RT / 64];
    inst_shift(bt.b, v, res, lxv, w1, w2, e, rsym, xbuf, dscal, part);
    const Scalars sc = bt.scal(blockIdx.z);
    double m = 0.0;
    for (int i = threadIdx.x; i < nparts; i += RT) m = fmax(m, part[i]);     // the other rows' part of the norm (k_refine_local)
    for (int i0 = threadIdx.x; i0 < d.NP; i0 += 4 * RT) {          // four of a thread's rows in flight together, combined in the same order
        double lv[4], a1[4], a2[4], vv[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * RT; const bool in = i < d.nx;
            lv[u] = in ? lxv[i] : 0.0; a1[u] = (in && have_m) ? w1[i] : 0.0; a2[u] = (in && have_m) ? w2[i] : 0.0; vv[u] = in ? v[i] : 0.0; rv[u] = in ? res[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * RT;
            if (i < d.nx) {
                const double hv = (lv[u] + (have_m ? a1[u] : 0.0)) + sc.ep * vv[u];
                const double r = rv[u] - hv;
                e[i] = r;
                rsym[i] = r;
                xbuf[i] = have_m ? r + a2[u] : r;
                m = fmax(m, rabs(r));
            } else if (i < d.NP) xbuf[i] = 0.0;
        }
    }
    const double mr = block_max(m, sm);
    if (threadIdx.x == 0) {
        dscal[7] = mr;
        if (hpub) {                        // a single handle: the norm goes straight to mapped host memory, then the sequence number the host spins on
            hpub[7] = mr;
            __threadfence_system();
            __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- tail of a condensed solve in ONE launch: t2 = [gx; hx] dx, back-substitution + recovery (k_recover), and the local rows of the refinement residual that
// follows (k_refine_local) -------------------------------------------------------------------------------------------------------------------------------
// Every refinement round used to run  k_gemv_n_partial -> k_gemv_n_reduce -> k_recover -> k_refine_local: four dependent launches of 4 - 12 us in which only the first
// moves data worth mentioning.  Here a workgroup owns up to 16 consecutive rows of [gx; hx] that are WHOLE constraints (the table `grp`, built on the host: equality
// rows and nonnegative entries one by one, second-order cones undivided — handles whose cones all have dimension <= 4), forms their entries of t2 completely (16 rows
// x 32 column parts, 32 columns per thread and pass, all loads of a pass in flight; partial sums combined in a fixed order through LDS), and the thread that owns a
// constraint goes straight on with what k_recover and k_refine_local do for it — same operations in the same order, the new step entries handed over in registers.
// The x entries of the step are spread over the workgroups.  Nothing crosses constraints, so no workgroup waits for another.
struct TailArgs {
    const double* w; const double* res; const double* resid; const double* wz; const double* Wsoc;
    double* rsym; double* dsym; double* step; double* accum; double* zsx; double* e; double* t1;
    int zsx_mode, do_refine;
};
// What the owner thread of a constraint reads that does NOT depend on t2 — requested BEFORE the mat-vec's stream of loads, so that it is there when t2 is: left where
// it is used, every read sat behind a store of the constraint code that may alias it (the arrays are not restrict: res and e ARE one array in a correction solve), a chain
// of four to five dependent memory round trips per row that starts only when t2 is complete (bench/tail_probe.sh: 6 us of the launch's 18.4).  Every location is read
// before the same thread writes it and no thread touches another constraint's rows, so reading early changes no value.  Second-order cones: the three index look-ups
// (entry -> cone -> start / dimension / offset) travel here; their operands follow in one batch behind t2 (64 doubles would have to be held across the mat-vec).
struct TailPre { double v[13]; int j, st, dim, woff; };
__device__ __forceinline__ void tail_prefetch(const Dims& d, const ConeDev& cd, const TailArgs& A, int row, TailPre& P) {
    if (row < d.ne) {
        const int k = row, ir = d.orr() + k, iy = d.oy() + k;
        P.v[0] = A.rsym[d.nx + k];
        if (A.zsx_mode != 1) P.v[1] = A.zsx[k];
        P.v[2] = A.res[ir];
        if (A.accum) { P.v[3] = A.accum[iy]; P.v[4] = A.accum[ir]; }
        if (A.do_refine) { P.v[5] = A.resid[ir]; P.v[6] = A.resid[iy]; }
    } else if (row < d.ne + d.q) {
        const int k = row - d.ne, is = d.os() + k, iz = d.oz() + k, it = d.ot() + k;
        P.v[0] = A.wz[k]; P.v[1] = A.rsym[d.nx + d.ne + k];
        if (A.zsx_mode != 1) P.v[2] = A.zsx[d.ne + k];
        P.v[3] = A.w[is]; P.v[4] = A.w[it]; P.v[5] = A.res[it]; P.v[6] = A.res[is];
        if (A.accum) { P.v[7] = A.accum[iz]; P.v[8] = A.accum[is]; P.v[9] = A.accum[it]; }
        if (A.do_refine) { P.v[10] = A.resid[is]; P.v[11] = A.resid[iz]; P.v[12] = A.resid[it]; }
    } else {
        const int c = row - d.ne;
        P.j = cd.entry_soc[c];
        P.st = cd.soc_start[P.j]; P.dim = cd.soc_dim[P.j]; P.woff = cd.soc_woff[P.j];
    }
}
__device__ __forceinline__ double tail_equality(const Dims& d, const Scalars& sc, const TailArgs& A, const TailPre& P, int k, double t2k) {
    const double Hrr = sc.rho + sc.ep;
    const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
    const double dy = -1.0 * omega_y * (P.v[0] - t2k);
    A.dsym[d.nx + k] = dy;
    double zk = t2k;
    if (A.zsx_mode == 2) zk = P.v[1] + t2k;
    if (A.zsx_mode) A.zsx[k] = zk; else zk = P.v[1];
    const double dr = (P.v[2] + dy) / Hrr;
    A.step[d.oy() + k] = dy;
    A.step[d.orr() + k] = dr;
    double vy = dy, vr = dr;
    if (A.accum) { vy = P.v[3] + dy; vr = P.v[4] + dr; A.accum[d.oy() + k] = vy; A.accum[d.orr() + k] = vr; }
    if (!A.do_refine) return 0.0;
    // k_refine_local, equality item
    const int ir = d.orr() + k, iy = d.oy() + k;
    const double hr = (sc.rho + sc.ep) * vr - vy;
    const double er = P.v[5] - hr;
    const double hy = zk + (-vr + (0.0 - sc.ed) * vy);
    const double ey = P.v[6] - hy;
    A.e[ir] = er; A.e[iy] = ey;
    double b = ey;
    b += er / Hrr;
    A.rsym[d.nx + k] = b;
    A.t1[k] = omega_y * b;
    return fmax(rabs(er), rabs(ey));
}
__device__ __forceinline__ double tail_nonnegative(const Dims& d, const Scalars& sc, const TailArgs& A, const TailPre& P, int k, double t2k) {
    const double Hss = 0.0 + sc.ep;
    const double dz = -1.0 * P.v[0] * (P.v[1] - t2k);
    A.dsym[d.nx + d.ne + k] = dz;
    double zk = t2k;
    if (A.zsx_mode == 2) zk = P.v[2] + t2k;
    if (A.zsx_mode) A.zsx[d.ne + k] = zk; else zk = P.v[2];
    const double Sb = P.v[3] - sc.ed, Ti = P.v[4], Pi = Hss;
    const double rt = P.v[5], rs = P.v[6];
    const double ds = (rt + Sb * (rs + dz)) / (Ti + Sb * Pi);
    const double dt = (rt - Ti * ds) / Sb;
    A.step[d.oz() + k] = dz; A.step[d.os() + k] = ds; A.step[d.ot() + k] = dt;
    double vz = dz, vs = ds, vt = dt;
    if (A.accum) {
        vz = P.v[7] + dz; vs = P.v[8] + ds; vt = P.v[9] + dt;
        A.accum[d.oz() + k] = vz; A.accum[d.os() + k] = vs; A.accum[d.ot() + k] = vt;
    }
    if (!A.do_refine) return 0.0;
    const int is = d.os() + k, iz = d.oz() + k, it = d.ot() + k;
    const double hs = (0.0 + sc.ep) * vs - vz - vt;
    const double es = P.v[10] - hs;
    const double hz = zk + (-vs + (0.0 - sc.ed) * vz);
    const double ez = P.v[11] - hz;
    const double ht = Ti * vs + (P.v[3] - sc.ed) * vt;
    const double et = P.v[12] - ht;
    A.e[is] = es; A.e[iz] = ez; A.e[it] = et;
    double b = ez;
    b += (et + Sb * es) / (Ti + Sb * Pi);
    A.rsym[d.nx + d.ne + k] = b;
    A.t1[d.ne + k] = P.v[0] * b;
    return fmax(fmax(rabs(es), rabs(ez)), rabs(et));
}
// second-order cone j of dimension <= 4; t2c[a] = entry of t2 of its row a
__device__ __forceinline__ double tail_soc_small(const Dims& d, const Scalars& sc, const TailArgs& A, const TailPre& P, const double* t2c) {
    constexpr int MD = 4;
    const double Hss = 0.0 + sc.ep;
    const int st = P.st, dim = P.dim, woff = P.woff;
    double sl[MD], t[MD], rs[MD], rt[MD], bb[MD], tt[MD], zz[MD], W[MD * MD], az[MD], as[MD], at[MD], res_s[MD], res_z[MD], res_t[MD];
#pragma unroll
    for (int a = 0; a < MD; ++a) {
        const bool in = a < dim;
        sl[a] = in ? A.w[d.os() + st + a] : 0.0; t[a] = in ? A.w[d.ot() + st + a] : 0.0;
        rs[a] = in ? A.res[d.os() + st + a] : 0.0; rt[a] = in ? A.res[d.ot() + st + a] : 0.0;
        bb[a] = in ? A.rsym[d.nx + d.ne + st + a] : 0.0; tt[a] = in ? t2c[a] : 0.0;
        zz[a] = (in && A.zsx_mode != 1) ? A.zsx[d.ne + st + a] : 0.0;
        az[a] = (in && A.accum) ? A.accum[d.oz() + st + a] : 0.0; as[a] = (in && A.accum) ? A.accum[d.os() + st + a] : 0.0; at[a] = (in && A.accum) ? A.accum[d.ot() + st + a] : 0.0;
        res_s[a] = (in && A.do_refine) ? A.resid[d.os() + st + a] : 0.0; res_z[a] = (in && A.do_refine) ? A.resid[d.oz() + st + a] : 0.0;
        res_t[a] = (in && A.do_refine) ? A.resid[d.ot() + st + a] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < MD * MD; ++e) W[e] = 0.0;
#pragma unroll
    for (int c = 0; c < MD; ++c)
#pragma unroll
        for (int a = 0; a < MD; ++a) if (a < dim && c < dim) W[a + c * MD] = A.Wsoc[woff + a + c * dim];
    double u[MD], v[MD], ds[MD], o[MD], dz[MD], zk[MD];
#pragma unroll
    for (int a = 0; a < MD; ++a) { o[a] = bb[a] - tt[a]; u[a] = 0.0; v[a] = 0.0; ds[a] = 0.0; dz[a] = 0.0; zk[a] = A.zsx_mode == 1 ? tt[a] : (A.zsx_mode == 2 ? zz[a] + tt[a] : zz[a]); }
    if (A.zsx_mode) {
#pragma unroll
        for (int a = 0; a < MD; ++a) if (a < dim) A.zsx[d.ne + st + a] = zk[a];
    }
#pragma unroll
    for (int a = 0; a < MD; ++a) if (a < dim) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < MD; ++c) if (c < dim) s += W[a + c * MD] * o[c];
        dz[a] = -1.0 * s;
        A.dsym[d.nx + d.ne + st + a] = dz[a];
    }
    const double sb1 = sl[0] - sc.ed;
    u[0] = t[0] + sb1 * Hss;
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) u[k] = t[k] + sl[k] * Hss;
    double acc = sb1 * (rs[0] + dz[0]);
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) acc += sl[k] * (rs[k] + dz[k]);
    v[0] = rt[0] + acc;
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) v[k] = rt[k] + (sl[k] * (rs[0] + dz[0]) + sb1 * (rs[k] + dz[k]));
    arrow_inverse_small<MD>(dim, u, v, ds);
    acc = t[0] * ds[0];
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) acc += t[k] * ds[k];
    v[0] = rt[0] - acc;
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) v[k] = rt[k] - (t[k] * ds[0] + t[0] * ds[k]);
    u[0] = sb1;
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) u[k] = sl[k];
    arrow_inverse_small<MD>(dim, u, v, o);                       // o = dt
    double lvz[MD], lvs[MD], lvt[MD];
#pragma unroll
    for (int k = 0; k < MD; ++k) {
        lvz[k] = dz[k]; lvs[k] = ds[k]; lvt[k] = o[k];
        if (k < dim) {
            A.step[d.oz() + st + k] = dz[k]; A.step[d.os() + st + k] = ds[k]; A.step[d.ot() + st + k] = o[k];
            if (A.accum) {
                lvz[k] = az[k] + dz[k]; lvs[k] = as[k] + ds[k]; lvt[k] = at[k] + o[k];
                A.accum[d.oz() + st + k] = lvz[k]; A.accum[d.os() + st + k] = lvs[k]; A.accum[d.ot() + st + k] = lvt[k];
            }
        }
    }
    if (!A.do_refine) return 0.0;
    // k_refine_local, small cone
    double m = 0.0;
    double qs[MD], qt[MD], qz[MD];
#pragma unroll
    for (int a = 0; a < MD; ++a) { qs[a] = 0.0; qt[a] = 0.0; qz[a] = 0.0; }
#pragma unroll
    for (int a = 0; a < MD; ++a) if (a < dim) {
        const int k = st + a;
        const double hs = (0.0 + sc.ep) * lvs[a] - lvz[a] - lvt[a];
        qs[a] = res_s[a] - hs;
        const double hz = zk[a] + (-lvs[a] + (0.0 - sc.ed) * lvz[a]);
        qz[a] = res_z[a] - hz;
        double ht;
        if (a == 0) {
            ht = t[0] * lvs[0] + (sl[0] - sc.ed) * lvt[0];
#pragma unroll
            for (int q = 1; q < MD; ++q) if (q < dim) ht += t[q] * lvs[q] + sl[q] * lvt[q];
        } else {
            ht = t[a] * lvs[0] + sl[a] * lvt[0];
            ht += t[0] * lvs[a] + (sl[0] - sc.ed) * lvt[a];
        }
        qt[a] = res_t[a] - ht;
        A.e[d.os() + k] = qs[a]; A.e[d.oz() + k] = qz[a]; A.e[d.ot() + k] = qt[a];
        m = fmax(m, fmax(fmax(rabs(qs[a]), rabs(qz[a])), rabs(qt[a])));
    }
    double uu[MD], vv[MD], oo[MD];
#pragma unroll
    for (int a = 0; a < MD; ++a) { uu[a] = 0.0; vv[a] = 0.0; oo[a] = 0.0; }
    uu[0] = t[0] + sb1 * Hss;
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) uu[k] = t[k] + sl[k] * Hss;
    double ac2 = sb1 * qs[0];
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) ac2 += sl[k] * qs[k];
    vv[0] = ac2 + qt[0];
#pragma unroll
    for (int k = 1; k < MD; ++k) if (k < dim) vv[k] = (sl[k] * qs[0] + sb1 * qs[k]) + qt[k];
    arrow_inverse_small<MD>(dim, uu, vv, oo);
#pragma unroll
    for (int k = 0; k < MD; ++k) if (k < dim) { oo[k] = qz[k] + oo[k]; A.rsym[d.nx + d.ne + st + k] = oo[k]; }
#pragma unroll
    for (int a = 0; a < MD; ++a) if (a < dim) {
        double ss = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < MD; ++b2) if (b2 < dim) ss += W[a + b2 * MD] * oo[b2];
        A.t1[d.ne + st + a] = ss;
    }
    return m;
}

#ifndef TAIL_EXP
#define TAIL_EXP 0       // (bench/tail_probe.sh: 1 = no constraint code, 2 = no constraint code and two workgroups per row group, 3 = no mat-vec loads)
#endif
constexpr int TAIL_ROWS = 16, TAIL_PARTS = 32, TAIL_CPT = 16;
// PT = column parts that are THREADS (32: 512 threads, one part each; 16: 256 threads, two parts each).  The 32 parts, their columns (part p: p, p + 32, ...) and the order
// of every sum are the same for both: the bits do not depend on PT.  One system alone takes PT = 32 (its ~160 workgroups do not fill the chip: the more loads each has in
// flight the better); a group's launch takes PT = 16: with ~200 registers per thread a compute unit holds ONE 512-thread workgroup but TWO of 256, and a workgroup streams
// only between its staging of dx and the constraint code of its 16 owner threads — 3.7 TB/s with one resident workgroup per unit (profiles/r06_kernel_stats_group.csv).
template <int PT>
__global__ __launch_bounds__(TAIL_ROWS * PT) void k_solve_tail(BatchSc bt, Dims d, ConeDev cd, const int* __restrict__ grp, int ngrp, const int* rowrange, const double* __restrict__ Z,
                                                                        const double* __restrict__ dx, const double* w, const double* res, const double* resid, const double* wz,
                                                                        const double* Wsoc, double* rsym, double* dsym, double* step, double* accum, double* zsx, double* e, double* t1,
                                                                        double* __restrict__ part, int zsx_mode, int do_refine, const int* __restrict__ gate = nullptr, int gate_epoch = 0) {
    if (gate && gate[0] == gate_epoch) return;        // (internal.hpp: gate)
    constexpr int ROWS = TAIL_ROWS, PARTS = TAIL_PARTS, CPT = TAIL_CPT, W = PARTS * CPT, NS = PARTS / PT, HB = CPT / NS, NT = ROWS * PT;
    static_assert(PT == 32 || PT == 16, "one or two column parts per thread");
    extern __shared__ __attribute__((aligned(16))) double xs[];      // dx, whole (nx rounded up to a multiple of W doubles, zero padded): ONE barrier for the mat-vec
    __shared__ double psum[PARTS][ROWS];
    __shared__ double t2s[ROWS];
    __shared__ double sm[ROWS * PARTS / 64];
    inst_shift(bt.b, Z, dx, w, res, resid, wz, Wsoc, rsym, dsym, step, zsx, e, t1, part);
    if (accum) inst_shift(bt.b, accum);
    if (rowrange) inst_shift_i(bt.b, rowrange);
    const Scalars sc = bt.scal(blockIdx.z);
    const int tid = threadIdx.x, r = tid % ROWS, p = tid / ROWS;
    // workgroup b runs on XCD b % 8 (dispatch order; used for speed only): every XCD takes a CONTIGUOUS eighth of the row groups — the 128-byte runs of 16 rows down a
    // column are not aligned to the cache lines (the leading dimension m is not a multiple of 16), so neighbouring row groups share the lines at their common edge, and
    // share them through an L2 only when they run on the same XCD
#if TAIL_EXP == 2
    const int per = (ngrp + 7) / 8, g = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 4), half = ((int)blockIdx.x >> 3) & 1;
#else
    const int per = (ngrp + 7) / 8, g = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
#endif
    if (g >= ngrp) return;
    const int r0 = grp[g], nrows = grp[g + 1] - r0;
    // ---- t2 rows r0 .. r0 + nrows - 1 ------------------------------------------------------------------------------
    const bool live = r < nrows;
    const double* Zr = Z + r0 + r;
    // an analysed structure (structure.hip): the columns of the row that can be non-zero — the loads outside are predicated off, the sums are those of the dense rows
    int jlo = 0, jhi = d.nx;
    if (rowrange && live) { jlo = rowrange[2 * (r0 + r)]; jhi = rowrange[2 * (r0 + r) + 1]; }
#if TAIL_EXP == 2
    { const int W2 = PARTS * CPT, np2 = (d.nx + W2 - 1) / W2, cut = ((np2 + 1) / 2) * W2; if (half == 0) jhi = cut; else jlo = cut; }
#endif
#if TAIL_EXP == 3
    jhi = 0;
#endif
    const int npass = (d.nx + W - 1) / W;
    const TailArgs A{w, res, resid, wz, Wsoc, rsym, dsym, step, accum, zsx, e, t1, zsx_mode, do_refine};
    TailPre P;
#if TAIL_EXP == 0
    if (tid < nrows) tail_prefetch(d, cd, A, r0 + tid, P);
#endif
    // the first batch's loads go out before dx is staged (they do not depend on it); from then on batch k + 1 travels while batch k is summed: the row is a stream
    // of loads with two batches in flight, not a chain of round trips.  A batch = HB consecutive entries (columns c0 + 32 q) of each of the thread's NS parts:
    // CPT loads per thread whatever PT; part p sums its entries q = 0, 1, ... of pass 0, then of pass 1, ... — one accumulator per part
    double va[NS][HB], vb[NS][HB];
    auto fetch = [&](int b, double (&v)[NS][HB]) {       // batch b: pass b / NS, entries (b % NS) HB .. + HB - 1
        const int c0 = (b / NS) * W, q0 = (b % NS) * HB;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
#pragma unroll
            for (int q = 0; q < HB; ++q) { const int c = c0 + (p + PT * sl) + PARTS * (q0 + q); v[sl][q] = (live && c < d.nx && c >= jlo && c < jhi) ? Zr[(size_t)c * d.m] : 0.0; }
    };
    const int nb = npass * NS;
    fetch(0, va);
    for (int i = tid; i < npass * W; i += NT) xs[i] = i < d.nx ? dx[i] : 0.0;
    __syncthreads();
    double acc[NS];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) acc[sl] = 0.0;
    auto sum = [&](int b, const double (&v)[NS][HB]) {
        const int c0 = (b / NS) * W, q0 = (b % NS) * HB;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl)
#pragma unroll
            for (int q = 0; q < HB; ++q) acc[sl] += v[sl][q] * xs[c0 + (p + PT * sl) + PARTS * (q0 + q)];
    };
    for (int b = 0; b < nb; b += 2) {
        if (b + 1 < nb) fetch(b + 1, vb);
        sum(b, va);
        if (b + 1 < nb) {
            if (b + 2 < nb) fetch(b + 2, va);
            sum(b + 1, vb);
        }
    }
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) psum[p + PT * sl][r] = acc[sl];
    // ---- x entries of the step (this workgroup's share) ---------------------------------------------------------------
    {
        const int xsz = (d.nx + ngrp - 1) / ngrp, xend = min(d.nx, (g + 1) * xsz);
        for (int i = g * xsz + tid; i < xend; i += NT) {
            const double v = dx[i];
            dsym[i] = v; step[i] = v;
            if (accum) accum[i] += v;
        }
    }
    __syncthreads();
    if (tid < ROWS) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += psum[q][tid];
        t2s[tid] = s;
    }
    __syncthreads();
#if TAIL_EXP == 1 || TAIL_EXP == 2
    if (tid < nrows) t1[r0 + tid] = t2s[tid];
    return;
#endif
    // ---- the constraints of these rows --------------------------------------------------------------------------------
    double m = 0.0;
    if (tid < nrows) {
#if TAIL_EXP != 0
        tail_prefetch(d, cd, A, r0 + tid, P);
#endif
        const int row = r0 + tid;
        if (row < d.ne) m = tail_equality(d, sc, A, P, row, t2s[tid]);
        else if (row < d.ne + d.q) m = tail_nonnegative(d, sc, A, P, row - d.ne, t2s[tid]);
        else if (row - d.ne == P.st) m = tail_soc_small(d, sc, A, P, t2s + tid);      // (the cone's rows follow its first one inside the group)
    }
    if (do_refine) {
        const double mr = block_max(m, sm);
        if (tid == 0) part[g] = mr;
    }
}
// can the solves of this handle (or of the group launch in progress) end in k_solve_tail?
static bool solve_tail_ok(const calipso_hip_solver* s) {
    static const bool env = [] { const char* e = getenv("CALIPSO_HIP_SOLVE_TAIL"); return !e || atoi(e) != 0; }();
    return env && s->zgrp && s->d.m > 0 && !s->compact && !(s->blocks.on && s->blocks_effective);
}
bool solve_tail_available(const calipso_hip_solver* s) { return solve_tail_ok(s); }
bool launch_solve_tail(calipso_hip_solver* s, int which, bool accumulate, bool with_refine) {
    if (!solve_tail_ok(s)) return false;
    const BatchSc B = batch_of(s);
    const double* res = which == 0 ? s->residual : s->residual_error;
    double* st = which == 0 ? s->step : s->step_correction;
    constexpr int TW = TAIL_PARTS * TAIL_CPT;
    const size_t lds = sizeof(double) * (size_t)((s->d.nx + TW - 1) / TW) * TW;
    // (bench/ab_libs.sh: which form a launch takes; CALIPSO_HIP_TAIL_PT=32 / 16 forces one — the bits are the same)
    static const int env_pt = [] { const char* e = getenv("CALIPSO_HIP_TAIL_PT"); return e ? atoi(e) : 0; }();
    const bool narrow = env_pt ? env_pt == 16 : B.b.n >= 4;
    const void* fn = narrow ? (const void*)k_solve_tail<16> : (const void*)k_solve_tail<32>;
    if (lds > 48 * 1024) {
        if (lds > 96 * 1024 || !lds_attribute(fn, 96 * 1024)) return false;      // (> 64 KB of dynamic LDS must be asked for, on every device; refused: the separate kernels)
    }
    const dim3 grid((s->n_zgrp + 7) / 8 * 8 * (TAIL_EXP == 2 ? 2 : 1), 1, B.b.n);
    auto go = [&](auto kern, int threads) {
        hipLaunchKernelGGL(kern, grid, dim3(threads), lds, s->stream, B, s->d, s->cone, s->zgrp, s->n_zgrp, s->band64 > 0 ? s->zrow : (const int*)nullptr, s->Z, s->xbuf, s->solution, res,
                           s->residual, s->wz, s->Wsoc, s->residual_symmetric, s->step_symmetric, st, accumulate ? s->step : (double*)nullptr, s->zsx, s->residual_error, s->t1, s->refpart,
                           which == 0 ? 1 : 2, with_refine ? 1 : 0, s->gate_epoch ? s->gate : (const int*)nullptr, s->gate_epoch);
    };
    if (narrow) go(k_solve_tail<16>, TAIL_ROWS * 16); else go(k_solve_tail<32>, TAIL_ROWS * 32);
    s->refine_local_done = with_refine;
    if (with_refine) s->refparts = s->n_zgrp;
    return true;
}
// the row groups of k_solve_tail: whole constraints, at most 16 rows each (host, at create)
void solve_tail_plan(const Dims& d, const std::vector<int>& soc_start, const std::vector<int>& soc_dim, std::vector<int>& grp) {
    grp.clear();
    if (d.m == 0 || d.n_wide > 0) return;
    grp.push_back(0);
    int fill = 0, row = 0;
    auto item = [&](int size) {
        if (fill + size > TAIL_ROWS) { grp.push_back(row); fill = 0; }
        fill += size; row += size;
    };
    for (int k = 0; k < d.ne + d.q; ++k) item(1);
    for (size_t j = 0; j < soc_dim.size(); ++j) item(soc_dim[j]);
    grp.push_back(row);
    (void)soc_start;
}

void launch_refine_local(calipso_hip_solver* s) {
    if (s->refine_local_done) { s->refine_local_done = false; return; }       // k_solve_tail formed these rows with the step it had just recovered
    const BatchSc B = batch_of(s);
    const int items = s->d.ne + s->d.q + s->d.n_soc;
    s->refparts = items ? (items + RL_THREADS - 1) / RL_THREADS + s->d.n_wide : 0;
    if (items == 0) return;
    hipLaunchKernelGGL(k_refine_local, dim3((items + RL_THREADS - 1) / RL_THREADS, 1, B.b.n), dim3(RL_THREADS), 0, s->stream, B, s->d, s->cone, s->solution, s->step,
                       s->residual, s->zsx, s->wz, s->Wsoc, s->residual_error, s->residual_symmetric, s->t1, s->refpart);
    launch_refine_local_wide(s, (items + RL_THREADS - 1) / RL_THREADS);      // their partial norms follow the ones of the kernel above in refpart
}
// k_refine_x with the reduction of the Hessian product folded in: the column-chunk partial sums of Lxx step_x (k_gemv_t2_and_n leaves nchunk of them per row) are
// combined HERE, in the order k_gemv_n_reduce combines them (four lanes per row, every fourth chunk each, then (0 + 1) + (2 + 3): the same bits), 64 rows per
// workgroup; the norm is a maximum over workgroups — exact in any order — through an atomic maximum on the bit pattern of the (non-negative) doubles, and the
// workgroup that arrives last (a ticket counter) adds the other rows' parts, stores dscal[7] and publishes.  One launch less per refinement residual.
__global__ __launch_bounds__(256) void k_refine_x_fused(BatchSc bt, Dims d, int have_m, int nchunk, const double* __restrict__ partial, const double* __restrict__ v,
                                                         const double* __restrict__ res, const double* __restrict__ w1, const double* __restrict__ w2, double* __restrict__ e,
                                                         double* __restrict__ rsym, double* __restrict__ xbuf, double* __restrict__ dscal, const double* __restrict__ part, int nparts,
                                                         double* __restrict__ hpub, unsigned long long* __restrict__ hseq, unsigned long long seq, int* __restrict__ gate = nullptr,
                                                         int gate_epoch = 0, int it = -1, int min_it = 0, double tol = 0.0, int last_queued = 0) {
    __shared__ double ps[4][64];
    __shared__ double sm[4];
    __shared__ int last;
    inst_shift(bt.b, partial, v, res, w1, w2, e, rsym, xbuf, dscal, part);
    // Speculative rounds (api.hip: do_refinement): gate[0] = epoch of the refinement that has converged, gate[1] = residuals evaluated (0 = only the initial one),
    // dscal[58] = first norm, dscal[59] = last norm.  A residual queued behind the converging one does nothing — except the LAST queued one, which reports.
    if (gate && gate[0] == gate_epoch) {
        if (last_queued && blockIdx.x == 0 && threadIdx.x == 0 && hpub) {
            hpub[7] = dscal[59]; hpub[20] = dscal[58]; hpub[21] = (double)gate[1]; hpub[22] = 1.0;
            __threadfence_system();
            __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const Scalars sc = bt.scal(blockIdx.z);
    const int r = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + r;
    double acc = 0.0;
    // the row's own operands travel with the partial sums (fetched behind the barrier they were one more memory round trip at the end of a 7 us kernel)
    double w1i = 0.0, w2i = 0.0, vi0 = 0.0, resi = 0.0;
    if (p == 0 && i < d.nx) { w1i = have_m ? w1[i] : 0.0; w2i = have_m ? w2[i] : 0.0; vi0 = v[i]; resi = res[i]; }
    if (i < d.nx) {
        for (int c0 = p; c0 < nchunk; c0 += 64) {
            double pv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int c = c0 + 4 * u; pv[u] = c < nchunk ? partial[(size_t)c * d.nx + i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (c0 + 4 * u < nchunk) acc += pv[u];
        }
    }
    ps[p][r] = acc;
    __syncthreads();
    double m = 0.0;
    if (p == 0) {
        if (i < d.nx) {
            const double lv = (ps[0][r] + ps[1][r]) + (ps[2][r] + ps[3][r]);
            const double hv = (lv + w1i) + sc.ep * vi0;
            const double rr = resi - hv;
            e[i] = rr;
            rsym[i] = rr;
            xbuf[i] = have_m ? rr + w2i : rr;
            m = rabs(rr);
        } else if (i < d.NP) xbuf[i] = 0.0;
    }
    const double mr = block_max(m, sm);
    unsigned long long* nb = reinterpret_cast<unsigned long long*>(dscal + 60);
    unsigned* ticket = reinterpret_cast<unsigned*>(dscal + 61);
    if (threadIdx.x == 0) {
        atomicMax(nb, (unsigned long long)__double_as_longlong(mr));          // (non-negative doubles order like their bit patterns)
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    double mm = 0.0;
    for (int k = threadIdx.x; k < nparts; k += 256) mm = fmax(mm, part[k]);     // the other rows' part of the norm (k_refine_local / k_solve_tail)
    const double mo = block_max(mm, sm);
    if (threadIdx.x == 0) {
        const double mx = fmax(mo, __longlong_as_double((long long)__hip_atomic_load(nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
        dscal[7] = mx;
        __hip_atomic_store(nb, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // for the next residual
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int stopped = 0;
        if (gate) {
            if (it == 0) dscal[58] = mx;
            dscal[59] = mx;
            gate[1] = it;
            if (mx <= tol && it >= min_it) { gate[0] = gate_epoch; stopped = 1; }      // iterative_refinement.jl:14-16: the loop ends here
        }
        if (hpub && (!gate || last_queued)) {
            hpub[7] = mx;
            if (gate) { hpub[20] = dscal[58]; hpub[21] = (double)it; hpub[22] = (double)stopped; }
            __threadfence_system();
            __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
void launch_refine_x_fused(calipso_hip_solver* s, bool publish, int nchunk, int it, bool last_queued) {
    const BatchSc B = batch_of(s);
    const bool spec = it >= 0 && s->gate_epoch != 0;
    const bool pub = publish && (!spec || last_queued);          // speculative rounds: only the last queued residual reports to the host
    hipLaunchKernelGGL(k_refine_x_fused, dim3((s->d.NP + 63) / 64, 1, B.b.n), dim3(256), 0, s->stream, B, s->d, s->d.m > 0 ? 1 : 0, nchunk, s->gemv_partial, s->step, s->residual, s->w1, s->w2,
                       s->residual_error, s->residual_symmetric, s->xbuf, s->dscal, s->refpart, s->refparts, pub ? s->hscal_dev : (double*)nullptr,
                       pub ? s->hseq_dev : (unsigned long long*)nullptr, pub ? ++s->pub_seq : 0ULL, spec ? s->gate : (int*)nullptr, s->gate_epoch, it,
                       (int)s->opt.min_iterative_refinement, s->opt.iterative_refinement_tolerance, last_queued ? 1 : 0);
}
void launch_refine_x(calipso_hip_solver* s, bool publish) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_refine_x, dim3(1, 1, B.b.n), dim3(RT), 0, s->stream, B, s->d, s->d.m > 0 ? 1 : 0, s->step, s->residual, s->lxv, s->w1, s->w2, s->residual_error,
                       s->residual_symmetric, s->xbuf, s->dscal, s->refpart, s->refparts,
                       publish ? s->hscal_dev : (double*)nullptr, publish ? s->hseq_dev : (unsigned long long*)nullptr, publish ? ++s->pub_seq : 0ULL);
}

__global__ void k_add(Batch bt, int n, double* __restrict__ y, const double* __restrict__ x) {
    inst_shift(bt, y, x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
void launch_add(calipso_hip_solver* s, double* y, const double* x, int len) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_add, dim3((len + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B.b, len, y, x);
}

// fills / copies of per-instance buffers (one launch covers every instance of the batch)
__global__ void k_fill_d(Batch bt, double* __restrict__ p, size_t n, double v) {
    inst_shift(bt, p);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_fill_i(Batch bt, int* __restrict__ p, size_t n, int v) {
    inst_shift_i(bt, p);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_copy_d(Batch bt, double* __restrict__ dst, const double* __restrict__ src, size_t n) {
    inst_shift(bt, dst, src);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
void fill_d(calipso_hip_solver* s, double* p, size_t n, double v) {
    if (!n) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_fill_d, dim3((unsigned)((n + 255) / 256), 1, B.b.n), dim3(256), 0, s->stream, B.b, p, n, v);
}
void fill_i(calipso_hip_solver* s, int* p, size_t n, int v) {
    if (!n) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_fill_i, dim3((unsigned)((n + 255) / 256), 1, B.b.n), dim3(256), 0, s->stream, B.b, p, n, v);
}
// four copies in one launch (the save / restore of the benchmark step): blockIdx.y selects the pair
struct Copy4 { double* dst[4]; const double* src[4]; unsigned long long n[4]; };
__global__ void k_copy4_d(Batch bt, Copy4 c) {
    const int q = blockIdx.y;
    double* dst = c.dst[q]; const double* src = c.src[q];
    inst_shift(bt, dst, src);
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c.n[q]) dst[i] = src[i];
}
void copy4_d(calipso_hip_solver* s, double* const dst[4], const double* const src[4], const size_t n[4]) {
    Copy4 c; size_t nmax = 0;
    for (int q = 0; q < 4; ++q) { c.dst[q] = dst[q]; c.src[q] = src[q]; c.n[q] = n[q]; nmax = std::max(nmax, n[q]); }
    if (!nmax) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_copy4_d, dim3((unsigned)((nmax + 255) / 256), 4, B.b.n), dim3(256), 0, s->stream, B.b, c);
}
void copy_d(calipso_hip_solver* s, double* dst, const double* src, size_t n) {
    if (!n) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_copy_d, dim3((unsigned)((n + 255) / 256), 1, B.b.n), dim3(256), 0, s->stream, B.b, dst, src, n);
}

// dense condensed K (both triangles, as residual_jacobian_variables.jl:110-167 writes it) for inspection / parity.
// Needs the cone blocks from launch_cone_weights (kzz for nonnegative entries, Bsoc for second-order cones).
__global__ void k_assemble_K(Dims d, Scalars sc, ConeDev cd, const double* __restrict__ Lxx, const double* __restrict__ gx,
                             const double* __restrict__ hx, const double* __restrict__ kzz, const double* __restrict__ Bsoc,
                             double* __restrict__ K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // row (fast, coalesced writes)
    const int j = blockIdx.y;
    if (i >= d.n) return;
    const int nx = d.nx, ne = d.ne;
    double v = 0.0;
    if (i < nx && j < nx) {
        v = Lxx[i + (size_t)j * nx];
        if (i == j) v += sc.ep;
    } else if (i < nx || j < nx) {
        const int c = i < nx ? j : i;      // constraint index
        const int xk = i < nx ? i : j;     // variable index
        v = (c < nx + ne) ? gx[(c - nx) + (size_t)xk * d.m] : hx[(c - nx - ne) + (size_t)xk * d.m];   // stacked Jacobian, ld = m
    } else if (i < nx + ne || j < nx + ne) {
        if (i == j) v = -1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed);
    } else {
        const int a = i - nx - ne, b = j - nx - ne;
        const int ja = cd.entry_soc[a], jb = cd.entry_soc[b];
        if (ja < 0 || jb < 0) {
            if (a == b) v = kzz[a];
        } else if (ja == jb) {
            const int st = cd.soc_start[ja], dim = cd.soc_dim[ja];
            v = Bsoc[cd.soc_woff[ja] + (a - st) + (b - st) * dim];
        }
    }
    K[i + (size_t)j * d.n] = v;
}
void launch_assemble_K(calipso_hip_solver* s) {
    hipLaunchKernelGGL(k_assemble_K, dim3((s->d.n + 255) / 256, s->d.n), dim3(256), 0, s->stream, s->d, s->sc, s->cone, s->Lxx, s->gx,
                       s->hx, s->kzz, s->Bsoc, s->Kdense);
}

}  // namespace calipso
