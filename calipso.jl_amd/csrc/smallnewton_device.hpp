// smallnewton_device.hpp — the device side of csrc/smallnewton.hip, included once per workgroup size inside a namespace of its own (SN_THREADS = threads per
// instance = per workgroup): two wavefronts per instance let a compute unit hold four C5-sized instances, four wavefronts are faster per instance when the LDS
// footprint allows two instances anyway.  Everything here is __device__ code over the common types (Dm, Lay, Args, the enums) of smallnewton.hip.
constexpr int NT = SN_THREADS, NW = NT / 64;
static_assert(NT == 64 || NT == 128 || NT == 256, "one, two or four wavefronts per instance");

// ---- workgroup-wide reductions (every thread calls; all get the result) ----------------------------------------------------------------------
// (result in lane 63 only; data-parallel moves, not shuffles through the LDS pipeline: device_utils.hpp)
__device__ __forceinline__ double wave_sum(double v) { return calipso::wave_sum_l63(v); }
__device__ __forceinline__ double wave_max(double v) { return calipso::wave_max_l63(v); }
template <int K> __device__ __forceinline__ void block_sum(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { const double s = wave_sum(v[k]); if (lane == 63) red[k * 4 + wave] = s; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { double t = red[k * 4]; for (int w = 1; w < NW; ++w) t += red[k * 4 + w]; v[k] = t; }
}
template <int K> __device__ __forceinline__ void block_max(double (&v)[K], double* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { const double s = wave_max(v[k]); if (lane == 63) red[k * 4 + wave] = s; }
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
    for (int k = 0; k < KS; ++k) { const double s = wave_sum(sv[k]); if (lane == 63) red[k * 4 + wave] = s; }
#pragma unroll
    for (int k = 0; k < KM; ++k) { const double s = wave_max(mv[k]); if (lane == 63) red[(KS + k) * 4 + wave] = s; }
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

// part[w n + r] = sum over the columns c = w, w + NW, ... of M[r + c n] x[c]: the Hessian block of the QP stays in GLOBAL memory (n^2 doubles per instance, four products a
// step: it lives in L2; in LDS it was a third of the instance's footprint and set how many instances a compute unit holds).  Wavefront w takes every NW-th column,
// lane -> row (coalesced), MVG_U loads in flight per thread; the caller adds the NW partial sums after its next barrier (mvg_sum)
#ifndef MVG_U
#define MVG_U (NT <= 128 ? 4 : 8)      // (more loads in flight cost registers the two-wavefront build does not have: 16 -> 159 spilled, -12 % throughput)
#endif
__device__ __forceinline__ void mvg_partial(const double* __restrict__ M, int n, const double* x, double* part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = lane; r < n; r += 64) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int c0 = wave; c0 < n; c0 += MVG_U * NW) {
            double mv[MVG_U], xv[MVG_U];
#pragma unroll
            for (int q = 0; q < MVG_U; ++q) { const int c = c0 + q * NW; const bool in = c < n; mv[q] = in ? M[r + (size_t)c * n] : 0.0; xv[q] = in ? x[c] : 0.0; }
#pragma unroll
            for (int q = 0; q < MVG_U; q += 4) { a0 += mv[q] * xv[q]; a1 += mv[q + 1] * xv[q + 1]; a2 += mv[q + 2] * xv[q + 2]; a3 += mv[q + 3] * xv[q + 3]; }
        }
        part[wave * n + r] = (a0 + a1) + (a2 + a3);
    }
}
__device__ __forceinline__ double mvg_sum(const double* part, int n, int r) {
    double a = part[r];
#pragma unroll
    for (int w = 1; w < NW; ++w) a += part[w * n + r];
    return a;
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
    const double* Lg;                               // the QP's Hessian block P of this instance, column-major nx x nx, in global memory
    double *Z, *S, *q, *bh, *lam, *sol, *cand, *step, *res, *rerr, *corr, *rsym, *fx, *gzx, *gh, *ghc, *cprod, *bgrad, *wz, *wsoc, *bsoc, *vsoc, *D, *Dinv, *xb, *t1, *t2, *ycol, *red;
    const int *soc_start, *soc_dim, *soc_woff;
    double* filt;                                   // global: [pairs theta | pairs merit | cache theta | cache merit | saved theta | saved merit], max_filter each
    double* stf;                                    // global: the slacks s and t (nc each) the cone Jacobians of the LAST search direction were formed at (differentiate!'s quirk B-12)
    // uniform scalars (every thread holds the same values)
    double kappa, tau, rho, ep, ep_last, ed, fcur, fcand, eqv, cpv, omega_y, kyy;
    long long filter_index, nfact_total, rfail, rmax, rlast, nsteps;
    int tid, mf;      // mf = options.max_filter
    // S: the lower triangle packed by rows, (i, k) at i (i + 1) / 2 + k.  Lanes walk a column (rows i, i + 1, ...: the triangular numbers are a permutation modulo a
    // power of two — no bank conflict) or a row (consecutive words)
    __device__ __forceinline__ static int si(int i, int k) { return ((i * (i + 1)) >> 1) + k; }
#ifdef SN_TRACE
    long long tph[12]; long long tlast;
    __device__ __forceinline__ void stamp(int k) { const long long t = wall_clock64(); tph[k] += t - tlast; tlast = t; }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif

    // ---- evaluate! of the QP (qp.hip): which = the point (sol / cand) ------------------------------------------------------------------------
    __device__ __forceinline__ double eval_objective(const double* p) {        // f = 1/2 x'Lxx x + q'x   (uses xb as scratch)
        mvg_partial(Lg, d.nx, p, ycol);
        __syncthreads();
        double v[2] = {0.0, 0.0};
        for (int i = tid; i < d.nx; i += NT) { v[0] += p[i] * mvg_sum(ycol, d.nx, i); v[1] += q[i] * p[i]; }
        block_sum(v, red);
        return 0.5 * v[0] + v[1];
    }
    __device__ __forceinline__ void eval_constraints(const double* p, double* out) {      // [g; h] = [A; -G] x + [-b; hvec]
        mv_n(Z, d.ldz, d.m, d.nx, p, out, bh);
        __syncthreads();
    }
    __device__ __forceinline__ void eval_gradients(const double* p) {                     // fx = Lxx x + q ; gzx = A'y + (-G)'z
        mvg_partial(Lg, d.nx, p, ycol);
        mv_t(Z, d.ldz, d.m, d.nx, p + d.oy(), gzx, nullptr);
        __syncthreads();
        for (int i = tid; i < d.nx; i += NT) fx[i] = mvg_sum(ycol, d.nx, i) + q[i];
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
        mvg_partial(Lg, d.nx, v, ycol);
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
            out[c] = (mvg_sum(ycol, d.nx, c) + ep * v[c]) + ((a0 + a1) + (a2 + a3));
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
                                for (int c = 0; c < JB; ++c) pr[r][c] = (i < d.nx && c < jb && j0 + c <= i) ? S[si(i, j0 + c)] : 0.0;
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
                                for (int c = 0; c < JB; ++c) if (i < d.nx && c < jb && j0 + c < i) S[si(i, j0 + c)] = pr[r][c];
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
            const double lxx = Lg[j + (size_t)i * d.nx];      // (issued first: the dot products below cover its trip to L2)
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
            double v = lxx + a;
            if (i == j) v += ep;
            S[e] = v;                // (e IS the packed index of (i, j))
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
                    for (int u = 0; u < JB; ++u) li[u] = u < jb ? S[si(i, j0 + u)] : 0.0;
                    // three entries of the row at a time: their chains of 8 dependent multiply-adds interleave (one entry alone is ~230 cycles of latency)
                    constexpr int KU = 3;
                    for (int k0 = base + tk; k0 <= i; k0 += 16 * KU) {
                        double v[KU];
#pragma unroll
                        for (int q = 0; q < KU; ++q) { const int k = k0 + 16 * q; v[q] = k <= i ? S[si(i, k)] : 0.0; }
#pragma unroll
                        for (int u = 0; u < JB; ++u) {
#pragma unroll
                            for (int q = 0; q < KU; ++q) { const int k = k0 + 16 * q; if (u < jb && k <= i) v[q] -= li[u] * pan[u * d.nx + k]; }
                        }
#pragma unroll
                        for (int q = 0; q < KU; ++q) { const int k = k0 + 16 * q; if (k <= i) S[si(i, k)] = v[q]; }
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
                    for (int q = 0; q < PF; ++q) l[u][q] = (u < nchunk && i < d.nx && k0 + q < d.nx && i > k0 + q) ? S[si(i, k0 + q)] : 0.0;
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
                    for (int q = 0; q < PF; ++q) { const int k = k1 - 1 - q; l[u][q] = (u < nchunk && k >= 0 && i < k) ? S[si(k, i)] : 0.0; }
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
    // :183-185 cone!(jacobian = true): the cone Jacobians of this search direction are functions of THIS point's s and t.  The reference keeps them as fields, and
    // differentiate! (differentiate.jl:13-16) reads them where the last search direction left them — at the iterate BEFORE the final one (quirk B-12): remember which
    for (int i = tid; i < d.nc; i += NT) { c.stf[i] = sol[d.os() + i]; c.stf[d.nc + i] = sol[d.ot() + i]; }
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
        mvg_partial(c.Lg, d.nx, cand, c.ycol);
        mv_n(c.Z, d.ldz, d.m, d.nx, cand, c.ghc, c.bh);
        __syncthreads();
        double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};      // Phi, lambda'r, r'r, theta numerator, x'Lxx x, q'x
        for (int i = tid; i < d.nx; i += NT) { v[4] += cand[i] * mvg_sum(c.ycol, d.nx, i); v[5] += c.q[i] * cand[i]; }
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

// the instance's context: the LDS carve, the problem data and the point into LDS (every thread of the workgroup; ends with the data written, not yet synchronised)
template <bool SOC> __device__ __forceinline__ void bind_instance(CtxT<SOC>& c, const Args& a, double* sm, int inst, int tid) {
    const Dm d = a.d;
    const Lay L = layout(d);
    c.d = d; c.o = &a.o; c.tid = tid;
    c.Lg = a.P + (size_t)inst * a.sP; c.Z = sm + L.Z; c.S = sm + L.S; c.q = sm + L.q; c.bh = sm + L.bh; c.lam = sm + L.lam; c.sol = sm + L.sol; c.cand = sm + L.cand; c.step = sm + L.step;
    c.res = sm + L.res; c.rerr = sm + L.rerr; c.corr = sm + L.corr; c.rsym = sm + L.rsym;
    c.fx = sm + L.fx; c.gzx = sm + L.gzx; c.gh = sm + L.gh; c.ghc = sm + L.ghc; c.cprod = sm + L.cprod; c.bgrad = sm + L.bgrad; c.wz = sm + L.wz; c.wsoc = sm + L.wsoc; c.bsoc = sm + L.bsoc; c.vsoc = sm + L.vsoc;
    c.soc_start = a.soc_start; c.soc_dim = a.soc_dim; c.soc_woff = a.soc_woff;
    c.D = sm + L.D; c.Dinv = sm + L.Dinv; c.xb = sm + L.xb; c.t1 = sm + L.t1; c.t2 = sm + L.t2; c.ycol = sm + L.ycol; c.red = sm + L.red;
    c.mf = (int)a.o.max_filter;
    c.filt = a.filt + (size_t)inst * 6 * (size_t)c.mf;
    c.stf = a.stf + (size_t)inst * 2 * (size_t)(d.nc > 0 ? d.nc : 1);
    const double* q = a.q + (size_t)inst * a.sq;
    const double* Zg = a.Z + (size_t)inst * a.sZ; const double* bh = a.bh + (size_t)inst * a.sbh;
    for (int e = tid; e < d.m * d.nx; e += NT) c.Z[(e % d.m) + (e / d.m) * d.ldz] = Zg[e];
    for (int i = tid; i < d.nx; i += NT) c.q[i] = q[i];
    for (int i = tid; i < d.m; i += NT) c.bh[i] = bh[i];
    const double* w = a.w + (size_t)inst * d.N;
    for (int i = tid; i < d.N; i += NT) c.sol[i] = w[i];
}

// differentiate!(solver) for every instance of the batch (differentiate.jl:1-61) at the resident point: the condensed matrix for the regularisation the last
// factorisation of solve! left (:13-20: residual_jacobian_variables!, the symmetric form, factorize!), then per parameter column search_direction_symmetric! on
// the column of dR/dtheta (:29-52) and sensitivity = -1.0 * the result (:55-57).  dR/dtheta comes from the caller (residual_jacobian_parameters.jl:1-40 is the
// caller's model: for the parametric QPs of the MPC loops it is constant); `count` = its columns.
template <bool SOC> __global__ __launch_bounds__(NT, 2) void k_smallnewton_diff(Args a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (inst >= a.batch) return;
    CtxT<SOC> c;
    bind_instance(c, a, sm, inst, tid);
    const Dm& d = c.d;
    const double* gsc = a.sc + (size_t)inst * SC_COUNT;
    c.kappa = gsc[SC_KAPPA]; c.tau = gsc[SC_TAU]; c.rho = gsc[SC_RHO]; c.ep = gsc[SC_EP]; c.ep_last = gsc[SC_EPLAST]; c.ed = gsc[SC_ED];
    c.eqv = 0.0; c.cpv = 0.0; c.fcur = 0.0; c.fcand = 0.0; c.omega_y = 0.0; c.kyy = 0.0;
    c.filter_index = 0; c.nfact_total = 0; c.rfail = 0; c.rmax = 0; c.rlast = 0; c.nsteps = 0;
#ifdef SN_TRACE
    for (int k = 0; k < 12; ++k) c.tph[k] = 0;
    c.tlast = wall_clock64();
#endif
    __syncthreads();
    // the cone Jacobians as the reference's differentiate! finds them: formed at the s, t of the last search direction (quirk B-12), not at the solution
    for (int i = tid; i < d.nc; i += NT) { c.sol[d.os() + i] = c.stf[i]; c.sol[d.ot() + i] = c.stf[d.nc + i]; }
    __syncthreads();
    int zero = 0;
    const bool inertia_ok = c.factorize(zero);
    const double* J = a.rtheta + (size_t)inst * (size_t)a.srtheta;
    double* Sn = a.sens + (size_t)inst * (size_t)d.N * (size_t)a.count;
    for (int j = 0; j < a.count; ++j) {
        for (int i = tid; i < d.N; i += NT) c.res[i] = J[(size_t)j * d.N + i];
        __syncthreads();
        // search_direction_symmetric!, then — for batches WITHOUT second-order cones — the correction rounds of iterative_refinement! (iterative_refinement.jl:1-52),
        // ONE loop as in the Newton step.  The reference's differentiate! does not refine: its solve is QDLDL on the (nx + ne + nc) symmetric matrix; this path solves
        // the CONDENSED nx system, the same thing in exact arithmetic when all cones are nonnegative orthants, but at a solution (penalty 1e7, central path 1e-7) five
        // digits less accurate: the rounds give them back (against the oracle: 1e-5 without, 1e-9 with).  With second-order cones the reference's answer IS the
        // unrefined solve with its triu-symmetrised cone blocks (quirk B-3): refining would move away from it, towards H^-1
        int it = 0;
        bool first = true;
        for (;;) {
            c.search_direction_symmetric(first ? c.res : c.rerr, first ? c.step : c.corr);
            if (SOC || !a.o.iterative_refinement) break;
            if (!first) { for (int i = tid; i < d.N; i += NT) c.step[i] += c.corr[i]; __syncthreads(); it += 1; }
            const double norm = c.residual_error();
            first = false;
            if (it > a.o.max_iterative_refinement) break;
            if (norm <= a.o.iterative_refinement_tolerance && it >= a.o.min_iterative_refinement) break;
        }
        for (int i = tid; i < d.N; i += NT) Sn[(size_t)j * d.N + i] = -1.0 * c.step[i];
        __syncthreads();
    }
    if (tid == 0) a.status[inst] = inertia_ok ? 0 : 1;      // (1: the factorisation's inertia is not (nx, ne + nc, 0); the reference does not look, the sensitivities are what they are)
}

template <bool SOC> __global__ __launch_bounds__(NT, 2) void k_smallnewton(Args a) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (inst >= a.batch) return;
    CtxT<SOC> c;
    bind_instance(c, a, sm, inst, tid);
    const Dm d = a.d;
    const long long mf_ = c.mf;
    const Options& o = a.o;
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
