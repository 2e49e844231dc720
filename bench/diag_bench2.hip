// diag_bench2.hip — round-3 harness for the 64 x 64 diagonal block of ldl.hip (diag_block): the LDL^T in four-column mini-panels with the inverse
// X = L^-1 grown alongside by the wavefronts that are done, then M = X' D^-1 X.  Variants of the mini-panel exchange are timed on one block in
// isolation (one workgroup of 1024 threads), with a timeline of the phases, and checked: L D L' = A, X L = I, M A = I.
//   hipcc -O3 --offload-arch=gfx950 bench/diag_bench2.hip -o /tmp/diag_bench2 && /tmp/diag_bench2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
constexpr int NB = 64, LDT = NB + 2;
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ long long g_ts[16];
#define TS(k) do { if (threadIdx.x == 0) g_ts[k] = wall_clock64(); } while (0)

__device__ __forceinline__ double fast_rcp(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void lds_barrier() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// EXCH: 0 = (y, 1/d) as 8-byte columns + reciprocal vector, pivot-row entries by broadcast LDS reads  (round 2 + X)
//       1 = (y, l) as 16-byte pairs, pivot-row entries by v_readlane from the wavefront's own rows
//       2 = (y, l) as 16-byte pairs, pivot-row entries by broadcast LDS reads
// XM:   0 = no inverse (L only; X, M wrong), 1 = X by the finished wavefronts (v_readlane), 2 = the finished wavefronts only READ the panel,
//       3 = the finished wavefronts only do the arithmetic (no panel read; X wrong)
// SLEEP: wavefronts other than the next owner wait s_sleep SLEEP after the barrier.   PRIO: s_setprio 3 for the owner and the next owner.
template <int EXCH, int XM, int SLEEP, int PRIO>
__global__ __launch_bounds__(1024) void k_diag(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout, double* __restrict__ Mout) {
    constexpr int WAVES = 16, CPW = 4;
    __shared__ double XT[NB * LDT], XTs[NB * LDT];
    __shared__ v2d pan[2][4][NB];
    __shared__ double ypan[2][4][NB], rpan[2][4];
    __shared__ double dpiv[NB], dinv[NB];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[CPW], x[CPW], lfin[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = 4 * cg + c;
        a[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
        x[c] = (i == k) ? 1.0 : 0.0; lfin[c] = 0.0;
    }
    TS(0);
#pragma unroll 1
    for (int P = 0; P < WAVES; ++P) {
        const int buf = P & 1;
        if (PRIO) { if (cg == P || cg == P + 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
        if (cg == P) {
            double y[4], rinv[4], dv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dv[j] = readlane_d(a[j], 4 * P + j);
                rinv[j] = fast_rcp(dv[j]);
                y[j] = a[j];
                const double li = a[j] * rinv[j];
                lfin[j] = li;
#pragma unroll
                for (int k = j + 1; k < 4; ++k) a[k] -= li * readlane_d(a[j], 4 * P + k);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (EXCH == 0) ypan[buf][j][i] = y[j]; else pan[buf][j][i] = (v2d){y[j], lfin[j]};
            }
            if (i < 4) {
                const double dd = i == 0 ? dv[0] : i == 1 ? dv[1] : i == 2 ? dv[2] : dv[3], rr = i == 0 ? rinv[0] : i == 1 ? rinv[1] : i == 2 ? rinv[2] : rinv[3];
                dpiv[4 * P + i] = dd; dinv[4 * P + i] = rr;
                if (EXCH == 0) rpan[buf][i] = rr;
            }
        }
        lds_barrier();
        if (SLEEP && cg != P + 1) __builtin_amdgcn_s_sleep(SLEEP);
        const bool upd = cg > P, xw = !upd && XM != 0;
        double l[4] = {0, 0, 0, 0}, yrow[4] = {0, 0, 0, 0};
        if (upd || XM == 1 || XM == 2) {
            if (EXCH == 0) {
                double yl[4], rp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { yl[j] = ypan[buf][j][i]; rp[j] = rpan[buf][j]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double v = yl[j] * rp[j]; l[j] = (i > 4 * P + j) ? v : 0.0; }
            } else {
                v2d yl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) yl[j] = pan[buf][j][i];
#pragma unroll
                for (int j = 0; j < 4; ++j) { l[j] = (i > 4 * P + j) ? yl[j].y : 0.0; yrow[j] = yl[j].x; }
            }
        } else if (XM == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) l[j] = (i > 4 * P + j) ? 1e-3 : 0.0;
        }
        if (upd) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double yr[CPW];
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    if (EXCH == 0) yr[c] = ypan[buf][j][4 * cg + c];
                    else if (EXCH == 1) yr[c] = readlane_d(yrow[j], 4 * cg + c);
                    else yr[c] = pan[buf][j][4 * cg + c].x;
                }
#pragma unroll
                for (int c = 0; c < CPW; ++c) a[c] -= l[j] * yr[c];
            }
        } else if (xw && XM != 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double xs[CPW];
#pragma unroll
                for (int c = 0; c < CPW; ++c) xs[c] = readlane_d(x[c], 4 * P + j);
#pragma unroll
                for (int c = 0; c < CPW; ++c) x[c] -= l[j] * xs[c];
            }
        } else if (xw) {
            x[0] += l[0] + l[1] + l[2] + l[3];      // keep the loads alive
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    TS(1);
    {
        const double di = dinv[i];
#pragma unroll
        for (int c = 0; c < CPW; ++c) { XT[(4 * cg + c) * LDT + i] = x[c]; XTs[(4 * cg + c) * LDT + i] = x[c] * di; }
    }
    lds_barrier();
    TS(2);
    {
        const int wa = cg >> 2, wb = cg & 3, fr = i & 15, fk = i >> 4;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const double xa = XTs[(wa * 16 + fr) * LDT + 4 * kk + fk];
            const double xb = XT[(wb * 16 + fr) * LDT + 4 * kk + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, acc, 0, 0, 0);
        }
        TS(3);
#pragma unroll
        for (int q = 0; q < 4; ++q) Mout[(wb * 16 + fr) + (size_t)(wa * 16 + fk + 4 * q) * NB] = acc[q];
    }
    if (tid < NB) Dx[tid] = dpiv[tid];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = 4 * cg + c;
        Xout[i + (size_t)k * NB] = x[c];
        if (i > k) S[i + (size_t)k * ld] = lfin[c];
    }
    TS(4);
}

template <typename K>
void run(const char* name, K kern, const std::vector<double>& A0, bool with_x) {
    const int ld = NB, reps = 200;
    double *S, *D, *X, *M;
    hipMalloc(&S, sizeof(double) * NB * NB); hipMalloc(&D, sizeof(double) * NB); hipMalloc(&X, sizeof(double) * NB * NB); hipMalloc(&M, sizeof(double) * NB * NB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(S, A0.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 0, 0, ld, S, D, X, M);   // (repeats refactor garbage: timing only)
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipMemcpy(S, A0.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), 0, 0, ld, S, D, X, M);
    std::vector<double> L(NB * NB), d(NB), Xh(NB * NB), Mh(NB * NB);
    hipMemcpy(L.data(), S, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), D, sizeof(double) * NB, hipMemcpyDeviceToHost);
    hipMemcpy(Xh.data(), X, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    hipMemcpy(Mh.data(), M, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    long long h[16]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts), sizeof(h));
    double err = 0, errx = 0, errm = 0;
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0;
        for (int k = 0; k <= j; ++k) s += (i == k ? 1.0 : L[i + k * NB]) * d[k] * (j == k ? 1.0 : L[j + k * NB]);
        err = fmax(err, fabs(s - A0[i + j * NB]));
        if (with_x) { double t = 0; for (int k = j; k <= i; ++k) t += Xh[i + k * NB] * (k == j ? 1.0 : L[k + j * NB]); errx = fmax(errx, fabs(t - (i == j ? 1.0 : 0.0))); }
    }
    if (with_x) for (int i = 0; i < NB; ++i) for (int j = 0; j < NB; ++j) { double t = 0; for (int k = 0; k < NB; ++k) t += Mh[i + k * NB] * A0[k + j * NB]; errm = fmax(errm, fabs(t - (i == j ? 1.0 : 0.0))); }
    printf("%-58s %6.2f us/launch | pivots %5.2f  X->LDS %4.2f  M %4.2f  stores %4.2f us | |LDL'-A| %.1e |XL-I| %.1e |MA-I| %.1e\n", name, best * 1e3 / reps,
           (h[1] - h[0]) / 100.0, (h[2] - h[1]) / 100.0, (h[3] - h[2]) / 100.0, (h[4] - h[3]) / 100.0, err, errx, errm);
    hipFree(S); hipFree(D); hipFree(X); hipFree(M);
}

int main() {
    std::vector<double> A(NB * NB);
    unsigned s = 12345;
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) {
        s = s * 1664525u + 1013904223u;
        double v = ((s >> 8) & 0xffff) / 65536.0 - 0.5;
        A[i + j * NB] = A[j + i * NB] = (i == j) ? 8.0 + v : v * 0.2;
    }
    run("e0: 8-byte panel + broadcast reads, no X", k_diag<0, 0, 0, 0>, A, false);
    run("e0: 8-byte panel + broadcast reads, X", k_diag<0, 1, 0, 0>, A, true);
    run("e0: X, prio", k_diag<0, 1, 0, 1>, A, true);
    run("e0: finished waves only read the panel", k_diag<0, 2, 0, 0>, A, false);
    run("e0: finished waves only do the arithmetic", k_diag<0, 3, 0, 0>, A, false);
    run("e0: X, sleep 1", k_diag<0, 1, 1, 0>, A, true);
    run("e0: X, sleep 2, prio", k_diag<0, 1, 2, 1>, A, true);
    run("e1: 16-byte pairs + readlane row, no X", k_diag<1, 0, 0, 0>, A, false);
    run("e1: 16-byte pairs + readlane row, X", k_diag<1, 1, 0, 0>, A, true);
    run("e1: X, sleep 2, prio", k_diag<1, 1, 2, 1>, A, true);
    run("e2: 16-byte pairs + broadcast reads, no X", k_diag<2, 0, 0, 0>, A, false);
    run("e2: 16-byte pairs + broadcast reads, X", k_diag<2, 1, 0, 0>, A, true);
    run("e2: X, prio", k_diag<2, 1, 0, 1>, A, true);
    run("e2: X, sleep 1, prio", k_diag<2, 1, 1, 1>, A, true);
    return 0;
}
