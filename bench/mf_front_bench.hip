// Timeline of one multifrontal front (sparse.hip: k_mf_factor) in isolation: a front of m rows with c pivot columns, packed lower triangle in LDS,
// two children of r x r update matrices extend-added from global memory.  Stamps (wall_clock64, 100 MHz) after every phase of workgroup 0.
//   hipcc -O3 --offload-arch=gfx950 bench/mf_front_bench.hip -o /tmp/mf_front && /tmp/mf_front [m c threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int tri(int i, int k) { return i * (i + 1) / 2 + k; }

template <int T, int V>
__global__ __launch_bounds__(T) void k_front(int m, int c, const double* __restrict__ A, const double* __restrict__ U1, const double* __restrict__ U2, const int* __restrict__ rel,
                                              double* __restrict__ P, double* __restrict__ Uo, double* __restrict__ D, long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) double F[];
    constexpr int RC = T / 16;
    const int tid = threadIdx.x, r = m - c, nt = m * (m + 1) / 2;
    double* rinv = F + nt;
    int sp = 0;
#define STAMP() do { __syncthreads(); if (tid == 0 && blockIdx.x == 0) stamps[sp] = wall_clock64(); ++sp; } while (0)
    STAMP();
    for (int e = tid; e < nt; e += T) F[e] = 0.0;
    STAMP();
    for (int e = tid; e < c * m; e += T) { const int i = e % m, k = e / m; if (i >= k) F[tri(i, k)] = A[e]; }      // the node's own columns
    STAMP();
    __shared__ int relS[256];
    for (int ch = 0; ch < 2; ++ch) {
        const double* U = ch ? U2 : U1;
        if (V == 0) {
            for (int e = tid; e < r * r; e += T) { const int a = e / r, b = e - a * r; if (a >= b) F[tri(rel[a], rel[b])] += U[e]; }
        } else {
            for (int a = tid; a < r; a += T) relS[a] = rel[a];
            __syncthreads();
            for (int a = tid >> 5; a < r; a += T / 32) {                  // a row of the child's update matrix per 32 lanes, coalesced along b
                const int ra = relS[a] * (relS[a] + 1) / 2;
                const double* Ua = U + (size_t)a * r;
                for (int b = tid & 31; b <= a; b += 32) F[ra + relS[b]] += Ua[b];
            }
        }
        __syncthreads();
    }
    STAMP();
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    long long tp = 0, tm = 0;
    for (int kb = 0; kb < c; kb += 16) {
        const int pe = min(kb + 16, c);
        const long long t0 = wall_clock64();
        if (V == 2) {
            // the pivot column travels through a contiguous double-buffered vector (conflict-free reads); whoever updates (i, j + 1) publishes it
            double* ycol = rinv + m;                                        // [2][m]
            if (kb == 0) { for (int i = tid; i < m; i += T) ycol[i] = F[tri(i, 0)]; }
            for (int j = kb; j < pe; ++j) {
                __syncthreads();
                const double* y = ycol + (j & 1) * m;
                double* yn = ycol + ((j + 1) & 1) * m;
                const double dj = y[j];
                const double rj = 1.0 / dj;
                if (tid == 0) { D[j] = dj; rinv[j] = rj; }
                const int k = j + 1 + (tid & 15);
                if (k < pe) {
                    const double ykj = y[k] * rj;
                    int i = j + 1 + (tid >> 4);
                    if (i < k) i += ((k - i + RC - 1) / RC) * RC;
                    for (; i < m; i += RC) {
                        double* Fi = F + i * (i + 1) / 2;
                        const double v = Fi[k] - y[i] * ykj;
                        Fi[k] = v;
                        if (k == j + 1) yn[i] = v;
                    }
                } else if (k == pe && pe < c) {                             // the first column of the NEXT panel is not touched by this panel's steps:
                    // it is refreshed by the matrix-core update below; its pivot column is re-read after that update (see below)
                }
            }
            __syncthreads();
        } else
        for (int j = kb; j < pe; ++j) {
            __syncthreads();
            const double dj = F[tri(j, j)];
            const double rj = 1.0 / dj;
            if (tid == 0) { D[j] = dj; rinv[j] = rj; }
            const int k = j + 1 + (tid & 15);
            if (k < pe) {
                const double ykj = F[tri(k, j)] * rj;
                int i = j + 1 + (tid >> 4);
                if (i < k) i += ((k - i + RC - 1) / RC) * RC;
                if (V == 0) { for (; i < m; i += RC) { double* Fi = F + i * (i + 1) / 2; Fi[k] -= Fi[j] * ykj; } }
                else {
                    for (; i + 3 * RC < m; i += 4 * RC) {                // four rows per round: their loads are independent, issue them together
                        double* F0 = F + i * (i + 1) / 2; double* F1 = F + (i + RC) * (i + RC + 1) / 2;
                        double* F2 = F + (i + 2 * RC) * (i + 2 * RC + 1) / 2; double* F3 = F + (i + 3 * RC) * (i + 3 * RC + 1) / 2;
                        const double y0 = F0[j], y1 = F1[j], y2 = F2[j], y3 = F3[j];
                        const double a0 = F0[k], a1 = F1[k], a2 = F2[k], a3 = F3[k];
                        F0[k] = a0 - y0 * ykj; F1[k] = a1 - y1 * ykj; F2[k] = a2 - y2 * ykj; F3[k] = a3 - y3 * ykj;
                    }
                    for (; i < m; i += RC) { double* Fi = F + i * (i + 1) / 2; Fi[k] -= Fi[j] * ykj; }
                }
            }
        }
        __syncthreads();
        const long long t1 = wall_clock64();
        const int ntl = (m - pe + 15) / 16;
        if (V != 0) {
            // a wave takes whole tile rows (largest first): the first operand's fragments are read once per row and reused for every tile of the row
            for (int q = wave; q < ntl; q += T / 64) {
                const int bi = ntl - 1 - q;
                const int ia = pe + 16 * bi + fr;
                const int ra = ia * (ia + 1) / 2;
                double av[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) { const int k = kb + 4 * kk + fk; av[kk] = (k < pe && ia < m) ? F[ra + k] * rinv[k] : 0.0; }
                for (int bj = 0; bj <= bi; ++bj) {
                    const int jb = pe + 16 * bj + fr;
                    const int rb = jb * (jb + 1) / 2;
                    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int k = kb + 4 * kk + fk;
                        const double bv = (k < pe && jb < m) ? F[rb + k] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) { const int i = pe + 16 * bi + fk + 4 * qq, j = pe + 16 * bj + fr; if (i < m && j <= i) F[tri(i, j)] -= acc[qq]; }
                }
            }
        } else
        for (int t = wave; t < ntl * (ntl + 1) / 2; t += T / 64) {
            int bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
            while (bi * (bi + 1) / 2 > t) --bi;
            const int bj = t - bi * (bi + 1) / 2;
            const int ia = pe + 16 * bi + fr, jb = pe + 16 * bj + fr;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = kb + 4 * kk + fk;
                const bool kin = k < pe;
                const double av = (kin && ia < m) ? F[tri(ia, k)] * rinv[k] : 0.0;
                const double bv = (kin && jb < m) ? F[tri(jb, k)] : 0.0;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = pe + 16 * bi + fk + 4 * q, j = pe + 16 * bj + fr; if (i < m && j <= i) F[tri(i, j)] -= acc[q]; }
        }
        __syncthreads();
        if (V == 2 && pe < c) { double* ycol = rinv + m; for (int i = tid; i < m; i += T) if (i >= pe) ycol[(pe & 1) * m + i] = F[tri(i, pe)]; }
        tp += t1 - t0; tm += wall_clock64() - t1;
    }
    STAMP();
    for (int e = tid; e < m * c; e += T) { const int i = e % m, k = e / m; P[e] = i > k ? F[tri(i, k)] * rinv[k] : 0.0; }
    for (int e = tid; e < r * r; e += T) { const int a = e / r, b = e - a * r; if (a >= b) Uo[e] = F[tri(c + a, c + b)]; }
    STAMP();
    if (tid == 0 && blockIdx.x == 0) { stamps[sp] = tp; stamps[sp + 1] = tm; }
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 168, c = argc > 2 ? atoi(argv[2]) : 56, T = argc > 3 ? atoi(argv[3]) : 512, V = argc > 4 ? atoi(argv[4]) : 0, r = m - c;
    std::vector<double> hA((size_t)m * c), hU((size_t)r * r, 0.01);
    for (int k = 0; k < c; ++k) for (int i = 0; i < m; ++i) hA[(size_t)i + (size_t)k * m] = i == k ? 4.0 * m : 1.0 / (1 + ((i * 7 + k * 13) % 23));
    std::vector<int> hrel(r); for (int a = 0; a < r; ++a) hrel[a] = c + a;
    double *A, *U1, *U2, *P, *Uo, *D; int* rel; long long* st;
    CK(hipMalloc(&A, hA.size() * 8)); CK(hipMalloc(&U1, hU.size() * 8)); CK(hipMalloc(&U2, hU.size() * 8)); CK(hipMalloc(&P, hA.size() * 8));
    CK(hipMalloc(&Uo, hU.size() * 8)); CK(hipMalloc(&D, m * 8)); CK(hipMalloc(&rel, r * 4)); CK(hipMalloc(&st, 16 * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(U1, hU.data(), hU.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(U2, hU.data(), hU.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rel, hrel.data(), r * 4, hipMemcpyHostToDevice));
    const size_t lds = sizeof(double) * ((size_t)m * (m + 1) / 2 + 3 * m);
    for (const void* fn : {(const void*)k_front<256, 0>, (const void*)k_front<256, 1>, (const void*)k_front<512, 0>, (const void*)k_front<512, 1>, (const void*)k_front<256, 2>, (const void*)k_front<512, 2>})
        CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    for (int rep = 0; rep < 3; ++rep) {
        if (T == 512 && V == 0) hipLaunchKernelGGL((k_front<512, 0>), dim3(16), dim3(512), lds, 0, m, c, A, U1, U2, rel, P, Uo, D, st);
        else if (T == 512 && V == 2) hipLaunchKernelGGL((k_front<512, 2>), dim3(16), dim3(512), lds, 0, m, c, A, U1, U2, rel, P, Uo, D, st);
        else if (T == 256 && V == 2) hipLaunchKernelGGL((k_front<256, 2>), dim3(16), dim3(256), lds, 0, m, c, A, U1, U2, rel, P, Uo, D, st);
        else if (T == 512) hipLaunchKernelGGL((k_front<512, 1>), dim3(16), dim3(512), lds, 0, m, c, A, U1, U2, rel, P, Uo, D, st);
        else if (V == 0) hipLaunchKernelGGL((k_front<256, 0>), dim3(16), dim3(256), lds, 0, m, c, A, U1, U2, rel, P, Uo, D, st);
        else hipLaunchKernelGGL((k_front<256, 1>), dim3(16), dim3(256), lds, 0, m, c, A, U1, U2, rel, P, Uo, D, st);
        CK(hipDeviceSynchronize());
    }
    long long h[16]; CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    const double u = 0.01;   // microseconds per tick
    printf("front m=%d c=%d threads=%d variant %d: zero %.1f  own columns %.1f  extend-add(2 children) %.1f  factor %.1f (panel columns %.1f, matrix-core updates %.1f)  write-out %.1f  total %.1f us\n",
           m, c, T, V, (h[1] - h[0]) * u, (h[2] - h[1]) * u, (h[3] - h[2]) * u, (h[4] - h[3]) * u, h[6] * u, h[7] * u, (h[5] - h[4]) * u, (h[5] - h[0]) * u);
    std::vector<double> hP(hA.size()), hD(m); CK(hipMemcpy(hP.data(), P, hP.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hD.data(), D, c * 8, hipMemcpyDeviceToHost));
    double cs = 0; for (double v : hP) cs += v; for (int k = 0; k < c; ++k) cs += hD[k];
    printf("  checksum %.17g\n", cs);
    return 0;
}
