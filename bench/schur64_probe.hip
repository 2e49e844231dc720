// schur64_probe.hip — the Schur complement's products S(i, j) = sum_k Z[k, i] w_k Z[k, j] as 64 x 64 tiles by workgroups of 256 threads (4 wavefronts, a 32 x 32 sub-tile =
// 2 x 2 v_mfma_f64_16x16x4 tiles each), operands staged through LDS in stages of KT constraint rows with a register prefetch one stage ahead — the viability test of
// DESIGN.md 5.00 / 9 "what comes next" item 0: (1) what does such a kernel sustain on RANDOM data against k_schur's 43 TFLOP/s (1024-thread workgroups, 128 x 128 tiles),
// (2) does a streaming kernel of another stream run beside it (it does not beside kernels of >= 512-thread workgroups that issue matrix instructions: overlap_probe3/4)?
//   hipcc --offload-arch=gfx950 -O3 bench/schur64_probe.hip -o /tmp/s64 && /tmp/s64
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KT, int WPE>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(WPE, WPE))) void k_schur64(int nx, int m, const double* __restrict__ Z, const double* __restrict__ w,
                                                                                                                  double* __restrict__ S, int ntl) {
    constexpr int LDK = KT + 2;
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * 64 * LDK];      // two stages of (A tile, B tile)
    // tile list: lower triangle of nb x nb, row-major; XCD-aware: block b works on item (b % 8) * per + b / 8
    const int per = (ntl + 7) / 8;
    const int item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= ntl || (int)(blockIdx.x >> 3) >= per) return;
    int bi = (int)((sqrt(8.0 * item + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= item) ++bi;
    while (bi * (bi + 1) / 2 > item) --bi;
    const int bj = item - bi * (bi + 1) / 2;
    const int i0 = bi * 64, j0 = bj * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fk = lane >> 4;
    // staging: thread -> (k = tid % KT, columns tid / KT + (256 / KT) * it)
    constexpr int CPT = 64 * KT / 256;      // columns per thread and operand
    const int sk = tid % KT, sc = tid / KT;
    double ra[CPT], rb[CPT];
    auto fetch = [&](int st) {
        const int k = st * KT + sk;
        const bool kin = k < m;
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int c = sc + (256 / KT) * it;
            const int ci = i0 + c < nx ? i0 + c : nx - 1, cj = j0 + c < nx ? j0 + c : nx - 1;
            const double a = Z[(kin ? k : 0) + (size_t)ci * m], b = Z[(kin ? k : 0) + (size_t)cj * m];
            ra[it] = (kin && i0 + c < nx) ? a : 0.0;
            rb[it] = (kin && j0 + c < nx) ? b * w[kin ? k : 0] : 0.0;
        }
    };
    auto park = [&](int buf) {
        double* As = lds + buf * 2 * 64 * LDK; double* Bs = As + 64 * LDK;
#pragma unroll
        for (int it = 0; it < CPT; ++it) { const int c = sc + (256 / KT) * it; As[c * LDK + sk] = ra[it]; Bs[c * LDK + sk] = rb[it]; }
    };
    v4d acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0, 0, 0, 0};
    const int nst = (m + KT - 1) / KT;
    fetch(0); park(0);
    if (nst > 1) fetch(1);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const double* As = lds + (st & 1) * 2 * 64 * LDK; const double* Bs = As + 64 * LDK;
        if (st + 1 < nst) park((st + 1) & 1);
        if (st + 2 < nst) fetch(st + 2);
#pragma unroll
        for (int kk = 0; kk < KT / 4; ++kk) {
            double fa[2], fb[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) fa[a] = As[(wr * 32 + a * 16 + fr) * LDK + kk * 4 + fk];
#pragma unroll
            for (int b = 0; b < 2; ++b) fb[b] = Bs[(wc * 32 + b * 16 + fr) * LDK + kk * 4 + fk];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[b], fa[a], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = j0 + wc * 32 + b * 16 + fk + 4 * r, gi = i0 + wr * 32 + a * 16 + fr;
                if (gi < nx && gj < nx) S[gi + (size_t)gj * nx] = acc[a][b][r];
            }
}

__global__ __launch_bounds__(256) void kB(const double* __restrict__ src, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 15 * stride < n; i += 16 * stride) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    if (s == 1.2345e-300) out[0] = s;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int KT, int WPE> void run(int nx, int m, int inst, const double* Z, const double* w, double* S, hipStream_t sa, hipStream_t sb, const double* src, size_t n, double* out) {
    const int nb = (nx + 63) / 64, ntl = nb * (nb + 1) / 2;
    const int grid = ((ntl + 7) / 8) * 8;
    auto A = [&] { for (int q = 0; q < inst; ++q) hipLaunchKernelGGL((k_schur64<KT, WPE>), dim3(grid), dim3(256), 0, sa, nx, m, Z + (size_t)q * m * nx, w, S + (size_t)q * nx * nx, ntl); };
    auto B = [&] { for (int r = 0; r < 16; ++r) hipLaunchKernelGGL(kB, dim3(512), dim3(256), 0, sb, src, n, out); };
    A(); B(); hipDeviceSynchronize();
    double t0 = now(); A(); hipStreamSynchronize(sa); const double a = now() - t0;
    t0 = now(); B(); hipStreamSynchronize(sb); const double b = now() - t0;
    t0 = now(); A(); B(); hipStreamSynchronize(sb); const double bdone = now() - t0; hipStreamSynchronize(sa); const double both = now() - t0;
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k_schur64<KT, WPE>);
    const double flop = 2.0 * (double)nx * (nx + 1) / 2.0 * m * inst, flop_exec = 2.0 * 64.0 * 64.0 * ((m + KT - 1) / KT * KT) * (double)ntl * inst;
    printf("k_schur64<KT %2d, %d waves/EU> (%3d VGPRs, %5zu B LDS): %d instance(s) alone %.3f ms = %.1f TFLOP/s useful (%.1f executed) | B alone %.2f ms | together: B done %.2f, all %.2f ms = %.2f x (A + B)\n",
           KT, WPE, fa.numRegs, fa.sharedSizeBytes, inst, a, flop / a * 1e-9, flop_exec / a * 1e-9, b, bdone, both, both / (a + b));
}

int main() {
    const int nx = 2500, m = 2500, inst = 4;
    std::vector<double> h((size_t)m * nx * inst), hw(m);
    srand(1);
    for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
    for (auto& v : hw) v = 0.5 + (double)rand() / RAND_MAX;
    double *Z, *w, *S, *src, *out;
    hipMalloc(&Z, h.size() * 8); hipMalloc(&w, m * 8); hipMalloc(&S, (size_t)nx * nx * inst * 8);
    hipMemcpy(Z, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), m * 8, hipMemcpyHostToDevice);
    const size_t n = (size_t)1 << 27;
    hipMalloc(&src, n * 8); hipMemset(src, 0, n * 8); hipMalloc(&out, 64);
    int least, greatest; hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t sa, sb;
    hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, greatest); hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, least);
    run<16, 2>(nx, m, inst, Z, w, S, sa, sb, src, n, out);
    run<16, 3>(nx, m, inst, Z, w, S, sa, sb, src, n, out);
    run<16, 4>(nx, m, inst, Z, w, S, sa, sb, src, n, out);
    run<32, 2>(nx, m, inst, Z, w, S, sa, sb, src, n, out);
    run<32, 3>(nx, m, inst, Z, w, S, sa, sb, src, n, out);
    return 0;
}
