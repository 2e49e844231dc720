// schur_loop_bench.hip — adds the ingredients of k_schur's stage loop one at a time to the bare MFMA loop (16 wavefronts per CU, one
// workgroup per CU, 16x16x4 fp64 MFMA, 4 accumulators per wavefront) to see which one costs the matrix pipe its throughput:
//   B = one __syncthreads per stage of 8 k-steps, W = the 8 ds_write_b64 per thread per stage (double-buffered LDS), G = the 8 global
//   loads per thread per stage (register prefetch two stages ahead), as in calipso.jl_amd/csrc/schur.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LDK = 34, TILE = 128;
template <bool BAR, bool WR, bool GL, int ACCS>
__global__ __launch_bounds__(1024) void k(int stages, const double* __restrict__ G, int ldg, double* out) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * TILE * LDK; i += 1024) sm[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    const int wi = wave & 7, jt0 = wave < 8 ? 0 : 4;
    v4d acc[4];
    for (int m = 0; m < 4; ++m) acc[m] = (v4d){0, 0, 0, 0};
    double ra[4], rb[4];
    const int k = tid % 32, cbase = tid / 32;
    const double* gp = G + (size_t)blockIdx.x * 128 * ldg;
    for (int q = 0; q < 4; ++q) { ra[q] = 1.0 + q; rb[q] = 2.0 + q; }
    for (int st = 0; st < stages; ++st) {
        const double* As = sm + (size_t)(st & 1) * 2 * TILE * LDK;
        const double* Bs = As + TILE * LDK;
        if (WR) {
            double* An = sm + (size_t)((st + 1) & 1) * 2 * TILE * LDK;
#pragma unroll
            for (int it = 0; it < 4; ++it) { An[(cbase + it * 32) * LDK + k] = ra[it]; An[TILE * LDK + (cbase + it * 32) * LDK + k] = rb[it]; }
        }
        if (GL) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                ra[it] = gp[(st * 32 + k) + (size_t)(cbase + it * 32) * ldg];
                rb[it] = gp[(st * 32 + k) + (size_t)(cbase + it * 32 + 64) * ldg] * 1.5;
            }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const double a = As[(wi * 16 + fr) * LDK + kk * 4 + fk];
            double b[4];
#pragma unroll
            for (int n = 0; n < ACCS; ++n) b[n] = Bs[((jt0 + n) * 16 + fr) * LDK + kk * 4 + fk];
#pragma unroll
            for (int n = 0; n < ACCS; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[n], a, acc[n], 0, 0, 0);
        }
        if (BAR) __syncthreads();
    }
    double s = 0;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r];
    out[blockIdx.x * 1024 + tid] = s + ra[0] + rb[1];
}
template <bool BAR, bool WR, bool GL, int ACCS> void run(const char* name, const double* G, int ldg) {
    const int blocks = 256, stages = 78;     // C3: 2500 constraint rows = 78 stages of 32
    double* out; hipMalloc(&out, sizeof(double) * blocks * 1024);
    hipFuncSetAttribute((const void*)k<BAR, WR, GL, ACCS>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE * LDK * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<BAR, WR, GL, ACCS><<<blocks, 1024, 4 * TILE * LDK * 8>>>(stages, G, ldg, out); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<BAR, WR, GL, ACCS><<<blocks, 1024, 4 * TILE * LDK * 8>>>(stages, G, ldg, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double flop = 2.0 * 16 * 16 * 4 * (double)ACCS * 8 * stages * 16.0 * blocks;
    printf("%-64s %7.3f ms  %6.2f TFLOP/s\n", name, ms, flop / ms * 1e-9);
    hipFree(out);
}
int main() {
    const int ldg = 2500;
    double* G; hipMalloc(&G, sizeof(double) * (size_t)ldg * (128 * 256 + 256)); hipMemset(G, 0, sizeof(double) * (size_t)ldg * (128 * 256 + 256));
    run<true, true, true, 4>("k_schur's loop, operands all zero", G, ldg);
    {   // random operands: fp64 matrix-core power (and with it the sustained clock) depends on the data
        const size_t n = (size_t)ldg * (128 * 256 + 256);
        double* h = (double*)malloc(n * sizeof(double));
        unsigned long long s = 88172645463325252ull;
        for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (double)(s >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
        hipMemcpy(G, h, n * sizeof(double), hipMemcpyHostToDevice);
        free(h);
    }
    run<true, true, true, 4>("k_schur's loop, random operands", G, ldg);
    run<true, true, true, 4>("k_schur's loop, random operands (again)", G, ldg);
    run<false, false, false, 4>("MFMA + 5 ds_read_b64 per k-step", G, ldg);
    run<true, false, false, 4>("+ barrier per stage", G, ldg);
    run<true, true, false, 4>("+ barrier + LDS stores", G, ldg);
    run<true, true, true, 4>("+ barrier + LDS stores + global loads (k_schur's loop)", G, ldg);
    run<false, true, true, 4>("LDS stores + global loads, no barrier (racy, timing only)", G, ldg);
    run<true, false, true, 4>("+ barrier + global loads, no LDS stores", G, ldg);
    return 0;
}
