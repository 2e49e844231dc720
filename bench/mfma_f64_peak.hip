// mfma_f64_peak.hip — micro-benchmark of v_mfma_f64_16x16x4_f64 issue rate on gfx950: the fp64 matrix-core ceiling the
// roofline of the Schur-complement / trailing-update kernels is priced against (the value is not tabulated in
// /opt/skills/guides/MI355X_MICROARCH.md).  Every SIMD runs `waves` wavefronts issuing independent MFMA chains.
//   hipcc --offload-arch=gfx950 -O3 bench/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak && gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, double* out) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 4096;
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(64, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;   // 4 waves per block
    printf("acc=%d blocks/CU=%d : %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", NACC, blocks_per_cu, ms, flop / ms * 1e-9,
           ms * 1e-3 * 2.4e9 / ((double)NACC * iters * blocks_per_cu));
    hipFree(out);
}

int main() {
    run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<16>(1); run<4>(2); run<8>(2); run<16>(2);
    return 0;
}
