// mfma_f64_peak.hip — micro-benchmark of v_mfma_f64_16x16x4_f64 on gfx950: the fp64 matrix-core ceiling that the roofline
// of the Schur-complement / trailing-update kernels is priced against (not tabulated in MI355X_MICROARCH.md).
// Reports wall-clock TFLOP/s, and — from s_memtime (shader clock) vs the constant 100 MHz wall_clock64 — the shader clock
// actually sustained and the cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 bench/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak && gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(int iters, double* out, long long* clk) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

// the 4x4x4 (4 blocks) form: 512 flop per instruction, one accumulator double per lane
template <int NACC>
__global__ __launch_bounds__(256) void k4(int iters, double* out, long long* clk) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = 0; clk[1] = 0; }
}
template <int NACC>
void run4(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    double* out; long long* clk;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k4<NACC><<<blocks, 256>>>(iters, out, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k4<NACC><<<blocks, 256>>>(iters, out, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 4 * 4 * 4 * 4 * (double)NACC * iters * 4.0 * blocks;
    printf("4x4x4: acc=%2d waves/SIMD=%d iters=%d: %8.3f ms  %6.2f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flop / ms * 1e-9);
    hipFree(out); hipFree(clk);
}

template <int NACC>
void run(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    double* out; long long* clk; long long h[2];
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(iters, out, clk);   // warm (clock ramp)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(iters, out, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks;
    const double shader_ghz = (double)h[0] / ((double)h[1] / 100e6) * 1e-9;
    printf("acc=%2d waves/SIMD=%d iters=%d: %8.3f ms  %6.2f TFLOP/s | s_memtime %.3f GHz, %.1f shader-cycles per MFMA per SIMD\n", NACC,
           blocks_per_cu, iters, ms, flop / ms * 1e-9, shader_ghz, (double)h[0] / ((double)NACC * iters * blocks_per_cu));
    hipFree(out); hipFree(clk);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s  CUs=%d  clockRate=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<1>(1, 20000); run<4>(1, 20000); run<8>(1, 20000); run<16>(1, 10000);
    run<4>(2, 20000); run<8>(2, 20000); run<4>(4, 10000); run<4>(8, 5000);
    run4<4>(1, 40000); run4<16>(1, 20000); run4<8>(2, 20000); run4<8>(4, 20000); run4<16>(4, 10000); run4<8>(8, 10000);
    return 0;
}
