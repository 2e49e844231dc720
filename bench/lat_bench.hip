// lat_bench.hip — dependent-issue latencies of the f64 operations on the pivot chain of the diagonal block (one wavefront, nothing else on the CU):
// cycles per operation of a chain of 64 dependent instructions, s_memtime around it (100 MHz wall clock is too coarse; s_memtime counts core clocks).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ long long g_t[32];
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define TIME(slot, body) { long long t0 = __builtin_readcyclecounter(); asm volatile(body : "+v"(x), "+v"(y) : "v"(z), "s"(sl)); long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) g_t[slot] = t1 - t0; }
__global__ void k(double* out, int sl) {
    double x = out[threadIdx.x], y = out[threadIdx.x + 64], z = out[threadIdx.x + 128];
    TIME(0, REP64("v_fma_f64 %0, %0, %2, %1\n\t"))
    TIME(1, REP64("v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"))
    TIME(2, REP64("v_rcp_f64 %0, %0\n\ts_nop 0\n\t"))
    TIME(3, REP64("v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"))
    TIME(4, REP64("v_fmac_f64_dpp %0, %2, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"))            // accumulate chain only (no DPP hazard on the source)
    TIME(5, REP64("v_fmac_f64_dpp %0, %2, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %2, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"))   // two independent
    TIME(6, REP64("v_fma_f64 %0, %0, %2, %0\n\tv_fma_f64 %1, %1, %2, %1\n\t"))                               // two independent chains
    TIME(7, REP64("v_mul_f64 %0, %0, %2\n\t"))
    { int xi = (int)x; long long t0 = __builtin_readcyclecounter(); asm volatile(REP64("s_nop 0\n\tv_readlane_b32 s4, %0, %1\n\tv_mov_b32 %0, s4\n\t") : "+v"(xi) : "s"(sl) : "s4"); long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) g_t[8] = t1 - t0; x += xi; }
    TIME(9, REP64("v_rcp_f64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"))
    TIME(10, REP64("s_nop 0\n\t"))
    TIME(11, REP64("v_fmac_f64 %0, %2, %2\n\tv_fmac_f64 %1, %2, %2\n\t"))
    { long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
      for (int it = 0; it < 200; ++it) asm volatile(REP64("v_fmac_f64 %0, %2, %2\n\tv_fmac_f64 %1, %2, %2\n\t") : "+v"(x), "+v"(y) : "v"(z));
      long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
      if (threadIdx.x == 0) { g_t[20] = w1 - w0; g_t[21] = c1 - c0; } }
    out[threadIdx.x] = x + y;
}
int main() {
    double* d; hipMalloc(&d, 8 * 256); hipMemset(d, 0, 8 * 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 3); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 3); hipDeviceSynchronize();
    long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof(h));
    const char* nm[] = {"v_fma_f64 dependent", "v_fmac_f64_dpp dependent through the DPP source (+s_nop 1)", "v_rcp_f64 dependent (+s_nop 0)", "v_mov_b64_dpp dependent (+s_nop 1)",
                        "v_fmac_f64_dpp, accumulator chain", "2 independent v_fmac_f64_dpp (per pair)", "2 independent v_fma_f64 (per pair)", "v_mul_f64 dependent",
                        "v_readlane_b32 + v_mov from SGPR (+s_nop 0)", "v_rcp_f64_dpp dependent (+s_nop 1)", "s_nop 0", "2 independent v_fmac_f64 (per pair)"};
    for (int i = 0; i < 12; ++i) printf("%-66s %6.1f clocks (s_memtime units) per step\n", nm[i], h[i] / 64.0);
    printf("calibration: 25600 v_fmac_f64 in %lld wall-clock ticks (10 ns) and %lld cycle-counter ticks: counter at %.0f MHz, one v_fmac_f64 = %.2f ns\n", h[20], h[21], h[21] / (h[20] * 0.01), h[20] * 10.0 / 25600);
    return 0;
}
