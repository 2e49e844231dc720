// lat_bench3.hip — latencies that bound the short phases of the diagonal block: a dependent chain of v_mfma_f64_16x16x4_f64 (accumulator feeds the next),
// an LDS round trip (ds_write_b64 -> ds_read_b64 of the same wavefront, and ds_read -> use), a workgroup barrier of 16 wavefronts.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ long long g_t[32];
__global__ __launch_bounds__(1024) void k(double* out) {
    __shared__ double lds[4096];
    const int tid = threadIdx.x;
    double x = out[tid & 63], y = out[64 + (tid & 63)];
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    lds[tid] = x;
    __syncthreads();
    long long t0, t1;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int it = 0; it < 16; ++it) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
    t1 = __builtin_readcyclecounter();
    if (tid == 0) g_t[0] = t1 - t0;
    // dependent through the B operand: the result of one MFMA is an operand of the next
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int it = 0; it < 16; ++it) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, acc[0], acc, 0, 0, 0); }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc));
    t1 = __builtin_readcyclecounter();
    if (tid == 0) g_t[1] = t1 - t0;
    // LDS: dependent read chain (address from the value read)
    int a = tid & 63;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int it = 0; it < 16; ++it) { const double v = lds[a]; a = (a + (int)v) & 1023; }
    t1 = __builtin_readcyclecounter();
    if (tid == 0) g_t[2] = t1 - t0;
    // LDS: write then read back, dependent
    double v = x;
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int it = 0; it < 16; ++it) { lds[1024 + tid] = v; asm volatile("" ::: "memory"); v = lds[1024 + (tid ^ 1)] + 1.0; }
    t1 = __builtin_readcyclecounter();
    if (tid == 0) g_t[3] = t1 - t0;
    // barrier
    t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int it = 0; it < 16; ++it) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }
    t1 = __builtin_readcyclecounter();
    if (tid == 0) g_t[4] = t1 - t0;
    out[tid] = acc[0] + acc[1] + v + a;
}
int main() {
    double* d; hipMalloc(&d, 8 * 2048); hipMemset(d, 0, 8 * 2048);
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d); hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d); hipDeviceSynchronize();
    long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof(h));
    const char* nm[] = {"v_mfma_f64_16x16x4 accumulate chain (16 wavefronts on the CU)", "v_mfma_f64_16x16x4, result feeds an operand of the next", "ds_read_b64 dependent (address from the value)", "ds_write_b64 -> ds_read_b64 dependent", "workgroup barrier (16 wavefronts, nothing outstanding)"};
    for (int i = 0; i < 5; ++i) printf("%-66s %7.1f cycles per step\n", nm[i], h[i] / 16.0);
    return 0;
}
