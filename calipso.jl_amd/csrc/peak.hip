// peak.hip — measured fp64 matrix-core ceiling of the device, for the roofline of bench.py (`roofline.peak_measured`).
// MI355X_MICROARCH.md does not tabulate fp64; the datasheet figure (78.6 TFLOP/s) is what `roofline.frac` is priced against, and
// this micro-benchmark says how much of it back-to-back independent v_mfma_f64_16x16x4_f64 instructions can sustain (8 wavefronts
// per SIMD, 4 independent accumulators each, nothing else in the loop): the ceiling any fp64 MFMA kernel on this chip sees.
#include <hip/hip_runtime.h>

#include "../../include/calipso_hip.h"

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma_f64_peak(int iters, double* __restrict__ out) {
    v4d acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int32_t calipso_hip_mfma_f64_peak(int32_t device, double* tflops) {
    if (!tflops) return CALIPSO_ERR_ARGUMENT;
    *tflops = 0.0;
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return CALIPSO_ERR_HIP;
    const int blocks = prop.multiProcessorCount * 8, iters = 5000;       // 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
    double* out = nullptr;
    hipEvent_t e0, e1;
    if (hipMalloc((void**)&out, sizeof(double) * (size_t)blocks * 256) != hipSuccess) return CALIPSO_ERR_HIP;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0.0;
    for (int rep = 0; rep < 4; ++rep) {                                   // the first launch ramps the clocks
        (void)hipEventRecord(e0, nullptr);
        hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, nullptr, iters, out);
        (void)hipEventRecord(e1, nullptr);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipFree(out); return CALIPSO_ERR_HIP; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * 16 * 4 * 4.0 * iters * 4.0 * blocks;
        if (rep > 0 && ms > 0.f) best = flop / (ms * 1e-3) * 1e-12 > best ? flop / (ms * 1e-3) * 1e-12 : best;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(out);
    *tflops = best;
    return CALIPSO_OK;
}
