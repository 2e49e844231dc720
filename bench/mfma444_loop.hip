// mfma444_loop.hip — which ingredient of a real GEMM inner loop costs v_mfma_f64_4x4x4_f64 its 76 TFLOP/s?  16 wavefronts per CU,
// per k-step: [LDS] 5 ds_read_b64 of operand fragments, [DPP] 3 row rotations (6 v_mov_b32_dpp), 16 MFMAs on 16 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ __forceinline__ double row_ror(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
typedef double v4d __attribute__((ext_vector_type(4)));
template <bool LDS, bool DPP, bool WIDE>
__global__ __launch_bounds__(1024) void k(int iters, double* out) {
    __shared__ double sm[2 * 128 * 34];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 34; i += 1024) sm[i] = 1e-3 * (i % 97);
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4, cj = wave & 7, rg = wave >> 3;
    double acc[4][4];
    v4d accw[4];
    for (int m = 0; m < 4; ++m) { accw[m] = (v4d){0, 0, 0, 0}; for (int r = 0; r < 4; ++r) acc[m][r] = 0.0; }
    double a[4] = {1.0 + lane, 2.0, 3.0, 4.0}, b0 = 0.5 + wave;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (LDS) {
                b0 = sm[128 * 34 + (cj * 16 + fr) * 34 + kk * 4 + fk];
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = sm[((4 * rg + m) * 16 + fr) * 34 + kk * 4 + fk];
            }
            if (WIDE) {
#pragma unroll
                for (int m = 0; m < 4; ++m) accw[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a[m], accw[m], 0, 0, 0);
            } else {
                double bq[4] = {b0, b0, b0, b0};
                if (DPP) { bq[1] = row_ror<4>(b0); bq[2] = row_ror<8>(b0); bq[3] = row_ror<12>(b0); }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[m][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(bq[r], a[m], acc[m][r], 0, 0, 0);
            }
        }
    }
    double s = 0;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r] + accw[m][r];
    out[blockIdx.x * 1024 + tid] = s;
}
template <bool LDS, bool DPP, bool WIDE> void run(const char* name, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, iters = 400;
    double* out; hipMalloc(&out, sizeof(double) * blocks * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<LDS, DPP, WIDE><<<blocks, 1024>>>(iters, out); hipDeviceSynchronize();
    hipEventRecord(e0); k<LDS, DPP, WIDE><<<blocks, 1024>>>(iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 16 * 16 * 4 * 4.0 * 8 * iters * 16.0 * blocks;     // 4 16x16x4-equivalents per k-step per wavefront
    printf("%-52s %d WG/CU: %7.3f ms  %6.2f TFLOP/s\n", name, wgs_per_cu, ms, flop / ms * 1e-9);
    hipFree(out);
}
int main() {
    run<false, false, true>("16x16x4, registers only", 1);
    run<true, false, true>("16x16x4 + 5 ds_read_b64 per k-step", 1);
    run<false, false, false>("4x4x4, registers only", 1);
    run<false, true, false>("4x4x4 + 3 DPP rotations", 1);
    run<true, false, false>("4x4x4 + 5 ds_read_b64 per k-step", 1);
    run<true, true, false>("4x4x4 + LDS + DPP (the k_schur inner loop)", 1);
    return 0;
}
