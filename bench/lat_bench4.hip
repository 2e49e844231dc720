// lat_bench4.hip — the pivot chain of one round of the diagonal wavefront (16 pivots of a 16 x 16 block, DPP updates; the code of diag_bench3.hip OWN = 4)
// timed alone (one wavefront in the workgroup) and with 15 more wavefronts waiting at a barrier, cold and warm instruction cache (first and later passes).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ long long g_t[64];
#define DPP_UPDG(K)                                                                                                                     \
    if constexpr ((K) < 16) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"                             \
                                         : "+v"(g[(K) & 15]) : "v"(gJ), "v"(nlg), "n"((K) & 15))
template <int J> struct PivA {
    static __device__ __forceinline__ void run(double (&g)[16], unsigned pub, unsigned tag, int tagval, double rinv) {
        const double gJ = g[J];
        const v2d pr = (v2d){gJ, rinv};
        asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(pub), "v"(pr), "n"(J * 1024) : "memory");
        asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(tag), "v"(tagval + J), "n"(J * 256) : "memory");
        if constexpr (J + 1 < 16) {
            double nlg, r0, dn, e, a1, e2, rn;
            asm volatile("v_mul_f64 %0, %1, -%2" : "=v"(nlg) : "v"(gJ), "v"(rinv));
            DPP_UPDG(J + 1);
            DPP_UPDG(J + 2);
            asm volatile("s_nop 1\n\tv_rcp_f64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r0) : "v"(g[J + 1]), "n"(J + 1));
            asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dn) : "v"(g[J + 1]), "n"(J + 1));
            DPP_UPDG(J + 3); DPP_UPDG(J + 4);
            asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(dn), "v"(r0));
            DPP_UPDG(J + 5); DPP_UPDG(J + 6);
            asm volatile("v_fma_f64 %0, %2, %3, %2\n\tv_mul_f64 %1, %3, %3" : "=&v"(a1), "=&v"(e2) : "v"(r0), "v"(e));
            DPP_UPDG(J + 7); DPP_UPDG(J + 8);
            asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(rn) : "v"(a1), "v"(e2));
            DPP_UPDG(J + 9); DPP_UPDG(J + 10); DPP_UPDG(J + 11); DPP_UPDG(J + 12); DPP_UPDG(J + 13); DPP_UPDG(J + 14); DPP_UPDG(J + 15);
            PivA<J + 1>::run(g, pub, tag, tagval, rn);
        }
    }
};
__global__ __launch_bounds__(1024) void k(double* out, int slot) {
    __shared__ v2d pubs[16 * 64];
    __shared__ int tags[16 * 64];
    const int tid = threadIdx.x, i = tid & 63, w = tid >> 6;
#pragma unroll 1
    for (int pass = 0; pass < 4; ++pass) {
        if (w == 0) {
            double g[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) g[c] = out[c * 64 + i] + (c == (i & 15) ? 8.0 : 0.1);
            const long long t0 = __builtin_readcyclecounter();
            PivA<0>::run(g, (unsigned)(uintptr_t)(pubs + i), (unsigned)(uintptr_t)(tags + i), 1, 0.125);
            asm volatile("s_nop 0" :: "v"(g[15]));
            const long long t1 = __builtin_readcyclecounter();
            if (i == 0) g_t[slot * 4 + pass] = t1 - t0;
            out[2048 + i] = g[15];
        }
        __syncthreads();
    }
}
int main() {
    double* d; hipMalloc(&d, 8 * 4096); hipMemset(d, 0, 8 * 4096);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, d, 1); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2); hipDeviceSynchronize();
    long long h[64]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof(h));
    const char* nm[] = {"one wavefront in the workgroup", "16 wavefronts, 15 of them at the barrier", "one wavefront again"};
    for (int s = 0; s < 3; ++s) printf("%-48s cycles for 16 pivots, passes 1-4: %lld %lld %lld %lld\n", nm[s], h[s * 4], h[s * 4 + 1], h[s * 4 + 2], h[s * 4 + 3]);
    return 0;
}
