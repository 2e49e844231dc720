// lat_bench2.hip — issue rate of v_fmac_f64 / v_fmac_f64_dpp with three distinct 64-bit register operands, by VGPR bank placement (register number mod 4)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ long long g_t[32];
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43"
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define T(slot, body) { long long t0 = __builtin_readcyclecounter(); asm volatile(REP16(body) ::: CLOB); long long t1 = __builtin_readcyclecounter(); if (threadIdx.x == 0) g_t[slot] = t1 - t0; }
__global__ void k(double* out) {
    asm volatile("v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n"
                 "v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
                 "v_mov_b32 v28, 0\n v_mov_b32 v29, 0\n v_mov_b32 v30, 0\n v_mov_b32 v31, 0\n v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n"
                 "v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n" ::: CLOB);
    // four different destinations per group (like the column updates), sources fixed
    T(0, "v_fmac_f64 v[8:9], v[40:41], v[36:37]\n v_fmac_f64 v[12:13], v[40:41], v[36:37]\n v_fmac_f64 v[16:17], v[40:41], v[36:37]\n v_fmac_f64 v[20:21], v[40:41], v[36:37]\n")       // all = 0 mod 4
    T(1, "v_fmac_f64 v[8:9], v[42:43], v[36:37]\n v_fmac_f64 v[12:13], v[42:43], v[36:37]\n v_fmac_f64 v[16:17], v[42:43], v[36:37]\n v_fmac_f64 v[20:21], v[42:43], v[36:37]\n")       // src0 = 2 mod 4
    T(2, "v_fmac_f64 v[10:11], v[40:41], v[36:37]\n v_fmac_f64 v[14:15], v[40:41], v[36:37]\n v_fmac_f64 v[18:19], v[40:41], v[36:37]\n v_fmac_f64 v[22:23], v[40:41], v[36:37]\n")     // dst = 2 mod 4, sources 0 mod 4
    T(3, "v_fmac_f64 v[8:9], v[40:41], v[36:37]\n v_fmac_f64 v[10:11], v[40:41], v[36:37]\n v_fmac_f64 v[12:13], v[40:41], v[36:37]\n v_fmac_f64 v[14:15], v[40:41], v[36:37]\n")       // consecutive destinations
    T(4, "v_fmac_f64_dpp v[8:9], v[40:41], v[36:37] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[12:13], v[40:41], v[36:37] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
         "v_fmac_f64_dpp v[16:17], v[40:41], v[36:37] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[20:21], v[40:41], v[36:37] row_newbcast:6 row_mask:0xf bank_mask:0xf\n")
    T(5, "v_fmac_f64_dpp v[8:9], v[42:43], v[36:37] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[12:13], v[42:43], v[36:37] row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
         "v_fmac_f64_dpp v[16:17], v[42:43], v[36:37] row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[20:21], v[42:43], v[36:37] row_newbcast:6 row_mask:0xf bank_mask:0xf\n")
    T(6, "v_fmac_f64_dpp v[8:9], v[40:41], v[36:37] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[10:11], v[40:41], v[38:39] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
         "v_fmac_f64_dpp v[12:13], v[40:41], v[36:37] row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[14:15], v[40:41], v[38:39] row_newbcast:4 row_mask:0xf bank_mask:0xf\n")   // the (a, g) pairs of OWN = 1
    T(7, "v_fma_f64 v[8:9], v[40:41], v[36:37], v[8:9]\n v_fma_f64 v[12:13], v[40:41], v[36:37], v[12:13]\n v_fma_f64 v[16:17], v[40:41], v[36:37], v[16:17]\n v_fma_f64 v[20:21], v[40:41], v[36:37], v[20:21]\n")
    T(8, "v_fma_f64 v[8:9], v[42:43], v[36:37], v[8:9]\n v_fma_f64 v[12:13], v[42:43], v[36:37], v[12:13]\n v_fma_f64 v[16:17], v[42:43], v[36:37], v[16:17]\n v_fma_f64 v[20:21], v[42:43], v[36:37], v[20:21]\n")
    T(9, "v_fma_f64 v[8:9], s[4:5], v[36:37], v[8:9]\n v_fma_f64 v[12:13], s[4:5], v[36:37], v[12:13]\n v_fma_f64 v[16:17], s[4:5], v[36:37], v[16:17]\n v_fma_f64 v[20:21], s[4:5], v[36:37], v[20:21]\n")   // one source from SGPRs
    T(10, "v_mul_f64 v[8:9], v[40:41], v[36:37]\n v_mul_f64 v[12:13], v[40:41], v[36:37]\n v_mul_f64 v[16:17], v[40:41], v[36:37]\n v_mul_f64 v[20:21], v[40:41], v[36:37]\n")
    T(11, "v_add_f64 v[8:9], v[40:41], v[8:9]\n v_add_f64 v[12:13], v[40:41], v[12:13]\n v_add_f64 v[16:17], v[40:41], v[16:17]\n v_add_f64 v[20:21], v[40:41], v[20:21]\n")
    out[threadIdx.x] = 0;
}
int main() {
    double* d; hipMalloc(&d, 8 * 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipDeviceSynchronize();
    long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof(h));
    const char* nm[] = {"v_fmac_f64, dst / src0 / src1 all = 0 mod 4", "v_fmac_f64, src0 = 2 mod 4", "v_fmac_f64, dst = 2 mod 4", "v_fmac_f64, consecutive destinations", "v_fmac_f64_dpp, all = 0 mod 4",
                        "v_fmac_f64_dpp, src0 = 2 mod 4", "v_fmac_f64_dpp, (a, g) pairs", "v_fma_f64 (VOP3), all = 0 mod 4", "v_fma_f64, src0 = 2 mod 4", "v_fma_f64, src0 from SGPRs", "v_mul_f64", "v_add_f64"};
    for (int i = 0; i < 12; ++i) printf("%-50s %5.2f cycles per instruction\n", nm[i], h[i] / 64.0);
    return 0;
}
