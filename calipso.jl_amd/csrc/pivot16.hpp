// pivot16.hpp — 16 pivot columns of a dense LDL^T factored by ONE wavefront in registers (lane = row), shared by the blocked factorisation of the Schur
// complement (ldl.hip: diag_block) and the fronts of the multifrontal factorisation (sparse.hip: k_mf_factor).  See ldl.hip for the design notes and
// bench/diag_bench3.hip, bench/lat_bench*.hip for the measurements behind it.
#pragma once
#include <hip/hip_runtime.h>

namespace calipso {

__device__ __forceinline__ double fast_rcp(double v) {   // v_rcp_f64 + 2 Newton steps (pivots are normal numbers; 0 -> inf as 1/0)
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// -- the owner of a round: 16 columns in registers.  Every statement is a volatile asm, so the order below IS the issue order: the updates of pivot J
// on column J + 1, the broadcast of the NEXT pivot, then its reciprocal chain threaded through the remaining updates of pivot J.  DPP reads of a VGPR
// need two wait states after a VALU write of it: only the pivot broadcast follows its producer that closely (s_nop 1 inside its string).
template <int K> __device__ __forceinline__ double bcast16(double v) {
    double m;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(v), "n"(K));
    return m;
}
#define DPP_UPD(K)                                                                                                                      \
    if constexpr ((K) < NC) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"                             \
                                         : "+v"(a[(K) & 15]) : "v"(yrep), "v"(nl), "n"((K) & 15))
template <int J, bool STORE_L, int NC = 16> struct Pivot {     // NC: columns of the panel (16, or 8 for a half panel)
    // on entry: rinv = reciprocal of pivot J; yrep (lane 16 m + k) = entry (16 r + k, p) of the pivot column p = 16 r + J — its rows of the diagonal
    // 16 x 16 block, replicated in every 16-lane row: what the row-local broadcast needs
    static __device__ __forceinline__ void run(double (&a)[16], unsigned yk_own, unsigned yk_rep, double* __restrict__ Lrow, int lane0, double rinv, double yrep) {
        const double nl = a[J] * -rinv;
        if constexpr (STORE_L) Lrow[J] = -nl;
        if constexpr (J + 1 < NC) {
            double rn, t, yn;
            int dlo, dhi;
            DPP_UPD(J + 1);
            // column J + 1 is final: publish it (the matrix-core update reads it from LDS anyway) and read it back replicated (one ds_read_b64; the LDS
            // queue of a wavefront is in order); the round trip hides behind the reciprocal chain, whose operand travels by v_readlane
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(yk_own), "v"(a[J + 1]), "n"((J + 1) * 8) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(yn) : "v"(yk_rep), "n"((J + 1) * 8) : "memory");
            asm volatile("s_nop 0\n\tv_readlane_b32 %0, %2, %4\n\tv_readlane_b32 %1, %3, %4" : "=&s"(dlo), "=&s"(dhi)
                         : "v"(__double2loint(a[J + 1])), "v"(__double2hiint(a[J + 1])), "s"(lane0 + J + 1));
            const double dn = __hiloint2double(dhi, dlo);
            asm volatile("v_rcp_f64 %0, %1" : "=v"(rn) : "s"(dn));
            DPP_UPD(J + 2); DPP_UPD(J + 3);
            asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(t) : "s"(dn), "v"(rn));
            DPP_UPD(J + 4);
            asm volatile("v_fmac_f64 %0, %1, %0" : "+v"(rn) : "v"(t));
            DPP_UPD(J + 5);
            asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(t) : "s"(dn), "v"(rn));
            DPP_UPD(J + 6);
            asm volatile("v_fmac_f64 %0, %1, %0" : "+v"(rn) : "v"(t));
            DPP_UPD(J + 7); DPP_UPD(J + 8); DPP_UPD(J + 9); DPP_UPD(J + 10); DPP_UPD(J + 11); DPP_UPD(J + 12); DPP_UPD(J + 13); DPP_UPD(J + 14); DPP_UPD(J + 15);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yn) :: "memory");
            Pivot<J + 1, STORE_L, NC>::run(a, yk_own, yk_rep, Lrow, lane0, rn, yn);
        }
    }
};
#undef DPP_UPD

}  // namespace calipso
