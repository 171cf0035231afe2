// issue_bench.hip -- lab tool (never part of libpzg.so): what one gfx950 CU issues per cycle, by instruction class and by
// the number of resident waves.  DESIGN.md's issue-port model of inflate_kernel is priced with these numbers.
//
//   hipcc -O3 --offload-arch=gfx950 tests/tools/issue_bench.hip -o build/issue_bench && build/issue_bench
//
// Every pattern is REPT copies of a short instruction group inside a loop; one workgroup = one wave; W workgroups per CU
// (the LDS allocation caps the residency at W, so the dispatcher cannot pile them up).  Reported: instructions of the
// pattern per CU-cycle (shader cycles from s_memtime, the clock from s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

#define STR2(x) #x
#define STR(x) STR2(x)
#define REPT 32

enum Pat {
    P_VALU_E32 = 0,
    P_VALU_VOP3,
    P_VALU_SDWA,
    P_VALU_DPP,
    P_READLANE,
    P_CNDMASK_S,
    P_VCMP_S,
    P_SALU,
    P_BITSET,
    P_BR_NT,
    P_BR_T,
    P_WALK,
    P_MIX_VS,
    P_BPERM,
    P_DSREAD,
    P_DSWRITE8,
    P_NOP,
    P_WAIT,
    P_VALU_DEP,
    P_SALU_DEP,
    P_MIX_V2S,
    P_WALK_T,
    P_VOP2_LIT,
    P_VOP1,
    P_MBCNT,
    P_ADD_E64,
    P_LSHL_ADD,
    P_VOPC_E32,
    P_CNDMASK_E32,
    P_SALU_LIT,
    P_SOPK,
    P_SALU64,
    P_MIX_48,
    P_MIX_8S,
    P_BFE,
    P_CMP_CND_E32,
    P_CND_E64_VCC,
    P_CND_E32_SETVCC,
    P_CMP_CND_E64,
    P_BITOP3,
    P_LSHL_E32,
    P_PERM,
    P_DSREAD_U8,
    P_DSPERMUTE,
    P_WALK2,
    P_CMP_4CND,
    P_SOR_CND_MIX,
    P_SOR_CND64_MIX,
    P_VCMP_CND_FAR,
    P_DECODER_MIX,
    P_COUNT
};
static const char *pat_name[P_COUNT] = {"valu_e32(4B,indep)", "valu_vop3 alignbit(8B)", "valu_sdwa", "valu_dpp row_shr", "v_readlane->s",
                                        "v_cndmask sgpr-mask", "v_cmp->sgpr", "salu s_add(indep)", "s_bitset1_b64", "s_cmp+cbranch(not taken)",
                                        "s_branch(taken,next)", "walk step x4 (nt)", "mix 1 valu+1 salu", "ds_bpermute", "ds_read_b32",
                                        "ds_write_b8", "s_nop 0", "s_waitcnt lgkm(0)", "valu_e32 dependent", "salu dependent", "mix 2 valu+1 salu",
                                        "walk step, exit taken/8", "vop2 e32 + literal(8B)", "vop1 v_mov e32", "v_mbcnt_lo (vop3)", "v_add_u32_e64 (vop3,2op)", "v_lshl_add_u32", "v_cmp e32 -> vcc", "v_cndmask e32 (vcc)", "s_add + literal(8B)", "s_addk (sopk)", "s_and_b64/s_bcnt1_b64", "mix 1 e32 + 1 vop3", "mix 1 vop3 + 1 salu", "v_bfe_u32 (vop3)", "v_cmp_e32+v_cndmask_e32", "v_cndmask_e64 vcc", "v_cndmask_e32 (vcc set once)", "v_cmp_e64+v_cndmask_e64 sgpr", "v_bitop3_b32", "v_lshlrev_b32_e32 const", "v_perm_b32 sgpr sel", "ds_read_u8", "ds_permute", "walk step, 26 bits/token", "1 v_cmp_e32 + 4 v_cndmask_e32", "s_or vcc + cndmask_e32 + 6 valu", "s_or sgpr + cndmask_e64 + 6 valu", "v_cmp_e32, 6 valu, cndmask_e32", "decoder-like: 6 vop3+2 e32+7 salu+2 br+1 bperm"};
// instructions per group (for the rate)
static const int pat_insts[P_COUNT] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 1, 4, 2, 1, 1, 1, 4, 1, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 2, 4, 2, 4, 4, 2, 4, 4, 4, 1, 1, 4, 5, 8, 8, 8, 18};

template <int PAT>
__global__ void __launch_bounds__(64) bench_kernel(uint32_t iters, uint64_t *out, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t v0 = lane + seed, v1 = lane * 3u + 1u, v2 = lane ^ 5u, v3 = seed, v4 = lane << 2;
    uint32_t s0 = seed, s1 = seed + 1u, s2 = 2u, s3 = 3u;
    uint64_t m = 0x5555555555555555ull ^ seed, m2 = 0;
    lds[lane] = lane;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
        if (PAT == P_VALU_E32)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_add_u32_e32 %2, %4, %2\n\tv_add_u32_e32 %3, %4, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_VALU_DEP)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_e32 %0, %1, %0\n\tv_add_u32_e32 %0, %1, %0\n\tv_add_u32_e32 %0, %1, %0\n\tv_add_u32_e32 %0, %1, %0\n\t.endr"
                         : "+v"(v0) : "v"(v4));
        if (PAT == P_VALU_VOP3)
            asm volatile(".rept " STR(REPT) "\n\tv_alignbit_b32 %0, %4, %0, %4\n\tv_alignbit_b32 %1, %4, %1, %4\n\tv_alignbit_b32 %2, %4, %2, %4\n\tv_alignbit_b32 %3, %4, %3, %4\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_VALU_SDWA)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_add_u32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_add_u32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                         "v_add_u32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_VALU_DPP)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %1, %4, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %2, %4, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_u32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_READLANE)
            asm volatile(".rept " STR(REPT) "\n\tv_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %4, 5\n\tv_readlane_b32 %2, %4, 7\n\tv_readlane_b32 %3, %4, 9\n\t.endr"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(v4));
        if (PAT == P_CNDMASK_S)
            asm volatile(".rept " STR(REPT) "\n\tv_cndmask_b32_e64 %0, %0, %4, %5\n\tv_cndmask_b32_e64 %1, %1, %4, %5\n\tv_cndmask_b32_e64 %2, %2, %4, %5\n\tv_cndmask_b32_e64 %3, %3, %4, %5\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "s"(m));
        if (PAT == P_VCMP_S)
            asm volatile(".rept " STR(REPT) "\n\tv_cmp_lt_u32_e64 %0, %1, %2\n\tv_cmp_lt_u32_e64 %0, %2, %3\n\tv_cmp_lt_u32_e64 %0, %3, %4\n\tv_cmp_lt_u32_e64 %0, %4, %1\n\t.endr"
                         : "=s"(m2) : "v"(v0), "v"(v1), "v"(v2), "v"(v3));
        if (PAT == P_SALU)
            asm volatile(".rept " STR(REPT) "\n\ts_add_u32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_add_u32 %3, %3, %4\n\t.endr"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed) : "scc");
        if (PAT == P_SALU_DEP)
            asm volatile(".rept " STR(REPT) "\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %0, %0, %1\n\t.endr"
                         : "+s"(s0) : "s"(seed) : "scc");
        if (PAT == P_BITSET)
            asm volatile(".rept " STR(REPT) "\n\ts_bitset1_b64 %0, %2\n\ts_bitset1_b64 %1, %3\n\ts_bitset1_b64 %0, %3\n\ts_bitset1_b64 %1, %2\n\t.endr"
                         : "+s"(m), "+s"(m2) : "s"(s2), "s"(s3));
        if (PAT == P_BR_NT)
            asm volatile(".rept " STR(REPT) "\n\ts_cmp_eq_u32 %0, -1\n\ts_cbranch_scc1 9f\n\t.endr\n9:" : : "s"(s2) : "scc");
        if (PAT == P_BR_T)
            asm volatile(".rept " STR(REPT) "\n\ts_branch 8f\n8:\n\t.endr" : : "s"(s2));
        if (PAT == P_WALK) {
            uint32_t kb = 0xffffffc0u, t;  // TB = 0 in every lane: the add never carries
            uint32_t tb = 0;
            asm volatile(".rept " STR(REPT) "\n\ts_bitset1_b64 %0, %1\n\tv_readlane_b32 %2, %3, %1\n\ts_add_u32 %1, %1, %2\n\ts_cbranch_scc1 9f\n\t.endr\n9:"
                         : "+s"(m), "+s"(kb), "=&s"(t) : "v"(tb) : "scc");
            s0 += kb;
        }
        if (PAT == P_WALK_T) {
            // a token of 8 bits at every offset: eight steps per half, the eighth leaves (taken branch), as in walk_half
            uint32_t tb = 8u;
            for (int h = 0; h < REPT / 8; ++h) {
                uint32_t kb = 0xffffffc0u, t;
                asm volatile("1:\n\t.rept 8\n\ts_bitset1_b64 %0, %1\n\tv_readlane_b32 %2, %3, %1\n\ts_add_u32 %1, %1, %2\n\ts_cbranch_scc1 2f\n\t.endr\n\ts_branch 1b\n2:"
                             : "+s"(m), "+s"(kb), "=&s"(t) : "v"(tb) : "scc");
                s0 += kb;
            }
        }
        if (PAT == P_MIX_VS)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_e32 %0, %2, %0\n\ts_add_u32 %1, %1, %3\n\t.endr" : "+v"(v0), "+s"(s0) : "v"(v4), "s"(seed) : "scc");
        if (PAT == P_MIX_V2S)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_e32 %0, %3, %0\n\tv_add_u32_e32 %1, %3, %1\n\ts_add_u32 %2, %2, %4\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+s"(s0) : "v"(v4), "s"(seed) : "scc");
        if (PAT == P_BPERM)
            asm volatile(".rept " STR(REPT) "\n\tds_bpermute_b32 %0, %1, %0\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0) : "v"(v4));
        if (PAT == P_DSREAD)
            asm volatile(".rept " STR(REPT) "\n\tds_read_b32 %0, %1\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "=v"(v0) : "v"(v4));
        if (PAT == P_DSWRITE8)
            asm volatile(".rept " STR(REPT) "\n\tds_write_b8 %1, %0\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : : "v"(v0), "v"(v4) : "memory");

        if (PAT == P_VOP2_LIT)
            asm volatile(".rept " STR(REPT) "\n\tv_and_b32_e32 %0, 0x3fc3fc, %0\n\tv_and_b32_e32 %1, 0x3fc3fc, %1\n\tv_and_b32_e32 %2, 0x3fc3fc, %2\n\tv_and_b32_e32 %3, 0x3fc3fc, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        if (PAT == P_VOP1)
            asm volatile(".rept " STR(REPT) "\n\tv_mov_b32_e32 %0, %4\n\tv_mov_b32_e32 %1, %4\n\tv_mov_b32_e32 %2, %4\n\tv_mov_b32_e32 %3, %4\n\t.endr"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(v4));
        if (PAT == P_MBCNT)
            asm volatile(".rept " STR(REPT) "\n\tv_mbcnt_lo_u32_b32 %0, %4, %0\n\tv_mbcnt_lo_u32_b32 %1, %4, %1\n\tv_mbcnt_lo_u32_b32 %2, %4, %2\n\tv_mbcnt_lo_u32_b32 %3, %4, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "s"(seed));
        if (PAT == P_ADD_E64)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_e64 %0, %4, %0\n\tv_add_u32_e64 %1, %4, %1\n\tv_add_u32_e64 %2, %4, %2\n\tv_add_u32_e64 %3, %4, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_LSHL_ADD)
            asm volatile(".rept " STR(REPT) "\n\tv_lshl_add_u32 %0, %4, 2, %0\n\tv_lshl_add_u32 %1, %4, 2, %1\n\tv_lshl_add_u32 %2, %4, 2, %2\n\tv_lshl_add_u32 %3, %4, 2, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_BFE)
            asm volatile(".rept " STR(REPT) "\n\tv_bfe_u32 %0, %0, %4, %4\n\tv_bfe_u32 %1, %1, %4, %4\n\tv_bfe_u32 %2, %2, %4, %4\n\tv_bfe_u32 %3, %3, %4, %4\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_VOPC_E32)
            asm volatile(".rept " STR(REPT) "\n\tv_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cmp_lt_u32_e32 vcc, %1, %2\n\tv_cmp_lt_u32_e32 vcc, %2, %3\n\tv_cmp_lt_u32_e32 vcc, %3, %0\n\t.endr"
                         : : "v"(v0), "v"(v1), "v"(v2), "v"(v3) : "vcc");
        if (PAT == P_CNDMASK_E32)
            asm volatile(".rept " STR(REPT) "\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc");
        if (PAT == P_SALU_LIT)
            asm volatile(".rept " STR(REPT) "\n\ts_add_u32 %0, %0, 0x12345\n\ts_add_u32 %1, %1, 0x12345\n\ts_add_u32 %2, %2, 0x12345\n\ts_add_u32 %3, %3, 0x12345\n\t.endr"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (PAT == P_SOPK)
            asm volatile(".rept " STR(REPT) "\n\ts_addk_i32 %0, 0x123\n\ts_addk_i32 %1, 0x123\n\ts_addk_i32 %2, 0x123\n\ts_addk_i32 %3, 0x123\n\t.endr"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (PAT == P_SALU64)
            asm volatile(".rept " STR(REPT) "\n\ts_and_b64 %0, %0, %1\n\ts_bcnt1_i32_b64 %2, %1\n\ts_or_b64 %1, %1, %0\n\ts_bcnt1_i32_b64 %3, %0\n\t.endr"
                         : "+s"(m), "+s"(m2), "=s"(s0), "=s"(s1) : : "scc");
        if (PAT == P_MIX_48)
            asm volatile(".rept " STR(REPT) "\n\tv_add_u32_e32 %0, %2, %0\n\tv_alignbit_b32 %1, %2, %1, %2\n\t.endr" : "+v"(v0), "+v"(v1) : "v"(v4));
        if (PAT == P_MIX_8S)
            asm volatile(".rept " STR(REPT) "\n\tv_alignbit_b32 %0, %2, %0, %2\n\ts_add_u32 %1, %1, %3\n\t.endr" : "+v"(v0), "+s"(s0) : "v"(v4), "s"(seed) : "scc");

        if (PAT == P_CMP_CND_E32)
            asm volatile(".rept " STR(REPT) "\n\tv_cmp_lt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %2, %2, %3, vcc\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2) : "v"(v4) : "vcc");
        if (PAT == P_CND_E64_VCC)
            asm volatile(".rept " STR(REPT) "\n\tv_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc\n\tv_cndmask_b32_e64 %2, %2, %4, vcc\n\tv_cndmask_b32_e64 %3, %3, %4, vcc\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc");
        if (PAT == P_CND_E32_SETVCC) {
            asm volatile("s_mov_b64 vcc, %0" : : "s"(m) : "vcc");
            asm volatile(".rept " STR(REPT) "\n\tv_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc");
        }
        if (PAT == P_CMP_CND_E64)
            asm volatile(".rept " STR(REPT) "\n\tv_cmp_lt_u32_e64 %4, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %2, %2, %3, %4\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2) : "v"(v4), "s"(m2));
        if (PAT == P_BITOP3)
            asm volatile(".rept " STR(REPT) "\n\tv_bitop3_b32 %0, %0, %4, %1 bitop3:0xca\n\tv_bitop3_b32 %1, %1, %4, %2 bitop3:0xca\n\tv_bitop3_b32 %2, %2, %4, %3 bitop3:0xca\n\tv_bitop3_b32 %3, %3, %4, %0 bitop3:0xca\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_LSHL_E32)
            asm volatile(".rept " STR(REPT) "\n\tv_lshlrev_b32_e32 %0, 2, %0\n\tv_lshrrev_b32_e32 %1, 8, %1\n\tv_ashrrev_i32_e32 %2, 31, %2\n\tv_sub_u32_e32 %3, %4, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4));
        if (PAT == P_PERM)
            asm volatile(".rept " STR(REPT) "\n\tv_perm_b32 %0, %0, %4, %5\n\tv_perm_b32 %1, %1, %4, %5\n\tv_perm_b32 %2, %2, %4, %5\n\tv_perm_b32 %3, %3, %4, %5\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "s"(seed));
        if (PAT == P_DSREAD_U8)
            asm volatile(".rept " STR(REPT) "\n\tds_read_u8 %0, %1\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "=v"(v0) : "v"(lane * 3u));
        if (PAT == P_DSPERMUTE)
            asm volatile(".rept " STR(REPT) "\n\tds_permute_b32 %0, %1, %0\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(v0) : "v"(v4));
        if (PAT == P_WALK2) {
            // tokens of 26 bits: a taken exit every 2.5 steps on average (the real walk's exits are a tenth of its branches)
            uint32_t tb = 26u;
            for (int h = 0; h < REPT / 2; ++h) {
                uint32_t kb = 0xffffffc0u + (h & 15), t;
                asm volatile("1:\n\t.rept 8\n\ts_bitset1_b64 %0, %1\n\tv_readlane_b32 %2, %3, %1\n\ts_add_u32 %1, %1, %2\n\ts_cbranch_scc1 2f\n\t.endr\n\ts_branch 1b\n2:"
                             : "+s"(m), "+s"(kb), "=&s"(t) : "v"(tb) : "scc");
                s0 += kb;
            }
        }

        if (PAT == P_CMP_4CND)
            asm volatile(".rept " STR(REPT) "\n\tv_cmp_lt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\tv_cndmask_b32_e32 %3, %3, %4, vcc\n\t"
                         "v_cndmask_b32_e32 %0, %0, %4, vcc\n\tv_cndmask_b32_e32 %1, %1, %4, vcc\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc");
        if (PAT == P_SOR_CND_MIX)   // what the emit code does: a mask formed on the scalar unit into vcc, one select, other vector work around it
            asm volatile(".rept " STR(REPT) "\n\ts_or_b64 vcc, %5, %6\n\tv_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\t"
                         "v_add_u32_e32 %3, %4, %3\n\tv_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_add_u32_e32 %3, %4, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4), "s"(m), "s"(m2) : "vcc", "scc");
        if (PAT == P_SOR_CND64_MIX) {
            uint64_t mm = 0;
            asm volatile(".rept " STR(REPT) "\n\ts_or_b64 %4, %6, %7\n\tv_add_u32_e32 %0, %5, %0\n\tv_add_u32_e32 %1, %5, %1\n\tv_cndmask_b32_e64 %2, %2, %5, %4\n\t"
                         "v_add_u32_e32 %3, %5, %3\n\tv_add_u32_e32 %0, %5, %0\n\tv_add_u32_e32 %1, %5, %1\n\tv_add_u32_e32 %3, %5, %3\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(mm) : "v"(v4), "s"(m), "s"(m2) : "scc");
            s0 += (uint32_t)mm;
        }
        if (PAT == P_VCMP_CND_FAR)
            asm volatile(".rept " STR(REPT) "\n\tv_cmp_lt_u32_e32 vcc, %0, %1\n\tv_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_add_u32_e32 %3, %4, %3\n\t"
                         "v_add_u32_e32 %0, %4, %0\n\tv_add_u32_e32 %1, %4, %1\n\tv_add_u32_e32 %3, %4, %3\n\tv_cndmask_b32_e32 %2, %2, %4, vcc\n\t.endr"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(v4) : "vcc");

        if (PAT == P_DECODER_MIX)  // the decoder's own proportions per 18 instructions: 8 vector (6 half-rate), 7 scalar, 2 branches, 1 crossbar gather
            asm volatile(".rept " STR(REPT) "\n\tv_alignbit_b32 %0, %6, %0, %6\n\ts_add_u32 %4, %4, %7\n\tv_bfe_u32 %1, %1, %6, %6\n\ts_add_u32 %5, %5, %7\n\t"
                         "v_add_u32_e32 %2, %6, %2\n\ts_cmp_eq_u32 %7, -1\n\ts_cbranch_scc1 9f\n\tv_perm_b32 %3, %3, %6, %7\n\ts_and_b32 %4, %4, %7\n\t"
                         "ds_bpermute_b32 %0, %6, %0\n\tv_mbcnt_lo_u32_b32 %1, %7, %1\n\ts_lshl_b32 %5, %5, 1\n\tv_sub_u32_e32 %2, %6, %2\n\ts_add_u32 %4, %4, %7\n\t"
                         "v_cndmask_b32_e64 %3, %3, %6, %8\n\ts_cmp_eq_u32 %7, -2\n\ts_cbranch_scc1 9f\n\ts_or_b32 %5, %5, %7\n\t.endr\n9:\n\ts_waitcnt lgkmcnt(0)"
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1) : "v"(v4), "s"(seed), "s"(m) : "scc");
        if (PAT == P_NOP) asm volatile(".rept " STR(REPT) "\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\t.endr");
        if (PAT == P_WAIT) asm volatile(".rept " STR(REPT) "\n\ts_waitcnt lgkmcnt(0)\n\t.endr");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    // keep every result alive
    uint32_t sink = v0 ^ v1 ^ v2 ^ v3 ^ s0 ^ s1 ^ s2 ^ s3 ^ (uint32_t)m ^ (uint32_t)m2;
    if (lane == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = (r1 - r0) | ((uint64_t)(sink & 1u) << 62);
    }
}

typedef void (*kern_t)(uint32_t, uint64_t *, uint32_t);
template <int P>
static void fill(kern_t *tab)
{
    tab[P] = bench_kernel<P>;
    if constexpr (P + 1 < P_COUNT) fill<P + 1>(tab);
}

int main(int argc, char **argv)
{
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
    kern_t tab[P_COUNT];
    fill<0>(tab);
    const int Ws[] = {4, 8, 16, 24, 28};
    uint64_t *d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 2 * ncu * 32));
    std::vector<uint64_t> h(2 * ncu * 32);
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000u;
    printf("%-28s", "pattern \\ waves per CU");
    for (int W : Ws) printf(" %13d", W);
    printf("   (instructions per CU-cycle; last column: MHz)\n");
    for (int p = argc > 2 ? P_COUNT - atoi(argv[2]) : 0; p < P_COUNT; ++p) {  // (argv[2] = n: only the last n patterns)
        printf("%-28s", pat_name[p]);
        double mhz = 0;
        for (int W : Ws) {
            const int grid = ncu * W;
            // LDS per workgroup so that at most W fit a CU (160 KiB); at least 256 B for the patterns' own use
            size_t ldsb = (160u * 1024u) / (size_t)W;
            if (ldsb > 64u * 1024u) ldsb = 64u * 1024u;
            ldsb &= ~(size_t)255;
            if (getenv("IB_NO_LDS_CAP")) ldsb = 1024;  // (residency then limited by wave slots only: 8 per SIMD)
            CK(hipFuncSetAttribute((const void *)tab[p], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            hipLaunchKernelGGL(tab[p], dim3(grid), dim3(64), ldsb, 0, 20u, d_out, 1u);  // warm-up
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(tab[p], dim3(grid), dim3(64), ldsb, 0, iters, d_out, 1u);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipEventDestroy(e0));
            CK(hipEventDestroy(e1));
            CK(hipMemcpy(h.data(), d_out, sizeof(uint64_t) * 2 * grid, hipMemcpyDeviceToHost));
            // median wave time in shader cycles
            std::vector<uint64_t> cyc(grid);
            double sc = 0, sr = 0;
            for (int i = 0; i < grid; ++i) {
                cyc[i] = h[2 * i];
                sc += (double)h[2 * i];
                sr += (double)(h[2 * i + 1] & 0xffffffffffffull);
            }
            std::sort(cyc.begin(), cyc.end());
            const double med = (double)cyc[grid / 2];
            const double insts = (double)iters * REPT * pat_insts[p];
            // W waves per CU each issue `insts` in `med` cycles
            mhz = sc / sr * 100.0;  // s_memrealtime ticks at 100 MHz
            // whole-launch rate: every wave's instructions over the launch's duration (HIP events) at the measured clock; beside it
            // the median wave's own rate x W (equal while no shared port saturates; above it when waves finish one after another)
            const double cyc_total = (double)ms * 1e-3 * mhz * 1e6;
            printf(" %6.3f/%-6.3f", insts * grid / (cyc_total * ncu), insts * W / med);
        }
        printf("   %6.0f\n", mhz);
    }
    return 0;
}
