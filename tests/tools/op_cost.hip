// op_cost.hip -- lab tool: issue cost of single vector / scalar / LDS instructions on gfx950, one opcode per row.
// 16 waves per CU (4 per SIMD) run REPT x 4 independent copies of the instruction in a loop; reported: SIMD-cycles per
// instruction (vector), CU-cycles per instruction (scalar, LDS), from the launch's duration (HIP events) and the measured clock.
//   hipcc -O3 --offload-arch=gfx950 tests/tools/op_cost.hip -o build/op_cost && build/op_cost
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define STR2(x) #x
#define STR(x) STR2(x)
#define REPT 32

// T: instruction template with D = destination/accumulator (one of four registers), A = a vector source, S = a scalar source,
// M = a 64-bit scalar mask
#define OPS(X) \
    X(v_add_u32_e32,      "v_add_u32_e32 \\D, %18, \\D") \
    X(v_sub_u32_e32,      "v_sub_u32_e32 \\D, %18, \\D") \
    X(v_and_b32_e32,      "v_and_b32_e32 \\D, %18, \\D") \
    X(v_and_b32_lit,      "v_and_b32_e32 \\D, 0x3fc3fc, \\D") \
    X(v_and_b32_sgpr,     "v_and_b32_e32 \\D, %13, \\D") \
    X(v_or_b32_e32,       "v_or_b32_e32 \\D, %18, \\D") \
    X(v_xor_b32_e32,      "v_xor_b32_e32 \\D, %18, \\D") \
    X(v_mov_b32_e32,      "v_mov_b32_e32 \\D, %18") \
    X(v_lshlrev_b32_c,    "v_lshlrev_b32_e32 \\D, 2, \\D") \
    X(v_lshlrev_b32_v,    "v_lshlrev_b32_e32 \\D, %18, \\D") \
    X(v_lshrrev_b32_c,    "v_lshrrev_b32_e32 \\D, 8, \\D") \
    X(v_ashrrev_i32_c,    "v_ashrrev_i32_e32 \\D, 31, \\D") \
    X(v_mul_u32_u24_c,    "v_mul_u32_u24_e32 \\D, 4, \\D") \
    X(v_mul_u32_u24_v,    "v_mul_u32_u24_e32 \\D, %18, \\D") \
    X(v_min_u32_e32,      "v_min_u32_e32 \\D, %18, \\D") \
    X(v_max_u32_e32,      "v_max_u32_e32 \\D, %18, \\D") \
    X(v_cndmask_e32,      "v_cndmask_b32_e32 \\D, \\D, %18, vcc") \
    X(v_cndmask_e64_s,    "v_cndmask_b32_e64 \\D, \\D, %18, %12") \
    X(v_cmp_lt_u32_e32,   "v_cmp_lt_u32_e32 vcc, %18, \\D") \
    X(v_cmp_lt_u32_e64,   "v_cmp_lt_u32_e64 %12, %18, \\D") \
    X(v_add_u32_e64,      "v_add_u32_e64 \\D, %18, \\D") \
    X(v_add3_u32,         "v_add3_u32 \\D, %18, \\D, %18") \
    X(v_add_lshl_u32,     "v_add_lshl_u32 \\D, %18, \\D, 2") \
    X(v_lshl_add_u32,     "v_lshl_add_u32 \\D, \\D, 2, %18") \
    X(v_lshl_or_b32,      "v_lshl_or_b32 \\D, \\D, 2, %18") \
    X(v_and_or_b32,       "v_and_or_b32 \\D, \\D, %18, %18") \
    X(v_or3_b32,          "v_or3_b32 \\D, \\D, %18, %18") \
    X(v_bfi_b32,          "v_bfi_b32 \\D, %18, \\D, %18") \
    X(v_bfe_u32,          "v_bfe_u32 \\D, \\D, %18, %18") \
    X(v_bfe_u32_c,        "v_bfe_u32 \\D, \\D, 8, 8") \
    X(v_alignbit_b32,     "v_alignbit_b32 \\D, %18, \\D, %18") \
    X(v_alignbyte_b32,    "v_alignbyte_b32 \\D, %18, \\D, %18") \
    X(v_perm_b32,         "v_perm_b32 \\D, \\D, %18, %13") \
    X(v_bitop3_b32,       "v_bitop3_b32 \\D, \\D, %18, %18 bitop3:0xca") \
    X(v_bitop3_b32_s,     "v_bitop3_b32 \\D, \\D, %13, %18 bitop3:0x80") \
    X(v_mad_u32_u24,      "v_mad_u32_u24 \\D, \\D, %18, %18") \
    X(v_mbcnt_lo,         "v_mbcnt_lo_u32_b32 \\D, %13, \\D") \
    X(v_mbcnt_hi,         "v_mbcnt_hi_u32_b32 \\D, %13, \\D") \
    X(v_readlane,         "v_readlane_b32 \\S2, \\D, 5") \
    X(v_readfirstlane,    "v_readfirstlane_b32 \\S2, \\D") \
    X(v_writelane,        "v_writelane_b32 \\D, %13, 5") \
    X(v_add_u32_sdwa,     "v_add_u32_sdwa \\D, %18, \\D dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
    X(v_lshlrev_sdwa,     "v_lshlrev_b32_sdwa \\D, %18, \\D dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0") \
    X(v_and_b32_sdwa,     "v_and_b32_sdwa \\D, %18, \\D dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(v_add_u32_dpp,      "v_add_u32_dpp \\D, %18, \\D row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(v_mov_b32_dpp,      "v_mov_b32_dpp \\D, %18 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
    X(v_dot4_u32_u8,      "v_dot4_u32_u8 \\D, %18, %13, \\D") \
    X(v_lshlrev_b64,      "v_lshlrev_b64 \\W, 3, \\W") \
    X(v_pk_add_u16,       "v_pk_add_u16 \\D, \\D, %18") \
    X(v_pk_lshrrev_b16,   "v_pk_lshrrev_b16 \\D, %18, \\D") \
    X(s_add_u32,          "s_add_u32 \\S2, \\S2, %13") \
    X(s_lshl_b32,         "s_lshl_b32 \\S2, \\S2, 1") \
    X(s_bfe_u32,          "s_bfe_u32 \\S2, \\S2, 0x80008") \
    X(s_bcnt1_i32_b64,    "s_bcnt1_i32_b64 \\S2, %12") \
    X(s_ff1_i32_b64,      "s_ff1_i32_b64 \\S2, %12") \
    X(s_bfm_b64,          "s_bfm_b64 %12, %13, %13") \
    X(s_and_b64,          "s_and_b64 %12, %12, %12") \
    X(s_cselect_b32,      "s_cselect_b32 \\S2, %13, \\S2") \
    X(s_cmp_lt_u32,       "s_cmp_lt_u32 \\S2, %13") \
    X(s_mov_b32,          "s_mov_b32 \\S2, %13") \
    X(s_mov_b64,          "s_mov_b64 %12, exec") \
    X(ds_read_b32,        "ds_read_b32 \\D, %18") \
    X(ds_read_u8,         "ds_read_u8 \\D, %18") \
    X(ds_read_b64,        "ds_read_b64 \\W, %18") \
    X(ds_read_b128,       "ds_read_b128 \\Q, %19") \
    X(ds_read2_b32,       "ds_read2_b32 \\W, %18 offset1:1") \
    X(ds_write_b8,        "ds_write_b8 %18, \\D") \
    X(ds_write_b16,       "ds_write_b16 %18, \\D") \
    X(ds_write_b32,       "ds_write_b32 %18, \\D") \
    X(ds_bpermute_b32,    "ds_bpermute_b32 \\D, %18, \\D") \
    X(ds_permute_b32,     "ds_permute_b32 \\D, %18, \\D") \
    X(ds_swizzle_b32,     "ds_swizzle_b32 \\D, \\D offset:0x8055")

enum { 
#define X(name, t) OP_##name,
OPS(X)
#undef X
OP_COUNT };
static const char *op_name[] = {
#define X(name, t) #name,
OPS(X)
#undef X
};
static const char *op_text[] = {
#define X(name, t) t,
OPS(X)
#undef X
};

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t iters, uint64_t *out, uint32_t seed)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;
    uint32_t v0 = lane + seed, v1 = lane * 3u + 1u, v2 = lane ^ 5u, v3 = seed + 7u, a = (lane << 2) & 0xfc, a16 = (lane << 4) & 0x3f0;
    uint64_t w0 = lane, w1 = lane + 1, w2 = lane + 2, w3 = lane + 3;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q0 = {lane, 1, 2, 3}, q1 = q0, q2 = q0, q3 = q0;
    uint32_t s = seed | 2u, s2a = 1, s2b = 2, s2c = 3, s2d = 4;
    uint64_t m = 0x5555555555555555ull ^ seed;
    for (uint32_t i = lane; i < 1024; i += 64) lds[i] = i;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t it = 0; it < iters; ++it) {
#define X(name, t)                                                                                                         \
        if (OP == OP_##name)                                                                                                \
            asm volatile(".macro ONE D, W, Q, S2\n\t" t "\n\t.endm\n\t"                                                       \
                         ".rept " STR(REPT) "\n\tONE %0, %4, %8, %14\n\tONE %1, %5, %9, %15\n\tONE %2, %6, %10, %16\n\tONE %3, %7, %11, %17\n\t.endr\n\t" \
                         ".purgem ONE\n\ts_waitcnt lgkmcnt(0)"                                                               \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(q0), "+v"(q1),   \
                           "+v"(q2), "+v"(q3), "+s"(m), "+s"(s), "+s"(s2a), "+s"(s2b), "+s"(s2c), "+s"(s2d)                         \
                         : "v"(a), "v"(a16)                                                                                 \
                         : "vcc", "scc", "memory");
        OPS(X)
#undef X
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t sink = v0 ^ v1 ^ v2 ^ v3 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3) ^ q0.x ^ q1.y ^ q2.z ^ q3.w ^ s ^ s2a ^ s2b ^ s2c ^ s2d ^ (uint32_t)m;
    if (lane == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = (r1 - r0) | ((uint64_t)(sink & 1u) << 62);
    }
}

typedef void (*kern_t)(uint32_t, uint64_t *, uint32_t);
template <int P>
static void fill(kern_t *tab)
{
    tab[P] = k<P>;
    if constexpr (P + 1 < OP_COUNT) fill<P + 1>(tab);
}

int main(int argc, char **argv)
{
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, W = argc > 2 ? atoi(argv[2]) : 16;
    const uint32_t iters = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000u;
    kern_t tab[OP_COUNT];
    fill<0>(tab);
    uint64_t *d_out;
    const int grid = ncu * W;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * 2 * grid));
    std::vector<uint64_t> h(2 * grid);
    printf("%d CUs, %d waves per CU; cycles per instruction: per SIMD (= 4 x per CU), per CU, lone-wave; clock MHz\n", ncu, W);
    for (int p = 0; p < OP_COUNT; ++p) {
        if (argc > 3) {  // only the opcodes named in argv[3] (comma-separated, exact names)
            const std::string want = std::string(",") + argv[3] + ",";
            if (want.find(std::string(",") + op_name[p] + ",") == std::string::npos) continue;
        }
        double lone = 0;
        double res[2] = {0, 0}, mhz = 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int g = pass == 0 ? ncu : grid;  // one wave per CU, then W
            const size_t ldsb = pass == 0 ? 65536 : 8192;
            CK(hipFuncSetAttribute((const void *)tab[p], hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            hipLaunchKernelGGL(tab[p], dim3(g), dim3(64), ldsb, 0, 20u, d_out, 1u);
            CK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(tab[p], dim3(g), dim3(64), ldsb, 0, iters, d_out, 1u);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_out, sizeof(uint64_t) * 2 * g, hipMemcpyDeviceToHost));
            double sc = 0, sr = 0;
            for (int i = 0; i < g; ++i) {
                sc += (double)h[2 * i];
                sr += (double)(h[2 * i + 1] & 0xffffffffffffull);
            }
            mhz = sc / sr * 100.0;
            const double insts = (double)iters * REPT * 4;
            if (pass == 0) lone = (sc / g) / insts;
            else {
                const double cyc_total = (double)ms * 1e-3 * mhz * 1e6;
                res[0] = cyc_total * ncu / (insts * g);  // CU-cycles per instruction
            }
            CK(hipEventDestroy(e0));
            CK(hipEventDestroy(e1));
        }
        printf("%-20s simd %6.2f  cu %6.2f  lone %6.2f  %5.0f MHz   %s\n", op_name[p], res[0] * 4.0, res[0], lone, mhz, op_text[p]);
    }
    return 0;
}
