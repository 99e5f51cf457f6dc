// valu_cycles.hip — cost IN SHADER CYCLES of the vector instructions the render kernel is (or could be) built from,
// measured with s_memtime inside the kernel so that DVFS (the clock the chip holds under each instruction mix) does not
// blur the answer the way wall-clock rates do (tools/valu_rates.hip).  Every wave runs `iters` rounds of 64 independent
// instructions of one kind over 8 accumulators; `waves` waves are resident per SIMD on every CU.
// Output per instruction: cycles per wave-instruction per SIMD (= wave's elapsed cycles / (instructions x waves per SIMD)),
// and the clock held meanwhile (cycles / wall time).  Memory: one float per thread written, nothing read.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_cycles.hip -o tools/bin/valu_cycles && tools/bin/valu_cycles 6 4000
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            std::exit(1);                                                                \
        }                                                                                \
    } while (0)

// X(name, statement using a[j] (float), p[j] (float2), m (unsigned), b, c (floats), sb (SGPR float), sm (64-bit SGPR pair))
#define OPS(X)                                                                                                              \
    X(v_fma_f32, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                                     \
    X(v_fmac_f32_e32, asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                               \
    X(v_fma_f32_sgpr, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "s"(sb), "v"(c)))                               \
    X(v_fma_f32_2vgpr, asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(a[j]) : "v"(b)))                                       \
    X(v_fma_f32_const, asm volatile("v_fma_f32 %0, %1, 2.0, %0" : "+v"(a[j]) : "v"(b)))                                      \
    X(v_fmac_f32_sgpr, asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[j]) : "s"(sb), "v"(c)))                             \
    X(v_fma_f32_dst_new, asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[j]) : "v"(b), "v"(c), "v"(tmp)))                   \
    X(v_pk_fma_f16_sgpr, asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[j]) : "s"(sb), "v"(c)))                         \
    X(v_pk_fma_f16_2vgpr, asm volatile("v_pk_fma_f16 %0, %1, %1, %0" : "+v"(a[j]) : "v"(b)))                                 \
    X(v_alignbit_sgpr, asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[j]) : "s"(sb)))                                 \
    X(v_mul_f32, asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_add_f32, asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_sub_f32, asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_max_f32, asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(pb), "v"(pc)))                             \
    X(v_pk_mul_f32, asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[j]) : "v"(pb)))                                          \
    X(v_pk_add_f32, asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[j]) : "v"(pb)))                                          \
    X(v_pk_fma_f16, asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                               \
    X(v_pk_mul_f16, asm volatile("v_pk_mul_f16 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                           \
    X(v_pk_add_f16, asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                           \
    X(v_pk_max_f16, asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                           \
    X(v_fma_f16, asm volatile("v_fma_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                                     \
    X(v_fma_mix_f32, asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(a[j]) : "v"(b), "v"(c)))           \
    X(v_dot2_f32_f16, asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                           \
    X(v_dot2c_f32_f16, asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                             \
    X(v_dot2_f32_bf16, asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                         \
    X(v_dot4_i32_i8, asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                             \
    X(v_alignbit_b32, asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[j]) : "v"(b)))                                   \
    X(v_lshl_or_b32, asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[j]) : "v"(b)))                                      \
    X(v_and_or_b32, asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                               \
    X(v_lshrrev_b32, asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(a[j])))                                               \
    X(v_lshlrev_b32, asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[j])))                                               \
    X(v_or_b32, asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                               \
    X(v_and_b32, asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_xor_b32, asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_or3_b32, asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                                     \
    X(v_bfi_b32, asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c)))                                     \
    X(v_perm_b32, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                                   \
    X(v_add_u32, asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                             \
    X(v_add3_u32, asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                                   \
    X(v_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[j]) : "v"(b)))                                    \
    X(v_mov_b32, asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[j]) : "v"(b)))                                                 \
    X(v_cndmask_b32_vcc, asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[j]) : "v"(b) : "vcc"))                    \
    X(v_cmp_lt_f32_vcc, asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc"))                          \
    X(v_cmp_lt_f32_sgpr, asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm[j]) : "v"(a[j]), "v"(b)))                      \
    X(v_cmp_gt_u32_vcc, asm volatile("v_cmp_gt_u32_e32 vcc, %0, %1" : : "v"(a[j]), "v"(b) : "vcc"))                          \
    X(v_cmpx_lt_f32, asm volatile("v_cmp_lt_f32_e64 %0, %1, %1" : "=s"(sm[j]) : "v"(a[j])))                                  \
    X(v_med3_f32, asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                                   \
    X(v_min3_f32, asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                                   \
    X(v_mad_u64_u32, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[j]) : "v"(b), "v"(c) : "vcc"))                \
    X(v_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))                                           \
    X(v_mul_hi_u32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))                                           \
    X(v_mul_u32_u24, asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))                                     \
    X(v_mad_u32_u24, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c)))                             \
    X(v_rcp_f32, asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(a[j])))                                                          \
    X(v_rsq_f32, asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(a[j])))                                                          \
    X(v_sqrt_f32, asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(a[j])))                                                        \
    X(v_exp_f32, asm volatile("v_exp_f32_e32 %0, %0" : "+v"(a[j])))                                                          \
    X(v_sin_f32, asm volatile("v_sin_f32_e32 %0, %0" : "+v"(a[j])))                                                          \
    X(v_cvt_f32_u32, asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(a[j])))                                                  \
    X(v_cvt_f16_f32, asm volatile("v_cvt_f16_f32_e32 %0, %0" : "+v"(a[j])))                                                  \
    X(v_cvt_pk_f16_f32, asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))                                \
    X(v_ffbh_u32, asm volatile("v_ffbh_u32_e32 %0, %0" : "+v"(a[j])))                                                        \
    X(v_bcnt_u32, asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))                                           \
    X(v_mbcnt_lo, asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[j]) : "v"(b)))                                       \
    X(v_readlane, asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sl[j]) : "v"(a[j])))                                        \
    X(v_readfirstlane, asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sl[j]) : "v"(a[j])))                                 \
    X(v_mov_dpp_rowshr, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[j])))              \
    X(v_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[j]) : "v"(db)))                                                \
    X(v_fma_f64, asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[j]) : "v"(db)))                                            \
    X(s_nop_0, asm volatile("s_nop 0"))                                                                                      \
    X(v_fma_f32_clamp, asm volatile("v_fma_f32 %0, -%1, %2, %0 clamp" : "+v"(a[j]) : "v"(b), "v"(c)))                         \
    X(v_fma_f32_double_add, asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a[j]) : "v"(b)))                                 \
    X(v_cvt_u32_f32, asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(a[j])))                                                  \
    X(v_cvt_pk_f16_f32_rtn, asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a[j]) : "v"(b)))                               \
    X(v_cvt_f32_f16, asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(a[j])))                                                  \
    X(v_cvt_f32_f16_sdwa_hi, asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[j])))   \
    X(v_perm_b32_sgpr_sel, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(sb)))                         \
    X(v_pk_mul_f32_sgpr, asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[j]) : "s"(sm[0])))                                  \
    X(v_mov_b64, asm volatile("v_mov_b64_e32 %0, %1" : "+v"(p[j]) : "v"(pb)))                                                \
    X(v_permlane32_swap, asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(a[j]), "+v"(a[(j + 4) & 7])))                \
    X(v_cndmask_b32_sgpr_mask, asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "s"(sm[0])))           \
    X(v_addc_co_u32, asm volatile("v_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(a[j]) : : "vcc"))                            \
    X(v_fmamk_f32, asm volatile("v_fmamk_f32 %0, %0, 0x3f7fff3f, %1" : "+v"(a[j]) : "v"(b)))                                 \
    X(v_med3_f32_sgpr, asm volatile("v_med3_f32 %0, %0, -%1, %1" : "+v"(a[j]) : "s"(sb)))                                    \
    X(v_subrev_f32_sgpr, asm volatile("v_subrev_f32_e32 %0, %1, %0" : "+v"(a[j]) : "s"(sb)))                                 \
    X(v_lshl_or_b32_imm16, asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[j]) : "v"(b)))                               \
    X(mix_class_7, MIXC7(j))                                                                                                 \
    X(mix_class_clamp_7, MIXCC(j))                                                                                           \
    X(mix_class_sgpr_7, MIXCS(j))                                                                                            \
    X(mix_class_sgpr_alignbit_7, MIXCSA(j))                                                                                  \
    X(mix_form1_9, MIXF1(j))                                                                                                 \
    X(mix_7fma_cmp, MIX7C(j))                                                                                                \
    X(mix_f16pair_10, MIXF2(j))

// one class-word filter test as shipped in r02-c: 6 FMA + alignbit (7 instructions)
#define MIXC7(j)                                                                                                                   \
    asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %3, %2, %1\n v_fma_f32 %0, %3, %3, %0\n v_fma_f32 %1, %2, %2, %1\n"     \
                 "v_fma_f32 %0, %0, %0, %3\n v_fma_f32 %0, %1, %1, %0\n"                                                           \
                 "v_alignbit_b32 %4, %4, %0, 31"                                                                                   \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c), "+v"(m))
// candidate: the sign collected by a clamped FMA (exactly 0 or 1) and a doubling FMA instead of the funnel shift (7 FMA-rate instructions)
#define MIXCC(j)                                                                                                                   \
    asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %3, %2, %1\n v_fma_f32 %0, %3, %3, %0\n v_fma_f32 %1, %2, %2, %1\n"     \
                 "v_fma_f32 %0, -%0, %0, %3\n v_fma_f32 %0, -%1, %1, %0 clamp\n"                                                   \
                 "v_fma_f32 %4, %4, 2.0, %0"                                                                                       \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c), "+v"(tmp))
// the clamp form with the sphere's operands in SGPRs (a different pair for every test), as the scalar-load filter issues it
#define MIXCS(j)                                                                                                                   \
    asm volatile("v_fma_f32 %0, %4, %2, %3\n v_fmac_f32_e32 %0, %5, %3\n v_fma_f32 %1, %4, %3, %2\n v_fmac_f32_e32 %1, %5, %2\n"     \
                 "v_fma_f32 %0, -%0, %0, %6\n v_fma_f32 %0, -%1, %1, %0 clamp\n"                                                   \
                 "v_fmac_f32_e32 %7, 2.0, %0"                                                                                      \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c) : "s"(sv[2 * j]), "s"(sv[2 * j + 1]), "s"(sb), "v"(tmp))
#define MIXCSA(j)                                                                                                                  \
    asm volatile("v_fma_f32 %0, %4, %2, %3\n v_fmac_f32_e32 %0, %5, %3\n v_fma_f32 %1, %4, %3, %2\n v_fmac_f32_e32 %1, %5, %2\n"     \
                 "v_fma_f32 %0, -%0, %0, %6\n v_fma_f32 %0, %1, %1, -%0\n"                                                         \
                 "v_alignbit_b32 %7, %7, %0, 31"                                                                                   \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c) : "s"(sv[2 * j]), "s"(sv[2 * j + 1]), "s"(sb), "v"(m))
// one filter test, form 1 as shipped in round 1: 8 FMA + alignbit (9 instructions)
#define MIXF1(j)                                                                                                                   \
    asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %3, %2, %1\n v_fma_f32 %0, %3, %3, %0\n v_fma_f32 %1, %2, %2, %1\n"     \
                 "v_fma_f32 %0, %2, %2, %0\n v_fma_f32 %1, %3, %3, %1\n v_fma_f32 %0, %0, %0, %3\n v_fma_f32 %0, %1, %1, %0\n"     \
                 "v_alignbit_b32 %4, %4, %0, 31"                                                                                   \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c), "+v"(m))
// candidate: 7 FMA + one compare into an SGPR pair (per-sphere lane mask), 8 instructions
#define MIX7C(j)                                                                                                                   \
    asm volatile("v_fma_f32 %0, %2, %3, %0\n v_fma_f32 %1, %3, %2, %1\n v_fma_f32 %0, %3, %3, %0\n v_fma_f32 %1, %2, %2, %1\n"     \
                 "v_fma_f32 %0, %2, %2, %0\n v_fma_f32 %1, %3, %3, %1\n v_fma_f32 %0, %1, %1, %0\n v_cmp_lt_f32_e64 %4, %0, %3"    \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c), "=s"(sm[j]))
// one PAIR of filter tests, packed f16: 8 pk_fma_f16 + shift + and_or (10 instructions)
#define MIXF2(j)                                                                                                                   \
    asm volatile("v_pk_fma_f16 %0, %2, %3, %0\n v_pk_fma_f16 %1, %3, %2, %1\n v_pk_fma_f16 %0, %3, %3, %0\n v_pk_fma_f16 %1, %2, %2, %1\n" \
                 "v_pk_fma_f16 %0, %2, %2, %0\n v_pk_fma_f16 %1, %3, %3, %1\n v_pk_fma_f16 %0, %0, %0, %3\n v_pk_fma_f16 %0, %1, %1, %0\n" \
                 "v_lshrrev_b32 %5, 3, %0\n v_and_or_b32 %4, %5, %2, %4"                                                           \
                 : "+v"(a[j]), "+v"(a[(j + 4) & 7]), "+v"(b), "+v"(c), "+v"(m), "=v"(tmp))

enum Op {
#define X(name, stmt) OP_##name,
    OPS(X)
#undef X
        N_OPS
};
static const char* kNames[N_OPS] = {
#define X(name, stmt) #name,
    OPS(X)
#undef X
};
static int instr_per_body(int op) { return op == OP_mix_form1_9 ? 9 : op == OP_mix_class_7 ? 7 : op == OP_mix_class_clamp_7 ? 7 : op == OP_mix_class_sgpr_7 ? 7 : op == OP_mix_class_sgpr_alignbit_7 ? 7 : op == OP_mix_7fma_cmp ? 8 : op == OP_mix_f16pair_10 ? 10 : 1; }

template <int OP>
__global__ __launch_bounds__(1024) void cyc_kernel(float* out, unsigned long long* cycles, int iters, float seed_s) {
    extern __shared__ float lds_pad[];      // sized by the host so that exactly the wanted number of blocks fits on a CU
    if (seed_s == -1.0f) lds_pad[threadIdx.x] = 0.0f;   // never true: keeps the allocation referenced
    float a[8];
    f2 p[8];
    double d[8];
    unsigned long long sm[8];
    unsigned sl[8];
    float b = 1.0f + threadIdx.x * 1e-9f, c = 0.999f, tmp = 0.0f;
    const f2 pb = {b, b}, pc = {c, c};
    const double db = 1.0 + threadIdx.x * 1e-12;
    const float sb = __builtin_amdgcn_readfirstlane(seed_s);
    float sv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) sv[j] = __builtin_amdgcn_readfirstlane(seed_s + 0.01f * j);
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = j * 0.125f + 0.5f;
        p[j] = f2{j * 0.125f, j * 0.25f};
        d[j] = j * 0.125;
        sm[j] = 0;
        sl[j] = 0;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#define X(name, stmt) \
    if (OP == OP_##name) { stmt; }
                OPS(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = __uint_as_float(m & 1u) + tmp * 0.0f + b * 0.0f + c * 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y + static_cast<float>(d[j]) + static_cast<float>(sm[j] & 1ull) + static_cast<float>(sl[j] & 1u);
    const size_t gid = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t1 - t0;
}

// What the clamp output modifier makes of the values a filter test can produce (kernels run with DX10_CLAMP set: NaN -> 0).
__global__ void clamp_semantics(const float* in, float* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        float r;
        asm volatile("v_fma_f32 %0, %1, 1.0, 0 clamp" : "=v"(r) : "v"(in[i]));
        out[i] = r;
    }
}

static int g_threads = 256, g_lds = 0;

template <int OP>
static void run(float* d_out, unsigned long long* d_cyc, std::vector<unsigned long long>& h_cyc, int grid, int iters, int waves) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cyc_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, g_lds));
    hipLaunchKernelGGL(cyc_kernel<OP>, dim3(grid), dim3(g_threads), g_lds, 0, d_out, d_cyc, iters / 8 + 1, 1.25f);   // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(cyc_kernel<OP>, dim3(grid), dim3(g_threads), g_lds, 0, d_out, d_cyc, iters, 1.25f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(h_cyc.data(), d_cyc, h_cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0, mx = 0, mn = 1e30;
    for (unsigned long long v : h_cyc) { sum += static_cast<double>(v); mx = v > mx ? static_cast<double>(v) : mx; mn = v < mn ? static_cast<double>(v) : mn; }
    const double avg = sum / h_cyc.size();
    const double n_instr = static_cast<double>(iters) * 64 * instr_per_body(OP);
    std::printf("%-22s %8.3f cycles per wave-instruction per SIMD   (%6.2f per body of %2d)   wave min %9.0f avg %9.0f max %9.0f cycles, %7.3f ms -> %5.2f GHz\n", kNames[OP],
                avg / (n_instr * waves), avg / (static_cast<double>(iters) * 64 * waves), instr_per_body(OP), mn, avg, mx, ms, mx / (ms * 1e6));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? std::atoi(argv[1]) : 4;     // resident waves per SIMD
    const int iters = argc > 2 ? std::atoi(argv[2]) : 4000;
    if (waves < 1 || waves > 8 || iters < 8 || iters > 200000) {
        std::fprintf(stderr, "usage: valu_cycles [waves per SIMD 1..8] [iterations 8..200000]\n");
        return 2;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    // placement is forced through LDS: up to 4 waves per SIMD = ONE block of 256 x waves threads per CU (96 KiB of LDS: a
    // second block does not fit); 6 or 8 = TWO blocks of 768 or 1024 threads (64 KiB each: a third does not fit)
    if (waves == 5 || waves == 7) { std::fprintf(stderr, "waves per SIMD: 1, 2, 3, 4, 6 or 8\n"); return 2; }
    const int blocks_per_cu = waves > 4 ? 2 : 1;
    g_threads = 256 * waves / blocks_per_cu;
    g_lds = blocks_per_cu == 1 ? 96 * 1024 : 64 * 1024;
    const int grid = n_cu * blocks_per_cu;
    const size_t n_threads = static_cast<size_t>(grid) * g_threads;
    {
        const float in[10] = {__builtin_nanf(""), __builtin_inff(), -__builtin_inff(), 0.5f, -0.0f, 2.0f, 1e-40f, -1e-40f, 1.0f, 0x1p-126f};
        float *d_in = nullptr, *d_res = nullptr, res[10];
        CHECK(hipMalloc(&d_in, sizeof(in)));
        CHECK(hipMalloc(&d_res, sizeof(in)));
        CHECK(hipMemcpy(d_in, in, sizeof(in), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(clamp_semantics, dim3(1), dim3(64), 0, 0, d_in, d_res, 10);
        CHECK(hipMemcpy(res, d_res, sizeof(in), hipMemcpyDeviceToHost));
        std::printf("clamp(x):");
        for (int i = 0; i < 10; ++i) std::printf("  %g -> %g", in[i], res[i]);
        std::printf("\n");
        CHECK(hipFree(d_in));
        CHECK(hipFree(d_res));
    }
    float* d_out = nullptr;
    unsigned long long* d_cyc = nullptr;
    CHECK(hipMalloc(&d_out, n_threads * sizeof(float)));
    CHECK(hipMalloc(&d_cyc, n_threads / 64 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h_cyc(n_threads / 64);
    std::printf("%s: %d CUs, %d wave(s) per SIMD, %d rounds of 64 bodies\n", prop.name, n_cu, waves, iters);
#define X(name, stmt) run<OP_##name>(d_out, d_cyc, h_cyc, grid, iters, waves);
    OPS(X)
#undef X
    CHECK(hipFree(d_out));
    CHECK(hipFree(d_cyc));
    return 0;
}
