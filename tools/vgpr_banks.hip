// vgpr_banks.hip — does a VALU instruction pay for reading two source VGPRs from the same register bank (number mod 4)?
// 64 independent FMAs per round with explicit registers: sources from three different banks, two in one bank (each
// pairing), all three in one bank, the same register twice.  N workgroups of 4 waves per CU (LDS-limited).
//   hipcc --offload-arch=gfx950 -O2 tools/vgpr_banks.hip -o tools/bin/vgpr_banks && tools/bin/vgpr_banks 5 2000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define CLOB "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23"
#define X8(A, B, C)                                                                                               \
    asm volatile("v_fma_f32 v16, " A ", " B ", " C "\n v_fma_f32 v17, " A ", " B ", " C "\n v_fma_f32 v18, " A ", " B ", " C "\n v_fma_f32 v19, " A ", " B ", " C "\n" \
                 "v_fma_f32 v20, " A ", " B ", " C "\n v_fma_f32 v21, " A ", " B ", " C "\n v_fma_f32 v22, " A ", " B ", " C "\n v_fma_f32 v23, " A ", " B ", " C ::: CLOB)

#define P8(INS, TAIL)                                                                                             \
    asm volatile(INS " v[16:17], " TAIL "\n" INS " v[18:19], " TAIL "\n" INS " v[20:21], " TAIL "\n" INS " v[22:23], " TAIL "\n" \
                 INS " v[16:17], " TAIL "\n" INS " v[18:19], " TAIL "\n" INS " v[20:21], " TAIL "\n" INS " v[22:23], " TAIL ::: CLOB)
#define S8(INS, TAIL)                                                                                             \
    asm volatile(INS " v16, " TAIL "\n" INS " v17, " TAIL "\n" INS " v18, " TAIL "\n" INS " v19, " TAIL "\n"       \
                 INS " v20, " TAIL "\n" INS " v21, " TAIL "\n" INS " v22, " TAIL "\n" INS " v23, " TAIL ::: CLOB)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cycles, int iters, float seed) {
    extern __shared__ float lds_pad[];
    if (seed == -1.0f) lds_pad[threadIdx.x] = 0.0f;
    asm volatile("v_mov_b32 v1, 1.0\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 2.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 2.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5" ::: CLOB);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) X8("v1", "v2", "v3");      // banks 1, 2, 3
            if (KIND == 1) X8("v1", "v5", "v3");      // src0 and src1 in bank 1
            if (KIND == 2) X8("v1", "v2", "v5");      // src0 and src2 in bank 1
            if (KIND == 3) X8("v2", "v1", "v5");      // src1 and src2 in bank 1
            if (KIND == 4) X8("v1", "v5", "v9");      // all three in bank 1
            if (KIND == 5) X8("v1", "v1", "v3");      // the same register twice
            if (KIND == 6) X8("v1", "v1", "v5");      // the same register twice + another of its bank
            if (KIND == 7) X8("v1", "v2", "2.0");     // two VGPRs, different banks, and a constant
            if (KIND == 8) X8("v1", "v5", "2.0");     // two VGPRs in one bank and a constant
            if (KIND == 9) P8("v_pk_fma_f32", "v[2:3], v[2:3], v[4:5]");      // packed: the same pair twice
            if (KIND == 10) P8("v_pk_fma_f32", "v[2:3], v[6:7], v[4:5]");     // packed: pairs in the same banks
            if (KIND == 11) P8("v_pk_fma_f32", "v[2:3], v[8:9], v[4:5]");     // packed: pairs in different banks
            if (KIND == 12) S8("v_mul_f32_e32", "v1, v1");
            if (KIND == 13) S8("v_mul_f32_e32", "v1, v2");
            if (KIND == 14) S8("v_mul_f32_dpp", "v1, v1 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf");
            if (KIND == 15) S8("v_mul_f32_sdwa", "v1, v1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD");
            if (KIND == 16) S8("v_add_f32_e64", "v1, v2 clamp");
            if (KIND == 17) S8("v_alignbit_b32", "v1, v2, 31");
            if (KIND == 18) S8("v_alignbit_b32", "v1, v5, 31");
            if (KIND == 19) S8("v_fma_f32", "v1, v5, v3 clamp");
            if (KIND == 20) S8("v_fmac_f32_e32", "v1, v1");
            if (KIND == 21) S8("v_fmac_f32_e32", "v1, v5");
            if (KIND == 22) S8("v_fmac_f32_e32", "v1, v2");
            if (KIND == 23) S8("v_mul_f32_e32", "v1, v5");
            if (KIND == 24) S8("v_mul_f32_e64", "v1, v1");
            if (KIND == 25) S8("v_mul_f32_e64", "v1, v5");
            if (KIND == 26) S8("v_mul_f32_e64", "-v1, v1 clamp");
            if (KIND == 27) S8("v_fmac_f32_e64", "v1, v1");
            if (KIND == 29) S8("v_subrev_f32_e32", "v1, v5");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s;
    asm volatile("v_add_f32 %0, v16, v23" : "=v"(s)::CLOB);
    const size_t gid = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t1 - t0;
}

// What `v_pk_fma_f32 ... clamp` with negated first operand makes of both halves: out = clamp(c - a*a), NaN -> 0.
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void pk_clamp_semantics(const float* a, const float* c, float* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        f2 p = {a[2 * i], a[2 * i + 1]}, q = {c[2 * i], c[2 * i + 1]}, r;
        asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(r) : "v"(p), "v"(q));
        out[2 * i] = r.x;
        out[2 * i + 1] = r.y;
    }
}

static int g_waves, g_iters, g_ncu;
static float* d_out;
static unsigned long long* d_cyc;
template <int KIND>
static void run(const char* name) {
    const int lds = (160 * 1024 / g_waves) & ~1023, grid = g_ncu * g_waves;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, 16, 1.25f);
    hipLaunchKernelGGL((k<KIND>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, g_iters, 1.25f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(static_cast<size_t>(grid) * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (auto c : h) mx = c > mx ? static_cast<double>(c) : mx;
    std::printf("%-58s %d waves/SIMD: %.2f SIMD cycles per FMA\n", name, g_waves, mx / (static_cast<double>(g_iters) * 64 * g_waves));
}

int main(int argc, char** argv) {
    g_waves = argc > 1 ? std::atoi(argv[1]) : 5;
    g_iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    if (g_waves < 1 || g_waves > 8 || g_iters < 16 || g_iters > 100000) { std::fprintf(stderr, "usage: vgpr_banks [waves per SIMD 1..8] [rounds]\n"); return 2; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_ncu = prop.multiProcessorCount;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(g_ncu) * g_waves * 256 * 4));
    CHECK(hipMalloc(&d_cyc, static_cast<size_t>(g_ncu) * g_waves * 4 * 8));
    {
        const float inf = __builtin_inff(), nan = __builtin_nanf("");
        const float a[12] = {1.0f, 2.0f, 0.5f, inf, 3.0f, 0.0f, nan, 1.0f, 0x1p70f, 1.0f, 0.0f, 0.0f};
        const float c[12] = {3.0f, 3.0f, 0.75f, inf, inf, -0.0f, 1.0f, nan, 0x1p100f, -inf, 0x1p-100f, 1.0f};
        float *da, *dc, *dr, r[12];
        CHECK(hipMalloc(&da, sizeof(a))); CHECK(hipMalloc(&dc, sizeof(c))); CHECK(hipMalloc(&dr, sizeof(a)));
        CHECK(hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, c, sizeof(c), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(pk_clamp_semantics, dim3(1), dim3(64), 0, 0, da, dc, dr, 6);
        CHECK(hipMemcpy(r, dr, sizeof(r), hipMemcpyDeviceToHost));
        std::printf("v_pk_fma_f32 clamp(c - a*a):");
        for (int i = 0; i < 12; ++i) std::printf("  (a %g, c %g) -> %g", a[i], c[i], r[i]);
        std::printf("\n");
    }
    run<0>("v_fma v1, v2, v3   (banks 1, 2, 3)");
    run<1>("v_fma v1, v5, v3   (src0, src1 share a bank)");
    run<2>("v_fma v1, v2, v5   (src0, src2 share a bank)");
    run<3>("v_fma v2, v1, v5   (src1, src2 share a bank)");
    run<4>("v_fma v1, v5, v9   (all three in one bank)");
    run<5>("v_fma v1, v1, v3   (same register twice)");
    run<6>("v_fma v1, v1, v5   (same register twice + its bank)");
    run<7>("v_fma v1, v2, 2.0  (two banks + constant)");
    run<8>("v_fma v1, v5, 2.0  (one bank + constant)");
    run<9>("v_pk_fma_f32 v[2:3], v[2:3], v[4:5]  (same pair twice)");
    run<10>("v_pk_fma_f32 v[2:3], v[6:7], v[4:5] (same banks)");
    run<11>("v_pk_fma_f32 v[2:3], v[8:9], v[4:5] (different banks)");
    run<12>("v_mul_f32 v1, v1");
    run<13>("v_mul_f32 v1, v2");
    run<14>("v_mul_f32_dpp v1, v1 (identity quad_perm)");
    run<15>("v_mul_f32_sdwa v1, v1");
    run<16>("v_add_f32 v1, v2 clamp");
    run<17>("v_alignbit_b32 v1, v2, 31");
    run<18>("v_alignbit_b32 v1, v5, 31 (one bank)");
    run<19>("v_fma v1, v5, v3 clamp (src0, src1 share a bank)");
    run<20>("v_fmac_f32_e32 v1, v1");
    run<21>("v_fmac_f32_e32 v1, v5 (one bank)");
    run<22>("v_fmac_f32_e32 v1, v2");
    run<23>("v_mul_f32_e32 v1, v5 (one bank)");
    run<24>("v_mul_f32_e64 v1, v1");
    run<25>("v_mul_f32_e64 v1, v5 (one bank)");
    run<26>("v_mul_f32_e64 -v1, v1 clamp");
    run<27>("v_fmac_f32_e64 v1, v1");
    run<29>("v_subrev_f32_e32 v1, v5 (one bank)");
    return 0;
}
