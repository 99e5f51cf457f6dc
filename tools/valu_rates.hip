// valu_rates.hip — issue rate of the vector instructions the filter could be built from, on the GPU it runs on.
// Decides between the filter forms of DESIGN.md §7 (f32 FMA chain vs f16 dot products) BEFORE any of them is written
// into the kernel.  Every wave runs `iters` rounds of 64 instructions of one kind on 8 independent accumulators;
// 256 CUs x 4 SIMDs x `waves` waves are resident.  Output: ns per wave-instruction per SIMD, and the same relative to
// v_fma_f32.  Memory: one float per thread is written to a buffer of exactly grid x block floats; nothing is read.
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
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

enum Op { FMA_F32, FMA_F32_SGPR, PK_FMA_F32, PK_FMA_F16, DOT2_F32_F16, DOT2C_F32_F16, DOT2_F32_BF16, ALIGNBIT, MIX_FORM1, MIX_FORM2, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_fma_f32 (SGPR src)", "v_pk_fma_f32", "v_pk_fma_f16", "v_dot2_f32_f16", "v_dot2c_f32_f16",
                                    "v_dot2_f32_bf16", "v_alignbit_b32", "form 1 body: 8 fma + alignbit (per sphere)", "form 2 body: 8 pk_fma_f16 + shift + and (per PAIR)",
                                    "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32 (Philox product)"};
static const int kInstrPerRound[N_OPS] = {64, 64, 64, 64, 64, 64, 64, 64, 72, 80, 64, 64, 64};

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed_s) {
    float a[8];
    f2 p[8];
    const float b = 1.0f + threadIdx.x * 1e-9f, c = 0.999f;
    const f2 pb = {b, b}, pc = {c, c};
    const float sb = __builtin_amdgcn_readfirstlane(seed_s);   // a value that lives in an SGPR
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = j * 0.125f;
        p[j] = f2{j * 0.125f, j * 0.25f};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == FMA_F32_SGPR) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[j]) : "s"(sb), "v"(c));
                if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[j]) : "v"(pb), "v"(pc));
                if (OP == PK_FMA_F16) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == DOT2_F32_F16) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == DOT2C_F32_F16) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == DOT2_F32_BF16) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(a[j]) : "v"(b), "v"(c));
                if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m) : "v"(a[j]));
                if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[j]) : "v"(b));
                if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(p[j]) : "v"(b), "v"(c) : "vcc");
            }
        }
        if (OP == MIX_FORM1) {
            // the shipped filter body for 8 spheres: 3 + 3 + 2 FMAs and one funnel shift each
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p1 = a[j], p2 = a[(j + 1) & 7], d;
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(c), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p1) : "v"(b), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(c), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p2) : "v"(b), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(d) : "v"(p1), "v"(c));
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(p2));
                asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(m) : "v"(d));
            }
        }
        if (OP == MIX_FORM2) {
            // the packed-f16 filter body (RTK_FILTER_FORM=2) for 8 PAIRS = 16 spheres, four pairs stage by stage
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* x = a;          // the eight accumulators stand in for four (p1, p2) pairs of chains
                float* y = a + 4;
#pragma unroll
                for (int st = 0; st < 3; ++st) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(x[j]) : "v"(b), "v"(c));
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(y[j]) : "v"(c), "v"(b));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_pk_fma_f16 %0, %0, %0, %1" : "+v"(x[j]) : "v"(c));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_pk_fma_f16 %0, %1, %1, %0" : "+v"(x[j]) : "v"(y[j]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned t;
                    asm volatile("v_lshrrev_b32 %0, 3, %1" : "=v"(t) : "v"(x[j]));
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(m) : "v"(t), "v"(0x10001000u));
                }
            }
        }
    }
    float s = __uint_as_float(m & 1u);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j] + p[j].x + p[j].y;
    out[blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x] = s;
}

template <int OP>
static double run(float* d_out, int grid, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters / 8, 1.25f);   // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(grid), dim3(256), 0, 0, d_out, iters, 1.25f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms;
}

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? std::atoi(argv[1]) : 4;     // resident waves per SIMD
    const int iters = argc > 2 ? std::atoi(argv[2]) : 20000;
    if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 8 || iters > 1000000) {
        std::fprintf(stderr, "usage: valu_rates [waves per SIMD 1..8] [iterations 8..1000000]\n");
        return 2;
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int grid = n_cu * waves_per_simd;     // one 256-thread block = one wave on each of the CU's 4 SIMDs
    float* d_out = nullptr;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(grid) * 256 * sizeof(float)));
    double ms[N_OPS];
    ms[FMA_F32] = run<FMA_F32>(d_out, grid, iters);
    ms[FMA_F32_SGPR] = run<FMA_F32_SGPR>(d_out, grid, iters);
    ms[PK_FMA_F32] = run<PK_FMA_F32>(d_out, grid, iters);
    ms[PK_FMA_F16] = run<PK_FMA_F16>(d_out, grid, iters);
    ms[DOT2_F32_F16] = run<DOT2_F32_F16>(d_out, grid, iters);
    ms[DOT2C_F32_F16] = run<DOT2C_F32_F16>(d_out, grid, iters);
    ms[DOT2_F32_BF16] = run<DOT2_F32_BF16>(d_out, grid, iters);
    ms[ALIGNBIT] = run<ALIGNBIT>(d_out, grid, iters);
    ms[MIX_FORM1] = run<MIX_FORM1>(d_out, grid, iters);
    ms[MIX_FORM2] = run<MIX_FORM2>(d_out, grid, iters);
    ms[MUL_LO_U32] = run<MUL_LO_U32>(d_out, grid, iters);
    ms[MUL_HI_U32] = run<MUL_HI_U32>(d_out, grid, iters);
    ms[MAD_U64_U32] = run<MAD_U64_U32>(d_out, grid, iters);
    std::printf("%s: %d CUs, %d wave(s) per SIMD, %d rounds\n", prop.name, n_cu, waves_per_simd, iters);
    const double base = ms[FMA_F32] * 1e6 / (static_cast<double>(iters) * 64 * waves_per_simd);
    for (int op = 0; op < N_OPS; ++op) {
        const double ns = ms[op] * 1e6 / (static_cast<double>(iters) * kInstrPerRound[op] * waves_per_simd);
        std::printf("%-42s %8.3f ms   %7.4f ns per wave-instruction per SIMD   x%.2f of v_fma_f32%s\n", kNames[op], ms[op], ns, ns / base,
                    op == MIX_FORM1 ? "  (one round = 8 spheres)" : op == MIX_FORM2 ? "  (one round = 16 spheres)" : "");
    }
    CHECK(hipFree(d_out));
    return 0;
}
