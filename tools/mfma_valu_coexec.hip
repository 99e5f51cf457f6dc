// mfma_valu_coexec.hip — do the matrix pipe and the vector ALU of one SIMD run side by side?  Role-split version
// (tools/mfma_valu_overlap.hip used self-dependent FMA chains that reach 2.45 cycles per FMA alone and so
// measured latency, not issue).  One workgroup of 16 waves per CU (LDS-limited to one): waves 0-7 issue only
// v_mfma_f32_32x32x16_f16 (four independent accumulator tiles), waves 8-15 only fp32 FMAs in the render kernel's form
// (v_fma_f32 v, v, s, v: 16 independent accumulators, multiplier in an SGPR); every SIMD holds two waves of each role.
// Three launches: matrix waves alone, vector waves alone, both.  Side by side: time(both) ~ max of the two; one after the
// other: ~ their sum.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coexec.hip -o tools/bin/mfma_valu_coexec && tools/bin/mfma_valu_coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMfmaPerIter = 8, kFmaPerIter = 128;      // per wave and iteration: 8 x 32 = 256 matrix cycles, 128 x 2 = 256 vector cycles

__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cycles, int iters, int run_mfma, int run_valu, float seed) {
    extern __shared__ float pad[];
    if (seed == -1.0f) pad[threadIdx.x] = 0.0f;
    const int wave = threadIdx.x >> 6;
    const bool matrix_role = wave < 8;
    unsigned long long t0 = 0, t1 = 0;
    float result = 0.0f;
    if (matrix_role) {
        if (run_mfma) {
            f16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = static_cast<_Float16>(seed + j); b[j] = static_cast<_Float16>(seed * 0.5f + (threadIdx.x & 7)); }
            f32x16 acc[4];
            for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < kMfmaPerIter; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
            }
            t1 = __builtin_readcyclecounter();
            for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) result += acc[t][j];
        }
    } else if (run_valu) {
        float v[16];
        for (int j = 0; j < 16; ++j) v[j] = 0.5f + 0.03125f * j;
        const float c = seed * 0.001f;
        float mult = __builtin_amdgcn_readfirstlane(__float_as_int(0.999f)) == 0 ? 1.0f : 0.999f;
        asm volatile("" : "+s"(mult));
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < kFmaPerIter; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r & 15]) : "s"(mult), "v"(c));
        }
        t1 = __builtin_readcyclecounter();
        for (int j = 0; j < 16; ++j) result += v[j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = result;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, lds = 96 * 1024;      // > half of the CU's LDS: one workgroup per CU
    float* d_out;
    unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, sizeof(float) * ncu * 1024));
    CHECK(hipMalloc(&d_cyc, 8 * ncu * 16));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    std::vector<unsigned long long> h(static_cast<size_t>(ncu) * 16);
    double per[3][2] = {};
    const char* names[3] = {"matrix waves alone", "vector waves alone", "both roles together"};
    for (int mode = 0; mode < 3; ++mode) {
        const int rm = mode != 1, rv = mode != 0;
        hipLaunchKernelGGL(k, dim3(ncu), dim3(1024), lds, 0, d_out, d_cyc, 64, rm, rv, 1.25f);          // warm-up
        hipLaunchKernelGGL(k, dim3(ncu), dim3(1024), lds, 0, d_out, d_cyc, iters, rm, rv, 1.25f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double mx[2] = {0, 0};
        for (size_t i = 0; i < h.size(); ++i) { const int role = (i % 16) < 8 ? 0 : 1; mx[role] = h[i] > mx[role] ? static_cast<double>(h[i]) : mx[role]; }
        per[mode][0] = mx[0] / iters; per[mode][1] = mx[1] / iters;
        // two waves of a role per SIMD: per iteration the SIMD's matrix pipe owes 2 x 8 x 32 = 512 cycles, its vector ALU 2 x 128 x 2 = 512
        std::printf("%-22s wave cycles per iteration (s_memtime ticks): matrix role %7.1f   vector role %7.1f\n", names[mode], per[mode][0], per[mode][1]);
    }
    std::printf("matrix pipe alone: %.2f ticks per MFMA and SIMD (2 waves x %d per iteration); vector ALU alone: %.2f ticks per FMA and SIMD (2 waves x %d)\n",
                per[0][0] / (2 * kMfmaPerIter), kMfmaPerIter, per[1][1] / (2 * kFmaPerIter), kFmaPerIter);
    std::printf("together: matrix role x%.2f, vector role x%.2f of their times alone (side by side: ~1.0 each; one after the other: ~2.0 each)\n",
                per[2][0] / per[0][0], per[2][1] / per[1][1]);
    return 0;
}
