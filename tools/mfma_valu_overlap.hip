// mfma_valu_overlap.hip — do v_mfma_f32_32x32x16_bf16 and fp32 vector FMAs of the waves of one SIMD run side by side?
// Every wave repeats a body of NM matrix instructions (independent accumulators) and NV vector FMAs (8 independent
// chains, two-VGPR-plus-constant form: 2.0 cycles each on their own); 4 waves per SIMD (LDS-limited), all CUs.
// Output: SIMD cycles per body.  Side by side: max(32 NM, 2 NV); one after the other: the sum.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cycles, int iters, float seed) {
    extern __shared__ float lds_pad[];
    if (seed == -1.0f) lds_pad[threadIdx.x] = 0.0f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = static_cast<__bf16>(seed + j); b[j] = static_cast<__bf16>(seed * 0.5f + threadIdx.x); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.0f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.125f * j;
    const float c = seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NV; ++r) asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(v[r & 7]) : "v"(c));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static int g_ncu, g_waves = 4, g_iters = 4000;
static float* d_out;
static unsigned long long* d_cyc;
template <int NM, int NV>
static void run() {
    const int lds = (160 * 1024 / g_waves) & ~1023, grid = g_ncu * g_waves;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NM, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k<NM, NV>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, 16, 1.25f);
    hipLaunchKernelGGL((k<NM, NV>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, g_iters, 1.25f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(static_cast<size_t>(grid) * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (auto c : h) mx = c > mx ? static_cast<double>(c) : mx;
    std::printf("%d MFMA + %3d FMA per body, %d waves/SIMD: %6.1f SIMD cycles per body (side by side %d, one after the other %d)\n", NM, NV, g_waves,
                mx / (static_cast<double>(g_iters) * g_waves), 32 * NM > 2 * NV ? 32 * NM : 2 * NV, 32 * NM + 2 * NV);
}

int main(int argc, char** argv) {
    if (argc > 1) g_waves = std::atoi(argv[1]);
    if (g_waves < 1 || g_waves > 8) return 2;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_ncu = prop.multiProcessorCount;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(g_ncu) * g_waves * 256 * 4));
    CHECK(hipMalloc(&d_cyc, static_cast<size_t>(g_ncu) * g_waves * 4 * 8));
    run<4, 0>(); run<8, 0>(); run<0, 96>(); run<0, 192>(); run<4, 96>(); run<4, 192>(); run<8, 96>(); run<8, 192>(); run<2, 192>();
    return 0;
}
