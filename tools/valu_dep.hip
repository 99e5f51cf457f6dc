// valu_dep.hip — how fast ONE wavefront can issue vector instructions that depend on each other, and how many
// wavefronts per SIMD it takes to fill the VALU with such code.  Every wave runs `iters` rounds of 64 FMAs arranged
// as ILP independent chains (ILP = 1: every FMA needs the previous one's result; 2, 4, 8: that many chains interleaved).
// N workgroups of 4 waves (one per SIMD) are made resident per CU by giving each 160/N KiB of LDS.
// Output: SIMD cycles per instruction = slowest wave's cycles / (instructions per wave x waves per SIMD).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_dep.hip -o tools/bin/valu_dep && tools/bin/valu_dep 5 2000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int ILP, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cycles, int iters, float seed) {
    extern __shared__ float lds_pad[];
    if (seed == -1.0f) lds_pad[threadIdx.x] = 0.0f;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.5f + 0.125f * j;
    float b = 1.0f + threadIdx.x * 1e-9f, c = 0.999f;
    unsigned m = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[r % ILP]) : "v"(b), "v"(c));          // 3 VGPR sources, result feeds the next of its chain
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, 2.0, %1" : "+v"(a[r % ILP]) : "v"(b));                  // 2 VGPR sources
            if (KIND == 2) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[r % ILP]) : "v"(b));             // the funnel shift
            if (KIND == 3) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[r % ILP]) : "v"(b));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = __uint_as_float(m & 1u);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    const size_t gid = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
    out[gid] = s;
    if ((threadIdx.x & 63) == 0) cycles[gid >> 6] = t1 - t0;
}

static int g_waves, g_iters, g_ncu;
static float* d_out;
static unsigned long long* d_cyc;
template <int ILP, int KIND>
static void run(const char* name) {
    const int lds = (160 * 1024 / g_waves) & ~1023;
    const int grid = g_ncu * g_waves;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<ILP, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, 16, 1.25f);
    hipLaunchKernelGGL((k<ILP, KIND>), dim3(grid), dim3(256), lds, 0, d_out, d_cyc, g_iters, 1.25f);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(static_cast<size_t>(grid) * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0, mn = 1e30;
    for (auto c : h) { mx = c > mx ? static_cast<double>(c) : mx; mn = c < mn ? static_cast<double>(c) : mn; }
    const double n = static_cast<double>(g_iters) * 64;
    std::printf("%-26s ILP %d, %d waves/SIMD: %.2f SIMD cycles per instruction (slowest wave), a wave issues every %.2f .. %.2f cycles\n", name, ILP, g_waves,
                mx / (n * g_waves), mn / n, mx / n);
}

int main(int argc, char** argv) {
    g_waves = argc > 1 ? std::atoi(argv[1]) : 5;
    g_iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    if (g_waves < 1 || g_waves > 8 || g_iters < 16 || g_iters > 100000) { std::fprintf(stderr, "usage: valu_dep [waves per SIMD 1..8] [rounds]\n"); return 2; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_ncu = prop.multiProcessorCount;
    CHECK(hipMalloc(&d_out, static_cast<size_t>(g_ncu) * g_waves * 256 * 4));
    CHECK(hipMalloc(&d_cyc, static_cast<size_t>(g_ncu) * g_waves * 4 * 8));
    run<1, 0>("v_fma_f32 (3 VGPR)"); run<2, 0>("v_fma_f32 (3 VGPR)"); run<4, 0>("v_fma_f32 (3 VGPR)"); run<8, 0>("v_fma_f32 (3 VGPR)");
    run<1, 1>("v_fma_f32 (2 VGPR, const)"); run<2, 1>("v_fma_f32 (2 VGPR, const)"); run<4, 1>("v_fma_f32 (2 VGPR, const)"); run<8, 1>("v_fma_f32 (2 VGPR, const)");
    run<1, 2>("v_alignbit_b32"); run<2, 2>("v_alignbit_b32"); run<4, 2>("v_alignbit_b32");
    run<1, 3>("v_mul_f32"); run<4, 3>("v_mul_f32");
    return 0;
}
