// mfma_filter_probe.hip — what one word of the filter would cost if the two projections of every (ray, sphere) pair came
// from the matrix pipe: per word and wave one 16-byte A fragment per lane from LDS, 4 x v_mfma_f32_32x32x16_bf16
// (2 groups of 32 rays x 2 projections), 48 packed FMAs for the squares, clamp and bit accumulation of the 2 x 16
// results a lane holds, one v_permlane32_swap to bring the two halves of a ray's mask word together, one LDS store.
// Reports SIMD cycles per word at N waves per SIMD; the shipped VALU filter measures ~540 (profiles/r02_valu_findings.md).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_filter_probe.hip -o tools/bin/mfma_filter_probe && tools/bin/mfma_filter_probe 4 200
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float F2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// The clamp modifier on a packed FMA is only reachable through inline assembly, and the compiler's hazard recognizer does
// not look inside it: the FIRST statement that reads a fresh MFMA result carries the wait itself (volatile: the
// statements keep their order, so the later ones are covered).
template <bool FIRST>
__device__ __forceinline__ F2 pk_sub_sq_clamp(F2 p, F2 q) {
    F2 r;
    if (FIRST) asm volatile("s_nop 15\n\tv_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(r) : "v"(p), "v"(q));
    else asm volatile("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(r) : "v"(p), "v"(q));
    return r;
}

constexpr int kWords = 16;

__global__ __launch_bounds__(256) void k(const uint4* a_tiles, const float* rays, unsigned* out, unsigned long long* cycles, int passes) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint4* lds_a = reinterpret_cast<uint4*>(smem);                       // [kWords][64] 16-byte A fragments
    unsigned* masks = reinterpret_cast<unsigned*>(smem + kWords * 1024) + (threadIdx.x >> 6) * (kWords * 64);
    for (int i = threadIdx.x; i < kWords * 64; i += blockDim.x) lds_a[i] = a_tiles[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5;
    unsigned total = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int p = 0; p < passes; ++p) {
        // per-pass B fragments (the ray side); values do not matter for timing
        bf16x8 b1[2], b2[2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                b1[g][j] = static_cast<__bf16>(rays[(lane * 8 + j + p) & 1023] * (g + 1));
                b2[g][j] = static_cast<__bf16>(rays[(lane * 8 + j + p + 3) & 1023] * (g + 2));
            }
        const F2 pr2 = {rays[lane & 7] + 1.0e4f, rays[lane & 7] + 1.0e4f};
        unsigned nz = 0;
#pragma unroll 1
        for (int w = 0; w < kWords; ++w) {
            const uint4 av = lds_a[w * 64 + lane];
            bf16x8 a;
            __builtin_memcpy(&a, &av, 16);
            unsigned half[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1[g], z, 0, 0, 0);
                const f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2[g], z, 0, 0, 0);
                F2 acc = {0.0f, 0.0f};
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const F2 p1 = {d1[2 * m], d1[2 * m + 1]}, p2 = {d2[2 * m], d2[2 * m + 1]};
                    const F2 q = __builtin_elementwise_fma(-p1, p1, pr2);
                    acc = __builtin_elementwise_fma(acc, F2{2.0f, 2.0f}, m == 0 ? pk_sub_sq_clamp<true>(p2, q) : pk_sub_sq_clamp<false>(p2, q));
                }
                half[g] = ((static_cast<unsigned>(acc.x) << 8) | static_cast<unsigned>(acc.y)) << (h ? 0 : 16);
            }
            const u2 sw = __builtin_amdgcn_permlane32_swap(half[0], half[1], false, false);
            const unsigned m = sw[0] | sw[1];
            if (m != 0) {
                masks[w * 64 + lane] = m;
                nz |= 0x80000000u >> w;
            }
        }
        total += nz;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = total;
    if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? std::atoi(argv[1]) : 4, passes = argc > 2 ? std::atoi(argv[2]) : 200;
    if (waves < 1 || waves > 8 || passes < 1 || passes > 100000) { std::fprintf(stderr, "usage: mfma_filter_probe [waves per SIMD 1..8] [passes]\n"); return 2; }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, grid = n_cu * waves;
    const int need = kWords * 1024 + 4 * kWords * 64 * 4;                    // A tiles + the four waves' mask words
    int lds = (160 * 1024 / waves) & ~1023;
    if (lds < need) { std::fprintf(stderr, "%d waves per SIMD leave %d B of LDS per workgroup, the probe needs %d\n", waves, lds, need); return 2; }
    std::vector<unsigned short> a(kWords * 64 * 8);
    for (size_t i = 0; i < a.size(); ++i) a[i] = static_cast<unsigned short>(0x3F80 + (i * 7) % 64);      // bf16 values near 1
    std::vector<float> rays(1024);
    for (int i = 0; i < 1024; ++i) rays[i] = 0.5f + 0.001f * i;
    uint4* d_a; float* d_r; unsigned* d_o; unsigned long long* d_c;
    CHECK(hipMalloc(&d_a, a.size() * 2)); CHECK(hipMalloc(&d_r, 4096)); CHECK(hipMalloc(&d_o, static_cast<size_t>(grid) * 256 * 4)); CHECK(hipMalloc(&d_c, static_cast<size_t>(grid) * 4 * 8));
    CHECK(hipMemcpy(d_a, a.data(), a.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_r, rays.data(), 4096, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_a, d_r, d_o, d_c, 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, d_a, d_r, d_o, d_c, passes);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(static_cast<size_t>(grid) * 4);
    CHECK(hipMemcpy(c.data(), d_c, c.size() * 8, hipMemcpyDeviceToHost));
    double mx = 0;
    for (auto v : c) mx = v > mx ? static_cast<double>(v) : mx;
    std::printf("%d waves/SIMD: %.0f SIMD cycles per word of 32 spheres x 64 rays (slowest wave %.0f cycles for %d passes of %d words)\n", waves,
                mx / (static_cast<double>(passes) * kWords * waves), mx, passes, kWords);
    return 0;
}
