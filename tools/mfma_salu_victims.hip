// mfma_salu_victims.hip — second standalone probe for the hazard of profiles/r03_hazard_notes.md.  The first probe
// (tools/mfma_salu_hazard.hip) showed that the MATRIX RESULTS of "MFMA; scalar-ALU instruction; MFMA" are right.  This one
// asks what else breaks: every wave alternates, pass by pass and at random, between
//   AGGRESSOR passes: 16 x { A fragment from LDS; v_mfma; <scalar-ALU instruction or not>; v_mfma; results checked }
//   VICTIM passes:    blocks of vector instructions that take a SCALAR register operand or produce one, each checked
//                     against the value it must have: fma with an SGPR constant and the clamp modifier (the filter's
//                     threshold instruction), v_mad_u64_u32 with an SGPR multiplier (Philox), v_readlane / v_writelane
//                     (SGPR spill code), v_cmp -> SGPR pair -> v_cndmask, v_readfirstlane,
// while 32 canary vector registers and 12 canary scalar registers per wave live through everything and are checked at
// the end.  4 workgroups per CU (one wave of each per SIMD), as the render kernel runs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_salu_victims.hip -o tools/bin/mfma_salu_victims
//   tools/bin/mfma_salu_victims [workgroups per CU, default 4] [passes, default 4000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTiles = 16;
constexpr int kKinds = 8;        // error counters: 0 matrix result, 1 fma+sgpr, 2 mad_u64+sgpr, 3 readlane, 4 writelane, 5 cmp/cndmask, 6 readfirstlane, 7 canaries

__host__ __device__ inline int a_of(int wg, int tile) { return 1 + (wg * kTiles + tile) % 61; }
__host__ __device__ inline int b_of(int gwave, int which) { return (which ? 31 : 1) + gwave % 29; }

template <int MODE>   // 0: MFMA; MFMA   1: MFMA; s_mov_b32; MFMA   2: MFMA; s_nop 0; MFMA
__global__ __launch_bounds__(256, 4) void k(unsigned long long* counts, int passes, float* sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    f16x8* tiles = reinterpret_cast<f16x8*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.x;
    const int gwave = wg * 4 + wave;
    for (int i = threadIdx.x; i < kTiles * 64; i += blockDim.x) {
        f16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((i & 63) < 32) f[0] = static_cast<_Float16>(a_of(wg, i >> 6));
        tiles[i] = f;
    }
    __syncthreads();
    f16x8 b0 = {0, 0, 0, 0, 0, 0, 0, 0}, b1 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) { b0[0] = static_cast<_Float16>(b_of(gwave, 0)); b1[0] = static_cast<_Float16>(b_of(gwave, 1)); }
    const float fb0 = static_cast<float>(b_of(gwave, 0)), fb1 = static_cast<float>(b_of(gwave, 1));
    unsigned bad[kKinds] = {0, 0, 0, 0, 0, 0, 0, 0};
    // canaries
    uint32_t cv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) cv[i] = 0xC0DE0000u + static_cast<uint32_t>(i) * 977u + static_cast<uint32_t>(lane) * 131u + static_cast<uint32_t>(gwave);
    uint32_t cs[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { cs[i] = 0x5CA10000u + static_cast<uint32_t>(i) * 7919u + static_cast<uint32_t>(gwave) * 3u; cs[i] = __builtin_amdgcn_readfirstlane(cs[i]); }
    unsigned lcg = static_cast<unsigned>(gwave) * 2654435761u + 12345u;
    // scalar constants of the victim blocks (wave-uniform, in SGPRs)
    uint32_t s_mul = __builtin_amdgcn_readfirstlane(0xD2511F53u), s_k = __builtin_amdgcn_readfirstlane(__float_as_uint(-0x1p48f));
    for (int p = 0; p < passes; ++p) {
        lcg = lcg * 1664525u + 1013904223u;
        const bool aggressor = __builtin_amdgcn_readfirstlane((lcg >> 20) & 1u) != 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(cv[i]));
#pragma unroll
        for (int i = 0; i < 12; ++i) asm volatile("" : "+s"(cs[i]));
        if (aggressor) {
            for (int t = 0; t < kTiles; ++t) {
                const f16x8 a = tiles[t * 64 + lane];
                f32x16 d0, d1;
#define PAIR(BETWEEN)                                                                                                                   \
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\t" BETWEEN "v_mfma_f32_32x32x16_f16 %1, %2, %4, 0\n\ts_nop 15\n\ts_nop 3"       \
                 : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b0), "v"(b1), "s"(cs[0]))
                if (MODE == 0) PAIR("");
                else if (MODE == 1) PAIR("s_mov_b32 %5, %5\n\t");
                else PAIR("s_nop 0\n\t");
#undef PAIR
                const float fa = static_cast<float>(a_of(wg, t));
                const float w0 = fa * fb0, w1 = fa * fb1;
#pragma unroll
                for (int r = 0; r < 16; ++r) bad[0] += (d0[r] != w0) + (d1[r] != w1);
            }
        } else {
            for (int rep = 0; rep < 24; ++rep) {
                const uint32_t x = lcg ^ (static_cast<uint32_t>(rep) * 0x9E3779B9u) ^ static_cast<uint32_t>(lane);
                // 1: the filter's threshold instruction: clamp(fma(d, -2^48 (SGPR), k)); d = +-2^-48 * small integer
                {
                    const float d = static_cast<float>(static_cast<int>(x & 7u) - 3) * 0x1p-48f, kk = (x & 8u) ? 1.0f : -1.0f;
                    float r;
                    asm volatile("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(d), "s"(s_k), "v"(kk));
                    const float want = fminf(fmaxf(-static_cast<float>(static_cast<int>(x & 7u) - 3) + kk, 0.0f), 1.0f);
                    bad[1] += r != want;
                }
                // 2: Philox's multiply: 32 x 32 -> 64 with the multiplier in an SGPR
                {
                    uint64_t r;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(r) : "v"(x), "s"(s_mul) : "vcc");
                    bad[2] += r != static_cast<uint64_t>(x) * 0xD2511F53ull;
                }
                // 3: v_readlane of a canary vector register (spill reload)
                {
                    uint32_t r;
                    asm volatile("v_readlane_b32 %0, %1, 5\n\ts_nop 1" : "=s"(r) : "v"(cv[rep & 31]));
                    bad[3] += r != 0xC0DE0000u + static_cast<uint32_t>(rep & 31) * 977u + 5u * 131u + static_cast<uint32_t>(gwave);
                }
                // 4: v_writelane of a scalar into a scratch vector register, read back by lane 7
                {
                    uint32_t v = x;
                    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, 7" : "+v"(v) : "s"(cs[rep % 12]));
                    const uint32_t want = lane == 7 ? 0x5CA10000u + static_cast<uint32_t>(rep % 12) * 7919u + static_cast<uint32_t>(gwave) * 3u : x;
                    bad[4] += v != want;
                }
                // 5: compare into an SGPR pair, select by it (VOP3 form)
                {
                    uint32_t r;
                    uint64_t m;
                    asm volatile("v_cmp_gt_u32_e64 %1, %2, %3\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, 17, 42, %1" : "=v"(r), "=&s"(m) : "v"(x & 0xFFFFu), "v"(0x8000u));
                    bad[5] += r != ((x & 0xFFFFu) > 0x8000u ? 42u : 17u);
                }
                // 6: v_readfirstlane
                {
                    const uint32_t r = __builtin_amdgcn_readfirstlane(cv[rep & 15]);
                    const int first = 0;
                    bad[6] += r != 0xC0DE0000u + static_cast<uint32_t>(rep & 15) * 977u + static_cast<uint32_t>(first) * 131u + static_cast<uint32_t>(gwave);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) bad[7] += cv[i] != 0xC0DE0000u + static_cast<uint32_t>(i) * 977u + static_cast<uint32_t>(lane) * 131u + static_cast<uint32_t>(gwave);
#pragma unroll
    for (int i = 0; i < 12; ++i) bad[7] += cs[i] != 0x5CA10000u + static_cast<uint32_t>(i) * 7919u + static_cast<uint32_t>(gwave) * 3u;
    for (int i = 0; i < kKinds; ++i)
        if (bad[i]) atomicAdd(counts + i, static_cast<unsigned long long>(bad[i]));
    if (lcg == 0x12345u) *sink = 1.0f;
}

static int g_ncu, g_per_cu = 4, g_passes = 4000;
static unsigned long long* d_counts;
static float* d_sink;

template <int MODE>
static void run(const char* what) {
    const int lds = ((160 * 1024 / g_per_cu) & ~1023) - 1024, grid = g_ncu * g_per_cu;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMemset(d_counts, 0, 8 * kKinds));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d_counts, g_passes, d_sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[kKinds];
    CHECK(hipMemcpy(h, d_counts, 8 * kKinds, hipMemcpyDeviceToHost));
    std::printf("%-26s %d workgroups/CU, %d passes/wave: wrong  matrix %llu  fma+sgpr %llu  mad_u64+sgpr %llu  readlane %llu  writelane %llu  cmp/cndmask %llu  readfirstlane %llu  canaries %llu\n",
                what, g_per_cu, g_passes, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
}

int main(int argc, char** argv) {
    if (argc > 1) g_per_cu = std::atoi(argv[1]);
    if (argc > 2) g_passes = std::atoi(argv[2]);
    if (g_per_cu < 1 || g_per_cu > 8 || g_passes < 1) return 2;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_ncu = prop.multiProcessorCount;
    CHECK(hipMalloc(&d_counts, 8 * kKinds));
    CHECK(hipMalloc(&d_sink, 4));
    run<0>("MFMA; MFMA");
    run<1>("MFMA; s_mov_b32; MFMA");
    run<2>("MFMA; s_nop 0; MFMA");
    return 0;
}
