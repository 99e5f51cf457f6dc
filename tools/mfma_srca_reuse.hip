// mfma_srca_reuse.hip — third standalone probe for the hazard of profiles/r03_hazard_notes.md.
// What the render kernel established (hand-edited code objects, tools/hazard_hunt/): two v_mfma_f32_32x32x16_f16 of one
// wave that NAME THE SAME A REGISTERS, with a scalar-ALU instruction issued between them, deliver wrong candidates once
// three or more waves share a SIMD; with the second MFMA reading a COPY of A in other registers the same code is exact.
// The first two probes (only MFMA and plain vector work on the SIMD) did not reproduce it.  Here every wave alternates at
// random between AGGRESSOR passes (16 x { A from LDS; v_mfma; s_mov_b32; v_mfma, same A registers; all 32 results per lane
// checked }) and DISTURBER passes of one kind, chosen by the template parameter:
//   0 none (plain FMAs)   1 global loads   2 global atomics without return   3 LDS reads and writes
//   4 transcendentals (v_sqrt, v_rcp, v_sin, v_exp, v_log)   5 v_permlane32_swap   6 v_mad_u64_u32 + v_readlane/v_writelane
//   7 all of them   8 scalar memory loads (s_load through the constant address space)   9 s_memrealtime + s_sleep
//   10 global atomics WITH return (the packet draw)   11 LDS atomics with return   12 kinds 8-11 together
//   13 per-lane divergent loops of IEEE square root and division on LDS operands (the exact stage)   14 kinds 5, 13 and f16 conversions (the coefficient build)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_srca_reuse.hip -o tools/bin/mfma_srca_reuse
//   tools/bin/mfma_srca_reuse [workgroups per CU, default 4] [passes, default 4000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTiles = 16;
constexpr int kRecords = 64;

struct Record {
    unsigned wg, wave, pass, tile, which, reg, lane;
    float got, want, a, b0, b1;
};

__host__ __device__ inline int a_of(int wg, int tile) { return 1 + (wg * kTiles + tile) % 61; }
__host__ __device__ inline int b_of(int gwave, int which) { return (which ? 31 : 1) + gwave % 29; }

template <int KIND, bool BETWEEN>
__global__ __launch_bounds__(256, 4) void k(unsigned long long* counts, Record* rec, unsigned* n_rec, int passes, float* gbuf, unsigned long long* gacc) {
    extern __shared__ __align__(16) unsigned char smem[];
    f16x8* tiles = reinterpret_cast<f16x8*>(smem);          // [kTiles][64] A fragments
    float* lscratch = reinterpret_cast<float*>(smem + kTiles * 64 * 16);      // [4 waves][8][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.x;
    const int gwave = wg * 4 + wave;
    for (int i = threadIdx.x; i < kTiles * 64; i += blockDim.x) {
        f16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((i & 63) < 32) f[0] = static_cast<_Float16>(a_of(wg, i >> 6));
        tiles[i] = f;
    }
    for (int i = threadIdx.x; i < 4 * 8 * 64; i += blockDim.x) lscratch[i] = 1.0f;
    __syncthreads();
    f16x8 b0 = {0, 0, 0, 0, 0, 0, 0, 0}, b1 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) { b0[0] = static_cast<_Float16>(b_of(gwave, 0)); b1[0] = static_cast<_Float16>(b_of(gwave, 1)); }
    const float fb0 = static_cast<float>(b_of(gwave, 0)), fb1 = static_cast<float>(b_of(gwave, 1));
    unsigned long long bad = 0;
    unsigned lcg = static_cast<unsigned>(gwave) * 2654435761u + 12345u;
    float filler = 1.0f + 0.001f * lane;
    uint32_t ifill = lane * 2654435761u;
    float* mine = lscratch + wave * 8 * 64;
    for (int p = 0; p < passes; ++p) {
        lcg = lcg * 1664525u + 1013904223u;
        const bool aggressor = __builtin_amdgcn_readfirstlane((lcg >> 20) & 1u) != 0;
        if (aggressor) {
            for (int t = 0; t < kTiles; ++t) {
                const f16x8 a = tiles[t * 64 + lane];
                f32x16 d0, d1;
                if (BETWEEN)
#ifdef SAME_REGISTERS          // the register numbers of the failing render kernel (tools/hazard_hunt): D v[0:15], v[16:31]; A v[46:49]; B v[32:35], v[36:39]
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\ts_mov_b32 s60, s60\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, 0\n\ts_nop 15\n\ts_nop 3"
                                 : "={v[0:15]}"(d0), "={v[16:31]}"(d1) : "{v[46:49]}"(a), "{v[32:35]}"(b0), "{v[36:39]}"(b1) : "s60");
#else
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\ts_mov_b32 s60, s60\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, 0\n\ts_nop 15\n\ts_nop 3"
                                 : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b0), "v"(b1) : "s60");
#endif
                else
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %4, 0\n\ts_nop 15\n\ts_nop 3"
                                 : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b0), "v"(b1));
                const float fa = static_cast<float>(a_of(wg, t));
                const float w0 = fa * fb0, w1 = fa * fb1;
                unsigned wrong = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    wrong |= (d0[r] != w0 ? 1u : 0u) << r;
                    wrong |= (d1[r] != w1 ? 1u : 0u) << (16 + r);
                }
                if (wrong) {
                    bad += __popc(wrong);
                    const unsigned slot = atomicAdd(n_rec, 1u);
                    if (slot < kRecords) {
                        const int bit = __ffs(wrong) - 1;
                        float got = 0.0f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { if (bit == r) got = d0[r]; if (bit == 16 + r) got = d1[r]; }
                        rec[slot] = {static_cast<unsigned>(wg), static_cast<unsigned>(wave), static_cast<unsigned>(p), static_cast<unsigned>(t), bit >= 16 ? 1u : 0u,
                                     static_cast<unsigned>(bit & 15), static_cast<unsigned>(lane), got, bit >= 16 ? w1 : w0, fa, fb0, fb1};
                    }
                }
            }
        } else {
            const int reps = 8 + static_cast<int>((lcg >> 24) & 31u);
            for (int e = 0; e < reps; ++e) {
                if (KIND == 0) {
                    asm volatile("v_fma_f32 %0, %0, 0.5, 1.0" : "+v"(filler));
                }
                if (KIND == 1 || KIND == 7) {          // gathers of 16 bytes from a 1 MiB table (L2 resident), as the material fetches
                    const uint32_t at = ((ifill >> 8) & 0xFFFFu) * 4u;
                    const float4 v = *reinterpret_cast<const float4*>(gbuf + at);
                    filler += v.x + v.w;
                    ifill = ifill * 1664525u + 1013904223u;
                }
                if (KIND == 2 || KIND == 7) {          // u64 atomics without return, as the accumulator adds
                    atomicAdd(gacc + ((ifill >> 10) & 0xFFFFu), static_cast<unsigned long long>(ifill & 3u));
                    ifill = ifill * 1664525u + 1013904223u;
                }
                if (KIND == 3 || KIND == 7) {
                    mine[(e & 7) * 64 + lane] = filler;
                    filler += mine[((e + 3) & 7) * 64 + (lane ^ 1)] * 0.25f;
                }
                if (KIND == 4 || KIND == 7) {
                    filler = __builtin_sqrtf(fabsf(filler) + 1.0f) + __builtin_amdgcn_rcpf(filler + 2.0f) + __builtin_amdgcn_sinf(filler) * 0.125f +
                             __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(fabsf(filler) + 1.5f) * 0.33f);
                }
                if (KIND == 5 || KIND == 7) {
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(ifill, __float_as_uint(filler), false, false);
                    ifill ^= sw[1] >> 3;
                    filler = filler * 0.5f + static_cast<float>(sw[0] & 0xFFu);
                }
                if (KIND == 8 || KIND == 12) {          // scalar loads at a wave-uniform, changing address (K$ misses), as the descriptor and kernarg reads
                    typedef const float __attribute__((address_space(4))) * CF;
                    const uint32_t at = __builtin_amdgcn_readfirstlane((ifill >> 7) & 0x3FFF0u);
                    const float sv = ((CF)(gbuf))[at] + ((CF)(gbuf))[at + 5];
                    filler += sv;
                    ifill = ifill * 1664525u + 1013904223u;
                }
                if (KIND == 9 || KIND == 12) {
                    const unsigned long long tnow = __builtin_amdgcn_s_memrealtime();
                    __builtin_amdgcn_s_sleep(1);
                    ifill ^= static_cast<uint32_t>(tnow);
                }
                if (KIND == 10 || KIND == 12) {
                    if (lane == 0) ifill += static_cast<uint32_t>(atomicAdd(gacc + 7, 1ull));
                    ifill = __builtin_amdgcn_readfirstlane(ifill) + lane;
                }
                if (KIND == 11 || KIND == 12) {
                    unsigned* q = reinterpret_cast<unsigned*>(lscratch);
                    if (lane == (e & 63)) ifill += atomicAdd(q + (wave & 3), 1u);
                }
                if (KIND == 13 || KIND == 14) {          // lanes leave the loop one by one (partial EXEC), v_div_scale / v_div_fmas use VCC
                    const int iters = 1 + static_cast<int>((ifill >> 5) & 7u);
                    for (int q = 0; q < iters; ++q) {
                        const float x = mine[((q + e) & 7) * 64 + (lane ^ q)] + 1.5f;
                        const float disc = fabsf(filler) * x - 0.25f;
                        if (disc > 0.0f) {
                            const float sq = __builtin_sqrtf(disc);
                            const float root = (-0.5f - sq) / x;
                            if (root > -100.0f) filler = filler * 0.5f + root * 0.001f;
                        }
                    }
                    ifill = ifill * 1664525u + 1013904223u;
                }
                if (KIND == 14) {
                    const _Float16 h = static_cast<_Float16>(filler);
                    const float back = static_cast<float>(h);
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(filler - back), ifill, false, false);
                    ifill ^= __builtin_amdgcn_perm(sw[0], sw[1], 0x05040100u);
                }
                if (KIND == 6 || KIND == 7) {
                    const uint64_t pr = static_cast<uint64_t>(0xD2511F53u) * ifill;
                    ifill = static_cast<uint32_t>(pr >> 32) ^ static_cast<uint32_t>(pr);
                    const uint32_t s = __builtin_amdgcn_readlane(ifill, 5);
                    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, 9" : "+v"(ifill) : "s"(s));
                }
            }
        }
    }
    if (bad) atomicAdd(counts, bad);
    if (filler == 123.0f || ifill == 77u) gbuf[0] = filler;
}

static int g_ncu, g_per_cu = 4, g_passes = 4000;
static unsigned long long* d_counts;
static Record* d_rec;
static unsigned* d_nrec;
static float* d_gbuf;
static unsigned long long* d_gacc;

template <int KIND, bool BETWEEN>
static void run(const char* what) {
    const int lds = ((160 * 1024 / g_per_cu) & ~1023) - 1024, grid = g_ncu * g_per_cu;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, BETWEEN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMemset(d_counts, 0, 8));
    CHECK(hipMemset(d_nrec, 0, 4));
    hipLaunchKernelGGL((k<KIND, BETWEEN>), dim3(grid), dim3(256), lds, 0, d_counts, d_rec, d_nrec, g_passes, d_gbuf, d_gacc);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    unsigned n = 0;
    CHECK(hipMemcpy(&bad, d_counts, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&n, d_nrec, 4, hipMemcpyDeviceToHost));
    std::printf("%-22s %-34s %d workgroups/CU: %llu wrong results in %u lane-tiles\n", BETWEEN ? "MFMA; s_mov_b32; MFMA" : "MFMA; MFMA", what, g_per_cu, bad, n);
    if (n) {
        std::vector<Record> h(kRecords);
        CHECK(hipMemcpy(h.data(), d_rec, sizeof(Record) * kRecords, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < n && i < 10; ++i) {
            const Record& r = h[i];
            const float b = r.which ? r.b1 : r.b0;
            std::printf("    wg %u wave %u pass %u tile %u D%u[%u] lane %u: got %g, want %g (a %g, b0 %g, b1 %g; got / b = %g, got / a = %g)\n", r.wg, r.wave, r.pass, r.tile, r.which,
                        r.reg, r.lane, r.got, r.want, r.a, r.b0, r.b1, r.got / b, r.got / r.a);
        }
    }
}

int main(int argc, char** argv) {
    if (argc > 1) g_per_cu = std::atoi(argv[1]);
    if (argc > 2) g_passes = std::atoi(argv[2]);
    if (g_per_cu < 1 || g_per_cu > 8 || g_passes < 1) return 2;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    g_ncu = prop.multiProcessorCount;
    CHECK(hipMalloc(&d_counts, 8));
    CHECK(hipMalloc(&d_rec, sizeof(Record) * kRecords));
    CHECK(hipMalloc(&d_nrec, 4));
    CHECK(hipMalloc(&d_gbuf, (1 << 20) + 64));
    CHECK(hipMemset(d_gbuf, 0, (1 << 20) + 64));
    CHECK(hipMalloc(&d_gacc, 8 << 16));
    CHECK(hipMemset(d_gacc, 0, 8 << 16));
    run<0, true>("disturbers: plain FMAs");
    run<1, true>("disturbers: global loads");
    run<2, true>("disturbers: global atomics");
    run<3, true>("disturbers: LDS reads and writes");
    run<4, true>("disturbers: transcendentals");
    run<5, true>("disturbers: v_permlane32_swap");
    run<6, true>("disturbers: mad_u64, read/writelane");
    run<7, true>("disturbers: all kinds");
    run<7, false>("disturbers: all kinds");
    run<8, true>("disturbers: scalar memory loads");
    run<9, true>("disturbers: s_memrealtime, s_sleep");
    run<10, true>("disturbers: global atomic with return");
    run<11, true>("disturbers: LDS atomic with return");
    run<12, true>("disturbers: kinds 8-11");
    run<13, true>("disturbers: divergent IEEE sqrt / div loops");
    run<14, true>("disturbers: 13 + conversions, lane swaps");
    return 0;
}
