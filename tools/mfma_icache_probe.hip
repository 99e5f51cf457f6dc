// mfma_icache_probe.hip — fourth standalone probe for the hazard of profiles/r03_hazard_notes.md.
// The three earlier probes are tight loops that live in the wave's instruction buffer; the render kernel is ~60 KB of code
// walked by 16 waves per CU at different places, so its waves stall on instruction fetch all the time, and the same pattern
// edited into the small intersect kernel does not fail.  Hypothesis tested here: the pair needs a FETCH STALL between its
// two MFMAs (with the scalar instruction there) to go wrong.  The kernel is 128 sections of ~1.5 KB each (> the 64 KB
// instruction cache); every pass a wave jumps to a random section, whose pattern sits at one of 16 alignments against the
// 64-byte fetch line:   s_nop x (section % 16);  v_mfma D0, A, B0;  [s_mov_b32];  v_mfma D1, A, B1;  wait;  180 FMAs.
// All 32 results per lane are checked against the values they must have (A differs by workgroup and tile, B by wave).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_icache_probe.hip -o tools/bin/mfma_icache_probe
//   tools/bin/mfma_icache_probe [workgroups per CU, default 4] [passes, default 20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTiles = 16;
constexpr int kRecords = 64;
constexpr int kSections = 128;

struct Record {
    unsigned wg, wave, pass, tile, which, reg, lane, section;
    float got, want, a, b0, b1;
};

__host__ __device__ inline int a_of(int wg, int tile) { return 1 + (wg * kTiles + tile) % 61; }
__host__ __device__ inline int b_of(int gwave, int which) { return (which ? 31 : 1) + gwave % 29; }

// the repeated lines are written out by the preprocessor (not .rept) so that the compiler's estimate of the statement's size,
// which it needs to choose between short and long branches, is not too small
#define PAD1(i) ".if %c6 > " #i "\n\ts_nop 0\n\t.endif\n\t"
#define PADS PAD1(0) PAD1(1) PAD1(2) PAD1(3) PAD1(4) PAD1(5) PAD1(6) PAD1(7) PAD1(8) PAD1(9) PAD1(10) PAD1(11) PAD1(12) PAD1(13) PAD1(14)
#define F1 "v_fma_f32 %2, %2, 0.5, 1.0\n\t"
#define F4 F1 F1 F1 F1
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16
#define FILLER F64 F64 F16 F16 F16 F4          // 180

template <int SEC, bool BETWEEN>
__device__ __forceinline__ void section(f32x16& d0, f32x16& d1, float& filler, f16x8 a, f16x8 b0, f16x8 b1) {
    if (BETWEEN)
        asm volatile(PADS "v_mfma_f32_32x32x16_f16 %0, %3, %4, 0\n\ts_mov_b32 s60, s60\n\tv_mfma_f32_32x32x16_f16 %1, %3, %5, 0\n\ts_nop 15\n\ts_nop 3\n\t" FILLER "s_nop 0"
                     : "=&v"(d0), "=&v"(d1), "+v"(filler) : "v"(a), "v"(b0), "v"(b1), "n"(SEC % 16) : "s60");
    else
        asm volatile(PADS "v_mfma_f32_32x32x16_f16 %0, %3, %4, 0\n\tv_mfma_f32_32x32x16_f16 %1, %3, %5, 0\n\ts_nop 15\n\ts_nop 3\n\t" FILLER "s_nop 0"
                     : "=&v"(d0), "=&v"(d1), "+v"(filler) : "v"(a), "v"(b0), "v"(b1), "n"(SEC % 16));
}

#define S1(n) case n: section<n, BETWEEN>(d0, d1, filler, a, b0, b1); break;
#define S4(n) S1(n) S1(n + 1) S1(n + 2) S1(n + 3)
#define S16(n) S4(n) S4(n + 4) S4(n + 8) S4(n + 12)
#define S64(n) S16(n) S16(n + 16) S16(n + 32) S16(n + 48)

template <bool BETWEEN>
__global__ __launch_bounds__(256, 4) void k(unsigned long long* counts, Record* rec, unsigned* n_rec, int passes, float* sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    f16x8* tiles = reinterpret_cast<f16x8*>(smem);          // [kTiles][64] A fragments
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
    unsigned long long bad = 0;
    unsigned lcg = static_cast<unsigned>(gwave) * 2654435761u + 12345u;
    float filler = 1.0f + 0.001f * lane;
    for (int p = 0; p < passes; ++p) {
        lcg = lcg * 1664525u + 1013904223u;
        const int sec = __builtin_amdgcn_readfirstlane((lcg >> 16) & (kSections - 1));
        const int t = __builtin_amdgcn_readfirstlane((lcg >> 8) & (kTiles - 1));
        const f16x8 a = tiles[t * 64 + lane];
        f32x16 d0, d1;
        switch (sec) {
            S64(0) S64(64)
            default: __builtin_unreachable();
        }
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
                             static_cast<unsigned>(bit & 15), static_cast<unsigned>(lane), static_cast<unsigned>(sec), got, bit >= 16 ? w1 : w0, fa, fb0, fb1};
            }
        }
    }
    if (bad) atomicAdd(counts, bad);
    if (filler == 123.0f) sink[0] = filler;
}

static int g_ncu, g_per_cu = 4, g_passes = 20000;
static unsigned long long* d_counts;
static Record* d_rec;
static unsigned* d_nrec;
static float* d_sink;

template <bool BETWEEN>
static void run() {
    const int lds = ((160 * 1024 / g_per_cu) & ~1023) - 1024, grid = g_ncu * g_per_cu;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<BETWEEN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMemset(d_counts, 0, 8));
    CHECK(hipMemset(d_nrec, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<BETWEEN>), dim3(grid), dim3(256), lds, 0, d_counts, d_rec, d_nrec, g_passes, d_sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long bad = 0;
    unsigned n = 0;
    CHECK(hipMemcpy(&bad, d_counts, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&n, d_nrec, 4, hipMemcpyDeviceToHost));
    // 181 FMAs + ~60 other instructions per pass: ~520 cycles of issue per wave when nothing stalls
    std::printf("%-22s %d workgroups/CU, %d passes per wave, %.1f ms (%.0f cycles per pass and wave at 2.4 GHz): %llu wrong results in %u lane-passes\n",
                BETWEEN ? "MFMA; s_mov_b32; MFMA" : "MFMA; MFMA", g_per_cu, g_passes, ms, ms * 1e-3 * 2.4e9 / g_passes, bad, n);
    if (n) {
        std::vector<Record> h(kRecords);
        CHECK(hipMemcpy(h.data(), d_rec, sizeof(Record) * kRecords, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < n && i < 12; ++i) {
            const Record& r = h[i];
            const float b = r.which ? r.b1 : r.b0;
            std::printf("    wg %u wave %u pass %u section %u (pad %u) tile %u D%u[%u] lane %u: got %g, want %g (a %g, b0 %g, b1 %g; got / b = %g, got / a = %g)\n", r.wg, r.wave,
                        r.pass, r.section, r.section % 16, r.tile, r.which, r.reg, r.lane, r.got, r.want, r.a, r.b0, r.b1, r.got / b, r.got / r.a);
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
    CHECK(hipMalloc(&d_sink, 64));
    run<true>();
    run<false>();
    return 0;
}
