// mfma_salu_hazard.hip — standalone reproducer for the wrong closest hits of the class-word loop (DESIGN.md §4.1,
// profiles/r03_hazard_notes.md): does a scalar-ALU instruction issued BETWEEN two v_mfma_f32_32x32x16_f16 of one wave
// change what the matrix instructions deliver, once several waves share a SIMD?
//
// Every wave walks T tiles per pass, as the render kernel's filter does: A fragment of the tile from LDS (shared by the
// workgroup), two MFMAs against the wave's own two B fragments, then every one of the 32 results per lane is compared
// with the value it must have.  The operands are small integers (exact in f16 and fp32), different for every tile,
// workgroup and wave:  A[row][k] = a(workgroup, tile) for k == 0, else 0;  B[k][col] = b(wave) for k == 0, else 0;
// so every element of D is a * b, and a wrong element says whose operands it was made from (another tile's A = a stale
// or early read; another wave's B; a mix).  Mismatches are counted and the first few are recorded in full.
//
//   MODE 0  MFMA; MFMA                      (the shipped order)
//   MODE 1  MFMA; s_mov_b32 s_x, s_x; MFMA  (a scalar-ALU instruction between)
//   MODE 2  MFMA; s_nop 0; MFMA
//   MODE 3  MFMA; v_mov_b32 v_x, v_x; MFMA
//   MODE 4  MFMA; MFMA; s_mov_b32 s_x, s_x  (scalar-ALU instruction after the pair)
//   MODE 5  MFMA; s_cmp_eq_u32 s_x, s_y; MFMA
//   MODE 6  MFMA; s_add_i32; MFMA        MODE 7  MFMA; s_cmp_eq_u32; s_nop 7; MFMA
// (the scalar registers involved hold non-zero patterns, so that a scalar operand leaking into a matrix instruction shows)
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_salu_hazard.hip -o tools/bin/mfma_salu_hazard
//   tools/bin/mfma_salu_hazard [workgroups per CU = waves per SIMD, default 4] [passes, default 2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTiles = 16;       // tiles per pass (the stock scene has 15 class words)
constexpr int kRecords = 64;     // mismatch records kept

struct Record {
    unsigned wg, wave, pass, tile, which, reg, lane;
    float got, want;
    float a, b0, b1;
};

__host__ __device__ inline int a_of(int wg, int tile) { return 1 + (wg * kTiles + tile) % 61; }
__host__ __device__ inline int b_of(int gwave, int which) { return (which ? 31 : 1) + gwave % 29; }

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* counts, Record* rec, unsigned* n_rec, int passes, float* sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    f16x8* tiles = reinterpret_cast<f16x8*>(smem);          // [kTiles][64] A fragments
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wg = blockIdx.x;
    const int gwave = wg * 4 + wave;
    for (int i = threadIdx.x; i < kTiles * 64; i += blockDim.x) {
        f16x8 f = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((i & 63) < 32) f[0] = static_cast<_Float16>(a_of(wg, i >> 6));          // lanes 0..31 hold k = 0..7 of their row
        tiles[i] = f;
    }
    __syncthreads();
    f16x8 b0 = {0, 0, 0, 0, 0, 0, 0, 0}, b1 = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) { b0[0] = static_cast<_Float16>(b_of(gwave, 0)); b1[0] = static_cast<_Float16>(b_of(gwave, 1)); }
    const float fb0 = static_cast<float>(b_of(gwave, 0)), fb1 = static_cast<float>(b_of(gwave, 1));
    unsigned long long bad = 0;
    unsigned lcg = static_cast<unsigned>(gwave) * 2654435761u + 12345u;
    float filler = 1.0f;
    for (int p = 0; p < passes; ++p) {
        // a wave-dependent amount of vector work between passes, so that the waves of a SIMD drift against each other as
        // the render kernel's do
        lcg = lcg * 1664525u + 1013904223u;
        const int extra = static_cast<int>((lcg >> 24) & 63u);
        for (int e = 0; e < extra; ++e) asm volatile("v_fma_f32 %0, %0, 0.5, 1.0" : "+v"(filler));
        for (int t = 0; t < kTiles; ++t) {
            const f16x8 a = tiles[t * 64 + lane];
            f32x16 d0, d1;
            // the pair under test, wait states for the results included (8-pass XDL: 12 before a reader)
            // the pair under test, wait states for the results included (8-pass XDL: 12 before a reader).  s60/s61 hold
            // non-zero patterns (pi, e) so that a scalar operand leaking into a matrix instruction would show.
#define PAIR(BETWEEN, AFTER)                                                                                                            \
    asm volatile("s_mov_b32 s60, 0x40490fdb\n\ts_mov_b32 s61, 0x402df854\n\ts_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\t" BETWEEN          \
                 "v_mfma_f32_32x32x16_f16 %1, %2, %4, 0\n\t" AFTER "s_nop 15\n\ts_nop 3"                                               \
                 : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b0), "v"(b1), "v"(filler) : "s60", "s61", "scc")
            if (MODE == 0) PAIR("", "");
            else if (MODE == 1) PAIR("s_mov_b32 s61, s60\n\t", "");
            else if (MODE == 2) PAIR("s_nop 0\n\t", "");
            else if (MODE == 3) PAIR("v_mov_b32 %5, %5\n\t", "");
            else if (MODE == 4) PAIR("", "s_mov_b32 s61, s60\n\t");
            else if (MODE == 5) PAIR("s_cmp_eq_u32 s60, s61\n\t", "");
            else if (MODE == 6) PAIR("s_add_i32 s61, s60, s61\n\t", "");
            else PAIR("s_cmp_eq_u32 s60, s61\n\ts_nop 7\n\t", "");
#undef PAIR
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
    }
    if (bad) atomicAdd(counts, bad);
    if (filler == 123.0f) *sink = filler;
}

static int g_ncu, g_per_cu = 4, g_passes = 2000;
static unsigned long long* d_counts;
static Record* d_rec;
static unsigned* d_nrec;
static float* d_sink;

template <int MODE>
static void run(const char* what) {
    // LDS sized so that exactly g_per_cu workgroups fit on a CU (160 KiB); the tiles need 16 KiB
    const int lds = ((160 * 1024 / g_per_cu) & ~1023) - 1024, grid = g_ncu * g_per_cu;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipMemset(d_counts, 0, 8));
    CHECK(hipMemset(d_nrec, 0, 4));
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), lds, 0, d_counts, d_rec, d_nrec, g_passes, d_sink);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0;
    unsigned n = 0;
    CHECK(hipMemcpy(&bad, d_counts, 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&n, d_nrec, 4, hipMemcpyDeviceToHost));
    const double pairs = static_cast<double>(grid) * 4 * g_passes * kTiles;
    std::printf("MODE %d  %-44s %d workgroups/CU: %llu wrong results in %u lane-tiles of %.3g MFMA pairs\n", MODE, what, g_per_cu, bad, n, pairs);
    if (n) {
        std::vector<Record> h(kRecords);
        CHECK(hipMemcpy(h.data(), d_rec, sizeof(Record) * kRecords, hipMemcpyDeviceToHost));
        for (unsigned i = 0; i < n && i < 12; ++i) {
            const Record& r = h[i];
            // whose operands give the value that was found?
            char note[96] = "";
            const float b = r.which ? r.b1 : r.b0;
            if (r.got == 0.0f) std::snprintf(note, sizeof note, "zero");
            else if (r.got == r.a * (r.which ? r.b0 : r.b1)) std::snprintf(note, sizeof note, "= a x the OTHER B of this wave");
            else if (b != 0.0f && r.got / b == static_cast<float>(static_cast<int>(r.got / b))) std::snprintf(note, sizeof note, "= %g x this B (a of another tile?)", r.got / b);
            else if (r.got / r.a == static_cast<float>(static_cast<int>(r.got / r.a))) std::snprintf(note, sizeof note, "= this a x %g (another wave's B?)", r.got / r.a);
            std::printf("    wg %u wave %u pass %u tile %u D%u[%u] lane %u: got %g, want %g (a %g, b0 %g, b1 %g) %s\n", r.wg, r.wave, r.pass, r.tile, r.which, r.reg, r.lane,
                        r.got, r.want, r.a, r.b0, r.b1, note);
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
    CHECK(hipMalloc(&d_sink, 4));
    run<0>("MFMA; MFMA");
    run<1>("MFMA; s_mov_b32; MFMA");
    run<2>("MFMA; s_nop 0; MFMA");
    run<3>("MFMA; v_mov_b32; MFMA");
    run<4>("MFMA; MFMA; s_mov_b32");
    run<5>("MFMA; s_cmp_eq_u32; MFMA");
    run<6>("MFMA; s_add_i32; MFMA");
    run<7>("MFMA; s_cmp_eq_u32; s_nop 7; MFMA");
    return 0;
}
