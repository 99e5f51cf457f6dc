// mfma_layout_check.hip — exact-integer check of the operand and result lane maps of v_mfma_f32_32x32x16_bf16 and of
// v_permlane32_swap_b32 as the filter relies on them (cdna_hip_programming.md "Fragment layout"):
//   A: lane l holds A[row l&31][k = 8 (l>>5) + j], j = 0..7;  B: lane l holds B[k = 8 (l>>5) + j][col l&31];
//   D: register reg of lane l is D[row (reg&3) + 8 (reg>>2) + 4 (l>>5)][col l&31];
//   permlane32_swap(x, y): x' = (x of lanes 0..31, y of lanes 0..31 moved to lanes 32..63), y' = (x of lanes 32..63 moved to 0..31, y of lanes 32..63).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_layout_check.hip -o tools/bin/mfma_layout_check && tools/bin/mfma_layout_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// A[row][k] = (k == k0) ? row + 1 : 0 ... plus a second non-zero column k1 with weight 64, B[k][col] = (k == k0) ? col + 1 : (k == k1) ? 1 : 0
// => D[row][col] = (row + 1)(col + 1) + 64 (row + 1) [k1 != k0]
__global__ void k(int k0, int k1, float* d_out, unsigned* swap_out) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int kk = 8 * h + j;
        a[j] = static_cast<__bf16>(kk == k0 ? static_cast<float>(r + 1) : kk == k1 ? 64.0f * (r + 1) : 0.0f);
        b[j] = static_cast<__bf16>(kk == k0 ? static_cast<float>(r + 1) : kk == k1 ? 1.0f : 0.0f);
    }
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) d_out[l * 16 + reg] = d[reg];
    const u2 s = __builtin_amdgcn_permlane32_swap(1000u + l, 2000u + l, false, false);
    swap_out[2 * l] = s[0];
    swap_out[2 * l + 1] = s[1];
}

int main() {
    float* d_out; unsigned* d_sw;
    CHECK(hipMalloc(&d_out, 64 * 16 * 4)); CHECK(hipMalloc(&d_sw, 128 * 4));
    float h[64 * 16]; unsigned sw[128];
    int bad = 0;
    for (int k0 = 0; k0 < 16; ++k0) {
        const int k1 = (k0 + 5) % 16;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, k0, k1, d_out, d_sw);
        CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
        for (int l = 0; l < 64; ++l)
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), col = l & 31;
                const float want = static_cast<float>((row + 1) * (col + 1) + 64 * (row + 1));
                if (h[l * 16 + reg] != want) { if (bad < 5) std::printf("k0 %d lane %d reg %d: %g, expected %g\n", k0, l, reg, h[l * 16 + reg], want); ++bad; }
            }
    }
    CHECK(hipMemcpy(sw, d_sw, sizeof(sw), hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) {
        const unsigned want0 = l < 32 ? 1000u + l : 2000u + (l - 32), want1 = l < 32 ? 1000u + (l + 32) : 2000u + l;
        if (sw[2 * l] != want0 || sw[2 * l + 1] != want1) { if (bad < 10) std::printf("swap lane %d: (%u, %u), expected (%u, %u)\n", l, sw[2 * l], sw[2 * l + 1], want0, want1); ++bad; }
    }
    std::printf(bad ? "MISMATCHES: %d\n" : "mfma_f32_32x32x16_bf16 A/B/D maps and permlane32_swap: as documented (%d mismatches)\n", bad);
    return bad ? 1 : 0;
}
