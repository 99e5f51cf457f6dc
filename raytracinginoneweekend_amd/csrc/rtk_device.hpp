// rtk_device.hpp — device-side building blocks of the MI355X path tracer (gfx950, wave64).
//
// This translation unit is compiled with -ffp-contract=off: every `a*b + c` below is two rounded
// fp32 operations exactly as in the reference (which is built without FMA contraction, SURVEY.md §7
// hard part 1); fused multiply-adds appear only where `fmaf` is written explicitly (the filter).
// IEEE-correct sqrt and divide are hipcc's default (-fhip-fp32-correctly-rounded-divide-sqrt) and
// f32 denormals are preserved (hipcc default), matching x86-64 SSE arithmetic.
//
// Reference citations are relative to /root/reference/RayTracingInOneWeekend/.
#pragma once

#ifdef RTK_HOST_EMULATION
// tests/cpu_emul/rtk_host_shim.hpp: a one-lane-per-wave CPU build of this header for the CPU test-suite
// (test infrastructure; the shipped library never defines RTK_HOST_EMULATION).
#include "rtk_host_shim.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#ifndef RTK_HOST_EMULATION
#define RTK_STAT(lane, name) ((void)0)
#define RTK_STAT_ANY(lane, name, pred) ((void)0)
#define RTK_STAT_EACH(name) ((void)0)
#endif

#ifndef RTK_LANES
#define RTK_LANES 64
#endif

namespace rtk {

constexpr int kWave = RTK_LANES;     // CDNA wavefront (64); 1 in the CPU emulation build
constexpr int kTilePixels = 64;      // a tile / packet always holds 64 pixels
constexpr int kBlockThreads = 4 * kWave;   // 4 waves per workgroup, one per SIMD
constexpr int kWavesPerBlock = 4;
constexpr int kListCap = 16;         // per-lane candidate list entries between flushes
constexpr int kEntryCap = 16;        // streamed local walk: per-lane (mask word, first list index) entries between exact passes
constexpr int kMaxLdsSpheres = 1024; // scenes up to this size are LDS-resident (32 B per sphere)

// Relative widening of the filter's radius term: r^2 (1+2^-15)^2 (host, rtk_scene_prep.hpp); it absorbs the rounding of
// the squares and sums of the projected distances.
constexpr float kFilterScaleSq = (1.0f + 0x1p-15f) * (1.0f + 0x1p-15f);

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(float t, V3 v) { return {t * v.x, t * v.y, t * v.z}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// Vector3.cpp:39-41 — (x x' + y y') + z z', each op rounded (TU is contract-off).
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// Vector3.cpp:35-37 and 61-63 — reciprocal then three multiplies.
__device__ __forceinline__ V3 div_by(V3 v, float t) { return (1.0f / t) * v; }
__device__ __forceinline__ V3 unit_vector(V3 v) { return div_by(v, __builtin_sqrtf(length_squared(v))); }
// Vector3.cpp:65-67.
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0f * dot(v, n)) * n; }
// Vector3.cpp:69-74.
__device__ __forceinline__ V3 refract(V3 uv, V3 n, float eta) {
    const float cos_theta = fminf(dot(-uv, n), 1.0f);
    const V3 perp = eta * (uv + cos_theta * n);
    const V3 par = (-__builtin_sqrtf(fabsf(1.0f - length_squared(perp)))) * n;
    return perp + par;
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11), replacing the two sequential mt19937
// streams of MathUtils.hpp:15-25.  key = seed, counter = (pixel, sample, bounce, stream).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 product each (v_mad_u64_u32) instead of separate low and high multiplies
        const uint64_t p0 = static_cast<uint64_t>(M0) * c.x, p1 = static_cast<uint64_t>(M1) * c.z;
        const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
        const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0;
        k.y += W1;
    }
    return c;
}
// 24 random bits -> [0, 1): exactly representable, never 1.
__device__ __forceinline__ float u01(uint32_t x) { return static_cast<float>(x >> 8) * 0x1p-24f; }

// Direct maps with the same distributions as the reference's rejection samplers.
// Uniform on the unit disk (Vector3.cpp:53-59).  v_sin/v_cos take revolutions.
__device__ __forceinline__ void sample_disk(float u1, float u2, float& x, float& y) {
    const float r = __builtin_sqrtf(u1);
    x = r * __builtin_amdgcn_cosf(u2);
    y = r * __builtin_amdgcn_sinf(u2);
}
// Uniform on the unit sphere: what unit_vector(random_in_unit_sphere()) produces (Vector3.cpp:49-51).
__device__ __forceinline__ V3 sample_unit_vector(float u1, float u2) {
    const float z = 1.0f - 2.0f * u1;
    const float rho = __builtin_sqrtf(fmaxf(0.0f, 1.0f - z * z));
    return {rho * __builtin_amdgcn_cosf(u2), rho * __builtin_amdgcn_sinf(u2), z};
}
// Uniform in the unit ball (Vector3.cpp:3-9): direction x cbrt(u).
__device__ __forceinline__ V3 sample_in_unit_sphere(float u1, float u2, float u3) {
    const float rad = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(u3) * (1.0f / 3.0f));
    return rad * sample_unit_vector(u1, u2);
}

// ---------------------------------------------------------------------------------------------
// Scene as the kernels see it.
// ---------------------------------------------------------------------------------------------
// Scene-wide bounds the filter needs (over the FILTERABLE spheres only; the others are candidates for every ray).
struct FilterBounds {
    float c_max;     // max |C| + |r|
    float r_max2;    // max r^2
    float r_min;     // min |r|
    // matrix-pipe layout (LDS-resident scenes):
    float sigma;     // power of two with sigma * c_max <= 1: the tiles hold monomials of the scaled centres C' = sigma C
    float cls_cy;    // scaled cy' shared by the spheres of every class word
    float cls_r2;    // their scaled, widened r'^2
    float acc_eps;   // relative bound of the matrix pipe's evaluation error for this scene's word kinds (kQuadEpsClass / General)
    int32_t kinds;   // bit 0: the scene has class words, bit 1: general words
    // products of the above that the margin needs per ray, made on the host: as kernel arguments they are scalar operands,
    // computed in the kernel they were loop invariants held in vector registers for the life of the kernel
    float cp;        // c' = sigma c_max (<= 1)
    float cp_twice;  // 2 c'
    float rp2;       // r'max^2 = sigma^2 r_max2
    float t_quad;    // sqrt(2) c'^2 + r'max^2, rounded up
    float cls_cy2;   // cls_cy^2
    // word-local frames (streamed scenes):
    float lsig_max;  // the largest sigma_w of the scene's class words (0: none)
};

struct SceneDev {
    // Filter operands, one block per WORD of 32 consecutive list entries (list order is kept: the exact stage's tie
    // rule depends on it).
    // Scenes of up to kMaxLdsSpheres spheres (wdesc != nullptr) — the whole filter test is ONE dot product on the
    // matrix pipe (quad_term below):
    //   wdesc[w] = (float4 index of the word's tile or records in filt, kind: 0 general, 1 class, 2 short tail,
    //   always-candidate bits, valid bits).
    //   A tile is 64 lane fragments per 16 terms: filt[off + 64 h + l] = lane l's 16-byte A fragment of
    //   v_mfma_f32_32x32x16_f16 — row l & 31, terms 16 h + 8 (l >> 5) ... + 7, eight f16.  Row rho holds sphere
    //   tile_row_sphere(rho), chosen so that the results a lane receives are 16 consecutive spheres.
    //   CLASS word (kind 1; one tile fragment set, K = 16): its filterable spheres share cy and radius with the scene's
    //   class (FilterBounds::cls_cy, cls_r2) — what every small sphere of the reference's random_scene() is
    //   (main.cpp:124).  GENERAL word (kind 0; two fragment sets, K = 32).  Pads and always-candidates are zero rows;
    //   the descriptor's valid bits mask what a zero row yields, its always bits are OR-ed in.
    //   A last word of at most kShortTail spheres gets no tile: its spheres follow as plain records
    //   (cx, cy, cz, r^2 (1+k)) and are tested on the vector ALUs (a tile would cost a full word for a few spheres).
    // Larger scenes (wdesc == nullptr), streamed through LDS chunks and filtered on the vector ALUs:
    //   filt[i] = (cx, cy, cz, r^2 (1+k)), i < npad; pads -inf, always-candidates +inf.
    const float4* filt;   // [filt_len]
    const float4* exact;  // [npad] cx, cy, cz, r*r  (fl(r*r), Sphere3.hpp:35)
    const float4* mat0;   // [n] colour r, g, b, roughness
    const float4* mat1;   // [n] metallic, ior, type (int bits), 1/radius (fl(1/r), Vector3.cpp:36)
    int32_t n;
    int32_t npad;         // n rounded up to a multiple of 32
    FilterBounds fb;
    const float4* wdesc = nullptr;   // [npad / 32]
    int32_t filt_len = 0;            // float4 entries of filt
    int32_t filt_lds = 0;            // matrix-pipe layout: entries of filt that the kernels keep in LDS (all of it)
    // Streamed scenes whose words are (nearly) all class words in their OWN frame (local_term below): per word a 64-fragment
    // tile and three float4 of descriptor; null when the scene streams through the vector-ALU filter alone.
    // The list is cut into `lwords` words of at most 32 consecutive entries (rtk_scene_prep.hpp).
    const float4* ltiles = nullptr;  // [lwords][64]
    const float4* lwdesc = nullptr;  // [lwords][3]: (frame centre x, y, z, 2 / sigma_w) (k0, 2 * first entry + kind, always bits, valid bits: one per entry, from the top) (s1, -2 sigma_w 2^13, sigma_w^2, -): rtk_scene_prep.hpp
    int32_t lwords = 0;
};
constexpr int kShortTail = 8;

// ---- the filter test as one dot product (matrix-pipe layout) --------------------------------------------------------
// With d the unit direction, p = o' - (o'.d) d the part of the scaled origin perpendicular to the ray and
// q = I - d d^T, the squared distance of a scaled centre C' from the ray's line is
//     P = sum_ij q_ij C'_i C'_j - 2 p.C' + |p|^2,
// bilinear in MONOMIALS of the centre (C'x^2, C'y^2, C'z^2, C'x C'y, C'x C'z, C'y C'z, C'x, C'y, C'z, and r'^2 for the
// threshold) and COEFFICIENTS of the ray.  Each monomial m (|m| <= 1, times 2^14) and each coefficient c (times t 2^13,
// t a power of two per ray that keeps |c t| < 4) is split into two f16 parts, m = m1 + m2, c = c1 + c2, both to 2^-24
// relative; the product is summed from the three terms (m1 c1, m1 c2, m2 c1), what is left out is below 2^-23 |m c|.
// A class word needs the five monomials without y (cy' and r'^2 are folded into the ray's coefficients and constant):
// 15 terms, K = 16.  A general word: nine monomials x 3 terms, then r'^2 as two parts against the one-part coefficient
// -t: 29 terms, K = 32.  The ray's constant stays on the vector ALUs (Intersector::closest_hit).
constexpr int kQuadClassMonomials = 5;     // xx zz xz x z
constexpr int kQuadGeneralMonomials = 9;   // xx yy zz xy xz yz x y z
constexpr float kQuadSphereScale = 0x1p14f, kQuadRayScale = 0x1p13f;
// Bound, relative to the sum T of the absolute values of all terms and the constant, of everything the evaluation does to
// the true value: 16 (32) fp32 accumulation steps of unspecified order and rounding, 2^-23 T each;
// left-out part products and part roundings, 2^-22.4 T; the ray's coefficients and constant in fp32, 2^-19 T.
constexpr float kQuadEpsClass = 0x1p-19f + 0x1p-19f + 0x1p-22f, kQuadEpsGeneral = 0x1p-18f + 0x1p-19f + 0x1p-22f;
// term k of a tile row: which monomial, which part (0, 1) of the sphere's monomial and of the ray's coefficient;
// monomial -1: unused term (zero), monomial `n_mono`: r'^2 (general words only)
__host__ __device__ inline void quad_term(int k, int n_mono, int& mono, int& sphere_part, int& ray_part) {
    if (k < 3 * n_mono) { mono = k / 3; sphere_part = (k % 3) == 2; ray_part = (k % 3) == 1; }
    else if (n_mono == kQuadGeneralMonomials && k < 3 * n_mono + 2) { mono = n_mono; sphere_part = k - 3 * n_mono; ray_part = 0; }
    else { mono = -1; sphere_part = 0; ray_part = 0; }
}
// Result register reg (0..15) of a lane in half hf (lane >> 5) is row (reg & 3) + 8 (reg >> 2) + 4 hf of the tile.  The
// filter accumulates even registers into one bit string and odd ones into another, so row rho holds this sphere of the
// word: lanes of half hf receive spheres 16 hf .. 16 hf + 15, register 2 m + c sphere 16 hf + 8 c + m.
__host__ __device__ inline int tile_row_sphere(int rho) {
    const int hf = (rho >> 2) & 1, r = rho - 4 * hf, c = r & 1, m = 2 * (r >> 3) + ((r >> 1) & 1);
    return 16 * hf + 8 * c + m;
}
// ---- word-local frames: the matrix-pipe filter for scenes beyond LDS --------------------------------
// In scene-scaled units a 0.2-unit sphere of a 10 000-sphere scene has r'^2 ~ 8e-6, below what 16 fp32 accumulation steps
// resolve (2^-19 T, T ~ 2).  So every word of 32 consecutive spheres that share cy and radius gets its OWN frame: centre c_w
// (cy exactly, the middle of the word's x and z ranges) and a power of two sigma_w with sigma_w |C - c_w|_inf <= kLocalB and
// sigma_w r~ <= kLocalB.  With delta = sigma_w (C - c_w) (delta_y = 0), v = sigma_w (o - c_w), p = v - (v.d) d:
//     P = qxx dx^2 + qzz dz^2 + qxz' dx dz - 2 px dx - 2 pz dz + |p|^2        (the class word's five monomials, cy' = 0)
// The three quadratic coefficients are the ray's alone (built once per pass); the two linear ones and the constant depend on
// the word: ~45 vector instructions per word rebuild their f16 parts and the constant, against the 192 that the
// vector-ALU form spends on the word's 32 tests beyond the 64 both need.  Term order of a local tile (local_term): the six
// terms of the linear coefficients sit in B dwords lo[2], lo[3], hi[3], so that two lane swaps per word rebuild the B
// operands.  Margin (all in the word's scaled units, u = 2^-24):
//   evaluation: kQuadEpsClass T,  T <= 3 B^2 + sqrt(2) B + r~'^2 + (1 + sqrt(2) B) |p|^2     (2 B (|px| + |pz|) <= sqrt(2) B (1 + |p|^2))
//   the reference's own rounding: 2^-19 T1,  T1 = |v - delta|^2 + r'^2 <= 1.25 |v|^2 + 10 B^2 + r'^2
//   v, v.d, p, |p|^2 in fp32: u |v| (28 |p| + 25) <= 2^-19 (0.5 |v|^2 + 0.81 |p|^2 + 0.64)
//   => marg = 1.25 (sum) + 2^-21 = A0_w + kLocalA1 |p|^2 + kLocalA2 |v|^2,   K = |p|^2 - r~'^2 - marg.
// p is clamped to 2 per component for the coefficients (|2 p| 2^13 must stay an f16 number).  A clamped ray passes at least
// 2 - sqrt(2) B > 1 from every centre of the word, and its (wrong) sum P_c = |delta_perp - p|^2 + 2 (p - p_c).delta is at
// least a^2 + dx^2 - 4 dx >= 1.89 (a = |p_x| >= 2, |dx| <= B), while r~'^2 + marg < B^2 + 0.3 for |v|^2 < kLocalSafeVV: still
// no candidate; beyond kLocalSafeVV, or NaN, the word is judged as a whole (Intersector::closest_hit).
constexpr float kLocalB = 0.625f;
constexpr float kLocalSafeVV = 0x1p14f;
constexpr int kLocalShortRun = 8;           // class runs of at most this many entries go into kind-0 words (vector ALUs, one batch of scalar loads)
constexpr double kLocalReach = 100.0;      // a class word's centres stay within this many radii of its frame centre (host cut)
constexpr float kLocalA1 = 1.25f * ((1.0f + 1.41422f * kLocalB) * kQuadEpsClass + 0.81f * 0x1p-19f);
constexpr float kLocalA2 = 1.25f * 1.75f * 0x1p-19f;
constexpr float kLocalRatio = -kLocalA2 / (1.0f - kLocalA1);      // K = (1 - A1) (|p|^2 + ratio |v|^2) - r~'^2 - A0_w
// the word's part of the margin, given its scaled widened radius^2 (host, rounded up by the caller)
inline double local_margin_const(double r2s) {
    const double B = kLocalB;
    return 1.25 * (static_cast<double>(kQuadEpsClass) * (3.0 * B * B + 1.41422 * B + r2s) + 0x1p-19 * (10.0 * B * B + 0.64 + r2s)) + 0x1p-21;
}
// term k of a local tile row: monomial (0 xx, 1 zz, 2 xz, 3 x, 4 z; -1 unused), part of the sphere's monomial, part of the ray's coefficient
__host__ __device__ inline void local_term(int k, int& mono, int& sphere_part, int& ray_part) {
    const int idx = k < 4 ? k : (k >= 8 && k <= 12 ? k - 4 : -1);          // the nine quadratic terms: k = 0..3, 8..12
    sphere_part = 0; ray_part = 0;
    if (idx >= 0) { mono = idx / 3; sphere_part = (idx % 3) == 2; ray_part = (idx % 3) == 1; return; }
    switch (k) {
        case 4: mono = 3; break;
        case 5: mono = 3; ray_part = 1; break;
        case 6: mono = 3; sphere_part = 1; break;
        case 7: mono = 4; sphere_part = 1; break;
        case 14: mono = 4; break;
        case 15: mono = 4; ray_part = 1; break;
        default: mono = -1;
    }
}
// float -> IEEE binary16 bits, round to nearest even (overflow -> inf; subnormals kept), and back
__host__ __device__ inline uint16_t f16_rne(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint16_t, static_cast<_Float16>(v));
#else
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, a = u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return static_cast<uint16_t>(sign | (a > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (a >= 0x477FF000u) return static_cast<uint16_t>(sign | 0x7C00u);          // rounds to 2^16 or more
    if (a < 0x33000001u) return static_cast<uint16_t>(sign);                      // at most 2^-25: rounds to zero
    const int e = static_cast<int>(a >> 23) - 127;
    uint32_t m = (a & 0x7FFFFFu) | 0x800000u;                                     // 24-bit significand
    const int shift = e >= -14 ? 13 : 13 + (-14 - e);                             // bits dropped
    const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
    m >>= shift;
    if (rest > half || (rest == half && (m & 1u))) ++m;
    // normal: m in [0x400, 0x800] with the implicit bit; a carry to 0x800 moves into the exponent by itself
    const uint32_t bits = e >= -14 ? (static_cast<uint32_t>(e + 15 - 1) << 10) + m : m;
    return static_cast<uint16_t>(sign | bits);
#endif
}
__host__ __device__ inline float f16_value(uint16_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<float>(__builtin_bit_cast(_Float16, b));
#else
    const uint32_t sign = static_cast<uint32_t>(b & 0x8000u) << 16, e = (b >> 10) & 0x1Fu, m = b & 0x3FFu;
    float v;
    if (e == 0) {
        v = static_cast<float>(m) * 0x1p-24f;
    } else if (e == 31) {
        const uint32_t u = 0x7F800000u | (m << 13);
        __builtin_memcpy(&v, &u, 4);
    } else {
        const uint32_t u = ((e + 112u) << 23) | (m << 13);
        __builtin_memcpy(&v, &u, 4);
    }
    uint32_t u;
    __builtin_memcpy(&u, &v, 4);
    u |= sign;
    __builtin_memcpy(&v, &u, 4);
    return v;
#endif
}
// v ~ p[0] + p[1] in f16 parts (|v - p[0] - p[1]| <= 2^-24 |v| while the second part is a normal number)
__host__ __device__ inline void f16_split2(float v, uint16_t p[2]) {
    p[0] = f16_rne(v);
    p[1] = f16_rne(v - f16_value(p[0]));
}

struct CameraDev {
    V3 origin, llc, horizontal, vertical, u, v, w;
    float lens_radius;
};

struct Ray {
    V3 o, d;
};

// Constant-rate device clock (100 MHz), for the kernel's drain statistics only.
#ifndef RTK_HOST_EMULATION
__device__ __forceinline__ unsigned long long rtk_wall_clock() { return wall_clock64(); }
#endif

// Wave-level helpers (wave64).
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ bool wave_any(bool p) { return __ballot(p) != 0ull; }
// sum of a small per-lane count over the wave, by votes on its bits (every lane gets the total)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    uint32_t s = 0;
    for (int b = 0; b < 32 && wave_any((v >> b) != 0u); ++b) s += static_cast<uint32_t>(__popcll(__ballot(((v >> b) & 1u) != 0u))) << b;
    return s;
}
__device__ __forceinline__ int lane_prefix(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0));
}

// One exact ray-sphere test: Sphere3.hpp:31-49 operation for operation (TU is contract-off; `a`
// is hoisted per ray and r*r precomputed per sphere, both bit-identical to the reference's
// per-call values).  Comparison polarity is the reference's: every test is false for NaN.
__device__ __forceinline__ bool sphere_root_exact(const Ray& r, float a, float4 s /*cx,cy,cz,rr*/, float t_min, float t_max,
                                                  float& root_out) {
    const V3 oc = r.o - V3{s.x, s.y, s.z};
    const float half_b = dot(oc, r.d);
    const float c = length_squared(oc) - s.w;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    const float sqrtd = __builtin_sqrtf(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    root_out = root;
    return true;
}


// hit_record construction (Sphere3.hpp:51-56, Hittable.hpp:15-18) + Material::scatter
// (Material.cpp:6-54).  u = four uniforms in [0,1).  Returns scatter's bool; `out` is the scattered ray.
// `ud` = unit_vector(in.d), evaluated once by the caller.
__device__ __forceinline__ bool hit_and_scatter(const Ray& in, const V3 ud, float t, float4 ex /*cx,cy,cz,rr*/, float4 m0, float4 m1,
                                                const float u[4], Ray& out) {
    const V3 center = {ex.x, ex.y, ex.z};
    const V3 p = in.o + t * in.d;                       // Ray3.hpp:19-21
    const V3 outward = m1.w * (p - center);             // (1/radius) * v
    const bool front_face = dot(in.d, outward) < 0.0f;
    const V3 normal = front_face ? outward : -outward;
    const int type = __float_as_int(m1.z);
    const float roughness = m0.w, metallic = m1.x, ior = m1.y;
    if (type == 1) {                                    // Lambertian, Material.cpp:8-19
        V3 dir = normal + roughness * sample_unit_vector(u[0], u[1]);
        // near_zero (Vector3.cpp:138-141): the reference compares |x| promoted to double with the double 1e-8.
        // The largest float below 1e-8 is 0x322BCC77 (9.99999993922529e-09) and the next one is above it, so for
        // an fp32 x:  (double)|x| < 1e-8  <=>  |x| <= 0x322BCC77  (checked exhaustively in tests/test_emul_render.py).
        const float eps = __uint_as_float(0x322BCC77u);
        if (fabsf(dir.x) <= eps && fabsf(dir.y) <= eps && fabsf(dir.z) <= eps) dir = normal;
        out = {p, dir};
        return true;
    }
    if (type == 2) {                                    // Metal, Material.cpp:20-25
        const V3 dir = metallic * reflect(ud, normal);
        out = {p, dir + roughness * sample_in_unit_sphere(u[0], u[1], u[2])};
        return dot(out.d, normal) > 0.0f;
    }
    if (type == 3) {                                    // Glass, Material.cpp:26-48
        const float ratio = front_face ? (1.0f / ior) : ior;
        const float cos_theta = fminf(dot(-ud, normal), 1.0f);
        const float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
        const bool cannot_refract = ratio * sin_theta > 1.0f;
        float r0 = (1.0f - ratio) / (1.0f + ratio);     // Schlick, Material.cpp:28-33
        r0 = r0 * r0;
        const float q = 1.0f - cos_theta;
        const float q2 = q * q;
        const float reflectance = r0 + (1.0f - r0) * (q2 * q2 * q);
        const V3 dir = (cannot_refract || reflectance > u[3]) ? reflect(ud, normal) : refract(ud, normal, ratio);
        out = {p, dir};
        return true;
    }
    out = {p, V3{0.0f, 0.0f, 0.0f}};                    // Type::None, Material.cpp:49-52
    return false;
}

// main.cpp:69-71 + Camera.hpp:25-29 for pixel column x, image row from the top `row` (y = H-1-row).
__device__ __forceinline__ Ray primary_ray(const CameraDev& cam, int x, int row, int W, int H, const float u[4]) {
    const int y = H - 1 - row;
    const float s = (static_cast<float>(x) + u[0]) / static_cast<float>(W - 1);
    const float t = (static_cast<float>(y) + u[1]) / static_cast<float>(H - 1);
    float dkx, dky;
    sample_disk(u[2], u[3], dkx, dky);
    const float rdx = cam.lens_radius * dkx, rdy = cam.lens_radius * dky;
    const V3 offset = rdx * cam.u + rdy * cam.v;
    Ray r;
    r.o = cam.origin + offset;
    r.d = cam.llc + s * cam.horizontal + t * cam.vertical - cam.origin - offset;
    return r;
}

// Sky of main.cpp:98-100 for the unit direction ud = unit_vector(r.direction()).
__device__ __forceinline__ V3 sky_unit(const V3 ud) {
    const float t = 0.5f * (ud.y + 1.0f);
    return (1.0f - t) * V3{1.0f, 1.0f, 1.0f} + t * V3{0.5f, 0.7f, 1.0f};
}

} // namespace rtk
