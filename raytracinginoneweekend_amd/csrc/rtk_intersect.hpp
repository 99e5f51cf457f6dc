// rtk_intersect.hpp — the closest-hit machinery of the render and intersect kernels: filter forms, compaction,
// exact stage (Intersector::closest_hit) and the uniform-grid walk (Intersector::closest_hit_grid).
// Device code; also built for the CPU by tests/cpu_emul.  Math, RNG, scene structs and the exact ray-sphere test are in
// rtk_device.hpp; the per-wave render loop that calls this is rtk_render.hpp.
#pragma once

#include "rtk_device.hpp"

namespace rtk {

// Closest hit over the whole list == HittableList::hit (HittableList.hpp:30-42).
//
// Stage 1 (filter, all spheres): a conservative test — is the sphere's centre within r of the ray's line.
// LDS-resident scenes evaluate it as ONE dot product per test on the matrix pipe: squared distance minus r^2 is
// bilinear in monomials of the centre (static, the words' tiles) and coefficients of the ray (built per call); per word
// of 32 spheres and 64 rays one v_mfma_f32_32x32x16_f16 per 32 rays (two for words of mixed radii), and two vector FMAs
// per test turn the sign into a mask bit.  Larger scenes stream through a wave-private LDS chunk and test on the vector
// ALUs with a basis of the plane perpendicular to the ray (e1.y == 0): 7 FMAs + one funnel shift per test.
// Stage 2 (candidates): LDS-resident scenes keep the mask words themselves — a lane stores each non-zero word in its LDS
// slot and notes the word in a register bitmap (no loop, no vote per word); streamed scenes compact the bits into a
// per-lane list of 16-bit indices (a full list triggers an early exact pass).  Ascending list index either way.
// Stage 3 (exact): each lane walks its own candidates with the reference's arithmetic and range tests
// against its running `closest`; ascending order reproduces the reference's tie rule (a later
// sphere with root == closest replaces, Sphere3.hpp:43 `t_max < root`) and its NaN behaviour.
// ---- how the matrix instructions of the filter are issued (profiles/r03_hazard_notes.md) -------------
// Two v_mfma_f32_32x32x16_f16 of one wave that NAME THE SAME A REGISTERS must follow each other directly.  With a
// scalar-ALU instruction (s_cmp, s_mov, ...) or a waiting s_waitcnt issued between them, the render kernel computed a few
// hundred wrong candidate words per 1.7e8 rays once three or more waves shared a SIMD: frames differed from run to run
// and from the uniform-grid kernel's.  Found by editing single instructions of the built code object
// (tools/asm_variant.py, tools/hazard_hunt/): the shipped loop with one s_mov_b32 s, s moved between its two MFMAs
// fails, with s_nop or a vector move in the same place it is exact, and with the second MFMA reading a COPY of the A
// fragment in other registers — or the same registers rewritten onto themselves by a vector move in between — it is
// exact again.  Which of the loop's instructions the compiler's scheduler puts
// between two builtin MFMAs is not under the source's control (three source forms of rounds 2 scheduled the loop
// compare there), so the MFMAs of one word are ONE asm statement: nothing can be scheduled between them, the result
// tiles are early-clobber outputs (a 32x32 result must not overlap its A or B operand: the matrix pipe reads them over
// several passes) and the wait states an 8-pass matrix result needs before any reader (12) are part of the statement —
// the compiler's hazard recogniser does not look inside asm.  tests/test_cabi.py checks the built code object: every
// v_mfma is directly followed by another v_mfma or by the statement's s_nop.
// RTK_EXP_SCALAR_BETWEEN_MFMAS (experiment builds only, tools/build_variants.py): the known-bad form, for
// tests/test_gpu_render.py::test_known_bad_loop_form_is_caught.
#ifndef RTK_HOST_EMULATION
typedef _Float16 rtk_f16x8 __attribute__((ext_vector_type(8)));
typedef float rtk_f32x16 __attribute__((ext_vector_type(16)));
#ifdef RTK_EXP_SCALAR_BETWEEN_MFMAS
#define RTK_BETWEEN_MFMAS "s_mov_b32 s32, s32\n\t"
#else
#define RTK_BETWEEN_MFMAS ""
#endif
// d0 = A x B0, d1 = A x B1: one tile of 32 spheres against the two groups of 32 rays (K = 16)
__device__ __forceinline__ void mfma_pair(rtk_f32x16& d0, rtk_f32x16& d1, rtk_f16x8 a, rtk_f16x8 b0, rtk_f16x8 b1) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, 0\n\t" RTK_BETWEEN_MFMAS "v_mfma_f32_32x32x16_f16 %1, %2, %4, 0\n\ts_nop 11"
                 : "=&v"(d0), "=&v"(d1)
                 : "v"(a), "v"(b0), "v"(b1));
}
// the same for K = 32 (general words): two fragment sets chained through the C operand
__device__ __forceinline__ void mfma_quad(rtk_f32x16& d0, rtk_f32x16& d1, rtk_f16x8 a0, rtk_f16x8 a1, rtk_f16x8 b00, rtk_f16x8 b01, rtk_f16x8 b10, rtk_f16x8 b11) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %4, 0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %5, 0\n\t"
                 "v_mfma_f32_32x32x16_f16 %0, %3, %6, %0\n\tv_mfma_f32_32x32x16_f16 %1, %3, %7, %1\n\ts_nop 11"
                 : "=&v"(d0), "=&v"(d1)
                 : "v"(a0), "v"(a1), "v"(b00), "v"(b01), "v"(b10), "v"(b11));
}
#endif
// Scenes beyond kMaxLdsSpheres stream the sphere list through a wave-private LDS chunk (one coalesced global load per
// 64 spheres) instead of reading every sphere with a broadcast global load (measured on the 10k-sphere scene of
// BASELINE.json configs[3]: 1.56x the broadcast-load walk, profiles/r02_config4.md).
constexpr int kStageSpheres = 128;   // spheres per staged chunk (a multiple of 32)
#ifdef RTK_HOST_EMULATION
#define RTK_WAVE_SYNC() ((void)__ballot(true))      // the fibers of a wave are not in lockstep between votes
#else
#define RTK_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif

// Uniform grid over the scene (no counterpart in the reference, whose HittableList::hit is O(N)).
// Built on the host (rtk_scene_prep.hpp build_grid); opt-in (rtk_context_set_accel), for lists far beyond LDS size.
// Every "gridable" sphere is registered in each cell its box overlaps, the box being inflated by everything the
// reference's rounding could still accept for ray origins inside the ball (centre, r_allow); the few spheres that
// are not gridable (outliers such as the 1000-unit ground sphere) are tested for every ray; a scene with a non-finite
// sphere gets no grid at all (build_grid).
struct GridDev {
    const uint32_t* cell_start;   // [zones][nx*ny*nz + 1]: one registration per zone of ray origins
    const uint32_t* items;        // sphere indices, ascending within a cell
    const uint32_t* always;       // [n_always] ascending: spheres outside the grid
    int32_t n_always;
    int32_t nx, ny, nz;
    float lox, loy, loz;          // grid box
    float hix, hiy, hiz;
    float hx, hy, hz;             // cell size
    float ihx, ihy, ihz;          // 1 / cell size
    int32_t znx, zny, znz;        // origin zones over the grid box (outer zones extend outwards)
    float izx, izy, izz;          // zones per unit length
    int32_t zone_stride;          // nx*ny*nz + 1
    float cx, cy, cz, r_allow2;   // ray origins inside this ball may use the grid; others take the brute-force path
    float slack;                  // distance by which traversal decisions are widened (float error of the walk)
};

template <bool LDS>
struct Intersector {
    const float4* filt;    // LDS == true: the words' tiles and the short tail's records in LDS; else SceneDev::filt in global memory
    const float4* wdesc = nullptr;   // LDS (with LDS == true) or null: word descriptors, SceneDev::wdesc
    const float4* exact;   // LDS or global (cx, cy, cz, r*r)
    uint16_t* list;        // !LDS: this wave's candidate list in LDS: [kListCap][64]
    uint32_t* masks = nullptr;   // LDS: this wave's candidate mask words in LDS: [npad / 32][64]
    float4* stage = nullptr;     // !LDS: this wave's LDS chunk of kStageSpheres filter operands
    const float4* ltiles = nullptr;   // !LDS: word-local tiles (SceneDev::ltiles) or null: the staged vector-ALU walk
    const float4* lwdesc = nullptr;   // !LDS: their descriptors
    int lwords = 0;                   // !LDS: words of the local cut
    uint2* entries = nullptr;         // !LDS local walk: this wave's pending candidates in LDS: [kEntryCap][64] (mask word, first list index); shares the memory of list and stage
    FilterBounds fb;
    int n;
    int lane;
#ifdef RTK_HOST_EMULATION
    uint8_t* dbg_flags = nullptr;   // emulation only: [n] set to 1 for every sphere the filter passes to the exact stage
#endif
#ifdef RTK_PROFILE_SECTIONS
    mutable unsigned long long prof[4] = {0, 0, 0, 0};     // profile builds only (tools/build_variants.py name=-DRTK_PROFILE_SECTIONS=1): shader cycles in the filter, in the exact stage; streamed local walk: class words, kind-0 words, compaction (with early exact passes), final exact pass
#endif
#ifdef RTK_PROFILE_WORD
    mutable unsigned long long wp[6] = {0, 0, 0, 0, 0, 0};     // profile builds only: parts of a class word of the streamed local walk
#endif

    // `list_base`: list entries are 16-bit offsets from it (the list is flushed before an offset could overflow).
    __device__ __forceinline__ void flush(const Ray& r, float a, float t_min, int list_base, int& cnt, float& closest, int& best,
                                          uint32_t& n_exact) const {
        for (int k = 0; wave_any(k < cnt); ++k) {
            RTK_STAT(lane, flush_iters);
            n_exact += static_cast<uint32_t>(__popcll(ballot(k < cnt)));
            if (k < cnt) {
                const int j = list_base + static_cast<int>(list[k * kWave + lane]);
                const float4 s = exact[j];
                float root;
                if (sphere_root_exact(r, a, s, t_min, closest, root)) {
                    closest = root;
                    best = j;
                }
                
            }
        }
        cnt = 0;
    }

    // Per-ray operands of the filter: the shrunk basis (e1, e2) of the plane perpendicular to the ray, e1.y == 0, and -o.e.
    struct FilterRay {
        float e1x, e1z, ns1, e2x, e2y, e2z, ns2;
    };

    // One ray-sphere filter test; shifts the result bit into `m` (1 = candidate).
    static __device__ __forceinline__ void filter_one(const FilterRay& f, float sx, float sy, float sz, float sw, uint32_t& m) {
        const float p1 = fmaf(sz, f.e1z, fmaf(sx, f.e1x, f.ns1));
        const float p2 = fmaf(sz, f.e2z, fmaf(sy, f.e2y, fmaf(sx, f.e2x, f.ns2)));
        const float dd = fmaf(p2, p2, fmaf(p1, p1, -sw));
        m = __builtin_amdgcn_alignbit(m, __float_as_uint(dd), 31);
    }
#ifndef RTK_HOST_EMULATION
    // 16 results of a lane -> 16 mask bits, first sphere in the highest bit: registers 2 m hold spheres 16 hf + m,
    // registers 2 m + 1 spheres 16 hf + 8 + m; one fp32 accumulator (16 bits: exact).  X = clamp(-2^48 D + k) is exactly 1.0
    // for a candidate, else 0.0.
    static __device__ __forceinline__ uint32_t mfma_bits16(const rtk_f32x16& d, float k) {
        const float nbig = -0x1p48f;
        float acc = __builtin_amdgcn_fmed3f(fmaf(d[0], nbig, k), 0.0f, 1.0f);
#pragma unroll
        for (int e = 1; e < 16; ++e) {
            const int reg = e < 8 ? 2 * e : 2 * (e - 8) + 1;
            acc = fmaf(acc, 2.0f, __builtin_amdgcn_fmed3f(fmaf(d[reg], nbig, k), 0.0f, 1.0f));
        }
        return static_cast<uint32_t>(acc);
    }
    // h0 / h1: the 16 bits of this lane's spheres for the ray of lane (lane & 31) / (lane | 32).  After the swap a
    // lane holds both halves of its own ray's word: spheres 0..15 (from the lower lane half) and 16..31.
    static __device__ __forceinline__ uint32_t mfma_join(uint32_t h0, uint32_t h1) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(h0, h1, false, false);
        return (sw[0] << 16) | sw[1];
    }
#endif
    // `n_exact`: the WAVE's running total of exact tests (every lane holds the same number).
    // ALLOW_LOCAL = false (the grid walk's fallback for lanes the grid cannot serve): streamed scenes take the staged walk even
    // when they have local tiles, so that the grid kernels do not carry the word-local walk's registers.
    template <bool ALLOW_LOCAL = true>
    __device__ __forceinline__ void closest_hit(const Ray& r, bool active, float t_min, float t_max, float& t_out, int& idx_out,
                                                uint32_t& n_exact) const {
        // Reference `a` (Sphere3.hpp:33), hoisted.
        const float a = length_squared(r.d);
        // A lane is "safe" when no intermediate of the filter can overflow, underflow to a wrong sign
        // or be NaN; unsafe lanes (NaN/inf/huge/zero-length rays) treat every sphere as a candidate.
        const float big = 0x1p30f;
        FilterRay f = {};
        const float inv_len = __builtin_amdgcn_rsqf(a);
        // Orthonormal basis (e1, e2) of the plane perpendicular to the ray with e1 horizontal:
        //   e1 = (dz, 0, -dx) / h,  e2 = d x e1 = (-dx dy / h, h, -dy dz / h),  h = sqrt(dx^2 + dz^2)   (d = unit direction).
        // No cancellation anywhere: every component is a product of rounded values, so e1 and e2 are orthogonal to d to
        // a few ulp whatever h is; only a ray within 2^-20 of the y axis (h^2 < 2^-40) has no usable horizontal
        // direction and tests every sphere.  A sphere can only be hit if its centre, projected on that plane, lies
        // within r of the ray.  The projected distances are shrunk by g < 1 so that everything the reference's own
        // rounding could still accept (E) and this filter's evaluation error (dp) stay inside r.
        // (Scenes beyond LDS only; LDS-resident scenes use the dot-product form below.)
        const float dx = r.d.x * inv_len, dy = r.d.y * inv_len, dz = r.d.z * inv_len;
        const float h2 = fmaf(dx, dx, dz * dz);
        const float inv_h = __builtin_amdgcn_rsqf(h2);
        const float o_len = __builtin_amdgcn_sqrtf(fmaf(r.o.x, r.o.x, fmaf(r.o.y, r.o.y, r.o.z * r.o.z)));
        // the dot-product form scales the ray's coefficients by 2^-2k, 2^k > sigma |o|, k <= 12
        const bool safe = (fabsf(r.o.x) < big) && (fabsf(r.o.y) < big) && (fabsf(r.o.z) < big) && (a > 0x1p-80f) && (a < 0x1p80f) &&
                          (LDS ? fb.sigma * o_len < 0x1p12f : h2 > 0x1p-40f);
        if (!LDS) {
            const float u1x = dz * inv_h, u1z = -dx * inv_h;                     // unshrunk basis
            const float u2x = -dy * (dx * inv_h), u2y = h2 * inv_h, u2z = -dy * (dz * inv_h);
            const float M = o_len + fb.c_max;
            const float E = 0x1p-19f * fmaf(M, M, fb.r_max2);
            const float dp = 0x1p-19f * M;          // this form's own error in a projected coordinate: three FMA roundings
            const float g = fb.r_min * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(fb.r_min, fb.r_min, E)) + 2.0f * dp) * (1.0f - 0x1p-16f);
            f.e1x = g * u1x; f.e1z = g * u1z;
            f.e2x = g * u2x; f.e2y = g * u2y; f.e2z = g * u2z;
            f.ns1 = -fmaf(r.o.z, f.e1z, r.o.x * f.e1x);
            f.ns2 = -fmaf(r.o.z, f.e2z, fmaf(r.o.y, f.e2y, r.o.x * f.e2x));
        }
        RTK_STAT(lane, closest_hit_calls);
#ifdef RTK_PROFILE_SECTIONS
        const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
        float closest = t_max;
        int best = -1;
        int cnt = 0;
        int list_base = 0;       // wave-uniform

        // compaction: candidate bits (MSB = first sphere of the word) -> this lane's list, ascending index;
        // when a lane's list is full the wave runs the exact stage early and carries on
        auto emit = [&](int base, uint32_t cand) {
#ifdef RTK_HOST_EMULATION
            if (dbg_flags)
                for (int b = 0; b < 32; ++b)
                    if (cand & (0x80000000u >> b)) dbg_flags[base + b] = 1;
#endif
            if (!wave_any(cand != 0)) return;          // most words of a large scene have no candidate on any lane: one vote
            for (;;) {
                while (wave_any(cand != 0 && cnt < kListCap)) {
                    RTK_STAT(lane, compact_iters);
                    if (cand != 0 && cnt < kListCap) {
                        const int pos = __clz(static_cast<int>(cand));
                        cand &= ~(0x80000000u >> pos);
                        list[cnt * kWave + lane] = static_cast<uint16_t>(base + pos - list_base);
                        ++cnt;
                    }
                }
                if (!wave_any(cand != 0)) break;
                RTK_STAT(lane, early_flushes);
                flush(r, a, t_min, list_base, cnt, closest, best, n_exact);
            }
        };
        auto to_cand = [&](uint32_t m, uint32_t valid) -> uint32_t {
            uint32_t cand = (safe ? m : 0xFFFFFFFFu) & valid;
            return active ? cand : 0u;
        };

        // filter + compaction over `count` consecutive spheres starting at list index `base`, all general words;
        // load(i) yields the operands of sphere base + i
        auto scan = [&](auto&& load, int base, int count) {
            const int full = count >> 5, rem = count & 31;
            for (int w = 0; w < full; ++w) {
                if (base + w * 32 - list_base > 65535 - 64) {     // scenes beyond 64k spheres: re-base the 16-bit list offsets
                    flush(r, a, t_min, list_base, cnt, closest, best, n_exact);
                    list_base = base + w * 32;
                }
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    const auto s = load(w * 32 + i);
                    filter_one(f, s.x, s.y, s.z, s.w, m);
                }
                RTK_STAT(lane, filter_words);
                emit(base + w * 32, to_cand(m, 0xFFFFFFFFu));
            }
            if (rem) {
                uint32_t m = 0;
#pragma unroll 4
                for (int i = 0; i < rem; ++i) {
                    const auto s = load(full * 32 + i);
                    filter_one(f, s.x, s.y, s.z, s.w, m);
                }
                m <<= (32 - rem);
                emit(base + full * 32, to_cand(m, 0xFFFFFFFFu << (32 - rem)));
            }
        };
        // The LDS-resident scene.  With C' = sigma C, o' = sigma o, d the unit direction, p = o' - (o'.d) d and
        // q = I - d d^T, a sphere can only be hit if  q(C', C') - 2 p.C' + |p|^2 - r'^2 <= (what the reference's rounding
        // adds).  The left side, times a power of two t per ray, is a dot product of the centre's monomials (the word's
        // tile, rtk_device.hpp quad_term) and the ray's coefficients, each split into two f16 parts: one
        // v_mfma_f32_32x32x16_f16 yields it for 32 spheres x 32 rays (two chained for general words), in fp32.  A lane then
        // holds the values D of 16 spheres for one ray; per sphere two vector FMAs remain: X = clamp(-2^48 (D + K)), K the
        // ray's constant including the margin (exactly 1.0 for a candidate, else 0.0: every D and K is a multiple of
        // 2^-48), and acc = 2 acc + X.  The two halves of a ray's mask word sit in lanes l and l ^ 32; one
        // v_permlane32_swap brings them together.  A lane keeps every non-zero mask word in its LDS slot and the set of
        // those words in `nz` (bit 31 - w: at most 32 words, kMaxLdsSpheres = 1024); then it walks its own candidates in
        // ascending list order.
        auto scan_words = [&]() {
            const int full = n >> 5, rem = n & 31;
            const bool short_tail = rem != 0 && rem <= kShortTail;
            const int words = short_tail ? full : full + (rem ? 1 : 0);       // words that have a tile
            uint32_t nz = 0;
            auto keep = [&](int w, uint32_t m) {          // the store is unconditional: nearly every word has a candidate on some lane
                masks[w * kWave + lane] = m;
                if (m != 0) nz |= 0x80000000u >> w;
            };
            const bool has_class = (fb.kinds & 1) != 0, has_general = (fb.kinds & 2) != 0;       // wave-uniform
            // ---- the ray's side of the dot product ----
            // t = 2^-2k with 2^k > |o'| (k >= 0) keeps every coefficient below 4: |q_ij| t <= 1, |2 p| t < 2.
            const float sg = fb.sigma;
            const float ox = r.o.x * sg, oy = r.o.y * sg, oz = r.o.z * sg;
            const float ol = o_len * sg;
            int ke = __builtin_amdgcn_frexp_expf(ol);
            ke = ke < 0 ? 0 : (ke > 12 ? 12 : ke);                           // beyond 12 the lane is not `safe`
            const float tk = __builtin_ldexpf(1.0f, -ke), t = tk * tk;
            const float sd = fmaf(oz, dz, fmaf(oy, dy, ox * dx));
            const float px = fmaf(-sd, dx, ox), py = fmaf(-sd, dy, oy), pz = fmaf(-sd, dz, oz);
            const float pp = fmaf(pz, pz, fmaf(py, py, px * px));
            // q without cancellation: 1 - dx^2 = dy^2 + dz^2 (|d| = 1 to a few ulp: part of the margin)
            const float qxx = fmaf(dy, dy, dz * dz), qyy = fmaf(dx, dx, dz * dz), qzz = fmaf(dx, dx, dy * dy);
            const float qxy = -2.0f * dx * dy, qxz = -2.0f * dx * dz, qyz = -2.0f * dy * dz;
            // Margin, in the same t-scaled units.  T bounds the sum of the absolute values of all terms and the constant
            // (|C'| <= c' = sigma c_max <= 1; sum |q_ij C'_i C'_j| <= sqrt(2) c'^2; w = |o'| 2^-k < 1), T1 the reference's
            // own |o - C|^2 + r^2:  what the reference's rounding could still accept is 2^-19 T1, this
            // evaluation is within acc_eps T + 2^-22 of the true value (rtk_device.hpp kQuadEps*; 2^-22: f16 parts below
            // the normal range, should the matrix pipe flush them).
            const float w = ol * tk;
            const float T = fmaf(fb.t_quad, t, fmaf(fb.cp_twice, tk, w) * w);
            const float T1 = fmaf(fmaf(tk, fb.cp, w), fmaf(tk, fb.cp, w), fb.rp2 * t);
            const float marg = 1.25f * fmaf(fb.acc_eps, T, 0x1p-19f * T1) + 0x1p-21f;
            const float tb = t * kQuadRayScale;
            // constants: K t 2^27, pushed away from zero so that it is a multiple of 2^-48, then times -2^48
            auto constant = [&](float k_t) {
                float ks = k_t * (kQuadSphereScale * kQuadRayScale);
                ks = fabsf(ks) < 0x1p-24f ? -0x1p-24f : ks;
                return ks * -0x1p48f;
            };
            const float cy = fb.cls_cy;
            const float kc = constant(fmaf(fmaf(-2.0f * py, cy, fmaf(qyy, fb.cls_cy2, pp)) - fb.cls_r2, t, -marg));
            const float kt = fmaf(pp, t, -marg);          // the general constant, t-scaled
            const float kg = constant(kt);
            // term vectors (quad_term): class xx zz xz x z; general xx yy zz xy xz yz x y z, then r'^2 against -t
            uint16_t vc[16], vg[32];
            if (has_class) {
                const float co[kQuadClassMonomials] = {qxx * tb, qzz * tb, qxz * tb, fmaf(qxy, cy, -2.0f * px) * tb, fmaf(qyz, cy, -2.0f * pz) * tb};
                uint16_t parts[kQuadClassMonomials][2];
#pragma unroll
                for (int m = 0; m < kQuadClassMonomials; ++m) f16_split2(co[m], parts[m]);
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    int mono, sp, rp;
                    quad_term(k, kQuadClassMonomials, mono, sp, rp);
                    vc[k] = mono < 0 ? static_cast<uint16_t>(0) : parts[mono][rp];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) vc[k] = 0;
            }
            // the general coefficients, t-scaled, in fp32: for general tiles (times 2^13, split) and for the short tail (as they are)
            const float cg[kQuadGeneralMonomials + 1] = {qxx * t, qyy * t, qzz * t, qxy * t, qxz * t, qyz * t, -2.0f * px * t, -2.0f * py * t, -2.0f * pz * t, -t};
            if (has_general) {
                uint16_t parts[kQuadGeneralMonomials + 1][2];
#pragma unroll
                for (int m = 0; m <= kQuadGeneralMonomials; ++m) f16_split2(cg[m] * kQuadRayScale, parts[m]);
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    int mono, sp, rp;
                    quad_term(k, kQuadGeneralMonomials, mono, sp, rp);
                    vg[k] = mono < 0 ? static_cast<uint16_t>(0) : parts[mono][rp];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 32; ++k) vg[k] = 0;
            }
            if (short_tail) {          // before the word loops: the fp32 coefficients need not live through them
                // the few spheres of the last word, on the vector ALUs: the same dot product from fp32 monomials (three
                // float4 per sphere: xx yy zz xy | xz yz x y | z r'^2 - -) and the fp32 coefficients, 11 FMAs, each
                // within 2^-24 T: inside the margin's allowance for the matrix pipe's 16 steps; the sign is the result.
                // A sphere the filter must not reject has r'^2 = +inf.
                const float4 wd = wdesc[full];
                const float4* q = filt + __builtin_amdgcn_readfirstlane(__float_as_int(wd.x));
                uint32_t m = 0;
                for (int i = 0; i < rem; ++i) {
                    const float4 ma = q[3 * i], mb = q[3 * i + 1], mc = q[3 * i + 2];
                    float acc = fmaf(ma.x, cg[0], kt);
                    acc = fmaf(ma.y, cg[1], acc); acc = fmaf(ma.z, cg[2], acc); acc = fmaf(ma.w, cg[3], acc);
                    acc = fmaf(mb.x, cg[4], acc); acc = fmaf(mb.y, cg[5], acc); acc = fmaf(mb.z, cg[6], acc); acc = fmaf(mb.w, cg[7], acc);
                    acc = fmaf(mc.x, cg[8], acc); acc = fmaf(mc.y, cg[9], acc);
                    m = __builtin_amdgcn_alignbit(m, __float_as_uint(acc), 31);
                }
                RTK_STAT(lane, tail_tests);
                keep(full, (m << (32 - rem)) | __float_as_uint(wd.z));
            }
#ifdef RTK_HOST_EMULATION
            // CPU build: the same part products summed in fp32, sphere by sphere (the matrix pipe's own summation order is
            // not modelled: the margin covers any)
            for (int wi = 0; wi < words; ++wi) {
                const float4 wd = wdesc[wi];
                const bool cls = __float_as_int(wd.y) != 0;
                const float4* tile = filt + __float_as_int(wd.x);
                const int kk = cls ? 16 : 32;
                uint32_t m = 0;
                for (int rho = 0; rho < 32; ++rho) {
                    const int i = tile_row_sphere(rho);
                    float dsum = 0.0f;
                    double exact_sum = 0.0, abs_sum = 0.0;          // exact in double: at most 32 products of 22 bits
                    for (int k = 0; k < kk; ++k) {
                        const uint16_t* frag = reinterpret_cast<const uint16_t*>(tile + 64 * (k >> 4) + 32 * ((k >> 3) & 1) + rho);
                        const float ak = f16_value(frag[k & 7]), bk = f16_value(cls ? vc[k] : vg[k]);
                        dsum = fmaf(ak, bk, dsum);
                        exact_sum += static_cast<double>(ak) * bk;
                        abs_sum += std::fabs(static_cast<double>(ak) * bk);
                    }
                    // with hardware noise switched on: the worst accumulation the margin allows for (kk steps of 2^-23 of
                    // the absolute sum at noise level 3), pushed towards "no candidate"
                    if (rtk_shim::ulp_noise > 0 && abs_sum < 1e30)
                        dsum = static_cast<float>(exact_sum + rtk_shim::ulp_noise / 3.0 * kk * 0x1p-23 * abs_sum) + 0.0f;
                    const float x = fmaf(dsum, -0x1p48f, cls ? kc : kg);
                    if (x > 0.0f && x < 1.0f) RTK_STAT_EACH(clamp_fractions);
                    if (x > 0.0f) m |= 0x80000000u >> i;
                }
                if (cls) RTK_STAT(lane, class_words);
                RTK_STAT(lane, filter_words);
                keep(wi, (m & __float_as_uint(wd.w)) | __float_as_uint(wd.z));
            }
#else
            // B fragments: lane l supplies B[k = 8 (l >> 5) + j][column l & 31].  For the first 32 rays lanes 0..31 give
            // their own k 0..7 and lanes 32..63 the k 8..15 of the ray of lane l - 32; for the second 32 rays the other
            // way round: exactly what v_permlane32_swap(lo, hi) yields.  The constants travel the same way.
            typedef rtk_f16x8 f16x8;
            typedef rtk_f32x16 f32x16;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            auto fragments = [](const uint16_t* v, f16x8 out[2]) {          // 16 terms -> the B operands of the two ray groups
                u32x4 g0, g1;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t lo = v[2 * d] | (static_cast<uint32_t>(v[2 * d + 1]) << 16), hi = v[8 + 2 * d] | (static_cast<uint32_t>(v[9 + 2 * d]) << 16);
                    const u32x2 sw = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
                    g0[d] = sw[0]; g1[d] = sw[1];
                }
                __builtin_memcpy(&out[0], &g0, 16); __builtin_memcpy(&out[1], &g1, 16);
            };
            float kcg[2], kgg[2];
            {
                const u32x2 a1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(kc), __float_as_uint(kc), false, false);
                const u32x2 a2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(kg), __float_as_uint(kg), false, false);
                kcg[0] = __uint_as_float(a1[0]); kcg[1] = __uint_as_float(a1[1]);
                kgg[0] = __uint_as_float(a2[0]); kgg[1] = __uint_as_float(a2[1]);
            }
            auto bits16 = [](const f32x16& d, float k) { return mfma_bits16(d, k); };
            auto join = [](uint32_t h0, uint32_t h1) { return mfma_join(h0, h1); };
            if (!has_general) {
                // every tile is a class tile: word w's fragments are filt[64 w ...]
                f16x8 bc[2];
                fragments(vc, bc);
                auto word = [&](int wi) {
                    const float4 wd = wdesc[wi];
                    const float4 av = filt[wi * kWave + lane];
                    f16x8 a0;
                    __builtin_memcpy(&a0, &av, 16);
                    f32x16 d0, d1;
                    mfma_pair(d0, d1, a0, bc[0], bc[1]);
                    const uint32_t m = join(bits16(d0, kcg[0]), bits16(d1, kcg[1]));
                    RTK_STAT(lane, class_words);
                    RTK_STAT(lane, filter_words);
                    keep(wi, (m & __float_as_uint(wd.w)) | __float_as_uint(wd.z));
                };
                // five words per trip (the compiler does not unroll a loop around an asm statement by itself): the LDS addresses of
                // the fragment, the descriptor and the mask slot become immediate offsets, 67.0 instead of 68.3 ms per frame
                constexpr int kUnroll = 5;
                int wi = 0;
                for (; wi + kUnroll <= words; wi += kUnroll) {
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) word(wi + u);
                }
                for (; wi < words; ++wi) word(wi);
            } else {
                f16x8 bc[2], bg0[2], bg1[2];
                fragments(vc, bc);
                fragments(vg, bg0);
                fragments(vg + 16, bg1);
                for (int wi = 0; wi < words; ++wi) {
                    const float4 wd = wdesc[wi];
                    const bool cls = __builtin_amdgcn_readfirstlane(__float_as_int(wd.y)) != 0;
                    const float4* tile = filt + __builtin_amdgcn_readfirstlane(__float_as_int(wd.x));
                    const float4 av = tile[lane];
                    f16x8 a0;
                    __builtin_memcpy(&a0, &av, 16);
                    f32x16 d0, d1;
                    float k0, k1;
                    if (cls) {
                        mfma_pair(d0, d1, a0, bc[0], bc[1]);
                        k0 = kcg[0]; k1 = kcg[1];
                    } else {
                        const float4 av1 = tile[kWave + lane];
                        f16x8 a1;
                        __builtin_memcpy(&a1, &av1, 16);
                        mfma_quad(d0, d1, a0, a1, bg0[0], bg0[1], bg1[0], bg1[1]);
                        k0 = kgg[0]; k1 = kgg[1];
                    }
                    const uint32_t m = join(bits16(d0, k0), bits16(d1, k1));
                    if (cls) RTK_STAT(lane, class_words);
                    RTK_STAT(lane, filter_words);
                    keep(wi, (m & __float_as_uint(wd.w)) | __float_as_uint(wd.z));
                }
            }
#endif
            // lanes whose ray the filter cannot judge test every sphere (rare: one vote per ray, outside the word loop)
            if (wave_any(active && !safe)) {
                if (!safe) {
                    for (int w = 0; w < full; ++w) masks[w * kWave + lane] = 0xFFFFFFFFu;
                    if (rem) masks[full * kWave + lane] = 0xFFFFFFFFu << (32 - rem);
                    const int words = full + (rem ? 1 : 0);
                    nz = words >= 32 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> words);
                }
            }
            if (!active) nz = 0;
#ifdef RTK_HOST_EMULATION
            if (dbg_flags)
                for (int w = 0; w < full + (rem ? 1 : 0); ++w)
                    if (nz & (0x80000000u >> w))
                        for (int b = 0; b < 32; ++b)
                            if (masks[w * kWave + lane] & (0x80000000u >> b)) dbg_flags[w * 32 + b] = 1;
#endif
#ifdef RTK_PROFILE_SECTIONS
            const unsigned long long pt1 = __builtin_readcyclecounter();
            prof[0] += pt1 - pt0;
#endif
            // exact stage: every lane pops its own candidates, lowest list index first
            uint32_t m = 0;
            int wb = 0;
            while (wave_any((m | nz) != 0)) {
                RTK_STAT(lane, flush_iters);
                if (m == 0 && nz != 0) {          // this lane's next non-empty word
                    const int w = __clz(static_cast<int>(nz));
                    nz &= ~(0x80000000u >> w);
                    m = masks[w * kWave + lane];
                    wb = w << 5;
                }
                n_exact += static_cast<uint32_t>(__popcll(ballot(m != 0)));
                if (m != 0) {
                    const int pos = __clz(static_cast<int>(m));
                    m &= ~(0x80000000u >> pos);
                    const int j = wb + pos;
                    const float4 s = exact[j];
                    float root;
                    if (sphere_root_exact(r, a, s, t_min, closest, root)) {
                        closest = root;
                        best = j;
                    }
                    
                }
            }
#ifdef RTK_PROFILE_SECTIONS
            prof[1] += __builtin_readcyclecounter() - pt1;
#endif
        };
        // Scenes beyond LDS whose words have local tiles (rtk_device.hpp "word-local frames"): per word the ray's linear
        // coefficients and constant are rebuilt in the word's frame (~45 vector instructions), then the class word's two MFMAs
        // and 64 FMAs as above; the A fragment comes straight from global memory (16 bytes per lane, the next word's requested
        // a word ahead), the descriptor through the scalar cache.  Words without a tile (kind 0: mixed radii, the list's
        // ragged end) run the projection form on the vector ALUs with the spheres as scalar operands.
        auto scan_local = [&]() {
            const bool safe_ray = (fabsf(r.o.x) < big) && (fabsf(r.o.y) < big) && (fabsf(r.o.z) < big) && (a > 0x1p-80f) && (a < 0x1p80f);
            // |v|^2 = sigma_w^2 |o - c_w|^2 stays below kLocalSafeVV = 2^14 in every word while (|o| + c_max) sigma_max < 120: then the
            // per-word check of far origins is skipped (one vote per pass instead of one per word)
#ifdef RTK_HOST_EMULATION
            const bool any_far = wave_any(active && !((o_len + fb.c_max) * fb.lsig_max < 120.0f));
#else
            // (through readfirstlane: a scalar the compiler branches on, instead of folding it into the per-lane condition
            // below and issuing that block under an empty EXEC for every word)
            int any_far = __builtin_amdgcn_readfirstlane(wave_any(active && !((o_len + fb.c_max) * fb.lsig_max < 120.0f)) ? 1 : 0);
            asm volatile("" : "+s"(any_far));
#endif
            const int words = lwords;
            // Pending candidates: one (mask word, first list index) entry per word that has any, written without a vote
            // (every lane writes its slot, only lanes with a candidate advance); the exact stage runs when a lane could run
            // out of slots within the next eight words, and at the end.  Ascending list order throughout: the tie rule.
            int ecnt = 0;
            auto push = [&](int base, uint32_t cand) {
#ifdef RTK_HOST_EMULATION
                if (dbg_flags)
                    for (int b = 0; b < 32; ++b)
                        if (cand & (0x80000000u >> b)) dbg_flags[base + b] = 1;
#endif
                entries[ecnt * kWave + lane] = make_uint2(cand, static_cast<uint32_t>(base));
                ecnt += cand != 0 ? 1 : 0;
            };
            auto drain = [&]() {
                uint32_t m = 0;
                int wb = 0, e = 0;
                while (wave_any(m != 0 || e < ecnt)) {
                    RTK_STAT(lane, flush_iters);
                    if (m == 0 && e < ecnt) {
                        const uint2 en = entries[e * kWave + lane];
                        m = en.x;
                        wb = static_cast<int>(en.y);
                        ++e;
                    }
                    n_exact += static_cast<uint32_t>(__popcll(ballot(m != 0)));
                    if (m != 0) {
                        const int pos = __clz(static_cast<int>(m));
                        m &= ~(0x80000000u >> pos);
                        const int j = wb + pos;
                        const float4 s = exact[j];
                        float root;
                        if (sphere_root_exact(r, a, s, t_min, closest, root)) {
                            closest = root;
                            best = j;
                        }
                    }
                }
                ecnt = 0;
            };
            // the quadratic coefficients: the ray's alone
            const float qc[3] = {fmaf(dy, dy, dz * dz) * kQuadRayScale, fmaf(dx, dx, dy * dy) * kQuadRayScale, -2.0f * dx * dz * kQuadRayScale};
            uint16_t qp[3][2];
#pragma unroll
            for (int m = 0; m < 3; ++m) f16_split2(qc[m], qp[m]);
            // B terms of this ray, k = 0..15 (local_term): the linear ones are filled in per word
            auto term = [&](int k, const uint16_t (&lp)[2][2]) -> uint32_t {
                int mono, sp, rp;
                local_term(k, mono, sp, rp);
                return mono < 0 ? 0u : (mono < 3 ? qp[mono][rp] : lp[mono - 3][rp]);
            };
            const uint16_t zero_l[2][2] = {{0, 0}, {0, 0}};
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                lo[d] = term(2 * d, zero_l) | (term(2 * d + 1, zero_l) << 16);
                hi[d] = term(8 + 2 * d, zero_l) | (term(9 + 2 * d, zero_l) << 16);
            }
#ifdef RTK_HOST_EMULATION
            typedef const float4* ConstQuads;
            const ConstQuads desc = lwdesc, src = filt;
#else
            typedef float Quad __attribute__((ext_vector_type(4)));
            typedef const Quad __attribute__((address_space(4))) * ConstQuads;
            const ConstQuads desc = (ConstQuads)(lwdesc), src = (ConstQuads)(filt);
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            // B fragments: lane l supplies B[k = 8 (l >> 5) + j][column l & 31]; dword d of the two ray groups is
            // v_permlane32_swap(lo[d], hi[d]).  Dwords 0 and 1 hold quadratic terms only: swapped once per pass.
            u32x4 g0, g1;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const u32x2 sw = __builtin_amdgcn_permlane32_swap(lo[d], hi[d], false, false);
                g0[d] = sw[0]; g1[d] = sw[1];
            }
            float4 av_next = ltiles[lane];
#endif
            // the next word's descriptor is requested before this word's work, its tile fragment likewise
            auto n0 = desc[0], n1 = desc[1], n2 = desc[2];
            // lanes the filter cannot judge take every sphere of a word, inactive lanes none: masks, set once per pass
            const uint32_t force_class = safe_ray ? 0u : 0xFFFFFFFFu, force_kind0 = safe ? 0u : 0xFFFFFFFFu, keep = active ? 0xFFFFFFFFu : 0u;
            for (int w = 0; w < words; ++w) {
#ifdef RTK_PROFILE_WORD
                const unsigned long long qtop = __builtin_readcyclecounter();
#endif
                const auto d0 = n0, d1 = n1, d2 = n2;
#ifndef RTK_HOST_EMULATION
                // scalar loads return out of order, so the only wait there is drains them all: wait for this word's descriptor
                // BEFORE the next one is requested, or the wait at its first use would take the new request with it
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
#endif
                {
                    const int wn = w + 1 < words ? w + 1 : w;
                    n0 = desc[3 * wn];
                    n1 = desc[3 * wn + 1];
                    n2 = desc[3 * wn + 2];
                }
#ifndef RTK_HOST_EMULATION
                __builtin_amdgcn_sched_barrier(0);
                const float4 av = av_next;
                av_next = ltiles[(w + 1 < words ? w + 1 : w) * 64 + lane];
                const int start_kind = __builtin_amdgcn_readfirstlane(__float_as_int(d1.y));
#else
                const int start_kind = __float_as_int(d1.y);
#endif
                const int count = __builtin_popcount(__float_as_uint(d1.w));
                const bool cls = (start_kind & 1) != 0;
                const int base = start_kind >> 1;
#ifdef RTK_PROFILE_WORD
                const unsigned long long q0 = __builtin_readcyclecounter();
                wp[0] += q0 - qtop;
#endif
#ifdef RTK_PROFILE_SECTIONS
                const unsigned long long pw0 = __builtin_readcyclecounter();
#endif
                const uint32_t always = __float_as_uint(d1.z), valid = __float_as_uint(d1.w);
                uint32_t m = 0;
                if (cls) {
                    // the ray in the word's frame, UNSCALED (sigma_w and the fixed-point scales are folded into the word's constants)
                    const float vx = r.o.x - d0.x, vy = r.o.y - d0.y, vz = r.o.z - d0.z;
                    const float sd = fmaf(vz, dz, fmaf(vy, dy, vx * dx));
                    const float px = fmaf(-sd, dx, vx), py = fmaf(-sd, dy, vy), pz = fmaf(-sd, dz, vz);
                    const float pp = fmaf(pz, pz, fmaf(py, py, px * px));
                    const float vv = fmaf(sd, sd, pp);
                    // constant: -2^48 2^27 (|p|^2 - r~'^2 - marg) in scaled units; pushed away from zero so that it is an integer
                    float kc = fmaf(fmaf(vv, kLocalRatio, pp), d2.x, d1.x);
                    kc = fabsf(kc) < 0x1p24f ? 0x1p24f : kc;
                    // linear coefficients -2 sigma p 2^13 from p clamped to 2 scaled units, in two f16 parts
                    const float lim = d0.w;
#ifdef RTK_HOST_EMULATION
                    const float lx = fminf(fmaxf(px, -lim), lim) * d2.y, lz = fminf(fmaxf(pz, -lim), lim) * d2.y;
#else
                    const float lx = __builtin_amdgcn_fmed3f(px, -lim, lim) * d2.y, lz = __builtin_amdgcn_fmed3f(pz, -lim, lim) * d2.y;
#endif
#ifdef RTK_HOST_EMULATION
                    uint16_t lp[2][2];
                    f16_split2(lx, lp[0]);
                    f16_split2(lz, lp[1]);
                    lo[2] = term(4, lp) | (term(5, lp) << 16);
                    lo[3] = term(6, lp) | (term(7, lp) << 16);
                    hi[3] = term(14, lp) | (term(15, lp) << 16);
#else
                    {
                        // the same split with packed conversions: H = (hi part of lx, of lz), L = the parts of the (exact) residuals;
                        // local_term puts them at lo[2] = (lx hi, lx lo), lo[3] = (lx hi, lz hi) = H, hi[3] = (lz hi, lz lo)
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                        const f32x2 l2 = {lx, lz};
                        const f16x2 H = __builtin_convertvector(l2, f16x2);
                        const f16x2 L = __builtin_convertvector(l2 - __builtin_convertvector(H, f32x2), f16x2);
                        const uint32_t hu = __builtin_bit_cast(uint32_t, H), lu = __builtin_bit_cast(uint32_t, L);
                        lo[2] = __builtin_amdgcn_perm(lu, hu, 0x05040100u);
                        lo[3] = hu;
                        hi[3] = __builtin_amdgcn_perm(lu, hu, 0x07060302u);
                    }
#endif
#ifdef RTK_HOST_EMULATION
                    const float4* tile = ltiles + static_cast<size_t>(w) * 64;
                    uint16_t vb[16];
                    for (int d = 0; d < 4; ++d) {
                        vb[2 * d] = static_cast<uint16_t>(lo[d]); vb[2 * d + 1] = static_cast<uint16_t>(lo[d] >> 16);
                        vb[8 + 2 * d] = static_cast<uint16_t>(hi[d]); vb[9 + 2 * d] = static_cast<uint16_t>(hi[d] >> 16);
                    }
                    for (int rho = 0; rho < 32; ++rho) {
                        const int i = tile_row_sphere(rho);
                        float dsum = 0.0f;
                        double exact_sum = 0.0, abs_sum = 0.0;
                        for (int k = 0; k < 16; ++k) {
                            const uint16_t* frag = reinterpret_cast<const uint16_t*>(tile + 32 * (k >> 3) + rho);
                            const float ak = f16_value(frag[k & 7]), bk = f16_value(vb[k]);
                            dsum = fmaf(ak, bk, dsum);
                            exact_sum += static_cast<double>(ak) * bk;
                            abs_sum += std::fabs(static_cast<double>(ak) * bk);
                        }
                        if (rtk_shim::ulp_noise > 0 && abs_sum < 1e30)
                            dsum = static_cast<float>(exact_sum + rtk_shim::ulp_noise / 3.0 * 16 * 0x1p-23 * abs_sum) + 0.0f;
                        const float x = fmaf(dsum, -0x1p48f, kc);
                        if (x > 0.0f && x < 1.0f) RTK_STAT_EACH(clamp_fractions);
                        if (x > 0.0f) m |= 0x80000000u >> i;
                    }
#else
                    u32x2 sw = __builtin_amdgcn_permlane32_swap(lo[2], hi[2], false, false);
                    g0[2] = sw[0]; g1[2] = sw[1];
                    sw = __builtin_amdgcn_permlane32_swap(lo[3], hi[3], false, false);
                    g0[3] = sw[0]; g1[3] = sw[1];
                    sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(kc), __float_as_uint(kc), false, false);
                    rtk_f16x8 a0, b0, b1;
                    __builtin_memcpy(&a0, &av, 16);
                    __builtin_memcpy(&b0, &g0, 16);
                    __builtin_memcpy(&b1, &g1, 16);
                    rtk_f32x16 t0, t1;
#ifdef RTK_PROFILE_WORD
                    const unsigned long long q1 = __builtin_readcyclecounter();
#endif
                    mfma_pair(t0, t1, a0, b0, b1);
#ifdef RTK_PROFILE_WORD
                    const unsigned long long q2 = __builtin_readcyclecounter();
#endif
                    m = mfma_join(mfma_bits16(t0, __uint_as_float(sw[0])), mfma_bits16(t1, __uint_as_float(sw[1])));
#ifdef RTK_PROFILE_WORD
                    const unsigned long long q3 = __builtin_readcyclecounter();
                    wp[1] += q1 - q0; wp[2] += q2 - q1; wp[3] += q3 - q2; wp[5] += 1;
#endif
#endif
                    // Origins so far from the word that the reference's rounding noise (2^-19 |v|^2) rivals the squares the dot product
                    // is made of (a bounce off the far side of the ground sphere, seen from a short word's fine frame): the clamped
                    // sums prove nothing there.  thr = |p|^2 - K is everything a centre's squared distance from the line may be
                    // for the reference to accept it; no centre is farther than sqrt(2) B from the frame's: the whole word is out
                    // when |p| > sqrt(2) B + sqrt(thr) (sqrt(x) <= (x + 1) / 2), else every sphere of it is a candidate.
                    if (any_far) {          // wave-uniform
                        const float pps = pp * d2.z, vvs = vv * d2.z;                                      // in scaled units
                        const float kraw = fmaf(fmaf(vv, kLocalRatio, pp), d2.x, d1.x) * -0x1p-75f;          // K, before the push away from zero
                        const float rf = fmaf(pps - kraw, 0.5f, 0.5f + 1.41422f * kLocalB);
                        const bool out = pps * (1.0f - 0x1p-10f) > rf * rf;
                        m = (vvs < kLocalSafeVV) ? m : (out ? 0u : 0xFFFFFFFFu);          // the second arm also for NaN
                    }
                    m = ((m | always | force_class) & valid) & keep;
                    RTK_STAT(lane, class_words);
                } else {
                    // batches of eight spheres, each requested whole (two s_load_dwordx16) and waited for once; entries past
                    // the word's end are read (the array is padded) and masked by `valid`
                    for (int b8 = 0; b8 < count; b8 += 8) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const auto sp = src[base + b8 + i];
                            filter_one(f, sp.x, sp.y, sp.z, sp.w, m);
                        }
                    }
                    m <<= (32 - ((count + 7) & ~7));
                    m = ((m | force_kind0) & valid) & keep;
                }
                RTK_STAT(lane, filter_words);
                const uint32_t cand = m;
#ifdef RTK_PROFILE_SECTIONS
                const unsigned long long pw1 = __builtin_readcyclecounter();
                prof[cls ? 0 : 1] += pw1 - pw0;
#endif
#ifdef RTK_PROFILE_WORD
                const unsigned long long q4 = __builtin_readcyclecounter();
#endif
                push(base, cand);
                if ((w & 7) == 7 && wave_any(ecnt > kEntryCap - 8)) drain();
#ifdef RTK_PROFILE_WORD
                if (cls) wp[4] += __builtin_readcyclecounter() - q4;
#endif
#ifdef RTK_PROFILE_SECTIONS
                prof[2] += __builtin_readcyclecounter() - pw1;
#endif
            }
#ifdef RTK_PROFILE_SECTIONS
            const unsigned long long pf0 = __builtin_readcyclecounter();
#endif
            drain();
#ifdef RTK_PROFILE_SECTIONS
            prof[3] += __builtin_readcyclecounter() - pf0;
#endif
        };
        if (!LDS) {
            // Scenes too large to live in LDS: the wave streams the list through a private LDS chunk — one coalesced
            // global load per 64 spheres instead of one broadcast load per sphere — and filters from LDS as usual.
            // Only this wave touches `stage`; DS operations of a wave execute in order, so the chunk's writes are
            // visible to its reads and the next chunk's writes cannot overtake this chunk's reads.
            // The next chunk is fetched into registers before the current one is scanned, so its
            // latency hides behind ~1150 filter instructions.
            if (ALLOW_LOCAL && lwdesc != nullptr) {
                scan_local();
            } else {
                constexpr int kPerLane = kStageSpheres / kWave;
#ifdef RTK_HOST_EMULATION
                typedef float4 Quad;
#else
                typedef float Quad __attribute__((ext_vector_type(4)));     // a register quad the compiler keeps out of scratch
#endif
                const Quad* src = reinterpret_cast<const Quad*>(filt);
                Quad* dst = reinterpret_cast<Quad*>(stage);
                Quad pre[kPerLane];
#pragma unroll
                for (int j = 0; j < kPerLane; ++j) {
                    const int i = j * kWave + lane;
                    pre[j] = src[i < n ? i : n - 1];
                }
                for (int base = 0; base < n; base += kStageSpheres) {
                    const int count = n - base < kStageSpheres ? n - base : kStageSpheres;
                    RTK_WAVE_SYNC();
#pragma unroll
                    for (int j = 0; j < kPerLane; ++j)
                        if (j * kWave + lane < count) dst[j * kWave + lane] = pre[j];
                    RTK_WAVE_SYNC();
#pragma unroll
                    for (int j = 0; j < kPerLane; ++j) {          // unconditional (clamped) loads: no branch, so no wait before the scan
                        const int i = base + kStageSpheres + j * kWave + lane;
                        pre[j] = src[i < n ? i : n - 1];
                    }
                    scan([&](int i) { return stage[i]; }, base, count);
                }
            }
            flush(r, a, t_min, list_base, cnt, closest, best, n_exact);
        } else {
            scan_words();
        }
        t_out = closest;
        idx_out = best;
    }

    // Closest hit through the uniform grid: same result as closest_hit(), bit for bit.
    // For a finite ray and range the reference's answer does not depend on the order the spheres are visited in:
    // a sphere's root is chosen against t_min alone whenever it could still win, so the list walk of
    // HittableList.hpp:34-38 yields (min root, then max index).  Lanes whose ray or range is not finite, or whose
    // origin lies outside the ball the grid's margins were sized for, take the brute-force path afterwards.
    __device__ __forceinline__ void closest_hit_grid(const GridDev& G, const Ray& r, bool active, float t_min, float t_max, float& t_out,
                                                     int& idx_out, uint32_t& n_exact) const {
        const float a = length_squared(r.d);
        const float big = 0x1p30f;
        const float qx = r.o.x - G.cx, qy = r.o.y - G.cy, qz = r.o.z - G.cz;
        // |d|^2 < 2^60: with every sphere below 2^30 (build_grid) no intermediate of the exact test overflows, so all
        // roots are finite numbers — the condition under which the visiting order does not matter
        const bool safe = (fabsf(r.o.x) < big) && (fabsf(r.o.y) < big) && (fabsf(r.o.z) < big) && (a > 0x1p-80f) && (a < 0x1p60f) &&
                          (fmaf(qx, qx, fmaf(qy, qy, qz * qz)) <= G.r_allow2) && (t_min == t_min) && (t_max == t_max);
        const bool use = active && safe;
        float closest = t_max;
        int best = -1;
        uint32_t mine = 0;          // this lane's exact tests (the walk below is divergent); added to the wave's total at the end
        auto test = [&](int j) {
            const float4 s = exact[j];
            float root;
            if (sphere_root_exact(r, a, s, t_min, closest, root) && (root < closest || j > best)) {
                closest = root;
                best = j;
            }
            ++mine;
        };
        for (int k = 0; k < G.n_always; ++k) {             // the same sphere on every lane: broadcast loads
            const int j = static_cast<int>(G.always[k]);
            if (use) test(j);
        }
        if (use) {
            const float len = __builtin_amdgcn_sqrtf(a);
            const float slack_t = 1.01f * G.slack * __builtin_amdgcn_rcpf(len);
            const float tiny = 0x1p-60f * len;             // a component this small cannot leave a cell within the scene
            const bool mx = fabsf(r.d.x) > tiny, my = fabsf(r.d.y) > tiny, mz = fabsf(r.d.z) > tiny;
            const float inf = __builtin_inff();
            const float ivx = mx ? 1.0f / r.d.x : inf, ivy = my ? 1.0f / r.d.y : inf, ivz = mz ? 1.0f / r.d.z : inf;
            // clip [t_min - slack, closest + slack] against the grid box
            float t0 = t_min - slack_t, t1 = closest + slack_t;
            bool miss = false;
            if (mx) { const float ta = (G.lox - r.o.x) * ivx, tb = (G.hix - r.o.x) * ivx; t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb)); }
            else miss = miss || r.o.x < G.lox || r.o.x > G.hix;
            if (my) { const float ta = (G.loy - r.o.y) * ivy, tb = (G.hiy - r.o.y) * ivy; t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb)); }
            else miss = miss || r.o.y < G.loy || r.o.y > G.hiy;
            if (mz) { const float ta = (G.loz - r.o.z) * ivz, tb = (G.hiz - r.o.z) * ivz; t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb)); }
            else miss = miss || r.o.z < G.loz || r.o.z > G.hiz;
            if (!miss && t0 <= t1 + slack_t) {
                auto cell_of = [](float p, float lo, float ih, int n) {
                    const float f = floorf((p - lo) * ih);
                    return !(f >= 0.0f) ? 0 : (f > static_cast<float>(n - 1) ? n - 1 : static_cast<int>(f));   // NaN and negatives -> 0
                };
                // the registration made for origins like this one
                const uint32_t* cs = G.cell_start + static_cast<size_t>((cell_of(r.o.z, G.loz, G.izz, G.znz) * G.zny + cell_of(r.o.y, G.loy, G.izy, G.zny)) * G.znx +
                                                                        cell_of(r.o.x, G.lox, G.izx, G.znx)) * static_cast<size_t>(G.zone_stride);
                int ix = cell_of(fmaf(t0, r.d.x, r.o.x), G.lox, G.ihx, G.nx);
                int iy = cell_of(fmaf(t0, r.d.y, r.o.y), G.loy, G.ihy, G.ny);
                int iz = cell_of(fmaf(t0, r.d.z, r.o.z), G.loz, G.ihz, G.nz);
                const int sx = r.d.x > 0.0f ? 1 : -1, sy = r.d.y > 0.0f ? 1 : -1, sz = r.d.z > 0.0f ? 1 : -1;
                // every step moves one cell forward along one axis, so the walk ends after at most nx + ny + nz steps
                for (int steps = G.nx + G.ny + G.nz; steps > 0; --steps) {
                    const int cell = (iz * G.ny + iy) * G.nx + ix;
                    const uint32_t b = cs[cell], e = cs[cell + 1];
                    for (uint32_t q = b; q < e; ++q) test(static_cast<int>(G.items[q]));
                    // parameter at which the ray leaves this cell, per axis (recomputed from the cell index: no drift)
                    const float ex = mx ? (fmaf(static_cast<float>(ix + (sx > 0 ? 1 : 0)), G.hx, G.lox) - r.o.x) * ivx : inf;
                    const float ey = my ? (fmaf(static_cast<float>(iy + (sy > 0 ? 1 : 0)), G.hy, G.loy) - r.o.y) * ivy : inf;
                    const float ez = mz ? (fmaf(static_cast<float>(iz + (sz > 0 ? 1 : 0)), G.hz, G.loz) - r.o.z) * ivz : inf;
                    const float t_exit = fminf(ex, fminf(ey, ez));
                    // nothing beyond closest (or beyond the clipped range) can win; both widened by the slack
                    if (!(t_exit - slack_t <= fminf(closest, t1))) break;
                    if (ex <= ey && ex <= ez) { ix += sx; if (ix < 0 || ix >= G.nx) break; }
                    else if (ey <= ez) { iy += sy; if (iy < 0 || iy >= G.ny) break; }
                    else { iz += sz; if (iz < 0 || iz >= G.nz) break; }
                }
            }
        }
        n_exact += wave_sum(mine);
        const bool brute = active && !safe;
        if (wave_any(brute)) {
            float t2;
            int i2;
            closest_hit<false>(r, brute, t_min, t_max, t2, i2, n_exact);
            if (brute) {
                closest = t2;
                best = i2;
            }
        }
        t_out = closest;
        idx_out = best;
    }
};

} // namespace rtk
