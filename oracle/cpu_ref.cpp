// oracle/cpu_ref.cpp — CPU restatement of the reference's per-pixel path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product path (raytracinginoneweekend_amd/, include/)
// may include, link, import or call this file.  Only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py load the library built from it (oracle/liboracle.so).
//
// What it is: a from-scratch, single-file restatement of the reference algorithm in flat
// structs and free functions (no class hierarchy, no shared_ptr, SoA scene), every function
// citing the reference file:line it follows (paths relative to
// /root/reference/RayTracingInOneWeekend/).  It must be compiled WITHOUT fp contraction
// (-ffp-contract=off, no -ffast-math): the reference image changes with FMA contraction and
// only with it (SURVEY.md §7 hard part 1).
//
// How it is pinned: tests/test_oracle_golden.py checks (1) the SHA-256 of the PPM this file
// renders for three argv sets against the hashes of the reference compiled unmodified from
// /root/reference (re-derived here by oracle/Makefile ->
// oracle/_ref/rtiow_ref), which pins RNG consumption order, scene, camera, intersection,
// scatter and quantiser together; (2) fixed ray batches / scatter vectors recorded from the
// compiled reference classes (oracle/ref_harness.cpp -> tests/golden/*.npz).
//
// The only third-party arithmetic is libstdc++'s std::mt19937 + uniform_real_distribution<float>
// (implementation-defined mapping; the scene is libstdc++-specific, SURVEY.md §7 hard part 6) and
// libm's sqrtf/tanf/powf/fminf/fabsf.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

namespace {

struct V3 {
    float x, y, z;
};

// Vector3.cpp:15-33 — componentwise operators, each product/sum rounded to fp32.
inline V3 add(V3 u, V3 v) { return {u.x + v.x, u.y + v.y, u.z + v.z}; }
inline V3 sub(V3 u, V3 v) { return {u.x - v.x, u.y - v.y, u.z - v.z}; }
inline V3 mulv(V3 u, V3 v) { return {u.x * v.x, u.y * v.y, u.z * v.z}; }
inline V3 muls(float t, V3 v) { return {t * v.x, t * v.y, t * v.z}; }
inline V3 neg(V3 v) { return {-v.x, -v.y, -v.z}; }
// Vector3.cpp:35-37 — division is reciprocal THEN three multiplies.
inline V3 divs(V3 v, float t) { return muls(1 / t, v); }
// Vector3.cpp:39-41 — (x*x' + y*y') + z*z'.
inline float dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
// Vector3.cpp:43-47.
inline V3 cross(V3 u, V3 v) {
    return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
// Vector3.cpp:122-128.
inline float length_squared(V3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline float length(V3 v) { return std::sqrt(length_squared(v)); }
// Vector3.cpp:61-63.
inline V3 unit_vector(V3 v) { return divs(v, length(v)); }
// Vector3.cpp:65-67 — v - ((2*dot)*n).
inline V3 reflect(V3 v, V3 n) { return sub(v, muls(2.0f * dot(v, n), n)); }
// Vector3.cpp:69-74.
inline V3 refract(V3 uv, V3 n, float eta_over_etaprime) {
    const float cos_theta = std::fmin(dot(neg(uv), n), 1.0f);
    const V3 perpendicular = muls(eta_over_etaprime, add(uv, muls(cos_theta, n)));
    const V3 parallel = muls(-std::sqrt(std::fabs(1.0f - length_squared(perpendicular))), n);
    return add(perpendicular, parallel);
}
// Vector3.cpp:138-141 — epsilon is a double literal; |float| promoted for the compare.
inline bool near_zero(V3 v) {
    const auto epsilon = 1e-8;
    return (std::abs(v.x) < epsilon) && (std::abs(v.y) < epsilon) && (std::abs(v.z) < epsilon);
}

// MathUtils.hpp:15-25 — two INDEPENDENT default-seeded mt19937 (function-local statics in the
// reference): G0 feeds random_float(), G1 feeds random_float(min,max).  A fresh distribution
// object per call.  The state lives for the whole program: scene construction consumes from both
// before rendering starts.
struct Rng {
    std::mt19937 g0;
    std::mt19937 g1;
    float f() {
        std::uniform_real_distribution<float> d(0.0f, 1.0f);
        return d(g0);
    }
    float f(float mn, float mx) {
        std::uniform_real_distribution<float> d(mn, mx);
        return d(g1);
    }
};

// Vector3.cpp:130-136 — braced init: draws are consumed x, y, z in order.
inline V3 random_v3(Rng& r) {
    const float a = r.f();
    const float b = r.f();
    const float c = r.f();
    return {a, b, c};
}
inline V3 random_v3(Rng& r, float mn, float mx) {
    const float a = r.f(mn, mx);
    const float b = r.f(mn, mx);
    const float c = r.f(mn, mx);
    return {a, b, c};
}
// Vector3.cpp:3-9 — reject only when |p|^2 > 1.
inline V3 random_in_unit_sphere(Rng& r) {
    for (;;) {
        const V3 p = random_v3(r, -1.0f, 1.0f);
        if (length_squared(p) > 1.0f) continue;
        return p;
    }
}
// Vector3.cpp:49-51.
inline V3 random_unit_vector(Rng& r) { return unit_vector(random_in_unit_sphere(r)); }
// Vector3.cpp:53-59 — reject when |p|^2 >= 1.
inline V3 random_in_unit_disk(Rng& r) {
    for (;;) {
        const float a = r.f(-1.0f, 1.0f);
        const float b = r.f(-1.0f, 1.0f);
        const V3 p = {a, b, 0.0f};
        if (length_squared(p) >= 1.0f) continue;
        return p;
    }
}

struct Ray {
    V3 o, d;
};
// Ray3.hpp:19-21.
inline V3 ray_at(const Ray& r, float t) { return add(r.o, muls(t, r.d)); }

// Material.hpp:10-15 enum order.
enum MatType : int32_t { MAT_NONE = 0, MAT_LAMBERTIAN = 1, MAT_METAL = 2, MAT_GLASS = 3 };

struct Mat {
    V3 color;
    float roughness, metallic, ior;
    int32_t type;
};

// Material.cpp:65-96 — roughness clamped to <= 1; glass forces metallic = 0.
inline Mat make_lambertian(V3 color, float roughness = 1.0f, float metallic = 0.0f, float ior = 1.0f) {
    return {color, roughness < 1.0f ? roughness : 1.0f, metallic, ior, MAT_LAMBERTIAN};
}
inline Mat make_metal(V3 color, float roughness, float metallic, float ior = 1.0f) {
    return {color, roughness < 1.0f ? roughness : 1.0f, metallic, ior, MAT_METAL};
}
inline Mat make_dielectric(V3 color, float roughness, float ior) {
    return {color, roughness < 1.0f ? roughness : 1.0f, 0.0f, ior, MAT_GLASS};
}

struct Sphere {
    V3 c;
    float r;
    Mat m;
};

struct Hit {
    V3 p, normal;
    float t;
    int32_t index;
    bool front_face;
};

// Sphere3.hpp:31-59 with Hittable.hpp:15-18.  Comparison polarity is kept exactly: every test is
// false for NaN, so a NaN discriminant or root is ACCEPTED (SURVEY.md §3.3 property 3).
inline bool sphere_hit(const Sphere& s, const Ray& r, float t_min, float t_max, Hit& rec) {
    const V3 oc = sub(r.o, s.c);
    const float a = length_squared(r.d);
    const float half_b = dot(oc, r.d);
    const float c = length_squared(oc) - s.r * s.r;
    const float discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0f) return false;
    const float sqrtd = std::sqrt(discriminant);
    float root = (-half_b - sqrtd) / a;
    if (root < t_min || t_max < root) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || t_max < root) return false;
    }
    rec.t = root;
    rec.p = ray_at(r, rec.t);
    const V3 outward_normal = divs(sub(rec.p, s.c), s.r);
    rec.front_face = dot(r.d, outward_normal) < 0;
    rec.normal = rec.front_face ? outward_normal : neg(outward_normal);
    return true;
}

// HittableList.hpp:30-42 — closest = t_max shrinks; a later sphere with root == closest replaces
// the earlier one (range test is the strict `t_max < root`).
inline bool list_hit(const std::vector<Sphere>& world, const Ray& r, float t_min, float t_max, Hit& rec) {
    Hit temp{};
    bool hit_anything = false;
    float closest = t_max;
    for (size_t i = 0; i < world.size(); ++i) {
        if (sphere_hit(world[i], r, t_min, closest, temp)) {
            hit_anything = true;
            closest = temp.t;
            rec = temp;
            rec.index = static_cast<int32_t>(i);
        }
    }
    return hit_anything;
}

// Material.cpp:28-33 — Schlick with std::pow(float, float).
inline float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * std::pow((1.0f - cosine), 5.0f);
}

// Material.cpp:6-54.  `u_glass` supplies the random_float() draw of line 40; it is evaluated
// lazily (only when refraction is possible) to keep the reference's G0 consumption order.
template <class RandUnit, class RandBall, class RandGlass>
inline bool scatter(const Mat& m, const Ray& in, const Hit& rec, Ray& out, RandUnit&& rand_unit, RandBall&& rand_ball,
                    RandGlass&& u_glass) {
    switch (m.type) {
    case MAT_LAMBERTIAN: {
        V3 dir = add(rec.normal, muls(m.roughness, rand_unit()));
        if (near_zero(dir)) dir = rec.normal;
        out = {rec.p, dir};
        return true;
    }
    case MAT_METAL: {
        const V3 dir = muls(m.metallic, reflect(unit_vector(in.d), rec.normal));
        out = {rec.p, add(dir, muls(m.roughness, rand_ball()))};
        return dot(out.d, rec.normal) > 0;
    }
    case MAT_GLASS: {
        const float ratio = rec.front_face ? (1.0f / m.ior) : m.ior;
        const V3 ud = unit_vector(in.d);
        const float cos_theta = std::fmin(dot(neg(ud), rec.normal), 1.0f);
        const float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        const bool cannot_refract = ratio * sin_theta > 1.0f;
        V3 dir;
        if (cannot_refract || reflectance(cos_theta, ratio) > u_glass()) {
            dir = reflect(ud, rec.normal);
        } else {
            dir = refract(ud, rec.normal, ratio);
        }
        out = {rec.p, dir};
        return true;
    }
    default:
        return false;
    }
}

// Camera.hpp:8-39.
struct Cam {
    V3 origin, lower_left_corner, horizontal, vertical, u, v, w;
    float lens_radius;
};

inline Cam make_camera(V3 lookFrom, V3 lookAt, V3 vUp, float vfovDegrees, float aspect, float aperture, float focus) {
    Cam c;
    const float pi = 3.14159265358979323846f;         // std::numbers::pi_v<float>, MathUtils.hpp:9
    const float theta = vfovDegrees * pi / 180.0f;     // MathUtils.hpp:11-13
    const float h = std::tan(theta * 0.5f);
    const float viewport_height = 2.0f * h;
    const float viewport_width = aspect * viewport_height;
    c.w = unit_vector(sub(lookFrom, lookAt));
    c.u = unit_vector(cross(vUp, c.w));
    c.v = cross(c.w, c.u);
    c.origin = lookFrom;
    c.horizontal = muls(focus * viewport_width, c.u);
    c.vertical = muls(focus * viewport_height, c.v);
    // Camera.hpp:21 — origin - horizontal*0.5 - vertical*0.5 - focus*w, left to right.
    c.lower_left_corner =
        sub(sub(sub(c.origin, muls(0.5f, c.horizontal)), muls(0.5f, c.vertical)), muls(focus, c.w));
    c.lens_radius = aperture * 0.5f;
    return c;
}

// Camera.hpp:25-29.
inline Ray get_ray(const Cam& c, float s, float t, V3 disk) {
    const V3 rd = muls(c.lens_radius, disk);
    const V3 offset = add(muls(rd.x, c.u), muls(rd.y, c.v));
    const V3 dir = sub(sub(add(add(c.lower_left_corner, muls(s, c.horizontal)), muls(t, c.vertical)), c.origin), offset);
    return {add(c.origin, offset), dir};
}

struct Ctx {
    Rng rng;
    std::vector<Sphere> world;
    uint64_t segments = 0;   // calls of list_hit == ray segments, SURVEY.md §8d
    uint64_t depth_outs = 0; // samples that ran out of depth (main.cpp:87-89)
};

// main.cpp:83-101 — recursion kept (right-associated colour product).
V3 ray_color(Ctx& ctx, const Ray& r, int depth) {
    if (depth <= 0) {
        ++ctx.depth_outs;
        return {0.0f, 0.0f, 0.0f};
    }
    Hit rec{};
    ++ctx.segments;
    if (list_hit(ctx.world, r, 0.001f, std::numeric_limits<float>::infinity(), rec)) {
        const Mat& m = ctx.world[rec.index].m;
        Ray scattered{};
        Rng& g = ctx.rng;
        const bool ok = scatter(
            m, r, rec, scattered, [&g] { return random_unit_vector(g); }, [&g] { return random_in_unit_sphere(g); },
            [&g] { return g.f(); });
        if (ok) return mulv(m.color, ray_color(ctx, scattered, depth - 1));
        return {0.0f, 0.0f, 0.0f};
    }
    const V3 direction = unit_vector(r.d);
    const float t = 0.5f * (direction.y + 1.0f);
    return add(muls(1.0f - t, V3{1.0f, 1.0f, 1.0f}), muls(t, V3{0.5f, 0.7f, 1.0f}));
}

// main.cpp:116-158.  `k`, `diffuse_below`, `metal_below` generalise the stock (11, 0.8, 0.95) for
// the stress configs C4/C5 of SURVEY.md §8d; stock values reproduce the reference scene exactly.
void random_scene(Ctx& ctx, int k, float diffuse_below, float metal_below) {
    ctx.world.clear();
    Rng& g = ctx.rng;
    ctx.world.push_back({{0.0f, -1000.0f, 0.0f}, 1000.0f, make_lambertian({0.5f, 0.5f, 0.5f})});
    for (int a = -k; a < k; ++a) {
        for (int b = -k; b < k; ++b) {
            const float choose_mat = g.f();
            const float cx = a + 0.9f * g.f();
            const float cz = b + 0.9f * g.f();
            const V3 center = {cx, 0.2f, cz};
            if (length(sub(center, V3{4.0f, 0.2f, 0.0f})) > 0.9f) {
                Mat m;
                if (choose_mat < diffuse_below) {
                    const V3 c1 = random_v3(g);
                    const V3 c2 = random_v3(g);
                    m = make_lambertian(mulv(c1, c2));
                } else if (choose_mat < metal_below) {
                    const V3 col = random_v3(g, 0.5f, 1.0f);
                    const float rough = g.f(0.0f, 0.5f);
                    m = make_metal(col, rough, 1.0f);
                } else {
                    m = make_dielectric({1.0f, 1.0f, 1.0f}, 0.0f, 1.5f);
                }
                ctx.world.push_back({center, 0.2f, m});
            }
        }
    }
    ctx.world.push_back({{0.0f, 1.0f, 0.0f}, 1.0f, make_dielectric({1.0f, 1.0f, 1.0f}, 0.0f, 1.5f)});
    ctx.world.push_back({{-4.0f, 1.0f, 0.0f}, 1.0f, make_lambertian({0.4f, 0.2f, 0.1f})});
    ctx.world.push_back({{4.0f, 1.0f, 0.0f}, 1.0f, make_metal({0.7f, 0.6f, 0.5f}, 0.0f, 1.0f)});
}

// Color.hpp:28-48 — sqrt(scale*c), clamp to [0,1], truncate 255*x to uint8.
inline uint8_t quantise(float c, int spp) {
    const float scale = 1.0f / spp;
    float v = std::sqrt(scale * c);
    v = (v < 0.0f) ? 0.0f : (1.0f < v) ? 1.0f : v;   // std::clamp(v, 0, 1)
    return static_cast<uint8_t>(255 * v);
}

} // namespace

// ------------------------------------------------------------------------------------------------
// C ABI used by the tests (ctypes).  Flat arrays; scene SoA mirrors include/rtk.h's rtk_scene.
// ------------------------------------------------------------------------------------------------
extern "C" {

void* oracle_create() { return new Ctx(); }
void oracle_destroy(void* h) { delete static_cast<Ctx*>(h); }

// Discard `n0` draws of G0 and `n1` of G1: lets the harness run decorrelated CPU renders of the
// SAME scene (a property of the harness, not of the reference; SURVEY.md §7 hard part 4).
void oracle_discard(void* h, uint64_t n0, uint64_t n1) {
    Ctx& c = *static_cast<Ctx*>(h);
    c.rng.g0.discard(n0);
    c.rng.g1.discard(n1);
}

int oracle_random_scene(void* h, int k, float diffuse_below, float metal_below) {
    Ctx& c = *static_cast<Ctx*>(h);
    random_scene(c, k, diffuse_below, metal_below);
    return static_cast<int>(c.world.size());
}

int oracle_scene_size(void* h) { return static_cast<int>(static_cast<Ctx*>(h)->world.size()); }

// Replace the context's scene by flat arrays (for parity runs on arbitrary scenes).
void oracle_set_scene(void* h, int n, const float* cx, const float* cy, const float* cz, const float* radius,
                      const float* mr, const float* mg, const float* mb, const float* rough, const float* metallic,
                      const float* ior, const int32_t* type) {
    Ctx& c = *static_cast<Ctx*>(h);
    c.world.resize(n);
    for (int i = 0; i < n; ++i) {
        c.world[i] = {{cx[i], cy[i], cz[i]}, radius[i], {{mr[i], mg[i], mb[i]}, rough[i], metallic[i], ior[i], type[i]}};
    }
}

void oracle_get_scene(void* h, float* cx, float* cy, float* cz, float* radius, float* mr, float* mg, float* mb,
                      float* rough, float* metallic, float* ior, int32_t* type) {
    Ctx& c = *static_cast<Ctx*>(h);
    for (size_t i = 0; i < c.world.size(); ++i) {
        const Sphere& s = c.world[i];
        cx[i] = s.c.x; cy[i] = s.c.y; cz[i] = s.c.z; radius[i] = s.r;
        mr[i] = s.m.color.x; mg[i] = s.m.color.y; mb[i] = s.m.color.z;
        rough[i] = s.m.roughness; metallic[i] = s.m.metallic; ior[i] = s.m.ior; type[i] = s.m.type;
    }
}

// 22 floats: origin, lower_left_corner, horizontal, vertical, u, v, w (3 each), lens_radius.
void oracle_camera(const float* look_from, const float* look_at, const float* vup, float vfov, float aspect,
                   float aperture, float focus, float* out22) {
    const Cam c = make_camera({look_from[0], look_from[1], look_from[2]}, {look_at[0], look_at[1], look_at[2]},
                              {vup[0], vup[1], vup[2]}, vfov, aspect, aperture, focus);
    std::memcpy(out22, &c, sizeof(float) * 22);
}

// HittableList::hit over the context's scene for a batch of rays (6 floats each: o, d).
void oracle_intersect_batch(void* h, const float* rays, uint64_t n, float t_min, float t_max, uint8_t* out_hit,
                            int32_t* out_index, float* out_t) {
    Ctx& c = *static_cast<Ctx*>(h);
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = {{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]}};
        Hit rec{};
        const bool hit = list_hit(c.world, r, t_min, t_max, rec);
        out_hit[i] = hit ? 1 : 0;
        out_index[i] = hit ? rec.index : -1;
        out_t[i] = hit ? rec.t : 0.0f;
    }
}

// For every (ray, sphere) pair: 1 when Sphere3::hit gets PAST its discriminant test (Sphere3.hpp:37-38), i.e.
// `discriminant < 0` is false (NaN included).  A GPU-side filter may reject a sphere only when this is 0.
void oracle_disc_pass(void* h, const float* rays, uint64_t n, uint8_t* out /* [n][scene size] */) {
    Ctx& c = *static_cast<Ctx*>(h);
    const size_t N = c.world.size();
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = {{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]}};
        for (size_t j = 0; j < N; ++j) {
            const Sphere& s = c.world[j];
            const V3 oc = sub(r.o, s.c);
            const float a = length_squared(r.d);
            const float half_b = dot(oc, r.d);
            const float cc = length_squared(oc) - s.r * s.r;
            const float discriminant = half_b * half_b - a * cc;
            out[i * N + j] = (discriminant < 0.0f) ? 0 : 1;
        }
    }
}

// Sphere3::hit for ONE sphere per ray (sphere index given per ray).  out9 = p(3), normal(3), t,
// front_face, hit.
void oracle_sphere_hit_batch(void* h, const float* rays, const int32_t* sphere, uint64_t n, float t_min, float t_max,
                             float* out9) {
    Ctx& c = *static_cast<Ctx*>(h);
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = {{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]}};
        Hit rec{};
        const bool hit = sphere_hit(c.world[sphere[i]], r, t_min, t_max, rec);
        float* o = out9 + 9 * i;
        o[0] = rec.p.x; o[1] = rec.p.y; o[2] = rec.p.z;
        o[3] = rec.normal.x; o[4] = rec.normal.y; o[5] = rec.normal.z;
        o[6] = rec.t; o[7] = rec.front_face ? 1.0f : 0.0f; o[8] = hit ? 1.0f : 0.0f;
    }
}

// Material::scatter with the random inputs supplied by the caller instead of drawn from G0/G1:
// per ray `rnd7` = unit vector (3) for Lambertian, in-ball vector (3) for Metal, uniform (1) for
// Glass.  Everything downstream of the draw is the reference's arithmetic.  out7 = scattered
// origin (3), direction (3), returned bool.
void oracle_scatter_batch(void* h, const float* rays, const int32_t* sphere, const float* t, const float* rnd7,
                          uint64_t n, float* out7) {
    Ctx& c = *static_cast<Ctx*>(h);
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = {{rays[6 * i + 0], rays[6 * i + 1], rays[6 * i + 2]}, {rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]}};
        const Sphere& s = c.world[sphere[i]];
        Hit rec{};
        rec.t = t[i];
        rec.p = ray_at(r, rec.t);                                  // Sphere3.hpp:53
        const V3 outward = divs(sub(rec.p, s.c), s.r);             // Sphere3.hpp:54
        rec.front_face = dot(r.d, outward) < 0;                    // Hittable.hpp:16
        rec.normal = rec.front_face ? outward : neg(outward);      // Hittable.hpp:17
        const float* q = rnd7 + 7 * i;
        Ray out{};
        const bool ok = scatter(
            s.m, r, rec, out, [q] { return V3{q[0], q[1], q[2]}; }, [q] { return V3{q[3], q[4], q[5]}; },
            [q] { return q[6]; });
        float* o = out7 + 7 * i;
        o[0] = out.o.x; o[1] = out.o.y; o[2] = out.o.z; o[3] = out.d.x; o[4] = out.d.y; o[5] = out.d.z;
        o[6] = ok ? 1.0f : 0.0f;
    }
}

// The samplers themselves (Vector3.cpp:3-9, 49-59), drawn from the context's G1: kind 0 = in unit
// sphere, 1 = unit vector, 2 = in unit disk.  For distribution tests of the GPU's direct maps.
void oracle_sample(void* h, int kind, uint64_t n, float* out3) {
    Ctx& c = *static_cast<Ctx*>(h);
    for (uint64_t i = 0; i < n; ++i) {
        const V3 v = kind == 0 ? random_in_unit_sphere(c.rng) : kind == 1 ? random_unit_vector(c.rng) : random_in_unit_disk(c.rng);
        out3[3 * i + 0] = v.x; out3[3 * i + 1] = v.y; out3[3 * i + 2] = v.z;
    }
}

// main.cpp:69-71 for given pixel positions and GIVEN random numbers: u = (x + xi0) / (W - 1), v = (y + xi1) / (H - 1)
// with y = H - 1 - row, then Camera::get_ray(u, v) (Camera.hpp:25-29) with `disk` standing in for
// random_in_unit_disk().  xy = (x, row) pairs, xi = pixel jitter pairs, disk = lens samples; out6 = origin, direction.
void oracle_primary_batch(const float* cam22, int W, int H, const int32_t* xy, const float* xi, const float* disk, uint64_t n, float* out6) {
    Cam cam;
    std::memcpy(&cam, cam22, sizeof(float) * 22);
    for (uint64_t i = 0; i < n; ++i) {
        const int x = xy[2 * i], y = H - 1 - xy[2 * i + 1];
        const float u = (x + xi[2 * i]) / (W - 1);
        const float v = (y + xi[2 * i + 1]) / (H - 1);
        const Ray r = get_ray(cam, u, v, V3{disk[2 * i], disk[2 * i + 1], 0.0f});
        float* o = out6 + 6 * i;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
    }
}

// main.cpp:62-78: the pixel/sample loops over rows [row_begin, row_end) counted from the TOP of the
// image (row 0 is y = H-1).  Writes the fp32 per-pixel colour sums (row-major from the top, RGB) and,
// if `out_u8` is non-null, the write_color_binary bytes.  The RNG streams continue from wherever the
// context left them, exactly as the reference's statics do after random_scene().
// Returns the number of ray segments traced (calls of HittableList::hit).
uint64_t oracle_render_rows(void* h, const float* cam22, int W, int H, int spp, int max_depth, int row_begin,
                            int row_end, float* out_sum, uint8_t* out_u8) {
    Ctx& c = *static_cast<Ctx*>(h);
    Cam cam;
    std::memcpy(&cam, cam22, sizeof(float) * 22);
    const uint64_t seg0 = c.segments;
    for (int row = row_begin; row < row_end; ++row) {
        const int y = H - 1 - row;
        for (int x = 0; x < W; ++x) {
            V3 pixel = {0.0f, 0.0f, 0.0f};
            for (int s = 0; s < spp; ++s) {
                const float u = (x + c.rng.f()) / (W - 1);
                const float v = (y + c.rng.f()) / (H - 1);
                const Ray r = get_ray(cam, u, v, random_in_unit_disk(c.rng));
                const V3 col = ray_color(c, r, max_depth);
                pixel = {pixel.x + col.x, pixel.y + col.y, pixel.z + col.z};   // Vector3.cpp:87-92
            }
            const size_t k = (static_cast<size_t>(row - row_begin) * W + x) * 3;
            if (out_sum) { out_sum[k] = pixel.x; out_sum[k + 1] = pixel.y; out_sum[k + 2] = pixel.z; }
            if (out_u8) { out_u8[k] = quantise(pixel.x, spp); out_u8[k + 1] = quantise(pixel.y, spp); out_u8[k + 2] = quantise(pixel.z, spp); }
        }
    }
    return c.segments - seg0;
}

uint64_t oracle_depth_outs(void* h) { return static_cast<Ctx*>(h)->depth_outs; }

// Color.hpp:28-48 applied to an array of colour sums.
void oracle_quantise(const float* sums, uint64_t n, int spp, uint8_t* out) {
    for (uint64_t i = 0; i < n; ++i) out[i] = quantise(sums[i], spp);
}

} // extern "C"
