// host_capi.cpp -> librtiow_host.so: a C view of the C++ host scene API (include/rtiow/*.hpp) so that
// Python (tests, bench.py) can build the same scenes and cameras the drop-in main builds, and so the
// host classes can be checked against the oracle without a GPU.  Links nothing from oracle/.
#include "Camera.hpp"
#include "Color.hpp"
#include "Flatten.hpp"
#include "HittableList.hpp"
#include "Render.hpp"
#include "Sphere3.hpp"

#include <cstring>
#include <random>
#include <sstream>

namespace {
int copy_out(const rtiow::FlatScene& f, int cap, float* cx, float* cy, float* cz, float* radius, float* mr, float* mg, float* mb, float* rough,
             float* metallic, float* ior, int32_t* type) {
    const int n = static_cast<int>(f.size());
    if (n > cap) return -n;
    const size_t b = sizeof(float) * n;
    std::memcpy(cx, f.cx.data(), b); std::memcpy(cy, f.cy.data(), b); std::memcpy(cz, f.cz.data(), b);
    std::memcpy(radius, f.radius.data(), b); std::memcpy(mr, f.mat_r.data(), b); std::memcpy(mg, f.mat_g.data(), b);
    std::memcpy(mb, f.mat_b.data(), b); std::memcpy(rough, f.roughness.data(), b); std::memcpy(metallic, f.metallic.data(), b);
    std::memcpy(ior, f.ior.data(), b); std::memcpy(type, f.mat_type.data(), sizeof(int32_t) * n);
    return n;
}
} // namespace

extern "C" {

// random_scene() (Render.hpp; reference main.cpp:116-158) flattened.  Uses the process-wide RNG
// statics, so it yields the stock scene only on its FIRST call in a process.  Returns n, or -n if
// `cap` is too small.
int rtiow_host_random_scene(int cap, float* cx, float* cy, float* cz, float* radius, float* mr, float* mg, float* mb, float* rough, float* metallic,
                            float* ior, int32_t* type) {
    const HittableList world = random_scene();
    return copy_out(rtiow::flatten(world), cap, cx, cy, cz, radius, mr, mg, mb, rough, metallic, ior, type);
}

// Stress scenes of BASELINE.json configs 4 and 5: the same generator with the grid widened to
// [-k, k)^2 and the material thresholds moved; private mt19937 streams seeded with `seed` (so the call
// is repeatable).  k = 11, 0.8, 0.95 and the default seed 5489 reproduce the stock scene.
int rtiow_host_scaled_scene(int k, float diffuse_below, float metal_below, uint32_t seed, int cap, float* cx, float* cy, float* cz, float* radius,
                            float* mr, float* mg, float* mb, float* rough, float* metallic, float* ior, int32_t* type) {
    std::mt19937 g0(seed), g1(seed);
    const auto r0 = [&g0] { return std::uniform_real_distribution<float>(0.0f, 1.0f)(g0); };
    const auto r1 = [&g1](float a, float b) { return std::uniform_real_distribution<float>(a, b)(g1); };
    HittableList world{};
    world.add(std::make_shared<Sphere3>(Point3{0.0f, -1000.0f, 0.0f}, 1000.0f, make_lambertian(MaterialDesc{Color{0.5f, 0.5f, 0.5f}})));
    for (int a = -k; a < k; ++a) {
        for (int b = -k; b < k; ++b) {
            const float choose = r0();
            const float x = a + 0.9f * r0();
            const float z = b + 0.9f * r0();
            const Point3 center{x, 0.2f, z};
            if (!((center - Point3{4.0f, 0.2f, 0.0f}).length() > 0.9f)) continue;
            MaterialDesc d{};
            Material m{};
            if (choose < diffuse_below) {
                const float a0 = r0(), a1 = r0(), a2 = r0(), b0 = r0(), b1 = r0(), b2 = r0();
                d.color = Color{a0, a1, a2} * Color{b0, b1, b2};
                m = make_lambertian(d);
            } else if (choose < metal_below) {
                const float c0 = r1(0.5f, 1.0f), c1 = r1(0.5f, 1.0f), c2 = r1(0.5f, 1.0f);
                d.color = Color{c0, c1, c2};
                d.roughness = r1(0.0f, 0.5f);
                d.metallic = 1.0f;
                m = make_metal(d);
            } else {
                d.refractionIndex = 1.5f;
                d.color = Color{1.0f, 1.0f, 1.0f};
                d.roughness = 0.0f;
                m = make_dielectric(d);
            }
            world.add(std::make_shared<Sphere3>(center, 0.2f, m));
        }
    }
    world.add(std::make_shared<Sphere3>(Point3{0.0f, 1.0f, 0.0f}, 1.0f, make_dielectric(MaterialDesc{Color{1.0f, 1.0f, 1.0f}, 0.0f, 0.0f, 1.5f})));
    world.add(std::make_shared<Sphere3>(Point3{-4.0f, 1.0f, 0.0f}, 1.0f, make_lambertian(MaterialDesc{Color{0.4f, 0.2f, 0.1f}})));
    world.add(std::make_shared<Sphere3>(Point3{4.0f, 1.0f, 0.0f}, 1.0f, make_metal(MaterialDesc{Color{0.7f, 0.6f, 0.5f}, 0.0f, 1.0f})));
    return copy_out(rtiow::flatten(world), cap, cx, cy, cz, radius, mr, mg, mb, rough, metallic, ior, type);
}

// Camera constructor -> the 22 floats of rtk_camera.
void rtiow_host_camera(const float* lf, const float* la, const float* vup, float vfov, float aspect, float aperture, float focus, float* out22) {
    const Camera c{Point3{lf[0], lf[1], lf[2]}, Point3{la[0], la[1], la[2]}, Vector3{vup[0], vup[1], vup[2]}, vfov, aspect, aperture, focus};
    const rtk_camera f = c.export_fields();
    std::memcpy(out22, &f, sizeof(float) * 22);
}

// HittableList::hit of the host classes on a scene given as arrays; when `nest` > 0 every `nest`
// consecutive spheres are wrapped in an inner HittableList (exercises flatten's in-place expansion:
// `flat_index` receives, per ray, the index the winning sphere has in rtiow::flatten(world)).
void rtiow_host_list_hit(int n, const float* cx, const float* cy, const float* cz, const float* radius, int nest, const float* rays, uint64_t nrays,
                         float t_min, float t_max, uint8_t* out_hit, int32_t* flat_index, float* out_t) {
    HittableList world{};
    std::shared_ptr<HittableList> inner;
    for (int i = 0; i < n; ++i) {
        Material m{};
        m.color = Color{static_cast<float>(i), 0.0f, 0.0f};
        auto s = std::make_shared<Sphere3>(Point3{cx[i], cy[i], cz[i]}, radius[i], m);
        if (nest > 0) {
            if (i % nest == 0) {
                inner = std::make_shared<HittableList>();
                world.add(inner);
            }
            inner->add(s);
        } else {
            world.add(s);
        }
    }
    const rtiow::FlatScene flat = rtiow::flatten(world);
    for (uint64_t i = 0; i < nrays; ++i) {
        const float* q = rays + 6 * i;
        hit_record rec{};
        const bool hit = world.hit(Ray3{Point3{q[0], q[1], q[2]}, Vector3{q[3], q[4], q[5]}}, t_min, t_max, rec);
        out_hit[i] = hit ? 1 : 0;
        out_t[i] = hit ? rec.t : 0.0f;
        int idx = -1;
        if (hit) {
            const float tag = rec.material.color.x();
            for (size_t k = 0; k < flat.size(); ++k)
                if (flat.mat_r[k] == tag) idx = static_cast<int>(k);
        }
        flat_index[i] = idx;
    }
}

// The nested world of oracle/ref_harness.cpp ref_nested_list_hit, built with THIS repository's host classes and
// flattened (rtiow::flatten expands nested lists in place): the SoA arrays rtk_set_scene takes.  Material: Lambertian
// whose red channel is the sphere's position in the build order (so a test can see the order flatten produced).
int rtiow_host_flatten_nested(int n, const float* cx, const float* cy, const float* cz, const float* radius, int nest, int cap, float* ocx, float* ocy,
                              float* ocz, float* oradius, float* mr, float* mg, float* mb, float* rough, float* metallic, float* ior, int32_t* type) {
    HittableList world{};
    std::shared_ptr<HittableList> inner;
    for (int i = 0; i < n; ++i) {
        Material m = make_lambertian(MaterialDesc{Color{static_cast<float>(i), 0.5f, 0.5f}});
        std::shared_ptr<Hittable> s = std::make_shared<Sphere3>(Point3{cx[i], cy[i], cz[i]}, radius[i], m);
        if (i % 7 == 3) s = std::make_shared<HittableList>(std::make_shared<HittableList>(s));
        if (nest > 0) {
            if (i % nest == 0) {
                inner = std::make_shared<HittableList>();
                world.add(inner);
            }
            inner->add(s);
        } else {
            world.add(s);
        }
    }
    return copy_out(rtiow::flatten(world), cap, ocx, ocy, ocz, oradius, mr, mg, mb, rough, metallic, ior, type);
}

// write_color_binary over an array of RGB sums.
void rtiow_host_write_color_binary(const float* sums, uint64_t npixels, int spp, uint8_t* out) {
    std::ostringstream os(std::ios_base::binary);
    for (uint64_t i = 0; i < npixels; ++i) write_color_binary(os, Color{sums[3 * i], sums[3 * i + 1], sums[3 * i + 2]}, spp);
    const std::string s = os.str();
    std::memcpy(out, s.data(), s.size());
}

// write_ppm_binary (Render.hpp): whole file image into `out` (capacity `cap`), returns its size.
int64_t rtiow_host_write_ppm(int W, int H, const float* sums, int spp, uint8_t* out, int64_t cap) {
    std::ostringstream os(std::ios_base::binary);
    rtiow::write_ppm_binary(os, W, H, sums, spp);
    const std::string s = os.str();
    if (static_cast<int64_t>(s.size()) > cap) return -static_cast<int64_t>(s.size());
    std::memcpy(out, s.data(), s.size());
    return static_cast<int64_t>(s.size());
}

// The reference program on the host classes, CPU only (main.cpp:64-76 with ray_color): first call in
// a process only (RNG statics).  For pinning the host API against the golden PPM hashes.
void rtiow_host_cpu_render(int W, int H, int spp, int depth, float aspect, uint8_t* out_u8) {
    const HittableList world = random_scene();
    const Camera camera{Point3{13.0f, 2.0f, 3.0f}, Point3{0.0f, 0.0f, 0.0f}, Vector3{0.0f, 1.0f, 0.0f}, 20, aspect, 0.1f, 10.0f};
    std::ostringstream os(std::ios_base::binary);
    for (int y = H - 1; y >= 0; --y) {
        for (int x = 0; x < W; ++x) {
            Color pixel{0.0f, 0.0f, 0.0f};
            for (int s = 0; s < spp; ++s) {
                const float u = (x + random_float()) / (W - 1);
                const float v = (y + random_float()) / (H - 1);
                pixel += ray_color(camera.get_ray(u, v), world, depth);
            }
            write_color_binary(os, pixel, spp);
        }
    }
    const std::string s = os.str();
    std::memcpy(out_u8, s.data(), s.size());
}

} // extern "C"
