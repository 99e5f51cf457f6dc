// oracle/ref_harness.cpp — thin C ABI around the UNMODIFIED reference classes, compiled in THIS
// container only, from the sources where they lie under /root/reference (never copied into the
// repo).  Output goes to oracle/_ref/libref_harness.so (git-ignored).  TEST INFRASTRUCTURE ONLY:
// used by tools/make_golden.py to record fixtures under tests/golden/ and by the CPU tests that
// validate oracle/cpu_ref.cpp against the real reference when /root/reference is present.
//
// Build (see oracle/Makefile):
//   g++ -std=c++20 -O2 -ffp-contract=off -fno-access-control -Dmain=rtiow_reference_main \
//       -I$REF ref_harness.cpp $REF/Material.cpp $REF/Vector3.cpp $REF/ProfileLogScope.cpp
// `-fno-access-control` lets the harness read HittableList::objects and Camera's fields
// (HittableList.hpp:26-27, Camera.hpp:31-39 are private with no accessor).  main.cpp is included
// whole because HittableList::hit / Sphere3::hit / write_color_binary are non-inline definitions
// in headers and main.cpp is their only legal includer.

#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <vector>

#include "main.cpp" // the reference's translation unit; its main() is renamed on the command line

extern "C" {

// random_scene() (main.cpp:116-158) from the process's fresh default-seeded RNG statics.
// Call at most once per process.  Returns the number of spheres; arrays must hold >= 1024.
int ref_random_scene(float* cx, float* cy, float* cz, float* radius, float* mr, float* mg, float* mb, float* rough,
                     float* metallic, float* ior, int32_t* type) {
    HittableList world = random_scene();
    int n = 0;
    for (const auto& obj : world.objects) {
        const auto* s = dynamic_cast<const Sphere3*>(obj.get());
        cx[n] = s->center.x(); cy[n] = s->center.y(); cz[n] = s->center.z(); radius[n] = s->radius;
        mr[n] = s->material.color.x(); mg[n] = s->material.color.y(); mb[n] = s->material.color.z();
        rough[n] = s->material.roughness; metallic[n] = s->material.metallic; ior[n] = s->material.refractionIndex;
        type[n] = static_cast<int32_t>(s->material.type);
        ++n;
    }
    return n;
}

// Camera ctor (Camera.hpp:8-23): 22 floats origin, lower_left_corner, horizontal, vertical, u, v, w, lens_radius.
void ref_camera(const float* lf, const float* la, const float* vup, float vfov, float aspect, float aperture,
                float focus, float* out22) {
    Camera c{Point3{lf[0], lf[1], lf[2]}, Point3{la[0], la[1], la[2]}, Vector3{vup[0], vup[1], vup[2]}, vfov, aspect, aperture, focus};
    const Vector3* f[7] = {&c.origin, &c.lower_left_corner, &c.horizontal, &c.vertical, &c.u, &c.v, &c.w};
    for (int i = 0; i < 7; ++i) { out22[3 * i] = f[i]->x(); out22[3 * i + 1] = f[i]->y(); out22[3 * i + 2] = f[i]->z(); }
    out22[21] = c.lens_radius;
}

// HittableList::hit (HittableList.hpp:30-42) over Sphere3 objects built from the arrays.  The
// winning sphere's list index is recovered from the reference's own hit_record by tagging each
// sphere's material.color.x with its index, so the tie rule is the reference's, not a restatement.
void ref_list_hit(int n, const float* cx, const float* cy, const float* cz, const float* radius, const float* rays,
                  uint64_t nrays, float t_min, float t_max, uint8_t* out_hit, int32_t* out_index, float* out_t) {
    HittableList world{};
    for (int i = 0; i < n; ++i) {
        Material m{};
        m.color = Color{static_cast<float>(i), 0.0f, 0.0f};
        world.add(std::make_shared<Sphere3>(Point3{cx[i], cy[i], cz[i]}, radius[i], m));
    }
    for (uint64_t i = 0; i < nrays; ++i) {
        const float* q = rays + 6 * i;
        const Ray3 r{Point3{q[0], q[1], q[2]}, Vector3{q[3], q[4], q[5]}};
        hit_record rec{};
        const bool hit = world.hit(r, t_min, t_max, rec);
        out_hit[i] = hit ? 1 : 0;
        out_index[i] = hit ? static_cast<int32_t>(rec.material.color.x()) : -1;
        out_t[i] = hit ? rec.t : 0.0f;
    }
}

// The same with NESTED lists (HittableList.hpp:20-27: a HittableList is itself a Hittable): every `nest` consecutive
// spheres sit in an inner reference HittableList; sphere k (k % 7 == 3) additionally gets a list of its own two levels
// down.  `tags` must list every sphere in the order the reference visits them (= the order a flatten must produce).
void ref_nested_list_hit(int n, const float* cx, const float* cy, const float* cz, const float* radius, int nest, const float* rays,
                         uint64_t nrays, float t_min, float t_max, uint8_t* out_hit, int32_t* out_index, float* out_t) {
    HittableList world{};
    std::shared_ptr<HittableList> inner;
    for (int i = 0; i < n; ++i) {
        Material m{};
        m.color = Color{static_cast<float>(i), 0.0f, 0.0f};
        std::shared_ptr<Hittable> s = std::make_shared<Sphere3>(Point3{cx[i], cy[i], cz[i]}, radius[i], m);
        if (i % 7 == 3) s = std::make_shared<HittableList>(std::make_shared<HittableList>(s));     // a list in a list around one sphere
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
    for (uint64_t i = 0; i < nrays; ++i) {
        const float* q = rays + 6 * i;
        const Ray3 r{Point3{q[0], q[1], q[2]}, Vector3{q[3], q[4], q[5]}};
        hit_record rec{};
        const bool hit = world.hit(r, t_min, t_max, rec);
        out_hit[i] = hit ? 1 : 0;
        out_index[i] = hit ? static_cast<int32_t>(rec.material.color.x()) : -1;
        out_t[i] = hit ? rec.t : 0.0f;
    }
}

// Sphere3::hit (Sphere3.hpp:31-59), one sphere per ray.  out9 = p, normal, t, front_face, hit.
void ref_sphere_hit(const float* centers, const float* radius, const float* rays, uint64_t n, float t_min, float t_max,
                    float* out9) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* q = rays + 6 * i;
        const Ray3 r{Point3{q[0], q[1], q[2]}, Vector3{q[3], q[4], q[5]}};
        const Sphere3 s{Point3{centers[3 * i], centers[3 * i + 1], centers[3 * i + 2]}, radius[i], Material{}};
        hit_record rec{};
        const bool hit = s.hit(r, t_min, t_max, rec);
        float* o = out9 + 9 * i;
        o[0] = rec.p.x(); o[1] = rec.p.y(); o[2] = rec.p.z();
        o[3] = rec.normal.x(); o[4] = rec.normal.y(); o[5] = rec.normal.z();
        o[6] = rec.t; o[7] = rec.front_face ? 1.0f : 0.0f; o[8] = hit ? 1.0f : 0.0f;
    }
}

// The reference samplers from the process's RNG statics (Vector3.cpp:3-9, 49-59; MathUtils.hpp:15-19).
// kind 0 in-sphere, 1 unit vector, 2 in-disk (3 floats each); kind 3 random_float() (1 float each).
void ref_sample(int kind, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        if (kind == 3) { out[i] = random_float(); continue; }
        const Vector3 v = kind == 0 ? random_in_unit_sphere() : kind == 1 ? random_unit_vector() : random_in_unit_disk();
        out[3 * i] = v.x(); out[3 * i + 1] = v.y(); out[3 * i + 2] = v.z();
    }
}

// Material::scatter (Material.cpp:6-54) after Sphere3::hit on one sphere per ray; draws come from the
// process's RNG statics, so in a FRESH process the i-th call sees the i-th output of ref_sample of
// the matching kind in another fresh process (all rays of one call must use one material type).
// mat7 = color(3), roughness, metallic, ior, type.  out7 = scattered o, d, returned bool.
// Rays that miss their sphere are skipped without consuming draws (out7[6] = -1).
void ref_hit_and_scatter(const float* centers, const float* radius, const float* mat7, const float* rays, uint64_t n,
                         float t_min, float t_max, float* out_t, float* out7) {
    for (uint64_t i = 0; i < n; ++i) {
        const float* q = rays + 6 * i;
        const float* mq = mat7 + 7 * i;
        const Ray3 r{Point3{q[0], q[1], q[2]}, Vector3{q[3], q[4], q[5]}};
        MaterialDesc desc{Color{mq[0], mq[1], mq[2]}, mq[3], mq[4], mq[5]};
        const int type = static_cast<int>(mq[6]);
        const Material m = type == 1 ? make_lambertian(desc) : type == 2 ? make_metal(desc) : type == 3 ? make_dielectric(desc) : make_material(desc);
        const Sphere3 s{Point3{centers[3 * i], centers[3 * i + 1], centers[3 * i + 2]}, radius[i], m};
        hit_record rec{};
        float* o = out7 + 7 * i;
        if (!s.hit(r, t_min, t_max, rec)) { out_t[i] = 0.0f; for (int k = 0; k < 6; ++k) o[k] = 0.0f; o[6] = -1.0f; continue; }
        out_t[i] = rec.t;
        Ray3 out{};
        const bool ok = rec.material.scatter(r, rec, out);
        o[0] = out.o.x(); o[1] = out.o.y(); o[2] = out.o.z(); o[3] = out.d.x(); o[4] = out.d.y(); o[5] = out.d.z();
        o[6] = ok ? 1.0f : 0.0f;
    }
}

// write_color_binary (Color.hpp:28-48) on an array of RGB colour sums.
void ref_write_color_binary(const float* sums, uint64_t npixels, int spp, uint8_t* out) {
    std::ostringstream os(std::ios_base::binary);
    for (uint64_t i = 0; i < npixels; ++i) write_color_binary(os, Color{sums[3 * i], sums[3 * i + 1], sums[3 * i + 2]}, spp);
    const std::string s = os.str();
    std::memcpy(out, s.data(), s.size());
}

// ray_color (main.cpp:83-101) for explicit rays against random_scene(); consumes RNG statics.
void ref_ray_color(const float* rays, uint64_t n, int depth, float* out3) {
    static HittableList world = random_scene();
    for (uint64_t i = 0; i < n; ++i) {
        const float* q = rays + 6 * i;
        const Color c = ray_color(Ray3{Point3{q[0], q[1], q[2]}, Vector3{q[3], q[4], q[5]}}, world, depth);
        out3[3 * i] = c.x(); out3[3 * i + 1] = c.y(); out3[3 * i + 2] = c.z();
    }
}

} // extern "C"
