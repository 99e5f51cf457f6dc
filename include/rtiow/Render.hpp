// rtiow/Render.hpp — the reference's free functions ray_color() and random_scene() (reference:
// main.cpp:83-101 and 116-158) for the host, and render_image(): the GPU replacement of the
// "Image Generation" scope (main.cpp:62-78) behind the same scene objects.
#pragma once

#include "Camera.hpp"
#include "Color.hpp"
#include "Flatten.hpp"
#include "HittableList.hpp"
#include "Material.hpp"
#include "Sphere3.hpp"
#include "rtk.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

// CPU radiance along one ray, at most `depth` segments.  The reference recurses and multiplies the
// material colours from the far end of the path back (c1*(c2*(...*sky))); this walks the path
// forward, remembers the colours, and folds them in that same order.
inline Color ray_color(const Ray3& r, const Hittable& world, int depth) {
    std::vector<Color> albedo;
    Ray3 ray = r;
    Color tail{0.0f, 0.0f, 0.0f};
    for (int left = depth;; --left) {
        if (left <= 0) break;                                   // bounce limit: no light
        hit_record rec{};
        if (!world.hit(ray, 0.001f, infinity, rec)) {
            const Vector3 dir = unit_vector(ray.direction());
            const float t = 0.5f * (dir.y() + 1.0f);
            tail = (1.0f - t) * Color(1.0f, 1.0f, 1.0f) + t * Color(0.5f, 0.7f, 1.0f);
            break;
        }
        Ray3 scattered{};
        if (!rec.material.scatter(ray, rec, scattered)) break;  // absorbed
        albedo.push_back(rec.material.color);
        ray = scattered;
    }
    for (auto it = albedo.rbegin(); it != albedo.rend(); ++it) tail = *it * tail;
    return tail;
}

// Book-1 final scene: ground, a 22 x 22 grid of small random spheres, three big ones.
inline HittableList random_scene() {
    HittableList world{};
    const auto sphere = [&world](Point3 c, float r, const Material& m) { world.add(std::make_shared<Sphere3>(c, r, m)); };
    sphere(Point3{0.0f, -1000.0f, 0.0f}, 1000.0f, make_lambertian(MaterialDesc{Color{0.5f, 0.5f, 0.5f}}));
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const float choose_mat = random_float();
            const float cx = a + 0.9f * random_float();
            const float cz = b + 0.9f * random_float();
            const Point3 center{cx, 0.2f, cz};
            if (!((center - Point3{4.0f, 0.2f, 0.0f}).length() > 0.9f)) continue;
            MaterialDesc desc{};
            Material material{};
            if (choose_mat < 0.8f) {
                const Color c1 = Color::random();
                const Color c2 = Color::random();
                desc.color = c1 * c2;
                material = make_lambertian(desc);
            } else if (choose_mat < 0.95f) {
                desc.color = Color::random(0.5f, 1.0f);
                desc.roughness = random_float(0.0f, 0.5f);
                desc.metallic = 1.0f;
                material = make_metal(desc);
            } else {
                desc.refractionIndex = 1.5f;
                desc.color = Color{1.0f, 1.0f, 1.0f};
                desc.roughness = 0.0f;
                material = make_dielectric(desc);
            }
            sphere(center, 0.2f, material);
        }
    }
    sphere(Point3{0.0f, 1.0f, 0.0f}, 1.0f, make_dielectric(MaterialDesc{Color{1.0f, 1.0f, 1.0f}, 0.0f, 0.0f, 1.5f}));
    sphere(Point3{-4.0f, 1.0f, 0.0f}, 1.0f, make_lambertian(MaterialDesc{Color{0.4f, 0.2f, 0.1f}}));
    sphere(Point3{4.0f, 1.0f, 0.0f}, 1.0f, make_metal(MaterialDesc{Color{0.7f, 0.6f, 0.5f}, 0.0f, 1.0f}));
    return world;
}

namespace rtiow {

// The reference's output file in one write: header of main.cpp:60 + write_color_binary's bytes (Color.hpp:28-48)
// for every pixel of `rgb_sum` ([height][width][3], row 0 = top row).  Byte-identical to streaming the pixels through
// write_color_binary one by one (3 ostream::write calls per pixel in the reference).
inline void write_ppm_binary(std::ostream& out, int image_width, int image_height, const float* rgb_sum, int samples_per_pixel) {
    out << "P6\n" << image_width << ' ' << image_height << '\n' << 255 << '\n';
    const size_t n = static_cast<size_t>(image_width) * image_height * 3;
    std::vector<char> bytes(n);
    for (size_t i = 0; i < n; ++i)
        bytes[i] = static_cast<char>(static_cast<uint8_t>(255 * rtiow_detail::gamma2(rgb_sum[i], samples_per_pixel)));
    out.write(bytes.data(), static_cast<std::streamsize>(n));
}

struct RenderResult {
    std::vector<float> rgb_sum;   // [height][width][3], row 0 = top image row
    rtk_stats stats{};
    int passes_done = 0;          // progressive renders: passes completed so far (== batches when the frame is finished)
};

// GPU render of `world` seen by `camera`: what main.cpp:64-76 computes, as per-pixel colour sums.
// The samples are rendered in `batches` progressive passes (disjoint sample indices, the per-pixel sums are
// added on the host in fp32); `progress(samples_done, samples_total)` is called after every pass — the analogue of
// the reference's per-scanline progress line (main.cpp:65), which has no meaning on a device that renders all
// scanlines at once.  Throws std::runtime_error with rtk_last_error() on failure (there is no CPU fallback).
//
// Resumable: with a `checkpoint_path` the running sums and the number of passes done are written to that file after
// every pass (to a temporary first, then renamed), and a later call with the same path, image size, spp, depth, seed,
// pass count AND the same scene, camera and accelerator (a hash of them is part of the file) CONTINUES after the last completed pass instead of starting over — the reference streams pixels to
// its output and loses everything when interrupted (main.cpp:74).  The finished frame has the bits of an
// uninterrupted one (same passes, same fp32 additions in the same order).  `max_passes_this_call` > 0 stops early
// (what an interruption looks like to the next call); RenderResult::passes_done tells how far the frame is.
struct ProgressiveCheckpoint {
    // file layout: this header, then width * height * 3 floats
    char magic[8];               // "RTKCKPT2"
    int32_t width, height, spp, max_depth, batches, passes_done;
    uint64_t seed;
    uint64_t rays, samples;
    uint64_t content_hash;       // FNV-1a over the flattened scene arrays, the camera's 22 floats, t_min and the accelerator: a
                                 // checkpoint of another scene or camera is NOT continued (the two images would be mixed)
};

// What the pixels depend on besides the header's integers: scene (in list order), camera, t_min, list walk or grid.
inline uint64_t checkpoint_content_hash(const FlatScene& flat, const rtk_camera& cam, float t_min, int accel) {
    uint64_t h = 1469598103934665603ull;
    const auto mix = [&h](const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    const rtk_scene v = flat.view();
    mix(&v.n, sizeof v.n);
    const float* arrays[] = {v.cx, v.cy, v.cz, v.radius, v.mat_r, v.mat_g, v.mat_b, v.roughness, v.metallic, v.ior};
    for (const float* a : arrays) mix(a, sizeof(float) * static_cast<size_t>(v.n));
    mix(v.mat_type, sizeof(int32_t) * static_cast<size_t>(v.n));
    mix(&cam, sizeof(float) * 22);
    mix(&t_min, sizeof t_min);
    mix(&accel, sizeof accel);
    return h;
}

inline bool load_checkpoint(const std::string& path, ProgressiveCheckpoint& h, std::vector<float>& sums) {
    std::ifstream in(path, std::ios_base::binary);
    if (!in) return false;
    in.read(reinterpret_cast<char*>(&h), sizeof h);
    if (!in || std::memcmp(h.magic, "RTKCKPT2", 8) != 0 || h.width <= 0 || h.height <= 0) return false;
    sums.resize(static_cast<size_t>(h.width) * h.height * 3);
    in.read(reinterpret_cast<char*>(sums.data()), static_cast<std::streamsize>(sums.size() * sizeof(float)));
    return static_cast<bool>(in);
}

inline void save_checkpoint(const std::string& path, const ProgressiveCheckpoint& h, const std::vector<float>& sums) {
    const std::string tmp = path + ".tmp";
    {
        std::ofstream out(tmp, std::ios_base::binary | std::ios_base::trunc);
        out.write(reinterpret_cast<const char*>(&h), sizeof h);
        out.write(reinterpret_cast<const char*>(sums.data()), static_cast<std::streamsize>(sums.size() * sizeof(float)));
        if (!out) throw std::runtime_error("cannot write checkpoint " + tmp);
    }
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename checkpoint to " + path);
}

template <class Progress>
inline RenderResult render_image_progressive(const Hittable& world, const Camera& camera, int image_width, int image_height,
                                             int samples_per_pixel, int max_depth, int batches, Progress&& progress, uint64_t seed = 0,
                                             int device = 0, int accel = RTK_ACCEL_LIST, const std::string& checkpoint_path = std::string(),
                                             int max_passes_this_call = 0) {
    const FlatScene flat = flatten(world);
    rtk_context* ctx = nullptr;
    const auto check = [&ctx](int status, const char* what) {
        if (status == RTK_OK) return;
        const std::string msg = std::string(what) + ": " + rtk_last_error();
        if (ctx) rtk_context_destroy(ctx);
        throw std::runtime_error(msg);
    };
    check(rtk_context_create(device, &ctx), "rtk_context_create");
    check(rtk_context_set_accel(ctx, accel), "rtk_context_set_accel");     // RTK_ACCEL_GRID: same image, far fewer sphere tests
    const rtk_scene view = flat.view();
    check(rtk_set_scene(ctx, &view), "rtk_set_scene");
    const rtk_camera cam = camera.export_fields();
    RenderResult out;
    const size_t n = static_cast<size_t>(image_width) * image_height * 3;
    out.rgb_sum.assign(n, 0.0f);
    std::vector<float> pass(n);
    if (batches < 1) batches = 1;
    if (batches > samples_per_pixel) batches = samples_per_pixel > 0 ? samples_per_pixel : 1;
    int done = 0, first_pass = 0;
    const uint64_t content = checkpoint_content_hash(flat, cam, 0.001f, accel);
    if (!checkpoint_path.empty()) {
        ProgressiveCheckpoint h{};
        std::vector<float> sums;
        if (load_checkpoint(checkpoint_path, h, sums) && h.width == image_width && h.height == image_height && h.spp == samples_per_pixel &&
            h.max_depth == max_depth && h.batches == batches && h.seed == seed && h.content_hash == content && h.passes_done > 0 && h.passes_done <= batches) {
            out.rgb_sum = sums;
            out.stats.rays = h.rays;
            out.stats.samples = h.samples;
            first_pass = h.passes_done;
            for (int b = 0; b < first_pass; ++b) done += (samples_per_pixel - done) / (batches - b);
            progress(done, samples_per_pixel);
        }
    }
    out.passes_done = first_pass;
    for (int b = first_pass; b < batches; ++b) {
        if (max_passes_this_call > 0 && b - first_pass >= max_passes_this_call) break;
        const int count = (samples_per_pixel - done) / (batches - b);
        rtk_params p{};
        p.width = image_width;
        p.height = image_height;
        p.spp = count;
        p.max_depth = max_depth;
        p.seed = seed;
        p.t_min = 0.001f;
        p.world_size = 1;
        p.sample_offset = done;
        rtk_stats st{};
        check(rtk_render(ctx, &cam, &p, pass.data(), &st), "rtk_render");
        for (size_t i = 0; i < n; ++i) out.rgb_sum[i] += pass[i];
        out.stats.rays += st.rays;
        out.stats.samples += st.samples;
        out.stats.sphere_tests += st.sphere_tests;
        out.stats.exact_tests += st.exact_tests;
        out.stats.lane_slots += st.lane_slots;
        out.stats.kernel_ms += st.kernel_ms;
        out.stats.total_ms += st.total_ms;
        out.stats.frames += 1;
        done += count;
        out.passes_done = b + 1;
        if (!checkpoint_path.empty()) {
            ProgressiveCheckpoint h{};
            std::memcpy(h.magic, "RTKCKPT2", 8);
            h.content_hash = content;
            h.width = image_width; h.height = image_height; h.spp = samples_per_pixel; h.max_depth = max_depth; h.batches = batches;
            h.passes_done = b + 1; h.seed = seed; h.rays = out.stats.rays; h.samples = out.stats.samples;
            save_checkpoint(checkpoint_path, h, out.rgb_sum);
        }
        progress(done, samples_per_pixel);
    }
    rtk_context_destroy(ctx);
    return out;
}

inline RenderResult render_image(const Hittable& world, const Camera& camera, int image_width, int image_height, int samples_per_pixel,
                                 int max_depth, uint64_t seed = 0, int device = 0, int accel = RTK_ACCEL_LIST) {
    return render_image_progressive(world, camera, image_width, image_height, samples_per_pixel, max_depth, 1, [](int, int) {}, seed, device, accel);
}

// The same frame on SEVERAL GPUs of one node from this one process (rtk_multi_*: interleaved tiles, one context and
// stream per device, framebuffer gathered on devices[0] over RCCL, de-interleaved and quantised there): writes the P6
// file like render_ppm.  The image is bit-identical to the single-device render for any device list.  Returns the
// frame's counters summed over the devices (kernel_ms = the slowest device's kernel).
inline rtk_stats render_ppm_multi(std::ostream& out, const Hittable& world, const Camera& camera, int image_width, int image_height,
                                  int samples_per_pixel, int max_depth, uint64_t seed, const std::vector<int32_t>& devices, int accel = RTK_ACCEL_LIST) {
    const FlatScene flat = flatten(world);
    rtk_multi* m = nullptr;
    const auto check = [&m](int status, const char* what) {
        if (status == RTK_OK) return;
        const std::string msg = std::string(what) + ": " + rtk_last_error();
        if (m) rtk_multi_destroy(m);
        throw std::runtime_error(msg);
    };
    check(rtk_multi_create(devices.data(), static_cast<int32_t>(devices.size()), &m), "rtk_multi_create");
    check(rtk_multi_set_accel(m, accel), "rtk_multi_set_accel");
    const rtk_scene view = flat.view();
    check(rtk_multi_set_scene(m, &view), "rtk_multi_set_scene");
    const rtk_camera cam = camera.export_fields();
    rtk_params p{};
    p.width = image_width;
    p.height = image_height;
    p.spp = samples_per_pixel;
    p.max_depth = max_depth;
    p.seed = seed;
    p.t_min = 0.001f;
    std::vector<uint8_t> bytes(static_cast<size_t>(image_width) * image_height * 3);
    std::vector<rtk_stats> per(devices.size());
    check(rtk_multi_render(m, &cam, &p, nullptr, bytes.data(), per.data()), "rtk_multi_render");
    rtk_multi_destroy(m);
    out << "P6\n" << image_width << ' ' << image_height << '\n' << 255 << '\n';
    out.write(reinterpret_cast<const char*>(bytes.data()), static_cast<std::streamsize>(bytes.size()));
    rtk_stats st{};
    for (const rtk_stats& d : per) {
        st.rays += d.rays; st.samples += d.samples; st.sphere_tests += d.sphere_tests; st.exact_tests += d.exact_tests; st.lane_slots += d.lane_slots;
        st.kernel_ms = d.kernel_ms > st.kernel_ms ? d.kernel_ms : st.kernel_ms;
        st.frames = 1;
    }
    return st;
}

// The reference's whole output step (main.cpp:57-76) in one call: render, apply write_color_binary ON THE DEVICE
// (rtk_render_u8: a quarter of the bytes cross PCIe) and write header + payload of the P6 file.  The bytes equal
// write_ppm_binary(render_image(...)) for the same seed.  Returns the frame's counters.
inline rtk_stats render_ppm(std::ostream& out, const Hittable& world, const Camera& camera, int image_width, int image_height,
                            int samples_per_pixel, int max_depth, uint64_t seed = 0, int device = 0, int accel = RTK_ACCEL_LIST) {
    const FlatScene flat = flatten(world);
    rtk_context* ctx = nullptr;
    const auto check = [&ctx](int status, const char* what) {
        if (status == RTK_OK) return;
        const std::string msg = std::string(what) + ": " + rtk_last_error();
        if (ctx) rtk_context_destroy(ctx);
        throw std::runtime_error(msg);
    };
    check(rtk_context_create(device, &ctx), "rtk_context_create");
    check(rtk_context_set_accel(ctx, accel), "rtk_context_set_accel");
    const rtk_scene view = flat.view();
    check(rtk_set_scene(ctx, &view), "rtk_set_scene");
    const rtk_camera cam = camera.export_fields();
    rtk_params p{};
    p.width = image_width;
    p.height = image_height;
    p.spp = samples_per_pixel;
    p.max_depth = max_depth;
    p.seed = seed;
    p.t_min = 0.001f;
    p.world_size = 1;
    std::vector<uint8_t> bytes(static_cast<size_t>(image_width) * image_height * 3);
    rtk_stats st{};
    check(rtk_render_u8(ctx, &cam, &p, bytes.data(), &st), "rtk_render_u8");
    rtk_context_destroy(ctx);
    out << "P6\n" << image_width << ' ' << image_height << '\n' << 255 << '\n';
    out.write(reinterpret_cast<const char*>(bytes.data()), static_cast<std::streamsize>(bytes.size()));
    return st;
}

} // namespace rtiow
