"""tools/prove_integration.py — INTEGRATION.md §A carried out: the maintainer's patch applied to a TEMPORARY copy of the
reference's own sources, built against librtk.so, the binary kept as oracle/_ref/rtiow_ref_patched.

    python tools/prove_integration.py            (in the build container: needs /root/reference; hipcc is not needed)

What it does, exactly as INTEGRATION.md §A describes:
  * HittableList.hpp: one accessor `items()` after `add()` (the class keeps `objects` private, HittableList.hpp:26-27);
  * Camera.hpp: `#include "rtk.h"` and `export_fields()` after `get_ray()` (the eight fields are private, Camera.hpp:31-39);
  * main.cpp: `#include "rtk.h"`, and the pixel / sample loops inside the "Image Generation" scope (main.cpp:62-78)
    replaced by flatten -> rtk_set_scene -> rtk_render -> write_color_binary over the returned sums.
Nothing else of the reference is touched: its random_scene(), its Camera constructor, its argv parsing, its PPM header and
its write_color_binary run as they are.  No reference source is copied into this repository: the copy lives in a
temporary directory that is deleted, only the built executable is kept (git-ignored, like oracle/_ref/rtiow_ref; it
travels to the GPU box with the tree).  tests/test_gpu_y_additions.py::test_patched_reference_executable_matches_rtk_render_u8
runs it on the GPU and compares its image_binary.ppm byte for byte with rtk_render_u8 of the same arguments.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RTIOW_REFERENCE", "/root/reference/RayTracingInOneWeekend")
OUT = os.path.join(ROOT, "oracle", "_ref", "rtiow_ref_patched")
LIBDIR = os.path.join(ROOT, "raytracinginoneweekend_amd")

ITEMS = '''
    // --- added for librtk (INTEGRATION.md A): read access to the list, in list order ---
    const std::vector<std::shared_ptr<Hittable>>& items() const { return objects; }
'''
EXPORT = '''
    // --- added for librtk (INTEGRATION.md A): the eight precomputed fields as the C ABI takes them ---
    rtk_camera export_fields() const {
        rtk_camera c{};
        auto put = [](float* d, const Vector3& a) { d[0] = a.x(); d[1] = a.y(); d[2] = a.z(); };
        put(c.origin, origin); put(c.lower_left_corner, lower_left_corner); put(c.horizontal, horizontal);
        put(c.vertical, vertical); put(c.u, u); put(c.v, v); put(c.w, w); c.lens_radius = lens_radius;
        return c;
    }
'''
LOOP = '''        // --- main.cpp:64-76 replaced (INTEGRATION.md A): the frame is rendered by librtk on the GPU ---
        std::vector<float> cx, cy, cz, rad, mr, mg, mb, rough, metal, ior; std::vector<int32_t> type;
        for (const auto& obj : world.items()) {          // list order preserved: the tie rule of HittableList.hpp:34-38 depends on it
            const auto* s = dynamic_cast<const Sphere3*>(obj.get());
            if (!s) { std::cerr << "not a Sphere3\\n"; return 1; }
            cx.push_back(s->center.x()); cy.push_back(s->center.y()); cz.push_back(s->center.z()); rad.push_back(s->radius);
            mr.push_back(s->material.color.x()); mg.push_back(s->material.color.y()); mb.push_back(s->material.color.z());
            rough.push_back(s->material.roughness); metal.push_back(s->material.metallic); ior.push_back(s->material.refractionIndex);
            type.push_back(static_cast<int32_t>(s->material.type));
        }
        rtk_scene scene{(int32_t)cx.size(), cx.data(), cy.data(), cz.data(), rad.data(), mr.data(), mg.data(), mb.data(),
                        rough.data(), metal.data(), ior.data(), type.data()};
        rtk_context* ctx = nullptr;
        if (rtk_context_create(0, &ctx) || rtk_set_scene(ctx, &scene)) { std::cerr << rtk_last_error() << '\\n'; return 1; }
        rtk_params p{}; p.width = image_width; p.height = image_height; p.spp = samples_per_pixel; p.max_depth = max_depth;
        p.seed = 0; p.t_min = 0.001f; p.world_size = 1;
        const rtk_camera cam = camera.export_fields();
        std::vector<float> sums(size_t(image_width) * image_height * 3);      // row 0 = y = H-1, as main.cpp:64 emits
        rtk_stats st{};
        if (rtk_render(ctx, &cam, &p, sums.data(), &st)) { std::cerr << rtk_last_error() << '\\n'; return 1; }
        for (size_t i = 0; i < sums.size(); i += 3)
            write_color_binary(bin_file, Color{sums[i], sums[i + 1], sums[i + 2]}, samples_per_pixel);   // Color.hpp:28, unchanged
        rtk_context_destroy(ctx);
'''


def patch(path, edits):
    text = open(path).read()
    for old, new in edits:
        if text.count(old) != 1:
            raise SystemExit(f"{os.path.basename(path)}: anchor not found exactly once:\n{old}")
        text = text.replace(old, new)
    open(path, "w").write(text)


def main():
    if not os.path.isdir(REF):
        raise SystemExit(f"{REF} is not here (the reference only exists in the build container); keeping any existing {OUT}")
    with tempfile.TemporaryDirectory(prefix="rtiow_patched_") as td:
        for f in os.listdir(REF):
            if f.endswith((".cpp", ".hpp")):
                shutil.copy(os.path.join(REF, f), td)
        patch(os.path.join(td, "HittableList.hpp"), [
            ("    virtual bool hit(const Ray3& r, float t_min, float t_max, hit_record& rec) const override;\nprotected:",
             ITEMS + "\n    virtual bool hit(const Ray3& r, float t_min, float t_max, hit_record& rec) const override;\nprotected:")])
        patch(os.path.join(td, "Camera.hpp"), [
            ('#include "Vector3.hpp"\n', '#include "Vector3.hpp"\n#include "rtk.h"\n'),
            ("protected:\nprivate:\n    Point3 origin;", EXPORT + "protected:\nprivate:\n    Point3 origin;")])
        main_cpp = os.path.join(td, "main.cpp")
        text = open(main_cpp).read()
        a = text.index("        for(int y = image_height - 1; y >= 0; --y) {")
        b = text.index('        std::cerr << "\\nDone.\\n";')
        text = text[:a] + LOOP + text[b:]
        text = text.replace('#include "ProfileLogScope.hpp"\n', '#include "ProfileLogScope.hpp"\n#include "rtk.h"\n#include <vector>\n#include <cstdint>\n', 1)
        open(main_cpp, "w").write(text)
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        srcs = [os.path.join(td, f) for f in ("main.cpp", "Material.cpp", "Vector3.cpp", "ProfileLogScope.cpp", "Ray3.cpp")]
        cmd = ["g++", "-std=c++20", "-O2", "-ffp-contract=off", "-w", "-I", os.path.join(ROOT, "include"), "-I", td, *srcs, "-o", OUT,
               "-L", LIBDIR, "-lrtk", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,$ORIGIN/../../raytracinginoneweekend_amd", "-Wl,-rpath,/opt/rocm/lib"]
        print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        # what changed, for the record (line counts only: the sources themselves stay where they are)
        for f in ("main.cpp", "HittableList.hpp", "Camera.hpp"):
            d = subprocess.run(["diff", os.path.join(REF, f), os.path.join(td, f)], capture_output=True, text=True).stdout
            added = sum(1 for l in d.splitlines() if l.startswith(">"))
            removed = sum(1 for l in d.splitlines() if l.startswith("<"))
            print(f"  {f}: {removed} reference lines replaced, {added} lines added")
    print("built", OUT)


if __name__ == "__main__":
    main()
