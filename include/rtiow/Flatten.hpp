// rtiow/Flatten.hpp — HittableList -> structure-of-arrays for the C ABI (no counterpart in the
// reference: its list is a vector of shared_ptr<Hittable> walked through virtual calls,
// HittableList.hpp:27,34).  List order is preserved, nested lists are expanded in place (a nested
// HittableList::hit visits its children in order with the caller's running closest-t as t_max, which
// is exactly what in-place expansion does), anything that is neither a Sphere3 nor a HittableList is
// an error because the GPU path only knows spheres.
#pragma once

#include "HittableList.hpp"
#include "Sphere3.hpp"
#include "rtk.h"

#include <cstdint>
#include <stdexcept>
#include <vector>

namespace rtiow {

struct FlatScene {
    std::vector<float> cx, cy, cz, radius, mat_r, mat_g, mat_b, roughness, metallic, ior;
    std::vector<int32_t> mat_type;

    size_t size() const { return cx.size(); }

    void push(const Sphere3& s) {
        cx.push_back(s.center.x());
        cy.push_back(s.center.y());
        cz.push_back(s.center.z());
        radius.push_back(s.radius);
        mat_r.push_back(s.material.color.x());
        mat_g.push_back(s.material.color.y());
        mat_b.push_back(s.material.color.z());
        roughness.push_back(s.material.roughness);
        metallic.push_back(s.material.metallic);
        ior.push_back(s.material.refractionIndex);
        mat_type.push_back(static_cast<int32_t>(s.material.type));
    }

    // View for rtk_set_scene; valid while this object is alive and unmodified.
    rtk_scene view() const {
        rtk_scene v;
        v.n = static_cast<int32_t>(size());
        v.cx = cx.data(); v.cy = cy.data(); v.cz = cz.data(); v.radius = radius.data();
        v.mat_r = mat_r.data(); v.mat_g = mat_g.data(); v.mat_b = mat_b.data();
        v.roughness = roughness.data(); v.metallic = metallic.data(); v.ior = ior.data();
        v.mat_type = mat_type.data();
        return v;
    }
};

inline void flatten_into(const Hittable& h, FlatScene& out) {
    if (const auto* s = dynamic_cast<const Sphere3*>(&h)) {
        out.push(*s);
    } else if (const auto* l = dynamic_cast<const HittableList*>(&h)) {
        for (const auto& child : l->items()) {
            if (!child) throw std::invalid_argument("rtiow::flatten: null object in HittableList");
            flatten_into(*child, out);
        }
    } else {
        throw std::invalid_argument("rtiow::flatten: only Sphere3 and HittableList can be rendered on the GPU");
    }
}

inline FlatScene flatten(const Hittable& world) {
    FlatScene out;
    flatten_into(world, out);
    return out;
}

} // namespace rtiow
