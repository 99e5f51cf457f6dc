// rtiow/Material.hpp — POD material with a type tag (reference: Material.hpp:8-39, Material.cpp).
#pragma once

#include "Vector3.hpp"

class Ray3;
struct hit_record;

class Material {
public:
    enum class Type { None, Lambertian, Metal, Glass };

    // CPU scatter, same contract as the reference's; defined in Hittable.hpp once hit_record is complete.
    bool scatter(const Ray3& ray_in, const hit_record& rec, Ray3& result);

    Vector3 attenuation{};
    Color color{};
    float roughness = 1.0f;
    float metallic = 0.0f;
    float refractionIndex = 1.0f;
    Material::Type type{};
};

struct MaterialDesc {
    Color color{1.0f, 1.0f, 1.0f};
    float roughness{1.0f};
    float metallic{0.0f};
    float refractionIndex{1.0f};
};

namespace rtiow_detail {
inline Material fill(const MaterialDesc& d, Material::Type t, Vector3 att, bool clamp_rough) {
    Material m{};
    m.color = d.color;
    m.attenuation = att;
    m.roughness = clamp_rough ? (d.roughness < 1.0f ? d.roughness : 1.0f) : d.roughness;
    m.metallic = d.metallic;
    m.refractionIndex = d.refractionIndex;
    m.type = t;
    return m;
}
} // namespace rtiow_detail

// Type::None: scatter() absorbs.  The reference leaves refractionIndex at its default here.
inline Material make_material(const MaterialDesc& desc) {
    Material m = rtiow_detail::fill(desc, Material::Type::None, Vector3{1.0f, 0.0f, 0.0f}, false);
    m.refractionIndex = 1.0f;
    return m;
}
// Roughness is clamped to <= 1 by all three typed factories.
inline Material make_lambertian(const MaterialDesc& desc) {
    return rtiow_detail::fill(desc, Material::Type::Lambertian, Vector3{1.0f, 0.0f, 0.0f}, true);
}
inline Material make_metal(const MaterialDesc& desc) {
    return rtiow_detail::fill(desc, Material::Type::Metal, Vector3{1.0f, 0.0f, 0.0f}, true);
}
// Glass forces metallic to 0.
inline Material make_dielectric(const MaterialDesc& desc) {
    Material m = rtiow_detail::fill(desc, Material::Type::Glass, Vector3{1.0f, 1.0f, 1.0f}, true);
    m.metallic = 0.0f;
    return m;
}
