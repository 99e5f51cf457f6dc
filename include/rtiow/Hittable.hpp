// rtiow/Hittable.hpp — hit_record, the Hittable interface and the CPU Material::scatter
// (reference: Hittable.hpp:7-28, Material.cpp:6-54).
#pragma once

#include "Material.hpp"
#include "Ray3.hpp"
#include "Vector3.hpp"

#include <cmath>

struct hit_record {
    Point3 p{};
    Vector3 normal{};
    Material material{};
    float t;
    bool hit{false};
    bool front_face{false};

    // The stored normal always opposes the incoming ray.
    void set_face_normal(const Ray3& r, const Vector3& outward_normal) {
        front_face = dot(r.direction(), outward_normal) < 0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

class Hittable {
public:
    virtual bool hit(const Ray3& r, float t_min, float t_max, hit_record& rec) const = 0;
    virtual ~Hittable() noexcept = default;
};

inline bool Material::scatter(const Ray3& ray_in, const hit_record& rec, Ray3& result) {
    if (type == Type::Lambertian) {
        Vector3 dir = rec.normal + roughness * random_unit_vector();
        if (dir.near_zero()) dir = rec.normal;
        result = Ray3{rec.p, dir};
        return true;
    }
    if (type == Type::Metal) {
        const Vector3 mirrored = metallic * reflect(unit_vector(ray_in.direction()), rec.normal);
        result = Ray3{rec.p, mirrored + roughness * random_in_unit_sphere()};
        return dot(result.direction(), rec.normal) > 0;
    }
    if (type == Type::Glass) {
        const float ratio = rec.front_face ? (1.0f / refractionIndex) : refractionIndex;
        const Vector3 ud = unit_vector(ray_in.direction());
        const float cos_theta = std::fmin(dot(-ud, rec.normal), 1.0f);
        const float sin_theta = std::sqrt(1.0f - cos_theta * cos_theta);
        bool mirror = ratio * sin_theta > 1.0f;   // total internal reflection
        if (!mirror) {
            // Schlick; the G0 draw happens only when refraction is possible.
            float r0 = (1.0f - ratio) / (1.0f + ratio);
            r0 = r0 * r0;
            mirror = r0 + (1.0f - r0) * std::pow(1.0f - cos_theta, 5.0f) > random_float();
        }
        result = Ray3{rec.p, mirror ? reflect(ud, rec.normal) : refract(ud, rec.normal, ratio)};
        return true;
    }
    return false;
}
