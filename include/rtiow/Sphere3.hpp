// rtiow/Sphere3.hpp — sphere primitive (reference: Sphere3.hpp:9-59).
#pragma once

#include "Hittable.hpp"
#include "Material.hpp"
#include "Vector3.hpp"

#include <cmath>

class Sphere3 : public Hittable {
public:
    Sphere3() = default;
    Sphere3(const Point3& c, float r, Material m) : center{c}, radius{r}, material{m} {}

    bool hit(const Ray3& r, float t_min, float t_max, hit_record& rec) const override {
        const Vector3 oc = r.origin() - center;
        const float a = r.direction().length_squared();
        const float half_b = dot(oc, r.direction());
        const float c = oc.length_squared() - radius * radius;
        const float discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0f) return false;   // rec.hit untouched on this exit, as in the reference
        const float sqrtd = std::sqrt(discriminant);
        // nearest root inside [t_min, t_max]; comparisons are written so that NaN is accepted
        float root = (-half_b - sqrtd) / a;
        if (root < t_min || t_max < root) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || t_max < root) {
                rec.hit = false;
                return false;
            }
        }
        rec.hit = true;
        rec.t = root;
        rec.p = r.at(root);
        rec.set_face_normal(r, (rec.p - center) / radius);
        rec.material = material;
        return true;
    }

    Point3 center{};
    float radius{1.0f};
    Material material{};
};
