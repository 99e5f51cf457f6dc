// rtiow/HittableList.hpp — brute-force list of hittables (reference: HittableList.hpp:9-42), plus the
// one hook the reference lacks: read access to the objects so the list can be flattened for the GPU.
#pragma once

#include "Hittable.hpp"
#include "Ray3.hpp"

#include <memory>
#include <vector>

class HittableList : public Hittable {
public:
    HittableList() = default;
    HittableList(std::shared_ptr<Hittable> object) { add(object); }

    void clear() { objects.clear(); }
    void add(std::shared_ptr<Hittable> object) { objects.emplace_back(object); }

    // Closest hit; a later object whose root equals the current closest replaces it.
    bool hit(const Ray3& r, float t_min, float t_max, hit_record& rec) const override {
        hit_record candidate{};
        bool any = false;
        float closest = t_max;
        for (const auto& object : objects) {
            if (!object->hit(r, t_min, closest, candidate)) continue;
            any = true;
            closest = candidate.t;
            rec = candidate;
        }
        return any;
    }

    // Not in the reference: const view for rtiow::flatten().
    const std::vector<std::shared_ptr<Hittable>>& items() const { return objects; }

private:
    std::vector<std::shared_ptr<Hittable>> objects;
};
