// rtiow/Ray3.hpp — origin + direction; the direction is NOT normalised (reference: Ray3.hpp:5-27).
#pragma once

#include "Vector3.hpp"

class Ray3 {
public:
    Ray3() = default;
    Ray3(const Point3& origin, const Vector3& direction) : o{origin}, d{direction} {}

    Point3 origin() const { return o; }
    Vector3 direction() const { return d; }
    Point3 at(float t) const { return o + t * d; }

    Point3 o;
    Vector3 d;
};
