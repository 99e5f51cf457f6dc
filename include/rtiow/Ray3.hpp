// rtiow/Ray3.hpp — a ray is an origin and an UNNORMALISED direction; points on it are o + t d
// (reference: Ray3.hpp:5-27; same public members and accessors).
#pragma once

#include "Vector3.hpp"

class Ray3 {
public:
    Point3 o;    // public in the reference too
    Vector3 d;

    Ray3() = default;
    Ray3(const Point3& origin, const Vector3& direction) : o{origin}, d{direction} {}

    Point3 origin() const { return o; }
    Vector3 direction() const { return d; }
    // t is in units of |d|: the reference never normalises directions, and t_min = 0.001 is in these units.
    Point3 at(float t) const { return o + t * d; }
};
