// rtiow/Camera.hpp — thin-lens camera (reference: Camera.hpp:6-40), plus export_fields() so the eight
// private fields can cross the C ABI.
#pragma once

#include "Ray3.hpp"
#include "Vector3.hpp"
#include "rtk.h"

#include <cmath>

class Camera {
public:
    Camera(Point3 lookFrom, Point3 lookAt, Vector3 vUp, float vfovDegrees, float aspectRatio, float aperture, float focusDistance) {
        const float half_height = std::tan(degrees_to_radians(vfovDegrees) * 0.5f);
        const float viewport_height = 2.0f * half_height;
        const float viewport_width = aspectRatio * viewport_height;
        w = unit_vector(lookFrom - lookAt);
        u = unit_vector(cross(vUp, w));
        v = cross(w, u);
        origin = lookFrom;
        horizontal = focusDistance * viewport_width * u;
        vertical = focusDistance * viewport_height * v;
        lower_left_corner = origin - horizontal * 0.5f - vertical * 0.5f - focusDistance * w;
        lens_radius = aperture * 0.5f;
    }

    Ray3 get_ray(float s, float t) const {
        const Vector3 rd = lens_radius * random_in_unit_disk();
        const Vector3 offset = u * rd.x() + v * rd.y();
        return Ray3{origin + offset, lower_left_corner + s * horizontal + t * vertical - origin - offset};
    }

    // Not in the reference: the precomputed fields, for rtk_render.
    rtk_camera export_fields() const {
        rtk_camera c;
        const Vector3* src[7] = {&origin, &lower_left_corner, &horizontal, &vertical, &u, &v, &w};
        float* dst[7] = {c.origin, c.lower_left_corner, c.horizontal, c.vertical, c.u, c.v, c.w};
        for (int i = 0; i < 7; ++i)
            for (int k = 0; k < 3; ++k) dst[i][k] = (*src[i])[k];
        c.lens_radius = lens_radius;
        return c;
    }

private:
    Point3 origin;
    Point3 lower_left_corner;
    Vector3 horizontal;
    Vector3 vertical;
    Vector3 u;
    Vector3 v;
    Vector3 w;
    float lens_radius;
};
