// rtiow/Camera.hpp — thin-lens camera with the reference's constructor and get_ray() (reference: Camera.hpp:6-40).
// The precomputed frame is kept directly in the C ABI's rtk_camera record, so handing it to the GPU is a copy.
#pragma once

#include "Ray3.hpp"
#include "Vector3.hpp"
#include "rtk.h"

#include <cmath>

class Camera {
public:
    Camera(Point3 lookFrom, Point3 lookAt, Vector3 vUp, float vfovDegrees, float aspectRatio, float aperture, float focusDistance) {
        // viewport size from the vertical field of view
        const float half_height = std::tan(degrees_to_radians(vfovDegrees) * 0.5f);
        const float view_h = 2.0f * half_height;
        const float view_w = aspectRatio * view_h;
        // right-handed frame: w points from the target to the eye
        const Vector3 back = unit_vector(lookFrom - lookAt);
        const Vector3 right = unit_vector(cross(vUp, back));
        const Vector3 up = cross(back, right);
        const Vector3 across = focusDistance * view_w * right;
        const Vector3 upward = focusDistance * view_h * up;
        const Point3 corner = lookFrom - across * 0.5f - upward * 0.5f - focusDistance * back;
        store(frame_.origin, lookFrom);
        store(frame_.lower_left_corner, corner);
        store(frame_.horizontal, across);
        store(frame_.vertical, upward);
        store(frame_.u, right);
        store(frame_.v, up);
        store(frame_.w, back);
        frame_.lens_radius = aperture * 0.5f;
    }

    // Ray through viewport coordinates (s, t) from a random point of the lens disk (stream G1).
    Ray3 get_ray(float s, float t) const {
        const Vector3 lens = frame_.lens_radius * random_in_unit_disk();
        const Vector3 shift = load(frame_.u) * lens.x() + load(frame_.v) * lens.y();
        const Point3 eye = load(frame_.origin);
        return Ray3{eye + shift, load(frame_.lower_left_corner) + s * load(frame_.horizontal) + t * load(frame_.vertical) - eye - shift};
    }

    // Not in the reference: the precomputed fields, for rtk_render.
    rtk_camera export_fields() const { return frame_; }

private:
    static void store(float (&dst)[3], const Vector3& v) {
        dst[0] = v.x();
        dst[1] = v.y();
        dst[2] = v.z();
    }
    static Vector3 load(const float (&src)[3]) { return Vector3{src[0], src[1], src[2]}; }

    rtk_camera frame_{};
};
