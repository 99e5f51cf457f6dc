// rtiow/Vector3.hpp — 3 x fp32 vector with the reference's public surface (reference:
// Vector3.hpp:8-78, Vector3.cpp).  Header-only; every operation rounds to fp32 step by step (build
// with -ffp-contract=off) so host-side results equal the reference's bit for bit.
#pragma once

#include "MathUtils.hpp"

#include <cmath>
#include <ostream>

class Vector3 {
public:
    Vector3() = default;
    Vector3(float e0, float e1, float e2) : m_e{e0, e1, e2} {}

    float x() const { return m_e[0]; }
    float y() const { return m_e[1]; }
    float z() const { return m_e[2]; }

    Vector3 operator-() const { return {-m_e[0], -m_e[1], -m_e[2]}; }
    float operator[](int i) const { return m_e[i]; }
    float& operator[](int i) { return m_e[i]; }

    Vector3& operator+=(const Vector3& rhs) {
        for (int i = 0; i < 3; ++i) m_e[i] += rhs.m_e[i];
        return *this;
    }
    Vector3& operator*=(const float t) {
        for (int i = 0; i < 3; ++i) m_e[i] *= t;
        return *this;
    }
    Vector3& operator/=(const float t) { return *this *= 1.0f / t; }

    float length_squared() const { return m_e[0] * m_e[0] + m_e[1] * m_e[1] + m_e[2] * m_e[2]; }
    float length() const { return std::sqrt(length_squared()); }

    // Components are drawn x, y, z in that order.
    static Vector3 random() {
        const float a = random_float(), b = random_float(), c = random_float();
        return {a, b, c};
    }
    static Vector3 random(float min, float max) {
        const float a = random_float(min, max), b = random_float(min, max), c = random_float(min, max);
        return {a, b, c};
    }

    // All three components strictly inside (-1e-8, 1e-8); the bound is a double, as in the reference.
    bool near_zero() const {
        constexpr double bound = 1e-8;
        return std::abs(m_e[0]) < bound && std::abs(m_e[1]) < bound && std::abs(m_e[2]) < bound;
    }

private:
    float m_e[3];
};

using Point3 = Vector3;
using Color = Vector3;

inline std::ostream& operator<<(std::ostream& out, const Vector3& v) { return out << v.x() << ' ' << v.y() << ' ' << v.z(); }

inline Vector3 operator+(const Vector3& u, const Vector3& v) { return {u.x() + v.x(), u.y() + v.y(), u.z() + v.z()}; }
inline Vector3 operator-(const Vector3& u, const Vector3& v) { return {u.x() - v.x(), u.y() - v.y(), u.z() - v.z()}; }
inline Vector3 operator*(const Vector3& u, const Vector3& v) { return {u.x() * v.x(), u.y() * v.y(), u.z() * v.z()}; }
inline Vector3 operator*(float t, const Vector3& v) { return {t * v.x(), t * v.y(), t * v.z()}; }
inline Vector3 operator*(const Vector3& v, float t) { return t * v; }
// Division multiplies by the rounded reciprocal (the reference does; it is not three divides).
inline Vector3 operator/(const Vector3& v, float t) { return (1 / t) * v; }

inline float dot(const Vector3& u, const Vector3& v) { return u.x() * v.x() + u.y() * v.y() + u.z() * v.z(); }
inline Vector3 cross(const Vector3& u, const Vector3& v) {
    return {u.y() * v.z() - u.z() * v.y(), u.z() * v.x() - u.x() * v.z(), u.x() * v.y() - u.y() * v.x()};
}
inline Vector3 unit_vector(Vector3 v) { return v / v.length(); }
inline Vector3 reflect(const Vector3& v, const Vector3& n) { return v - 2.0f * dot(v, n) * n; }
inline Vector3 refract(const Vector3& uv, const Vector3& n, float eta_over_etaprime) {
    const float cos_theta = std::fmin(dot(-uv, n), 1.0f);
    const Vector3 across = eta_over_etaprime * (uv + cos_theta * n);
    const Vector3 along = -std::sqrt(std::fabs(1.0f - across.length_squared())) * n;
    return across + along;
}

// Rejection samplers on stream G1.
inline Vector3 random_in_unit_sphere() {
    Vector3 p = Vector3::random(-1.0f, 1.0f);
    while (p.length_squared() > 1.0f) p = Vector3::random(-1.0f, 1.0f);
    return p;
}
inline Vector3 random_unit_vector() { return unit_vector(random_in_unit_sphere()); }
inline Vector3 random_in_unit_disk() {
    for (;;) {
        const float a = random_float(-1.0f, 1.0f), b = random_float(-1.0f, 1.0f);
        const Vector3 p{a, b, 0.0f};
        if (p.length_squared() < 1.0f) return p;
    }
}
