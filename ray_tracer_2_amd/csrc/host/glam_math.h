// glam_math.h -- the subset of glam 0.30.5 (Cargo.lock pin of the reference)
// that the scene pipeline's results depend on, restated in scalar IEEE
// binary32.  glam's sources are not vendored in the reference checkout
// ("parity unpinned"), so these follow glam's published scalar
// algorithms; tests/test_host_math.py checks them against the identities of
// SURVEY.md 8a-8.
//
// Call sites in the reference: Transform::to_matrix / Transform::cam
// (src/scene/components/transform.rs:9-20), BVH::build_per_mesh
// (src/core/bvh.rs:190-193), Camera::to_uniform (src/scene/camera.rs:81-91),
// scene definitions (src/scene/scene.rs:595,610,887).
#ifndef RT_GLAM_MATH_H
#define RT_GLAM_MATH_H

#include <cmath>

namespace rt2 {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
// f32::min / f32::max: the non-NaN operand wins
inline float fmin32(float a, float b) { return std::fmin(a, b); }
inline float fmax32(float a, float b) { return std::fmax(a, b); }
inline Vec3 vmin(Vec3 a, Vec3 b) { return {fmin32(a.x, b.x), fmin32(a.y, b.y), fmin32(a.z, b.z)}; }
inline Vec3 vmax(Vec3 a, Vec3 b) { return {fmax32(a.x, b.x), fmax32(a.y, b.y), fmax32(a.z, b.z)}; }
// glam scalar Vec3::dot: (x*x + y*y) + z*z
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
// glam Vec3::cross
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
// glam Vec3::normalize = self * length().recip()
inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }

struct Quat {
    float x = 0, y = 0, z = 0, w = 1;
};

// glam Quat::mul_quat (scalar)
inline Quat qmul(Quat a, Quat b) {
    Quat r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
    r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}
inline Quat quat_rotation_x(float a) { return {std::sin(a * 0.5f), 0, 0, std::cos(a * 0.5f)}; }
inline Quat quat_rotation_y(float a) { return {0, std::sin(a * 0.5f), 0, std::cos(a * 0.5f)}; }
inline Quat quat_rotation_z(float a) { return {0, 0, std::sin(a * 0.5f), std::cos(a * 0.5f)}; }

enum class EulerRot { XYX, YXZ, XYZ };
// Quat::from_euler: intrinsic rotations, first letter applied first
// (q = q_a * q_b * q_c).  glam 0.30 goes through a rotation matrix; the
// result agrees to rounding (unpinned).
inline Quat quat_from_euler(EulerRot order, float a, float b, float c) {
    switch (order) {
        case EulerRot::XYX: return qmul(qmul(quat_rotation_x(a), quat_rotation_y(b)), quat_rotation_x(c));
        case EulerRot::YXZ: return qmul(qmul(quat_rotation_y(a), quat_rotation_x(b)), quat_rotation_z(c));
        case EulerRot::XYZ: return qmul(qmul(quat_rotation_x(a), quat_rotation_y(b)), quat_rotation_z(c));
    }
    return Quat{};
}

// glam Quat::from_rotation_axes (after DirectXMath XMQuaternionRotationMatrix)
inline Quat quat_from_rotation_axes(Vec3 xa, Vec3 ya, Vec3 za) {
    float m00 = xa.x, m01 = xa.y, m02 = xa.z;
    float m10 = ya.x, m11 = ya.y, m12 = ya.z;
    float m20 = za.x, m21 = za.y, m22 = za.z;
    if (m22 <= 0.0f) {
        float dif10 = m11 - m00;
        float omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) {
            float four_xsq = omm22 - dif10;
            float inv4x = 0.5f / std::sqrt(four_xsq);
            return {four_xsq * inv4x, (m01 + m10) * inv4x, (m02 + m20) * inv4x, (m12 - m21) * inv4x};
        }
        float four_ysq = omm22 + dif10;
        float inv4y = 0.5f / std::sqrt(four_ysq);
        return {(m01 + m10) * inv4y, four_ysq * inv4y, (m12 + m21) * inv4y, (m20 - m02) * inv4y};
    }
    float sum10 = m11 + m00;
    float opm22 = 1.0f + m22;
    if (sum10 <= 0.0f) {
        float four_zsq = opm22 - sum10;
        float inv4z = 0.5f / std::sqrt(four_zsq);
        return {(m02 + m20) * inv4z, (m12 + m21) * inv4z, four_zsq * inv4z, (m01 - m10) * inv4z};
    }
    float four_wsq = opm22 + sum10;
    float inv4w = 0.5f / std::sqrt(four_wsq);
    return {(m12 - m21) * inv4w, (m20 - m02) * inv4w, (m01 - m10) * inv4w, four_wsq * inv4w};
}

// glam Quat::look_to_rh / look_at_lh (0.30): look_at_lh(eye, center, up) =
// look_to_lh((center - eye).normalize(), up) = look_to_rh(-dir, up)
inline Quat quat_look_to_rh(Vec3 dir, Vec3 up) {
    Vec3 f = dir;
    Vec3 s = normalize(cross(f, up));
    Vec3 u = cross(s, f);
    return quat_from_rotation_axes({s.x, u.x, -f.x}, {s.y, u.y, -f.y}, {s.z, u.z, -f.z});
}
inline Quat quat_look_at_lh(Vec3 eye, Vec3 center, Vec3 up) {
    return quat_look_to_rh(-normalize(center - eye), up);
}

// Column-major 4x4: c[col][row], like glam::Mat4::to_cols_array_2d.
struct Mat4 {
    float c[4][4];
};

// glam Mat4::from_scale_rotation_translation via quat_to_axes
inline Mat4 mat4_from_srt(Vec3 scale, Quat q, Vec3 t) {
    float x = q.x, y = q.y, z = q.z, w = q.w;
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2;
    float yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    float xa[4] = {1.0f - (yy + zz), xy + wz, xz - wy, 0.0f};
    float ya[4] = {xy - wz, 1.0f - (xx + zz), yz + wx, 0.0f};
    float za[4] = {xz + wy, yz - wx, 1.0f - (xx + yy), 0.0f};
    Mat4 m;
    for (int i = 0; i < 4; ++i) {
        m.c[0][i] = xa[i] * scale.x;
        m.c[1][i] = ya[i] * scale.y;
        m.c[2][i] = za[i] * scale.z;
    }
    m.c[3][0] = t.x;
    m.c[3][1] = t.y;
    m.c[3][2] = t.z;
    m.c[3][3] = 1.0f;
    return m;
}

// glam scalar Mat4::inverse (cofactor expansion after GLM compute_inverse)
inline Mat4 mat4_inverse(const Mat4& s) {
    float m00 = s.c[0][0], m01 = s.c[0][1], m02 = s.c[0][2], m03 = s.c[0][3];
    float m10 = s.c[1][0], m11 = s.c[1][1], m12 = s.c[1][2], m13 = s.c[1][3];
    float m20 = s.c[2][0], m21 = s.c[2][1], m22 = s.c[2][2], m23 = s.c[2][3];
    float m30 = s.c[3][0], m31 = s.c[3][1], m32 = s.c[3][2], m33 = s.c[3][3];

    float coef00 = m22 * m33 - m32 * m23;
    float coef02 = m12 * m33 - m32 * m13;
    float coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23;
    float coef06 = m11 * m33 - m31 * m13;
    float coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22;
    float coef10 = m11 * m32 - m31 * m12;
    float coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23;
    float coef14 = m10 * m33 - m30 * m13;
    float coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22;
    float coef18 = m10 * m32 - m30 * m12;
    float coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21;
    float coef22 = m10 * m31 - m30 * m11;
    float coef23 = m10 * m21 - m20 * m11;

    float fac0[4] = {coef00, coef00, coef02, coef03};
    float fac1[4] = {coef04, coef04, coef06, coef07};
    float fac2[4] = {coef08, coef08, coef10, coef11};
    float fac3[4] = {coef12, coef12, coef14, coef15};
    float fac4[4] = {coef16, coef16, coef18, coef19};
    float fac5[4] = {coef20, coef20, coef22, coef23};
    float vec0[4] = {m10, m00, m00, m00};
    float vec1[4] = {m11, m01, m01, m01};
    float vec2[4] = {m12, m02, m02, m02};
    float vec3[4] = {m13, m03, m03, m03};
    const float sign_a[4] = {1.0f, -1.0f, 1.0f, -1.0f};
    const float sign_b[4] = {-1.0f, 1.0f, -1.0f, 1.0f};

    Mat4 inv;
    for (int i = 0; i < 4; ++i) {
        float inv0 = (vec1[i] * fac0[i] - vec2[i] * fac1[i]) + vec3[i] * fac2[i];
        float inv1 = (vec0[i] * fac0[i] - vec2[i] * fac3[i]) + vec3[i] * fac4[i];
        float inv2 = (vec0[i] * fac1[i] - vec1[i] * fac3[i]) + vec3[i] * fac5[i];
        float inv3 = (vec0[i] * fac2[i] - vec1[i] * fac4[i]) + vec2[i] * fac5[i];
        inv.c[0][i] = inv0 * sign_a[i];
        inv.c[1][i] = inv1 * sign_b[i];
        inv.c[2][i] = inv2 * sign_a[i];
        inv.c[3][i] = inv3 * sign_b[i];
    }
    float d0 = m00 * inv.c[0][0], d1 = m01 * inv.c[1][0], d2 = m02 * inv.c[2][0], d3 = m03 * inv.c[3][0];
    float det = ((d0 + d1) + d2) + d3;
    float rcp_det = 1.0f / det;
    for (int cidx = 0; cidx < 4; ++cidx)
        for (int r = 0; r < 4; ++r) inv.c[cidx][r] = inv.c[cidx][r] * rcp_det;
    return inv;
}

struct Transform {  // src/scene/components/transform.rs:3-8
    Vec3 pos{0, 0, 0};
    Quat rot{};
    Vec3 scale{1, 1, 1};
    Mat4 to_matrix() const { return mat4_from_srt(scale, rot, pos); }  // transform.rs:10-12
    static Transform cam(Vec3 origin, Vec3 look_at) {                   // transform.rs:13-19
        Transform t;
        t.pos = origin;
        t.rot = quat_look_at_lh(origin, look_at, Vec3{0, 1, 0});
        return t;
    }
};

}  // namespace rt2

#endif
