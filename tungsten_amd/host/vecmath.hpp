// Host-side float vector/matrix helpers.  Operation order follows the reference
// (src/core/math/Vec.hpp:127-193, Mat4f.hpp, Mat4f.cpp) so that baked scene
// constants (camera basis, quad frames, world-space vertices) come out with the
// same bits as the reference computes in prepareForRender.
#pragma once
#include <cmath>

namespace tungsten_amd {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float a) : x(a), y(a), z(a) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
    Vec3 operator+(const Vec3 &o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3 &o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(const Vec3 &o) const { return {x*o.x, y*o.y, z*o.z}; }
    Vec3 operator/(const Vec3 &o) const { return {x/o.x, y/o.y, z/o.z}; }
    Vec3 operator*(float s) const { return {x*s, y*s, z*s}; }
    Vec3 operator/(float s) const { return {x/s, y/s, z/s}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
    Vec3 &operator+=(const Vec3 &o) { x += o.x; y += o.y; z += o.z; return *this; }
    Vec3 &operator-=(const Vec3 &o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    Vec3 &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    Vec3 &operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
    float dot(const Vec3 &o) const { float s = x*o.x; s += y*o.y; s += z*o.z; return s; }
    float lengthSq() const { float r = 0.0f; r += x*x; r += y*y; r += z*z; return r; }
    float length() const { return std::sqrt(lengthSq()); }
    Vec3 cross(const Vec3 &o) const { return {y*o.z - z*o.y, z*o.x - x*o.z, x*o.y - y*o.x}; }
    Vec3 normalized() const { float inv = 1.0f/length(); return {x*inv, y*inv, z*inv}; }
    void normalize() { float inv = 1.0f/length(); x *= inv; y *= inv; z *= inv; }
    float max() const { float m = x; if (y > m) m = y; if (z > m) m = z; return m; }
};
inline Vec3 operator*(float s, const Vec3 &v) { return {v.x*s, v.y*s, v.z*s}; }

// Row-major 4x4, element names as in the reference (a11..a44).
struct Mat4 {
    float a[16];
    Mat4() { for (int i = 0; i < 16; ++i) a[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
    Mat4(float a11, float a12, float a13, float a14, float a21, float a22, float a23, float a24,
         float a31, float a32, float a33, float a34, float a41, float a42, float a43, float a44)
    {
        float v[16] = {a11, a12, a13, a14, a21, a22, a23, a24, a31, a32, a33, a34, a41, a42, a43, a44};
        for (int i = 0; i < 16; ++i) a[i] = v[i];
    }
    Mat4(const Vec3 &right, const Vec3 &up, const Vec3 &fwd)
    : Mat4(right.x, up.x, fwd.x, 0.0f, right.y, up.y, fwd.y, 0.0f, right.z, up.z, fwd.z, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f)
    {
    }
    float operator()(int i, int j) const { return a[i*4 + j]; }
    Vec3 right() const { return {a[0], a[4], a[8]}; }
    Vec3 up() const { return {a[1], a[5], a[9]}; }
    Vec3 fwd() const { return {a[2], a[6], a[10]}; }
    void setRight(const Vec3 &v) { a[0] = v.x; a[4] = v.y; a[8] = v.z; }
    Vec3 translation() const { return {a[3], a[7], a[11]}; }
    // Mat4f::operator*(Vec3f) : point transform (Mat4f.hpp:320-327)
    Vec3 operator*(const Vec3 &b) const
    {
        return {a[0]*b.x + a[1]*b.y + a[2]*b.z + a[3],
                a[4]*b.x + a[5]*b.y + a[6]*b.z + a[7],
                a[8]*b.x + a[9]*b.y + a[10]*b.z + a[11]};
    }
    Vec3 transformVector(const Vec3 &b) const
    {
        return {a[0]*b.x + a[1]*b.y + a[2]*b.z,
                a[4]*b.x + a[5]*b.y + a[6]*b.z,
                a[8]*b.x + a[9]*b.y + a[10]*b.z};
    }
    Mat4 operator*(const Mat4 &b) const
    {
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int t = 0; t < 4; ++t)
                r.a[i*4 + t] = a[i*4 + 0]*b.a[0*4 + t] + a[i*4 + 1]*b.a[1*4 + t] + a[i*4 + 2]*b.a[2*4 + t] + a[i*4 + 3]*b.a[3*4 + t];
        return r;
    }
    Mat4 transpose() const
    {
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) r.a[i*4 + j] = a[j*4 + i];
        return r;
    }
    static Mat4 scale(const Vec3 &s)
    {
        return Mat4(s.x, 0, 0, 0, 0, s.y, 0, 0, 0, 0, s.z, 0, 0, 0, 0, 1);
    }
    // Mat4f.cpp:48-55
    Mat4 extractRotation() const { return Mat4(right().normalized(), up().normalized(), fwd().normalized()); }
    Mat4 extractScale() const { return scale(Vec3(right().length(), up().length(), fwd().length())); }
    // Mat4f.cpp:10-13
    Mat4 toNormalMatrix() const
    {
        return scale(Vec3(1.0f)/Vec3(right().lengthSq(), up().lengthSq(), fwd().lengthSq()))*(*this);
    }
    // Mat4f.cpp:118-130 ; PI = 3.1415926536f (Angle.hpp:8)
    static Mat4 rotYXZ(const Vec3 &rot)
    {
        const float PI = 3.1415926536f;
        Vec3 r = rot*PI/180.0f;
        float c[] = {std::cos(r.x), std::cos(r.y), std::cos(r.z)};
        float s[] = {std::sin(r.x), std::sin(r.y), std::sin(r.z)};
        return Mat4(c[1]*c[2] - s[1]*s[0]*s[2], -c[1]*s[2] - s[1]*s[0]*c[2], -s[1]*c[0], 0.0f,
                    c[0]*s[2], c[0]*c[2], -s[0], 0.0f,
                    s[1]*c[2] + c[1]*s[0]*s[2], -s[1]*s[2] + c[1]*s[0]*c[2], c[1]*c[0], 0.0f,
                    0.0f, 0.0f, 0.0f, 1.0f);
    }
};

} // namespace tungsten_amd
