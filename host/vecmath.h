// vecmath.h — the little float vector/matrix algebra the host façade needs
// (the reference uses glm; this is a self-contained stand-in with glm's
// column-major mat4 conventions and float arithmetic order).
#pragma once
#include <cmath>

namespace rth {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float k) { return {a.x * k, a.y * k, a.z * k}; }
inline vec3 operator*(float k, vec3 a) { return {a.x * k, a.y * k, a.z * k}; }
inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
inline vec3 &operator-=(vec3 &a, vec3 b) { a = a - b; return a; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }  // glm: v * inversesqrt(dot(v,v))
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// column-major 4x4: m.c[col][row]
struct mat4 {
    float c[4][4];
    mat4() : mat4(1.0f) {}
    explicit mat4(float d) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) c[i][j] = i == j ? d : 0.0f;
    }
};
struct col4 { float v[4]; };
inline col4 column(const mat4 &m, int i) { return {{m.c[i][0], m.c[i][1], m.c[i][2], m.c[i][3]}}; }
inline col4 operator*(col4 a, float k) { return {{a.v[0] * k, a.v[1] * k, a.v[2] * k, a.v[3] * k}}; }
inline col4 operator+(col4 a, col4 b) { return {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]}}; }
inline void set_column(mat4 &m, int i, col4 a) { for (int j = 0; j < 4; j++) m.c[i][j] = a.v[j]; }

// glm::translate / rotate / scale: post-multiply (m * T)
inline mat4 translate(const mat4 &m, vec3 v) {
    mat4 r = m;
    set_column(r, 3, column(m, 0) * v.x + column(m, 1) * v.y + column(m, 2) * v.z + column(m, 3));
    return r;
}
inline mat4 scale(const mat4 &m, vec3 v) {
    mat4 r = m;
    set_column(r, 0, column(m, 0) * v.x);
    set_column(r, 1, column(m, 1) * v.y);
    set_column(r, 2, column(m, 2) * v.z);
    return r;
}
inline mat4 rotate(const mat4 &m, float angle, vec3 axis_in) {
    const float c = std::cos(angle), s = std::sin(angle);
    vec3 axis = normalize(axis_in);
    vec3 t = axis * (1.0f - c);
    float rot[3][3];
    rot[0][0] = c + t.x * axis.x;
    rot[0][1] = t.x * axis.y + s * axis.z;
    rot[0][2] = t.x * axis.z - s * axis.y;
    rot[1][0] = t.y * axis.x - s * axis.z;
    rot[1][1] = c + t.y * axis.y;
    rot[1][2] = t.y * axis.z + s * axis.x;
    rot[2][0] = t.z * axis.x + s * axis.y;
    rot[2][1] = t.z * axis.y - s * axis.x;
    rot[2][2] = c + t.z * axis.z;
    mat4 r = m;
    for (int i = 0; i < 3; i++)
        set_column(r, i, column(m, 0) * rot[i][0] + column(m, 1) * rot[i][1] + column(m, 2) * rot[i][2]);
    return r;
}

}  // namespace rth
