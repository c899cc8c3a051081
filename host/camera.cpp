// camera.cpp — see camera.h.  Behaviour restated from src/camera.cpp of the reference:
// world up is −y; w from yaw/pitch, u = normalize(w × up), v = u × w;
// lower_left_corner = w − (hw·u + hh·v), horizontal = 2·hw·u, vertical = 2·hh·v.
#include "camera.h"

#include <cmath>

using rth::vec3;

namespace {
const float kSpeedSlow = 0.3f, kSpeedNormal = 1.0f, kSpeedFast = 5.0f;
const float kMouseSensitivity = 0.2f;
const float kZoomMin = 90.0f, kZoomMax = 10.0f, kZoomSpeed = 0.5f;
const vec3 kUp(0.0f, -1.0f, 0.0f);
}  // namespace

void Camera::updateVectors() {
    float rp = rth::radians(pitch), ry = rth::radians(yaw);
    w = rth::normalize(vec3(std::cos(rp) * std::sin(ry), std::sin(rp), std::cos(rp) * std::cos(ry)));
    u = rth::normalize(rth::cross(w, kUp));
    v = rth::cross(u, w);
    lower_left_corner = w - (half_width * u + half_height * v);
    horizontal = 2.0f * half_width * u;
    vertical = 2.0f * half_height * v;
}

void Camera::setFov() {
    // the reference approximates 1/180 by 0.0055556f here (src/camera.cpp:40) ...
    float angle = (float)(fov * M_PI * 0.0055556f);
    // The reference calls an UNQUALIFIED tan on a float (src/camera.cpp:41,48): the float overload where the platform's
    // headers put one in the global namespace (libc++, the author's platform), otherwise ::tan(double) rounded to float —
    // the two can differ in the last bit.  Here, and in camera.py: the float function (tanf).  Unpinned, like every libm call.
    half_height = std::tan(angle * 0.5f);
    half_width = aspect * half_height;
    updateVectors();
}

Camera::Camera(int camera_fov, float camera_aspect, const vec3 &pos, float y, float p)
    : fov((float)camera_fov), aspect(camera_aspect), speed(kSpeedSlow), yaw(y), pitch(p), position(pos) {
    // ... and divides by 180 exactly in the constructor (src/camera.cpp:47)
    float angle = (float)(fov * M_PI / 180.0f);
    half_height = std::tan(angle * 0.5f);
    half_width = aspect * half_height;
    updateVectors();
}

void Camera::move(CameraMovementDirection dir, float dt) {
    float ds = speed * dt;
    switch (dir) {
        case FORWARD: position += w * ds; break;
        case BACK: position -= w * ds; break;
        case LEFT: position -= u * ds; break;
        case RIGHT: position += u * ds; break;
    }
}

void Camera::rotate(float x, float y) {
    yaw += x * kMouseSensitivity * fov / kZoomMax;
    pitch += y * kMouseSensitivity * fov / kZoomMax;
    if (pitch > 89.0f) pitch = 89.0f;
    else if (pitch < -89.0f) pitch = -89.0f;
    yaw = std::fmod(yaw, 360.0f);
    updateVectors();
}

void Camera::zoom(float scroll) {
    fov += scroll * kZoomSpeed;
    if (fov < kZoomMax) fov = kZoomMax;
    else if (fov > kZoomMin) fov = kZoomMin;
    setFov();
}

void Camera::setFasterSpeed(bool speed_up) { speed = speed_up ? kSpeedFast : kSpeedNormal; }
void Camera::setSlowerSpeed(bool speed_down) { speed = speed_down ? kSpeedSlow : kSpeedNormal; }

void Camera::setSize(float new_aspect) {
    aspect = new_aspect;
    setFov();
}

float *Camera::transferData() const {
    static float data[12];
    const vec3 *src[4] = {&position, &lower_left_corner, &horizontal, &vertical};
    for (int i = 0; i < 4; i++) {
        data[3 * i] = src[i]->x;
        data[3 * i + 1] = src[i]->y;
        data[3 * i + 2] = src[i]->z;
    }
    return data;
}
