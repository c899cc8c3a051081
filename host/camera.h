// camera.h — Camera with the reference's public surface (include/camera.h:20-53,
// src/camera.cpp:26-118 of antoni-wojcik/OpenCL-Raytracing), without glm.
// What the trace kernels see of it is transferData(): 12 floats = position,
// lower_left_corner, horizontal, vertical (kernels/raytracer.cl:129-134,503).
#pragma once
#include "vecmath.h"

enum CameraMovementDirection { FORWARD, BACK, LEFT, RIGHT };

class Camera {
    float fov, aspect;
    float speed;
    float half_height, half_width;
    float yaw, pitch;
    rth::vec3 u, v, w;
    rth::vec3 position;
    rth::vec3 horizontal, vertical, lower_left_corner;

    void updateVectors();
    void setFov();

public:
    Camera(int camera_fov, float camera_aspect, const rth::vec3 &pos = rth::vec3(0.0f, 0.0f, 0.0f), float y = 0.0f,
           float p = 0.0f);

    void move(CameraMovementDirection dir, float dt);
    void rotate(float x, float y);
    void zoom(float scroll);

    void setFasterSpeed(bool speed_up);
    void setSlowerSpeed(bool speed_down);
    void setSize(float new_aspect);

    // pointer to a function-static float[12]: valid until the next call, not thread-safe
    // (the reference's contract, src/camera.cpp:94-110)
    float *transferData() const;
};
