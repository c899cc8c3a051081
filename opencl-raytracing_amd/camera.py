"""Camera — mirror of the reference's class (include/camera.h:20-53,
src/camera.cpp:26-118) without glm.  The only thing the trace kernels see of it
is the 12-float block of ``transferData()``: position, lower_left_corner,
horizontal, vertical (kernels/raytracer.cl:129-134,503).

All arithmetic is float32 in the reference's order; ``tan/sin/cos`` come from the
host libm (as in the reference), so the block is reproducible per host, and the
kernels' parity is pinned on the block, not on libm.
"""
import ctypes
import ctypes.util
import math

import numpy as np

f32 = np.float32

# the reference calls cos/sin/tan on floats (glm::cos(float) → cosf): use the host libm's
# single-precision functions, as its C++ does, rather than rounding double results
_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("cosf", "sinf", "tanf"):
    getattr(_libm, _n).restype = ctypes.c_float
    getattr(_libm, _n).argtypes = [ctypes.c_float]


def cosf(x):
    return f32(_libm.cosf(float(f32(x))))


def sinf(x):
    return f32(_libm.sinf(float(f32(x))))


def tanf(x):
    return f32(_libm.tanf(float(f32(x))))

FORWARD, BACK, LEFT, RIGHT = range(4)  # enum CameraMovementDirection, camera.h:13-18

CAMERA_SPEED_SLOW = f32(0.3)
CAMERA_SPEED_NORMAL = f32(1.0)
CAMERA_SPEED_FAST = f32(5.0)
MOUSE_SENSITIVITY = f32(0.2)
ZOOM_MIN = f32(90.0)
ZOOM_MAX = f32(10.0)
ZOOM_SPEED = f32(0.5)
UP_DIR = np.array([0.0, -1.0, 0.0], dtype=f32)  # world up is -y, camera.cpp:23


def _norm(v):
    # glm::normalize = v * inversesqrt(dot(v, v))
    return v * (f32(1) / f32(np.sqrt(f32(np.dot(v, v)))))


def _cross(a, b):
    return np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], dtype=f32)


class Camera:
    def __init__(self, camera_fov, camera_aspect, pos=(0.0, 0.0, 0.0), y=0.0, p=0.0):
        self.fov = f32(int(camera_fov))  # ctor takes int, camera.h:37
        self.aspect = f32(camera_aspect)
        self.position = np.asarray(pos, dtype=f32).copy()
        self.yaw, self.pitch = f32(y), f32(p)
        self.speed = CAMERA_SPEED_SLOW
        angle = f32(float(self.fov) * math.pi / 180.0)  # double arithmetic, camera.cpp:47
        self.half_height = tanf(angle * f32(0.5))
        self.half_width = self.aspect * self.half_height
        self._updateVectors()

    def _updateVectors(self):  # camera.cpp:26-37
        rp = self.pitch * f32(0.01745329251994329576923690768489)
        ry = self.yaw * f32(0.01745329251994329576923690768489)
        w = np.array([cosf(rp) * sinf(ry), sinf(rp), cosf(rp) * cosf(ry)], dtype=f32)
        self.w = _norm(w)
        self.u = _norm(_cross(self.w, UP_DIR))
        self.v = _cross(self.u, self.w)
        self.lower_left_corner = self.w - (self.half_width * self.u + self.half_height * self.v)
        self.horizontal = f32(2.0) * self.half_width * self.u
        self.vertical = f32(2.0) * self.half_height * self.v

    def _setFov(self):  # camera.cpp:39-44 (uses the 0.0055556f approximation of 1/180)
        angle = f32(float(self.fov) * math.pi * float(f32(0.0055556)))
        self.half_height = tanf(angle * f32(0.5))
        self.half_width = self.aspect * self.half_height
        self._updateVectors()

    def move(self, dir, dt):
        ds = self.speed * f32(dt)
        if dir == FORWARD:
            self.position = self.position + self.w * ds
        elif dir == BACK:
            self.position = self.position - self.w * ds
        elif dir == LEFT:
            self.position = self.position - self.u * ds
        elif dir == RIGHT:
            self.position = self.position + self.u * ds

    def rotate(self, x, y):
        self.yaw = self.yaw + f32(x) * MOUSE_SENSITIVITY * self.fov / ZOOM_MAX
        self.pitch = self.pitch + f32(y) * MOUSE_SENSITIVITY * self.fov / ZOOM_MAX
        self.pitch = min(max(self.pitch, f32(-89.0)), f32(89.0))
        self.yaw = f32(math.fmod(self.yaw, 360.0))
        self._updateVectors()

    def zoom(self, scroll):
        self.fov = self.fov + f32(scroll) * ZOOM_SPEED
        self.fov = min(max(self.fov, ZOOM_MAX), ZOOM_MIN)
        self._setFov()

    def setFasterSpeed(self, speed_up):
        self.speed = CAMERA_SPEED_FAST if speed_up else CAMERA_SPEED_NORMAL

    def setSlowerSpeed(self, speed_down):
        self.speed = CAMERA_SPEED_SLOW if speed_down else CAMERA_SPEED_NORMAL

    def setSize(self, new_aspect):
        self.aspect = f32(new_aspect)
        self._setFov()

    def transferData(self):
        """float32[12]: position, lower_left_corner, horizontal, vertical (camera.cpp:94-110)."""
        return np.concatenate([self.position, self.lower_left_corner, self.horizontal, self.vertical]).astype(f32)
