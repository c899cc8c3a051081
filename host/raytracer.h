// raytracer.h — RayTracer with the reference's public surface (include/raytracer.h:17-47,
// src/raytracer.cpp:24-174 of antoni-wojcik/OpenCL-Raytracing) over the C ABI of
// librt_amd.so (include/rt_amd.h).  No OpenCL / OpenGL: the image lives in HBM and is
// read back on request.  Errors print the library's message and exit(-1), as the
// reference does (src/kernelgl.cpp:47-56), unless exceptions are enabled with
// RayTracer::throwOnError(true).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/rt_amd.h"
#include "camera.h"
#include "scene.h"

struct Screen;  // the reference's display quad (include/screen.h) — out of scope, kept as an opaque name

struct RayTracerError : std::runtime_error {
    int code;
    RayTracerError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

class RayTracer {
    int width, height;
    rt_context *ctx = nullptr;
    SceneCreator scene;
    std::vector<float> pixels;  // last transferImage(): gamma-space RGBA32F, row 0 = y 0
    static bool throw_on_error;

    void check(int rc) const;
    void upload();

public:
    // the reference's constructor loads this path, relative to the working directory (src/raytracer.cpp:95); the
    // repository ships its own file of that name (assets/scenes/scene.scene)
    static const char *defaultScenePath() { return "assets/scenes/scene.scene"; }
    static void throwOnError(bool on) { throw_on_error = on; }

    // kernel_path is accepted for source compatibility with
    // RayTracer(w, h, "kernels/raytracer.cl") and ignored: the kernels are in librt_amd.so.
    RayTracer(int w, int h, const char *kernel_path);
    RayTracer(int w, int h, const char *kernel_path, const std::string &scene_path, int device = 0,
              uint64_t seed = 0xC0FFEE);
    RayTracer(int w, int h, SceneCreator &&scene_, int device = 0, uint64_t seed = 0xC0FFEE);
    ~RayTracer();
    RayTracer(const RayTracer &) = delete;
    RayTracer &operator=(const RayTracer &) = delete;

    void render(const Camera *camera);       // sample_counter = 0, kernel `trace`
    void renderAgain(const Camera *camera);  // ++sample_counter, kernel `retrace`
    // The reference binds a GL texture here (src/raytracer.cpp:167-174); this one reads the image
    // back and returns it (both arguments may be null).
    const float *transferImage(Screen *screen = nullptr, const char *shader_tex_id = nullptr);
    void setTime(float time);  // declared, never defined in the reference (raytracer.h:45): a stub
    void resize(int w, int h);

    // native additions
    void renderSamples(const Camera *camera, uint32_t first_sample, uint32_t n_samples);  // fused, asynchronous
    const float *renderFrame(const Camera *camera, uint32_t spp);  // clear + fused + resolve + read back
    uint32_t sampleCounter() const;
    rt_context *context() { return ctx; }
    SceneCreator &sceneCreator() { return scene; }
    int getWidth() const { return width; }
    int getHeight() const { return height; }
};
