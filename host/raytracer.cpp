// raytracer.cpp — see raytracer.h.
#include "raytracer.h"

#include <cstdlib>
#include <iostream>

bool RayTracer::throw_on_error = false;

void RayTracer::check(int rc) const {
    if (rc == RT_OK) return;
    std::string msg = rt_last_error(ctx);
    if (throw_on_error) throw RayTracerError(rc, msg);
    std::cerr << msg << std::endl;  // KernelGL::processError: message, then exit(-1)
    std::exit(-1);
}

void RayTracer::upload() {
    rt_scene_desc d = scene.describe();
    check(rt_set_scene(ctx, &d));
    check(rt_set_textures(ctx, scene.texels(), scene.texW(), scene.texH(), scene.texLayers()));
}

RayTracer::RayTracer(int w, int h, const char *kernel_path) : RayTracer(w, h, kernel_path, defaultScenePath()) {}

RayTracer::RayTracer(int w, int h, const char *, const std::string &scene_path, int device, uint64_t seed)
    : width(w), height(h) {
    check(rt_create(device, w, h, &ctx));
    check(rt_set_seed(ctx, seed));
    try {
        scene.loadScene(scene_path);  // the reference hard-codes its path, src/raytracer.cpp:95
        scene.loadTextures();
    } catch (const SceneError &e) {
        if (throw_on_error) throw;
        std::cerr << e.what() << std::endl;  // processError of src/scene.cpp:29-32
        std::exit(-1);
    }
    upload();
}

RayTracer::RayTracer(int w, int h, SceneCreator &&scene_, int device, uint64_t seed)
    : width(w), height(h), scene(std::move(scene_)) {
    check(rt_create(device, w, h, &ctx));
    check(rt_set_seed(ctx, seed));
    upload();
}

RayTracer::~RayTracer() { rt_destroy(ctx); }

void RayTracer::render(const Camera *camera) { check(rt_render(ctx, camera->transferData())); }
void RayTracer::renderAgain(const Camera *camera) { check(rt_render_again(ctx, camera->transferData())); }

const float *RayTracer::transferImage(Screen *, const char *) {
    pixels.resize((size_t)width * height * 4);
    check(rt_read_image(ctx, pixels.data(), pixels.size() * sizeof(float)));
    return pixels.data();
}

void RayTracer::setTime(float) {}

void RayTracer::resize(int w, int h) {
    check(rt_resize(ctx, w, h));
    width = w;
    height = h;
}

void RayTracer::renderSamples(const Camera *camera, uint32_t first_sample, uint32_t n_samples) {
    check(rt_render_spp(ctx, camera->transferData(), first_sample, n_samples));
}

const float *RayTracer::renderFrame(const Camera *camera, uint32_t spp) {
    check(rt_clear(ctx));
    check(rt_render_spp(ctx, camera->transferData(), 0, spp));
    check(rt_resolve(ctx));
    return transferImage();
}

uint32_t RayTracer::sampleCounter() const {
    uint32_t v = 0;
    rt_sample_counter(ctx, &v);
    return v;
}
