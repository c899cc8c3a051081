// rt_cli.cpp — headless driver replacing the reference's GLFW application
// (main.cpp:62-116 render loop, :262-289 screenshot) for the trace path:
//   rt_cli --scene assets/scenes/c2_cornell.scene --size 1920x1080 --spp 64
//          --camera=-8,-1,-8,45,0 [--fov 60] [--seed 12648430] [--progressive]
//          [--out frame.tga] [--pfm frame.pfm] [--raw frame.f32] [--device 0]
// --progressive renders like the interactive app (render + spp-1 × renderAgain, one launch
// per sample); the default is the fused path (all samples in one launch).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "raytracer.h"

static void usage() {
    std::cerr << "usage: rt_cli --scene FILE [--size WxH] [--spp N] [--camera=x,y,z,yaw,pitch] [--fov DEG] "
                 "[--seed N] [--progressive] [--out F.tga] [--pfm F.pfm] [--raw F.f32] [--device N]\n";
    std::exit(2);
}

int main(int argc, char **argv) {
    std::string scene_path = "assets/scenes/c2_cornell.scene", out_tga   // (the façade's own default is the reference's path, assets/scenes/scene.scene)
        , out_pfm, out_raw, dump_scene;
    int w = 1200, h = 800, spp = 16, device = 0, fov = 60;  // the reference's window is 1200x800 (main.cpp:9-12)
    float cam[5] = {0, 0, 0, 0, 0};
    unsigned long long seed = 0xC0FFEE;
    bool progressive = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            size_t n = std::strlen(name);
            if (a.compare(0, n, name) == 0 && a.size() > n && a[n] == '=') return argv[i] + n + 1;
            if (a == name && i + 1 < argc) return argv[++i];
            return nullptr;
        };
        const char *v;
        if ((v = val("--scene"))) scene_path = v;
        else if ((v = val("--size"))) { if (std::sscanf(v, "%dx%d", &w, &h) != 2) usage(); }
        else if ((v = val("--spp"))) spp = std::atoi(v);
        else if ((v = val("--camera"))) { if (std::sscanf(v, "%f,%f,%f,%f,%f", cam, cam + 1, cam + 2, cam + 3, cam + 4) != 5) usage(); }
        else if ((v = val("--fov"))) fov = std::atoi(v);
        else if ((v = val("--seed"))) seed = std::strtoull(v, nullptr, 0);
        else if ((v = val("--out"))) out_tga = v;
        else if ((v = val("--pfm"))) out_pfm = v;
        else if ((v = val("--raw"))) out_raw = v;
        else if ((v = val("--device"))) device = std::atoi(v);
        else if ((v = val("--dump-scene"))) dump_scene = v;
        else if (a == "--progressive") progressive = true;
        else usage();
    }
    if (spp < 1 || w < 1 || h < 1) usage();

    Camera camera(fov, (float)w / (float)h, rth::vec3(cam[0], cam[1], cam[2]), cam[3], cam[4]);
    if (!dump_scene.empty()) {
        // host-only mode (no GPU): the arrays rt_set_scene would receive + the camera block, for parity
        // checks of the parser / OBJ reader / camera against other host implementations
        SceneCreator sc;
        try {
            sc.loadScene(scene_path);
            sc.loadTextures();
        } catch (const SceneError &e) {
            std::cerr << e.what() << std::endl;
            return 1;
        }
        rt_scene_desc d = sc.describe();
        std::ofstream f(dump_scene, std::ios::binary);
        uint32_t counts[12] = {d.material_count, d.sphere_count, d.plane_count, d.lens_count, d.vertex_count, d.uv_count,
                               d.index_count, d.mesh_count, d.model_count, (uint32_t)sc.texW(), (uint32_t)sc.texH(),
                               (uint32_t)sc.texLayers()};
        f.write((const char *)counts, sizeof counts);
        f.write((const char *)camera.transferData(), 12 * sizeof(float));
        f.write((const char *)d.materials, sizeof(rt_material) * d.material_count);
        f.write((const char *)d.spheres, sizeof(rt_sphere) * d.sphere_count);
        f.write((const char *)d.planes, sizeof(rt_plane) * d.plane_count);
        f.write((const char *)d.lenses, sizeof(rt_lens) * d.lens_count);
        f.write((const char *)d.vertices, sizeof(rt_float3) * d.vertex_count);
        f.write((const char *)d.uvs, sizeof(rt_float2) * d.uv_count);
        f.write((const char *)d.indices, sizeof(uint32_t) * d.index_count);
        f.write((const char *)d.meshes, sizeof(rt_mesh) * d.mesh_count);
        f.write((const char *)d.models, sizeof(rt_model) * d.model_count);
        if (sc.texLayers()) f.write((const char *)sc.texels(), sizeof(float) * 4 * sc.texW() * sc.texH() * sc.texLayers());
        return f ? 0 : 1;
    }
    RayTracer tracer(w, h, "kernels/raytracer.cl", scene_path, device, seed);

    auto t0 = std::chrono::steady_clock::now();
    const float *img;
    if (progressive) {
        tracer.render(&camera);
        for (int s = 1; s < spp; s++) tracer.renderAgain(&camera);
        img = tracer.transferImage();
    } else {
        img = tracer.renderFrame(&camera, (uint32_t)spp);
    }
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << w << "x" << h << " " << spp << " spp " << (progressive ? "progressive" : "fused") << ": " << sec * 1e3
              << " ms incl. read-back, " << (double)w * h * spp / sec / 1e6 << " Msamples/s" << std::endl;

    if (!out_raw.empty()) {
        std::ofstream f(out_raw, std::ios::binary);
        f.write((const char *)img, (size_t)w * h * 4 * sizeof(float));
    }
    if (!out_pfm.empty()) {  // PFM stores rows bottom-up; image row 0 is the bottom of the picture already
        std::ofstream f(out_pfm, std::ios::binary);
        f << "PF\n" << w << " " << h << "\n-1.0\n";
        for (size_t i = 0; i < (size_t)w * h; i++) f.write((const char *)(img + 4 * i), 3 * sizeof(float));
    }
    if (!out_tga.empty()) {
        // uncompressed 24-bit TGA, the header of main.cpp:266; bottom-up rows, BGR bytes
        short header[9] = {0, 2, 0, 0, 0, 0, (short)w, (short)h, 24};
        std::ofstream f(out_tga, std::ios::binary);
        f.write((const char *)header, sizeof header);
        std::string row((size_t)w * 3, '\0');
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                const float *p = img + 4 * ((size_t)y * w + x);
                for (int c = 0; c < 3; c++) {
                    float v = p[2 - c];
                    v = v < 0 ? 0 : (v > 1 ? 1 : v);
                    row[3 * x + c] = (char)(unsigned char)std::lround(v * 255.0f);
                }
            }
            f.write(row.data(), row.size());
        }
    }
    return 0;
}
