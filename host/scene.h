// scene.h — SceneCreator with the reference's public surface (include/scene.h:83-153,
// src/scene.cpp:110-461 of antoni-wojcik/OpenCL-Raytracing) over the C ABI of
// librt_amd.so.  The POD structs are the device layouts of include/rt_amd.h
// (cl_float3 == rt_float3, 16 bytes), so the vectors are handed to
// rt_set_scene as they are.  No OpenCL, no Assimp, no stb_image:
//   * loadModel is a minimal OBJ reader reproducing what Assimp yields for
//     aiProcess_Triangulate | aiProcess_FlipUVs (src/scene.cpp:195): one vertex
//     per face corner in file order, fan triangulation, v → 1 − v;
//   * loadTextures decodes 8-bit RGBA PNG (zlib inflate + the five scanline filters) with
//     stb_image's published 8-bit → float rule, colour = (float)pow(byte/255.0f, 2.2f),
//     alpha = byte/255.0f; a map_Kd path that does not exist (the reference's .mtl files
//     name an absolute path on the author's machine) is looked up by base name next to
//     the models / the scene; binary PPM (P6) / PFM with the same stem are an extension.
// Errors throw SceneError (the reference prints and calls exit(-1), scene.cpp:29-32).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/rt_amd.h"
#include "vecmath.h"

typedef rt_float3 cl_float3;
typedef rt_float2 cl_float2;
typedef float cl_float;
typedef uint32_t cl_uint;

enum MatType { t_refractive, t_reflective, t_dielectric, t_diffuse, t_textured, t_light };

typedef rt_material Material;
typedef rt_sphere Sphere;
typedef rt_plane Plane;
typedef rt_lens Lens;
typedef rt_mesh Mesh;
typedef rt_model Model;

struct SceneError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline cl_float3 make_float3(float x, float y, float z) { return cl_float3{x, y, z, 0.0f}; }

class SceneCreator {
    std::vector<Material> materials;
    std::vector<Sphere> spheres;
    std::vector<Plane> planes;
    std::vector<Lens> lenses;
    std::vector<Model> models;
    std::vector<cl_float3> vertices;
    std::vector<cl_float2> texture_uv;
    std::vector<cl_uint> indices;
    std::vector<Mesh> meshes;
    std::vector<std::string> texture_paths;
    std::vector<float> texture_data;  // layers × h × w × RGBA32F
    int tex_w = 0, tex_h = 0, tex_layers = 0;
    cl_uint mesh_count_total = 0;
    std::string base_dir, scene_dir;
    std::vector<std::string> model_dirs;  // directories of the OBJ files loaded so far (texture lookup)
    std::string resolveModelPath(const std::string &p) const;

public:
    void addMaterial(MatType type, const cl_float3 &color, cl_float extra_data);
    void addSphere(const cl_float3 &pos, cl_float r, cl_uint mat_ID);
    void addPlane(const cl_float3 &pos, const cl_float3 &normal, cl_uint mat_ID);
    void addLens(const cl_float3 &pos, const cl_float3 &normal, cl_float r1, cl_float r2, cl_float h, cl_uint mat_ID);
    void loadModel(const std::string &path, cl_uint mat_ID, const rth::mat4 &transform = rth::mat4(1.0f));
    void loadTextures();
    void setTextures(const float *rgba, int w, int h, int layers);
    void loadScene(const std::string &path);
    void loadSceneText(const std::string &text);
    void setBaseDir(const std::string &dir) { base_dir = dir; }  // fallback for "load:" / map_Kd paths that do not exist as written

    // what replaces setupBuffers/createScene/setKernelArgs (src/scene.cpp:46-108)
    rt_scene_desc describe();
    const float *texels() const { return tex_layers ? texture_data.data() : nullptr; }
    int texW() const { return tex_w; }
    int texH() const { return tex_h; }
    int texLayers() const { return tex_layers; }

    size_t sphereCount() const { return spheres.size(); }
    size_t materialCount() const { return materials.size(); }
    size_t modelCount() const { return models.size(); }
    size_t faceCount() const { return indices.size() / 3; }
    const std::vector<std::string> &texturePaths() const { return texture_paths; }
};
