// scene.cpp — see scene.h.  The .scene grammar is restated from src/scene.cpp:297-461 of
// the reference: '#' comments; section headers MATERIALS/SPHERES/PLANES/LENSES/MODELS;
// fields split at commas outside parentheses and consumed strictly left to right;
// vectors "(x, y, z)" and floats in plain decimal notation (no exponent); material ids
// are a SINGLE digit; model operations translate/rotate/scale post-multiply the
// current transform and `load` resets it.
#include "scene.h"

#include <cctype>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

#include <zlib.h>

using rth::mat4;
using rth::vec3;

namespace {

[[noreturn]] void fail(const std::string &msg) { throw SceneError(msg); }

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) a++;
    while (b > a && std::isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// fields of one line: commas inside "(...)" do not split
std::vector<std::string> split_fields(const std::string &line) {
    std::vector<std::string> out;
    std::string cur;
    int depth = 0;
    for (char ch : line) {
        if (ch == '(') depth++;
        if (ch == ')' && depth > 0) depth--;
        if (ch == ',' && depth == 0) {
            out.push_back(cur);
            cur.clear();
        } else cur.push_back(ch);
    }
    out.push_back(cur);
    return out;
}

// [-+]?digits*(.digits+)? with at least one character — the reference's float pattern
bool parse_decimal(const std::string &tok, float &value) {
    std::string t = trim(tok);
    size_t i = 0, n = t.size();
    if (n == 0) return false;
    if (t[i] == '+' || t[i] == '-') i++;
    size_t int_digits = 0, frac_digits = 0;
    while (i < n && std::isdigit((unsigned char)t[i])) { i++; int_digits++; }
    if (i < n && t[i] == '.') {
        i++;
        while (i < n && std::isdigit((unsigned char)t[i])) { i++; frac_digits++; }
        if (frac_digits == 0) return false;
    }
    if (i != n || (int_digits == 0 && frac_digits == 0)) return false;
    try {
        value = std::stof(t);
    } catch (...) { return false; }
    return true;
}

struct Fields {
    std::vector<std::string> f;
    size_t i = 0;
    explicit Fields(const std::string &line) : f(split_fields(line)) {}
    const std::string &next() {
        if (i >= f.size()) fail("ERROR: SCENE: NOT ENOUGH PARAMETERS");
        return f[i++];
    }
    float getFloat() {
        const std::string &w = next();
        float v;
        if (!parse_decimal(w, v)) fail("ERROR: SCENE: IMPROPER FLOAT: " + w);
        return v;
    }
    cl_float3 getVec() {
        const std::string &w = next();
        std::string t = trim(w);
        if (t.size() < 2 || t.front() != '(' || t.back() != ')') fail("ERROR: SCENE: IMPROPER VECTOR: " + w);
        std::vector<std::string> parts = split_fields(t.substr(1, t.size() - 2));
        float v[3];
        if (parts.size() != 3) fail("ERROR: SCENE: IMPROPER VECTOR: " + w);
        for (int k = 0; k < 3; k++)
            if (!parse_decimal(parts[k], v[k])) fail("ERROR: SCENE: IMPROPER VECTOR: " + w);
        return make_float3(v[0], v[1], v[2]);
    }
    cl_uint getUInt() {  // one digit only (src/scene.cpp:455)
        const std::string &w = next();
        std::string t = trim(w);
        if (t.size() != 1 || !std::isdigit((unsigned char)t[0])) fail("ERROR: SCENE: IMPROPER UNSIGNED INT: " + w);
        return (cl_uint)(t[0] - '0');
    }
    std::string getPath() {
        const std::string &w = next();
        std::string t = trim(w);
        if (t.size() < 2 || t.front() != '"' || t.back() != '"') fail("ERROR: SCENE: IMPROPER PATH: " + w);
        return t.substr(1, t.size() - 2);
    }
};

std::string join_path(const std::string &dir, const std::string &p) {
    if (dir.empty() || (!p.empty() && p[0] == '/')) return p;
    return dir + "/" + p;
}
std::string dir_of(const std::string &p) {
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string() : p.substr(0, k);
}
std::string base_of(const std::string &p) {
    size_t k = p.find_last_of("/\\");
    return k == std::string::npos ? p : p.substr(k + 1);
}
bool file_exists(const std::string &p) { return std::ifstream(p).good(); }

struct ObjMesh {
    std::vector<float> pos, uv;  // per corner
    std::vector<cl_uint> idx;
    bool has_uv = true;
    std::string material;
};

std::map<std::string, std::string> read_mtl(const std::string &path) {
    std::map<std::string, std::string> out;
    std::ifstream in(path);
    std::string line, name;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "newmtl") {
            std::getline(ss, name);
            name = trim(name);
        } else if (key == "map_Kd" && !name.empty()) {
            std::string rest;
            std::getline(ss, rest);
            out[name] = trim(rest);
        }
    }
    return out;
}

std::vector<ObjMesh> read_obj(const std::string &path, std::map<std::string, std::string> &mtl_tex) {
    std::ifstream in(path);
    if (!in) fail("ERROR: Assimp: Unable to open file \"" + path + "\".");
    std::vector<float> v, vt;
    std::vector<ObjMesh> meshes;
    ObjMesh cur;
    auto flush = [&]() {
        if (!cur.idx.empty()) meshes.push_back(cur);
        std::string m = cur.material;
        cur = ObjMesh();
        cur.material = m;
    };
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key) || key[0] == '#') continue;
        if (key == "v") {
            float a, b, c;
            ss >> a >> b >> c;
            v.insert(v.end(), {a, b, c});
        } else if (key == "vt") {
            float a = 0, b = 0;
            ss >> a >> b;
            vt.insert(vt.end(), {a, b});
        } else if (key == "mtllib") {
            std::string rest;
            std::getline(ss, rest);
            auto m = read_mtl(join_path(dir_of(path), trim(rest)));
            mtl_tex.insert(m.begin(), m.end());
        } else if (key == "o" || key == "g") {
            flush();
        } else if (key == "usemtl") {
            flush();
            std::string rest;
            std::getline(ss, rest);
            cur.material = trim(rest);
        } else if (key == "f") {
            std::vector<std::pair<int, int>> corners;
            std::string tok;
            while (ss >> tok) {
                int vi = 0, ti = 0;
                bool has_t = false;
                size_t s1 = tok.find('/');
                vi = std::stoi(tok.substr(0, s1));
                if (s1 != std::string::npos) {
                    size_t s2 = tok.find('/', s1 + 1);
                    std::string t = tok.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) { ti = std::stoi(t); has_t = true; }
                }
                vi = vi > 0 ? vi - 1 : (int)(v.size() / 3) + vi;
                if (has_t) ti = ti > 0 ? ti - 1 : (int)(vt.size() / 2) + ti;
                corners.push_back({vi, has_t ? ti : -1});
            }
            cl_uint b = (cl_uint)(cur.pos.size() / 3);
            for (auto &c : corners) {
                cur.pos.insert(cur.pos.end(), {v[3 * c.first], v[3 * c.first + 1], v[3 * c.first + 2]});
                if (c.second < 0) {
                    cur.has_uv = false;
                    cur.uv.insert(cur.uv.end(), {0.0f, 0.0f});
                } else cur.uv.insert(cur.uv.end(), {vt[2 * c.second], 1.0f - vt[2 * c.second + 1]});  // FlipUVs
            }
            for (size_t k = 1; k + 1 < corners.size(); k++) cur.idx.insert(cur.idx.end(), {b, b + (cl_uint)k, b + (cl_uint)k + 1});
        }
    }
    flush();
    if (meshes.empty()) fail("ERROR: Assimp: OBJ: no faces in \"" + path + "\"");
    return meshes;
}

// stb_image's 8-bit → float rule (stbi__ldr_to_hdr with the default l2h gamma 2.2f / scale 1.0f, which is what
// stbi_loadf applies, src/scene.cpp:158): colour = (float)pow(byte / 255.0f, 2.2f) — a FLOAT quotient and the
// float constant 2.2f, both promoted to double for pow() — and alpha = byte / 255.0f.  stb_image itself is not
// in the reference tree (un-vendored), so this restates its published formula; parity with it is unpinned.
struct LdrToHdr {
    float colour[256], alpha[256];
    LdrToHdr() {
        for (int i = 0; i < 256; i++) {
            float q = (float)i / 255.0f;
            colour[i] = (float)std::pow((double)q, (double)2.2f);
            alpha[i] = q;
        }
    }
};
const LdrToHdr &ldr_to_hdr() {
    static const LdrToHdr t;
    return t;
}

// PNG → 8-bit samples, `channels` per pixel.  Non-interlaced, 8 bits per sample, colour types 0 / 2 / 4 / 6
// (grey, RGB, grey+alpha, RGBA) — what the reference's textures are (assets/textures/die.png: 1024², RGBA8).
// zlib inflates the IDAT stream; the five scanline filters are undone here.  Returns "" or an error text.
std::string read_png(const std::string &path, std::vector<unsigned char> &pix, int &w, int &h, int &channels) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return "cannot open";
    std::vector<unsigned char> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) return "not a PNG file";
    auto be32 = [&](size_t at) {
        return ((uint32_t)file[at] << 24) | ((uint32_t)file[at + 1] << 16) | ((uint32_t)file[at + 2] << 8) | (uint32_t)file[at + 3];
    };
    std::vector<unsigned char> idat;
    bool have_ihdr = false;
    int depth = 0, ctype = 0, interlace = 0;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        uint32_t len = be32(pos);
        if (pos + 12 + (size_t)len > file.size()) return "truncated chunk";
        const unsigned char *type = &file[pos + 4], *data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) return "bad IHDR";
            w = (int)be32(pos + 8);
            h = (int)be32(pos + 12);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w < 1 || h < 1 || w > 32768 || h > 32768) return "bad header";
    if (depth != 8 || interlace != 0) return "only 8-bit non-interlaced PNG is supported";
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return "palette PNG is not supported";
    }
    const size_t stride = (size_t)w * channels;
    std::vector<unsigned char> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return "inflate failed";
    pix.assign(stride * (size_t)h, 0);
    const int bpp = channels;
    for (int y = 0; y < h; y++) {
        const unsigned char *src = &raw[(stride + 1) * (size_t)y];
        unsigned char *cur = &pix[stride * (size_t)y];
        const unsigned char *up = y ? cur - stride : nullptr;
        const int filter = src[0];
        for (size_t x = 0; x < stride; x++) {
            int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= (size_t)bpp) ? up[x - bpp] : 0;
            int pred;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return "bad scanline filter";
            }
            cur[x] = (unsigned char)(src[1 + x] + pred);
        }
    }
    return "";
}

// binary PPM (P6, maxval 255) or PFM (PF, little endian) → RGBA32F (alpha 1): an EXTENSION for this build's own
// synthetic assets — the reference only takes 4-channel images (src/scene.cpp:165-179)
bool read_ppm_pfm(const std::string &path, std::vector<float> &rgba, int &w, int &h) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::string magic;
    in >> magic;
    auto skip = [&]() {
        while (std::isspace(in.peek())) in.get();
        while (in.peek() == '#') { std::string c; std::getline(in, c); while (std::isspace(in.peek())) in.get(); }
    };
    if (magic == "P6") {
        int maxv;
        skip(); in >> w; skip(); in >> h; skip(); in >> maxv;
        in.get();
        if (!in || maxv != 255 || w < 1 || h < 1) return false;
        std::vector<unsigned char> buf((size_t)w * h * 3);
        in.read((char *)buf.data(), buf.size());
        if (!in) return false;
        const float *lut = ldr_to_hdr().colour;
        rgba.resize((size_t)w * h * 4);
        for (size_t i = 0; i < (size_t)w * h; i++) {
            rgba[4 * i] = lut[buf[3 * i]];
            rgba[4 * i + 1] = lut[buf[3 * i + 1]];
            rgba[4 * i + 2] = lut[buf[3 * i + 2]];
            rgba[4 * i + 3] = 1.0f;
        }
        return true;
    }
    if (magic == "PF") {
        float scale;
        in >> w >> h >> scale;
        in.get();
        if (!in || w < 1 || h < 1 || scale >= 0) return false;
        std::vector<float> buf((size_t)w * h * 3);
        in.read((char *)buf.data(), buf.size() * 4);
        if (!in) return false;
        rgba.resize((size_t)w * h * 4);
        for (int y = 0; y < h; y++)  // PFM rows run bottom to top
            for (int x = 0; x < w; x++) {
                const float *s = &buf[3 * ((size_t)(h - 1 - y) * w + x)];
                float *d = &rgba[4 * ((size_t)y * w + x)];
                d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = 1.0f;
            }
        return true;
    }
    return false;
}

}  // namespace

void SceneCreator::addMaterial(MatType type, const cl_float3 &color, cl_float extra_data) {
    Material m;
    std::memset(&m, 0, sizeof m);
    m.type = (int32_t)type;
    m.color = color;
    m.extra_data = extra_data;
    materials.push_back(m);
}

void SceneCreator::addSphere(const cl_float3 &pos, cl_float r, cl_uint mat_ID) {
    Sphere s;
    std::memset(&s, 0, sizeof s);
    s.pos = pos;
    s.r = r;
    s.mat_ID = mat_ID;
    spheres.push_back(s);
}

void SceneCreator::addPlane(const cl_float3 &pos, const cl_float3 &normal, cl_uint mat_ID) {
    Plane p;
    std::memset(&p, 0, sizeof p);
    p.pos = pos;
    p.normal = normal;
    p.mat_ID = mat_ID;
    planes.push_back(p);
}

// the lens is the intersection of two spheres of radii r1, r2 whose centres sit on the
// axis at ±sqrt(r² − h²) from pos, h = half aperture (src/scene.cpp:122-143)
void SceneCreator::addLens(const cl_float3 &pos, const cl_float3 &normal, cl_float r1, cl_float r2, cl_float h, cl_uint mat_ID) {
    if (!(r1 >= h && r2 >= h)) fail("ERROR: SCENE: LENS RADII MUST BE >= h");
    Lens l;
    std::memset(&l, 0, sizeof l);
    l.pos = pos;
    float d1 = (float)std::sqrt(r1 * r1 - h * h), d2 = (float)std::sqrt(r2 * r2 - h * h);
    l.p1 = make_float3(pos.x + normal.x * d1, pos.y + normal.y * d1, pos.z + normal.z * d1);
    l.p2 = make_float3(pos.x - normal.x * d2, pos.y - normal.y * d2, pos.z - normal.z * d2);
    l.r1 = r1;
    l.r2 = r2;
    l.mat_ID = mat_ID;
    lenses.push_back(l);
}

void SceneCreator::loadModel(const std::string &path, cl_uint mat_ID, const mat4 &m) {
    if (materials.size() <= mat_ID) fail("ERROR: MATERIAL OF ID: " + std::to_string(mat_ID) + " DOES NOT EXIST");
    std::map<std::string, std::string> mtl_tex;
    std::vector<ObjMesh> parts = read_obj(path, mtl_tex);
    {
        std::string d = dir_of(path);
        if (d.empty()) d = ".";
        bool seen = false;
        for (const std::string &e : model_dirs) seen = seen || e == d;
        if (!seen) model_dirs.push_back(d);
    }
    bool textured = materials[mat_ID].type == t_textured;
    for (const ObjMesh &om : parts) {
        Mesh mesh;
        mesh.vertex_anchor = (cl_uint)vertices.size();
        mesh.index_anchor = (cl_uint)indices.size();
        mesh.face_count = (cl_uint)(om.idx.size() / 3);
        mesh.texture_ID = (cl_uint)-1;
        if (texture_uv.size() < vertices.size()) texture_uv.resize(vertices.size(), cl_float2{0.0f, 0.0f});
        for (size_t i = 0; i < om.pos.size() / 3; i++) {
            float x = om.pos[3 * i], y = om.pos[3 * i + 1], z = om.pos[3 * i + 2];
            // column-major mat4 × (x, y, z, 1), src/scene.cpp:226-232
            vertices.push_back(make_float3(m.c[0][0] * x + m.c[1][0] * y + m.c[2][0] * z + m.c[3][0],
                                           m.c[0][1] * x + m.c[1][1] * y + m.c[2][1] * z + m.c[3][1],
                                           m.c[0][2] * x + m.c[1][2] * y + m.c[2][2] * z + m.c[3][2]));
            texture_uv.push_back(om.has_uv ? cl_float2{om.uv[2 * i], om.uv[2 * i + 1]} : cl_float2{0.0f, 0.0f});
        }
        indices.insert(indices.end(), om.idx.begin(), om.idx.end());
        if (textured) {
            auto it = mtl_tex.find(om.material);
            if (it == mtl_tex.end()) fail("ERROR: MESH HAS NO TEXTURE APPLIED, USE A DIFFERENT MATERIAL");
            size_t j = 0;
            for (; j < texture_paths.size(); j++)
                if (texture_paths[j] == it->second) break;  // dedup by path string, src/scene.cpp:272-283
            if (j == texture_paths.size()) texture_paths.push_back(it->second);
            mesh.texture_ID = (cl_uint)j;
        }
        meshes.push_back(mesh);
    }
    models.push_back(Model{mesh_count_total, (cl_uint)parts.size(), mat_ID});
    mesh_count_total += (cl_uint)parts.size();
}

void SceneCreator::setTextures(const float *rgba, int w, int h, int layers) {
    if (!rgba || w < 1 || h < 1 || layers < 1) fail("ERROR: TEXTURES: bad array");
    texture_data.assign(rgba, rgba + (size_t)w * h * layers * 4);
    tex_w = w;
    tex_h = h;
    tex_layers = layers;
}

// Texture files.  The reference hands every map_Kd string to stbi_loadf as it is (src/scene.cpp:158); its own
// .mtl files name an absolute path on the author's machine (assets/cube/cube.mtl:13), so a path that does not
// exist is looked up again by its BASE NAME next to the models and the scene: <model dir>/../textures,
// <model dir>, <scene dir>/../textures, ./assets/textures (the reference's tree seen from its working
// directory), <base dir>/textures.  Relative paths are also tried against the base / model directories.
// A PNG must be 4-channel, as in the reference (src/scene.cpp:165-179); PPM/PFM stand-ins (same stem) are an
// extension for synthetic assets.
void SceneCreator::loadTextures() {
    if (models.empty()) {
        tex_layers = 0;
        return;
    }
    // The reference exits here with "ERROR: TEXTURE COUNT = 0" whenever a scene has models but no texture path was
    // collected (src/scene.cpp:147-148) — i.e. for EVERY model whose material is not t_textured (processMesh only
    // collects paths for that type, :264), although the kernel never fetches a texel for such a model.  BASELINE.json's
    // configuration 5 is exactly that (an OBJ mesh with a dielectric material), so this is accepted: no layers.
    if (texture_paths.empty()) {
        tex_layers = 0;
        texture_data.clear();
        return;
    }
    texture_data.clear();
    tex_layers = 0;
    std::vector<std::string> dirs;
    auto add_dir = [&](const std::string &d) {
        for (const std::string &e : dirs) if (e == d) return;
        dirs.push_back(d);
    };
    for (const std::string &d : model_dirs) { add_dir(join_path(d, "../textures")); add_dir(d); }
    if (!scene_dir.empty()) add_dir(join_path(scene_dir, "../textures"));
    add_dir("assets/textures");
    if (!base_dir.empty()) { add_dir(join_path(base_dir, "textures")); add_dir(base_dir); }
    for (size_t id = 0; id < texture_paths.size(); id++) {
        const std::string &p = texture_paths[id];
        std::vector<std::string> candidates = {p};
        if (!p.empty() && p[0] != '/') {
            if (!base_dir.empty()) candidates.push_back(join_path(base_dir, p));
            for (const std::string &d : model_dirs) candidates.push_back(join_path(d, p));
        }
        for (const std::string &d : dirs) candidates.push_back(join_path(d, base_of(p)));
        const size_t n_as_named = candidates.size();
        for (size_t k = 0; k < n_as_named; k++) {  // PPM / PFM stand-ins with the same stem
            std::string stem = candidates[k].substr(0, candidates[k].find_last_of('.'));
            candidates.push_back(stem + ".ppm");
            candidates.push_back(stem + ".pfm");
        }
        std::vector<float> img;
        int w = 0, h = 0;
        bool ok = false;
        for (const std::string &c : candidates) {
            if (!file_exists(c)) continue;
            std::vector<unsigned char> pix;
            int ch = 0;
            std::string err = read_png(c, pix, w, h, ch);
            if (err.empty()) {
                if (ch != 4) fail("ERROR: STBimage: TEXTURE HAS A WRONG FORMAT: " + std::to_string(ch) + " INSTEAD OF 4 (RGBA)");
                const LdrToHdr &t = ldr_to_hdr();
                img.resize((size_t)w * h * 4);
                for (size_t i = 0; i < (size_t)w * h; i++) {
                    img[4 * i] = t.colour[pix[4 * i]];
                    img[4 * i + 1] = t.colour[pix[4 * i + 1]];
                    img[4 * i + 2] = t.colour[pix[4 * i + 2]];
                    img[4 * i + 3] = t.alpha[pix[4 * i + 3]];
                }
                ok = true;
                break;
            }
            if (err != "not a PNG file") fail("ERROR: STBimage: " + err + ": " + c);
            if (read_ppm_pfm(c, img, w, h)) { ok = true; break; }
        }
        if (!ok) fail("ERROR: STBimage: COULD NOT FIND THE TEXTURE: " + p);
        if (id == 0) { tex_w = w; tex_h = h; }
        else if (w != tex_w || h != tex_h)
            fail("ERROR: TEXTURES HAVE DIFFERENT SIZES: TEMPLATE: " + std::to_string(tex_w) + " x " + std::to_string(tex_h) +
                 ", TEXTURE ID(" + std::to_string(id) + "): " + std::to_string(w) + " x " + std::to_string(h));
        texture_data.insert(texture_data.end(), img.begin(), img.end());
        tex_layers++;
    }
}

void SceneCreator::loadScene(const std::string &path) {
    std::ifstream in(path);
    if (!in) fail("ERROR: SCENE: NOT SUCCESFULLY READ: " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    scene_dir = dir_of(path);
    if (base_dir.empty()) base_dir = dir_of(scene_dir);  // assets/scenes/x.scene → assets/
    loadSceneText(ss.str());
}

// `load:` paths are used AS WRITTEN, i.e. relative to the working directory, like the reference
// (src/scene.cpp:355 → :195; assets/scenes/scene.scene:32 says "assets/cube/cube.obj").  Only when that file
// does not exist is the path tried against the base directory and the scene's own directory.
std::string SceneCreator::resolveModelPath(const std::string &p) const {
    if (file_exists(p) || (!p.empty() && p[0] == '/')) return p;
    if (!base_dir.empty() && file_exists(join_path(base_dir, p))) return join_path(base_dir, p);
    if (!scene_dir.empty() && file_exists(join_path(scene_dir, p))) return join_path(scene_dir, p);
    return p;
}

void SceneCreator::loadSceneText(const std::string &text) {
    enum Section { none, mats, sph, pla, len, mod } section = none;
    mat4 model(1.0f);
    std::istringstream in(text);
    std::string line;
    while (std::getline(in, line)) {
        size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        if (line.empty()) continue;
        size_t colon = line.find(':');
        if (colon != std::string::npos) {
            std::string word = line.substr(0, colon);
            if (word == "MATERIALS") { section = mats; continue; }
            if (word == "SPHERES") { section = sph; continue; }
            if (word == "PLANES") { section = pla; continue; }
            if (word == "LENSES") { section = len; continue; }
            if (word == "MODELS") { section = mod; continue; }
            if (section != mod) fail("ERROR: SCENE: OPERATION " + word + " DOES NOT EXIST");
            Fields f(line.substr(colon + 1));
            if (word == "translate") {
                cl_float3 v = f.getVec();
                model = rth::translate(model, vec3(v.x, v.y, v.z));
            } else if (word == "rotate") {
                float deg = f.getFloat();
                cl_float3 v = f.getVec();
                model = rth::rotate(model, rth::radians(deg), vec3(v.x, v.y, v.z));
            } else if (word == "scale") {
                cl_float3 v = f.getVec();
                model = rth::scale(model, vec3(v.x, v.y, v.z));
            } else if (word == "load") {
                std::string p = f.getPath();
                cl_uint mat = f.getUInt();
                loadModel(resolveModelPath(p), mat, model);
                model = mat4(1.0f);
            }
            continue;
        }
        Fields f(line);
        switch (section) {
            case mats: {
                std::string word = f.next();
                static const std::map<std::string, MatType> names = {
                    {"reflective", t_reflective}, {"refractive", t_refractive}, {"diffuse", t_diffuse},
                    {"dielectric", t_dielectric}, {"light", t_light},           {"textured", t_textured}};
                auto it = names.find(word);
                if (it == names.end()) fail("ERROR: SCENE: MATERIAL: " + word + " DOES NOT EXIST");
                cl_float3 col = f.getVec();
                float extra = f.getFloat();
                addMaterial(it->second, col, extra);
                break;
            }
            case sph: {
                cl_float3 p = f.getVec();
                float r = f.getFloat();
                addSphere(p, r, f.getUInt());
                break;
            }
            case pla: {
                cl_float3 p = f.getVec();
                cl_float3 n = f.getVec();
                addPlane(p, n, f.getUInt());
                break;
            }
            case len: {
                cl_float3 p = f.getVec();
                cl_float3 n = f.getVec();
                float r1 = f.getFloat(), r2 = f.getFloat(), h = f.getFloat();
                addLens(p, n, r1, r2, h, f.getUInt());
                break;
            }
            default: fail("ERROR: SCENE: OPERATION NOT SPECIFIED");
        }
    }
}

rt_scene_desc SceneCreator::describe() {
    if (texture_uv.size() < vertices.size()) texture_uv.resize(vertices.size(), cl_float2{0.0f, 0.0f});
    rt_scene_desc d;
    std::memset(&d, 0, sizeof d);
    d.materials = materials.data();   d.material_count = (uint32_t)materials.size();
    d.spheres = spheres.data();       d.sphere_count = (uint32_t)spheres.size();
    d.planes = planes.data();         d.plane_count = (uint32_t)planes.size();
    d.lenses = lenses.data();         d.lens_count = (uint32_t)lenses.size();
    d.vertices = vertices.data();     d.vertex_count = (uint32_t)vertices.size();
    d.uvs = texture_uv.data();        d.uv_count = (uint32_t)texture_uv.size();
    d.indices = indices.data();       d.index_count = (uint32_t)indices.size();
    d.meshes = meshes.data();         d.mesh_count = (uint32_t)meshes.size();
    d.models = models.data();         d.model_count = (uint32_t)models.size();
    return d;
}
