// pt_scene.h — reference-format scene file (Renderer/main.cpp:117-191) -> what the two path-tracing consumers need:
// geometry for shading, the 4 hard-coded materials (main.cpp:165-168), the camera (Camera::lookAt, Camera.cpp:13-26) and
// the environment probe.  Host-only; shared by pathtracer.cpp and pt_device.hip so both see identical inputs.
#ifndef RACC_PT_SCENE_H
#define RACC_PT_SCENE_H

#include "pt_shade.h"

#include <cmath>
#include <cstdio>
#include <vector>

namespace ptscene {

#pragma pack(push, 1)
struct SceneHeader {   // Renderer/main.cpp:118-133
    uint32_t maxDepth, vertexCount, triangleCount;
    uint16_t viewportWidth, viewportHeight, environmentWidth, environmentHeight;
    float origin[3], target[3], up[3], fov;
};
#pragma pack(pop)
static_assert(sizeof(SceneHeader) == 60, "scene header layout");

struct Scene {
    SceneHeader hdr;
    std::vector<float> vertices;           // xyzw per vertex (16-byte records, Scene.cpp:187)
    std::vector<uint32_t> indices;
    std::vector<uint16_t> triangleMaterials;
    std::vector<float> normals;            // xyzw per vertex
    std::vector<float> env;                // RGBA32F
    ptshade::Materials mat;
    ptshade::Camera cam;
};

// Returns 0, or -2 with one line on stderr.
inline int load(const char* path, uint32_t width, uint32_t height, Scene& s) {
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(&s.hdr, sizeof(s.hdr), 1, f) != 1) { if (f) std::fclose(f); std::fprintf(stderr, "racc_pt: cannot read %s\n", path); return -2; }
    const uint32_t T = s.hdr.triangleCount, V = s.hdr.vertexCount;
    s.indices.resize(size_t(T) * 3); s.triangleMaterials.resize(T); s.vertices.resize(size_t(V) * 4); s.normals.resize(size_t(V) * 4);
    s.env.resize(size_t(s.hdr.environmentWidth) * s.hdr.environmentHeight * 4);
    bool ok = std::fread(s.indices.data(), 12, T, f) == T;
    ok = ok && std::fread(s.triangleMaterials.data(), 2, T, f) == T;
    ok = ok && std::fseek(f, long(T) * 16, SEEK_CUR) == 0;                       // per-triangle normals: recomputed from the vertices
    ok = ok && std::fread(s.vertices.data(), 16, V, f) == V;
    ok = ok && std::fread(s.normals.data(), 16, V, f) == V;
    ok = ok && std::fseek(f, long(V) * 8, SEEK_CUR) == 0;                        // texture coordinates: unused by the 4 materials
    ok = ok && std::fread(s.env.data(), 16, s.env.size() / 4, f) == s.env.size() / 4;
    std::fclose(f);
    if (!ok) { std::fprintf(stderr, "racc_pt: short scene file\n"); return -2; }

    const float mats[4][4] = {{0.8f, 0.8f, 0.8f, 1.0f / 1.4f}, {0.1f, 0.1f, 0.1f, 1.0f / 1.4f},    // main.cpp:165-168
                              {0.6f, 0.6f, 0.6f, 1.0f / 1.2f}, {0.3f, 0.3f, 0.3f, 1.0f / 1.2f}};
    for (int m = 0; m < 4; ++m) { for (int ch = 0; ch < 3; ++ch) s.mat.kd[m][ch] = mats[m][ch]; s.mat.eta[m] = mats[m][3]; }
    {   // Camera::lookAt, Camera.cpp:13-26
        using ptshade::Vec;
        const Vec o{s.hdr.origin[0], s.hdr.origin[1], s.hdr.origin[2]};
        const Vec fwd = normalize(Vec{s.hdr.target[0], s.hdr.target[1], s.hdr.target[2]} - o);
        const Vec right = normalize(cross(fwd, Vec{s.hdr.up[0], s.hdr.up[1], s.hdr.up[2]}));
        const Vec up = cross(right, fwd);
        const float ey = std::tan(0.5f * s.hdr.fov * 3.14159265f / 180.0f), ex = ey * float(width) / float(height);
        s.cam.origin = o;
        s.cam.right = right * (-2.0f / float(width) * ex);
        s.cam.up = up * (-2.0f / float(height) * ey);
        s.cam.view = fwd + right * ex + up * ey;
    }
    return 0;
}

}  // namespace ptscene
#endif
