// pt_shade.h — the path tracer's per-ray arithmetic, shared VERBATIM by the host-shading consumer (pathtracer.cpp,
// callbacks on CPU threads as in the reference) and the device-resident consumer (pt_device.hip, one HIP kernel per
// bounce).  Everything here is plain IEEE binary32/binary64 arithmetic in a fixed order (both translation units are
// compiled with -ffp-contract=off, fused operations are written as fmaf), sqrt and division are correctly rounded on
// both sides, and sine/cosine come from the polynomial below rather than from two different math libraries — so the
// two consumers produce the same image bit for bit, which is what tests/test_gpu_pathtracer.py asserts.
//
// What it mirrors (file:line into the reference checkout):
//   camera ray                  Renderer/Camera.cpp:55-85
//   payload                     Renderer/LightPath.h:14-17  weight[3] + pixel (low 24 bits) | depth (high 8)
//   shade                       Renderer/PathTracingRenderer.cpp:72-566, material Renderer/Materials.cpp:39-151
#ifndef RACC_PT_SHADE_H
#define RACC_PT_SHADE_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RACC_HD __host__ __device__ __forceinline__
#else
#define RACC_HD inline
#endif

namespace ptshade {

struct Vec { float x, y, z; };
RACC_HD Vec operator+(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RACC_HD Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RACC_HD Vec operator*(Vec a, float s) { return {a.x * s, a.y * s, a.z * s}; }
RACC_HD float dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RACC_HD Vec cross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RACC_HD Vec normalize(Vec a) { return a * (1.0f / sqrtf(dot(a, a))); }

// Counter-based RNG keyed by (pixel, sample, depth, stream): a path's random numbers do not depend on which thread,
// lane or GPU shades it (the reference seeds per-thread streams with rand(), Camera.cpp:58, PathTracingRenderer.cpp:102).
RACC_HD uint32_t pcg(uint32_t x) {
    x = x * 747796405u + 2891336453u;
    const uint32_t w = ((x >> ((x >> 28) + 4)) ^ x) * 277803737u;
    return (w >> 22) ^ w;
}
RACC_HD uint32_t pathKey(uint32_t pixel, uint32_t sample) { return pcg(pcg(pixel) ^ (sample * 0x9E3779B9u)); }
RACC_HD float uniformKeyed(uint32_t key, uint32_t depth, uint32_t stream) {
    const uint32_t h = pcg(key ^ (depth * 0x85EBCA6Bu + stream * 0xC2B2AE35u));
    return float(h >> 8) * (1.0f / 16777216.0f);
}

// sin and cos of 2*pi*r for r in [0,1): exact quadrant reduction (r has 24 significant bits, so 4r, floor and the
// difference are exact), then fixed polynomials on [0, pi/2] evaluated with fmaf.  |error| < 1e-7.
RACC_HD void sincos2pi(float r, float& s, float& c) {
    const float t = r * 4.0f;
    const float q = floorf(t);
    const float a = (t - q) * 1.57079632679489662f;
    const float a2 = a * a;
    float sp = -2.50521083854417188e-8f;                         // -1/11!
    sp = fmaf(sp, a2, 2.75573192239858907e-6f);                   // 1/9!
    sp = fmaf(sp, a2, -1.98412698412698413e-4f);                  // -1/7!
    sp = fmaf(sp, a2, 8.33333333333333333e-3f);                   // 1/5!
    sp = fmaf(sp, a2, -1.66666666666666667e-1f);                  // -1/3!
    const float sa = fmaf(sp * a2, a, a);
    float cp = 2.08767569878680990e-9f;                          // 1/12!
    cp = fmaf(cp, a2, -2.75573192239858907e-7f);                  // -1/10!
    cp = fmaf(cp, a2, 2.48015873015873016e-5f);                   // 1/8!
    cp = fmaf(cp, a2, -1.38888888888888889e-3f);                  // -1/6!
    cp = fmaf(cp, a2, 4.16666666666666667e-2f);                   // 1/4!
    cp = fmaf(cp, a2, -0.5f);
    const float ca = fmaf(cp, a2, 1.0f);
    const int qi = int(q) & 3;
    s = qi == 0 ? sa : qi == 1 ? ca : qi == 2 ? -sa : -ca;
    c = qi == 0 ? ca : qi == 1 ? -sa : qi == 2 ? -ca : sa;
}

struct Camera { Vec origin, right, up, view; };      // Camera::lookAt products, Camera.cpp:13-26

struct RayRec { float origin[3], minT, dir[3], maxT; };                   // RayAccelerator.h:59-64
struct HitRec { uint32_t triangle; float t, u, v; };                       // RayAccelerator.h:66-76 (miss: rgb in t,u,v)
struct PathRec { float weight[3]; uint32_t pixelDepth; };                  // LightPath.h:14-17

struct Materials { float kd[4][3]; float eta[4]; };                        // Renderer/main.cpp:165-168

struct SceneView {                                                          // shading inputs (host or device pointers)
    const uint32_t* indices;            // 3 per triangle
    const uint16_t* triangleMaterials;
    const float* normals;               // xyzw per vertex
    const float* vertices;              // xyzw per vertex
    uint32_t triangleCount;
};

constexpr double kFixed = 1048576.0;    // 2^20: frame-buffer resolution of one accumulated contribution

// Camera.cpp:55-85 with jittered samples from the counter RNG.
RACC_HD void primaryRay(const Camera& cam, uint32_t x, uint32_t y, uint32_t pixel, uint32_t sample, RayRec& ray, PathRec& path) {
    const uint32_t key = pathKey(pixel, sample);
    const float px = float(x) + uniformKeyed(key, 0, 1), py = float(y) + uniformKeyed(key, 0, 2);
    const Vec d = normalize(cam.view + cam.right * px + cam.up * py);
    ray.origin[0] = cam.origin.x; ray.origin[1] = cam.origin.y; ray.origin[2] = cam.origin.z; ray.minT = 0.0f;
    ray.dir[0] = d.x; ray.dir[1] = d.y; ray.dir[2] = d.z; ray.maxT = 1e6f;
    path.weight[0] = path.weight[1] = path.weight[2] = 1.0f;
    path.pixelDepth = pixel;
}

// Materials.cpp:39-151, scalar: Fresnel-weighted choice between the mirror direction and a cosine-weighted diffuse
// direction.  Returns false if the path dies.
RACC_HD bool sampleMaterial(const Materials& mat, unsigned m, Vec n, Vec wo, float r1, float r2, float r3, Vec& wi, float colour[3]) {
    const float d0 = dot(n, wo);
    const float cosi = d0 > 0.0f ? d0 : 0.0f;
    const Vec refl = n * (2.0f * cosi) - wo;
    const float e = mat.eta[m];
    const float k = 1.0f + e * e * (cosi * cosi - 1.0f);
    float fresnel = 1.0f;                      // total internal reflection (Materials.cpp:83: blendv on the sign of k)
    if (k >= 0.0f) {
        const float cost = sqrtf(k);
        const float rper = (e * cosi - cost) / (e * cosi + cost);
        const float rpar = -(e * cost - cosi) / (e * cost + cosi);
        fresnel = 0.5f * (rpar * rpar + rper * rper);
    }
    Vec bu = fabsf(n.x) > 0.1f ? Vec{-n.z, 0.0f, n.x} : Vec{0.0f, -n.z, n.y};   // Materials.cpp:86-93
    bu = normalize(bu);
    const Vec bv = cross(n, bu);
    float sn, cs;
    sincos2pi(r1, sn, cs);
    const float s = sqrtf(r2), c = sqrtf(1.0f - r2);
    const Vec diffuse = normalize(n * c + (bu * cs + bv * sn) * s);
    const float s0 = 3.0f * fresnel, s1 = mat.kd[m][0] + mat.kd[m][1] + mat.kd[m][2], sum = s0 + s1;    // Materials.cpp:121-128
    const bool pickDiffuse = r3 * sum >= s0;
    wi = pickDiffuse ? diffuse : refl;
    float rgb[3];
    for (int ch = 0; ch < 3; ++ch) rgb[ch] = pickDiffuse ? mat.kd[m][ch] : fresnel;
    const float denom = rgb[0] + rgb[1] + rgb[2];
    if (!(denom > 0.0f)) return false;
    const float scale = sum / denom;                                                                   // Materials.cpp:138
    for (int ch = 0; ch < 3; ++ch) colour[ch] = rgb[ch] * scale;
    return true;
}

RACC_HD bool finiteF(float v) { return (v - v) == 0.0f; }
RACC_HD bool finiteD(double v) { return (v - v) == 0.0; }

// Fixed-point contribution of a path that left the scene (PathTracingRenderer.cpp:505-563): out[ch] to be ADDED to
// frame[pixel*3 + ch]; valid[ch] false when the product is not finite.
RACC_HD void missContribution(const HitRec& hit, const PathRec& path, long long out[3], bool valid[3]) {
    const float env[3] = {hit.t, hit.u, hit.v};
    for (int ch = 0; ch < 3; ++ch) {
        const double v = double(env[ch]) * double(path.weight[ch]);
        valid[ch] = finiteD(v);
        out[ch] = valid[ch] ? llround(v * kFixed) : 0;
    }
}

// Unnormalised-orientation geometric normal of a triangle from its three xyzw vertices (the side flip happens per ray).
RACC_HD Vec geometricNormal(const float* a, const float* b, const float* c) {
    return normalize(cross(Vec{b[0] - a[0], b[1] - a[1], b[2] - a[2]}, Vec{c[0] - a[0], c[1] - a[1], c[2] - a[2]}));
}

// One surface interaction (PathTracingRenderer.cpp:113-422) given the triangle's three vertex normals, its geometric
// normal and its material.  Returns true and fills (nextRay, nextPath) when the path continues.  `sample` is the path's
// sample index (RNG key).  The caller has already checked depth < maxDepth and the triangle index (:113-114).
RACC_HD bool shadeSurface(const Materials& mat, const RayRec& ray, const HitRec& hit, const PathRec& path, uint32_t sample,
                          const float* n0, const float* n1, const float* n2, Vec ng, unsigned m, RayRec& nextRay, PathRec& nextPath) {
    const uint32_t pixel = path.pixelDepth & 0xFFFFFFu, depth = path.pixelDepth >> 24;
    const float u = hit.u, v = hit.v, w = 1.0f - u - v;                                       // :218-227: w,u,v weight index 0,1,2
    Vec n = normalize(Vec{n0[0] * w + n1[0] * u + n2[0] * v, n0[1] * w + n1[1] * u + n2[1] * v, n0[2] * w + n1[2] * u + n2[2] * v});
    const Vec d{ray.dir[0], ray.dir[1], ray.dir[2]};
    const Vec wo = d * -1.0f;
    if (dot(ng, wo) < 0.0f) ng = ng * -1.0f;        // geometric normal toward the viewer side
    if (dot(n, wo) < 0.0f) n = n * -1.0f;
    Vec wi;
    float colour[3];
    const uint32_t key = pathKey(pixel, sample);
    if (!sampleMaterial(mat, m, n, wo, uniformKeyed(key, depth + 1, 3), uniformKeyed(key, depth + 1, 4), uniformKeyed(key, depth + 1, 5), wi, colour)) return false;
    const float wgt[3] = {path.weight[0] * colour[0], path.weight[1] * colour[1], path.weight[2] * colour[2]};
    if (!(wgt[0] > 0.01f || wgt[1] > 0.01f || wgt[2] > 0.01f)) return false;                   // :394-399
    if (!(dot(wi, ng) > 0.0f)) return false;                                                   // :401-403 (no transmission)
    const Vec p = Vec{ray.origin[0], ray.origin[1], ray.origin[2]} + d * hit.t + ng * 1e-4f;   // :410-412
    if (!(finiteF(p.x + p.y + p.z) && finiteF(wi.x + wi.y + wi.z))) return false;               // :416-418
    nextRay.origin[0] = p.x; nextRay.origin[1] = p.y; nextRay.origin[2] = p.z; nextRay.minT = 1e-3f;
    nextRay.dir[0] = wi.x; nextRay.dir[1] = wi.y; nextRay.dir[2] = wi.z; nextRay.maxT = 1e6f;
    nextPath.weight[0] = wgt[0]; nextPath.weight[1] = wgt[1]; nextPath.weight[2] = wgt[2];
    nextPath.pixelDepth = pixel | ((depth + 1) << 24);                                          // :414
    return true;
}

// The same, gathering the triangle's data from the scene arrays (what the host consumer does per hit).
RACC_HD bool shadeHit(const SceneView& sc, const Materials& mat, uint32_t maxDepth, const RayRec& ray, const HitRec& hit,
                      const PathRec& path, uint32_t sample, RayRec& nextRay, PathRec& nextPath) {
    if ((path.pixelDepth >> 24) >= maxDepth || hit.triangle >= sc.triangleCount) return false;   // :113-114
    const uint32_t* tri = sc.indices + size_t(hit.triangle) * 3;
    const Vec ng = geometricNormal(sc.vertices + size_t(tri[0]) * 4, sc.vertices + size_t(tri[1]) * 4, sc.vertices + size_t(tri[2]) * 4);
    return shadeSurface(mat, ray, hit, path, sample, sc.normals + size_t(tri[0]) * 4, sc.normals + size_t(tri[1]) * 4,
                        sc.normals + size_t(tri[2]) * 4, ng, sc.triangleMaterials[hit.triangle] & 3u, nextRay, nextPath);
}

// Per-triangle shading record of the device consumer: everything shadeSurface needs in ONE aligned 64 B gather instead
// of 3 indices + 3 normals + 3 vertices in seven different cache lines.  Built on the host with the functions above, so
// the values are the ones the host consumer computes per hit.
struct alignas(64) ShadeTri { float n0[3], n1[3], n2[3], ng[3]; uint32_t material; uint32_t pad[3]; };

inline void buildShadeTri(const SceneView& sc, uint32_t triangle, ShadeTri& out) {
    const uint32_t* tri = sc.indices + size_t(triangle) * 3;
    for (int k = 0; k < 3; ++k) {
        out.n0[k] = sc.normals[size_t(tri[0]) * 4 + k]; out.n1[k] = sc.normals[size_t(tri[1]) * 4 + k]; out.n2[k] = sc.normals[size_t(tri[2]) * 4 + k];
    }
    const Vec ng = geometricNormal(sc.vertices + size_t(tri[0]) * 4, sc.vertices + size_t(tri[1]) * 4, sc.vertices + size_t(tri[2]) * 4);
    out.ng[0] = ng.x; out.ng[1] = ng.y; out.ng[2] = ng.z;
    out.material = sc.triangleMaterials[triangle] & 3u;
    out.pad[0] = out.pad[1] = out.pad[2] = 0;
}

}  // namespace ptshade
#endif
