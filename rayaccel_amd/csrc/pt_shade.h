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


// ------------------------------------------------------------------------------------------------------------------------------------
// Eight surface interactions at a time (host consumer only; AVX2 + FMA — the reference's own shading code is 8-wide AVX2,
// Renderer/PathTracingRenderer.cpp:72-566, Renderer/Materials.cpp:39-151).  Lane i of every operation below is the scalar code above
// applied to hit i: the same IEEE operations in the same order — vmulps/vaddps where the scalar code multiplies and adds, vfmadd only
// where it says fmaf, vdivps/vsqrtps (correctly rounded, like / and sqrtf), branches turned into selects of values both sides compute —
// so shadeSurface8 returns bit for bit what eight calls of shadeSurface return (racc_pt_test_shade8, tests/test_host_build.py) and the
// host and device consumers keep rendering identical images.
#if !defined(__HIPCC__) && defined(__AVX2__) && defined(__FMA__)
}  // namespace ptshade
#include <immintrin.h>
namespace ptshade {
namespace simd {

typedef __m256 F8;
typedef __m256i I8;
struct Vec8 { F8 x, y, z; };
inline F8 set1(float v) { return _mm256_set1_ps(v); }
inline Vec8 operator+(Vec8 a, Vec8 b) { return {_mm256_add_ps(a.x, b.x), _mm256_add_ps(a.y, b.y), _mm256_add_ps(a.z, b.z)}; }
inline Vec8 operator-(Vec8 a, Vec8 b) { return {_mm256_sub_ps(a.x, b.x), _mm256_sub_ps(a.y, b.y), _mm256_sub_ps(a.z, b.z)}; }
inline Vec8 operator*(Vec8 a, F8 s) { return {_mm256_mul_ps(a.x, s), _mm256_mul_ps(a.y, s), _mm256_mul_ps(a.z, s)}; }
inline F8 dot(Vec8 a, Vec8 b) { return _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(a.x, b.x), _mm256_mul_ps(a.y, b.y)), _mm256_mul_ps(a.z, b.z)); }
inline Vec8 cross(Vec8 a, Vec8 b) {
    return {_mm256_sub_ps(_mm256_mul_ps(a.y, b.z), _mm256_mul_ps(a.z, b.y)), _mm256_sub_ps(_mm256_mul_ps(a.z, b.x), _mm256_mul_ps(a.x, b.z)),
            _mm256_sub_ps(_mm256_mul_ps(a.x, b.y), _mm256_mul_ps(a.y, b.x))};
}
inline Vec8 normalize(Vec8 a) { return a * _mm256_div_ps(set1(1.0f), _mm256_sqrt_ps(dot(a, a))); }
inline F8 sel(F8 mask, F8 yes, F8 no) { return _mm256_blendv_ps(no, yes, mask); }      // mask ? yes : no
inline Vec8 sel(F8 mask, Vec8 yes, Vec8 no) { return {sel(mask, yes.x, no.x), sel(mask, yes.y, no.y), sel(mask, yes.z, no.z)}; }
inline F8 neg(F8 a) { return _mm256_xor_ps(a, set1(-0.0f)); }                          // unary minus: the sign bit, NaNs included

inline I8 pcg8(I8 x) {
    x = _mm256_add_epi32(_mm256_mullo_epi32(x, _mm256_set1_epi32(int(747796405u))), _mm256_set1_epi32(int(2891336453u)));
    const I8 sh = _mm256_add_epi32(_mm256_srli_epi32(x, 28), _mm256_set1_epi32(4));
    const I8 w = _mm256_mullo_epi32(_mm256_xor_si256(_mm256_srlv_epi32(x, sh), x), _mm256_set1_epi32(int(277803737u)));
    return _mm256_xor_si256(_mm256_srli_epi32(w, 22), w);
}
inline I8 pathKey8(I8 pixel, I8 sample) { return pcg8(_mm256_xor_si256(pcg8(pixel), _mm256_mullo_epi32(sample, _mm256_set1_epi32(int(0x9E3779B9u))))); }
inline F8 uniformKeyed8(I8 key, I8 depth, uint32_t stream) {
    const I8 h = pcg8(_mm256_xor_si256(key, _mm256_add_epi32(_mm256_mullo_epi32(depth, _mm256_set1_epi32(int(0x85EBCA6Bu))), _mm256_set1_epi32(int(stream * 0xC2B2AE35u)))));
    return _mm256_mul_ps(_mm256_cvtepi32_ps(_mm256_srli_epi32(h, 8)), set1(1.0f / 16777216.0f));      // (h >> 8 < 2^24: the signed conversion is the unsigned one)
}

inline void sincos2pi8(F8 r, F8& s, F8& c) {
    const F8 t = _mm256_mul_ps(r, set1(4.0f));
    const F8 q = _mm256_floor_ps(t);
    const F8 a = _mm256_mul_ps(_mm256_sub_ps(t, q), set1(1.57079632679489662f));
    const F8 a2 = _mm256_mul_ps(a, a);
    F8 sp = set1(-2.50521083854417188e-8f);
    sp = _mm256_fmadd_ps(sp, a2, set1(2.75573192239858907e-6f));
    sp = _mm256_fmadd_ps(sp, a2, set1(-1.98412698412698413e-4f));
    sp = _mm256_fmadd_ps(sp, a2, set1(8.33333333333333333e-3f));
    sp = _mm256_fmadd_ps(sp, a2, set1(-1.66666666666666667e-1f));
    const F8 sa = _mm256_fmadd_ps(_mm256_mul_ps(sp, a2), a, a);
    F8 cp = set1(2.08767569878680990e-9f);
    cp = _mm256_fmadd_ps(cp, a2, set1(-2.75573192239858907e-7f));
    cp = _mm256_fmadd_ps(cp, a2, set1(2.48015873015873016e-5f));
    cp = _mm256_fmadd_ps(cp, a2, set1(-1.38888888888888889e-3f));
    cp = _mm256_fmadd_ps(cp, a2, set1(4.16666666666666667e-2f));
    cp = _mm256_fmadd_ps(cp, a2, set1(-0.5f));
    const F8 ca = _mm256_fmadd_ps(cp, a2, set1(1.0f));
    const I8 qi = _mm256_and_si256(_mm256_cvttps_epi32(q), _mm256_set1_epi32(3));
    const F8 q0 = _mm256_castsi256_ps(_mm256_cmpeq_epi32(qi, _mm256_setzero_si256())), q1 = _mm256_castsi256_ps(_mm256_cmpeq_epi32(qi, _mm256_set1_epi32(1))),
             q2 = _mm256_castsi256_ps(_mm256_cmpeq_epi32(qi, _mm256_set1_epi32(2)));
    s = sel(q0, sa, sel(q1, ca, sel(q2, neg(sa), neg(ca))));
    c = sel(q0, ca, sel(q1, neg(sa), sel(q2, neg(ca), sa)));
}

// sampleMaterial for eight hits; m = material index per lane (0..3).  Returns the lanes whose path goes on.
inline F8 sampleMaterial8(const Materials& mat, I8 m, Vec8 n, Vec8 wo, F8 r1, F8 r2, F8 r3, Vec8& wi, F8 colour[3]) {
    const F8 zero = _mm256_setzero_ps(), one = set1(1.0f);
    const F8 d0 = dot(n, wo);
    const F8 cosi = sel(_mm256_cmp_ps(d0, zero, _CMP_GT_OQ), d0, zero);
    const Vec8 refl = n * _mm256_mul_ps(set1(2.0f), cosi) - wo;
    const F8 etaTab = _mm256_setr_ps(mat.eta[0], mat.eta[1], mat.eta[2], mat.eta[3], 0.f, 0.f, 0.f, 0.f);
    const F8 e = _mm256_permutevar8x32_ps(etaTab, m);
    const F8 k = _mm256_add_ps(one, _mm256_mul_ps(_mm256_mul_ps(e, e), _mm256_sub_ps(_mm256_mul_ps(cosi, cosi), one)));
    const F8 cost = _mm256_sqrt_ps(k);                                    // (NaN where k < 0: those lanes keep fresnel = 1)
    const F8 ec = _mm256_mul_ps(e, cosi), et = _mm256_mul_ps(e, cost);
    const F8 rper = _mm256_div_ps(_mm256_sub_ps(ec, cost), _mm256_add_ps(ec, cost));
    const F8 rpar = neg(_mm256_div_ps(_mm256_sub_ps(et, cosi), _mm256_add_ps(et, cosi)));
    const F8 fr = _mm256_mul_ps(set1(0.5f), _mm256_add_ps(_mm256_mul_ps(rpar, rpar), _mm256_mul_ps(rper, rper)));
    const F8 fresnel = sel(_mm256_cmp_ps(k, zero, _CMP_GE_OQ), fr, one);
    const F8 wide = _mm256_cmp_ps(_mm256_andnot_ps(set1(-0.0f), n.x), set1(0.1f), _CMP_GT_OQ);      // fabsf(n.x) > 0.1f
    Vec8 bu = sel(wide, Vec8{neg(n.z), zero, n.x}, Vec8{zero, neg(n.z), n.y});
    bu = normalize(bu);
    const Vec8 bv = cross(n, bu);
    F8 sn, cs;
    sincos2pi8(r1, sn, cs);
    const F8 s = _mm256_sqrt_ps(r2), c = _mm256_sqrt_ps(_mm256_sub_ps(one, r2));
    const Vec8 diffuse = normalize(n * c + (bu * cs + bv * sn) * s);
    F8 kd[3];
    for (int ch = 0; ch < 3; ++ch) kd[ch] = _mm256_permutevar8x32_ps(_mm256_setr_ps(mat.kd[0][ch], mat.kd[1][ch], mat.kd[2][ch], mat.kd[3][ch], 0.f, 0.f, 0.f, 0.f), m);
    const F8 s0 = _mm256_mul_ps(set1(3.0f), fresnel), s1 = _mm256_add_ps(_mm256_add_ps(kd[0], kd[1]), kd[2]), sum = _mm256_add_ps(s0, s1);
    const F8 pickDiffuse = _mm256_cmp_ps(_mm256_mul_ps(r3, sum), s0, _CMP_GE_OQ);
    wi = sel(pickDiffuse, diffuse, refl);
    F8 rgb[3];
    for (int ch = 0; ch < 3; ++ch) rgb[ch] = sel(pickDiffuse, kd[ch], fresnel);
    const F8 denom = _mm256_add_ps(_mm256_add_ps(rgb[0], rgb[1]), rgb[2]);
    const F8 alive = _mm256_cmp_ps(denom, zero, _CMP_GT_OQ);
    const F8 scale = _mm256_div_ps(sum, denom);
    for (int ch = 0; ch < 3; ++ch) colour[ch] = _mm256_mul_ps(rgb[ch], scale);
    return alive;
}

// In-register transpose of eight rows of eight floats.
inline void transpose8(F8 r[8]) {
    const F8 t0 = _mm256_unpacklo_ps(r[0], r[1]), t1 = _mm256_unpackhi_ps(r[0], r[1]), t2 = _mm256_unpacklo_ps(r[2], r[3]), t3 = _mm256_unpackhi_ps(r[2], r[3]);
    const F8 t4 = _mm256_unpacklo_ps(r[4], r[5]), t5 = _mm256_unpackhi_ps(r[4], r[5]), t6 = _mm256_unpacklo_ps(r[6], r[7]), t7 = _mm256_unpackhi_ps(r[6], r[7]);
    const F8 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE), u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    const F8 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE), u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
    r[0] = _mm256_permute2f128_ps(u0, u4, 0x20); r[1] = _mm256_permute2f128_ps(u1, u5, 0x20); r[2] = _mm256_permute2f128_ps(u2, u6, 0x20); r[3] = _mm256_permute2f128_ps(u3, u7, 0x20);
    r[4] = _mm256_permute2f128_ps(u0, u4, 0x31); r[5] = _mm256_permute2f128_ps(u1, u5, 0x31); r[6] = _mm256_permute2f128_ps(u2, u6, 0x31); r[7] = _mm256_permute2f128_ps(u3, u7, 0x31);
}

// shadeSurface for the eight hits ray[k], hit[k], path[k], sample[k], tri[k] (k = 0..7; the caller has checked depth < maxDepth and the
// triangle index, as for shadeSurface).  Bit k of the result says path k goes on; its next ray / payload are nextRay[k] / nextPath[k]
// (all eight are written; those of dead lanes are garbage).
inline unsigned shadeSurface8(const Materials& mat, const RayRec* const ray[8], const HitRec* const hit[8], const PathRec* const path[8], const uint32_t sample[8],
                              const ShadeTri* const tri[8], RayRec nextRay[8], PathRec nextPath[8]) {
    F8 R[8], A[8], B[8];
    for (int k = 0; k < 8; ++k) {
        R[k] = _mm256_loadu_ps(ray[k]->origin);                                                        // ox oy oz minT dx dy dz maxT
        A[k] = _mm256_load_ps(reinterpret_cast<const float*>(tri[k]));                                 // n0.xyz n1.xyz n2.xy
        B[k] = _mm256_load_ps(reinterpret_cast<const float*>(tri[k]) + 8);                             // n2.z ng.xyz material pad
    }
    transpose8(R); transpose8(A); transpose8(B);
    alignas(32) float hu[8], hv[8], ht[8], w0[8], w1[8], w2[8];
    alignas(32) uint32_t pd[8], sm[8];
    for (int k = 0; k < 8; ++k) {
        ht[k] = hit[k]->t; hu[k] = hit[k]->u; hv[k] = hit[k]->v;
        w0[k] = path[k]->weight[0]; w1[k] = path[k]->weight[1]; w2[k] = path[k]->weight[2]; pd[k] = path[k]->pixelDepth; sm[k] = sample[k];
    }
    const F8 zero = _mm256_setzero_ps(), one = set1(1.0f), minus1 = set1(-1.0f);
    const I8 pixelDepth = _mm256_load_si256(reinterpret_cast<const I8*>(pd));
    const I8 pixel = _mm256_and_si256(pixelDepth, _mm256_set1_epi32(0xFFFFFF)), depth = _mm256_srli_epi32(pixelDepth, 24);
    const F8 u = _mm256_load_ps(hu), v = _mm256_load_ps(hv), t = _mm256_load_ps(ht);
    const F8 w = _mm256_sub_ps(_mm256_sub_ps(one, u), v);
    const Vec8 n0{A[0], A[1], A[2]}, n1{A[3], A[4], A[5]}, n2{A[6], A[7], B[0]};
    Vec8 ng{B[1], B[2], B[3]};
    const I8 m = _mm256_castps_si256(B[4]);
    auto mix = [&](F8 a, F8 b, F8 c) { return _mm256_add_ps(_mm256_add_ps(_mm256_mul_ps(a, w), _mm256_mul_ps(b, u)), _mm256_mul_ps(c, v)); };
    Vec8 n = normalize(Vec8{mix(n0.x, n1.x, n2.x), mix(n0.y, n1.y, n2.y), mix(n0.z, n1.z, n2.z)});
    const Vec8 o{R[0], R[1], R[2]}, d{R[4], R[5], R[6]};
    const Vec8 wo = d * minus1;
    ng = sel(_mm256_cmp_ps(dot(ng, wo), zero, _CMP_LT_OQ), ng * minus1, ng);
    n = sel(_mm256_cmp_ps(dot(n, wo), zero, _CMP_LT_OQ), n * minus1, n);
    const I8 key = pathKey8(pixel, _mm256_load_si256(reinterpret_cast<const I8*>(sm)));
    const I8 depth1 = _mm256_add_epi32(depth, _mm256_set1_epi32(1));
    Vec8 wi;
    F8 colour[3];
    F8 alive = sampleMaterial8(mat, m, n, wo, uniformKeyed8(key, depth1, 3), uniformKeyed8(key, depth1, 4), uniformKeyed8(key, depth1, 5), wi, colour);
    const F8 g0 = _mm256_mul_ps(_mm256_load_ps(w0), colour[0]), g1 = _mm256_mul_ps(_mm256_load_ps(w1), colour[1]), g2 = _mm256_mul_ps(_mm256_load_ps(w2), colour[2]);
    const F8 thr = set1(0.01f);
    alive = _mm256_and_ps(alive, _mm256_or_ps(_mm256_or_ps(_mm256_cmp_ps(g0, thr, _CMP_GT_OQ), _mm256_cmp_ps(g1, thr, _CMP_GT_OQ)), _mm256_cmp_ps(g2, thr, _CMP_GT_OQ)));
    alive = _mm256_and_ps(alive, _mm256_cmp_ps(dot(wi, ng), zero, _CMP_GT_OQ));
    const Vec8 p = o + d * t + ng * set1(1e-4f);
    auto finite = [](F8 x) { return _mm256_cmp_ps(_mm256_sub_ps(x, x), _mm256_setzero_ps(), _CMP_EQ_OQ); };
    alive = _mm256_and_ps(alive, _mm256_and_ps(finite(_mm256_add_ps(_mm256_add_ps(p.x, p.y), p.z)), finite(_mm256_add_ps(_mm256_add_ps(wi.x, wi.y), wi.z))));
    F8 O[8] = {p.x, p.y, p.z, set1(1e-3f), wi.x, wi.y, wi.z, set1(1e6f)};
    transpose8(O);
    for (int k = 0; k < 8; ++k) _mm256_storeu_ps(nextRay[k].origin, O[k]);
    alignas(32) float o0[8], o1[8], o2[8];
    alignas(32) uint32_t opd[8];
    _mm256_store_ps(o0, g0); _mm256_store_ps(o1, g1); _mm256_store_ps(o2, g2);
    _mm256_store_si256(reinterpret_cast<I8*>(opd), _mm256_or_si256(pixel, _mm256_slli_epi32(depth1, 24)));
    for (int k = 0; k < 8; ++k) { nextPath[k].weight[0] = o0[k]; nextPath[k].weight[1] = o1[k]; nextPath[k].weight[2] = o2[k]; nextPath[k].pixelDepth = opd[k]; }
    return unsigned(_mm256_movemask_ps(alive));
}

// primaryRay for the eight pixels (x .. x+7, y), pixel indices pixel .. pixel+7, of one sample: lane k = primaryRay(cam, x + k, y, pixel + k, sample).
inline void primaryRay8(const Camera& cam, uint32_t x, uint32_t y, uint32_t pixel, uint32_t sample, RayRec ray[8], PathRec path[8]) {
    const I8 lane = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const I8 key = pathKey8(_mm256_add_epi32(_mm256_set1_epi32(int(pixel)), lane), _mm256_set1_epi32(int(sample)));
    const I8 zero = _mm256_setzero_si256();
    const F8 px = _mm256_add_ps(_mm256_cvtepi32_ps(_mm256_add_epi32(_mm256_set1_epi32(int(x)), lane)), uniformKeyed8(key, zero, 1));
    const F8 py = _mm256_add_ps(set1(float(y)), uniformKeyed8(key, zero, 2));
    const Vec8 view{set1(cam.view.x), set1(cam.view.y), set1(cam.view.z)}, right{set1(cam.right.x), set1(cam.right.y), set1(cam.right.z)}, up{set1(cam.up.x), set1(cam.up.y), set1(cam.up.z)};
    const Vec8 d = normalize(view + right * px + up * py);
    F8 O[8] = {set1(cam.origin.x), set1(cam.origin.y), set1(cam.origin.z), _mm256_setzero_ps(), d.x, d.y, d.z, set1(1e6f)};
    transpose8(O);
    for (int k = 0; k < 8; ++k) {
        _mm256_storeu_ps(ray[k].origin, O[k]);
        path[k].weight[0] = path[k].weight[1] = path[k].weight[2] = 1.0f;
        path[k].pixelDepth = pixel + uint32_t(k);
    }
}

}  // namespace simd
#define RACC_PT_SHADE8 1
#endif

}  // namespace ptshade
#endif
