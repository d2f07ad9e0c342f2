// Kernels of the gfx950 path tracer, compiled once per arithmetic mode (see
// caster_device.hpp).  The including .hip file defines CASTER_STRICT and
// CASTER_SFX (strict / fast).
#pragma once

#include "caster_device.hpp"
#include "caster_launch.h"

#define CASTER_CAT2(a, b) a##b
#define CASTER_CAT(a, b) CASTER_CAT2(a, b)
#define CASTER_NAME(base) CASTER_CAT(base, CASTER_SFX)

namespace caster {

static_assert(sizeof(MatRec) == sizeof(Mat), "material record layout");
static_assert(sizeof(ImgRec) == sizeof(Img), "image record layout");
static_assert(sizeof(CamRec) == sizeof(Cam), "camera record layout");

CDEV SceneView view_of(const SceneDev& s)
{
    SceneView v;
    v.sph = s.sph; v.sph_meta = s.sph_meta;
    v.rect_a = s.rect_a; v.rect_b = s.rect_b; v.rect_meta = s.rect_meta;
    v.mats = reinterpret_cast<const Mat*>(s.mats);
    v.imgs = reinterpret_cast<const Img*>(s.imgs);
    v.n_sph = s.n_sph; v.n_rect = s.n_rect; v.n_mat = s.n_mat; v.n_img = s.n_img;
    v.has_noise = false; v.noise_vec = nullptr; v.noise_perm = nullptr;      // see stage_noise
    v.filter_img = s.filter_img[CASTER_STRICT];
    v.mfma_tab = s.mfma_tab[CASTER_STRICT];
    return v;
}

// Copies the scene's Perlin tables (if it has any) into the kernel's LDS block `lds` of
// kNoiseLdsBytes and points the view at the copy.  Every thread of the block must call it; a
// __syncthreads() has to follow before the first texture evaluation.
CDEV void stage_noise(SceneView& v, const SceneDev& s, unsigned char* lds)
{
    if (!s.noise_vec) return;
    float4* vec = reinterpret_cast<float4*>(lds);
    unsigned char* perm = lds + 256 * sizeof(float4);
    for (int k = threadIdx.x; k < 256; k += blockDim.x) vec[k] = s.noise_vec[k];
    for (int k = threadIdx.x; k < 768; k += blockDim.x) perm[k] = s.noise_perm[k];
    v.has_noise = true;
    v.noise_vec = (lds_float4_ptr)vec;
    v.noise_perm = (lds_u8_ptr)perm;
}

CDEV Cam cam_of(const CamRec& c)
{
    Cam k;
    k.o = mk(c.o[0], c.o[1], c.o[2]);
    k.llc = mk(c.llc[0], c.llc[1], c.llc[2]);
    k.h = mk(c.h[0], c.h[1], c.h[2]);
    k.v = mk(c.v[0], c.v[1], c.v[2]);
    k.u0 = mk(c.u0[0], c.u0[1], c.u0[2]);
    k.v0 = mk(c.v0[0], c.v0[1], c.v0[2]);
    k.lr = c.lr;
    return k;
}

}  // namespace caster

#include "caster_render.hpp"

namespace caster {

// Closest hit of the rect list, merged into the result of the sphere loop.
// best_ord is the list position of the current winner; an equal t goes to the
// later list entry, which is what the reference's in-order loop produces.
CDEV void closest_rects(const SceneView& sc, const Ray& r, float& closest, int& best_ord,
                        int& best_rect)
{
    for (int k = 0; k < sc.n_rect; ++k) {
        float t;
        if (rect_root(sc.rect_a[k], sc.rect_b[k], r, kTMin, closest, t)) {
            int ord = sc.rect_meta[k].x;
            if (t < closest || ord > best_ord || best_ord < 0) {
                closest = t; best_ord = ord; best_rect = k;
            }
        }
    }
}

// ------------------------------------------------------------ test kernels --
__global__ void CASTER_NAME(k_test_hit_sphere_)(int n, const float* sph4, const float* ray6, const float* tr2,
                                                int32_t* hit, float* t, float* p3, float* n3, int32_t* face)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Ray r;
    r.o = mk(ray6[6 * k], ray6[6 * k + 1], ray6[6 * k + 2]);
    r.d = mk(ray6[6 * k + 3], ray6[6 * k + 4], ray6[6 * k + 5]);
    float radius = sph4[4 * k + 3];
    float4 s = make_float4(sph4[4 * k], sph4[4 * k + 1], sph4[4 * k + 2], radius * radius);
    float root;
    bool ok = sphere_root(s, r.o, r.d, dot(r.d, r.d), tr2[2 * k], tr2[2 * k + 1], root);
    HitRec rec;
    rec.t = 0; rec.f = 0; rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0);
    if (ok) finish_sphere(rec, r, root, mk(s.x, s.y, s.z), radius, 0);
    hit[k] = ok; t[k] = rec.t; face[k] = rec.f;
    p3[3 * k] = rec.p.x; p3[3 * k + 1] = rec.p.y; p3[3 * k + 2] = rec.p.z;
    n3[3 * k] = rec.n.x; n3[3 * k + 1] = rec.n.y; n3[3 * k + 2] = rec.n.z;
}

__global__ void CASTER_NAME(k_test_hit_rect_)(int n, const float* rect5, const float* ray6, const float* tr2,
                                              int32_t* hit, float* t, float* p3, float* n3, int32_t* face,
                                              float* uv2)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Ray r;
    r.o = mk(ray6[6 * k], ray6[6 * k + 1], ray6[6 * k + 2]);
    r.d = mk(ray6[6 * k + 3], ray6[6 * k + 4], ray6[6 * k + 5]);
    float4 ra = make_float4(rect5[5 * k], rect5[5 * k + 1], rect5[5 * k + 2], rect5[5 * k + 3]);
    float4 rb = make_float4(rect5[5 * k + 4], 0, 0, 0);
    float tt;
    bool ok = rect_root(ra, rb, r, tr2[2 * k], tr2[2 * k + 1], tt);
    HitRec rec;
    rec.t = 0; rec.f = 0; rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.uv.u = 0; rec.uv.v = 0;
    if (ok) finish_rect(rec, r, tt, ra, rb, 0);
    hit[k] = ok; t[k] = rec.t; face[k] = rec.f;
    p3[3 * k] = rec.p.x; p3[3 * k + 1] = rec.p.y; p3[3 * k + 2] = rec.p.z;
    n3[3 * k] = rec.n.x; n3[3 * k + 1] = rec.n.y; n3[3 * k + 2] = rec.n.z;
    uv2[2 * k] = rec.uv.u; uv2[2 * k + 1] = rec.uv.v;
}

// hit_list_hit_all straight from global memory (no LDS staging): the same
// device functions as the render kernel, one ray per thread.
CDEV bool closest_hit_global(const SceneView& sc, const Ray& r, HitRec& rec)
{
    float closest = kTMax;
    int best_sph = -1;
    const float aa = dot(r.d, r.d);
    for (int k = 0; k < sc.n_sph; ++k) {
        float4 s = sc.sph[k];
        s.w = s.w * s.w;
        float root;
        if (sphere_root(s, r.o, r.d, aa, kTMin, closest, root)) { closest = root; best_sph = k; }
    }
    int best_ord = (best_sph >= 0) ? sc.sph_meta[best_sph].x : -1;
    int best_rect = -1;
    closest_rects(sc, r, closest, best_ord, best_rect);
    if (best_ord < 0) return false;
    if (best_rect >= 0) {
        finish_rect(rec, r, closest, sc.rect_a[best_rect], sc.rect_b[best_rect], sc.rect_meta[best_rect].y);
    } else {
        const float4 sp = sc.sph[best_sph];
        finish_sphere(rec, r, closest, mk(sp.x, sp.y, sp.z), sp.w, sc.sph_meta[best_sph].y);
    }
    return true;
}

__global__ void CASTER_NAME(k_test_hit_list_)(SceneDev sd, int n, const float* ray6, int32_t* hit, float* t,
                                              float* p3, float* n3, int32_t* face, int32_t* mat)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const SceneView sc = view_of(sd);
    Ray r;
    r.o = mk(ray6[6 * k], ray6[6 * k + 1], ray6[6 * k + 2]);
    r.d = mk(ray6[6 * k + 3], ray6[6 * k + 4], ray6[6 * k + 5]);
    HitRec rec;
    rec.t = 0; rec.f = 0; rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.m = -1;
    bool ok = closest_hit_global(sc, r, rec);
    hit[k] = ok; t[k] = rec.t; face[k] = rec.f; mat[k] = ok ? rec.m : -1;
    p3[3 * k] = rec.p.x; p3[3 * k + 1] = rec.p.y; p3[3 * k + 2] = rec.p.z;
    n3[3 * k] = rec.n.x; n3[3 * k + 1] = rec.n.y; n3[3 * k + 2] = rec.n.z;
}

__global__ void CASTER_NAME(k_test_scatter_)(SceneDev sd, int n, const float* ray6, const float* p3,
                                             const float* n3, const int32_t* face, const int32_t* mat,
                                             uint32_t seed, int32_t* ok, float* out6, int32_t* draws)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const SceneView sc = view_of(sd);
    Ray r;
    r.o = mk(ray6[6 * k], ray6[6 * k + 1], ray6[6 * k + 2]);
    r.d = mk(ray6[6 * k + 3], ray6[6 * k + 4], ray6[6 * k + 5]);
    HitRec rec;
    rec.p = mk(p3[3 * k], p3[3 * k + 1], p3[3 * k + 2]);
    rec.n = mk(n3[3 * k], n3[3 * k + 1], n3[3 * k + 2]);
    rec.f = face[k];
    rec.m = mat[k];
    Rng g;
    rng_seed(g, seed, (uint32_t)k, 0u);
    Ray out;
    out.o = mk(0, 0, 0); out.d = mk(0, 0, 0);
    bool res = false;
    const Mat m = sc.mats[rec.m];
    if (m.type == 0) res = scatter_lambertian(g, rec, out);
    else if (m.type == 1) res = scatter_metal(g, r, rec, m.fuzz, out);
    else if (m.type == 2) res = scatter_dielectric(g, r, rec, m.ior, out);
    ok[k] = res;
    out6[6 * k] = out.o.x; out6[6 * k + 1] = out.o.y; out6[6 * k + 2] = out.o.z;
    out6[6 * k + 3] = out.d.x; out6[6 * k + 4] = out.d.y; out6[6 * k + 5] = out.d.z;
    draws[k] = (int32_t)g.draws;
}

__global__ void CASTER_NAME(k_test_camera_)(CamRec cr, int n, const float* uv2, uint32_t seed, float* out6,
                                            int32_t* draws)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const Cam cam = cam_of(cr);
    Rng g;
    rng_seed(g, seed, (uint32_t)k, 0u);
    Ray r = ray_from_cam(cam, g, uv2[2 * k], uv2[2 * k + 1]);
    out6[6 * k] = r.o.x; out6[6 * k + 1] = r.o.y; out6[6 * k + 2] = r.o.z;
    out6[6 * k + 3] = r.d.x; out6[6 * k + 4] = r.d.y; out6[6 * k + 5] = r.d.z;
    draws[k] = (int32_t)g.draws;
}

__global__ void CASTER_NAME(k_test_texture_)(SceneDev sd, int n, const int32_t* mat, const float* uv2,
                                             const float* p3, float* rgb3)
{
    __shared__ __align__(16) unsigned char s_noise[kNoiseLdsBytes];
    SceneView sc = view_of(sd);
    stage_noise(sc, sd, s_noise);
    __syncthreads();
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    HitRec rec;
    rec.p = mk(p3[3 * k], p3[3 * k + 1], p3[3 * k + 2]);
    rec.uv.u = uv2[2 * k]; rec.uv.v = uv2[2 * k + 1];
    rec.uv_sphere = false;
    V3 c = texture_color(sc, sc.mats[mat[k]], rec);
    rgb3[3 * k] = c.x; rgb3[3 * k + 1] = c.y; rgb3[3 * k + 2] = c.z;
}

// The small vector and optics functions of src/common.c and get_uv_coord of src/material.c, one
// item per thread: in8 holds up to eight floats of arguments, out3 up to three of results.
//   fn 0 reflectance(cosine, ratio)                 common.c:170-175   -> out[0]
//   fn 1 v3_f32_refract(unit_v, n, ratio), cos_theta as material.c:30 hands it in (in[7])   common.c:163-168 -> out[0..2]
//   fn 2 v3_f32_reflect(v, n)                        common.c:158-161   -> out[0..2]
//   fn 3 get_uv_coord of a point on the unit sphere  material.c:45-56   -> out[0..1]
//   fn 4 v3_f32_near_zero(v)                         common.c:152-156   -> out[0] = 0 or 1
//   fn 5 v3_f32_to_unit(v)                           common.c:89-97     -> out[0..2]
__global__ void CASTER_NAME(k_test_math_)(int fn, int n, const float* in8, float* out3)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* a = in8 + 8 * (size_t)k;
    V3 r = mk(0.0f, 0.0f, 0.0f);
    switch (fn) {
    case 0: r.x = reflectance(a[0], a[1]); break;
    case 1: r = refract(mk(a[0], a[1], a[2]), mk(a[3], a[4], a[5]), a[6], a[7]); break;
    case 2: r = reflect(mk(a[0], a[1], a[2]), mk(a[3], a[4], a[5])); break;
    case 3: { const V2 uv = get_uv_coord_sphere(mk(a[0], a[1], a[2])); r.x = uv.u; r.y = uv.v; break; }
    case 4: r.x = near_zero(mk(a[0], a[1], a[2])) ? 1.0f : 0.0f; break;
    case 5: r = to_unit(mk(a[0], a[1], a[2])); break;
    default: break;
    }
    out3[3 * k] = r.x; out3[3 * k + 1] = r.y; out3[3 * k + 2] = r.z;
}

#if defined(CASTER_STATS) || defined(CASTER_TIMERS)
extern "C" void CASTER_NAME(caster_debug_stats_)(unsigned long long* out32, int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_caster_stats), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_caster_stats), z, sizeof z);
    }
}
#endif
#ifdef CASTER_TIMERS
extern "C" void CASTER_NAME(caster_debug_wavelog_)(unsigned long long* out65536)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out65536, HIP_SYMBOL(g_caster_wavelog), sizeof(unsigned long long) * 65536);
}
#endif

static inline dim3 grid_for(int n) { return dim3((unsigned)((n + 255) / 256)); }

hipError_t CASTER_NAME(launch_test_hit_sphere_)(int n, const float* sph4, const float* ray6, const float* tr2,
                                                int32_t* hit, float* t, float* p3, float* n3, int32_t* face,
                                                hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_hit_sphere_), grid_for(n), dim3(256), 0, st, n, sph4, ray6, tr2, hit,
                       t, p3, n3, face);
    return hipGetLastError();
}
hipError_t CASTER_NAME(launch_test_hit_rect_)(int n, const float* rect5, const float* ray6, const float* tr2,
                                              int32_t* hit, float* t, float* p3, float* n3, int32_t* face,
                                              float* uv2, hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_hit_rect_), grid_for(n), dim3(256), 0, st, n, rect5, ray6, tr2, hit,
                       t, p3, n3, face, uv2);
    return hipGetLastError();
}
hipError_t CASTER_NAME(launch_test_hit_list_)(const SceneDev& sc, int n, const float* ray6, int32_t* hit,
                                              float* t, float* p3, float* n3, int32_t* face, int32_t* mat,
                                              hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_hit_list_), grid_for(n), dim3(256), 0, st, sc, n, ray6, hit, t, p3,
                       n3, face, mat);
    return hipGetLastError();
}
hipError_t CASTER_NAME(launch_test_scatter_)(const SceneDev& sc, int n, const float* ray6, const float* p3,
                                             const float* n3, const int32_t* face, const int32_t* mat,
                                             uint32_t seed, int32_t* ok, float* out6, int32_t* draws,
                                             hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_scatter_), grid_for(n), dim3(256), 0, st, sc, n, ray6, p3, n3, face,
                       mat, seed, ok, out6, draws);
    return hipGetLastError();
}
hipError_t CASTER_NAME(launch_test_camera_)(const CamRec& cam, int n, const float* uv2, uint32_t seed,
                                            float* out6, int32_t* draws, hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_camera_), grid_for(n), dim3(256), 0, st, cam, n, uv2, seed, out6,
                       draws);
    return hipGetLastError();
}
hipError_t CASTER_NAME(launch_test_texture_)(const SceneDev& sc, int n, const int32_t* mat, const float* uv2,
                                             const float* p3, float* rgb3, hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_texture_), grid_for(n), dim3(256), 0, st, sc, n, mat, uv2, p3, rgb3);
    return hipGetLastError();
}
hipError_t CASTER_NAME(launch_test_math_)(int fn, int n, const float* in8, float* out3, hipStream_t st)
{
    hipLaunchKernelGGL(CASTER_NAME(k_test_math_), grid_for(n), dim3(256), 0, st, fn, n, in8, out3);
    return hipGetLastError();
}

}  // namespace caster
