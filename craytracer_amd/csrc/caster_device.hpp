// Device-side math, random stream, intersection, scattering and texturing for
// the gfx950 path tracer.  This header is compiled twice:
//   * caster_strict.hip  (CASTER_STRICT=1, -ffp-contract=off): every expression
//     keeps the reference's association order, so no FMA is formed, divide and
//     sqrt are the correctly rounded forms, and libm calls are evaluated in
//     double and rounded once;
//   * caster_fast.hip    (CASTER_STRICT=0): the compiler may contract to FMA.
// Each function cites the reference lines whose result it must reproduce.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef CASTER_STRICT
#error "define CASTER_STRICT to 0 or 1 before including caster_device.hpp"
#endif

#define CDEV __device__ __forceinline__

// Developer builds, off in the product: -DCASTER_STATS counts wave-level events, -DCASTER_TIMERS sums the
// shader clock ticks the waves spend in regions of the render kernel (slot 16+i).
#if defined(CASTER_STATS) || defined(CASTER_TIMERS)
__device__ unsigned long long g_caster_stats[32];
#endif
#ifdef CASTER_STATS
#define CASTER_STAT(i)                                                                                   \
    do {                                                                                                 \
        const unsigned long long ex_ = __builtin_amdgcn_read_exec();                                     \
        if (__builtin_amdgcn_mbcnt_hi((unsigned)(ex_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ex_, 0u)) == 0) \
            atomicAdd(&g_caster_stats[i], 1ull);                                                         \
    } while (0)
#else
#define CASTER_STAT(i) do { } while (0)
#endif
// Region timers of the render kernel.  In the product build every member is an empty inline
// function and the object holds nothing.
#ifdef CASTER_TIMERS
__device__ unsigned long long g_caster_wavelog[4096 * 16];  // one record per resident wave, see WaveTimers::report
struct WaveTimers {
    enum Region { kSchedule = 0, kBounce, kRefill, kIntersect, kTotal, kLists, kRegions };
    unsigned long long ticks[kRegions] = {0, 0, 0, 0, 0, 0};   // shader clock ticks per region
    unsigned long long tail[kRegions] = {0, 0, 0, 0, 0, 0};    // the same after mark()
    unsigned long long trips = 0, units = 0;
    unsigned long long wall0 = 0, wall_last_fetch = 0, wall_empty = 0;  // 100 MHz
    unsigned long long last_unit = 0, trips_at_empty = 0;
    __device__ __forceinline__ unsigned long long now() const { return __builtin_readcyclecounter(); }
    __device__ __forceinline__ void start() { wall0 = wall_clock64(); }
    __device__ __forceinline__ void add(Region r, unsigned long long t0) { const unsigned long long d = now() - t0; ticks[r] += d; if (marked) tail[r] += d; }
    __device__ __forceinline__ void trip() { ++trips; }
    unsigned long long marked = 0, trips_at_mark = 0;
    __device__ __forceinline__ void mark() { marked = 1; trips_at_mark = trips; }       // from here on the regions also add to `tail`
    __device__ __forceinline__ void fetched(bool got, int chunk, int tile_x, int tile_y)
    {
        if (got) { ++units; wall_last_fetch = wall_clock64(); last_unit = (unsigned long long)chunk * 1000000ull + tile_y * 1000 + tile_x; }
        else { wall_empty = wall_clock64(); trips_at_empty = trips; }
    }
    // sums into g_caster_stats[16..22] (+ trips into [0] when the event counters are off) and one record
    // per wave: start, end, units, trips, last fetch, last unit, queue-empty time, trips then,
    // bounce, refill, intersect, lists, total ticks, rays | trips at mark() << 32, and after mark():
    // intersect | lists << 32, bounce | refill << 32
    __device__ __forceinline__ void report(int wave_in_block, unsigned int wave_rays)
    {
        if ((threadIdx.x & 63) != 0) return;
        for (int r = 0; r < kRegions; ++r) atomicAdd(&g_caster_stats[16 + r], ticks[r]);
        const unsigned long long wall1 = wall_clock64();
        atomicAdd(&g_caster_stats[16 + kRegions], wall1 - wall0);
#ifndef CASTER_STATS
        atomicAdd(&g_caster_stats[0], trips);
#endif
        if (blockIdx.x >= 1024) return;
        unsigned long long* wl = g_caster_wavelog + (blockIdx.x * 4 + wave_in_block) * 16;
        wl[0] = wall0; wl[1] = wall1; wl[2] = units; wl[3] = trips;
        wl[4] = wall_last_fetch; wl[5] = last_unit; wl[6] = wall_empty; wl[7] = trips_at_empty;
        wl[8] = ticks[kBounce]; wl[9] = ticks[kRefill]; wl[10] = ticks[kIntersect]; wl[11] = ticks[kLists];
        wl[12] = ticks[kTotal]; wl[13] = wave_rays | (trips_at_mark << 32);
        wl[14] = tail[kIntersect] | (tail[kLists] << 32); wl[15] = tail[kBounce] | (tail[kRefill] << 32);
    }
};
#else
struct WaveTimers {
    enum Region { kSchedule = 0, kBounce, kRefill, kIntersect, kTotal, kLists, kRegions };
    __device__ __forceinline__ unsigned long long now() const { return 0; }
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void add(Region, unsigned long long) {}
    __device__ __forceinline__ void trip() {}
    __device__ __forceinline__ void mark() {}
    __device__ __forceinline__ void fetched(bool, int, int, int) {}
    __device__ __forceinline__ void report(int, unsigned int) {}
};
#endif

namespace caster {

constexpr float kPi = 3.141592654f;      // inc/common.h:10
constexpr float kTMin = 0.001f;          // src/caster.c:441
constexpr float kTMax = 16777215.0f;     // src/caster.c:441 (0xFFFFFF)

struct V3 { float x, y, z; };
struct V2 { float u, v; };
struct Ray { V3 o, d; };

// Flat material / image records as the host uploads them.
struct Mat {
    int32_t type, tex;
    float c0[3], c1[3];
    float fuzz, ior;
    int32_t image, pad;
};
struct Img {
    const uint8_t* data;
    uint32_t w, h, bpp, pad;
};
struct Cam { V3 o, llc, h, v, u0, v0; float lr; };

// Scene as the kernels see it.  Spheres and rects keep their list positions
// (ord) so that a tie in t can go to the later list entry, as the reference's
// in-order loop does (src/caster.c:114,136,204).
struct SceneView {
    const float4* sph;      // cx, cy, cz, r
    const int2*   sph_meta; // list position, material
    const float4* filter_img;
    const uint4* mfma_tab;
    const float4* rect_a;   // cx, cy, cz, du
    const float4* rect_b;   // dv, -, -, -
    const int2*   rect_meta;
    const Mat*    mats;
    const Img*    imgs;
    int n_sph, n_rect, n_mat, n_img;
    // Perlin tables, staged in LDS by every kernel that evaluates textures (stage_noise).  An LDS
    // address may be 0, which is also the null value of that address space: has_noise says
    // whether the tables are there.
    bool has_noise;
    const __attribute__((address_space(3))) float4* noise_vec;
    const __attribute__((address_space(3))) uint8_t* noise_perm;
};

typedef __attribute__((address_space(3))) float4* lds_float4_ptr;
typedef __attribute__((address_space(3))) uint8_t* lds_u8_ptr;
constexpr int kNoiseLdsBytes = 256 * (int)sizeof(float4) + 768;    // Perlin vectors and permutation tables

CDEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
CDEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }   // common.c:45-53
CDEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }   // common.c:55-63
CDEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }   // common.c:65-72
CDEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }      // common.c:33
// common.c:34.  Three correctly rounded quotients over one denominator.  The compiler's division is a
// reciprocal refined once and a chain of four fused corrections, wrapped in a scaling of operands in
// extreme ranges and a fix-up of zeros, infinities and NaNs; when all four magnitudes lie in
// [2^-40, 2^40] neither does anything, and the same chain on one shared reciprocal gives the same
// bits (tools/div_check.hip compares 1e11 quotients).  Anything else takes the plain division.
CDEV float div_chain(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

CDEV V3 operator/(V3 a, float s)
{
#if CASTER_STRICT
    const float as = __builtin_fabsf(s), ax = __builtin_fabsf(a.x), ay = __builtin_fabsf(a.y), az = __builtin_fabsf(a.z);
    const float lo = __builtin_fminf(__builtin_fminf(ax, ay), __builtin_fminf(az, as));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(ax, ay), __builtin_fmaxf(az, as));
    if (lo >= 0x1p-40f && hi <= 0x1p40f) {
        const float r0 = __builtin_amdgcn_rcpf(s);
        const float r = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
        return mk(div_chain(a.x, s, r), div_chain(a.y, s, r), div_chain(a.z, s, r));
    }
#endif
    return mk(a.x / s, a.y / s, a.z / s);
}
CDEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }                              // common.c:36-43
CDEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y + a.z * b.z); }      // common.c:84-87
CDEV V3 to_unit(V3 v) { return v / sqrtf(dot(v, v)); }                          // common.c:89-97

// ------------------------------------------------------------------ RNG --
// Replaces rand() (src/common.c:10-13).  One stream per path: the 64-bit xoroshiro engine (two
// 32-bit words; s1 ^= s0, s0 = rotl(s0, 26) ^ s1 ^ (s1 << 9), s1 = rotl(s1, 13)) with the sum of the
// two words as output, seeded by two chained 32-bit avalanche hashes of (seed, pixel, sample).  A
// draw is the top 23 bits of the output, random_f32 = n / 2^23: exact in binary32, never 1.  A draw
// costs six full-rate integer operations and two to make the float, and no multiply -- which
// matters because the rejection samplers make the wave pay for its slowest lane -- and a path
// carries two state words.
struct Rng { uint32_t s0, s1; uint32_t draws; };

CDEV uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

CDEV void rng_seed(Rng& g, uint32_t seed, uint32_t pixel, uint32_t sample)
{
    const uint32_t a = mix32(pixel ^ mix32(seed + 0x9E3779B9u));
    const uint32_t b = mix32(sample + a);
    g.s0 = a; g.s1 = b | 1u;            // never the all-zero state
    g.draws = 0;
}

CDEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }

// One step; returns the 32-bit output.
CDEV uint32_t rng_step(Rng& g)
{
    const uint32_t result = g.s0 + g.s1;
    const uint32_t x = g.s1 ^ g.s0;
    g.s0 = xor3(__builtin_amdgcn_alignbit(g.s0, g.s0, 6u), x, x << 9);     // rotl(s0, 26) ^ x ^ (x << 9)
    g.s1 = __builtin_amdgcn_alignbit(x, x, 19u);                           // rotl(x, 13)
    return result;
}

// The integer of a draw: the top 23 bits of the output.
CDEV uint32_t rng_next23(Rng& g) { return rng_step(g) >> 9; }

// 1 + n / 2^23 for the next draw n: the 23 bits dropped into the mantissa of a float in [1, 2)
// (one v_alignbit_b32: {0x7F : output} >> 9).
CDEV float rng_one_plus(Rng& g)
{
    g.draws++;
    return __uint_as_float(__builtin_amdgcn_alignbit(0x7Fu, rng_step(g), 9u));
}

CDEV float random_f32(Rng& g)                                   // common.c:10-13
{
    return rng_one_plus(g) - 1.0f;                              // n / 2^23, exact
}

CDEV float random_range_f32(Rng& g, float lo, float hi)         // common.c:15-18
{
    return lo + (hi - lo) * random_f32(g);
}

// random_range_f32(g, -1, 1) = -1 + 2 * (n / 2^23) (common.c:15-18): every step of it is exact, and
// so is 2 * (1 + n / 2^23) - 3 in one fused operation: the same bits, fewer operations in loops where
// the wave pays for its slowest lane.
CDEV float random_pm1(Rng& g)
{
    return __builtin_fmaf(rng_one_plus(g), 2.0f, -3.0f);
}

CDEV V3 random_unit_sphere(Rng& g)                              // common.c:108-126
{
    for (;;) {
        CASTER_STAT(4);
        V3 p;
        p.x = random_pm1(g);
        p.y = random_pm1(g);
        p.z = random_pm1(g);
        if (dot(p, p) < 1.0f) return p;
    }
}

CDEV V3 random_unit_disk(Rng& g)                                // common.c:128-136, x first
{
    for (;;) {
        CASTER_STAT(5);
        V3 p;
        p.x = random_pm1(g);
        p.y = random_pm1(g);
        p.z = 0.0f;
        if (dot(p, p) < 1.0f) return p;
    }
}

CDEV V3 random_unit_norm(Rng& g) { return to_unit(random_unit_sphere(g)); }     // common.c:138-141

CDEV bool near_zero(V3 v)                                       // common.c:152-156
{
    const float s = 1E-8f;
    return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}

CDEV V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }              // common.c:158-161

CDEV V3 refract(V3 unit_v, V3 n, float ratio, float cos_theta)  // common.c:163-168
{
    V3 orto = (unit_v + n * cos_theta) * ratio;
    V3 para = n * -sqrtf(fabsf(1.0f - dot(orto, orto)));
    return orto + para;
}

// libm calls.  The strict build evaluates them in double and rounds once,
// which agrees with glibc's float results except where glibc itself is not
// correctly rounded (last-ulp cases).
CDEV float m_pow5(float x)
{
#if CASTER_STRICT
    double d = (double)x;
    return (float)(d * d * d * d * d);
#else
    float x2 = x * x;
    return x2 * x2 * x;
#endif
}
CDEV float m_sin(float x)
{
#if CASTER_STRICT
    return (float)sin((double)x);
#else
    return sinf(x);
#endif
}
CDEV float m_atan2(float y, float x)
{
#if CASTER_STRICT
    return (float)atan2((double)y, (double)x);
#else
    return atan2f(y, x);
#endif
}
CDEV float m_acos(float x)
{
#if CASTER_STRICT
    return (float)acos((double)x);
#else
    return acosf(x);
#endif
}

CDEV float reflectance(float cosine, float ratio)               // common.c:170-175
{
    float r0 = (1.0f - ratio) / (1.0f + ratio);
    r0 = r0 * r0;
    return r0 + (1.0f - r0) * m_pow5(1.0f - cosine);
}

// ---------------------------------------------------------- intersection --
CDEV V3 ray_at(const Ray& r, float t) { return r.o + r.d * t; } // caster.c:5-8

// One sphere against one ray inside the closest-hit loop (caster.c:99-121 up
// to the accepted root).  s = {cx, cy, cz, r*r}.  Returns true and the root if
// the sphere is accepted against the closest t so far.
CDEV bool sphere_root(const float4 s, const V3& o, const V3& d, float a, float tmin, float tmax,
                      float& root_out)
{
    V3 oc = mk(o.x - s.x, o.y - s.y, o.z - s.z);
    float hb = dot(oc, d);
    float c = dot(oc, oc) - s.w;
    float disc = hb * hb - a * c;
    if (disc < 0.0f) return false;
    float sq = sqrtf(disc);
    float root = (-hb - sq) / a;
    if (root < tmin || root > tmax) {
        root = (-hb + sq) / a;
        if (root < tmin || root > tmax) return false;
    }
    root_out = root;
    return true;
}

struct HitRec {
    V3 p, n;
    float t;
    int f;
    int m;
    V2 uv;
    bool uv_sphere;   // uv not yet computed: a sphere's uv is derived from the normal on demand
    V3 outn;
};

CDEV void set_face(HitRec& rec, const Ray& r, V3 outn)          // caster.c:10-14
{
    rec.f = dot(r.d, outn) < 0.0f;
    rec.n = rec.f ? outn : neg(outn);
}

// caster.c:123-128 for an accepted root.  The reference also evaluates
// get_uv_coord here for every candidate; only image textures read uv, so it
// is evaluated lazily from outn (same value).
CDEV void finish_sphere(HitRec& rec, const Ray& r, float root, V3 center, float radius, int mat)
{
    rec.t = root;
    rec.p = ray_at(r, root);
    V3 outn = (rec.p - center) / radius;
    set_face(rec, r, outn);
    rec.outn = outn;
    rec.uv_sphere = true;
    rec.uv.u = 0.0f; rec.uv.v = 0.0f;
    rec.m = mat;
}

CDEV V2 get_uv_coord_sphere(V3 p)                               // material.c:45-56
{
    V2 uv;
    uv.u = (m_atan2(-p.z, p.x) + kPi) / (2.0f * kPi);
    uv.v = m_acos(-p.y) / kPi;
    return uv;
}

// caster.c:132-151: candidate t of an xy rect, independent of the closest hit.
CDEV bool rect_root(const float4 ra, const float4 rb, const Ray& r, float tmin, float tmax, float& t_out)
{
    float t = (ra.z - r.o.z) / r.d.z;
    if (t < tmin || t > tmax) return false;
    float x = r.o.x + t * r.d.x;
    float y = r.o.y + t * r.d.y;
    float half_du = ra.w / 2.0f;
    float half_dv = rb.x / 2.0f;
    float x0 = ra.x - half_du, x1 = ra.x + half_du;
    float y0 = ra.y - half_dv, y1 = ra.y + half_dv;
    if (x < x0 || x > x1 || y < y0 || y > y1) return false;
    t_out = t;
    return true;
}

CDEV void finish_rect(HitRec& rec, const Ray& r, float t, const float4 ra, const float4 rb, int mat)
{                                                               // caster.c:139-161
    float x = r.o.x + t * r.d.x;
    float y = r.o.y + t * r.d.y;
    float x0 = ra.x - ra.w / 2.0f;
    float y0 = ra.y - rb.x / 2.0f;
    rec.uv.u = (x - x0) / ra.w;
    rec.uv.v = (y - y0) / rb.x;
    rec.uv_sphere = false;
    rec.t = t;
    set_face(rec, r, mk(0.0f, 0.0f, 1.0f));
    rec.outn = mk(0.0f, 0.0f, 1.0f);
    rec.m = mat;
    rec.p = ray_at(r, t);
}

// -------------------------------------------------------------- materials --
CDEV bool scatter_lambertian(Rng& g, const HitRec& rec, Ray& out)               // material.c:3-12
{
    out.o = rec.p;
    out.d = rec.n + random_unit_norm(g);
    if (near_zero(out.d)) out.d = rec.n;
    return true;
}

CDEV bool scatter_metal(Rng& g, const Ray& r, const HitRec& rec, float fuzz, Ray& out)  // material.c:14-19
{
    out.o = rec.p;
    V3 refl = reflect(to_unit(r.d), rec.n);
    V3 fz = random_unit_sphere(g) * fuzz;
    out.d = refl + fz;
    return dot(out.d, rec.n) > 0.0f;
}

// material.c:21-43 with unit_vector(r->direction) handed in (shade() computes it once for all materials)
CDEV bool scatter_dielectric_unit(Rng& g, V3 udir, const HitRec& rec, float ior, Ray& out)
{
    float rratio = rec.f ? (1.0f / ior) : ior;
    out.o = rec.p;
    V3 iudir = neg(udir);
    float cos_theta = fminf(dot(iudir, rec.n), 1.0f);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    bool not_refract = rratio * sin_theta > 1.0f;
    if (not_refract || reflectance(cos_theta, rratio) > random_f32(g))
        out.d = reflect(udir, rec.n);
    else
        out.d = refract(udir, rec.n, rratio, cos_theta);
    return true;
}

CDEV bool scatter_dielectric(Rng& g, const Ray& r, const HitRec& rec, float ior, Ray& out)  // material.c:21-43
{
    return scatter_dielectric_unit(g, to_unit(r.d), rec, ior, out);
}

CDEV float clampf(float x, float lo, float hi)                  // common.c:25-29
{
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

// Gradient noise and turbulence for TTYPE_NOISE.  No counterpart in the reference (inc/noise.h:4-7
// is a placeholder, src/noise.c is empty): the published construction of "Ray Tracing: The Next
// Week".  Eight lattice corners around p each hold a unit vector chosen by xor-ing three
// permutation tables; the value interpolates dot(vector, offset from the corner) with
// Hermite-smoothed weights.
CDEV float noise_value(const SceneView& sc, V3 p)
{
    const float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    const float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const int i = (int)fx, j = (int)fy, k = (int)fz;
    const float uu = u * u * (3.0f - 2.0f * u);
    const float vv = v * v * (3.0f - 2.0f * v);
    const float ww = w * w * (3.0f - 2.0f * w);
    const __attribute__((address_space(3))) uint8_t* px = sc.noise_perm;
    const __attribute__((address_space(3))) uint8_t* py = px + 256;
    const __attribute__((address_space(3))) uint8_t* pz = px + 512;
    // the six table bytes as 32-bit values, xor-ed three at a time in one instruction each
    const unsigned int hx[2] = {px[i & 255], px[(i + 1) & 255]};
    const unsigned int hy[2] = {py[j & 255], py[(j + 1) & 255]};
    const unsigned int hz[2] = {pz[k & 255], pz[(k + 1) & 255]};
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; di++) {
        const float wi = di ? uu : 1.0f - uu;
#pragma unroll
        for (int dj = 0; dj < 2; dj++) {
            const float wj = dj ? vv : 1.0f - vv;
#pragma unroll
            for (int dk = 0; dk < 2; dk++) {
                const __attribute__((address_space(3))) float4* cp = sc.noise_vec + xor3(hx[di], hy[dj], hz[dk]);
                const float cx = cp->x, cy = cp->y, cz = cp->z;
                const float wk = dk ? ww : 1.0f - ww;
                const float ox = u - (float)di, oy = v - (float)dj, oz = w - (float)dk;
                const float d = (cx * ox + cy * oy) + cz * oz;
                accum += ((wi * wj) * wk) * d;
            }
        }
    }
    return accum;
}

CDEV float noise_turb(const SceneView& sc, V3 p)
{
    float accum = 0.0f, weight = 1.0f;
    V3 temp = p;
#pragma unroll 1
    for (int k = 0; k < 7; k++) {
        accum += weight * noise_value(sc, temp);
        weight *= 0.5f;
        temp = temp * 2.0f;
    }
    return fabsf(accum);
}

// The marble pattern's brightness at p: what TTYPE_NOISE multiplies the material's colour with.
CDEV float marble_scale(const SceneView& sc, V3 p, float stripes)
{
    return 0.5f * (1.0f + m_sin(stripes * p.z + 10.0f * noise_turb(sc, p)));
}

// NOISE false: for callers that evaluate the noise textures elsewhere (shade<TEX, true>) and never
// come here with one that has its tables.
template <bool NOISE = true>
CDEV V3 texture_color(const SceneView& sc, const Mat& m, HitRec& rec)           // material.c:65-101
{
    switch (m.tex) {
    case 0: // TTYPE_COLOR
        return mk(m.c0[0], m.c0[1], m.c0[2]);
    case 1: { // TTYPE_CHECKER
        float sines = m_sin(10.0f * rec.p.x) * m_sin(10.0f * rec.p.y) * m_sin(10.0f * rec.p.z);
        if (sines < 0.0f) return mk(m.c0[0], m.c0[1], m.c0[2]);
        return mk(m.c1[0], m.c1[1], m.c1[2]);
    }
    case 2: { // TTYPE_IMAGE_ALBEDO
        if (m.image < 0 || m.image >= sc.n_img) return mk(0.0f, 0.0f, 0.0f);
        const Img im = sc.imgs[m.image];
        V2 uv = rec.uv_sphere ? get_uv_coord_sphere(rec.outn) : rec.uv;
        uv.u = clampf(uv.u, 0.0f, 1.0f);
        uv.v = clampf(uv.v, 0.0f, 1.0f);
        uint32_t i = (uint32_t)(uv.u * (float)im.w);
        uint32_t j = (uint32_t)(uv.v * (float)im.h);
        if (i >= im.w) i = im.w - 1;
        if (j >= im.h) j = im.h - 1;
        const float color_scale = 1.0f / 255.0f;
        const uint8_t* pix = im.data + (size_t)j * im.w * im.bpp + (size_t)i * im.bpp;
        return mk(color_scale * (float)pix[0], color_scale * (float)pix[1], color_scale * (float)pix[2]);
    }
    case 4: { // TTYPE_NOISE (extension): marble
        if (!NOISE || !sc.has_noise) return mk(0.0f, 0.0f, 0.0f);
        const float s = marble_scale(sc, rec.p, m.c1[0]);
        return mk(m.c0[0] * s, m.c0[1] * s, m.c0[2] * s);
    }
    default:
        return mk(0.0f, 0.0f, 0.0f);
    }
}

// ----------------------------------------------------------------- camera --
CDEV Ray ray_from_cam(const Cam& ci, Rng& g, float u, float v)  // camera.c:32-42
{
    Ray r;
    V3 rd = random_unit_disk(g) * ci.lr;
    V3 offset = ci.u0 * rd.x + ci.v0 * rd.y;
    r.o = ci.o + offset;
    r.d = ci.llc + (ci.h * u + ((ci.v * v - ci.o) - offset));
    return r;
}

// --------------------------------------------------------------- shading --
// What ray_color does after hit_list_hit_all returns a hit (caster.c:441-480),
// unrolled from recursion: returns true when the path continues along `next`
// with its throughput multiplied by `atten`; false when the path ends with
// radiance `emit` (black for a failed scatter or an unknown tag).
// TEX false: for scenes whose every texture is a solid colour (TTYPE_COLOR), the instance leaves out
// the checker, image and noise lookups altogether -- same values, a smaller kernel.
// DEFER: a noise texture is not evaluated here.  Its value only scales the path's throughput, which
// nothing reads before the path's next bounce, so the caller may evaluate it later, for several
// slots at once: `later` then names the material (-1: nothing left to do), `atten` is 1 and the
// point to evaluate at is next.o.
template <bool TEX = true, bool DEFER = false>
CDEV bool shade(const SceneView& sc, Rng& g, const Ray& r, HitRec& rec, Ray& next, V3& atten, V3& emit, int& later)
{
    later = -1;
    const Mat m = sc.mats[rec.m];
    emit = mk(0.0f, 0.0f, 0.0f);
    atten = mk(1.0f, 1.0f, 1.0f);
    // Lambertian and metal both start with one unit-sphere rejection sample and
    // nothing else is drawn before it (material.c:6,17), so the sampling loop is
    // shared: the wave runs it once for the lanes of both materials.
    V3 us = mk(0.0f, 0.0f, 0.0f);
    if (m.type == 0 || m.type == 1) us = random_unit_sphere(g);
    // Each scatter function normalises one vector first: the unit-sphere sample (lambertian,
    // material.c:6 through random_unit_norm) or the incoming direction (metal :16, dielectric :24).
    // One normalisation serves the lanes of all three.
    V3 unit = mk(0.0f, 0.0f, 0.0f);
    if (m.type >= 0 && m.type <= 2) unit = to_unit(m.type == 0 ? us : r.d);
    // The texture is looked up once, after the switch, for the lanes of every material that wants
    // it: the wave then runs the lookup (a long one for noise textures) once, not per material.
    bool cont = false, textured = false;
    switch (m.type) {
    case 0: // MTYPE_LAMBERTIAN, material.c:3-12
        CASTER_STAT(7);
        next.o = rec.p;
        next.d = rec.n + unit;
        if (near_zero(next.d)) next.d = rec.n;
        cont = true;
        textured = true;
        break;
    case 1: // MTYPE_METAL, material.c:14-19
        CASTER_STAT(8);
        next.o = rec.p;
        next.d = reflect(unit, rec.n) + us * m.fuzz;
        cont = dot(next.d, rec.n) > 0.0f;
        textured = cont;
        break;
    case 2: // MTYPE_DIELECTRIC
        CASTER_STAT(9);
        scatter_dielectric_unit(g, unit, rec, m.ior, next);
        cont = true;
        break;
    case 3: // MTYPE_DIFF_LIGHT, caster.c:470-479
        CASTER_STAT(10);
        if (m.tex == 0) emit = mk(m.c0[0], m.c0[1], m.c0[2]);
        else if (m.tex == 2) textured = true;
        break;
    default:
        break;
    }
    if (TEX && DEFER && textured && m.tex == 4 && sc.has_noise && m.type != 3) {
        later = rec.m;
        textured = false;
    }
    if (textured) {
        const V3 c = TEX ? texture_color<!DEFER>(sc, m, rec) : mk(m.c0[0], m.c0[1], m.c0[2]);
        if (m.type == 3) emit = c;
        else atten = c;
    }
    return cont;
}

template <bool TEX = true>
CDEV bool shade(const SceneView& sc, Rng& g, const Ray& r, HitRec& rec, Ray& next, V3& atten, V3& emit)
{
    int later;
    return shade<TEX, false>(sc, g, r, rec, next, atten, emit, later);
}

}  // namespace caster
