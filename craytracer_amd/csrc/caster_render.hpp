// The path-tracing kernel.  Replaces the recursion of ray_color
// (src/caster.c:425-491) and the per-pixel sample loop of job_func
// (src/dispatcher.c:12-30) by an iterative loop over a pool of paths.
//
// Work decomposition
//   * a work unit is an 8x8 pixel tile times a chunk of the sample range.  The
//     grid is persistent: every wave pulls units from a queue (one atomic add
//     per unit) until it is empty; a 256-thread block holds four waves that
//     share one copy of the sphere list in LDS (the operands of the exact test);
//   * every lane carries TWO path slots, so a wave traces 128 rays per loop
//     trip.  The conservative sphere test is a 32 x 32 matrix product per 32
//     spheres and 32 rays on the matrix cores (v_mfma_f32_32x32x16_f16, see "sphere
//     loop on the matrix cores"); small scenes and scenes that do not qualify run
//     it on packed FP32 (v_pk_*_f32: slot A in the low half, slot B in the high
//     half) with the sphere's numbers as scalar operands from the scalar cache;
//   * when a path ends, its radiance is added to the accumulator of the wave's
//     current tile in LDS and the slot takes the next (pixel, sample) of the
//     unit's pool: the finished slots are found with a ballot and numbered with
//     mbcnt, so the wave stays full.  When the pool runs dry the tile's sums go
//     to the frame buffer and the wave starts on its next unit at once; the
//     paths of the old unit that are still in flight add their radiance to the
//     frame buffer themselves when they end.  The wave never idles waiting for
//     long paths;
//   * all sums are 64-bit fixed point (2^-32 units): integer adds commute, so
//     a pixel's sum does not depend on the order in which paths end, on the
//     route their radiance takes or on the unit-to-wave assignment, and a
//     render is bit-reproducible;
//   * the frame buffer is a fixed-point buffer in HBM, one plane per channel
//     (64-bit integer atomics); a later kernel converts it to float once.
//
// Every ray is tested against every sphere (src/caster.c:204 has no early
// out): the loop evaluates a conservative candidate test, notes the few
// candidates in lists in LDS and runs the reference's root arithmetic on those.
#pragma once

#include "caster_device.hpp"
#include "caster_launch.h"

namespace caster {

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kBlock = 256;                 // 4 waves
constexpr int kWaves = kBlock / 64;
constexpr int kListCap = 16;                // candidate list entries per lane (u16 each) before the lists are worked off
constexpr int kTileW = 8, kTileH = 8;       // pixels per work unit

CDEV f2 splat(float x) { return f2{x, x}; }

struct Slot {
    V3 o, d;        // strict: d as the reference has it; fast: unit length
#if CASTER_STRICT
    float aa_;      // d.d
#else
    float tmin_;    // 0.001 * |d| of the ray as the reference has it (t is measured along the unit direction)
    float tmax_;    // 16777215 * |d|
#endif
    V3 thr;         // product of the attenuations so far
    Rng g;
    int q;          // pixel this path belongs to: column in bits 0-15, row in bits 16-31
    int depth;      // rays this path may still shoot; 0: the slot holds no path
    CDEV bool alive() const { return depth > 0; }
    // d.d, and the range of t (src/caster.c:441): constants of the mode where they are not stored
#if CASTER_STRICT
    CDEV float aa() const { return aa_; }
    CDEV float tmin() const { return kTMin; }
    CDEV float tmax() const { return kTMax; }
#else
    CDEV float aa() const { return 1.0f; }
    CDEV float tmin() const { return tmin_; }
    CDEV float tmax() const { return tmax_; }
#endif
};

// A slot without a path: a ray that starts far outside any scene and misses
// every sphere with a finite negative discriminant, so it never takes the
// candidate branch.
CDEV void park(Slot& s)
{
    s.o = mk(0.0f, 1.0e15f, 0.0f);
    s.d = mk(1.0f, 0.0f, 0.0f);
#if CASTER_STRICT
    s.aa_ = 1.0f;
#else
    s.tmin_ = kTMin;
    s.tmax_ = kTMax;
#endif
    s.depth = 0;
}

CDEV void set_ray(Slot& s, const Ray& r)
{
    s.o = r.o;
#if CASTER_STRICT
    s.d = r.d;
    s.aa_ = dot(r.d, r.d);
#else
    // t is measured along the unit direction: t' = t * |d|
    const float len2 = dot(r.d, r.d);
    const float inv = __frsqrt_rn(len2);
    const float len = len2 * inv;
    s.d = r.d * inv;
    s.tmin_ = kTMin * len;
    s.tmax_ = kTMax * len;
#endif
}

struct TileGeom {
    int x0, y0;     // pixel of the tile's corner
    int tw, th;     // valid pixels (partial tiles at the right / top edge)
};

// A work unit as a wave sees it (all fields wave-uniform).
struct Unit {
    int chunk;      // index of the sample chunk, -1 = no unit
    TileGeom tg;
    int s0, cs;     // first sample and number of samples of the chunk
    int next, total;// pool cursor / pool size (pixels x samples)
};

constexpr int kAccStride = kTileW * kTileH * 3;     // one accumulator set: 64 pixels x RGB
typedef unsigned long long acc_t;                   // fixed point, 32 fractional bits

// Pulls the next unit off the queue.  Units are numbered chunk-major.
CDEV Unit fetch_unit(const RenderArgs& a, int tiles_x, int n_tiles, int lane)
{
    unsigned int u = 0;
    if (lane == 0) u = atomicAdd(a.queue, 1u);
    u = (unsigned int)__builtin_amdgcn_readfirstlane((int)u);
    Unit un;
    un.chunk = -1; un.s0 = 0; un.cs = 0; un.next = 0; un.total = 0;
    un.tg.x0 = un.tg.y0 = 0; un.tg.tw = un.tg.th = 0;
    if (u >= (unsigned int)a.n_units) return un;
    const int chunk = (int)(u / (unsigned int)n_tiles);
    const int tile = (int)u - chunk * n_tiles;
    un.chunk = chunk;
    un.tg.x0 = (tile % tiles_x) * kTileW;
    un.tg.y0 = (tile / tiles_x) * kTileH;
    un.tg.tw = (a.width - un.tg.x0 < kTileW) ? a.width - un.tg.x0 : kTileW;
    un.tg.th = (a.height - un.tg.y0 < kTileH) ? a.height - un.tg.y0 : kTileH;
    un.s0 = a.s_begin + a.chunk_start[chunk];
    un.cs = a.chunk_start[chunk + 1] - a.chunk_start[chunk];
    un.total = (a.max_depth > 0) ? un.tg.tw * un.tg.th * un.cs : 0;
    return un;
}

// The divisions (i + jitter) / (width - 1) and (j + jitter) / (height - 1) of src/dispatcher.c:21-22
// share their denominators over the whole frame.  The compiler's correctly rounded division is a
// reciprocal refined once and a chain of four fused corrections, wrapped in a scaling of operands in
// extreme ranges; for a numerator in [0, 65536) and an integer denominator in [1, 65535] nothing is
// scaled, so the same chain on a reciprocal kept from the kernel's start gives the same bits
// (tools/div_check.hip compares 7e10 quotients).  A frame one pixel wide or high divides by zero as
// the reference does: the plain division then.
struct PixelScale {
    float dw, rw, dh, rh;   // denominators and their refined reciprocals (wave-uniform)
    bool chain;
};

CDEV PixelScale pixel_scale(int width, int height)
{
    PixelScale p;
    p.dw = (float)(width - 1);
    p.dh = (float)(height - 1);
    p.chain = width > 1 && height > 1;
    const float w0 = __builtin_amdgcn_rcpf(p.dw), h0 = __builtin_amdgcn_rcpf(p.dh);
    const float rw = __builtin_fmaf(__builtin_fmaf(-p.dw, w0, 1.0f), w0, w0);
    const float rh = __builtin_fmaf(__builtin_fmaf(-p.dh, h0, 1.0f), h0, h0);
    p.rw = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rw)));
    p.rh = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rh)));
    return p;
}

CDEV float divide_by(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

// Camera ray for pool entry m of the unit: src/dispatcher.c:21-24.
CDEV void start_pool_path(Slot& s, int m, const Unit& un, const RenderArgs& a, const Cam& cam, const PixelScale& ps)
{
    const int nvalid = un.tg.tw * un.tg.th;
    int smp, qx, qy;
    if (nvalid == 64) {
        smp = m >> 6; qx = m & 7; qy = (m >> 3) & 7;
    } else {
        smp = m / nvalid;
        const int rem = m - smp * nvalid;
        qy = rem / un.tg.tw;
        qx = rem - qy * un.tg.tw;
    }
    const int i = un.tg.x0 + qx, j = un.tg.y0 + qy;
    rng_seed(s.g, a.seed, (uint32_t)(j * a.width + i), (uint32_t)(un.s0 + smp));
    const float un_ = (float)i + random_f32(s.g);
    const float vn_ = (float)j + random_f32(s.g);
    float u, v;
    if (ps.chain) {
        u = divide_by(un_, ps.dw, ps.rw);
        v = divide_by(vn_, ps.dh, ps.rh);
    } else {
        u = un_ / ps.dw;
        v = vn_ / ps.dh;
    }
    const Ray r = ray_from_cam(cam, s.g, u, v);
    set_ray(s, r);
    s.thr = mk(1.0f, 1.0f, 1.0f);
    s.q = (j << 16) | i;
    s.depth = a.max_depth;
}

// Adds the sums of a unit's tile to the frame's fixed-point buffer and clears the accumulator.  The
// buffer is three planes (R, G, B) of one word per pixel: the eight lanes of a tile row then add to
// eight consecutive words, 64 bytes, which leave the L2 as one atomic request (with the channels
// interleaved per pixel every lane's atomic was a 64-byte request of its own: 8 x the traffic).
CDEV void flush_unit(const Unit& un, acc_t* acc_set, const RenderArgs& a, int lane)
{
    const int qx = lane & 7, qy = lane >> 3;
    if (qx < un.tg.tw && qy < un.tg.th) {
        const size_t plane = (size_t)a.width * (size_t)a.height;
        acc_t* px = a.fixed + ((size_t)(un.tg.y0 + qy) * a.width + (un.tg.x0 + qx));
        const acc_t r = acc_set[lane * 3 + 0], g = acc_set[lane * 3 + 1], b = acc_set[lane * 3 + 2];
        if (r) atomicAdd(px, r);
        if (g) atomicAdd(px + plane, g);
        if (b) atomicAdd(px + 2 * plane, b);
    }
    acc_set[lane * 3 + 0] = 0; acc_set[lane * 3 + 1] = 0; acc_set[lane * 3 + 2] = 0;
    __builtin_amdgcn_wave_barrier();
}

// floor(x * 2^32) for a radiance value: a product of colours and an emission, finite and >= 0 for
// any sane scene (clamped to [0, 1e9], NaN -> 0).  The integer part and the fraction are converted
// separately: x - trunc(x) is exact in binary32, and so is its product with 2^32, so the result
// equals (u64)((double)x * 2^32) bit for bit without a double-to-u64 conversion or 64-bit shifts.
CDEV acc_t to_fixed(float x)
{
    const float c = fminf(fmaxf(x, 0.0f), 1.0e9f);
    const unsigned int hi = (unsigned int)c;                    // truncates; c < 2^32
    const float frac = c - (float)hi;                           // exact (0 when c >= 2^24)
    const unsigned int lo = (unsigned int)(frac * 4294967296.0f);
    return ((acc_t)hi << 32) | (acc_t)lo;
}

// ---- sphere loop ---------------------------------------------------------
// Two tables describe the spheres:
//   filter[k] = {cx-Px, cy-Py, g*(C+r*r)-(C-r*r), cz-Pz}, with C = |c-P|^2 and P a reference
//               point near the scene's spheres: built by the host, read by every wave through
//               the scalar cache (the entries are wave-uniform and live in SGPRs);
//   exact[k]  = {cx, cy, r*r, cz}: the operands of the reference's test (src/caster.c:101-105),
//               a tile of them staged in LDS for the candidate code.
// The value order puts each operand where a packed instruction can broadcast it to both
// halves of a register pair.
//
// Every ray meets every sphere (the reference has no early out), but only a
// few spheres per ray can be hit.  The loop therefore evaluates, for both slots
// of the lane at once, a conservative candidate test (see PairRays: 7 packed
// FP32 instructions per sphere and lane pair) and runs the reference's own
// arithmetic only for the candidates.  The test carries a margin
// g*(|o'|^2 + C + r*r) that exceeds the rounding error of both its own form and
// the reference's form, so a sphere the reference's test would accept is
// always a candidate; what the candidate code then decides is exactly what the
// reference decides (strict mode) or the same test on the unit direction (fast
// mode).  Spheres much larger than the rest (ground, sky) are candidates for
// every ray and are tested directly (test_always).
constexpr int kGroup = 4;               // spheres per software-pipeline stage
#if CASTER_STRICT
constexpr float kFilterMargin = kFilterMarginStrict;
#else
constexpr float kFilterMargin = kFilterMarginFast;
#endif
constexpr int kLdsSlack = 2 * kGroup;   // read-ahead beyond the padded tile

CDEV float4 exact_image(const float4 s /* cx, cy, cz, r */) { return make_float4(s.x, s.y, s.w * s.w, s.z); }
// Pad entries: r*r = -1e30 gives a finite negative discriminant for every ray.
CDEV float4 exact_pad() { return make_float4(0.0f, 0.0f, -1.0e30f, 0.0f); }
struct SphereLds {
    const float4* exact;
    const float4* filter; // the whole scene's filter entries, in device memory (read through the scalar cache)
    int count;            // spheres in the tile, without the padding
};

// The filter works in coordinates relative to the reference point P (o' = o - P, c' = c - P) with
// the unit direction u.  With w = o' - (o'.u)u, the part of o' perpendicular to the ray,
//     discriminant / (d.d) = (c'.u)^2 + 2 c'.w - (|c'|^2 - r^2) - |w|^2 ,
// so a sphere can only be hit if  (c'.u)^2 + (2w).c' - K  >=  |w|^2  (K = |c'|^2 - r^2 from the
// table): seven packed operations per sphere and lane on the left, a number per ray on the right.
// Rounding is covered by the margin g*(|o'|^2 + |c'|^2 + r^2): its sphere part is folded into the
// table entry, its ray part into the threshold.
struct PairRays {     // filter operands of both slots of a lane, packed
    f2 ux, uy, uz, p2x, p2y, p2z;
    float thrA, thrB;           // a key below its slot's threshold: no hit possible
    bool finiteA, finiteB;      // a ray with a non-finite component takes no part in the sphere loop (see non_finite_rule)
};

CDEV PairRays pack_rays(const Slot& sa, const Slot& sb, const float3 ref)
{
    PairRays r;
    const f2 qx = f2{sa.o.x, sb.o.x} - splat(ref.x);
    const f2 qy = f2{sa.o.y, sb.o.y} - splat(ref.y);
    const f2 qz = f2{sa.o.z, sb.o.z} - splat(ref.z);
#if CASTER_STRICT
    // the slots keep the reference's unnormalised direction; the filter wants a unit one
    const f2 inv = f2{__frsqrt_rn(sa.aa()), __frsqrt_rn(sb.aa())};
    r.ux = f2{sa.d.x, sb.d.x} * inv; r.uy = f2{sa.d.y, sb.d.y} * inv; r.uz = f2{sa.d.z, sb.d.z} * inv;
#else
    r.ux = f2{sa.d.x, sb.d.x}; r.uy = f2{sa.d.y, sb.d.y}; r.uz = f2{sa.d.z, sb.d.z};
#endif
    const f2 oo = qx * qx + qy * qy + qz * qz;
    const f2 od = qx * r.ux + qy * r.uy + qz * r.uz;
    const f2 wx = qx - od * r.ux, wy = qy - od * r.uy, wz = qz - od * r.uz;
    const f2 thr = (wx * wx + wy * wy + wz * wz) - oo * splat(kFilterMargin);
    r.p2x = wx + wx; r.p2y = wy + wy; r.p2z = wz + wz;
    r.thrA = thr.x; r.thrB = thr.y;
    // the threshold is finite exactly when every component of the ray is (overflow aside, which
    // makes the thresholds NaN or inf: the ray is then treated like a non-finite one)
    r.finiteA = (thr.x - thr.x == 0.0f);
    r.finiteB = (thr.y - thr.y == 0.0f);
    return r;
}

// Per sphere: the left-hand side of the candidate test.
CDEV f2 candidate_key(const PairRays& r, const float4 f /* c'x, c'y, g*scale - K, c'z */)
{
    f2 a = r.ux * splat(f.x);
    a = __builtin_elementwise_fma(r.uy, splat(f.y), a);
    a = __builtin_elementwise_fma(r.uz, splat(f.w), a);
    // b = p2z * c'z + K: both scalars sit in one SGPR pair, which a single instruction may read twice
    // (left to itself the compiler copies K to a VGPR first: one more VALU instruction per sphere)
    f2 b;
    const f2 zw = {f.z, f.w};
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(b) : "v"(r.p2z), "s"(zw));
    b = __builtin_elementwise_fma(r.p2y, splat(f.y), b);
    b = __builtin_elementwise_fma(r.p2x, splat(f.x), b);
    return __builtin_elementwise_fma(a, a, b);
}

// The t at which the reference would report this sphere for this ray, if any, REGARDLESS of
// the closest hit so far (src/caster.c:101-121): the first root if it is not below tmin, else
// the second root if that is not below tmin.  The reference's in-order loop, which accepts a
// sphere when that t does not exceed the closest t so far (an equal t replaces the earlier
// sphere), ends with the smallest such t and, among equal ones, the latest sphere; a first
// root beyond the closest hit is rejected there together with the second root, which is
// never smaller.  So the closest hit is a minimum over (t, later-sphere-first) and can be taken
// in any order.
// Second half of candidate_root: from oc.d (hb), oc.oc - r*r (c) and the discriminant to the root.
CDEV bool root_tail(float hb, float c, float disc, float aa, float tmin, float& t)
{
#if CASTER_STRICT
    // The reference's arithmetic, operation for operation.  One early return that cannot change
    // its outcome: hb >= 0 and c >= 0 give sqrt_d <= hb (sqrt(fl(hb*hb)) == hb exactly and the
    // discriminant is not larger), so both numerators are <= 0 and both roots are below tmin > 0.
    if (disc < 0.0f) return false;
    if (hb >= 0.0f && c >= 0.0f) return false;
    const float sq = sqrtf(disc);
    float root = (-hb - sq) / aa;
    if (root < tmin) {
        root = (-hb + sq) / aa;
        if (root < tmin) return false;
    }
    t = root;
    return true;
#else
    (void)aa;   // the direction has unit length
    if (!(disc >= 0.0f && (hb < 0.0f || c < 0.0f))) return false;
    const float sq = __builtin_amdgcn_sqrtf(disc);
    float root = -hb - sq;
    if (root < tmin) root = -hb + sq;
    if (root < tmin) return false;
    t = root;
    return true;
#endif
}

CDEV bool candidate_root(const float4 e /* cx, cy, r*r, cz */, V3 o, V3 d, float aa, float tmin, float& t)
{
    const float ocx = o.x - e.x, ocy = o.y - e.y, ocz = o.z - e.w;
#if CASTER_STRICT
    const float hb = (ocx * d.x + ocy * d.y) + ocz * d.z;
    const float c = ((ocx * ocx + ocy * ocy) + ocz * ocz) - e.z;
    const float disc = hb * hb - aa * c;
#else
    const float hb = fmaf(ocz, d.z, fmaf(ocy, d.y, ocx * d.x));
    const float c = fmaf(ocz, ocz, fmaf(ocy, ocy, fmaf(ocx, ocx, -e.z)));
    const float disc = fmaf(hb, hb, -c);
#endif
    return root_tail(hb, c, disc, aa, tmin, t);
}

// Takes the hit at t on sphere `index` if it beats what the slot has: the reference's in-order
// loop keeps the smallest t and, among equal ones, the latest sphere (src/caster.c:209-214).
CDEV void take_hit(float t, int index, float& closest, int& best)
{
    if (t < closest || (t == closest && index > best)) { closest = t; best = index; }
}

// Candidates are not examined where they are found: a sphere is a candidate for a handful of
// the wave's 128 rays, and the root arithmetic (sqrt, divisions) would run for the whole wave
// each time.  A lane that has a candidate in a group of kGroup spheres appends one 16-bit entry
// to its own list in LDS -- group number and one bit per (sphere, slot) -- and the lists are
// worked off after the tile, all lanes together, one candidate per lane and turn.
struct CandLists {
    unsigned short* mem;    // [kListCap][64], this wave's; entry i of a lane at mem[i * 64 + lane]
    int cnt;                // entries in this lane's list
};

CDEV void drain_candidates(CandLists& cl, const SphereLds& lds, int base, const Slot& sa, const Slot& sb,
                           float& closestA, float& closestB, int& bestA, int& bestB, WaveTimers& tm)
{
    const unsigned long long t0 = tm.now();
    int cur = 0;
    unsigned int bits = 0, group = 0;
    for (;;) {
        if (bits == 0 && cur < cl.cnt) {
            const unsigned int e = cl.mem[cur * 64];
            ++cur;
            bits = e & 0xFFu;
            group = e >> 8;
        }
        if (!__any(bits != 0)) break;
        CASTER_STAT(1);
        if (bits != 0) {
            // bit 7 - 2j - s stands for sphere j of the group and slot s
            const int b = 31 - __clz((int)bits);
            bits &= ~(1u << b);
            const int j = 7 - b;
            const int k = (int)group * kGroup + (j >> 1);
            const bool second = (j & 1) != 0;
            V3 o, d;
            o.x = second ? sb.o.x : sa.o.x; o.y = second ? sb.o.y : sa.o.y; o.z = second ? sb.o.z : sa.o.z;
            d.x = second ? sb.d.x : sa.d.x; d.y = second ? sb.d.y : sa.d.y; d.z = second ? sb.d.z : sa.d.z;
            const float aa = second ? sb.aa() : sa.aa();
            const float tmin = second ? sb.tmin() : sa.tmin();
            float t;
            if (candidate_root(lds.exact[k], o, d, aa, tmin, t)) {
                // selects on values, not on which variable to update: the four stay in registers
                float closest = second ? closestB : closestA;
                int best = second ? bestB : bestA;
                take_hit(t, base + k, closest, best);
                closestA = second ? closestA : closest; bestA = second ? bestA : best;
                closestB = second ? closest : closestB; bestB = second ? best : bestB;
            }
        }
    }
    cl.cnt = 0;
    tm.add(WaveTimers::kLists, t0);
}

// A pipeline stage: kGroup spheres held in named registers (no indexed array,
// so nothing can end up in scratch memory).
struct Group {
    float4 g0, g1, g2, g3;
};

// The filter entries are the same for every lane: they are read through the scalar cache into
// SGPRs (constant address space -> s_load_dwordx16 per group) and enter the packed instructions
// as scalar operands.  Against a broadcast read from LDS this saves the vector-register writes and
// 32 VGPRs, and measured 5 % of the frame time.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* filter_ptr;
CDEV float4 entry(filter_ptr src, int k) { const v4f v = src[k]; return make_float4(v.x, v.y, v.z, v.w); }

CDEV Group load_group(filter_ptr src, int k)
{
    Group g;
    g.g0 = entry(src, k); g.g1 = entry(src, k + 1);
    g.g2 = entry(src, k + 2); g.g3 = entry(src, k + 3);
    return g;
}

// Shifts the sign bit of x into m from the right.
CDEV unsigned int or3(unsigned int a, unsigned int b, unsigned int c) { return a | b | c; }
CDEV unsigned int push_sign(unsigned int m, float x) { return __builtin_amdgcn_alignbit(m, __float_as_uint(x), 31u); }

// Evaluates the candidate keys of a group and looks at them together: a sphere is a candidate
// for some ray of the wave only a few percent of the time, so one test of the group's largest
// key per slot (max3 chains) replaces two compares per sphere.  A non-finite ray has NaN
// thresholds and is never a candidate.
CDEV void process_group(const Group& g, int k, const PairRays& r, CandLists& cl, const SphereLds& lds, int base,
                        const Slot& sa, const Slot& sb, float& closestA, float& closestB, int& bestA, int& bestB,
                        WaveTimers& tm)
{
    const f2 k0 = candidate_key(r, g.g0);
    const f2 k1 = candidate_key(r, g.g1);
    const f2 k2 = candidate_key(r, g.g2);
    const f2 k3 = candidate_key(r, g.g3);
    const float mA = __builtin_fmaxf(__builtin_fmaxf(k0.x, k1.x), __builtin_fmaxf(k2.x, k3.x));
    const float mB = __builtin_fmaxf(__builtin_fmaxf(k0.y, k1.y), __builtin_fmaxf(k2.y, k3.y));
    const bool mine = (mA >= r.thrA) | (mB >= r.thrB);
    if (__any(mine)) {
        CASTER_STAT(12);
        if (__any(cl.cnt >= kListCap)) drain_candidates(cl, lds, base, sa, sb, closestA, closestB, bestA, bestB, tm);
        // the sign bits of key - threshold, first key in bit 7; a clear bit is a candidate
        unsigned int neg = push_sign(0u, k0.x - r.thrA);
        neg = push_sign(neg, k0.y - r.thrB);
        neg = push_sign(neg, k1.x - r.thrA);
        neg = push_sign(neg, k1.y - r.thrB);
        neg = push_sign(neg, k2.x - r.thrA);
        neg = push_sign(neg, k2.y - r.thrB);
        neg = push_sign(neg, k3.x - r.thrA);
        neg = push_sign(neg, k3.y - r.thrB);
        if (mine) {
            cl.mem[cl.cnt * 64] = (unsigned short)(((unsigned int)(k / kGroup) << 8) | (~neg & 0xFFu));
            ++cl.cnt;
        }
    }
}

// Sphere loop over one tile for the two slots of a lane.  cnt is a multiple of
// 2*kGroup (the tables are padded) and kLdsSlack entries beyond it are readable.
// Two register sets alternate: while one group is evaluated the next one is
// already on its way from the scalar cache.
CDEV void isect_tile(const SphereLds lds, int cnt, int base, const PairRays& r, CandLists& cl,
                     const Slot& sa, const Slot& sb, float& closestA, float& closestB, int& bestA, int& bestB,
                     WaveTimers& tm)
{
    filter_ptr src = (filter_ptr)(lds.filter + base);
    Group ga = load_group(src, 0);
#pragma unroll 1
    for (int k = 0; k < cnt; k += 2 * kGroup) {
        const Group gb = load_group(src, k + kGroup);
        process_group(ga, k, r, cl, lds, base, sa, sb, closestA, closestB, bestA, bestB, tm);
        ga = load_group(src, k + 2 * kGroup);
        process_group(gb, k + kGroup, r, cl, lds, base, sa, sb, closestA, closestB, bestA, bestB, tm);
    }
    drain_candidates(cl, lds, base, sa, sb, closestA, closestB, bestA, bestB, tm);
}

// ---- sphere loop on the matrix cores -------------------------------------
// The candidate key is bilinear in a feature vector of the sphere and one of the ray:
//     (c'.u)^2 + (2w).c' + Z  =  sum_f S_f R_f + Z ,
//     S = (cx^2, cy^2, cz^2, cx cy, cx cz, cy cz, cx, cy, cz)          c' = scale * (c - P)
//     R = (ux^2, uy^2, uz^2, 2 ux uy, 2 ux uz, 2 uy uz, 2 wx, 2 wy, 2 wz)
// so the keys of 32 spheres and 32 rays are a 32 x 32 matrix product, and the matrix cores
// compute it while the vector unit only looks for the largest key.  The operands are
// half-precision: every feature is split into two halves x = hi + lo (22 bits together) and
// the three products hi*hi, hi*lo, lo*hi are summed (the matrix unit multiplies exactly and
// accumulates in FP32): 27 of the 32 k-slots of two v_mfma_f32_32x32x16_f16, two more for
// Z = Zhi + Zlo against a ray entry of 1.  The power-of-two scale keeps every sphere feature
// within half-precision range; what is lost against the FP32 form of the filter is covered by
// a larger margin (kFilterMarginMfmaExtra) and an absolute term in Z.
//
// Operand layout (lane l, r = l & 31, h = l >> 5): the sphere operand of block b and k-half q
// comes ready-made from the table, mfma_tab[(2b + q) * 64 + l] = slots 16q + 8h .. + 7 of
// sphere 32b + r.  A wave's 128 rays form four column blocks: block 2s + g holds slot s of
// lanes 32g .. 32g + 31.  Each lane builds the 32 slots of its own two rays; one
// v_permlane32_swap per register then leaves lane l with slots 16q + 8h .. of ray r of the
// column block, which is where the instruction wants them.  The result has the ray on the
// lane and 16 spheres in the registers: rows 8(v >> 2) + 4h + (v & 3), v = 0..15.
//
// A lane that finds a candidate is not the lane that owns the ray, and the candidates are spread
// unevenly over the rays: they go to a queue of the wave and are worked off by all lanes together
// (see MfmaQueue).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));

#if CASTER_STRICT
constexpr float kFilterMarginMfma = kFilterMarginStrict + kFilterMarginMfmaExtra;
#else
constexpr float kFilterMarginMfma = kFilterMarginFast + kFilterMarginMfmaExtra;
#endif

struct RayHalves {
    h8 f00, f01, f10, f11;      // slots 0-7, 8-15, 16-23, 24-31
};
constexpr float kThrScale = 1024.0f;    // the threshold enters the product as (threshold / kThrScale) * kThrScale: keeps it in half-precision range

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// Two floats to two halves (round to nearest even), in one register: v_cvt_pk_f16_f32.
CDEV unsigned int halves(float lo, float hi) { return __builtin_bit_cast(unsigned int, __builtin_convertvector(f2{lo, hi}, h2)); }
CDEV float low_half(unsigned int pair) { return (float)__builtin_bit_cast(h2, pair).x; }
CDEV float high_half(unsigned int pair) { return (float)__builtin_bit_cast(h2, pair).y; }

// The 32 k-slots of one ray from its nine features f (already zero for a ray that is never, or
// always, a candidate) and the three pieces of its threshold.  Feature k takes slots 3k (its upper
// half h, against the sphere's upper half), 3k + 1 (the rest l = f - h, against the upper half) and
// 3k + 2 (h again, against the sphere's rest); slots 27, 28 hold 1 (against the two halves of the
// sphere's constant), 29-31 the threshold.  Pairs of slots that sit in one register are converted
// together.
CDEV RayHalves split_features(const float (&f)[9], float t0, float t1, float t2)
{
    const unsigned int h01 = halves(f[0], f[1]), h23 = halves(f[2], f[3]), h45 = halves(f[4], f[5]),
                       h67 = halves(f[6], f[7]), h8o = halves(f[8], 1.0f);
    const float r0 = f[0] - low_half(h01), r1 = f[1] - high_half(h01), r2 = f[2] - low_half(h23),
                r3 = f[3] - high_half(h23), r4 = f[4] - low_half(h45), r5 = f[5] - high_half(h45),
                r6 = f[6] - low_half(h67), r7 = f[7] - high_half(h67), r8 = f[8] - low_half(h8o);
    RayHalves r;
    r.f00 = __builtin_bit_cast(h8, make_uint4(halves(f[0], r0), h01, halves(r1, f[1]), halves(f[2], r2)));
    r.f01 = __builtin_bit_cast(h8, make_uint4(h23, halves(r3, f[3]), halves(f[4], r4), h45));
    r.f10 = __builtin_bit_cast(h8, make_uint4(halves(r5, f[5]), halves(f[6], r6), h67, halves(r7, f[7])));
    r.f11 = __builtin_bit_cast(h8, make_uint4(halves(f[8], r8), h8o, halves(1.0f, t0), halves(t1, t2)));
    return r;
}

// x keeps its lower 32 lanes and takes those of y into its upper half; y keeps its upper 32
// lanes and takes those of x into its lower half.
CDEV void swap_halves(unsigned int& x, unsigned int& y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
CDEV void swap_halves(h8& x, h8& y)
{
    uint4 a = __builtin_bit_cast(uint4, x), b = __builtin_bit_cast(uint4, y);
    swap_halves(a.x, b.x); swap_halves(a.y, b.y); swap_halves(a.z, b.z); swap_halves(a.w, b.w);
    x = __builtin_bit_cast(h8, a);
    y = __builtin_bit_cast(h8, b);
}

struct MfmaRays {
    h8 b00, b01, b10, b11, b20, b21, b30, b31;      // b<column block><k-half>
    bool finiteA, finiteB;
};

// The filter's ray operand of one slot: features, threshold, k-slots.  (Scalar FP32 on purpose: with
// packed operations in the kernel that holds the matrix instructions this toolchain has produced
// code that misses candidates, differently from run to run.)
CDEV RayHalves ray_halves(const Slot& s, const float3 ref, float scale, bool& finite)
{
    const float qx = (s.o.x - ref.x) * scale, qy = (s.o.y - ref.y) * scale, qz = (s.o.z - ref.z) * scale;
#if CASTER_STRICT
    const float inv = __frsqrt_rn(s.aa());
    const float ux = s.d.x * inv, uy = s.d.y * inv, uz = s.d.z * inv;
#else
    const float ux = s.d.x, uy = s.d.y, uz = s.d.z;
#endif
    const float oo = qx * qx + qy * qy + qz * qz;
    const float od = qx * ux + qy * uy + qz * uz;
    const float wx = qx - od * ux, wy = qy - od * uy, wz = qz - od * uz;
    const float thr = (wx * wx + wy * wy + wz * wz) - oo * kFilterMarginMfma;
    finite = (thr - thr == 0.0f);
    float f[9] = {ux * ux, uy * uy, uz * uz, 2.0f * ux * uy, 2.0f * ux * uz, 2.0f * uy * uz, wx + wx, wy + wy, wz + wz};
    const float big = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(f[6]), __builtin_fabsf(f[7])), __builtin_fabsf(f[8]));
    // A slot without a path, or a non-finite ray (non_finite_rule decides for it): never a
    // candidate.  A ray so far out that its features leave half-precision range: a candidate for
    // every sphere (the queue code then runs the exact test on all of them).
    const bool never = !s.alive() || !finite;
    const float tn = -thr * (1.0f / kThrScale);             // three halves of it go against kThrScale in the sphere operand
    const bool all = !never && !(big < 60000.0f && __builtin_fabsf(tn) < 60000.0f);
    // the threshold in three halves: t0 + t1 + t2
    const float r1 = tn - (float)(_Float16)tn;
    const float r2 = r1 - (float)(_Float16)r1;
    float t0 = tn, t1 = r1, t2 = r2;
    if (never || all) {
#pragma unroll
        for (int i = 0; i < 9; ++i) f[i] = 0.0f;
        // the key is then the sphere's constant alone (at most 65504 in size) plus 3 * 60000 * kThrScale of either sign
        t0 = t1 = t2 = never ? -60000.0f : 60000.0f;
    }
    return split_features(f, t0, t1, t2);
}

CDEV MfmaRays pack_rays_mfma(const Slot& sa, const Slot& sb, const float3 ref, float scale)
{
    MfmaRays m;
    RayHalves ra = ray_halves(sa, ref, scale, m.finiteA);
    RayHalves rb = ray_halves(sb, ref, scale, m.finiteB);
    swap_halves(ra.f00, ra.f01); swap_halves(ra.f10, ra.f11);
    swap_halves(rb.f00, rb.f01); swap_halves(rb.f10, rb.f11);
    m.b00 = ra.f00; m.b10 = ra.f01; m.b01 = ra.f10; m.b11 = ra.f11;
    m.b20 = rb.f00; m.b30 = rb.f01; m.b21 = rb.f10; m.b31 = rb.f11;
    return m;
}

// Candidates of the matrix loop.  A lane that finds candidates in a column block appends one entry --
// whose ray, which block, the 16-bit mask -- to a queue of the WAVE in LDS (the position comes from a
// ballot, no atomics).  The queue is worked off by all lanes together, 64 entries at a time: a lane
// takes an entry, fetches the ray from its owner's registers (ds_bpermute), runs the reference's
// arithmetic (candidate_root) on the entry's spheres and merges a hit into the owner's result in LDS
// with a 64-bit atomic maximum over (~t, sphere + 1): the smallest t and, among equal ones, the latest
// sphere -- what take_hit does.  A ray with many candidates thus costs the wave no more than a ray with
// one: the work is spread over the lanes.
//
// Entry: bits 0-15 mask (register v in bit 15 - v), 16-22 block within the tile, 23-27 ray r of the
// column block, 28 column block & 1 (owner lane = 32 * that + r), 29 slot, 30 writer half h (rows
// 8 (v >> 2) + 4 h + (v & 3) of the block).
constexpr int kMfmaTileBlocks = 128;        // blocks per tile (7 bits in an entry)
constexpr int kMfmaQueueCap = 512;          // entries; one block of 32 spheres adds 256 at most
constexpr int kMfmaQueueBytes = kMfmaQueueCap * 4 + 128 * 8;       // per wave: the queue and the 128 results
struct MfmaQueue {
    unsigned int* entries;          // this wave's [kMfmaQueueCap]
    unsigned long long* result;     // this wave's [128]: slot * 64 + lane
    int count;                      // wave-uniform
    unsigned int head;              // this lane's part of an entry: ray r = lane & 31 in bits 23-27, writer half in bit 30
};

CDEV unsigned long long hit_key(float t, int index)
{
    return ((unsigned long long)(~__float_as_uint(t)) << 32) | (unsigned long long)(unsigned int)(index + 1);
}

// Before the loop: the results start from what the directly tested spheres gave.
CDEV void queue_begin(MfmaQueue& q, float closestA, float closestB, int bestA, int bestB)
{
    const int lane = threadIdx.x & 63;
    q.result[lane] = hit_key(closestA, bestA);
    q.result[64 + lane] = hit_key(closestB, bestB);
    q.count = 0;
    q.head = (((unsigned int)lane & 31u) << 23) | (((unsigned int)lane >> 5) << 30);
}

// After the last drain: every lane takes its two results back.
CDEV void queue_end(const MfmaQueue& q, float& closestA, float& closestB, int& bestA, int& bestB)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long ka = q.result[lane], kb = q.result[64 + lane];
    closestA = __uint_as_float(~(unsigned int)(ka >> 32)); bestA = (int)(unsigned int)ka - 1;
    closestB = __uint_as_float(~(unsigned int)(kb >> 32)); bestB = (int)(unsigned int)kb - 1;
}

CDEV float fetch(int from, float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(from << 2, __float_as_int(x))); }

// base: first sphere of the tile.  The exact operands of a candidate come from the scene's image in
// LDS, or (EXACT_LDS false: scenes larger than that image) from the sphere table in device memory.
template <bool EXACT_LDS>
CDEV void drain_mfma(MfmaQueue& q, const SceneView& sc, const float4* lds_exact, int base, const Slot& sa, const Slot& sb,
                     WaveTimers& tm)
{
    const unsigned long long t0 = tm.now();
    const int lane = threadIdx.x & 63;
    // LDS instructions of one wave complete in order: the reads below see the queue as written
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int first = 0; first < q.count; first += 64) {
        const unsigned int e = (first + lane < q.count) ? q.entries[first + lane] : 0u;
        unsigned int bits = e & 0xFFFFu;
        const int owner = (int)((e >> 23) & 63u);           // ray r and the column block's parity
        const bool second = ((e >> 29) & 1u) != 0u;
        // the ray: every lane reads its owner's registers (all lanes take part in a ds_bpermute)
        V3 o, d;
        {
            const float ax = fetch(owner, sa.o.x), ay = fetch(owner, sa.o.y), az = fetch(owner, sa.o.z);
            const float bx = fetch(owner, sb.o.x), by = fetch(owner, sb.o.y), bz = fetch(owner, sb.o.z);
            o.x = second ? bx : ax; o.y = second ? by : ay; o.z = second ? bz : az;
        }
        {
            const float ax = fetch(owner, sa.d.x), ay = fetch(owner, sa.d.y), az = fetch(owner, sa.d.z);
            const float bx = fetch(owner, sb.d.x), by = fetch(owner, sb.d.y), bz = fetch(owner, sb.d.z);
            d.x = second ? bx : ax; d.y = second ? by : ay; d.z = second ? bz : az;
        }
        const float aa_a = fetch(owner, sa.aa()), aa_b = fetch(owner, sb.aa());
        const float tm_a = fetch(owner, sa.tmin()), tm_b = fetch(owner, sb.tmin());
        const float aa = second ? aa_b : aa_a;
        const float tmin = second ? tm_b : tm_a;
        unsigned long long* dst = q.result + (second ? 64 : 0) + owner;
        const int row0 = base + (int)((e >> 16) & 127u) * 32 + (int)((e >> 30) & 1u) * 4;
        while (__any(bits != 0u)) {
            CASTER_STAT(1);
            if (bits != 0u) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1u;
                const int v = 15 - b;
                const int k = row0 + (v >> 2) * 8 + (v & 3);
                if (k < sc.n_sph) {
                    const float4 ex = EXACT_LDS ? lds_exact[k] : exact_image(sc.sph[k]);
                    float t;
                    if (candidate_root(ex, o, d, aa, tmin, t)) atomicMax(dst, hit_key(t, k));
                }
            }
        }
    }
    q.count = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    tm.add(WaveTimers::kLists, t0);
}

// One column block against one sphere block: two matrix instructions leave threshold - key of 16
// spheres in the lane's registers (the table holds the negated sphere operand); a set sign bit is a
// candidate.  The 16 signs are gathered into a mask (one v_alignbit each; register v in bit 15 - v)
// and the lanes with candidates append their entries to the wave's queue.
template <int CB>
CDEV void mfma_column(const h8 a0, const h8 a1, const h8 b0, const h8 b1, unsigned int block_head, MfmaQueue& q)
{
    f16x acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
    // Three column blocks in four have no candidate at all: the sign of the OR of the 16 results
    // tells (8 three-input ORs), and the masks are gathered only where some lane has one.
    unsigned int any = __float_as_uint(acc[0]);
#pragma unroll
    for (int v = 1; v < 15; v += 2) any = or3(any, __float_as_uint(acc[v]), __float_as_uint(acc[v + 1]));
    any |= __float_as_uint(acc[15]);
    const unsigned long long hits = __ballot((int)any < 0);
    if (hits != 0ull) {
        CASTER_STAT(12);
        unsigned int cand = 0u;
#pragma unroll
        for (int v = 0; v < 16; ++v) cand = push_sign(cand, acc[v]);
        if ((int)any < 0) {         // the lanes of `hits`: those with a bit in cand
            // position: entries of this column block in lane order behind what the queue holds (a
            // scalar address plus the lane's rank)
            const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(hits >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hits, 0u));
            unsigned int* tail = q.entries + q.count;
            // block_head: the lane's part (ray r = lane & 31, writer half) and the block; the rest is
            // the parity and slot of the column block
            tail[rank] = cand | block_head | (((unsigned int)(CB & 1) << 28) | ((unsigned int)(CB >> 1) << 29));
        }
        q.count += __popcll(hits);
    }
}

// The sphere loop: all blocks of the table against the wave's four column blocks, two blocks per
// turn so that the operand registers alternate (the next block is on its way while one is worked
// on).  The queue is worked off when the next block might not fit, at the end of a tile of blocks
// (an entry names a block within its tile) and at the end.  The table holds one block more than
// the scene needs (read ahead).
template <bool EXACT_LDS>
CDEV void mfma_block(const uint4 a0, const uint4 a1, int blk, int& base, const SceneView& sc, const float4* lds_exact,
                     const MfmaRays& mr, MfmaQueue& q, const Slot& sa, const Slot& sb, bool first_column_only, WaveTimers& tm)
{
    const h8 A0 = __builtin_bit_cast(h8, a0), A1 = __builtin_bit_cast(h8, a1);
    const int in_tile = blk & (kMfmaTileBlocks - 1);
    if (q.count > kMfmaQueueCap - 256 || (in_tile == 0 && blk != 0)) {
        drain_mfma<EXACT_LDS>(q, sc, lds_exact, base, sa, sb, tm);
        base = (blk - in_tile) * 32;
    }
    const unsigned int block_head = q.head | ((unsigned int)in_tile << 16);
    mfma_column<0>(A0, A1, mr.b00, mr.b01, block_head, q);
    if (first_column_only) return;          // wave-uniform: the other three column blocks hold no path (gather_paths)
    mfma_column<1>(A0, A1, mr.b10, mr.b11, block_head, q);
    mfma_column<2>(A0, A1, mr.b20, mr.b21, block_head, q);
    mfma_column<3>(A0, A1, mr.b30, mr.b31, block_head, q);
}

template <bool EXACT_LDS>
CDEV void isect_mfma(const SceneView& sc, const float4* lds_exact, int n_blocks, const MfmaRays& mr, MfmaQueue& q,
                     const Slot& sa, const Slot& sb, float& closestA, float& closestB, int& bestA, int& bestB,
                     WaveTimers& tm)
{
    queue_begin(q, closestA, closestB, bestA, bestB);
    const uint4* tab = sc.mfma_tab + (threadIdx.x & 63);
    uint4 a0 = tab[0], a1 = tab[64];
    int base = 0;
    int blk = 0;
#pragma unroll 1
    for (; blk + 1 < n_blocks; blk += 2) {
        const uint4 b0 = tab[(blk + 1) * 128], b1 = tab[(blk + 1) * 128 + 64];
        mfma_block<EXACT_LDS>(a0, a1, blk, base, sc, lds_exact, mr, q, sa, sb, false, tm);
        a0 = tab[(blk + 2) * 128]; a1 = tab[(blk + 2) * 128 + 64];
        mfma_block<EXACT_LDS>(b0, b1, blk + 1, base, sc, lds_exact, mr, q, sa, sb, false, tm);
    }
    if (blk < n_blocks) mfma_block<EXACT_LDS>(a0, a1, blk, base, sc, lds_exact, mr, q, sa, sb, false, tm);
    drain_mfma<EXACT_LDS>(q, sc, lds_exact, base, sa, sb, tm);
    queue_end(q, closestA, closestB, bestA, bestB);
}

// The sphere loop of a wave whose paths have been gathered into the first column block
// (gather_paths).  With a quarter of the work per block the table reads are what such a wave waits
// for, and the registers of the three idle column blocks are free: four blocks are on their way
// instead of one.  Reads past the table's last block (it holds one more than the scene needs) are
// redirected to that block.
template <bool EXACT_LDS>
CDEV void isect_mfma_gathered(const SceneView& sc, const float4* lds_exact, int n_blocks, const MfmaRays& mr, MfmaQueue& q,
                              const Slot& sa, const Slot& sb, float& closestA, float& closestB, int& bestA, int& bestB,
                              WaveTimers& tm)
{
    constexpr int kAhead = 4;
    queue_begin(q, closestA, closestB, bestA, bestB);
    const uint4* tab = sc.mfma_tab + (threadIdx.x & 63);
    uint4 p0[kAhead], p1[kAhead];
#pragma unroll
    for (int i = 0; i < kAhead; ++i) {
        const int b = i < n_blocks ? i : n_blocks;
        p0[i] = tab[b * 128]; p1[i] = tab[b * 128 + 64];
    }
    int base = 0;
#pragma unroll 1
    for (int blk = 0; blk < n_blocks; blk += kAhead) {
#pragma unroll
        for (int i = 0; i < kAhead; ++i) {
            if (blk + i < n_blocks) {                       // wave-uniform
                const uint4 a0 = p0[i], a1 = p1[i];
                const int b = blk + i + kAhead < n_blocks ? blk + i + kAhead : n_blocks;
                p0[i] = tab[b * 128]; p1[i] = tab[b * 128 + 64];
                mfma_block<EXACT_LDS>(a0, a1, blk + i, base, sc, lds_exact, mr, q, sa, sb, true, tm);
            }
        }
    }
    drain_mfma<EXACT_LDS>(q, sc, lds_exact, base, sa, sb, tm);
    queue_end(q, closestA, closestB, bestA, bestB);
}

// The spheres that are candidates for every ray (SceneDev::n_always of them) do not go through
// the lists: each lane tests them for its two slots directly.  Their exact images and list
// positions sit in LDS (stage_always).
struct AlwaysLds {
    float4 exact[kMaxAlways];
    int index[kMaxAlways];
};

CDEV void stage_always(AlwaysLds* al, const SceneDev& sd)
{
    if (threadIdx.x < kMaxAlways) {
        const int k = (int)threadIdx.x;
        const bool used = k < sd.n_always;
        al->index[k] = used ? sd.always[k] : 0;
        al->exact[k] = used ? exact_image(sd.sph[sd.always[k]]) : exact_pad();
    }
}

CDEV void test_always(const AlwaysLds* al, int n_always, const Slot& sa, const Slot& sb, float& closestA,
                      float& closestB, int& bestA, int& bestB)
{
    for (int h = 0; h < n_always; ++h) {
        const int index = al->index[h];
        const float4 e = al->exact[h];
        float t;
        if (candidate_root(e, sa.o, sa.d, sa.aa(), sa.tmin(), t)) take_hit(t, index, closestA, bestA);
        if (candidate_root(e, sb.o, sb.d, sb.aa(), sb.tmin(), t)) take_hit(t, index, closestB, bestB);
    }
}

// A ray with a non-finite component makes every sphere test "accept" in the reference (NaN passes
// the range checks, src/caster.c:107,114): it reports its last sphere with t = NaN.
CDEV void non_finite_rule(bool finite, int n_sph, float& closest, int& best)
{
    if (!finite) { closest = __builtin_nanf(""); best = n_sph - 1; }
}

// Copies the exact image of spheres [base, base+cnt) into LDS, padded to a multiple of 2*kGroup
// plus the read-ahead slack.  Returns the padded count.
CDEV int stage_spheres(float4* exact, const SceneView& sc, int base, int cnt)
{
    const int padded = (cnt + 2 * kGroup - 1) / (2 * kGroup) * (2 * kGroup);
    for (int k = threadIdx.x; k < padded + kLdsSlack; k += kBlock)
        exact[k] = (k < cnt) ? exact_image(sc.sph[base + k]) : exact_pad();
    return padded;
}

// After the sphere loop: merge the rects into the closest hit and build the
// hit record (src/caster.c:216-223, 123-128, 153-161).  False on a miss.
CDEV bool resolve_hit(const Slot& s, const Ray& r, float closest, int best_sph, const SceneView& sc, HitRec& rec)
{
    // A non-finite ray makes every test "accept" (NaN passes the range checks, src/caster.c:107,114),
    // the pad entries of the last LDS tile included; the reference would report its last sphere.
    if (best_sph >= sc.n_sph) best_sph = sc.n_sph - 1;
    int best_ord = (best_sph >= 0) ? sc.sph_meta[best_sph].x : -1;
    int best_rect = -1;
    for (int k = 0; k < sc.n_rect; ++k) {
        float t;
        if (rect_root(sc.rect_a[k], sc.rect_b[k], r, s.tmin(), closest, t)) {
            const int ord = sc.rect_meta[k].x;
            if (t < closest || ord > best_ord || best_ord < 0) { closest = t; best_ord = ord; best_rect = k; }
        }
    }
    if (best_ord < 0) return false;
    if (best_rect >= 0) {
        finish_rect(rec, r, closest, sc.rect_a[best_rect], sc.rect_b[best_rect], sc.rect_meta[best_rect].y);
    } else {
        const float4 sp = sc.sph[best_sph];
        finish_sphere(rec, r, closest, mk(sp.x, sp.y, sp.z), sp.w, sc.sph_meta[best_sph].y);
    }
    return true;
}

// A path has ended: its radiance goes to the accumulator of the wave's current tile in LDS if it
// belongs to that tile, else (the wave has moved on to another unit meanwhile) straight to the
// frame's fixed-point buffer.  Both are integer additions: the sum does not depend on the route.
CDEV void deposit(int q, V3 radiance, const TileGeom& tg, acc_t* wave_acc, const RenderArgs& a)
{
    const int i = q & 0xFFFF, j = (int)((unsigned int)q >> 16);
    const unsigned int qx = (unsigned int)(i - tg.x0), qy = (unsigned int)(j - tg.y0);
    const acc_t fr = to_fixed(radiance.x), fg = to_fixed(radiance.y), fb = to_fixed(radiance.z);
    if (qx < (unsigned int)tg.tw && qy < (unsigned int)tg.th) {
        acc_t* px = wave_acc + (qy * kTileW + qx) * 3;
        if (fr) atomicAdd(px + 0, fr);
        if (fg) atomicAdd(px + 1, fg);
        if (fb) atomicAdd(px + 2, fb);
    } else {
        const size_t plane = (size_t)a.width * (size_t)a.height;
        acc_t* px = a.fixed + ((size_t)j * a.width + i);
        if (fr) atomicAdd(px, fr);
        if (fg) atomicAdd(px + plane, fg);
        if (fb) atomicAdd(px + 2 * plane, fb);
    }
}

// One bounce of a slot's path: shade the hit and either continue the path or
// end it (adding its radiance to the frame).
// DEFER: a noise texture met by a path that goes on is left to the caller (see shade and
// marble_stage): `later` then holds the material whose colour still has to scale the throughput.
template <bool TEX, bool DEFER = false>
CDEV void bounce(Slot& s, float closest, int best_sph, const SceneView& sc, const TileGeom& tg, acc_t* wave_acc,
                 const RenderArgs& a, unsigned int& rays, int& later)
{
    later = -1;
    const unsigned long long tracing = __ballot(s.alive());
    if (tracing == 0ull) return;
    rays += (unsigned int)__popcll(tracing);     // counted per wave: the same in all lanes
    if (!s.alive()) return;
    CASTER_STAT(6);
    Ray r;
    r.o = s.o;
    r.d = s.d;
    bool cont = false;
    V3 radiance = mk(0.0f, 0.0f, 0.0f);
    HitRec rec;
    if (resolve_hit(s, r, closest, best_sph, sc, rec)) {
        Ray next;
        V3 atten, emit;
        cont = shade<TEX, DEFER>(sc, s.g, r, rec, next, atten, emit, later);
        if (cont) {
            s.thr = s.thr * atten;
            set_ray(s, next);
            s.depth--;
            if (s.depth <= 0) cont = false;          // ray_color at depth 0 is black and is not counted
        } else {
            radiance = s.thr * emit;
        }
    }
    if (!cont) {
        deposit(s.q, radiance, tg, wave_acc, a);
        park(s);
        later = -1;                                  // an ended path's throughput is read no more
    }
}

template <bool TEX>
CDEV void bounce(Slot& s, float closest, int best_sph, const SceneView& sc, const TileGeom& tg, acc_t* wave_acc,
                 const RenderArgs& a, unsigned int& rays)
{
    int later;
    bounce<TEX, false>(s, closest, best_sph, sc, tg, wave_acc, a, rays, later);
}

// ---- primary rays ----------------------------------------------------------
// The camera rays of a work unit all leave the lens disk and pass through the same 8x8-pixel patch
// of the focus plane (src/camera.c:32-42): a thin beam, which only a handful of the scene's spheres
// can touch.  They do not go through the all-pairs sphere loop.  When a wave takes a unit it bounds
// the beam once and notes the spheres the beam can touch in a list in LDS (build_tile_list: a
// superset, by the argument there); the unit's paths are then started 64 at a time, all lanes busy
// (primary_batch): camera ray, the reference's root arithmetic (candidate_root) on the directly
// tested spheres and on the list, closest hit by take_hit's rule, and the first bounce.  What the
// exact test decides is what it decides in the sphere loop -- a sphere off the list cannot be hit,
// so leaving it out changes nothing -- and the frames are bit-identical to tracing the camera rays
// through the loop.  Paths that survive their first bounce wait in a ring of the wave in device
// memory (PathRing) until a slot of the main loop is free; the loop itself only ever traces rays
// from the second on, about two thirds of all rays in the benchmark scene.
constexpr int kTileListCap = 128;           // spheres on a tile's list; beyond that the batch tests all spheres

struct TileList {
    unsigned short* idx;    // this wave's [kTileListCap] in LDS
    int count;              // wave-uniform
    bool all;               // no usable bound (degenerate frame or camera, list overflow): every sphere is tested
};

CDEV int lane_rank(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Every camera ray of the unit runs from a point a of the lens disk, |a - a0| <= rho_l around the
// camera position a0, through a point b of the tile's patch of the focus plane, |b - b0| <= rho_q
// around the patch centre b0: x(s) = a + s (b - a), and s is the ray parameter t of the reference
// (its direction is b - a, unnormalised), so s > 0.  Against the axis x0(s) = a0 + s (b0 - a0):
// |x(s) - x0(s)| <= f(s) = |1 - s| rho_l + s rho_q.  A sphere (c, r) touched at parameter s therefore
// has |x0(s) - c| <= r + f(s).  With z and dperp the axial and the perpendicular coordinate of c and
// L = |b0 - a0|, that needs |s L - z| <= r + f(s) <= (r + rho_l) + s (rho_l + rho_q), which confines
// s to [s_lo, s_hi] (empty: the sphere lies behind the lens), and dperp <= r + max f over that
// interval; f is convex, so the maximum sits at an end.  Every quantity is inflated by 1e-3 of its
// size and 1e-5 of the distances involved: orders of magnitude more than the rounding of this test
// or of the reference's, so a sphere the reference's arithmetic would report is on the list.  The
// test is written as "not provably out of reach", so a sphere with NaN data stays on the list.
CDEV void build_tile_list(TileList& tl, const Unit& un, const RenderArgs& a, const Cam& cam, const SceneView& sc, int lane)
{
    tl.count = 0;
    tl.all = true;
    if (!a.tile_lists) return;
    if (a.width < 2 || a.height < 2) return;                     // the pixel coordinates divide by zero: no beam to speak of
    const float iw = __builtin_amdgcn_rcpf((float)(a.width - 1)), ih = __builtin_amdgcn_rcpf((float)(a.height - 1));
    const float hu = 0.5f * (float)un.tg.tw * iw, hv = 0.5f * (float)un.tg.th * ih;
    const float uc = (float)un.tg.x0 * iw + hu, vc = (float)un.tg.y0 * ih + hv;
    const V3 ax = (cam.llc + (cam.h * uc + cam.v * vc)) - cam.o;
    const float L = __builtin_amdgcn_sqrtf(dot(ax, ax));
    const float rho_q = (hu * __builtin_amdgcn_sqrtf(dot(cam.h, cam.h)) + hv * __builtin_amdgcn_sqrtf(dot(cam.v, cam.v))) * 1.002f;
    const float rho_l = __builtin_fabsf(cam.lr) * 1.002f *
                        __builtin_amdgcn_sqrtf(__builtin_fmaxf(dot(cam.u0, cam.u0), dot(cam.v0, cam.v0)) + __builtin_fabsf(dot(cam.u0, cam.v0)));
    const float kk = rho_l + rho_q;
    if (!(L - kk > 1.0e-3f * L) || !(L < 1.0e30f)) return;       // a beam as wide as it is long, or no finite axis
    const float inv_l = __builtin_amdgcn_rcpf(L);
    const V3 e = ax * inv_l;
    const float inv_lo = __builtin_amdgcn_rcpf(L + kk) * 0.999f, inv_hi = __builtin_amdgcn_rcpf(L - kk) * 1.001f;
    tl.all = false;
    for (int base = 0; base < sc.n_sph; base += 64) {
        const int k = base + lane;
        bool cand = false;
        if (k < sc.n_sph) {
            const float4 sp = sc.sph[k];
            const V3 v = mk(sp.x, sp.y, sp.z) - cam.o;
            const float z = dot(v, e);
            const V3 w = v - e * z;
            const float dperp = __builtin_amdgcn_sqrtf(dot(w, w));
            const float r = __builtin_fabsf(sp.w);
            const float slack = 1.0e-5f * ((__builtin_fabsf(z) + dperp) + L);
            const float R0 = (r + rho_l) * 1.001f + slack;
            const float s_hi = (z + R0) * inv_hi;                                // only looked at when z + R0 >= 0
            const float s_lo = __builtin_fmaxf((z - R0) * inv_lo, 0.0f);
            const float f_lo = __builtin_fabsf(1.0f - s_lo) * rho_l + s_lo * rho_q;
            const float f_hi = __builtin_fabsf(1.0f - s_hi) * rho_l + s_hi * rho_q;
            const float bound = (r + __builtin_fmaxf(f_lo, f_hi)) * 1.001f + slack;
            const bool out = (z + R0 < 0.0f) || (dperp > bound);
            cand = !out;
            for (int h = 0; h < a.scene.n_always; ++h) cand = cand && (k != a.scene.always[h]);   // tested directly
        }
        const unsigned long long m = __ballot(cand);
        if (cand) {
            const int pos = tl.count + lane_rank(m);
            if (pos < kTileListCap) tl.idx[pos] = (unsigned short)k;
        }
        tl.count += __popcll(m);
    }
    if (tl.count > kTileListCap || sc.n_sph > 65535) tl.all = true;
    // the list is read back by all lanes of this wave; LDS instructions of one wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Between two bounces a slot's throughput and pixel are of no use to the sphere loop, where registers
// are scarcest: the main loop keeps them in LDS (one float4 per slot and lane) and has them in
// registers only while the slot bounces.
CDEV void cold_out(const Slot& s, float4* home) { *home = make_float4(s.thr.x, s.thr.y, s.thr.z, __int_as_float(s.q)); }
CDEV void cold_in(Slot& s, const float4* home)
{
    const float4 v = *home;
    s.thr = mk(v.x, v.y, v.z);
    s.q = __float_as_int(v.w);
}

// Paths waiting for a slot: up to kRingEntries states of this wave in device memory, word-major (the
// lanes of a push or a pop touch consecutive addresses), kept as a stack: the free slots take the
// paths started last.  The order in which paths are traced changes no result, and a stack keeps
// re-using its lowest entries -- a few KB per wave that stay in the L2 cache instead of a ring's
// whole extent travelling to memory and back.  count is the same in all lanes.  A state is written
// and read by different lanes of the same wave: the fences order the two.
struct PathRing {
    uint32_t* mem;
    int count;
};

// The words of a slot's state, in ring order: 14 in the strict mode, 15 in the fast mode.
static_assert(kRingWords >= (CASTER_STRICT ? 14 : 15), "a ring entry holds a slot's whole state");
template <typename F>
CDEV void for_each_word(Slot& s, F f)
{
    int w = 0;
    auto fl = [&](float& x) { uint32_t u = __float_as_uint(x); f(w++, u); x = __uint_as_float(u); };
    auto in = [&](int& x) { uint32_t u = (uint32_t)x; f(w++, u); x = (int)u; };
    fl(s.o.x); fl(s.o.y); fl(s.o.z); fl(s.d.x); fl(s.d.y); fl(s.d.z);
#if CASTER_STRICT
    fl(s.aa_);
#else
    fl(s.tmin_); fl(s.tmax_);
#endif
    fl(s.thr.x); fl(s.thr.y); fl(s.thr.z);
    f(w++, s.g.s0); f(w++, s.g.s1);
    in(s.q); in(s.depth);
}

CDEV void ring_push(PathRing& r, Slot& s)
{
    const unsigned long long m = __ballot(s.alive());
    if (m != 0ull) {
        if (s.alive()) {
            uint32_t* p = r.mem + (r.count + lane_rank(m));
            for_each_word(s, [&](int w, uint32_t& u) { p[w * kRingEntries] = u; });
        }
        r.count += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// The free lanes of a slot take the paths on top of the stack.
CDEV void ring_pop(PathRing& r, Slot& s, float4* home)
{
    if (r.count == 0) return;                                   // wave-uniform
    const unsigned long long want = __ballot(!s.alive());
    if (want == 0ull) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int rank = lane_rank(want);
    if (!s.alive() && rank < r.count) {
        CASTER_STAT(11);
        const uint32_t* p = r.mem + (r.count - 1 - rank);
        for_each_word(s, [&](int w, uint32_t& u) { u = p[w * kRingEntries]; });
        s.g.draws = 0;
        cold_out(s, home);
    }
    int n = __popcll(want);
    if (n > r.count) n = r.count;
    r.count -= n;
}

// Starts the next 64 paths of the unit (fewer at the end of its pool), traces their camera rays and
// shades the first hit; the survivors join the ring.  A camera ray with a non-finite component is
// handled as everywhere else (non_finite_rule).
template <bool TEX, bool EXACT_LDS>
CDEV void primary_batch(Slot& sp, Unit& cur, const TileList& tl, const RenderArgs& a, const Cam& cam, const PixelScale& ps,
                        const SceneView& sc, const float4* lds_exact, const AlwaysLds* al, acc_t* wave_acc, unsigned int& rays,
                        int lane)
{
    sp.thr = mk(1.0f, 1.0f, 1.0f);
    sp.g.s0 = sp.g.s1 = sp.g.draws = 0;
    sp.q = 0; sp.depth = 0;
    park(sp);
    const int m = cur.next + lane;
    if (m < cur.total) { CASTER_STAT(13); start_pool_path(sp, m, cur, a, cam, ps); }
    cur.next = (cur.next + 64 < cur.total) ? cur.next + 64 : cur.total;
    float closest = sp.tmax();                    // a root equal to tmax is still accepted
    int best = -1;
    for (int h = 0; h < a.scene.n_always; ++h) {
        float t;
        if (candidate_root(al->exact[h], sp.o, sp.d, sp.aa(), sp.tmin(), t)) take_hit(t, al->index[h], closest, best);
    }
    const int n = tl.all ? sc.n_sph : tl.count;
    for (int i = 0; i < n; ++i) {
        int k = i;
        if (tl.all) {
            bool direct = false;
            for (int h = 0; h < a.scene.n_always; ++h) direct |= (a.scene.always[h] == i);
            if (direct) continue;
        } else {
            k = __builtin_amdgcn_readfirstlane((int)tl.idx[i]);
        }
        CASTER_STAT(14);
        const float4 ex = EXACT_LDS ? lds_exact[k] : exact_image(sc.sph[k]);
        float t;
        if (candidate_root(ex, sp.o, sp.d, sp.aa(), sp.tmin(), t)) take_hit(t, k, closest, best);
    }
    const bool finite = __builtin_isfinite(sp.o.x) && __builtin_isfinite(sp.o.y) && __builtin_isfinite(sp.o.z) &&
                        __builtin_isfinite(sp.d.x) && __builtin_isfinite(sp.d.y) && __builtin_isfinite(sp.d.z);
    non_finite_rule(finite, sc.n_sph, closest, best);
    bounce<TEX>(sp, closest, best, sc, cur.tg, wave_acc, a, rays);
}

// The survivors of a batch go straight into the free slots of the main loop, lane to lane through the
// LDS crossbar (ds_bpermute), without a trip through device memory; only those that find no free slot
// wait on the stack.  Survivor number r (in lane order) goes to free slot number r, slot A's free lanes
// first; `scratch` (64 words of this wave in LDS, idle outside the sphere loop) maps r to the survivor's lane.
CDEV uint32_t slot_bits(float x) { return __float_as_uint(x); }
CDEV uint32_t slot_bits(uint32_t x) { return x; }
CDEV uint32_t slot_bits(int x) { return (uint32_t)x; }
CDEV void slot_put(float& x, uint32_t u) { x = __uint_as_float(u); }
CDEV void slot_put(uint32_t& x, uint32_t u) { x = u; }
CDEV void slot_put(int& x, uint32_t u) { x = (int)u; }
CDEV uint32_t from_lane(int src, uint32_t x) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)x); }

CDEV void hand_over(Slot& sp, Slot& sa, float4* homeA, Slot& sb, float4* homeB, PathRing& ring, unsigned int* scratch, int lane)
{
    const unsigned long long surv = __ballot(sp.alive());
    if (surv == 0ull) return;
    const int ns = __popcll(surv);
    const unsigned long long fa = __ballot(!sa.alive()), fb = __ballot(!sb.alive());
    const int na = __popcll(fa), nfree = na + __popcll(fb);
    const int rs = lane_rank(surv);
    if (sp.alive()) scratch[rs] = (unsigned int)lane;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // LDS instructions of one wave complete in order
    __builtin_amdgcn_wave_barrier();
    const int ra = lane_rank(fa), rb = na + lane_rank(fb);
    const bool takeA = !sa.alive() && ra < ns, takeB = !sb.alive() && rb < ns;
    const int srcA = takeA ? (int)scratch[ra] : lane;
    const int srcB = takeB ? (int)scratch[rb] : lane;
    // every lane takes part in a ds_bpermute; the lanes that take a path keep what they fetched
#define CASTER_MOVE(field)                                                          \
    {                                                                               \
        const uint32_t u = slot_bits(sp.field);                                     \
        const uint32_t va = from_lane(srcA, u), vb = from_lane(srcB, u);            \
        if (takeA) slot_put(sa.field, va);                                          \
        if (takeB) slot_put(sb.field, vb);                                          \
    }
    CASTER_MOVE(o.x) CASTER_MOVE(o.y) CASTER_MOVE(o.z) CASTER_MOVE(d.x) CASTER_MOVE(d.y) CASTER_MOVE(d.z)
#if CASTER_STRICT
    CASTER_MOVE(aa_)
#else
    CASTER_MOVE(tmin_) CASTER_MOVE(tmax_)
#endif
    CASTER_MOVE(g.s0) CASTER_MOVE(g.s1) CASTER_MOVE(depth)
#undef CASTER_MOVE
    {   // throughput and pixel live in LDS between bounces (cold_out)
        const uint32_t tx = slot_bits(sp.thr.x), ty = slot_bits(sp.thr.y), tz = slot_bits(sp.thr.z), tq = slot_bits(sp.q);
        const float4 ca = make_float4(__uint_as_float(from_lane(srcA, tx)), __uint_as_float(from_lane(srcA, ty)),
                                      __uint_as_float(from_lane(srcA, tz)), __uint_as_float(from_lane(srcA, tq)));
        const float4 cb = make_float4(__uint_as_float(from_lane(srcB, tx)), __uint_as_float(from_lane(srcB, ty)),
                                      __uint_as_float(from_lane(srcB, tz)), __uint_as_float(from_lane(srcB, tq)));
        if (takeA) *homeA = ca;
        if (takeB) *homeB = cb;
    }
    // the survivors without a slot wait on the stack
    if (rs < nfree) sp.depth = 0;
    ring_push(ring, sp);
    ring.count = __builtin_amdgcn_readfirstlane(ring.count);
}

// The noise textures the two slots left for later (bounce<TEX, true>), evaluated for both slots at
// once.  A marble texel is about a thousand vector instructions whatever the number of lanes that
// want one, and in either slot only some do: the lanes of slot A without a texel of their own take
// the texels of slot B (B's r-th texel goes to A's r-th idle lane, through `items`, 64 float4 of this
// wave in LDS), so that one pass serves both slots whenever the two together want no more than 64; what
// does not fit is evaluated by its own lane in a second pass.  A texel's value depends on its point
// and its material alone, not on the lane that evaluates it.  The point is the new ray's origin
// (the scatter functions start the next ray at the hit point, src/material.c:8,18); the throughput
// is scaled where it lives between bounces (cold_out).
CDEV void marble_stage(const SceneView& sc, const Slot& sa, int laterA, float4* homeA, const Slot& sb, int laterB, float4* homeB,
                       float4* items, int lane)
{
    const unsigned long long ma = __ballot(laterA >= 0), mb = __ballot(laterB >= 0);
    if ((ma | mb) == 0ull) return;
    const int idle = 64 - __popcll(ma), nb = __popcll(mb);
    const int moved = nb < idle ? nb : idle;                    // wave-uniform
    const int rb = lane_rank(mb), ri = lane_rank(~ma);
    float stripesB = 0.0f;
    if (laterB >= 0) stripesB = sc.mats[laterB].c1[0];
    if (laterB >= 0 && rb < moved) items[rb] = make_float4(sb.o.x, sb.o.y, sb.o.z, stripesB);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // LDS instructions of one wave complete in order
    __builtin_amdgcn_wave_barrier();
    const bool guest = laterA < 0 && ri < moved;                // this lane evaluates a texel of slot B
    float scaleA = 1.0f, scaleB = 1.0f;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        V3 p = sa.o;
        float stripes = 0.0f;
        bool busy;
        if (pass == 0) {
            busy = laterA >= 0 || guest;
            if (laterA >= 0) stripes = sc.mats[laterA].c1[0];
            if (guest) { const float4 it = items[ri]; p = mk(it.x, it.y, it.z); stripes = it.w; }
        } else {
            busy = laterB >= 0 && rb >= moved;
            p = sb.o;
            stripes = stripesB;
        }
        float v = 1.0f;
        if (busy) v = marble_scale(sc, p, stripes);
        if (pass == 0) {
            if (laterA >= 0) scaleA = v;
            if (guest) items[ri].x = v;
            if (moved == nb) break;                             // nothing left for a second pass
        } else {
            if (busy) scaleB = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (laterB >= 0 && rb < moved) scaleB = items[rb].x;
    if (laterA >= 0) {
        const Mat m = sc.mats[laterA];
        const float4 c = *homeA;
        *homeA = make_float4(c.x * (m.c0[0] * scaleA), c.y * (m.c0[1] * scaleA), c.z * (m.c0[2] * scaleA), c.w);
    }
    if (laterB >= 0) {
        const Mat m = sc.mats[laterB];
        const float4 c = *homeB;
        *homeB = make_float4(c.x * (m.c0[0] * scaleB), c.y * (m.c0[1] * scaleB), c.z * (m.c0[2] * scaleB), c.w);
    }
}

// The end of a frame: the queue is empty, nothing waits, and what a wave still holds are the few
// paths that run on towards the depth limit (in the benchmark scene one path in 200 lives past 30
// rays -- light caught inside a glass sphere -- and those hold a tenth of the slots at any time),
// scattered over its 128 slots.  A trip costs the same instructions whatever the number of live
// lanes, so once no more than 32 paths are left they are gathered into the first 32 lanes of slot A
// -- over the stack, whose pop fills the free lanes in order -- and the sphere loop runs its first
// column block only from then on: a quarter of the matrix instructions and sign scans, and one
// bounce instead of two.  Which lane traces a path changes nothing about it.
CDEV void gather_paths(Slot& sa, float4* homeA, Slot& sb, float4* homeB, PathRing& ring)
{
    cold_in(sa, homeA);
    ring_push(ring, sa);
    park(sa);
    cold_in(sb, homeB);
    ring_push(ring, sb);
    park(sb);
    ring.count = __builtin_amdgcn_readfirstlane(ring.count);
    ring_pop(ring, sa, homeA);
}

// Resident blocks per CU = waves per SIMD.  Measured on the 1080p benchmark (strict, ms per frame):
// 3 (168 VGPRs, no spills) 291, 4 (128 VGPRs) 276, 5 (96 VGPRs) 290, 6 (80 VGPRs) 314.
#ifndef CASTER_BLOCKS_PER_CU
#define CASTER_BLOCKS_PER_CU 4
#endif
// The same for the kernel of the matrix loop.
#ifndef CASTER_MATRIX_BLOCKS_PER_CU
#define CASTER_MATRIX_BLOCKS_PER_CU 4
#endif
constexpr int lds_fixed_bytes(int kind)
{
    return (int)sizeof(unsigned long long) * kWaves * kAccStride +
           kWaves * (kind >= kLoopMatrix ? kMfmaQueueBytes : 2 * kListCap * 64) +
           kWaves * kTileListCap * (int)sizeof(unsigned short) +
           kWaves * 2 * 64 * (int)sizeof(float4);
}

template <int KIND, bool TEX>
__global__ __launch_bounds__(kBlock, KIND >= kLoopMatrix ? CASTER_MATRIX_BLOCKS_PER_CU : CASTER_BLOCKS_PER_CU)
void CASTER_NAME(render_kernel_)(RenderArgs a)
{
    constexpr unsigned int kPerCU = KIND >= kLoopMatrix ? CASTER_MATRIX_BLOCKS_PER_CU : CASTER_BLOCKS_PER_CU;
    constexpr bool TILED = KIND == kLoopVectorTiled;
    constexpr bool MATRIX = KIND >= kLoopMatrix;
    extern __shared__ __align__(16) unsigned char s_dyn[];
    __shared__ AlwaysLds s_always;
    // layout: accumulators [kWaves][kAccStride] (acc_t), candidate lists (u16; [kWaves][kListCap][64], or
    // the results and the queues of MfmaQueue for the matrix loop), the tile lists of the camera rays
    // (u16 [kWaves][kTileListCap]), the slots' throughput and pixel between bounces (cold_out), the exact
    // sphere image of one tile, the noise tables
    acc_t* s_acc = reinterpret_cast<acc_t*>(s_dyn);
    unsigned short* s_cand = reinterpret_cast<unsigned short*>(s_acc + kWaves * kAccStride);
    float4* s_exact = reinterpret_cast<float4*>(s_dyn + lds_fixed_bytes(KIND));
    float4* s_cold = reinterpret_cast<float4*>(s_dyn + lds_fixed_bytes(KIND)) - kWaves * 2 * 64;       // [kWaves][2 slots][64 lanes]
    unsigned short* s_tile = reinterpret_cast<unsigned short*>(s_cold) - kWaves * kTileListCap;
    float4* s_noise_vec = s_exact + a.lds_spheres;                              // only if the scene has noise tables

    SceneView sc = view_of(a.scene);
    stage_noise(sc, a.scene, reinterpret_cast<unsigned char*>(s_noise_vec));   // read 98 times per shaded noise texel
    const Cam cam = cam_of(a.cam);
    const PixelScale ps = pixel_scale(a.width, a.height);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // the same in all lanes: per-wave LDS addresses stay scalar
    const int tiles_x = (a.width + kTileW - 1) / kTileW;
    const int tiles_y = (a.height + kTileH - 1) / kTileH;
    const int n_tiles = tiles_x * tiles_y;

    acc_t* wave_acc = s_acc + wave * kAccStride;
    for (int k = lane; k < kAccStride; k += 64) wave_acc[k] = 0;
    stage_always(&s_always, a.scene);
    int n_lds = 0;
    if (KIND == kLoopVector || KIND == kLoopMatrix) n_lds = stage_spheres(s_exact, sc, 0, sc.n_sph);
    SphereLds lds;
    lds.exact = s_exact;
    lds.filter = sc.filter_img;
    lds.count = sc.n_sph;
    const float3 ref = make_float3(a.scene.ref[0], a.scene.ref[1], a.scene.ref[2]);
    __syncthreads();

    Slot sa, sb;
    sa.thr = sb.thr = mk(1.0f, 1.0f, 1.0f);
    sa.g.s0 = sa.g.s1 = sa.g.draws = 0;
    sb.g = sa.g;
    sa.q = sb.q = 0; sa.depth = sb.depth = 0;
    park(sa);
    park(sb);
    unsigned int rays = 0;
    unsigned int trips = 0;                     // wave-uniform
    CandLists cl;
    cl.mem = s_cand + wave * kListCap * 64 + lane;
    cl.cnt = 0;
    MfmaQueue ml;
    ml.result = reinterpret_cast<unsigned long long*>(s_cand) + wave * 128;
    ml.entries = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned long long*>(s_cand) + kWaves * 128) + wave * kMfmaQueueCap;
    ml.count = 0;
    WaveTimers tm;
    tm.start();
    const unsigned long long t_begin = tm.now();

    const unsigned int age = blockIdx.x * kPerCU / gridDim.x;       // blocks are placed in launch order, kPerCU per CU
    // The unit that feeds the slots.  When its pool is empty its sums so far go to the frame buffer
    // and the wave starts on the next unit at once; the paths still in flight then belong to a tile
    // that is no longer the current one and deposit their radiance in the frame buffer themselves.
    Unit cur;
    cur.chunk = -1;
    cur.next = cur.total = 0;
    cur.cs = cur.s0 = 0;
    cur.tg.x0 = cur.tg.y0 = cur.tg.tw = cur.tg.th = 0;
    bool queue_empty = false;
    bool gathered = false;                      // wave-uniform: see gather_paths

    // Camera rays are traced apart from the loop (see "primary rays"); paths enter the slots with their
    // second ray, from the ring.
    constexpr bool kExactLds = KIND == kLoopVector || KIND == kLoopMatrix;      // the whole scene's exact image sits in LDS
    PathRing ring;
    ring.mem = a.ring + (size_t)(blockIdx.x * kWaves + (unsigned)wave) * (size_t)(kRingEntries * kRingWords);
    ring.count = 0;
    TileList tl;
    tl.idx = s_tile + wave * kTileListCap;
    tl.count = 0;
    tl.all = true;
    unsigned int batches = 0;                   // wave-uniform
    // 2 KB of this wave in LDS that are idle outside the sphere loop (its candidate queue or lists)
    unsigned int* wave_scratch = MATRIX ? ml.entries : reinterpret_cast<unsigned int*>(s_cand + wave * kListCap * 64);
    float4* homeA = s_cold + wave * 128 + lane;
    float4* homeB = homeA + 64;
    cold_out(sa, homeA);
    cold_out(sb, homeB);

    for (;;) {
        // Supply: the free slots take the paths that wait on the stack; while slots are still free the
        // current unit starts another batch of 64 paths, whose survivors go straight into the free
        // slots (hand_over) -- those that find none wait on the stack, 64 at most.  A unit whose pool is
        // empty hands its sums to the frame buffer and the wave takes the next one; its paths still
        // waiting or in flight add their radiance to the frame buffer themselves when they end (deposit).
        {
            const unsigned long long t_refill = tm.now();
            for (;;) {
                const int need = __popcll(__ballot(!sa.alive())) + __popcll(__ballot(!sb.alive()));
                if (need == 0) break;
                if (ring.count > 0) {                   // waiting paths first
                    ring_pop(ring, sa, homeA);
                    ring_pop(ring, sb, homeB);
                    continue;
                }
                if (cur.chunk >= 0 && cur.next >= cur.total) {
                    flush_unit(cur, wave_acc, a, lane);
                    cur.chunk = -1;
                    cur.tg.tw = cur.tg.th = 0;      // no current tile
                }
                if (cur.chunk < 0) {
                    if (queue_empty) break;
                    cur = fetch_unit(a, tiles_x, n_tiles, lane);
                    queue_empty = cur.chunk < 0;
                    tm.fetched(!queue_empty, cur.chunk, cur.tg.x0 / kTileW, cur.tg.y0 / kTileH);
                    if (queue_empty) break;
                    build_tile_list(tl, cur, a, cam, sc, lane);
                }
                Slot sp;
                primary_batch<TEX, kExactLds>(sp, cur, tl, a, cam, ps, sc, s_exact, &s_always, wave_acc, rays, lane);
                ++batches;
                hand_over(sp, sa, homeA, sb, homeB, ring, wave_scratch, lane);
            }
            tm.add(WaveTimers::kRefill, t_refill);
        }

        const bool done = queue_empty && cur.chunk < 0 && ring.count == 0;
        if (MATRIX && done && !gathered) {
            const int left = __popcll(__ballot(sa.alive())) + __popcll(__ballot(sb.alive()));
            if (left <= 32) {
                if (left > 0) gather_paths(sa, homeA, sb, homeB, ring);
                gathered = true;
                tm.mark();
            }
        }
        const int live = (sa.alive() || sb.alive()) ? 1 : 0;
        if (TILED) {
            // the tile loop below has block-wide barriers: all four waves leave together
            if (__syncthreads_or((live || !done) ? 1 : 0) == 0) break;
        } else {
            if (__any(live) == 0) {
                if (done) break;
                continue;           // nothing to trace; units still have to be retired or fetched
            }
        }

        // The SIMD issues from its oldest wave first, and these waves hardly ever stall: left alone,
        // the youngest of the four waves on a SIMD runs several times slower than the oldest and
        // its units finish last.  The issue priority goes round instead, by the clock (a turn is
        // 2^20 cycles, a few trips) so that the four waves of a SIMD, which differ in age, hold
        // four different levels at any time.  (Going round by the trip count left ties, which the
        // hardware again breaks by age: the youngest wave still ran 1.7 times slower.)
        switch (((unsigned int)(__builtin_readcyclecounter() >> 20) + age) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
        CASTER_STAT(0);
        tm.trip();
        ++trips;
        const unsigned long long t_isect = tm.now();
        float closestA = sa.tmax(), closestB = sb.tmax();       // a root equal to tmax is still accepted
        int bestA = -1, bestB = -1;
        test_always(&s_always, a.scene.n_always, sa, sb, closestA, closestB, bestA, bestB);
        if (MATRIX) {
            const MfmaRays mr = pack_rays_mfma(sa, sb, ref, a.scene.mfma_scale);
            if (gathered) isect_mfma_gathered<KIND == kLoopMatrix>(sc, s_exact, a.scene.mfma_blocks, mr, ml, sa, sb, closestA, closestB, bestA, bestB, tm);
            else isect_mfma<KIND == kLoopMatrix>(sc, s_exact, a.scene.mfma_blocks, mr, ml, sa, sb, closestA, closestB, bestA, bestB, tm);
            non_finite_rule(mr.finiteA, sc.n_sph, closestA, bestA);
            non_finite_rule(mr.finiteB, sc.n_sph, closestB, bestB);
        } else {
            const PairRays pr = pack_rays(sa, sb, ref);
            if (!TILED) {
                isect_tile(lds, n_lds, 0, pr, cl, sa, sb, closestA, closestB, bestA, bestB, tm);
            } else {
                for (int base = 0; base < sc.n_sph; base += kSphTile) {
                    const int cnt = (sc.n_sph - base < kSphTile) ? sc.n_sph - base : kSphTile;
                    if (base) __syncthreads();           // everyone is done with the previous tile
                    const int padded = stage_spheres(s_exact, sc, base, cnt);
                    lds.count = cnt;
                    __syncthreads();
                    isect_tile(lds, padded, base, pr, cl, sa, sb, closestA, closestB, bestA, bestB, tm);
                }
            }
            non_finite_rule(pr.finiteA, sc.n_sph, closestA, bestA);
            non_finite_rule(pr.finiteB, sc.n_sph, closestB, bestB);
        }
        tm.add(WaveTimers::kIntersect, t_isect);

        // Both slots bounce.  The code is the same for the two (one source, two inlined copies), and
        // the tests hold a path to the same bits in either slot, in both arithmetic modes.
        const unsigned long long t_bounce = tm.now();
        int laterA, laterB;
        cold_in(sa, homeA);
        bounce<TEX, TEX>(sa, closestA, bestA, sc, cur.tg, wave_acc, a, rays, laterA);
        cold_out(sa, homeA);
        cold_in(sb, homeB);
        bounce<TEX, TEX>(sb, closestB, bestB, sc, cur.tg, wave_acc, a, rays, laterB);
        cold_out(sb, homeB);
        if (TEX) marble_stage(sc, sa, laterA, homeA, sb, laterB, homeB, reinterpret_cast<float4*>(wave_scratch), lane);
        tm.add(WaveTimers::kBounce, t_bounce);
    }
    tm.add(WaveTimers::kTotal, t_begin);

    const unsigned int wrays = rays;            // bounce counts for the wave
    if (lane == 0 && wrays) atomicAdd(a.rays, (unsigned long long)wrays);
    if (lane == 0 && trips) atomicAdd(a.trips, (unsigned long long)trips);
    if (lane == 0 && batches) atomicAdd(a.batches, (unsigned long long)batches);
    tm.report(wave, wrays);
}

// Sphere slots of the exact image in LDS: one tile, padded, plus the read-ahead slack.
static inline int lds_sphere_slots(int n_sph)
{
    const int tile = n_sph < kSphTile ? n_sph : kSphTile;
    return (tile + 2 * kGroup - 1) / (2 * kGroup) * (2 * kGroup) + kLdsSlack;
}

hipError_t CASTER_NAME(launch_render_)(const RenderArgs& a_in, int resident_blocks, hipStream_t st)
{
    RenderArgs a = a_in;
    const int tiles_x = (a.width + kTileW - 1) / kTileW;
    const int tiles_y = (a.height + kTileH - 1) / kTileH;
    const long long units = (long long)tiles_x * tiles_y * a.n_chunks;
    if (units <= 0 || units > 0x7fffffffLL) return hipErrorInvalidValue;
    a.n_units = (int)units;
    const int kind = pick_loop_kind(a.scene, CASTER_STRICT);
    a.lds_spheres = kind == kLoopMatrixLarge ? 0 : lds_sphere_slots(a.scene.n_sph);
    const size_t lds_bytes = lds_fixed_bytes(kind) + sizeof(float4) * (size_t)a.lds_spheres +
                             (a.scene.noise_vec ? kNoiseLdsBytes : 0);
    long long blocks = (units + kWaves - 1) / kWaves;
    // resident_blocks counts CASTER_BLOCKS_PER_CU per CU
    if (kind >= kLoopMatrix) resident_blocks = resident_blocks / CASTER_BLOCKS_PER_CU * CASTER_MATRIX_BLOCKS_PER_CU;
    if (blocks > resident_blocks) blocks = resident_blocks;      // persistent: the waves loop over the queue
    if (!a.ring || blocks * kWaves > a.ring_waves) return hipErrorInvalidValue;     // every wave needs its ring
    const dim3 grid((unsigned)blocks), block(kBlock);
    // A scene whose every texture is a solid colour runs instances without the texture code (the
    // benchmark's case).  The choice depends on the scene alone, never on the loop: the two loops of a
    // scene must run the same shading code to give the same bits in the fast mode, where the
    // compiler's choice of fused operations may differ between instances.
#define CASTER_LAUNCH_KIND(K)                                                                                            \
    do {                                                                                                                 \
        if (a.scene.all_solid) hipLaunchKernelGGL((CASTER_NAME(render_kernel_)<K, false>), grid, block, lds_bytes, st, a); \
        else hipLaunchKernelGGL((CASTER_NAME(render_kernel_)<K, true>), grid, block, lds_bytes, st, a);                   \
    } while (0)
    switch (kind) {
    case kLoopMatrix: CASTER_LAUNCH_KIND(kLoopMatrix); break;
    case kLoopMatrixLarge: CASTER_LAUNCH_KIND(kLoopMatrixLarge); break;
    case kLoopVectorTiled: CASTER_LAUNCH_KIND(kLoopVectorTiled); break;
    default: CASTER_LAUNCH_KIND(kLoopVector); break;
    }
#undef CASTER_LAUNCH_KIND
    return hipGetLastError();
}

// hit_list_hit_all for an array of rays through the render kernel's own
// machinery: LDS staging, candidate filter, packed two-slot sphere loop, rect
// merge.  Lane t of a block traces rays 2t and 2t+1 of the block's 512.
__global__ __launch_bounds__(kBlock, 4)
void CASTER_NAME(k_test_trace_)(SceneDev sd, int n, const float* ray6, int32_t* hit, float* t, float* p3,
                                float* n3, int32_t* face, int32_t* mat)
{
    __shared__ float4 s_exact[kSphTile + 2 * kLdsSlack];
    __shared__ __align__(8) unsigned short s_cand[kWaves * kMfmaQueueBytes / 2];      // the larger of the two loops' lists
    __shared__ AlwaysLds s_always;
    stage_always(&s_always, sd);
    __syncthreads();
    const SceneView sc = view_of(sd);
    SphereLds lds;
    lds.exact = s_exact;
    lds.filter = sc.filter_img;
    lds.count = 0;
    const float3 ref = make_float3(sd.ref[0], sd.ref[1], sd.ref[2]);
    const int i0 = (blockIdx.x * kBlock + threadIdx.x) * 2;
    Slot sl[2];
    Ray rr[2];
    float len[2];
    for (int k = 0; k < 2; ++k) {
        park(sl[k]);
        len[k] = 1.0f;
        rr[k].o = mk(0, 0, 0); rr[k].d = mk(0, 0, 1);
        if (i0 + k < n) {
            const float* q = ray6 + 6 * (size_t)(i0 + k);
            rr[k].o = mk(q[0], q[1], q[2]);
            rr[k].d = mk(q[3], q[4], q[5]);
            set_ray(sl[k], rr[k]);
            sl[k].depth = 1;
#if !CASTER_STRICT
            len[k] = sl[k].tmin() / kTMin;
#endif
        }
    }
    float closest[2] = {sl[0].tmax(), sl[1].tmax()};
    int best[2] = {-1, -1};
    const PairRays pr = pack_rays(sl[0], sl[1], ref);
    CandLists cl;
    cl.mem = s_cand + (threadIdx.x >> 6) * kListCap * 64 + (threadIdx.x & 63);
    cl.cnt = 0;
    WaveTimers tm;
    test_always(&s_always, sd.n_always, sl[0], sl[1], closest[0], closest[1], best[0], best[1]);
    const bool matrix = sd.mfma_blocks > 0 && sc.mfma_tab != nullptr;
    if (matrix) {
        if (sc.n_sph <= kSphTile) stage_spheres(s_exact, sc, 0, sc.n_sph);
        __syncthreads();
        MfmaQueue ml;
        ml.result = reinterpret_cast<unsigned long long*>(s_cand) + (threadIdx.x >> 6) * 128;
        ml.entries = reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned long long*>(s_cand) + kWaves * 128) + (threadIdx.x >> 6) * kMfmaQueueCap;
        ml.count = 0;
        const MfmaRays mr = pack_rays_mfma(sl[0], sl[1], ref, sd.mfma_scale);
        if (sc.n_sph <= kSphTile)
            isect_mfma<true>(sc, s_exact, sd.mfma_blocks, mr, ml, sl[0], sl[1], closest[0], closest[1], best[0], best[1], tm);
        else
            isect_mfma<false>(sc, s_exact, sd.mfma_blocks, mr, ml, sl[0], sl[1], closest[0], closest[1], best[0], best[1], tm);
    }
    for (int base = 0; base < (matrix ? 0 : sc.n_sph); base += kSphTile) {
        const int cnt = (sc.n_sph - base < kSphTile) ? sc.n_sph - base : kSphTile;
        __syncthreads();
        const int padded = stage_spheres(s_exact, sc, base, cnt);
        lds.count = cnt;
        __syncthreads();
        isect_tile(lds, padded, base, pr, cl, sl[0], sl[1], closest[0], closest[1], best[0], best[1], tm);
    }
    non_finite_rule(pr.finiteA, sc.n_sph, closest[0], best[0]);
    non_finite_rule(pr.finiteB, sc.n_sph, closest[1], best[1]);
    for (int k = 0; k < 2; ++k) {
        if (i0 + k >= n) continue;
        const int i = i0 + k;
        HitRec rec;
        rec.t = 0; rec.f = 0; rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.m = -1;
        Ray r;
        r.o = sl[k].o; r.d = sl[k].d;
        const bool ok = resolve_hit(sl[k], r, closest[k], best[k], sc, rec);
        hit[i] = ok; t[i] = ok ? rec.t / len[k] : 0.0f; face[i] = ok ? rec.f : 0; mat[i] = ok ? rec.m : -1;
        p3[3 * i] = rec.p.x; p3[3 * i + 1] = rec.p.y; p3[3 * i + 2] = rec.p.z;
        n3[3 * i] = rec.n.x; n3[3 * i + 1] = rec.n.y; n3[3 * i + 2] = rec.n.z;
    }
}

hipError_t CASTER_NAME(launch_test_trace_)(const SceneDev& sc, int n, const float* ray6, int32_t* hit, float* t,
                                           float* p3, float* n3, int32_t* face, int32_t* mat, hipStream_t st)
{
    const int per_block = kBlock * 2;
    hipLaunchKernelGGL(CASTER_NAME(k_test_trace_), dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), 0,
                       st, sc, n, ray6, hit, t, p3, n3, face, mat);
    return hipGetLastError();
}

// ---- filter audit ----------------------------------------------------------
// The conservative sphere filter must never reject a pair the exact test accepts: a false
// negative silently drops a hit.  This kernel takes that from belief to a count.  For every ray
// and every sphere that goes through the filter (not the directly tested ones) it runs the exact
// test (candidate_root: the reference's root arithmetic) and evaluates the filter exactly as the
// render kernel does -- the packed vector form, or the real matrix instructions on the real
// operand table -- and reports per ray: pairs the exact test accepts, accepted pairs the filter
// rejected (must be 0), and the smallest slack (key - threshold) / (|o'|^2 + |c'|^2 + r^2) over the
// accepted pairs: a dimensionless number to hold against the filter's margin g -- the filter
// errs by (g - slack) on a grazing pair, and fails when that reaches g.
CDEV int ordered_bits(float x) { const int i = __float_as_int(x); return i < 0 ? (i ^ 0x7fffffff) : i; }
CDEV float ordered_float(int i) { return __int_as_float(i < 0 ? (i ^ 0x7fffffff) : i); }

struct AuditLds {
    unsigned int accepted[kWaves][128];
    unsigned int missed[kWaves][128];
    int slack[kWaves][128];             // ordered_bits of the smallest slack
};

CDEV bool audit_listed(const SceneDev& sd, int k)
{
    bool direct = false;
    for (int h = 0; h < sd.n_always; ++h) direct |= (sd.always[h] == k);
    return k < sd.n_sph && !direct;
}

CDEV void audit_note(AuditLds* al, int wave, int ray, bool candidate, float slack)
{
    atomicAdd(&al->accepted[wave][ray], 1u);
    if (!candidate) atomicAdd(&al->missed[wave][ray], 1u);
    atomicMin(&al->slack[wave][ray], ordered_bits(slack));
}

// One column block of the wave against one sphere block, audited.  CB as in mfma_column.
template <int CB>
CDEV void audit_column(const h8 a0, const h8 a1, const h8 b0, const h8 b1, int blk, const SceneDev& sd, const SceneView& sc,
                       const Slot& s, bool valid, float oo_scaled, const float3 ref, AuditLds* al, int wave)
{
    f16x acc = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
    const int lane = threadIdx.x & 63;
    const int owner = 32 * (CB & 1) + (lane & 31);
    V3 o, d;
    o.x = fetch(owner, s.o.x); o.y = fetch(owner, s.o.y); o.z = fetch(owner, s.o.z);
    d.x = fetch(owner, s.d.x); d.y = fetch(owner, s.d.y); d.z = fetch(owner, s.d.z);
    const float aa = fetch(owner, s.aa()), tmin = fetch(owner, s.tmin());
    const float oo = fetch(owner, oo_scaled);
    const bool ok = fetch(owner, valid ? 1.0f : 0.0f) != 0.0f;
    const int ray = (CB >> 1) * 64 + owner;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int k = blk * 32 + (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
        if (!ok || !audit_listed(sd, k)) continue;
        const float4 sp = sc.sph[k];
        float t;
        if (!candidate_root(exact_image(sp), o, d, aa, tmin, t)) continue;
        const float cx = (sp.x - ref.x) * sd.mfma_scale, cy = (sp.y - ref.y) * sd.mfma_scale, cz = (sp.z - ref.z) * sd.mfma_scale;
        const float rs = sp.w * sd.mfma_scale;
        const float scale = oo + ((cx * cx + cy * cy + cz * cz) + rs * rs);
        // the accumulator holds threshold - key: a set sign bit is a candidate
        audit_note(al, wave, ray, (__float_as_int(acc[v]) < 0), -acc[v] / scale);
    }
}

__global__ __launch_bounds__(kBlock, 2)
void CASTER_NAME(k_filter_audit_)(SceneDev sd, int n, const float* ray6, uint32_t* accepted, uint32_t* missed, float* slack)
{
    __shared__ AuditLds s_audit;
    const SceneView sc = view_of(sd);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    for (int k = lane; k < 128; k += 64) {
        s_audit.accepted[wave][k] = 0u;
        s_audit.missed[wave][k] = 0u;
        s_audit.slack[wave][k] = ordered_bits(__builtin_inff());
    }
    __syncthreads();
    const float3 ref = make_float3(sd.ref[0], sd.ref[1], sd.ref[2]);
    const int i0 = (blockIdx.x * kBlock + threadIdx.x) * 2;
    Slot sl[2];
    for (int k = 0; k < 2; ++k) {
        park(sl[k]);
        if (i0 + k < n) {
            const float* q = ray6 + 6 * (size_t)(i0 + k);
            Ray r;
            r.o = mk(q[0], q[1], q[2]);
            r.d = mk(q[3], q[4], q[5]);
            set_ray(sl[k], r);
            sl[k].depth = 1;
        }
    }
    const bool matrix = sd.mfma_blocks > 0 && sc.mfma_tab != nullptr;
    if (matrix) {
        const MfmaRays mr = pack_rays_mfma(sl[0], sl[1], ref, sd.mfma_scale);
        float oo[2];
        for (int k = 0; k < 2; ++k) {
            const float qx = (sl[k].o.x - ref.x) * sd.mfma_scale, qy = (sl[k].o.y - ref.y) * sd.mfma_scale,
                        qz = (sl[k].o.z - ref.z) * sd.mfma_scale;
            oo[k] = qx * qx + qy * qy + qz * qz;
        }
        const bool validA = sl[0].alive() && mr.finiteA, validB = sl[1].alive() && mr.finiteB;
        const uint4* tab = sc.mfma_tab + lane;
#pragma unroll 1
        for (int blk = 0; blk < sd.mfma_blocks; ++blk) {
            const h8 A0 = __builtin_bit_cast(h8, tab[blk * 128]), A1 = __builtin_bit_cast(h8, tab[blk * 128 + 64]);
            audit_column<0>(A0, A1, mr.b00, mr.b01, blk, sd, sc, sl[0], validA, oo[0], ref, &s_audit, wave);
            audit_column<1>(A0, A1, mr.b10, mr.b11, blk, sd, sc, sl[0], validA, oo[0], ref, &s_audit, wave);
            audit_column<2>(A0, A1, mr.b20, mr.b21, blk, sd, sc, sl[1], validB, oo[1], ref, &s_audit, wave);
            audit_column<3>(A0, A1, mr.b30, mr.b31, blk, sd, sc, sl[1], validB, oo[1], ref, &s_audit, wave);
        }
    } else {
        const PairRays pr = pack_rays(sl[0], sl[1], ref);
        const f2 qx = f2{sl[0].o.x, sl[1].o.x} - splat(ref.x);
        const f2 qy = f2{sl[0].o.y, sl[1].o.y} - splat(ref.y);
        const f2 qz = f2{sl[0].o.z, sl[1].o.z} - splat(ref.z);
        const f2 oo = qx * qx + qy * qy + qz * qz;
        filter_ptr src = (filter_ptr)sc.filter_img;
#pragma unroll 1
        for (int k = 0; k < sc.n_sph; ++k) {
            if (!audit_listed(sd, k)) continue;                 // wave-uniform
            const f2 key = candidate_key(pr, entry(src, k));
            const float4 sp = sc.sph[k];
            const float cx = sp.x - ref.x, cy = sp.y - ref.y, cz = sp.z - ref.z;
            const float cc = (cx * cx + cy * cy + cz * cz) + sp.w * sp.w;
            float t;
            if (sl[0].alive() && pr.finiteA && candidate_root(exact_image(sp), sl[0].o, sl[0].d, sl[0].aa(), sl[0].tmin(), t))
                audit_note(&s_audit, wave, lane, key.x >= pr.thrA, (key.x - pr.thrA) / (oo.x + cc));
            if (sl[1].alive() && pr.finiteB && candidate_root(exact_image(sp), sl[1].o, sl[1].d, sl[1].aa(), sl[1].tmin(), t))
                audit_note(&s_audit, wave, 64 + lane, key.y >= pr.thrB, (key.y - pr.thrB) / (oo.y + cc));
        }
    }
    __syncthreads();
    for (int k = 0; k < 2; ++k) {
        if (i0 + k >= n) continue;
        accepted[i0 + k] = s_audit.accepted[wave][k * 64 + lane];
        missed[i0 + k] = s_audit.missed[wave][k * 64 + lane];
        slack[i0 + k] = ordered_float(s_audit.slack[wave][k * 64 + lane]);
    }
}

hipError_t CASTER_NAME(launch_filter_audit_)(const SceneDev& sc, int n, const float* ray6, uint32_t* accepted,
                                             uint32_t* missed, float* slack, hipStream_t st)
{
    const int per_block = kBlock * 2;
    hipLaunchKernelGGL(CASTER_NAME(k_filter_audit_), dim3((unsigned)((n + per_block - 1) / per_block)), dim3(kBlock), 0,
                       st, sc, n, ray6, accepted, missed, slack);
    return hipGetLastError();
}

}  // namespace caster
