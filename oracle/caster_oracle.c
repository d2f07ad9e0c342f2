/*
 * caster_oracle.c -- CPU restatement of CRaytracer's per-pixel path-tracing path.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see caster_oracle.h for both).
 * Every function names the reference lines it restates.  Expressions keep the
 * reference's association order and are evaluated in binary32 without FMA
 * (compile with -ffp-contract=off, no -march flags that enable FMA).
 */
#include "caster_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PI 3.141592654f /* inc/common.h:10 */

/* ------------------------------------------------------------------ RNG -- */
/*
 * The reference draws from libc rand() (src/common.c:10-13), seeded from the
 * wall clock, so its images are not reproducible.  Mode ORC_RNG_COUNTER is the
 * replacement the HIP path shares: one stream per path on the 64-bit xoroshiro
 * engine (two 32-bit words) with the sum of the words as output, seeded by two
 * chained 32-bit avalanche hashes of (seed, pixel, sample); one draw is the top
 * 23 bits of the output, so random_f32 = n / 2^23 lies in [0, 1) and is exact in
 * binary32.
 */
static uint32_t orc_mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

void orc_rng_seed(orng* g, uint32_t seed, uint32_t pixel, uint32_t sample)
{
    const uint32_t a = orc_mix32(pixel ^ orc_mix32(seed + 0x9E3779B9u));
    const uint32_t b = orc_mix32(sample + a);
    g->s[0] = a; g->s[1] = b | 1u;
    g->draws = 0;
    g->mode = ORC_RNG_COUNTER;
}

void orc_rng_libc(orng* g)
{
    g->s[0] = g->s[1] = 0;
    g->draws = 0;
    g->mode = ORC_RNG_LIBC;
}

static uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

uint32_t orc_rng_next23(orng* g)
{
    uint32_t* s = g->s;
    const uint32_t result = s[0] + s[1];
    const uint32_t x = s[1] ^ s[0];
    s[0] = rotl32(s[0], 26) ^ x ^ (x << 9);
    s[1] = rotl32(x, 13);
    return result >> 9;
}

/* src/common.c:10-13 */
float orc_random_f32(orng* g)
{
    g->draws++;
    if (g->mode == ORC_RNG_LIBC)
        return rand() / (RAND_MAX + 1.0f);
    return orc_rng_next23(g) / 8388608.0f;
}

/* src/common.c:15-18 */
float orc_random_range_f32(orng* g, float lo, float hi)
{
    return lo + (hi - lo) * orc_random_f32(g);
}

/* src/common.c:25-29 */
float orc_clamp(float x, float lo, float hi)
{
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}

/* ----------------------------------------------------------- vector ops -- */
static ov3 v3(float x, float y, float z) { ov3 r; r.x = x; r.y = y; r.z = z; return r; }
static ov3 v3_add(ov3 a, ov3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }     /* common.c:45-53 */
static ov3 v3_sub(ov3 a, ov3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }     /* common.c:55-63 */
static ov3 v3_mul(ov3 a, ov3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }     /* common.c:65-72 */
static ov3 v3_smul(ov3 v, float s) { return v3(v.x * s, v.y * s, v.z * s); }        /* common.c:33 */
static ov3 v3_sdiv(ov3 v, float s) { return v3(v.x / s, v.y / s, v.z / s); }        /* common.c:34 */
static ov3 v3_neg(ov3 v) { return v3(-v.x, -v.y, -v.z); }                           /* common.c:36-43 */

static ov3 v3_cross(ov3 a, ov3 b)                                                   /* common.c:74-82 */
{
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* src/common.c:84-87: (x*x' + y*y') + z*z', left to right */
float orc_dot(ov3 a, ov3 b)
{
    return (a.x * b.x + a.y * b.y + a.z * b.z);
}

/* src/common.c:89-97: three divisions by the length */
ov3 orc_to_unit(ov3 v)
{
    return v3_sdiv(v, sqrtf(orc_dot(v, v)));
}

/* src/common.c:108-126: x, y, z drawn by three sequenced statements */
ov3 orc_random_unit_sphere(orng* g)
{
    for (;;) {
        ov3 pt;
        pt.x = orc_random_range_f32(g, -1, 1);
        pt.y = orc_random_range_f32(g, -1, 1);
        pt.z = orc_random_range_f32(g, -1, 1);
        if (orc_dot(pt, pt) < 1)
            return pt;
    }
}

/* src/common.c:128-136.  The reference writes x and y in one initializer list,
 * whose evaluation order C leaves unspecified; gcc evaluates left to right, so
 * x is drawn first here. */
ov3 orc_random_unit_disk(orng* g)
{
    for (;;) {
        ov3 pt;
        pt.x = orc_random_range_f32(g, -1, 1);
        pt.y = orc_random_range_f32(g, -1, 1);
        pt.z = 0;
        if (orc_dot(pt, pt) < 1)
            return pt;
    }
}

/* src/common.c:138-141 */
ov3 orc_random_unit_norm(orng* g)
{
    return orc_to_unit(orc_random_unit_sphere(g));
}

/* src/common.c:152-156: fabs() promotes to double; the compare is value-exact */
int orc_near_zero(ov3 v)
{
    const float s = 1E-8f;
    return (fabs(v.x) < s) && (fabs(v.y) < s) && (fabs(v.z) < s);
}

/* src/common.c:158-161 */
ov3 orc_reflect(ov3 v, ov3 n)
{
    return v3_sub(v, v3_smul(n, 2 * orc_dot(v, n)));
}

/* src/common.c:163-168 */
ov3 orc_refract(ov3 unit_v, ov3 n, float ratio, float cos_theta)
{
    ov3 orto = v3_smul(v3_add(unit_v, v3_smul(n, cos_theta)), ratio);
    ov3 para = v3_smul(n, -sqrtf(fabsf(1.0f - orc_dot(orto, orto))));
    return v3_add(orto, para);
}

/* src/common.c:170-175 */
float orc_reflectance(float cosine, float ratio)
{
    float r0 = (1 - ratio) / (1 + ratio);
    r0 = r0 * r0;
    return r0 + (1 - r0) * powf(1 - cosine, 5);
}

/* ---------------------------------------------------------- intersection -- */
/* src/caster.c:5-8 */
ov3 orc_ray_at(const oray* r, float t)
{
    return v3_add(r->o, v3_smul(r->d, t));
}

/* src/caster.c:10-14 */
static void set_face(ohit* rec, const oray* r, ov3 outn)
{
    rec->f = orc_dot(r->d, outn) < 0.0f;
    rec->n = rec->f ? outn : v3_neg(outn);
}

/* src/material.c:45-56 (sphere case) */
ov2 orc_get_uv_coord_sphere(ov3 p)
{
    ov2 uv;
    uv.u = (atan2f(-p.z, p.x) + ORC_PI) / (2 * ORC_PI);
    uv.v = acosf(-p.y) / ORC_PI;
    return uv;
}

/* src/caster.c:99-130 */
int orc_hit_sphere(const oobj* s, const oray* r, float tmin, float tmax, ohit* rec)
{
    ov3 center = v3(s->g[0], s->g[1], s->g[2]);
    float radius = s->g[3];
    ov3 oc = v3_sub(r->o, center);
    float a = orc_dot(r->d, r->d);
    float hb = orc_dot(oc, r->d);
    float c = orc_dot(oc, oc) - radius * radius;
    float d = hb * hb - a * c;

    if (d < 0)
        return 0;

    float sqrt_d = sqrtf(d);
    float root = (-hb - sqrt_d) / a;
    if (root < tmin || root > tmax) {
        root = (-hb + sqrt_d) / a;
        if (root < tmin || root > tmax)
            return 0;
    }

    rec->t = root;
    rec->p = orc_ray_at(r, root);
    ov3 outn = v3_sdiv(v3_sub(rec->p, center), radius);
    set_face(rec, r, outn);
    rec->uv = orc_get_uv_coord_sphere(outn);
    rec->m = s->material;
    return 1;
}

/* src/caster.c:132-164 */
int orc_hit_xy_rect(const oobj* rt, const oray* r, float tmin, float tmax, ohit* rec)
{
    float cx = rt->g[0], cy = rt->g[1], cz = rt->g[2], du = rt->g[3], dv = rt->g[4];
    float t = (cz - r->o.z) / r->d.z;

    if (t < tmin || t > tmax)
        return 0;

    float x = r->o.x + t * r->d.x;
    float y = r->o.y + t * r->d.y;

    float half_du = du / 2;
    float half_dv = dv / 2;

    float x0 = cx - half_du;
    float x1 = cx + half_du;
    float y0 = cy - half_dv;
    float y1 = cy + half_dv;

    if (x < x0 || x > x1 || y < y0 || y > y1)
        return 0;

    rec->uv.u = (x - x0) / du;
    rec->uv.v = (y - y0) / dv;
    rec->t = t;

    set_face(rec, r, v3(0, 0, 1));
    rec->m = rt->material;
    rec->p = orc_ray_at(r, t);
    return 1;
}

/* src/caster.c:198-236: every object is tested, in list order, against the
 * closest t so far; an equal t is accepted, so a tie goes to the later object. */
int orc_hit_list_hit_all(const oscene* sc, const oray* r, float tmin, float tmax, ohit* rec)
{
    ohit tmp;
    int hit_smt = 0;
    float closest = tmax;

    for (int32_t i = 0; i < sc->count; i++) {
        const oobj* ob = &sc->objs[i];
        switch (ob->type) {
        case ORC_HTYPE_SPHERE:
            if (orc_hit_sphere(ob, r, tmin, closest, &tmp)) {
                hit_smt = 1;
                closest = tmp.t;
                *rec = tmp;
            }
            break;
        case ORC_HTYPE_RECT_XY:
            if (orc_hit_xy_rect(ob, r, tmin, closest, &tmp)) {
                hit_smt = 1;
                closest = tmp.t;
                *rec = tmp;
            }
            break;
        }
    }
    return hit_smt;
}

/* -------------------------------------------------------------- materials -- */
/* src/material.c:3-12 */
int orc_scatter_lambertian(const oscene* sc, orng* g, const oray* r, const ohit* rec, oray* out)
{
    (void)sc; (void)r;
    out->o = rec->p;
    out->d = v3_add(rec->n, orc_random_unit_norm(g));
    if (orc_near_zero(out->d))
        out->d = rec->n;
    return 1;
}

/* src/material.c:14-19: the unit-sphere sample is drawn even when fuzziness is 0 */
int orc_scatter_metal(const oscene* sc, orng* g, const oray* r, const ohit* rec, oray* out)
{
    out->o = rec->p;
    ov3 refl = orc_reflect(orc_to_unit(r->d), rec->n);
    ov3 fuzz = v3_smul(orc_random_unit_sphere(g), sc->mats[rec->m].fuzziness);
    out->d = v3_add(refl, fuzz);
    return orc_dot(out->d, rec->n) > 0;
}

/* src/material.c:21-43: random_f32() is reached only when not totally reflecting */
int orc_scatter_dielectric(const oscene* sc, orng* g, const oray* r, const ohit* rec, oray* out)
{
    float n_refract = sc->mats[rec->m].n_refract;
    float rratio = rec->f ? (1.0f / n_refract) : n_refract;
    out->o = rec->p;

    ov3 udir = orc_to_unit(r->d);
    ov3 iudir = v3_neg(orc_to_unit(r->d));

    float cos_theta = fminf(orc_dot(iudir, rec->n), 1.0);
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);

    int not_refract = rratio * sin_theta > 1.0f;

    if (not_refract || orc_reflectance(cos_theta, rratio) > orc_random_f32(g))
        out->d = orc_reflect(udir, rec->n);
    else
        out->d = orc_refract(udir, rec->n, rratio, cos_theta);
    return 1;
}

/* src/material.c:58-101 */
/* Gradient noise at p (extension, see onoise in the header): the eight lattice corners around p
 * each hold a unit vector chosen by xor-ing three permutation tables; the result interpolates
 * dot(vector, offset from the corner) with Hermite-smoothed weights.  Zero at lattice points. */
float orc_noise_value(const onoise* n, ov3 p)
{
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = u * u * (3 - 2 * u);
    float vv = v * v * (3 - 2 * v);
    float ww = w * w * (3 - 2 * w);
    float accum = 0.0f;
    for (int di = 0; di < 2; di++)
        for (int dj = 0; dj < 2; dj++)
            for (int dk = 0; dk < 2; dk++) {
                const float* c = n->ranvec[n->perm_x[(i + di) & 255] ^ n->perm_y[(j + dj) & 255] ^ n->perm_z[(k + dk) & 255]];
                float wx = u - di, wy = v - dj, wz = w - dk;
                float d = c[0] * wx + c[1] * wy + c[2] * wz;
                accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) * (dk * ww + (1 - dk) * (1 - ww)) * d;
            }
    return accum;
}

float orc_noise_turb(const onoise* n, ov3 p, int depth)
{
    float accum = 0.0f;
    ov3 temp = p;
    float weight = 1.0f;
    for (int k = 0; k < depth; k++) {
        accum += weight * orc_noise_value(n, temp);
        weight *= 0.5f;
        temp = v3_smul(temp, 2.0f);
    }
    return fabsf(accum);
}

ov3 orc_texture_get_color_at(const oscene* sc, const omat* m, ov2 uv, ov3 p)
{
    switch (m->tex_type) {
    case ORC_TTYPE_COLOR:
        return v3(m->color0[0], m->color0[1], m->color0[2]);
    case ORC_TTYPE_CHECKER: {
        float sines = sinf(10 * p.x) * sinf(10 * p.y) * sinf(10 * p.z);
        if (sines < 0)
            return v3(m->color0[0], m->color0[1], m->color0[2]);
        return v3(m->color1[0], m->color1[1], m->color1[2]);
    }
    case ORC_TTYPE_IMAGE_ALBEDO: {
        const oimage* img = &sc->images[m->image];
        uv.u = orc_clamp(uv.u, 0.0f, 1.0f);
        uv.v = orc_clamp(uv.v, 0.0f, 1.0f);
        uint32_t i = (uint32_t)(uv.u * img->w);
        uint32_t j = (uint32_t)(uv.v * img->h);
        if (i >= img->w) i = img->w - 1;
        if (j >= img->h) j = img->h - 1;
        const float color_scale = 1.0f / 255.0f;
        const uint8_t* pix = img->data + j * img->w * img->bpp + i * img->bpp;
        return v3(color_scale * pix[0], color_scale * pix[1], color_scale * pix[2]);
    }
    case ORC_TTYPE_NOISE: {     /* extension, see onoise in the header */
        float s = 0.5f * (1.0f + sinf(m->color1[0] * p.z + 10 * orc_noise_turb(sc->noise, p, 7)));
        return v3(m->color0[0] * s, m->color0[1] * s, m->color0[2] * s);
    }
    }
    /* The reference falls off the end of the function here (undefined value);
     * no scene on the path reaches it.  Black is returned. */
    return v3(0, 0, 0);
}

/* src/caster.c:425-491.  The profiler hook at :427 is left out: it touches
 * neither the ray nor the random stream. */
ov3 orc_ray_color(const oscene* sc, orng* g, const oray* r, int depth, int64_t* total_rays)
{
    const ov3 black = v3(0.0f, 0.0f, 0.0f);
    const ov3 white = v3(1.0f, 1.0f, 1.0f);
    ohit rec;

    if (depth <= 0)
        return black;

    *total_rays += 1;

    if (orc_hit_list_hit_all(sc, r, 0.001f, 0xFFFFFF, &rec)) {
        oray scattered;
        const omat* m = &sc->mats[rec.m];

        switch (m->type) {
        case ORC_MTYPE_LAMBERTIAN:
            if (orc_scatter_lambertian(sc, g, r, &rec, &scattered)) {
                ov3 color = orc_texture_get_color_at(sc, m, rec.uv, rec.p);
                return v3_mul(color, orc_ray_color(sc, g, &scattered, depth - 1, total_rays));
            }
            break;
        case ORC_MTYPE_METAL:
            if (orc_scatter_metal(sc, g, r, &rec, &scattered)) {
                ov3 color = orc_texture_get_color_at(sc, m, rec.uv, rec.p);
                return v3_mul(color, orc_ray_color(sc, g, &scattered, depth - 1, total_rays));
            }
            break;
        case ORC_MTYPE_DIELECTRIC:
            if (orc_scatter_dielectric(sc, g, r, &rec, &scattered))
                return v3_mul(white, orc_ray_color(sc, g, &scattered, depth - 1, total_rays));
            break;
        case ORC_MTYPE_DIFF_LIGHT:
            switch (m->tex_type) {
            case ORC_TTYPE_COLOR:
                return v3(m->color0[0], m->color0[1], m->color0[2]);
            case ORC_TTYPE_IMAGE_ALBEDO:
                return orc_texture_get_color_at(sc, m, rec.uv, rec.p);
            }
        }
        return black;
    }
    return black;
}

/* ----------------------------------------------------------------- camera -- */
/* src/camera.c:3-30 */
ocam orc_calculate_cam_info(ov3 pos, ov3 look_at, ov3 up, float fov, float ar, float aperture, float focus)
{
    ocam ci;
    float theta = fov * 3.141592654f / 180.0f;
    float h = tanf(theta / 2.0f);
    float viewport_height = 2.0f * h;
    float viewport_width = ar * viewport_height;

    ov3 w = orc_to_unit(v3_sub(pos, look_at));
    ov3 u = orc_to_unit(v3_cross(up, w));
    ov3 v = v3_cross(w, u);

    ci.u0 = u;
    ci.v0 = v;
    ci.o = pos;
    ci.h = v3_smul(u, focus * viewport_width);
    ci.v = v3_smul(v, focus * viewport_height);

    ov3 hh = v3_sdiv(ci.h, 2);
    ov3 hv = v3_sdiv(ci.v, 2);
    ci.llc = v3_sub(v3_sub(v3_sub(ci.o, hh), hv), v3_smul(w, focus));
    ci.lr = aperture / 2.0f;
    return ci;
}

/* src/camera.c:32-42 */
oray orc_ray_from_cam_info(const ocam* ci, orng* g, float u, float v)
{
    oray r;
    ov3 rd = v3_smul(orc_random_unit_disk(g), ci->lr);
    ov3 offset = v3_add(v3_smul(ci->u0, rd.x), v3_smul(ci->v0, rd.y));
    r.o = v3_add(ci->o, offset);
    r.d = v3_add(ci->llc, v3_add(v3_smul(ci->h, u), v3_sub(v3_sub(v3_smul(ci->v, v), ci->o), offset)));
    return r;
}

/* ------------------------------------------------------ pixel loop + store -- */
/* src/image.c:30-36 */
void orc_quantize_pixel(ov3 color, uint8_t out_rgba[4])
{
    out_rgba[0] = (uint8_t)(orc_clamp(color.x, 0.0f, 0.999f) * 256);
    out_rgba[1] = (uint8_t)(orc_clamp(color.y, 0.0f, 0.999f) * 256);
    out_rgba[2] = (uint8_t)(orc_clamp(color.z, 0.0f, 0.999f) * 256);
    out_rgba[3] = 0x00;
}

/* src/dispatcher.c:18-27 */
ov3 orc_pixel_sum(const oscene* sc, const ocam* cam, int width, int height, int i, int j,
                  int s_begin, int s_count, int max_depth, uint32_t seed, int rng_mode,
                  int64_t* rays)
{
    ov3 pixel_color = v3(0, 0, 0);
    orng g;
    if (rng_mode == ORC_RNG_LIBC)
        orc_rng_libc(&g);
    for (int s = s_begin; s < s_begin + s_count; s++) {
        if (rng_mode == ORC_RNG_COUNTER)
            orc_rng_seed(&g, seed, (uint32_t)(j * width + i), (uint32_t)s);
        float u = (float)(i + orc_random_f32(&g)) / (width - 1);
        float v = (float)(j + orc_random_f32(&g)) / (height - 1);
        oray r = orc_ray_from_cam_info(cam, &g, u, v);
        pixel_color = v3_add(pixel_color, orc_ray_color(sc, &g, &r, max_depth, rays));
    }
    return pixel_color;
}

int64_t orc_render_rows(const oscene* sc, const ocam* cam, int width, int height, int j0, int j1,
                        int s_begin, int s_count, int max_depth, uint32_t seed, int rng_mode,
                        float* sums)
{
    int64_t rays = 0;
    for (int j = j0; j < j1; j++) {
        for (int i = 0; i < width; i++) {
            ov3 c = orc_pixel_sum(sc, cam, width, height, i, j, s_begin, s_count, max_depth, seed,
                                  rng_mode, &rays);
            float* px = sums + 4 * ((size_t)j * width + i);
            px[0] += c.x;
            px[1] += c.y;
            px[2] += c.z;
            px[3] += (float)s_count;
        }
    }
    return rays;
}

/* src/dispatcher.c:11,29-34: scale = 1.0f / spp; multiply; component sqrt; store at row height-j-1 */
void orc_resolve(const float* sums, int width, int height, int spp, uint8_t* rgba)
{
    float scale = 1.0f / spp;
    for (int j = 0; j < height; j++) {
        for (int i = 0; i < width; i++) {
            const float* px = sums + 4 * ((size_t)j * width + i);
            ov3 c = v3_smul(v3(px[0], px[1], px[2]), scale);
            c = v3(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z));
            int y = height - j - 1;
            orc_quantize_pixel(c, rgba + 4 * ((size_t)y * width + i));
        }
    }
}

/* ------------------------------------------------- row-block dispatcher port -- */
typedef struct {
    const oscene* sc;
    const ocam* cam;
    int width, height, spp, max_depth, j0, j1;
    uint32_t seed;
    int rng_mode;
    float* sums;
    int64_t rays;
} orc_job;

static void* orc_job_main(void* arg)
{
    orc_job* job = (orc_job*)arg;
    job->rays = orc_render_rows(job->sc, job->cam, job->width, job->height, job->j0, job->j1, 0,
                                job->spp, job->max_depth, job->seed, job->rng_mode, job->sums);
    return NULL;
}

/* src/dispatcher.c:12,61-113: block_size = height / blocks (integer division, the
 * remainder rows are never rendered); one job per thread, job k owns rows
 * [k*block_size, (k+1)*block_size).  Threads are pthreads here. */
int64_t orc_dispatch(const oscene* sc, const ocam* cam, int width, int height, int spp,
                     int max_depth, uint32_t seed, int threads, int blocks,
                     float* sums, uint8_t* rgba)
{
    return orc_dispatch_rng(sc, cam, width, height, spp, max_depth, seed, threads, blocks,
                            ORC_RNG_COUNTER, sums, rgba);
}

/* The same dispatcher with the random source chosen.  ORC_RNG_LIBC is the reference as it
 * ships: every thread draws from the one libc rand() (src/common.c:10-13), whose hidden state
 * glibc guards with a lock, so the threads serialise on it and the image is not reproducible. */
int64_t orc_dispatch_rng(const oscene* sc, const ocam* cam, int width, int height, int spp,
                         int max_depth, uint32_t seed, int threads, int blocks, int rng_mode,
                         float* sums, uint8_t* rgba)
{
    if (threads > 64) threads = 64; /* MAX_THREADS, inc/dispatcher.h:10 */
    if (threads < 1) threads = 1;
    if (blocks < 1) blocks = 1;
    int block_size = height / blocks;
    orc_job jobs[64];
    pthread_t tids[64];
    int64_t total = 0;
    for (int k = 0; k < threads; k++) {
        orc_job* jb = &jobs[k];
        jb->sc = sc; jb->cam = cam; jb->width = width; jb->height = height;
        jb->spp = spp; jb->max_depth = max_depth; jb->seed = seed; jb->sums = sums;
        jb->rng_mode = rng_mode;
        jb->j0 = k * block_size;
        jb->j1 = (k + 1) * block_size;
        if (jb->j0 > height) jb->j0 = height;
        if (jb->j1 > height) jb->j1 = height;
        jb->rays = 0;
        pthread_create(&tids[k], NULL, orc_job_main, jb);
    }
    for (int k = 0; k < threads; k++) {
        pthread_join(tids[k], NULL);
        total += jobs[k].rays;
    }
    if (rgba)
        orc_resolve(sums, width, height, spp, rgba);
    return total;
}

/* ------------------------------------------------------------- BMP bytes -- */
/* src/image.c:98-196: 24-bit, rows written from the buffer's last row to its
 * first, each pixel's bytes in buffer order R,G,B (the format's slots are
 * B,G,R, so red and blue appear swapped in viewers); rows padded to 4 bytes. */
int64_t orc_bmp_bytes(const uint8_t* rgba, int width, int height, uint8_t* out)
{
    int width_in_bytes = width * 3;
    int padding = (4 - (width_in_bytes) % 4) % 4;
    int stride = width_in_bytes + padding;
    int file_size = 14 + 40 + stride * height;
    uint8_t* p = out;
    memset(p, 0, 54);
    p[0] = 'B'; p[1] = 'M';
    p[2] = (uint8_t)(file_size); p[3] = (uint8_t)(file_size >> 8);
    p[4] = (uint8_t)(file_size >> 16); p[5] = (uint8_t)(file_size >> 24);
    p[10] = 54;
    p += 14;
    p[0] = 40;
    p[4] = (uint8_t)(width); p[5] = (uint8_t)(width >> 8);
    p[6] = (uint8_t)(width >> 16); p[7] = (uint8_t)(width >> 24);
    p[8] = (uint8_t)(height); p[9] = (uint8_t)(height >> 8);
    p[10] = (uint8_t)(height >> 16); p[11] = (uint8_t)(height >> 24);
    p[12] = 1;
    p[14] = 24;
    p += 40;
    for (int y = height - 1; y >= 0; y--) {
        for (int x = 0; x < width; x++) {
            const uint8_t* px = rgba + 4 * ((size_t)y * width + x);
            *p++ = px[0]; *p++ = px[1]; *p++ = px[2];
        }
        for (int k = 0; k < padding; k++) *p++ = 0;
    }
    return (int64_t)(p - out);
}

/* ---------------------------------------------------------- batch drivers -- */
static oray ray_from6(const float* r6)
{
    oray r;
    r.o = v3(r6[0], r6[1], r6[2]);
    r.d = v3(r6[3], r6[4], r6[5]);
    return r;
}

static void hit_to_arrays(int k, int ok, const ohit* rec, int32_t* hit, float* t, float* p3,
                          float* n3, int32_t* face, float* uv2)
{
    hit[k] = ok;
    if (!ok) {
        t[k] = 0; face[k] = 0;
        p3[3 * k] = p3[3 * k + 1] = p3[3 * k + 2] = 0;
        n3[3 * k] = n3[3 * k + 1] = n3[3 * k + 2] = 0;
        if (uv2) uv2[2 * k] = uv2[2 * k + 1] = 0;
        return;
    }
    t[k] = rec->t; face[k] = rec->f;
    p3[3 * k] = rec->p.x; p3[3 * k + 1] = rec->p.y; p3[3 * k + 2] = rec->p.z;
    n3[3 * k] = rec->n.x; n3[3 * k + 1] = rec->n.y; n3[3 * k + 2] = rec->n.z;
    if (uv2) { uv2[2 * k] = rec->uv.u; uv2[2 * k + 1] = rec->uv.v; }
}

void orc_batch_hit_sphere(int n, const float* sph4, const float* ray6, const float* trange2,
                          int32_t* hit, float* t, float* p3, float* n3, int32_t* face, float* uv2)
{
    for (int k = 0; k < n; k++) {
        oobj s;
        memset(&s, 0, sizeof s);
        s.type = ORC_HTYPE_SPHERE;
        memcpy(s.g, sph4 + 4 * k, 4 * sizeof(float));
        oray r = ray_from6(ray6 + 6 * k);
        ohit rec;
        memset(&rec, 0, sizeof rec);
        int ok = orc_hit_sphere(&s, &r, trange2[2 * k], trange2[2 * k + 1], &rec);
        hit_to_arrays(k, ok, &rec, hit, t, p3, n3, face, uv2);
    }
}

void orc_batch_hit_xy_rect(int n, const float* rect5, const float* ray6, const float* trange2,
                           int32_t* hit, float* t, float* p3, float* n3, int32_t* face, float* uv2)
{
    for (int k = 0; k < n; k++) {
        oobj s;
        memset(&s, 0, sizeof s);
        s.type = ORC_HTYPE_RECT_XY;
        memcpy(s.g, rect5 + 5 * k, 5 * sizeof(float));
        oray r = ray_from6(ray6 + 6 * k);
        ohit rec;
        memset(&rec, 0, sizeof rec);
        int ok = orc_hit_xy_rect(&s, &r, trange2[2 * k], trange2[2 * k + 1], &rec);
        hit_to_arrays(k, ok, &rec, hit, t, p3, n3, face, uv2);
    }
}

void orc_batch_hit_list(const oscene* sc, int n, const float* ray6,
                        int32_t* hit, float* t, float* p3, float* n3, int32_t* face, int32_t* mat)
{
    for (int k = 0; k < n; k++) {
        oray r = ray_from6(ray6 + 6 * k);
        ohit rec;
        memset(&rec, 0, sizeof rec);
        int ok = orc_hit_list_hit_all(sc, &r, 0.001f, 0xFFFFFF, &rec);
        hit_to_arrays(k, ok, &rec, hit, t, p3, n3, face, NULL);
        mat[k] = ok ? rec.m : -1;
    }
}

void orc_batch_scatter(const oscene* sc, int n, const float* ray6, const float* p3, const float* n3,
                       const int32_t* face, const int32_t* mat, uint32_t seed,
                       int32_t* ok, float* out_ray6, int32_t* draws)
{
    for (int k = 0; k < n; k++) {
        oray r = ray_from6(ray6 + 6 * k);
        ohit rec;
        memset(&rec, 0, sizeof rec);
        rec.p = v3(p3[3 * k], p3[3 * k + 1], p3[3 * k + 2]);
        rec.n = v3(n3[3 * k], n3[3 * k + 1], n3[3 * k + 2]);
        rec.f = face[k];
        rec.m = mat[k];
        orng g;
        orc_rng_seed(&g, seed, (uint32_t)k, 0);
        oray out;
        memset(&out, 0, sizeof out);
        int res = 0;
        switch (sc->mats[rec.m].type) {
        case ORC_MTYPE_LAMBERTIAN: res = orc_scatter_lambertian(sc, &g, &r, &rec, &out); break;
        case ORC_MTYPE_METAL:      res = orc_scatter_metal(sc, &g, &r, &rec, &out); break;
        case ORC_MTYPE_DIELECTRIC: res = orc_scatter_dielectric(sc, &g, &r, &rec, &out); break;
        default: break;
        }
        ok[k] = res;
        out_ray6[6 * k] = out.o.x; out_ray6[6 * k + 1] = out.o.y; out_ray6[6 * k + 2] = out.o.z;
        out_ray6[6 * k + 3] = out.d.x; out_ray6[6 * k + 4] = out.d.y; out_ray6[6 * k + 5] = out.d.z;
        draws[k] = (int32_t)g.draws;
    }
}

void orc_batch_camera_ray(const ocam* cam, int n, const float* uv2, uint32_t seed,
                          float* out_ray6, int32_t* draws)
{
    for (int k = 0; k < n; k++) {
        orng g;
        orc_rng_seed(&g, seed, (uint32_t)k, 0);
        oray r = orc_ray_from_cam_info(cam, &g, uv2[2 * k], uv2[2 * k + 1]);
        out_ray6[6 * k] = r.o.x; out_ray6[6 * k + 1] = r.o.y; out_ray6[6 * k + 2] = r.o.z;
        out_ray6[6 * k + 3] = r.d.x; out_ray6[6 * k + 4] = r.d.y; out_ray6[6 * k + 5] = r.d.z;
        draws[k] = (int32_t)g.draws;
    }
}

void orc_batch_texture(const oscene* sc, int n, const int32_t* mat, const float* uv2, const float* p3,
                       float* rgb3)
{
    for (int k = 0; k < n; k++) {
        ov2 uv; uv.u = uv2[2 * k]; uv.v = uv2[2 * k + 1];
        ov3 c = orc_texture_get_color_at(sc, &sc->mats[mat[k]], uv, v3(p3[3 * k], p3[3 * k + 1], p3[3 * k + 2]));
        rgb3[3 * k] = c.x; rgb3[3 * k + 1] = c.y; rgb3[3 * k + 2] = c.z;
    }
}

void orc_batch_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out23)
{
    orng g;
    orc_rng_seed(&g, seed, pixel, sample);
    for (int k = 0; k < n; k++)
        out23[k] = orc_rng_next23(&g);
}

/* The small vector and optics functions, array at a time (item k reads in8[8k..], writes out3[3k..]):
 * 0 reflectance, 1 refract (cos_theta in in[7]), 2 reflect, 3 get_uv_coord of a sphere normal,
 * 4 near_zero, 5 to_unit.  src/common.c:89-97,152-175, src/material.c:45-56. */
void orc_batch_math(int fn, int n, const float* in8, float* out3)
{
    for (int k = 0; k < n; k++) {
        const float* a = in8 + 8 * (size_t)k;
        ov3 r = v3(0.0f, 0.0f, 0.0f);
        switch (fn) {
        case 0: r.x = orc_reflectance(a[0], a[1]); break;
        case 1: r = orc_refract(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), a[6], a[7]); break;
        case 2: r = orc_reflect(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5])); break;
        case 3: { ov2 uv = orc_get_uv_coord_sphere(v3(a[0], a[1], a[2])); r.x = uv.u; r.y = uv.v; break; }
        case 4: r.x = orc_near_zero(v3(a[0], a[1], a[2])) ? 1.0f : 0.0f; break;
        case 5: r = orc_to_unit(v3(a[0], a[1], a[2])); break;
        default: break;
        }
        out3[3 * k] = r.x; out3[3 * k + 1] = r.y; out3[3 * k + 2] = r.z;
    }
}
