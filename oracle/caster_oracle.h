/*
 * caster_oracle.h -- CPU restatement of CRaytracer's per-pixel path-tracing path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (libcaster_hip, the C host,
 * the Python package) may include, link or call this.  Only tests/, the smoke
 * check and the cpu_baseline leg of bench.py use it, and only as the checker
 * or as the timed CPU baseline.
 *
 * PARITY UNPINNED.  The reference holds no tests, golden vectors or fixtures
 * for this path, and its sources cannot be compiled in this image: every
 * translation unit includes <windows.h> (inc/common.h:8), which the image
 * lacks.  This file is therefore a by-hand restatement of the reference
 * algorithm; each function cites the reference lines it follows.  The golden
 * vectors under tests/golden/ were produced by this restatement and guard it
 * against regressions; they do not pin it to the reference.
 *
 * All arithmetic is IEEE binary32 evaluated in the reference's association
 * order; build with -ffp-contract=off so no FMA is formed.
 */
#ifndef CASTER_ORACLE_H
#define CASTER_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ov3;          /* inc/common.h:45-56 */
typedef struct { float u, v; } ov2;             /* inc/common.h:32-42 */
typedef struct { ov3 o, d; } oray;              /* inc/caster.h:6-10  */

/* inc/caster.h:18-26; the material pointer is restated as a table index. */
typedef struct {
    ov3 p, n;
    float t;
    int32_t f;
    int32_t m;
    ov2 uv;
} ohit;

/* Object / material / texture tags: inc/caster.h:71-75, inc/material.h:17-20,32-35 */
enum { ORC_HTYPE_SPHERE = 0, ORC_HTYPE_RECT_XY = 1 };
enum { ORC_MTYPE_LAMBERTIAN = 0, ORC_MTYPE_METAL = 1, ORC_MTYPE_DIELECTRIC = 2, ORC_MTYPE_DIFF_LIGHT = 3 };
enum { ORC_TTYPE_COLOR = 0, ORC_TTYPE_CHECKER = 1, ORC_TTYPE_IMAGE_ALBEDO = 2, ORC_TTYPE_IMAGE_EMIT = 3,
       ORC_TTYPE_NOISE = 4 /* extension, see onoise */ };

/* inc/material.h:10-30 flattened: two colour units are all the path reads. */
typedef struct {
    int32_t type;
    int32_t tex_type;
    float color0[3];
    float color1[3];
    float fuzziness;
    float n_refract;
    int32_t image;      /* index into oscene.images, -1 if none */
    int32_t pad;
} omat;                 /* 48 bytes */

/* One entry of the hit_list (inc/caster.h:39-55,64-69), list order preserved.
 * sphere: g = {cx, cy, cz, radius}; xy rect: g = {cx, cy, cz, du, dv}. */
typedef struct {
    int32_t type;
    int32_t material;
    float g[6];
} oobj;                 /* 32 bytes */

typedef struct {        /* inc/image.h:6-12 */
    const uint8_t* data;
    uint32_t w, h, bpp;
} oimage;

/* Perlin noise tables.  EXTENSION WITHOUT A REFERENCE COUNTERPART: the reference declares
 * `noise_perlin { int todo; }` (inc/noise.h:4-7) and src/noise.c is empty.  What is restated here
 * is the published gradient-noise construction of "Ray Tracing: The Next Week" (256 random unit
 * vectors, three permutation tables, Hermite-smoothed trilinear interpolation, 7-octave
 * turbulence, marble texture colour * 0.5 * (1 + sin(scale * p.z + 10 * turb(p)))).  A noise
 * texture is a material with tex_type ORC_TTYPE_NOISE, color0 = colour, color1[0] = scale. */
typedef struct {
    float ranvec[256][3];
    uint8_t perm_x[256], perm_y[256], perm_z[256];
} onoise;               /* 3840 bytes */

typedef struct {
    int32_t count;
    int32_t nmats;
    int32_t nimages;
    int32_t pad;
    const oobj* objs;
    const omat* mats;
    const oimage* images;
    const onoise* noise;    /* NULL unless a material uses ORC_TTYPE_NOISE */
} oscene;

typedef struct {        /* inc/camera.h:6-13 */
    ov3 o, llc, h, v, u0, v0;
    float lr;
} ocam;

/* Random source.  mode 0: counter generator keyed by (seed, pixel, sample);
 * mode 1: libc rand() exactly as src/common.c:10-13 (not thread safe). */
enum { ORC_RNG_COUNTER = 0, ORC_RNG_LIBC = 1 };
typedef struct {
    uint32_t s[4];
    uint64_t draws;
    int32_t mode;
    int32_t pad;
} orng;

void     orc_rng_seed(orng* g, uint32_t seed, uint32_t pixel, uint32_t sample);
void     orc_rng_libc(orng* g);
uint32_t orc_rng_next23(orng* g);

float orc_random_f32(orng* g);
float orc_random_range_f32(orng* g, float lo, float hi);
float orc_clamp(float x, float lo, float hi);
ov3   orc_random_unit_sphere(orng* g);
ov3   orc_random_unit_disk(orng* g);
ov3   orc_random_unit_norm(orng* g);
int   orc_near_zero(ov3 v);
ov3   orc_reflect(ov3 v, ov3 n);
ov3   orc_refract(ov3 unit_v, ov3 n, float ratio, float cos_theta);
float orc_reflectance(float cosine, float ratio);
float orc_dot(ov3 a, ov3 b);
ov3   orc_to_unit(ov3 v);

ov3  orc_ray_at(const oray* r, float t);
int  orc_hit_sphere(const oobj* s, const oray* r, float tmin, float tmax, ohit* rec);
int  orc_hit_xy_rect(const oobj* rt, const oray* r, float tmin, float tmax, ohit* rec);
int  orc_hit_list_hit_all(const oscene* sc, const oray* r, float tmin, float tmax, ohit* rec);
int  orc_scatter_lambertian(const oscene* sc, orng* g, const oray* r, const ohit* rec, oray* out);
int  orc_scatter_metal(const oscene* sc, orng* g, const oray* r, const ohit* rec, oray* out);
int  orc_scatter_dielectric(const oscene* sc, orng* g, const oray* r, const ohit* rec, oray* out);
ov2  orc_get_uv_coord_sphere(ov3 p);
ov3  orc_texture_get_color_at(const oscene* sc, const omat* m, ov2 uv, ov3 p);
float orc_noise_value(const onoise* n, ov3 p);
float orc_noise_turb(const onoise* n, ov3 p, int depth);
ov3  orc_ray_color(const oscene* sc, orng* g, const oray* r, int depth, int64_t* total_rays);

ocam orc_calculate_cam_info(ov3 pos, ov3 look_at, ov3 up, float fov, float ar, float aperture, float focus);
oray orc_ray_from_cam_info(const ocam* ci, orng* g, float u, float v);

void orc_quantize_pixel(ov3 color, uint8_t out_rgba[4]);

/* The sample loop of src/dispatcher.c:19-30 for pixel (i, j), samples
 * [s_begin, s_begin + s_count): returns the linear RGB SUM (no scale, no sqrt). */
ov3 orc_pixel_sum(const oscene* sc, const ocam* cam, int width, int height, int i, int j,
                  int s_begin, int s_count, int max_depth, uint32_t seed, int rng_mode,
                  int64_t* rays);

/* Rows [j0, j1) of the frame.  sums: width*height*4 floats indexed [(j*width+i)*4 + c]
 * (j = 0 is the image bottom, as in src/dispatcher.c:12), channel 3 receives the
 * sample count; sums are ADDED to what the buffer holds.  Returns rays shot. */
int64_t orc_render_rows(const oscene* sc, const ocam* cam, int width, int height, int j0, int j1,
                        int s_begin, int s_count, int max_depth, uint32_t seed, int rng_mode,
                        float* sums);

/* src/dispatcher.c:29-34 + src/image.c:30-36: scale, gamma, quantise, y flip. */
void orc_resolve(const float* sums, int width, int height, int spp, uint8_t* rgba);

/* pthread re-host of the dispatcher's row-block split (src/dispatcher.c:12,86,103-113):
 * `threads` jobs, job k renders rows [k*(height/blocks), (k+1)*(height/blocks)). */
int64_t orc_dispatch(const oscene* sc, const ocam* cam, int width, int height, int spp,
                     int max_depth, uint32_t seed, int threads, int blocks,
                     float* sums, uint8_t* rgba);
/* The same with the random source chosen: ORC_RNG_LIBC has every thread draw from the one libc
 * rand() as src/common.c:10-13 does (glibc serialises the calls on a lock). */
int64_t orc_dispatch_rng(const oscene* sc, const ocam* cam, int width, int height, int spp,
                         int max_depth, uint32_t seed, int threads, int blocks, int rng_mode,
                         float* sums, uint8_t* rgba);

/* Bytes of the BMP that src/image.c:98-196 writes for an RGBA8 buffer.
 * out must hold 54 + height*stride bytes; returns the byte count. */
int64_t orc_bmp_bytes(const uint8_t* rgba, int width, int height, uint8_t* out);

/* Batch drivers for array-at-a-time parity tests. */
void orc_batch_hit_sphere(int n, const float* sph4, const float* ray6, const float* trange2,
                          int32_t* hit, float* t, float* p3, float* n3, int32_t* face, float* uv2);
void orc_batch_hit_xy_rect(int n, const float* rect5, const float* ray6, const float* trange2,
                           int32_t* hit, float* t, float* p3, float* n3, int32_t* face, float* uv2);
void orc_batch_hit_list(const oscene* sc, int n, const float* ray6,
                        int32_t* hit, float* t, float* p3, float* n3, int32_t* face, int32_t* mat);
void orc_batch_scatter(const oscene* sc, int n, const float* ray6, const float* p3, const float* n3,
                       const int32_t* face, const int32_t* mat, uint32_t seed,
                       int32_t* ok, float* out_ray6, int32_t* draws);
void orc_batch_camera_ray(const ocam* cam, int n, const float* uv2, uint32_t seed,
                          float* out_ray6, int32_t* draws);
void orc_batch_texture(const oscene* sc, int n, const int32_t* mat, const float* uv2, const float* p3,
                       float* rgb3);
void orc_batch_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out23);
void orc_batch_math(int fn, int n, const float* in8, float* out3);

#ifdef __cplusplus
}
#endif
#endif
