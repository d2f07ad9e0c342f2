/*
 * caster_hip.h -- C ABI of libcaster_hip, the MI355X (gfx950) implementation of
 * CRaytracer's per-pixel path-tracing path.
 *
 * The reference has no plugin or FFI layer; its seam is the dispatcher job API
 * (inc/dispatcher.h:49-53) whose worker, job_func (src/dispatcher.c:9-53), runs
 * the per-pixel sample loop.  This library replaces what job_func calls:
 * ray_from_cam_info (src/camera.c:32-42), ray_color (src/caster.c:425-491),
 * hit_list_hit_all / hit_sphere / hit_xy_rect (src/caster.c:99-236),
 * ray_scatter_* and texture_get_color_at (src/material.c:3-101), the sample
 * accumulation and gamma (src/dispatcher.c:19-30) and the pixel store
 * (src/image.c:30-36).
 *
 * Plain C: opaque handle, plain pointers and sizes, int status (0 = success).
 * After a non-zero status caster_hip_last_error() describes the failure
 * (thread-local string).  There is no CPU fallback: without a HIP device every
 * compute entry point fails with CASTER_HIP_ERR_DEVICE.
 */
#ifndef CASTER_HIP_H
#define CASTER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    CASTER_HIP_OK = 0,
    CASTER_HIP_ERR_ARG = 1,     /* null pointer, bad size, unknown tag      */
    CASTER_HIP_ERR_DEVICE = 2,  /* HIP runtime error, or no device          */
    CASTER_HIP_ERR_STATE = 3    /* call order (e.g. render before a scene)  */
};

/* Render flags. */
enum {
    /* Arithmetic in the reference's operation order, no FMA contraction,
     * correctly rounded divide/sqrt.  Without it the kernel may use FMA,
     * reciprocals and a normalised ray direction (same image statistically). */
    CASTER_HIP_STRICT = 1u,
    /* Evaluate the conservative sphere filter on the vector unit even where the scene qualifies
     * for the matrix cores.  Both forms hand the same candidates' exact tests to the same code, so
     * the results are identical; the flag exists for scenes' A/B timing and for the tests. */
    CASTER_HIP_VECTOR_LOOP = 2u
};

/* Object, material and texture tags: inc/caster.h:71-75, inc/material.h:17-20,32-35. */
enum { CASTER_HTYPE_SPHERE = 0, CASTER_HTYPE_RECT_XY = 1 };
enum { CASTER_MTYPE_LAMBERTIAN = 0, CASTER_MTYPE_METAL = 1, CASTER_MTYPE_DIELECTRIC = 2, CASTER_MTYPE_DIFF_LIGHT = 3 };
enum { CASTER_TTYPE_COLOR = 0, CASTER_TTYPE_CHECKER = 1, CASTER_TTYPE_IMAGE_ALBEDO = 2, CASTER_TTYPE_IMAGE_EMIT = 3,
       CASTER_TTYPE_NOISE = 4 /* extension: see caster_noise */ };

/* One hit_list entry (inc/caster.h:39-55,64-69) with pointers flattened.
 * sphere: g = {cx, cy, cz, radius, -, -};  xy rect: g = {cx, cy, cz, du, dv, -}.
 * Array order is list order (ties in t go to the later entry, src/caster.c:114,204). */
typedef struct {
    int32_t type;
    int32_t material;   /* index into the material table */
    float g[6];
} caster_object;

/* inc/material.h:10-30, flattened.  color0/color1 are texture.color[0]/[1]. */
typedef struct {
    int32_t type;
    int32_t tex_type;
    float color0[3];
    float color1[3];
    float fuzziness;
    float n_refract;
    int32_t image;      /* index into the image table, -1 if none */
    int32_t pad;
} caster_material;

/* inc/image.h:6-12; data is HOST memory, row-major, bpp bytes per texel. */
typedef struct {
    const uint8_t* data;
    uint32_t w, h, bpp;
} caster_image;

/* Perlin noise tables for CASTER_TTYPE_NOISE materials (color0 = colour, color1[0] = scale):
 * colour * 0.5 * (1 + sin(scale * p.z + 10 * turbulence(p))), 7 octaves of gradient noise over
 * 256 unit vectors chosen through three xor-ed permutation tables.  The reference only declares
 * a placeholder (inc/noise.h:4-7, src/noise.c is empty): there is nothing in it to match, the
 * construction is the published one of "Ray Tracing: The Next Week". */
typedef struct {
    float ranvec[256][3];
    uint8_t perm_x[256], perm_y[256], perm_z[256];
} caster_noise;

/* inc/camera.h:6-13 (cam_info): 19 floats, same order. */
typedef struct {
    float o[3], llc[3], h[3], v[3], u0[3], v0[3];
    float lr;
} caster_camera;

typedef struct caster_hip_ctx caster_hip_ctx;

const char* caster_hip_last_error(void);
int caster_hip_device_count(int* count);

/* One context per GPU; it owns the uploaded scene and scratch slabs. */
int caster_hip_create(int device, caster_hip_ctx** out);
int caster_hip_destroy(caster_hip_ctx* ctx);

/* Kernels are launched on this hipStream_t (NULL = the default stream). */
int caster_hip_set_stream(caster_hip_ctx* ctx, void* hip_stream);

/* Tuning knob: samples of one 8x8 pixel tile handled by one work unit.  0 (the default) lets the
 * library plan the chunks itself: up to 64 samples first, smaller ones towards the end of the
 * call so that the waves finish together.  It changes scheduling only; the rendered paths and
 * the sums do not depend on it. */
int caster_hip_set_chunk(caster_hip_ctx* ctx, int samples_per_unit);

/* Replaces the pointer-chasing hit_list the dispatcher holds (inc/dispatcher.h:31):
 * objects in list order plus the material and image tables they index. */
int caster_hip_upload_scene(caster_hip_ctx* ctx,
                            int n_objects, const caster_object* objects,
                            int n_materials, const caster_material* materials,
                            int n_images, const caster_image* images);

/* The noise tables of the scene uploaded last (inc/noise.h:4-7 is the reference's placeholder for
 * them).  Required before rendering a scene that has a CASTER_TTYPE_NOISE material;
 * caster_hip_upload_scene forgets them. */
int caster_hip_set_noise(caster_hip_ctx* ctx, const caster_noise* noise);

/* Device memory helpers for hosts that do not manage HIP memory themselves. */
int caster_hip_malloc(caster_hip_ctx* ctx, size_t bytes, void** dev_ptr);
int caster_hip_free(caster_hip_ctx* ctx, void* dev_ptr);
int caster_hip_memset_zero(caster_hip_ctx* ctx, void* dev_ptr, size_t bytes);
int caster_hip_download(caster_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
int caster_hip_upload(caster_hip_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes);
int caster_hip_synchronize(caster_hip_ctx* ctx);

/* dst_dev[i] += src_dev[i] for n_floats floats, where src_dev lives on device src_device (memory
 * of another context, whose work on it has completed) and dst_dev on ctx's device: a peer-to-peer
 * copy (xGMI between the GPUs of a node) into a staging buffer, then one add kernel.  This is the
 * sum of the per-GPU sample partitions; the reference has no counterpart, its threads write
 * disjoint row blocks (src/dispatcher.c:12).  Adding the partitions in a fixed order gives
 * reproducible sums. */
int caster_hip_add_from_device(caster_hip_ctx* ctx, float* dst_dev, int src_device, const float* src_dev, size_t n_floats);

/*
 * The sample loop of job_func (src/dispatcher.c:12-27) for every pixel and the
 * samples [sample_begin, sample_begin + sample_count): linear RGB sums are
 * ADDED into accum_dev, a DEVICE buffer of width*height*4 floats indexed
 * [(j*width + i)*4 + c] with j = 0 the image bottom row; channel 3 gains
 * sample_count.  Random draws are keyed by (seed, j*width+i, sample), so any
 * split of the sample range over calls or GPUs renders the same paths.
 * Asynchronous: returns once the work is queued on the context's stream.
 */
int caster_hip_render(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height,
                      int sample_begin, int sample_count, int max_depth,
                      uint32_t seed, uint32_t flags, float* accum_dev);

/* Waits for the last render on this context.  rays = ray_color calls with
 * depth > 0 (src/caster.c:439); kernel_ms = HIP-event time of the path-tracing
 * kernel alone.  Either pointer may be NULL. */
int caster_hip_render_stats(caster_hip_ctx* ctx, uint64_t* rays, float* kernel_ms);

/* src/dispatcher.c:11,29-34 + src/image.c:30-36: x (1.0f/spp), component sqrt,
 * clamp to [0, 0.999], x256, truncate; alpha 0; row y = height-j-1.
 * rgba is width*height*4 bytes in HOST memory (resolve) or DEVICE memory (_dev). */
int caster_hip_resolve(caster_hip_ctx* ctx, const float* accum_dev, int width, int height,
                       int spp, uint8_t* rgba_host);
int caster_hip_resolve_dev(caster_hip_ctx* ctx, const float* accum_dev, int width, int height,
                           int spp, uint8_t* rgba_dev);

/*
 * Array-at-a-time entry points that run single device functions, so per-ray
 * parity with the reference functions can be checked.  All pointers are HOST
 * memory; n items each.  flags: CASTER_HIP_STRICT or 0.
 */
/* hit_sphere, src/caster.c:99-130.  sph4 = {cx,cy,cz,r}, ray6 = {o,d}, trange2 = {tmin,tmax}. */
int caster_hip_test_hit_sphere(caster_hip_ctx* ctx, uint32_t flags, int n,
                               const float* sph4, const float* ray6, const float* trange2,
                               int32_t* hit, float* t, float* p3, float* n3, int32_t* face);
/* hit_xy_rect, src/caster.c:132-164.  rect5 = {cx,cy,cz,du,dv}. */
int caster_hip_test_hit_xy_rect(caster_hip_ctx* ctx, uint32_t flags, int n,
                                const float* rect5, const float* ray6, const float* trange2,
                                int32_t* hit, float* t, float* p3, float* n3, int32_t* face,
                                float* uv2);
/* hit_list_hit_all over the uploaded scene, tmin 0.001, tmax 16777215 (src/caster.c:198-236,441). */
int caster_hip_test_hit_list(caster_hip_ctx* ctx, uint32_t flags, int n, const float* ray6,
                             int32_t* hit, float* t, float* p3, float* n3, int32_t* face,
                             int32_t* mat);
/* The same closest-hit query through the render kernel's own sphere loop (LDS staging,
 * candidate filter, two packed ray slots per lane, rect merge), so that the production
 * intersection path can be checked ray by ray against hit_list_hit_all. */
int caster_hip_test_trace(caster_hip_ctx* ctx, uint32_t flags, int n, const float* ray6,
                          int32_t* hit, float* t, float* p3, float* n3, int32_t* face,
                          int32_t* mat);
/* ray_scatter_{lambertian,metal,dielectric} by mat[k]'s type (src/material.c:3-43);
 * item k draws from the stream keyed (seed, k, 0).  draws = random_f32 calls made. */
int caster_hip_test_scatter(caster_hip_ctx* ctx, uint32_t flags, int n,
                            const float* ray6, const float* p3, const float* n3,
                            const int32_t* face, const int32_t* mat, uint32_t seed,
                            int32_t* ok, float* out_ray6, int32_t* draws);
/* ray_from_cam_info (src/camera.c:32-42); item k draws from the stream (seed, k, 0). */
int caster_hip_test_camera_ray(caster_hip_ctx* ctx, uint32_t flags, const caster_camera* cam,
                               int n, const float* uv2, uint32_t seed,
                               float* out_ray6, int32_t* draws);
/* texture_get_color_at (src/material.c:65-101) for material mat[k] of the uploaded scene. */
int caster_hip_test_texture(caster_hip_ctx* ctx, uint32_t flags, int n, const int32_t* mat,
                            const float* uv2, const float* p3, float* rgb3);
/* First n 24-bit draws of the stream keyed (seed, pixel, sample). */
int caster_hip_test_rng(caster_hip_ctx* ctx, uint32_t seed, uint32_t pixel, uint32_t sample,
                        int n, uint32_t* out24);

#ifdef __cplusplus
}
#endif
#endif
