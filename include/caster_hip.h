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
    CASTER_HIP_VECTOR_LOOP = 2u,
    /* Camera rays are traced against a short list of spheres per 8x8-pixel tile: the spheres the
     * tile's beam of camera rays (lens disk to the tile's patch of the focus plane) can touch, by a
     * conservative bound.  With this flag every camera ray meets every sphere instead.  The results
     * are identical; the flag exists so that the tests can hold the two against each other. */
    CASTER_HIP_NO_TILE_LISTS = 4u
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
/* Free and total device memory of the context's GPU right now (hipMemGetInfo); either pointer may
 * be NULL.  The reference has no counterpart (it allocates on the host heap, src/image.c:5); hosts
 * use it to size frames, the tests to show that caster_hip_render_noalloc allocates nothing. */
int caster_hip_mem_info(caster_hip_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

/* dst_dev[i] += src_dev[i] for n_floats floats, where src_dev lives on device src_device (memory
 * of another context, whose work on it has completed) and dst_dev on ctx's device: a peer-to-peer
 * copy (xGMI between the GPUs of a node) into a staging buffer, then one add kernel.  This is the
 * sum of the per-GPU sample partitions; the reference has no counterpart, its threads write
 * disjoint row blocks (src/dispatcher.c:12).  Adding the partitions in a fixed order gives
 * reproducible sums. */
int caster_hip_add_from_device(caster_hip_ctx* ctx, float* dst_dev, int src_device, const float* src_dev, size_t n_floats);

/* The sum of the per-GPU sample partitions as one collective: accum_dev[0] (on the device of
 * ctxs[0]) receives the element-wise sum of accum_dev[0..n-1], n_floats floats each (width*height*4),
 * through one RCCL ncclReduce(float32, sum, root 0) per context, issued as a group on the contexts'
 * streams -- over xGMI between the GPUs of a node.  The contexts must sit on n distinct devices
 * (buffers that share a device are summed with caster_hip_add_from_device first).  Asynchronous like
 * caster_hip_render: work queued later on the stream of ctxs[0] sees the sum; the other buffers are
 * unchanged.  The linear sums must be reduced BEFORE the 1/spp scale and the sqrt of
 * src/dispatcher.c:29-30, which caster_hip_resolve applies afterwards.  The reference has no
 * counterpart: its threads share one img_buffer and write disjoint row blocks (src/dispatcher.c:12).
 * RCCL (librccl.so.1) is opened on the first call; without it the call fails with
 * CASTER_HIP_ERR_DEVICE and nothing else in the library is affected. */
int caster_hip_reduce(caster_hip_ctx* const* ctxs, float* const* accum_dev, int n, size_t n_floats);

/*
 * The sample loop of job_func (src/dispatcher.c:12-27) for every pixel and the
 * samples [sample_begin, sample_begin + sample_count): linear RGB sums are
 * ADDED into accum_dev, a DEVICE buffer of width*height*4 floats indexed
 * [(j*width + i)*4 + c] with j = 0 the image bottom row; channel 3 gains
 * sample_count.  Random draws are keyed by (seed, j*width+i, sample), so any
 * split of the sample range over calls or GPUs renders the same paths.
 * Asynchronous: returns once the work is queued on the context's stream.
 *
 * Range of the sums.  Within a call a pixel's radiance is summed in 64-bit fixed point (units of
 * 2^-32; integer adds commute, which makes a frame bit-reproducible) and converted to float once at
 * the end.  The radiance a finished path contributes (product of the textures' colours along the
 * path and the emission, src/caster.c:451-477) is clamped to [0, 1e9] per channel on its way in, a
 * NaN counts as 0, and a channel's sum over the samples of ONE call must stay below 2^32 (4.29e9):
 * beyond that the integer wraps silently.  The reference would carry an inf or NaN into the pixel
 * instead (src/dispatcher.c:24-30).  Scenes with emitters brighter than 1e9, or more than 4e9 of
 * summed radiance per pixel and call, are outside what this entry point represents; split the
 * sample range over calls for the latter.
 */
int caster_hip_render(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height,
                      int sample_begin, int sample_count, int max_depth,
                      uint32_t seed, uint32_t flags, float* accum_dev);

/* caster_hip_render for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc, code that must not allocate because a HIP graph is being captured elsewhere
 * in the process).  It has no counterpart in the reference.  The call itself is an eager call: it
 * records the two timing events that caster_hip_render_stats / _info read, and those two wait on the
 * stream -- neither is allowed ON a capturing stream, and the call has not been exercised there.
 * Same arguments, same result bit for bit, same asynchrony; it calls no hipMalloc/hipFree or any
 * other allocating routine: the scratch that caster_hip_render keeps in buffers it allocates on
 * first use or with the context (the per-pixel fixed-point sums of the call, the ray and trip
 * counters, the work queue's head, the rings in which paths wait between their first and their
 * second ray) lives in a static device array of CASTER_HIP_NOALLOC_MAX_PIXELS * 24 + 32 +
 * CASTER_HIP_NOALLOC_MAX_WAVES * 3840 bytes (24 MiB + 15 MiB) that every GPU the library runs on holds
 * until the process ends.  Frames of more pixels are refused with CASTER_HIP_ERR_ARG, and so is a GPU
 * that runs more than CASTER_HIP_NOALLOC_MAX_WAVES waves of the render kernel at a time (16 per
 * compute unit: 4 096 on the 256 units of MI355X).  All contexts on one GPU share that GPU's array: calls that
 * overlap on one GPU (two streams, two contexts, two threads) race; calls in stream order do not,
 * each clears what it needs.  caster_hip_render_stats / _info afterwards report this call. */
#define CASTER_HIP_NOALLOC_MAX_PIXELS (1u << 20)
#define CASTER_HIP_NOALLOC_MAX_WAVES 4096u
int caster_hip_render_noalloc(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height,
                              int sample_begin, int sample_count, int max_depth,
                              uint32_t seed, uint32_t flags, float* accum_dev);

/* Waits for the last render on this context.  rays = ray_color calls with
 * depth > 0 (src/caster.c:439); kernel_ms = HIP-event time of the path-tracing
 * kernel alone.  Either pointer may be NULL. */
int caster_hip_render_stats(caster_hip_ctx* ctx, uint64_t* rays, float* kernel_ms);

/* What the last render on this context ran (waits for it), for measurement: the reference publishes
 * per-thread ray counts and times the same way (src/dispatcher.c:36-40, src/display.c:190-219).
 * loop_kind: 0 sphere filter on the vector unit, 1 the same with the sphere list streamed through
 * LDS in tiles, 2 filter on the matrix cores, 3 the same for lists of more than 512 spheres.
 * wave_trips counts the trips of all waves through the sphere loop: each trip meets 128 ray slots
 * with every sphere of the list (in the matrix loop: mfma_blocks blocks of 32 spheres x 4 blocks of
 * 32 rays x 2 v_mfma_f32_32x32x16_f16).  The loop traces a path from its second ray on; the camera
 * rays are traced in batches of 64 against the sphere list of their 8x8-pixel tile, and
 * primary_batches counts those batches. */
typedef struct {
    int32_t loop_kind;
    int32_t mfma_blocks;
    int32_t n_spheres;
    float kernel_ms;
    uint64_t rays;
    uint64_t wave_trips;
    uint64_t primary_batches;
} caster_render_info;
int caster_hip_render_info(caster_hip_ctx* ctx, caster_render_info* out);

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
/* The small vector and optics functions, item k reading in8[8k..] and writing out3[3k..]:
 *   CASTER_MATH_REFLECTANCE  reflectance(cosine = in[0], ratio = in[1])            src/common.c:170-175  -> out[0]
 *   CASTER_MATH_REFRACT      v3_f32_refract(uv = in[0..2], n = in[3..5], ratio = in[6]) with the
 *                            cos_theta of src/material.c:30 in in[7]                src/common.c:163-168  -> out[0..2]
 *   CASTER_MATH_REFLECT      v3_f32_reflect(v = in[0..2], n = in[3..5])             src/common.c:158-161  -> out[0..2]
 *   CASTER_MATH_UV_SPHERE    get_uv_coord of a sphere's outward normal in[0..2]     src/material.c:45-56  -> out[0..1]
 *   CASTER_MATH_NEAR_ZERO    v3_f32_near_zero(in[0..2])                             src/common.c:152-156  -> out[0] = 0 or 1
 *   CASTER_MATH_TO_UNIT      v3_f32_to_unit(in[0..2])                               src/common.c:89-97    -> out[0..2] */
enum { CASTER_MATH_REFLECTANCE = 0, CASTER_MATH_REFRACT = 1, CASTER_MATH_REFLECT = 2, CASTER_MATH_UV_SPHERE = 3,
       CASTER_MATH_NEAR_ZERO = 4, CASTER_MATH_TO_UNIT = 5 };
int caster_hip_test_math(caster_hip_ctx* ctx, uint32_t flags, int fn, int n, const float* in8, float* out3);
/* Audit of the conservative sphere filter that stands in front of hit_sphere in the render kernel
 * (src/caster.c:204 tests every sphere; the kernel runs the exact test only on the filter's
 * candidates, so a filter that rejects a sphere the exact test accepts would drop a hit).  For ray k
 * and every sphere of the uploaded scene that goes through the filter: accepted[k] = spheres the
 * exact test reports (any root >= tmin), missed[k] = those of them the filter rejected -- must be 0
 * -- and slack[k] = the smallest (key - threshold) / (|o'|^2 + |c'|^2 + r^2) over the accepted ones
 * (+inf if none): the filter's margin g minus its arithmetic error on the most critical pair, to
 * hold against g.  The filter runs as a render call with these flags would run it: on the matrix
 * cores where the scene qualifies, on the vector unit otherwise or with CASTER_HIP_VECTOR_LOOP. */
int caster_hip_filter_audit(caster_hip_ctx* ctx, uint32_t flags, int n, const float* ray6,
                            uint32_t* accepted, uint32_t* missed, float* slack);
/* First n 23-bit draws of the stream keyed (seed, pixel, sample). */
int caster_hip_test_rng(caster_hip_ctx* ctx, uint32_t seed, uint32_t pixel, uint32_t sample,
                        int n, uint32_t* out23);

#ifdef __cplusplus
}
#endif
#endif
