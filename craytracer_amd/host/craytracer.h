/*
 * craytracer.h -- host-side mirror of the CRaytracer interfaces that sit on
 * either side of the per-pixel hot path, so that code written against the
 * reference's headers (scene setup, camera, image buffer, dispatcher job API)
 * compiles against this one and runs its sample loop on MI355X GPUs through
 * libcaster_hip.
 *
 * Type and function names, field names, argument order and struct layouts
 * follow the reference headers cited at each declaration.  What changed is
 * behind the dispatcher: ray_dispatcher_run_jobs() starts one host thread per
 * GPU instead of one Win32 pool thread per row block, and each "job" is a
 * sample partition of the whole frame rather than a block of rows.
 */
#ifndef CRAYTRACER_HOST_H
#define CRAYTRACER_HOST_H

#include <stdint.h>
#include <stdio.h>
#include <time.h>

#ifdef __cplusplus
extern "C" {
#endif

/* inc/common.h:14-20 */
typedef float f32;
typedef unsigned char u8;
typedef unsigned int u32;
typedef unsigned long long u64;
typedef signed char i8;
typedef signed int i16;   /* sic: the reference declares i16 as int */
typedef int i32;

#define PI 3.141592654f   /* inc/common.h:10 */

/* inc/common.h:32-56 */
typedef struct { union { struct { f32 u, v; }; f32 p[2]; }; } v2_f32;
typedef struct { union { struct { f32 x, y, z; }; f32 p[3]; }; } v3_f32;

/* inc/image.h:6-23 */
typedef struct { u8* data; u32 w, h; u8 bpp; } img_buffer;
img_buffer* new_image_buffer(u32 w, u32 h, u8 bpp);
void image_buffer_set_all(img_buffer* buffer, v3_f32 color);
void image_buffer_set_pixel(img_buffer* buffer, i32 x, i32 y, v3_f32 color);
void free_image_buffer(img_buffer* buffer);
void write_img_buffer_to_file(img_buffer* buffer, const char* filename);
/* The reference reads bitmaps through Win32 LoadImage (src/image.c:73-96); this
 * one parses uncompressed 24/32-bit BMP files itself and returns rows bottom-up,
 * bytes B,G,R as a DIB section holds them.  NULL if the file cannot be read. */
img_buffer* read_bitmap_image(const char* filename);

/* inc/material.h:8-35 */
#define TEXTURE_COLOR_UNITS 8
/* inc/noise.h:4-7 declares the placeholder `noise_perlin { int todo; }` and src/noise.c is empty.
 * Here the type holds the tables of gradient ("Perlin") noise as "Ray Tracing: The Next Week"
 * builds them: 256 random unit vectors and three permutations of 0..255. */
typedef struct noise_perlin { v3_f32 ranvec[256]; u8 perm_x[256], perm_y[256], perm_z[256]; } noise_perlin;
noise_perlin* new_noise_perlin(u32 seed);
void free_noise_perlin(noise_perlin* noise);

/* `noise` is an extension of the reference's struct (last field): the tables of a TTYPE_NOISE
 * texture, whose colour is color[0] * 0.5 * (1 + sin(color[1].x * p.z + 10 * turbulence(p))).
 * All noise textures of a scene share one table. */
typedef struct texture { v3_f32 color[TEXTURE_COLOR_UNITS]; u8 type; img_buffer* image; noise_perlin* noise; } texture;
#define TTYPE_COLOR 0
#define TTYPE_CHECKER 1
#define TTYPE_IMAGE_ALBEDO 2
#define TTYPE_IMAGE_EMIT 3
#define TTYPE_NOISE 4
typedef struct material { i8 type; texture texture; f32 fuzziness; f32 n_refract; } material;
#define MTYPE_LAMBERTIAN 0
#define MTYPE_METAL 1
#define MTYPE_DIELECTRIC 2
#define MTYPE_DIFF_LIGHT 3

/* inc/caster.h:6-10,39-79 */
typedef struct { v3_f32 o, d; } ray;
typedef struct { v3_f32 center; f32 radius; material* material; } sphere;
struct rect { v3_f32 center; v2_f32 dims; material* material; };
typedef struct rect xy_rect;
typedef struct { void** head; i8* types; i32 count; } hit_list;
#define HTYPE_SPHERE 0
#define HTYPE_RECT_XY 1
hit_list new_hit_list(void);
void free_hit_list(hit_list* list);
void hit_list_add_tail(hit_list* list, void* hittable, i8 hit_type);

/* inc/camera.h:6-15 */
typedef struct { v3_f32 o, llc, h, v, u0, v0; f32 lr; } cam_info;
cam_info calculate_cam_info(v3_f32 cam_pos, v3_f32 look_at, v3_f32 cam_up, f32 cam_fov, f32 cam_ar,
                            f32 aperture, f32 focus);

/* inc/dispatcher.h:10-53.  The public fields keep their names and meaning;
 * the Win32 pool handles are replaced by an opaque backend pointer. */
#define MAX_THREADS 64
#define DMODE_PARALLELW 0
typedef struct {
    void* backend;            /* GPU jobs, one per device                        */
    u8 thread_count;          /* number of GPU jobs                              */
    u8 dispatch_mode;
    u8 last_id;
    i32 block_size;           /* samples per GPU job (was rows per job)          */
    i32 block_width;
    i32 block_height;
    i32 spp;
    i32 max_depth;
    cam_info* cam;
    img_buffer* image;
    hit_list* list;

    i16 atomic_window_running;                 /* set to 0 to cancel            */

    u64 atomic_spp_counter[MAX_THREADS];       /* us spent in the last batch    */
    i32 atomic_spp_rays[MAX_THREADS];          /* rays shot in the last batch   */
    u64 atomic_spp_counter_total[MAX_THREADS]; /* us spent so far               */
    i32 atomic_spp_rays_total[MAX_THREADS];    /* rays so far (wraps at 2^31, as the reference's i32 does) */
    u64 rays_total[MAX_THREADS];               /* the same count without the wrap */
    int status;                                /* 0, or the first libcaster_hip error */
    u64 fence_sum_us;                          /* the fence's sum of the jobs' buffers: wall time ...  */
    const char* fence_sum_how;                 /* ... and the route it took (static string)            */
} ray_dispatcher;

ray_dispatcher new_ray_dispatcher(u8 num_threads, u8 blocks, i32 width, i32 height, i32 spp, i32 max_depth,
                                  cam_info* cam, img_buffer* img, hit_list* list);
void free_ray_dispatcher(ray_dispatcher* rd);
void ray_dispatcher_run_jobs(ray_dispatcher* rd);
clock_t ray_dispatcher_worker_fence(ray_dispatcher* rd);

/* Render options the reference fixes at compile time; call before run_jobs. */
void ray_dispatcher_set_seed(ray_dispatcher* rd, u32 seed);
void ray_dispatcher_set_strict(ray_dispatcher* rd, int strict);   /* default 1: the reference's operation order (the parity-checked mode); 0 = fast */
void ray_dispatcher_set_batch(ray_dispatcher* rd, i32 samples_per_batch);
/* More jobs than devices (job g on device g % devices, own context and sample range each). */
void ray_dispatcher_set_jobs(ray_dispatcher* rd, i32 jobs);

/* The order in which the fence adds the jobs' buffers (see dispatcher.c): same-device adds into the
 * first job of each device, then one RCCL reduce over the devices' leads or, without RCCL, a pairwise
 * tree.  Pure functions of the job -> device map; exported so that the order is testable without a GPU. */
typedef struct { int dst, src, round; } crt_sum_step;
int crt_sum_plan_local(const int* device, int n_jobs, int* lead, crt_sum_step* steps, int* n_steps);
int crt_sum_plan_tree(int n_lead, crt_sum_step* steps);

/* Progressive accumulation across runs (the reference plans a "gradual image add mode",
 * src/dispatcher.c:6-7).  set_checkpoint: the fence writes the frame's radiance sums and the sample
 * ranges they hold to the file, also after a cancel.  resume: the sums of the file become the start
 * of the frame and the jobs render only the samples of [0, spp) that it lacks; returns 0, or -1 if
 * the file is unreadable or belongs to another frame size, seed, arithmetic mode, depth, camera or scene
 * (objects, materials, texture images, noise tables).
 * Call both before run_jobs. */
void ray_dispatcher_set_checkpoint(ray_dispatcher* rd, const char* path);
int ray_dispatcher_resume(ray_dispatcher* rd, const char* path);

/* Sample ranges and the checkpoint file. */
#define CRT_MAX_RANGES 256
typedef struct { i32 begin, count; } crt_range;
typedef struct {
    u32 width, height, seed;
    i32 strict, max_depth;
    u64 scene_hash;
    i32 n_ranges;
    crt_range ranges[CRT_MAX_RANGES];
    f32* sums;                   /* width * height * 4, owned */
} crt_checkpoint;
int crt_ranges_normalize(crt_range* r, int n);
i32 crt_ranges_total(const crt_range* r, int n);
int crt_ranges_complement(const crt_range* done, int n_done, i32 spp, crt_range* out, int cap);
int crt_ranges_slice(const crt_range* r, int n, i32 first, i32 count, crt_range* out, int cap);
int crt_checkpoint_write(const char* path, const crt_checkpoint* cp);
crt_checkpoint* crt_checkpoint_read(const char* path);
void crt_checkpoint_free(crt_checkpoint* cp);

/* ---- scenes ---------------------------------------------------------- */
/* A scene owns its spheres, rects, materials and textures; world lists them. */
typedef struct {
    hit_list world;
    sphere* spheres; i32 n_spheres;
    xy_rect* rects; i32 n_rects;
    material* materials; i32 n_materials;
    img_buffer* texture_image;   /* owned, may be NULL */
    noise_perlin* noise;         /* owned, may be NULL */
} crt_scene;

/* src/main.c:14-55: ground checker sphere, glass sphere, image-textured sphere,
 * emissive xy rect.  earth may be NULL: the third sphere then takes the solid
 * albedo (0.4, 0.2, 0.1) of mat1 (src/main.c:25). */
crt_scene* crt_scene_0(img_buffer* earth);
/* RTIOW final scene built from the reference's primitives, lit by an enclosing
 * emissive sphere (the reference returns black on a miss, src/caster.c:482). */
crt_scene* crt_scene_rtiow(u32 scene_seed);
/* The same layout with marble (TTYPE_NOISE) textures on the ground, the diffuse small spheres
 * and the big diffuse sphere. */
crt_scene* crt_scene_perlin(u32 scene_seed);
/* n small spheres scattered over the ground plus ground and sky spheres. */
crt_scene* crt_scene_stress(i32 n, u32 scene_seed);
void crt_scene_free(crt_scene* sc);
/* The camera of src/main.c:71-78 for the given aspect ratio. */
cam_info crt_default_camera(f32 aspect);
/* A w x h RGB test texture (continents-like bands), 3 bytes per texel. */
img_buffer* crt_make_test_texture(u32 w, u32 h);

/* ---- flattening (pointer graph -> libcaster_hip tables) -------------- */
struct crt_flat_scene;
typedef struct crt_flat_scene crt_flat_scene;
crt_flat_scene* crt_flatten(const hit_list* world);
void crt_flat_free(crt_flat_scene* fs);
int crt_flat_counts(const crt_flat_scene* fs, int* n_objects, int* n_materials, int* n_images);
const void* crt_flat_objects(const crt_flat_scene* fs);    /* caster_object[]   */
const void* crt_flat_materials(const crt_flat_scene* fs);  /* caster_material[] */
const void* crt_flat_images(const crt_flat_scene* fs);     /* caster_image[]    */
const void* crt_flat_noise(const crt_flat_scene* fs);      /* caster_noise* or NULL */
u64 crt_flat_hash(const crt_flat_scene* fs);               /* of the object, material, image and noise tables */
u64 crt_hash_bytes(u64 h, const void* data, size_t n);     /* FNV-1a, continued from h */

/* Bytes write_img_buffer_to_file() writes, into out (54 + h*stride bytes). */
long crt_bmp_encode(const img_buffer* buffer, u8* out);

#ifdef __cplusplus
}
#endif
#endif
