/* Command-line host (reference: src/main.c:58-124).  Same four positional
 * arguments -- height, samples per pixel, max depth, "threads" (here: GPUs) --
 * same camera, same output file; the preview window is left out.
 *
 *   craytracer <height> <spp> <max_depth> <gpus> [--scene 0|rtiow|perlin|stress:N]
 *              [--seed S] [--strict | --fast] [--batch B] [--jobs J] [--texture file.bmp] [--out file.bmp]
 *              [--checkpoint file] [--resume file]
 *
 * --checkpoint writes the frame's radiance sums and sample ranges; --resume starts from such a
 * file and renders only the samples of [0, spp) it lacks (e.g. 100 spp today, 500 tomorrow).
 */
#include "craytracer.h"

#include <stdlib.h>
#include <string.h>

int main(int argc, char* argv[])
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <height> <spp> <max_depth> <gpus> [--scene 0|rtiow|perlin|stress:N] [--seed S] "
                        "[--strict | --fast] [--batch B] [--jobs J] [--texture file.bmp] [--out file.bmp] [--checkpoint file] "
                        "[--resume file]\n", argv[0]);
        return 2;
    }
    const f32 aspect = 16.0f / 9.0f;                 /* src/main.c:64-68 */
    const i32 height = atoi(argv[1]);
    const i32 width = (i32)(height * aspect);
    const i32 spp = atoi(argv[2]);
    const i32 max_depth = atoi(argv[3]);
    const u8 gpus = (u8)atoi(argv[4]);               /* src/main.c:89 */

    const char* scene_name = "0";
    const char* out_name = "output.bmp";             /* src/main.c:113 */
    const char* texture_name = NULL;
    const char* checkpoint_name = NULL;
    const char* resume_name = NULL;
    u32 seed = 0xC0FFEEu;
    int strict = 1;                                  /* the reference's arithmetic; --fast trades it for speed */
    i32 batch = 0, jobs = 0;
    for (int k = 5; k < argc; k++) {
        if (!strcmp(argv[k], "--scene") && k + 1 < argc) scene_name = argv[++k];
        else if (!strcmp(argv[k], "--seed") && k + 1 < argc) seed = (u32)strtoul(argv[++k], NULL, 0);
        else if (!strcmp(argv[k], "--strict")) strict = 1;
        else if (!strcmp(argv[k], "--fast")) strict = 0;
        else if (!strcmp(argv[k], "--batch") && k + 1 < argc) batch = atoi(argv[++k]);
        else if (!strcmp(argv[k], "--jobs") && k + 1 < argc) jobs = atoi(argv[++k]);
        else if (!strcmp(argv[k], "--texture") && k + 1 < argc) texture_name = argv[++k];
        else if (!strcmp(argv[k], "--out") && k + 1 < argc) out_name = argv[++k];
        else if (!strcmp(argv[k], "--checkpoint") && k + 1 < argc) checkpoint_name = argv[++k];
        else if (!strcmp(argv[k], "--resume") && k + 1 < argc) resume_name = argv[++k];
        else { fprintf(stderr, "unknown argument: %s\n", argv[k]); return 2; }
    }
    if (height < 2 || width < 2 || spp < 1) { fprintf(stderr, "height >= 2 and spp >= 1 are required\n"); return 2; }

    cam_info cam = crt_default_camera(aspect);       /* src/main.c:71-78 */

    img_buffer* image = new_image_buffer((u32)width, (u32)height, 4);   /* src/main.c:81-84 */
    v3_f32 black = {{{0.0f, 0.0f, 0.0f}}};
    image_buffer_set_all(image, black);

    crt_scene* scene = NULL;
    if (!strcmp(scene_name, "0")) {
        /* src/main.c:46 loads "earthmap.bmp", which the reference repository does
         * not ship: use --texture, else a generated test texture. */
        img_buffer* earth = texture_name ? read_bitmap_image(texture_name) : crt_make_test_texture(256, 128);
        if (texture_name && !earth) { fprintf(stderr, "cannot read %s\n", texture_name); return 1; }
        scene = crt_scene_0(earth);
    } else if (!strcmp(scene_name, "rtiow")) {
        scene = crt_scene_rtiow(1);
    } else if (!strcmp(scene_name, "perlin")) {
        scene = crt_scene_perlin(1);
    } else if (!strncmp(scene_name, "stress:", 7)) {
        scene = crt_scene_stress(atoi(scene_name + 7), 2);
    }
    if (!scene) { fprintf(stderr, "unknown scene %s\n", scene_name); return 2; }

    ray_dispatcher dispatcher = new_ray_dispatcher(gpus, gpus, width, height, spp, max_depth, &cam, image,
                                                   &scene->world);           /* src/main.c:90-94 */
    ray_dispatcher_set_seed(&dispatcher, seed);
    ray_dispatcher_set_strict(&dispatcher, strict);
    ray_dispatcher_set_batch(&dispatcher, batch);
    if (jobs > 0) ray_dispatcher_set_jobs(&dispatcher, jobs);      /* more jobs than GPUs: they share the devices */
    if (checkpoint_name) ray_dispatcher_set_checkpoint(&dispatcher, checkpoint_name);
    if (resume_name && ray_dispatcher_resume(&dispatcher, resume_name) != 0) {
        free_ray_dispatcher(&dispatcher);
        return 1;
    }

    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    ray_dispatcher_run_jobs(&dispatcher);                                     /* src/main.c:100 */
    ray_dispatcher_worker_fence(&dispatcher);                                 /* src/main.c:107 */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const int status = dispatcher.status;
    if (dispatcher.thread_count > 1 && dispatcher.fence_sum_how)
        printf("Fence: %d buffers summed in %.3f ms (%s)\n", (int)dispatcher.thread_count, dispatcher.fence_sum_us * 1e-3,
               dispatcher.fence_sum_how);
    u64 rays = 0;
    for (int g = 0; g < dispatcher.thread_count; g++) rays += dispatcher.rays_total[g];
    free_ray_dispatcher(&dispatcher);

    const double time_spent = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    printf("App time: %lf\n", time_spent);                                    /* src/main.c:111 */
    if (!status) {
        printf("Rays: %llu  (%.1f Mrays/s)\n", rays, time_spent > 0 ? rays / time_spent * 1e-6 : 0.0);
        write_img_buffer_to_file(image, out_name);                            /* src/main.c:113 */
    }
    crt_scene_free(scene);
    free_image_buffer(image);
    return status ? 1 : 0;
}
