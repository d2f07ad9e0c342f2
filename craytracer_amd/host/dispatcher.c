/* GPU re-host of the dispatcher job API (reference: src/dispatcher.c,
 * inc/dispatcher.h).  The reference splits the frame into row blocks, one
 * Win32 pool thread each, and every thread runs the per-pixel sample loop.
 * Here a job is a GPU: job g renders samples [g*spp/G, (g+1)*spp/G) of every
 * pixel through libcaster_hip, in batches so that progress counters can be
 * published and a cancel request (atomic_window_running = 0) is honoured
 * between batches; the fence sums the per-GPU radiance sums in GPU order and
 * resolves them into the image buffer. */
#include "craytracer.h"
#include "caster_hip.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

typedef struct {
    ray_dispatcher* rd;
    int id, device;
    crt_range seg[CRT_MAX_RANGES];   /* the samples this job renders, in order */
    int n_seg;
    crt_range done[CRT_MAX_RANGES];  /* what it has rendered (shorter than seg after a cancel) */
    caster_hip_ctx* ctx;
    float* accum_dev;
    int samples_done;
    int status;
    int thread_started;              /* pthread_create succeeded: the fence joins only these */
    char err[256];
} gpu_job;

typedef struct {
    int n_jobs;
    int n_dev;
    int started;
    pthread_t threads[MAX_THREADS];
    gpu_job jobs[MAX_THREADS];
    crt_flat_scene* flat;
    u32 seed;
    int strict;
    i32 batch;
    crt_checkpoint* resume;          /* sums and sample ranges of an earlier run, or NULL */
    char* checkpoint_path;           /* where the fence writes the frame's checkpoint, or NULL */
} gpu_backend;

/* Identity of a frame for checkpoints: the scene as uploaded and the camera's 19 floats. */
static u64 frame_hash(const gpu_backend* be, const ray_dispatcher* rd)
{
    return crt_hash_bytes(crt_flat_hash(be->flat), rd->cam, sizeof(cam_info));
}

static u64 now_us(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (u64)tv.tv_sec * 1000000ull + (u64)tv.tv_usec;
}

static void job_fail(gpu_job* job, int status)
{
    job->status = status;
    strncpy(job->err, caster_hip_last_error(), sizeof job->err - 1);
    job->err[sizeof job->err - 1] = 0;
    fprintf(stderr, "GPU job #%d (device %d): %s\n", job->id, job->device, job->err);
}

static void* job_main(void* arg)
{
    gpu_job* job = (gpu_job*)arg;
    ray_dispatcher* rd = job->rd;
    gpu_backend* be = (gpu_backend*)rd->backend;
    const int id = job->id;
    const size_t n_px = (size_t)rd->block_width * rd->block_height;
    const size_t accum_bytes = n_px * 4 * sizeof(float);
    int rc;

    int n_obj = 0, n_mat = 0, n_img = 0;
    crt_flat_counts(be->flat, &n_obj, &n_mat, &n_img);
    if ((rc = caster_hip_create(job->device, &job->ctx)) ||
        (rc = caster_hip_upload_scene(job->ctx, n_obj, (const caster_object*)crt_flat_objects(be->flat), n_mat,
                                      (const caster_material*)crt_flat_materials(be->flat), n_img,
                                      (const caster_image*)crt_flat_images(be->flat))) ||
        (crt_flat_noise(be->flat) &&
         (rc = caster_hip_set_noise(job->ctx, (const caster_noise*)crt_flat_noise(be->flat)))) ||
        (rc = caster_hip_malloc(job->ctx, accum_bytes, (void**)&job->accum_dev)) ||
        (rc = caster_hip_memset_zero(job->ctx, job->accum_dev, accum_bytes))) {
        job_fail(job, rc);
        goto done;
    }

    /* the first job continues the sums of the run that is resumed */
    if (id == 0 && be->resume && (rc = caster_hip_upload(job->ctx, job->accum_dev, be->resume->sums, accum_bytes))) {
        job_fail(job, rc);
        goto done;
    }

    caster_camera cam;
    memcpy(&cam, rd->cam, sizeof cam);   /* cam_info and caster_camera share one layout */
    for (int k = 0; k < job->n_seg; k++) {
        const int s_begin = job->seg[k].begin, s_count = job->seg[k].count;
        const i32 batch = be->batch > 0 ? be->batch : s_count;
        job->done[k].begin = s_begin;
        job->done[k].count = 0;
        for (int s = s_begin; s < s_begin + s_count; s += batch) {
            /* src/dispatcher.c:42-45: stop when the window was closed */
            if (!__atomic_load_n(&rd->atomic_window_running, __ATOMIC_RELAXED)) break;
            const int cnt = (s + batch <= s_begin + s_count) ? batch : s_begin + s_count - s;
            const u64 t0 = now_us();
            uint64_t rays = 0;
            if ((rc = caster_hip_render(job->ctx, &cam, rd->block_width, rd->block_height, s, cnt, rd->max_depth,
                                        be->seed, be->strict ? CASTER_HIP_STRICT : 0u, job->accum_dev)) ||
                (rc = caster_hip_render_stats(job->ctx, &rays, NULL))) {
                job_fail(job, rc);
                goto done;
            }
            const u64 us = now_us() - t0;
            job->samples_done += cnt;
            job->done[k].count += cnt;
            /* src/dispatcher.c:36-40, per batch instead of per pixel */
            __atomic_store_n(&rd->atomic_spp_counter[id], us ? us : 1, __ATOMIC_RELAXED);
            __atomic_fetch_add(&rd->atomic_spp_counter_total[id], us, __ATOMIC_RELAXED);
            __atomic_store_n(&rd->atomic_spp_rays[id], (i32)rays, __ATOMIC_RELAXED);
            __atomic_fetch_add(&rd->atomic_spp_rays_total[id], (i32)rays, __ATOMIC_RELAXED);
            __atomic_fetch_add(&rd->rays_total[id], (u64)rays, __ATOMIC_RELAXED);
        }
    }

    /* the sums stay on the device; the fence adds them GPU to GPU once every job's work is complete */
    if ((rc = caster_hip_synchronize(job->ctx))) job_fail(job, rc);

done:
    printf("Thread #%d: Job finished!\n", id);                       /* src/dispatcher.c:50 */
    __atomic_store_n(&rd->atomic_spp_counter[id], 1, __ATOMIC_RELAXED);  /* src/dispatcher.c:51-52 */
    __atomic_store_n(&rd->atomic_spp_rays[id], 0, __ATOMIC_RELAXED);
    return NULL;
}

ray_dispatcher new_ray_dispatcher(u8 num_threads, u8 blocks, i32 width, i32 height, i32 spp, i32 max_depth,
                                  cam_info* cam, img_buffer* img, hit_list* list)
{
    (void)blocks;   /* the row-block count has no meaning for sample partitions */
    ray_dispatcher d;
    memset(&d, 0, sizeof d);
    if (num_threads > MAX_THREADS) num_threads = MAX_THREADS;      /* src/dispatcher.c:63 */
    int n_dev = 0;
    if (caster_hip_device_count(&n_dev) != CASTER_HIP_OK || n_dev <= 0) {
        fprintf(stderr, "new_ray_dispatcher: %s\n", caster_hip_last_error());
        d.status = CASTER_HIP_ERR_DEVICE;
        n_dev = 0;
    }
    int n_jobs = num_threads < 1 ? 1 : num_threads;
    if (n_jobs > n_dev) n_jobs = n_dev;
    if (spp > 0 && n_jobs > spp) n_jobs = spp;

    gpu_backend* be = (gpu_backend*)calloc(1, sizeof(gpu_backend));
    d.backend = be;
    d.thread_count = (u8)n_jobs;
    d.dispatch_mode = DMODE_PARALLELW;
    d.last_id = 0;
    d.max_depth = max_depth;
    d.list = list;
    d.atomic_window_running = 1;
    d.block_size = n_jobs > 0 ? (spp + n_jobs - 1) / n_jobs : 0;
    d.block_width = width;
    d.block_height = height;
    d.spp = spp;
    d.cam = cam;
    d.image = img;
    if (be) {
        be->n_jobs = n_jobs;
        be->n_dev = n_dev > 0 ? n_dev : 1;
        be->seed = 0xC0FFEEu;
        be->strict = 1;                       /* the reference's operation order: the parity-checked mode */
        be->batch = 0;
        be->flat = crt_flatten(list);
        if (!be->flat) {
            fprintf(stderr, "new_ray_dispatcher: the hit_list holds an object type the renderer does not define\n");
            d.status = CASTER_HIP_ERR_ARG;
        }
    } else {
        d.status = CASTER_HIP_ERR_ARG;
    }
    return d;
}

void ray_dispatcher_set_seed(ray_dispatcher* rd, u32 seed) { if (rd->backend) ((gpu_backend*)rd->backend)->seed = seed; }
void ray_dispatcher_set_strict(ray_dispatcher* rd, int strict) { if (rd->backend) ((gpu_backend*)rd->backend)->strict = strict; }
void ray_dispatcher_set_batch(ray_dispatcher* rd, i32 n) { if (rd->backend) ((gpu_backend*)rd->backend)->batch = n; }

/* More jobs than devices: job g runs on device g % devices, each with its own context and sample
 * range.  new_ray_dispatcher() never does this by itself. */
void ray_dispatcher_set_jobs(ray_dispatcher* rd, i32 jobs)
{
    gpu_backend* be = (gpu_backend*)rd->backend;
    if (!be || be->started || rd->status) return;
    if (jobs < 1) jobs = 1;
    if (jobs > MAX_THREADS) jobs = MAX_THREADS;
    if (rd->spp > 0 && jobs > rd->spp) jobs = rd->spp;
    be->n_jobs = jobs;
    rd->thread_count = (u8)jobs;
    rd->block_size = (rd->spp + jobs - 1) / jobs;
}

void ray_dispatcher_run_jobs(ray_dispatcher* rd)
{
    gpu_backend* be = (gpu_backend*)rd->backend;
    if (!be || rd->status || be->started) return;
    be->started = 1;
    const int G = be->n_jobs;
    /* the samples still to render: all of [0, spp), or what the resumed run lacks */
    crt_range todo[CRT_MAX_RANGES];
    const int n_todo = crt_ranges_complement(be->resume ? be->resume->ranges : NULL, be->resume ? be->resume->n_ranges : 0,
                                             rd->spp, todo, CRT_MAX_RANGES);
    if (n_todo < 0) {
        /* more gaps than the range table holds: rendering nothing and writing an image would be a silent lie */
        fprintf(stderr, "ray_dispatcher_run_jobs: the resumed frame's missing samples do not fit %d ranges\n", CRT_MAX_RANGES);
        rd->status = CASTER_HIP_ERR_ARG;
        be->started = 0;
        return;
    }
    const long long total = n_todo > 0 ? crt_ranges_total(todo, n_todo) : 0;
    for (int g = 0; g < G; g++) {
        gpu_job* job = &be->jobs[g];
        memset(job, 0, sizeof *job);
        job->rd = rd;
        job->id = g;
        job->device = g % be->n_dev;
        const i32 first = (i32)((total * g) / G);
        const i32 count = (i32)((total * (g + 1)) / G) - first;
        job->n_seg = n_todo > 0 ? crt_ranges_slice(todo, n_todo, first, count, job->seg, CRT_MAX_RANGES) : 0;
        if (job->n_seg < 0) {
            fprintf(stderr, "ray_dispatcher_run_jobs: job %d's share of the samples does not fit %d ranges\n", g, CRT_MAX_RANGES);
            job->n_seg = 0;
            job->status = CASTER_HIP_ERR_ARG;
            rd->status = CASTER_HIP_ERR_ARG;
            continue;                                                 /* not launched */
        }
        if (pthread_create(&be->threads[g], NULL, job_main, job) != 0) {
            job->status = CASTER_HIP_ERR_STATE;
            rd->status = CASTER_HIP_ERR_STATE;
        } else {
            job->thread_started = 1;
        }
        if (rd->last_id <= rd->thread_count) rd->last_id++;           /* src/dispatcher.c:111 */
    }
}

/* The order in which a frame's buffers are added, as data (so that it can be checked without a GPU).
 * crt_sum_plan_local: the first job on each device leads it; every other job is added into its
 * device's lead (steps {dst job, src job}, in job order).  lead[] receives the leads' job numbers in
 * the order the devices first appear; job 0 is always lead[0].  Returns the number of leads.
 * crt_sum_plan_tree: the pairwise tree over n_lead buffers, rounds of stride 1, 2, 4, ...: in a round
 * lead l (a multiple of 2*stride) takes lead l + stride.  Steps hold positions in lead[]; the
 * adds of one round touch disjoint buffers and may run concurrently; after the last round lead[0]
 * holds the sum.  Returns the number of steps. */
int crt_sum_plan_local(const int* device, int n_jobs, int* lead, crt_sum_step* steps, int* n_steps)
{
    int n_lead = 0, n = 0;
    for (int g = 0; g < n_jobs; g++) {
        int l = 0;
        while (l < n_lead && device[lead[l]] != device[g]) l++;
        if (l == n_lead) { lead[n_lead++] = g; continue; }
        steps[n].dst = lead[l]; steps[n].src = g; steps[n].round = 0;
        n++;
    }
    *n_steps = n;
    return n_lead;
}

int crt_sum_plan_tree(int n_lead, crt_sum_step* steps)
{
    int n = 0, round = 1;
    for (int stride = 1; stride < n_lead; stride *= 2, round++)
        for (int l = 0; l + stride < n_lead; l += 2 * stride) {
            steps[n].dst = l; steps[n].src = l + stride; steps[n].round = round;
            n++;
        }
    return n;
}

/* The sum of the jobs' radiance sums into job 0's buffer; they are linear and must be added before
 * the scale and the gamma (src/dispatcher.c:29-30).  Jobs that share a GPU are added into the first
 * job on that GPU with an add kernel; the per-GPU sums then meet in ONE RCCL reduce over the node's
 * xGMI links (caster_hip_reduce), everything queued on the GPUs' streams without a host wait in
 * between.  Without RCCL the per-GPU sums are added pairwise over the peer links instead -- a tree of
 * log2(G) rounds whose adds run concurrently -- and a line on stderr says so.  A failed
 * caster_hip_reduce has queued nothing (it ends its communicators before any part of the collective
 * is launched), so the streams are free for the fallback. */
static int sum_jobs(gpu_backend* be, size_t n_floats, const char** how)
{
    int device[MAX_THREADS], lead[MAX_THREADS];
    crt_sum_step steps[MAX_THREADS];
    int n_steps = 0, rc = 0;
    *how = "one buffer";
    for (int g = 0; g < be->n_jobs; g++) device[g] = be->jobs[g].device;
    const int n_lead = crt_sum_plan_local(device, be->n_jobs, lead, steps, &n_steps);
    for (int k = 0; k < n_steps; k++) {
        gpu_job* dst = &be->jobs[steps[k].dst];
        gpu_job* src = &be->jobs[steps[k].src];
        *how = "add kernels";
        if ((rc = caster_hip_add_from_device(dst->ctx, dst->accum_dev, src->device, src->accum_dev, n_floats))) return rc;
    }
    if (n_lead > 1) {
        caster_hip_ctx* ctxs[MAX_THREADS];
        float* bufs[MAX_THREADS];
        for (int l = 0; l < n_lead; l++) { ctxs[l] = be->jobs[lead[l]].ctx; bufs[l] = be->jobs[lead[l]].accum_dev; }
        *how = "RCCL reduce";
        rc = caster_hip_reduce(ctxs, bufs, n_lead, n_floats);
        if (rc) {
            fprintf(stderr, "fence: %s; adding the %d GPUs' sums pairwise over the peer links instead\n", caster_hip_last_error(), n_lead);
            *how = "pairwise peer adds";
            rc = 0;
            const int n_tree = crt_sum_plan_tree(n_lead, steps);
            int round = 0;
            for (int k = 0; k < n_tree && !rc; k++) {
                if (steps[k].round != round) {                   /* a new round: its sources must be complete */
                    round = steps[k].round;
                    for (int l = 0; l < n_lead && !rc; l++) rc = caster_hip_synchronize(ctxs[l]);
                }
                if (!rc)
                    rc = caster_hip_add_from_device(ctxs[steps[k].dst], bufs[steps[k].dst], be->jobs[lead[steps[k].src]].device,
                                                    bufs[steps[k].src], n_floats);
            }
        }
    }
    if (!rc) rc = caster_hip_synchronize(be->jobs[0].ctx);
    return rc;
}

clock_t ray_dispatcher_worker_fence(ray_dispatcher* rd)
{
    gpu_backend* be = (gpu_backend*)rd->backend;
    if (!be || !be->started) return clock();
    const size_t n_px = (size_t)rd->block_width * rd->block_height;
    for (int g = 0; g < be->n_jobs; g++) {
        if (be->jobs[g].thread_started) pthread_join(be->threads[g], NULL);
        if (be->jobs[g].status && !rd->status) rd->status = be->jobs[g].status;
    }
    /* the sample ranges the frame now holds */
    static _Thread_local crt_range held[CRT_MAX_RANGES * 2];
    int n_held = 0;
    if (be->resume)
        for (int k = 0; k < be->resume->n_ranges; k++) held[n_held++] = be->resume->ranges[k];
    for (int g = 0; g < be->n_jobs; g++) {
        int m = crt_ranges_normalize(be->jobs[g].done, be->jobs[g].n_seg);
        if (n_held + m > CRT_MAX_RANGES * 2) n_held = crt_ranges_normalize(held, n_held);
        for (int k = 0; k < m && n_held < CRT_MAX_RANGES * 2; k++) held[n_held++] = be->jobs[g].done[k];
    }
    n_held = crt_ranges_normalize(held, n_held);
    if (!rd->status && be->n_jobs > 0) {
        gpu_job* root = &be->jobs[0];
        int done = root->samples_done + (be->resume ? crt_ranges_total(be->resume->ranges, be->resume->n_ranges) : 0);
        for (int g = 1; g < be->n_jobs; g++) done += be->jobs[g].samples_done;
        const char* how = "";
        const u64 t_sum = now_us();
        int rc = sum_jobs(be, n_px * 4, &how);
        rd->fence_sum_us = now_us() - t_sum;
        rd->fence_sum_how = how;
        if (!rc && done > 0 && rd->image && rd->image->data)
            rc = caster_hip_resolve(root->ctx, root->accum_dev, rd->block_width, rd->block_height, done, rd->image->data);
        if (!rc && be->checkpoint_path) {
            crt_checkpoint cp;
            memset(&cp, 0, sizeof cp);
            cp.width = (u32)rd->block_width; cp.height = (u32)rd->block_height;
            cp.seed = be->seed; cp.strict = be->strict ? 1 : 0; cp.max_depth = rd->max_depth;
            cp.scene_hash = frame_hash(be, rd);
            cp.n_ranges = n_held < CRT_MAX_RANGES ? n_held : CRT_MAX_RANGES;
            memcpy(cp.ranges, held, (size_t)cp.n_ranges * sizeof(crt_range));
            cp.sums = (f32*)malloc(n_px * 4 * sizeof(f32));
            if (!cp.sums || n_held > CRT_MAX_RANGES) {
                fprintf(stderr, "checkpoint %s: not written (out of memory or too many sample ranges)\n", be->checkpoint_path);
                rd->status = CASTER_HIP_ERR_ARG;
            } else if ((rc = caster_hip_download(root->ctx, cp.sums, root->accum_dev, n_px * 4 * sizeof(f32))) == 0 &&
                       crt_checkpoint_write(be->checkpoint_path, &cp) != 0) {
                fprintf(stderr, "checkpoint %s: write failed\n", be->checkpoint_path);
                rd->status = CASTER_HIP_ERR_ARG;
            }
            free(cp.sums);
        }
        if (rc) { job_fail(root, rc); rd->status = rc; }
    }
    for (int g = 0; g < be->n_jobs; g++) {
        gpu_job* job = &be->jobs[g];
        if (job->ctx) {
            if (job->accum_dev) caster_hip_free(job->ctx, job->accum_dev);
            caster_hip_destroy(job->ctx);
        }
        memset(job, 0, sizeof *job);
    }
    be->started = 0;
    return clock();
}

void ray_dispatcher_set_checkpoint(ray_dispatcher* rd, const char* path)
{
    gpu_backend* be = (gpu_backend*)rd->backend;
    if (!be || be->started) return;
    free(be->checkpoint_path);
    be->checkpoint_path = path ? strdup(path) : NULL;
}

int ray_dispatcher_resume(ray_dispatcher* rd, const char* path)
{
    gpu_backend* be = (gpu_backend*)rd->backend;
    if (!be || be->started || rd->status) return -1;
    crt_checkpoint* cp = crt_checkpoint_read(path);
    if (!cp) { fprintf(stderr, "resume: cannot read %s\n", path ? path : "(null)"); return -1; }
    if (cp->width != (u32)rd->block_width || cp->height != (u32)rd->block_height || cp->seed != be->seed ||
        (cp->strict != 0) != (be->strict != 0) || cp->max_depth != rd->max_depth ||
        cp->scene_hash != frame_hash(be, rd)) {
        fprintf(stderr, "resume: %s belongs to another frame (size, seed, arithmetic mode, depth, camera or scene differ)\n", path);
        crt_checkpoint_free(cp);
        return -1;
    }
    crt_checkpoint_free(be->resume);
    be->resume = cp;
    return 0;
}

void free_ray_dispatcher(ray_dispatcher* rd)
{
    gpu_backend* be = (gpu_backend*)rd->backend;
    if (!be) return;
    if (be->started) ray_dispatcher_worker_fence(rd);
    crt_checkpoint_free(be->resume);
    free(be->checkpoint_path);
    crt_flat_free(be->flat);
    free(be);
    rd->backend = NULL;
}
