/* Progressive accumulation across runs.  The reference only plans it ("gradual image add mode",
 * src/dispatcher.c:6-7); with sample-range jobs it falls out of the data: the linear radiance sums
 * of a frame plus the list of sample ranges they hold are a checkpoint, and a later run renders
 * the ranges that are missing.  Because every (pixel, sample) has its own random stream, the
 * paths of the resumed frame are the paths of an uninterrupted one.
 *
 * File layout (little endian): "CRTACC01", u32 width, height, seed, i32 strict, max_depth,
 * u64 scene hash, i32 n_ranges, i32 reserved, n_ranges x {i32 begin, i32 count},
 * width*height*4 floats (R, G, B sums and the sample count per pixel, rows bottom-up as the
 * renderer numbers them). */
#include "craytracer.h"
#include "caster_hip.h"

#include <stdlib.h>
#include <string.h>

static int range_cmp(const void* a, const void* b)
{
    const crt_range* x = (const crt_range*)a;
    const crt_range* y = (const crt_range*)b;
    return (x->begin > y->begin) - (x->begin < y->begin);
}

/* Sorts, drops empty ranges, merges ranges that touch or overlap.  Returns the new length. */
int crt_ranges_normalize(crt_range* r, int n)
{
    int m = 0;
    for (int k = 0; k < n; k++)
        if (r[k].count > 0) r[m++] = r[k];
    qsort(r, (size_t)m, sizeof *r, range_cmp);
    int out = 0;
    for (int k = 0; k < m; k++) {
        if (out > 0 && r[k].begin <= r[out - 1].begin + r[out - 1].count) {
            const i32 end = r[k].begin + r[k].count;
            if (end > r[out - 1].begin + r[out - 1].count) r[out - 1].count = end - r[out - 1].begin;
        } else {
            r[out++] = r[k];
        }
    }
    return out;
}

i32 crt_ranges_total(const crt_range* r, int n)
{
    i32 t = 0;
    for (int k = 0; k < n; k++) t += r[k].count;
    return t;
}

/* The samples of [0, spp) that a normalized list does not hold.  Returns the length of out, -1 if
 * cap is too small. */
int crt_ranges_complement(const crt_range* done, int n_done, i32 spp, crt_range* out, int cap)
{
    int n = 0;
    i32 at = 0;
    for (int k = 0; k <= n_done; k++) {
        const i32 stop = (k < n_done) ? (done[k].begin < spp ? done[k].begin : spp) : spp;
        if (stop > at) {
            if (n == cap) return -1;
            out[n].begin = at;
            out[n].count = stop - at;
            n++;
        }
        if (k < n_done && done[k].begin + done[k].count > at) at = done[k].begin + done[k].count;
    }
    return n;
}

/* Entries first .. first+count-1 of the samples a list enumerates, as a list.  Returns its length,
 * -1 if cap is too small. */
int crt_ranges_slice(const crt_range* r, int n, i32 first, i32 count, crt_range* out, int cap)
{
    int m = 0;
    i32 seen = 0;
    for (int k = 0; k < n && count > 0; k++) {
        const i32 lo = first > seen ? first - seen : 0;       /* offset into r[k] */
        if (lo < r[k].count) {
            i32 take = r[k].count - lo;
            if (take > count) take = count;
            if (m == cap) return -1;
            out[m].begin = r[k].begin + lo;
            out[m].count = take;
            m++;
            first += take;
            count -= take;
        }
        seen += r[k].count;
    }
    return m;
}

/* FNV-1a, continued from h. */
u64 crt_hash_bytes(u64 h, const void* data, size_t n)
{
    const u8* p = (const u8*)data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

/* Hash of everything of the flattened scene that reaches the GPU -- objects, materials, every image
 * texture's size and texels, the noise tables: a resumed run must render the scene it started with. */
u64 crt_flat_hash(const crt_flat_scene* fs)
{
    u64 h = 1469598103934665603ull;
    if (!fs) return h;
    int n_obj = 0, n_mat = 0, n_img = 0;
    crt_flat_counts(fs, &n_obj, &n_mat, &n_img);
    h = crt_hash_bytes(h, crt_flat_objects(fs), (size_t)n_obj * sizeof(caster_object));
    h = crt_hash_bytes(h, crt_flat_materials(fs), (size_t)n_mat * sizeof(caster_material));
    const caster_image* img = (const caster_image*)crt_flat_images(fs);
    for (int k = 0; k < n_img; k++) {
        const u32 dims[3] = { img[k].w, img[k].h, img[k].bpp };
        h = crt_hash_bytes(h, dims, sizeof dims);
        h = crt_hash_bytes(h, img[k].data, (size_t)img[k].w * img[k].h * img[k].bpp);
    }
    if (crt_flat_noise(fs)) h = crt_hash_bytes(h, crt_flat_noise(fs), sizeof(caster_noise));
    return h;
}

static const char kMagic[8] = { 'C', 'R', 'T', 'A', 'C', 'C', '0', '2' };

int crt_checkpoint_write(const char* path, const crt_checkpoint* cp)
{
    if (!path || !cp || !cp->sums || cp->n_ranges < 0 || cp->n_ranges > CRT_MAX_RANGES) return -1;
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    const i32 reserved = 0;
    const size_t n = (size_t)cp->width * cp->height * 4;
    int ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&cp->width, 4, 1, f) == 1 && fwrite(&cp->height, 4, 1, f) == 1 &&
             fwrite(&cp->seed, 4, 1, f) == 1 && fwrite(&cp->strict, 4, 1, f) == 1 && fwrite(&cp->max_depth, 4, 1, f) == 1 &&
             fwrite(&cp->scene_hash, 8, 1, f) == 1 && fwrite(&cp->n_ranges, 4, 1, f) == 1 && fwrite(&reserved, 4, 1, f) == 1 &&
             fwrite(cp->ranges, sizeof(crt_range), (size_t)cp->n_ranges, f) == (size_t)cp->n_ranges &&
             fwrite(cp->sums, sizeof(f32), n, f) == n;
    if (fclose(f) != 0) ok = 0;
    return ok ? 0 : -1;
}

crt_checkpoint* crt_checkpoint_read(const char* path)
{
    FILE* f = path ? fopen(path, "rb") : NULL;
    if (!f) return NULL;
    crt_checkpoint* cp = (crt_checkpoint*)calloc(1, sizeof *cp);
    char magic[8];
    i32 reserved = 0;
    int ok = cp && fread(magic, 1, 8, f) == 8 && !memcmp(magic, kMagic, 8) && fread(&cp->width, 4, 1, f) == 1 &&
             fread(&cp->height, 4, 1, f) == 1 && fread(&cp->seed, 4, 1, f) == 1 && fread(&cp->strict, 4, 1, f) == 1 &&
             fread(&cp->max_depth, 4, 1, f) == 1 && fread(&cp->scene_hash, 8, 1, f) == 1 &&
             fread(&cp->n_ranges, 4, 1, f) == 1 && fread(&reserved, 4, 1, f) == 1 &&
             cp->n_ranges >= 0 && cp->n_ranges <= CRT_MAX_RANGES && cp->width > 0 && cp->height > 0 &&
             cp->width <= 65535u && cp->height <= 65535u &&
             fread(cp->ranges, sizeof(crt_range), (size_t)cp->n_ranges, f) == (size_t)cp->n_ranges;
    if (ok) {
        for (int k = 0; k < cp->n_ranges; k++)
            if (cp->ranges[k].begin < 0 || cp->ranges[k].count < 0 || cp->ranges[k].begin > 0x3fffffff ||
                cp->ranges[k].count > 0x3fffffff) ok = 0;
    }
    if (ok) {
        const size_t n = (size_t)cp->width * cp->height * 4;
        cp->sums = (f32*)malloc(n * sizeof(f32));
        char extra;
        ok = cp->sums && fread(cp->sums, sizeof(f32), n, f) == n && fread(&extra, 1, 1, f) == 0;   /* exact length */
    }
    fclose(f);
    if (!ok) { crt_checkpoint_free(cp); return NULL; }
    cp->n_ranges = crt_ranges_normalize(cp->ranges, cp->n_ranges);
    return cp;
}

void crt_checkpoint_free(crt_checkpoint* cp)
{
    if (!cp) return;
    free(cp->sums);
    free(cp);
}
