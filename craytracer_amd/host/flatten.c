/* Turns the pointer graph a hit_list refers to (spheres/rects -> materials ->
 * textures -> images) into the index-based tables libcaster_hip uploads.
 * List order is kept; materials and images are de-duplicated by address. */
#include "craytracer.h"
#include "caster_hip.h"

#include <stdlib.h>
#include <string.h>

struct crt_flat_scene {
    caster_object* objects; int n_objects;
    caster_material* materials; int n_materials;
    caster_image* images; int n_images;
    const material** mat_keys;
    const img_buffer** img_keys;
    const noise_perlin* noise;      /* the table the scene's noise textures share, or NULL */
    int noise_conflict;             /* two noise textures with different tables */
};

/* noise_perlin is handed to libcaster_hip as it is */
_Static_assert(sizeof(noise_perlin) == sizeof(caster_noise), "noise table layout");

static int find_or_add_image(crt_flat_scene* fs, const img_buffer* img)
{
    if (!img) return -1;
    for (int k = 0; k < fs->n_images; k++)
        if (fs->img_keys[k] == img) return k;
    const int k = fs->n_images++;
    fs->img_keys[k] = img;
    fs->images[k].data = img->data;
    fs->images[k].w = img->w; fs->images[k].h = img->h; fs->images[k].bpp = img->bpp;
    return k;
}

static int find_or_add_material(crt_flat_scene* fs, const material* m)
{
    for (int k = 0; k < fs->n_materials; k++)
        if (fs->mat_keys[k] == m) return k;
    const int k = fs->n_materials++;
    fs->mat_keys[k] = m;
    caster_material* o = &fs->materials[k];
    memset(o, 0, sizeof *o);
    o->type = m->type;
    o->tex_type = m->texture.type;
    for (int c = 0; c < 3; c++) { o->color0[c] = m->texture.color[0].p[c]; o->color1[c] = m->texture.color[1].p[c]; }
    o->fuzziness = m->fuzziness;
    o->n_refract = m->n_refract;
    if (m->texture.type == TTYPE_NOISE) {
        if (!m->texture.noise || (fs->noise && fs->noise != m->texture.noise)) fs->noise_conflict = 1;
        fs->noise = m->texture.noise;
    }
    o->image = (m->texture.type == TTYPE_IMAGE_ALBEDO || m->texture.type == TTYPE_IMAGE_EMIT)
                   ? find_or_add_image(fs, m->texture.image) : -1;
    return k;
}

crt_flat_scene* crt_flatten(const hit_list* world)
{
    if (!world || world->count < 0) return NULL;
    const size_t n = (size_t)(world->count > 0 ? world->count : 1);
    crt_flat_scene* fs = (crt_flat_scene*)calloc(1, sizeof *fs);
    if (!fs) return NULL;
    fs->objects = (caster_object*)calloc(n, sizeof(caster_object));
    fs->materials = (caster_material*)calloc(n, sizeof(caster_material));
    fs->images = (caster_image*)calloc(n, sizeof(caster_image));
    fs->mat_keys = (const material**)calloc(n, sizeof(void*));
    fs->img_keys = (const img_buffer**)calloc(n, sizeof(void*));
    if (!fs->objects || !fs->materials || !fs->images || !fs->mat_keys || !fs->img_keys) { crt_flat_free(fs); return NULL; }
    for (i32 k = 0; k < world->count; k++) {
        caster_object* o = &fs->objects[fs->n_objects];
        memset(o, 0, sizeof *o);
        if (world->types[k] == HTYPE_SPHERE) {
            const sphere* s = (const sphere*)world->head[k];
            o->type = CASTER_HTYPE_SPHERE;
            o->g[0] = s->center.x; o->g[1] = s->center.y; o->g[2] = s->center.z; o->g[3] = s->radius;
            o->material = find_or_add_material(fs, s->material);
        } else if (world->types[k] == HTYPE_RECT_XY) {
            const xy_rect* r = (const xy_rect*)world->head[k];
            o->type = CASTER_HTYPE_RECT_XY;
            o->g[0] = r->center.x; o->g[1] = r->center.y; o->g[2] = r->center.z;
            o->g[3] = r->dims.u; o->g[4] = r->dims.v;
            o->material = find_or_add_material(fs, r->material);
        } else {
            crt_flat_free(fs);   /* xz/yz rects and BVH nodes have no definition in the reference */
            return NULL;
        }
        fs->n_objects++;
    }
    if (fs->noise_conflict) {       /* a noise texture without a table, or two different tables */
        crt_flat_free(fs);
        return NULL;
    }
    return fs;
}

void crt_flat_free(crt_flat_scene* fs)
{
    if (!fs) return;
    free(fs->objects); free(fs->materials); free(fs->images);
    free((void*)fs->mat_keys); free((void*)fs->img_keys);
    free(fs);
}

int crt_flat_counts(const crt_flat_scene* fs, int* n_objects, int* n_materials, int* n_images)
{
    if (!fs) return 1;
    if (n_objects) *n_objects = fs->n_objects;
    if (n_materials) *n_materials = fs->n_materials;
    if (n_images) *n_images = fs->n_images;
    return 0;
}
const void* crt_flat_objects(const crt_flat_scene* fs) { return fs ? fs->objects : NULL; }
const void* crt_flat_materials(const crt_flat_scene* fs) { return fs ? fs->materials : NULL; }
const void* crt_flat_images(const crt_flat_scene* fs) { return fs ? fs->images : NULL; }
const void* crt_flat_noise(const crt_flat_scene* fs) { return fs ? fs->noise : NULL; }
