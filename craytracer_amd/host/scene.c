/* Scene setup of the host: the hit_list helpers (reference: src/caster.c:169-245),
 * the reference's own scene (src/main.c:14-55) and the two benchmark scenes,
 * all built from the reference's primitives (spheres, xy rects, the four
 * material types). */
#include "craytracer.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>

hit_list new_hit_list(void)                                  /* src/caster.c:169-176 */
{
    hit_list l;
    l.head = NULL; l.types = NULL; l.count = 0;
    return l;
}

void hit_list_add_tail(hit_list* list, void* hittable, i8 hit_type)   /* src/caster.c:178-196 */
{
    list->head = (void**)realloc(list->head, sizeof(void*) * (size_t)(list->count + 1));
    list->types = (i8*)realloc(list->types, sizeof(i8) * (size_t)(list->count + 1));
    list->head[list->count] = hittable;
    list->types[list->count] = hit_type;
    list->count++;
}

void free_hit_list(hit_list* list)                           /* src/caster.c:238-245 */
{
    free(list->head);
    free(list->types);
    list->head = NULL; list->types = NULL; list->count = 0;
}

/* Scene-building random numbers: a SplitMix64-seeded PCG-XSH-RR stream, 24-bit
 * draws scaled to [0,1).  Separate from the render stream. */
typedef struct { u64 s; } scene_rng;
static void srng_init(scene_rng* g, u32 seed)
{
    u64 z = 0xFFFFFFFF00000000ull ^ ((u64)seed * 0x9E3779B97F4A7C15ull);
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    g->s = z ^ (z >> 31);
}
static f32 srnd(scene_rng* g)
{
    const u64 old = g->s;
    g->s = old * 6364136223846793005ull + 1442695040888963407ull;
    const u32 xs = (u32)(((old >> 18) ^ old) >> 27);
    const u32 rot = (u32)(old >> 59);
    const u32 out = (xs >> rot) | (xs << ((32u - rot) & 31u));
    return (f32)(out >> 8) / 16777216.0f;
}
static f32 srnd_range(scene_rng* g, f32 lo, f32 hi) { return lo + (hi - lo) * srnd(g); }

static crt_scene* scene_alloc(i32 max_spheres, i32 max_rects, i32 max_materials)
{
    crt_scene* sc = (crt_scene*)calloc(1, sizeof(crt_scene));
    if (!sc) return NULL;
    sc->world = new_hit_list();
    sc->spheres = (sphere*)calloc((size_t)(max_spheres > 0 ? max_spheres : 1), sizeof(sphere));
    sc->rects = (xy_rect*)calloc((size_t)(max_rects > 0 ? max_rects : 1), sizeof(xy_rect));
    sc->materials = (material*)calloc((size_t)(max_materials > 0 ? max_materials : 1), sizeof(material));
    if (!sc->spheres || !sc->rects || !sc->materials) { crt_scene_free(sc); return NULL; }
    return sc;
}

void crt_scene_free(crt_scene* sc)
{
    if (!sc) return;
    free_hit_list(&sc->world);
    free(sc->spheres);
    free(sc->rects);
    free(sc->materials);
    free_image_buffer(sc->texture_image);
    free_noise_perlin(sc->noise);
    free(sc);
}

static material* add_material(crt_scene* sc, i8 type, u8 tex_type, f32 r, f32 g, f32 b, f32 fuzz, f32 ior)
{
    material* m = &sc->materials[sc->n_materials++];
    memset(m, 0, sizeof *m);
    m->type = type;
    m->texture.type = tex_type;
    m->texture.color[0].x = r; m->texture.color[0].y = g; m->texture.color[0].z = b;
    m->fuzziness = fuzz;
    m->n_refract = ior;
    return m;
}

static void add_sphere(crt_scene* sc, f32 x, f32 y, f32 z, f32 radius, material* m)
{
    sphere* s = &sc->spheres[sc->n_spheres++];
    s->center.x = x; s->center.y = y; s->center.z = z;
    s->radius = radius;
    s->material = m;
    hit_list_add_tail(&sc->world, s, HTYPE_SPHERE);
}

static void add_xy_rect(crt_scene* sc, f32 x, f32 y, f32 z, f32 du, f32 dv, material* m)
{
    xy_rect* r = &sc->rects[sc->n_rects++];
    r->center.x = x; r->center.y = y; r->center.z = z;
    r->dims.u = du; r->dims.v = dv;
    r->material = m;
    hit_list_add_tail(&sc->world, r, HTYPE_RECT_XY);
}

/* src/main.c:14-55.  List order: ground, s[1] (glass), s[2] (earth), r[0]. */
crt_scene* crt_scene_0(img_buffer* earth)
{
    crt_scene* sc = scene_alloc(3, 1, 4);
    if (!sc) return NULL;
    material* ground = add_material(sc, MTYPE_LAMBERTIAN, TTYPE_CHECKER, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f);
    ground->texture.color[1].x = 0.0f; ground->texture.color[1].y = 0.0f; ground->texture.color[1].z = 0.0f;
    material* glass = add_material(sc, MTYPE_DIELECTRIC, TTYPE_COLOR, 0.0f, 0.0f, 0.0f, 0.0f, 1.5f);
    material* globe;
    if (earth) {
        globe = add_material(sc, MTYPE_LAMBERTIAN, TTYPE_IMAGE_ALBEDO, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f);
        globe->texture.image = earth;
        sc->texture_image = earth;
    } else {
        globe = add_material(sc, MTYPE_LAMBERTIAN, TTYPE_COLOR, 0.4f, 0.2f, 0.1f, 0.0f, 0.0f);
    }
    material* light = add_material(sc, MTYPE_DIFF_LIGHT, TTYPE_COLOR, 2.0f, 2.0f, 2.0f, 0.0f, 0.0f);
    add_sphere(sc, 0.0f, -1000.0f, 0.0f, 1000.0f, ground);
    add_sphere(sc, 4.0f, 1.0f, 0.0f, 1.0f, glass);
    add_sphere(sc, 4.0f, 1.0f, 3.0f, 1.0f, globe);
    add_xy_rect(sc, 4.0f, 2.0f, -2.0f, 2.0f, 2.0f, light);
    return sc;
}

/* "Ray Tracing in One Weekend" final scene: 22 x 22 grid of small spheres
 * (80 % diffuse, 15 % metal, 5 % glass), the three unit spheres whose
 * materials the reference declares as mat0..mat2 (src/main.c:24-26), and an
 * enclosing emissive sphere as the only light. */
crt_scene* crt_scene_rtiow(u32 scene_seed)
{
    crt_scene* sc = scene_alloc(22 * 22 + 5, 0, 22 * 22 + 5);
    if (!sc) return NULL;
    scene_rng g;
    srng_init(&g, scene_seed);
    add_sphere(sc, 0.0f, -1000.0f, 0.0f, 1000.0f,
               add_material(sc, MTYPE_LAMBERTIAN, TTYPE_COLOR, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            const f32 choose = srnd(&g);
            const f32 cx = (f32)a + 0.9f * srnd(&g);
            const f32 cz = (f32)b + 0.9f * srnd(&g);
            const f32 dx = cx - 4.0f, dz = cz;
            if (dx * dx + dz * dz <= 0.9f * 0.9f) continue;
            material* m;
            if (choose < 0.8f) {
                const f32 r = srnd(&g) * srnd(&g), gg = srnd(&g) * srnd(&g), bb = srnd(&g) * srnd(&g);
                m = add_material(sc, MTYPE_LAMBERTIAN, TTYPE_COLOR, r, gg, bb, 0.0f, 0.0f);
            } else if (choose < 0.95f) {
                const f32 r = srnd_range(&g, 0.5f, 1.0f), gg = srnd_range(&g, 0.5f, 1.0f), bb = srnd_range(&g, 0.5f, 1.0f);
                const f32 fuzz = srnd_range(&g, 0.0f, 0.5f);
                m = add_material(sc, MTYPE_METAL, TTYPE_COLOR, r, gg, bb, fuzz, 0.0f);
            } else {
                m = add_material(sc, MTYPE_DIELECTRIC, TTYPE_COLOR, 0.0f, 0.0f, 0.0f, 0.0f, 1.5f);
            }
            add_sphere(sc, cx, 0.2f, cz, 0.2f, m);
        }
    }
    add_sphere(sc, 0.0f, 1.0f, 0.0f, 1.0f, add_material(sc, MTYPE_DIELECTRIC, TTYPE_COLOR, 0, 0, 0, 0.0f, 1.5f));
    add_sphere(sc, -4.0f, 1.0f, 0.0f, 1.0f, add_material(sc, MTYPE_LAMBERTIAN, TTYPE_COLOR, 0.4f, 0.2f, 0.1f, 0.0f, 0.0f));
    add_sphere(sc, 4.0f, 1.0f, 0.0f, 1.0f, add_material(sc, MTYPE_METAL, TTYPE_COLOR, 0.7f, 0.6f, 0.5f, 0.0f, 0.0f));
    add_sphere(sc, 0.0f, 0.0f, 0.0f, 100000.0f, add_material(sc, MTYPE_DIFF_LIGHT, TTYPE_COLOR, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f));
    return sc;
}

/* Gradient-noise tables: 256 unit vectors with components drawn from [-1, 1) and three
 * Fisher-Yates permutations of 0..255 (the construction of "Ray Tracing: The Next Week"; the
 * reference's inc/noise.h:4-7 is a placeholder). */
noise_perlin* new_noise_perlin(u32 seed)
{
    noise_perlin* n = (noise_perlin*)calloc(1, sizeof *n);
    if (!n) return NULL;
    scene_rng g;
    srng_init(&g, seed ^ 0x50455231u);
    for (int k = 0; k < 256; k++) {
        f32 x, y, z, len2;
        do {
            x = srnd_range(&g, -1.0f, 1.0f); y = srnd_range(&g, -1.0f, 1.0f); z = srnd_range(&g, -1.0f, 1.0f);
            len2 = x * x + y * y + z * z;
        } while (len2 < 1.0e-6f);
        const f32 len = sqrtf(len2);
        n->ranvec[k].x = x / len; n->ranvec[k].y = y / len; n->ranvec[k].z = z / len;
    }
    u8* perms[3] = {n->perm_x, n->perm_y, n->perm_z};
    for (int t = 0; t < 3; t++) {
        u8* p = perms[t];
        for (int k = 0; k < 256; k++) p[k] = (u8)k;
        for (int k = 255; k > 0; k--) {
            int target = (int)(srnd(&g) * (f32)(k + 1));
            if (target > k) target = k;
            const u8 tmp = p[k]; p[k] = p[target]; p[target] = tmp;
        }
    }
    return n;
}

void free_noise_perlin(noise_perlin* noise) { free(noise); }

static material* add_noise_material(crt_scene* sc, i8 type, f32 r, f32 g, f32 b, f32 scale, f32 fuzz)
{
    material* m = add_material(sc, type, TTYPE_NOISE, r, g, b, fuzz, 0.0f);
    m->texture.color[1].x = scale;
    m->texture.noise = sc->noise;
    return m;
}

/* The layout of crt_scene_rtiow with marble textures: ground, diffuse small spheres and the
 * big diffuse sphere are TTYPE_NOISE, the metals keep a solid colour. */
crt_scene* crt_scene_perlin(u32 scene_seed)
{
    crt_scene* sc = scene_alloc(22 * 22 + 5, 0, 22 * 22 + 5);
    if (!sc) return NULL;
    sc->noise = new_noise_perlin(scene_seed);
    if (!sc->noise) { crt_scene_free(sc); return NULL; }
    scene_rng g;
    srng_init(&g, scene_seed);
    add_sphere(sc, 0.0f, -1000.0f, 0.0f, 1000.0f, add_noise_material(sc, MTYPE_LAMBERTIAN, 1.0f, 1.0f, 1.0f, 4.0f, 0.0f));
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            const f32 choose = srnd(&g);
            const f32 cx = (f32)a + 0.9f * srnd(&g);
            const f32 cz = (f32)b + 0.9f * srnd(&g);
            const f32 dx = cx - 4.0f, dz = cz;
            if (dx * dx + dz * dz <= 0.9f * 0.9f) continue;
            material* m;
            if (choose < 0.8f) {
                const f32 r = srnd_range(&g, 0.3f, 1.0f), gg = srnd_range(&g, 0.3f, 1.0f), bb = srnd_range(&g, 0.3f, 1.0f);
                m = add_noise_material(sc, MTYPE_LAMBERTIAN, r, gg, bb, srnd_range(&g, 4.0f, 40.0f), 0.0f);
            } else if (choose < 0.95f) {
                const f32 r = srnd_range(&g, 0.5f, 1.0f), gg = srnd_range(&g, 0.5f, 1.0f), bb = srnd_range(&g, 0.5f, 1.0f);
                const f32 fuzz = srnd_range(&g, 0.0f, 0.5f);
                m = add_material(sc, MTYPE_METAL, TTYPE_COLOR, r, gg, bb, fuzz, 0.0f);
            } else {
                m = add_material(sc, MTYPE_DIELECTRIC, TTYPE_COLOR, 0.0f, 0.0f, 0.0f, 0.0f, 1.5f);
            }
            add_sphere(sc, cx, 0.2f, cz, 0.2f, m);
        }
    }
    add_sphere(sc, 0.0f, 1.0f, 0.0f, 1.0f, add_material(sc, MTYPE_DIELECTRIC, TTYPE_COLOR, 0, 0, 0, 0.0f, 1.5f));
    add_sphere(sc, -4.0f, 1.0f, 0.0f, 1.0f, add_noise_material(sc, MTYPE_LAMBERTIAN, 1.0f, 0.9f, 0.8f, 8.0f, 0.0f));
    add_sphere(sc, 4.0f, 1.0f, 0.0f, 1.0f, add_material(sc, MTYPE_METAL, TTYPE_COLOR, 0.7f, 0.6f, 0.5f, 0.0f, 0.0f));
    add_sphere(sc, 0.0f, 0.0f, 0.0f, 100000.0f, add_material(sc, MTYPE_DIFF_LIGHT, TTYPE_COLOR, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f));
    return sc;
}

/* Brute-force stress scene: ground, n small spheres (70 % diffuse, 20 % metal,
 * 10 % glass) with centres uniform in [-50,50] x [0.2,10] x [-50,50] and radii
 * in [0.1,0.5], then the sky sphere. */
crt_scene* crt_scene_stress(i32 n, u32 scene_seed)
{
    if (n < 0) n = 0;
    crt_scene* sc = scene_alloc(n + 2, 0, n + 2);
    if (!sc) return NULL;
    scene_rng g;
    srng_init(&g, scene_seed);
    add_sphere(sc, 0.0f, -1000.0f, 0.0f, 1000.0f,
               add_material(sc, MTYPE_LAMBERTIAN, TTYPE_COLOR, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f));
    for (i32 k = 0; k < n; k++) {
        const f32 cx = srnd_range(&g, -50.0f, 50.0f);
        const f32 cy = srnd_range(&g, 0.2f, 10.0f);
        const f32 cz = srnd_range(&g, -50.0f, 50.0f);
        const f32 rad = srnd_range(&g, 0.1f, 0.5f);
        const f32 choose = srnd(&g);
        material* m;
        if (choose < 0.7f)
            m = add_material(sc, MTYPE_LAMBERTIAN, TTYPE_COLOR, srnd(&g), srnd(&g), srnd(&g), 0.0f, 0.0f);
        else if (choose < 0.9f)
            m = add_material(sc, MTYPE_METAL, TTYPE_COLOR, srnd_range(&g, 0.5f, 1.0f), srnd_range(&g, 0.5f, 1.0f),
                             srnd_range(&g, 0.5f, 1.0f), srnd_range(&g, 0.0f, 0.5f), 0.0f);
        else
            m = add_material(sc, MTYPE_DIELECTRIC, TTYPE_COLOR, 0, 0, 0, 0.0f, 1.5f);
        add_sphere(sc, cx, cy, cz, rad, m);
    }
    add_sphere(sc, 0.0f, 0.0f, 0.0f, 100000.0f, add_material(sc, MTYPE_DIFF_LIGHT, TTYPE_COLOR, 1.0f, 1.0f, 1.0f, 0.0f, 0.0f));
    return sc;
}
