/* Host-side camera setup (the reference's src/camera.c:3-30): runs once per
 * frame on the CPU; its result, cam_info, is a kernel argument. */
#include "craytracer.h"

#include <math.h>

static v3_f32 vsub(v3_f32 a, v3_f32 b) { v3_f32 r; r.x = a.x - b.x; r.y = a.y - b.y; r.z = a.z - b.z; return r; }
static v3_f32 vscale(v3_f32 a, f32 s) { v3_f32 r; r.x = a.x * s; r.y = a.y * s; r.z = a.z * s; return r; }
static v3_f32 vdiv(v3_f32 a, f32 s) { v3_f32 r; r.x = a.x / s; r.y = a.y / s; r.z = a.z / s; return r; }
static v3_f32 vcross(v3_f32 a, v3_f32 b)
{
    v3_f32 r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
static v3_f32 vunit(v3_f32 a) { return vdiv(a, sqrtf(a.x * a.x + a.y * a.y + a.z * a.z)); }

cam_info calculate_cam_info(v3_f32 cam_pos, v3_f32 look_at, v3_f32 cam_up, f32 cam_fov, f32 cam_ar,
                            f32 aperture, f32 focus)
{
    cam_info ci;
    const f32 theta = cam_fov * 3.141592654f / 180.0f;
    const f32 half_h = tanf(theta / 2.0f);
    const f32 vp_h = 2.0f * half_h;
    const f32 vp_w = cam_ar * vp_h;

    const v3_f32 w = vunit(vsub(cam_pos, look_at));
    const v3_f32 u = vunit(vcross(cam_up, w));
    const v3_f32 v = vcross(w, u);

    ci.o = cam_pos;
    ci.u0 = u;
    ci.v0 = v;
    ci.h = vscale(u, focus * vp_w);
    ci.v = vscale(v, focus * vp_h);
    ci.llc = vsub(vsub(vsub(ci.o, vdiv(ci.h, 2)), vdiv(ci.v, 2)), vscale(w, focus));
    ci.lr = aperture / 2.0f;
    return ci;
}

/* src/main.c:71-78 */
cam_info crt_default_camera(f32 aspect)
{
    v3_f32 pos = {{{13, 2, 3}}}, at = {{{0, 0, 0}}}, up = {{{0, 1, 0}}};
    return calculate_cam_info(pos, at, up, 20.0f, aspect, 0.1f, 10.0f);
}
