// C ABI of libcaster_hip (see include/caster_hip.h).  Host side only: argument
// checks, scene flattening into device arrays, the fixed-point frame buffer, kernel launches.
#include "caster_hip.h"
#include "caster_launch.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is opened at run time (see Rccl)

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(CASTER_HIP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int kCounterWords = 4;

}  // namespace

// Scratch of caster_hip_render_noalloc: the counters and the per-pixel fixed-point sums of a call,
// which caster_hip_render keeps in buffers it allocates.  A file-scope device array: every GPU the
// library runs on holds its own copy (24 MiB + 32 B) at its own address until the process ends.  In
// host code the name is only a handle for hipGetSymbolAddress (caster_hip_create asks once per
// device and keeps the pointer); kernels and hipMemsetAsync are given that pointer, never the name.
constexpr size_t kNoAllocMaxPixels = (size_t)CASTER_HIP_NOALLOC_MAX_PIXELS;
constexpr size_t kNoAllocMaxWaves = (size_t)CASTER_HIP_NOALLOC_MAX_WAVES;
constexpr size_t kRingWords64 = caster::kRingBytesPerWave / sizeof(unsigned long long);     // per wave
static __device__ __align__(256) unsigned long long scratch[kCounterWords + 3 * kNoAllocMaxPixels + kNoAllocMaxWaves * kRingWords64];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned long long* g_scratch_ptr[kMaxDevices] = {};    // per device, filled in with that device current
}  // namespace

struct caster_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // scene
    bool has_scene = false;
    caster::SceneDev scene{};
    float* staging = nullptr;   // peer-copy target of caster_hip_add_from_device
    size_t staging_bytes = 0;
    std::vector<void*> scene_allocs;
    bool needs_noise = false;   // the scene has a CASTER_TTYPE_NOISE material
    // scratch
    unsigned long long* fixed = nullptr;      // per-pixel fixed-point sums of the call in flight
    size_t fixed_bytes = 0;
    unsigned long long* counters = nullptr;   // kCounterWords words: [0] rays, [1] low word: work-unit queue head, [2] wave trips, [3] camera-ray batches
    uint32_t* ring = nullptr;                 // the waves' rings of waiting paths: kRingBytesPerWave for each of resident_blocks * 4 waves
    unsigned long long* scratch = nullptr;    // this device's copy of the static scratch array (caster_hip_render_noalloc)
    int resident_blocks = 1024;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool render_pending = false;
    int chunk = 0;  // samples per work unit; 0 = automatic
    // what the last render launched
    const unsigned long long* last_counters = nullptr;
    int last_kind = -1;
    int last_blocks = 0;
    int last_spheres = 0;
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

void free_scene(caster_hip_ctx* ctx)
{
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->scene = caster::SceneDev{};
    ctx->has_scene = false;
    ctx->needs_noise = false;
}

template <typename T>
int upload_array(caster_hip_ctx* ctx, const std::vector<T>& host, const T** dev_out)
{
    *dev_out = nullptr;
    if (host.empty()) return CASTER_HIP_OK;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, host.size() * sizeof(T)));
    ctx->scene_allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev_out = static_cast<const T*>(d);
    return CASTER_HIP_OK;
}

// RAII device buffer for the array-at-a-time test entry points.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    hipError_t put(const void* src, size_t bytes)
    {
        hipError_t e = alloc(bytes);
        if (e != hipSuccess) return e;
        return bytes ? hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    }
    hipError_t get(void* dst, size_t bytes) const
    {
        return bytes ? hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost) : hipSuccess;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

caster::CamRec cam_rec(const caster_camera* c)
{
    caster::CamRec r;
    static_assert(sizeof(caster::CamRec) == sizeof(caster_camera), "camera layout");
    memcpy(&r, c, sizeof r);
    return r;
}

}  // namespace

extern "C" {

const char* caster_hip_last_error(void) { return g_err; }

int caster_hip_device_count(int* count)
{
    if (!count) return fail(CASTER_HIP_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(CASTER_HIP_ERR_DEVICE, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
    }
    *count = n;
    return CASTER_HIP_OK;
}

int caster_hip_create(int device, caster_hip_ctx** out)
{
    if (!out) return fail(CASTER_HIP_ERR_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(CASTER_HIP_ERR_DEVICE, "no HIP device available (%s); libcaster_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= n) return fail(CASTER_HIP_ERR_ARG, "device %d out of range [0, %d)", device, n);
    DeviceGuard guard(device);
    if (!guard.ok) return fail(CASTER_HIP_ERR_DEVICE, "hipSetDevice(%d) failed", device);
    caster_hip_ctx* ctx = new (std::nothrow) caster_hip_ctx();
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "out of host memory");
    ctx->device = device;
    {
        // persistent grid: 4 blocks of 4 waves per CU (the kernel is built for 4 waves per SIMD)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            ctx->resident_blocks = prop.multiProcessorCount * 4;     // CASTER_BLOCKS_PER_CU of caster_render.hpp
    }
    const char* env_chunk = getenv("CASTER_HIP_CHUNK");
    if (env_chunk) ctx->chunk = atoi(env_chunk);
    if (hipMalloc(reinterpret_cast<void**>(&ctx->ring), (size_t)ctx->resident_blocks * caster::kWavesPerBlock * caster::kRingBytesPerWave) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&ctx->counters), kCounterWords * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(ctx->counters, 0, kCounterWords * sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        caster_hip_destroy(ctx);
        return fail(CASTER_HIP_ERR_DEVICE, "context allocation on device %d failed", device);
    }
    if (device < kMaxDevices) {
        // the address of this device's copy of the scratch array (the device is current: DeviceGuard above)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void* p = nullptr;
            const hipError_t se = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (se != hipSuccess || !p) {
                caster_hip_destroy(ctx);
                return fail(CASTER_HIP_ERR_DEVICE, "hipGetSymbolAddress(scratch) on device %d failed: %s", device,
                            hipGetErrorString(se));
            }
            g_scratch_ptr[device] = static_cast<unsigned long long*>(p);
        }
        ctx->scratch = g_scratch_ptr[device];
    }
    *out = ctx;
    return CASTER_HIP_OK;
}

int caster_hip_destroy(caster_hip_ctx* ctx)
{
    if (!ctx) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    if (ctx->fixed) (void)hipFree(ctx->fixed);
    if (ctx->staging) (void)hipFree(ctx->staging);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->ring) (void)hipFree(ctx->ring);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    delete ctx;
    return CASTER_HIP_OK;
}

int caster_hip_set_stream(caster_hip_ctx* ctx, void* hip_stream)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return CASTER_HIP_OK;
}

int caster_hip_set_chunk(caster_hip_ctx* ctx, int samples_per_unit)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    if (samples_per_unit < 0) return fail(CASTER_HIP_ERR_ARG, "samples_per_unit must be >= 0");
    ctx->chunk = samples_per_unit;
    return CASTER_HIP_OK;
}

// Operand tables of the matrix-core filter (layout and arithmetic: caster_render.hpp).  A scene
// qualifies when every sphere that is not tested directly has finite filter entries and the list
// entries can name its blocks; otherwise mfma_blocks stays 0 and the vector loop runs.
static void split_half_host(double x, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)(float)x;
    lo = (_Float16)(float)(x - (double)(float)hi);
}

static int build_matrix_tables(caster_hip_ctx* ctx, const std::vector<float4>& sph, const std::vector<float4>& sph_filter)
{
    caster::SceneDev& s = ctx->scene;
    s.mfma_tab[0] = s.mfma_tab[1] = nullptr;
    s.mfma_blocks = 0;
    s.mfma_scale = 1.0f;
    const size_t n = sph.size();
    const size_t n_blocks = (n + 31) / 32;
    if (n == 0) return CASTER_HIP_OK;
    double extent = 0.0;
    for (size_t k = 0; k < n; ++k) {
        const float4 f = sph_filter[k];            // c'x, c'y, c'z, C - r*r (+inf: tested directly, -inf: huge but not listed)
        if (f.w == INFINITY) continue;
        if (f.w == -INFINITY || !std::isfinite(f.x) || !std::isfinite(f.y) || !std::isfinite(f.z)) return CASTER_HIP_OK;
        extent = std::max(extent, std::max(std::max(fabs((double)f.x), fabs((double)f.y)), fabs((double)f.z)));
        extent = std::max(extent, fabs((double)sph[k].w));
    }
    // scaled extents in (32, 64]: squares and K stay below 2^14, far inside half-precision range
    int e = 0;
    if (extent > 0.0 && std::isfinite(extent)) {
        (void)frexp(extent, &e);                   // extent = m * 2^e, m in [0.5, 1)
        e = 6 - e;
        if (e > 100) e = 100;
        if (e < -100) e = -100;
    }
    const double scale = ldexp(1.0, e);
    s.mfma_scale = (float)scale;
    for (int mode = 0; mode < 2; ++mode) {
        const double g = (double)(mode ? caster::kFilterMarginStrict : caster::kFilterMarginFast) +
                         (double)caster::kFilterMarginMfmaExtra;
        std::vector<_Float16> tab((n_blocks + 1) * 2 * 64 * 8, (_Float16)0.0f);
        for (size_t k = 0; k < (n_blocks + 1) * 32; ++k) {
            _Float16 slot[32];
            for (int i = 0; i < 32; ++i) slot[i] = (_Float16)0.0f;
            const bool live = k < n && sph_filter[k].w != INFINITY;
            if (live) {
                const double cx = (double)sph_filter[k].x * scale, cy = (double)sph_filter[k].y * scale,
                             cz = (double)sph_filter[k].z * scale;
                const double r = (double)sph[k].w * scale;
                const double C = cx * cx + cy * cy + cz * cz, r2 = r * r;
                const double feat[9] = {cx * cx, cy * cy, cz * cz, cx * cy, cx * cz, cy * cz, cx, cy, cz};
                for (int f = 0; f < 9; ++f) {
                    _Float16 hi, lo;
                    split_half_host(feat[f], hi, lo);
                    slot[3 * f] = hi; slot[3 * f + 1] = hi; slot[3 * f + 2] = lo;
                }
                // the constant of the key, with the sphere's share of the margin and an absolute
                // term for what half-precision subnormals lose
                const double Z = g * (C + r2) - (C - r2) + 1.0e-4;
                split_half_host(Z, slot[27], slot[28]);
            } else {
                slot[27] = (_Float16)(-60000.0f);   // never a candidate: pad rows and the spheres that are tested directly
            }
            slot[29] = slot[30] = slot[31] = (_Float16)1024.0f;   // against the three halves of -threshold / 1024
            // The table holds the NEGATED operand: the product is then threshold - key, and a set
            // sign bit marks a candidate (the kernel gathers sign bits; no complement needed).
            const size_t blk = k / 32, r = k % 32;
            for (int q = 0; q < 2; ++q)
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 8; ++j)
                        tab[(((blk * 2 + q) * 64) + (size_t)h * 32 + r) * 8 + j] = -slot[16 * q + 8 * h + j];
        }
        static_assert(sizeof(_Float16) == 2, "half size");
        std::vector<uint4> img(tab.size() / 8);
        memcpy(img.data(), tab.data(), img.size() * sizeof(uint4));
        int rc;
        if ((rc = upload_array(ctx, img, &s.mfma_tab[mode]))) return rc;
    }
    s.mfma_blocks = (int)n_blocks;
    return CASTER_HIP_OK;
}

int caster_hip_upload_scene(caster_hip_ctx* ctx, int n_objects, const caster_object* objects,
                            int n_materials, const caster_material* materials,
                            int n_images, const caster_image* images)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    if (n_objects < 0 || n_materials < 0 || n_images < 0) return fail(CASTER_HIP_ERR_ARG, "negative count");
    if ((n_objects && !objects) || (n_materials && !materials) || (n_images && !images))
        return fail(CASTER_HIP_ERR_ARG, "NULL table with non-zero count");

    std::vector<float4> sph, rect_a, rect_b;
    std::vector<int2> sph_meta, rect_meta;
    for (int k = 0; k < n_objects; ++k) {
        const caster_object& o = objects[k];
        if (o.material < 0 || o.material >= n_materials)
            return fail(CASTER_HIP_ERR_ARG, "object %d: material %d out of range [0, %d)", k, o.material, n_materials);
        if (o.type == CASTER_HTYPE_SPHERE) {
            sph.push_back(make_float4(o.g[0], o.g[1], o.g[2], o.g[3]));
            sph_meta.push_back(make_int2(k, o.material));
        } else if (o.type == CASTER_HTYPE_RECT_XY) {
            rect_a.push_back(make_float4(o.g[0], o.g[1], o.g[2], o.g[3]));
            rect_b.push_back(make_float4(o.g[4], 0.f, 0.f, 0.f));
            rect_meta.push_back(make_int2(k, o.material));
        } else {
            return fail(CASTER_HIP_ERR_ARG, "object %d: unsupported type tag %d", k, o.type);
        }
    }
    for (int k = 0; k < n_materials; ++k) {
        const caster_material& m = materials[k];
        if (m.type < 0 || m.type > CASTER_MTYPE_DIFF_LIGHT)
            return fail(CASTER_HIP_ERR_ARG, "material %d: unknown type tag %d", k, m.type);
        if (m.tex_type < 0 || m.tex_type > CASTER_TTYPE_NOISE)
            return fail(CASTER_HIP_ERR_ARG, "material %d: unknown texture tag %d", k, m.tex_type);
        if (m.tex_type == CASTER_TTYPE_IMAGE_ALBEDO && (m.image < 0 || m.image >= n_images))
            return fail(CASTER_HIP_ERR_ARG, "material %d: image %d out of range [0, %d)", k, m.image, n_images);
    }
    for (int k = 0; k < n_images; ++k) {
        if (!images[k].data || images[k].w == 0 || images[k].h == 0 || images[k].bpp < 3)
            return fail(CASTER_HIP_ERR_ARG, "image %d: needs data, non-zero size and bpp >= 3", k);
    }

    DeviceGuard guard(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    free_scene(ctx);

    std::vector<caster::MatRec> mats(n_materials);
    static_assert(sizeof(caster::MatRec) == sizeof(caster_material), "material layout");
    if (n_materials) memcpy(mats.data(), materials, sizeof(caster_material) * (size_t)n_materials);

    std::vector<caster::ImgRec> imgs(n_images);
    for (int k = 0; k < n_images; ++k) {
        const size_t bytes = (size_t)images[k].w * images[k].h * images[k].bpp;
        void* d = nullptr;
        HIP_TRY(hipMalloc(&d, bytes));
        ctx->scene_allocs.push_back(d);
        HIP_TRY(hipMemcpy(d, images[k].data, bytes, hipMemcpyHostToDevice));
        imgs[k].data = static_cast<const uint8_t*>(d);
        imgs[k].w = images[k].w; imgs[k].h = images[k].h; imgs[k].bpp = images[k].bpp; imgs[k].pad = 0;
    }

    // Candidate filter tables (see caster_render.hpp): reference point P = mean
    // centre of the ordinary spheres; per sphere c' = c - P, K = |c'|^2 - r*r and
    // the error scale |c'|^2 + r*r.  A sphere much larger than the median one
    // is a candidate for every ray: the first few of them are listed in `always`
    // and tested directly (K = +inf keeps them out of the filter), the others
    // get K = -inf.
    std::vector<float4> sph_filter(sph.size());
    std::vector<float> sph_scale(sph.size());
    ctx->scene.n_always = 0;
    double ref[3] = {0.0, 0.0, 0.0};
    if (!sph.empty()) {
        std::vector<float> radii;
        for (const float4& q : sph) radii.push_back(fabsf(q.w));
        std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
        const double huge_r = 32.0 * (double)radii[radii.size() / 2];
        size_t n_ord = 0;
        for (const float4& q : sph)
            if (fabs((double)q.w) <= huge_r) { ref[0] += q.x; ref[1] += q.y; ref[2] += q.z; ++n_ord; }
        if (n_ord) { ref[0] /= (double)n_ord; ref[1] /= (double)n_ord; ref[2] /= (double)n_ord; }
        const float pf[3] = {(float)ref[0], (float)ref[1], (float)ref[2]};
        ctx->scene.n_always = 0;
        for (size_t k = 0; k < sph.size(); ++k) {
            const float4 q = sph[k];
            const float cx = q.x - pf[0], cy = q.y - pf[1], cz = q.z - pf[2];
            const double C = (double)cx * cx + (double)cy * cy + (double)cz * cz;
            const double r2 = (double)q.w * (double)q.w;
            const bool huge = !(fabs((double)q.w) <= huge_r) || !std::isfinite(C) || !std::isfinite(r2);
            const bool direct = huge && ctx->scene.n_always < caster::kMaxAlways;
            if (direct) ctx->scene.always[ctx->scene.n_always++] = (int)k;
            sph_filter[k] = make_float4(cx, cy, cz, direct ? INFINITY : huge ? -INFINITY : (float)(C - r2));
            sph_scale[k] = huge ? 0.0f : (float)(C + r2);
        }
        ref[0] = pf[0]; ref[1] = pf[1]; ref[2] = pf[2];
    }

    caster::SceneDev& s = ctx->scene;
    int rc;
    if ((rc = upload_array(ctx, sph, &s.sph))) return rc;
    for (int mode = 0; mode < 2; ++mode) {
        const float margin = mode ? caster::kFilterMarginStrict : caster::kFilterMarginFast;
        std::vector<float4> img(sph.size() + 16, make_float4(0.0f, 0.0f, -1.0e30f, 0.0f));
        for (size_t k = 0; k < sph.size(); ++k) {
            const float4 h = sph_filter[k];
            img[k] = make_float4(h.x, h.y, (h.w == INFINITY) ? -1.0e30f : margin * sph_scale[k] - h.w, h.z);
        }
        if ((rc = upload_array(ctx, img, &s.filter_img[mode]))) return rc;
    }
    s.ref[0] = (float)ref[0]; s.ref[1] = (float)ref[1]; s.ref[2] = (float)ref[2];
    if ((rc = build_matrix_tables(ctx, sph, sph_filter))) return rc;
    if ((rc = upload_array(ctx, sph_meta, &s.sph_meta))) return rc;
    if ((rc = upload_array(ctx, rect_a, &s.rect_a))) return rc;
    if ((rc = upload_array(ctx, rect_b, &s.rect_b))) return rc;
    if ((rc = upload_array(ctx, rect_meta, &s.rect_meta))) return rc;
    if ((rc = upload_array(ctx, mats, &s.mats))) return rc;
    if ((rc = upload_array(ctx, imgs, &s.imgs))) return rc;
    s.n_sph = (int)sph.size();
    s.n_rect = (int)rect_a.size();
    s.n_mat = n_materials;
    s.n_img = n_images;
    ctx->has_scene = true;
    s.all_solid = 1;
    for (int k = 0; k < n_materials; ++k) {
        if (materials[k].tex_type == CASTER_TTYPE_NOISE) ctx->needs_noise = true;
        if (materials[k].tex_type != CASTER_TTYPE_COLOR) s.all_solid = 0;
    }
    return CASTER_HIP_OK;
}

int caster_hip_set_noise(caster_hip_ctx* ctx, const caster_noise* noise)
{
    if (!ctx || !noise) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "caster_hip_set_noise before caster_hip_upload_scene");
    for (int k = 0; k < 256; ++k)
        for (int c = 0; c < 3; ++c)
            if (!std::isfinite(noise->ranvec[k][c])) return fail(CASTER_HIP_ERR_ARG, "noise vector %d is not finite", k);
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<float4> vec(256);
    for (int k = 0; k < 256; ++k) vec[k] = make_float4(noise->ranvec[k][0], noise->ranvec[k][1], noise->ranvec[k][2], 0.0f);
    std::vector<uint8_t> perm(768);
    memcpy(perm.data(), noise->perm_x, 256);
    memcpy(perm.data() + 256, noise->perm_y, 256);
    memcpy(perm.data() + 512, noise->perm_z, 256);
    int rc;
    if ((rc = upload_array(ctx, vec, &ctx->scene.noise_vec))) return rc;
    if ((rc = upload_array(ctx, perm, &ctx->scene.noise_perm))) return rc;
    return CASTER_HIP_OK;
}

int caster_hip_malloc(caster_hip_ctx* ctx, size_t bytes, void** dev_ptr)
{
    if (!ctx || !dev_ptr) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return CASTER_HIP_OK;
}

int caster_hip_free(caster_hip_ctx* ctx, void* dev_ptr)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipFree(dev_ptr));
    return CASTER_HIP_OK;
}

int caster_hip_memset_zero(caster_hip_ctx* ctx, void* dev_ptr, size_t bytes)
{
    if (!ctx || !dev_ptr) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipMemsetAsync(dev_ptr, 0, bytes, ctx->stream));
    return CASTER_HIP_OK;
}

int caster_hip_download(caster_hip_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes)
{
    if (!ctx || !host_dst || !dev_src) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CASTER_HIP_OK;
}

int caster_hip_upload(caster_hip_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes)
{
    if (!ctx || !dev_dst || !host_src) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CASTER_HIP_OK;
}

int caster_hip_synchronize(caster_hip_ctx* ctx)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CASTER_HIP_OK;
}

int caster_hip_mem_info(caster_hip_ctx* ctx, size_t* free_bytes, size_t* total_bytes)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    DeviceGuard guard(ctx->device);
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return CASTER_HIP_OK;
}

// Splits the sample range of a call into the chunks of the work units (unit = 8x8 pixel tile x chunk;
// the queue hands the units out chunk by chunk).  A unit's sums are collected in LDS and reach the
// frame buffer with one atomic per pixel and channel; paths that outlive their unit pay three
// atomics each, so a chunk should hold many more paths than a wave has slots.  Against that, the
// costliest tiles of a frame take many times the average (20 rays per path against 3 where two
// spheres touch in the 1080p benchmark scene), and a unit that starts late and runs long leaves the
// other waves idle at the end.  So every chunk is at most the share of the remaining samples that a
// wave would still finish if its unit were 16 times the average: with many tiles per wave that is one
// half, the chunks then halve towards the end.  The sums are integers, so the split does not change
// the result.
static int plan_chunks(const caster_hip_ctx* ctx, int width, int height, int sample_count, int* start)
{
    const int cap = caster::kMaxChunks;
    int n = 0;
    start[0] = 0;
    if (ctx->chunk > 0) {           // caller's choice: equal chunks
        int chunk = ctx->chunk;
        if ((sample_count + chunk - 1) / chunk > cap) chunk = (sample_count + cap - 1) / cap;
        for (int s = 0; s < sample_count; s += chunk) start[++n] = (s + chunk < sample_count) ? s + chunk : sample_count;
        return n;
    }
    const double tiles = (double)((width + 7) / 8) * (double)((height + 7) / 8);
    const double waves = (double)ctx->resident_blocks * 4.0;
    double share = tiles / (16.0 * waves);
    if (share > 0.5) share = 0.5;
    int min_chunk = 4;
    if (sample_count / min_chunk > cap - 32) min_chunk = (sample_count + cap - 33) / (cap - 32);
    int left = sample_count;
    while (left > 0) {
        int c = (int)((double)left * share);
        if (c > (min_chunk > 64 ? min_chunk : 64)) c = min_chunk > 64 ? min_chunk : 64;
        if (c < min_chunk) c = min_chunk;
        if (c > left || n == cap - 1) c = left;
        start[n + 1] = start[n] + c;
        ++n;
        left -= c;
    }
    return n;
}

// The body of both render entry points.  `fixed` (three planes of width * height words) and `counters`
// (kCounterWords words) are the call's scratch: the context's own buffers, or this device's copy of
// the static array.  Nothing here allocates.
static int render_on(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height, int sample_begin,
                     int sample_count, int max_depth, uint32_t seed, uint32_t flags, float* accum_dev,
                     unsigned long long* fixed, unsigned long long* counters, uint32_t* ring, int ring_waves)
{
    HIP_TRY(hipMemsetAsync(counters, 0, kCounterWords * sizeof(unsigned long long), ctx->stream));
    ctx->render_pending = true;
    ctx->last_counters = counters;
    ctx->last_kind = -1;
    ctx->last_blocks = 0;
    ctx->last_spheres = ctx->scene.n_sph;
    if (sample_count == 0) {
        HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
        HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
        return CASTER_HIP_OK;
    }
    const size_t n_pixels = (size_t)width * height;
    HIP_TRY(hipMemsetAsync(fixed, 0, n_pixels * 3 * sizeof(unsigned long long), ctx->stream));

    caster::RenderArgs a;
    a.cam = cam_rec(cam);
    a.scene = ctx->scene;
    if (flags & CASTER_HIP_VECTOR_LOOP) a.scene.mfma_blocks = 0;
    a.width = width; a.height = height;
    a.s_begin = sample_begin; a.s_count = sample_count;
    a.n_chunks = plan_chunks(ctx, width, height, sample_count, a.chunk_start);
    a.max_depth = max_depth;
    a.seed = seed;
    a.fixed = fixed;
    a.rays = counters;
    a.queue = reinterpret_cast<unsigned int*>(counters + 1);
    a.trips = counters + 2;
    a.batches = counters + 3;
    a.ring = ring;
    a.ring_waves = ring_waves;
    a.tile_lists = (flags & CASTER_HIP_NO_TILE_LISTS) ? 0 : 1;
    a.n_units = 0;
    a.lds_spheres = 0;
    const int strict = (flags & CASTER_HIP_STRICT) ? 1 : 0;
    ctx->last_kind = caster::pick_loop_kind(a.scene, strict);
    ctx->last_blocks = ctx->last_kind >= caster::kLoopMatrix ? a.scene.mfma_blocks : 0;
    // the events bracket the path-tracing kernel alone
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    if (strict) HIP_TRY(caster::launch_render_strict(a, ctx->resident_blocks, ctx->stream));
    else HIP_TRY(caster::launch_render_fast(a, ctx->resident_blocks, ctx->stream));
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(caster::launch_fold(fixed, (int)n_pixels, sample_count, reinterpret_cast<float4*>(accum_dev), ctx->stream));
    return CASTER_HIP_OK;
}

static int check_render_args(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height, int sample_begin,
                             int sample_count, uint32_t flags, float* accum_dev)
{
    if (!ctx || !cam || !accum_dev) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "render before caster_hip_upload_scene");
    if (ctx->needs_noise && !ctx->scene.noise_vec)
        return fail(CASTER_HIP_ERR_STATE, "the scene has a noise texture: call caster_hip_set_noise first");
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535)
        return fail(CASTER_HIP_ERR_ARG, "frame size %dx%d out of range", width, height);
    if (sample_begin < 0 || sample_count < 0) return fail(CASTER_HIP_ERR_ARG, "negative sample range");
    if ((flags & ~(uint32_t)(CASTER_HIP_STRICT | CASTER_HIP_VECTOR_LOOP | CASTER_HIP_NO_TILE_LISTS)) != 0)
        return fail(CASTER_HIP_ERR_ARG, "unknown flag bits 0x%x", flags);
    return CASTER_HIP_OK;
}

int caster_hip_render(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height,
                      int sample_begin, int sample_count, int max_depth,
                      uint32_t seed, uint32_t flags, float* accum_dev)
{
    int rc = check_render_args(ctx, cam, width, height, sample_begin, sample_count, flags, accum_dev);
    if (rc) return rc;
    DeviceGuard guard(ctx->device);
    const size_t need = (size_t)width * height * 3 * sizeof(unsigned long long);
    if (sample_count > 0 && need > ctx->fixed_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->fixed) HIP_TRY(hipFree(ctx->fixed));
        ctx->fixed = nullptr;
        ctx->fixed_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->fixed), need));
        ctx->fixed_bytes = need;
    }
    return render_on(ctx, cam, width, height, sample_begin, sample_count, max_depth, seed, flags, accum_dev,
                     ctx->fixed, ctx->counters, ctx->ring, ctx->resident_blocks * caster::kWavesPerBlock);
}

int caster_hip_render_noalloc(caster_hip_ctx* ctx, const caster_camera* cam, int width, int height,
                              int sample_begin, int sample_count, int max_depth,
                              uint32_t seed, uint32_t flags, float* accum_dev)
{
    int rc = check_render_args(ctx, cam, width, height, sample_begin, sample_count, flags, accum_dev);
    if (rc) return rc;
    if ((size_t)width * (size_t)height > kNoAllocMaxPixels)
        return fail(CASTER_HIP_ERR_ARG, "caster_hip_render_noalloc: %dx%d = %zu pixels exceed the static scratch of %zu pixels; "
                                        "use caster_hip_render or render the frame in parts",
                    width, height, (size_t)width * (size_t)height, kNoAllocMaxPixels);
    if (!ctx->scratch) return fail(CASTER_HIP_ERR_STATE, "no scratch address for device %d", ctx->device);
    if ((size_t)ctx->resident_blocks * caster::kWavesPerBlock > kNoAllocMaxWaves)
        return fail(CASTER_HIP_ERR_ARG, "caster_hip_render_noalloc: this GPU runs %d waves of the render kernel at a time, the static "
                                        "scratch holds the path rings of %zu", ctx->resident_blocks * caster::kWavesPerBlock, kNoAllocMaxWaves);
    DeviceGuard guard(ctx->device);
    return render_on(ctx, cam, width, height, sample_begin, sample_count, max_depth, seed, flags, accum_dev,
                     ctx->scratch + kCounterWords, ctx->scratch,
                     reinterpret_cast<uint32_t*>(ctx->scratch + kCounterWords + 3 * kNoAllocMaxPixels), (int)kNoAllocMaxWaves);
}

int caster_hip_add_from_device(caster_hip_ctx* ctx, float* dst_dev, int src_device, const float* src_dev, size_t n_floats)
{
    if (!ctx || !dst_dev || !src_dev) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || src_device < 0 || src_device >= n_dev)
        return fail(CASTER_HIP_ERR_ARG, "source device %d out of range", src_device);
    if (n_floats == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    if (src_device == ctx->device) {        // the same GPU: no copy, the add kernel reads the source in place
        HIP_TRY(caster::launch_add(dst_dev, src_dev, n_floats, ctx->stream));
        return CASTER_HIP_OK;
    }
    const size_t bytes = n_floats * sizeof(float);
    if (bytes > ctx->staging_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->staging) HIP_TRY(hipFree(ctx->staging));
        ctx->staging = nullptr;
        ctx->staging_bytes = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->staging), bytes));
        ctx->staging_bytes = bytes;
    }
    HIP_TRY(hipMemcpyPeerAsync(ctx->staging, ctx->device, src_dev, src_device, bytes, ctx->stream));
    HIP_TRY(caster::launch_add(dst_dev, ctx->staging, n_floats, ctx->stream));
    return CASTER_HIP_OK;
}

int caster_hip_render_stats(caster_hip_ctx* ctx, uint64_t* rays, float* kernel_ms)
{
    if (!ctx) return fail(CASTER_HIP_ERR_ARG, "ctx is NULL");
    if (!ctx->render_pending) return fail(CASTER_HIP_ERR_STATE, "no render has been issued on this context");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (rays) {
        unsigned long long r = 0;
        HIP_TRY(hipMemcpy(&r, ctx->last_counters, sizeof r, hipMemcpyDeviceToHost));
        *rays = (uint64_t)r;
    }
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, ctx->ev0, ctx->ev1));
    return CASTER_HIP_OK;
}

int caster_hip_render_info(caster_hip_ctx* ctx, caster_render_info* out)
{
    if (!ctx || !out) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->render_pending) return fail(CASTER_HIP_ERR_STATE, "no render has been issued on this context");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long c[kCounterWords] = {};
    HIP_TRY(hipMemcpy(c, ctx->last_counters, sizeof c, hipMemcpyDeviceToHost));
    memset(out, 0, sizeof *out);
    out->loop_kind = ctx->last_kind;
    out->mfma_blocks = ctx->last_blocks;
    out->n_spheres = ctx->last_spheres;
    out->rays = (uint64_t)c[0];
    out->wave_trips = (uint64_t)c[2];
    out->primary_batches = (uint64_t)c[3];
    HIP_TRY(hipEventElapsedTime(&out->kernel_ms, ctx->ev0, ctx->ev1));
    return CASTER_HIP_OK;
}

int caster_hip_resolve_dev(caster_hip_ctx* ctx, const float* accum_dev, int width, int height,
                           int spp, uint8_t* rgba_dev)
{
    if (!ctx || !accum_dev || !rgba_dev) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (width <= 0 || height <= 0 || height > 65535) return fail(CASTER_HIP_ERR_ARG, "bad frame size");
    if (spp <= 0) return fail(CASTER_HIP_ERR_ARG, "spp must be positive");
    DeviceGuard guard(ctx->device);
    HIP_TRY(caster::launch_resolve(reinterpret_cast<const float4*>(accum_dev), width, height, spp, rgba_dev,
                                   ctx->stream));
    return CASTER_HIP_OK;
}

int caster_hip_resolve(caster_hip_ctx* ctx, const float* accum_dev, int width, int height,
                       int spp, uint8_t* rgba_host)
{
    if (!ctx || !accum_dev || !rgba_host) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (width <= 0 || height <= 0) return fail(CASTER_HIP_ERR_ARG, "bad frame size");
    DeviceGuard guard(ctx->device);
    const size_t bytes = (size_t)width * height * 4;
    DevBuf tmp;
    HIP_TRY(tmp.alloc(bytes));
    int rc = caster_hip_resolve_dev(ctx, accum_dev, width, height, spp, tmp.as<uint8_t>());
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rgba_host, tmp.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CASTER_HIP_OK;
}

// ---------------------------------------------------- per-function entry points
#define PICK(flags, base) (((flags) & CASTER_HIP_STRICT) ? caster::base##strict : caster::base##fast)

int caster_hip_test_hit_sphere(caster_hip_ctx* ctx, uint32_t flags, int n,
                               const float* sph4, const float* ray6, const float* trange2,
                               int32_t* hit, float* t, float* p3, float* n3, int32_t* face)
{
    if (!ctx || !sph4 || !ray6 || !trange2 || !hit || !t || !p3 || !n3 || !face)
        return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf ds, dr, dt, dhit, dtt, dp, dn, df;
    HIP_TRY(ds.put(sph4, sizeof(float) * 4 * n));
    HIP_TRY(dr.put(ray6, sizeof(float) * 6 * n));
    HIP_TRY(dt.put(trange2, sizeof(float) * 2 * n));
    HIP_TRY(dhit.alloc(sizeof(int32_t) * n));
    HIP_TRY(dtt.alloc(sizeof(float) * n));
    HIP_TRY(dp.alloc(sizeof(float) * 3 * n));
    HIP_TRY(dn.alloc(sizeof(float) * 3 * n));
    HIP_TRY(df.alloc(sizeof(int32_t) * n));
    HIP_TRY(PICK(flags, launch_test_hit_sphere_)(n, ds.as<float>(), dr.as<float>(), dt.as<float>(),
                                                 dhit.as<int32_t>(), dtt.as<float>(), dp.as<float>(),
                                                 dn.as<float>(), df.as<int32_t>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dhit.get(hit, sizeof(int32_t) * n));
    HIP_TRY(dtt.get(t, sizeof(float) * n));
    HIP_TRY(dp.get(p3, sizeof(float) * 3 * n));
    HIP_TRY(dn.get(n3, sizeof(float) * 3 * n));
    HIP_TRY(df.get(face, sizeof(int32_t) * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_hit_xy_rect(caster_hip_ctx* ctx, uint32_t flags, int n,
                                const float* rect5, const float* ray6, const float* trange2,
                                int32_t* hit, float* t, float* p3, float* n3, int32_t* face, float* uv2)
{
    if (!ctx || !rect5 || !ray6 || !trange2 || !hit || !t || !p3 || !n3 || !face || !uv2)
        return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf ds, dr, dt, dhit, dtt, dp, dn, df, duv;
    HIP_TRY(ds.put(rect5, sizeof(float) * 5 * n));
    HIP_TRY(dr.put(ray6, sizeof(float) * 6 * n));
    HIP_TRY(dt.put(trange2, sizeof(float) * 2 * n));
    HIP_TRY(dhit.alloc(sizeof(int32_t) * n));
    HIP_TRY(dtt.alloc(sizeof(float) * n));
    HIP_TRY(dp.alloc(sizeof(float) * 3 * n));
    HIP_TRY(dn.alloc(sizeof(float) * 3 * n));
    HIP_TRY(df.alloc(sizeof(int32_t) * n));
    HIP_TRY(duv.alloc(sizeof(float) * 2 * n));
    HIP_TRY(PICK(flags, launch_test_hit_rect_)(n, ds.as<float>(), dr.as<float>(), dt.as<float>(),
                                               dhit.as<int32_t>(), dtt.as<float>(), dp.as<float>(),
                                               dn.as<float>(), df.as<int32_t>(), duv.as<float>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dhit.get(hit, sizeof(int32_t) * n));
    HIP_TRY(dtt.get(t, sizeof(float) * n));
    HIP_TRY(dp.get(p3, sizeof(float) * 3 * n));
    HIP_TRY(dn.get(n3, sizeof(float) * 3 * n));
    HIP_TRY(df.get(face, sizeof(int32_t) * n));
    HIP_TRY(duv.get(uv2, sizeof(float) * 2 * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_hit_list(caster_hip_ctx* ctx, uint32_t flags, int n, const float* ray6,
                             int32_t* hit, float* t, float* p3, float* n3, int32_t* face, int32_t* mat)
{
    if (!ctx || !ray6 || !hit || !t || !p3 || !n3 || !face || !mat) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "no scene uploaded");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf dr, dhit, dtt, dp, dn, df, dm;
    HIP_TRY(dr.put(ray6, sizeof(float) * 6 * n));
    HIP_TRY(dhit.alloc(sizeof(int32_t) * n));
    HIP_TRY(dtt.alloc(sizeof(float) * n));
    HIP_TRY(dp.alloc(sizeof(float) * 3 * n));
    HIP_TRY(dn.alloc(sizeof(float) * 3 * n));
    HIP_TRY(df.alloc(sizeof(int32_t) * n));
    HIP_TRY(dm.alloc(sizeof(int32_t) * n));
    HIP_TRY(PICK(flags, launch_test_hit_list_)(ctx->scene, n, dr.as<float>(), dhit.as<int32_t>(), dtt.as<float>(),
                                               dp.as<float>(), dn.as<float>(), df.as<int32_t>(), dm.as<int32_t>(),
                                               ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dhit.get(hit, sizeof(int32_t) * n));
    HIP_TRY(dtt.get(t, sizeof(float) * n));
    HIP_TRY(dp.get(p3, sizeof(float) * 3 * n));
    HIP_TRY(dn.get(n3, sizeof(float) * 3 * n));
    HIP_TRY(df.get(face, sizeof(int32_t) * n));
    HIP_TRY(dm.get(mat, sizeof(int32_t) * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_trace(caster_hip_ctx* ctx, uint32_t flags, int n, const float* ray6,
                          int32_t* hit, float* t, float* p3, float* n3, int32_t* face, int32_t* mat)
{
    if (!ctx || !ray6 || !hit || !t || !p3 || !n3 || !face || !mat) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "no scene uploaded");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf dr, dhit, dtt, dp, dn, df, dm;
    HIP_TRY(dr.put(ray6, sizeof(float) * 6 * n));
    HIP_TRY(dhit.alloc(sizeof(int32_t) * n));
    HIP_TRY(dtt.alloc(sizeof(float) * n));
    HIP_TRY(dp.alloc(sizeof(float) * 3 * n));
    HIP_TRY(dn.alloc(sizeof(float) * 3 * n));
    HIP_TRY(df.alloc(sizeof(int32_t) * n));
    HIP_TRY(dm.alloc(sizeof(int32_t) * n));
    caster::SceneDev scene = ctx->scene;
    if (flags & CASTER_HIP_VECTOR_LOOP) scene.mfma_blocks = 0;
    HIP_TRY(PICK(flags, launch_test_trace_)(scene, n, dr.as<float>(), dhit.as<int32_t>(), dtt.as<float>(),
                                            dp.as<float>(), dn.as<float>(), df.as<int32_t>(), dm.as<int32_t>(),
                                            ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dhit.get(hit, sizeof(int32_t) * n));
    HIP_TRY(dtt.get(t, sizeof(float) * n));
    HIP_TRY(dp.get(p3, sizeof(float) * 3 * n));
    HIP_TRY(dn.get(n3, sizeof(float) * 3 * n));
    HIP_TRY(df.get(face, sizeof(int32_t) * n));
    HIP_TRY(dm.get(mat, sizeof(int32_t) * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_scatter(caster_hip_ctx* ctx, uint32_t flags, int n,
                            const float* ray6, const float* p3, const float* n3,
                            const int32_t* face, const int32_t* mat, uint32_t seed,
                            int32_t* ok, float* out_ray6, int32_t* draws)
{
    if (!ctx || !ray6 || !p3 || !n3 || !face || !mat || !ok || !out_ray6 || !draws)
        return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "no scene uploaded");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    for (int k = 0; k < n; ++k)
        if (mat[k] < 0 || mat[k] >= ctx->scene.n_mat)
            return fail(CASTER_HIP_ERR_ARG, "mat[%d] = %d out of range [0, %d)", k, mat[k], ctx->scene.n_mat);
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf dr, dp, dn, df, dm, dok, dout, dd;
    HIP_TRY(dr.put(ray6, sizeof(float) * 6 * n));
    HIP_TRY(dp.put(p3, sizeof(float) * 3 * n));
    HIP_TRY(dn.put(n3, sizeof(float) * 3 * n));
    HIP_TRY(df.put(face, sizeof(int32_t) * n));
    HIP_TRY(dm.put(mat, sizeof(int32_t) * n));
    HIP_TRY(dok.alloc(sizeof(int32_t) * n));
    HIP_TRY(dout.alloc(sizeof(float) * 6 * n));
    HIP_TRY(dd.alloc(sizeof(int32_t) * n));
    HIP_TRY(PICK(flags, launch_test_scatter_)(ctx->scene, n, dr.as<float>(), dp.as<float>(), dn.as<float>(),
                                              df.as<int32_t>(), dm.as<int32_t>(), seed, dok.as<int32_t>(),
                                              dout.as<float>(), dd.as<int32_t>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dok.get(ok, sizeof(int32_t) * n));
    HIP_TRY(dout.get(out_ray6, sizeof(float) * 6 * n));
    HIP_TRY(dd.get(draws, sizeof(int32_t) * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_camera_ray(caster_hip_ctx* ctx, uint32_t flags, const caster_camera* cam,
                               int n, const float* uv2, uint32_t seed, float* out_ray6, int32_t* draws)
{
    if (!ctx || !cam || !uv2 || !out_ray6 || !draws) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf duv, dout, dd;
    HIP_TRY(duv.put(uv2, sizeof(float) * 2 * n));
    HIP_TRY(dout.alloc(sizeof(float) * 6 * n));
    HIP_TRY(dd.alloc(sizeof(int32_t) * n));
    HIP_TRY(PICK(flags, launch_test_camera_)(cam_rec(cam), n, duv.as<float>(), seed, dout.as<float>(),
                                             dd.as<int32_t>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dout.get(out_ray6, sizeof(float) * 6 * n));
    HIP_TRY(dd.get(draws, sizeof(int32_t) * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_texture(caster_hip_ctx* ctx, uint32_t flags, int n, const int32_t* mat,
                            const float* uv2, const float* p3, float* rgb3)
{
    if (!ctx || !mat || !uv2 || !p3 || !rgb3) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "no scene uploaded");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    for (int k = 0; k < n; ++k)
        if (mat[k] < 0 || mat[k] >= ctx->scene.n_mat)
            return fail(CASTER_HIP_ERR_ARG, "mat[%d] = %d out of range [0, %d)", k, mat[k], ctx->scene.n_mat);
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf dm, duv, dp, dout;
    HIP_TRY(dm.put(mat, sizeof(int32_t) * n));
    HIP_TRY(duv.put(uv2, sizeof(float) * 2 * n));
    HIP_TRY(dp.put(p3, sizeof(float) * 3 * n));
    HIP_TRY(dout.alloc(sizeof(float) * 3 * n));
    HIP_TRY(PICK(flags, launch_test_texture_)(ctx->scene, n, dm.as<int32_t>(), duv.as<float>(), dp.as<float>(),
                                              dout.as<float>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dout.get(rgb3, sizeof(float) * 3 * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_rng(caster_hip_ctx* ctx, uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out23)
{
    if (!ctx || !out23) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf d;
    HIP_TRY(d.alloc(sizeof(uint32_t) * n));
    HIP_TRY(caster::launch_test_rng(seed, pixel, sample, n, d.as<uint32_t>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(d.get(out23, sizeof(uint32_t) * n));
    return CASTER_HIP_OK;
}

int caster_hip_test_math(caster_hip_ctx* ctx, uint32_t flags, int fn, int n, const float* in8, float* out3)
{
    if (!ctx || !in8 || !out3) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if (fn < 0 || fn > CASTER_MATH_TO_UNIT) return fail(CASTER_HIP_ERR_ARG, "unknown function %d", fn);
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf di, dout;
    HIP_TRY(di.put(in8, sizeof(float) * 8 * n));
    HIP_TRY(dout.alloc(sizeof(float) * 3 * n));
    HIP_TRY(PICK(flags, launch_test_math_)(fn, n, di.as<float>(), dout.as<float>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(dout.get(out3, sizeof(float) * 3 * n));
    return CASTER_HIP_OK;
}

int caster_hip_filter_audit(caster_hip_ctx* ctx, uint32_t flags, int n, const float* ray6,
                            uint32_t* accepted, uint32_t* missed, float* slack)
{
    if (!ctx || !ray6 || !accepted || !missed || !slack) return fail(CASTER_HIP_ERR_ARG, "NULL argument");
    if (!ctx->has_scene) return fail(CASTER_HIP_ERR_STATE, "no scene uploaded");
    if (n < 0) return fail(CASTER_HIP_ERR_ARG, "n < 0");
    if ((flags & ~(uint32_t)(CASTER_HIP_STRICT | CASTER_HIP_VECTOR_LOOP)) != 0)
        return fail(CASTER_HIP_ERR_ARG, "unknown flag bits 0x%x", flags);
    if (n == 0) return CASTER_HIP_OK;
    DeviceGuard guard(ctx->device);
    DevBuf dr, da, dm, ds;
    HIP_TRY(dr.put(ray6, sizeof(float) * 6 * n));
    HIP_TRY(da.alloc(sizeof(uint32_t) * n));
    HIP_TRY(dm.alloc(sizeof(uint32_t) * n));
    HIP_TRY(ds.alloc(sizeof(float) * n));
    caster::SceneDev scene = ctx->scene;
    const int strict = (flags & CASTER_HIP_STRICT) ? 1 : 0;
    // the same choice of loop as a render call makes
    if ((flags & CASTER_HIP_VECTOR_LOOP) || caster::pick_loop_kind(scene, strict) < caster::kLoopMatrix) scene.mfma_blocks = 0;
    HIP_TRY(PICK(flags, launch_filter_audit_)(scene, n, dr.as<float>(), da.as<uint32_t>(), dm.as<uint32_t>(),
                                              ds.as<float>(), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(da.get(accepted, sizeof(uint32_t) * n));
    HIP_TRY(dm.get(missed, sizeof(uint32_t) * n));
    HIP_TRY(ds.get(slack, sizeof(float) * n));
    return CASTER_HIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ RCCL reduce
// RCCL is opened at run time, so that the library loads (and everything but this entry point works)
// in processes and on machines without it, and so that a process which already holds an RCCL (a
// PyTorch process) uses that one.
namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::vector<int> devices;           // the communicators below were made for these devices, in this order
    std::vector<ncclComm_t> comms;
};
std::mutex g_rccl_mutex;
Rccl g_rccl;

int rccl_open()
{
    if (g_rccl.handle) return CASTER_HIP_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) return fail(CASTER_HIP_ERR_DEVICE, "RCCL not found (librccl.so.1): %s", dlerror());
#define RCCL_SYM(field, name)                                                             \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                 \
    if (!g_rccl.field) return fail(CASTER_HIP_ERR_DEVICE, "RCCL symbol %s not found", name)
    RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RCCL_SYM(CommAbort, "ncclCommAbort");
    RCCL_SYM(Reduce, "ncclReduce");
    RCCL_SYM(GroupStart, "ncclGroupStart");
    RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
    g_rccl.handle = h;
    return CASTER_HIP_OK;
}

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return fail(CASTER_HIP_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));  \
    } while (0)

}  // namespace

extern "C" int caster_hip_reduce(caster_hip_ctx* const* ctxs, float* const* accum_dev, int n, size_t n_floats)
{
    if (!ctxs || !accum_dev || n <= 0) return fail(CASTER_HIP_ERR_ARG, "NULL argument or n <= 0");
    std::vector<int> devices((size_t)n);
    for (int k = 0; k < n; ++k) {
        if (!ctxs[k] || !accum_dev[k]) return fail(CASTER_HIP_ERR_ARG, "context or buffer %d is NULL", k);
        devices[(size_t)k] = ctxs[k]->device;
        for (int j = 0; j < k; ++j)
            if (devices[(size_t)j] == devices[(size_t)k])
                return fail(CASTER_HIP_ERR_ARG, "contexts %d and %d are on the same device %d: sum those with "
                                                "caster_hip_add_from_device, RCCL wants one rank per GPU", j, k, devices[(size_t)k]);
    }
    if (n_floats == 0) return CASTER_HIP_OK;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int rc = rccl_open();
    if (rc) return rc;
    int prev = -1;
    (void)hipGetDevice(&prev);
    // Everything that can fail without RCCL's doing is checked before the group is opened: a group
    // that has taken some ranks' calls and not the others' must never be launched (the ranks that
    // were launched would wait for the missing ones for ever, on streams the caller then waits on).
    for (int k = 0; k < n; ++k) {
        if (hipSetDevice(devices[(size_t)k]) != hipSuccess) {
            if (prev >= 0) (void)hipSetDevice(prev);
            return fail(CASTER_HIP_ERR_DEVICE, "hipSetDevice(%d) failed", devices[(size_t)k]);
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    if (g_rccl.devices != devices) {
        for (ncclComm_t c : g_rccl.comms) (void)g_rccl.CommDestroy(c);
        g_rccl.comms.assign((size_t)n, nullptr);
        g_rccl.devices.clear();
        RCCL_TRY(g_rccl.CommInitAll(g_rccl.comms.data(), n, devices.data()));
        g_rccl.devices = devices;
    }
    // If a call of the group fails all the same, nothing of it is launched: the communicators are
    // ended first (their queued work is cancelled with them), the group is closed only to leave
    // RCCL's thread state clean, and the cache is dropped so that the next call makes new ones.
    auto give_up = [&]() {
        for (ncclComm_t c : g_rccl.comms)
            if (c) (void)g_rccl.CommAbort(c);
        (void)g_rccl.GroupEnd();
        g_rccl.comms.clear();
        g_rccl.devices.clear();
        if (prev >= 0) (void)hipSetDevice(prev);
    };
    // one rank per GPU, all driven from this thread: the calls of a group are issued together
    RCCL_TRY(g_rccl.GroupStart());
    for (int k = 0; k < n; ++k) {
        if (hipSetDevice(devices[(size_t)k]) != hipSuccess) {
            give_up();
            return fail(CASTER_HIP_ERR_DEVICE, "hipSetDevice(%d) failed inside the group; the communicators were ended, nothing was launched",
                        devices[(size_t)k]);
        }
        const ncclResult_t r = g_rccl.Reduce(accum_dev[k], accum_dev[k], n_floats, ncclFloat32, ncclSum, 0,
                                             g_rccl.comms[(size_t)k], ctxs[k]->stream);
        if (r != ncclSuccess) {
            const char* why = g_rccl.GetErrorString(r);
            give_up();
            return fail(CASTER_HIP_ERR_DEVICE, "ncclReduce on device %d failed: %s; the communicators were ended, nothing was launched",
                        devices[(size_t)k], why);
        }
    }
    const ncclResult_t r = g_rccl.GroupEnd();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (r != ncclSuccess) {
        // the group did not start as a whole: the communicators are not to be trusted again
        for (ncclComm_t c : g_rccl.comms)
            if (c) (void)g_rccl.CommAbort(c);
        g_rccl.comms.clear();
        g_rccl.devices.clear();
        return fail(CASTER_HIP_ERR_DEVICE, "ncclGroupEnd failed: %s; the communicators were ended", g_rccl.GetErrorString(r));
    }
    return CASTER_HIP_OK;
}
