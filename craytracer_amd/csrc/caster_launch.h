// Host-side launch interface between the C ABI (caster_api.cpp) and the two
// device translation units (caster_strict.hip, caster_fast.hip).  Both units
// export the same set of functions under a _strict / _fast suffix.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace caster {

struct MatRec {            // == caster_material
    int32_t type, tex;
    float c0[3], c1[3];
    float fuzz, ior;
    int32_t image, pad;
};
struct ImgRec {
    const uint8_t* data;   // device pointer
    uint32_t w, h, bpp, pad;
};
struct CamRec { float o[3], llc[3], h[3], v[3], u0[3], v0[3]; float lr; };

constexpr int kMaxAlways = 4;
constexpr float kFilterMarginStrict = 1.0e-5f;   // candidate-filter margins of the two arithmetic modes
constexpr float kFilterMarginFast = 2.5e-6f;
// The matrix-core form of the filter (see caster_render.hpp) adds the error of its two-piece
// half-precision operands and of the matrix unit's accumulation to those margins.
constexpr float kFilterMarginMfmaExtra = 3.0e-5f;
constexpr int kMfmaMinSpheres = 33;   // below this the vector loop is as fast or faster (measured: equal at 32, 1.4 x slower at 256)
constexpr int kSphTile = 512;         // spheres per tile: their exact image (8 KiB) is staged in LDS

struct SceneDev {
    const float4* sph;       // cx, cy, cz, r            (list order among spheres)
    const int2*   sph_meta;  // list position, material
    const float4* filter_img[2]; // the filter entries as the loop reads them (fast, strict), padded with 16 never-candidates
    const float4* rect_a;    // cx, cy, cz, du
    const float4* rect_b;    // dv, 0, 0, 0
    const int2*   rect_meta;
    const MatRec* mats;
    const ImgRec* imgs;
    int n_sph, n_rect, n_mat, n_img;
    float ref[3];            // P, the reference point of the candidate filter
    const float4* noise_vec; // 256 unit vectors (w unused), or null
    const uint8_t* noise_perm; // perm_x[256], perm_y[256], perm_z[256], or null
    const uint4* mfma_tab[2];// matrix-core filter operands (fast, strict): [block][2][64 lanes] x 8 halves, or null
    int mfma_blocks;         // blocks of 32 spheres in mfma_tab; 0: the scene does not qualify
    float mfma_scale;        // power of two that brings the scene into half-precision range
    int all_solid;           // every material's texture is a solid colour (TTYPE_COLOR): the kernel may leave the texture code out
    int n_always;            // spheres tested directly for every ray, outside the filter
    int always[kMaxAlways];  // their indices
};

// The sphere loop of a launch: the packed vector filter with the exact sphere image in LDS, the
// same streaming the image through LDS in tiles (more than kSphTile spheres), or the matrix-core
// filter with the exact image in LDS (up to kSphTile spheres) or in device memory (more), for
// scenes that qualify.
enum LoopKind { kLoopVector = 0, kLoopVectorTiled = 1, kLoopMatrix = 2, kLoopMatrixLarge = 3 };

// The one rule that picks the loop, used by the launcher and reported by caster_hip_render_info.
// `strict` indexes the per-mode tables (0 fast, 1 strict).
inline int pick_loop_kind(const SceneDev& s, int strict)
{
    const bool large = s.n_sph > kSphTile;
    const bool matrix = s.n_sph >= kMfmaMinSpheres && s.mfma_blocks > 0 && s.mfma_tab[strict] != nullptr;
    return matrix ? (large ? kLoopMatrixLarge : kLoopMatrix) : (large ? kLoopVectorTiled : kLoopVector);
}

constexpr int kMaxChunks = 256;

// Paths between their first and their second ray wait in a ring of the wave in device memory (see
// "primary rays" in caster_render.hpp): kRingEntries path states of kRingWords 32-bit words, word-major.
constexpr int kRingEntries = 64;
constexpr int kRingWords = 15;
constexpr size_t kRingBytesPerWave = (size_t)kRingEntries * kRingWords * sizeof(uint32_t);
constexpr int kWavesPerBlock = 4;

struct RenderArgs {
    CamRec cam;
    SceneDev scene;
    int width, height;
    int s_begin, s_count;    // sample range of this call
    int n_chunks;            // work units per pixel tile
    int chunk_start[kMaxChunks + 1];   // chunk c covers samples s_begin + [chunk_start[c], chunk_start[c+1])
    int max_depth;
    uint32_t seed;
    unsigned long long* fixed; // three planes (R, G, B) of width * height fixed-point sums (2^-32 units), zero at launch
    unsigned long long* rays;
    unsigned int* queue;     // work-unit queue head, zero at launch
    unsigned long long* trips; // wave trips through the sphere loop (128 ray slots each), zero at launch
    unsigned long long* batches; // batches of 64 camera rays traced against their tile's sphere list, zero at launch
    uint32_t* ring;          // kRingBytesPerWave bytes for every wave of the launch; contents need no initialisation
    int ring_waves;          // waves the ring buffer has room for
    int tile_lists;          // 0: camera rays meet every sphere (CASTER_HIP_NO_TILE_LISTS)
    int n_units;             // filled in by the launcher
    int lds_spheres;         // float4 slots per sphere image in LDS (filled in by the launcher)
};

#define CASTER_DECLARE_LAUNCHERS(SFX)                                                             \
    hipError_t launch_render_##SFX(const RenderArgs& a, int resident_blocks, hipStream_t st);                          \
    hipError_t launch_test_hit_sphere_##SFX(int n, const float* sph4, const float* ray6,          \
                                            const float* tr2, int32_t* hit, float* t, float* p3,  \
                                            float* n3, int32_t* face, hipStream_t st);            \
    hipError_t launch_test_hit_rect_##SFX(int n, const float* rect5, const float* ray6,           \
                                          const float* tr2, int32_t* hit, float* t, float* p3,    \
                                          float* n3, int32_t* face, float* uv2, hipStream_t st);  \
    hipError_t launch_test_hit_list_##SFX(const SceneDev& sc, int n, const float* ray6,           \
                                          int32_t* hit, float* t, float* p3, float* n3,           \
                                          int32_t* face, int32_t* mat, hipStream_t st);           \
    hipError_t launch_test_trace_##SFX(const SceneDev& sc, int n, const float* ray6,              \
                                       int32_t* hit, float* t, float* p3, float* n3,              \
                                       int32_t* face, int32_t* mat, hipStream_t st);              \
    hipError_t launch_test_scatter_##SFX(const SceneDev& sc, int n, const float* ray6,            \
                                         const float* p3, const float* n3, const int32_t* face,   \
                                         const int32_t* mat, uint32_t seed, int32_t* ok,          \
                                         float* out6, int32_t* draws, hipStream_t st);            \
    hipError_t launch_test_camera_##SFX(const CamRec& cam, int n, const float* uv2, uint32_t seed,\
                                        float* out6, int32_t* draws, hipStream_t st);             \
    hipError_t launch_test_texture_##SFX(const SceneDev& sc, int n, const int32_t* mat,           \
                                         const float* uv2, const float* p3, float* rgb3,          \
                                         hipStream_t st);                                         \
    hipError_t launch_test_math_##SFX(int fn, int n, const float* in8, float* out3,               \
                                      hipStream_t st);                                            \
    hipError_t launch_filter_audit_##SFX(const SceneDev& sc, int n, const float* ray6,            \
                                         uint32_t* accepted, uint32_t* missed, float* slack,      \
                                         hipStream_t st);

CASTER_DECLARE_LAUNCHERS(strict)
CASTER_DECLARE_LAUNCHERS(fast)

// Mode-independent kernels (defined in the strict unit).
hipError_t launch_fold(const unsigned long long* fixed, int n_pixels, int sample_count, float4* accum, hipStream_t st);
hipError_t launch_add(float* dst, const float* src, size_t n, hipStream_t st);
hipError_t launch_resolve(const float4* accum, int width, int height, int spp, uint8_t* rgba, hipStream_t st);
hipError_t launch_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out23, hipStream_t st);

}  // namespace caster
