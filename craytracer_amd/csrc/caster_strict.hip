// Strict arithmetic mode (reference operation order, no FMA; build this file
// with -ffp-contract=off) plus the kernels that do not depend on the mode.
#define CASTER_STRICT 1
#define CASTER_SFX strict
#pragma clang fp contract(off)
#include "caster_kernels.hpp"

namespace caster {

// accum[p] += the call's fixed-point sums, converted once; channel 3 counts samples.
__global__ void k_fold(const unsigned long long* fixed, int n_pixels, int sample_count, float4* accum)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    const unsigned long long fr = fixed[p], fg = fixed[(size_t)n_pixels + p], fb = fixed[2 * (size_t)n_pixels + p];   // planes R, G, B
    const double k = 1.0 / 4294967296.0;
    float4 acc = accum[p];
    acc.x = acc.x + (float)((double)fr * k);
    acc.y = acc.y + (float)((double)fg * k);
    acc.z = acc.z + (float)((double)fb * k);
    acc.w = acc.w + (float)sample_count;
    accum[p] = acc;
}

// dst[i] += src[i]: the sum of two GPUs' sample partitions (HBM bound: 12 bytes per float).
__global__ void k_add(float* dst, const float* src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = dst[i] + src[i];
}

// src/dispatcher.c:11,29-34 and src/image.c:30-36.
__global__ void k_resolve(const float4* accum, int width, int height, int spp, uint8_t* rgba)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= width || y >= height) return;
    const float scale = 1.0f / (float)spp;
    const int j = height - y - 1;
    const float4 a = accum[(size_t)j * width + x];
    float cr = sqrtf(a.x * scale), cg = sqrtf(a.y * scale), cb = sqrtf(a.z * scale);
    uchar4 o;
    o.x = (unsigned char)(clampf(cr, 0.0f, 0.999f) * 256.0f);
    o.y = (unsigned char)(clampf(cg, 0.0f, 0.999f) * 256.0f);
    o.z = (unsigned char)(clampf(cb, 0.0f, 0.999f) * 256.0f);
    o.w = 0;
    reinterpret_cast<uchar4*>(rgba)[(size_t)y * width + x] = o;
}

__global__ void k_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out23)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng g;
    rng_seed(g, seed, pixel, sample);
    for (int k = 0; k < n; ++k) out23[k] = rng_next23(g);
}

hipError_t launch_fold(const unsigned long long* fixed, int n_pixels, int sample_count, float4* accum, hipStream_t st)
{
    hipLaunchKernelGGL(k_fold, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, st, fixed, n_pixels,
                       sample_count, accum);
    return hipGetLastError();
}

hipError_t launch_add(float* dst, const float* src, size_t n, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_add, dim3((unsigned)blocks), dim3(256), 0, st, dst, src, n);
    return hipGetLastError();
}

hipError_t launch_resolve(const float4* accum, int width, int height, int spp, uint8_t* rgba, hipStream_t st)
{
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((width + 255) / 256), (unsigned)height), dim3(256), 0, st,
                       accum, width, height, spp, rgba);
    return hipGetLastError();
}

hipError_t launch_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out23, hipStream_t st)
{
    hipLaunchKernelGGL(k_test_rng, dim3(1), dim3(64), 0, st, seed, pixel, sample, n, out23);
    return hipGetLastError();
}

}  // namespace caster
