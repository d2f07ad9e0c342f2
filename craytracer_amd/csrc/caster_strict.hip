// Strict arithmetic mode (reference operation order, no FMA; build this file
// with -ffp-contract=off) plus the kernels that do not depend on the mode.
#define CASTER_STRICT 1
#define CASTER_SFX strict
#pragma clang fp contract(off)
#include "caster_kernels.hpp"

namespace caster {

// accum[p] += slab[0][p] + slab[1][p] + ... in chunk order: the order is fixed,
// so a render is bit-reproducible whatever order the units ran in.
__global__ void k_combine(const float4* slabs, int n_chunks, int n_pixels, float4* accum)
{
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    float4 acc = accum[p];
    for (int c = 0; c < n_chunks; ++c) {
        const float4 s = slabs[(size_t)c * n_pixels + p];
        acc.x = acc.x + s.x; acc.y = acc.y + s.y; acc.z = acc.z + s.z; acc.w = acc.w + s.w;
    }
    accum[p] = acc;
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

__global__ void k_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out24)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng g;
    rng_seed(g, seed, pixel, sample);
    for (int k = 0; k < n; ++k) out24[k] = rng_next24(g);
}

hipError_t launch_combine(const float4* slabs, int n_chunks, int n_pixels, float4* accum, hipStream_t st)
{
    hipLaunchKernelGGL(k_combine, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, st, slabs, n_chunks,
                       n_pixels, accum);
    return hipGetLastError();
}

hipError_t launch_resolve(const float4* accum, int width, int height, int spp, uint8_t* rgba, hipStream_t st)
{
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((width + 255) / 256), (unsigned)height), dim3(256), 0, st,
                       accum, width, height, spp, rgba);
    return hipGetLastError();
}

hipError_t launch_test_rng(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out24, hipStream_t st)
{
    hipLaunchKernelGGL(k_test_rng, dim3(1), dim3(64), 0, st, seed, pixel, sample, n, out24);
    return hipGetLastError();
}

}  // namespace caster
