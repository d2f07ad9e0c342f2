// Micro-benchmark: the matrix loop of the render kernel in isolation, at the render kernel's
// occupancy (4 blocks of 4 waves per CU): per column block two dependent v_mfma_f32_32x32x16_f16
// (variant 0) or four independent v_mfma_f32_16x16x32_f16 (variant 1), then the 16 sign masks and
// the test for candidates (never taken); variant 2 issues the matrix instructions of variant 0 alone
// (one register of the result is looked at), variant 3 the sign masks and the test alone (on the
// operand registers), variant 4 the matrix instructions and the OR of the 16 results that tells
// whether the column block has a candidate (what the render kernel does where it has none).  Prints
// the time per column block and SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4x __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned push_sign(unsigned m, float x) { return __builtin_amdgcn_alignbit(m, __float_as_uint(x), 31u); }

template <int V>
__device__ __forceinline__ unsigned column(const h8 a0, const h8 a1, const h8 b0, const h8 b1)
{
    unsigned cand = 0u;
    if (V == 0) {
        f16x acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) cand = push_sign(cand, acc[v]);
    } else if (V == 4) {
        f16x acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
        unsigned any = __float_as_uint(acc[0]);
#pragma unroll
        for (int v = 1; v < 15; v += 2) any = any | __float_as_uint(acc[v]) | __float_as_uint(acc[v + 1]);
        any |= __float_as_uint(acc[15]);
        cand = any >> 31;
    } else if (V == 2) {
        f16x acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
        cand = push_sign(cand, acc[0]);
    } else if (V == 3) {
        const uint4 x = __builtin_bit_cast(uint4, a0), y = __builtin_bit_cast(uint4, b0), z = __builtin_bit_cast(uint4, b1), w = __builtin_bit_cast(uint4, a1);
        const unsigned r[16] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w, z.x, z.y, z.z, z.w, w.x, w.y, w.z, w.w};
#pragma unroll
        for (int v = 0; v < 16; ++v) cand = __builtin_amdgcn_alignbit(cand, r[v], 31u);
    } else {
        const f4x z = {0, 0, 0, 0};
        const f4x c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, z, 0, 0, 0);
        const f4x c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, z, 0, 0, 0);
        const f4x c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, z, 0, 0, 0);
        const f4x c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, z, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) cand = push_sign(cand, c0[v]);
#pragma unroll
        for (int v = 0; v < 4; ++v) cand = push_sign(cand, c1[v]);
#pragma unroll
        for (int v = 0; v < 4; ++v) cand = push_sign(cand, c2[v]);
#pragma unroll
        for (int v = 0; v < 4; ++v) cand = push_sign(cand, c3[v]);
    }
    return cand;
}

template <int V>
__global__ __launch_bounds__(256, 4) void k(const uint4* tab, int n_blocks, int trips, const uint4* rays, unsigned* out)
{
    const int lane = threadIdx.x & 63;
    h8 b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = __builtin_bit_cast(h8, rays[i * 64 + lane]);
    unsigned found = 0;
    const uint4* t = tab + lane;
    for (int trip = 0; trip < trips; ++trip) {
        uint4 a0 = t[0], a1 = t[64];
#pragma unroll 1
        for (int blk = 0; blk < n_blocks; ++blk) {
            const uint4 n0 = t[(blk + 1) * 128], n1 = t[(blk + 1) * 128 + 64];
            const h8 A0 = __builtin_bit_cast(h8, a0), A1 = __builtin_bit_cast(h8, a1);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const unsigned cand = column<V>(A0, A1, b[2 * cb], b[2 * cb + 1]);
                if (__ballot(cand != 0u) != 0ull) found += cand;
            }
            a0 = n0; a1 = n1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = found;
}

int main()
{
    const int n_blocks = 16, trips = 4000;
    // all products positive: no sign bit set, the candidate branch is never taken
    std::vector<_Float16> tab((n_blocks + 1) * 128 * 8, (_Float16)0.25f), rays(8 * 64 * 8, (_Float16)0.5f);
    uint4 *d_tab, *d_rays; unsigned* d_out;
    (void)hipMalloc(&d_tab, tab.size() * 2); (void)hipMalloc(&d_rays, rays.size() * 2);
    (void)hipMemcpy(d_tab, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_rays, rays.data(), rays.size() * 2, hipMemcpyHostToDevice);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 4;
    (void)hipMalloc(&d_out, blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[5] = {"2 x 32x32x16, dependent", "4 x 16x16x32, independent", "matrix instructions only", "sign masks only",
                            "2 x 32x32x16 and the OR of the results"};
    for (int v = 0; v < 5; ++v) {
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_tab, n_blocks, trips, d_rays, d_out);
            else if (v == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_tab, n_blocks, trips, d_rays, d_out);
            else if (v == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d_tab, n_blocks, trips, d_rays, d_out);
            else if (v == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d_tab, n_blocks, trips, d_rays, d_out);
            else hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d_tab, n_blocks, trips, d_rays, d_out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: 4 waves x trips x n_blocks x 4 column blocks
            const double per_cb = ms * 1e6 / (4.0 * trips * n_blocks * 4.0);
            printf("variant %d (%s) rep %d: %.2f ms, %.1f ns per column block and SIMD (%.0f cycles at 2.1 GHz)\n", v,
                   names[v], rep, ms, per_cb, per_cb * 2.1);
        }
    }
    unsigned h = 0; (void)hipMemcpy(&h, d_out, 4, hipMemcpyDeviceToHost);
    printf("check %u\n", h);
    return 0;
}
