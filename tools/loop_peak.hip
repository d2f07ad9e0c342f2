// Micro-benchmark: the sphere loop of the render kernel in isolation (7 packed FP32
// instructions per sphere and lane pair, operands through the scalar cache, one
// max/compare per group of four, candidate branch never taken), at the render kernel's
// occupancy.  Prints shader cycles per group of four spheres and per-SIMD issue figures.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) v4f* cptr;
__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }

struct R { f2 ux, uy, uz, px, py, pz; float thrA, thrB; };

__device__ __forceinline__ f2 key(const R& r, const v4f f)
{
    f2 a = r.ux * splat(f.x);
    a = __builtin_elementwise_fma(r.uy, splat(f.y), a);
    a = __builtin_elementwise_fma(r.uz, splat(f.w), a);
    f2 b;
    const f2 zw = {f.z, f.w};
    asm("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(b) : "v"(r.pz), "s"(zw));
    b = __builtin_elementwise_fma(r.py, splat(f.y), b);
    b = __builtin_elementwise_fma(r.px, splat(f.x), b);
    return __builtin_elementwise_fma(a, a, b);
}

__global__ __launch_bounds__(256, 4) void k(const float4* tab, int n, int trips, const float* in, float* out,
                                            unsigned long long* cyc, int prio)
{
    cptr t = (cptr)tab;
    const int i = threadIdx.x;
    R r;
    r.ux = f2{in[i], in[i + 1]}; r.uy = f2{in[i + 2], in[i + 3]}; r.uz = f2{in[i + 4], in[i + 5]};
    r.px = f2{in[i + 6], in[i + 7]}; r.py = f2{in[i + 8], in[i + 9]}; r.pz = f2{in[i + 10], in[i + 11]};
    r.thrA = 1e30f; r.thrB = 1e30f;
    int cnt = 0;
    const unsigned long long c0 = clock64();
    for (int trip = 0; trip < trips; ++trip) {
        if (prio) {
            switch ((trip + blockIdx.x * 4 / gridDim.x) & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        v4f g0 = t[0], g1 = t[1], g2 = t[2], g3 = t[3];
#pragma unroll 1
        for (int kk = 0; kk < n; kk += 8) {
            const v4f h0 = t[kk + 4], h1 = t[kk + 5], h2 = t[kk + 6], h3 = t[kk + 7];
            {
                const f2 k0 = key(r, g0), k1 = key(r, g1), k2 = key(r, g2), k3 = key(r, g3);
                const float mA = fmaxf(fmaxf(k0.x, k1.x), fmaxf(k2.x, k3.x));
                const float mB = fmaxf(fmaxf(k0.y, k1.y), fmaxf(k2.y, k3.y));
                if (__any((mA >= r.thrA) | (mB >= r.thrB))) cnt++;
            }
            g0 = t[kk + 8]; g1 = t[kk + 9]; g2 = t[kk + 10]; g3 = t[kk + 11];
            {
                const f2 k0 = key(r, h0), k1 = key(r, h1), k2 = key(r, h2), k3 = key(r, h3);
                const float mA = fmaxf(fmaxf(k0.x, k1.x), fmaxf(k2.x, k3.x));
                const float mB = fmaxf(fmaxf(k0.y, k1.y), fmaxf(k2.y, k3.y));
                if (__any((mA >= r.thrA) | (mB >= r.thrB))) cnt++;
            }
        }
        r.ux += splat(1e-7f);      // keep the trips from being merged
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 256 + i] = cnt + r.ux.x;
    if (i == 0) atomicAdd(cyc, c1 - c0);
}

int main()
{
    const int n = 488, trips = 2000;
    std::vector<float4> tab(n + 16);
    for (int k = 0; k < n + 16; ++k) tab[k] = make_float4(0.01f * k, 0.2f, -1e3f, 0.02f * k);
    std::vector<float> in(512, 0.5f);
    float4* d_tab; float *d_in, *d_out; unsigned long long* d_cyc;
    hipMalloc(&d_tab, tab.size() * sizeof(float4)); hipMalloc(&d_in, in.size() * 4); hipMalloc(&d_out, 1024 * 256 * 4);
    hipMalloc(&d_cyc, 8);
    hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(float4), hipMemcpyHostToDevice);
    hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    for (int prio = 0; prio < 2; ++prio)
        for (int blocks : {256, 512, 1024}) {
            hipMemset(d_cyc, 0, 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_tab, n, 10, d_in, d_out, d_cyc, prio);
            hipDeviceSynchronize();
            hipMemset(d_cyc, 0, 8);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_tab, n, trips, d_in, d_out, d_cyc, prio);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
            const double per_group = (double)cyc / blocks / trips / (n / 4);
            const double waves_per_simd = blocks / 256.0;
            printf("prio %d, %d waves/SIMD: %.1f wall cycles per group and wave = %.1f per SIMD; %.2f ms, clock %.0f MHz\n", prio,
                   (int)waves_per_simd, per_group, per_group / waves_per_simd, ms, (double)cyc / blocks / (ms * 1e-3) * 1e-6);
        }
    return 0;
}
