// Micro-benchmark: sustained FP32 VALU rate on gfx950 for v_fma_f32 and
// v_pk_fma_f32, at several waves per SIMD.  Prints TFLOP/s (FMA = 2 flop) and the
// shader clock each kernel ran at (cycle counter against the 100 MHz real-time counter).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ unsigned long long g_probe[4];
#define PROBE_BEGIN() \
    unsigned long long c0 = 0, w0 = 0; \
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
#define PROBE_END() \
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_probe[0] = clock64() - c0; g_probe[1] = wall_clock64() - w0; }

template <int ILP>
__global__ void k_fma(float* out, int iters, float a, float b)
{
    float acc[ILP];
    PROBE_BEGIN();
#pragma unroll
    for (int k = 0; k < ILP; ++k) acc[k] = threadIdx.x * 0.001f + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(a), "v"(b));
    }
    PROBE_END();
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ILP>
__global__ void k_pkfma(float* out, int iters, float a, float b)
{
    f2 acc[ILP];
    f2 va = {a, a}, vb = {b, b};
    PROBE_BEGIN();
#pragma unroll
    for (int k = 0; k < ILP; ++k) acc[k] = f2{threadIdx.x * 0.001f + k, threadIdx.x * 0.002f + k};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) acc[k] = __builtin_elementwise_fma(acc[k], va, vb);
    }
    PROBE_END();
    float s = 0;
#pragma unroll
    for (int k = 0; k < ILP; ++k) s += acc[k].x + acc[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static double run(K kern, int blocks, int threads, int iters, float* d_out, double* mhz)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0001f, 0.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1.0001f, 0.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long pr[4];
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof pr);
    *mhz = pr[1] ? 100.0 * (double)pr[0] / (double)pr[1] : 0.0;
    return ms * 1e-3;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device: %s, CUs %d, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float* d_out;
    hipMalloc(&d_out, sizeof(float) * 256 * 32 * 1024);
    const int iters = 400000;
    constexpr int ILP = 8;
    double m1, m2;
    for (int cus : {8, p.multiProcessorCount}) {    // a few CUs busy, then all of them
        for (int wps : {1, 2, 4, 8}) {              // waves per SIMD
            const int threads = 256;                // 4 waves per block -> 1 per SIMD
            const int blocks = cus * wps;
            double s1 = run(k_fma<ILP>, blocks, threads, iters, d_out, &m1);
            double f1 = 2.0 * ILP * iters * (double)blocks * threads / s1 * 1e-12;
            double s2 = run(k_pkfma<ILP>, blocks, threads, iters, d_out, &m2);
            double f2_ = 4.0 * ILP * iters * (double)blocks * threads / s2 * 1e-12;
            printf("blocks %5d waves/SIMD %d: v_fma_f32 %.1f TFLOP/s (%.3f ms, %.0f MHz)   v_pk_fma_f32 %.1f TFLOP/s (%.3f ms, %.0f MHz)\n",
                   blocks, wps, f1, s1 * 1e3, m1, f2_, s2 * 1e3, m2);
        }
    }
    return 0;
}
