// Developer check: division of a pixel coordinate plus jitter by (size - 1), the quotient the
// camera code needs once per path start, through a reciprocal refined once per denominator and the
// compiler's own correction chain without its range scaling -- must equal the correctly rounded
// `/` bit for bit for numerators in [0, 65536) and integer denominators in [1, 65535].
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float div_by(float n, float d, float r)
{
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
}

__global__ void k(unsigned long long* bad, unsigned long long* seen, unsigned seed)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = tid * 2654435761u + seed;
    unsigned long long nbad = 0, nseen = 0;
    for (int it = 0; it < 4096; ++it) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const unsigned den = 1u + (x % 65535u);
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const unsigned i = x % (den + 1u);                      // pixel index 0 .. den
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const float jitter = (float)(x >> 8) * 5.9604644775390625e-8f;
        const float d = (float)den;
        const float n = (float)i + jitter;
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float e = __builtin_fmaf(-d, r0, 1.0f);
        const float r = __builtin_fmaf(e, r0, r0);
        const float a = n / d, b = div_by(n, d, r);
        nbad += (__float_as_uint(a) != __float_as_uint(b));
        ++nseen;
        // the exact pixel centres and edges too
        const float a2 = (float)i / d, b2 = div_by((float)i, d, r);
        nbad += (__float_as_uint(a2) != __float_as_uint(b2));
        ++nseen;
    }
    atomicAdd(bad, nbad);
    atomicAdd(seen, nseen);
}

// Second domain: three components over one denominator (a normal (p - c) / r, a vector over its
// length), taken through the chain when all four magnitudes lie in [2^-40, 2^40].
__global__ void k3(unsigned long long* bad, unsigned long long* seen, unsigned seed)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = tid * 2654435761u + seed;
    unsigned long long nbad = 0, nseen = 0;
    for (int it = 0; it < 4096; ++it) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        // exponent 87 .. 167 (2^-40 .. 2^40), any mantissa, either sign; every 16th mantissa all ones or all zeros
        unsigned mant = x & 0x7FFFFFu;
        if ((x >> 28) == 0u) mant = 0x7FFFFFu;
        if ((x >> 28) == 1u) mant = 0u;
        const unsigned ed = 87u + (x >> 8) % 80u;
        const float d = __uint_as_float((ed << 23) | mant);
        const float r0 = __builtin_amdgcn_rcpf(d);
        const float e = __builtin_fmaf(-d, r0, 1.0f);
        const float r = __builtin_fmaf(e, r0, r0);
        for (int c = 0; c < 3; ++c) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            unsigned mn = x & 0x7FFFFFu;
            if ((x >> 28) == 2u) mn = 0x7FFFFFu;
            if ((x >> 28) == 3u) mn = 0u;
            if ((x >> 28) == 4u) mn = mant;                     // quotient a power of two
            const unsigned en = 87u + (x >> 9) % 80u;
            const float n = __uint_as_float((x & 0x80000000u) | (en << 23) | mn);
            const float a = n / d, b = div_by(n, d, r);
            nbad += (__float_as_uint(a) != __float_as_uint(b));
            ++nseen;
        }
    }
    atomicAdd(bad, nbad);
    atomicAdd(seen, nseen);
}

int main()
{
    unsigned long long *d_bad, *d_seen, h[2] = {0, 0};
    (void)hipMalloc(&d_bad, 8); (void)hipMalloc(&d_seen, 8);
    (void)hipMemset(d_bad, 0, 8); (void)hipMemset(d_seen, 0, 8);
    for (unsigned s = 0; s < 8; ++s) hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d_bad, d_seen, s * 977u + 1u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h[0], d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h[1], d_seen, 8, hipMemcpyDeviceToHost);
    printf("pixel coordinates: quotients compared %llu, different %llu\n", h[1], h[0]);
    const unsigned long long first = h[0];
    (void)hipMemset(d_bad, 0, 8); (void)hipMemset(d_seen, 0, 8);
    for (unsigned s = 0; s < 8; ++s) hipLaunchKernelGGL(k3, dim3(4096), dim3(256), 0, 0, d_bad, d_seen, s * 1013u + 7u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h[0], d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h[1], d_seen, 8, hipMemcpyDeviceToHost);
    printf("vector over scalar, magnitudes in [2^-40, 2^40]: quotients compared %llu, different %llu\n", h[1], h[0]);
    return (first | h[0]) != 0;
}
