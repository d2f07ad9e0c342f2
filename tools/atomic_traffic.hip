// Calibration of rocprofv3's WRITE_SIZE for 64-bit integer atomics (no return value), the access the
// render kernel makes into its fixed-point frame buffer: N lanes add to consecutive words
// (k_contiguous: a wave covers 512 contiguous bytes), to every fourth word (k_stride4: one word per
// 32 bytes, the interleaved-channel layout) and to words 4 KiB apart (k_scattered).
// Build: hipcc --offload-arch=gfx950 -O2 tools/atomic_traffic.hip -o tools/atomic_traffic
// Run under: rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- tools/atomic_traffic
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_contiguous(unsigned long long* buf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(buf + i, 1ull);
}
__global__ void k_stride4(unsigned long long* buf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(buf + 4 * i, 1ull);
}
__global__ void k_scattered(unsigned long long* buf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(buf + 512 * (i % (n / 512)) + i / (n / 512), 1ull);
}

int main()
{
    const size_t n = (size_t)1 << 24;      // 16 Mi atomics per kernel = 128 MiB of payload
    unsigned long long* buf;
    if (hipMalloc(&buf, n * 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (hipMemset(buf, 0, n * 4 * sizeof(unsigned long long)) != hipSuccess) return 1;
    const dim3 grid((unsigned)(n / 256)), block(256);
    hipLaunchKernelGGL(k_contiguous, grid, block, 0, 0, buf, n);
    hipLaunchKernelGGL(k_stride4, grid, block, 0, 0, buf, n);
    hipLaunchKernelGGL(k_scattered, grid, block, 0, 0, buf, n);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    printf("atomics per kernel: %zu, payload %zu bytes\n", n, n * 8);
    (void)hipFree(buf);
    return 0;
}
