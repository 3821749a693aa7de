// fetch_calib -- calibrates rocprofv3's FETCH_SIZE for THIS code's access pattern (MI355X_MICROARCH.md: the
// factor 2 is measured for wide streaming reads, "other access widths are uncalibrated").
//   ./build/fetch_calib [table_MiB]      (hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o build/fetch_calib)
// Two kernels over one table far larger than the 256 MiB Infinity Cache, each reading every byte exactly once:
//   k_stream  16 B per lane, consecutive lanes consecutive addresses (the pattern the factor 2 was measured on)
//   k_gather  every lane reads one whole 64-byte record (4 x 16 B, as a BVH node fetch) at a pseudo-random
//             index: a bijective multiplicative permutation of the record index, so no record is read twice
//   k_gather32  the same with 32-byte records (2 x 16 B: a node of the tight tree since round 2's 16-bit grid)
// Known bytes per launch = table size.  Run under `rocprofv3 --pmc FETCH_SIZE`; FETCH_SIZE_KB * 1024 / bytes
// is the fraction the counter reports for each pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_stream(const float4* __restrict__ t, size_t n_f4, float* sink) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = t[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void k_gather(const float4* __restrict__ t, unsigned n_rec, unsigned mult, float* sink) {
    float acc = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const unsigned r = (unsigned)(((unsigned long long)i * mult) & (n_rec - 1));  // n_rec = 2^k, mult odd: a bijection
        const float4 a = t[4 * (size_t)r], b = t[4 * (size_t)r + 1], c = t[4 * (size_t)r + 2], d = t[4 * (size_t)r + 3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) *sink = acc;
}

__global__ void k_gather32(const float4* __restrict__ t, unsigned n_rec, unsigned mult, float* sink) {
    float acc = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const unsigned r = (unsigned)(((unsigned long long)i * mult) & (n_rec - 1));
        const float4 a = t[2 * (size_t)r], b = t[2 * (size_t)r + 1];
        acc += a.x + b.y;
    }
    if (acc == 12345.678f) *sink = acc;
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 2048;   // power of two
    const size_t bytes = mib << 20;
    float4* t;
    float* sink;
    if (hipMalloc(&t, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(t, 0, bytes);
    hipDeviceSynchronize();
    const unsigned n_rec = (unsigned)(bytes / 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, t, bytes / 16, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_stream  %zu MiB in %.3f ms = %.0f GB/s\n", mib, ms, bytes / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(256 * 16), dim3(256), 0, 0, t, n_rec, 2654435761u, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather  %zu MiB in %.3f ms = %.0f GB/s\n", mib, ms, bytes / ms / 1e6);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather32, dim3(256 * 16), dim3(256), 0, 0, t, (unsigned)(bytes / 32), 2654435761u, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("k_gather32 %zu MiB in %.3f ms = %.0f GB/s\n", mib, ms, bytes / ms / 1e6);
    }
    printf("bytes_per_launch %zu\n", bytes);
    return 0;
}
