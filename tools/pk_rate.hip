// pk_rate -- issue rate of packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against the scalar forms on
// this GPU: 8 independent chains per lane, 4 waves per SIMD, every CU busy.  Prints ns per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/pk_rate.hip -o build/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define N_ITER 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float s) {
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 m = {s, s + 1.0f};
    for (int it = 0; it < N_ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i].x = a[i].x * m.x;                                   // v_mul_f32
            if (MODE == 1) a[i] = a[i] * m;                                         // v_pk_mul_f32
            if (MODE == 2) a[i].x = __builtin_fmaf(a[i].x, m.x, m.y);               // v_fma_f32
            if (MODE == 3) a[i] = __builtin_elementwise_fma(a[i], m, m);            // v_pk_fma_f32
            if (MODE == 4) a[i].x = fminf(a[i].x, m.x + (float)it);                 // v_min + add
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = 4.0 * N_ITER * 8;  // waves per SIMD x instructions per wave
    printf("%-14s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / insts_per_simd,
           ms * 1e6 / insts_per_simd * 2.4);
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    run<0>("v_mul_f32", out);
    run<1>("v_pk_mul_f32", out);
    run<2>("v_fma_f32", out);
    run<3>("v_pk_fma_f32", out);
    run<4>("v_min+v_add", out);
    return 0;
}
