// valu_rate -- issue interval of the VALU instructions the render kernels are made of, on this GPU.
//
// The roofline of a cache-resident scene is VALU issue.  Round 2 priced every VALU instruction at the
// 2.5 cycles tools/pk_rate.hip measured for v_mul/v_fma; the kernels' correctly rounded `/` and `sqrt` expand to
// v_rcp/v_sqrt (transcendental unit), v_div_scale (two results), v_div_fmas (reads VCC), v_div_fixup, and the
// traversal is full of v_cmp + v_cndmask.  This program measures each of them the way pk_rate measured v_mul:
// 8 independent chains per lane, 32 workgroups of 4 waves per CU (every SIMD holds as many waves as it can take
// for the whole run: with exactly 4 waves per SIMD's worth of workgroups the dispatcher filled some CUs to 8 waves per
// SIMD and left others empty, and the figures depended on it), one instruction kind per kernel (inline asm, so
// the compiler cannot fuse or drop anything), and the two composites as the compiler expands them (x = y / x and
// x = sqrtf(x) under -fhip-fp32-correctly-rounded-divide-sqrt).  Run under `rocprofv3 --pmc SQ_INSTS_VALU
// SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_WAVES` the same kernels calibrate what
// the counters mean (one row per kernel: its name carries the mode).
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt tools/valu_rate.hip -o build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 512   // chain steps per lane and chain of the short run; the long run takes 4 x as many
#define WG_PER_CU 32
enum { M_MUL, M_FMA, M_ADD, M_MAX, M_RCP, M_SQRT, M_RSQ, M_EXP, M_LOG, M_DIV_SCALE, M_DIV_FMAS, M_DIV_FIXUP, M_CNDMASK, M_CMP,
       M_CMP_CNDMASK, M_CVT, M_MOV, M_MUL_LEGACY, M_FULL_DIV, M_FULL_SQRT, M_FULL_POW, M_N };
static const char* NAMES[M_N] = {"v_mul_f32", "v_fma_f32", "v_add_f32", "v_max_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32",
                                 "v_exp_f32", "v_log_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32", "v_cndmask_b32",
                                 "v_cmp_lt_f32", "v_cmp+v_cndmask", "v_cvt_f32_u32", "v_mov_b32", "v_sub_f32(sdwa-free)",
                                 "x = y / x  (correctly rounded)", "x = sqrtf(x) (correctly rounded)", "x = powf(x, s)"};
// instructions of one kind issued per chain step (the composites: counted in the ISA, see main)
template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float* out, float s, int n_iter) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0f + 0.001f * (float)(threadIdx.x + 1) + (float)i;
    const float m = s, c = s + 0.25f;
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == M_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == M_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == M_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == M_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == M_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (MODE == M_DIV_SCALE) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == M_DIV_FMAS) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            if (MODE == M_DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (MODE == M_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m) : "vcc");
            if (MODE == M_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            if (MODE == M_CMP_CNDMASK)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(m), "v"(c) : "vcc");
            if (MODE == M_CVT) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            if (MODE == M_MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == M_MUL_LEGACY) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == M_FULL_DIV) a[i] = c / a[i];
            if (MODE == M_FULL_SQRT) a[i] = sqrtf(a[i]) + m;
            if (MODE == M_FULL_POW) a[i] = powf(a[i], m);
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// Two runs of different length: the difference of their times is free of the launch's fixed cost (~10-40 us here,
// as much as the short run itself takes for the one-cycle instructions).
template <int MODE>
void run(float* out, double per_step) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int blocks = 256 * WG_PER_CU;
    float ms[2];
    for (int r = 0; r < 2; r++) {
        const int n = r ? 4 * N_ITER : N_ITER;
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, n);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, n);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[r], e0, e1);
    }
    const double steps_per_simd = (double)WG_PER_CU * (3 * N_ITER) * 8;  // waves per SIMD x extra chain steps per wave of the long run
    const double ns = (ms[1] - ms[0]) * 1e6 / steps_per_simd;
    printf("mode %2d %-34s %.3f / %.3f ms  %7.3f ns per step per SIMD = %6.2f cycles at 2.4 GHz", MODE, NAMES[MODE], ms[0], ms[1], ns, ns * 2.4);
    if (per_step > 1.0) printf("  (%.1f VALU instructions per step: %.2f cycles each)", per_step, ns * 2.4 / per_step);
    printf("\n");
}
int main(int argc, char** argv) {
    float* out;
    (void)hipMalloc(&out, (size_t)256 * WG_PER_CU * 256 * 4);
    // VALU instructions per step of the composites, counted in this build's ISA (tools/valu_rate_isa.sh prints them);
    // pass them as arguments to have the per-instruction figure printed
    // (from the PMC pass of round 3, SQ_INSTS_VALU per step: 11.0, 16.5 -- the sqrt and the add beside it -- and 165.5)
    const double n_div = argc > 1 ? atof(argv[1]) : 11.0, n_sqrt = argc > 2 ? atof(argv[2]) : 16.5, n_pow = argc > 3 ? atof(argv[3]) : 165.5;
    run<M_MUL>(out, 1), run<M_FMA>(out, 1), run<M_ADD>(out, 1), run<M_MAX>(out, 1), run<M_RCP>(out, 1), run<M_SQRT>(out, 1);
    run<M_RSQ>(out, 1), run<M_EXP>(out, 1), run<M_LOG>(out, 1), run<M_DIV_SCALE>(out, 1), run<M_DIV_FMAS>(out, 1);
    run<M_DIV_FIXUP>(out, 1), run<M_CNDMASK>(out, 1), run<M_CMP>(out, 1), run<M_CMP_CNDMASK>(out, 2), run<M_CVT>(out, 1);
    run<M_MOV>(out, 1), run<M_MUL_LEGACY>(out, 1), run<M_FULL_DIV>(out, n_div), run<M_FULL_SQRT>(out, n_sqrt), run<M_FULL_POW>(out, n_pow);
    return 0;
}
