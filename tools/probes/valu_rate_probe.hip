// VALU issue-rate probe (gfx950): cycles per wave instruction and SIMD for the instructions the tile kernels are made of --
// v_fma_f64, v_rsq_f64, v_rsq_f32 (+ the two conversions), v_mov_b32 with a DPP wave rotation.  Four waves per SIMD, independent
// chains, so that latencies are hidden and the issue rate shows.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe valu_rate_probe.hip ; run: ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k_probe(double *out, int iters) {
    double a[8];
    float f[8];
    int q[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { a[k] = 1.0+threadIdx.x*1e-3+k; f[k] = (float)a[k]; q[k] = threadIdx.x+k; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) a[k] = __builtin_fma(a[k], 0.999, 1e-3);
            if (MODE == 1) a[k] = __builtin_amdgcn_rsq(a[k]);
            if (MODE == 2) f[k] = __builtin_amdgcn_rsqf(f[k]);
            if (MODE == 3) a[k] = (double)__builtin_amdgcn_rsqf((float)a[k]);
            if (MODE == 4) q[k] = __builtin_amdgcn_update_dpp(0, q[k], 0x134, 0xf, 0xf, true);
            if (MODE == 5) q[k] = __builtin_amdgcn_update_dpp(0, q[k], 0x121, 0xf, 0xf, true);     // row_ror:1
            if (MODE == 6) a[k] = a[k]*1.0000001;
        }
    }
    double s = 0.;
#pragma unroll
    for (int k = 0; k < 8; k++) s += a[k]+f[k]+q[k];
    out[blockIdx.x*256+threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, double *out, int per_iter) {
    const int iters = 20000, grid = 256*4;            // four workgroups of four waves per CU: four waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double clk = 2.4e9;
    printf("%-28s %8.3f ms, %6.2f cycles per wave instruction and SIMD (%d instructions per step)\n", name, ms,
           ms*1e-3*clk/((double)iters*8*per_iter*4), per_iter);
}

int main() {
    double *out;
    if (hipMalloc(&out, sizeof(double)*256*1024) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    run<0>("v_fma_f64", out, 1);
    run<6>("v_mul_f64", out, 1);
    run<1>("v_rsq_f64", out, 1);
    run<2>("v_rsq_f32", out, 1);
    run<3>("cvt + v_rsq_f32 + cvt", out, 3);
    run<4>("v_mov_b32 dpp wave_rol:1", out, 1);
    run<5>("v_mov_b32 dpp row_ror:1", out, 1);
    (void)hipFree(out);
    return 0;
}
