// LDS throughput probe (gfx950): cycles per wave instruction and CU for ds_add_f64 / ds_add_f32 / ds_write_b64 / ds_read_b64,
// nine addresses per iteration like the cross block of a P1 pair.  Patterns: 0 every lane its own consecutive element (no bank
// conflict), 1 lane l -> row l of a sub-block with row stride 65 doubles (the tile kernels), 2 the same with pairs of lanes sharing
// the address (two cells with a common DoF), 3 all lanes one address.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip ; run: ./lds_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define NEL 8192
template <int MODE>
__global__ void __launch_bounds__(256) k_probe(double *out, int iters, int pattern) {
    __shared__ double s[NEL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < NEL; t += 256) s[t] = 0.;
    __syncthreads();
    int base;
    if (pattern == 0) base = wave*64+lane;
    else if (pattern == 1) base = (lane*65+wave) % (NEL-16);
    else if (pattern == 2) base = ((lane >> 1)*65+wave) % (NEL-16);
    else if (pattern == 3) base = wave;
    else base = (lane*(pattern-3)+wave) % (NEL-64);            // pattern 4 + k: lane stride k+1 doubles (bank structure)
    double v = 1.0+tid, acc = 0.;
    float vf = 1.f+tid;
    float *sf = (float*)s;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int a = base+k*7;
            if (MODE == 0) (void)__hip_atomic_fetch_add(&s[a], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 1) (void)__hip_atomic_fetch_add(&sf[a], vf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 2) ((volatile double*)s)[a] = v;
            if (MODE == 3) acc += ((volatile double*)s)[a];
            if (MODE == 4) { const double t = ((volatile double*)s)[a]; ((volatile double*)s)[a] = t+v; }
        }
        v += 1.0;
    }
    __syncthreads();
    out[blockIdx.x*256+tid] = acc+s[tid];
}

template <int MODE>
static void run(const char *name, double *out, int pattern) {
    const int iters = 20000, grid = 256;              // one workgroup per CU: the LDS pipe of a CU serves its four waves
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, pattern);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, pattern);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double clk = 2.4e9;                         // nominal
    const double per_instr = ms*1e-3*clk/((double)iters*9*4);     // 4 waves per CU share the pipe
    printf("%-14s pattern %d: %8.3f ms, %6.1f cycles per wave instruction and CU\n", name, pattern, ms, per_instr);
}

int main() {
    double *out;
    if (hipMalloc(&out, sizeof(double)*256*256) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    for (int p = 0; p < 4; p++) {
        run<0>("ds_add_f64", out, p);
        run<1>("ds_add_f32", out, p);
        run<2>("ds_write_b64", out, p);
        run<3>("ds_read_b64", out, p);
        run<4>("read+add+write", out, p);
    }
    // lane strides 1, 2, 4, 8, 16, 32, 64 doubles and 3, 5, 17, 33
    const int strides[] = {1, 2, 4, 8, 16, 32, 64, 3, 5, 17, 33};
    for (int k = 0; k < 11; k++) run<0>("ds_add_f64 stride", out, 3+strides[k]);
    hipFree(out);
    return 0;
}
