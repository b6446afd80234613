// valu_rate.hip -- round-3 experiment: how many cycles does a SIMD of gfx950 need per wave64 vector instruction when several
// waves share it?  256 workgroups x W waves per SIMD, every wave issues N independent v_fma_f32 (or v_mul + v_add, or
// v_pk_fma_f32 / v_pk_mul_f32 on register pairs); reported: cycles per instruction per SIMD from the kernel time at the clock the
// kernel itself measures with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, long long *cyc)
{
    float a[8], b = 1.0001f + threadIdx.x * 1e-7f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = i + threadIdx.x; p[i] = f2{a[i], a[i] + 1}; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) a[i] = __builtin_fmaf(a[i], b, c);
            else if (MODE == 1) { a[i] = a[i] * b; asm volatile("" : "+v"(a[i])); a[i] = a[i] + c; }
            else if (MODE == 2) p[i] = __builtin_elementwise_fma(p[i], pb, pc);
            else { p[i] = p[i] * pb; asm volatile("" : "+v"(p[i])); p[i] = p[i] + pc; }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc((void **)&out, 4096); hipMalloc((void **)&cyc, 8);
    const int iters = 20000;
    const char *names[4] = {"v_fma_f32", "v_mul_f32 + v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32 + v_pk_add_f32"};
    for (int mode = 0; mode < 4; ++mode)
        for (int waves : {1, 2, 4, 8}) {          // waves per SIMD = threads / 256
            const int threads = waves * 256;
            if (threads > 1024) { continue; }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                else hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double instr_per_wave = (double)iters * 8 * (mode == 1 || mode == 3 ? 2 : 1);
            printf("%-30s %d wave(s)/SIMD: %.2f cycles per instruction per SIMD (wave lifetime %lld ticks, %.3f ms)\n", names[mode], waves,
                   (double)c / (instr_per_wave * waves), c, ms);
        }
    return 0;
}
