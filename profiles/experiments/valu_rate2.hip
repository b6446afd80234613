// valu_rate2.hip -- round-4 experiment: issue cost of the fp64 and conversion instructions the RRC-on-load path uses, next to f32
// ones, at four waves per SIMD (256 workgroups x 1024 threads, one per CU).  Each wave issues ITERS x 8 independent instructions of
// one kind; reported: cycles per instruction per SIMD at 2.4 GHz from the kernel time (s_memtime does not follow the shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP8(stmt) _Pragma("unroll") for (int i = 0; i < 8; ++i) { stmt; }

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters)
{
    float a[8]; double d[8]; unsigned u[8]; int n[8];
    const double kk = 1.000001 + threadIdx.x * 1e-9, bb = 0.25;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = i + threadIdx.x; d[i] = 3.5 + i + threadIdx.x; u[i] = i * 77 + threadIdx.x; n[i] = i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) OP8(a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f))
        if (MODE == 1) OP8(d[i] = __builtin_fma(d[i], kk, bb))
        if (MODE == 2) OP8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(kk)))
        if (MODE == 3) OP8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bb)))
        if (MODE == 4) OP8(asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i])))
        if (MODE == 5) OP8(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(d[i])))
        if (MODE == 6) OP8(asm volatile("v_trunc_f64 %0, %1" : "=v"(d[i]) : "v"(d[i])))
        if (MODE == 7) OP8(asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i])))
        if (MODE == 8) OP8(asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(a[i]) : "v"(u[i])))
        if (MODE == 9) OP8(asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(u[i]) : "v"(n[i])))
        if (MODE == 10) OP8(asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])))
        if (MODE == 11) OP8(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(n[i]) : "v"(a[i])))
        if (MODE == 12) OP8(asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(n[i]) : "v"(u[i])))
        if (MODE == 13) OP8(asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(d[i]) : "v"(kk)))
        if (MODE == 14) OP8(asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(u[i])))
        if (MODE == 15) OP8(asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(u[i]) : "v"(n[i])))
        if (MODE == 16) OP8(asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i])))
        if (MODE == 17) OP8(asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(n[i])))
        if (MODE == 18) OP8(asm volatile("v_cvt_pk_u16_u32 %0, %0, %1" : "+v"(u[i]) : "v"(n[i])))
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + (float)u[i] + (float)n[i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int M> static void run(const char *name, float *out)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<M>, dim3(256), dim3(1024), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double per_simd = (double)iters * 8 * 4;
    printf("%-22s %.3f ms  %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    float *out; hipMalloc((void **)&out, 4096);
    run<0>("v_fma_f32", out); run<1>("v_fma_f64", out); run<2>("v_mul_f64", out); run<3>("v_add_f64", out);
    run<4>("v_cvt_f64_u32", out); run<5>("v_cvt_i32_f64", out); run<6>("v_trunc_f64", out); run<7>("v_cvt_f32_u32", out);
    run<8>("v_cvt_f32_u32_sdwa", out); run<9>("v_alignbit_b32", out); run<10>("v_rndne_f32", out); run<11>("v_cvt_i32_f32", out);
    run<12>("v_med3_i32", out); run<13>("v_pk_mov_b32", out); run<14>("v_and_b32", out); run<15>("v_lshl_or_b32", out);
    run<16>("v_cvt_u32_f32", out); run<17>("v_mov_b32", out); run<18>("v_cvt_pk_u16_u32", out);
    return 0;
}
