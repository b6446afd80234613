// strided_rows_read.hip -- round-3 experiment: what bounds the last inverse column pass (fft_col128_peak_kernel, 3.7 TB/s)?
// The pass reads, per workgroup, 128 row segments of 128 bytes that are 125 rows (3 MB) apart; the 125-point pass reads
// 125 CONSECUTIVE rows (24 KB apart) and runs at the copy ceiling (5.8 TB/s).  Column panels (Infinity Cache residency) and
// 256-byte segments changed nothing (profiles/experiments/r03_sweep1_panels_lanes.txt, r03_sweep2).  This program times the
// bare access pattern -- 16 lanes x 128 rows per 256-thread workgroup, 8 loads of 8 bytes per thread, nothing computed --
// for three row layouts of the same 16000 x 3008 float2 array:
//   a  row = o + 125 n                       the pass as it is (n: point index 0..127, o: offset 0..124)
//   b  row = ((n >> 3) 125 + o) 8 + (n & 7)  a two-level layout: 16 groups of 8 consecutive rows
//   c  row = 128 o + n                       128 consecutive rows (the friendly pattern)
// and the same with the loads of one thread issued for 16-byte (float4, 8 lanes per row) accesses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int M = 16000, P = 3008, LT = P / 16;

template <int MODE>
__global__ __launch_bounds__(256) void read_kernel(const float2 *__restrict__ data, float *__restrict__ sink)
{
    const int lt = blockIdx.x, o = blockIdx.y;
    const int q = threadIdx.x >> 4, v = threadIdx.x & 15;
    float2 x[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int n = q + 16 * m;
        long row;
        if (MODE == 0) row = o + 125L * n;
        else if (MODE == 1) row = ((long)(n >> 3) * 125 + o) * 8 + (n & 7);
        else row = 128L * o + n;
        x[m] = data[row * P + lt * 16 + v];
    }
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m) s += x[m].x + x[m].y;
    if (s == 12345.678f) sink[blockIdx.x] = s;         // never true: keeps the loads
}

int main()
{
    float2 *d;
    float *sink;
    const size_t n = (size_t)M * P;
    hipMalloc((void **)&d, n * sizeof(float2));
    hipMalloc((void **)&sink, 4096);
    hipMemset(d, 0, n * sizeof(float2));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"a  rows o + 125 n (3 MB apart)", "b  16 groups of 8 consecutive rows", "c  128 consecutive rows"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(read_kernel<0>, dim3(LT, 125), dim3(256), 0, 0, d, sink);
            else if (mode == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(LT, 125), dim3(256), 0, 0, d, sink);
            else hipLaunchKernelGGL(read_kernel<2>, dim3(LT, 125), dim3(256), 0, 0, d, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) best = ms < best ? ms : best;
        }
        printf("%-40s %.4f ms  %.2f TB/s\n", names[mode], best, (double)LT * 125 * 128 * 128 / (best * 1e-3) / 1e12);
    }
    return 0;
}
