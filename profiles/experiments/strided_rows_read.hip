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

// MAP 1: the product's XCD-chunked lane-tile numbering (workgroup i runs on XCD i % 8; XCD j owns a contiguous chunk of lane tiles)
// LDSB > 0: the workgroup also holds LDSB bytes of LDS and passes a barrier, like the pass it imitates
template <int MODE, int MAP = 0, int LDSB = 0>
__global__ __launch_bounds__(256) void read_kernel(const float2 *__restrict__ data, float *__restrict__ sink)
{
    __shared__ float lds[LDSB > 0 ? LDSB / 4 : 1];
    constexpr int chunk = (LT + 7) / 8;
    const int lt = MAP ? (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int o = blockIdx.y;
    if (lt >= LT) return;
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
    if (LDSB > 0) {
        lds[threadIdx.x] = s;
        __syncthreads();
        s += lds[threadIdx.x ^ 64];
    }
    if (s == 12345.678f) sink[blockIdx.x] = s;         // never true: keeps the loads
}


// Step-by-step towards the product's pass (pattern a, XCD-chunked, the tile buffer in LDS):
//   STEP 1  + the inter-pass twiddle gather  twT[o * tid]  (128 threads, stride 8 o bytes in a 128 KB table) -> LDS, barrier
//   STEP 2  + the first stage's data flow: x[m] *= twj[q + 16 m] (LDS reads), 8 stores of 8 bytes into the padded tile, barrier
//   STEP 3  + a second barrier-separated LDS round trip (what stages 2 and 3 do), nothing computed
// VEC4: the tile's data through four 16-byte loads per thread (two neighbouring columns of rows q + 16 (2 j + h), h = v >> 3)
// instead of eight 8-byte loads -- half the vector-memory instructions for the same bytes; TWROW: the twiddles as one
// contiguous row (twT + 128 o) instead of the gather
template <int STEP, bool VEC4 = false, bool TWROW = false>
__global__ __launch_bounds__(256) void pass_like_kernel(const float2 *__restrict__ data, const float2 *__restrict__ twT, float *__restrict__ sink)
{
    __shared__ float2 buf[128 * 17];
    __shared__ float2 twj[128];
    constexpr int chunk = (LT + 7) / 8;
    const int lt = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    const int o = blockIdx.y;
    if (lt >= LT) return;
    const int q = threadIdx.x >> 4, v = threadIdx.x & 15;
    float2 x[8];
    if (VEC4) {
        const int h = v >> 3, c = 2 * (v & 7);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *reinterpret_cast<const float4 *>(data + (o + 125L * (q + 16 * (2 * j + h))) * P + lt * 16 + c);
            x[2 * j] = make_float2(t.x, t.y);
            x[2 * j + 1] = make_float2(t.z, t.w);
        }
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = data[(o + 125L * (q + 16 * m)) * P + lt * 16 + v];
    }
    float2 rtw = make_float2(1.f, 0.f);
    if (threadIdx.x < 128) rtw = TWROW ? twT[(o * 128 + (int)threadIdx.x) % 16000] : twT[o * (int)threadIdx.x];
    if (threadIdx.x < 128) twj[threadIdx.x] = rtw;
    __syncthreads();
    float s = 0.f;
    if (STEP >= 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const float2 w = twj[q + 16 * m];
            x[m] = make_float2(x[m].x * w.x + x[m].y * w.y, x[m].x * w.y - x[m].y * w.x);
            buf[(q * 8 + m) * 17 + v] = x[m];
        }
        __syncthreads();
        if (STEP >= 3) {
            float2 y[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) y[m] = buf[(q + 16 * m) * 17 + v];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < 8; ++m) buf[(q * 8 + m) * 17 + v] = make_float2(y[m].y, y[m].x);
            __syncthreads();
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) { const float2 y = buf[(q + 16 * m) * 17 + v]; s += y.x + y.y; }
    } else {
#pragma unroll
        for (int m = 0; m < 8; ++m) s += x[m].x + x[m].y + twj[(q + 16 * m) & 127].x;
    }
    if (s == 12345.678f) sink[blockIdx.x] = s;
}

template <typename K>
static void time_kernel(const char *name, K launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) best = ms < best ? ms : best;
    }
    printf("%-64s %.4f ms  %.2f TB/s\n", name, best, (double)LT * 125 * 128 * 128 / (best * 1e-3) / 1e12);
}

int main(int argc, char **)
{
    float2 *d;
    float *sink;
    const size_t n = (size_t)M * P;
    hipMalloc((void **)&d, n * sizeof(float2));
    hipMalloc((void **)&sink, 4096);
    hipMemset(d, 0, n * sizeof(float2));
    if (argc > 1) {                       // any argument: random data instead of zeros
        std::vector<float2> h(n);
        unsigned r = 12345u;
        for (auto &z : h) { r = r * 1664525u + 1013904223u; z.x = (float)(r >> 8) * 1e-3f; r = r * 1664525u + 1013904223u; z.y = -(float)(r >> 8) * 1e-3f; }
        hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice);
        printf("random data\n");
    }
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
    constexpr int GX = 8 * ((LT + 7) / 8);
    time_kernel("a, XCD-chunked lane tiles (the product's numbering)", [&] { hipLaunchKernelGGL((read_kernel<0, 1, 0>), dim3(GX, 125), dim3(256), 0, 0, d, sink); });
    time_kernel("a, plain numbering, 18 KB of LDS + a barrier per workgroup", [&] { hipLaunchKernelGGL((read_kernel<0, 0, 18432>), dim3(LT, 125), dim3(256), 0, 0, d, sink); });
    time_kernel("a, XCD-chunked, 18 KB of LDS + a barrier", [&] { hipLaunchKernelGGL((read_kernel<0, 1, 18432>), dim3(GX, 125), dim3(256), 0, 0, d, sink); });
    time_kernel("c, XCD-chunked", [&] { hipLaunchKernelGGL((read_kernel<2, 1, 0>), dim3(GX, 125), dim3(256), 0, 0, d, sink); });
    float2 *tw;
    hipMalloc((void **)&tw, 16000 * sizeof(float2));
    hipMemset(tw, 0, 16000 * sizeof(float2));
    time_kernel("pass-like 1: + twiddle gather -> LDS, barrier", [&] { hipLaunchKernelGGL((pass_like_kernel<1>), dim3(GX, 125), dim3(256), 0, 0, d, tw, sink); });
    time_kernel("pass-like 2: + first-stage data flow (LDS tile), barrier", [&] { hipLaunchKernelGGL((pass_like_kernel<2>), dim3(GX, 125), dim3(256), 0, 0, d, tw, sink); });
    time_kernel("pass-like 3: + one more barrier-separated LDS round trip", [&] { hipLaunchKernelGGL((pass_like_kernel<3>), dim3(GX, 125), dim3(256), 0, 0, d, tw, sink); });
    time_kernel("pass-like 3, twiddles as a contiguous row", [&] { hipLaunchKernelGGL((pass_like_kernel<3, false, true>), dim3(GX, 125), dim3(256), 0, 0, d, tw, sink); });
    time_kernel("pass-like 3, contiguous row, 16-byte data loads", [&] { hipLaunchKernelGGL((pass_like_kernel<3, true, true>), dim3(GX, 125), dim3(256), 0, 0, d, tw, sink); });
    time_kernel("pass-like 3, gather, 16-byte data loads", [&] { hipLaunchKernelGGL((pass_like_kernel<3, true, false>), dim3(GX, 125), dim3(256), 0, 0, d, tw, sink); });
    return 0;
}
