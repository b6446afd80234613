// Pass-count experiment for the column transform (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 single_pass_column.hip -o single_pass_column && ./single_pass_column
// The shipped column transform of a 16000 x 3000 complex array is two HBM passes (128 x 125).  A single-pass
// transform keeps a whole 16000-point column (128 KB) in one workgroup's LDS, so every access to the row-major
// array is 8 bytes at a 24 KB stride.  This probe times exactly that access pattern with the transform left
// out (column -> LDS -> column), so that the number is a lower bound for any single-pass kernel:
//   copy      read a column into LDS, write it back to a second array
//   read      read only (the last inverse pass stores nothing)
//   write     write only
// in two workgroup -> column orders: "xcd" gives the 32 workgroups resident on one XCD 32 neighbouring columns
// (the 16 columns of a 128-byte line are in flight on one L2 together), "plain" numbers them in launch order.
// The two-pass form's own traffic moves the array twice in 128-byte segments; its times are in DESIGN.md §5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kRows = 16000, kCols = 3000, kThreads = 1024;

template <int MODE, bool XCD>      // MODE 0 copy, 1 read, 2 write
__global__ __launch_bounds__(kThreads) void column_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, float *sink)
{
    extern __shared__ float2 col[];
    int id = blockIdx.x;
    int c = XCD ? (id & 7) * (kCols / 8) + (id >> 3) : id;
    if (MODE != 2) {
        for (int r = threadIdx.x; r < kRows; r += kThreads) col[r] = in[(long)r * kCols + c];
    } else {
        for (int r = threadIdx.x; r < kRows; r += kThreads) col[r] = make_float2((float)r, (float)c);
    }
    __syncthreads();
    if (MODE != 1) {
        // a transform leaves its output in digit-reversed LDS order; the read below is the unit-stride case
        for (int r = threadIdx.x; r < kRows; r += kThreads) out[(long)r * kCols + c] = col[r];
    } else {
        float acc = 0.f;
        for (int r = threadIdx.x; r < kRows; r += kThreads) acc += col[(r * 7) % kRows].x;
        if (acc == 1.2345f) *sink = acc;
    }
}

// the same bytes in 128-byte segments, as one of the two passes of the shipped form moves them (16 columns x
// 128 rows per workgroup, rows 125 apart): calibration of what "one pass" costs on this box
__global__ __launch_bounds__(256) void tile_kernel(const float2 *__restrict__ in, float2 *__restrict__ out)
{
    __shared__ float2 t[128 * 17];
    int tiles_x = (kCols + 15) / 16;
    int o1 = blockIdx.x / tiles_x, cx = (blockIdx.x % tiles_x) * 16;
    int lane = threadIdx.x & 15, j0 = threadIdx.x >> 4;
    for (int j = j0; j < 128; j += 16)
        if (cx + lane < kCols) t[j * 17 + lane] = in[(long)(o1 + 125 * j) * kCols + cx + lane];
    __syncthreads();
    for (int j = j0; j < 128; j += 16)
        if (cx + lane < kCols) out[(long)(o1 + 125 * j) * kCols + cx + lane] = t[((j * 5) & 127) * 17 + lane];
}

template <typename F> static float time_it(F launch, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char **argv)
{
    int reps = argc > 1 ? atoi(argv[1]) : 10;
    size_t bytes = (size_t)kRows * kCols * sizeof(float2);
    float2 *in, *out; float *sink;
    CK(hipMalloc(&in, bytes)); CK(hipMalloc(&out, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 0, bytes)); CK(hipMemset(out, 0, bytes));
    size_t lds = (size_t)kRows * sizeof(float2);
#define ATTR(k) CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
    ATTR((column_kernel<0, true>)); ATTR((column_kernel<0, false>)); ATTR((column_kernel<1, true>));
    ATTR((column_kernel<1, false>)); ATTR((column_kernel<2, true>)); ATTR((column_kernel<2, false>));
    double gb = bytes / 1e9;
#define RUN(name, k, moved) { float ms = time_it([&] { hipLaunchKernelGGL(k, dim3(kCols), dim3(kThreads), lds, 0, in, out, sink); }, reps); \
        printf("%-26s %.4f ms  %.2f TB/s of array bytes\n", name, ms, moved * gb / ms); }
    RUN("single-pass copy  xcd", (column_kernel<0, true>), 2);
    RUN("single-pass copy  plain", (column_kernel<0, false>), 2);
    RUN("single-pass read  xcd", (column_kernel<1, true>), 1);
    RUN("single-pass read  plain", (column_kernel<1, false>), 1);
    RUN("single-pass write xcd", (column_kernel<2, true>), 1);
    RUN("single-pass write plain", (column_kernel<2, false>), 1);
    {
        int grid = 125 * ((kCols + 15) / 16);
        float ms = time_it([&] { hipLaunchKernelGGL(tile_kernel, dim3(grid), dim3(256), 0, 0, in, out); }, reps);
        printf("%-26s %.4f ms  %.2f TB/s of array bytes\n", "one 128-B-segment pass", ms, 2 * gb / ms);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
