// cu_mask_probe.hip -- round-3 experiment: can the compute-bound row stage and the HBM-bound column passes of neighbouring
// unit pairs run SIDE BY SIDE on disjoint sets of CUs (hipExtStreamCreateWithCUMask)?
//   1. which CUs does a mask select: a kernel records (XCC_ID, SE_ID, CU_ID) of its workgroups per mask pattern
//   2. the strided read of the last inverse pass (strided_rows_read.hip, pattern a) on n of 256 CUs: TB/s against n
//   3. a compute-only kernel (fma chain, fixed work per workgroup, persistent over a work list) on the complementary CUs:
//      alone, and beside the read kernel -- do the two keep their stand-alone times?
// Build on the box:  hipcc --offload-arch=gfx950 -O3 profiles/experiments/cu_mask_probe.hip -o /tmp/cu_mask_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <set>
#include <vector>

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e = (x);                                                                        \
        if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); }     \
    } while (0)

constexpr int M = 16000, P = 3008, LT = P / 16;

__global__ __launch_bounds__(256) void read_kernel(const float2 *__restrict__ data, float *__restrict__ sink, int tiles)
{
    // persistent over the (lane tile, o) list so that the grid can be sized to the CUs of the stream
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int lt = t % LT, o = t / LT;
        const int q = threadIdx.x >> 4, v = threadIdx.x & 15;
        float2 x[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = data[(o + 125L * (q + 16 * m)) * P + lt * 16 + v];
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) s += x[m].x + x[m].y;
        if (s == 12345.678f) sink[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(512) void spin_kernel(float *__restrict__ sink, int items, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int t = blockIdx.x; t < items; t += gridDim.x)
        for (int i = 0; i < iters; ++i) {
            a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.999f, 1e-4f);
        }
    if (a + c + d == 12345.678f) sink[blockIdx.x] = a;
}

__global__ void where_kernel(unsigned *__restrict__ out)
{
    if (threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x] = (hwid & 0xffffu) | ((xcc & 0xfu) << 16);       // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    }
    // stay a little so that the workgroups spread
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

static float time_ms(hipStream_t s, const std::function<void()> &f)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s));
        f();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = ms < best ? ms : best;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("CUs %d\n", ncu);
    const int words = (ncu + 31) / 32;
    float2 *d;
    float *sink;
    unsigned *where;
    const size_t n = (size_t)M * P;
    CK(hipMalloc((void **)&d, n * sizeof(float2)));
    CK(hipMalloc((void **)&sink, 1 << 16));
    CK(hipMalloc((void **)&where, 4096 * 4));
    CK(hipMemset(d, 0, n * sizeof(float2)));

    auto make_stream = [&](const std::vector<uint32_t> &mask) {
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
        return s;
    };
    auto mask_first = [&](int k) {           // bits 0 .. k-1
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < k; ++i) m[i / 32] |= 1u << (i % 32);
        return m;
    };
    auto mask_not = [&](std::vector<uint32_t> m) {
        for (int i = 0; i < ncu; ++i) m[i / 32] ^= 1u << (i % 32);
        return m;
    };
    auto mask_mod = [&](int period, int keep) {       // bit i set when i % period < keep
        std::vector<uint32_t> m(words, 0u);
        for (int i = 0; i < ncu; ++i)
            if (i % period < keep) m[i / 32] |= 1u << (i % 32);
        return m;
    };

    // 1. where do the masks land
    auto census = [&](const char *name, const std::vector<uint32_t> &mask) {
        hipStream_t s = make_stream(mask);
        CK(hipMemsetAsync(where, 0xff, 4096 * 4, s));
        hipLaunchKernelGGL(where_kernel, dim3(2048), dim3(64), 0, s, where);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned> h(2048);
        CK(hipMemcpy(h.data(), where, 2048 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus;
        int per_xcc[16] = {0};
        for (unsigned w : h) {
            const unsigned key = (w >> 8) & 0xfffu;       // xcc | se | sh | cu
            if (cus.insert(key).second) per_xcc[(w >> 16) & 0xf]++;
        }
        printf("%-34s -> %3zu distinct CUs; per XCC:", name, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
        printf("\n");
        CK(hipStreamDestroy(s));
    };
    census("all bits", mask_first(ncu));
    census("bits 0..127", mask_first(128));
    census("bits 0..63", mask_first(64));
    census("bits i % 8 < 5", mask_mod(8, 5));
    census("bits i % 32 < 20", mask_mod(32, 20));
    census("bits i % 4 < 3", mask_mod(4, 3));

    // 2. the strided read on k CUs
    const int tiles = LT * 125;
    for (int k : {256, 192, 160, 128, 96, 64}) {
        if (k > ncu) continue;
        hipStream_t s = make_stream(mask_first(k));
        const float ms = time_ms(s, [&] { hipLaunchKernelGGL(read_kernel, dim3(k * 8), dim3(256), 0, s, d, sink, tiles); });
        printf("strided read on bits 0..%d: %.4f ms  %.2f TB/s\n", k - 1, ms, (double)tiles * 128 * 128 / (ms * 1e-3) / 1e12);
        CK(hipStreamDestroy(s));
    }
    for (int keep : {24, 20, 16, 12}) {
        hipStream_t s = make_stream(mask_mod(32, keep));
        const int k = ncu * keep / 32;
        const float ms = time_ms(s, [&] { hipLaunchKernelGGL(read_kernel, dim3(k * 8), dim3(256), 0, s, d, sink, tiles); });
        printf("strided read on bits i %% 32 < %d (%d CUs): %.4f ms  %.2f TB/s\n", keep, k, ms, (double)tiles * 128 * 128 / (ms * 1e-3) / 1e12);
        CK(hipStreamDestroy(s));
    }

    // 3. compute beside memory on complementary masks
    for (int keep : {20, 16}) {
        const std::vector<uint32_t> mc = mask_mod(32, keep), mm = mask_not(mc);
        hipStream_t sc = make_stream(mc), sm = make_stream(mm);
        const int kc = ncu * keep / 32, km = ncu - kc;
        const int items = 4096, iters = 20000;
        const float tc = time_ms(sc, [&] { hipLaunchKernelGGL(spin_kernel, dim3(kc), dim3(512), 0, sc, sink, items, iters); });
        const int reps = 12;
        const float tm = time_ms(sm, [&] { for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel, dim3(km * 8), dim3(256), 0, sm, d, sink, tiles); });
        // both: start together, wait for both
        hipEvent_t e0, e1, e2;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sc));
        CK(hipStreamWaitEvent(sm, e0, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(kc), dim3(512), 0, sc, sink, items, iters);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel, dim3(km * 8), dim3(256), 0, sm, d, sink, tiles);
        CK(hipEventRecord(e1, sc));
        CK(hipEventRecord(e2, sm));
        CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(e2));
        float a, b;
        CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e0, e2));
        printf("compute on %d CUs alone %.3f ms | %d reads on %d CUs alone %.3f ms | side by side: compute %.3f ms, reads %.3f ms\n", kc, tc, reps, km, tm, a, b);
        CK(hipStreamDestroy(sc)); CK(hipStreamDestroy(sm));
    }
    return 0;
}
