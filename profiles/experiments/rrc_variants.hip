// Streaming-pattern experiment for the RRC kernel (not part of the product build).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off rrc_variants.hip -o rrc_variants && ./rrc_variants
// Variants over a 30000 x 65536 u16 raster (BASELINE config 2): plain copy for calibration, the
// shipped column-owned kernel, non-temporal accesses, rows in flight, block shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned rrc_px(double k, double b, unsigned s)
{
    double v = __dadd_rn(__dmul_rn(k, (double)s), b);
    int t = (v < 2147483648.0) ? (int)v : 0;
    return (unsigned)t & 0xffffu;
}
__device__ __forceinline__ uint4 rrc_vec(uint4 in, const double *k, const double *b)
{
    unsigned w[4];
    const unsigned *p = reinterpret_cast<const unsigned *>(&in);
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = rrc_px(k[2 * i], b[2 * i], p[i] & 0xffffu) | (rrc_px(k[2 * i + 1], b[2 * i + 1], p[i] >> 16) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld(const uint4 *p)
{
    if (!NT) return *p;
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool NT> __device__ __forceinline__ void st(uint4 *p, uint4 v)
{
    if (!NT) { *p = v; return; }
    u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4 *>(p));
}

// plain flat copy, grid-stride, U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_flat(const uint4 *src, uint4 *dst, long n)
{
    long i = (long)blockIdx.x * 256 * U + threadIdx.x;
    const long stride = (long)gridDim.x * 256 * U;
    for (; i + 256 * (U - 1) < n; i += stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(dst + i + u * 256, v[u]);
    }
}

// column-owned RRC (the shipped design): ROWS rows in flight, optional math (COMPUTE) and nt
template <int ROWS, bool NT, bool COMPUTE>
__global__ __launch_bounds__(256) void rrc_cols(const uint16_t *src, uint16_t *dst, int w, long h, const double2 *kb, long rpb)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (x0 >= w) return;
    double k[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { double2 p = kb[x0 + i]; k[i] = p.x; b[i] = p.y; }
    const long r0 = (long)blockIdx.y * rpb;
    long r1 = r0 + rpb; if (r1 > h) r1 = h;
    const uint16_t *s = src + r0 * (long)w + x0;
    uint16_t *d = dst + r0 * (long)w + x0;
    for (long r = r0; r + ROWS <= r1; r += ROWS) {
        uint4 v[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) v[u] = ld<NT>(reinterpret_cast<const uint4 *>(s + (long)u * w));
#pragma unroll
        for (int u = 0; u < ROWS; ++u) st<NT>(reinterpret_cast<uint4 *>(d + (long)u * w), COMPUTE ? rrc_vec(v[u], k, b) : v[u]);
        s += (long)ROWS * w; d += (long)ROWS * w;
    }
}

// row-interleaved blocks: block (bx, by) handles rows by, by+gridDim.y, ... (all blocks sweep the
// raster front to back together, so concurrently active rows are neighbours in memory)
template <int ROWS, bool NT>
__global__ __launch_bounds__(256) void rrc_sweep(const uint16_t *src, uint16_t *dst, int w, long h, const double2 *kb)
{
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (x0 >= w) return;
    double k[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { double2 p = kb[x0 + i]; k[i] = p.x; b[i] = p.y; }
    const long step = (long)gridDim.y * ROWS;
    for (long r = (long)blockIdx.y * ROWS; r + ROWS <= h; r += step) {
        const uint16_t *s = src + r * (long)w + x0;
        uint16_t *d = dst + r * (long)w + x0;
        uint4 v[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; ++u) v[u] = ld<NT>(reinterpret_cast<const uint4 *>(s + (long)u * w));
#pragma unroll
        for (int u = 0; u < ROWS; ++u) st<NT>(reinterpret_cast<uint4 *>(d + (long)u * w), rrc_vec(v[u], k, b));
    }
}

// flat, line-aligned, column-fixed: a wave always covers 64 consecutive 16-byte chunks starting
// at a multiple of 64 chunks (1 KiB aligned), and steps by `sr` = lcm(chunks per row, 64) chunks
// (a whole number of rows), so its lanes keep their columns -- LUT in registers -- while every
// access is a full, aligned 1 KiB.
template <int R, bool NT, bool SNT = NT>
__global__ __launch_bounds__(256) void rrc_flat(const uint16_t *src, uint16_t *dst, int P /*chunks per row*/, long nchunks,
                                                long sr /*chunks per super-row*/, const double2 *kb, long nsuper, long super_per_block)
{
    const long f0 = ((long)blockIdx.x * 256 + threadIdx.x);     // chunk inside the first super-row
    if (f0 >= sr) return;
    const int col = (int)(f0 % P) * 8;
    double k[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { double2 p = kb[col + i]; k[i] = p.x; b[i] = p.y; }
    const long s0 = (long)blockIdx.y * super_per_block;
    long s1 = s0 + super_per_block; if (s1 > nsuper) s1 = nsuper;
    const uint4 *s = reinterpret_cast<const uint4 *>(src) + f0 + s0 * sr;
    uint4 *d = reinterpret_cast<uint4 *>(dst) + f0 + s0 * sr;
    long f = f0 + s0 * sr;
    for (long q = s0; q < s1; q += R) {
        uint4 v[R];
#pragma unroll
        for (int u = 0; u < R; ++u) if (q + u < s1 && f + u * sr < nchunks) v[u] = ld<NT>(s + u * sr);
#pragma unroll
        for (int u = 0; u < R; ++u) if (q + u < s1 && f + u * sr < nchunks) st<SNT>(d + u * sr, rrc_vec(v[u], k, b));
        s += R * sr; d += R * sr; f += R * sr;
    }
}

template <typename F> float timeit(F f, int iters = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / iters;
}

int main()
{
    const int w = 30000; const long h = 65536;
    const long npx = (long)w * h; const size_t bytes = npx * 2;
    uint16_t *src, *dst; double2 *kb;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&kb, sizeof(double2) * w));
    std::vector<uint16_t> hs(1 << 20); for (auto &v : hs) v = rand() & 4095;
    for (size_t o = 0; o < bytes; o += hs.size() * 2) CK(hipMemcpy((char *)src + o, hs.data(), std::min(hs.size() * 2, bytes - o), hipMemcpyHostToDevice));
    std::vector<double2> hk(w); for (int i = 0; i < w; ++i) hk[i] = make_double2(0.9 + 0.2 * (rand() / (double)RAND_MAX), 8.0 * (rand() / (double)RAND_MAX));
    CK(hipMemcpy(kb, hk.data(), sizeof(double2) * w, hipMemcpyHostToDevice));
    auto report = [&](const char *name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s (rd+wr)\n", name, ms, 2.0 * bytes / ms / 1e6); fflush(stdout); };
    const long n16 = bytes / 16;
    for (int blocks : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "copy_flat U4 grid %d", blocks); report(nm, timeit([&] { copy_flat<4, false><<<blocks, 256>>>((uint4 *)src, (uint4 *)dst, n16); }));
        snprintf(nm, 64, "copy_flat U4 nt grid %d", blocks); report(nm, timeit([&] { copy_flat<4, true><<<blocks, 256>>>((uint4 *)src, (uint4 *)dst, n16); }));
        snprintf(nm, 64, "copy_flat U8 nt grid %d", blocks); report(nm, timeit([&] { copy_flat<8, true><<<blocks, 256>>>((uint4 *)src, (uint4 *)dst, n16); }));
    }
    const int gx = (w / 8 + 255) / 256;
    for (int gy : {136, 2048}) {
        long rpb = (h + gy - 1) / gy; rpb = (rpb + 7) / 8 * 8; int g = (int)((h + rpb - 1) / rpb);
        char nm[64];
        snprintf(nm, 64, "rrc_cols R4 gy %d", g); report(nm, timeit([&] { rrc_cols<4, false, true><<<dim3(gx, g), 256>>>(src, dst, w, h, kb, rpb); }));
        snprintf(nm, 64, "rrc_cols R4 nt gy %d", g); report(nm, timeit([&] { rrc_cols<4, true, true><<<dim3(gx, g), 256>>>(src, dst, w, h, kb, rpb); }));
        snprintf(nm, 64, "rrc_cols R8 nt gy %d", g); report(nm, timeit([&] { rrc_cols<8, true, true><<<dim3(gx, g), 256>>>(src, dst, w, h, kb, rpb); }));
        snprintf(nm, 64, "rrc_cols R4 nt nocompute gy %d", g); report(nm, timeit([&] { rrc_cols<4, true, false><<<dim3(gx, g), 256>>>(src, dst, w, h, kb, rpb); }));
    }
    for (int gy : {512}) {
        char nm[64];
        snprintf(nm, 64, "rrc_sweep R4 gy %d", gy); report(nm, timeit([&] { rrc_sweep<4, false><<<dim3(gx, gy), 256>>>(src, dst, w, h, kb); }));
        snprintf(nm, 64, "rrc_sweep R4 nt gy %d", gy); report(nm, timeit([&] { rrc_sweep<4, true><<<dim3(gx, gy), 256>>>(src, dst, w, h, kb); }));
        snprintf(nm, 64, "rrc_sweep R2 nt gy %d", gy); report(nm, timeit([&] { rrc_sweep<2, true><<<dim3(gx, gy), 256>>>(src, dst, w, h, kb); }));
        snprintf(nm, 64, "rrc_sweep R8 nt gy %d", gy); report(nm, timeit([&] { rrc_sweep<8, true><<<dim3(gx, gy), 256>>>(src, dst, w, h, kb); }));
    }
    // misaligned flat copy: does losing 128-byte line alignment explain the gap?
    report("copy_flat U4 nt grid 32768 +96B misaligned", timeit([&] { copy_flat<4, true><<<32768, 256>>>((uint4 *)src + 6, (uint4 *)dst + 6, n16 - 8); }));
    report("copy_flat U4 nt grid 32768 +32B misaligned", timeit([&] { copy_flat<4, true><<<32768, 256>>>((uint4 *)src + 2, (uint4 *)dst + 2, n16 - 8); }));
    report("copy_flat U4 nt grid 32768 src only +96B", timeit([&] { copy_flat<4, true><<<32768, 256>>>((uint4 *)src + 6, (uint4 *)dst, n16 - 8); }));
    report("copy_flat U4 nt grid 32768 dst only +96B", timeit([&] { copy_flat<4, true><<<32768, 256>>>((uint4 *)src, (uint4 *)dst + 6, n16 - 8); }));
    {
        const int P = w / 8;
        long g = P, t = 64; while (t) { long r = g % t; g = t; t = r; }
        const long sr = (long)P / g * 64;                 // lcm(P, 64)
        const long nch = npx / 8;
        const long nsuper = (nch + sr - 1) / sr;
        const int gx2 = (int)((sr + 255) / 256);
        for (int gy : {64, 128, 256, 512, 1024}) {
            long spb = (nsuper + gy - 1) / gy;
            char nm[64];
            snprintf(nm, 64, "rrc_flat R4 gy %d", gy); report(nm, timeit([&] { rrc_flat<4, false><<<dim3(gx2, gy), 256>>>(src, dst, P, nch, sr, kb, nsuper, spb); }));
            snprintf(nm, 64, "rrc_flat R2 gy %d", gy); report(nm, timeit([&] { rrc_flat<2, false><<<dim3(gx2, gy), 256>>>(src, dst, P, nch, sr, kb, nsuper, spb); }));
            snprintf(nm, 64, "rrc_flat R1 gy %d", gy); report(nm, timeit([&] { rrc_flat<1, false><<<dim3(gx2, gy), 256>>>(src, dst, P, nch, sr, kb, nsuper, spb); }));
            snprintf(nm, 64, "rrc_flat R4 ntload gy %d", gy); report(nm, timeit([&] { rrc_flat<4, true, false><<<dim3(gx2, gy), 256>>>(src, dst, P, nch, sr, kb, nsuper, spb); }));
            snprintf(nm, 64, "rrc_flat R4 ntstore gy %d", gy); report(nm, timeit([&] { rrc_flat<4, false, true><<<dim3(gx2, gy), 256>>>(src, dst, P, nch, sr, kb, nsuper, spb); }));
            snprintf(nm, 64, "rrc_flat R4 in place gy %d", gy); report(nm, timeit([&] { rrc_flat<4, false><<<dim3(gx2, gy), 256>>>(src, src, P, nch, sr, kb, nsuper, spb); }));
        }
    }
    // in-place (the reference seam): half the footprint
    report("rrc_sweep R4 nt gy 256 in place", timeit([&] { rrc_sweep<4, true><<<dim3(gx, 256), 256>>>(src, src, w, h, kb); }));
    report("rrc_cols R4 gy 136 in place", timeit([&] { rrc_cols<4, false, true><<<dim3(gx, 136), 256>>>(src, src, w, h, kb, 488); }));
    return 0;
}
