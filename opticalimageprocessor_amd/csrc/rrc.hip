// rrc.hip -- relative radiometric correction on gfx950.
//
// Replaces IMO::InplaceRRC (imageop.h:129-138) and, fused with the BIL split of
// PreProcessor::LoadMSS (preproc.h:62-75), PreProcessor::DoRRC4MSS (preproc.h:202-222).
//
//   dst = (uint16_t)(k[x] * src + b[x])            -- fp64, two roundings, no FMA
//
// Layout / mapping.  HBM-bound: 2 B read + 2 B written per pixel.  A lane owns 8 consecutive
// pixels (one 16-byte chunk) and keeps their 8 (k,b) pairs in VGPRs; no LDS (no reuse beyond
// the LUT).  What decides the bandwidth is line alignment of the STORES: a 30000-pixel line is
// 60000 B, not a multiple of 128 B, so a wave that follows line boundaries writes partial cache
// lines (measured: 4.4-4.7 TB/s, profiles/experiments/rrc_variants.hip).  The main kernel
// therefore treats the raster as a flat array of 16-byte chunks: a wave always covers 64
// consecutive chunks starting on a 1 KiB boundary of the destination, and a lane advances by
// lcm(chunks per line, 64) chunks -- a whole number of lines -- so its 8 columns, hence its
// LUT registers, never change while every access is a full aligned KiB (measured 5.7 TB/s =
// 72 % of the 8 TB/s HBM3E peak on 30000 x 65536).  Widths that are not a multiple of 8 fall
// back to the column-owned 8-byte kernel, anything else to a scalar kernel.
//
// Conversion.  The reference's double->uint16_t cast is what x86-64 compilers emit for it:
// cvttsd2si (32-bit, truncating; 0x80000000 when out of range or NaN) followed by a 16-bit
// truncation.  v_cvt_i32_f64 truncates the same way but saturates, so only the positive
// overflow case needs a select (negative overflow and NaN already give low half 0).
#include "oip_internal.h"

#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kRowsInFlight = 4;

__device__ __forceinline__ unsigned rrc_px(double k, double b, unsigned s) { return oip_rrc_px(k, b, s); }

template <int V> struct Vec;
template <> struct Vec<8> { using type = uint4; };
template <> struct Vec<4> { using type = uint2; };

template <int V>
__device__ __forceinline__ typename Vec<V>::type rrc_vec(typename Vec<V>::type in, const double *k, const double *b)
{
    unsigned w[V / 2];
    const unsigned *p = reinterpret_cast<const unsigned *>(&in);
#pragma unroll
    for (int i = 0; i < V / 2; ++i) {
        unsigned lo = rrc_px(k[2 * i], b[2 * i], p[i] & 0xffffu);
        unsigned hi = rrc_px(k[2 * i + 1], b[2 * i + 1], p[i] >> 16);
        w[i] = lo | (hi << 16);
    }
    typename Vec<V>::type out;
    unsigned *q = reinterpret_cast<unsigned *>(&out);
#pragma unroll
    for (int i = 0; i < V / 2; ++i) q[i] = w[i];
    return out;
}

// flat, line-aligned, column-fixed (see header).  f = chunk index in the raster, g = f + a its
// index in the destination's 1 KiB-aligned frame (a = misalignment of dst in chunks, 0..63);
// lane g0 of super-row 0 handles g0, g0 + sr, g0 + 2 sr, ...  src may alias dst.
__global__ __launch_bounds__(kBlock) void rrc_u16_flat_kernel(const uint16_t *src, uint16_t *dst, int P, long nchunks,
                                                              long sr, int a, const double2 *__restrict__ kb,
                                                              long nsuper, long super_per_block)
{
    const long g0 = (long)blockIdx.x * kBlock + threadIdx.x;
    if (g0 >= sr) return;
    const long f0 = g0 - a;
    int cc = (int)(f0 % P);
    if (cc < 0) cc += P;
    const int col = cc * 8;
    double k[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double2 p = kb[col + i];
        k[i] = p.x;
        b[i] = p.y;
    }
    const long s0 = (long)blockIdx.y * super_per_block;
    long s1 = s0 + super_per_block;
    if (s1 > nsuper) s1 = nsuper;
    long f = f0 + s0 * sr;
    const uint4 *s = reinterpret_cast<const uint4 *>(src) + f;
    uint4 *d = reinterpret_cast<uint4 *>(dst) + f;
    for (long q = s0; q < s1; q += kRowsInFlight) {
        uint4 v[kRowsInFlight];
        bool ok[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
            const long fu = f + u * sr;
            ok[u] = q + u < s1 && fu >= 0 && fu < nchunks;
            if (ok[u]) v[u] = s[u * sr];
        }
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u)
            if (ok[u]) d[u * sr] = rrc_vec<8>(v[u], k, b);
        s += kRowsInFlight * sr;
        d += kRowsInFlight * sr;
        f += kRowsInFlight * sr;
    }
}

// planar raster, column-owned.  src may alias dst (in-place, as the reference).
template <int V>
__global__ __launch_bounds__(kBlock) void rrc_u16_kernel(const uint16_t *src, uint16_t *dst, int w, long h,
                                                         const double2 *__restrict__ kb, long rows_per_block)
{
    using VT = typename Vec<V>::type;
    const int x0 = (blockIdx.x * kBlock + threadIdx.x) * V;
    if (x0 >= w) return;
    double k[V], b[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        double2 p = kb[x0 + i];
        k[i] = p.x;
        b[i] = p.y;
    }
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > h) r1 = h;
    const uint16_t *s = src + r0 * (long)w + x0;
    uint16_t *d = dst + r0 * (long)w + x0;
    long r = r0;
    for (; r + kRowsInFlight <= r1; r += kRowsInFlight) {
        VT v[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) v[u] = *reinterpret_cast<const VT *>(s + (long)u * w);
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) *reinterpret_cast<VT *>(d + (long)u * w) = rrc_vec<V>(v[u], k, b);
        s += (long)kRowsInFlight * w;
        d += (long)kRowsInFlight * w;
    }
    for (; r < r1; ++r) {
        VT v = *reinterpret_cast<const VT *>(s);
        *reinterpret_cast<VT *>(d) = rrc_vec<V>(v, k, b);
        s += w;
        d += w;
    }
}

// any width / alignment: one pixel per lane over the flat raster
__global__ __launch_bounds__(kBlock) void rrc_u16_scalar_kernel(const uint16_t *src, uint16_t *dst, int w, long n,
                                                                const double2 *__restrict__ kb)
{
    long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; i < n; i += stride) {
        int x = (int)(i % w);
        double2 p = kb[x];
        dst[i] = (uint16_t)rrc_px(p.x, p.y, src[i]);
    }
}

// BIL MSS line (4 bands x bw px) -> 4 planar bands, RRC applied on the way (kb indexed by
// BIL column == band*bw + column-in-band).  A lane owns 8 BIL columns; each 4-pixel half
// lies inside one band because bw % 4 == 0, and goes out as one 8-byte store.
template <bool RRC>
__global__ __launch_bounds__(kBlock) void mss_split_rrc_kernel(const uint16_t *__restrict__ bil,
                                                               uint16_t *__restrict__ planes, size_t plane_stride,
                                                               int w, int bw, long lines,
                                                               const double2 *__restrict__ kb, long rows_per_block)
{
    const int x0 = (blockIdx.x * kBlock + threadIdx.x) * 8;
    if (x0 >= w) return;
    double k[8], b[8];
    if (RRC) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double2 p = kb[x0 + i];
            k[i] = p.x;
            b[i] = p.y;
        }
    }
    const int band0 = x0 / bw, band1 = (x0 + 4) / bw;
    uint16_t *d0 = planes + (size_t)band0 * plane_stride + (x0 - band0 * bw);
    uint16_t *d1 = planes + (size_t)band1 * plane_stride + (x0 + 4 - band1 * bw);
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > lines) r1 = lines;
    const uint16_t *s = bil + r0 * (long)w + x0;
    d0 += r0 * (long)bw;
    d1 += r0 * (long)bw;
    long r = r0;
    for (; r + kRowsInFlight <= r1; r += kRowsInFlight) {
        uint4 v[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) v[u] = *reinterpret_cast<const uint4 *>(s + (long)u * w);
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
            uint4 o = RRC ? rrc_vec<8>(v[u], k, b) : v[u];
            *reinterpret_cast<uint2 *>(d0 + (long)u * bw) = make_uint2(o.x, o.y);
            *reinterpret_cast<uint2 *>(d1 + (long)u * bw) = make_uint2(o.z, o.w);
        }
        s += (long)kRowsInFlight * w;
        d0 += (long)kRowsInFlight * bw;
        d1 += (long)kRowsInFlight * bw;
    }
    for (; r < r1; ++r) {
        uint4 v = *reinterpret_cast<const uint4 *>(s);
        uint4 o = RRC ? rrc_vec<8>(v, k, b) : v;
        *reinterpret_cast<uint2 *>(d0) = make_uint2(o.x, o.y);
        *reinterpret_cast<uint2 *>(d1) = make_uint2(o.z, o.w);
        s += w;
        d0 += bw;
        d1 += bw;
    }
}

// The same split walked FLAT over the destination planes (round 4): a band plane is a contiguous bw x lines raster whose
// 15000-byte lines (bw = 7500) are not multiples of 16, so the line-owned kernel above can only store 8 bytes per lane and
// its waves write half-filled cache-line sectors.  Here a plane is an array of 4-pixel UNITS (bw % 4 == 0: a unit never
// straddles a line); a lane owns a 16-byte CHUNK = two units of the plane's 1 KiB-aligned frame (a = units between the frame
// and the plane's first pixel), a wave stores one aligned KiB, and a lane advances by sr chunks with 2 sr a multiple of the
// units per line, so its 8 columns -- its LUT registers -- never change.  The two units of a chunk may lie on consecutive
// BIL lines: two 8-byte loads.  blockIdx.z = band.
template <bool RRC>
__global__ __launch_bounds__(kBlock) void mss_split_flat_kernel(const uint16_t *__restrict__ bil, uint16_t *__restrict__ planes,
                                                                size_t plane_stride, int w, int upl, long lines, const double2 *__restrict__ kb,
                                                                long sr, long nsuper, long super_per_block)
{
    const int band = blockIdx.z;
    const long g0 = (long)blockIdx.x * kBlock + threadIdx.x;
    if (g0 >= sr) return;
    uint16_t *plane = planes + (size_t)band * plane_stride;
    const long a = (long)(((uintptr_t)plane & 1023) >> 3);          // units
    const long U = (long)upl * lines;                                 // units of the plane
    uint2 *frame = reinterpret_cast<uint2 *>(plane) - a;             // unit g of the frame is unit g - a of the plane
    // the columns of the lane's two units (fixed: 2 sr is a multiple of upl)
    long u0 = 2 * g0 - a;
    int c0 = (int)(((u0 % upl) + upl) % upl), c1 = c0 + 1 == upl ? 0 : c0 + 1;
    double k[8], b[8];
    if (RRC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double2 p0 = kb[band * (upl * 4) + c0 * 4 + i], p1 = kb[band * (upl * 4) + c1 * 4 + i];
            k[i] = p0.x; b[i] = p0.y; k[4 + i] = p1.x; b[4 + i] = p1.y;
        }
    }
    const long s0 = (long)blockIdx.y * super_per_block;
    long s1 = s0 + super_per_block;
    if (s1 > nsuper) s1 = nsuper;
    const long lines_per_step = 2 * sr / upl;
    // unit u of the plane = BIL line u / upl, columns band * bw + 4 (u % upl)
    long u = u0 + 2 * s0 * sr;
    long r0 = (u - c0) / upl;                                        // line of unit u (exact: u - c0 is a multiple of upl, also below 0)
    const int wrap = c1 == 0 ? 1 : 0;                                // the second unit starts the next line
    const uint2 *src0 = reinterpret_cast<const uint2 *>(bil + (long)band * upl * 4 + c0 * 4), *src1 = reinterpret_cast<const uint2 *>(bil + (long)band * upl * 4 + c1 * 4);
    const long wq = w / 4;                                           // BIL line pitch in units
    for (long q = s0; q < s1; q += kRowsInFlight) {
        uint2 v0[kRowsInFlight], v1[kRowsInFlight];
        bool ok0[kRowsInFlight], ok1[kRowsInFlight];
#pragma unroll
        for (int t = 0; t < kRowsInFlight; ++t) {
            const long ut = u + 2 * t * sr, rt = r0 + t * lines_per_step;
            ok0[t] = q + t < s1 && ut >= 0 && ut < U;
            ok1[t] = q + t < s1 && ut + 1 >= 0 && ut + 1 < U;
            if (ok0[t]) v0[t] = src0[rt * wq];
            if (ok1[t]) v1[t] = src1[(rt + wrap) * wq];
        }
#pragma unroll
        for (int t = 0; t < kRowsInFlight; ++t) {
            const long gt = 2 * (g0 + (q + t) * sr);
            uint4 o = make_uint4(v0[t].x, v0[t].y, v1[t].x, v1[t].y);
            if (RRC) o = rrc_vec<8>(o, k, b);
            if (ok0[t] && ok1[t]) *reinterpret_cast<uint4 *>(frame + gt) = o;
            else if (ok0[t]) frame[gt] = make_uint2(o.x, o.y);
            else if (ok1[t]) frame[gt + 1] = make_uint2(o.z, o.w);
        }
        u += 2 * kRowsInFlight * sr;
        r0 += kRowsInFlight * lines_per_step;
    }
}

template <bool RRC>
__global__ __launch_bounds__(kBlock) void mss_split_rrc_scalar_kernel(const uint16_t *__restrict__ bil,
                                                                      uint16_t *__restrict__ planes,
                                                                      size_t plane_stride, int w, int bw, long n,
                                                                      const double2 *__restrict__ kb)
{
    long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; i < n; i += stride) {
        long row = i / w;
        int x = (int)(i - row * w);
        int band = x / bw;
        if (band > 3) continue;             // w % 4 != 0: trailing pixels belong to no band
        unsigned v = bil[i];
        if (RRC) {
            double2 p = kb[x];
            v = rrc_px(p.x, p.y, v);
        }
        planes[(size_t)band * plane_stride + row * bw + (x - band * bw)] = (uint16_t)v;
    }
}

// RRC of a w-column window of one raster into a window of another (separate pitches): the fused prestitch -> stitch
// form writes RRC(PAN1)[:, 0 : W - fold] straight into the left half of the stitched raster (imageop.h:340-351), so
// .RRC.RAW of CCD 1 is never materialised.  A lane owns 8 columns (LUT in registers) and walks a block of lines, four
// lines in flight; the last group of a window whose width is not a multiple of 8 stores its pixels one by one.
__global__ __launch_bounds__(kBlock) void rrc_u16_window_kernel(const uint16_t *__restrict__ src, long src_pitch, uint16_t *__restrict__ dst,
                                                                long dst_pitch, int w, long h, const double2 *__restrict__ kb,
                                                                long rows_per_block, int period)
{
    // Store alignment decides the bandwidth (see the header of this file): a destination line pitch that is not a multiple
    // of 128 bytes shifts every line against the cache lines.  Lines r = rho (mod period) share one shift
    // (period = 128 / gcd(128, pitch bytes mod 128)), so blockIdx.z = rho owns those lines and starts its 8-column groups
    // where THEIR cache lines start: every wave then stores one aligned KiB per line, while a lane still owns 8 fixed
    // columns (LUT in registers).
    const int rho = blockIdx.z;
    const int shift = (int)(((128u - (unsigned)(((uintptr_t)dst + (size_t)rho * (size_t)dst_pitch * 2) & 127u)) & 127u) >> 1);    // columns, multiple of 8
    const int x0 = shift - 64 + (blockIdx.x * kBlock + threadIdx.x) * 8;
    if (x0 < 0 || x0 >= w) return;
    double k[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2 p = kb[x0 + i < w ? x0 + i : w - 1];
        k[i] = p.x;
        b[i] = p.y;
    }
    const bool whole = x0 + 8 <= w;
    const long r0 = (long)blockIdx.y * rows_per_block + rho;         // rows_per_block is a multiple of period
    long r1 = (long)(blockIdx.y + 1) * rows_per_block;
    if (r1 > h) r1 = h;
    const uint16_t *s = src + r0 * src_pitch + x0;
    uint16_t *d = dst + r0 * dst_pitch + x0;
    const long sstep = (long)period * src_pitch, dstep = (long)period * dst_pitch;
    auto put = [&](uint16_t *q, uint4 o) {
        if (whole) { *reinterpret_cast<uint4 *>(q) = o; return; }
        const unsigned wv[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (x0 + j < w) q[j] = (uint16_t)((wv[j >> 1] >> (16 * (j & 1))) & 0xffffu);
    };
    // (a partial group still loads 16 bytes: the source window is followed by the rest of its line -- host-checked)
    long r = r0;
    for (; r + (long)(kRowsInFlight - 1) * period < r1; r += (long)kRowsInFlight * period) {
        uint4 v[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) v[u] = *reinterpret_cast<const uint4 *>(s + u * sstep);
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) put(d + u * dstep, rrc_vec<8>(v[u], k, b));
        s += kRowsInFlight * sstep;
        d += kRowsInFlight * dstep;
    }
    for (; r < r1; r += period) {
        put(d, rrc_vec<8>(*reinterpret_cast<const uint4 *>(s), k, b));
        s += sstep;
        d += dstep;
    }
}

__global__ __launch_bounds__(kBlock) void rrc_u16_window_scalar_kernel(const uint16_t *__restrict__ src, long src_pitch, uint16_t *__restrict__ dst,
                                                                       long dst_pitch, int w, long n, const double2 *__restrict__ kb)
{
    long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; i < n; i += stride) {
        const long r = i / w;
        const int x = (int)(i - r * w);
        const double2 p = kb[x];
        dst[r * dst_pitch + x] = (uint16_t)rrc_px(p.x, p.y, src[r * src_pitch + x]);
    }
}

// grid.y so that the launch has ~16 blocks per CU while each block keeps >= 64 rows
inline void row_blocks(const oip_ctx *ctx, int gx, long h, long *rows_per_block, int *gy)
{
    long want = (long)ctx->cu_count * 16 / (gx > 0 ? gx : 1);
    if (want < 1) want = 1;
    long rpb = (h + want - 1) / want;
    if (rpb < 64) rpb = 64;
    rpb = (rpb + kRowsInFlight - 1) / kRowsInFlight * kRowsInFlight;
    long g = (h + rpb - 1) / rpb;
    if (g < 1) g = 1;
    if (g > 65535) { g = 65535; rpb = (h + g - 1) / g; rpb = (rpb + kRowsInFlight - 1) / kRowsInFlight * kRowsInFlight; g = (h + rpb - 1) / rpb; }
    *rows_per_block = rpb;
    *gy = (int)g;
}

}  // namespace

// The launcher proper, on an explicit stream and without touching the context's profiler or error-free state: the
// staging layer (oip_rrc_u16_host) calls it from a second host thread on streams of its own while the first thread
// drives kernels through ctx->stream.
int oip_rrc_launch(oip_ctx *ctx, hipStream_t stream, const uint16_t *d_src, uint16_t *d_dst, int w, long h, const double *d_kb)
{
    if (w <= 0 || h < 0 || !d_src || !d_dst || !d_kb) return oip_fail(ctx, OIP_E_INVALID, "oip_rrc_u16: bad argument");
    if (h == 0) return OIP_OK;
    const double2 *kb = reinterpret_cast<const double2 *>(d_kb);
    const uintptr_t align = (uintptr_t)d_src | (uintptr_t)d_dst;
    if (w % 8 == 0 && (align & 15) == 0) {
        const int P = w / 8;
        long g = P, t = 64;
        while (t) { long r = g % t; g = t; t = r; }
        const long sr = (long)P / g * 64;                          // lcm(P, 64) chunks = whole lines
        const long nchunks = (long)P * h;
        const int a = (int)(((uintptr_t)d_dst >> 4) & 63);
        const long nsuper = (nchunks + a + sr - 1) / sr;
        long spb = 16;                                             // chunks per lane per LUT load
        long gy = (nsuper + spb - 1) / spb;
        if (gy > 65535) { gy = 65535; spb = (nsuper + gy - 1) / gy; spb = (spb + kRowsInFlight - 1) / kRowsInFlight * kRowsInFlight; gy = (nsuper + spb - 1) / spb; }
        const long gx = (sr + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(rrc_u16_flat_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, stream, d_src, d_dst,
                           P, nchunks, sr, a, kb, nsuper, spb);
    } else if (w % 4 == 0 && (align & 7) == 0) {
        int gx = (w / 4 + kBlock - 1) / kBlock, gy;
        long rpb;
        row_blocks(ctx, gx, h, &rpb, &gy);
        hipLaunchKernelGGL(rrc_u16_kernel<4>, dim3(gx, gy), dim3(kBlock), 0, stream, d_src, d_dst, w, h, kb, rpb);
    } else {
        long n = (long)w * h;
        long blocks = (n + kBlock - 1) / kBlock;
        long cap = (long)ctx->cu_count * 32;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(rrc_u16_scalar_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, stream, d_src, d_dst, w, n, kb);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

extern "C" int oip_rrc_u16(oip_ctx *ctx, const uint16_t *d_src, uint16_t *d_dst, int w, long h, const double *d_kb)
{
    OIP_CHECK_CTX(ctx);
    if (w <= 0 || h < 0 || !d_src || !d_dst || !d_kb) return oip_fail(ctx, OIP_E_INVALID, "oip_rrc_u16: bad argument");
    const uintptr_t align = (uintptr_t)d_src | (uintptr_t)d_dst;
    OipProfScope prof(ctx, (w % 8 == 0 && (align & 15) == 0) ? "rrc_u16_flat_kernel" : "rrc_u16_kernel");
    return oip_rrc_launch(ctx, ctx->stream, d_src, d_dst, w, h, d_kb);
}

// IMO::InplaceRRC (imageop.h:129-138) on a window: columns [0, w) of h lines of a raster of pitch src_pitch, written to a
// raster of pitch dst_pitch (pitches in pixels; d_kb: the w (k, b) pairs of the window's columns).
extern "C" int oip_rrc_u16_window(oip_ctx *ctx, const uint16_t *d_src, long src_pitch, uint16_t *d_dst, long dst_pitch, int w, long h,
                                  const double *d_kb)
{
    OIP_CHECK_CTX(ctx);
    if (w <= 0 || h < 0 || !d_src || !d_dst || !d_kb || src_pitch < w || dst_pitch < w)
        return oip_fail(ctx, OIP_E_INVALID, "oip_rrc_u16_window: bad argument");
    if (h == 0) return OIP_OK;
    const double2 *kb = reinterpret_cast<const double2 *>(d_kb);
    OipProfScope prof(ctx, "rrc_u16_window_kernel");
    // vector form: 16-byte loads and stores on both sides, and a partial last group may read up to 7 pixels past the window
    const bool vec = src_pitch % 8 == 0 && dst_pitch % 8 == 0 && (((uintptr_t)d_src | (uintptr_t)d_dst) & 15) == 0 &&
                     (w % 8 == 0 || src_pitch >= (long)(w + 7) / 8 * 8);
    if (vec) {
        long g = (dst_pitch * 2) % 128, t = 128;
        while (g) { const long r = t % g; t = g; g = r; }
        const int period = (int)(128 / t);                             // lines with the same shift against the cache lines
        int gx = ((w + 7) / 8 + 8 + kBlock - 1) / kBlock, gy;          // + 8 groups: the shifted column origin
        long rpb;
        row_blocks(ctx, gx * period, h, &rpb, &gy);
        rpb = (rpb + period - 1) / period * period;
        gy = (int)((h + rpb - 1) / rpb);
        hipLaunchKernelGGL(rrc_u16_window_kernel, dim3(gx, gy, period), dim3(kBlock), 0, ctx->stream, d_src, src_pitch, d_dst, dst_pitch, w, h, kb,
                           rpb, period);
    } else {
        const long n = (long)w * h;
        long blocks = (n + kBlock - 1) / kBlock;
        const long cap = (long)ctx->cu_count * 32;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(rrc_u16_window_scalar_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d_src, src_pitch, d_dst, dst_pitch, w,
                           n, kb);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

extern "C" int oip_mss_split_rrc_u16(oip_ctx *ctx, const uint16_t *d_bil, uint16_t *d_planes, size_t plane_stride,
                                     int w, long lines, const double *d_kb4)
{
    OIP_CHECK_CTX(ctx);
    if (w <= 0 || lines < 0 || !d_bil || !d_planes) return oip_fail(ctx, OIP_E_INVALID, "oip_mss_split_rrc_u16: bad argument");
    const int bw = w / OIP_MSS_BANDS;
    if (bw <= 0) return oip_fail(ctx, OIP_E_INVALID, "oip_mss_split_rrc_u16: line narrower than 4 pixels");
    if (plane_stride < (size_t)bw * (size_t)lines) return oip_fail(ctx, OIP_E_INVALID, "oip_mss_split_rrc_u16: plane_stride too small");
    if (lines == 0) return OIP_OK;
    OipProfScope prof(ctx, "mss_split_rrc_kernel");
    const double2 *kb = reinterpret_cast<const double2 *>(d_kb4);
    const bool fast = (w % 8 == 0) && (bw % 4 == 0) && (w == bw * 4) && (((uintptr_t)d_bil & 15) == 0) &&
                      (((uintptr_t)d_planes & 7) == 0) && (plane_stride % 4 == 0);
    // the flat form (16-byte stores, aligned KiB per wave) wherever its index arithmetic holds; OIP_MSS_SPLIT_FLAT=0: the line-owned kernel
    const char *ef = getenv("OIP_MSS_SPLIT_FLAT");
    const int upl = bw / 4;
    long sr = 0;
    if (fast && !(ef && atoi(ef) == 0)) {
        long g = upl % 2 == 0 ? upl / 2 : upl, t = 64, base = g;
        while (t) { long r = g % t; g = t; t = r; }
        sr = base / g * 64;                                          // lcm(upl / gcd(upl, 2), 64): 2 sr units = whole lines
        if (sr > (1L << 22)) sr = 0;                                 // absurd widths: the line-owned kernel
    }
    if (fast && sr > 0) {
        const long U = (long)upl * lines;
        const long nsuper = (U + 127 + 2 * sr - 1) / (2 * sr) + 1;   // chunks g = g0 + q sr cover units up to U + a (a < 128)
        long spb = 16;
        long gy = (nsuper + spb - 1) / spb;
        if (gy > 65535) { gy = 65535; spb = (nsuper + gy - 1) / gy; spb = (spb + kRowsInFlight - 1) / kRowsInFlight * kRowsInFlight; gy = (nsuper + spb - 1) / spb; }
        const long gx = (sr + kBlock - 1) / kBlock;
        if (kb)
            hipLaunchKernelGGL(mss_split_flat_kernel<true>, dim3((unsigned)gx, (unsigned)gy, OIP_MSS_BANDS), dim3(kBlock), 0, ctx->stream, d_bil, d_planes,
                               plane_stride, w, upl, lines, kb, sr, nsuper, spb);
        else
            hipLaunchKernelGGL(mss_split_flat_kernel<false>, dim3((unsigned)gx, (unsigned)gy, OIP_MSS_BANDS), dim3(kBlock), 0, ctx->stream, d_bil, d_planes,
                               plane_stride, w, upl, lines, kb, sr, nsuper, spb);
    } else if (fast) {
        int gx = (w / 8 + kBlock - 1) / kBlock, gy;
        long rpb;
        row_blocks(ctx, gx, lines, &rpb, &gy);
        if (kb)
            hipLaunchKernelGGL(mss_split_rrc_kernel<true>, dim3(gx, gy), dim3(kBlock), 0, ctx->stream, d_bil, d_planes,
                               plane_stride, w, bw, lines, kb, rpb);
        else
            hipLaunchKernelGGL(mss_split_rrc_kernel<false>, dim3(gx, gy), dim3(kBlock), 0, ctx->stream, d_bil, d_planes,
                               plane_stride, w, bw, lines, kb, rpb);
    } else {
        long n = (long)w * lines;
        long blocks = (n + kBlock - 1) / kBlock;
        long cap = (long)ctx->cu_count * 32;
        if (blocks > cap) blocks = cap;
        if (kb)
            hipLaunchKernelGGL(mss_split_rrc_scalar_kernel<true>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                               d_bil, d_planes, plane_stride, w, bw, n, kb);
        else
            hipLaunchKernelGGL(mss_split_rrc_scalar_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream,
                               d_bil, d_planes, plane_stride, w, bw, n, kb);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// oip_rrc_u16_host (the host-buffer form of this seam) lives in staging.hip with the rest of the raster I/O staging
