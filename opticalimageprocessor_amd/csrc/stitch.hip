// stitch.hip -- RAW strip stitch on gfx950.
//
// Replaces the line loop of IMO::StitchBigRaw (imageop.h:340-351): output line i is
// left[i][0 : W-fold] followed by right[i][fold : W]  (fold = --fold-cols / 2, main.cpp:189).
// Pure 2 B in / 2 B out copy; the only difficulty is alignment: the right half starts
// `fold` pixels into its source line and W-fold pixels into the output line, so source and
// destination are in general only 2-byte aligned relative to each other.  A lane owns 8
// output pixels (one 16-byte aligned store); it reads the five dwords covering its source
// span and funnel-shifts by 16 bits when the source is odd-pixel aligned.
#include "oip_internal.h"

namespace {

constexpr int kBlock = 256;

// 8 consecutive u16 starting at element index `e` of `p` (e may be odd), as 4 dwords.
// The dwords covering the span are clamped to [lo, hi) element indices (whole dwords inside
// the allocation); lanes never dereference outside the raster.
__device__ __forceinline__ uint4 load8_u16(const uint16_t *__restrict__ p, long e, long n_elems)
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    long d0 = e >> 1;                       // first dword
    const long dmax = (n_elems - 1) >> 1;   // last dword that holds a valid element
    uint32_t w[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        long di = d0 + i;
        w[i] = q[di > dmax ? dmax : di];
    }
    uint4 o;
    if (e & 1) {
        o.x = __builtin_amdgcn_alignbit(w[1], w[0], 16);
        o.y = __builtin_amdgcn_alignbit(w[2], w[1], 16);
        o.z = __builtin_amdgcn_alignbit(w[3], w[2], 16);
        o.w = __builtin_amdgcn_alignbit(w[4], w[3], 16);
    } else {
        o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    }
    return o;
}

// Flat over the OUTPUT raster: lane handles 16-byte chunk f of the output (always a full, aligned
// 16-byte store; a wave writes one aligned KiB whatever the line pitch -- 2*(W-fold) pixels is
// rarely a multiple of 64), finds its line / column by one integer division and gathers the 8
// source pixels from the left or the right strip.  4 chunks in flight per lane.
// requires: out base 16-byte aligned, (2*half) % 8 == 0, left/right bases 4-byte aligned.
__global__ __launch_bounds__(kBlock) void stitch_rows_kernel(const uint16_t *__restrict__ left,
                                                             const uint16_t *__restrict__ right,
                                                             uint16_t *__restrict__ out, int W, long L, int fold)
{
    const int half = W - fold;
    const int cpr = (2 * half) / 8;               // chunks per output line
    const long nchunks = (long)cpr * L;
    const long n_elems = (long)W * L;
    constexpr int U = 4;
    const long stride = (long)gridDim.x * kBlock;
    for (long f0 = (long)blockIdx.x * kBlock + threadIdx.x; f0 < nchunks; f0 += stride * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long f = f0 + u * stride;
            if (f >= nchunks) break;
            const long r = f / cpr;
            const int x0 = (int)(f - r * cpr) * 8;
            if (x0 + 8 <= half) {
                v[u] = load8_u16(left, r * W + x0, n_elems);
            } else if (x0 >= half) {
                v[u] = load8_u16(right, r * W + fold + (x0 - half), n_elems);
            } else {
                unsigned short t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    int x = x0 + i;
                    t[i] = x < half ? left[r * W + x] : right[r * W + fold + (x - half)];
                }
                v[u].x = t[0] | ((unsigned)t[1] << 16); v[u].y = t[2] | ((unsigned)t[3] << 16);
                v[u].z = t[4] | ((unsigned)t[5] << 16); v[u].w = t[6] | ((unsigned)t[7] << 16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long f = f0 + u * stride;
            if (f >= nchunks) break;
            reinterpret_cast<uint4 *>(out)[f] = v[u];
        }
    }
}

__global__ __launch_bounds__(kBlock) void stitch_rows_scalar_kernel(const uint16_t *__restrict__ left,
                                                                    const uint16_t *__restrict__ right,
                                                                    uint16_t *__restrict__ out, int W, long L, int fold)
{
    const int half = W - fold;
    const long ow = 2L * half;
    const long n = ow * L;
    long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long stride = (long)gridDim.x * kBlock;
    for (; i < n; i += stride) {
        long r = i / ow;
        int x = (int)(i - r * ow);
        out[i] = x < half ? left[r * W + x] : right[r * W + fold + (x - half)];
    }
}

// In-place sample permutation of an interleaved 4-channel image: sample i of every pixel becomes sample order[i].  A lane
// owns two pixels (one 16-byte load and store of its own: in place is safe); `sel` holds the four selectors, 4 bits each.
__global__ __launch_bounds__(kBlock) void permute_u16x4_kernel(uint16_t *__restrict__ img, long npairs, long npix, unsigned sel)
{
    const long stride = (long)gridDim.x * kBlock;
    for (long f = (long)blockIdx.x * kBlock + threadIdx.x; f < npairs; f += stride) {
        if (2 * f + 1 < npix) {
            uint4 v = reinterpret_cast<uint4 *>(img)[f], o;
            const unsigned a[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
            const unsigned b[4] = {v.z & 0xffffu, v.z >> 16, v.w & 0xffffu, v.w >> 16};
            o.x = a[sel & 3] | (a[(sel >> 4) & 3] << 16);
            o.y = a[(sel >> 8) & 3] | (a[(sel >> 12) & 3] << 16);
            o.z = b[sel & 3] | (b[(sel >> 4) & 3] << 16);
            o.w = b[(sel >> 8) & 3] | (b[(sel >> 12) & 3] << 16);
            reinterpret_cast<uint4 *>(img)[f] = o;
        } else {                                                        // the odd last pixel
            uint2 v = reinterpret_cast<uint2 *>(img)[2 * f], o;
            const unsigned a[4] = {v.x & 0xffffu, v.x >> 16, v.y & 0xffffu, v.y >> 16};
            o.x = a[sel & 3] | (a[(sel >> 4) & 3] << 16);
            o.y = a[(sel >> 8) & 3] | (a[(sel >> 12) & 3] << 16);
            reinterpret_cast<uint2 *>(img)[2 * f] = o;
        }
    }
}

}  // namespace

extern "C" int oip_permute_u16x4(oip_ctx *ctx, uint16_t *d_img, size_t npixels, const int *order)
{
    OIP_CHECK_CTX(ctx);
    if (!order || (!d_img && npixels)) return oip_fail(ctx, OIP_E_INVALID, "oip_permute_u16x4: bad argument");
    unsigned sel = 0;
    for (int i = 0; i < 4; ++i) {
        if (order[i] < 0 || order[i] > 3) return oip_fail(ctx, OIP_E_INVALID, "oip_permute_u16x4: sample index outside 0..3");
        sel |= (unsigned)order[i] << (4 * i);
    }
    if (((uintptr_t)d_img & 15) != 0) return oip_fail(ctx, OIP_E_INVALID, "oip_permute_u16x4: image not 16-byte aligned");
    if (npixels == 0 || sel == 0x3210u) return OIP_OK;
    OipProfScope prof(ctx, "permute_u16x4_kernel");
    const long npairs = (long)((npixels + 1) / 2);
    long blocks = (npairs + kBlock - 1) / kBlock;
    const long cap = (long)ctx->cu_count * 64;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(permute_u16x4_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d_img, npairs, (long)npixels, sel);
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

extern "C" int oip_stitch_rows_u16(oip_ctx *ctx, const uint16_t *d_left, const uint16_t *d_right, uint16_t *d_out,
                                   int W, long L, int fold)
{
    OIP_CHECK_CTX(ctx);
    if (!d_left || !d_right || !d_out || W <= 0 || L < 0 || fold < 0 || fold >= W)
        return oip_fail(ctx, OIP_E_INVALID, "oip_stitch_rows_u16: bad argument");
    if (L == 0) return OIP_OK;
    OipProfScope prof(ctx, "stitch_rows_kernel");
    const int half = W - fold;
    const int ow = 2 * half;
    const bool fast = (ow % 8 == 0) && (((uintptr_t)d_out & 15) == 0) && (((uintptr_t)d_left & 3) == 0) &&
                      (((uintptr_t)d_right & 3) == 0) && ((long)W * L >= 16 && ((long)W * L) % 2 == 0);
    if (fast) {
        const long nchunks = (long)(ow / 8) * L;
        long blocks = (nchunks + (long)kBlock * 4 - 1) / ((long)kBlock * 4);
        const long cap = (long)ctx->cu_count * 128;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(stitch_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d_left, d_right,
                           d_out, W, L, fold);
    } else {
        long n = (long)ow * L;
        long blocks = (n + kBlock - 1) / kBlock;
        long cap = (long)ctx->cu_count * 32;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(stitch_rows_scalar_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d_left,
                           d_right, d_out, W, L, fold);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}
