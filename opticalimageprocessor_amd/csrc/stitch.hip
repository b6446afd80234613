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

// requires: out base 16-byte aligned and (2*half) % 8 == 0 so every output line starts on a
// 16-byte boundary; left/right bases 4-byte aligned.
__global__ __launch_bounds__(kBlock) void stitch_rows_kernel(const uint16_t *__restrict__ left,
                                                             const uint16_t *__restrict__ right,
                                                             uint16_t *__restrict__ out, int W, long L, int fold,
                                                             long rows_per_block)
{
    const int half = W - fold;
    const int ow = 2 * half;
    const int x0 = (blockIdx.x * kBlock + threadIdx.x) * 8;   // first output pixel of this lane
    if (x0 >= ow) return;
    const long n_elems = (long)W * L;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > L) r1 = L;
    for (long r = r0; r < r1; ++r) {
        uint4 v;
        if (x0 + 8 <= half) {
            v = load8_u16(left, r * W + x0, n_elems);
        } else if (x0 >= half) {
            v = load8_u16(right, r * W + fold + (x0 - half), n_elems);
        } else {
            // the 16-byte chunk straddles the seam: element-wise
            unsigned short t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int x = x0 + i;
                t[i] = x < half ? left[r * W + x] : right[r * W + fold + (x - half)];
            }
            v.x = t[0] | ((unsigned)t[1] << 16); v.y = t[2] | ((unsigned)t[3] << 16);
            v.z = t[4] | ((unsigned)t[5] << 16); v.w = t[6] | ((unsigned)t[7] << 16);
        }
        *reinterpret_cast<uint4 *>(out + r * (long)ow + x0) = v;
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

}  // namespace

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
        int gx = (ow / 8 + kBlock - 1) / kBlock;
        long want = (long)ctx->cu_count * 16 / gx;
        if (want < 1) want = 1;
        long rpb = (L + want - 1) / want;
        if (rpb < 16) rpb = 16;
        long gy = (L + rpb - 1) / rpb;
        if (gy > 65535) { gy = 65535; rpb = (L + gy - 1) / gy; gy = (L + rpb - 1) / rpb; }
        hipLaunchKernelGGL(stitch_rows_kernel, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_left, d_right,
                           d_out, W, L, fold, rpb);
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
