// remap.hip -- constant-shift bicubic resampling of CCD-2 (the .PRESTT raster) on gfx950.
//
// Replaces Stitcher::PreStitch (stitcher.h:83-139) driving IMO::SectionaryRemap
// (imageop.h:230-275) and cv::remap(INTER_CUBIC, BORDER_CONSTANT) (imageop.h:258).
//
// The reference streams two section_rows x W float maps (8 B/px) through cv::remap; here the
// maps never exist: mapx depends only on the column and mapy only on the section-relative
// line, both through the f32 rounding of (x + dx) / (y + dy) (stitcher.h:93-99), so
//   * a tiny pre-kernel turns every output line into {4 source lines, y phase, path flags}
//     (oip_geom.h: section seams, cuts and the stale tail of the reused section buffer);
//   * the pixel kernel computes the column's integer offset / x phase once per lane and walks
//     down its line tile with a 4x4 register window, so each new output line costs one new
//     source line per lane (HBM traffic: 2 B read + 2 B written per pixel).
#include "oip_bicubic.h"
#include "oip_geom.h"
#include "oip_internal.h"

namespace {

constexpr int kBlock = 256;

struct RowInfo {
    int src[4];     // source line of each vertical tap relative to d_src, -1 = constant border
    int fy;         // y phase
    int flags;      // bit0: all four taps inside the section buffer (interior path in y)
                    // bit1: window entirely above/below the buffer (output 0)
    int pad[2];
};

__global__ void shift_rows_kernel(RowInfo *rows, OipShiftGeom g, long out_row0, long out_rows, long src_row0,
                                  long src_rows)
{
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= out_rows) return;
    OipShiftRow o = oip_shift_row(g, out_row0 + r);
    RowInfo ri;
    for (int t = 0; t < 4; ++t) {
        long s = o.src[t];
        if (s >= 0) {
            s -= src_row0;
            if (s < 0 || s >= src_rows) s = -1;   // unreachable: the host checked the halo
        }
        ri.src[t] = (int)s;
    }
    ri.fy = o.fy;
    int flags = 0;
    if ((unsigned)o.iy < (unsigned)(g.section_rows - 3 > 0 ? g.section_rows - 3 : 0)) flags |= 1;
    if (o.iy >= g.section_rows || o.iy + 4 <= 0) flags |= 2;
    ri.flags = flags;
    ri.pad[0] = ri.pad[1] = 0;
    rows[r] = ri;
}

__device__ __forceinline__ void load_tap_row(const uint16_t *__restrict__ src, int row, int W, const int c[4],
                                             unsigned xmask, float out[4])
{
    if (row < 0) {
        out[0] = out[1] = out[2] = out[3] = 0.f;
        return;
    }
    const uint16_t *p = src + (long)row * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (xmask & (1u << j)) ? (float)p[c[j]] : 0.f;
}

__global__ __launch_bounds__(kBlock) void remap_shift_kernel(const uint16_t *__restrict__ src,
                                                             uint16_t *__restrict__ dst,
                                                             const RowInfo *__restrict__ rows, int W, long out_rows,
                                                             double dx, const float *__restrict__ tab1d,
                                                             int rows_per_block)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= W) return;
    // stitcher.h:96  mapx = (float)(x + mDeltaX); imgwarp.cpp: sx = cvRound(mapx*32)
    const float mapx = (float)((double)x + dx);
    const int sx = oip_cvround(mapx * 32.0f);
    const int ix = oip_sat_short(sx >> 5) - 1;
    const int fx = sx & 31;
    float wx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wx[j] = tab1d[fx * 4 + j];
    unsigned xmask = 0;
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int cj = ix + j;
        if (cj >= 0 && cj < W) xmask |= 1u << j;
        c[j] = cj < 0 ? 0 : (cj > W - 1 ? W - 1 : cj);
    }
    const bool x_inside = (unsigned)ix < (unsigned)(W - 3 > 0 ? W - 3 : 0);
    const bool x_out = (ix >= W) || (ix + 4 <= 0);

    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > out_rows) r1 = out_rows;

    float v[4][4];
    int cur[4] = {-2, -2, -2, -2};
    int cur_fy = -1;
    float wy[4] = {0.f, 0.f, 0.f, 0.f};
    for (long r = r0; r < r1; ++r) {
        const RowInfo ri = rows[r];
        if (ri.src[0] == cur[1] && ri.src[1] == cur[2] && ri.src[2] == cur[3] && cur[1] != -2) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[t][j] = v[t + 1][j];
            }
            load_tap_row(src, ri.src[3], W, c, xmask, v[3]);
        } else if (!(ri.src[0] == cur[0] && ri.src[1] == cur[1] && ri.src[2] == cur[2] && ri.src[3] == cur[3])) {
#pragma unroll
            for (int t = 0; t < 4; ++t) load_tap_row(src, ri.src[t], W, c, xmask, v[t]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) cur[t] = ri.src[t];
        if (ri.fy != cur_fy) {
            cur_fy = ri.fy;
#pragma unroll
            for (int j = 0; j < 4; ++j) wy[j] = tab1d[cur_fy * 4 + j];
        }
        float sum;
        if ((ri.flags & 2) || x_out) {
            sum = 0.f;
        } else if ((ri.flags & 1) && x_inside) {
            sum = oip_bicubic_interior(v, wx, wy);
        } else {
            unsigned ymask = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (ri.src[t] >= 0) ymask |= 1u << t;
            sum = oip_bicubic_border(v, wx, wy, xmask, ymask);
        }
        dst[r * (long)W + x] = (uint16_t)oip_sat_u16(sum);
    }
}

}  // namespace

// Source lines read by a window of output lines.  Inside one piece of the output (top cut,
// one section's body, bottom cut) the tap lines grow monotonically with the output line, so
// only piece end points need evaluating -- plus, one by one, the last lines of the final
// section and the bottom cut, whose taps can fall back into the previous section's lines
// (the stale tail of the reused section buffer).
static void shift_range_accum(const OipShiftGeom &g, long gy, long *lo, long *hi)
{
    OipShiftRow o = oip_shift_row(g, gy);
    for (int t = 0; t < 4; ++t) {
        if (o.src[t] < 0) continue;
        if (*lo < 0 || o.src[t] < *lo) *lo = o.src[t];
        if (o.src[t] + 1 > *hi) *hi = o.src[t] + 1;
    }
}

extern "C" int oip_remap_shift_src_range(long out_row0, long out_rows, long L, double dy, int section_rows,
                                         long *first, long *last)
{
    if (out_rows < 0 || out_row0 < 0 || out_row0 + out_rows > L || section_rows <= 0 || !(dy == dy)) return OIP_E_INVALID;
    OipShiftGeom g = oip_shift_geom(1, L, 0.0, dy, section_rows);
    if (g.step <= 0 || g.nsec <= 0) return OIP_E_INVALID;
    long lo = -1, hi = -1;
    const long a = out_row0, b = out_row0 + out_rows;   // [a, b)
    auto piece = [&](long p0, long p1, bool every) {     // [p0, p1) clipped to [a, b)
        if (p0 < a) p0 = a;
        if (p1 > b) p1 = b;
        if (p0 >= p1) return;
        if (every) { for (long r = p0; r < p1; ++r) shift_range_accum(g, r, &lo, &hi); return; }
        shift_range_accum(g, p0, &lo, &hi);
        shift_range_accum(g, p1 - 1, &lo, &hi);
    };
    piece(0, g.ucut, true);
    for (int s = 0; s < g.nsec; ++s) {
        long p0 = (long)s * g.step + g.ucut;
        long p1 = (s == g.nsec - 1) ? L - g.bcut : (long)(s + 1) * g.step + g.ucut;
        if (s == g.nsec - 1) {
            long tail = p1 - 8 > p0 ? p1 - 8 : p0;
            piece(p0, tail, false);
            piece(tail, p1, true);
        } else {
            piece(p0, p1, false);
        }
    }
    piece(L - g.bcut, L, true);
    if (lo < 0) lo = hi = 0;
    if (first) *first = lo;
    if (last) *last = hi;
    return OIP_OK;
}

extern "C" int oip_remap_shift_bicubic_u16(oip_ctx *ctx, const uint16_t *d_src, long src_row0, long src_rows,
                                           uint16_t *d_dst, long out_row0, long out_rows, int W, long L, double dx,
                                           double dy, int section_rows, int row_guard)
{
    OIP_CHECK_CTX(ctx);
    if (!d_src || !d_dst || W <= 0 || L <= 0 || section_rows <= 3)
        return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_bicubic_u16: bad argument");
    if (L <= row_guard) return oip_fail(ctx, OIP_E_INVALID, "too few data rows, please use cv::remap()");   // imageop.h:242-244
    if (row_guard < section_rows) return oip_fail(ctx, OIP_E_INVALID, "row_guard must be >= section_rows");
    if (section_rows > 32767 || W > 32767) return oip_fail(ctx, OIP_E_INVALID, "cv::remap cannot address more than 32767 rows/cols");
    if (out_row0 < 0 || out_rows < 0 || out_row0 + out_rows > L || src_row0 < 0 || src_rows < 0 || src_row0 + src_rows > L)
        return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_bicubic_u16: row window outside the raster");
    if (!(dy == dy) || !(dx == dx)) return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_bicubic_u16: NaN shift");
    OipShiftGeom g = oip_shift_geom(W, L, dx, dy, section_rows);
    if (g.step <= 0 || g.nsec <= 0) return oip_fail(ctx, OIP_E_INVALID, "shift larger than a remap section");
    if (out_rows == 0) return OIP_OK;

    if (src_row0 != 0 || src_rows != L) {
        long first = 0, last = 0;
        if (oip_remap_shift_src_range(out_row0, out_rows, L, dy, section_rows, &first, &last) != OIP_OK)
            return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_bicubic_u16: bad row window");
        if (first < src_row0 || last > src_row0 + src_rows)
            return oip_fail(ctx, OIP_E_INVALID,
                            "oip_remap_shift_bicubic_u16: source window [%ld,%ld) lacks halo lines, need [%ld,%ld)",
                            src_row0, src_row0 + src_rows, first, last);
    }
    void *ws = nullptr;
    int rc = oip_workspace(ctx, (size_t)out_rows * sizeof(RowInfo), &ws);
    if (rc) return rc;
    RowInfo *rows = reinterpret_cast<RowInfo *>(ws);
    {
        OipProfScope prof(ctx, "shift_rows_kernel");
        int blocks = (int)((out_rows + 255) / 256);
        hipLaunchKernelGGL(shift_rows_kernel, dim3(blocks), dim3(256), 0, ctx->stream, rows, g, out_row0, out_rows,
                           src_row0, src_rows);
    }
    {
        OipProfScope prof(ctx, "remap_shift_kernel");
        int gx = (W + kBlock - 1) / kBlock;
        long want = (long)ctx->cu_count * 16 / gx;
        if (want < 1) want = 1;
        long rpb = (out_rows + want - 1) / want;
        if (rpb < 32) rpb = 32;
        long gy = (out_rows + rpb - 1) / rpb;
        if (gy > 65535) { gy = 65535; rpb = (out_rows + gy - 1) / gy; gy = (out_rows + rpb - 1) / rpb; }
        hipLaunchKernelGGL(remap_shift_kernel, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_src, d_dst, rows,
                           W, out_rows, dx, ctx->d_tab1d, (int)rpb);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}
