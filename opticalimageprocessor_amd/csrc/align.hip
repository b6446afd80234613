// align.hip -- inter-band (PAN<->MSS) bicubic alignment of the 4 MSS bands on gfx950.
//
// Replaces PreProcessor::DoInterBandAlignment, outer (preproc.h:351-425) and inner
// (preproc.h:428-468): per 20000-line section and per band the reference builds two float
// maps from the fitted polynomials, calls cv::remap(INTER_CUBIC, BORDER_CONSTANT), merges
// the four planes with cv::merge and memcpy's the section minus its leading overlap into
// the final 16UC4 image.  Here one kernel does all of it per output pixel:
//   mapX = (float)((cX1*xx + cX0 + xx)/4)                     xx = 4x   (fp64, left to right)
//   mapY = (float)((cY2*xx*xx + cY1*xx + cY0 + yy)/4)         yy = 4y, y section relative
// evaluated in fp64 in registers (never stored), OpenCV's 1/32-px bicubic on each planar
// band, and one 8-byte interleaved store (B0,B1,B2,B3) -- 4 x 2 B read + 8 B written per
// output pixel.  Lines of skipped tail sections are written as zeros (the reference leaves
// them uninitialised, SURVEY App.B-6).
#include "oip_bicubic.h"
#include "oip_geom.h"
#include "oip_internal.h"

#include <climits>
#include <cmath>

namespace {

constexpr int kBlock = 256;

struct AlignRow {
    int valid;      // 0: line of a skipped section -> zeros
    int yrel;       // section-relative line
    int base;       // section's first MSS line relative to d_planes' first line
    int lines;      // section's line count (cv::remap source height)
};

struct AlignCoef {
    double cx[4][2];
    double cy[4][3];
};

__global__ void align_rows_kernel(AlignRow *rows, OipAlignGeom g, long out_row0, long out_rows, long src_row0)
{
    long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= out_rows) return;
    AlignRow a;
    int sec;
    long yrel, off, lines;
    if (oip_align_row(g, out_row0 + r, &sec, &yrel, &off, &lines)) {
        a.valid = 1; a.yrel = (int)yrel; a.base = (int)(off - src_row0); a.lines = (int)lines;
    } else {
        a.valid = 0; a.yrel = 0; a.base = 0; a.lines = 0;
    }
    rows[r] = a;
}

// One lane = one (output column, band): lane 16b+p of a wave handles pixel p, band b, so a wave
// writes 128 contiguous bytes of the interleaved 16UC4 line (as 2-byte stores 8 bytes apart per
// 16-lane group) and keeps a single 4x4 source window per lane (~50 VGPRs instead of ~190 for four
// bands per lane: 8 waves/SIMD instead of 2).
// For a fixed column the first tap line iy advances by exactly one per output line except where
// the f32 rounding of mapY flips (rare) or a section seam restarts the section-relative line,
// so the window normally loads only the newest source line (4 taps per pixel instead of 16).
__device__ __forceinline__ void align_load_row(const uint16_t *__restrict__ pl, long lr, bool yok, int Wb, int cix,
                                               unsigned xmask, float out[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int cj = cix + j;
        cj = cj < 0 ? 0 : (cj > Wb - 1 ? Wb - 1 : cj);
        out[j] = (yok && (xmask & (1u << j))) ? (float)pl[lr * Wb + cj] : 0.f;
    }
}

template <int KB>
__global__ __launch_bounds__(kBlock) void align_mss_kernel(const uint16_t *__restrict__ planes, size_t plane_stride,
                                                           long src_rows, uint16_t *__restrict__ dst,
                                                           const AlignRow *__restrict__ rows, int Wb, long out_rows,
                                                           AlignCoef co, const float *__restrict__ tab1d,
                                                           int rows_per_block)
{
    // lane = 16 * band + pixel: the 2-byte taps of a 16-lane group are 32 contiguous bytes of one
    // plane (band-minor lanes put the four lanes of every quad on four different planes, which the
    // texture addresser serialises -- measured 2.4x slower)
    const int gid = blockIdx.x * kBlock + threadIdx.x;
    const int x = (gid >> 6) * 16 + (gid & 15), b = (gid >> 4) & 3;
    if (x >= Wb) return;
    const int xx = x * 4;
    const double dxx = (double)xx;
    // column-only part of the maps (preproc.h:447-448), fp64, left to right
    // the lane's band picks its coefficients from the kernel arguments (selects, no memory)
    const double cx0 = b == 0 ? co.cx[0][0] : (b == 1 ? co.cx[1][0] : (b == 2 ? co.cx[2][0] : co.cx[3][0]));
    const double cx1 = b == 0 ? co.cx[0][1] : (b == 1 ? co.cx[1][1] : (b == 2 ? co.cx[2][1] : co.cx[3][1]));
    const double cy0 = b == 0 ? co.cy[0][0] : (b == 1 ? co.cy[1][0] : (b == 2 ? co.cy[2][0] : co.cy[3][0]));
    const double cy1 = b == 0 ? co.cy[0][1] : (b == 1 ? co.cy[1][1] : (b == 2 ? co.cy[2][1] : co.cy[3][1]));
    const double cy2 = b == 0 ? co.cy[0][2] : (b == 1 ? co.cy[1][2] : (b == 2 ? co.cy[2][2] : co.cy[3][2]));
    const double mx = __dadd_rn(__dadd_rn(__dmul_rn(cx1, dxx), cx0), dxx) * 0.25;
    const int sx = oip_cvround((float)mx * 32.0f);
    const int cix = oip_sat_short(sx >> 5) - 1;
    const int fx = sx & 31;
    float wx[4];
    unsigned xmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        wx[j] = tab1d[fx * 4 + j];
        const int cj = cix + j;
        if (cj >= 0 && cj < Wb) xmask |= 1u << j;
    }
    const double coly = __dadd_rn(__dadd_rn(__dmul_rn(__dmul_rn(cy2, dxx), dxx), __dmul_rn(cy1, dxx)), cy0);
    const uint16_t *pl = planes + (size_t)b * plane_stride;
    const bool x_out = cix >= Wb || cix + 4 <= 0;
    const bool x_in = (unsigned)cix < (unsigned)(Wb - 3 > 0 ? Wb - 3 : 0);

    float win[4][4];               // [tap row][tap col]
    int cur_iy = INT_MIN, cur_base = INT_MIN;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > out_rows) r1 = out_rows;

    // first-tap line and y phase of output line r for this lane's column and band
    auto map_y = [&](const AlignRow &a, int *iy, int *fy) {
        const double yy = (double)((long)a.yrel * 4);
        const double my = __dadd_rn(coly, yy) * 0.25;
        const int sy = oip_cvround((float)my * 32.0f);
        *iy = oip_sat_short(sy >> 5) - 1;
        *fy = sy & 31;
    };
    // the 16-tap sum for the window as it stands (interior / border / fully outside)
    auto resample = [&](const AlignRow &a, int iy, int fy) -> unsigned {
        float sum;
        if (x_out || iy >= a.lines || iy + 4 <= 0) {
            sum = 0.f;
        } else {
            float wy[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wy[j] = tab1d[fy * 4 + j];
            if (x_in && (unsigned)iy < (unsigned)(a.lines - 3 > 0 ? a.lines - 3 : 0)) {
                sum = oip_bicubic_interior(win, wx, wy);
            } else {
                unsigned ymask = 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = iy + t;
                    const long lr = (long)a.base + rr;
                    if (rr >= 0 && rr < a.lines && lr >= 0 && lr < src_rows) ymask |= 1u << t;
                }
                sum = oip_bicubic_border(win, wx, wy, xmask, ymask);
            }
        }
        return oip_sat_u16(sum);
    };
    // one output line the careful way: any jump of the first tap line reloads what is missing
    auto one_row = [&](long r, const AlignRow &a) {
        unsigned res = 0;
        if (a.valid) {
            if (a.base != cur_base) { cur_base = a.base; cur_iy = INT_MIN; }       // new section: window stale
            int iy, fy;
            map_y(a, &iy, &fy);
            if (iy == cur_iy + 1 && cur_iy != INT_MIN) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) win[t][j] = win[t + 1][j];
                }
                const int rr = iy + 3;
                const long lr = (long)a.base + rr;
                align_load_row(pl, lr, rr >= 0 && rr < a.lines && lr >= 0 && lr < src_rows, Wb, cix, xmask, win[3]);
            } else if (iy != cur_iy) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rr = iy + t;
                    const long lr = (long)a.base + rr;
                    align_load_row(pl, lr, rr >= 0 && rr < a.lines && lr >= 0 && lr < src_rows, Wb, cix, xmask, win[t]);
                }
            }
            cur_iy = iy;
            res = resample(a, iy, fy);
        }
        dst[((size_t)r * Wb + x) * 4 + b] = (uint16_t)res;
    };

    // Lines can be taken KB at a time: in the regular case -- same section, first tap line advancing
    // by one per output line, window already primed -- the new source lines are known before any of
    // them is used, so their loads go out together.  Measured, the extra registers cost more
    // (occupancy) than the shorter dependency chains gain, so KB = 1 is the default.
    if (KB == 0) {
        // One-line lookahead: after the window for output line r is complete, the source line that line
        // r + 1 will need in the regular case (same section, first tap line + 1) is requested BEFORE the 16-tap
        // sum of line r, so its latency hides under ~50 f32 operations instead of stalling the next iteration.
        // A wrong guess (section seam, rounding flip of the map) is detected by (base, line) tags and reloaded.
        float nl[4] = {0.f, 0.f, 0.f, 0.f};
        int nl_iy = INT_MIN, nl_base = INT_MIN;
        for (long r = r0; r < r1; ++r) {
            const AlignRow a = rows[r];
            unsigned res = 0;
            if (a.valid) {
                if (a.base != cur_base) { cur_base = a.base; cur_iy = INT_MIN; }
                int iy, fy;
                map_y(a, &iy, &fy);
                if (iy == cur_iy + 1 && cur_iy != INT_MIN) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) win[t][j] = win[t + 1][j];
                    }
                    if (nl_base == a.base && nl_iy == iy) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) win[3][j] = nl[j];
                    } else {
                        const int rr = iy + 3;
                        const long lr = (long)a.base + rr;
                        align_load_row(pl, lr, rr >= 0 && rr < a.lines && lr >= 0 && lr < src_rows, Wb, cix, xmask, win[3]);
                    }
                } else if (iy != cur_iy) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rr = iy + t;
                        const long lr = (long)a.base + rr;
                        align_load_row(pl, lr, rr >= 0 && rr < a.lines && lr >= 0 && lr < src_rows, Wb, cix, xmask, win[t]);
                    }
                }
                cur_iy = iy;
                {
                    const int rr = iy + 4;
                    const long lr = (long)a.base + rr;
                    align_load_row(pl, lr, rr >= 0 && rr < a.lines && lr >= 0 && lr < src_rows, Wb, cix, xmask, nl);
                    nl_iy = iy + 1;
                    nl_base = a.base;
                }
                res = resample(a, iy, fy);
            }
            dst[((size_t)r * Wb + x) * 4 + b] = (uint16_t)res;
        }
        return;
    }
    constexpr int kBatch = KB > 0 ? KB : 1;
    long r = r0;
    for (; r + kBatch <= r1; r += kBatch) {
        AlignRow a[kBatch];
        int iy[kBatch], fy[kBatch];
        bool regular = cur_iy != INT_MIN;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            a[k] = rows[r + k];
            map_y(a[k], &iy[k], &fy[k]);
            regular = regular && a[k].valid && a[k].base == cur_base && iy[k] == cur_iy + 1 + k;
        }
        if (regular) {
            float nl[kBatch][4];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int rr = iy[k] + 3;
                const long lr = (long)a[k].base + rr;
                align_load_row(pl, lr, rr >= 0 && rr < a[k].lines && lr >= 0 && lr < src_rows, Wb, cix, xmask, nl[k]);
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) win[t][j] = win[t + 1][j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) win[3][j] = nl[k][j];
                cur_iy = iy[k];
                dst[((size_t)(r + k) * Wb + x) * 4 + b] = (uint16_t)resample(a[k], iy[k], fy[k]);
            }
        } else {
#pragma unroll 1
            for (int k = 0; k < kBatch; ++k) one_row(r + k, a[k]);
        }
    }
    for (; r < r1; ++r) one_row(r, rows[r]);
}

// host mirror of the kernel's first-tap line for one band/column/line (range queries)
inline int host_iy(const double *cy3, int x, long yrel)
{
    double dxx = (double)(x * 4);
    double coly = ((cy3[2] * dxx) * dxx + cy3[1] * dxx) + cy3[0];
    double my = (coly + (double)(yrel * 4)) * 0.25;
    int sy = (int)rintf((float)my * 32.0f);
    int iy = sy >> 5;
    iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    return iy - 1;
}

}  // namespace

// MSS source lines needed by a window of output lines.  mapY is quadratic in the column, so
// its extremes over a line are at the ends or at the vertex; they are evaluated per band and
// widened by one line for the f32 rounding of the map.
extern "C" int oip_align_mss_src_range(long out_row0, long out_rows, long Lm, const double *cy, int Wb,
                                       int lines_per_section, int line_offset, int overlap, int keep_leading,
                                       int min_lines, long *first, long *last)
{
    if (!cy || Wb <= 0 || out_rows < 0 || out_row0 < 0) return OIP_E_INVALID;
    OipAlignGeom g = oip_align_geom(Wb, Lm, lines_per_section, line_offset, overlap, keep_leading, min_lines);
    if (out_row0 + out_rows > g.out_rows) return OIP_E_INVALID;
    long lo = -1, hi = -1;
    // candidate columns: ends and the vertex of each band's parabola
    int cand[4][3];
    for (int b = 0; b < 4; ++b) {
        const double *c = cy + b * 3;
        cand[b][0] = 0;
        cand[b][1] = Wb - 1;
        int xv = 0;
        if (c[2] != 0.0) {
            double v = -c[1] / (2.0 * c[2]) / 4.0;
            xv = v < 0 ? 0 : (v > Wb - 1 ? Wb - 1 : (int)v);
        }
        cand[b][2] = xv;
    }
    auto accum = [&](long orow) {
        int sec;
        long yrel, off, lines;
        if (!oip_align_row(g, orow, &sec, &yrel, &off, &lines)) return;
        for (int b = 0; b < 4; ++b)
            for (int k = 0; k < 3; ++k)
                for (int dxc = -1; dxc <= 1; ++dxc) {
                    int x = cand[b][k] + dxc;
                    if (x < 0 || x >= Wb) continue;
                    int iy = host_iy(cy + b * 3, x, yrel);
                    long r_lo = iy - 1, r_hi = iy + 4 + 1;          // widened by one line each side
                    if (r_lo < 0) r_lo = 0;
                    if (r_hi > lines) r_hi = lines;
                    if (r_lo >= r_hi) continue;
                    if (lo < 0 || off + r_lo < lo) lo = off + r_lo;
                    if (off + r_hi > hi) hi = off + r_hi;
                }
    };
    // tap lines are monotone in the output line inside one section: end points suffice
    long a = out_row0, bnd = out_row0 + out_rows;
    long r = a;
    while (r < bnd) {
        int sec;
        long yrel, off, lines;
        if (!oip_align_row(g, r, &sec, &yrel, &off, &lines)) break;   // zero tail from here on
        long sec_end = r + (lines - yrel);                             // first output line of the next section
        if (g.keep_leading && sec == 0 && r < g.overlap) sec_end = g.overlap + (lines - g.overlap);
        long e = sec_end < bnd ? sec_end : bnd;
        accum(r);
        accum(e - 1);
        r = e;
    }
    if (lo < 0) lo = hi = 0;
    if (first) *first = lo;
    if (last) *last = hi;
    return OIP_OK;
}

extern "C" int oip_align_mss_bicubic_u16x4(oip_ctx *ctx, const uint16_t *d_planes, size_t plane_stride, long src_row0,
                                           long src_rows, uint16_t *d_dst, long out_row0, long out_rows, int Wb,
                                           long Lm, const double *cx, const double *cy, int lines_per_section,
                                           int line_offset, int overlap, int keep_leading, int min_lines,
                                           long *rows_valid)
{
    OIP_CHECK_CTX(ctx);
    if (!d_planes || !d_dst || !cx || !cy || Wb <= 0 || Lm <= 0)
        return oip_fail(ctx, OIP_E_INVALID, "oip_align_mss_bicubic_u16x4: bad argument");
    // preproc.h:355-367
    if (overlap > OIP_IBPA_MAX_LINEOVERLAP)
        return oip_fail(ctx, OIP_E_INVALID, "Overlap value %d exceeds maximum allowed value(%d)", overlap, OIP_IBPA_MAX_LINEOVERLAP);
    if (lines_per_section > 32767) return oip_fail(ctx, OIP_E_INVALID, "Row number exceeds OpenCV allowed value");
    if (lines_per_section < overlap * 2 || overlap < 0 || lines_per_section <= overlap)
        return oip_fail(ctx, OIP_E_INVALID, "Lines per section too small or section overlapped lines too large");
    if (Lm - line_offset < min_lines || line_offset < 0)
        return oip_fail(ctx, OIP_E_INVALID, "Too few image lines left to process");
    if (Wb > 32767) return oip_fail(ctx, OIP_E_INVALID, "cv::remap cannot address more than 32767 columns");
    for (int i = 0; i < 8; ++i) if (!(cx[i] == cx[i])) return oip_fail(ctx, OIP_E_INVALID, "NaN polynomial coefficient");
    for (int i = 0; i < 12; ++i) if (!(cy[i] == cy[i])) return oip_fail(ctx, OIP_E_INVALID, "NaN polynomial coefficient");
    OipAlignGeom g = oip_align_geom(Wb, Lm, lines_per_section, line_offset, overlap, keep_leading, min_lines);
    if (rows_valid) *rows_valid = g.rows_valid;
    if (out_row0 < 0 || out_rows < 0 || out_row0 + out_rows > g.out_rows || src_row0 < 0 || src_rows < 0 ||
        src_row0 + src_rows > Lm)
        return oip_fail(ctx, OIP_E_INVALID, "oip_align_mss_bicubic_u16x4: row window outside the raster");
    if (plane_stride < (size_t)src_rows * Wb) return oip_fail(ctx, OIP_E_INVALID, "plane_stride too small");
    if (out_rows == 0) return OIP_OK;
    if (src_row0 != 0 || src_rows != Lm) {
        long first = 0, last = 0;
        int rc = oip_align_mss_src_range(out_row0, out_rows, Lm, cy, Wb, lines_per_section, line_offset, overlap,
                                         keep_leading, min_lines, &first, &last);
        if (rc) return oip_fail(ctx, rc, "oip_align_mss_bicubic_u16x4: bad row window");
        if (first < src_row0 || last > src_row0 + src_rows)
            return oip_fail(ctx, OIP_E_INVALID,
                            "oip_align_mss_bicubic_u16x4: source window [%ld,%ld) lacks halo lines, need [%ld,%ld)",
                            src_row0, src_row0 + src_rows, first, last);
    }
    void *ws = nullptr;
    int rc = oip_workspace(ctx, (size_t)out_rows * sizeof(AlignRow) + 512, &ws);
    if (rc) return rc;
    AlignRow *rows = reinterpret_cast<AlignRow *>(ws);
    {
        OipProfScope prof(ctx, "align_rows_kernel");
        int blocks = (int)((out_rows + 255) / 256);
        hipLaunchKernelGGL(align_rows_kernel, dim3(blocks), dim3(256), 0, ctx->stream, rows, g, out_row0, out_rows, src_row0);
    }
    AlignCoef co;
    memcpy(co.cx, cx, sizeof co.cx);
    memcpy(co.cy, cy, sizeof co.cy);
    {
        OipProfScope prof(ctx, "align_mss_kernel");
        int gx = (Wb + kBlock / 4 - 1) / (kBlock / 4);          // 16 pixels x 4 bands per wave
        long want = (long)ctx->cu_count * 16 / gx;
        if (want < 1) want = 1;
        long rpb = (out_rows + want - 1) / want;
        if (rpb < 32) rpb = 32;
        long gy = (out_rows + rpb - 1) / rpb;
        if (gy > 65535) { gy = 65535; rpb = (out_rows + gy - 1) / gy; gy = (out_rows + rpb - 1) / rpb; }
        static const char *envb = getenv("OIP_ALIGN_BATCH");                 // experiment knob
        const int kb = envb ? atoi(envb) : 0;       // measured on MI355X: 3.9 ms lookahead (0); 4.4 / 5.4 / 6.8 ms for batches of 1 / 2 / 4
        auto fn = kb == 0 ? align_mss_kernel<0> : (kb == 1 ? align_mss_kernel<1> : (kb == 2 ? align_mss_kernel<2> : align_mss_kernel<4>));
        hipLaunchKernelGGL(fn, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_planes, plane_stride,
                           src_rows, d_dst, rows, Wb, out_rows, co, ctx->d_tab1d, (int)rpb);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}
