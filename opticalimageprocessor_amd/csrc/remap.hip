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

#include <type_traits>
#include <vector>

namespace {

constexpr int kBlock = 256;

// Where output pixel (line r of the call's window, column x of the strip geometry) goes: dst[r * pitch + x + shift], and
// only columns x >= col0 are stored.  The plain call is {W, 0, 0, vec}; the fused prestitch -> stitch form writes the
// columns [fold, W) of the resampled CCD-2 line straight into the right half of the stitched raster (pitch 2 (W - fold),
// shift W - 2 fold): .RRC.PRESTT.RAW is never materialised.  vec: 16-byte stores are aligned (host-checked).
struct DstWin {
    long pitch;
    int col0;
    long shift;
    int vec;
};

struct RowInfo {
    int src[4];     // source line of each vertical tap relative to d_src, -1 = constant border
    int fy;         // y phase
    int flags;      // bit0: all four taps inside the section buffer (interior path in y)
                    // bit1: window entirely above/below the buffer (output 0)
    int pad[2];
};

constexpr int kMaxBadRows = 4096;

__global__ void shift_rows_kernel(RowInfo *rows, OipShiftGeom g, long out_row0, long out_rows, long src_row0,
                                  long src_rows, int *bad_count, int *bad_rows)
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
    if (flags != 1 && bad_count) {          // window touches the section border (or lies outside): fix-up list
        int i = atomicAdd(bad_count, 1);
        if (i < kMaxBadRows) bad_rows[i] = (int)r;
    }
}

// kb != nullptr: the source is the RAW strip and every sample is corrected on load (IMO::InplaceRRC's pixel, exact) -- the
// fused prestitch -> stitch form, where .RRC.RAW of CCD 2 is not materialised either
__device__ __forceinline__ void load_tap_row(const uint16_t *__restrict__ src, int row, int W, const int c[4],
                                             unsigned xmask, float out[4], const double2 *__restrict__ kb)
{
    if (row < 0) {
        out[0] = out[1] = out[2] = out[3] = 0.f;
        return;
    }
    const uint16_t *p = src + (long)row * W;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned v = (xmask & (1u << j)) ? (unsigned)p[c[j]] : 0u;
        if (kb && (xmask & (1u << j))) { const double2 q = kb[c[j]]; v = oip_rrc_px(q.x, q.y, v); }
        out[j] = (float)v;
    }
}

// one output column over a run of output lines (4x4 register window, one new source line per
// output line): the general path -- any width, any alignment, every border case
__device__ __forceinline__ void remap_column(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, DstWin dw,
                             const RowInfo *__restrict__ rows, int W, double dx, const float *__restrict__ tab1d,
                             int x, long r0, long r1, const double2 *__restrict__ kb = nullptr)
{
    if (x < dw.col0) return;
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
            load_tap_row(src, ri.src[3], W, c, xmask, v[3], kb);
        } else if (!(ri.src[0] == cur[0] && ri.src[1] == cur[1] && ri.src[2] == cur[2] && ri.src[3] == cur[3])) {
#pragma unroll
            for (int t = 0; t < 4; ++t) load_tap_row(src, ri.src[t], W, c, xmask, v[t], kb);
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
        dst[r * dw.pitch + x + dw.shift] = (uint16_t)oip_sat_u16(sum);
    }
}


__global__ __launch_bounds__(kBlock) void remap_shift_kernel(const uint16_t *__restrict__ src,
                                                             uint16_t *__restrict__ dst, DstWin dw,
                                                             const RowInfo *__restrict__ rows, int W, long out_rows,
                                                             double dx, const float *__restrict__ tab1d,
                                                             int rows_per_block, const double2 *__restrict__ kb)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= W) return;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > out_rows) r1 = out_rows;
    remap_column(src, dst, dw, rows, W, dx, tab1d, x, r0, r1, kb);
}

// ---- v2: 8 output pixels per lane ------------------------------------------------------------------
// A lane owns output columns x0..x0+7 (one aligned 16-byte store per line).  For a constant
// shift its 8 columns read source columns ix0 .. ix0+10: 11 consecutive u16 = 6 dwords from a
// 4-byte aligned address, one such load per NEW source line (the four tap lines live in a
// rotating register window; the loop is unrolled by 4 so the rotation is static).  When all 8
// columns share the x phase -- the normal case, the phase only changes where the f32 rounding
// of x+dx crosses a binade -- the sixteen 2-D weights wy*wx are shared by the 8 pixels and are
// rebuilt only when the line's y phase changes.  Column groups that are irregular (phase change
// inside the 8, image borders) and lines whose window touches a section border are NOT written
// here: two small fix-up launches of the general per-column code handle them, which keeps this
// kernel free of calls and within 128 VGPRs.  Arithmetic order per pixel is oip_bicubic_interior.
__host__ __device__ inline bool shift_group_regular(int x0, int W, double dx, int *ix0_out, int *fx0_out)
{
    int ix0 = 0, fx0 = 0;
    bool regular = true;
    for (int j = 0; j < 8; ++j) {
        const float mapx = (float)((double)(x0 + j) + dx);
        const int sx = (int)__builtin_rintf(mapx * 32.0f);
        int ix = sx >> 5;
        ix = (ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix)) - 1;
        const int fx = sx & 31;
        if (j == 0) { ix0 = ix; fx0 = fx; }
        else regular = regular && ix == ix0 + j && fx == fx0;
    }
    *ix0_out = ix0;
    *fx0_out = fx0;
    // every tap of every pixel inside the image in x; + 11 (not + 10): the sixth dword a lane reads from an even first column
    // ends one sample past its last tap, and with it inside the line no load of a regular group can leave the buffer
    return regular && ix0 >= 0 && ix0 + 11 < W;
}

// One source line of a lane = 11 consecutive u16 = 6 dwords from a 4-byte aligned address.  The load is split
// from its use: the raw dwords of the NEXT output line's new source line are requested before the current
// line's 16-tap sums, so twice the bytes are in flight per wave (these kernels are bound by memory-level
// parallelism: 1 KiB per wave and line, 16-24 waves per CU, against ~35 KiB per CU that the latency-bandwidth
// product of HBM asks for).
__device__ __forceinline__ void load_raw6(const uint32_t *__restrict__ lane_base, int row, int half_pitch, uint32_t w[6])
{
    // lane_base: the dword holding the lane's first tap column on line 0 (W is even: a line is half_pitch dwords).  A regular
    // group's six dwords stay inside its line (shift_group_regular), so the address is one 64-bit multiply-add with six
    // immediate offsets and nothing is clamped.  Round 3 had dropped the per-dword clamps for all lines but the buffer's last
    // (708 -> 530 vector instructions per line and wave); the remaining two-branch form still made the compiler carry six
    // 64-bit addresses through the merge (84 v_lshl_add_u64 per line).
    const uint32_t *q = lane_base + (long)row * half_pitch;
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = q[i];
}

__global__ __launch_bounds__(kBlock, 4) void remap_shift8_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, DstWin dw,
                                                                 const RowInfo *__restrict__ rows, int W, long out_rows,
                                                                 long src_elems, double dx, const float *__restrict__ tab1d,
                                                                 int rows_per_block)
{
    const int x0 = (blockIdx.x * kBlock + threadIdx.x) * 8;
    if (x0 >= W || x0 + 8 <= dw.col0) return;
    int c0, fx0;
    if (!shift_group_regular(x0, W, dx, &c0, &fx0)) return;        // fix-up launch A
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > out_rows) r1 = out_rows;
    float wx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wx[j] = tab1d[fx0 * 4 + j];

    const uint32_t *lane_base = reinterpret_cast<const uint32_t *>(src) + (c0 >> 1);
    const int half_pitch = W >> 1;
    const bool odd = c0 & 1;
    oip_f2 win[4][7];                             // tap line t at unrolled step k lives in win[(k+t)&3], as sample pairs
    float w2d[16];
    int cur1 = -2, cur2 = -2, cur3 = -2;
    int cur_fy = -1;
    uint32_t nraw[6] = {0u, 0u, 0u, 0u, 0u, 0u};  // raw dwords of source line `nline`, requested one output line ahead
    int nline = -2;
    for (long rb = r0; rb < r1; rb += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long r = rb + k;
            if (r >= r1) break;
            const RowInfo ri = rows[r];
            if (ri.flags != 1) { cur1 = cur2 = cur3 = -2; continue; }      // fix-up launch B
            const bool slide = cur1 != -2 && ri.src[0] == cur1 && ri.src[1] == cur2 && ri.src[2] == cur3;
            if (slide) {
                if (ri.src[3] == nline) oip_expand_pairs(nraw, odd, win[(k + 3) & 3]);
                else { uint32_t w[6]; load_raw6(lane_base, ri.src[3], half_pitch, w); oip_expand_pairs(w, odd, win[(k + 3) & 3]); }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) { uint32_t w[6]; load_raw6(lane_base, ri.src[t], half_pitch, w); oip_expand_pairs(w, odd, win[(k + t) & 3]); }
            }
            cur1 = ri.src[1]; cur2 = ri.src[2]; cur3 = ri.src[3];
            // the line the next output line will add in the regular case (its taps one line further down)
            nline = -2;
            if (r + 1 < r1 && (long)(ri.src[3] + 1) * W < src_elems) {
                nline = ri.src[3] + 1;
                load_raw6(lane_base, nline, half_pitch, nraw);
            }
            if (ri.fy != cur_fy) {
                cur_fy = ri.fy;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const float wy = tab1d[cur_fy * 4 + ky];
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) w2d[ky * 4 + kx] = __fmul_rn(wy, wx[kx]);
                }
            }
            uint4 o;
            {
                oip_f2 sum[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) oip_row_taps8(win[(k + t) & 3], w2d + t * 4, t == 0, sum);
                o = oip_sat_pack8(sum);
            }
            uint16_t *drow = dst + r * dw.pitch + x0 + dw.shift;
            if (x0 >= dw.col0 && dw.vec) {
                *reinterpret_cast<uint4 *>(drow) = o;
            } else {
                // the group that straddles col0, or a destination whose 16-byte stores would be misaligned
                const unsigned d[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (x0 + j >= dw.col0) drow[j] = (uint16_t)(d[j >> 1] >> (16 * (j & 1)));
            }
        }
    }
}

// ---- RRC on load: the RAW strip goes through the workgroup's LDS once -------------------------------------
// The source is CCD 2's RAW strip.  The block's 256 lanes fetch one aligned 16-byte chunk each of every new source line
// (2048 consecutive columns from cbase = 2032 * blockIdx.x + ixmin8, ixmin8 = 8 * floor((floor(dx) - 2) / 8): the taps of
// the block's 254 output groups lie inside whatever the rounding of x + dx does; the host checks it), correct its 8 samples
// with the (k, b) pairs of the lane's own columns held in registers (IMO::InplaceRRC's pixel, exact) and write them to LDS;
// after one barrier the 254 output lanes read their 11 samples from there.  Every sample is corrected ONCE (the register
// form would correct 11 per 8 output pixels and needs the pairs of 11 columns per lane: 6.0 ms with the pairs in LDS,
// profiles/experiments/remap_rrc_on_load.diff.txt), so Stitcher::DoRRC of CCD 2 rides PreStitch's pass and <pan2>.RRC.RAW
// is never written.  Lines are double-buffered in LDS (a write to buffer p follows the barrier of buffer p ^ 1, which
// every reader of the previous use of p has passed); chunks are requested two output lines ahead.  Everything that decides
// a barrier is uniform over the block (the row table and blockIdx); arithmetic per pixel is remap_shift8_kernel's.
// The same staging WITHOUT the correction is slower than remap_shift8_kernel's register window (3.47 against 3.19 ms on
// two 30000 x 100000 segments, at 3 or 4 workgroups per CU): the plain call keeps the register form.
constexpr int kLdsOut = kBlock - 2;

// ---- fp16-accumulate variant (BASELINE config 5: "fp16 accumulate, tolerance stated") ----------------------
// Same geometry, phases, tap positions and border rules as remap_shift8_kernel; only the 16-tap sum of
// the regular interior pixels changes: samples and the sixteen 2-D weights are rounded to fp16 and the sum
// is a chain of packed fp16 FMAs (v_pk_fma_f16: two output pixels per instruction, 16 instructions per
// pixel pair instead of 62 unfused f32 operations).  NOT the parity mode: fp16 carries 11 significant bits.
// Samples enter as (sample - 2048) -- exact integers for 12-bit data -- and the running sum of 16 products
// rounds to 0.5..2 DN steps depending on its magnitude.  Measured against the f32 kernel on the 12-bit
// synthetic strips: tests/test_gpu_config5.py::test_remap_f16acc_tolerance prints max and mean |delta|
// (DESIGN.md section 4.2 records them: max 5, mean 0.25 DN); the bound asserted for arbitrary data is
// |delta| <= 6 + max|sample - 2048| / 64.  The mode is specified for data up to 15 bits (the biased
// sample must fit int16).  Irregular column groups and section-border lines still go through the f32
// fix-up kernels.
typedef _Float16 oip_h2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ oip_h2 h2_from_u16pair(uint32_t w)
{
    oip_h2 r;
    r.x = (_Float16)(unsigned short)(w & 0xffffu);
    r.y = (_Float16)(unsigned short)(w >> 16);
    return r;
}
// (a.y, b.x): the pair one sample further along the line
__device__ __forceinline__ oip_h2 h2_shift(oip_h2 a, oip_h2 b)
{
    const uint32_t ua = __builtin_bit_cast(uint32_t, a), ub = __builtin_bit_cast(uint32_t, b);
    return __builtin_bit_cast(oip_h2, __builtin_amdgcn_alignbit(ub, ua, 16));
}

// one source line as packed fp16 pairs of (sample - kF16Bias): E[i] = (g[2i], g[2i+1]), O[i] = (g[2i+1], g[2i+2]),
// g[q] = sample c0 + q.  The bias is taken off in 16-bit integer arithmetic (exact), so 12-bit data enters fp16
// as integers in [-2048, 2047] -- all exactly representable -- and the partial sums stay small; the bicubic
// weights sum to one, so the bias is added back to the finished sum.
constexpr int kF16Bias = 2048;
__device__ __forceinline__ oip_h2 h2_from_biased_pair(uint32_t w)
{
    oip_h2 r;
    r.x = (_Float16)(short)((w & 0xffffu) - kF16Bias);
    r.y = (_Float16)(short)((w >> 16) - kF16Bias);
    return r;
}
__device__ __forceinline__ void expand_h(const uint32_t w[6], int c0, oip_h2 E[6], oip_h2 O[5])
{
    // odd first column: one funnel shift per dword brings the line to the even layout (as oip_expand_pairs does) -- selecting
    // between the two layouts after the conversion cost ten v_cndmask per line
    const unsigned sh = (c0 & 1) ? 16u : 0u;
#pragma unroll
    for (int i = 0; i < 5; ++i) E[i] = h2_from_biased_pair(__builtin_amdgcn_alignbit(w[i + 1], w[i], sh));
    E[5] = h2_from_biased_pair(w[5] >> sh);       // only its first half (sample 10) is used, through O[4]
#pragma unroll
    for (int i = 0; i < 5; ++i) O[i] = h2_shift(E[i], E[i + 1]);
}
// The 8 fp16 sums (acc[p] = pixels 2p, 2p+1) biased back, saturated and packed.  (float)acc + 2048 is exact in f32 for every
// fp16 value that can round to a different integer than its neighbour (|acc| >= 0.5 has an ulp >= 2^-11; below that the sum
// stays strictly inside (2047.5, 2048.5)), so adding 2048 + 1.5 * 2^23 in one step rounds exactly as rintf((float)acc + 2048)
// does and leaves the integer in the low bits (oip_sat_pack8's trick); the integer clamp maps +inf to 65535 and -inf to 0 as
// the fminf / fmaxf pair it replaces did.  A NaN sum -- not reachable: 16 products of int16 samples with weights whose
// magnitudes add up to 1.6 stay far below fp16's 65504 -- would saturate by its sign bit.
__device__ __forceinline__ uint4 h2_sat_pack8(const oip_h2 acc[4])
{
    unsigned c[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int lo = (int)__float_as_uint((float)acc[p].x + (12582912.0f + (float)kF16Bias));
        const int hi = (int)__float_as_uint((float)acc[p].y + (12582912.0f + (float)kF16Bias));
        c[2 * p] = (unsigned)(lo < 0x4B400000 ? 0x4B400000 : (lo > 0x4B40FFFF ? 0x4B40FFFF : lo));
        c[2 * p + 1] = (unsigned)(hi < 0x4B400000 ? 0x4B400000 : (hi > 0x4B40FFFF ? 0x4B40FFFF : hi));
    }
    uint4 o;
    o.x = __builtin_amdgcn_perm(c[1], c[0], 0x05040100u);
    o.y = __builtin_amdgcn_perm(c[3], c[2], 0x05040100u);
    o.z = __builtin_amdgcn_perm(c[5], c[4], 0x05040100u);
    o.w = __builtin_amdgcn_perm(c[7], c[6], 0x05040100u);
    return o;
}
__device__ __forceinline__ void load_src_line11_h(const uint32_t *__restrict__ lane_base, int row, int half_pitch, int c0,
                                                  oip_h2 E[6], oip_h2 O[5])
{
    uint32_t w[6];
    load_raw6(lane_base, row, half_pitch, w);
    expand_h(w, c0, E, O);
}

__global__ __launch_bounds__(kBlock, 4) void remap_shift8_f16_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, DstWin dw,
                                                                     const RowInfo *__restrict__ rows, int W, long out_rows,
                                                                     long src_elems, double dx, const float *__restrict__ tab1d,
                                                                     int rows_per_block)
{
    const int x0 = (blockIdx.x * kBlock + threadIdx.x) * 8;
    if (x0 >= W || x0 + 8 <= dw.col0) return;
    int c0, fx0;
    if (!shift_group_regular(x0, W, dx, &c0, &fx0)) return;        // fix-up launch A (f32)
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > out_rows) r1 = out_rows;
    float wx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wx[j] = tab1d[fx0 * 4 + j];

    const uint32_t *lane_base = reinterpret_cast<const uint32_t *>(src) + (c0 >> 1);
    const int half_pitch = W >> 1;
    oip_h2 E[4][6], O[4][5];                      // tap line t at unrolled step k lives in slot (k+t)&3
    oip_h2 w2d[16];
    int cur1 = -2, cur2 = -2, cur3 = -2;
    int cur_fy = -1;
    uint32_t nraw[6] = {0u, 0u, 0u, 0u, 0u, 0u};  // one-line lookahead, as in remap_shift8_kernel
    int nline = -2;
    for (long rb = r0; rb < r1; rb += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long r = rb + k;
            if (r >= r1) break;
            const RowInfo ri = rows[r];
            if (ri.flags != 1) { cur1 = cur2 = cur3 = -2; continue; }      // fix-up launch B (f32)
            const bool slide = cur1 != -2 && ri.src[0] == cur1 && ri.src[1] == cur2 && ri.src[2] == cur3;
            if (slide) {
                if (ri.src[3] == nline) expand_h(nraw, c0, E[(k + 3) & 3], O[(k + 3) & 3]);
                else load_src_line11_h(lane_base, ri.src[3], half_pitch, c0, E[(k + 3) & 3], O[(k + 3) & 3]);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) load_src_line11_h(lane_base, ri.src[t], half_pitch, c0, E[(k + t) & 3], O[(k + t) & 3]);
            }
            cur1 = ri.src[1]; cur2 = ri.src[2]; cur3 = ri.src[3];
            nline = -2;
            if (r + 1 < r1 && (long)(ri.src[3] + 1) * W < src_elems) {
                nline = ri.src[3] + 1;
                load_raw6(lane_base, nline, half_pitch, nraw);
            }
            if (ri.fy != cur_fy) {
                cur_fy = ri.fy;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const float wy = tab1d[cur_fy * 4 + ky];
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        const _Float16 h = (_Float16)__fmul_rn(wy, wx[kx]);
                        oip_h2 hh = {h, h};
                        w2d[ky * 4 + kx] = hh;
                    }
                }
            }
            oip_h2 acc[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {             // output pixels 2p, 2p+1
                acc[p] = oip_h2{(_Float16)0.f, (_Float16)0.f};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const oip_h2 *Et = E[(k + t) & 3], *Ot = O[(k + t) & 3];
                    acc[p] = __builtin_elementwise_fma(Et[p], w2d[t * 4 + 0], acc[p]);
                    acc[p] = __builtin_elementwise_fma(Ot[p], w2d[t * 4 + 1], acc[p]);
                    acc[p] = __builtin_elementwise_fma(Et[p + 1], w2d[t * 4 + 2], acc[p]);
                    acc[p] = __builtin_elementwise_fma(Ot[p + 1], w2d[t * 4 + 3], acc[p]);
                }
            }
            const uint4 o = h2_sat_pack8(acc);
            uint16_t *drow = dst + r * dw.pitch + x0 + dw.shift;
            if (x0 >= dw.col0 && dw.vec) {
                *reinterpret_cast<uint4 *>(drow) = o;
            } else {
                // the group that straddles col0, or a destination whose 16-byte stores would be misaligned
                const unsigned d[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (x0 + j >= dw.col0) drow[j] = (uint16_t)(d[j >> 1] >> (16 * (j & 1)));
            }
        }
    }
}

// RRC on load, both accumulate modes (the block comment "RRC on load" above the fp16 helpers describes the kernel)
// IMO::InplaceRRC's pixel with or without the range test of the conversion (SAFE: the caller has proved |k s + b| < 2^31)
template <bool SAFE> __device__ __forceinline__ unsigned rrc_px_t(double k, double b, unsigned s)
{
    if constexpr (!SAFE) return oip_rrc_px(k, b, s);
    else {
        const double v = __dadd_rn(__dmul_rn(k, (double)s), b);
        return (unsigned)(int)v & 0xffffu;
    }
}

// F16: the fp16-accumulate sums of remap_shift8_f16_kernel on the corrected samples.  (The same staging WITHOUT the correction,
// for plain calls, was measured again with this loop: 2.70 ms f32 / 2.65 fp16 against the register-window kernels' 2.55 / 2.70 on
// two 30000 x 100000 segments -- every form of this pass now sits at 4.5-4.8 TB/s, 85 % of what the plain copy kernels reach,
// and the plain calls keep the register form.)
template <bool F16>
__global__ __launch_bounds__(kBlock, 3) void remap_shift8_rrc_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, DstWin dw,
                                                                     const RowInfo *__restrict__ rows, int W, long out_rows,
                                                                     long src_elems, double dx, const float *__restrict__ tab1d,
                                                                     int rows_per_block, const double2 *__restrict__ kb, int ixmin8)
{
    __shared__ uint32_t lds[2][kBlock * 4];
    const int X0 = blockIdx.x * kLdsOut * 8;
    if (X0 + kLdsOut * 8 <= dw.col0) return;                         // nothing of this block is stored (uniform)
    const int cbase = X0 + ixmin8;
    const int cc = cbase + (int)threadIdx.x * 8;                     // the lane's chunk of source columns
    const bool chunk_ok = cc >= 0 && cc + 8 <= W;                    // W % 8 == 0: a chunk is inside the line or outside it
    double2 q[8];
    bool in_range = true;                         // |k| 65535 + |b| < 2^31 for the lane's columns: k s + b converts without the range test
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        q[j] = chunk_ok ? kb[cc + j] : make_double2(0.0, 0.0);
        in_range = in_range && __dadd_rn(__dmul_rn(fabs(q[j].x), 65535.0), fabs(q[j].y)) < 2147483648.0;
    }
    // The double -> uint16_t cast of IMO::InplaceRRC (oip_rrc_px) tests both bounds before it converts: two fp64 compares and a
    // select per sample.  When every pair of the workgroup's columns keeps k s + b inside the int32 range for any 16-bit s --
    // every real LUT does -- the loop is instantiated without the test (same bits: the test could never fire); otherwise with it.
    const bool safe_lut = __syncthreads_and(in_range) != 0;
    const int x0 = X0 + (int)threadIdx.x * 8;
    int c0 = 0, fx0 = 0;
    const bool active = (int)threadIdx.x < kLdsOut && x0 < W && x0 + 8 > dw.col0 && shift_group_regular(x0, W, dx, &c0, &fx0);
    const int d0 = active ? (c0 - cbase) >> 1 : 0;                   // host: 0 <= c0 - cbase, c0 - cbase + 10 < 2048
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > out_rows) r1 = out_rows;
    const long src_lines = src_elems / W;
    float wx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wx[j] = tab1d[fx0 * 4 + j];

    auto body = [&](auto safe_tag) __attribute__((always_inline)) {
    constexpr bool SAFE = decltype(safe_tag)::value;
    oip_f2 win[F16 ? 1 : 4][7];                   // tap line t at step k of a quad lives in slot (k+t)&3, as sample pairs
    oip_h2 E[F16 ? 4 : 1][6], O[F16 ? 4 : 1][5];
    typename std::conditional<F16, oip_h2, float>::type w2d[16];
    auto fetch = [&](long row) __attribute__((always_inline)) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (chunk_ok && row >= 0 && row < src_lines) v = *reinterpret_cast<const uint4 *>(src + row * W + cc);
        return v;
    };
    // correct the lane's chunk, put it into LDS buffer `buf`, and after the barrier read the lane's 11 samples into window slot
    // `slot`.  Both indices are compile-time constants at every call (see the buffer rule at `prime`).
    auto stage = [&](uint4 v, int slot, int buf) __attribute__((always_inline)) {
        v.x = rrc_px_t<SAFE>(q[0].x, q[0].y, v.x & 0xffffu) | (rrc_px_t<SAFE>(q[1].x, q[1].y, v.x >> 16) << 16);
        v.y = rrc_px_t<SAFE>(q[2].x, q[2].y, v.y & 0xffffu) | (rrc_px_t<SAFE>(q[3].x, q[3].y, v.y >> 16) << 16);
        v.z = rrc_px_t<SAFE>(q[4].x, q[4].y, v.z & 0xffffu) | (rrc_px_t<SAFE>(q[5].x, q[5].y, v.z >> 16) << 16);
        v.w = rrc_px_t<SAFE>(q[6].x, q[6].y, v.w & 0xffffu) | (rrc_px_t<SAFE>(q[7].x, q[7].y, v.w >> 16) << 16);
        reinterpret_cast<uint4 *>(lds[buf])[threadIdx.x] = v;
        __syncthreads();
        if (active) {
            uint32_t w[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) w[i] = lds[buf][d0 + i];
            if constexpr (F16) expand_h(w, c0, E[slot], O[slot]);
            else oip_expand_pairs(w, c0 & 1, win[slot]);
        }
    };
    // Everything that steers the loop is wave-uniform and lives in scalar registers: the row table comes in through scalar loads
    // (oip_uniform) and the state below only ever takes values from it.  (The form this replaces -- a nested run / unrolled-by-4
    // loop left through breaks, with the inactive lanes skipping the sums through `continue` -- made the compiler keep the line
    // counter, the window state and a loop-state variable in VECTOR registers and rotate a three-deep chunk queue with 16 moves
    // per line: about 100 of the 1500 issue cycles of a line.)
    int cur_fy = -1;
    int cur1 = -2, cur2 = -2, cur3 = -2;          // the three newest tap lines in the window, when `primed`
    int qline = -2;                               // source line whose chunk is in qa; qb, qc, qd hold the next three
    bool primed = false;                          // the window is at step 0 of a quad
    uint4 qa = make_uint4(0u, 0u, 0u, 0u), qb = qa, qc = qa, qd = qa;
    // LDS buffer rule: a write to buffer b must follow the barrier of a stage into b ^ 1 (every reader of the previous use of b
    // has passed it).  A quad stages into 1, 0, 1, 0 and prime into 0, 1, 0 behind a barrier of its own, so whatever ran before,
    // the next stage call is allowed to write where it does.
    auto prime = [&](const RowInfo &a) __attribute__((always_inline)) {
        __syncthreads();
        const uint4 f0 = fetch(a.src[0]), f1 = fetch(a.src[1]), f2 = fetch(a.src[2]);
        qline = a.src[3];
        qa = fetch(qline); qb = fetch((long)qline + 1); qc = fetch((long)qline + 2); qd = fetch((long)qline + 3);
        stage(f0, 0, 0);
        stage(f1, 1, 1);
        stage(f2, 2, 0);
        cur1 = a.src[0]; cur2 = a.src[1]; cur3 = a.src[2];
        primed = true;
    };
    // output line rr_ at step k of a quad: stage its newest tap line (in `qk`), request the chunk four lines on into the same
    // register, then the 16-tap sums of the active lanes
    auto line = [&](auto ktag, const RowInfo &ri, long rr_, uint4 &qk) __attribute__((always_inline)) {
        constexpr int k = decltype(ktag)::value;
        stage(qk, (k + 3) & 3, (k + 1) & 1);
        qk = fetch((long)qline + 4);
        qline += 1;
        cur1 = ri.src[1]; cur2 = ri.src[2]; cur3 = ri.src[3];
        if (active) {
            if (ri.fy != cur_fy) {
                cur_fy = ri.fy;
#pragma unroll
                for (int ky = 0; ky < 4; ++ky) {
                    const float wy = tab1d[cur_fy * 4 + ky];
#pragma unroll
                    for (int kx = 0; kx < 4; ++kx) {
                        if constexpr (F16) {
                            const _Float16 h = (_Float16)__fmul_rn(wy, wx[kx]);
                            w2d[ky * 4 + kx] = oip_h2{h, h};
                        } else {
                            w2d[ky * 4 + kx] = __fmul_rn(wy, wx[kx]);
                        }
                    }
                }
            }
            uint4 o;
            if constexpr (F16) {
                oip_h2 acc[4];
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {          // output pixels 2pp, 2pp+1 (remap_shift8_f16_kernel's sums)
                    acc[pp] = oip_h2{(_Float16)0.f, (_Float16)0.f};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const oip_h2 *Et = E[(k + t) & 3], *Ot = O[(k + t) & 3];
                        acc[pp] = __builtin_elementwise_fma(Et[pp], w2d[t * 4 + 0], acc[pp]);
                        acc[pp] = __builtin_elementwise_fma(Ot[pp], w2d[t * 4 + 1], acc[pp]);
                        acc[pp] = __builtin_elementwise_fma(Et[pp + 1], w2d[t * 4 + 2], acc[pp]);
                        acc[pp] = __builtin_elementwise_fma(Ot[pp + 1], w2d[t * 4 + 3], acc[pp]);
                    }
                }
                o = h2_sat_pack8(acc);
            } else {
                oip_f2 sum[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) oip_row_taps8(win[(k + t) & 3], w2d + t * 4, t == 0, sum);
                o = oip_sat_pack8(sum);
            }
            uint16_t *drow = dst + rr_ * dw.pitch + x0 + dw.shift;
            if (x0 >= dw.col0 && dw.vec) {
                *reinterpret_cast<uint4 *>(drow) = o;
            } else {
                const unsigned d[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (x0 + j >= dw.col0) drow[j] = (uint16_t)(d[j >> 1] >> (16 * (j & 1)));
            }
        }
    };
    auto follows = [](const RowInfo &n, const RowInfo &p) __attribute__((always_inline)) {     // n's window = p's, one line down
        return n.flags == 1 && n.src[0] == p.src[1] && n.src[1] == p.src[2] && n.src[2] == p.src[3] && n.src[3] == p.src[3] + 1;
    };
    long r = r0;
    while (r < r1) {
        const RowInfo a = rows[oip_uniform(r)];
        if (a.flags != 1) { ++r; primed = false; continue; }             // fix-up launch B
        if (r + 4 <= r1) {
            // a quad: four consecutive regular lines whose windows slide by one source line each -- all but a handful of lines
            const RowInfo b = rows[oip_uniform(r + 1)], c = rows[oip_uniform(r + 2)], d = rows[oip_uniform(r + 3)];
            if (follows(b, a) && follows(c, b) && follows(d, c)) {
                if (!(primed && a.src[0] == cur1 && a.src[1] == cur2 && a.src[2] == cur3 && a.src[3] == qline)) prime(a);
                line(std::integral_constant<int, 0>{}, a, r, qa);
                line(std::integral_constant<int, 1>{}, b, r + 1, qb);
                line(std::integral_constant<int, 2>{}, c, r + 2, qc);
                line(std::integral_constant<int, 3>{}, d, r + 3, qd);
                r += 4;
                continue;
            }
        }
        // a line that is not part of a quad (before a section border, the last lines of the block): window staged afresh
        prime(a);
        line(std::integral_constant<int, 0>{}, a, r, qa);
        primed = false;
        ++r;
    }
    };
    if (safe_lut) body(std::true_type{}); else body(std::false_type{});
}


// fix-up A: irregular 8-column groups, all lines.  blockIdx.x = index into `groups`; the 256
// lanes are 8 columns x 32 line sub-ranges.
__global__ __launch_bounds__(kBlock) void remap_fix_cols_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, DstWin dw,
                                                                const RowInfo *__restrict__ rows, int W, long out_rows,
                                                                double dx, const float *__restrict__ tab1d,
                                                                const int *__restrict__ groups, int rows_per_block,
                                                                const double2 *__restrict__ kb)
{
    const int x = groups[blockIdx.x] * 8 + (threadIdx.x & 7);
    if (x >= W) return;
    const int sub = threadIdx.x >> 3;
    const long per = (rows_per_block + 31) / 32;
    long r0 = (long)blockIdx.y * rows_per_block + sub * per;
    long r1 = r0 + per;
    const long rend = (long)(blockIdx.y + 1) * rows_per_block;
    if (r1 > rend) r1 = rend;
    if (r1 > out_rows) r1 = out_rows;
    if (r0 < r1) remap_column(src, dst, dw, rows, W, dx, tab1d, x, r0, r1, kb);
}

// fix-up B: lines whose window touches a section border, all columns
__global__ __launch_bounds__(kBlock) void remap_fix_rows_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, DstWin dw,
                                                                const RowInfo *__restrict__ rows, int W, double dx,
                                                                const float *__restrict__ tab1d,
                                                                const int *__restrict__ bad_count,
                                                                const int *__restrict__ bad_rows, const double2 *__restrict__ kb)
{
    int n = *bad_count;
    if (n > kMaxBadRows) n = kMaxBadRows;
    const int x = blockIdx.x * kBlock + threadIdx.x;
    if ((int)blockIdx.y >= n || x >= W) return;
    const long r = bad_rows[blockIdx.y];
    remap_column(src, dst, dw, rows, W, dx, tab1d, x, r, r + 1, kb);
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

static int remap_shift_impl(oip_ctx *ctx, const uint16_t *d_src, long src_row0, long src_rows, uint16_t *d_dst,
                            long out_row0, long out_rows, int W, long L, double dx, double dy, int section_rows,
                            int row_guard, bool f16acc, long dst_pitch = 0, int dst_col0 = 0, long dst_col_off = 0, const double *d_kb = nullptr)
{
    const double2 *kb = reinterpret_cast<const double2 *>(d_kb);
    OIP_CHECK_CTX(ctx);
    if (dst_pitch <= 0) { dst_pitch = W; dst_col0 = 0; dst_col_off = 0; }
    if (dst_col0 < 0 || dst_col0 >= W || dst_col_off < 0 || dst_col_off + (W - dst_col0) > dst_pitch)
        return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_bicubic_u16: destination window outside its raster");
    DstWin dw;
    dw.pitch = dst_pitch;
    dw.col0 = dst_col0;
    dw.shift = dst_col_off - dst_col0;
    dw.vec = (dw.shift % 8 == 0) && (dst_pitch % 8 == 0) && ((((uintptr_t)d_dst) & 15) == 0);
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
    bool v8 = W % 8 == 0 && (((uintptr_t)d_dst) & 1) == 0 && (((uintptr_t)d_src) & 3) == 0 && src_rows * (long)W >= 16 &&
              (dw.vec || dst_pitch != W);             // the plain call keeps its rule: misaligned destination -> generic kernel
    // RRC on load: the LDS-staged kernel fetches 16-byte chunks of the source lines; otherwise the general kernel (it
    // corrects on load as well)
    const bool lds = v8 && kb && (((uintptr_t)d_src) & 15) == 0;
    if (kb && !lds && f16acc)
        return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_remap_shift_rrc_bicubic_u16_window: the fp16-accumulate form needs W % 8 == 0 and a 16-byte aligned source");
    if (kb && !lds) v8 = false;
    const long fl = (long)floor(dx);
    const int ixmin8 = (int)(8 * ((fl - 2 >= 0 ? fl - 2 : fl - 2 - 7) / 8));
    // irregular 8-column groups (host arithmetic identical to the kernel's)
    std::vector<int> bad_groups;
    if (v8)
        for (int gidx = 0; gidx < W / 8; ++gidx) {
            int a0, f0;
            if (!shift_group_regular(gidx * 8, W, dx, &a0, &f0)) { bad_groups.push_back(gidx); continue; }
            if (lds) {                                // the taps of a regular group lie inside its block's 2048 staged columns
                const int li = a0 - (gidx / kLdsOut * kLdsOut * 8 + ixmin8);
                if (li < 0 || li + 10 >= kBlock * 8) return oip_fail(ctx, OIP_E_RUNTIME, "remap: staged columns do not cover group %d", gidx);
            }
        }
    const size_t rows_bytes = ((size_t)out_rows * sizeof(RowInfo) + 255) / 256 * 256;
    const size_t list_bytes = 256 + sizeof(int) * (size_t)kMaxBadRows + sizeof(int) * (bad_groups.size() + 64);
    void *ws = nullptr;
    int rc = oip_workspace(ctx, rows_bytes + list_bytes, &ws);
    if (rc) return rc;
    RowInfo *rows = reinterpret_cast<RowInfo *>(ws);
    int *d_bad_count = reinterpret_cast<int *>((char *)ws + rows_bytes);
    int *d_bad_rows = d_bad_count + 64;
    int *d_bad_groups = d_bad_rows + kMaxBadRows;
    if (v8) {
        ctx->prof_chain = nullptr;
        OIP_HIP(ctx, hipMemsetAsync(d_bad_count, 0, sizeof(int), ctx->stream));
        if (!bad_groups.empty())
            OIP_HIP(ctx, hipMemcpyAsync(d_bad_groups, bad_groups.data(), sizeof(int) * bad_groups.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    {
        OipProfScope prof(ctx, "shift_rows_kernel");
        int blocks = (int)((out_rows + 255) / 256);
        hipLaunchKernelGGL(shift_rows_kernel, dim3(blocks), dim3(256), 0, ctx->stream, rows, g, out_row0, out_rows,
                           src_row0, src_rows, v8 ? d_bad_count : (int *)nullptr, d_bad_rows);
    }
    if (v8 && (long)g.nsec * 8 + g.ucut + g.bcut + 16 > kMaxBadRows)
        return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_remap_shift_bicubic_u16: too many section-border lines");
    if (v8) {
        int gx = lds ? (W / 8 + kLdsOut - 1) / kLdsOut : (W / 8 + kBlock - 1) / kBlock;
        // workgroups per CU: 5..48 measured again in round 4 (profiles/experiments/r04_grid_sweep.txt): flat within 3 %
        long want = (long)ctx->cu_count * (lds ? 12 : 16) / gx;
        if (want < 1) want = 1;
        long rpb = (out_rows + want - 1) / want;
        if (rpb < 32) rpb = 32;
        rpb = (rpb + 3) / 4 * 4;
        long gy = (out_rows + rpb - 1) / rpb;
        if (gy > 65535) { gy = 65535; rpb = ((out_rows + gy - 1) / gy + 3) / 4 * 4; gy = (out_rows + rpb - 1) / rpb; }
        {
            OipProfScope prof(ctx, lds ? (f16acc ? "remap_shift8_rrc_f16_kernel" : "remap_shift8_rrc_kernel") : (f16acc ? "remap_shift8_f16_kernel" : "remap_shift8_kernel"));
            if (lds && f16acc)
                hipLaunchKernelGGL(remap_shift8_rrc_kernel<true>, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_src, d_dst, dw, rows, W,
                                   out_rows, src_rows * (long)W, dx, ctx->d_tab1d, (int)rpb, kb, ixmin8);
            else if (lds)
                hipLaunchKernelGGL(remap_shift8_rrc_kernel<false>, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_src, d_dst, dw, rows, W,
                                   out_rows, src_rows * (long)W, dx, ctx->d_tab1d, (int)rpb, kb, ixmin8);
            else if (f16acc)
                hipLaunchKernelGGL(remap_shift8_f16_kernel, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_src, d_dst, dw, rows, W, out_rows,
                                   src_rows * (long)W, dx, ctx->d_tab1d, (int)rpb);
            else
                hipLaunchKernelGGL(remap_shift8_kernel, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_src, d_dst, dw, rows, W, out_rows,
                                   src_rows * (long)W, dx, ctx->d_tab1d, (int)rpb);
        }
        if (!bad_groups.empty()) {
            OipProfScope prof(ctx, "remap_fix_cols_kernel");
            long rpb2 = 2048;
            long gy2 = (out_rows + rpb2 - 1) / rpb2;
            if (gy2 > 65535) { gy2 = 65535; rpb2 = (out_rows + gy2 - 1) / gy2; gy2 = (out_rows + rpb2 - 1) / rpb2; }
            hipLaunchKernelGGL(remap_fix_cols_kernel, dim3((unsigned)bad_groups.size(), (unsigned)gy2), dim3(kBlock), 0,
                               ctx->stream, d_src, d_dst, dw, rows, W, out_rows, dx, ctx->d_tab1d, d_bad_groups, (int)rpb2, kb);
        }
        {
            OipProfScope prof(ctx, "remap_fix_rows_kernel");
            const long nb = (long)g.nsec * 8 + g.ucut + g.bcut + 16;      // upper bound on listed lines
            hipLaunchKernelGGL(remap_fix_rows_kernel, dim3((W + kBlock - 1) / kBlock, (unsigned)nb), dim3(kBlock), 0, ctx->stream,
                               d_src, d_dst, dw, rows, W, dx, ctx->d_tab1d, d_bad_count, d_bad_rows, kb);
        }
    } else {
        OipProfScope prof(ctx, "remap_shift_kernel");
        int gx = (W + kBlock - 1) / kBlock;
        long want = (long)ctx->cu_count * 16 / gx;
        if (want < 1) want = 1;
        long rpb = (out_rows + want - 1) / want;
        if (rpb < 32) rpb = 32;
        long gy = (out_rows + rpb - 1) / rpb;
        if (gy > 65535) { gy = 65535; rpb = (out_rows + gy - 1) / gy; gy = (out_rows + rpb - 1) / rpb; }
        hipLaunchKernelGGL(remap_shift_kernel, dim3(gx, (unsigned)gy), dim3(kBlock), 0, ctx->stream, d_src, d_dst, dw, rows,
                           W, out_rows, dx, ctx->d_tab1d, (int)rpb, kb);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

extern "C" int oip_remap_shift_bicubic_u16(oip_ctx *ctx, const uint16_t *d_src, long src_row0, long src_rows,
                                           uint16_t *d_dst, long out_row0, long out_rows, int W, long L, double dx,
                                           double dy, int section_rows, int row_guard)
{
    return remap_shift_impl(ctx, d_src, src_row0, src_rows, d_dst, out_row0, out_rows, W, L, dx, dy, section_rows, row_guard, false);
}

extern "C" int oip_remap_shift_bicubic_u16_f16acc(oip_ctx *ctx, const uint16_t *d_src, long src_row0, long src_rows,
                                                  uint16_t *d_dst, long out_row0, long out_rows, int W, long L, double dx,
                                                  double dy, int section_rows, int row_guard)
{
    return remap_shift_impl(ctx, d_src, src_row0, src_rows, d_dst, out_row0, out_rows, W, L, dx, dy, section_rows, row_guard, true);
}

// PreStitch's resampled CCD-2 lines written into a window of another raster: column x >= dst_col0 of output line r goes to
// d_dst[r * dst_pitch + dst_col_off + (x - dst_col0)], columns below dst_col0 are not stored.  With dst_pitch = 2 (W - fold),
// dst_col0 = fold, dst_col_off = W - fold this is the right half of IMO::StitchBigRaw's output line (imageop.h:340-351):
// prestitch -> stitch without materialising .RRC.PRESTT.RAW (the fused single-pass pipeline).
extern "C" int oip_remap_shift_bicubic_u16_window(oip_ctx *ctx, const uint16_t *d_src, long src_row0, long src_rows,
                                                  uint16_t *d_dst, long dst_pitch, int dst_col0, long dst_col_off, long out_row0,
                                                  long out_rows, int W, long L, double dx, double dy, int section_rows, int row_guard,
                                                  int f16acc)
{
    if (ctx && dst_pitch <= 0) return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_bicubic_u16_window: bad destination pitch");
    return remap_shift_impl(ctx, d_src, src_row0, src_rows, d_dst, out_row0, out_rows, W, L, dx, dy, section_rows, row_guard, f16acc != 0,
                            dst_pitch, dst_col0, dst_col_off);
}

// The same with the source being the RAW CCD-2 strip: every sample is corrected on load (IMO::InplaceRRC's pixel, exact;
// d_kb: the W (k, b) pairs), so Stitcher::DoRRC of CCD 2, PreStitch and the right half of StitchBigRaw are ONE pass over the
// strip and <pan2>.RRC.RAW is not materialised either.  Bits are those of RRC followed by the plain call in the same
// accumulate mode (f16acc != 0: the fp16-accumulate sums; that mode needs a 16-byte aligned source and W % 8 == 0).
extern "C" int oip_remap_shift_rrc_bicubic_u16_window(oip_ctx *ctx, const uint16_t *d_src_raw, long src_row0, long src_rows, const double *d_kb,
                                                      uint16_t *d_dst, long dst_pitch, int dst_col0, long dst_col_off, long out_row0,
                                                      long out_rows, int W, long L, double dx, double dy, int section_rows, int row_guard,
                                                      int f16acc)
{
    if (ctx && (dst_pitch <= 0 || !d_kb)) return oip_fail(ctx, OIP_E_INVALID, "oip_remap_shift_rrc_bicubic_u16_window: bad argument");
    return remap_shift_impl(ctx, d_src_raw, src_row0, src_rows, d_dst, out_row0, out_rows, W, L, dx, dy, section_rows, row_guard, f16acc != 0,
                            dst_pitch, dst_col0, dst_col_off, d_kb);
}
