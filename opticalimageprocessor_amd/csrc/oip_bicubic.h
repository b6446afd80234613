// oip_bicubic.h -- OpenCV-exact 4x4 bicubic accumulation for the remap kernels.
//
// cv::remap(..., INTER_CUBIC, BORDER_CONSTANT, 0) on 16U data is
// remapBicubic<Cast<float,ushort>, float, 1> (OpenCV imgproc/imgwarp.cpp):
//   * the 2-D weight table entry is w[ky*4+kx] = wy[ky] * wx[kx]   (one f32 product)
//   * fully-inside window:   sum  = S00*w0 + S01*w1 + S02*w2 + S03*w3;      (row 0)
//                            sum += S10*w4 + ... ;  sum += row2;  sum += row3;
//     i.e. each row is summed left to right on its own, then added to the running sum
//   * window touching the border: sum = 0, then tap by tap  sum += S*w  for the taps that
//     exist (constant border value 0 contributes nothing and is skipped)
//   * result = saturate_cast<ushort>(cvRound(sum))
// All products and sums are separate f32 roundings (the x86-64 baseline build of OpenCV
// has no FMA), hence the explicit __fmul_rn/__fadd_rn: nothing here may contract.
#pragma once

#include <hip/hip_runtime.h>

__device__ __forceinline__ float oip_row_dot(float s0, float s1, float s2, float s3, float wy, const float *wx)
{
    float w0 = __fmul_rn(wy, wx[0]), w1 = __fmul_rn(wy, wx[1]), w2 = __fmul_rn(wy, wx[2]), w3 = __fmul_rn(wy, wx[3]);
    float r = __fadd_rn(__fmul_rn(s0, w0), __fmul_rn(s1, w1));
    r = __fadd_rn(r, __fmul_rn(s2, w2));
    r = __fadd_rn(r, __fmul_rn(s3, w3));
    return r;
}

// v[ky][kx]: source samples as f32 (zeros where invalid)
__device__ __forceinline__ float oip_bicubic_interior(const float v[4][4], const float *wx, const float *wy)
{
    float sum = oip_row_dot(v[0][0], v[0][1], v[0][2], v[0][3], wy[0], wx);
    sum = __fadd_rn(sum, oip_row_dot(v[1][0], v[1][1], v[1][2], v[1][3], wy[1], wx));
    sum = __fadd_rn(sum, oip_row_dot(v[2][0], v[2][1], v[2][2], v[2][3], wy[2], wx));
    sum = __fadd_rn(sum, oip_row_dot(v[3][0], v[3][1], v[3][2], v[3][3], wy[3], wx));
    return sum;
}

// xmask/ymask: bit i set when tap column/row i exists
__device__ __forceinline__ float oip_bicubic_border(const float v[4][4], const float *wx, const float *wy,
                                                    unsigned xmask, unsigned ymask)
{
    float sum = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        if (!(ymask & (1u << ky))) continue;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            if (!(xmask & (1u << kx))) continue;
            sum = __fadd_rn(sum, __fmul_rn(v[ky][kx], __fmul_rn(wy[ky], wx[kx])));
        }
    }
    return sum;
}

// ---- 8 output pixels per lane, taps as packed pairs (align.hip's and remap.hip's fast kernels) ----------------------------
// These kernels are bound by the issue rate of vector instructions (profiles/experiments/r03_valu_rate.txt and
// r04_valu_rate2.txt give the cycles per instruction), so the 16-tap sums use gfx950's packed f32 instructions (v_pk_mul_f32 /
// v_pk_add_f32: two pixels per issue slot, 4.7 cycles against 2 x 3.1) on operands that ARE register pairs already -- a pair the
// compiler has to assemble costs a v_pk_mov_b32 (5.5 cycles) or two v_mov_b32, which is what its own vectoriser did to the
// scalar form.  A lane's 8 pixels are paired (j, j + 4): the pair's tap kx is then (s[j + kx], s[j + 4 + kx]) for every kx, so
// with the window line held as the seven pairs D[i] = (s[i], s[i + 4]) every operand of every tap is one of them.
typedef float oip_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void oip_expand_pairs(const uint32_t w[6], bool odd, oip_f2 D[7])
{
    // samples start at the low half of w[0] for an even first column, at its high half for an odd one: bring the
    // odd case to the even layout with one funnel shift per dword, then one conversion per sample
    const unsigned sh = odd ? 16u : 0u;
    uint32_t e[6];
#pragma unroll
    for (int i = 0; i < 5; ++i) e[i] = __builtin_amdgcn_alignbit(w[i + 1], w[i], sh);
    e[5] = w[5] >> sh;
    float s[11];
#pragma unroll
    for (int q = 0; q < 11; ++q) s[q] = (q & 1) ? (float)(e[q >> 1] >> 16) : (float)(e[q >> 1] & 0xffffu);
#pragma unroll
    for (int i = 0; i < 7; ++i) { D[i].x = s[i]; D[i].y = s[i + 4]; }
}

// One tap row of the 8 pixels: sum[j] = (pixel j, pixel j + 4).  Per pair 4 packed multiplies and 3 packed adds, each half in
// oip_bicubic_interior's order: products rounded, row sum left to right, rows added in order (-ffp-contract=off keeps the
// compiler from fusing them).
__device__ __forceinline__ void oip_row_taps8(const oip_f2 D[7], const float *w4, bool first, oip_f2 sum[4])
{
    const oip_f2 w0 = {w4[0], w4[0]}, w1 = {w4[1], w4[1]}, w2 = {w4[2], w4[2]}, w3 = {w4[3], w4[3]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        oip_f2 rr = D[j] * w0 + D[j + 1] * w1;
        rr = rr + D[j + 2] * w2;
        rr = rr + D[j + 3] * w3;
        sum[j] = first ? rr : sum[j] + rr;
    }
}

// saturate_cast<ushort> of the 8 sums and their 16 bytes in pixel order.  clamp(cvRound(v), 0, 65535) as three instructions per
// pixel pair less than rndne + cvt + med3 + pack: v + 1.5 * 2^23 rounds v to an integer exactly as rintf does (ties to even;
// |v| < 2^22: the sums of 16 products of 16-bit samples with weights of magnitude < 1.3 are far inside) and leaves it in the low
// bits of the float, 0x4B400000 + cvRound(v); clamping those bits between 0x4B400000 and 0x4B40FFFF as integers clamps the value,
// and the low halves of two results are one v_perm_b32.
__device__ __forceinline__ uint4 oip_sat_pack8(const oip_f2 sum[4])
{
    unsigned c[8];
    const oip_f2 magic = {12582912.0f, 12582912.0f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const oip_f2 t = sum[j] + magic;
        const int lo = (int)__float_as_uint(t.x), hi = (int)__float_as_uint(t.y);
        c[j] = (unsigned)(lo < 0x4B400000 ? 0x4B400000 : (lo > 0x4B40FFFF ? 0x4B40FFFF : lo));          // one v_med3_i32
        c[j + 4] = (unsigned)(hi < 0x4B400000 ? 0x4B400000 : (hi > 0x4B40FFFF ? 0x4B40FFFF : hi));
    }
    uint4 o;
    o.x = __builtin_amdgcn_perm(c[1], c[0], 0x05040100u);
    o.y = __builtin_amdgcn_perm(c[3], c[2], 0x05040100u);
    o.z = __builtin_amdgcn_perm(c[5], c[4], 0x05040100u);
    o.w = __builtin_amdgcn_perm(c[7], c[6], 0x05040100u);
    return o;
}
