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
