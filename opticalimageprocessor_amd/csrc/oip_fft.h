// oip_fft.h -- internal interface of the 2-D FFT engine (fft.hip) used by phasecorr.hip
#pragma once

#include <hip/hip_runtime.h>

#include <functional>
#include <vector>

struct oip_ctx;

struct OipFftPass {
    int F;              // sub-transform length of this pass
    int nradix;
    int radix[16];      // Stockham radices, product == F
    int vshift;         // log2(V): V adjacent transforms per workgroup tile
    int Vp;             // LDS row pitch (V+1, or 1 when V == 1)
    int mode;           // 0: points strided, lanes contiguous; 1: points contiguous
    long nstride;       // mode 0: elements between consecutive points
    long lanes;         // mode 0: lanes in the contiguous direction; mode 1: number of vectors
    int lane_tiles;
    int O1;             // mode 0 outer dimension 1 (count, element stride)
    long o1_stride;
    int O2;
    long o2_stride;
    int tw_mode;        // 0 none; 1: j = lane index; 2: j = o1 index  (twiddle w_T^(j*k))
    int T;
    int inverse;
};

struct OipFft2dPlan {
    int M, N;
    std::vector<int> xf, yf;          // pass factors per axis, in forward order
    std::vector<OipFftPass> passes;   // forward order: x passes then y passes
    int n_x;
};

int oip_fft2d_plan(oip_ctx *ctx, int M, int N, const OipFft2dPlan **out);
int oip_fft2d_exec(oip_ctx *ctx, const OipFft2dPlan *plan, float2 *data, int inverse);

// position <-> frequency of one axis after the forward transform.  With factors
// (F1, F2, ..) position p = k1*(F2*F3..) + k2*(F3..) + .. holds frequency
// k = k1 + F1*(k2 + F2*(k3 ..)).
struct OipAxisDigits {
    int n;            // number of factors (<= 4)
    int f[4];
    int L;
};

__host__ __device__ inline int oip_pos_to_freq(const OipAxisDigits &a, int p)
{
    int d[4];
    int rem = p;
    int stride = a.L;
    for (int i = 0; i < a.n; ++i) {
        stride /= a.f[i];
        d[i] = rem / stride;
        rem -= d[i] * stride;
    }
    int k = 0;
    for (int i = a.n - 1; i >= 0; --i) k = k * a.f[i] + d[i];
    return k;
}

__host__ __device__ inline int oip_freq_to_pos(const OipAxisDigits &a, int k)
{
    int p = 0;
    int stride = a.L;
    int rem = k;
    for (int i = 0; i < a.n; ++i) {
        stride /= a.f[i];
        int d = rem % a.f[i];
        rem /= a.f[i];
        p += d * stride;
    }
    return p;
}
