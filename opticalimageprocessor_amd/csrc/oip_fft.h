// oip_fft.h -- internal interface of the 2-D FFT engine (fft.hip) used by phasecorr.hip
#pragma once

#include <hip/hip_runtime.h>
#include <cstring>

#include <functional>
#include <vector>

struct oip_ctx;

// Arg-max of a correlation surface without a reduction pass: every workgroup of the last inverse pass
// folds its tile maximum into one of kPeakSlots 64-bit slots per surface with atomicMax.  A slot packs
// (order-preserving bits of the f32 value) << 32 | (0xFFFFFFFF - key), key = row-major index in the
// fftShift-ed image (< 2^32), so the largest value wins and, among equal values, the smallest key -- the
// first maximum in minMaxLoc's scan order.  0 = empty (NaN never enters).  The window kernel reduces the
// slots itself; the centroid kernel zeroes them for the next surface.
constexpr int kPeakSlots = 256;
__host__ __device__ inline unsigned long long oip_peak_pack(float v, long key)
{
    if (v != v) return 0ull;                                   // NaN
    if (v == 0.f) v = 0.f;                                     // -0 == +0 for minMaxLoc
    unsigned u;
    memcpy(&u, &v, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)key);
}
__host__ __device__ inline long oip_peak_key(unsigned long long packed, long none)
{
    return packed == 0ull ? none : (long)(0xFFFFFFFFu - (unsigned)(packed & 0xFFFFFFFFull));
}

// What a pass reads instead of / writes besides the complex array (fusions)
struct OipFftIo {
    int load_kind;      // 0: the complex array; 1: pack two real f32 images (zero padded)
    const float *re;    // f32 image, pitch == cols ...
    const float *im;    // ... may be null
    const uint16_t *re16;   // ... or a u16 raster window (pointer at its first pixel, own pitch)
    const uint16_t *im16;
    long pitch_re16, pitch_im16;
    // ... or an image that is only vertically up-sampled so far: V is rows x v_cols f32 (pitch ==
    // v_cols) and the loader applies the horizontal four cubic taps
    // value(y, x) = sum_j alpha[x][j] * V[y][clamp(xofs[x] - 1 + j)]
    const float *re_v;
    const float *im_v;
    int v_cols;
    const int *xofs;
    const float *alpha; // [cols][4]
    int x4;             // exact x4 geometry (host-verified): xofs[x] == (x - 2) >> 2, no table look-up needed
    int rows, cols;     // extent of the real images
    int store_kind;     // 0: the complex array; 1: peak partials only (nothing stored);
                        // 2: 5x5 window around a known peak (25 workgroups, nothing else stored)
    unsigned long long *slots;  // store_kind 1 and 2: [2][kPeakSlots] arg-max slots (real part, imaginary part)
    long *peak_key;             // store_kind 2: out, shifted key of the peak per part (for the centroid kernel)
    float *window;              // store_kind 2: [25] values, row-major dy,dx (NaN = outside image)
    int part;                   // store_kind 2: number of parts (1: real surface only, 2: real and imaginary), 25 tiles each
};

struct OipFftPass {
    int F;              // sub-transform length of this pass
    int nradix;
    int radix[16];      // Stockham radices, product == F (generic kernel)
    int vshift;         // log2(V): V adjacent transforms per workgroup tile
    int Vp;             // LDS row pitch (V+1, or 1 when V == 1)
    int mode;           // 0: points strided, lanes contiguous; 1: points contiguous
    int axis;           // 0: x (rows), 1: y (columns)
    long nstride;       // mode 0: elements between consecutive points
    long lanes;         // mode 0: lanes in the contiguous direction; mode 1: number of vectors
    int lane_tiles;
    int O1;             // mode 0 outer dimension 1 (count, element stride)
    long o1_stride;
    int O2;
    long o2_stride;
    int tw_mode;        // 0 none; 1: j = lane index; 2: j = o1 index  (twiddle w_T^(j*k))
    int T;
    int S;              // stride of the sub-transform in axis elements (T / F)
    int N;              // row length of the array (to turn offsets into coordinates)
    int M;
    int P;              // row pitch in elements (N rounded up to whole 128-byte lines)
    int inverse;
    long ntiles;        // tiles of this launch (persistent specialised kernels walk them)
    long total_tiles;   // tiles of the whole pass (peak partial slots)
    int lt0, ltn;       // lane-tile window of this launch (column panel); ltn == 0: all
    int fast;           // index of a compile-time specialised kernel, -1: generic
    int grid3;          // mode 0 launched as a (lane tile, o1, o2) grid: the tile needs no divisions to decode
    int dbg;            // experiment mask for the fused-loader kernels (OIP_PACK_DBG): 1 no loads, 2 no stages, 4 no stores
    int xcd_chunk;      // grid3: lane tiles per XCD (grid x = 8 * xcd_chunk >= lane tiles); see decode_tile
    int tw_rows;        // tw_mode 2, specialised kernels: the inter-pass table is [O1][F] (row o1 contiguous) instead of table T gathered at o1 * n
};

struct OipFft2dPlan {
    int M, N;
    int P;                            // row pitch of the complex array, elements
    std::vector<int> xf, yf;          // pass factors per axis, in forward order
    std::vector<OipFftPass> passes;   // forward order: y passes then x passes
    int n_y;
    const int *d_ypos;                // device: row position of frequency line ky in the scrambled spectrum, ky in [0, M)
};

int oip_fft2d_plan(oip_ctx *ctx, int M, int N, const OipFft2dPlan **out);
// forward: io (optional) applies to the FIRST pass' load; inverse: to the LAST pass' store
int oip_fft2d_exec(oip_ctx *ctx, const OipFft2dPlan *plan, float2 *data, int inverse, const OipFftIo *io, int rows_done = 0);
int oip_fft_table(oip_ctx *ctx, int T, const float2 **out);     // exp(-2 pi i t / T), t in [0, T)
// re-run the last inverse pass for the 25 tiles holding the 5x5 window around *peak_key
int oip_fft2d_window(oip_ctx *ctx, const OipFft2dPlan *plan, float2 *data, const OipFftIo *io);

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
    if (a.n == 1) return p;
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
    if (a.n == 1) return k;
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
