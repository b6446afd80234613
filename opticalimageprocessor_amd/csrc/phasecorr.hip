// phasecorr.hip -- cv::phaseCorrelate and its two drivers on gfx950.
//
// Replaces the loop bodies of Stitcher::CalcSttParameters (stitcher.h:166-191) and
// PreProcessor::CalcInterBandCorrelation (preproc.h:251-329) including the OpenCV calls they
// make: Mat1w->Mat1f window conversion, cv::resize(INTER_CUBIC) x4 up-sampling of the MSS
// window (preproc.h:302-307) and cv::phaseCorrelate (stitcher.h:180, preproc.h:316).
//
// cv::phaseCorrelate (OpenCV imgproc/phasecorr.cpp), restated:
//   pad to getOptimalDFTSize with zeros (bottom/right) -> F1 = dft(a), F2 = dft(b)
//   P = F1 conj(F2)            (mulSpectrums, f32)
//   Pm = |P|                   (magSpectrums; the purely real bins store P*P instead)
//   C = P Pm / (Pm^2 + eps)    (divSpectrums: f32 formula in the row body, fp64 formula in the
//                               first/last column, C = P/(P*P+eps) in the purely real bins)
//   c = idft(C) unscaled -> fftShift -> first maximum -> 5x5 weighted centroid (fp64)
//   response = sum(5x5)/(M N);  shift = (N/2 - cx, M/2 - cy)
// Here two real images ride one complex FFT (fft.hip): for z = a + i b the spectra are
//   A(k) = (Z(k) + conj Z(-k))/2,   B(k) = (Z(k) - conj Z(-k))/(2i)
// and two correlation surfaces ride one inverse FFT as Y = C1 + i C2.  The FFT round-off
// differs from OpenCV's own DFT, so shifts agree to ~1e-4 px, not bitwise (parity unpinned:
// OpenCV is not in the reference tree).
#include "oip_fft.h"
#include "oip_fft_dev.h"
#include "oip_internal.h"

#include <cfloat>
#include <cmath>

namespace {

constexpr int kBlock = 256;

// ---- window / resize -----------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void window_u16_to_f32_kernel(const uint16_t *__restrict__ img, size_t pitch,
                                                                   long row0, int col0, int rows, int cols,
                                                                   float *__restrict__ out)
{
    long i = (long)blockIdx.x * kBlock + threadIdx.x;
    const long n = (long)rows * cols;
    const long stride = (long)gridDim.x * kBlock;
    for (; i < n; i += stride) {
        long y = i / cols;
        int x = (int)(i - y * cols);
        out[i] = (float)img[(size_t)(row0 + y) * pitch + col0 + x];
    }
}

// cv::resize(f32, INTER_CUBIC) (OpenCV imgproc/resize.cpp).  The coefficient set-up is done on
// the host exactly as cv::hal::resize does it -- fx = (float)((dx+0.5)*scale - 0.5),
// sx = floor(fx), fx -= sx, interpolateCubic(fx) -- and cached per shape; the kernel is the
// HResizeCubic + VResizeCubic pair: 4 horizontal taps (edge replicated) into an f32 row value,
// then 4 vertical taps, every product and sum a separate f32 rounding.
template <typename SrcT>
__global__ __launch_bounds__(kBlock) void resize_cubic_kernel(const SrcT *__restrict__ src, long spitch, int sw, int sh,
                                                              float *__restrict__ dst, int dw, int dh,
                                                              const int *__restrict__ xofs, const float4 *__restrict__ alpha,
                                                              const int *__restrict__ yofs, const float4 *__restrict__ beta)
{
    const int dx = blockIdx.x * kBlock + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= dw) return;
    const int sx = xofs[dx], sy = yofs[dy];
    const float4 a = alpha[dx], b = beta[dy];
    int cx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = sx - 1 + j;
        cx[j] = c < 0 ? 0 : (c > sw - 1 ? sw - 1 : c);
    }
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int yk = sy - 1 + k;
        yk = yk < 0 ? 0 : (yk > sh - 1 ? sh - 1 : yk);
        const SrcT *S = src + (size_t)yk * spitch;
        float v = __fmul_rn((float)S[cx[0]], a.x);
        v = __fadd_rn(v, __fmul_rn((float)S[cx[1]], a.y));
        v = __fadd_rn(v, __fmul_rn((float)S[cx[2]], a.z));
        v = __fadd_rn(v, __fmul_rn((float)S[cx[3]], a.w));
        r[k] = v;
    }
    float o = __fmul_rn(r[0], b.x);
    o = __fadd_rn(o, __fmul_rn(r[1], b.y));
    o = __fadd_rn(o, __fmul_rn(r[2], b.z));
    o = __fadd_rn(o, __fmul_rn(r[3], b.w));
    dst[(size_t)dy * dw + dx] = o;
}

// Vertical half of the cubic up-sampling only: V[dy][x] = sum_k beta[dy][k] * S[clamp(yofs[dy] - 1 + k)][x].
// The horizontal half is applied by the loader of the first FFT pass (OipFftIo::re_v / im_v), so the
// fully up-sampled image -- 16x the band window -- never exists in memory; V is 4x the window, written
// once and read once.  cv::resize interpolates horizontally first; doing the vertical taps first
// changes the f32 rounding of an output by an ulp or so, far inside what the FFT's own round-off
// already does to the correlation (the separate oip_resize_cubic_f32 entry keeps cv::resize's order
// bit for bit).
constexpr int kVRows = 4;       // output rows per thread of the vertical pass (they share most source rows)
constexpr int kVBatch = 8;      // images per launch (the four bands of two units)
template <typename SrcT> struct VBatch {
    const SrcT *src[kVBatch];
    float *dst[kVBatch];
    long spitch[kVBatch];       // source pitch per image (raster windows and compact unit windows mix in one batch)
};
template <typename SrcT>
__global__ __launch_bounds__(kBlock) void resize_cubic_v_kernel(VBatch<SrcT> vb, int sw, int sh, int dh,
                                                                const int *__restrict__ yofs, const float4 *__restrict__ beta)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    if (x >= sw) return;
    const SrcT *__restrict__ src = vb.src[blockIdx.z];
    float *__restrict__ dst = vb.dst[blockIdx.z];
    const long spitch = vb.spitch[blockIdx.z];
#pragma unroll
    for (int j = 0; j < kVRows; ++j) {
        const int dy = blockIdx.y * kVRows + j;
        if (dy >= dh) break;
        const int sy = yofs[dy];
        const float4 b = beta[dy];
        int y0 = sy - 1, y1 = sy, y2 = sy + 1, y3 = sy + 2;
        y0 = y0 < 0 ? 0 : (y0 > sh - 1 ? sh - 1 : y0);
        y1 = y1 < 0 ? 0 : (y1 > sh - 1 ? sh - 1 : y1);
        y2 = y2 < 0 ? 0 : (y2 > sh - 1 ? sh - 1 : y2);
        y3 = y3 < 0 ? 0 : (y3 > sh - 1 ? sh - 1 : y3);
        float v = __fmul_rn((float)src[(size_t)y0 * spitch + x], b.x);
        v = __fadd_rn(v, __fmul_rn((float)src[(size_t)y1 * spitch + x], b.y));
        v = __fadd_rn(v, __fmul_rn((float)src[(size_t)y2 * spitch + x], b.z));
        v = __fadd_rn(v, __fmul_rn((float)src[(size_t)y3 * spitch + x], b.w));
        dst[(size_t)dy * sw + x] = v;
    }
}

// Exact x4 up-sampling (the reference geometry: MSS GSD = 4 x PAN GSD).  One lane owns one
// source pixel (q, p) and produces its 4x4 block of outputs from the 5x5 source neighbourhood:
// the four outputs of a row share their horizontal taps, the four rows share the horizontal
// sums.  Taps, clamping and the order of every f32 product/sum are those of the generic kernel
// (the host verified that each output's first tap lies in the lane's 5 columns / rows);
// 25 loads and 4 16-byte stores per 16 outputs instead of 16 loads and one 4-byte store each.
template <typename SrcT>
__global__ __launch_bounds__(kBlock) void resize_cubic_x4_kernel(const SrcT *__restrict__ src, long spitch, int sw, int sh,
                                                                 float *__restrict__ dst, int dw,
                                                                 const int *__restrict__ xofs, const float4 *__restrict__ alpha,
                                                                 const int *__restrict__ yofs, const float4 *__restrict__ beta)
{
    const int q = blockIdx.x * kBlock + threadIdx.x;
    const int p = blockIdx.y;
    if (q >= sw) return;
    float S[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        int yr = p - 2 + r;
        yr = yr < 0 ? 0 : (yr > sh - 1 ? sh - 1 : yr);
        const SrcT *row = src + (size_t)yr * spitch;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            int xc = q - 2 + c;
            xc = xc < 0 ? 0 : (xc > sw - 1 ? sw - 1 : xc);
            S[r][c] = (float)row[xc];
        }
    }
    float H[5][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int dx = 4 * q + i;
        const float4 a = alpha[dx];
        const bool hi = (xofs[dx] - 1) - (q - 2) != 0;         // first tap is column q-1 instead of q-2
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const float s0 = hi ? S[r][1] : S[r][0], s1 = hi ? S[r][2] : S[r][1];
            const float s2 = hi ? S[r][3] : S[r][2], s3 = hi ? S[r][4] : S[r][3];
            float v = __fmul_rn(s0, a.x);
            v = __fadd_rn(v, __fmul_rn(s1, a.y));
            v = __fadd_rn(v, __fmul_rn(s2, a.z));
            v = __fadd_rn(v, __fmul_rn(s3, a.w));
            H[r][i] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int dy = 4 * p + j;
        const float4 b = beta[dy];
        const bool hi = (yofs[dy] - 1) - (p - 2) != 0;
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h0 = hi ? H[1][i] : H[0][i], h1 = hi ? H[2][i] : H[1][i];
            const float h2 = hi ? H[3][i] : H[2][i], h3 = hi ? H[4][i] : H[3][i];
            float v = __fmul_rn(h0, b.x);
            v = __fadd_rn(v, __fmul_rn(h1, b.y));
            v = __fadd_rn(v, __fmul_rn(h2, b.z));
            v = __fadd_rn(v, __fmul_rn(h3, b.w));
            o[i] = v;
        }
        *reinterpret_cast<float4 *>(dst + (size_t)dy * dw + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- cross-power spectrum -----------------------------------------------------------------------
struct SpecRef {
    const float2 *z;    // packed spectrum (scrambled order)
    int part;           // 0: the image in the real slot, 1: the image in the imaginary slot
};
struct XpowerJob {
    SpecRef a[2], b[2]; // correlation c: A = a[c], B = b[c]
    int ncorr;          // 1 or 2 (second goes to the imaginary slot of the output)
};

__device__ __forceinline__ float2 spec_of(int part, float2 zk, float2 zm)
{
    if (part == 0) return make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    return make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
}

// C for one bin, following mulSpectrums/magSpectrums/divSpectrums.  The formulas are odd in the
// imaginary part, so C(-k) == conj(C(k)) bit for bit: the CCS-implied half needs no own path.
__device__ __forceinline__ float2 cross_power_bin(float2 A, float2 B, bool real_bin, bool edge_col)
{
    const float eps = FLT_EPSILON;
    if (real_bin) {
        float p = __fmul_rn(A.x, B.x);
        float m = __fmul_rn(p, p);
        return make_float2(__fdiv_rn(p, __fadd_rn(m, eps)), 0.f);
    }
    float pr = __fadd_rn(__fmul_rn(A.x, B.x), __fmul_rn(A.y, B.y));
    float pi = __fsub_rn(__fmul_rn(A.y, B.x), __fmul_rn(A.x, B.y));
    float mag = (float)__dsqrt_rn(__dadd_rn(__dmul_rn((double)pr, (double)pr), __dmul_rn((double)pi, (double)pi)));
    if (edge_col) {
        double denom = __dadd_rn(__dmul_rn((double)mag, (double)mag), (double)eps);
        double re = __dmul_rn((double)pr, (double)mag);
        double im = __dmul_rn((double)pi, (double)mag);
        return make_float2((float)__ddiv_rn(re, denom), (float)__ddiv_rn(im, denom));
    }
    double denom = (double)__fadd_rn(__fmul_rn(mag, mag), eps);
    double re = (double)__fmul_rn(pr, mag);
    double im = (double)__fmul_rn(pi, mag);
    return make_float2((float)__ddiv_rn(re, denom), (float)__ddiv_rn(im, denom));
}

// The same bin for the fused row stage.  The spectra feeding it already differ from OpenCV's in
// their last bits (another FFT factorisation), so correctly rounded double-precision sqrt and
// divisions buy nothing there; what must survive is the structure: the magnitude formed in double
// (no overflow before the float rounding), mag * mag and p * mag rounded to float (where the
// reference overflows to inf and produces 0 or NaN), the + eps, and the special bins.  sqrt and the
// reciprocal use the hardware approximations (relative error about 1e-7, below the FFT round-off);
// inf and NaN propagate as in the division: x * (1 / inf) = 0, inf * 0 = NaN.
__device__ __forceinline__ float2 cross_power_bin_fast(float2 A, float2 B, bool real_bin, bool edge_col)
{
    if (real_bin || edge_col) return cross_power_bin(A, B, real_bin, edge_col);
    const float eps = FLT_EPSILON;
    float pr = __fadd_rn(__fmul_rn(A.x, B.x), __fmul_rn(A.y, B.y));
    float pi = __fsub_rn(__fmul_rn(A.y, B.x), __fmul_rn(A.x, B.y));
    float mag = (float)__builtin_amdgcn_sqrt(__dadd_rn(__dmul_rn((double)pr, (double)pr), __dmul_rn((double)pi, (double)pi)));
    float r = __builtin_amdgcn_rcpf(__fadd_rn(__fmul_rn(mag, mag), eps));
    return make_float2(__fmul_rn(__fmul_rn(pr, mag), r), __fmul_rn(__fmul_rn(pi, mag), r));
}

// One workgroup row handles the spectrum line of frequency ky AND its mirror -ky: every input
// line is read once, C(k) is computed once and written as Y(k) and Y(-k) = conj(C1) + i conj(C2).
// The x axis of the reference shapes is a single pass (natural order), so the mirrored
// column N-kx is a reversed, still coalesced, access.
__global__ __launch_bounds__(kBlock) void cross_power_kernel(float2 *__restrict__ out, XpowerJob job, int M, int N,
                                                             int P, OipAxisDigits yd, OipAxisDigits xd)
{
    const int px = blockIdx.x * kBlock + threadIdx.x;
    const int ky = blockIdx.y;                        // 0 .. M/2
    if (px >= N) return;
    const int kx = oip_pos_to_freq(xd, px);
    const int nkx = kx ? N - kx : 0, nky = ky ? M - ky : 0;
    const long r1 = (long)oip_freq_to_pos(yd, ky) * P, r2 = (long)oip_freq_to_pos(yd, nky) * P;
    const int px2 = oip_freq_to_pos(xd, nkx);
    const bool edge_col = (kx == 0) || (2 * kx == N);
    const bool real_bin = edge_col && (ky == 0 || 2 * ky == M);
    float2 y = make_float2(0.f, 0.f), ym = make_float2(0.f, 0.f);
    float2 zk[3], zm[3];
    const float2 *arr[3] = {nullptr, nullptr, nullptr};
    int narr = 0;
    // distinct input arrays of the job (at most 3)
    for (int c = 0; c < job.ncorr; ++c)
        for (int s = 0; s < 2; ++s) {
            const float2 *z = s ? job.b[c].z : job.a[c].z;
            bool seen = false;
            for (int i = 0; i < narr; ++i) seen = seen || arr[i] == z;
            if (!seen && narr < 3) arr[narr++] = z;
        }
    for (int i = 0; i < narr; ++i) { zk[i] = arr[i][r1 + px]; zm[i] = arr[i][r2 + px2]; }
    for (int c = 0; c < job.ncorr; ++c) {
        int ia = 0, ib = 0;
        for (int i = 0; i < narr; ++i) { if (arr[i] == job.a[c].z) ia = i; if (arr[i] == job.b[c].z) ib = i; }
        float2 A = spec_of(job.a[c].part, zk[ia], zm[ia]);
        float2 B = spec_of(job.b[c].part, zk[ib], zm[ib]);
        float2 C = cross_power_bin(A, B, real_bin, edge_col);
        if (c == 0) { y.x += C.x; y.y += C.y; ym.x += C.x; ym.y -= C.y; }        // Y = C1 + i C2
        else { y.x -= C.y; y.y += C.x; ym.x += C.y; ym.y += C.x; }               // i*conj(C2) = (C2.y, C2.x)
    }
    out[r1 + px] = y;
    if (r1 != r2) out[r2 + px2] = ym;
}

// Cross-power jobs flattened on the host: up to three distinct packed spectra, up to two output
// arrays, each output Y = C(a0,b0) + i C(a1,b1); per correlation which spectrum and which slot
// (real/imaginary) hold A and B
struct FusedJob {
    const float2 *z[3];
    float2 *out[2];
    int narr, nout;
    int ncorr[2];
    int ia[4], ib[4], pa[4], pb[4];     // correlation 2*o + c of output o
    int dbg;                            // experiment mask (OIP_ROWS_DBG): skip phases to time the rest; results are wrong
};

// Cross-power + inverse row pass (OIP_FUSED_ROWS=1: the forward row pass stays a separate launch; kept as
// the measured intermediate step towards corr_rows_kernel below).  The workgroup of frequency line ky owns
// the lines ky and -ky of every spectrum of the job:
// the cross-power lines Y(ky,.) and Y(-ky,.) of each output are formed in LDS straight from the
// registers, inverse row-transformed and stored -- Y never exists in the spectral domain.
template <int F, int NT, int WPE, int... Rs>
__global__ __launch_bounds__(NT, WPE) void xpower_rows_kernel(FusedJob fj, int M, int P, OipAxisDigits yd,
                                                         const float2 *__restrict__ twF)
{
    constexpr int TWN = oipfft::TwTable<F, Rs...>::value();
    constexpr int N = F;
    constexpr int NIT = (N + NT - 1) / NT;
    __shared__ float2 buf[2 * F];          // [point][line]: line 0 = ky, line 1 = -ky, interleaved
    __shared__ float2 tw[TWN];
    const int ky = blockIdx.x;
    const int nky = ky ? M - ky : 0;
    const long r1 = (long)oip_freq_to_pos(yd, ky) * P, r2 = (long)oip_freq_to_pos(yd, nky) * P;
    const bool pair = r1 != r2;
    for (int i = threadIdx.x; i < TWN; i += NT) tw[i] = twF[i];
    const int narr = fj.narr;
    // one register array per spectrum and line: static indices only (no scratch)
    float2 zk0[NIT], zk1[NIT], zk2[NIT], zm0[NIT], zm1[NIT], zm2[NIT];
    const float2 zero = make_float2(0.f, 0.f);
    const float2 *z0 = fj.z[0], *z1 = fj.z[1], *z2 = fj.z[2];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int kx = threadIdx.x + it * NT;
        const int nkx = kx ? N - kx : 0;
        const bool ok = kx < N;
        zk0[it] = ok ? z0[r1 + kx] : zero;
        zm0[it] = ok ? z0[r2 + nkx] : zero;
        zk1[it] = ok && narr > 1 ? z1[r1 + kx] : zero;
        zm1[it] = ok && narr > 1 ? z1[r2 + nkx] : zero;
        zk2[it] = ok && narr > 2 ? z2[r1 + kx] : zero;
        zm2[it] = ok && narr > 2 ? z2[r2 + nkx] : zero;
    }
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (o >= fj.nout) break;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int kx = threadIdx.x + it * NT;
            if (kx >= N) continue;
            const int nkx = kx ? N - kx : 0;
            const bool edge_col = (kx == 0) || (2 * kx == N);
            const bool real_bin = edge_col && (ky == 0 || 2 * ky == M);
            float2 y = zero, ym = zero;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c >= fj.ncorr[o]) break;
                const int ia = fj.ia[2 * o + c], ib = fj.ib[2 * o + c];
                const int pa = fj.pa[2 * o + c], pb = fj.pb[2 * o + c];
                float2 zka = ia == 0 ? zk0[it] : (ia == 1 ? zk1[it] : zk2[it]);
                float2 zma = ia == 0 ? zm0[it] : (ia == 1 ? zm1[it] : zm2[it]);
                float2 zkb = ib == 0 ? zk0[it] : (ib == 1 ? zk1[it] : zk2[it]);
                float2 zmb = ib == 0 ? zm0[it] : (ib == 1 ? zm1[it] : zm2[it]);
                float2 A = spec_of(pa, zka, zma);
                float2 B = spec_of(pb, zkb, zmb);
                float2 C = cross_power_bin(A, B, real_bin, edge_col);
                if (c == 0) { y.x += C.x; y.y += C.y; ym.x += C.x; ym.y -= C.y; }        // Y = C1 + i C2
                else { y.x -= C.y; y.y += C.x; ym.x += C.y; ym.y += C.x; }               // i*conj(C2) = (C2.y, C2.x)
            }
            // inverse = conj(forward(conj(.)))
            buf[2 * kx] = make_float2(y.x, -y.y);
            buf[2 * nkx + 1] = pair ? make_float2(ym.x, -ym.y) : zero;
        }
        __syncthreads();
        oipfft::Stages<F, 1, 2, NT, 1, Rs...>::run(buf, tw);        // both lines in one Stockham network
        float2 *out = fj.out[o];
        for (int e = threadIdx.x; e < 2 * N; e += NT) {
            const int line = e >= N, x = e - line * N;
            if (line && !pair) break;
            float2 a = buf[2 * x + line];
            out[(line ? r2 : r1) + x] = make_float2(a.x, -a.y);
        }
        __syncthreads();
    }
}

// Row stage of the whole correlation in one persistent kernel, for shapes whose row axis is a single
// factor (natural order along x).  The spectra arrive with only their column passes done.  A
// workgroup owns the lines ky and -ky of every spectrum of the job, each spectrum in its own
// two-line LDS buffer:
//   1. the prefetched lines (registers) go to LDS; the loads of the workgroup's NEXT line pair are
//      issued and stay in flight under everything below;
//   2. forward row transform of all spectra together (one barrier pair per stage);
//   3. cross-power in place: the thread of column kx is the only reader of bins (kx, ky) and
//      (-kx, -ky), so Y_o(kx, ky) / Y_o(-kx, -ky) overwrite them in buffer o;
//   4. inverse row transform of the outputs together; both lines of each output are stored.
// Against separate passes this saves a read + write of every spectrum (forward rows), a write +
// read of every Y (cross-power) and the re-read of spectra shared by two outputs; one workgroup
// per CU (3 x 48 KB of LDS at F = 3000) hides HBM latency by the register prefetch instead of
// occupancy.
template <int F, int NT, int NARR, int NOUT, int... Rs>
__global__ __launch_bounds__(NT) void corr_rows_kernel(FusedJob fj, int M, int P, const int *__restrict__ ypos,
                                                       const float2 *__restrict__ twF)
{
    constexpr int TWN = oipfft::TwTable<F, Rs...>::value();
    constexpr int N = F;
    constexpr int NIT = (N + NT - 1) / NT;
    __shared__ __align__(16) float2 buf[NARR * 2 * F];   // [spectrum][point][line]: line 0 = ky, line 1 = -ky
    __shared__ float2 tw[TWN];
    const int dbg = fj.dbg;
    const int half = M / 2;
    int ky = blockIdx.x;
    if (ky > half) return;
    for (int i = threadIdx.x; i < TWN; i += NT) tw[i] = twF[i];
    // Lines move 16 bytes (two points) per lane: half the vector-memory and LDS instructions of 8-byte
    // accesses -- their operand traffic shares the SIMD-to-LDS path with the stages' ds_writes.
    static_assert(N % 2 == 0, "two points per lane");
    constexpr int NIT2 = (N / 2 + NT - 1) / NT;
    float4 la[NARR][NIT2], lb[NARR][NIT2];
    float4 *buf4 = reinterpret_cast<float4 *>(buf);
    // rows of the line pair whose loads are in la/lb (n1, n2) and of the pair in LDS (s1, s2)
    // (row positions come from a per-plan table: decoding them with the digit loop cost ~300 scalar
    // instructions and four dependent scalar loads per iteration, on every wave at the same time)
    long n1 = (long)ypos[ky] * P, n2 = (long)ypos[ky ? M - ky : 0] * P;
    auto fetch = [&](int tid) {
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                // lanes past the end of the line re-read its last pair (never committed): a select on
                // the loaded value would need the load to have completed -- a wait inside the prefetch
                int q = tid + it * NT;
                q = q < N / 2 ? q : N / 2 - 1;
                la[a][it] = *reinterpret_cast<const float4 *>(fj.z[a] + n1 + 2 * q);
                lb[a][it] = *reinterpret_cast<const float4 *>(fj.z[a] + n2 + 2 * q);
            }
        }
    };
    auto commit = [&](int tid) {
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int q = tid + it * NT;
                if (q < N / 2) {
                    // [point][line] interleave: (2q, line 0), (2q, line 1), (2q+1, line 0), (2q+1, line 1):
                    // two 16-byte writes 32 bytes apart per lane (2-way bank conflict; four 8-byte writes
                    // would be 4-way).  The repacking moves sit here, after the loads have landed.
                    buf4[a * F + 2 * q] = make_float4(la[a][it].x, la[a][it].y, lb[a][it].x, lb[a][it].y);
                    buf4[a * F + 2 * q + 1] = make_float4(la[a][it].z, la[a][it].w, lb[a][it].z, lb[a][it].w);
                }
            }
        }
    };
    fetch(threadIdx.x);
    commit(threadIdx.x);
    long s1 = n1, s2 = n2;
    __syncthreads();
    for (; ky <= half; ky += gridDim.x) {
        // opaque per iteration: keeps the stage address arithmetic from being hoisted out of this
        // loop, where it would occupy registers for the whole kernel
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const bool pair = s1 != s2;
        // The lines of the NEXT pair are requested first and committed at the bottom of this same
        // iteration: the registers holding them then never cross the loop's back edge.  (Carried across
        // it, the register allocator gave the loop-header values other registers than the loads'
        // destinations and copied right after the loads -- waiting for them at the point of issue.)
        const int kn = ky + gridDim.x;
        const bool more = kn <= half;
        if (more && !(dbg & 8)) {
            n1 = (long)ypos[kn] * P;
            n2 = (long)ypos[M - kn] * P;
            fetch(tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 1)) oipfft::StagesPipe<F, NT, NARR, 1, Rs...>::run(buf, tw, tid);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int kx = tid + it * NT;
            if (kx >= N || (dbg & 2)) continue;
            const int nkx = kx ? N - kx : 0;
            const bool edge_col = (kx == 0) || (2 * kx == N);
            const bool real_bin = edge_col && (ky == 0 || 2 * ky == M);
            // The two job shapes are fixed (xpower_stage checks them on the host), so which spectrum and
            // slot feeds which correlation is known here -- no run-time selects:
            //   1 spectrum : Y0 = C(z0.re, z0.im)
            //   3 spectra  : A = z0.re against z0.im, z1.re | z1.im, z2.(re or im: fj.pb[3])
            const float2 zk0 = buf[2 * kx], zm0 = buf[2 * nkx + 1];
            const float2 A = spec_of(0, zk0, zm0);
            float2 y0, y0m, y1 = make_float2(0.f, 0.f), y1m = make_float2(0.f, 0.f);
            {
                const float2 C = cross_power_bin_fast(A, spec_of(1, zk0, zm0), real_bin, edge_col);
                y0 = C;
                y0m = make_float2(C.x, -C.y);
            }
            if (NARR == 3) {
                const float2 zk1 = buf[2 * F + 2 * kx], zm1 = buf[2 * F + 2 * nkx + 1];
                const float2 zk2 = buf[4 * F + 2 * kx], zm2 = buf[4 * F + 2 * nkx + 1];
                float2 C = cross_power_bin_fast(A, spec_of(0, zk1, zm1), real_bin, edge_col);
                y0.x -= C.y; y0.y += C.x; y0m.x += C.y; y0m.y += C.x;                        // + i C, + i conj(C)
                C = cross_power_bin_fast(A, spec_of(1, zk1, zm1), real_bin, edge_col);
                y1 = C;
                y1m = make_float2(C.x, -C.y);
                const float2 B3 = fj.pb[3] ? spec_of(1, zk2, zm2) : spec_of(0, zk2, zm2);
                C = cross_power_bin_fast(A, B3, real_bin, edge_col);
                y1.x -= C.y; y1.y += C.x; y1m.x += C.y; y1m.y += C.x;
            }
            // inverse = conj(forward(conj(.)))
            buf[2 * kx] = make_float2(y0.x, -y0.y);
            buf[2 * nkx + 1] = pair ? make_float2(y0m.x, -y0m.y) : make_float2(0.f, 0.f);
            if (NOUT == 2) {
                buf[2 * F + 2 * kx] = make_float2(y1.x, -y1.y);
                buf[2 * F + 2 * nkx + 1] = pair ? make_float2(y1m.x, -y1m.y) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        asm volatile("" : "+v"(tid));
        if (!(dbg & 4)) oipfft::StagesPipe<F, NT, NOUT, 1, Rs...>::run(buf, tw, tid);
        if (NOUT == 1) __syncthreads();
        // Results leave LDS through registers so that the next pair can be committed before the stores
        // are issued.
        float4 ya[NOUT][NIT2], yb[NOUT][NIT2];      // line ky / line -ky, two points each
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int q = tid + it * NT;
                if (q < N / 2) {
                    const float4 u = buf4[o * F + 2 * q], v = buf4[o * F + 2 * q + 1];
                    ya[o][it] = make_float4(u.x, -u.y, v.x, -v.y);
                    yb[o][it] = make_float4(u.z, -u.w, v.z, -v.w);
                }
            }
        }
        __syncthreads();
        if (more) commit(tid);
        __builtin_amdgcn_sched_barrier(0);
        const long o1 = s1, o2 = s2;
        s1 = n1; s2 = n2;
        if (!(dbg & 16)) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float2 *out = fj.out[o];
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int q = tid + it * NT;
                    if (q < N / 2) {
                        *reinterpret_cast<float4 *>(out + o1 + 2 * q) = ya[o][it];
                        if (pair) *reinterpret_cast<float4 *>(out + o2 + 2 * q) = yb[o][it];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// The same row stage with a three-stage factorisation (3000 = 25 * 15 * 8: composite butterflies in registers,
// oip_fft_dev.h) and every stage dealt over all buffers at once: a point crosses LDS three times per transform instead
// of five, a stage costs two barriers for all buffers instead of one per buffer.  Single-line items keep LDS accesses
// 8 bytes wide and conflict-free (the first stage stores at a 25-point stride: 100 dwords, odd multiple of 4).
// Measured per launch: 0.80 -> 0.76 ms at 3000 points.  The 1250-point geometry (25 * 25 * 2) was measured too and is
// slower this way (0.46 vs 0.33 ms: eight rounds of radix-2 items), so it keeps the five-stage kernel.
template <int F, int NT, int NARR, int NOUT, int... Rs>
__global__ __launch_bounds__(NT) void corr_rows3_kernel(FusedJob fj, int M, int P, const int *__restrict__ ypos,
                                                       const float2 *__restrict__ twF)
{
    constexpr int TWN = oipfft::TwTable<F, Rs...>::value();
    constexpr int N = F;
    constexpr int NIT = (N + NT - 1) / NT;
    __shared__ __align__(16) float2 buf[NARR * 2 * F];   // [spectrum][point][line]: line 0 = ky, line 1 = -ky
    __shared__ float2 tw[TWN];
    const int dbg = fj.dbg;
    const int half = M / 2;
    int ky = blockIdx.x;
    if (ky > half) return;
    for (int i = threadIdx.x; i < TWN; i += NT) tw[i] = twF[i];
    // Lines move 16 bytes (two points) per lane: half the vector-memory and LDS instructions of 8-byte
    // accesses -- their operand traffic shares the SIMD-to-LDS path with the stages' ds_writes.
    static_assert(N % 2 == 0, "two points per lane");
    constexpr int NIT2 = (N / 2 + NT - 1) / NT;
    float4 la[NARR][NIT2], lb[NARR][NIT2];
    float4 *buf4 = reinterpret_cast<float4 *>(buf);
    // rows of the line pair whose loads are in la/lb (n1, n2) and of the pair in LDS (s1, s2)
    // (row positions come from a per-plan table: decoding them with the digit loop cost ~300 scalar
    // instructions and four dependent scalar loads per iteration, on every wave at the same time)
    long n1 = (long)ypos[ky] * P, n2 = (long)ypos[ky ? M - ky : 0] * P;
    auto fetch = [&](int tid) {
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                // lanes past the end of the line re-read its last pair (never committed): a select on
                // the loaded value would need the load to have completed -- a wait inside the prefetch
                int q = tid + it * NT;
                q = q < N / 2 ? q : N / 2 - 1;
                la[a][it] = *reinterpret_cast<const float4 *>(fj.z[a] + n1 + 2 * q);
                lb[a][it] = *reinterpret_cast<const float4 *>(fj.z[a] + n2 + 2 * q);
            }
        }
    };
    auto commit = [&](int tid) {
#pragma unroll
        for (int a = 0; a < NARR; ++a) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int q = tid + it * NT;
                if (q < N / 2) {
                    // [point][line] interleave: (2q, line 0), (2q, line 1), (2q+1, line 0), (2q+1, line 1):
                    // two 16-byte writes 32 bytes apart per lane (2-way bank conflict; four 8-byte writes
                    // would be 4-way).  The repacking moves sit here, after the loads have landed.
                    buf4[a * F + 2 * q] = make_float4(la[a][it].x, la[a][it].y, lb[a][it].x, lb[a][it].y);
                    buf4[a * F + 2 * q + 1] = make_float4(la[a][it].z, la[a][it].w, lb[a][it].z, lb[a][it].w);
                }
            }
        }
    };
    fetch(threadIdx.x);
    commit(threadIdx.x);
    long s1 = n1, s2 = n2;
    __syncthreads();
    for (; ky <= half; ky += gridDim.x) {
        // opaque per iteration: keeps the stage address arithmetic from being hoisted out of this
        // loop, where it would occupy registers for the whole kernel
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const bool pair = s1 != s2;
        // The lines of the NEXT pair are requested first and committed at the bottom of this same
        // iteration: the registers holding them then never cross the loop's back edge.  (Carried across
        // it, the register allocator gave the loop-header values other registers than the loads'
        // destinations and copied right after the loads -- waiting for them at the point of issue.)
        const int kn = ky + gridDim.x;
        const bool more = kn <= half;
        if (more && !(dbg & 8)) {
            n1 = (long)ypos[kn] * P;
            n2 = (long)ypos[M - kn] * P;
            fetch(tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 1)) oipfft::StagesAll<F, NT, NARR, 1, Rs...>::run(buf, tw, tid);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int kx = tid + it * NT;
            if (kx >= N || (dbg & 2)) continue;
            const int nkx = kx ? N - kx : 0;
            const bool edge_col = (kx == 0) || (2 * kx == N);
            const bool real_bin = edge_col && (ky == 0 || 2 * ky == M);
            // The two job shapes are fixed (xpower_stage checks them on the host), so which spectrum and
            // slot feeds which correlation is known here -- no run-time selects:
            //   1 spectrum : Y0 = C(z0.re, z0.im)
            //   3 spectra  : A = z0.re against z0.im, z1.re | z1.im, z2.(re or im: fj.pb[3])
            const float2 zk0 = buf[2 * kx], zm0 = buf[2 * nkx + 1];
            const float2 A = spec_of(0, zk0, zm0);
            float2 y0, y0m, y1 = make_float2(0.f, 0.f), y1m = make_float2(0.f, 0.f);
            {
                const float2 C = cross_power_bin_fast(A, spec_of(1, zk0, zm0), real_bin, edge_col);
                y0 = C;
                y0m = make_float2(C.x, -C.y);
            }
            if (NARR == 3) {
                const float2 zk1 = buf[2 * F + 2 * kx], zm1 = buf[2 * F + 2 * nkx + 1];
                const float2 zk2 = buf[4 * F + 2 * kx], zm2 = buf[4 * F + 2 * nkx + 1];
                float2 C = cross_power_bin_fast(A, spec_of(0, zk1, zm1), real_bin, edge_col);
                y0.x -= C.y; y0.y += C.x; y0m.x += C.y; y0m.y += C.x;                        // + i C, + i conj(C)
                C = cross_power_bin_fast(A, spec_of(1, zk1, zm1), real_bin, edge_col);
                y1 = C;
                y1m = make_float2(C.x, -C.y);
                const float2 B3 = fj.pb[3] ? spec_of(1, zk2, zm2) : spec_of(0, zk2, zm2);
                C = cross_power_bin_fast(A, B3, real_bin, edge_col);
                y1.x -= C.y; y1.y += C.x; y1m.x += C.y; y1m.y += C.x;
            }
            // inverse = conj(forward(conj(.)))
            buf[2 * kx] = make_float2(y0.x, -y0.y);
            buf[2 * nkx + 1] = pair ? make_float2(y0m.x, -y0m.y) : make_float2(0.f, 0.f);
            if (NOUT == 2) {
                buf[2 * F + 2 * kx] = make_float2(y1.x, -y1.y);
                buf[2 * F + 2 * nkx + 1] = pair ? make_float2(y1m.x, -y1m.y) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        asm volatile("" : "+v"(tid));
        if (!(dbg & 4)) oipfft::StagesAll<F, NT, NOUT, 1, Rs...>::run(buf, tw, tid);
        // Results leave LDS through registers so that the next pair can be committed before the stores
        // are issued.
        float4 ya[NOUT][NIT2], yb[NOUT][NIT2];      // line ky / line -ky, two points each
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int q = tid + it * NT;
                if (q < N / 2) {
                    const float4 u = buf4[o * F + 2 * q], v = buf4[o * F + 2 * q + 1];
                    ya[o][it] = make_float4(u.x, -u.y, v.x, -v.y);
                    yb[o][it] = make_float4(u.z, -u.w, v.z, -v.w);
                }
            }
        }
        __syncthreads();
        if (more) commit(tid);
        __builtin_amdgcn_sched_barrier(0);
        const long o1 = s1, o2 = s2;
        s1 = n1; s2 = n2;
        if (!(dbg & 16)) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                float2 *out = fj.out[o];
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int q = tid + it * NT;
                    if (q < N / 2) {
                        *reinterpret_cast<float4 *>(out + o1 + 2 * q) = ya[o][it];
                        if (pair) *reinterpret_cast<float4 *>(out + o2 + 2 * q) = yb[o][it];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- row stage with the x4 cubic up-sampling of the bands applied to their spectra -------------------------------
//
// cv::resize(INTER_CUBIC) by exactly 4 is linear: along one axis, out = R s with R = C + E, where C is the circulant
// operator "zero-stuff by 4, convolve with the 16-tap kernel h" and E holds what edge replication changes -- it is
// non-zero in the columns of the source samples {0, 1, n-2, n-1} only (the taps that leave the image are clamped to
// sample 0 or n-1 instead of wrapping to n-2, n-1, 0, 1).  So the N = 4n point transform of an up-sampled line is
//     DFT_N(R s)[k] = H[k] DFT_n(s)[k mod n] + sum_j G_j[k] s[J_j],       H = DFT_N(h), G_j = DFT_N(E[:, J_j])
// -- a 750-point transform, one complex multiply and four multiply-adds per bin instead of a 3000-point transform of
// the up-sampled line.  The same holds along the columns (VEXP): the band windows are transformed at their own size
// (4000 x 750) and line ky of the up-sampled band's column transform is Hv[ky] B^[ky mod 4000] + sum_i Gv_i[ky] raw_i.
// Without VEXP (OIP_SPECTRAL_UP=1) the vertical taps are applied in the image domain (resize_cubic_v) and the columns of
// the resulting 16000 x 750 images are transformed.  The identity is exact; what changes against transforming the
// up-sampled image is the rounding (the f32 rounding of every up-sampled pixel is replaced by the rounding of H, G and
// the products), 4e-6 px on the shifts -- the size of the difference between any two float FFTs
// (tests/test_gpu_correlation.py compares the routes with each other and with the oracle, which up-samples first).
//
// One launch serves a pair of units: zp = PAN_A + i PAN_B (full width), four narrow arrays (bands 0|1 and 2|3 of unit
// A, then of unit B) and four outputs Y = C(PAN, band 2a) + i C(PAN, band 2a+1).  LDS holds the PAN line pair, one
// more full-width buffer and the eight narrow lines (144 KB): after the forward transforms the PAN spectra of the
// thread's bins move to registers, and the outputs are formed, inverse-transformed and stored two at a time in the two
// full-width buffers.  A thread owns the same bins of every line pair: kx <= N/2 and N - kx of line ky (and their
// mirrors in line -ky), which share H and G up to conjugation.
struct UpRowsJob {
    const float2 *zp;       // pitch P
    const float2 *zn;       // four narrow arrays, zn_stride elements apart, pitch Pn
    long zn_stride;
    int Pn;
    float2 *out[4];
    const float2 *xtab;     // [5][N]: H, G_0 .. G_3
    int nout;               // 4, or 2 for a single unit (arrays 2 and 3 are not read)
    int dbg;                // experiment mask like FusedJob::dbg; 32: skip the vertical expansion (results are wrong)
    // VEXP: the vertical up-sampling is applied to the spectra as well.  zn then holds the column transforms of the
    // band windows themselves (m = M / 4 rows): line ky of the up-sampled band is
    //     Hv[ky] zn[ky mod m] + sum_i Gv_i[ky] raw[i],      raw = band rows {0, 1, m-2, m-1}
    const float2 *vtab;     // [5][M]: Hv, Gv_0 .. Gv_3
    const uint2 *raw16;     // [4 rows][4 arrays x n / 2]: (bX | bY << 16) of two neighbouring points
    const int *ypos_s;      // row position of frequency line k in zn, k in [0, m)
    int m;
};

// conj(a) b, acc + a b, acc + conj(a) b on packed-f32 instructions (two each; operand selection as in oipfft::cmul).
// The spectral operator is part of the transform: multiply-adds fuse (see oip_fft_dev.h).
__device__ __forceinline__ float2 cmulj(float2 a, float2 b)
{
    oipfft::oip_v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(oipfft::to_v(a)), "v"(oipfft::to_v(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(oipfft::to_v(a)), "v"(oipfft::to_v(b)), "v"(t));
    return oipfft::to_f2(r);
}
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 acc)
{
    oipfft::oip_v2f t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(oipfft::to_v(a)), "v"(oipfft::to_v(b)), "v"(oipfft::to_v(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(oipfft::to_v(a)), "v"(oipfft::to_v(b)), "v"(t));
    return oipfft::to_f2(r);
}
__device__ __forceinline__ float2 cfmaj(float2 a, float2 b, float2 acc)
{
    oipfft::oip_v2f t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(oipfft::to_v(a)), "v"(oipfft::to_v(b)), "v"(oipfft::to_v(acc)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(oipfft::to_v(a)), "v"(oipfft::to_v(b)), "v"(t));
    return oipfft::to_f2(r);
}

// spec_of without its factor 1/2 (the row stage's horizontal tables carry it)
__device__ __forceinline__ float2 spec2_of(int part, float2 zk, float2 zm)
{
    if (part == 0) return make_float2(zk.x + zm.x, zk.y - zm.y);
    return make_float2(zk.y + zm.y, zm.x - zk.x);
}
// acc + s v for a real s (one packed multiply-add)
__device__ __forceinline__ float2 sfma(float s, float2 v, float2 acc)
{
    OIP_FFT_FMA
    return make_float2(acc.x + s * v.x, acc.y + s * v.y);
}

template <int NT, bool VEXP>
__global__ __launch_bounds__(NT) void corr_rows_up_kernel(UpRowsJob fj, int M, int P, const int *__restrict__ ypos,
                                                          const float2 *__restrict__ twF, const float2 *__restrict__ twS)
{
    constexpr int F = 3000, S = F / 4;
    constexpr int TWF = oipfft::TwTable<F, 25, 15, 8>::value(), TWS = oipfft::TwTable<S, 25, 15, 2>::value();
    // A thread owns the bins kx = tid + NT r <= N/2 of line ky and their mirrors N - kx of the same line: H and G are
    // transforms of real sequences, so the mirror's table values are the conjugates -- half the table registers.
    constexpr int NB = (F / 2 + 1 + NT - 1) / NT;
    __shared__ __align__(16) float2 buf[2 * 2 * F + 4 * 2 * S];    // PAN / output 0 | output 1 | narrow arrays; [point][line]
    __shared__ float2 tw[TWF], tws[TWS];
    __shared__ float2 edge[4][2][4];                               // [array][line][j]: narrow line samples 0, 1, S-2, S-1
    __shared__ float2 edgeA[2][2];                                 // PAN spectra (unit A, unit B) at kx = 0 and N/2
    __shared__ float2 edgeT[2][5];                                 // H, G_0..3 at kx = 0 and N/2
    float2 *bufN = buf + 4 * F;
    float4 *buf4 = reinterpret_cast<float4 *>(buf), *buf4N = reinterpret_cast<float4 *>(bufN);
    const int dbg = fj.dbg;
    const int half = M / 2;
    int ky = blockIdx.x;
    if (ky > half) return;
    for (int i = threadIdx.x; i < TWF; i += NT) tw[i] = twF[i];
    for (int i = threadIdx.x; i < TWS; i += NT) tws[i] = twS[i];
    if (threadIdx.x < 10) edgeT[threadIdx.x / 5][threadIdx.x % 5] = fj.xtab[(threadIdx.x % 5) * F + (threadIdx.x / 5) * (F / 2)];
    // H and G of the thread's bins: re-read from the L2-resident table (120 KB) for each of the two cross-power rounds
    // (held across the inverse transforms they cost the registers the prefetched lines need).  Vector-memory
    // operations complete in order: the second load is issued before the stores of the first round and before the
    // prefetch, so that waiting for it waits for nothing else.
    float2 H[NB], G[NB][4];
    auto load_tables = [&](int tid) {
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            int kx = tid + NT * r;
            kx = kx <= F / 2 ? kx : 0;
            H[r] = fj.xtab[kx];
#pragma unroll
            for (int j = 0; j < 4; ++j) G[r][j] = fj.xtab[(1 + j) * F + kx];
        }
    };
    constexpr int NIT2 = (F / 2 + NT - 1) / NT;
    constexpr int NQ = 4 * (S / 2);                 // 16-byte pieces of one line of the four narrow arrays
    constexpr int NITN = (NQ + NT - 1) / NT;
    float4 la[NIT2], lb[NIT2], na[NITN], nb[NITN];
    long n1 = ypos[ky], n2 = ypos[ky ? M - ky : 0];     // rows (the pitches differ between zp and zn)
    // rows of the narrow arrays: the same lines, or (VEXP) the lines ky mod m and -ky mod m of the band transforms
    auto narrow_row = [&](int k, long full) { return VEXP ? (long)fj.ypos_s[k % fj.m] : full; };
    long m1 = narrow_row(ky, n1), m2 = narrow_row(ky ? M - ky : 0, n2);
    int kyc = ky;                                       // the frequency line whose loads are in the registers
    float2 vt[5];                                       // VEXP: Hv, Gv_0..3 of line kyc
    uint2 rawreg[NITN][4];                              // VEXP: the raw band rows at this thread's points (bX | bY << 16, two points)
    // VEXP: the narrow lines, their coefficients and the raw rows are requested at the commit itself -- their
    // registers (58 per thread) held across the second round's transforms spill, and a spilled load is waited for
    // where it is issued.  They are a quarter of the input bytes and mostly L2 / Infinity Cache hits (every line of
    // a band transform serves four frequency lines, the raw rows serve all).
    auto fetch_narrow = [&](int tid) {
        if (VEXP) {
#pragma unroll
            for (int r = 0; r < 5; ++r) vt[r] = fj.vtab[r * M + kyc];
#pragma unroll
            for (int it = 0; it < NITN; ++it) {
                int q = tid + it * NT;
                q = q < NQ ? q : NQ - 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) rawreg[it][r] = fj.raw16[(long)r * NQ + q];
            }
        }
#pragma unroll
        for (int it = 0; it < NITN; ++it) {
            int q = tid + it * NT;
            q = q < NQ ? q : NQ - 1;
            const int a = q / (S / 2), i = q - a * (S / 2);
            const float2 *z = fj.zn + a * fj.zn_stride + 2 * i;
            na[it] = *reinterpret_cast<const float4 *>(z + m1 * fj.Pn);
            nb[it] = *reinterpret_cast<const float4 *>(z + m2 * fj.Pn);
        }
    };
    auto fetch = [&](int tid) {
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            int q = tid + it * NT;
            q = q < F / 2 ? q : F / 2 - 1;
            la[it] = *reinterpret_cast<const float4 *>(fj.zp + n1 * P + 2 * q);
            lb[it] = *reinterpret_cast<const float4 *>(fj.zp + n2 * P + 2 * q);
        }
        if (!VEXP) fetch_narrow(tid);
    };
    auto commit = [&](int tid) {
        if (VEXP) fetch_narrow(tid);
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            const int q = tid + it * NT;
            if (q < F / 2) {
                buf4[2 * q] = make_float4(la[it].x, la[it].y, lb[it].x, lb[it].y);
                buf4[2 * q + 1] = make_float4(la[it].z, la[it].w, lb[it].z, lb[it].w);
            }
        }
#pragma unroll
        for (int it = 0; it < NITN; ++it) {
            const int q = tid + it * NT;
            if (q < NQ) {
                const int a = q / (S / 2), i = q - a * (S / 2);
                if (VEXP && !(dbg & 32)) {
                    // line ky of the vertically up-sampled band pair from line ky mod m of its transform; Hv and Gv of
                    // line -ky are the conjugates.  The raw band rows of this thread's points are kernel-long
                    // constants (two u16 pairs per row and piece).
                    const float2 hv = vt[0];
                    float2 x0 = oipfft::cmul(hv, make_float2(na[it].x, na[it].y)), x1 = oipfft::cmul(hv, make_float2(na[it].z, na[it].w));
                    float2 y0 = cmulj(hv, make_float2(nb[it].x, nb[it].y)), y1 = cmulj(hv, make_float2(nb[it].z, nb[it].w));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float2 gv = vt[1 + r];
                        const uint2 u = rawreg[it][r];
                        const float2 w0 = make_float2((float)(u.x & 0xffffu), (float)(u.x >> 16)), w1 = make_float2((float)(u.y & 0xffffu), (float)(u.y >> 16));
                        x0 = cfma(gv, w0, x0); x1 = cfma(gv, w1, x1);
                        y0 = cfmaj(gv, w0, y0); y1 = cfmaj(gv, w1, y1);
                    }
                    na[it] = make_float4(x0.x, x0.y, x1.x, x1.y);
                    nb[it] = make_float4(y0.x, y0.y, y1.x, y1.y);
                }
                buf4N[a * S + 2 * i] = make_float4(na[it].x, na[it].y, nb[it].x, nb[it].y);
                buf4N[a * S + 2 * i + 1] = make_float4(na[it].z, na[it].w, nb[it].z, nb[it].w);
                if (i == 0 || i == S / 2 - 1) {
                    const int j = i ? 2 : 0;
                    edge[a][0][j] = make_float2(na[it].x, na[it].y); edge[a][0][j + 1] = make_float2(na[it].z, na[it].w);
                    edge[a][1][j] = make_float2(nb[it].x, nb[it].y); edge[a][1][j + 1] = make_float2(nb[it].z, nb[it].w);
                }
            }
        }
    };
    fetch(threadIdx.x);
    commit(threadIdx.x);
    long s1 = n1, s2 = n2;
    __syncthreads();
    bool first = true;
    for (; ky <= half; ky += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const bool pair = s1 != s2;
        const int kn = ky + gridDim.x;
        const bool more = kn <= half;
        if (first) load_tables(tid);            // later iterations: requested before the previous iteration's last stores
        first = false;
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 1)) {
            oipfft::StagesDual3<F, 1, S, 4, NT, 25, 15, 8, 2>::run(buf, tw, bufN, tws, tid);
        }
        // PAN spectra of this thread's bins (unit A in the real slot, unit B in the imaginary one); those of the two
        // edge columns (kx = 0, N/2: divSpectrums' double-precision and real-only formulas) also go to edgeA
        float2 Aa[2 * NB], Ab[2 * NB];                  // [2 r]: bin kx, [2 r + 1]: bin N - kx
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const int kx = tid + NT * r, nkx = kx ? F - kx : 0;
            if (kx <= F / 2) {
                const float2 zk = buf[2 * kx], zm = buf[2 * nkx + 1];
                Aa[2 * r] = spec_of(0, zk, zm);
                Ab[2 * r] = spec_of(1, zk, zm);
                const float2 zk2 = buf[2 * nkx], zm2 = buf[2 * kx + 1];
                Aa[2 * r + 1] = spec_of(0, zk2, zm2);
                Ab[2 * r + 1] = spec_of(1, zk2, zm2);
                if (kx == 0 || 2 * kx == F) { edgeA[kx ? 1 : 0][0] = Aa[2 * r]; edgeA[kx ? 1 : 0][1] = Ab[2 * r]; }
            }
        }
        // outputs o0, o0 + 1 (narrow arrays of the same index) into the two full-width buffers
        auto xround = [&](int o0, const float2 (&A)[2 * NB]) {
            if (dbg & 2) return;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float2 *zn = bufN + (o0 + h) * 2 * S;
                float2 e0[4], e1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { e0[j] = edge[o0 + h][0][j]; e1[j] = edge[o0 + h][1][j]; }
                float2 *ob = buf + h * 2 * F;
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const int kx = tid + NT * r, nkx = kx ? F - kx : 0;
                    if (kx > F / 2) continue;
                    const int c = kx % S, cm = c ? S - c : 0;      // the narrow bin of kx and of -kx
                    // sum_j G_j e_j = S + i T and sum_j conj(G_j) e_j = S - i T with S = sum Re(G_j) e_j, T = sum Im(G_j) e_j:
                    // bin kx and its mirror N - kx (conjugate table values) share the products, line by line
                    float2 S0 = make_float2(0.f, 0.f), T0 = S0, S1 = S0, T1 = S0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        S0 = sfma(G[r][j].x, e0[j], S0); T0 = sfma(G[r][j].y, e0[j], T0);
                        S1 = sfma(G[r][j].x, e1[j], S1); T1 = sfma(G[r][j].y, e1[j], T1);
                    }
                    {
                        // packed band spectrum at (ky, kx) and at (-ky, -kx): H and G of -kx are the conjugates
                        const float2 Z0 = oipfft::csub_rot(oipfft::cadd(oipfft::cmul(H[r], zn[2 * c]), S0), T0);          // + S0 + i T0
                        const float2 Z1 = oipfft::cadd_rot(oipfft::cadd(cmulj(H[r], zn[2 * cm + 1]), S1), T1);            // + S1 - i T1
                        const float2 C1 = cross_power_bin_fast(A[2 * r], spec2_of(0, Z0, Z1), false, false);
                        const float2 C2 = cross_power_bin_fast(A[2 * r], spec2_of(1, Z0, Z1), false, false);
                        // Y = C1 + i C2 at the bin, conj(C1) + i conj(C2) at its mirror; inverse = conj(forward(conj(.)))
                        ob[2 * kx] = make_float2(C1.x - C2.y, -(C1.y + C2.x));
                        ob[2 * nkx + 1] = pair ? make_float2(C1.x + C2.y, -(C2.x - C1.y)) : make_float2(0.f, 0.f);
                    }
                    if (kx != 0 && 2 * kx != F) {
                        // the same for bin (ky, N - kx) and its mirror (-ky, kx)
                        const float2 Z0 = oipfft::cadd_rot(oipfft::cadd(cmulj(H[r], zn[2 * cm]), S0), T0);                // + S0 - i T0
                        const float2 Z1 = oipfft::csub_rot(oipfft::cadd(oipfft::cmul(H[r], zn[2 * c + 1]), S1), T1);      // + S1 + i T1
                        const float2 C1 = cross_power_bin_fast(A[2 * r + 1], spec2_of(0, Z0, Z1), false, false);
                        const float2 C2 = cross_power_bin_fast(A[2 * r + 1], spec2_of(1, Z0, Z1), false, false);
                        ob[2 * nkx] = make_float2(C1.x - C2.y, -(C1.y + C2.x));
                        ob[2 * kx + 1] = pair ? make_float2(C1.x + C2.y, -(C2.x - C1.y)) : make_float2(0.f, 0.f);
                    }
                    // one bin pair at a time (keeps the LDS reads of all bins from being hoisted above the arithmetic)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the two edge columns again with their own formulas (one copy of that code, two threads of the block)
#pragma unroll 1
            for (int r = 0; r < NB; ++r) {
                const int kx = tid + NT * r;
                if (kx != 0 && 2 * kx != F) continue;
                const bool real_bin = ky == 0 || 2 * ky == M;
                const float2 Ae = edgeA[kx ? 1 : 0][o0 ? 1 : 0];
                // (from LDS: a vector-memory load here would be younger than the prefetch and wait for it)
                float2 He = edgeT[kx ? 1 : 0][0], Ge[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) Ge[j] = edgeT[kx ? 1 : 0][1 + j];
#pragma unroll 1
                for (int h = 0; h < 2; ++h) {
                    const float2 *zn = bufN + (o0 + h) * 2 * S;
                    float2 Z0 = oipfft::cmul(He, zn[0]), Z1 = cmulj(He, zn[1]);       // kx = 0 and N/2 both map to narrow bin 0
#pragma unroll
                    for (int j = 0; j < 4; ++j) { Z0 = cfma(Ge[j], edge[o0 + h][0][j], Z0); Z1 = cfmaj(Ge[j], edge[o0 + h][1][j], Z1); }
                    const float2 C1 = cross_power_bin(Ae, spec2_of(0, Z0, Z1), real_bin, true);
                    const float2 C2 = cross_power_bin(Ae, spec2_of(1, Z0, Z1), real_bin, true);
                    float2 *ob = buf + h * 2 * F;
                    ob[2 * kx] = make_float2(C1.x - C2.y, -(C1.y + C2.x));
                    ob[2 * kx + 1] = pair ? make_float2(C1.x + C2.y, -(C2.x - C1.y)) : make_float2(0.f, 0.f);   // -kx == kx here
                }
            }
        };
        float4 ya[2][NIT2], yb[2][NIT2];
        auto finish = [&]() {           // inverse row transform of the two buffers; results to registers
            __syncthreads();
            asm volatile("" : "+v"(tid));
            if (!(dbg & 4)) oipfft::StagesAll<F, NT, 2, 1, 25, 15, 8>::run(buf, tw, tid);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int q = tid + it * NT;
                    if (q < F / 2) {
                        const float4 u = buf4[o * F + 2 * q], v = buf4[o * F + 2 * q + 1];
                        ya[o][it] = make_float4(u.x, -u.y, v.x, -v.y);
                        yb[o][it] = make_float4(u.z, -u.w, v.z, -v.w);
                    }
                }
            }
            __syncthreads();
        };
        auto store = [&](int o0) {
            if (dbg & 16) return;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                float2 *out = fj.out[o0 + o];
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int q = tid + it * NT;
                    if (q < F / 2) {
                        *reinterpret_cast<float4 *>(out + s1 * P + 2 * q) = ya[o][it];
                        if (pair) *reinterpret_cast<float4 *>(out + s2 * P + 2 * q) = yb[o][it];
                    }
                }
            }
        };
        xround(0, Aa);
        finish();
        if (fj.nout == 4) load_tables(tid);
        __builtin_amdgcn_sched_barrier(0);
        store(0);
        if (fj.nout == 4) xround(2, Ab);
        __builtin_amdgcn_sched_barrier(0);
        // The lines of the NEXT pair are requested here -- after the last consumer of anything the compiler may have
        // spilled (a reload is a vector-memory load: younger than the prefetch, it would wait for it) -- and committed at
        // the bottom of this same iteration: three quarters of the iteration run without their registers.
        if (more && !(dbg & 8)) {
            // (scalar loads: as vector loads these look-ups put an s_waitcnt vmcnt(0) -- the stores of store(0) and the table
            // loads -- in front of the prefetch.  ABAB on one box: 1.1677 / 1.1696 -> 1.1509 / 1.1427 ms per launch)
            n1 = oip_sload_i32(ypos, kn);
            n2 = oip_sload_i32(ypos, M - kn);
            m1 = VEXP ? (long)oip_sload_i32(fj.ypos_s, kn % fj.m) : n1;
            m2 = VEXP ? (long)oip_sload_i32(fj.ypos_s, (M - kn) % fj.m) : n2;
            kyc = kn;
            fetch(tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (fj.nout == 4) finish();
        if (more) commit(tid);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_tables(tid);             // ahead of the stores: waiting for them then waits for nothing younger
        __builtin_amdgcn_sched_barrier(0);
        if (fj.nout == 4) store(2);
        s1 = n1; s2 = n2;
        __syncthreads();
    }
}

// ---- row stage with the VERTICAL x4 up-sampling applied to the band spectra (padded row lengths) ---------------------
//
// The reference's own strips are 12288 pixels wide: ten slices of 1228 columns, band windows of 307, and
// cv::phaseCorrelate pads the 1228-column images to 1250 (preproc.h:302-316).  1250 is not four times anything the
// band window has, so the horizontal taps stay in the image domain -- but 16000 = 4 x 4000 still holds on the vertical
// axis: cv::resize interpolates horizontally first (f32, reproduced exactly by hpack_bands_kernel on the 4000 band
// rows), and the vertical operator V = C + E acts on those rows, so along a column
//     DFT_M(V h)[ky] = Hv[ky] DFT_m(h)[ky mod m] + sum_i Gv_i[ky] h[J_i],       J = {0, 1, m-2, m-1}
// exactly (corr_rows_up_kernel, VEXP).  The horizontally up-sampled bands of a pair of units are transformed at
// their own height (four arrays bX + i bY side by side, m x 4N) and line ky of an up-sampled band's column transform
// is rebuilt on the way into LDS.  Against the image-domain route a pair of units costs 2 array-sized forward column
// transforms instead of 5, no vertical up-sampling kernel and one row-stage launch (5 spectra in, 4 out) instead of two
// (3 in, 2 out each).
//
// LDS: five two-line buffers [PAN_A + i PAN_B | array 0 .. 3]; output o = C(PAN_u, band 2a) + i C(PAN_u, band 2a+1)
// (arrays 0, 1: unit A = real slot of the PAN buffer; arrays 2, 3: unit B) overwrites array o's buffer in place -- the
// thread of column kx is the only reader of bins (kx, ky) and (-kx, -ky).
struct VRowsJob {
    const float2 *zp;       // pitch P
    const float2 *zn;       // m rows x four arrays, zn_stride elements apart, pitch Pn (column transforms done)
    long zn_stride;
    int Pn;
    float2 *out[4];
    int nout;               // 4, or 2 for a single unit (arrays 2 and 3 are neither read nor written)
    int dbg;
    const float2 *vtab;     // [5][M]: Hv, Gv_0 .. Gv_3
    const float2 *raw;      // [4 rows][4 arrays][zn_stride]: (bX, bY) of the horizontally up-sampled band rows {0, 1, m-2, m-1}
    const int *ypos_s;      // row position of frequency line k in zn, k in [0, m)
    int m;
};

template <int F, int NT, int... Rs>
__global__ __launch_bounds__(NT, 2 * NT / 256) void corr_rows_v_kernel(VRowsJob fj, int a0, int part, int M, int P, const int *__restrict__ ypos,
                                                                      const float2 *__restrict__ twF)
{
    // One launch serves ONE unit: its PAN image (slot `part` of the packed PAN buffer) against band arrays a0, a0 + 1.
    // Three two-line buffers = 60 KB of LDS and <= 128 VGPRs, so two workgroups share a CU: one computes while the other
    // waits at a barrier or for its lines.  (All four arrays of a pair in one workgroup -- 105 KB, one workgroup per CU --
    // was measured first: 0.96 ms per pair against 2 x 0.33 ms for two launches of the image-domain kernel of this shape.)
    constexpr int TWN = oipfft::TwTable<F, Rs...>::value();
    constexpr int N = F, NARR = 3, NB = 2;
    constexpr int NIT = (N + NT - 1) / NT;
    static_assert(N % 2 == 0, "two points per lane");
    constexpr int NIT2 = (N / 2 + NT - 1) / NT;
    constexpr int NQ = NB * (N / 2);                // 16-byte pieces of one line of the two band arrays
    constexpr int NITN = (NQ + NT - 1) / NT;
    __shared__ __align__(16) float2 buf[NARR * 2 * F];   // [PAN | array a0 | array a0 + 1][point][line]: line 0 = ky, line 1 = -ky
    __shared__ float2 tw[TWN];
    float4 *buf4 = reinterpret_cast<float4 *>(buf);
    const int dbg = fj.dbg;
    const int half = M / 2;
    int ky = blockIdx.x;
    if (ky > half) return;
    for (int i = threadIdx.x; i < TWN; i += NT) tw[i] = twF[i];
    float4 la[NIT2], lb[NIT2], na[NITN], nb[NITN];
    float2 vt[5];
    long n1 = ypos[ky], n2 = ypos[ky ? M - ky : 0];
    long m1 = fj.ypos_s[ky % fj.m], m2 = fj.ypos_s[(ky ? M - ky : 0) % fj.m];
    int kyc = ky;                                   // the frequency line whose loads are in the registers
    const float2 *zn = fj.zn + (long)a0 * fj.zn_stride;
    const float2 *raw = fj.raw + (long)a0 * fj.zn_stride;
    auto fetch = [&](int tid) {                     // the PAN line pair, the band lines and their coefficients: one iteration ahead
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            int q = tid + it * NT;
            q = q < N / 2 ? q : N / 2 - 1;
            la[it] = *reinterpret_cast<const float4 *>(fj.zp + n1 * P + 2 * q);
            lb[it] = *reinterpret_cast<const float4 *>(fj.zp + n2 * P + 2 * q);
        }
#pragma unroll
        for (int it = 0; it < NITN; ++it) {
            int q = tid + it * NT;
            q = q < NQ ? q : NQ - 1;
            const int a = q / (N / 2), i = q - a * (N / 2);
            const float2 *z = zn + a * fj.zn_stride + 2 * i;
            na[it] = *reinterpret_cast<const float4 *>(z + m1 * fj.Pn);
            nb[it] = *reinterpret_cast<const float4 *>(z + m2 * fj.Pn);
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) vt[r] = fj.vtab[r * M + kyc];
    };
    auto commit = [&](int tid) {
#pragma unroll
        for (int it = 0; it < NIT2; ++it) {
            const int q = tid + it * NT;
            if (q < N / 2) {
                buf4[2 * q] = make_float4(la[it].x, la[it].y, lb[it].x, lb[it].y);
                buf4[2 * q + 1] = make_float4(la[it].z, la[it].w, lb[it].z, lb[it].w);
            }
        }
        // line ky of the vertically up-sampled band pair: Hv[ky] zn[ky mod m] + sum_i Gv_i[ky] raw_i; Hv and Gv of line
        // -ky are the conjugates.  The raw rows (160 KB for a pair of units: L2 hits) are read here, not held.
#pragma unroll
        for (int it = 0; it < NITN; ++it) {
            const int q = tid + it * NT;
            if (q >= NQ) continue;
            const int a = q / (N / 2), i = q - a * (N / 2);
            const float2 hv = vt[0];
            float2 x0 = oipfft::cmul(hv, make_float2(na[it].x, na[it].y)), x1 = oipfft::cmul(hv, make_float2(na[it].z, na[it].w));
            float2 y0 = cmulj(hv, make_float2(nb[it].x, nb[it].y)), y1 = cmulj(hv, make_float2(nb[it].z, nb[it].w));
            if (!(dbg & 32)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float4 rw = *reinterpret_cast<const float4 *>(raw + ((long)r * 4 + a) * fj.zn_stride + 2 * i);
                    const float2 gv = vt[1 + r];
                    const float2 w0 = make_float2(rw.x, rw.y), w1 = make_float2(rw.z, rw.w);
                    x0 = cfma(gv, w0, x0); x1 = cfma(gv, w1, x1);
                    y0 = cfmaj(gv, w0, y0); y1 = cfmaj(gv, w1, y1);
                }
            }
            buf4[(1 + a) * F + 2 * i] = make_float4(x0.x, x0.y, y0.x, y0.y);
            buf4[(1 + a) * F + 2 * i + 1] = make_float4(x1.x, x1.y, y1.x, y1.y);
        }
    };
    fetch(threadIdx.x);
    commit(threadIdx.x);
    long s1 = n1, s2 = n2;
    __syncthreads();
    for (; ky <= half; ky += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));               // per-iteration opaque copy: no stage addressing hoisted out of the loop
        const bool pair = s1 != s2;
        const int kn = ky + gridDim.x;
        const bool more = kn <= half;
        if (more && !(dbg & 8)) {
            // (plain loads here: this kernel requests its next lines at the TOP of the iteration, nothing of its own is in
            // flight yet, and the explicit scalar loads of corr_rows_up_kernel cost it 4.5 % -- 0.740 against 0.708 ms)
            n1 = ypos[kn];
            n2 = ypos[M - kn];
            m1 = fj.ypos_s[kn % fj.m];
            m2 = fj.ypos_s[(M - kn) % fj.m];
            kyc = kn;
            fetch(tid);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 1)) oipfft::StagesPipe<F, NT, NARR, 1, Rs...>::run(buf, tw, tid);
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            const int kx = tid + it * NT;
            if (kx >= N || (dbg & 2)) continue;
            const int nkx = kx ? N - kx : 0;
            const bool edge_col = (kx == 0) || (2 * kx == N);
            const bool real_bin = edge_col && (ky == 0 || 2 * ky == M);
            const float2 A = spec_of(part, buf[2 * kx], buf[2 * nkx + 1]);
#pragma unroll
            for (int a = 0; a < NB; ++a) {
                float2 *b = buf + (1 + a) * 2 * F;
                const float2 zk = b[2 * kx], zm = b[2 * nkx + 1];
                const float2 C1 = cross_power_bin_fast(A, spec_of(0, zk, zm), real_bin, edge_col);
                const float2 C2 = cross_power_bin_fast(A, spec_of(1, zk, zm), real_bin, edge_col);
                // Y = C1 + i C2 at the bin, conj(C1) + i conj(C2) at its mirror; inverse = conj(forward(conj(.)))
                b[2 * kx] = make_float2(C1.x - C2.y, -(C1.y + C2.x));
                b[2 * nkx + 1] = pair ? make_float2(C1.x + C2.y, -(C2.x - C1.y)) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        asm volatile("" : "+v"(tid));
        if (!(dbg & 4)) oipfft::StagesPipe<F, NT, NB, 1, Rs...>::run(buf + 2 * F, tw, tid);
        float4 ya[NB][NIT2], yb[NB][NIT2];          // line ky / line -ky, two points each
#pragma unroll
        for (int o = 0; o < NB; ++o) {
#pragma unroll
            for (int it = 0; it < NIT2; ++it) {
                const int q = tid + it * NT;
                if (q < N / 2) {
                    const float4 u = buf4[(1 + o) * F + 2 * q], v = buf4[(1 + o) * F + 2 * q + 1];
                    ya[o][it] = make_float4(u.x, -u.y, v.x, -v.y);
                    yb[o][it] = make_float4(u.z, -u.w, v.z, -v.w);
                }
            }
        }
        __syncthreads();
        if (more) commit(tid);
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 16)) {
#pragma unroll
            for (int o = 0; o < NB; ++o) {
                float2 *out = fj.out[a0 + o];
#pragma unroll
                for (int it = 0; it < NIT2; ++it) {
                    const int q = tid + it * NT;
                    if (q < N / 2) {
                        *reinterpret_cast<float4 *>(out + s1 * P + 2 * q) = ya[o][it];
                        if (pair) *reinterpret_cast<float4 *>(out + s2 * P + 2 * q) = yb[o][it];
                    }
                }
            }
        }
        s1 = n1; s2 = n2;
        __syncthreads();
    }
}

// Horizontal half of cv::resize(INTER_CUBIC) x4 on the u16 band windows of a launch (HResizeCubic: four taps, edge
// replicated, every product and sum a separate f32 rounding -- the order of resize_cubic_kernel), packed two bands per
// complex value: array a = H(bX) + i H(bY) occupies columns [a stride, a stride + cols) of an m x (4 stride) array of
// pitch Pn, columns [cols, stride) zero (cv::phaseCorrelate's right padding); rows {0, 1, m-2, m-1} also go to
// raw[r][a][x] for the vertical operator's edge terms.
struct HPackJob {
    const uint16_t *bx[4], *by[4];
    long pitch[4];
};
__global__ __launch_bounds__(128) void hpack_bands_kernel(HPackJob job, int m, int n, int cols, int stride, int Pn, const int *__restrict__ xofs,
                                                          const float4 *__restrict__ alpha, float2 *__restrict__ z, float2 *__restrict__ raw)
{
    const int x = blockIdx.x * 128 + threadIdx.x;
    const int a = blockIdx.z;
    if (x >= stride) return;
    const bool in = x < cols;
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        const int sx = xofs[x];
        w = alpha[x];
        c0 = sx - 1; c1 = sx; c2 = sx + 1; c3 = sx + 2;
        c0 = c0 < 0 ? 0 : (c0 > n - 1 ? n - 1 : c0);
        c1 = c1 < 0 ? 0 : (c1 > n - 1 ? n - 1 : c1);
        c2 = c2 < 0 ? 0 : (c2 > n - 1 ? n - 1 : c2);
        c3 = c3 < 0 ? 0 : (c3 > n - 1 ? n - 1 : c3);
    }
    const uint16_t *bx = job.bx[a], *by = job.by[a];
    const long pitch = job.pitch[a];
    auto taps = [&](const uint16_t *S) {
        float v = __fmul_rn((float)S[c0], w.x);
        v = __fadd_rn(v, __fmul_rn((float)S[c1], w.y));
        v = __fadd_rn(v, __fmul_rn((float)S[c2], w.z));
        v = __fadd_rn(v, __fmul_rn((float)S[c3], w.w));
        return v;
    };
#pragma unroll 2
    for (int y = blockIdx.y; y < m; y += gridDim.y) {
        float2 v = make_float2(0.f, 0.f);
        if (in) v = make_float2(taps(bx + y * pitch), taps(by + y * pitch));
        z[(long)y * Pn + (long)a * stride + x] = v;
        const int r = y < 2 ? y : (y >= m - 2 ? y - (m - 4) : -1);
        if (r >= 0) raw[((long)r * 4 + a) * stride + x] = v;
    }
}

struct FusedRow {
    int F, threads, fwd_threads;
    void (*fwd1)(FusedJob, int, int, const int *, const float2 *);        // one spectrum -> one output
    void (*fwd3)(FusedJob, int, int, const int *, const float2 *);        // three spectra -> two outputs
    void (*inv)(FusedJob, int, int, OipAxisDigits, const float2 *);
};
// power-of-two radices last: their stores are then contiguous in LDS (a leading radix-8 stage
// stores at a 128-byte stride, an 8-way bank conflict for ds_write_b64)
const FusedRow kFusedRow[] = {
    {3000, 512, 768, corr_rows_kernel<3000, 768, 1, 1, 3, 5, 5, 5, 8>, corr_rows_kernel<3000, 768, 3, 2, 3, 5, 5, 5, 8>,
     xpower_rows_kernel<3000, 512, 2, 3, 8, 5, 5, 5>},
    {1250, 256, 512, corr_rows_kernel<1250, 512, 1, 1, 5, 5, 5, 5, 2>, corr_rows_kernel<1250, 512, 3, 2, 5, 5, 5, 5, 2>,
     xpower_rows_kernel<1250, 256, 2, 2, 5, 5, 5, 5>},
    {200, 256, 256, corr_rows_kernel<200, 256, 1, 1, 5, 5, 8>, corr_rows_kernel<200, 256, 3, 2, 5, 5, 8>,
     xpower_rows_kernel<200, 256, 2, 8, 5, 5>},
};

// ---- peak: first maximum of the fftShift-ed surface + 5x5 weighted centroid ----------------------

// weightedCentroid(C, peak, Size(5,5), &response) (phasecorr.cpp) on the recomputed window;
// NaN marks window cells outside the image (the reference clamps the window to the image).
__global__ void centroid_kernel(const float *__restrict__ window, const long *__restrict__ key, int M, int N,
                                double *__restrict__ result, unsigned long long *__restrict__ slots)
{
    // last consumer of this surface's arg-max slots: leave them empty for the next surface
    // (both surfaces: the pass fills the imaginary one's slots even when only the real one is wanted)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 2 * kPeakSlots; i += blockDim.x) slots[i] = 0ull;
    if (threadIdx.x != 0) return;
    window += 32 * blockIdx.x;                       // one block per part
    key += blockIdx.x;
    result += 3 * blockIdx.x;
    const int py = (int)(*key / N), px = (int)(*key - (long)py * N);
    double cxs = 0.0, cys = 0.0, si = 0.0;
    for (int dy = 0; dy < 5; ++dy)
        for (int dx = 0; dx < 5; ++dx) {
            const float f = window[dy * 5 + dx];
            const int y = py - 2 + dy, x = px - 2 + dx;
            if (y < 0 || y >= M || x < 0 || x >= N) continue;
            const double v = (double)f;
            cxs = __dadd_rn(cxs, __dmul_rn((double)x, v));
            cys = __dadd_rn(cys, __dmul_rn((double)y, v));
            si = __dadd_rn(si, v);
        }
    double response = si;
    si = __dadd_rn(si, DBL_EPSILON);
    const double cx = cxs / si, cy = cys / si;
    response = response / (double)((long)M * N);
    result[0] = (double)N / 2.0 - cx;
    result[1] = (double)M / 2.0 - cy;
    result[2] = response;
}

// The same in one launch, without re-running the last inverse pass on 25 tiles per part: block `part` reduces the
// arg-max slots of its surface, evaluates the 25 window cells directly -- the last inverse column pass is
//     c(y, x) = sum_{n < F1} data[(y mod S) + S n][x] exp(+2 pi i n y / M)        (S = M / F1; nothing was stored by it)
// so a cell is a 128-term sum for the 16000-line geometry -- and runs weightedCentroid on them.  The cell values differ
// from the pass's own in the last bits (another summation order), like any two float transforms.
constexpr int kPeakWinThreads = 1024;
// up to four complex arrays per launch (the four outputs of a pair of units): block = array * nparts + part; array j uses
// the slot set slots + j * 2 * kPeakSlots
struct PeakWinJob {
    const float2 *data[4];
    double *result[4];
};
__global__ __launch_bounds__(kPeakWinThreads) void peak_window_kernel(PeakWinJob job, int M, int N, int P, int F1, int S,
                                                                     const float2 *__restrict__ twM, unsigned long long *__restrict__ slots,
                                                                     int nparts)
{
    constexpr int NW = kPeakWinThreads / 64;
    const int arr = blockIdx.x / nparts, part = blockIdx.x - arr * nparts;
    const float2 *__restrict__ data = job.data[arr];
    double *__restrict__ result = job.result[arr];
    slots += (long)arr * 2 * kPeakSlots;
    __shared__ unsigned long long sbest[NW];
    __shared__ float win[25];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long best = 0ull;
    for (int i = threadIdx.x; i < kPeakSlots; i += kPeakWinThreads) {
        const unsigned long long v = slots[part * kPeakSlots + i];
        best = v > best ? v : best;
        // last reader of this surface's slots: leave them empty for the next surface (the pass fills the imaginary
        // one's slots even when only the real surface is wanted)
        slots[part * kPeakSlots + i] = 0ull;
        if (nparts == 1) slots[kPeakSlots + i] = 0ull;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    if (lane == 0) sbest[wave] = best;
    __syncthreads();
    best = sbest[0];
    for (int w = 1; w < NW; ++w) best = sbest[w] > best ? sbest[w] : best;
    const long key = oip_peak_key(best, 0);           // an all-NaN surface has no entry: minMaxLoc leaves (0, 0)
    const int py = (int)(key / N), px = (int)(key - (long)py * N);
    // a wave per window cell (all loads of the window in flight together)
    for (int cell = wave; cell < 25; cell += NW) {
        const int ys = py - 2 + cell / 5, xs = px - 2 + cell % 5;
        float acc = 0.f;
        const bool inside = ys >= 0 && ys < M && xs >= 0 && xs < N;     // weightedCentroid clamps the window to the image
        if (inside) {
            int yo = ys - (M >> 1); if (yo < 0) yo += M;                 // fftShift
            int xo = xs - (N >> 1); if (xo < 0) xo += N;
            const int o1 = yo % S;
            for (int n = lane; n < F1; n += 64) {
                const float2 z = data[(long)(o1 + S * n) * P + xo];
                const float2 w = twM[(int)(((long)n * yo) % M)];         // exp(-2 pi i n yo / M); the sum wants its conjugate
                acc += part ? __fsub_rn(__fmul_rn(z.y, w.x), __fmul_rn(z.x, w.y)) : __fadd_rn(__fmul_rn(z.x, w.x), __fmul_rn(z.y, w.y));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if (lane == 0) win[cell] = inside ? acc : NAN;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double cxs = 0.0, cys = 0.0, si = 0.0;
    for (int dy = 0; dy < 5; ++dy)
        for (int dx = 0; dx < 5; ++dx) {
            const int y = py - 2 + dy, x = px - 2 + dx;
            if (y < 0 || y >= M || x < 0 || x >= N) continue;
            const double v = (double)win[dy * 5 + dx];
            cxs = __dadd_rn(cxs, __dmul_rn((double)x, v));
            cys = __dadd_rn(cys, __dmul_rn((double)y, v));
            si = __dadd_rn(si, v);
        }
    double response = si;
    si = __dadd_rn(si, DBL_EPSILON);
    const double cx = cxs / si, cy = cys / si;
    response = response / (double)((long)M * N);
    result[3 * part + 0] = (double)N / 2.0 - cx;
    result[3 * part + 1] = (double)M / 2.0 - cy;
    result[3 * part + 2] = response;
}

int optimal_dft_size(int n)
{
    // cv::getOptimalDFTSize: smallest 2^a 3^b 5^c >= n
    long best = -1;
    for (long p5 = 1; p5 < 2L * n + 1; p5 *= 5)
        for (long p35 = p5; p35 < 2L * n + 1; p35 *= 3) {
            long v = p35;
            while (v < n) v *= 2;
            if (best < 0 || v < best) best = v;
        }
    return (int)best;
}

OipAxisDigits digits_of(const std::vector<int> &f, int L)
{
    OipAxisDigits d;
    d.n = (int)f.size();
    d.L = L;
    for (int i = 0; i < 4; ++i) d.f[i] = i < d.n ? f[i] : 1;
    return d;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline bool smooth_len(long n) { for (int p : {2, 3, 5}) while (n % p == 0) n /= p; return n == 1; }

// Workspace carve-up for one correlation unit
struct PcWork {
    float2 *z[5];
    float2 *y[4];
    float *fa;          // base window, f32
    float *fb[8];       // second images, f32 (up-sampled bands; two units' worth)
    float *fsmall;      // MSS window before resize
    unsigned long long *slots;  // [4 arrays][2][kPeakSlots] arg-max slots, empty between surfaces
    long *keys;         // peak key scratch
    float *window;      // 25 floats
};

int carve(oip_ctx *ctx, const OipFft2dPlan *pl, int rows, int cols, int small_elems, int nz, int ny, int nfb, PcWork *w)
{
    const int M = pl->M;
    const size_t zbytes = align_up(sizeof(float2) * (size_t)M * pl->P, 256);
    const size_t fbytes = align_up(sizeof(float) * (size_t)rows * cols, 256);
    const size_t sbytes = align_up(sizeof(float) * (size_t)(small_elems > 0 ? small_elems : 1), 256);
    const size_t pbytes = align_up(sizeof(unsigned long long) * 4 * 2 * kPeakSlots, 256);
    size_t total = zbytes * (nz + ny) + fbytes * (1 + nfb) + sbytes + pbytes + 512;
    void *ws;
    int rc = oip_workspace(ctx, total, &ws);
    if (rc) return rc;
    char *p = (char *)ws;
    for (int i = 0; i < 5; ++i) { w->z[i] = i < nz ? (float2 *)p : nullptr; if (i < nz) p += zbytes; }
    for (int i = 0; i < 4; ++i) { w->y[i] = i < ny ? (float2 *)p : nullptr; if (i < ny) p += zbytes; }
    w->fa = (float *)p; p += fbytes;
    for (int i = 0; i < 8; ++i) { w->fb[i] = i < nfb ? (float *)p : nullptr; if (i < nfb) p += fbytes; }
    w->fsmall = (float *)p; p += sbytes;
    w->slots = (unsigned long long *)p; p += pbytes;
    w->keys = (long *)p; p += 256;
    w->window = (float *)p;
    // the slots must start empty (later surfaces are cleaned by the centroid kernel)
    ctx->prof_chain = nullptr;
    OIP_HIP(ctx, hipMemsetAsync(w->slots, 0, sizeof(unsigned long long) * 4 * 2 * kPeakSlots, ctx->stream));
    return OIP_OK;
}

int launch_window(oip_ctx *ctx, const uint16_t *img, size_t pitch, long row0, int col0, int rows, int cols, float *out)
{
    long n = (long)rows * cols;
    long blocks = (n + kBlock - 1) / kBlock;
    long cap = (long)ctx->cu_count * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    OipProfScope prof(ctx, "window_u16_to_f32_kernel");
    hipLaunchKernelGGL(window_u16_to_f32_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, img, pitch, row0,
                       col0, rows, cols, out);
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

int resize_tables(oip_ctx *ctx, int sw, int sh, int dw, int dh, const OipResizeTab **out)
{
    for (auto &t : ctx->resize_tabs)
        if (t.sw == sw && t.sh == sh && t.dw == dw && t.dh == dh) { *out = &t; return OIP_OK; }
    // cv::hal::resize coefficient set-up (resize.cpp), on the host in the same float/double mix
    const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<float> alpha((size_t)dw * 4), beta((size_t)dh * 4);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        xofs[dx] = sx;
        oip_interpolate_cubic_host(fx, &alpha[(size_t)dx * 4]);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        oip_interpolate_cubic_host(fy, &beta[(size_t)dy * 4]);
    }
    OipResizeTab t;
    t.sw = sw; t.sh = sh; t.dw = dw; t.dh = dh;
    t.d_xspec = nullptr; t.xspec_state = 0;
    t.d_yspec = nullptr; t.yspec_state = 0;
    // the x4 kernel needs every output's first tap inside its source pixel's 5-wide window
    t.x4 = dw == 4 * sw && dh == 4 * sh;
    for (int dx = 0; dx < dw && t.x4; ++dx) { int o = (xofs[dx] - 1) - (dx / 4 - 2); t.x4 = o == 0 || o == 1; }
    for (int dy = 0; dy < dh && t.x4; ++dy) { int o = (yofs[dy] - 1) - (dy / 4 - 2); t.x4 = o == 0 || o == 1; }
    t.x4h = 1;
    for (int dx = 0; dx < dw && t.x4h; ++dx) t.x4h = xofs[dx] == ((dx - 2) >> 2);
    t.x4v = 1;
    for (int dy = 0; dy < dh && t.x4v; ++dy) t.x4v = yofs[dy] == ((dy - 2) >> 2);
    OIP_HIP(ctx, hipMalloc((void **)&t.d_xofs, sizeof(int) * dw));
    OIP_HIP(ctx, hipMalloc((void **)&t.d_alpha, sizeof(float) * 4 * dw));
    OIP_HIP(ctx, hipMalloc((void **)&t.d_yofs, sizeof(int) * dh));
    OIP_HIP(ctx, hipMalloc((void **)&t.d_beta, sizeof(float) * 4 * dh));
    OIP_HIP(ctx, hipMemcpy(t.d_xofs, xofs.data(), sizeof(int) * dw, hipMemcpyHostToDevice));
    OIP_HIP(ctx, hipMemcpy(t.d_alpha, alpha.data(), sizeof(float) * 4 * dw, hipMemcpyHostToDevice));
    OIP_HIP(ctx, hipMemcpy(t.d_yofs, yofs.data(), sizeof(int) * dh, hipMemcpyHostToDevice));
    OIP_HIP(ctx, hipMemcpy(t.d_beta, beta.data(), sizeof(float) * 4 * dh, hipMemcpyHostToDevice));
    ctx->resize_tabs.push_back(t);
    *out = &ctx->resize_tabs.back();
    return OIP_OK;
}

template <typename SrcT>
int launch_resize(oip_ctx *ctx, const SrcT *src, long spitch, int sw, int sh, float *dst, int dw, int dh)
{
    const OipResizeTab *t;
    int rc = resize_tables(ctx, sw, sh, dw, dh, &t);
    if (rc) return rc;
    if (t->x4 && ((uintptr_t)dst & 15) == 0) {
        OipProfScope prof(ctx, "resize_cubic_x4_kernel");
        hipLaunchKernelGGL(resize_cubic_x4_kernel<SrcT>, dim3((sw + kBlock - 1) / kBlock, sh), dim3(kBlock), 0, ctx->stream,
                           src, spitch, sw, sh, dst, dw, t->d_xofs, reinterpret_cast<const float4 *>(t->d_alpha), t->d_yofs,
                           reinterpret_cast<const float4 *>(t->d_beta));
        OIP_HIP(ctx, hipGetLastError());
        return OIP_OK;
    }
    OipProfScope prof(ctx, "resize_cubic_kernel");
    hipLaunchKernelGGL(resize_cubic_kernel<SrcT>, dim3((dw + kBlock - 1) / kBlock, dh), dim3(kBlock), 0, ctx->stream, src,
                       spitch, sw, sh, dst, dw, dh, t->d_xofs, reinterpret_cast<const float4 *>(t->d_alpha), t->d_yofs,
                       reinterpret_cast<const float4 *>(t->d_beta));
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// The same for the exact x4 geometry (host-verified: yofs[dy] == (dy - 2) >> 2, even widths and pitch): a
// thread owns two adjacent columns and kV4Blocks source rows; the four outputs of source row p use rows
// p-2 .. p+2, so a 5-row window slides down one row per 4 outputs -- 1.25 loads of a pixel pair per 4 x 2
// outputs instead of 32 -- and the outputs leave as 8-byte stores.  Same taps, same f32 order.
constexpr int kV4Blocks = 8;        // source rows (= 4 output rows each) per thread
constexpr int kV4Threads = 128;
template <typename SrcT>
__global__ __launch_bounds__(kV4Threads) void resize_cubic_v_x4_kernel(VBatch<SrcT> vb, int sw, int sh, int dh,
                                                                      const float4 *__restrict__ beta)
{
    const int c = blockIdx.x * kV4Threads + threadIdx.x;            // column pair
    if (2 * c >= sw) return;
    const SrcT *__restrict__ src = vb.src[blockIdx.z] + 2 * c;
    float *__restrict__ dst = vb.dst[blockIdx.z] + 2 * c;
    const long spitch = vb.spitch[blockIdx.z];
    const int p0 = blockIdx.y * kV4Blocks;
    auto load = [&](int r) {                                        // clamped source row as two floats
        r = r < 0 ? 0 : (r > sh - 1 ? sh - 1 : r);
        const SrcT *q = src + (size_t)r * spitch;
        return make_float2((float)q[0], (float)q[1]);
    };
    float2 w0 = load(p0 - 2), w1 = load(p0 - 1), w2 = load(p0), w3 = load(p0 + 1), w4;
#pragma unroll
    for (int k = 0; k < kV4Blocks; ++k) {
        const int p = p0 + k;
        if (4 * p >= dh) break;
        w4 = load(p + 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dy = 4 * p + j;
            const float4 b = beta[dy];
            // rows (dy - 2 >> 2) - 1 ... + 2: p-2..p+1 for j = 0, 1 and p-1..p+2 for j = 2, 3
            const float2 t0 = j < 2 ? w0 : w1, t1 = j < 2 ? w1 : w2, t2 = j < 2 ? w2 : w3, t3 = j < 2 ? w3 : w4;
            float2 o;
            o.x = __fmul_rn(t0.x, b.x);
            o.x = __fadd_rn(o.x, __fmul_rn(t1.x, b.y));
            o.x = __fadd_rn(o.x, __fmul_rn(t2.x, b.z));
            o.x = __fadd_rn(o.x, __fmul_rn(t3.x, b.w));
            o.y = __fmul_rn(t0.y, b.x);
            o.y = __fadd_rn(o.y, __fmul_rn(t1.y, b.y));
            o.y = __fadd_rn(o.y, __fmul_rn(t2.y, b.z));
            o.y = __fadd_rn(o.y, __fmul_rn(t3.y, b.w));
            if (dy < dh) *reinterpret_cast<float2 *>(dst + (size_t)dy * sw) = o;
        }
        w0 = w1; w1 = w2; w2 = w3; w3 = w4;
    }
}

// vertical half only (see resize_cubic_v_kernel) of `count` equally shaped images in one launch; *tab
// carries the horizontal taps for the FFT loader
template <typename SrcT>
int launch_resize_v(oip_ctx *ctx, const SrcT *const *src, float *const *dst, int count, const long *spitch, int sw, int sh, int dw, int dh,
                    const OipResizeTab **tab)
{
    int rc = resize_tables(ctx, sw, sh, dw, dh, tab);
    if (rc) return rc;
    if (count < 1 || count > kVBatch) return oip_fail(ctx, OIP_E_RUNTIME, "launch_resize_v: bad batch");
    VBatch<SrcT> vb;
    for (int i = 0; i < kVBatch; ++i) { vb.src[i] = src[i < count ? i : 0]; vb.dst[i] = dst[i < count ? i : 0]; vb.spitch[i] = spitch[i < count ? i : 0]; }
    OipProfScope prof(ctx, "resize_cubic_v_kernel");
    bool aligned = (sw & 1) == 0 && dh == 4 * sh && (*tab)->x4v;
    for (int i = 0; i < count; ++i)
        aligned = aligned && (spitch[i] & 1) == 0 && ((size_t)src[i] % (2 * sizeof(SrcT))) == 0 && ((size_t)dst[i] & 7) == 0;
    { const char *e = getenv("OIP_V_GENERIC"); if (e && atoi(e)) aligned = false; }      // test knob: force the generic kernel
    if (aligned) {
        hipLaunchKernelGGL(resize_cubic_v_x4_kernel<SrcT>, dim3((sw / 2 + kV4Threads - 1) / kV4Threads, (sh + kV4Blocks - 1) / kV4Blocks, count),
                           dim3(kV4Threads), 0, ctx->stream, vb, sw, sh, dh, reinterpret_cast<const float4 *>((*tab)->d_beta));
        OIP_HIP(ctx, hipGetLastError());
        return OIP_OK;
    }
    hipLaunchKernelGGL(resize_cubic_v_kernel<SrcT>, dim3((sw + kBlock - 1) / kBlock, (dh + kVRows - 1) / kVRows, count), dim3(kBlock), 0,
                       ctx->stream, vb, sw, sh, dh, (*tab)->d_yofs, reinterpret_cast<const float4 *>((*tab)->d_beta));
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// H and G_j of the x4 cubic up-sampling n -> N = 4 n along one axis (0: horizontal, 1: vertical) as an operator on
// spectra (see corr_rows_up_kernel; oip_upsample_operator in host.cpp builds them), cached on the device per resize
// geometry.  *out stays null when the geometry has no such form.
int upsample_spectrum_tables(oip_ctx *ctx, const OipResizeTab *ctab, int axis, const float2 **out)
{
    OipResizeTab *t = const_cast<OipResizeTab *>(ctab);
    *out = nullptr;
    int &state = axis ? t->yspec_state : t->xspec_state;
    void *&d_spec = axis ? t->d_yspec : t->d_xspec;
    if (state == 1) { *out = (const float2 *)d_spec; return OIP_OK; }
    if (state < 0) return OIP_OK;
    state = -1;
    const int n = axis ? t->sh : t->sw, N = axis ? t->dh : t->dw;
    if (N != 4 * n || n < 8 || !(axis ? t->x4v : t->x4h)) return OIP_OK;
    std::vector<float2> tab((size_t)5 * N);
    if (oip_upsample_operator(n, reinterpret_cast<float *>(tab.data())) != OIP_OK) return OIP_OK;
    if (axis == 0)          // the horizontal tables carry the 1/2 of the unpacking of a packed band pair (spec2_of): exact
        for (auto &v : tab) { v.x *= 0.5f; v.y *= 0.5f; }
    OIP_HIP(ctx, hipMalloc(&d_spec, sizeof(float2) * tab.size()));
    OIP_HIP(ctx, hipMemcpy(d_spec, tab.data(), sizeof(float2) * tab.size(), hipMemcpyHostToDevice));
    state = 1;
    *out = (const float2 *)d_spec;
    return OIP_OK;
}

// one real image of a packed pair: an f32 image (pitch == cols) or a u16 raster window
struct RealSrc {
    const float *f32;
    const uint16_t *u16;
    long pitch16;
    const float *v;     // vertically up-sampled image; the horizontal taps are applied by the loader
};
inline RealSrc src_f32(const float *p) { return RealSrc{p, nullptr, 0, nullptr}; }
inline RealSrc src_u16(const uint16_t *p, long pitch) { return RealSrc{nullptr, p, pitch, nullptr}; }
inline RealSrc src_v(const float *p) { return RealSrc{nullptr, nullptr, 0, p}; }
inline RealSrc src_none() { return RealSrc{nullptr, nullptr, 0, nullptr}; }

// horizontal-tap tables of the V sources of a forward_packed call (all V sources of one call share them)
struct HTaps {
    int v_cols;
    const int *xofs;
    const float *alpha;
    int x4;
};

// forward transform of z = re + i im, the two f32 images read directly by the first pass;
// skip_rows: leave the row passes to the fused row-stage kernel
int forward_packed(oip_ctx *ctx, const OipFft2dPlan *pl, float2 *z, RealSrc re, RealSrc im, int rows, int cols, bool skip_rows,
                   const HTaps *vt = nullptr)
{
    OipFftIo io;
    memset(&io, 0, sizeof io);
    io.load_kind = 1;
    io.re = re.f32; io.re16 = re.u16; io.pitch_re16 = re.pitch16;
    io.im = im.f32; io.im16 = im.u16; io.pitch_im16 = im.pitch16;
    if (re.v || im.v) {
        if (!vt) return oip_fail(ctx, OIP_E_RUNTIME, "forward_packed: V source without horizontal taps");
        io.re_v = re.v; io.im_v = im.v;
        io.v_cols = vt->v_cols; io.xofs = vt->xofs; io.alpha = vt->alpha; io.x4 = vt->x4;
    }
    io.rows = rows; io.cols = cols;
    return oip_fft2d_exec(ctx, pl, z, 0, &io, skip_rows ? 1 : 0);
}

int launch_xpower(oip_ctx *ctx, float2 *out, const XpowerJob &job, const OipFft2dPlan *pl)
{
    OipProfScope prof(ctx, "cross_power_kernel");
    hipLaunchKernelGGL(cross_power_kernel, dim3((pl->N + kBlock - 1) / kBlock, pl->M / 2 + 1), dim3(kBlock), 0,
                       ctx->stream, out, job, pl->M, pl->N, pl->P, digits_of(pl->yf, pl->M), digits_of(pl->xf, pl->N));
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// How the row stage runs for this plan: 2 = forward rows + cross-power + inverse rows in one kernel,
// 1 = cross-power + inverse rows fused (forward rows as a separate pass), 0 = nothing fused.
// OIP_FUSED_ROWS=0|1|2 is an experiment knob.
struct RowStage {
    const FusedRow *k;
    int level;
};
RowStage row_stage(const OipFft2dPlan *pl)
{
    const char *env = getenv("OIP_FUSED_ROWS");          // read per call: a test switches it between two correlations
    int level = env ? atoi(env) : 2;
    RowStage rs{nullptr, 0};
    if (level <= 0 || pl->xf.size() != 1) return rs;
    for (const FusedRow &k : kFusedRow)
        if (k.F == pl->N) { rs.k = &k; rs.level = level > 1 ? 2 : 1; }
    return rs;
}

// Cross-power of ncorr (A, B) spectrum pairs into nout = ceil(ncorr / 2) arrays y[o] =
// C(2o) + i C(2o+1), with whatever row-stage fusion the plan allows.  At most three distinct
// spectra per call.
int xpower_stage(oip_ctx *ctx, const OipFft2dPlan *pl, const RowStage &rs, const SpecRef *a, const SpecRef *b, int ncorr,
                 float2 *const y[2])
{
    const int nout = (ncorr + 1) / 2;
    if (!rs.k) {
        for (int o = 0; o < nout; ++o) {
            XpowerJob job;
            memset(&job, 0, sizeof job);
            job.ncorr = ncorr - 2 * o > 1 ? 2 : 1;
            for (int c = 0; c < job.ncorr; ++c) { job.a[c] = a[2 * o + c]; job.b[c] = b[2 * o + c]; }
            int rc = launch_xpower(ctx, y[o], job, pl);
            if (rc) return rc;
        }
        return OIP_OK;
    }
    const float2 *twF;
    int rc = oip_fft_table(ctx, rs.k->F, &twF);
    if (rc) return rc;
    FusedJob fj;
    memset(&fj, 0, sizeof fj);
    fj.nout = nout;
    auto slot = [&](const float2 *z) {
        for (int i = 0; i < fj.narr; ++i) if (fj.z[i] == z) return i;
        if (fj.narr == 3) return -1;
        fj.z[fj.narr] = z;
        return fj.narr++;
    };
    for (int c = 0; c < ncorr; ++c) {
        fj.ia[c] = slot(a[c].z); fj.pa[c] = a[c].part;
        fj.ib[c] = slot(b[c].z); fj.pb[c] = b[c].part;
        if (fj.ia[c] < 0 || fj.ib[c] < 0) return oip_fail(ctx, OIP_E_RUNTIME, "xpower_stage: more than three spectra");
        fj.ncorr[c / 2]++;
    }
    for (int o = 0; o < nout; ++o) fj.out[o] = y[o];
    { const char *e = getenv("OIP_ROWS_DBG"); fj.dbg = e ? atoi(e) : 0; }
    if (rs.level == 2) {
        // persistent: as many workgroups as fit the CUs at once (LDS- or thread-limited)
        // the kernel hard-wires which spectrum and slot feeds which correlation
        const bool one = fj.narr == 1 && fj.nout == 1 && ncorr == 1 && fj.ia[0] == 0 && fj.pa[0] == 0 && fj.ib[0] == 0 && fj.pb[0] == 1;
        bool three = fj.narr == 3 && fj.nout == 2 && ncorr == 4;
        const int want_ib[4] = {0, 1, 1, 2}, want_pb[3] = {1, 0, 1};
        for (int c = 0; c < 4 && three; ++c)
            three = fj.ia[c] == 0 && fj.pa[c] == 0 && fj.ib[c] == want_ib[c] && (c == 3 || fj.pb[c] == want_pb[c]);
        if (!one && !three) return oip_fail(ctx, OIP_E_RUNTIME, "xpower_stage: unsupported job shape");
        const size_t lds = sizeof(float2) * ((size_t)fj.narr * 2 * rs.k->F + rs.k->F / 2);
        long per_cu = (long)(160 * 1024 / lds);
        if (per_cu > 2048 / rs.k->fwd_threads) per_cu = 2048 / rs.k->fwd_threads;
        if (per_cu < 1) per_cu = 1;
        long grid = (long)ctx->cu_count * per_cu;
        if (grid > pl->M / 2 + 1) grid = pl->M / 2 + 1;
        OipProfScope prof(ctx, "corr_rows_kernel");
        const char *env3 = getenv("OIP_ROWS3");                         // experiment knob: 0 = the five-stage kernel
        if (three && rs.k->F == 3000 && !(env3 && atoi(env3) == 0)) {
            long g3 = ctx->cu_count;
            if (g3 > pl->M / 2 + 1) g3 = pl->M / 2 + 1;
            hipLaunchKernelGGL((corr_rows3_kernel<3000, 768, 3, 2, 25, 15, 8>), dim3((unsigned)g3), dim3(768), 0, ctx->stream, fj, pl->M, pl->P,
                               pl->d_ypos, twF);
        } else
        hipLaunchKernelGGL(one ? rs.k->fwd1 : rs.k->fwd3, dim3((unsigned)grid), dim3(rs.k->fwd_threads), 0, ctx->stream, fj, pl->M,
                           pl->P, pl->d_ypos, twF);
    } else {
        OipProfScope prof(ctx, "xpower_rows_kernel");
        hipLaunchKernelGGL(rs.k->inv, dim3(pl->M / 2 + 1), dim3(rs.k->threads), 0, ctx->stream, fj, pl->M, pl->P,
                           digits_of(pl->yf, pl->M), twF);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// arg-max -> 5x5 window -> centroid of both parts (the real and imaginary surface) of `narr` arrays whose last inverse
// pass left per-tile maxima in slot sets 0 .. narr-1: one launch
int peak_windows(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, float2 *const *y, double *const *d_results, int narr, int nparts)
{
    const float2 *twM;
    int rc;
    if ((rc = oip_fft_table(ctx, pl->M, &twM))) return rc;
    const int F1 = pl->yf[0];
    PeakWinJob job;
    for (int j = 0; j < 4; ++j) { job.data[j] = y[j < narr ? j : 0]; job.result[j] = d_results[j < narr ? j : 0]; }
    OipProfScope prof(ctx, "peak_window_kernel");
    hipLaunchKernelGGL(peak_window_kernel, dim3(narr * nparts), dim3(kPeakWinThreads), 0, ctx->stream, job, pl->M, pl->N, pl->P, F1, pl->M / F1, twM,
                       w.slots, nparts);
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// inverse transform of y whose last pass only leaves per-tile maxima (in slot set `slot_set`), then -- unless `defer`:
// the caller batches the arrays of a launch through peak_windows -- for each wanted part:
// arg-max -> recompute the 5x5 window -> centroid -> result slot
int inverse_and_peaks(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, float2 *y, int nparts, double *d_results,
                      bool rows_done, int slot_set = 0, bool defer = false)
{
    OipFftIo io;
    memset(&io, 0, sizeof io);
    io.store_kind = 1;
    io.slots = w.slots + (long)slot_set * 2 * kPeakSlots;
    int rc = oip_fft2d_exec(ctx, pl, y, 1, &io, rows_done ? 1 : 0);
    if (rc) return rc;
    // arg-max -> 5x5 window -> centroid of all parts (the real and imaginary surface of y) in one launch
    // (OIP_WINDOW_FFT=1: the earlier form -- re-run the last pass on the 25 tiles of the window, then a centroid launch)
    static const char *envw = getenv("OIP_WINDOW_FFT");
    if (!(envw && atoi(envw))) {
        if (defer) return OIP_OK;
        float2 *const ys[1] = {y};
        double *const rs[1] = {d_results};
        PcWork w1 = w;
        w1.slots = io.slots;
        return peak_windows(ctx, pl, w1, ys, rs, 1, nparts);
    }
    OipFftIo wio;
    memset(&wio, 0, sizeof wio);
    wio.peak_key = w.keys;
    wio.slots = io.slots;
    wio.window = w.window;
    wio.part = nparts;
    if ((rc = oip_fft2d_window(ctx, pl, y, &wio))) return rc;
    {
        OipProfScope prof(ctx, "centroid_kernel");
        hipLaunchKernelGGL(centroid_kernel, dim3(nparts), dim3(64), 0, ctx->stream, w.window, w.keys, pl->M, pl->N, d_results, io.slots);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// one pair (a, b) -> result slot
int correlate_pair(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, RealSrc a, RealSrc b, int rows,
                   int cols, double *d_result)
{
    const RowStage rs = row_stage(pl);
    int rc = forward_packed(ctx, pl, w.z[0], a, b, rows, cols, rs.level == 2);
    if (rc) return rc;
    const SpecRef sa[1] = {{w.z[0], 0}}, sb[1] = {{w.z[0], 1}};
    float2 *const y[2] = {w.y[0], nullptr};
    if ((rc = xpower_stage(ctx, pl, rs, sa, sb, 1, y))) return rc;
    return inverse_and_peaks(ctx, pl, w, w.y[0], 1, d_result, rs.k != nullptr);
}

// base image (real slot of zp) against (imag slot of zp, both slots of zq, slot `last_part` of zl)
int correlate_four(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, const RowStage &rs, float2 *zp, float2 *zq,
                   float2 *zl, int last_part, double *d_results /* 4 x 3 */)
{
    const SpecRef sa[4] = {{zp, 0}, {zp, 0}, {zp, 0}, {zp, 0}};
    const SpecRef sb[4] = {{zp, 1}, {zq, 0}, {zq, 1}, {zl, last_part}};
    float2 *const y[2] = {w.y[0], w.y[1]};
    int rc;
    if ((rc = xpower_stage(ctx, pl, rs, sa, sb, 4, y))) return rc;
    static const char *envw = getenv("OIP_WINDOW_FFT");
    const bool batched = !(envw && atoi(envw));
    if ((rc = inverse_and_peaks(ctx, pl, w, w.y[0], 2, d_results, rs.k != nullptr, 0, batched))) return rc;
    if ((rc = inverse_and_peaks(ctx, pl, w, w.y[1], 2, d_results + 6, rs.k != nullptr, batched ? 1 : 0, batched))) return rc;
    if (!batched) return OIP_OK;
    double *const res[2] = {d_results, d_results + 6};
    return peak_windows(ctx, pl, w, y, res, 2, 2);
}

// base image a against four images b0..b3: 3 forward + 2 inverse complex transforms
int correlate_one_to_four(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, RealSrc a, const RealSrc b[4],
                          int rows, int cols, double *d_results /* 4 x 3 */, const HTaps *vt)
{
    const RowStage rs = row_stage(pl);
    const bool skip = rs.level == 2;
    int rc;
    if ((rc = forward_packed(ctx, pl, w.z[0], a, b[0], rows, cols, skip, vt))) return rc;
    if ((rc = forward_packed(ctx, pl, w.z[1], b[1], b[2], rows, cols, skip, vt))) return rc;
    if ((rc = forward_packed(ctx, pl, w.z[2], b[3], src_none(), rows, cols, skip, vt))) return rc;
    return correlate_four(ctx, pl, w, rs, w.z[0], w.z[1], w.z[2], 0, d_results);
}

// Two units (base image + four bands each) at once: the fourth bands of the two units share one
// complex transform (b3 of unit A in the real slot, b3 of unit B in the imaginary slot), so the
// pair costs 5 forward transforms instead of 6.
int correlate_two_units(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, RealSrc aA, const RealSrc bA[4], RealSrc aB,
                        const RealSrc bB[4], int rows, int cols, double *d_resA /* 4 x 3 */, double *d_resB, const HTaps *vt)
{
    const RowStage rs = row_stage(pl);
    const bool skip = rs.level == 2;
    int rc;
    if ((rc = forward_packed(ctx, pl, w.z[0], aA, bA[0], rows, cols, skip, vt))) return rc;
    if ((rc = forward_packed(ctx, pl, w.z[1], bA[1], bA[2], rows, cols, skip, vt))) return rc;
    if ((rc = forward_packed(ctx, pl, w.z[2], aB, bB[0], rows, cols, skip, vt))) return rc;
    if ((rc = forward_packed(ctx, pl, w.z[3], bB[1], bB[2], rows, cols, skip, vt))) return rc;
    if ((rc = forward_packed(ctx, pl, w.z[4], bA[3], bB[3], rows, cols, skip, vt))) return rc;
    if ((rc = correlate_four(ctx, pl, w, rs, w.z[0], w.z[1], w.z[4], 0, d_resA))) return rc;
    return correlate_four(ctx, pl, w, rs, w.z[2], w.z[3], w.z[4], 1, d_resB);
}

// The same pair of units with the horizontal up-sampling applied to the band spectra (corr_rows_up_kernel): one
// full-width forward transform (PAN_A + i PAN_B; aB may be absent), four quarter-width column transforms of the
// vertically up-sampled bands (V images, rows x band_cols, packed two by two), one row-stage launch, four inverse
// column transforms.  Against correlate_two_units: 2 instead of 5 array-sized forward column transforms, 6 instead of
// 10 row transforms of 3000 points (in units of one).
// One inter-band unit: a PAN window (rows x cols u16) and the matching windows of the four bands
// (band_rows x band_cols u16), each with its own pitch -- a window of the resident raster or a compact
// copy received from another rank.
struct IbUnit {
    const uint16_t *pan;
    long pan_pitch;
    const uint16_t *band[OIP_MSS_BANDS];
    long band_pitch;
};

struct UpPath {
    const OipFft2dPlan *narrow;     // M x band_cols (vertical taps in the image domain), or
    const OipFft2dPlan *small;      // band_rows x (4 band_cols): both axes on the spectra (vtab set)
    const float2 *xtab, *vtab;
    bool vonly;                     // padded row lengths (1228 -> 1250): horizontal taps in the image domain, vertical axis on
                                    // the spectra (corr_rows_v_kernel); small = band_rows x (4 N)
};

// The band windows of a launch as one complex array: array a = bX + i bY occupies columns [n a, n a + n) of an
// m x (4 n) array of pitch Pn (the column passes then run over all four at once); rows {0, 1, m-2, m-1} of each array
// also go to raw16[r][a n / 2 + p / 2] as (bX | bY << 16) of the points p, p + 1 (n even).
struct BandPackJob {
    const uint16_t *bx[4], *by[4];
    long pitch[4];
};
template <bool PAIRS>        // PAIRS: two neighbouring columns per thread through 4-byte loads (host-checked alignment)
__global__ __launch_bounds__(128) void pack_bands_kernel(BandPackJob job, int m, int n, int Pn, float2 *__restrict__ z, unsigned *__restrict__ raw16)
{
    constexpr int W = PAIRS ? 2 : 1;
    const int p = W * (blockIdx.x * 128 + threadIdx.x);
    const int a = blockIdx.z;
    if (p >= n) return;
    const uint16_t *bx = job.bx[a] + p, *by = job.by[a] + p;
    const long pitch = job.pitch[a];
#pragma unroll 4
    for (int y = blockIdx.y; y < m; y += gridDim.y) {
        const int r = y < 2 ? y : (y >= m - 2 ? y - (m - 4) : -1);
        if (PAIRS) {
            const unsigned ux = *reinterpret_cast<const unsigned *>(bx + y * pitch), uy = *reinterpret_cast<const unsigned *>(by + y * pitch);
            *reinterpret_cast<float4 *>(z + (long)y * Pn + a * n + p) =
                make_float4((float)(ux & 0xffffu), (float)(uy & 0xffffu), (float)(ux >> 16), (float)(uy >> 16));
            if (r >= 0) *reinterpret_cast<uint2 *>(raw16 + ((long)r * 4 + a) * n + p) = make_uint2((ux & 0xffffu) | (uy << 16), (ux >> 16) | (uy & 0xffff0000u));
        } else {
            const unsigned ux = bx[y * pitch], uy = by[y * pitch];
            z[(long)y * Pn + a * n + p] = make_float2((float)ux, (float)uy);
            if (r >= 0) raw16[((long)r * 4 + a) * n + p] = ux | (uy << 16);
        }
    }
}

int correlate_units_up(oip_ctx *ctx, const OipFft2dPlan *pl, const UpPath &up, const PcWork &w, RealSrc aA, RealSrc aB, const RealSrc *v /* 4 or 8: V images */,
                       const IbUnit *const *units, int nunits, int rows, int cols, int band_rows, int band_cols, double *d_resA, double *d_resB)
{
    int rc;
    if ((rc = forward_packed(ctx, pl, w.z[0], aA, nunits > 1 ? aB : src_none(), rows, cols, true))) return rc;
    const int narr = 2 * nunits;
    long zn_stride;
    int Pn;
    const int *ypos_s = nullptr;
    unsigned *raw = nullptr;
    if (up.vtab) {
        // band windows -> one m x 3000 complex array (four arrays side by side) -> its column passes
        Pn = up.small->P;
        zn_stride = band_cols;
        raw = reinterpret_cast<unsigned *>(w.z[1] + (long)up.small->M * Pn);
        BandPackJob bj;
        for (int a = 0; a < 4; ++a) {
            const IbUnit *u = units[(a < narr ? a : 0) / 2];
            bj.bx[a] = u->band[2 * (a & 1)]; bj.by[a] = u->band[2 * (a & 1) + 1]; bj.pitch[a] = u->band_pitch;
        }
        {
            OipProfScope prof(ctx, "pack_bands_kernel");
            bool pairs = (band_cols & 1) == 0 && (Pn & 1) == 0;
            for (int a = 0; a < 4; ++a)
                pairs = pairs && (bj.pitch[a] & 1) == 0 && ((size_t)bj.bx[a] & 3) == 0 && ((size_t)bj.by[a] & 3) == 0;
            const int gy = band_rows < 128 ? band_rows : 128;
            if (pairs)
                hipLaunchKernelGGL(pack_bands_kernel<true>, dim3((band_cols / 2 + 127) / 128, gy, 4), dim3(128), 0, ctx->stream, bj, band_rows, band_cols, Pn,
                                   w.z[1], raw);
            else
                hipLaunchKernelGGL(pack_bands_kernel<false>, dim3((band_cols + 127) / 128, gy, 4), dim3(128), 0, ctx->stream, bj, band_rows, band_cols, Pn,
                                   w.z[1], raw);
            OIP_HIP(ctx, hipGetLastError());
        }
        ctx->prof_tag = "_band";
        rc = oip_fft2d_exec(ctx, up.small, w.z[1], 0, nullptr, 1);
        ctx->prof_tag = nullptr;
        if (rc) return rc;
        ypos_s = up.small->d_ypos;
    } else {
        Pn = up.narrow->P;
        zn_stride = (long)up.narrow->M * Pn;
        ctx->prof_tag = "_quarter";
        for (int a = 0; a < narr && !rc; ++a)
            rc = forward_packed(ctx, up.narrow, w.z[1] + a * zn_stride, src_f32(v[2 * a].v), src_f32(v[2 * a + 1].v), rows, band_cols, true);
        ctx->prof_tag = nullptr;
        if (rc) return rc;
    }
    const float2 *twF, *twS;
    if ((rc = oip_fft_table(ctx, 3000, &twF)) || (rc = oip_fft_table(ctx, 750, &twS))) return rc;
    UpRowsJob fj;
    memset(&fj, 0, sizeof fj);
    fj.zp = w.z[0];
    fj.zn = w.z[1];
    fj.zn_stride = zn_stride;
    fj.Pn = Pn;
    for (int o = 0; o < 4; ++o) fj.out[o] = w.y[o];
    fj.xtab = up.xtab;
    fj.nout = narr;
    fj.vtab = up.vtab;
    fj.raw16 = reinterpret_cast<const uint2 *>(raw);
    fj.ypos_s = ypos_s;
    fj.m = band_rows;
    { const char *e = getenv("OIP_ROWS_DBG"); fj.dbg = e ? atoi(e) : 0; }
    {
        OipProfScope prof(ctx, "corr_rows_up_kernel");
        long grid = ctx->cu_count;
        if (grid > pl->M / 2 + 1) grid = pl->M / 2 + 1;
        const char *et = getenv("OIP_UP_THREADS");                      // experiment knob: 512 | 768 threads
        const dim3 g((unsigned)grid);
        if (up.vtab && et && atoi(et) == 768)
            hipLaunchKernelGGL((corr_rows_up_kernel<768, true>), g, dim3(768), 0, ctx->stream, fj, pl->M, pl->P, pl->d_ypos, twF, twS);
        else if (up.vtab)       // measured in bench.py: 1.275 ms with 512 threads (244 VGPRs, no scratch), 1.36 with 768 (168, 28 spilled)
            hipLaunchKernelGGL((corr_rows_up_kernel<512, true>), g, dim3(512), 0, ctx->stream, fj, pl->M, pl->P, pl->d_ypos, twF, twS);
        else if (et && atoi(et) == 768)
            hipLaunchKernelGGL((corr_rows_up_kernel<768, false>), g, dim3(768), 0, ctx->stream, fj, pl->M, pl->P, pl->d_ypos, twF, twS);
        else
            hipLaunchKernelGGL((corr_rows_up_kernel<512, false>), g, dim3(512), 0, ctx->stream, fj, pl->M, pl->P, pl->d_ypos, twF, twS);
        OIP_HIP(ctx, hipGetLastError());
    }
    // the inverse column passes of the outputs, each into its own slot set; one window launch for all of them
    // (with OIP_WINDOW_FFT=1 every output still runs its own window passes: slot set 0 each time)
    static const char *envw = getenv("OIP_WINDOW_FFT");
    const bool batched = !(envw && atoi(envw));
    double *res[4];
    for (int o = 0; o < narr; ++o) {
        res[o] = (o < 2 ? d_resA : d_resB) + 6 * (o & 1);
        if ((rc = inverse_and_peaks(ctx, pl, w, w.y[o], 2, res[o], true, batched ? o : 0, batched))) return rc;
    }
    if (batched) return peak_windows(ctx, pl, w, w.y, res, narr, 2);
    return OIP_OK;
}

// A pair of units (or one) of a padded geometry -- the reference's 12288-wide strips: 1228-column slices, 1250-point
// rows -- with the vertical up-sampling on the spectra (corr_rows_v_kernel): one full-height forward column transform
// (PAN_A + i PAN_B), the horizontally up-sampled bands transformed at their own height (four arrays side by side), one
// row-stage launch, four inverse column transforms.
int correlate_units_vup(oip_ctx *ctx, const OipFft2dPlan *pl, const UpPath &up, const PcWork &w, const OipResizeTab *tab, RealSrc aA, RealSrc aB,
                        const IbUnit *const *units, int nunits, int rows, int cols, int band_rows, int band_cols, double *d_resA, double *d_resB)
{
    int rc;
    if ((rc = forward_packed(ctx, pl, w.z[0], aA, nunits > 1 ? aB : src_none(), rows, cols, true))) return rc;
    const int narr = 2 * nunits;
    const int N = pl->N, Pn = up.small->P;
    // z[1] holds the m x (4 N) band array; the four raw rows of the four arrays live in the small scratch (carve: 32 N floats)
    float2 *zn = w.z[1];
    float2 *raw = reinterpret_cast<float2 *>(w.fsmall);
    if ((long)up.small->M * Pn > (long)pl->M * pl->P) return oip_fail(ctx, OIP_E_RUNTIME, "correlate_units_vup: band array exceeds its slot");
    HPackJob hj;
    for (int a = 0; a < 4; ++a) {
        const IbUnit *u = units[(a < narr ? a : 0) / 2];
        hj.bx[a] = u->band[2 * (a & 1)]; hj.by[a] = u->band[2 * (a & 1) + 1]; hj.pitch[a] = u->band_pitch;
    }
    {
        OipProfScope prof(ctx, "hpack_bands_kernel");
        const int gy = band_rows < 256 ? band_rows : 256;
        hipLaunchKernelGGL(hpack_bands_kernel, dim3((N + 127) / 128, gy, 4), dim3(128), 0, ctx->stream, hj, band_rows, band_cols, cols, N, Pn, tab->d_xofs,
                           reinterpret_cast<const float4 *>(tab->d_alpha), zn, raw);
        OIP_HIP(ctx, hipGetLastError());
    }
    ctx->prof_tag = "_band";
    rc = oip_fft2d_exec(ctx, up.small, zn, 0, nullptr, 1);
    ctx->prof_tag = nullptr;
    if (rc) return rc;
    const float2 *twF;
    if ((rc = oip_fft_table(ctx, N, &twF))) return rc;
    VRowsJob fj;
    memset(&fj, 0, sizeof fj);
    fj.zp = w.z[0];
    fj.zn = zn;
    fj.zn_stride = N;
    fj.Pn = Pn;
    for (int o = 0; o < 4; ++o) fj.out[o] = w.y[o];
    fj.nout = narr;
    fj.vtab = up.vtab;
    fj.raw = raw;
    fj.ypos_s = up.small->d_ypos;
    fj.m = band_rows;
    { const char *e = getenv("OIP_ROWS_DBG"); fj.dbg = e ? atoi(e) : 0; }
    {
        OipProfScope prof(ctx, "corr_rows_v_kernel");
        const char *eg = getenv("OIP_VROWS_WG_PER_CU");        // experiment knob: 1 = one workgroup per CU (what co-residency buys)
        long grid = (eg && atoi(eg) == 1 ? 1L : 2L) * ctx->cu_count;   // two workgroups per CU (60 KB of LDS, <= 128 VGPRs each)
        if (grid > pl->M / 2 + 1) grid = pl->M / 2 + 1;
        if (N != 1250) return oip_fail(ctx, OIP_E_RUNTIME, "correlate_units_vup: no row stage for %d-point rows", N);
        for (int u = 0; u < nunits; ++u)
            hipLaunchKernelGGL((corr_rows_v_kernel<1250, 512, 5, 5, 5, 5, 2>), dim3((unsigned)grid), dim3(512), 0, ctx->stream, fj, 2 * u, u, pl->M, pl->P,
                               pl->d_ypos, twF);
        OIP_HIP(ctx, hipGetLastError());
    }
    double *res[4];
    for (int o = 0; o < narr; ++o) {
        res[o] = (o < 2 ? d_resA : d_resB) + 6 * (o & 1);
        if ((rc = inverse_and_peaks(ctx, pl, w, w.y[o], 2, res[o], true, o, true))) return rc;
    }
    return peak_windows(ctx, pl, w, w.y, res, narr, 2);
}

int fetch_results(oip_ctx *ctx, int count, double *host_out)
{
    ctx->prof_chain = nullptr;
    OIP_HIP(ctx, hipMemcpyAsync(ctx->h_small, ctx->d_small, sizeof(double) * count, hipMemcpyDeviceToHost, ctx->stream));
    OIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(host_out, ctx->h_small, sizeof(double) * count);
    return OIP_OK;
}

}  // namespace

extern "C" int oip_window_u16_to_f32(oip_ctx *ctx, const uint16_t *d_img, size_t pitch, long row0, int col0, int rows,
                                     int cols, float *d_out)
{
    OIP_CHECK_CTX(ctx);
    if (!d_img || !d_out || rows <= 0 || cols <= 0 || row0 < 0 || col0 < 0 || (size_t)(col0 + cols) > pitch)
        return oip_fail(ctx, OIP_E_INVALID, "oip_window_u16_to_f32: bad argument");
    return launch_window(ctx, d_img, pitch, row0, col0, rows, cols, d_out);
}

extern "C" int oip_resize_cubic_f32(oip_ctx *ctx, const float *d_src, int sw, int sh, float *d_dst, int dw, int dh)
{
    OIP_CHECK_CTX(ctx);
    if (!d_src || !d_dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0 || dh > 65535)
        return oip_fail(ctx, OIP_E_INVALID, "oip_resize_cubic_f32: bad argument");
    return launch_resize<float>(ctx, d_src, sw, sw, sh, d_dst, dw, dh);
}

extern "C" int oip_phase_correlate_f32(oip_ctx *ctx, const float *d_a, const float *d_b, int rows, int cols, double *dx,
                                       double *dy, double *response)
{
    OIP_CHECK_CTX(ctx);
    if (!d_a || !d_b || rows <= 0 || cols <= 0) return oip_fail(ctx, OIP_E_INVALID, "oip_phase_correlate_f32: bad argument");
    const int M = optimal_dft_size(rows), N = optimal_dft_size(cols);
    if (M > 65535) return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_phase_correlate_f32: more than 65535 rows");
    const OipFft2dPlan *pl;
    int rc = oip_fft2d_plan(ctx, M, N, &pl);
    if (rc) return rc;
    PcWork w;
    if ((rc = carve(ctx, pl, 1, 1, 0, 1, 1, 0, &w))) return rc;
    double *d_res = (double *)ctx->d_small;
    if ((rc = correlate_pair(ctx, pl, w, src_f32(d_a), src_f32(d_b), rows, cols, d_res))) return rc;
    double r[3];
    if ((rc = fetch_results(ctx, 3, r))) return rc;
    if (dx) *dx = r[0];
    if (dy) *dy = r[1];
    if (response) *response = r[2];
    return OIP_OK;
}

// CCD-pair correlations of n window pairs (the loop body of stitcher.h:166-191 after the colRange): window i is
// rows x cols u16 at a[i] / b[i] with its own pitch; results go to d_res[3 * i]
static int stt_pairs(oip_ctx *ctx, const uint16_t *const *a, const size_t *pitch_a, const uint16_t *const *b,
                     const size_t *pitch_b, int n, int rows, int cols, double *host_out)
{
    const int M = optimal_dft_size(rows), N = optimal_dft_size(cols);
    if (M > 65535) return oip_fail(ctx, OIP_E_UNSUPPORTED, "correlation window taller than 65535 lines");
    const OipFft2dPlan *pl;
    int rc = oip_fft2d_plan(ctx, M, N, &pl);
    if (rc) return rc;
    PcWork w;
    if ((rc = carve(ctx, pl, 1, 1, 0, 1, 1, 0, &w))) return rc;
    if ((rc = oip_small(ctx, sizeof(double) * 3 * (size_t)(n > 0 ? n : 1)))) return rc;
    double *d_res = (double *)ctx->d_small;
    for (int i = 0; i < n; ++i)
        if ((rc = correlate_pair(ctx, pl, w, src_u16(a[i], (long)pitch_a[i]), src_u16(b[i], (long)pitch_b[i]), rows, cols, d_res + 3 * i)))
            return rc;
    if (n > 0 && (rc = fetch_results(ctx, 3 * n, host_out))) return rc;
    return OIP_OK;
}

extern "C" int oip_stt_correlate_windows(oip_ctx *ctx, const uint16_t *const *d_a, const size_t *pitch_a,
                                         const uint16_t *const *d_b, const size_t *pitch_b, int n, int rows, int cols,
                                         double *out)
{
    OIP_CHECK_CTX(ctx);
    if (n < 0 || (n > 0 && (!d_a || !d_b || !pitch_a || !pitch_b || !out)) || rows <= 0 || cols <= 0)
        return oip_fail(ctx, OIP_E_INVALID, "oip_stt_correlate_windows: bad argument");
    for (int i = 0; i < n; ++i)
        if (!d_a[i] || !d_b[i] || pitch_a[i] < (size_t)cols || pitch_b[i] < (size_t)cols)
            return oip_fail(ctx, OIP_E_INVALID, "oip_stt_correlate_windows: window %d has a null pointer or a pitch below %d", i, cols);
    return stt_pairs(ctx, d_a, pitch_a, d_b, pitch_b, n, rows, cols, out);
}

extern "C" int oip_stt_correlate(oip_ctx *ctx, const uint16_t *d_pan1, const uint16_t *d_pan2, int W, long L, long row0,
                                 long nrows, int sections, int lines_per_section, int overlap_cols, int edge_cols,
                                 double *out)
{
    OIP_CHECK_CTX(ctx);
    if (!d_pan1 || !d_pan2 || !out || W <= 0 || L <= 0 || sections <= 0 || lines_per_section <= 0 || overlap_cols <= 0 ||
        overlap_cols > W || edge_cols < 0 || edge_cols >= overlap_cols || sections > 100000)
        return oip_fail(ctx, OIP_E_INVALID, "oip_stt_correlate: bad argument");
    // stitcher.h:75-77
    if (L < (long)sections * lines_per_section)
        return oip_fail(ctx, OIP_E_INVALID, "PAN line count less than sections times line-per-section, use smaller -s and/or -l value(s)");
    const int rows = lines_per_section, cols = overlap_cols - edge_cols;
    // stitcher.h:151-152, :167
    const long gap = (L - (long)sections * lines_per_section) / (sections + 1);
    const long step = gap + lines_per_section;
    std::vector<int> which;
    std::vector<const uint16_t *> a, b;
    for (int s = 0; s < sections; ++s) {
        const long off = gap + (long)s * step;
        if (off < row0 || off + rows > row0 + nrows) continue;     // another rank's section
        which.push_back(s);
        // stitcher.h:175-176: PAN1 cols [W-ov, W-edge), PAN2 cols [edge, ov); the u16->f32
        // conversion happens in the first FFT pass' loader
        a.push_back(d_pan1 + (size_t)(off - row0) * W + (W - overlap_cols));
        b.push_back(d_pan2 + (size_t)(off - row0) * W + edge_cols);
    }
    std::vector<size_t> pitch(which.size(), (size_t)W);
    std::vector<double> r(3 * which.size() + 3);
    int rc = stt_pairs(ctx, a.data(), pitch.data(), b.data(), pitch.data(), (int)which.size(), rows, cols, r.data());
    if (rc) return rc;
    for (int s = 0; s < sections; ++s)
        for (int k = 0; k < 3; ++k) out[3 * s + k] = NAN;
    for (size_t i = 0; i < which.size(); ++i)
        for (int k = 0; k < 3; ++k) out[3 * which[i] + k] = r[3 * i + k];
    return OIP_OK;
}

// The loop body of preproc.h:262-329 for n units: per unit and band (dx, dy, response) -> host_out[12 * u + 3 * b].
// Units go two at a time (five forward and four inverse transforms per pair).
static int interband_units(oip_ctx *ctx, const IbUnit *units, int n, int rows, int cols, int band_rows, int band_cols,
                           double *host_out)
{
    const int M = optimal_dft_size(rows), N = optimal_dft_size(cols);
    if (M > 65535) return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_interband_correlate: more than 65535 correlation lines");
    const OipFft2dPlan *pl;
    int rc = oip_fft2d_plan(ctx, M, N, &pl);
    if (rc) return rc;
    // PAN window: read as u16 by the FFT loader.  MSS windows: the vertical cubic pass runs as a kernel
    // (u16 -> f32, rows x band_cols), the horizontal pass inside the FFT loader -- or, for the exact x4 geometry at
    // 3000 columns, on the band spectra in the row stage (corr_rows_up_kernel; OIP_SPECTRAL_UP=0 keeps the loader).
    const OipResizeTab *tab = nullptr;
    if ((rc = resize_tables(ctx, band_cols, band_rows, cols, rows, &tab))) return rc;
    UpPath up{nullptr, nullptr, nullptr, nullptr, false};
    {
        // OIP_SPECTRAL_UP: 0 = image domain, 1 = horizontal axis on the spectra, 2 (default) = both axes
        const char *e = getenv("OIP_SPECTRAL_UP");
        const int want = e ? atoi(e) : 2;
        if (want > 0 && rows == 4 * band_rows && cols == 4 * band_cols && M == rows && N == cols && N == 3000 && row_stage(pl).level == 2) {
            if ((rc = upsample_spectrum_tables(ctx, tab, 0, &up.xtab))) return rc;
            if (up.xtab) {
                if ((rc = oip_fft2d_plan(ctx, M, band_cols, &up.narrow))) return rc;
                if (4 * up.narrow->P > pl->P || up.narrow->N != band_cols) up.xtab = nullptr;      // the four narrow arrays share one array-sized slot
            }
            if (up.xtab && want > 1 && optimal_dft_size(band_rows) == band_rows && band_rows >= 8) {
                if ((rc = upsample_spectrum_tables(ctx, tab, 1, &up.vtab))) return rc;
                if (up.vtab && (rc = oip_fft2d_plan(ctx, band_rows, cols, &up.small))) return rc;      // four arrays side by side
            }
        }
        // the reference's own 12288-wide strips: 1228-column slices padded to 1250-point rows.  The horizontal taps stay in
        // the image domain (1250 is not 4 x the band width); the vertical axis (16000 = 4 x 4000) goes to the spectra
        // (corr_rows_v_kernel).  OIP_SPECTRAL_V=0 keeps the image-domain route.
        const char *ev = getenv("OIP_SPECTRAL_V");
        if (!up.xtab && !(ev && atoi(ev) == 0) && want > 0 && rows == 4 * band_rows && cols == 4 * band_cols && M == rows && N == 1250 &&
            (cols & 1) == 0 && row_stage(pl).level == 2 && optimal_dft_size(band_rows) == band_rows && band_rows >= 8 && smooth_len(4 * N)) {
            if ((rc = upsample_spectrum_tables(ctx, tab, 1, &up.vtab))) return rc;
            if (up.vtab) {
                if ((rc = oip_fft2d_plan(ctx, band_rows, 4 * N, &up.small))) return rc;
                up.vonly = (long)up.small->M * up.small->P <= (long)pl->M * pl->P;
                if (!up.vonly) up.vtab = nullptr;
            }
        }
    }
    PcWork w;
    const bool spectral = up.xtab || up.vonly;
    if ((rc = carve(ctx, pl, rows, band_cols, up.vonly ? 32 * N : 0, spectral ? 2 : 5, spectral ? 4 : 2, 8, &w))) return rc;      // f32 scratch: the V images
    if ((rc = oip_small(ctx, sizeof(double) * 12 * (size_t)(n > 0 ? n : 1)))) return rc;
    double *d_res = (double *)ctx->d_small;
    // vertical passes of all bands of one or two units in one launch
    auto upsample = [&](const IbUnit *const *uns, int nun, float *const *fb, RealSrc *out) -> int {
        const uint16_t *srcs[kVBatch];
        long pitches[kVBatch];
        for (int u = 0; u < nun; ++u)
            for (int b = 0; b < OIP_MSS_BANDS; ++b) {
                srcs[4 * u + b] = uns[u]->band[b];
                pitches[4 * u + b] = uns[u]->band_pitch;
                out[4 * u + b] = src_v(fb[4 * u + b]);
            }
        return launch_resize_v<uint16_t>(ctx, srcs, fb, 4 * nun, pitches, band_cols, band_rows, cols, rows, &tab);
    };
    int k = 0;
    RealSrc sAB[8];
    for (; k + 1 < n; k += 2) {
        const IbUnit &A = units[k], &B = units[k + 1];
        const IbUnit *two[2] = {&A, &B};
        if (up.vonly) {
            if ((rc = correlate_units_vup(ctx, pl, up, w, tab, src_u16(A.pan, A.pan_pitch), src_u16(B.pan, B.pan_pitch), two, 2, rows, cols, band_rows,
                                          band_cols, d_res + 12 * k, d_res + 12 * (k + 1)))) return rc;
            continue;
        }
        if (!up.vtab && (rc = upsample(two, 2, w.fb, sAB))) return rc;
        if (up.xtab) {
            if ((rc = correlate_units_up(ctx, pl, up, w, src_u16(A.pan, A.pan_pitch), src_u16(B.pan, B.pan_pitch), sAB, two, 2, rows, cols, band_rows,
                                         band_cols, d_res + 12 * k, d_res + 12 * (k + 1)))) return rc;
            continue;
        }
        const HTaps vt{band_cols, tab->d_xofs, tab->d_alpha, tab->x4h};
        if ((rc = correlate_two_units(ctx, pl, w, src_u16(A.pan, A.pan_pitch), sAB, src_u16(B.pan, B.pan_pitch), sAB + 4, rows, cols,
                                      d_res + 12 * k, d_res + 12 * (k + 1), &vt))) return rc;
    }
    if (k < n) {
        const IbUnit &A = units[k];
        const IbUnit *one[1] = {&A};
        if (up.vonly) {
            if ((rc = correlate_units_vup(ctx, pl, up, w, tab, src_u16(A.pan, A.pan_pitch), src_none(), one, 1, rows, cols, band_rows, band_cols,
                                          d_res + 12 * k, nullptr))) return rc;
        } else {
        if (!up.vtab && (rc = upsample(one, 1, w.fb, sAB))) return rc;
        if (up.xtab) {
            if ((rc = correlate_units_up(ctx, pl, up, w, src_u16(A.pan, A.pan_pitch), src_none(), sAB, one, 1, rows, cols, band_rows, band_cols,
                                         d_res + 12 * k, nullptr))) return rc;
        } else {
            const HTaps vt{band_cols, tab->d_xofs, tab->d_alpha, tab->x4h};
            if ((rc = correlate_one_to_four(ctx, pl, w, src_u16(A.pan, A.pan_pitch), sAB, rows, cols, d_res + 12 * k, &vt))) return rc;
        }
        }
    }
    if (n > 0 && (rc = fetch_results(ctx, 12 * n, host_out))) return rc;
    return OIP_OK;
}

extern "C" int oip_interband_correlate_units(oip_ctx *ctx, const uint16_t *const *d_pan, const size_t *pan_pitch,
                                             const uint16_t *const *d_bands, const size_t *band_pitch, int n, int rows,
                                             int cols, double *out)
{
    OIP_CHECK_CTX(ctx);
    if (n < 0 || (n > 0 && (!d_pan || !pan_pitch || !d_bands || !band_pitch || !out)) || rows <= 0 || cols <= 0)
        return oip_fail(ctx, OIP_E_INVALID, "oip_interband_correlate_units: bad argument");
    // preproc.h:274-276: the band window is the PAN window integer-divided by 4
    const int band_rows = rows / OIP_MSS_BANDS, band_cols = cols / OIP_MSS_BANDS;
    if (band_rows <= 0 || band_cols <= 0) return oip_fail(ctx, OIP_E_INVALID, "oip_interband_correlate_units: window too small");
    std::vector<IbUnit> units((size_t)n);
    for (int i = 0; i < n; ++i) {
        units[i].pan = d_pan[i];
        units[i].pan_pitch = (long)pan_pitch[i];
        units[i].band_pitch = (long)band_pitch[i];
        bool ok = d_pan[i] && pan_pitch[i] >= (size_t)cols && band_pitch[i] >= (size_t)band_cols;
        for (int b = 0; b < OIP_MSS_BANDS; ++b) { units[i].band[b] = d_bands[4 * i + b]; ok = ok && d_bands[4 * i + b]; }
        if (!ok) return oip_fail(ctx, OIP_E_INVALID, "oip_interband_correlate_units: unit %d has a null pointer or a short pitch", i);
    }
    return interband_units(ctx, units.data(), n, rows, cols, band_rows, band_cols, out);
}

extern "C" int oip_interband_correlate(oip_ctx *ctx, const uint16_t *d_pan, long Lp, long prow0, long pn,
                                       const uint16_t *d_planes, size_t plane_stride, long mrow0, long mn, int W,
                                       int slices, int sections, int corr_lines, double *out)
{
    OIP_CHECK_CTX(ctx);
    if (!d_pan || !d_planes || !out || W <= 0 || Lp <= 0 || corr_lines <= 0)
        return oip_fail(ctx, OIP_E_INVALID, "oip_interband_correlate: bad argument");
    // preproc.h:228-237
    if (slices < OIP_IBCV_MIN_SLICES)
        return oip_fail(ctx, OIP_E_INVALID, "CalcInterBandCorrelation: at lease %d slice needed", OIP_IBCV_MIN_SLICES);
    if (sections <= 0) return oip_fail(ctx, OIP_E_INVALID, "CalcInterBandCorrelation: section count should be a positive integer");
    if (sections > 1 && (long)sections * corr_lines > Lp)
        return oip_fail(ctx, OIP_E_INVALID, "CalcInterBandCorrelation: too many sections (%d lines per section), not enough total PAN data lines", corr_lines);
    // preproc.h:245-247, :274-276
    const int baseRows = (int)(Lp < corr_lines ? Lp : corr_lines);
    const long baseRowGap = (Lp - (long)baseRows * sections) / (sections + 1);
    const int baseSliceCols = W / slices;
    const int bandRows = baseRows / OIP_MSS_BANDS;
    const long bandRowGap = baseRowGap / OIP_MSS_BANDS;
    const int bandSliceCols = baseSliceCols / OIP_MSS_BANDS;
    const int Wb = W / OIP_MSS_BANDS;
    if (bandRows <= 0 || bandSliceCols <= 0) return oip_fail(ctx, OIP_E_INVALID, "oip_interband_correlate: slice too small");
    const int n = slices * sections;
    std::vector<int> which;
    std::vector<IbUnit> units;
    for (int sec = 0; sec < sections; ++sec) {
        const long secRowStart = baseRowGap + (long)sec * (baseRows + baseRowGap);        // preproc.h:257
        const long secBandRowStart = bandRowGap + (long)sec * (bandRows + bandRowGap);    // preproc.h:284
        if (secRowStart < prow0 || secRowStart + baseRows > prow0 + pn) continue;
        if (secBandRowStart < mrow0 || secBandRowStart + bandRows > mrow0 + mn) continue;
        for (int i = 0; i < slices; ++i) {
            which.push_back(sec * slices + i);
            IbUnit u;
            u.pan = d_pan + (size_t)(secRowStart - prow0) * W + (size_t)i * baseSliceCols;
            u.pan_pitch = W;
            for (int b = 0; b < OIP_MSS_BANDS; ++b)
                u.band[b] = d_planes + (size_t)b * plane_stride + (size_t)(secBandRowStart - mrow0) * Wb + (size_t)i * bandSliceCols;
            u.band_pitch = Wb;
            units.push_back(u);
        }
    }
    std::vector<double> r(12 * units.size() + 12);
    int rc = interband_units(ctx, units.data(), (int)units.size(), baseRows, baseSliceCols, bandRows, bandSliceCols, r.data());
    if (rc) return rc;
    for (int b = 0; b < OIP_MSS_BANDS; ++b)
        for (int u = 0; u < n; ++u) {
            double *o = out + ((size_t)b * n + u) * 4;
            const int i = u % slices;
            o[0] = o[1] = o[2] = NAN;
            o[3] = (double)(i * baseSliceCols + baseSliceCols / 2);                        // preproc.h:326
        }
    for (size_t j = 0; j < which.size(); ++j)
        for (int b = 0; b < OIP_MSS_BANDS; ++b)
            for (int k = 0; k < 3; ++k) out[((size_t)b * n + which[j]) * 4 + k] = r[12 * j + 3 * b + k];
    return OIP_OK;
}
