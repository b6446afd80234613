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

// OpenCV imgwarp.cpp/resize.cpp interpolateCubic in f32, operation by operation
__device__ __forceinline__ void interp_cubic(float x, float *c)
{
    const float A = -0.75f;
    float x1 = __fadd_rn(x, 1.f);
    c[0] = __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(A, x1), 5 * A), x1), 8 * A), x1), 4 * A);
    c[1] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2, x), A + 3), x), x), 1.f);
    float y = __fsub_rn(1.f, x);
    c[2] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(A + 2, y), A + 3), y), y), 1.f);
    c[3] = __fsub_rn(__fsub_rn(__fsub_rn(1.f, c[0]), c[1]), c[2]);
}

// cv::resize(f32, INTER_CUBIC): fx = (float)((dx+0.5)*scale - 0.5), sx = floor(fx), fx -= sx;
// horizontal 4 taps (edge replicated) into an f32 row value, then vertical 4 taps.
__global__ __launch_bounds__(kBlock) void resize_cubic_f32_kernel(const float *__restrict__ src, int sw, int sh,
                                                                  float *__restrict__ dst, int dw, int dh,
                                                                  double scale_x, double scale_y)
{
    const int dx = blockIdx.x * kBlock + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= dw) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx = __fsub_rn(fx, (float)sx);
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy = __fsub_rn(fy, (float)sy);
    float a[4], b[4];
    interp_cubic(fx, a);
    interp_cubic(fy, b);
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
        const float *S = src + (size_t)yk * sw;
        float v = __fmul_rn(S[cx[0]], a[0]);
        v = __fadd_rn(v, __fmul_rn(S[cx[1]], a[1]));
        v = __fadd_rn(v, __fmul_rn(S[cx[2]], a[2]));
        v = __fadd_rn(v, __fmul_rn(S[cx[3]], a[3]));
        r[k] = v;
    }
    float o = __fmul_rn(r[0], b[0]);
    o = __fadd_rn(o, __fmul_rn(r[1], b[1]));
    o = __fadd_rn(o, __fmul_rn(r[2], b[2]));
    o = __fadd_rn(o, __fmul_rn(r[3], b[3]));
    dst[(size_t)dy * dw + dx] = o;
}

// ---- pack two real images into one zero-padded complex image ----------------------------------
__global__ __launch_bounds__(kBlock) void pack_kernel(float2 *__restrict__ z, int M, int N, const float *__restrict__ re,
                                                      const float *__restrict__ im, int rows, int cols)
{
    const int x = blockIdx.x * kBlock + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= N) return;
    float2 v = make_float2(0.f, 0.f);
    if (y < rows && x < cols) {
        v.x = re[(size_t)y * cols + x];
        if (im) v.y = im[(size_t)y * cols + x];
    }
    z[(size_t)y * N + x] = v;
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

__device__ __forceinline__ float2 spec_of(const SpecRef &s, long pk, long pmk)
{
    float2 zk = s.z[pk], zm = s.z[pmk];
    if (s.part == 0) return make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));
    return make_float2(0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x));
}

// C for one stored bin, following mulSpectrums/magSpectrums/divSpectrums
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

__global__ __launch_bounds__(kBlock) void cross_power_kernel(float2 *__restrict__ out, XpowerJob job, int M, int N,
                                                             OipAxisDigits yd, OipAxisDigits xd)
{
    const int px = blockIdx.x * kBlock + threadIdx.x;
    const int py = blockIdx.y;
    if (px >= N) return;
    const int kx = oip_pos_to_freq(xd, px);
    const int ky = oip_pos_to_freq(yd, py);
    const int nkx = kx ? N - kx : 0, nky = ky ? M - ky : 0;
    // CCS stores kx in [0, N/2]; in the first / Nyquist column only ky <= M/2
    const bool edge_col = (kx == 0) || (2 * kx == N);
    bool stored;
    if (edge_col) stored = 2 * ky <= M;
    else stored = 2 * kx < N;
    const int cky = stored ? ky : nky, ckx = stored ? kx : nkx;
    const int mky = stored ? nky : ky, mkx = stored ? nkx : kx;
    const long pk = (long)oip_freq_to_pos(yd, cky) * N + oip_freq_to_pos(xd, ckx);
    const long pmk = (long)oip_freq_to_pos(yd, mky) * N + oip_freq_to_pos(xd, mkx);
    const bool real_bin = edge_col && (cky == 0 || 2 * cky == M);
    float2 y = make_float2(0.f, 0.f);
    for (int c = 0; c < job.ncorr; ++c) {
        float2 A = spec_of(job.a[c], pk, pmk);
        float2 B = spec_of(job.b[c], pk, pmk);
        float2 C = cross_power_bin(A, B, real_bin, edge_col);
        if (!stored) C.y = -C.y;
        if (c == 0) { y.x += C.x; y.y += C.y; }       // Y = C1 + i C2
        else { y.x -= C.y; y.y += C.x; }
    }
    out[(long)py * N + px] = y;
}

// ---- peak: first maximum of the fftShift-ed surface + 5x5 weighted centroid ----------------------
struct PeakPartial {
    float val;
    int pad;
    long key;           // index in the shifted image, row-major
};

__device__ __forceinline__ bool peak_better(float v, long k, float bv, long bk)
{
    return v > bv || (v == bv && k < bk);
}

__global__ __launch_bounds__(kBlock) void peak_partial_kernel(const float2 *__restrict__ c, int part, int M, int N,
                                                              PeakPartial *__restrict__ partials)
{
    __shared__ float sval[kBlock];
    __shared__ long skey[kBlock];
    const long n = (long)M * N;
    float bv = -INFINITY;
    long bk = n;            // any real element beats this
    const int ym = M >> 1, xm = N >> 1;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        int y = (int)(i / N), x = (int)(i - (long)y * N);
        float v = part ? c[i].y : c[i].x;
        int ys = y + ym; if (ys >= M) ys -= M;
        int xs = x + xm; if (xs >= N) xs -= N;
        long key = (long)ys * N + xs;
        // minMaxLoc skips nothing; NaN never compares greater, like the reference's scan
        if (peak_better(v, key, bv, bk)) { bv = v; bk = key; }
    }
    sval[threadIdx.x] = bv;
    skey[threadIdx.x] = bk;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            if (peak_better(sval[threadIdx.x + s], skey[threadIdx.x + s], sval[threadIdx.x], skey[threadIdx.x])) {
                sval[threadIdx.x] = sval[threadIdx.x + s];
                skey[threadIdx.x] = skey[threadIdx.x + s];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[blockIdx.x].val = sval[0];
        partials[blockIdx.x].key = skey[0];
    }
}

__global__ __launch_bounds__(kBlock) void peak_final_kernel(const float2 *__restrict__ c, int part, int M, int N,
                                                            const PeakPartial *__restrict__ partials, int npart,
                                                            double *__restrict__ result)
{
    __shared__ float sval[kBlock];
    __shared__ long skey[kBlock];
    float bv = -INFINITY;
    long bk = (long)M * N;
    for (int i = threadIdx.x; i < npart; i += kBlock)
        if (peak_better(partials[i].val, partials[i].key, bv, bk)) { bv = partials[i].val; bk = partials[i].key; }
    sval[threadIdx.x] = bv;
    skey[threadIdx.x] = bk;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            if (peak_better(sval[threadIdx.x + s], skey[threadIdx.x + s], sval[threadIdx.x], skey[threadIdx.x])) {
                sval[threadIdx.x] = sval[threadIdx.x + s];
                skey[threadIdx.x] = skey[threadIdx.x + s];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    long key = skey[0];
    if (key >= (long)M * N) key = 0;                 // all-NaN surface: minMaxLoc leaves (0,0)
    const int py = (int)(key / N), px = (int)(key - (long)py * N);
    // weightedCentroid(C, peak, Size(5,5), &response)   (phasecorr.cpp)
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > M - 1) maxr = M - 1;
    if (maxc > N - 1) maxc = N - 1;
    const int ym = M >> 1, xm = N >> 1;
    double cxs = 0.0, cys = 0.0, si = 0.0;
    for (int y = minr; y <= maxr; ++y) {
        int yo = y - ym; if (yo < 0) yo += M;        // shifted(y) = original((y - yMid) mod M)
        for (int x = minc; x <= maxc; ++x) {
            int xo = x - xm; if (xo < 0) xo += N;
            float2 e = c[(long)yo * N + xo];
            double v = (double)(part ? e.y : e.x);
            cxs = __dadd_rn(cxs, __dmul_rn((double)x, v));
            cys = __dadd_rn(cys, __dmul_rn((double)y, v));
            si = __dadd_rn(si, v);
        }
    }
    double response = si;
    si = __dadd_rn(si, DBL_EPSILON);
    double cx = cxs / si, cy = cys / si;
    response = response / (double)((long)M * N);
    result[0] = (double)N / 2.0 - cx;
    result[1] = (double)M / 2.0 - cy;
    result[2] = response;
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

// Workspace carve-up for one correlation unit
struct PcWork {
    float2 *z[3];
    float2 *y[2];
    float *fa;          // base window, f32
    float *fb[4];       // second images, f32 (up-sampled bands)
    float *fsmall;      // MSS window before resize
    PeakPartial *partials;
    int npart;
};

int carve(oip_ctx *ctx, int M, int N, int rows, int cols, int small_elems, int nz, int ny, int nfb, PcWork *w)
{
    const size_t zbytes = align_up(sizeof(float2) * (size_t)M * N, 256);
    const size_t fbytes = align_up(sizeof(float) * (size_t)rows * cols, 256);
    const size_t sbytes = align_up(sizeof(float) * (size_t)(small_elems > 0 ? small_elems : 1), 256);
    w->npart = ctx->cu_count * 8;
    const size_t pbytes = align_up(sizeof(PeakPartial) * w->npart, 256);
    size_t total = zbytes * (nz + ny) + fbytes * (1 + nfb) + sbytes + pbytes;
    void *ws;
    int rc = oip_workspace(ctx, total, &ws);
    if (rc) return rc;
    char *p = (char *)ws;
    for (int i = 0; i < 3; ++i) { w->z[i] = i < nz ? (float2 *)p : nullptr; if (i < nz) p += zbytes; }
    for (int i = 0; i < 2; ++i) { w->y[i] = i < ny ? (float2 *)p : nullptr; if (i < ny) p += zbytes; }
    w->fa = (float *)p; p += fbytes;
    for (int i = 0; i < 4; ++i) { w->fb[i] = i < nfb ? (float *)p : nullptr; if (i < nfb) p += fbytes; }
    w->fsmall = (float *)p; p += sbytes;
    w->partials = (PeakPartial *)p;
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

int launch_resize(oip_ctx *ctx, const float *src, int sw, int sh, float *dst, int dw, int dh)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    OipProfScope prof(ctx, "resize_cubic_f32_kernel");
    hipLaunchKernelGGL(resize_cubic_f32_kernel, dim3((dw + kBlock - 1) / kBlock, dh), dim3(kBlock), 0, ctx->stream, src,
                       sw, sh, dst, dw, dh, scale_x, scale_y);
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

int launch_pack(oip_ctx *ctx, float2 *z, int M, int N, const float *re, const float *im, int rows, int cols)
{
    OipProfScope prof(ctx, "pack_kernel");
    hipLaunchKernelGGL(pack_kernel, dim3((N + kBlock - 1) / kBlock, M), dim3(kBlock), 0, ctx->stream, z, M, N, re, im,
                       rows, cols);
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

int launch_xpower(oip_ctx *ctx, float2 *out, const XpowerJob &job, const OipFft2dPlan *pl)
{
    OipProfScope prof(ctx, "cross_power_kernel");
    hipLaunchKernelGGL(cross_power_kernel, dim3((pl->N + kBlock - 1) / kBlock, pl->M), dim3(kBlock), 0, ctx->stream, out,
                       job, pl->M, pl->N, digits_of(pl->yf, pl->M), digits_of(pl->xf, pl->N));
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

int launch_peak(oip_ctx *ctx, const float2 *c, int part, int M, int N, const PcWork &w, double *d_result)
{
    long n = (long)M * N;
    int blocks = (int)((n + kBlock - 1) / kBlock < w.npart ? (n + kBlock - 1) / kBlock : w.npart);
    {
        OipProfScope prof(ctx, "peak_partial_kernel");
        hipLaunchKernelGGL(peak_partial_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, c, part, M, N, w.partials);
    }
    {
        OipProfScope prof(ctx, "peak_final_kernel");
        hipLaunchKernelGGL(peak_final_kernel, dim3(1), dim3(kBlock), 0, ctx->stream, c, part, M, N, w.partials, blocks,
                           d_result);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// one pair (a, b) -> result slot
int correlate_pair(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, const float *a, const float *b, int rows,
                   int cols, double *d_result)
{
    int rc = launch_pack(ctx, w.z[0], pl->M, pl->N, a, b, rows, cols);
    if (rc) return rc;
    if ((rc = oip_fft2d_exec(ctx, pl, w.z[0], 0))) return rc;
    XpowerJob job;
    memset(&job, 0, sizeof job);
    job.ncorr = 1;
    job.a[0] = {w.z[0], 0};
    job.b[0] = {w.z[0], 1};
    if ((rc = launch_xpower(ctx, w.y[0], job, pl))) return rc;
    if ((rc = oip_fft2d_exec(ctx, pl, w.y[0], 1))) return rc;
    return launch_peak(ctx, w.y[0], 0, pl->M, pl->N, w, d_result);
}

// base image a against four images b0..b3: 3 forward + 2 inverse complex transforms
int correlate_one_to_four(oip_ctx *ctx, const OipFft2dPlan *pl, const PcWork &w, const float *a, float *const b[4],
                          int rows, int cols, double *d_results /* 4 x 3 */)
{
    int rc;
    if ((rc = launch_pack(ctx, w.z[0], pl->M, pl->N, a, b[0], rows, cols))) return rc;
    if ((rc = launch_pack(ctx, w.z[1], pl->M, pl->N, b[1], b[2], rows, cols))) return rc;
    if ((rc = launch_pack(ctx, w.z[2], pl->M, pl->N, b[3], nullptr, rows, cols))) return rc;
    for (int i = 0; i < 3; ++i)
        if ((rc = oip_fft2d_exec(ctx, pl, w.z[i], 0))) return rc;
    XpowerJob j0, j1;
    memset(&j0, 0, sizeof j0);
    memset(&j1, 0, sizeof j1);
    j0.ncorr = 2;
    j0.a[0] = {w.z[0], 0}; j0.b[0] = {w.z[0], 1};
    j0.a[1] = {w.z[0], 0}; j0.b[1] = {w.z[1], 0};
    j1.ncorr = 2;
    j1.a[0] = {w.z[0], 0}; j1.b[0] = {w.z[1], 1};
    j1.a[1] = {w.z[0], 0}; j1.b[1] = {w.z[2], 0};
    if ((rc = launch_xpower(ctx, w.y[0], j0, pl))) return rc;
    if ((rc = launch_xpower(ctx, w.y[1], j1, pl))) return rc;
    for (int i = 0; i < 2; ++i)
        if ((rc = oip_fft2d_exec(ctx, pl, w.y[i], 1))) return rc;
    for (int c = 0; c < 4; ++c)
        if ((rc = launch_peak(ctx, w.y[c >> 1], c & 1, pl->M, pl->N, w, d_results + 3 * c))) return rc;
    return OIP_OK;
}

int fetch_results(oip_ctx *ctx, int count, double *host_out)
{
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
    return launch_resize(ctx, d_src, sw, sh, d_dst, dw, dh);
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
    if ((rc = carve(ctx, M, N, 1, 1, 0, 1, 1, 0, &w))) return rc;
    double *d_res = (double *)ctx->d_small;
    if ((rc = correlate_pair(ctx, pl, w, d_a, d_b, rows, cols, d_res))) return rc;
    double r[3];
    if ((rc = fetch_results(ctx, 3, r))) return rc;
    if (dx) *dx = r[0];
    if (dy) *dy = r[1];
    if (response) *response = r[2];
    return OIP_OK;
}

extern "C" int oip_stt_correlate(oip_ctx *ctx, const uint16_t *d_pan1, const uint16_t *d_pan2, int W, long L, long row0,
                                 long nrows, int sections, int lines_per_section, int overlap_cols, int edge_cols,
                                 double *out)
{
    OIP_CHECK_CTX(ctx);
    if (!d_pan1 || !d_pan2 || !out || W <= 0 || L <= 0 || sections <= 0 || lines_per_section <= 0 || overlap_cols <= 0 ||
        overlap_cols > W || edge_cols < 0 || edge_cols >= overlap_cols || sections > 1000)
        return oip_fail(ctx, OIP_E_INVALID, "oip_stt_correlate: bad argument");
    // stitcher.h:75-77
    if (L < (long)sections * lines_per_section)
        return oip_fail(ctx, OIP_E_INVALID, "PAN line count less than sections times line-per-section, use smaller -s and/or -l value(s)");
    const int rows = lines_per_section, cols = overlap_cols - edge_cols;
    const int M = optimal_dft_size(rows), N = optimal_dft_size(cols);
    if (M > 65535) return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_stt_correlate: section taller than 65535 lines");
    const OipFft2dPlan *pl;
    int rc = oip_fft2d_plan(ctx, M, N, &pl);
    if (rc) return rc;
    PcWork w;
    if ((rc = carve(ctx, M, N, rows, cols, 0, 1, 1, 1, &w))) return rc;
    // stitcher.h:151-152, :167
    const long gap = (L - (long)sections * lines_per_section) / (sections + 1);
    const long step = gap + lines_per_section;
    double *d_res = (double *)ctx->d_small;
    std::vector<int> have(sections, 0);
    for (int s = 0; s < sections; ++s) {
        const long off = gap + (long)s * step;
        if (off < row0 || off + rows > row0 + nrows) continue;     // another rank's section
        have[s] = 1;
        // stitcher.h:175-176: PAN1 cols [W-ov, W-edge), PAN2 cols [edge, ov)
        if ((rc = launch_window(ctx, d_pan1, W, off - row0, W - overlap_cols, rows, cols, w.fa))) return rc;
        if ((rc = launch_window(ctx, d_pan2, W, off - row0, edge_cols, rows, cols, w.fb[0]))) return rc;
        if ((rc = correlate_pair(ctx, pl, w, w.fa, w.fb[0], rows, cols, d_res + 3 * s))) return rc;
    }
    std::vector<double> r(3 * sections);
    if ((rc = fetch_results(ctx, 3 * sections, r.data()))) return rc;
    for (int s = 0; s < sections; ++s)
        for (int k = 0; k < 3; ++k) out[3 * s + k] = have[s] ? r[3 * s + k] : NAN;
    return OIP_OK;
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
    if ((long)slices * sections * 12 * sizeof(double) > 65536)
        return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_interband_correlate: too many slices x sections");
    // preproc.h:245-247, :274-276
    const int baseRows = (int)(Lp < corr_lines ? Lp : corr_lines);
    const long baseRowGap = (Lp - (long)baseRows * sections) / (sections + 1);
    const int baseSliceCols = W / slices;
    const int bandRows = baseRows / OIP_MSS_BANDS;
    const long bandRowGap = baseRowGap / OIP_MSS_BANDS;
    const int bandSliceCols = baseSliceCols / OIP_MSS_BANDS;
    const int Wb = W / OIP_MSS_BANDS;
    if (bandRows <= 0 || bandSliceCols <= 0) return oip_fail(ctx, OIP_E_INVALID, "oip_interband_correlate: slice too small");
    const int M = optimal_dft_size(baseRows), N = optimal_dft_size(baseSliceCols);
    if (M > 65535) return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_interband_correlate: more than 65535 correlation lines");
    const OipFft2dPlan *pl;
    int rc = oip_fft2d_plan(ctx, M, N, &pl);
    if (rc) return rc;
    PcWork w;
    if ((rc = carve(ctx, M, N, baseRows, baseSliceCols, bandRows * bandSliceCols, 3, 2, 4, &w))) return rc;
    double *d_res = (double *)ctx->d_small;
    const int n = slices * sections;
    std::vector<int> have(n, 0);
    for (int sec = 0; sec < sections; ++sec) {
        const long secRowStart = baseRowGap + (long)sec * (baseRows + baseRowGap);        // preproc.h:257
        const long secBandRowStart = bandRowGap + (long)sec * (bandRows + bandRowGap);    // preproc.h:284
        if (secRowStart < prow0 || secRowStart + baseRows > prow0 + pn) continue;
        if (secBandRowStart < mrow0 || secBandRowStart + bandRows > mrow0 + mn) continue;
        for (int i = 0; i < slices; ++i) {
            const int u = sec * slices + i;
            have[u] = 1;
            if ((rc = launch_window(ctx, d_pan, W, secRowStart - prow0, i * baseSliceCols, baseRows, baseSliceCols, w.fa))) return rc;
            for (int b = 0; b < OIP_MSS_BANDS; ++b) {
                if ((rc = launch_window(ctx, d_planes + (size_t)b * plane_stride, Wb, secBandRowStart - mrow0,
                                        i * bandSliceCols, bandRows, bandSliceCols, w.fsmall))) return rc;
                if ((rc = launch_resize(ctx, w.fsmall, bandSliceCols, bandRows, w.fb[b], baseSliceCols, baseRows))) return rc;
            }
            if ((rc = correlate_one_to_four(ctx, pl, w, w.fa, w.fb, baseRows, baseSliceCols, d_res + 12 * u))) return rc;
        }
    }
    std::vector<double> r(12 * n);
    if ((rc = fetch_results(ctx, 12 * n, r.data()))) return rc;
    for (int b = 0; b < OIP_MSS_BANDS; ++b)
        for (int u = 0; u < n; ++u) {
            double *o = out + ((size_t)b * n + u) * 4;
            const int i = u % slices;
            for (int k = 0; k < 3; ++k) o[k] = have[u] ? r[12 * u + 3 * b + k] : NAN;
            o[3] = (double)(i * baseSliceCols + baseSliceCols / 2);                        // preproc.h:326
        }
    return OIP_OK;
}
