// fft.hip -- batched complex 2-D FFT for the phase-correlation path on gfx950.
//
// cv::phaseCorrelate (call sites stitcher.h:180, preproc.h:316) spends its time in three
// real 2-D DFTs per call (OpenCV imgproc/phasecorr.cpp -> core dft()).  Sizes are
// getOptimalDFTSize values, i.e. 2^a 3^b 5^c: 16000 x {200, 1250, 3000} for the reference
// geometry and the BASELINE configs.
//
// Design (not OpenCV's CCS-packed real transform):
//   * two real images are packed as one complex image z = a + i b and transformed together;
//     the per-image spectra are separated algebraically inside the cross-power kernel
//     (phasecorr.hip).  5 real images -> 3 complex FFTs, 4 real correlation surfaces -> 2.
//   * every axis length L is split into factors F1*F2(*F3).  One "pass" kernel does all
//     F-point sub-transforms of one factor with a Stockham network in LDS, for a tile of V
//     independent transforms that are ADJACENT IN MEMORY, so global loads and stores are
//     V*8-byte (256 B) contiguous segments whatever the axis:
//        mode A  (strided points, contiguous lanes): column passes
//        mode B  (contiguous points):                the row pass (whole F-point rows)
//   * forward = columns then rows, decimation in frequency; inverse = rows then columns,
//     decimation in time.  The forward transform leaves each axis in digit-scrambled order
//     (position p = k1*F2 + k2 holds frequency k1 + F1*k2), the point-wise kernel works on
//     that order and the inverse consumes it and returns natural order: no transposes, no
//     reordering passes.  A 16000 x 3000 transform is 3 passes over HBM (column x2, row),
//     each 8 B read + 8 B written per point.
//   * fusions: the first forward pass reads the two real f32 images directly (zero padding
//     included) instead of a packed complex array; the last inverse pass does not store the
//     correlation surface at all -- it reduces it to per-tile maxima (the arg-max of the
//     fftShift-ed surface), and the 5x5 window around the winner is recomputed from 25 tiles.
//   * the shapes of the reference geometry run compile-time specialised kernels (radices,
//     tile and strides are constants: no integer division in the butterfly loops, one LDS
//     buffer with register staging, 4-5 workgroups per CU); everything else runs the generic
//     kernel with run-time radices.
//   * twiddles come from fp64-computed tables.
// The inverse is unnormalised, like cv::idft without DFT_SCALE (phasecorr.cpp).
#include "oip_fft.h"
#include "oip_fft_dev.h"
#include "oip_internal.h"

#include <cmath>
#include <map>

using namespace oipfft;

namespace {

constexpr int kFftBlock = 256;
constexpr int kMaxTileElems = 5120;       // generic kernel: F * (V+1) complex elements per LDS buffer

// ---- tile addressing shared by the generic and the specialised kernels -------------------------
struct Tile {
    long base;      // element offset of (point 0, lane 0)
    int nv;         // live vectors in the tile
    int lane0;      // mode 0: first lane
    int o1;         // mode 0: outer index 1
    int o2;         // mode 0: outer index 2
    long vec0;      // mode 1: first vector
    long gtile;     // tile index over the whole pass (peak partial slot), independent of panelling
};

__device__ __forceinline__ Tile decode_tile(const OipFftPass &p, long bid)
{
    Tile t;
    const int V = 1 << p.vshift;
    if (p.mode == 0) {
        int lt;
        long rest;
        if (p.grid3) {
            // (lane tile, o1, o2) straight from the grid: two 64-bit divisions by run-time values are
            // ~150 scalar instructions, and these passes issue about as many scalar as vector ones
            // Workgroups go to the 8 XCDs round-robin in launch order and every XCD has its own L2.
            // Neighbouring lane tiles share cache lines whenever a tile row is narrower than a line
            // (2-byte PAN pixels: 32 B per row; the 32-byte tap runs of the up-sampled bands), so XCD j
            // is given the contiguous lane tiles [j * chunk, (j + 1) * chunk) instead of every eighth
            // one: the sharing becomes L2 hits instead of 2-4x over-fetch from the fabric.  The grid's
            // x extent is padded to 8 * chunk; surplus workgroups leave before any barrier.
            const int rel = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
            lt = p.lt0 + rel;
            t.o1 = blockIdx.y;
            t.o2 = blockIdx.z;
            rest = (long)(t.o2 * p.O1 + t.o1);
        } else {
            const int ltn = p.ltn > 0 ? p.ltn : p.lane_tiles;      // lane-tile window of this launch (column panel)
            lt = p.lt0 + (int)(bid % ltn);
            rest = bid / ltn;
            t.o1 = (int)(rest % p.O1);
            t.o2 = (int)(rest / p.O1);
        }
        t.gtile = rest * p.lane_tiles + lt;
        t.lane0 = lt << p.vshift;
        t.nv = p.lanes - t.lane0 < V ? (int)(p.lanes - t.lane0) : V;
        if (p.grid3 && lt - p.lt0 >= (p.ltn > 0 ? p.ltn : p.lane_tiles)) t.nv = 0;      // padding workgroup
        t.base = (long)t.o2 * p.o2_stride + (long)t.o1 * p.o1_stride + t.lane0;
        t.vec0 = 0;
    } else {
        t.vec0 = bid << p.vshift;
        t.gtile = bid;
        t.nv = p.lanes - t.vec0 < V ? (int)(p.lanes - t.vec0) : V;
        t.base = 0;                      // mode 1 addresses come from vec_offset()
        t.lane0 = t.o1 = t.o2 = 0;
    }
    return t;
}

// mode 1: element offset of point n of vector (vec0 + v): vectors are the F-point pieces of
// the rows, rows are P elements apart
__device__ __forceinline__ long vec_offset(const OipFftPass &p, const Tile &t, int n, int v)
{
    const int vec = (int)t.vec0 + v;                 // M * (N / F) < 2^31
    if (p.N == p.F) return (long)vec * p.P + n;      // whole rows: no division
    const int per_row = p.N / p.F;
    const int y = vec / per_row;
    return (long)y * p.P + (vec - y * per_row) * p.F + n;
}

// (row, column) of tile element (point n, vector v) in the M x N array
__device__ __forceinline__ void elem_coord(const OipFftPass &p, const Tile &t, int n, int v, int *y, int *x)
{
    if (p.mode == 0) {
        if (p.axis == 1) { *y = t.o2 * p.T + t.o1 + n * p.S; *x = t.lane0 + v; }
        else { *y = t.o2; *x = t.o1 * p.T + t.lane0 + v + n * p.S; }
    } else {
        const int per_row = p.N / p.F;
        const long vec = t.vec0 + v;
        *y = (int)(vec / per_row);
        *x = (int)(vec - (long)(*y) * per_row) * p.F + n;
    }
}

// horizontal four cubic taps on an image whose vertical up-sampling is already done: the four
// values sit in one 16- or 28-byte run of row y, so a 16-lane tile row reads a single cache line
__device__ __forceinline__ float hresize_tap(const float *__restrict__ V, const OipFftIo &io, int y, int x)
{
    const int sx = io.xofs[x];
    const float4 a = reinterpret_cast<const float4 *>(io.alpha)[x];
    const int last = io.v_cols - 1;
    int c0 = sx - 1, c1 = sx, c2 = sx + 1, c3 = sx + 2;
    c0 = c0 < 0 ? 0 : (c0 > last ? last : c0);
    c1 = c1 < 0 ? 0 : (c1 > last ? last : c1);
    c2 = c2 < 0 ? 0 : (c2 > last ? last : c2);
    c3 = c3 < 0 ? 0 : (c3 > last ? last : c3);
    const float *row = V + (size_t)y * io.v_cols;
    float v = __fmul_rn(row[c0], a.x);
    v = __fadd_rn(v, __fmul_rn(row[c1], a.y));
    v = __fadd_rn(v, __fmul_rn(row[c2], a.z));
    v = __fadd_rn(v, __fmul_rn(row[c3], a.w));
    return v;
}

__device__ __forceinline__ float2 load_elem(const float2 *__restrict__ data, const OipFftPass &p, const OipFftIo &io,
                                            const Tile &t, int n, int v, long off)
{
    if (io.load_kind == 0) return data[off];
    int y, x;
    elem_coord(p, t, n, v, &y, &x);
    float2 z = make_float2(0.f, 0.f);
    if (y < io.rows && x < io.cols) {
        const size_t i = (size_t)y * io.cols + x;
        if (io.re) z.x = io.re[i];
        else if (io.re16) z.x = (float)io.re16[(size_t)y * io.pitch_re16 + x];
        else if (io.re_v) z.x = hresize_tap(io.re_v, io, y, x);
        if (io.im) z.y = io.im[i];
        else if (io.im16) z.y = (float)io.im16[(size_t)y * io.pitch_im16 + x];
        else if (io.im_v) z.y = hresize_tap(io.im_v, io, y, x);
    }
    return z;
}

__device__ __forceinline__ bool peak_better(float v, long k, float bv, long bk) { return v > bv || (v == bv && k < bk); }

// Reduce the tile to its maximum (first occurrence in the fftShift-ed scan order) for the
// real and the imaginary part; one partial per workgroup and part.
__device__ void store_peak(const float2 *buf, int Vp, const OipFftPass &p, const OipFftIo &io,
                           const Tile &t, float *sval, long *skey, long tile, long ntiles, const int kFftBlock = 256)
{
    const int V = 1 << p.vshift;
    const int total = p.F << p.vshift;
    const int ym = p.M >> 1, xm = p.N >> 1;
    float bv[2] = {-INFINITY, -INFINITY};
    const long none = (long)p.M * p.N;
    // the scan keeps the tile-local element index of the best value; the 64-bit key of the shifted
    // position is only formed for the winner (and in the rare exact ties, where the smaller key wins)
    auto key_of = [&](int e) -> long {
        if (e < 0) return none;
        const int v = p.mode == 0 ? (e & (V - 1)) : e / p.F;
        const int n = p.mode == 0 ? (e >> p.vshift) : e - v * p.F;
        int y, x;
        elem_coord(p, t, n, v, &y, &x);
        int ys = y + ym; if (ys >= p.M) ys -= p.M;
        int xs = x + xm; if (xs >= p.N) xs -= p.N;
        return (long)ys * p.N + xs;
    };
    int be[2] = {-1, -1};
    for (int e = threadIdx.x; e < total; e += kFftBlock) {
        const int v = p.mode == 0 ? (e & (V - 1)) : e / p.F;
        const int n = p.mode == 0 ? (e >> p.vshift) : e - v * p.F;
        if (v >= t.nv) continue;
        float2 z = buf[n * Vp + v];
        z.y = -z.y;                              // inverse = conj(forward(conj))
        if (z.x > bv[0]) { bv[0] = z.x; be[0] = e; }
        else if (z.x == bv[0] && key_of(e) < key_of(be[0])) be[0] = e;
        if (z.y > bv[1]) { bv[1] = z.y; be[1] = e; }
        else if (z.y == bv[1] && key_of(e) < key_of(be[1])) be[1] = e;
    }
    long bk[2] = {key_of(be[0]), key_of(be[1])};
    // wave-level arg-max with shuffles, then one LDS hand-off between the waves of the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = kFftBlock >> 6;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
        float v = bv[part];
        long k = bk[part];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(v, off, 64);
            const long ok = __shfl_xor(k, off, 64);
            if (peak_better(ov, ok, v, k)) { v = ov; k = ok; }
        }
        bv[part] = v;
        bk[part] = k;
    }
    __syncthreads();                          // every lane is done reading the tile: scratch may overlay it
    if (lane == 0) {
        sval[wave * 2] = bv[0]; sval[wave * 2 + 1] = bv[1];
        skey[wave * 2] = bk[0]; skey[wave * 2 + 1] = bk[1];
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const int part = threadIdx.x;
        float v = sval[part];
        long k = skey[part];
        for (int w = 1; w < nwaves; ++w)
            if (peak_better(sval[w * 2 + part], skey[w * 2 + part], v, k)) { v = sval[w * 2 + part]; k = skey[w * 2 + part]; }
        const unsigned long long packed = oip_peak_pack(v, k);
        if (packed) atomicMax(&io.slots[part * kPeakSlots + (int)(tile & (kPeakSlots - 1))], packed);
    }
}

// store_kind 2: the peak of part `part` from the slots (block-wide: every thread gets the key); the first
// window block of a part publishes it for the centroid kernel.  scratch: >= 8 * 16 bytes of LDS.
__device__ long window_peak_key(const OipFftPass &p, const OipFftIo &io, int blk, unsigned long long *scratch)
{
    const int part = blk / 25;
    unsigned long long best = 0ull;
    for (int i = threadIdx.x; i < kPeakSlots; i += blockDim.x) {
        const unsigned long long s = io.slots[part * kPeakSlots + i];
        best = s > best ? s : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o > best ? o : best;
    }
    const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) scratch[wave] = best;
    __syncthreads();
    best = scratch[0];
    for (int w = 1; w < nwaves; ++w) best = scratch[w] > best ? scratch[w] : best;
    __syncthreads();                          // scratch is the tile buffer
    // an all-NaN surface has no entry: minMaxLoc leaves (0, 0)
    const long key = oip_peak_key(best, 0);
    if (blk % 25 == 0 && threadIdx.x == 0) io.peak_key[part] = key;
    return key;
}

// store_kind 2: which tile holds window element w of the 5x5 window around the peak
__device__ __forceinline__ bool window_tile(const OipFftPass &p, const OipFftIo &io, int blk, long key, Tile *t, int *n0, int *v0)
{
    // block = 25 * part + w: the windows of all parts of one array go in a single launch
    const int w = blk % 25;
    const int py = (int)(key / p.N), px = (int)(key - (long)py * p.N);
    const int ys = py - 2 + w / 5, xs = px - 2 + w % 5;
    if (ys < 0 || ys >= p.M || xs < 0 || xs >= p.N) return false;     // weightedCentroid clamps the window
    int yo = ys - (p.M >> 1); if (yo < 0) yo += p.M;
    int xo = xs - (p.N >> 1); if (xo < 0) xo += p.N;
    // the last inverse pass is a column pass over the whole axis: T == M, O2 == 1
    const int V = 1 << p.vshift;
    t->o2 = 0;
    t->o1 = yo % p.S;
    *n0 = yo / p.S;
    t->lane0 = (xo >> p.vshift) << p.vshift;
    *v0 = xo - t->lane0;
    t->nv = p.lanes - t->lane0 < V ? (int)(p.lanes - t->lane0) : V;
    t->base = (long)t->o1 * p.o1_stride + t->lane0;
    t->vec0 = 0;
    t->gtile = 0;
    return true;
}

// ---- generic kernel: run-time radices, ping-pong LDS buffers --------------------------------------
template <int R>
__device__ __forceinline__ void stockham_stage(const float2 *__restrict__ in, float2 *__restrict__ out,
                                               const float2 *__restrict__ tw, int F, int Ns, int vshift, int Vp)
{
    const int V = 1 << vshift;
    const int nb = F / R;                  // butterflies per transform
    const int items = nb << vshift;
    const int twstep = F / (Ns * R);
    for (int item = threadIdx.x; item < items; item += kFftBlock) {
        const int v = item & (V - 1);
        const int b = item >> vshift;
        const int k = b % Ns;
        float2 x[R];
#pragma unroll
        for (int m = 0; m < R; ++m) x[m] = in[(b + m * nb) * Vp + v];
        if (Ns > 1) {
#pragma unroll
            for (int m = 1; m < R; ++m) x[m] = cmul(x[m], tw[k * m * twstep]);
        }
        butterfly<R>(x);
        const int j0 = (b - k) * R + k;    // (b / Ns) * Ns * R + k
#pragma unroll
        for (int m = 0; m < R; ++m) out[(j0 + m * Ns) * Vp + v] = x[m];
    }
}

__global__ __launch_bounds__(kFftBlock) void fft_pass_kernel(float2 *__restrict__ data, OipFftPass p, OipFftIo io,
                                                             const float2 *__restrict__ twF,
                                                             const float2 *__restrict__ twT)
{
    extern __shared__ float2 smem[];
    __shared__ float sval[kFftBlock];
    __shared__ long skey[kFftBlock];
    const int F = p.F;
    const int V = 1 << p.vshift;
    const int Vp = p.Vp;
    float2 *bufA = smem;
    float2 *bufB = smem + F * Vp;
    float2 *tw = smem + 2 * F * Vp;

    Tile t;
    int wn0 = 0, wv0 = 0;
    if (io.store_kind == 2) {
        const long key = window_peak_key(p, io, blockIdx.x, reinterpret_cast<unsigned long long *>(smem));
        if (!window_tile(p, io, blockIdx.x, key, &t, &wn0, &wv0)) {
            if (threadIdx.x == 0) io.window[(blockIdx.x / 25) * 32 + blockIdx.x % 25] = NAN;
            return;
        }
    } else {
        t = decode_tile(p, blockIdx.x);
        if (p.grid3 && t.nv <= 0) return;       // padding workgroup of the XCD-chunked grid
    }
    for (int i = threadIdx.x; i < F; i += kFftBlock) tw[i] = twF[i];

    const int total = F << p.vshift;
    for (int e = threadIdx.x; e < total; e += kFftBlock) {
        const int v = p.mode == 0 ? (e & (V - 1)) : e / F;
        const int n = p.mode == 0 ? (e >> p.vshift) : e - v * F;
        float2 z = make_float2(0.f, 0.f);
        if (v < t.nv) {
            const long off = p.mode == 0 ? t.base + (long)n * p.nstride + v : vec_offset(p, t, n, v);
            z = load_elem(data, p, io, t, n, v, off);
            if (p.inverse) {
                z.y = -z.y;
                if (p.tw_mode) z = cmul(z, twT[(long)(p.tw_mode == 1 ? t.lane0 + v : t.o1) * n]);
            }
        }
        bufA[n * Vp + v] = z;
    }
    __syncthreads();

    int Ns = 1;
    for (int s = 0; s < p.nradix; ++s) {
        const int r = p.radix[s];
        if (r == 4) stockham_stage<4>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else if (r == 5) stockham_stage<5>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else if (r == 2) stockham_stage<2>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else if (r == 8) stockham_stage<8>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else stockham_stage<3>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        __syncthreads();
        float2 *tmp = bufA; bufA = bufB; bufB = tmp;
        Ns *= r;
    }

    if (io.store_kind == 1) { store_peak(bufA, Vp, p, io, t, sval, skey, t.gtile, p.total_tiles); return; }
    if (io.store_kind == 2) {
        if (threadIdx.x == 0) {
            float2 z = bufA[wn0 * Vp + wv0];
            io.window[(blockIdx.x / 25) * 32 + blockIdx.x % 25] = (blockIdx.x / 25) ? -z.y : z.x;
        }
        return;
    }
    for (int e = threadIdx.x; e < total; e += kFftBlock) {
        const int v = p.mode == 0 ? (e & (V - 1)) : e / F;
        const int n = p.mode == 0 ? (e >> p.vshift) : e - v * F;
        if (v >= t.nv) continue;
        float2 z = bufA[n * Vp + v];
        if (p.inverse) z.y = -z.y;
        else if (p.tw_mode) z = cmul(z, twT[(long)(p.tw_mode == 1 ? t.lane0 + v : t.o1) * n]);
        const long off = p.mode == 0 ? t.base + (long)n * p.nstride + v : vec_offset(p, t, n, v);
        data[off] = z;
    }
}

// ---- specialised kernels: compile-time F, radices, tile; one LDS buffer ---------------------------
// raw (unconverted) element load for the prefetch: u16 sources stay as bit patterns so that no
// conversion -- hence no wait on the load -- happens before the values are committed to LDS
__device__ __forceinline__ float2 load_elem_raw(const float2 *__restrict__ data, const OipFftPass &p, const OipFftIo &io,
                                                const Tile &t, int n, int v, long off)
{
    if (io.load_kind == 0) return data[off];
    int y, x;
    elem_coord(p, t, n, v, &y, &x);
    float2 z = make_float2(0.f, 0.f);
    if (y < io.rows && x < io.cols) {
        const size_t i = (size_t)y * io.cols + x;
        if (io.re) z.x = io.re[i];
        else if (io.re16) z.x = __uint_as_float((unsigned)io.re16[(size_t)y * io.pitch_re16 + x]);
        else if (io.re_v) z.x = hresize_tap(io.re_v, io, y, x);
        if (io.im) z.y = io.im[i];
        else if (io.im16) z.y = __uint_as_float((unsigned)io.im16[(size_t)y * io.pitch_im16 + x]);
        else if (io.im_v) z.y = hresize_tap(io.im_v, io, y, x);
    }
    return z;
}

// IOK 0: plain pass (complex array in, complex array out); 1: fused loader (io.load_kind 1), plain
// store; 2: plain load, fused store (io.store_kind 1 or 2).  Separate instantiations because the
// kernel arguments of the fusions cost scalar registers -- past 100 a wave of occupancy goes, and
// these passes are latency-bound enough to lose 15-20 % with it.
template <int F, int VS, int MODE, int NT, int IOK, int... Rs>
__global__ __launch_bounds__(NT) void fft_pass_ct_kernel(float2 *__restrict__ data, OipFftPass p, OipFftIo io,
                                                         const float2 *__restrict__ twF,
                                                         const float2 *__restrict__ twT)
{
    constexpr int kFftBlock = NT;      // shadows the generic kernel's block size
    constexpr int V = 1 << VS;
    constexpr int Vp = VS > 1 ? V + 1 : V;      // 1 or 2 vectors: interleaved without padding
    constexpr int TWN = TwTable<F, Rs...>::value();
    constexpr int TOTAL = F << VS;
    constexpr int NLD = (TOTAL + kFftBlock - 1) / kFftBlock;
    __shared__ float2 buf[F * Vp];
    __shared__ float2 tw[TWN];
    __shared__ float2 twj[MODE == 0 ? F : 1];       // inter-pass twiddles of this tile (column passes)
    static_assert(sizeof(float2) * F * Vp >= kFftBlock * 12 + 16, "tile too small to host the reduction scratch");

    const long ntiles = (IOK == 2 && io.store_kind == 2) ? gridDim.x : p.ntiles;
    long tile = blockIdx.x;
    if (MODE != 0 && tile >= ntiles) return;
    Tile t;
    int wn0 = 0, wv0 = 0;
    if ((IOK == 2 && io.store_kind == 2)) {
        const long key = window_peak_key(p, io, blockIdx.x, reinterpret_cast<unsigned long long *>(buf));
        if (!window_tile(p, io, blockIdx.x, key, &t, &wn0, &wv0)) {
            if (threadIdx.x == 0) io.window[(blockIdx.x / 25) * 32 + blockIdx.x % 25] = NAN;
            return;
        }
    } else if (MODE == 0) {
        OipFftPass pg = p;
        pg.grid3 = 1;                  // launch_pass always gives mode-0 passes the 3-D grid
        t = decode_tile(pg, 0);
        if (t.nv <= 0) return;         // padding workgroup of the XCD-chunked grid
    } else {
        t = decode_tile(p, tile);
    }
    const bool tile_tw = MODE == 0 && p.tw_mode == 2;
    const bool raw_re16 = (IOK == 1 && io.load_kind == 1) && !io.re && io.re16;
    const bool raw_im16 = (IOK == 1 && io.load_kind == 1) && !io.im && io.im16;

    float2 zz[NLD];
    auto issue_loads = [&](const Tile &tt) {
        if (IOK == 1 && MODE == 0 && kFftBlock % V == 0 && (io.re_v || io.im_v) && p.axis == 1) {
            // Fused loader with horizontal cubic taps: a thread's elements all sit in one image
            // column x, so the four clamped source columns and weights are formed once.
            const int v = threadIdx.x & (V - 1);
            const int x = tt.lane0 + v;
            const bool xin = v < tt.nv && x < io.cols;
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xin) {
                const int sx = io.x4 ? (x - 2) >> 2 : io.xofs[x];
                const int last = io.v_cols - 1;
                a = reinterpret_cast<const float4 *>(io.alpha)[x];
                c0 = sx - 1; c1 = sx; c2 = sx + 1; c3 = sx + 2;
                c0 = c0 < 0 ? 0 : (c0 > last ? last : c0);
                c1 = c1 < 0 ? 0 : (c1 > last ? last : c1);
                c2 = c2 < 0 ? 0 : (c2 > last ? last : c2);
                c3 = c3 < 0 ? 0 : (c3 > last ? last : c3);
            }
            const int y0 = tt.o2 * p.T + tt.o1;
            // The taps of a 16-lane tile row come from at most kSeg consecutive source columns
            // (8 for the x4 geometry).  Staging those runs in LDS -- the tile buffer is still free --
            // replaces four gathers per element and source by 1/4 staging load; per-lane gathers
            // stay as the fall-back for wider footprints.
            constexpr int kSeg = V >= 16 ? V / 2 : 8;
            const int xl = tt.lane0, xr = tt.lane0 + tt.nv - 1 < io.cols - 1 ? tt.lane0 + tt.nv - 1 : io.cols - 1;
            int cbase = 0, cend = -1;
            if (xl < io.cols) {
                const int lastc = io.v_cols - 1;
                int f = (io.x4 ? (xl - 2) >> 2 : io.xofs[xl]) - 1, l = (io.x4 ? (xr - 2) >> 2 : io.xofs[xr]) + 2;
                cbase = f < 0 ? 0 : (f > lastc ? lastc : f);
                cend = l < 0 ? 0 : (l > lastc ? lastc : l);
            }
            const bool staged = cend - cbase < kSeg && F * kSeg * 2 * sizeof(float) <= sizeof(float2) * F * Vp;
            if (staged) {
                float *seg = reinterpret_cast<float *>(buf);            // [2][F][kSeg]
                // The PAN window (u16 in the real slot) goes through LDS too when its rows are 16-byte
                // aligned: one 16-byte load per 8 pixels instead of eight 2-byte loads per lane -- the
                // loader is bound by the number of vector-memory instructions, not by bytes.
                unsigned short *seg16 = reinterpret_cast<unsigned short *>(seg + 2 * F * kSeg);      // [F][V]
                const bool wide16 = io.re16 && !io.re_v && V == 16 && (io.cols & 7) == 0 && (io.pitch_re16 & 7) == 0 &&
                                    ((size_t)io.re16 & 15) == 0 &&
                                    sizeof(float) * 2 * F * kSeg + sizeof(unsigned short) * F * V <= sizeof(float2) * F * Vp;
                // directly readable components first: their loads and the staging loads below are
                // then one round of memory latency, not two
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int n = (threadIdx.x >> VS) + i * (kFftBlock >> VS);
                    const int y = y0 + n * p.S;
                    float2 z = make_float2(0.f, 0.f);
                    if ((TOTAL % kFftBlock == 0 || n < F) && xin && y < io.rows) {
                        if (io.re_v) {}
                        else if (io.re) z.x = io.re[(size_t)y * io.cols + x];
                        else if (io.re16 && !wide16) z.x = __uint_as_float((unsigned)io.re16[(size_t)y * io.pitch_re16 + x]);
                        if (io.im_v) {}
                        else if (io.im) z.y = io.im[(size_t)y * io.cols + x];
                        else if (io.im16) z.y = __uint_as_float((unsigned)io.im16[(size_t)y * io.pitch_im16 + x]);
                    }
                    zz[i] = z;
                }
                if (wide16) {
                    for (int e = threadIdx.x; e < 2 * F; e += kFftBlock) {
                        const int n = e >> 1, half = e & 1;
                        const int y = y0 + n * p.S, xc = tt.lane0 + 8 * half;
                        uint4 px = make_uint4(0u, 0u, 0u, 0u);
                        if (y < io.rows && xc < io.cols) px = *reinterpret_cast<const uint4 *>(io.re16 + (size_t)y * io.pitch_re16 + xc);
                        *reinterpret_cast<uint4 *>(seg16 + n * V + 8 * half) = px;
                    }
                }
                // the tap runs: 8-byte loads where the run starts on an even column of an even-pitch image
                const bool pair_ok = (io.v_cols & 1) == 0 && (cbase & 1) == 0 && (kSeg & 1) == 0;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const float *__restrict__ Vs = sl ? io.im_v : io.re_v;
                    if (!Vs) continue;
                    if (pair_ok) {
                        constexpr int PERS2 = (F * kSeg / 2 + kFftBlock - 1) / kFftBlock;
#pragma unroll
                        for (int i = 0; i < PERS2; ++i) {
                            const int e = threadIdx.x + i * kFftBlock;
                            const int n = e / (kSeg / 2), j = 2 * (e % (kSeg / 2));
                            const int y = y0 + n * p.S;
                            if ((F * kSeg / 2 % kFftBlock == 0 || n < F) && y < io.rows) {
                                const float *row = Vs + (size_t)y * io.v_cols;
                                float2 v2;
                                if (cbase + j + 1 < io.v_cols) {
                                    v2 = *reinterpret_cast<const float2 *>(row + cbase + j);
                                } else {
                                    const int last = io.v_cols - 1;
                                    v2 = make_float2(row[cbase + j < last ? cbase + j : last], row[last]);
                                }
                                *reinterpret_cast<float2 *>(seg + (sl * F + n) * kSeg + j) = v2;
                            }
                        }
                    } else {
                        constexpr int PERS = (F * kSeg + kFftBlock - 1) / kFftBlock;
#pragma unroll
                        for (int i = 0; i < PERS; ++i) {
                            const int e = threadIdx.x + i * kFftBlock;
                            const int n = e / kSeg, j = e % kSeg;
                            const int y = y0 + n * p.S;
                            if ((F * kSeg % kFftBlock == 0 || n < F) && y < io.rows) {
                                const int c = cbase + j < io.v_cols ? cbase + j : io.v_cols - 1;
                                seg[(sl * F + n) * kSeg + j] = Vs[(size_t)y * io.v_cols + c];
                            }
                        }
                    }
                }
                __syncthreads();
                const int r0 = c0 - cbase, r1 = c1 - cbase, r2 = c2 - cbase, r3 = c3 - cbase;
                auto taps = [&](const float *__restrict__ row) {
                    float r = __fmul_rn(row[r0], a.x);
                    r = __fadd_rn(r, __fmul_rn(row[r1], a.y));
                    r = __fadd_rn(r, __fmul_rn(row[r2], a.z));
                    r = __fadd_rn(r, __fmul_rn(row[r3], a.w));
                    return r;
                };
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int n = (threadIdx.x >> VS) + i * (kFftBlock >> VS);
                    const int y = y0 + n * p.S;
                    if ((TOTAL % kFftBlock == 0 || n < F) && xin && y < io.rows) {
                        if (io.re_v) zz[i].x = taps(seg + n * kSeg);
                        else if (wide16) zz[i].x = __uint_as_float((unsigned)seg16[n * V + v]);
                        if (io.im_v) zz[i].y = taps(seg + (F + n) * kSeg);
                    }
                }
                __syncthreads();            // the staging area is the tile buffer the commit overwrites
                return;
            }
            auto taps = [&](const float *__restrict__ row) {
                float r = __fmul_rn(row[c0], a.x);
                r = __fadd_rn(r, __fmul_rn(row[c1], a.y));
                r = __fadd_rn(r, __fmul_rn(row[c2], a.z));
                r = __fadd_rn(r, __fmul_rn(row[c3], a.w));
                return r;
            };
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int n = (threadIdx.x >> VS) + i * (kFftBlock >> VS);
                const int y = y0 + n * p.S;
                float2 z = make_float2(0.f, 0.f);
                if ((TOTAL % kFftBlock == 0 || n < F) && xin && y < io.rows) {
                    if (io.re_v) z.x = taps(io.re_v + (size_t)y * io.v_cols);
                    else if (io.re) z.x = io.re[(size_t)y * io.cols + x];
                    else if (io.re16) z.x = __uint_as_float((unsigned)io.re16[(size_t)y * io.pitch_re16 + x]);
                    if (io.im_v) z.y = taps(io.im_v + (size_t)y * io.v_cols);
                    else if (io.im) z.y = io.im[(size_t)y * io.cols + x];
                    else if (io.im16) z.y = __uint_as_float((unsigned)io.im16[(size_t)y * io.pitch_im16 + x]);
                }
                zz[i] = z;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = threadIdx.x + i * kFftBlock;
            const int v = MODE == 0 ? (e & (V - 1)) : e / F;
            const int n = MODE == 0 ? (e >> VS) : e - v * F;
            zz[i] = make_float2(0.f, 0.f);
            if ((TOTAL % kFftBlock == 0 || e < TOTAL) && v < tt.nv) {
                const long off = MODE == 0 ? tt.base + (long)n * p.nstride + v : vec_offset(p, tt, n, v);
                zz[i] = IOK == 1 ? load_elem_raw(data, p, io, tt, n, v, off) : data[off];
            }
        }
    };
    if (IOK == 1 && (p.dbg & 1)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) zz[i] = make_float2(1.f, 0.f);
    } else {
        issue_loads(t);
    }
    for (int i = threadIdx.x; i < TWN; i += kFftBlock) tw[i] = twF[i];

    {
        if (tile_tw) {
            for (int i = threadIdx.x; i < F; i += kFftBlock) twj[i] = p.tw_rows ? twT[(long)t.o1 * F + i] : twT[(long)t.o1 * i];
            if (p.inverse) __syncthreads();
        }
        // commit the prefetched tile to LDS
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = threadIdx.x + i * kFftBlock;
            const int v = MODE == 0 ? (e & (V - 1)) : e / F;
            const int n = MODE == 0 ? (e >> VS) : e - v * F;
            if (TOTAL % kFftBlock == 0 || e < TOTAL) {
                float2 z = zz[i];
                if (raw_re16) z.x = (float)__float_as_uint(z.x);
                if (raw_im16) z.y = (float)__float_as_uint(z.y);
                if (p.inverse && v < t.nv) {
                    z.y = -z.y;
                    if (tile_tw) z = cmul(z, twj[n]);
                    else if (p.tw_mode == 1) z = cmul(z, twT[(long)(t.lane0 + v) * n]);
                }
                buf[n * Vp + v] = z;
            }
        }
        __syncthreads();

        if (!(IOK == 1 && (p.dbg & 2))) Stages<F, VS, Vp, NT, 1, Rs...>::run(buf, tw);

        if ((IOK == 2 && io.store_kind == 1)) {
            // the scan of the tile ends (barrier inside store_peak) before the scratch is written
            store_peak(buf, Vp, p, io, t, reinterpret_cast<float *>(buf + kFftBlock), reinterpret_cast<long *>(buf), t.gtile,
                       p.total_tiles, NT);
        } else if ((IOK == 2 && io.store_kind == 2)) {
            if (threadIdx.x == 0) {
                float2 z = buf[wn0 * Vp + wv0];
                io.window[(blockIdx.x / 25) * 32 + blockIdx.x % 25] = (blockIdx.x / 25) ? -z.y : z.x;
            }
        } else {
            for (int e = threadIdx.x; e < TOTAL; e += kFftBlock) {
                const int v = MODE == 0 ? (e & (V - 1)) : e / F;
                const int n = MODE == 0 ? (e >> VS) : e - v * F;
                if (v >= t.nv || (IOK == 1 && (p.dbg & 4))) continue;
                float2 z = buf[n * Vp + v];
                if (p.inverse) z.y = -z.y;
                else if (tile_tw) z = cmul(z, twj[n]);
                else if (p.tw_mode == 1) z = cmul(z, twT[(long)(t.lane0 + v) * n]);
                const long off = MODE == 0 ? t.base + (long)n * p.nstride + v : vec_offset(p, t, n, v);
                data[off] = z;
            }
        }
    }
}

// ---- last inverse column pass (peak role), 128 points x 16 lanes, first and last stage in registers ----------
// The generic pass kernel moves every point through LDS four times (commit, three stages) and reads it a fifth
// time for the arg-max scan; with eight workgroups per CU the SIMDs' vector issue is the limit (VALU active
// 12 % of the wave cycles x 8 waves per SIMD; profiles/r02_pmc_passes_before.json), not HBM.  Here thread
// (q, v) = (threadIdx >> 4, threadIdx & 15) loads the eight points q + 16 m of lane v straight from memory (for
// a fixed m a wave covers four 128-byte row segments) -- exactly the inputs of its radix-8 butterfly of the
// first Stockham stage -- and the outputs of the last radix-4 stage are scanned in registers: two LDS round
// trips and four barriers instead of five and nine.  The tile maximum is folded into the arg-max slots per
// WAVE (a value reduction on the DPP path, then an atomicMax by the lanes that hold the maximum -- one lane
// unless values tie, and atomicMax orders ties by key itself): no LDS hand-off between the waves.
// Round 2: 0.128 -> 0.104 ms per launch.  Round 3 took the pass apart (profiles/experiments/peak_dbg.sh,
// strided_rows_read.hip): its loads alone take 0.064 ms (the bare pattern reads at 6 TB/s), its arithmetic alone 0.069,
// loads + first stage 0.0955 -- and 19 % of the bare pattern's time came back when the tile's 128 inter-pass twiddles
// were GATHERED from table T at a stride of 8 o bytes: 128 more cache lines per tile, requested behind the data and
// requested behind the data.  They are now one contiguous 1 KiB row per tile row (get_pass_table): 0.104 -> 0.088 ms.
// Walking several tile rows per workgroup with the next tile's loads in flight (OIP_PEAK_TILES), or fetching two or three
// tiles up front, changes nothing.
//
// maximum over the 64 lanes of a wave on the DPP path (no LDS traffic): quad swaps, the two row mirrors, then the row
// broadcasts of gfx9 bring the maximum of everything to lane 63.  fmaxf semantics per step (a NaN loses against a number);
// the callers compare their own value with the result.
__device__ __forceinline__ float wave_max_f32(float v)
{
#define OIP_DPP_MAX(ctrl, rmask)                                                                                         \
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), \
                                                                        ctrl, rmask, 0xf, false)))
    OIP_DPP_MAX(0xB1, 0xf);       // quad_perm [1,0,3,2]
    OIP_DPP_MAX(0x4E, 0xf);       // quad_perm [2,3,0,1]
    OIP_DPP_MAX(0x141, 0xf);      // row_half_mirror
    OIP_DPP_MAX(0x140, 0xf);      // row_mirror
    OIP_DPP_MAX(0x142, 0xa);      // row_bcast15 into rows 1 and 3
    OIP_DPP_MAX(0x143, 0xc);      // row_bcast31 into rows 2 and 3
#undef OIP_DPP_MAX
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int VS>
__global__ __launch_bounds__(16 << VS) void fft_col128_peak_kernel(const float2 *__restrict__ data, OipFftPass p, OipFftIo io,
                                                              const float2 *__restrict__ twF, const float2 *__restrict__ twT)
{
    // VS = 4: 16 lanes (128-byte row segments), 256 threads; VS = 5: 32 lanes (256-byte segments), 512 threads.  Thread
    // (q, v) = (tid >> VS, tid & (V - 1)), q < 16, owns the points q + 16 m of lane v either way.
    constexpr int F = 128, V = 1 << VS, Vp = V + 1;
    __shared__ float2 buf[F * Vp];
    __shared__ float2 tw[96];                  // w_128^i, i < 96: the stage twiddles w, w^2, w^3 of both radix-4 stages are table entries
    __shared__ float2 twj[F];
    // lane tile: contiguous chunks per XCD (see decode_tile); the workgroup keeps it and walks the row offsets
    // o1 = blockIdx.y, + gridDim.y, ...: the loads of the next tile are issued before the later stages of the current one
    const int ltn = p.ltn > 0 ? p.ltn : p.lane_tiles;
    const int rel = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
    if (rel >= ltn) return;
    const int lt = p.lt0 + rel;
    const int lane0 = lt << VS;
    const int nv = p.lanes - lane0 < V ? (int)(p.lanes - lane0) : V;
    const int v = threadIdx.x & (V - 1);
    const bool lane_ok = v < nv;
    const long none = (long)p.M * p.N;
    const int ym = p.M >> 1, xm = p.N >> 1;
    int xs = lane0 + v + xm; if (xs >= p.N) xs -= p.N;

    int o1 = blockIdx.y;
    if (o1 >= p.O1) return;
    float2 x[8];
    float2 rtw = make_float2(1.f, 0.f);
    // Loads: a uniform 64-bit base per (tile row, m) plus ONE 32-bit byte offset per thread (the host checks that the array
    // is smaller than 2 GiB), so a load costs no vector arithmetic; lanes past the last column re-read the last valid one
    // (their values never reach the scan) instead of branching around every load.  The tile's 128 inter-pass twiddles are
    // one contiguous row of the [O1][F] table (get_pass_table), read by the first 128 threads and passed through LDS: one
    // coalesced KiB -- gathered from table T at o n they were 128 more cache lines per tile (0.103 -> 0.088 ms); a thread
    // loading its own eight straight from the row, without the LDS hop, was SLOWER (0.106 ms).
    const int vc = lane_ok ? v : nv - 1;
    auto fetch = [&](int o, int tid) {
        const unsigned voff = (unsigned)((tid >> VS) * (int)p.nstride + lane0 + vc) * 8u;
        const char *tb = reinterpret_cast<const char *>(data) + (long)o * p.o1_stride * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = *reinterpret_cast<const float2 *>(tb + (long)m * 16 * p.nstride * 8 + voff);
        if (tid < F) rtw = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(twT) + (unsigned)(p.tw_rows ? o * F + tid : o * tid) * 8u);
    };
    // inverse = conj(forward(conj(.))); the inter-pass twiddle of point n multiplies the conjugated input
    auto stage1 = [&](int tid) {
        const int qq = tid >> VS;
#pragma unroll
        for (int m = 0; m < 8; ++m) x[m] = cmul(make_float2(x[m].x, -x[m].y), twj[qq + 16 * m]);
        bf8(x);
#pragma unroll
        for (int m = 0; m < 8; ++m) buf[(qq * 8 + m) * Vp + v] = x[m];
    };
    fetch(o1, threadIdx.x);
    if (threadIdx.x < 96) tw[threadIdx.x] = twF[threadIdx.x];
    if (threadIdx.x < F) twj[threadIdx.x] = rtw;
    __syncthreads();
    stage1(threadIdx.x);
    __syncthreads();
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));           // per-iteration opaque copy: no loop-invariant stage addressing held in registers
        const int qq = tid >> VS;
        const int o1n = o1 + gridDim.y;
        const bool more = o1n < p.O1;
        if (more) fetch(o1n, tid);              // in flight under stages 2 and 3 and the scan of the current tile
        __builtin_amdgcn_sched_barrier(0);
        // stage 2: radix 4, Ns = 8
        float2 y[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = qq + 16 * i;
#pragma unroll
            for (int m = 0; m < 4; ++m) y[i][m] = buf[(b + 32 * m) * Vp + v];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = qq + 16 * i, k = b & 7;
            const float2 w1 = tw[k * 4], w2 = tw[k * 8], w3 = tw[k * 12];
            y[i][1] = cmul(y[i][1], w1);
            y[i][2] = cmul(y[i][2], w2);
            y[i][3] = cmul(y[i][3], w3);
            bf4(y[i]);
            const int j0 = (b - k) * 4 + k;
#pragma unroll
            for (int m = 0; m < 4; ++m) buf[(j0 + 8 * m) * Vp + v] = y[i][m];
        }
        __syncthreads();
        // stage 3: radix 4, Ns = 32: outputs n = b + 32 m stay in registers
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int b = qq + 16 * i;
#pragma unroll
            for (int m = 0; m < 4; ++m) y[i][m] = buf[(b + 32 * m) * Vp + v];
            const float2 w1 = tw[b], w2 = tw[2 * b], w3 = tw[3 * b];
            y[i][1] = cmul(y[i][1], w1);
            y[i][2] = cmul(y[i][2], w2);
            y[i][3] = cmul(y[i][3], w3);
            bf4(y[i]);
        }
        // Arg-max of the thread's eight points in fftShift-ed scan order: row y = o1 + n S moves to (y + M/2) mod M, so
        // the points n >= 64 (m = 2, 3) come first, each half in increasing n: (i, m) = (0,2) (1,2) (0,3) (1,3) (0,0)
        // (1,0) (0,1) (1,1).  Strict comparisons keep the first maximum; NaN never wins.
        float bv0 = -INFINITY, bv1 = -INFINITY;
        int bn0 = -1, bn1 = -1;
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
            const int i = s2 & 1, m = ((s2 >> 1) + 2) & 3;
            const int n = qq + 16 * i + 32 * m;
            const float re = y[i][m].x, im = -y[i][m].y;         // conj back
            if (re > bv0) { bv0 = re; bn0 = n; }
            if (im > bv1) { bv1 = im; bn1 = n; }
        }
        if (!lane_ok) { bv0 = bv1 = -INFINITY; bn0 = bn1 = -1; }
        const long gtile = (long)o1 * p.lane_tiles + lt;            // as decode_tile numbers them (O2 == 1)
        const int slot = (int)(gtile & (kPeakSlots - 1));
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            const float mine = part ? bv1 : bv0;
            const int bn = part ? bn1 : bn0;
            const float wmax = wave_max_f32(mine);
            // the generic kernel publishes (-inf, none) for a tile without any comparable value: keep that (lane 0 only)
            const bool holder = mine == wmax && (mine > -INFINITY || (tid & 63) == 0);
            if (holder) {
                long key = none;
                if (bn >= 0 && lane_ok) {
                    int ys = o1 + bn * p.S + ym; if (ys >= p.M) ys -= p.M;
                    key = (long)ys * p.N + xs;
                }
                const unsigned long long packed = oip_peak_pack(mine, key);
                if (packed) atomicMax(&io.slots[part * kPeakSlots + slot], packed);
            }
        }
        if (!more) break;
        __syncthreads();                        // every stage-3 read of buf and every use of twj is done
        if (tid < F) twj[tid] = rtw;
        __syncthreads();
        stage1(tid);
        __syncthreads();
        o1 = o1n;
    }
}

// ---- first forward column pass with fused x4 up-sampling: persistent over rows, register prefetch ----
// The generic fused loader above spends most of its time waiting: a tile's source rows are S rows apart
// (a DRAM page and a TLB entry each), eight workgroups per CU are all the LDS allows, and each of them
// sits through load -> LDS -> barrier -> taps -> barrier before its transform starts (measured: 0.09 of
// 0.22 ms).  This kernel is the same pass for the one geometry that matters (16-lane tiles, exact x4
// up-sampling, 16-byte aligned u16 rows; the host checks and otherwise falls back): a workgroup keeps its
// lane tile and walks the tile rows o1 = blockIdx.y, + gridDim.y, ..., and the raw source data of the
// NEXT tile -- one 16-byte load of PAN pixels, two 8-byte loads per up-sampled band, one twiddle -- is
// requested before the transform of the current one, so it lands while the butterflies and stores run.
template <int F, int NT, int... Rs>
__global__ __launch_bounds__(NT) void fft_first_pass_up_kernel(float2 *__restrict__ data, OipFftPass p, OipFftIo io,
                                                              const float2 *__restrict__ twF, const float2 *__restrict__ twT)
{
    constexpr int VS = 4, V = 16, Vp = V + 1, kSeg = 8;
    constexpr int TWN0 = TwTable<F, Rs...>::value();
    constexpr int TWN = (F == 128 && NT == 256 && TWN0 < 96) ? 96 : TWN0;       // the register-staged form reads w, w^2, w^3 of both radix-4 stages from the table
    constexpr int NLD = F * V / NT;                     // tile elements per thread
    constexpr int NRV = F * kSeg / 2 / NT;              // 8-byte tap-run loads per thread and band
    static_assert(F * V % NT == 0 && F * kSeg / 2 % NT == 0 && 2 * F <= NT && F <= NT && NT % V == 0, "tile / block shape");
    static_assert(sizeof(float) * 2 * F * kSeg + 2 * sizeof(unsigned short) * F * V <= sizeof(float2) * F * Vp, "staging fits the tile buffer");
    __shared__ float2 buf[F * Vp];
    __shared__ float2 tw[TWN];
    __shared__ float2 twj[F];
    float *seg = reinterpret_cast<float *>(buf);                                         // [2][F][kSeg] tap runs
    unsigned short *seg16 = reinterpret_cast<unsigned short *>(seg + 2 * F * kSeg);       // [F][V] PAN pixels
    unsigned short *seg16b = seg16 + F * V;                                               // [F][V] a second PAN window (imaginary slot)

    // lane tile: contiguous chunks per XCD (see decode_tile)
    const int ltn = p.ltn > 0 ? p.ltn : p.lane_tiles;
    const int rel = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
    if (rel >= ltn) return;
    const int lane0 = (p.lt0 + rel) << VS;
    const int nv = p.lanes - lane0 < V ? (int)(p.lanes - lane0) : V;
    const int o2 = blockIdx.z;
    const int v = threadIdx.x & (V - 1);
    const int x = lane0 + v;
    const bool xin = v < nv && x < io.cols;
    // horizontal taps of this thread's column (exact x4: first tap column = ((x - 2) >> 2) - 1)
    const int lastc = io.v_cols - 1;
    int cb = ((lane0 - 2) >> 2) - 1;
    const int cbase = cb < 0 ? 0 : (cb > lastc ? lastc : cb);
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xin && (io.re_v || io.im_v)) {
        const int sx = (x - 2) >> 2;
        a = reinterpret_cast<const float4 *>(io.alpha)[x];
        int c0 = sx - 1, c1 = sx, c2 = sx + 1, c3 = sx + 2;
        c0 = c0 < 0 ? 0 : (c0 > lastc ? lastc : c0);
        c1 = c1 < 0 ? 0 : (c1 > lastc ? lastc : c1);
        c2 = c2 < 0 ? 0 : (c2 > lastc ? lastc : c2);
        c3 = c3 < 0 ? 0 : (c3 > lastc ? lastc : c3);
        r0 = c0 - cbase; r1 = c1 - cbase; r2 = c2 - cbase; r3 = c3 - cbase;
    }
    const bool has16 = io.re16 != nullptr, has16b = io.im16 != nullptr;
    const float *__restrict__ V0 = io.re_v, *__restrict__ V1 = io.im_v;
    for (int i = threadIdx.x; i < TWN; i += NT) tw[i] = twF[i];

    // raw data of one tile: everything is loaded unconditionally from clamped addresses (a select on a
    // loaded value would wait for the load where it is issued); validity is applied when it is used
    uint4 rpx = make_uint4(0u, 0u, 0u, 0u), rpxb = make_uint4(0u, 0u, 0u, 0u);
    float2 rv[2][NRV];
    float2 rtw = make_float2(1.f, 0.f);
    auto fetch = [&](int o1, int tid) {
        const int y0 = o2 * p.T + o1;
        if (has16) {
            const int n = tid >> 1, half = tid & 1;
            int y = y0 + (n < F ? n : F - 1) * p.S;
            y = y < io.rows ? y : io.rows - 1;
            int xc = lane0 + 8 * half;
            xc = xc < io.cols ? xc : io.cols - 8;
            rpx = *reinterpret_cast<const uint4 *>(io.re16 + (size_t)y * io.pitch_re16 + xc);
            if (has16b) rpxb = *reinterpret_cast<const uint4 *>(io.im16 + (size_t)y * io.pitch_im16 + xc);
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const float *__restrict__ Vs = sl ? V1 : V0;
            if (!Vs) continue;
#pragma unroll
            for (int i = 0; i < NRV; ++i) {
                const int e = tid + i * NT;
                const int n = e / (kSeg / 2), j = 2 * (e % (kSeg / 2));
                int y = y0 + n * p.S;
                y = y < io.rows ? y : io.rows - 1;
                int c = cbase + j;
                c = c + 1 <= lastc ? c : lastc - 1;                 // the pair stays inside the row
                rv[sl][i] = *reinterpret_cast<const float2 *>(Vs + (size_t)y * io.v_cols + c);
            }
        }
        if (tid < F) rtw = p.tw_rows ? twT[(long)o1 * F + tid] : twT[(long)o1 * tid];
    };
    float2 zz[NLD];
    auto expand = [&](int o1, int tid) {
        const int y0 = o2 * p.T + o1;
        if (has16) {
            const int n = tid >> 1, half = tid & 1;
            if (n < F) {
                *reinterpret_cast<uint4 *>(seg16 + n * V + 8 * half) = rpx;
                if (has16b) *reinterpret_cast<uint4 *>(seg16b + n * V + 8 * half) = rpxb;
            }
        }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (!(sl ? V1 : V0)) continue;
#pragma unroll
            for (int i = 0; i < NRV; ++i) {
                const int e = tid + i * NT;
                const int n = e / (kSeg / 2), j = 2 * (e % (kSeg / 2));
                float2 v2 = rv[sl][i];
                // a pair pulled back from the row end holds column lastc in its second half
                if (cbase + j + 1 > lastc) v2 = make_float2(v2.y, v2.y);
                *reinterpret_cast<float2 *>(seg + (sl * F + n) * kSeg + j) = v2;
            }
        }
        if (tid < F) twj[tid] = rtw;
        __syncthreads();
        auto taps = [&](const float *__restrict__ row) {
            float r = __fmul_rn(row[r0], a.x);
            r = __fadd_rn(r, __fmul_rn(row[r1], a.y));
            r = __fadd_rn(r, __fmul_rn(row[r2], a.z));
            r = __fadd_rn(r, __fmul_rn(row[r3], a.w));
            return r;
        };
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int n = (tid >> VS) + i * (NT >> VS);
            const int y = y0 + n * p.S;
            float2 z = make_float2(0.f, 0.f);
            if (xin && y < io.rows) {
                if (has16) z.x = (float)seg16[n * V + v];
                else if (V0) z.x = taps(seg + n * kSeg);
                if (has16b) z.y = (float)seg16b[n * V + v];
                else if (V1) z.y = taps(seg + (F + n) * kSeg);
            }
            zz[i] = z;
        }
        __syncthreads();                    // the staging area is the tile buffer
    };

    int o1 = blockIdx.y;
    if (o1 >= p.O1) return;
    fetch(o1, threadIdx.x);
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));       // per-iteration opaque copy: no loop-invariant stage addressing in registers
        expand(o1, tid);
        const int o1n = o1 + gridDim.y;
        const bool more = o1n < p.O1;
        const long base = (long)o2 * p.o2_stride + (long)o1 * p.o1_stride + lane0;
        if (F == 128 && NT == 256 && !(p.dbg & 8)) {
            // The element a thread has just built, zz[i], is point (tid >> 4) + 16 i of lane v: the eight inputs of ITS
            // radix-8 butterfly of the first stage.  So the first stage runs on the registers and only its result goes
            // to LDS; the third stage's outputs are multiplied by the inter-pass twiddles and stored from registers
            // (a wave covers four 128-byte row segments per store instruction): two LDS round trips instead of four.
            const int qq = tid >> 4;
            bf8(zz);
#pragma unroll
            for (int m = 0; m < 8; ++m) buf[(qq * 8 + m) * Vp + v] = zz[m];
            __syncthreads();
            if (more) fetch(o1n, tid);               // in flight during the two remaining stages and the stores
            __builtin_amdgcn_sched_barrier(0);
            float2 y[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int b = qq + 16 * i;
#pragma unroll
                for (int m = 0; m < 4; ++m) y[i][m] = buf[(b + 32 * m) * Vp + v];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {            // radix 4, Ns = 8
                const int b = qq + 16 * i, k = b & 7;
                const float2 w1 = tw[k * 4], w2 = tw[k * 8], w3 = tw[k * 12];
                y[i][1] = cmul(y[i][1], w1);
                y[i][2] = cmul(y[i][2], w2);
                y[i][3] = cmul(y[i][3], w3);
                bf4(y[i]);
                const int j0 = (b - k) * 4 + k;
#pragma unroll
                for (int m = 0; m < 4; ++m) buf[(j0 + 8 * m) * Vp + v] = y[i][m];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i) {            // radix 4, Ns = 32: outputs n = b + 32 m
                const int b = qq + 16 * i;
#pragma unroll
                for (int m = 0; m < 4; ++m) y[i][m] = buf[(b + 32 * m) * Vp + v];
                const float2 w1 = tw[b], w2 = tw[2 * b], w3 = tw[3 * b];
                y[i][1] = cmul(y[i][1], w1);
                y[i][2] = cmul(y[i][2], w2);
                y[i][3] = cmul(y[i][3], w3);
                bf4(y[i]);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int n = b + 32 * m;
                    if (v < nv) data[base + (long)n * p.nstride + v] = cmul(y[i][m], twj[n]);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int n = (tid >> VS) + i * (NT >> VS);
            buf[n * Vp + (tid & (V - 1))] = zz[i];
        }
        __syncthreads();
        if (more) fetch(o1n, tid);               // in flight during the transform and the stores below
        __builtin_amdgcn_sched_barrier(0);
        Stages<F, VS, Vp, NT, 1, Rs...>::run(buf, tw, tid);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int n = (tid >> VS) + i * (NT >> VS);
            if (v < nv) data[base + (long)n * p.nstride + v] = cmul(buf[n * Vp + (tid & (V - 1))], twj[n]);
        }
        }
        if (!more) break;
        __syncthreads();                    // buf and twj are free for the next tile
        o1 = o1n;
    }
}

struct FirstUpKernel {
    int F, threads;
    void (*fn)(float2 *, OipFftPass, OipFftIo, const float2 *, const float2 *);
};
const FirstUpKernel kFirstUp[] = {
    {128, 256, fft_first_pass_up_kernel<128, 256, 8, 4, 4>},
};

// table of specialisations: (F, log2 V, mode) -> kernel
struct FastKernel {
    int F, vshift, mode, threads;
    void (*fn[3])(float2 *, OipFftPass, OipFftIo, const float2 *, const float2 *);         // by IOK: plain, fused load, fused store
};
const FastKernel kFast[] = {
    // column passes of 16000 = 125 * 128 (and other 5^3 / 2^7 factors): 16 lanes = 128-byte
    // segments, 17 KiB of LDS per workgroup -> 8 workgroups per CU (measured faster than 32 lanes)
    {125, 4, 0, 256, {fft_pass_ct_kernel<125, 4, 0, 256, 0, 5, 5, 5>, fft_pass_ct_kernel<125, 4, 0, 256, 1, 5, 5, 5>, fft_pass_ct_kernel<125, 4, 0, 256, 2, 5, 5, 5>}},
    {128, 4, 0, 256, {fft_pass_ct_kernel<128, 4, 0, 256, 0, 8, 4, 4>, fft_pass_ct_kernel<128, 4, 0, 256, 1, 8, 4, 4>, fft_pass_ct_kernel<128, 4, 0, 256, 2, 8, 4, 4>}},
    {125, 5, 0, 256, {fft_pass_ct_kernel<125, 5, 0, 256, 0, 5, 5, 5>, fft_pass_ct_kernel<125, 5, 0, 256, 1, 5, 5, 5>, fft_pass_ct_kernel<125, 5, 0, 256, 2, 5, 5, 5>}},
    {128, 5, 0, 256, {fft_pass_ct_kernel<128, 5, 0, 256, 0, 8, 4, 4>, fft_pass_ct_kernel<128, 5, 0, 256, 1, 8, 4, 4>, fft_pass_ct_kernel<128, 5, 0, 256, 2, 8, 4, 4>}},
    {100, 4, 0, 256, {fft_pass_ct_kernel<100, 4, 0, 256, 0, 4, 5, 5>, fft_pass_ct_kernel<100, 4, 0, 256, 1, 4, 5, 5>, fft_pass_ct_kernel<100, 4, 0, 256, 2, 4, 5, 5>}},
    {160, 4, 0, 256, {fft_pass_ct_kernel<160, 4, 0, 256, 0, 4, 8, 5>, fft_pass_ct_kernel<160, 4, 0, 256, 1, 4, 8, 5>, fft_pass_ct_kernel<160, 4, 0, 256, 2, 4, 8, 5>}},
    {64, 5, 0, 256, {fft_pass_ct_kernel<64, 5, 0, 256, 0, 4, 4, 4>, fft_pass_ct_kernel<64, 5, 0, 256, 1, 4, 4, 4>, fft_pass_ct_kernel<64, 5, 0, 256, 2, 4, 4, 4>}},
    // 4000 = 32 * 125: the column transform of the band windows themselves (a quarter of the correlation lines)
    {32, 5, 0, 256, {fft_pass_ct_kernel<32, 5, 0, 256, 0, 8, 4>, fft_pass_ct_kernel<32, 5, 0, 256, 1, 8, 4>, fft_pass_ct_kernel<32, 5, 0, 256, 2, 8, 4>}},
    // row passes: 30000/10, 12288/10 -> 1250, the 200-column stitch overlap
    {3000, 1, 1, 512, {fft_pass_ct_kernel<3000, 1, 1, 512, 0, 3, 8, 5, 5, 5>, fft_pass_ct_kernel<3000, 1, 1, 512, 1, 3, 8, 5, 5, 5>, fft_pass_ct_kernel<3000, 1, 1, 512, 2, 3, 8, 5, 5, 5>}},
    {3000, 0, 1, 512, {fft_pass_ct_kernel<3000, 0, 1, 512, 0, 3, 8, 5, 5, 5>, fft_pass_ct_kernel<3000, 0, 1, 512, 1, 3, 8, 5, 5, 5>, fft_pass_ct_kernel<3000, 0, 1, 512, 2, 3, 8, 5, 5, 5>}},
    {3000, 0, 1, 256, {fft_pass_ct_kernel<3000, 0, 1, 256, 0, 3, 8, 5, 5, 5>, fft_pass_ct_kernel<3000, 0, 1, 256, 1, 3, 8, 5, 5, 5>, fft_pass_ct_kernel<3000, 0, 1, 256, 2, 3, 8, 5, 5, 5>}},
    {1250, 1, 1, 256, {fft_pass_ct_kernel<1250, 1, 1, 256, 0, 2, 5, 5, 5, 5>, fft_pass_ct_kernel<1250, 1, 1, 256, 1, 2, 5, 5, 5, 5>, fft_pass_ct_kernel<1250, 1, 1, 256, 2, 2, 5, 5, 5, 5>}},
    {200, 4, 1, 256, {fft_pass_ct_kernel<200, 4, 1, 256, 0, 8, 5, 5>, fft_pass_ct_kernel<200, 4, 1, 256, 1, 8, 5, 5>, fft_pass_ct_kernel<200, 4, 1, 256, 2, 8, 5, 5>}},
};
constexpr int kNumFast = sizeof(kFast) / sizeof(kFast[0]);

// ---- host-side planning --------------------------------------------------------------------
bool smooth235(long n)
{
    if (n < 1) return false;
    for (int p : {2, 3, 5}) while (n % p == 0) n /= p;
    return n == 1;
}

std::vector<int> radix_list(int F)
{
    std::vector<int> r;
    int twos = 0;
    while (F % 2 == 0) { F /= 2; ++twos; }
    while (F % 3 == 0) { F /= 3; r.push_back(3); }
    for (; twos >= 3; twos -= 3) r.push_back(8);
    if (twos == 2) r.push_back(4);
    if (twos == 1) r.push_back(2);
    while (F % 5 == 0) { F /= 5; r.push_back(5); }
    return r;
}

// split L into pass factors: all but the last limited by max_a (mode A tiles), the last by
// max_last; fewest passes, then the most balanced split; 125*128 preferred for 16000
bool split_axis(int L, int max_a, int max_last, std::vector<int> *out)
{
    if (L <= max_last) { *out = {L}; return true; }
    for (int passes = 2; passes <= 4; ++passes) {
        std::vector<int> best;
        double best_score = -1;
        std::vector<int> cur;
        std::function<void(int, int)> rec = [&](int rem, int left) {
            if (left == 1) {
                if (rem <= max_last && rem >= 2) {
                    cur.push_back(rem);
                    int mn = 1 << 30;
                    for (int f : cur) mn = f < mn ? f : mn;
                    if (mn > best_score) { best_score = mn; best = cur; }
                    cur.pop_back();
                }
                return;
            }
            for (int f = 2; f <= max_a && f <= rem; ++f) {
                if (rem % f) continue;
                cur.push_back(f);
                rec(rem / f, left - 1);
                cur.pop_back();
            }
        };
        rec(L, passes);
        if (!best.empty()) { *out = best; return true; }
    }
    return false;
}

int pick_vshift(int F, int want)
{
    int vs = 0;
    while ((1 << (vs + 1)) <= want && (long)F * ((1 << (vs + 1)) + 1) <= kMaxTileElems) ++vs;
    return vs;
}

void choose_kernel(OipFftPass *p, int want_v)
{
    p->fast = -1;
    // experiment knobs (first match in the table is the default): lanes per column tile,
    // threads per row-pass workgroup
    static const char *env = getenv("OIP_FFT_LANES");
    static const char *envt = getenv("OIP_FFT_ROW_THREADS");
    const int want_vs = env ? (atoi(env) == 16 ? 4 : 5) : -1;
    const int want_nt = envt ? atoi(envt) : -1;
    static const char *envr = getenv("OIP_FFT_ROW_VS");
    const int want_rvs = envr ? atoi(envr) : -1;
    for (int i = 0; i < kNumFast; ++i)
        if (kFast[i].F == p->F && kFast[i].mode == p->mode && (p->mode == 1 || want_vs < 0 || kFast[i].vshift == want_vs) &&
            (p->mode == 0 || want_nt < 0 || kFast[i].threads == want_nt) &&
            (p->mode == 0 || want_rvs < 0 || kFast[i].vshift == want_rvs)) {
            p->fast = i;
            p->vshift = kFast[i].vshift;
            p->Vp = p->vshift > 1 ? (1 << p->vshift) + 1 : (1 << p->vshift);
            return;
        }
    p->vshift = pick_vshift(p->F, want_v);
    p->Vp = p->vshift ? (1 << p->vshift) + 1 : 1;
}

}  // namespace

struct oip_fft_state {
    std::map<int, float2 *> tables;       // exp(-2 pi i t / T), t in [0, T)
    std::map<std::pair<int, int>, float2 *> pass_tables;   // (T, F): [T / F][F], row o = exp(-2 pi i o n / T), n in [0, F)
    std::map<std::pair<int, int>, OipFft2dPlan> plans;
};

static int get_table(oip_ctx *ctx, int T, const float2 **out)
{
    if (!ctx->fft) ctx->fft = new oip_fft_state();
    auto it = ctx->fft->tables.find(T);
    if (it == ctx->fft->tables.end()) {
        std::vector<float2> h(T);
        for (int t = 0; t < T; ++t) {
            double a = -2.0 * M_PI * (double)t / (double)T;
            h[t] = make_float2((float)cos(a), (float)sin(a));
        }
        float2 *d = nullptr;
        OIP_HIP(ctx, hipMalloc((void **)&d, sizeof(float2) * T));
        OIP_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * T, hipMemcpyHostToDevice));
        it = ctx->fft->tables.emplace(T, d).first;
    }
    *out = it->second;
    return OIP_OK;
}

// The inter-pass twiddles of a tw_mode 2 pass, one contiguous row of F values per tile row o (the same values as
// table T at o n, computed the same way).  A tile used to gather its F values from table T at a stride of 8 o bytes:
// F different cache lines per tile -- as many line requests as the tile's data (128 rows of one line each), on the
// dependent end of the tile's loads; the bare access pattern of the last inverse pass lost 19 % to that gather alone
// (profiles/experiments/strided_rows_read.hip, "pass-like 1").
static int get_pass_table(oip_ctx *ctx, int T, int F, const float2 **out)
{
    if (!ctx->fft) ctx->fft = new oip_fft_state();
    auto key = std::make_pair(T, F);
    auto it = ctx->fft->pass_tables.find(key);
    if (it == ctx->fft->pass_tables.end()) {
        const int O = T / F;
        std::vector<float2> h((size_t)O * F);
        for (int o = 0; o < O; ++o)
            for (int n = 0; n < F; ++n) {
                const long t = ((long)o * n) % T;
                double a = -2.0 * M_PI * (double)t / (double)T;
                h[(size_t)o * F + n] = make_float2((float)cos(a), (float)sin(a));
            }
        float2 *d = nullptr;
        OIP_HIP(ctx, hipMalloc((void **)&d, sizeof(float2) * h.size()));
        OIP_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice));
        it = ctx->fft->pass_tables.emplace(key, d).first;
    }
    *out = it->second;
    return OIP_OK;
}

void oip_fft_destroy(oip_ctx *ctx)
{
    if (!ctx->fft) return;
    for (auto &kv : ctx->fft->tables) hipFree(kv.second);
    for (auto &kv : ctx->fft->pass_tables) hipFree(kv.second);
    for (auto &kv : ctx->fft->plans) hipFree(const_cast<int *>(kv.second.d_ypos));
    delete ctx->fft;
    ctx->fft = nullptr;
}

static void fill_radix(OipFftPass *p)
{
    std::vector<int> r = radix_list(p->F);
    p->nradix = (int)r.size();
    for (int i = 0; i < p->nradix; ++i) p->radix[i] = r[i];
}

int oip_fft2d_plan(oip_ctx *ctx, int M, int N, const OipFft2dPlan **out)
{
    if (!ctx->fft) ctx->fft = new oip_fft_state();
    auto key = std::make_pair(M, N);
    auto it = ctx->fft->plans.find(key);
    if (it != ctx->fft->plans.end()) { *out = &it->second; return OIP_OK; }
    if (!smooth235(M) || !smooth235(N)) return oip_fail(ctx, OIP_E_INVALID, "fft2d: %d x %d is not 2^a3^b5^c", M, N);
    if (M < 2) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: fewer than 2 rows");
    OipFft2dPlan pl;
    pl.M = M; pl.N = N;
    // row pitch padded to whole 128-byte lines (16 complex): every pass stores full lines
    const int P = (N + 15) / 16 * 16;
    pl.P = P;
    if (!split_axis(N, 256, 4096, &pl.xf)) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: cannot factor row length %d", N);
    // 16000 = 128 * 125 with the 128-point pass first: its points are 125 rows apart, an odd multiple of
    // 512 B for the padded pitches, so a tile spreads over the HBM channels (125 * 128 would put all
    // points of a tile 47 * 2^16 B apart -- same channel -- and ran at half the bandwidth)
    static const char *envo = getenv("OIP_FFT_Y_ORDER");
    if (M == 16000) pl.yf = (envo && atoi(envo) == 125) ? std::vector<int>{125, 128} : std::vector<int>{128, 125};
    else if (M == 4000) pl.yf = std::vector<int>{32, 125};          // the band windows of 16000 correlation lines
    else if (!split_axis(M, 256, 256, &pl.yf)) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: cannot factor column length %d", M);
    // column (y) passes first: always mode A with lanes = x
    {
        int T = M;
        for (size_t i = 0; i < pl.yf.size(); ++i) {
            OipFftPass p;
            memset(&p, 0, sizeof p);
            p.F = pl.yf[i];
            p.M = M; p.N = N; p.P = P;
            p.axis = 1;
            fill_radix(&p);
            const int S = T / p.F;
            p.mode = 0;
            choose_kernel(&p, 32);
            p.nstride = (long)S * P;
            p.lanes = N;
            p.lane_tiles = (N + (1 << p.vshift) - 1) >> p.vshift;
            p.O1 = S;     p.o1_stride = P;              // j: row offset inside the block
            p.O2 = M / T; p.o2_stride = (long)T * P;    // blocks
            p.tw_mode = S > 1 ? 2 : 0; p.T = T; p.S = S;
            pl.passes.push_back(p);
            T = S;
        }
    }
    pl.n_y = (int)pl.passes.size();
    // row (x) passes: leading factors in mode A (lanes = j), last factor as whole contiguous
    // sub-rows in mode B
    {
        int T = N;
        for (size_t i = 0; i < pl.xf.size(); ++i) {
            OipFftPass p;
            memset(&p, 0, sizeof p);
            p.F = pl.xf[i];
            p.M = M; p.N = N; p.P = P;
            p.axis = 0;
            fill_radix(&p);
            const int S = T / p.F;
            p.T = T; p.S = S;
            if (S == 1) {
                p.mode = 1;
                choose_kernel(&p, 32);
                p.lanes = (long)M * (N / p.F);          // contiguous F-point vectors
                p.tw_mode = 0;
            } else {
                p.mode = 0;
                choose_kernel(&p, 32);
                p.nstride = S;
                p.lanes = S;
                p.lane_tiles = (S + (1 << p.vshift) - 1) >> p.vshift;
                p.O1 = N / T; p.o1_stride = T;          // blocks of the current sub-problem
                p.O2 = M;     p.o2_stride = P;          // rows
                p.tw_mode = 1;
            }
            pl.passes.push_back(p);
            T = S;
        }
    }
    for (auto &p : pl.passes) {
        if (p.fast < 0 && (long)p.F * p.Vp > kMaxTileElems)
            return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: factor %d too long for one LDS tile", p.F);
        const float2 *t;
        int rc = get_table(ctx, p.F, &t);
        if (rc) return rc;
        if (p.tw_mode) { rc = get_table(ctx, p.T, &t); if (rc) return rc; }
    }
    // row positions of the frequency lines (the row stage of the correlation walks ky and M - ky)
    {
        OipAxisDigits d;
        d.n = (int)pl.yf.size();
        d.L = M;
        for (int i = 0; i < 4; ++i) d.f[i] = i < d.n ? pl.yf[i] : 1;
        std::vector<int> ypos(M);
        for (int k = 0; k < M; ++k) ypos[k] = oip_freq_to_pos(d, k);
        int *dp = nullptr;
        OIP_HIP(ctx, hipMalloc((void **)&dp, sizeof(int) * M));
        OIP_HIP(ctx, hipMemcpy(dp, ypos.data(), sizeof(int) * M, hipMemcpyHostToDevice));
        pl.d_ypos = dp;
    }
    it = ctx->fft->plans.emplace(key, pl).first;
    *out = &it->second;
    return OIP_OK;
}

static long pass_blocks(const OipFftPass &p)
{
    if (p.mode == 0) return (long)(p.ltn > 0 ? p.ltn : p.lane_tiles) * p.O1 * p.O2;
    return (p.lanes + (1 << p.vshift) - 1) >> p.vshift;
}


static int launch_pass(oip_ctx *ctx, float2 *data, OipFftPass p, int inverse, const OipFftIo &io, long blocks_override)
{
    p.inverse = inverse;
    const float2 *twF = nullptr, *twT = nullptr;
    int rc = get_table(ctx, p.F, &twF);
    if (rc) return rc;
    if (p.tw_mode) { rc = get_table(ctx, p.T, &twT); if (rc) return rc; }
    // the specialised kernels of a tw_mode 2 pass read their F inter-pass twiddles as one contiguous row (get_pass_table)
    const float2 *twR = twT;
    bool rows_ok = false;
    p.tw_rows = 0;
    if (p.tw_mode == 2 && p.F > 0 && p.T % p.F == 0 && p.O1 == p.T / p.F && (long)p.T * 8 <= (16L << 20)) {
        const char *envr = getenv("OIP_TW_ROWS");                        // read per call (a test compares the forms): 0 = gather from table T
        if (!(envr && atoi(envr) == 0)) {
            rc = get_pass_table(ctx, p.T, p.F, &twR);
            if (rc) return rc;
            rows_ok = true;
        }
    }
    {
        OipFftPass whole = p;
        whole.ltn = 0;
        p.total_tiles = pass_blocks(whole);
    }
    long blocks = blocks_override > 0 ? blocks_override : pass_blocks(p);
    if (blocks <= 0 || blocks > 0x7fffffffL) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft pass grid too large");
    char pname[64];
    snprintf(pname, sizeof pname, blocks_override > 0 ? "fft_window_F%d" : (p.fast >= 0 ? "fft_pass_ct_kernel_F%d%s%s" : "fft_pass_kernel_F%d%s%s"), p.F,
             io.load_kind == 1 ? "_pack" : (io.store_kind == 1 ? "_peak" : ""), blocks_override > 0 || !ctx->prof_tag ? "" : ctx->prof_tag);
    OipProfScope prof(ctx, pname);
    { const char *e = getenv("OIP_PACK_DBG"); p.dbg = e ? atoi(e) : 0; }
    dim3 grid3((unsigned)blocks);
    p.grid3 = 0;
    if (p.mode == 0 && (p.O1 > 65535 || p.O2 > 65535)) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft pass: more than 65535 rows or blocks");
    if (p.mode == 0 && blocks_override <= 0) {
        p.grid3 = 1;
        const int ltn = p.ltn > 0 ? p.ltn : p.lane_tiles;
        p.xcd_chunk = (ltn + 7) / 8;
        grid3 = dim3((unsigned)(8 * p.xcd_chunk), (unsigned)p.O1, (unsigned)p.O2);
    }
    // the persistent first pass with fused up-sampling, when the geometry allows (see the kernel)
    // sources it takes: up-sampled V images (exact x4) with or without a u16 window in the real slot, or one / two u16
    // windows alone (the PAN windows of a pair of units)
    const bool up_v = (io.re_v || io.im_v) && io.x4 && !io.im16 && !(io.re16 && io.re_v) && (io.v_cols & 1) == 0 && io.v_cols >= 8 &&
                      ((size_t)io.re_v & 7) == 0 && ((size_t)io.im_v & 7) == 0;
    const bool pure16 = io.re16 && !io.re_v && !io.im_v &&
                        (!io.im16 || ((io.pitch_im16 & 7) == 0 && ((size_t)io.im16 & 15) == 0));
    if (blocks_override <= 0 && !inverse && io.load_kind == 1 && p.mode == 0 && p.axis == 1 && p.vshift == 4 && p.tw_mode == 2 &&
        (up_v || pure16) && !io.re && !io.im && io.cols >= 8 && io.rows >= 1 &&
        (!io.re16 || ((io.cols & 7) == 0 && (io.pitch_re16 & 7) == 0 && ((size_t)io.re16 & 15) == 0))) {
        static const char *envu = getenv("OIP_FIRST_UP");                 // experiment knob: 0 disables, N = tile rows per workgroup
        const int tiles = envu ? atoi(envu) : 3;
        for (const FirstUpKernel &k : kFirstUp)
            if (k.F == p.F && tiles > 0) {
                const int ltn = p.ltn > 0 ? p.ltn : p.lane_tiles;
                p.xcd_chunk = (ltn + 7) / 8;
                const int gy = (p.O1 + tiles - 1) / tiles;
                p.tw_rows = rows_ok;
                hipLaunchKernelGGL(k.fn, dim3((unsigned)(8 * p.xcd_chunk), (unsigned)gy, (unsigned)p.O2), dim3(k.threads), 0, ctx->stream,
                                   data, p, io, twF, twR);
                OIP_HIP(ctx, hipGetLastError());
                return OIP_OK;
            }
    }
    // the register-staged peak pass for the 128-point inverse column pass (see the kernel)
    if (blocks_override <= 0 && inverse && io.store_kind == 1 && p.mode == 0 && p.axis == 1 && p.F == 128 && (p.vshift == 4 || p.vshift == 5) &&
        p.tw_mode == 2 && p.grid3 && (16 * p.nstride + p.lanes) * 8 < (1L << 31)) {       // 32-bit byte offsets inside a tile row block
        static const char *envk = getenv("OIP_PEAK_V2");                  // experiment knob: 0 = generic pass kernel
        if (!(envk && atoi(envk) == 0)) {
            static const char *envt = getenv("OIP_PEAK_TILES");             // experiment knob: tile rows per workgroup
            const int tiles = envt && atoi(envt) > 0 ? atoi(envt) : 1;      // measured: 0.104 / 0.106 / 0.106 / 0.110 ms for 1 / 2 / 3 / 5
            if (p.O2 != 1) return oip_fail(ctx, OIP_E_RUNTIME, "peak pass: the last inverse pass spans the whole axis");
            // experiment knob: 32-lane tiles (256-byte row segments, 512 threads) for this pass alone, whatever the plan chose
            static const char *envl = getenv("OIP_PEAK_LANES");
            if (envl && atoi(envl) == 32 && p.vshift == 4 && p.ltn <= 0) {
                p.vshift = 5;
                p.Vp = 33;
                p.lane_tiles = (int)((p.lanes + 31) >> 5);
                p.xcd_chunk = (p.lane_tiles + 7) / 8;
                grid3.x = (unsigned)(8 * p.xcd_chunk);
            }
            grid3.y = (unsigned)((p.O1 + tiles - 1) / tiles);
            p.tw_rows = rows_ok;
            if (p.vshift == 4) hipLaunchKernelGGL(fft_col128_peak_kernel<4>, grid3, dim3(256), 0, ctx->stream, data, p, io, twF, twR);
            else hipLaunchKernelGGL(fft_col128_peak_kernel<5>, grid3, dim3(512), 0, ctx->stream, data, p, io, twF, twR);
            OIP_HIP(ctx, hipGetLastError());
            return OIP_OK;
        }
    }
    if (p.fast >= 0 && kFast[p.fast].fn[io.load_kind ? 1 : (io.store_kind ? 2 : 0)]) {
        p.ntiles = blocks;
        long grid = blocks;
        // the LDS-staged pass kernels keep the gather: with rows the 125-point passes were 1.4 % SLOWER (0.1328 -> 0.1347 ms, A/B twice on
        // one box), OIP_TW_ROWS=2 turns the rows on for them too
        const char *envr2 = getenv("OIP_TW_ROWS");
        p.tw_rows = rows_ok && p.mode == 0 && envr2 && atoi(envr2) == 2;
        hipLaunchKernelGGL(kFast[p.fast].fn[io.load_kind ? 1 : (io.store_kind ? 2 : 0)], p.grid3 ? grid3 : dim3((unsigned)grid), dim3(kFast[p.fast].threads), 0, ctx->stream, data, p, io, twF, p.tw_rows ? twR : twT);
    } else {
        size_t lds = sizeof(float2) * ((size_t)2 * p.F * p.Vp + p.F);
        hipLaunchKernelGGL(fft_pass_kernel, p.grid3 ? grid3 : dim3((unsigned)blocks), dim3(kFftBlock), lds, ctx->stream, data, p, io, twF, twT);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// In-place complex 2-D transform of an M x N row-major float2 array.
//   inverse == 0: natural order in, digit-scrambled spectrum out (columns first, then rows);
//                 io->load_kind applies to the first pass
//   inverse == 1: scrambled spectrum in, natural order out, unnormalised (rows, then columns);
//                 io->store_kind applies to the last pass
int oip_fft2d_exec(oip_ctx *ctx, const OipFft2dPlan *pl, float2 *data, int inverse, const OipFftIo *io, int rows_done)
{
    // rows_done (inverse only): the row passes were already applied by a fused kernel
    const int np = rows_done ? pl->n_y : (int)pl->passes.size();
    OipFftIo plain;
    memset(&plain, 0, sizeof plain);
    // Column passes are independent per column, so they can run panel by panel (a fraction of
    // the columns through all column passes, then the next fraction): the later pass then re-reads
    // what the earlier one just wrote while it is still in the 256 MiB Infinity Cache.
    static const char *envp = getenv("OIP_FFT_PANELS");
    int panels = envp ? atoi(envp) : 1;
    const int lane_tiles = pl->n_y > 0 ? pl->passes[0].lane_tiles : 1;
    if (panels < 1) panels = 1;
    if (panels > lane_tiles) panels = lane_tiles;
    auto column_passes = [&](bool inv) -> int {
        for (int pn = 0; pn < panels; ++pn) {
            for (int k = 0; k < pl->n_y; ++k) {
                const int i = inv ? pl->n_y - 1 - k : k;
                OipFftPass p = pl->passes[i];
                // the panel in THIS pass' lane tiles (the passes of one axis may use different tile widths); a panel
                // boundary must fall on a boundary of every pass, so panels are cut in units of the widest tile
                int vmax = 0;
                for (int q = 0; q < pl->n_y; ++q) vmax = pl->passes[q].vshift > vmax ? pl->passes[q].vshift : vmax;
                const int wide = (int)((p.lanes + (1L << vmax) - 1) >> vmax);              // tiles of the widest pass
                const int w0 = (int)((long)wide * pn / panels), w1 = (int)((long)wide * (pn + 1) / panels);
                const int lt0 = w0 << (vmax - p.vshift);
                int lt1 = w1 << (vmax - p.vshift);
                if (lt1 > p.lane_tiles || pn == panels - 1) lt1 = p.lane_tiles;
                p.lt0 = lt0;
                p.ltn = lt1 - lt0;
                OipFftIo use = plain;
                if (!inv && i == 0 && io) { use = *io; use.store_kind = 0; }
                if (inv && i == 0 && io) { use.store_kind = io->store_kind; use.slots = io->slots; }
                int rc = launch_pass(ctx, data, p, inv ? 1 : 0, use, 0);
                if (rc) return rc;
            }
        }
        return OIP_OK;
    };
    if (!inverse) {
        int rc = column_passes(false);
        if (rc) return rc;
        for (int i = pl->n_y; i < np; ++i) {
            OipFftIo use = plain;
            if (i == 0 && io) { use = *io; use.store_kind = 0; }
            rc = launch_pass(ctx, data, pl->passes[i], 0, use, 0);
            if (rc) return rc;
        }
    } else {
        for (int i = np - 1; i >= pl->n_y; --i) {
            OipFftIo use = plain;
            if (i == 0 && io) { use.store_kind = io->store_kind; use.slots = io->slots; }
            int rc = launch_pass(ctx, data, pl->passes[i], 1, use, 0);
            if (rc) return rc;
        }
        int rc = column_passes(true);
        if (rc) return rc;
    }
    return OIP_OK;
}

int oip_fft_table(oip_ctx *ctx, int T, const float2 **out) { return get_table(ctx, T, out); }

int oip_fft2d_window(oip_ctx *ctx, const OipFft2dPlan *pl, float2 *data, const OipFftIo *io)
{
    OipFftIo use;
    memset(&use, 0, sizeof use);
    use.store_kind = 2;
    use.peak_key = io->peak_key;
    use.slots = io->slots;
    use.window = io->window;
    use.part = io->part;
    return launch_pass(ctx, data, pl->passes[0], 1, use, 25L * (io->part > 0 ? io->part : 1));
}
