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
//     F-point sub-transforms of one factor with a Stockham radix-{2,3,4,5} network in LDS,
//     for a tile of V independent transforms that are ADJACENT IN MEMORY, so global loads
//     and stores are V*8-byte (128..256 B) contiguous segments whatever the axis:
//        mode A  (strided points, contiguous lanes): column passes and the leading row passes
//        mode B  (contiguous points):                the last row pass (whole F-point rows)
//   * decimation in frequency forward / decimation in time inverse: the forward transform
//     leaves each axis in digit-scrambled order (position p = k1*F2 + k2 holds frequency
//     k1 + F1*k2), the point-wise kernels work on that order, and the inverse consumes it and
//     returns natural order.  No transposes, no bit-reversal passes: a 16000 x 3000 transform
//     is 3 passes over HBM (row, column x2), each pass 8 B read + 8 B written per point.
//   * twiddles come from fp64-computed tables (per sub-length in LDS, per pass in HBM/L2).
// The inverse is unnormalised, like cv::idft without DFT_SCALE (phasecorr.cpp).
#include "oip_internal.h"
#include "oip_fft.h"

#include <cmath>
#include <map>

namespace {

constexpr int kFftBlock = 256;
constexpr int kMaxTileElems = 5120;       // F * (V+1) complex elements per LDS buffer

__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmuli_neg(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// forward butterflies, w = exp(-2 pi i / r)
__device__ __forceinline__ void bf2(float2 *x)
{
    float2 a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}
__device__ __forceinline__ void bf3(float2 *x)
{
    const float s = 0.86602540378443864676f;
    float2 t = cadd(x[1], x[2]);
    float2 d = cscale(cmuli_neg(csub(x[1], x[2])), s);
    float2 m = make_float2(x[0].x - 0.5f * t.x, x[0].y - 0.5f * t.y);
    x[0] = cadd(x[0], t);
    x[1] = cadd(m, d);
    x[2] = csub(m, d);
}
__device__ __forceinline__ void bf4(float2 *x)
{
    float2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
    float2 s13 = cadd(x[1], x[3]), d13 = cmuli_neg(csub(x[1], x[3]));
    x[0] = cadd(s02, s13);
    x[2] = csub(s02, s13);
    x[1] = cadd(d02, d13);
    x[3] = csub(d02, d13);
}
__device__ __forceinline__ void bf5(float2 *x)
{
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]);
    float2 t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
    float2 m1 = make_float2(x[0].x + c1 * t1.x + c2 * t2.x, x[0].y + c1 * t1.y + c2 * t2.y);
    float2 m2 = make_float2(x[0].x + c2 * t1.x + c1 * t2.x, x[0].y + c2 * t1.y + c1 * t2.y);
    float2 u1 = cmuli_neg(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));
    float2 u2 = cmuli_neg(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));
    x[0] = cadd(x[0], cadd(t1, t2));
    x[1] = cadd(m1, u1);
    x[4] = csub(m1, u1);
    x[2] = cadd(m2, u2);
    x[3] = csub(m2, u2);
}

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
        if (R == 2) bf2(x);
        else if (R == 3) bf3(x);
        else if (R == 4) bf4(x);
        else bf5(x);
        const int j0 = (b - k) * R + k;    // (b / Ns) * Ns * R + k
#pragma unroll
        for (int m = 0; m < R; ++m) out[(j0 + m * Ns) * Vp + v] = x[m];
    }
}

__global__ __launch_bounds__(kFftBlock) void fft_pass_kernel(float2 *__restrict__ data, OipFftPass p,
                                                             const float2 *__restrict__ twF,
                                                             const float2 *__restrict__ twT)
{
    extern __shared__ float2 smem[];
    const int F = p.F;
    const int V = 1 << p.vshift;
    const int Vp = p.Vp;
    float2 *bufA = smem;
    float2 *bufB = smem + F * Vp;
    float2 *tw = smem + 2 * F * Vp;
    for (int i = threadIdx.x; i < F; i += kFftBlock) tw[i] = twF[i];

    const long bid = blockIdx.x;
    long base;
    int nv, lane0 = 0, o1 = 0;
    if (p.mode == 0) {
        const int lt = (int)(bid % p.lane_tiles);
        const long rest = bid / p.lane_tiles;
        o1 = (int)(rest % p.O1);
        const long o2 = rest / p.O1;
        lane0 = lt << p.vshift;
        nv = p.lanes - lane0 < V ? p.lanes - lane0 : V;
        base = o2 * p.o2_stride + (long)o1 * p.o1_stride + lane0;
        const int total = F << p.vshift;
        for (int e = threadIdx.x; e < total; e += kFftBlock) {
            const int v = e & (V - 1), n = e >> p.vshift;
            float2 z = make_float2(0.f, 0.f);
            if (v < nv) {
                z = data[base + (long)n * p.nstride + v];
                if (p.inverse) {
                    z.y = -z.y;
                    if (p.tw_mode) {
                        const int j = p.tw_mode == 1 ? lane0 + v : o1;
                        z = cmul(z, twT[(long)j * n]);
                    }
                }
            }
            bufA[n * Vp + v] = z;
        }
    } else {
        const long vec0 = bid << p.vshift;
        nv = p.lanes - vec0 < V ? (int)(p.lanes - vec0) : V;
        base = vec0 * F;
        const int total = nv * F;
        for (int e = threadIdx.x; e < total; e += kFftBlock) {
            const int v = e / F, n = e - v * F;
            float2 z = data[base + e];
            if (p.inverse) z.y = -z.y;
            bufA[n * Vp + v] = z;
        }
        for (int e = total + threadIdx.x; e < (F << p.vshift); e += kFftBlock) {
            const int v = e / F, n = e - v * F;
            bufA[n * Vp + v] = make_float2(0.f, 0.f);
        }
    }
    __syncthreads();

    int Ns = 1;
    for (int s = 0; s < p.nradix; ++s) {
        const int r = p.radix[s];
        if (r == 4) stockham_stage<4>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else if (r == 5) stockham_stage<5>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else if (r == 2) stockham_stage<2>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        else stockham_stage<3>(bufA, bufB, tw, F, Ns, p.vshift, Vp);
        __syncthreads();
        float2 *t = bufA; bufA = bufB; bufB = t;
        Ns *= r;
    }

    if (p.mode == 0) {
        const int total = F << p.vshift;
        for (int e = threadIdx.x; e < total; e += kFftBlock) {
            const int v = e & (V - 1), n = e >> p.vshift;
            if (v >= nv) continue;
            float2 z = bufA[n * Vp + v];
            if (p.inverse) {
                z.y = -z.y;
            } else if (p.tw_mode) {
                const int j = p.tw_mode == 1 ? lane0 + v : o1;
                z = cmul(z, twT[(long)j * n]);
            }
            data[base + (long)n * p.nstride + v] = z;
        }
    } else {
        const int total = nv * F;
        for (int e = threadIdx.x; e < total; e += kFftBlock) {
            const int v = e / F, n = e - v * F;
            float2 z = bufA[n * Vp + v];
            if (p.inverse) z.y = -z.y;
            data[base + e] = z;
        }
    }
}

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
    for (; twos >= 2; twos -= 2) r.push_back(4);
    if (twos) r.push_back(2);
    while (F % 3 == 0) { F /= 3; r.push_back(3); }
    while (F % 5 == 0) { F /= 5; r.push_back(5); }
    return r;
}

// split L into pass factors: all but the last limited by max_a (mode A tiles), the last by
// max_last; fewest passes, then the most balanced split
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

}  // namespace

struct oip_fft_state {
    std::map<int, float2 *> tables;       // exp(-2 pi i t / T), t in [0, T)
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

void oip_fft_destroy(oip_ctx *ctx)
{
    if (!ctx->fft) return;
    for (auto &kv : ctx->fft->tables) hipFree(kv.second);
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
    OipFft2dPlan pl;
    pl.M = M; pl.N = N;
    // rows (x axis): leading factors in mode A (lanes = j, >= 8 contiguous points wanted),
    // last factor as whole contiguous sub-rows in mode B
    if (!split_axis(N, 256, 4096, &pl.xf)) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: cannot factor row length %d", N);
    if (!split_axis(M, 256, 256, &pl.yf)) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: cannot factor column length %d", M);
    // x passes
    {
        int T = N;                       // length of the sub-problem the pass starts from
        for (size_t i = 0; i < pl.xf.size(); ++i) {
            OipFftPass p;
            memset(&p, 0, sizeof p);
            p.F = pl.xf[i];
            fill_radix(&p);
            const int S = T / p.F;
            if (S == 1) {
                p.mode = 1;
                p.vshift = pick_vshift(p.F, 32);
                p.Vp = p.vshift ? (1 << p.vshift) + 1 : 1;
                p.lanes = (long)M * (N / p.F);          // contiguous F-point vectors
                p.tw_mode = 0; p.T = 0;
            } else {
                p.mode = 0;
                p.vshift = pick_vshift(p.F, 32);
                p.Vp = (1 << p.vshift) + 1;
                p.nstride = S;
                p.lanes = S;
                p.lane_tiles = (S + (1 << p.vshift) - 1) >> p.vshift;
                p.O1 = N / T; p.o1_stride = T;          // blocks of the current sub-problem
                p.O2 = M;     p.o2_stride = N;          // rows
                p.tw_mode = 1; p.T = T;
            }
            pl.passes.push_back(p);
            T = S;
        }
    }
    pl.n_x = (int)pl.passes.size();
    // y passes: always mode A with lanes = x
    {
        int T = M;
        for (size_t i = 0; i < pl.yf.size(); ++i) {
            OipFftPass p;
            memset(&p, 0, sizeof p);
            p.F = pl.yf[i];
            fill_radix(&p);
            const int S = T / p.F;
            p.mode = 0;
            p.vshift = pick_vshift(p.F, 32);
            p.Vp = (1 << p.vshift) + 1;
            p.nstride = (long)S * N;
            p.lanes = N;
            p.lane_tiles = (N + (1 << p.vshift) - 1) >> p.vshift;
            p.O1 = S;     p.o1_stride = N;              // j: row offset inside the block
            p.O2 = M / T; p.o2_stride = (long)T * N;    // blocks
            p.tw_mode = S > 1 ? 2 : 0; p.T = S > 1 ? T : 0;
            pl.passes.push_back(p);
            T = S;
        }
    }
    for (auto &p : pl.passes) {
        if ((long)p.F * p.Vp > kMaxTileElems) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft2d: factor %d too long for one LDS tile", p.F);
        const float2 *t;
        int rc = get_table(ctx, p.F, &t);
        if (rc) return rc;
        if (p.T) { rc = get_table(ctx, p.T, &t); if (rc) return rc; }
    }
    it = ctx->fft->plans.emplace(key, pl).first;
    *out = &it->second;
    return OIP_OK;
}

static int launch_pass(oip_ctx *ctx, float2 *data, OipFftPass p, int inverse)
{
    p.inverse = inverse;
    const float2 *twF = nullptr, *twT = nullptr;
    int rc = get_table(ctx, p.F, &twF);
    if (rc) return rc;
    if (p.T) { rc = get_table(ctx, p.T, &twT); if (rc) return rc; }
    long blocks;
    if (p.mode == 0) blocks = (long)p.lane_tiles * p.O1 * p.O2;
    else blocks = (p.lanes + (1 << p.vshift) - 1) >> p.vshift;
    if (blocks <= 0 || blocks > 0x7fffffffL) return oip_fail(ctx, OIP_E_UNSUPPORTED, "fft pass grid too large");
    size_t lds = sizeof(float2) * ((size_t)2 * p.F * p.Vp + p.F);
    OipProfScope prof(ctx, "fft_pass_kernel");
    hipLaunchKernelGGL(fft_pass_kernel, dim3((unsigned)blocks), dim3(kFftBlock), lds, ctx->stream, data, p, twF, twT);
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}

// In-place complex 2-D transform of an M x N row-major float2 array.
//   inverse == 0: natural order in, digit-scrambled spectrum out (rows first, then columns)
//   inverse == 1: scrambled spectrum in, natural order out, unnormalised (columns, then rows)
int oip_fft2d_exec(oip_ctx *ctx, const OipFft2dPlan *pl, float2 *data, int inverse)
{
    const int np = (int)pl->passes.size();
    if (!inverse) {
        for (int i = 0; i < np; ++i) {
            int rc = launch_pass(ctx, data, pl->passes[i], 0);
            if (rc) return rc;
        }
    } else {
        for (int i = np - 1; i >= 0; --i) {
            int rc = launch_pass(ctx, data, pl->passes[i], 1);
            if (rc) return rc;
        }
    }
    return OIP_OK;
}
