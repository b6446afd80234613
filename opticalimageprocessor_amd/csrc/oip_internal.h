// oip_internal.h -- shared by the translation units of liboipgpu.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "oip_c.h"

struct oip_prof_entry {
    std::string name;
    double total_ms = 0.0;
    long launches = 0;
};
struct oip_prof_pending {
    int entry;
    hipEvent_t e0, e1;
    bool own_e0;        // false: e0 is the previous scope's e1 (back-to-back kernels share the event)
};

struct oip_fft_state;   // fft.hip
struct oip_stage_state; // staging.hip

// cv::resize coefficient tables (resize.cpp builds the same xofs/alpha/yofs/beta on the host)
struct OipResizeTab {
    int sw, sh, dw, dh;
    bool x4;            // exact x4 up-sampling: the 4x4-per-lane kernel applies
    int x4h;            // xofs[dx] == (dx - 2) >> 2 for every dx: the FFT loader needs no xofs look-up
    int x4v;            // yofs[dy] == (dy - 2) >> 2 for every dy: the sliding-window vertical kernel applies
    int *d_xofs;
    float *d_alpha;     // dw x 4
    int *d_yofs;
    float *d_beta;      // dh x 4
    void *d_xspec;      // float2 [5][dw]: the horizontal up-sampling as an operator on spectra (built on first use)
    int xspec_state;    // 0 not tried, 1 built, -1 the geometry has no such form
    void *d_yspec;      // float2 [5][dh]: the same for the vertical axis
    int yspec_state;
};

// OpenCV imgwarp.cpp interpolateCubic, f32, evaluated on the host exactly as OpenCV does
// (x86-64, no contraction: every TU is built with -ffp-contract=off)
static inline void oip_interpolate_cubic_host(float x, float *coeffs)
{
    const float A = -0.75f;
    coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

struct oip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    char err[1024] = {0};
    int cu_count = 256;

    // bicubic 1-D coefficient table (32 phases x 4 taps, f32) in HBM
    float *d_tab1d = nullptr;

    // scratch for small host<->device results
    void *d_small = nullptr;       // device scratch for result tables (64 KiB, grown on demand: oip_small)
    void *h_small = nullptr;       // pinned host mirror of it
    size_t small_bytes = 0;

    // growable device workspace (FFT planes, windows)
    void *d_work = nullptr;
    size_t work_bytes = 0;

    // raster I/O staging (staging.hip): pinned ring + staging streams, and the two device blocks of oip_rrc_u16_host
    oip_stage_state *stage = nullptr;
    void *d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;

    oip_fft_state *fft = nullptr;
    std::vector<OipResizeTab> resize_tabs;

    // profiling
    bool prof_on = false;
    std::string prof_filter;            // non-empty: only this kernel name is timed
    std::vector<oip_prof_entry> prof;
    std::vector<oip_prof_pending> prof_pending;
    hipEvent_t prof_chain = nullptr;    // end event of the previous scope while nothing else has been enqueued since
    const char *prof_tag = nullptr;     // appended to the FFT pass names (the quarter-width band transforms report apart)
};

int oip_fail(oip_ctx *ctx, int code, const char *fmt, ...);
int oip_prof_begin(oip_ctx *ctx, const char *name);   // returns pending index or -1
void oip_prof_end(oip_ctx *ctx, int pending);
int oip_workspace(oip_ctx *ctx, size_t bytes, void **out);   // grow-only workspace
int oip_small(oip_ctx *ctx, size_t bytes);                   // make d_small / h_small hold at least `bytes`
void oip_fft_destroy(oip_ctx *ctx);
void oip_stage_destroy(oip_ctx *ctx);
// rrc.hip: the RRC launch on an explicit stream (no profiler scope; safe from a staging thread)
int oip_rrc_launch(oip_ctx *ctx, hipStream_t stream, const uint16_t *d_src, uint16_t *d_dst, int w, long h, const double *d_kb);

#define OIP_HIP(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess)                                                          \
            return oip_fail((ctx), OIP_E_DEVICE, "%s failed: %s (%s:%d)", #call,        \
                            hipGetErrorString(e__), __FILE__, __LINE__);                \
    } while (0)

#define OIP_CHECK_CTX(ctx)                                                              \
    do {                                                                                \
        if (!(ctx)) return OIP_E_INVALID;                                               \
        (ctx)->prof_chain = nullptr; /* the caller may have put its own work on the stream */ \
    } while (0)

// scoped profiling region around one or more launches
struct OipProfScope {
    oip_ctx *ctx;
    int id;
    OipProfScope(oip_ctx *c, const char *name) : ctx(c), id(oip_prof_begin(c, name)) {}
    ~OipProfScope() { oip_prof_end(ctx, id); }
};

// ---- device helpers shared by the resampling kernels -----------------------------------
// A line index that is the same in every lane, told to the compiler: the per-line row tables are then read with scalar loads
// (s_load, counted by lgkmcnt) instead of one vector load per lane of the same 16-24 bytes -- whose s_waitcnt vmcnt(0) also
// waits for every line the kernel has requested ahead (round 4: the row table of remap_shift8_rrc_kernel and
// align_mss8_kernel came in through global_load_dwordx3/x4 because the compiler could not prove the loop counter uniform).
__device__ __forceinline__ long oip_uniform(long r) { return (long)__builtin_amdgcn_readfirstlane((int)r); }
// One int of a table at a wave-uniform index through the scalar cache, whatever the compiler can prove about aliasing: where
// a kernel also stores through pointers it was handed in a struct, the compiler reads such a table with a vector load, and the
// s_waitcnt vmcnt(0) behind it waits for every store and prefetch the wave has in flight.  The table must not be written by the
// kernel (the scalar cache is not coherent with vector stores).  Waits for the value (lgkmcnt: LDS and scalar loads only).
__device__ __forceinline__ int oip_sload_i32(const int *__restrict__ table, long index)
{
    int v;
    const int *p = table + oip_uniform(index);
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// OpenCV cvRound(float): round half to even; v_rndne_f32 + v_cvt_i32_f32
__device__ __forceinline__ int oip_cvround(float v) { return (int)__builtin_rintf(v); }

__device__ __forceinline__ int oip_sat_short(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

// saturate_cast<ushort>(float) = clamp(cvRound(v), 0, 65535)
__device__ __forceinline__ unsigned oip_sat_u16(float v)
{
    int iv = oip_cvround(v);
    iv = iv < 0 ? 0 : (iv > 65535 ? 65535 : iv);
    return (unsigned)iv;
}

// IMO::InplaceRRC's pixel (imageop.h:134): (uint16_t)(k * s + b) in fp64, two roundings, then what x86-64 compilers emit for
// the double -> uint16_t cast: cvttsd2si (32-bit, truncating; 0x80000000 when out of range or NaN) followed by a 16-bit
// truncation.  Both bounds explicit: (int)v is only evaluated where the C++ conversion is defined.  (rrc.hip; also the
// RRC-on-load form of the resampling kernel, remap.hip)
__device__ __forceinline__ unsigned oip_rrc_px(double k, double b, unsigned s)
{
    double v = __dadd_rn(__dmul_rn(k, (double)s), b);
    int t = (v > -2147483649.0 && v < 2147483648.0) ? (int)v : 0;
    return (unsigned)t & 0xffffu;
}
