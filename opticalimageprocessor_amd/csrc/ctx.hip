// ctx.hip -- context, stream, memory, profiling for liboipgpu.so
#include "oip_internal.h"

#include <cstdlib>

int oip_fail(oip_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
        va_end(ap);
    }
    return code;
}

extern "C" int oip_version(void) { return 0x0101; }

extern "C" int oip_create(int device, oip_ctx **out)
{
    if (!out) return OIP_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        fprintf(stderr, "oip_create: no usable HIP device %d (found %d); liboipgpu has no CPU fallback\n",
                device, n);
        return OIP_E_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return OIP_E_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return OIP_E_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "oip_create: device %d is %s; liboipgpu is built for gfx950 only\n", device,
                prop.gcnArchName);
        return OIP_E_DEVICE;
    }
    oip_ctx *ctx = new oip_ctx();
    ctx->device = device;
    ctx->cu_count = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return OIP_E_DEVICE;
    }
    float tab[32 * 4];
    for (int i = 0; i < 32; ++i) oip_interpolate_cubic_host(i * (1.f / 32), tab + i * 4);
    if (hipMalloc(&ctx->d_tab1d, sizeof tab) != hipSuccess ||
        hipMemcpy(ctx->d_tab1d, tab, sizeof tab, hipMemcpyHostToDevice) != hipSuccess ||
        hipMalloc(&ctx->d_small, 65536) != hipSuccess ||
        hipHostMalloc(&ctx->h_small, 65536, hipHostMallocDefault) != hipSuccess) {
        oip_destroy(ctx);
        return OIP_E_DEVICE;
    }
    ctx->small_bytes = 65536;
    *out = ctx;
    return OIP_OK;
}

int oip_small(oip_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->small_bytes) return OIP_OK;
    OIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    size_t want = (bytes + 65535) / 65536 * 65536;
    void *d = nullptr, *h = nullptr;
    if (hipMalloc(&d, want) != hipSuccess || hipHostMalloc(&h, want, hipHostMallocDefault) != hipSuccess) {
        if (d) hipFree(d);
        return oip_fail(ctx, OIP_E_NOMEM, "result scratch of %zu bytes failed", want);
    }
    hipFree(ctx->d_small);
    hipHostFree(ctx->h_small);
    ctx->d_small = d;
    ctx->h_small = h;
    ctx->small_bytes = want;
    return OIP_OK;
}

extern "C" void oip_destroy(oip_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    oip_fft_destroy(ctx);
    for (auto &t : ctx->resize_tabs) { hipFree(t.d_xofs); hipFree(t.d_alpha); hipFree(t.d_yofs); hipFree(t.d_beta); if (t.d_xspec) hipFree(t.d_xspec); if (t.d_yspec) hipFree(t.d_yspec); }
    for (auto &p : ctx->prof_pending) { if (p.own_e0) hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
    if (ctx->d_tab1d) hipFree(ctx->d_tab1d);
    if (ctx->d_small) hipFree(ctx->d_small);
    if (ctx->h_small) hipHostFree(ctx->h_small);
    if (ctx->d_work) hipFree(ctx->d_work);
    oip_stage_destroy(ctx);
    for (int i = 0; i < 2; ++i)
        if (ctx->d_stage[i]) hipFree(ctx->d_stage[i]);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *oip_last_error(const oip_ctx *ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int oip_set_stream(oip_ctx *ctx, void *s)
{
    OIP_CHECK_CTX(ctx);
    OIP_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->stream) OIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (s == nullptr) {
        if (!ctx->own_stream) {
            OIP_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
            ctx->own_stream = true;
        }
        return OIP_OK;
    }
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)s;
    ctx->own_stream = false;
    return OIP_OK;
}

extern "C" void *oip_get_stream(oip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int oip_sync(oip_ctx *ctx)
{
    OIP_CHECK_CTX(ctx);
    OIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OIP_OK;
}

extern "C" int oip_malloc(oip_ctx *ctx, void **d_ptr, size_t bytes)
{
    OIP_CHECK_CTX(ctx);
    if (!d_ptr) return oip_fail(ctx, OIP_E_INVALID, "oip_malloc: null out pointer");
    OIP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) return oip_fail(ctx, OIP_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return OIP_OK;
}

extern "C" int oip_free(oip_ctx *ctx, void *d_ptr)
{
    OIP_CHECK_CTX(ctx);
    if (d_ptr) OIP_HIP(ctx, hipFree(d_ptr));
    return OIP_OK;
}

extern "C" int oip_memset(oip_ctx *ctx, void *d_ptr, int value, size_t bytes)
{
    OIP_CHECK_CTX(ctx);
    ctx->prof_chain = nullptr;
    OIP_HIP(ctx, hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return OIP_OK;
}

extern "C" int oip_memcpy_h2d(oip_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    OIP_CHECK_CTX(ctx);
    OIP_HIP(ctx, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OIP_OK;
}

extern "C" int oip_memcpy_d2h(oip_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    OIP_CHECK_CTX(ctx);
    OIP_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return OIP_OK;
}

extern "C" int oip_host_alloc(oip_ctx *ctx, void **ptr, size_t bytes)
{
    OIP_CHECK_CTX(ctx);
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return oip_fail(ctx, OIP_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return OIP_OK;
}

extern "C" int oip_host_free(oip_ctx *ctx, void *ptr)
{
    OIP_CHECK_CTX(ctx);
    if (ptr) OIP_HIP(ctx, hipHostFree(ptr));
    return OIP_OK;
}

int oip_workspace(oip_ctx *ctx, size_t bytes, void **out)
{
    if (bytes > ctx->work_bytes) {
        // grow-only; callers on this context's stream are serialised, so syncing before
        // the free is enough to retire earlier users of the old block
        OIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_work) OIP_HIP(ctx, hipFree(ctx->d_work));
        ctx->d_work = nullptr;
        ctx->work_bytes = 0;
        size_t want = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(&ctx->d_work, want);
        if (e != hipSuccess) return oip_fail(ctx, OIP_E_NOMEM, "workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        ctx->work_bytes = want;
    }
    *out = ctx->d_work;
    return OIP_OK;
}

// ---- profiling -----------------------------------------------------------------------
int oip_prof_begin(oip_ctx *ctx, const char *name)
{
    if (!ctx || !ctx->prof_on) return -1;
    if (!ctx->prof_filter.empty() && ctx->prof_filter != name) {
        ctx->prof_chain = nullptr;          // an untimed kernel follows: the next timed one needs its own start event
        return -1;
    }
    int entry = -1;
    for (size_t i = 0; i < ctx->prof.size(); ++i)
        if (ctx->prof[i].name == name) { entry = (int)i; break; }
    if (entry < 0) {
        ctx->prof.push_back(oip_prof_entry());
        ctx->prof.back().name = name;
        entry = (int)ctx->prof.size() - 1;
    }
    // Consecutive kernels of one API call run back to back on the stream, so the end event of one is
    // the start event of the next: one hipEventRecord per kernel instead of two (each record is a
    // few microseconds of stream time -- 5 % of a correlation batch with two per kernel).
    oip_prof_pending p;
    p.entry = entry;
    p.own_e0 = ctx->prof_chain == nullptr;
    if (p.own_e0) {
        if (hipEventCreate(&p.e0) != hipSuccess) return -1;
        hipEventRecord(p.e0, ctx->stream);
    } else {
        p.e0 = ctx->prof_chain;
    }
    if (hipEventCreate(&p.e1) != hipSuccess) return -1;
    ctx->prof_chain = nullptr;
    ctx->prof_pending.push_back(p);
    return (int)ctx->prof_pending.size() - 1;
}

void oip_prof_end(oip_ctx *ctx, int pending)
{
    if (pending < 0 || !ctx) return;
    hipEventRecord(ctx->prof_pending[pending].e1, ctx->stream);
    ctx->prof_chain = ctx->prof_pending[pending].e1;
}

static void prof_resolve(oip_ctx *ctx)
{
    if (ctx->prof_pending.empty()) return;
    hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            ctx->prof[p.entry].total_ms += ms;
            ctx->prof[p.entry].launches += 1;
        }
    }
    for (auto &p : ctx->prof_pending) {
        if (p.own_e0) hipEventDestroy(p.e0);
        hipEventDestroy(p.e1);
    }
    ctx->prof_pending.clear();
    ctx->prof_chain = nullptr;
}

extern "C" int oip_profile_enable(oip_ctx *ctx, int on)
{
    OIP_CHECK_CTX(ctx);
    prof_resolve(ctx);
    ctx->prof_on = on != 0;
    return OIP_OK;
}

extern "C" int oip_profile_filter(oip_ctx *ctx, const char *kernel_name)
{
    OIP_CHECK_CTX(ctx);
    ctx->prof_filter = kernel_name ? kernel_name : "";
    return OIP_OK;
}

extern "C" int oip_profile_reset(oip_ctx *ctx)
{
    OIP_CHECK_CTX(ctx);
    prof_resolve(ctx);
    ctx->prof.clear();
    return OIP_OK;
}

extern "C" int oip_profile_count(oip_ctx *ctx)
{
    if (!ctx) return 0;
    prof_resolve(ctx);
    return (int)ctx->prof.size();
}

extern "C" int oip_profile_get(oip_ctx *ctx, int i, char *name, int namelen, double *total_ms, long *launches)
{
    OIP_CHECK_CTX(ctx);
    prof_resolve(ctx);
    if (i < 0 || i >= (int)ctx->prof.size()) return oip_fail(ctx, OIP_E_INVALID, "profile index out of range");
    if (name && namelen > 0) snprintf(name, namelen, "%s", ctx->prof[i].name.c_str());
    if (total_ms) *total_ms = ctx->prof[i].total_ms;
    if (launches) *launches = ctx->prof[i].launches;
    return OIP_OK;
}
