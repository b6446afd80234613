// staging.hip -- raster I/O staging between files / pageable host buffers and HBM.
//
// Replaces the staging half of ImageOperations: ReadFileContent (imageop.h:52-82: one heap buffer
// filled by 8 MiB fread calls), LoadRawImage (:110-127), WriteBufferToFile (:84-97) and the per-section
// fseek/fread/fwrite of Stitcher::PreStitch (stitcher.h:103-120).  The reference moves every raster
// through one pageable heap buffer, serially with the arithmetic; here a raster travels in blocks
// through a ring of pinned buffers on a stream of its own:
//     file --fread--> pinned slot --DMA--> HBM        (the next block is being read while this one flies)
//     HBM --DMA--> pinned slot --fwrite--> file
//     pageable buffer --pool memcpy--> pinned slot --DMA--> HBM   (and back)
// so disk, host copies, PCIe and the kernels of the context's compute stream overlap.  Staging calls
// use only staging streams and pinned slots of their own and never change the context's compute stream or
// profiler: they may run on other host threads while the first one drives kernels through the same
// context.  Three lanes, each serialised by its own mutex (a second caller of the same lane waits):
//   ring lane      oip_read_file_to_device, oip_write_device_to_file, oip_upload_staged, oip_rrc_u16_host
//                  (four pinned slots; stream `stream`, oip_rrc_u16_host also `rrc_stream`)
//   download lane  oip_download_staged (two pinned slots, stream `stream2`)
// so an upload thread and a download thread run full duplex.  The state is created once under a lock.
// A call returns a TICKET; oip_stage_wait(ctx, ticket) makes the compute stream wait -- on the device, not the
// host -- for the transfers up to that ticket, and oip_stage_sync(ctx) blocks the host until they are done.
// Ordering against the compute stream: downloads and file writes start after the compute-stream work
// enqueued before the call.  UPLOADS DO NOT WAIT for the compute stream (an uploader thread must keep the
// link busy while a 60-ms correlation is queued): re-uploading into a buffer that queued kernels still read
// is the caller's hazard -- call oip_stage_order_after_compute(ctx) first, or upload into another buffer.
#include "oip_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace {

// ---- a small persistent pool for the pageable <-> pinned copies ---------------------------------
// (one thread moves ~10 GB/s, the link ~55: the copies, not the link, bound a pageable host buffer)
class CopyPool {
public:
    static CopyPool &get()
    {
        static CopyPool p;
        return p;
    }
    int threads() const { return (int)mWorkers.size() + 1; }
    void copy(void *dst, const void *src, size_t bytes)
    {
        const int n = threads();
        if (n == 1 || bytes < ((size_t)4 << 20)) { memcpy(dst, src, bytes); return; }
        std::lock_guard<std::mutex> one_at_a_time(mCall);          // an upload and a download thread may both be here
        const size_t part = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
        {
            std::unique_lock<std::mutex> lk(mMu);
            mDst = (char *)dst; mSrc = (const char *)src; mBytes = bytes; mPart = part; mWidth = 0;
            mPending = (int)mWorkers.size();
            ++mGen;
        }
        mCv.notify_all();
        memcpy(dst, src, part < bytes ? part : bytes);                 // the caller takes piece 0
        std::unique_lock<std::mutex> lk(mMu);
        mDone.wait(lk, [&] { return mPending == 0; });
    }
    // rows of `width` bytes, the pitches in bytes: the rows are dealt over the threads in contiguous runs
    void copy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows)
    {
        const int n = threads();
        auto run = [&](size_t r0, size_t r1) {
            for (size_t r = r0; r < r1; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
        };
        if (n == 1 || width * rows < ((size_t)4 << 20)) { run(0, rows); return; }
        std::lock_guard<std::mutex> one_at_a_time(mCall);
        const size_t part = (rows + n - 1) / n;
        {
            std::unique_lock<std::mutex> lk(mMu);
            mDst = (char *)dst; mSrc = (const char *)src; mBytes = rows; mPart = part; mWidth = width; mDPitch = dpitch; mSPitch = spitch;
            mPending = (int)mWorkers.size();
            ++mGen;
        }
        mCv.notify_all();
        run(0, part < rows ? part : rows);
        std::unique_lock<std::mutex> lk(mMu);
        mDone.wait(lk, [&] { return mPending == 0; });
    }

private:
    CopyPool()
    {
        const char *e = getenv("OIP_HOST_COPY_THREADS");
        int n = e ? atoi(e) : (int)std::thread::hardware_concurrency() / 2;
        n = n < 1 ? 1 : (n > 16 ? 16 : n);
        for (int i = 1; i < n; ++i) mWorkers.emplace_back([this, i] { work(i); });
    }
    ~CopyPool()
    {
        {
            std::unique_lock<std::mutex> lk(mMu);
            mStop = true;
            ++mGen;
        }
        mCv.notify_all();
        for (auto &t : mWorkers) t.join();
    }
    void work(int idx)
    {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mMu);
            mCv.wait(lk, [&] { return mGen != seen; });
            seen = mGen;
            if (mStop) return;
            char *d = mDst; const char *s = mSrc; const size_t bytes = mBytes, part = mPart, width = mWidth, dp = mDPitch, sp = mSPitch;
            lk.unlock();
            const size_t off = (size_t)idx * part;
            if (width == 0) {
                if (off < bytes) memcpy(d + off, s + off, bytes - off < part ? bytes - off : part);
            } else {                                                    // 2-D: bytes = rows, part = rows per thread
                const size_t r1 = off + part < bytes ? off + part : bytes;
                for (size_t r = off; r < r1; ++r) memcpy(d + r * dp, s + r * sp, width);
            }
            lk.lock();
            if (--mPending == 0) mDone.notify_one();
        }
    }
    std::vector<std::thread> mWorkers;
    std::mutex mMu, mCall;
    std::condition_variable mCv, mDone;
    unsigned long mGen = 0;
    int mPending = 0;
    bool mStop = false;
    char *mDst = nullptr;
    const char *mSrc = nullptr;
    size_t mBytes = 0, mPart = 0, mWidth = 0, mDPitch = 0, mSPitch = 0;
};

constexpr int kSlots = 4;
constexpr size_t kSlotBytes = (size_t)32 << 20;
constexpr int kTicketRing = 64;

}  // namespace

// the staging state lives behind the context (opaque to the other translation units)
struct oip_stage_state {
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // the download lane's stream
    hipStream_t rrc_stream = nullptr;       // second stream of oip_rrc_u16_host (up and down transfers of neighbouring blocks overlap)
    std::mutex ring_mu, down_mu;            // one caller per lane at a time
    void *slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t slot_free[kSlots] = {nullptr, nullptr, nullptr, nullptr};   // recorded after the DMA that last used the slot
    bool slot_used[kSlots] = {false, false, false, false};
    int next = 0;
    hipEvent_t ticket_ev[kTicketRing];
    hipEvent_t compute_ev = nullptr;        // marks the compute stream's position for uploads
    // the download lane: its own two slots, stream2 and event, so that oip_download_staged on one host thread and an
    // upload on another never touch the same state (full duplex over the link)
    void *dslot[2] = {nullptr, nullptr};
    hipEvent_t dslot_free[2] = {nullptr, nullptr};
    bool dslot_used[2] = {false, false};
    int dnext = 0;
    hipEvent_t down_compute_ev = nullptr;
    std::atomic<long> ticket{0};
    // where the host side of the ring lane spends its time (oip_stage_stats): pageable <-> pinned copies on the pool, and
    // waiting for a slot whose DMA has not finished (the link is the limit then)
    std::atomic<long> copy_ns{0}, wait_ns{0}, ring_bytes{0}, ring_calls{0};
    // LUT cache of oip_rrc_u16_host
    double *d_kb = nullptr;
    std::vector<double> kb_host;
};

static std::mutex g_stage_init_mu;
static int stage_init(oip_ctx *ctx)
{
    // first staging calls may come from two threads at once: the state is created exactly once
    std::lock_guard<std::mutex> once(g_stage_init_mu);
    if (ctx->stage) return OIP_OK;
    OIP_HIP(ctx, hipSetDevice(ctx->device));
    oip_stage_state *s = new oip_stage_state();
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->rrc_stream, hipStreamNonBlocking) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging stream"); }
    for (int i = 0; i < kSlots; ++i) {
        if (hipHostMalloc(&s->slot[i], kSlotBytes, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s->slot_free[i], hipEventDisableTiming) != hipSuccess) {
            delete s;
            return oip_fail(ctx, OIP_E_NOMEM, "pinned staging ring (%d x %zu MiB) failed", kSlots, kSlotBytes >> 20);
        }
    }
    for (int i = 0; i < kTicketRing; ++i)
        if (hipEventCreateWithFlags(&s->ticket_ev[i], hipEventDisableTiming) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging events"); }
    if (hipEventCreateWithFlags(&s->compute_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->down_compute_ev, hipEventDisableTiming) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging events"); }
    for (int i = 0; i < 2; ++i)
        if (hipHostMalloc(&s->dslot[i], kSlotBytes, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&s->dslot_free[i], hipEventDisableTiming) != hipSuccess) {
            delete s;
            return oip_fail(ctx, OIP_E_NOMEM, "pinned download slots failed");
        }
    ctx->stage = s;
    return OIP_OK;
}

void oip_stage_destroy(oip_ctx *ctx)
{
    oip_stage_state *s = ctx->stage;
    if (!s) return;
    if (s->stream) hipStreamSynchronize(s->stream);
    if (s->stream2) { hipStreamSynchronize(s->stream2); hipStreamDestroy(s->stream2); }
    if (s->rrc_stream) { hipStreamSynchronize(s->rrc_stream); hipStreamDestroy(s->rrc_stream); }
    for (int i = 0; i < kSlots; ++i) { if (s->slot[i]) hipHostFree(s->slot[i]); if (s->slot_free[i]) hipEventDestroy(s->slot_free[i]); }
    for (int i = 0; i < kTicketRing; ++i) hipEventDestroy(s->ticket_ev[i]);
    if (s->compute_ev) hipEventDestroy(s->compute_ev);
    if (s->down_compute_ev) hipEventDestroy(s->down_compute_ev);
    for (int i = 0; i < 2; ++i) { if (s->dslot[i]) hipHostFree(s->dslot[i]); if (s->dslot_free[i]) hipEventDestroy(s->dslot_free[i]); }
    if (s->d_kb) hipFree(s->d_kb);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
    ctx->stage = nullptr;
}

// next ring slot, once the DMA that last read or wrote it has finished
static int slot_acquire(oip_ctx *ctx, oip_stage_state *s, int *out)
{
    const int i = s->next;
    s->next = (s->next + 1) % kSlots;
    if (s->slot_used[i]) {
        const auto t0 = std::chrono::steady_clock::now();
        OIP_HIP(ctx, hipEventSynchronize(s->slot_free[i]));
        s->wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    *out = i;
    return OIP_OK;
}

// pool copy with its time booked to the lane's statistics
static void timed_copy(oip_stage_state *s, void *dst, const void *src, size_t bytes)
{
    const auto t0 = std::chrono::steady_clock::now();
    CopyPool::get().copy(dst, src, bytes);
    s->copy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    s->ring_bytes += (long)bytes;
}

extern "C" int oip_stage_stats(oip_ctx *ctx, double *out, int reset)
{
    if (!ctx || !out) return OIP_E_INVALID;
    oip_stage_state *s = ctx->stage;
    out[0] = s ? s->copy_ns.load() * 1e-9 : 0.0;
    out[1] = s ? s->wait_ns.load() * 1e-9 : 0.0;
    out[2] = s ? (double)s->ring_bytes.load() : 0.0;
    out[3] = s ? (double)s->ring_calls.load() : 0.0;
    if (s && reset) { s->copy_ns = 0; s->wait_ns = 0; s->ring_bytes = 0; s->ring_calls = 0; }
    return OIP_OK;
}

static long ticket_issue(oip_ctx *ctx, oip_stage_state *s)
{
    const long t = s->ticket.load() + 1;
    hipEventRecord(s->ticket_ev[t % kTicketRing], s->stream);
    s->ticket.store(t);
    return t;
}

// the ring lane's stream waits for the compute stream's work enqueued so far: file writes read what kernels
// produced; uploads only on request (oip_stage_order_after_compute -- see the note at the top of the file)
static int order_after_compute(oip_ctx *ctx, oip_stage_state *s)
{
    hipEvent_t e = s->compute_ev;
    OIP_HIP(ctx, hipEventRecord(e, ctx->stream));
    OIP_HIP(ctx, hipStreamWaitEvent(s->stream, e, 0));
    return OIP_OK;
}

extern "C" int oip_stage_order_after_compute(oip_ctx *ctx)
{
    if (!ctx) return OIP_E_INVALID;
    int rc = stage_init(ctx);
    if (rc) return rc;
    std::lock_guard<std::mutex> lane(ctx->stage->ring_mu);
    return order_after_compute(ctx, ctx->stage);
}

extern "C" int oip_stage_wait(oip_ctx *ctx, long ticket)
{
    if (!ctx) return OIP_E_INVALID;
    oip_stage_state *s = ctx->stage;
    if (!s || ticket <= 0) return OIP_OK;
    const long now = s->ticket.load();
    if (ticket > now) return oip_fail(ctx, OIP_E_INVALID, "oip_stage_wait: ticket %ld not issued yet", ticket);
    // an old ticket whose event was recycled is implied by any later one on the same stream
    const long use = now - ticket >= kTicketRing - 2 ? now : ticket;
    OIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ticket_ev[use % kTicketRing], 0));
    return OIP_OK;
}

extern "C" int oip_stage_sync(oip_ctx *ctx)
{
    if (!ctx) return OIP_E_INVALID;
    if (ctx->stage) OIP_HIP(ctx, hipStreamSynchronize(ctx->stage->stream));
    return OIP_OK;
}

extern "C" int oip_stage_threads(void) { return CopyPool::get().threads(); }

// ---- file <-> device -------------------------------------------------------------------------------
extern "C" int oip_read_file_to_device(oip_ctx *ctx, const char *path, size_t offset, size_t bytes, void *d_dst,
                                       size_t *bytes_read, long *ticket)
{
    if (!ctx) return OIP_E_INVALID;
    if (!path || (!d_dst && bytes)) return oip_fail(ctx, OIP_E_INVALID, "oip_read_file_to_device: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    std::lock_guard<std::mutex> lane(s->ring_mu);
    FILE *f = fopen(path, "rb");
    if (!f) return oip_fail(ctx, OIP_E_INVALID, "cannot open file [%s]: %d", path, errno);      // imageop.h:55-57
    if (bytes == 0) {                                                                            // all available (imageop.h:59-63)
        if (fseeko(f, 0, SEEK_END)) { fclose(f); return oip_fail(ctx, OIP_E_INVALID, "ReadFileContent(): seek2end failed"); }
        const off_t end = ftello(f);
        bytes = (size_t)end > offset ? (size_t)end - offset : 0;
    }
    if (fseeko(f, (off_t)offset, SEEK_SET)) { fclose(f); return oip_fail(ctx, OIP_E_INVALID, "ReadFileContent(): rewind failed"); }
    size_t done = 0;
    while (done < bytes) {
        int i;
        if ((rc = slot_acquire(ctx, s, &i))) { fclose(f); return rc; }
        const size_t want = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        size_t got = 0;
        while (got < want) {                                                                     // imageop.h:69-79, 8 MiB units
            const size_t unit = want - got < ((size_t)8 << 20) ? want - got : ((size_t)8 << 20);
            const size_t rn = fread((char *)s->slot[i] + got, 1, unit, f);
            got += rn;
            if (rn == 0) break;
        }
        if (got) {
            if (hipMemcpyAsync((char *)d_dst + done, s->slot[i], got, hipMemcpyHostToDevice, s->stream) != hipSuccess) {
                fclose(f);
                return oip_fail(ctx, OIP_E_DEVICE, "H2D of a staged block failed");
            }
            hipEventRecord(s->slot_free[i], s->stream);
            s->slot_used[i] = true;
        }
        done += got;
        if (got < want) break;                                                                   // short file
    }
    fclose(f);
    if (bytes_read) *bytes_read = done;
    const long t = ticket_issue(ctx, s);
    if (ticket) *ticket = t;
    else OIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ticket_ev[t % kTicketRing], 0));        // no ticket wanted: order the compute stream now
    return OIP_OK;
}

extern "C" int oip_write_device_to_file(oip_ctx *ctx, const void *d_src, size_t bytes, const char *path, int append)
{
    if (!ctx) return OIP_E_INVALID;
    if (!path || (!d_src && bytes)) return oip_fail(ctx, OIP_E_INVALID, "oip_write_device_to_file: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    std::lock_guard<std::mutex> lane(s->ring_mu);
    FILE *f = fopen(path, append ? "ab" : "wb");
    if (!f) return oip_fail(ctx, OIP_E_RUNTIME, "open file [%s] failed: %d", path, errno);      // imageop.h:86-88
    if ((rc = order_after_compute(ctx, s))) { fclose(f); return rc; }
    // block k's DMA is in flight while block k-1 is written
    int prev = -1;
    size_t prev_bytes = 0, done = 0;
    auto flush = [&](int i, size_t n) -> int {
        if (hipEventSynchronize(s->slot_free[i]) != hipSuccess) return oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
        size_t w = 0;
        while (w < n) {                                                                          // imageop.h:88-95, 8 MiB units
            const size_t unit = n - w < ((size_t)8 << 20) ? n - w : ((size_t)8 << 20);
            const size_t wb = fwrite((const char *)s->slot[i] + w, 1, unit, f);
            if (wb == 0) return oip_fail(ctx, OIP_E_RUNTIME, "write file failed: %d", errno);
            w += wb;
        }
        return OIP_OK;
    };
    while (done < bytes) {
        int i;
        if ((rc = slot_acquire(ctx, s, &i))) break;
        const size_t n = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        if (hipMemcpyAsync(s->slot[i], (const char *)d_src + done, n, hipMemcpyDeviceToHost, s->stream) != hipSuccess) {
            rc = oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
            break;
        }
        hipEventRecord(s->slot_free[i], s->stream);
        s->slot_used[i] = true;
        if (prev >= 0 && (rc = flush(prev, prev_bytes))) break;
        prev = i; prev_bytes = n;
        done += n;
    }
    if (rc == OIP_OK && prev >= 0) rc = flush(prev, prev_bytes);
    if (fclose(f) != 0 && rc == OIP_OK) rc = oip_fail(ctx, OIP_E_RUNTIME, "close file [%s] failed: %d", path, errno);
    return rc;
}

// ---- pageable host buffer <-> device ---------------------------------------------------------------
extern "C" int oip_upload_staged(oip_ctx *ctx, void *d_dst, const void *host, size_t bytes, long *ticket)
{
    if (!ctx) return OIP_E_INVALID;
    if ((!d_dst || !host) && bytes) return oip_fail(ctx, OIP_E_INVALID, "oip_upload_staged: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    std::lock_guard<std::mutex> lane(s->ring_mu);
    ++s->ring_calls;
    size_t done = 0;
    while (done < bytes) {
        int i;
        if ((rc = slot_acquire(ctx, s, &i))) return rc;
        const size_t n = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        timed_copy(s, s->slot[i], (const char *)host + done, n);
        OIP_HIP(ctx, hipMemcpyAsync((char *)d_dst + done, s->slot[i], n, hipMemcpyHostToDevice, s->stream));
        hipEventRecord(s->slot_free[i], s->stream);
        s->slot_used[i] = true;
        done += n;
    }
    const long t = ticket_issue(ctx, s);
    if (ticket) *ticket = t;
    else OIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ticket_ev[t % kTicketRing], 0));
    return OIP_OK;
}

// The same for a 2-D block: `rows` rows of `width` bytes, host rows src_pitch bytes apart, device rows dst_pitch bytes apart
// (a column block of a raster: the units of a correlation section are column slices, so a section can arrive slice group by
// slice group and the first group's correlation runs under the upload of the second).
extern "C" int oip_upload_staged_2d(oip_ctx *ctx, void *d_dst, size_t dst_pitch, const void *host, size_t src_pitch, size_t width, size_t rows,
                                    long *ticket)
{
    if (!ctx) return OIP_E_INVALID;
    if (((!d_dst || !host) && width && rows) || width > dst_pitch || width > src_pitch || width > kSlotBytes)
        return oip_fail(ctx, OIP_E_INVALID, "oip_upload_staged_2d: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    std::lock_guard<std::mutex> lane(s->ring_mu);
    ++s->ring_calls;
    size_t done = 0;
    const size_t per = width ? kSlotBytes / width : rows;
    while (done < rows && width) {
        int i;
        if ((rc = slot_acquire(ctx, s, &i))) return rc;
        const size_t n = rows - done < per ? rows - done : per;
        {
            const auto t0 = std::chrono::steady_clock::now();
            CopyPool::get().copy2d(s->slot[i], width, (const char *)host + done * src_pitch, src_pitch, width, n);
            s->copy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            s->ring_bytes += (long)(width * n);
        }
        OIP_HIP(ctx, hipMemcpy2DAsync((char *)d_dst + done * dst_pitch, dst_pitch, s->slot[i], width, width, n, hipMemcpyHostToDevice, s->stream));
        hipEventRecord(s->slot_free[i], s->stream);
        s->slot_used[i] = true;
        done += n;
    }
    const long t = ticket_issue(ctx, s);
    if (ticket) *ticket = t;
    else OIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ticket_ev[t % kTicketRing], 0));
    return OIP_OK;
}

extern "C" int oip_download_staged(oip_ctx *ctx, void *host, const void *d_src, size_t bytes)
{
    if (!ctx) return OIP_E_INVALID;
    if ((!d_src || !host) && bytes) return oip_fail(ctx, OIP_E_INVALID, "oip_download_staged: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    std::lock_guard<std::mutex> lane(s->down_mu);
    // the download lane (see oip_stage_state): ordered after what the compute stream has enqueued so far
    OIP_HIP(ctx, hipEventRecord(s->down_compute_ev, ctx->stream));
    OIP_HIP(ctx, hipStreamWaitEvent(s->stream2, s->down_compute_ev, 0));
    int prev = -1;
    size_t prev_bytes = 0, prev_off = 0, done = 0;
    auto drain = [&](int i, size_t off, size_t n) -> int {
        if (hipEventSynchronize(s->dslot_free[i]) != hipSuccess) return oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
        CopyPool::get().copy((char *)host + off, s->dslot[i], n);
        return OIP_OK;
    };
    while (done < bytes) {
        const int i = s->dnext;
        s->dnext ^= 1;
        if (s->dslot_used[i]) OIP_HIP(ctx, hipEventSynchronize(s->dslot_free[i]));
        const size_t n = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        OIP_HIP(ctx, hipMemcpyAsync(s->dslot[i], (const char *)d_src + done, n, hipMemcpyDeviceToHost, s->stream2));
        hipEventRecord(s->dslot_free[i], s->stream2);
        s->dslot_used[i] = true;
        if (prev >= 0 && (rc = drain(prev, prev_off, prev_bytes))) return rc;
        prev = i; prev_off = done; prev_bytes = n;
        done += n;
    }
    if (prev >= 0) rc = drain(prev, prev_off, prev_bytes);
    return rc;
}

// ---- IMO::InplaceRRC on the reference's heap buffer (imageop.h:129-138 as DoRRC4RAW calls it, :194-228) -----------
// In place on a pageable raster: block k+1 goes up, block k is corrected and block k-1 comes down at the same time;
// the LUT is uploaded once per distinct table (cached), the host copies run on the persistent pool.
extern "C" int oip_rrc_u16_host(oip_ctx *ctx, uint16_t *buff, int w, long h, const double *kb_host)
{
    if (!ctx) return OIP_E_INVALID;         // (no OIP_CHECK_CTX: the compute thread's profiler chain is not this call's to reset)
    if (w <= 0 || h < 0 || !buff || !kb_host) return oip_fail(ctx, OIP_E_INVALID, "oip_rrc_u16_host: bad argument");
    if (h == 0) return OIP_OK;
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    std::lock_guard<std::mutex> lane(s->ring_mu);
    OIP_HIP(ctx, hipSetDevice(ctx->device));
    if (s->kb_host.size() != (size_t)w * 2 || memcmp(s->kb_host.data(), kb_host, sizeof(double) * 2 * w) != 0) {
        OIP_HIP(ctx, hipStreamSynchronize(s->stream));
        OIP_HIP(ctx, hipStreamSynchronize(s->rrc_stream));
        if (s->d_kb) OIP_HIP(ctx, hipFree(s->d_kb));
        s->d_kb = nullptr;
        OIP_HIP(ctx, hipMalloc((void **)&s->d_kb, (size_t)w * 16));
        OIP_HIP(ctx, hipMemcpy(s->d_kb, kb_host, (size_t)w * 16, hipMemcpyHostToDevice));
        s->kb_host.assign(kb_host, kb_host + (size_t)w * 2);
    }
    const size_t row_bytes = (size_t)w * 2;
    long rows = (long)(kSlotBytes / row_bytes);
    if (rows < 1) return oip_fail(ctx, OIP_E_UNSUPPORTED, "oip_rrc_u16_host: a line exceeds the staging block");
    if (rows > h) rows = h;
    const size_t block = (size_t)rows * row_bytes;
    // device blocks: two, inside the context's staging allocations
    if (ctx->stage_bytes < block) {
        for (int i = 0; i < 2; ++i) {
            if (ctx->d_stage[i]) OIP_HIP(ctx, hipFree(ctx->d_stage[i]));
            ctx->d_stage[i] = nullptr;
            OIP_HIP(ctx, hipMalloc(&ctx->d_stage[i], block));
        }
        ctx->stage_bytes = block;
    }
    const long nchunks = (h + rows - 1) / rows;
    // per chunk: pinned slot up -> device block (slot c & 1) -> kernel -> same pinned slot down; the ring has four
    // pinned slots, so the host copies of chunk c+1 (in) and c-1 (out) run while chunk c is on the device
    int slot_of[3] = {-1, -1, -1};
    long chunk_of[3] = {-1, -1, -1};
    auto drain = [&](int k) -> int {          // bring chunk chunk_of[k] home
        if (chunk_of[k] < 0) return OIP_OK;
        const long c = chunk_of[k];
        const long r0 = c * rows, n = h - r0 < rows ? h - r0 : rows;
        if (hipEventSynchronize(s->slot_free[slot_of[k]]) != hipSuccess) return oip_fail(ctx, OIP_E_DEVICE, "staged RRC block failed");
        CopyPool::get().copy(buff + r0 * (long)w, s->slot[slot_of[k]], (size_t)n * row_bytes);
        chunk_of[k] = -1;
        return OIP_OK;
    };
    for (long c = 0; c < nchunks && rc == OIP_OK; ++c) {
        const int k = (int)(c % 3);
        if ((rc = drain(k))) break;                                   // chunk c-3 (its slot is about to be reused)
        int i;
        if ((rc = slot_acquire(ctx, s, &i))) break;
        const long r0 = c * rows, n = h - r0 < rows ? h - r0 : rows;
        const size_t nb = (size_t)n * row_bytes;
        CopyPool::get().copy(s->slot[i], buff + r0 * (long)w, nb);
        void *d = ctx->d_stage[c & 1];
        hipStream_t st = (c & 1) ? s->rrc_stream : s->stream;       // device block and stream alternate: neighbours overlap
        if (hipMemcpyAsync(d, s->slot[i], nb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = oip_fail(ctx, OIP_E_DEVICE, "H2D failed"); break; }
        // explicit stream: the context's compute stream and profiler stay the compute thread's
        if ((rc = oip_rrc_launch(ctx, st, (uint16_t *)d, (uint16_t *)d, w, n, s->d_kb)) != OIP_OK) break;
        if (hipMemcpyAsync(s->slot[i], d, nb, hipMemcpyDeviceToHost, st) != hipSuccess) { rc = oip_fail(ctx, OIP_E_DEVICE, "D2H failed"); break; }
        hipEventRecord(s->slot_free[i], st);
        s->slot_used[i] = true;
        slot_of[k] = i; chunk_of[k] = c;
        // bring home the chunk before this one while this one is on the device
        if (c >= 1 && (rc = drain((int)((c - 1) % 3)))) break;
    }
    for (int k = 0; k < 3 && rc == OIP_OK; ++k) rc = drain(k);
    if (rc != OIP_OK) { hipStreamSynchronize(s->stream); hipStreamSynchronize(s->rrc_stream); }
    return rc;
}
