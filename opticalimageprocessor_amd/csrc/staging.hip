// staging.hip -- raster I/O staging between files / pageable host buffers and HBM.
//
// Replaces the staging half of ImageOperations: ReadFileContent (imageop.h:52-82: one heap buffer
// filled by 8 MiB fread calls), LoadRawImage (:110-127), WriteBufferToFile (:84-97) and the per-section
// fseek/fread/fwrite of Stitcher::PreStitch (stitcher.h:103-120).  The reference moves every raster
// through one pageable heap buffer, serially with the arithmetic; here a raster travels in blocks
// through a ring of pinned buffers on a stream of its own:
//     file --pread x T--> pinned slot --DMA--> HBM    (the next block is being read while this one flies; the slot is filled
//                                                      by T pool threads, each with a pread of its own share)
//     HBM --DMA--> pinned slot --pwrite--> file       (block k+1 comes down while block k is written; ONE writer per file:
//                                                      buffered writes hold the inode lock and the faults of a shared mapping
//                                                      contend -- measured, see write_device_to_file_at)
//     pageable buffer --pool memcpy--> pinned slot --DMA--> HBM   (and back)
// so disk, host copies, PCIe and the kernels of the context's compute stream overlap.  Staging calls
// use only staging streams and pinned slots of their own and never change the context's compute stream or
// profiler: they may run on other host threads while the first one drives kernels through the same
// context.  Three lanes, each serialised by its own mutex (a second caller of the same lane waits):
//   ring lane      oip_read_file_to_device, oip_upload_staged, oip_rrc_u16_host
//                  (four pinned slots; stream `stream`, oip_rrc_u16_host also `rrc_stream`)
//   download lanes oip_download_staged[_after], oip_write_device_to_file[_at] (three lanes of two pinned slots and a stream each:
//                  a call takes a free one, so two products can go out at the same time)
// so a reader thread and a writer thread run full duplex beside the compute thread.  The state is created once under a lock.
// A call returns a TICKET; oip_stage_wait(ctx, ticket) makes the compute stream wait -- on the device, not the
// host -- for the transfers up to that ticket, and oip_stage_sync(ctx) blocks the host until they are done.
// Ordering against the compute stream: downloads and file writes start after the compute-stream work
// enqueued before the call -- or, given a MARK (oip_compute_mark, taken by the compute thread right after the
// kernel that produced the data), after that mark only: a writer thread then does not queue behind a
// correlation batch that was enqueued later.  UPLOADS DO NOT WAIT for the compute stream (an uploader thread must keep the
// link busy while a 60-ms correlation is queued): re-uploading into a buffer that queued kernels still read
// is the caller's hazard -- call oip_stage_order_after_compute(ctx) first, or upload into another buffer.
#include "oip_internal.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace {

// ---- a small persistent pool for the host side of the ring: pageable <-> pinned copies, pread into a slot, copies into a
// file mapping (one thread moves ~10 GB/s, the link ~55: the host side, not the link, bounds a staged raster)
class CopyPool {
public:
    static CopyPool &get()
    {
        static CopyPool p;
        return p;
    }
    int threads() const { return (int)mWorkers.size() + 1; }
    int copy_threads() const { return mCopyThreads; }
    // fn(part, nparts) on the first `limit` threads of the pool (0: all; part 0 on the caller); returns when all are done
    void run(const std::function<void(int, int)> &fn, int limit = 0)
    {
        const int n = limit > 0 && limit < threads() ? limit : threads();
        if (n == 1) { fn(0, 1); return; }
        std::lock_guard<std::mutex> one_at_a_time(mCall);          // a reader and a writer thread may both be here
        {
            std::unique_lock<std::mutex> lk(mMu);
            mFn = &fn;
            mParts = n;
            mPending = (int)mWorkers.size();
            ++mGen;
        }
        mCv.notify_all();
        fn(0, n);
        std::unique_lock<std::mutex> lk(mMu);
        mDone.wait(lk, [&] { return mPending == 0; });
        mFn = nullptr;
    }
    void copy(void *dst, const void *src, size_t bytes)
    {
        if (threads() == 1 || bytes < ((size_t)4 << 20)) { memcpy(dst, src, bytes); return; }
        run([&](int i, int n) {
            const size_t part = ((bytes + n - 1) / n + 4095) & ~(size_t)4095, off = (size_t)i * part;
            if (off < bytes) memcpy((char *)dst + off, (const char *)src + off, bytes - off < part ? bytes - off : part);
        }, mCopyThreads);
    }
    // rows of `width` bytes, the pitches in bytes: the rows are dealt over the threads in contiguous runs
    void copy2d(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t rows)
    {
        auto rows_of = [&](size_t r0, size_t r1) {
            for (size_t r = r0; r < r1; ++r) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, width);
        };
        if (threads() == 1 || width * rows < ((size_t)4 << 20)) { rows_of(0, rows); return; }
        run([&](int i, int n) {
            const size_t part = (rows + n - 1) / n, r0 = (size_t)i * part;
            if (r0 < rows) rows_of(r0, r0 + part < rows ? r0 + part : rows);
        }, mCopyThreads);
    }
    // pread of [off, off + bytes) of fd into dst, every thread its own share; returns the bytes read when the file covered
    // the range, less when it ended early (the contiguous prefix), (size_t)-1 on a read error (errno kept in *err)
    size_t pread_all(int fd, void *dst, size_t bytes, size_t off, int *err)
    {
        std::atomic<size_t> short_at{bytes};
        std::atomic<int> failed{0};
        auto share = [&](int i, int n) {
            const size_t part = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
            size_t o = (size_t)i * part;
            const size_t end = o + part < bytes ? o + part : bytes;
            while (o < end) {
                const ssize_t r = pread(fd, (char *)dst + o, end - o, (off_t)(off + o));
                if (r < 0) { if (errno == EINTR) continue; failed = errno ? errno : EIO; return; }
                if (r == 0) {                                          // end of file inside this share
                    size_t cur = short_at.load();
                    while (o < cur && !short_at.compare_exchange_weak(cur, o)) {}
                    return;
                }
                o += (size_t)r;
            }
        };
        if (bytes < ((size_t)1 << 20)) share(0, 1); else run(share);
        if (failed.load()) { *err = failed.load(); return (size_t)-1; }
        return short_at.load();
    }

private:
    // Two sizes, both measured on the GPU box: pageable <-> pinned memcpy is best on 8-16 threads (more only
    // contend), pread from the page cache into a slot keeps gaining up to 32 (profiles/r04_io_*).  OIP_HOST_COPY_THREADS sets
    // the first, OIP_HOST_READ_THREADS the pool size (both at most 64).
    CopyPool()
    {
        auto env_int = [](const char *name, int def) { const char *e = getenv(name); const int v = e ? atoi(e) : def; return v < 1 ? 1 : (v > 64 ? 64 : v); };
        const int half = (int)std::thread::hardware_concurrency() / 2;
        mCopyThreads = env_int("OIP_HOST_COPY_THREADS", half > 16 ? 16 : half);
        int n = env_int("OIP_HOST_READ_THREADS", half > 32 ? 32 : half);
        if (n < mCopyThreads) n = mCopyThreads;
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
            const std::function<void(int, int)> *fn = mFn;
            const int n = mParts;
            lk.unlock();
            if (idx < n) (*fn)(idx, n);
            lk.lock();
            if (--mPending == 0) mDone.notify_one();
        }
    }
    std::vector<std::thread> mWorkers;
    std::mutex mMu, mCall;
    std::condition_variable mCv, mDone;
    unsigned long mGen = 0;
    int mPending = 0, mParts = 1, mCopyThreads = 1;
    bool mStop = false;
    const std::function<void(int, int)> *mFn = nullptr;
};

constexpr int kSlots = 4;
constexpr size_t kSlotBytes = (size_t)32 << 20;
constexpr int kTicketRing = 64;

}  // namespace

// ---- pinned slots ---------------------------------------------------------------------------------------------------
// hipHostMalloc pins 4 KiB pages one by one: 48-60 ms for the ring's 128 MiB on the GPU box (OIP_STAGE_TRACE), all of it in
// front of the first byte a fresh process moves.  An anonymous mapping backed by transparent huge pages
// (/sys/kernel/mm/transparent_hugepage/enabled = madvise on the box) registered with hipHostRegister pins 2 MiB pages
// instead.  OIP_STAGE_PIN=malloc keeps hipHostMalloc; any failure of the huge-page route falls back to it as well.
struct PinnedSlot {
    void *p = nullptr;
    size_t bytes = 0;
    void *map = nullptr;            // non-null: our own mapping, registered (else hipHostMalloc)
    size_t map_bytes = 0;
};
static bool pinned_alloc(PinnedSlot *s, size_t bytes)
{
    const char *e = getenv("OIP_STAGE_PIN");
    if (!(e && !strcmp(e, "malloc"))) {
        const size_t huge = (size_t)2 << 20;
        const size_t len = bytes + huge;
        void *m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            char *al = (char *)(((uintptr_t)m + huge - 1) & ~(uintptr_t)(huge - 1));
            madvise(al, bytes, MADV_HUGEPAGE);
            for (size_t o = 0; o < bytes; o += 4096) al[o] = 0;                 // fault it in (a huge page per first touch)
            if (hipHostRegister(al, bytes, hipHostRegisterDefault) == hipSuccess) {
                s->p = al; s->bytes = bytes; s->map = m; s->map_bytes = len;
                return true;
            }
            (void)hipGetLastError();
            munmap(m, len);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return false;
    s->p = p; s->bytes = bytes; s->map = nullptr; s->map_bytes = 0;
    return true;
}
static void pinned_free(PinnedSlot *s)
{
    if (!s->p) return;
    if (s->map) { hipHostUnregister(s->p); munmap(s->map, s->map_bytes); }
    else hipHostFree(s->p);
    s->p = nullptr;
}

// the staging state lives behind the context (opaque to the other translation units)
struct oip_stage_state {
    hipStream_t stream = nullptr;
    hipStream_t rrc_stream = nullptr;       // second stream of oip_rrc_u16_host (up and down transfers of neighbouring blocks overlap)
    std::mutex ring_mu;                     // one caller per lane at a time
    void *slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    PinnedSlot slot_mem[kSlots];
    hipEvent_t slot_free[kSlots] = {nullptr, nullptr, nullptr, nullptr};   // recorded after the DMA that last used the slot
    bool slot_used[kSlots] = {false, false, false, false};
    int next = 0;
    hipEvent_t ticket_ev[kTicketRing];
    hipEvent_t compute_ev = nullptr;        // marks the compute stream's position for uploads
    // the download lanes: each its own two slots, stream and event, so that a download or a file write on one host thread
    // and an upload on another never touch the same state (full duplex over the link); three of them, so that the
    // three strip-sized products of `prestitch` (different files: writers of ONE file serialise in the kernel anyway) go out at once
    struct DownLane {
        std::mutex mu;
        hipStream_t stream = nullptr;
        void *slot[2] = {nullptr, nullptr};
        PinnedSlot slot_mem[2];
        hipEvent_t slot_free[2] = {nullptr, nullptr};
        bool slot_used[2] = {false, false};
        int next = 0;
        hipEvent_t compute_ev = nullptr;
    } down[3];
    // marks of the compute stream (oip_compute_mark): taken by the compute thread, waited for by the download lane
    hipEvent_t mark_ev[kTicketRing];
    std::atomic<long> mark{0};
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
    // every staging entry point comes through here first: the calling thread -- possibly a fresh reader or writer thread --
    // gets the context's device before it touches a stream or an event
    OIP_HIP(ctx, hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> once(g_stage_init_mu);
    if (ctx->stage) return OIP_OK;
    const auto t_init = std::chrono::steady_clock::now();
    oip_stage_state *s = new oip_stage_state();
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->down[0].stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->down[1].stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->down[2].stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->rrc_stream, hipStreamNonBlocking) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging stream"); }
    // (pinning the four slots on a thread each was measured: the same 33 ms through huge pages, 59-128 ms instead of 43-65
    // through hipHostMalloc -- the registration serialises in the driver)
    for (int i = 0; i < kSlots; ++i) {
        if (!pinned_alloc(&s->slot_mem[i], kSlotBytes) ||
            hipEventCreateWithFlags(&s->slot_free[i], hipEventDisableTiming) != hipSuccess) {
            delete s;
            return oip_fail(ctx, OIP_E_NOMEM, "pinned staging ring (%d x %zu MiB) failed", kSlots, kSlotBytes >> 20);
        }
        s->slot[i] = s->slot_mem[i].p;
    }
    for (int i = 0; i < kTicketRing; ++i)
        if (hipEventCreateWithFlags(&s->ticket_ev[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->mark_ev[i], hipEventDisableTiming) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging events"); }
    if (hipEventCreateWithFlags(&s->compute_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->down[0].compute_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->down[1].compute_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->down[2].compute_ev, hipEventDisableTiming) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging events"); }
    // (the download lanes pin their slots on first use -- down_lane_ready: a reader that never writes does not pay for them,
    // and a writer thread pins its own while the reader is already moving data)
    for (auto &d : s->down)
        for (int i = 0; i < 2; ++i)
            if (hipEventCreateWithFlags(&d.slot_free[i], hipEventDisableTiming) != hipSuccess) { delete s; return oip_fail(ctx, OIP_E_DEVICE, "staging events"); }
    ctx->stage = s;
    if (getenv("OIP_STAGE_TRACE"))
        fprintf(stderr, "oip staging: ring of %d x %zu MiB pinned in %.1f ms (%s)\n", kSlots, kSlotBytes >> 20,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_init).count(),
                s->slot_mem[0].map ? "huge pages, hipHostRegister" : "hipHostMalloc");
    return OIP_OK;
}

// called with the lane's lock held
static int down_lane_ready(oip_ctx *ctx, oip_stage_state::DownLane &d)
{
    for (int i = 0; i < 2; ++i)
        if (!d.slot[i]) {
            if (!pinned_alloc(&d.slot_mem[i], kSlotBytes)) return oip_fail(ctx, OIP_E_NOMEM, "pinned download slots failed");
            d.slot[i] = d.slot_mem[i].p;
        }
    return OIP_OK;
}

void oip_stage_destroy(oip_ctx *ctx)
{
    oip_stage_state *s = ctx->stage;
    if (!s) return;
    if (s->stream) hipStreamSynchronize(s->stream);
    for (auto &d : s->down) if (d.stream) { hipStreamSynchronize(d.stream); hipStreamDestroy(d.stream); }
    if (s->rrc_stream) { hipStreamSynchronize(s->rrc_stream); hipStreamDestroy(s->rrc_stream); }
    for (int i = 0; i < kSlots; ++i) { pinned_free(&s->slot_mem[i]); if (s->slot_free[i]) hipEventDestroy(s->slot_free[i]); }
    for (int i = 0; i < kTicketRing; ++i) { hipEventDestroy(s->ticket_ev[i]); hipEventDestroy(s->mark_ev[i]); }
    if (s->compute_ev) hipEventDestroy(s->compute_ev);
    for (auto &d : s->down) {
        if (d.compute_ev) hipEventDestroy(d.compute_ev);
        for (int i = 0; i < 2; ++i) { pinned_free(&d.slot_mem[i]); if (d.slot_free[i]) hipEventDestroy(d.slot_free[i]); }
    }
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

extern "C" int oip_stage_threads(void) { return CopyPool::get().copy_threads(); }

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
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return oip_fail(ctx, OIP_E_INVALID, "cannot open file [%s]: %d", path, errno);  // imageop.h:55-57
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return oip_fail(ctx, OIP_E_INVALID, "ReadFileContent(): seek2end failed"); }
    const size_t avail = (size_t)st.st_size > offset ? (size_t)st.st_size - offset : 0;
    if (bytes == 0 || bytes > avail) bytes = avail;                                              // all available (imageop.h:59-63); a short file reads short
    size_t done = 0;
    while (done < bytes) {
        int i;
        if ((rc = slot_acquire(ctx, s, &i))) { close(fd); return rc; }
        const size_t want = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        // the slot is filled by the pool: every thread preads its own share (imageop.h:69-79 reads 8 MiB units on one thread)
        int err = 0;
        const auto t0 = std::chrono::steady_clock::now();
        const size_t got = CopyPool::get().pread_all(fd, s->slot[i], want, offset + done, &err);
        s->copy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        if (got == (size_t)-1) { close(fd); return oip_fail(ctx, OIP_E_RUNTIME, "read of file [%s] failed: %d", path, err); }
        s->ring_bytes += (long)got;
        if (got) {
            if (hipMemcpyAsync((char *)d_dst + done, s->slot[i], got, hipMemcpyHostToDevice, s->stream) != hipSuccess) {
                close(fd);
                return oip_fail(ctx, OIP_E_DEVICE, "H2D of a staged block failed");
            }
            hipEventRecord(s->slot_free[i], s->stream);
            s->slot_used[i] = true;
        }
        done += got;
        if (got < want) break;                                                                   // the file shrank under us
    }
    close(fd);
    ++s->ring_calls;
    if (bytes_read) *bytes_read = done;
    const long t = ticket_issue(ctx, s);
    if (ticket) *ticket = t;
    else OIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, s->ticket_ev[t % kTicketRing], 0));        // no ticket wanted: order the compute stream now
    return OIP_OK;
}

// the download lane's stream waits for the compute stream: for `mark` when one is given (and still in the ring), else for
// everything enqueued so far
static int down_order(oip_ctx *ctx, oip_stage_state *s, oip_stage_state::DownLane &d, long mark)
{
    if (mark > 0 && mark <= s->mark.load() && s->mark.load() - mark < kTicketRing - 2) {
        OIP_HIP(ctx, hipStreamWaitEvent(d.stream, s->mark_ev[mark % kTicketRing], 0));
        return OIP_OK;
    }
    OIP_HIP(ctx, hipEventRecord(d.compute_ev, ctx->stream));
    OIP_HIP(ctx, hipStreamWaitEvent(d.stream, d.compute_ev, 0));
    return OIP_OK;
}

// a free download lane (the first one when both are busy: the caller waits there)
struct DownLaneLock {
    oip_stage_state::DownLane *d;
    explicit DownLaneLock(oip_stage_state *s)
    {
        d = nullptr;
        for (auto &l : s->down)
            if (l.mu.try_lock()) { d = &l; break; }
        if (!d) { s->down[0].mu.lock(); d = &s->down[0]; }
    }
    ~DownLaneLock() { d->mu.unlock(); }
};

extern "C" int oip_compute_mark(oip_ctx *ctx, long *mark)
{
    if (!ctx || !mark) return OIP_E_INVALID;
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    const long m = s->mark.load() + 1;
    OIP_HIP(ctx, hipEventRecord(s->mark_ev[m % kTicketRing], ctx->stream));
    s->mark.store(m);
    *mark = m;
    return OIP_OK;
}

extern "C" int oip_compute_mark_sync(oip_ctx *ctx, long mark)
{
    if (!ctx) return OIP_E_INVALID;
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    if (mark > 0 && mark <= s->mark.load() && s->mark.load() - mark < kTicketRing - 2) {
        OIP_HIP(ctx, hipEventSynchronize(s->mark_ev[mark % kTicketRing]));
        return OIP_OK;
    }
    OIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return OIP_OK;
}

// HBM -> pinned slot -> file at `file_offset` (SIZE_MAX: at the end).  Block k+1 comes down while block k is written.
// Measured on the GPU box (profiles/r04_io_probe.txt): a NEW tmpfs file takes 5.3-6.7 GB/s from one writing thread (page
// allocation under the inode lock); 8-32 threads of pwrite on the one inode stay at 3.5-5.6 GB/s, a MAP_SHARED mapping
// filled by 1-64 threads at 3.4 -> 1.3 GB/s (the faults contend).  So: one writer, plain pwrite of whole slots
// (OIP_FILE_WRITE=mmap keeps the mapping route for file systems where it pays).
static int write_device_to_file_at(oip_ctx *ctx, const void *d_src, size_t bytes, const char *path, size_t file_offset, bool truncate, long mark)
{
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    DownLaneLock lane(s);
    oip_stage_state::DownLane &d = *lane.d;
    if ((rc = down_lane_ready(ctx, d))) return rc;
    const int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC | (truncate ? O_TRUNC : 0), 0644);
    if (fd < 0) return oip_fail(ctx, OIP_E_RUNTIME, "open file [%s] failed: %d", path, errno);  // imageop.h:86-88
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return oip_fail(ctx, OIP_E_RUNTIME, "open file [%s] failed: %d", path, errno); }
    if (file_offset == (size_t)-1) file_offset = (size_t)st.st_size;
    char *map = nullptr;
    size_t map_len = 0, map_skew = 0;
    const char *mode = getenv("OIP_FILE_WRITE");
    if (mode && !strcmp(mode, "mmap") && bytes) {
        // a mapping reports a full file system as SIGBUS, not as an error: only with room to spare
        struct statvfs vfs;
        const bool room = fstatvfs(fd, &vfs) == 0 && (double)vfs.f_bavail * (double)vfs.f_frsize > (double)bytes + 256e6;
        if (room && ((size_t)st.st_size >= file_offset + bytes || ftruncate(fd, (off_t)(file_offset + bytes)) == 0)) {
            const size_t page = (size_t)sysconf(_SC_PAGESIZE);
            map_skew = file_offset % page;
            map_len = bytes + map_skew;
            void *m = mmap(nullptr, map_len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)(file_offset - map_skew));
            if (m != MAP_FAILED) { map = (char *)m; madvise(m, map_len, MADV_HUGEPAGE); }
        }
    }
    if ((rc = down_order(ctx, s, d, mark))) { if (map) munmap(map, map_len); close(fd); return rc; }
    int prev = -1;
    size_t prev_bytes = 0, prev_off = 0, done = 0;
    auto flush = [&](int i, size_t off, size_t n) -> int {
        if (hipEventSynchronize(d.slot_free[i]) != hipSuccess) return oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
        if (map) { CopyPool::get().copy(map + map_skew + off, d.slot[i], n); return OIP_OK; }
        size_t w = 0;
        while (w < n) {                                                                          // imageop.h:88-95
            const ssize_t wb = pwrite(fd, (const char *)d.slot[i] + w, n - w, (off_t)(file_offset + off + w));
            if (wb < 0 && errno == EINTR) continue;
            if (wb <= 0) return oip_fail(ctx, OIP_E_RUNTIME, "write file failed: %d", errno);
            w += (size_t)wb;
        }
        return OIP_OK;
    };
    while (done < bytes) {
        const int i = d.next;
        d.next ^= 1;
        if (d.slot_used[i] && hipEventSynchronize(d.slot_free[i]) != hipSuccess) { rc = oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed"); break; }
        const size_t n = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        if (hipMemcpyAsync(d.slot[i], (const char *)d_src + done, n, hipMemcpyDeviceToHost, d.stream) != hipSuccess) {
            rc = oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
            break;
        }
        hipEventRecord(d.slot_free[i], d.stream);
        d.slot_used[i] = true;
        if (prev >= 0 && (rc = flush(prev, prev_off, prev_bytes))) break;
        prev = i; prev_off = done; prev_bytes = n;
        done += n;
    }
    if (rc == OIP_OK && prev >= 0) rc = flush(prev, prev_off, prev_bytes);
    if (rc != OIP_OK) hipStreamSynchronize(d.stream);
    if (map) munmap(map, map_len);
    if (close(fd) != 0 && rc == OIP_OK) rc = oip_fail(ctx, OIP_E_RUNTIME, "close file [%s] failed: %d", path, errno);
    return rc;
}

// ---- a product file prepared ahead of its pixels --------------------------------------------------------------------
// What a buffered write of a NEW file costs is allocating its pages, one by one under the inode lock (5-7 GB/s on the box's
// tmpfs, and no second thread helps: r04_io_probe).  A product whose size is known before its pixels exist -- the aligned
// image while the strip is still being read -- is prepared by a thread that has nothing else to do then: the file is
// created, its blocks are reserved (posix_fallocate: a full file system fails HERE, cleanly) and mapped MAP_SHARED.  When
// the pixels arrive they go HBM -> pinned slot -> mapping with the slot copied by the pool's threads: memory copies into
// pages that exist (first touches map them, in parallel), several times the rate of the write() path.  Any step that fails leaves a sink
// that writes through pwrite instead.
struct oip_file_sink {
    int fd = -1;
    char *map = nullptr;
    size_t bytes = 0;
    std::string path;
};

extern "C" int oip_file_sink_open(oip_ctx *ctx, const char *path, size_t bytes, oip_file_sink **out)
{
    if (!ctx) return OIP_E_INVALID;
    if (!path || !out) return oip_fail(ctx, OIP_E_INVALID, "oip_file_sink_open: bad argument");
    *out = nullptr;
    const int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0644);       // (no O_TRUNC: a header written before stays)
    if (fd < 0) return oip_fail(ctx, OIP_E_RUNTIME, "open file [%s] failed: %d", path, errno);
    oip_file_sink *k = new oip_file_sink();
    k->fd = fd;
    k->bytes = bytes;
    k->path = path;
    const char *mode = getenv("OIP_FILE_WRITE");
    if (bytes && !(mode && !strcmp(mode, "pwrite")) && posix_fallocate(fd, 0, (off_t)bytes) == 0) {
        // NOT MAP_POPULATE: populating 1.5 GB holds the process's mmap_lock for ~0.2 s and every thread that maps, unmaps or
        // faults meanwhile -- the reader, the compute thread's allocations -- stands still (measured: the second correlation
        // section started 230 ms late).  The reserved pages are mapped by the copy threads' own first touches instead, which
        // run in parallel; OIP_SINK_POPULATE=1 populates ahead in 16 MiB steps (MADV_POPULATE_WRITE) for comparison.
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m != MAP_FAILED) {
            k->map = (char *)m;
            const char *pop = getenv("OIP_SINK_POPULATE");
#ifdef MADV_POPULATE_WRITE
            if (pop && atoi(pop) == 1)
                for (size_t o = 0; o < bytes; o += (size_t)16 << 20)
                    if (madvise(k->map + o, bytes - o < ((size_t)16 << 20) ? bytes - o : ((size_t)16 << 20), MADV_POPULATE_WRITE) != 0) break;
#else
            (void)pop;
#endif
        }
    }
    *out = k;
    return OIP_OK;
}

extern "C" int oip_file_sink_close(oip_ctx *ctx, oip_file_sink *k)
{
    if (!k) return OIP_OK;
    int rc = OIP_OK;
    if (k->map) munmap(k->map, k->bytes);
    if (k->fd >= 0 && close(k->fd) != 0 && ctx) rc = oip_fail(ctx, OIP_E_RUNTIME, "close file [%s] failed: %d", k->path.c_str(), errno);
    delete k;
    return rc;
}

extern "C" int oip_file_sink_write(oip_ctx *ctx, oip_file_sink *k, size_t file_offset, const void *d_src, size_t bytes, long mark)
{
    if (!ctx) return OIP_E_INVALID;
    if (!k || (!d_src && bytes) || file_offset + bytes > k->bytes) return oip_fail(ctx, OIP_E_INVALID, "oip_file_sink_write: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    DownLaneLock lane(s);
    oip_stage_state::DownLane &d = *lane.d;
    if ((rc = down_lane_ready(ctx, d))) return rc;
    if ((rc = down_order(ctx, s, d, mark))) return rc;
    int prev = -1;
    size_t prev_bytes = 0, prev_off = 0, done = 0;
    auto flush = [&](int i, size_t off, size_t n) -> int {
        if (hipEventSynchronize(d.slot_free[i]) != hipSuccess) return oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
        if (k->map) { CopyPool::get().copy(k->map + file_offset + off, d.slot[i], n); return OIP_OK; }
        size_t w = 0;
        while (w < n) {
            const ssize_t wb = pwrite(k->fd, (const char *)d.slot[i] + w, n - w, (off_t)(file_offset + off + w));
            if (wb < 0 && errno == EINTR) continue;
            if (wb <= 0) return oip_fail(ctx, OIP_E_RUNTIME, "write file failed: %d", errno);
            w += (size_t)wb;
        }
        return OIP_OK;
    };
    while (done < bytes) {
        const int i = d.next;
        d.next ^= 1;
        if (d.slot_used[i] && hipEventSynchronize(d.slot_free[i]) != hipSuccess) { rc = oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed"); break; }
        const size_t n = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        if (hipMemcpyAsync(d.slot[i], (const char *)d_src + done, n, hipMemcpyDeviceToHost, d.stream) != hipSuccess) {
            rc = oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
            break;
        }
        hipEventRecord(d.slot_free[i], d.stream);
        d.slot_used[i] = true;
        if (prev >= 0 && (rc = flush(prev, prev_off, prev_bytes))) break;
        prev = i; prev_off = done; prev_bytes = n;
        done += n;
    }
    if (rc == OIP_OK && prev >= 0) rc = flush(prev, prev_off, prev_bytes);
    if (rc != OIP_OK) hipStreamSynchronize(d.stream);
    return rc;
}

extern "C" int oip_write_device_to_file(oip_ctx *ctx, const void *d_src, size_t bytes, const char *path, int append)
{
    if (!ctx) return OIP_E_INVALID;
    if (!path || (!d_src && bytes)) return oip_fail(ctx, OIP_E_INVALID, "oip_write_device_to_file: bad argument");
    return write_device_to_file_at(ctx, d_src, bytes, path, append ? (size_t)-1 : 0, !append, 0);
}

extern "C" int oip_write_device_to_file_at(oip_ctx *ctx, const void *d_src, size_t bytes, const char *path, size_t file_offset, long mark)
{
    if (!ctx) return OIP_E_INVALID;
    if (!path || (!d_src && bytes)) return oip_fail(ctx, OIP_E_INVALID, "oip_write_device_to_file_at: bad argument");
    return write_device_to_file_at(ctx, d_src, bytes, path, file_offset, false, mark);
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

extern "C" int oip_download_staged_after(oip_ctx *ctx, void *host, const void *d_src, size_t bytes, long mark)
{
    if (!ctx) return OIP_E_INVALID;
    if ((!d_src || !host) && bytes) return oip_fail(ctx, OIP_E_INVALID, "oip_download_staged: bad argument");
    int rc = stage_init(ctx);
    if (rc) return rc;
    oip_stage_state *s = ctx->stage;
    DownLaneLock lane(s);
    oip_stage_state::DownLane &d = *lane.d;
    if ((rc = down_lane_ready(ctx, d))) return rc;
    // a download lane (see oip_stage_state): ordered after the mark, or after what the compute stream has enqueued so far
    if ((rc = down_order(ctx, s, d, mark))) return rc;
    int prev = -1;
    size_t prev_bytes = 0, prev_off = 0, done = 0;
    auto drain = [&](int i, size_t off, size_t n) -> int {
        if (hipEventSynchronize(d.slot_free[i]) != hipSuccess) return oip_fail(ctx, OIP_E_DEVICE, "D2H of a staged block failed");
        CopyPool::get().copy((char *)host + off, d.slot[i], n);
        return OIP_OK;
    };
    while (done < bytes) {
        const int i = d.next;
        d.next ^= 1;
        if (d.slot_used[i]) OIP_HIP(ctx, hipEventSynchronize(d.slot_free[i]));
        const size_t n = bytes - done < kSlotBytes ? bytes - done : kSlotBytes;
        OIP_HIP(ctx, hipMemcpyAsync(d.slot[i], (const char *)d_src + done, n, hipMemcpyDeviceToHost, d.stream));
        hipEventRecord(d.slot_free[i], d.stream);
        d.slot_used[i] = true;
        if (prev >= 0 && (rc = drain(prev, prev_off, prev_bytes))) return rc;
        prev = i; prev_off = done; prev_bytes = n;
        done += n;
    }
    if (prev >= 0) rc = drain(prev, prev_off, prev_bytes);
    return rc;
}

extern "C" int oip_download_staged(oip_ctx *ctx, void *host, const void *d_src, size_t bytes)
{
    return oip_download_staged_after(ctx, host, d_src, bytes, 0);
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
