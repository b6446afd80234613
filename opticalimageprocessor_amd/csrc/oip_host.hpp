// oip_host.hpp -- C++ host layer above the C ABI (include/oip_c.h), mirroring the reference's
// class interface for the hot path: ImageOperations (IMO), Stitcher, PreProcessor -- same
// method names, argument meaning and error behaviour (exception types -> exit codes of
// main.cpp:320-343), with the raster loops running on the MI355X.  Used by the `oip` CLI.
//
// Differences kept deliberately small:
//   * the line width is a constructor/run-time argument (reference: PIXELS_PER_LINE 12288);
//   * TIFF input and output go through a dependency-free TIFF/BigTIFF codec (oip_tiff.hpp) instead of
//     GDAL / cv::imread / cv::imwrite;
//   * timing lines are logged like the reference's (seconds, MBps) through a plain logger.
#pragma once

#include <sys/stat.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <filesystem>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "oip_c.h"
#include "oip_tiff.hpp"

namespace OIPGPU {

struct errno_error : public std::runtime_error {      // libimsux errno_error stand-in (same role)
    explicit errno_error(const std::string &s) : std::runtime_error(s + ": " + std::strerror(errno)) {}
    errno_error(const std::string &complete, int) : std::runtime_error(complete) {}   // the C ABI's message already names the errno
};
struct usage_error : public std::invalid_argument {   // main.cpp:20-23
    explicit usage_error(const std::string &s) : std::invalid_argument(s) {}
};

// ---- logging (libimsux logger stand-in: timestamped lines to stdout and $LOGFILE / oip.log) ----
inline FILE *&log_file() { static FILE *f = nullptr; return f; }
inline void log_line(bool stamped, const char *fmt, ...)
{
    char msg[2048];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    char ts[64] = "";
    if (stamped) {
        time_t t = time(nullptr);
        struct tm tmv;
        localtime_r(&t, &tmv);
        strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S ", &tmv);
    }
    printf("%s%s\n", ts, msg);
    if (log_file()) { fprintf(log_file(), "%s%s\n", ts, msg); fflush(log_file()); }
}
#define OLOG(...) ::OIPGPU::log_line(true, __VA_ARGS__)
#define RLOG(...) ::OIPGPU::log_line(false, __VA_ARGS__)

struct stop_watch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double tick()
    {
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s > 0 ? s : 1e-9;
    }
};

// Which compression a TIFF product gets.  "reference" (default) = what the reference's libraries do for that product:
// cv::imwrite -> LZW + horizontal predictor (ALIGNED.TIFF, stitched MSS without -g), GDAL with COMPRESS=LZW
// PREDICTOR=2 (stitched MSS with -g, imageop.h:471-472), GDAL with default options -> none (stitched PAN,
// imageop.h:316-328).  --tiff-compress none|lzw (or $OIP_TIFF_COMPRESS) forces one for every product.
inline int &tiff_policy() { static int p = -1; return p; }       // -1 reference, else TIFF_NONE / TIFF_LZW
inline int tiff_compression(int reference_choice)
{
    if (tiff_policy() < 0) {
        const char *e = getenv("OIP_TIFF_COMPRESS");
        if (e && std::string(e) == "none") return TIFF_NONE;
        if (e && std::string(e) == "lzw") return TIFF_LZW;
        return reference_choice;
    }
    return tiff_policy();
}

inline std::string to_lower(std::string s)
{
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

// ---- small threading helpers of the pipelined actions ----------------------------------------------------
template <typename T> class BlockingQueue {
public:
    void push(T v)
    {
        { std::lock_guard<std::mutex> lk(mMu); mQ.push_back(std::move(v)); }
        mCv.notify_one();
    }
    T pop()
    {
        std::unique_lock<std::mutex> lk(mMu);
        mCv.wait(lk, [&] { return !mQ.empty(); });
        T v = std::move(mQ.front());
        mQ.pop_front();
        return v;
    }
private:
    std::mutex mMu;
    std::condition_variable mCv;
    std::deque<T> mQ;
};

// a thread that runs jobs in the order they are posted (a product writer: HBM -> file beside the compute thread); the
// first exception stops it and is re-thrown by finish()
class JobThread {
public:
    JobThread() : mThread([this] { loop(); }) {}
    ~JobThread() { try { finish(); } catch (...) {} }
    void post(std::function<void()> job) { mQ.push(std::move(job)); }
    void finish()
    {
        if (mThread.joinable()) { mQ.push(nullptr); mThread.join(); }
        if (mError) { auto e = mError; mError = nullptr; std::rethrow_exception(e); }
    }
private:
    void loop()
    {
        for (;;) {
            std::function<void()> job = mQ.pop();
            if (!job) return;
            if (mError) continue;                 // drain after a failure
            try { job(); } catch (...) { mError = std::current_exception(); }
        }
    }
    BlockingQueue<std::function<void()>> mQ;
    std::exception_ptr mError;
    std::thread mThread;
};

inline double seconds_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
inline std::chrono::steady_clock::time_point &process_start()
{
    static std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    return t;
}

// ---- device context + error mapping ----------------------------------------------------------
class Device {
public:
    static Device &get()
    {
        static Device d;
        return d;
    }
    oip_ctx *ctx()
    {
        if (!mCtx) {
            const auto t0 = std::chrono::steady_clock::now();
            if (oip_create(0, &mCtx) != OIP_OK)
                throw std::runtime_error("no usable MI355X (gfx950) device: this build has no CPU fallback");
            mCreateSeconds = seconds_since(t0);
        }
        return mCtx;
    }
    double create_seconds() const { return mCreateSeconds; }
    void check(int rc)
    {
        if (rc == OIP_OK) return;
        std::string m = oip_last_error(mCtx);
        switch (rc) {
            case OIP_E_INVALID: throw std::invalid_argument(m);
            case OIP_E_IO: throw errno_error(m, 0);
            default: throw std::runtime_error(m);
        }
    }
    ~Device() { if (mCtx) oip_destroy(mCtx); }
private:
    oip_ctx *mCtx = nullptr;
    double mCreateSeconds = 0.0;
};

template <typename T> struct DevBuf {              // RAII device buffer
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    void alloc(size_t count)
    {
        release();
        void *v = nullptr;
        Device::get().check(oip_malloc(Device::get().ctx(), &v, count * sizeof(T)));
        p = (T *)v;
        n = count;
    }
    void release() { if (p) oip_free(Device::get().ctx(), p); p = nullptr; n = 0; }
    // pageable host memory moves through the pinned staging ring (include/oip_c.h, raster I/O staging); small
    // tables take the plain copy
    void upload(const T *h, size_t count)
    {
        if (count * sizeof(T) >= ((size_t)8 << 20)) Device::get().check(oip_upload_staged(Device::get().ctx(), p, h, count * sizeof(T), nullptr));
        else Device::get().check(oip_memcpy_h2d(Device::get().ctx(), p, h, count * sizeof(T)));
    }
    void download(T *h, size_t count) const
    {
        if (count * sizeof(T) >= ((size_t)8 << 20)) { Device::get().check(oip_download_staged(Device::get().ctx(), h, p, count * sizeof(T))); return; }
        Device::get().check(oip_memcpy_d2h(Device::get().ctx(), h, p, count * sizeof(T)));
        Device::get().check(oip_sync(Device::get().ctx()));
    }
    // ReadFileContent / LoadRawImage with the buffer in HBM (imageop.h:52-82, :110-127): the whole file, which
    // must hold exactly `count` elements
    void load_file(const std::string &filePath, size_t count)
    {
        OLOG("Reading raw image from file `%s' ...", filePath.c_str());
        stop_watch sw;
        size_t got = 0;
        Device::get().check(oip_read_file_to_device(Device::get().ctx(), filePath.c_str(), 0, count * sizeof(T), p, &got, nullptr));
        if (got != count * sizeof(T))
            throw std::runtime_error("file size(" + std::to_string(count * sizeof(T)) + ") doesn't match with read byte count(" +
                                     std::to_string(got) + ")");
        double es = sw.tick();
        OLOG("%zu bytes read in %.3f seconds (%.1f MBps).", got, es, got / es / 1024.0 / 1024.0);
    }
    // WriteBufferToFile with the buffer in HBM (imageop.h:84-97)
    void save_file(const std::string &filePath, size_t count) const
    {
        Device::get().check(oip_write_device_to_file(Device::get().ctx(), p, count * sizeof(T), filePath.c_str(), 0));
    }
    void swap(DevBuf &o) { std::swap(p, o.p); std::swap(n, o.n); }
    ~DevBuf() { release(); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

constexpr int BYTES_PER_PIXEL = 2;
constexpr int MSS_BANDS = OIP_MSS_BANDS;

// rows x rowSamples u16 in HBM -> the writer, 256 MiB of lines at a time, the next block coming down (staging download lane)
// while the strips of the current one are encoded (a few threads) and the previous one's are written (TiffWriterU16's own
// thread): download || encode || write
// OIP_TIFF_GPU_LZW=0: LZW strips on the host's threads (the encoder of oip_tiff.hpp; same file bytes) -- test and A/B knob
inline bool gpu_lzw()
{
    static const bool on = [] { const char *e = getenv("OIP_TIFF_GPU_LZW"); return !(e && atoi(e) == 0); }();
    return on;
}

inline void tiff_rows_from_device(TiffWriterU16 &tw, const uint16_t *d_img, long rows, size_t rowSamples, long mark)
{
    oip_ctx *ctx = Device::get().ctx();
    size_t chunkBytes = (size_t)256 << 20;
    if (const char *e = getenv("OIP_TIFF_CHUNK_MB")) {                 // test hook: several blocks on a small image
        const long mb = atol(e);
        if (mb > 0) chunkBytes = (size_t)mb << 20;
    }
    const long chunk = std::max<long>(1, (long)(chunkBytes / (rowSamples * 2)));
    const size_t cap = (size_t)std::min(chunk, rows) * rowSamples;
    std::unique_ptr<uint16_t[]> buf[2] = {std::unique_ptr<uint16_t[]>(new uint16_t[cap]), std::unique_ptr<uint16_t[]>(rows > chunk ? new uint16_t[cap] : nullptr)};
    auto fetch = [&](long r0, uint16_t *dst) {
        const long nr = std::min(chunk, rows - r0);
        return oip_download_staged_after(ctx, dst, d_img + (size_t)r0 * rowSamples, (size_t)nr * rowSamples * 2, mark);
    };
    Device::get().check(fetch(0, buf[0].get()));
    int cur = 0;
    for (long r0 = 0; r0 < rows; r0 += chunk) {
        const long nr = std::min(chunk, rows - r0);
        std::future<int> next;
        if (r0 + chunk < rows) next = std::async(std::launch::async, fetch, r0 + chunk, buf[cur ^ 1].get());
        try {
            tw.write_rows(buf[cur].get(), nr);
        } catch (...) {
            if (next.valid()) next.wait();
            throw;
        }
        if (next.valid()) Device::get().check(next.get());
        cur ^= 1;
    }
}

// An LZW product whose pixels are in HBM, in file sample order: the strips are encoded where the image is (csrc/tifflzw.hip:
// a lane per strip) and leave the device packed, as one block that goes into the file behind the header.  The encoder runs on
// the compute stream: everything the image waits for is ahead of it there.
// TiffLzwPrep: what a caller that knows the product's geometry before its pixels exist prepares meanwhile (the pipelined
// default action, on its product writer's thread while the strip is still being read): the device buffers, and the file with
// the worst-case size reserved and mapped -- writing a NEW file is bound by the allocation of its pages; the
// file is cut back to the encoded size afterwards.
struct TiffLzwPrep {
    DevBuf<uint8_t> payload, scratch;
    oip_file_sink *sink = nullptr;
    uint64_t payloadAt = 0;
    void prepare(TiffWriterU16 &tw, const std::string &path, int width, long height, int spp, bool withSink)
    {
        const long rps = tw.rows_per_strip();
        payload.alloc(oip_tiff_lzw_worst_bytes(height, width, spp, rps));
        scratch.alloc(oip_tiff_lzw_scratch_bytes(height, width, spp, rps));
        if (withSink) {
            payloadAt = tw.begin_external_strips();
            Device::get().check(oip_file_sink_open(Device::get().ctx(), path.c_str(), (size_t)payloadAt + payload.n, &sink));
        }
    }
    ~TiffLzwPrep() { if (sink) oip_file_sink_close(nullptr, sink); }
};

inline void tiff_lzw_from_device(TiffWriterU16 &tw, const std::string &path, const uint16_t *d_img, int width, long height, int spp,
                                 TiffLzwPrep *prep = nullptr)
{
    oip_ctx *ctx = Device::get().ctx();
    stop_watch sw;
    const long rps = tw.rows_per_strip();
    const size_t nstrips = (size_t)((height + rps - 1) / rps);
    TiffLzwPrep own;
    std::future<int> reserving;
    size_t reserved = 0;
    if (!prep) {
        // nothing was prepared: the product file is reserved and mapped by a helper thread WHILE the strips are being encoded
        // -- at the size sensor data encodes to (raw + 3 %: LZW does not compress it); a payload that turns out larger goes
        // the plain way.  (Writing a new file is bound by the allocation of its pages)
        prep = &own;
        const size_t worst = oip_tiff_lzw_worst_bytes(height, width, spp, rps);
        prep->payload.alloc(worst);
        const size_t raw = (size_t)height * width * spp * 2;
        if (raw >= ((size_t)64 << 20)) {
            reserved = std::min(worst, raw + raw / 32 + ((size_t)64 << 10));
            own.payloadAt = tw.begin_external_strips();
            TiffLzwPrep *op = &own;
            reserving = std::async(std::launch::async, [=] { return oip_file_sink_open(ctx, path.c_str(), (size_t)op->payloadAt + reserved, &op->sink); });
        }
    } else if (prep->sink) {
        reserved = prep->payload.n;
    }
    std::vector<uint64_t> off(nstrips), len(nstrips);
    size_t bytes = 0;
    const double tAlloc = sw.tick();
    const int rcEncode = oip_tiff_lzw_strips_u16(ctx, d_img, height, width, spp, rps, prep->payload.p, prep->payload.n, off.data(), len.data(), &bytes,
                                                 prep->scratch.p, prep->scratch.n);
    if (reserving.valid() && reserving.get() != OIP_OK && prep->sink) { oip_file_sink_close(nullptr, prep->sink); prep->sink = nullptr; }
    Device::get().check(rcEncode);
    const double tEncode = sw.tick();
    const size_t even = (bytes + 1) & ~(size_t)1;
    if (prep->sink && even <= reserved) {
        Device::get().check(oip_file_sink_write(ctx, prep->sink, (size_t)prep->payloadAt, prep->payload.p, even, 0));
        oip_file_sink *k = prep->sink;
        prep->sink = nullptr;
        Device::get().check(oip_file_sink_close(ctx, k));
        if (::truncate(path.c_str(), (off_t)(prep->payloadAt + even)) != 0) throw errno_error("truncate() of the TIFF product failed");
    } else {
        uint64_t at = prep->payloadAt;
        if (prep->sink) {                                              // reserved too little: drop the reservation, write the plain way
            oip_file_sink *k = prep->sink;
            prep->sink = nullptr;
            Device::get().check(oip_file_sink_close(ctx, k));
            if (::truncate(path.c_str(), (off_t)at) != 0) throw errno_error("truncate() of the TIFF product failed");
        } else if (!reserved) {
            at = tw.begin_external_strips();
        }
        Device::get().check(oip_write_device_to_file_at(ctx, prep->payload.p, even, path.c_str(), (size_t)at, 0));
    }
    tw.end_external_strips(off.data(), len.data(), nstrips, bytes);
    const double tWrite = sw.tick();
    RLOG("TIMING tiff_lzw strips=%zu alloc=%.4f encode=%.4f write=%.4f bytes=%zu prepared=%d", nstrips, tAlloc, tEncode, tWrite, bytes, prep != &own);
}

// A TIFF product whose pixels are in HBM (rows x width x spp u16, interleaved).  Uncompressed: the pixel payload goes from
// the device into the file behind the header (TiffWriterU16::begin_external_payload + oip_write_device_to_file_at) -- no
// strip-sized heap copy, as the reference's cv::imwrite / GDAL paths make (imageop.h:316-328, preproc.h:167-185).  LZW: the
// lines come down 256 MiB at a time and their strips are encoded on a few threads.  opencv_order (4 samples): the image takes
// cv::imwrite's on-disk sample order (c2,c1,c0,c3) ON THE DEVICE first -- d_img is modified; `order` (4 samples, GDAL band
// maps): out sample i = in sample order[i], likewise.  mark: 0, or the compute-stream mark the image is complete at.
inline void write_tiff_from_device(const std::string &path, uint16_t *d_img, int width, long height, int spp, int compression,
                                   bool opencv_order, const int *order = nullptr, long mark = 0, bool permute_done = false)
{
    oip_ctx *ctx = Device::get().ctx();
    if (spp == MSS_BANDS && !permute_done && (opencv_order || order)) {
        const int cvOrder[4] = {2, 1, 0, 3};
        Device::get().check(oip_permute_u16x4(ctx, d_img, (size_t)width * height, order ? order : cvOrder));
        mark = 0;                                                      // (the permutation is younger than the mark)
    }
    TiffWriterU16 tw(path, width, height, spp, false, compression);
    const size_t rowSamples = (size_t)width * spp;
    if (compression == TIFF_NONE) {
        const uint64_t at = tw.begin_external_payload();
        Device::get().check(oip_write_device_to_file_at(ctx, d_img, (size_t)height * rowSamples * 2, path.c_str(), (size_t)at, mark));
        tw.end_external_payload();
    } else if (gpu_lzw()) {
        tiff_lzw_from_device(tw, path, d_img, width, height, spp);
    } else {
        tiff_rows_from_device(tw, d_img, height, rowSamples, mark);
    }
    tw.close();
}


// cv::imread of a 16-bit TIFF (imageop.h:380-388) with the image ending up in HBM.  LZW files -- what the reference's own
// products are -- go from the file into device memory as they are and their strips are decoded there (csrc/tifflzw.hip: a
// strip per lane); anything else, and OIP_TIFF_GPU_LZW=0, decodes on the host's threads (read_tiff_u16) and uploads.  Same
// checks and error texts either way: the header is validated by the host reader in both.
inline void read_tiff_to_device(const std::string &path, int *width, long *height, int *spp, DevBuf<uint16_t> &img)
{
    TiffLayout lay;
    read_tiff_u16(path, width, height, spp, nullptr, &lay);
    const size_t samples = (size_t)lay.width * lay.height * lay.spp;
    bool device = gpu_lzw() && lay.compression == TIFF_LZW && (lay.spp == 1 || lay.spp == 4) &&
                  lay.rows_per_strip * lay.width * lay.spp * 2 < ((uint64_t)1 << 32);
    uint64_t lo = ~(uint64_t)0, hi = 0;
    for (size_t k = 0; k < lay.offs.size(); ++k) { lo = std::min(lo, lay.offs[k]); hi = std::max(hi, lay.offs[k] + lay.lens[k]); }
    if (device && hi - lo > 3 * samples + ((uint64_t)64 << 20)) device = false;       // strips scattered over a far larger file
    if (!device) {
        std::vector<uint16_t> host;
        read_tiff_u16(path, width, height, spp, &host);
        img.alloc(host.size());
        img.upload(host.data(), host.size());
        return;
    }
    oip_ctx *ctx = Device::get().ctx();
    DevBuf<uint8_t> file((size_t)(hi - lo));
    size_t got = 0;
    Device::get().check(oip_read_file_to_device(ctx, path.c_str(), (size_t)lo, (size_t)(hi - lo), file.p, &got, nullptr));
    if (got != (size_t)(hi - lo)) throw std::runtime_error("read TIFF [" + path + "]: truncated file");
    Device::get().check(oip_stage_sync(ctx));
    for (auto &o : lay.offs) o -= lo;
    img.alloc(samples);
    const int rc = oip_tiff_lzw_decode_u16(ctx, file.p, file.n, lay.offs.data(), lay.lens.data(), (long)lay.offs.size(), (long)lay.height,
                                           (int)lay.width, (int)lay.spp, (long)lay.rows_per_strip, (int)lay.predictor, img.p);
    if (rc != OIP_OK) throw std::runtime_error("read TIFF [" + path + "]: " + std::string(oip_last_error(ctx)));
}

// ---- ImageOperations (imageop.h:33-568, hot-path subset) -----------------------------------------
struct RRCParam { double k; double b; };     // imageop.h:26-29

class ImageOperations {
public:
    static size_t FileSize(const std::string &filePath)            // imageop.h:43-47
    {
        struct stat st;
        memset(&st, 0, sizeof st);
        if (stat(filePath.c_str(), &st)) throw errno_error("stat() call for file failed");
        return (size_t)st.st_size;
    }

    // imageop.h:52-82: size = bytes read; offset; total = 0 for all available
    static char *ReadFileContent(const std::string &filePath, size_t &size, size_t offset = 0, size_t total = 0,
                                 char *buff = nullptr)
    {
        FILE *f = fopen(filePath.c_str(), "rb");
        if (!f) throw std::invalid_argument("cannot open file [" + filePath + "]: " + std::to_string(errno));
        size_t want = total;
        if (total == 0) {
            if (fseek(f, 0, SEEK_END)) { fclose(f); throw std::invalid_argument("ReadFileContent(): seek2end failed"); }
            want = (size_t)ftello(f) - offset;
        }
        if (fseeko(f, (off_t)offset, SEEK_SET)) { fclose(f); throw std::invalid_argument("ReadFileContent(): rewind failed"); }
        if (!buff) buff = new char[want ? want : 1];
        const size_t unit = 8u * 1024 * 1024;
        size_t rb = 0;
        for (char *p = buff;;) {
            size_t rn = fread(p, 1, std::min(unit, want - rb), f);
            p += rn;
            rb += rn;
            if (rn == 0 || rb == want) { size = rb; break; }
        }
        fclose(f);
        return buff;
    }

    static size_t WriteBufferToFile(const char *buff, size_t size, const std::string &saveFilePath)   // imageop.h:84-97
    {
        FILE *f = fopen(saveFilePath.c_str(), "wb");
        if (!f) throw std::runtime_error("open file [" + saveFilePath + "] failed: " + std::to_string(errno));
        const size_t unit = std::min((size_t)8 * 1024 * 1024, size);
        size_t written = 0;
        for (const char *p = buff; written < size;) {
            size_t wb = fwrite(p, 1, std::min(unit, size - written), f);
            if (wb == 0) { fclose(f); throw std::runtime_error("write file failed: " + std::to_string(errno)); }
            written += wb;
            p += wb;
        }
        fclose(f);
        return written;
    }

    // imageop.h:99-108: cwd / stem(template) + stemExtension + (replaceExtension | ext(template))
    static std::string BuildOutputFilePath(const std::string &templatePath, const std::string &stemExtension,
                                           const char *replaceExtension = nullptr)
    {
        auto cd = std::filesystem::current_path();
        std::filesystem::path tmpl = templatePath;
        auto out = cd / tmpl.stem();
        out += stemExtension;
        out += replaceExtension ? std::string(replaceExtension) : tmpl.extension().string();
        return out.string();
    }

    static void *LoadRawImage(const std::string &filePath, size_t offset = 0, size_t bytes = 0, size_t expectedSize = 0)
    {                                                                // imageop.h:110-127
        OLOG("Reading raw image from file `%s' ...", filePath.c_str());
        size_t size = 0;
        stop_watch sw;
        void *content = ReadFileContent(filePath, size, offset, bytes);
        if (expectedSize > 0 && size != expectedSize) {
            delete[] (char *)content;
            throw std::runtime_error("file size(" + std::to_string(expectedSize) + ") doesn't match with read byte count(" +
                                     std::to_string(size) + ")");
        }
        double es = sw.tick();
        OLOG("%zu bytes read in %.3f seconds (%.1f MBps).", size, es, size / es / 1024.0 / 1024.0);
        return content;
    }

    // imageop.h:129-138 -- on the GPU, through pinned double-buffered line blocks
    static void InplaceRRC(uint16_t *buff, int w, int h, const RRCParam *rrcParam)
    {
        Device::get().check(oip_rrc_u16_host(Device::get().ctx(), buff, w, h, reinterpret_cast<const double *>(rrcParam)));
    }

    static RRCParam *LoadRRCParamFile(const char *paramFilePath, int expectedLines)      // imageop.h:140-192
    {
        OLOG("Loading RRC paramter from file `%s' ...", paramFilePath);
        std::unique_ptr<RRCParam[]> p(new RRCParam[expectedLines]);
        char err[1024];
        int rc = oip_load_rrc_param_file(paramFilePath, expectedLines, reinterpret_cast<double *>(p.get()), err, sizeof err);
        if (rc == OIP_E_IO) throw errno_error(err);
        if (rc != OIP_OK) throw std::runtime_error(err);
        OLOG("LoadRRCParamFile(): loaded.");
        return p.release();
    }

    // imageop.h:194-228.  The raster goes file -> HBM -> file through the staging ring (no whole-file heap buffer);
    // keepBuffer hands back a heap copy like the reference does.
    static uint16_t *DoRRC4RAW(const std::string &raw, int pixelPerLine, const std::string &rrc,
                               const std::string saveRaw = "", bool keepBuffer = false)
    {
        size_t size = FileSize(raw);
        const size_t npx = size / BYTES_PER_PIXEL;
        DevBuf<uint16_t> image(npx ? npx : 1);
        image.load_file(raw, npx);
        long lines = (long)(size / ((size_t)pixelPerLine * BYTES_PER_PIXEL));
        std::unique_ptr<RRCParam[]> rrcParam(LoadRRCParamFile(rrc.c_str(), pixelPerLine));
        DevBuf<double> kb((size_t)pixelPerLine * 2);
        kb.upload((double *)rrcParam.get(), (size_t)pixelPerLine * 2);
        OLOG("Do inplace RRC ...");
        stop_watch sw;
        Device::get().check(oip_rrc_u16(Device::get().ctx(), image.p, image.p, pixelPerLine, lines, kb.p));
        Device::get().check(oip_sync(Device::get().ctx()));
        double es = sw.tick();
        OLOG("Done for %zu bytes in %.3f seconds (%.1f MBps).", size, es, size / es / (1024.0 * 1024.0));
        if (!saveRaw.empty()) {
            OLOG("Write RRC result as file \"%s\" ...", saveRaw.c_str());
            sw.tick();
            image.save_file(saveRaw, npx);
            es = sw.tick();
            OLOG("%zu bytes written in %.3f seconds (%.1f MBps).", size, es, size / es / (1024.0 * 1024.0));
        }
        if (!keepBuffer) return nullptr;
        std::unique_ptr<uint16_t[]> out(new uint16_t[npx ? npx : 1]);
        image.download(out.get(), npx);
        return out.release();
    }

    // imageop.h:277-363, RAW output only (the GTiff writer is "next")
    static std::string StitchBigRaw(const std::string &leftImagePath, const std::string &rightImagePath,
                                    const std::string &stitchedFilePath, int pixelPerLine, int foldColPixels)
    {
        size_t szl = FileSize(leftImagePath), szr = FileSize(rightImagePath);
        if (szl != szr)
            throw std::invalid_argument("RAW image sizes not match: left = " + std::to_string(szl) + " bytes, right = " +
                                        std::to_string(szr) + " bytes");
        const size_t bytesPerLine = (size_t)pixelPerLine * BYTES_PER_PIXEL;
        const long imageLines = (long)(szl / bytesPerLine);
        const int outputFullLinePixels = (pixelPerLine - foldColPixels) * 2;
        bool outputIsTiff = true;                                      // imageop.h:297-306
        std::string outputFilePath = stitchedFilePath;
        if (stitchedFilePath.empty()) {
            outputFilePath = (std::filesystem::current_path() /
                              ("stitched_" + std::to_string(outputFullLinePixels) + "n" + std::to_string(BYTES_PER_PIXEL * 8) + "b.TIFF")).string();
        } else {
            outputIsTiff = to_lower(std::filesystem::path(stitchedFilePath).extension().string()) == ".tiff";
        }
        const size_t npx = (size_t)pixelPerLine * imageLines, nout = (size_t)outputFullLinePixels * imageLines;
        DevBuf<uint16_t> dl(npx), dr(npx), dout(nout);
        dl.load_file(leftImagePath, npx);
        dr.load_file(rightImagePath, npx);
        OLOG("Begin stitching two images ...");
        stop_watch sw;
        Device::get().check(oip_stitch_rows_u16(Device::get().ctx(), dl.p, dr.p, dout.p, pixelPerLine, imageLines, foldColPixels));
        if (outputIsTiff) {                                            // 1-band GTiff (imageop.h:316-328), straight from the device
            write_tiff_from_device(outputFilePath, dout.p, outputFullLinePixels, imageLines, 1, tiff_compression(TIFF_NONE), false);
        } else {
            dout.save_file(outputFilePath, nout);
        }
        double es = sw.tick();
        OLOG("%zu bytes written in %.3f seconds (%.1f MBps).", nout * 2, es, nout * 2 / es / (1024.0 * 1024.0));
        return outputFilePath;
    }

    // imageop.h:365-457 (+ StitchTiffGDAL :460-567): two 16UC4 images -> column-range concat.  cv::imread
    // hands the reference BGRA-ordered Mats, i.e. the original channel order c0..c3; the same is rebuilt
    // here from the on-disk RGBA order.  Without -g the result is re-encoded like cv::imwrite; with -g
    // GDAL writes band b from channel bandMap[b]-1 (imageop.h:529).
    static std::string StitchTiff(const std::string &leftImagePath, const std::string &rightImagePath,
                                  const std::string &stitchedFilePath, int foldColPixels, bool useGDAL = false,
                                  const int *bandMap = nullptr)
    {
        std::string outputFilePath = stitchedFilePath;
        if (stitchedFilePath.empty()) outputFilePath = (std::filesystem::current_path() / "stitched.TIFF").string();
        else if (to_lower(std::filesystem::path(stitchedFilePath).extension().string()) != ".tiff")
            throw std::invalid_argument("Output file should be a tiff image");
        int wl, wr, sl, sr;
        long hl, hr;
        DevBuf<uint16_t> dl, dr;
        OLOG("Reading tiff image from file `%s' ...", leftImagePath.c_str());
        read_tiff_to_device(leftImagePath, &wl, &hl, &sl, dl);
        OLOG("Reading tiff image from file `%s' ...", rightImagePath.c_str());
        read_tiff_to_device(rightImagePath, &wr, &hr, &sr, dr);
        if (hl != hr || wl != wr) throw std::runtime_error("images have different sizes");
        if (sl != MSS_BANDS || sr != MSS_BANDS) throw std::runtime_error("StitchTiff(): 4-channel 16-bit images expected");
        if (foldColPixels < 0 || foldColPixels >= wl) throw std::invalid_argument("fold columns exceed the image width");
        // the stitch itself: 4 interleaved samples per pixel are just 4x wider u16 lines
        const int W4 = wl * 4, fold4 = foldColPixels * 4;
        const size_t nin = (size_t)W4 * hl, nout = (size_t)2 * (W4 - fold4) * hl;
        (void)nin;
        DevBuf<uint16_t> dout(nout);
        Device::get().check(oip_stitch_rows_u16(Device::get().ctx(), dl.p, dr.p, dout.p, W4, hl, fold4));
        const int ow = 2 * (wl - foldColPixels);
        OLOG("Write stitched image to file '%s' ...", outputFilePath.c_str());
        // file order is RGBA = (c2, c1, c0, c3) of the reference's Mat.  The stitched image stays on the device: its samples are
        // put in file order there and the product goes out like every other TIFF of the CLI (LZW strips encoded on the device)
        const size_t bytes = nout * 2;
        if (!useGDAL && bytes / 2 < 4000000000ull) {
            write_tiff_from_device(outputFilePath, dout.p, ow, hl, MSS_BANDS, tiff_compression(TIFF_LZW), false);   // same on-disk order in and out
        } else {
            const int mat2file[4] = {2, 1, 0, 3};              // Mat channel c lives at file sample mat2file[c]
            int order[4];
            for (int b = 0; b < 4; ++b) order[b] = mat2file[bandMap ? bandMap[b] - 1 : b];
            write_tiff_from_device(outputFilePath, dout.p, ow, hl, MSS_BANDS, tiff_compression(TIFF_LZW), false, order);
        }
        return outputFilePath;
    }
};
typedef ImageOperations IMO;

// ---- Stitcher (stitcher.h:18-223) ------------------------------------------------------------------
class Stitcher {
public:
    // stitcher.h:21-46; foldCols is the already-halved value (main.cpp:189)
    static std::string Stitch(const std::string &leftImagePath, const std::string &rightImagePath,
                              const std::string &outputPath = "", int foldCols = 0, int pixelsPerLine = OIP_PIXELS_PER_LINE,
                              bool useGDAL = false, const int *bandMap = nullptr)
    {
        std::string leftExt = to_lower(std::filesystem::path(leftImagePath).extension().string());
        std::string rightExt = to_lower(std::filesystem::path(rightImagePath).extension().string());
        if (leftExt != rightExt) throw std::invalid_argument("Stitch(): two images should be same type");
        if (leftExt != ".tiff" && leftExt != ".raw") throw std::invalid_argument("Stitch(): only RAW and TIFF image supported");
        if (leftExt == ".raw") return IMO::StitchBigRaw(leftImagePath, rightImagePath, outputPath, pixelsPerLine, foldCols);
        return IMO::StitchTiff(leftImagePath, rightImagePath, outputPath, foldCols, useGDAL, bandMap);
    }

    Stitcher(const std::string &pan1, const std::string &pan2, const std::string &rrc1, const std::string &rrc2,
             int sections = OIP_STT_DEF_SECTIONS, int linePerSection = OIP_STT_DEF_SECLINES,
             int overlapCols = OIP_STT_DEF_OVERLAPPX, int pixelsPerLine = OIP_PIXELS_PER_LINE)
        : mFilePAN1(pan1), mFilePAN2(pan2), mParamFileRRC1(rrc1), mParamFileRRC2(rrc2), mSections(sections),
          mLinePerSection(linePerSection), mOverlapCols(overlapCols), mW(pixelsPerLine)
    {                                                                  // stitcher.h:49-81
        size_t s1 = IMO::FileSize(pan1);
        if ((size_t)sections * linePerSection * BYTES_PER_PIXEL > s1)
            throw std::invalid_argument("PAN1 size too small for SECTION & LINE_PER_SECTION argument");
        size_t s2 = IMO::FileSize(pan2);
        if ((size_t)sections * linePerSection * BYTES_PER_PIXEL > s2)
            throw std::invalid_argument("PAN2 size too small for SECTION & LINE_PER_SECTION argument");
        if (s1 != s2) throw std::invalid_argument("PAN1 size doesn't match PAN2 size");
        mSizePAN = s1;
        mLinesPAN = (int)(s1 / ((size_t)mW * BYTES_PER_PIXEL));
        OLOG("PAN: %d lines total.", mLinesPAN);
        if (mLinesPAN < sections * linePerSection)
            throw std::invalid_argument("PAN line count less than sections times line-per-section, use smaller -s and/or -l value(s)");
        mRrcFilePAN1 = mFilePAN1;
        mRrcFilePAN2 = mFilePAN2;
    }

    // The reference reads both PAN files three times (CalcSttParameters, DoRRC, PreStitch) and writes two of them
    // back in between; here each raw strip is read ONCE into HBM and every step works on the resident copy -- the
    // .RRC.RAW / .RRC.PRESTT.RAW products are still written, from the device.
    void LoadRawOnce()
    {
        const size_t npx = (size_t)mW * mLinesPAN;
        if (!mD1.p) { mD1.alloc(npx); mD1.load_file(mFilePAN1, npx); }
        if (!mD2.p) { mD2.alloc(npx); mD2.load_file(mFilePAN2, npx); }
    }

    // stitcher.h:148-201 (runs on the files mRrcFilePAN1/2 point at: the raw ones, App.B-1)
    void CalcSttParameters(double threshold = OIP_STT_DEF_PHCTHRHLD, double maxDeltaY = 0.0, int edgeCols = 0)
    {
        LoadRawOnce();
        std::vector<double> r(3 * (size_t)mSections);
        Device::get().check(oip_stt_correlate(Device::get().ctx(), mD1.p, mD2.p, mW, mLinesPAN, 0, mLinesPAN, mSections,
                                              mLinePerSection, mOverlapCols, edgeCols, r.data()));
        const int gapLines = (mLinesPAN - mSections * mLinePerSection) / (mSections + 1);
        const int stepLines = gapLines + mLinePerSection;
        OLOG("Calculating stitching delta values ...");
        RLOG("| offset |  delta x |  delta y | response | r |");
        RLOG("-----------------------------------------------");
        for (int i = 0; i < mSections; ++i) {
            double dx = r[3 * i], dy = r[3 * i + 1], resp = r[3 * i + 2];
            bool isValid = resp >= threshold && (maxDeltaY <= 0.0 || std::abs(dy) <= maxDeltaY);
            RLOG("|%7d |%10.4f|%10.4f|%10.4f|%s|", gapLines + i * stepLines, dx, dy, resp, isValid ? " Y " : " N ");
        }
        int valid = 0;                                                  // stitcher.h:181-198
        if (oip_stt_mean(r.data(), mSections, threshold, maxDeltaY, &mDeltaX, &mDeltaY, &mResponse, &valid) != OIP_OK)
            throw std::runtime_error("No valid delta value found for stitching parameter calculating");
        OLOG("Total %d valid delta value pairs found, everage value:", valid);
        OLOG("    dx: %.5f, dy: %.5f, r: %.5f", mDeltaX, mDeltaY, mResponse);
    }

    // The three strip-sized products of `prestitch` (.RRC.RAW x 2, .RRC.PRESTT.RAW: 6 GB each at the 30000 x 100000 geometry) go
    // out on writer threads behind marks of the compute stream, each on a download lane of its own: a NEW file takes 6-7 GB/s
    // from one writer whatever else the process does, so three files at once are what shortens the command.
    // Finish() -- called by PreStitch, by the destructor and by the CLI -- waits for them and logs the reference's lines.
    void DoRRC()                                                        // stitcher.h:141-146
    {
        LoadRawOnce();
        mRrcFilePAN1 = IMO::BuildOutputFilePath(mFilePAN1, ".RRC");
        mRrcFilePAN2 = IMO::BuildOutputFilePath(mFilePAN2, ".RRC");
        const size_t npx = (size_t)mW * mLinesPAN;
        DevBuf<double> kb((size_t)mW * 2);
        oip_ctx *ctx = Device::get().ctx();
        for (int c = 0; c < 2; ++c) {                                   // IMO::DoRRC4RAW on the resident strip
            std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile((c ? mParamFileRRC2 : mParamFileRRC1).c_str(), mW));
            kb.upload((double *)prm.get(), (size_t)mW * 2);
            DevBuf<uint16_t> &d = c ? mD2 : mD1;
            OLOG("Do inplace RRC ...");
            stop_watch sw;
            Device::get().check(oip_rrc_u16(ctx, d.p, d.p, mW, (long)mLinesPAN, kb.p));
            Device::get().check(oip_sync(ctx));
            double es = sw.tick();
            OLOG("Done for %zu bytes in %.3f seconds (%.1f MBps).", mSizePAN, es, mSizePAN / es / (1024.0 * 1024.0));
            const std::string save = c ? mRrcFilePAN2 : mRrcFilePAN1;
            OLOG("Write RRC result as file \"%s\" ...", save.c_str());
            long mark = 0;
            Device::get().check(oip_compute_mark(ctx, &mark));
            const uint16_t *src = d.p;
            const size_t bytes = mSizePAN;
            (void)npx;
            mWriter[c].post([=] {
                stop_watch w;
                { FILE *f = fopen(save.c_str(), "wb"); if (!f) throw std::runtime_error("open file [" + save + "] failed: " + std::to_string(errno)); fclose(f); }
                Device::get().check(oip_write_device_to_file_at(ctx, src, bytes, save.c_str(), 0, mark));
                const double s = w.tick();
                OLOG("%zu bytes written in %.3f seconds (%.1f MBps).", bytes, s, bytes / s / (1024.0 * 1024.0));
            });
        }
    }

    // fp16acc: the fp16-accumulate resampling variant (BASELINE config 5; not the parity mode)
    int PreStitch(bool fp16acc = false)                                 // stitcher.h:83-139
    {
        LoadRawOnce();                                                  // mD2 holds what mRrcFilePAN2 names (raw with --no-rrc)
        mPreSttFilePAN2 = IMO::BuildOutputFilePath(mRrcFilePAN2, ".PRESTT");
        const size_t npx = (size_t)mW * mLinesPAN;
        oip_ctx *ctx = Device::get().ctx();
        stop_watch sw;
        mPre.alloc(npx);
        auto fn = fp16acc ? oip_remap_shift_bicubic_u16_f16acc : oip_remap_shift_bicubic_u16;
        Device::get().check(fn(ctx, mD2.p, 0, mLinesPAN, mPre.p, 0, mLinesPAN, mW, mLinesPAN, mDeltaX, mDeltaY,
                               OIP_REMAP_SECTION_ROWS, OIP_REMAP_ROW_GUARD));
        long mark = 0;
        Device::get().check(oip_compute_mark(ctx, &mark));
        const std::string save = mPreSttFilePAN2;
        const uint16_t *src = mPre.p;
        const size_t bytes = mSizePAN;
        mWriter[2].post([=] {
            { FILE *f = fopen(save.c_str(), "wb"); if (!f) throw std::runtime_error("open file [" + save + "] failed: " + std::to_string(errno)); fclose(f); }
            Device::get().check(oip_write_device_to_file_at(ctx, src, bytes, save.c_str(), 0, mark));
        });
        Finish();
        double es = sw.tick();
        OLOG("Pre-stitched PAN2 written to file '%s'.", mPreSttFilePAN2.c_str());
        OLOG("%zu bytes processed & written in %.3f seconds (%.1f MBps).", mSizePAN, es, mSizePAN / es / (1024.0 * 1024.0));
        const int ucut = mDeltaY >= 0.0 ? 0 : (int)(-mDeltaY) + 1, bcut = mDeltaY >= 0.0 ? (int)mDeltaY + 1 : 0;
        return mLinesPAN - (ucut + bcut);        // SectionaryRemap's returned row_offset
    }

    // every product posted so far is on disk when this returns (the first failure of a writer is re-thrown)
    void Finish()
    {
        for (auto &w : mWriter) w.finish();
    }

    double deltaX() const { return mDeltaX; }
    double deltaY() const { return mDeltaY; }
    double response() const { return mResponse; }

private:
    std::string mFilePAN1, mFilePAN2, mParamFileRRC1, mParamFileRRC2, mRrcFilePAN1, mRrcFilePAN2, mPreSttFilePAN2;
    double mDeltaX = 0, mDeltaY = 0, mResponse = 0;
    size_t mSizePAN = 0;
    int mSections, mLinePerSection, mOverlapCols, mLinesPAN = 0, mW;
    DevBuf<uint16_t> mD1, mD2, mPre;            // the two strips, resident from the first step that needs them; the resampled CCD 2
    JobThread mWriter[3];                       // (declared after the buffers: joined before they are released)
};

// ---- PreProcessor (preproc.h:30-599) ---------------------------------------------------------------
struct InterBandShift { double dx, dy, rs; int cx; };                   // preproc.h:23-28

class PreProcessor {
public:
    PreProcessor(const std::string &panFile, const std::string &mssFile, const std::string &rrcFile4PAN,
                 const std::string rrcFile4MSSBand[MSS_BANDS], int pixelsPerLine = OIP_PIXELS_PER_LINE)
        : mPanFile(panFile), mMssFile(mssFile), mRrcPanFile(rrcFile4PAN), mW(pixelsPerLine)
    {
        for (int i = 0; i < MSS_BANDS; ++i) mRrcMssBndFile[i] = rrcFile4MSSBand[i];
        CheckFilesAttributes();
    }

    void LoadPAN()                                                      // preproc.h:51-54
    {
        OLOG("Loading PAN raw image ...");
        mPAN.alloc(mSizePAN / 2);
        mPAN.load_file(mPanFile, mSizePAN / 2);
    }

    // Fused task: the PAN strip is already on the device (RRC'd / pre-stitched by the
    // previous step of the same process) -- no file round trip.  Not owned.
    void UseDevicePAN(const uint16_t *d_pan) { mPanView = d_pan; }

    // --fit reference (default): NumCpp Poly1d::fit as the reference calls it; --fit lstsq: QR on a scaled abscissa
    void SetFitMode(int mode) { mFitMode = mode; }

    void LoadMSS()                                                      // preproc.h:56-80 (split deferred to DoRRC4MSS)
    {
        OLOG("Loading MSS raw image ...");
        mMssBil.alloc(mSizeMSS / 2);
        mMssBil.load_file(mMssFile, mSizeMSS / 2);
        mPlaneStride = (size_t)(mW / MSS_BANDS) * mLinesMSS;
        mPlanes.alloc(mPlaneStride * MSS_BANDS);
        Device::get().check(oip_sync(Device::get().ctx()));
        mSplitDone = false;
    }

    void DoRRC4PAN()                                                    // preproc.h:188-200
    {
        if (!mPAN.p) throw std::logic_error("PAN raw image data not loaded, call `LoadPAN()' first");
        std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(mRrcPanFile.c_str(), mW));
        DevBuf<double> kb((size_t)mW * 2);
        kb.upload((double *)prm.get(), (size_t)mW * 2);
        OLOG("Begin inplace RRC for PAN data ... ");
        stop_watch sw;
        Device::get().check(oip_rrc_u16(Device::get().ctx(), mPAN.p, mPAN.p, mW, (long)mLinesPAN, kb.p));
        Device::get().check(oip_sync(Device::get().ctx()));
        double es = sw.tick();
        OLOG("RRC for PAN done in %.4f seconds (%.1f MBps).", es, mSizePAN / es / 1024.0 / 1024.0);
    }

    // preproc.h:202-222; doRRC=false is --no-rrc4mss (split only)
    void DoRRC4MSS(bool doRRC = true)
    {
        if (!mMssBil.p) throw std::logic_error("MSS raw image data not loaded, call `LoadMSS()' first");
        const int bw = mW / MSS_BANDS;
        DevBuf<double> kb;
        if (doRRC) {
            std::vector<double> all((size_t)mW * 2);
            for (int i = 0; i < MSS_BANDS; ++i) {
                std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(mRrcMssBndFile[i].c_str(), bw));
                memcpy(&all[(size_t)i * bw * 2], prm.get(), sizeof(double) * 2 * bw);
            }
            kb.alloc((size_t)mW * 2);
            kb.upload(all.data(), (size_t)mW * 2);
        }
        OLOG("Splitting %d bands of MSS image%s ...", MSS_BANDS, doRRC ? " with inplace RRC" : "");
        stop_watch sw;
        Device::get().check(oip_mss_split_rrc_u16(Device::get().ctx(), mMssBil.p, mPlanes.p, mPlaneStride, mW, (long)mLinesMSS,
                                                  doRRC ? kb.p : nullptr));
        Device::get().check(oip_sync(Device::get().ctx()));
        double es = sw.tick();
        OLOG("RRC done for MSS bands in %.4f seconds (%.1f MBps).", es, mSizeMSS / es / 1024.0 / 1024.0);
        mMssBil.release();
        mSplitDone = true;
    }

    void CalcInterBandCorrelation(int slices = OIP_IBCV_DEF_SLICES, int sections = OIP_IBCV_DEF_SECTIONS,
                                  double threshold = OIP_IBCV_DEF_THRESHOLD, bool autoUnloadPAN = true)
    {                                                                   // preproc.h:224-347
        if (!mSplitDone) DoRRC4MSS(false);
        OLOG("Calculating inter-band correlation with %d slices in %d section(s) ...", slices, sections);
        const int n = slices * sections;
        std::vector<double> t((size_t)MSS_BANDS * n * 4);
        stop_watch sw;
        const uint16_t *pan = mPanView ? mPanView : mPAN.p;
        if (!pan) throw std::logic_error("PAN raw image data not loaded, call `LoadPAN()' first");
        Device::get().check(oip_interband_correlate(Device::get().ctx(), pan, (long)mLinesPAN, 0, (long)mLinesPAN, mPlanes.p,
                                                    mPlaneStride, 0, (long)mLinesMSS, mW, slices, sections,
                                                    OIP_CORRELATION_LINES, t.data()));
        OLOG("Inter-band correlation finished in %.3f seconds, result:", sw.tick());
        for (int b = 0; b < MSS_BANDS; ++b) {
            mBandShift[b].resize(n);
            for (int i = 0; i < n; ++i) {
                const double *s = &t[((size_t)b * n + i) * 4];
                mBandShift[b][i] = InterBandShift{s[0], s[1], s[2], (int)s[3]};
            }
        }
        DumpInterBandShiftValues(slices, sections);
        OLOG("Filter invalid correlation values & try polynomial fitting ...");
        char err[512];
        int rc = oip_filter_and_fit_mode(t.data(), n, threshold, OIP_IBCV_MIN_COUNT, mFitMode, &mDeltaXcoeffs[0][0], &mDeltaYcoeffs[0][0], err, sizeof err);
        if (rc != OIP_OK) { OLOG("%s.", err); throw std::runtime_error(err); }
        for (int b = 0; b < MSS_BANDS; ++b) {
            OLOG("BAND %d\tdeltaX coeff: [1] %.15f, [0] %.9f", b, mDeltaXcoeffs[b][1], mDeltaXcoeffs[b][0]);
            OLOG("\tdeltaY coeff: [2] %.15f, [1] %.15f, [0] %.9f", mDeltaYcoeffs[b][2], mDeltaYcoeffs[b][1], mDeltaYcoeffs[b][0]);
        }
        OLOG("CalcInterBandCorrelation(): done.");
        if (autoUnloadPAN) mPAN.release();
    }

    // preproc.h:351-425 (+ inner :428-468); writes <stem>.ALIGNED.TIFF
    // keepOnDevice (fused task): the aligned 16UC4 image stays on the device (swapped into *keepOnDevice,
    // *keptRows lines) and no file is written
    void DoInterBandAlignment(int linePerSection, int lineOffset = 0, int sectionOverlap = OIP_IBPA_DEFAULT_LINEOVERLAP,
                              bool keepLeadingLines = false, bool autoUnloadRawMSS = true, DevBuf<uint16_t> *keepOnDevice = nullptr,
                              long *keptRows = nullptr)
    {
        OLOG("Doing inter-band alignment ...");
        const int Wb = mW / MSS_BANDS;
        const long rows = (long)mLinesMSS - lineOffset - (keepLeadingLines ? 0 : sectionOverlap);
        if (rows <= 0) throw std::invalid_argument("Too few image lines left to process");
        DevBuf<uint16_t> out((size_t)rows * Wb * MSS_BANDS);
        long processed = 0;
        stop_watch sw;
        Device::get().check(oip_align_mss_bicubic_u16x4(Device::get().ctx(), mPlanes.p, mPlaneStride, 0, (long)mLinesMSS, out.p, 0,
                                                        rows, Wb, (long)mLinesMSS, &mDeltaXcoeffs[0][0], &mDeltaYcoeffs[0][0],
                                                        linePerSection, lineOffset, sectionOverlap, keepLeadingLines ? 1 : 0,
                                                        OIP_IBPA_MIN_PROCESSLINES, &processed));
        if (keepOnDevice) {
            Device::get().check(oip_sync(Device::get().ctx()));
            OLOG("Alignment done in %.3f seconds (%ld lines valid of %ld).", sw.tick(), processed, rows);
            keepOnDevice->swap(out);
            if (keptRows) *keptRows = rows;
            if (autoUnloadRawMSS) mPlanes.release();
            OLOG("DoInterBandAlignment(): done.");
            return;
        }
        Device::get().check(oip_sync(Device::get().ctx()));
        double es = sw.tick();
        OLOG("Alignment done in %.3f seconds (%ld lines valid of %ld).", es, processed, rows);
        // preproc.h:167-185 WriteAlignedMSS_TIFF: 4-channel 16-bit TIFF, samples in OpenCV's on-disk order
        auto save = IMO::BuildOutputFilePath(mMssFile, ".ALIGNED", ".TIFF");
        OLOG("Outputing aligned TIFF image (%d x %ld x 4) to [%s] ...", Wb, rows, save.c_str());
        write_tiff_from_device(save, out.p, Wb, rows, MSS_BANDS, tiff_compression(TIFF_LZW), true);
        OLOG("Output done.");
        if (autoUnloadRawMSS) mPlanes.release();
        OLOG("DoInterBandAlignment(): done.");
    }

    void WriteRRCedPAN()                                                // preproc.h:93-105
    {
        auto save = IMO::BuildOutputFilePath(mPanFile, ".RRC");
        mPAN.save_file(save, mSizePAN / 2);
        OLOG("Written to file [%s].", save.c_str());
    }

    // ---- the default action (main.cpp:288-317) as ONE pipeline ------------------------------------------------------
    // The reference runs LoadPAN, LoadMSS, DoRRC4PAN, DoRRC4MSS, CalcInterBandCorrelation and DoInterBandAlignment one after
    // the other, each over the whole strip (preproc.h:51-80, :188-222, :224-347, :351-425), and so do the step methods above.
    // Here three host threads share the context:
    //   reader   the two RAW files go file -> pinned ring -> HBM (oip_read_file_to_device, parallel pread) in the order the
    //            arithmetic needs them: per correlation section its MSS lines, then its PAN lines in blocks; the MSS lines
    //            between the sections; the PAN lines between the sections last.  Every block carries a ticket.
    //   compute  (this thread) waits for a block's ticket on the device, corrects it (RRC in place / BIL split + RRC), runs a
    //            section's phase correlations as soon as its lines are resident (the pairs of units of the one-call form are
    //            kept, so the shifts are the serial run's bits), then filter + fit, the alignment kernel and the sample
    //            permutation of the product -- while the reader is still bringing the PAN lines between the sections.
    //   writers  <pan>.RRC.RAW goes out block by block behind the RRC kernel of each block (--write-rrcpan), the aligned image
    //            goes HBM -> pinned -> <mss>.ALIGNED.TIFF (uncompressed: straight into the file's pixel payload; LZW: strips
    //            encoded on a few threads as their lines come down).
    // Products are byte-identical to the step-by-step flow's (tests/test_gpu_cli.py runs both; OIP_PIPELINE=0 selects the steps).
    struct DefaultActionOptions {
        bool doRRC4PAN = false, writeRrcPan = false, doRRC4MSS = true, keepLeading = false;
        int slices = OIP_IBCV_DEF_SLICES, sections = OIP_IBCV_DEF_SECTIONS;
        double threshold = OIP_IBCV_DEF_THRESHOLD;
        int linesSection = OIP_IBPA_DEFAULT_BATCHLINES, lineOffset = 0, overlapLines = OIP_IBPA_DEFAULT_LINEOVERLAP;
    };

    void RunPipelined(const DefaultActionOptions &o)
    {
        oip_ctx *ctx = Device::get().ctx();
        auto ck = [](int rc) { Device::get().check(rc); };
        const auto t0 = std::chrono::steady_clock::now();
        const int W = mW, Wb = W / MSS_BANDS;
        const long Lp = (long)mLinesPAN, Lm = (long)mLinesMSS;
        // the argument checks of CalcInterBandCorrelation (preproc.h:228-237) and DoInterBandAlignment, before a byte is read
        char msg[256];
        if (o.slices < OIP_IBCV_MIN_SLICES) {
            snprintf(msg, sizeof msg, "CalcInterBandCorrelation: at lease %d slice needed", OIP_IBCV_MIN_SLICES);
            throw std::invalid_argument(msg);
        }
        if (o.sections <= 0) throw std::invalid_argument("CalcInterBandCorrelation: section count should be a positive integer");
        if (o.sections > 1 && (long)o.sections * OIP_CORRELATION_LINES > Lp) {
            snprintf(msg, sizeof msg, "CalcInterBandCorrelation: too many sections (%d lines per section), not enough total PAN data lines", OIP_CORRELATION_LINES);
            throw std::invalid_argument(msg);
        }
        const long outRows = Lm - o.lineOffset - (o.keepLeading ? 0 : o.overlapLines);
        if (outRows <= 0) throw std::invalid_argument("Too few image lines left to process");
        // preproc.h:245-247, :274-276
        const int baseRows = (int)std::min<long>(Lp, OIP_CORRELATION_LINES), baseCols = W / o.slices;
        const long baseGap = (Lp - (long)baseRows * o.sections) / (o.sections + 1);
        const int bandRows = baseRows / MSS_BANDS, bandCols = baseCols / MSS_BANDS;
        const long bandGap = baseGap / MSS_BANDS;
        if (bandRows <= 0 || bandCols <= 0) throw std::invalid_argument("oip_interband_correlate: slice too small");
        std::vector<long> p0(o.sections), m0(o.sections);
        for (int sec = 0; sec < o.sections; ++sec) {
            p0[sec] = baseGap + (long)sec * (baseRows + baseGap);                 // preproc.h:257
            m0[sec] = bandGap + (long)sec * (bandRows + bandGap);                 // preproc.h:284
            if (p0[sec] + baseRows > Lp || m0[sec] + bandRows > Lm) throw std::invalid_argument("oip_interband_correlate: section outside the strip");
        }
        mPAN.alloc((size_t)W * Lp);
        mMssBil.alloc((size_t)W * Lm);
        mPlaneStride = (size_t)Wb * Lm;
        mPlanes.alloc(mPlaneStride * MSS_BANDS);
        DevBuf<uint16_t> out((size_t)outRows * Wb * MSS_BANDS);
        DevBuf<double> kbPan, kbMss;
        const double tSetup = seconds_since(t0);

        // ---- the order of arrival
        enum Kind { kMss, kPan, kUnits, kFit };
        struct Item { Kind kind; long a, b; };
        std::vector<Item> order;
        const long blockLines = std::max<long>(1, (long)(((size_t)256 << 20) / ((size_t)W * BYTES_PER_PIXEL)));
        auto pan_blocks = [&](long a, long b) { for (long r = a; r < b; r += blockLines) order.push_back({kPan, r, std::min(b, r + blockLines)}); };
        for (int sec = 0; sec < o.sections; ++sec) {
            order.push_back({kMss, m0[sec], m0[sec] + bandRows});
            pan_blocks(p0[sec], p0[sec] + baseRows);
            order.push_back({kUnits, sec, 0});
        }
        {   // the MSS lines between the sections: only the alignment reads them
            long prev = 0;
            for (int sec = 0; sec <= o.sections; ++sec) {
                const long a = sec < o.sections ? m0[sec] : Lm;
                if (prev < a) order.push_back({kMss, prev, a});
                prev = sec < o.sections ? m0[sec] + bandRows : Lm;
            }
        }
        order.push_back({kFit, 0, 0});
        {   // the PAN lines between the sections: no arithmetic of this action reads them (they are read and corrected as
            // the reference does -- the strip's size is checked, --write-rrcpan wants them)
            long prev = 0;
            for (int sec = 0; sec <= o.sections; ++sec) {
                const long a = sec < o.sections ? p0[sec] : Lp;
                if (prev < a) pan_blocks(prev, a);
                prev = sec < o.sections ? p0[sec] + baseRows : Lp;
            }
        }

        // ---- reader thread
        OLOG("Loading PAN raw image ...");
        OLOG("Loading MSS raw image ...");
        OLOG("Reading raw image from file `%s' ...", mPanFile.c_str());
        OLOG("Reading raw image from file `%s' ...", mMssFile.c_str());
        BlockingQueue<long> tickets;                       // one per read item, in order; -1: the reader failed
        std::atomic<bool> cancel{false};
        std::exception_ptr readerError;
        double tReadDone = 0.0;
        const size_t lineBytes = (size_t)W * BYTES_PER_PIXEL;
        std::thread reader([&] {
            try {
                for (const Item &it : order) {
                    if (it.kind != kMss && it.kind != kPan) continue;
                    if (cancel.load()) break;
                    const bool pan = it.kind == kPan;
                    const size_t off = (size_t)it.a * lineBytes, n = (size_t)(it.b - it.a) * lineBytes;
                    size_t got = 0;
                    long t = 0;
                    const int rc = oip_read_file_to_device(ctx, (pan ? mPanFile : mMssFile).c_str(), off, n,
                                                           (char *)(pan ? mPAN.p : mMssBil.p) + off, &got, &t);
                    if (rc != OIP_OK) Device::get().check(rc);
                    if (got != n)
                        throw std::runtime_error("file size(" + std::to_string(pan ? mSizePAN : mSizeMSS) + ") doesn't match with read byte count(" +
                                                 std::to_string(off + got) + ")");
                    tickets.push(t);
                }
                tReadDone = seconds_since(t0);
            } catch (...) {
                readerError = std::current_exception();
                tickets.push(-1);
            }
        });
        double tCorrDone = 0.0, tAlignDone = 0.0;
        // the product file exists from the start (its blocks are reserved while the strip is still being read); a run that fails
        // takes it away again
        struct ProductGuard {
            std::string path;
            bool done = false;
            oip_file_sink *sink = nullptr;         // the product's prepared file (uncompressed products)
            uint64_t payloadAt = 0;
            ~ProductGuard()
            {
                if (sink) oip_file_sink_close(nullptr, sink);
                if (!done && !path.empty()) ::remove(path.c_str());
            }
        } productGuard;
        const std::string alignedPath = IMO::BuildOutputFilePath(mMssFile, ".ALIGNED", ".TIFF");
        const int comp = tiff_compression(TIFF_LZW);
        std::unique_ptr<TiffWriterU16> tiff(new TiffWriterU16(alignedPath, Wb, outRows, MSS_BANDS, false, comp));   // (declared before the writers: their jobs use it)
        productGuard.path = alignedPath;
        std::unique_ptr<TiffLzwPrep> lzwPrep;              // (declared before the writers too)
        JobThread panWriter, productWriter;
        const size_t productBytes = (size_t)outRows * Wb * MSS_BANDS * 2;
        // (an empty transfer: the writer's download lane pins its two slots now, not when the product is waiting for them)
        productWriter.post([=] { Device::get().check(oip_download_staged_after(ctx, nullptr, nullptr, 0, 0)); });
        if (comp == TIFF_LZW && gpu_lzw()) {
            // likewise for the LZW product: device buffers of the strip encoder, the file reserved at its worst-case size
            lzwPrep.reset(new TiffLzwPrep());
            TiffWriterU16 *twp = tiff.get();
            TiffLzwPrep *lp = lzwPrep.get();
            productWriter.post([=] { lp->prepare(*twp, alignedPath, Wb, outRows, MSS_BANDS, true); });
        }
        if (comp == TIFF_NONE) {
            // the writer thread has nothing to do until the fit is in: it prepares the product file -- header out, blocks
            // reserved, pages mapped and populated (oip_file_sink_open) -- so that the pixels, when they exist, are copied
            // into pages that exist (the allocation is what bounds a buffered write of a new file)
            TiffWriterU16 *twp = tiff.get();
            ProductGuard *pg = &productGuard;
            productWriter.post([=] {
                pg->payloadAt = twp->begin_external_payload();
                Device::get().check(oip_file_sink_open(ctx, alignedPath.c_str(), (size_t)pg->payloadAt + productBytes, &pg->sink));
            });
        }
        struct Joiner {                                    // whatever happens below, the threads are stopped and joined
            std::thread &reader; std::atomic<bool> &cancel;
            ~Joiner() { cancel = true; if (reader.joinable()) reader.join(); }
        } joiner{reader, cancel};

        // while the first section is on its way: the LUTs, and everything the first correlation call would otherwise set up behind
        // the data -- FFT plans and twiddles, the up-sampling operator's tables, the workspace (a call with no units does just that)
        if (o.doRRC4PAN) {
            std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(mRrcPanFile.c_str(), W));
            kbPan.alloc((size_t)W * 2);
            kbPan.upload((double *)prm.get(), (size_t)W * 2);
        }
        if (o.doRRC4MSS) {
            std::vector<double> all((size_t)W * 2);
            for (int i = 0; i < MSS_BANDS; ++i) {
                std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(mRrcMssBndFile[i].c_str(), Wb));
                memcpy(&all[(size_t)i * Wb * 2], prm.get(), sizeof(double) * 2 * Wb);
            }
            kbMss.alloc((size_t)W * 2);
            kbMss.upload(all.data(), (size_t)W * 2);
        }
        ck(oip_interband_correlate_units(ctx, nullptr, nullptr, nullptr, nullptr, 0, baseRows, baseCols, nullptr));
        const double tPrepared = seconds_since(t0);
        const std::string rrcPanPath = IMO::BuildOutputFilePath(mPanFile, ".RRC");
        if (o.writeRrcPan) {                               // sized up front: blocks land at their offsets in any order
            FILE *f = fopen(rrcPanPath.c_str(), "wb");
            if (!f) throw std::runtime_error("open file [" + rrcPanPath + "] failed: " + std::to_string(errno));
            fclose(f);
        }
        if (o.doRRC4PAN) OLOG("Begin inplace RRC for PAN data ... ");
        OLOG("Splitting %d bands of MSS image%s ...", MSS_BANDS, o.doRRC4MSS ? " with inplace RRC" : "");
        OLOG("Calculating inter-band correlation with %d slices in %d section(s) ...", o.slices, o.sections);

        const int n = o.slices * o.sections;
        std::vector<double> table((size_t)MSS_BANDS * n * 4, NAN);
        for (int b = 0; b < MSS_BANDS; ++b)
            for (int u = 0; u < n; ++u) table[((size_t)b * n + u) * 4 + 3] = (double)((u % o.slices) * baseCols + baseCols / 2);   // preproc.h:326
        int unitsDone = 0;
        std::string corrCalls;                             // start+duration of every correlation call, ms (TIMING line)
        for (const Item &it : order) {
            if (it.kind == kMss || it.kind == kPan) {
                const long t = tickets.pop();
                if (t < 0) { reader.join(); std::rethrow_exception(readerError); }
                ck(oip_stage_wait(ctx, t));
                if (it.kind == kMss) {
                    ck(oip_mss_split_rrc_u16(ctx, mMssBil.p + (size_t)it.a * W, mPlanes.p + (size_t)it.a * Wb, mPlaneStride, W, it.b - it.a,
                                             o.doRRC4MSS ? kbMss.p : nullptr));
                } else {
                    uint16_t *blk = mPAN.p + (size_t)it.a * W;
                    if (o.doRRC4PAN) ck(oip_rrc_u16(ctx, blk, blk, W, it.b - it.a, kbPan.p));
                    if (o.writeRrcPan) {
                        long mark = 0;
                        ck(oip_compute_mark(ctx, &mark));
                        const size_t off = (size_t)it.a * lineBytes, nb = (size_t)(it.b - it.a) * lineBytes;
                        panWriter.post([=] { Device::get().check(oip_write_device_to_file_at(ctx, blk, nb, rrcPanPath.c_str(), off, mark)); });
                    }
                }
            } else if (it.kind == kUnits) {
                // the units of the sections resident so far, in the pairs (2k, 2k+1) of the one-call form (a pair that spans two
                // sections -- odd slice counts -- waits for the second)
                const int sec = (int)it.a;
                int hi = (sec + 1) * o.slices;
                if (sec != o.sections - 1) hi &= ~1;
                const int cnt = hi - unitsDone;
                if (cnt <= 0) continue;
                std::vector<const uint16_t *> dPan(cnt), dBands((size_t)cnt * MSS_BANDS);
                std::vector<size_t> panPitch(cnt, (size_t)W), bandPitch(cnt, (size_t)Wb);
                for (int j = 0; j < cnt; ++j) {
                    const int u = unitsDone + j, us = u / o.slices, i = u % o.slices;
                    dPan[j] = mPAN.p + (size_t)p0[us] * W + (size_t)i * baseCols;
                    for (int b = 0; b < MSS_BANDS; ++b) dBands[(size_t)j * MSS_BANDS + b] = mPlanes.p + (size_t)b * mPlaneStride + (size_t)m0[us] * Wb + (size_t)i * bandCols;
                }
                std::vector<double> r((size_t)12 * cnt);
                const double tc0 = seconds_since(t0);
                ck(oip_interband_correlate_units(ctx, dPan.data(), panPitch.data(), dBands.data(), bandPitch.data(), cnt, baseRows, baseCols, r.data()));
                corrCalls += (corrCalls.empty() ? "" : ",") + std::to_string((int)(tc0 * 1e3)) + "+" + std::to_string((int)((seconds_since(t0) - tc0) * 1e3));
                for (int j = 0; j < cnt; ++j)
                    for (int b = 0; b < MSS_BANDS; ++b)
                        for (int k = 0; k < 3; ++k) table[((size_t)b * n + unitsDone + j) * 4 + k] = r[(size_t)12 * j + 3 * b + k];
                unitsDone = hi;
                if (sec == o.sections - 1) tCorrDone = seconds_since(t0);
            } else {
                // every section and every MSS line is in: filter, fit, align, product
                OLOG("Inter-band correlation finished in %.3f seconds, result:", tCorrDone);
                for (int b = 0; b < MSS_BANDS; ++b) {
                    mBandShift[b].resize(n);
                    for (int i = 0; i < n; ++i) {
                        const double *sv = &table[((size_t)b * n + i) * 4];
                        mBandShift[b][i] = InterBandShift{sv[0], sv[1], sv[2], (int)sv[3]};
                    }
                }
                DumpInterBandShiftValues(o.slices, o.sections);
                OLOG("Filter invalid correlation values & try polynomial fitting ...");
                char err[512];
                const int rc = oip_filter_and_fit_mode(table.data(), n, o.threshold, OIP_IBCV_MIN_COUNT, mFitMode, &mDeltaXcoeffs[0][0], &mDeltaYcoeffs[0][0], err, sizeof err);
                if (rc != OIP_OK) { OLOG("%s.", err); throw std::runtime_error(err); }
                for (int b = 0; b < MSS_BANDS; ++b) {
                    OLOG("BAND %d\tdeltaX coeff: [1] %.15f, [0] %.9f", b, mDeltaXcoeffs[b][1], mDeltaXcoeffs[b][0]);
                    OLOG("\tdeltaY coeff: [2] %.15f, [1] %.15f, [0] %.9f", mDeltaYcoeffs[b][2], mDeltaYcoeffs[b][1], mDeltaYcoeffs[b][0]);
                }
                OLOG("CalcInterBandCorrelation(): done.");
                OLOG("Doing inter-band alignment ...");
                long processed = 0;
                ck(oip_align_mss_bicubic_u16x4(ctx, mPlanes.p, mPlaneStride, 0, Lm, out.p, 0, outRows, Wb, Lm, &mDeltaXcoeffs[0][0], &mDeltaYcoeffs[0][0],
                                               o.linesSection, o.lineOffset, o.overlapLines, o.keepLeading ? 1 : 0, OIP_IBPA_MIN_PROCESSLINES, &processed));
                // preproc.h:167-185 WriteAlignedMSS_TIFF: cv::imwrite stores the Mat's channels (c0,c1,c2,c3) as samples (c2,c1,c0,c3);
                // the image takes that order on the device, so its lines go from HBM into the file as they are
                const int fileOrder[4] = {2, 1, 0, 3};
                ck(oip_permute_u16x4(ctx, out.p, (size_t)outRows * Wb, fileOrder));
                long mark = 0;
                ck(oip_compute_mark(ctx, &mark));
                TiffWriterU16 *tw = tiff.get();
                ProductGuard *pg = &productGuard;
                TiffLzwPrep *lp = lzwPrep.get();
                const uint16_t *img = out.p;
                const size_t rowSamples = (size_t)Wb * MSS_BANDS;
                const long rows = outRows;
                productWriter.post([=, &tAlignDone] {
                    Device::get().check(oip_compute_mark_sync(ctx, mark));
                    tAlignDone = seconds_since(t0);
                    OLOG("Alignment done in %.3f seconds (%ld lines valid of %ld).", tAlignDone, processed, rows);
                    OLOG("Outputing aligned TIFF image (%d x %ld x 4) to [%s] ...", Wb, rows, alignedPath.c_str());
                    if (comp == TIFF_NONE) {
                        Device::get().check(oip_file_sink_write(ctx, pg->sink, (size_t)pg->payloadAt, img, (size_t)rows * rowSamples * 2, mark));
                        oip_file_sink *k = pg->sink;
                        pg->sink = nullptr;
                        Device::get().check(oip_file_sink_close(ctx, k));
                        tw->end_external_payload();
                    } else if (gpu_lzw()) {
                        tiff_lzw_from_device(*tw, alignedPath, img, Wb, rows, MSS_BANDS, lp);
                    } else {
                        tiff_rows_from_device(*tw, img, rows, rowSamples, mark);     // download || encode || write
                    }
                    tw->close();
                    OLOG("Output done.");
                });
            }
        }
        reader.join();
        if (readerError) std::rethrow_exception(readerError);
        ck(oip_sync(ctx));
        const double tComputeDone = seconds_since(t0);
        OLOG("%zu bytes read in %.3f seconds (%.1f MBps).", mSizePAN + mSizeMSS, tReadDone, (mSizePAN + mSizeMSS) / tReadDone / 1024.0 / 1024.0);
        if (o.doRRC4PAN) OLOG("RRC for PAN done in %.4f seconds (%.1f MBps).", tComputeDone, mSizePAN / tComputeDone / 1024.0 / 1024.0);
        OLOG("RRC done for MSS bands in %.4f seconds (%.1f MBps).", tComputeDone, mSizeMSS / tComputeDone / 1024.0 / 1024.0);
        panWriter.finish();
        if (o.writeRrcPan) OLOG("Written to file [%s].", rrcPanPath.c_str());
        // the strips are done with: their 9 GB go back while the product is still on its way to the file (releasing them is a
        // good part of what a process of this size pays at exit)
        mMssBil.release();
        mPlanes.release();
        mPAN.release();
        productWriter.finish();
        productGuard.done = true;
        const double tAll = seconds_since(t0);
        OLOG("DoInterBandAlignment(): done.");
        // one line for harnesses (bench.py's `cli` object): seconds since the action started
        double lane[4] = {0, 0, 0, 0};                       // the reader's own time: in pread (all pool threads), waiting for a ring slot's DMA
        oip_stage_stats(ctx, lane, 0);
        RLOG("TIMING default_action pipelined=1 setup=%.4f prepared=%.4f read_done=%.4f correlation_done=%.4f aligned=%.4f compute_done=%.4f products_written=%.4f "
             "device_create=%.4f since_process_start=%.4f reader_in_pread=%.4f reader_waiting_for_slot=%.4f bytes_read=%zu bytes_written=%zu "
             "correlation_calls_ms=%s",
             tSetup, tPrepared, tReadDone, tCorrDone, tAlignDone, tComputeDone, tAll, Device::get().create_seconds(), seconds_since(process_start()), lane[0], lane[1],
             mSizePAN + mSizeMSS, (size_t)outRows * Wb * MSS_BANDS * 2 + (o.writeRrcPan ? mSizePAN : 0), corrCalls.c_str());
    }

    const double *deltaXcoeffs() const { return &mDeltaXcoeffs[0][0]; }
    const double *deltaYcoeffs() const { return &mDeltaYcoeffs[0][0]; }

private:
    void DumpInterBandShiftValues(int slices, int sections)             // preproc.h:470-490
    {
        RLOG("|#SLC|Start|Center| End |   B1.x   |   B2.x   |   B3.x   |   B4.x   |   B1.y   |   B2.y   |   B3.y   |   B4.y   "
             "|   B1.r   |   B2.r   |   B3.r   |   B4.r   |");
        int sliceCols = mW / slices;
        for (int s = 0; s < sections; ++s)
            for (int i = 0; i < slices; ++i) {
                int ii = i + s * slices;
                RLOG("|%4d|%5d|%6d|%5d|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|%10.4f|", i,
                     i * sliceCols, mBandShift[0][ii].cx, (i + 1) * sliceCols, mBandShift[0][ii].dx, mBandShift[1][ii].dx,
                     mBandShift[2][ii].dx, mBandShift[3][ii].dx, mBandShift[0][ii].dy, mBandShift[1][ii].dy, mBandShift[2][ii].dy,
                     mBandShift[3][ii].dy, mBandShift[0][ii].rs, mBandShift[1][ii].rs, mBandShift[2][ii].rs, mBandShift[3][ii].rs);
            }
    }

    void CheckFilesAttributes()                                         // preproc.h:552-572
    {
        OLOG("Checking PAN raw file attributes ...");
        mSizePAN = IMO::FileSize(mPanFile);
        const size_t lineBytes = (size_t)mW * BYTES_PER_PIXEL;
        mLinesPAN = mSizePAN / lineBytes;
        OLOG("Checking MSS raw file attributes ...");
        mSizeMSS = IMO::FileSize(mMssFile);
        mLinesMSS = mSizeMSS / lineBytes;
        if (mSizePAN != MSS_BANDS * mSizeMSS)
            throw std::runtime_error("PAN file size does not match MSS file size: PAN file should be " + std::to_string(MSS_BANDS) +
                                     "x as large as MSS file");
        if (mSizePAN % lineBytes != 0)
            throw std::runtime_error("PAN file size invalid: should be multiplies of " + std::to_string(lineBytes));
        OLOG("CheckFilesAttributes(): OK.");
    }

    const std::string mPanFile, mMssFile, mRrcPanFile;
    std::string mRrcMssBndFile[MSS_BANDS];
    int mW;
    size_t mSizePAN = 0, mSizeMSS = 0, mLinesPAN = 0, mLinesMSS = 0, mPlaneStride = 0;
    DevBuf<uint16_t> mPAN, mMssBil, mPlanes;
    const uint16_t *mPanView = nullptr;
    bool mSplitDone = false;
    int mFitMode = OIP_FIT_REFERENCE;
    std::vector<InterBandShift> mBandShift[MSS_BANDS];
    double mDeltaXcoeffs[MSS_BANDS][2] = {};
    double mDeltaYcoeffs[MSS_BANDS][3] = {};
};

// ---- fused task ---------------------------------------------------------------
// DOC/sample-task.sh runs five commands -- prestitch, stitch (PAN), the default action for each CCD,
// stitch (MSS) -- that hand 6 strip-sized files to one another through the file system.  Here the same
// steps run in one process with every intermediate resident in HBM: four RAW inputs are read once, two
// TIFFs are written.  Each step is the very code path of the stand-alone command (same C-ABI calls, same
// parameters), so the two products are identical to the five-command flow's; tests/test_gpu_cli.py
// compares them.
struct TaskOptions {
    int width = OIP_PIXELS_PER_LINE;
    // prestitch
    int sections = OIP_STT_DEF_SECTIONS, sectionLines = OIP_STT_DEF_SECLINES, overlapCols = OIP_STT_DEF_OVERLAPPX, edgeCols = 0;
    double sttThreshold = OIP_STT_DEF_PHCTHRHLD, sttMaxDeltaY = 0.0;
    // stitching (as given on the command line: halved like main.cpp:189)
    int foldColsPAN = 0, foldColsMSS = 0;
    bool useGDAL = false;
    const int *bandMap = nullptr;
    // default action
    int slices = OIP_IBCV_DEF_SLICES, ibcSections = OIP_IBCV_DEF_SECTIONS, linesSection = OIP_IBPA_DEFAULT_BATCHLINES, lineOffset = 0,
        overlapLines = OIP_IBPA_DEFAULT_LINEOVERLAP;
    double ibcThreshold = OIP_IBCV_DEF_THRESHOLD;
    bool keepLeading = false;
    int fitMode = OIP_FIT_REFERENCE;
    bool fp16acc = false;
    bool panOnly = false;       // stitched PAN product only: fused RRC / resampling straight into the stitched raster
};

inline void RunFusedTask(const std::string &pan1, const std::string &pan2, const std::string &rrc1, const std::string &rrc2,
                         const std::string &mss1, const std::string &mss2, const std::string rrcMss1[MSS_BANDS],
                         const std::string rrcMss2[MSS_BANDS], const std::string &outPAN, const std::string &outMSS,
                         const TaskOptions &o)
{
    oip_ctx *ctx = Device::get().ctx();
    auto ck = [](int rc) { Device::get().check(rc); };
    stop_watch total;
    const int W = o.width;
    // ---- step 1: prestitch.  Sizes and checks as Stitcher's constructor (stitcher.h:49-81)
    const size_t s1 = IMO::FileSize(pan1), s2 = IMO::FileSize(pan2);
    if ((size_t)o.sections * o.sectionLines * BYTES_PER_PIXEL > s1) throw std::invalid_argument("PAN1 size too small for SECTION & LINE_PER_SECTION argument");
    if (s1 != s2) throw std::invalid_argument("PAN1 size doesn't match PAN2 size");
    const long L = (long)(s1 / ((size_t)W * BYTES_PER_PIXEL));
    if (L < (long)o.sections * o.sectionLines)
        throw std::invalid_argument("PAN line count less than sections times line-per-section, use smaller -s and/or -l value(s)");
    const size_t npx = (size_t)W * L;
    DevBuf<uint16_t> p1(npx), p2(npx), p2s;
    p1.load_file(pan1, npx);
    p2.load_file(pan2, npx);
    // CalcSttParameters on the raw strips (App. B-1), same filter and mean as stitcher.h:181-198
    std::vector<double> r(3 * (size_t)o.sections);
    ck(oip_stt_correlate(ctx, p1.p, p2.p, W, L, 0, L, o.sections, o.sectionLines, o.overlapCols, o.edgeCols, r.data()));
    double dx = 0, dy = 0, resp = 0;
    int valid = 0;
    if (oip_stt_mean(r.data(), o.sections, o.sttThreshold, o.sttMaxDeltaY, &dx, &dy, &resp, &valid) != OIP_OK)
        throw std::runtime_error("No valid delta value found for stitching parameter calculating");
    OLOG("Total %d valid delta value pairs found, everage value:", valid);
    OLOG("    dx: %.5f, dy: %.5f, r: %.5f", dx, dy, resp);
    if (o.panOnly) {
        // stitched PAN alone: no corrected strip is needed afterwards, so the left half of every stitched line is the RRC of
        // the raw CCD-1 line (oip_rrc_u16_window) and the right half the resampled CCD-2 line, corrected on load
        // (oip_remap_shift_rrc_bicubic_u16_window): neither .RRC.RAW nor .RRC.PRESTT.RAW is materialised.  Same bits as the flow below.
        const int fold = o.foldColsPAN / 2;
        if (fold < 0 || fold >= W) throw std::invalid_argument("fold columns exceed the image width");
        const long ow = 2L * (W - fold);
        const size_t nout = (size_t)ow * L;
        DevBuf<uint16_t> st(nout);
        DevBuf<double> kb((size_t)W * 2);
        std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(rrc1.c_str(), W));
        kb.upload((double *)prm.get(), (size_t)W * 2);
        ck(oip_rrc_u16_window(ctx, p1.p, W, st.p, ow, W - fold, L, kb.p));
        ck(oip_sync(ctx));
        p1.release();
        prm.reset(IMO::LoadRRCParamFile(rrc2.c_str(), W));
        kb.upload((double *)prm.get(), (size_t)W * 2);
        // the raw CCD-2 samples are corrected on load by the resampling kernel itself (either accumulate mode): one pass over the strip
        ck(oip_remap_shift_rrc_bicubic_u16_window(ctx, p2.p, 0, L, kb.p, st.p, ow, fold, W - fold, 0, L, W, L, dx, dy, OIP_REMAP_SECTION_ROWS,
                                                  OIP_REMAP_ROW_GUARD, o.fp16acc ? 1 : 0));
        OLOG("Write stitched image to file '%s' ...", outPAN.c_str());
        write_tiff_from_device(outPAN, st.p, (int)ow, L, 1, tiff_compression(TIFF_NONE), false);
        OLOG("Fused task (PAN only) done in %.3f seconds.", total.tick());
        return;
    }
    JobThread panWriter;                       // (declared before the buffers its job shares)
    // DoRRC (both strips, in place) + PreStitch of PAN2
    {
        DevBuf<double> kb((size_t)W * 2);
        for (int c = 0; c < 2; ++c) {
            std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile((c ? rrc2 : rrc1).c_str(), W));
            kb.upload((double *)prm.get(), (size_t)W * 2);
            ck(oip_rrc_u16(ctx, c ? p2.p : p1.p, c ? p2.p : p1.p, W, L, kb.p));
            ck(oip_sync(ctx));
        }
    }
    p2s.alloc(npx);
    ck((o.fp16acc ? oip_remap_shift_bicubic_u16_f16acc : oip_remap_shift_bicubic_u16)(ctx, p2.p, 0, L, p2s.p, 0, L, W, L, dx, dy,
                                                                                     OIP_REMAP_SECTION_ROWS, OIP_REMAP_ROW_GUARD));
    p2.release();
    // ---- step 2: stitched PAN product
    {
        const int fold = o.foldColsPAN / 2;
        if (fold < 0 || fold >= W) throw std::invalid_argument("fold columns exceed the image width");
        const size_t nout = (size_t)2 * (W - fold) * L;
        auto st = std::make_shared<DevBuf<uint16_t>>(nout);
        ck(oip_stitch_rows_u16(ctx, p1.p, p2s.p, st->p, W, L, fold));
        long mark = 0;
        ck(oip_compute_mark(ctx, &mark));
        OLOG("Write stitched image to file '%s' ...", outPAN.c_str());
        // the largest product of the task (4 (W - fold) L bytes) goes out on a writer thread, behind the mark of the stitch
        // kernel, while the two default actions below run: a new file takes ~6 GB/s whatever else the process does
        const int comp = tiff_compression(TIFF_NONE);
        panWriter.post([=] { write_tiff_from_device(outPAN, st->p, 2 * (W - fold), L, 1, comp, false, nullptr, mark); });
    }
    // ---- step 3: inter-band alignment per CCD, PAN taken from the device
    DevBuf<uint16_t> aligned[2];
    long arows[2] = {0, 0};
    for (int c = 0; c < 2; ++c) {
        PreProcessor pp(c ? pan2 : pan1, c ? mss2 : mss1, "", c ? rrcMss2 : rrcMss1, W);
        pp.UseDevicePAN(c ? p2s.p : p1.p);
        pp.SetFitMode(o.fitMode);
        pp.LoadMSS();
        pp.DoRRC4MSS(true);
        pp.CalcInterBandCorrelation(o.slices, o.ibcSections, o.ibcThreshold, false);
        pp.DoInterBandAlignment(o.linesSection, o.lineOffset, o.overlapLines, o.keepLeading, true, &aligned[c], &arows[c]);
        (c ? p2s : p1).release();
    }
    // ---- step 4: stitched MSS product (imageop.h:365-457 / :460-567 on the aligned 16UC4 images)
    {
        if (arows[0] != arows[1]) throw std::runtime_error("images have different sizes");
        const int Wb = W / MSS_BANDS, fold = o.foldColsMSS / 2;
        if (fold < 0 || fold >= Wb) throw std::invalid_argument("fold columns exceed the image width");
        const int W4 = Wb * MSS_BANDS, fold4 = fold * MSS_BANDS;
        const size_t nout = (size_t)2 * (W4 - fold4) * arows[0];
        DevBuf<uint16_t> st(nout);
        ck(oip_stitch_rows_u16(ctx, aligned[0].p, aligned[1].p, st.p, W4, arows[0], fold4));
        OLOG("Write stitched image to file '%s' ...", outMSS.c_str());
        const int ow = 2 * (Wb - fold);
        if (!o.useGDAL && nout < 4000000000ull) {
            write_tiff_from_device(outMSS, st.p, ow, arows[0], MSS_BANDS, tiff_compression(TIFF_LZW), true);     // as cv::imwrite of the stitched Mat
        } else {
            int order[4];
            for (int b = 0; b < 4; ++b) order[b] = o.bandMap ? o.bandMap[b] - 1 : b;  // band b <- Mat channel map[b]-1
            write_tiff_from_device(outMSS, st.p, ow, arows[0], MSS_BANDS, tiff_compression(TIFF_LZW), false, order);
        }
    }
    panWriter.finish();
    OLOG("Fused task done in %.3f seconds.", total.tick());
}

}  // namespace OIPGPU
