// oip_tiff.hpp -- dependency-free codec for the 16-bit TIFF / BigTIFF files on either side of the hot path.
//
// SURVEY 8f ranks 1 and 2.  The reference writes and reads its TIFFs through two libraries that are not
// available here:
//   * cv::imwrite(".TIFF") of a CV_16UC4 Mat (preproc.h:167-185; StitchTiff, imageop.h:390-402): OpenCV's TIFF
//     encoder defaults to LZW with the horizontal predictor, and converts BGRA -> RGBA on the way out;
//   * GDAL GTiff: stitched PAN as a 1-band file with default creation options (imageop.h:316-328: no
//     compression), stitched MSS with COMPRESS=LZW PREDICTOR=2 (imageop.h:460-567);
//   * cv::imread of those files (imageop.h:380-388).
// This codec writes and reads little-endian baseline strips, chunky samples, unsigned 16 bit, classic TIFF or
// BigTIFF (when the file would pass 4 GiB, as GDAL does with BIGTIFF=IF_NEEDED), uncompressed (259 = 1) or LZW
// (259 = 5) with predictor 1 or 2 (317) -- so the files the reference itself produces can be consumed and the
// files written here open in cv::imread / GDAL.  Pixel payloads match the reference's; file bytes need not
// (strip sizes and tag order are the libraries' own).  Strips are encoded / decoded on a few threads.
//
// LZW as TIFF 6.0 section 13 specifies it and libtiff implements it: MSB-first codes of 9..12 bits, ClearCode
// 256, EndOfInformation 257, first free code 258, every strip starts with ClearCode; the encoder widens the
// code after assigning entry 511 / 1023 / 2047 (the decoder, whose table lags by one entry, widens when ITS
// next free code is 511 / 1023 / 2047 -- TIFF's "early change") and emits ClearCode when entry 4093 is assigned.
// Predictor 2 on 16-bit samples: each sample minus the same channel of the previous pixel, modulo 2^16.
//
// Channel order: OpenCV stores a 4-channel Mat (c0,c1,c2,c3) as samples (c2,c1,c0,c3); `opencv_order`
// reproduces that, so files written here round-trip through cv::imread exactly like the reference's.
#pragma once

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <future>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace OIPGPU {

enum TiffCompression { TIFF_NONE = 1, TIFF_LZW = 5 };

namespace tiffdetail {

// CPUs this process may actually use: the cgroup's CPU quota where there is one (a container that shows all 256 hardware threads
// of the host and grants 16 CPUs of time -- cpu.max = "1600000 100000" -- is the normal case for one rank of a GPU node), else
// the hardware's count.  More runnable threads than the quota only get the whole group throttled.
inline int cpu_budget()
{
    static const int n = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                  // cgroup v2: "<quota|max> <period>"
            char q[32] = {0};
            if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%ld", &period) != 1) period = 0; fclose(h); }
        }
        if (quota > 0 && period > 0) {
            const int c = (int)((quota + period - 1) / period);
            if (c >= 1 && c < hw) hw = c;
        }
        return hw;
    }();
    return n;
}

// threads of the host strip encoder / decoder: the CPUs the process may use, at most 64 (OIP_TIFF_THREADS overrides, up to 128).
// Device-resident products and LZW inputs do not come here: their strips are coded on the device (csrc/tifflzw.hip).
inline int worker_count()
{
    static const int n = [] {
        const char *e = getenv("OIP_TIFF_THREADS");
        int v = e ? atoi(e) : cpu_budget();
        const int cap = e ? 128 : 64;
        return v < 1 ? 1 : (v > cap ? cap : v);
    }();
    return n;
}

// run fn(i, t) for i in [0, n) on a few threads; t < worker_count() is the index of the thread that runs it
template <typename F> inline void parallel_for_t(size_t n, F fn)
{
    const int nt = (int)std::min<size_t>(n, (size_t)worker_count());
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i, 0); return; }
    std::vector<std::thread> th;
    std::vector<std::string> errs(nt);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            try { for (size_t i = t; i < n; i += nt) fn(i, t); }
            catch (const std::exception &e) { errs[t] = e.what(); }
        });
    for (auto &x : th) x.join();
    for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
}
template <typename F> inline void parallel_for(size_t n, F fn) { parallel_for_t(n, [&](size_t i, int) { fn(i); }); }

// The string table of the encoder: open addressing over 16384 slots (4x the 4094 entries), a slot = (generation << 20) |
// (prefix code << 8) | byte.  ClearCode starts a new GENERATION instead of wiping the table -- sensor data barely compresses
// (a code per 1.3 bytes), so the table fills every ~5 KB of input and wiping 64 KB each time cost more than the coding itself.
struct LzwTable {
    static constexpr int kHash = 1 << 14;
    uint32_t key[kHash];
    uint16_t code[kHash];
    uint32_t gen = 0;
    LzwTable() { memset(key, 0, sizeof key); }
    void clear()
    {
        if (++gen == (1u << 12)) { memset(key, 0, sizeof key); gen = 1; }      // generation 0 is never current: zeroed slots are empty
    }
};

// encoded strip: a buffer that is NOT value-initialised (a vector would zero 1.5 x the strip before the encoder overwrites it)
struct LzwStrip {
    std::unique_ptr<uint8_t[]> p;
    size_t n = 0;
    const uint8_t *data() const { return p.get(); }
    size_t size() const { return n; }
};

// worst case of an encoded strip: one 12-bit code per byte, plus Clear / EOI codes
inline size_t lzw_worst(size_t n) { return n + n / 2 + n / 1024 + 64; }

// encodes n bytes into dst (lzw_worst(n) bytes); returns the encoded size
inline size_t lzw_encode_to(const uint8_t *src, size_t n, uint8_t *dst)
{
    static thread_local std::unique_ptr<LzwTable> tab;  // ~96 KB, once per encoding thread
    if (!tab) tab.reset(new LzwTable());
    tab->clear();
    uint8_t *o = dst;
    uint64_t acc = 0;
    int nbits = 0;
    auto put = [&](unsigned code, int width) {
        acc = (acc << width) | code;
        nbits += width;
        while (nbits >= 8) { *o++ = (uint8_t)(acc >> (nbits - 8)); nbits -= 8; }
    };
    int width = 9, next = 258;
    put(256, width);
    if (n == 0) {
        put(257, width);
        if (nbits > 0) *o++ = (uint8_t)(acc << (8 - nbits));
        return (size_t)(o - dst);
    }
    uint32_t gen = tab->gen << 20;
    int ent = src[0];
    for (size_t i = 1; i < n; ++i) {
        const int c = src[i];
        const uint32_t key = gen | ((uint32_t)ent << 8) | (uint32_t)c;
        unsigned h = ((((uint32_t)ent << 8) | (uint32_t)c) * 2654435761u) >> 18;
        bool found = false;
        for (;;) {
            const uint32_t k = tab->key[h];
            if (k == key) { ent = tab->code[h]; found = true; break; }
            if ((k >> 20) != (gen >> 20)) break;                            // empty in this generation
            h = (h + 1) & (LzwTable::kHash - 1);
        }
        if (found) continue;
        put((unsigned)ent, width);
        ent = c;
        tab->key[h] = key;
        tab->code[h] = (uint16_t)next++;
        if (next == 4094) {                        // table full: clear (libtiff: free_ent == CODE_MAX - 1)
            put(256, width);
            tab->clear();
            gen = tab->gen << 20;
            width = 9;
            next = 258;
        } else if (next == (1 << width) && width < 12) {
            ++width;
        }
    }
    put((unsigned)ent, width);
    ++next;                                        // libtiff's LZWPostEncode: the last code counts as an entry too
    if (next == 4094) { put(256, width); width = 9; }
    else if (next == (1 << width) && width < 12) ++width;
    put(257, width);
    if (nbits > 0) *o++ = (uint8_t)(acc << (8 - nbits));
    return (size_t)(o - dst);
}
inline void lzw_encode(const uint8_t *src, size_t n, LzwStrip &out)
{
    out.p.reset(new uint8_t[lzw_worst(n)]);
    out.n = lzw_encode_to(src, n, out.p.get());
}

// returns the number of bytes produced (at most cap); throws on a corrupt stream
inline size_t lzw_decode(const uint8_t *src, size_t n, uint8_t *dst, size_t cap)
{
    uint16_t prefix[4096];
    uint8_t suffix[4096], first[4096];
    uint16_t length[4096];
    for (int i = 0; i < 256; ++i) { prefix[i] = 0xFFFF; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; length[i] = 1; }
    int width = 9, next = 258, old = -1;
    uint64_t acc = 0;
    int nbits = 0;
    size_t pos = 0, produced = 0;
    for (;;) {
        while (nbits < width) {
            if (pos >= n) return produced;          // stream ends without EOI: tolerated, as libtiff does
            acc = (acc << 8) | src[pos++];
            nbits += 8;
        }
        const int code = (int)((acc >> (nbits - width)) & ((1u << width) - 1));
        nbits -= width;
        if (code == 257) break;
        if (code == 256) { width = 9; next = 258; old = -1; continue; }
        if (old < 0) {
            if (code >= 256) throw std::runtime_error("corrupt LZW stream (bad first code)");
            if (produced < cap) dst[produced] = (uint8_t)code;
            ++produced;
            old = code;
            continue;
        }
        // A conforming encoder sends ClearCode when it assigns entry 4093, so the decoder's table never reaches 4096
        // entries; a stream that keeps sending codes without clearing is corrupt (libtiff: "Corrupted LZW table").
        if (next >= 4096) throw std::runtime_error("corrupt LZW stream (table full without ClearCode)");
        int emit = code;
        if (code >= next) {
            if (code != next) throw std::runtime_error("corrupt LZW stream (code beyond the table)");
            // KwKwK: old string + its first byte
            prefix[next] = (uint16_t)old; suffix[next] = first[old]; first[next] = first[old]; length[next] = (uint16_t)(length[old] + 1);
            emit = next;
        } else {
            prefix[next] = (uint16_t)old; suffix[next] = first[code]; first[next] = first[old]; length[next] = (uint16_t)(length[old] + 1);
        }
        const size_t len = length[emit];
        if (produced + len <= cap) {
            uint8_t *p = dst + produced + len;
            for (int c = emit; c != 0xFFFF; c = prefix[c]) *--p = suffix[c];
        } else {
            // tail of a strip that decodes to more than expected: keep what fits
            std::vector<uint8_t> tmp(len);
            uint8_t *p = tmp.data() + len;
            for (int c = emit; c != 0xFFFF; c = prefix[c]) *--p = suffix[c];
            if (produced < cap) memcpy(dst + produced, tmp.data(), cap - produced);
        }
        produced += len;
        ++next;
        if (next >= (1 << width) - 1 && width < 12) ++width;       // early change
        old = code;
        if (produced >= cap && pos >= n) break;
    }
    return produced;
}

inline void predictor2_encode(uint16_t *row, size_t width, int spp)
{
    for (size_t i = width * spp - 1; i >= (size_t)spp; --i) row[i] = (uint16_t)(row[i] - row[i - spp]);
}
inline void predictor2_decode(uint16_t *row, size_t width, int spp)
{
    for (size_t i = spp; i < width * spp; ++i) row[i] = (uint16_t)(row[i] + row[i - spp]);
}

}  // namespace tiffdetail

class TiffWriterU16 {
public:
    // rows are appended top to bottom with write_rows(); close() writes the directory.
    // compression: TIFF_NONE, or TIFF_LZW (always with predictor 2, what cv::imwrite and the reference's GDAL call use)
    TiffWriterU16(const std::string &path, int width, long height, int spp, bool opencv_order, int compression = TIFF_NONE)
        : mW(width), mH(height), mSpp(spp), mSwap(opencv_order && spp == 4), mComp(compression)
    {
        if (width <= 0 || height <= 0 || (spp != 1 && spp != 4)) throw std::invalid_argument("TiffWriterU16: bad geometry");
        if (compression != TIFF_NONE && compression != TIFF_LZW) throw std::invalid_argument("TiffWriterU16: unsupported compression");
        mRowBytes = (size_t)width * spp * 2;
        const size_t data = mRowBytes * (size_t)height;
        // classic TIFF offsets are 32 bit.  LZW can also GROW a strip: at worst one 12-bit code per byte (x1.5), so the
        // choice is made on the worst case and a classic file can never overflow half-way (BigTIFF is always valid).
        const size_t worst = compression == TIFF_LZW ? data + data / 2 : data;
        // Strips: 8 MiB uncompressed (their offsets are arithmetic); LZW: whole rows up to 64 KiB (cv::imwrite and GDAL write 8
        // KB strips; libtiff's table restarts every ~5 KB of sensor data, so the strip size does not change the ratio).  An
        // LZW strip is the unit of parallel work of whoever encodes it: a lane of the device encoder (csrc/tifflzw.hip: 25000
        // strips for a 7500 x 25000 x 4 product), a thread of the host encoder below -- both write the same strips, so a file
        // is the same bytes whoever encoded it.
        mRowsPerStrip = compression == TIFF_LZW ? (long)((64u << 10) / mRowBytes) : (long)((8u << 20) / mRowBytes);
        if (mRowsPerStrip < 1) mRowsPerStrip = 1;
        if (mRowsPerStrip > height) mRowsPerStrip = height;
        const size_t nstrips = ((size_t)height + mRowsPerStrip - 1) / mRowsPerStrip;
        mBig = worst + nstrips * 16 + 16384 > 0xFFFFF000ull;                  // header, page-aligned payload, strip tables, directory
        if (const char *e = getenv("OIP_TIFF_FORCE_BIG")) mBig = mBig || atoi(e) != 0;   // test hook: BigTIFF at any size
        mF = fopen(path.c_str(), "wb");
        if (!mF) throw std::runtime_error("open file [" + path + "] failed");
        if (mBig) {
            const unsigned char h[16] = {'I', 'I', 43, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // directory offset patched in close()
            put(h, 16);
        } else {
            const unsigned char h[8] = {'I', 'I', 42, 0, 0, 0, 0, 0};
            put(h, 8);
        }
        mPos = mBig ? 16 : 8;
    }

    void write_rows(const uint16_t *rows, long count)
    {
        if (mRowsDone + count > mH) throw std::logic_error("TiffWriterU16: too many rows");
        if (mComp == TIFF_NONE) {
            std::vector<uint16_t> tmp;
            for (long r = 0; r < count; ++r) {
                if (mRowsDone % mRowsPerStrip == 0) { mStripOff.push_back(mPos); mStripLen.push_back(0); }
                const uint16_t *src = rows + (size_t)r * mW * mSpp;
                if (mSwap) { tmp.resize((size_t)mW * 4); swap_row(src, tmp.data()); src = tmp.data(); }
                put(src, mRowBytes);
                mStripLen.back() += mRowBytes;
                mPos += mRowBytes;
                ++mRowsDone;
            }
            return;
        }
        // compressed: whole strips are encoded straight from the caller's rows in bounded batches (a few strips per
        // worker thread at a time) and written in order; only the rows of a strip left incomplete by this call are kept
        const size_t rw = (size_t)mW * mSpp;
        mRowsDone += count;
        const bool last = mRowsDone == mH;
        long r = 0;
        if (!mPending.empty()) {
            long have = (long)(mPending.size() / rw);
            const long take = std::min<long>(count, mRowsPerStrip - have);
            mPending.insert(mPending.end(), rows, rows + (size_t)take * rw);
            r = take;
            have += take;
            if (have == mRowsPerStrip || (last && r == count)) {
                encode_strips(mPending.data(), have);
                mPending.clear();
            }
        }
        const long full = (count - r) / mRowsPerStrip;
        // a batch: a few strips per worker thread and at least 64 MiB of rows, so that a batch is worth the threads it starts
        const long batch = std::max<long>(2 * (long)tiffdetail::worker_count(), (long)(((size_t)64 << 20) / (mRowBytes * (size_t)mRowsPerStrip)) + 1);
        for (long k0 = 0; k0 < full; k0 += batch) {
            const long nk = std::min(batch, full - k0);
            encode_strips(rows + (size_t)(r + k0 * mRowsPerStrip) * rw, nk * mRowsPerStrip);
        }
        r += full * mRowsPerStrip;
        if (r < count) {
            if (last) encode_strips(rows + (size_t)r * rw, count - r);           // the last, short strip
            else mPending.insert(mPending.end(), rows + (size_t)r * rw, rows + (size_t)count * rw);
        }
    }

    void close()
    {
        if (!mF) return;
        wait_write();
        if (mRowsDone != mH || !mPending.empty()) { fclose(mF); mF = nullptr; throw std::logic_error("TiffWriterU16: rows missing"); }
        if (mPositioned && fseeko(mF, (off_t)mPos, SEEK_SET) != 0) { fclose(mF); mF = nullptr; throw std::runtime_error("TiffWriterU16: seek failed"); }
        if (mPos & 1) { const unsigned char z = 0; put(&z, 1); ++mPos; }
        // out-of-line arrays first
        const uint64_t nstrips = mStripOff.size();
        uint64_t offStripOff = 0, offStripLen = 0, offBits = 0, offFmt = 0;
        const size_t osz = mBig ? 8 : 4, inl = mBig ? 8 : 4;
        if (!mBig && mPos + nstrips * 8 + 4096 > 0xFFFFFFFFull) { fclose(mF); mF = nullptr; throw std::runtime_error("TiffWriterU16: classic TIFF overflow"); }
        if (nstrips * osz > inl) {
            offStripOff = mPos; for (uint64_t v : mStripOff) putv(v, osz);
            offStripLen = mPos; for (uint64_t v : mStripLen) putv(v, osz);
        }
        if ((size_t)mSpp * 2 > inl) {
            offBits = mPos; for (int i = 0; i < mSpp; ++i) putv(16, 2);
            offFmt = mPos; for (int i = 0; i < mSpp; ++i) putv(1, 2);
        }
        if (mPos & 1) { const unsigned char z = 0; put(&z, 1); ++mPos; }
        const uint64_t ifd = mPos;
        struct Tag { uint16_t id, type; uint64_t count, value; bool is_offset; };
        const uint16_t SHORT = 3, LONG = 4, LONG8 = 16;
        const uint16_t otype = mBig ? LONG8 : LONG;
        std::vector<Tag> tags;
        auto inline_shorts = [&](int n, uint16_t v) { uint64_t x = 0; for (int i = 0; i < n; ++i) x |= (uint64_t)v << (16 * i); return x; };
        tags.push_back({256, LONG, 1, (uint64_t)mW, false});
        tags.push_back({257, LONG, 1, (uint64_t)mH, false});
        tags.push_back({258, SHORT, (uint64_t)mSpp, offBits ? offBits : inline_shorts(mSpp, 16), offBits != 0});
        tags.push_back({259, SHORT, 1, (uint64_t)mComp, false});
        tags.push_back({262, SHORT, 1, (uint64_t)(mSpp == 4 ? 2 : 1), false});            // RGB / BlackIsZero
        tags.push_back({273, otype, nstrips, offStripOff ? offStripOff : mStripOff[0], offStripOff != 0});
        tags.push_back({277, SHORT, 1, (uint64_t)mSpp, false});
        tags.push_back({278, LONG, 1, (uint64_t)mRowsPerStrip, false});
        tags.push_back({279, otype, nstrips, offStripLen ? offStripLen : mStripLen[0], offStripLen != 0});
        tags.push_back({284, SHORT, 1, 1, false});                                        // chunky
        if (mComp == TIFF_LZW) tags.push_back({317, SHORT, 1, 2, false});                 // horizontal predictor
        if (mSpp == 4) tags.push_back({338, SHORT, 1, 2, false});                         // unassociated alpha
        tags.push_back({339, SHORT, (uint64_t)mSpp, offFmt ? offFmt : inline_shorts(mSpp, 1), offFmt != 0});
        if (mBig) {
            putv(tags.size(), 8);
            for (auto &t : tags) { putv(t.id, 2); putv(t.type, 2); putv(t.count, 8); putv(t.value, 8); }
            putv(0, 8);
            fseeko(mF, 8, SEEK_SET);
            putv_raw(ifd, 8);
        } else {
            putv(tags.size(), 2);
            for (auto &t : tags) { putv(t.id, 2); putv(t.type, 2); putv(t.count, 4); putv(t.value, 4); }
            putv(0, 4);
            fseeko(mF, 4, SEEK_SET);
            putv_raw(ifd, 4);
        }
        if (fclose(mF) != 0) { mF = nullptr; throw std::runtime_error("TiffWriterU16: close failed"); }
        mF = nullptr;
    }

    ~TiffWriterU16()
    {
        try { wait_write(); } catch (...) {}
        if (mF) fclose(mF);
    }
    bool bigtiff() const { return mBig; }

    // Reserve the file's blocks ahead of the pixels (uncompressed files: header + payload are known up front).  On tmpfs and
    // on extent file systems allocating the pages is the larger part of a buffered write: a product writer
    // that is idle while the strip is still being read does it then, and the payload write that follows is a copy.  Best effort.
    void preallocate()
    {
        if (mComp != TIFF_NONE || !mF) return;
        fflush(mF);
        (void)posix_fallocate(fileno(mF), 0, (off_t)(mPos + mRowBytes * (uint64_t)mH));
    }

    // Uncompressed files only: the whole pixel payload (height x width x spp samples, already in FILE sample order) is written
    // by someone else at the returned byte offset -- the staging layer, straight from HBM (oip_write_device_to_file_at) --
    // between begin_external_payload() and end_external_payload(); close() then appends the directory.
    uint64_t begin_external_payload()
    {
        if (mComp != TIFF_NONE || mRowsDone != 0) throw std::logic_error("TiffWriterU16: external payload needs an empty uncompressed file");
        if (mSwap) throw std::logic_error("TiffWriterU16: external payload is written in file sample order");
        // right behind the header, where write_rows() would put it: the file is the same bytes either way
        if (fflush(mF) != 0) throw std::runtime_error("TiffWriterU16: write failed");
        return mPos;
    }
    void end_external_payload()
    {
        for (long r = 0; r < mH; r += mRowsPerStrip) {
            const long n = std::min<long>(mRowsPerStrip, mH - r);
            mStripOff.push_back(mPos);
            mStripLen.push_back((uint64_t)n * mRowBytes);
            mPos += (uint64_t)n * mRowBytes;
        }
        mRowsDone = mH;
        if (fseeko(mF, (off_t)mPos, SEEK_SET) != 0) throw std::runtime_error("TiffWriterU16: seek failed");
    }

    // LZW files whose strips were encoded elsewhere (the device: oip_tiff_lzw_strips_u16 on an image in file sample order):
    // the packed strips are written by the caller at the offset begin_external_strips() returns -- where write_rows() would
    // have put the first one -- and end_external_strips() takes their offsets inside that block and their sizes.
    long rows_per_strip() const { return mRowsPerStrip; }
    uint64_t begin_external_strips()
    {
        if (mComp != TIFF_LZW || mRowsDone != 0) throw std::logic_error("TiffWriterU16: external strips need an empty LZW file");
        if (mSwap) throw std::logic_error("TiffWriterU16: external strips are written in file sample order");
        if (fflush(mF) != 0) throw std::runtime_error("TiffWriterU16: write failed");
        return mPos;
    }
    void end_external_strips(const uint64_t *off, const uint64_t *len, size_t n, uint64_t payload_bytes)
    {
        if (n != (size_t)((mH + mRowsPerStrip - 1) / mRowsPerStrip)) throw std::logic_error("TiffWriterU16: strip count");
        if (!mBig && mPos + payload_bytes > 0xFFFFF000ull) throw std::runtime_error("TiffWriterU16: classic TIFF overflow");
        for (size_t k = 0; k < n; ++k) {
            if ((off[k] & 1) || off[k] + len[k] > payload_bytes) throw std::logic_error("TiffWriterU16: strip outside its payload");
            mStripOff.push_back(mPos + off[k]);
            mStripLen.push_back(len[k]);
        }
        mPos += payload_bytes;
        mRowsDone = mH;
        mPositioned = true;
    }

private:
    // encode `nrows` rows (whole strips, the last one possibly short) on a few threads and append them to the file.
    // No allocation per strip: a worker copies + differences a strip in a scratch buffer of its own and encodes it into its
    // region of one of two arenas that alternate between batches (the previous batch is still being written from the other).
    // Per-strip `new` of 1 + 1.5 MB on 32-64 threads made the encoder wait for the kernel's page-fault and unmap paths
    // instead of coding (round 4: the LZW product did not get faster with more threads or smaller strips until this went).
    void encode_strips(const uint16_t *rows, long nrows)
    {
        const size_t rw = (size_t)mW * mSpp;
        const long nstrips = (nrows + mRowsPerStrip - 1) / mRowsPerStrip;
        const size_t stripBytes = (size_t)mRowsPerStrip * rw * 2, worst = tiffdetail::lzw_worst(stripBytes);
        Arena &ar = mArena[mArenaCur];
        mArenaCur ^= 1;
        if (ar.cap < (size_t)nstrips * worst) { ar.p.reset(new uint8_t[(size_t)nstrips * worst]); ar.cap = (size_t)nstrips * worst; }
        if (mScratch.size() < (size_t)tiffdetail::worker_count()) mScratch.resize((size_t)tiffdetail::worker_count());
        std::vector<size_t> len((size_t)nstrips);
        uint8_t *base = ar.p.get();
        tiffdetail::parallel_for_t((size_t)nstrips, [&](size_t k, int t) {
            const long r0 = (long)k * mRowsPerStrip;
            const long n = std::min<long>(mRowsPerStrip, nrows - r0);
            if (!mScratch[(size_t)t]) mScratch[(size_t)t].reset(new uint16_t[(size_t)mRowsPerStrip * rw]);
            uint16_t *buf = mScratch[(size_t)t].get();
            for (long r = 0; r < n; ++r) {
                const uint16_t *src = rows + (size_t)(r0 + r) * rw;
                uint16_t *d = buf + (size_t)r * rw;
                if (mSwap) swap_row(src, d); else memcpy(d, src, rw * 2);
                tiffdetail::predictor2_encode(d, (size_t)mW, mSpp);
            }
            len[k] = tiffdetail::lzw_encode_to((const uint8_t *)buf, (size_t)n * rw * 2, base + k * worst);
        });
        // the batch goes to the file on a thread of its own (positioned writes) while the caller brings down and encodes the
        // next one; the previous batch's write is waited for -- and its error raised -- first
        wait_write();
        std::vector<uint64_t> at((size_t)nstrips);
        for (long k = 0; k < nstrips; ++k) {
            if (mPos & 1) ++mPos;                                       // strips start on even offsets (the gap reads as zero)
            if (!mBig && mPos + len[(size_t)k] > 0xFFFFF000ull) throw std::runtime_error("TiffWriterU16: classic TIFF overflow");
            at[(size_t)k] = mPos;
            mStripOff.push_back(mPos);
            mStripLen.push_back(len[(size_t)k]);
            mPos += len[(size_t)k];
        }
        if (fflush(mF) != 0) throw std::runtime_error("TiffWriterU16: write failed");
        mPositioned = true;
        const int fd = fileno(mF);
        mWrite = std::async(std::launch::async, [fd, base, worst, len, at] {
            for (size_t k = 0; k < len.size(); ++k) {
                const uint8_t *p = base + k * worst;
                size_t n = len[k], w = 0;
                while (w < n) {
                    const ssize_t r = pwrite(fd, p + w, n - w, (off_t)(at[k] + w));
                    if (r < 0 && errno == EINTR) continue;
                    if (r <= 0) throw std::runtime_error("TiffWriterU16: write failed");
                    w += (size_t)r;
                }
            }
        });
    }
    void wait_write() { if (mWrite.valid()) mWrite.get(); }

    void swap_row(const uint16_t *src, uint16_t *dst) const
    {
        for (int x = 0; x < mW; ++x) {
            dst[4 * x + 0] = src[4 * x + 2];
            dst[4 * x + 1] = src[4 * x + 1];
            dst[4 * x + 2] = src[4 * x + 0];
            dst[4 * x + 3] = src[4 * x + 3];
        }
    }
    void put(const void *p, size_t n)
    {
        if (fwrite(p, 1, n, mF) != n) throw std::runtime_error("TiffWriterU16: write failed");
    }
    void putv_raw(uint64_t v, size_t n)
    {
        unsigned char b[8];
        for (size_t i = 0; i < n; ++i) b[i] = (unsigned char)(v >> (8 * i));
        put(b, n);
    }
    void putv(uint64_t v, size_t n) { putv_raw(v, n); mPos += n; }

    FILE *mF = nullptr;
    int mW;
    long mH;
    int mSpp;
    bool mSwap, mBig = false;
    int mComp;
    size_t mRowBytes = 0;
    long mRowsPerStrip = 1, mRowsDone = 0;
    struct Arena { std::unique_ptr<uint8_t[]> p; size_t cap = 0; };
    Arena mArena[2];                                    // encoded strips of the batch being encoded / being written
    int mArenaCur = 0;
    std::vector<std::unique_ptr<uint16_t[]>> mScratch;  // one strip of differenced samples per worker
    uint64_t mPos = 0;
    std::vector<uint64_t> mStripOff, mStripLen;
    std::vector<uint16_t> mPending;
    std::future<void> mWrite;           // the batch of encoded strips being written
    bool mPositioned = false;           // positioned writes have moved past the FILE's own position
};

// Reader for little-endian, chunky, unsigned 16-bit strip TIFF / BigTIFF, uncompressed or LZW (predictor 1 or
// 2): what the writer above, cv::imwrite and GDAL's GTiff driver (INTERLEAVE=PIXEL, untiled) produce.  Samples
// are returned in file order.  Every header field is checked before it sizes an allocation or a read.
// layout_only: the checked header alone -- geometry, encoding and strip table -- for a caller that brings the strips in itself
// (the device decoder: oip_host.hpp::read_tiff_to_device)
struct TiffLayout {
    uint64_t width = 0, height = 0, spp = 0, compression = 1, predictor = 1, rows_per_strip = 0;
    std::vector<uint64_t> offs, lens;
};
inline void read_tiff_u16(const std::string &path, int *width, long *height, int *spp, std::vector<uint16_t> *out,
                          TiffLayout *layout_only = nullptr)
{
    struct FileGuard {
        FILE *f;
        ~FileGuard() { if (f) fclose(f); }
    } g{fopen(path.c_str(), "rb")};
    FILE *f = g.f;
    if (!f) throw std::runtime_error("cannot open file [" + path + "]");
    auto fail = [&](const std::string &m) -> void { throw std::runtime_error("read TIFF [" + path + "]: " + m); };
    if (fseeko(f, 0, SEEK_END)) fail("seek failed");
    const uint64_t fsize = (uint64_t)ftello(f);
    auto rd = [&](uint64_t off, void *p, size_t n) {
        if (off > fsize || n > fsize - off) fail("truncated file");
        if (fseeko(f, (off_t)off, SEEK_SET) || fread(p, 1, n, f) != n) fail("truncated file");
    };
    unsigned char h[16];
    rd(0, h, 8);
    if (h[0] != 'I' || h[1] != 'I') fail("only little-endian TIFF is supported");
    const bool big = h[2] == 43;
    if ((h[2] != 42 && !big) || h[3] != 0) fail("not a TIFF file");
    uint64_t ifd;
    if (big) { rd(8, h, 8); memcpy(&ifd, h, 8); } else { uint32_t v; memcpy(&v, h + 4, 4); ifd = v; }
    uint64_t n = 0;
    if (big) rd(ifd, &n, 8); else { uint16_t v; rd(ifd, &v, 2); n = v; }
    if (n == 0 || n > 4096) fail("implausible directory");
    const uint64_t ent = ifd + (big ? 8 : 2);
    const size_t esz = big ? 20 : 12, osz = big ? 8 : 4;
    auto tsize = [](int t) { return t == 3 ? 2 : (t == 4 ? 4 : (t == 16 ? 8 : 0)); };
    uint64_t W = 0, H = 0, comp = 1, planar = 1, S = 1, bits = 0, fmt = 1, pred = 1, rps = 0;
    std::vector<uint64_t> offs, lens;
    for (uint64_t i = 0; i < n; ++i) {
        unsigned char e[20];
        rd(ent + i * esz, e, esz);
        uint16_t id, type;
        memcpy(&id, e, 2); memcpy(&type, e + 2, 2);
        uint64_t cnt = 0;
        memcpy(&cnt, e + 4, osz);
        const int ts = tsize(type);
        if (!ts || cnt == 0) continue;                                          // unknown type / empty tag: not ours
        if (cnt > fsize / (uint64_t)ts) fail("tag larger than the file");
        std::vector<unsigned char> raw((size_t)cnt * ts);
        if (cnt * ts <= osz) memcpy(raw.data(), e + 4 + osz, (size_t)cnt * ts);
        else { uint64_t o = 0; memcpy(&o, e + 4 + osz, osz); rd(o, raw.data(), raw.size()); }
        auto val = [&](uint64_t k) { uint64_t v = 0; memcpy(&v, raw.data() + k * ts, ts); return v; };
        switch (id) {
            case 256: W = val(0); break;
            case 257: H = val(0); break;
            case 258: bits = val(0); for (uint64_t k = 1; k < cnt; ++k) if (val(k) != bits) fail("mixed sample depths"); break;
            case 259: comp = val(0); break;
            case 273: offs.resize(cnt); for (uint64_t k = 0; k < cnt; ++k) offs[k] = val(k); break;
            case 277: S = val(0); break;
            case 278: rps = val(0); break;
            case 279: lens.resize(cnt); for (uint64_t k = 0; k < cnt; ++k) lens[k] = val(k); break;
            case 284: planar = val(0); break;
            case 317: pred = val(0); break;
            case 339: fmt = val(0); break;
            case 322: case 323: case 324: case 325: fail("tiled TIFF is not supported (strips only)"); break;
            default: break;
        }
    }
    if (comp != TIFF_NONE && comp != TIFF_LZW) fail("compression " + std::to_string(comp) + " is not supported (none and LZW are)");
    if (pred != 1 && pred != 2) fail("predictor " + std::to_string(pred) + " is not supported");
    if (bits != 16 || fmt != 1 || planar != 1) fail("only chunky unsigned 16-bit samples are supported");
    if (!W || !H || W > 0x7FFFFFFFull || H > 0x7FFFFFFFull || S == 0 || S > 16) fail("missing or implausible geometry");
    if (offs.empty() || offs.size() != lens.size()) fail("missing strip tags");
    const uint64_t row_bytes = W * S * 2;
    if (H > (~(uint64_t)0) / row_bytes || row_bytes * H > ((uint64_t)1 << 46)) fail("implausible image size");
    if (rps == 0 || rps > H) rps = H;
    const uint64_t nstrips = (H + rps - 1) / rps;
    if (offs.size() != nstrips) fail("strip count does not match RowsPerStrip");
    for (size_t k = 0; k < offs.size(); ++k)
        if (offs[k] > fsize || lens[k] > fsize - offs[k]) fail("strip outside the file");
    if (comp == TIFF_NONE) {
        uint64_t total = 0;
        for (uint64_t l : lens) { if (l > row_bytes * H - total) fail("strip sizes exceed the image"); total += l; }
        if (total != row_bytes * H) fail("strip sizes do not cover the image");
    }
    if (comp == TIFF_LZW) {
        // header fields alone must not size the allocation: a 12-bit code (1.5 bytes) expands to at most 4096 - 258
        // bytes, so the strips present bound what the image can decode to
        uint64_t packed = 0;
        for (uint64_t l : lens) packed += l;
        if (row_bytes * H / 2560 > packed + 16) fail("image size is implausible for its compressed strips");
    }
    if (layout_only) {
        layout_only->width = W; layout_only->height = H; layout_only->spp = S; layout_only->compression = comp;
        layout_only->predictor = pred; layout_only->rows_per_strip = rps;
        layout_only->offs = offs; layout_only->lens = lens;
        *width = (int)W; *height = (long)H; *spp = (int)S;
        return;
    }
    try {
        out->resize((size_t)(W * H * S));
    } catch (const std::bad_alloc &) {
        fail("not enough memory for a " + std::to_string(W) + " x " + std::to_string(H) + " x " + std::to_string(S) + " image");
    }
    unsigned char *dstb = reinterpret_cast<unsigned char *>(out->data());
    if (comp == TIFF_NONE) {
        size_t pos = 0;
        for (size_t k = 0; k < offs.size(); ++k) {
            rd(offs[k], dstb + pos, (size_t)lens[k]);
            pos += (size_t)lens[k];
        }
    } else {
        // read the compressed strips in batches, decode them on a few threads
        const size_t batch = 64;
        for (size_t k0 = 0; k0 < offs.size(); k0 += batch) {
            const size_t kn = std::min(batch, offs.size() - k0);
            std::vector<std::vector<unsigned char>> raw(kn);
            for (size_t j = 0; j < kn; ++j) { raw[j].resize((size_t)lens[k0 + j]); rd(offs[k0 + j], raw[j].data(), raw[j].size()); }
            tiffdetail::parallel_for(kn, [&](size_t j) {
                const uint64_t k = k0 + j;
                const uint64_t r0 = k * rps, nr = std::min<uint64_t>(rps, H - r0);
                const size_t want = (size_t)(nr * row_bytes);
                const size_t got = tiffdetail::lzw_decode(raw[j].data(), raw[j].size(), dstb + (size_t)(r0 * row_bytes), want);
                if (got != want) throw std::runtime_error("read TIFF [" + path + "]: strip " + std::to_string(k) + " decodes to " +
                                                          std::to_string(got) + " bytes, " + std::to_string(want) + " expected");
            });
        }
    }
    if (pred == 2) {
        uint16_t *px = out->data();
        tiffdetail::parallel_for((size_t)H, [&](size_t r) { tiffdetail::predictor2_decode(px + r * (size_t)(W * S), (size_t)W, (int)S); });
    }
    *width = (int)W; *height = (long)H; *spp = (int)S;
}

// like the writer class, but with an explicit sample order: out sample i = in sample order[i]
inline void write_tiff_u16_mapped(const std::string &path, const uint16_t *data, int width, long height, const int order[4],
                                  int compression = TIFF_NONE)
{
    TiffWriterU16 w(path, width, height, 4, false, compression);
    const long chunk = std::max<long>(1, (long)(((size_t)32 << 20) / ((size_t)width * 8)));
    std::vector<uint16_t> rows((size_t)chunk * width * 4);
    for (long r0 = 0; r0 < height; r0 += chunk) {
        const long n = std::min(chunk, height - r0);
        for (long r = 0; r < n; ++r) {
            const uint16_t *src = data + (size_t)(r0 + r) * width * 4;
            uint16_t *dst = rows.data() + (size_t)r * width * 4;
            for (int x = 0; x < width; ++x)
                for (int c = 0; c < 4; ++c) dst[4 * (size_t)x + c] = src[4 * (size_t)x + order[c]];
        }
        w.write_rows(rows.data(), n);
    }
    w.close();
}

inline void write_tiff_u16(const std::string &path, const uint16_t *data, int width, long height, int spp, bool opencv_order,
                           int compression = TIFF_NONE)
{
    TiffWriterU16 w(path, width, height, spp, opencv_order, compression);
    w.write_rows(data, height);             // (batches its strips itself: 2 per encoder thread at a time)
    w.close();
}

}  // namespace OIPGPU
