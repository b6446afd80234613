// oip_tiff.hpp -- dependency-free writer for uncompressed 16-bit TIFF / BigTIFF.
//
// SURVEY 8f rank 1.  The reference writes its results through two libraries that are not
// available here: the stitched PAN strip as a 1-band GTiff via GDAL (imageop.h:316-328,
// GDT_UInt16, default creation options = uncompressed, one strip per row block) and the aligned
// MSS image via cv::imwrite(".TIFF") of a CV_16UC4 Mat (preproc.h:167-185).  This writer produces
// files every TIFF reader opens (little endian, baseline tags, contiguous strips; BigTIFF when the
// file would pass 4 GiB, as GDAL does with BIGTIFF=IF_NEEDED) -- not byte-identical files: OpenCV
// additionally LZW-compresses, which no consumer depends on.
//
// Channel order: OpenCV's TIFF encoder converts BGRA -> RGBA on write (and imread converts back),
// so a 4-channel Mat (c0,c1,c2,c3) is stored as samples (c2,c1,c0,c3).  `opencv_order` reproduces
// that, so files written here round-trip through cv::imread exactly like the reference's.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace OIPGPU {

class TiffWriterU16 {
public:
    // rows are appended top to bottom with write_rows(); close() writes the directory
    TiffWriterU16(const std::string &path, int width, long height, int spp, bool opencv_order)
        : mW(width), mH(height), mSpp(spp), mSwap(opencv_order && spp == 4)
    {
        if (width <= 0 || height <= 0 || (spp != 1 && spp != 4)) throw std::invalid_argument("TiffWriterU16: bad geometry");
        mRowBytes = (size_t)width * spp * 2;
        const size_t data = mRowBytes * (size_t)height;
        mBig = data + (size_t)height * 16 / 64 + 4096 > 0xFFFFF000ull;      // classic TIFF offsets are 32 bit
        mRowsPerStrip = (long)((8u << 20) / mRowBytes);
        if (mRowsPerStrip < 1) mRowsPerStrip = 1;
        if (mRowsPerStrip > height) mRowsPerStrip = height;
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
        std::vector<uint16_t> tmp;
        for (long r = 0; r < count; ++r) {
            if (mRowsDone % mRowsPerStrip == 0) { mStripOff.push_back(mPos); mStripLen.push_back(0); }
            const uint16_t *src = rows + (size_t)r * mW * mSpp;
            if (mSwap) {
                tmp.resize((size_t)mW * 4);
                for (int x = 0; x < mW; ++x) {
                    tmp[4 * x + 0] = src[4 * x + 2];
                    tmp[4 * x + 1] = src[4 * x + 1];
                    tmp[4 * x + 2] = src[4 * x + 0];
                    tmp[4 * x + 3] = src[4 * x + 3];
                }
                src = tmp.data();
            }
            put(src, mRowBytes);
            mStripLen.back() += mRowBytes;
            mPos += mRowBytes;
            ++mRowsDone;
        }
    }

    void close()
    {
        if (!mF) return;
        if (mRowsDone != mH) { fclose(mF); mF = nullptr; throw std::logic_error("TiffWriterU16: rows missing"); }
        if (mPos & 1) { const unsigned char z = 0; put(&z, 1); ++mPos; }
        // out-of-line arrays first
        const uint64_t nstrips = mStripOff.size();
        uint64_t offStripOff = 0, offStripLen = 0, offBits = 0, offFmt = 0;
        const size_t osz = mBig ? 8 : 4, inl = mBig ? 8 : 4;
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
        tags.push_back({259, SHORT, 1, 1, false});                                        // no compression
        tags.push_back({262, SHORT, 1, (uint64_t)(mSpp == 4 ? 2 : 1), false});            // RGB / BlackIsZero
        tags.push_back({273, otype, nstrips, offStripOff ? offStripOff : mStripOff[0], offStripOff != 0});
        tags.push_back({277, SHORT, 1, (uint64_t)mSpp, false});
        tags.push_back({278, LONG, 1, (uint64_t)mRowsPerStrip, false});
        tags.push_back({279, otype, nstrips, offStripLen ? offStripLen : mStripLen[0], offStripLen != 0});
        tags.push_back({284, SHORT, 1, 1, false});                                        // chunky
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

    ~TiffWriterU16() { if (mF) fclose(mF); }
    bool bigtiff() const { return mBig; }

private:
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
    size_t mRowBytes = 0;
    long mRowsPerStrip = 1, mRowsDone = 0;
    uint64_t mPos = 0;
    std::vector<uint64_t> mStripOff, mStripLen;
};

// Reader for what the writer above produces (and any other little-endian, uncompressed, chunky
// 16-bit strip TIFF / BigTIFF).  Compressed files -- e.g. cv::imwrite's default LZW -- are refused
// with a clear message rather than decoded.  Samples are returned in file order.
inline void read_tiff_u16(const std::string &path, int *width, long *height, int *spp, std::vector<uint16_t> *out)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open file [" + path + "]");
    auto fail = [&](const std::string &m) { fclose(f); throw std::runtime_error("read TIFF [" + path + "]: " + m); };
    auto rd = [&](uint64_t off, void *p, size_t n) { if (fseeko(f, (off_t)off, SEEK_SET) || fread(p, 1, n, f) != n) fail("truncated file"); };
    unsigned char h[16];
    rd(0, h, 8);
    if (h[0] != 'I' || h[1] != 'I') fail("only little-endian TIFF is supported");
    const bool big = h[2] == 43;
    if (h[2] != 42 && !big) fail("not a TIFF file");
    uint64_t ifd;
    if (big) { rd(8, h, 8); memcpy(&ifd, h, 8); } else { uint32_t v; memcpy(&v, h + 4, 4); ifd = v; }
    uint64_t n = 0;
    if (big) rd(ifd, &n, 8); else { uint16_t v; rd(ifd, &v, 2); n = v; }
    const uint64_t ent = ifd + (big ? 8 : 2);
    const size_t esz = big ? 20 : 12, osz = big ? 8 : 4;
    auto tsize = [](int t) { return t == 3 ? 2 : (t == 4 ? 4 : (t == 16 ? 8 : 0)); };
    uint64_t W = 0, H = 0, comp = 1, planar = 1, S = 1, bits = 0, fmt = 1;
    std::vector<uint64_t> offs, lens;
    for (uint64_t i = 0; i < n; ++i) {
        unsigned char e[20];
        rd(ent + i * esz, e, esz);
        uint16_t id, type;
        memcpy(&id, e, 2); memcpy(&type, e + 2, 2);
        uint64_t cnt = 0;
        memcpy(&cnt, e + 4, osz);
        const int ts = tsize(type);
        if (!ts) continue;
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
            case 279: lens.resize(cnt); for (uint64_t k = 0; k < cnt; ++k) lens[k] = val(k); break;
            case 284: planar = val(0); break;
            case 339: fmt = val(0); break;
            default: break;
        }
    }
    if (comp != 1) fail("compressed TIFF input (compression " + std::to_string(comp) + ") is not supported by this build; re-save uncompressed");
    if (bits != 16 || fmt != 1 || planar != 1) fail("only chunky unsigned 16-bit samples are supported");
    if (!W || !H || offs.empty() || offs.size() != lens.size()) fail("missing geometry or strip tags");
    out->resize((size_t)W * H * S);
    size_t pos = 0;
    for (size_t k = 0; k < offs.size(); ++k) {
        if (pos + lens[k] > out->size() * 2) fail("strip sizes exceed the image");
        rd(offs[k], (char *)out->data() + pos, (size_t)lens[k]);
        pos += (size_t)lens[k];
    }
    if (pos != out->size() * 2) fail("strip sizes do not cover the image");
    fclose(f);
    *width = (int)W; *height = (long)H; *spp = (int)S;
}

// like the writer class, but with an explicit sample order: out sample i = in sample order[i]
inline void write_tiff_u16_mapped(const std::string &path, const uint16_t *data, int width, long height, const int order[4])
{
    TiffWriterU16 w(path, width, height, 4, false);
    std::vector<uint16_t> row((size_t)width * 4);
    for (long r = 0; r < height; ++r) {
        const uint16_t *src = data + (size_t)r * width * 4;
        for (int x = 0; x < width; ++x)
            for (int c = 0; c < 4; ++c) row[4 * (size_t)x + c] = src[4 * (size_t)x + order[c]];
        w.write_rows(row.data(), 1);
    }
    w.close();
}

inline void write_tiff_u16(const std::string &path, const uint16_t *data, int width, long height, int spp, bool opencv_order)
{
    TiffWriterU16 w(path, width, height, spp, opencv_order);
    const long chunk = 4096;
    for (long r = 0; r < height; r += chunk)
        w.write_rows(data + (size_t)r * width * spp, height - r < chunk ? height - r : chunk);
    w.close();
}

}  // namespace OIPGPU
