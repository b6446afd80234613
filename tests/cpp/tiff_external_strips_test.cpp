// tiff_external_strips_test.cpp -- TiffWriterU16's external-strips interface (what the device LZW encoder feeds,
// csrc/tifflzw.hip) against its own write_rows(): the same strips, encoded here with the host coder and packed the way
// oip_tiff_lzw_strips_u16 packs them (strip order, even offsets, zero pad), must give the same file, byte for byte.
// Built with ASan + UBSan by tests/test_cli_cpu.py.  usage: tiff_external_strips_test DIR
#include "oip_tiff.hpp"

#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

using namespace OIPGPU;

static std::vector<uint8_t> slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    int bad = 0, n = 0;
    const int cases[][3] = {{7, 5, 1}, {720, 300, 4}, {7500, 9, 4}, {33, 4, 1}, {2048, 70, 1}, {16384, 3, 4}, {64, 150, 4}};
    for (auto &c : cases) {
        const int w = c[0], spp = c[2];
        const long h = c[1];
        std::vector<uint16_t> img((size_t)w * h * spp);
        for (size_t i = 0; i < img.size(); ++i) img[i] = (uint16_t)(((i * 2654435761u) >> 9) & (i % 7 ? 0x0fff : 0xffff));
        const std::string a = dir + "/a.tiff", b = dir + "/b.tiff";
        {
            TiffWriterU16 tw(a, w, h, spp, false, TIFF_LZW);
            tw.write_rows(img.data(), h);
            tw.close();
        }
        {
            TiffWriterU16 tw(b, w, h, spp, false, TIFF_LZW);
            const long rps = tw.rows_per_strip();
            const size_t rw = (size_t)w * spp, nstrips = (size_t)((h + rps - 1) / rps);
            std::vector<uint64_t> off(nstrips), len(nstrips);
            std::vector<uint8_t> payload;
            for (size_t k = 0; k < nstrips; ++k) {
                const long r0 = (long)k * rps, nr = std::min<long>(rps, h - r0);
                std::vector<uint16_t> rows(img.begin() + (size_t)r0 * rw, img.begin() + (size_t)(r0 + nr) * rw);
                for (long r = 0; r < nr; ++r) tiffdetail::predictor2_encode(rows.data() + (size_t)r * rw, (size_t)w, spp);
                std::vector<uint8_t> enc(tiffdetail::lzw_worst((size_t)nr * rw * 2));
                const size_t m = tiffdetail::lzw_encode_to((const uint8_t *)rows.data(), (size_t)nr * rw * 2, enc.data());
                if (payload.size() & 1) payload.push_back(0);
                off[k] = payload.size();
                len[k] = m;
                payload.insert(payload.end(), enc.begin(), enc.begin() + (long)m);
            }
            const uint64_t bytes = payload.size();
            if (payload.size() & 1) payload.push_back(0);
            const uint64_t at = tw.begin_external_strips();
            FILE *f = fopen(b.c_str(), "r+b");
            if (!f || fseeko(f, (off_t)at, SEEK_SET) != 0 || fwrite(payload.data(), 1, payload.size(), f) != payload.size()) { printf("io error\n"); return 2; }
            fclose(f);
            tw.end_external_strips(off.data(), len.data(), nstrips, bytes);
            tw.close();
        }
        const auto fa = slurp(a), fb = slurp(b);
        ++n;
        if (fa.size() < 100 || fa != fb) { ++bad; printf("case %d x %ld x %d: files differ (%zu / %zu bytes)\n", w, h, spp, fa.size(), fb.size()); }
        // and the file reads back
        int rw_ = 0, rs = 0; long rh = 0;
        std::vector<uint16_t> back;
        read_tiff_u16(b, &rw_, &rh, &rs, &back);
        if (rw_ != w || rh != h || rs != spp || back != img) { ++bad; printf("case %d x %ld x %d: read-back differs\n", w, h, spp); }
    }
    // misuse is refused
    try { TiffWriterU16 tw(dir + "/c.tiff", 8, 2, 1, false, TIFF_NONE); tw.begin_external_strips(); ++bad; printf("uncompressed writer took external strips\n"); }
    catch (const std::logic_error &) {}
    try {
        TiffWriterU16 tw(dir + "/c.tiff", 8, 2, 1, false, TIFF_LZW);
        tw.begin_external_strips();
        uint64_t off[1] = {1}, len[1] = {4};
        tw.end_external_strips(off, len, 1, 8);
        ++bad; printf("odd strip offset accepted\n");
    } catch (const std::logic_error &) {}
    printf("%d cases, %d bad\n", n, bad);
    return bad ? 1 : 0;
}
