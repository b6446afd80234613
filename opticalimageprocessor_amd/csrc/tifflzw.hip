// tifflzw.hip -- the strips of an LZW TIFF product encoded on the GPU (round 4).
//
// The reference writes its aligned MSS product through cv::imwrite (preproc.h:167-185: OpenCV's TIFF encoder, LZW with the
// horizontal predictor) and the stitched MSS through GDAL with COMPRESS=LZW PREDICTOR=2 (imageop.h:460-567).  Both libraries
// encode on the host, strip by strip.  With the image already in HBM and one process per GPU -- a sixteenth of a node's host
// cores each -- that encoder was what the product waited for (0.45-0.5 s for 1.5 GB on 16 cores, csrc/oip_tiff.hpp).  TIFF
// strips are independent LZW streams, so a strip is a lane here: 25000 strips of a 7500 x 25000 x 4 image are 25000 lanes
// that each walk their 60 KB once.
//
// Per lane: the horizontal differences of its rows (predictor 2: sample minus the same channel of the previous pixel, modulo
// 2^16, restarting at every row) as a little-endian byte stream, through the string-table coder TIFF 6.0 section 13
// specifies and libtiff implements (csrc/oip_tiff.hpp::lzw_encode_to is the same coder on the host and the parity check of
// this one: tests/test_gpu_tifflzw.py compares files byte for byte): MSB-first codes of 9..12 bits, ClearCode 256 first,
// EndOfInformation 257 last, first free code 258, the code widens after entry 511 / 1023 / 2047 is assigned, ClearCode when
// entry 4093 is.  The string table is a lane's own 8192 x 8 B open-addressing table in HBM: a slot is
// (generation : 12 | prefix code : 12 | byte : 8) in its high word and the entry's code in its low word; ClearCode starts a new generation instead of wiping 64 KB.
// Every step is a dependent 8-byte load from a 2 GB region: the kernel is bound by HBM latency, not bandwidth -- ~1 us a
// byte, 60 ms for any number of strips up to the 32768 a launch takes -- which is still eight times what the host cores did.
// The encoded strips land in fixed slots (worst case: 1.5 bytes per byte); a second kernel packs them behind one another
// (even offsets, as the host writer places them) so that the payload leaves the device as one block.
#include "oip_internal.h"

#include <cstdint>
#include <vector>

namespace {

constexpr int kSlots = 8192;                     // per-lane table slots (4094 entries at most: load factor < 0.5)
constexpr long kStripsPerLaunch = 32768;

struct LzwJob {
    const uint16_t *img;
    long rows;
    int width, spp;
    long rps;                                    // rows per strip
    long nstrips, strip0;                        // strips of the image, first strip of this launch
    uint8_t *slots;
    size_t slot_bytes;                           // multiple of 4
    unsigned long long *tab;
    unsigned *len;                               // [nstrips]
};

struct LzwState {
    unsigned long long *tab;
    uint8_t *out;
    unsigned long long acc;
    unsigned o;                                  // bytes written
    int nbits, width, next, ent;
    unsigned gen;
};

__device__ __forceinline__ void lzw_put(LzwState &s, unsigned code)
{
    s.acc = (s.acc << s.width) | code;
    s.nbits += s.width;
    if (s.nbits >= 32) {
        const unsigned w = (unsigned)(s.acc >> (s.nbits - 32));
        *reinterpret_cast<unsigned *>(s.out + s.o) = __builtin_bswap32(w);       // MSB first
        s.o += 4;
        s.nbits -= 32;
    }
}

__device__ __forceinline__ void lzw_byte(LzwState &s, unsigned c)
{
    const unsigned key = ((unsigned)s.ent << 8) | c;                             // 20 bits
    const unsigned want = (s.gen << 20) | key;
    unsigned h = (key * 2654435761u) >> (32 - 13);
    for (;;) {
        const unsigned long long slot = s.tab[h];
        const unsigned hi = (unsigned)(slot >> 32);
        if (hi == want) { s.ent = (int)(unsigned)slot; return; }
        if ((hi >> 20) != s.gen) break;                                          // empty in this generation
        h = (h + 1) & (kSlots - 1);
    }
    lzw_put(s, (unsigned)s.ent);
    s.ent = (int)c;
    s.tab[h] = ((unsigned long long)want << 32) | (unsigned)s.next;
    ++s.next;
    if (s.next == 4094) {                        // table full: clear (libtiff: free_ent == CODE_MAX - 1)
        lzw_put(s, 256u);
        ++s.gen;
        s.width = 9;
        s.next = 258;
    } else if (s.next == (1 << s.width) && s.width < 12) {
        ++s.width;
    }
}

__global__ __launch_bounds__(64) void lzw_strips_kernel(LzwJob j)
{
    const long local = (long)blockIdx.x * 64 + threadIdx.x;
    const long k = j.strip0 + local;
    if (k >= j.nstrips) return;
    LzwState s;
    s.tab = j.tab + (size_t)local * kSlots;
    s.out = j.slots + (size_t)local * j.slot_bytes;
    s.acc = 0; s.o = 0; s.nbits = 0; s.width = 9; s.next = 258; s.ent = -1; s.gen = 1;
    lzw_put(s, 256u);
    const long r0 = k * j.rps;
    long r1 = r0 + j.rps;
    if (r1 > j.rows) r1 = j.rows;
    const size_t rowSamples = (size_t)j.width * j.spp;
    // the first byte of the strip only seeds the prefix; every later byte goes through the table
    auto feed = [&](unsigned c) {
        if (s.ent < 0) s.ent = (int)c; else lzw_byte(s, c);
    };
    if (j.spp == 4) {
        for (long r = r0; r < r1; ++r) {
            const unsigned long long *row = reinterpret_cast<const unsigned long long *>(j.img + (size_t)r * rowSamples);
            unsigned long long prev = 0, cur = row[0];
            for (int x = 0; x < j.width; ++x) {
                const unsigned long long nxt = x + 1 < j.width ? row[x + 1] : 0ull;     // in flight under this pixel's 8 bytes
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned d = ((unsigned)(cur >> (16 * c)) - (unsigned)(prev >> (16 * c))) & 0xffffu;
                    feed(d & 0xffu);
                    feed(d >> 8);
                }
                prev = cur;
                cur = nxt;
            }
        }
    } else {                                     // one sample per pixel
        for (long r = r0; r < r1; ++r) {
            const uint16_t *row = j.img + (size_t)r * rowSamples;
            unsigned prev = 0;
            for (int x = 0; x < j.width; ++x) {
                const unsigned cur = row[x];
                const unsigned d = (cur - prev) & 0xffffu;
                feed(d & 0xffu);
                feed(d >> 8);
                prev = cur;
            }
        }
    }
    lzw_put(s, (unsigned)s.ent);
    ++s.next;                                    // libtiff's LZWPostEncode: the last code counts as an entry too
    if (s.next == 4094) { lzw_put(s, 256u); s.width = 9; }
    else if (s.next == (1 << s.width) && s.width < 12) ++s.width;
    lzw_put(s, 257u);
    while (s.nbits >= 8) { s.out[s.o++] = (uint8_t)(s.acc >> (s.nbits - 8)); s.nbits -= 8; }
    if (s.nbits > 0) s.out[s.o++] = (uint8_t)(s.acc << (8 - s.nbits));
    j.len[k] = s.o;
}

// strip k of this launch: slot -> payload + off[k] (even), (len + 1) / 2 two-byte units; an odd length's pad byte is zero
__global__ __launch_bounds__(256) void lzw_pack_kernel(const uint8_t *__restrict__ slots, size_t slot_bytes, const unsigned *__restrict__ len,
                                                       const unsigned long long *__restrict__ off, long strip0, long nstrips,
                                                       uint8_t *__restrict__ payload)
{
    const long k = strip0 + blockIdx.x;
    if (k >= nstrips) return;
    const unsigned n = len[k];
    const uint16_t *src = reinterpret_cast<const uint16_t *>(slots + (size_t)blockIdx.x * slot_bytes);
    uint16_t *dst = reinterpret_cast<uint16_t *>(payload + off[k]);
    const unsigned units = (n + 1) / 2;
    for (unsigned i = threadIdx.x; i < units; i += 256) {
        uint16_t v = src[i];
        if (2 * i + 1 >= n) v &= 0x00ffu;
        dst[i] = v;
    }
}

}  // namespace

extern "C" size_t oip_tiff_lzw_worst_bytes(long rows, int width, int spp, long rows_per_strip)
{
    if (rows <= 0 || width <= 0 || spp <= 0 || rows_per_strip <= 0) return 0;
    const size_t strip = (size_t)rows_per_strip * width * spp * 2;
    const size_t nstrips = ((size_t)rows + rows_per_strip - 1) / rows_per_strip;
    return nstrips * (strip + strip / 2 + strip / 1024 + 64 + 2);
}

// device scratch of one call: the lanes' tables, their output slots, the strip lengths and offsets
static void lzw_scratch_layout(long nstrips, size_t slot_bytes, size_t *tab, size_t *slots, size_t *len, size_t *off, size_t *total)
{
    const long per = nstrips < kStripsPerLaunch ? nstrips : kStripsPerLaunch;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    *tab = 0;
    *slots = up((size_t)per * kSlots * sizeof(unsigned long long));
    *len = *slots + up((size_t)per * slot_bytes);
    *off = *len + up((size_t)nstrips * sizeof(unsigned));
    *total = *off + up((size_t)nstrips * sizeof(unsigned long long));
}

extern "C" size_t oip_tiff_lzw_scratch_bytes(long rows, int width, int spp, long rows_per_strip)
{
    if (rows <= 0 || width <= 0 || spp <= 0 || rows_per_strip <= 0) return 0;
    const long nstrips = (rows + rows_per_strip - 1) / rows_per_strip;
    const size_t strip = (size_t)rows_per_strip * width * spp * 2;
    const size_t slot_bytes = (strip + strip / 2 + strip / 1024 + 64 + 3) / 4 * 4;
    size_t a, b, c, d, total;
    lzw_scratch_layout(nstrips, slot_bytes, &a, &b, &c, &d, &total);
    return total;
}

extern "C" int oip_tiff_lzw_strips_u16(oip_ctx *ctx, const uint16_t *d_img, long rows, int width, int spp, long rows_per_strip,
                                       uint8_t *d_payload, size_t payload_cap, uint64_t *strip_off, uint64_t *strip_len,
                                       size_t *payload_bytes, void *d_scratch, size_t scratch_bytes)
{
    OIP_CHECK_CTX(ctx);
    if (!d_img || !d_payload || !strip_off || !strip_len || !payload_bytes || rows <= 0 || width <= 0 || (spp != 1 && spp != 4) ||
        rows_per_strip <= 0)
        return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_strips_u16: bad argument");
    if (spp == 4 && (((uintptr_t)d_img) & 7)) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_strips_u16: image not 8-byte aligned");
    if ((((uintptr_t)d_payload) & 1)) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_strips_u16: payload not 2-byte aligned");
    const long nstrips = (rows + rows_per_strip - 1) / rows_per_strip;
    const size_t strip = (size_t)rows_per_strip * width * spp * 2;
    if (strip > (1u << 30)) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_strips_u16: strip larger than 1 GiB");
    const size_t slot_bytes = (strip + strip / 2 + strip / 1024 + 64 + 3) / 4 * 4;
    const long per = nstrips < kStripsPerLaunch ? nstrips : kStripsPerLaunch;
    size_t o_tab, o_slots, o_len, o_off, need;
    lzw_scratch_layout(nstrips, slot_bytes, &o_tab, &o_slots, &o_len, &o_off, &need);
    void *own = nullptr;                         // scratch of this call's own when the caller brought none
    auto release = [&] { if (own) (void)hipFree(own); };
#define OIP_LZW_HIP(call)                                                                                     \
    do {                                                                                                      \
        hipError_t e__ = (call);                                                                              \
        if (e__ != hipSuccess) {                                                                              \
            release();                                                                                        \
            return oip_fail(ctx, OIP_E_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
        }                                                                                                     \
    } while (0)
    if (d_scratch) {
        if (scratch_bytes < need || (((uintptr_t)d_scratch) & 7)) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_strips_u16: scratch too small or misaligned");
    } else {
        OIP_LZW_HIP(hipMalloc(&own, need));
        d_scratch = own;
    }
    unsigned long long *d_tab = reinterpret_cast<unsigned long long *>((char *)d_scratch + o_tab);
    uint8_t *d_slots = reinterpret_cast<uint8_t *>((char *)d_scratch + o_slots);
    unsigned *d_len = reinterpret_cast<unsigned *>((char *)d_scratch + o_len);
    unsigned long long *d_off = reinterpret_cast<unsigned long long *>((char *)d_scratch + o_off);
    std::vector<unsigned> len((size_t)nstrips);
    std::vector<unsigned long long> off((size_t)nstrips);
    size_t pos = 0;
    for (long s0 = 0; s0 < nstrips; s0 += per) {
        const long ns = nstrips - s0 < per ? nstrips - s0 : per;
        LzwJob j{d_img, rows, width, spp, rows_per_strip, nstrips, s0, d_slots, slot_bytes, d_tab, d_len};
        // generation 0 is never current: zeroed slots are empty
        OIP_LZW_HIP(hipMemsetAsync(d_tab, 0, (size_t)ns * kSlots * sizeof(unsigned long long), ctx->stream));
        {
            OipProfScope prof(ctx, "lzw_strips_kernel");
            hipLaunchKernelGGL(lzw_strips_kernel, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, ctx->stream, j);
        }
        OIP_LZW_HIP(hipGetLastError());
        OIP_LZW_HIP(hipMemcpyAsync(len.data() + s0, d_len + s0, (size_t)ns * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
        OIP_LZW_HIP(hipStreamSynchronize(ctx->stream));
        for (long k = s0; k < s0 + ns; ++k) {
            if (pos & 1) ++pos;                                                  // strips start on even offsets (oip_tiff.hpp)
            off[(size_t)k] = pos;
            pos += len[(size_t)k];
        }
        const size_t end = (pos + 1) & ~(size_t)1;
        if (end > payload_cap) { release(); return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_strips_u16: payload buffer too small"); }
        OIP_LZW_HIP(hipMemcpyAsync(d_off + s0, off.data() + s0, (size_t)ns * sizeof(unsigned long long), hipMemcpyHostToDevice, ctx->stream));
        {
            OipProfScope prof(ctx, "lzw_pack_kernel");
            hipLaunchKernelGGL(lzw_pack_kernel, dim3((unsigned)ns), dim3(256), 0, ctx->stream, d_slots, slot_bytes, d_len, d_off, s0, nstrips, d_payload);
        }
        OIP_LZW_HIP(hipGetLastError());
        OIP_LZW_HIP(hipStreamSynchronize(ctx->stream));                          // the slots and off[] are reused by the next launch
    }
#undef OIP_LZW_HIP
    release();
    for (long k = 0; k < nstrips; ++k) { strip_off[k] = off[(size_t)k]; strip_len[k] = len[(size_t)k]; }
    *payload_bytes = pos;
    return OIP_OK;
}
