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
    int active;                                  // lanes of a wave that take a strip
};

struct LzwState {
    unsigned long long *tab;
    unsigned *occ;                               // LDS: this lane's "slot holds an entry of the current generation" bits, word w at occ[w * stride]
    int stride;
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

__device__ __forceinline__ void lzw_clear_occ(LzwState &s)
{
#pragma unroll 8
    for (int w = 0; w < kSlots / 32; ++w) s.occ[w * s.stride] = 0u;
}

// One byte through the coder.  Three steps out of four of sensor data are misses, and a miss that lands on an EMPTY slot --
// most of them: the table is at most half full -- does not need the slot's contents to know it: a bit per slot in LDS (1 KB per
// lane, lane-interleaved: no bank conflicts) says so, and the step costs an LDS read and two stores nobody waits for instead
// of a dependent load from HBM.  Only occupied slots are loaded (hits, and collisions with another string).
__device__ __forceinline__ void lzw_byte(LzwState &s, unsigned c)
{
    const unsigned key = ((unsigned)s.ent << 8) | c;                             // 20 bits
    const unsigned want = (s.gen << 20) | key;
    unsigned h = (key * 2654435761u) >> (32 - 13);
    for (;;) {
        const unsigned bits = s.occ[(h >> 5) * s.stride];
        if (!((bits >> (h & 31)) & 1u)) break;                                   // empty in this generation
        const unsigned long long slot = s.tab[h];
        if ((unsigned)(slot >> 32) == want) { s.ent = (int)(unsigned)slot; return; }
        h = (h + 1) & (kSlots - 1);
    }
    lzw_put(s, (unsigned)s.ent);
    s.ent = (int)c;
    s.tab[h] = ((unsigned long long)want << 32) | (unsigned)s.next;
    s.occ[(h >> 5) * s.stride] |= 1u << (h & 31);
    ++s.next;
    if (s.next == 4094) {                        // table full: clear (libtiff: free_ent == CODE_MAX - 1)
        lzw_put(s, 256u);
        ++s.gen;
        lzw_clear_occ(s);
        s.width = 9;
        s.next = 258;
    } else if (s.next == (1 << s.width) && s.width < 12) {
        ++s.width;
    }
}

__global__ __launch_bounds__(64) void lzw_strips_kernel(LzwJob j)
{
    extern __shared__ unsigned occ[];                                            // kSlots / 32 words per active lane; a workgroup is one wave
    if ((int)threadIdx.x >= j.active) return;
    const long local = (long)blockIdx.x * j.active + threadIdx.x;
    const long k = j.strip0 + local;
    if (k >= j.nstrips) return;
    LzwState s;
    s.tab = j.tab + (size_t)local * kSlots;
    s.occ = occ + threadIdx.x;
    s.stride = j.active;
    lzw_clear_occ(s);
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


// ---- the other direction: the LZW strips of a TIFF file decoded on the device ------------------------------------------------
// cv::imread of the two ALIGNED.TIFF inputs of the MSS stitch (imageop.h:380-388) decodes on the host; here the strips' bytes go
// from the file into HBM as they are (oip_read_file_to_device) and a lane decodes a strip straight into the image.  An LZW
// string is always a substring of what has already been decoded -- entry `next` is the previous string plus the first byte of
// the current one, and those are adjacent in the output -- so a table entry is (position, length) in the lane's own output and
// emitting a code is a forward byte copy inside the image (which also makes the KwKwK case, code == next, nothing special).
// Same acceptance rules as tiffdetail::lzw_decode (oip_tiff.hpp): a stream that ends without EndOfInformation is tolerated, a
// first code >= 256, a code beyond the table or a table that fills without ClearCode are corrupt; a strip must decode to
// exactly its rows.  Predictor 2 is undone by the same lane afterwards (running sums along its rows).
struct LzwDecJob {
    const uint8_t *file;
    const unsigned long long *off, *len;         // [nstrips]: strip offsets inside `file`, byte counts
    long rows;
    int width, spp;
    long rps, nstrips, strip0;
    int predictor;
    uint16_t *img;
    unsigned long long *tab;                     // per lane 4096 x (position | length << 32)
    unsigned *got;                               // [nstrips] bytes a strip decoded to
    int *status;                                 // [nstrips] 0 ok, 1 bad first code, 2 code beyond the table, 3 table full
    int active;                                  // lanes of a wave that take a strip
};

__global__ __launch_bounds__(64) void lzw_decode_kernel(LzwDecJob j)
{
    if ((int)threadIdx.x >= j.active) return;
    const long local = (long)blockIdx.x * j.active + threadIdx.x;
    const long k = j.strip0 + local;
    if (k >= j.nstrips) return;
    unsigned long long *tab = j.tab + (size_t)local * 4096;
    const uint8_t *in = j.file + j.off[k];
    const unsigned n = (unsigned)j.len[k];
    const size_t rowBytes = (size_t)j.width * j.spp * 2;
    const long r0 = k * j.rps;
    long nr = j.rows - r0;
    if (nr > j.rps) nr = j.rps;
    uint8_t *out = reinterpret_cast<uint8_t *>(j.img) + (size_t)r0 * rowBytes;
    const unsigned cap = (unsigned)((size_t)nr * rowBytes);
    int width = 9, next = 258, status = 0;
    bool have_old = false;
    unsigned long long acc = 0;
    int nbits = 0;
    unsigned ip = 0, pos = 0, old_pos = 0, old_len = 0;
    for (;;) {
        while (nbits < width && ip < n) { acc = (acc << 8) | in[ip++]; nbits += 8; }
        if (nbits < width) break;                                              // stream ends without EOI: tolerated, as libtiff does
        const unsigned code = (unsigned)(acc >> (nbits - width)) & ((1u << width) - 1);
        nbits -= width;
        if (code == 257u) break;
        if (code == 256u) { width = 9; next = 258; have_old = false; continue; }
        if (!have_old) {
            if (code >= 256u) { status = 1; break; }
            if (pos < cap) out[pos] = (uint8_t)code;
            old_pos = pos; old_len = 1; ++pos;
            have_old = true;
            continue;
        }
        if (next >= 4096) { status = 3; break; }
        if (code > (unsigned)next) { status = 2; break; }
        const unsigned long long fresh = (unsigned long long)old_pos | ((unsigned long long)(old_len + 1) << 32);
        tab[next] = fresh;                                                     // previous string + this string's first byte
        unsigned clen = 1;
        if (code < 256u) {
            if (pos < cap) out[pos] = (uint8_t)code;
        } else {
            const unsigned long long e = code == (unsigned)next ? fresh : tab[code];
            const unsigned src = (unsigned)e;
            clen = (unsigned)(e >> 32);
            for (unsigned i = 0; i < clen; ++i)
                if (pos + i < cap) out[pos + i] = out[src + i];               // src + i < pos + i: forward copy, overlap included
        }
        old_pos = pos; old_len = clen; pos += clen;
        ++next;
        if (next >= (1 << width) - 1 && width < 12) ++width;                   // early change
        if (pos >= cap && ip >= n) break;
    }
    j.got[k] = pos;
    j.status[k] = status;
    if (status != 0 || pos != cap || j.predictor != 2) return;
    // predictor 2: every sample plus the same channel of the previous pixel, modulo 2^16, row by row
    if (j.spp == 4) {
        for (long r = 0; r < nr; ++r) {
            unsigned long long *row = reinterpret_cast<unsigned long long *>(out + (size_t)r * rowBytes);
            unsigned long long prev = row[0];
            for (int x = 1; x < j.width; ++x) {
                const unsigned long long d = row[x];
                // four 16-bit adds in one register: the low 15 bits of every lane carry on their own, the top bits by xor
                const unsigned long long m = 0x7fff7fff7fff7fffull;
                prev = ((prev & m) + (d & m)) ^ ((prev ^ d) & ~m);
                row[x] = prev;
            }
        }
    } else {
        const int spp = j.spp;
        for (long r = 0; r < nr; ++r) {
            uint16_t *row = reinterpret_cast<uint16_t *>(out + (size_t)r * rowBytes);
            for (long i = spp; i < (long)j.width * spp; ++i) row[i] = (uint16_t)(row[i] + row[i - spp]);
        }
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
        static const char *env_active = getenv("OIP_LZW_LANES");
        // Lanes of a wave that take a strip.  A wave steps at the pace of its slowest lane -- the longest probe chain, a code to
        // flush -- so 64 coders in lockstep are slower than 4 (116 -> 86 ms at 25000 strips, 76 -> 45 ms at 2000: the vector ALU
        // idles either way, and more, emptier waves give the memory system more independent chains to overlap).
        const int active = env_active && atoi(env_active) > 0 && atoi(env_active) <= 64 ? atoi(env_active) : 4;
        LzwJob j{d_img, rows, width, spp, rows_per_strip, nstrips, s0, d_slots, slot_bytes, d_tab, d_len, active};
        // No slot is read before the lane's occupancy bit says it was written in the current generation, so the tables need no
        // initial VALUE -- but memory fresh from hipMalloc is first touched far cheaper by a linear fill than by the kernel's
        // scattered stores (a launch on untouched scratch: 250-350 ms instead of 90-120); tables and output slots alike.
        if (s0 == 0) OIP_LZW_HIP(hipMemsetAsync(d_scratch, 0, o_len, ctx->stream));
        {
            OipProfScope prof(ctx, "lzw_strips_kernel");
            hipLaunchKernelGGL(lzw_strips_kernel, dim3((unsigned)((ns + active - 1) / active)), dim3(64), (size_t)active * (kSlots / 32) * 4, ctx->stream, j);
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

extern "C" int oip_tiff_lzw_decode_u16(oip_ctx *ctx, const uint8_t *d_file, size_t file_bytes, const uint64_t *strip_off,
                                       const uint64_t *strip_len, long nstrips, long rows, int width, int spp, long rows_per_strip,
                                       int predictor, uint16_t *d_img)
{
    OIP_CHECK_CTX(ctx);
    if (!d_file || !strip_off || !strip_len || !d_img || nstrips <= 0 || rows <= 0 || width <= 0 || spp <= 0 || spp > 16 || rows_per_strip <= 0 ||
        (predictor != 1 && predictor != 2))
        return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_decode_u16: bad argument");
    if (nstrips != (rows + rows_per_strip - 1) / rows_per_strip) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_decode_u16: strip count does not match RowsPerStrip");
    const size_t rowBytes = (size_t)width * spp * 2;
    if ((size_t)rows_per_strip * rowBytes >= ((size_t)1 << 32)) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_decode_u16: strip of 4 GiB or more");
    if (spp == 4 && predictor == 2 && (((uintptr_t)d_img) & 7)) return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_decode_u16: image not 8-byte aligned");
    for (long k = 0; k < nstrips; ++k)
        if (strip_off[k] > file_bytes || strip_len[k] > file_bytes - strip_off[k] || strip_len[k] >= ((uint64_t)1 << 32))
            return oip_fail(ctx, OIP_E_INVALID, "oip_tiff_lzw_decode_u16: strip %ld outside the buffer", k);
    const long per = nstrips < kStripsPerLaunch ? nstrips : kStripsPerLaunch;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t o_off = up((size_t)per * 4096 * 8), o_len = o_off + up((size_t)nstrips * 8), o_got = o_len + up((size_t)nstrips * 8),
                 o_status = o_got + up((size_t)nstrips * 4), need = o_status + up((size_t)nstrips * 4);
    void *scratch = nullptr;
    OIP_HIP(ctx, hipMalloc(&scratch, need));
    auto fail = [&](int rc) { (void)hipFree(scratch); return rc; };
    unsigned long long *d_tab = reinterpret_cast<unsigned long long *>(scratch);
    unsigned long long *d_off = reinterpret_cast<unsigned long long *>((char *)scratch + o_off);
    unsigned long long *d_len = reinterpret_cast<unsigned long long *>((char *)scratch + o_len);
    unsigned *d_got = reinterpret_cast<unsigned *>((char *)scratch + o_got);
    int *d_status = reinterpret_cast<int *>((char *)scratch + o_status);
    static_assert(sizeof(uint64_t) == sizeof(unsigned long long), "strip tables are copied as they are");
    if (hipMemcpyAsync(d_off, strip_off, (size_t)nstrips * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_len, strip_len, (size_t)nstrips * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(oip_fail(ctx, OIP_E_DEVICE, "oip_tiff_lzw_decode_u16: copying the strip tables failed"));
    for (long s0 = 0; s0 < nstrips; s0 += per) {
        const long ns = nstrips - s0 < per ? nstrips - s0 : per;
        // (the decoder does not gain from emptier waves as the encoder does: 64 / 16 / 8 / 4 lanes 98 / 102 / 115 / 120 ms)
        static const char *env_active = getenv("OIP_LZW_DECODE_LANES");
        const int active = env_active && atoi(env_active) > 0 && atoi(env_active) <= 64 ? atoi(env_active) : 64;
        LzwDecJob j{d_file, d_off, d_len, rows, width, spp, rows_per_strip, nstrips, s0, predictor, d_img, d_tab, d_got, d_status, active};
        OipProfScope prof(ctx, "lzw_decode_kernel");
        hipLaunchKernelGGL(lzw_decode_kernel, dim3((unsigned)((ns + active - 1) / active)), dim3(64), 0, ctx->stream, j);
    }
    std::vector<unsigned> got((size_t)nstrips);
    std::vector<int> status((size_t)nstrips);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpyAsync(got.data(), d_got, (size_t)nstrips * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(status.data(), d_status, (size_t)nstrips * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess)
        return fail(oip_fail(ctx, OIP_E_DEVICE, "oip_tiff_lzw_decode_u16: decoding failed on the device"));
    (void)hipFree(scratch);
    for (long k = 0; k < nstrips; ++k) {
        const long nr = rows - k * rows_per_strip < rows_per_strip ? rows - k * rows_per_strip : rows_per_strip;
        const size_t want = (size_t)nr * rowBytes;
        if (status[(size_t)k] == 1) return oip_fail(ctx, OIP_E_RUNTIME, "corrupt LZW stream (bad first code) in strip %ld", k);
        if (status[(size_t)k] == 2) return oip_fail(ctx, OIP_E_RUNTIME, "corrupt LZW stream (code beyond the table) in strip %ld", k);
        if (status[(size_t)k] == 3) return oip_fail(ctx, OIP_E_RUNTIME, "corrupt LZW stream (table full without ClearCode) in strip %ld", k);
        if (got[(size_t)k] != want) return oip_fail(ctx, OIP_E_RUNTIME, "strip %ld decodes to %u bytes, %zu expected", k, got[(size_t)k], want);
    }
    return OIP_OK;
}
