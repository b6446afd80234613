// auxmerge.hip -- sub-image merge + byte-order pass of the down-link de-framer on gfx950.
//
// Replaces, for UNCOMPRESSED frames, the data movement of AuxSeparator::WriteImageData
// (aux_separator.h:341-364): every frame carries (4 PAN + 1 MSS) x 8 sub-images of 256 x 1536 big-endian
// u16; InflateSubImage (:374-393, ratio IMGSIG_ZRTO_NONE) copies a sub-image and swaps the bytes of every
// word, MergeSubImage (:366-372) copies its rows side by side into a 12288-pixel stripe that is then
// written to the PAN or MSS RAW file.  Here the whole frame payload is one streaming pass:
//     out[(r * sub_lines + line) * (hparts * sub_cols) + c * sub_cols + col] = bswap16(tile[r * hparts + c][line][col])
// 2 B read + 2 B written per pixel.  The framing itself (sync-word scan, CRC, IMTR re-framing) and JPEG 2000
// decoding stay on the host: byte-serial parsing, out of scope.
#include "oip_internal.h"

namespace {

constexpr int kBlock = 256;

// 8 pixels per lane: 16-byte load and store, one v_perm_b32 per pixel pair
__global__ __launch_bounds__(kBlock) void merge_be16_kernel(const uint4 *__restrict__ tiles, uint4 *__restrict__ out, int hparts,
                                                            int sub_lines, int chunks_per_tile_row, long total_chunks)
{
    const long stride = (long)gridDim.x * kBlock;
    const int row_chunks = hparts * chunks_per_tile_row;                 // chunks per output line
    const long tile_chunks = (long)sub_lines * chunks_per_tile_row;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total_chunks; i += stride) {
        const long row = i / row_chunks;                                // output line over all stripes
        const int cc = (int)(i - row * row_chunks);
        const int c = cc / chunks_per_tile_row, k = cc - c * chunks_per_tile_row;
        const long r = row / sub_lines;
        const int line = (int)(row - r * sub_lines);
        uint4 v = tiles[(r * hparts + c) * tile_chunks + (long)line * chunks_per_tile_row + k];
        v.x = __builtin_amdgcn_perm(v.x, v.x, 0x02030001u);
        v.y = __builtin_amdgcn_perm(v.y, v.y, 0x02030001u);
        v.z = __builtin_amdgcn_perm(v.z, v.z, 0x02030001u);
        v.w = __builtin_amdgcn_perm(v.w, v.w, 0x02030001u);
        out[i] = v;
    }
}

// any geometry / alignment: one pixel per lane
__global__ __launch_bounds__(kBlock) void merge_be16_scalar_kernel(const uint16_t *__restrict__ tiles, uint16_t *__restrict__ out,
                                                                   int hparts, int sub_lines, int sub_cols, long total)
{
    const long stride = (long)gridDim.x * kBlock;
    const long W = (long)hparts * sub_cols;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
        const long row = i / W;
        const int x = (int)(i - row * W);
        const int c = x / sub_cols, col = x - c * sub_cols;
        const long r = row / sub_lines;
        const int line = (int)(row - r * sub_lines);
        const unsigned w = tiles[((r * hparts + c) * sub_lines + line) * (long)sub_cols + col];
        out[i] = (uint16_t)(((w & 0xFFu) << 8) | (w >> 8));
    }
}

}  // namespace

extern "C" int oip_merge_subimages_be16(oip_ctx *ctx, const uint16_t *d_tiles, uint16_t *d_out, int vparts, int hparts,
                                        int sub_lines, int sub_cols)
{
    OIP_CHECK_CTX(ctx);
    if (!d_tiles || !d_out || vparts <= 0 || hparts <= 0 || sub_lines <= 0 || sub_cols <= 0)
        return oip_fail(ctx, OIP_E_INVALID, "oip_merge_subimages_be16: bad argument");
    if (d_tiles == d_out) return oip_fail(ctx, OIP_E_INVALID, "oip_merge_subimages_be16: in-place merge is not possible");
    const long total = (long)vparts * hparts * sub_lines * sub_cols;
    OipProfScope prof(ctx, "merge_be16_kernel");
    if ((sub_cols & 7) == 0 && (((size_t)d_tiles | (size_t)d_out) & 15) == 0) {
        const long chunks = total / 8;
        long blocks = (chunks + kBlock - 1) / kBlock;
        const long cap = (long)ctx->cu_count * 16;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(merge_be16_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, reinterpret_cast<const uint4 *>(d_tiles),
                           reinterpret_cast<uint4 *>(d_out), hparts, sub_lines, sub_cols / 8, chunks);
    } else {
        long blocks = (total + kBlock - 1) / kBlock;
        const long cap = (long)ctx->cu_count * 32;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(merge_be16_scalar_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, d_tiles, d_out, hparts, sub_lines,
                           sub_cols, total);
    }
    OIP_HIP(ctx, hipGetLastError());
    return OIP_OK;
}
