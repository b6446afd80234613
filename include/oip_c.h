/*
 * oip_c.h -- C ABI of liboipgpu.so, the MI355X (gfx950) implementation of the
 * arloan/OpticalImageProcessor hot path: per-column relative radiometric correction,
 * cross-CCD / inter-band phase correlation, bicubic resampling and strip stitching.
 *
 * The reference has no FFI: the path sits behind header-only C++ classes with static
 * methods (IMO, Stitcher, PreProcessor).  Each entry point below names the reference
 * seam it replaces (file:line under OpticalImageProcessor/).  INTEGRATION.md shows the
 * few lines a maintainer adds to the reference to call them.
 *
 * Conventions
 *   - plain pointers and sizes only; `d_*` arguments are device (HBM) pointers, everything
 *     else is host memory.  Rasters are headerless row-major uint16 (little endian), pitch ==
 *     width, exactly the reference's RAW layout (oipshared.h:27-32).
 *   - every call returns an oip_status; oip_last_error(ctx) gives the message.  The status
 *     classes mirror the exception types the reference throws so a C++ shell can re-throw
 *     them and keep the exit codes of main.cpp:320-343.
 *   - kernels are enqueued on the context's stream and are asynchronous unless the
 *     function returns host values (then it synchronises the stream itself).
 *   - one oip_ctx per device; a context is not thread-safe, distinct contexts are independent.
 *   - there is NO CPU fallback: without a gfx950 device oip_create fails.
 */
#ifndef OIP_C_H
#define OIP_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oip_ctx oip_ctx;

typedef enum oip_status {
    OIP_OK = 0,
    OIP_E_INVALID = 1,   /* std::invalid_argument in the reference                       */
    OIP_E_RUNTIME = 2,   /* std::runtime_error                                           */
    OIP_E_IO = 3,        /* errno_error (open/read/write/stat)                           */
    OIP_E_DEVICE = 4,    /* HIP failure / no gfx950 device                               */
    OIP_E_NOMEM = 5,
    OIP_E_UNSUPPORTED = 6
} oip_status;

/* reference constants (oipshared.h:27-54, imageop.h:19-20); width is a run-time argument
 * everywhere below, these are only the defaults */
#define OIP_PIXELS_PER_LINE      12288
#define OIP_MSS_BANDS            4
#define OIP_CORRELATION_LINES    16000
#define OIP_IBCV_DEF_THRESHOLD   0.4
#define OIP_IBCV_MIN_COUNT       5
#define OIP_IBCV_DEF_SECTIONS    5
#define OIP_IBCV_DEF_SLICES      10
#define OIP_IBCV_MIN_SLICES      8
#define OIP_IBPA_DEFAULT_BATCHLINES  20000
#define OIP_IBPA_DEFAULT_LINEOVERLAP 520
#define OIP_IBPA_MAX_LINEOVERLAP     3000
#define OIP_IBPA_MIN_PROCESSLINES    1500
#define OIP_STT_DEF_SECTIONS     10
#define OIP_STT_DEF_SECLINES     16000
#define OIP_STT_DEF_OVERLAPPX    200
#define OIP_STT_DEF_PHCTHRHLD    0.4
#define OIP_REMAP_ROW_GUARD      32767
#define OIP_REMAP_SECTION_ROWS   30000

/* ---- context, stream, memory ------------------------------------------------------ */
int         oip_version(void);                       /* 0x0101 == "1.1" (main.cpp:94)  */
int         oip_create(int device, oip_ctx **out);
void        oip_destroy(oip_ctx *ctx);
const char *oip_last_error(const oip_ctx *ctx);
int         oip_set_stream(oip_ctx *ctx, void *hip_stream);  /* borrow a hipStream_t (NULL: own) */
void       *oip_get_stream(oip_ctx *ctx);
int         oip_sync(oip_ctx *ctx);
int         oip_malloc(oip_ctx *ctx, void **d_ptr, size_t bytes);
int         oip_free(oip_ctx *ctx, void *d_ptr);
int         oip_memset(oip_ctx *ctx, void *d_ptr, int value, size_t bytes);
int         oip_memcpy_h2d(oip_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int         oip_memcpy_d2h(oip_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int         oip_host_alloc(oip_ctx *ctx, void **ptr, size_t bytes);   /* pinned staging */
int         oip_host_free(oip_ctx *ctx, void *ptr);

/* ---- RRC --------------------------------------------------------------------------- */
/* IMO::LoadRRCParamFile(path, expectedLines)  imageop.h:140-192.  kb_out receives
 * expected_lines (k,b) pairs == RRCParam[expected_lines].  No context needed. */
int oip_load_rrc_param_file(const char *path, int expected_lines, double *kb_out,
                            char *err, int errlen);

/* IMO::InplaceRRC(buff, w, h, rrcParam)  imageop.h:129-138; callers imageop.h:207,
 * preproc.h:195, :215.  dst[y*w+x] = (uint16_t)(k[x]*src[y*w+x] + b[x]) in fp64 with two
 * roundings and x86-64 truncate/wrap conversion; bit-exact.  d_dst may equal d_src
 * (in place, as the reference).  d_kb: w (k,b) pairs in HBM. */
int oip_rrc_u16(oip_ctx *ctx, const uint16_t *d_src, uint16_t *d_dst, int w, long h,
                const double *d_kb);

/* The same seam on a WINDOW: columns [0, w) of h lines of a raster of pitch src_pitch, written to a raster of pitch
 * dst_pitch (pitches in pixels; d_kb: the w (k,b) pairs of the window's columns).  With dst = the stitched raster
 * (pitch 2 (W - fold), w = W - fold) this is the left half of IMO::StitchBigRaw's output line (imageop.h:340-351)
 * taken straight from the raw CCD-1 strip: prestitch -> stitch without materialising <pan1>.RRC.RAW. */
int oip_rrc_u16_window(oip_ctx *ctx, const uint16_t *d_src, long src_pitch, uint16_t *d_dst, long dst_pitch,
                       int w, long h, const double *d_kb);

/* Host-buffer form of the same seam: `buff` is the reference's heap buffer, corrected in
 * place through pinned, double-buffered line blocks (H2D || kernel || D2H). */
int oip_rrc_u16_host(oip_ctx *ctx, uint16_t *buff, int w, long h, const double *kb);

/* ---- raster I/O staging (imageop.h:43-127, stitcher.h:103-120) ---------------------------------------
 * IMO::ReadFileContent + LoadRawImage / WriteBufferToFile move a raster through one pageable heap buffer,
 * serially with the arithmetic (8 MiB fread / fwrite units on the calling thread, imageop.h:69-79, :88-95).
 * These entry points move it in 32 MiB blocks through a ring of pinned buffers on a staging stream of the
 * context's own -- a slot is filled from the file by parallel pread on the host copy pool -- so disk, host
 * copies, PCIe and the kernels of the compute stream overlap.  Staging calls may run on other host threads
 * while the first drives kernels through the same context: they use streams and pinned slots of their own,
 * set the context's device for the calling thread, and never touch the compute stream's state.  Calls of one
 * lane serialise on a lock (ring lane: read_file / upload_staged / rrc_u16_host; download lane:
 * download_staged[_after] / write_file[_at] / file_sink_write: three of them, a call takes a free one), ring and download
 * lanes run concurrently (full duplex): a reader thread,
 * the compute thread and a writer thread form the pipeline of the `oip` CLI's default action.
 * Ordering: downloads and file writes start after the compute-stream work enqueued before the call, or after
 * a MARK of the compute stream (oip_compute_mark) when one is given; UPLOADS DO NOT wait for the compute
 * stream -- before re-uploading into a buffer that queued kernels still read, call
 * oip_stage_order_after_compute(ctx) (or upload elsewhere).
 *   ticket != NULL : the call returns once the last block's DMA is ENQUEUED and *ticket identifies it;
 *                    oip_stage_wait(ctx, t) makes the compute stream wait (on the device) for everything up
 *                    to t.  ticket == NULL: the compute stream is ordered behind the transfer by the call itself.
 *   downloads start after the compute-stream work enqueued before the call and return when the host side
 *   (file or buffer) is complete. */
/* ReadFileContent(filePath, size, offset, total, buff) with `buff` in HBM; bytes == 0: to the end of the file */
int oip_read_file_to_device(oip_ctx *ctx, const char *path, size_t offset, size_t bytes, void *d_dst,
                            size_t *bytes_read, long *ticket);
/* WriteBufferToFile(buff, size, saveFilePath) with `buff` in HBM (append != 0: "ab", as the section writes of
 * stitcher.h:114-120 accumulate one output file) */
int oip_write_device_to_file(oip_ctx *ctx, const void *d_src, size_t bytes, const char *path, int append);
/* the same into an EXISTING or new file at byte `file_offset` without truncating it (the file grows as needed): a product
 * written block by block as its lines become final -- <pan>.RRC.RAW while the strip is still being read, the pixel payload
 * of an uncompressed TIFF (oip_tiff.hpp) behind its header.  mark: 0, or a mark of the compute stream to wait for. */
int oip_write_device_to_file_at(oip_ctx *ctx, const void *d_src, size_t bytes, const char *path, size_t file_offset, long mark);
/* A product file PREPARED ahead of its pixels: created, its blocks reserved (a full file system fails at open, cleanly) and
 * mapped -- by a thread that has time for it, e.g. while the strip is still being read -- so that the later write is
 * HBM -> pinned slot -> parallel memory copies into pages that exist (on page-cache-backed files several
 * times the rate of allocating them during the write, which is what bounds WriteBufferToFile's loop, imageop.h:84-97).
 * The file is not truncated: a header written before stays.  bytes: the final size of the file.  A sink whose reservation or
 * mapping failed (or OIP_FILE_WRITE=pwrite) writes through pwrite.  mark as oip_write_device_to_file_at. */
typedef struct oip_file_sink oip_file_sink;
int oip_file_sink_open(oip_ctx *ctx, const char *path, size_t bytes, oip_file_sink **out);
int oip_file_sink_write(oip_ctx *ctx, oip_file_sink *sink, size_t file_offset, const void *d_src, size_t bytes, long mark);
int oip_file_sink_close(oip_ctx *ctx, oip_file_sink *sink);
/* A MARK names the compute-stream work enqueued so far (an event from a ring of 64; taken by the compute thread, e.g. right
 * after the RRC kernel of a line block).  A download-lane transfer given the mark starts once that work is done, not after
 * what the compute thread enqueued later (a 12-ms correlation batch, say).  A mark that has left the ring means "everything
 * enqueued so far". */
int oip_compute_mark(oip_ctx *ctx, long *mark);
int oip_compute_mark_sync(oip_ctx *ctx, long mark);   /* the calling host thread waits for the mark */
/* the same between a pageable host buffer and HBM (copies to/from the pinned ring run on a thread pool) */
int oip_upload_staged(oip_ctx *ctx, void *d_dst, const void *host, size_t bytes, long *ticket);
/* the same for a 2-D block (a column block of a raster): `rows` rows of `width` bytes; host rows src_pitch bytes apart,
 * device rows dst_pitch bytes apart; width at most 32 MiB */
int oip_upload_staged_2d(oip_ctx *ctx, void *d_dst, size_t dst_pitch, const void *host, size_t src_pitch, size_t width,
                         size_t rows, long *ticket);
int oip_download_staged(oip_ctx *ctx, void *host, const void *d_src, size_t bytes);
int oip_download_staged_after(oip_ctx *ctx, void *host, const void *d_src, size_t bytes, long mark);
int oip_stage_wait(oip_ctx *ctx, long ticket);
int oip_stage_sync(oip_ctx *ctx);
int oip_stage_order_after_compute(oip_ctx *ctx);      /* the ring lane's later transfers wait for the compute stream's work enqueued so far */
int oip_stage_threads(void);                          /* threads of the host copy pool (OIP_HOST_COPY_THREADS) */
/* where the host side of the upload lane spent its time since the last reset: out[0] seconds in pageable -> pinned copies,
 * out[1] seconds waiting for a ring slot whose DMA had not finished (the link is the limit then), out[2] bytes, out[3] calls */
int oip_stage_stats(oip_ctx *ctx, double *out, int reset);

/* PreProcessor::LoadMSS split (preproc.h:62-75) fused with DoRRC4MSS (preproc.h:202-222):
 * one pass over the BIL MSS raster (each line = 4 bands x w/4 px) writing 4 planar,
 * RRC-corrected bands (band b at d_planes + b*plane_stride).  d_kb4: 4 x (w/4) (k,b)
 * pairs, band-major; NULL = split only (--no-rrc4mss). */
int oip_mss_split_rrc_u16(oip_ctx *ctx, const uint16_t *d_bil, uint16_t *d_planes,
                          size_t plane_stride, int w, long lines, const double *d_kb4);

/* ---- correlation ------------------------------------------------------------------- */
/* cv::phaseCorrelate(src1, src2, noArray(), &response)  call sites stitcher.h:180,
 * preproc.h:316.  d_a/d_b: continuous rows x cols f32.  Synchronises; host outputs. */
int oip_phase_correlate_f32(oip_ctx *ctx, const float *d_a, const float *d_b, int rows,
                            int cols, double *dx, double *dy, double *response);

/* Mat1w.colRange -> Mat1f conversion (stitcher.h:175-176, preproc.h:258-293) */
int oip_window_u16_to_f32(oip_ctx *ctx, const uint16_t *d_img, size_t pitch, long row0,
                          int col0, int rows, int cols, float *d_out);

/* cv::resize(f32, dsize, 0, 0, INTER_CUBIC)  call site preproc.h:302-307 */
int oip_resize_cubic_f32(oip_ctx *ctx, const float *d_src, int sw, int sh, float *d_dst,
                         int dw, int dh);

/* Loop body of Stitcher::CalcSttParameters (stitcher.h:166-191) for all sections:
 * out[s*3 + {0,1,2}] = dx, dy, response of section s (host).  d_pan1/d_pan2 hold global
 * lines [row0, row0+nrows) of the W-wide rasters of L lines; sections not fully inside
 * that range are skipped and reported as NaN (multi-GPU: each rank computes the sections
 * it owns, results are all-gathered). */
int oip_stt_correlate(oip_ctx *ctx, const uint16_t *d_pan1, const uint16_t *d_pan2, int W,
                      long L, long row0, long nrows, int sections, int lines_per_section,
                      int overlap_cols, int edge_cols, double *out);

/* The same loop body for n explicit window pairs (multi-GPU: a section whose lines live on several ranks
 * is gathered into compact windows on the rank that computes it): window i of CCD 1 / CCD 2 is rows x cols
 * u16 at d_a[i] / d_b[i] with row pitch pitch_a[i] / pitch_b[i] (elements) -- the Mat1w.colRange views of
 * stitcher.h:175-176.  out[3*i + {0,1,2}] = dx, dy, response.  The pointer arrays are host arrays of
 * device pointers. */
int oip_stt_correlate_windows(oip_ctx *ctx, const uint16_t *const *d_a, const size_t *pitch_a,
                              const uint16_t *const *d_b, const size_t *pitch_b, int n, int rows, int cols,
                              double *out);

/* Loop body of PreProcessor::CalcInterBandCorrelation (preproc.h:251-329):
 * out[((b*sections + sec)*slices + i)*4 + {0..3}] = dx, dy, rs, cx.  PAN lines
 * [prow0, prow0+pn) and MSS band lines [mrow0, mrow0+mn) are resident; sections not fully
 * inside are reported as NaN.
 * cv::resize(INTER_CUBIC) of the band window (preproc.h:302-307) followed by the transform of the up-sampled
 * image is computed, for slices of 3000 columns whose window is exactly 4 x the band window, as the transform of
 * the band window itself expanded by the up-sampling operator's own transform (an exact identity, see DESIGN.md
 * 4.3); results agree with "up-sample, then transform" to ~4e-6 px.  Environment: OIP_SPECTRAL_UP=0 keeps the
 * up-sampling in the image domain (1: horizontal axis only on the spectra). */
int oip_interband_correlate(oip_ctx *ctx, const uint16_t *d_pan, long Lp, long prow0, long pn,
                            const uint16_t *d_planes, size_t plane_stride, long mrow0, long mn,
                            int W, int slices, int sections, int corr_lines, double *out);

/* The (section, slice) body of the same loop (preproc.h:262-329) for n explicit units: unit u is a PAN
 * window of rows x cols u16 at d_pan[u] (pitch pan_pitch[u]) and the four band windows of (rows/4) x
 * (cols/4) u16 at d_bands[4*u + b] (pitch band_pitch[u]).  out[12*u + 3*b + {0,1,2}] = dx, dy, rs of band
 * b.  Lets a multi-GPU host hand any unit to any rank (a unit needs 96 MB + 4 x 6 MB of windows at the
 * 30000-wide geometry).  Units are processed two at a time (2u, 2u+1 share transforms), so the last
 * digits of a unit's result (~1e-6 px) depend on its partner: a host that wants the bits of the
 * single-GPU run keeps the pairs of oip_interband_correlate's order (section-major, slices in order). */
int oip_interband_correlate_units(oip_ctx *ctx, const uint16_t *const *d_pan, const size_t *pan_pitch,
                                  const uint16_t *const *d_bands, const size_t *band_pitch, int n, int rows,
                                  int cols, double *out);

/* The x4 cubic up-sampling of cv::resize (preproc.h:302-307) along one axis as an operator on spectra -- what
 * oip_interband_correlate applies to the transforms of the band windows instead of transforming the up-sampled
 * image.  Host only.  out: 5 x (4 n) complex floats, rows H, G_0 .. G_3:
 *   DFT_4n(up-sampled s)[k] = H[k] DFT_n(s)[k mod n] + sum_j G_j[k] s[J_j],   J = {0, 1, n-2, n-1}.
 * OIP_E_UNSUPPORTED for n < 8. */
int oip_upsample_operator(int n, float *out);

/* The validity filter and means of Stitcher::CalcSttParameters (stitcher.h:181-198), host: table[3*s +
 * {0,1,2}] = dx, dy, response of section s, in section order.  OIP_E_RUNTIME when no section is valid
 * ("No valid delta value found for stitching parameter calculating"). */
int oip_stt_mean(const double *table, int sections, double threshold, double max_delta_y, double *dx,
                 double *dy, double *response, int *valid);

/* FilterInterBandShiftValues + DoCorrelationPolynomialFitting (preproc.h:492-550), host.
 * shifts: [4][n][4] (dx,dy,rs,cx).  cx_out[4][2], cy_out[4][3] ascending coefficients.
 * oip_filter_and_fit fits like the reference (OIP_FIT_REFERENCE). */
#define OIP_FIT_REFERENCE 0  /* NumCpp Poly1d::fit as called at preproc.h:535-536: inv(A^T A) A^T y, raw abscissa */
#define OIP_FIT_LSTSQ     1  /* the same least-squares problem by Householder QR on a scaled abscissa       */
int oip_filter_and_fit(const double *shifts, int n, double threshold, int min_count,
                       double *cx_out, double *cy_out, char *err, int errlen);
int oip_filter_and_fit_mode(const double *shifts, int n, double threshold, int min_count, int fit_mode,
                            double *cx_out, double *cy_out, char *err, int errlen);
/* nc::polynomial::Poly1d<double>::fit(x, y, deg), ascending coefficients: the reference's operation
 * order (oip_polyfit_reference) and the well-conditioned solver (oip_polyfit) */
int oip_polyfit_reference(const double *x, const double *y, int n, int deg, double *coeffs);
int oip_polyfit(const double *x, const double *y, int n, int deg, double *coeffs);

/* ---- resampling -------------------------------------------------------------------- */
/* Stitcher::PreStitch (stitcher.h:83-139) + IMO::SectionaryRemap (imageop.h:230-275) +
 * cv::remap(INTER_CUBIC, BORDER_CONSTANT) (imageop.h:258): out(x,y) = bicubic(src, x+dx,
 * y+dy) with OpenCV's 1/32-px quantisation, per-section borders and cuts.  The float maps
 * are never materialised.  d_src holds global lines [src_row0, src_row0+src_rows) and
 * d_dst receives output lines [out_row0, out_row0+out_rows) of the W x L raster (whole
 * strip: 0, L, 0, L).  OIP_E_INVALID if L <= row_guard (imageop.h:242-244). */
int oip_remap_shift_bicubic_u16(oip_ctx *ctx, const uint16_t *d_src, long src_row0,
                                long src_rows, uint16_t *d_dst, long out_row0, long out_rows,
                                int W, long L, double dx, double dy, int section_rows,
                                int row_guard);
/* The same call with the 16-tap sums of the regular interior pixels accumulated in packed fp16 (BASELINE
 * config 5: "fp16 accumulate (tolerance stated)").  NOT the parity mode: |result - fp32 result| <= 6 DN on
 * 12-bit data (measured: max 5, mean 0.25 DN, 20-25 % of the pixels differ), <= 6 + max|sample - 2048|/64
 * DN in general; samples enter as (sample - 2048), so the mode is specified for data up to 15 bits.
 * Geometry, 1/32-px phases, section borders and the irregular columns are identical (and computed in
 * f32); widths that are not a multiple of 8 fall back to the fp32 kernel altogether. */
int oip_remap_shift_bicubic_u16_f16acc(oip_ctx *ctx, const uint16_t *d_src, long src_row0,
                                       long src_rows, uint16_t *d_dst, long out_row0, long out_rows,
                                       int W, long L, double dx, double dy, int section_rows,
                                       int row_guard);
/* The same resampling written into a WINDOW of another raster: column x >= dst_col0 of output line r goes to
 * d_dst[r * dst_pitch + dst_col_off + (x - dst_col0)], columns below dst_col0 are not stored (d_dst: first output line
 * of the destination raster, dst_pitch in pixels).  With dst_pitch = 2 (W - fold), dst_col0 = fold, dst_col_off = W - fold
 * the resampled CCD-2 line lands in the right half of IMO::StitchBigRaw's output line (imageop.h:340-351): prestitch ->
 * stitch without materialising .RRC.PRESTT.RAW (the fused single-pass pipeline of DOC/sample-task.sh; `oip task`).
 * f16acc != 0 selects the fp16-accumulate variant.  Pixels are those of the two plain calls, bit for bit. */
int oip_remap_shift_bicubic_u16_window(oip_ctx *ctx, const uint16_t *d_src, long src_row0, long src_rows,
                                       uint16_t *d_dst, long dst_pitch, int dst_col0, long dst_col_off,
                                       long out_row0, long out_rows, int W, long L, double dx, double dy,
                                       int section_rows, int row_guard, int f16acc);
/* The same with the source being the RAW CCD-2 strip: every sample is corrected on load (IMO::InplaceRRC's pixel, exact;
 * d_kb: the W (k,b) pairs in HBM), so Stitcher::DoRRC of CCD 2, PreStitch and the right half of StitchBigRaw are ONE pass
 * over the strip and <pan2>.RRC.RAW is not materialised either.  Bits are those of oip_rrc_u16 followed by
 * oip_remap_shift_bicubic_u16_window with the same f16acc (f16acc != 0 needs W % 8 == 0 and a 16-byte aligned source:
 * OIP_E_UNSUPPORTED otherwise). */
int oip_remap_shift_rrc_bicubic_u16_window(oip_ctx *ctx, const uint16_t *d_src_raw, long src_row0, long src_rows,
                                           const double *d_kb, uint16_t *d_dst, long dst_pitch, int dst_col0,
                                           long dst_col_off, long out_row0, long out_rows, int W, long L, double dx,
                                           double dy, int section_rows, int row_guard, int f16acc);
/* source lines [first, last) that output lines [out_row0, out_row0+out_rows) read: the halo
 * a row-block shard has to hold (host arithmetic only) */
int oip_remap_shift_src_range(long out_row0, long out_rows, long L, double dy,
                              int section_rows, long *first, long *last);

/* PreProcessor::DoInterBandAlignment outer (preproc.h:351-425) + inner (:428-468) incl.
 * cv::remap and cv::merge: 4 planar bands -> interleaved 16UC4, polynomial maps evaluated
 * in fp64 in the kernel.  cx[4][2], cy[4][3] host.  d_planes holds MSS lines [src_row0,
 * src_row0+src_rows); d_dst receives output lines [out_row0, out_row0+out_rows) of the
 * (Lm - line_offset - (keep?0:overlap)) x Wb x 4 result; skipped tail lines are zero.
 * rows_valid (may be NULL): the reference's processedLines. */
int oip_align_mss_bicubic_u16x4(oip_ctx *ctx, const uint16_t *d_planes, size_t plane_stride,
                                long src_row0, long src_rows, uint16_t *d_dst, long out_row0,
                                long out_rows, int Wb, long Lm, const double *cx,
                                const double *cy, int lines_per_section, int line_offset,
                                int overlap, int keep_leading, int min_lines, long *rows_valid);
int oip_align_mss_src_range(long out_row0, long out_rows, long Lm, const double *cy, int Wb,
                            int lines_per_section, int line_offset, int overlap,
                            int keep_leading, int min_lines, long *first, long *last);

/* IMO::StitchBigRaw line loop (imageop.h:340-351), RAW output: out line = left[0:W-fold] ||
 * right[fold:W]; `fold` is the already-halved value (main.cpp:189). */
int oip_stitch_rows_u16(oip_ctx *ctx, const uint16_t *d_left, const uint16_t *d_right,
                        uint16_t *d_out, int W, long L, int fold);

/* In-place sample permutation of an interleaved 4-channel u16 image: sample i of every pixel becomes the former sample
 * order[i].  cv::imwrite stores a 4-channel Mat (c0,c1,c2,c3) as samples (c2,c1,c0,c3) (preproc.h:167-185 WriteAlignedMSS_TIFF;
 * imageop.h:390-402), GDAL writes band b from channel bandMap[b]-1 (imageop.h:529): with the image already in file order on the
 * device, its lines go from HBM into the TIFF's pixel payload without a host pass.  d_img 16-byte aligned. */
int oip_permute_u16x4(oip_ctx *ctx, uint16_t *d_img, size_t npixels, const int *order);

/* The strips of an LZW TIFF product, encoded on the device (cv::imwrite's TIFF encoder behind preproc.h:167-185 and GDAL's
 * COMPRESS=LZW PREDICTOR=2 behind imageop.h:460-567 do this on the host, strip by strip).  d_img: rows x width x spp u16,
 * interleaved, in file sample order (oip_permute_u16x4 first where cv::imwrite / a band map reorder); spp 1 or 4; strip k
 * holds rows [k rows_per_strip, (k + 1) rows_per_strip).  Every strip is the horizontal-predictor differences of its rows
 * through TIFF 6.0's LZW as libtiff writes it (MSB-first 9..12-bit codes, ClearCode first, early change, EndOfInformation).
 * The encoded strips are packed into d_payload at even offsets in strip order; strip_off / strip_len (host arrays, one
 * entry per strip) say where, *payload_bytes is the end of the last one.  payload_cap >= oip_tiff_lzw_worst_bytes().
 * d_scratch: NULL (the call allocates and frees its own) or >= oip_tiff_lzw_scratch_bytes() of device memory, 8-byte aligned,
 * that a caller who knows the product's geometry early prepares off the critical path.  Synchronises the context's stream. */
size_t oip_tiff_lzw_worst_bytes(long rows, int width, int spp, long rows_per_strip);
size_t oip_tiff_lzw_scratch_bytes(long rows, int width, int spp, long rows_per_strip);
int oip_tiff_lzw_strips_u16(oip_ctx *ctx, const uint16_t *d_img, long rows, int width, int spp, long rows_per_strip,
                            uint8_t *d_payload, size_t payload_cap, uint64_t *strip_off, uint64_t *strip_len,
                            size_t *payload_bytes, void *d_scratch, size_t scratch_bytes);

/* ... and read: the LZW strips of a TIFF file (cv::imread of the stitch inputs, imageop.h:380-388) decoded on the device.
 * d_file: the file's bytes from some base offset on, already in HBM (oip_read_file_to_device); strip_off / strip_len (host):
 * every strip's offset inside d_file and its size; chunky u16 samples, predictor 1 or 2.  d_img receives rows x width x spp
 * samples in file order.  A stream that ends without EndOfInformation is tolerated (as libtiff does); a corrupt stream or a
 * strip that does not decode to exactly its rows is OIP_E_RUNTIME with the strip named.  Synchronises the stream. */
int oip_tiff_lzw_decode_u16(oip_ctx *ctx, const uint8_t *d_file, size_t file_bytes, const uint64_t *strip_off,
                            const uint64_t *strip_len, long nstrips, long rows, int width, int spp, long rows_per_strip,
                            int predictor, uint16_t *d_img);

/* ---- instrumentation --------------------------------------------------------------- */
/* name + accumulated device time of the kernels launched through this context since the
 * last reset, measured with HIP events on the context's stream (off by default). */
/* ---- 8f rank 4: sub-image merge of the down-link de-framer ------------------------------------------
 * AuxSeparator::WriteImageData / MergeSubImage / the byte-order pass of InflateSubImage
 * (aux_separator.h:341-393) for uncompressed frames: d_tiles holds vparts*hparts sub-images of
 * sub_lines x sub_cols BIG-endian u16, tile r*hparts + c after tile r*hparts + c - 1; d_out receives
 * vparts stripes of sub_lines lines of hparts*sub_cols little-endian pixels
 * (reference frame: vparts = 4 PAN + 1 MSS, hparts = 8, 256 x 1536 sub-images).  Not in place. */
int oip_merge_subimages_be16(oip_ctx *ctx, const uint16_t *d_tiles, uint16_t *d_out, int vparts, int hparts,
                             int sub_lines, int sub_cols);

int oip_profile_enable(oip_ctx *ctx, int on);
int oip_profile_reset(oip_ctx *ctx);
/* Time only the kernels profiled under this name (NULL or "": every kernel).  The events themselves cost
   stream time (about 2 % of a correlation batch when every kernel is timed). */
int oip_profile_filter(oip_ctx *ctx, const char *kernel_name);
int oip_profile_count(oip_ctx *ctx);
int oip_profile_get(oip_ctx *ctx, int i, char *name, int namelen, double *total_ms, long *launches);

#ifdef __cplusplus
}
#endif
#endif /* OIP_C_H */
