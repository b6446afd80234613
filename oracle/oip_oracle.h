/*
 * oip_oracle.h -- CPU restatement of the reference hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This library is the *checker*, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (liboipgpu.so) never
 * links, loads or falls back to anything in oracle/.
 *
 * Each function restates one piece of arloan/OpticalImageProcessor (citations are
 * file:line under /root/reference/OpticalImageProcessor/) or of the third-party
 * routine the reference calls there (OpenCV imgproc, un-vendored, version unpinned:
 * CMakeLists.txt:8).
 *
 * Parity status
 *   orc_inplace_rrc ............ PINNED  (checked bit-for-bit against oracle/_ref, the
 *                                reference's own InplaceRRC lines compiled in place, and
 *                                against tests/golden/rrc_*.bin generated from it)
 *   everything else ............ PARITY UNPINNED: the arithmetic lives in OpenCV/NumCpp,
 *                                which are absent from the reference tree and this image;
 *                                the reference holds no tests or golden vectors.  These
 *                                functions restate OpenCV 4.x's published algorithm and
 *                                are anchored on analytic known answers (tests/golden/).
 */
#ifndef OIP_ORACLE_H
#define OIP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_rrc_param { double k; double b; } orc_rrc_param; /* imageop.h:26-29 */

/* imageop.h:129-138  IMO::InplaceRRC */
void orc_inplace_rrc(uint16_t *buff, int w, int h, const orc_rrc_param *p);
/* same loop, rows split over `threads` pthreads (the "1-socket" CPU baseline of SURVEY 8d) */
void orc_inplace_rrc_mt(uint16_t *buff, int w, int h, const orc_rrc_param *p, int threads);

/* imageop.h:140-192  IMO::LoadRRCParamFile.  Returns 0 on success, else a code
 * (1 open, 2 header read, 3 count mismatch, 4 bad row, 5 row count) and fills err. */
int orc_load_rrc_param_file(const char *path, int expected_lines, orc_rrc_param *out,
                            char *err, int errlen);

/* preproc.h:62-75  PreProcessor::LoadMSS split of a BIL line into 4 planar bands */
void orc_split_mss(const uint16_t *bil, uint16_t *b0, uint16_t *b1, uint16_t *b2,
                   uint16_t *b3, int pixels_per_line, size_t lines);

/* OpenCV imgwarp.cpp interpolateCubic / initInterTab2D(INTER_CUBIC): tab[32*32][16] f32 */
void orc_bicubic_tab(float *tab);
void orc_interpolate_cubic(float x, float *coeffs4);

/* OpenCV cv::remap(src 16UC1, dst, mapx 32F, mapy 32F, INTER_CUBIC, BORDER_CONSTANT, 0)
 * call sites imageop.h:258, preproc.h:453-457.  sstep/dstep in elements. */
void orc_remap_cubic_u16(const uint16_t *src, int sw, int sh, size_t sstep,
                         uint16_t *dst, int dw, int dh, size_t dstep,
                         const float *mapx, const float *mapy, size_t mstep);

/* stitcher.h:83-139 Stitcher::PreStitch + imageop.h:230-275 IMO::SectionaryRemap,
 * with the reference's constants as parameters (section_rows = REMAP_SECTION_ROWS 30000,
 * row_guard = REMAP_ROW_GUARD 32767).  src/dst are whole W x L rasters.
 * Returns the row_offset SectionaryRemap returns, or -1 if total_rows <= row_guard. */
long orc_prestitch(const uint16_t *src, uint16_t *dst, int W, int L, double dx, double dy,
                   int section_rows, int row_guard);

/* preproc.h:351-425 + :428-468 DoInterBandAlignment (outer + inner) incl. cv::merge.
 * dst is (Lm - line_offset - (keep?0:overlap)) x Wb x 4 interleaved, zero-filled first
 * (the reference leaves skipped tail rows uninitialised: SURVEY App.B-6).
 * min_lines = IBPA_MIN_PROCESSLINES (1500).  Returns rows actually written. */
long orc_align_mss(const uint16_t *b0, const uint16_t *b1, const uint16_t *b2,
                   const uint16_t *b3, uint16_t *dst, int Wb, long Lm,
                   const double *cx /*4x2*/, const double *cy /*4x3*/,
                   int lines_per_section, int line_offset, int overlap, int keep_leading,
                   int min_lines);

/* imageop.h:340-351 IMO::StitchBigRaw line loop (RAW output), fold already halved */
void orc_stitch_raw(const uint16_t *left, const uint16_t *right, uint16_t *out, int W,
                    long L, int fold);

/* OpenCV cv::resize(32F, INTER_CUBIC) (resize.cpp), call site preproc.h:302-307 */
void orc_resize_cubic_f32(const float *src, int sw, int sh, float *dst, int dw, int dh);

/* stitcher.h:175-176 / preproc.h:258-266: strided u16 window -> contiguous f32 */
void orc_window_u16_to_f32(const uint16_t *img, size_t pitch, long row0, int col0, int rows,
                           int cols, float *out);

#ifdef __cplusplus
}
#endif
/* AuxSeparator::WriteImageData + MergeSubImage + InflateSubImage's byte-order pass, uncompressed frames
 * (aux_separator.h:341-393): the reference copies sub-image c of stripe r row by row to column offset
 * c * sub_cols of a hparts*sub_cols-wide stripe after swapping the two bytes of every word. */
void orc_merge_subimages_be16(const uint16_t *tiles, uint16_t *out, int vparts, int hparts, int sub_lines, int sub_cols);

#endif
