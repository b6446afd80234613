/*
 * oip_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY:
 * see oip_oracle.h for who may load this and for the parity status of each function.
 *
 * Build: gcc -O2 -ffp-contract=off (no -march, no -ffast-math): the reference's CMake
 * Release build targets baseline x86-64, which has no FMA, so every a*b+c below rounds
 * twice -- exactly like the reference binary on Linux x86-64 (SURVEY App.B-10).
 */
#include "oip_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * RRC  (imageop.h:129-138)
 *   uint16_t dst = (uint16_t)(rrcParam[x].k * src + rrcParam[x].b);
 * double -> uint16_t is undefined in C++ when out of range; on x86-64 every compiler the
 * survey tried lowers it to cvttsd2si (32-bit) followed by a 16-bit truncation, i.e.
 * (uint16_t)(int32_t)trunc(v), with cvttsd2si's "integer indefinite" 0x80000000 (low
 * half 0) for NaN / |v| >= 2^31.  That de-facto behaviour is what is restated here, in
 * defined C.
 * ---------------------------------------------------------------------------------- */
static inline uint16_t rrc_px(double k, double b, uint16_t src)
{
    double v = k * (double)src + b;            /* two roundings, no FMA */
    int32_t t;
    if (v > -2147483649.0 && v < 2147483648.0) t = (int32_t)v;   /* trunc toward zero */
    else t = INT32_MIN;                         /* cvttsd2si indefinite (also NaN)   */
    return (uint16_t)(uint32_t)t;
}

void orc_inplace_rrc(uint16_t *buff, int w, int h, const orc_rrc_param *p)
{
    for (size_t y = 0; y < (size_t)h; ++y)
        for (size_t x = 0; x < (size_t)w; ++x) {
            size_t idx = (size_t)w * y + x;
            buff[idx] = rrc_px(p[x].k, p[x].b, buff[idx]);
        }
}

struct rrc_job { uint16_t *buff; int w; int h0, h1; const orc_rrc_param *p; };
static void *rrc_worker(void *arg)
{
    struct rrc_job *j = (struct rrc_job *)arg;
    orc_inplace_rrc(j->buff + (size_t)j->w * j->h0, j->w, j->h1 - j->h0, j->p);
    return NULL;
}
void orc_inplace_rrc_mt(uint16_t *buff, int w, int h, const orc_rrc_param *p, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t tid[256];
    struct rrc_job job[256];
    for (int t = 0; t < threads; ++t) {
        job[t].buff = buff; job[t].w = w; job[t].p = p;
        job[t].h0 = (int)((long)h * t / threads);
        job[t].h1 = (int)((long)h * (t + 1) / threads);
        pthread_create(&tid[t], NULL, rrc_worker, &job[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
}

/* ------------------------------------------------------------------------------------
 * RRC parameter file  (imageop.h:140-192)
 * 3 header lines ("1", column count, "0" -- lines 1 and 3 are only asserted under DEBUG,
 * SURVEY App.B-9), then one "k , b" per column via sscanf(" %lf , %lf"); fgets buffer
 * 1024; the row count must equal expected_lines exactly; any unparsable line (incl. a
 * trailing blank one) is an error.
 * ---------------------------------------------------------------------------------- */
int orc_load_rrc_param_file(const char *path, int expected_lines, orc_rrc_param *out,
                            char *err, int errlen)
{
    FILE *f = fopen(path, "rb");
    if (!f) { snprintf(err, errlen, "open RRC Param file failed"); return 1; }
    char buff[1024];
    if (!fgets(buff, sizeof buff, f)) { fclose(f); snprintf(err, errlen, "LoadRRCParamFile([1]): read file content failed"); return 2; }
    if (!fgets(buff, sizeof buff, f)) { fclose(f); snprintf(err, errlen, "LoadRRCParamFile([2]): read file content failed"); return 2; }
    int lines = atoi(buff);
    if (lines != expected_lines) {
        fclose(f);
        snprintf(err, errlen, "LoadRRCParamFile([2]): expected %d lines while %d found in file content", expected_lines, lines);
        return 3;
    }
    if (!fgets(buff, sizeof buff, f)) { fclose(f); snprintf(err, errlen, "LoadRRCParamFile([3]): read file content failed"); return 2; }
    int index = 0;
    double k = .0, b = .0;
    for (; fgets(buff, sizeof buff, f); ++index) {
        if (sscanf(buff, " %lf , %lf", &k, &b) != 2) {
            fclose(f);
            snprintf(err, errlen, "line #%d of RRC param file [%s] found invalid", index, path);
            return 4;
        }
        if (index < expected_lines) { out[index].k = k; out[index].b = b; }
        /* the reference writes params[index] unchecked (heap overflow past
         * expected_lines); the restatement only counts the extra rows */
    }
    fclose(f);
    if (index != expected_lines) {
        snprintf(err, errlen, "RRC Param file [%s] invalid: %d lines of param expected, %d lines parsed.", path, expected_lines, index);
        return 5;
    }
    return 0;
}

/* preproc.h:62-75 */
void orc_split_mss(const uint16_t *bil, uint16_t *b0, uint16_t *b1, uint16_t *b2,
                   uint16_t *b3, int ppl, size_t lines)
{
    uint16_t *band[4] = { b0, b1, b2, b3 };
    int bpl = ppl / 4;
    for (size_t i = 0; i < lines; ++i)
        for (int b = 0; b < 4; ++b)
            memcpy(band[b] + i * bpl, bil + i * ppl + (size_t)b * bpl, (size_t)bpl * 2);
}

/* ------------------------------------------------------------------------------------
 * OpenCV bicubic tables (imgproc/imgwarp.cpp: interpolateCubic, initInterTab1D,
 * initInterTab2D with INTER_BITS=5, INTER_TAB_SIZE=32, float table).
 * ---------------------------------------------------------------------------------- */
void orc_interpolate_cubic(float x, float *coeffs)
{
    const float A = -0.75f;
    coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

void orc_bicubic_tab(float *tab)
{
    float t1[32 * 4];
    const float scale = 1.f / 32;
    for (int i = 0; i < 32; ++i) orc_interpolate_cubic(i * scale, t1 + i * 4);
    for (int i = 0; i < 32; ++i)           /* y phase */
        for (int j = 0; j < 32; ++j) {     /* x phase */
            float *w = tab + (size_t)(i * 32 + j) * 16;
            for (int k1 = 0; k1 < 4; ++k1) {
                float vy = t1[i * 4 + k1];
                for (int k2 = 0; k2 < 4; ++k2) w[k1 * 4 + k2] = vy * t1[j * 4 + k2];
            }
        }
}

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* cvRound: RNE */
static inline int16_t sat_short(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
static inline uint16_t sat_ushort_f(float v)
{
    int iv = cv_round_f(v);
    return (uint16_t)(iv < 0 ? 0 : iv > 65535 ? 65535 : iv);
}

/* ------------------------------------------------------------------------------------
 * cv::remap, 16UC1 source, two CV_32FC1 maps, INTER_CUBIC, BORDER_CONSTANT(0)
 * (imgwarp.cpp: RemapInvoker map conversion + remapBicubic<Cast<float,ushort>,float,1>).
 *   sx = cvRound(mapx*32); sy = cvRound(mapy*32); phase = (sy&31)*32 + (sx&31)
 *   X = saturate_cast<short>(sx>>5) - 1, Y likewise
 *   interior (0<=X<sw-3, 0<=Y<sh-3): sum = row0; sum += row1; sum += row2; sum += row3
 *        with rowk = S0*w0 + S1*w1 + S2*w2 + S3*w3 (left to right, f32)
 *   fully outside: 0;  partly outside: sum=0, then tap by tap sum += S*w for taps inside.
 * ---------------------------------------------------------------------------------- */
static float g_tab[32 * 32 * 16];
static int g_tab_ready = 0;
static pthread_once_t g_tab_once = PTHREAD_ONCE_INIT;
static void tab_init(void) { orc_bicubic_tab(g_tab); g_tab_ready = 1; }

void orc_remap_cubic_u16(const uint16_t *src, int sw, int sh, size_t sstep,
                         uint16_t *dst, int dw, int dh, size_t dstep,
                         const float *mapx, const float *mapy, size_t mstep)
{
    pthread_once(&g_tab_once, tab_init);
    unsigned width1 = (unsigned)(sw - 3 > 0 ? sw - 3 : 0);
    unsigned height1 = (unsigned)(sh - 3 > 0 ? sh - 3 : 0);
    for (int dy = 0; dy < dh; ++dy) {
        const float *mx = mapx + (size_t)dy * mstep;
        const float *my = mapy + (size_t)dy * mstep;
        uint16_t *D = dst + (size_t)dy * dstep;
        for (int dx = 0; dx < dw; ++dx) {
            int isx = cv_round_f(mx[dx] * 32);
            int isy = cv_round_f(my[dx] * 32);
            const float *w = g_tab + (size_t)((isy & 31) * 32 + (isx & 31)) * 16;
            int sx = sat_short(isx >> 5) - 1;
            int sy = sat_short(isy >> 5) - 1;
            if ((unsigned)sx < width1 && (unsigned)sy < height1) {
                const uint16_t *S = src + (size_t)sy * sstep + sx;
                float sum = S[0] * w[0] + S[1] * w[1] + S[2] * w[2] + S[3] * w[3];
                S += sstep;
                sum += S[0] * w[4] + S[1] * w[5] + S[2] * w[6] + S[3] * w[7];
                S += sstep;
                sum += S[0] * w[8] + S[1] * w[9] + S[2] * w[10] + S[3] * w[11];
                S += sstep;
                sum += S[0] * w[12] + S[1] * w[13] + S[2] * w[14] + S[3] * w[15];
                D[dx] = sat_ushort_f(sum);
            } else {
                if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) { D[dx] = 0; continue; }
                int x[4], y[4];
                for (int i = 0; i < 4; ++i) {
                    x[i] = (sx + i >= 0 && sx + i < sw) ? sx + i : -1;
                    y[i] = (sy + i >= 0 && sy + i < sh) ? sy + i : -1;
                }
                float cv = 0.f, sum = cv * 1;
                for (int i = 0; i < 4; ++i, w += 4) {
                    int yi = y[i];
                    if (yi < 0) continue;
                    const uint16_t *S = src + (size_t)yi * sstep;
                    if (x[0] >= 0) sum += (S[x[0]] - cv) * w[0];
                    if (x[1] >= 0) sum += (S[x[1]] - cv) * w[1];
                    if (x[2] >= 0) sum += (S[x[2]] - cv) * w[2];
                    if (x[3] >= 0) sum += (S[x[3]] - cv) * w[3];
                }
                D[dx] = sat_ushort_f(sum);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Stitcher::PreStitch (stitcher.h:83-139) driving IMO::SectionaryRemap
 * (imageop.h:230-275), restated literally:
 *   - one section_rows x W source buffer `buff` reused for every section; get_src reads
 *     `rows` lines into its head and returns the WHOLE buffer, so in a short last section
 *     rows [rows, section_rows) still hold the previous section's lines (App.B-4);
 *   - mapx/mapy are section_rows x W float maps of (x + dx), (y + dy) built once, y being
 *     section-relative (stitcher.h:93-99); remap's dst therefore always has section_rows rows;
 *   - ucut/bcut from stitcher.h:122-123; first section also emits its top ucut rows, every
 *     section emits rows [ucut, rows-bcut), and after the loop the last dst's final bcut rows.
 * ---------------------------------------------------------------------------------- */
long orc_prestitch(const uint16_t *src, uint16_t *dst, int W, int L, double dx, double dy,
                   int section_rows, int row_guard)
{
    if (L <= row_guard) return -1;     /* imageop.h:242-244 throws invalid_argument */
    size_t n = (size_t)section_rows * W;
    uint16_t *buff = (uint16_t *)calloc(n, 2);     /* cv::Mat1w buff (uninitialised there) */
    uint16_t *out = (uint16_t *)calloc(n, 2);
    float *mapx = (float *)malloc(n * sizeof(float));
    float *mapy = (float *)malloc(n * sizeof(float));
    for (int y = 0; y < section_rows; ++y)
        for (int x = 0; x < W; ++x) {
            size_t idx = (size_t)y * W + x;
            mapx[idx] = (float)(x + dx);
            mapy[idx] = (float)(y + dy);
        }
    int ucut = dy >= 0.0 ? 0 : (int)(-dy) + 1;
    int bcut = dy >= 0.0 ? (int)dy + 1 : 0;
    int total_cut = ucut + bcut;
    int row_offset = 0;
    size_t written = 0;                 /* output rows are appended sequentially (fwrite) */
    for (int s = 0;; ++s) {
        int rows = section_rows < L - row_offset ? section_rows : L - row_offset;
        if (rows <= total_cut) break;
        memcpy(buff, src + (size_t)row_offset * W, (size_t)rows * W * 2);   /* fseek+fread */
        orc_remap_cubic_u16(buff, W, section_rows, W, out, W, section_rows, W, mapx, mapy, W);
        if (s == 0 && ucut > 0) {
            memcpy(dst + written * W, out, (size_t)ucut * W * 2);
            written += ucut;
        }
        memcpy(dst + written * W, out + (size_t)ucut * W, (size_t)(rows - bcut - ucut) * W * 2);
        written += rows - bcut - ucut;
        row_offset += rows - total_cut;
    }
    if (bcut > 0) {
        memcpy(dst + written * W, out + (size_t)(section_rows - bcut) * W, (size_t)bcut * W * 2);
        written += bcut;
    }
    free(buff); free(out); free(mapx); free(mapy);
    (void)written;
    return row_offset;
}

/* ------------------------------------------------------------------------------------
 * PreProcessor::DoInterBandAlignment, outer (preproc.h:351-425) + inner (:428-468).
 * Inner: per band, maps evaluated in fp64 left to right then cast to f32
 *   mapX = (float)((cX1*xx + cX0 + xx)/4),  mapY = (float)((cY2*xx*xx + cY1*xx + cY0 + yy)/4)
 * with xx = 4x (int), yy = 4y (size_t), y relative to the section; cv::remap of the
 * section's rows; cv::merge -> 4-channel interleaved.  Outer: sections of
 * lines_per_section advancing lines_per_section - overlap; the first `overlap` rows of
 * each section are dropped (kept once with keep_leading); a trailing section shorter than
 * min_lines is skipped.
 * ---------------------------------------------------------------------------------- */
long orc_align_mss(const uint16_t *b0, const uint16_t *b1, const uint16_t *b2,
                   const uint16_t *b3, uint16_t *dst, int Wb, long Lm,
                   const double *cx, const double *cy,
                   int lps, int line_offset, int overlap, int keep_leading, int min_lines)
{
    const uint16_t *band[4] = { b0, b1, b2, b3 };
    long out_rows = Lm - line_offset - (keep_leading ? 0 : overlap);
    if (out_rows <= 0) return 0;
    memset(dst, 0, (size_t)out_rows * Wb * 4 * 2);
    size_t n = (size_t)lps * Wb;
    float *mapX = (float *)malloc(n * sizeof(float));
    float *mapY = (float *)malloc(n * sizeof(float));
    uint16_t *aligned = (uint16_t *)malloc(n * 2);
    uint16_t *merged = (uint16_t *)malloc(n * 4 * 2);
    long processed = 0;
    size_t offset = (size_t)line_offset;
    for (int i = 0;; ++i) {
        size_t rem = (size_t)Lm - offset;              /* size_t wrap as in the reference */
        size_t lines = rem < (size_t)lps ? rem : (size_t)lps;
        if ((size_t)Lm < offset || lines < (size_t)min_lines) break;
        int rows = (int)lines;
        for (int b = 0; b < 4; ++b) {
            const double *coeffX = cx + b * 2;
            const double *coeffY = cy + b * 3;
            for (size_t y = 0; y < (size_t)rows; ++y)
                for (int x = 0; x < Wb; ++x) {
                    size_t yy = y * 4;
                    int xx = x * 4;
                    mapX[y * Wb + x] = (float)((coeffX[1] * xx + coeffX[0] + xx) / 4);
                    mapY[y * Wb + x] = (float)((coeffY[2] * xx * xx + coeffY[1] * xx + coeffY[0] + yy) / 4);
                }
            orc_remap_cubic_u16(band[b] + offset * Wb, Wb, rows, Wb, aligned, Wb, rows, Wb,
                                mapX, mapY, Wb);
            for (size_t p = 0; p < (size_t)rows * Wb; ++p) merged[p * 4 + b] = aligned[p];   /* cv::merge */
        }
        if (i == 0 && keep_leading) {
            memcpy(dst, merged, (size_t)overlap * Wb * 8);
            processed += overlap;
        }
        memcpy(dst + (size_t)processed * Wb * 4, merged + (size_t)overlap * Wb * 4,
               (size_t)(rows - overlap) * Wb * 8);
        processed += rows - overlap;
        offset += (size_t)(lps - overlap);
    }
    free(mapX); free(mapY); free(aligned); free(merged);
    return processed;
}

/* imageop.h:340-351: per line, left[0:W-fold] then right[fold:W] */
void orc_stitch_raw(const uint16_t *left, const uint16_t *right, uint16_t *out, int W,
                    long L, int fold)
{
    int half = W - fold;
    for (long i = 0; i < L; ++i) {
        memcpy(out + (size_t)i * 2 * half, left + (size_t)i * W, (size_t)half * 2);
        memcpy(out + (size_t)i * 2 * half + half, right + (size_t)i * W + fold, (size_t)half * 2);
    }
}

/* ------------------------------------------------------------------------------------
 * cv::resize(CV_32FC1, INTER_CUBIC) (imgproc/resize.cpp: the coefficient set-up of
 * cv::hal::resize + HResizeCubic<float,float,float> + VResizeCubic<float,float,float>).
 *   fx = (float)((dx+0.5)*scale_x - 0.5); sx = floor(fx); fx -= sx;  (scale = 1/inv_scale,
 *   inv_scale = (double)dsize/ssize); taps sx-1..sx+2 clamped to the edge; horizontal pass
 *   to an f32 row buffer, then vertical  S0*b0 + S1*b1 + S2*b2 + S3*b3  (scalar order;
 *   OpenCV's SIMD body associates differently per version -- rounding-level only).
 * ---------------------------------------------------------------------------------- */
void orc_resize_cubic_f32(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    float *alpha = (float *)malloc(sizeof(float) * 4 * dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        xofs[dx] = sx;
        orc_interpolate_cubic(fx, alpha + dx * 4);
    }
    float *rows[4];
    for (int k = 0; k < 4; ++k) rows[k] = (float *)malloc(sizeof(float) * dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        float beta[4];
        orc_interpolate_cubic(fy, beta);
        for (int k = 0; k < 4; ++k) {
            int syk = sy - 1 + k;
            if (syk < 0) syk = 0;
            if (syk > sh - 1) syk = sh - 1;
            const float *S = src + (size_t)syk * sw;
            float *D = rows[k];
            for (int dx = 0; dx < dw; ++dx) {
                int sx = xofs[dx] - 1;
                const float *a = alpha + dx * 4;
                float v = 0;
                for (int j = 0; j < 4; ++j) {
                    int sxj = sx + j;
                    if (sxj < 0) sxj = 0;
                    if (sxj > sw - 1) sxj = sw - 1;
                    v += S[sxj] * a[j];
                }
                D[dx] = v;
            }
        }
        float *D = dst + (size_t)dy * dw;
        for (int x = 0; x < dw; ++x)
            D[x] = rows[0][x] * beta[0] + rows[1][x] * beta[1] + rows[2][x] * beta[2] + rows[3][x] * beta[3];
    }
    for (int k = 0; k < 4; ++k) free(rows[k]);
    free(xofs); free(alpha);
}

void orc_window_u16_to_f32(const uint16_t *img, size_t pitch, long row0, int col0, int rows,
                           int cols, float *out)
{
    for (int y = 0; y < rows; ++y) {
        const uint16_t *S = img + (size_t)(row0 + y) * pitch + col0;
        for (int x = 0; x < cols; ++x) out[(size_t)y * cols + x] = (float)S[x];
    }
}


/* aux_separator.h:341-393, ratio == IMGSIG_ZRTO_NONE: InflateSubImage memcpy's the sub-image and swaps the
 * bytes of every 16-bit word (:386-392); MergeSubImage (:366-372) memcpy's each of its rows to
 * image + r * BYTES_PER_PANLINE + vSlice * IMGSIG_IMBASE_COLS * BYTES_PER_PIXEL; WriteImageData (:347-363)
 * writes one full stripe per row of sub-images, PAN stripes first. */
void orc_merge_subimages_be16(const uint16_t *tiles, uint16_t *out, int vparts, int hparts, int sub_lines, int sub_cols)
{
    const size_t sub = (size_t)sub_lines * sub_cols;
    const size_t linepx = (size_t)hparts * sub_cols;
    for (int r = 0; r < vparts; ++r)
        for (int c = 0; c < hparts; ++c) {
            const uint16_t *t = tiles + ((size_t)r * hparts + c) * sub;
            uint16_t *stripe = out + (size_t)r * sub_lines * linepx;
            for (int line = 0; line < sub_lines; ++line)
                for (int col = 0; col < sub_cols; ++col) {
                    uint16_t w = t[(size_t)line * sub_cols + col];
                    w = (uint16_t)((w & 0x00FF) << 8 | (w & 0xFF00) >> 8);
                    stripe[(size_t)line * linepx + (size_t)c * sub_cols + col] = w;
                }
        }
}
