// oip_geom.h -- section geometry of the two sectioned remaps, shared by host and device.
//
// Both reference drivers cut the strip into line sections because cv::remap addresses
// source pixels with `short` coordinates (imageop.h:19-20, preproc.h:349-361).  Each output
// line therefore belongs to exactly one section, its map value is relative to that
// section's first line, and taps falling outside the section's buffer read the constant
// border (0).  These helpers turn a global output line into (section, relative line,
// source-line mapping) so that a kernel -- or a row-block shard of it -- reproduces the
// per-section results without ever materialising a section buffer or a float map.
#pragma once

#include <hip/hip_runtime.h>

// ---- Stitcher::PreStitch + IMO::SectionaryRemap (stitcher.h:83-139, imageop.h:230-275) ----
struct OipShiftGeom {
    int W;
    long L;
    double dx, dy;
    int section_rows;     // REMAP_SECTION_ROWS: rows of buff / mapx / mapy / dst
    int ucut, bcut;       // stitcher.h:122-123
    long step;            // section_rows - (ucut + bcut): advance per section (imageop.h:266)
    int nsec;             // sections the loop processes (imageop.h:249-267)
};

struct OipShiftRow {
    int fy;               // y phase 0..31
    int iy;               // first tap's buffer row (section relative), already -1
    long src[4];          // global source line of each vertical tap, -1 = outside buffer (zero)
};

__host__ __device__ inline OipShiftGeom oip_shift_geom(int W, long L, double dx, double dy, int section_rows)
{
    OipShiftGeom g;
    g.W = W; g.L = L; g.dx = dx; g.dy = dy; g.section_rows = section_rows;
    g.ucut = dy >= 0.0 ? 0 : (int)(-dy) + 1;
    g.bcut = dy >= 0.0 ? (int)dy + 1 : 0;
    int total_cut = g.ucut + g.bcut;
    g.step = (long)section_rows - total_cut;
    // for (s=0;;++s) { rows=min(SR, L-off); if (rows<=cut) break; ...; off += rows-cut; }
    int n = 0;
    if (g.step > 0) {
        long off = 0;
        for (;;) {
            long rows = (long)section_rows < L - off ? (long)section_rows : L - off;
            if (rows <= total_cut) break;
            ++n;
            off += rows - total_cut;
        }
    }
    g.nsec = n;
    return g;
}

// rows the section's fread delivered (stitcher.h:103-111)
__host__ __device__ inline long oip_shift_section_rows(const OipShiftGeom &g, int s)
{
    long off = (long)s * g.step;
    return (long)g.section_rows < g.L - off ? (long)g.section_rows : g.L - off;
}

// Output line `gy` (0 <= gy < L) of the .PRESTT raster -> section, relative line, taps.
//   [0, ucut)            : top rows of section 0's dst        (imageop.h:260-263)
//   [s*step+ucut, ...)   : dst rows [ucut, rows-bcut) of section s   (imageop.h:265)
//   [L-bcut, L)          : last dst's rows [SR-bcut, SR)      (imageop.h:269-272)
// Source buffer row r of section s holds global line s*step + r for r < rows_s; in a short
// last section rows_s <= r < SR still hold the previous section's lines (s-1)*step + r
// (SURVEY App.B-4: get_src returns the whole reused buffer).
__host__ __device__ inline OipShiftRow oip_shift_row(const OipShiftGeom &g, long gy)
{
    OipShiftRow o;
    int s;
    long yr;
    if (gy < g.ucut) {
        s = 0;
        yr = gy;
    } else if (gy >= g.L - g.bcut) {
        s = g.nsec - 1;
        yr = (long)g.section_rows - g.bcut + (gy - (g.L - g.bcut));
    } else {
        long q = (gy - g.ucut) / g.step;
        s = q > g.nsec - 1 ? g.nsec - 1 : (int)q;
        yr = gy - (long)s * g.step;
    }
    float mapy = (float)((double)yr + g.dy);            // stitcher.h:97 (y is section relative)
#if defined(__HIP_DEVICE_COMPILE__)
    int sy = (int)__builtin_rintf(mapy * 32.0f);
#else
    int sy = (int)__builtin_rintf(mapy * 32.0f);
#endif
    int iy = sy >> 5;
    iy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    iy -= 1;
    o.fy = sy & 31;
    o.iy = iy;
    long rows_s = oip_shift_section_rows(g, s);
    for (int t = 0; t < 4; ++t) {
        long r = (long)iy + t;
        if (r < 0 || r >= g.section_rows) o.src[t] = -1;
        else if (r < rows_s) o.src[t] = (long)s * g.step + r;
        else o.src[t] = s > 0 ? (long)(s - 1) * g.step + r : -1;
    }
    return o;
}

// ---- PreProcessor::DoInterBandAlignment (preproc.h:351-468) ----------------------------
struct OipAlignGeom {
    int Wb;
    long Lm;
    int lps, line_offset, overlap, keep_leading, min_lines;
    long adv;             // lps - overlap
    int nsec;             // sections processed (preproc.h:379-408)
    long out_rows;        // rows of mAlignedMSS (preproc.h:375-376)
    long rows_valid;      // processedLines after the loop
};

__host__ __device__ inline OipAlignGeom oip_align_geom(int Wb, long Lm, int lps, int line_offset, int overlap,
                                                       int keep_leading, int min_lines)
{
    OipAlignGeom g;
    g.Wb = Wb; g.Lm = Lm; g.lps = lps; g.line_offset = line_offset; g.overlap = overlap;
    g.keep_leading = keep_leading; g.min_lines = min_lines;
    g.adv = (long)lps - overlap;
    g.out_rows = Lm - line_offset - (keep_leading ? 0 : overlap);
    if (g.out_rows < 0) g.out_rows = 0;
    int n = 0;
    long processed = 0;
    long offset = line_offset;
    for (;;) {
        if (Lm < offset) break;
        long lines = Lm - offset < (long)lps ? Lm - offset : (long)lps;
        if (lines < min_lines) break;
        if (n == 0 && keep_leading) processed += overlap;
        processed += lines - overlap;
        offset += g.adv;
        ++n;
    }
    g.nsec = n;
    g.rows_valid = processed;
    return g;
}

// output row -> (section i, section-relative line y, section's first global line, lines)
__host__ __device__ inline bool oip_align_row(const OipAlignGeom &g, long orow, int *sec, long *yrel, long *sec_off,
                                              long *sec_lines)
{
    long o = orow;
    int i;
    long y;
    if (g.keep_leading) {
        if (o < g.overlap) { i = 0; y = o; }
        else { o -= g.overlap; i = (int)(o / g.adv); y = o - (long)i * g.adv + g.overlap; }
    } else {
        i = (int)(o / g.adv);
        y = o - (long)i * g.adv + g.overlap;
    }
    if (i >= g.nsec) return false;
    long off = (long)g.line_offset + (long)i * g.adv;
    long lines = g.Lm - off < (long)g.lps ? g.Lm - off : (long)g.lps;
    if (y >= lines) return false;
    *sec = i; *yrel = y; *sec_off = off; *sec_lines = lines;
    return true;
}
