// host.cpp -- host-side pieces of the path that carry no raster arithmetic:
// the RRC parameter file loader and the shift filtering / polynomial fit.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oip_c.h"

// IMO::LoadRRCParamFile (imageop.h:140-192): three header lines (only the second -- the
// column count -- is checked outside DEBUG builds), then one "k , b" row per column parsed
// with sscanf(" %lf , %lf") from a 1024-byte fgets buffer; the row count must match exactly
// and any unparsable line (a trailing blank one included) is an error.  errno_error maps to
// OIP_E_IO, std::runtime_error to OIP_E_RUNTIME.
extern "C" int oip_load_rrc_param_file(const char *path, int expected_lines, double *kb_out, char *err, int errlen)
{
    auto fail = [&](int code, const char *fmt, auto... a) {
        if (err && errlen > 0) snprintf(err, errlen, fmt, a...);
        return code;
    };
    if (!path || expected_lines <= 0 || !kb_out) return fail(OIP_E_INVALID, "%s", "oip_load_rrc_param_file: bad argument");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(OIP_E_IO, "%s", "open RRC Param file failed");
    const int bn = 1024;
    char buff[bn];
    if (!fgets(buff, bn, f)) { fclose(f); return fail(OIP_E_IO, "%s", "LoadRRCParamFile([1]): read file content failed"); }
    if (!fgets(buff, bn, f)) { fclose(f); return fail(OIP_E_IO, "%s", "LoadRRCParamFile([2]): read file content failed"); }
    int lines = atoi(buff);
    if (lines != expected_lines) {
        fclose(f);
        return fail(OIP_E_RUNTIME, "LoadRRCParamFile([2]): expected %d lines while %d found in file content", expected_lines, lines);
    }
    if (!fgets(buff, bn, f)) { fclose(f); return fail(OIP_E_IO, "%s", "LoadRRCParamFile([3]): read file content failed"); }
    int index = 0;
    double k = .0, b = .0;
    for (; fgets(buff, bn, f); ++index) {
        if (sscanf(buff, " %lf , %lf", &k, &b) != 2) {
            fclose(f);
            return fail(OIP_E_RUNTIME, "line #%d of RRC param file [%s] found invalid", index, path);
        }
        if (index < expected_lines) {   // the reference overruns its array here; we only count
            kb_out[2 * index] = k;
            kb_out[2 * index + 1] = b;
        }
    }
    fclose(f);
    if (index != expected_lines)
        return fail(OIP_E_RUNTIME, "RRC Param file [%s] invalid: %d lines of param expected, %d lines parsed.", path, expected_lines, index);
    return OIP_OK;
}

// Least-squares polynomial, coefficients in ascending order, solved WELL: Householder QR on the
// abscissa centred and scaled to [-1,1], then the coefficients are expanded back to powers of x --
// the exact least-squares solution to fp64 accuracy (OIP_FIT_LSTSQ).  This is NOT what the reference
// computes (see oip_polyfit_reference below); it is offered as `--fit lstsq`.  Every rank calls this
// with identical inputs and gets identical bits (single thread, fixed order).
extern "C" int oip_polyfit(const double *x, const double *y, int n, int deg, double *coeffs)
{
    if (!x || !y || !coeffs || deg < 0 || deg > 8 || n <= deg) return OIP_E_INVALID;
    const int m = deg + 1;
    double mu = 0.0;
    for (int i = 0; i < n; ++i) mu += x[i];
    mu /= n;
    double sc = 0.0;
    for (int i = 0; i < n; ++i) sc = std::fmax(sc, std::fabs(x[i] - mu));
    if (sc == 0.0) sc = 1.0;
    std::vector<double> A((size_t)n * m), rhs(y, y + n);
    for (int i = 0; i < n; ++i) {
        double t = (x[i] - mu) / sc, p = 1.0;
        for (int j = 0; j < m; ++j) { A[(size_t)i * m + j] = p; p *= t; }
    }
    // Householder QR, applied to rhs on the fly
    for (int j = 0; j < m; ++j) {
        double norm = 0.0;
        for (int i = j; i < n; ++i) norm += A[(size_t)i * m + j] * A[(size_t)i * m + j];
        norm = std::sqrt(norm);
        if (norm == 0.0) return OIP_E_RUNTIME;
        double alpha = A[(size_t)j * m + j] > 0 ? -norm : norm;
        std::vector<double> v(n - j);
        for (int i = j; i < n; ++i) v[i - j] = A[(size_t)i * m + j];
        v[0] -= alpha;
        double vnorm2 = 0.0;
        for (double e : v) vnorm2 += e * e;
        if (vnorm2 == 0.0) continue;
        for (int c = j; c < m; ++c) {
            double dot = 0.0;
            for (int i = j; i < n; ++i) dot += v[i - j] * A[(size_t)i * m + c];
            double f = 2.0 * dot / vnorm2;
            for (int i = j; i < n; ++i) A[(size_t)i * m + c] -= f * v[i - j];
        }
        double dot = 0.0;
        for (int i = j; i < n; ++i) dot += v[i - j] * rhs[i];
        double f = 2.0 * dot / vnorm2;
        for (int i = j; i < n; ++i) rhs[i] -= f * v[i - j];
    }
    std::vector<double> c(m);
    for (int j = m - 1; j >= 0; --j) {
        double s = rhs[j];
        for (int k = j + 1; k < m; ++k) s -= A[(size_t)j * m + k] * c[k];
        double d = A[(size_t)j * m + j];
        if (d == 0.0) return OIP_E_RUNTIME;
        c[j] = s / d;
    }
    // sum_j c_j ((x-mu)/sc)^j  ->  ascending powers of x
    std::vector<double> p(m, 0.0);
    for (int j = 0; j < m; ++j) {
        double binom = 1.0;                       // C(j, i)
        double scj = std::pow(sc, j);
        for (int i = 0; i <= j; ++i) {
            if (i > 0) binom = binom * (j - i + 1) / i;
            p[i] += c[j] * binom * std::pow(-mu, j - i) / scj;
        }
    }
    for (int j = 0; j < m; ++j) coeffs[j] = p[j];
    return OIP_OK;
}

// nc::polynomial::Poly1d<double>::fit(x, y, deg) as the reference calls it (preproc.h:535-536), restated
// from NumCpp (un-vendored, version unpinned: PARITY UNPINNED):
//   A[i][j] = x_i^j by repeated multiplication (utils::power), raw abscissa;
//   non-square A:  aInv = inv(A^T A) . A^T ;  coefficients = aInv . y
//   NdArray::dot = one std::inner_product per element (k ascending, starting from 0);
//   linalg::inv  = in-place Gauss-Jordan sweep over the diagonal without pivoting (rows are only
//                  swapped for exactly-zero diagonal entries, which cannot occur for A^T A here).
// At cx up to 12288..30000 and degree 2 cond(A^T A) is 1e16..1e18, so the low digits of the result
// are set by this very operation order -- which is why it is reproduced operation by operation
// instead of being replaced by a better-conditioned solver: the product's default (OIP_FIT_REFERENCE)
// must give the maps the reference gives.  No FMA (the library is built -ffp-contract=off, like the
// reference's x86-64 build).
extern "C" int oip_polyfit_reference(const double *x, const double *y, int n, int deg, double *coeffs)
{
    if (!x || !y || !coeffs || deg < 0 || deg > 8 || n <= deg) return OIP_E_INVALID;
    const int m = deg + 1;
    std::vector<double> A((size_t)n * m);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double v = 1.0;
            if (j > 0) { v = x[i]; for (int e = 1; e < j; ++e) v *= x[i]; }
            A[(size_t)i * m + j] = v;
        }
    // aT.dot(a)
    std::vector<double> G((size_t)m * m), R((size_t)m * m);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = acc + A[(size_t)k * m + i] * A[(size_t)k * m + j];
            G[(size_t)i * m + j] = acc;
        }
    // linalg::inv: sweep k = 0..m-1, each sweep builds `result` from the current matrix and replaces it
    for (int k = 0; k < m; ++k) {
        if (G[(size_t)k * m + k] == 0.0) return OIP_E_RUNTIME;      // degenerate abscissae
        R[(size_t)k * m + k] = -1.0 / G[(size_t)k * m + k];
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < m; ++j) {
                if (i != k && j != k) R[(size_t)i * m + j] = G[(size_t)i * m + j] + G[(size_t)k * m + j] * G[(size_t)i * m + k] * R[(size_t)k * m + k];
                else if (i != k && j == k) R[(size_t)i * m + k] = G[(size_t)i * m + k] * R[(size_t)k * m + k];
                else if (i == k && j != k) R[(size_t)k * m + j] = G[(size_t)k * m + j] * R[(size_t)k * m + k];
            }
        G = R;
    }
    for (double &v : R) v *= -1.0;
    // aTaInv.dot(aT), then .dot(y)
    std::vector<double> P((size_t)m * n);
    for (int i = 0; i < m; ++i)
        for (int c = 0; c < n; ++c) {
            double acc = 0.0;
            for (int k = 0; k < m; ++k) acc = acc + R[(size_t)i * m + k] * A[(size_t)c * m + k];
            P[(size_t)i * n + c] = acc;
        }
    for (int i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int c = 0; c < n; ++c) acc = acc + P[(size_t)i * n + c] * y[c];
        if (!(acc == acc) || std::isinf(acc)) return OIP_E_RUNTIME;
        coeffs[i] = acc;
    }
    return OIP_OK;
}

// FilterInterBandShiftValues (preproc.h:492-512) + DoCorrelationPolynomialFitting
// (preproc.h:514-550): per band keep the shifts whose response reaches the threshold,
// require at least min_count of them, fit dx(cx) with degree 1 and dy(cx) with degree 2.
extern "C" int oip_filter_and_fit_mode(const double *shifts, int n, double threshold, int min_count, int fit_mode,
                                       double *cx_out, double *cy_out, char *err, int errlen)
{
    if (!shifts || n <= 0 || !cx_out || !cy_out || (fit_mode != OIP_FIT_REFERENCE && fit_mode != OIP_FIT_LSTSQ)) return OIP_E_INVALID;
    auto fit = fit_mode == OIP_FIT_LSTSQ ? oip_polyfit : oip_polyfit_reference;
    std::vector<double> cxv(n), xv(n), yv(n);
    for (int b = 0; b < OIP_MSS_BANDS; ++b) {
        // Two tests, as the reference has them: FilterInterBandShiftValues counts a unit unless `rs < threshold`
        // (preproc.h:498-503) -- a NaN response, which the f32 cross-power spectrum produces when |P|^2 overflows, is
        // not below the threshold and counts towards min_count -- while DoCorrelationPolynomialFitting takes a unit
        // into the fit when `rs >= threshold` (preproc.h:527), which a NaN fails.
        int vvi = 0, fc = 0;
        for (int i = 0; i < n; ++i) {
            const double *s = shifts + ((size_t)b * n + i) * 4;
            if (!(s[2] < threshold)) ++fc;
            if (s[2] >= threshold) {
                cxv[vvi] = s[3];
                xv[vvi] = s[0];
                yv[vvi] = s[1];
                ++vvi;
            }
        }
        if (fc < min_count) {
            if (err && errlen > 0)
                snprintf(err, errlen, "Not enough valid correlation values for band#%d: %d valid values found, %d expected at least",
                         b + 1, fc, min_count);
            return OIP_E_RUNTIME;
        }
        int rc = fit(cxv.data(), xv.data(), vvi, 1, cx_out + b * 2);
        if (rc == OIP_OK) rc = fit(cxv.data(), yv.data(), vvi, 2, cy_out + b * 3);
        if (rc != OIP_OK) {
            if (err && errlen > 0) snprintf(err, errlen, "polynomial fit failed for band#%d (degenerate abscissae)", b + 1);
            return OIP_E_RUNTIME;
        }
    }
    return OIP_OK;
}

extern "C" int oip_filter_and_fit(const double *shifts, int n, double threshold, int min_count, double *cx_out,
                                  double *cy_out, char *err, int errlen)
{
    return oip_filter_and_fit_mode(shifts, n, threshold, min_count, OIP_FIT_REFERENCE, cx_out, cy_out, err, errlen);
}

// The accumulation of Stitcher::CalcSttParameters (stitcher.h:181-198): sections in order, a section is
// valid when resp >= threshold and (max_delta_y <= 0 or |dy| <= max_delta_y); arithmetic means in fp64.
// NaN rows (sections no rank computed) fail the comparison like any low response.  Every rank of a
// multi-GPU run calls this on the all-gathered table and gets identical bits.
extern "C" int oip_stt_mean(const double *table, int sections, double threshold, double max_delta_y, double *dx,
                            double *dy, double *response, int *valid_out)
{
    if (!table || sections <= 0) return OIP_E_INVALID;
    double sx = 0.0, sy = 0.0, sr = 0.0;
    int valid = 0;
    for (int i = 0; i < sections; ++i) {
        const double x = table[3 * i], y = table[3 * i + 1], r = table[3 * i + 2];
        const bool ok = r >= threshold && (max_delta_y <= 0.0 || std::fabs(y) <= max_delta_y);
        if (ok) { sx += x; sy += y; sr += r; ++valid; }
    }
    if (valid_out) *valid_out = valid;
    if (valid == 0) return OIP_E_RUNTIME;       // "No valid delta value found for stitching parameter calculating"
    if (dx) *dx = sx / valid;
    if (dy) *dy = sy / valid;
    if (response) *response = sr / valid;
    return OIP_OK;
}

// cv::resize(INTER_CUBIC) by exactly 4 along one axis as an operator on spectra: with R = C + E the
// n -> N = 4 n up-sampling matrix (C circulant: zero-stuff, convolve with the 16-tap kernel h; E: what clamping instead
// of wrapping the out-of-image taps adds, non-zero in columns J = {0, 1, n-2, n-1} only),
//     DFT_N(R s)[k] = H[k] DFT_n(s)[k mod n] + sum_j G_j[k] s[J_j],     H = DFT_N(h), G_j = DFT_N(E[:, J_j]).
// out[(t * N + k) * 2 + {0,1}] = re, im of H (t = 0) and G_0..G_3 (t = 1..4), as float.  Built in double from the f32
// taps of cv::hal::resize's coefficient set-up (fx = (float)((d + 0.5) * scale - 0.5), interpolateCubic(fx - floor)),
// the same the image-domain kernels apply.  OIP_E_UNSUPPORTED when the taps are not 4-periodic or n < 8.
static inline void host_interpolate_cubic(float x, float *c)
{
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
extern "C" int oip_upsample_operator(int n, float *out)
{
    if (n < 8 || !out) return OIP_E_UNSUPPORTED;
    const int N = 4 * n;
    const double scale = 1. / ((double)N / n);
    std::vector<double> h(N, 0.0), g[4];
    std::vector<char> hset(N, 0);
    for (auto &v : g) v.assign(N, 0.0);
    const int J[4] = {0, 1, n - 2, n - 1};
    auto jidx = [&](int p) { for (int j = 0; j < 4; ++j) if (J[j] == p) return j; return -1; };
    for (int d = 0; d < N; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)floorf(fx);
        fx -= sx;
        float w[4];
        host_interpolate_cubic(fx, w);
        for (int j = 0; j < 4; ++j) {
            const int p = sx - 1 + j;
            const int e = ((d - 4 * p) % N + N) % N;
            if (!hset[e]) { h[e] = w[j]; hset[e] = 1; }
            else if (h[e] != (double)w[j]) return OIP_E_UNSUPPORTED;     // taps not periodic in d: no circulant part
            if (p < 0 || p >= n) {
                const int a = jidx(p < 0 ? 0 : n - 1), b = jidx(((p % n) + n) % n);
                if (a < 0 || b < 0) return OIP_E_UNSUPPORTED;
                g[a][d] += w[j];
                g[b][d] -= w[j];
            }
        }
    }
    const double step = -2.0 * 3.14159265358979323846 / N;
    auto dft = [&](const std::vector<double> &v, float *dst) {
        std::vector<int> nz;
        for (int d = 0; d < N; ++d) if (v[d] != 0.0) nz.push_back(d);
        for (int k = 0; k < N; ++k) {
            double re = 0.0, im = 0.0;
            for (int d : nz) {
                const double ang = step * (double)(((long)k * d) % N);
                re += v[d] * cos(ang);
                im += v[d] * sin(ang);
            }
            dst[2 * k] = (float)re;
            dst[2 * k + 1] = (float)im;
        }
    };
    dft(h, out);
    for (int j = 0; j < 4; ++j) dft(g[j], out + (size_t)(1 + j) * N * 2);
    return OIP_OK;
}
