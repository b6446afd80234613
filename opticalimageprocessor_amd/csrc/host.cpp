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

// nc::polynomial::Poly1d<double>::fit (call sites preproc.h:535-536): least-squares
// polynomial, coefficients in ascending order.  NumCpp solves it on the raw Vandermonde
// matrix, which at cx up to 12288..30000 and degree 2 is conditioned ~1e16..1e18, so its low
// digits are an artefact of its own inverse.  Here: Householder QR on the abscissa centred
// and scaled to [-1,1], then the coefficients are expanded back to powers of x -- the exact
// least-squares solution to fp64 accuracy.  Every rank calls this with identical inputs and
// gets identical bits (single thread, fixed order).
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

// FilterInterBandShiftValues (preproc.h:492-512) + DoCorrelationPolynomialFitting
// (preproc.h:514-550): per band keep the shifts whose response reaches the threshold,
// require at least min_count of them, fit dx(cx) with degree 1 and dy(cx) with degree 2.
extern "C" int oip_filter_and_fit(const double *shifts, int n, double threshold, int min_count, double *cx_out,
                                  double *cy_out, char *err, int errlen)
{
    if (!shifts || n <= 0 || !cx_out || !cy_out) return OIP_E_INVALID;
    std::vector<double> cxv(n), xv(n), yv(n);
    for (int b = 0; b < OIP_MSS_BANDS; ++b) {
        int vvi = 0;
        for (int i = 0; i < n; ++i) {
            const double *s = shifts + ((size_t)b * n + i) * 4;
            if (s[2] >= threshold) {            // NaN responses (sections another rank owns) fail this test
                cxv[vvi] = s[3];
                xv[vvi] = s[0];
                yv[vvi] = s[1];
                ++vvi;
            }
        }
        if (vvi < min_count) {
            if (err && errlen > 0)
                snprintf(err, errlen, "Not enough valid correlation values for band#%d: %d valid values found, %d expected at least",
                         b + 1, vvi, min_count);
            return OIP_E_RUNTIME;
        }
        int rc = oip_polyfit(cxv.data(), xv.data(), vvi, 1, cx_out + b * 2);
        if (rc == OIP_OK) rc = oip_polyfit(cxv.data(), yv.data(), vvi, 2, cy_out + b * 3);
        if (rc != OIP_OK) {
            if (err && errlen > 0) snprintf(err, errlen, "polynomial fit failed for band#%d (degenerate abscissae)", b + 1);
            return OIP_E_RUNTIME;
        }
    }
    return OIP_OK;
}
