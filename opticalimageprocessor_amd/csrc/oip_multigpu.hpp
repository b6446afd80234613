// oip_multigpu.hpp -- the two strip work-flows on the N GPUs of one node, in C++ over RCCL.
//
// `oip --gpus N ...` (default action, BASELINE config 4) and `oip prestitch --gpus N ...` (cross-CCD path,
// BASELINE config 5).  One process, one host thread and one oip_ctx per GPU, one RCCL communicator per GPU
// (ncclCommInitAll); librccl is loaded by the first --gpus run (dlopen, below): neither the CLI nor liboipgpu.so links it.  The strip is cut into
// scan-line blocks: rank r owns lines [r L/N, (r+1) L/N) of every raster and reads exactly those
// bytes of the input files.  Three exchange steps, none of them a reduction over pixels -- the same plan as
// opticalimageprocessor_amd/dist.py, whose gloo tests check it against the single-process result bit for bit;
// tests/test_cli_cpu.py compares the two plans through `oip plan`:
//   1. correlation windows: the reference correlates a fixed number of windows per strip (preproc.h:245-259,
//      stitcher.h:151-168); the (section, slice) units are placed by predicted cost (assign_groups_by_cost: a pair
//      moves only when bytes / link bandwidth beats computing it at home) and the lines a unit's rank lacks arrive as
//      compact windows: one grouped ncclSend / ncclRecv per pair on a communication stream, under the resident
//      pairs' kernels; a received pair is computed behind its own event;
//   2. ncclAllGather of the per-unit results, then the identical fixed-order host step on every rank
//      (filter + polynomial fit, or the CCD shift mean): bit-identical maps everywhere;
//   3. resampling halo (oip_align_mss_src_range / oip_remap_shift_src_range): whole lines, ncclSend / ncclRecv.
// Section seams come from GLOBAL line indices, so the blocks put together are the single-GPU product.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <functional>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <thread>

#include <dlfcn.h>

#include "oip_host.hpp"
#include "oip_rankguard.hpp"

// RCCL is loaded when the first --gpus run asks for it, not with the executable: a single-GPU `oip` run then depends on
// libamdhip64 alone (what liboipgpu.so depends on), and a machine without RCCL can still run it.  It is not a start-up
// optimisation: an executable linked against the 570 MB librccl.so starts in 13.8 ms, this one in 12.5 ms
// (profiles/experiments/rccl_link_probe, `--version`, best of five each).  Types come from <rccl/rccl.h>; the entry points
// are resolved from librccl.so.1 on first use.
namespace OIPGPU {
struct RcclApi {
    decltype(&::ncclCommInitAll) CommInitAll;
    decltype(&::ncclCommDestroy) CommDestroy;
    decltype(&::ncclCommAbort) CommAbort;
    decltype(&::ncclGroupStart) GroupStart;
    decltype(&::ncclGroupEnd) GroupEnd;
    decltype(&::ncclSend) Send;
    decltype(&::ncclRecv) Recv;
    decltype(&::ncclAllGather) AllGather;
    decltype(&::ncclGetErrorString) GetErrorString;
    static RcclApi &get()
    {
        static RcclApi a;
        return a;
    }
private:
    RcclApi()
    {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) throw std::runtime_error(std::string("--gpus needs RCCL: ") + dlerror());
        auto sym = [&](const char *name) {
            void *p = dlsym(h, name);
            if (!p) throw std::runtime_error(std::string("librccl.so.1 lacks ") + name);
            return p;
        };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    }
};
}  // namespace OIPGPU
#define ncclCommInitAll ::OIPGPU::RcclApi::get().CommInitAll
#define ncclCommDestroy ::OIPGPU::RcclApi::get().CommDestroy
#define ncclCommAbort ::OIPGPU::RcclApi::get().CommAbort
#define ncclGroupStart ::OIPGPU::RcclApi::get().GroupStart
#define ncclGroupEnd ::OIPGPU::RcclApi::get().GroupEnd
#define ncclSend ::OIPGPU::RcclApi::get().Send
#define ncclRecv ::OIPGPU::RcclApi::get().Recv
#define ncclAllGather ::OIPGPU::RcclApi::get().AllGather
#define ncclGetErrorString ::OIPGPU::RcclApi::get().GetErrorString

namespace OIPGPU {

// ---- plans (pure arithmetic; mirrors dist.py's assign_groups_by_cost / StripPlan / CcdPlan) ---------------------------------
struct Piece {
    int src, dst;
    int kind;           // 0 pan (or pan1), 1 mss (all four planes) (or pan2)
    int unit;
    long row0, rows;
    int col0, cols;
    long dst_row;
};
struct LineTransfer {
    int src, dst;
    long row0, rows;
};

// ---- cost-aware placement (mirrors dist.py's assign_groups_by_cost operation by operation; integer microseconds) -----
// A group (a pair of inter-band units, or one CCD section) costs compute_us wherever it runs, plus the time to bring in the
// window bytes that rank lacks: they arrive over one xGMI link at bytes_per_us, one group after the other from time 0, under
// the resident groups' kernels; a rank computes its resident groups first, then the others in index order as they arrive.
// Every group starts where most of its bytes are; then single groups move, best move first, while a move lowers the ranks'
// finish times sorted from the latest down.  The constants are dist.py's (LINK_GBS etc.; unmeasured on hardware).
// OIP_LINK_GBS (an integer, GB/s) overrides the link figure in both hosts; `oip plan` prints what was used.
constexpr long kPairUs16000x3000 = 2500, kCcdSectionUs16000x200 = 150;
inline long link_gbs()
{
    const char *e = getenv("OIP_LINK_GBS");
    const long v = e ? atol(e) : 50;
    return v > 0 ? v : 50;
}

inline long rank_finish_us(const std::vector<long> &costs, long compute_us, long bytes_per_us)
{
    long t = 0, arrived = 0;
    for (long c : costs) if (c == 0) t += compute_us;
    for (long c : costs)
        if (c) {
            arrived += (c + bytes_per_us - 1) / bytes_per_us;
            t = std::max(t, arrived) + compute_us;
        }
    return t;
}

inline std::vector<int> assign_groups_by_cost(const std::vector<std::vector<long>> &missing, int world, long compute_us, long bytes_per_us,
                                              std::vector<long> *finish_out = nullptr)
{
    const int ng = (int)missing.size();
    std::vector<int> where(ng, 0);
    for (int g = 0; g < ng; ++g)
        for (int q = 1; q < world; ++q) if (missing[g][q] < missing[g][where[g]]) where[g] = q;
    auto finish = [&](int r, const std::vector<int> &w) {
        std::vector<long> costs;
        for (int g = 0; g < ng; ++g) if (w[g] == r) costs.push_back(missing[g][r]);
        return rank_finish_us(costs, compute_us, bytes_per_us);
    };
    std::vector<long> fin(world);
    for (int r = 0; r < world; ++r) fin[r] = finish(r, where);
    auto sorted_desc = [](std::vector<long> v) { std::sort(v.begin(), v.end(), std::greater<long>()); return v; };
    for (int it = 0; it < 4 * ng * world; ++it) {
        const std::vector<long> cur = sorted_desc(fin);
        bool have = false;
        std::vector<long> best_key, best_fin;
        int best_g = -1, best_q = -1;
        for (int g = ng - 1; g >= 0; --g) {
            const int r = where[g];
            for (int q = 0; q < world; ++q) {
                if (q == r) continue;
                std::vector<int> w2 = where;
                w2[g] = q;
                std::vector<long> f2 = fin;
                f2[r] = finish(r, w2);
                f2[q] = finish(q, w2);
                const std::vector<long> key = sorted_desc(f2);
                if (key < cur && (!have || key < best_key)) { have = true; best_key = key; best_fin = f2; best_g = g; best_q = q; }
            }
        }
        if (!have) break;
        where[best_g] = best_q;
        fin = best_fin;
    }
    if (finish_out) *finish_out = fin;
    return where;
}

// bytes (u16) of a window of `rows` lines x `cols` columns x `planes` planes starting at line row0 that each rank lacks
inline void add_missing_bytes(std::vector<long> *out, long row0, long rows, int cols, int planes, long block, int world)
{
    for (int q = 0; q < world; ++q) {
        const long held = std::max(0L, std::min(row0 + rows, (q + 1) * block) - std::max(row0, q * block));
        (*out)[q] += (rows - held) * cols * planes * 2;
    }
}

inline void window_pieces(std::vector<Piece> *out, int kind, int unit, int dst, long row0, long rows, int col0, int cols,
                          long block, int world)
{
    for (int r = 0; r < world; ++r) {
        const long lo = std::max(row0, r * block), hi = std::min(row0 + rows, (r + 1) * block);
        if (lo < hi) out->push_back(Piece{r, dst, kind, unit, lo, hi - lo, col0, cols, lo - row0});
    }
}

struct StripPlanC {
    int W, world, slices, sections, corr_lines, lps, line_offset, overlap, keep, min_lines, halo_cap;
    long Lp, Lm, pb, mb, base_gap, band_gap, out_rows;
    int base_rows, band_rows, base_cols, band_cols, n_units;
    std::vector<int> assign;

    StripPlanC(int W_, long Lp_total, int world_, int slices_ = OIP_IBCV_DEF_SLICES, int sections_ = OIP_IBCV_DEF_SECTIONS,
               int corr = OIP_CORRELATION_LINES, int lps_ = OIP_IBPA_DEFAULT_BATCHLINES, int line_offset_ = 0,
               int overlap_ = OIP_IBPA_DEFAULT_LINEOVERLAP, bool keep_ = false, int min_lines_ = OIP_IBPA_MIN_PROCESSLINES,
               int halo_cap_ = 64)
        : W(W_), world(world_), slices(slices_), sections(sections_), corr_lines(corr), lps(lps_), line_offset(line_offset_),
          overlap(overlap_), keep(keep_), min_lines(min_lines_), halo_cap(halo_cap_), Lp(Lp_total)
    {
        if (Lp_total % (4L * world)) throw std::invalid_argument("PAN line count must be a multiple of 4 x the GPU count");
        if (sections > 1 && (long)sections * corr > Lp_total)                         // preproc.h:234-237
            throw std::invalid_argument("CalcInterBandCorrelation: too many sections, not enough total PAN data lines");
        Lm = Lp / 4; pb = Lp / world; mb = pb / 4;
        base_rows = (int)std::min<long>(Lp, corr);                                    // preproc.h:245-247, :274-276
        base_gap = (Lp - (long)base_rows * sections) / (sections + 1);
        band_rows = base_rows / 4; band_gap = base_gap / 4;
        base_cols = W / slices; band_cols = base_cols / 4;
        out_rows = Lm - line_offset - (keep ? 0 : overlap);
        n_units = sections * slices;
        // placement by predicted cost; pairs of units stay together (the kernels process two units per launch)
        const long compute_us = std::max(1L, kPairUs16000x3000 * base_rows * base_cols / (16000L * 3000L));
        std::vector<std::vector<long>> missing;
        for (int g = 0; g < n_units; g += 2) {
            std::vector<long> m(world, 0);
            for (int u = g; u < std::min(g + 2, n_units); ++u) {
                long p0, m0;
                section(u / slices, &p0, &m0);
                add_missing_bytes(&m, p0, base_rows, base_cols, 1, pb, world);
                add_missing_bytes(&m, m0, band_rows, band_cols, 4, mb, world);
            }
            missing.push_back(m);
        }
        const std::vector<int> where = assign_groups_by_cost(missing, world, compute_us, link_gbs() * 1000, &predicted_finish_us);
        assign.resize(n_units);
        for (int u = 0; u < n_units; ++u) assign[u] = where[u / 2];
    }
    std::vector<long> predicted_finish_us;
    void section(int sec, long *p0, long *m0) const
    {
        *p0 = base_gap + (long)sec * (base_rows + base_gap);
        *m0 = band_gap + (long)sec * (band_rows + band_gap);
    }
    int owner(int sec) const
    {
        long p0, m0;
        section(sec, &p0, &m0);
        return (int)std::min<long>(p0 / pb, world - 1);
    }
    std::vector<int> units_of(int r) const
    {
        std::vector<int> v;
        for (int u = 0; u < n_units; ++u) if (assign[u] == r) v.push_back(u);
        return v;
    }
    std::vector<Piece> unit_pieces(int u) const
    {
        std::vector<Piece> out;
        long p0, m0;
        section(u / slices, &p0, &m0);
        const int i = u % slices;
        window_pieces(&out, 0, u, assign[u], p0, base_rows, i * base_cols, base_cols, pb, world);
        window_pieces(&out, 1, u, assign[u], m0, band_rows, i * band_cols, band_cols, mb, world);
        return out;
    }
    bool unit_is_local(int u) const
    {
        for (const Piece &p : unit_pieces(u)) if (p.src != p.dst) return false;
        return true;
    }
    std::vector<Piece> correlation_pieces() const
    {
        std::vector<Piece> out;
        for (int u = 0; u < n_units; ++u)
            if (!unit_is_local(u)) for (const Piece &p : unit_pieces(u)) out.push_back(p);
        return out;
    }
    // the same pieces pair by pair, in unit order: the posting order of the overlapped exchange (identical on all ranks)
    std::vector<std::pair<std::vector<int>, std::vector<Piece>>> exchange_groups() const
    {
        std::vector<std::pair<std::vector<int>, std::vector<Piece>>> out;
        for (int g = 0; g < n_units; g += 2) {
            std::vector<int> us;
            std::vector<Piece> ps;
            for (int u = g; u < std::min(g + 2, n_units); ++u)
                if (!unit_is_local(u)) { us.push_back(u); for (const Piece &p : unit_pieces(u)) ps.push_back(p); }
            if (!us.empty()) out.push_back({us, ps});
        }
        return out;
    }
    void align_out_rows(int r, long *o0, long *o1) const
    {
        const long shift = line_offset + (keep ? 0 : overlap);
        const long b0 = r * mb, b1 = (r + 1) * mb;
        *o0 = r == 0 ? 0 : std::min(std::max(b0 - shift, 0L), out_rows);
        *o1 = r == world - 1 ? out_rows : std::min(std::max(b1 - shift, 0L), out_rows);
        if (*o1 < *o0) *o1 = *o0;
    }
    // need[r] = [first, last) MSS lines rank r's output rows read
    std::vector<LineTransfer> align_transfers(const double *cy, std::vector<std::pair<long, long>> *need) const
    {
        std::vector<LineTransfer> out;
        need->clear();
        for (int r = 0; r < world; ++r) {
            long o0, o1, f = 0, l = 0;
            align_out_rows(r, &o0, &o1);
            if (o1 > o0 && oip_align_mss_src_range(o0, o1 - o0, Lm, cy, W / 4, lps, line_offset, overlap, keep, min_lines, &f, &l) != OIP_OK)
                throw std::runtime_error("oip_align_mss_src_range failed");
            need->push_back({f, l});
            for (int q = 0; q < world; ++q) {
                if (q == r || l <= f) continue;
                const long lo = std::max(f, q * mb), hi = std::min(l, (q + 1) * mb);
                if (lo < hi) out.push_back(LineTransfer{q, r, lo, hi - lo});
            }
        }
        return out;
    }
};

struct CcdPlanC {
    int W, world, sections, lps, ov, edge, cols, section_rows, row_guard, fold, n_units;
    long L, pb, gap, step;
    std::vector<int> assign;

    CcdPlanC(int W_, long L_, int world_, int sections_ = OIP_STT_DEF_SECTIONS, int lps_ = OIP_STT_DEF_SECLINES,
             int ov_ = OIP_STT_DEF_OVERLAPPX, int edge_ = 0, int section_rows_ = OIP_REMAP_SECTION_ROWS,
             int row_guard_ = OIP_REMAP_ROW_GUARD)
        : W(W_), world(world_), sections(sections_), lps(lps_), ov(ov_), edge(edge_), cols(ov_ - edge_), section_rows(section_rows_),
          row_guard(row_guard_), fold(ov_ / 2), n_units(sections_), L(L_)
    {
        if (L % world) throw std::invalid_argument("line count must be a multiple of the GPU count");
        if (L < (long)sections * lps)                                                // stitcher.h:75-77
            throw std::invalid_argument("PAN line count less than sections times line-per-section, use smaller -s and/or -l value(s)");
        pb = L / world;
        gap = (L - (long)sections * lps) / (sections + 1);                            // stitcher.h:151-152, :167
        step = gap + lps;
        const long compute_us = std::max(1L, kCcdSectionUs16000x200 * lps * cols / (16000L * 200L));
        std::vector<std::vector<long>> missing;
        for (int s = 0; s < sections; ++s) {
            std::vector<long> m(world, 0);
            add_missing_bytes(&m, gap + (long)s * step, lps, cols, 2, pb, world);
            missing.push_back(m);
        }
        assign = assign_groups_by_cost(missing, world, compute_us, link_gbs() * 1000, &predicted_finish_us);
    }
    std::vector<long> predicted_finish_us;
    long section_start(int s) const { return gap + (long)s * step; }
    std::vector<int> units_of(int r) const
    {
        std::vector<int> v;
        for (int u = 0; u < n_units; ++u) if (assign[u] == r) v.push_back(u);
        return v;
    }
    std::vector<Piece> unit_pieces(int u) const
    {
        std::vector<Piece> out;
        const long a = section_start(u);
        window_pieces(&out, 0, u, assign[u], a, lps, W - ov, cols, pb, world);       // stitcher.h:175: PAN1 cols [W-ov, W-edge)
        window_pieces(&out, 1, u, assign[u], a, lps, edge, cols, pb, world);         // stitcher.h:176: PAN2 cols [edge, ov)
        return out;
    }
    bool unit_is_local(int u) const
    {
        for (const Piece &p : unit_pieces(u)) if (p.src != p.dst) return false;
        return true;
    }
    std::vector<Piece> correlation_pieces() const
    {
        std::vector<Piece> out;
        for (int u = 0; u < n_units; ++u)
            if (!unit_is_local(u)) for (const Piece &p : unit_pieces(u)) out.push_back(p);
        return out;
    }
    std::vector<std::pair<std::vector<int>, std::vector<Piece>>> exchange_groups() const
    {
        std::vector<std::pair<std::vector<int>, std::vector<Piece>>> out;
        for (int u = 0; u < n_units; ++u)
            if (!unit_is_local(u)) out.push_back({std::vector<int>{u}, unit_pieces(u)});
        return out;
    }
    std::vector<LineTransfer> remap_transfers(double dy, std::vector<std::pair<long, long>> *need) const
    {
        std::vector<LineTransfer> out;
        need->clear();
        for (int r = 0; r < world; ++r) {
            long f = 0, l = 0;
            if (oip_remap_shift_src_range(r * pb, pb, L, dy, section_rows, &f, &l) != OIP_OK)
                throw std::runtime_error("oip_remap_shift_src_range failed");
            need->push_back({f, l});
            for (int q = 0; q < world; ++q) {
                if (q == r) continue;
                const long lo = std::max(f, q * pb), hi = std::min(l, (q + 1) * pb);
                if (lo < hi) out.push_back(LineTransfer{q, r, lo, hi - lo});
            }
        }
        return out;
    }
};

// `oip plan`: the plans as JSON, for the test that compares them with dist.py's
inline void print_pieces(const std::vector<Piece> &ps)
{
    printf("[");
    for (size_t i = 0; i < ps.size(); ++i)
        printf("%s[%d,%d,%d,%d,%ld,%ld,%d,%d,%ld]", i ? "," : "", ps[i].src, ps[i].dst, ps[i].kind, ps[i].unit, ps[i].row0, ps[i].rows,
               ps[i].col0, ps[i].cols, ps[i].dst_row);
    printf("]");
}
inline void print_int_list(const std::vector<int> &v)
{
    printf("[");
    for (size_t i = 0; i < v.size(); ++i) printf("%s%d", i ? "," : "", v[i]);
    printf("]");
}
inline void PrintStripPlan(const StripPlanC &p, const double *cy)
{
    printf("{\"assign\":");
    print_int_list(p.assign);
    printf(",\"pieces\":");
    print_pieces(p.correlation_pieces());
    printf(",\"align_rows\":[");
    for (int r = 0; r < p.world; ++r) { long a, b; p.align_out_rows(r, &a, &b); printf("%s[%ld,%ld]", r ? "," : "", a, b); }
    printf("],\"align_transfers\":[");
    std::vector<std::pair<long, long>> need;
    auto tr = p.align_transfers(cy, &need);
    for (size_t i = 0; i < tr.size(); ++i) printf("%s[%d,%d,%ld,%ld]", i ? "," : "", tr[i].src, tr[i].dst, tr[i].row0, tr[i].rows);
    printf("],\"link_gbs\":%ld,\"predicted_finish_us\":", link_gbs());
    { std::vector<int> v(p.predicted_finish_us.begin(), p.predicted_finish_us.end()); print_int_list(v); }
    printf("}\n");
}
inline void PrintCcdPlan(const CcdPlanC &p, double dy)
{
    printf("{\"assign\":");
    print_int_list(p.assign);
    printf(",\"pieces\":");
    print_pieces(p.correlation_pieces());
    printf(",\"remap_transfers\":[");
    std::vector<std::pair<long, long>> need;
    auto tr = p.remap_transfers(dy, &need);
    for (size_t i = 0; i < tr.size(); ++i) printf("%s[%d,%d,%ld,%ld]", i ? "," : "", tr[i].src, tr[i].dst, tr[i].row0, tr[i].rows);
    printf("],\"need\":[");
    for (size_t i = 0; i < need.size(); ++i) printf("%s[%ld,%ld]", i ? "," : "", need[i].first, need[i].second);
    printf("],\"link_gbs\":%ld,\"predicted_finish_us\":", link_gbs());
    { std::vector<int> v(p.predicted_finish_us.begin(), p.predicted_finish_us.end()); print_int_list(v); }
    printf("}\n");
}

// ---- stage times of a rank's step -------------------------------------------------------------------
// The placement model (predicted_finish_us) has never met a multi-GPU node: every N-GPU run logs, per GPU, where its step's
// time went -- host time between points at which the rank's stream is drained anyway (RRC done, a correlation call returned,
// the all-gather or halo exchange synchronised, the product block downloaded) -- next to the model's prediction.
struct StageClock {
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    std::vector<std::pair<std::string, double>> ms;
    double since_rrc = 0.0, correlation_finish = 0.0;
    bool rrc_done = false;
    void tick(const char *name)
    {
        const auto now = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        for (auto &e : ms) if (e.first == name) { e.second += d; goto booked; }
        ms.push_back({name, d});
    booked:
        if (rrc_done) since_rrc += d;
        if (!strcmp(name, "rrc")) rrc_done = true;
        if (!strncmp(name, "correlate", 9)) correlation_finish = since_rrc;
    }
    std::string line() const
    {
        std::string s;
        char b[96];
        for (auto &e : ms) { snprintf(b, sizeof b, "%s%s %.2f", s.empty() ? "" : " | ", e.first.c_str(), e.second); s += b; }
        return s;
    }
};
inline void LogStageClocks(const std::vector<StageClock> &clk, const std::vector<long> &predicted_us)
{
    OLOG("Stage times per GPU in ms (host clock at points where the GPU's stream is drained; link figure assumed by the placement: %ld GB/s):", link_gbs());
    for (size_t r = 0; r < clk.size(); ++r) RLOG("  GPU %zu: %s", r, clk[r].line().c_str());
    std::string p, m;
    char b[48];
    for (size_t r = 0; r < clk.size(); ++r) {
        snprintf(b, sizeof b, "%s%ld", r ? ", " : "", r < predicted_us.size() ? predicted_us[r] : -1L); p += b;
        snprintf(b, sizeof b, "%s%.0f", r ? ", " : "", clk[r].correlation_finish * 1e3); m += b;
    }
    RLOG("  correlation stage finished (us after the RRC): predicted [%s], measured [%s]", p.c_str(), m.c_str());
}

// ---- the node: one context, stream and communicator per GPU, one host thread per GPU -------------------------------
class Node {
public:
    explicit Node(int n) : N(n), bar(n), ctx(n, nullptr), comms(n), err(n)
    {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || have < n)
            throw std::runtime_error("--gpus " + std::to_string(n) + ": only " + std::to_string(have) + " GPUs visible");
        std::vector<int> devs(n);
        for (int i = 0; i < n; ++i) {
            devs[i] = i;
            if (oip_create(i, &ctx[i]) != OIP_OK) throw std::runtime_error("no usable MI355X (gfx950) device " + std::to_string(i));
        }
        if (ncclCommInitAll(comms.comm.data(), n, devs.data()) != ncclSuccess) throw std::runtime_error("ncclCommInitAll failed");
        // a communication stream per GPU: the window exchange runs on it, beside the compute stream's kernels
        cstream.assign(n, nullptr);
        for (int i = 0; i < n; ++i) {
            if (hipSetDevice(i) != hipSuccess || hipStreamCreateWithFlags(&cstream[i], hipStreamNonBlocking) != hipSuccess)
                throw std::runtime_error("communication stream of GPU " + std::to_string(i));
        }
    }
    ~Node()
    {
        for (size_t i = 0; i < cstream.size(); ++i) if (cstream[i]) { hipSetDevice((int)i); hipStreamDestroy(cstream[i]); }
        if (!comms.aborted()) for (auto c : comms.comm) if (c) ncclCommDestroy(c);       // ncclCommAbort has already freed them otherwise
        for (auto c : ctx) if (c) oip_destroy(c);
    }
    hipStream_t stream(int r) { return (hipStream_t)oip_get_stream(ctx[r]); }
    void check(int r, int rc)
    {
        if (rc == OIP_OK) return;
        const std::string m = oip_last_error(ctx[r]);
        if (rc == OIP_E_INVALID) throw std::invalid_argument(m);
        throw std::runtime_error(m);
    }
    // run fn(rank) on one thread per GPU; the first exception is re-thrown on the caller
    template <typename F> void run(F fn)
    {
        std::vector<std::thread> th;
        std::vector<std::exception_ptr> ex(N);
        for (int r = 0; r < N; ++r)
            th.emplace_back([&, r] {
                try { hipSetDevice(r); fn(r); }
                catch (...) {
                    // A rank that gives up after a pre-exchange barrier leaves its peers inside a grouped send/recv, blocked in
                    // oip_sync on a kernel that waits for this rank for ever: setting a flag does not unblock a device-side wait.
                    // Aborting every communicator makes those kernels exit, the peers' streams drain, their oip_sync returns
                    // and they leave through sync_point(); the process then exits non-zero with this rank's message.
                    ex[r] = std::current_exception(); failed = true; bar.abort(); abort_comms();
                }
            });
        for (auto &t : th) t.join();
        // the rank that failed first tells why; the others only report that a peer gave up
        for (auto &e : ex)
            if (e) {
                try { std::rethrow_exception(e); }
                catch (const PeerFailed &) { continue; }
            }
        for (auto &e : ex) if (e) std::rethrow_exception(e);
    }
    // every rank thread calls this at the same point; a failure on any rank is raised on all (no rank is left waiting
    // in a collective for a peer that has given up)
    void sync_point()
    {
        if (!bar.wait() || failed) throw PeerFailed();
    }
    // waits for the RCCL calls in flight on the peers' threads (CommGuard), then aborts every communicator once; from then
    // on with_comm() throws PeerFailed instead of touching a freed communicator
    void abort_comms()
    {
        comms.abort_all([](ncclComm_t c) { if (c) ncclCommAbort(c); });
    }
    // every use of a communicator -- one call or a whole ncclGroupStart .. ncclGroupEnd sequence -- goes through here
    template <typename F> void with_comm(int r, F f)
    {
        if (failed) throw PeerFailed();
        comms.use(r, f);
    }
    // an RCCL call that fails is this rank's failure: raised at once (run() then aborts the communicators), never carried
    // into a stream synchronisation that a peer may not be able to complete
    static void nccl_ok(ncclResult_t rc, const char *what)
    {
        if (rc != ncclSuccess) throw std::runtime_error(std::string(what) + " failed: " + ncclGetErrorString(rc));
    }

    // exchange step 1, overlapped with the correlation: every group of units that is not entirely on its rank is packed and
    // posted as ONE grouped send/recv on the rank's communication stream, in the global order every rank iterates; an event
    // per group lets the compute stream wait for that group's bytes only.  The caller computes its resident groups first,
    // then each received group behind its event, and finally calls finish_pieces (drains the stream, frees the packing
    // buffers, meets the other ranks).
    struct PendingGroup {
        std::vector<int> units;
        hipEvent_t ev;
    };
    struct PendingExchange {
        std::vector<PendingGroup> groups;
        std::vector<void *> temps;
    };
    template <typename SrcOf, typename DstOf>
    PendingExchange post_pieces(int r, const std::vector<std::pair<std::vector<int>, std::vector<Piece>>> &groups, int planes_of_kind1,
                                SrcOf src_of, DstOf dst_of)
    {
        PendingExchange pe;
        hipStream_t st = stream(r), cs = cstream[r];
        // the communication stream starts behind what the compute stream has produced so far (the corrected lines)
        hipEvent_t ready;
        if (hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(ready, st) != hipSuccess ||
            hipStreamWaitEvent(cs, ready, 0) != hipSuccess) throw std::runtime_error("communication stream ordering failed");
        hipEventDestroy(ready);
        sync_point();
        for (const auto &g : groups) {
            struct Xfer { void *buf; size_t bytes; int peer; };
            std::vector<Xfer> sends, recvs;
            for (const Piece &p : g.second) {
                const int planes = p.kind == 1 ? planes_of_kind1 : 1;
                for (int b = 0; b < planes; ++b) {
                    const size_t bytes = (size_t)p.rows * p.cols * 2;
                    if (p.src == r && p.dst == r) {
                        auto s = src_of(p, b);
                        if (hipMemcpy2DAsync(dst_of(p, b), (size_t)p.cols * 2, s.first, s.second * 2, (size_t)p.cols * 2, p.rows,
                                             hipMemcpyDeviceToDevice, cs) != hipSuccess) throw std::runtime_error("window copy failed");
                    } else if (p.src == r) {
                        void *t = nullptr;
                        check(r, oip_malloc(ctx[r], &t, bytes));
                        pe.temps.push_back(t);
                        auto s = src_of(p, b);
                        if (hipMemcpy2DAsync(t, (size_t)p.cols * 2, s.first, s.second * 2, (size_t)p.cols * 2, p.rows,
                                             hipMemcpyDeviceToDevice, cs) != hipSuccess) throw std::runtime_error("window pack failed");
                        sends.push_back({t, bytes, p.dst});
                    } else if (p.dst == r) {
                        recvs.push_back({dst_of(p, b), bytes, p.src});
                    }
                }
            }
            if (!sends.empty() || !recvs.empty())
                with_comm(r, [&](ncclComm_t c) {
                    fault_point(r, "exchange");
                    nccl_ok(ncclGroupStart(), "ncclGroupStart");
                    for (auto &x : sends) nccl_ok(ncclSend(x.buf, x.bytes, ncclUint8, x.peer, c, cs), "ncclSend (window piece)");
                    for (auto &x : recvs) nccl_ok(ncclRecv(x.buf, x.bytes, ncclUint8, x.peer, c, cs), "ncclRecv (window piece)");
                    nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
                });
            PendingGroup pg;
            pg.units = g.first;
            if (hipEventCreateWithFlags(&pg.ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(pg.ev, cs) != hipSuccess)
                throw std::runtime_error("exchange event failed");
            pe.groups.push_back(pg);
        }
        return pe;
    }
    // the compute stream waits (on the device) for one group's bytes
    void wait_group(int r, const PendingGroup &g)
    {
        // the host waits too (the correlation call that follows could not start earlier anyway): the wait for a pair's bytes is
        // then booked apart from the pair's correlation (StageClock)
        if (hipEventSynchronize(g.ev) != hipSuccess || hipStreamWaitEvent(stream(r), g.ev, 0) != hipSuccess)
            throw std::runtime_error("exchange wait failed");
    }
    void finish_pieces(int r, PendingExchange *pe)
    {
        if (hipStreamSynchronize(cstream[r]) != hipSuccess) throw std::runtime_error("window exchange failed");
        for (auto &g : pe->groups) hipEventDestroy(g.ev);
        for (void *t : pe->temps) oip_free(ctx[r], t);
        pe->groups.clear();
        pe->temps.clear();
        sync_point();
    }

    // exchange step 3: whole lines into the halo rows of a raster.  ptr_of(global line, plane) -> device pointer on
    // this rank; bytes_per_line per plane
    template <typename PtrOf>
    void exchange_lines(int r, const std::vector<LineTransfer> &tr, int planes, size_t bytes_per_line, PtrOf ptr_of)
    {
        hipStream_t st = stream(r);
        sync_point();
        bool any = false;
        for (const LineTransfer &t : tr) any = any || t.src == r || t.dst == r;
        if (any)
            with_comm(r, [&](ncclComm_t c) {
                fault_point(r, "halo");
                nccl_ok(ncclGroupStart(), "ncclGroupStart");
                for (const LineTransfer &t : tr)
                    for (int b = 0; b < planes; ++b) {
                        if (t.src == r) nccl_ok(ncclSend(ptr_of(t.row0, b), (size_t)t.rows * bytes_per_line, ncclUint8, t.dst, c, st), "ncclSend (halo lines)");
                        if (t.dst == r) nccl_ok(ncclRecv(ptr_of(t.row0, b), (size_t)t.rows * bytes_per_line, ncclUint8, t.src, c, st), "ncclRecv (halo lines)");
                    }
                nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
            });
        check(r, oip_sync(ctx[r]));
        sync_point();
    }

    // exchange step 2: every rank contributes a table with NaN for what it did not compute; all ranks end with the merged
    // table (rank order, first non-NaN wins -- each entry is computed by exactly one rank)
    void allgather_table(int r, std::vector<double> *table, int stride)
    {
        const size_t n = table->size();
        double *d_send = nullptr, *d_recv = nullptr;
        check(r, oip_malloc(ctx[r], (void **)&d_send, n * sizeof(double)));
        check(r, oip_malloc(ctx[r], (void **)&d_recv, n * sizeof(double) * N));
        check(r, oip_memcpy_h2d(ctx[r], d_send, table->data(), n * sizeof(double)));
        check(r, oip_sync(ctx[r]));
        sync_point();
        with_comm(r, [&](ncclComm_t c) {
            fault_point(r, "allgather");
            nccl_ok(ncclAllGather(d_send, d_recv, n, ncclDouble, c, stream(r)), "ncclAllGather");
        });
        std::vector<double> all(n * N);
        check(r, oip_memcpy_d2h(ctx[r], all.data(), d_recv, n * N * sizeof(double)));
        check(r, oip_sync(ctx[r]));
        for (size_t i = 0; i < n; i += stride)
            for (int q = 0; q < N; ++q)
                if (all[q * n + i] == all[q * n + i]) { for (int k = 0; k < stride; ++k) (*table)[i + k] = all[q * n + i + k]; break; }
        oip_free(ctx[r], d_send);
        oip_free(ctx[r], d_recv);
        sync_point();
    }

    const int N;
    HostBarrier bar;
    // OIP_FAULT_INJECT="<rank>:<point>" (point: exchange, halo, allgather) makes that rank throw at that point -- behind the
    // barrier that precedes the exchange, its peers already posting: what the failure path has to survive (test_gpu_cli.py)
    static void fault_point(int r, const char *point)
    {
        const char *e = getenv("OIP_FAULT_INJECT");
        if (!e) return;
        char want[32] = "";
        int rank = -1;
        if (sscanf(e, "%d:%31s", &rank, want) == 2 && rank == r && !strcmp(want, point))
            throw std::runtime_error(std::string("injected failure on GPU ") + std::to_string(r) + " at " + point);
    }

    std::vector<oip_ctx *> ctx;
    CommGuard<ncclComm_t> comms;
    std::vector<hipStream_t> cstream;
    std::vector<std::string> err;
    std::atomic<bool> failed{false};
};

struct MultiGpuDefaultOptions {
    int width = OIP_PIXELS_PER_LINE, gpus = 1;
    bool doRRC4PAN = false, doRRC4MSS = true, keepLeading = false;
    int slices = OIP_IBCV_DEF_SLICES, sections = OIP_IBCV_DEF_SECTIONS, linesSection = OIP_IBPA_DEFAULT_BATCHLINES, lineOffset = 0,
        overlapLines = OIP_IBPA_DEFAULT_LINEOVERLAP, fitMode = OIP_FIT_REFERENCE;
    double threshold = OIP_IBCV_DEF_THRESHOLD;
};

// main.cpp:288-317 on N GPUs: <stem>.ALIGNED.TIFF is the single-GPU product
inline void RunDefaultActionMultiGpu(const std::string &panFile, const std::string &mssFile, const std::string &rrcPan,
                                     const std::string rrcMss[MSS_BANDS], const MultiGpuDefaultOptions &o)
{
    const int W = o.width, Wb = W / MSS_BANDS, N = o.gpus;
    const size_t lineBytes = (size_t)W * BYTES_PER_PIXEL;
    const size_t sizePAN = IMO::FileSize(panFile), sizeMSS = IMO::FileSize(mssFile);
    if (sizePAN != MSS_BANDS * sizeMSS)                                              // preproc.h:565-567
        throw std::runtime_error("PAN file size does not match MSS file size: PAN file should be " + std::to_string(MSS_BANDS) + "x as large as MSS file");
    if (sizePAN % lineBytes != 0) throw std::runtime_error("PAN file size invalid: should be multiplies of " + std::to_string(lineBytes));
    const long Lp = (long)(sizePAN / lineBytes);
    StripPlanC plan(W, Lp, N, o.slices, o.sections, OIP_CORRELATION_LINES, o.linesSection, o.lineOffset, o.overlapLines, o.keepLeading);
    OLOG("Strip of %ld PAN lines on %d GPUs: %ld lines per GPU, %d correlation units dealt %s.", Lp, N, plan.pb, plan.n_units,
         plan.correlation_pieces().empty() ? "without moving a line" : "with window exchange");
    std::vector<double> kbPan, kbMss;
    if (o.doRRC4PAN) {
        std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(rrcPan.c_str(), W));
        kbPan.assign((double *)prm.get(), (double *)prm.get() + (size_t)W * 2);
    }
    if (o.doRRC4MSS) {
        kbMss.resize((size_t)W * 2);
        for (int b = 0; b < MSS_BANDS; ++b) {
            std::unique_ptr<RRCParam[]> prm(IMO::LoadRRCParamFile(rrcMss[b].c_str(), Wb));
            memcpy(&kbMss[(size_t)b * Wb * 2], prm.get(), sizeof(double) * 2 * Wb);
        }
    }
    Node node(N);
    std::vector<std::vector<uint16_t>> blocks(N);             // aligned rows of each rank, on the host
    std::vector<std::pair<long, long>> out_rows(N);
    double cxAll[MSS_BANDS][2] = {}, cyAll[MSS_BANDS][3] = {};
    std::vector<StageClock> clocks(N);
    node.run([&](int r) {
        StageClock &clk = clocks[r];
        oip_ctx *c = node.ctx[r];
        auto ck = [&](int rc) { node.check(r, rc); };
        // the rank's own lines, read from the files at the block's offset
        uint16_t *pan = nullptr, *bil = nullptr, *planes = nullptr;
        const long b0 = r * plan.mb, b1 = (r + 1) * plan.mb;
        const long head = std::min<long>(plan.halo_cap, b0), tail = std::min<long>(plan.halo_cap, plan.Lm - b1);
        const long m_first = b0 - head, m_cap = head + plan.mb + tail;
        const size_t plane_stride = (size_t)m_cap * Wb;
        ck(oip_malloc(c, (void **)&pan, (size_t)plan.pb * lineBytes));
        ck(oip_malloc(c, (void **)&bil, (size_t)plan.mb * lineBytes));
        ck(oip_malloc(c, (void **)&planes, plane_stride * MSS_BANDS * 2));
        ck(oip_memset(c, planes, 0, plane_stride * MSS_BANDS * 2));
        // a short read would leave uninitialised HBM under the correlation: the block must arrive whole (as DevBuf::load_file checks)
        auto read_block = [&](const std::string &file, size_t offset, size_t bytes, void *dst) {
            size_t got = 0;
            ck(oip_read_file_to_device(c, file.c_str(), offset, bytes, dst, &got, nullptr));
            if (got != bytes)
                throw std::runtime_error("read file [" + file + "] failed: " + std::to_string(got) + " of " + std::to_string(bytes) +
                                         " bytes at offset " + std::to_string(offset));
        };
        read_block(panFile, (size_t)r * plan.pb * lineBytes, (size_t)plan.pb * lineBytes, pan);
        read_block(mssFile, (size_t)r * plan.mb * lineBytes, (size_t)plan.mb * lineBytes, bil);
        ck(oip_sync(c));
        clk.tick("read");
        double *d_kb = nullptr;
        ck(oip_malloc(c, (void **)&d_kb, (size_t)W * 16));
        if (o.doRRC4PAN) {
            ck(oip_memcpy_h2d(c, d_kb, kbPan.data(), (size_t)W * 16));
            ck(oip_rrc_u16(c, pan, pan, W, plan.pb, d_kb));
            ck(oip_sync(c));
        }
        if (o.doRRC4MSS) ck(oip_memcpy_h2d(c, d_kb, kbMss.data(), (size_t)W * 16));
        uint16_t *own = planes + (size_t)head * Wb;
        ck(oip_mss_split_rrc_u16(c, bil, own, plane_stride, W, plan.mb, o.doRRC4MSS ? d_kb : nullptr));
        ck(oip_sync(c));
        clk.tick("rrc");
        ck(oip_free(c, bil));
        // exchange 1: compact windows of the units this rank computes but does not hold entirely
        const std::vector<int> mine = plan.units_of(r);
        std::vector<uint16_t *> wpan(plan.n_units, nullptr), wmss(plan.n_units, nullptr);
        for (int u : mine)
            if (!plan.unit_is_local(u)) {
                ck(oip_malloc(c, (void **)&wpan[u], (size_t)plan.base_rows * plan.base_cols * 2));
                ck(oip_malloc(c, (void **)&wmss[u], (size_t)plan.band_rows * plan.band_cols * 2 * MSS_BANDS));
            }
        auto pending = node.post_pieces(
            r, plan.exchange_groups(), MSS_BANDS,
            [&](const Piece &p, int b) -> std::pair<const void *, size_t> {
                if (p.kind == 0) return {pan + (size_t)(p.row0 - r * plan.pb) * W + p.col0, (size_t)W};
                return {planes + (size_t)b * plane_stride + (size_t)(p.row0 - m_first) * Wb + p.col0, (size_t)Wb};
            },
            [&](const Piece &p, int b) -> void * {
                if (p.kind == 0) return wpan[p.unit] + (size_t)p.dst_row * plan.base_cols;
                return wmss[p.unit] + (size_t)b * plan.band_rows * plan.band_cols + (size_t)p.dst_row * plan.band_cols;
            });
        clk.tick("exchange_post");
        // The units, pairs as in the single-GPU order (a pair is resident or not as a whole): the resident pairs are computed
        // while the exchange runs on the communication stream, then every received pair behind its own event.
        std::vector<double> table((size_t)plan.n_units * 12, std::numeric_limits<double>::quiet_NaN());
        auto correlate = [&](const std::vector<int> &units) {
            if (units.empty()) return;
            std::vector<const uint16_t *> up, ub;
            std::vector<size_t> pp, pbm;
            for (int u : units) {
                long p0, m0;
                plan.section(u / plan.slices, &p0, &m0);
                const int i = u % plan.slices;
                if (wpan[u]) {
                    up.push_back(wpan[u]); pp.push_back(plan.base_cols);
                    for (int b = 0; b < MSS_BANDS; ++b) ub.push_back(wmss[u] + (size_t)b * plan.band_rows * plan.band_cols);
                    pbm.push_back(plan.band_cols);
                } else {
                    up.push_back(pan + (size_t)(p0 - r * plan.pb) * W + (size_t)i * plan.base_cols); pp.push_back(W);
                    for (int b = 0; b < MSS_BANDS; ++b) ub.push_back(planes + (size_t)b * plane_stride + (size_t)(m0 - m_first) * Wb + (size_t)i * plan.band_cols);
                    pbm.push_back(Wb);
                }
            }
            std::vector<double> res(12 * units.size() + 12);
            ck(oip_interband_correlate_units(c, up.data(), pp.data(), ub.data(), pbm.data(), (int)units.size(), plan.base_rows, plan.base_cols, res.data()));
            for (size_t j = 0; j < units.size(); ++j) memcpy(&table[(size_t)units[j] * 12], &res[12 * j], sizeof(double) * 12);
        };
        {
            // A pair (2k, 2k+1) is resident only when BOTH its units are: with an odd slice count a pair spans two sections and one
            // unit can be local while its partner is not -- computing the local one alone would shift every later pairing on this
            // rank, and a unit's last digits depend on its partner.  Such a pair waits for its exchange group and is
            // computed whole.
            auto pair_of = [&](int u) { std::vector<int> v; for (int w = u & ~1; w < std::min((u & ~1) + 2, plan.n_units); ++w) v.push_back(w); return v; };
            auto pair_local = [&](int u) { for (int w : pair_of(u)) if (!plan.unit_is_local(w)) return false; return true; };
            std::vector<int> resident;
            for (int u : mine) if (pair_local(u)) resident.push_back(u);
            correlate(resident);
            clk.tick("correlate_resident");
            for (const auto &g : pending.groups) {
                std::vector<int> here;
                for (int u : pair_of(g.units[0])) if (plan.assign[u] == r) here.push_back(u);
                if (here.empty()) continue;
                node.wait_group(r, g);
                clk.tick("exchange_wait");
                correlate(here);
                clk.tick("correlate_received");
            }
            node.finish_pieces(r, &pending);
            clk.tick("exchange_drain");
        }
        // exchange 2: the table [unit][band][dx, dy, rs]
        node.allgather_table(r, &table, 12);
        clk.tick("allgather");
        std::vector<double> shifts((size_t)MSS_BANDS * plan.n_units * 4);
        for (int b = 0; b < MSS_BANDS; ++b)
            for (int u = 0; u < plan.n_units; ++u) {
                double *s = &shifts[((size_t)b * plan.n_units + u) * 4];
                for (int k = 0; k < 3; ++k) s[k] = table[(size_t)u * 12 + 3 * b + k];
                s[3] = (double)((u % plan.slices) * plan.base_cols + plan.base_cols / 2);        // preproc.h:326
            }
        double cx[MSS_BANDS][2], cy[MSS_BANDS][3];
        char err[512];
        if (oip_filter_and_fit_mode(shifts.data(), plan.n_units, o.threshold, OIP_IBCV_MIN_COUNT, o.fitMode, &cx[0][0], &cy[0][0], err, sizeof err) != OIP_OK)
            throw std::runtime_error(err);
        if (r == 0) { memcpy(cxAll, cx, sizeof cx); memcpy(cyAll, cy, sizeof cy); }
        // exchange 3: align halo
        std::vector<std::pair<long, long>> need;
        auto tr = plan.align_transfers(&cy[0][0], &need);
        for (const LineTransfer &t : tr)
            if (t.dst == r && (t.row0 < m_first || t.row0 + t.rows > m_first + m_cap)) throw std::runtime_error("MSS halo exceeds buffer capacity");
        node.exchange_lines(r, tr, MSS_BANDS, (size_t)Wb * 2, [&](long line, int b) -> void * {
            return planes + (size_t)b * plane_stride + (size_t)(line - m_first) * Wb;
        });
        clk.tick("fit+halo");
        long v0 = b0, v1 = b1;
        for (const LineTransfer &t : tr) if (t.dst == r) { v0 = std::min(v0, t.row0); v1 = std::max(v1, t.row0 + t.rows); }
        long o0, o1;
        plan.align_out_rows(r, &o0, &o1);
        out_rows[r] = {o0, o1};
        if (o1 > o0) {
            uint16_t *out = nullptr;
            const size_t n = (size_t)(o1 - o0) * Wb * MSS_BANDS;
            ck(oip_malloc(c, (void **)&out, n * 2));
            ck(oip_align_mss_bicubic_u16x4(c, planes + (size_t)(v0 - m_first) * Wb, plane_stride, v0, v1 - v0, out, o0, o1 - o0, Wb, plan.Lm,
                                           &cx[0][0], &cy[0][0], o.linesSection, o.lineOffset, o.overlapLines, o.keepLeading ? 1 : 0,
                                           OIP_IBPA_MIN_PROCESSLINES, nullptr));
            blocks[r].resize(n);
            ck(oip_download_staged(c, blocks[r].data(), out, n * 2));
            ck(oip_free(c, out));
        }
        clk.tick("align+download");
        for (int u : mine) { if (wpan[u]) oip_free(c, wpan[u]); if (wmss[u]) oip_free(c, wmss[u]); }
        oip_free(c, d_kb); oip_free(c, planes); oip_free(c, pan);
    });
    LogStageClocks(clocks, plan.predicted_finish_us);
    for (int b = 0; b < MSS_BANDS; ++b) {
        OLOG("BAND %d\tdeltaX coeff: [1] %.15f, [0] %.9f", b, cxAll[b][1], cxAll[b][0]);
        OLOG("\tdeltaY coeff: [2] %.15f, [1] %.15f, [0] %.9f", cyAll[b][2], cyAll[b][1], cyAll[b][0]);
    }
    // preproc.h:167-185: one ALIGNED.TIFF, the blocks in rank order
    auto save = IMO::BuildOutputFilePath(mssFile, ".ALIGNED", ".TIFF");
    OLOG("Outputing aligned TIFF image (%d x %ld x 4) to [%s] ...", Wb, plan.out_rows, save.c_str());
    TiffWriterU16 tw(save, Wb, plan.out_rows, MSS_BANDS, true, tiff_compression(TIFF_LZW));
    for (int r = 0; r < N; ++r)
        if (out_rows[r].second > out_rows[r].first) tw.write_rows(blocks[r].data(), out_rows[r].second - out_rows[r].first);
    tw.close();
    OLOG("Output done.");
}

struct MultiGpuPrestitchOptions {
    int width = OIP_PIXELS_PER_LINE, gpus = 1, sections = OIP_STT_DEF_SECTIONS, sectionLines = OIP_STT_DEF_SECLINES,
        overlapCols = OIP_STT_DEF_OVERLAPPX, edgeCols = 0;
    double threshold = OIP_STT_DEF_PHCTHRHLD, maxDeltaY = 0.0;
    bool doRRC = true, onlyCalc = false, fp16acc = false;
};

// main.cpp:270-286 on N GPUs: .RRC.RAW x 2 and .RRC.PRESTT.RAW are the single-GPU products
inline void RunPrestitchMultiGpu(const std::string &pan1, const std::string &pan2, const std::string &rrc1, const std::string &rrc2,
                                 const MultiGpuPrestitchOptions &o)
{
    const int W = o.width, N = o.gpus;
    const size_t lineBytes = (size_t)W * BYTES_PER_PIXEL;
    const size_t s1 = IMO::FileSize(pan1), s2 = IMO::FileSize(pan2);
    if (s1 != s2) throw std::invalid_argument("PAN1 size doesn't match PAN2 size");
    const long L = (long)(s1 / lineBytes);
    CcdPlanC plan(W, L, N, o.sections, o.sectionLines, o.overlapCols, o.edgeCols);
    if (!o.onlyCalc && L <= OIP_REMAP_ROW_GUARD) throw std::invalid_argument("too few data rows, please use cv::remap()");   // imageop.h:242-244
    std::vector<double> kb1, kb2;
    if (o.doRRC && !o.onlyCalc) {
        std::unique_ptr<RRCParam[]> a(IMO::LoadRRCParamFile(rrc1.c_str(), W)), b(IMO::LoadRRCParamFile(rrc2.c_str(), W));
        kb1.assign((double *)a.get(), (double *)a.get() + (size_t)W * 2);
        kb2.assign((double *)b.get(), (double *)b.get() + (size_t)W * 2);
    }
    const std::string f1 = o.doRRC ? IMO::BuildOutputFilePath(pan1, ".RRC") : pan1, f2 = o.doRRC ? IMO::BuildOutputFilePath(pan2, ".RRC") : pan2;
    const std::string fp = IMO::BuildOutputFilePath(f2, ".PRESTT");
    Node node(N);
    double dxAll = 0, dyAll = 0;
    std::vector<StageClock> clocks(N);
    node.run([&](int r) {
        StageClock &clk = clocks[r];
        oip_ctx *c = node.ctx[r];
        auto ck = [&](int rc) { node.check(r, rc); };
        const long b0 = r * plan.pb;
        uint16_t *p1 = nullptr, *p2 = nullptr;
        const size_t blockBytes = (size_t)plan.pb * lineBytes;
        ck(oip_malloc(c, (void **)&p1, blockBytes));
        ck(oip_malloc(c, (void **)&p2, blockBytes));
        auto read_block = [&](const std::string &file, size_t offset, size_t bytes, void *dst) {
            size_t got = 0;
            ck(oip_read_file_to_device(c, file.c_str(), offset, bytes, dst, &got, nullptr));
            if (got != bytes)
                throw std::runtime_error("read file [" + file + "] failed: " + std::to_string(got) + " of " + std::to_string(bytes) +
                                         " bytes at offset " + std::to_string(offset));
        };
        read_block(pan1, (size_t)b0 * lineBytes, blockBytes, p1);
        read_block(pan2, (size_t)b0 * lineBytes, blockBytes, p2);
        ck(oip_sync(c));
        // exchange 1 + correlation on the RAW lines (App. B-1)
        const std::vector<int> mine = plan.units_of(r);
        std::vector<uint16_t *> wa(plan.n_units, nullptr), wb(plan.n_units, nullptr);
        for (int u : mine)
            if (!plan.unit_is_local(u)) {
                ck(oip_malloc(c, (void **)&wa[u], (size_t)plan.lps * plan.cols * 2));
                ck(oip_malloc(c, (void **)&wb[u], (size_t)plan.lps * plan.cols * 2));
            }
        auto pending = node.post_pieces(
            r, plan.exchange_groups(), 1,
            [&](const Piece &p, int) -> std::pair<const void *, size_t> {
                return {(p.kind == 0 ? p1 : p2) + (size_t)(p.row0 - b0) * W + p.col0, (size_t)W};
            },
            [&](const Piece &p, int) -> void * { return (p.kind == 0 ? wa : wb)[p.unit] + (size_t)p.dst_row * plan.cols; });
        ck(oip_sync(c));
        clk.tick("read");
        clk.tick("rrc");                                  // CalcSttParameters works on the RAW lines: the model's time 0 is here
        std::vector<double> table((size_t)plan.sections * 3, std::numeric_limits<double>::quiet_NaN());
        auto correlate = [&](const std::vector<int> &units) {
            if (units.empty()) return;
            std::vector<const uint16_t *> pa, pbv;
            std::vector<size_t> qa, qb;
            for (int u : units) {
                if (wa[u]) { pa.push_back(wa[u]); pbv.push_back(wb[u]); qa.push_back(plan.cols); qb.push_back(plan.cols); }
                else {
                    const long a = plan.section_start(u);
                    pa.push_back(p1 + (size_t)(a - b0) * W + (W - plan.ov)); pbv.push_back(p2 + (size_t)(a - b0) * W + plan.edge);
                    qa.push_back(W); qb.push_back(W);
                }
            }
            std::vector<double> res(3 * units.size() + 3);
            ck(oip_stt_correlate_windows(c, pa.data(), qa.data(), pbv.data(), qb.data(), (int)units.size(), plan.lps, plan.cols, res.data()));
            for (size_t j = 0; j < units.size(); ++j) memcpy(&table[(size_t)units[j] * 3], &res[3 * j], sizeof(double) * 3);
        };
        {
            std::vector<int> resident;
            for (int u : mine) if (plan.unit_is_local(u)) resident.push_back(u);
            correlate(resident);
            clk.tick("correlate_resident");
            for (const auto &g : pending.groups) {
                std::vector<int> here;
                for (int u : g.units) if (plan.assign[u] == r) here.push_back(u);
                if (here.empty()) continue;
                node.wait_group(r, g);
                clk.tick("exchange_wait");
                correlate(here);
                clk.tick("correlate_received");
            }
            node.finish_pieces(r, &pending);
            clk.tick("exchange_drain");
        }
        node.allgather_table(r, &table, 3);
        clk.tick("allgather");
        double dx, dy, resp;
        int valid = 0;
        if (oip_stt_mean(table.data(), plan.sections, o.threshold, o.maxDeltaY, &dx, &dy, &resp, &valid) != OIP_OK)      // stitcher.h:181-198
            throw std::runtime_error("No valid delta value found for stitching parameter calculating");
        if (r == 0) {
            dxAll = dx; dyAll = dy;
            RLOG("| offset |  delta x |  delta y | response | r |");
            for (int i = 0; i < plan.sections; ++i)
                RLOG("|%7ld |%10.4f|%10.4f|%10.4f|%s|", plan.section_start(i), table[3 * i], table[3 * i + 1], table[3 * i + 2],
                     (table[3 * i + 2] >= o.threshold && (o.maxDeltaY <= 0.0 || std::abs(table[3 * i + 1]) <= o.maxDeltaY)) ? " Y " : " N ");
            OLOG("Total %d valid delta value pairs found, everage value:", valid);
            OLOG("    dx: %.5f, dy: %.5f, r: %.5f", dx, dy, resp);
        }
        for (int u : mine) { if (wa[u]) oip_free(c, wa[u]); if (wb[u]) oip_free(c, wb[u]); }
        if (o.onlyCalc) { oip_free(c, p1); oip_free(c, p2); return; }
        // DoRRC of the rank's lines; CCD 2 lands in a buffer with room for the remap halo
        std::vector<std::pair<long, long>> need;
        auto tr = plan.remap_transfers(dy, &need);
        const long f = std::min(need[r].first, b0), l = std::max(need[r].second, b0 + plan.pb);
        uint16_t *r2 = nullptr;
        ck(oip_malloc(c, (void **)&r2, (size_t)(l - f) * lineBytes));
        ck(oip_memset(c, r2, 0, (size_t)(l - f) * lineBytes));
        uint16_t *own2 = r2 + (size_t)(b0 - f) * W;
        if (o.doRRC) {
            double *d_kb = nullptr;
            ck(oip_malloc(c, (void **)&d_kb, (size_t)W * 16));
            ck(oip_memcpy_h2d(c, d_kb, kb1.data(), (size_t)W * 16));
            ck(oip_rrc_u16(c, p1, p1, W, plan.pb, d_kb));
            ck(oip_sync(c));
            ck(oip_memcpy_h2d(c, d_kb, kb2.data(), (size_t)W * 16));
            ck(oip_rrc_u16(c, p2, own2, W, plan.pb, d_kb));
            ck(oip_sync(c));
            oip_free(c, d_kb);
        } else {
            if (hipMemcpyAsync(own2, p2, blockBytes, hipMemcpyDeviceToDevice, node.stream(r)) != hipSuccess) throw std::runtime_error("copy failed");
            ck(oip_sync(c));
        }
        oip_free(c, p2);
        clk.tick("rrc_x2");
        node.exchange_lines(r, tr, 1, lineBytes, [&](long line, int) -> void * { return r2 + (size_t)(line - f) * W; });
        clk.tick("halo");
        uint16_t *dst = nullptr;
        ck(oip_malloc(c, (void **)&dst, blockBytes));
        auto fn = o.fp16acc ? oip_remap_shift_bicubic_u16_f16acc : oip_remap_shift_bicubic_u16;
        ck(fn(c, r2, f, l - f, dst, b0, plan.pb, W, L, dx, dy, OIP_REMAP_SECTION_ROWS, OIP_REMAP_ROW_GUARD));
        ck(oip_sync(c));
        clk.tick("remap");
        // products: the ranks append their blocks to the files in rank order (every rank passes N turns)
        for (int t = 0; t < N; ++t) {
            if (t == r) {
                if (o.doRRC) {
                    ck(oip_write_device_to_file(c, p1, blockBytes, f1.c_str(), r > 0));
                    ck(oip_write_device_to_file(c, own2, blockBytes, f2.c_str(), r > 0));
                }
                ck(oip_write_device_to_file(c, dst, blockBytes, fp.c_str(), r > 0));
            }
            node.sync_point();
        }
        clk.tick("write");
        oip_free(c, dst); oip_free(c, r2); oip_free(c, p1);
    });
    LogStageClocks(clocks, plan.predicted_finish_us);
    if (!o.onlyCalc) OLOG("Pre-stitched PAN2 written to file '%s' (dx %.5f, dy %.5f).", fp.c_str(), dxAll, dyAll);
}

}  // namespace OIPGPU
