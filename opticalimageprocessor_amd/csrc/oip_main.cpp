// oip_main.cpp -- the reference's command line (main.cpp:92-343) rebuilt on the C ABI.
//
//   oip prestitch --pan1 A.RAW --pan2 B.RAW [--rrc1 --rrc2 -s -l --stitch-overlap --stt-threshold
//                 --stt-maxdeltay -e -r/--rrc/--no-rrc -c]                     (main.cpp:112-150)
//   oip stitch --image1 L.RAW --image2 R.RAW -c/--fold-cols N [-o OUT.RAW]       (main.cpp:159-190)
//   oip --pan P.RAW --mss M.RAW [--do-rrc4pan --rrc-pan F --no-rrc4mss --rrc-msb1..4 F --slices
//       --ibc-sections --ibc-threshold --line-offset --lines-section --overlap-lines -k]   (:193-252)
//   oip task ...                fused flow of DOC/sample-task.sh, see run_task()
//   oip -v | --version          prints 1.1
// plus --width N (pixels per PAN line; the reference hard-codes 12288, oipshared.h:28).
// `auxsep` is outside this build.  TIFF input and output go through oip_tiff.hpp (uncompressed and LZW, with
// or without the horizontal predictor).  Two options are not the reference's: --fit and --fp16-accumulate.
//
// Exit codes as the reference: usage_error -> "USAGE ERROR" + 254; any std::exception -> 2; unknown
// -> 1; help/version -> 255 (CLI11's Success + 255, main.cpp:262-263); argument errors -> CLI11's
// codes (RequiredError 106, ValidationError 105, ExtrasError 109, ConversionError 104).
#include <cstdlib>
#include <unistd.h>
#include <map>
#include <set>

#include "oip_host.hpp"
#include "oip_multigpu.hpp"

using namespace OIPGPU;

namespace {

struct cli_error : public std::runtime_error {
    int code;
    cli_error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// option table: name -> takes a value?
struct Spec {
    std::map<std::string, std::string> alias;       // "-s" -> "--sections"
    std::set<std::string> valued, flags;
};

struct Parsed {
    std::map<std::string, std::string> val;
    std::set<std::string> flag;
    bool has(const std::string &k) const { return val.count(k) || flag.count(k); }
    std::string str(const std::string &k, const std::string &def = "") const { auto it = val.find(k); return it == val.end() ? def : it->second; }
    int integer(const std::string &k, int def) const
    {
        auto it = val.find(k);
        if (it == val.end()) return def;
        char *e = nullptr;
        long v = strtol(it->second.c_str(), &e, 10);
        if (!e || *e) throw cli_error(104, "Could not convert: " + k + " = " + it->second);
        return (int)v;
    }
    double real(const std::string &k, double def) const
    {
        auto it = val.find(k);
        if (it == val.end()) return def;
        char *e = nullptr;
        double v = strtod(it->second.c_str(), &e);
        if (!e || *e) throw cli_error(104, "Could not convert: " + k + " = " + it->second);
        return v;
    }
};

Parsed parse(const Spec &sp, const std::vector<std::string> &args)
{
    Parsed p;
    for (size_t i = 0; i < args.size(); ++i) {
        std::string a = args[i], v;
        bool has_v = false;
        auto eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) { v = a.substr(eq + 1); a = a.substr(0, eq); has_v = true; }
        auto al = sp.alias.find(a);
        if (al != sp.alias.end()) a = al->second;
        if (sp.flags.count(a)) { p.flag.insert(a); continue; }
        if (sp.valued.count(a)) {
            if (!has_v) {
                if (i + 1 >= args.size()) throw cli_error(114, a + ": 1 required value missing");
                v = args[++i];
            }
            p.val[a] = v;
            continue;
        }
        throw cli_error(109, "The following argument was not expected: " + args[i]);
    }
    return p;
}

void require(const Parsed &p, const std::string &k)
{
    if (!p.has(k)) throw cli_error(106, k + " is required");
}

void existing_file(const Parsed &p, const std::string &k)
{
    if (!p.val.count(k)) return;
    struct stat st;
    if (stat(p.val.at(k).c_str(), &st) != 0 || !S_ISREG(st.st_mode))
        throw cli_error(105, k + ": File does not exist: " + p.val.at(k));
}

// --fit reference|lstsq (not in the reference): which solver fits the shift polynomials.  `reference` (default)
// restates NumCpp's Poly1d::fit operation by operation (preproc.h:535-536) so the maps are the reference's;
// `lstsq` solves the same least-squares problem by Householder QR on a scaled abscissa.
int fit_mode(const Parsed &p)
{
    const std::string f = p.str("--fit", "reference");
    if (f == "reference") return OIP_FIT_REFERENCE;
    if (f == "lstsq") return OIP_FIT_LSTSQ;
    throw cli_error(105, "--fit: reference or lstsq expected");
}

void usage()
{
    puts("Optical Satellite Image Pre-Processing/Processing Utility (MI355X build)\n"
         "Usage: oip [OPTIONS] [SUBCOMMAND]\n\n"
         "Options:\n"
         "  -h,--help  -v,--version  --width N  --tiff-compress reference|none|lzw\n"
         "  --pan FILE --mss FILE [--do-rrc4pan --rrc-pan FILE --write-rrcpan/--no-rrcpan] [--no-rrc4mss]\n"
         "  --rrc-msb1 FILE --rrc-msb2 FILE --rrc-msb3 FILE --rrc-msb4 FILE\n"
         "  --slices N --ibc-sections N --ibc-threshold X --line-offset N --lines-section N --overlap-lines N -k,--keep-leading\n"
         "  --fit reference|lstsq   (polynomial fit: the reference's NumCpp formulation [default] or QR least squares)\n\n"
         "Subcommands:\n"
         "  prestitch  --pan1 FILE --pan2 FILE [--rrc1 FILE --rrc2 FILE -s N -l N --stitch-overlap N\n"
         "             --stt-threshold X --stt-maxdeltay X -e N -r,--rrc/--no-rrc -c,--only-calculate --fp16-accumulate]\n"
         "  stitch     --image1 FILE --image2 FILE -c,--fold-cols N [-o,--out FILE] [-g,--GDAL -m,--band-map a,b,c,d]\n"
         "  --gpus N   (default action and prestitch) scan-line blocks over the N GPUs of the node, RCCL exchanges\n"
         "  plan       strip|ccd --width W --lines L --gpus N ...: print the multi-GPU row plan as JSON\n"
         "  task       prestitch + stitch + default action x2 + stitch in one process (intermediates stay on the GPU;\n"
         "             --pan-only: the stitched PAN product alone, RRC and resampling written straight into it):\n"
         "             --pan1 --pan2 --rrc1 --rrc2 --mss1 --mss2 --rrc-mss{1,2}-b{1..4} FILE --fold-cols-pan N --fold-cols-mss N\n"
         "             --out-pan FILE.TIFF --out-mss FILE.TIFF [prestitch, default-action and stitch options]");
}

int run_prestitch(const std::vector<std::string> &args, int width)
{
    Spec sp;
    sp.valued = {"--pan1", "--pan2", "--rrc1", "--rrc2", "--sections", "--section-lines", "--stitch-overlap", "--stt-threshold",
                 "--stt-maxdeltay", "--edge-cols", "--width", "--gpus"};
    sp.flags = {"--rrc", "--no-rrc", "--only-calculate", "--fp16-accumulate"};
    sp.alias = {{"-s", "--sections"}, {"-l", "--section-lines"}, {"-e", "--edge-cols"}, {"-r", "--rrc"}, {"-c", "--only-calculate"}};
    Parsed p = parse(sp, args);
    require(p, "--pan1");
    require(p, "--pan2");
    for (auto k : {"--pan1", "--pan2", "--rrc1", "--rrc2"}) existing_file(p, k);
    width = p.integer("--width", width);
    const int sections = p.integer("--sections", OIP_STT_DEF_SECTIONS);
    const int sectionLines = p.integer("--section-lines", OIP_STT_DEF_SECLINES);
    const int overlapCols = p.integer("--stitch-overlap", OIP_STT_DEF_OVERLAPPX);
    const int edgeCols = p.integer("--edge-cols", 0);
    if (edgeCols < 0 || edgeCols > overlapCols / 2) throw cli_error(105, "--edge-cols: invalid edge cols");      // main.cpp:135-141
    const double thr = p.real("--stt-threshold", OIP_STT_DEF_PHCTHRHLD), maxdy = p.real("--stt-maxdeltay", 0.0);
    const bool doRRC = !p.flag.count("--no-rrc");
    const bool onlyCalc = p.flag.count("--only-calculate") != 0;
    const int gpus = p.integer("--gpus", 1);
    if (gpus < 1 || gpus > 64) throw cli_error(105, "--gpus: GPU count expected");
    if (p.has("--gpus")) {                        // scan-line blocks over the GPUs of the node (oip_multigpu.hpp); --gpus 1 included
        MultiGpuPrestitchOptions mo;
        mo.width = width; mo.gpus = gpus; mo.sections = sections; mo.sectionLines = sectionLines; mo.overlapCols = overlapCols;
        mo.edgeCols = edgeCols; mo.threshold = thr; mo.maxDeltaY = maxdy; mo.doRRC = doRRC; mo.onlyCalc = onlyCalc;
        mo.fp16acc = p.flag.count("--fp16-accumulate") != 0;
        RunPrestitchMultiGpu(p.str("--pan1"), p.str("--pan2"), p.str("--rrc1"), p.str("--rrc2"), mo);
        return 0;
    }
    // main.cpp:270-286
    Stitcher stt(p.str("--pan1"), p.str("--pan2"), p.str("--rrc1"), p.str("--rrc2"), sections, sectionLines, overlapCols, width);
    stt.CalcSttParameters(thr, maxdy, edgeCols);
    if (!onlyCalc) {
        if (doRRC) stt.DoRRC();
        stt.PreStitch(p.flag.count("--fp16-accumulate") != 0);      // not in the reference: BASELINE config 5's resampling variant
    }
    stt.Finish();                                                   // the products' writer threads
    return 0;
}

int run_stitch(const std::vector<std::string> &args, int width)
{
    Spec sp;
    sp.valued = {"--image1", "--image2", "--out", "--fold-cols", "--band-map", "--width"};
    sp.flags = {"--GDAL"};
    sp.alias = {{"-o", "--out"}, {"-c", "--fold-cols"}, {"-g", "--GDAL"}, {"-m", "--band-map"}};
    Parsed p = parse(sp, args);
    require(p, "--image1");
    require(p, "--image2");
    require(p, "--fold-cols");
    width = p.integer("--width", width);
    const int foldCols = p.integer("--fold-cols", 0);
    if (foldCols < 2) throw cli_error(105, "--fold-cols: fold column value too small");                           // main.cpp:166-170
    if (p.has("--band-map") && !p.has("--GDAL")) throw cli_error(107, "--band-map requires --GDAL");              // ->needs(gdal)
    int map[MSS_BANDS] = {0, 0, 0, 0};
    const std::string bandMap = p.str("--band-map");
    if (!bandMap.empty()) {                                                                                      // main.cpp:177-188
        if (sscanf(bandMap.c_str(), "%d,%d,%d,%d", map, map + 1, map + 2, map + 3) != 4) throw cli_error(105, "-m: need 4 band indices");
        for (int i = 0; i < MSS_BANDS; ++i)
            if (map[i] <= 0 || map[i] > MSS_BANDS) throw cli_error(105, "-m: invalid band index");
    }
    Stitcher::Stitch(p.str("--image1"), p.str("--image2"), p.str("--out"), foldCols / 2, width, p.has("--GDAL"),
                     bandMap.empty() ? nullptr : map);                                                           // main.cpp:189
    return 0;
}

int run_default(const std::vector<std::string> &args, int width)
{
    Spec sp;
    sp.valued = {"--pan", "--mss", "--rrc-pan", "--rrc-msb1", "--rrc-msb2", "--rrc-msb3", "--rrc-msb4", "--slices", "--ibc-sections",
                 "--ibc-threshold", "--line-offset", "--lines-section", "--overlap-lines", "--width", "--fit", "--gpus"};
    sp.flags = {"--do-rrc4pan", "--write-rrcpan", "--no-rrcpan", "--no-rrc4mss", "--keep-leading"};
    sp.alias = {{"-k", "--keep-leading"}};
    Parsed p = parse(sp, args);
    for (auto k : {"--pan", "--mss", "--rrc-msb1", "--rrc-msb2", "--rrc-msb3", "--rrc-msb4"}) existing_file(p, k);
    width = p.integer("--width", width);
    const double thr = p.real("--ibc-threshold", OIP_IBCV_DEF_THRESHOLD);
    if (thr < 0.0 || thr >= 1.0) throw cli_error(105, "--ibc-threshold: invalid threshold value");               // main.cpp:233-239
    if ((p.has("--rrc-pan") || p.has("--write-rrcpan") || p.has("--no-rrcpan")) && !p.has("--do-rrc4pan"))
        throw cli_error(107, "--rrc-pan requires --do-rrc4pan");                                                  // ->needs(rrc4pan)
    const bool doRRC4PAN = p.flag.count("--do-rrc4pan") != 0;
    const bool doRRC4MSS = !p.flag.count("--no-rrc4mss");
    // main.cpp:288-299
    if (doRRC4PAN && p.str("--rrc-pan").empty()) throw usage_error("RRC parameter file of PAN needed");
    std::string msb[MSS_BANDS] = {p.str("--rrc-msb1"), p.str("--rrc-msb2"), p.str("--rrc-msb3"), p.str("--rrc-msb4")};
    if (doRRC4MSS && (msb[0].empty() || msb[1].empty() || msb[2].empty() || msb[3].empty()))
        throw usage_error("RRC parameter file of all MSS Bands needed");
    if (p.has("--gpus")) {                        // scan-line blocks over the GPUs of the node (oip_multigpu.hpp); --gpus 1 included
        MultiGpuDefaultOptions mo;
        mo.width = width; mo.gpus = p.integer("--gpus", 1);
        if (mo.gpus < 1 || mo.gpus > 64) throw cli_error(105, "--gpus: GPU count expected");
        if (p.flag.count("--write-rrcpan")) throw cli_error(105, "--write-rrcpan is not available with --gpus");
        require(p, "--pan");
        require(p, "--mss");
        mo.doRRC4PAN = doRRC4PAN; mo.doRRC4MSS = doRRC4MSS; mo.keepLeading = p.flag.count("--keep-leading") != 0;
        mo.slices = p.integer("--slices", OIP_IBCV_DEF_SLICES); mo.sections = p.integer("--ibc-sections", OIP_IBCV_DEF_SECTIONS);
        mo.linesSection = p.integer("--lines-section", OIP_IBPA_DEFAULT_BATCHLINES); mo.lineOffset = p.integer("--line-offset", 0);
        mo.overlapLines = p.integer("--overlap-lines", OIP_IBPA_DEFAULT_LINEOVERLAP); mo.fitMode = fit_mode(p); mo.threshold = thr;
        RunDefaultActionMultiGpu(p.str("--pan"), p.str("--mss"), p.str("--rrc-pan"), msb, mo);
        return 0;
    }
    // main.cpp:301-316
    PreProcessor pp(p.str("--pan"), p.str("--mss"), p.str("--rrc-pan"), msb, width);
    pp.SetFitMode(fit_mode(p));
    const char *pl = getenv("OIP_PIPELINE");
    if (!(pl && atoi(pl) == 0)) {
        // the same steps as one pipeline: read || RRC || correlation || product writes (PreProcessor::RunPipelined);
        // OIP_PIPELINE=0 runs them one after the other as the reference does -- the products are the same bytes
        PreProcessor::DefaultActionOptions o;
        o.doRRC4PAN = doRRC4PAN; o.writeRrcPan = doRRC4PAN && p.flag.count("--write-rrcpan") != 0; o.doRRC4MSS = doRRC4MSS;
        o.keepLeading = p.flag.count("--keep-leading") != 0;
        o.slices = p.integer("--slices", OIP_IBCV_DEF_SLICES); o.sections = p.integer("--ibc-sections", OIP_IBCV_DEF_SECTIONS); o.threshold = thr;
        o.linesSection = p.integer("--lines-section", OIP_IBPA_DEFAULT_BATCHLINES); o.lineOffset = p.integer("--line-offset", 0);
        o.overlapLines = p.integer("--overlap-lines", OIP_IBPA_DEFAULT_LINEOVERLAP);
        pp.RunPipelined(o);
        return 0;
    }
    pp.LoadPAN();
    pp.LoadMSS();
    if (doRRC4PAN) {
        pp.DoRRC4PAN();
        if (p.flag.count("--write-rrcpan")) pp.WriteRRCedPAN();      // RAW instead of the reference's TIFF
    }
    pp.DoRRC4MSS(doRRC4MSS);
    pp.CalcInterBandCorrelation(p.integer("--slices", OIP_IBCV_DEF_SLICES), p.integer("--ibc-sections", OIP_IBCV_DEF_SECTIONS), thr);
    pp.DoInterBandAlignment(p.integer("--lines-section", OIP_IBPA_DEFAULT_BATCHLINES), p.integer("--line-offset", 0),
                            p.integer("--overlap-lines", OIP_IBPA_DEFAULT_LINEOVERLAP), p.flag.count("--keep-leading") != 0);
    return 0;
}

// oip plan strip|ccd ...: prints the multi-GPU row plan as JSON (no device is touched) -- what each rank computes,
// which window pieces and halo lines move.  tests/test_cli_cpu.py compares it with dist.py's plan.
int run_plan(const std::vector<std::string> &args)
{
    if (args.empty() || (args[0] != "strip" && args[0] != "ccd")) throw cli_error(105, "plan: strip or ccd expected");
    Spec sp;
    sp.valued = {"--width", "--lines", "--gpus", "--slices", "--ibc-sections", "--corr-lines", "--lines-section", "--line-offset", "--overlap-lines",
                 "--min-lines", "--halo-cap", "--cy", "--sections", "--section-lines", "--stitch-overlap", "--edge-cols", "--dy", "--section-rows"};
    sp.flags = {"--keep-leading"};
    Parsed p = parse(sp, {args.begin() + 1, args.end()});
    const int W = p.integer("--width", OIP_PIXELS_PER_LINE), gpus = p.integer("--gpus", 1);
    const long L = (long)p.real("--lines", 0);
    if (args[0] == "strip") {
        StripPlanC plan(W, L, gpus, p.integer("--slices", OIP_IBCV_DEF_SLICES), p.integer("--ibc-sections", OIP_IBCV_DEF_SECTIONS),
                        p.integer("--corr-lines", OIP_CORRELATION_LINES), p.integer("--lines-section", OIP_IBPA_DEFAULT_BATCHLINES),
                        p.integer("--line-offset", 0), p.integer("--overlap-lines", OIP_IBPA_DEFAULT_LINEOVERLAP), p.flag.count("--keep-leading") != 0,
                        p.integer("--min-lines", OIP_IBPA_MIN_PROCESSLINES), p.integer("--halo-cap", 64));
        double cy[12] = {0};
        const std::string c = p.str("--cy", "0,0,0");
        double a = 0, b = 0, d = 0;
        if (sscanf(c.c_str(), "%lf,%lf,%lf", &a, &b, &d) != 3) throw cli_error(105, "--cy: three coefficients expected");
        for (int k = 0; k < 4; ++k) { cy[3 * k] = a; cy[3 * k + 1] = b; cy[3 * k + 2] = d; }
        PrintStripPlan(plan, cy);
    } else {
        CcdPlanC plan(W, L, gpus, p.integer("--sections", OIP_STT_DEF_SECTIONS), p.integer("--section-lines", OIP_STT_DEF_SECLINES),
                      p.integer("--stitch-overlap", OIP_STT_DEF_OVERLAPPX), p.integer("--edge-cols", 0),
                      p.integer("--section-rows", OIP_REMAP_SECTION_ROWS));
        PrintCcdPlan(plan, p.real("--dy", 0.0));
    }
    return 0;
}

// oip task: DOC/sample-task.sh's five commands in one process, intermediates resident on the GPU
int run_task(const std::vector<std::string> &args, int width)
{
    Spec sp;
    sp.valued = {"--pan1", "--pan2", "--rrc1", "--rrc2", "--mss1", "--mss2", "--out-pan", "--out-mss", "--fold-cols-pan", "--fold-cols-mss",
                 "--sections", "--section-lines", "--stitch-overlap", "--stt-threshold", "--stt-maxdeltay", "--edge-cols", "--band-map",
                 "--slices", "--ibc-sections", "--ibc-threshold", "--line-offset", "--lines-section", "--overlap-lines", "--width", "--fit"};
    for (int c = 1; c <= 2; ++c)
        for (int b = 1; b <= MSS_BANDS; ++b) sp.valued.insert("--rrc-mss" + std::to_string(c) + "-b" + std::to_string(b));
    sp.flags = {"--GDAL", "--keep-leading", "--fp16-accumulate", "--pan-only"};
    sp.alias = {{"-s", "--sections"}, {"-l", "--section-lines"}, {"-e", "--edge-cols"}, {"-g", "--GDAL"}, {"-m", "--band-map"}, {"-k", "--keep-leading"}};
    Parsed p = parse(sp, args);
    std::string msb[2][MSS_BANDS];
    // --pan-only: the stitched PAN product alone (steps 1-2 of DOC/sample-task.sh).  No corrected strip is needed afterwards,
    // so RRC of CCD 1 and the resampled CCD-2 lines are written straight into the stitched raster (one pass each).
    const bool panOnly = p.flag.count("--pan-only") != 0;
    for (auto k : {"--pan1", "--pan2", "--rrc1", "--rrc2", "--out-pan", "--fold-cols-pan"}) require(p, k);
    if (!panOnly)
        for (auto k : {"--mss1", "--mss2", "--out-mss", "--fold-cols-mss"}) require(p, k);
    for (int c = 0; c < 2 && !panOnly; ++c)
        for (int b = 0; b < MSS_BANDS; ++b) {
            const std::string k = "--rrc-mss" + std::to_string(c + 1) + "-b" + std::to_string(b + 1);
            require(p, k);
            existing_file(p, k);
            msb[c][b] = p.str(k);
        }
    for (auto k : {"--pan1", "--pan2", "--rrc1", "--rrc2", "--mss1", "--mss2"}) existing_file(p, k);
    TaskOptions o;
    o.width = p.integer("--width", width);
    o.sections = p.integer("--sections", o.sections);
    o.sectionLines = p.integer("--section-lines", o.sectionLines);
    o.overlapCols = p.integer("--stitch-overlap", o.overlapCols);
    o.edgeCols = p.integer("--edge-cols", 0);
    if (o.edgeCols < 0 || o.edgeCols > o.overlapCols / 2) throw cli_error(105, "--edge-cols: invalid edge cols");
    o.sttThreshold = p.real("--stt-threshold", o.sttThreshold);
    o.sttMaxDeltaY = p.real("--stt-maxdeltay", 0.0);
    o.foldColsPAN = p.integer("--fold-cols-pan", 0);
    o.foldColsMSS = p.integer("--fold-cols-mss", panOnly ? 2 : 0);
    o.panOnly = panOnly;
    if (o.foldColsPAN < 2 || o.foldColsMSS < 2) throw cli_error(105, "--fold-cols: fold column value too small");
    o.useGDAL = p.has("--GDAL");
    if (p.has("--band-map") && !o.useGDAL) throw cli_error(107, "--band-map requires --GDAL");
    int map[MSS_BANDS] = {0, 0, 0, 0};
    const std::string bandMap = p.str("--band-map");
    if (!bandMap.empty()) {
        if (sscanf(bandMap.c_str(), "%d,%d,%d,%d", map, map + 1, map + 2, map + 3) != 4) throw cli_error(105, "-m: need 4 band indices");
        for (int i = 0; i < MSS_BANDS; ++i)
            if (map[i] <= 0 || map[i] > MSS_BANDS) throw cli_error(105, "-m: invalid band index");
        o.bandMap = map;
    }
    o.slices = p.integer("--slices", o.slices);
    o.ibcSections = p.integer("--ibc-sections", o.ibcSections);
    o.ibcThreshold = p.real("--ibc-threshold", o.ibcThreshold);
    if (o.ibcThreshold < 0.0 || o.ibcThreshold >= 1.0) throw cli_error(105, "--ibc-threshold: invalid threshold value");
    o.linesSection = p.integer("--lines-section", o.linesSection);
    o.lineOffset = p.integer("--line-offset", 0);
    o.overlapLines = p.integer("--overlap-lines", o.overlapLines);
    o.keepLeading = p.flag.count("--keep-leading") != 0;
    o.fitMode = fit_mode(p);
    o.fp16acc = p.flag.count("--fp16-accumulate") != 0;
    for (auto k : {"--out-pan", "--out-mss"})
        if (p.has(k) && to_lower(std::filesystem::path(p.str(k)).extension().string()) != ".tiff") throw std::invalid_argument("Output file should be a tiff image");
    RunFusedTask(p.str("--pan1"), p.str("--pan2"), p.str("--rrc1"), p.str("--rrc2"), p.str("--mss1"), p.str("--mss2"), msb[0], msb[1],
                 p.str("--out-pan"), p.str("--out-mss"), o);
    return 0;
}

}  // namespace

static int oip_main(int argc, const char *argv[]);

// The products are on disk and the log is flushed when oip_main returns; what is left is tear-down -- hipFree of ~10 GB,
// un-pinning the staging ring, the HIP runtime's own shutdown (measured: DESIGN.md 4.5) -- which the kernel does faster
// for a process that simply leaves.  OIP_FAST_EXIT=0 runs the destructors.
int main(int argc, const char *argv[])
{
    process_start();
    const int rc = oip_main(argc, argv);
    fflush(stdout);
    if (log_file()) fflush(log_file());
    const char *fe = getenv("OIP_FAST_EXIT");
    if (!(fe && atoi(fe) == 0)) _exit(rc);
    return rc;
}

static int oip_main(int argc, const char *argv[])
{
    try {
        const char *lf = getenv("LOGFILE");                             // main.cpp:322-329
        log_file() = fopen(lf ? lf : "oip.log", "a");
        std::vector<std::string> args(argv + 1, argv + argc);
        int width = OIP_PIXELS_PER_LINE;
        try {
            for (auto &a : args) {
                if (a == "-h" || a == "--help") { usage(); return 255; }
                if (a == "-v" || a == "--version") { puts("1.1"); return 255; }
            }
            // --tiff-compress reference|none|lzw (not in the reference; anywhere on the line)
            for (size_t i = 0; i < args.size(); ++i) {
                std::string v;
                if (args[i] == "--tiff-compress" && i + 1 < args.size()) { v = args[i + 1]; args.erase(args.begin() + i, args.begin() + i + 2); }
                else if (args[i].rfind("--tiff-compress=", 0) == 0) { v = args[i].substr(16); args.erase(args.begin() + i); }
                else continue;
                if (v == "reference") tiff_policy() = -1;
                else if (v == "none") tiff_policy() = TIFF_NONE;
                else if (v == "lzw") tiff_policy() = TIFF_LZW;
                else throw cli_error(105, "--tiff-compress: reference, none or lzw expected");
                break;
            }
            if (!args.empty() && args[0] == "prestitch") return run_prestitch({args.begin() + 1, args.end()}, width);
            if (!args.empty() && args[0] == "stitch") return run_stitch({args.begin() + 1, args.end()}, width);
            if (!args.empty() && args[0] == "task") return run_task({args.begin() + 1, args.end()}, width);
            if (!args.empty() && args[0] == "plan") return run_plan({args.begin() + 1, args.end()});
            if (!args.empty() && args[0] == "auxsep")
                throw std::invalid_argument("auxsep (down-link de-framing) is outside this build: run the reference's auxsep, then this tool");
            if (args.empty()) { usage(); return 0; }
            return run_default(args, width);
        } catch (const cli_error &e) {
            fprintf(stderr, "%s\nRun with --help for more information.\n", e.what());
            return e.code;
        }
    } catch (usage_error &ex) {
        printf("USAGE ERROR: %s.\n", ex.what());
        return 254;
    } catch (std::exception &ex) {
        OLOG("[ERROR] %s.", ex.what());
        return 2;
    } catch (...) {
        OLOG("[FATAL] UNKOWN FATAL ERROR OCCURED.");
        return 1;
    }
}
