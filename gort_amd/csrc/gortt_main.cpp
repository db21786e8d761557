// gortt_main.cpp -- `gortt [options] < angles.dat > output.dat`, the drop-in executable.
//
// Keeps the command line, stdin/stdout formats, -W/-P probability LUT files, messages
// and exit codes of the reference's main() / gortt_cl_parser (gortt.c:9-382,1003-1136);
// all numerics run on the GPU through libgort_amd.so (include/gort_amd.h).  Angle lines
// are batched to the device instead of being evaluated one by one.
//
// Deliberate deviations (DESIGN.md "CLI deviations"):
//   * a value flag given as the last argument is an error (the reference reads argv[argc]
//     and crashes);
//   * the header line may be longer than 999 characters (the reference cannot take more
//     than ~190 wavelengths per run);
//   * -energy works for any number of wavelengths (the reference overflows its heap
//     beyond 32);
//   * -soil_spectra, which in the reference only dumps a table and exits with failure,
//     is rejected with a message.
// Extensions (double-dash, unknown options to the reference): --binary-in, --binary-out, --lut-hex, --lut-cache DIR, --gpus N, --help.
//
// The per-line loop of the reference (read, evaluate, print: gortt.c:232-329) is a three-stage pipeline here:
// this thread reads and parses chunk i+1 into a pinned slot of a gort_pipe while the GPU evaluates chunk i and
// a second thread formats and writes chunk i-1; the copies to and from the device run on streams of their own.
#include <algorithm>
#include <cctype>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <sched.h>
#include <string>
#include <strings.h>
#include <sys/uio.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "gort_amd.h"

namespace {

const char *g_prog = "gortt";

[[noreturn]] void die(const char *fmt, const char *a = nullptr, const char *b = nullptr)
{
    std::fflush(stdout);
    std::fprintf(stderr, fmt, a, b);
    std::exit(EXIT_FAILURE);
}

void usage(const char *bin)
{
    // text of gortt_usage (gortt.c:1140-1234), printed on stderr
    std::fprintf(stderr, "usage: %s [options] < angles.dat\n\n", bin);
    std::fputs(
        "The first line of the input data reads:\nN M W_1 W_2 [...] W_M\n"
        "where N is the number of view--illumination geometries\nM is the number of wavelengths and\n"
        "W_i (i=1,M) are the wavelengths at which to predict the canopy reflectance\n"
        "The rest of the input data is four columns of ascii:\nview_zenith view_azimuth solar_zenith solar azimuth\n\n"
        "The command line options are:\n"
        "\n============ Crown geometry options:\n"
        "-beta arg\tforce the proportion of mutual shadowing to arg\n"
        "         \t[n.b. if beta is not set gortt uses the model by Li and Strahler (IGARSS'92) to determine mutual shadowing]\n"
        "\n------------ EITHER (old style):\n"
        "-h1 arg    \tset the lower boundary of the crown centres (m) to arg\n"
        "-h2 arg    \tset the upper boundary of the crown centres (m) to arg\n"
        "-b arg     \tset the vertical crown radius (m) to arg\n"
        "-r arg     \tset the horizontal crown radius (m) to arg\n"
        "-lambda arg\tset the tree stem density (1/m2) to arg\n"
        "\n------------ OR (new style):\n"
        "-HB  arg\tset the ratio of the centroid height range to the vertical crown radius to arg\n"
        "-BR  arg\tset the ratio of the vertical to horizontal crown radius to arg\n"
        "-PCC arg\tset the projected crown cover (at nadir) to arg\n"
        "\nThe above old style options refer to the original GORT paper and the \n"
        "new style options are as they are expressed in the Quaife et al. (2008) \n"
        "DALEC paper. As soon as a new style option is specified all old style \n"
        "options are ignored. Note that they are same thing - the code simply uses\n"
        "the new style options to calculate the old ones but the new ones are \n"
        "preferred because it reduces equifinality. \n"
        "\n============ Amount of leaf material:\n"
        "------------ EITHER:\n"
        "-favd arg\tset the foliage volume area density (1/m2) within crown to arg\n"
        "\n------------ OR:\n"
        "-LAI  arg\tset the leaf area index (m2/m2) for the scene to arg\n"
        "\n============ Prospect leaf options:\n"
        "-N arg  \tset the leaf structure variable to arg\n"
        "-Cab arg\tset the leaf chlorophyl content (\xc2\xb5g.cm-2) to arg\n"
        "-Cw arg \tset the equivelant leaf water thickness (cm) to arg\n"
        "-Car arg \tset the carotenoid content (\xc2\xb5g.cm-2) to arg\n"
        "-Anth arg \tset the anthocyanin content (\xc2\xb5g.cm-2) to arg\n"
        "-Cbrown arg \tset the /brown pigment content (arbitrary units) to arg\n"
        "-Cm arg \tset the leaf mass per unit area (g.cm-2) to arg\n"
        "\n============ Price soil spectra options:\n"
        "-rsl1 arg \tset the weight of the first soil vector to arg\n"
        "-rsl2 arg \tset the weight of the second soil vector to arg\n"
        "-rsl3 arg \tset the weight of the third soil vector to arg\n"
        "-rsl4 arg \tset the weight of the fourth soil vector to arg\n"
        "\n============ User override for spectral properties:\n"
        "-alb_leaf     arg \tset leaf albedo to arg (turns prospect off)**\n"
        "-alb_soil     arg \tset soil albedo to arg (turns price off)**\n"
        "-soil_spectra arg \tread soil spectra from file arg (turns price off)\n"
        "                  \t[** n.b. use only one of the above soil options]\n"
        "\n============ Read/write gap probabilities:\n"
        "-W       \tgo as far as calculating the gap probabilities and write these to the stdout and exit\n"
        "-P file  \tread gap probabilities from file that have been written using the -W option\n"
        "         \tn.b. the above two options are included to allow fast BRF calculation based on\n"
        "         \tpre-computed gap probabilities. The model must be run using the same crown and canopy\n"
        "         \tgeometry options for the read and write. Spectral options (i.e. Prospect and Price options)\n"
        "         \tmay be varied whist running using a give probability file.\n"
        "\n============ Input/output options:\n"
        "-prnspec\tprint the scene component spectra for each wavelength inside {}\n"
        "-prnprop\tprint the viewed proportions of scene components inside []\n"
        "-energy \tprint the spectral albedo, absoption by veg and absorption by soil for each wavelength after other outputs.\n"
        "-u      \tprint this message and exit\n"
        "\n",
        stderr);
}

struct Options {
    gort_canopy canopy;
    gort_leaf_soil leaf;
    bool prnspec = false, prnprop = false, energy = false, write_lut = false, read_lut = false;
    // extensions (double dash: the reference rejects them as unknown options, so no valid reference
    // command line changes meaning)
    bool binary_in = false;    // --binary-in : after the text header, angle lines are records of 4 raw doubles
    bool binary_out = false;   // --binary-out: rows are raw doubles in print order (angles, then per band ..., K, energy)
    bool lut_hex = false;      // --lut-hex   : -W writes C99 hex floats (exact; -P of either program reads them)
    std::string lut_cache;     // --lut-cache DIR : gap tables kept per crown geometry in DIR (gort_lut_cache_*)
    std::vector<int> devices;  // --gpus N (devices 0..N-1) or GORTT_DEVICES="0,2,3": chunks go round the devices
    std::string lut_file;
};

// Prefix rules and their ORDER are those of gortt_cl_parser (gortt.c:1022-1115).
void parse_args(int argc, char **argv, Options &o)
{
    bool use_true_p = false, use_lai = false;
    float hb = 2.0f, br = 1.0f, pcc = 0.5f, lai = 2.0f;
    auto ci = [](const char *a, const char *flag, size_t n) { return !strncasecmp(a, flag, n); };
    auto cs = [](const char *a, const char *flag, size_t n) { return !strncmp(a, flag, n); };
    for (int i = 1; i < argc; ++i) {
        const char *a = argv[i];
        if (*a != '-') {
            // (sic) the reference reports argv[1], not the offending argument
            std::fprintf(stderr, "%s: unknown argument on command line: %s\n", argv[0], argv[1]);
            std::fprintf(stderr, "(use the option -u to see brief usage instructions)\n");
            std::exit(EXIT_FAILURE);
        }
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "%s: option %s needs a value\n", argv[0], a);
                std::exit(EXIT_FAILURE);
            }
            return argv[++i];
        };
        if (!std::strcmp(a, "--binary-in")) o.binary_in = true;
        else if (!std::strcmp(a, "--binary-out")) o.binary_out = true;
        else if (!std::strcmp(a, "--help")) {
            // -u prints the reference's text and nothing else (compared byte for byte); the extensions are listed here
            usage(argv[0]);
            std::fputs("============ Extensions of this implementation (the reference rejects double-dash options):\n"
                       "--binary-in      \tangle records on stdin as raw doubles (vza vaa sza saa) behind the text header line\n"
                       "--binary-out     \toutput rows as raw doubles, same field order as the text row\n"
                       "--lut-hex        \twith -W: write the gap probabilities as C99 hex floats (exact; -P reads them)\n"
                       "--lut-cache DIR  \tkeep / look up the gap probabilities per crown geometry in DIR\n"
                       "--gpus N         \tsend the chunks of the stream round N GPUs (or GORTT_DEVICES=0,2,..), rows in input order\n"
                       "environment: GORTT_CHUNK_MB, GORTT_THREADS, GORTT_VERBOSE=1 (stage times on stderr)\n\n",
                       stderr);
            std::exit(EXIT_SUCCESS);
        }
        else if (!std::strcmp(a, "--lut-hex")) o.lut_hex = true;
        else if (!std::strcmp(a, "--lut-cache")) o.lut_cache = val();
        else if (!std::strcmp(a, "--gpus")) {
            const int n = atoi(val());
            if (n < 1 || n > 64) {
                std::fprintf(stderr, "%s: --gpus needs a count between 1 and 64\n", argv[0]);
                std::exit(EXIT_FAILURE);
            }
            o.devices.clear();
            for (int d = 0; d < n; ++d) o.devices.push_back(d);
        }
        else if (ci(a, "-favd", 5)) o.canopy.favd = atof(val());
        else if (ci(a, "-h1", 3)) o.canopy.h1 = atof(val());
        else if (ci(a, "-h2", 3)) o.canopy.h2 = atof(val());
        else if (ci(a, "-lambda", 7)) o.canopy.lambda = atof(val());
        else if (cs(a, "-HB", 3)) { use_true_p = true; hb = (float)atof(val()); }
        else if (cs(a, "-BR", 3)) { use_true_p = true; br = (float)atof(val()); }
        else if (cs(a, "-PCC", 7)) { use_true_p = true; pcc = (float)atof(val()); }
        else if (cs(a, "-LAI", 7)) { use_lai = true; lai = (float)atof(val()); }
        else if (ci(a, "-beta", 5)) { o.canopy.use_user_beta = 1; o.canopy.beta = atof(val()); }
        else if (ci(a, "-diffuse", 5)) { o.canopy.use_user_fd = 1; o.canopy.fd_user = 1.0 - atof(val()); }
        else if (cs(a, "-alb_leaf", 9)) { o.leaf.use_alb_leaf = 1; o.leaf.alb_leaf = atof(val()); }
        else if (cs(a, "-alb_soil", 9)) { o.leaf.use_alb_soil = 1; o.leaf.alb_soil = atof(val()); }
        else if (cs(a, "-soil_spectra", 10)) {
            std::fprintf(stderr, "%s: -soil_spectra is not supported (in the reference it only dumps the table and fails)\n", argv[0]);
            std::exit(EXIT_FAILURE);
        }
        else if (cs(a, "-prnspec", 7)) o.prnspec = true;
        else if (cs(a, "-prnprop", 7)) o.prnprop = true;
        else if (cs(a, "-energy", 7)) o.energy = true;
        else if (cs(a, "-q08_pn_kopen", 7)) o.canopy.use_q08 = 1;
        else if (cs(a, "-lidar", 6)) { /* accepted; the reference's lidar routine is an empty stub */ }
        else if (cs(a, "-P", 2)) { o.read_lut = true; o.lut_file = val(); }
        else if (cs(a, "-W", 2)) o.write_lut = true;
        else if (ci(a, "-N", 2)) o.leaf.N = atof(val());
        else if (ci(a, "-cab", 4)) o.leaf.Cab = atof(val());
        else if (ci(a, "-car", 4)) o.leaf.Car = atof(val());
        else if (ci(a, "-canth", 3)) o.leaf.Anth = atof(val());
        else if (ci(a, "-cbrown", 3)) o.leaf.Cbrown = atof(val());
        else if (ci(a, "-cw", 3)) o.leaf.Cw = atof(val());
        else if (ci(a, "-cm", 3)) o.leaf.Cm = atof(val());
        else if (ci(a, "-rsl1", 5)) o.leaf.rsl[0] = atof(val());
        else if (ci(a, "-rsl2", 5)) o.leaf.rsl[1] = atof(val());
        else if (ci(a, "-rsl3", 5)) o.leaf.rsl[2] = atof(val());
        else if (ci(a, "-rsl4", 5)) o.leaf.rsl[3] = atof(val());
        else if (ci(a, "-b", 2)) o.canopy.b = atof(val());
        else if (ci(a, "-r", 2)) o.canopy.r = atof(val());
        else if (ci(a, "-u", 2)) { usage(argv[0]); std::exit(EXIT_SUCCESS); }
        else {
            std::fprintf(stderr, "%s: unknown option on command line: %s\n", argv[0], a);
            std::fprintf(stderr, "(use the option -u to see brief usage instructions)\n");
            std::exit(EXIT_FAILURE);
        }
    }
    if (o.devices.empty())
        if (const char *v = std::getenv("GORTT_DEVICES")) {
            // a list of device ordinals; one may appear more than once (two pipes on one GPU)
            for (const char *q = v; *q;) {
                char *end;
                const long d = std::strtol(q, &end, 10);
                if (end == q || d < 0) break;
                o.devices.push_back((int)d);
                q = *end == ',' ? end + 1 : end;
                if (*end && *end != ',') break;
            }
        }
    if (use_true_p) gort_canopy_newstyle(&o.canopy, hb, br, pcc);
    if (use_lai) gort_canopy_set_lai(&o.canopy, lai);
}

bool read_line(FILE *fp, std::string &line)
{
    static char *buf = nullptr;
    static size_t cap = 0;
    const ssize_t n = getline(&buf, &cap, fp);
    if (n <= 0) { line.clear(); return false; }
    line.assign(buf, (size_t)n);
    return true;
}

// Angle lines in bulk: stdin is read in 8 MB blocks, lines are found with memchr and NUL-terminated in place (strtod
// must not run on into the next line) - a getline() per line was half of the text mode's run time.  A last line
// without a newline counts, as with fgets (gortt.c:232).
struct LineReader {
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    bool eof = false;
    LineReader() : buf((8u << 20) + 1) {}
    // up to max_lines lines: their offsets into buf (valid until the next call); fewer only at end of input
    long take(FILE *fp, long max_lines, std::vector<size_t> &offsets)
    {
        offsets.clear();
        while ((long)offsets.size() < max_lines) {
            char *nl = end > pos ? (char *)std::memchr(buf.data() + pos, '\n', end - pos) : nullptr;
            if (nl) {
                *nl = '\0';
                offsets.push_back(pos);
                pos = (size_t)(nl - buf.data()) + 1;
                continue;
            }
            if (eof) {
                if (end > pos) {                         // last line, no newline
                    buf[end] = '\0';
                    offsets.push_back(pos);
                    pos = end;
                }
                break;
            }
            // out of complete lines: keep what is in use (the lines already handed out and the partial one),
            // make room behind it and read on
            const size_t base = offsets.empty() ? pos : offsets[0];
            if (base > 0) {
                std::memmove(buf.data(), buf.data() + base, end - base);
                for (size_t &o : offsets) o -= base;
                pos -= base;
                end -= base;
            }
            if (buf.size() - 1 - end < (4u << 20)) buf.resize(buf.size() * 2 + 1);
            const size_t got = std::fread(buf.data() + end, 1, buf.size() - 1 - end, fp);
            end += got;
            if (got == 0) eof = true;
        }
        return (long)offsets.size();
    }
};

// "%lf %lf %lf %lf" of the reference's sscanf (gortt.c:234): four numbers, anything after them ignored.
// Lines whose first four fields are PLAIN decimal numbers standing alone ([+-]digits[.digits][e[+-]digits] followed
// by white space or the end of the line - every line a program writes) are converted with strtod, for which scanf and
// strtod agree by definition.  Everything else - hex floats, inf / nan(...), a dangling exponent marker ("1e", "0x1p"),
// junk glued to a number, "0x." - goes through sscanf itself: scanf's greedy matching differs from strtod's longest
// valid prefix in exactly those corners (it swallows "1e" and "0x" + nothing, refuses "nan()" and "infinit"), and the
// reference's behaviour there, error or shifted fields, is whatever its scanf does.
static bool plain_field(const char *&p)
{
    while (std::isspace((unsigned char)*p)) ++p;
    if (*p == '+' || *p == '-') ++p;
    int digits = 0;
    while (std::isdigit((unsigned char)*p)) { ++p; ++digits; }
    if (*p == '.') {
        ++p;
        while (std::isdigit((unsigned char)*p)) { ++p; ++digits; }
    }
    if (!digits) return false;
    if (*p == 'e' || *p == 'E') {
        ++p;
        if (*p == '+' || *p == '-') ++p;
        if (!std::isdigit((unsigned char)*p)) return false;
        while (std::isdigit((unsigned char)*p)) ++p;
    }
    return *p == '\0' || std::isspace((unsigned char)*p);
}

bool parse_angles(const char *s, double v[4])
{
    const char *p = s;
    bool plain = true;
    for (int q = 0; q < 4 && plain; ++q) plain = plain_field(p);
    if (!plain) return std::sscanf(s, "%lf %lf %lf %lf", v, v + 1, v + 2, v + 3) == 4;
    for (int q = 0; q < 4; ++q) {
        char *end;
        v[q] = std::strtod(s, &end);
        s = end;
    }
    return true;
}

// whitespace-separated tokens, as get_first_string_token hands them out (gortt.c:1237-1283)
std::vector<std::string> tokens(const std::string &line)
{
    std::vector<std::string> t;
    size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && std::isspace((unsigned char)line[i])) ++i;
        size_t j = i;
        while (j < line.size() && !std::isspace((unsigned char)line[j])) ++j;
        if (j > i) t.emplace_back(line, i, j - i);
        i = j;
    }
    return t;
}

struct Out {
    std::vector<char> buf;                   // raw bytes; `len` of them are text
    size_t len = 0;
    char *room(size_t n)                     // at least n writable bytes behind the text
    {
        if (buf.size() < len + n) buf.resize((len + n) * 2 + 4096);
        return buf.data() + len;
    }
    void text(const char *s, size_t n) { std::memcpy(room(n), s, n); len += n; }
    void nums(const double *v, long n)       // computed values: "%f " each, NaN as -nan (x86 default NaN)
    {
        long k = gort_format_f6_row(v, n, room((size_t)n * 24 + 8), (size_t)n * 24 + 8);
        if (k < 0) {                         // a value beyond 4e9 in the row: one by one (up to 318 characters each)
            k = 0;
            for (long i = 0; i < n; ++i) {
                char *o = room(400);
                int m = gort_format_f6(v[i], o);
                o[m++] = ' ';
                len += (size_t)m;
            }
        }
        len += (size_t)k;
    }
    void num(double v) { nums(&v, 1); }
    void raw(double v)                       // echoed input: plain "%f " (a NaN typed in prints as printf prints it)
    {
        if (!std::isnan(v)) { nums(&v, 1); return; }
        char *o = room(400);
        int n = std::snprintf(o, 399, "%f", v);
        o[n++] = ' ';
        len += (size_t)n;
    }
    void flush()
    {
        std::fwrite(buf.data(), 1, len, stdout);
        len = 0;
    }
};

// worker threads for formatting `values` numbers: one per ~16k values, at most the cores we may run on
unsigned format_threads(size_t values)
{
    if (const char *v = std::getenv("GORTT_THREADS")) {
        const int t = atoi(v);
        if (t > 0) return (unsigned)t;
    }
    static const unsigned hw = [] {
        unsigned n = std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) n = (unsigned)CPU_COUNT(&set);
        // a container's CPU share (cgroup v2 cpu.max / v1 cfs quota): more threads than that only contend
        // (1M lines x 180 bands on a 16-core share of a 256-thread host: 16 threads 0.89 s, 32 1.13 s, 64 1.46 s)
        long quota = -1, period = -1;
        if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32];
            if (std::fscanf(f, "%31s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = atol(q);
            std::fclose(f);
        } else if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(g, "%ld", &quota) != 1) quota = -1;
            std::fclose(g);
            if (FILE *h = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(h, "%ld", &period) != 1) period = -1;
                std::fclose(h);
            }
        }
        if (quota > 0 && period > 0) {
            const unsigned share = (unsigned)((quota + period - 1) / period);
            if (share >= 1 && share < n) n = share;
        }
        if (n == 0) n = 1;
        return n > 32 ? 32u : n;
    }();
    const size_t want = values / 16384;
    return want < 2 ? 1u : (want < hw ? (unsigned)want : hw);
}

void check(int rc)
{
    if (rc != GORT_OK) die("%s: %s\n", g_prog, gort_last_error());
}

}  // namespace

int main(int argc, char **argv)
{
    g_prog = argv[0];
    const auto t_main = std::chrono::steady_clock::now();
    Options o;
    gort_canopy_defaults(&o.canopy);
    gort_leaf_soil_defaults(&o.leaf);
    parse_args(argc, argv, o);
    check(gort_canopy_init(&o.canopy));

    // 1) gap probabilities on the device unless they come from a file (gortt.c:116-120)
    if (!o.read_lut) {
        const bool cached = !o.lut_cache.empty() && gort_lut_cache_load(o.lut_cache.c_str(), &o.canopy) == GORT_OK;
        if (!cached) {
            check(gort_gap_probabilities(&o.canopy, 1));
            // a cache that cannot be written is not an error of the run
            if (!o.lut_cache.empty() && gort_lut_cache_store(o.lut_cache.c_str(), &o.canopy) != GORT_OK)
                std::fprintf(stderr, "%s: warning: %s\n", argv[0], gort_last_error());
        }
        if (getenv("GORTT_VERBOSE") && !o.lut_cache.empty())
            std::fprintf(stderr, "gortt: gap tables %s %s/gap-%016llx.lut\n", cached ? "from" : "computed, kept in", o.lut_cache.c_str(),
                         (unsigned long long)gort_canopy_key(&o.canopy));
    }
    // 2) -W: write them and stop, before stdin is touched (gortt.c:123-128)
    if (o.write_lut) {
        if (o.lut_hex) {
            // same rows, exact: "%a" keeps every bit, and fscanf("%lf") of the reference parses it, so the
            // horizon values that "%0.40f" flushes to zero (p_n0 at 89 deg ~ 4e-65) survive the round trip
            for (int j = 0; j < 90; ++j) std::printf("%d %a %a\n", j, o.canopy.p_n0[j], o.canopy.epgap[j]);
            std::printf("-1 %a %a\n", o.canopy.k_open, o.canopy.k_openep);
            return EXIT_SUCCESS;
        }
        std::vector<char> text(1 << 15);
        long n = gort_lut_format(&o.canopy, text.data(), text.size());
        if (n < 0) check((int)n);
        std::fwrite(text.data(), 1, (size_t)n, stdout);
        return EXIT_SUCCESS;
    }
    // 3) -P file (gortt.c:131-146)
    if (o.read_lut && gort_lut_read(o.lut_file.c_str(), &o.canopy) != GORT_OK)
        die("%s: error opening probability file: %s\n", argv[0], o.lut_file.c_str());

    // header: N M W_1 .. W_M (gortt.c:153-184)
    std::string header;
    if (!read_line(stdin, header)) die("%s: error reading data on stdin\n", argv[0]);
    std::vector<std::string> tk = tokens(header);
    if (tk.empty()) die("%s: error reading number of angles from line 1\n", argv[0]);
    const int na_check = atoi(tk[0].c_str());
    if (tk.size() < 2) die("%s: error reading number of wavebands from line 1\n", argv[0]);
    const int nw_check = atoi(tk[1].c_str());
    std::vector<double> wl;
    for (size_t i = 2; i < tk.size(); ++i) wl.push_back(atof(tk[i].c_str()));
    const int nw = (int)wl.size();
    if (nw_check != nw) {
        std::fprintf(stderr, "%s: expected number of wavelengths (%d) does not match with number found (%d)\n",
                     argv[0], nw_check, nw);
        return EXIT_FAILURE;
    }

    std::vector<double> rsoil(nw), rleaf(nw), tleaf(nw);
    if (nw > 0 && gort_spectra(&o.leaf, wl.data(), nw, rsoil.data(), rleaf.data(), tleaf.data()) != GORT_OK)
        die("%s\n", gort_last_error());

    // one engine + one pipe of chunks in flight per device (--gpus N / GORTT_DEVICES="0,1,..": chunk k goes to
    // device k mod N, rows leave in input order)
    std::vector<int> devices = o.devices;
    if (devices.empty()) devices.push_back(-1);                 // -1: whatever device is current
    for (int d : devices)
        if (d >= gort_device_count()) {
            std::fflush(stdout);
            std::fprintf(stderr, "%s: device %d asked for (--gpus / GORTT_DEVICES), this machine has %d\n", argv[0], d, gort_device_count());
            return EXIT_FAILURE;
        }
    const size_t per_line_out = (size_t)nw * (1 + (o.prnspec ? 4 : 0) + (o.energy ? 3 : 0)) + 8;
    size_t chunk_mb = 48;                                        // GORTT_CHUNK_MB: output bytes per chunk
    if (const char *v = std::getenv("GORTT_CHUNK_MB")) { const long m = atol(v); if (m >= 1 && m <= 4096) chunk_mb = (size_t)m; }
    const bool verbose = std::getenv("GORTT_VERBOSE") != nullptr;   // stage timings on stderr
    const auto t_start = std::chrono::steady_clock::now();
    const double t_before = std::chrono::duration<double>(t_start - t_main).count();    // flags, gap tables (first HIP call), header, spectra
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    double t_acquire = 0, t_read = 0, t_submit = 0, t_wait = 0, t_write = 0;
    long CHUNK = (long)((chunk_mb << 20) / (sizeof(double) * per_line_out));
    CHUNK = CHUNK < 1024 ? 1024 : (CHUNK > 262144 ? 262144 : CHUNK);
    // small inputs: small pinned buffers and one slot (a stream longer than its header says still works, unpipelined)
    if (na_check > 0 && na_check < CHUNK) CHUNK = na_check < 256 ? 256 : na_check;
    const int depth = (na_check > 0 && na_check <= CHUNK) ? 1 : 3;
    const unsigned pflags = (o.prnspec ? GORT_PIPE_SCOMP : 0u) | (o.energy ? GORT_PIPE_ENERGY : 0u);
    struct Dev { gort_engine *eng = nullptr; gort_pipe *pipe = nullptr; int id = -1; };
    std::vector<Dev> devs(devices.size());
    for (size_t d = 0; d < devs.size(); ++d) {
        devs[d].id = devices[d];
        if (devices[d] >= 0) check(gort_set_device(devices[d]));
        check(gort_engine_create(&devs[d].eng));
        check(gort_engine_set_canopy(devs[d].eng, &o.canopy));
        if (nw > 0) check(gort_engine_set_spectra(devs[d].eng, nw, rsoil.data(), rleaf.data(), tleaf.data()));
        check(gort_pipe_create(devs[d].eng, CHUNK, depth, pflags, &devs[d].pipe));
    }

    std::fputs(header.c_str(), stdout);
    std::fflush(stdout);
    const double t_setup = since(t_start);

    // ---- consumer: collects the chunks in order, formats (or gathers, --binary-out) and writes them ----
    std::mutex mu;
    std::condition_variable cv;
    long chunks_submitted = 0;
    bool producer_done = false;
    std::string consumer_error;
    long na = 0;
    auto format_lines = [&](const gort_pipe_chunk &c, long a0, long a1, Out &dst) {
        for (long a = a0; a < a1; ++a) {
            for (int q = 0; q < 4; ++q) dst.raw(c.angles[4 * a + q]);
            if (!o.prnspec) {
                dst.nums(c.rsurf + (size_t)a * nw, nw);
            } else {
                for (int i = 0; i < nw; ++i) {
                    dst.num(c.rsurf[(size_t)a * nw + i]);
                    dst.text("{ ", 2);
                    dst.nums(c.scomp + ((size_t)a * nw + i) * 4, 4);
                    dst.text("} ", 2);
                }
            }
            if (o.prnprop) {
                dst.text("[ ", 2);
                dst.nums(c.K + 4 * a, 4);            // without wavelengths too (`N 0`): gortt.c:424-449 run in front of the loop
                dst.text("] ", 2);
            }
            if (o.energy) dst.nums(c.energy + (size_t)a * nw * 3, 3L * nw);
            dst.text("\n", 1);
        }
    };
    // --binary-out rows are gathered with writev straight from the pinned buffers (same field order as the text row)
    auto write_binary = [&](const gort_pipe_chunk &c) -> bool {
        std::vector<struct iovec> iov;
        iov.reserve(1024);
        auto flush = [&]() -> bool {
            size_t i = 0;
            while (i < iov.size()) {
                const ssize_t w = writev(STDOUT_FILENO, &iov[i], (int)std::min<size_t>(iov.size() - i, 1024));
                if (w < 0) { if (errno == EINTR) continue; return false; }
                size_t left = (size_t)w;
                while (left > 0 && i < iov.size()) {
                    if (left >= iov[i].iov_len) { left -= iov[i].iov_len; ++i; }
                    else { iov[i].iov_base = (char *)iov[i].iov_base + left; iov[i].iov_len -= left; left = 0; }
                }
            }
            iov.clear();
            return true;
        };
        auto add = [&](const double *p, size_t n) { if (n) iov.push_back({(void *)p, n * sizeof(double)}); };
        for (long a = 0; a < c.n; ++a) {
            add(c.angles + 4 * a, 4);
            if (!o.prnspec) {
                add(c.rsurf + (size_t)a * nw, (size_t)nw);
            } else {
                for (int i = 0; i < nw; ++i) {
                    add(c.rsurf + (size_t)a * nw + i, 1);
                    add(c.scomp + ((size_t)a * nw + i) * 4, 4);
                }
            }
            if (o.prnprop) add(c.K + 4 * a, 4);
            if (o.energy) add(c.energy + (size_t)a * nw * 3, (size_t)nw * 3);
            if (iov.size() + 8 + (o.prnspec ? 2 * (size_t)nw : 0) > 1024 && !flush()) return false;
        }
        return flush();
    };
    std::thread consumer([&] {
        long k = 0;
        Out out;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return chunks_submitted > k || producer_done; });
                if (chunks_submitted <= k) break;
            }
            Dev &dv = devs[(size_t)(k % (long)devs.size())];
            if (dv.id >= 0) gort_set_device(dv.id);
            gort_pipe_chunk c;
            const auto tw = std::chrono::steady_clock::now();
            const int wrc = gort_pipe_wait(dv.pipe, &c);
            t_wait += since(tw);
            const auto tf = std::chrono::steady_clock::now();
            if (wrc != GORT_OK) {
                std::lock_guard<std::mutex> lk(mu);
                if (consumer_error.empty()) consumer_error = gort_last_error();
                gort_pipe_release(dv.pipe);
                ++k;
                continue;
            }
            const long n = c.n;
            if (o.binary_out) {
                if (!write_binary(c)) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (consumer_error.empty()) consumer_error = "write error on stdout";
                }
            } else {
                // text rows: the formatting (exact "%f", gort_format_f6) is the slowest stage of the whole program, so
                // the lines of a chunk are formatted by several threads into their own buffers and written in order
                const size_t per_line = 4 + (size_t)nw * (o.prnspec ? 5 : 1) + (o.prnprop ? 4 : 0) + (o.energy ? 3 * (size_t)nw : 0);
                const unsigned workers = format_threads((size_t)n * per_line);
                if (workers <= 1) {
                    for (long a0 = 0; a0 < n; a0 += 4096) {
                        format_lines(c, a0, a0 + 4096 < n ? a0 + 4096 : n, out);
                        out.flush();
                    }
                } else {
                    std::vector<Out> parts(workers);
                    std::vector<std::thread> pool;
                    for (unsigned t = 0; t < workers; ++t)
                        pool.emplace_back([&, t] { format_lines(c, n * (long)t / workers, n * (long)(t + 1) / workers, parts[t]); });
                    for (auto &th : pool) th.join();
                    for (auto &part : parts) part.flush();
                }
                std::fflush(stdout);
            }
            t_write += since(tf);
            gort_pipe_release(dv.pipe);
            ++k;
        }
    });

    // ---- producer (this thread): reads and parses chunk after chunk into the pipes' pinned slots (gortt.c:232-237) ----
    bool bad_line = false, eof = false;
    LineReader reader;
    std::vector<size_t> offsets;
    std::string producer_error;
    long k_chunk = 0;
    while (!eof && !bad_line && producer_error.empty()) {
        Dev &dv = devs[(size_t)(k_chunk % (long)devs.size())];
        if (dv.id >= 0) gort_set_device(dv.id);
        double *ang = nullptr;
        const auto ta = std::chrono::steady_clock::now();
        if (gort_pipe_acquire(dv.pipe, &ang) != GORT_OK) { producer_error = gort_last_error(); break; }
        t_acquire += since(ta);
        const auto tr = std::chrono::steady_clock::now();
        long n = 0;
        if (o.binary_in) {
            const size_t got = std::fread(ang, sizeof(double), (size_t)CHUNK * 4, stdin);
            if (got % 4 != 0) bad_line = true;           // truncated record
            n = (long)(got / 4);
            if (got < (size_t)CHUNK * 4) eof = true;
        } else {
            // the chunk's lines are collected NUL-terminated (strtod must not run on into the next line) and
            // parsed by several threads; the first line that does not hold four numbers ends the input there
            const long nl = reader.take(stdin, CHUNK, offsets);
            if (nl < CHUNK) eof = true;
            const char *text = reader.buf.data();
            const unsigned workers = format_threads((size_t)nl * 64);            // ~4 strtod calls per line
            std::vector<long> first_bad(workers, nl);
            auto parse_lines = [&](unsigned t) {
                const long a0 = nl * (long)t / workers, a1 = nl * (long)(t + 1) / workers;
                for (long a = a0; a < a1; ++a)
                    if (!parse_angles(text + offsets[(size_t)a], &ang[(size_t)a * 4])) { first_bad[t] = a; break; }
            };
            if (workers <= 1) {
                parse_lines(0);
            } else {
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < workers; ++t) pool.emplace_back(parse_lines, t);
                for (auto &th : pool) th.join();
            }
            long good = nl;
            for (unsigned t = 0; t < workers; ++t) good = first_bad[t] < good ? first_bad[t] : good;
            if (good < nl) bad_line = true;
            n = good;
        }
        t_read += since(tr);
        const auto ts = std::chrono::steady_clock::now();
        if (gort_pipe_submit(dv.pipe, n) != GORT_OK) producer_error = gort_last_error();
        t_submit += since(ts);
        na += n;
        {
            std::lock_guard<std::mutex> lk(mu);
            ++chunks_submitted;
        }
        cv.notify_all();
        ++k_chunk;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        producer_done = true;
    }
    cv.notify_all();
    consumer.join();
    std::fflush(stdout);
    if (verbose)
        std::fprintf(stderr, "gortt: %ld lines in %ld chunks of <= %ld; setup %.3f s, total %.3f s; producer: slot wait %.3f, "
                     "read+parse %.3f, submit %.3f; consumer: chunk wait %.3f, format+write %.3f\n", na, k_chunk, CHUNK,
                     t_setup, since(t_start), t_acquire, t_read, t_submit, t_wait, t_write);
    // Every chunk has been collected and written: nothing is in flight on any device.  The process ends the orderly way
    // - pipes and engines destroyed, exit handlers and static destructors run - so that whatever finalises at exit
    // (rocprofv3's tool library, gcov, a leak checker) gets its turn.  GORTT_FAST_EXIT=1 skips the teardown (~0.03 s:
    // ~150 MB of pinned memory, device buffers, the HIP runtime's state - all reclaimed by the driver anyway) and leaves
    // through _Exit once the streams are flushed; it is ignored when a preloaded or profiling library is in the process.
    const bool fast_exit = std::getenv("GORTT_FAST_EXIT") && std::atoi(std::getenv("GORTT_FAST_EXIT")) != 0 &&
                           !std::getenv("LD_PRELOAD") && !std::getenv("ROCP_TOOL_LIBRARIES") && !std::getenv("HSA_TOOLS_LIB") &&
                           !std::getenv("ASAN_OPTIONS") && !std::getenv("LSAN_OPTIONS");
    const bool orderly = !fast_exit;
    const auto t_down = std::chrono::steady_clock::now();
    if (orderly)
        for (Dev &dv : devs) {
            if (dv.id >= 0) gort_set_device(dv.id);
            gort_pipe_destroy(dv.pipe);
            gort_engine_destroy(dv.eng);
        }
    if (verbose)
        std::fprintf(stderr, "gortt: %.3f s from main() to the first chunk's setup (HIP start-up, gap tables, header, spectra), "
                     "%.3f s to free pipes and engines%s\n", t_before, since(t_down), orderly ? "" : " (skipped: GORTT_FAST_EXIT)");
    auto leave = [&](int rc) -> int {
        std::fflush(stdout);
        std::fflush(stderr);
        if (!orderly) std::_Exit(rc);
        return rc;
    };
    if (!producer_error.empty()) { std::fflush(stdout); std::fprintf(stderr, "%s: %s\n", g_prog, producer_error.c_str()); return leave(EXIT_FAILURE); }
    if (!consumer_error.empty()) { std::fflush(stdout); std::fprintf(stderr, "%s: %s\n", g_prog, consumer_error.c_str()); return leave(EXIT_FAILURE); }
    if (bad_line) {
        std::fflush(stdout);
        std::fprintf(stderr, "%s: error on input, line %ld\n", argv[0], na + 1);
        return leave(EXIT_FAILURE);
    }
    if (na_check != na) {
        std::fflush(stdout);
        std::fprintf(stderr, "%s: expected number of angles (%d) does not match with number found (%ld)\n", argv[0],
                     na_check, na);
        return leave(EXIT_FAILURE);
    }
    return leave(EXIT_SUCCESS);
}
