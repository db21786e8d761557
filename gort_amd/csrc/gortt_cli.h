// gortt_cli.h -- the GPU-free front end of the `gortt` drop-in: the command line (prefix rules and their order as in
// gortt_cl_parser, gortt.c:1003-1136; usage text gortt.c:1140-1234), the header line (gortt.c:153-184), the angle-line
// reader with its scanf("%lf") emulation (gortt.c:232-237) and the row formatter glue (gortt.c:310-327).  Everything
// here runs on the host and touches no device entry point, so that it can be built with AddressSanitizer + UBSan and fed
// the hostile command lines and oddly spelled numbers of tests/golden/ on the CPU (tests/test_sanitizers.py,
// tests/support/cli_dry_run.cpp); gortt_main.cpp adds the device pipeline.
#ifndef GORTT_CLI_H
#define GORTT_CLI_H

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <string>
#include <strings.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "gort_amd.h"

namespace gortt_cli {

[[noreturn]] inline void die(const char *fmt, const char *a = nullptr, const char *b = nullptr)
{
    std::fflush(stdout);
    std::fprintf(stderr, fmt, a, b);
    std::exit(EXIT_FAILURE);
}

inline void usage(const char *bin)
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
inline void parse_args(int argc, char **argv, Options &o)
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

inline bool read_line(FILE *fp, std::string &line)
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
inline bool plain_field(const char *&p)
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

inline bool parse_angles(const char *s, double v[4])
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
inline std::vector<std::string> tokens(const std::string &line)
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
        if (len) std::fwrite(buf.data(), 1, len, stdout);       // (an empty chunk has no buffer yet: fwrite's pointer must not be null)
        len = 0;
    }
};

// worker threads for formatting `values` numbers: one per ~16k values, at most the cores we may run on
inline unsigned format_threads(size_t values)
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


// the header line `N M W_1 .. W_M` (gortt.c:153-184): messages and exits as the reference's main()
struct Header {
    std::string text;
    int na_check = 0, nw_check = 0;
    std::vector<double> wl;
};

// returns false where the reference returns EXIT_FAILURE from main() after its message (wavelength count mismatch)
inline bool read_header(FILE *fp, const char *prog, Header &h)
{
    if (!read_line(fp, h.text)) die("%s: error reading data on stdin\n", prog);
    std::vector<std::string> tk = tokens(h.text);
    if (tk.empty()) die("%s: error reading number of angles from line 1\n", prog);
    h.na_check = atoi(tk[0].c_str());
    if (tk.size() < 2) die("%s: error reading number of wavebands from line 1\n", prog);
    h.nw_check = atoi(tk[1].c_str());
    for (size_t i = 2; i < tk.size(); ++i) h.wl.push_back(atof(tk[i].c_str()));
    if (h.nw_check != (int)h.wl.size()) {
        std::fprintf(stderr, "%s: expected number of wavelengths (%d) does not match with number found (%d)\n", prog, h.nw_check,
                     (int)h.wl.size());
        return false;
    }
    return true;
}

// one chunk of angle lines from the reader into ang[4 * n]: lines parsed by several threads; returns the number of good
// lines, *bad = true if the line behind them does not hold four numbers, *eof when the input ended inside the chunk
inline long parse_chunk(LineReader &reader, FILE *fp, long max_lines, std::vector<size_t> &offsets, double *ang, bool *bad, bool *eof)
{
    const long nl = reader.take(fp, max_lines, offsets);
    if (nl < max_lines) *eof = true;
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
    if (good < nl) *bad = true;
    return good;
}

}  // namespace gortt_cli
#endif
