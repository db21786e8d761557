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
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <sys/uio.h>

#include "gortt_cli.h"

using namespace gortt_cli;

namespace {

const char *g_prog = "gortt";

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
    Header hd;
    if (!read_header(stdin, argv[0], hd)) return EXIT_FAILURE;
    const std::string &header = hd.text;
    const int na_check = hd.na_check;
    const std::vector<double> &wl = hd.wl;
    const int nw = (int)wl.size();

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
    // -energy: what gortt_energy computes depends on the line's sun direction only (gortt_albedo.c:62-138), so a chunk's albedo
    // rows come from the device in the INDEXED form - each distinct row once + one index per line (GORT_PIPE_ENERGY_INDEXED) -
    // and each is formatted once; GORTT_ENERGY_DENSE=1 keeps a row per line on the device, over PCIe and in the formatter
    const bool energy_dense = o.energy && std::getenv("GORTT_ENERGY_DENSE") && std::atoi(std::getenv("GORTT_ENERGY_DENSE")) != 0;
    // numbers a line leaves in the OUTPUT: in text every line carries its own copy of its albedo row's formatted bytes whether the row
    // came once per chunk or once per line (without the 3 nw term a chunk of 180 bands was 220 MB of text in the
    // formatting threads' buffers); only the binary output of the indexed form points at shared rows
    const size_t per_line_out = (size_t)nw * (1 + (o.prnspec ? 4 : 0) + (o.energy && (energy_dense || !o.binary_out) ? 3 : 0)) + 8;
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
    const unsigned pflags = (o.prnspec ? GORT_PIPE_SCOMP : 0u) | (o.energy ? GORT_PIPE_ENERGY : 0u) |
                            (o.energy && !energy_dense ? GORT_PIPE_ENERGY_INDEXED : 0u);
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
    // the text of a chunk's distinct albedo rows, each formatted once (indexed form)
    std::vector<Out> energy_text;
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
            if (o.energy && c.energy_index) {
                const Out &row = energy_text[c.energy_index[a]];
                dst.text(row.buf.data(), row.len);
            } else if (o.energy) {
                dst.nums(c.energy + (size_t)a * nw * 3, 3L * nw);
            }
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
            if (o.energy) add(c.energy + (size_t)(c.energy_index ? c.energy_index[a] : a) * nw * 3, (size_t)nw * 3);
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
                const bool indexed = o.energy && c.energy_index && nw > 0;
                if (indexed) {
                    // each distinct row once, by as many threads as the numbers are worth
                    const long nr = c.energy_rows;
                    if ((long)energy_text.size() < nr) energy_text.resize((size_t)nr);
                    auto rows_text = [&](long r0, long r1) {
                        for (long r = r0; r < r1; ++r) {
                            energy_text[(size_t)r].len = 0;
                            energy_text[(size_t)r].nums(c.energy + (size_t)r * nw * 3, 3L * nw);
                        }
                    };
                    const unsigned rw = format_threads((size_t)nr * 3 * (size_t)nw);
                    if (rw <= 1) {
                        rows_text(0, nr);
                    } else {
                        std::vector<std::thread> pool;
                        for (unsigned t = 0; t < rw; ++t) pool.emplace_back(rows_text, nr * (long)t / rw, nr * (long)(t + 1) / rw);
                        for (auto &th : pool) th.join();
                    }
                }
                // copying a formatted row costs about a tenth of formatting it
                const size_t per_line = 4 + (size_t)nw * (o.prnspec ? 5 : 1) + (o.prnprop ? 4 : 0) +
                                        (o.energy ? (indexed ? 3 * (size_t)nw / 8 : 3 * (size_t)nw) : 0);
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
            n = parse_chunk(reader, stdin, CHUNK, offsets, ang, &bad_line, &eof);
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
