// The front end of the `gortt` drop-in without a device (gort_amd/csrc/gortt_cli.h + the host precompute of
// gort_host.cpp): command line, -P file, header line, spectra, angle lines - and, where the device would evaluate a
// line, only the echo of its four angles ("%f %f %f %f " of the raw input, gortt.c:310) and the newline.  Messages and
// exit codes of everything in front of the device are those of the executable.  Built with AddressSanitizer + UBSan by
// tests/test_sanitizers.py and fed every reference-generated CLI case.  Test infrastructure.
#include "gortt_cli.h"

using namespace gortt_cli;

int main(int argc, char **argv)
{
    Options o;
    gort_canopy_defaults(&o.canopy);
    gort_leaf_soil_defaults(&o.leaf);
    parse_args(argc, argv, o);
    if (gort_canopy_init(&o.canopy) != GORT_OK) die("%s: %s\n", argv[0], gort_last_error());
    // what the gap-probability entry point refuses before it touches the device (a degenerate crown)
    if (!o.read_lut && gort_canopy_check_geometry(&o.canopy) != GORT_OK) die("%s: %s\n", argv[0], gort_last_error());
    if (o.write_lut) return EXIT_SUCCESS;                          // the table itself is the device's
    if (o.read_lut && gort_lut_read(o.lut_file.c_str(), &o.canopy) != GORT_OK)
        die("%s: error opening probability file: %s\n", argv[0], o.lut_file.c_str());
    Header hd;
    if (!read_header(stdin, argv[0], hd)) return EXIT_FAILURE;
    const int nw = (int)hd.wl.size();
    std::vector<double> rsoil(nw), rleaf(nw), tleaf(nw);
    if (nw > 0 && gort_spectra(&o.leaf, hd.wl.data(), nw, rsoil.data(), rleaf.data(), tleaf.data()) != GORT_OK)
        die("%s\n", gort_last_error());
    std::fputs(hd.text.c_str(), stdout);
    const long CHUNK = 1000;
    std::vector<double> ang((size_t)CHUNK * 4);
    LineReader reader;
    std::vector<size_t> offsets;
    bool bad_line = false, eof = false;
    long na = 0;
    Out out;
    while (!eof && !bad_line) {
        long n = 0;
        if (o.binary_in) {
            const size_t got = std::fread(ang.data(), sizeof(double), (size_t)CHUNK * 4, stdin);
            if (got % 4 != 0) bad_line = true;
            n = (long)(got / 4);
            if (got < (size_t)CHUNK * 4) eof = true;
        } else {
            n = parse_chunk(reader, stdin, CHUNK, offsets, ang.data(), &bad_line, &eof);
        }
        for (long a = 0; a < n; ++a) {
            for (int q = 0; q < 4; ++q) out.raw(ang[(size_t)4 * a + q]);
            out.text("\n", 1);
        }
        out.flush();
        na += n;
    }
    std::fflush(stdout);
    if (bad_line) {
        std::fprintf(stderr, "%s: error on input, line %ld\n", argv[0], na + 1);
        return EXIT_FAILURE;
    }
    if (hd.na_check != na) {
        std::fprintf(stderr, "%s: expected number of angles (%d) does not match with number found (%ld)\n", argv[0], hd.na_check, na);
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
