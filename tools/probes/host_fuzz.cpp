// host_fuzz.cpp -- the host-side entry points of libgort_amd (gort_host.cpp: no device code) under AddressSanitizer and
// UBSan with hostile inputs: the exact "%f" formatter over random bit patterns against snprintf, the LUT readers and
// the gap-table cache on damaged files, wavelength ranges, tiny output buffers.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -Igort_amd/csrc \
//       -DGORT_DATA_DIR=\"$PWD/gort_amd/data\" gort_amd/csrc/gort_host.cpp tools/probes/host_fuzz.cpp -o /tmp/host_fuzz && /tmp/host_fuzz
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <unistd.h>

#include "gort_amd.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

int main(int argc, char **argv)
{
    const double scale = argc > 1 ? atof(argv[1]) : 1.0;     // share of the default number of trials
    std::mt19937_64 rng(12345);
    // 1) formatter: random bit patterns, random magnitudes, ties, against snprintf (NaN prints as -nan)
    {
        char a[400], b[400];
        auto one = [&](double v) {
            const int n = gort_format_f6(v, a);
            a[n] = 0;
            if (std::isnan(v)) std::strcpy(b, "-nan");
            else std::snprintf(b, sizeof b, "%f", v);
            if (std::strcmp(a, b) != 0) { std::printf("format mismatch %a: %s vs %s\n", v, a, b); ++failures; }
        };
        for (int i = 0; i < (int)(2000000 * scale); ++i) {
            uint64_t bits = rng();
            double v;
            std::memcpy(&v, &bits, 8);
            one(v);
        }
        std::uniform_real_distribution<double> u(-12.0, 12.0);
        for (int i = 0; i < (int)(2000000 * scale); ++i) one(std::ldexp(u(rng), (int)(rng() % 40) - 20));
        for (long k = -3000000; k <= 3000000; k += (long)(7 / scale)) one((k + 0.5) * 1e-6);     // decimal ties (not representable: round by the true value)
        for (int e = -1080; e <= 1030; ++e) { one(std::ldexp(1.0, e)); one(-std::ldexp(1.5, e)); }
        one(0.0); one(-0.0); one(INFINITY); one(-INFINITY); one(4.0e9); one(3.9999999999e9); one(1e22); one(1.7976931348623157e308);
    }
    // 2) row formatter: exact capacity boundaries
    {
        std::vector<double> v(257);
        for (size_t i = 0; i < v.size(); ++i) v[i] = (i % 7 == 0) ? -123456789.987654 : (i % 5 == 0 ? NAN : 1.0 / (double)(i + 1));
        for (size_t cap : {size_t(0), size_t(1), size_t(17), size_t(359), size_t(360), size_t(24 * 257 - 1), size_t(24 * 257), size_t(1 << 16)}) {
            std::vector<char> buf(cap + 1, 'x');
            const long n = gort_format_f6_row(v.data(), (long)v.size(), buf.data(), cap);
            CHECK(n < 0 || (size_t)n <= cap);
            CHECK(buf[cap] == 'x');
        }
        v[3] = 1e300;                                       // 301 digits: the slow path
        std::vector<char> buf(1 << 16);
        CHECK(gort_format_f6_row(v.data(), (long)v.size(), buf.data(), buf.size()) > 300);
        CHECK(gort_format_f6_row(v.data(), (long)v.size(), buf.data(), 500) < 0);
        CHECK(gort_format_f6_row(nullptr, 1, buf.data(), 10) < 0);
    }
    // 3) LUT reader and cache loader on damaged files
    {
        char dir[] = "/tmp/gort_fuzz_XXXXXX";
        CHECK(mkdtemp(dir) != nullptr);
        gort_canopy c;
        gort_canopy_defaults(&c);
        gort_canopy_set_lai(&c, 4.0f);
        CHECK(gort_canopy_init(&c) == GORT_OK);
        for (int t = 0; t < GORT_NTH; ++t) { c.p_n0[t] = std::exp(-0.1 * t * t); c.epgap[t] = 0.5 * c.p_n0[t]; }
        c.k_open = 0.25; c.k_openep = 0.125;
        CHECK(gort_lut_cache_store(dir, &c) == GORT_OK);
        char name[64];
        std::snprintf(name, sizeof name, "/gap-%016llx.lut", (unsigned long long)gort_canopy_key(&c));
        const std::string path = std::string(dir) + name;
        std::string text;
        {
            FILE *f = std::fopen(path.c_str(), "rb");
            CHECK(f != nullptr);
            char blk[4096];
            size_t k;
            while ((k = std::fread(blk, 1, sizeof blk, f)) > 0) text.append(blk, k);
            std::fclose(f);
        }
        gort_canopy d = c;
        CHECK(gort_lut_cache_load(dir, &d) == GORT_OK && std::memcmp(d.p_n0, c.p_n0, sizeof c.p_n0) == 0);
        for (int trial = 0; trial < (int)(3000 * scale); ++trial) {
            std::string bad = text;
            const int kind = trial % 4;
            if (kind == 0) bad.resize(rng() % (bad.size() + 1));                                         // truncated
            else if (kind == 1) for (int k = 0; k < 8; ++k) bad[rng() % bad.size()] = (char)(rng() % 256);  // corrupted bytes
            else if (kind == 2) bad.insert(rng() % bad.size(), std::string(1 + rng() % 300, "0123456789-.e +xXpP\n#"[rng() % 21]));
            else { bad = std::to_string((long)(rng() % 400) - 100) + " " + std::string(rng() % 5000, '9') + " 1\n" + bad; }   // huge numbers, wild row indices
            FILE *f = std::fopen(path.c_str(), "wb");
            std::fwrite(bad.data(), 1, bad.size(), f);
            std::fclose(f);
            gort_canopy e = c;
            const int rc = gort_lut_cache_load(dir, &e);
            CHECK(rc == GORT_OK || rc == 1);
            if (rc == GORT_OK) CHECK(std::memcmp(e.p_n0, c.p_n0, sizeof(double) * 90) == 0);       // a hit must be the true entry
            gort_canopy g = c;
            CHECK(gort_lut_read(path.c_str(), &g) == GORT_OK);                                   // the -P reader takes what it can, in bounds
        }
        std::remove(path.c_str());
        CHECK(gort_lut_cache_load(dir, &d) == 1);
        CHECK(gort_lut_read(path.c_str(), &d) == GORT_EIO);
        char small[100];
        CHECK(gort_lut_format(&c, small, sizeof small) < 0);
        rmdir(dir);
    }
    // 4) spectra: wavelengths at and beyond the range, zero and many bands
    {
        gort_leaf_soil s;
        gort_leaf_soil_defaults(&s);
        std::vector<double> wl = {400.0, 2500.0, 400.0000001, 2499.9999999, 1234.5678};
        std::vector<double> rs(wl.size()), rl(wl.size()), tl(wl.size());
        CHECK(gort_spectra(&s, wl.data(), (int)wl.size(), rs.data(), rl.data(), tl.data()) == GORT_OK);
        for (double bad : {399.9999, 2500.0001, -1.0, 1e9, (double)NAN}) {
            wl[2] = bad;
            CHECK(gort_spectra(&s, wl.data(), (int)wl.size(), rs.data(), rl.data(), tl.data()) == GORT_ERANGE);
        }
        CHECK(gort_spectra(&s, wl.data(), 0, rs.data(), rl.data(), tl.data()) == GORT_OK);
    }
    std::printf(failures ? "host_fuzz: %d FAILURES\n" : "host_fuzz: ok\n", failures);
    return failures != 0;
}
