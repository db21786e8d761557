/* sanitize_driver.c -- exercises the CPU-side code under AddressSanitizer + UBSan
 * (tests/test_sanitizers.py builds it twice: against oracle/gort_oracle.c and against the product's
 * host translation unit gort_amd/csrc/gort_host.cpp).  GPU AddressSanitizer is not available on this
 * pool, so the device kernels are covered by index-logic emulation and bitwise tests instead. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef DRIVE_ORACLE
#include "gort_oracle.h"

int main(void)
{
    static gort_o_canopy c;
    float geo[4][4] = {{2.0f, 1.0f, 0.5f, 4.0f}, {2.0f, 2.0f, 0.6f, 3.3f}, {0.5f, 0.4f, 0.95f, 9.0f}, {4.0f, 4.0f, 0.05f, 0.1f}};
    double wl[7] = {400.0, 400.25, 555.5, 1650.75, 2499.99, 2500.0, 703.1};
    double rs[7], rl[7], tl[7], RT[2 * GORT_O_NBANDS], rsurf[3 * 7], scomp[3 * 4 * 7], K[3 * 4], en[3 * 7 * 3];
    double rsl[4] = {0.2, 0.1, 0.03726, -0.002426};
    double ang[3 * 4] = {10, 0, 30, 20, -55, 40, 62, 300, 90, 0, 90, 0};
    double chk = 0.0;
    for (int g = 0; g < 4; ++g) {
        gort_o_canopy_defaults(&c);
        if (g) gort_o_canopy_newstyle(&c, geo[g][0], geo[g][1], geo[g][2]);
        gort_o_canopy_set_lai(&c, geo[g][3]);
        gort_o_canopy_init(&c);
        if (g == 3) gort_o_gap_probabilities_q08(&c);
        else if (gort_o_gap_probabilities(&c) != 0) { fprintf(stderr, "histogram overflow\n"); return 2; }
        if (gort_o_price_soil(wl, 7, rsl, rs) != 0) return 3;
        gort_o_prospect_d(1.2 + g, 30., 10., 1.0, 0.0, 0.015, 0.009, RT);
        if (gort_o_leaf_interp(wl, 7, RT, rl, tl) != 0) return 4;
        gort_o_rsurf_stream(&c, ang, 3, 7, rs, rl, tl, rsurf, scomp, K);
        gort_o_energy_stream(&c, ang, 3, 7, rs, rl, tl, en);
        for (int i = 0; i < 7; ++i) if (en[i * 3] == en[i * 3]) chk += en[i * 3] + rsurf[i] + K[0];
    }
    printf("oracle sanitize driver ok %.6f\n", chk);
    return 0;
}
#else
#include "gort_amd.h"

int main(void)
{
    gort_canopy c;
    gort_leaf_soil s;
    double wl[7] = {400.0, 400.25, 555.5, 1650.75, 2499.99, 2500.0, 703.1};
    double rs[7], rl[7], tl[7], x[32], w[32];
    static char buf[1 << 15];
    char num[400];
    double chk = 0.0;
    gort_canopy_defaults(&c);
    gort_canopy_newstyle(&c, 2.0f, 2.0f, 0.6f);
    gort_canopy_set_lai(&c, 3.3f);
    if (gort_canopy_init(&c) != 0) return 2;
    gort_leaf_soil_defaults(&s);
    if (gort_spectra(&s, wl, 7, rs, rl, tl) != 0) return 3;
    wl[0] = 399.0;
    if (gort_spectra(&s, wl, 7, rs, rl, tl) != GORT_ERANGE) return 4;
    gort_gauleg(-1., 1., x, w, 32);
    for (int t = 0; t < GORT_NTH; ++t) { c.p_n0[t] = exp(-0.7 * t); c.epgap[t] = 1e-40 * t; }
    c.k_open = 0.1; c.k_openep = 0.03;
    long n = gort_lut_format(&c, buf, sizeof buf);
    if (n <= 0) return 5;
    if (gort_lut_format(&c, buf, 100) >= 0) return 6;                 /* too small: must fail, not overflow */
    FILE *fp = fopen("/tmp/gort_sanitize_lut.dat", "w");
    fwrite(buf, 1, (size_t)n, fp); fputs("120 0.5 0.5\n-3 0.25 0.125\n", fp); fclose(fp);   /* out-of-range row index */
    if (gort_lut_read("/tmp/gort_sanitize_lut.dat", &c) != 0) return 7;
    if (gort_lut_read("/nonexistent", &c) != GORT_EIO) return 8;
    double vals[] = {0.0, -0.0, 1e-9, -1e-9, 0.0078125, 123456.789, 3.9999999e9, 4.0e9, 1e300, -1e300, INFINITY, -INFINITY, NAN};
    for (unsigned i = 0; i < sizeof vals / sizeof *vals; ++i) chk += gort_format_f6(vals[i], num);
    for (int i = 0; i < 7; ++i) chk += rs[i] + rl[i] + tl[i];
    printf("host sanitize driver ok %.6f %s\n", chk + x[0] + w[31] + c.k_open, gort_version());
    return 0;
}
#endif
