/*
 * gort_oracle.c -- TEST INFRASTRUCTURE: CPU restatement of the GORT hot path.
 * See gort_oracle.h for the rules (oracle only; never linked into the product).
 *
 * Written from the behaviour documented in SURVEY.md section 8(a) and the reference
 * sources cited per function (paths relative to /root/reference).  Structure is
 * ours (one flat canopy struct, each named quantity evaluated once, live products
 * only); the arithmetic of every formula keeps the reference's association so the
 * two agree to rounding.
 */
#include "gort_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef GORT_DATA_DIR
#error "compile with -DGORT_DATA_DIR=\"<repo>/gort_amd/data\""
#endif

/* coefficient tables: gort_amd/data/ (.f32, .f64) (tools/extract_spectral_tables.py) */
__asm__(".section .rodata\n"
        ".balign 16\n"
        ".global gort_o_prospect_tab\n"
        "gort_o_prospect_tab:\n"
        ".incbin \"" GORT_DATA_DIR "/prospect_d_coeffs.f32\"\n"
        ".balign 16\n"
        ".global gort_o_soil_tab\n"
        "gort_o_soil_tab:\n"
        ".incbin \"" GORT_DATA_DIR "/price_soil_eofs.f64\"\n"
        ".previous\n");
extern const float  gort_o_prospect_tab[7 * GORT_O_NBANDS];
extern const double gort_o_soil_tab[4 * 421];

#define PI M_PI
static double dtor(double x) { return x * PI / 180.0; }          /* gortt.h:6  */
static double sec(double x)  { return 1.0 / cos(x); }            /* gortt.h:8  */
static double dmax(double x, double y) { return x > y ? x : y; } /* gortt.h:9  */
static double dmin(double x, double y) { return x < y ? x : y; } /* gortt.h:10 */

/* ------------------------------------------------------------------ canopy */

void gort_o_canopy_defaults(gort_o_canopy *c)
{
    memset(c, 0, sizeof *c);
    c->lambda = 0.405;               /* gortt.c:67 */
    c->r = 0.76;                     /* gortt.c:68 */
    c->b = 3.55263 * c->r;           /* gortt.c:69 */
    c->h1 = 3.0;  c->h2 = 8.5;       /* gortt.c:70-71 */
    c->favd = 0.858;                 /* gortt.c:72 */
    c->dth = dtor(1);                /* gortt.c:76 */
    c->nlayers = GORT_O_NLAYERS;     /* gortt.c:78 */
    c->maxcrowns = GORT_O_MAXCROWNS; /* gortt.c:89 */
    c->nh_es = GORT_O_NH_ES;         /* gortt.c:90 */
}

/* -HB/-BR/-PCC are parsed into C floats (gortt.c:1014,1032-1034) */
void gort_o_canopy_newstyle(gort_o_canopy *c, float hb, float br, float pcc)
{
    c->r = 10.;
    c->b = br * c->r;
    c->h1 = c->b * 2.;
    c->h2 = hb * c->b + c->h1;
    c->lambda = pcc / (c->r * c->r * PI);
}

void gort_o_canopy_set_lai(gort_o_canopy *c, float lai)
{
    c->favd = lai * 3. / (c->lambda * c->r * c->r * PI * c->b * 4.0);
}

void gort_o_canopy_init(gort_o_canopy *c)
{
    int i;
    c->ell = c->b / c->r;
    c->rr = c->r * c->r;
    c->rrr = c->rr * c->r;
    c->h = 2.0 * c->r * c->ell + c->h2 - c->h1;
    c->k = 0.5;                                          /* LAD_05, gortt.c:85,622-623 */
    c->elai = c->favd * ((1.333333) * c->lambda * PI * c->ell * c->rrr);   /* sic, gortt.c:657 */
    c->tau = c->k * c->favd;
    c->z1 = c->h1 - c->r * c->ell;
    c->z2 = c->h2 + c->r * c->ell;
    c->lv = c->lambda / (c->h2 - c->h1);
    c->favd_p = c->favd * c->ell;
    c->tau_p = c->k * c->favd_p;
    c->lv_p = c->lv * c->ell;
    c->z1_p = c->z1 / c->ell;  c->z2_p = c->z2 / c->ell;
    c->h1_p = c->h1 / c->ell;  c->h2_p = c->h2 / c->ell;
    c->dz = (double)(c->z2 - c->z1) / ((double)c->nlayers - 1.0);
    c->ds = c->dz;                                       /* gortt.c:696 */
    c->dz_p = c->dz / c->ell;
    c->nth = (int)(dtor(90.0) / c->dth + 0.5) + 1;
    c->factorial[0] = 1;
    for (i = 1; i <= c->maxcrowns; i++) c->factorial[i] = c->factorial[i - 1] * (double)i;
    for (i = c->nlayers - 1; i >= 0; i--) {
        c->height[i] = c->z2 - c->dz * (double)(c->nlayers - 1 - i);
        c->height_p[i] = c->height[i] / c->ell;
    }
    for (i = 0; i < c->nth; i++) {
        c->theta[i] = c->dth * (double)i;
        if (c->theta[i] >= PI / 2.0) c->theta[i] = PI / 2.0 - 1.0 * PI / 180.0;
        c->theta_p[i] = atan(tan(c->theta[i]) * c->ell);
        if (c->theta_p[i] >= PI / 2.0) c->theta_p[i] = PI / 2.0 - 1.0 * PI / 180.0;
    }
}

/* ---------------------------------------------- Pn: crown projection volume */

/* gortt_pn_kopen.c:285-305 */
static double left_circle_area(double r, double x_cut)
{
    double area_tot = PI * r * r;
    double ang_sector = acos(fabs(x_cut) / r) * 2.0;
    double area_sector = area_tot * ang_sector / (2.0 * PI);
    double area_triangle = fabs(x_cut) * sqrt(r * r - x_cut * x_cut);
    return (x_cut > 0.0) ? area_tot - (area_sector - area_triangle)
                         : area_sector - area_triangle;
}

/* gortt_pn_kopen.c:309-323 */
static double right_ellipse_area(double r, double b, double x_cut)
{
    double x_cut_p = x_cut / (b / r);
    double a_p = PI * r * r;
    a_p -= left_circle_area(r, x_cut_p);
    return a_p * (b / r);
}

/* gortt_pn_kopen.c:170-229 (+ the circle/ellipse mix, :233-282) */
static double cross_section(const gort_o_canopy *c, double t, double h, double z)
{
    double h_low, h_high;
    if (z < h - c->r) return 0.0;
    h_low = h - c->r * sin(t);
    h_high = h + c->r * sin(t);
    if (z <= h_low) {
        double a = c->rr - (h - z) * (h - z);
        double r_p = (a <= 0) ? 0 : sqrt(a);
        return PI * r_p * r_p;
    } else if (z > h_low && z < h_high) {
        double zdiff = h - z;
        double r_p = sqrt(c->rr - zdiff * zdiff);
        double x_cc = zdiff * tan(t);
        double x_p = x_cc / (1.0 - cos(t) * cos(t));
        double a_cp = left_circle_area(r_p, x_p - x_cc);
        double a_ep = right_ellipse_area(c->r, c->r * sec(t), x_p);
        return a_cp + a_ep;
    }
    return PI * c->rr * sec(t);
}

double gort_o_crown_proj_volume(const gort_o_canopy *c, double t, double h)
{
    double vol = 0.0, z;
    /* loop variable accumulated in floating point, as in :162 */
    for (z = c->h1_p + c->dz_p / 2.0; z <= c->h2_p; z += c->dz_p)
        vol += cross_section(c, t, h, z) * (c->dz_p);
    return vol;
}

/* ------------------------------------------------------------ ES(z, theta) */

/* gortt_pn_kopen.c:566-645 */
static double mean_chord(const gort_o_canopy *c, int z, double h, int th)
{
    double hz = c->height_p[z];
    if (hz > h + c->r - 0.0001) return 0.0;
    if (hz < h - c->r + 0.0001) return 4.0 * c->r / 3.0;
    {
        double V_sphere = 4.0 * PI * c->rrr / 3.0;
        double zdiff = fabs(h - hz);
        double ht = c->r - zdiff;
        double V_slice = PI * ht * ht / 3.0 * (3.0 * c->r - ht);
        double V_tot = (hz > h) ? V_slice : V_sphere - V_slice;
        double proj_area;
        V_tot /= cos(c->theta_p[th]);
        if (h < hz) proj_area = cross_section(c, c->theta_p[th], h, (h - zdiff));
        else        proj_area = cross_section(c, c->theta_p[th], h, (h + zdiff));
        return V_tot / proj_area;
    }
}

double gort_o_get_es(const gort_o_canopy *c, int z, int t)
{
    double dh = (c->h2_p - c->h1_p) / (double)c->nh_es;
    double pcc = 1.0 / (c->h2_p - c->h1_p);            /* :648-659 */
    double ES = 0.0, h;
    for (h = c->h1_p + dh / 2.0; h <= c->h2_p; h += dh)
        ES += mean_chord(c, z, h, t) * (pcc * dh);
    return ES;
}

/* ------------------------------------------------ capsule volume below h_b */

/* gortt_pn_kopen.c:858-872 */
static double triang_fcn(double x, double b, double r, double the)
{
    double a1 = tan(the) * (x - b);
    double a2 = r * r - x * x;
    double a3 = a2 - a1 * a1;
    if (fabs(a3) < 0.0000000001) a3 = 0.0;
    return 2.0 * a1 * sqrt(a3);
}

/* composite Simpson, 20 double-intervals; gortt_pn_kopen.c:811-854 */
static double triang(double b, double r, double the, int noint)
{
    double sint = sin(the), cost = cos(the);
    double a1 = r * r - b * b * sint * sint;
    double x0 = b * (sint * sint) + sqrt(a1) * cost;
    int m = noint, i;
    double h = .50 * (x0 - b) / (float)m;
    double sum1 = 0.0, sum2 = 0.0, volume;
    for (i = 0; i < m; i++) sum1 += triang_fcn(b + (float)(2 * i + 1) * h, b, r, the);
    volume = 4.0 * sum1;
    for (i = 0; i < m - 1; i++) sum2 += triang_fcn(b + (float)(2 * (i + 1)) * h, b, r, the);
    volume += 2.0 * sum2;
    volume += triang_fcn(x0, b, r, the);
    volume += triang_fcn(b, b, r, the);
    volume *= h / 3.0;
    return volume;
}

/* gortt_pn_kopen.c:796-806 */
static double sector(double a1, double a2, double r)
{
    double b1 = r * r * a1 - (a1 * a1 * a1) / 3.0;
    double b2 = r * r * a2 - (a2 * a2 * a2) / 3.0;
    return PI * (b2 - b1) / 2.0;
}

/* gortt_pn_kopen.c:771-792 */
static double trisec(double hh, double hh_b, double th, double r)
{
    double tmp = hh - hh_b;
    double x = -1.0 * tmp * sin(th) + sqrt(r * r - tmp * tmp) * cos(th);
    double b = -tmp / sin(th);
    return triang(b, r, th, 20) + sector(x, r, r);
}

/* gortt_pn_kopen.c:876-886 */
static double cylind_fcn(double x, double r)
{
    return .50 * x * sqrt(r * r - x * x) + .50 * r * r * asin(x / r);
}

/* gortt_pn_kopen.c:891-924 */
static double cylind(double r, double h1, double h2, double h)
{
    double slope = h / (h2 - h1);
    double tmp1 = sqrt(r * r - h1 * h1);
    double tmp2 = sqrt(r * r - h2 * h2);
    double volume = tmp1 * tmp1 * tmp1 - tmp2 * tmp2 * tmp2;
    volume /= 3.0;
    volume -= h1 * (cylind_fcn(h2, r) - cylind_fcn(h1, r));
    volume *= 2.0 * slope;
    if (h2 < r) {
        double phi = acos(h2 / r);
        double s1 = r * r * phi;
        double s2 = r * sin(phi) * h2;
        volume += (s1 - s2) * h;
    }
    return volume;
}

/* gortt_pn_kopen.c:665-768: seven-way piecewise in where plane h_b cuts the capsule */
double gort_o_vol(const gort_o_canopy *c, int h, int h_s, int t, double h_b)
{
    const double r = c->r, th = c->theta_p[t];
    const double zh = c->height_p[h], zs = c->height_p[h_s];
    double V, V_0, V_sp1, V_sp2, V_cyln, tmp_s, h_t, h_tt;

    tmp_s = (zs - zh) / cos(th);
    V_0 = PI * c->rr * tmp_s;
    V_0 += (4.0 / 3.0) * PI * c->rrr;

    if ((zh - r) >= h_b) {
        V = 0.0;
    } else if ((zh - r * sin(th)) >= h_b) {
        h_t = r - (zh - h_b);
        V = (PI / 3.0) * h_t * h_t * (3.0 * r - h_t);
    } else if ((zh + r * sin(th)) >= h_b) {
        V_sp1 = (2.0 / 3.0) * PI * c->rrr;
        V_sp1 -= trisec(zh, h_b, th, r);
        h_tt = (h_b - (zh - r * sin(th))) / cos(th);
        if (zs - r * sin(th) >= h_b) {
            double hh1 = (zh - h_b) / sin(th);
            V_cyln = cylind(r, hh1, r, h_tt);
            V_sp2 = 0.0;
        } else {
            double hh1 = (zh - h_b) / sin(th);
            double hh2 = (zs - h_b) / sin(th);
            double hh = (zs - zh) / cos(th);
            V_cyln = cylind(r, hh1, hh2, hh);
            V_sp2 = trisec(h_b, zs, th, r);
        }
        V = V_sp1 + V_cyln + V_sp2;
    } else if (zs - r * sin(th) >= h_b) {
        double tmp_h = (h_b - zh) / cos(th);
        V_cyln = PI * r * r * tmp_h;
        V_sp1 = (2.0 / 3.0) * PI * c->rrr;
        V = V_sp1 + V_cyln;
    } else if (zs + r * sin(th) >= h_b) {
        double hh1, tmp_h;
        h_tt = (zs + r * sin(th) - h_b) / cos(th);
        hh1 = (h_b - zs) / sin(th);
        tmp_h = (zs - zh) / cos(th);
        V_cyln = PI * r * r * tmp_h - cylind(r, hh1, r, h_tt);
        V_sp2 = trisec(h_b, zs, th, r);
        V_sp1 = (2.0 / 3.0) * PI * c->rrr;
        V = V_cyln + V_sp2 + V_sp1;
    } else if (zs + r >= h_b) {
        h_t = r - (h_b - zs);
        V_sp1 = (PI / 3.0) * h_t * h_t * (3.0 * r - h_t);
        V = V_0 - V_sp1;
    } else {
        V = V_0;
    }
    return V;
}

/* ---------------------------------------------- gap probabilities (live set) */

static int s_to_index(const gort_o_canopy *c, double s) { return (int)(s / c->ds + 0.5); }   /* :134-139 */

/*
 * Live products only: p_n0[h][t] for all h, p_s0, the path-length
 * histogram pd_s for h=0 only, epgap[0][t<nth-1], k_open[0], k_openep[0].  The
 * reference additionally fills tables nothing reads (vb, fb, t_open, pd_s for h>0).
 * Returns 0, or -1 when a path-length index falls outside the histogram (the
 * reference would write out of bounds there).
 */
int gort_o_gap_probabilities(gort_o_canopy *c)
{
    int h, t, n, sp_i, s, rc = 0;

    /* :24-32 */
    for (t = 0; t < c->nth; t++)
        for (h = 0; h < c->nlayers; h++) {
            c->v_g[h][t] = gort_o_crown_proj_volume(c, c->theta_p[t], c->height_p[h]);
            c->p_n0[h][t] = exp(-1.0 * c->lv_p * c->v_g[h][t]);
        }
    /* :40-45 */
    for (t = 0; t < c->nth; t++) {
        c->p_s0[c->nlayers - 1][t] = 0.0;
        for (h = c->nlayers - 2; h >= 0; h--) c->p_s0[h][t] = c->p_n0[h + 1][t] - c->p_n0[h][t];
    }

    for (t = 0; t < c->nth; t++) c->epgap0[t] = 0.0;

    /* :50-71 with h=0, then :1083-1125 */
    for (t = 0; t < c->nth; t++) {
        int s_max = s_to_index(c, (c->z2_p - c->height_p[0]) / cos(c->theta_p[t]));
        int nbin = s_max + 3;                                  /* PD_S_BUFF, gortt.h:4 */
        double *pd = (double *)calloc((size_t)nbin, sizeof(double));
        double es = gort_o_get_es(c, 0, t);                    /* :445 */
        c->es0[t] = es;

        for (sp_i = c->nlayers - 1; sp_i > 0; sp_i--) {        /* :457 */
            double s_p = (double)(c->height_p[sp_i] - c->height_p[0]) / cos(c->theta_p[t]);
            double P_s_p, temp1;
            if (sp_i == c->nlayers - 1) { pd[0] += c->p_s0[sp_i][t]; continue; }
            P_s_p = c->p_s0[sp_i][t];
            /* n-invariant; the reference recomputes it for every n (:496-497) */
            temp1 = gort_o_vol(c, 0, sp_i, t, c->h2_p) - gort_o_vol(c, 0, sp_i, t, c->h1_p);
            temp1 *= c->lv_p;
            for (n = 1; n <= c->maxcrowns; n++) {
                double P_n = (pow(temp1, (double)n) * exp(-temp1)) /
                             (c->factorial[n] * (1.0 - exp(-temp1)));
                double sl = s_p * (1.0 - exp(-1.0 * (double)n * es / s_p));
                int idx = s_to_index(c, sl);
                if (idx < 0 || idx >= nbin) { rc = -1; continue; }
                pd[idx] += P_n * P_s_p;
            }
        }
        if (t < c->nth - 1) {                                   /* :1099 */
            double e = 0.0;
            for (s = 0; s <= s_max; s++)
                e += exp(-((double)s * c->ds) * c->tau_p) * pd[s];   /* :1113,:1138,:145 */
            c->epgap0[t] = e;
        }
        free(pd);
    }

    /* :329-391, h=0 */
    {
        double ko = 0.0, kep = 0.0;
        double l1 = c->p_n0[0][0] * sin(2.0 * c->theta[0]);
        double l2 = c->epgap0[0] * sin(2.0 * c->theta[0]);
        for (t = 1; t < c->nth; t++) {
            double t1 = c->p_n0[0][t] * sin(2.0 * c->theta[t]);
            double t2 = c->epgap0[t] * sin(2.0 * c->theta[t]);
            ko += (t1 + l1) / 2.0 * c->dth;   l1 = t1;
            kep += (t2 + l2) / 2.0 * c->dth;  l2 = t2;
        }
        c->k_open0 = ko;  c->k_openep0 = kep;
    }
    return rc;
}

/* gortt_pn_kopen.c:1144-1200 */
void gort_o_gap_probabilities_q08(gort_o_canopy *c)
{
    int t;
    double cc = PI * c->rr * c->lambda;
    double l = c->favd * c->b * 4. / 3. * cc;
    double k2 = 0.348535 * pow(cc, (-1.08069 - 0.0874595 * cc));
    double k1 = 0.0014166;
    double a = cc * (exp(k1 * cc * cc) - exp(-k2 * l));
    double ko = 0.0, kep = 0.0, l1, l2;

    memset(c->p_n0, 0, sizeof c->p_n0);
    memset(c->p_s0, 0, sizeof c->p_s0);
    /* the "last" terms are formed from the still-zero tables (:1176-1177) */
    l1 = 0.0 * sin(2.0 * c->theta[0]);
    l2 = 0.0 * sin(2.0 * c->theta[0]);
    c->p_n0[0][0] = exp(-cc / (cos(c->theta_p[0])));
    c->epgap0[0] = exp(-a / (cos(c->theta_p[0]))) - c->p_n0[0][0];
    for (t = 1; t < c->nth; t++) {
        double t1, t2;
        c->p_n0[0][t] = exp(-cc / (cos(c->theta_p[t])));
        c->epgap0[t] = exp(-a / (cos(c->theta_p[t]))) - c->p_n0[0][t];
        t1 = c->p_n0[0][t] * sin(2.0 * c->theta[t]);
        ko += (t1 + l1) / 2.0 * c->dth;   l1 = t1;
        t2 = c->epgap0[t] * sin(2.0 * c->theta[t]);
        kep += (t2 + l2) / 2.0 * c->dth;  l2 = t2;
    }
    c->k_open0 = ko;  c->k_openep0 = kep;
}

/* ---------------------------------------------------------------- geometry */

static double prime_theta(const gort_o_canopy *c, double za)      /* gortt.c:581-588 */
{
    return atan((c->b / c->r) * tan(za));
}

/* table lookup; the reference indexes past the table for zenith > 90 deg (undefined);
 * the oracle defines that as NaN. gortt.c:889-897 */
static void interp_tables(const gort_o_canopy *c, double za, double *pn0, double *epg)
{
    double pos = fabs(za) / c->dth;
    int ci = (int)ceil(pos), fi = (int)floor(pos);
    double d = pos - fi;
    if (ci >= c->nth || !(pos == pos)) { *pn0 = NAN; *epg = NAN; return; }
    *pn0 = d * c->p_n0[0][ci] + (1.0 - d) * c->p_n0[0][fi];
    *epg = d * c->epgap0[ci] + (1.0 - d) * c->epgap0[fi];
}

void gort_o_set_zenith_probabilities(const gort_o_canopy *c, gort_o_geom *g)
{
    interp_tables(c, g->sza, &g->pn0_s, &g->epgap_s);
    interp_tables(c, g->vza, &g->pn0_v, &g->epgap_v);
}

void gort_o_normalise_angles(const gort_o_canopy *c, double vza_deg, double vaa_deg,
                             double sza_deg, double saa_deg, gort_o_geom *g)
{
    memset(g, 0, sizeof *g);
    g->vza = dtor(vza_deg);  g->vaa = dtor(vaa_deg);
    g->sza = dtor(sza_deg);  g->saa = dtor(saa_deg);
    if (g->sza < 0.0) { g->saa += PI; g->sza *= -1.0; }
    if (g->vza < 0.0) { g->vaa += PI; g->vza *= -1.0; }
    while (g->saa > 2 * PI) g->saa -= 2 * PI;
    while (g->vaa > 2 * PI) g->vaa -= 2 * PI;
    while (g->saa < 0) g->saa += 2 * PI;
    while (g->vaa < 0) g->vaa += 2 * PI;
    g->raa = g->saa - g->vaa;
    g->raa = fabs((g->raa - 2 * PI * (int)(0.5 + g->raa * M_1_PI * 0.5)));   /* C truncation */
    g->vza_p = prime_theta(c, g->vza);
    g->sza_p = prime_theta(c, g->sza);
    g->fd = c->use_user_fd ? c->fd_user : cos(g->sza) / (cos(g->sza) + 0.09);
    gort_o_set_zenith_probabilities(c, g);
}

/* -------------------------------------------------- areal proportions (K's) */

/* gortt_brdf.c:23-100 (ambrals t2, Li&Strahler'92 t1 - the compiled-in branches) */
static double overlap_fn(const gort_o_canopy *c, double sza_p, double vza_p, double raa)
{
    double d = pow(tan(sza_p), 2) + pow(tan(vza_p), 2)
             - 2.0 * tan(sza_p) * tan(vza_p) * cos(raa);
    double D = sqrt(dmax(0.0, d));
    double t2 = sqrt(D * D + pow((tan(sza_p) * tan(vza_p) * sin(raa)), 2));
    double t1 = (sec(sza_p) + sec(vza_p));
    double cos_t = (c->h / c->b) * t2 / t1;
    double t;
    cos_t = dmax(-1.0, cos_t);
    cos_t = dmin(1.0, cos_t);
    t = acos(cos_t);
    return dmax(0.0, (t - sin(t) * cos_t) * (sec(sza_p) + sec(vza_p)) / PI);
}

/* gortt_brdf.c:7-20 */
static double kg_fn(const gort_o_canopy *c, double sza_p, double vza_p, double raa)
{
    double ov = overlap_fn(c, sza_p, vza_p, raa);
    return exp(-(c->lambda * pow(c->r, 2) * PI * (sec(sza_p) + sec(vza_p) - ov)));
}

/* gortt_brdf.c:171-238 */
static void kc_fFbeta(const gort_o_canopy *c, const gort_o_geom *g, double raa, double Kg,
                      double *f, double *F, double *beta)
{
    const double sp = g->sza_p, vp = g->vza_p;
    double ov = overlap_fn(c, sp, vp, raa);
    double phase_prime = cos(vp) * cos(sp) + sin(vp) * sin(sp) * cos(raa);
    double Mi = (1.0 - (1.0 - exp(-c->lambda * PI * c->rr * sec(sp))) / (c->lambda * PI * c->rr * sec(sp)));
    double Mv = (1.0 - (1.0 - exp(-c->lambda * PI * c->rr * sec(vp))) / (c->lambda * PI * c->rr * sec(vp)));
    double Gamma   = PI * c->rr * (sec(sp) + sec(vp) - ov);
    double Gamma_c = PI * c->rr * sec(vp) * 0.5 * (1.0 + phase_prime);
    double Gamma_v = PI * c->rr * sec(vp);
    double M, theta_Mi, theta_Mv, Gamma_i, PiMi, PvMv, Po;
    (void)theta_Mv;

    *F = Gamma_c / Gamma;
    M = 1.0 - (1.0 - Kg) / (c->lambda * Gamma);
    theta_Mi = acos(1.0 - 2.0 * Mi);
    theta_Mv = acos(1.0 - 2.0 * Mv);
    Gamma_i = Gamma_v;
    PiMi = (1 - cos(theta_Mi * (1 - (sp - vp * cos(raa)) / PI))) / 2.0;
    PvMv = Mv - (1.0 - cos(vp * cos(raa) - sp)) / 2.0;

    if ((raa < dtor(270.)) && (raa > dtor(90.))) Po = PvMv;
    else if (fabs(g->vza) > fabs(g->sza)) Po = PiMi;
    else Po = PvMv;

    if (sp < 0.000000001) {
        *beta = 0.0;
    } else {
        double D = c->r * (1.0 / tan(sp / 2.0));
        *beta = (c->lambda * Gamma_i) / (c->lambda * Gamma_i + (c->h2 - c->h1) / D)
              * (1.0 - exp(-c->lambda * Gamma_i - (c->h2 - c->h1) / D))
              / (1.0 - exp(-c->lambda * Gamma_i));
    }
    *f = *F * (1.0 - Gamma_v * (PvMv + PiMi - Po) / Gamma_c) / (1.0 - M);
}

/* gortt_brdf.c:118-169 */
static double kc_fn(const gort_o_canopy *c, const gort_o_geom *g, double Kg)
{
    double f, F, beta, junk, f0, F0, f180, F180, Kg0, Kg180, frac;
    kc_fFbeta(c, g, g->raa, Kg, &f, &F, &beta);
    Kg0 = kg_fn(c, g->sza_p, g->vza_p, dtor(0.));
    kc_fFbeta(c, g, dtor(0.), Kg0, &f0, &F0, &junk);
    Kg180 = kg_fn(c, g->sza_p, g->vza_p, dtor(180.));
    kc_fFbeta(c, g, dtor(180.), Kg180, &f180, &F180, &junk);
    frac = g->raa / PI;
    if (frac > 1.0) frac = 2.0 - frac;
    if (c->use_user_beta) beta = c->beta;
    f = (1. - frac) * f0 * F0 + frac * f180 * F180;
    f = beta * f + (1.0 - beta) * F;
    return f * (1.0 - Kg);
}

/* gortt_brdf.c:638-702: Kuusk hot-spot; UNPRIMED angles in cos(xi) */
static double kuusk_fn(const gort_o_canopy *c, const gort_o_geom *g)
{
    const double k_vza = 0.5;                                 /* gortt.c:287 */
    double cos_xi = cos(g->sza) * cos(g->vza) + sin(g->sza) * sin(g->vza) * cos(g->raa);
    double lsza = -log(g->epgap_s) / (c->k * c->favd);
    double lvza = -log(g->epgap_v) / (k_vza * c->favd);
    double t1, t2, H;
    if ((lsza * lsza + lvza * lvza - 2. * lsza * lvza * cos_xi) > 0.0) {
        double lsv = sqrt(lsza * lsza + lvza * lvza - 2. * lsza * lvza * cos_xi);
        t2 = (1.0 - exp(-lsv / c->r)) / (lsv / c->r);
    } else {
        t2 = 1.0;
    }
    t1 = ((lsza * lvza) > 0.0) ? sqrt(lsza * lvza) : 0.0;
    H = exp(c->k * c->favd * t1 * t2);
    return g->epgap_s * g->epgap_v * H;
}

/* ------------------------------------------------------------------- rsurf */

void gort_o_rsurf(const gort_o_canopy *c, gort_o_geom *g, int nw,
                  const double *rsoil, const double *rleaf, const double *tleaf,
                  double *rsurf, double *scomp)
{
    const double fd = g->fd;
    const double ko = c->k_open0, kep = c->k_openep0;
    double Kc, Kg, Kt, Kz, Kprime_g, Kprime_z, kuusk, mu, t_0, t_prime_0;
    int i;

    g->vza_p = prime_theta(c, g->vza);                       /* gortt.c:424-425 */
    g->sza_p = prime_theta(c, g->sza);

    Kg = kg_fn(c, g->sza_p, g->vza_p, g->raa);
    Kc = kc_fn(c, g, Kg);
    Kz = exp(-(c->lambda * PI * pow(c->r, 2)) / cos(g->vza_p)) - Kg;
    Kt = 1.0 - Kc - Kz - Kg;
    Kt = dmax(0.0, Kt);
    Kprime_g = exp(-(c->lambda * PI * c->rr) / cos(g->sza_p)) - Kg;
    Kprime_z = 1.0 - exp(-(c->lambda * PI * c->rr) / cos(g->vza_p)) - Kprime_g;

    /* wavelength-independent terms the reference re-evaluates inside its band loop */
    kuusk = kuusk_fn(c, g);
    mu = cos(g->sza_p);
    t_0 = exp(-(c->k * c->elai * sec(g->sza_p)));            /* gortt_brdf.c:534 */
    t_prime_0 = g->pn0_s + g->epgap_s;                       /* gortt_brdf.c:447 */

    for (i = 0; i < nw; i++) {
        const double rs = rsoil[i];
        double omega = rleaf[i] + tleaf[i];                  /* gortt.c:469 */
        double gam = sqrt(1 - omega);                        /* gortt.c:470 */
        double R_ff = (1.0 - gam) / (1.0 + gam);             /* gortt_brdf.c:574 */
        double R_df = (1.0 - gam) / (1.0 + 2.0 * mu * gam);  /* gortt_brdf.c:552 */
        double T_ff = exp(-(2.0 * gam * c->k * c->elai));    /* gortt_brdf.c:492 */
        double T_df, t_ff, p_ff, t_df, p_df, t_p_df, t_p_ff, kopen, gfun;
        double G, Zd, Zf, Z, CdC, CfC, CdG, CfG, CdCG, CfCG, Cd, Cf, C, Td, Tf, T;

        T_df = (omega / 2.0);                                /* gortt_brdf.c:467-471 */
        T_df *= (1. + 2. * mu) / (1. - pow((2. * gam * mu), 2));
        T_df *= (T_ff - t_0);

        t_ff = T_ff;                                         /* gortt_brdf.c:401-403 */
        t_ff *= (1. - pow(R_ff, 2));
        t_ff /= (1. - pow(R_ff * T_ff, 2));

        p_ff = R_ff;                                         /* gortt_brdf.c:510-512 */
        p_ff *= (1. - pow(T_ff, 2));
        p_ff /= (1. - pow(T_ff * R_ff, 2));

        t_df = T_df - p_ff * (t_0 * R_df + T_df * R_ff);     /* gortt_brdf.c:423-424 */
        p_df = R_df - t_ff * (t_0 * R_df + T_df * R_ff);     /* gortt_brdf.c:628-630 */

        t_p_df = t_df * (1 - t_prime_0);                     /* gortt_brdf.c:361 */
        kopen = ko + kep;                                    /* gortt_brdf.c:381 */
        t_p_ff = t_ff * (1.0 - kopen) + kopen;               /* gortt_brdf.c:382 */
        gfun = -(4.0 / 9.0) * (rleaf[i] - tleaf[i]) / (omega);   /* gortt_brdf.c:591 */

        G = fd * rs + (1 - fd) * rs;                         /* gortt.c:481-484 */
        Zd = (t_p_df + g->epgap_s) * rs;                     /* gortt.c:491 */
        Zf = (t_p_ff - kep) * rs;                            /* gortt.c:492 */
        Z = fd * Zd + (1 - fd) * Zf;

        CdC = p_df + ((1.0 - omega) * kuusk * omega * (1.0 - gfun))
                     / (2.0 * cos(g->sza_p) * cos(g->vza_p)); /* gortt.c:504-507 */
        CfC = p_ff;
        CdG = (Z * Kprime_z + G * Kprime_g) * kep;           /* gortt.c:514 */
        CfG = ((kep + ko) * G + (1 - (kep + ko)) * Z) * kep; /* gortt.c:516-517 */
        CdCG = (t_p_df + t_prime_0) * (rs / (1.0 - rs * p_ff)) * (t_p_ff - ko);  /* gortt.c:519-521 */
        CfCG = t_p_ff * (rs / (1.0 - rs * p_ff)) * (t_p_ff - ko);               /* gortt.c:523-525 */
        Cd = CdC + CdG + CdCG;
        Cf = CfC + CfG + CfCG;
        C = fd * Cd + (1 - fd) * Cf;
        Td = CdCG;  Tf = CfCG;                               /* gortt.c:541-547: same expressions */
        T = fd * Td + (1 - fd) * Tf;

        rsurf[i] = Kc * C + Kg * G + Kt * T + Kz * Z;        /* gortt.c:557 */
        if (scomp) { scomp[4 * i] = C; scomp[4 * i + 1] = G; scomp[4 * i + 2] = T; scomp[4 * i + 3] = Z; }
    }
    g->Kc = Kc;  g->Kg = Kg;  g->Kt = Kt;  g->Kz = Kz;
}

/* --------------------------------------------------------- albedo / energy */

/* Numerical-Recipes style Gauss-Legendre; gortt_albedo.c:141-199 */
void gort_o_gauleg(double x1, double x2, double *x, double *w, int n)
{
    int m = (n + 1) / 2, i, j;
    double xm = 0.5 * (x2 + x1), xl = 0.5 * (x2 - x1);
    for (i = 0; i < m; i++) {
        double z = cos(3.141592654 * (i + 0.75) / (n + 0.5)), z1, pp, p1, p2, p3;
        do {
            p1 = 1.0;  p2 = 0.0;
            for (j = 1; j <= n; j++) {
                p3 = p2;  p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            z1 = z;
            z = z1 - p1 / pp;
        } while (fabs(z - z1) > 3.0e-11);
        x[i] = xm - xl * z;
        x[n - 1 - i] = xm + xl * z;
        w[i] = 2.0 * xl / ((1.0 - z * z) * pp * pp);
        w[n - 1 - i] = w[i];
    }
}

/*
 * gortt_albedo.c:62-138 then :7-60.  g must be normalised; on return g->vza/vaa/raa
 * hold the last quadrature node, as in the reference.  Unlike the reference the
 * partial sums are sized nw (it sizes them npoints and overflows the heap for nw>32).
 */
void gort_o_energy(const gort_o_canopy *c, gort_o_geom *g, int nw,
                   const double *rsoil, const double *rleaf, const double *tleaf,
                   const double *abscissa, const double *weights,
                   double *albedo, double *favegt, double *fasoil)
{
    const int np = GORT_O_NPOINTS;
    double *sum_y = (double *)calloc((size_t)nw, sizeof(double));
    double *sum_x = (double *)calloc((size_t)nw, sizeof(double));
    double *rsurf = (double *)calloc((size_t)nw, sizeof(double));
    double *scomp = (double *)calloc((size_t)nw * 4, sizeof(double));
    double xm = 0.5 * (1. - 1.), xr = 0.5 * (1. + 1.);
    double ym = 0.5 * (2. * PI - 0.), yr = 0.5 * (2. * PI + 0.);
    int i, j, k;

    for (i = 0; i < np; i++) {
        double y = ym + yr * abscissa[i];
        g->vaa = y;
        while (g->vaa > 2 * PI) g->vaa -= 2 * PI;
        g->raa = g->saa - g->vaa;
        g->raa = fabs((g->raa - 2 * PI * (int)(0.5 + g->raa * M_1_PI * 0.5)));
        for (k = 0; k < nw; k++) sum_x[k] = 0.;
        for (j = (int)(np / 2.); j < np; j++) {
            double x = xm + xr * abscissa[j];
            g->vza = acos(x);
            if (g->vza < 0.0) { g->vaa += PI; g->vza *= -1.0; }
            g->vza_p = prime_theta(c, g->vza);
            g->sza_p = prime_theta(c, g->sza);
            gort_o_set_zenith_probabilities(c, g);
            gort_o_rsurf(c, g, nw, rsoil, rleaf, tleaf, rsurf, scomp);
            for (k = 0; k < nw; k++)
                sum_x[k] = sum_x[k] + rsurf[k] * weights[j] * fabs(x) * xr;
        }
        for (k = 0; k < nw; k++) sum_y[k] = sum_y[k] + sum_x[k] * weights[i] * yr;
    }
    for (k = 0; k < nw; k++) albedo[k] = sum_y[k] / PI;

    {   /* gortt_albedo.c:37-58; G,Z from the LAST rsurf call (they depend on sza, band only) */
        double Fd1 = 1., Pn0 = g->pn0_s;
        for (k = 0; k < nw; k++) {
            double Fu1 = albedo[k], rs = rsoil[k];
            double G = scomp[k * 4 + 1], Z = scomp[k * 4 + 3];
            double Fu2 = G * Pn0 + Z * (1. - Pn0);
            double Fd2 = Pn0 + Z * (1. - Pn0) / rs;
            favegt[k] = Fd1 - Fu1 - Fd2 + Fu2;
            fasoil[k] = Fd2 - Fu2;
        }
    }
    free(sum_y); free(sum_x); free(rsurf); free(scomp);
}

/* ----------------------------------------------------------------- spectra */

/* gortt.c:1286-1328. Returns -1 on a wavelength outside [400,2500]. */
int gort_o_price_soil(const double *wl, int nw, const double rsl[4], double *rsoil)
{
    const double *v1 = gort_o_soil_tab, *v2 = v1 + 421, *v3 = v2 + 421, *v4 = v3 + 421;
    int i;
    for (i = 0; i < nw; i++) {
        int upper, lower;
        double fraction, lo, up;
        if (!(wl[i] >= 400 && wl[i] <= 2500)) return -1;      /* NaN too (the reference would index with (int)NaN) */
        upper = (int)(1. + (wl[i] - 400) / 5.0);
        lower = (int)((wl[i] - 400) / 5.0);
        fraction = (double)(wl[i] - 400.) / 5.0 - lower;
        lo = rsl[0] * v1[lower] + rsl[1] * v2[lower] + rsl[2] * v3[lower] + rsl[3] * v4[lower];
        /* at 2500 nm the reference reads one element past each table, times fraction = 0 */
        up = (upper > 420) ? 0.0
           : rsl[0] * v1[upper] + rsl[1] * v2[upper] + rsl[2] * v3[upper] + rsl[3] * v4[upper];
        rsoil[i] = lo * (1 - fraction) + up * fraction;
    }
    return 0;
}

/* tav_abs.f90:16-60; pi and the tables are single precision there */
static double tav(double theta, double nr)
{
    const double pi = (double)(atanf(1.0f) * 4.0f);
    double rd = pi / 180.;
    double n2 = nr * nr, np = n2 + 1., nm = n2 - 1.;   /* flang lowers x**2. to x*x (no pow call) */
    double a = (nr + 1) * (nr + 1.) / 2.;
    double k = -((n2 - 1) * (n2 - 1.) / 4.);
    double sa = sin(theta * rd);
    double b1, b2, b, b3, a3, ts, tp1, tp2, tp3, tp4, tp5, tp;
    if (theta == 90.) b1 = 0.;
    else b1 = sqrt((sa * sa - np / 2) * (sa * sa - np / 2) + k);
    b2 = sa * sa - np / 2;
    b = b1 - b2;
    b3 = b * b * b;
    a3 = a * a * a;
    ts = (k * k / (6 * b3) + k / b - b / 2) - (k * k / (6 * a3) + k / a - a / 2);
    tp1 = -(2 * n2 * (b - a) / (np * np));
    tp2 = -(2 * n2 * np * log(b / a) / (nm * nm));
    tp3 = n2 * (1. / b - 1. / a) / 2;
    tp4 = 16 * (n2 * n2) * (n2 * n2 + 1) * log((2 * np * b - nm * nm) / (2 * np * a - nm * nm))
          / ((np * np * np) * (nm * nm));
    tp5 = 16 * (n2 * n2 * n2) * (1. / (2 * np * b - nm * nm) - 1. / (2 * np * a - nm * nm)) / (np * np * np);
    tp = tp1 + tp2 + tp3 + tp4 + tp5;
    return (ts + tp) / (2 * (sa * sa));
}

/* prospect_DB.f90:94-189 */
void gort_o_prospect_d(double N, double Cab, double Car, double Anth, double Cbrown,
                       double Cw, double Cm, double *RT)
{
    const float *nr_t = gort_o_prospect_tab;
    const float *kCab = nr_t + GORT_O_NBANDS, *kCar = kCab + GORT_O_NBANDS;
    const float *kAnth = kCar + GORT_O_NBANDS, *kBrown = kAnth + GORT_O_NBANDS;
    const float *kCw = kBrown + GORT_O_NBANDS, *kCm = kCw + GORT_O_NBANDS;
    int i;
    for (i = 0; i < GORT_O_NBANDS; i++) {
        double nr = nr_t[i];
        double k = (Cab * kCab[i] + Car * kCar[i] + Anth * kAnth[i] + Cbrown * kBrown[i]
                    + Cw * kCw[i] + Cm * kCm[i]) / N;
        double tau, xx, yy;
        double t12, talf, ralf, r12, t21, r21, denom, Ta, Ra, t, r;
        double D, rq, tq, a, b, bNm1, bN2, a2, Rsub, Tsub;

        if (k <= 0.0) {
            tau = 1;
        } else if (k <= 4.0) {
            xx = 0.5 * k - 1.0;
            yy = (((((((((((((((-3.60311230482612224e-13
                * xx + 3.46348526554087424e-12) * xx - 2.99627399604128973e-11)
                * xx + 2.57747807106988589e-10) * xx - 2.09330568435488303e-9)
                * xx + 1.59501329936987818e-8) * xx - 1.13717900285428895e-7)
                * xx + 7.55292885309152956e-7) * xx - 4.64980751480619431e-6)
                * xx + 2.63830365675408129e-5) * xx - 1.37089870978830576e-4)
                * xx + 6.47686503728103400e-4) * xx - 2.76060141343627983e-3)
                * xx + 1.05306034687449505e-2) * xx - 3.57191348753631956e-2)
                * xx + 1.07774527938978692e-1) * xx - 2.96997075145080963e-1;
            yy = (yy * xx + 8.64664716763387311e-1) * xx + 7.42047691268006429e-1;
            yy = yy - log(k);
            tau = (1.0 - k) * exp(-k) + k * k * yy;
        } else if (k <= 85.0) {
            xx = 14.5 / (k + 3.25) - 1.0;
            yy = (((((((((((((((-1.62806570868460749e-12
                * xx - 8.95400579318284288e-13) * xx - 4.08352702838151578e-12)
                * xx - 1.45132988248537498e-11) * xx - 8.35086918940757852e-11)
                * xx - 2.13638678953766289e-10) * xx - 1.10302431467069770e-9)
                * xx - 3.67128915633455484e-9) * xx - 1.66980544304104726e-8)
                * xx - 6.11774386401295125e-8) * xx - 2.70306163610271497e-7)
                * xx - 1.05565006992891261e-6) * xx - 4.72090467203711484e-6)
                * xx - 1.95076375089955937e-5) * xx - 9.16450482931221453e-5)
                * xx - 4.05892130452128677e-4) * xx - 2.14213055000334718e-3;
            yy = ((yy * xx - 1.06374875116569657e-2) * xx - 8.50699154984571871e-2) * xx
                 + 9.23755307807784058e-1;
            yy = exp(-k) * yy / k;
            tau = (1.0 - k) * exp(-k) + k * k * yy;
        } else {
            tau = 0;
        }

        t12 = tav(90., nr);
        talf = tav(40., nr);
        ralf = 1. - talf;
        r12 = 1. - t12;
        t21 = t12 / (nr * nr);
        r21 = 1 - t21;
        denom = 1 - r21 * r21 * (tau * tau);
        Ta = talf * tau * t21 / denom;
        Ra = ralf + r21 * tau * Ta;
        t = t12 * tau * t21 / denom;
        r = r12 + r21 * tau * t;

        D = sqrt((1. + r + t) * (1. + r - t) * (1. - r + t) * (1. - r - t));
        rq = r * r;  tq = t * t;
        a = (1. + rq - tq + D) / (2 * r);
        b = (1. - rq + tq + D) / (2 * t);
        bNm1 = pow(b, (N - 1));
        bN2 = bNm1 * bNm1;
        a2 = a * a;
        denom = a2 * bN2 - 1.;
        Rsub = a * (bN2 - 1.) / denom;
        Tsub = bNm1 * (a2 - 1.) / denom;
        if (r + t >= 1.0) {
            Tsub = t / (t + (1. - t) * (N - 1));
            Rsub = 1 - Tsub;
        }
        denom = 1 - Rsub * r;
        RT[GORT_O_NBANDS + i] = Ta * Tsub / denom;
        RT[i] = Ra + Ta * Rsub * t / denom;
    }
}

/* gortt.c:1349-1371: linear interpolation with a FLOAT fraction. -1 if out of range. */
int gort_o_leaf_interp(const double *wl, int nw, const double *RT, double *rleaf, double *tleaf)
{
    int i;
    for (i = 0; i < nw; i++) {
        int upper, lower;
        float fraction, omf;
        double ru, tu;
        if (!(wl[i] >= 400 && wl[i] <= 2500)) return -1;      /* NaN too (the reference would index with (int)NaN) */
        upper = (int)(1 + (wl[i] - 400.0) / 1.0);
        lower = (int)((wl[i] - 400.0) / 1.0);
        fraction = (float)((float)(wl[i] - 400.0) / 1.0 - lower);
        omf = 1 - fraction;                    /* float arithmetic, as (1-fraction) in C */
        /* at 2500 nm the reference reads past the R block (into T) / past the array, times 0 */
        ru = (upper > GORT_O_NBANDS - 1) ? 0.0 : RT[upper];
        tu = (upper > GORT_O_NBANDS - 1) ? 0.0 : RT[upper + GORT_O_NBANDS];
        rleaf[i] = RT[lower] * omf + ru * fraction;
        tleaf[i] = RT[lower + GORT_O_NBANDS] * omf + tu * fraction;
    }
    return 0;
}

/* ------------------------------------------------------------ batch drivers */

void gort_o_rsurf_stream(const gort_o_canopy *c, const double *angles_deg, long nA, int nw,
                         const double *rsoil, const double *rleaf, const double *tleaf,
                         double *rsurf, double *scomp, double *K)
{
    long a;
    for (a = 0; a < nA; a++) {
        gort_o_geom g;
        const double *q = angles_deg + 4 * a;
        gort_o_normalise_angles(c, q[0], q[1], q[2], q[3], &g);
        gort_o_rsurf(c, &g, nw, rsoil, rleaf, tleaf, rsurf + a * nw,
                     scomp ? scomp + a * 4 * nw : NULL);
        if (K) { K[4 * a] = g.Kc; K[4 * a + 1] = g.Kg; K[4 * a + 2] = g.Kt; K[4 * a + 3] = g.Kz; }
    }
}

void gort_o_energy_stream(const gort_o_canopy *c, const double *angles_deg, long nA, int nw,
                          const double *rsoil, const double *rleaf, const double *tleaf,
                          double *energy)
{
    double x[GORT_O_NPOINTS], w[GORT_O_NPOINTS];
    double *alb = (double *)calloc((size_t)nw * 3, sizeof(double));
    long a;
    int k;
    gort_o_gauleg(-1., 1., x, w, GORT_O_NPOINTS);
    for (a = 0; a < nA; a++) {
        gort_o_geom g;
        const double *q = angles_deg + 4 * a;
        gort_o_normalise_angles(c, q[0], q[1], q[2], q[3], &g);
        gort_o_energy(c, &g, nw, rsoil, rleaf, tleaf, x, w, alb, alb + nw, alb + 2 * nw);
        for (k = 0; k < nw; k++) {
            energy[(a * nw + k) * 3 + 0] = alb[k];
            energy[(a * nw + k) * 3 + 1] = alb[nw + k];
            energy[(a * nw + k) * 3 + 2] = alb[2 * nw + k];
        }
    }
    free(alb);
}
