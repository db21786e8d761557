/*
 * gort_oracle.h -- TEST INFRASTRUCTURE: CPU restatement of the GORT hot path.
 *
 * This is the parity oracle.  It is NOT the product and nothing under gort_amd/
 * may include, link or call it; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py do, and only as the checker.
 *
 * Pinned (tests/test_oracle_golden.py) against tests/golden/ fixtures, which were
 * produced by tools/make_golden.py from the REAL reference compiled in place
 * (oracle/Makefile `ref` target): CLI runs of oracle/_ref/gortt_fp (full-precision
 * stdout of the reference main()) and function-level dumps through
 * oracle/_ref/libgortt_ref.so.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 * Plain scalar IEEE double, strictly sequential, evaluation order of each formula
 * kept as in the reference so that results agree to the last bits (compile with
 * -ffp-contract=off).
 */
#ifndef GORT_ORACLE_H
#define GORT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define GORT_O_NLAYERS   15
#define GORT_O_NTH       91
#define GORT_O_MAXCROWNS 30
#define GORT_O_NH_ES     20
#define GORT_O_NPOINTS   32
#define GORT_O_NBANDS    2101

typedef struct {
    /* ---- inputs (gortt.c:67-72 defaults; flags gortt.c:1026-1131) ---- */
    double r, b, h1, h2, lambda, favd;
    int    use_user_beta;  double beta;        /* -beta     gortt.c:1039 */
    int    use_user_fd;    double fd_user;     /* -diffuse  gortt.c:1041 (fd = 1-arg) */
    /* ---- derived scalars (gortt_init_params, gortt.c:641-714) ---- */
    double ell, rr, rrr, h, k, elai, tau, z1, z2, lv;
    double favd_p, tau_p, lv_p, z1_p, z2_p, h1_p, h2_p;
    double dz, ds, dz_p, dth;
    int    nlayers, nth, maxcrowns, nh_es;
    double height[GORT_O_NLAYERS], height_p[GORT_O_NLAYERS];
    double theta[GORT_O_NTH], theta_p[GORT_O_NTH];
    double factorial[GORT_O_MAXCROWNS + 1];
    /* ---- gap-probability products (gortt_pn_kopen.c) ---- */
    double v_g [GORT_O_NLAYERS][GORT_O_NTH];
    double p_n0[GORT_O_NLAYERS][GORT_O_NTH];
    double p_s0[GORT_O_NLAYERS][GORT_O_NTH];
    double es0[GORT_O_NTH];            /* ES(z=height[0], t) */
    double epgap0[GORT_O_NTH];         /* epgap[0][t]; [90] stays 0 */
    double k_open0, k_openep0;
} gort_o_canopy;

typedef struct {                       /* gortt_geometry, gortt.h:53-69 (radians) */
    double vza, vaa, sza, saa, raa, vza_p, sza_p;
    double fd;                         /* gortt.c:290-291 */
    double pn0_s, epgap_s, pn0_v, epgap_v;   /* gortt.c:896-910 */
    double Kc, Kg, Kt, Kz;
} gort_o_geom;

/* canopy ------------------------------------------------------------------ */
void gort_o_canopy_defaults(gort_o_canopy *c);                       /* gortt.c:67-96 */
void gort_o_canopy_newstyle(gort_o_canopy *c, float hb, float br, float pcc); /* gortt.c:1117-1125 */
void gort_o_canopy_set_lai(gort_o_canopy *c, float lai);             /* gortt.c:1127-1131 */
void gort_o_canopy_init(gort_o_canopy *c);                           /* gortt.c:632-868 */
int  gort_o_gap_probabilities(gort_o_canopy *c);                     /* gortt_pn_kopen.c:7-129 (live products) */
void gort_o_gap_probabilities_q08(gort_o_canopy *c);                 /* gortt_pn_kopen.c:1144-1200 */

/* building blocks exposed for unit goldens */
double gort_o_crown_proj_volume(const gort_o_canopy *c, double t, double h);      /* :149-167 */
double gort_o_get_es(const gort_o_canopy *c, int z, int t);                        /* :534-563 */
double gort_o_vol(const gort_o_canopy *c, int h, int h_s, int t, double h_b);      /* :665-768 */

/* geometry ---------------------------------------------------------------- */
void gort_o_normalise_angles(const gort_o_canopy *c, double vza_deg, double vaa_deg,
                             double sza_deg, double saa_deg, gort_o_geom *g);       /* gortt.c:240-294 */
void gort_o_set_zenith_probabilities(const gort_o_canopy *c, gort_o_geom *g);      /* gortt.c:872-915 */

/* BRDF -------------------------------------------------------------------- */
void gort_o_rsurf(const gort_o_canopy *c, gort_o_geom *g, int nw,
                  const double *rsoil, const double *rleaf, const double *tleaf,
                  double *rsurf, double *scomp /* 4*nw or NULL */);                 /* gortt.c:385-578 */
void gort_o_energy(const gort_o_canopy *c, gort_o_geom *g, int nw,
                   const double *rsoil, const double *rleaf, const double *tleaf,
                   const double *abscissa, const double *weights,
                   double *albedo, double *favegt, double *fasoil);                 /* gortt_albedo.c:7-138 */
void gort_o_gauleg(double x1, double x2, double *x, double *w, int n);             /* gortt_albedo.c:141-199 */

/* spectra ----------------------------------------------------------------- */
int  gort_o_price_soil(const double *wl, int nw, const double rsl[4], double *rsoil);   /* gortt.c:1286-1328 */
void gort_o_prospect_d(double N, double Cab, double Car, double Anth, double Cbrown,
                       double Cw, double Cm, double *RT /* [2*2101]: R then T */);     /* prospect_DB.f90:72-191 */
int  gort_o_leaf_interp(const double *wl, int nw, const double *RT,
                        double *rleaf, double *tleaf);                                  /* gortt.c:1349-1371 */

/* convenience batch drivers (used by tests and bench.py cpu_baseline "port") -- */
/* angles_deg[nA][4] = vza vaa sza saa as on the gortt stdin stream; out rsurf[nA][nw],
 * optional scomp[nA][4nw], K[nA][4]. */
void gort_o_rsurf_stream(const gort_o_canopy *c, const double *angles_deg, long nA, int nw,
                         const double *rsoil, const double *rleaf, const double *tleaf,
                         double *rsurf, double *scomp, double *K);
/* energy[nA][3*nw] = albedo,favegt,fasoil interleaved per band, as printed (gortt.c:323-324) */
void gort_o_energy_stream(const gort_o_canopy *c, const double *angles_deg, long nA, int nw,
                          const double *rsoil, const double *rleaf, const double *tleaf,
                          double *energy);

#ifdef __cplusplus
}
#endif
#endif
