/*
 * ref_shim.c -- TEST INFRASTRUCTURE (oracle side). Not part of the product.
 *
 * Flat C entry points (ctypes-friendly) over the REAL reference functions, so that
 * tools/make_golden.py can dump function-level values at full precision:
 * spectra providers, the gap-probability tables and gortt_rsurf / gortt_energy for
 * an already-normalised geometry.  Built only by `make -C oracle ref`, which
 * compiles the reference sources in place (gortt.c with -Dmain=gortt_ref_main) and
 * links them with this file into oracle/_ref/libgortt_ref.so.  <gortt.h> is the
 * reference's own header, found through -I/root/reference/include at compile
 * time; no reference source text lives in this repository.
 *
 * The defaults below restate the assignments at the top of the reference's main()
 * (/root/reference/gortt.c:32-96), which are not callable; the CLI-level goldens
 * (oracle/_ref/gortt_fp) cross-check them.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <gortt.h>

extern double default_soil_vector_1[], default_soil_vector_2[];
extern double default_soil_vector_3[], default_soil_vector_4[];
void gortt_gap_probabilities_Q08(gortt_parameters *, gortt_geometry *);

static gortt_parameters P;
static gortt_geometry   G;
static gortt_spectra    S;
static gortt_control    C;
static int have_canopy = 0;

static void defaults(void)
{
    memset(&P, 0, sizeof P); memset(&G, 0, sizeof G);
    memset(&S, 0, sizeof S); memset(&C, 0, sizeof C);
    S.rsl1 = 0.2; S.rsl2 = 0.1; S.rsl3 = 0.03726; S.rsl4 = -0.002426;
    S.p_N = 1.2; S.p_Cab = 30.; S.p_Car = 10.; S.p_Anth = 1.0;
    S.p_Cbrown = 0.0; S.p_Cw = 0.015; S.p_Cm = 0.009;
    P.lambda = 0.405; P.r = 0.76; P.b = 3.55263 * P.r;
    P.h1 = 3.0; P.h2 = 8.5; P.favd = 0.858;
    P.dz = 0.20; P.ds = 0.20; P.dth = DTOR(1);
    P.nlayers = 15; P.lad = LAD_05; P.maxcrowns = 30; P.nh_es = 20;
    P.npoints = 32; P.use_user_fd = FALSE;
}

/* argv[0] is a program name, as for main(). Runs parser + init + gap probabilities. */
int refshim_canopy(int argc, char **argv)
{
    defaults();
    gortt_cl_parser(argc, argv, &P, &G, &S, &C);
    gortt_init_params(&P, &G);
    if (C.use_q08_pn_kopen) gortt_gap_probabilities_Q08(&P, &G);
    else                    gortt_gap_probabilities(&P, &G);
    P.abscissa = (double *)malloc(P.npoints * sizeof(double));
    P.weights  = (double *)malloc(P.npoints * sizeof(double));
    gauleg(-1., 1., P.abscissa, P.weights, P.npoints);
    have_canopy = 1;
    return 0;
}

/* scalars[32]: derived canopy scalars in a fixed order (see tools/make_golden.py) */
void refshim_canopy_scalars(double *sc)
{
    int i = 0;
    sc[i++] = P.r; sc[i++] = P.b; sc[i++] = P.h1; sc[i++] = P.h2; sc[i++] = P.lambda;
    sc[i++] = P.favd; sc[i++] = P.ellipticity; sc[i++] = P.h; sc[i++] = P.elai;
    sc[i++] = P.tau; sc[i++] = P.z1; sc[i++] = P.z2; sc[i++] = P.lv; sc[i++] = P.favd_p;
    sc[i++] = P.tau_p; sc[i++] = P.lv_p; sc[i++] = P.z1_p; sc[i++] = P.z2_p;
    sc[i++] = P.h1_p; sc[i++] = P.h2_p; sc[i++] = P.dz; sc[i++] = P.ds; sc[i++] = P.dz_p;
    sc[i++] = P.dth; sc[i++] = (double)P.nth; sc[i++] = (double)P.nlayers;
    sc[i++] = P.k; sc[i++] = P.rr; sc[i++] = P.rrr;
    sc[i++] = (double)P.use_user_beta; sc[i++] = P.beta; sc[i++] = (double)P.use_user_fd;
}

/* tables: p_n0[15*91], p_s0[15*91], v_g[15*91], epgap0[91], theta[91], theta_p[91],
 * height[15], height_p[15], kk[2] = k_open[0], k_openep[0] */
void refshim_gap_tables(double *p_n0, double *p_s0, double *v_g, double *epgap0,
                        double *theta, double *theta_p, double *height, double *height_p,
                        double *kk)
{
    int h, t;
    for (h = 0; h < P.nlayers; h++)
        for (t = 0; t < P.nth; t++) {
            p_n0[h * P.nth + t] = P.p_n0[h][t];
            p_s0[h * P.nth + t] = P.p_s0[h][t];
            v_g[h * P.nth + t]  = P.v_g[h][t];
        }
    for (t = 0; t < P.nth; t++) {
        epgap0[t] = P.epgap[0][t]; theta[t] = P.theta[t]; theta_p[t] = P.theta_p[t];
    }
    for (h = 0; h < P.nlayers; h++) { height[h] = P.height[h]; height_p[h] = P.height_p[h]; }
    kk[0] = P.k_open[0]; kk[1] = P.k_openep[0];
}

void refshim_gauleg(double *x, double *w, int n) { gauleg(-1., 1., x, w, n); }

/* PROSPECT-D straight from the Fortran: out[0..2100]=R, out[2101..4201]=T */
void refshim_prospect_raw(double N, double Cab, double Car, double Anth, double Cbrown,
                          double Cw, double Cm, double *out)
{
    prospect_DB_(&N, &Cab, &Car, &Anth, &Cbrown, &Cw, &Cm, out);
}

/* spectra for the current flags (call refshim_canopy first so -cab/-rsl1/-alb_* are parsed) */
int refshim_spectra(const double *wl, int nw, double *rsoil, double *rleaf, double *tleaf)
{
    if (!have_canopy) return -1;
    S.nw = nw;
    S.wavelength = (double *)malloc(nw * sizeof(double));
    memcpy(S.wavelength, wl, nw * sizeof(double));
    S.rsurf = (double *)calloc(nw, sizeof(double));
    S.rsoil = (double *)calloc(nw, sizeof(double));
    S.rleaf = (double *)calloc(nw, sizeof(double));
    S.tleaf = (double *)calloc(nw, sizeof(double));
    S.scomp = (double *)calloc(4 * nw, sizeof(double));
    /* sized nw (not npoints) on purpose: avoids the reference's own overflow at nw>32
       only for the arrays WE own; gortt_albedo's internal sum_x/sum_y still overflow,
       so refshim_energy refuses nw>32. */
    S.albedo = (double *)calloc(nw, sizeof(double));
    S.favegt = (double *)calloc(nw, sizeof(double));
    S.fasoil = (double *)calloc(nw, sizeof(double));
    gortt_price_soil(&S, default_soil_vector_1, default_soil_vector_2,
                     default_soil_vector_3, default_soil_vector_4);
    gortt_prospect_interface(&S);
    memcpy(rsoil, S.rsoil, nw * sizeof(double));
    memcpy(rleaf, S.rleaf, nw * sizeof(double));
    memcpy(tleaf, S.tleaf, nw * sizeof(double));
    return 0;
}

/* geometry already normalised as main() would (radians; vza,sza>=0; raa given). */
static void set_geom(double vza, double vaa, double sza, double saa, double raa)
{
    G.vza = vza; G.vaa = vaa; G.sza = sza; G.saa = saa; G.raa = raa;
    G.vza_prime = gortt_prime_theta(&P, G.vza);
    G.sza_prime = gortt_prime_theta(&P, G.sza);
    P.k_vza = gortt_leaf_angle_distribution(&P, G.vza);
    if (!P.use_user_fd) P.fd = cos(G.sza) / (cos(G.sza) + 0.09);
    gortt_set_zenith_dependant_probabilities(&P, &G);
}

/* out: rsurf[nw], scomp[4nw], K[4]=Kc,Kg,Kt,Kz, pr[4]=Pn0_s,EPgap_s,Pn0_v,EPgap_v */
void refshim_rsurf(double vza, double vaa, double sza, double saa, double raa,
                   double *rsurf, double *scomp, double *K, double *pr)
{
    set_geom(vza, vaa, sza, saa, raa);
    gortt_rsurf(&P, &G, &S);
    memcpy(rsurf, S.rsurf, S.nw * sizeof(double));
    memcpy(scomp, S.scomp, 4 * S.nw * sizeof(double));
    K[0] = G.Kc; K[1] = G.Kg; K[2] = G.Kt; K[3] = G.Kz;
    pr[0] = P.p_neq0_heq0_sza; pr[1] = P.p_ngt0_heq0_sza;
    pr[2] = P.p_neq0_heq0_vza; pr[3] = P.p_ngt0_heq0_vza;
}

int refshim_energy(double vza, double vaa, double sza, double saa, double raa,
                   double *albedo, double *favegt, double *fasoil)
{
    if (S.nw > 32) return -1;   /* reference heap overflow, gortt_albedo.c:79-88 */
    set_geom(vza, vaa, sza, saa, raa);
    gortt_rsurf(&P, &G, &S);
    gortt_energy(&P, &G, &S);
    memcpy(albedo, S.albedo, S.nw * sizeof(double));
    memcpy(favegt, S.favegt, S.nw * sizeof(double));
    memcpy(fasoil, S.fasoil, S.nw * sizeof(double));
    return 0;
}
