/* ORACLE / TEST INFRASTRUCTURE -- not product code.  See mw_oracle.h.
 *
 * Each function cites the lines of /root/reference/molint.F90 it restates.
 * The arithmetic keeps the reference's operation order (and this file is
 * built with -ffp-contract=off) so that agreement with the compiled
 * reference is at the 1e-15 level, far inside the 1e-10 parity bar.
 */
#include "mw_oracle.h"
#include <math.h>
#include <stddef.h>

#define MWO_MAXNEIGH_HARD 256

/* molint.F90:64-74, constants.f90:42-43 */
static const double ANG_TO_BOHR = 1.0 / 0.5291772108;
#define MW_SIGMA   (2.3925 * ANG_TO_BOHR)
#define MW_EPSILON (6.189 / 627.509469)
static const double MW_LAMBDA = 23.15;
static const double SW_BIGA   = 7.049556277;
static const double SW_B      = 0.6022245584;
static const double SW_GAMMA  = 1.2;
static const double SW_A      = 1.8;
/* molint.F90:74: the literal has no _dp suffix, so the reference holds the
 * single-precision value widened to double. */
#define COS0 ((double)(-0.33331324756f))

void mwo_constants(double out[8])
{
    out[0] = MW_SIGMA; out[1] = MW_EPSILON; out[2] = MW_LAMBDA; out[3] = SW_BIGA;
    out[4] = SW_B;     out[5] = SW_GAMMA;   out[6] = SW_A;      out[7] = COS0;
}

/* molint.F90:174-217 */
int mwo_compute_ivects(const double h[9], double *ivect, int max_ivect)
{
    const double sigma = MW_SIGMA;
    const double *h1 = h, *h2 = h + 3, *h3 = h + 6;
    /* :189-191 */
    int im = (int)floor(SW_A * sigma / sqrt(h1[0]*h1[0] + h1[1]*h1[1] + h1[2]*h1[2])) + 1;
    int jm = (int)floor(SW_A * sigma / sqrt(h2[0]*h2[0] + h2[1]*h2[1] + h2[2]*h2[2])) + 1;
    int km = (int)floor(SW_A * sigma / sqrt(h3[0]*h3[0] + h3[1]*h3[1] + h3[2]*h3[2])) + 1;
    int nivect = (2*im + 1) * (2*jm + 1) * (2*km + 1);       /* :193 */
    if (nivect > max_ivect) return -1;

    ivect[0] = ivect[1] = ivect[2] = 0.0;                    /* :197 central cell first */
    int k = 1;
    for (int ic = -im; ic <= im; ++ic) {                     /* :200-213 */
        double sx[3] = { (double)ic * h1[0], (double)ic * h1[1], (double)ic * h1[2] };
        for (int jc = -jm; jc <= jm; ++jc) {
            double sy[3] = { (double)jc * h2[0], (double)jc * h2[1], (double)jc * h2[2] };
            for (int kc = -km; kc <= km; ++kc) {
                double sz[3] = { (double)kc * h3[0], (double)kc * h3[1], (double)kc * h3[2] };
                if (ic == 0 && jc == 0 && kc == 0) continue; /* :207 */
                for (int d = 0; d < 3; ++d) ivect[3*k + d] = (sx[d] + sy[d]) + sz[d]; /* :208 */
                ++k;
            }
        }
    }
    return nivect;
}

/* molint.F90:501-559 */
int mwo_compute_neighbours(int n, const double *xyz, const double *ivect, int nivect,
                           int maxneigh, int *nn, int *jn, int *vn)
{
    const double rn = SW_A * MW_SIGMA * 1.18;                /* :516 */
    const double rn2 = rn * rn;                              /* :537 */
    int maxnn = 0, overflow = 0;
    for (int i = 0; i < n; ++i) {                            /* :520 */
        const double *ri = xyz + 3*i;
        int cnt = 0;
        for (int j = 0; j < n; ++j) {                        /* :525 */
            const double *rj = xyz + 3*j;
            double v[3] = { rj[0] - ri[0], rj[1] - ri[1], rj[2] - ri[2] };  /* :529 */
            for (int k = 0; k < nivect; ++k) {               /* :531 */
                if (k == 0 && j == i) continue;              /* :532 */
                double t0 = v[0] + ivect[3*k], t1 = v[1] + ivect[3*k+1], t2 = v[2] + ivect[3*k+2]; /* :534 */
                double r2 = t0*t0 + t1*t1 + t2*t2;           /* :535 */
                if (r2 < rn2) {                              /* :537-542 */
                    if (cnt < maxneigh) {
                        jn[(size_t)maxneigh*i + cnt] = j + 1;
                        vn[(size_t)maxneigh*i + cnt] = k + 1;
                    } else {
                        overflow = 1;
                    }
                    ++cnt;
                }
            }
        }
        nn[i] = cnt < maxneigh ? cnt : maxneigh;
        for (int s = nn[i]; s < maxneigh; ++s) { jn[(size_t)maxneigh*i + s] = 0; vn[(size_t)maxneigh*i + s] = 0; }
        if (cnt > maxnn) maxnn = cnt;
    }
    return overflow ? -1 : maxnn;
}

/* molint.F90:407-499 */
double mwo_model_energy(int n, const double *xyz, const double *ivect,
                        int maxneigh, const int *nn, const int *jn, const int *vn,
                        long long counts[2])
{
    const double sigma = MW_SIGMA, eps = MW_EPSILON;
    const double rcsq = sigma * SW_A * sigma * SW_A;         /* :432 */
    const double sig_a = sigma * SW_A;
    const double Aeps = SW_BIGA * eps, lam_eps = MW_LAMBDA * eps, gam_sig = SW_GAMMA * sigma;
    double Evdw = 0.0;
    long long npair = 0, ntrip = 0;

    for (int i = 0; i < n; ++i) {                            /* :438 */
        const double *ri = xyz + 3*i;
        const int *jl = jn + (size_t)maxneigh*i, *vl = vn + (size_t)maxneigh*i;
        for (int ln = 0; ln < nn[i]; ++ln) {                 /* :442 */
            const double *rj = xyz + 3*(jl[ln] - 1);
            const double *iv = ivect + 3*(vl[ln] - 1);
            double a0 = (rj[0] + iv[0]) - ri[0];             /* :447,450 */
            double a1 = (rj[1] + iv[1]) - ri[1];
            double a2 = (rj[2] + iv[2]) - ri[2];
            double r2_ij = a0*a0 + a1*a1 + a2*a2;            /* :451 */
            if (r2_ij < rcsq) {                              /* :454 */
                double r1_ij = sqrt(r2_ij);                  /* :456 */
                double exp2 = exp(sigma / (r1_ij - sig_a));  /* :459 */
                double q = sigma * sigma / r2_ij;
                double tmpE = Aeps * (SW_B * (q*q) - 1.0);   /* :460 */
                tmpE = tmpE * exp2;                          /* :461 */
                exp2 = exp(gam_sig / (r1_ij - sig_a));       /* :462 */
                Evdw = Evdw + 0.5 * tmpE;                    /* :464 */
                ++npair;
                for (int ln2 = ln + 1; ln2 < nn[i]; ++ln2) { /* :467 */
                    const double *rk = xyz + 3*(jl[ln2] - 1);
                    const double *kv = ivect + 3*(vl[ln2] - 1);
                    double b0 = (rk[0] + kv[0]) - ri[0];     /* :472,474 */
                    double b1 = (rk[1] + kv[1]) - ri[1];
                    double b2 = (rk[2] + kv[2]) - ri[2];
                    double r2_ik = b0*b0 + b1*b1 + b2*b2;    /* :475 */
                    if (r2_ik < rcsq) {                      /* :477 */
                        double r1_ik = sqrt(r2_ik);
                        double ctheta = (a0*b0 + a1*b1 + a2*b2) / (r1_ik * r1_ij);  /* :480 */
                        double d = ctheta - COS0;
                        double csq = d * d;                  /* :481 */
                        double exp1 = exp(gam_sig / (r1_ik - sig_a));               /* :482 */
                        Evdw = Evdw + lam_eps * exp1 * exp2 * csq;                  /* :483 */
                        ++ntrip;
                    }
                }
            }
        }
    }
    if (counts) { counts[0] = npair; counts[1] = ntrip; }
    return Evdw;                                             /* :495 */
}

/* molint.F90:220-404 */
double mwo_local_energy(int imol, int n, const double *xyz, const double *ivect,
                        int maxneigh, const int *nn, const int *jn, const int *vn,
                        long long counts[2])
{
    (void)n;
    const double sigma = MW_SIGMA, eps = MW_EPSILON;
    const double rcsq = sigma * SW_A * sigma * SW_A;         /* :255 */
    const double sig_a = sigma * SW_A;
    const double Aeps = SW_BIGA * eps, gam_sig = SW_GAMMA * sigma;
    double sqlist[2*MWO_MAXNEIGH_HARD], cthetalist[2*MWO_MAXNEIGH_HARD];
    double Evdw = 0.0, Etb = 0.0;
    long long npair = 0, ntrip = 0;

    const int i = imol - 1;
    const double *ri = xyz + 3*i;                            /* :258 */
    const int nni = nn[i];
    const int *jl = jn + (size_t)maxneigh*i, *vl = vn + (size_t)maxneigh*i;

    for (int ln = 0; ln < nni; ++ln) {                       /* :260 */
        double iEtb = 0.0;
        const int j = jl[ln] - 1;                            /* :264-265 */
        const double *jiv = ivect + 3*(vl[ln] - 1);          /* :268 */
        const double *rj0 = xyz + 3*j;
        double rj[3] = { rj0[0] + jiv[0], rj0[1] + jiv[1], rj0[2] + jiv[2] };  /* :269 */
        double a0 = rj[0] - ri[0], a1 = rj[1] - ri[1], a2 = rj[2] - ri[2];     /* :272 */
        double r2_ij = a0*a0 + a1*a1 + a2*a2;                /* :273 */

        if (r2_ij < rcsq) {                                  /* :276 */
            double ir1_ij = 1.0 / sqrt(r2_ij);               /* :278 */
            double r1_ij = ir1_ij * r2_ij;                   /* :286 */
            double isr1_ij = 1.0 / (r1_ij - sig_a);          /* :288 */
            double exp2 = exp(sigma * isr1_ij);              /* :291 */
            double exp3 = exp(gam_sig * isr1_ij);            /* :292 */
            double q = sigma * sigma * ir1_ij * ir1_ij;
            double tmpE = Aeps * (SW_B * (q*q) - 1.0);       /* :294 */
            tmpE = tmpE * exp2;                              /* :295 */
            Evdw = Evdw + tmpE;                              /* :297 */
            ++npair;

            int vlen = 0;
            /* j--i--k: the remaining entries of imol's own list, :302-318 */
            for (int ln2 = ln + 1; ln2 < nni; ++ln2) {
                const double *rk0 = xyz + 3*(jl[ln2] - 1);
                const double *kv = ivect + 3*(vl[ln2] - 1);
                double b0 = (rk0[0] + kv[0]) - ri[0];        /* :307,309 */
                double b1 = (rk0[1] + kv[1]) - ri[1];
                double b2 = (rk0[2] + kv[2]) - ri[2];
                int s = ln2 - ln - 1;                        /* :312 (0-based) */
                sqlist[s] = b0*b0 + b1*b1 + b2*b2;           /* :310,314 */
                cthetalist[s] = (a0*b0 + a1*b1 + a2*b2) * ir1_ij;   /* :316 */
            }
            /* i--j--k: every entry of jmol's list, shifted by j's image, :320-343 */
            double c0 = -a0, c1 = -a1, c2 = -a2;             /* :320 */
            const int nnj = nn[j];
            const int *jl2 = jn + (size_t)maxneigh*j, *vl2 = vn + (size_t)maxneigh*j;
            for (int ln2 = 0; ln2 < nnj; ++ln2) {            /* :324 */
                const double *rk0 = xyz + 3*(jl2[ln2] - 1);
                const double *kv = ivect + 3*(vl2[ln2] - 1);
                double b0 = ((rk0[0] + kv[0]) + jiv[0]) - rj[0];   /* :332,334 */
                double b1 = ((rk0[1] + kv[1]) + jiv[1]) - rj[1];
                double b2 = ((rk0[2] + kv[2]) + jiv[2]) - rj[2];
                int s = nni - (ln + 1) + ln2;                /* :337 (0-based) */
                sqlist[s] = b0*b0 + b1*b1 + b2*b2;           /* :335,339 */
                cthetalist[s] = (c0*b0 + c1*b1 + c2*b2) * ir1_ij;   /* :341 */
            }
            vlen = nni - (ln + 1) + nnj;                     /* :346 */

            /* :354-386; an out-of-range slot contributes exactly 0 (G2) */
            for (int s = 0; s < vlen; ++s) {
                if (sqlist[s] < rcsq) {                      /* :361 */
                    double vinv = 1.0 / sqrt(sqlist[s]);     /* :355 */
                    double vexp = vinv * sqlist[s] - sig_a;  /* :363 */
                    vexp = gam_sig / vexp;                   /* :364 */
                    double ct = cthetalist[s] * vinv;        /* :365 */
                    if (ct < 0.99) {                         /* :367 */
                        double d = ct - COS0;
                        iEtb = iEtb + (d * d) * exp(vexp);   /* :368,385 */
                        ++ntrip;
                    }
                }
            }
            iEtb = iEtb * exp3;                              /* :387 */
        }
        Etb = Etb + iEtb;                                    /* :392 */
    }
    Evdw = Evdw + MW_LAMBDA * eps * Etb;                     /* :397 */
    if (counts) { counts[0] = npair; counts[1] = ntrip; }
    return Evdw;                                             /* :400 */
}

void mwo_local_energy_all(int n, const double *xyz, const double *ivect,
                          int maxneigh, const int *nn, const int *jn, const int *vn,
                          double *e_out, long long counts[2])
{
    long long c[2], tot[2] = { 0, 0 };
    for (int i = 1; i <= n; ++i) {
        e_out[i - 1] = mwo_local_energy(i, n, xyz, ivect, maxneigh, nn, jn, vn, c);
        tot[0] += c[0]; tot[1] += c[1];
    }
    if (counts) { counts[0] = tot[0]; counts[1] = tot[1]; }
}

void mwo_trial_moves(int nmoves, const int *imol, const double *trial,
                     int n, double *xyz, const double *ivect,
                     int maxneigh, const int *nn, const int *jn, const int *vn,
                     double *e_old, double *e_new)
{
    for (int m = 0; m < nmoves; ++m) {
        double *r = xyz + 3*(imol[m] - 1);
        double keep[3] = { r[0], r[1], r[2] };
        e_old[m] = mwo_local_energy(imol[m], n, xyz, ivect, maxneigh, nn, jn, vn, NULL);   /* mc_moves.F90:1010 */
        r[0] = trial[3*m]; r[1] = trial[3*m+1]; r[2] = trial[3*m+2];                        /* mc_moves.F90:1079 */
        e_new[m] = mwo_local_energy(imol[m], n, xyz, ivect, maxneigh, nn, jn, vn, NULL);   /* mc_moves.F90:1083 */
        r[0] = keep[0]; r[1] = keep[1]; r[2] = keep[2];                                     /* mc_moves.F90:1186 */
    }
}

/* =========================================================================================
 * the translation-move driver (mc_moves.F90:966-1213) with its helpers
 * eta_weight (:893-964) and mu_to_bin (:2187-2215).  Restated from the cited lines.  PINNED against the
 * reference program itself: oracle/_ref/mc_water_ref_rng is the reference's unmodified
 * main/mc_moves/io/... with random_uniform_random interposed at link time (ref_wrap_rng.c) so that it
 * draws this file's Philox stream; tests/test_sweep_pin.py replays its runs (single box; two lattices
 * with interpolated weights) and lands on the same configuration to 1e-10 after ~2000 trial moves.
 * ========================================================================================= */
#include <stdint.h>

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

static double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);   /* [0,1) */
}

void mwo_move_uniforms8(uint64_t seed, uint32_t walker, uint64_t move, double u[8])
{
    for (uint32_t call = 0; call < 4; ++call) {
        uint32_t c[4] = { (uint32_t)move, (uint32_t)(move >> 32), walker, call };
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[2 * call]     = u53(c[0], c[1]);
        u[2 * call + 1] = u53(c[2], c[3]);
    }
}

void mwo_move_uniforms(uint64_t seed, uint32_t walker, uint64_t move, double u[6])
{
    for (uint32_t call = 0; call < 3; ++call) {
        uint32_t c[4] = { (uint32_t)move, (uint32_t)(move >> 32), walker, call };
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        u[2 * call]     = u53(c[0], c[1]);
        u[2 * call + 1] = u53(c[2], c[3]);
    }
}

/* r**n for integer n by repeated multiplication, as Fortran evaluates r_pos**Ns / r_pos**k */
static double ipow(double r, int n)
{
    double acc = 1.0, b = r;
    for (int e = n; e > 0; e >>= 1) { if (e & 1) acc *= b; b *= b; }
    return acc;
}

static double gp_ratio(double a, double s, int Ns)           /* mc_moves.F90:583-596 */
{
    double r = 1.1;
    for (int k = 1; ; ++k) {
        const double tmpsum = a * (1.0 - ipow(r, Ns)) / (1.0 - r);
        const double r_new = r * pow(s / tmpsum, 1.0 / (double)Ns);
        if (fabs(r_new - r) <= 2.0 * 2.220446049250313e-16) break;
        if (k > 1000000) break;
        r = r_new;
    }
    return r;
}

void mwo_mu_grid(int nbins, double mu_min, double mu_max, double *mu_bin, double *binwidth, double gp[4])
{
    const double s_pos = fabs(mu_max) - 0.5, s_neg = fabs(mu_min) - 0.5;     /* :571-572 */
    const double a_pos = 1.0, a_neg = 1.0;
    const int Ns = nbins / 2;
    const double r_pos = gp_ratio(a_pos, s_pos, Ns), r_neg = gp_ratio(a_neg, s_neg, Ns);
    double mu_u = -0.5, mu_l;
    int k = 0;
    for (int ibin = nbins / 2; ibin >= 1; --ibin) {                          /* :625-633 */
        mu_l = mu_u - a_neg * ipow(r_neg, k);
        mu_bin[ibin - 1] = 0.5 * (mu_u + mu_l);
        binwidth[ibin - 1] = mu_u - mu_l;
        mu_u = mu_l; ++k;
    }
    mu_bin[nbins / 2] = 0.0; binwidth[nbins / 2] = 1.0;                      /* :636-637 */
    mu_l = 0.5; k = 0;
    for (int ibin = nbins / 2 + 2; ibin <= nbins; ++ibin) {                  /* :641-649 */
        mu_u = mu_l + a_pos * ipow(r_pos, k);
        mu_bin[ibin - 1] = 0.5 * (mu_u + mu_l);
        binwidth[ibin - 1] = mu_u - mu_l;
        mu_l = mu_u; ++k;
    }
    gp[0] = r_pos; gp[1] = a_pos; gp[2] = r_neg; gp[3] = a_neg;
}

int mwo_mu_to_bin(const mwo_eta *g, double mu)                               /* mc_moves.F90:2187-2215 */
{
    if (fabs(mu) <= 0.5) return g->nbins / 2 + 1;
    if (mu > 0.0) {
        const double arg = 1.0 - (mu - 0.5) * (1.0 - g->r_pos) / g->a_pos;
        return g->nbins / 2 + 2 + (int)(log(arg) / log(g->r_pos));
    }
    const double arg = 1.0 - (fabs(mu) - 0.5) * (1.0 - g->r_neg) / g->a_neg;
    return g->nbins / 2 - (int)(log(arg) / log(g->r_neg));
}

double mwo_eta_weight(const mwo_eta *g, double mu)                           /* mc_moves.F90:893-964 */
{
    if (mu < g->mu_lo) return 1.7976931348623157e308;                        /* huge(1.0_dp), :915-918 */
    if (mu > g->mu_hi) return 1.7976931348623157e308;
    const int k = mwo_mu_to_bin(g, mu);
    const double *w = g->weight - 1, *mb = g->mu_bin - 1, *bw = g->binwidth - 1;   /* 1-based views */
    if (!g->eta_interp) return w[k];
    double gradient;
    if (k == g->start_bin) {                                                 /* :929-933 */
        gradient = 2.0 * (w[k + 1] - w[k]) / (bw[k] + bw[k + 1]);
        return w[k] + (mu - mb[k]) * gradient;
    }
    if (k == g->end_bin) {                                                   /* :935-939 */
        gradient = 2.0 * (w[k] - w[k - 1]) / (bw[k] + bw[k - 1]);
        return w[k] + (mu - mb[k]) * gradient;
    }
    if (mu > mb[k]) {                                                        /* :942-945 */
        gradient = 2.0 * (w[k + 1] - w[k]) / (bw[k] + bw[k + 1]);
        return w[k] + (mu - mb[k]) * gradient;
    }
    gradient = 2.0 * (w[k] - w[k - 1]) / (bw[k] + bw[k - 1]);               /* :947-949 */
    return w[k - 1] + (mu - mb[k - 1]) * gradient;
}

#define HM(m, r, c) ((m)[((c) - 1) * 3 + ((r) - 1)])     /* Fortran (r,c), column-major 3x3 */
void mwo_recipmatrix(const double h[9], double rc[9])                        /* util.f90:43-77 */
{
    HM(rc,1,1) = HM(h,2,2)*HM(h,3,3) - HM(h,2,3)*HM(h,3,2);
    HM(rc,1,2) = HM(h,2,3)*HM(h,3,1) - HM(h,2,1)*HM(h,3,3);
    HM(rc,1,3) = HM(h,2,1)*HM(h,3,2) - HM(h,2,2)*HM(h,3,1);
    HM(rc,2,1) = HM(h,1,3)*HM(h,3,2) - HM(h,1,2)*HM(h,3,3);
    HM(rc,2,2) = HM(h,1,1)*HM(h,3,3) - HM(h,1,3)*HM(h,3,1);
    HM(rc,2,3) = HM(h,1,2)*HM(h,3,1) - HM(h,1,1)*HM(h,3,2);
    HM(rc,3,1) = HM(h,1,2)*HM(h,2,3) - HM(h,1,3)*HM(h,2,2);
    HM(rc,3,2) = HM(h,1,3)*HM(h,2,1) - HM(h,1,1)*HM(h,2,3);
    HM(rc,3,3) = HM(h,1,1)*HM(h,2,2) - HM(h,1,2)*HM(h,2,1);
    const double vol = HM(h,1,1)*HM(rc,1,1) + HM(h,1,2)*HM(rc,1,2) + HM(h,1,3)*HM(rc,1,3);
    const double Pi = 3.141592653589793238462643383279502884197;
    for (int i = 0; i < 9; ++i) rc[i] = rc[i] * 2.0 * Pi / vol;
}

/* ---- run options the reference keeps in module variables (userparams / mc_moves), set by the tests ------------------
 * leshift (userparams.f90:41; main.f90:146-150,173): g_dref = ref_enthalpy(1) - ref_enthalpy(2), 0 = off.
 * wl_swetnam (mc_moves.F90:1636-1653): increment recomputed from the histogram after every recorded move.
 * parallel_strategy = 'dd' (mc_moves.F90:181-210,243-248,872,913): walker_in_window and the equilibration rules. */
static double g_dref = 0.0;
static int g_swetnam = 0;
static double g_wl_alpha = 1.0, g_orig_wl_factor = 0.0, g_mu_min = 0.0, g_mu_max = 0.0, g_sumhist = 0.0, g_wl_factor_now = 0.0;
static int g_dd = 0, g_eq_cycles = 0, g_in_window = 1, g_not_in_window_at_eq = 0;

/* -DMINU (mc_moves.F90:1119-1140,1168-1170,1385-1401,1426-1429; a compile-time variant of the reference, off in its
 * examples): every accepted translation or volume move also moves the walker to the lattice of lower enthalpy. */
static int g_minu = 0, g_minu_lsn = 0;
static double g_ref1 = 0.0, g_ref2 = 0.0;
void mwo_set_leshift(double ref1, double ref2) { g_dref = ref1 - ref2; g_ref1 = ref1; g_ref2 = ref2; }
void mwo_set_minu(int on) { g_minu = on; }
/* the MINU branch shared by both move types: returns the lattice the move would end in, rewrites diffkT if it differs */
static int minu_branch(int ls, const double *E, const double *V, const double *E_ls_backup, const double *V_ls_old,
                       int npt_terms, int n, double beta, double pressure, double new_eta, double old_eta, double *diffkT)
{
    const double h1 = E[0] + pressure * V[0] - g_ref1, h2 = E[1] + pressure * V[1] - g_ref2;   /* minloc, :1122-1126 (refs 0 without leshift) */
    const int lsn = h2 < h1 ? 2 : 1;
    if (lsn != ls) {
        double d;
        if (npt_terms)                                                        /* :1131-1133, :1396-1397 */
            d = beta * E[lsn - 1] - beta * *E_ls_backup + beta * pressure * (V[lsn - 1] - *V_ls_old)
                - (double)n * log(V[lsn - 1] / *V_ls_old) + new_eta - old_eta;
        else                                                                  /* :1135 */
            d = beta * E[lsn - 1] - beta * *E_ls_backup + new_eta - old_eta;
        if (g_ref1 != 0.0 || g_ref2 != 0.0)                                   /* leshift, :1134,1136,1398 */
            d = d - beta * (lsn == 1 ? g_ref1 : g_ref2) + beta * (ls == 1 ? g_ref1 : g_ref2);
        *diffkT = d;
    }
    return lsn;
}
void mwo_set_swetnam(int on, double alpha, double orig_wl_factor, double mu_min, double mu_max, double sumhist)
{
    g_swetnam = on; g_wl_alpha = alpha; g_orig_wl_factor = orig_wl_factor; g_mu_min = mu_min; g_mu_max = mu_max; g_sumhist = sumhist;
}
void mwo_get_swetnam(double *sumhist, double *wl_factor) { *sumhist = g_sumhist; *wl_factor = g_wl_factor_now; }
void mwo_set_dd(int on, int eq_cycles, int in_window) { g_dd = on; g_eq_cycles = eq_cycles; g_in_window = on ? in_window : 1; g_not_in_window_at_eq = 0; }
void mwo_get_dd(int *in_window, int *failed) { *in_window = g_in_window; *failed = g_not_in_window_at_eq; }

/* eta_weight with the 'dd' rule of :913 in front: a walker that has not reached its window carries no weight (the
 * reference returns there WITHOUT assigning the function result; 0 is what its comment asks for) */
static double eta_w(const mwo_eta *g, double mu) { return g_in_window ? mwo_eta_weight(g, mu) : 0.0; }

/* mc_update_wl_bins, mc_moves.F90:1597-1689 */
static void update_wl_bins(const mwo_eta *eta, const mwo_cycle_opts *o, double ls_mu,
                           double *histogram, double *unbiased_hist, double *weight)
{
    if (!o->record) return;                                                   /* :1614 */
    const int k = mwo_mu_to_bin(eta, ls_mu);                                  /* :1616 */
    if (k < 1 || k > eta->nbins) return;                                      /* :1619 */
    histogram[k - 1] = histogram[k - 1] + o->av_binwidth / eta->binwidth[k - 1];          /* :1621 */
    if (o->samplerun) {                                                       /* :1625-1631 */
        const double incr = o->av_binwidth / eta->binwidth[k - 1];
        unbiased_hist[k - 1] = unbiased_hist[k - 1] + incr * exp(eta_w(eta, ls_mu) - o->log_unbiased_norm);
        return;
    }
    double wl_factor = o->wl_factor;
    if (g_swetnam) {                                                          /* :1636-1653 */
        g_sumhist = g_sumhist + 1.0;
        wl_factor = 0.0;
        for (int i = 1; i <= eta->nbins; ++i) {
            const double binfrac = eta->binwidth[i - 1] / (g_mu_max - g_mu_min - 1.0);
            const double dev = histogram[i - 1] * eta->binwidth[i - 1] / g_sumhist - binfrac;
            wl_factor = wl_factor + dev * dev;
        }
        wl_factor = sqrt(wl_factor / (double)eta->nbins);
        wl_factor = log(wl_factor);
        wl_factor = wl_factor * g_wl_alpha * (double)eta->nbins;
        wl_factor = wl_factor < g_orig_wl_factor ? wl_factor : g_orig_wl_factor;
        g_wl_factor_now = wl_factor;
    }
    const double incr = wl_factor;                                            /* :1677 */
    weight[k - 1] = weight[k - 1] + o->av_binwidth * incr / eta->binwidth[k - 1];           /* :1680 */
    double minbin = weight[eta->start_bin - 1];                               /* :1682-1685 */
    for (int i = eta->start_bin; i <= eta->end_bin; ++i) if (weight[i - 1] < minbin) minbin = weight[i - 1];
    for (int i = eta->start_bin; i <= eta->end_bin; ++i) weight[i - 1] = weight[i - 1] - minbin;
}

/* mc_lattice_switch, mc_moves.F90:1536-1594; returns 1 if accepted */
static int lattice_switch(const mwo_eta *eta, const mwo_cycle_opts *o, double beta, int n, double x,
                          const double *model_energy, int *ls, double *ls_mu)
{
    const int lsn = 3 - *ls;                                                  /* :1555 */
    const double old_eta = eta_w(eta, *ls_mu), new_eta = eta_w(eta, *ls_mu);                     /* :1557-1558 */
    const double *E = model_energy - 1, *V = o->volume - 1;                   /* 1-based views */
    double diffkT;
    if (o->npt)                                                               /* :1561-1563 */
        diffkT = beta * E[lsn] - beta * E[*ls] + beta * o->pressure * (V[lsn] - V[*ls])
                 - (double)n * log(V[lsn] / V[*ls]) + new_eta - old_eta;
    else                                                                      /* :1568 */
        diffkT = beta * E[lsn] - beta * E[*ls] + new_eta - old_eta;
    diffkT = diffkT + (*ls == 1 ? beta * g_dref : -(beta * g_dref));          /* leshift: - beta ref(lsn) + beta ref(ls), :1567,1572 */
    double compare = exp(-diffkT);
    if (compare > 1.0) compare = 1.0;
    if (x < compare) {                                                        /* :1576-1590 */
        double mu = (E[1] + o->pressure * V[1]) - (E[2] + o->pressure * V[2]);
        mu = mu - g_dref;                                                     /* :1584 */
        mu = mu * beta - (double)n * log(V[1] / V[2]);
        *ls_mu = mu;
        *ls = lsn;
        return 1;
    }
    return 0;
}

/* Top of mc_cycle for the move with global index mg (n moves per cycle; cycles count from 1), mc_moves.F90:181-210:
 * the first move of a cycle re-evaluates walker_in_window.  Returns the cycle number (0 when 'dd' is off). */
static int dd_cycle_top(uint64_t mg, int n, double ls_mu, const mwo_eta *eta)
{
    if (!g_dd) return 0;
    const int cyc = (int)(mg / (uint64_t)n) + 1;
    if (mg % (uint64_t)n == 0) {
        if (cyc < g_eq_cycles) g_in_window = (ls_mu > eta->mu_lo && ls_mu < eta->mu_hi) ? 1 : 0;
        else if (cyc == g_eq_cycles) { if (!g_in_window) g_not_in_window_at_eq = 1; }
        else g_in_window = 1;
    }
    return cyc;
}

static void sweep_impl(int nmoves, uint64_t seed, uint32_t walker, uint64_t move0,
                       int nlat, int n, double *xyz, const double *h,
                       const double *ivect, int ivstride, int maxneigh,
                       const int *nn, const int *jn, const int *vn,
                       double beta, double max_trans, const mwo_eta *eta, const mwo_cycle_opts *opt,
                       double *histogram, double *unbiased_hist, double *weight,
                       int *ls_io, double *ls_mu_io, double *model_energy,
                       long long *accepted, long long *switches, double *log);

void mwo_sweep_translation(int nmoves, uint64_t seed, uint32_t walker, uint64_t move0,
                           int nlat, int n, double *xyz, const double *h,
                           const double *ivect, int ivstride, int maxneigh,
                           const int *nn, const int *jn, const int *vn,
                           double beta, double max_trans, const mwo_eta *eta,
                           int *ls_io, double *ls_mu_io, double *model_energy,
                           long long *accepted, double *log)
{
    sweep_impl(nmoves, seed, walker, move0, nlat, n, xyz, h, ivect, ivstride, maxneigh, nn, jn, vn, beta, max_trans,
               eta, NULL, NULL, NULL, NULL, ls_io, ls_mu_io, model_energy, accepted, NULL, log);
}

void mwo_sweep_cycle(int nmoves, uint64_t seed, uint32_t walker, uint64_t move0,
                     int nlat, int n, double *xyz, const double *h,
                     const double *ivect, int ivstride, int maxneigh,
                     const int *nn, const int *jn, const int *vn,
                     double beta, double max_trans, const mwo_eta *eta, const mwo_cycle_opts *opt,
                     double *histogram, double *unbiased_hist, double *weight,
                     int *ls_io, double *ls_mu_io, double *model_energy,
                     long long *accepted, long long *switches, double *log)
{
    sweep_impl(nmoves, seed, walker, move0, nlat, n, xyz, h, ivect, ivstride, maxneigh, nn, jn, vn, beta, max_trans,
               eta, opt, histogram, unbiased_hist, weight, ls_io, ls_mu_io, model_energy, accepted, switches, log);
}

static void sweep_impl(int nmoves, uint64_t seed, uint32_t walker, uint64_t move0,
                       int nlat, int n, double *xyz, const double *h,
                       const double *ivect, int ivstride, int maxneigh,
                       const int *nn, const int *jn, const int *vn,
                       double beta, double max_trans, const mwo_eta *eta, const mwo_cycle_opts *opt,
                       double *histogram, double *unbiased_hist, double *weight,
                       int *ls_io, double *ls_mu_io, double *model_energy,
                       long long *accepted, long long *switches, double *log)
{
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;     /* constants.f90:23 */
    double recip[2][9];
    for (int l = 0; l < nlat; ++l) mwo_recipmatrix(h + 9 * l, recip[l]);
    int ls = *ls_io;                                                          /* 1-based */
    double ls_mu = *ls_mu_io;
    long long acc = 0;
#define LAT(l, arr, stride) ((arr) + (size_t)(l) * (stride))
    for (int mv = 0; mv < nmoves; ++mv) {
        double u[8];
        mwo_move_uniforms8(seed, walker, move0 + (uint64_t)mv, u);            /* u[0..5] as mwo_move_uniforms */
        const int cyc = dd_cycle_top(move0 + (uint64_t)mv, n, ls_mu, eta);
        const int lsn = nlat == 2 ? 3 - ls : 1;                               /* partner_lattice, :866-868 */
        int imol = (int)(u[0] * (double)n) + 1;                               /* :1001-1002 */
        if (imol > n) imol = n;
        double old_e[2] = {0, 0}, new_e[2] = {0, 0}, backup[2] = {0, 0}, deltaE[2] = {0, 0};
        for (int l = 0; l < nlat; ++l) {                                      /* :1007-1018 */
            old_e[l] = mwo_local_energy(imol, n, LAT(l, xyz, 3 * n), LAT(l, ivect, 3 * ivstride), maxneigh,
                                        LAT(l, nn, n), LAT(l, jn, (size_t)n * maxneigh), LAT(l, vn, (size_t)n * maxneigh), NULL);
            backup[l] = model_energy[l];
            model_energy[l] = model_energy[l] - old_e[l];
        }
        double x = 2.0 * u[1] - 1.0, y = 2.0 * u[2] - 1.0, z = 2.0 * u[3] - 1.0;   /* :1021-1027 */
        const double norm = 1.0 / sqrt(x * x + y * y + z * z);                /* :1029 */
        x = x * norm; y = y * norm; z = z * norm;
        const double r = u[4] * 2.0 - 1.0;                                    /* :1035 */
        x = x * max_trans * r; y = y * max_trans * r; z = z * max_trans * r;  /* :1037-1039 */
        const double *rc = recip[ls - 1];
        double sx = HM(rc,1,1) * x + HM(rc,2,1) * y + HM(rc,3,1) * z;         /* :1042-1050 */
        double sy = HM(rc,1,2) * x + HM(rc,2,2) * y + HM(rc,3,2) * z;
        double sz = HM(rc,1,3) * x + HM(rc,2,3) * y + HM(rc,3,3) * z;
        sx = sx * 0.5 * invPi; sy = sy * 0.5 * invPi; sz = sz * 0.5 * invPi;  /* :1052-1054 */
        double transvec[2][3];
        transvec[ls - 1][0] = x; transvec[ls - 1][1] = y; transvec[ls - 1][2] = z;
        if (nlat == 2) {                                                      /* :1061-1067 */
            const double *hn = h + 9 * (lsn - 1);
            for (int d = 1; d <= 3; ++d)
                transvec[lsn - 1][d - 1] = HM(hn,d,1) * sx + HM(hn,d,2) * sy + HM(hn,d,3) * sz;
        }
        for (int l = 0; l < nlat; ++l) {                                      /* :1076-1092 */
            double *p = LAT(l, xyz, 3 * n) + 3 * (imol - 1);
            p[0] += transvec[l][0]; p[1] += transvec[l][1]; p[2] += transvec[l][2];
            new_e[l] = mwo_local_energy(imol, n, LAT(l, xyz, 3 * n), LAT(l, ivect, 3 * ivstride), maxneigh,
                                        LAT(l, nn, n), LAT(l, jn, (size_t)n * maxneigh), LAT(l, vn, (size_t)n * maxneigh), NULL);
            model_energy[l] = model_energy[l] + new_e[l];
            deltaE[l] = new_e[l] - old_e[l];
        }
        double diffkT;
        int minu_ls = ls;
        if (nlat == 1) {
            diffkT = beta * deltaE[0];                                        /* :1106 */
        } else {
            const double eta_old = eta_w(eta, ls_mu);                         /* :1112-1116 */
            ls_mu = ls_mu + (deltaE[0] - deltaE[1]) * beta;
            const double eta_new = eta_w(eta, ls_mu);
            diffkT = deltaE[ls - 1] * beta + eta_new - eta_old;
            if (g_minu)                                                       /* :1119-1140 */
                minu_ls = minu_branch(ls, model_energy, opt->volume, &backup[ls - 1], &opt->volume[ls - 1], opt->npt, n, beta,
                                      opt->pressure, eta_new, eta_old, &diffkT);
        }
        const double zeta = u[5];                                             /* :1145 */
        double pacc = exp(-diffkT);
        if (pacc > 1.0) pacc = 1.0;
        const int ok = zeta < pacc;                                           /* :1146 (false for NaN) */
        if (ok) {
            ++acc;
            ls = minu_ls;                                                     /* :1168-1170 */
        } else {                                                              /* :1182-1195 */
            for (int l = 0; l < nlat; ++l) {
                double *p = LAT(l, xyz, 3 * n) + 3 * (imol - 1);
                p[0] -= transvec[l][0]; p[1] -= transvec[l][1]; p[2] -= transvec[l][2];
                model_energy[l] = backup[l];
            }
            if (nlat == 2) ls_mu = ls_mu - (deltaE[0] - deltaE[1]) * beta;
        }
        int sw = 0;
        if (opt && nlat == 2) {
            update_wl_bins(eta, opt, ls_mu, histogram, unbiased_hist, weight);                 /* mc_moves.F90:230 */
            if (opt->always_switch && !(g_dd && cyc < g_eq_cycles)) {                         /* :243-248 */
                sw = lattice_switch(eta, opt, beta, n, u[6], model_energy, &ls, &ls_mu);
                if (sw && switches) ++*switches;
            }
        }
        if (log) {
            double *q = log + 8 * (size_t)mv;
            q[0] = imol; q[1] = ok + 2 * sw; q[2] = old_e[0]; q[3] = new_e[0]; q[4] = old_e[1]; q[5] = new_e[1]; q[6] = ls_mu; q[7] = diffkT;
        }
    }
#undef LAT
    *ls_io = ls; *ls_mu_io = ls_mu; *accepted += acc;
}


/* ------------------------------------------------------------------------------------------------
 * mc_volume, mc_moves.F90:1216-1534 (MINU off; ref_ljr, which only chain synchronisation
 * reads, is not carried).
 * ------------------------------------------------------------------------------------------------ */
static double det3(const double *m)                                           /* util.f90:16-41 */
{
    double det = HM(m,1,1) * (HM(m,2,2) * HM(m,3,3) - HM(m,2,3) * HM(m,3,2));
    det = det - HM(m,1,2) * (HM(m,2,1) * HM(m,3,3) - HM(m,2,3) * HM(m,3,1));
    det = det + HM(m,1,3) * (HM(m,2,1) * HM(m,3,2) - HM(m,2,2) * HM(m,3,1));
    return det;
}

/* positions -> fractional coordinates with `recip`, back to Cartesian with `hnew`; ljr += (that - ljr), :1288-1316 */
static void rescale_positions(int n, double *xyz, const double *recip, const double *hnew)
{
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;
    for (int i = 0; i < n; ++i) {
        double *p = xyz + 3 * i;
        const double o0 = p[0], o1 = p[1], o2 = p[2];
        double s0 = HM(recip,1,1) * o0 + HM(recip,2,1) * o1 + HM(recip,3,1) * o2;
        double s1 = HM(recip,1,2) * o0 + HM(recip,2,2) * o1 + HM(recip,3,2) * o2;
        double s2 = HM(recip,1,3) * o0 + HM(recip,2,3) * o1 + HM(recip,3,3) * o2;
        s0 = s0 * 0.5 * invPi; s1 = s1 * 0.5 * invPi; s2 = s2 * 0.5 * invPi;
        double t[3];
        for (int d = 1; d <= 3; ++d) t[d - 1] = HM(hnew,d,1) * s0 + HM(hnew,d,2) * s1 + HM(hnew,d,3) * s2;
        t[0] = t[0] - o0; t[1] = t[1] - o1; t[2] = t[2] - o2;
        p[0] = p[0] + t[0]; p[1] = p[1] + t[1]; p[2] = p[2] + t[2];
    }
}

/* reference positions carried through volume moves (mc_moves.F90:1318-1349, 1461-1492); NULL = not carried */
static double *g_ref_xyz = NULL;

int mwo_volume_move(const double u[4], int nlat, int n, double *xyz, double *h, double *volume,
                    double *ivect, int ivstride, int *nivect, int maxneigh,
                    const int *nn, const int *jn, const int *vn,
                    double beta, double dv_max, double pressure, const mwo_eta *eta,
                    int ls, double *ls_mu, double *model_energy)
{
    double backup_e[2], old_e[2], new_e[2], deltaE[2] = {0, 0}, old_vol[2], old_h[2][9], old_recip[2][9], recip[2][9];
    for (int l = 0; l < nlat; ++l) {                                          /* :1243-1262 */
        backup_e[l] = model_energy[l]; old_e[l] = model_energy[l];
        mwo_recipmatrix(h + 9 * l, recip[l]);
        for (int t = 0; t < 9; ++t) { old_h[l][t] = h[9 * l + t]; old_recip[l][t] = recip[l][t]; }
        old_vol[l] = volume[l];
    }
    const int idim = (int)(u[0] * 3.0) + 1;                                   /* :1269-1272 */
    const int jdim = (int)(u[1] * 3.0) + 1;
    double delta[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    HM(delta, idim, jdim) = (2.0 * u[2] - 1.0) * dv_max;                      /* :1276-1278 */
    HM(delta, jdim, idim) = HM(delta, idim, jdim);
    for (int l = 0; l < nlat; ++l) for (int t = 0; t < 9; ++t) h[9 * l + t] = h[9 * l + t] + delta[t];   /* :1281-1282 */
    for (int l = 0; l < nlat; ++l) {                                          /* :1285-1358 */
        rescale_positions(n, xyz + (size_t)3 * n * l, recip[l], h + 9 * l);
        if (g_ref_xyz) rescale_positions(n, g_ref_xyz + (size_t)3 * n * l, recip[l], h + 9 * l);      /* :1318-1349 */
        volume[l] = fabs(det3(h + 9 * l));
        mwo_recipmatrix(h + 9 * l, recip[l]);
        const int niv = mwo_compute_ivects(h + 9 * l, ivect + (size_t)3 * ivstride * l, ivstride);
        if (niv < 0) return -1;
        nivect[l] = niv;
        new_e[l] = mwo_model_energy(n, xyz + (size_t)3 * n * l, ivect + (size_t)3 * ivstride * l, maxneigh,
                                    nn + (size_t)n * l, jn + (size_t)n * maxneigh * l, vn + (size_t)n * maxneigh * l, NULL);
        model_energy[l] = new_e[l];
    }
    for (int l = 0; l < nlat; ++l) deltaE[l] = new_e[l] - old_e[l];           /* :1361 */
    double old_eta = 0.0, new_eta = 0.0;
    if (nlat == 2) {                                                          /* :1363-1371 */
        old_eta = eta_w(eta, *ls_mu);
        double mu = (model_energy[0] + pressure * volume[0]) - (model_energy[1] + pressure * volume[1]);
        mu = mu - g_dref;                                                     /* :1371 */
        mu = mu * beta - (double)n * log(volume[0] / volume[1]);
        *ls_mu = mu;
        new_eta = eta_w(eta, *ls_mu);
    }
    const double x = u[3];                                                    /* :1378 */
    double diffkT = beta * deltaE[ls - 1] + new_eta - old_eta + beta * pressure * (volume[ls - 1] - old_vol[ls - 1])
                    - (double)n * log(volume[ls - 1] / old_vol[ls - 1]);         /* :1381-1382 */
    g_minu_lsn = ls;
    if (g_minu && nlat == 2)                                                  /* :1385-1401 */
        g_minu_lsn = minu_branch(ls, model_energy, volume, &backup_e[ls - 1], &old_vol[ls - 1], 1, n, beta, pressure,
                                 new_eta, old_eta, &diffkT);
    double compare = exp(-diffkT);
    if (compare > 1.0) compare = 1.0;
    if (x < compare) return 1;                                                /* :1410 (with MINU the caller takes g_minu_lsn, :1426-1429) */
    g_minu_lsn = ls;
    /* rejected, :1426-1530 */
    for (int l = 0; l < nlat; ++l) {
        volume[l] = old_vol[l];
        for (int t = 0; t < 9; ++t) h[9 * l + t] = old_h[l][t];
    }
    for (int l = 0; l < nlat; ++l) {
        rescale_positions(n, xyz + (size_t)3 * n * l, recip[l], h + 9 * l);   /* recip is the NEW one here */
        if (g_ref_xyz) rescale_positions(n, g_ref_xyz + (size_t)3 * n * l, recip[l], h + 9 * l);      /* :1461-1492 */
    }
    for (int l = 0; l < nlat; ++l) {
        const int niv = mwo_compute_ivects(h + 9 * l, ivect + (size_t)3 * ivstride * l, ivstride);       /* :1510-1512 */
        if (niv < 0) return -1;
        nivect[l] = niv;
        model_energy[l] = backup_e[l];                                        /* :1514 */
    }
    if (nlat == 2) {                                                          /* :1516-1520 */
        double mu = (model_energy[0] + pressure * volume[0]) - (model_energy[1] + pressure * volume[1]);
        mu = mu - g_dref;                                                     /* :1526 */
        mu = mu * beta - (double)n * log(volume[0] / volume[1]);
        *ls_mu = mu;
    }
    (void)old_recip;
    return 0;
}

void mwo_sweep_full(int nmoves, uint64_t seed, uint32_t walker, uint64_t move0, double transP, double dv_max,
                    int nlat, int n, double *xyz, double *h, double *volume,
                    double *ivect, int ivstride, int *nivect, int maxneigh,
                    const int *nn, const int *jn, const int *vn,
                    double beta, double max_trans, const mwo_eta *eta, mwo_cycle_opts *opt,
                    double *histogram, double *unbiased_hist, double *weight,
                    int *ls, double *ls_mu, double *model_energy,
                    long long *accepted, long long *switches, long long *nvol, double *log)
{
    for (int mv = 0; mv < nmoves; ++mv) {
        double u[8];
        mwo_move_uniforms8(seed, walker, move0 + (uint64_t)mv, u);
        if (u[7] < transP) {                                                  /* mc_moves.F90:226-231 */
            opt->volume[0] = volume[0]; if (nlat == 2) opt->volume[1] = volume[1];
            sweep_impl(1, seed, walker, move0 + (uint64_t)mv, nlat, n, xyz, h, ivect, ivstride, maxneigh, nn, jn, vn,
                       beta, max_trans, eta, opt, histogram, unbiased_hist, weight, ls, ls_mu, model_energy,
                       accepted, switches, log ? log + 8 * (size_t)mv : NULL);
        } else {                                                              /* :232-235 */
            const int cyc = dd_cycle_top(move0 + (uint64_t)mv, n, *ls_mu, eta);
            const int ok = mwo_volume_move(u, nlat, n, xyz, h, volume, ivect, ivstride, nivect, maxneigh, nn, jn, vn,
                                           beta, dv_max, opt->pressure, eta, *ls, ls_mu, model_energy);
            if (nvol) { ++nvol[0]; if (ok == 1) ++nvol[1]; }
            if (ok == 1 && g_minu && nlat == 2) *ls = g_minu_lsn;                                      /* :1426-1429 */
            int sw = 0;
            if (nlat == 2) {
                opt->volume[0] = volume[0]; opt->volume[1] = volume[1];
                update_wl_bins(eta, opt, *ls_mu, histogram, unbiased_hist, weight);                    /* :234 */
                if (opt->always_switch && !(g_dd && cyc < g_eq_cycles)) {                              /* :243-248 */
                    sw = lattice_switch(eta, opt, beta, n, u[6], model_energy, ls, ls_mu);
                    if (sw && switches) ++*switches;
                }
            }
            if (log) {
                double *q = log + 8 * (size_t)mv;
                q[0] = 0.0; q[1] = 4 + (ok == 1) + 2 * sw; q[2] = model_energy[0]; q[3] = volume[0];
                q[4] = nlat == 2 ? model_energy[1] : 0.0; q[5] = nlat == 2 ? volume[1] : 0.0; q[6] = *ls_mu; q[7] = 0.0;
            }
        }
    }
}


void mwo_sweep_full_ref(int nmoves, uint64_t seed, uint32_t walker, uint64_t move0, double transP, double dv_max,
                        int nlat, int n, double *xyz, double *ref_xyz, double *h, double *volume,
                        double *ivect, int ivstride, int *nivect, int maxneigh,
                        const int *nn, const int *jn, const int *vn,
                        double beta, double max_trans, const mwo_eta *eta, mwo_cycle_opts *opt,
                        double *histogram, double *unbiased_hist, double *weight,
                        int *ls, double *ls_mu, double *model_energy,
                        long long *accepted, long long *switches, long long *nvol, double *log)
{
    g_ref_xyz = ref_xyz;
    mwo_sweep_full(nmoves, seed, walker, move0, transP, dv_max, nlat, n, xyz, h, volume, ivect, ivstride, nivect, maxneigh,
                   nn, jn, vn, beta, max_trans, eta, opt, histogram, unbiased_hist, weight, ls, ls_mu, model_energy,
                   accepted, switches, nvol, log);
    g_ref_xyz = NULL;
}

/* mc_check_chain_synchronisation, mc_moves.F90:2217-2416 */
int mwo_chain_sync(int n, double *xyz, const double *ref_xyz, double *h, const double *ref_h, double *volume,
                   double *ivect, int ivstride, int *nivect, int maxneigh,
                   const int *nn, const int *jn, const int *vn,
                   double beta, double pressure, double *ls_mu, double *model_energy)
{
    const double invPi = 1.0 / 3.141592653589793238462643383279502884197;
    double hmat_diff[2][9], recip[2][9];
    /* :2248-2259: energies and ls_mu of the state as it is (ls_mu is overwritten again at the end) */
    for (int l = 0; l < 2; ++l)
        model_energy[l] = mwo_model_energy(n, xyz + (size_t)3 * n * l, ivect + (size_t)3 * ivstride * l, maxneigh,
                                           nn + (size_t)n * l, jn + (size_t)n * maxneigh * l, vn + (size_t)n * maxneigh * l, NULL);
    for (int l = 0; l < 2; ++l) for (int t = 0; t < 9; ++t) hmat_diff[l][t] = h[9 * l + t] - ref_h[9 * l + t];   /* :2261-2263 */
    for (int t = 0; t < 9; ++t) h[9 + t] = ref_h[9 + t] + hmat_diff[0][t];                                        /* :2277 */
    mwo_recipmatrix(h, recip[0]);                                                                                 /* :2279-2280 */
    mwo_recipmatrix(h + 9, recip[1]);
    for (int i = 0; i < n; ++i) {                                                                                 /* :2289-2343 */
        double svect[2][3], ref_svect[2][3], spos_diff0[3];
        for (int l = 0; l < 2; ++l) {
            const double *p = xyz + (size_t)3 * n * l + 3 * i, *q = ref_xyz + (size_t)3 * n * l + 3 * i;
            const double *rc = recip[l];
            svect[l][0] = (HM(rc,1,1) * p[0] + HM(rc,2,1) * p[1] + HM(rc,3,1) * p[2]) * 0.5 * invPi;
            svect[l][1] = (HM(rc,1,2) * p[0] + HM(rc,2,2) * p[1] + HM(rc,3,2) * p[2]) * 0.5 * invPi;
            svect[l][2] = (HM(rc,1,3) * p[0] + HM(rc,2,3) * p[1] + HM(rc,3,3) * p[2]) * 0.5 * invPi;
            ref_svect[l][0] = (HM(rc,1,1) * q[0] + HM(rc,2,1) * q[1] + HM(rc,3,1) * q[2]) * 0.5 * invPi;
            ref_svect[l][1] = (HM(rc,1,2) * q[0] + HM(rc,2,2) * q[1] + HM(rc,3,2) * q[2]) * 0.5 * invPi;
            ref_svect[l][2] = (HM(rc,1,3) * q[0] + HM(rc,2,3) * q[1] + HM(rc,3,3) * q[2]) * 0.5 * invPi;
            if (l == 0) for (int d = 0; d < 3; ++d) spos_diff0[d] = svect[0][d] - ref_svect[0][d];
        }
        double s2[3];
        for (int d = 0; d < 3; ++d) s2[d] = ref_svect[1][d] + spos_diff0[d];                                      /* :2339 */
        double *p2 = xyz + (size_t)3 * n + 3 * i;
        const double *h2 = h + 9;
        for (int d = 1; d <= 3; ++d) p2[d - 1] = HM(h2,d,1) * s2[0] + HM(h2,d,2) * s2[1] + HM(h2,d,3) * s2[2];    /* :2341 matmul */
    }
    for (int l = 0; l < 2; ++l) {                                                                                 /* :2385-2390 */
        volume[l] = fabs(det3(h + 9 * l));
        const int niv = mwo_compute_ivects(h + 9 * l, ivect + (size_t)3 * ivstride * l, ivstride);
        if (niv < 0) return -1;
        nivect[l] = niv;
    }
    for (int l = 0; l < 2; ++l)                                                                                   /* :2395-2396 */
        model_energy[l] = mwo_model_energy(n, xyz + (size_t)3 * n * l, ivect + (size_t)3 * ivstride * l, maxneigh,
                                           nn + (size_t)n * l, jn + (size_t)n * maxneigh * l, vn + (size_t)n * maxneigh * l, NULL);
    double mu = model_energy[0] + pressure * volume[0] - model_energy[1] - pressure * volume[1];                  /* :2398-2401 */
    mu = mu - g_dref;
    *ls_mu = mu * beta - (double)n * log(volume[0] / volume[1]);
    return 0;
}
