/*
 * c3d_oracle.c — CPU restatement (plain C, fp64, single thread) of the Chromosome3D hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (chromosome3d_amd/, libc3d.so,
 * c3d_solve, the Perl driver) may include, link, import or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * What it restates (file:line under /root/reference):
 *   front half   chromosome3D.pl:110-162 (IF2dist_new), :164-179 (calc_len_IF),
 *                :181-206 (dist2rr), :340-362 (carr2tbl)            -> PINNED bit-exact by
 *                tests/golden/<id>.dist|.rr|.contact.tbl + front_half_golden.json (made by
 *                running the reference's own Perl subs, tests/golden/make_golden.pl)
 *   assessment   chromosome3D.pl:447-485, 487-554, 581-600, 716-729  -> PINNED by the known
 *                answers in front_half_golden.json (satisfied n/R, sum of deviations)
 *   metric       spearman_IF_pdb.pl:26-70                            -> restated (its Perl deps
 *                are absent from the reference tree); checked against scipy in tests
 *   solver       the CNS dgsa protocol the deck emits, chromosome3D.pl:1395-1426 (nbonds,
 *                fbeta, mass), :1631-1700 (regularise + hot stages), :1729-1782 (slow cool),
 *                :1790-1803 (final minimisation), :1806-1816 (centre).  CNS itself (cns_solve
 *                1.3, third party, licence-gated, not in the tree, no Fortran here) cannot
 *                run: PARITY UNPINNED for the solver except statistically, through the 46
 *                bundled output_models PDB files (Spearman(IF,d) table in BASELINE.md).
 *
 * Solver model ("bead model", one particle per Hi-C bin):
 *   E = w * [ S * sum_{restrained i<j} softsq(d_ij - t_ij) + k_b * sum_i (d_{i,i+1} - b0)^2 ]
 *       + w_vdw * sum_{|i-j| >= rep_sep} max(0, (s R0)^2 - d_ij^2)^2
 *   MD     : leap-frog, all masses 100 amu, a = 418.4 F / m; Berendsen weak coupling (hot,
 *            1/tau = fbeta = 10 /ps) or hard velocity rescaling (cool) using T of the previous
 *            half step; centre-of-mass velocity removed each step.
 *   minimise: FIRE (Bitzek et al. 2006) instead of CNS's L-BFGS — same stationary points,
 *            one force evaluation per step and no line search (GPU-friendly; deviation
 *            recorded in DESIGN.md).
 *   RNG    : Philox4x32-10 keyed by (seed, replica), counter = (bead, purpose).
 * The HIP path implements exactly this arithmetic in fp32 (reductions in fp32/fp64 as
 * documented); tests compare the two on identical inputs.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define C3O_OK 0
#define C3O_ERR (-1)

/* ------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) — counter-based, identical on host and device.    */
/* ------------------------------------------------------------------------------------ */
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k[0];
    const uint32_t n1 = lo1;
    const uint32_t n2 = hi0 ^ c[3] ^ k[1];
    const uint32_t n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void c3o_philox4x32(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr_in[0], ctr_in[1], ctr_in[2], ctr_in[3]};
    uint32_t k[2] = {key_in[0], key_in[1]};
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(c));
}
/* uniform in (0,1): (u + 0.5) / 2^32 */
static inline double u01(uint32_t u) { return ((double)u + 0.5) * (1.0 / 4294967296.0); }
/* four standard normals for (seed, replica, bead, purpose) via Box-Muller */
static void normals4(uint64_t seed, uint32_t replica, uint32_t bead, uint32_t purpose, double g[4]) {
    uint32_t ctr[4] = {bead, purpose, 0u, 0u};
    uint32_t key[2] = {(uint32_t)(seed & 0xFFFFFFFFu) ^ (replica * 0x9E3779B9u),
                       (uint32_t)(seed >> 32) + replica};
    uint32_t r[4];
    c3o_philox4x32(ctr, key, r);
    const double two_pi = 6.283185307179586476925286766559;
    double a = sqrt(-2.0 * log(u01(r[0]))), b = two_pi * u01(r[1]);
    g[0] = a * cos(b); g[1] = a * sin(b);
    a = sqrt(-2.0 * log(u01(r[2]))); b = two_pi * u01(r[3]);
    g[2] = a * cos(b); g[3] = a * sin(b);
}

/* ------------------------------------------------------------------------------------ */
/* Front half: IF matrix text -> N, distances (%.1f), restraint rows, contact.tbl         */
/* ------------------------------------------------------------------------------------ */

/* chromosome3D.pl:164-179 + :116-129: whitespace-split numbers, N = fields on line 1.
 * Returns N (>0) and a malloc'ed row-major n*n array in *out, or C3O_ERR. */
int c3o_parse_if_text(const char* text, size_t len, double** out) {
    /* first line field count */
    size_t p = 0;
    int n = 0;
    {
        size_t q = 0;
        int in_tok = 0;
        while (q < len && text[q] != '\n') {
            const int ws = (text[q] == ' ' || text[q] == '\t' || text[q] == '\r');
            if (!ws && !in_tok) { ++n; in_tok = 1; }
            if (ws) in_tok = 0;
            ++q;
        }
    }
    if (n < 1) return C3O_ERR;
    double* m = (double*)malloc(sizeof(double) * (size_t)n * n);
    if (!m) return C3O_ERR;
    size_t cnt = 0;
    char* buf = (char*)malloc(len + 1);
    memcpy(buf, text, len);
    buf[len] = 0;
    char* s = buf;
    (void)p;
    while (*s) {
        while (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n') ++s;
        if (!*s) break;
        char* e;
        const double v = strtod(s, &e);
        if (e == s) { free(buf); free(m); return C3O_ERR; }
        if (cnt < (size_t)n * n) m[cnt] = v;
        ++cnt;
        s = e;
    }
    free(buf);
    if (cnt != (size_t)n * n) { free(m); return C3O_ERR; }
    *out = m;
    return n;
}
void c3o_free(void* p) { free(p); }

/* chromosome3D.pl:130-161: P = IF**alpha, running sum row-major, mean over all N*N incl.
 * diagonal, P /= mean, D = K / P (P == 0 -> -1), printed "%.1f".  dist10[i*n+j] receives the
 * printed value in tenths of an Angstrom as an integer (-10 for the -1 sentinel), obtained
 * by formatting with "%.1f" exactly as the reference does and reading the digits back. */
int c3o_if_to_dist10(const double* IF, int n, double alpha, double K, int32_t* dist10) {
    const size_t nn = (size_t)n * n;
    double* P = (double*)malloc(sizeof(double) * nn);
    if (!P) return C3O_ERR;
    double sum = 0.0;
    for (size_t k = 0; k < nn; ++k) { P[k] = pow(IF[k], alpha); sum += P[k]; }
    const double mean = sum / ((double)n * (double)n);
    for (size_t k = 0; k < nn; ++k) {
        double v = P[k] / mean;
        if (v == 0) v = -1; else v = K / v;
        char b[64];
        snprintf(b, sizeof b, "%.1f", v);
        /* parse "[-]ddd.d" into tenths */
        int neg = 0; const char* s = b; if (*s == '-') { neg = 1; ++s; }
        long long ip = 0; while (*s >= '0' && *s <= '9') { ip = ip * 10 + (*s - '0'); ++s; }
        int frac = 0; if (*s == '.') { ++s; frac = *s - '0'; }
        long long t = ip * 10 + frac;
        if (t > 2000000000LL) t = 2000000000LL;   /* inf/huge */
        dist10[k] = (int32_t)(neg ? -t : t);
    }
    free(P);
    return C3O_OK;
}

/* write <ID>.dist exactly as chromosome3D.pl:156-161 */
int c3o_write_dist(const char* path, const int32_t* dist10, int n) {
    FILE* f = fopen(path, "w");
    if (!f) return C3O_ERR;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            const int32_t t = dist10[(size_t)i * n + j];
            const int32_t a = t < 0 ? -t : t;
            fprintf(f, "%s%d.%d ", t < 0 ? "-" : "", a / 10, a % 10);
        }
        fputc('\n', f);
    }
    fclose(f);
    return C3O_OK;
}

typedef struct { int i, j; int32_t t10; char key[24]; } rr_row;
static int rr_cmp(const void* a, const void* b) { return strcmp(((const rr_row*)a)->key, ((const rr_row*)b)->key); }

/* chromosome3D.pl:181-206: i<j, |j-i| >= min_sep, D > 0; rows ordered by Perl `sort keys`
 * (byte-wise string order of "i j", 1-based).  Returns R; *ri,*rj (1-based) and *rt10 are
 * malloc'ed. */
int c3o_dist_to_rr(const int32_t* dist10, int n, int min_sep, int** ri, int** rj, int32_t** rt10) {
    size_t cap = 0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j)
            if (j - i >= min_sep && dist10[(size_t)i * n + j] > 0) ++cap;
    rr_row* rows = (rr_row*)malloc(sizeof(rr_row) * (cap ? cap : 1));
    size_t r = 0;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const int32_t t = dist10[(size_t)i * n + j];
            if (j - i < min_sep || t <= 0) continue;
            rows[r].i = i + 1; rows[r].j = j + 1; rows[r].t10 = t;
            snprintf(rows[r].key, sizeof rows[r].key, "%d %d", i + 1, j + 1);
            ++r;
        }
    qsort(rows, r, sizeof(rr_row), rr_cmp);
    *ri = (int*)malloc(sizeof(int) * (r ? r : 1));
    *rj = (int*)malloc(sizeof(int) * (r ? r : 1));
    *rt10 = (int32_t*)malloc(sizeof(int32_t) * (r ? r : 1));
    for (size_t k = 0; k < r; ++k) { (*ri)[k] = rows[k].i; (*rj)[k] = rows[k].j; (*rt10)[k] = rows[k].t10; }
    free(rows);
    return (int)r;
}

/* <ID>.rr rows "i j %.2f %.2f 1.0" (chromosome3D.pl:204) */
int c3o_write_rr(const char* path, const int* ri, const int* rj, const int32_t* rt10, int R) {
    FILE* f = fopen(path, "w");
    if (!f) return C3O_ERR;
    for (int k = 0; k < R; ++k)
        fprintf(f, "%d %d %d.%d0 %d.%d0 1.0\n", ri[k], rj[k], rt10[k] / 10, rt10[k] % 10, rt10[k] / 10, rt10[k] % 10);
    fclose(f);
    return C3O_OK;
}
/* contact.tbl rows (chromosome3D.pl:352-360): d=(hi+lo)/2, dminus=dplus=(hi-lo)/2=0.00 */
int c3o_write_tbl(const char* path, const int* ri, const int* rj, const int32_t* rt10, int R) {
    FILE* f = fopen(path, "w");
    if (!f) return C3O_ERR;
    for (int k = 0; k < R; ++k)
        fprintf(f, "assign45 (resid %3d and name ca) (resid %3d and name ca) %d.%d0 0.00 0.00\n",
                ri[k], rj[k], rt10[k] / 10, rt10[k] % 10);
    fclose(f);
    return C3O_OK;
}

/* ------------------------------------------------------------------------------------ */
/* Energy model                                                                          */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int n;
    int min_sep;      /* 5    chromosome3D.pl:20,65 */
    int noe_pot;      /* 0 symmetric soft-square (Huber); 1 X-PLOR soft-square (upper side soft,
                         lower side square); 2 pure square */
    int rep_sep;      /* repel acts on |i-j| >= rep_sep */
    int ang_mode;     /* (i,i+2) term: 0 lower bound only (d < a0), 1 harmonic both sides */
    double s_noe;     /* 10   chromosome3D.pl:66,1111,1120 */
    double rswitch;   /* 1.0  CNS readdata default [CNS-UNVERIFIED] */
    double asym;      /* 2.0  asymptote slope      [CNS-UNVERIFIED] */
    double masym;     /* lower-side tail slope for noe_pot 3 (CNS masymptote) [CNS-UNVERIFIED] */
    double mrswitch;  /* lower-side switch distance for noe_pot 3 (CNS mrswitch) [CNS-UNVERIFIED] */
    double k_bond;    /* pseudo-bond force constant (calibrated) */
    double b0;        /* 3.8 A */
    double r0_rep;    /* repel contact distance R0 (scaled by `repel` s) */
    double k_rep;     /* bead-level repel force-constant multiplier (calibrated) */
    double k_ang;     /* soft (i,i+2) lower-bound force constant (pseudo-angle), 0 = off */
    double a0;        /* (i,i+2) lower bound distance */
    double mass;      /* 100  chromosome3D.pl:1416 */
    double fbeta;     /* 10   chromosome3D.pl:1415 */
    int msoexp;       /* noe_pot 3: exponent of the lower side's soft form, 1 or 2 (CNS msoexponent) [CNS-UNVERIFIED] */
} c3o_model;

typedef struct {
    int kind;         /* 0 MD + Berendsen T-coupling, 1 MD + velocity rescale, 2 FIRE minimise, 5 two-point step-size minimise (c3o_bb_step), FIRE after g_two_point_steps of them */
    int nsteps;
    double dt;        /* ps */
    double w_all;     /* `weights * w`            */
    double w_vdw;     /* absolute vdw weight       */
    double repel_s;   /* nbonds repel=            */
    double t_bath;    /* K */
} c3o_stage;

#define KBOLTZ 0.0019872      /* kcal/mol/K  (X-PLOR/CNS AKMA) */
#define ACCEL 418.4           /* (kcal/mol/A/amu) -> A/ps^2   */

/* derivative helper: returns dE/dd (without S and w) and adds energy (without S,w) */
static inline double softsq(const c3o_model* m, double delta, double* e) {
    const double rs = m->rswitch;
    const double ad = fabs(delta);
    /* continuity constants for the linear tail: e = a + c*|delta| with c = asym*rs ... for
       sqexponent 2, soexponent 1: a + b/D + c D, b chosen 0 when c = 2 rs */
    const double c = m->asym * rs;               /* slope of the tail, = 2 at defaults */
    const double b = (c - 2.0 * rs) * rs * rs;   /* from -b/rs^2 + c = 2 rs  -> b = (c - 2rs) rs^2 */
    const double a = rs * rs - b / rs - c * rs;  /* from a + b/rs + c rs = rs^2 */
    int soft;
    if (m->noe_pot == 0) soft = ad > rs;
    else if (m->noe_pot == 1) soft = delta > rs;
    else if (m->noe_pot == 3) {
        const double mrs = m->mrswitch;
        if (delta < -mrs) {  /* lower side: CNS minus-side soft form a + b/D^p + c D, c = masym, p = msoexponent
                                (1 or 2), C1-continuous at D = mrswitch */
            const double mc = m->masym;
            if (m->msoexp == 2) {
                const double mb = (mc - 2.0 * mrs) * mrs * mrs * mrs / 2.0;      /* from c - 2 b / mrs^3 = 2 mrs */
                const double ma = mrs * mrs - mb / (mrs * mrs) - mc * mrs;
                *e += ma + mb / (ad * ad) + mc * ad;
                return -(-2.0 * mb / (ad * ad * ad) + mc);
            }
            const double mb = (mc - 2.0 * mrs) * mrs * mrs;
            const double ma = mrs * mrs - mb / mrs - mc * mrs;
            *e += ma + mb / ad + mc * ad;
            return -(-mb / (ad * ad) + mc);
        }
        soft = delta > rs;
    }
    else soft = 0;
    if (!soft) { *e += delta * delta; return 2.0 * delta; }
    *e += a + b / ad + c * ad;
    const double g = -b / (ad * ad) + c;
    return delta > 0 ? g : -g;
}

typedef struct { double e_noe, e_bond, e_rep; } c3o_energy;

/* tgt10: n*n tenths of Angstrom (<=0 none).  F (n*3) receives the TOTAL weighted force.
 * Energies are returned unweighted by w_all / w_vdw but include S, k_b and the repel form. */
void c3o_energy_force(const c3o_model* m, const int32_t* tgt10, const double* x, double w_all,
                      double w_vdw, double repel_s, double* F, c3o_energy* en) {
    const int n = m->n;
    double e_noe = 0, e_bond = 0, e_rep = 0;
    if (F) memset(F, 0, sizeof(double) * 3 * (size_t)n);
    const double R2 = (repel_s * m->r0_rep) * (repel_s * m->r0_rep);
    for (int i = 0; i < n; ++i) {
        for (int j = i + 1; j < n; ++j) {
            const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < 1e-12) r2 = 1e-12;
            double coef = 0.0;    /* F_i += coef * dvec ; F_j -= coef * dvec */
            const int sep = j - i;
            const int32_t t10 = tgt10[(size_t)i * n + j];
            if (sep >= m->min_sep && t10 > 0) {
                const double d = sqrt(r2);
                double e = 0;
                const double g = softsq(m, d - 0.1 * t10, &e);
                e_noe += m->s_noe * e;
                coef -= w_all * m->s_noe * g / d;
            }
            if (sep == 1) {
                const double d = sqrt(r2);
                const double dl = d - m->b0;
                e_bond += m->k_bond * dl * dl;
                coef -= w_all * 2.0 * m->k_bond * dl / d;
            }
            if (sep >= m->rep_sep && r2 < R2) {
                const double q = R2 - r2;
                e_rep += m->k_rep * q * q;
                coef += w_vdw * m->k_rep * 4.0 * q;
            }
            if (sep == 2 && m->k_ang > 0 && (m->ang_mode == 1 || r2 < m->a0 * m->a0)) {
                const double d = sqrt(r2);
                const double dl = d - m->a0;
                e_bond += m->k_ang * dl * dl;
                coef -= w_all * 2.0 * m->k_ang * dl / d;
            }
            if (F && coef != 0.0) {
                F[3 * i] += coef * dx; F[3 * i + 1] += coef * dy; F[3 * i + 2] += coef * dz;
                F[3 * j] -= coef * dx; F[3 * j + 1] -= coef * dy; F[3 * j + 2] -= coef * dz;
            }
        }
    }
    if (en) { en->e_noe = e_noe; en->e_bond = e_bond; en->e_rep = e_rep; }
}

/* ------------------------------------------------------------------------------------ */
/* Initial state: random coil (step b0) + Maxwell velocities, Philox keyed (seed, replica) */
/* ------------------------------------------------------------------------------------ */
void c3o_init_coords(const c3o_model* m, uint64_t seed, uint32_t replica, double* x) {
    const int n = m->n;
    double px = 0, py = 0, pz = 0;
    for (int i = 0; i < n; ++i) {
        if (i > 0) {
            double g[4];
            normals4(seed, replica, (uint32_t)i, 0u, g);
            double nrm = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            if (nrm < 1e-12) { g[0] = 1; g[1] = g[2] = 0; nrm = 1; }
            px += m->b0 * g[0] / nrm; py += m->b0 * g[1] / nrm; pz += m->b0 * g[2] / nrm;
        }
        x[3 * i] = px; x[3 * i + 1] = py; x[3 * i + 2] = pz;
    }
    double cx = 0, cy = 0, cz = 0;
    for (int i = 0; i < n; ++i) { cx += x[3 * i]; cy += x[3 * i + 1]; cz += x[3 * i + 2]; }
    cx /= n; cy /= n; cz /= n;
    for (int i = 0; i < n; ++i) { x[3 * i] -= cx; x[3 * i + 1] -= cy; x[3 * i + 2] -= cz; }
}
/* A5, the reference's start structure restated for beads: extn.inp lays the chain out along x (`ident (x) (all)`,
 * `do (x=x/5.)`, chromosome3D.pl:2413-2414), gives every atom a random y and z in [0, 0.5) (`do (y=random(0.5))`,
 * `do (z=random(0.5))`, :2415-2416) and regularises the result to covalent geometry (:2424-2517).  Bead i sits b0 * i along x
 * (the regularised Calpha spacing), y and z uniform in [0, 0.5) from Philox words 0 and 1 of counter (i, purpose 2) under the
 * replica's key; centred like the coil start. */
void c3o_init_coords_extended(const c3o_model* m, uint64_t seed, uint32_t replica, double* x) {
    const int n = m->n;
    const uint32_t key[2] = {(uint32_t)(seed & 0xFFFFFFFFu) ^ (replica * 0x9E3779B9u), (uint32_t)(seed >> 32) + replica};
    for (int i = 0; i < n; ++i) {
        const uint32_t ctr[4] = {(uint32_t)i, 2u, 0u, 0u};
        uint32_t u[4];
        c3o_philox4x32(ctr, key, u);
        x[3 * i] = m->b0 * i; x[3 * i + 1] = 0.5 * u01(u[0]); x[3 * i + 2] = 0.5 * u01(u[1]);
    }
    double cx = 0, cy = 0, cz = 0;
    for (int i = 0; i < n; ++i) { cx += x[3 * i]; cy += x[3 * i + 1]; cz += x[3 * i + 2]; }
    cx /= n; cy /= n; cz /= n;
    for (int i = 0; i < n; ++i) { x[3 * i] -= cx; x[3 * i + 1] -= cy; x[3 * i + 2] -= cz; }
}
/* v ~ Maxwell(T): sigma = sqrt(kB T * ACCEL / m)  [A/ps]   (deck :1646-1648, T = 0.5 K) */
void c3o_init_velocities(const c3o_model* m, uint64_t seed, uint32_t replica, double temp, double* v) {
    const double sigma = sqrt(KBOLTZ * temp * ACCEL / m->mass);
    for (int i = 0; i < m->n; ++i) {
        double g[4];
        normals4(seed, replica, (uint32_t)i, 1u, g);
        v[3 * i] = sigma * g[0]; v[3 * i + 1] = sigma * g[1]; v[3 * i + 2] = sigma * g[2];
    }
}

/* ------------------------------------------------------------------------------------ */
/* Integrators                                                                           */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    double dt, alpha;
    int npos;
    int started;
} c3o_fire_state;

typedef struct {
    double dt_start, dt_max, f_inc, f_dec, alpha_start, f_alpha, max_step;
    int n_min;
} c3o_fire_params;

static double temperature_of(const c3o_model* m, const double* v) {
    double ke2 = 0;
    for (int k = 0; k < 3 * m->n; ++k) ke2 += v[k] * v[k];
    const int ndf = 3 * m->n - 3;
    return m->mass * ke2 / ACCEL / ((ndf > 0 ? ndf : 1) * KBOLTZ);
}

/* one leap-frog MD step (see header).  F is scratch (n*3). */
void c3o_md_step(const c3o_model* m, const int32_t* tgt10, const c3o_stage* st, double* x, double* v, double* F) {
    const int n = m->n;
    c3o_energy_force(m, tgt10, x, st->w_all, st->w_vdw, st->repel_s, F, NULL);
    double tprev = temperature_of(m, v);
    if (tprev < 1e-2) tprev = 1e-2;
    double lam;
    if (st->kind == 0) {
        double l2 = 1.0 + st->dt * m->fbeta * (st->t_bath / tprev - 1.0);
        if (l2 < 0) l2 = 0;
        lam = sqrt(l2);
    } else {
        lam = sqrt(st->t_bath / tprev);
    }
    double cm[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) { cm[0] += v[3 * i]; cm[1] += v[3 * i + 1]; cm[2] += v[3 * i + 2]; }
    cm[0] /= n; cm[1] /= n; cm[2] /= n;
    const double acc = st->dt * ACCEL / m->mass;
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            const int k = 3 * i + c;
            v[k] = lam * (v[k] - cm[c]) + acc * F[k];
            x[k] += st->dt * v[k];
        }
}

/* one FIRE step in the form the device runs it (one force evaluation, one launch per step):
 *   F = F(x);  the power / norm test of Bitzek et al. 2006 uses the sums L = (v.F, F.F, v.v) of
 *   the PREVIOUS step (a one-step lag: a global sum of this step's F is only available to the
 *   next launch), then the usual mixing, adaptive dt / alpha, semi-implicit Euler move with a
 *   per-bead displacement clamp.  L is replaced by this step's sums.  A stage starts with
 *   v = 0, L = 0, state (dt_start, alpha_start, 0). */
void c3o_fire_step(const c3o_model* m, const int32_t* tgt10, const c3o_stage* st, const c3o_fire_params* fp,
                   c3o_fire_state* fs, double* L, double* x, double* v, double* F) {
    const int n = m->n;
    c3o_energy_force(m, tgt10, x, st->w_all, st->w_vdw, st->repel_s, F, NULL);
    double vf = 0, ff = 0, vv = 0;
    for (int k = 0; k < 3 * n; ++k) { vf += v[k] * F[k]; ff += F[k] * F[k]; vv += v[k] * v[k]; }
    if (L[0] > 0) {
        const double mix = fs->alpha * sqrt(L[2] / (L[1] > 1e-30 ? L[1] : 1e-30));
        for (int k = 0; k < 3 * n; ++k) v[k] = (1.0 - fs->alpha) * v[k] + mix * F[k];
        if (fs->npos > fp->n_min) {
            fs->dt = fs->dt * fp->f_inc < fp->dt_max ? fs->dt * fp->f_inc : fp->dt_max;
            fs->alpha *= fp->f_alpha;
        }
        fs->npos += 1;
    } else {
        for (int k = 0; k < 3 * n; ++k) v[k] = 0;
        fs->alpha = fp->alpha_start;
        fs->dt *= fp->f_dec;
        fs->npos = 0;
    }
    const double acc = fs->dt * ACCEL / m->mass;
    for (int i = 0; i < n; ++i) {
        double dr[3];
        double d2 = 0;
        for (int c = 0; c < 3; ++c) {
            const int k = 3 * i + c;
            v[k] += acc * F[k];
            dr[c] = fs->dt * v[k];
            d2 += dr[c] * dr[c];
        }
        const double sc = d2 > fp->max_step * fp->max_step ? fp->max_step / sqrt(d2) : 1.0;
        for (int c = 0; c < 3; ++c) x[3 * i + c] += sc * dr[c];
    }
    L[0] = vf; L[1] = ff; L[2] = vv;
}

/* One step of the two-point step-size (Barzilai-Borwein 1988) minimiser in the form the device runs it — stage kind 5, round 5's
 * opt-in alternative to FIRE for the final minimisation (deck chromosome3D.pl:1790-1803 runs L-BFGS there; this is the scalar,
 * memoryless member of the same secant family: the step length is the inverse of a one-number curvature estimate from the last
 * move s and the change of the gradient y over it).  One force evaluation a step, no energy, no line search:
 *   F = F(x_k);   s = the previous move (a_prev F_old, per-bead clamp max_step),  y = F_old - F (the gradient is -F);
 *   this step's sums  L' = (s.y, F.F, y.y, s.s);
 *   the length of THIS move comes from the sums L of the PREVIOUS evaluation (the device's one-step lag, as in c3o_fire_step:
 *   a "gradient method with retards", Friedlander et al. 1999): s.y > 0: a = s.s / s.y on even evaluations, s.y / y.y on odd ones
 *   (the two Barzilai-Borwein lengths, alternated); s.y <= 0 (negative curvature along the move): twice the previous length;
 *   kept inside [1e-7, 1e2];  evaluations 0 and 1 of a stage, which have no completed pair yet, move with
 *   a0 = dt_start^2 ACCEL / mass (FIRE's first displacement per unit force);
 *   x += clamp(a F) per bead, F_old = F (kept in the velocity array), L = L'.
 * fs->dt holds the length of the last move, fs->npos the evaluation count of the stage. */
void c3o_bb_step(const c3o_model* m, const int32_t* tgt10, const c3o_stage* st, const c3o_fire_params* fp,
                 c3o_fire_state* fs, double* L, double* x, double* v, double* F) {
    const int n = m->n;
    c3o_energy_force(m, tgt10, x, st->w_all, st->w_vdw, st->repel_s, F, NULL);
    const int k = fs->npos;
    const double a_prev = fs->dt;
    double a = a_prev;
    if (k == 0) a = fp->dt_start * fp->dt_start * ACCEL / m->mass;
    else if (k >= 2) {
        const double sy = L[0];
        if (sy > 0) a = (k % 2 == 0) ? L[3] / sy : sy / L[2];
        else a = 2.0 * a_prev;
        if (!(a >= 1e-7)) a = 1e-7;
        if (a > 1e2) a = 1e2;
    }
    double q[4] = {0, 0, 0, 0};
    const double ms2 = fp->max_step * fp->max_step;
    for (int i = 0; i < n; ++i) {
        double s[3] = {0, 0, 0}, d2 = 0;
        if (k > 0) {
            for (int c = 0; c < 3; ++c) { s[c] = a_prev * v[3 * i + c]; d2 += s[c] * s[c]; }
            const double sc = d2 > ms2 ? fp->max_step / sqrt(d2) : 1.0;
            for (int c = 0; c < 3; ++c) {
                s[c] *= sc;
                const double y = v[3 * i + c] - F[3 * i + c];
                q[0] += s[c] * y; q[2] += y * y; q[3] += s[c] * s[c];
            }
        }
        double dr[3];
        d2 = 0;
        for (int c = 0; c < 3; ++c) { q[1] += F[3 * i + c] * F[3 * i + c]; dr[c] = a * F[3 * i + c]; d2 += dr[c] * dr[c]; }
        const double sc = d2 > ms2 ? fp->max_step / sqrt(d2) : 1.0;
        for (int c = 0; c < 3; ++c) { x[3 * i + c] += sc * dr[c]; v[3 * i + c] = F[3 * i + c]; }
    }
    fs->dt = a; fs->npos = k + 1;
    for (int c = 0; c < 4; ++c) L[c] = q[c];
}

static int g_two_point_steps = 1000;     /* device option final_minimiser_steps */
void c3o_set_two_point_steps(int k) { g_two_point_steps = k < 2 ? 2 : k; }

/* Run a schedule of stages on one replica.  x (n*3) in/out, v scratch (n*3).
 * Returns number of force evaluations.  If gtol > 0, a FIRE stage exits when the RMS force
 * drops below gtol (checked every `check_every` steps, as the device path does). */
long c3o_run_schedule(const c3o_model* m, const int32_t* tgt10, const c3o_stage* stages, int n_stages,
                      const c3o_fire_params* fp, double gtol, int check_every, uint64_t seed,
                      uint32_t replica, double* x, double* v_out) {
    const int n = m->n;
    double* v = (double*)calloc((size_t)3 * n, sizeof(double));
    double* F = (double*)calloc((size_t)3 * n, sizeof(double));
    long evals = 0;
    int prev_kind = -1;
    for (int s = 0; s < n_stages; ++s) {
        const c3o_stage* st = &stages[s];
        if (st->kind == 2 || st->kind == 5) {
            /* kind 5: two-point steps for the first g_two_point_steps evaluations of the stage, then FIRE from a fresh state (v = 0) for
             * the rest — the two-point method has no descent guarantee, FIRE takes over what it has not finished (device: build_program) */
            c3o_fire_state fs = {fp->dt_start, fp->alpha_start, 0, 1};
            double L[4] = {0, 0, 0, 0};
            const int nbb = st->kind == 5 ? (st->nsteps < g_two_point_steps ? st->nsteps : g_two_point_steps) : 0;
            for (int k = 0; k < 3 * n; ++k) v[k] = 0;
            for (int it = 0; it < st->nsteps; ++it) {
                if (it == nbb && nbb > 0) {
                    const c3o_fire_state fresh = {fp->dt_start, fp->alpha_start, 0, 1};
                    fs = fresh;
                    L[0] = L[1] = L[2] = L[3] = 0;
                    for (int k = 0; k < 3 * n; ++k) v[k] = 0;
                }
                if (it >= nbb) c3o_fire_step(m, tgt10, st, fp, &fs, L, x, v, F);
                else c3o_bb_step(m, tgt10, st, fp, &fs, L, x, v, F);
                ++evals;
                if (gtol > 0 && check_every > 0 && (it + 1) % check_every == 0 && sqrt(L[1] / (3.0 * n)) < gtol) break;
            }
        } else {
            if (prev_kind == 2 || prev_kind == 5 || prev_kind == -1) c3o_init_velocities(m, seed, replica, 0.5, v);
            for (int it = 0; it < st->nsteps; ++it) { c3o_md_step(m, tgt10, st, x, v, F); ++evals; }
        }
        prev_kind = st->kind;
    }
    /* centre (deck :1806-1816) */
    double c[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) c[k] += x[3 * i + k];
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) x[3 * i + k] -= c[k] / n;
    if (v_out) memcpy(v_out, v, sizeof(double) * 3 * (size_t)n);
    free(v); free(F);
    return evals;
}

/* ------------------------------------------------------------------------------------ */
/* Assessment (chromosome3D.pl:447-485, 581-600, 716-729) and Spearman metric             */
/* ------------------------------------------------------------------------------------ */
static double round3(double d) {  /* sprintf "%.3f" then numeric use (calc_dist :727) */
    char b[64];
    snprintf(b, sizeof b, "%.3f", d);
    return strtod(b, NULL);
}
/* xyz here are the PDB-rounded coordinates (the reference reads them back from %8.3f text) */
void c3o_assess(const double* x, int R, const int* ri, const int* rj, const int32_t* rt10, double relax,
                int* satisfied, double* sum_dev) {
    int count = 0;
    double sdev = 0;
    for (int k = 0; k < R; ++k) {
        const int i = ri[k] - 1, j = rj[k] - 1;
        const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
        const double d = round3(sqrt(dx * dx + dy * dy + dz * dz));
        /* the tbl text "%d.%d0" read back by Perl = nearest double to t10/10 = this quotient */
        const double t = rt10[k] / 10.0;
        const double dminus = 0.0, dplus = 0.0;
        if (d < t + dplus + relax) ++count;
        if (d < t - dminus - relax) --count;
        if (d > t + dplus + 0.2) sdev += d - (t + dplus);
        if (d < t - dminus - 0.2) sdev += (t - dminus) - d;
    }
    *satisfied = count;
    *sum_dev = sdev;
}

typedef struct { double v; int idx; } rk;
static int rk_cmp(const void* a, const void* b) {
    const double x = ((const rk*)a)->v, y = ((const rk*)b)->v;
    return (x > y) - (x < y);
}
static void avg_ranks(const double* v, size_t m, double* r) {
    rk* a = (rk*)malloc(sizeof(rk) * m);
    for (size_t k = 0; k < m; ++k) { a[k].v = v[k]; a[k].idx = (int)k; }
    qsort(a, m, sizeof(rk), rk_cmp);
    size_t k = 0;
    while (k < m) {
        size_t e = k;
        while (e + 1 < m && a[e + 1].v == a[k].v) ++e;
        const double rank = 0.5 * ((double)k + (double)e) + 1.0;
        for (size_t q = k; q <= e; ++q) r[a[q].idx] = rank;
        k = e + 1;
    }
    free(a);
}
/* spearman_IF_pdb.pl:42-70: all ORDERED pairs |r1-r2| >= range, d rounded "%.3f";
 * Spearman = Pearson of average ranks. */
double c3o_spearman_if_dist(const double* IF, const double* x, int n, int range) {
    size_t m = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (abs(i - j) >= range) ++m;
    if (m < 2) return 0.0;
    double* a = (double*)malloc(sizeof(double) * m);
    double* b = (double*)malloc(sizeof(double) * m);
    double* ra = (double*)malloc(sizeof(double) * m);
    double* rb = (double*)malloc(sizeof(double) * m);
    size_t k = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (abs(i - j) < range) continue;
            const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
            a[k] = IF[(size_t)i * n + j];
            b[k] = round3(sqrt(dx * dx + dy * dy + dz * dz));
            ++k;
        }
    avg_ranks(a, m, ra);
    avg_ranks(b, m, rb);
    double ma = 0, mb = 0;
    for (k = 0; k < m; ++k) { ma += ra[k]; mb += rb[k]; }
    ma /= m; mb /= m;
    double sab = 0, saa = 0, sbb = 0;
    for (k = 0; k < m; ++k) { sab += (ra[k] - ma) * (rb[k] - mb); saa += (ra[k] - ma) * (ra[k] - ma); sbb += (rb[k] - mb) * (rb[k] - mb); }
    free(a); free(b); free(ra); free(rb);
    return sab / sqrt(saa * sbb);
}

/* ------------------------------------------------------------------------------------ */
/* A7: metric-matrix distance geometry at bead level (deck :1471-1525, knobs :1008-1090)   */
/*   bounds  -> triangle smoothing (shortest paths) -> random trial distances ->           */
/*   metric (Gram) matrix -> top-3 eigenvectors (orthogonal iteration) -> coordinates.     */
/* The reference embeds a substructure of the pseudo-protein with CNS `mmdg`; here every   */
/* bead is embedded: bond pairs are fixed at b0, restrained pairs at their target          */
/* (dminus = dplus = 0, chromosome3D.pl:352-354), the rest gets [lower_default, +inf).     */
/* ------------------------------------------------------------------------------------ */
#define C3O_DG_INF 1.0e30
void c3o_dg_bounds(const c3o_model* m, const int32_t* tgt10, double lower_default, double* U, double* L) {
    const int n = m->n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int sep = abs(i - j);
            double u = C3O_DG_INF, l = lower_default;
            const int32_t t10 = tgt10[(size_t)i * n + j];
            if (sep == 0) { u = 0; l = 0; }
            else if (sep == 1) { u = m->b0; l = m->b0; }
            else if (sep >= m->min_sep && t10 > 0) { u = 0.1 * t10; l = 0.1 * t10; }
            U[(size_t)i * n + j] = u;
            L[(size_t)i * n + j] = l;
        }
}
/* upper bounds: all-pairs shortest paths (Floyd-Warshall); lower bounds: inverse triangle
 * inequality with the smoothed upper bounds; finally L <= U is enforced */
void c3o_dg_smooth(int n, double* U, double* L) {
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i) {
            const double uik = U[(size_t)i * n + k];
            for (int j = 0; j < n; ++j) {
                const double v = uik + U[(size_t)k * n + j];
                if (v < U[(size_t)i * n + j]) U[(size_t)i * n + j] = v;
            }
        }
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                const double a = L[(size_t)i * n + k] - U[(size_t)k * n + j];
                const double b = L[(size_t)k * n + j] - U[(size_t)i * n + k];
                double v = L[(size_t)i * n + j];
                if (a > v) v = a;
                if (b > v) v = b;
                L[(size_t)i * n + j] = v;
            }
    for (size_t q = 0; q < (size_t)n * n; ++q) if (L[q] > U[q]) L[q] = U[q];
}
/* trial distance of pair (i<j): uniform in [L, U], Philox counter (i, j, 2) */
static double dg_uniform(uint64_t seed, uint32_t replica, uint32_t i, uint32_t j) {
    uint32_t ctr[4] = {i, j, 2u, 0u};
    uint32_t key[2] = {(uint32_t)(seed & 0xFFFFFFFFu) ^ (replica * 0x9E3779B9u), (uint32_t)(seed >> 32) + replica};
    uint32_t r[4];
    c3o_philox4x32(ctr, key, r);
    return u01(r[0]);
}
void c3o_dg_trial_d2(int n, const double* U, const double* L, uint64_t seed, uint32_t replica, double* D2) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) { D2[(size_t)i * n + j] = 0; continue; }
            const int a = i < j ? i : j, b = i < j ? j : i;
            const double lo = L[(size_t)a * n + b], hi = U[(size_t)a * n + b];
            const double d = lo + dg_uniform(seed, replica, (uint32_t)a, (uint32_t)b) * (hi - lo);
            D2[(size_t)i * n + j] = d * d;
        }
}
/* y = B v with B = -1/2 J D2 J (J = centring projector) */
static void dg_matvec(int n, const double* D2, const double* v, double* y) {
    double mv = 0;
    for (int i = 0; i < n; ++i) mv += v[i];
    mv /= n;
    double mz = 0;
    for (int i = 0; i < n; ++i) {
        double z = 0;
        for (int j = 0; j < n; ++j) z += D2[(size_t)i * n + j] * (v[j] - mv);
        y[i] = z;
        mz += z;
    }
    mz /= n;
    for (int i = 0; i < n; ++i) y[i] = -0.5 * (y[i] - mz);
}
/* orthogonal (subspace) iteration for the 3 leading eigenpairs; x = sqrt(lambda_k) v_k, centred */
void c3o_dg_embed(int n, const double* D2, uint64_t seed, uint32_t replica, int iters, double* x) {
    double* V = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double* W = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double g[4];
        normals4(seed, replica, (uint32_t)i, 3u, g);
        for (int k = 0; k < 3; ++k) V[(size_t)k * n + i] = g[k];
    }
    double lam[3] = {0, 0, 0};
    for (int it = 0; it <= iters; ++it) {
        for (int k = 0; k < 3; ++k) dg_matvec(n, D2, V + (size_t)k * n, W + (size_t)k * n);
        if (it == iters) {   /* Rayleigh quotients with the current orthonormal V */
            for (int k = 0; k < 3; ++k) {
                double s = 0;
                for (int i = 0; i < n; ++i) s += V[(size_t)k * n + i] * W[(size_t)k * n + i];
                lam[k] = s;
            }
            break;
        }
        for (int k = 0; k < 3; ++k) {   /* modified Gram-Schmidt */
            for (int q = 0; q < k; ++q) {
                double s = 0;
                for (int i = 0; i < n; ++i) s += W[(size_t)k * n + i] * V[(size_t)q * n + i];
                for (int i = 0; i < n; ++i) W[(size_t)k * n + i] -= s * V[(size_t)q * n + i];
            }
            double nn = 0;
            for (int i = 0; i < n; ++i) nn += W[(size_t)k * n + i] * W[(size_t)k * n + i];
            nn = sqrt(nn > 1e-300 ? nn : 1e-300);
            for (int i = 0; i < n; ++i) V[(size_t)k * n + i] = W[(size_t)k * n + i] / nn;
        }
    }
    double c[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        const double s = sqrt(lam[k] > 0 ? lam[k] : 0);
        for (int i = 0; i < n; ++i) { x[3 * i + k] = s * V[(size_t)k * n + i]; c[k] += x[3 * i + k]; }
    }
    for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) x[3 * i + k] -= c[k] / n;
    free(V); free(W);
}
