/*
 * wfsim_oracle.c -- CPU restatement (plain C, scalar, one thread) of WFSim's photon -> raw_records hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (wfsim_amd/csrc, HIP) never links or calls it.
 *
 * Parity status: PINNED.  Every deterministic stage below is checked bit-for-bit against vectors produced by
 * running the reference itself (tests/golden/make_golden.py, WFSim v1.2.2 imported from /root/reference in the
 * build container): add_current currents, pulse bounds, digitised rows, channel masks, ZLE tuples, add_noise, the
 * Pulse-call structure of run sets and of the electron-afterpulse feedback loop (golden chains A-H).  The random
 * stages cannot be stream-compatible with numpy's legacy generator; they follow the reference's arithmetic
 * (same truncations, term by term) on a counter-based Philox4x32-10 stream (layout in DESIGN.md "RNG streams")
 * and are pinned statistically against histograms of the reference's own draws (tests/golden/dists.npz, and
 * dists_models.npz for the timing model variants).  Unpinned: the 244-byte raw_record layout of orc_pack_records (strax is
 * not installed: restated from its published dtype).
 *
 * Each function names the reference lines it restates (paths relative to /root/reference/wfsim).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;
typedef int32_t i32;
typedef uint32_t u32;
typedef uint64_t u64;

/* ---------------------------------------------------------------- configuration ------------------ */
typedef struct {
    /* digitiser / pulse geometry */
    i32 dt, samples_before, samples_after, store_before, store_after, tlen;
    i32 trigger_window, baseline, n_rows;           /* n_rows = 801 */
    i32 n_tpc, n_top, he_first, he_factor, sum_channel, last_bottom, detector_nt;
    i32 n_spe_channels, noise_len, noise_channels, enable_noise;
    /* models */
    i32 s1_simple, s2_time_model /*0 zero_delay, 1 spread around zero*/, n_lum, enable_pmt_ap, n_ap_elements;
    i32 tile_gen;                                   /* tile-local generation of eligible S2s (RNG spec v9): see gen_s2 */
    i32 tile_gen_min;                               /* ... from this many photons per tile (amp * gain * max p_ch) on */
    i32 fma;                                        /* 1: add_current accumulates with fma(template, gain, current) -- one rounding per term instead
                                                       of two; the switch of wfs_config.fma (north_star tolerance: 1 ADC count).  0: the reference's
                                                       separately rounded product and sum (bit-exact currents) */
    double c2a;                                     /* current_2_adc, pulse.py:33-35 */
    double tts_mean, tts_sigma, p_dpe;
    double s1_decay_time, s1_decay_spread;
    double sf_gas, t1_gas, t3_gas, s2_time_spread;
    double trap_time, gain_spread;
    double pmt_ap_modifier, pmt_ap_t_modifier;
    double rext, drift_velocity;
    u64 seed;
} orc_config;

typedef struct {
    i32 n_bins_delay, n_bins_amp, amp_2d, is_uniform;
    double delay_bin, amp_bin;
    const double *delay_cdf;   /* [n_ch][n_bins_delay] */
    const double *amp_cdf;     /* [n_ch][n_bins_amp] or [n_bins_amp] */
} orc_ap_element;

typedef struct { i64 *p; i64 n, cap; } vec_i64;
typedef struct { double *p; i64 n, cap; } vec_f64;
typedef struct { i32 *p; i64 n, cap; } vec_i32;
typedef struct { int16_t *p; i64 n, cap; } vec_i16;
typedef struct { uint8_t *p; i64 n, cap; } vec_u8;

#define VEC_PUSH(v, T, x) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? (v).cap * 2 : 1024; \
    (v).p = (T *)realloc((v).p, (size_t)(v).cap * sizeof(T)); } (v).p[(v).n++] = (x); } while (0)
#define VEC_RESERVE(v, T, m) do { if ((v).n + (m) > (v).cap) { while ((v).n + (m) > (v).cap) (v).cap = (v).cap ? (v).cap * 2 : 1024; \
    (v).p = (T *)realloc((v).p, (size_t)(v).cap * sizeof(T)); } } while (0)

/* a discrete delay distribution: cumulative probabilities (cum[i] = P(X <= vmin + i)) and -- for the tables photons are
 * drawn from -- its Walker alias table (K = 2^lg cells of {threshold, alias}; see alias_build / alias_sample) */
typedef struct { double *cum; i64 n; i64 vmin; u32 *thr, *alias; int lg; } orc_tab;

typedef struct orc_session_s {
    orc_config c;
    const double *templates;     /* [10][tlen]                       pulse.py:146-187 */
    const double *spe;           /* [n_spe_channels][2001]           pulse.py:189-223 */
    const double *gains;         /* [n_tpc]                                          */
    const double *thr_truth;     /* [n_rows] zle/special threshold - 0.5, pulse.py:240-243 */
    const i64 *thr_zle;          /* [n_rows] baseline - thr - 1, rawdata.py:290-294  */
    const double *lum_x, *lum_t; /* [n_lum] normalised cdf, emission time, s2.py:333-338 */
    const int16_t *noise;        /* [noise_len][noise_channels] */
    const double *noise_f;       /* float noise (orc_set_noise_float): the truncated sum is stored, rawdata.py:436 */
    orc_ap_element ap[8];
    double current_max[16];      /* per ns remainder: sample_duration <= 16 ns */

    /* pulse cache (rawdata.py:180-190) */
    vec_i32 pl_ch, pl_runset; vec_i64 pl_left, pl_right, pl_cur_off, pl_nph; vec_f64 cur;
    i64 first_uncommitted_pulse;
    i64 last_end; int has_pulse;              /* RawData.last_pulse_end_time (exact integer once a pulse exists) */
    /* all generated photons (channel sorted per pulse call) */
    vec_i64 ph_t; vec_i16 ph_ch; vec_u8 ph_dpe; vec_f64 ph_gain; vec_i64 call_ph_off; vec_i32 call_kind, call_runset;
    vec_i64 e_t; vec_i64 call_e_off;
    /* digitise groups */
    vec_i64 dg_left, dg_right, dg_first_pulse, dg_n_pulses, dg_ix_rand, dg_row_off;
    vec_i32 row_ch; vec_i64 row_left, row_right, row_data_off; vec_i32 row_data;
    /* ZLE intervals */
    vec_i64 zl_digit, zl_left, zl_right, zl_data_off; vec_i32 zl_ch; vec_i32 zl_data;
    /* truth accumulators per pulse call (pulse.py:229-271): 12 doubles each */
    vec_f64 truth;
    i64 n_pe_total;
    const i64 *noise_override; i64 n_noise_override;
    u32 win_gid; int win_gid_set; /* noise stream of the open window: gid of the first instruction of its first cluster that made a pulse */
    int save_full_truth;         /* rawdata.py:42: 1 (default) = every instruction is its own Pulse call */
    orc_tab tab[10];
    /* model variants of the photon delays (orc_set_delay_models ...): extra tables, per-instruction choice, S1 propagation */
    orc_tab *xtab; i32 n_xtab;
    const i32 *ins_tab, *ins_tabb, *ins_pzi; const double *ins_pzf; i64 n_ins_models;
    double *prop_top, *prop_bot; i32 prop_nz, prop_nu; double prop_u0, prop_du;
    /* s2_luminescence_model 'garfield_gas_gap' (s2.py:413-483): inverse CDFs of the excitation time per tabulated gas gap,
     * per instruction the lower table and the interpolation weight towards the next one */
    double *gg_inv; i32 gg_n, gg_L; const i32 *ins_gg; const double *ins_ggw; i64 n_ins_gg; i32 cur_gg; double cur_ggw;
    i32 cur_tab, cur_tabb, cur_pzi; double cur_pzf;   /* trunc()-ed delay variates: the individual terms and their sums, see TAB_* */    /* tests: ix_rand per digitise call instead of the Philox draw */
} orc_session;

/* ---------------------------------------------------------------- Philox4x32-10 ------------------ */
/* Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3" (SC'11); constants of Random123. */
static inline void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 out[4])
{
    for (int r = 0; r < 10; r++) {
        u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
        u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n1 = (u32)p1, n2 = (u32)(p0 >> 32) ^ c3 ^ k1, n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(const u32 *ctr, const u32 *key, u32 *out) { philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out); }

/* draw sites (counter word 3); counter = (emitter, instruction gid, item, site) -- DESIGN.md "RNG streams" */
/* RNG spec v6, photon streams: counter (em_base, gid, P >> 2, site) with P the index of the photon among its instruction's
 * photons; photon P owns word P & 3 of SITE_DELAY -> summed delay (alias table), SITE_CH -> channel, SITE_GAIN -> first SPE
 * index, double-PE flag, second SPE index.  SITE_PH, counter (0, gid, item): photons that arrive with time and channel
 * (optical input): x -> transit time, y -> gains.  SITE_PH_X, counter (emitter, gid, item): S1 optical propagation. */
enum { SITE_S1_HIT = 1, SITE_S2_SURVIVE = 2, SITE_EL_A = 3, SITE_EL_B = 4, SITE_EL_POIS = 5,
       SITE_EL_DIFF = 6 /* per electron: Box-Muller pair -> radial / azimuthal transverse diffusion, s2.py:588-589 */,
       SITE_DELAY = 16, SITE_CH = 17, SITE_GAIN = 18, SITE_PH = 19, SITE_PH_X = 20, SITE_LUM = 21 /* P-indexed like SITE_DELAY: garfield gas gap excitation time */,
       /* tile-local generation (spec v9): counter (em_base + channel, gid, item, site) */
       SITE_TILE_N = 24 /* photons of the tile: Poisson, item = iteration */, SITE_TILE_E = 25 /* item = P >> 2, word P & 3 -> surviving electron */,
       SITE_TILE_DELAY = 26, SITE_TILE_GAIN = 27,
       SITE_AP = 32, SITE_AP_SCREEN = 40, SITE_AP_X = 48, SITE_NOISE = 64 };

static inline void draw(const orc_session *s, u32 emitter, u32 gid, u32 item, u32 site, u32 w[4])
{
    philox4x32_10(emitter, gid, item, site, (u32)s->c.seed, (u32)(s->c.seed >> 32), w);
}
static inline double u53(u32 a, u32 b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0; }
static inline u64 bern_threshold(double p)
{
    if (!(p > 0)) return 0;
    if (p >= 1) return 4294967296ull;
    return (u64)(p * 4294967296.0);
}
static inline void box_muller(const u32 w[4], double *z0, double *z1)
{
    double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
    double r = sqrt(-2.0 * log(1.0 - u1));
    double a = 6.283185307179586476925286766559 * u2;
    *z0 = r * cos(a); *z1 = r * sin(a);
}

/* Poisson: Hoermann PTRS for lam >= 10, Knuth multiplication below (the two algorithms numpy's legacy
 * generator uses, s2.py:308 draws np.random.poisson).  Uniform pairs come from SITE_EL_POIS items. */
static i64 poisson_site(const orc_session *s, u32 emitter, u32 gid, u32 site, double lam);
static i64 poisson_draw(const orc_session *s, u32 emitter, u32 gid, double lam) { return poisson_site(s, emitter, gid, SITE_EL_POIS, lam); }
static i64 poisson_site(const orc_session *s, u32 emitter, u32 gid, u32 site, double lam)
{
    u32 w[4]; u32 it = 0;
    if (!(lam > 0)) return 0;
    if (lam < 10) {
        double enlam = exp(-lam), prod = 1.0; i64 x = 0;
        for (;;) {
            draw(s, emitter, gid, it++, site, w);
            prod *= u53(w[0], w[1]);
            if (prod > enlam) x++; else return x;
            prod *= u53(w[2], w[3]);
            if (prod > enlam) x++; else return x;
        }
    }
    double slam = sqrt(lam), loglam = log(lam);
    double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2);
    for (;;) {
        draw(s, emitter, gid, it++, site, w);
        double U = u53(w[0], w[1]) - 0.5, V = u53(w[2], w[3]);
        double us = 0.5 - fabs(U);
        i64 k = (i64)floor((2 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + k * loglam - lgamma((double)k + 1)) return k;
    }
}

/* Photons per electron (s2.py:308): the Poisson distribution of the instruction's secondary gain is tabulated once per gain --
 * pmf(k) = exp(k ln(lam) - lam - lgamma(k + 1)) on the 256 values from floor(lam - 8 sqrt(lam)) - 4 on, summed in order, normalised --
 * and inverted with ONE uniform by bisection (the device: k_poisson_tables / poisson_table_draw); gains above POIS_LAM_MAX keep PTRS. */
#define POIS_W 256
#define POIS_LAM_MAX 217.0
static double g_pois_lam = -1.0, g_pois_cdf[POIS_W]; static i64 g_pois_kmin = 0;
static i64 poisson_any(const orc_session *s, u32 emitter, u32 gid, double lam)
{
    if (!(lam > 0)) return 0;
    if (lam > POIS_LAM_MAX) return poisson_draw(s, emitter, gid, lam);
    if (lam != g_pois_lam) {
        i64 k0 = (i64)floor(lam - 8.0 * sqrt(lam)) - 4; if (k0 < 0) k0 = 0;
        double run = 0;
        for (int j = 0; j < POIS_W; j++) { const double k = (double)(k0 + j); run += exp(k * log(lam) - lam - lgamma(k + 1.0)); g_pois_cdf[j] = run; }
        const double tot = g_pois_cdf[POIS_W - 1];
        for (int j = 0; j < POIS_W; j++) g_pois_cdf[j] = g_pois_cdf[j] / tot;
        g_pois_lam = lam; g_pois_kmin = k0;
    }
    u32 w[4]; draw(s, emitter, gid, 0, SITE_EL_POIS, w);
    const double u = u53(w[0], w[1]);
    int lo = 0, hi = POIS_W - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (u < g_pois_cdf[mid]) hi = mid; else lo = mid + 1; }
    return g_pois_kmin + lo;
}

enum { TAB_TTS = 0, TAB_S1_EXP, TAB_S1_SPREAD, TAB_T1, TAB_T3, TAB_S2_SPREAD, TAB_LUM, TAB_S1_TOTAL, TAB_S2_TOTAL, TAB_S2_NOLUM, TAB_N };
static void tab_exp(struct orc_session_s *s, int slot, double tau);
static void tab_totals(struct orc_session_s *s);
static void tab_normal(struct orc_session_s *s, int slot, double mu, double sigma);

/* ---------------------------------------------------------------- session ------------------------ */
orc_session *orc_new(const orc_config *c, const double *templates, const double *spe, const double *gains,
                     const double *thr_truth, const i64 *thr_zle, const double *lum_x, const double *lum_t,
                     const int16_t *noise)
{
    orc_session *s = (orc_session *)calloc(1, sizeof(orc_session));
    s->cur_tab = s->cur_tabb = s->cur_pzi = -1; s->cur_gg = -1;
    s->c = *c; s->templates = templates; s->spe = spe; s->gains = gains; s->thr_truth = thr_truth; s->thr_zle = thr_zle;
    s->lum_x = lum_x; s->lum_t = lum_t; s->noise = noise; s->save_full_truth = 1;
    tab_normal(s, TAB_TTS, c->tts_mean, c->tts_sigma); tab_exp(s, TAB_S1_EXP, c->s1_decay_time); tab_normal(s, TAB_S1_SPREAD, 0.0, c->s1_decay_spread);
    tab_exp(s, TAB_T1, c->t1_gas); tab_exp(s, TAB_T3, c->t3_gas); tab_normal(s, TAB_S2_SPREAD, 0.0, c->s2_time_spread);
    tab_totals(s);
    for (int r = 0; r < 10; r++) {              /* pulse.py:32 current_max */
        double m = templates[r * c->tlen];
        for (int k = 1; k < c->tlen; k++) if (templates[r * c->tlen + k] > m) m = templates[r * c->tlen + k];
        s->current_max[r] = m;
    }
    VEC_PUSH(s->pl_cur_off, i64, 0); VEC_PUSH(s->call_ph_off, i64, 0); VEC_PUSH(s->call_e_off, i64, 0);
    VEC_PUSH(s->dg_row_off, i64, 0); VEC_PUSH(s->row_data_off, i64, 0); VEC_PUSH(s->zl_data_off, i64, 0);
    return s;
}

void orc_set_ap_element(orc_session *s, int e, int n_bins_delay, int n_bins_amp, int amp_2d, int is_uniform,
                        double delay_bin, double amp_bin, const double *delay_cdf, const double *amp_cdf)
{
    orc_ap_element *a = &s->ap[e];
    a->n_bins_delay = n_bins_delay; a->n_bins_amp = n_bins_amp; a->amp_2d = amp_2d; a->is_uniform = is_uniform;
    a->delay_bin = delay_bin; a->amp_bin = amp_bin; a->delay_cdf = delay_cdf; a->amp_cdf = amp_cdf;
}

void orc_set_noise_float(orc_session *s, const double *noise) { s->noise_f = noise; if (noise && !s->noise) s->noise = (const int16_t *)noise; }
void orc_set_noise_override(orc_session *s, const i64 *ix, i64 n) { s->noise_override = ix; s->n_noise_override = n; }
void orc_set_save_full_truth(orc_session *s, int on) { s->save_full_truth = on; }

void orc_free(orc_session *s)
{
    void **ptrs[] = { (void **)&s->pl_ch.p, (void **)&s->pl_runset.p, (void **)&s->pl_left.p, (void **)&s->pl_right.p,
        (void **)&s->pl_cur_off.p, (void **)&s->pl_nph.p, (void **)&s->cur.p, (void **)&s->ph_t.p, (void **)&s->ph_ch.p,
        (void **)&s->ph_dpe.p, (void **)&s->ph_gain.p, (void **)&s->call_ph_off.p, (void **)&s->call_kind.p,
        (void **)&s->call_runset.p, (void **)&s->e_t.p, (void **)&s->call_e_off.p, (void **)&s->dg_left.p,
        (void **)&s->dg_right.p, (void **)&s->dg_first_pulse.p, (void **)&s->dg_n_pulses.p, (void **)&s->dg_ix_rand.p,
        (void **)&s->dg_row_off.p, (void **)&s->row_ch.p, (void **)&s->row_left.p, (void **)&s->row_right.p,
        (void **)&s->row_data_off.p, (void **)&s->row_data.p, (void **)&s->zl_digit.p, (void **)&s->zl_left.p,
        (void **)&s->zl_right.p, (void **)&s->zl_data_off.p, (void **)&s->zl_ch.p, (void **)&s->zl_data.p,
        (void **)&s->truth.p };
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); i++) free(*ptrs[i]);
    for (int q = 0; q < TAB_N; q++) { free(s->tab[q].cum); free(s->tab[q].thr); free(s->tab[q].alias); }
    for (i32 k = 0; k < s->n_xtab; k++) { free(s->xtab[k].cum); free(s->xtab[k].thr); free(s->xtab[k].alias); }
    free(s->xtab); free(s->prop_top); free(s->prop_bot); free(s->gg_inv);
    free(s);
}

/* ---------------------------------------------------------------- add_current -------------------- */
static i64 floordiv(i64 a, i64 b) { i64 q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
static i64 floormod(i64 a, i64 b) { i64 m = a % b; return (m != 0 && ((m < 0) != (b < 0))) ? m + b : m; }

typedef struct { i64 t; i64 i; } tkey;
static int tkey_cmp(const void *a, const void *b)
{
    const tkey *x = (const tkey *)a, *y = (const tkey *)b;
    if (x->t != y->t) return x->t < y->t ? -1 : 1;
    return x->i < y->i ? -1 : (x->i > y->i);
}

/* pulse.py:276-318  Pulse.add_current.  Photons are visited in ascending time; photons of equal ns are merged
 * (gains summed) and placed once: cur[start : start+tlen] += templates[t % dt] * gain_total, product and sum
 * rounded separately (numpy does not fuse them).  The reference's argsort is numpy's default introsort, whose
 * order among equal keys is unspecified; ties are taken here in input order. */
static void add_current_impl(const i64 *t, const double *g, i64 n, i64 pulse_left, i64 dt,
                             const double *templates, i64 tlen, double *cur, int fused)
{
    if (n == 0) return;
    tkey *k = (tkey *)malloc((size_t)n * sizeof(tkey));
    for (i64 i = 0; i < n; i++) { k[i].t = t[i]; k[i].i = i; }
    qsort(k, (size_t)n, sizeof(tkey), tkey_cmp);
    double gain_total = 0;
    i64 tmp = k[0].t;
    for (i64 j = 0; j <= n; j++) {
        if (j == n || k[j].t > tmp) {
            i64 start = floordiv(tmp, dt) - pulse_left, rem = floormod(tmp, dt);
            const double *T = templates + rem * tlen;
            if (fused) for (i64 q = 0; q < tlen; q++) cur[start + q] = fma(T[q], gain_total, cur[start + q]);      /* one rounding per term */
            else for (i64 q = 0; q < tlen; q++) {
                volatile double prod = T[q] * gain_total;      /* separate rounding of product and sum */
                cur[start + q] += prod;
            }
            if (j == n) break;
            gain_total = g[k[j].i]; tmp = k[j].t;
        } else {
            gain_total += g[k[j].i];
        }
    }
    free(k);
}
void orc_add_current(const i64 *t, const double *g, i64 n, i64 pulse_left, i64 dt,
                     const double *templates, i64 tlen, double *cur)
{
    add_current_impl(t, g, n, pulse_left, dt, templates, tlen, cur, 0);
}
/* the same walk with fused multiply-adds (wfs_config.fma): what the HIP kernels compute when the switch is on */
void orc_add_current_fma(const i64 *t, const double *g, i64 n, i64 pulse_left, i64 dt,
                         const double *templates, i64 tlen, double *cur)
{
    add_current_impl(t, g, n, pulse_left, dt, templates, tlen, cur, 1);
}

/* np.sum over a contiguous float64 array: numpy's pairwise summation (blocks of 128, 8 partial sums),
 * needed to reproduce raw_area of pulse.py:256-257 to the last bit */
static double np_pairwise_sum(const double *a, i64 n)
{
    if (n < 8) { double r = 0.; for (i64 i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8]; i64 i;
        for (int k = 0; k < 8; k++) r[k] = a[k];
        for (i = 8; i < n - (n % 8); i += 8) for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    i64 n2 = n / 2; n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* ---------------------------------------------------------------- Pulse.__call__ ----------------- */
/* pulse.py:81-144 (per-channel loop) + pulse.py:229-271 (add_truth).  Input photons are sorted by channel, times
 * are post transit-time-spread, gains already drawn (pulse.py:97-107).  n_dpe_ch = number of DPE photons in the
 * channel slice (only enters the truth quirk pulse.py:255).  Appends pulses to the cache and one truth record. */
void orc_pulse_call(orc_session *s, int kind, int runset, i64 n, const i64 *t, const int16_t *ch,
                    const uint8_t *dpe, const double *gain, int gains_preassigned)
{
    const orc_config *c = &s->c;
    double tr[12]; memset(tr, 0, sizeof tr);
    /* tr: n_photon n_pe n_photon_trigger n_pe_trigger raw_area raw_area_trigger, then the same for bottom */
    i64 a = 0;
    while (a < n) {
        i64 b = a; int channel = ch[a];
        while (b < n && ch[b] == channel) b++;
        i64 cnt = b - a;
        if (channel >= 0 && channel < c->n_tpc && s->gains[channel] == 0) { a = b; continue; }   /* turned off, pulse.py:89 */
        i64 n_dpe = 0;
        if (!gains_preassigned) for (i64 i = a; i < b; i++) n_dpe += dpe[i] ? 1 : 0;
        /* add_truth */
        double thr = s->thr_truth[channel]; i64 n_trig = 0, n_trig_dpe = 0;
        double *gtrig = (double *)malloc((size_t)cnt * 8);
        for (i64 i = a; i < b; i++) {
            i64 rem = floormod(t[i], c->dt);
            double amp = gain[i] * s->current_max[rem] * c->c2a;
            int above = amp > thr;
            if (above) { gtrig[n_trig++] = gain[i]; if (i - a < n_dpe) n_trig_dpe++; }
        }
        double sumg = np_pairwise_sum(gain + a, cnt), sumg_trig = np_pairwise_sum(gtrig, n_trig);
        free(gtrig);
        double vals[6] = { (double)cnt, (double)(cnt + n_dpe), (double)n_trig, (double)(n_trig + n_trig_dpe),
                           sumg / s->gains[channel], sumg_trig / s->gains[channel] };
        for (int f = 0; f < 6; f++) { tr[f] += vals[f]; if (channel >= c->n_top && channel <= c->last_bottom) tr[6 + f] += vals[f]; }
        if (!gains_preassigned) s->n_pe_total += cnt + n_dpe;
        /* tile bounds pulse.py:118-128 */
        i64 tmin = t[a], tmax = t[a];
        for (i64 i = a; i < b; i++) { if (t[i] < tmin) tmin = t[i]; if (t[i] > tmax) tmax = t[i]; }
        i64 left = floordiv(tmin, c->dt) - c->store_before - c->samples_before;
        i64 right = floordiv(tmax, c->dt) + c->store_after + c->samples_after;
        i64 len = right - left + 1;
        VEC_RESERVE(s->cur, double, len);
        double *cur = s->cur.p + s->cur.n; memset(cur, 0, (size_t)len * sizeof(double)); s->cur.n += len;
        add_current_impl(t + a, gain + a, cnt, left, c->dt, s->templates, c->tlen, cur, c->fma);
        VEC_PUSH(s->pl_ch, i32, channel); VEC_PUSH(s->pl_runset, i32, runset); VEC_PUSH(s->pl_left, i64, left);
        VEC_PUSH(s->pl_right, i64, right); VEC_PUSH(s->pl_nph, i64, cnt); VEC_PUSH(s->pl_cur_off, i64, s->cur.n);
        i64 end = right * c->dt;                                 /* rawdata.py:188-190 */
        if (!s->has_pulse || end > s->last_end) { s->last_end = end; s->has_pulse = 1; }
        a = b;
    }
    for (int f = 0; f < 12; f++) VEC_PUSH(s->truth, double, tr[f]);
    VEC_RESERVE(s->ph_t, i64, n); VEC_RESERVE(s->ph_ch, int16_t, n); VEC_RESERVE(s->ph_dpe, uint8_t, n); VEC_RESERVE(s->ph_gain, double, n);
    memcpy(s->ph_t.p + s->ph_t.n, t, (size_t)n * 8); s->ph_t.n += n;
    memcpy(s->ph_ch.p + s->ph_ch.n, ch, (size_t)n * 2); s->ph_ch.n += n;
    memcpy(s->ph_dpe.p + s->ph_dpe.n, dpe, (size_t)n); s->ph_dpe.n += n;
    memcpy(s->ph_gain.p + s->ph_gain.n, gain, (size_t)n * 8); s->ph_gain.n += n;
    VEC_PUSH(s->call_ph_off, i64, s->ph_t.n); VEC_PUSH(s->call_kind, i32, kind); VEC_PUSH(s->call_runset, i32, runset);
    VEC_PUSH(s->call_e_off, i64, s->e_t.n);
}

/* ---------------------------------------------------------------- digitise ----------------------- */
static double round_half_even(double x) { return nearbyint(x); }   /* np.around, default rounding mode */

/* utils.py:13-58 find_intervals_below_threshold, the sequential state machine as written there */
i64 orc_find_intervals_below_threshold(const i64 *w, i64 n, i64 threshold, i64 holdoff, i64 *result, i64 result_size)
{
    i64 last = n - 1, cur = 0, start = -1, end = -1; int in = 0;
    for (i64 i = 0; i < n; i++) {
        i64 x = w[i];
        if (x < threshold) { if (!in) { in = 1; start = i; } end = i; }
        if ((i == last && in) || (x >= threshold && i >= end + holdoff && in)) {
            in = 0; result[cur * 2] = start; result[cur * 2 + 1] = end; cur++;
            if (cur == result_size) { /* utils.py:54-55 writes past the buffer; never reached with a 50000 buffer */ return cur; }
        }
    }
    return cur;
}

/* rawdata.py:398-437 add_noise, the part after ix_rand is drawn: per masked channel noise[(ix_rand + ix - ch_left) mod N, ch]
 * is added to the samples ch_left..ch_right that exist (the reference guards against ix >= row length), channels past the
 * noise columns are skipped. */
static const double *g_noise_f = NULL;       /* set around add_noise_rows by the session that has a float noise array */
static void add_noise_rows(i64 *raw, i64 R, i64 L, const uint8_t *mask, const i64 *ml, const i64 *mr,
                           const int16_t *noise, i64 N, i64 noise_channels, i64 ix_rand)
{
    for (i64 ch = 0; ch < R; ch++) {
        if (ch >= noise_channels || !mask[ch]) continue;
        for (i64 ix = ml[ch]; ix <= mr[ch]; ix++) {
            if (ix >= L) continue;
            i64 in = ix_rand + ix - ml[ch];
            if (in >= N) in -= N * (in / N);
            if (g_noise_f) raw[ch * L + ix] = (i64)((double)raw[ch * L + ix] + g_noise_f[in * noise_channels + ch]);
            else raw[ch * L + ix] += noise[in * noise_channels + ch];
        }
    }
}
/* upper bound of np.random.randint(0, high) for ix_rand (rawdata.py:407-417); <= 0 means ix_rand = 0 */
i64 orc_noise_high(i64 R, const uint8_t *mask, const i64 *ml, const i64 *mr, i64 N)
{
    i64 nl = INT64_MAX, nr = INT64_MIN; int any = 0;
    for (i64 r = 0; r < R; r++) if (mask[r]) { any = 1; if (ml[r] < nl) nl = ml[r]; if (mr[r] > nr) nr = mr[r]; }
    if (!any) return -1;
    return (N - nr + nl - 1 < 0) ? N - 1 : N - nr + nl - 1;
}
void orc_add_noise(i64 *raw, i64 R, i64 L, const uint8_t *mask, const i64 *ml, const i64 *mr, const int16_t *noise, i64 N,
                   i64 noise_channels, i64 ix_rand)
{
    add_noise_rows(raw, R, L, mask, ml, mr, noise, N, noise_channels, ix_rand);
}
void orc_add_noise_float(i64 *raw, i64 R, i64 L, const uint8_t *mask, const i64 *ml, const i64 *mr, const double *noise, i64 N,
                         i64 noise_channels, i64 ix_rand)
{
    g_noise_f = noise;
    add_noise_rows(raw, R, L, mask, ml, mr, (const int16_t *)noise, N, noise_channels, ix_rand);
    g_noise_f = NULL;
}

/* rawdata.py:204-272 digitize_pulse_cache + :398-458 add_noise/add_baseline/digitizer_saturation, then
 * rawdata.py:274-311 ZLE over the same window.  Consumes all pulses cached since the previous digitise. */
void orc_digitize_and_zle(orc_session *s, u32 noise_gid)
{
    const orc_config *c = &s->c;
    i64 p0 = s->first_uncommitted_pulse, p1 = s->pl_ch.n;
    if (p1 == p0) return;
    i64 tw = c->trigger_window;
    i64 left = s->pl_left.p[p0], right = s->pl_right.p[p0];
    for (i64 p = p0; p < p1; p++) { if (s->pl_left.p[p] < left) left = s->pl_left.p[p]; if (s->pl_right.p[p] > right) right = s->pl_right.p[p]; }
    left -= tw; right += tw;
    if (floormod(left, 2) != 0) left -= 1;                        /* rawdata.py:221-222 */
    i64 L = right - left + 1, R = c->n_rows;
    i64 *raw = (i64 *)calloc((size_t)(R * L), sizeof(i64));       /* rawdata.py:224 */
    uint8_t *mask = (uint8_t *)calloc((size_t)R, 1);
    i64 *ml = (i64 *)malloc((size_t)R * 8), *mr = (i64 *)calloc((size_t)R, 8);
    for (i64 r = 0; r < R; r++) ml[r] = INT64_MAX;
    for (i64 p = p0; p < p1; p++) {
        int ch = s->pl_ch.p[p]; if (ch < 0) ch += (int)R;         /* python negative index, SURVEY B.12 */
        i64 pl = s->pl_left.p[p], pr = s->pl_right.p[p];
        mask[ch] = 1; if (pl < ml[ch]) ml[ch] = pl; if (pr > mr[ch]) mr[ch] = pr;
        const double *cur = s->cur.p + s->pl_cur_off.p[p];
        i64 *row = raw + ch * L + (pl - left);
        for (i64 i = 0; i <= pr - pl; i++) {
            i64 adc = -(i64)round_half_even(cur[i] * c->c2a);     /* rawdata.py:236, per pulse rounding */
            row[i] += adc;
            if (c->detector_nt) {
                i64 he = adc * c->he_factor;                      /* int(high_energy_deamplification_factor) */
                if (ch < c->n_top) raw[(c->he_first + ch) * L + (pl - left) + i] += he;
                else if (ch <= c->last_bottom) raw[c->sum_channel * L + (pl - left) + i] += he;
            }
        }
        if (c->detector_nt && ch < c->n_top) { int h = c->he_first + ch; mask[h] = 1; ml[h] = ml[ch]; mr[h] = mr[ch]; }
    }
    for (i64 r = 0; r < R; r++) if (mask[r]) { ml[r] -= left + tw; mr[r] -= left - tw; }   /* rawdata.py:258-259 */
    /* add_noise rawdata.py:398-437 */
    i64 ix_rand = -1;
    if (c->enable_noise) {
        i64 nl = INT64_MAX, nr = INT64_MIN; int any = 0;
        for (i64 r = 0; r < R; r++) if (mask[r]) { any = 1; if (ml[r] < nl) nl = ml[r]; if (mr[r] > nr) nr = mr[r]; }
        if (any) {
            i64 N = c->noise_len, high = (N - nr + nl - 1 < 0) ? N - 1 : N - nr + nl - 1;
            if (high <= 0) ix_rand = 0;
            else { u32 w[4]; draw(s, 0, noise_gid, 0, SITE_NOISE, w); ix_rand = (i64)(u53(w[0], w[1]) * (double)high); }
            if (s->noise_override && s->dg_left.n < s->n_noise_override && s->noise_override[s->dg_left.n] >= 0) ix_rand = s->noise_override[s->dg_left.n];
            g_noise_f = s->noise_f;
            add_noise_rows(raw, R, L, mask, ml, mr, s->noise, N, c->noise_channels, ix_rand);
            g_noise_f = NULL;
        }
    }
    i64 dg = s->dg_left.n;
    VEC_PUSH(s->dg_left, i64, left); VEC_PUSH(s->dg_right, i64, right); VEC_PUSH(s->dg_first_pulse, i64, p0);
    VEC_PUSH(s->dg_n_pulses, i64, p1 - p0); VEC_PUSH(s->dg_ix_rand, i64, ix_rand);
    i64 *itv = (i64 *)malloc(50000 * 2 * sizeof(i64));            /* rawdata.py:280 */
    for (i64 ch = 0; ch < R; ch++) {
        if (!mask[ch]) continue;
        i64 cl = ml[ch], cr = mr[ch];
        i64 *row = raw + ch * L;
        for (i64 ix = cl; ix <= cr; ix++) { row[ix] += c->baseline; if (row[ix] < 0) row[ix] = 0; }   /* :439-458 */
        i64 n = cr - cl + 1;
        VEC_PUSH(s->row_ch, i32, (i32)ch); VEC_PUSH(s->row_left, i64, cl); VEC_PUSH(s->row_right, i64, cr);
        VEC_RESERVE(s->row_data, i32, n);
        for (i64 i = 0; i < n; i++) s->row_data.p[s->row_data.n + i] = (i32)row[cl + i];
        s->row_data.n += n; VEC_PUSH(s->row_data_off, i64, s->row_data.n);
        /* ZLE rawdata.py:282-311 */
        i64 holdoff = tw + tw + 1;
        i64 nit = orc_find_intervals_below_threshold(row + cl, n, s->thr_zle[ch], holdoff, itv, 50000);
        for (i64 k = 0; k < nit; k++) {
            i64 a = itv[2 * k] - tw, b = itv[2 * k + 1] + tw;
            if (a < 0) a = 0; if (a > n - 1) a = n - 1; if (b < 0) b = 0; if (b > n - 1) b = n - 1;   /* np.clip */
            a = (i64)(ceil((double)a / 2.0) * 2); b = (i64)(floor((double)b / 2.0) * 2);           /* even landing */
            VEC_PUSH(s->zl_digit, i64, dg); VEC_PUSH(s->zl_ch, i32, (i32)ch);
            VEC_PUSH(s->zl_left, i64, left + cl + a); VEC_PUSH(s->zl_right, i64, left + cl + b);
            i64 m = b - a + 1; if (m < 0) m = 0;
            VEC_RESERVE(s->zl_data, i32, m);
            for (i64 i = 0; i < m; i++) s->zl_data.p[s->zl_data.n + i] = (i32)row[cl + a + i];
            s->zl_data.n += m; VEC_PUSH(s->zl_data_off, i64, s->zl_data.n);
        }
    }
    VEC_PUSH(s->dg_row_off, i64, s->row_ch.n);
    free(itv); free(raw); free(mask); free(ml); free(mr);
    s->first_uncommitted_pulse = p1;
}

/* schedulers: digitise the open window with its noise stream; note_cluster() names the stream after the first cluster
 * of the window that produced a pulse (clusters without pulses must not matter: the answer would otherwise depend on
 * how a run is cut into batches) */
static void digitize_window(orc_session *s) { orc_digitize_and_zle(s, s->win_gid); s->win_gid_set = 0; }
static void note_cluster(orc_session *s, i64 pulses_before, u32 first_gid)
{
    if (!s->win_gid_set && s->pl_ch.n > pulses_before) { s->win_gid = first_gid; s->win_gid_set = 1; }
}

/* ---------------------------------------------------------------- integer delay variates --------- */
/* The reference draws a float variate and truncates it to int64 (pulse.py:54-56 normal, s1.py:193-194 exponential and
 * normal, pulse.py:339-341 exponential * lifetime, s2.py:550 normal).  trunc(Y) is sampled here directly: from one
 * uniform by inverse CDF over cum[i] = P(trunc(Y) <= vmin + i) for the NORMAL terms (transit time, S1/S2 spread):
 * P(X <= k) = Phi((k+1-mu)/sigma) for k >= 0 and Phi((k-mu)/sigma) for k < 0 (the C cast truncates toward zero).
 * Exponential terms use the closed form trunc(-log(1-u) * tau). */
static void tab_normal(orc_session *s, int slot, double mu, double sigma)
{
    if (!(sigma > 0)) { s->tab[slot].cum = (double *)malloc(8); s->tab[slot].cum[0] = 1.0; s->tab[slot].n = 1; s->tab[slot].vmin = (i64)mu; return; }
    i64 lo = (i64)floor(mu - 8.5 * sigma) - 1, hi = (i64)ceil(mu + 8.5 * sigma) + 1, n = hi - lo + 1;
    double *c = (double *)malloc((size_t)n * 8);
    for (i64 k = lo; k <= hi; k++) { double x = ((double)(k >= 0 ? k + 1 : k) - mu) / sigma; c[k - lo] = 0.5 * erfc(-x / 1.4142135623730951); }
    c[n - 1] = 1.0;
    s->tab[slot].cum = c; s->tab[slot].n = n; s->tab[slot].vmin = lo;
}
static void tab_exp(orc_session *s, int slot, double tau)
{
    double *c = (double *)malloc(60000 * 8); i64 n = 0;
    if (!(tau > 0)) { c[n++] = 1.0; }
    else for (i64 k = 0; k < 60000; k++) { double v = -expm1(-(double)(k + 1) / tau); c[n++] = v; if (v >= 1.0) break; }
    c[n - 1] = 1.0;
    s->tab[slot].cum = c; s->tab[slot].n = n; s->tab[slot].vmin = 0;
}
static i64 sample_tab(const orc_session *s, int slot, double u)
{
    const double *c = s->tab[slot].cum; i64 lo = 0, hi = s->tab[slot].n - 1;
    while (lo < hi) { i64 mid = (lo + hi) >> 1; if (u < c[mid]) hi = mid; else lo = mid + 1; }      /* first i with u < cum[i] */
    return s->tab[slot].vmin + lo;
}
/* The delay of a photon is a SUM of independent terms, each truncated to an integer on its own, and only
 * the sum reaches the pulse.  The sum of independent integer variates is sampled from one uniform through the
 * convolution of their probability mass functions -- the same distribution as adding separately drawn terms:
 *   S1 (s1.py:193-194 + pulse.py:54-56):  trunc(Exp * s1_decay_time) + trunc(N(0, s1_decay_spread)) + trunc(N(tts))
 *   S2 (s2.py:338, pulse.py:339-341, s2.py:550, pulse.py:54-56):
 *        trunc(luminescence) + trunc(Exp * (t1 w.p. sf, else t3)) + trunc(N(0, s2_time_spread)) + trunc(N(tts))
 * pmf of trunc(np.interp(u, lum_x, lum_t)): P(trunc(L) <= k) = F(k + 1) for k >= 0, F(k) for k < 0 with
 * F(x) = P(L <= x) the inverse of the piecewise linear map u -> L. */
typedef struct { double *p; i64 n, vmin; } pmf_t;
static pmf_t pmf_of_tab(const orc_session *s, int slot)
{
    pmf_t r; r.n = s->tab[slot].n; r.vmin = s->tab[slot].vmin; r.p = (double *)malloc((size_t)r.n * 8);
    for (i64 i = 0; i < r.n; i++) r.p[i] = s->tab[slot].cum[i] - (i ? s->tab[slot].cum[i - 1] : 0.0);
    return r;
}
static pmf_t pmf_delta(i64 v) { pmf_t r; r.n = 1; r.vmin = v; r.p = (double *)malloc(8); r.p[0] = 1.0; return r; }
static pmf_t pmf_mix(pmf_t a, double wa, pmf_t b, double wb)
{
    pmf_t r; r.vmin = a.vmin < b.vmin ? a.vmin : b.vmin;
    i64 hi = (a.vmin + a.n > b.vmin + b.n) ? a.vmin + a.n : b.vmin + b.n;
    r.n = hi - r.vmin; r.p = (double *)calloc((size_t)r.n, 8);
    for (i64 i = 0; i < a.n; i++) r.p[a.vmin - r.vmin + i] += wa * a.p[i];
    for (i64 i = 0; i < b.n; i++) r.p[b.vmin - r.vmin + i] += wb * b.p[i];
    return r;
}
static pmf_t pmf_conv(pmf_t a, pmf_t b)
{
    pmf_t r; r.vmin = a.vmin + b.vmin; r.n = a.n + b.n - 1; r.p = (double *)calloc((size_t)r.n, 8);
    for (i64 i = 0; i < a.n; i++) { const double ai = a.p[i]; if (ai == 0.0) continue; for (i64 j = 0; j < b.n; j++) r.p[i + j] += ai * b.p[j]; }
    return r;
}
static double lum_cdf(const orc_session *s, double x)
{
    const double *xp = s->lum_x, *fp = s->lum_t; const int n = s->c.n_lum;
    if (x < fp[0]) return 0.0;
    if (x >= fp[n - 1]) return 1.0;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (fp[mid] <= x) lo = mid; else hi = mid; }
    return xp[lo] + (x - fp[lo]) / (fp[lo + 1] - fp[lo]) * (xp[lo + 1] - xp[lo]);
}
static void tab_lum(orc_session *s)
{
    const int n = s->c.n_lum;
    if (n < 2 || !s->lum_x || !s->lum_t) { s->tab[TAB_LUM].cum = (double *)malloc(8); s->tab[TAB_LUM].cum[0] = 1.0; s->tab[TAB_LUM].n = 1; s->tab[TAB_LUM].vmin = 0; return; }
    const i64 lo = (i64)floor(s->lum_t[0]) - 1, hi = (i64)ceil(s->lum_t[n - 1]) + 1;
    double *c = (double *)malloc((size_t)(hi - lo + 1) * 8);
    for (i64 k = lo; k <= hi; k++) c[k - lo] = lum_cdf(s, (double)(k >= 0 ? k + 1 : k));
    c[hi - lo] = 1.0;
    s->tab[TAB_LUM].cum = c; s->tab[TAB_LUM].n = hi - lo + 1; s->tab[TAB_LUM].vmin = lo;
}
static void tab_from_pmf(orc_session *s, int slot, pmf_t a)
{
    i64 first = 0; while (first < a.n - 1 && a.p[first] == 0.0) first++;
    double *c = (double *)malloc((size_t)(a.n - first) * 8); double acc = 0; i64 n = 0;
    for (i64 i = first; i < a.n; i++) { acc += a.p[i]; c[n++] = acc; if (acc >= 1.0) break; }
    while (n > 1 && c[n - 2] >= 1.0) n--;
    c[n - 1] = 1.0;
    s->tab[slot].cum = c; s->tab[slot].n = n; s->tab[slot].vmin = a.vmin + first;
}
/* Walker's alias method for the tables photons are drawn from (the summed delays): K = 2^lg >= max(n, 2) cells.  Vose's
 * construction in a fixed order: q[i] = p[i] * K; cells with q < 1 ("small") and the others ("large") go on two stacks
 * in ascending index order; the small cell on top gets {thr = floor(q * 2^32), alias = large cell on top}, the large
 * cell keeps (q_l + q_s) - 1 and moves to the small stack when that is below 1; leftovers keep their own outcome.
 * A draw uses ONE 32-bit word: its top lg bits pick the cell c, the remaining bits (left aligned) are compared with thr[c]:
 * the outcome is c if they are below, else alias[c]. */
static void alias_build(orc_tab *t)
{
    i64 K = 2; int lg = 1;
    while (K < t->n) { K <<= 1; lg++; }
    free(t->thr); free(t->alias);
    t->lg = lg; t->thr = (u32 *)malloc((size_t)K * 4); t->alias = (u32 *)malloc((size_t)K * 4);
    double *q = (double *)calloc((size_t)K, 8);
    u32 *small = (u32 *)malloc((size_t)K * 4), *large = (u32 *)malloc((size_t)K * 4); i64 ns = 0, nl = 0;
    for (i64 i = 0; i < t->n; i++) q[i] = (t->cum[i] - (i ? t->cum[i - 1] : 0.0)) * (double)K;
    for (i64 i = 0; i < K; i++) { if (q[i] < 1.0) small[ns++] = (u32)i; else large[nl++] = (u32)i; t->thr[i] = 0xffffffffu; t->alias[i] = (u32)i; }
    while (ns > 0 && nl > 0) {
        const u32 si = small[--ns], l = large[nl - 1];
        const double x = q[si] * 4294967296.0;
        t->thr[si] = x >= 4294967295.0 ? 0xffffffffu : (u32)x; t->alias[si] = l;
        q[l] = (q[l] + q[si]) - 1.0;
        if (q[l] < 1.0) { nl--; small[ns++] = l; }
    }
    free(q); free(small); free(large);
}
static i64 alias_sample(const orc_tab *t, u32 w)
{
    const u32 c = w >> (32 - t->lg);
    return t->vmin + (i64)((u32)(w << t->lg) < t->thr[c] ? c : t->alias[c]);
}
/* the distribution an alias table samples, cell by cell (tests: it must equal the table's pmf to 2^-32 per outcome) */
void orc_alias_pmf(const orc_session *s, int slot, i32 xtab, double *pmf_alias, double *pmf_cum, i64 cap, i64 *n_out, i64 *vmin_out)
{
    const orc_tab *t = xtab >= 0 ? &s->xtab[xtab] : &s->tab[slot];
    const i64 K = 1ll << t->lg;
    *n_out = t->n; *vmin_out = t->vmin;
    if (cap < K) return;
    for (i64 i = 0; i < K; i++) { pmf_alias[i] = 0.0; pmf_cum[i] = i < t->n ? t->cum[i] - (i ? t->cum[i - 1] : 0.0) : 0.0; }
    for (i64 i = 0; i < K; i++) {
        const double pa = t->alias[i] == (u32)i ? 1.0 : (double)t->thr[i] / 4294967296.0;
        pmf_alias[i] += pa / (double)K; pmf_alias[t->alias[i]] += (1.0 - pa) / (double)K;
    }
}

static void tab_totals(orc_session *s)
{
    const orc_config *c = &s->c;
    tab_lum(s);
    pmf_t tts = pmf_of_tab(s, TAB_TTS);
    /* S1 */
    pmf_t a = c->s1_simple ? pmf_of_tab(s, TAB_S1_EXP) : pmf_delta(0);
    if (c->s1_simple && c->s1_decay_spread != 0) { pmf_t b = pmf_of_tab(s, TAB_S1_SPREAD), r = pmf_conv(a, b); free(a.p); free(b.p); a = r; }
    { pmf_t r = pmf_conv(a, tts); free(a.p); tab_from_pmf(s, TAB_S1_TOTAL, r); free(r.p); }
    /* S2 */
    pmf_t t1 = pmf_of_tab(s, TAB_T1), t3 = pmf_of_tab(s, TAB_T3), lum = pmf_of_tab(s, TAB_LUM);
    pmf_t st = pmf_mix(t1, c->sf_gas, t3, 1.0 - c->sf_gas);
    pmf_t b = pmf_conv(st, lum);
    if (c->s2_time_model == 1 && c->s2_time_spread != 0) { pmf_t sp = pmf_of_tab(s, TAB_S2_SPREAD), r = pmf_conv(b, sp); free(b.p); free(sp.p); b = r; }
    { pmf_t r = pmf_conv(b, tts); tab_from_pmf(s, TAB_S2_TOTAL, r); free(r.p); }
    {   /* the same without the 'simple' luminescence term: base of the garfield tables (s2.py:512-528) */
        pmf_t nb = pmf_conv(st, tts);
        if (c->s2_time_model == 1 && c->s2_time_spread != 0) { pmf_t sp = pmf_of_tab(s, TAB_S2_SPREAD), r = pmf_conv(nb, sp); free(nb.p); free(sp.p); nb = r; }
        tab_from_pmf(s, TAB_S2_NOLUM, nb); free(nb.p);
    }
    free(t1.p); free(t3.p); free(lum.p); free(st.p); free(b.p); free(tts.p);
    alias_build(&s->tab[TAB_TTS]); alias_build(&s->tab[TAB_S1_TOTAL]); alias_build(&s->tab[TAB_S2_TOTAL]);
}

/* Model variants of S1.photon_timings (s1.py:162-238: 'custom' recoil models, optical propagation) and
 * S2.photon_timings (s2.py:504-557: garfield luminescence, optical propagation): every one of them adds one more independent
 * integer-truncated term.  Table k = base[k] (*) extra pmf k with base 0: transit time only, 1: the S1 terms of wfs_config,
 * 2: the S2 terms, 3: the S2 terms without the 'simple' luminescence.  An instruction names the table of its photons on
 * top-array channels and the one for bottom-array channels (the S2 propagation spline differs, s2.py:496-500). */
void orc_set_delay_models(orc_session *s, i32 n_tables, const i32 *base, const i64 *pmf_off, const double *pmf, const i32 *vmin)
{
    static const int slot_of_base[4] = {TAB_TTS, TAB_S1_TOTAL, TAB_S2_TOTAL, TAB_S2_NOLUM};
    for (i32 k = 0; k < s->n_xtab; k++) { free(s->xtab[k].cum); free(s->xtab[k].thr); free(s->xtab[k].alias); }
    free(s->xtab); s->xtab = NULL; s->n_xtab = n_tables;
    if (n_tables <= 0) return;
    s->xtab = calloc((size_t)n_tables, sizeof *s->xtab);
    for (i32 k = 0; k < n_tables; k++) {
        pmf_t e; e.n = pmf_off[k + 1] - pmf_off[k]; e.vmin = vmin[k]; e.p = (double *)(pmf + pmf_off[k]);
        pmf_t b = pmf_of_tab(s, slot_of_base[base[k]]), r = pmf_conv(b, e);
        /* tab_from_pmf writes s->tab[slot]: build in a scratch slot-shaped struct */
        i64 first = 0; while (first < r.n - 1 && r.p[first] == 0.0) first++;
        double *c = (double *)malloc((size_t)(r.n - first) * 8); double acc = 0; i64 n = 0;
        for (i64 i = first; i < r.n; i++) { acc += r.p[i]; c[n++] = acc; if (acc >= 1.0) break; }
        while (n > 1 && c[n - 2] >= 1.0) n--;
        c[n - 1] = 1.0;
        s->xtab[k].cum = c; s->xtab[k].n = n; s->xtab[k].vmin = r.vmin + first;
        alias_build(&s->xtab[k]);
        free(b.p); free(r.p);
    }
}
/* S1 optical propagation (s1.py:241-260): spline([z, u]) per photon, a RegularGridInterpolator (load_resource.py:356-357:
 * multilinear, extrapolating) over (z, u) for the top and the bottom array.  The host hands over the node values
 * [nz][nu], the u grid (u0, du) and per instruction the z cell and the normalised distance in it. */
void orc_set_s1_propagation(orc_session *s, i32 nz, i32 nu, double u0, double du, const double *top, const double *bot)
{
    free(s->prop_top); free(s->prop_bot); s->prop_top = s->prop_bot = NULL; s->prop_nz = nz; s->prop_nu = nu; s->prop_u0 = u0; s->prop_du = du;
    if (nz <= 0) return;
    s->prop_top = (double *)malloc((size_t)nz * nu * 8); memcpy(s->prop_top, top, (size_t)nz * nu * 8);
    s->prop_bot = (double *)malloc((size_t)nz * nu * 8); memcpy(s->prop_bot, bot, (size_t)nz * nu * 8);
}
/* per instruction (same indexing as the arrays of the next orc_simulate* call; the caller keeps them alive):
 * tab / tabb: delay table for top / bottom channels (-1: the default table of the type); pzi, pzf: z cell and fraction */
void orc_set_instruction_models(orc_session *s, i64 n, const i32 *tab, const i32 *tabb, const i32 *pzi, const double *pzf)
{
    s->n_ins_models = n; s->ins_tab = tab; s->ins_tabb = tabb; s->ins_pzi = pzi; s->ins_pzf = pzf;
}
/* garfield_gas_gap luminescence: timing_inv_cdf [n_gg][L] (load_resource.py:284-291); per instruction (same indexing as the
 * arrays of the next orc_simulate* call) the index of the table at or below its gas gap (-1: not this model) and
 * (gas gap - tabulated gas gap) / spacing (s2.py:474-476) */
void orc_set_gas_gap_model(orc_session *s, i32 n_gg, i32 L, const double *inv)
{
    free(s->gg_inv); s->gg_inv = NULL; s->gg_n = n_gg; s->gg_L = L;
    if (n_gg <= 0) return;
    s->gg_inv = (double *)malloc((size_t)n_gg * L * 8); memcpy(s->gg_inv, inv, (size_t)n_gg * L * 8);
}
void orc_set_instruction_gas_gap(orc_session *s, i64 n, const i32 *idx, const double *w) { s->n_ins_gg = n; s->ins_gg = idx; s->ins_ggw = w; }
/* excitation time of one photon, s2.py:443-446: a uniform position on the inverse CDF interpolated between the two tables */
static double gg_time(const orc_session *s, i32 lo, double wgt, u32 w)
{
    const i32 L = s->gg_L, hi = lo + 1 < s->gg_n ? lo + 1 : s->gg_n - 1;
    const double samples = ((double)w + 0.5) * (1.0 / 4294967296.0) * (double)(L - 2);
    const double fl = floor(samples); const i32 i0 = (i32)fl, i1 = (i32)ceil(samples);
    const double *A = s->gg_inv + (i64)lo * L, *B = s->gg_inv + (i64)hi * L;
    const double t1 = (B[i0] - A[i0]) * wgt + A[i0], t2 = (B[i1] - A[i1]) * wgt + A[i1];
    return (t2 - t1) * (samples - fl) + t1;
}

static void set_cur(orc_session *s, i64 i)
{
    s->cur_gg = (i < s->n_ins_gg && s->ins_gg && s->gg_inv) ? s->ins_gg[i] : -1; s->cur_ggw = s->cur_gg >= 0 ? s->ins_ggw[i] : 0.0;
    const int on = i < s->n_ins_models;
    s->cur_tab = on && s->ins_tab ? s->ins_tab[i] : -1; s->cur_tabb = on && s->ins_tabb ? s->ins_tabb[i] : s->cur_tab;
    s->cur_pzi = on && s->ins_pzi ? s->ins_pzi[i] : -1; s->cur_pzf = on && s->ins_pzf ? s->ins_pzf[i] : 0.0;
}
/* multilinear interpolation as scipy's RegularGridInterpolator evaluates it: sum over the cell's corners of value * weights */
static double s1_propagation(const orc_session *s, int bottom, i32 zi, double zf, double u)
{
    const double *T = bottom ? s->prop_bot : s->prop_top; const i32 nu = s->prop_nu;
    i32 ui = (i32)floor((u - s->prop_u0) / s->prop_du);
    if (ui < 0) ui = 0; if (ui > nu - 2) ui = nu - 2;
    const double uf = (u - (s->prop_u0 + (double)ui * s->prop_du)) / s->prop_du;
    const double *r0 = T + (i64)zi * nu + ui, *r1 = r0 + nu;
    double v = r0[0] * ((1.0 - zf) * (1.0 - uf));
    v += r0[1] * ((1.0 - zf) * uf);
    v += r1[0] * (zf * (1.0 - uf));
    v += r1[1] * (zf * uf);
    return v;
}

/* ---------------------------------------------------------------- photon generation -------------- */
static int cmp_ch_stable(const void *a, const void *b)
{
    const i64 *x = (const i64 *)a, *y = (const i64 *)b;     /* key = channel << 40 | index */
    return *x < *y ? -1 : (*x > *y);
}

typedef struct { vec_i64 t; vec_i16 ch; vec_u8 dpe; vec_f64 gain; vec_i32 g1, g2; } photon_buf;

/* np.random.choice(channels, p=...) == searchsorted(cdf, u, side='right') on the normalised cumulative sum
 * (s1.py:154-158, s2.py:673-677); cdf row prepared by the host exactly as numpy does. */
__attribute__((unused)) static int channel_from_cdf(const double *cdf, int n, double u)
{
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
    return lo < n ? lo : n - 1;
}

static double interp_lum(const orc_session *s, double u)
{
    /* np.interp(u, y / y[-1], t), s2.py:338 */
    const double *xp = s->lum_x, *fp = s->lum_t; int n = s->c.n_lum;
    if (u <= xp[0]) return fp[0];
    if (u >= xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (xp[mid] <= u) lo = mid; else hi = mid; }
    if (xp[lo] == u) return fp[lo];
    double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
    return slope * (u - xp[lo]) + fp[lo];
}

/* SPE indices and double-PE flag from one word (pulse.py:76-79, 97-103, 226): w * 2000 = g * 2^32 + frac; g + 1 =
 * int(u * 2000) + 1 is the first index; frac < floor(p_dpe * 2^32) is the double-PE trial, and given that, frac is uniform
 * below the threshold: int(frac * 2000 / thr) + 1 is the second index. */
static inline void gain_code(const orc_session *s, u32 w, int *g1, int *g2)
{
    const u64 prod = (u64)w * 2000u, thr = bern_threshold(s->c.p_dpe);
    const u32 frac = (u32)prod;
    *g1 = (int)(prod >> 32) + 1; *g2 = 0;
    if ((u64)frac < thr) { u32 v = (u32)((double)frac * (2000.0 / (double)thr)) + 1u; *g2 = (int)(v > 2000u ? 2000u : v); }
}
/* One photon: pulse.py:53-56 (TTS), :76-79 (DPE), :97-103 (SPE gain, idx = int(u*2000)+1), plus the timing terms of
 * s1.py:180-194 (simple) or s2.py:504-557 (luminescence simple + singlet/triplet + spread + electron time).
 * Every term is truncated to int64 on its own before it is added: the sum is drawn from the table of the sum.
 * RNG spec v6: photon P of its instruction owns word P & 3 of the calls (em_base, gid, P >> 2, SITE_CH / _DELAY / _GAIN). */
/* Channel of a photon: Walker's alias method over the 2^lg >= n_tpc cells of the instruction's cumulative row, built by alias_build
 * exactly as the device builds its cells (k_chan_alias) -- RNG spec v8.  The row is that of the instruction in hand (chan_row_set). */
static orc_tab g_chan_tab;
static void chan_row_set(const orc_session *s, const double *cdf)
{
    g_chan_tab.cum = (double *)cdf; g_chan_tab.n = s->c.n_tpc; g_chan_tab.vmin = 0;
    alias_build(&g_chan_tab);
}

/* the distribution the channel alias cells of a cumulative row encode (tests), and draws from it exactly as one_photon() makes them */
void orc_chan_alias_pmf(const orc_session *s, const double *cdf, double *pmf_alias)
{
    chan_row_set(s, cdf);
    const i64 K = 1ll << g_chan_tab.lg;
    for (i64 i = 0; i < s->c.n_tpc; i++) pmf_alias[i] = 0.0;
    for (i64 i = 0; i < K; i++) {
        const double pa = g_chan_tab.alias[i] == (u32)i ? 1.0 : (double)g_chan_tab.thr[i] / 4294967296.0;
        if (i < s->c.n_tpc) pmf_alias[i] += pa / (double)K;
        if (g_chan_tab.alias[i] < (u32)s->c.n_tpc) pmf_alias[g_chan_tab.alias[i]] += (1.0 - pa) / (double)K;
    }
}
void orc_sample_channels(const orc_session *s, const double *cdf, i64 n, i32 *out)
{
    chan_row_set(s, cdf);
    for (i64 i = 0; i < n; i++) { u32 C[4]; draw(s, 0, 777777u, (u32)(i >> 2), SITE_CH, C); out[i] = (i32)alias_sample(&g_chan_tab, C[i & 3]); }
}

static void one_photon(const orc_session *s, int is_s2, u32 emitter, u32 gid, u32 item, u32 em_base, u32 P, i64 t0,
                       const double *cdf, i64 *t_out, int *ch_out, int *dpe_out, double *gain_out)
{
    const orc_config *c = &s->c;
    u32 C[4], D[4], G[4];
    draw(s, em_base, gid, P >> 2, SITE_CH, C); draw(s, em_base, gid, P >> 2, SITE_DELAY, D); draw(s, em_base, gid, P >> 2, SITE_GAIN, G);
    int ch = (int)alias_sample(&g_chan_tab, C[P & 3u]);
    int g1, g2; gain_code(s, G[P & 3u], &g1, &g2);
    int is_dpe = g2 != 0;
    i64 t;                                                /* all delay terms from one table, see tab_totals / orc_set_delay_models */
    if (s->cur_tab >= 0) t = t0 + alias_sample(&s->xtab[ch >= c->n_top ? s->cur_tabb : s->cur_tab], D[P & 3u]);
    else t = t0 + alias_sample(&s->tab[is_s2 ? TAB_S2_TOTAL : TAB_S1_TOTAL], D[P & 3u]);
    if (!is_s2 && s->prop_top && s->cur_pzi >= 0) {       /* s1.py:185-188; prop_time is an int64 array: the assignment truncates */
        u32 X[4]; draw(s, emitter, gid, item, SITE_PH_X, X);
        t += (i64)s1_propagation(s, ch >= c->n_top, s->cur_pzi, s->cur_pzf, ((double)X[0] + 0.5) * (1.0 / 4294967296.0));
    }
    int sc = c->n_spe_channels > ch ? ch : 0;
    const double *row = s->spe + (i64)sc * 2001;
    double G0 = s->gains[ch], gain = G0 * row[g1];
    if (is_dpe) gain += G0 * row[g2];
    *t_out = t; *ch_out = ch; *dpe_out = is_dpe; *gain_out = gain;
}

static void sort_by_channel_and_call(orc_session *s, int kind, int runset, photon_buf *pb, int preassigned)
{
    i64 n = pb->t.n;
    i64 *key = (i64 *)malloc((size_t)(n ? n : 1) * 8);
    for (i64 i = 0; i < n; i++) key[i] = ((i64)pb->ch.p[i] << 40) | i;
    qsort(key, (size_t)n, 8, cmp_ch_stable);
    i64 *t = (i64 *)malloc((size_t)(n ? n : 1) * 8); int16_t *ch = (int16_t *)malloc((size_t)(n ? n : 1) * 2);
    uint8_t *dpe = (uint8_t *)malloc((size_t)(n ? n : 1)); double *g = (double *)malloc((size_t)(n ? n : 1) * 8);
    for (i64 i = 0; i < n; i++) { i64 j = key[i] & ((1ll << 40) - 1); t[i] = pb->t.p[j]; ch[i] = pb->ch.p[j]; dpe[i] = pb->dpe.p[j]; g[i] = pb->gain.p[j]; }
    orc_pulse_call(s, kind, runset, n, t, ch, dpe, g, preassigned);
    free(key); free(t); free(ch); free(dpe); free(g);
}

/* afterpulse.py:172-249 PMT_Afterpulse.photon_afterpulse for the photons of the pulse call just made */
static void pmt_afterpulse_call(orc_session *s, int runset, const u32 *gid_of, i64 call_index, const u32 *emitter_of, const u32 *item_of)
{
    const orc_config *c = &s->c;
    i64 a = s->call_ph_off.p[call_index], b = s->call_ph_off.p[call_index + 1];
    if (b == a) return;                                    /* afterpulse.py:162-164 */
    photon_buf pb; memset(&pb, 0, sizeof pb);
    for (int e = 0; e < c->n_ap_elements; e++) {
        const orc_ap_element *ap = &s->ap[e];
        for (i64 i = a; i < b; i++) {
            int ch = s->ph_ch.p[i]; u32 w[4];
            /* RNG spec v10: the top 32 bits of the first uniform of elements 4k .. 4k + 3 are the words of ONE call (the device screens
             * four elements with it); the low bits and the second uniform come from the element's own call */
            u32 sw[4]; draw(s, emitter_of[i - a], gid_of[i - a], item_of[i - a], SITE_AP_SCREEN + (u32)(e >> 2), sw);
            draw(s, emitter_of[i - a], gid_of[i - a], item_of[i - a], SITE_AP + (u32)e, w);
            double rU0 = 1.0 - u53(sw[e & 3], w[1]), rU1 = 1.0 - u53(w[2], w[3]);
            const double *dc = ap->delay_cdf + (i64)ch * ap->n_bins_delay;
            double prob = dc[ap->n_bins_delay - 1];
            rU0 /= c->pmt_ap_modifier;
            if (s->ph_dpe.p[i]) rU0 /= 2;
            if (!(rU0 <= prob)) continue;
            double delay, amp;
            if (ap->is_uniform) {
                u32 x[4]; draw(s, emitter_of[i - a], gid_of[i - a], item_of[i - a], SITE_AP_X + (u32)e, x);
                double lo = dc[0], hi = dc[1];
                delay = (lo + (hi - lo) * u53(x[0], x[1])) * ap->delay_bin; amp = 1.0;
            } else {
                int best = 0; double bd = fabs(dc[0] - rU0);                  /* np.argmin(|cdf - u|) : first minimum */
                for (int k = 1; k < ap->n_bins_delay; k++) { double d = fabs(dc[k] - rU0); if (d < bd) { bd = d; best = k; } }
                delay = best * ap->delay_bin - c->pmt_ap_t_modifier;
                const double *ac = ap->amp_2d ? ap->amp_cdf + (i64)ch * ap->n_bins_amp : ap->amp_cdf;
                int ba = 0; double bad = fabs(ac[0] - rU1);
                for (int k = 1; k < ap->n_bins_amp; k++) { double d = fabs(ac[k] - rU1); if (d < bad) { bad = d; ba = k; } }
                amp = ba * ap->amp_bin;
            }
            /* afterpulse.py:235: int64 timing + float delay; the sum is float and is cast back to int64 in
             * pulse.py:130 (astype(np.int64)) after the bounds were taken on the float values */
            double tf = (double)s->ph_t.p[i] + delay;
            VEC_PUSH(pb.t, i64, (i64)tf); VEC_PUSH(pb.ch, int16_t, (int16_t)ch); VEC_PUSH(pb.dpe, uint8_t, 0);
            VEC_PUSH(pb.gain, double, s->gains[ch] * amp);
        }
    }
    sort_by_channel_and_call(s, 3, runset, &pb, 1);
    free(pb.t.p); free(pb.ch.p); free(pb.dpe.p); free(pb.gain.p);
}

/* One Pulse call can cover several instructions (save_full_truth=False groups them, rawdata.py:108-127): the photons of
 * the instructions are appended in instruction order (s1.py:95-103 / s2.py:107-136 loop over the instruction array) and
 * every photon remembers its Philox coordinates (gid, emitter, item) for the afterpulse stage. */
typedef struct { photon_buf pb; vec_i64 gid, em, it; } call_ctx;
static void ctx_push(call_ctx *x, i64 t, int ch, int dpe, double g, u32 gid, i64 em, i64 it)
{
    VEC_PUSH(x->pb.t, i64, t); VEC_PUSH(x->pb.ch, int16_t, (int16_t)ch); VEC_PUSH(x->pb.dpe, uint8_t, (uint8_t)dpe); VEC_PUSH(x->pb.gain, double, g);
    VEC_PUSH(x->gid, i64, (i64)gid); VEC_PUSH(x->em, i64, em); VEC_PUSH(x->it, i64, it);
}
/* s1.py:117-238 for one instruction: Binomial(amp, ly) hits as Bernoulli trials, then one_photon() each */
static i64 gen_s1(orc_session *s, call_ctx *x, u32 gid, i64 time, i64 amp, double p_hit, const double *cdf)
{
    u64 T = bern_threshold(p_hit); i64 n_hits = 0; u32 w[4];
    chan_row_set(s, cdf);
    for (i64 j = 0; j < amp; j++) {                   /* s1.py:133 Binomial(amp, ly) as a sum of Bernoulli trials */
        if ((j & 3) == 0) draw(s, 0, gid, (u32)(j >> 2), SITE_S1_HIT, w);
        n_hits += (u64)w[j & 3] < T;
    }
    for (i64 k = 0; k < n_hits; k++) {
        i64 t; int ch, dpe; double g;
        one_photon(s, 0, 0, gid, (u32)k, 0, (u32)k, time, cdf, &t, &ch, &dpe, &g);
        ctx_push(x, t, ch, dpe, g, gid, 0, k);
    }
    return n_hits;
}
/* s2.py:73-136 S2.__call__ (luminescence 'simple') for one instruction.  Candidate electron j < amp survives with
 * probability cy (s2.py:254 Binomial as Bernoulli trials); survivors draw s2.py:280-282 arrival time and
 * s2.py:308-310 photon count; photons as one_photon(). */
/* Tile-local generation (RNG spec v9, v11: tiles of any size; device: wfs_tilegen.h).  With one secondary gain for all electrons (s2_gain_spread == 0) the
 * photon counts per (electron, channel) are independent Poisson(g p_ch) variates (Poisson splitting of s2.py:308 + the
 * np.random.choice of :673), so a tile (instruction, channel) holds Poisson(n_surviving g p_ch) photons, each from a uniformly
 * drawn surviving electron -- the same joint distribution, drawn tile by tile.  The rule below is evaluated identically on the device. */
static int fuse_eligible(const orc_session *s, int type, u32 em_base, i64 amp, double sc_gain, const double *cdf)
{
    const orc_config *c = &s->c;
    if (!c->tile_gen || c->gain_spread != 0.0) return 0;      /* (save_full_truth off: the callers pass type -1 for an instruction that shares its Pulse call) */
    if (s->cur_gg >= 0) return 0;                   /* 'garfield_gas_gap' luminescence: the instruction's photons share a mean (s2.py:447-450) */
    if (type != 2 || em_base != 0u || amp <= 0 || !(sc_gain > 0)) return 0;
    double pmax = 0.0;
    for (int ch = 0; ch < c->n_tpc; ch++) { const double p = cdf[ch] - (ch ? cdf[ch - 1] : 0.0); pmax = p > pmax ? p : pmax; }
    const double lam = (double)amp * sc_gain * pmax;
    return lam >= (double)c->tile_gen_min && lam < 1.0e9;      /* (tile counts are 32-bit; which kernel makes a tile is the device's business) */
}
static i64 gen_s2(orc_session *s, call_ctx *x, int type, u32 gid, u32 em_base, i64 time, i64 amp, double cy, double drift_mean, double drift_spread,
                  double sc_gain, const double *cdf)
{
    const orc_config *c = &s->c;
    u64 T = bern_threshold(cy); u32 w[4]; i64 n0 = x->pb.t.n; u32 P = 0;       /* P: photons of this instruction so far */
    const int fused = fuse_eligible(s, type, em_base, amp, sc_gain, cdf);
    vec_i64 surv; memset(&surv, 0, sizeof surv);                                /* fused: arrival times of the surviving electrons, in candidate order */
    chan_row_set(s, cdf);
    vec_f64 lum; memset(&lum, 0, sizeof lum);                                   /* garfield gas gap: excitation time of every photon */
    for (i64 j = 0; j < amp; j++) {
        if ((j & 3) == 0) draw(s, em_base, gid, (u32)(j >> 2), SITE_S2_SURVIVE, w);
        if (!((u64)w[j & 3] < T)) continue;
        u32 A[4], B[4]; double z_drift, z_gain;
        const u32 je = em_base + (u32)j;            /* emitter id in the Philox counters */
        draw(s, je, gid, 0, SITE_EL_A, A);
        draw(s, je, gid, 0, SITE_EL_B, B);
        box_muller(B, &z_drift, &z_gain);
        double timing = -log(1.0 - u53(A[0], A[1])) * c->trap_time;
        timing += drift_mean + drift_spread * z_drift;
        i64 et = time + (i64)timing;
        if (fused) { VEC_PUSH(s->e_t, i64, et); VEC_PUSH(surv, i64, et); continue; }
        i64 nph = poisson_any(s, je, gid, sc_gain);
        nph += (i64)(0.0 + c->gain_spread * z_gain);
        if (nph < 0) nph = 0;
        VEC_PUSH(s->e_t, i64, et);
        for (i64 m = 0; m < nph; m++) {
            i64 t; int ch, dpe; double g;
            if (s->cur_gg >= 0) { u32 Lw[4]; draw(s, em_base, gid, P >> 2, SITE_LUM, Lw); VEC_PUSH(lum, double, gg_time(s, s->cur_gg, s->cur_ggw, Lw[P & 3u])); }
            one_photon(s, 1, je, gid, (u32)m, em_base, P++, et, cdf, &t, &ch, &dpe, &g);
            ctx_push(x, t, ch, dpe, g, gid, (i64)je, m);
        }
    }
    if (fused) {
        for (int ch = 0; ch < c->n_tpc && surv.n > 0; ch++) {
            /* the instruction's table for this channel's array when timing-model variants are set (s2.py:504-557), else all S2 terms */
            const orc_tab *tab = s->cur_tab >= 0 ? &s->xtab[ch >= c->n_top ? s->cur_tabb : s->cur_tab] : &s->tab[TAB_S2_TOTAL];
            const double p = cdf[ch] - (ch ? cdf[ch - 1] : 0.0);
            const double lam = (double)(i32)surv.n * sc_gain * p;
            const u32 c0 = em_base + (u32)ch;
            i64 N = poisson_site(s, c0, gid, SITE_TILE_N, lam);
            if (N > 0x3fffffffLL) N = 0x3fffffffLL;
            int sc = c->n_spe_channels > ch ? ch : 0;
            const double *row = s->spe + (i64)sc * 2001;
            for (i64 q = 0; q < N; q++) {                                       /* photon q of the tile: word q & 3 of three calls */
                u32 E[4], D[4], G[4];
                draw(s, c0, gid, (u32)(q >> 2), SITE_TILE_E, E); draw(s, c0, gid, (u32)(q >> 2), SITE_TILE_DELAY, D); draw(s, c0, gid, (u32)(q >> 2), SITE_TILE_GAIN, G);
                const u32 e = (u32)(((u64)E[q & 3] * (u64)(u32)surv.n) >> 32);
                const i64 t = surv.p[e] + alias_sample(tab, D[q & 3]);
                int g1, g2; gain_code(s, G[q & 3], &g1, &g2);
                double gain = s->gains[ch] * row[g1];
                if (g2) gain += s->gains[ch] * row[g2];
                ctx_push(x, t, ch, g2 != 0, gain, gid, (i64)c0, q);                   /* (c0, gid, q): the photon's coordinates for its PMT afterpulse draws */
            }
        }
        free(surv.p);
    }
    if (lum.n) {
        /* s2.py:447-450: T - mean(T) over the instruction's photons, then the int64 cast of photon_timings (s2.py:532-533).
         * The mean is taken in fixed point (2^-20 ns) so that it does not depend on the order of summation. */
        i64 sum = 0;
        for (i64 k = 0; k < lum.n; k++) sum += llrint(lum.p[k] * 1048576.0);
        const double mean = (double)sum / (double)lum.n / 1048576.0;
        for (i64 k = 0; k < lum.n; k++) x->pb.t.p[n0 + k] += (i64)(lum.p[k] - mean);
    }
    free(lum.p);
    return x->pb.t.n - n0;
}
/* the Pulse call over the collected photons (+ its PMT afterpulse call, rawdata.py:176-178) */
static void finish_call(orc_session *s, int kind, int runset, call_ctx *x)
{
    i64 n = x->pb.t.n; i64 *key = (i64 *)malloc((size_t)(n ? n : 1) * 8);
    u32 *g2 = (u32 *)malloc((size_t)(n ? n : 1) * 4), *em2 = (u32 *)malloc((size_t)(n ? n : 1) * 4), *it2 = (u32 *)malloc((size_t)(n ? n : 1) * 4);
    for (i64 i = 0; i < n; i++) key[i] = ((i64)x->pb.ch.p[i] << 40) | i;
    qsort(key, (size_t)n, 8, cmp_ch_stable);            /* channel-sorted order, as Pulse.__call__ sees the photons */
    for (i64 i = 0; i < n; i++) { i64 q = key[i] & ((1ll << 40) - 1); g2[i] = (u32)x->gid.p[q]; em2[i] = (u32)x->em.p[q]; it2[i] = (u32)x->it.p[q]; }
    i64 call = s->call_kind.n;
    sort_by_channel_and_call(s, kind, runset, &x->pb, 0);
    if (s->c.enable_pmt_ap) pmt_afterpulse_call(s, runset, g2, call, em2, it2);
    free(key); free(g2); free(em2); free(it2);
    free(x->pb.t.p); free(x->pb.ch.p); free(x->pb.dpe.p); free(x->pb.gain.p); free(x->gid.p); free(x->em.p); free(x->it.p);
    memset(x, 0, sizeof *x);
}

/* rawdata.py:475-493 RawDataOptical.sim_primary for one instruction: supplied photons (ns relative to the instruction,
 * channel), cut to 0 <= t < cutoff, sorted by channel, then the plain Pulse.__call__ (transit time, DPE, SPE gain). */
i64 orc_optical(orc_session *s, u32 gid, int runset, i64 time, i64 n, const i64 *t_rel, const i32 *chan, i64 cutoff)
{
    const orc_config *c = &s->c;
    photon_buf pb; memset(&pb, 0, sizeof pb);
    for (i64 k = 0; k < n; k++) {
        if (t_rel[k] < 0 || t_rel[k] >= cutoff) continue;
        u32 B[4];
        draw(s, 0, gid, (u32)k, SITE_PH, B);
        int g1, g2; gain_code(s, B[1], &g1, &g2);
        int is_dpe = g2 != 0;
        i64 t = time + t_rel[k];
        t += alias_sample(&s->tab[TAB_TTS], B[0]);
        int ch = chan[k];
        int sc = c->n_spe_channels > ch ? ch : 0;
        const double *row = s->spe + (i64)sc * 2001;
        double G = s->gains[ch], gain = G * row[g1];
        if (is_dpe) gain += G * row[g2];
        VEC_PUSH(pb.t, i64, t); VEC_PUSH(pb.ch, int16_t, (int16_t)ch); VEC_PUSH(pb.dpe, uint8_t, (uint8_t)is_dpe); VEC_PUSH(pb.gain, double, gain);
    }
    i64 np = pb.t.n;
    sort_by_channel_and_call(s, 0, runset, &pb, 0);
    free(pb.t.p); free(pb.ch.p); free(pb.dpe.p); free(pb.gain.p);
    return np;
}

/* scheduler for optical instructions (all type 1, rawdata.py:38-157 with RawDataOptical.sim_primary) */
void orc_simulate_optical(orc_session *s, i64 n, const i64 *time, const u32 *gid, const i32 *first, const i32 *last,
                          const i32 *channels, const i64 *timings, i64 cutoff)
{
    const orc_config *c = &s->c;
    if (n == 0) return;
    tkey *ord = (tkey *)malloc((size_t)n * sizeof(tkey));
    for (i64 i = 0; i < n; i++) { ord[i].t = time[i]; ord[i].i = i; }
    qsort(ord, (size_t)n, sizeof(tkey), tkey_cmp);
    i64 a = 0; int runset = 0;
    while (a < n) {
        i64 b = a + 1;
        while (b < n && !((double)(ord[b].t - ord[b - 1].t) > c->rext)) b++;
        if (s->has_pulse && (double)(ord[a].t - s->last_end) > c->rext) digitize_window(s);
        const i64 np0 = s->pl_ch.n;
        for (i64 k = a; k < b; k++) {
            i64 i = ord[k].i;
            orc_optical(s, gid[i], runset++, time[i], last[i] - first[i], timings + first[i], channels + first[i], cutoff);
        }
        note_cluster(s, np0, gid[ord[a].i]);
        a = b;
    }
    digitize_window(s);
    free(ord);
}

/* ---------------------------------------------------------------- scheduler ---------------------- */
/* rawdata.py:38-157 RawData.__call__ with electron afterpulses off: instructions are sorted by
 * time - z/v*[S2], split where the gap exceeds rext; clusters run in order, S1 run-sets then S2 run-sets (one
 * instruction per run-set, save_full_truth); the cache is digitised when the next cluster starts more than rext
 * after the end of the last pulse (rawdata.py:96-98) and at the end (rawdata.py:154-155). */
void orc_simulate(orc_session *s, i64 n, const int8_t *type, const i64 *time, const float *z, const i32 *amp,
                  const u32 *gid, const double *p_hit, const double *drift_mean, const double *drift_spread,
                  const double *sc_gain, const i32 *cdf_row, const double *cdf_table, const u32 *em_base)
{
    const orc_config *c = &s->c;
    if (n == 0) return;
    i64 *it = (i64 *)malloc((size_t)n * 8); tkey *ord = (tkey *)malloc((size_t)n * sizeof(tkey));
    for (i64 i = 0; i < n; i++) {
        /* rawdata.py:61: time + (z / v * (type % 2 - 1)).astype(int64); z is a float32 array and v a python float,
         * so numpy evaluates the quotient and the product in float32 */
        int sgn = ((type[i] % 2 + 2) % 2) - 1;
        volatile float q = z[i] / (float)c->drift_velocity;
        volatile float qs = q * (float)sgn;
        it[i] = time[i] + (i64)qs;
        ord[i].t = it[i]; ord[i].i = i;
    }
    qsort(ord, (size_t)n, sizeof(tkey), tkey_cmp);
    i64 a = 0; int runset = 0;
    while (a < n) {
        i64 b = a + 1;
        while (b < n && !((double)(ord[b].t - ord[b - 1].t) > c->rext)) b++;
        if (s->has_pulse && (double)(ord[a].t - s->last_end) > c->rext) digitize_window(s);    /* rawdata.py:96-98 */
        const i64 np0 = s->pl_ch.n;
        static const int ptypes[4] = {1, 2, 4, 6};              /* rawdata.py:102: S1, S2, photo-ionisation electrons, gate electrons */
        for (int pq = 0; pq < 4; pq++) {
            const int ptype = ptypes[pq];
            /* run sets (rawdata.py:106-127): every S1 / S2 on its own, or -- save_full_truth off -- S1s whose keys are at most
             * 100 ns apart / S2s at most int(0.2 / v) ns apart in one Pulse call; electron afterpulses: one call per cluster */
            const i64 gap = ptype == 1 ? 100 : (i64)(0.2 / c->drift_velocity);
            call_ctx x; memset(&x, 0, sizeof x); int open = 0; i64 last_key = 0;
            for (i64 k = a; k < b; k++) {
                i64 i = ord[k].i; if (type[i] != ptype) continue;
                const int starts = !open || (ptype <= 2 && (s->save_full_truth || ord[k].t - last_key > gap));      /* this instruction opens a Pulse call */
                if (open && starts) { finish_call(s, ptype, runset++, &x); open = 0; }
                /* alone in its call (the tile-local generator's condition, as on the device: k_fuse_decide): it opens one and the next
                 * instruction of its type in the cluster, if any, opens another */
                int alone = starts && ptype == 2;
                if (alone && !s->save_full_truth)
                    for (i64 q = k + 1; q < b; q++) if (type[ord[q].i] == ptype) { alone = ord[q].t - ord[k].t > gap; break; }
                const double *cdf = cdf_table + (i64)cdf_row[i] * c->n_tpc;
                set_cur(s, i);
                if (ptype == 1) gen_s1(s, &x, gid[i], time[i], amp[i], p_hit[i], cdf);
                else gen_s2(s, &x, alone ? ptype : -1, gid[i], em_base ? em_base[i] : 0u, time[i], amp[i], p_hit[i], drift_mean[i], drift_spread[i], sc_gain[i], cdf);
                open = 1; last_key = ord[k].t;
            }
            if (open) finish_call(s, ptype == 6 ? 5 : ptype, runset++, &x);      /* call kinds: 1 S1, 2 S2, 3 PMT afterpulse, 4 PI electrons, 5 gate electrons */
        }
        note_cluster(s, np0, gid[ord[a].i]);
        a = b;
    }
    digitize_window(s);                                       /* rawdata.py:154-155 */
    free(it); free(ord);
}

/* The same with the schedule given: instructions in processing order, cluster[] non-decreasing, tmin[] the key the
 * window rule sees (rawdata.py:96-98; the host raises it for clusters that are always preceded by a digitisation,
 * rawdata.py:148-149), run_set[] non-decreasing inside a cluster.  Used for runs with electron afterpulses, whose
 * order comes from the host's replay of the feedback loop (wfsim_amd/scheduler.py: feedback_schedule). */
void orc_simulate_scheduled(orc_session *s, i64 n, const int8_t *type, const i64 *time, const i32 *amp, const u32 *gid,
                            const double *p_hit, const double *drift_mean, const double *drift_spread, const double *sc_gain,
                            const i32 *cdf_row, const double *cdf_table, const u32 *em_base, const i32 *cluster, const i64 *tmin,
                            const i32 *run_set)
{
    const orc_config *c = &s->c;
    if (n == 0) return;
    i64 a = 0;
    while (a < n) {
        i64 b = a + 1, cmin = tmin[a];
        while (b < n && cluster[b] == cluster[a]) { if (tmin[b] < cmin) cmin = tmin[b]; b++; }
        if (s->has_pulse && (double)(cmin - s->last_end) > c->rext) digitize_window(s);
        const i64 np0 = s->pl_ch.n;
        i64 k = a;
        while (k < b) {
            i64 e = k + 1;
            while (e < b && run_set[e] == run_set[k]) e++;
            call_ctx x; memset(&x, 0, sizeof x);
            for (i64 i = k; i < e; i++) {
                const double *cdf = cdf_table + (i64)cdf_row[i] * c->n_tpc;
                set_cur(s, i);
                if (type[i] == 1) gen_s1(s, &x, gid[i], time[i], amp[i], p_hit[i], cdf);
                /* an instruction that shares its Pulse call is never generated tile by tile; one that is alone in its call follows the
                 * usual rule (the device: k_fuse_decide, set i = {instruction i}) */
                else gen_s2(s, &x, e - k == 1 ? type[i] : -1, gid[i], em_base ? em_base[i] : 0u, time[i], amp[i], p_hit[i], drift_mean[i], drift_spread[i], sc_gain[i], cdf);
            }
            finish_call(s, type[k] == 6 ? 5 : type[k], run_set[k], &x);
            k = e;
        }
        note_cluster(s, np0, gid[a]);
        a = b;
    }
    digitize_window(s);
}

/* ---------------------------------------------------------------- record packing ----------------- */
/* strax_interface.py:391-436: one ZLE interval -> ceil(len/110) raw_records (244-byte packed layout). */
i64 orc_pack_records(const orc_session *s, i64 samples_per_record, uint8_t *out, i64 capacity)
{
    i64 nrec = 0, rec_bytes = 24 + 2 * samples_per_record;
    for (i64 k = 0; k < s->zl_ch.n; k++) {
        i64 left = s->zl_left.p[k], plen = s->zl_right.p[k] - left + 1;
        if (plen <= 0) continue;
        i64 need = (plen + samples_per_record - 1) / samples_per_record;
        const i32 *d = s->zl_data.p + s->zl_data_off.p[k];
        for (i64 i = 0; i < need; i++, nrec++) {
            if (!out) continue;
            if (nrec >= capacity) return -1;
            uint8_t *r = out + nrec * rec_bytes; memset(r, 0, (size_t)rec_bytes);
            i64 tm = s->c.dt * (left + samples_per_record * i);
            i32 len = (i32)((plen < samples_per_record * (i + 1) ? plen : samples_per_record * (i + 1)) - samples_per_record * i);
            int16_t dt16 = (int16_t)s->c.dt, ch16 = (int16_t)s->zl_ch.p[k], ri = (int16_t)i; i32 pl32 = (i32)plen;
            memcpy(r, &tm, 8); memcpy(r + 8, &len, 4); memcpy(r + 12, &dt16, 2); memcpy(r + 14, &ch16, 2);
            memcpy(r + 16, &pl32, 4); memcpy(r + 20, &ri, 2);
            for (i32 q = 0; q < len; q++) { int16_t v = (int16_t)d[samples_per_record * i + q]; memcpy(r + 24 + 2 * q, &v, 2); }
        }
    }
    return nrec;
}

/* ---------------------------------------------------------------- accessors ---------------------- */
#define GETTER(name, field, T) T *orc_##name(orc_session *s, i64 *n) { *n = s->field.n; return s->field.p; }
GETTER(pl_ch, pl_ch, i32) GETTER(pl_runset, pl_runset, i32) GETTER(pl_left, pl_left, i64) GETTER(pl_right, pl_right, i64)
GETTER(pl_cur_off, pl_cur_off, i64) GETTER(pl_nph, pl_nph, i64) GETTER(cur, cur, double)
GETTER(ph_t, ph_t, i64) GETTER(ph_ch, ph_ch, int16_t) GETTER(ph_dpe, ph_dpe, uint8_t) GETTER(ph_gain, ph_gain, double)
GETTER(call_ph_off, call_ph_off, i64) GETTER(call_kind, call_kind, i32) GETTER(call_runset, call_runset, i32)
GETTER(e_t, e_t, i64) GETTER(call_e_off, call_e_off, i64)
GETTER(dg_left, dg_left, i64) GETTER(dg_right, dg_right, i64) GETTER(dg_first_pulse, dg_first_pulse, i64)
GETTER(dg_n_pulses, dg_n_pulses, i64) GETTER(dg_ix_rand, dg_ix_rand, i64) GETTER(dg_row_off, dg_row_off, i64)
GETTER(row_ch, row_ch, i32) GETTER(row_left, row_left, i64) GETTER(row_right, row_right, i64)
GETTER(row_data_off, row_data_off, i64) GETTER(row_data, row_data, i32)
GETTER(zl_digit, zl_digit, i64) GETTER(zl_ch, zl_ch, i32) GETTER(zl_left, zl_left, i64) GETTER(zl_right, zl_right, i64)
GETTER(zl_data_off, zl_data_off, i64) GETTER(zl_data, zl_data, i32) GETTER(truth, truth, double)
i64 orc_n_pe(const orc_session *s) { return s->n_pe_total; }

/* stand-alone samplers used by the distribution tests: the individual random terms exactly as one_photon() and
 * orc_s2() compute them (same draw sites), so each can be compared with a histogram of the reference's own draws.
 * kind: 0 luminescence delay (s2.py:338), 1 gas singlet/triplet delay (pulse.py:339-341), 2 transit time (pulse.py:54-56),
 *       3 S1 'simple' delay (s1.py:193-194), 4 electron arrival (s2.py:280-282, p0 = drift mean, p1 = drift spread),
 *       5 / 6 the summed delay of an S1 / S2 photon (tab_totals), 7 / 8 trunc(Exp * t3) / trunc(np.interp) evaluated directly,
 *       9 trunc(Exp * t3) from its table */
void orc_sample_term(orc_session *s, int kind, i64 n, double p0, double p1, i64 *out)
{
    const orc_config *c = &s->c;
    for (i64 i = 0; i < n; i++) {
        u32 A[4], B[4]; u32 em = (u32)(i >> 20), item = (u32)(i & 0xfffff), gid = 777u;
        draw(s, em, gid, item, SITE_CH, A); draw(s, em, gid, item, SITE_DELAY, B);
        double z0, z1;
        if (kind == 0) out[i] = sample_tab(s, TAB_LUM, u53(B[0], B[1]));
        else if (kind == 1) out[i] = sample_tab(s, ((u64)B[3] < bern_threshold(c->sf_gas)) ? TAB_T1 : TAB_T3, u53(B[0], B[1]));
        else if (kind == 2) out[i] = alias_sample(&s->tab[TAB_TTS], B[0]);                /* as photons with given times draw it */
        else if (kind == 10) out[i] = sample_tab(s, TAB_TTS, u53(B[0], B[1]));
        else if (kind == 3) out[i] = sample_tab(s, TAB_S1_EXP, u53(B[0], B[1])) + sample_tab(s, TAB_S1_SPREAD, u53(A[0], A[1]));
        else if (kind == 5) out[i] = alias_sample(&s->tab[TAB_S1_TOTAL], B[0]);          /* as one_photon() draws them */
        else if (kind == 6) out[i] = alias_sample(&s->tab[TAB_S2_TOTAL], B[0]);
        else if (kind == 7) out[i] = (i64)(-log(1.0 - u53(B[0], B[1])) * c->t3_gas);          /* the reference's expressions on the same uniform, */
        else if (kind == 8) out[i] = (i64)interp_lum(s, u53(B[0], B[1]));                      /* to check the tables sample by sample */
        else if (kind == 9) out[i] = sample_tab(s, TAB_T3, u53(B[0], B[1]));
        else {
            draw(s, (u32)i, gid, 0, SITE_EL_A, A); draw(s, (u32)i, gid, 0, SITE_EL_B, B);
            box_muller(B, &z0, &z1);
            double timing = -log(1.0 - u53(A[0], A[1])) * c->trap_time;
            timing += p0 + p1 * z0;
            out[i] = (i64)timing;
        }
    }
}

/* the delay of a photon exactly as one_photon() adds it: table `tab` (-1: the default of the type), S1 propagation cell */
void orc_sample_delay(orc_session *s, i64 n, int is_s2, i32 tab, int bottom, i32 pzi, double pzf, i64 *out)
{
    for (i64 i = 0; i < n; i++) {
        u32 B[4], X[4]; draw(s, 0, 424242u, (u32)(i >> 2), SITE_DELAY, B); draw(s, 0, 424242u, (u32)i, SITE_PH_X, X);
        i64 t = tab >= 0 ? alias_sample(&s->xtab[tab], B[i & 3]) : alias_sample(&s->tab[is_s2 ? TAB_S2_TOTAL : TAB_S1_TOTAL], B[i & 3]);
        if (!is_s2 && s->prop_top && pzi >= 0) t += (i64)s1_propagation(s, bottom, pzi, pzf, ((double)X[0] + 0.5) * (1.0 / 4294967296.0));
        out[i] = t;
    }
}
void orc_sample_poisson(orc_session *s, double lam, i64 n, i64 *out) { for (i64 i = 0; i < n; i++) out[i] = poisson_any(s, (u32)i, 12345u, lam); }

/* electrons of one S2 instruction as gen_s2 and the device see them (tests of the transverse diffusion, s2.py:560-613): survival flag
 * (s2.py:254) and the two standard normals of the radial / azimuthal displacement */
void orc_sample_diffusion(orc_session *s, u32 gid, u32 em_base, i64 amp, double cy, uint8_t *survive, double *z0, double *z1)
{
    const u64 T = bern_threshold(cy); u32 w[4];
    for (i64 j = 0; j < amp; j++) {
        if ((j & 3) == 0) draw(s, em_base, gid, (u32)(j >> 2), SITE_S2_SURVIVE, w);
        survive[j] = (u64)w[j & 3] < T;
        u32 B[4]; draw(s, em_base + (u32)j, gid, 0, SITE_EL_DIFF, B);
        box_muller(B, &z0[j], &z1[j]);
    }
}

/* the garfield gas gap luminescence term of n photons of one instruction exactly as gen_s2 adds it (tests) */
void orc_sample_gas_gap(orc_session *s, i64 n, i32 lo, double wgt, i64 *out)
{
    double *T = (double *)malloc((size_t)(n ? n : 1) * 8); i64 sum = 0;
    for (i64 k = 0; k < n; k++) { u32 w[4]; draw(s, 0, 515151u, (u32)(k >> 2), SITE_LUM, w); T[k] = gg_time(s, lo, wgt, w[k & 3]); sum += llrint(T[k] * 1048576.0); }
    const double mean = n ? (double)sum / (double)n / 1048576.0 : 0.0;
    for (i64 k = 0; k < n; k++) out[k] = (i64)(T[k] - mean);
    free(T);
}
