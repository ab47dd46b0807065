// wfs_device.h -- device-side helpers of the MI355X WFSim hot path (gfx950 only).
//
// RNG: Philox4x32-10 (Salmon et al., SC'11), counter = (emitter, instruction gid, item, site), key = seed.
// The stream layout is specified in DESIGN.md "RNG streams"; the CPU oracle implements the same layout
// independently (oracle/wfsim_oracle.c) so that GPU and oracle results can be compared photon by photon.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef long long i64;
typedef unsigned long long u64;
typedef int i32;
typedef unsigned int u32;

#define WFS_MAX_CH 1024          // channels that fit the LDS histograms of the bucketing kernels
#define WFS_MAX_AP 8
#define WFS_DT 10                // sample_duration [ns]; wfs_create rejects anything else

enum WfsSite : u32 {
    SITE_S1_HIT = 1, SITE_S2_SURVIVE = 2, SITE_EL_A = 3, SITE_EL_B = 4, SITE_EL_POIS = 5,
    SITE_PH_A = 16, SITE_PH_B = 17, SITE_AP = 32, SITE_AP_X = 48, SITE_NOISE = 64
};

struct u32x4 { u32 x, y, z, w; };
struct __attribute__((aligned(8))) PhotonRec { i32 t; u32 code; };   // one bucketed photon: ns relative to its set, SPE codes

__device__ __forceinline__ u32x4 philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;     // one v_mad_u64_u32 each
        const u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n2 = (u32)(p0 >> 32) ^ c3 ^ k1;
        c0 = n0; c1 = (u32)p1; c2 = n2; c3 = (u32)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(u32 a, u32 b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

__device__ __forceinline__ u64 bern_threshold(double p)
{
    if (!(p > 0)) return 0;
    if (p >= 1) return 4294967296ull;
    return (u64)(p * 4294967296.0);
}

// first variate only (the second one is not needed when its spread is zero: (i64)(0.0 + 0 * z) == 0)
__device__ __forceinline__ double box_muller_z0(u32x4 w)
{
    double u1 = u53(w.x, w.y), u2 = u53(w.z, w.w);
    double r = sqrt(-2.0 * log(1.0 - u1));
    return r * cos(6.283185307179586476925286766559 * u2);
}

__device__ __forceinline__ void box_muller(u32x4 w, double &z0, double &z1)
{
    double u1 = u53(w.x, w.y), u2 = u53(w.z, w.w);
    double r = sqrt(-2.0 * log(1.0 - u1));
    double a = 6.283185307179586476925286766559 * u2;
    double s, c;
    sincos(a, &s, &c);
    z0 = r * c; z1 = r * s;
}

// Integer-valued normal delays.  The reference draws a normal variate and truncates it to int64 before adding it
// (pulse.py:54-56 transit time, s1.py:194 and s2.py:550 spreads).  trunc(Y) is a discrete variate; it is sampled here by
// inverse CDF from one uniform: cum[i] = P(X <= vmin + i), result = vmin + (first i with u < cum[i]).  The tables are
// built on the host (wfs_engine.hip, build_time_tables) from erfc; the CPU oracle builds its own the same way.
// (Exponential delays use the closed form trunc(-log(1-u) * tau).)
#ifndef DISC_G
#define DISC_G 4096
#endif
struct DiscTab { const double *cum; const unsigned short *guide; i32 vmin, n; };

// Lookup in two steps so that several lookups of one photon can have their loads in flight together:
// disc_begin issues the two guide loads, disc_finish the (usually 4-entry) window of cumulative probabilities.
struct DiscReq { const double *cum; int lo, hi; double u; };

__device__ __forceinline__ DiscReq disc_begin(const DiscTab &t, double u)
{
    const int c = (int)(u * DISC_G);
    DiscReq r; r.cum = t.cum; r.u = u; r.lo = t.guide[c]; r.hi = t.guide[c + 1];       // answer in [lo, hi]
    return r;
}

__device__ __forceinline__ i64 disc_finish(const DiscTab &t, DiscReq r)
{
    int lo = r.lo, hi = r.hi;
    while (hi - lo > 3) { const int mid = (lo + hi) >> 1; if (r.u < r.cum[mid]) hi = mid; else lo = mid + 1; }
    // at most 4 candidates left: fetch them together (cum[n-1] == 1 > u, indices clamped to the table)
    const int n1 = t.n - 1;
    const double c0 = r.cum[lo], c1 = r.cum[lo + 1 < n1 ? lo + 1 : n1], c2 = r.cum[lo + 2 < n1 ? lo + 2 : n1];
    const int k = r.u < c0 ? 0 : (r.u < c1 ? 1 : (r.u < c2 ? 2 : 3));
    return (i64)t.vmin + lo + k;
}

__device__ __forceinline__ i64 sample_disc(const DiscTab &t, double u) { return disc_finish(t, disc_begin(t, u)); }

// python-style floor division / modulo on int64 (numpy // and % on int64, pulse.py:305-306)
__host__ __device__ __forceinline__ i64 floordiv(i64 a, i64 b) { i64 q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
__host__ __device__ __forceinline__ i64 floormod(i64 a, i64 b) { i64 m = a % b; return (m != 0 && ((m < 0) != (b < 0))) ? m + b : m; }

// Everything a kernel needs, passed by value (fits the kernarg segment).
struct WfsDev {
    // scalars (wfs_config)
    i32 dt, samples_before, samples_after, store_before, store_after, tlen, tw, baseline, n_rows;
    i32 n_tpc, n_top, he_first, he_factor, last_bottom, detector_nt, enable_noise, s1_simple, s2_time_model, enable_pmt_ap;
    i32 n_spe, n_lum, noise_len, noise_channels, n_ap, he_rows /* HE rows materialised */, row_slots /* per group */;
    double c2a, tts_mean, tts_sigma, p_dpe, s1_decay_time, s1_decay_spread, sf_gas, t1_gas, t3_gas, s2_time_spread;
    double trap_time, gain_spread, pmt_ap_modifier, pmt_ap_t_modifier, rext;
    u32 k0, k1;
    u64 thr_dpe;                               // Bernoulli threshold on a 32-bit word: floor(p * 2^32)
    double current_max[10];
    DiscTab tab_tts, tab_s1, tab_s2;           // inverse-CDF tables of integer delays: transit time alone, all terms of an S1 / S2 photon
    // tables
    const double *templates, *spe, *gains, *thr_truth, *lum_x, *lum_t;
    const i64 *thr_zle;
    const int16_t *noise;
};
