// wfs_device.h -- device-side helpers of the MI355X WFSim hot path (gfx950 only).
//
// RNG: Philox4x32-10 (Salmon et al., SC'11), counter = (emitter, instruction gid, item, site), key = seed.
// The stream layout is specified in DESIGN.md "RNG streams" (spec v6: channel, delay and gain words of a photon come from
// three streams indexed by the photon's position P among its instruction's photons, four photons per call -- 0.75 calls
// per photon); the CPU oracle implements the same layout independently (oracle/wfsim_oracle.c) so that GPU and oracle
// results can be compared photon by photon.
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
    SITE_EL_DIFF = 6,    // per electron (emitter, gid, 0): Box-Muller pair -> radial / azimuthal transverse diffusion (s2.py:588-589)
    // photon streams, counter (em_base, gid, P >> 2, site), P = index of the photon among its instruction's photons; photon P owns word P & 3
    SITE_DELAY = 16,     // -> summed delay (alias table)
    SITE_CH = 17,        // -> channel
    SITE_GAIN = 18,      // -> first SPE index, double-PE flag, second SPE index
    SITE_PH = 19,        // photons that arrive with time and channel (optical input), counter (0, gid, item): x -> transit time, y -> gains
    SITE_PH_X = 20,      // per photon (emitter, gid, item), S1 optical propagation only: x -> spline coordinate
    SITE_LUM = 21,       // photon stream (as SITE_DELAY): position on the excitation-time inverse CDF ('garfield_gas_gap' luminescence)
    // tile-local generation (spec v9, wfs_tilegen.h): counter (em_base + channel, gid, item, site)
    SITE_TILE_N = 24,    // photons of the tile: Poisson draw, item = iteration of the sampler
    SITE_TILE_E = 25,    // item = P >> 2 (P: photon of the tile), word P & 3 -> surviving electron
    SITE_TILE_DELAY = 26,    // -> summed delay (alias table)
    SITE_TILE_GAIN = 27,     // -> SPE indices and double-PE flag
    SITE_AP = 32, SITE_AP_SCREEN = 40, SITE_AP_X = 48, SITE_NOISE = 64
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

// Integer-valued delays.  The reference draws float variates and truncates each to int64 before adding it (pulse.py:54-56
// transit time, s1.py:193-194, s2.py:338, pulse.py:339-341, s2.py:550); only the SUM reaches the pulse, and the sum of
// independent integer variates is one discrete variate whose probability mass function the host builds by convolution
// (wfs_engine.hip, build_time_tables).  It is sampled with Walker's alias method from ONE 32-bit word: K = 2^k >= n cells,
// cell c holds {thr, alias}; the word's top k bits pick the cell, the remaining bits (left aligned) are compared with thr.
// One 8-byte gather per photon (an inverse-CDF search needs a guide cell and three cumulative values: five gathers).
// The table is built by the same sequential algorithm on the host and in the CPU oracle.
struct AliasTab { const uint2 *cell; i32 vmin, shift; };      // shift = 32 - k

__device__ __forceinline__ u32 alias_cell(const AliasTab &t, u32 w) { return w >> t.shift; }
__device__ __forceinline__ i32 alias_pick(const AliasTab &t, u32 w, u32 c, uint2 e) { return t.vmin + (i32)(((w << (32 - t.shift)) < e.x) ? c : e.y); }
__device__ __forceinline__ i32 alias_sample(const AliasTab &t, u32 w) { const u32 c = alias_cell(t, w); return alias_pick(t, w, c, t.cell[c]); }

// SPE indices and double-PE flag of a photon from ONE word (pulse.py:76-79, 97-103, 226): w * 2000 = g * 2^32 + frac;
// g + 1 = int(u * 2000) + 1 is the first index; frac (the 32 low bits, equidistributed for every g) < thr is the double-PE
// Bernoulli trial, and given frac < thr it is uniform on [0, thr): int(frac * 2000 / thr) + 1 is the second index.
// Returns g1 | g2 << 16 with g2 = 0 when the photon makes a single PE.
__device__ __forceinline__ u32 gain_code(u32 w, u64 thr_dpe, double dpe_inv)
{
    const u64 prod = (u64)w * 2000u;
    const u32 g1 = (u32)(prod >> 32) + 1u, frac = (u32)prod;
    if (!((u64)frac < thr_dpe)) return g1;
    u32 g2 = (u32)((double)frac * dpe_inv) + 1u;
    g2 = g2 > 2000u ? 2000u : g2;
    return g1 | (g2 << 16);
}

// One term of Pulse.add_current (pulse.py:303-318): current += template * gain.  FMA = false: numpy's two roundings (the product,
// then the sum; the library is compiled with -ffp-contract=off, so the two stay apart) -- currents bit-exact with the reference.
// FMA = true (wfs_config.fma): one v_fma_f64, one rounding; half the f64 issue slots of every gather loop.
template <bool FMA> __device__ __forceinline__ double mac(double a, double b, double c)
{
    if constexpr (FMA) return __builtin_fma(a, b, c);
    else { const double prod = a * b; return c + prod; }
}

// Wave votes straight on the lane mask: HIP's __ballot / __any / __all take an int, which costs a v_cndmask (bool -> 0 / 1) and a v_cmp per
// vote on the vector unit; the builtin compares into the scalar mask directly.  (Votes are over the ACTIVE lanes, as the HIP ones.)
__device__ __forceinline__ u64 ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool any64(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ bool all64(bool p) { return __builtin_amdgcn_ballot_w64(!p) == 0ull; }

// Index of the wave inside its workgroup as a SCALAR: threadIdx.x >> 6 is the same in all 64 lanes but the compiler cannot know, and
// everything computed from it (row descriptors, base pointers, loop bounds of the wave-per-row kernels) would sit in vector
// registers and be recomputed by the vector unit; through readfirstlane it becomes SGPR work and scalar loads.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// Wave-wide reductions on the DPP path (quad permutes, half-row and row mirrors: VALU moves, no LDS crossbar round trips as with
// the shuffles; readlane joins the four 16-lane rows).  Every lane -- and the scalar result -- ends up with the value of all 64.
template <int CTRL> __device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = dpp_mov<CTRL>((int)(u32)b), hi = dpp_mov<CTRL>((int)(u32)((u64)b >> 32));
    return __longlong_as_double((long long)(((u64)(u32)hi << 32) | (u32)lo));
}
// lane ^ J of the wave without a trip through the LDS crossbar where the hardware has a permute for it: quad permutes (J = 1, 2) and the
// rotation of a 16-lane row by 8 are DPP moves on the vector unit, J = 4 / 16 swizzles (LDS unit, no address register), J = 32 the
// generic shuffle.  All 64 lanes must be active (the callers sit in wave-uniform code).
template <int J> __device__ __forceinline__ u32 lane_xor(u32 v)
{
    if constexpr (J == 1) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);          // quad_perm [1, 0, 3, 2]
    else if constexpr (J == 2) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);     // quad_perm [2, 3, 0, 1]
    else if constexpr (J == 8) return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);    // row_ror:8
    else if constexpr (J == 4) return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                       // bit mode: and 0x1f, xor 4
    else if constexpr (J == 16) return (u32)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                      // bit mode: and 0x1f, xor 16
    else return (u32)__shfl_xor((int)v, J, 64);
}
// one compare-exchange step of a bitonic network on keys held one per lane (ascending over the wave)
template <int K, int J> __device__ __forceinline__ u32 bitonic_step(u32 key, int lane)
{
    const u32 other = lane_xor<J>(key);
    const bool up = (lane & K) == 0, low = (lane & J) == 0;
    return (low == up) ? (key < other ? key : other) : (key > other ? key : other);
}
template <int K, int J = (K >> 1)> __device__ __forceinline__ u32 bitonic_merge(u32 key, int lane)
{
    key = bitonic_step<K, J>(key, lane);
    if constexpr (J > 1) return bitonic_merge<K, (J >> 1)>(key, lane);
    else return key;
}

// (every DPP move is made ONCE, by all lanes, before its value is used: inside a divergent region a DPP read of a switched-off
// lane returns the old value instead)
#define WFS_DPP_STEPS(T, COMBINE) \
    { const T o = dpp_mov<0xB1>(v); v = COMBINE; } { const T o = dpp_mov<0x4E>(v); v = COMBINE; } \
    { const T o = dpp_mov<0x141>(v); v = COMBINE; } { const T o = dpp_mov<0x140>(v); v = COMBINE; }
__device__ __forceinline__ int wave_sum(int v)
{
    WFS_DPP_STEPS(int, v + o)
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_min(int v)
{
    WFS_DPP_STEPS(int, min(v, o))
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max(int v)
{
    WFS_DPP_STEPS(int, max(v, o))
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// (a fixed order of additions: the result does not depend on anything but the 64 values)
__device__ __forceinline__ double wave_sum(double v)
{
    WFS_DPP_STEPS(double, v + o)
    const long long b = __double_as_longlong(v);
    const int lo = (int)(u32)b, hi = (int)(u32)((u64)b >> 32);
    double r[4];
#pragma unroll
    for (int q = 0; q < 4; q++) r[q] = __longlong_as_double((long long)(((u64)(u32)__builtin_amdgcn_readlane(hi, 16 * q) << 32) | (u32)__builtin_amdgcn_readlane(lo, 16 * q)));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// In-kernel phase stamps (diagnostic builds, -DWFS_STAMPS; MI355X_MICROARCH.md "In-kernel stamps"): thread 0 of every
// workgroup adds the shader cycles since its previous stamp to slot i.  Production builds compile them away.
#ifdef WFS_STAMPS
#define STAMP_INIT unsigned long long st_last_ = __builtin_amdgcn_s_memtime()
#define STAMP(d, i) do { if (threadIdx.x == 0 && (d).stamps) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    unsigned long long *row_ = (d).stamps + (size_t)(blockIdx.x & 4095u) * 64; \
    atomicAdd(&row_[i], t_ - st_last_); atomicAdd(&row_[(i) + 32], 1ull); st_last_ = t_; } } while (0)
#define STAMP_WAIT() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")      // (charges the phase with the latency of what it issued)
#else
#define STAMP_INIT
#define STAMP(d, i)
#define STAMP_WAIT()
#endif

// python-style floor division / modulo on int64 (numpy // and % on int64, pulse.py:305-306)
__host__ __device__ __forceinline__ i64 floordiv(i64 a, i64 b) { i64 q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
__host__ __device__ __forceinline__ i64 floormod(i64 a, i64 b) { i64 m = a % b; return (m != 0 && ((m < 0) != (b < 0))) ? m + b : m; }

// A lane of k_zle reads 4 consecutive noise samples from any start below noise_len: every channel row of the device copy is
// followed by its own first NOISE_PAD samples, so the read never meets the wrap (rawdata.py:433-434 indexes modulo noise_len).
constexpr int NOISE_PAD = 4;
constexpr int WFS_SPR = 110;                   // samples per strax record (strax DEFAULT_RECORD_LENGTH; a compile-time constant: the divisions by it are in every row kernel)
constexpr int NOISE_MIN_FAST = 512;            // shortest noise table of the fast row kernels (a block of 256 samples wraps at most once)

// Everything a kernel needs, passed by value (fits the kernarg segment).
struct WfsDev {
    // scalars (wfs_config)
    i32 dt, samples_before, samples_after, store_before, store_after, tlen, tw, baseline, n_rows;
    i32 n_tpc, n_top, he_first, he_factor, last_bottom, detector_nt, enable_noise, s1_simple, s2_time_model, enable_pmt_ap;
    i32 n_spe, n_lum, noise_len, noise_channels, n_ap, he_rows /* HE rows materialised */, row_slots /* per group */;
    i32 noise_stride;                          // samples between the noise rows of two channels: noise_len + NOISE_PAD, the pad repeats the row's start
    double c2a, tts_mean, tts_sigma, p_dpe, s1_decay_time, s1_decay_spread, sf_gas, t1_gas, t3_gas, s2_time_spread;
    double trap_time, gain_spread, pmt_ap_modifier, pmt_ap_t_modifier, rext;
    u32 k0, k1;
    unsigned long long *stamps;                // -DWFS_STAMPS builds only: per-phase cycle sums of the instrumented kernels (nullptr otherwise)
    u64 thr_dpe;                               // Bernoulli threshold on a 32-bit word: floor(p * 2^32)
    double dpe_inv;                            // 2000 / thr_dpe
    double current_max[16];                    // pulse.py:32, per ns remainder (sample_duration <= 16 ns)
    AliasTab tab_tts, tab_s1, tab_s2;          // alias tables of integer delays: transit time alone, all terms of an S1 / S2 photon
    // tables
    const double *templates, *spe, *gains, *thr_truth, *lum_x, *lum_t;
    const i64 *thr_zle;
    const int16_t *noise;
    const double *noise_f;       // a float noise array (wfs_set_noise_float): the sum is stored truncated (rawdata.py:436 into an int64 row)
};
