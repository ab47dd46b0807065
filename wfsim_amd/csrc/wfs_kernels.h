// wfs_kernels.h -- HIP kernels of the MI355X WFSim hot path (gfx950).  Included once by wfs_engine.hip.
//
// Data layout in HBM (one batch):
//   pulse set   = one Pulse.__call__ of the reference (one S1 or S2 instruction, or its PMT afterpulses)
//   tile        = (pulse set, channel): the photons of one pulse in one PMT; tile id = set * n_tpc + channel
//   photons     = bucketed by tile: 8-byte records {t i32 (ns relative to the set's t0), code u32 (SPE table indices
//                 g1 | g2 << 16, g2 = 0: no double-PE)}; ph_gain f64 only for pre-assigned gains
//   group       = digitise window (one digitize_pulse_cache call): clusters merged by the rule of rawdata.py:96-98
//   row         = (group, channel) active range [row_lo - tw, row_hi + tw]; int32 accumulators in one arena
//   intervals   = ZLE output per row, reserved slots; records = 244-byte strax raw_records
#pragma once
#include "wfs_device.h"
#include <cstddef>

#define I64_MAX 0x7fffffffffffffffLL
#define I64_MIN (-I64_MAX - 1)

// ------------------------------------------------------------------------------------------------ fills
__global__ void k_fill_i64(i64 *p, i64 n, i64 v) { i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_fill_i32(i32 *p, i64 n, i32 v) { i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_fill_minmax(i64 *p, i64 n) { i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = (i & 1) ? I64_MIN : I64_MAX; }

// totals for wfs_get_counts, reduced on the device (copying itv_n back cost 300 MB per 10^5-cluster batch):
// scal[20] = sum of itv_n (ZLE intervals), scal[21] = sum of the per-set n_pe (truth[s][1], integral doubles)
__global__ void k_counts(const i32 *itv_n, i64 n_itv, const double *truth, i64 n_sets, i64 *scal)
{
    i64 a = 0, b = 0;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n_itv; i += (i64)gridDim.x * blockDim.x) a += itv_n[i];
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n_sets; i += (i64)gridDim.x * blockDim.x) b += (i64)(truth[i * 16 + 1] + 0.5);
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (a) atomicAdd((unsigned long long *)&scal[20], (unsigned long long)a); if (b) atomicAdd((unsigned long long *)&scal[21], (unsigned long long)b); }
}

// ------------------------------------------------------------------------------------------------ scan
// exclusive scan i32[n] -> i64[n+1] in three launches (reduce / spine / down-sweep); 4096 items per block
#define SCAN_TPB 256
#define SCAN_IPT 16
#define SCAN_TILE (SCAN_TPB * SCAN_IPT)

__device__ __forceinline__ i64 block_reduce_sum(i64 v, i64 *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    i64 r = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) r += sh[i];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(SCAN_TPB) void k_scan_reduce(const i32 *__restrict__ in, i64 n, i64 *__restrict__ block_sums)
{
    __shared__ i64 sh[SCAN_TPB / 64];
    i64 base = (i64)blockIdx.x * SCAN_TILE;
    i64 s = 0;
    // (branch-free loads: behind `if (i < n)` every load is waited for before the next is issued)
    i32 x[SCAN_IPT];
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { const i64 i = base + (i64)k * SCAN_TPB + threadIdx.x; x[k] = in[i < n ? i : 0]; }
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { const i64 i = base + (i64)k * SCAN_TPB + threadIdx.x; s += i < n ? x[k] : 0; }
    s = block_reduce_sum(s, sh);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_scan_spine(i64 *__restrict__ block_sums, i64 nb, i64 *__restrict__ total)
{
    __shared__ i64 sh[1024];
    __shared__ i64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (i64 base = 0; base < nb; base += 1024) {
        i64 i = base + threadIdx.x;
        i64 v = i < nb ? block_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            i64 t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(SCAN_TPB) void k_scan_down(const i32 *__restrict__ in, i64 n, const i64 *__restrict__ block_sums,
                                                        i64 *__restrict__ out, i64 offset)
{
    __shared__ i64 sh[SCAN_TPB];
    i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_IPT;
    i64 v[SCAN_IPT], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { const i64 i = base + k; v[k] = in[i < n ? i : 0]; }
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { if (base + k >= n) v[k] = 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < SCAN_TPB; o <<= 1) {
        i64 t = threadIdx.x >= (unsigned)o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    i64 run = offset + block_sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < SCAN_IPT; k++) { i64 i = base + k; if (i < n) out[i] = run; run += v[k]; if (i == n - 1) out[n] = run; }
}

// ------------------------------------------------------------------------------------------------ geometry
struct GeomArgs {
    i64 n_sets, n_tiles, n_clusters;
    i64 n_gslots;        // capacity of the per-group arrays: n_clusters + 1 (a carried-in window can open group 1 at cluster 0)
    const i32 *tile_count, *tile_tmin, *tile_tmax;
    const i32 *set_cluster; const i64 *set_t0;
    const i64 *cl_tmin; const u32 *cl_gid;
    i64 *cl_end; i32 *cl_group;
    i64 *grp_lo, *grp_hi, *grp_left, *grp_right, *grp_ixrand; u32 *grp_gid;
    i64 *row_lo, *row_hi;
    i32 *acc_len, *itv_cap; i32 *active_rows;
    i64 *scal;           // [0] n_groups [1] error flag [2] n_active_rows [3] n sparse tiles [4] their max start bins
                         // [5] their max photons [11] n dense tiles [12] their max start bins
    i32 *active_tiles;   // tiny tiles (thread per tile); the other two lists are copied behind them by the host side
    i32 *sparse_tiles;   // sparse-class tiles (few photons per start bin: sorted-list kernel)
    i32 *dense_tiles;    // everything else (dense H-table kernel, windows over time)
    i32 *wave_tiles;     // medium tiles (at most 64 photons, any width: wave per tile); scal[17] = their number
    const i32 *tile_done;       // [n_done] 1: the tile's pulse was made by k_s2_tile<FULL> (wfs_tilegen.h): on no work list (or nullptr)
    i64 n_done;                 // primary tiles (the tiles of the afterpulse sets, which follow them, are never done)
    i32 *row_cnt, *row_tile;    // [groups * n_tpc] tiles in the row; one of them
    const i32 *ins_bcap; const i64 *ins_boff;      // tile sample buffers (wfs_tilegen.h)
    // resident rows (k_row_pulse): rows no longer than res_max_len whose tiles all fit a wave are made, finished and zero-suppressed by
    // ONE wave in LDS; their tiles are on no work list but in a per-row list (res_toff: row -> first entry of res_desc)
    i32 res_on, res_max_len;
    i32 *row_bad;               // [groups * n_tpc] 1: the row holds a tile that needs one of the tile kernels
    i32 *fin_len, *res_cnt;     // [groups * n_tpc] finished int16 samples reserved for a resident row (a multiple of 4) / its tiles
    const i64 *res_toff;        // exclusive scan of res_cnt
    struct TileDesc *res_desc;  // [resident tiles] descriptors in row order (k_tile_assign)
    i64 rows_cap;               // capacity of active_rows: resident rows of at most RES_SHORT_LEN samples are appended from its end backwards,
    i32 *res_long;              // the longer ones to this list (two launches of k_row_pulse: its LDS goes with the longest row)
    i32 force_dense;     // debug: send every tile to the dense kernel
    i32 init_has; i64 init_runmax;      // last_pulse_end_time carried in from earlier batches
    const i64 *noise_override; i64 n_noise_override;
};

#define SPARSE_MAX_PHOTONS 32      // tiles with a handful of photons (S1-like) go to the sorted-list kernel
#define SPARSE_MAX_BINS 64
#define TINY_MAX_PHOTONS 4         // tiles of a few photons in a few start bins (the bulk of an S1): one THREAD per tile
#define TINY_MAX_BINS 32
#define TINY_LANES 16              // lanes that share the samples of one tiny tile (k_pulse_tiny)
#define WAVE_MAX_PHOTONS 64        // medium tiles: photons fit the lanes of one wave (any width): k_pulse_wave
#define WAVE_MAX_BINS 16384
#define RES_MAX_TILE_VISITS 256      // a resident row longer than the LDS of its wave: tiles x segments it may cost
#define RES_SHORT_LEN 768          // resident rows up to this length: 4 waves x 3 KB of LDS per workgroup, eight workgroups per CU

__device__ __forceinline__ void tile_bounds(const WfsDev &d, i64 t0, i32 tmin, i32 tmax, i64 &left, i64 &right, i64 &bin0, i64 &nb)
{
    // pulse.py:118-127
    bin0 = floordiv(t0 + tmin, (i64)d.dt);
    i64 bin1 = floordiv(t0 + tmax, (i64)d.dt);
    nb = bin1 - bin0 + 1;
    left = bin0 - d.store_before - d.samples_before;
    right = bin1 + d.store_after + d.samples_after;
}

// work lists and maxima: aggregated per workgroup in LDS, then one global atomic per workgroup and counter
// (76k wave-level atomics on one cache line cost ~3 ms; the counters all live in the same line of scal[])
__device__ __forceinline__ void tile_list_append(const GeomArgs &a, bool listed, int cls, i64 nb, i32 cnt, i64 tile)
{
    __shared__ i32 s_n[4]; __shared__ i64 s_b[4], s_mx[4];
    if (threadIdx.x == 0) { s_n[0] = 0; s_n[1] = 0; s_n[2] = 0; s_n[3] = 0; s_mx[0] = 0; s_mx[1] = 0; s_mx[2] = 0; s_mx[3] = 0; }
    __syncthreads();
    i32 rk = 0;
    if (listed) {
        rk = atomicAdd(&s_n[cls], 1);
        if (cls == 1) { atomicMax(&s_mx[0], nb); atomicMax(&s_mx[1], (i64)cnt); } else if (cls == 2) { atomicMax(&s_mx[2], nb); atomicMax(&s_mx[3], (i64)cnt); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        s_b[0] = s_n[0] ? (i64)atomicAdd((u64 *)&a.scal[16], (u64)s_n[0]) : 0;
        s_b[1] = s_n[1] ? (i64)atomicAdd((u64 *)&a.scal[3], (u64)s_n[1]) : 0;
        s_b[2] = s_n[2] ? (i64)atomicAdd((u64 *)&a.scal[11], (u64)s_n[2]) : 0;
        s_b[3] = s_n[3] ? (i64)atomicAdd((u64 *)&a.scal[17], (u64)s_n[3]) : 0;
        if (s_mx[0]) atomicMax(&a.scal[4], s_mx[0]);
        if (s_mx[1]) atomicMax(&a.scal[5], s_mx[1]);
        if (s_mx[2]) atomicMax(&a.scal[12], s_mx[2]);
        if (s_mx[3]) atomicMax(&a.scal[15], s_mx[3]);
    }
    __syncthreads();
    if (listed) (cls == 0 ? a.active_tiles : (cls == 1 ? a.sparse_tiles : (cls == 2 ? a.dense_tiles : a.wave_tiles)))[s_b[cls] + rk] = (i32)tile;
}

__device__ __forceinline__ int tile_class(const GeomArgs &a, i32 cnt, i64 nb)      // 0 tiny, 1 sparse, 2 dense, 3 wave (medium)
{
    if (a.force_dense) return 2;
    return (cnt <= TINY_MAX_PHOTONS && nb <= TINY_MAX_BINS) ? 0 : ((cnt <= SPARSE_MAX_PHOTONS && nb <= SPARSE_MAX_BINS) ? 1
           : ((cnt <= WAVE_MAX_PHOTONS && nb <= WAVE_MAX_BINS) ? 3 : 2));
}

// per tile: end time of its pulse -> cluster (rawdata.py:188-190); work lists of non-empty tiles.
// List appends and the two maxima are aggregated per wave (one atomic per wave instead of one per tile).
__global__ void k_tile_geom(WfsDev d, GeomArgs a)
{
    const i64 tile = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = tile < a.n_tiles && a.tile_count[tile] > 0;
    int cls = 2; i64 nb = 0; i32 cnt = 0; i32 cl = -1; i64 end = I64_MIN;      // class 0 tiny, 1 sparse, 2 dense, 3 wave (medium)
    if (live) {
        const i64 set = tile / d.n_tpc;
        i64 left, right, bin0;
        tile_bounds(d, a.set_t0[set], a.tile_tmin[tile], a.tile_tmax[tile], left, right, bin0, nb);
        cl = a.set_cluster[set]; end = right * d.dt;
        cnt = a.tile_count[tile];
        cls = tile_class(a, cnt, nb);
        if (a.tile_done && tile < a.n_done && a.tile_done[tile]) cls = -1;      // its pulse exists already (k_s2_tile): no work list
    }
    {   // cluster end time: the live lanes of a wave usually belong to one cluster -> one atomic for the wave
        const u64 ml = ballot64(live);
        if (ml) {
            const i32 cl0 = __shfl(cl, __ffsll((long long)ml) - 1, 64);
            if (all64(!live || cl == cl0)) {
                i64 e = end;
                for (int o = 32; o > 0; o >>= 1) { const i64 x = __shfl_down(e, o, 64); e = x > e ? x : e; }
                if (lane == 0) atomicMax(&a.cl_end[cl0], e);
            } else if (live) atomicMax(&a.cl_end[cl], end);
        }
    }
    if (!a.res_on) tile_list_append(a, live && cls >= 0, cls, nb, cnt, tile);      // (resident rows on: k_tile_assign makes the lists, once the rows are known)
}

// Digitise groups.  The cache is digitised before cluster k when min(instruction key of k) - last_pulse_end_time > rext
// and a pulse exists (rawdata.py:96-98); last_pulse_end_time is a running maximum over everything simulated so far
// (rawdata.py:188-190).  One workgroup: exclusive running maximum of the cluster end times, then a prefix sum of the flags.
#define GROUPS_TPB 1024
__global__ __launch_bounds__(GROUPS_TPB) void k_groups(WfsDev d, GeomArgs a)
{
    __shared__ i64 smax[GROUPS_TPB]; __shared__ i32 shas[GROUPS_TPB]; __shared__ i32 ssum[GROUPS_TPB];
    const int tid = threadIdx.x;
    const i64 C = a.n_clusters, per = (C + GROUPS_TPB - 1) / GROUPS_TPB;
    const i64 k0 = tid * per, k1 = (k0 + per < C) ? k0 + per : C;
    // 1) per-thread maximum of its clusters' end times
    i64 m = I64_MIN; i32 hs = 0;
    for (i64 k = k0; k < k1; k++) { const i64 e = a.cl_end[k]; if (e != I64_MIN) { m = hs ? (e > m ? e : m) : e; hs = 1; } }
    smax[tid] = m; shas[tid] = hs;
    __syncthreads();
    // 2) exclusive running maximum over threads (plus the carry from earlier batches)
    for (int o = 1; o < GROUPS_TPB; o <<= 1) {
        i64 pm = I64_MIN; i32 ph = 0;
        if (tid >= o) { pm = smax[tid - o]; ph = shas[tid - o]; }
        __syncthreads();
        if (ph) { smax[tid] = shas[tid] ? (pm > smax[tid] ? pm : smax[tid]) : pm; shas[tid] = 1; }
        __syncthreads();
    }
    i64 run = a.init_runmax; i32 has = a.init_has;
    if (tid > 0 && shas[tid - 1]) { const i64 pm = smax[tid - 1]; run = has ? (pm > run ? pm : run) : pm; has = 1; }
    // 3) flags of my clusters, sequentially inside the thread
    i32 nflag = 0;
    {
        i64 r = run; i32 h = has;
        for (i64 k = k0; k < k1; k++) {
            if (h && (double)(a.cl_tmin[k] - r) > d.rext) nflag++;
            const i64 e = a.cl_end[k];
            if (e != I64_MIN) { r = h ? (e > r ? e : r) : e; h = 1; }
        }
    }
    ssum[tid] = nflag;
    __syncthreads();
    for (int o = 1; o < GROUPS_TPB; o <<= 1) {
        i32 t = tid >= o ? ssum[tid - o] : 0;
        __syncthreads();
        ssum[tid] += t;
        __syncthreads();
    }
    // 4) group of every cluster; the first cluster of a group that made a pulse names its noise stream (grp_gid holds its
    //    index here; clusters without pulses must not matter, or the stream would depend on how a run is cut into batches)
    i32 g = ssum[tid] - nflag;
    {
        i64 r = run; i32 h = has;
        for (i64 k = k0; k < k1; k++) {
            const bool flag = h && (double)(a.cl_tmin[k] - r) > d.rext;
            if (flag) g++;
            a.cl_group[k] = g;
            const i64 e = a.cl_end[k];
            if (e != I64_MIN) { atomicMin(&a.grp_gid[g], (u32)k); r = h ? (e > r ? e : r) : e; h = 1; }     // first cluster of the group with a pulse
        }
    }
    if (tid == GROUPS_TPB - 1) a.scal[0] = ssum[tid] + 1;
}

__global__ void k_tile_rows(WfsDev d, GeomArgs a)
{
    i64 tile = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = tile < a.n_tiles && a.tile_count[tile] > 0;
    i64 left = I64_MAX, right = I64_MIN, g = -1;
    if (live) {
        i64 set = tile / d.n_tpc; i32 ch = (i32)(tile - set * d.n_tpc);
        i64 bin0, nb;
        tile_bounds(d, a.set_t0[set], a.tile_tmin[tile], a.tile_tmax[tile], left, right, bin0, nb);
        g = a.cl_group[a.set_cluster[set]];
        atomicMin(&a.row_lo[g * d.n_tpc + ch], left); atomicMax(&a.row_hi[g * d.n_tpc + ch], right);
        if (a.row_cnt) { atomicAdd(&a.row_cnt[g * d.n_tpc + ch], 1); a.row_tile[g * d.n_tpc + ch] = (i32)tile; }      // (row_tile: only read when the row has ONE tile)
        if (a.res_on) {         // a tile of more than a wave's photons, or one whose pulse exists already, keeps its row off the resident path
            const bool done = a.tile_done && tile < a.n_done && a.tile_done[tile];
            if (done || a.tile_count[tile] > WAVE_MAX_PHOTONS || nb > WAVE_MAX_BINS) a.row_bad[g * d.n_tpc + ch] = 1;
        }
    }
    // the group's range: the 494 tiles of a pulse set all aim at one address -- reduce inside the wave first when its
    // live lanes share the group (the usual case), one atomic pair per wave instead of 64
    const u64 lm = ballot64(live);
    if (lm == 0) return;
    const i64 g0 = __shfl(g, __ffsll((long long)lm) - 1, 64);
    if (all64(!live || g == g0)) {
        for (int o = 32; o > 0; o >>= 1) {
            const i64 l2 = __shfl_xor(left, o, 64), r2 = __shfl_xor(right, o, 64);
            left = l2 < left ? l2 : left; right = r2 > right ? r2 : right;
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&a.grp_lo[g0], left); atomicMax(&a.grp_hi[g0], right); }
    } else if (live) { atomicMin(&a.grp_lo[g], left); atomicMax(&a.grp_hi[g], right); }
}

// per group: window (rawdata.py:215-224) and the noise offset (rawdata.py:407-417)
__global__ void k_group_final(WfsDev d, GeomArgs a)
{
    i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.scal[0]) return;
    if (a.grp_lo[g] == I64_MAX) { a.grp_left[g] = 0; a.grp_right[g] = -1; a.grp_ixrand[g] = -1; return; }
    i64 left = a.grp_lo[g] - d.tw, right = a.grp_hi[g] + d.tw;
    if (!(right - left < 1000000)) atomicMax(&a.scal[1], (i64)1);           // "Pulse cache too long", rawdata.py:219
    if (floormod(left, 2) != 0) left -= 1;
    a.grp_left[g] = left; a.grp_right[g] = right;
    i64 ix = -1;
    if (d.enable_noise) {
        i64 nl = a.grp_lo[g] - left - d.tw, nr = a.grp_hi[g] - left + d.tw;
        i64 N = d.noise_len, high = (N - nr + nl - 1 < 0) ? N - 1 : N - nr + nl - 1;
        if (high <= 0) ix = 0;
        else { u32x4 w = philox4x32_10(0, a.cl_gid[a.grp_gid[g]], 0, SITE_NOISE, d.k0, d.k1); ix = (i64)(u53(w.x, w.y) * (double)high); }
        // (reduced modulo the noise length: add_noise wraps the index, rawdata.py:433-434, and the single-subtract wrap of
        // load_sample relies on ix < noise_len)
        if (a.noise_override && g < a.n_noise_override && a.noise_override[g] >= 0) ix = N > 0 ? a.noise_override[g] % N : 0;
    }
    a.grp_ixrand[g] = ix;
}

__device__ __forceinline__ bool row_is_direct(const i32 *tile_done, i64 n_done, const i32 *row_cnt, const i32 *row_tile, i64 ridx)
{
    if (!tile_done || row_cnt[ridx] != 1) return false;
    const i64 tile = row_tile[ridx];
    return tile < n_done && tile_done[tile] != 0;
}

// per (group, row slot): accumulator length and reserved ZLE interval slots; list of rows with data
// (appended with one global atomic per workgroup)
__global__ void k_row_len(WfsDev d, GeomArgs a)
{
    __shared__ i32 s_n, s_nres, s_nlong, s_direct, s_shared, s_maxres, s_reslen; __shared__ i64 s_base, s_rbase, s_lbase;
    if (threadIdx.x == 0) { s_n = 0; s_nres = 0; s_nlong = 0; s_direct = 0; s_shared = 0; s_maxres = 0; s_reslen = 0; }
    __syncthreads();
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n = a.scal[0] * d.row_slots;
    i32 cap = 0, rk = -1; bool resident = false, longrow = false;
    if (idx < n) {
        const i64 g = idx / d.row_slots; const i32 slot = (i32)(idx - g * d.row_slots);
        const i32 ch = slot < d.n_tpc ? slot : slot - d.n_tpc;         // HE slot -> its top channel
        const i64 lo = a.row_lo[g * d.n_tpc + ch];
        if (lo != I64_MAX) {
            const i64 len = a.row_hi[g * d.n_tpc + ch] - lo + 1 + 2 * (i64)d.tw;
            i64 hold = 2 * (i64)d.tw + 1; if (hold < 1) hold = 1;
            cap = (i32)((len + hold) / (hold + 1));
            // a row made by ONE tile whose samples exist already (k_s2_tile) is read from the tile's buffer: no accumulators
            const bool direct = row_is_direct(a.tile_done, a.n_done, a.row_cnt, a.row_tile, g * d.n_tpc + ch);
            // a short row of small tiles is made in LDS by one wave (k_row_pulse): no accumulators either, 16-bit finished samples instead
            // (a row longer than res_max_len samples is made in segments of that length, its tiles looked at once per segment)
            resident = a.res_on && !direct && !a.row_bad[g * d.n_tpc + ch]
                       && (len <= a.res_max_len || (i64)a.row_cnt[g * d.n_tpc + ch] * ((len + a.res_max_len - 1) / a.res_max_len) <= RES_MAX_TILE_VISITS);
            if (slot < d.n_tpc) {
                a.acc_len[g * d.n_tpc + ch] = (direct || resident) ? 0 : (i32)len; if (direct) atomicAdd(&s_direct, (i32)len);
                if (a.res_on) { a.fin_len[g * d.n_tpc + ch] = resident ? (i32)((len + 3) & ~(i64)3) : 0; a.res_cnt[g * d.n_tpc + ch] = resident ? a.row_cnt[g * d.n_tpc + ch] : 0; }
                if (resident) { atomicMax(&s_maxres, (i32)len); atomicAdd(&s_reslen, (i32)len); longrow = len > RES_SHORT_LEN; }
            }
            if (slot < d.n_tpc && !resident && a.row_cnt && a.row_cnt[g * d.n_tpc + ch] > 1) atomicAdd(&s_shared, 1);
            rk = atomicAdd(resident ? (longrow ? &s_nlong : &s_nres) : &s_n, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = (i64)atomicAdd((u64 *)&a.scal[2], (u64)s_n);
    if (threadIdx.x == 0 && s_nres) s_rbase = (i64)atomicAdd((u64 *)&a.scal[32], (u64)s_nres);
    if (threadIdx.x == 0 && s_nlong) s_lbase = (i64)atomicAdd((u64 *)&a.scal[37], (u64)s_nlong);
    if (threadIdx.x == 0 && s_reslen) { atomicMax(&a.scal[33], (i64)s_maxres); atomicAdd((u64 *)&a.scal[36], (u64)s_reslen); }
    if (threadIdx.x == 0 && s_direct) atomicAdd((u64 *)&a.scal[26], (u64)s_direct);      // samples of the rows read in place
    if (threadIdx.x == 0 && s_shared) atomicAdd((u64 *)&a.scal[27], (u64)s_shared);      // rows made by several tiles (k_tile_add is needed)
    __syncthreads();
    if (rk >= 0) {
        if (!resident) a.active_rows[s_base + rk] = (i32)idx;
        else if (!longrow) a.active_rows[a.rows_cap - 1 - (s_rbase + rk)] = (i32)idx;
        else a.res_long[s_lbase + rk] = (i32)idx;
    }
    if (idx < a.n_gslots * d.row_slots) a.itv_cap[idx] = cap;
}

// ------------------------------------------------------------------------------------------------ pulse kernel
// Pulse.__call__ per-channel body + add_current (pulse.py:82-144, 276-318) + the per-pulse rounding and
// accumulation of digitize_pulse_cache (rawdata.py:231-239), one workgroup per tile.
//
// add_current visits photons in ascending time, merges photons of equal ns and adds templates[t % dt] * gain_total
// at sample t // dt - pulse_left.  Here the merged gains are built first in LDS, H[r][j] = sum of gains of photons
// with t % dt = r in start bin j (one LDS float atomic per photon), then every sample gathers its 22 x 10 possible
// contributions in ascending (j, r) = ascending time order with a separate multiply and add.  Adding a zero
// contribution does not change a float sum, so the result equals the reference's current bit for bit (except for
// the order in which three or more photons of the SAME ns are merged, which numpy's unstable argsort leaves open).
// Everything a pulse workgroup needs to know about its tile, resolved once per tile by k_tile_desc (five levels of
// dependent lookups otherwise: work list -> tile -> pulse set -> cluster -> window row).  64 bytes, read with scalar loads.
struct TileDesc {
    i64 off;              // first photon
    i64 dst;              // index in raw[] of the tile's sample 0
    i64 rel0;             // ns of the tile's first start bin relative to the set's t0
    double G, thr;        // PMT gain, truth threshold of the channel
    i32 n, nb, L, tile;   // photons, start bins, samples, tile id
    i32 ch, mode;
};

struct PulseArgs {
    const i32 *active_tiles; i64 n_active;
    const TileDesc *desc;   // per work-list entry (dense kernel)
    const i32 *tile_count, *tile_tmin, *tile_tmax; const i64 *tile_off;
    const i32 *set_cluster; const i64 *set_t0; const i32 *set_mode;       // mode 0: SPE codes, 1: explicit gains
    const PhotonRec *ph; const double *ph_gain;
    const i32 *cl_group; const i64 *row_lo; const i64 *acc_off;
    i32 *raw;
    double *tile_truth;   // [n_tiles][8] per-tile partial sums: n, n_dpe, n_trig, n_trig_dpe, sum g, sum g trig, sum t, sum t^2 -> k_truth_reduce
    double *currents; const i64 *cur_off;     // debug: f64 tile currents
    i32 W;                // bins capacity of the sparse kernel's LDS tables
    i32 NP;               // photon capacity of the sparse kernel's LDS list
    i32 n_win;            // dense kernel: workgroups per tile (each takes every n_win-th chunk of TPB samples)
    i32 spe_lds;          // dense resident kernel: the H table has room for the channel's SPE row (2001 doubles), staged there before the gains are looked up
    i32 sparse_max;       // dense kernel: a wave whose samples see at most this many occupied cells walks them instead of gathering all 220 per sample (tap_block)
};

#define DENSE_PPT 8        // photons per thread per batch held in registers

struct TemplateArg { double t[22 * WFS_DT]; };    // t[k * dt + r] = templates[r][k]; kernarg segment -> scalar loads, the taps live in SGPRs

struct DescArgs { const i32 *list; i64 n; const i32 *tile_count, *tile_tmin, *tile_tmax; const i64 *tile_off; const i32 *set_cluster; const i64 *set_t0;
                  const i32 *set_mode; const i32 *cl_group; const i64 *row_lo; const i64 *acc_off; TileDesc *desc; };
__global__ void k_tile_desc(WfsDev d, DescArgs a)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const i64 tile = a.list[i];
    const i64 set = tile / d.n_tpc; const i32 ch = (i32)(tile - set * d.n_tpc);
    const i64 t0 = a.set_t0[set];
    i64 left, right, bin0, nb;
    tile_bounds(d, t0, a.tile_tmin[tile], a.tile_tmax[tile], left, right, bin0, nb);
    const i64 ridx = (i64)a.cl_group[a.set_cluster[set]] * d.n_tpc + ch;
    TileDesc t;
    t.off = a.tile_off[tile]; t.dst = a.acc_off[ridx] + (left - (a.row_lo[ridx] - d.tw)); t.rel0 = bin0 * d.dt - t0;
    t.G = d.gains[ch]; t.thr = d.thr_truth[ch];
    t.n = a.tile_count[tile]; t.nb = (i32)nb; t.L = (i32)(right - left + 1); t.tile = (i32)tile;
    t.ch = ch; t.mode = a.set_mode[set];
    a.desc[i] = t;
}

// The gather of one sample (pulse.py:303-318 in the H-table form): thread `tid` owns the sample whose start bins are the H rows
// tid (tap k = 21) .. tid + 21 (tap k = 0); it adds its tlen x dt possible contributions in ascending time, separate multiply and
// add.  The tap loop is outermost so that only the dt taps of one k are live in SGPRs.
template <bool FMA>
__device__ __forceinline__ double tap_gather(const double *H, const TemplateArg &tp, int tid)
{
    constexpr int dt = WFS_DT, tlen = 22;
    const double *Hs = H + (tid + (tlen - 1)) * dt;
    double c = 0.0;
#pragma unroll 1
    for (int k = tlen - 1; k >= 0; k--) {
        const double *Tk = tp.t + k * dt;          // the dt taps of one k are contiguous: two wide scalar loads
        const double T0 = Tk[0], T1 = Tk[1], T2 = Tk[2], T3 = Tk[3], T4 = Tk[4], T5 = Tk[5], T6 = Tk[6], T7 = Tk[7], T8 = Tk[8], T9 = Tk[9];
        const double2 *hp = (const double2 *)(Hs - k * dt);
        const double2 h0 = hp[0], h1 = hp[1], h2 = hp[2], h3 = hp[3], h4 = hp[4];
        c = mac<FMA>(T0, h0.x, c); c = mac<FMA>(T1, h0.y, c);
        c = mac<FMA>(T2, h1.x, c); c = mac<FMA>(T3, h1.y, c);
        c = mac<FMA>(T4, h2.x, c); c = mac<FMA>(T5, h2.y, c);
        c = mac<FMA>(T6, h3.x, c); c = mac<FMA>(T7, h3.y, c);
        c = mac<FMA>(T8, h4.x, c); c = mac<FMA>(T9, h4.y, c);
    }
    return c;
}

// The same sample from the OCCUPIED cells alone, for waves whose 64 samples see few photons (the tails of an S2 tile, where most of
// the 220 cells a sample gathers are empty: the dense gather spends 8 f64 issue cycles on every one of them).  The wave scans the
// 850 cells its samples can reach (H rows 64 w .. 64 w + 84) with ballots -- two rounds in the middle first: a wave in the
// bulk of the tile is recognised there and leaves for the dense gather.  With at most `sparse_max` (<= TAP_LIST_LEN) occupied cells
// it lists them in ascending time (LDS, 2 bytes each), lane i takes entry i (cell, merged gain) into registers, and the entries
// are then broadcast one by one with readlane: every lane looks up ITS tap W2[(10 (lane + 21) + 9) - cell] (the zero tap W2[220]
// outside the reach of its sample) and adds the product with a separate multiply and add -- the additions of the dense gather in
// the same order minus the additions of +0.0, hence the same bits (pulse.py:303-318).  W2[k * dt + (dt - 1 - r)] = templates[r][k].
// A cell counts as occupied when the high word of its merged gain is non-zero (any gain of at least 2^-1022).
#define TAP_W2_LEN (22 * WFS_DT + 2)
#define TAP_LIST_LEN 64                // entries of a wave's list (unsigned short): one per lane
#define TAP_LDS_BYTES(tpb) (TAP_W2_LEN * 8 + ((tpb) / 64) * TAP_LIST_LEN * 2)
__device__ __forceinline__ void tap_w2_fill(double *W2, const double *templates, int tid, int tpb)
{
    constexpr int dt = WFS_DT, tlen = 22;
    for (int i = tid; i < TAP_W2_LEN; i += tpb) W2[i] = i < tlen * dt ? templates[(dt - 1 - i % dt) * tlen + i / dt] : 0.0;
}
// W2 is followed by the waves' lists: [TAP_W2_LEN doubles][tpb / 64][TAP_LIST_LEN] unsigned short
// n_cells: the cells of H that hold data (the caller may have cleared only the rows its live samples can see)
template <bool FMA>
__device__ __forceinline__ double tap_block(const double *H, double *W2, const TemplateArg &tp, int tid, int sparse_max, int n_cells)
{
    constexpr int dt = WFS_DT, tlen = 22, REACH = (64 + tlen - 1) * dt, NR = (REACH + 63) / 64;
    const int lane = tid & 63;
    if (sparse_max < 0) return tap_gather<FMA>(H, tp, tid);
    const double *Hw = H + (tid & ~63) * dt;            // first cell of the wave's reach
    const u32 *Hhi = (const u32 *)Hw + 1;               // high words
    unsigned short *list = (unsigned short *)(W2 + TAP_W2_LEN) + (tid >> 6) * TAP_LIST_LEN;
    const u64 below = (1ull << lane) - 1ull;
    const int c_end = min(REACH, n_cells - (tid & ~63) * dt);      // cells of the reach that hold data
    {   // rounds 6 and 7 (the middle of the reach) first
        const u32 a = 6 * 64 + lane < c_end ? Hhi[2 * (6 * 64 + lane)] : 0u, b = 7 * 64 + lane < c_end ? Hhi[2 * (7 * 64 + lane)] : 0u;
        if (__popcll(ballot64(a != 0u)) + __popcll(ballot64(b != 0u)) > (sparse_max >> 2)) return tap_gather<FMA>(H, tp, tid);      // wave-uniform
    }
    int m = 0;
#pragma unroll 1
    for (int r0 = 0; r0 < NR; r0 += 4) {                 // (four loads in flight at a time)
        u32 hw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int c = (r0 + u) * 64 + lane; hw[u] = c < c_end ? Hhi[2 * c] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u64 mk = ballot64(hw[u] != 0u);
            if (mk) {                                    // wave-uniform
                const int pos = m + __popcll(mk & below);
                if (hw[u] != 0u && pos < TAP_LIST_LEN) list[pos] = (unsigned short)((r0 + u) * 64 + lane);
                m += __popcll(mk);
            }
        }
    }
    if (m > sparse_max) return tap_gather<FMA>(H, tp, tid);   // wave-uniform
    if (m == 0) return 0.0;
    // lane i <- entry i (behind the last entry: a cell no sample reaches)
    const int ci = lane < m ? (int)list[lane] : 0x7fff;
    const double gi = Hw[lane < m ? ci : 0];
    const int g_lo = (int)(u32)__double_as_longlong(gi), g_hi = (int)(u32)((u64)__double_as_longlong(gi) >> 32);
    const u32 A = (u32)((lane + tlen - 1) * dt + (dt - 1));
    double acc = 0.0;
    for (int i0 = 0; i0 < m; i0 += 4) {                  // four entries in flight (one LDS round trip per four)
        double w[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = (i0 + u) & 63;                 // (entries behind the last one carry the cell 0x7fff: the zero tap for every lane)
            const int c = __builtin_amdgcn_readlane(ci, i);
            g[u] = __longlong_as_double((long long)(((u64)(u32)__builtin_amdgcn_readlane(g_hi, i) << 32) | (u32)__builtin_amdgcn_readlane(g_lo, i)));
            u32 idx = A - (u32)c;
            idx = idx < (u32)(tlen * dt) ? idx : (u32)(tlen * dt);
            w[u] = W2[idx];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc = mac<FMA>(w[u], g[u], acc);
    }
    return acc;
}

template <int TPB, bool RESIDENT, bool FMA>
__global__ __launch_bounds__(TPB) void k_pulse(WfsDev d, PulseArgs a, TemplateArg tp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int dt = WFS_DT, tlen = 22;
    // The live samples of a tile (sample s' = s - lead sees the start bins s' - 21 .. s') are worked off in chunks of TPB
    // consecutive samples, one per thread; a chunk needs the TPB + tlen - 1 start bins from 21 before its first sample on.
    constexpr int HROWS = TPB + tlen - 1;
    double *H = (double *)smem;                           // [HROWS][dt]: merged gain per (start bin, ns remainder)
    double *red = H + (size_t)HROWS * dt;                 // [TPB / 64][8]
    u32 *wsum = (u32 *)(red + 8 * (TPB / 64));            // [16]
    double *W2 = (double *)(wsum + 16);                   // [TAP_W2_LEN] taps by time difference + the waves' records (tap_block)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // pulse.py:32 current_max, indexed per photon: from LDS (indexing the kernarg copy per lane becomes a global load)
    __shared__ double s_cmax[WFS_DT];
    if (tid < WFS_DT) s_cmax[tid] = d.current_max[tid];
    tap_w2_fill(W2, d.templates, tid, TPB);

    // grid = tiles x a.n_win: workgroup `win` of a tile takes the chunks win, win + n_win, ... (n_win = 1: the resident form)
    const i64 tidx = blockIdx.x / a.n_win;
    const int win = (int)(blockIdx.x - tidx * a.n_win);
    const TileDesc td = a.desc[tidx];                     // block-uniform: scalar loads
    const i64 tile = td.tile; const i32 n = td.n; const i64 off = td.off; const i32 ch = td.ch;
    const i64 nb = td.nb, L = td.L;
    const int lead = d.store_before + d.samples_before;
    i32 *dst = a.raw + td.dst;
    const int mode = td.mode;
    const double G = td.G;
    const double *spe_row = d.spe + (size_t)(d.n_spe > 1 ? ch : 0) * 2001;
    const double thr = td.thr;
    const i64 rel0 = td.rel0;                             // ns of the tile's first start bin relative to t0

    const i64 n_live = nb + (tlen - 1);                   // samples lead .. lead + n_live - 1 can be non-zero
    if (win > 0 && (i64)win * TPB >= n_live) return;      // no chunk for this workgroup (block-uniform)
    // number of DPE photons of the tile (truth quirk pulse.py:255); truth is window 0's job.  A tile that fits one
    // register batch is counted from the registers inside the photon loop, longer ones with a scan of their own.
    i32 n_dpe_tile = 0;
    const bool one_batch = n <= TPB * DENSE_PPT;
    if (!RESIDENT && win == 0 && !one_batch) {
        i32 c = 0;
        for (i32 p = tid; p < n; p += TPB) c += (a.ph[off + p].code >> 16) != 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) wsum[wid] = (u32)c;
        __syncthreads();
        for (int w = 0; w < TPB / 64; w++) n_dpe_tile += (i32)wsum[w];
    }
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // n, n_dpe, n_trig, n_trig_dpe, sum g, sum g trig, sum t, sum t^2
    STAMP_INIT;

    // A tile that fits one register batch and is worked on by a single workgroup (n_win == 1) is read ONCE: its photons
    // (ns, gain) stay in registers over all windows, and the truth sums are taken here.
    i32 r_ns[DENSE_PPT]; double r_gain[DENSE_PPT];
    if (RESIDENT) {
        u32 code[DENSE_PPT];
        // The channel's SPE row (16 KB) goes through LDS (H is still free): two data-dependent 8-byte gathers per photon from
        // global memory cost the texture addresser ~64 cycles per wave instruction, from LDS a few.  Its loads are issued
        // together with the photon loads: one global round trip for both.
        // They bypass the registers (global_load_lds: 16 bytes per lane, a wave fills 1 KB of LDS per instruction).
        const bool spe_lds = mode == 0 && a.spe_lds;         // block-uniform
        if (spe_lds) {
            constexpr int NIT = (2001 * 8 + TPB * 16 - 1) / (TPB * 16);
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                // first double of this lane's 16 bytes; the lanes behind the row re-read its last entry (index 2000: its second
                // half is the first entry of the next row or, behind the last row, the slack every device buffer has)
                const int i = (it * TPB + tid) * 2;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(spe_row + (i < 2001 ? i : 2000)),
                                                 (__attribute__((address_space(3))) void *)(H + (it * TPB + (tid & ~63)) * 2), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < DENSE_PPT; k++) {
            const i32 p = tid + k * TPB;
            const bool v = p < n;
            const PhotonRec rec = a.ph[off + (v ? p : 0)];            // branch-free: tiles on the work list have n >= 1
            r_ns[k] = v ? (i32)(rec.t - rel0) : -1;
            code[k] = v ? rec.code : 0u;
            r_gain[k] = 0.0;
        }
        // explicit gains (afterpulse / injected photons) in a loop of their own: a conditional load inside the loop above makes
        // the compiler wait for every photon load before the branch -- eight memory round trips in a row instead of one
        if (mode != 0) {
#pragma unroll
            for (int k = 0; k < DENSE_PPT; k++) { const i32 p = tid + k * TPB; const double g = a.ph_gain[off + (p < n ? p : 0)]; r_gain[k] = p < n ? g : 0.0; }
        }
        {
            i32 c = 0;
#pragma unroll
            for (int k = 0; k < DENSE_PPT; k++) c += (code[k] >> 16) != 0;
            for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
            if (lane == 0) wsum[wid] = (u32)c;
            if (spe_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the SPE row has landed in LDS
            __syncthreads();
            for (int w = 0; w < TPB / 64; w++) n_dpe_tile += (i32)wsum[w];
        }
        if (mode == 0) {
            if (spe_lds) {
                // (in two halves: sixteen 8-byte values in flight at once cost the registers of a seventh wave per SIMD)
#pragma unroll
                for (int h0 = 0; h0 < DENSE_PPT; h0 += DENSE_PPT / 2) {
                    double s1[DENSE_PPT / 2], s2[DENSE_PPT / 2];
#pragma unroll
                    for (int k = 0; k < DENSE_PPT / 2; k++) { s1[k] = H[code[h0 + k] & 0xffffu]; s2[k] = H[code[h0 + k] >> 16]; }     // H[0] for "no second PE": unused
#pragma unroll
                    for (int k = 0; k < DENSE_PPT / 2; k++) {
                        double gk = G * s1[k];                          // pulse.py:97-98
                        if (code[h0 + k] >> 16) gk += G * s2[k];        // pulse.py:101-103
                        r_gain[h0 + k] = gk;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();                             // H is reused below
            } else {
                double s1[DENSE_PPT], s2[DENSE_PPT];
#pragma unroll
                for (int k = 0; k < DENSE_PPT; k++) {
                    s1[k] = spe_row[code[k] & 0xffffu];                // index 0 for the unused slots: a valid address
                    s2[k] = 0;
                    if (code[k] >> 16) s2[k] = spe_row[code[k] >> 16];
                }
#pragma unroll
                for (int k = 0; k < DENSE_PPT; k++) {
                    double gk = G * s1[k];                              // pulse.py:97-98
                    if (code[k] >> 16) gk += G * s2[k];                 // pulse.py:101-103
                    r_gain[k] = gk;
                }
            }
        }
        STAMP(d, 8);
        // truth sums of the tile (pulse.py:229-271).  The four counts need no vector reduction (tile_count, the DPE count
        // above, wave ballots); the four float sums go through LDS (H is still free) and one shuffle tree per quantity.
        u32 c_trig = 0, c_trig_dpe = 0;                   // wave-uniform
        double sg = 0, sgt = 0, st = 0, st2 = 0;
#pragma unroll
        for (int k = 0; k < DENSE_PPT; k++) {
            const bool v = r_ns[k] >= 0;
            const int r = v ? r_ns[k] % dt : 0;
            const bool above = v && (r_gain[k] * s_cmax[r] * d.c2a > thr);
            c_trig += (u32)__popcll(ballot64(above));
            c_trig_dpe += (u32)__popcll(ballot64(above && tid + k * TPB < n_dpe_tile));
            if (v) {
                sg += r_gain[k];
                if (above) sgt += r_gain[k];
                const double tr = (double)(r_ns[k] + rel0);
                st += tr; st2 += tr * tr;
            }
        }
        if (a.tile_truth) {
            constexpr int NW = TPB / 64;
            double *S = H;                                 // [4][TPB]
            S[0 * TPB + tid] = sg; S[1 * TPB + tid] = sgt; S[2 * TPB + tid] = st; S[3 * TPB + tid] = st2;
            if (lane == 0) { wsum[NW + wid * 2] = c_trig; wsum[NW + wid * 2 + 1] = c_trig_dpe; }
            __syncthreads();
            for (int q = wid; q < 4; q += NW) {
                double x = 0;
#pragma unroll
                for (int j = 0; j < NW; j++) x += S[q * TPB + j * 64 + lane];
                for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
                if (lane == 0) a.tile_truth[tile * 8 + 4 + q] = x;
            }
            if (tid == 0) {
                u32 t2 = 0, t3 = 0;
                for (int w = 0; w < NW; w++) { t2 += wsum[NW + w * 2]; t3 += wsum[NW + w * 2 + 1]; }
                double *o = a.tile_truth + tile * 8;
                o[0] = (double)n; o[1] = (double)n_dpe_tile; o[2] = (double)t2; o[3] = (double)t3;
            }
        }
    }

    STAMP(d, 9);
    // samples before the first start bin can reach (lead of them) and behind the reach of the last one are exactly zero and the
    // accumulators start from zero: nothing to do for them (the debug copy of the currents wants every sample)
    if (a.currents && win == 0) for (i64 sz = tid; sz < L; sz += TPB) if (sz < lead || sz >= lead + n_live) a.currents[a.cur_off[tidx] + sz] = 0.0;
    for (i64 c0 = (i64)win * TPB; c0 < n_live; c0 += (i64)a.n_win * TPB) {
        const bool first = (c0 == 0);
        const i64 b_lo = c0 - (tlen - 1);                  // start bin of H row 0 (may be negative: those rows stay empty)
        __syncthreads();
        for (int i = tid; i < HROWS * dt; i += TPB) H[i] = 0.0;
        __syncthreads();
        STAMP(d, 10);
        if (RESIDENT) {
            const i32 ns_lo = (i32)b_lo * dt, ns_hi = ns_lo + HROWS * dt;   // the chunk's start bins in ns relative to the tile's first start bin
#pragma unroll
            for (int k = 0; k < DENSE_PPT; k++) {
                if (r_ns[k] < 0 || r_ns[k] < ns_lo || r_ns[k] >= ns_hi) continue;
                atomicAdd(&H[r_ns[k] - ns_lo], r_gain[k]);                  // (start bin - b_lo) * dt + r
            }
        } else
        // ---- photons -> H, in register batches so that the global loads of a batch are all in flight together
        for (i32 base = 0; base < n; base += TPB * DENSE_PPT) {
            i32 ns[DENSE_PPT]; u32 code[DENSE_PPT]; double gain[DENSE_PPT]; bool use[DENSE_PPT];
#pragma unroll
            for (int k = 0; k < DENSE_PPT; k++) {
                const i32 p = base + tid + k * TPB;
                const bool v = p < n;
                const PhotonRec rec = a.ph[off + (v ? p : 0)];        // branch-free: tiles on the work list have n >= 1
                ns[k] = v ? (i32)(rec.t - rel0) : -1;
                code[k] = v ? rec.code : 0u;
                const i64 jw = (i64)(ns[k] / dt) - b_lo;
                use[k] = v && (first || (jw >= 0 && jw < HROWS));      // chunk 0 needs every photon (truth), the others only their own
                gain[k] = 0.0;
            }
            if (mode != 0) {                                 // (a loop of its own: see the resident form)
#pragma unroll
                for (int k = 0; k < DENSE_PPT; k++) { const i32 p = base + tid + k * TPB; const double g = a.ph_gain[off + (p < n ? p : 0)]; gain[k] = use[k] ? g : 0.0; }
            }
            if (first && one_batch) {                // (one batch, several workgroups per tile)
                i32 c = 0;
#pragma unroll
                for (int k = 0; k < DENSE_PPT; k++) c += (code[k] >> 16) != 0;
                for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
                if (lane == 0) wsum[wid] = (u32)c;
                __syncthreads();
                for (int w = 0; w < TPB / 64; w++) n_dpe_tile += (i32)wsum[w];
            }
            if (mode == 0) {
                double s1[DENSE_PPT], s2[DENSE_PPT];
#pragma unroll
                for (int k = 0; k < DENSE_PPT; k++) {
                    s1[k] = 0; s2[k] = 0;
                    if (use[k]) { s1[k] = spe_row[code[k] & 0xffffu]; if (code[k] >> 16) s2[k] = spe_row[code[k] >> 16]; }
                }
#pragma unroll
                for (int k = 0; k < DENSE_PPT; k++) {
                    double gk = G * s1[k];                              // pulse.py:97-98
                    if (code[k] >> 16) gk += G * s2[k];                 // pulse.py:101-103
                    gain[k] = gk;
                }
            }
#pragma unroll
            for (int k = 0; k < DENSE_PPT; k++) {
                if (!use[k]) continue;
                const int bin = ns[k] / dt, r = ns[k] - bin * dt;
                const i64 jw = bin - b_lo;
                if (jw >= 0 && jw < HROWS) atomicAdd(&H[(int)jw * dt + r], gain[k]);
                if (first) {
                    const bool above = gain[k] * s_cmax[r] * d.c2a > thr;
                    const bool is_dpe = (code[k] >> 16) != 0;
                    acc[0] += 1; acc[1] += is_dpe; acc[4] += gain[k];
                    if (above) { acc[2] += 1; acc[5] += gain[k]; if (base + tid + k * TPB < n_dpe_tile) acc[3] += 1; }
                    const double tr = (double)(ns[k] + rel0);
                    acc[6] += tr; acc[7] += tr * tr;
                }
            }
        }
        __syncthreads();
        STAMP(d, 11);
        // ---- every sample gathers its tlen x dt possible contributions in ascending time (pulse.py:303-318): sample c0 + tid
        // reads the rows tid (tap k = 21) .. tid + 21 (tap k = 0).  The tap loop is outermost so that only the dt taps of one
        // k are live in SGPRs.
        const bool act = c0 + tid < n_live;
        if (any64(act)) {                                  // wave-uniform
            const double c = tap_block<FMA>(H, W2, tp, tid, a.sparse_max, HROWS * dt);
            if (act) {
                const i64 sx = lead + c0 + tid;            // sample of the tile
                if (a.currents) a.currents[a.cur_off[tidx] + sx] = c;
                const i64 adc = -(i64)rint(c * d.c2a);     // rawdata.py:236, np.around = round half to even
                if (adc != 0) atomicAdd(&dst[sx], (i32)adc);
            }
        }
        STAMP(d, 12);
    }

    if (!RESIDENT && a.tile_truth && win == 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 64);
        __syncthreads();
        if (lane == 0) for (int q = 0; q < 8; q++) red[wid * 8 + q] = acc[q];
        __syncthreads();
        if (tid < 8) { double sum = 0; for (int w = 0; w < TPB / 64; w++) sum += red[w * 8 + tid]; a.tile_truth[tile * 8 + tid] = sum; }
    }
}

// Any digitiser geometry (sample_duration dt <= WFS_MAX_DT ns, template length tlen <= WFS_MAX_TLEN samples; pulse.py:146-187 builds the
// templates for whatever the config says): the same H-table scheme with the geometry taken from the config at run time -- merged
// gains H[start bin][ns remainder] in LDS, every sample gathers its tlen x dt possible contributions in ascending time with a
// separate multiply and add, templates read from LDS.  One workgroup per tile, chunks of TPB samples, the tile's photons re-read per
// chunk; none of the specialisations of the 10 ns / 22 tap kernels (they keep the taps in SGPRs and unroll over dt): this is the
// path of every tile when the configuration differs from the XENONnT TPC digitiser.  Same bits as the reference (add_current).
#define WFS_MAX_DT 16
#define WFS_MAX_TLEN 256
template <int TPB, bool FMA>
__global__ __launch_bounds__(TPB) void k_pulse_generic(WfsDev d, PulseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int dt = d.dt, tlen = d.tlen;
    const int HROWS = TPB + tlen - 1;
    double *H = (double *)smem;                           // [HROWS][dt]
    double *T = H + (size_t)HROWS * dt;                   // [dt][tlen] templates
    double *red = T + (size_t)dt * tlen;                  // [TPB / 64][8]
    __shared__ double s_cmax[WFS_MAX_DT];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < dt * tlen; i += TPB) T[i] = d.templates[i];
    if (tid < dt) s_cmax[tid] = d.current_max[tid];
    const TileDesc td = a.desc[blockIdx.x];
    const i64 tile = td.tile; const i32 n = td.n; const i64 off = td.off;
    const i64 nb = td.nb;
    const int lead = d.store_before + d.samples_before;
    i32 *dst = a.raw + td.dst;
    const double G = td.G, thr = td.thr;
    const double *spe_row = d.spe + (size_t)(d.n_spe > 1 ? td.ch : 0) * 2001;
    const i64 rel0 = td.rel0;
    const i64 n_live = nb + (tlen - 1);
    // number of DPE photons of the tile (truth quirk pulse.py:255)
    __shared__ i32 s_ndpe;
    if (tid == 0) s_ndpe = 0;
    __syncthreads();
    {
        i32 c = 0;
        for (i32 p = tid; p < n; p += TPB) c += (a.ph[off + p].code >> 16) != 0;
        c = wave_sum(c);
        if (lane == 0 && c) atomicAdd(&s_ndpe, c);
    }
    __syncthreads();
    const i32 n_dpe_tile = s_ndpe;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // n, n_dpe, n_trig, n_trig_dpe, sum g, sum g trig, sum t, sum t^2
    if (a.currents) for (i64 sz = tid; sz < td.L; sz += TPB) if (sz < lead || sz >= lead + n_live) a.currents[a.cur_off[blockIdx.x] + sz] = 0.0;
    for (i64 c0 = 0; c0 < n_live; c0 += TPB) {
        const bool first = c0 == 0;
        const i64 b_lo = c0 - (tlen - 1);
        __syncthreads();
        for (int i = tid; i < HROWS * dt; i += TPB) H[i] = 0.0;
        __syncthreads();
        for (i32 p = tid; p < n; p += TPB) {
            const PhotonRec rec = a.ph[off + p];
            const i32 ns = (i32)(rec.t - rel0);
            const int bin = ns / dt, r = ns - bin * dt;
            const i64 jw = bin - b_lo;
            const bool mine = jw >= 0 && jw < HROWS;
            if (!first && !mine) continue;
            double gain;
            if (td.mode != 0) gain = a.ph_gain[off + p];
            else { gain = G * spe_row[rec.code & 0xffffu]; if (rec.code >> 16) gain += G * spe_row[rec.code >> 16]; }      // pulse.py:97-103
            if (mine) atomicAdd(&H[(int)jw * dt + r], gain);
            if (first) {
                const bool above = gain * s_cmax[r] * d.c2a > thr;
                acc[0] += 1; acc[1] += (rec.code >> 16) != 0; acc[4] += gain;
                if (above) { acc[2] += 1; acc[5] += gain; if (p < n_dpe_tile) acc[3] += 1; }
                const double tr = (double)(ns + rel0);
                acc[6] += tr; acc[7] += tr * tr;
            }
        }
        __syncthreads();
        if (c0 + tid < n_live) {
            // sample c0 + tid sees the start bins (rows) tid (tap tlen - 1) .. tid + tlen - 1 (tap 0), ascending time (pulse.py:303-318)
            double c = 0.0;
            for (int k = tlen - 1; k >= 0; k--) {
                const double *hp = H + (size_t)(tid + (tlen - 1) - k) * dt;
                for (int r = 0; r < dt; r++) c = mac<FMA>(T[r * tlen + k], hp[r], c);
            }
            const i64 sx = lead + c0 + tid;
            if (a.currents) a.currents[a.cur_off[blockIdx.x] + sx] = c;
            const i64 adc = -(i64)rint(c * d.c2a);         // rawdata.py:236
            if (adc != 0) atomicAdd(&dst[sx], (i32)adc);
        }
    }
    if (a.tile_truth) {
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = wave_sum(acc[q]);
        __syncthreads();
        if (lane == 0) for (int q = 0; q < 8; q++) red[wid * 8 + q] = acc[q];
        __syncthreads();
        if (tid < 8) { double sum = 0; for (int w = 0; w < TPB / 64; w++) sum += red[w * 8 + tid]; a.tile_truth[tile * 8 + tid] = sum; }
    }
}

// Tiny tiles (a few photons in a few start bins: almost every tile of an S1): one THREAD per tile.  The photons are
// sorted by time in registers, equal-ns photons merged (first of the run carries the summed gain), and every sample in
// reach of a photon adds templates[r][k] * gain in ascending time with a separate multiply and add: the arithmetic of
// add_current (pulse.py:276-318), hence the same bits as the other two kernels.  Truth sums need no reduction.
// The pulse of a tile of at most TINY_MAX_PHOTONS photons (see k_pulse_tiny) in two steps.  tiny_tile_prepare: the photons sorted by
// time, runs of equal ns merged, start bin and ns remainder of each, the samples they reach -- and the tile's truth sums (sub == 0).
// tiny_tile_samples: lane `sub` of LANES takes the samples s_first + sub, + LANES, ...; tap(r, k) = templates[r][k], sink(s, adc)
// receives the non-zero ADC counts.
struct TinyPrep { int jb[TINY_MAX_PHOTONS], rr[TINY_MAX_PHOTONS]; double g[TINY_MAX_PHOTONS]; int n, s_first, s_last; };
__device__ __forceinline__ TinyPrep tiny_tile_prepare(const WfsDev &d, const PulseArgs &a, const TileDesc &td, int sub, const double *s_cmax)
{
    constexpr int dt = WFS_DT, tlen = 22, NP = TINY_MAX_PHOTONS;
    const int n = td.n, L = td.L;
    const int lead = d.store_before + d.samples_before;
    const double G = td.G;
    const double *spe_row = d.spe + (size_t)(d.n_spe > 1 ? td.ch : 0) * 2001;
    i32 ns[NP]; double g[NP]; u32 code[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        const bool v = k < n;
        const PhotonRec rec = a.ph[td.off + (v ? k : 0)];
        ns[k] = v ? (i32)(rec.t - td.rel0) : 0x7fffffff;
        code[k] = v ? rec.code : 0u;
        g[k] = 0.0;
        if (v) {
            if (td.mode != 0) g[k] = a.ph_gain[td.off + k];
            else { g[k] = G * spe_row[code[k] & 0xffffu]; if (code[k] >> 16) g[k] += G * spe_row[code[k] >> 16]; }      // pulse.py:97-103
        }
    }
    if (a.tile_truth && sub == 0) {        // pulse.py:229-271, photons in their order in the channel slice
        i32 n_dpe = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) n_dpe += (code[k] >> 16) != 0;
        double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NP; k++) {
            if (k >= n) continue;
            const bool above = g[k] * s_cmax[ns[k] % dt] * d.c2a > td.thr;
            v[0] += 1; v[1] += (code[k] >> 16) != 0; v[4] += g[k];
            if (above) { v[2] += 1; v[5] += g[k]; if (k < n_dpe) v[3] += 1; }
            const double tr = (double)(ns[k] + td.rel0);
            v[6] += tr; v[7] += tr * tr;
        }
        for (int q = 0; q < 8; q++) a.tile_truth[(i64)td.tile * 8 + q] = v[q];
    }
    // ---- ascending time (stable: equal ns keep their order), then merge runs of equal ns
#pragma unroll
    for (int i = 1; i < NP; i++)
#pragma unroll
        for (int j = i; j > 0; j--)
            if (ns[j] < ns[j - 1]) { const i32 t = ns[j]; ns[j] = ns[j - 1]; ns[j - 1] = t; const double x = g[j]; g[j] = g[j - 1]; g[j - 1] = x; }
    // run heads collect the gains of their run in order; the others become 0.0 (adding 0.0 changes nothing)
    {
        int head = 0;
#pragma unroll
        for (int k = 1; k < NP; k++) {
            if (k < n && ns[k] == ns[k - 1]) {
#pragma unroll
                for (int q = 0; q < NP; q++) if (q == head) g[q] += g[k];
                g[k] = 0.0;
            } else head = k;
        }
    }
    int jb[NP], rr[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) { jb[k] = (k < n) ? ns[k] / dt : 0x3fffffff; rr[k] = (k < n) ? ns[k] - jb[k] * dt : 0; }
    // samples in reach of a photon: start bin j touches samples j + lead .. j + lead + tlen - 1
    const int s_first = a.currents ? 0 : jb[0] + lead;
    int jmax = 0;
#pragma unroll
    for (int k = 0; k < NP; k++) if (k < n) jmax = jb[k] > jmax ? jb[k] : jmax;
    const int s_last = a.currents ? L - 1 : (jmax + lead + tlen - 1 < L - 1 ? jmax + lead + tlen - 1 : L - 1);
    TinyPrep q; q.n = n; q.s_first = s_first; q.s_last = s_last;
#pragma unroll
    for (int k = 0; k < NP; k++) { q.jb[k] = jb[k]; q.rr[k] = rr[k]; q.g[k] = g[k]; }
    return q;
}

template <bool FMA, int LANES, bool CUR, class Tap, class Sink>
__device__ __forceinline__ void tiny_tile_samples(const WfsDev &d, const PulseArgs &a, const TinyPrep &q, i64 idx, int sub, Tap tap, Sink sink)
{
    constexpr int tlen = 22, NP = TINY_MAX_PHOTONS;
    const int lead = d.store_before + d.samples_before;
    for (int s = q.s_first + sub; s <= q.s_last; s += LANES) {
        double cur = 0.0;
#pragma unroll
        for (int k = 0; k < NP; k++) {
            const int kk = s - lead - q.jb[k];
            if (k < q.n && kk >= 0 && kk < tlen) cur = mac<FMA>(tap(q.rr[k], kk), q.g[k], cur);
        }
        if (CUR && a.currents) a.currents[a.cur_off[idx] + s] = cur;
        const i64 adc = -(i64)rint(cur * d.c2a);                 // rawdata.py:236
        if (adc != 0) sink(s, (i32)adc);
    }
}

template <bool FMA, int LANES, class Tap, class Sink>
__device__ __forceinline__ void tiny_tile_pulse(const WfsDev &d, const PulseArgs &a, const TileDesc &td, i64 idx, int sub, const double *s_cmax, Tap tap, Sink sink)
{
    const TinyPrep q = tiny_tile_prepare(d, a, td, sub, s_cmax);
    tiny_tile_samples<FMA, LANES, true>(d, a, q, idx, sub, tap, sink);
}

// A prepared tiny tile of a resident row (k_tile_assign -> k_row_pulse), in the 64 bytes of a TileDesc; n_neg = -photons sits where
// TileDesc::n does and tells the two apart.
struct __attribute__((aligned(64))) TinyDesc { double g[TINY_MAX_PHOTONS]; i32 dst, s_first, n_neg, s_last; u32 jr[TINY_MAX_PHOTONS]; };      // jr = start bin << 4 | ns remainder
static_assert(sizeof(TinyDesc) == 64 && sizeof(TileDesc) == 64 && offsetof(TinyDesc, n_neg) == offsetof(TileDesc, n), "descriptor layouts");

// Resident rows on: every live tile goes either into its row's list (descriptor written here, dst relative to the row's first
// sample) or onto the work list of its class -- the choice k_tile_geom cannot make, because rows exist only after k_groups.
__global__ void k_tile_assign(WfsDev d, GeomArgs a, DescArgs da, PulseArgs pa)
{
    const i64 tile = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = tile < a.n_tiles && a.tile_count[tile] > 0;
    int cls = 2; i64 nb = 0; i32 cnt = 0; bool listed = false;
    if (live) {
        const i64 set = tile / d.n_tpc; const i32 ch = (i32)(tile - set * d.n_tpc);
        const i64 t0 = a.set_t0[set];
        i64 left, right, bin0;
        tile_bounds(d, t0, a.tile_tmin[tile], a.tile_tmax[tile], left, right, bin0, nb);
        cnt = a.tile_count[tile];
        const i64 ridx = (i64)a.cl_group[a.set_cluster[set]] * d.n_tpc + ch;
        const i64 first = a.res_toff[ridx];
        if (a.res_toff[ridx + 1] > first) {
            const i32 k = atomicSub(&a.res_cnt[ridx], 1) - 1;          // (the scan has consumed the counts: they serve as the rows' cursors)
            TileDesc t;
            t.off = da.tile_off[tile]; t.dst = left - (a.row_lo[ridx] - d.tw); t.rel0 = bin0 * d.dt - t0;
            t.G = d.gains[ch]; t.thr = d.thr_truth[ch];
            t.n = cnt; t.nb = (i32)nb; t.L = (i32)(right - left + 1); t.tile = (i32)tile;
            t.ch = ch; t.mode = da.set_mode[set];
            if (cnt <= TINY_MAX_PHOTONS && nb <= TINY_MAX_BINS) {       // (the tiny class of the work lists: the same tiles, the same order of their truth sums)
                // a handful of photons: everything that does not depend on the sample is done here, one LANE per tile (sorted and merged
                // photons, their start bins, the tile's truth sums) -- in k_row_pulse a whole wave would do it for one tile
                const TinyPrep q = tiny_tile_prepare(d, pa, t, 0, d.current_max);
                TinyDesc y;
                y.dst = (i32)t.dst; y.s_first = q.s_first; y.s_last = q.s_last; y.n_neg = -q.n;
#pragma unroll
                for (int j = 0; j < TINY_MAX_PHOTONS; j++) { y.g[j] = q.g[j]; y.jr[j] = ((u32)q.jb[j] << 4) | (u32)q.rr[j]; }
                *(TinyDesc *)&a.res_desc[first + k] = y;
            } else a.res_desc[first + k] = t;
        } else {
            cls = tile_class(a, cnt, nb);
            listed = !(a.tile_done && tile < a.n_done && a.tile_done[tile]);
        }
    }
    tile_list_append(a, listed, cls, nb, cnt, tile);
}

template <bool FMA>
__global__ __launch_bounds__(256) void k_pulse_tiny(WfsDev d, PulseArgs a, i64 n_tiny)
{
    constexpr int dt = WFS_DT, tlen = 22, NP = TINY_MAX_PHOTONS;
    __shared__ double sT[dt * tlen];                      // templates[r][k]
    __shared__ double s_cmax[dt];
    for (int i = threadIdx.x; i < dt * tlen; i += blockDim.x) sT[i] = d.templates[i];
    if (threadIdx.x < dt) s_cmax[threadIdx.x] = d.current_max[threadIdx.x];
    __syncthreads();
    // TINY_LANES lanes per tile: all of them load, sort and merge the tile's (at most 4) photons -- the same registers in each, no
    // exchange -- and share the tile's samples, lane j taking s_first + j, + TINY_LANES, ...: the adds of a tile land in runs of 64
    // bytes instead of 64 rows per instruction, and a wave's loop is as long as its longest tile divided by TINY_LANES.
    const i64 gtid = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 idx = gtid / TINY_LANES; const int sub = (int)(gtid % TINY_LANES);
    if (idx >= n_tiny) return;
    const TileDesc td = a.desc[idx];
    i32 *dst = a.raw + td.dst;
    tiny_tile_pulse<FMA, TINY_LANES>(d, a, td, idx, sub, s_cmax, [&](int r_, int k_) { return sT[r_ * tlen + k_]; }, [&](int s_, i32 adc) { atomicAdd(&dst[s_], adc); });
}

// Medium tiles: at most 64 photons over any number of start bins -- the tiles of a small S2 (a few hundred to a few thousand
// electrons: ~30 photons per PMT over microseconds), far too wide for the counting-sort kernel below and far too empty for
// the dense kernel, whose cost goes with the samples (~600 x 220 multiply-adds for 30 photons x 22).  One WAVE per tile, no
// LDS traffic beyond the template table: lane = photon while the photons are loaded, sorted by time (bitonic network on
// keys ns << 6 | lane) and merged (equal ns: the first of the run takes the summed gain, the others 0.0, which add nothing);
// then lane = sample, block by block of 64: the photons that can reach the block are a contiguous range of the sorted
// list, their (bin, ns remainder, gain) are broadcast one at a time from the lane that holds them, and every sample in
// reach adds templates[r][k] * gain with a separate multiply and add, in ascending time: the arithmetic of add_current
// (pulse.py:276-318), the same bits as the other kernels.
struct __attribute__((aligned(16))) WavePhoton { i32 bin, row; double g; };       // start bin, first tap of its template row in sTz, merged gain
#define PW_U 2
#define WAVE_TZ_LEN (WFS_DT * (22 + 2))
__device__ __forceinline__ void wave_tables_fill(const WfsDev &d, double *sTz, double *s_cmax)      // (every thread of the workgroup; a barrier follows)
{
    constexpr int dt = WFS_DT, tlen = 22;
    for (int i = threadIdx.x; i < dt * (tlen + 2); i += blockDim.x) { const int r = i / (tlen + 2), k = i - r * (tlen + 2) - 1; sTz[i] = (k >= 0 && k < tlen) ? d.templates[r * tlen + k] : 0.0; }
    if (threadIdx.x < dt) s_cmax[threadIdx.x] = d.current_max[threadIdx.x];
}

// The pulse of one tile made by one wave (see k_pulse_wave): sink(s, adc) receives the rounded, non-zero ADC counts of the tile's
// sample s, every lane a different sample.
template <bool FMA, bool CUR, class Sink>
__device__ __forceinline__ void wave_tile_pulse(const WfsDev &d, const PulseArgs &a, const TileDesc &td, i64 idx, WavePhoton *wph, const double *sTz,
                                                const double *s_cmax, int lane, Sink sink)
{
    constexpr int dt = WFS_DT, tlen = 22;
    const int n = td.n, L = td.L;
    const int lead = d.store_before + d.samples_before;
    const bool v = lane < n;
    // ---- lane = photon
    const PhotonRec rec = a.ph[td.off + (v ? lane : 0)];
    const i32 ns0 = v ? (i32)(rec.t - td.rel0) : 0x7fffff;                  // ns relative to the tile's first start bin
    const u32 code = v ? rec.code : 0u;
    double g = 0.0;
    if (td.mode != 0) { if (v) g = a.ph_gain[td.off + lane]; }
    else {
        const double *spe_row = d.spe + (size_t)(d.n_spe > 1 ? td.ch : 0) * 2001;
        const double s1 = spe_row[code & 0xffffu], s2 = spe_row[code >> 16];
        g = td.G * s1;                                                       // pulse.py:97-98
        if (code >> 16) g += td.G * s2;                                      // pulse.py:101-103
        if (!v) g = 0.0;
    }
    if (a.tile_truth) {        // pulse.py:229-271, photons in their order in the channel slice (= lane order)
        const bool is_dpe = v && (code >> 16) != 0;
        const i32 n_dpe = __popcll(ballot64(is_dpe));
        const bool above = v && (g * s_cmax[v ? ns0 % dt : 0] * d.c2a > td.thr);
        const i32 n_trig = __popcll(ballot64(above)), n_trig_dpe = __popcll(ballot64(above && lane < n_dpe));
        const double tr = v ? (double)(ns0 + td.rel0) : 0.0;
        // (DPP: no LDS crossbar trips; every photon above the threshold -- the usual tile -- : the triggered area IS the area, same bits)
        const double sg = wave_sum(g), sgt = all64(!v || above) ? sg : wave_sum(above ? g : 0.0), st = wave_sum(tr), st2 = wave_sum(tr * tr);
        if (lane == 0) {
            double *o = a.tile_truth + (i64)td.tile * 8;
            o[0] = (double)n; o[1] = (double)n_dpe; o[2] = (double)n_trig; o[3] = (double)n_trig_dpe; o[4] = sg; o[5] = sgt; o[6] = st; o[7] = st2;
        }
    }
    // ---- ascending time: bitonic sort of the keys ns << 6 | lane (unique: a deterministic order), then the gains follow
    u32 key = ((u32)ns0 << 6) | (u32)lane;
    // (the photons sit in lanes 0 .. n - 1 and every other lane holds a larger key: with n <= 32 (16) the merges of 64 (32) lanes have
    // nothing to do -- the lower half is sorted ascending after the stages before them, which is all that is read)
    // (the exchanges with lane ^ 1, 2, 8 are DPP moves, ^ 4, 16 swizzles: 19 of the 21 steps without an LDS address, wfs_device.h)
    key = bitonic_merge<2>(key, lane); key = bitonic_merge<4>(key, lane); key = bitonic_merge<8>(key, lane); key = bitonic_merge<16>(key, lane);
    if (n > 16) { key = bitonic_merge<32>(key, lane); if (n > 32) key = bitonic_merge<64>(key, lane); }      // wave-uniform
    const int src = (int)(key & 63u);
    i32 ns = (i32)(key >> 6);
    g = __shfl(g, src, 64);
    // runs of equal ns (pulse.py:303-313): the head of a run collects the gains of the run in order, the others become 0.0
    {
        const i32 prev = __shfl_up(ns, 1, 64);
        const bool head = lane == 0 || prev != ns;
        if (any64(!head && lane < n)) {
            double tot = g;
            for (int step = 1; step < 64; step++) {
                const i32 nsq = __shfl_down(ns, step, 64); const double gq = __shfl_down(g, step, 64);
                const bool same = lane + step < n && nsq == ns;
                if (!any64(head && same)) break;
                if (head && same) tot += gq;
            }
            g = head ? tot : 0.0;
        }
    }
    const i32 bin = lane < n ? ns / dt : 0x3fffffff, r = lane < n ? ns - (ns / dt) * dt : 0;
    if (lane < PW_U) wph[64 + lane] = WavePhoton{0x3fffffff, 1, 0.0};
    wph[lane] = WavePhoton{bin, r * (tlen + 2) + 1, g};    // (read back by every lane of this wave: one 16-byte LDS broadcast per photon
                                                          //  instead of three readlanes and their scalar unpacking on the vector unit)
    // ---- lane = sample.  Live samples: s' = 0 .. nb + 20 (sample lead + s' of the tile sees the start bins s' - 21 .. s').
    // Blocks of 64 samples start where something lands: behind a block the next one begins at the first start bin that can still
    // reach a sample not yet written (a photon of the thin tail of an S2 is alone in its 22 samples; whole blocks between the
    // photons are skipped).  Inside a block every sample walks the photons in reach, in ascending time; sTz has a zero tap on
    // either side of every template row, so a photon out of a sample's reach adds +0.0, which changes nothing (no exec masking).
    const int n_live = td.nb + tlen - 1;
    if (CUR && a.currents) for (int s = lane; s < L; s += 64) a.currents[a.cur_off[idx] + s] = 0.0;
    int s0 = 0;
    while (s0 < n_live) {
        // sorted photons with a start bin in [s0 - 21, s0 + 63]: a contiguous range [q_lo, q_hi) of lanes
        const int q_lo = __popcll(ballot64(bin < s0 - (tlen - 1))), q_hi = __popcll(ballot64(bin <= s0 + 63));
        if (q_lo >= n) break;                                 // nothing reaches s0 or anything behind it
        if (q_hi == q_lo) { s0 = __builtin_amdgcn_readlane(bin, q_lo); continue; }      // (bin[q_lo] > s0 + 63: the next block starts at that photon)
        const int sp = s0 + lane;
        double c = 0.0;
        // (PW_U photons per trip, their LDS reads in flight together; a photon past q_hi is out of reach of this block -- or one of
        // the pad entries behind the list -- and adds +0.0)
        for (int q = q_lo; q < q_hi; q += PW_U) {
            WavePhoton w[PW_U]; double tap[PW_U];
#pragma unroll
            for (int u = 0; u < PW_U; u++) w[u] = wph[q + u];
#pragma unroll
            for (int u = 0; u < PW_U; u++) { int kk = sp - w[u].bin; kk = kk < -1 ? -1 : (kk > tlen ? tlen : kk); tap[u] = sTz[w[u].row + kk]; }
#pragma unroll
            for (int u = 0; u < PW_U; u++) c = mac<FMA>(tap[u], w[u].g, c);
        }
        if (sp < n_live) {
            if (CUR && a.currents) a.currents[a.cur_off[idx] + lead + sp] = c;
            const double x = rint(c * d.c2a);                  // rawdata.py:236
            // (|x| < 2^31 unless a current is absurd: the one-instruction conversion; the general one otherwise, same value)
            const i32 adc = fabs(x) < 2147483648.0 ? -(i32)x : (i32)(-(i64)x);
            if (adc != 0) sink(lead + sp, adc);
        }
        s0 += 64;
    }
}

template <bool FMA>
__global__ __launch_bounds__(256) void k_pulse_wave(WfsDev d, PulseArgs a, i64 n_wave)
{
    constexpr int dt = WFS_DT, tlen = 22;
    __shared__ double sTz[WAVE_TZ_LEN];                   // templates[r][k] with a zero tap in front of and behind every row
    __shared__ double s_cmax[dt];
    __shared__ WavePhoton s_ph[4][64 + PW_U];             // the sorted, merged photons of each wave's tile, pad entries that reach nothing behind them
    wave_tables_fill(d, sTz, s_cmax);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const i64 idx = (i64)blockIdx.x * 4 + wave_in_block();       // (an SGPR: the tile descriptor stays scalar)
    WavePhoton *wph = s_ph[wave_in_block()];
    if (idx >= n_wave) return;                            // wave-uniform
    const TileDesc td = a.desc[idx];
    i32 *dst = a.raw + td.dst;
    wave_tile_pulse<FMA, true>(d, a, td, idx, wph, sTz, s_cmax, lane, [&](int s_, i32 adc) { atomicAdd(&dst[s_], adc); });
}

// Sparse form of the same computation, for tiles with few photons per start bin (every S1, S2s up to ~10^6 PE):
// the photons are counting-sorted by their ns in LDS (one counter per ns of the tile, two u16 counters per dword),
// photons of equal ns are merged (first of the run gets the summed gain, the others 0.0, which add nothing) --
// i.e. the sorted, merged photon list add_current walks (pulse.py:297-318) -- and every sample gathers its
// contributions from the contiguous slice of that list in ascending time.  Same bits as the dense kernel.
#define SPARSE_PPT 16      // photons per thread held in registers: tile photons <= TPB * SPARSE_PPT

template <int TPB, bool FMA>
__global__ __launch_bounds__(TPB) void k_pulse_sparse(WfsDev d, PulseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tlen = d.tlen; constexpr int dt = WFS_DT;      // sample_duration is 10 ns (checked by wfs_create)
    // T replicas: 8 copies of the templates, rows padded to tlen + 2 with a zero tap on either side, copies offset by
    // 4 bank pairs so that the 64 lanes' data-dependent template reads spread over the LDS banks
    constexpr int TROW = 24, TREP = 260, NREP = 1;       // requires tlen == 22 (tiny tiles: bank conflicts do not matter)
    double *cg = (double *)smem;                         // [NP] merged gain of the sorted photons
    double *red = cg + a.NP;                             // [TPB / 64][8]
    i32 *ctoff = (i32 *)(red + 8 * (TPB / 64));          // [NP] r * TROW - start bin
    unsigned short *boff = (unsigned short *)(ctoff + a.NP);           // [dt * W + 2] exclusive offsets
    u32 *wsum = (u32 *)(boff + ((dt * a.W + 2 + 3) & ~3));             // [TPB / 64]
    u32 *cnt = wsum + 8;                                 // [dt * W / 2] packed u16 pairs: photons per ns ...
    double *Trep = (double *)cnt;                        // ... later reused for [NREP][TREP] template copies
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    __shared__ double s_cmax[WFS_DT];                    // pulse.py:32 current_max (see k_pulse)
    if (tid < WFS_DT) s_cmax[tid] = d.current_max[tid];

    const i64 tile = a.active_tiles[blockIdx.x];
    const i32 n = a.tile_count[tile];
    const i64 off = a.tile_off[tile];
    const i64 set = tile / d.n_tpc; const i32 ch = (i32)(tile - set * d.n_tpc);
    const i64 t0 = a.set_t0[set];
    i64 left, right, bin0, nb64;
    tile_bounds(d, t0, a.tile_tmin[tile], a.tile_tmax[tile], left, right, bin0, nb64);
    const int nb = (int)nb64, nns = nb * dt;
    const int L = (int)(right - left + 1);
    const int lead = d.store_before + d.samples_before;
    const int mode = a.set_mode[set];
    const double G = d.gains[ch];
    const double *spe_row = d.spe + (size_t)(d.n_spe > 1 ? ch : 0) * 2001;
    const double thr = d.thr_truth[ch];
    const i32 rel0 = (i32)(bin0 * dt - t0);              // ns of the tile's first start bin relative to t0

    // ---- all photons of the tile into registers: independent loads, all in flight together
    i32 ns[SPARSE_PPT]; u32 code[SPARSE_PPT]; double gain[SPARSE_PPT];
    const int kmax = (n + TPB - 1) / TPB;                // block-uniform: registers beyond it stay empty
#pragma unroll
    for (int k = 0; k < SPARSE_PPT; k++) {
        const i32 p = tid + k * TPB;
        const bool v = p < n;
        const PhotonRec rec = a.ph[off + (v ? p : 0)];
        ns[k] = v ? rec.t - rel0 : -1;
        code[k] = v ? rec.code : 0u;
        gain[k] = (v && mode != 0) ? a.ph_gain[off + p] : 0.0;
    }
    if (mode == 0) {
        double s1[SPARSE_PPT], s2[SPARSE_PPT];
#pragma unroll
        for (int k = 0; k < SPARSE_PPT; k++) {
            s1[k] = 0; s2[k] = 0;
            if (k < kmax) { s1[k] = spe_row[code[k] & 0xffffu]; if (any64((code[k] >> 16) != 0)) s2[k] = spe_row[code[k] >> 16]; }
        }
#pragma unroll
        for (int k = 0; k < SPARSE_PPT; k++) {
            double gk = G * s1[k];                                  // pulse.py:97-98
            if (code[k] >> 16) gk += G * s2[k];                     // pulse.py:101-103
            gain[k] = gk;
        }
    }
    for (int i = tid; i < (nns + 1) / 2; i += TPB) cnt[i] = 0;
    i32 n_dpe_tile = 0;
    {                         // number of DPE photons of the tile (truth quirk pulse.py:255)
        i32 c = 0;
#pragma unroll
        for (int k = 0; k < SPARSE_PPT; k++) c += (code[k] >> 16) != 0;
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) wsum[wid] = (u32)c;
        __syncthreads();
        for (int w = 0; w < TPB / 64; w++) n_dpe_tile += (i32)wsum[w];
    }
    __syncthreads();

    // ---- pass 1: photons per ns, truth accumulators (pulse.py:229-271)
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // n, n_dpe, n_trig, n_trig_dpe, sum g, sum g trig, sum t, sum t^2
#pragma unroll
    for (int k = 0; k < SPARSE_PPT; k++) {
        if (ns[k] < 0) continue;
        atomicAdd(&cnt[ns[k] >> 1], 1u << ((ns[k] & 1) * 16));
        const int r = ns[k] % dt;
        const bool above = gain[k] * s_cmax[r] * d.c2a > thr;
        const bool is_dpe = (code[k] >> 16) != 0;
        acc[0] += 1; acc[1] += is_dpe; acc[4] += gain[k];
        if (above) { acc[2] += 1; acc[5] += gain[k]; if (tid + k * TPB < n_dpe_tile) acc[3] += 1; }
        const double tr = (double)(ns[k] + rel0);
        acc[6] += tr; acc[7] += tr * tr;
    }
    __syncthreads();

    // ---- exclusive scan of the per-ns counts -> boff[0, nns]
    {
        const int C = (nns + TPB - 1) / TPB;
        const int j0 = tid * C;
        u32 sum = 0;
        for (int k = 0; k < C; k++) { int j = j0 + k; if (j < nns) sum += (cnt[j >> 1] >> ((j & 1) * 16)) & 0xffffu; }
        u32 incl = sum;
        for (int o = 1; o < 64; o <<= 1) { u32 t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        u32 run = incl - sum;
        for (int w = 0; w < wid; w++) run += wsum[w];
        for (int k = 0; k < C; k++) { int j = j0 + k; if (j < nns) { boff[j] = (unsigned short)run; run += (cnt[j >> 1] >> ((j & 1) * 16)) & 0xffffu; } }
        if (tid == TPB - 1) boff[nns] = (unsigned short)n;
        __syncthreads();
        for (int i = tid; i < (nns + 1) / 2; i += TPB) cnt[i] = 0;
        __syncthreads();
    }

    // ---- pass 2: place in time order
#pragma unroll
    for (int k = 0; k < SPARSE_PPT; k++) {
        if (ns[k] < 0) continue;
        const u32 old = atomicAdd(&cnt[ns[k] >> 1], 1u << ((ns[k] & 1) * 16));
        const u32 pos = boff[ns[k]] + ((old >> ((ns[k] & 1) * 16)) & 0xffffu);
        cg[pos] = gain[k]; ctoff[pos] = (ns[k] % dt) * TROW - ns[k] / dt;
    }
    __syncthreads();

    // ---- merge photons of equal ns (pulse.py:303-313)
    for (int j = tid; j < nns; j += TPB) {
        const int q0 = boff[j], m = (int)boff[j + 1] - q0;
        if (m >= 2) {
            double tot = cg[q0];
            for (int q = 1; q < m; q++) { tot += cg[q0 + q]; cg[q0 + q] = 0.0; }
            cg[q0] = tot;
        }
    }
    __syncthreads();

    // ---- templates into LDS (the counters are dead now)
    for (int i = tid; i < NREP * TREP; i += TPB) {
        const int c = i / TREP, x = i - c * TREP, r = x / TROW, kk = x - r * TROW;
        Trep[i] = (x < dt * TROW && kk >= 1 && kk <= tlen) ? d.templates[r * tlen + kk - 1] : 0.0;
    }
    __syncthreads();

    // ---- gather: two samples per lane; sample s <- photons of start bins [s - lead - (tlen-1), s - lead], ascending time
    const i64 g = a.cl_group[a.set_cluster[set]];
    const i64 ridx = g * d.n_tpc + ch;
    i32 *dst = a.raw + a.acc_off[ridx] + (left - (a.row_lo[ridx] - d.tw));
    for (int sa = 2 * tid; sa < L; sa += 2 * TPB) {
        int jlo = sa - lead - (tlen - 1), jhi = sa + 1 - lead;
        if (jlo < 0) jlo = 0;
        if (jhi > nb - 1) jhi = nb - 1;
        double cur0 = 0.0, cur1 = 0.0;
        if (jhi >= jlo) {
            const int q1 = boff[(jhi + 1) * dt];
            // k + 1 of sample sa for a photon in start bin j is (sa - lead - j) + 1 in [0, tlen]; sample sa + 1 reads the next tap
            const double *Ts = Trep + (lane & (NREP - 1)) * TREP + (sa - lead + 1);
            int q = boff[jlo * dt];
            for (; q + 2 <= q1; q += 2) {
                const i32 o0 = ctoff[q], o1 = ctoff[q + 1];
                const double g0 = cg[q], g1 = cg[q + 1];
                cur0 = mac<FMA>(Ts[o0], g0, cur0); cur1 = mac<FMA>(Ts[o0 + 1], g0, cur1);
                cur0 = mac<FMA>(Ts[o1], g1, cur0); cur1 = mac<FMA>(Ts[o1 + 1], g1, cur1);
            }
            if (q < q1) { const i32 o0 = ctoff[q]; const double g0 = cg[q]; cur0 = mac<FMA>(Ts[o0], g0, cur0); cur1 = mac<FMA>(Ts[o0 + 1], g0, cur1); }
        }
        if (a.currents) { a.currents[a.cur_off[blockIdx.x] + sa] = cur0; if (sa + 1 < L) a.currents[a.cur_off[blockIdx.x] + sa + 1] = cur1; }
        const i64 adc0 = -(i64)rint(cur0 * d.c2a), adc1 = -(i64)rint(cur1 * d.c2a);      // rawdata.py:236
        if (adc0 != 0) atomicAdd(&dst[sa], (i32)adc0);
        if (adc1 != 0 && sa + 1 < L) atomicAdd(&dst[sa + 1], (i32)adc1);
    }

    if (a.tile_truth) {
        // (wave_sum: the order of wave_tile_pulse -- a tile of this class holds at most 32 photons, photon p in lane p of the first wave,
        // so its f64 truth sums are the same bits whether this kernel, k_pulse_wave or k_row_pulse made the tile)
#pragma unroll
        for (int q = 0; q < 8; q++) acc[q] = wave_sum(acc[q]);
        if (lane == 0) for (int q = 0; q < 8; q++) red[wid * 8 + q] = acc[q];
        __syncthreads();
        if (tid < 8) { double sum = 0; for (int w = 0; w < TPB / 64; w++) sum += red[w * 8 + tid]; a.tile_truth[tile * 8 + tid] = sum; }
    }
}

// Truth accumulators of a pulse set (pulse.py:229-271 add_truth: totals, bottom-array totals; rawdata.py:330-345 photon
// time moments) from the per-tile partial sums, one wave per set.  Channels are summed in a fixed order (lane, then the
// DPP tree of wave_sum), so the result does not depend on the order the tiles were processed in.
struct TruthArgs { i64 n_sets; const i32 *tile_count, *tile_tmin, *tile_tmax; const i64 *set_t0; const double *tile_truth;
                   double *truth; i64 *tminmax; };
__global__ __launch_bounds__(256) void k_truth_reduce(WfsDev d, TruthArgs a)
{
    const i64 set = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (set >= a.n_sets) return;
    double v[15];
#pragma unroll
    for (int q = 0; q < 15; q++) v[q] = 0.0;
    i64 tmin = I64_MAX, tmax = I64_MIN;
    const i64 t0 = a.set_t0[set];
    for (int ch = lane; ch < d.n_tpc; ch += 64) {
        const i64 tile = set * d.n_tpc + ch;
        if (a.tile_count[tile] <= 0) continue;
        const double *p = a.tile_truth + tile * 8;
        const double G = d.gains[ch];
        const double vals[6] = {p[0], p[0] + p[1], p[2], p[2] + p[3], p[4] / G, p[5] / G};
        const bool bottom = ch >= d.n_top && ch <= d.last_bottom;
#pragma unroll
        for (int q = 0; q < 6; q++) { v[q] += vals[q]; if (bottom) v[6 + q] += vals[q]; }
        v[12] += p[0]; v[13] += p[6]; v[14] += p[7];
        const i64 lo = t0 + a.tile_tmin[tile], hi = t0 + a.tile_tmax[tile];
        tmin = lo < tmin ? lo : tmin; tmax = hi > tmax ? hi : tmax;
    }
#pragma unroll
    for (int q = 0; q < 15; q++) v[q] = wave_sum(v[q]);         // (DPP: a fixed order, no LDS crossbar trips -- 180 of them per set otherwise)
    for (int o = 32; o > 0; o >>= 1) {
        const i64 lo = __shfl_down(tmin, o, 64), hi = __shfl_down(tmax, o, 64);
        tmin = lo < tmin ? lo : tmin; tmax = hi > tmax ? hi : tmax;
    }
    if (lane == 0) {
        for (int q = 0; q < 15; q++) a.truth[set * 16 + q] = v[q];
        a.tminmax[set * 2] = tmin; a.tminmax[set * 2 + 1] = tmax;
    }
}

// ------------------------------------------------------------------------------------------------ ZLE + records
struct ZleArgs {
    const i32 *active_rows; i64 n_active_rows;
    const i64 *row_lo, *row_hi, *acc_off; const i32 *raw;
    const i64 *grp_left, *grp_ixrand;
    const i64 *itv_off;
    i64 *itv_left, *itv_right; i32 *itv_n, *row_nrec;
    const i64 *rec_off;          // per row slot: first record
    uint8_t *records; i64 rec_capacity;
    i32 *row_dbg; const i64 *row_dbg_off;   // debug: finished rows
    i32 spr;                     // samples per record
    struct RowDesc *desc;        // [n_active_rows] everything a row's wave needs, prepared by k_row_desc
    const u32 *rec_dest;         // record order by (time, channel): slot of record r in the output (nullptr: row order)
    u64 *rec_key; u32 *rec_val; i64 *key_base;     // k_rec_keys: sort key (sample - *key_base) << 12 | channel, and the record index; *key_base = first sample of the batch (k_row_desc)
    const i32 *tile_done, *row_cnt, *row_tile; const i32 *ins_bcap; const i64 *ins_boff; const i32 *tbuf; i64 n_done;      // rows read from a tile buffer in place (wfs_tilegen.h)
    // resident rows (k_row_pulse): the rows behind the first n_front of the row list; their finished 16-bit samples
    i64 n_front, n_short, rows_cap; const i64 *res_toff, *fin_off; int16_t *fin; const i32 *res_long; struct ResRow *res_rows;
    struct PackDesc *pdesc;      // [n_active_rows] what k_pack needs of a row, in one scalar load (k_pack_desc)
};

// One 64-byte descriptor per active row (thread per row: the divisions and the five dependent look-ups of a row are
// done 64 rows at a time instead of once per wave; with 10^7 short S1 rows the per-row prologue was most of k_zle / k_pack)
struct __attribute__((aligned(64))) RowDesc {
    i64 acc_off;                 // first accumulated sample of the row in raw[]
    i64 row_abs;                 // absolute sample index of the row's first sample
    i64 ixr;                     // noise offset of the row's window
    i64 itv_base;                // first interval slot of the row
    i64 thr;                     // ZLE threshold of the channel
    i64 idx;                     // row slot (group * row_slots + slot): index of itv_n / row_nrec / rec_off
    i32 len, channel, he, src;   // src 1: acc_off points into the tile buffers (a row made by one k_s2_tile tile); 2: into the finished
                                 // 16-bit samples of the resident rows (k_row_pulse)
};

struct __attribute__((aligned(16))) ResRow { i64 t0; i32 n, pad; };       // a resident row's tiles: res_desc[t0 .. t0 + n)

__global__ void k_row_desc(WfsDev d, ZleArgs a)
{
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_active_rows) return;
    // row list order: accumulator rows | short resident rows (from the end of active_rows backwards, k_row_len) | long resident rows
    const i64 idx = r < a.n_front ? a.active_rows[r] : (r < a.n_front + a.n_short ? a.active_rows[a.rows_cap - 1 - (r - a.n_front)] : a.res_long[r - a.n_front - a.n_short]);
    const i64 g = idx / d.row_slots; const i32 slot = (i32)(idx - g * d.row_slots);
    const bool he = slot >= d.n_tpc; const i32 acc_ch = he ? slot - d.n_tpc : slot, channel = he ? d.he_first + acc_ch : slot;
    const i64 ridx = g * d.n_tpc + acc_ch;
    RowDesc q;
    q.acc_off = a.acc_off[ridx]; q.row_abs = a.row_lo[ridx] - d.tw; q.ixr = a.grp_ixrand[g]; q.itv_base = a.itv_off[idx];
    q.src = 0;
    if (r >= a.n_front) {
        q.acc_off = a.fin_off[ridx]; q.src = 2;
        const i64 t0 = a.res_toff[ridx];
        a.res_rows[r - a.n_front] = ResRow{t0, (i32)(a.res_toff[ridx + 1] - t0), 0};
    }
    if (row_is_direct(a.tile_done, a.n_done, a.row_cnt, a.row_tile, ridx)) {
        const i32 tile = a.row_tile[ridx]; const i32 ins = tile / d.n_tpc;
        q.acc_off = a.ins_boff[ins] + (i64)(tile - ins * d.n_tpc) * a.ins_bcap[ins]; q.src = 1;
    }
    q.thr = d.thr_zle[channel]; q.idx = idx; q.len = (i32)(a.row_hi[ridx] - a.row_lo[ridx] + 1 + 2 * (i64)d.tw);
    q.channel = channel; q.he = he ? 1 : 0;
    a.desc[r] = q;
    if (a.key_base) { atomicMin(a.key_base, q.row_abs); atomicMax(a.key_base + 7, q.row_abs + (i64)q.len); }      // (scal[22], scal[29])
}

// finished sample of a row: accumulated ADC + noise + baseline, clamped at 0
// (rawdata.py:398-458 add_noise / add_baseline / digitizer_saturation, fused into the read)
__device__ __forceinline__ i32 finish_sample(const WfsDev &d, const i32 *acc, i64 i, i32 slot_ch, bool he, i64 ix_rand)
{
    i64 v = acc[i];
    if (he) v *= d.he_factor;                                   // rawdata.py:242-246
    if (d.enable_noise && slot_ch < d.noise_channels) {
        const u32 in = ((u32)ix_rand + (u32)i) % (u32)d.noise_len;      // rawdata.py:433-434 (ix_rand < noise_len <= 2^31, i < 2^20: 32 bits hold the sum)
        // channel-major copy (wfs_set_tables): consecutive samples, consecutive addresses
        if (d.noise_f) v = (i64)((double)v + d.noise_f[(i64)slot_ch * d.noise_stride + in]);     // numba: int64 += float64 stores the truncated sum
        else v += d.noise[(i64)slot_ch * d.noise_stride + in];
    }
    v += d.baseline;
    return v < 0 ? 0 : (i32)v;
}

__device__ __forceinline__ void row_of_slot(const WfsDev &d, i64 idx, i64 &g, i32 &channel, i32 &acc_ch, bool &he)
{
    g = idx / d.row_slots; i32 slot = (i32)(idx - g * d.row_slots);
    he = slot >= d.n_tpc; acc_ch = he ? slot - d.n_tpc : slot; channel = he ? d.he_first + acc_ch : slot;
}

// The same in two steps, for loops that want ALL their loads in flight before the first use: with the load inside finish_sample
// the compiler waits for every load right behind it (a branch follows), and a wave spends one memory round trip per load.
// A row no longer than the noise array wraps at most once (ix_rand < noise_len, i < len <= noise_len): no loop on the load path.
// NK, the noise table of the batch, is a template parameter (0: none, 1: int16, 2: float64) and a row outside the table
// (noisy false, wave-uniform) reads row 0 and drops the value: NO branch sits between two loads -- with a branch around the
// noise load the compiler drained the memory counter after every one of them.
template <int NK> struct RawSample;
template <> struct RawSample<0> { i32 acc; };
template <> struct RawSample<1> { i32 acc; i32 nz; };
template <> struct RawSample<2> { i32 acc; double nzf; };
template <int NK>
__device__ __forceinline__ RawSample<NK> load_sample(const WfsDev &d, const i32 *acc, i32 i, const void *noise_row, u32 nstart, u32 noff)
{
    // (32-bit offsets from wave-uniform row pointers: the address arithmetic is one add per load, the bases stay in SGPRs)
    // noise sample nstart + noff of the row, nstart < noise_len a scalar, noff < noise_len: one conditional subtraction wraps it
    RawSample<NK> s; s.acc = acc[(u32)i];
    if constexpr (NK != 0) {
        u32 in = nstart + noff;
        in = in >= (u32)d.noise_len ? in - (u32)d.noise_len : in;
        if constexpr (NK == 2) s.nzf = ((const double *)noise_row)[in]; else s.nz = ((const int16_t *)noise_row)[in];
    }
    return s;
}
template <int NK>
__device__ __forceinline__ i32 finish_loaded(const WfsDev &d, const RawSample<NK> &s, bool he, bool noisy)
{
    i64 v = s.acc;
    if (he) v *= d.he_factor;
    if constexpr (NK == 1) v += noisy ? s.nz : 0;
    if constexpr (NK == 2) { const i64 w = (i64)((double)v + s.nzf); v = noisy ? w : v; }
    v += d.baseline;
    return v < 0 ? 0 : (i32)v;
}

// Four consecutive samples of a row per lane (k_zle): 16 bytes of the row at a 4-byte aligned address, 8 (32) bytes of the noise at a
// 2 (8)-byte aligned one -- the hardware takes both as single loads; NOISE_PAD keeps the noise read off the wrap.
template <int NK> struct Raw4;
template <> struct Raw4<0> { i32 acc[4]; };
template <> struct Raw4<1> { i32 acc[4]; int16_t nz[4]; };
template <> struct Raw4<2> { i32 acc[4]; double nzf[4]; };
template <int NK>
__device__ __forceinline__ Raw4<NK> load_four(const WfsDev &d, const i32 *acc, i32 i, const void *noise_row, u32 nstart, u32 noff)
{
    Raw4<NK> s; __builtin_memcpy(s.acc, acc + (u32)i, 16);
    if constexpr (NK != 0) {
        u32 in = nstart + noff;
        in = in >= (u32)d.noise_len ? in - (u32)d.noise_len : in;
        if constexpr (NK == 2) __builtin_memcpy(s.nzf, (const double *)noise_row + in, 32); else __builtin_memcpy(s.nz, (const int16_t *)noise_row + in, 8);
    }
    return s;
}
template <int NK>
__device__ __forceinline__ i32 finish_four(const WfsDev &d, const Raw4<NK> &s, int j, bool he, bool noisy)
{
    i64 v = s.acc[j];
    if (he) v *= d.he_factor;
    if constexpr (NK == 1) v += noisy ? (i32)s.nz[j] : 0;
    if constexpr (NK == 2) { const i64 w = (i64)((double)v + s.nzf[j]); v = noisy ? w : v; }
    v += d.baseline;
    return v < 0 ? 0 : (i32)v;
}

// The interval bookkeeping of a row whose hold-off is at least a chunk of 64 samples: two hits of one chunk are never more than the
// hold-off apart, so only the FIRST hit of a chunk can open an interval -- scalar work on the ballot of the lanes' hit nibbles.
// Closed intervals wait in lane (k mod 64) and leave in one store (flush / finish).  A store inside a loop of loads -- even one that
// is almost never executed -- may be outstanding together with loads, the two kinds retire out of order, and the compiler then waits
// for EVERYTHING at the top of the loop; the flush of a row with more than 64 intervals drains the counter itself.
struct ZleFast {
    i32 s_last = -1, s_left = -1, s_count = 0, s_nrec = 0, my_l = 0, my_r = 0;
    // close interval k = [rawl, rawr] (first / last hit): window, clip, even landing (rawdata.py:302-308); returns the records it needs
    __device__ __forceinline__ i32 close(const WfsDev &d, const ZleArgs &a, int lane, i64 base, i64 row_abs, i32 len32, i32 k, i32 rawl, i32 rawr)
    {
        i32 l = rawl - d.tw, rr = rawr + d.tw;
        l = l < 0 ? 0 : (l > len32 - 1 ? len32 - 1 : l); rr = rr < 0 ? 0 : (rr > len32 - 1 ? len32 - 1 : rr);
        l = (l + 1) / 2 * 2; rr = rr / 2 * 2;                  // ceil(l/2)*2, floor(r/2)*2 for non-negative ints
        const int slot = k & 63;
        if (lane == slot) { my_l = l; my_r = rr; }
        if (slot == 63) {
            a.itv_left[base + (k - 63) + lane] = row_abs + my_l; a.itv_right[base + (k - 63) + lane] = row_abs + my_r;      // absolute sample indices (rawdata.py:311)
            __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
        }
        const i32 plen = rr - l + 1;
        return plen > 0 ? (plen + WFS_SPR - 1) / WFS_SPR : 0;
    }
    // 256 samples: lane q holds the hits of samples i .. i + 3 (i = block start + 4 q) as a nibble; 16 lanes are a chunk of 64 samples,
    // whose first and last hit come from the ballot of the non-empty nibbles and two readlanes
    __device__ __forceinline__ void block(const WfsDev &d, const ZleArgs &a, int lane, i64 base, i64 row_abs, i32 len32, i32 hold32, u32 nib, i32 i)
    {
        const u64 mask = ballot64(nib != 0);
        if (mask == 0) return;                              // wave-uniform
        const i32 fpos = i + (i32)__builtin_ctz(nib | 16u), lpos = i + 31 - (i32)__builtin_clz(nib | 1u);      // first / last hit of the lane
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const u32 m = (u32)(mask >> (16 * c)) & 0xffffu;
            if (m == 0) continue;
            const i32 first = __builtin_amdgcn_readlane(fpos, 16 * c + (i32)__builtin_ctz(m));
            const i32 last = __builtin_amdgcn_readlane(lpos, 16 * c + 31 - (i32)__builtin_clz(m));
            if (s_last < 0 || first - s_last > hold32) {
                if (s_count > 0) s_nrec += close(d, a, lane, base, row_abs, len32, s_count - 1, s_left, s_last);
                s_left = first; s_count++;
            }
            s_last = last;
        }
    }
    __device__ __forceinline__ void finish(const WfsDev &d, const ZleArgs &a, int lane, i64 base, i64 row_abs, i32 len32, i64 idx)
    {
        if (s_count > 0) s_nrec += close(d, a, lane, base, row_abs, len32, s_count - 1, s_left, s_last);
        if (lane < (s_count & 63)) { const i64 k0 = base + (s_count & ~63); a.itv_left[k0 + lane] = row_abs + my_l; a.itv_right[k0 + lane] = row_abs + my_r; }
        if (lane == 0) { a.itv_n[idx] = s_count; a.row_nrec[idx] = s_nrec; }
    }
};

// one wave per row: find_intervals_below_threshold (utils.py:13-58) in its parallel form: consecutive
// hit samples a < b belong to one interval iff b - a <= max(holdoff, 1); then the window of rawdata.py:302-308.
template <int NK>
__global__ __launch_bounds__(256) void k_zle(WfsDev d, ZleArgs a)
{
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * 4 + wave_in_block();         // (an SGPR: the descriptor and everything derived from it stay scalar)
    if (r >= a.n_front) return;                             // (the rows behind n_front are resident: k_row_pulse has made their intervals)
    const RowDesc q = a.desc[r];
    const i64 idx = q.idx; const i32 channel = q.channel; const bool he = q.he != 0;
    const i64 len = q.len;
    const i32 *acc = (q.src ? a.tbuf : a.raw) + q.acc_off;
    const i64 thr = q.thr;
    const i64 ixr = q.ixr;
    i64 hold = 2 * (i64)d.tw + 1; if (hold < 1) hold = 1;
    const i64 base = q.itv_base;
    const i64 row_abs = q.row_abs;
    // close interval k = [rawl, rawr] (first / last hit): window, clip, even landing (rawdata.py:302-308),
    // absolute sample indices (rawdata.py:311); returns the number of records it needs
    auto close_interval = [&](i32 k, i64 rawl, i64 rawr) -> i32 {
        i64 l = rawl - d.tw, rr = rawr + d.tw;
        l = l < 0 ? 0 : (l > len - 1 ? len - 1 : l); rr = rr < 0 ? 0 : (rr > len - 1 ? len - 1 : rr);
        l = (l + 1) / 2 * 2; rr = rr / 2 * 2;                  // ceil(l/2)*2, floor(r/2)*2 for non-negative ints
        a.itv_left[base + k] = row_abs + l; a.itv_right[base + k] = row_abs + rr;
        const i32 plen = (i32)(rr - l + 1);                     // a row is shorter than 10^6 samples (k_group_final)
        return plen > 0 ? (plen + WFS_SPR - 1) / WFS_SPR : 0;
    };
    i32 carry_last = -1, open_left = -1; i32 count = 0, nrec = 0;      // (a row is shorter than 10^6 samples: 32-bit indices)
    const i32 len32 = (i32)len, hold32 = (i32)hold;
    const bool noisy = NK != 0 && channel < d.noise_channels;
    // the noise index of a row's sample i is (ix + i) mod noise_len (rawdata.py:433-434): a scalar start per block of samples that
    // steps with the block and wraps, plus the lane's offset inside the block -- rows may be any number of noise lengths long.
    // (A table shorter than NOISE_MIN_FAST samples would wrap more than once inside a block: the general path below.)
    const bool fast_loads = len32 >= 1 && (NK == 0 || d.noise_len >= NOISE_MIN_FAST);
    const i64 noise_off = noisy ? (i64)channel * d.noise_stride : 0; const u32 nixr = noisy ? (u32)ixr : 0u;
    const void *noise_row = NK == 2 ? (const void *)(d.noise_f + noise_off) : (const void *)(d.noise + noise_off);
    if (hold32 >= 63 && fast_loads && !a.row_dbg) {
        // The usual geometry (hold-off of at least a chunk): the interval bookkeeping is scalar work on the ballot mask (ZleFast), and
        // the vector unit is left with load, finish and compare.
        constexpr int G = 4;
        ZleFast z;
        // A lane takes FOUR consecutive samples (one 16-byte load of the row, one 8-byte load of the noise): a wave instruction moves
        // 1 KB of the row, and the hits of a lane are a nibble.
        // G such loads per trip: 1024 samples, most rows in one trip (6 KB in flight per wave; a second register set filled ahead
        // bought nothing -- the register allocator reuses the first set's registers for addresses and waits for its loads anyway).
        // Loads are unconditional (lanes past the end of the row read its first samples): behind a branch the compiler drains the
        // memory counter instead of counting.
        Raw4<NK> A[G];
        u32 nrun = nixr;                                        // noise index of the next block's first sample (a scalar)
        auto fetch = [&](Raw4<NK> *buf, i32 g0) {
#pragma unroll
            for (int u = 0; u < G; u++) {
                const i32 i = g0 + 256 * u + 4 * lane;
                buf[u] = load_four<NK>(d, acc, i < len32 ? i : 0, noise_row, nrun, 4u * (u32)lane);
                if constexpr (NK != 0) { nrun += 256u; nrun = nrun >= (u32)d.noise_len ? nrun - (u32)d.noise_len : nrun; }
            }
        };
        auto look = [&](const Raw4<NK> *buf, i32 g0) {
#pragma unroll
            for (int u = 0; u < G; u++) {
                const i32 b0 = g0 + 256 * u;
                if (b0 >= len32) break;                         // wave-uniform
                const i32 i = b0 + 4 * lane;
                u32 nib = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) { const i32 v = finish_four<NK>(d, buf[u], j, he, noisy); nib |= (i + j < len32 && (i64)v < thr) ? (1u << j) : 0u; }
                z.block(d, a, lane, base, row_abs, len32, hold32, nib, i);
            }
        };
        for (i32 g0 = 0; g0 < len32; g0 += 256 * G) { fetch(A, g0); look(A, g0); }
        z.finish(d, a, lane, base, row_abs, len32, idx);
        return;
    }
    // chunks of 64 samples; the loads of ZLE_GROUP chunks are issued together (one dependent load per chunk leaves the wave
    // waiting a memory round trip per 64 samples)
    constexpr int ZLE_GROUP = 2;
    u32 nrun1 = nixr;
    for (i32 g0 = 0; g0 < len32; g0 += 64 * ZLE_GROUP) {
        i32 vv[ZLE_GROUP];
        if (fast_loads) {                                       // (wave-uniform) every load of the group first, then the arithmetic
            RawSample<NK> rs[ZLE_GROUP];
#pragma unroll
            for (int u = 0; u < ZLE_GROUP; u++) {
                const i32 i = g0 + 64 * u + lane;
                rs[u] = load_sample<NK>(d, acc, i < len32 ? i : len32 - 1, noise_row, nrun1, (u32)lane);
                if constexpr (NK != 0) { nrun1 += 64u; nrun1 = nrun1 >= (u32)d.noise_len ? nrun1 - (u32)d.noise_len : nrun1; }
            }
#pragma unroll
            for (int u = 0; u < ZLE_GROUP; u++) { const i32 i = g0 + 64 * u + lane; vv[u] = i < len32 ? finish_loaded(d, rs[u], he, noisy) : 0x7fffffff; }
        } else {
#pragma unroll
            for (int u = 0; u < ZLE_GROUP; u++) { const i32 i = g0 + 64 * u + lane; vv[u] = i < len32 ? finish_sample(d, acc, i, channel, he, ixr) : 0x7fffffff; }
        }
        if (a.row_dbg) {
#pragma unroll
            for (int u = 0; u < ZLE_GROUP; u++) { const i32 i = g0 + 64 * u + lane; if (i < len32) a.row_dbg[a.row_dbg_off[r] + i] = vv[u]; }
        }
#pragma unroll
        for (int u = 0; u < ZLE_GROUP; u++) {
            const i32 c0 = g0 + 64 * u;
            if (c0 >= len32) break;                             // wave-uniform
            const i32 i = c0 + lane;
            const bool hit = i < len32 && (i64)vv[u] < thr;
            const u64 mask = ballot64(hit);
            const u64 lt = (1ull << lane) - 1ull;
            const u64 below = mask & lt;
            const i32 prev = below ? c0 + 63 - __clzll(below) : carry_last;       // last hit before this sample
            const bool start = hit && (prev < 0 || i - prev > hold32);
            const u64 smask = ballot64(start);
            if (start) {
                const u64 sbelow = smask & lt;
                const i32 k = count + __popcll(sbelow);
                const i32 prev_left = sbelow ? c0 + 63 - __clzll(sbelow) : open_left;
                if (k > 0) nrec += close_interval(k - 1, prev_left, prev);
            }
            count += __popcll(smask);
            if (smask) open_left = c0 + 63 - __clzll(smask);
            if (mask) carry_last = c0 + 63 - __clzll(mask);
        }
    }
    if (lane == 0 && count > 0) nrec += close_interval(count - 1, open_left, carry_last);
    for (int o = 32; o > 0; o >>= 1) nrec += __shfl_down(nrec, o, 64);
    if (lane == 0) { a.itv_n[idx] = count; a.row_nrec[idx] = nrec; }
}

// Resident rows: ONE wave makes a whole (window, channel) row -- zeroes it in LDS, adds the rounded pulses of the row's tiles
// (wave_tile_pulse: the arithmetic of k_pulse_wave; integer sums, so the order of the tiles does not matter), finishes the samples
// (noise, baseline, clamp: rawdata.py:398-458), runs the zero-length encoding on them and writes them ONCE, as 16-bit samples.
// Against the accumulator path (memset 4 B + atomics + 6 B read by k_zle + 6 B read by k_pack per sample) a row costs 2 B of noise
// read and 2 B written here and 2 B read by k_pack.  rawdata.py:231-239 (per-pulse rounding into the row), :302-311 (intervals).
#define ROW_LDS_FIXED ((WAVE_TZ_LEN + 16) * 8 + 4 * (64 + PW_U) * 16)
template <int NK, bool FMA, bool SEG>
__global__ __launch_bounds__(256) void k_row_pulse(WfsDev d, PulseArgs a, ZleArgs z, i64 first, i64 n_res, i32 region)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char row_lds[];
    double *sTz = (double *)row_lds; double *s_cmax = sTz + WAVE_TZ_LEN;
    WavePhoton *s_ph = (WavePhoton *)(s_cmax + 16);
    i32 *s_acc = (i32 *)(s_ph + 4 * (64 + PW_U));
    wave_tables_fill(d, sTz, s_cmax);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = wave_in_block();
    const i64 r = (i64)blockIdx.x * 4 + w;
    if (r >= n_res) return;                               // wave-uniform; no barrier below
    const RowDesc q = z.desc[z.n_front + first + r];
    const ResRow rr = z.res_rows[first + r];
    i32 *acc = s_acc + (size_t)w * region;
    WavePhoton *wph = s_ph + w * (64 + PW_U);
    const i32 len32 = q.len;
    const i64 thr = q.thr; const i64 base = q.itv_base, row_abs = q.row_abs, idx = q.idx;
    const i32 thr32 = thr > 0x7fffffffLL ? 0x7fffffff : (thr < -0x7fffffffLL ? -0x7fffffff : (i32)thr);      // (finished samples are 0 .. 2^31 - 1: the comparison is the same)
    i64 hold = 2 * (i64)d.tw + 1; if (hold < 1) hold = 1;
    const i32 hold32 = (i32)hold;
    const bool noisy = NK != 0 && q.channel < d.noise_channels;
    const i64 noise_off = noisy ? (i64)q.channel * d.noise_stride : 0;
    const void *noise_row = NK == 2 ? (const void *)(d.noise_f + noise_off) : (const void *)(d.noise + noise_off);
    u32 nrun = noisy ? (u32)q.ixr : 0u;                    // noise index of the next block's first sample (rawdata.py:433-434)
    int16_t *fin = z.fin + q.acc_off;
    const i32 fin_len = (len32 + 3) & ~3;
    auto load_noise = [&](Raw4<NK> &o) {
        if constexpr (NK != 0) {
            u32 in = nrun + 4u * (u32)lane;
            in = in >= (u32)d.noise_len ? in - (u32)d.noise_len : in;
            if constexpr (NK == 2) __builtin_memcpy(o.nzf, (const double *)noise_row + in, 32); else __builtin_memcpy(o.nz, (const int16_t *)noise_row + in, 8);
            nrun += 256u; nrun = nrun >= (u32)d.noise_len ? nrun - (u32)d.noise_len : nrun;
        }
    };
    Raw4<NK> s4;
    load_noise(s4);                                       // the first block's noise: on its way under the pulses
    const i64 t1 = rr.t0 + rr.n;
    ZleFast zf;
    // A row longer than the wave's LDS (SEG) is made segment by segment: the tiles that reach into the segment are made again for it,
    // their samples outside it dropped; the interval bookkeeping and the noise index run on across the segments.
    for (i32 seg0 = 0; seg0 < len32; seg0 += region) {
        const i32 seg_len = SEG ? (len32 - seg0 < region ? len32 - seg0 : region) : len32;
        for (i32 i = 4 * lane; i < ((seg_len + 255) & ~255); i += 256) *(int4 *)(acc + i) = make_int4(0, 0, 0, 0);
        // ---- the row's tiles (no HE rows on this path: row slot == row index)
        TileDesc td = a.desc[rr.t0];
        for (i64 t = rr.t0; t < t1; t++) {
            if (t > rr.t0) td = a.desc[t];
            if (td.n < 0) {                               // a prepared tiny tile (k_tile_assign): the samples its photons reach, nothing else
                TinyDesc y; __builtin_memcpy(&y, &td, 64);
                if (SEG && (y.dst + y.s_last < seg0 || y.dst + y.s_first >= seg0 + seg_len)) continue;
                TinyPrep p; p.n = -y.n_neg; p.s_first = y.s_first; p.s_last = y.s_last;
#pragma unroll
                for (int j = 0; j < TINY_MAX_PHOTONS; j++) { p.g[j] = y.g[j]; p.jb[j] = (int)(y.jr[j] >> 4); p.rr[j] = (int)(y.jr[j] & 15u); }
                const i32 o = y.dst - seg0;
                tiny_tile_samples<FMA, 64, false>(d, a, p, t, lane, [&](int r_, int k_) { return sTz[r_ * (22 + 2) + 1 + k_]; },
                                                  [&](int s_, i32 adc) { if (!SEG || (u32)(o + s_) < (u32)seg_len) atomicAdd(&acc[o + s_], adc); });
            } else {
                if (SEG && (td.dst + td.L <= seg0 || td.dst >= seg0 + seg_len)) continue;
                const i32 o = (i32)td.dst - seg0;
                wave_tile_pulse<FMA, false>(d, a, td, t, wph, sTz, s_cmax, lane,
                                            [&](int s_, i32 adc) { if (!SEG || (u32)(o + s_) < (u32)seg_len) atomicAdd(&acc[o + s_], adc); });      // (ds_add_u32: every lane its own sample)
            }
        }
        // ---- finish, zero-length encoding, 16-bit samples
        for (i32 b0 = 0; b0 < seg_len; b0 += 256) {
            const i32 i = seg0 + b0 + 4 * lane;           // the lane's first sample in the row
            Raw4<NK> nxt = s4;
            if (i - 4 * lane + 256 < len32) load_noise(nxt);      // the next block's noise on its way under this block's arithmetic
            *(int4 *)s4.acc = *(const int4 *)(acc + b0 + 4 * lane);
            u32 nib = 0; u32 v16[4];
            if constexpr (NK == 2) {
#pragma unroll
                for (int j = 0; j < 4; j++) { const i32 v = finish_four<NK>(d, s4, j, false, noisy); nib |= (i + j < len32 && (i64)v < thr) ? (1u << j) : 0u; v16[j] = (u32)(uint16_t)v; }
            } else {
                // integer noise: 32-bit arithmetic with a saturating add -- the values of finish_four for every accumulator that is not within
                // 2^16 of overflowing an int32 (a row of a 14-bit digitiser is some 10^5 times smaller)
                const i32 live = len32 - i;                  // samples of the lane inside the row
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    i32 add = d.baseline;
                    if constexpr (NK == 1) add += noisy ? (i32)s4.nz[j] : 0;
                    i32 v = __builtin_elementwise_add_sat(s4.acc[j], add);
                    v = v < 0 ? 0 : v;
                    nib |= (j < live && v < thr32) ? (1u << j) : 0u; v16[j] = (u32)v;
                }
            }
            if (i < fin_len) *(uint2 *)(fin + i) = make_uint2(__builtin_amdgcn_perm(v16[1], v16[0], 0x05040100u), __builtin_amdgcn_perm(v16[3], v16[2], 0x05040100u));
            zf.block(d, z, lane, base, row_abs, len32, hold32, nib, i);
            s4 = nxt;
        }
        if (!SEG) break;
    }
    zf.finish(d, z, lane, base, row_abs, len32, idx);
}

// Records ordered by (time, channel) as strax.sort_by_time leaves them (strax_interface.py:453): one key per record; the
// windows of a batch do not overlap in time, so one sort of the whole batch keeps every window contiguous.
__global__ __launch_bounds__(256) void k_rec_keys(WfsDev d, ZleArgs a)
{
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * 4 + wave_in_block();
    if (r >= a.n_active_rows) return;
    const RowDesc q = a.desc[r];
    const i32 count = a.itv_n[q.idx];
    if (count == 0) return;
    i64 rec = a.rec_off[q.idx];
    const i64 key_base = *a.key_base;
    for (i32 k = 0; k < count; k++) {
        const i64 left = a.itv_left[q.itv_base + k]; const i32 plen = (i32)(a.itv_right[q.itv_base + k] - left + 1);
        if (plen <= 0) continue;
        const i32 need = (plen + WFS_SPR - 1) / WFS_SPR;
        for (i32 f = lane; f < need; f += 64) {
            if (rec + f >= a.rec_capacity) break;
            a.rec_key[rec + f] = ((u64)(left + (i64)WFS_SPR * f - key_base) << 12) | (u64)(u32)q.channel;
            a.rec_val[rec + f] = (u32)(rec + f);
        }
        rec += need;
    }
}
__global__ void k_invert_perm(const u32 *val, u32 *dest, i64 n) { const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x; if (p < n) dest[val[p]] = (u32)p; }

// One 64-byte descriptor per active row for k_pack, made once the record offsets are known (thread per row): the row's place, its
// first record, its first interval.  With ~10^7 short rows a wave of k_pack spent most of its life in the chain row descriptor ->
// interval count and record offset -> interval bounds -> samples; now it is descriptor -> samples.
struct __attribute__((aligned(64))) PackDesc {
    i64 acc_off, rec, row_abs, left0, ixr, itv_base;
    i32 plen0, count, channel, len_he_src;       // len | he << 20 | src << 21 (a row is shorter than 10^6 samples)
};
__global__ void k_pack_desc(ZleArgs a)
{
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_active_rows) return;
    const RowDesc q = a.desc[r];
    PackDesc p;
    p.acc_off = q.acc_off; p.rec = a.rec_off[q.idx]; p.row_abs = q.row_abs; p.ixr = q.ixr; p.itv_base = q.itv_base;
    p.count = a.itv_n[q.idx]; p.channel = q.channel; p.len_he_src = q.len | (q.he << 20) | (q.src << 21);
    p.left0 = 0; p.plen0 = 0;
    if (p.count > 0) { p.left0 = a.itv_left[q.itv_base]; p.plen0 = (i32)(a.itv_right[q.itv_base] - p.left0 + 1); }
    a.pdesc[r] = p;
}

// The records of the resident rows: their finished 16-bit samples exist (k_row_pulse), two of them are a dword of a record.  A kernel
// of its own with the few arguments it needs: in the general k_pack the scalar unit (one per CU) was the bottleneck -- ~350 scalar
// instructions per row for address arithmetic, spilled descriptors and the branches around four unrolled records.  Here the loads of
// up to four records are unconditional (indices clamped into the row), the header dwords are selected, not branched to, and the
// stores are predicated per lane.
struct PackResArgs { const PackDesc *pdesc; const int16_t *fin; const i64 *itv_left, *itv_right; uint8_t *records; const u32 *rec_dest; i64 rec_capacity, n_rows; i32 spr, dt; };
__global__ __launch_bounds__(256) void k_pack_res(PackResArgs a)
{
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * 4 + wave_in_block();
    if (r >= a.n_rows) return;
    const PackDesc q = a.pdesc[r];
    if (q.count == 0) return;
    const int16_t *fin = a.fin + q.acc_off;
    const i32 last = (((q.len_he_src & 0xfffff) + 3) & ~3) - 2;
    constexpr int spr = WFS_SPR; constexpr int rec_dwords = (24 + 2 * spr) / 4;
    const i64 rec_bytes = 24 + 2 * (i64)spr;
    const u32 w3 = ((u32)(uint16_t)a.dt) | ((u32)(uint16_t)q.channel << 16);
    i64 rec = q.rec;
    for (i32 k = 0; k < q.count; k++) {
        const i64 left = k == 0 ? q.left0 : a.itv_left[q.itv_base + k]; const i32 plen = k == 0 ? q.plen0 : (i32)(a.itv_right[q.itv_base + k] - left + 1);
        if (plen <= 0) continue;
        const i32 need = (plen + spr - 1) / spr;
        const i32 off = (i32)(left - q.row_abs);                   // even (rawdata.py:305-306), and the row starts on a multiple of four samples of fin
        for (int q0 = 0; q0 < rec_dwords; q0 += 64) {              // (one pass unless a record is longer than 64 dwords)
            const int qd = q0 + lane;
            const int s0 = (qd - 6) * 2;                            // first of the lane's two samples (lanes of the header: negative)
            const i32 lane_off = off + (s0 < 0 ? 0 : s0);
            for (i32 f0 = 0; f0 < need; f0 += 4) {
                u32 two[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { i32 i0 = lane_off + spr * (f0 + u); i0 = i0 > last ? last : i0; two[u] = *(const u32 *)(fin + i0); }      // (past the row: a valid pair, dropped)
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const i32 f = f0 + u;
                    const i64 slot = a.rec_dest ? (i64)a.rec_dest[rec + (f < need ? f : 0)] : rec + f;
                    u32 *out = (u32 *)(a.records + slot * rec_bytes);
                    const i64 time = (i64)a.dt * (left + spr * f);
                    const i32 rest = plen - spr * f, length = rest < spr ? rest : spr;
                    u32 w = (s0 < length ? two[u] & 0xffffu : 0u) | (s0 + 1 < length ? two[u] & 0xffff0000u : 0u);
                    w = qd == 0 ? (u32)(u64)time : w; w = qd == 1 ? (u32)((u64)time >> 32) : w; w = qd == 2 ? (u32)length : w;
                    w = qd == 3 ? w3 : w; w = qd == 4 ? (u32)plen : w; w = qd == 5 ? (u32)(uint16_t)f : w;          // record_i; baseline = 0
                    if (f < need && qd < rec_dwords && rec + f < a.rec_capacity) out[qd] = w;
                }
            }
        }
        rec += need;
    }
}

// one wave per row: write its intervals as strax raw_records (strax_interface.py:425-435).  A record is 24 bytes of header and
// samples_per_record int16: lane q writes dword q of the record (61 dwords with 110 samples: three idle lanes, no division by the
// record length anywhere), PACK_U records in flight so that the loads of one do not wait behind the stores of the one before.
template <int NK>
__global__ __launch_bounds__(256) void k_pack(WfsDev d, ZleArgs a)
{
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * 4 + wave_in_block();
    if (r >= a.n_front) return;                             // (the rows behind n_front: k_pack_res)
    const PackDesc q = a.pdesc[r];
    const i32 count = q.count;
    if (count == 0) return;
    const i32 channel = q.channel; const bool he = ((q.len_he_src >> 20) & 1) != 0; const int src = q.len_he_src >> 21;
    const i32 *acc = (src ? a.tbuf : a.raw) + q.acc_off;
    const i64 row_abs = q.row_abs;
    const i64 ixr = q.ixr;
    const i64 base = q.itv_base;
    const i32 len32 = q.len_he_src & 0xfffff;
    constexpr int spr = WFS_SPR; constexpr int rec_dwords = (24 + 2 * spr) / 4;
    const i64 rec_bytes = 24 + 2 * (i64)spr;
    const bool noisy = NK != 0 && channel < d.noise_channels;
    const bool fast_loads = NK == 0 || (d.noise_len >= NOISE_MIN_FAST && spr <= NOISE_MIN_FAST / 2);       // (as in k_zle: a scalar noise start per record)
    const i64 noise_off = noisy ? (i64)channel * d.noise_stride : 0; const u32 nixr = noisy ? (u32)ixr : 0u;
    const void *noise_row = NK == 2 ? (const void *)(d.noise_f + noise_off) : (const void *)(d.noise + noise_off);
    const u32 w3 = ((u32)(uint16_t)d.dt) | ((u32)(uint16_t)channel << 16);
    constexpr int PACK_U = 4;
    i64 rec = q.rec;
    for (i32 k = 0; k < count; k++) {
        const i64 left = k == 0 ? q.left0 : a.itv_left[base + k]; const i32 plen = k == 0 ? q.plen0 : (i32)(a.itv_right[base + k] - left + 1);
        if (plen <= 0) continue;
        const i32 need = (plen + spr - 1) / spr;
        const i32 off = (i32)(left - row_abs);                     // first sample of the interval inside the row
        for (int q0 = 0; q0 < rec_dwords; q0 += 64) {              // (one pass unless a record is longer than 64 dwords)
            const int qd = q0 + lane;
            const int s0 = (qd - 6) * 2;                            // first of the lane's two samples (lanes of the header: negative)
            u32 nrun = 0;                                           // noise index of the next record's first sample (a scalar)
            if constexpr (NK != 0) { if (fast_loads) nrun = (nixr + (u32)off) % (u32)d.noise_len; }
            const u32 noff = (u32)(s0 < 0 ? 0 : s0);
            for (i32 f0 = 0; f0 < need; f0 += PACK_U) {
                RawSample<NK> lo[PACK_U], hi[PACK_U];
                if (fast_loads) {
#pragma unroll
                    for (int u = 0; u < PACK_U; u++) {
                        const i32 f = f0 + u;
                        if (f >= need) break;                       // wave-uniform
                        i32 i0 = off + spr * f + (i32)noff;
                        i0 = i0 > len32 - 1 ? len32 - 1 : i0;       // (lanes past the record's samples read a valid sample and drop it)
                        const i32 i1 = i0 + 1 > len32 - 1 ? len32 - 1 : i0 + 1;
                        lo[u] = load_sample<NK>(d, acc, i0, noise_row, nrun, noff); hi[u] = load_sample<NK>(d, acc, i1, noise_row, nrun, noff + 1u);
                        if constexpr (NK != 0) { nrun += (u32)spr; nrun = nrun >= (u32)d.noise_len ? nrun - (u32)d.noise_len : nrun; }
                    }
                }
#pragma unroll
                for (int u = 0; u < PACK_U; u++) {
                    const i32 f = f0 + u;
                    if (f >= need) break;
                    if (rec + f >= a.rec_capacity) break;
                    u32 *out = (u32 *)(a.records + (a.rec_dest ? (i64)a.rec_dest[rec + f] : rec + f) * rec_bytes);
                    const i64 time = (i64)d.dt * (left + spr * f);
                    const i32 length = (plen < spr * (f + 1) ? plen : spr * (f + 1)) - spr * f;
                    u32 w;
                    if (qd >= 6) {
                        u32 vlo, vhi;
                        if (fast_loads) { vlo = (u32)(uint16_t)finish_loaded(d, lo[u], he, noisy); vhi = (u32)(uint16_t)finish_loaded(d, hi[u], he, noisy); }
                        else {
                            const i64 i0 = (i64)off + spr * f + s0;
                            vlo = s0 < length ? (u32)(uint16_t)finish_sample(d, acc, i0, channel, he, ixr) : 0u;
                            vhi = s0 + 1 < length ? (u32)(uint16_t)finish_sample(d, acc, i0 + 1, channel, he, ixr) : 0u;
                        }
                        w = (s0 < length ? vlo : 0u) | ((s0 + 1 < length ? vhi : 0u) << 16);
                    }
                    else if (qd == 0) w = (u32)(u64)time;
                    else if (qd == 1) w = (u32)((u64)time >> 32);
                    else if (qd == 2) w = (u32)length;
                    else if (qd == 3) w = w3;
                    else if (qd == 4) w = (u32)plen;
                    else w = (u32)(uint16_t)f;                        // record_i, baseline = 0
                    if (qd < rec_dwords) out[qd] = w;
                }
            }
        }
        rec += need;
    }
}

// ------------------------------------------------------------------------------------------------ generation
// One 64-byte descriptor per photon block (k_block_emitters): a generator workgroup starts from ONE scalar load instead of a
// chain of dependent look-ups (block -> emitters -> instruction -> pulse set ...), each a global round trip.
struct __attribute__((aligned(64))) BlockDesc {
    i64 e_lo;             // first emitter of the block
    i64 itime;            // origin of the photon times of the instruction's pulse set
    i64 R0;               // index of the block's first photon among its instruction's photons
    i32 ins;              // instruction of a single-instruction block, -1: the block spans instructions (generic path)
    i32 nwin;             // emitters of the block + 2 (<= GEN_WIN)
    i32 set, row;         // pulse set, channel CDF row
    u32 gid, eb, jbase;   // Philox coordinates: instruction id, emitter base, emitter id of e_lo
    i32 is_s2;
    u32 sbase;            // photons of the instructions in front of this one in its pulse set (order keys count through the set)
    i32 multi;            // the pulse set holds several instructions: order keys are written for every photon (the tile is sorted as a whole)
};

struct GenArgs {
    i64 n_ins, n_psets, n_emitters, n_photons;
    const u32 *ins_embase;        // [n_ins] offset of the instruction's emitter ids in the Philox counters (0 for primaries; the k-th
                                  // electron-afterpulse instruction of a parent shares the parent's gid and uses (k + 1) << 20)
    const i32 *ins_set;           // [n_ins] pulse set (one Pulse.__call__, rawdata.py:108-127) of every instruction; tiles are (set, channel)
    const i64 *set_ins_off; const i32 *set_ins_list;      // set -> its instructions (CSR)
    const i64 *set_t0;            // [n_sets] time origin of the set's photon times
    const int8_t *ins_type; const i64 *ins_time; const i32 *ins_amp; const u32 *ins_gid;
    const double *ins_p, *ins_dm, *ins_ds, *ins_sc; const i32 *ins_cdfrow; const double *cdf_table;
    const unsigned short *cdf_guide;      // [n_cdf][CDF_G + 2] guide table of every CDF row (host; no longer used by the generator)
    const uint2 *chan_alias; i32 ch_lg;   // [n_cdf][1 << ch_lg] Walker alias cells of every channel CDF row (k_chan_alias): {threshold, alias channel}
    const i64 *em_off;            // [n_ins + 1] first emitter of each instruction
    i64 *em_time; i32 *em_nph; i32 *em_ins; const i64 *em_ph_off;
    double *em_zg;                // [n_emitters] gain-spread normal of a surviving electron whose photon number k_s2_photons draws (PTRS), NaN elsewhere
    const double *pois_cdf; const i32 *pois_kmin;     // [n_ins][POIS_W] cumulative Poisson table of the instruction's secondary gain, first k (-1: PTRS, -2: gain <= 0)
    i32 *tile_count; const i64 *tile_off; i32 *tile_cursor; i32 *tile_tmin, *tile_tmax;
    PhotonRec *ph;
    u32 *ph_idx;                  // [photons] order key of every stored photon: its index among its pulse set's photons, instruction by
                                  // instruction (afterpulses: element << 29 | key of the parent) -- k_tile_order sorts every tile by it
    u32 *ins_sbase;               // [n_ins] photons of the instructions in front of this one in its pulse set (k_set_bases)
    i32 *tile_tail;               // [primary tiles] photons that the generic path puts BEHIND the tile's block ranges: those of an instruction that
                                  // began in an earlier block (count pass; the fill pass uses it as their cursor)
    i32 *tile_tailbase;           // [primary tiles] first slot of those photons inside the tile (k_block_ranges)
    i32 *ins_fullsort;            // [n_ins] 1: the tiles of this instruction are sorted as a whole (several instructions in its pulse set, or a block
                                  // in the MIDDLE of the instruction that takes the generic path: more emitters than the LDS window holds)
    i64 n_blocks;
    i64 xcd_chunk;                // XCD x (workgroup id % 8) walks the photon blocks [x * xcd_chunk, (x + 1) * xcd_chunk) in order
    i32 *eblk_ins;                // [ceil(n_emitters / 256) + 1] instruction of the first emitter of every block of k_s2_electrons
    i64 *blk_e;                   // [n_blocks][2] first / last emitter of every photon block
    u32 *blk_base;                // [n_blocks][n_tpc] start of the block's photons inside each tile (written by the count pass)
    unsigned short *blk_cnt;      // [n_blocks][n_tpc] the block's photons per channel (count pass, single-instruction blocks)
    i32 *blk_ins;                 // [n_blocks] instruction of a single-instruction block, -1 otherwise (k_block_emitters)
    struct BlockDesc *blk_desc;   // [n_blocks] everything a generator workgroup needs to know about its block
    const i32 *ins_fused;         // [n_ins] 1: the instruction's photons are generated tile by tile (wfs_tilegen.h), or nullptr
    i64 *ins_ph0;                 // [n_ins + 1] first photon (generation order) of every instruction: em_ph_off[em_off[i]]
    double *el_stat;              // [n_ins][4] electrons: n, sum t, sum t^2 ; el_minmax [n_ins][2]
    i64 *el_minmax;
    i64 *scal;
    // model variants of the photon delays (wfs_set_delay_models / wfs_set_instruction_models); tabs == nullptr: the two default tables
    const AliasTab *tabs;         // [n_tables + 2] user tables, then the default S1 and S2 tables
    const i32 *ins_tab, *ins_tabb;        // [n_ins] table of the instruction's photons on top / bottom array channels
    const i32 *ins_pzi; const double *ins_pzf;    // [n_ins] S1 optical propagation: z cell of the spline grid and normalised distance in it (-1: none)
    const double *prop_top, *prop_bot; i32 prop_nu; double prop_u0, prop_du;      // spline node values [nz][nu], u grid
    // 'garfield_gas_gap' luminescence (s2.py:413-483): inverse CDFs of the excitation time per tabulated gas gap [gg_n][gg_L];
    // per instruction the table at or below its gas gap (-1: not this model), the interpolation weight towards the next
    // table, and the sum of its photons' excitation times in units of 2^-20 ns (k_gg_sum; the mean is subtracted, s2.py:447)
    const double *gg_inv; i32 gg_n, gg_L; const i32 *ins_gg; const double *ins_ggw; i64 *ins_ggsum;
};

// S1 optical propagation delay (s1.py:241-260): multilinear interpolation of the spline nodes, evaluated as scipy's
// RegularGridInterpolator does (sum over the cell's corners of value * weights); same operation order as the oracle
__device__ __forceinline__ double s1_propagation(const GenArgs &a, bool bottom, i32 zi, double zf, u32 w)
{
    const double u = ((double)w + 0.5) * (1.0 / 4294967296.0);
    const double *T = bottom ? a.prop_bot : a.prop_top; const i32 nu = a.prop_nu;
    i32 ui = (i32)floor((u - a.prop_u0) / a.prop_du);
    ui = ui < 0 ? 0 : (ui > nu - 2 ? nu - 2 : ui);
    const double uf = (u - (a.prop_u0 + (double)ui * a.prop_du)) / a.prop_du;
    const double *r0 = T + (i64)zi * nu + ui, *r1 = r0 + nu;
    double v = r0[0] * ((1.0 - zf) * (1.0 - uf));
    v += r0[1] * ((1.0 - zf) * uf);
    v += r1[0] * (zf * (1.0 - uf));
    v += r1[1] * (zf * uf);
    return v;
}

// excitation time of one photon from the garfield gas gap tables (s2.py:436-446): a uniform position on the inverse CDF,
// interpolated between the two tables that bracket the instruction's gas gap; same operation order as the oracle
__device__ __forceinline__ double gg_time(const GenArgs &a, i32 lo, double wgt, u32 w)
{
    const i32 L = a.gg_L, hi = lo + 1 < a.gg_n ? lo + 1 : a.gg_n - 1;
    const double samples = ((double)w + 0.5) * (1.0 / 4294967296.0) * (double)(L - 2);
    const double fl = floor(samples); const i32 i0 = (i32)fl, i1 = (i32)ceil(samples);
    const double *A = a.gg_inv + (i64)lo * L, *B = a.gg_inv + (i64)hi * L;
    const double t1 = (B[i0] - A[i0]) * wgt + A[i0], t2 = (B[i1] - A[i1]) * wgt + A[i1];
    return (t2 - t1) * (samples - fl) + t1;
}
// mean excitation time of an instruction's photons (fixed point sum -> independent of the order of summation)
__device__ __forceinline__ double gg_mean(const GenArgs &a, i32 ins)
{
    const i64 n = a.ins_ph0[ins + 1] - a.ins_ph0[ins];
    return n > 0 ? (double)a.ins_ggsum[ins] / (double)n / 1048576.0 : 0.0;
}

// Poisson: PTRS (Hoermann 1993) for lam >= 10, multiplication method below; same algorithm and uniforms as the oracle
__device__ i64 poisson_draw(const WfsDev &d, u32 emitter, u32 gid, double lam)
{
    u32 it = 0;
    if (lam <= 0) return 0;
    if (lam < 10) {
        double enlam = exp(-lam), prod = 1.0; i64 x = 0;
        for (;;) {
            u32x4 w = philox4x32_10(emitter, gid, it++, SITE_EL_POIS, d.k0, d.k1);
            prod *= u53(w.x, w.y);
            if (prod > enlam) x++; else return x;
            prod *= u53(w.z, w.w);
            if (prod > enlam) x++; else return x;
        }
    }
    double slam = sqrt(lam), loglam = log(lam);
    double b = 0.931 + 2.53 * slam, aa = -0.059 + 0.02483 * b;
    double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2);
    for (;;) {
        u32x4 w = philox4x32_10(emitter, gid, it++, SITE_EL_POIS, d.k0, d.k1);
        double U = u53(w.x, w.y) - 0.5, V = u53(w.z, w.w);
        double us = 0.5 - fabs(U);
        i64 k = (i64)floor((2 * aa / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(aa / (us * us) + b) <= -lam + k * loglam - lgamma((double)k + 1)) return k;
    }
}

// S1: n_hits = Binomial(amp, ly) as a sum of Bernoulli trials (s1.py:133), one wave per instruction
__global__ __launch_bounds__(256) void k_s1_hits(WfsDev d, GenArgs a)
{
    const int lane = threadIdx.x & 63;
    i64 i = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= a.n_ins || a.ins_type[i] != 1) return;
    const u64 T = bern_threshold(a.ins_p[i]);
    const i64 amp = a.ins_amp[i]; const u32 gid = a.ins_gid[i];
    i32 hits = 0;
    for (i64 q = lane; q * 4 < amp; q += 64) {
        u32x4 w = philox4x32_10(0, gid, (u32)q, SITE_S1_HIT, d.k0, d.k1);
        i64 j = q * 4;
        hits += (u64)w.x < T;
        if (j + 1 < amp) hits += (u64)w.y < T;
        if (j + 2 < amp) hits += (u64)w.z < T;
        if (j + 3 < amp) hits += (u64)w.w < T;
    }
    for (int o = 32; o > 0; o >>= 1) hits += __shfl_down(hits, o, 64);
    if (lane == 0) { i64 e = a.em_off[i]; a.em_nph[e] = hits; a.em_time[e] = a.ins_time[i]; a.em_ins[e] = (i32)i; }
}

// Photons per electron, Poisson(sc_gain) (s2.py:308).  The secondary gain is a property of the INSTRUCTION, so its Poisson
// distribution is tabulated once per instruction -- pmf(k) = exp(k ln(lam) - lam - lgamma(k + 1)) on the 256 values from
// floor(lam - 8 sqrt(lam)) - 4 on (what lies outside is below 1e-15), summed in order, normalised -- and an electron inverts it with
// ONE uniform (bisection, 8 steps).  numpy's PTRS rejection loop with its logs and lgamma per trial cost 2000 VALU instructions per
// electron (every wave takes the slow path and the longest loop of its 64 lanes) and 198 VGPRs; it is kept, in a kernel of its own,
// for gains above POIS_LAM_MAX.  Same distribution (tests: KS against np.random.poisson draws of the reference).
#define POIS_W 256
#define POIS_LAM_MAX 217.0
__global__ __launch_bounds__(POIS_W) void k_poisson_tables(GenArgs a, double *cdf, i32 *kmin)
{
    __shared__ double p[POIS_W];
    const i64 i = blockIdx.x; const int t = threadIdx.x;
    const double lam = a.ins_sc[i];
    if (a.ins_type[i] == 1 || !(lam > 0) || lam > POIS_LAM_MAX) { if (t == 0) kmin[i] = (a.ins_type[i] != 1 && !(lam > 0)) ? -2 : -1; return; }
    i64 k0 = (i64)floor(lam - 8.0 * sqrt(lam)) - 4; if (k0 < 0) k0 = 0;
    const double k = (double)(k0 + t);
    p[t] = exp(k * log(lam) - lam - lgamma(k + 1.0));
    __syncthreads();
    if (t == 0) { double run = 0; for (int j = 0; j < POIS_W; j++) { run += p[j]; p[j] = run; } kmin[i] = (i32)k0; }
    __syncthreads();
    cdf[i * POIS_W + t] = p[t] / p[POIS_W - 1];
}
__device__ __forceinline__ i64 poisson_table_draw(const WfsDev &d, const double *cdf, i32 kmin, u32 emitter, u32 gid)
{
    const u32x4 w = philox4x32_10(emitter, gid, 0, SITE_EL_POIS, d.k0, d.k1);
    const double u = u53(w.x, w.y);
    int lo = 0, hi = POIS_W - 1;                         // first j with u < cdf[j] (cdf[POIS_W - 1] == 1)
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
    return (i64)kmin + lo;
}

// S2: one thread per candidate electron (s2.py:254, 258-286, 308-310)
// instruction of the first emitter of every 256-emitter block of k_s2_electrons (+ a sentinel: the last instruction)
__global__ void k_emitter_blocks(GenArgs a, i64 n_blocks)
{
    const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_blocks) return;
    if (b == n_blocks) { a.eblk_ins[b] = (i32)(a.n_ins - 1); return; }
    const i64 e = b * 256;
    i64 lo = 0, hi = a.n_ins;
    while (hi - lo > 1) { i64 mid = (lo + hi) >> 1; if (a.em_off[mid] <= e) lo = mid; else hi = mid; }
    a.eblk_ins[b] = (i32)lo;
}

__global__ __launch_bounds__(256) void k_s2_electrons(WfsDev d, GenArgs a)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    i64 i = -1; bool is_s2 = false;
    // instruction of emitter e: em_off[i] <= e < em_off[i+1].  The instruction of every block's first emitter comes from
    // k_emitter_blocks (all those searches in flight at once); inside the block the search runs over what is left,
    // usually nothing (10^4 candidate electrons per S2).
    if (e < a.n_emitters) {
        i64 lo = a.eblk_ins[blockIdx.x], hi = a.eblk_ins[blockIdx.x + 1] + 1;
        while (hi - lo > 1) { i64 mid = (lo + hi) >> 1; if (a.em_off[mid] <= e) lo = mid; else hi = mid; }
        i = lo; is_s2 = a.ins_type[i] != 1;          // types 2, 4, 6 are S2-like (electrons drifting to the gas gap)
    }
    double st_n = 0, st_t = 0, st_t2 = 0; i64 st_min = I64_MAX, st_max = I64_MIN;
    if (is_s2) {
        const u32 j0 = (u32)(e - a.em_off[i]), jb = a.ins_embase[i], j = jb + j0; const u32 gid = a.ins_gid[i];
        a.em_ins[e] = (i32)i;
        u32x4 sv = philox4x32_10(jb, gid, j0 >> 2, SITE_S2_SURVIVE, d.k0, d.k1);
        u32 word = (j0 & 3) == 0 ? sv.x : (j0 & 3) == 1 ? sv.y : (j0 & 3) == 2 ? sv.z : sv.w;
        if (!((u64)word < bern_threshold(a.ins_p[i]))) { a.em_nph[e] = 0; a.em_time[e] = I64_MIN; }
        else {
            u32x4 A = philox4x32_10(j, gid, 0, SITE_EL_A, d.k0, d.k1);
            u32x4 B = philox4x32_10(j, gid, 0, SITE_EL_B, d.k0, d.k1);
            double z_drift, z_gain;
            box_muller(B, z_drift, z_gain);
            double timing = -log(1.0 - u53(A.x, A.y)) * d.trap_time;
            timing += a.ins_dm[i] + a.ins_ds[i] * z_drift;
            i64 et = a.ins_time[i] + (i64)timing;                         // s2.py:282
            a.em_time[e] = et;
            const i32 pk = a.pois_kmin[i];
            if (a.ins_fused && a.ins_fused[i]) a.em_nph[e] = 0;                // its photons are made tile by tile (wfs_tilegen.h)
            else if (pk == -1) a.em_zg[e] = z_gain;                             // gain above POIS_LAM_MAX: k_s2_photons (PTRS)
            else {
                i64 nph = pk >= 0 ? poisson_table_draw(d, a.pois_cdf + i * POIS_W, pk, j, gid) : 0;
                nph += (i64)(0.0 + d.gain_spread * z_gain);               // s2.py:309
                if (nph < 0) nph = 0;
                a.em_nph[e] = (i32)nph;
            }
            double tr = (double)(et - a.ins_time[i]);
            st_n = 1; st_t = tr; st_t2 = tr * tr; st_min = et; st_max = et;
        }
    }
    // electron time statistics of the truth row (rawdata.py:325-332): one set of atomics per wave when the
    // whole wave works on one instruction, else one per lane
    const i64 i_first = __shfl(i, 0, 64);
    bool leader = true;
    if (all64(i == i_first)) {
        for (int o = 32; o > 0; o >>= 1) {
            st_n += __shfl_down(st_n, o, 64); st_t += __shfl_down(st_t, o, 64); st_t2 += __shfl_down(st_t2, o, 64);
            i64 mn = __shfl_down(st_min, o, 64), mx = __shfl_down(st_max, o, 64);
            st_min = mn < st_min ? mn : st_min; st_max = mx > st_max ? mx : st_max;
        }
        leader = (threadIdx.x & 63) == 0;
    }
    if (leader && st_n > 0) {
        atomicAdd(&a.el_stat[i * 4 + 0], st_n); atomicAdd(&a.el_stat[i * 4 + 1], st_t); atomicAdd(&a.el_stat[i * 4 + 2], st_t2);
        atomicMin(&a.el_minmax[i * 2], st_min); atomicMax(&a.el_minmax[i * 2 + 1], st_max);
    }
}

// photons of the surviving electrons of instructions with a gain above POIS_LAM_MAX: numpy's PTRS + int(N(0, gain_spread)) (s2.py:308-310)
__global__ __launch_bounds__(256) void k_s2_photons(WfsDev d, GenArgs a)
{
    const i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_emitters) return;
    if (a.em_time[e] == I64_MIN) return;                     // did not survive (em_nph = 0 already); S1 emitters are not touched: em_zg is NaN
    const double zg = a.em_zg[e];
    if (!(zg == zg)) return;
    const i32 i = a.em_ins[e];
    const u32 j = a.ins_embase[i] + (u32)(e - a.em_off[i]);
    i64 nph = poisson_draw(d, j, a.ins_gid[i], a.ins_sc[i]);
    nph += (i64)(0.0 + d.gain_spread * zg);                  // s2.py:309
    if (nph < 0) nph = 0;
    a.em_nph[e] = (i32)nph;
}

__device__ __forceinline__ int channel_from_cdf(const double *cdf, int n, double u)
{
    int lo = 0, hi = n;                             // searchsorted(cdf, u, side='right')
    while (lo < hi) { int mid = (lo + hi) >> 1; if (u < cdf[mid]) hi = mid; else lo = mid + 1; }
    return lo < n ? lo : n - 1;
}

#ifndef GEN_LOG
#define GEN_LOG 11
#endif
#define GEN_BLOCK (1 << GEN_LOG)   // photons per block of the generator
#define GEN_WIN 512                // emitter offsets staged in LDS per block
#define CDF_G 512                  // guide cells of the per-block channel search

// Guide of a non-decreasing CDF row: cell j = int(u * scale), clamped to AP_GUIDE - 1, holds a bracket [lo, hi] of "the first entry >= u"
// for every u of the cell (built on the host from the row itself with a margin on either side of the cell, wfs_set_ap_element): the
// bisection of argmin_abs_diff starts from two entries instead of the row -- ~8 dependent, divergent loads per search fewer.
#define AP_GUIDE 256
struct ApGuide { double scale; unsigned short lo[AP_GUIDE], hi[AP_GUIDE]; };

// PMT afterpulse element tables (afterpulse.py:181-186) and the staging list of generated afterpulse photons
struct ApElemDev { i32 n_bins_delay, n_bins_amp, amp_2d, is_uniform; double delay_bin, amp_bin; const double *delay_cdf, *amp_cdf;
                   i32 delay_sorted, amp_sorted;        // the rows are non-decreasing (checked on the host): argmin by bisection
                   const ApGuide *delay_guide, *amp_guide; };     // per row of the two tables (or nullptr): where the bisection starts

// np.argmin(|cdf - u|) (afterpulse.py:222, 229): the FIRST index of the smallest distance.  On a non-decreasing row the
// distance falls up to the first entry >= u and rises behind it, so the answer is that entry or the one in front of it
// (lower index on a tie), moved to the start of its plateau of equal values -- two bisections instead of a scan of the row
// (a scan costs every wave that holds one accepted photon ~n_bins iterations: 20 of 23 ms of the fill pass with
// afterpulses on).  Rows that are not sorted (never seen; the host checks) keep the scan.
__device__ __forceinline__ int argmin_abs_diff(const double *c, int n, double u, bool sorted, const ApGuide *guide = nullptr)
{
    if (!sorted) {
        int best = 0; double bd = fabs(c[0] - u);
        for (int k = 1; k < n; k++) { const double dd = fabs(c[k] - u); if (dd < bd) { bd = dd; best = k; } }
        return best;
    }
    int lo = 0, hi = n;                                  // first k with c[k] >= u (n: none)
    if (guide && u >= 0.0) {
        const double x = u * guide->scale;
        const int j = x < (double)(AP_GUIDE - 1) ? (int)x : AP_GUIDE - 1;
        lo = guide->lo[j]; hi = guide->hi[j];
    }
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (c[mid] >= u) hi = mid; else lo = mid + 1; }
    int best = lo;
    if (lo == n) best = n - 1;
    else if (lo > 0 && !(fabs(c[lo] - u) < fabs(c[lo - 1] - u))) best = lo - 1;
    const double v = c[best];
    if (best == 0 || c[best - 1] < v) return best;       // (no plateau in front of it: the usual case, one load instead of a second bisection)
    int a = 0, b = best;                                 // first k with c[k] == v: first k with c[k] >= v
    while (a < b) { const int mid = (a + b) >> 1; if (c[mid] >= v) b = mid; else a = mid + 1; }
    return a;
}
struct ApArgs {
    i32 n; i32 pad;
    ApElemDev el[WFS_MAX_AP];
    const double *prob[WFS_MAX_AP]; // [n_tpc] per element: afterpulse probability of every channel (last entry of its delay CDF row), contiguous
    const u32 *thr[WFS_MAX_AP];     // [n_tpc][2] per element: screening thresholds (single / double photoelectron parent), ap_threshold
    i64 cap;                        // capacity of the staging list
    i32 *ap_ins; i32 *ap_ch; i32 *ap_t; double *ap_gain;     // [cap] instruction, channel, ns relative to the instruction, gain
    u32 *ap_key;                    // [cap] order key: element << 29 | index of the parent photon among its instruction's photons
    i64 *count;                     // number of afterpulse candidates in the list (device scalar, scal[13])
    struct ApSeg *seg; i64 n_seg;   // candidates of a k_s2_tile workgroup: one contiguous, key-ordered stretch of the list per tile (k_ap_seg)
    struct ApCand *cand;            // [cap] candidates of the generator (k_ap_finish turns entry i into afterpulse photon i, or a hole: ap_ch[i] = -1)
};
#define AP_STAGE 128               // afterpulse candidates a block parks in LDS (48 bytes each; a block of 2048 photons has ~60)

// first and last emitter of every photon block: one bisection per thread, all in flight together (a block doing
// its own two bisections serially costs ~20 us of dependent HBM latency before its 2048 photons can start)
__global__ void k_block_emitters(GenArgs a)
{
    const i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.n_blocks) return;
    const i64 p0 = b * GEN_BLOCK;
    const i64 p1 = (p0 + GEN_BLOCK < a.n_photons) ? p0 + GEN_BLOCK : a.n_photons;
    for (int w = 0; w < 2; w++) {
        const i64 p = w == 0 ? p0 : p1 - 1;
        i64 lo = 0, hi = a.n_emitters;
        while (hi - lo > 1) { i64 mid = (lo + hi) >> 1; if (a.em_ph_off[mid] <= p) lo = mid; else hi = mid; }
        a.blk_e[2 * b + w] = lo;
    }
    // the fast path of the generator needs all photons of the block in one instruction and the block's emitters in its LDS window
    const i64 e_lo = a.blk_e[2 * b], e_hi = a.blk_e[2 * b + 1];
    const i32 i0 = a.em_ins[e_lo];
    const bool single = e_hi - e_lo + 2 <= GEN_WIN && i0 == a.em_ins[e_hi];
    a.blk_ins[b] = single ? i0 : -1;
    if (!single && i0 == a.em_ins[e_hi] && a.ins_ph0[i0] < p0 && a.ins_ph0[i0 + 1] > p1) a.ins_fullsort[i0] = 1;      // generic block inside one instruction
    BlockDesc bd{};
    bd.ins = -1;
    if (single) {
        bd.ins = i0; bd.e_lo = e_lo; bd.nwin = (i32)(e_hi - e_lo + 2); bd.set = a.ins_set[i0]; bd.itime = a.set_t0[bd.set];
        bd.R0 = p0 - a.ins_ph0[i0]; bd.row = a.ins_cdfrow[i0]; bd.gid = a.ins_gid[i0]; bd.eb = a.ins_embase[i0];
        bd.jbase = (u32)(e_lo - a.em_off[i0]) + bd.eb; bd.is_s2 = a.ins_type[i0] != 1; bd.sbase = a.ins_sbase[i0];
        bd.multi = 0;      // (whether the tile is sorted as a whole is only known when every block has been looked at: ins_fullsort, read by the fill pass)
    }
    a.blk_desc[b] = bd;
}

__global__ void k_ins_ph0(GenArgs a)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= a.n_ins) a.ins_ph0[i] = a.em_ph_off[a.em_off[i]];
}
// order keys run through a pulse set instruction by instruction (the reference appends the photons of a Pulse call's instructions
// in that order, s1.py:95-103 / s2.py:107-136): photons in front of every instruction inside its set
__global__ void k_set_bases(GenArgs a)
{
    const i64 s = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_psets) return;
    i64 run = 0;
    for (i64 q = a.set_ins_off[s]; q < a.set_ins_off[s + 1]; q++) {
        const i32 i = a.set_ins_list[q];
        if (run > 0xffffffffLL) atomicMax(&a.scal[1], (i64)2);
        a.ins_sbase[i] = (u32)run;
        a.ins_fullsort[i] = a.set_ins_off[s + 1] - a.set_ins_off[s] > 1;       // (k_block_emitters adds the instructions with a generic block in their middle)
        run += a.em_ph_off[a.em_off[i + 1]] - a.em_ph_off[a.em_off[i]];
    }
}

__device__ __forceinline__ u32 word_of(const u32x4 &W, int k) { return k == 0 ? W.x : (k == 1 ? W.y : (k == 2 ? W.z : W.w)); }

// PMT afterpulses of one photon (afterpulse.py:172-249): one uniform pair per element.  RNG spec v10: the top 32 bits of the FIRST
// uniform of elements 4k .. 4k + 3 are the four words of one call (site SITE_AP_SCREEN + k); its low bits and the second uniform come
// from the element's own call (SITE_AP + e: words y, z, w), which only a candidate ever makes.  The generator only SCREENS
// (ap_generate, inside the photon loop): one Philox call per four elements and an integer comparison of the screen word with a
// threshold of the (element, channel) that no accepted uniform can miss (ap_threshold in wfs_engine.hip); the rare candidate is
// parked in LDS and leaves with the block's other candidates for the global list.  k_ap_finish, one thread per candidate, makes
// the reference's own floating-point comparison and turns the accepted ones into afterpulse photons (two bisections of cumulative
// rows in global memory, ~20 dependent loads).  Inside the photon loop a wave paid that chain whenever ONE of its 64 lanes had a
// candidate, which with p ~ 1 % per element is most of the time; behind the loop, once per block, it still kept the block's LDS
// and its eight mostly idle waves on the CU for tens of microseconds (fill pass 5.1 -> 2.7 ms -> see DESIGN.md for the last step).
struct __attribute__((aligned(8))) ApCand { i64 itime; u32 x; i32 ins, ch, t; u32 key, j, m, gid; i32 e_dpe, pad; };     // (48 bytes)
struct ApStage { i32 *n; ApCand *cand; };
struct ApSeg { i64 base; i32 n, tile; };     // list entries [base, base + n) belong to afterpulse tile `tile`
#define AP_SEG_MARK (1 << 30)                // in ApCand.ins / ap_ins: the entry belongs to a segment (k_ap_seg places it, k_ap_count / k_ap_place skip it)
// the uniform of the acceptance test, exactly as the reference forms and scales it (afterpulse.py:196-204); w: the element's own call
__device__ __forceinline__ double ap_uniform(const WfsDev &d, const ApCand &q, const u32x4 &w)
{
    double rU0 = 1.0 - u53(q.x, w.y);
    rU0 /= d.pmt_ap_modifier;
    if (q.e_dpe >> 8) rU0 /= 2;
    return rU0;
}
__device__ __forceinline__ u32x4 ap_call(const WfsDev &d, const ApCand &q) { return philox4x32_10(q.j, q.gid, q.m, SITE_AP + (u32)(q.e_dpe & 0xff), d.k0, d.k1); }
__device__ __forceinline__ bool ap_accept(const WfsDev &d, const ApArgs &ap, const ApCand &q, const u32x4 &w) { return ap_uniform(d, q, w) <= ap.prob[q.e_dpe & 0xff][q.ch]; }
__device__ __forceinline__ void ap_finish(const WfsDev &d, i64 *scal, const ApArgs &ap, const ApCand &q, const u32x4 &w, i64 gk)
{
    const int e = q.e_dpe & 0xff;
    const ApElemDev &el = ap.el[e];
    const double *dc = el.delay_cdf + (size_t)q.ch * el.n_bins_delay;
    double delay, amp;
    if (el.is_uniform) {
        const u32x4 x = philox4x32_10(q.j, q.gid, q.m, SITE_AP_X + (u32)e, d.k0, d.k1);
        delay = (dc[0] + (dc[1] - dc[0]) * u53(x.x, x.y)) * el.delay_bin; amp = 1.0;
    } else {
        const int best = argmin_abs_diff(dc, el.n_bins_delay, ap_uniform(d, q, w), el.delay_sorted != 0, el.delay_guide ? el.delay_guide + q.ch : nullptr);      // np.argmin(|cdf - u|): first minimum
        delay = best * el.delay_bin - d.pmt_ap_t_modifier;
        const double *ac = el.amp_2d ? el.amp_cdf + (size_t)q.ch * el.n_bins_amp : el.amp_cdf;
        const int ba = argmin_abs_diff(ac, el.n_bins_amp, 1.0 - u53(w.z, w.w), el.amp_sorted != 0, el.amp_guide ? el.amp_guide + (el.amp_2d ? q.ch : 0) : nullptr);
        amp = ba * el.amp_bin;
    }
    const double tf = (double)(q.itime + q.t) + delay;          // afterpulse.py:235, int64 + float
    i64 tap = (i64)tf - q.itime;
    if (tap > 0x7fffffffLL || tap < -0x7fffffffLL) { atomicMax(&scal[1], (i64)2); tap = 0; }
    if (gk < ap.cap) { ap.ap_ins[gk] = q.ins; ap.ap_ch[gk] = q.ch; ap.ap_t[gk] = (i32)tap; ap.ap_gain[gk] = d.gains[q.ch] * amp; ap.ap_key[gk] = q.key; }
}
// a candidate into the block's LDS list (or, past AP_STAGE, straight to the end)
__device__ __forceinline__ void ap_park(const ApArgs &ap, const ApStage &st, u32 x, i32 e_dpe, u32 j, u32 gid, u32 m,
                                     i32 ins, i32 ch, i64 itime, i32 t, u32 key, bool seg = false)
{
    ApCand q;
    q.x = x; q.itime = itime; q.ins = ins; q.ch = ch; q.t = t; q.j = j; q.m = m; q.gid = gid; q.e_dpe = e_dpe; q.pad = 0;
    q.key = key;                                                    // element << 29 | parent: the reference walks element by element, parent by parent (afterpulse.py:189-207)
    const i32 kq = atomicAdd(st.n, 1);
    if (kq < AP_STAGE) { if (seg) q.ins |= AP_SEG_MARK; st.cand[kq] = q; }
    else { const i64 gk = (i64)atomicAdd((u64 *)ap.count, 1ull); if (gk < ap.cap) ap.cand[gk] = q; }      // (a block with more candidates than the stage holds)
}
// the screen alone: bit e set when element e of this photon is a candidate (k_s2_tile keeps eight photons in registers and parks the
// rare candidates in a loop of its own); ap_screen_word recomputes a candidate's screen word
__device__ __forceinline__ u32 ap_screen_mask(const WfsDev &d, const ApArgs &ap, u32 j, u32 gid, u32 m, int ch, bool is_dpe)
{
    u32 mask = 0;
    for (int e0 = 0; e0 < ap.n; e0 += 4) {
        const u32x4 S = philox4x32_10(j, gid, m, SITE_AP_SCREEN + (u32)(e0 >> 2), d.k0, d.k1);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = e0 + k;
            if (e >= ap.n) break;
            if ((word_of(S, k) >> 5) >= ap.thr[e][ch * 2 + (is_dpe ? 1 : 0)]) mask |= 1u << e;
        }
    }
    return mask;
}
__device__ __forceinline__ u32 ap_screen_word(const WfsDev &d, u32 j, u32 gid, u32 m, int e)
{
    return word_of(philox4x32_10(j, gid, m, SITE_AP_SCREEN + (u32)(e >> 2), d.k0, d.k1), e & 3);
}
__device__ __forceinline__ void ap_generate(const WfsDev &d, const ApArgs &ap, const ApStage &st,
                                            u32 j, u32 gid, u32 m, i32 ins, int ch, bool is_dpe, i64 itime, i64 t, u32 P)
{
    for (int e0 = 0; e0 < ap.n; e0 += 4) {
        const u32x4 S = philox4x32_10(j, gid, m, SITE_AP_SCREEN + (u32)(e0 >> 2), d.k0, d.k1);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = e0 + k;
            if (e >= ap.n) break;
            const u32 x = word_of(S, k);
            const u32 thr = ap.thr[e][ch * 2 + (is_dpe ? 1 : 0)];
            if ((x >> 5) < thr) continue;                           // surely rejected (ap_threshold)
            ap_park(ap, st, x, e | (is_dpe ? 256 : 0), j, gid, m, ins, ch, itime, (i32)t, ((u32)e << 29) | (P & 0x1fffffffu));
        }
    }
}

// Photon generator, two passes over the same photon index space (GEN_BLOCK consecutive photons per block, generation
// order = emitter by emitter).  RNG spec v6: photon P of an instruction (P = its index among the
// instruction's photons) owns word P & 3 of three calls (em_base, gid, P >> 2, site): SITE_CH -> channel (s1.py:154-158 /
// s2.py:673-677), SITE_DELAY -> summed delay (alias table; s1.py:180-194 / s2.py:504-557, pulse.py:53-56), SITE_GAIN ->
// SPE indices and double-PE flag (pulse.py:76-103).  A thread that takes four consecutive photons spends 0.75 Philox
// calls per photon (the generator is bound by VALU issue: 20 quarter-rate multiplies per call).
// Count pass (k_photon_count): channel words only -> photons per (block, channel); no per-photon intermediate is left in
//   HBM (the first version wrote a packed word per photon and read it back: 5.8 GB per 10^9-PE batch).
// k_block_ranges: where in its tile every block puts its photons (consecutive blocks, consecutive ranges).
// Fill pass (k_photon_fill): all three calls per quad in generation order; every photon is ranked inside its channel with
//   LDS counters and staged at its bucket position in LDS; the block then stores in BUCKET order: neighbouring lanes write
//   photons of the same tile to consecutive addresses (a scattered 8-byte store per lane is bound by the L2 request rate).
// Fast path ("single": all photons of the block belong to one instruction, the normal case for an S2): emitter window,
//   channel thresholds + guide table and tile offsets of the block live in LDS.
// Generic path (a block spanning instructions: S1s, small S2s): per-photon global lookups and atomics.

// exclusive prefix sum over n <= 4 * TPB LDS integers, in place; v[n] receives the total.  All TPB threads call it.
template <int TPB>
__device__ __forceinline__ void block_excl_scan(i32 *v, int n, i32 *wtmp)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int per = (n + TPB - 1) / TPB, b = tid * per;
    i32 loc[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { loc[k] = (k < per && b + k < n) ? v[b + k] : 0; s += loc[k]; }
    i32 x = s;
    for (int o = 1; o < 64; o <<= 1) { const i32 y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
    if (lane == 63) wtmp[wid] = x;
    __syncthreads();
    i32 run = x - s;
    for (int w = 0; w < wid; w++) run += wtmp[w];
#pragma unroll
    for (int k = 0; k < 4; k++) if (k < per && b + k < n) { v[b + k] = run; run += loc[k]; }
    if (tid == TPB - 1) v[n] = run;
    __syncthreads();
}

// ---- pattern maps on the device (S1.photon_channels s1.py:138-159, S2.photon_channels s2.py:616-682; the maps are
// straxen.InterpolatingMap objects built by load_resource.make_patternmap, method WeightedNearestNeighbors): the hit
// pattern at a position is the inverse-distance weighted average of the 2 * dims nearest grid nodes (distance clipped at
// 1e-6).  On the host that costs ~0.4 ms per instruction (10^4 S1: seconds); here one thread finds the neighbours of
// an instruction and one workgroup turns them into the instruction's channel CDF row and its guide table.
#define MAP_K 6                    // 2 * dims neighbours, dims <= 3
struct MapArgs {
    i32 dims, n[3], w[3];          // regular grid: nodes per axis; half-width of the candidate block per axis
    double lo[3], h[3];            // first node and spacing per axis
    i64 n_points; const double *points;    // point-list map (an irregular coordinate system): [n_points][dims]; else nullptr
    // 2-D point lists: a uniform cell index over the points (cell (i, j) = cell_start[i * cny + j] .. of cell_pts, ascending point index)
    const i32 *cell_start, *cell_pts; i32 cnx, cny; double cell_lo[2], cell_h;
    const float *values;           // [nodes][n_map_ch]
    i32 n_map_ch;                  // channels stored in the map; channels beyond get weight 1 (s2.py:648-650: top-only maps)
    i64 n_rows;                    // positions to evaluate
    const i32 *row_ins;            // [n_rows] instruction of the row (pattern rows)
    const float *x, *y, *z;        // [n_ins] positions of the instructions (pattern rows)
    const double *pos;             // or: [n_rows][dims] positions (scalar maps)
    const i64 *row_id;             // [n_rows] cdf row to write
    i64 *nb_idx; double *nb_w;     // [n_rows][MAP_K]
    double *cdf_table; unsigned short *cdf_guide;
    const double *gains;           // [n_tpc] 0: turned-off PMT
    const double *aft;             // [n_ins] s2_aft_sigma: factor on the top-array fraction of the instruction's pattern, or nullptr
    i32 n_top;
    const double *pre;             // [n_rows][n_map_ch] patterns already averaged (k_diffuse_patterns) instead of neighbour lists, or nullptr
};

__device__ __forceinline__ void map_position(const MapArgs &m, i64 r, double pos[3])
{
    if (m.pos) { for (int a = 0; a < 3; a++) pos[a] = a < m.dims ? m.pos[r * m.dims + a] : 0.0; return; }
    const i32 ins = m.row_ins[r];
    pos[0] = (double)m.x[ins]; pos[1] = (double)m.y[ins]; pos[2] = m.dims > 2 ? (double)m.z[ins] : 0.0;
}

__device__ __forceinline__ void map_store_neighbours(const MapArgs &m, i64 r, int K, const double *bd, const i64 *bi)
{
    for (int k = 0; k < MAP_K; k++) {
        const double dist = k < K ? sqrt(bd[k]) : 0.0;
        m.nb_idx[r * MAP_K + k] = k < K ? bi[k] : -1;
        m.nb_w[r * MAP_K + k] = (k < K && bi[k] >= 0) ? 1.0 / (dist < 1e-6 ? 1e-6 : dist) : 0.0;
    }
}

// the K = 2 * dims nearest nodes of a regular grid (squared distances ascending in bd, node indices in bi)
__device__ __forceinline__ void grid_nearest(const MapArgs &m, const double *pos, double *bd, i64 *bi)
{
    i32 c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
    for (int a = 0; a < m.dims; a++) {
        i32 c = (i32)floor((pos[a] - m.lo[a]) / m.h[a]);
        c = c < 0 ? 0 : (c > m.n[a] - 2 ? m.n[a] - 2 : c);
        c0[a] = c - m.w[a] + 1 < 0 ? 0 : c - m.w[a] + 1;
        c1[a] = c + m.w[a] > m.n[a] - 1 ? m.n[a] - 1 : c + m.w[a];
    }
    const int K = 2 * m.dims;
    for (int k = 0; k < MAP_K; k++) { bd[k] = 1e300; bi[k] = -1; }
    for (i32 i = c0[0]; i <= c1[0]; i++) {
        const double dx = pos[0] - (m.lo[0] + i * m.h[0]);
        for (i32 j = c0[1]; j <= c1[1]; j++) {
            const double dy = m.dims > 1 ? pos[1] - (m.lo[1] + j * m.h[1]) : 0.0;
            for (i32 l = c0[2]; l <= c1[2]; l++) {
                const double dz = m.dims > 2 ? pos[2] - (m.lo[2] + l * m.h[2]) : 0.0;
                const double d2 = dx * dx + dy * dy + dz * dz;
                if (!(d2 < bd[K - 1])) continue;            // ties keep the node met first (lower linear index)
                const i64 node = ((i64)i * (m.dims > 1 ? m.n[1] : 1) + j) * (m.dims > 2 ? m.n[2] : 1) + l;
                int q = K - 1;
                while (q > 0 && d2 < bd[q - 1]) { bd[q] = bd[q - 1]; bi[q] = bi[q - 1]; q--; }
                bd[q] = d2; bi[q] = node;
            }
        }
    }
}

// The same for a 2-D point list, through the cell index: rings of cells around the position's cell until the K-th neighbour found is
// closer than anything in the rings not yet visited can be (a cell at Chebyshev distance k lies at least (k - 1) cell widths away);
// equal distances: the lower point index, as the brute-force search of k_map_neighbours_points.
__device__ __forceinline__ void points_nearest_2d(const MapArgs &m, const double *pos, double *bd, i64 *bi)
{
    constexpr int K = 4;
    for (int k = 0; k < MAP_K; k++) { bd[k] = 1e300; bi[k] = -1; }
    i32 cx = (i32)floor((pos[0] - m.cell_lo[0]) / m.cell_h), cy = (i32)floor((pos[1] - m.cell_lo[1]) / m.cell_h);
    cx = cx < 0 ? 0 : (cx > m.cnx - 1 ? m.cnx - 1 : cx); cy = cy < 0 ? 0 : (cy > m.cny - 1 ? m.cny - 1 : cy);
    const i32 rmax = m.cnx > m.cny ? m.cnx : m.cny;
    for (i32 r = 0; r <= rmax; r++) {
        for (i32 i = cx - r; i <= cx + r; i++) {
            if (i < 0 || i >= m.cnx) continue;
            const bool whole = i == cx - r || i == cx + r;          // the ring's two full columns; in between only its top and bottom cell
            for (i32 j = cy - r; j <= cy + r; j += (whole || r == 0) ? 1 : 2 * r) {
                if (j < 0 || j >= m.cny) continue;
                const i64 cell = (i64)i * m.cny + j;
                for (i32 q = m.cell_start[cell]; q < m.cell_start[cell + 1]; q++) {
                    const i64 p = m.cell_pts[q];
                    const double dx = pos[0] - m.points[2 * p], dy = pos[1] - m.points[2 * p + 1];
                    const double d2 = dx * dx + dy * dy;
                    if (!(d2 < bd[K - 1] || (d2 == bd[K - 1] && p < bi[K - 1]))) continue;
                    int t = K - 1;
                    while (t > 0 && (d2 < bd[t - 1] || (d2 == bd[t - 1] && p < bi[t - 1]))) { bd[t] = bd[t - 1]; bi[t] = bi[t - 1]; t--; }
                    bd[t] = d2; bi[t] = p;
                }
            }
        }
        const double reach = (double)r * m.cell_h;
        if (bi[K - 1] >= 0 && bd[K - 1] < reach * reach) break;
    }
}

__global__ void k_map_neighbours(MapArgs m)
{
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m.n_rows) return;
    double pos[3]; map_position(m, r, pos);
    double bd[MAP_K]; i64 bi[MAP_K];
    grid_nearest(m, pos, bd, bi);
    map_store_neighbours(m, r, 2 * m.dims, bd, bi);
}

// point-list maps (straxen.InterpolatingMap on an irregular coordinate system: a KD-tree query for the 2 * dims nearest
// points): one wave per position, every lane scans a stride of the points keeping its own K best, then K rounds of a
// wave-wide argmin merge the 64 lists (ties: the lower point index).
__global__ __launch_bounds__(256) void k_map_neighbours_points(MapArgs m)
{
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= m.n_rows) return;                              // the whole wave leaves
    double pos[3]; map_position(m, r, pos);
    const int K = 2 * m.dims;
    double bd[MAP_K]; i64 bi[MAP_K];
#pragma unroll
    for (int k = 0; k < MAP_K; k++) { bd[k] = 1e300; bi[k] = 0x7fffffffffffffffLL; }
    for (i64 q = lane; q < m.n_points; q += 64) {
        double d2 = 0;
        for (int a = 0; a < m.dims; a++) { const double d = pos[a] - m.points[q * m.dims + a]; d2 += d * d; }
        if (!(d2 < bd[MAP_K - 1])) continue;
        // insertion into the sorted list of MAP_K (only the first K are used); fixed trip counts keep the lists in registers
        bool placed = false; double cd = d2; i64 ci = q;
#pragma unroll
        for (int k = 0; k < MAP_K; k++) {
            if (placed || cd < bd[k]) { const double td = bd[k]; const i64 ti = bi[k]; bd[k] = cd; bi[k] = ci; cd = td; ci = ti; placed = true; }
        }
    }
    double od[MAP_K]; i64 oi[MAP_K];
    for (int k = 0; k < MAP_K; k++) {
        double d = bd[0]; i64 i = bi[0];
        for (int o = 32; o > 0; o >>= 1) {
            const double xd = __shfl_xor(d, o, 64); const i64 xi = __shfl_xor(i, o, 64);
            if (xd < d || (xd == d && xi < i)) { d = xd; i = xi; }
        }
        if (bi[0] == i && i != 0x7fffffffffffffffLL) {     // the winner pops its head
#pragma unroll
            for (int q = 0; q < MAP_K - 1; q++) { bd[q] = bd[q + 1]; bi[q] = bi[q + 1]; }
            bd[MAP_K - 1] = 1e300; bi[MAP_K - 1] = 0x7fffffffffffffffLL;
        }
        od[k] = d; oi[k] = i == 0x7fffffffffffffffLL ? -1 : i;
    }
    if (lane == 0) map_store_neighbours(m, r, K, od, oi);
}

// Transverse diffusion with field maps (S2.s2_pattern_map_diffuse, s2.py:560-613): the pattern of an instruction is the average
// of the pattern map over its SURVIVING electrons' positions, each displaced by N(0, sigma_r) along the radius and N(0, sigma_a)
// across it; electrons that end outside tpc_radius do not count.  One workgroup per instruction, after k_s2_electrons: batches of
// 256 electrons -- a thread displaces one electron and finds its 4 neighbours (grid nodes, or the points of a point-list map through its cell index), then the threads, one or two
// channels each, add the 256 weighted patterns in electron order (a fixed order: reproducible sums).  An instruction without an
// electron inside gets no pattern: its electrons make no photons (the reference's NaN pattern sends them to channel -1).
struct DiffArgs { i64 n_rows; const i32 *row_ins; const double *sig_r, *sig_a; double r2max; double *pre; };

__global__ __launch_bounds__(256) void k_diffuse_patterns(WfsDev d, GenArgs a, MapArgs m, DiffArgs q)
{
    __shared__ i32 s_idx[256][4]; __shared__ double s_w[256][4]; __shared__ i32 s_n, s_wn[4];
    const i64 r = blockIdx.x; const int tid = threadIdx.x;
    const i32 ins = q.row_ins[r];
    const double x0 = (double)m.x[ins], y0 = (double)m.y[ins];
    const double th = atan2(y0, x0), ct = cos(th), st = sin(th);
    const double sr = q.sig_r[ins], sa = q.sig_a[ins];
    const i64 e0 = a.em_off[ins], e1 = a.em_off[ins + 1];
    const u32 jb = a.ins_embase[ins], gid = a.ins_gid[ins];
    double acc[2] = {0.0, 0.0};
    i64 n_in = 0;
    for (i64 base = e0; base < e1; base += 256) {
        const i64 e = base + tid;
        bool ok = e < e1 && a.em_time[e] != I64_MIN;
        double w[4] = {0, 0, 0, 0}, ws = 1; i64 bi[MAP_K];
        if (ok) {
            const u32x4 B = philox4x32_10(jb + (u32)(e - e0), gid, 0, SITE_EL_DIFF, d.k0, d.k1);
            double z0, z1;
            box_muller(B, z0, z1);
            const double hr = z0 * sr, ha = z1 * sa;
            const double pos[3] = {x0 + (ct * hr - st * ha), y0 + (st * hr + ct * ha), 0.0};
            ok = pos[0] * pos[0] + pos[1] * pos[1] <= q.r2max;
            if (ok) {
                double bd[MAP_K];
                if (m.points) points_nearest_2d(m, pos, bd, bi); else grid_nearest(m, pos, bd, bi);
                ws = 0;
                for (int k = 0; k < 4; k++) { const double dist = sqrt(bd[k]); w[k] = bi[k] >= 0 ? 1.0 / (dist < 1e-6 ? 1e-6 : dist) : 0.0; ws += w[k]; }
            }
        }
        // slots in electron order (ballot ranks): the sums below run in a fixed order
        const u64 mask = ballot64(ok);
        if ((tid & 63) == 0) s_wn[tid >> 6] = __popcll(mask);
        __syncthreads();
        int slot = __popcll(mask & ((1ull << (tid & 63)) - 1ull));
        for (int v = 0; v < (tid >> 6); v++) slot += s_wn[v];
        if (ok) for (int k = 0; k < 4; k++) { s_idx[slot][k] = bi[k] >= 0 ? (i32)bi[k] : 0; s_w[slot][k] = w[k] / ws; }
        if (tid == 0) s_n = s_wn[0] + s_wn[1] + s_wn[2] + s_wn[3];
        __syncthreads();
        const int nb = s_n;
        n_in += nb;
        for (int c = tid, u = 0; c < m.n_map_ch && u < 2; c += 256, u++) {
            double sum = acc[u];
            for (int p = 0; p < nb; p++)
                for (int k = 0; k < 4; k++) sum += s_w[p][k] * (double)m.values[(i64)s_idx[p][k] * m.n_map_ch + c];
            acc[u] = sum;
        }
        __syncthreads();
    }
    for (int c = tid, u = 0; c < m.n_map_ch && u < 2; c += 256, u++) q.pre[r * m.n_map_ch + c] = n_in > 0 ? acc[u] / (double)n_in : 0.0;
    if (n_in == 0) for (i64 e = e0 + tid; e < e1; e += 256) a.em_nph[e] = 0;
}

// scalar maps (LCE, S2 correction, SE gain, longitudinal diffusion ...: s1.py:125, s2.py:170-234): the weighted average itself
// (nv values per node: array-valued maps, load_resource.py:383-401 make_map -- one thread per (position, value))
__global__ void k_map_scalar(MapArgs m, const double *values, double *out, int nv)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m.n_rows * nv) return;
    const i64 r = t / nv; const int v = (int)(t - r * nv);
    double num = 0, den = 0;
    for (int k = 0; k < MAP_K; k++) {
        const i64 i = m.nb_idx[r * MAP_K + k];
        if (i < 0) continue;
        const double w = m.nb_w[r * MAP_K + k];
        num += values[i * nv + v] * w; den += w;
    }
    out[t] = den > 0 ? num / den : __builtin_nan("");
}

// scipy.interpolate.RegularGridInterpolator(method='linear', bounds_error=False, fill_value=None) -- what straxen's InterpolatingMap
// builds for method 'RegularGridInterpolator' (load_resource.py:357, 383-401): multilinear inside the grid, the edge cell's plane
// continued outside it.  Per axis the cell (clamped to [0, n - 2]) and the normalised distance to its lower node, then the sum over
// the cell's 2^dims corners of value * product of weights, corners in scipy's order (itertools.product of (i, i + 1) per axis).
struct LinearMapArgs { i32 dims, nv; i32 n[3]; double lo[3], h[3]; const double *values; i64 n_rows; const double *pos; double *out; };
__global__ void k_map_linear(LinearMapArgs a)
{
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_rows * a.nv) return;
    const i64 r = t / a.nv; const int v = (int)(t - r * a.nv);
    i64 cell[3] = {0, 0, 0}; double f[3] = {0, 0, 0};
    for (int q = 0; q < a.dims; q++) {
        const double x = (a.pos[r * a.dims + q] - a.lo[q]) / a.h[q];
        i64 c = (i64)floor(x);
        c = c < 0 ? 0 : (c > a.n[q] - 2 ? a.n[q] - 2 : c);
        cell[q] = c; f[q] = x - (double)c;
    }
    double sum = 0.0;
    for (int corner = 0; corner < (1 << a.dims); corner++) {
        double w = 1.0; i64 idx = 0;
        for (int q = 0; q < a.dims; q++) {
            const int up = (corner >> (a.dims - 1 - q)) & 1;
            w *= up ? f[q] : 1.0 - f[q];
            idx = idx * a.n[q] + cell[q] + up;
        }
        sum += a.values[idx * a.nv + v] * w;
    }
    a.out[t] = sum;
}

// scipy.interpolate.RectBivariateSpline.ev (the field-dependence and COMSOL distortion maps, load_resource.py:316, 326): FITPACK's
// bispeu on the spline's own knots and coefficients -- arguments clamped to the knot range, knot span by search, the
// k + 1 non-zero B-splines by the stable recurrence of de Boor and Cox (fpbspl), coefficients summed in fpbisp's order.
struct SplineArgs { i32 nx, ny, kx, ky; const double *tx, *ty, *c; i64 n; const double *pos; double *out; };

__device__ __forceinline__ int spline_span(const double *t, int n, int k, double &x, double h[6])
{
    const int nk1 = n - k - 1;
    const double tb = t[k], te = t[nk1];
    if (x < tb) x = tb;
    if (x > te) x = te;
    int l = k;
    while (!(x < t[l + 1]) && l + 1 != nk1) l++;
    double hh[5];
    h[0] = 1.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 0; i < j; i++) {
            const int li = l + i + 1, lj = li - j;
            if (t[li] == t[lj]) { h[i + 1] = 0.0; continue; }
            const double f = hh[i] / (t[li] - t[lj]);
            h[i] += f * (t[li] - x);
            h[i + 1] = f * (x - t[lj]);
        }
    }
    return l - k;
}

__global__ void k_map_spline(SplineArgs a)
{
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n) return;
    double x = a.pos[2 * r], y = a.pos[2 * r + 1], hx[6], hy[6];
    const int lx = spline_span(a.tx, a.nx, a.kx, x, hx), ly = spline_span(a.ty, a.ny, a.ky, y, hy);
    const int nky1 = a.ny - a.ky - 1;
    double sp = 0;
    for (int i = 0; i <= a.kx; i++)
        for (int j = 0; j <= a.ky; j++) sp += a.c[(i64)(lx + i) * nky1 + ly + j] * hx[i] * hy[j];
    a.out[r] = sp;
}

// one workgroup per row: weighted average of the neighbours' patterns, turned-off PMTs removed, normalised, cumulative sum,
// guide table (guide[c] = first channel whose cumulative probability exceeds c / CDF_G, as the host builds it)
__global__ __launch_bounds__(256) void k_map_rows(MapArgs m, int nch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *p = (double *)smem;                    // [nch]
    __shared__ double s_part[4]; __shared__ double s_sum;
    const i64 r = blockIdx.x; const int tid = threadIdx.x;
    double w[MAP_K]; i64 idx[MAP_K]; double wsum = 0;
    for (int k = 0; k < MAP_K; k++) { w[k] = m.pre ? 0.0 : m.nb_w[r * MAP_K + k]; idx[k] = m.pre ? -1 : m.nb_idx[r * MAP_K + k]; wsum += w[k]; }
    double part = 0;
    for (int c = tid; c < nch; c += 256) {
        double v = 1.0;
        if (c < m.n_map_ch && m.pre) v = m.pre[r * m.n_map_ch + c];
        else if (c < m.n_map_ch) {
            double acc = 0;
            for (int k = 0; k < MAP_K; k++) if (idx[k] >= 0) acc += w[k] * (double)m.values[idx[k] * m.n_map_ch + c];
            v = acc / wsum;
        }
        if (m.gains[c] == 0) v = 0;
        p[c] = v; part += v;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
    if ((tid & 63) == 0) s_part[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        // s2_aft_sigma (s2.py:660-665): the top-array fraction of the normalised pattern times the instruction's skew-normal
        // factor, clipped to [0, 1]; top channels scaled by new / cur, the others by (1 - new) / (1 - cur)
        double s_top = 1.0, s_bot = 1.0;
        if (m.aft && tot > 0) {
            double top = 0;
            for (int c = 0; c < m.n_top; c++) top += p[c];
            const double cur = top / tot, f = m.aft[m.row_ins[r]];
            if (cur > 0 && cur < 1 && f == f) {
                double nw = cur * f; nw = nw < 0 ? 0 : (nw > 1 ? 1 : nw);
                s_top = nw / cur; s_bot = (1 - nw) / (1 - cur);
            }
        }
        // np.random.choice: cdf = cumsum(p / sum); cdf /= cdf[-1]  (sequential, as numpy's cumsum)
        double run = 0;
        for (int c = 0; c < nch; c++) { run += tot > 0 ? p[c] / tot * (c < m.n_top ? s_top : s_bot) : 1.0 / nch; p[c] = run; }
        s_sum = run;
    }
    __syncthreads();
    const i64 rid = m.row_id[r];
    double *row = m.cdf_table + rid * nch;
    for (int c = tid; c < nch; c += 256) { p[c] = p[c] / s_sum; row[c] = p[c]; }
    __syncthreads();
    unsigned short *g = m.cdf_guide + rid * (CDF_G + 2);
    for (int c = tid; c <= CDF_G + 1; c += 256) {
        const double x = (double)c / CDF_G;
        int lo = 0, hi = nch - 1;                   // first channel with p[ch] > x, at most nch - 1
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (p[mid] > x) hi = mid; else lo = mid + 1; }
        g[c] = (unsigned short)lo;
    }
}

// Between the passes: where in its tile every single-instruction block puts its photons.  Thread = (instruction,
// channel): a running sum over the instruction's blocks in block order, starting behind whatever the generic path
// already counted into the tile.  Consecutive blocks get consecutive ranges (no atomics, a reproducible layout).
__global__ void k_block_ranges(WfsDev d, GenArgs a)
{
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int nch = d.n_tpc;
    if (idx >= a.n_psets * nch) return;
    const i32 set = (i32)(idx / nch); const int c = (int)(idx - (i64)set * nch);
    u32 run = (u32)a.tile_count[idx];
    for (i64 q = a.set_ins_off[set]; q < a.set_ins_off[set + 1]; q++) {      // the set's instructions, in order
        const i32 ins = a.set_ins_list[q];
        const i64 pa = a.ins_ph0[ins], pb = a.ins_ph0[ins + 1];     // photons [pa, pb) of the instruction
        if (pb <= pa) continue;
        const i64 b1 = (pb - 1) / GEN_BLOCK;
        constexpr int NB = 8;                                    // blocks at a time: their loads are in flight together
        for (i64 b0 = pa / GEN_BLOCK; b0 <= b1; b0 += NB) {
            u32 cnt[NB]; bool mine[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                const i64 b = b0 + k < b1 ? b0 + k : b1;
                const i32 bi = a.blk_ins[b];                     // (unconditional: a load behind && is waited for before the next one)
                cnt[k] = a.blk_cnt[b * nch + c];
                mine[k] = b0 + k <= b1 && bi == ins;
            }
#pragma unroll
            for (int k = 0; k < NB; k++) if (mine[k]) { a.blk_base[(b0 + k) * nch + c] = run; run += cnt[k]; }
        }
    }
    a.tile_tailbase[idx] = (i32)run;                             // the last photons of the instruction (generic path) go behind the block ranges
    run += (u32)a.tile_tail[idx]; a.tile_tail[idx] = 0;         // (the fill pass counts them again as it places them)
    a.tile_count[idx] = (i32)run;
}

// channel thresholds of a CDF row for 32-bit words: channel = first c with w / 2^32 < cdf[c], i.e. w < cdf[c] * 2^32, i.e.
// (w an integer) w <= ceil(cdf[c] * 2^32) - 1 =: T[c] -- exact, and the same decision as np.random.choice's
// searchsorted(cdf, u, 'right') on u = w / 2^32.  Channels with cdf == 0 are never examined: the search starts at
// guide[w >> 23], the first channel whose cumulative probability exceeds (w >> 23) / 512 >= 0.
__device__ __forceinline__ u32 cdf_threshold(double c)
{
    const double x = ceil(c * 4294967296.0);
    return x >= 4294967296.0 ? 0xffffffffu : (x >= 1.0 ? (u32)x - 1u : 0u);
}
// Channel of a photon from its 32-bit word: Walker's alias method over the 2^lg >= n_tpc cells of the instruction's row (built by
// k_chan_alias from the cumulative row, identically in the oracle): the top lg bits pick the cell, the remaining bits, left-aligned,
// are compared with the cell's threshold -- ONE 8-byte look-up instead of a guide read and a threshold search (RNG spec v8; the
// probabilities are those of np.random.choice's row to 2^-32, as with the thresholds before).
__device__ __forceinline__ int channel_lookup(const uint2 *A, int lg, u32 w)
{
    const u32 cell = w >> (32 - lg);
    const uint2 e = A[cell];
    return (int)(((w << lg) < e.x) ? cell : e.y);
}

// Alias cells of every channel CDF row.  One workgroup per row, thread 0 runs Vose's construction exactly as the host does for the
// delay tables (build_alias: cells below 1 and the others on two stacks, ascending fill, last in first out) so that the oracle,
// which restates it in C, gets the same cells bit for bit (IEEE adds and compares only).
__global__ __launch_bounds__(64) void k_chan_alias(const double *cdf_table, int nch, int lg, uint2 *out)
{
    __shared__ double q[WFS_MAX_CH];
    __shared__ unsigned short st_small[WFS_MAX_CH], st_large[WFS_MAX_CH];
    __shared__ uint2 cell[WFS_MAX_CH];                     // (built in LDS, written out together: the sequential part touches no global memory)
    const i64 r = blockIdx.x; const int K = 1 << lg;
    const double *cum = cdf_table + r * nch;
    // the two stacks, filled in ascending cell order by the whole wave (ballot ranks), then the sequential pairing on lane 0 with the
    // open "large" cell kept in registers
    int ns = 0, nl = 0;
    for (int i0 = 0; i0 < K; i0 += 64) {
        const int i = i0 + threadIdx.x;
        // (K = 2^lg can be below the wave width: lanes behind the last cell take no part -- phantom zero-mass cells on the small
        // stack would drain the large ones and leave the row uniform over all K cells, channels >= n_tpc included)
        const bool valid = i < K;
        const double qi = i < nch ? (cum[i] - (i ? cum[i - 1] : 0.0)) * (double)K : 0.0;
        if (valid) { q[i] = qi; cell[i] = uint2{0xffffffffu, (u32)i}; }
        const bool sm = valid && qi < 1.0, lgc = valid && !(qi < 1.0);
        const u64 ms = ballot64(sm), ml = ballot64(lgc), below = (1ull << threadIdx.x) - 1ull;
        if (sm) st_small[ns + __popcll(ms & below)] = (unsigned short)i; else if (lgc) st_large[nl + __popcll(ml & below)] = (unsigned short)i;
        ns += __popcll(ms); nl += __popcll(ml);
    }
    __syncthreads();
    if (threadIdx.x == 0 && ns > 0 && nl > 0) {
        int l = st_large[nl - 1]; double ql = q[l];
        for (;;) {
            const int sidx = st_small[--ns];
            const double qs = q[sidx];
            const double t = qs * 4294967296.0;
            cell[sidx] = uint2{t >= 4294967295.0 ? 0xffffffffu : (u32)t, (u32)l};
            ql = (ql + qs) - 1.0;
            if (ql < 1.0) { q[l] = ql; nl--; st_small[ns++] = (unsigned short)l; if (nl == 0) break; l = st_large[nl - 1]; ql = q[l]; }
            if (ns == 0) break;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 64) out[r * K + i] = cell[i];
}

// photon p (generation order) -> emitter, instruction, Philox coordinates; generic path and k_photon_times
struct PhotonId { i64 em; i32 ins; u32 gid, eb, j, m, P; };
__device__ __forceinline__ PhotonId photon_id(const GenArgs &a, i64 p, i64 lo = 0, i64 hi = -1)
{
    if (hi < 0) hi = a.n_emitters;                 // emitter of p lies in [lo, hi): the caller may know a narrower range
    while (hi - lo > 1) { i64 mid = (lo + hi) >> 1; if (a.em_ph_off[mid] <= p) lo = mid; else hi = mid; }
    PhotonId r; r.em = lo; r.ins = a.em_ins[lo]; r.gid = a.ins_gid[r.ins]; r.eb = a.ins_embase[r.ins];
    r.j = (u32)(lo - a.em_off[r.ins]) + r.eb; r.m = (u32)(p - a.em_ph_off[lo]); r.P = (u32)(p - a.ins_ph0[r.ins]);
    return r;
}
__device__ __forceinline__ u32 photon_word(const WfsDev &d, const PhotonId &id, u32 site)
{
    return word_of(philox4x32_10(id.eb, id.gid, id.P >> 2, site, d.k0, d.k1), (int)(id.P & 3u));
}
__device__ __forceinline__ int photon_channel_global(const WfsDev &d, const GenArgs &a, const PhotonId &id)
{
    return channel_lookup(a.chan_alias + ((size_t)a.ins_cdfrow[id.ins] << a.ch_lg), a.ch_lg, photon_word(d, id, SITE_CH));
}

// A block that spans instructions (generic path) holds the LAST photons of the instruction it starts in and the FIRST photons of the
// ones that begin in it.  The tile of a single-instruction pulse set is laid out [first photons (generic) | block ranges in block
// order | last photons (generic)]: photon order up to the order inside the two small generic groups, which k_tile_order_scan repairs.
__device__ __forceinline__ bool generic_is_tail(const GenArgs &a, const PhotonId &id, i64 p0)
{
    return a.ins_ph0[id.ins] < p0 && !a.ins_fullsort[id.ins];
}

#define COUNT_TPB (GEN_BLOCK / 8)
#define FILL_TPB (GEN_BLOCK / 4)   // one quad of photons per thread: many waves per workgroup keep enough table gathers in flight

// Workgroups are dealt to the 8 XCDs round robin: XCD x (= workgroup id % 8) takes the photon blocks [x * chunk, (x + 1) * chunk)
// in order, so that the blocks writing neighbouring ranges of a tile run close together in time behind the same L2.
__device__ __forceinline__ i64 block_of_workgroup(const GenArgs &a) { return (i64)(blockIdx.x & 7) * a.xcd_chunk + (blockIdx.x >> 3); }

// 'garfield_gas_gap' luminescence: sum of the excitation times of every instruction's photons (block by block like the count
// pass; only launched when a loaded instruction uses the model)
__global__ __launch_bounds__(256) void k_gg_sum(WfsDev d, GenArgs a)
{
    const i64 vb = blockIdx.x;
    if (vb >= a.n_blocks) return;
    const BlockDesc bd = a.blk_desc[vb];
    const i64 p0 = vb * GEN_BLOCK;
    const int np = (int)((p0 + GEN_BLOCK < a.n_photons) ? GEN_BLOCK : a.n_photons - p0);
    if (bd.ins >= 0) {
        const i32 lo = a.ins_gg[bd.ins];
        if (lo < 0) return;                                  // block-uniform
        const double wgt = a.ins_ggw[bd.ins];
        const u32 q0 = (u32)(bd.R0 >> 2); const int r = (int)(bd.R0 & 3);
        i64 sum = 0;
        for (int qi = threadIdx.x; 4 * qi - r < np; qi += 256) {
            const u32x4 W = philox4x32_10(bd.eb, bd.gid, q0 + (u32)qi, SITE_LUM, d.k0, d.k1);
#pragma unroll
            for (int k = 0; k < 4; k++) { const int pr = 4 * qi + k - r; if (pr >= 0 && pr < np) sum += __double2ll_rn(gg_time(a, lo, wgt, word_of(W, k)) * 1048576.0); }
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
        if ((threadIdx.x & 63) == 0 && sum) atomicAdd((u64 *)&a.ins_ggsum[bd.ins], (u64)sum);
    } else {
        for (int pr = threadIdx.x; pr < np; pr += 256) {
            const PhotonId id = photon_id(a, p0 + pr, a.blk_e[2 * vb], a.blk_e[2 * vb + 1] + 1);
            const i32 lo = a.ins_gg[id.ins];
            if (lo >= 0) atomicAdd((u64 *)&a.ins_ggsum[id.ins], (u64)__double2ll_rn(gg_time(a, lo, a.ins_ggw[id.ins], photon_word(d, id, SITE_LUM)) * 1048576.0));
        }
    }
}

#define GEN_COUNT_LDS(nch, lg) ((size_t)(8u << (lg)) + (size_t)(nch) * 4)

__global__ __launch_bounds__(COUNT_TPB) void k_photon_count(WfsDev d, GenArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nch = d.n_tpc, tid = threadIdx.x;
    const int lg = a.ch_lg, K = 1 << lg;
    uint2 *A = (uint2 *)smem;                                       // [K] alias cells of the instruction's channel row
    i32 *hist = (i32 *)(smem + (size_t)K * 8);                      // [nch]
    const i64 vb = block_of_workgroup(a);
    if (vb >= a.n_blocks) return;                            // block-uniform (padding of the XCD order)
    const BlockDesc bd = a.blk_desc[vb];                     // block-uniform: scalar loads
    const i64 p0 = vb * GEN_BLOCK;
    const int np = (int)((p0 + GEN_BLOCK < a.n_photons) ? GEN_BLOCK : a.n_photons - p0);
    if (bd.ins >= 0) {
        for (int c = tid; c < nch; c += COUNT_TPB) hist[c] = 0;
        for (int c = tid; c < K; c += COUNT_TPB) A[c] = a.chan_alias[((size_t)bd.row << lg) + c];
        __syncthreads();
        const u32 q0 = (u32)(bd.R0 >> 2); const int r = (int)(bd.R0 & 3);
        for (int qi = tid; 4 * qi - r < np; qi += COUNT_TPB) {  // quad qi of the block: photons 4 * qi - r .. + 3 (block-relative)
            const u32x4 W = philox4x32_10(bd.eb, bd.gid, q0 + (u32)qi, SITE_CH, d.k0, d.k1);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int pr = 4 * qi + k - r;
                if (pr >= 0 && pr < np) atomicAdd(&hist[channel_lookup(A, lg, word_of(W, k))], 1);
            }
        }
        __syncthreads();
        for (int c = tid; c < nch; c += COUNT_TPB) a.blk_cnt[vb * nch + c] = (unsigned short)hist[c];     // -> k_block_ranges
    } else {
        for (int pr = tid; pr < np; pr += COUNT_TPB) {
            const PhotonId id = photon_id(a, p0 + pr, a.blk_e[2 * vb], a.blk_e[2 * vb + 1] + 1);      // (23 dependent loads over all emitters otherwise)
            const int ch = photon_channel_global(d, a, id);
            const i64 tile = (i64)a.ins_set[id.ins] * nch + ch;
            atomicAdd(generic_is_tail(a, id, p0) ? &a.tile_tail[tile] : &a.tile_count[tile], 1);
        }
    }
}

// LDS layout of the fill pass (byte offsets; plain integer offsets: a pointer that went through an integer cast loses its
// LDS address space)
struct GenFillLds { int wtime, T, hist, cur, hmin, hmax, hoff, chmap, pidx, stage, ap, total; };
__host__ __device__ inline GenFillLds gen_fill_lds(int nch, int lg, bool with_ap)
{
    const int nch1 = nch + 1 + ((nch + 1) & 1);              // even: keeps what follows 8-byte aligned
    GenFillLds o;
    o.wtime = GEN_WIN * 4;                                    // win i32[GEN_WIN] at 0: first photon of the block's emitters, relative to the block
    o.T = o.wtime + GEN_WIN * 4;                              // wtime i32[GEN_WIN]: emitter times relative to the set's origin
    o.hist = o.T + (8 << lg);                                 // T: uint2[2^lg] alias cells of the channel row; hist i32[nch1]: photons per channel, then their prefix sums
    o.cur = o.hist + nch1 * 4;                                // i32[nch]: rank counters
    o.hmin = o.cur + nch * 4;                                 // i32[nch], i32[nch]: earliest / latest photon of the block per channel
    o.hmax = o.hmin + nch * 4;
    o.hoff = o.hmax + nch * 4;                                // i32[nch]: slot of bucket position 0 of every channel, relative to the set's first photon
    o.chmap = o.hoff + nch * 4;                               // u16[GEN_BLOCK]: channel of every bucket position
    o.pidx = o.chmap + GEN_BLOCK * 2;                         // u16[GEN_BLOCK]: block-relative photon index of every bucket position
    o.stage = (o.pidx + GEN_BLOCK * 2 + 7) & ~7;              // PhotonRec[GEN_BLOCK]: the block's photons in bucket order
    o.ap = o.stage + GEN_BLOCK * 8;                           // afterpulse staging
    o.total = ((o.ap + (with_ap ? AP_STAGE * (int)sizeof(ApCand) : 0) + 7) & ~7) + 16;
    return o;
}

// EXT: delay tables per instruction and array (model variants), S1 optical propagation term
template <bool AP, bool EXT>
__global__ __launch_bounds__(FILL_TPB) void k_photon_fill(WfsDev d, GenArgs a, ApArgs ap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TPB = FILL_TPB, CPT = (WFS_MAX_CH + TPB - 1) / TPB;      // channels per thread
    const int nch = d.n_tpc, tid = threadIdx.x;
    const int lg = a.ch_lg;
    const GenFillLds o = gen_fill_lds(nch, lg, AP);
    i32 *win = (i32 *)smem, *wtime = (i32 *)(smem + o.wtime);
    uint2 *T = (uint2 *)(smem + o.T);                        // alias cells of the block's channel row
    i32 *hist = (i32 *)(smem + o.hist), *cur = (i32 *)(smem + o.cur), *hmin = (i32 *)(smem + o.hmin), *hmax = (i32 *)(smem + o.hmax);
    i32 *hoff = (i32 *)(smem + o.hoff);
    unsigned short *chmap = (unsigned short *)(smem + o.chmap), *pidx = (unsigned short *)(smem + o.pidx);
    PhotonRec *stage = (PhotonRec *)(smem + o.stage);
    __shared__ i32 s_wtmp[TPB / 64];
    ApStage aps;
    aps.cand = (ApCand *)(smem + o.ap);
    __shared__ i32 s_apn; __shared__ i64 s_apbase;
    aps.n = &s_apn;
    if (AP && tid == 0) s_apn = 0;
    const i64 vb = block_of_workgroup(a);
    if (vb >= a.n_blocks) return;
    STAMP_INIT;
    const BlockDesc bd = a.blk_desc[vb];                     // block-uniform: scalar loads
    const i64 p0 = vb * GEN_BLOCK;
    const int np = (int)((p0 + GEN_BLOCK < a.n_photons) ? GEN_BLOCK : a.n_photons - p0);

    if (bd.ins >= 0) {
        // ---- fast path: all photons of the block belong to one instruction.  Loads that depend on the block alone go first.
        i32 cnt_r[CPT]; u32 base_r[CPT];
#pragma unroll
        for (int q = 0; q < CPT; q++) { const int c = tid + q * TPB; cnt_r[q] = c < nch ? (i32)a.blk_cnt[vb * nch + c] : 0; base_r[q] = c < nch ? a.blk_base[vb * nch + c] : 0u; }
        const int nwin = bd.nwin;                            // <= GEN_WIN (k_block_emitters)
        const u32 gid = bd.gid, eb = bd.eb; const i32 set_lo = bd.set; const i64 itime = bd.itime;     // photon times are relative to the pulse set's origin
        const AliasTab &tab = bd.is_s2 ? d.tab_s2 : d.tab_s1;
        AliasTab tab_t = tab, tab_b = tab; i32 pzi = -1; double pzf = 0.0;
        if (EXT) {
            tab_t = a.tabs[a.ins_tab[bd.ins]]; tab_b = a.tabs[a.ins_tabb[bd.ins]];
            if (a.prop_top && !bd.is_s2) { pzi = a.ins_pzi[bd.ins]; pzf = a.ins_pzf[bd.ins]; }
        }
        i32 gg_lo = -1; double gg_w = 0.0, gg_m = 0.0;       // garfield gas gap luminescence of this instruction
        if (EXT && a.gg_inv && a.ins_gg[bd.ins] >= 0) { gg_lo = a.ins_gg[bd.ins]; gg_w = a.ins_ggw[bd.ins]; gg_m = gg_mean(a, bd.ins); }
#define TAB_OF(c) (EXT ? ((c) >= d.n_top ? tab_b : tab_t) : tab)
        const i64 tbase = (i64)set_lo * nch;
        const i64 set_ph0 = a.tile_off[tbase];               // first photon slot of the set: per-channel offsets fit 32 bits
        PhotonRec *out = a.ph + set_ph0;
        const u32 q0 = (u32)(bd.R0 >> 2); const int r = (int)(bd.R0 & 3);
        i32 toff_r[CPT];
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            const int c = tid + q * TPB;
            if (c < nch) {
                toff_r[q] = (i32)(a.tile_off[tbase + c] - set_ph0);
                hist[c] = cnt_r[q]; cur[c] = 0; hmin[c] = 0x7fffffff; hmax[c] = (i32)0x80000000;
            }
        }
        for (int c = tid; c < (1 << lg); c += TPB) T[c] = a.chan_alias[((size_t)bd.row << lg) + c];
        for (int k = tid; k < nwin; k += TPB) {
            win[k] = (i32)(a.em_ph_off[bd.e_lo + k] - p0);
            // the block's own emitters only (slot nwin - 1 is the sentinel behind them); I64_MIN: an electron that did not survive (no photons)
            i64 wt = (k < nwin - 1) ? a.em_time[bd.e_lo + k] : I64_MIN;
            wt = wt == I64_MIN ? 0 : wt - itime;
            if (wt > 0x3fffffffLL || wt < -0x3fffffffLL) atomicMax(&a.scal[1], (i64)2);      // emitter further than 2^30 ns from its set's origin
            wtime[k] = (i32)wt;
        }
        __syncthreads();                                     // hist is complete
        STAMP(d, 0);
        block_excl_scan<TPB>(hist, nch, s_wtmp);             // (ends with a barrier)
#pragma unroll
        for (int q = 0; q < CPT; q++) {
            const int c = tid + q * TPB;
            if (c < nch) {
                hoff[c] = toff_r[q] + (i32)base_r[q] - hist[c];      // (the channel of every bucket position is noted when its photon is staged)
            }
        }
        STAMP(d, 1);
        // ---- generation order: a thread takes a quad of consecutive photons; their channel, delay and gain words come from
        // three Philox calls.  The finished photon goes to its bucket position (rank inside its channel) in LDS.
        for (int qi = tid; 4 * qi - r < np; qi += TPB) {
            const u32x4 C = philox4x32_10(eb, gid, q0 + (u32)qi, SITE_CH, d.k0, d.k1);
            const u32x4 D = philox4x32_10(eb, gid, q0 + (u32)qi, SITE_DELAY, d.k0, d.k1);
            const int pr0 = 4 * qi - r;
            int ch[4]; u32 cell[4]; uint2 e[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {                    // (photons outside the block: harmless values, nothing stored)
                ch[k] = channel_lookup(T, lg, word_of(C, k));
                cell[k] = alias_cell(TAB_OF(ch[k]), word_of(D, k));
                e[k] = TAB_OF(ch[k]).cell[cell[k]];
            }
            const u32x4 G = philox4x32_10(eb, gid, q0 + (u32)qi, SITE_GAIN, d.k0, d.k1);      // (under the latency of the table gathers)
            u32x4 LW = u32x4{0u, 0u, 0u, 0u};
            if (EXT && gg_lo >= 0) LW = philox4x32_10(eb, gid, q0 + (u32)qi, SITE_LUM, d.k0, d.k1);
            // emitter of the quad's first photon: last slot with win[slot] <= pr (emitters without photons are skipped);
            // photons are spread evenly over the block's emitters, so an interpolated guess lands within a step or two
            int slot;
            {
                const int prf = pr0 < 0 ? 0 : pr0;
                slot = (int)(((u32)prf * (u32)(nwin - 1)) >> GEN_LOG);   // GEN_BLOCK >= np: slot <= nwin - 2
                while (win[slot] > prf) slot--;                          // win[0] <= 0 <= prf
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int pr = pr0 + k;
                if (pr < 0 || pr >= np) continue;
                while (win[slot + 1] <= pr) slot++;          // win[nwin - 1] = first photon behind the block's last emitter > pr
                const int c = ch[k];
                i32 t = wtime[slot] + alias_pick(TAB_OF(c), word_of(D, k), cell[k], e[k]);
                const u32 code = gain_code(word_of(G, k), d.thr_dpe, d.dpe_inv);
                const u32 j = bd.jbase + (u32)slot, m = (u32)(pr - win[slot]);
                if (EXT && pzi >= 0) {                       // s1.py:185-188: int64 array, the assignment truncates
                    const u32x4 X = philox4x32_10(j, gid, m, SITE_PH_X, d.k0, d.k1);
                    t += (i32)(i64)s1_propagation(a, c >= d.n_top, pzi, pzf, X.x);
                }
                if (EXT && gg_lo >= 0) t += (i32)(i64)(gg_time(a, gg_lo, gg_w, word_of(LW, k)) - gg_m);      // s2.py:447-450, the cast of :532
                if (AP) ap_generate(d, ap, aps, j, gid, m, set_lo, c, (code >> 16) != 0, itime, (i64)t, bd.sbase + (u32)(bd.R0 + pr));
                atomicMin(&hmin[c], t); atomicMax(&hmax[c], t);
                const int pos = hist[c] + atomicAdd(&cur[c], 1);
                stage[pos] = PhotonRec{t, code}; chmap[pos] = (unsigned short)c; pidx[pos] = (unsigned short)pr;
            }
        }
        STAMP(d, 2);
        __syncthreads();
        // the few photons of every (block, channel) segment into photon order (an insertion sort per channel): with consecutive
        // blocks owning consecutive ranges the tile is then in generation order as it is written, k_tile_order finds nothing to do
        for (int c = tid; c < nch; c += TPB) {
            const int s0 = hist[c], s1 = hist[c + 1];
            if (s1 - s0 < 2 || s1 - s0 > 64) continue;       // (longer segments: left to k_tile_order)
            for (int i = s0 + 1; i < s1; i++) {
                const unsigned short key = pidx[i]; const PhotonRec r = stage[i];
                int j = i - 1;
                while (j >= s0 && pidx[j] > key) { pidx[j + 1] = pidx[j]; stage[j + 1] = stage[j]; j--; }
                pidx[j + 1] = key; stage[j + 1] = r;
            }
        }
        __syncthreads();
        STAMP(d, 3);
        // ---- bucket order: neighbouring lanes store photons of the same tile to consecutive addresses
        // (order keys only where the tile is sorted as a whole: the block ranges of a single-instruction set are in order as written)
        if (a.ins_fullsort[bd.ins]) { u32 *out_idx = a.ph_idx + set_ph0; for (int i = tid; i < np; i += TPB) { const int at = hoff[chmap[i]] + i; out[at] = stage[i]; out_idx[at] = bd.sbase + (u32)(bd.R0 + pidx[i]); } }
        else for (int i = tid; i < np; i += TPB) out[hoff[chmap[i]] + i] = stage[i];
        STAMP(d, 4);
        for (int c = tid; c < nch; c += TPB)
            if (hist[c + 1] > hist[c]) { atomicMin(&a.tile_tmin[tbase + c], hmin[c]); atomicMax(&a.tile_tmax[tbase + c], hmax[c]); }
        STAMP(d, 5);
#undef TAB_OF
    } else {
        // ---- generic path: per-photon global lookups and atomics
        if (AP) __syncthreads();                             // s_apn
        for (int pr = tid; pr < np; pr += TPB) {
            const PhotonId id = photon_id(a, p0 + pr, a.blk_e[2 * vb], a.blk_e[2 * vb + 1] + 1);
            const i32 ins = id.ins;
            const int ch = photon_channel_global(d, a, id);
            const i32 set = a.ins_set[ins];
            const i64 tile = (i64)set * nch + ch;
            const i64 itime = a.set_t0[set];
            const u32 code = gain_code(photon_word(d, id, SITE_GAIN), d.thr_dpe, d.dpe_inv);
            i64 t = a.em_time[id.em] - itime;
            const AliasTab &tb = EXT ? a.tabs[ch >= d.n_top ? a.ins_tabb[ins] : a.ins_tab[ins]] : (a.ins_type[ins] != 1 ? d.tab_s2 : d.tab_s1);
            t += alias_sample(tb, photon_word(d, id, SITE_DELAY));
            if (EXT && a.prop_top && a.ins_type[ins] == 1 && a.ins_pzi[ins] >= 0) {
                const u32x4 X = philox4x32_10(id.j, id.gid, id.m, SITE_PH_X, d.k0, d.k1);
                t += (i64)s1_propagation(a, ch >= d.n_top, a.ins_pzi[ins], a.ins_pzf[ins], X.x);
            }
            if (EXT && a.gg_inv && a.ins_gg[ins] >= 0) t += (i64)(gg_time(a, a.ins_gg[ins], a.ins_ggw[ins], photon_word(d, id, SITE_LUM)) - gg_mean(a, ins));
            if (t > 0x7fffffffLL || t < -0x7fffffffLL) { atomicMax(&a.scal[1], (i64)2); t = 0; }
            if (AP) ap_generate(d, ap, aps, id.j, id.gid, id.m, set, ch, (code >> 16) != 0, itime, t, a.ins_sbase[ins] + id.P);
            atomicMin(&a.tile_tmin[tile], (i32)t); atomicMax(&a.tile_tmax[tile], (i32)t);
            // photons of multi-instruction blocks take the first slots of their tile (k_block_ranges starts behind them)
            const i64 at = a.tile_off[tile] + (generic_is_tail(a, id, p0) ? a.tile_tailbase[tile] + atomicAdd(&a.tile_tail[tile], 1) : atomicAdd(&a.tile_cursor[tile], 1));
            a.ph[at] = PhotonRec{(i32)t, code}; a.ph_idx[at] = a.ins_sbase[ins] + id.P;
        }
    }
    if (AP) {
        __syncthreads();
        const int nst = s_apn < AP_STAGE ? s_apn : AP_STAGE;
        if (tid == 0 && nst > 0) s_apbase = (i64)atomicAdd((u64 *)ap.count, (u64)nst);
        __syncthreads();
        for (int k = tid; k < nst; k += TPB) { const i64 gk = s_apbase + k; if (gk < ap.cap) ap.cand[gk] = aps.cand[k]; }
    }
}

// one thread per candidate: the reference's acceptance comparison, then delay and amplitude of the accepted ones (ap_finish); a rejected
// candidate leaves a hole that k_ap_count / k_ap_place skip
__global__ void k_ap_finish(WfsDev d, GenArgs a, ApArgs ap)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 n = *ap.count < ap.cap ? *ap.count : ap.cap;
    if (i >= n) return;
    const ApCand q = ap.cand[i];
    const u32x4 w = ap_call(d, q);                          // the element's own call: low bits of the first uniform, the second uniform
    const bool ok = ap_accept(d, ap, q, w);
    if (ok) ap_finish(d, a.scal, ap, q, w, i); else ap.ap_ch[i] = -1;
    // (no counter of the accepted ones here: one atomic per wave on one address was 5.5 of this kernel's 6.8 ms with 3.5 x 10^7
    // candidates -- the scan over the afterpulse tiles' counts yields the total anyway)
}

// afterpulse photons -> tiles of the afterpulse pulse set of their primary set (set n_psets + set): count, then place
// Runs of consecutive list entries that go to the same tile (the candidates of a k_s2_tile workgroup all do, in key order) share one
// atomic: the first lane of a run reserves for the run, the others take their place behind it -- in list order, so a tile filled by
// one run is in generation order as it is written.  (70 afterpulses per tile of a 10^6-PE S2: one atomic per photon on the tile's
// cursor and two on its time range serialised the kernel, 8.3 ms for 3.5 x 10^7 afterpulses.)
struct ApRun { int head, rank, len; };
__device__ __forceinline__ ApRun ap_run(i64 tile, int lane)
{
    const i64 prev = __shfl_up(tile, 1, 64);
    const u64 heads = ballot64(lane == 0 || tile != prev);
    ApRun r;
    r.head = 63 - __clzll(heads & ((2ull << lane) - 1ull));
    const u64 behind = r.head == 63 ? 0ull : heads & ~((2ull << r.head) - 1ull);
    r.len = (behind ? __builtin_ctzll(behind) : 64) - r.head;
    r.rank = lane - r.head;
    return r;
}
__global__ __launch_bounds__(256) void k_ap_count(WfsDev d, GenArgs a, ApArgs ap)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const i64 n = *ap.count < ap.cap ? *ap.count : ap.cap;
    const bool v = i < n && ap.ap_ch[i] >= 0 && !(ap.ap_ins[i] & AP_SEG_MARK);
    const i64 tile = v ? ((i64)a.n_psets + ap.ap_ins[i]) * d.n_tpc + ap.ap_ch[i] : -1 - lane;       // (a hole is a run of its own)
    const ApRun r = ap_run(tile, lane);
    if (v && r.rank == 0) atomicAdd(&a.tile_count[tile], r.len);
}

__global__ __launch_bounds__(256) void k_ap_place(WfsDev d, GenArgs a, ApArgs ap, double *ph_gain_base)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const i64 n = *ap.count < ap.cap ? *ap.count : ap.cap;
    const bool v = i < n && ap.ap_ch[i] >= 0 && !(ap.ap_ins[i] & AP_SEG_MARK);
    const i64 tile = v ? ((i64)a.n_psets + ap.ap_ins[i]) * d.n_tpc + ap.ap_ch[i] : -1 - lane;
    const ApRun r = ap_run(tile, lane);
    const i32 t = v ? ap.ap_t[i] : 0;
    i32 first = 0;
    if (v && r.rank == 0) first = atomicAdd(&a.tile_cursor[tile], r.len);
    first = __shfl(first, r.head, 64);
    // earliest / latest photon of the run, reduced towards its first lane
    i32 lo = t, hi = t;
    for (int o = 1; o < 64; o <<= 1) {
        const i32 l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
        if (r.rank + o < r.len) { lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
    }
    if (!v) return;
    const i64 pos = a.tile_off[tile] + first + r.rank;
    a.ph[pos] = PhotonRec{t, 0u}; ph_gain_base[pos] = ap.ap_gain[i]; a.ph_idx[pos] = ap.ap_key[i];
    if (r.rank == 0) { atomicMin(&a.tile_tmin[tile], lo); atomicMax(&a.tile_tmax[tile], hi); }
}

// The afterpulses of a tile-generated tile: ONE wave per segment of the list (ApSeg).  PLACE false: the accepted entries are counted
// into the afterpulse tile; PLACE true: they go to the tile in list order = key order -- the tile is in generation order as written,
// whatever order the atomics of other workgroups happen in (k_tile_order finds nothing to do).
template <bool PLACE>
__global__ __launch_bounds__(256) void k_ap_seg(WfsDev d, GenArgs a, ApArgs ap, double *ph_gain_base)
{
    const int lane = threadIdx.x & 63;
    const i64 w = (i64)blockIdx.x * 4 + wave_in_block();
    if (w >= ap.n_seg) return;
    const ApSeg sg = ap.seg[w];
    if (sg.n <= 0) return;
    i32 total = 0;
    for (i32 c = 0; c < sg.n; c += 64) {
        const i64 i = sg.base + c + lane;
        total += __popcll(ballot64(c + lane < sg.n && i < ap.cap && ap.ap_ch[i] >= 0));
    }
    if (total == 0) return;
    if (!PLACE) { if (lane == 0) atomicAdd(&a.tile_count[sg.tile], total); return; }
    i32 first = 0;
    if (lane == 0) first = atomicAdd(&a.tile_cursor[sg.tile], total);
    first = __builtin_amdgcn_readfirstlane(first);
    const i64 off = a.tile_off[sg.tile] + first;
    i32 done = 0, lo = 0x7fffffff, hi = (i32)0x80000000;
    for (i32 c = 0; c < sg.n; c += 64) {
        const i64 i = sg.base + c + lane;
        const bool v = c + lane < sg.n && i < ap.cap && ap.ap_ch[i] >= 0;
        const u64 m = ballot64(v);
        if (v) {
            const i64 pos = off + done + __popcll(m & ((1ull << lane) - 1ull));
            const i32 t = ap.ap_t[i];
            a.ph[pos] = PhotonRec{t, 0u}; ph_gain_base[pos] = ap.ap_gain[i]; a.ph_idx[pos] = ap.ap_key[i];
            lo = t < lo ? t : lo; hi = t > hi ? t : hi;
        }
        done += __popcll(m);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    if (lane == 0) { atomicMin(&a.tile_tmin[sg.tile], lo); atomicMax(&a.tile_tmax[sg.tile], hi); }
}

// Generation order inside every tile.  The bucketing above leaves the photons of a tile in the order the atomics happened to
// hand out slots; the reference's Pulse call sees them in channel-sorted GENERATION order (a stable argsort would; the oracle
// does), and two things depend on that order: the truth quirk that counts the triggered photons among the FIRST n_dpe of the
// channel slice (pulse.py:255) and the order in which three or more photons of one ns are merged.  Every stored photon carries
// its order key (ph_idx); tiles whose keys are out of order are listed (one thread per tile) and sorted: by a wave up to 64
// photons (rank by counting smaller keys), by a workgroup in LDS up to TILE_ORDER_MAX; larger ones -- tiles of the per-electron
// generator beyond 4096 photons: S2s of 10^5 electrons with a gain spread or inside run sets -- go on a third list and through a
// segmented radix sort over (order key, position) pairs (k_order_huge_pack / rocPRIM / k_order_huge_apply, wfs_engine.hip).
// Explicit gains (afterpulses) move along.
#define TILE_ORDER_MAX 4096
#define TILE_ORDER_INLINE 12        // ranges up to this many photons are sorted by the scanning thread itself
struct OrderRange { i64 start; i32 n, pad; };          // photons [start, start + n) of the photon array
struct OrderArgs { i64 n_tiles, n_ptiles; const i32 *tile_count; const i64 *tile_off; PhotonRec *ph; u32 *ph_idx; double *ph_gain; i64 gain_first;
                   OrderRange *wave_list, *big_list; i64 *scal;      // ph_gain[p - gain_first] for photons p >= gain_first (afterpulses), or nullptr;
                                                                     // scal[30] / scal[31]: ranges on the wave list / the workgroup list; ranges beyond
                                                                     // TILE_ORDER_MAX photons: big_list[2 n_tiles - 1 - k], scal[19] of them
                   const i32 *skip_ins; i32 nch;                     // [n_ins] 1: the tiles of this instruction (tile / nch) come from k_s2_tile, already in order (or nullptr)
                   const i32 *tile_cursor, *tile_tailbase; const i32 *ins_fullsort; const i64 *set_ins_off; const i32 *set_ins_list; };
                   // primary tiles of a set whose (first) instruction is not flagged ins_fullsort: head = [0, cursor), tail = [tailbase, n)
// pass 1, one THREAD per tile.  A primary tile of a single-instruction pulse set is [first photons | block ranges | last photons]
// with the block ranges in order as written: only the two small generic groups are looked at.  Every other tile (afterpulse sets,
// sets of several instructions) is looked at as a whole.  A range that is out of order is sorted on the spot when it is tiny,
// listed for a wave (<= 64) or a workgroup (<= TILE_ORDER_MAX) otherwise.
__device__ __forceinline__ void order_range(const OrderArgs &a, i64 start, i32 n, int &cls)
{
    cls = -1;
    if (n < 2) return;
    const u32 *k = a.ph_idx + start;
    bool sorted = true;
    u32 prev = k[0];
    for (i32 q = 1; q < n; q++) { const u32 x = k[q]; if (x < prev) { sorted = false; break; } prev = x; }
    if (sorted) return;
    if (n > TILE_ORDER_INLINE) { cls = n <= 64 ? 0 : (n <= TILE_ORDER_MAX ? 1 : 2); return; }
    const bool has_gain = a.ph_gain && start >= a.gain_first;
    for (i32 i = 1; i < n; i++) {                            // insertion sort of a handful of records
        const u32 key = a.ph_idx[start + i]; const PhotonRec r = a.ph[start + i];
        const double g = has_gain ? a.ph_gain[start - a.gain_first + i] : 0.0;
        i32 jj = i - 1;
        while (jj >= 0 && a.ph_idx[start + jj] > key) {
            a.ph_idx[start + jj + 1] = a.ph_idx[start + jj]; a.ph[start + jj + 1] = a.ph[start + jj];
            if (has_gain) a.ph_gain[start - a.gain_first + jj + 1] = a.ph_gain[start - a.gain_first + jj];
            jj--;
        }
        a.ph_idx[start + jj + 1] = key; a.ph[start + jj + 1] = r;
        if (has_gain) a.ph_gain[start - a.gain_first + jj + 1] = g;
    }
}
__global__ __launch_bounds__(256) void k_tile_order_scan(OrderArgs a)
{
    __shared__ i32 s_n[3]; __shared__ i64 s_base[3];
    if (threadIdx.x < 3) s_n[threadIdx.x] = 0;
    __syncthreads();
    const i64 tile = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    OrderRange rg[2] = {{0, 0, 0}, {0, 0, 0}}; int cls[2] = {-1, -1}, rk[2] = {0, 0};
    if (tile < a.n_tiles) {
        const i32 n = a.tile_count[tile];
        const i64 set = tile / a.nch;
        if (n >= 2 && !(a.skip_ins && tile < a.n_ptiles && a.skip_ins[set])) {
            const i64 off = a.tile_off[tile];
            if (tile < a.n_ptiles && !a.ins_fullsort[a.set_ins_list[a.set_ins_off[set]]]) {
                rg[0] = OrderRange{off, a.tile_cursor[tile], 0};
                rg[1] = OrderRange{off + a.tile_tailbase[tile], n - a.tile_tailbase[tile], 0};
            } else rg[0] = OrderRange{off, n, 0};
            for (int q = 0; q < 2; q++) { order_range(a, rg[q].start, rg[q].n, cls[q]); if (cls[q] >= 0) rk[q] = atomicAdd(&s_n[cls[q]], 1); }
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_n[threadIdx.x]) s_base[threadIdx.x] = (i64)atomicAdd((u64 *)&a.scal[threadIdx.x < 2 ? 30 + threadIdx.x : 19], (u64)s_n[threadIdx.x]);
    __syncthreads();
    for (int q = 0; q < 2; q++) {
        if (cls[q] == 0) a.wave_list[s_base[0] + rk[q]] = rg[q];
        else if (cls[q] == 1) a.big_list[s_base[1] + rk[q]] = rg[q];
        else if (cls[q] == 2) a.big_list[2 * a.n_tiles - 1 - (s_base[2] + rk[q])] = rg[q];
    }
}
// Ranges beyond TILE_ORDER_MAX photons: their order keys, positions and records are copied to compact arrays (range k at cbeg[k]), the
// (key, position) pairs are sorted range by range (rocPRIM's segmented radix sort), and the records move to their ranks.
struct HugeOrderArgs { i64 n_ranges, total; const i64 *start; const i64 *cbeg; const PhotonRec *ph; const u32 *ph_idx; const double *ph_gain; i64 gain_first;
                       u32 *keys, *vals; PhotonRec *rec; double *gain; };
__device__ __forceinline__ i64 huge_range_of(const HugeOrderArgs &a, i64 i)
{
    i64 lo = 0, hi = a.n_ranges;
    while (hi - lo > 1) { const i64 mid = (lo + hi) >> 1; if (a.cbeg[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
__global__ void k_order_huge_pack(HugeOrderArgs a)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const i64 r = huge_range_of(a, i), p = a.start[r] + (i - a.cbeg[r]);
    a.keys[i] = a.ph_idx[p]; a.vals[i] = (u32)(i - a.cbeg[r]); a.rec[i] = a.ph[p];
    if (a.ph_gain && p >= a.gain_first) a.gain[i] = a.ph_gain[p - a.gain_first];
}
__global__ void k_order_huge_apply(HugeOrderArgs a, const u32 *keys_sorted, const u32 *vals_sorted, PhotonRec *ph, u32 *ph_idx, double *ph_gain)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.total) return;
    const i64 r = huge_range_of(a, i), p = a.start[r] + (i - a.cbeg[r]);
    const i64 src = a.cbeg[r] + vals_sorted[i];
    ph[p] = a.rec[src]; ph_idx[p] = keys_sorted[i];
    if (ph_gain && p >= a.gain_first) ph_gain[p - a.gain_first] = a.gain[src];
}
// pass 2, one wave per listed range of at most 64 photons: rank = number of smaller keys (the keys of a range are distinct)
__global__ __launch_bounds__(256) void k_tile_order(OrderArgs a, i64 n_list)
{
    const int lane = threadIdx.x & 63;
    const i64 li = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (li >= n_list) return;
    const OrderRange rg = a.wave_list[li];
    const i32 n = rg.n; const i64 off = rg.start;
    const bool v = lane < n;
    const u32 k0 = v ? a.ph_idx[off + lane] : 0xffffffffu;
    const PhotonRec r0 = a.ph[off + (v ? lane : 0)];
    const bool has_gain = a.ph_gain && off >= a.gain_first;
    const double g0 = has_gain ? a.ph_gain[off - a.gain_first + (v ? lane : 0)] : 0.0;
    int rank = 0;
    for (int j = 0; j < n; j++) rank += (u32)__builtin_amdgcn_readlane((int)k0, j) < k0;
    if (v) {
        a.ph[off + rank] = r0; a.ph_idx[off + rank] = k0;
        if (has_gain) a.ph_gain[off - a.gain_first + rank] = g0;
    }
}
// ranges of 65 .. TILE_ORDER_MAX photons: (key, position) pairs sorted in LDS (bitonic), then the records move
__global__ __launch_bounds__(256) void k_tile_order_big(OrderArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *key = (u64 *)smem;                                  // [npow2] key << 32 | position
    const OrderRange rg = a.big_list[blockIdx.x];
    const i32 n = rg.n; const i64 off = rg.start;
    int np2 = 128; while (np2 < n) np2 <<= 1;
    const int tid = threadIdx.x;
    for (int i = tid; i < np2; i += 256) key[i] = i < n ? ((u64)a.ph_idx[off + i] << 32) | (u32)i : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int l = i ^ j;
                if (l > i) {
                    const u64 x = key[i], y = key[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { key[i] = y; key[l] = x; }
                }
            }
            __syncthreads();
        }
    const bool has_gain = a.ph_gain && off >= a.gain_first;
    PhotonRec *ph = a.ph + off; u32 *idx = a.ph_idx + off; double *pg = has_gain ? a.ph_gain + (off - a.gain_first) : nullptr;
    PhotonRec *cr = (PhotonRec *)(key + np2);                 // [n] copies: every record is read before any is written
    double *cg = (double *)(cr + np2);
    for (int i = tid; i < n; i += 256) { cr[i] = ph[i]; if (has_gain) cg[i] = pg[i]; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int src = (int)(u32)key[i];
        ph[i] = cr[src]; idx[i] = (u32)(key[i] >> 32);
        if (has_gain) pg[i] = cg[src];
    }
}

// Arrival times of chosen photons, addressed by their index in GENERATION order (emitter by emitter, the order of
// em_ph_off): electron afterpulses pick random detected photons of their parent S2 as time zeros (afterpulse.py:44-47).
// The time is recomputed from the photon's own Philox draw (site B), so the choice does not depend on where the
// bucketing put the photon.
__global__ void k_photon_times(WfsDev d, GenArgs a, i64 n, const i64 *index, i64 *out)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PhotonId id = photon_id(a, index[i]);
    const i32 ins = id.ins;
    const u32 w = photon_word(d, id, SITE_DELAY);
    if (a.tabs) {                                            // model variants: the table depends on the photon's array
        const int ch = photon_channel_global(d, a, id);
        i64 t = a.em_time[id.em] + alias_sample(a.tabs[ch >= d.n_top ? a.ins_tabb[ins] : a.ins_tab[ins]], w);
        if (a.prop_top && a.ins_type[ins] == 1 && a.ins_pzi[ins] >= 0) {
            const u32x4 X = philox4x32_10(id.j, id.gid, id.m, SITE_PH_X, d.k0, d.k1);
            t += (i64)s1_propagation(a, ch >= d.n_top, a.ins_pzi[ins], a.ins_pzf[ins], X.x);
        }
        if (a.gg_inv && a.ins_gg[ins] >= 0) t += (i64)(gg_time(a, a.ins_gg[ins], a.ins_ggw[ins], photon_word(d, id, SITE_LUM)) - gg_mean(a, ins));
        out[i] = t;
        return;
    }
    out[i] = a.em_time[id.em] + alias_sample(a.ins_type[ins] != 1 ? d.tab_s2 : d.tab_s1, w);
}

// Optical input (RawDataOptical.sim_primary, rawdata.py:475-493): photons are supplied, already bucketed by the host
// side of the ABI; what is left of Pulse.__call__ before add_current is drawn here per photon: transit time spread
// (pulse.py:53-56), double-PE flag (pulse.py:76-79) and the SPE gain indices (pulse.py:97-103): one Philox call.  One thread per tile.
struct OpticalArgs { i64 n_tiles; const i32 *tile_count; const i64 *tile_off; i32 *tile_tmin, *tile_tmax; const u32 *set_gid;
                     const i32 *in_t; const u32 *in_item; PhotonRec *ph; i64 *scal; };

// Bucketing of the supplied photons by (instruction, channel) -- one thread per instruction, which owns the instruction's tiles
// (plain read-modify-write).  The host used to do this over dense per-tile arrays: 0.4 s of 0.5 s per 4 x 10^5 nVeto instructions.
// rawdata.py:485-486: photons with a negative time or beyond the cutoff are dropped; pulse.py:89-90: so are those of turned-off PMTs.
struct OptLoadArgs { i64 n; const i32 *first, *last, *channels; const i64 *timings; i64 cutoff; const double *gains;
                     i32 *tile_count; const i64 *tile_off; i32 *tile_cursor; i32 *in_t; u32 *in_item; i64 *scal; };

template <bool PLACE>
__global__ void k_optical_bucket(WfsDev d, OptLoadArgs a)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const i32 f = a.first[i], l = a.last[i];
    for (i32 k = f; k < l; k++) {
        const i64 t = a.timings[k];
        if (t < 0 || t >= a.cutoff) continue;
        const i32 ch = a.channels[k];
        if (ch < 0 || ch >= d.n_tpc) { atomicMax(&a.scal[20], (i64)1); continue; }      // photon channel out of range
        if (a.gains[ch] == 0) continue;
        if (t > 0x7ffffff0LL) { atomicMax(&a.scal[20], (i64)2); continue; }              // photon time beyond 2^31 ns
        const i64 tile = i * d.n_tpc + ch;
        if (!PLACE) a.tile_count[tile]++;
        else { const i64 pos = a.tile_off[tile] + a.tile_cursor[tile]++; a.in_t[pos] = (i32)t; a.in_item[pos] = (u32)(k - f); }
    }
}

__global__ void k_optical_finish(WfsDev d, OpticalArgs a)
{
    const i64 tile = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= a.n_tiles) return;
    const i32 n = a.tile_count[tile];
    if (n <= 0) return;
    const i64 off = a.tile_off[tile];
    const u32 gid = a.set_gid[tile / d.n_tpc];
    i32 tmin = 0x7fffffff, tmax = (i32)0x80000000;
    for (i32 p = 0; p < n; p++) {
        const u32 item = a.in_item[off + p];                 // index of the photon inside its instruction's range
        const u32x4 B = philox4x32_10(0, gid, item, SITE_PH, d.k0, d.k1);
        i64 t = a.in_t[off + p];
        t += alias_sample(d.tab_tts, B.x);
        if (t > 0x7fffffffLL || t < -0x7fffffffLL) { atomicMax(&a.scal[1], (i64)2); t = 0; }
        a.ph[off + p] = PhotonRec{(i32)t, gain_code(B.y, d.thr_dpe, d.dpe_inv)};
        tmin = (i32)t < tmin ? (i32)t : tmin; tmax = (i32)t > tmax ? (i32)t : tmax;
    }
    a.tile_tmin[tile] = tmin; a.tile_tmax[tile] = tmax;
}

