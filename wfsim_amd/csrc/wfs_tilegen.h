// wfs_tilegen.h -- tile-local photon generation fused with the pulse kernel (RNG spec v9).  Included by
// wfs_engine.hip behind wfs_kernels.h.
//
// A primary S2 whose electrons all share one secondary gain (s2_gain_spread == 0) makes Poisson(g) photons per surviving electron
// (s2.py:308) and sends each to a channel drawn from the instruction's pattern row (np.random.choice, s2.py:673).  By Poisson
// splitting the photon counts per (electron, channel) are independent Poisson(g p_ch) variates, hence
//     * the number of photons of a TILE (instruction, channel) is Poisson(n_surviving g p_ch), independent of the other tiles,
//     * given that number every photon of the tile belongs to a uniformly drawn surviving electron,
// which is the same joint distribution of (electron, channel) pairs as the reference's.  A pulse workgroup can therefore make
// its own photons: count from one Poisson draw (k_tile_counts), then per photon one word each for the electron, the summed delay
// (the alias table of every other S2 photon) and the SPE gain code.  Nothing per photon touches HBM: no channel word, no count
// pass, no bucket ranks, no 8-byte record written and read back (13.6 GB of 21 GB per headline batch went there).
//
// The tile's time range only exists once its photons do, and the window / row layout (k_tile_geom ... k_row_len) needs it:
// the kernel therefore runs BEFORE the geometry and writes its rounded ADC samples into a buffer of its own, sized from what
// is known beforehand (electron time range of the instruction + support of the delay table) and padded with trigger_window
// zeros on either side -- exactly the row a tile makes when it is alone in its (window, channel), which then needs no copy:
// ZLE and record packing read the tile buffer in place.  Tiles that share a row with others are added into the row's
// accumulators afterwards (k_tile_add).
//
// Which instructions take this path is a rule both the device and the CPU oracle evaluate (fuse_eligible): type 2, no
// emitter offset (not an electron afterpulse), s2_gain_spread == 0, default delay table, and a tile that
// fits the 2048 photon registers of a workgroup with eight standard deviations to spare -- and is worth a workgroup: at least
// wfs_config.tile_gen_min photons expected on the brightest channel (below ~60 per tile the block generator is faster).
#pragma once

#define TILE_MAX_PHOTONS 2048      // photons a pulse workgroup holds in registers (256 threads x 8)

struct __attribute__((aligned(16))) FTile {
    i64 e0;               // first compacted electron time of the instruction (index into et32)
    i64 t0;               // origin of the set's photon times (absolute ns)
    i64 boff;             // first int of the tile's sample buffer
    double G, thr;        // PMT gain, truth threshold of the channel
    i32 n, n_surv;        // photons of the tile, surviving electrons of the instruction
    i32 ch, tile;
    u32 gid, c0;          // Philox coordinates: instruction id, emitter base + channel
    i32 cap, pad;         // samples the buffer holds
};

struct FuseArgs {
    i64 n_ins; i32 nch; i32 table_span;      // cells of the S2 delay table (bound of its support)
    i32 lam_min, pad0;    // wfs_config.tile_gen_min
    i64 n_ptiles;         // primary tiles: the afterpulse tile of tile t is t + n_ptiles
    const int8_t *ins_type; const i32 *ins_amp; const double *ins_sc; const u32 *ins_embase, *ins_gid; const i32 *ins_cdfrow;
    const double *cdf_table; const double *row_pmax; const i64 *ins_time; const i64 *em_off; const i64 *em_time; const i64 *el_minmax;
    i32 *ins_fused;       // [n_ins] 1: the instruction's photons are generated tile by tile
    i32 *ins_nsurv;       // [n_ins] surviving electrons
    i32 *ins_bcap;        // [n_ins] samples reserved per tile buffer
    i32 *ins_bcap_all;    // [n_ins] n_tpc * ins_bcap (input of the scan)
    const i64 *ins_boff;  // [n_ins + 1] first int of the instruction's tile buffers
    i32 *et32;            // [n_emitters] arrival times of the surviving electrons relative to the set origin, compacted per instruction
    i32 *tile_count; const i64 *tile_off; i32 *tile_tmin, *tile_tmax; double *tile_truth;
    FTile *tiles;         // work list of the tiles with photons
    i32 *tbuf;            // tile sample buffers
    PhotonRec *ph; i32 keep_ph;      // debug: the photons are also stored, tile by tile, in generation order
    i32 sparse_max;       // occupied cells up to which a wave walks them instead of the dense gather (tap_block)
    i64 *scal;            // [1] error flag, [23] ints of all tile buffers, [24] photons of all tiles, [25] listed tiles
};

// Both sides of the parity tests evaluate this rule (oracle/wfsim_oracle.c: fuse_eligible): the same IEEE operations (a maximum does
// not depend on the order it is taken in).
__device__ __forceinline__ bool fuse_eligible(int type, u32 em_base, i32 amp, double sc, double pmax, i32 lam_min)
{
    if (type != 2 || em_base != 0u || amp <= 0 || !(sc > 0)) return false;
    const double lam = (double)amp * sc * pmax;
    return lam >= (double)lam_min && lam + 8.0 * sqrt(lam) + 8.0 <= (double)TILE_MAX_PHOTONS;
}

// largest channel probability of every pattern row: one wave per row (a thread walking a row alone pays one load latency per channel)
__global__ __launch_bounds__(256) void k_row_pmax(const double *cdf_table, int nch, i64 n_rows, double *row_pmax)
{
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_rows) return;
    const double *row = cdf_table + r * nch;
    double pmax = 0.0;
    for (int c = lane; c < nch; c += 64) { const double p = row[c] - (c ? row[c - 1] : 0.0); pmax = p > pmax ? p : pmax; }
    for (int o = 32; o > 0; o >>= 1) { const double x = __shfl_xor(pmax, o, 64); pmax = x > pmax ? x : pmax; }
    if (lane == 0) row_pmax[r] = pmax;
}

__global__ void k_fuse_decide(FuseArgs f)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f.n_ins) return;
    f.ins_fused[i] = fuse_eligible(f.ins_type[i], f.ins_embase[i], f.ins_amp[i], f.ins_sc[i], f.row_pmax[f.ins_cdfrow[i]], f.lam_min) ? 1 : 0;
}

// Surviving electrons of a tile-generated instruction, compacted in candidate order: arrival times relative to the set origin
// (the instruction's time: such an instruction is alone in its pulse set).  One workgroup per instruction.
__global__ __launch_bounds__(256) void k_fuse_electrons(WfsDev d, FuseArgs f)
{
    __shared__ i32 s_w[4];
    const i64 i = blockIdx.x; const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (!f.ins_fused[i]) return;
    const i64 e0 = f.em_off[i], e1 = f.em_off[i + 1], t0 = f.ins_time[i];
    i32 run = 0;
    for (i64 base = e0; base < e1; base += 256) {
        const i64 e = base + tid;
        const i64 et = e < e1 ? f.em_time[e] : I64_MIN;
        const bool ok = et != I64_MIN;
        const u64 m = __ballot(ok);
        if (lane == 0) s_w[wid] = __popcll(m);
        __syncthreads();
        i32 pos = run + __popcll(m & ((1ull << lane) - 1ull));
        for (int w = 0; w < wid; w++) pos += s_w[w];
        if (ok) {
            i64 rel = et - t0;
            if (rel > 0x3fffffffLL || rel < -0x3fffffffLL) { atomicMax(&f.scal[1], (i64)2); rel = 0; }      // electron further than 2^30 ns from its instruction
            f.et32[e0 + pos] = (i32)rel;
        }
        run += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (tid == 0) {
        f.ins_nsurv[i] = run;
        i32 cap = 0;
        if (run > 0) {
            // every photon lies in [first electron + table minimum, last electron + table minimum + cells): start bins, the samples
            // in front of / behind them (pulse.py:118-127), trigger_window zeros on either side (the row of a tile that is alone)
            const i64 span = f.el_minmax[2 * i + 1] - f.el_minmax[2 * i] + (i64)f.table_span;
            const i64 c = span / d.dt + 2 + d.store_before + d.samples_before + d.store_after + d.samples_after + 2 * (i64)d.tw + 4;
            if (c * f.nch > 0x7fffffffLL) atomicMax(&f.scal[1], (i64)1); else cap = (i32)c;
        }
        f.ins_bcap[i] = cap; f.ins_bcap_all[i] = cap * f.nch;
    }
}

// numpy's legacy Poisson (s2.py:308 draws np.random.poisson): PTRS (Hoermann 1993) for lam >= 10, multiplication method below;
// uniform pairs from the items of one Philox stream.  Same algorithm and uniforms as the oracle's poisson_site.
__device__ i64 poisson_site(const WfsDev &d, u32 emitter, u32 gid, u32 site, double lam)
{
    u32 it = 0;
    if (!(lam > 0)) return 0;
    if (lam < 10) {
        const double enlam = exp(-lam); double prod = 1.0; i64 x = 0;
        for (;;) {
            const u32x4 w = philox4x32_10(emitter, gid, it++, site, d.k0, d.k1);
            prod *= u53(w.x, w.y);
            if (prod > enlam) x++; else return x;
            prod *= u53(w.z, w.w);
            if (prod > enlam) x++; else return x;
        }
    }
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam, aa = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2);
    for (;;) {
        const u32x4 w = philox4x32_10(emitter, gid, it++, site, d.k0, d.k1);
        const double U = u53(w.x, w.y) - 0.5, V = u53(w.z, w.w);
        const double us = 0.5 - fabs(U);
        const i64 k = (i64)floor((2 * aa / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(aa / (us * us) + b) <= -lam + k * loglam - lgamma((double)k + 1)) return k;
    }
}

// Photons of every tile of the tile-generated instructions: Poisson(n_surviving * gain * p_channel), one thread per tile; tiles
// with photons go on the work list with everything their workgroup needs (one scalar load).
__global__ __launch_bounds__(256) void k_tile_counts(WfsDev d, FuseArgs f)
{
    __shared__ i32 s_n, s_tot; __shared__ i64 s_base;
    if (threadIdx.x == 0) { s_n = 0; s_tot = 0; }
    __syncthreads();
    const i64 idx = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    const i64 i = idx / f.nch; const int c = (int)(idx - i * f.nch);
    i64 N = 0; i32 rk = -1; FTile ft{};
    if (i < f.n_ins && f.ins_fused[i]) {
        const i32 ns = f.ins_nsurv[i];
        const double *row = f.cdf_table + (i64)f.ins_cdfrow[i] * f.nch;
        const double p = row[c] - (c ? row[c - 1] : 0.0);
        const double lam = (double)ns * f.ins_sc[i] * p;
        const u32 gid = f.ins_gid[i], c0 = f.ins_embase[i] + (u32)c;
        N = poisson_site(d, c0, gid, SITE_TILE_N, lam);
        if (N > TILE_MAX_PHOTONS) N = TILE_MAX_PHOTONS;           // (beyond eight standard deviations of the largest admitted tile)
        f.tile_count[idx] = (i32)N;                               // tile id = instruction * n_tpc + channel: one pulse set per instruction
        if (N > 0) {
            rk = atomicAdd(&s_n, 1);
            ft.e0 = f.em_off[i]; ft.t0 = f.ins_time[i]; ft.cap = f.ins_bcap[i]; ft.boff = f.ins_boff[i] + (i64)c * ft.cap;
            ft.G = d.gains[c]; ft.thr = d.thr_truth[c]; ft.n = (i32)N; ft.n_surv = ns; ft.ch = c; ft.tile = (i32)idx; ft.gid = gid; ft.c0 = c0;
        }
    }
    const int tot = wave_sum((int)N);                        // (a workgroup's 256 tiles hold at most 2^19 photons)
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&s_tot, tot);
    __syncthreads();
    if (threadIdx.x == 0 && s_tot) atomicAdd((u64 *)&f.scal[24], (u64)s_tot);      // one atomic per workgroup on each of the two counters
    if (threadIdx.x == 0 && s_n) s_base = (i64)atomicAdd((u64 *)&f.scal[25], (u64)s_n);
    __syncthreads();
    if (rk >= 0) f.tiles[s_base + rk] = ft;
}

// One workgroup per tile: photons (electron, delay, gain code: three Philox calls per quad of photons, spec v9), the tile's time
// range, and -- FULL -- everything k_pulse does for a resident tile (SPE gains, truth sums, the H-table gather in chunks of
// 256 samples, per-pulse rounding), written to the tile's own sample buffer.  !FULL: generation only (debug modes: the
// photons go to the photon array and the ordinary pulse kernels take the tile from there).
// AP: PMT afterpulses of the tile's photons (same channel, the afterpulse set of the instruction): screened here, photon by photon
// (ap_screen_mask: counter (c0, gid, q) of the parent photon q of the tile), the candidates leave for the global list that
// k_ap_finish works through (wfs_kernels.h); their photons become ordinary tiles of the afterpulse sets.
template <bool FULL, bool AP, bool FMA>
__global__ __launch_bounds__(256, 6) void k_s2_tile(WfsDev d, FuseArgs f, TemplateArg tp, const ApArgs *app, int ap_lds_off)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TPB = 256, dt = WFS_DT, tlen = 22, HROWS = TPB + tlen - 1, NPH = 8, NW = TPB / 64;
    double *H = (double *)smem;                           // [HROWS][dt]; before the chunks: the channel's SPE row, then the truth partial sums
    u32 *wsum = (u32 *)(H + (size_t)HROWS * dt);          // [4 * NW]
    double *W2 = (double *)(wsum + 4 * NW);               // [TAP_W2_LEN] taps by time difference (tap_block)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    __shared__ double s_cmax[WFS_DT];
    __shared__ i32 s_mm[2 * NW];
    if (tid < WFS_DT) s_cmax[tid] = d.current_max[tid];
    if (FULL) tap_w2_fill(W2, d.templates, tid, TPB);
    const FTile ft = f.tiles[blockIdx.x];                 // block-uniform: scalar loads
    const i32 n = ft.n;
    __shared__ i32 s_apn; __shared__ i64 s_apbase;
    if (AP) { if (tid == 0) s_apn = 0; __syncthreads(); }
    const double *spe_row = d.spe + (size_t)(d.n_spe > 1 ? ft.ch : 0) * 2001;
    STAMP_INIT;
    if (FULL) {                                           // the SPE row (16 KB) on its way to LDS under the generation (see k_pulse)
        constexpr int NIT = (2001 * 8 + TPB * 16 - 1) / (TPB * 16);
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = (it * TPB + tid) * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(spe_row + (i < 2001 ? i : 2000)),
                                             (__attribute__((address_space(3))) void *)(H + (it * TPB + (tid & ~63)) * 2), 16, 0, 0);
        }
    }
    // ---- generation: thread tid owns the quads tid and tid + TPB, i.e. the photons P = 4 q .. 4 q + 3 (register k = 4 r + j)
    const AliasTab tab = d.tab_s2;
    u32 eidx[NPH], dw[NPH], cell[NPH], code[NPH];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const u32 q = (u32)(tid + r * TPB);
#pragma unroll
        for (int j = 0; j < 4; j++) { eidx[4 * r + j] = 0; dw[4 * r + j] = 0; cell[4 * r + j] = 0; code[4 * r + j] = 0; }
        if (__any((i32)(4 * q) < n)) {                    // wave-uniform: the second round is empty for most waves of a 1500-photon tile
            const u32x4 E = philox4x32_10(ft.c0, ft.gid, q, SITE_TILE_E, d.k0, d.k1);
            const u32x4 D = philox4x32_10(ft.c0, ft.gid, q, SITE_TILE_DELAY, d.k0, d.k1);
            const u32x4 G = philox4x32_10(ft.c0, ft.gid, q, SITE_TILE_GAIN, d.k0, d.k1);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 4 * r + j;
                eidx[k] = (u32)(((u64)word_of(E, j) * (u64)(u32)ft.n_surv) >> 32);      // uniform surviving electron
                dw[k] = word_of(D, j); cell[k] = alias_cell(tab, dw[k]);
                code[k] = gain_code(word_of(G, j), d.thr_dpe, d.dpe_inv);
            }
        }
    }
    i32 et[NPH]; uint2 ce[NPH];                           // all sixteen gathers in flight together
#pragma unroll
    for (int k = 0; k < NPH; k++) { et[k] = f.et32[ft.e0 + eidx[k]]; ce[k] = tab.cell[cell[k]]; }
    i32 r_t[NPH]; i32 tmin = 0x7fffffff, tmax = (i32)0x80000000; i32 ndpe = 0;
#pragma unroll
    for (int k = 0; k < NPH; k++) {
        const i32 P = 4 * (tid + (k >> 2) * TPB) + (k & 3);
        const bool v = P < n;
        const i32 t = et[k] + alias_pick(tab, dw[k], cell[k], ce[k]);
        r_t[k] = t;
        if (v) { tmin = t < tmin ? t : tmin; tmax = t > tmax ? t : tmax; ndpe += (code[k] >> 16) != 0; }
        else code[k] = 0;
    }
    if constexpr (AP) {
        const ApArgs &ap = *app;
        ApStage aps; aps.n = &s_apn; aps.cand = (ApCand *)(smem + ap_lds_off);
        const i32 set = ft.tile / f.nch;
        // the tile's thresholds (one channel: two per element) once, in scalar registers: a dependent pair of global loads per photon and
        // element otherwise, in the middle of every tile's chain
        const int n_ap = ap.n;
        u32 thr_s[WFS_MAX_AP], thr_d[WFS_MAX_AP];
#pragma unroll
        for (int e = 0; e < WFS_MAX_AP; e++) { thr_s[e] = 0xffffffffu; thr_d[e] = 0xffffffffu; if (e < n_ap) { thr_s[e] = ap.thr[e][ft.ch * 2]; thr_d[e] = ap.thr[e][ft.ch * 2 + 1]; } }
        u64 cm = 0;                                       // bit 8 k + e: element e of photon k is a candidate
#pragma unroll
        for (int k = 0; k < NPH; k++) {
            const i32 P = 4 * (tid + (k >> 2) * TPB) + (k & 3);
            if (P >= n) continue;
            const bool dpe = (code[k] >> 16) != 0;
            u32 m = 0;
#pragma unroll
            for (int e0 = 0; e0 < WFS_MAX_AP; e0 += 4) {
                if (e0 >= n_ap) break;                     // block-uniform
                const u32x4 S = philox4x32_10(ft.c0, ft.gid, (u32)P, SITE_AP_SCREEN + (u32)(e0 >> 2), d.k0, d.k1);
#pragma unroll
                for (int j = 0; j < 4; j++) if ((word_of(S, j) >> 5) >= (dpe ? thr_d[e0 + j] : thr_s[e0 + j])) m |= 1u << (e0 + j);      // (elements past n_ap: threshold 2^32 - 1, never met)
            }
            cm |= (u64)m << (8 * k);
        }
        while (cm) {                                      // (one photon in a hundred: the parking code once, not eight times)
            const int b = __builtin_ctzll(cm); cm &= cm - 1;
            const int k = b >> 3, e = b & 7;
            i32 tk = r_t[0]; u32 ck = code[0];
#pragma unroll
            for (int q = 1; q < NPH; q++) { tk = k == q ? r_t[q] : tk; ck = k == q ? code[q] : ck; }
            const u32 P = (u32)(4 * (tid + (k >> 2) * TPB) + (k & 3));
            ap_park(ap, aps, ap_screen_word(d, ft.c0, ft.gid, P, e), e | ((ck >> 16) ? 256 : 0), ft.c0, ft.gid, P, set, ft.ch, ft.t0, tk, ((u32)e << 29) | (P & 0x1fffffffu), true);
        }
    }
    tmin = wave_min(tmin); tmax = wave_max(tmax); ndpe = wave_sum(ndpe);
    if (lane == 0) { s_mm[2 * wid] = tmin; s_mm[2 * wid + 1] = tmax; wsum[wid] = (u32)ndpe; }
    if (FULL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the SPE row has landed in LDS
    __syncthreads();
    i32 n_dpe_tile = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) { tmin = s_mm[2 * w] < tmin ? s_mm[2 * w] : tmin; tmax = s_mm[2 * w + 1] > tmax ? s_mm[2 * w + 1] : tmax; n_dpe_tile += (i32)wsum[w]; }
    if (tid == 0) { f.tile_tmin[ft.tile] = tmin; f.tile_tmax[ft.tile] = tmax; }
    if constexpr (AP) {                                   // the block's candidates to the global list (all parked: the barrier above)
        const ApArgs &ap = *app;
        const ApCand *cand = (const ApCand *)(smem + ap_lds_off);
        const int nst = s_apn < AP_STAGE ? s_apn : AP_STAGE;
        if (tid == 0 && nst > 0) s_apbase = (i64)atomicAdd((u64 *)ap.count, (u64)nst);
        __syncthreads();
        if (tid == 0) ap.seg[blockIdx.x] = ApSeg{nst > 0 ? s_apbase : 0, nst, ft.tile + (i32)f.n_ptiles};     // the tile's stretch of the list (k_ap_seg)
        // in key order (element, parent photon): the tile's afterpulses then reach their tile in generation order (k_ap_place)
        for (int k = tid; k < nst; k += TPB) {
            const ApCand q = cand[k];
            int rank = 0;
            for (int j = 0; j < nst; j++) rank += cand[j].key < q.key;
            const i64 gk = s_apbase + rank; if (gk < ap.cap) ap.cand[gk] = q;
        }
    }
    if (!FULL || f.keep_ph) {
        PhotonRec *out = f.ph + f.tile_off[ft.tile];
#pragma unroll
        for (int k = 0; k < NPH; k++) { const i32 P = 4 * (tid + (k >> 2) * TPB) + (k & 3); if (P < n) out[P] = PhotonRec{r_t[k], code[k]}; }
    }
    if (!FULL) return;
    STAMP(d, 16);

    // ---- the tile (pulse.py:118-127): start bins, samples, ns relative to the first start bin
    const i64 bin0 = floordiv(ft.t0 + tmin, (i64)dt), bin1 = floordiv(ft.t0 + tmax, (i64)dt);
    const i64 nb = bin1 - bin0 + 1;
    const i32 rel0 = (i32)(bin0 * dt - ft.t0);
    const int lead = d.store_before + d.samples_before;
    const i64 L = nb + lead + d.store_after + d.samples_after;
    const i64 n_live = nb + (tlen - 1);
    if (nb <= 0 || L + 2 * (i64)d.tw > (i64)ft.cap) { if (tid == 0) atomicMax(&f.scal[1], (i64)3); return; }      // (cannot happen: the buffer was sized from the bounds)
    i32 *tb = f.tbuf + ft.boff;                           // [tw zeros][L samples][tw zeros]
    const double G = ft.G, thr = ft.thr;
    i32 r_ns[NPH]; double r_gain[NPH];
#pragma unroll
    for (int k = 0; k < NPH; k++) { const i32 P = 4 * (tid + (k >> 2) * TPB) + (k & 3); r_ns[k] = P < n ? r_t[k] - rel0 : -1; }
    // SPE gains from the row in LDS (pulse.py:97-103), in two halves (registers: see k_pulse)
#pragma unroll
    for (int h0 = 0; h0 < NPH; h0 += NPH / 2) {
        double s1[NPH / 2], s2[NPH / 2];
#pragma unroll
        for (int k = 0; k < NPH / 2; k++) { s1[k] = H[code[h0 + k] & 0xffffu]; s2[k] = H[code[h0 + k] >> 16]; }
#pragma unroll
        for (int k = 0; k < NPH / 2; k++) {
            double gk = G * s1[k];
            if (code[h0 + k] >> 16) gk += G * s2[k];
            r_gain[h0 + k] = r_ns[h0 + k] >= 0 ? gk : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                      // H is reused below
    STAMP(d, 17);
    // ---- truth sums of the tile (pulse.py:229-271); photons in their order in the channel slice = generation order P
    {
        u32 c_trig = 0, c_trig_dpe = 0;
        double sg = 0, sgt = 0, st = 0, st2 = 0;
#pragma unroll
        for (int k = 0; k < NPH; k++) {
            const bool v = r_ns[k] >= 0;
            const int r = v ? r_ns[k] % dt : 0;
            const bool above = v && (r_gain[k] * s_cmax[r] * d.c2a > thr);
            const i32 P = 4 * (tid + (k >> 2) * TPB) + (k & 3);
            c_trig += (u32)__popcll(__ballot(above));
            c_trig_dpe += (u32)__popcll(__ballot(above && P < n_dpe_tile));
            if (v) {
                sg += r_gain[k];
                if (above) sgt += r_gain[k];
                const double tr = (double)(r_ns[k] + rel0);
                st += tr; st2 += tr * tr;
            }
        }
        if (f.tile_truth) {
            // wave sums on the DPP path, then the four waves in a fixed order
            sg = wave_sum(sg); sgt = wave_sum(sgt); st = wave_sum(st); st2 = wave_sum(st2);
            double *S = H;                                 // [NW][4]
            if (lane == 0) { S[wid * 4 + 0] = sg; S[wid * 4 + 1] = sgt; S[wid * 4 + 2] = st; S[wid * 4 + 3] = st2; wsum[NW + wid * 2] = c_trig; wsum[NW + wid * 2 + 1] = c_trig_dpe; }
            __syncthreads();
            if (tid < 4) f.tile_truth[(i64)ft.tile * 8 + 4 + tid] = (S[0 * 4 + tid] + S[1 * 4 + tid]) + (S[2 * 4 + tid] + S[3 * 4 + tid]);
            if (tid == 0) {
                u32 t2 = 0, t3 = 0;
                for (int w = 0; w < NW; w++) { t2 += wsum[NW + w * 2]; t3 += wsum[NW + w * 2 + 1]; }
                double *o = f.tile_truth + (i64)ft.tile * 8;
                o[0] = (double)n; o[1] = (double)n_dpe_tile; o[2] = (double)t2; o[3] = (double)t3;
            }
        }
    }
    STAMP(d, 18);
    // samples nothing can reach: the padding and the samples before / behind the live range
    {
        const i32 live0 = d.tw + lead, live1 = live0 + (i32)n_live, tot = (i32)L + 2 * d.tw;
        for (i32 i = tid; i < live0; i += TPB) tb[i] = 0;
        for (i32 i = live1 + tid; i < tot; i += TPB) tb[i] = 0;
    }
    for (i64 c0 = 0; c0 < n_live; c0 += TPB) {
        const i64 b_lo = c0 - (tlen - 1);                  // start bin of H row 0
        __syncthreads();
        // (the rows this chunk's live samples can see: nothing behind them is read, tap_block is told)
        const i64 rows = n_live - c0 + (tlen - 1);
        const int ncell = (int)(rows < HROWS ? rows : HROWS) * dt;
        for (int i = tid; i < ncell; i += TPB) H[i] = 0.0;
        __syncthreads();
        STAMP(d, 19);
        const i32 ns_lo = (i32)b_lo * dt, ns_hi = ns_lo + HROWS * dt;
#pragma unroll
        for (int k = 0; k < NPH; k++) {
            if (r_ns[k] < 0 || r_ns[k] < ns_lo || r_ns[k] >= ns_hi) continue;
            atomicAdd(&H[r_ns[k] - ns_lo], r_gain[k]);
        }
        __syncthreads();
        STAMP(d, 20);
        const bool act = c0 + tid < n_live;
        if (__any(act)) {                                  // wave-uniform
            const double c = tap_block<FMA>(H, W2, tp, tid, f.sparse_max, ncell);
            if (act) tb[d.tw + lead + c0 + tid] = (i32)(-(i64)rint(c * d.c2a));        // rawdata.py:236, np.around = round half to even
        }
        STAMP(d, 21);
    }
}

// Tiles that are not alone in their (window, channel) row: their samples are added into the row's accumulators like any other pulse
// (rawdata.py:231-239).  One workgroup per listed tile; tiles whose row reads the tile buffer in place leave at once.
struct TileAddArgs { const i32 *set_cluster; const i32 *cl_group; const i64 *row_lo; const i64 *acc_off; const i32 *row_cnt; i32 *raw; };
__global__ __launch_bounds__(256) void k_tile_add(WfsDev d, FuseArgs f, TileAddArgs a)
{
    const FTile ft = f.tiles[blockIdx.x];
    const i64 set = ft.tile / f.nch;
    const i64 ridx = (i64)a.cl_group[a.set_cluster[set]] * f.nch + ft.ch;
    if (a.row_cnt[ridx] == 1) return;                      // block-uniform
    i64 left, right, bin0, nb;
    tile_bounds(d, ft.t0, f.tile_tmin[ft.tile], f.tile_tmax[ft.tile], left, right, bin0, nb);
    const i32 L = (i32)(right - left + 1);
    const i32 *src = f.tbuf + ft.boff + d.tw;
    i32 *dst = a.raw + a.acc_off[ridx] + (left - (a.row_lo[ridx] - d.tw));
    for (i32 s = threadIdx.x; s < L; s += 256) { const i32 v = src[s]; if (v != 0) atomicAdd(&dst[s], v); }
}
