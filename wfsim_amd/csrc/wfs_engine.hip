// wfs_engine.hip -- host side of libwfsim_amd.so: the C ABI of include/wfsim_amd.h, device memory arenas,
// stage orchestration on one HIP stream.  gfx950 (MI355X) only.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include "wfs_kernels.h"
#include "wfs_tilegen.h"
#include "../../include/wfsim_amd.h"

// Every pulse kernel exists in two forms: FMA = true (wfs_config.fma, one rounding per template * gain term) and FMA = false (numpy's two
// roundings, currents bit-exact with the reference).  The handle's switch picks one at launch time.
#define K_S2_TILE_FULL(F) k_s2_tile<true, false, F>
#define K_S2_TILE_FULL_AP(F) k_s2_tile<true, true, F>
#define K_S2_TILE_GEN(F) k_s2_tile<false, false, F>
#define K_S2_TILE_GEN_AP(F) k_s2_tile<false, true, F>
#define K_PULSE_256_RES(F) k_pulse<256, true, F>
#define K_PULSE_128_RES(F) k_pulse<128, true, F>
#define K_PULSE_256_WIN(F) k_pulse<256, false, F>
#define K_PULSE_128_WIN(F) k_pulse<128, false, F>
#define K_PULSE_GENERIC(F) k_pulse_generic<256, F>
#define K_PULSE_SPARSE_64(F) k_pulse_sparse<64, F>
#define K_PULSE_SPARSE_256(F) k_pulse_sparse<256, F>
#define K_PULSE_TINY(F) k_pulse_tiny<F>
#define K_PULSE_WAVE(F) k_pulse_wave<F>
#define WFS_LAUNCH_F(h, KM, grid, block, lds, ...) do { \
    if ((h)->cfg.fma) hipLaunchKernelGGL(HIP_KERNEL_NAME(KM(true)), grid, block, lds, (h)->stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(KM(false)), grid, block, lds, (h)->stream, __VA_ARGS__); } while (0)

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

namespace {

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    template <class T> T *as() const { return (T *)p; }
};
struct Pmf { std::vector<double> p; long vmin = 0; };      // probability mass function of an integer delay term

struct ApElem { int n_bins_delay = 0, n_bins_amp = 0, amp_2d = 0, is_uniform = 0; double delay_bin = 0, amp_bin = 0; DevBuf delay_cdf, amp_cdf, prob, thr, delay_guide, amp_guide;
                std::vector<double> prob_h; double thr_mod = -1.0;      // host copy of the probability column; modifier the thresholds were built for
                int delay_sorted = 0, amp_sorted = 0; };

struct KernelTime { std::string name; hipEvent_t a, b; };

}  // namespace

struct wfs_handle {
    wfs_config cfg;
    int device = 0, n_cus = 256;
    hipStream_t stream = nullptr; bool own_stream = false;
    std::string err;
    WfsDev dev;
    bool tables_set = false, batch_loaded = false, injected = false, optical = false, ran = false;
    DevBuf set_gid, opt_t, opt_item, opt_first, opt_last, opt_ch, opt_time;
    int keep_currents = 0, profiling = 0;
    i32 res_env = -1;                // WFS_ROW_RESIDENT=0/1 overrides the config switch (A/B runs)
    i32 res_max_len = 1024;          // segment of a resident row (k_row_pulse: 4 bytes of LDS per sample and wave; longer rows take several); WFS_RES_MAX_LEN overrides
    bool gen_done = false;
    bool gen_order_ready = false;          // host tables of the generation order (gen_order_tables), valid for the current generation
    std::vector<i64> go_off, go_block0; std::vector<i32> go_fused, go_tile_count;
    GenArgs gen_args{};           // the generator's view of the batch (kept for wfs_gather_photon_times)
    int carry_has = 0; i64 carry_runmax = 0;
    // tables
    DevBuf t_templates, t_spe, t_gains, t_thr_truth, t_thr_zle, t_lumx, t_lumt, t_noise, t_noise_f;
    ApElem ap[WFS_MAX_AP];
    // instructions
    i64 n_ins = 0, n_psets = 0, n_sets = 0, n_clusters = 0, n_emitters = 0, n_photons = 0, n_tiles = 0;
    DevBuf ins_type, ins_time, ins_amp, ins_gid, ins_p, ins_dm, ins_ds, ins_sc, ins_cdfrow, cdf_table, cdf_guide, chan_alias, em_off, em_zg, ins_embase, ins_set, set_ins_off, set_ins_list;
    DevBuf set_cluster, set_t0, set_mode, cl_tmin, cl_gid, cl_end, cl_group;
    DevBuf em_time, em_nph, em_ins, em_ph_off, el_stat, el_minmax, blk_e, blk_base, blk_cnt, blk_ins, eblk_ins, ins_ph0, blk_desc;
    DevBuf tile_count, tile_off, tile_cursor, tile_tmin, tile_tmax, active_tiles, sparse_tiles, dense_tiles, wave_tiles;
    DevBuf ph, ph_gain, ph_idx, ap_key, order_list, order_list2, ins_sbase, tile_tail, tile_tailbase, ins_fullsort;
    DevBuf grp_lo, grp_hi, grp_left, grp_right, grp_ixrand, grp_gid;
    DevBuf row_lo, row_hi, acc_len, acc_off, itv_cap, itv_off, active_rows, raw;
    DevBuf itv_left, itv_right, itv_n, row_nrec, rec_off;
    // packed records: two arenas used in turn, so that the device -> host copy of one batch (copy stream, pinned host memory)
    // runs under the kernels of the next batch
    DevBuf records_ab[2]; int rec_cur = 0; hipStream_t copy_stream = nullptr; hipEvent_t rec_copied[2] = {nullptr, nullptr}; bool rec_pending[2] = {false, false};
    DevBuf &records_buf() { return records_ab[rec_cur]; }
    DevBuf truth, tminmax, tile_truth, tile_desc, gather_idx, gather_out, currents, cur_len, cur_off, row_dbg, row_dbg_len, row_dbg_off;
    DevBuf stamps;
    DevBuf row_desc, rec_key, rec_key2, rec_val, rec_val2, rec_dest, sort_tmp, scan_tmp, scal, noise_override;
    bool sort_records = false; i64 n_noise_override = 0;
    // host mirrors
    std::vector<i64> h_set_off;       // injected photons: per set photon offsets (channel sorted input order)
    wfs_counts counts{};
    i64 *h_scal = nullptr;            // [64] host copy of scal in page-locked memory: read_scal's copy needs no staging buffer (five of them per batch)
    DevBuf pack_desc;
    DevBuf row_bad, fin_len, res_cnt, fin_off, res_toff, res_desc, fin, res_long, res_rows;      // resident rows (k_row_pulse)
    i64 n_front_rows = 0, n_res_rows = 0, n_short_rows = 0, n_res_tiles = 0, max_res_len = 0, s_fin = 0, s_res = 0; bool res_on = false;
    i64 n_active_tiles = 0, n_tiny_tiles = 0, n_sparse_tiles = 0, n_dense_tiles = 0, n_wave_tiles = 0, max_nb_dense = 0, n_active_rows = 0, n_groups = 0, s_raw = 0, n_itv_slots = 0, n_records = 0, max_nb = 0, max_tile = 0, max_tile_dense = 0;
    i64 cur_total = 0, row_dbg_total = 0;
    std::vector<KernelTime> times;
    std::vector<double> h_templates;      // [dt][tlen]
    bool generic_geom = false;            // sample_duration / template length other than 10 ns / 22 samples: k_pulse_generic for every tile
    std::vector<double> h_gains;
    i64 zero64 = 0;
    DevBuf tt_alias[6];
    std::vector<i64> h_rs_off; std::vector<i32> h_rs_list;        // run set -> instructions (host copy)
    std::vector<double> h_lum_x, h_lum_t;          // luminescence table (host copy, enters the S2 delay table)
    DevBuf ap_ins, ap_ch, ap_t, ap_gain, ap_cand, ap_args_dev, ap_seg; i64 n_ap_photons = 0; bool ap_active = false;
    // model variants of the photon delays
    std::vector<DevBuf> x_alias; std::vector<AliasTab> h_tabs; DevBuf d_tabs; i32 n_user_tabs = 0;
    std::vector<Pmf> base_pmf;           // transit time only, S1 terms, S2 terms, S2 terms without the 'simple' luminescence
    DevBuf prop_top, prop_bot; i32 prop_nz = 0, prop_nu = 0; double prop_u0 = 0, prop_du = 1;
    DevBuf ins_tab, ins_tabb, ins_pzi, ins_pzf; bool ins_models = false;
    DevBuf pois_cdf, pois_kmin; bool any_ptrs = false;         // Poisson tables of the secondary gain per instruction (k_poisson_tables)
    DevBuf gg_inv, ins_gg, ins_ggw, ins_ggsum; i32 gg_n = 0, gg_L = 0; bool ins_gg_set = false;       // 'garfield_gas_gap' luminescence
    // pattern maps evaluated on the device
    struct PatternMap { bool set = false; i32 dims = 0, n[3] = {1, 1, 1}, w[3] = {0, 0, 0}, n_map_ch = 0; double lo[3] = {0, 0, 0}, hgrid[3] = {1, 1, 1}; DevBuf values;
                        i64 n_points = 0; DevBuf points;
                        DevBuf cell_start, cell_pts; i32 cnx = 0, cny = 0; double cell_lo[2] = {0, 0}, cell_h = 1; } pmap[2];       // 2-D point lists: cell index (transverse diffusion)
    // scalar maps (kind 0: weighted nearest neighbours on a regular grid, 1: on a point list, 2: RectBivariateSpline)
    struct ScalarMap { int kind = 0; int nv = 1; PatternMap g; i32 nx = 0, ny = 0, kx = 0, ky = 0; DevBuf tx, ty, c; };      // kind 0 / 1: nearest neighbours on a grid / a point list, 2: spline, 3: multilinear grid
    std::vector<std::unique_ptr<ScalarMap>> smaps;
    DevBuf smap_pos, smap_out, smap_nb_idx, smap_nb_w, ins_aft; bool ins_aft_set = false;
    // transverse diffusion with field maps: instructions whose pattern is averaged over their electrons (k_diffuse_patterns)
    DevBuf ins_sigr, ins_siga, diff_row_ins, diff_row_id, diff_pre; std::vector<uint8_t> ins_diff; i64 n_diff_rows = 0; double diff_r2 = 0;
    std::vector<i32> dev_row_ins; std::vector<int8_t> h_ins_type; i64 n_host_rows = 0; bool dev_rows_pending = false;
    DevBuf map_row_ins[2], map_row_id[2], map_x, map_y, map_z, map_nb_idx[2], map_nb_w[2];

    // tile-local generation (wfs_tilegen.h): S2 instructions whose photons are made inside the pulse workgroup
    DevBuf huge_start, huge_cbeg, huge_keys, huge_keys2, huge_vals, huge_vals2, huge_rec, huge_gain;      // ordering of tiles beyond TILE_ORDER_MAX photons
    DevBuf row_pmax, ins_fused, ins_nsurv, ins_bcap, ins_bcap_all, ins_boff, et32, ftiles, tbuf, row_cnt, row_tile, tile_done;
    bool fuse_on = false, fuse_full = false, run_sets_given = false, sets_aligned = true, any_s2 = false;
    i64 n_fused_tiles = 0 /* made by k_s2_tile<FULL> */, n_gen_tiles = 0 /* tile-generated, pulse by the ordinary kernels */, p_fused = 0, s_raw_direct = 0;
    int tap_sparse_max = 48;     // tap_block: occupied cells up to which a wave of the dense pulse kernels walks them (WFS_TAP_SPARSE_MAX)
    FuseArgs fuse_args{};

    std::string launch_err;      // first failed launch of the current call (wfs_set_debug bit 3: every launch is checked)
    int fail(int code, const std::string &msg) { err = msg; return code; }
};

namespace {

#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    return h->fail(WFS_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } } while (0)

// Screening threshold of the afterpulse generator for one (element, channel, single / double PE parent): the uniform of the
// acceptance test is rU0 = 1 - u53(x, y) (afterpulse.py:196), accepted when rU0 / modifier [/ 2] <= prob.  With a = x >> 5 (the top
// 27 bits of u53) rU0 > 1 - (a + 1) 2^-27, so a < floor((1 - prob * modifier * k * (1 + 1e-9)) 2^27) - 2 cannot be accepted whatever
// the low bits and the two roundings of the divisions are.  Everything else is a candidate and gets the reference's comparison.
static u32 ap_threshold(double prob, double modifier, bool dpe)
{
    if (modifier == 0.0) return 134217728u;                        // rU0 / 0 = inf <= prob never holds (afterpulse.py:198): no photon is a candidate
    if (!(modifier > 0.0) || !(prob == prob)) return 0u;           // (negative / NaN: nothing screened, the exact comparison decides)
    const double pm = prob * modifier * (dpe ? 2.0 : 1.0) * (1.0 + 1e-9);
    if (!(pm < 1.0)) return 0u;
    const double a = std::floor((1.0 - pm) * 134217728.0) - 2.0;
    return a <= 0.0 ? 0u : (a >= 134217728.0 ? 134217728u : (u32)a);
}

int ensure(wfs_handle *h, DevBuf &b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return WFS_OK;
    if (b.p) HIPCHK(hipFree(b.p));
    size_t want = bytes + bytes / 8 + 256;
    b.p = nullptr; b.cap = 0;
    HIPCHK(hipMalloc(&b.p, want));
    b.cap = want;
    return WFS_OK;
}

int upload(wfs_handle *h, DevBuf &b, const void *src, size_t bytes)
{
    int rc = ensure(h, b, bytes); if (rc) return rc;
    if (bytes) HIPCHK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
    return WFS_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// Brackets one kernel launch: HIP events when profiling is on; with wfs_set_debug bit 3 the launch is checked on the spot
// (hipGetLastError + a stream synchronisation), so that a failed launch or a faulting kernel is reported under its own name
// instead of ~25 launches later at the end of wfs_run.
struct Timer {
    wfs_handle *h; bool on; const char *name;
    Timer(wfs_handle *h_, const char *name_) : h(h_), on(h_->profiling != 0), name(name_)
    {
        if (!on) return;
        KernelTime kt; kt.name = name;
        hipEventCreate(&kt.a); hipEventCreate(&kt.b);
        hipEventRecord(kt.a, h->stream);
        h->times.push_back(kt);
    }
    ~Timer()
    {
        if (on) hipEventRecord(h->times.back().b, h->stream);
        if (h->keep_currents & 8) {
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess && h->launch_err.empty()) { try { h->launch_err = std::string(name) + ": " + hipGetErrorString(e); } catch (...) {} }
        }
    }
};
#define CHECK_LAUNCHES() do { if (!h->launch_err.empty()) { const std::string m_ = h->launch_err; h->launch_err.clear(); return h->fail(WFS_E_HIP, m_); } } while (0)

// grid size for n items; never 0 (a launch with an empty grid is an error, every kernel bounds-checks its index)
inline unsigned nblocks(i64 n, int tpb) { return n > 0 ? (unsigned)((n + tpb - 1) / tpb) : 1u; }

int fill64(wfs_handle *h, DevBuf &b, i64 n, i64 v)
{
    TRY(ensure(h, b, (size_t)n * 8));
    if (n) { Timer t(h, "k_fill_i64"); hipLaunchKernelGGL(k_fill_i64, dim3(nblocks(n, 256)), dim3(256), 0, h->stream, b.as<i64>(), n, v); }
    return WFS_OK;
}
int fill32(wfs_handle *h, DevBuf &b, i64 n, i32 v)
{
    TRY(ensure(h, b, (size_t)n * 4));
    if (n) { Timer t(h, "k_fill_i32"); hipLaunchKernelGGL(k_fill_i32, dim3(nblocks(n, 256)), dim3(256), 0, h->stream, b.as<i32>(), n, v); }
    return WFS_OK;
}

// exclusive scan i32[n] -> i64[n+1]; total written to scal[slot]
int scan_into(wfs_handle *h, const i32 *in, i64 n, i64 *outp, int scal_slot, i64 offset);

int scan(wfs_handle *h, const i32 *in, i64 n, DevBuf &out, int scal_slot)
{
    TRY(ensure(h, out, (size_t)(n + 1) * 8));
    return scan_into(h, in, n, out.as<i64>(), scal_slot, 0);
}

// exclusive scan into a caller-provided range, every output shifted by offset
int scan_into(wfs_handle *h, const i32 *in, i64 n, i64 *outp, int scal_slot, i64 offset)
{
    i64 nb = (n + SCAN_TILE - 1) / SCAN_TILE; if (nb < 1) nb = 1;
    TRY(ensure(h, h->scan_tmp, (size_t)nb * 8));
    i64 *total = h->scal.as<i64>() + scal_slot;
    if (n == 0) { h->zero64 = offset; HIPCHK(hipMemcpyAsync(outp, &h->zero64, 8, hipMemcpyHostToDevice, h->stream)); HIPCHK(hipMemsetAsync(total, 0, 8, h->stream)); return WFS_OK; }
    { Timer t(h, "k_scan_reduce"); hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_TPB), 0, h->stream, in, n, h->scan_tmp.as<i64>()); }
    { Timer t(h, "k_scan_spine"); hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(1024), 0, h->stream, h->scan_tmp.as<i64>(), nb, total); }
    { Timer t(h, "k_scan_down"); hipLaunchKernelGGL(k_scan_down, dim3((unsigned)nb), dim3(SCAN_TPB), 0, h->stream, in, n, h->scan_tmp.as<i64>(), outp, offset); }
    return WFS_OK;
}

int read_scal(wfs_handle *h)
{
    HIPCHK(hipMemcpyAsync(h->h_scal, h->scal.p, 512, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!h->launch_err.empty()) { const std::string m = h->launch_err; h->launch_err.clear(); return h->fail(WFS_E_HIP, m); }     // (wfs_set_debug bit 3)
    return WFS_OK;
}

// cum[i] = P(trunc(Y) <= vmin + i) for Y ~ N(mu, sigma) (trunc = C cast, toward zero)
static void normal_trunc_table(double mu, double sigma, std::vector<double> &cum, int &vmin)
{
    cum.clear();
    if (!(sigma > 0)) { vmin = (int)mu; cum.push_back(1.0); return; }
    const int lo = (int)floor(mu - 8.5 * sigma) - 1, hi = (int)ceil(mu + 8.5 * sigma) + 1;
    vmin = lo;
    for (int k = lo; k <= hi; k++) {
        const double x = ((k >= 0 ? k + 1 : k) - mu) / sigma;
        cum.push_back(0.5 * erfc(-x / 1.4142135623730951));
    }
    cum.back() = 1.0;
}

// Walker alias table of a discrete distribution given by its cumulative probabilities (Vose's construction, sequential and
// in a fixed order so that the CPU oracle -- which builds its own -- arrives at the same cells): K = 2^k >= max(n, 2) cells,
// q[i] = p[i] * K; "small" (q < 1) and "large" cells on two stacks filled in ascending index order; pop a small s, pair it
// with the large l on top: cell s = {thr = floor(q[s] * 2^32), alias = l}, q[l] = (q[l] + q[s]) - 1, l moves to the small
// stack when it drops below 1; what is left over keeps its own outcome.  Sampling: wfs_device.h alias_sample.
static void build_alias(const std::vector<double> &cum, std::vector<uint2> &cell, int &shift)
{
    const size_t n = cum.size();
    size_t K = 2; int lg = 1;
    while (K < n) { K <<= 1; lg++; }
    shift = 32 - lg;
    std::vector<double> q(K, 0.0);
    for (size_t i = 0; i < n; i++) q[i] = (cum[i] - (i ? cum[i - 1] : 0.0)) * (double)K;
    std::vector<u32> small, large; small.reserve(K); large.reserve(K);
    for (size_t i = 0; i < K; i++) (q[i] < 1.0 ? small : large).push_back((u32)i);
    cell.assign(K, uint2{0xffffffffu, 0u});
    for (size_t i = 0; i < K; i++) cell[i].y = (u32)i;                       // default: own outcome with certainty
    while (!small.empty() && !large.empty()) {
        const u32 sidx = small.back(); small.pop_back();
        const u32 l = large.back();
        const double t = q[sidx] * 4294967296.0;
        cell[sidx].x = t >= 4294967295.0 ? 0xffffffffu : (u32)t; cell[sidx].y = l;
        q[l] = (q[l] + q[sidx]) - 1.0;
        if (q[l] < 1.0) { large.pop_back(); small.push_back(l); }
    }
}

static int upload_disc(wfs_handle *h, DevBuf &buf, const std::vector<double> &cum, int vmin, AliasTab &out)
{
    if (cum.size() > 65000) return h->fail(WFS_E_CAPACITY, "delay table too long (time constant above ~1.5 us)");
    std::vector<uint2> cell; int shift = 31;
    build_alias(cum, cell, shift);
    TRY(upload(h, buf, cell.data(), cell.size() * sizeof(uint2)));
    HIPCHK(hipStreamSynchronize(h->stream));
    out.cell = buf.as<uint2>(); out.vmin = vmin; out.shift = shift;
    return WFS_OK;
}

// ---- delay of a photon = sum of independent, separately truncated terms: sampled from ONE uniform through
// the convolution of the terms' probability mass functions (same distribution as adding separately drawn terms)

static Pmf pmf_from_cum(const std::vector<double> &cum, long vmin)
{
    Pmf r; r.vmin = vmin; r.p.resize(cum.size());
    for (size_t i = 0; i < cum.size(); i++) r.p[i] = cum[i] - (i ? cum[i - 1] : 0.0);
    return r;
}
static Pmf pmf_normal(double mu, double sigma) { std::vector<double> cum; int vmin; normal_trunc_table(mu, sigma, cum, vmin); return pmf_from_cum(cum, vmin); }
// trunc(Exp * tau) (s1.py:193, pulse.py:341): P(X <= k) = 1 - exp(-(k + 1) / tau)
static Pmf pmf_exp(double tau)
{
    std::vector<double> cum;
    if (!(tau > 0)) cum.push_back(1.0);
    else for (long k = 0; k < 60000; k++) { const double v = -expm1(-(double)(k + 1) / tau); cum.push_back(v); if (v >= 1.0) break; }
    cum.back() = 1.0;
    return pmf_from_cum(cum, 0);
}
// trunc(np.interp(u, x, t)) (s2.py:338): P(L <= y) is the inverse of the increasing piecewise linear map u -> L
static Pmf pmf_luminescence(const std::vector<double> &x, const std::vector<double> &t)
{
    const int n = (int)x.size();
    if (n < 2) { Pmf r; r.p = {1.0}; return r; }
    auto cdf = [&](double y) {
        if (y < t[0]) return 0.0;
        if (y >= t[n - 1]) return 1.0;
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t[mid] <= y) lo = mid; else hi = mid; }
        return x[lo] + (y - t[lo]) / (t[lo + 1] - t[lo]) * (x[lo + 1] - x[lo]);
    };
    const long lo = (long)floor(t[0]) - 1, hi = (long)ceil(t[n - 1]) + 1;
    std::vector<double> cum;
    for (long k = lo; k <= hi; k++) cum.push_back(cdf((double)(k >= 0 ? k + 1 : k)));      // the cast truncates toward zero
    cum.back() = 1.0;
    return pmf_from_cum(cum, lo);
}
static Pmf pmf_mix(const Pmf &a, double wa, const Pmf &b, double wb)
{
    Pmf r; r.vmin = std::min(a.vmin, b.vmin);
    const long hi = std::max(a.vmin + (long)a.p.size(), b.vmin + (long)b.p.size());
    r.p.assign((size_t)(hi - r.vmin), 0.0);
    for (size_t i = 0; i < a.p.size(); i++) r.p[(size_t)(a.vmin - r.vmin) + i] += wa * a.p[i];
    for (size_t i = 0; i < b.p.size(); i++) r.p[(size_t)(b.vmin - r.vmin) + i] += wb * b.p[i];
    return r;
}
static Pmf pmf_conv(const Pmf &a, const Pmf &b)
{
    Pmf r; r.vmin = a.vmin + b.vmin; r.p.assign(a.p.size() + b.p.size() - 1, 0.0);
    for (size_t i = 0; i < a.p.size(); i++) { const double ai = a.p[i]; if (ai == 0.0) continue; for (size_t j = 0; j < b.p.size(); j++) r.p[i + j] += ai * b.p[j]; }
    return r;
}
// cumulative table of a pmf: leading zeros dropped, cut where the running sum reaches 1 (the oracle does the same)
static void cum_of_pmf(const Pmf &a, std::vector<double> &cum, long &vmin)
{
    size_t first = 0; while (first + 1 < a.p.size() && a.p[first] == 0.0) first++;
    cum.clear(); double acc = 0;
    for (size_t i = first; i < a.p.size(); i++) { acc += a.p[i]; cum.push_back(acc); if (acc >= 1.0) break; }
    while (cum.size() > 1 && cum[cum.size() - 2] >= 1.0) cum.pop_back();
    cum.back() = 1.0;
    vmin = a.vmin + (long)first;
}
static int upload_pmf(wfs_handle *h, int slot, const Pmf &a, AliasTab &out)
{
    std::vector<double> cum; long vmin;
    cum_of_pmf(a, cum, vmin);
    return upload_disc(h, h->tt_alias[slot], cum, (int)vmin, out);
}

// tab_tts: transit time alone (photons that arrive with their times: RawDataOptical); tab_s1 / tab_s2: every delay term of
// an S1 / S2 photon relative to its emitter (s1.py:193-194, s2.py:338 + pulse.py:339-341 + s2.py:550, pulse.py:54-56)
int build_time_tables(wfs_handle *h)
{
    const wfs_config &c = h->cfg; WfsDev &d = h->dev;
    const Pmf tts = pmf_normal(c.tts_mean, c.tts_sigma);
    TRY(upload_pmf(h, 0, tts, d.tab_tts));
    Pmf s1; s1.p = {1.0};
    if (c.s1_simple) { s1 = pmf_exp(c.s1_decay_time); if (c.s1_decay_spread != 0.0) s1 = pmf_conv(s1, pmf_normal(0.0, c.s1_decay_spread)); }
    TRY(upload_pmf(h, 1, pmf_conv(s1, tts), d.tab_s1));
    const Pmf st = pmf_mix(pmf_exp(c.t1_gas), c.sf_gas, pmf_exp(c.t3_gas), 1.0 - c.sf_gas);
    Pmf s2 = pmf_conv(st, pmf_luminescence(h->h_lum_x, h->h_lum_t)), s2n = st;
    if (c.s2_time_model == 1 && c.s2_time_spread != 0.0) { s2 = pmf_conv(s2, pmf_normal(0.0, c.s2_time_spread)); s2n = pmf_conv(s2n, pmf_normal(0.0, c.s2_time_spread)); }
    TRY(upload_pmf(h, 2, pmf_conv(s2, tts), d.tab_s2));
    // bases of the model-variant tables (wfs_set_delay_models): 0 transit time, 1 S1 terms, 2 S2 terms, 3 S2 without luminescence
    h->base_pmf = {tts, pmf_conv(s1, tts), pmf_conv(s2, tts), pmf_conv(s2n, tts)};
    return WFS_OK;
}

void refresh_dev(wfs_handle *h)
{
    const wfs_config &c = h->cfg; WfsDev &d = h->dev;
    d.dt = c.dt; d.samples_before = c.samples_before; d.samples_after = c.samples_after; d.store_before = c.store_before;
    d.store_after = c.store_after; d.tlen = c.tlen; d.tw = c.trigger_window; d.baseline = c.baseline; d.n_rows = c.n_rows;
    d.n_tpc = c.n_tpc; d.n_top = c.n_top; d.he_first = c.he_first; d.he_factor = c.he_factor; d.last_bottom = c.last_bottom;
    d.detector_nt = c.detector_nt; d.s1_simple = c.s1_simple; d.s2_time_model = c.s2_time_model; d.enable_pmt_ap = c.enable_pmt_ap;
    d.c2a = c.c2a; d.tts_mean = c.tts_mean; d.tts_sigma = c.tts_sigma; d.p_dpe = c.p_dpe; d.s1_decay_time = c.s1_decay_time;
    d.s1_decay_spread = c.s1_decay_spread; d.sf_gas = c.sf_gas; d.t1_gas = c.t1_gas; d.t3_gas = c.t3_gas;
    d.s2_time_spread = c.s2_time_spread; d.trap_time = c.trap_time; d.gain_spread = c.gain_spread;
    d.pmt_ap_modifier = c.pmt_ap_modifier; d.pmt_ap_t_modifier = c.pmt_ap_t_modifier; d.rext = c.rext;
    d.k0 = (u32)c.seed; d.k1 = (u32)(c.seed >> 32);
    auto thr = [](double p) -> u64 { if (!(p > 0)) return 0; if (p >= 1) return 4294967296ull; return (u64)(p * 4294967296.0); };
    d.thr_dpe = thr(c.p_dpe); d.dpe_inv = d.thr_dpe ? 2000.0 / (double)d.thr_dpe : 0.0;
    // HE rows are only materialised when they can differ from a flat baseline: a non-zero int(factor)
    // (rawdata.py:242) or noise columns for the HE channels
    d.enable_noise = (c.enable_noise && d.noise != nullptr) ? 1 : 0;
    bool he_noise = d.enable_noise && d.noise_channels > c.he_first;
    d.he_rows = (c.detector_nt && c.n_top > 0 && (c.he_factor != 0 || he_noise)) ? 1 : 0;
    d.row_slots = c.n_tpc + (d.he_rows ? c.n_top : 0);
}

}  // namespace


// Exception firewall of the C ABI: every extern "C" entry point is a function-try-block, so that a std::bad_alloc /
// std::length_error from the host-side containers becomes an error code + wfs_last_error instead of std::terminate.
static int wfs_caught(wfs_handle *h, const char *fn) noexcept
{
    int code = WFS_E_INVALID; const char *what = "unknown C++ exception";
    char buf[256];
    try { throw; }
    catch (const std::bad_alloc &) { code = WFS_E_NOMEM; what = "out of host memory (std::bad_alloc)"; }
    catch (const std::exception &e) { snprintf(buf, sizeof buf, "%s", e.what()); what = buf; }
    catch (...) {}
    if (h) { try { h->err = std::string(fn) + ": " + what; } catch (...) {} }
    return code;
}
#define WFS_CATCH(h) catch (...) { return wfs_caught((h), __func__); }

extern "C" {

int wfs_device_count(int *n) try { return hipGetDeviceCount(n) == hipSuccess ? WFS_OK : WFS_E_HIP; } WFS_CATCH(nullptr)

int wfs_create(const wfs_config *cfg, int device, wfs_handle **out)
try {
    if (!cfg || !out) return WFS_E_INVALID;
    // (the fast kernels are specialised for 10 ns samples and 22-tap templates; any other geometry runs through k_pulse_generic)
    if (cfg->n_tpc <= 0 || cfg->n_tpc > WFS_MAX_CH || cfg->dt < 1 || cfg->dt > WFS_MAX_DT || cfg->tlen < 1 || cfg->tlen > WFS_MAX_TLEN
        || cfg->samples_before + cfg->samples_after != cfg->tlen) return WFS_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return WFS_E_HIP;      // fail loudly: no CPU fallback
    if (device < 0 || device >= ndev) return WFS_E_INVALID;
    if (hipSetDevice(device) != hipSuccess) return WFS_E_HIP;
    wfs_handle *h = new wfs_handle();
    h->cfg = *cfg; h->device = device;
    h->generic_geom = cfg->dt != WFS_DT || cfg->tlen != 22;
    { hipDeviceProp_t prop; h->n_cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256; }
    memset(&h->dev, 0, sizeof(h->dev));
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return WFS_E_HIP; }
    h->own_stream = true;
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->rec_copied[0], hipEventDisableTiming) != hipSuccess
        || hipEventCreateWithFlags(&h->rec_copied[1], hipEventDisableTiming) != hipSuccess) { delete h; return WFS_E_HIP; }
    if (hipMalloc(&h->scal.p, 512) != hipSuccess) { delete h; return WFS_E_HIP; }
    if (hipHostMalloc((void **)&h->h_scal, 512, hipHostMallocDefault) != hipSuccess) { hipFree(h->scal.p); delete h; return WFS_E_HIP; }
    memset(h->h_scal, 0, 512);
    h->scal.cap = 256;
#ifdef WFS_STAMPS
    if (hipMalloc(&h->stamps.p, 4096 * 64 * 8) != hipSuccess) { delete h; return WFS_E_HIP; }
    h->stamps.cap = 4096 * 64 * 8; hipMemset(h->stamps.p, 0, 4096 * 64 * 8); h->dev.stamps = h->stamps.as<unsigned long long>();
#endif
    refresh_dev(h);
    if (const char *e = getenv("WFS_TAP_SPARSE_MAX")) h->tap_sparse_max = std::min(atoi(e), TAP_LIST_LEN - 1);      // tuning knob of tap_block (results do not depend on it)
    if (build_time_tables(h) != WFS_OK) { delete h; return WFS_E_HIP; }
    // the pulse kernel stages up to 1024 start bins per tile: (10 * 1024 + 220) * 8 + 1024 * 4 bytes of LDS
#define WFS_BIG_LDS(K, B) hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, (B) * 1024)
#define WFS_BIG_LDS_F(KM, B) do { WFS_BIG_LDS(HIP_KERNEL_NAME(KM(true)), B); WFS_BIG_LDS(HIP_KERNEL_NAME(KM(false)), B); } while (0)
    WFS_BIG_LDS_F(K_PULSE_256_RES, 100); WFS_BIG_LDS_F(K_PULSE_128_RES, 100); WFS_BIG_LDS_F(K_PULSE_256_WIN, 100); WFS_BIG_LDS_F(K_PULSE_128_WIN, 100);
    hipFuncSetAttribute((const void *)k_photon_fill<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void *)k_photon_fill<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void *)k_photon_fill<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void *)k_photon_fill<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void *)k_tile_order_big, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    WFS_BIG_LDS_F(K_PULSE_GENERIC, 128);
#define K_ROW_PULSE_0(F) k_row_pulse<0, F, true>
#define K_ROW_PULSE_1(F) k_row_pulse<1, F, true>
#define K_ROW_PULSE_2(F) k_row_pulse<2, F, true>
    WFS_BIG_LDS_F(K_ROW_PULSE_0, 128); WFS_BIG_LDS_F(K_ROW_PULSE_1, 128); WFS_BIG_LDS_F(K_ROW_PULSE_2, 128);
    if (const char *e = getenv("WFS_ROW_RESIDENT")) h->res_env = atoi(e);          // 0 / 1 / 2 as the config switch (A/B runs)
    if (const char *e = getenv("WFS_RES_MAX_LEN")) h->res_max_len = std::max(256, std::min(atoi(e), 7168)) / 256 * 256;      // tuning knob (results do not depend on it)
    WFS_BIG_LDS_F(K_S2_TILE_FULL, 100); WFS_BIG_LDS_F(K_S2_TILE_FULL_AP, 100); WFS_BIG_LDS_F(K_S2_TILE_GEN, 100); WFS_BIG_LDS_F(K_S2_TILE_GEN_AP, 100);
    WFS_BIG_LDS_F(K_PULSE_SPARSE_64, 100); WFS_BIG_LDS_F(K_PULSE_SPARSE_256, 100);
    *out = h;
    return WFS_OK;
} WFS_CATCH(nullptr)

int wfs_destroy(wfs_handle *h)
try {
    if (!h) return WFS_OK;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);

#ifdef WFS_STAMPS
    {   // diagnostic build: average cycles per workgroup and phase
        std::vector<unsigned long long> all((size_t)4096 * 64); unsigned long long v[64] = {0};
        if (hipMemcpy(all.data(), h->stamps.p, all.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            for (size_t r = 0; r < 4096; r++) for (int i = 0; i < 64; i++) v[i] += all[r * 64 + i];
            for (int i = 0; i < 32; i++) if (v[i + 32]) fprintf(stderr, "stamp %2d: %10.0f cycles/wg  (%llu wgs, %.3e cycles total)\n", i, (double)v[i] / (double)v[i + 32], v[i + 32], (double)v[i]);
        }
        hipFree(h->stamps.p);
    }
#endif
    DevBuf *all[] = {&h->rec_key, &h->rec_key2, &h->rec_val, &h->rec_val2, &h->rec_dest, &h->sort_tmp, &h->row_desc, &h->pmap[0].values, &h->pmap[1].values, &h->map_row_ins[0], &h->map_row_ins[1], &h->map_row_id[0], &h->map_row_id[1], &h->map_x, &h->map_y, &h->map_z, &h->map_nb_idx[0], &h->map_nb_idx[1], &h->map_nb_w[0], &h->map_nb_w[1], &h->d_tabs, &h->gg_inv, &h->ins_gg, &h->ins_ggw, &h->ins_ggsum, &h->prop_top, &h->prop_bot, &h->ins_tab, &h->ins_tabb, &h->ins_pzi, &h->ins_pzf, &h->blk_e, &h->blk_base, &h->blk_cnt, &h->blk_ins, &h->eblk_ins, &h->ins_ph0, &h->blk_desc, &h->noise_override, &h->ap_ins, &h->ap_ch, &h->ap_t, &h->ap_gain, &h->ap_cand, &h->ap_args_dev, &h->ap_seg, &h->set_gid, &h->opt_t, &h->opt_item, &h->opt_first, &h->opt_last, &h->opt_ch, &h->opt_time, &h->t_templates, &h->t_spe, &h->t_gains, &h->t_thr_truth, &h->t_thr_zle, &h->t_lumx, &h->t_lumt, &h->t_noise, &h->t_noise_f,
        &h->ins_type, &h->ins_time, &h->ins_amp, &h->ins_gid, &h->ins_p, &h->ins_dm, &h->ins_ds, &h->ins_sc, &h->ins_cdfrow, &h->cdf_table, &h->cdf_guide, &h->chan_alias, &h->ins_embase, &h->ins_set, &h->set_ins_off, &h->set_ins_list,
        &h->em_off, &h->em_zg, &h->pois_cdf, &h->pois_kmin, &h->set_cluster, &h->set_t0, &h->set_mode, &h->cl_tmin, &h->cl_gid, &h->cl_end, &h->cl_group, &h->em_time, &h->em_nph,
        &h->em_ins, &h->em_ph_off, &h->el_stat, &h->el_minmax, &h->tile_count, &h->tile_off, &h->tile_cursor, &h->tile_tmin, &h->tile_tmax,
        &h->active_tiles, &h->sparse_tiles, &h->dense_tiles, &h->wave_tiles, &h->ph, &h->ph_gain, &h->grp_lo, &h->grp_hi, &h->grp_left, &h->grp_right, &h->grp_ixrand,
        &h->grp_gid, &h->row_lo, &h->row_hi, &h->acc_len, &h->acc_off, &h->itv_cap, &h->itv_off, &h->active_rows, &h->raw, &h->itv_left,
        &h->itv_right, &h->itv_n, &h->row_nrec, &h->rec_off, &h->records_ab[0], &h->records_ab[1], &h->truth, &h->tminmax, &h->tile_truth, &h->tile_desc, &h->gather_idx, &h->gather_out, &h->currents, &h->cur_len, &h->cur_off,
        &h->row_dbg, &h->row_dbg_len, &h->row_dbg_off, &h->scan_tmp, &h->scal,
        &h->ph_idx, &h->ap_key, &h->order_list, &h->order_list2, &h->ins_sbase, &h->tile_tail, &h->tile_tailbase, &h->ins_fullsort, &h->row_pmax, &h->ins_fused, &h->ins_nsurv, &h->ins_bcap, &h->ins_bcap_all, &h->ins_boff, &h->et32, &h->ftiles, &h->tbuf, &h->row_cnt, &h->row_tile, &h->tile_done, &h->huge_start, &h->huge_cbeg, &h->huge_keys, &h->huge_keys2, &h->huge_vals, &h->huge_vals2, &h->huge_rec, &h->huge_gain,
        &h->pack_desc, &h->row_bad, &h->fin_len, &h->res_cnt, &h->fin_off, &h->res_toff, &h->res_desc, &h->fin, &h->res_long, &h->res_rows,
        &h->pmap[0].cell_start, &h->pmap[0].cell_pts, &h->pmap[1].cell_start, &h->pmap[1].cell_pts};
    for (DevBuf *b : all) if (b->p) hipFree(b->p);
    if (h->h_scal) hipHostFree(h->h_scal);
    for (DevBuf *b : {&h->pmap[0].points, &h->pmap[1].points, &h->smap_pos, &h->smap_out, &h->smap_nb_idx, &h->smap_nb_w, &h->ins_aft, &h->ins_sigr, &h->ins_siga, &h->diff_row_ins, &h->diff_row_id, &h->diff_pre}) if (b->p) hipFree(b->p);
    for (auto &m : h->smaps) for (DevBuf *b : {&m->g.values, &m->g.points, &m->tx, &m->ty, &m->c}) if (b->p) hipFree(b->p);
    for (int q = 0; q < 6; q++) if (h->tt_alias[q].p) hipFree(h->tt_alias[q].p);
    for (auto &b : h->x_alias) if (b.p) hipFree(b.p);
    for (auto &a : h->ap) { if (a.delay_cdf.p) hipFree(a.delay_cdf.p); if (a.amp_cdf.p) hipFree(a.amp_cdf.p); if (a.prob.p) hipFree(a.prob.p); if (a.thr.p) hipFree(a.thr.p); if (a.delay_guide.p) hipFree(a.delay_guide.p); if (a.amp_guide.p) hipFree(a.amp_guide.p); }
    for (auto &t : h->times) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    if (h->own_stream) hipStreamDestroy(h->stream);
    if (h->copy_stream) { hipStreamSynchronize(h->copy_stream); hipStreamDestroy(h->copy_stream); }
    for (int q = 0; q < 2; q++) if (h->rec_copied[q]) hipEventDestroy(h->rec_copied[q]);
    delete h;
    return WFS_OK;
} WFS_CATCH(nullptr)

const char *wfs_last_error(const wfs_handle *h) { return h ? h->err.c_str() : "null handle"; }

int wfs_set_stream(wfs_handle *h, void *s)
try {
    if (!h) return WFS_E_INVALID;
    if (h->own_stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); h->own_stream = false; }
    h->stream = (hipStream_t)s;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_synchronize(wfs_handle *h) try { if (!h) return WFS_E_INVALID; HIPCHK(hipStreamSynchronize(h->stream)); return WFS_OK; } WFS_CATCH(h)
// a noise array of floats (resource.noise_data of a float dtype with non-integral values): add_noise (rawdata.py:436) adds it into
// the int64 row inside numba, which stores the TRUNCATED SUM -- not the sum with the truncated noise.  Replaces the int16 table.
int wfs_set_noise_float(wfs_handle *h, const double *noise, int32_t noise_len, int32_t noise_channels)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->tables_set) return h->fail(WFS_E_STATE, "wfs_set_tables must be called first");
    if (!noise || noise_len <= 0 || noise_channels <= 0) return h->fail(WFS_E_INVALID, "wfs_set_noise_float: empty noise array");
    HIPCHK(hipSetDevice(h->device));
    const size_t stride = (size_t)noise_len + NOISE_PAD;        // (every row followed by its own first samples: wfs_device.h NOISE_PAD)
    std::vector<double> nt(stride * noise_channels);
    for (int64_t i = 0; i < (int64_t)stride; i++) for (int c = 0; c < noise_channels; c++) nt[(size_t)c * stride + i] = noise[(size_t)(i % noise_len) * noise_channels + c];
    TRY(upload(h, h->t_noise_f, nt.data(), sizeof(double) * nt.size()));
    HIPCHK(hipStreamSynchronize(h->stream));
    WfsDev &d = h->dev;
    d.noise_f = h->t_noise_f.as<double>(); d.noise = (const int16_t *)d.noise_f; d.noise_len = noise_len; d.noise_channels = noise_channels; d.noise_stride = (i32)stride;
    refresh_dev(h);
    return WFS_OK;
} WFS_CATCH(h)

int wfs_set_noise_offsets(wfs_handle *h, const int64_t *ix, int64_t n)
try {
    if (!h || n < 0 || (n > 0 && !ix)) return WFS_E_INVALID;
    h->n_noise_override = n;
    if (n) { TRY(upload(h, h->noise_override, ix, (size_t)n * 8)); HIPCHK(hipStreamSynchronize(h->stream)); }
    return WFS_OK;
} WFS_CATCH(h)
int wfs_set_window_carry(wfs_handle *h, int32_t has, int64_t t) try { if (!h) return WFS_E_INVALID; h->carry_has = has; h->carry_runmax = t; return WFS_OK; } WFS_CATCH(h)
int wfs_copy_cluster_groups(wfs_handle *h, int32_t *group, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (cap < h->n_clusters) return h->fail(WFS_E_CAPACITY, "cluster buffer too small");
    HIPCHK(hipMemcpy(group, h->cl_group.p, (size_t)h->n_clusters * 4, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)
int wfs_set_debug(wfs_handle *h, int32_t keep) try { if (!h) return WFS_E_INVALID; h->keep_currents = keep; return WFS_OK; } WFS_CATCH(h)
int wfs_set_profiling(wfs_handle *h, int32_t on) try { if (!h) return WFS_E_INVALID; h->profiling = on; return WFS_OK; } WFS_CATCH(h)

int wfs_set_tables(wfs_handle *h, const double *templates, const double *spe, int32_t n_spe, const double *gains,
                   const double *thr_truth, const int64_t *thr_zle, const double *lum_x, const double *lum_t, int32_t n_lum,
                   const int16_t *noise, int32_t noise_len, int32_t noise_channels)
try {
    if (!h || !templates || !spe || !gains || !thr_truth || !thr_zle || n_spe < 1) return h ? h->fail(WFS_E_INVALID, "wfs_set_tables: null table") : WFS_E_INVALID;
    if (n_spe != 1 && n_spe < h->cfg.n_tpc) return h->fail(WFS_E_INVALID, "wfs_set_tables: n_spe must be 1 or >= n_tpc");
    HIPCHK(hipSetDevice(h->device));
    const wfs_config &c = h->cfg;
    TRY(upload(h, h->t_templates, templates, sizeof(double) * c.dt * c.tlen));
    TRY(upload(h, h->t_spe, spe, sizeof(double) * 2001 * (size_t)n_spe));
    TRY(upload(h, h->t_gains, gains, sizeof(double) * c.n_tpc));
    TRY(upload(h, h->t_thr_truth, thr_truth, sizeof(double) * c.n_rows));
    TRY(upload(h, h->t_thr_zle, thr_zle, sizeof(int64_t) * c.n_rows));
    WfsDev &d = h->dev;
    d.n_lum = 0; h->h_lum_x.clear(); h->h_lum_t.clear();
    if (lum_x && lum_t && n_lum >= 2) {
        for (int i = 0; i + 1 < n_lum; i++)
            if (!(lum_x[i + 1] >= lum_x[i]) || !(lum_t[i + 1] > lum_t[i])) return h->fail(WFS_E_INVALID, "luminescence table must be increasing");
        TRY(upload(h, h->t_lumx, lum_x, sizeof(double) * n_lum)); TRY(upload(h, h->t_lumt, lum_t, sizeof(double) * n_lum));
        d.n_lum = n_lum; h->h_lum_x.assign(lum_x, lum_x + n_lum); h->h_lum_t.assign(lum_t, lum_t + n_lum);
    }
    TRY(build_time_tables(h));         // the S2 delay table contains the luminescence term
    d.noise = nullptr; d.noise_f = nullptr; d.noise_len = 0; d.noise_channels = 0; d.noise_stride = 0;
    if (noise && noise_len > 0 && noise_channels > 0) {
        // channel-major on the device ([channel][sample]; the reference's array is [sample][channel], rawdata.py:429): a row reads
        // consecutive samples of ONE channel -- time-major that is one cache line per sample
        // (and every row followed by its own first samples: wfs_device.h NOISE_PAD)
        const size_t stride = (size_t)noise_len + NOISE_PAD;
        std::vector<int16_t> nt(stride * noise_channels);
        for (int64_t i = 0; i < (int64_t)stride; i++) for (int c = 0; c < noise_channels; c++) nt[(size_t)c * stride + i] = noise[(size_t)(i % noise_len) * noise_channels + c];
        TRY(upload(h, h->t_noise, nt.data(), sizeof(int16_t) * nt.size()));
        HIPCHK(hipStreamSynchronize(h->stream));
        d.noise = h->t_noise.as<int16_t>(); d.noise_len = noise_len; d.noise_channels = noise_channels; d.noise_stride = (i32)stride;
    }
    d.templates = h->t_templates.as<double>(); d.spe = h->t_spe.as<double>(); d.n_spe = n_spe; d.gains = h->t_gains.as<double>();
    d.thr_truth = h->t_thr_truth.as<double>(); d.thr_zle = h->t_thr_zle.as<i64>(); d.lum_x = h->t_lumx.as<double>(); d.lum_t = h->t_lumt.as<double>();
    h->h_templates.assign(templates, templates + (size_t)c.dt * c.tlen);
    h->h_gains.assign(gains, gains + c.n_tpc);
    for (int r = 0; r < c.dt; r++) {                       // pulse.py:32 current_max
        double m = templates[r * c.tlen];
        for (int k = 1; k < c.tlen; k++) m = std::max(m, templates[r * c.tlen + k]);
        d.current_max[r] = m;
    }
    refresh_dev(h);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->tables_set = true;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_set_ap_element(wfs_handle *h, int32_t e, int32_t n_bins_delay, int32_t n_bins_amp, int32_t amp_2d, int32_t is_uniform,
                       double delay_bin, double amp_bin, const double *delay_cdf, const double *amp_cdf)
try {
    if (!h || e < 0 || e >= WFS_MAX_AP || !delay_cdf || !amp_cdf) return WFS_E_INVALID;
    ApElem &a = h->ap[e];
    a.n_bins_delay = n_bins_delay; a.n_bins_amp = n_bins_amp; a.amp_2d = amp_2d; a.is_uniform = is_uniform; a.delay_bin = delay_bin; a.amp_bin = amp_bin;
    TRY(upload(h, a.delay_cdf, delay_cdf, sizeof(double) * (size_t)h->cfg.n_tpc * n_bins_delay));
    TRY(upload(h, a.amp_cdf, amp_cdf, sizeof(double) * (size_t)(amp_2d ? h->cfg.n_tpc : 1) * n_bins_amp));
    // the probability column on its own (last entry of every channel's delay row): the generator keeps it in LDS
    std::vector<double> prob((size_t)h->cfg.n_tpc);
    for (int c = 0; c < h->cfg.n_tpc; c++) prob[c] = delay_cdf[(size_t)c * n_bins_delay + n_bins_delay - 1];
    TRY(upload(h, a.prob, prob.data(), sizeof(double) * prob.size()));
    a.prob_h = prob; a.thr_mod = -1.0;
    // non-decreasing rows (every cumulative distribution is): np.argmin(|cdf - u|) by bisection on the device
    auto rows_sorted = [](const double *c, size_t rows, int n) {
        for (size_t r = 0; r < rows; r++) for (int k = 1; k < n; k++) if (!(c[r * n + k] >= c[r * n + k - 1])) return 0;
        return 1;
    };
    a.delay_sorted = rows_sorted(delay_cdf, (size_t)h->cfg.n_tpc, n_bins_delay);
    a.amp_sorted = rows_sorted(amp_cdf, (size_t)(amp_2d ? h->cfg.n_tpc : 1), n_bins_amp);
    // guides of the sorted rows (ApGuide): for every cell of u * scale a bracket of "first entry >= u" that holds for every u the
    // device can put into the cell -- the cell's ends moved outwards by 2^-40 (the one rounding of u * scale is 2^-53)
    auto build_guides = [&](DevBuf &buf, const double *c, size_t rows, int n) -> int {
        if (buf.p) { hipFree(buf.p); buf.p = nullptr; buf.cap = 0; }
        if (n < 1 || n > 65535) return WFS_OK;
        std::vector<ApGuide> g(rows);
        const double margin = 9.094947017729282e-13;            // 2^-40
        for (size_t r = 0; r < rows; r++) {
            const double *row = c + r * n;
            auto first_ge = [&](double x) { return (int)(std::lower_bound(row, row + n, x) - row); };
            const double top = row[n - 1];
            g[r].scale = top > 0.0 ? (double)AP_GUIDE / top : 0.0;
            for (int j = 0; j < AP_GUIDE; j++) {
                if (!(g[r].scale > 0.0)) { g[r].lo[j] = 0; g[r].hi[j] = (unsigned short)n; continue; }
                const double x_lo = (double)j / g[r].scale * (1.0 - margin), x_hi = (double)(j + 1) / g[r].scale * (1.0 + margin);
                g[r].lo[j] = (unsigned short)first_ge(x_lo);
                g[r].hi[j] = j == AP_GUIDE - 1 ? (unsigned short)n : (unsigned short)first_ge(x_hi);       // (the last cell also takes every u above the row)
            }
        }
        int rc = upload(h, buf, g.data(), g.size() * sizeof(ApGuide)); if (rc) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        return WFS_OK;
    };
    if (a.delay_sorted) TRY(build_guides(a.delay_guide, delay_cdf, (size_t)h->cfg.n_tpc, n_bins_delay)); else if (a.delay_guide.p) { hipFree(a.delay_guide.p); a.delay_guide.p = nullptr; a.delay_guide.cap = 0; }
    if (a.amp_sorted) TRY(build_guides(a.amp_guide, amp_cdf, (size_t)(amp_2d ? h->cfg.n_tpc : 1), n_bins_amp)); else if (a.amp_guide.p) { hipFree(a.amp_guide.p); a.amp_guide.p = nullptr; a.amp_guide.cap = 0; }
    h->dev.n_ap = std::max(h->dev.n_ap, e + 1);
    HIPCHK(hipStreamSynchronize(h->stream));
    return WFS_OK;
} WFS_CATCH(h)

// ---------------------------------------------------------------------------------------------- batch input
static int load_clusters(wfs_handle *h, i64 n, const int32_t *cluster, const int64_t *tmin, const uint32_t *gid)
{
    // clusters are contiguous runs of the (sorted) input; first member gives the key and the noise stream id
    std::vector<i64> cl_tmin; std::vector<u32> cl_gid;
    for (i64 i = 0; i < n; i++) {
        if (cluster[i] < 0 || (i > 0 && (cluster[i] < cluster[i - 1] || cluster[i] > cluster[i - 1] + 1)) || (i == 0 && cluster[0] != 0))
            return h->fail(WFS_E_INVALID, "cluster ids must start at 0 and be non-decreasing without gaps");
        if ((i64)cl_tmin.size() <= cluster[i]) { cl_tmin.push_back(tmin[i]); cl_gid.push_back(gid ? gid[i] : (u32)i); }
        else cl_tmin[cluster[i]] = std::min(cl_tmin[cluster[i]], (i64)tmin[i]);
    }
    h->n_clusters = (i64)cl_tmin.size();
    TRY(upload(h, h->cl_tmin, cl_tmin.data(), cl_tmin.size() * 8));
    TRY(upload(h, h->cl_gid, cl_gid.data(), cl_gid.size() * 4));
    HIPCHK(hipStreamSynchronize(h->stream));       // vectors go out of scope
    return WFS_OK;
}

int wfs_load_instructions(wfs_handle *h, int64_t n, const int8_t *type, const int64_t *time, const int32_t *amp, const uint32_t *gid,
                          const int32_t *cluster, const int64_t *tmin, const double *p_hit, const double *drift_mean,
                          const double *drift_spread, const double *sc_gain, const int32_t *cdf_row, const double *cdf_table, int32_t n_cdf,
                          const int32_t *run_set, int64_t n_run_sets, const uint32_t *em_base)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->tables_set) return h->fail(WFS_E_STATE, "wfs_set_tables must be called first");
    if (n <= 0 || !type || !time || !amp || !gid || !cluster || !tmin || !p_hit || !drift_mean || !drift_spread || !sc_gain || !cdf_row || !cdf_table || n_cdf <= 0)
        return h->fail(WFS_E_INVALID, "wfs_load_instructions: null or empty input");
    HIPCHK(hipSetDevice(h->device));
    std::vector<i64> em_off((size_t)n + 1);
    em_off[0] = 0;
    bool any_ptrs = false;
    for (i64 i = 0; i < n; i++) {
        if (type[i] != 1 && type[i] != 2 && type[i] != 4 && type[i] != 6)      // 4 / 6: electron afterpulses, simulated like an S2 (afterpulse.py:14, 94)
            return h->fail(WFS_E_INVALID, "instruction types: 1 (S1), 2 (S2), 4 / 6 (photo-ionisation / photo-electric electrons)");
        if (amp[i] < 0) return h->fail(WFS_E_INVALID, "negative amp");
        if (cdf_row[i] < -1 || cdf_row[i] >= n_cdf) return h->fail(WFS_E_INVALID, "cdf_row out of range");
        if (cdf_row[i] == -1 && !h->pmap[type[i] == 1 ? 0 : 1].set) return h->fail(WFS_E_STATE, "cdf_row -1 needs wfs_set_pattern_map for the instruction type");
        if (type[i] != 1 && h->dev.n_lum < 2) return h->fail(WFS_E_STATE, "S2 instructions need the luminescence table");
        em_off[i + 1] = em_off[i] + (type[i] == 1 ? 1 : (i64)amp[i]);
        if (type[i] != 1 && sc_gain[i] > 217.0) any_ptrs = true;          // POIS_LAM_MAX: the electrons of such an instruction draw with PTRS (k_s2_photons)
    }
    // pulse sets (one Pulse.__call__ each, rawdata.py:108-127): the caller's run sets, by default one per instruction
    std::vector<i32> ins_set((size_t)n);
    i64 PS = n;
    if (run_set) {
        if (n_run_sets <= 0 || n_run_sets > n) return h->fail(WFS_E_INVALID, "wfs_load_instructions: bad number of run sets");
        PS = n_run_sets;
        for (i64 i = 0; i < n; i++) { if (run_set[i] < 0 || run_set[i] >= PS) return h->fail(WFS_E_INVALID, "run set index out of range"); ins_set[i] = run_set[i]; }
    } else for (i64 i = 0; i < n; i++) ins_set[i] = (i32)i;
    std::vector<i64> set_off((size_t)PS + 1, 0); std::vector<i32> set_list((size_t)n);        // set -> its instructions (CSR)
    for (i64 i = 0; i < n; i++) set_off[ins_set[i] + 1]++;
    // (a run set may be empty: a caller that numbers the sets by their first instruction -- what lets single-instruction sets take the
    // tile-local generator next to the shared Pulse calls of electron afterpulses -- leaves the numbers of the other members unused)
    for (i64 q = 0; q < PS; q++) set_off[q + 1] += set_off[q];
    { std::vector<i64> cur(set_off.begin(), set_off.end() - 1); for (i64 i = 0; i < n; i++) set_list[cur[ins_set[i]]++] = (i32)i; }
    for (i64 q = 0; q < PS; q++)
        for (i64 k = set_off[q] + 1; k < set_off[q + 1]; k++)
            if (cluster[set_list[k]] != cluster[set_list[set_off[q]]] || type[set_list[k]] != type[set_list[set_off[q]]])
                return h->fail(WFS_E_INVALID, "the instructions of a run set must share cluster and type");
    // PMT afterpulses are a second pulse set per primary set (rawdata.py:176-178): set PS + q belongs to set q
    h->ap_active = h->cfg.enable_pmt_ap && h->dev.n_ap > 0;
    const i64 S = h->ap_active ? 2 * PS : PS;
    h->any_ptrs = any_ptrs;
    h->run_sets_given = run_set != nullptr;
    h->sets_aligned = PS == n;                     // every set carries the index of its first instruction
    for (i64 q = 0; q < PS && h->sets_aligned; q++) if (set_off[q + 1] > set_off[q] && set_list[set_off[q]] != (i32)q) h->sets_aligned = false;
    h->any_s2 = false; for (i64 i = 0; i < n; i++) if (type[i] == 2) { h->any_s2 = true; break; }
    h->n_ins = n; h->n_psets = PS; h->n_sets = S; h->n_emitters = em_off[n]; h->n_tiles = S * h->cfg.n_tpc;
    h->h_rs_off = set_off; h->h_rs_list = set_list;
    if (h->n_tiles > 0x7fffffffLL) return h->fail(WFS_E_CAPACITY, "too many tiles in one batch");
    TRY(upload(h, h->ins_type, type, (size_t)n)); TRY(upload(h, h->ins_time, time, (size_t)n * 8)); TRY(upload(h, h->ins_amp, amp, (size_t)n * 4));
    { std::vector<u32> eb((size_t)n, 0u); if (em_base) eb.assign(em_base, em_base + n); TRY(upload(h, h->ins_embase, eb.data(), (size_t)n * 4)); HIPCHK(hipStreamSynchronize(h->stream)); }
    TRY(upload(h, h->ins_gid, gid, (size_t)n * 4)); TRY(upload(h, h->ins_p, p_hit, (size_t)n * 8)); TRY(upload(h, h->ins_dm, drift_mean, (size_t)n * 8));
    TRY(upload(h, h->ins_ds, drift_spread, (size_t)n * 8)); TRY(upload(h, h->ins_sc, sc_gain, (size_t)n * 8));
    // rows -1: the channel CDF of the instruction comes from the device pattern map (wfs_eval_pattern_rows), stored behind the host rows
    h->dev_row_ins.clear(); h->h_ins_type.assign(type, type + n); h->n_host_rows = n_cdf;
    {
        std::vector<i32> rows(cdf_row, cdf_row + n);
        for (i64 i = 0; i < n; i++) if (rows[i] < 0) { rows[i] = (i32)(n_cdf + (i64)h->dev_row_ins.size()); h->dev_row_ins.push_back((i32)i); }
        h->dev_rows_pending = !h->dev_row_ins.empty(); h->ins_aft_set = false; h->n_diff_rows = 0; h->ins_diff.clear();
        const size_t total = (size_t)n_cdf + h->dev_row_ins.size();
        TRY(ensure(h, h->cdf_table, total * h->cfg.n_tpc * 8)); TRY(ensure(h, h->cdf_guide, total * (CDF_G + 2) * 2));
        TRY(upload(h, h->ins_cdfrow, rows.data(), (size_t)n * 4)); HIPCHK(hipStreamSynchronize(h->stream));
    }
    TRY(upload(h, h->cdf_table, cdf_table, (size_t)n_cdf * h->cfg.n_tpc * 8));
    {   // guide table of every channel-CDF row: guide[c] = first channel whose cumulative probability exceeds c / CDF_G
        const int nch = h->cfg.n_tpc;
        std::vector<unsigned short> guide((size_t)n_cdf * (CDF_G + 2));
        for (int r = 0; r < n_cdf; r++) {
            const double *row = cdf_table + (size_t)r * nch; int ch = 0;
            for (int c = 0; c <= CDF_G + 1; c++) {
                const double x = (double)c / CDF_G;
                while (ch < nch - 1 && row[ch] <= x) ch++;
                guide[(size_t)r * (CDF_G + 2) + c] = (unsigned short)ch;
            }
        }
        TRY(upload(h, h->cdf_guide, guide.data(), guide.size() * 2));
    }
    TRY(upload(h, h->em_off, em_off.data(), em_off.size() * 8));
    {
        std::vector<i32> sc((size_t)S), sm((size_t)S, 0); std::vector<i64> st((size_t)S);
        for (i64 q = 0; q < S; q++) {
            const i64 ps = q % PS;
            sm[q] = q >= PS ? 1 : 0;
            if (set_off[ps + 1] == set_off[ps]) { sc[q] = 0; st[q] = 0; continue; }      // an unused set number: no instructions, no photons, no tiles
            i64 t0 = time[set_list[set_off[ps]]];
            for (i64 k = set_off[ps]; k < set_off[ps + 1]; k++) t0 = std::min<i64>(t0, time[set_list[k]]);
            sc[q] = cluster[set_list[set_off[ps]]]; st[q] = t0;
        }
        TRY(upload(h, h->ins_set, ins_set.data(), (size_t)n * 4)); TRY(upload(h, h->set_ins_off, set_off.data(), set_off.size() * 8));
        TRY(upload(h, h->set_ins_list, set_list.data(), (size_t)n * 4));
        TRY(upload(h, h->set_cluster, sc.data(), (size_t)S * 4)); TRY(upload(h, h->set_t0, st.data(), (size_t)S * 8));
        TRY(upload(h, h->set_mode, sm.data(), (size_t)S * 4));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    TRY(load_clusters(h, n, cluster, tmin, gid));
    h->injected = false; h->optical = false; h->batch_loaded = true; h->ran = false; h->gen_done = false; h->ins_models = false; h->ins_gg_set = false;
    return WFS_OK;
} WFS_CATCH(h)

// ---- model variants of the photon delays: S1 'custom' recoil models and optical propagation (s1.py:162-260), S2 garfield
// luminescence and optical propagation (s2.py:380-557).  Each adds one independent integer-truncated term; the host builds
// its probability mass function, the table of the sum is the convolution with the terms wfs_config describes.
int wfs_set_delay_models(wfs_handle *h, int32_t n_tables, const int32_t *base, const int64_t *pmf_off, const double *pmf, const int32_t *vmin)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->tables_set) return h->fail(WFS_E_STATE, "wfs_set_tables must be called first");
    if (n_tables < 0 || n_tables > 65536 || (n_tables > 0 && (!base || !pmf_off || !pmf || !vmin))) return h->fail(WFS_E_INVALID, "wfs_set_delay_models: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    for (auto &b : h->x_alias) if (b.p) hipFree(b.p);
    h->x_alias.assign((size_t)n_tables, DevBuf{});
    h->h_tabs.assign((size_t)n_tables + 2, AliasTab{});
    h->n_user_tabs = n_tables; h->ins_models = false;
    for (int k = 0; k < n_tables; k++) {
        if (base[k] < 0 || base[k] > 3 || pmf_off[k + 1] <= pmf_off[k]) return h->fail(WFS_E_INVALID, "wfs_set_delay_models: bad base or empty pmf");
        Pmf e; e.vmin = vmin[k]; e.p.assign(pmf + pmf_off[k], pmf + pmf_off[k + 1]);
        std::vector<double> cum; long v0;
        cum_of_pmf(pmf_conv(h->base_pmf[base[k]], e), cum, v0);
        if (cum.size() > 65000) return h->fail(WFS_E_CAPACITY, "delay table too long");
        TRY(upload_disc(h, h->x_alias[k], cum, (int)v0, h->h_tabs[k]));
    }
    h->h_tabs[n_tables] = h->dev.tab_s1; h->h_tabs[n_tables + 1] = h->dev.tab_s2;       // "-1": the default table of the type
    TRY(upload(h, h->d_tabs, h->h_tabs.data(), h->h_tabs.size() * sizeof(AliasTab)));
    HIPCHK(hipStreamSynchronize(h->stream));
    return WFS_OK;
} WFS_CATCH(h)

// ---- pattern maps evaluated on the device (make_patternmap, load_resource.py:403-435; WeightedNearestNeighbors) ----
static int map_set_grid(wfs_handle *h, wfs_handle::PatternMap &m, int dims, int min_dims, const int32_t *n_nodes, const double *lo, const double *hi, size_t &nodes)
{
    if (dims < min_dims || dims > 3 || !n_nodes || !lo || !hi) return h->fail(WFS_E_INVALID, "map: bad number of dimensions / missing grid");
    nodes = 1; double diag2 = 0;
    for (int a = 0; a < 3; a++) { m.n[a] = 1; m.lo[a] = 0; m.hgrid[a] = 1; m.w[a] = 0; }
    for (int a = 0; a < dims; a++) {
        if (n_nodes[a] < 2 || !(hi[a] > lo[a])) return h->fail(WFS_E_INVALID, "map: every axis needs at least 2 nodes and hi > lo");
        m.n[a] = n_nodes[a]; m.lo[a] = lo[a]; m.hgrid[a] = (hi[a] - lo[a]) / (n_nodes[a] - 1); nodes *= (size_t)n_nodes[a];
        diag2 += m.hgrid[a] * m.hgrid[a];
    }
    // the 2 * dims nearest nodes lie within one cell diagonal (a cell has 2^dims >= 2 * dims corners): candidate block per axis
    for (int a = 0; a < dims; a++) m.w[a] = (i32)ceil(sqrt(diag2) / m.hgrid[a]) + 1;
    m.dims = dims; m.n_points = 0;
    return WFS_OK;
}

static void map_args(const wfs_handle::PatternMap &pm, MapArgs &m)
{
    m.dims = pm.dims; for (int q = 0; q < 3; q++) { m.n[q] = pm.n[q]; m.w[q] = pm.w[q]; m.lo[q] = pm.lo[q]; m.h[q] = pm.hgrid[q]; }
    m.n_points = pm.n_points; m.points = pm.n_points ? pm.points.as<double>() : nullptr;
    m.values = pm.values.as<float>(); m.n_map_ch = pm.n_map_ch;
    m.cell_start = pm.cnx ? pm.cell_start.as<i32>() : nullptr; m.cell_pts = pm.cnx ? pm.cell_pts.as<i32>() : nullptr;
    m.cnx = pm.cnx; m.cny = pm.cny; m.cell_lo[0] = pm.cell_lo[0]; m.cell_lo[1] = pm.cell_lo[1]; m.cell_h = pm.cell_h;
}

static void launch_neighbours(wfs_handle *h, const MapArgs &m)
{
    if (m.points) { Timer t(h, "k_map_neighbours_points"); hipLaunchKernelGGL(k_map_neighbours_points, dim3(nblocks(m.n_rows, 4)), dim3(256), 0, h->stream, m); }
    else { Timer t(h, "k_map_neighbours"); hipLaunchKernelGGL(k_map_neighbours, dim3(nblocks(m.n_rows, 128)), dim3(128), 0, h->stream, m); }
}

int wfs_set_pattern_map(wfs_handle *h, int32_t which, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi,
                        const float *values, int32_t n_map_channels)
try {
    if (!h) return WFS_E_INVALID;
    if (which != 1 && which != 2) return h->fail(WFS_E_INVALID, "wfs_set_pattern_map: which = 1 (S1 map) or 2 (S2 map)");
    auto &m = h->pmap[which - 1];
    if (dims == 0) { m.set = false; return WFS_OK; }
    if (!values || n_map_channels <= 0 || n_map_channels > h->cfg.n_tpc) return h->fail(WFS_E_INVALID, "wfs_set_pattern_map: at most n_tpc channels");
    HIPCHK(hipSetDevice(h->device));
    size_t nodes = 0;
    TRY(map_set_grid(h, m, dims, 2, n_nodes, lo, hi, nodes));
    m.n_map_ch = n_map_channels;
    TRY(upload(h, m.values, values, nodes * (size_t)n_map_channels * 4));
    HIPCHK(hipStreamSynchronize(h->stream));
    m.set = true;
    return WFS_OK;
} WFS_CATCH(h)

// a pattern map on an irregular coordinate system (a list of points; straxen queries a KD-tree for the 2 * dims nearest)
int wfs_set_pattern_map_points(wfs_handle *h, int32_t which, int32_t dims, int64_t n_points, const double *points, const float *values, int32_t n_map_channels)
try {
    if (!h) return WFS_E_INVALID;
    if (which != 1 && which != 2) return h->fail(WFS_E_INVALID, "wfs_set_pattern_map_points: which = 1 (S1 map) or 2 (S2 map)");
    auto &m = h->pmap[which - 1];
    if (dims < 2 || dims > 3 || n_points < 2 * dims || !points || !values || n_map_channels <= 0 || n_map_channels > h->cfg.n_tpc)
        return h->fail(WFS_E_INVALID, "wfs_set_pattern_map_points: 2 or 3 dimensions, at least 2 * dims points, at most n_tpc channels");
    HIPCHK(hipSetDevice(h->device));
    m.dims = dims; m.n_points = n_points; m.n_map_ch = n_map_channels;
    TRY(upload(h, m.points, points, (size_t)n_points * dims * 8));
    TRY(upload(h, m.values, values, (size_t)n_points * (size_t)n_map_channels * 4));
    m.cnx = m.cny = 0;
    if (dims == 2) {
        // cell index for the per-electron searches of the transverse diffusion (k_diffuse_patterns): square cells of about two points
        double lo[2] = {points[0], points[1]}, hi[2] = {points[0], points[1]};
        for (i64 q = 0; q < n_points; q++) for (int a = 0; a < 2; a++) { lo[a] = std::min(lo[a], points[2 * q + a]); hi[a] = std::max(hi[a], points[2 * q + a]); }
        const double span = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), 1e-9);
        const double ch = span / std::max<double>(1.0, std::ceil(std::sqrt((double)n_points / 2.0)));
        const i32 cnx = (i32)std::floor((hi[0] - lo[0]) / ch) + 1, cny = (i32)std::floor((hi[1] - lo[1]) / ch) + 1;
        std::vector<i32> start((size_t)cnx * cny + 1, 0), pts((size_t)n_points), cell((size_t)n_points);
        for (i64 q = 0; q < n_points; q++) {
            i32 cx = (i32)std::floor((points[2 * q] - lo[0]) / ch), cy = (i32)std::floor((points[2 * q + 1] - lo[1]) / ch);
            cx = std::min(std::max(cx, 0), cnx - 1); cy = std::min(std::max(cy, 0), cny - 1);
            cell[(size_t)q] = cx * cny + cy; start[(size_t)cell[(size_t)q] + 1]++;
        }
        for (size_t c = 0; c + 1 < start.size(); c++) start[c + 1] += start[c];
        { std::vector<i32> cur(start.begin(), start.end() - 1); for (i64 q = 0; q < n_points; q++) pts[(size_t)cur[(size_t)cell[(size_t)q]]++] = (i32)q; }      // (ascending point index inside a cell)
        TRY(upload(h, m.cell_start, start.data(), start.size() * 4)); TRY(upload(h, m.cell_pts, pts.data(), pts.size() * 4));
        m.cnx = cnx; m.cny = cny; m.cell_lo[0] = lo[0]; m.cell_lo[1] = lo[1]; m.cell_h = ch;
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    m.set = true;
    return WFS_OK;
} WFS_CATCH(h)

// s2_aft_sigma (s2.py:660-665): the skew-normal factor of every instruction of the loaded batch (NaN: none), applied to the
// rows the device evaluates.  Call between wfs_load_instructions and wfs_eval_pattern_rows.
int wfs_set_instruction_aft(wfs_handle *h, int64_t n, const double *factor)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->batch_loaded || h->injected || h->optical || n != h->n_ins) return h->fail(WFS_E_STATE, "wfs_set_instruction_aft follows wfs_load_instructions of the same batch");
    if (!factor) { h->ins_aft_set = false; return WFS_OK; }
    HIPCHK(hipSetDevice(h->device));
    TRY(upload(h, h->ins_aft, factor, (size_t)n * 8));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ins_aft_set = true;
    return WFS_OK;
} WFS_CATCH(h)

// diffusion_constant_transverse with enable_field_dependencies['diffusion_transverse_map'] (S2.s2_pattern_map_diffuse, s2.py:560-613):
// sigma_r[i], sigma_a[i] = sqrt(2 D t) of instruction i along / across the radius (cm; NaN: not this path).  The pattern of such an
// instruction is averaged over its surviving electrons inside wfs_run (k_diffuse_patterns), not by wfs_eval_pattern_rows.
// Needs the S2 pattern map on the device (a 2-D regular grid or point list) and the instruction loaded with cdf_row = -1.
int wfs_set_instruction_diffusion(wfs_handle *h, int64_t n, const double *sigma_r, const double *sigma_a, double tpc_radius)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->batch_loaded || h->injected || h->optical || n != h->n_ins || !sigma_r || !sigma_a) return h->fail(WFS_E_STATE, "wfs_set_instruction_diffusion follows wfs_load_instructions of the same batch");
    const auto &pm = h->pmap[1];
    if (!pm.set || pm.dims != 2 || (pm.n_points && !pm.cnx)) return h->fail(WFS_E_STATE, "wfs_set_instruction_diffusion needs a two-dimensional S2 pattern map on the device (wfs_set_pattern_map / wfs_set_pattern_map_points)");
    HIPCHK(hipSetDevice(h->device));
    h->ins_diff.assign((size_t)n, 0);
    std::vector<char> is_dev((size_t)n, 0);
    for (i32 i : h->dev_row_ins) is_dev[i] = 1;
    std::vector<i32> rows; std::vector<i64> ids;
    for (size_t k = 0; k < h->dev_row_ins.size(); k++) {
        const i32 i = h->dev_row_ins[k];
        if (h->h_ins_type[i] == 1 || !(sigma_r[i] == sigma_r[i]) || !(sigma_a[i] == sigma_a[i])) continue;
        h->ins_diff[i] = 1; rows.push_back(i); ids.push_back(h->n_host_rows + (i64)k);
    }
    for (i64 i = 0; i < n; i++)
        if (h->h_ins_type[i] != 1 && sigma_r[i] == sigma_r[i] && !is_dev[i]) return h->fail(WFS_E_STATE, "transverse diffusion: the instruction must take its pattern from the device map (cdf_row = -1)");
    h->n_diff_rows = (i64)rows.size(); h->diff_r2 = tpc_radius * tpc_radius;
    TRY(upload(h, h->ins_sigr, sigma_r, (size_t)n * 8)); TRY(upload(h, h->ins_siga, sigma_a, (size_t)n * 8));
    TRY(upload(h, h->diff_row_ins, rows.data(), rows.size() * 4)); TRY(upload(h, h->diff_row_id, ids.data(), ids.size() * 8));
    HIPCHK(hipStreamSynchronize(h->stream));
    return WFS_OK;
} WFS_CATCH(h)

// ---- scalar maps on the device: LCE (s1.py:125), S2 correction / SE gain (s2.py:193-234), longitudinal diffusion (s2.py:170),
// field-dependence splines (s2.py:150, 248), field distortion maps (s2.py:41, 66) ----
static int smap_new(wfs_handle *h, std::unique_ptr<wfs_handle::ScalarMap> m, int32_t *map_id)
{
    HIPCHK(hipStreamSynchronize(h->stream));
    h->smaps.push_back(std::move(m));
    *map_id = (int32_t)h->smaps.size() - 1;
    return WFS_OK;
}

static int scalar_map_grid(wfs_handle *h, int kind, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi, const double *values, int32_t n_values, int32_t *map_id)
{
    if (!h || !map_id || !values) return WFS_E_INVALID;
    if (n_values < 1 || n_values > 4096) return h->fail(WFS_E_INVALID, "scalar map: 1 .. 4096 values per node");
    HIPCHK(hipSetDevice(h->device));
    std::unique_ptr<wfs_handle::ScalarMap> m(new wfs_handle::ScalarMap()); size_t nodes = 0;
    m->kind = kind; m->nv = n_values;
    TRY(map_set_grid(h, m->g, dims, 1, n_nodes, lo, hi, nodes));
    TRY(upload(h, m->g.values, values, nodes * (size_t)n_values * 8));
    return smap_new(h, std::move(m), map_id);
}

int wfs_scalar_map_grid(wfs_handle *h, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi, const double *values, int32_t *map_id)
try { return scalar_map_grid(h, 0, dims, n_nodes, lo, hi, values, 1, map_id); } WFS_CATCH(h)

int wfs_scalar_map_grid_array(wfs_handle *h, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi, const double *values, int32_t n_values, int32_t *map_id)
try { return scalar_map_grid(h, 0, dims, n_nodes, lo, hi, values, n_values, map_id); } WFS_CATCH(h)

int wfs_scalar_map_linear(wfs_handle *h, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi, const double *values, int32_t n_values, int32_t *map_id)
try { return scalar_map_grid(h, 3, dims, n_nodes, lo, hi, values, n_values, map_id); } WFS_CATCH(h)

static int scalar_map_points(wfs_handle *h, int32_t dims, int64_t n_points, const double *points, const double *values, int32_t n_values, int32_t *map_id)
{
    if (!h || !map_id || !values || !points || dims < 1 || dims > 3 || n_points < 2 * dims) return h ? h->fail(WFS_E_INVALID, "wfs_scalar_map_points: 1..3 dimensions, at least 2 * dims points") : WFS_E_INVALID;
    if (n_values < 1 || n_values > 4096) return h->fail(WFS_E_INVALID, "scalar map: 1 .. 4096 values per node");
    HIPCHK(hipSetDevice(h->device));
    std::unique_ptr<wfs_handle::ScalarMap> m(new wfs_handle::ScalarMap());
    m->kind = 1; m->nv = n_values; m->g.dims = dims; m->g.n_points = n_points;
    TRY(upload(h, m->g.points, points, (size_t)n_points * dims * 8));
    TRY(upload(h, m->g.values, values, (size_t)n_points * (size_t)n_values * 8));
    return smap_new(h, std::move(m), map_id);
}

int wfs_scalar_map_points(wfs_handle *h, int32_t dims, int64_t n_points, const double *points, const double *values, int32_t *map_id)
try { return scalar_map_points(h, dims, n_points, points, values, 1, map_id); } WFS_CATCH(h)

int wfs_scalar_map_points_array(wfs_handle *h, int32_t dims, int64_t n_points, const double *points, const double *values, int32_t n_values, int32_t *map_id)
try { return scalar_map_points(h, dims, n_points, points, values, n_values, map_id); } WFS_CATCH(h)

int wfs_scalar_map_spline(wfs_handle *h, int32_t nx, const double *tx, int32_t ny, const double *ty, int32_t kx, int32_t ky, const double *c, int32_t *map_id)
try {
    if (!h || !map_id || !tx || !ty || !c) return WFS_E_INVALID;
    if (kx < 1 || kx > 5 || ky < 1 || ky > 5 || nx < 2 * kx + 2 || ny < 2 * ky + 2) return h->fail(WFS_E_INVALID, "wfs_scalar_map_spline: degrees 1..5, at least 2 (k + 1) knots per axis");
    HIPCHK(hipSetDevice(h->device));
    std::unique_ptr<wfs_handle::ScalarMap> m(new wfs_handle::ScalarMap());
    m->kind = 2; m->nx = nx; m->ny = ny; m->kx = kx; m->ky = ky;
    TRY(upload(h, m->tx, tx, (size_t)nx * 8)); TRY(upload(h, m->ty, ty, (size_t)ny * 8));
    TRY(upload(h, m->c, c, (size_t)(nx - kx - 1) * (size_t)(ny - ky - 1) * 8));
    return smap_new(h, std::move(m), map_id);
} WFS_CATCH(h)

static int scalar_map_eval(wfs_handle *h, int32_t map_id, int64_t n, const double *pos, double *out, int32_t n_values)
{
    if (!h) return WFS_E_INVALID;
    if (map_id < 0 || (size_t)map_id >= h->smaps.size() || n < 0 || (n && (!pos || !out))) return h->fail(WFS_E_INVALID, "wfs_scalar_map_eval: unknown map or missing arrays");
    const auto &sm = *h->smaps[map_id];
    if (sm.nv != n_values) return h->fail(WFS_E_INVALID, "wfs_scalar_map_eval: the map has another number of values per node (wfs_scalar_map_eval_array)");
    if (n == 0) return WFS_OK;
    HIPCHK(hipSetDevice(h->device));
    const int dims = sm.kind == 2 ? 2 : sm.g.dims, nv = sm.nv;
    TRY(upload(h, h->smap_pos, pos, (size_t)n * dims * 8)); TRY(ensure(h, h->smap_out, (size_t)n * nv * 8));
    if (sm.kind == 2) {
        SplineArgs a{sm.nx, sm.ny, sm.kx, sm.ky, sm.tx.as<double>(), sm.ty.as<double>(), sm.c.as<double>(), n, h->smap_pos.as<double>(), h->smap_out.as<double>()};
        Timer t(h, "k_map_spline"); hipLaunchKernelGGL(k_map_spline, dim3(nblocks(n, 128)), dim3(128), 0, h->stream, a);
    } else if (sm.kind == 3) {
        LinearMapArgs a{};
        a.dims = dims; a.nv = nv; a.values = sm.g.values.as<double>(); a.n_rows = n; a.pos = h->smap_pos.as<double>(); a.out = h->smap_out.as<double>();
        for (int q = 0; q < 3; q++) { a.n[q] = sm.g.n[q]; a.lo[q] = sm.g.lo[q]; a.h[q] = sm.g.hgrid[q]; }
        Timer t(h, "k_map_linear"); hipLaunchKernelGGL(k_map_linear, dim3(nblocks(n * nv, 128)), dim3(128), 0, h->stream, a);
    } else {
        TRY(ensure(h, h->smap_nb_idx, (size_t)n * MAP_K * 8)); TRY(ensure(h, h->smap_nb_w, (size_t)n * MAP_K * 8));
        MapArgs m{};
        map_args(sm.g, m);
        m.n_rows = n; m.pos = h->smap_pos.as<double>(); m.nb_idx = h->smap_nb_idx.as<i64>(); m.nb_w = h->smap_nb_w.as<double>();
        launch_neighbours(h, m);
        { Timer t(h, "k_map_scalar"); hipLaunchKernelGGL(k_map_scalar, dim3(nblocks(n * nv, 128)), dim3(128), 0, h->stream, m, sm.g.values.as<double>(), h->smap_out.as<double>(), nv); }
    }
    HIPCHK(hipMemcpyAsync(out, h->smap_out.p, (size_t)n * nv * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    return WFS_OK;
}

int wfs_scalar_map_eval(wfs_handle *h, int32_t map_id, int64_t n, const double *pos, double *out)
try { return scalar_map_eval(h, map_id, n, pos, out, 1); } WFS_CATCH(h)

int wfs_scalar_map_eval_array(wfs_handle *h, int32_t map_id, int64_t n, const double *pos, double *out, int32_t n_values)
try { return scalar_map_eval(h, map_id, n, pos, out, n_values); } WFS_CATCH(h)

int wfs_eval_pattern_rows(wfs_handle *h, int64_t n, const float *x, const float *y, const float *z)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->batch_loaded || h->injected || h->optical || n != h->n_ins || !x || !y || !z) return h->fail(WFS_E_STATE, "wfs_eval_pattern_rows follows wfs_load_instructions of the same batch");
    if (h->dev_row_ins.empty()) return WFS_OK;
    HIPCHK(hipSetDevice(h->device));
    const int nch = h->cfg.n_tpc;
    TRY(upload(h, h->map_x, x, (size_t)n * 4)); TRY(upload(h, h->map_y, y, (size_t)n * 4)); TRY(upload(h, h->map_z, z, (size_t)n * 4));
    i64 first = h->n_host_rows;
    // rows were numbered in instruction order; evaluate them map by map, each over its own (ordered) subset
    std::vector<i32> rows_of[2]; std::vector<i64> row_id[2];
    for (size_t k = 0; k < h->dev_row_ins.size(); k++) {
        if (!h->ins_diff.empty() && h->ins_diff[h->dev_row_ins[k]]) continue;       // averaged over its electrons inside wfs_run
        const int w = h->h_ins_type[h->dev_row_ins[k]] == 1 ? 0 : 1; rows_of[w].push_back(h->dev_row_ins[k]); row_id[w].push_back(first + (i64)k);
    }
    for (int w = 0; w < 2; w++) {
        if (rows_of[w].empty()) continue;
        const auto &pm = h->pmap[w];
        const i64 nr = (i64)rows_of[w].size();
        TRY(upload(h, h->map_row_ins[w], rows_of[w].data(), (size_t)nr * 4)); TRY(upload(h, h->map_row_id[w], row_id[w].data(), (size_t)nr * 8));
        TRY(ensure(h, h->map_nb_idx[w], (size_t)nr * MAP_K * 8)); TRY(ensure(h, h->map_nb_w[w], (size_t)nr * MAP_K * 8));
        MapArgs m{};
        map_args(pm, m);
        m.n_rows = nr; m.row_ins = h->map_row_ins[w].as<i32>(); m.row_id = h->map_row_id[w].as<i64>();
        m.aft = (w == 1 && h->ins_aft_set) ? h->ins_aft.as<double>() : nullptr; m.n_top = h->cfg.n_top;
        m.x = h->map_x.as<float>(); m.y = h->map_y.as<float>(); m.z = h->map_z.as<float>();
        m.nb_idx = h->map_nb_idx[w].as<i64>(); m.nb_w = h->map_nb_w[w].as<double>();
        m.cdf_table = h->cdf_table.as<double>(); m.cdf_guide = h->cdf_guide.as<unsigned short>(); m.gains = h->t_gains.as<double>();
        launch_neighbours(h, m);
        { Timer t(h, "k_map_rows"); hipLaunchKernelGGL(k_map_rows, dim3((unsigned)nr), dim3(256), (size_t)nch * 8, h->stream, m, nch); }
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    h->dev_rows_pending = false; h->gen_done = false;
    return WFS_OK;
} WFS_CATCH(h)

// channel CDF rows of the loaded batch as the generator sees them (host rows followed by the device-evaluated ones)
int wfs_copy_cdf_rows(wfs_handle *h, int32_t *cdf_row, double *cdf_table, int64_t cap_rows)
try {
    if (!h || !h->batch_loaded || h->injected || h->optical) return WFS_E_STATE;
    if (h->dev_rows_pending) return h->fail(WFS_E_STATE, "wfs_eval_pattern_rows has not been called for this batch");
    const i64 total = h->n_host_rows + (i64)h->dev_row_ins.size();
    if (cap_rows < total) return h->fail(WFS_E_CAPACITY, "cdf row buffer too small");
    HIPCHK(hipSetDevice(h->device));
    if (cdf_row) HIPCHK(hipMemcpy(cdf_row, h->ins_cdfrow.p, (size_t)h->n_ins * 4, hipMemcpyDeviceToHost));
    if (cdf_table) HIPCHK(hipMemcpy(cdf_table, h->cdf_table.p, (size_t)total * h->cfg.n_tpc * 8, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)

int wfs_set_s1_propagation(wfs_handle *h, int32_t nz, int32_t nu, double u0, double du, const double *top, const double *bottom)
try {
    if (!h) return WFS_E_INVALID;
    if (nz == 0) { h->prop_nz = 0; return WFS_OK; }
    if (nz < 2 || nu < 2 || !(du > 0) || !top || !bottom) return h->fail(WFS_E_INVALID, "wfs_set_s1_propagation: needs a grid of at least 2 x 2 nodes");
    HIPCHK(hipSetDevice(h->device));
    TRY(upload(h, h->prop_top, top, (size_t)nz * nu * 8)); TRY(upload(h, h->prop_bot, bottom, (size_t)nz * nu * 8));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->prop_nz = nz; h->prop_nu = nu; h->prop_u0 = u0; h->prop_du = du;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_set_instruction_models(wfs_handle *h, int64_t n, const int32_t *tab, const int32_t *tab_bottom, const int32_t *prop_zi, const double *prop_zf)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->batch_loaded || h->injected || h->optical || n != h->n_ins) return h->fail(WFS_E_STATE, "wfs_set_instruction_models follows wfs_load_instructions of the same batch");
    if (h->d_tabs.p == nullptr) TRY(wfs_set_delay_models(h, 0, nullptr, nullptr, nullptr, nullptr));
    HIPCHK(hipSetDevice(h->device));
    std::vector<int8_t> type((size_t)n);
    HIPCHK(hipMemcpy(type.data(), h->ins_type.p, (size_t)n, hipMemcpyDeviceToHost));
    std::vector<i32> t((size_t)n), tb((size_t)n), zi((size_t)n, -1); std::vector<double> zf((size_t)n, 0.0);
    for (i64 i = 0; i < n; i++) {
        const i32 dflt = h->n_user_tabs + (type[i] == 1 ? 0 : 1);
        const i32 a = tab ? tab[i] : -1, b = tab_bottom ? tab_bottom[i] : a;
        if (a < -1 || a >= h->n_user_tabs || b < -1 || b >= h->n_user_tabs) return h->fail(WFS_E_INVALID, "delay table index out of range");
        t[i] = a < 0 ? dflt : a; tb[i] = b < 0 ? dflt : b;
        if (prop_zi && prop_zi[i] >= 0) {
            if (h->prop_nz < 2 || prop_zi[i] > h->prop_nz - 2 || !prop_zf) return h->fail(WFS_E_INVALID, "S1 propagation cell out of range (wfs_set_s1_propagation first)");
            zi[i] = prop_zi[i]; zf[i] = prop_zf[i];
        }
    }
    TRY(upload(h, h->ins_tab, t.data(), (size_t)n * 4)); TRY(upload(h, h->ins_tabb, tb.data(), (size_t)n * 4));
    TRY(upload(h, h->ins_pzi, zi.data(), (size_t)n * 4)); TRY(upload(h, h->ins_pzf, zf.data(), (size_t)n * 8));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ins_models = true; h->gen_done = false;
    return WFS_OK;
} WFS_CATCH(h)

// ---- s2_luminescence_model 'garfield_gas_gap' (s2.py:413-483, load_resource.py:284-291) ----
int wfs_set_gas_gap_model(wfs_handle *h, int32_t n_gas_gaps, int32_t n_points, const double *timing_inv_cdf)
try {
    if (!h) return WFS_E_INVALID;
    if (n_gas_gaps == 0) { h->gg_n = 0; return WFS_OK; }
    if (n_gas_gaps < 1 || n_points < 3 || !timing_inv_cdf) return h->fail(WFS_E_INVALID, "wfs_set_gas_gap_model: at least one table of at least 3 points");
    HIPCHK(hipSetDevice(h->device));
    TRY(upload(h, h->gg_inv, timing_inv_cdf, (size_t)n_gas_gaps * n_points * 8));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->gg_n = n_gas_gaps; h->gg_L = n_points;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_set_instruction_gas_gap(wfs_handle *h, int64_t n, const int32_t *table, const double *weight)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->batch_loaded || h->injected || h->optical || n != h->n_ins || !table || !weight) return h->fail(WFS_E_STATE, "wfs_set_instruction_gas_gap follows wfs_load_instructions of the same batch");
    if (h->gg_n < 1) return h->fail(WFS_E_STATE, "wfs_set_gas_gap_model first");
    for (i64 i = 0; i < n; i++) if (table[i] < -1 || table[i] >= h->gg_n) return h->fail(WFS_E_INVALID, "gas gap table index out of range");
    HIPCHK(hipSetDevice(h->device));
    TRY(upload(h, h->ins_gg, table, (size_t)n * 4)); TRY(upload(h, h->ins_ggw, weight, (size_t)n * 8));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->ins_gg_set = true; h->gen_done = false;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_load_photons(wfs_handle *h, int64_t n_sets, const int32_t *set_cluster, const int64_t *set_tmin, const int64_t *set_off,
                     const int64_t *t, const int16_t *ch, const double *gain, const uint8_t *dpe)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->tables_set) return h->fail(WFS_E_STATE, "wfs_set_tables must be called first");
    if (n_sets <= 0 || !set_cluster || !set_tmin || !set_off) return h->fail(WFS_E_INVALID, "wfs_load_photons: null or empty input");
    HIPCHK(hipSetDevice(h->device));
    const i64 P = set_off[n_sets]; const int nch = h->cfg.n_tpc;
    h->n_ins = 0; h->n_sets = n_sets; h->n_tiles = n_sets * nch; h->n_photons = P; h->n_emitters = 0;
    if (h->n_tiles > 0x7fffffffLL) return h->fail(WFS_E_CAPACITY, "too many tiles in one batch");
    // host bucketing: the input is already channel sorted inside each set
    std::vector<i32> count((size_t)h->n_tiles, 0), tmn((size_t)h->n_tiles, 0x7fffffff), tmx((size_t)h->n_tiles, (i32)0x80000000), rel((size_t)P);
    std::vector<i64> t0((size_t)n_sets, 0);
    std::vector<i32> mode((size_t)n_sets, 1);
    for (i64 s = 0; s < n_sets; s++) {
        i64 a = set_off[s], b = set_off[s + 1];
        if (b < a) return h->fail(WFS_E_INVALID, "set_off must be non-decreasing");
        if (b == a) { t0[s] = set_tmin[s]; continue; }
        i64 mn = t[a];
        for (i64 p = a; p < b; p++) mn = std::min(mn, (i64)t[p]);
        t0[s] = mn;
        for (i64 p = a; p < b; p++) {
            if (ch[p] < 0 || ch[p] >= nch) return h->fail(WFS_E_INVALID, "photon channel out of range");
            if (h->h_gains[ch[p]] == 0) return h->fail(WFS_E_INVALID, "photon on a turned-off PMT (gain 0): drop it before wfs_load_photons (pulse.py:89-90)");
            if (p > a && ch[p] < ch[p - 1]) return h->fail(WFS_E_INVALID, "photons of a set must be sorted by channel");
            i64 r = t[p] - mn;
            if (r > 0x7fffffffLL) return h->fail(WFS_E_CAPACITY, "photon time span of a set exceeds 2^31 ns");
            rel[p] = (i32)r;
            size_t tile = (size_t)(s * nch + ch[p]);
            count[tile]++; tmn[tile] = std::min(tmn[tile], (i32)r); tmx[tile] = std::max(tmx[tile], (i32)r);
        }
    }
    h->h_set_off.assign(set_off, set_off + n_sets + 1);
    TRY(upload(h, h->set_cluster, set_cluster, (size_t)n_sets * 4)); TRY(upload(h, h->set_t0, t0.data(), (size_t)n_sets * 8));
    TRY(upload(h, h->set_mode, mode.data(), (size_t)n_sets * 4));
    TRY(upload(h, h->tile_count, count.data(), count.size() * 4)); TRY(upload(h, h->tile_tmin, tmn.data(), tmn.size() * 4));
    TRY(upload(h, h->tile_tmax, tmx.data(), tmx.size() * 4));
    TRY(upload(h, h->ph_gain, gain, (size_t)P * 8));
    {   // DPE flags ride in the code word (only the truth quirk pulse.py:255 reads them in explicit-gain mode)
        std::vector<PhotonRec> recs((size_t)P);
        for (i64 p = 0; p < P; p++) recs[p] = PhotonRec{rel[p], (dpe && dpe[p]) ? (1u << 16) : 0u};
        TRY(upload(h, h->ph, recs.data(), (size_t)P * 8));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    TRY(load_clusters(h, n_sets, set_cluster, set_tmin, nullptr));
    h->injected = true; h->optical = false; h->batch_loaded = true; h->ran = false; h->gen_done = false;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_load_optical(wfs_handle *h, int64_t n, const int64_t *time, const uint32_t *gid, const int32_t *cluster, const int64_t *tmin,
                     const int32_t *first, const int32_t *last, const int32_t *channels, const int64_t *timings, int64_t n_ph, int64_t cutoff)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->tables_set) return h->fail(WFS_E_STATE, "wfs_set_tables must be called first");
    if (n <= 0 || !time || !gid || !cluster || !tmin || !first || !last || (n_ph > 0 && (!channels || !timings)))
        return h->fail(WFS_E_INVALID, "wfs_load_optical: null or empty input");
    HIPCHK(hipSetDevice(h->device));
    const int nch = h->cfg.n_tpc;
    h->n_ins = n; h->n_sets = n; h->n_tiles = n * nch; h->n_emitters = 0; h->ap_active = false;
    if (h->n_tiles > 0x7fffffffLL) return h->fail(WFS_E_CAPACITY, "too many tiles in one batch");
    for (i64 i = 0; i < n; i++)
        if (first[i] < 0 || last[i] < first[i] || last[i] > n_ph) return h->fail(WFS_E_INVALID, "_first/_last out of range");
    // the photons are bucketed by (instruction, channel) on the device: count, scan, place (k_optical_bucket)
    const i64 T = h->n_tiles;
    std::vector<i32> mode((size_t)n, 0);
    TRY(upload(h, h->set_cluster, cluster, (size_t)n * 4)); TRY(upload(h, h->set_t0, time, (size_t)n * 8)); TRY(upload(h, h->set_mode, mode.data(), (size_t)n * 4));
    TRY(upload(h, h->set_gid, gid, (size_t)n * 4)); TRY(upload(h, h->ins_time, time, (size_t)n * 8));
    TRY(upload(h, h->opt_first, first, (size_t)n * 4)); TRY(upload(h, h->opt_last, last, (size_t)n * 4));
    TRY(upload(h, h->opt_ch, channels, (size_t)n_ph * 4)); TRY(upload(h, h->opt_time, timings, (size_t)n_ph * 8));
    TRY(ensure(h, h->tile_count, (size_t)T * 4)); TRY(ensure(h, h->tile_cursor, (size_t)T * 4)); TRY(ensure(h, h->tile_off, (size_t)(T + 1) * 8));
    TRY(ensure(h, h->opt_t, (size_t)n_ph * 4)); TRY(ensure(h, h->opt_item, (size_t)n_ph * 4));
    HIPCHK(hipMemsetAsync(h->tile_count.p, 0, (size_t)T * 4, h->stream)); HIPCHK(hipMemsetAsync(h->tile_cursor.p, 0, (size_t)T * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->scal.p, 0, 512, h->stream));
    OptLoadArgs oa{n, h->opt_first.as<i32>(), h->opt_last.as<i32>(), h->opt_ch.as<i32>(), h->opt_time.as<i64>(), cutoff, h->t_gains.as<double>(),
                   h->tile_count.as<i32>(), h->tile_off.as<i64>(), h->tile_cursor.as<i32>(), h->opt_t.as<i32>(), h->opt_item.as<u32>(), h->scal.as<i64>()};
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_optical_bucket<false>), dim3(nblocks(n, 256)), dim3(256), 0, h->stream, h->dev, oa);
    TRY(scan_into(h, h->tile_count.as<i32>(), T, h->tile_off.as<i64>(), 7, 0));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_optical_bucket<true>), dim3(nblocks(n, 256)), dim3(256), 0, h->stream, h->dev, oa);
    TRY(read_scal(h));
    HIPCHK(hipGetLastError());
    if (h->h_scal[20] == 1) return h->fail(WFS_E_INVALID, "photon channel out of range");
    if (h->h_scal[20] == 2) return h->fail(WFS_E_CAPACITY, "photon time beyond 2^31 ns");
    const i64 P = h->h_scal[7];
    h->n_photons = P;
    TRY(ensure(h, h->ph, (size_t)P * 8));
    TRY(ensure(h, h->tile_tmin, (size_t)h->n_tiles * 4)); TRY(ensure(h, h->tile_tmax, (size_t)h->n_tiles * 4));
    TRY(ensure(h, h->el_stat, (size_t)n * 32)); TRY(ensure(h, h->el_minmax, (size_t)n * 16));
    HIPCHK(hipMemsetAsync(h->el_stat.p, 0, (size_t)n * 32, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    TRY(load_clusters(h, n, cluster, tmin, gid));
    h->injected = false; h->optical = true; h->batch_loaded = true; h->ran = false; h->gen_done = false;
    return WFS_OK;
} WFS_CATCH(h)

// ---------------------------------------------------------------------------------------------- run
static int run_generation(wfs_handle *h)
{
    const WfsDev &d = h->dev;
    const i64 N = h->n_ins, E = h->n_emitters, T = h->n_tiles;
    TRY(ensure(h, h->em_time, (size_t)E * 8)); TRY(ensure(h, h->em_nph, (size_t)E * 4)); TRY(ensure(h, h->em_ins, (size_t)E * 4));
    TRY(ensure(h, h->el_stat, (size_t)N * 32)); TRY(ensure(h, h->el_minmax, (size_t)N * 16));
    HIPCHK(hipMemsetAsync(h->el_stat.p, 0, (size_t)N * 32, h->stream));
    hipLaunchKernelGGL(k_fill_minmax, dim3(nblocks(2 * N, 256)), dim3(256), 0, h->stream, h->el_minmax.as<i64>(), 2 * N);      // (min, max) pairs: (I64_MAX, I64_MIN)
    GenArgs g{};
    g.n_ins = N; g.n_psets = h->n_psets; g.n_emitters = E;
    g.ins_embase = h->ins_embase.as<u32>(); g.ins_set = h->ins_set.as<i32>(); g.set_ins_off = h->set_ins_off.as<i64>(); g.set_ins_list = h->set_ins_list.as<i32>(); g.set_t0 = h->set_t0.as<i64>();
    g.ins_type = h->ins_type.as<int8_t>(); g.ins_time = h->ins_time.as<i64>(); g.ins_amp = h->ins_amp.as<i32>(); g.ins_gid = h->ins_gid.as<u32>();
    g.ins_p = h->ins_p.as<double>(); g.ins_dm = h->ins_dm.as<double>(); g.ins_ds = h->ins_ds.as<double>(); g.ins_sc = h->ins_sc.as<double>();
    g.ins_cdfrow = h->ins_cdfrow.as<i32>(); g.cdf_table = h->cdf_table.as<double>(); g.cdf_guide = h->cdf_guide.as<unsigned short>(); g.em_off = h->em_off.as<i64>();
    g.em_time = h->em_time.as<i64>(); g.em_nph = h->em_nph.as<i32>(); g.em_ins = h->em_ins.as<i32>();
    g.el_stat = h->el_stat.as<double>(); g.el_minmax = h->el_minmax.as<i64>(); g.scal = h->scal.as<i64>();
    const bool ext = h->ins_models;
    if (ext) {
        g.tabs = h->d_tabs.as<AliasTab>(); g.ins_tab = h->ins_tab.as<i32>(); g.ins_tabb = h->ins_tabb.as<i32>();
        g.ins_pzi = h->ins_pzi.as<i32>(); g.ins_pzf = h->ins_pzf.as<double>();
        if (h->prop_nz >= 2) { g.prop_top = h->prop_top.as<double>(); g.prop_bot = h->prop_bot.as<double>(); g.prop_nu = h->prop_nu; g.prop_u0 = h->prop_u0; g.prop_du = h->prop_du; }
        if (h->ins_gg_set && h->gg_n >= 1) {
            TRY(ensure(h, h->ins_ggsum, (size_t)(N + 1) * 8)); HIPCHK(hipMemsetAsync(h->ins_ggsum.p, 0, (size_t)(N + 1) * 8, h->stream));
            g.gg_inv = h->gg_inv.as<double>(); g.gg_n = h->gg_n; g.gg_L = h->gg_L; g.ins_gg = h->ins_gg.as<i32>(); g.ins_ggw = h->ins_ggw.as<double>(); g.ins_ggsum = h->ins_ggsum.as<i64>();
        }
    }
    TRY(ensure(h, h->em_zg, (size_t)E * 8)); HIPCHK(hipMemsetAsync(h->em_zg.p, 0xff, (size_t)E * 8, h->stream)); g.em_zg = h->em_zg.as<double>();
    // tile-local generation (wfs_tilegen.h): which instructions take it is decided before the electrons are drawn -- theirs get no
    // photon numbers.  Debug modes that need the per-photon arrays (currents, generation only) run the generation half alone.
    const bool ap_cfg = h->ap_active;
    h->fuse_on = h->cfg.tile_gen && !h->generic_geom && h->any_s2 && d.gain_spread == 0.0 && (!h->run_sets_given || h->sets_aligned) && h->n_diff_rows == 0;
    (void)ap_cfg;                                // (PMT afterpulses of tile-generated photons are screened inside k_s2_tile)
    h->fuse_full = h->fuse_on && !(h->keep_currents & 5);
    h->n_fused_tiles = 0; h->n_gen_tiles = 0; h->p_fused = 0;
    FuseArgs f{};
    if (h->fuse_on) {
        f.lam_min = h->cfg.tile_gen_min; f.n_ins = N; f.nch = d.n_tpc; f.table_span = (i32)(1u << (32 - d.tab_s2.shift));
        f.set_ins_off = h->run_sets_given ? h->set_ins_off.as<i64>() : nullptr;
        f.n_top = d.n_top;
        if (ext) {       // timing-model variants: the instruction's tables (the longest one bounds every tile buffer)
            f.tabs = g.tabs; f.ins_tab = g.ins_tab; f.ins_tabb = g.ins_tabb; f.ins_gg = g.gg_inv ? g.ins_gg : nullptr;
            for (const AliasTab &t : h->h_tabs) if (t.cell) f.table_span = std::max(f.table_span, (i32)(1u << (32 - t.shift)));
        }
        f.ins_type = g.ins_type; f.ins_amp = g.ins_amp; f.ins_sc = g.ins_sc; f.ins_embase = g.ins_embase; f.ins_gid = g.ins_gid; f.ins_cdfrow = g.ins_cdfrow;
        f.cdf_table = g.cdf_table; f.ins_time = g.ins_time; f.em_off = g.em_off; f.em_time = g.em_time; f.el_minmax = g.el_minmax; f.scal = g.scal;
        TRY(ensure(h, h->ins_fused, (size_t)N * 4)); TRY(ensure(h, h->ins_nsurv, (size_t)N * 4)); TRY(ensure(h, h->ins_bcap, (size_t)N * 4));
        TRY(ensure(h, h->ins_bcap_all, (size_t)N * 4)); TRY(ensure(h, h->et32, (size_t)E * 4));
        HIPCHK(hipMemsetAsync(h->ins_bcap.p, 0, (size_t)N * 4, h->stream)); HIPCHK(hipMemsetAsync(h->ins_bcap_all.p, 0, (size_t)N * 4, h->stream));
        f.ins_fused = h->ins_fused.as<i32>(); f.ins_nsurv = h->ins_nsurv.as<i32>(); f.ins_bcap = h->ins_bcap.as<i32>(); f.ins_bcap_all = h->ins_bcap_all.as<i32>();
        f.et32 = h->et32.as<i32>();
        {
            const i64 n_rows = h->n_host_rows + (i64)h->dev_row_ins.size();
            TRY(ensure(h, h->row_pmax, (size_t)n_rows * 8)); f.row_pmax = h->row_pmax.as<double>();
            Timer t(h, "k_fuse_decide");
            hipLaunchKernelGGL(k_row_pmax, dim3(nblocks(n_rows, 4)), dim3(256), 0, h->stream, f.cdf_table, d.n_tpc, n_rows, h->row_pmax.as<double>());
            hipLaunchKernelGGL(k_fuse_decide, dim3(nblocks(N, 256)), dim3(256), 0, h->stream, f);
        }
        g.ins_fused = f.ins_fused;
    }
    { Timer t(h, "k_s1_hits"); hipLaunchKernelGGL(k_s1_hits, dim3(nblocks(N, 4)), dim3(256), 0, h->stream, d, g); }
    {
        const i64 neb = (E + 255) / 256;
        TRY(ensure(h, h->eblk_ins, (size_t)(neb + 1) * 4)); g.eblk_ins = h->eblk_ins.as<i32>();
        { Timer t(h, "k_emitter_blocks"); hipLaunchKernelGGL(k_emitter_blocks, dim3(nblocks(neb + 1, 256)), dim3(256), 0, h->stream, g, neb); }
        TRY(ensure(h, h->pois_cdf, (size_t)N * POIS_W * 8)); TRY(ensure(h, h->pois_kmin, (size_t)N * 4));
        g.pois_cdf = h->pois_cdf.as<double>(); g.pois_kmin = h->pois_kmin.as<i32>();
        { Timer t(h, "k_poisson_tables"); hipLaunchKernelGGL(k_poisson_tables, dim3((unsigned)N), dim3(POIS_W), 0, h->stream, g, h->pois_cdf.as<double>(), h->pois_kmin.as<i32>()); }
        { Timer t(h, "k_s2_electrons"); hipLaunchKernelGGL(k_s2_electrons, dim3(nblocks(E, 256)), dim3(256), 0, h->stream, d, g); }
        if (h->any_ptrs) { Timer t(h, "k_s2_photons"); hipLaunchKernelGGL(k_s2_photons, dim3(nblocks(E, 256)), dim3(256), 0, h->stream, d, g); }
    }
    if (h->n_diff_rows > 0) {
        // transverse diffusion maps: the pattern of these instructions is the average over their surviving electrons (s2.py:560-613)
        const auto &pm = h->pmap[1];
        const i64 nr = h->n_diff_rows;
        TRY(ensure(h, h->diff_pre, (size_t)nr * pm.n_map_ch * 8));
        MapArgs m{};
        map_args(pm, m);
        m.n_rows = nr; m.row_ins = h->diff_row_ins.as<i32>(); m.row_id = h->diff_row_id.as<i64>();
        m.x = h->map_x.as<float>(); m.y = h->map_y.as<float>(); m.z = h->map_z.as<float>();
        m.cdf_table = h->cdf_table.as<double>(); m.cdf_guide = h->cdf_guide.as<unsigned short>(); m.gains = h->t_gains.as<double>();
        m.aft = h->ins_aft_set ? h->ins_aft.as<double>() : nullptr; m.n_top = h->cfg.n_top;
        DiffArgs q{nr, m.row_ins, h->ins_sigr.as<double>(), h->ins_siga.as<double>(), h->diff_r2, h->diff_pre.as<double>()};
        { Timer t(h, "k_diffuse_patterns"); hipLaunchKernelGGL(k_diffuse_patterns, dim3((unsigned)nr), dim3(256), 0, h->stream, d, g, m, q); }
        m.pre = h->diff_pre.as<double>();
        { Timer t(h, "k_map_rows"); hipLaunchKernelGGL(k_map_rows, dim3((unsigned)nr), dim3(256), (size_t)d.n_tpc * 8, h->stream, m, d.n_tpc); }
    }
    const i64 TP = h->n_psets * d.n_tpc;        // primary tiles; afterpulse tiles follow
    TRY(ensure(h, h->tile_count, (size_t)T * 4)); TRY(ensure(h, h->tile_cursor, (size_t)T * 4));
    HIPCHK(hipMemsetAsync(h->tile_count.p, 0, (size_t)T * 4, h->stream)); HIPCHK(hipMemsetAsync(h->tile_cursor.p, 0, (size_t)T * 4, h->stream));
    TRY(fill32(h, h->tile_tmin, T, 0x7fffffff)); TRY(fill32(h, h->tile_tmax, T, (i32)0x80000000));
    TRY(ensure(h, h->tile_off, (size_t)(T + 1) * 8));
    if (h->fuse_on) {
        // surviving electrons compacted per instruction, tile buffers sized from their time range, photons per tile (Poisson)
        { Timer t(h, "k_fuse_electrons"); hipLaunchKernelGGL(k_fuse_electrons, dim3((unsigned)N), dim3(256), 0, h->stream, d, f); }
        TRY(scan(h, h->ins_bcap_all.as<i32>(), N, h->ins_boff, 23));
        TRY(ensure(h, h->ftiles, (size_t)TP * sizeof(FTile))); TRY(ensure(h, h->tile_done, (size_t)TP * 4));
        f.ins_boff = h->ins_boff.as<i64>(); f.tile_count = h->tile_count.as<i32>(); f.tiles = h->ftiles.as<FTile>();
        f.tile_done = h->tile_done.as<i32>(); f.full = h->fuse_full ? 1 : 0; f.n_list = TP;
        { Timer t(h, "k_tile_counts"); hipLaunchKernelGGL(k_tile_counts, dim3(nblocks(TP, 256)), dim3(256), 0, h->stream, d, f); }
    }
    TRY(scan(h, h->em_nph.as<i32>(), E, h->em_ph_off, 6));
    TRY(read_scal(h));
    const i64 P = h->h_scal[6];                 // photons of the block generator; the tiles' own photons come on top
    if (h->fuse_on) { h->p_fused = h->h_scal[24]; h->n_fused_tiles = h->h_scal[25]; h->n_gen_tiles = h->h_scal[28]; }
    h->n_photons = P + h->p_fused; h->n_ap_photons = 0;
    g.em_ph_off = h->em_ph_off.as<i64>(); g.n_photons = P;
    const bool ap_on = h->ap_active;
    const i64 ap_cap = ap_on ? (P + h->p_fused) / 8 + 65536 : 0;
    TRY(ensure(h, h->ph, (size_t)(P + h->p_fused + ap_cap) * 8));
    TRY(ensure(h, h->ph_idx, (size_t)(P + h->p_fused + ap_cap) * 4)); g.ph_idx = h->ph_idx.as<u32>();
    g.tile_count = h->tile_count.as<i32>(); g.tile_cursor = h->tile_cursor.as<i32>(); g.tile_tmin = h->tile_tmin.as<i32>();
    g.tile_tmax = h->tile_tmax.as<i32>(); g.ph = h->ph.as<PhotonRec>();
    g.tile_off = h->tile_off.as<i64>();
    ApArgs ap{};
    if (ap_on) {
        ap.n = d.n_ap;
        for (int e = 0; e < d.n_ap; e++) {
            const ApElem &s = h->ap[e];
            ap.el[e] = ApElemDev{s.n_bins_delay, s.n_bins_amp, s.amp_2d, s.is_uniform, s.delay_bin, s.amp_bin, s.delay_cdf.as<double>(), s.amp_cdf.as<double>(),
                                 s.delay_sorted, s.amp_sorted, s.delay_guide.as<ApGuide>(), s.amp_guide.as<ApGuide>()};
            ap.prob[e] = s.prob.as<double>();
            ApElem &sm = h->ap[e];
            if (sm.thr_mod != d.pmt_ap_modifier || !sm.thr.p) {           // screening thresholds of the generator (ap_threshold), once per modifier
                std::vector<u32> th(sm.prob_h.size() * 2);
                for (size_t c = 0; c < sm.prob_h.size(); c++) for (int dpe = 0; dpe < 2; dpe++) th[2 * c + dpe] = ap_threshold(sm.prob_h[c], d.pmt_ap_modifier, dpe != 0);
                TRY(upload(h, sm.thr, th.data(), th.size() * 4));
                HIPCHK(hipStreamSynchronize(h->stream));
                sm.thr_mod = d.pmt_ap_modifier;
            }
            ap.thr[e] = sm.thr.as<u32>();
        }
        TRY(ensure(h, h->ap_ins, (size_t)ap_cap * 4)); TRY(ensure(h, h->ap_ch, (size_t)ap_cap * 4)); TRY(ensure(h, h->ap_t, (size_t)ap_cap * 4));
        TRY(ensure(h, h->ap_gain, (size_t)ap_cap * 8)); TRY(ensure(h, h->ph_gain, (size_t)ap_cap * 8)); TRY(ensure(h, h->ap_key, (size_t)ap_cap * 4));
        ap.ap_key = h->ap_key.as<u32>();
        ap.cap = ap_cap; ap.ap_ins = h->ap_ins.as<i32>(); ap.ap_ch = h->ap_ch.as<i32>(); ap.ap_t = h->ap_t.as<i32>(); ap.ap_gain = h->ap_gain.as<double>();
        ap.count = h->scal.as<i64>() + 13;
        TRY(ensure(h, h->ap_cand, (size_t)ap_cap * sizeof(ApCand))); ap.cand = h->ap_cand.as<ApCand>();
    }
    TRY(ensure(h, h->ins_ph0, (size_t)(N + 1) * 8)); g.ins_ph0 = h->ins_ph0.as<i64>();
    { Timer t(h, "k_ins_ph0"); hipLaunchKernelGGL(k_ins_ph0, dim3(nblocks(N + 1, 256)), dim3(256), 0, h->stream, g); }
    TRY(ensure(h, h->ins_sbase, (size_t)N * 4)); g.ins_sbase = h->ins_sbase.as<u32>();
    TRY(ensure(h, h->ins_fullsort, (size_t)N * 4)); g.ins_fullsort = h->ins_fullsort.as<i32>();
    TRY(ensure(h, h->tile_tail, (size_t)TP * 4)); TRY(ensure(h, h->tile_tailbase, (size_t)TP * 4));
    HIPCHK(hipMemsetAsync(h->tile_tail.p, 0, (size_t)TP * 4, h->stream)); HIPCHK(hipMemsetAsync(h->tile_tailbase.p, 0, (size_t)TP * 4, h->stream));
    g.tile_tail = h->tile_tail.as<i32>(); g.tile_tailbase = h->tile_tailbase.as<i32>();
    { Timer t(h, "k_set_bases"); hipLaunchKernelGGL(k_set_bases, dim3(nblocks(h->n_psets, 256)), dim3(256), 0, h->stream, g); }
    if (P > 0) {
        const unsigned nb = (unsigned)((P + GEN_BLOCK - 1) / GEN_BLOCK);
        g.n_blocks = nb;
        TRY(ensure(h, h->blk_e, (size_t)nb * 16)); TRY(ensure(h, h->blk_base, (size_t)nb * d.n_tpc * 4)); TRY(ensure(h, h->blk_cnt, (size_t)nb * d.n_tpc * 2)); TRY(ensure(h, h->blk_ins, (size_t)nb * 4));
        g.blk_e = h->blk_e.as<i64>(); g.blk_base = h->blk_base.as<u32>(); g.blk_cnt = h->blk_cnt.as<unsigned short>(); g.blk_ins = h->blk_ins.as<i32>();
        TRY(ensure(h, h->blk_desc, (size_t)nb * sizeof(BlockDesc))); g.blk_desc = h->blk_desc.as<BlockDesc>();
        { Timer t(h, "k_block_emitters"); hipLaunchKernelGGL(k_block_emitters, dim3(nblocks(nb, 256)), dim3(256), 0, h->stream, g); }
        // XCD x (workgroup id % 8) walks the photon blocks [x * chunk, (x + 1) * chunk) in order: the blocks that share
        // cache lines of a tile (consecutive ranges, k_block_ranges) run close together in time on the same L2
        g.xcd_chunk = (nb + 7) / 8;
        const unsigned nbx = (unsigned)(g.xcd_chunk * 8);
        {   // alias cells of every channel CDF row (host rows + rows from the device maps, the electron-averaged ones included)
            const i64 n_rows = h->n_host_rows + (i64)h->dev_row_ins.size();
            int lg = 1; while ((1 << lg) < d.n_tpc) lg++;
            TRY(ensure(h, h->chan_alias, (size_t)n_rows * ((size_t)8 << lg)));
            g.chan_alias = h->chan_alias.as<uint2>(); g.ch_lg = lg;
            Timer t(h, "k_chan_alias"); hipLaunchKernelGGL(k_chan_alias, dim3((unsigned)n_rows), dim3(64), 0, h->stream, g.cdf_table, d.n_tpc, lg, h->chan_alias.as<uint2>());
        }
        if (g.gg_inv) { Timer t(h, "k_gg_sum"); hipLaunchKernelGGL(k_gg_sum, dim3(nb), dim3(256), 0, h->stream, d, g); }
        { Timer t(h, "k_photon_count"); hipLaunchKernelGGL(k_photon_count, dim3(nbx), dim3(COUNT_TPB), GEN_COUNT_LDS(d.n_tpc, g.ch_lg), h->stream, d, g); }
        { Timer t(h, "k_block_ranges"); hipLaunchKernelGGL(k_block_ranges, dim3(nblocks(TP, 256)), dim3(256), 0, h->stream, d, g); }
        TRY(scan_into(h, h->tile_count.as<i32>(), TP, h->tile_off.as<i64>(), 7, 0));
        const size_t gen_lds = (size_t)gen_fill_lds(d.n_tpc, g.ch_lg, ap_on).total;
        { Timer t(h, "k_photon_fill");
          if (ext && ap_on) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_fill<true, true>), dim3(nbx), dim3(FILL_TPB), gen_lds, h->stream, d, g, ap);
          else if (ext) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_fill<false, true>), dim3(nbx), dim3(FILL_TPB), gen_lds, h->stream, d, g, ap);
          else if (ap_on) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_fill<true, false>), dim3(nbx), dim3(FILL_TPB), gen_lds, h->stream, d, g, ap);
          else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_photon_fill<false, false>), dim3(nbx), dim3(FILL_TPB), gen_lds, h->stream, d, g, ap); }
    } else {
        TRY(scan_into(h, h->tile_count.as<i32>(), TP, h->tile_off.as<i64>(), 7, 0));
    }
    // (photon counts only -- wfs_set_debug bit 2 without bit 4, the electron-afterpulse pre-pass: the tiles' photon numbers are drawn
    // (k_tile_counts) and single photons are recomputed on request (wfs_gather_photon_times); no photon of a tile is generated)
    const bool counts_only = (h->keep_currents & 4) && !(h->keep_currents & 16);
    if (h->fuse_on && h->n_fused_tiles + h->n_gen_tiles > 0 && !counts_only) {
        // the tiles' photons and pulses (before the geometry: the tile time ranges come out of this kernel)
        TRY(ensure(h, h->tbuf, (size_t)h->h_scal[23] * 4 + 64));
        TRY(ensure(h, h->tile_truth, (size_t)T * 8 * 8));
        f.tile_off = h->tile_off.as<i64>(); f.tile_tmin = h->tile_tmin.as<i32>(); f.tile_tmax = h->tile_tmax.as<i32>(); f.tile_truth = h->tile_truth.as<double>();
        f.tbuf = h->tbuf.as<i32>(); f.ph = h->ph.as<PhotonRec>(); f.keep_ph = (h->keep_currents & 16) ? 1 : 0;
        TemplateArg tp;
        for (int k = 0; k < 22; k++) for (int r = 0; r < WFS_DT; r++) tp.t[k * WFS_DT + r] = h->h_templates[r * 22 + k];
        // H table + the waves' counters; the exact form adds the tables of tap_block (the fused form gathers every sample densely)
        size_t lds = (size_t)(WFS_TILE_CHUNK + d.tlen - 1) * d.dt * 8 + 4 * 4 * 4 + (h->cfg.fma ? 16 : TAP_LDS_BYTES(256) + 64);
        lds = (lds + 15) / 16 * 16;
        const int ap_lds_off = (int)lds;                         // afterpulse candidates of the tile (AP variants)
        const ApArgs *app = nullptr;
        if (ap_on) {
            lds += (size_t)AP_STAGE * sizeof(ApCand);
            TRY(ensure(h, h->ap_seg, (size_t)(h->n_fused_tiles + h->n_gen_tiles) * sizeof(ApSeg))); ap.seg = h->ap_seg.as<ApSeg>(); ap.n_seg = h->n_fused_tiles + h->n_gen_tiles;
            f.n_ptiles = TP;
            TRY(upload(h, h->ap_args_dev, &ap, sizeof ap)); app = h->ap_args_dev.as<ApArgs>();      // (`ap` outlives the copy: read_scal below)
        }
        f.sparse_max = h->tap_sparse_max;
        // tiles of up to 2048 photons: photons and pulse in one workgroup (listed from the front); brighter ones -- and every tile in the
        // debug modes -- generation only (listed from the back), their pulses are made by the ordinary kernels below
        if (h->n_fused_tiles > 0) {
            Timer t(h, "k_s2_tile");
            const dim3 grid((unsigned)h->n_fused_tiles);
            if (ap_on) WFS_LAUNCH_F(h, K_S2_TILE_FULL_AP, grid, dim3(256), lds, d, f, tp, app, ap_lds_off, (i64)0, 1, 0);
            else WFS_LAUNCH_F(h, K_S2_TILE_FULL, grid, dim3(256), lds, d, f, tp, app, ap_lds_off, (i64)0, 1, 0);
        }
        if (h->n_gen_tiles > 0) {
            Timer t(h, "k_s2_tile_gen");
            const dim3 grid((unsigned)h->n_gen_tiles);
            if (ap_on) WFS_LAUNCH_F(h, K_S2_TILE_GEN_AP, grid, dim3(256), lds, d, f, tp, app, ap_lds_off, TP - 1, -1, (int)h->n_fused_tiles);
            else WFS_LAUNCH_F(h, K_S2_TILE_GEN, grid, dim3(256), lds, d, f, tp, app, ap_lds_off, TP - 1, -1, (int)h->n_fused_tiles);
        }
    }
    h->fuse_args = f;
    if (ap_on) {
        // afterpulse photons: count per tile of the afterpulse sets, offsets behind the primary photons, place
        { Timer t(h, "k_ap_finish"); hipLaunchKernelGGL(k_ap_finish, dim3(nblocks(ap_cap, 256)), dim3(256), 0, h->stream, d, g, ap); }
        { Timer t(h, "k_ap_count"); hipLaunchKernelGGL(k_ap_count, dim3(nblocks(ap_cap, 256)), dim3(256), 0, h->stream, d, g, ap);
          if (ap.n_seg > 0) hipLaunchKernelGGL(k_ap_seg<false>, dim3(nblocks(ap.n_seg, 4)), dim3(256), 0, h->stream, d, g, ap, (double *)nullptr); }
        TRY(scan_into(h, h->tile_count.as<i32>() + TP, TP, h->tile_off.as<i64>() + TP, 14, P + h->p_fused));      // (behind the slots of every primary tile, the tile-generated ones included)
        { Timer t(h, "k_ap_place"); hipLaunchKernelGGL(k_ap_place, dim3(nblocks(ap_cap, 256)), dim3(256), 0, h->stream, d, g, ap, h->ph_gain.as<double>() - (P + h->p_fused));
          if (ap.n_seg > 0) hipLaunchKernelGGL(k_ap_seg<true>, dim3(nblocks(ap.n_seg, 4)), dim3(256), 0, h->stream, d, g, ap, h->ph_gain.as<double>() - (P + h->p_fused)); }
        TRY(read_scal(h));
        if (h->h_scal[13] > ap_cap) return h->fail(WFS_E_CAPACITY, "more PMT afterpulse photons than 1/8 of the primary photons: afterpulse probability unreasonably high");
        h->n_ap_photons = h->h_scal[14];                             // (the accepted candidates: total of the afterpulse tiles' counts, the scan above)
    }
    if (P > 0 || ap_on) {
        // every tile of the block generator into generation order (k_tile_order): the order the reference's Pulse call sees
        TRY(ensure(h, h->order_list, (size_t)T * 2 * sizeof(OrderRange))); TRY(ensure(h, h->order_list2, (size_t)T * 2 * sizeof(OrderRange)));
        OrderArgs oa{T, TP, h->tile_count.as<i32>(), h->tile_off.as<i64>(), h->ph.as<PhotonRec>(), h->ph_idx.as<u32>(),
                     ap_on ? h->ph_gain.as<double>() : nullptr, P + h->p_fused, h->order_list.as<OrderRange>(), h->order_list2.as<OrderRange>(), h->scal.as<i64>(),
                     (h->fuse_on && h->n_fused_tiles + h->n_gen_tiles > 0) ? h->ins_fused.as<i32>() : nullptr, d.n_tpc,
                     h->tile_cursor.as<i32>(), h->tile_tailbase.as<i32>(), h->ins_fullsort.as<i32>(), h->set_ins_off.as<i64>(), h->set_ins_list.as<i32>()};
        { Timer t(h, "k_tile_order_scan"); hipLaunchKernelGGL(k_tile_order_scan, dim3(nblocks(T, 256)), dim3(256), 0, h->stream, oa); }
        TRY(read_scal(h));
        const i64 n_wave = h->h_scal[30], n_big = h->h_scal[31];
        if (n_wave > 0) { Timer t(h, "k_tile_order"); hipLaunchKernelGGL(k_tile_order, dim3(nblocks(n_wave, 4)), dim3(256), 0, h->stream, oa, n_wave); }
        if (n_big > 0) {
            Timer t(h, "k_tile_order_big");
            hipLaunchKernelGGL(k_tile_order_big, dim3((unsigned)n_big), dim3(256), (size_t)TILE_ORDER_MAX * 24, h->stream, oa);
        }
        const i64 n_huge = h->h_scal[19];
        if (n_huge > 0) {
            // ranges beyond the workgroup sort: compact copies, one segmented radix sort over (order key, position), records to their ranks
            std::vector<OrderRange> rg((size_t)n_huge);
            HIPCHK(hipMemcpy(rg.data(), h->order_list2.as<OrderRange>() + (2 * T - n_huge), (size_t)n_huge * sizeof(OrderRange), hipMemcpyDeviceToHost));
            std::sort(rg.begin(), rg.end(), [](const OrderRange &x, const OrderRange &y) { return x.start < y.start; });      // (appended with atomics: a fixed order for the offsets)
            std::vector<i64> start((size_t)n_huge), cbeg((size_t)n_huge + 1, 0);
            for (i64 k = 0; k < n_huge; k++) { start[(size_t)k] = rg[(size_t)k].start; cbeg[(size_t)k + 1] = cbeg[(size_t)k] + rg[(size_t)k].n; }
            const i64 tot = cbeg[(size_t)n_huge];
            if (tot > 0xffffffffLL) return h->fail(WFS_E_CAPACITY, "more than 2^32 photons in tiles beyond 4096 photons");
            TRY(upload(h, h->huge_start, start.data(), start.size() * 8)); TRY(upload(h, h->huge_cbeg, cbeg.data(), cbeg.size() * 8));
            TRY(ensure(h, h->huge_keys, (size_t)tot * 4)); TRY(ensure(h, h->huge_keys2, (size_t)tot * 4)); TRY(ensure(h, h->huge_vals, (size_t)tot * 4)); TRY(ensure(h, h->huge_vals2, (size_t)tot * 4));
            TRY(ensure(h, h->huge_rec, (size_t)tot * 8)); TRY(ensure(h, h->huge_gain, (size_t)tot * 8));
            HugeOrderArgs ha{n_huge, tot, h->huge_start.as<i64>(), h->huge_cbeg.as<i64>(), oa.ph, oa.ph_idx, oa.ph_gain, oa.gain_first,
                             h->huge_keys.as<u32>(), h->huge_vals.as<u32>(), h->huge_rec.as<PhotonRec>(), h->huge_gain.as<double>()};
            Timer t(h, "k_tile_order_huge");
            hipLaunchKernelGGL(k_order_huge_pack, dim3(nblocks(tot, 256)), dim3(256), 0, h->stream, ha);
            size_t bytes = 0;
            HIPCHK(rocprim::segmented_radix_sort_pairs(nullptr, bytes, h->huge_keys.as<u32>(), h->huge_keys2.as<u32>(), h->huge_vals.as<u32>(), h->huge_vals2.as<u32>(),
                                                       (unsigned)tot, (unsigned)n_huge, h->huge_cbeg.as<i64>(), h->huge_cbeg.as<i64>() + 1, 0u, 32u, h->stream));
            TRY(ensure(h, h->sort_tmp, bytes));
            HIPCHK(rocprim::segmented_radix_sort_pairs(h->sort_tmp.p, bytes, h->huge_keys.as<u32>(), h->huge_keys2.as<u32>(), h->huge_vals.as<u32>(), h->huge_vals2.as<u32>(),
                                                       (unsigned)tot, (unsigned)n_huge, h->huge_cbeg.as<i64>(), h->huge_cbeg.as<i64>() + 1, 0u, 32u, h->stream));
            hipLaunchKernelGGL(k_order_huge_apply, dim3(nblocks(tot, 256)), dim3(256), 0, h->stream, ha, h->huge_keys2.as<u32>(), h->huge_vals2.as<u32>(), h->ph.as<PhotonRec>(), h->ph_idx.as<u32>(),
                               ap_on ? h->ph_gain.as<double>() : nullptr);
        }
    }
    h->gen_args = g;
    return WFS_OK;
}

int wfs_run(wfs_handle *h)
try {
    if (!h) return WFS_E_INVALID;
    if (!h->batch_loaded) return h->fail(WFS_E_STATE, "no batch loaded");
    if (h->dev_rows_pending && !h->injected && !h->optical) return h->fail(WFS_E_STATE, "instructions with cdf_row -1: call wfs_eval_pattern_rows before wfs_run");
    h->ran = false; h->gen_done = false; h->gen_order_ready = false;
    HIPCHK(hipSetDevice(h->device));
    for (auto &t : h->times) { hipEventDestroy(t.a); hipEventDestroy(t.b); }
    h->times.clear();
    const WfsDev &d = h->dev;
    const i64 T = h->n_tiles, S = h->n_sets, C = h->n_clusters;
    HIPCHK(hipMemsetAsync(h->scal.p, 0, 512, h->stream));
    if (h->optical) {
        OpticalArgs oa{T, h->tile_count.as<i32>(), h->tile_off.as<i64>(), h->tile_tmin.as<i32>(), h->tile_tmax.as<i32>(), h->set_gid.as<u32>(),
                       h->opt_t.as<i32>(), h->opt_item.as<u32>(), h->ph.as<PhotonRec>(), h->scal.as<i64>()};
        Timer t(h, "k_optical_finish");
        hipLaunchKernelGGL(k_optical_finish, dim3(nblocks(T, 256)), dim3(256), 0, h->stream, d, oa);
    }
    else if (!h->injected) TRY(run_generation(h));
    else { h->ap_active = false; TRY(scan(h, h->tile_count.as<i32>(), T, h->tile_off, 7)); }
    h->gen_done = true;
    if (h->keep_currents & 4) { HIPCHK(hipStreamSynchronize(h->stream)); return WFS_OK; }      // wfs_set_debug bit 2: photon generation only

    // ---- geometry: tiles -> clusters -> groups -> rows
    const i64 CG = C + 1;         // group slots
    TRY(fill64(h, h->cl_end, C, I64_MIN)); TRY(ensure(h, h->cl_group, (size_t)C * 4));
    TRY(fill64(h, h->grp_lo, CG, I64_MAX)); TRY(fill64(h, h->grp_hi, CG, I64_MIN));
    TRY(ensure(h, h->grp_left, (size_t)CG * 8)); TRY(ensure(h, h->grp_right, (size_t)CG * 8)); TRY(ensure(h, h->grp_ixrand, (size_t)CG * 8));
    TRY(ensure(h, h->grp_gid, (size_t)CG * 4)); HIPCHK(hipMemsetAsync(h->grp_gid.p, 0xff, (size_t)CG * 4, h->stream));
    TRY(fill64(h, h->row_lo, CG * d.n_tpc, I64_MAX)); TRY(fill64(h, h->row_hi, CG * d.n_tpc, I64_MIN));
    TRY(ensure(h, h->acc_len, (size_t)CG * d.n_tpc * 4)); HIPCHK(hipMemsetAsync(h->acc_len.p, 0, (size_t)CG * d.n_tpc * 4, h->stream));
    TRY(ensure(h, h->itv_cap, (size_t)CG * d.row_slots * 4)); TRY(ensure(h, h->active_rows, (size_t)CG * d.row_slots * 4));
    TRY(ensure(h, h->active_tiles, (size_t)T * 4)); TRY(ensure(h, h->sparse_tiles, (size_t)T * 4)); TRY(ensure(h, h->dense_tiles, (size_t)T * 4)); TRY(ensure(h, h->wave_tiles, (size_t)T * 4));
    GeomArgs ga{};
    ga.n_sets = S; ga.n_tiles = T; ga.n_clusters = C; ga.n_gslots = CG;
    ga.tile_count = h->tile_count.as<i32>(); ga.tile_tmin = h->tile_tmin.as<i32>(); ga.tile_tmax = h->tile_tmax.as<i32>();
    ga.set_cluster = h->set_cluster.as<i32>(); ga.set_t0 = h->set_t0.as<i64>(); ga.cl_tmin = h->cl_tmin.as<i64>(); ga.cl_gid = h->cl_gid.as<u32>();
    ga.cl_end = h->cl_end.as<i64>(); ga.cl_group = h->cl_group.as<i32>(); ga.grp_lo = h->grp_lo.as<i64>(); ga.grp_hi = h->grp_hi.as<i64>();
    ga.grp_left = h->grp_left.as<i64>(); ga.grp_right = h->grp_right.as<i64>(); ga.grp_ixrand = h->grp_ixrand.as<i64>(); ga.grp_gid = h->grp_gid.as<u32>();
    ga.row_lo = h->row_lo.as<i64>(); ga.row_hi = h->row_hi.as<i64>(); ga.acc_len = h->acc_len.as<i32>(); ga.itv_cap = h->itv_cap.as<i32>();
    ga.active_rows = h->active_rows.as<i32>(); ga.scal = h->scal.as<i64>(); ga.active_tiles = h->active_tiles.as<i32>(); ga.sparse_tiles = h->sparse_tiles.as<i32>(); ga.dense_tiles = h->dense_tiles.as<i32>(); ga.wave_tiles = h->wave_tiles.as<i32>(); ga.force_dense = ((h->keep_currents & 2) || h->generic_geom) ? 1 : 0; ga.init_has = h->carry_has; ga.init_runmax = h->carry_runmax;
    ga.noise_override = h->n_noise_override ? h->noise_override.as<i64>() : nullptr; ga.n_noise_override = h->n_noise_override;
    const bool tiles_done = !h->injected && !h->optical && h->fuse_full && h->n_fused_tiles > 0;      // pulses made by k_s2_tile (wfs_tilegen.h)
    // resident rows (k_row_pulse): the usual digitiser geometry, a hold-off of at least a chunk and a noise table the fast row loads
    // can walk (as the fast path of k_zle), no HE rows, no debug copies of currents or rows
    // 2 (auto, the default): on when the batch holds at least two photons per (pulse set, channel) slot -- rows that collect several
    // pulses are where the accumulators cost (memset, atomics, two more reads); a batch of sparse S1 or nVeto hits is as fast through
    // them, and the resident path's extra pass over the tiles does not pay there
    // (photons in the photon array: the tiles k_s2_tile made in one go hold theirs in their sample buffers and can never be resident)
    const i64 p_all = h->n_photons - ((!h->injected && !h->optical && h->fuse_full) ? h->p_fused : 0) + ((!h->injected && !h->optical && h->ap_active) ? h->n_ap_photons : 0);
    const bool res_auto = T > 0 && p_all >= 2 * T;
    const int res_mode = h->res_env >= 0 ? h->res_env : h->cfg.row_resident;
    h->res_on = (res_mode == 2 ? res_auto : res_mode != 0) && !(h->keep_currents & 3) && !h->generic_geom && !d.he_rows && 2 * (i64)d.tw + 1 >= 63
                && (!d.enable_noise || d.noise_len >= NOISE_MIN_FAST);
    if (tiles_done || h->res_on) {
        TRY(ensure(h, h->row_cnt, (size_t)CG * d.n_tpc * 4)); TRY(ensure(h, h->row_tile, (size_t)CG * d.n_tpc * 4));
        HIPCHK(hipMemsetAsync(h->row_cnt.p, 0, (size_t)CG * d.n_tpc * 4, h->stream));
        ga.row_cnt = h->row_cnt.as<i32>(); ga.row_tile = h->row_tile.as<i32>();
    }
    if (tiles_done) {
        ga.tile_done = h->tile_done.as<i32>(); ga.n_done = h->n_psets * d.n_tpc;
        ga.ins_bcap = h->ins_bcap.as<i32>(); ga.ins_boff = h->ins_boff.as<i64>();
    }
    if (h->res_on) {
        const size_t nr = (size_t)CG * d.n_tpc;
        TRY(ensure(h, h->row_bad, nr * 4)); TRY(ensure(h, h->fin_len, nr * 4)); TRY(ensure(h, h->res_cnt, nr * 4));
        HIPCHK(hipMemsetAsync(h->row_bad.p, 0, nr * 4, h->stream)); HIPCHK(hipMemsetAsync(h->fin_len.p, 0, nr * 4, h->stream)); HIPCHK(hipMemsetAsync(h->res_cnt.p, 0, nr * 4, h->stream));
        ga.res_on = 1; ga.res_max_len = h->res_max_len; ga.row_bad = h->row_bad.as<i32>(); ga.fin_len = h->fin_len.as<i32>(); ga.res_cnt = h->res_cnt.as<i32>();
        ga.rows_cap = CG * d.row_slots;
        TRY(ensure(h, h->res_long, nr * 4)); ga.res_long = h->res_long.as<i32>();
    }
    { Timer t(h, "k_tile_geom"); hipLaunchKernelGGL(k_tile_geom, dim3(nblocks(T, 1024)), dim3(1024), 0, h->stream, d, ga); }
    { Timer t(h, "k_groups"); hipLaunchKernelGGL(k_groups, dim3(1), dim3(GROUPS_TPB), 0, h->stream, d, ga); }
    { Timer t(h, "k_tile_rows"); hipLaunchKernelGGL(k_tile_rows, dim3(nblocks(T, 256)), dim3(256), 0, h->stream, d, ga); }
    { Timer t(h, "k_group_final"); hipLaunchKernelGGL(k_group_final, dim3(nblocks(CG, 256)), dim3(256), 0, h->stream, d, ga); }
    { Timer t(h, "k_row_len"); hipLaunchKernelGGL(k_row_len, dim3(nblocks(CG * d.row_slots, 1024)), dim3(1024), 0, h->stream, d, ga); }
    TRY(scan(h, h->acc_len.as<i32>(), CG * d.n_tpc, h->acc_off, 8));
    TRY(scan(h, h->itv_cap.as<i32>(), CG * d.row_slots, h->itv_off, 9));
    if (h->res_on) {
        TRY(scan(h, h->fin_len.as<i32>(), CG * d.n_tpc, h->fin_off, 34)); TRY(scan(h, h->res_cnt.as<i32>(), CG * d.n_tpc, h->res_toff, 35));
        // rows are known: tiles onto their row's list or the work list of their class (descriptor slots for every tile: no host
        // round trip for the number of resident ones)
        TRY(ensure(h, h->res_desc, (size_t)std::max<i64>(T, 1) * sizeof(TileDesc)));
        ga.res_toff = h->res_toff.as<i64>(); ga.res_desc = h->res_desc.as<TileDesc>();
        DescArgs da{}; da.tile_off = h->tile_off.as<i64>(); da.set_mode = h->set_mode.as<i32>();
        TRY(ensure(h, h->tile_truth, (size_t)T * 8 * 8));
        PulseArgs pt{}; pt.ph = h->ph.as<PhotonRec>(); pt.tile_truth = h->tile_truth.as<double>();
        pt.ph_gain = (!h->injected && h->ap_active) ? h->ph_gain.as<double>() - h->n_photons : h->ph_gain.as<double>();
        Timer t(h, "k_tile_assign"); hipLaunchKernelGGL(k_tile_assign, dim3(nblocks(T, 256)), dim3(256), 0, h->stream, d, ga, da, pt);
    }
    TRY(read_scal(h));
    if (h->h_scal[1] == 1) return h->fail(WFS_E_CAPACITY, "Pulse cache too long (digitise window of 10^6 samples or more, rawdata.py:219)");
    if (h->h_scal[1] == 2) return h->fail(WFS_E_CAPACITY, "photon time further than 2^31 ns from its instruction");
    if (h->h_scal[1] == 3) return h->fail(WFS_E_STATE, "internal: a tile of k_s2_tile did not fit its sample buffer");
    h->n_front_rows = h->h_scal[2]; h->n_short_rows = h->res_on ? h->h_scal[32] : 0; h->n_res_rows = h->res_on ? h->h_scal[32] + h->h_scal[37] : 0; h->max_res_len = h->h_scal[33]; h->s_fin = h->res_on ? h->h_scal[34] : 0;
    h->n_res_tiles = h->res_on ? h->h_scal[35] : 0; h->s_res = h->res_on ? h->h_scal[36] : 0;
    h->n_groups = h->h_scal[0]; h->n_active_rows = h->n_front_rows + h->n_res_rows;
    if (h->res_on && getenv("WFS_RES_STATS"))
        fprintf(stderr, "resident rows: %lld short + %lld long of %lld rows, %lld tiles of %lld listed + resident, longest %lld samples, %lld finished samples, accumulators %lld samples\n",
                (long long)h->n_short_rows, (long long)(h->n_res_rows - h->n_short_rows), (long long)h->n_active_rows, (long long)h->n_res_tiles,
                (long long)(h->n_res_tiles + h->h_scal[16] + h->h_scal[3] + h->h_scal[11] + h->h_scal[17]), (long long)h->max_res_len, (long long)h->s_fin, (long long)h->h_scal[8]); h->n_sparse_tiles = h->h_scal[3]; h->max_nb = h->h_scal[4]; h->max_tile = h->h_scal[5]; h->max_tile_dense = h->h_scal[15];
    h->n_dense_tiles = h->h_scal[11]; h->max_nb_dense = h->h_scal[12]; h->n_tiny_tiles = h->h_scal[16];
    h->n_wave_tiles = h->h_scal[17];
    h->n_active_tiles = h->n_tiny_tiles + h->n_sparse_tiles + h->n_dense_tiles + h->n_wave_tiles;
    // one work list: tiny tiles, then sparse, then dense
    if (h->n_sparse_tiles > 0)
        HIPCHK(hipMemcpyAsync(h->active_tiles.as<i32>() + h->n_tiny_tiles, h->sparse_tiles.p, (size_t)h->n_sparse_tiles * 4, hipMemcpyDeviceToDevice, h->stream));
    if (h->n_dense_tiles > 0)
        HIPCHK(hipMemcpyAsync(h->active_tiles.as<i32>() + h->n_tiny_tiles + h->n_sparse_tiles, h->dense_tiles.p, (size_t)h->n_dense_tiles * 4, hipMemcpyDeviceToDevice, h->stream));
    if (h->n_wave_tiles > 0)
        HIPCHK(hipMemcpyAsync(h->active_tiles.as<i32>() + h->n_tiny_tiles + h->n_sparse_tiles + h->n_dense_tiles, h->wave_tiles.p, (size_t)h->n_wave_tiles * 4, hipMemcpyDeviceToDevice, h->stream));
    h->s_raw = h->h_scal[8]; h->n_itv_slots = h->h_scal[9];
    h->s_raw_direct = tiles_done ? h->h_scal[26] : 0;      // samples of the rows that are read from a tile buffer in place
    // deterministic processing order of the work lists (they were appended with atomics)
    // (results do not depend on it; sorting keeps profiles and debug dumps reproducible)

    // ---- pulses
    TRY(ensure(h, h->raw, (size_t)h->s_raw * 4 + 64)); HIPCHK(hipMemsetAsync(h->raw.p, 0, (size_t)h->s_raw * 4, h->stream));       // (+64: a lane of k_zle reads 4 samples from its first valid one)
    TRY(ensure(h, h->truth, (size_t)S * 16 * 8)); HIPCHK(hipMemsetAsync(h->truth.p, 0, (size_t)S * 16 * 8, h->stream));
    TRY(ensure(h, h->tminmax, (size_t)S * 16)); TRY(ensure(h, h->tile_truth, (size_t)T * 8 * 8));
    PulseArgs pa{};
    pa.active_tiles = h->active_tiles.as<i32>(); pa.n_active = h->n_active_tiles;
    pa.tile_count = h->tile_count.as<i32>(); pa.tile_tmin = h->tile_tmin.as<i32>(); pa.tile_tmax = h->tile_tmax.as<i32>(); pa.tile_off = h->tile_off.as<i64>();
    pa.set_cluster = h->set_cluster.as<i32>(); pa.set_t0 = h->set_t0.as<i64>(); pa.set_mode = h->set_mode.as<i32>();
    pa.ph = h->ph.as<PhotonRec>();
    pa.ph_gain = (!h->injected && h->ap_active) ? h->ph_gain.as<double>() - h->n_photons : h->ph_gain.as<double>();
    pa.cl_group = h->cl_group.as<i32>(); pa.row_lo = h->row_lo.as<i64>(); pa.acc_off = h->acc_off.as<i64>(); pa.raw = h->raw.as<i32>();
    pa.tile_truth = h->tile_truth.as<double>();
    h->cur_total = 0;
    if ((h->keep_currents & 1) && h->n_active_tiles > 0) {
        // debug: tile lengths in work-list order -> offsets
        std::vector<i32> at((size_t)h->n_active_tiles), tmn((size_t)T), tmx((size_t)T);
        std::vector<i64> t0((size_t)S);
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(at.data(), h->active_tiles.p, at.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tmn.data(), h->tile_tmin.p, tmn.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(tmx.data(), h->tile_tmax.p, tmx.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(t0.data(), h->set_t0.p, t0.size() * 8, hipMemcpyDeviceToHost));
        std::vector<i64> off(at.size() + 1, 0);
        for (size_t k = 0; k < at.size(); k++) {
            i64 set = at[k] / d.n_tpc;
            i64 b0 = floordiv(t0[set] + tmn[at[k]], (i64)d.dt), b1 = floordiv(t0[set] + tmx[at[k]], (i64)d.dt);
            off[k + 1] = off[k] + (b1 - b0 + 1) + d.store_before + d.samples_before + d.store_after + d.samples_after;
        }
        h->cur_total = off.back();
        TRY(upload(h, h->cur_off, off.data(), off.size() * 8));
        TRY(ensure(h, h->currents, (size_t)h->cur_total * 8));
        HIPCHK(hipStreamSynchronize(h->stream));
        pa.currents = h->currents.as<double>(); pa.cur_off = h->cur_off.as<i64>();
    }
    if (h->n_active_tiles > 0) {        // tile descriptors of the whole work list (used by the tiny and the dense kernel)
        TRY(ensure(h, h->tile_desc, (size_t)h->n_active_tiles * sizeof(TileDesc)));
        DescArgs da{pa.active_tiles, h->n_active_tiles, pa.tile_count, pa.tile_tmin, pa.tile_tmax, pa.tile_off, pa.set_cluster, pa.set_t0, pa.set_mode,
                    pa.cl_group, pa.row_lo, pa.acc_off, h->tile_desc.as<TileDesc>()};
        Timer t(h, "k_tile_desc");
        hipLaunchKernelGGL(k_tile_desc, dim3(nblocks(h->n_active_tiles, 256)), dim3(256), 0, h->stream, d, da);
    }
    if (h->n_tiny_tiles > 0) {
        PulseArgs pt = pa;
        pt.desc = h->tile_desc.as<TileDesc>();
        Timer t(h, "k_pulse_tiny");
        WFS_LAUNCH_F(h, K_PULSE_TINY, dim3(nblocks(h->n_tiny_tiles * TINY_LANES, 256)), dim3(256), 0, d, pt, h->n_tiny_tiles);
    }
    if (h->n_wave_tiles > 0) {          // work list order: tiny | sparse | dense | wave
        PulseArgs pw = pa;
        const i64 first = h->n_tiny_tiles + h->n_sparse_tiles + h->n_dense_tiles;
        pw.desc = h->tile_desc.as<TileDesc>() + first;
        if (pw.cur_off) pw.cur_off += first;
        Timer t(h, "k_pulse_wave");
        WFS_LAUNCH_F(h, K_PULSE_WAVE, dim3(nblocks(h->n_wave_tiles, 4)), dim3(256), 0, d, pw, h->n_wave_tiles);
    }
    if (h->n_sparse_tiles > 0) {
        PulseArgs ps = pa;
        ps.active_tiles = h->active_tiles.as<i32>() + h->n_tiny_tiles;
        if (ps.cur_off) ps.cur_off += h->n_tiny_tiles;
        ps.W = (int)((h->max_nb + 7) / 8 * 8); ps.NP = (int)((h->max_tile + 7) / 8 * 8);
        const bool small = h->max_tile <= 256 && h->max_nb <= 256;
        const int tpb = small ? 64 : 256;
        size_t lds = (size_t)ps.NP * 12 + (size_t)8 * (tpb / 64) * 8 + ((size_t)d.dt * ps.W + 8) * 2 + 8 * 4 + std::max((size_t)d.dt * ps.W * 2, (size_t)1 * 260 * 8) + 16;
        lds = (lds + 15) / 16 * 16;
        Timer t(h, "k_pulse_sparse");
        if (small) WFS_LAUNCH_F(h, K_PULSE_SPARSE_64, dim3((unsigned)h->n_sparse_tiles), dim3(64), lds, d, ps);
        else WFS_LAUNCH_F(h, K_PULSE_SPARSE_256, dim3((unsigned)h->n_sparse_tiles), dim3(256), lds, d, ps);
    }
    if (h->n_dense_tiles > 0 && h->generic_geom) {        // any digitiser geometry: one kernel for every tile
        PulseArgs pd = pa;
        pd.desc = h->tile_desc.as<TileDesc>() + h->n_tiny_tiles + h->n_sparse_tiles;
        if (pd.cur_off) pd.cur_off += h->n_tiny_tiles + h->n_sparse_tiles;
        size_t lds = (size_t)(256 + d.tlen - 1) * d.dt * 8 + (size_t)d.dt * d.tlen * 8 + 4 * 8 * 8 + 64;
        lds = (lds + 15) / 16 * 16;
        if (lds > 128 * 1024) return h->fail(WFS_E_CAPACITY, "sample_duration x template length too large for the LDS tables of k_pulse_generic");
        Timer t(h, "k_pulse_generic");
        WFS_LAUNCH_F(h, K_PULSE_GENERIC, dim3((unsigned)h->n_dense_tiles), dim3(256), lds, d, pd);
    } else if (h->n_dense_tiles > 0) {
        PulseArgs pd = pa;
        pd.active_tiles = h->active_tiles.as<i32>() + h->n_tiny_tiles + h->n_sparse_tiles;
        pd.desc = h->tile_desc.as<TileDesc>() + h->n_tiny_tiles + h->n_sparse_tiles;
        if (pd.cur_off) pd.cur_off += h->n_tiny_tiles + h->n_sparse_tiles;
        // chunks of TPB live samples per pass (TPB + 21 start-bin rows of 80 bytes in LDS: 22 KB for 256 -> 7 workgroups per CU)
        const i64 n_live_max = h->max_nb_dense + d.tlen - 1;
        const bool small = n_live_max <= 128;
        const int tpb = small ? 128 : 256;
        const int NWIN_MAX = 8;
        // tiles that fit one register batch: one workgroup per tile walks the chunks with the photons resident in
        // registers; longer tiles: several workgroups per tile (each re-reads the tile's photons for its chunks)
        pd.n_win = (h->max_tile_dense <= (i64)tpb * DENSE_PPT) ? 1 : (int)std::min<i64>(NWIN_MAX, std::max<i64>(1, (n_live_max + tpb - 1) / tpb));
        pd.W = tpb;
        TemplateArg tp;
        for (int k = 0; k < 22; k++) for (int r = 0; r < WFS_DT; r++) tp.t[k * WFS_DT + r] = h->h_templates[(size_t)r * 22 + k];
        size_t lds = (size_t)(tpb + d.tlen - 1) * d.dt * 8 + (size_t)8 * (tpb / 64) * 8 + 64 + TAP_LDS_BYTES(256) + 16;
        lds = (lds + 15) / 16 * 16;
        pd.sparse_max = (h->keep_currents & 2) ? -1 : h->tap_sparse_max;      // (force_dense: every wave takes the dense gather)
        pd.spe_lds = ((size_t)(tpb + d.tlen - 1) * d.dt >= 2001) ? 1 : 0;
        const unsigned grid = (unsigned)(h->n_dense_tiles * pd.n_win);
        Timer t(h, "k_pulse_dense");
        if (pd.n_win == 1) {
            if (small) WFS_LAUNCH_F(h, K_PULSE_128_RES, dim3(grid), dim3(128), lds, d, pd, tp);
            else WFS_LAUNCH_F(h, K_PULSE_256_RES, dim3(grid), dim3(256), lds, d, pd, tp);
        } else {
            if (small) WFS_LAUNCH_F(h, K_PULSE_128_WIN, dim3(grid), dim3(128), lds, d, pd, tp);
            else WFS_LAUNCH_F(h, K_PULSE_256_WIN, dim3(grid), dim3(256), lds, d, pd, tp);
        }
    }

    if (tiles_done && h->h_scal[27] > 0) {   // tiles of k_s2_tile that share their row with other pulses: added into the row's accumulators
        TileAddArgs ta{h->set_cluster.as<i32>(), h->cl_group.as<i32>(), h->row_lo.as<i64>(), h->acc_off.as<i64>(), h->row_cnt.as<i32>(), h->raw.as<i32>()};
        Timer t(h, "k_tile_add");
        hipLaunchKernelGGL(k_tile_add, dim3((unsigned)h->n_fused_tiles), dim3(256), 0, h->stream, d, h->fuse_args, ta);
    }

    // ---- ZLE + records
    const i64 RS = CG * d.row_slots;
    TRY(ensure(h, h->itv_left, (size_t)h->n_itv_slots * 8 + 16)); TRY(ensure(h, h->itv_right, (size_t)h->n_itv_slots * 8 + 16));      // (+16: k_pack reads the first two slots of a row whatever its capacity)
    TRY(ensure(h, h->itv_n, (size_t)RS * 4)); TRY(ensure(h, h->row_nrec, (size_t)RS * 4));
    HIPCHK(hipMemsetAsync(h->itv_n.p, 0, (size_t)RS * 4, h->stream)); HIPCHK(hipMemsetAsync(h->row_nrec.p, 0, (size_t)RS * 4, h->stream));
    ZleArgs za{};
    za.active_rows = h->active_rows.as<i32>(); za.n_active_rows = h->n_active_rows; za.row_lo = h->row_lo.as<i64>(); za.row_hi = h->row_hi.as<i64>();
    za.acc_off = h->acc_off.as<i64>(); za.raw = h->raw.as<i32>(); za.grp_left = h->grp_left.as<i64>(); za.grp_ixrand = h->grp_ixrand.as<i64>();
    za.itv_off = h->itv_off.as<i64>(); za.itv_left = h->itv_left.as<i64>(); za.itv_right = h->itv_right.as<i64>();
    za.itv_n = h->itv_n.as<i32>(); za.row_nrec = h->row_nrec.as<i32>(); za.spr = WFS_SPR;
    if (tiles_done) { za.tile_done = ga.tile_done; za.n_done = ga.n_done; za.row_cnt = ga.row_cnt; za.row_tile = ga.row_tile; za.ins_bcap = ga.ins_bcap; za.ins_boff = ga.ins_boff; za.tbuf = h->tbuf.as<i32>(); }
    za.n_front = h->n_front_rows; za.rows_cap = CG * d.row_slots;
    if (h->n_res_rows > 0) {
        TRY(ensure(h, h->fin, (size_t)h->s_fin * 2 + 16));
        TRY(ensure(h, h->res_rows, (size_t)h->n_res_rows * sizeof(ResRow)));
        za.res_toff = h->res_toff.as<i64>(); za.fin_off = h->fin_off.as<i64>(); za.fin = h->fin.as<int16_t>();
        za.n_short = h->n_short_rows; za.res_long = h->res_long.as<i32>(); za.res_rows = h->res_rows.as<ResRow>();
    }
    h->row_dbg_total = 0;
    if ((h->keep_currents & 1) && h->n_active_rows > 0) {
        std::vector<i32> ar((size_t)h->n_active_rows);
        std::vector<i64> lo((size_t)CG * d.n_tpc), hi((size_t)CG * d.n_tpc);
        HIPCHK(hipMemcpy(ar.data(), h->active_rows.p, ar.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lo.data(), h->row_lo.p, lo.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hi.data(), h->row_hi.p, hi.size() * 8, hipMemcpyDeviceToHost));
        std::vector<i64> off(ar.size() + 1, 0);
        for (size_t k = 0; k < ar.size(); k++) {
            i64 g = ar[k] / d.row_slots; i32 slot = (i32)(ar[k] - g * d.row_slots); i32 c = slot < d.n_tpc ? slot : slot - d.n_tpc;
            off[k + 1] = off[k] + hi[g * d.n_tpc + c] - lo[g * d.n_tpc + c] + 1 + 2 * d.tw;
        }
        h->row_dbg_total = off.back();
        TRY(upload(h, h->row_dbg_off, off.data(), off.size() * 8));
        TRY(ensure(h, h->row_dbg, (size_t)h->row_dbg_total * 4));
        HIPCHK(hipStreamSynchronize(h->stream));
        za.row_dbg = h->row_dbg.as<i32>(); za.row_dbg_off = h->row_dbg_off.as<i64>();
    }
    TRY(ensure(h, h->row_desc, (size_t)h->n_active_rows * sizeof(RowDesc))); za.desc = h->row_desc.as<RowDesc>();
    if (h->sort_records) {      // scal[22] / scal[29]: first sample / end of the last row of the batch (k_row_desc): origin and span of the sort keys
        za.key_base = h->scal.as<i64>() + 22;
        hipLaunchKernelGGL(k_fill_i64, dim3(1), dim3(64), 0, h->stream, za.key_base, (i64)1, I64_MAX);
        hipLaunchKernelGGL(k_fill_i64, dim3(1), dim3(64), 0, h->stream, h->scal.as<i64>() + 29, (i64)1, I64_MIN);
    }
    if (h->n_active_rows > 0) { Timer t(h, "k_row_desc"); hipLaunchKernelGGL(k_row_desc, dim3(nblocks(h->n_active_rows, 256)), dim3(256), 0, h->stream, d, za); }
    const int noise_kind = !d.enable_noise ? 0 : (d.noise_f ? 2 : 1);      // (a template parameter of the row kernels: no branch between their loads)
    if (h->n_res_rows > 0) {        // resident rows: pulses, finished samples and intervals by one wave per row
        PulseArgs pr = pa;
        pr.desc = h->res_desc.as<TileDesc>(); pr.currents = nullptr; pr.cur_off = nullptr;
        // two launches: the rows of at most RES_SHORT_LEN samples at full occupancy, the longer ones with LDS for the longest of them
        Timer t(h, "k_row_pulse");
#define WFS_ROW_PULSE2(NK, F) do { if (part == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_row_pulse<NK, F, false>), grid, dim3(256), lds, h->stream, d, pr, za, first, n_rows, region); \
                                   else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_row_pulse<NK, F, true>), grid, dim3(256), lds, h->stream, d, pr, za, first, n_rows, region); } while (0)
#define WFS_ROW_PULSE(NK) do { if (h->cfg.fma) WFS_ROW_PULSE2(NK, true); else WFS_ROW_PULSE2(NK, false); } while (0)
        for (int part = 0; part < 2; part++) {
            const i64 first = part == 0 ? 0 : h->n_short_rows, n_rows = part == 0 ? h->n_short_rows : h->n_res_rows - h->n_short_rows;
            if (n_rows <= 0) continue;
            const i32 region = part == 0 ? RES_SHORT_LEN : (i32)((std::min<i64>(h->max_res_len, h->res_max_len) + 255) / 256 * 256);
            const size_t lds = ROW_LDS_FIXED + (size_t)4 * region * 4;
            const dim3 grid(nblocks(n_rows, 4));
            if (noise_kind == 0) WFS_ROW_PULSE(0); else if (noise_kind == 1) WFS_ROW_PULSE(1); else WFS_ROW_PULSE(2);
        }
    }
    {   // truth accumulators of every pulse set from the per-tile partial sums
        TruthArgs ta{S, h->tile_count.as<i32>(), h->tile_tmin.as<i32>(), h->tile_tmax.as<i32>(), h->set_t0.as<i64>(), h->tile_truth.as<double>(),
                     h->truth.as<double>(), h->tminmax.as<i64>()};
        Timer t(h, "k_truth_reduce");
        hipLaunchKernelGGL(k_truth_reduce, dim3(nblocks(S, 4)), dim3(256), 0, h->stream, d, ta);
    }

    if (h->n_front_rows > 0) {
        Timer t(h, "k_zle"); const dim3 grid(nblocks(h->n_front_rows, 4));
        if (noise_kind == 0) hipLaunchKernelGGL(k_zle<0>, grid, dim3(256), 0, h->stream, d, za);
        else if (noise_kind == 1) hipLaunchKernelGGL(k_zle<1>, grid, dim3(256), 0, h->stream, d, za);
        else hipLaunchKernelGGL(k_zle<2>, grid, dim3(256), 0, h->stream, d, za);
    }
    TRY(scan(h, h->row_nrec.as<i32>(), RS, h->rec_off, 10));
    TRY(read_scal(h));
    h->n_records = h->h_scal[10];
    // the other arena: the previous batch's records may still be on their way to the host
    h->rec_cur ^= 1;
    if (h->rec_pending[h->rec_cur]) { HIPCHK(hipEventSynchronize(h->rec_copied[h->rec_cur])); h->rec_pending[h->rec_cur] = false; }
    TRY(ensure(h, h->records_buf(), (size_t)h->n_records * 244));
    za.rec_off = h->rec_off.as<i64>(); za.records = h->records_buf().as<uint8_t>(); za.rec_capacity = h->n_records;
    if (h->sort_records && h->n_records > 1 && h->n_active_rows > 0) {
        // records by (time, channel): keys per record, one radix sort of the batch, k_pack writes to the sorted slots
        const i64 NR = h->n_records;
        if (NR > 0x7fffffffLL) return h->fail(WFS_E_CAPACITY, "more than 2^31 records in one batch (the device sort takes an int count)");
        TRY(ensure(h, h->rec_key, (size_t)NR * 8)); TRY(ensure(h, h->rec_key2, (size_t)NR * 8));
        TRY(ensure(h, h->rec_val, (size_t)NR * 4)); TRY(ensure(h, h->rec_val2, (size_t)NR * 4)); TRY(ensure(h, h->rec_dest, (size_t)NR * 4));
        za.rec_key = h->rec_key.as<u64>(); za.rec_val = h->rec_val.as<u32>();
        { Timer t(h, "k_rec_keys"); hipLaunchKernelGGL(k_rec_keys, dim3(nblocks(h->n_active_rows, 4)), dim3(256), 0, h->stream, d, za); }
        // rocPRIM's radix sort, over the key bits that are in use only: (samples the batch spans) << 12 | channel
        unsigned end_bit = 13;
        { const u64 span = (u64)std::max<i64>(h->h_scal[29] - h->h_scal[22], 1); while (end_bit < 64 && ((span << 12) >> end_bit) != 0) end_bit++; }
        size_t bytes = 0;
        HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, h->rec_key.as<u64>(), h->rec_key2.as<u64>(), h->rec_val.as<u32>(), h->rec_val2.as<u32>(), (size_t)NR, 0u, end_bit, h->stream));
        TRY(ensure(h, h->sort_tmp, bytes));
        { Timer t(h, "record_sort"); HIPCHK(rocprim::radix_sort_pairs(h->sort_tmp.p, bytes, h->rec_key.as<u64>(), h->rec_key2.as<u64>(), h->rec_val.as<u32>(), h->rec_val2.as<u32>(), (size_t)NR, 0u, end_bit, h->stream)); }
        { Timer t(h, "k_invert_perm"); hipLaunchKernelGGL(k_invert_perm, dim3(nblocks(NR, 256)), dim3(256), 0, h->stream, h->rec_val2.as<u32>(), h->rec_dest.as<u32>(), NR); }
        za.rec_dest = h->rec_dest.as<u32>();
    }
    if (h->n_active_rows > 0 && h->n_records > 0) {
        TRY(ensure(h, h->pack_desc, (size_t)h->n_active_rows * sizeof(PackDesc))); za.pdesc = h->pack_desc.as<PackDesc>();
        { Timer t(h, "k_pack_desc"); hipLaunchKernelGGL(k_pack_desc, dim3(nblocks(h->n_active_rows, 256)), dim3(256), 0, h->stream, za); }
        if (h->n_res_rows > 0) {
            PackResArgs pr{h->pack_desc.as<PackDesc>() + h->n_front_rows, h->fin.as<int16_t>(), za.itv_left, za.itv_right, za.records, za.rec_dest, za.rec_capacity, h->n_res_rows, za.spr, (i32)d.dt};
            Timer t(h, "k_pack_res"); hipLaunchKernelGGL(k_pack_res, dim3(nblocks(h->n_res_rows, 4)), dim3(256), 0, h->stream, pr);
        }
        if (h->n_front_rows > 0) {
        Timer t(h, "k_pack"); const dim3 grid(nblocks(h->n_front_rows, 4));
        if (noise_kind == 0) hipLaunchKernelGGL(k_pack<0>, grid, dim3(256), 0, h->stream, d, za);
        else if (noise_kind == 1) hipLaunchKernelGGL(k_pack<1>, grid, dim3(256), 0, h->stream, d, za);
        else hipLaunchKernelGGL(k_pack<2>, grid, dim3(256), 0, h->stream, d, za);
        }
    }
    {   // totals of wfs_get_counts: afterpulse sets carry no truth (rawdata.py:322-323)
        const i64 n_prim = (!h->injected && h->ap_active) ? h->n_psets : h->n_sets;
        const i64 work = std::max<i64>(RS, n_prim);
        hipLaunchKernelGGL(k_counts, dim3((unsigned)std::min<i64>(nblocks(work, 256), 128)), dim3(256), 0, h->stream, h->itv_n.as<i32>(), RS, h->truth.as<double>(), n_prim, h->scal.as<i64>());
    }
    TRY(read_scal(h));
    HIPCHK(hipGetLastError());
    CHECK_LAUNCHES();
    h->ran = true;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_get_counts(wfs_handle *h, wfs_counts *out)
try {
    if (!h || !out) return WFS_E_INVALID;
    if (!h->ran) return h->fail(WFS_E_STATE, "wfs_run has not completed");
    wfs_counts c{};
    c.n_instructions = h->n_ins; c.n_pulse_sets = h->n_sets; c.n_emitters = h->n_emitters; c.n_photons = h->n_photons + ((!h->injected && h->ap_active) ? h->n_ap_photons : 0);
    const bool tiles_done = !h->injected && !h->optical && h->fuse_full && h->n_fused_tiles > 0;
    c.n_tiles = h->n_active_tiles + (tiles_done ? h->n_fused_tiles : 0) + h->n_res_tiles; c.n_groups = h->n_groups; c.n_rows = h->n_active_rows;
    c.n_raw_samples = h->s_raw + (tiles_done ? h->s_raw_direct : 0) + h->s_res;
    c.n_records = h->n_records;
    c.n_intervals = h->h_scal[20]; c.n_pe = h->h_scal[21];          // reduced on the device at the end of wfs_run (k_counts)
    *out = c; h->counts = c;
    return WFS_OK;
} WFS_CATCH(h)

const void *wfs_records_dev_ptr(wfs_handle *h) { return h ? h->records_buf().p : nullptr; }

int wfs_copy_records(wfs_handle *h, void *dst, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (cap < h->n_records) return h->fail(WFS_E_CAPACITY, "record buffer too small");
    if (h->n_records) HIPCHK(hipMemcpy(dst, h->records_buf().p, (size_t)h->n_records * 244, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_records_range(wfs_handle *h, void *dst, int64_t first, int64_t count)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (first < 0 || count < 0 || first + count > h->n_records) return h->fail(WFS_E_INVALID, "record range outside the batch");
    if (count) HIPCHK(hipMemcpy(dst, (const uint8_t *)h->records_buf().p + (size_t)first * 244, (size_t)count * 244, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)

// Pinned host memory + the copy stream: the records of a batch travel to the host while the next batch's kernels run
// (strax_interface.py:360-364: the caller owns the record buffer; pinning it once removes the staging copy of pageable
// transfers -- 15 GB/s -- and lets the transfer overlap).  wfs_run leaves the records of the last TWO batches intact.
int wfs_host_register(void *ptr, int64_t bytes)
try {
    if (!ptr || bytes <= 0) return WFS_E_INVALID;
    return hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault) == hipSuccess ? WFS_OK : WFS_E_HIP;
} WFS_CATCH(nullptr)
int wfs_host_unregister(void *ptr) try { return (ptr && hipHostUnregister(ptr) == hipSuccess) ? WFS_OK : WFS_E_HIP; } WFS_CATCH(nullptr)

int wfs_copy_records_range_async(wfs_handle *h, void *dst, int64_t first, int64_t count)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (first < 0 || count < 0 || first + count > h->n_records) return h->fail(WFS_E_INVALID, "record range outside the batch");
    HIPCHK(hipSetDevice(h->device));
    if (count) HIPCHK(hipMemcpyAsync(dst, (const uint8_t *)h->records_buf().p + (size_t)first * 244, (size_t)count * 244, hipMemcpyDeviceToHost, h->copy_stream));
    HIPCHK(hipEventRecord(h->rec_copied[h->rec_cur], h->copy_stream));
    h->rec_pending[h->rec_cur] = true;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_wait_records(wfs_handle *h)
try {
    if (!h) return WFS_E_INVALID;
    // May be called from the consumer's thread while a worker thread is inside wfs_run for the next batch (RawData.iter_batches):
    // it only waits for the copy stream and leaves the handle's state alone -- rec_pending stays with the thread that runs and copies
    // (wfs_run's own wait on an event that has completed returns at once).
    if (hipStreamSynchronize(h->copy_stream) != hipSuccess) return WFS_E_HIP;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_records_dev(wfs_handle *h, void *dst, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (cap < h->n_records) return h->fail(WFS_E_CAPACITY, "record buffer too small");
    if (h->n_records) { HIPCHK(hipMemcpyAsync(dst, h->records_buf().p, (size_t)h->n_records * 244, hipMemcpyDeviceToDevice, h->stream)); HIPCHK(hipStreamSynchronize(h->stream)); }
    return WFS_OK;
} WFS_CATCH(h)

// records of a batch in the order strax.sort_by_time gives them ((time, channel); strax_interface.py:453) instead of the
// order the reference yields pulses (window, channel, interval)
int wfs_set_record_order(wfs_handle *h, int32_t by_time)
try {
    if (!h) return WFS_E_INVALID;
    h->sort_records = by_time != 0;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_groups(wfs_handle *h, int64_t *left, int64_t *right, int64_t *first_record, int64_t *ix_rand)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    const i64 G = h->n_groups;
    if (left) HIPCHK(hipMemcpy(left, h->grp_left.p, (size_t)G * 8, hipMemcpyDeviceToHost));
    if (right) HIPCHK(hipMemcpy(right, h->grp_right.p, (size_t)G * 8, hipMemcpyDeviceToHost));
    if (ix_rand) HIPCHK(hipMemcpy(ix_rand, h->grp_ixrand.p, (size_t)G * 8, hipMemcpyDeviceToHost));
    if (first_record && G > 0)      // rec_off[g * row_slots]: a strided copy of one value per window (the whole array is 6 KB per cluster)
        HIPCHK(hipMemcpy2D(first_record, 8, h->rec_off.p, (size_t)h->dev.row_slots * 8, 8, (size_t)G, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_intervals(wfs_handle *h, int32_t *group, int32_t *channel, int64_t *left, int64_t *right, int64_t *data_off, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    const WfsDev &d = h->dev;
    const i64 RS = (h->n_clusters + 1) * d.row_slots;
    std::vector<i32> n((size_t)RS); std::vector<i64> off((size_t)RS + 1), L((size_t)h->n_itv_slots), R((size_t)h->n_itv_slots);
    HIPCHK(hipMemcpy(n.data(), h->itv_n.p, n.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(off.data(), h->itv_off.p, off.size() * 8, hipMemcpyDeviceToHost));
    if (h->n_itv_slots) {
        HIPCHK(hipMemcpy(L.data(), h->itv_left.p, L.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(R.data(), h->itv_right.p, R.size() * 8, hipMemcpyDeviceToHost));
    }
    i64 k = 0, doff = 0;
    for (i64 idx = 0; idx < RS; idx++) {
        i64 g = idx / d.row_slots; i32 slot = (i32)(idx - g * d.row_slots);
        i32 chn = slot < d.n_tpc ? slot : d.he_first + slot - d.n_tpc;
        for (i32 q = 0; q < n[idx]; q++) {
            if (k >= cap) return h->fail(WFS_E_CAPACITY, "interval buffer too small");
            i64 l = L[off[idx] + q], r = R[off[idx] + q];
            group[k] = (i32)g; channel[k] = chn; left[k] = l; right[k] = r; data_off[k] = doff;
            doff += (r - l + 1 > 0) ? r - l + 1 : 0; k++;
        }
    }
    if (k < cap) data_off[k] = doff;
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_interval_data(wfs_handle *h, int16_t *data, int64_t cap)
try {
    // unpack the records: fragments of one interval are consecutive
    if (!h || !h->ran) return WFS_E_STATE;
    std::vector<uint8_t> rec((size_t)h->n_records * 244);
    if (h->n_records) HIPCHK(hipMemcpy(rec.data(), h->records_buf().p, rec.size(), hipMemcpyDeviceToHost));
    i64 k = 0;
    for (i64 r = 0; r < h->n_records; r++) {
        i32 len; memcpy(&len, &rec[(size_t)r * 244 + 8], 4);
        if (k + len > cap) return h->fail(WFS_E_CAPACITY, "interval data buffer too small");
        memcpy(data + k, &rec[(size_t)r * 244 + 24], (size_t)len * 2);
        k += len;
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_pulses(wfs_handle *h, int32_t *set, int32_t *channel, int64_t *left, int64_t *right, int64_t *nph, int64_t *cur_off, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    const WfsDev &d = h->dev;
    const i64 A = h->n_active_tiles;
    if (cap < A) return h->fail(WFS_E_CAPACITY, "pulse buffer too small");
    std::vector<i32> at((size_t)A), cnt((size_t)h->n_tiles), tmn((size_t)h->n_tiles), tmx((size_t)h->n_tiles);
    std::vector<i64> t0((size_t)h->n_sets);
    if (A) HIPCHK(hipMemcpy(at.data(), h->active_tiles.p, at.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cnt.data(), h->tile_count.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tmn.data(), h->tile_tmin.p, tmn.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(tmx.data(), h->tile_tmax.p, tmx.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(t0.data(), h->set_t0.p, t0.size() * 8, hipMemcpyDeviceToHost));
    i64 off = 0;
    for (i64 k = 0; k < A; k++) {
        i64 tile = at[k], s = tile / d.n_tpc;
        i64 b0 = floordiv(t0[s] + tmn[tile], (i64)d.dt), b1 = floordiv(t0[s] + tmx[tile], (i64)d.dt);
        set[k] = (i32)s; channel[k] = (i32)(tile - s * d.n_tpc);
        left[k] = b0 - d.store_before - d.samples_before; right[k] = b1 + d.store_after + d.samples_after;
        nph[k] = cnt[tile]; cur_off[k] = off; off += right[k] - left[k] + 1;
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_currents(wfs_handle *h, double *cur, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (!(h->keep_currents & 1)) return h->fail(WFS_E_STATE, "wfs_set_debug(h, 1) before wfs_run");
    if (cap < h->cur_total) return h->fail(WFS_E_CAPACITY, "current buffer too small");
    if (h->cur_total) HIPCHK(hipMemcpy(cur, h->currents.p, (size_t)h->cur_total * 8, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_rows(wfs_handle *h, int32_t *group, int32_t *channel, int64_t *left, int64_t *right, int64_t *data_off, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (!(h->keep_currents & 1)) return h->fail(WFS_E_STATE, "wfs_set_debug(h, 1) before wfs_run");
    const WfsDev &d = h->dev;
    const i64 A = h->n_active_rows;
    if (cap < A) return h->fail(WFS_E_CAPACITY, "row buffer too small");
    std::vector<i32> ar((size_t)A); std::vector<i64> lo((size_t)(h->n_clusters + 1) * d.n_tpc), hi(lo.size()), gl((size_t)h->n_groups);
    if (A) HIPCHK(hipMemcpy(ar.data(), h->active_rows.p, ar.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(lo.data(), h->row_lo.p, lo.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hi.data(), h->row_hi.p, hi.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gl.data(), h->grp_left.p, gl.size() * 8, hipMemcpyDeviceToHost));
    i64 off = 0;
    for (i64 k = 0; k < A; k++) {
        i64 g = ar[k] / d.row_slots; i32 slot = (i32)(ar[k] - g * d.row_slots); i32 c = slot < d.n_tpc ? slot : slot - d.n_tpc;
        group[k] = (i32)g; channel[k] = slot < d.n_tpc ? slot : d.he_first + c;
        // relative to the window like the reference's channel mask (rawdata.py:258-259)
        left[k] = lo[g * d.n_tpc + c] - gl[g] - d.tw; right[k] = hi[g * d.n_tpc + c] - gl[g] + d.tw;
        data_off[k] = off; off += right[k] - left[k] + 1;
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_row_data(wfs_handle *h, int32_t *data, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (cap < h->row_dbg_total) return h->fail(WFS_E_CAPACITY, "row data buffer too small");
    if (h->row_dbg_total) HIPCHK(hipMemcpy(data, h->row_dbg.p, (size_t)h->row_dbg_total * 4, hipMemcpyDeviceToHost));
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_photons(wfs_handle *h, int64_t *set_off, int64_t *t, int16_t *ch, double *gain, uint8_t *dpe, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    if (!h->injected && !h->optical && h->fuse_full && h->n_fused_tiles > 0 && !(h->keep_currents & 16))
        return h->fail(WFS_E_STATE, "the photons of tile-generated instructions were not kept: wfs_set_debug bit 4 before wfs_run");
    const WfsDev &d = h->dev;
    const i64 P = h->n_photons + ((!h->injected && h->ap_active) ? h->n_ap_photons : 0), T = h->n_tiles;
    if (cap < P) return h->fail(WFS_E_CAPACITY, "photon buffer too small");
    std::vector<i64> off((size_t)T + 1), t0((size_t)h->n_sets);
    std::vector<PhotonRec> recs((size_t)P);
    HIPCHK(hipMemcpy(off.data(), h->tile_off.p, off.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(t0.data(), h->set_t0.p, t0.size() * 8, hipMemcpyDeviceToHost));
    if (P) HIPCHK(hipMemcpy(recs.data(), h->ph.p, recs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> gains((size_t)d.n_tpc), spe((size_t)2001 * d.n_spe), pg;
    HIPCHK(hipMemcpy(gains.data(), h->t_gains.p, gains.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(spe.data(), h->t_spe.p, spe.size() * 8, hipMemcpyDeviceToHost));
    const i64 gain_first = h->injected ? 0 : h->n_photons;         // ph_gain covers the explicit-gain photons only
    if (P - gain_first > 0 && (h->injected || h->ap_active)) { pg.resize((size_t)(P - gain_first)); HIPCHK(hipMemcpy(pg.data(), h->ph_gain.p, pg.size() * 8, hipMemcpyDeviceToHost)); }
    if (off[(size_t)T] != P) { char msg[160]; snprintf(msg, sizeof msg, "wfs_copy_photons: tile offsets end at %lld, %lld photons (T %lld, sets %lld)", (long long)off[(size_t)T], (long long)P, (long long)T, (long long)h->n_sets); return h->fail(WFS_E_STATE, msg); }
    for (i64 s = 0; s <= h->n_sets; s++) set_off[s] = off[(size_t)s * d.n_tpc];
    for (i64 tile = 0; tile < T; tile++) {
        i64 s = tile / d.n_tpc; int c = (int)(tile - s * d.n_tpc);
        const double *row = &spe[(size_t)(d.n_spe > 1 ? c : 0) * 2001];
        for (i64 p = off[tile]; p < off[tile + 1]; p++) {
            t[p] = t0[s] + recs[p].t; ch[p] = (int16_t)c;
            u32 g1 = recs[p].code & 0xffffu, g2 = recs[p].code >> 16;
            if (h->injected || p >= h->n_photons) { gain[p] = pg[(size_t)(p - gain_first)]; dpe[p] = g2 != 0; }
            else { double g = gains[c] * row[g1]; if (g2) g += gains[c] * row[g2]; gain[p] = g; dpe[p] = g2 != 0; }
        }
    }
    return WFS_OK;
} WFS_CATCH(h)

// Generation order of a batch's photons as the electron-afterpulse pre-pass sees it (rawdata.py:133-145 hands the parent S2's photons to
// afterpulse.py): instruction by instruction; the photons of an instruction of the per-electron generator in its generation order, those
// of a tile-generated instruction tile by tile (channel ascending), photon q of a tile at position q.  Host-side tables of that order,
// built once per generation.
static int gen_order_tables(wfs_handle *h)
{
    if (h->gen_order_ready) return WFS_OK;
    const i64 N = h->n_ins; const int nch = h->dev.n_tpc;
    std::vector<i64> emo((size_t)N + 1), epo((size_t)h->n_emitters + 1);
    HIPCHK(hipMemcpy(emo.data(), h->em_off.p, emo.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(epo.data(), h->em_ph_off.p, epo.size() * 8, hipMemcpyDeviceToHost));
    h->go_fused.assign((size_t)N, 0); h->go_tile_count.clear();
    const bool any_fused = h->fuse_on && h->n_fused_tiles + h->n_gen_tiles > 0;
    if (any_fused) {
        HIPCHK(hipMemcpy(h->go_fused.data(), h->ins_fused.p, (size_t)N * 4, hipMemcpyDeviceToHost));
        h->go_tile_count.resize((size_t)N * nch);
        HIPCHK(hipMemcpy(h->go_tile_count.data(), h->tile_count.p, h->go_tile_count.size() * 4, hipMemcpyDeviceToHost));      // (one pulse set per instruction: tile = instruction * n_tpc + channel)
    }
    h->go_off.assign((size_t)N + 1, 0); h->go_block0.assign((size_t)N, 0);
    for (i64 i = 0; i < N; i++) {
        i64 n = epo[(size_t)emo[i + 1]] - epo[(size_t)emo[i]];           // (0 for a tile-generated instruction: its electrons carry no photon numbers)
        h->go_block0[i] = epo[(size_t)emo[i]];
        if (any_fused && h->go_fused[i]) for (int c = 0; c < nch; c++) n += h->go_tile_count[(size_t)i * nch + c];
        h->go_off[i + 1] = h->go_off[i] + n;
    }
    h->gen_order_ready = true;
    return WFS_OK;
}

int wfs_copy_instruction_photon_offsets(wfs_handle *h, int64_t *off, int64_t cap)
try {
    if (!h || !h->gen_done || h->injected || h->optical) return WFS_E_STATE;
    const i64 N = h->n_ins;
    if (cap < N + 1) return h->fail(WFS_E_CAPACITY, "offset buffer too small");
    TRY(gen_order_tables(h));
    for (i64 i = 0; i <= N; i++) off[i] = h->go_off[(size_t)i];
    return WFS_OK;
} WFS_CATCH(h)

int wfs_gather_photon_times(wfs_handle *h, int64_t n, const int64_t *index, int64_t *t_out)
try {
    if (!h || !h->gen_done || h->injected || h->optical) return WFS_E_STATE;
    if (n <= 0) return WFS_OK;
    if (!index || !t_out) return h->fail(WFS_E_INVALID, "wfs_gather_photon_times: null argument");
    TRY(gen_order_tables(h));
    const i64 N = h->n_ins; const int nch = h->dev.n_tpc;
    for (i64 i = 0; i < n; i++) if (index[i] < 0 || index[i] >= h->go_off[(size_t)N]) return h->fail(WFS_E_INVALID, "photon index out of range");
    HIPCHK(hipSetDevice(h->device));
    // requests of the per-electron generator (index in ITS photon order) and of tile-generated instructions (instruction, channel, q)
    std::vector<i64> blk_idx, blk_pos, tile_pos; std::vector<TileTimeReq> treq;
    for (i64 k = 0; k < n; k++) {
        const i64 i = (i64)(std::upper_bound(h->go_off.begin(), h->go_off.end(), (i64)index[k]) - h->go_off.begin()) - 1;
        i64 local = index[k] - h->go_off[(size_t)i];
        if (!h->go_fused[(size_t)i]) { blk_idx.push_back(h->go_block0[(size_t)i] + local); blk_pos.push_back(k); continue; }
        int c = 0;
        const i32 *cnt = &h->go_tile_count[(size_t)i * nch];
        while (c < nch - 1 && local >= cnt[c]) { local -= cnt[c]; c++; }
        treq.push_back(TileTimeReq{(i32)i, (i32)c, (i32)local, 0}); tile_pos.push_back(k);
    }
    std::vector<i64> got;
    if (!blk_idx.empty()) {
        const i64 m = (i64)blk_idx.size();
        TRY(upload(h, h->gather_idx, blk_idx.data(), (size_t)m * 8)); TRY(ensure(h, h->gather_out, (size_t)m * 8));
        hipLaunchKernelGGL(k_photon_times, dim3(nblocks(m, 256)), dim3(256), 0, h->stream, h->dev, h->gen_args, m, h->gather_idx.as<i64>(), h->gather_out.as<i64>());
        got.resize((size_t)m);
        HIPCHK(hipMemcpyAsync(got.data(), h->gather_out.p, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (i64 k = 0; k < m; k++) t_out[blk_pos[(size_t)k]] = got[(size_t)k];
    }
    if (!treq.empty()) {
        const i64 m = (i64)treq.size();
        TRY(upload(h, h->gather_idx, treq.data(), (size_t)m * sizeof(TileTimeReq))); TRY(ensure(h, h->gather_out, (size_t)m * 8));
        hipLaunchKernelGGL(k_tile_photon_times, dim3(nblocks(m, 256)), dim3(256), 0, h->stream, h->dev, h->fuse_args, m, h->gather_idx.as<TileTimeReq>(), h->gather_out.as<i64>());
        got.resize((size_t)m);
        HIPCHK(hipMemcpyAsync(got.data(), h->gather_out.p, (size_t)m * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (i64 k = 0; k < m; k++) t_out[tile_pos[(size_t)k]] = got[(size_t)k];
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_truth(wfs_handle *h, double *acc12, double *tstat5, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    const i64 S = h->n_sets;
    if (cap < S) return h->fail(WFS_E_CAPACITY, "truth buffer too small");
    std::vector<double> tr((size_t)S * 16); std::vector<i64> mm((size_t)S * 2), t0((size_t)S);
    HIPCHK(hipMemcpy(tr.data(), h->truth.p, tr.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mm.data(), h->tminmax.p, mm.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(t0.data(), h->set_t0.p, t0.size() * 8, hipMemcpyDeviceToHost));
    for (i64 s = 0; s < S; s++) {
        for (int f = 0; f < 12; f++) acc12[s * 12 + f] = tr[s * 16 + f];
        double n = tr[s * 16 + 12], st = tr[s * 16 + 13], st2 = tr[s * 16 + 14];
        double mean = n > 0 ? st / n : 0, var = n > 0 ? st2 / n - mean * mean : 0;
        tstat5[s * 5 + 0] = n; tstat5[s * 5 + 1] = n > 0 ? (double)t0[s] + mean : NAN;
        tstat5[s * 5 + 2] = n > 0 ? (double)mm[2 * s] : NAN; tstat5[s * 5 + 3] = n > 0 ? (double)mm[2 * s + 1] : NAN;
        tstat5[s * 5 + 4] = n > 0 ? sqrt(var > 0 ? var : 0) : NAN;
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_truth_per_pmt(wfs_handle *h, double *acc6, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    const i64 S = h->n_sets, nch = h->cfg.n_tpc, T = S * nch;
    if (cap < S) return h->fail(WFS_E_CAPACITY, "per-PMT truth buffer too small");
    std::vector<double> tt((size_t)T * 8); std::vector<i32> cnt((size_t)T); std::vector<double> gains((size_t)nch);
    HIPCHK(hipMemcpy(tt.data(), h->tile_truth.p, tt.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cnt.data(), h->tile_count.p, cnt.size() * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(gains.data(), h->t_gains.p, gains.size() * 8, hipMemcpyDeviceToHost));
    for (i64 t = 0; t < T; t++) {
        double *o = acc6 + t * 6;
        if (cnt[t] <= 0) { for (int q = 0; q < 6; q++) o[q] = 0.0; continue; }     // tiles without photons were never written
        const double *p = tt.data() + t * 8, G = gains[t % nch];
        o[0] = p[0]; o[1] = p[0] + p[1]; o[2] = p[2]; o[3] = p[2] + p[3]; o[4] = p[4] / G; o[5] = p[5] / G;       // pulse.py:259-271
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_copy_electron_stats(wfs_handle *h, double *estat5, int64_t cap)
try {
    if (!h || !h->ran) return WFS_E_STATE;
    const i64 N = h->n_ins;
    if (h->optical) { for (i64 i = 0; i < N && i < cap; i++) { estat5[i * 5] = 0; for (int q = 1; q < 5; q++) estat5[i * 5 + q] = NAN; } return WFS_OK; }
    const i64 PS = h->n_psets;
    if (cap < PS) return h->fail(WFS_E_CAPACITY, "electron stats buffer too small");
    if (N == 0) return WFS_OK;
    std::vector<double> st((size_t)N * 4); std::vector<i64> mm((size_t)N * 2), t0((size_t)N);
    HIPCHK(hipMemcpy(st.data(), h->el_stat.p, st.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mm.data(), h->el_minmax.p, mm.size() * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(t0.data(), h->ins_time.p, t0.size() * 8, hipMemcpyDeviceToHost));
    for (i64 q = 0; q < PS; q++) {
        // moments of the electrons of all instructions of the run set (rawdata.py:334-341), about the first instruction's time
        const i64 ref = h->h_rs_off[q + 1] > h->h_rs_off[q] ? t0[h->h_rs_list[h->h_rs_off[q]]] : 0;      // (an unused set number: no instructions, zero electrons)
        double n = 0, s1 = 0, s2 = 0; i64 lo = I64_MAX, hi = I64_MIN;
        for (i64 k = h->h_rs_off[q]; k < h->h_rs_off[q + 1]; k++) {
            const i64 i = h->h_rs_list[k];
            const double ni = st[i * 4], a1 = st[i * 4 + 1], a2 = st[i * 4 + 2], dlt = (double)(t0[i] - ref);
            if (!(ni > 0)) continue;
            n += ni; s1 += a1 + ni * dlt; s2 += a2 + 2 * dlt * a1 + ni * dlt * dlt;
            lo = std::min(lo, mm[2 * i]); hi = std::max(hi, mm[2 * i + 1]);
        }
        double mean = n > 0 ? s1 / n : 0, var = n > 0 ? s2 / n - mean * mean : 0;
        estat5[q * 5 + 0] = n; estat5[q * 5 + 1] = n > 0 ? (double)ref + mean : NAN;
        estat5[q * 5 + 2] = n > 0 ? (double)lo : NAN; estat5[q * 5 + 3] = n > 0 ? (double)hi : NAN;
        estat5[q * 5 + 4] = n > 0 ? sqrt(var > 0 ? var : 0) : NAN;
    }
    return WFS_OK;
} WFS_CATCH(h)

int wfs_kernel_times(wfs_handle *h, char *names, int64_t names_cap, float *ms, int32_t *n_launches, int32_t *n_kernels)
try {
    if (!h) return WFS_E_INVALID;
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<std::string> uniq; std::vector<float> tot; std::vector<int> cnt;
    for (auto &t : h->times) {
        float m = 0; hipEventElapsedTime(&m, t.a, t.b);
        size_t k = 0; for (; k < uniq.size(); k++) if (uniq[k] == t.name) break;
        if (k == uniq.size()) { uniq.push_back(t.name); tot.push_back(0); cnt.push_back(0); }
        tot[k] += m; cnt[k]++;
    }
    i64 pos = 0;
    for (size_t k = 0; k < uniq.size(); k++) {
        if (pos + (i64)uniq[k].size() + 1 > names_cap) return h->fail(WFS_E_CAPACITY, "names buffer too small");
        memcpy(names + pos, uniq[k].c_str(), uniq[k].size() + 1); pos += (i64)uniq[k].size() + 1;
        ms[k] = tot[k]; n_launches[k] = cnt[k];
    }
    *n_kernels = (int32_t)uniq.size();
    return WFS_OK;
} WFS_CATCH(h)

}  // extern "C"
