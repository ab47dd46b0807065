/*
 * wfsim_amd.h -- C ABI of the MI355X-native WFSim hot path (libwfsim_amd.so, HIP, gfx950 only).
 *
 * The reference (XENONnT/WFSim v1.2.2) is pure Python and has no FFI; its hot path lives behind
 * wfsim.RawData / wfsim.ChunkRawRecords.  This header is what a binding for that path binds: every entry point
 * names the reference routine(s) it replaces (paths relative to the reference's wfsim/ directory).  The Python
 * host side of this repository (wfsim_amd/engine.py, ctypes) is such a binding; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain C types only; no exceptions cross the ABI: every entry point is a function-try-block that turns a C++
 *     exception of the host-side containers into a code (std::bad_alloc -> WFS_E_NOMEM, anything else -> WFS_E_INVALID)
 *     and a message.  Every function returns 0 on success or a negative WFS_E_* code; wfs_last_error(h) gives the message.
 *   - a handle owns one GPU (hipSetDevice at creation) and one HIP stream; it is not thread safe.
 *   - "host" pointers are ordinary CPU memory (numpy buffers), copied by the call; "dev" pointers are HBM
 *     addresses on the handle's GPU (e.g. torch tensors' data_ptr()).
 *   - instructions of one batch must be sorted by the scheduler key time - z/v*[S2] (rawdata.py:61) and carry
 *     their time-cluster id (gap > right_raw_extension, rawdata.py:63); wfsim_amd/scheduler.py does that.
 */
#ifndef WFSIM_AMD_H
#define WFSIM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WFS_OK 0
#define WFS_E_INVALID (-1)   /* bad argument / shape                                      */
#define WFS_E_HIP (-2)       /* HIP runtime error (message has the call)                  */
#define WFS_E_CAPACITY (-3)  /* caller buffer too small, or the 10^6-sample window assert */
#define WFS_E_STATE (-4)     /* call order (tables / batch not loaded)                    */
#define WFS_E_NOMEM (-5)     /* host allocation failed inside the library (std::bad_alloc) */

typedef struct wfs_handle wfs_handle;

/* Scalars of the hot path.  Filled from the fax config by wfsim_amd.config.kernel_params();
 * the config keys are listed in SURVEY.md appendix A. */
typedef struct wfs_config {
    int32_t dt;                /* sample_duration                                   pulse.py:58      */
    int32_t samples_before;    /* samples_before_pulse_center                       pulse.py:121-127 */
    int32_t samples_after;     /* samples_after_pulse_center                                         */
    int32_t store_before;      /* samples_to_store_before                                            */
    int32_t store_after;       /* samples_to_store_after                                             */
    int32_t tlen;              /* template length = before + after (22)             pulse.py:175     */
    int32_t trigger_window;    /* rawdata.py:215-216, 299-304                                        */
    int32_t baseline;          /* digitizer_reference_baseline                      rawdata.py:270   */
    int32_t n_rows;            /* rows of the digitiser array (801)                 rawdata.py:224   */
    int32_t n_tpc;             /* len(gains)                                                         */
    int32_t n_top;             /* n_top_pmts                                        rawdata.py:243   */
    int32_t he_first;          /* channel_map['he'][0]                              rawdata.py:244   */
    int32_t he_factor;         /* int(high_energy_deamplification_factor)           rawdata.py:242   */
    int32_t sum_channel;       /* channel_map['sum_signal'] (never emitted)              */
    int32_t last_bottom;       /* channels_bottom[-1]                               rawdata.py:250   */
    int32_t detector_nt;       /* detector == 'XENONnT'                             rawdata.py:241   */
    int32_t enable_noise;      /* rawdata.py:263                                                     */
    int32_t s1_simple;         /* 'simple' in s1_model_type                         s1.py:191        */
    int32_t s2_time_model;     /* 0 zero_delay, 1 "s2_time_spread around zero"      s2.py:545-550    */
    int32_t enable_pmt_ap;     /* enable_pmt_afterpulses                            rawdata.py:176   */
    int32_t tile_gen;          /* 1: primary S2s whose tiles fit a pulse workgroup draw their photons there (tile-local generation,
                                  RNG spec v9): Poisson(n_e g p_ch) photons per (instruction, channel), a uniform surviving
                                  electron per photon -- the distribution of s2.py:308 + :673 when s2_gain_spread == 0.  0: every
                                  photon comes from the per-electron generator (needed by the electron-afterpulse pre-pass)   */
    int32_t tile_gen_min;      /* ... and only where amp * gain * max(p_ch) reaches this many photons: below it a workgroup per tile
                                  costs more than the block generator (crossover measured at ~60 photons per tile)        */
    int32_t fma;               /* 1 (config 'fused_multiply_add', the default): Pulse.add_current (pulse.py:303-318) accumulates
                                  template * gain into the current with ONE rounding per term (v_fma_f64) instead of the separately
                                  rounded product and sum of numpy: currents within 1 ulp of the tile maximum per term of the
                                  reference's, ADC samples / ZLE / records equal unless a current lands within that distance of a
                                  rounding tie (tolerance of the path: 1 ADC count).  0: the reference's two roundings, currents
                                  bit-exact                                                                   */
    int32_t row_resident;      /* config 'row_resident'.  1: a (window, channel) row whose pulses all
                                  hold at most 64 photons is made (in segments of 1024 samples), finished and zero-suppressed by one wave in LDS (k_row_pulse);
                                  0: every row through the integer accumulators in HBM; 2 ('auto', the default): 1 for a batch with
                                  at least two photons in the photon array per (pulse set, channel) slot, else 0.  Same records either way.          */
    double c2a;                /* current_2_adc                                     pulse.py:33-35   */
    double tts_mean, tts_sigma;/* pmt_transit_time_mean, spread/2.35482             pulse.py:53-56   */
    double p_dpe;              /* p_double_pe_emision                               pulse.py:76      */
    double s1_decay_time, s1_decay_spread;                    /*                    s1.py:193-194    */
    double sf_gas, t1_gas, t3_gas;                            /* singlet/triplet    pulse.py:333-341 */
    double s2_time_spread;                                    /*                    s2.py:550        */
    double trap_time;          /* electron_trapping_time                            s2.py:280        */
    double gain_spread;        /* s2_gain_spread                                    s2.py:309        */
    double pmt_ap_modifier, pmt_ap_t_modifier;                /*                    afterpulse.py:198,223 */
    double rext;               /* right_raw_extension                               rawdata.py:47    */
    double drift_velocity;     /* drift_velocity_liquid                             rawdata.py:46    */
    uint64_t seed;             /* Philox key (the reference seeds numpy's global generator, strax_interface.py:589) */
} wfs_config;

/* ---- life cycle: replaces RawData.__init__ / Pulse.__init__ (rawdata.py:27-36, pulse.py:24-37) ---------- */
int wfs_create(const wfs_config *cfg, int device, wfs_handle **out);
int wfs_destroy(wfs_handle *h);
const char *wfs_last_error(const wfs_handle *h);
int wfs_device_count(int *n);

/* Init-time tables (host pointers, copied to HBM once).
 *   templates  f64[10][tlen]        Pulse.init_pmt_current_templates            pulse.py:146-187
 *   spe        f64[n_spe][2001]     Pulse.init_spe_scaling_factor_distributions pulse.py:189-223 (n_spe == 1: shared)
 *   gains      f64[n_tpc]           config['gains']
 *   thr_truth  f64[n_rows]          zle/special threshold - 0.5                 pulse.py:240-243
 *   thr_zle    i64[n_rows]          baseline - threshold - 1                    rawdata.py:290-294
 *   lum_x/t    f64[n_lum]           inverse-CDF table of the simple luminescence model, s2.py:317-341
 *   noise      i16[noise_len][noise_channels] or NULL                           load_resource.py:375-376 */
int wfs_set_tables(wfs_handle *h, const double *templates, const double *spe, int32_t n_spe,
                   const double *gains, const double *thr_truth, const int64_t *thr_zle,
                   const double *lum_x, const double *lum_t, int32_t n_lum,
                   const int16_t *noise, int32_t noise_len, int32_t noise_channels);

/* PMT afterpulse element tables (resource.uniform_to_pmt_ap[element], afterpulse.py:181-186), element < 8 */
int wfs_set_ap_element(wfs_handle *h, int32_t element, int32_t n_bins_delay, int32_t n_bins_amp, int32_t amp_2d,
                       int32_t is_uniform, double delay_bin, double amp_bin,
                       const double *delay_cdf, const double *amp_cdf);

/* ---- batch input ------------------------------------------------------------------------------------- */
/* One batch of primary instructions (host pointers; SoA because instruction_dtype is a packed 70-byte record
 * with an unaligned int64, strax_interface.py:25-42).  Sorted by the scheduler key; cluster[] non-decreasing.
 *   type i8, time i64, amp i32               instruction fields
 *   gid u32                                  run-wide instruction index (RNG stream id; shard independent)
 *   cluster i32                              time-cluster index within the batch (rawdata.py:61-63)
 *   tmin i64                                 scheduler key (time - z/v*[S2]) of the instruction
 *   p_hit f64                                S1: light yield (s1.py:125-131); S2: electron survival (s2.py:241-252)
 *   drift_mean/drift_spread f64              s2.py:158-179
 *   sc_gain f64                              s2.py:182-209
 *   cdf_row i32, cdf_table f64[n_cdf][n_tpc] cumulative channel probabilities (np.random.choice, s1.py:154, s2.py:673)
 *   run_set i32 (or NULL), n_run_sets       pulse set of every instruction: the instructions the reference hands to ONE
 *                                           Pulse call (rawdata.py:108-127; save_full_truth=False groups S1s within 100 ns
 *                                           and S2s within 2 mm).  Sets are numbered 0..n_run_sets-1; their instructions
 *                                           share cluster and type.  NULL: one set per instruction (the default).
 *                                           A set number may stay unused (its truth row is empty).  With n_run_sets = n
 *                                           and every set numbered by its FIRST instruction, an S2 that is alone in its
 *                                           set takes the tile-local generator like an S2 of a batch without run sets
 *                                           (electron afterpulses on: the primaries, next to the shared calls of the
 *                                           secondaries); with any other numbering no instruction does.
 *   em_base u32 (or NULL)                   offset of the instruction's emitter ids in the Philox counters: an electron-
 *                                           afterpulse instruction carries its parent's gid and (k + 1) << 20 for the k-th
 *                                           secondary of that parent (streams independent of batching / sharding) */
int wfs_load_instructions(wfs_handle *h, int64_t n, const int8_t *type, const int64_t *time, const int32_t *amp,
                          const uint32_t *gid, const int32_t *cluster, const int64_t *tmin,
                          const double *p_hit, const double *drift_mean, const double *drift_spread,
                          const double *sc_gain, const int32_t *cdf_row, const double *cdf_table, int32_t n_cdf,
                          const int32_t *run_set, int64_t n_run_sets, const uint32_t *em_base);

/* ---- model variants of the photon delays ---------------------------------------------------------------
 * S1.photon_timings (s1.py:162-238: 'custom' recoil models er / nr / alpha / led, s1.py:262-337) and S2.photon_timings
 * (s2.py:504-557: 'garfield' luminescence s2.py:380-411, optical propagation s2.py:486-502) add one more independent,
 * integer-truncated delay term per photon.  The host hands over that term's probability mass function; table k is the
 * convolution of pmf k (support vmin[k] .. vmin[k] + len - 1, pmf[pmf_off[k] .. pmf_off[k + 1])) with base[k]:
 *   0 transit time only, 1 the S1 terms of wfs_config, 2 the S2 terms, 3 the S2 terms without the 'simple' luminescence.
 * Call after wfs_set_tables; n_tables = 0 drops the tables. */
int wfs_set_delay_models(wfs_handle *h, int32_t n_tables, const int32_t *base, const int64_t *pmf_off, const double *pmf,
                         const int32_t *vmin);

/* S1 optical propagation (S1.optical_propagation, s1.py:241-260; the spline is a RegularGridInterpolator over (z, u),
 * load_resource.py:356-357): node values [nz][nu] for top- and bottom-array channels, the uniform u grid u0 + k * du.
 * The delay of a photon is trunc(multilinear interpolation at (z of the instruction, u)), u from the photon's stream. */
int wfs_set_s1_propagation(wfs_handle *h, int32_t nz, int32_t nu, double u0, double du, const double *top, const double *bottom);

/* Per instruction of the batch just loaded (wfs_load_instructions resets them): delay table of its photons on top-array
 * and on bottom-array channels (-1: the default table of the instruction type; tab_bottom NULL: same as tab), and for
 * S1 propagation the z cell of the spline grid and the normalised distance inside it (prop_zi -1 / NULL: none). */
int wfs_set_instruction_models(wfs_handle *h, int64_t n, const int32_t *tab, const int32_t *tab_bottom,
                               const int32_t *prop_zi, const double *prop_zf);
/* s2_luminescence_model 'garfield_gas_gap' (s2.py:413-483; resource s2_luminescence_gg, load_resource.py:284-291): the
 * excitation-time inverse CDFs timing_inv_cdf[n_gas_gaps][n_points] once; then per batch, for every instruction, the table at
 * or below the gas gap under it (np.digitize(gap, gas_gap) - 1; -1: the instruction does not use the model) and
 * (gap - gas_gap[table]) / spacing.  Every photon draws its excitation time from the interpolated table; the mean over the
 * instruction's photons is subtracted (s2.py:447) and the result truncated (s2.py:532).  Needs wfs_set_instruction_models
 * for the same batch (the remaining delay terms come from its tables, base 3). */
int wfs_set_gas_gap_model(wfs_handle *h, int32_t n_gas_gaps, int32_t n_points, const double *timing_inv_cdf);
int wfs_set_instruction_gas_gap(wfs_handle *h, int64_t n, const int32_t *table, const double *weight);

/* ---- pattern maps evaluated on the device ----------------------------------------------------------------
 * The hit pattern of an instruction is resource.s1_pattern_map(x, y, z) / s2_pattern_map(x, y) (s1.py:148, s2.py:640):
 * straxen InterpolatingMaps on a regular grid, method WeightedNearestNeighbors (load_resource.make_patternmap,
 * load_resource.py:403-435): inverse-distance weighted average of the 2 * dims nearest nodes, distances clipped at 1e-6.
 * which: 1 S1 map (3-D), 2 S2 map (2-D); nodes per axis, first / last node per axis; values f32[nodes][n_map_channels]
 * (row-major over the axes, PMT mask already applied).  n_map_channels < n_tpc: the missing (bottom) channels count 1
 * (s2.py:648-650).  Turned-off PMTs (gain 0) are removed on the device (s1.py:150, s2.py:653).  dims = 0 drops the map.
 * An instruction loaded with cdf_row = -1 takes its channel CDF from the map of its type: call wfs_eval_pattern_rows
 * with the positions (float32 as in instruction_dtype; z unused for S2) after wfs_load_instructions, before wfs_run.
 * wfs_copy_cdf_rows returns the rows the generator uses (parity tests feed them to the oracle). */
int wfs_set_pattern_map(wfs_handle *h, int32_t which, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi,
                        const float *values, int32_t n_map_channels);
int wfs_eval_pattern_rows(wfs_handle *h, int64_t n, const float *x, const float *y, const float *z);
/* The same for a map on an irregular coordinate system (a list of points: straxen queries a KD-tree for the 2 * dims
 * nearest points): points f64[n_points][dims], values f32[n_points][n_map_channels]. */
int wfs_set_pattern_map_points(wfs_handle *h, int32_t which, int32_t dims, int64_t n_points, const double *points,
                               const float *values, int32_t n_map_channels);
/* s2_aft_sigma / s2_aft_skewness (S2.photon_channels, s2.py:660-665): factor[i] = the skew-normal draw of instruction i
 * (NaN: leave the pattern alone); on the rows the device evaluates the top-array fraction cur becomes
 * new = clip(cur * factor, 0, 1), top channels are scaled by new / cur and the others by (1 - new) / (1 - cur).
 * Call between wfs_load_instructions and wfs_eval_pattern_rows; factor = NULL clears it. */
int wfs_set_instruction_aft(wfs_handle *h, int64_t n, const double *factor);
/* diffusion_constant_transverse with enable_field_dependencies['diffusion_transverse_map'] (S2.s2_pattern_map_diffuse,
 * s2.py:560-613): the pattern of an S2 instruction is the average of the pattern map over its surviving electrons, each
 * displaced by N(0, sigma_r[i]) along the radius and N(0, sigma_a[i]) across it (cm; sqrt(2 D t) from the field maps, NaN: not
 * this path); electrons that end outside tpc_radius do not count.  Evaluated inside wfs_run, behind the electron survival
 * draws.  Needs the S2 pattern map as a regular grid (wfs_set_pattern_map) and the instruction loaded with cdf_row = -1;
 * call between wfs_load_instructions and wfs_eval_pattern_rows (which still supplies the positions). */
int wfs_set_instruction_diffusion(wfs_handle *h, int64_t n, const double *sigma_r, const double *sigma_a, double tpc_radius);

/* ---- scalar maps evaluated on the device -------------------------------------------------------------------
 * The per-instruction inputs of the generator that the reference reads from straxen InterpolatingMaps: S1 light yield
 * (resource.s1_lce_correction_map, s1.py:125), S2 correction / single-electron gain (s2.py:193-196, 229-234), longitudinal
 * diffusion (s2.py:170), field-distortion corrections (s2.py:41), all WeightedNearestNeighbors on a regular grid
 * (wfs_scalar_map_grid) or a point list (wfs_scalar_map_points); and the field-dependence / COMSOL maps, which
 * load_resource.py:316,326 builds with method RectBivariateSpline (wfs_scalar_map_spline: the knots tx[nx], ty[ny], degrees
 * and coefficients c[(nx-kx-1)*(ny-ky-1)] of scipy's spline object; evaluated like its .ev(): FITPACK bispeu, arguments
 * clamped to the knot range).  Each call registers a map for the lifetime of the handle and returns its id.
 * wfs_scalar_map_eval: pos f64[n][dims] -> out f64[n].  Tolerance against the host evaluation: rtol 1e-6 (different
 * summation order; the nearest neighbours of a position equidistant to several nodes may be chosen differently). */
int wfs_scalar_map_grid(wfs_handle *h, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi,
                        const double *values, int32_t *map_id);
int wfs_scalar_map_points(wfs_handle *h, int32_t dims, int64_t n_points, const double *points, const double *values, int32_t *map_id);
int wfs_scalar_map_spline(wfs_handle *h, int32_t nx, const double *tx, int32_t ny, const double *ty, int32_t kx, int32_t ky,
                          const double *c, int32_t *map_id);
int wfs_scalar_map_eval(wfs_handle *h, int32_t map_id, int64_t n, const double *pos, double *out);
/* The other shapes make_map / InterpolatingMap can take (load_resource.py:357, 383-401): array-valued maps -- n_values entries per node,
 * values f64[nodes][n_values], WeightedNearestNeighbors on a grid or a point list -- and method 'RegularGridInterpolator' on a regular
 * grid (wfs_scalar_map_linear: scipy's multilinear interpolation with bounds_error=False, fill_value=None, i.e. the edge cell's plane
 * continued outside the grid; scalar or array-valued).  wfs_scalar_map_eval_array: pos f64[n][dims] -> out f64[n][n_values];
 * n_values must be the map's (wfs_scalar_map_eval is the n_values = 1 form).  Tolerance as above (multilinear: rtol 1e-12). */
int wfs_scalar_map_grid_array(wfs_handle *h, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi,
                              const double *values, int32_t n_values, int32_t *map_id);
int wfs_scalar_map_points_array(wfs_handle *h, int32_t dims, int64_t n_points, const double *points, const double *values,
                                int32_t n_values, int32_t *map_id);
int wfs_scalar_map_linear(wfs_handle *h, int32_t dims, const int32_t *n_nodes, const double *lo, const double *hi,
                          const double *values, int32_t n_values, int32_t *map_id);
int wfs_scalar_map_eval_array(wfs_handle *h, int32_t map_id, int64_t n, const double *pos, double *out, int32_t n_values);
int wfs_copy_cdf_rows(wfs_handle *h, int32_t *cdf_row, double *cdf_table, int64_t cap_rows);

/* Order of the packed records of a batch: 0 (default) as the reference yields pulses (window, channel, interval,
 * rawdata.py:282-311); 1 as strax.sort_by_time leaves them in ChunkRawRecords.final_results (strax_interface.py:453):
 * by (time, channel) -- the windows of a batch do not overlap in time, so they stay contiguous.  Sorted on the device. */
int wfs_set_record_order(wfs_handle *h, int32_t by_time);

/* Parity entry: photons supplied instead of generated -- what RawDataOptical.sim_primary hands to Pulse
 * (rawdata.py:475-493) and what the golden vectors inject.  One "pulse set" = one Pulse.__call__ (pulse.py:39).
 *   set_cluster i32[n_sets], set_tmin i64[n_sets]   cluster id and scheduler key of the set's instruction
 *   set_off i64[n_sets+1]                           photon ranges; inside a set photons are sorted by channel
 *   t i64 (post transit-time spread), ch i16, gain f64 (pulse.py:97-107), dpe u8 */
int wfs_load_photons(wfs_handle *h, int64_t n_sets, const int32_t *set_cluster, const int64_t *set_tmin,
                     const int64_t *set_off, const int64_t *t, const int16_t *ch, const double *gain,
                     const uint8_t *dpe);

/* Optical input: replaces RawDataOptical.sim_primary (rawdata.py:475-493).  One pulse set per instruction; its photons
 * are channels[first[i]:last[i]] / timings[...] (ns relative to the instruction), cut to 0 <= t < time_cutoff
 * (nveto_time_max_cutoff).  Transit time spread, double-PE and SPE gains are drawn on the GPU (Pulse.__call__). */
int wfs_load_optical(wfs_handle *h, int64_t n, const int64_t *time, const uint32_t *gid, const int32_t *cluster,
                     const int64_t *tmin, const int32_t *first, const int32_t *last,
                     const int32_t *channels, const int64_t *timings, int64_t n_photons, int64_t time_cutoff);

/* State carried across batches: RawData.last_pulse_end_time (rawdata.py:55, 188-190), the running maximum of the end of
 * every pulse simulated so far, which decides whether the first clusters of this batch open a new digitise window. */
int wfs_set_window_carry(wfs_handle *h, int32_t has_pulse, int64_t last_pulse_end_time);

/* ---- run: replaces RawData.__call__ for the loaded batch (rawdata.py:38-157) ---------------------------- */
/* Stages: S1/S2 photon generation (s1.py:60-114, s2.py:73-136) [skipped after wfs_load_photons] ->
 * Pulse.__call__ + add_current (pulse.py:39-144, 276-318) -> digitize_pulse_cache (rawdata.py:204-272) ->
 * ZLE (rawdata.py:274-311) -> record packing (strax_interface.py:391-436).  Asynchronous on the handle's
 * stream except for the size read-backs between stages. */
int wfs_run(wfs_handle *h);

/* n_pulse_sets: rows of the per-set outputs (wfs_copy_truth, wfs_copy_truth_per_pmt, wfs_copy_electron_stats; set offsets of
 * wfs_copy_photons) = n_run_sets as given to wfs_load_instructions, unused set numbers included (their rows are empty), twice that with
 * PMT afterpulses (the afterpulse set of set q is n_run_sets + q).  n_raw_samples: samples of every digitised row, whichever way it was
 * made (accumulators, tile buffer read in place, resident row). */
typedef struct wfs_counts {
    int64_t n_instructions, n_pulse_sets, n_emitters, n_photons /* primary + PMT afterpulse */, n_pe, n_tiles, n_groups, n_rows,
            n_raw_samples, n_intervals, n_records;
} wfs_counts;
int wfs_get_counts(wfs_handle *h, wfs_counts *out);

/* Electron afterpulses need, per parent S2, the number of detected photons and the arrival times of randomly chosen ones
 * (afterpulse.py:37-47, 106-121: len(signal_pulse._photon_timings), _photon_timings[randint]).
 * off[n + 1]: first generated photon of every instruction (generation order: instruction by instruction; inside an instruction of
 * the per-electron generator emitter by emitter, inside a tile-generated one (wfs_config.tile_gen) tile by tile -- channel ascending,
 * photon q of a tile at position q); index: photons in that numbering, t_out: their arrival times in ns (recomputed from the photon's
 * own draws, so the answer does not depend on the order inside the channel buckets -- and, for tile-generated instructions, not on
 * any photon having been generated: debug bit 2 without bit 4 only draws the tiles' photon numbers).  After wfs_run (also with debug bit 2). */
int wfs_copy_instruction_photon_offsets(wfs_handle *h, int64_t *off, int64_t capacity);
int wfs_gather_photon_times(wfs_handle *h, int64_t n, const int64_t *index, int64_t *t_out);

/* Per-PMT truth (config 'per_pmt_truth', pulse.py:61-66, 268-269): acc6[set][channel][6] = n_photon, n_pe, n_photon_trigger,
 * n_pe_trigger, raw_area, raw_area_trigger of every pulse set and TPC channel.  cap = sets the buffer holds. */
int wfs_copy_truth_per_pmt(wfs_handle *h, double *acc6, int64_t cap);

/* ---- results (host copies) ---------------------------------------------------------------------------- */
/* raw_records, 244-byte packed strax layout, in the order the reference yields them (group, channel, interval,
 * fragment); dst may be a host or a device pointer (wfs_copy_records_dev). */
int wfs_copy_records(wfs_handle *h, void *dst_host, int64_t capacity_records);
/* the records [first, first + count) of the batch into a caller buffer (e.g. straight into the chunker's record buffer) */
int wfs_copy_records_range(wfs_handle *h, void *dst, int64_t first, int64_t count);
/* The same without waiting: the copy runs on the handle's copy stream and overlaps the kernels of the next wfs_run (the
 * engine keeps the records of the last two batches); dst should be pinned (wfs_host_register: e.g. the chunker's record
 * buffer, strax_interface.py:360-364, registered once) -- from pageable memory the driver stages the copy and nothing
 * overlaps.  wfs_wait_records returns when every copy issued so far has landed. */
int wfs_copy_records_range_async(wfs_handle *h, void *dst, int64_t first, int64_t count);
int wfs_wait_records(wfs_handle *h);
int wfs_host_register(void *ptr, int64_t bytes);
int wfs_host_unregister(void *ptr);
int wfs_copy_records_dev(wfs_handle *h, void *dst_dev, int64_t capacity_records);
const void *wfs_records_dev_ptr(wfs_handle *h);
/* digitise windows: rawdata.left / rawdata.right and the first record of each window (strax_interface.py:394-399) */
int wfs_copy_groups(wfs_handle *h, int64_t *left, int64_t *right, int64_t *first_record, int64_t *ix_rand);
/* digitise window of every time-cluster of the batch (rawdata.py:96-98 decided on the GPU) */
int wfs_copy_cluster_groups(wfs_handle *h, int32_t *group, int64_t capacity_clusters);
/* ZLE intervals (channel, left, right) and their samples: the tuples RawData.__call__ yields (rawdata.py:311) */
int wfs_copy_intervals(wfs_handle *h, int32_t *group, int32_t *channel, int64_t *left, int64_t *right,
                       int64_t *data_off, int64_t capacity);
int wfs_copy_interval_data(wfs_handle *h, int16_t *data, int64_t capacity);
/* pulses (tiles): channel, left, right, photons, and the f64 currents (pulse.py:138-144) -- parity/debug */
int wfs_copy_pulses(wfs_handle *h, int32_t *set, int32_t *channel, int64_t *left, int64_t *right,
                    int64_t *n_photons, int64_t *cur_off, int64_t capacity);
int wfs_copy_currents(wfs_handle *h, double *cur, int64_t capacity);   /* needs wfs_set_debug(h, 1) before wfs_run */
/* digitised rows in their active range (rawdata.py:258-259) -- parity/debug */
int wfs_copy_rows(wfs_handle *h, int32_t *group, int32_t *channel, int64_t *left, int64_t *right,
                  int64_t *data_off, int64_t capacity);
int wfs_copy_row_data(wfs_handle *h, int32_t *data, int64_t capacity);
/* generated photons per pulse set, channel sorted: t i64, ch i16, gain f64, dpe u8 (after wfs_run) */
int wfs_copy_photons(wfs_handle *h, int64_t *set_off, int64_t *t, int16_t *ch, double *gain, uint8_t *dpe,
                     int64_t capacity);
/* truth accumulators per pulse set: 12 f64 (n_photon n_pe n_photon_trigger n_pe_trigger raw_area raw_area_trigger,
 * then the same for the bottom array; pulse.py:229-271) + photon time stats (n, mean, min, max, std; rawdata.py:325-332) */
int wfs_copy_truth(wfs_handle *h, double *acc12, double *tstat5, int64_t capacity_sets);
/* electron arrival-time statistics per run set (= per instruction by default): n, mean, min, max, std over the electrons of
 * all its instructions (rawdata.py:325-341; NaN when there are none) */
int wfs_copy_electron_stats(wfs_handle *h, double *estat5, int64_t capacity_sets);

/* ---- instrumentation ---------------------------------------------------------------------------------- */
/* flags: bit 0 keep f64 tile currents and finished rows for wfs_copy_currents / wfs_copy_rows; bit 1 send every tile
 * to the dense pulse kernel (both kernels give the same bits; used by the parity tests); bit 2 wfs_run stops after the
 * photon generation (enough for wfs_copy_set_photon_counts / wfs_gather_photon_times: the electron-afterpulse pre-pass);
 * bit 3 check every kernel launch on the spot (hipGetLastError + stream synchronisation after each one: a failed launch or a
 * faulting kernel is reported under its own name by wfs_run; slow, for debugging); bit 4 the photons of tile-generated instructions
 * (wfs_config.tile_gen) are also stored for wfs_copy_photons -- by default they only ever exist in registers */
int wfs_set_debug(wfs_handle *h, int32_t flags);
/* parity tests: noise start index per digitise window (rawdata.py:417) instead of the Philox draw; entries < 0 keep the draw.
 * Indexed by the window number of wfs_copy_groups (host pointer, copied). n = 0 clears the override. */
int wfs_set_noise_offsets(wfs_handle *h, const int64_t *ix_rand, int64_t n);
/* A noise array of floats, f64[noise_len][noise_channels] (call after wfs_set_tables; replaces its int16 table): add_noise
 * (rawdata.py:436, numba) adds noise_data[ix, ch] into the int64 row, which stores the truncated SUM. */
int wfs_set_noise_float(wfs_handle *h, const double *noise, int32_t noise_len, int32_t noise_channels);
int wfs_set_stream(wfs_handle *h, void *hip_stream);
int wfs_synchronize(wfs_handle *h);
/* HIP-event timing of the kernels of the last wfs_run: names (NUL separated) and milliseconds */
int wfs_kernel_times(wfs_handle *h, char *names, int64_t names_cap, float *ms, int32_t *n_launches, int32_t *n_kernels);
int wfs_set_profiling(wfs_handle *h, int32_t on);

#ifdef __cplusplus
}
#endif
#endif
