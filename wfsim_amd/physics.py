"""Per-instruction quantities evaluated on the host from (callable) maps before a batch goes to the GPU.

Maps are host callables at the drop-in boundary (straxen.InterpolatingMap in production, DummyMap in tests),
so everything that touches a map is evaluated here, vectorised over the batch, exactly as the reference does:

* S1 hit probability        -- S1.get_n_photons,  /root/reference/wfsim/core/s1.py:117-135 (the Binomial is drawn on the GPU)
* S1 channel probabilities  -- S1.photon_channels, s1.py:138-159 (the categorical draw happens on the GPU)
* S2 drift time mean/spread -- S2.get_s2_drift_time_params, /root/reference/wfsim/core/s2.py:158-179
* S2 electron survival      -- S2.get_electron_yield, s2.py:212-256 (Binomial drawn on the GPU)
* S2 secondary gain         -- S2.get_s2_light_yield, s2.py:182-209
* S2 channel probabilities  -- S2.photon_channels, s2.py:616-682
"""
import numpy as np

from .resource import DummyMap
from .tables import choice_cdf


def s1_hit_probability(instructions, config, resource):
    positions = np.array([instructions['x'], instructions['y'], instructions['z']]).T
    ly = resource.s1_lce_correction_map(positions)
    if len(ly.shape) != 1:
        ly = np.squeeze(ly, axis=-1)
    ly = ly / (1 + config['p_double_pe_emision'])
    ly = ly * config['s1_detection_efficiency']
    return np.asarray(ly, dtype=np.float64)


def _turned_off(config):
    gains = np.asarray(config['gains'])
    return np.arange(len(gains))[gains == 0]


def s1_channel_probabilities(instructions, config, resource):
    positions = np.array([instructions['x'], instructions['y'], instructions['z']]).T
    channels = np.arange(config['n_tpc_pmts'])
    p = np.array(resource.s1_pattern_map(positions), dtype=np.float64)
    p[:, np.isin(channels, _turned_off(config))] = 0
    return p / np.sum(p, axis=1)[:, None]


def s2_drift_time_params(z, xy, config, resource):
    efd = config['enable_field_dependencies']
    if efd.get('drift_speed_map', False):
        v = resource.field_dependencies_map(z, xy, map_name='drift_speed_map')
        v = v * 1e-4
        v = v * resource.drift_velocity_scaling
    else:
        v = config['drift_velocity_liquid']
    if efd.get('diffusion_longitudinal_map', False):
        dl = resource.diffusion_longitudinal_map(z, xy)
    else:
        dl = config['diffusion_constant_longitudinal']
    mean = - z / v + config['drift_time_gate']
    mean = np.clip(mean, 0, np.inf)
    spread = np.sqrt(2 * dl * mean)
    spread /= v
    return mean, spread


def s2_electron_survival(z, xy, positions, config, resource):
    mean, _ = s2_drift_time_params(z, xy, config, resource)
    if config.get('ext_eff_from_map', False):
        rel = resource.s2_correction_map(positions).flatten()
        if config.get('se_gain_from_map', False):
            se = resource.se_gain_map(positions)
        else:
            se = rel * config['s2_secondary_sc_gain']
        cy = config['g2_mean'] * rel / se
    else:
        cy = config['electron_extraction_yield']
    cy = cy * np.exp(- 1 * mean / config['electron_lifetime_liquid'])
    if config['enable_field_dependencies'].get('survival_probability_map', False):
        p_surv = resource.field_dependencies_map(z, xy, map_name='survival_probability_map')
        if np.any(p_surv < 0) or np.any(p_surv > 1):
            p_surv = np.clip(p_surv, a_min=0, a_max=1)
        cy = cy * p_surv
    return np.asarray(cy, dtype=np.float64) * np.ones(len(z))


def s2_secondary_gain(positions, config, resource):
    if config.get('se_gain_from_map', False):
        sc_gain = np.array(resource.se_gain_map(positions), dtype=np.float64)
    else:
        sc_gain = np.array(resource.s2_correction_map(positions), dtype=np.float64)
        sc_gain *= config['s2_secondary_sc_gain']
    if len(sc_gain.shape) != 1:
        sc_gain = np.squeeze(sc_gain, axis=-1)
    sc_gain /= 1 + config['p_double_pe_emision']
    sc_gain[np.isnan(sc_gain)] = 0
    return sc_gain


def s2_channel_probabilities(positions, config, resource, gids=None):
    """S2.photon_channels up to the categorical draw, /root/reference/wfsim/core/s2.py:616-682.

    ``diffusion_constant_transverse > 0`` sends the reference through s2_pattern_map_diffuse (s2.py:560-613), which reads
    the constant with ``getattr(config, ...)`` on a dict and therefore always diffuses by 0: every electron sits at the
    instruction's xy and the averaged pattern IS the pattern at xy (instructions outside ``tpc_radius`` get no pattern).
    Only with the ``diffusion_transverse_map`` field maps is there a real spread: that average over the surviving electrons
    is made on the device (``s2_transverse_sigmas``, wfs_set_instruction_diffusion) and needs the pattern map there.
    ``s2_aft_sigma``: the top-array fraction of every instruction's pattern is rescaled by a skew-normal factor
    (s2.py:660-665); the draw comes from a host Philox stream keyed by (seed, run-wide instruction id)."""
    channels = np.arange(config['n_tpc_pmts']).astype(np.int64)
    bottom_index = np.array(config['channels_bottom'])
    pattern = np.array(resource.s2_pattern_map(positions), dtype=np.float64)
    if config.get('diffusion_constant_transverse', 0) > 0:
        if config.get('enable_field_dependencies', {}).get('diffusion_transverse_map', False):
            raise NotImplementedError('diffusion_transverse_map (s2.py:575-579) averages the pattern over the electrons the GPU draws: '
                                      'it needs the S2 pattern map on the device as a regular grid (an InterpolatingMap, device_pattern_maps)')
        outside = np.sum(np.asarray(positions, dtype=np.float64) ** 2, axis=1) > config['tpc_radius'] ** 2     # s2.py:598
        pattern[outside] = 0
    if pattern.shape[1] - 1 not in bottom_index:
        pattern = np.pad(pattern, [[0, 0], [0, len(bottom_index)]], 'constant', constant_values=1)
    pattern[:, np.isin(channels, _turned_off(config))] = 0
    sum_pat = np.sum(pattern, axis=1).reshape(-1, 1)
    pattern = np.divide(pattern, sum_pat, out=np.zeros_like(pattern), where=sum_pat != 0)
    assert pattern.shape[1] == len(channels)
    if config.get('s2_aft_sigma', 0.0) != 0:
        top_index = np.arange(config['n_top_pmts'])
        factor = s2_aft_factors(len(pattern), config, gids)
        for i, pat in enumerate(pattern):
            if pat.sum() == 0:
                continue
            cur_aft = np.sum(pat[top_index]) / np.sum(pat)
            new_aft = np.clip(cur_aft * factor[i], 0, 1)
            pat[top_index] *= (new_aft / cur_aft)
            pat[bottom_index] *= (1 - new_aft) / (1 - cur_aft)
    return pattern


def s2_transverse_sigmas(z_obs, xy_obs, config, resource):
    """(sigma_radial, sigma_azimuthal) in cm of every instruction's electron cloud at the liquid surface,
    S2.s2_pattern_map_diffuse (s2.py:572-589) with the ``diffusion_transverse_map`` field maps: sqrt(2 D t), D from the
    radial / azimuthal diffusion maps (cm^2/s) at the observed position, t = -z / v with get_avg_drift_velocity (s2.py:139-155)"""
    z_obs = np.asarray(z_obs, dtype=np.float64)
    assert np.all(z_obs < 0), 'All S2 in liquid should have z < 0'
    if config['enable_field_dependencies'].get('drift_speed_map', False):
        v = resource.field_dependencies_map(z_obs, xy_obs, map_name='drift_speed_map') * 1e-4 * resource.drift_velocity_scaling
    else:
        v = config['drift_velocity_liquid']
    d_r = resource.field_dependencies_map(z_obs, xy_obs, map_name='diffusion_radial_map') * 1e-9          # cm^2 / ns
    d_a = resource.field_dependencies_map(z_obs, xy_obs, map_name='diffusion_azimuthal_map') * 1e-9
    t = - z_obs / v
    return np.sqrt(2 * d_r * t), np.sqrt(2 * d_a * t)


def s2_aft_factors(n, config, gids=None):
    """the skew-normal factor on the top-array fraction of every S2 instruction's pattern (s2.py:661: one
    ``skewnorm.rvs(loc=1, scale=s2_aft_sigma, a=s2_aft_skewness)`` per instruction), from a host Philox stream keyed by
    (seed, run-wide instruction id): independent of batching and of whether the rows are made on the host or the device"""
    from numpy.random import Generator, Philox
    from scipy.stats import skewnorm
    g = np.arange(n) if gids is None else np.asarray(gids)
    seed, sigma, skew = int(config.get('seed', 0) or 0), config['s2_aft_sigma'], config.get('s2_aft_skewness', 0.0)
    return np.array([skewnorm.rvs(loc=1.0, scale=sigma, a=skew, random_state=Generator(Philox(key=[seed, (int(g[i]) << 8) | 0x41])))
                     for i in range(n)], dtype=np.float64)


def s2_observed_positions(instructions, config, resource):
    """(z_obs, xy_obs) of S2-like instructions under the field distortion models of S2.__call__
    (/root/reference/wfsim/core/s2.py:29-71, 81-88): ``inverse_fdc`` (XENON1T: the data-driven correction map applied
    backwards, six damped iterations), ``comsol`` (radial distortion map), anything else: the true positions."""
    x, y, z = (instructions[f].astype(np.float64) for f in ('x', 'y', 'z'))
    model = config.get('field_distortion_model', 'none')
    if model == 'inverse_fdc':
        positions = np.array([x, y, z]).T
        for i_iter in range(6):
            dr = resource.fdc_3d(positions)
            if i_iter > 0:
                dr = 0.5 * dr + 0.5 * dr_pre
            dr_pre = dr
            r_obs = np.sqrt(x ** 2 + y ** 2) - dr
            x_obs = x * r_obs / (r_obs + dr)
            y_obs = y * r_obs / (r_obs + dr)
            z_obs = - np.sqrt(z ** 2 + dr ** 2)
            positions = np.array([x_obs, y_obs, z_obs]).T
        return z_obs, np.array([x_obs, y_obs]).T
    if model == 'comsol':
        theta = np.arctan2(y, x)
        r_obs = resource.fd_comsol(np.array([np.sqrt(x ** 2 + y ** 2), z]).T, map_name='r_distortion_map')
        return z, np.array([r_obs * np.cos(theta), r_obs * np.sin(theta)]).T
    return z, np.array([x, y]).T


def instruction_time(instructions, config):
    """Ordering key of the scheduler, /root/reference/wfsim/core/rawdata.py:61 (float32 arithmetic, as numpy does)."""
    v = config['drift_velocity_liquid']
    return instructions['time'] + (instructions['z'] / v * (instructions['type'] % 2 - 1)).astype(np.int64)


def instruction_params(instructions, config, resource, gids=None, device_maps=(), device_aft=True):
    """Batch arrays for the device: hit/survival probability, drift parameters, secondary gain and the
    cumulative channel table of every instruction (rows de-duplicated).  ``device_maps``: kinds ('s1', 's2') whose pattern
    map lives on the device (Engine.device_maps): their instructions get ``cdf_row = -1`` and no host row."""
    n = len(instructions)
    is_s1 = instructions['type'] == 1
    # types 4 (photo-ionisation electrons) and 6 (photo-electric electrons) are simulated by S2.__call__ (afterpulse.py:14, 94)
    is_s2 = np.isin(instructions['type'], (2, 4, 6))
    if not np.all(is_s1 | is_s2):
        raise NotImplementedError('instruction types on the MI355X path: 1 (S1), 2 (S2), 4 / 6 (electron afterpulses)')
    p_hit = np.zeros(n)
    drift_mean = np.zeros(n)
    drift_spread = np.zeros(n)
    sc_gain = np.zeros(n)
    n_ch = config['n_tpc_pmts']
    rows = []
    cdf_row = np.zeros(n, dtype=np.int32)
    pattern_xy = np.array([instructions['x'], instructions['y']], dtype=np.float64).T      # where the pattern map is evaluated
    outside = np.zeros(n, dtype=bool)         # S2 positions without a pattern (s2.py:598, see s2_channel_probabilities)
    aft_factor = np.full(n, np.nan)
    diff_sigma = (np.full(n, np.nan), np.full(n, np.nan))      # diffusion_transverse_map: the pattern is averaged over the electrons on the device
    for sel, kind in ((is_s1, 's1'), (is_s2, 's2')):
        if not sel.any():
            continue
        ins = instructions[sel]
        idx = np.where(sel)[0]
        if kind == 's1':
            p_hit[sel] = s1_hit_probability(ins, config, resource)
            pmap = resource.s1_pattern_map

            def probs(k):                      # channel probabilities of the first k instructions of this kind
                return s1_channel_probabilities(ins[:k], config, resource)
            per_instruction = False
        else:
            xy = np.array([ins['x'], ins['y']]).T
            # survival and drift use the true position, the S2 maps (correction, gain, pattern) the observed one (s2.py:81-103)
            xy_obs = s2_observed_positions(ins, config, resource)[1] if config.get('field_distortion_model', 'none') in ('inverse_fdc', 'comsol') else xy
            pattern_xy[sel] = xy_obs
            p_hit[sel] = s2_electron_survival(ins['z'], xy, xy_obs, config, resource)
            drift_mean[sel], drift_spread[sel] = s2_drift_time_params(ins['z'], xy, config, resource)
            sc_gain[sel] = s2_secondary_gain(xy_obs, config, resource)
            pmap = resource.s2_pattern_map
            sel_gids = None if gids is None else np.asarray(gids)[sel]

            def probs(k):
                return s2_channel_probabilities(xy_obs[:k], config, resource, None if sel_gids is None else sel_gids[:k])
            transverse_maps = (config.get('diffusion_constant_transverse', 0) > 0
                               and config.get('enable_field_dependencies', {}).get('diffusion_transverse_map', False))
            if transverse_maps:
                if 's2' not in device_maps:
                    raise NotImplementedError('diffusion_transverse_map (s2.py:575-579) needs the S2 pattern map on the device (a two-dimensional WeightedNearestNeighbors map: regular grid or point list)')
                z_obs = s2_observed_positions(ins, config, resource)[0] if config.get('field_distortion_model', 'none') in ('inverse_fdc', 'comsol') else ins['z']
                diff_sigma[0][idx], diff_sigma[1][idx] = s2_transverse_sigmas(z_obs, np.asarray(xy_obs, dtype=np.float64), config, resource)
            elif config.get('diffusion_constant_transverse', 0) > 0:
                outside[idx] = np.sum(np.asarray(xy_obs, dtype=np.float64) ** 2, axis=1) > config['tpc_radius'] ** 2
            per_instruction = config.get('s2_aft_sigma', 0.0) != 0 or bool(outside[idx].any())
        if kind in device_maps and not (kind == 's2' and config.get('s2_aft_sigma', 0.0) != 0 and not device_aft):
            cdf_row[idx] = -1                  # the row comes from the device map; instructions without a pattern make no photons
            p_hit[idx[outside[idx]]] = 0.0
            if kind == 's2' and config.get('s2_aft_sigma', 0.0) != 0:        # the device rescales its rows (wfs_set_instruction_aft)
                aft_factor[idx] = s2_aft_factors(len(idx), config, sel_gids)
        elif isinstance(pmap, DummyMap) and not per_instruction:
            cdf_row[idx] = len(rows)           # a constant map: one shared row
            rows.append(choice_cdf(probs(1))[0])
        else:
            p = probs(len(ins))
            empty = p.sum(axis=1) == 0          # no pattern (all PMTs off / outside the TPC): the instruction makes no photons
            if empty.any():
                p[empty] = 1.0
                p_hit[idx[empty]] = 0.0
            cdf_row[idx] = len(rows) + np.arange(len(idx))
            rows.extend(list(choice_cdf(p)))
    cdf_table = np.ascontiguousarray(np.stack(rows)) if rows else np.ones((1, n_ch))
    return dict(p_hit=p_hit, drift_mean=drift_mean, drift_spread=drift_spread, sc_gain=sc_gain,
                cdf_row=cdf_row, cdf_table=cdf_table, pattern_xy=pattern_xy,
                aft_factor=None if np.all(np.isnan(aft_factor)) else aft_factor,
                diff_sigma=None if np.all(np.isnan(diff_sigma[0])) else diff_sigma)
