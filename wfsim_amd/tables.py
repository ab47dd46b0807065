"""Init-time tables of the hot path (host, numpy/scipy like the reference; uploaded once to HBM).

Each builder restates the reference routine it names and is pinned against tables dumped from the reference
(tests/golden/tables.npz, dists.npz):

* ``pmt_current_templates``  -- Pulse.init_pmt_current_templates, /root/reference/wfsim/core/pulse.py:146-187
* ``spe_scaling_table``      -- Pulse.init_spe_scaling_factor_distributions, pulse.py:189-223
* ``luminescence_table``     -- S2.luminescence_timings_simple + the table part of _luminescence_timings_simple,
                                /root/reference/wfsim/core/s2.py:317-378
* ``choice_cdf``             -- the cumulative table np.random.choice(p=...) searches (s1.py:154-158, s2.py:673-677)
* ``thresholds``             -- per-channel ZLE / truth thresholds, rawdata.py:290-294, pulse.py:240-243
"""
import numpy as np
from scipy.interpolate import interp1d

# pax unit system, the four constants of /root/reference/wfsim/units.py used by s2.py:355-359
_electron_charge_SI = 1.602176565 * 10 ** (-19)
_boltzmann_SI = 1.3806488 * 10 ** (-23)
_m, _s, _eV, _K = 10 ** 2, 10 ** 9, 1, 1
_C = 1 / _electron_charge_SI
_J = _eV / _electron_charge_SI
_V = _J / _C
_N = _J / _m
_Pa = _N / _m ** 2
UNIT_bar = 10 ** 5 * _Pa
UNIT_kV = 10 ** 3 * _V
UNIT_cm = 10 ** (-2) * _m
UNIT_boltzmann = _boltzmann_SI * _J / _K


def pmt_current_templates(config):
    """f64[10, 22]: template r is the SPE current for a photon at r ns past a sample edge."""
    pe_pulse_function = interp1d(config.get('pe_pulse_ts'), np.cumsum(config.get('pe_pulse_ys')),
                                 bounds_error=False, fill_value=(0, 1))
    dt = config.get('sample_duration', 10)
    before = config.get('samples_before_pulse_center', 2)
    after = config.get('samples_after_pulse_center', 20)
    rounding = config.get('pmt_pulse_time_rounding', 1.0)
    assert rounding == 1
    samples = np.linspace(-before * dt, +after * dt, 1 + before + after)
    templates = []
    for r in np.arange(0, dt, rounding):
        cur = np.diff(pe_pulse_function(samples - r)) / dt
        cur *= (1 / dt) / np.sum(cur)
        templates.append(cur)
    return np.ascontiguousarray(np.array(templates), dtype=np.float64)


def spe_scaling_table(charge, pdfs):
    """f64[n_ch, 2001]: inverse CDF of the SPE area distribution on a 2001-point uniform grid."""
    rows = []
    grid_cdf = np.linspace(0, 1, 2001)
    for pdf in pdfs:
        if pdf.sum() > 0:
            scaled_bins = charge
            cdf = np.cumsum(pdf) / np.sum(pdf)
        else:
            cdf = np.linspace(0, 1, 10)
            scaled_bins = np.zeros_like(cdf)
        rows.append(interp1d(cdf, scaled_bins, kind='next', bounds_error=False,
                             fill_value=(scaled_bins[0], scaled_bins[-1]))(grid_cdf))
    return np.ascontiguousarray(np.stack(rows), dtype=np.float64)


def luminescence_table(config, gas_gap=None):
    """(x, t): np.interp(u, x, t) is the emission delay of the 'simple' luminescence model for one gas gap."""
    c = config
    number_density_gas = c['pressure'] / (UNIT_boltzmann * c['temperature'])
    alpha = c['gas_drift_velocity_slope'] / number_density_gas
    uE = UNIT_kV / UNIT_cm
    pressure = c['pressure'] / UNIT_bar
    dG = np.ones(1) * (c['elr_gas_gap_length'] if gas_gap is None else gas_gap)
    rA = c['anode_field_domination_distance']
    rW = c['anode_wire_radius']
    dL = c['gate_to_anode_distance'] - dG
    VG = c['anode_voltage'] / (1 + dL / dG / c['lxe_dielectric_constant'])
    E0 = VG / ((dG - rA) / rA + np.log(rA / rW))
    dr = 0.0001
    r = np.arange(np.max(dG), rW, -dr)
    rr = np.clip(1 / r, 1 / rA, 1 / rW)
    dt = dr / (alpha * E0[0] * rr)
    dy = E0[0] * rr / uE - 0.8 * pressure
    avgt = np.sum(np.cumsum(dt) * dy) / np.sum(dy)
    j = np.argmax(r <= dG[0])
    t = np.cumsum(dt[j:]) - avgt
    y = np.cumsum(dy[j:])
    return np.ascontiguousarray(y / y[-1]), np.ascontiguousarray(t)


def choice_cdf(p):
    """Rows of cumulative probabilities searched with side='right', as numpy's legacy ``choice`` builds them."""
    p = np.asarray(p, dtype=np.float64)
    cdf = p.cumsum(axis=-1)
    cdf /= cdf[..., -1:]
    return np.ascontiguousarray(cdf)


def thresholds(config, n_rows):
    """(thr_truth f64[n_rows], thr_zle i64[n_rows])"""
    special = config.get('special_thresholds', {})
    thr = np.full(n_rows, config['zle_threshold'], dtype=np.float64)
    for k, v in special.items():
        thr[int(k)] = v
    thr_truth = thr - 0.5
    thr_zle = (config['digitizer_reference_baseline'] - thr - 1).astype(np.int64)
    return thr_truth, thr_zle
