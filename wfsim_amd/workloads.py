"""Synthetic workloads of the BASELINE configurations: instruction arrays, configurations and the synthetic resource tables
(PMT afterpulse CDFs, noise, a position dependent S2 pattern map) that stand in for the private XENONnT resource files.

One place for `bench.py`, the profiling tools and the tests (the builders used to live in the test package).  Everything is a
plain numpy construction from a fixed seed, so two processes build identical workloads:

* ``s2_batch`` / ``bench_config``            BASELINE configs[2]: 10^4-electron S2s (~10^6 PE each), dummy (flat) pattern map
* ``s2map_config``                           the same batch under a position dependent S2 pattern map (the PMT above the event takes a
                                             few percent of the light: tiles from a few hundred to several 10^4 photons)
* ``s1_batch``                               configs[1]: 10^4 S1s of ~200 PE
* ``mixed_batch`` / ``mixed_config``         configs[3]: S1 + S2 pairs over the TPC, PMT afterpulses and noise on
* ``nveto_config`` / ``optical_instructions``  configs[4]: optical nVeto instructions at a given rate on 120 channels
"""
import numpy as np

from .config import xenonnt_test_config
from .dtypes import instruction_dtype, optical_extra_dtype

N_TPC, N_TOP = 494, 253


# ---------------------------------------------------------------------------------------------- synthetic resource tables
def synthetic_afterpulse_tables(seed=6, n_channels=N_TPC):
    """PMT afterpulse tables of the shape resource.uniform_to_pmt_ap has (/root/reference/wfsim/core/afterpulse.py:181-186): element ->
    delaytime_cdf[n_ch, n_bins] (NOT normalised: the last value is the afterpulse probability), amplitude_cdf[n_ch, n_bins] or [n_bins],
    bin sizes; the 'Uniform' element has delaytime_cdf[n_ch, 2].  Made up: Gaussian delay / amplitude shapes, 2 % and 1.2 % probability
    scattered by +-50 % over the channels.  (The golden fixture tests/golden/pmt_ap_tables.npz holds exactly these arrays.)"""
    rng = np.random.default_rng(seed)
    nb = 200
    out = {}
    for name, p_ap, mean_delay, amp2d in [('He', 0.02, 60, True), ('Xe', 0.012, 140, False)]:
        x = np.arange(nb)
        shape = np.cumsum(np.exp(-0.5 * ((x - mean_delay) / 15.0) ** 2))
        shape /= shape[-1]
        prob = p_ap * rng.uniform(0.5, 1.5, n_channels)
        dcdf = shape[None, :] * prob[:, None]
        a = np.cumsum(np.exp(-0.5 * ((np.arange(100) - 25) / 8.0) ** 2))
        a /= a[-1]
        acdf = np.repeat(a[None, :], n_channels, axis=0) if amp2d else a
        out[name] = dict(delaytime_cdf=dcdf, amplitude_cdf=acdf, delaytime_bin_size=10.0, amplitude_bin_size=0.04)
    out['Uniform'] = dict(delaytime_cdf=np.stack([np.full(n_channels, 0.004), np.full(n_channels, 0.008)], axis=1),
                          amplitude_cdf=np.ones(3), delaytime_bin_size=1000.0, amplitude_bin_size=1.0)
    return out


def synthetic_noise(seed=5, n_samples=3000, n_channels=N_TPC):
    """int16 noise array [n_samples, n_channels], N(0, 2.2 ADC) rounded (the shape load_resource.py:375-376 loads)"""
    rng = np.random.default_rng(seed)
    return np.round(rng.normal(0, 2.2, (n_samples, n_channels))).astype(np.int16)


def synthetic_electron_afterpulses(n_bins=140, t_max=150e3, tau_bins=30.0, total=3e-3):
    """(histogram, bin_edges) of the photo-ionisation delay: the shape load_resource.py:233 reads from a private file -- a
    falling delay spectrum over 150 us whose integral is the probability per photon"""
    hist = np.exp(-np.arange(n_bins) / tau_bins)
    hist *= total / hist.sum()
    return hist, np.linspace(0.0, t_max, n_bins + 1)


def synthetic_garfield_table(seed=4242, n_x=11, n_samples=4000):
    """dict(t[n_x, n_samples], x[n_x]): sampled luminescence delays per distance to the anode wire, the layout of the garfield
    file of load_resource.py:293-309 (the file itself is private)"""
    rng = np.random.default_rng(seed)
    x = np.linspace(-0.25, 0.25, n_x)
    return dict(t=np.stack([rng.gamma(3.0 + 8 * abs(v), 60.0, n_samples) + 900 for v in x]), x=x)


def synthetic_pmt_positions(n_tpc=N_TPC, n_top=N_TOP):
    """a made-up PMT layout: sunflower rings of 48 cm radius, top array first"""
    def rings(n):
        k = np.arange(n)
        r = 48.0 * np.sqrt((k + 0.5) / n)
        phi = k * 2.399963229728653
        return np.stack([r * np.cos(phi), r * np.sin(phi)], axis=1)
    return np.concatenate([rings(n_top), rings(n_tpc - n_top)])


def synthetic_s2_pattern_map(n_grid=61, top_width=7.0, aft=0.75):
    """An S2 hit pattern on a regular (x, y) grid in the format of the pattern-map files (coordinate_system + map[nx][ny][n_pmt],
    load_resource.py:404-433): a top PMT sees 1 / (1 + d^2 / w^2)^1.5 of the light emitted d cm away from its axis (the PMT above the
    event takes a few percent of all photons), the bottom array sees it almost uniformly; `aft` of the light goes to the top array."""
    xy = synthetic_pmt_positions()
    g = np.linspace(-66.0, 66.0, n_grid)
    d2 = (g[:, None, None] - xy[None, None, :, 0]) ** 2 + (g[None, :, None] - xy[None, None, :, 1]) ** 2       # [nx][ny][pmt]
    top = 1.0 / (1.0 + d2[..., :N_TOP] / top_width ** 2) ** 1.5
    bottom = 1.0 / (1.0 + d2[..., N_TOP:] / 60.0 ** 2)
    top *= aft / top.sum(axis=-1, keepdims=True)
    bottom *= (1.0 - aft) / bottom.sum(axis=-1, keepdims=True)
    m = np.concatenate([top, bottom], axis=-1).astype(np.float32)
    return dict(coordinate_system=[['x', [-66.0, 66.0, n_grid]], ['y', [-66.0, 66.0, n_grid]]], map=m)


# ---------------------------------------------------------------------------------------------- instruction arrays
def s2_batch(n, first_gid=0, t0=0, electrons=10_000, spread_xy=False, seed=2):
    """BASELINE configs[2]: n S2 instructions, 10^4 electrons each, 1 ms apart, z = -10 cm; at the centre of the TPC, or (spread_xy)
    uniformly over a disc of 45 cm radius (a pattern map then gives every instruction its own hit pattern)"""
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = 2
    ins['time'] = t0 + 1_000_000 * (1 + np.arange(n))
    ins['z'] = -10.0
    ins['amp'] = electrons
    ins['recoil'] = 7
    ins['event_number'] = first_gid + np.arange(n)
    if spread_xy:
        rng = np.random.default_rng(seed + first_gid)
        r, phi = 45 * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
        ins['x'], ins['y'] = r * np.cos(phi), r * np.sin(phi)
    return ins


def s1_batch(n, first_gid=0, quanta=1667, seed=2):
    """BASELINE configs[1]: n S1 instructions of ~200 PE, 1 ms apart, uniform over the TPC"""
    rng = np.random.default_rng(seed + first_gid)
    ins = np.zeros(n, dtype=instruction_dtype)
    ins['type'] = 1
    ins['time'] = 1_000_000 * (1 + np.arange(n))
    r, phi = 50 * np.sqrt(rng.random(n)), rng.uniform(0, 2 * np.pi, n)
    ins['x'], ins['y'], ins['z'] = r * np.cos(phi), r * np.sin(phi), -rng.uniform(0, 97, n)
    ins['amp'] = quanta
    ins['recoil'] = 7
    ins['event_number'] = first_gid + np.arange(n)
    return ins


def mixed_batch(n, first_gid=0):
    """BASELINE configs[3]: n / 2 events, an S1 (3000 quanta) and an S2 (1500 electrons) each, 1 ms apart, all over the TPC"""
    n_ev = max(n // 2, 1)
    rng = np.random.default_rng(4 + first_gid)
    ins = np.zeros(2 * n_ev, dtype=instruction_dtype)
    ins['type'] = np.tile([1, 2], n_ev)
    ins['time'] = np.repeat(1_000_000 * (1 + np.arange(n_ev)), 2)
    r, phi = 45 * np.sqrt(rng.random(n_ev)), rng.uniform(0, 2 * np.pi, n_ev)
    ins['x'], ins['y'], ins['z'] = np.repeat(r * np.cos(phi), 2), np.repeat(r * np.sin(phi), 2), np.repeat(-rng.uniform(1, 95, n_ev), 2)
    ins['amp'] = np.tile([3000, 1500], n_ev)
    ins['recoil'] = 7
    ins['event_number'] = first_gid + np.arange(2 * n_ev)
    return ins


def optical_instructions(n, rate_ns, seed):
    """BASELINE configs[4]: n optical instructions (mean spacing rate_ns), ~10 photons each on 120 channels, arrival times
    exponential with 60 ns, 1 % of the photons outside the accepted window on either side (rawdata.py:470-476)"""
    rng = np.random.default_rng(seed)
    ins = np.zeros(n, dtype=instruction_dtype + optical_extra_dtype)
    ins['type'] = 1
    ins['time'] = 1_000_000 + np.cumsum(rng.exponential(rate_ns, n)).astype(np.int64)
    nph = rng.poisson(10, n)
    ins['_first'] = np.concatenate([[0], np.cumsum(nph)[:-1]])
    ins['_last'] = np.cumsum(nph)
    ins['amp'] = nph
    ins['event_number'] = np.arange(n)
    tot = int(nph.sum())
    channels = rng.integers(0, 120, tot)
    timings = rng.exponential(60, tot).astype(np.int64)
    timings[rng.random(tot) < 0.01] = -5          # a few photons outside the accepted window
    timings[rng.random(tot) < 0.01] = 2_000_000
    return ins, channels, timings


# ---------------------------------------------------------------------------------------------- configurations
def bench_config(seed, pmt_afterpulses=False, reference_defaults=False, **overrides):
    """configs[2]: E[PE] = 10^4 e- * survival * sc_gain 100 = ~10^6 PE per instruction; dummy maps, noise and
    afterpulses off (pmt_afterpulses: the side measurement with synthetic afterpulse tables; reference_defaults: the side measurement
    with electron afterpulses on, the reference's default (rawdata.py:194), and the garfield luminescence model XENONnT's fax config
    selects -- both on synthetic tables)"""
    kw = dict(s2_secondary_sc_gain=100.0, seed=seed)
    if pmt_afterpulses:
        kw.update(enable_pmt_afterpulses=True, uniform_to_pmt_ap=synthetic_afterpulse_tables())
    if reference_defaults:
        kw.update(enable_electron_afterpulses=True, uniform_to_ele_ap=synthetic_electron_afterpulses(),
                  s2_luminescence_model='garfield', s2_luminescence=synthetic_garfield_table())
    kw.update(overrides)
    return xenonnt_test_config(**kw)


def s2map_config(seed, **overrides):
    """configs[2] under the synthetic position dependent S2 pattern map (evaluated on the device)"""
    return bench_config(seed, s2_pattern_map=synthetic_s2_pattern_map(), **overrides)


def mixed_config(seed, **overrides):
    """configs[3]: PMT afterpulses and noise on, synthetic tables (the real resource files are private)"""
    kw = dict(seed=seed, enable_pmt_afterpulses=True, uniform_to_pmt_ap=synthetic_afterpulse_tables(), enable_noise=True, noise_data=synthetic_noise())
    kw.update(overrides)
    return xenonnt_test_config(**kw)


def nveto_config(**kw):
    """configs[4]: 120 nVeto channels (one dead PMT), right_raw_extension 2 us so that a 1 MHz instruction stream still breaks into
    clusters"""
    kw.setdefault('right_raw_extension', 2000)
    c = xenonnt_test_config(detector='XENONnT_neutron_veto', **kw)
    n = 120
    c['gains'] = np.full(n, 2e6)
    c['gains'][7] = 0.0                       # one dead PMT
    c['n_tpc_pmts'], c['n_top_pmts'] = n, 0
    c['channels_bottom'] = np.array([], dtype=np.int64)
    c['channel_map'] = dict(nveto=(2000, 2119), sum_signal=800, he=(500, 752))
    c['photon_area_distribution'] = dict(c['photon_area_distribution'], n_channels=n)
    return c
