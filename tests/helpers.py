"""Shared helpers of the test-suite: build oracle / device sessions from a config, replay golden chains."""
import os

import numpy as np

from wfsim_amd.config import xenonnt_test_config  # noqa: F401

from wfsim_amd import tables as T
from wfsim_amd.config import kernel_params, N_ROWS
from wfsim_amd.resource import Resource

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


# Pulse.add_current with fused multiply-adds (config 'fused_multiply_add', the default of the HIP path; wfs_config.fma): every
# term template * gain is added to the current with one rounding instead of numpy's two.  A sample sums at most tlen x (photons in
# reach) terms; the two forms differ by at most half an ulp of the running sum per term, so by a few ulp of the tile maximum at the
# end.  Tolerance of the golden comparisons in that mode (measured maximum on the golden chains: 2 ulp); everything downstream of the
# per-pulse rounding (rows, ZLE intervals, records) is asserted EQUAL in both modes.
FMA_CURRENT_TOL_ULP = 8


def with_fma(config, on):
    return dict(config, fused_multiply_add=bool(on))


def assert_currents_close(cur, ref, what=''):
    tol = FMA_CURRENT_TOL_ULP * np.spacing(np.abs(ref).max()) if len(ref) else 0.0
    assert np.all(np.abs(cur - ref) <= tol), f'{what}: max diff {np.abs(cur - ref).max()} > {tol}'


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def params_chain_config():
    """the config of golden chain F (tests/golden/make_golden.py: params_overrides): non-default digitiser / ZLE / pulse
    window settings, non-uniform gains, three turned-off PMTs"""
    import json
    ov = json.load(open(os.path.join(GOLDEN, 'chain_params_config.json')))
    ov['gains'] = np.asarray(ov['gains'], dtype=np.float64)
    ov['turned_off_pmts'] = np.asarray(ov['turned_off_pmts'])
    return xenonnt_test_config(**ov)


def geometry_chain_config():
    """the config of golden chain I (make_golden.py: geometry_overrides): 5 ns samples, 3 + 37 template samples"""
    import json
    return xenonnt_test_config(**json.load(open(os.path.join(GOLDEN, 'chain_geometry_config.json'))))


def host_tables(config, resource=None):
    resource = resource or Resource(config)
    thr_truth, thr_zle = T.thresholds(config, N_ROWS)
    lum_x, lum_t = T.luminescence_table(config)
    return dict(templates=T.pmt_current_templates(config), spe=T.spe_scaling_table(resource.spe_charge, resource.spe_pdfs),
                gains=np.asarray(config['gains'], dtype=np.float64), thr_truth=thr_truth, thr_zle=thr_zle,
                lum_x=lum_x, lum_t=lum_t, noise=getattr(resource, 'noise_data', None))


def ap_tables_from_golden():
    d = golden('pmt_ap_tables.npz')
    out = {}
    for name in ['He', 'Xe', 'Uniform']:
        out[name] = dict(delaytime_cdf=d[f'{name}_delaytime_cdf'], amplitude_cdf=d[f'{name}_amplitude_cdf'],
                         delaytime_bin_size=float(d[f'{name}_delaytime_bin_size']),
                         amplitude_bin_size=float(d[f'{name}_amplitude_bin_size']))
    return out


def make_oracle(config, ap_tables=None, resource=None):
    from oracle.oracle import Oracle
    orc = Oracle(kernel_params(config), host_tables(config), ap_tables)
    orc.set_save_full_truth(config.get('save_full_truth', True))
    from wfsim_amd.delay_models import DelayModels
    orc.set_delay_models(DelayModels(config, resource or Resource(config)))
    return orc


def replay_chain_on_oracle(orc, d):
    """Feed the photons the reference generated (per Pulse.__call__) and digitise where the reference did."""
    first = d['dg_first_pulse']
    npl = d['dg_n_pulses']
    boundaries = set((first + npl).tolist())
    n_pulses = 0
    for k in range(len(d['call_kind'])):
        a, b = d['call_ph_off'][k], d['call_ph_off'][k + 1]
        orc.pulse_call(int(d['call_kind'][k]), k, d['ph_t'][a:b], d['ph_ch'][a:b], d['ph_dpe'][a:b], d['ph_gain'][a:b],
                       bool(d['call_has_gains'][k]))
        n_pulses += d['call_pulse_off'][k + 1] - d['call_pulse_off'][k]
        if n_pulses in boundaries:
            orc.digitize_and_zle(0)
    return orc.results()


# ------------------------------------------------------------------------------------------------ device side
def make_engine(config, seed=None, resource=None):
    from wfsim_amd.engine import Engine
    return Engine(config, resource or Resource(config), device=0, seed=seed, keep_photons=True)      # (photons(): tests compare photon by photon)


def chain_union(d):
    """primaries + recorded secondaries of a golden chain and, per instruction, its parent (-1 for primaries): the
    reference copies the parent S2's fields into its secondaries (afterpulse.py:49), event_number identifies it"""
    prim = d['instructions']
    if 'secondaries' not in d.files or len(d['secondaries']) == 0:
        return prim, np.full(len(prim), -1, dtype=np.int64)
    sec = d['secondaries']
    parent = np.full(len(prim) + len(sec), -1, dtype=np.int64)
    for k in range(len(sec)):
        parent[len(prim) + k] = int(np.where((prim['type'] == 2) & (prim['event_number'] == sec['event_number'][k]))[0][0])
    return np.concatenate([prim, sec]), parent


def chain_sets(d, config):
    """Pulse sets of a golden chain: (set_cluster, set_tmin) per recorded Pulse.__call__ -- the host scheduler's clusters,
    window-rule keys and run sets for the chain's instructions, matched to the recorded calls in processing order."""
    from wfsim_amd.scheduler import feedback_schedule
    ins, parent = chain_union(d)
    order, key, cluster, rs = feedback_schedule(ins, parent, config)
    n_sets = int(rs.max()) + 1
    cl_of = np.array([cluster[np.where(rs == q)[0][0]] for q in range(n_sets)])
    tmin_of = np.array([key[rs == q].min() for q in range(n_sets)])
    set_cluster, set_tmin = [], []
    k = -1
    for kind in d['call_kind']:
        if kind != 3:          # a primary Pulse call; kind 3 = PMT afterpulses of the previous primary
            k += 1
        set_cluster.append(cl_of[k])
        set_tmin.append(tmin_of[k])
    assert k == n_sets - 1, (k, n_sets)
    return np.asarray(set_cluster, np.int32), np.asarray(set_tmin, np.int64)


def replay_chain_on_engine(eng, d, config, debug=True, force_dense=False):
    set_cluster, set_tmin = chain_sets(d, config)
    eng.set_debug(debug, force_dense)
    dpe = np.array(d['ph_dpe'], dtype=np.uint8)
    for k in np.where(d['call_has_gains'])[0]:        # pre-assigned gains: n_double_pe = 0 (pulse.py:105-106)
        dpe[d['call_ph_off'][k]:d['call_ph_off'][k + 1]] = 0
    eng.load_photons(set_cluster, set_tmin, d['call_ph_off'], d['ph_t'], d['ph_ch'], d['ph_gain'], dpe)
    return eng.run()


def canonical_intervals(group, ch, left, right, data_off, data):
    """list of (group, ch, left, right, bytes) sorted -- order independent comparison"""
    out = []
    for k in range(len(ch)):
        out.append((int(group[k]), int(ch[k]), int(left[k]), int(right[k]),
                    np.asarray(data[data_off[k]:data_off[k + 1]], dtype=np.int64).tobytes()))
    return sorted(out)


def host_diffuse_patterns(orc, pattern_map, xy, gids, amps, p_survive, sigma_r, sigma_a, tpc_radius, em_base=0):
    """S2.s2_pattern_map_diffuse (s2.py:583-611) on the host with the electrons the oracle / the device draw: per instruction
    (averaged pattern of the surviving electrons inside the TPC, number of them, per-channel standard error of that average)"""
    out, n_in, err = [], [], []
    for i in range(len(xy)):
        sv, z0, z1 = orc.sample_diffusion(int(gids[i]), em_base, int(amps[i]), float(p_survive[i]))
        th = np.arctan2(xy[i, 1], xy[i, 0])
        hr, ha = z0[sv] * sigma_r[i], z1[sv] * sigma_a[i]
        pos = np.array([xy[i, 0] + np.cos(th) * hr - np.sin(th) * ha, xy[i, 1] + np.sin(th) * hr + np.cos(th) * ha]).T
        pos = pos[np.sum(pos ** 2, axis=1) <= tpc_radius ** 2]
        p = np.asarray(pattern_map(pos), dtype=np.float64) if len(pos) else np.zeros((0, 1))
        out.append(p.mean(axis=0) if len(pos) else None)
        n_in.append(len(pos))
        err.append(p.std(axis=0) / np.sqrt(max(len(pos), 1)) if len(pos) else None)
    return out, n_in, err
