"""Fax-config handling for the hot path (host side, Python like the reference).

* ``load_fax_config``  -- reads a fax JSON that may carry ``//`` / ``#`` comments and trailing commas, the way
  the bundled ``files/XENONnT_wfsim_config.json`` does (the reference goes through ``straxen.get_resource``,
  /root/reference/wfsim/strax_interface.py:567).
* ``xenonnt_test_config`` -- the reference's bundled test config plus the keys its plugin injects at run time
  (``gains``, ``channel_map``, ``n_tpc_pmts`` ... /root/reference/wfsim/strax_interface.py:566-608), with dummy maps.
* ``kernel_params`` -- flattens the config keys the hot path consumes into the scalars
  of the C-ABI ``wfs_config`` struct.
"""
import json
import os
import re

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')
N_ROWS = 801      # rows of the digitiser array, /root/reference/wfsim/core/rawdata.py:224


def load_fax_config(path):
    lines = []
    with open(path) as f:
        for line in f:
            if line.lstrip().startswith('//'):
                continue
            line = re.sub(r'#.*$', '', line)
            line = re.sub(r'(?<!:)//.*$', '', line)     # keep the // of URLs
            lines.append(line)
    txt = re.sub(r',\s*([\]}])', r'\1', ''.join(lines))
    return json.loads(txt)


def xenonnt_test_config(**overrides):
    """Bundled XENONnT test configuration (dummy maps, 494 PMTs, gains 2e6)."""
    with open(os.path.join(DATA_DIR, 'xenonnt_test_config.json')) as f:
        c = json.load(f)
    c['gains'] = np.asarray(c['gains'], dtype=np.float64)
    c['channels_bottom'] = np.asarray(c['channels_bottom'], dtype=np.int64)
    c['channel_map'] = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c['channel_map'].items()}
    spe = np.load(os.path.join(DATA_DIR, 'spe_single_channel.npz'))
    c['photon_area_distribution'] = dict(charge=spe['charge'], pdf=spe['pdf'], n_channels=c['n_tpc_pmts'])
    c.update(overrides)
    return c


def current_2_adc(config):
    """/root/reference/wfsim/core/pulse.py:33-35"""
    return (config['pmt_circuit_load_resistor'] * config['external_amplification']
            / (config['digitizer_voltage_range'] / 2 ** (config['digitizer_bits'])))


def afterpulse_switches(config):
    """The three afterpulse switches with the defaults RawData.sim_data gives them (rawdata.py:176, 194, 198): PMT and
    photo-ionisation afterpulses ON, gate afterpulses OFF when a key is missing.  One place for every user (RawData,
    kernel_params, the shard planner), so that they cannot disagree."""
    return dict(pmt=bool(config.get('enable_pmt_afterpulses', True)),
                electron=bool(config.get('enable_electron_afterpulses', True)),
                gate=bool(config.get('enable_gate_afterpulses', False)))


def tile_local_generation(config):
    """Whether primary S2s may draw their photons tile by tile inside the pulse workgroup (RNG spec v9; switch:
    config['tile_local_generation'], default on).  Off wherever something needs the photons of an instruction electron by electron:
    the transverse-diffusion field maps (the pattern of an instruction is then known only after its electrons) and any digitiser geometry
    other than 10 ns samples / 22-sample templates (the fused kernel is specialised for it).  Pulse calls that cover several instructions
    (save_full_truth=False, the shared calls of electron afterpulses) do not switch it off: an instruction that shares its call keeps the
    per-electron generator, one that is alone in its call takes the tile path (run sets numbered by their first instruction,
    engine.load_instructions; k_fuse_decide / the oracle's callers).  Electron / gate afterpulses (the reference's defaults: on / off, rawdata.py:194-200) do NOT
    switch it off: their pre-pass needs the photon number of every parent S2 and the arrival times of a few picked photons, which the
    tile path serves from the tiles' photon numbers and the photons' own Philox coordinates (wfs_copy_instruction_photon_offsets,
    wfs_gather_photon_times) without generating a photon.  The per-instruction conditions (s2_gain_spread == 0, default delay table)
    are checked where the instructions are: wfs_tilegen.h fuse_eligible and the oracle's twin."""
    transverse_maps = (config.get('diffusion_constant_transverse', 0) > 0
                       and config.get('enable_field_dependencies', {}).get('diffusion_transverse_map', False))
    tpc_digitiser = (int(config.get('sample_duration', 10)) == 10
                     and int(config.get('samples_before_pulse_center', 2)) + int(config.get('samples_after_pulse_center', 20)) == 22)
    return bool(config.get('tile_local_generation', True) and not transverse_maps and tpc_digitiser)


def kernel_params(config):
    """Scalars of the hot path, named as the fields of ``wfs_config`` (include/wfsim_amd.h)."""
    c = config
    detector_nt = c['detector'] == 'XENONnT'
    s1_model = c.get('s1_model_type', 'simple')
    for part in re.split(r'[+ ,]+', s1_model):
        # /root/reference/wfsim/core/s1.py:50 lists simple, custom, optical_propagation, nest
        assert part in ('', 'simple', 'custom', 'optical_propagation', 'nest'), f'Model type "{part}" not in the valid S1 model types'
        if part == 'nest':
            raise NotImplementedError('s1_model_type "nest" needs nestpy and is outside the MI355X hot path (SURVEY.md 2.1 row 2)')
    if c.get('s2_luminescence_model', 'simple') not in ('simple', 'garfield', 'garfield_gas_gap'):
        raise KeyError(f"{c['s2_luminescence_model']} is not valid! Use 'simple' or 'garfield' or 'garfield_gas_gap'")
    # same substring tests, in the same order, as s2.py:539-552; the propagation term itself is a delay table (delay_models.py)
    if 'optical_propagation' in c['s2_time_model'] or 'zero_delay' in c['s2_time_model']:
        s2_time_model = 0
    elif 's2_time_spread around zero' in c['s2_time_model']:
        s2_time_model = 1
    else:
        raise KeyError(f"{c['s2_time_model']} is not in any of the valid s2 time models")
    he = c.get('channel_map', {}).get('he', (500, 752))
    bottom = np.asarray(c.get('channels_bottom', []))
    return dict(
        dt=int(c.get('sample_duration', 10)),
        samples_before=int(c.get('samples_before_pulse_center', 2)),
        samples_after=int(c.get('samples_after_pulse_center', 20)),
        store_before=int(c['samples_to_store_before']),
        store_after=int(c['samples_to_store_after']),
        tlen=int(c.get('samples_before_pulse_center', 2)) + int(c.get('samples_after_pulse_center', 20)),
        trigger_window=int(c['trigger_window']),
        baseline=int(c['digitizer_reference_baseline']),
        n_rows=N_ROWS,
        n_tpc=len(c['gains']),
        n_top=int(c.get('n_top_pmts', 0)),
        he_first=int(he[0]),
        he_factor=int(c.get('high_energy_deamplification_factor', 0)),     # int(), rawdata.py:242
        sum_channel=int(c.get('channel_map', {}).get('sum_signal', 800)),
        last_bottom=int(bottom[-1]) if len(bottom) else -1,
        detector_nt=int(detector_nt),
        enable_noise=int(bool(c.get('enable_noise', True))),
        s1_simple=int('simple' in s1_model),
        s2_time_model=s2_time_model,
        enable_pmt_ap=int(afterpulse_switches(c)['pmt']),
        tile_gen=int(tile_local_generation(c)), tile_gen_min=int(c.get('tile_local_min_photons', 64)),
        fma=int(bool(c.get('fused_multiply_add', True))), row_resident=(2 if c.get('row_resident', 'auto') == 'auto' else int(bool(c.get('row_resident')))),
        c2a=float(current_2_adc(c)),
        tts_mean=float(c['pmt_transit_time_mean']),
        tts_sigma=float(c['pmt_transit_time_spread'] / 2.35482),          # pulse.py:52-56
        p_dpe=float(c['p_double_pe_emision']),
        s1_decay_time=float(c.get('s1_decay_time', 0.0)),
        s1_decay_spread=float(c.get('s1_decay_spread', 0.0)),
        sf_gas=float(c['singlet_fraction_gas']),
        t1_gas=float(c['singlet_lifetime_gas']),
        t3_gas=float(c['triplet_lifetime_gas']),
        s2_time_spread=float(c.get('s2_time_spread', 0.0)),
        trap_time=float(c['electron_trapping_time']),
        gain_spread=float(c.get('s2_gain_spread', 0)),
        pmt_ap_modifier=float(c.get('pmt_ap_modifier', 1)),
        pmt_ap_t_modifier=float(c.get('pmt_ap_t_modifier', 0)),
        rext=float(c.get('right_raw_extension', 100000)),
        drift_velocity=float(c['drift_velocity_liquid']),
        seed=int(c.get('seed', 0) or 0),
    )
