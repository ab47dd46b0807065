"""Host preparation of optical (photon-list) input before RawDataOptical.

Reference: /root/reference/wfsim/utils.py:61-165 (``find_optical_t_range``, ``split_long_optical_pulse``,
``optical_adjustment``), called by ``read_optical`` (strax_interface.py:332) on what it read from the Geant4 file.
``read_optical`` itself needs ``uproot`` and a ROOT file and stays outside; this is the array arithmetic behind it.
"""
import numpy as np

PULSE_MAX_DURATION = int(1e3)        # utils.py:9
N_SPLIT_LOOP = 5                     # utils.py:10


def optical_adjustment(instructions, timings, channels):
    """1) every entry's ``time`` moves to its first photon and its photon timings become relative to it;
    2) an entry spanning more than PULSE_MAX_DURATION ns is split: the photons later than that are moved to the front of
    the entry's range and handed to a NEW instruction appended at the end, the entry keeps the early ones.

    As in the reference (utils.py:133-162) the loop ends after the first split round, because the rows it appends are
    never looked at again: their photon timings stay relative to the original entry's first photon (> PULSE_MAX_DURATION)
    and they keep the original entry's ``time``.  ``timings`` and ``channels`` are modified in place; the (longer)
    instruction array is returned."""
    instructions = np.array(instructions)
    first, last = instructions['_first'].astype(np.int64), instructions['_last'].astype(np.int64)
    n = len(instructions)
    tmins, tmaxs = np.full(n, -1, dtype=np.int64), np.full(n, -1, dtype=np.int64)
    for ix in range(n):                                   # utils.py:62-87
        a, b = first[ix], last[ix]
        if a == b:
            continue
        seg = timings[a:b]
        tmins[ix], tmaxs[ix] = seg.min(), seg.max()
        seg -= tmins[ix]
    instructions['time'] += tmins
    long_pulse = (tmaxs - tmins) > PULSE_MAX_DURATION
    extra = []
    for ix in np.where(long_pulse)[0]:                    # utils.py:90-119
        a, b = first[ix], last[ix]
        late = [iy for iy in range(a, b) if timings[iy] > PULSE_MAX_DURATION]
        if not late:
            continue
        cnt = a
        for k, iy in enumerate(late):
            cnt = a + k
            if iy > cnt:
                timings[cnt], timings[iy] = timings[iy], timings[cnt]
                channels[cnt], channels[iy] = channels[iy], channels[cnt]
        tmp = instructions[ix].copy()
        tmp['_first'], tmp['_last'] = a, cnt + 1
        instructions[ix]['_first'] = cnt + 1
        extra.append(tmp)
    if extra:
        instructions = np.append(instructions, np.array(extra, dtype=instructions.dtype))
    return instructions


def nveto_qe_table(qe_data, nveto_channels):
    """(channel, wavelength in nm) -> quantum efficiency in percent, interpolated to every nm below 1000
    (strax_interface.py:251-266); ``qe_data``: dict(nv_pmt_qe_wavelength=[...], nv_pmt_qe={str(channel): [...]}) or None
    (all 100 %)"""
    if qe_data is None:
        return np.ones([len(nveto_channels), 1000]) * 100
    from scipy.interpolate import interp1d
    out = np.zeros([len(nveto_channels), 1000])
    for ich, channel in enumerate(nveto_channels):
        out[ich] = interp1d(qe_data['nv_pmt_qe_wavelength'], qe_data['nv_pmt_qe'][str(channel)], bounds_error=False,
                            kind='linear', fill_value=(0, 0))(np.arange(1000))
    return out


def read_optical_events(config, events, qe_data=None, rng=None):
    """``read_optical`` (strax_interface.py:282-333, 235-279) behind the ROOT reader: ``events`` holds what the reference
    takes from the Geant4 tree -- ``eventid`` [n], per-event arrays ``pmthitID``, ``pmthitTime`` (s), ``pmthitEnergy``
    (eV, nVeto only), and ``xp_pri`` / ``yp_pri`` / ``zp_pri`` (mm).  Selects the entries, applies the nVeto quantum /
    collection efficiency thinning (a uniform draw per photon, here from ``rng``), builds the optical instructions and
    runs ``optical_adjustment``.  Returns (instructions, channels, timings)."""
    from .dtypes import instruction_dtype, optical_extra_dtype
    g4id = np.asarray(events['eventid'])
    if config.get('entry_stop', None) is None:
        config['entry_stop'] = int(np.max(g4id)) + 1
    mask = (g4id < config.get('entry_stop', int(2 ** 63 - 1))) & (g4id >= config.get('entry_start', 0))
    sel = np.where(mask)[0]
    n_events = len(sel)
    hit_id = [np.asarray(events['pmthitID'][i]) for i in sel]
    channels = np.hstack(hit_id).astype(np.int64) if n_events else np.zeros(0, np.int64)
    timings = np.hstack([np.asarray(events['pmthitTime'][i]) * 1e9 for i in sel]).astype(np.int64) if n_events else np.zeros(0, np.int64)
    lengths = np.array([len(h) for h in hit_id], dtype=np.int64)
    if config['detector'] == 'XENONnT_neutron_veto':
        lo, hi = config['channel_map']['nveto']
        nveto_channels = np.arange(lo, hi + 1)
        wavelengths = np.hstack([1239.841984 / np.asarray(events['pmthitEnergy'][i], dtype=np.float64) for i in sel])   # h * c / E, nm
        qe = nveto_qe_table(qe_data, nveto_channels)
        hit_mask = (channels >= nveto_channels[0]) & (channels <= nveto_channels[-1])
        channels[~hit_mask] = nveto_channels[0]
        wavelengths[(wavelengths < 0) | (wavelengths >= 999)] = 0
        qes = qe[channels - nveto_channels[0], np.around(wavelengths).astype(np.int64)]
        rng = rng or np.random.default_rng(int(config.get('seed', 0) or 0))
        hit_mask &= rng.random(len(qes)) <= qes * config.get('nv_pmt_ce_factor', 1.0) / 100
        ends = np.cumsum(lengths)
        csum = np.concatenate([[0], np.cumsum(hit_mask)])
        amplitudes = csum[ends] - csum[ends - lengths]
        channels, timings = channels[hit_mask] - lo, timings[hit_mask]         # shifted to 0-based for the simulation
    else:
        amplitudes = lengths
    ins = np.zeros(n_events, dtype=instruction_dtype + optical_extra_dtype)
    for f, k in (('x', 'xp_pri'), ('y', 'yp_pri'), ('z', 'zp_pri')):
        ins[f] = np.asarray(events[k]).flatten()[mask] / 10.
    ins['event_number'] = np.arange(n_events)
    ins['g4id'] = g4id[mask]
    ins['type'] = 1
    ins['recoil'] = 1
    ins['_first'] = np.cumsum(amplitudes) - amplitudes
    ins['_last'] = np.cumsum(amplitudes)
    ins = optical_adjustment(ins, timings, channels)
    return ins, channels, timings


def read_optical(config):
    """strax_interface.py:282-333: the Geant4 optical file named by ``config['fax_file']`` -> (instructions, channels,
    timings).  Needs ``uproot`` for the ROOT file; everything behind the reader is ``read_optical_events``."""
    try:
        import uproot
    except ImportError as e:
        raise NotImplementedError('read_optical needs uproot to open the Geant4 file; pass the event arrays to read_optical_events') from e
    events = uproot.open(config['fax_file']).get('events')
    keys = ['eventid', 'pmthitID', 'pmthitTime', 'xp_pri', 'yp_pri', 'zp_pri'] + (['pmthitEnergy'] if config['detector'] == 'XENONnT_neutron_veto' else [])
    return read_optical_events(config, {k: events[k].array(library='np') for k in keys}, qe_data=config.get('nv_pmt_qe_data'))
