"""SURVEY 8f.4 pinned on the reference: the host front end of the optical / nVeto path -- read_optical + _read_optical_nveto
(/root/reference/wfsim/strax_interface.py:234-333) and RawRecordsFromMcChain.set_timing (:824-863) -- against
tests/golden/optical_frontend.npz, made by tests/golden/make_golden.py `frontend` by RUNNING those reference functions on an
array-backed stand-in for the Geant4 tree.  The reference draws from numpy's seeded legacy generator; a RandomState with the
same seed hands this package's restatement the same uniforms, so everything compares exactly.  CPU only."""
import numpy as np
import pytest

from tests.helpers import golden
from wfsim_amd import synchronise_timing
from wfsim_amd.optical import read_optical_events

CH_LO, CH_HI = 2000, 2119


def _events(d, tag, with_energy):
    def ragged(k):
        flat, n = d[f'{tag}_{k}_flat'], d[f'{tag}_{k}_len']
        return [flat[a:b] for a, b in zip(np.cumsum(n) - n, np.cumsum(n))]
    ev = dict(eventid=d[f'{tag}_eventid'], pmthitID=ragged('pmthitID'), pmthitTime=ragged('pmthitTime'),
              xp_pri=d[f'{tag}_xp_pri'], yp_pri=d[f'{tag}_yp_pri'], zp_pri=d[f'{tag}_zp_pri'])
    if with_energy:
        ev['pmthitEnergy'] = ragged('pmthitEnergy')
    return ev


@pytest.mark.parametrize('tag,detector,use_qe', [('nveto', 'XENONnT_neutron_veto', True), ('nveto_noqe', 'XENONnT_neutron_veto', False),
                                                 ('tpc', 'XENONnT', None)])
def test_read_optical_equals_the_reference(tag, detector, use_qe):
    d = golden('optical_frontend.npz')
    qe = None
    if use_qe:
        qe = dict(nv_pmt_qe_wavelength=d['qe_wavelength'].tolist(),
                  nv_pmt_qe={str(c): d['qe_table'][c - CH_LO].tolist() for c in range(CH_LO, CH_HI + 1)})
    e0, e1 = (int(x) for x in d[f'{tag}_entry'])
    cfg = dict(detector=detector, channel_map=dict(nveto=(CH_LO, CH_HI), tpc=(0, 493)), entry_start=e0,
               entry_stop=None if tag == 'nveto' else e1, nv_pmt_ce_factor=float(d[f'{tag}_ce']))
    ins, channels, timings = read_optical_events(cfg, _events(d, tag, detector != 'XENONnT'), qe_data=qe,
                                                 rng=np.random.RandomState(int(d[f'{tag}_seed'])))
    ref = d[f'{tag}_ins']
    assert cfg['entry_stop'] == e1
    assert ins.dtype == ref.dtype and len(ins) == len(ref) > 20
    for f in ref.dtype.names:
        assert np.array_equal(ins[f], ref[f]), f
    assert np.array_equal(channels, d[f'{tag}_channels']) and np.array_equal(timings, d[f'{tag}_timings'])
    if detector != 'XENONnT':
        n_in = int(d[f'{tag}_pmthitID_len'].sum())
        assert 0 < len(channels) < n_in and channels.min() >= 0 and channels.max() <= CH_HI - CH_LO      # thinned, 0-based
    assert len(ins) > len(np.unique(ins['g4id']))          # optical_adjustment appended the split pulses


@pytest.mark.parametrize('tag', ['both', 'nv_only', 'tpc_only'])
def test_set_timing_equals_the_reference(tag):
    d = golden('optical_frontend.npz')
    e_in = [None if x < 0 else int(x) for x in d[f'timing_{tag}_entry_in']]
    cfg = dict(event_rate=1000.0, entry_start=e_in[0] if e_in[0] is not None else 0, entry_stop=e_in[1])
    tpc = d[f'timing_{tag}_epix_in'] if f'timing_{tag}_epix_in' in d.files else None
    nv = d[f'timing_{tag}_nveto_in'] if f'timing_{tag}_nveto_in' in d.files else None
    a, b, t = synchronise_timing(cfg, tpc, nv, rng=np.random.RandomState(int(d[f'timing_{tag}_seed'])))
    assert [cfg['entry_start'], cfg['entry_stop']] == d[f'timing_{tag}_entry_out'].tolist()
    for got, key in ((a, 'epix_out'), (b, 'nveto_out')):
        if f'timing_{tag}_{key}' not in d.files:
            assert got is None
            continue
        ref = d[f'timing_{tag}_{key}']
        assert got.dtype == ref.dtype and len(got) == len(ref) > 10
        for f in ref.dtype.names:
            assert np.array_equal(got[f], ref[f]), f
    if tpc is not None:
        assert len(a) < len(tpc)                            # the instructions behind the last event slot were removed
