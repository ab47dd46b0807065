"""Transverse diffusion with field maps (S2.s2_pattern_map_diffuse, s2.py:560-613): the host restatement that the GPU rows are
compared with (tests/helpers.host_diffuse_patterns, fed with the oracle's electron draws) against patterns the REFERENCE
produced for the same inputs (tests/golden/diffusion.npz, 4 instructions x 20000 electrons; make_golden.fixture_diffusion)."""
import numpy as np

from tests.golden.make_golden import diffusion_case
from tests.helpers import golden, make_oracle, host_diffuse_patterns
from wfsim_amd.config import xenonnt_test_config
from wfsim_amd.itp_map import InterpolatingMap
from wfsim_amd.physics import s2_transverse_sigmas


class _FieldMaps:
    drift_velocity_scaling = 1.0

    def __init__(self, par):
        self.par = par

    def field_dependencies_map(self, z, xy, map_name='map'):
        return np.full(len(z), dict(diffusion_radial_map=self.par['d_r'], diffusion_azimuthal_map=self.par['d_a'])[map_name])


def test_averaged_patterns_agree_with_the_reference():
    d = golden('diffusion.npz')
    pattern, xy, z, par = diffusion_case()
    cfg = xenonnt_test_config(seed=17, tpc_radius=par['tpc_radius'], drift_velocity_liquid=par['v'], diffusion_constant_transverse=1.0,
                              enable_field_dependencies=dict(diffusion_transverse_map=True, drift_speed_map=False))
    sr, sa = s2_transverse_sigmas(z, xy, cfg, _FieldMaps(par))
    assert np.allclose(sr, np.sqrt(2 * par['d_r'] * 1e-9 * -z / par['v'])) and sr.max() > 1.5 and np.all(sa < sr)
    ne = d['n_electron']
    orc = make_oracle(cfg)
    pm = InterpolatingMap(pattern)
    mine, n_in, err = host_diffuse_patterns(orc, pm, xy, np.arange(len(z)), ne, np.ones(len(z)), sr, sa, par['tpc_radius'])
    assert n_in[0] == ne[0] and n_in[3] < 0.8 * ne[3]                # the instruction at the edge loses electrons to the cut
    plain = pm(xy)
    for i in range(len(z)):
        ref = d['patterns'][i]
        # two independent averages over 20000 electrons: channel by channel within 6 standard errors of the difference
        assert np.all(np.abs(mine[i] - ref) <= 6 * np.sqrt(2) * err[i] + 1e-9), i
        assert abs(mine[i].sum() / ref.sum() - 1) < 6 * np.sqrt(2) * np.sqrt(np.sum(err[i] ** 2)) / ref.sum() + 1e-3
    # and the diffusion is visible: the averaged pattern is not the pattern at the instruction's position
    assert np.abs(d['patterns'][3] - plain[3]).max() > 20 * err[3].max()
