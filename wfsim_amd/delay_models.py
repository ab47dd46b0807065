"""Model variants of the photon delays: the extra terms of S1.photon_timings and S2.photon_timings as tables.

The reference adds, per photon, independent delay terms that are each truncated to an integer before they are added
(SURVEY.md appendix B.2).  The scalars of ``wfs_config`` describe the terms of the default models (S1 'simple',
S2 'simple' luminescence + singlet/triplet + time spread, PMT transit time).  Every other model the reference offers
adds ONE more such term, described here by its probability mass function on the integers; the device convolves it with
the default terms and samples the sum from one uniform (include/wfsim_amd.h: wfs_set_delay_models):

* S1 ``custom`` (/root/reference/wfsim/core/s1.py:197-214): per recoil class ER / NR / alpha / LED (s1.py:262-337)
* S2 ``garfield`` luminescence (/root/reference/wfsim/core/s2.py:380-411): per wire-distance row of the garfield table
* S2 ``optical_propagation`` (s2.py:486-502, 540-541): per PMT array (top / bottom)
* S1 ``optical_propagation`` (s1.py:185-188, 241-260) depends on z continuously: evaluated per photon on the device from
  the spline nodes (wfs_set_s1_propagation), not a table.

* S2 ``garfield_gas_gap`` luminescence (s2.py:413-483) is NOT an independent term -- the mean of the instruction's own draws is
  subtracted (s2.py:447-448): the device draws it per photon from the interpolated excitation-time tables and subtracts the
  instruction's mean (wfs_set_gas_gap_model / wfs_set_instruction_gas_gap); the remaining terms come from a table.

``nest`` (needs nestpy) stays outside the path.
"""
import numpy as np

# wfsim/units.py: distances in cm, times in ns, energies in eV, charge in electrons
_ELECTRON_CHARGE_SI = 1.602176565 * 10 ** (-19)
_J = 1 / _ELECTRON_CHARGE_SI
_KG = _J * (10 ** 9) ** 2 / (10 ** 2) ** 2
G_PER_CM3 = (10 ** (-3) * _KG) / (10 ** (-2) * 10 ** 2) ** 3
V_PER_CM = 1.0          # V = J / C = 1, cm = 1

RECOIL_CLASSES = {'NR': [0], 'ALPHA': [6], 'ER': [7, 8, 11, 12], 'LED': [20]}          # NestId, s1.py:21-30


# ----------------------------------------------------------------------------------------------- pmf algebra
class Pmf:
    """probabilities of the integers vmin, vmin + 1, ..."""

    def __init__(self, p, vmin=0):
        self.p = np.asarray(p, dtype=np.float64)
        self.vmin = int(vmin)

    def conv(self, other):
        return Pmf(np.convolve(self.p, other.p), self.vmin + other.vmin)

    @staticmethod
    def mix(parts):
        lo = min(q.vmin for _, q in parts)
        hi = max(q.vmin + len(q.p) for _, q in parts)
        p = np.zeros(hi - lo)
        for w, q in parts:
            p[q.vmin - lo:q.vmin - lo + len(q.p)] += w * q.p
        return Pmf(p, lo)

    def cdf_at(self, k):
        c = np.cumsum(self.p)
        i = np.clip(np.asarray(k) - self.vmin, -1, len(c) - 1)
        return np.where(i < 0, 0.0, c[np.maximum(i, 0)])


def pmf_delta(v=0):
    return Pmf([1.0], v)


def pmf_exp(tau):
    """trunc(Exp(1) * tau): P(X <= k) = 1 - exp(-(k + 1) / tau)   (pulse.py:341, s1.py:193)"""
    if not tau > 0:
        return pmf_delta(0)
    k = np.arange(0, int(np.ceil(40 * tau)) + 2)
    cum = -np.expm1(-(k + 1) / tau)
    cum[-1] = 1.0
    return Pmf(np.diff(cum, prepend=0.0), 0)


def pmf_singlet_triplet(singlet_fraction, t1, t3):
    """Pulse.singlet_triplet_delays, /root/reference/wfsim/core/pulse.py:320-341"""
    return Pmf.mix([(singlet_fraction, pmf_exp(t1)), (1 - singlet_fraction, pmf_exp(t3))])


def pmf_uniform(length):
    """trunc(Uniform(0, L)), s1.py:272-279 (LED)"""
    if not length > 0:
        return pmf_delta(0)
    k = np.arange(0, int(np.ceil(length)))
    return Pmf((np.minimum(k + 1, length) - k) / length, 0)


def pmf_recombination(reco_time, maximum=1000):
    """trunc(clip(reco_time / (1 / U - 1), 0, maximum)), s1.py:318-322: P(R <= y) = y / (reco_time + y) below the clip"""
    k = np.arange(0, int(maximum) + 1)
    cum = (k + 1) / (reco_time + k + 1.0)
    cum[-1] = 1.0                       # everything above the clip lands on `maximum`
    return Pmf(np.diff(cum, prepend=0.0), 0)


def pmf_piecewise_linear(u, t):
    """trunc(f(U)), U uniform on [u[0], u[-1]], f piecewise linear through (u, t), not necessarily monotone.

    trunc is the C cast (toward zero): value k > 0 for f in [k, k + 1), 0 for f in (-1, 1), k < 0 for f in (k - 1, k]."""
    u, t = np.asarray(u, dtype=np.float64), np.asarray(t, dtype=np.float64)
    lo, hi = int(np.floor(t.min())) - 1, int(np.ceil(t.max())) + 1
    edges = np.arange(lo, hi + 2, dtype=np.float64)
    F_le, F_lt = np.zeros(len(edges)), np.zeros(len(edges))      # P(f(U) <= y), P(f(U) < y) at the integers y
    span = u[-1] - u[0]
    for a, b, du in zip(t[:-1], t[1:], np.diff(u)):
        w = du / span
        if a == b:                                               # a flat piece is an atom
            F_le += w * (edges >= a)
            F_lt += w * (edges > a)
        else:
            m, M = min(a, b), max(a, b)
            frac = w * np.clip((edges - m) / (M - m), 0.0, 1.0)
            F_le += frac
            F_lt += frac
    ks = np.arange(lo, hi + 1)
    # trunc(f) <= k  <=>  f < k + 1 for k >= 0,  f <= k for k < 0
    cum = np.where(ks >= 0, F_lt[(ks + 1 - lo).clip(0, len(edges) - 1)], F_le[(ks - lo).clip(0, len(edges) - 1)])
    cum[-1] = 1.0
    return Pmf(np.diff(cum, prepend=0.0), lo)


def pmf_interp(x, t):
    """trunc(np.interp(U, x, t)) for increasing x and t (the luminescence quantile table, s2.py:338): P(L <= y) is the inverse
    of the piecewise linear map, evaluated at the integers; u below x[0] gives t[0] (np.interp clamps)"""
    x, t = np.asarray(x, dtype=np.float64), np.asarray(t, dtype=np.float64)
    lo, hi = int(np.floor(t[0])) - 1, int(np.ceil(t[-1])) + 1
    ks = np.arange(lo, hi + 1)
    y = np.where(ks >= 0, ks + 1, ks).astype(np.float64)          # trunc(L) <= k  <=>  L < k + 1 (k >= 0), L <= k (k < 0)
    cum = np.where(y < t[0], 0.0, np.where(y >= t[-1], 1.0, np.interp(y, t, x)))
    cum[-1] = 1.0
    return Pmf(np.diff(cum, prepend=0.0), lo)


def pmf_samples(values):
    """a value drawn uniformly from ``values`` (already integers)"""
    v = np.asarray(values, dtype=np.int64)
    lo = int(v.min())
    return Pmf(np.bincount(v - lo) / len(v), lo)


# ----------------------------------------------------------------------------------------------- the models
def s1_custom_pmf(kind, config):
    """delay term of S1.er / nr / alpha / led (s1.py:262-337) in the liquid phase"""
    c = config
    t1, t3 = c['singlet_lifetime_liquid'], c['triplet_lifetime_liquid']
    if kind == 'ALPHA':
        return pmf_singlet_triplet(c['s1_ER_alpha_singlet_fraction'], t1, t3)
    if kind == 'NR':
        return pmf_singlet_triplet(c['s1_NR_singlet_fraction'], t1, t3)
    if kind == 'LED':
        return pmf_uniform(c['led_pulse_length'])
    if kind == 'ER':
        density = c.get('liquid_density', 1.872452802978054e+30) / G_PER_CM3
        excfrac = 0.4 - 0.11131 * density - 0.0026651 * density ** 2
        excfrac = 1 / (1 + excfrac)
        excfrac /= 1 - (1 - excfrac) * (1 - c['s1_ER_recombination_fraction'])
        efield = c['drift_field'] / V_PER_CM
        reco_time = 3.5 / 0.18 * (1 / 20 + 0.41) * np.exp(-0.009 * efield)
        primary = pmf_singlet_triplet(c['s1_ER_primary_singlet_fraction'], t1, t3)
        # float recombination time + integer excimer delay, truncated once: trunc(R + I) = trunc(R) + I for R >= 0
        secondary = pmf_recombination(reco_time, 1000).conv(pmf_singlet_triplet(c['s1_ER_secondary_singlet_fraction'], t1, t3))
        return Pmf.mix([(excfrac, primary), (1 - excfrac, secondary)])
    raise ValueError(kind)


def spline_nodes_1d(spline, map_name, n=4097):
    """(u, t) nodes of a 1-D propagation spline: the grid of an InterpolatingMap (between two nodes the inverse-distance
    weighting of the two nearest points IS linear interpolation), else the callable sampled on n points of [0, 1]"""
    grid = getattr(spline, 'grid', None)
    if grid is not None and len(grid) == 1 and getattr(spline, 'method', '') in ('WeightedNearestNeighbors', 'RegularGridInterpolator'):
        u = grid[0]
        t = np.asarray(spline.data[map_name], dtype=np.float64).reshape(-1)
        if u[0] <= 0 and u[-1] >= 1:
            # restrict to [0, 1): the uniform never leaves it
            uu = np.unique(np.concatenate([[0.0], u[(u > 0) & (u < 1)], [1.0]]))
            return uu, np.interp(uu, u, t)
    u = np.linspace(0, 1, n)
    return u, np.asarray(spline(u[:, None], map_name=map_name), dtype=np.float64).reshape(-1)


class DelayModels:
    """Tables and per-instruction choices for one configuration; ``None``-like (``active == False``) for the defaults."""

    def __init__(self, config, resource):
        c = config
        self.config = c
        s1_model = c.get('s1_model_type', 'simple')
        s2_time_model = c.get('s2_time_model', '')
        lum_model = c.get('s2_luminescence_model', 'simple')
        self.base, self.pmfs = [], []
        self.s1_tables = {}                 # recoil class -> table
        self.s1_prop = None
        self.s2_rows = None                 # garfield: x grid of the rows
        self.gas_gap = None                 # garfield_gas_gap: tabulated gas gaps, inverse CDFs, the (x, y) -> gas gap map
        self.s2_tables = None               # [n_rows or 1][2] table index per row and array
        if 'custom' in s1_model:
            for kind in RECOIL_CLASSES:
                if kind == 'LED' and 'led_pulse_length' not in c:
                    continue
                self.s1_tables[kind] = self._add(1, s1_custom_pmf(kind, c))
        if 'optical_propagation' in s1_model:
            spline = resource.s1_optical_propagation_spline
            grid = getattr(spline, 'grid', None)
            if grid is not None and len(grid) == 2 and getattr(spline, 'method', '') == 'RegularGridInterpolator':
                zg, ug = grid
                top = np.asarray(spline.data['top'], dtype=np.float64).reshape(len(zg), len(ug))
                bot = np.asarray(spline.data['bottom'], dtype=np.float64).reshape(len(zg), len(ug))
            else:               # any callable: resampled on a fine regular grid, multilinear in between
                zg = np.linspace(-float(c['tpc_length']), 0.0, 257)
                ug = np.linspace(0.0, 1.0, 1025)
                pts = np.array(np.meshgrid(zg, ug, indexing='ij')).reshape(2, -1).T
                top = np.asarray(spline(pts, map_name='top'), dtype=np.float64).reshape(len(zg), len(ug))
                bot = np.asarray(spline(pts, map_name='bottom'), dtype=np.float64).reshape(len(zg), len(ug))
            self.s1_prop = dict(z=np.asarray(zg, dtype=np.float64), u0=float(ug[0]), du=float(ug[1] - ug[0]), nu=len(ug),
                                top=np.ascontiguousarray(top), bottom=np.ascontiguousarray(bot))
        s2_prop = None
        if 'optical_propagation' in s2_time_model:
            spline = resource.s2_optical_propagation_spline
            s2_prop = [pmf_piecewise_linear(*spline_nodes_1d(spline, name)) for name in ('top', 'bottom')]
        if lum_model == 'garfield':
            lum = resource.s2_luminescence
            t, x = np.asarray(lum['t']), np.asarray(lum['x'], dtype=np.float64)
            assert t.ndim == 2, 'Timing data is expected to have D2'
            avgt = int(np.average(t).astype(int))                                   # s2.py:410
            self.s2_rows = x
            self.s2_tables = []
            for r in range(len(x)):
                row = pmf_samples(t[r].astype(np.int64) - avgt)
                self.s2_tables.append([self._add(3, row.conv(s2_prop[k]) if s2_prop else row) for k in range(2 if s2_prop else 1)])
        elif lum_model == 'garfield_gas_gap':
            gg = resource.s2_luminescence_gg
            self.gas_gap = dict(gas_gap=np.asarray(gg['gas_gap'], dtype=np.float64), inv=np.ascontiguousarray(gg['timing_inv_cdf'], dtype=np.float64),
                                map=resource.garfield_gas_gap_map)
            assert self.gas_gap['inv'].ndim == 2 and len(self.gas_gap['gas_gap']) == len(self.gas_gap['inv']) >= 2
            # everything but the luminescence: singlet / triplet + spread + transit time (base 3) (+ optical propagation)
            self.s2_tables = [[self._add(3, s2_prop[k] if s2_prop else pmf_delta(0)) for k in range(2 if s2_prop else 1)]]
        elif lum_model != 'simple':
            raise NotImplementedError(f's2_luminescence_model "{lum_model}" is outside the MI355X hot path (delay_models.py)')
        elif s2_prop:
            self.s2_tables = [[self._add(2, s2_prop[0]), self._add(2, s2_prop[1])]]
        # gas gap warping (s2.py:360-378): the 'simple' luminescence depends on the gas gap under the instruction -> one table
        # per S2 instruction, rebuilt for every batch (a functional path: ~1 ms of host work per instruction)
        self.warp = bool(c.get('enable_gas_gap_warping', False)) and lum_model == 'simple'
        self._s2_prop = s2_prop
        self._gas_gap = getattr(resource, 'gas_gap_length', None)
        if self.warp:
            assert self._gas_gap is not None, 'enable_gas_gap_warping needs resource.gas_gap_length (config gas_gap_map)'
            self.s2_tables = None               # the static top / bottom pair is replaced by per-instruction tables
            self.base, self.pmfs = [b for b, q in zip(self.base, self.pmfs) if b != 2], [q for b, q in zip(self.base, self.pmfs) if b != 2]
        self._n_static = len(self.base)
        self.per_batch = self.warp
        self.active = bool(self.base) or self.s1_prop is not None or self.warp

    def _add(self, base, pmf):
        self.base.append(base)
        self.pmfs.append(pmf)
        return len(self.base) - 1

    # what wfs_set_delay_models / orc_set_delay_models take
    def table_arrays(self):
        off = np.concatenate([[0], np.cumsum([len(q.p) for q in self.pmfs])]).astype(np.int64)
        pmf = np.concatenate([q.p for q in self.pmfs]) if self.pmfs else np.zeros(0)
        return (np.asarray(self.base, dtype=np.int32), off, np.ascontiguousarray(pmf, dtype=np.float64),
                np.asarray([q.vmin for q in self.pmfs], dtype=np.int32))

    def garfield_rows(self, instructions, gids=None):
        """row of the garfield table for every instruction: nearest tabulated distance to a wire (s2.py:395-406)"""
        c = self.config
        xy = np.array([instructions['x'], instructions['y']], dtype=np.float64).T
        confine = c.get('s2_garfield_confine_position', 0.0)
        if isinstance(confine, float) and confine > 0.0:
            # a uniform draw per instruction (s2.py:396); its stream is keyed by the run-wide instruction id
            from numpy.random import Generator, Philox
            g = np.arange(len(xy)) if gids is None else np.asarray(gids)
            distance = np.array([Generator(Philox(key=[int(c.get('seed', 0) or 0), (int(q) << 8) | 0x47])).uniform(-confine, confine) for q in g])
        else:
            tilt = c.get('anode_xaxis_angle', np.pi / 4)
            pitch = c.get('anode_pitch', 0.5)
            rot = np.array(((np.cos(tilt), -np.sin(tilt)), (np.sin(tilt), np.cos(tilt))))
            distance = (np.matmul(xy, rot)[:, 1] + pitch / 2) % pitch - pitch / 2
        return np.array([np.argmin(np.abs(d - self.s2_rows)) for d in distance], dtype=np.int64)

    def instruction_gas_gap(self, instructions):
        """(table, weight) per instruction for wfs_set_instruction_gas_gap: s2.py:470-476 -- the tabulated gas gap at or below the
        one under the instruction (np.digitize - 1) and the distance to it in units of the spacing; -1 for S1s"""
        n = len(instructions)
        idx, w = np.full(n, -1, dtype=np.int32), np.zeros(n, dtype=np.float64)
        s2 = np.where(instructions['type'] != 1)[0]
        if len(s2):
            gaps = self.gas_gap['gas_gap']
            d_gas_gap = gaps[1] - gaps[0]
            xy = np.array([instructions['x'][s2], instructions['y'][s2]]).T
            cont = np.asarray(self.gas_gap['map'](xy), dtype=np.float64).reshape(len(s2), -1)[:, 0]
            draw = np.digitize(cont, gaps) - 1
            if np.any(draw < 0):
                # s2.py:476-477: index -1 wraps to the LAST table with the distance taken from gaps[-1] -- an artefact of numpy's
                # negative indexing, not physics; here such positions use the first table (distance < 0: extrapolated)
                import warnings
                warnings.warn(f'garfield_gas_gap: {int(np.sum(draw < 0))} instruction(s) with a gas gap below the first tabulated value '
                              f'({gaps[0]}): the first table is used where the reference wraps to the last one (s2.py:476)')
            draw = np.clip(draw, 0, len(gaps) - 1)
            idx[s2] = draw
            w[s2] = (cont - gaps[draw]) / d_gas_gap
        return idx, w

    def instruction_tables(self, instructions, gids=None):
        """(tab, tab_bottom, prop_zi, prop_zf) for wfs_set_instruction_models"""
        n = len(instructions)
        tab = np.full(n, -1, dtype=np.int32)
        tabb = np.full(n, -1, dtype=np.int32)
        zi = np.full(n, -1, dtype=np.int32)
        zf = np.zeros(n, dtype=np.float64)
        is_s1 = instructions['type'] == 1
        if self.s1_tables and is_s1.any():
            recoil = instructions['recoil']
            known = np.zeros(n, dtype=bool)
            for kind, ids in RECOIL_CLASSES.items():
                sel = is_s1 & np.isin(recoil, ids)
                if sel.any():
                    if kind not in self.s1_tables:
                        raise KeyError('led_pulse_length')
                    tab[sel] = tabb[sel] = self.s1_tables[kind]
                known |= sel
            if np.any(is_s1 & ~known):
                raise AttributeError(f'Recoil type must be ER, NR, alpha or LED, not {np.unique(recoil[is_s1 & ~known])}. Check nest ids')
        if self.s1_prop is not None and is_s1.any():
            zg = self.s1_prop['z']
            z = instructions['z'][is_s1].astype(np.float64)
            i = np.clip(np.searchsorted(zg, z) - 1, 0, len(zg) - 2)          # scipy RegularGridInterpolator._find_indices
            zi[is_s1] = i
            zf[is_s1] = (z - zg[i]) / (zg[i + 1] - zg[i])
        if self.warp and (~is_s1).any():
            from .tables import luminescence_table
            del self.base[self._n_static:], self.pmfs[self._n_static:]
            s2 = np.where(~is_s1)[0]
            xy = np.array([instructions['x'][s2], instructions['y'][s2]], dtype=np.float64).T
            gaps = np.asarray(self._gas_gap(xy), dtype=np.float64).reshape(len(s2), -1)[:, 0]
            if len(np.unique(gaps)) * (2 if self._s2_prop else 1) > 65536 - self._n_static:
                raise ValueError('enable_gas_gap_warping: more than 65536 delay tables in one batch; lower RawData.max_batch_quanta')
            made = {}
            for i, g in zip(s2, gaps):
                if g not in made:
                    lum = pmf_interp(*luminescence_table(self.config, gas_gap=float(g)))
                    made[g] = [self._add(3, lum.conv(p) if p is not None else lum) for p in (self._s2_prop or [None])]
                tab[i], tabb[i] = made[g][0], made[g][-1]
        if self.s2_tables is not None and (~is_s1).any():
            s2 = np.where(~is_s1)[0]
            rows = self.garfield_rows(instructions[s2], None if gids is None else np.asarray(gids)[s2]) if self.s2_rows is not None else np.zeros(len(s2), dtype=np.int64)
            t = np.asarray(self.s2_tables, dtype=np.int32)
            tab[s2] = t[rows, 0]
            tabb[s2] = t[rows, -1]
        return tab, tabb, zi, zf
