"""Electron afterpulses: secondary instructions made from a simulated S2 (host side).

Reference: /root/reference/wfsim/core/afterpulse.py:14-139 (``PhotoIonization_Electron``, ``PhotoElectric_Electron``) and
the scheduler feedback loop /root/reference/wfsim/core/rawdata.py:133-145, 192-202.  After every S2 pulse set the
reference draws a number of delayed electrons from the number of DETECTED photons of that set, picks random photons of
the set as time zeros and queues ``type = 4`` (photo-ionisation in the liquid) / ``type = 6`` (photo-electric effect on
the gate) instructions, which are later simulated by ``S2.__call__`` like any S2.

Here the secondaries of the whole run are made in a pre-pass (``RawData`` generates the photons of the primaries once,
without pulses), then primaries and secondaries go through the normal path together.  What the device contributes is
the photon count of every parent set and the arrival times of the chosen photons (``Engine.set_photon_counts`` /
``gather_photon_times``).  Random streams: numpy's Philox generator keyed by (seed, parent gid, kind) -- one stream
per parent, so the secondaries do not depend on batching or sharding; a secondary carries its parent's gid and the
emitter offset ``(k + 1) << 20`` for its own photon Monte Carlo (include/wfsim_amd.h, ``em_base``).
"""
import numpy as np

from .config import afterpulse_switches

MAX_SECONDARIES_PER_PARENT = 4094          # emitter offsets (k + 1) << 20 must fit 32 bits


class DelayHistogram:
    """The few members of ``multihist.Hist1d`` the reference touches on ``resource.uniform_to_ele_ap``
    (afterpulse.py:33-41, 61-73): ``n`` (sum of the bin contents = electrons per detected photon), ``bin_centers`` and
    ``get_random`` (bin by bin content, uniform inside the bin).  Wraps any object with ``histogram`` and ``bin_edges``."""

    def __init__(self, histogram, bin_edges):
        self.histogram = np.asarray(histogram, dtype=np.float64)
        self.bin_edges = np.asarray(bin_edges, dtype=np.float64)
        assert len(self.bin_edges) == len(self.histogram) + 1
        self.bin_centers = 0.5 * (self.bin_edges[1:] + self.bin_edges[:-1])
        self.n = float(self.histogram.sum())
        self._p = self.histogram / self.n if self.n > 0 else None

    @classmethod
    def wrap(cls, obj):
        return obj if isinstance(obj, cls) else cls(obj.histogram, obj.bin_edges)

    def get_random(self, rng, size):
        if size == 0 or self._p is None:
            return np.zeros(0)
        i = rng.choice(len(self.bin_centers), size=size, p=self._p)
        return self.bin_centers[i] + rng.uniform(-0.5, 0.5, size) * np.diff(self.bin_edges)[i]


def coarse_delay_grid(hist, config):
    """afterpulse.py:61-71: delay bins as wide as the longitudinal diffusion at that drift time, starting at 100 ns"""
    spread = np.sqrt(2 * config['diffusion_constant_longitudinal'] * hist.bin_centers) / config['drift_velocity_liquid']
    grid, t = [], 100
    while t < hist.bin_centers[-1]:
        grid.append(t)
        t += spread[np.argmin(np.abs(t - hist.bin_centers))]
    return np.array(grid)


def _rng(config, gid, kind):
    return np.random.Generator(np.random.Philox(key=int(config.get('seed', 0)) & (2 ** 64 - 1),
                                                counter=[int(gid), int(kind), 0xE1EC, 0]))


def _positions(rng, n, config):
    r = np.sqrt(rng.uniform(0, config['tpc_radius'] ** 2, n))                  # afterpulse.py:78-84
    a = rng.uniform(-np.pi, np.pi, n)
    return r * np.cos(a), r * np.sin(a)


def plan_secondaries(parent, gid, n_photons, config, hist=None, grid=None):
    """Everything about the secondaries of one parent set except their time zeros.  Returns a list of
    (kind, photon_index[n], delay[n], amp[n], x[n], y[n]) with kind 4 (photo-ionisation) / 6 (gate)."""
    out = []
    if n_photons <= 0:
        return out                                                               # afterpulse.py:24-26
    sw = afterpulse_switches(config)
    if sw['electron'] and hist is not None:
        rng = _rng(config, gid, 4)
        n_el = rng.poisson(hist.n * n_photons * config['photoionization_modifier'])          # afterpulse.py:37-39
        delay = hist.get_random(rng, n_el)
        delay = delay[delay < grid[-1]]
        idx, cnt = np.unique(np.digitize(delay, grid), return_counts=True)                   # afterpulse.py:73-76
        n = len(idx)
        pick = rng.integers(0, n_photons, n)
        x, y = _positions(rng, n, config)
        out.append((4, pick, grid[idx], cnt.astype(np.int64), x, y))
    if sw['gate']:
        rng = _rng(config, gid, 6)
        n = rng.poisson(config['photoelectric_p'] * n_photons * config['photoelectric_modifier'])   # afterpulse.py:108-110
        delay = np.clip(rng.normal(config['photoelectric_t_center'] + config['drift_time_gate'],
                                   config['photoelectric_t_spread'], n), 0, None)
        pick = rng.integers(0, n_photons, n)
        x, y = _positions(rng, n, config)
        out.append((6, pick, delay, np.ones(n, dtype=np.int64), x, y))
    return out


def build_instructions(parent, plans, t_zeros_per_plan, config):
    """afterpulse.py:49-58 / 123-131: copies of the parent's (first) instruction with type, time, position and amp set"""
    rows = []
    for (kind, pick, delay, amp, x, y), t0 in zip(plans, t_zeros_per_plan):
        ins = np.repeat(parent, len(pick))
        ins['type'] = kind
        ins['time'] = t0 - config['drift_time_gate'] if kind == 4 else t0 + config['drift_time_gate']
        ins['x'], ins['y'] = x, y
        ins['z'] = -delay * config['drift_velocity_liquid']
        ins['amp'] = amp
        rows.append(ins)
    return np.concatenate(rows) if rows else np.zeros(0, dtype=parent.dtype)
