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
