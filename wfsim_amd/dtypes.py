"""Structured dtypes of the drop-in boundary.

Same field names, order, types and titles as the reference interface so arrays are interchangeable:

* ``instruction_dtype``        -- /root/reference/wfsim/strax_interface.py:25-42
* ``optical_extra_dtype``      -- /root/reference/wfsim/strax_interface.py:45-46
* ``truth_extra_dtype``        -- /root/reference/wfsim/strax_interface.py:49-73
* ``extra_truth_dtype_per_pmt``-- /root/reference/wfsim/strax_interface.py:76-116
* ``raw_record_dtype``         -- strax.raw_record_dtype (third party, strax>=1.6.0, not vendored by the
  reference; layout restated from its published definition: 244-byte packed record)
"""
import numpy as np

DEFAULT_RECORD_LENGTH = 110   # strax.DEFAULT_RECORD_LENGTH

instruction_dtype = [
    (('Waveform simulator event number.', 'event_number'), np.int32),
    (('Quanta type (S1 photons or S2 electrons)', 'type'), np.int8),
    (('Time of the interaction [ns]', 'time'), np.int64),
    (('X position of the cluster [cm]', 'x'), np.float32),
    (('Y position of the cluster [cm]', 'y'), np.float32),
    (('Z position of the cluster [cm]', 'z'), np.float32),
    (('Number of quanta', 'amp'), np.int32),
    (('Recoil type of interaction.', 'recoil'), np.int8),
    (('Energy deposit of interaction', 'e_dep'), np.float32),
    (('Total energy deposit in the sensitive volume', 'tot_e'), np.float32),
    (('Eventid like in geant4 output rootfile', 'g4id'), np.int32),
    (('Volume id giving the detector subvolume', 'vol_id'), np.int32),
    (('Local field [ V / cm ]', 'local_field'), np.float64),
    (('Number of excitons', 'n_excitons'), np.int32),
    (('X position of the primary particle [cm]', 'x_pri'), np.float32),
    (('Y position of the primary particle [cm]', 'y_pri'), np.float32),
    (('Z position of the primary particle [cm]', 'z_pri'), np.float32),
]

optical_extra_dtype = [
    (('first optical input index', '_first'), np.int32),
    (('last optical input index +1', '_last'), np.int32),
]

truth_extra_dtype = [
    (('End time of the interaction [ns]', 'endtime'), np.int64),
    (('Number of simulated electrons', 'n_electron'), np.int32),
    (('Number of photons reaching PMT', 'n_photon'), np.int32),
    (('Number of photons + dpe passing', 'n_pe'), np.int32),
    (('Number of photons passing trigger', 'n_photon_trigger'), np.int32),
    (('Number of photons + dpe passing trigger', 'n_pe_trigger'), np.int32),
    (('Raw area in pe', 'raw_area'), np.float64),
    (('Raw area in pe passing trigger', 'raw_area_trigger'), np.float64),
    (('Number of photons reaching PMT (bottom)', 'n_photon_bottom'), np.int32),
    (('Number of photons + dpe passing (bottom)', 'n_pe_bottom'), np.int32),
    (('Number of photons passing trigger (bottom)', 'n_photon_trigger_bottom'), np.int32),
    (('Number of photons + dpe passing trigger (bottom)', 'n_pe_trigger_bottom'), np.int32),
    (('Raw area in pe (bottom)', 'raw_area_bottom'), np.float64),
    (('Raw area in pe passing trigger (bottom)', 'raw_area_trigger_bottom'), np.float64),
    (('Arrival time of the first photon [ns]', 't_first_photon'), np.float64),
    (('Arrival time of the last photon [ns]', 't_last_photon'), np.float64),
    (('Mean time of the photons [ns]', 't_mean_photon'), np.float64),
    (('Standard deviation of photon arrival times [ns]', 't_sigma_photon'), np.float64),
    (('X field-distorted mean position of the electrons [cm]', 'x_mean_electron'), np.float32),
    (('Y field-distorted mean position of the electrons [cm]', 'y_mean_electron'), np.float32),
    (('Arrival time of the first electron [ns]', 't_first_electron'), np.float64),
    (('Arrival time of the last electron [ns]', 't_last_electron'), np.float64),
    (('Mean time of the electrons [ns]', 't_mean_electron'), np.float64),
    (('Standard deviation of electron arrival times [ns]', 't_sigma_electron'), np.float64),
]


def extra_truth_dtype_per_pmt(n_pmt):
    """Truth dtype with per-PMT fields instead of the total/bottom split when ``n_pmt`` is an int.

    Mirrors /root/reference/wfsim/strax_interface.py:76-116.
    """
    if not n_pmt:
        return truth_extra_dtype
    per_pmt = [
        (('Number of photons reaching PMT', 'n_photon_per_pmt'), (np.int32, n_pmt)),
        (('Number of photons + dpe passing', 'n_pe_per_pmt'), (np.int32, n_pmt)),
        (('Number of photons passing trigger', 'n_photon_trigger_per_pmt'), (np.int32, n_pmt)),
        (('Number of photons + dpe passing trigger', 'n_pe_trigger_per_pmt'), (np.int32, n_pmt)),
        (('Raw area in pe', 'raw_area_per_pmt'), (np.float64, n_pmt)),
        (('Raw area in pe passing trigger', 'raw_area_trigger_per_pmt'), (np.float64, n_pmt)),
    ]
    total = [
        (('Number of photons reaching PMT (total)', 'n_photon'), np.int32),
        (('Number of photons + dpe passing (total)', 'n_pe'), np.int32),
        (('Number of photons passing trigger (total)', 'n_photon_trigger'), np.int32),
        (('Number of photons + dpe passing trigger (total)', 'n_pe_trigger'), np.int32),
        (('Raw area in pe (total)', 'raw_area'), np.float64),
        (('Raw area in pe passing trigger (total)', 'raw_area_trigger'), np.float64),
    ]
    return truth_extra_dtype[:2] + per_pmt + total + truth_extra_dtype[14:]


def raw_record_dtype(samples_per_record=DEFAULT_RECORD_LENGTH):
    """strax raw-record layout: 24-byte header + int16 data[samples_per_record] (244 B at 110)."""
    return [
        (('Start time since unix epoch [ns]', 'time'), np.int64),
        (('Length of the interval in samples', 'length'), np.int32),
        (('Width of one sample [ns]', 'dt'), np.int16),
        (('Channel/PMT number', 'channel'), np.int16),
        (('Length of pulse to which the record belongs (without zero-padding)', 'pulse_length'), np.int32),
        (('Fragment number in the pulse', 'record_i'), np.int16),
        (('Baseline determined by the digitizer (if this is supported)', 'baseline'), np.int16),
        (('Waveform data in raw ADC counts', 'data'), np.int16, samples_per_record),
    ]


RAW_RECORD_NBYTES = np.dtype(raw_record_dtype()).itemsize
assert RAW_RECORD_NBYTES == 244
assert np.dtype(instruction_dtype).itemsize == 70
