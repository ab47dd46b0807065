"""wfsim_amd -- MI355X-native photon -> raw_records hot path of the XENONnT waveform simulator.

Public surface mirrors the reference's (``wfsim``) for this path:
``instruction_dtype``, ``truth_extra_dtype``, ``RawData``, ``RawDataOptical``, ``ChunkRawRecords``,
``RawRecordsFromFaxNT``, ``RawRecordsFromFax1T``.  Importing the package does not touch the GPU; constructing
``RawData`` / ``ChunkRawRecords`` does and fails loudly without libwfsim_amd.so or without an MI355X.
"""
from .dtypes import (instruction_dtype, optical_extra_dtype, truth_extra_dtype, extra_truth_dtype_per_pmt,  # noqa: F401
                     raw_record_dtype, DEFAULT_RECORD_LENGTH)
from .resource import DummyMap, make_map, Resource  # noqa: F401
from .config import load_fax_config, xenonnt_test_config, kernel_params  # noqa: F401
from .rawdata import RawData, RawDataOptical, PULSE_TYPE_NAMES  # noqa: F401
from .optical import optical_adjustment, read_optical, read_optical_events  # noqa: F401
from .strax_interface import (ChunkRawRecords, SimulatorPlugin, RawRecordsFromFaxNT, RawRecordsFromFax1T,  # noqa: F401
                              RawRecordsFromFaxOpticalNT, RawRecordsFromFaxnVeto, RawRecordsFromMcChain,
                              synchronise_timing, instruction_from_csv)

__version__ = '0.1.0'
