"""air_modes -- MI355X-native drop-in for the gr-air-modes receive hot path.

Same names as the reference's Python surface for this path (python/__init__.py:34,43):
    air_modes.preamble(rate, threshold_db)      swig/air_modes_swig.i:15
    air_modes.slicer(queue)                     swig/air_modes_swig.i:16
    air_modes.rx_path(rate, threshold, queue, use_pmf=False, use_dcblock=False)
plus the two GNU Radio runtime types the path hands data over with (gr.msg_queue,
gr.message), because GNU Radio itself is not a dependency here, and -- the callers either side of the path --
the message consumers: make_parser / modes_reply (python/parse.py), output_print
(python/msprint.py), cpr_decoder (python/cpr.py), decode_alt (python/altitude.py) and the
file-source command line `python -m air_modes.modes_rx` (apps/modes_rx).

All computation happens in libairmodes_hip.so (HIP, gfx950) through the C ABI of
include/airmodes_hip.h; importing this package does not load the library, constructing
a block does, and raises if the library or a HIP device is missing (no CPU fallback).
"""
from .msg_queue import message, msg_queue
from .blocks import preamble, slicer
from .rx_path import rx_path, rx_path_bank
from ._capi import AirModesError, Context, Library, PACKET_DTYPE, TAG_DTYPE, EXIT_DTYPE, shard_entries
# message consumers (host side, after the hot path): python/__init__.py:45-63 exports the same names
from .exceptions import *            # noqa: F401,F403
from .exceptions import (ADSBError, MetricAltError, ParserError, NoHandlerError, MlatNonConvergeError,
                         CPRNoPositionError, CPRBoundaryStraddleError, FieldNotInPacket)
from .altitude import decode_alt, gray2bin, encode_alt_modes
from .modes_types import stamp, modes_report, llh, mlat_report
from .cpr import cpr_decoder, cpr_encode, cpr_resolve_local, cpr_resolve_global, range_bearing
from .parse import (data_field, modes_reply, me_reply, mb_reply, mv_reply, bds09_reply, tcas_reply, make_parser,
                    decode_id, charmap, parseBDS08, parseBDS05, parseBDS06, parseBDS09_0, parseBDS09_1,
                    parseBDS09_3, parseBDS62, parseMB_id, parseMB_TCAS_resolutions, parseMB_TCAS_threatid,
                    parseMB_TCAS_threatloc, parse_TCAS_CRM)
from .msprint import output_print
from .pubsub import pubsub

__all__ = ["message", "msg_queue", "preamble", "slicer", "rx_path", "rx_path_bank", "AirModesError", "Context",
           "Library", "PACKET_DTYPE", "TAG_DTYPE", "EXIT_DTYPE", "shard_entries",
           "make_parser", "modes_reply", "output_print", "cpr_decoder", "decode_alt", "decode_id", "stamp",
           "modes_report", "pubsub", "ADSBError"]
