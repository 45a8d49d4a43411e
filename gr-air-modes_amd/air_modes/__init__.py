"""air_modes -- MI355X-native drop-in for the gr-air-modes receive hot path.

Same names as the reference's Python surface for this path (python/__init__.py:34,43):
    air_modes.preamble(rate, threshold_db)      swig/air_modes_swig.i:15
    air_modes.slicer(queue)                     swig/air_modes_swig.i:16
    air_modes.rx_path(rate, threshold, queue, use_pmf=False, use_dcblock=False)
plus the two GNU Radio runtime types the path hands data over with (gr.msg_queue,
gr.message), because GNU Radio itself is not a dependency here.

All computation happens in libairmodes_hip.so (HIP, gfx950) through the C ABI of
include/airmodes_hip.h; importing this package does not load the library, constructing
a block does, and raises if the library or a HIP device is missing (no CPU fallback).
"""
from .msg_queue import message, msg_queue
from .blocks import preamble, slicer
from .rx_path import rx_path
from ._capi import AirModesError, Context, Library, PACKET_DTYPE, TAG_DTYPE, EXIT_DTYPE, shard_entries

__all__ = ["message", "msg_queue", "preamble", "slicer", "rx_path", "AirModesError", "Context",
           "Library", "PACKET_DTYPE", "TAG_DTYPE", "EXIT_DTYPE", "shard_entries"]
