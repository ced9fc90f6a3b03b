"""MI355X-native Gen2 RFID receive path -- Python host layer.

Mirrors the Python surface of the reference's `rfid` module for the receive path
(gr-rfid/swig/rfid_swig.i:16-23: rfid.gate / rfid.tag_decoder / rfid.reader) on top of
librfid_mi355x.so (include/rfid_mi355x.h).  Import never touches the GPU; creating a
Context (or a gate block) does, and fails loudly without a gfx950 device.
"""
from . import _capi as capi
from .blocks import gate, matched_filter, reader, tag_decoder
from .context import Context, unpack_bits
from .flowgraph import reader_top_block
from . import batch, shard

__all__ = ["capi", "Context", "unpack_bits", "gate", "tag_decoder", "reader", "matched_filter",
           "reader_top_block", "batch", "shard"]
