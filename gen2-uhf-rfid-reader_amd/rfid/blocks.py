"""Block-shaped mirror of the reference's Python surface for the receive path.

    rfid.gate(sample_rate)          <- gr::rfid::gate::make          (include/rfid/gate.h:51)
    rfid.tag_decoder(sample_rate)   <- gr::rfid::tag_decoder::make   (include/rfid/tag_decoder.h:48)
    rfid.reader(sample_rate, dac)   <- gr::rfid::reader::make        (include/rfid/reader.h:51)
    rfid.matched_filter(decim,taps) <- filter.fir_filter_ccc         (apps/reader.py:75)

As in the reference, the gate block owns the shared reader state (lib/gate_impl.cc:67-69): here that state
lives in an rfid.Context (one per RX stream).  The reference's factories take no stream argument
(`rfid.tag_decoder(int)`, `rfid.reader(int, int)`), so the binding rule is the reference's construction order
(apps/reader.py:75-78), made explicit and per thread instead of a process-wide "current context":
  * tag_decoder / reader bind, at construction, to the stream of the most recently constructed gate of this thread;
  * a matched_filter binds to that stream too if it has no matched filter yet, otherwise (or when no gate exists
    yet -- apps/reader.py:75 builds the filter BEFORE the gate) it waits for the next gate of this thread.
Several flowgraphs in one process therefore never share or steal state; passing `ctx=` binds explicitly.
Each block's general_work() hands its buffer through the C-ABI to the HIP kernels and returns what the C++
block would pass to consume_each()/produce().
"""
from __future__ import annotations

import threading
from typing import Optional, Tuple

import numpy as np

from .context import Context


class _Binding(threading.local):
    def __init__(self):
        self.current: Optional[Context] = None      # stream of the most recent gate of this thread
        self.has_filter = False                     # ... already has its matched filter
        self.pending_filters = []                   # matched filters built before their gate


_bind = _Binding()


def _ctx(ctx: Optional[Context]) -> Context:
    c = ctx or _bind.current
    if c is None:
        raise RuntimeError("construct rfid.gate(...) first: it owns the shared reader state "
                           "(reference: gate_impl.cc:67-69, apps/reader.py:76-78)")
    return c


class _Block:
    def __init__(self, ctx: Optional[Context]):
        self.ctx = ctx

    def forecast(self, noutput_items: int) -> int:
        return noutput_items          # gate_impl.cc:79-83, tag_decoder_impl.cc:72-76


class matched_filter(_Block):
    def __init__(self, decim: int = 5, taps=(1,) * 25, ctx: Optional[Context] = None):
        if decim != 5 or len(taps) != 25 or any(complex(t) != 1 for t in taps):
            raise ValueError("only fir_filter_ccc(5, [1]*25) is built (apps/reader.py:65,75)")
        # (a gate built before its filter takes it -- unless its stream is closed already, or it said that its filter is
        # somebody else's: filter_is_external())
        if ctx is None and _bind.current is not None and not _bind.has_filter and getattr(_bind.current, "_h", None):
            ctx = _bind.current
            _bind.has_filter = True
        super().__init__(ctx)
        if ctx is None:
            _bind.pending_filters.append(self)      # bound by the next gate (apps/reader.py:75-76 order)

    def work(self, x) -> np.ndarray:
        if self.ctx is None:
            raise RuntimeError("matched_filter is not bound to a stream yet: construct rfid.gate(...)")
        return self.ctx.mf_work(x)


class gate(_Block):
    def __init__(self, sample_rate: int, ctx: Optional[Context] = None, device: int = 0, **params):
        if ctx is None:
            ctx = Context(device=device, sample_rate=int(sample_rate), **params)
        super().__init__(ctx)
        _bind.current = ctx
        _bind.has_filter = False
        if _bind.pending_filters:
            _bind.pending_filters.pop(0).ctx = ctx
            _bind.has_filter = True

    def filter_is_external(self) -> None:
        """This gate is fed by a filter that is not this library's (apps/reader.py:75 as it stands): a rfid.matched_filter built
        later belongs to the NEXT gate, not to this one."""
        if _bind.current is self.ctx:
            _bind.has_filter = True

    def general_work(self, x) -> Tuple[int, np.ndarray]:
        return self.ctx.gate_work(x)


class tag_decoder(_Block):
    def __init__(self, sample_rate: int, ctx: Optional[Context] = None):
        super().__init__(_ctx(ctx))
        if int(sample_rate) != self.ctx.params.sample_rate:
            raise ValueError("tag_decoder sample_rate differs from the gate's")

    def general_work(self, x):
        """-> (consumed, port0_floats, result, scores); port 1 (complex debug) never produces,
        as in the reference (tag_decoder_impl.cc:227-234)."""
        return self.ctx.decoder_work(x)


class reader(_Block):
    def __init__(self, sample_rate: int, dac_rate: int, ctx: Optional[Context] = None):
        super().__init__(_ctx(ctx))
        self.dac_rate = int(dac_rate)

    def forecast(self, noutput_items: int) -> int:
        return 0                      # reader_impl.cc:194-198

    def general_work(self, n_in: int) -> int:
        """The state transitions of reader_impl::general_work alone -> consumed."""
        return self.ctx.reader_work(n_in)

    def general_work_tx(self, in_bits=None):
        """reader_impl::general_work complete: transitions + the transmit waveform written to the block's
        output (float32 samples at dac_rate) -> (consumed, samples)."""
        return self.ctx.reader_work_tx(in_bits, dac_rate=self.dac_rate)

    def print_results(self) -> None:
        print(self.ctx.print_results(), end="")
