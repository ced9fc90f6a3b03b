"""Block-shaped mirror of the reference's Python surface for the receive path.

    rfid.gate(sample_rate)          <- gr::rfid::gate::make          (include/rfid/gate.h:51)
    rfid.tag_decoder(sample_rate)   <- gr::rfid::tag_decoder::make   (include/rfid/tag_decoder.h:48)
    rfid.reader(sample_rate, dac)   <- gr::rfid::reader::make        (include/rfid/reader.h:51)
    rfid.matched_filter(decim,taps) <- filter.fir_filter_ccc         (apps/reader.py:75)

As in the reference, the gate block is constructed first and owns the shared reader state
(lib/gate_impl.cc:67-69): here that state lives in an rfid.Context (one per RX stream); the
other blocks attach to the context of the most recently constructed gate unless one is
passed explicitly.  Each block's general_work() hands its buffer through the C-ABI to the
HIP kernels and returns what the C++ block would pass to consume_each()/produce().
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .context import Context

_current: Optional[Context] = None


def _ctx(ctx: Optional[Context]) -> Context:
    c = ctx or _current
    if c is None:
        raise RuntimeError("construct rfid.gate(...) first: it owns the shared reader state "
                           "(reference: gate_impl.cc:67-69, apps/reader.py:76-78)")
    return c


class _Block:
    def __init__(self, ctx: Context):
        self.ctx = ctx

    def forecast(self, noutput_items: int) -> int:
        return noutput_items          # gate_impl.cc:79-83, tag_decoder_impl.cc:72-76


class matched_filter(_Block):
    def __init__(self, decim: int = 5, taps=(1,) * 25, ctx: Optional[Context] = None, device: int = 0):
        global _current
        if ctx is None and _current is None:
            _current = Context(device=device)
        super().__init__(_ctx(ctx))
        if decim != 5 or len(taps) != 25 or any(complex(t) != 1 for t in taps):
            raise ValueError("only fir_filter_ccc(5, [1]*25) is built (apps/reader.py:65,75)")

    def work(self, x) -> np.ndarray:
        return self.ctx.mf_work(x)


class gate(_Block):
    def __init__(self, sample_rate: int, ctx: Optional[Context] = None, device: int = 0, **params):
        global _current
        if ctx is None:
            ctx = Context(device=device, sample_rate=int(sample_rate), **params)
            _current = ctx
        super().__init__(ctx)

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
