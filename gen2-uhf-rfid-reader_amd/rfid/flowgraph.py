"""Offline flowgraph: the DEBUG=True topology of apps/reader.py:101-112
(file_source -> matched_filter -> gate -> tag_decoder -> reader), driven by a
single-threaded scheduler (the README's GR_SCHEDULER=STS mode, README.md:40).

Every block call goes through the C-ABI into the HIP kernels.  This is the drop-in,
call-per-buffer form; bulk offline decoding uses rfid.batch instead.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _capi as capi
from . import blocks


def fir_filter_ccc_ones(x: np.ndarray, decim: int = 5, n_taps: int = 25) -> np.ndarray:
    """Somebody else's matched filter: what filter.fir_filter_ccc(5, [1]*25) (apps/reader.py:65,75) computes, in plain
    numpy on the host -- history of n_taps - 1 zeros, one output per complete group of `decim` inputs, the taps (all one)
    summed in tap order in binary32.  For the flowgraph whose filter is NOT this library's block (external_filter=True)."""
    x = np.ascontiguousarray(x, dtype=np.complex64)
    n_out = len(x) // decim
    xp = np.concatenate([np.zeros(n_taps - 1, dtype=np.complex64), x])
    re = np.zeros(n_out, dtype=np.float32)
    im = np.zeros(n_out, dtype=np.float32)
    for k in range(n_taps):
        seg = xp[k: k + decim * n_out: decim]
        re = re + seg.real
        im = im + seg.imag
    return (re + 1j * im).astype(np.complex64)


class reader_top_block:
    def __init__(self, source_path: Optional[str] = None, samples: Optional[np.ndarray] = None,
                 device: int = 0, chunk: int = 8192, lookahead: bool = False, external_filter: bool = False, **params):
        # variables of apps/reader.py:52-65
        self.dac_rate = 1e6
        self.adc_rate = 100e6 / 50
        self.decim = 5
        self.num_taps = [1] * 25
        self.chunk = int(chunk)
        if samples is None:
            if source_path is None:
                raise ValueError("need source_path (interleaved float32 I,Q file) or samples")
            samples = np.fromfile(source_path, dtype=np.complex64)   # blocks.file_source, reader.py:102
        self.samples = np.ascontiguousarray(samples, dtype=np.complex64)
        # the blocks of apps/reader.py:75-78, in that order and with those arguments (device / params are what
        # the reference fixes at compile time); the gate owns the stream, the others bind to it (rfid/blocks.py)
        # external_filter: apps/reader.py as it stands -- the matched filter is GNU Radio's own block, not this library's;
        # here fir_filter_ccc_ones() on the host, and only gate / tag_decoder / reader are made
        self.external_filter = bool(external_filter)
        self.matched_filter = None if external_filter else blocks.matched_filter(self.decim, self.num_taps)
        self.gate = blocks.gate(int(self.adc_rate / self.decim), device=device, **params)
        if external_filter:
            self.gate.filter_is_external()
        self.tag_decoder = blocks.tag_decoder(int(self.adc_rate / self.decim))
        self.reader = blocks.reader(int(self.adc_rate / self.decim), int(self.dac_rate))
        self.ctx = self.gate.ctx
        assert (external_filter or self.matched_filter.ctx is self.ctx) and self.tag_decoder.ctx is self.ctx and self.reader.ctx is self.ctx
        self.decoded = []        # (result, scores) per decoded window, for inspection
        if lookahead and external_filter:   # ... keyed on the gate's own input: gate -> tag_decoder per buffer it is shown
            self.ctx.lookahead_enable_gate(2 * self.chunk)
        elif lookahead:          # the library answers the gate / decoder calls from one whole-chain pass per filter call
            self.ctx.lookahead_enable(self.chunk * self.decim)

    def _reader_until_idle(self, q: int) -> None:
        for _ in range(8):
            before = self.ctx.state().gen2_logic_status
            if before == capi.IDLE:
                break
            self.reader.general_work(q)
            q = 0
            if self.ctx.state().gen2_logic_status == before:
                break

    def run(self) -> None:
        self._reader_until_idle(0)                    # START -> SEND_QUERY -> IDLE
        dq = np.zeros(0, dtype=np.complex64)          # decoder input buffer
        gq = np.zeros(0, dtype=np.complex64)          # gate input buffer
        pos = 0
        src = fir_filter_ccc_ones(self.samples, self.decim, len(self.num_taps)) if self.external_filter else self.samples
        per = 1 if self.external_filter else self.decim
        view = 2 * self.chunk if self.external_filter else self.chunk   # (a gate behind a foreign filter sees a scheduler's buffer:
        n = len(src)                                                     #  what it has not consumed yet and what came in since)
        flushed = False
        idle = 0
        while pos < n or len(gq):
            if pos < n:
                blk = src[pos:pos + self.chunk * per]
                pos += len(blk)
                y = blk if self.external_filter else self.matched_filter.work(blk)
                gq = np.concatenate([gq, y]) if len(gq) else y
            progressed = False
            while len(gq):
                take = gq if pos >= n else gq[:view]      # (at the end of the input the gate is shown everything that is left)
                consumed, out = self.gate.general_work(take)
                progressed = progressed or consumed > 0 or len(out) > 0
                gq = gq[consumed:]
                if len(out):
                    dq = np.concatenate([dq, out]) if len(dq) else out
                while True:
                    dcons, bits, res, sc = self.tag_decoder.general_work(dq)
                    if dcons == 0:
                        break
                    self.decoded.append((res, sc))
                    dq = dq[dcons:]
                    self._reader_until_idle(len(bits))
                if consumed == 0:
                    break
            if pos >= n:
                idle = 0 if progressed else idle + 1
                if not len(gq) or (flushed and idle > 4):     # (a gate-keyed look-ahead carries the flush out on the third idle call)
                    break
                if not flushed:
                    self.ctx.lookahead_flush()       # the source has run dry: what the look-ahead holds back is decided now
                    flushed = True

    def start(self) -> None:          # gr.top_block.start() analogue (apps/reader.py:123)
        self.run()

    def stop(self) -> None:
        pass
