"""Synthetic EPC Gen2 receive traces (2 Msps complex64, FM0 @ 40 kHz BLF).

Signal model:  x = L*tx + h*tag*tx + sigma*(N(0,1)+jN(0,1))
where `tx` is the reader's PIE envelope as reader_impl would transmit it
(timings: gr-rfid/lib/reader_impl.cc:51-71,84-125; bit fields:
gr-rfid/include/rfid/global_vars.h:113-133) at 1 us resolution, upsampled x2, and `tag`
is the FM0 backscatter level (preamble gr-rfid/include/rfid/global_vars.h:136).

This is a data generator (host side, numpy) for tests and bench inputs.  It is not on
the hot path and contains no decode logic.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

# durations in us == samples at the 1 Msps DAC rate (reader_impl.cc:51-60)
PW = 12
DELIM = 12
DATA0 = 24          # 12 high + 12 low
DATA1 = 48          # 36 high + 12 low
RTCAL = 72          # 60 high + 12 low
TRCAL = 200         # 188 high + 12 low
CW_QUERY = 240 + 480 + (17 + 6) * 25      # 1295  (reader_impl.cc:69)
CW_ACK = 3 * 240 + 480 + (129 + 6) * 25   # 4575  (reader_impl.cc:70)
T1_US = 250                               # tag reply delay after last rising edge
HALF_BIT_RAW = 25                         # 12.5 us at 2 Msps
TAG_PREAMBLE = (1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 1)


def _sym(high: int, total: int) -> np.ndarray:
    s = np.zeros(total, dtype=np.float32)
    s[:high] = 1.0
    return s


_D0 = _sym(12, DATA0)
_D1 = _sym(36, DATA1)
_DELIM = np.zeros(DELIM, dtype=np.float32)
_RTCAL = _sym(RTCAL - PW, RTCAL)
_TRCAL = _sym(TRCAL - PW, TRCAL)
_PREAMBLE = np.concatenate([_DELIM, _D0, _RTCAL, _TRCAL])
_FRAME_SYNC = np.concatenate([_DELIM, _D0, _RTCAL])


def crc5(bits: Sequence[int]) -> List[int]:
    """Gen2 CRC-5 (poly x^5+x^3+1, preset 01001) over `bits`, MSB first."""
    reg = 0b01001
    for b in bits:
        msb = (reg >> 4) & 1
        reg = (reg << 1) & 0x1F
        if msb ^ int(b):
            reg ^= 0b01001
    return [(reg >> i) & 1 for i in range(4, -1, -1)]


def crc16(bits: Sequence[int]) -> List[int]:
    """Gen2 CRC-16 (CCITT 0x1021, preset 0xFFFF, complemented) over `bits`, MSB first."""
    reg = 0xFFFF
    for b in bits:
        msb = (reg >> 15) & 1
        reg = (reg << 1) & 0xFFFF
        if msb ^ int(b):
            reg ^= 0x1021
    reg ^= 0xFFFF
    return [(reg >> i) & 1 for i in range(15, -1, -1)]


def pie(bits: Sequence[int]) -> np.ndarray:
    return np.concatenate([_D1 if b else _D0 for b in bits]) if len(bits) else np.zeros(0, np.float32)


def query_cmd(q: int) -> np.ndarray:
    # 1000 | DR | M(2) | TRext | Sel(2) | Session(2) | Target | Q(4) | CRC-5
    bits = [1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0] + [(q >> i) & 1 for i in (3, 2, 1, 0)]
    bits = bits + crc5(bits)
    return np.concatenate([_PREAMBLE, pie(bits)])


def query_rep_cmd() -> np.ndarray:
    return np.concatenate([_FRAME_SYNC, pie([0, 0, 0, 0])])


def ack_cmd(rn16: Sequence[int]) -> np.ndarray:
    return np.concatenate([_FRAME_SYNC, pie([0, 1] + list(rn16))])


def fm0_levels(bits: Sequence[int]) -> np.ndarray:
    """Half-bit backscatter levels: preamble, data bits, dummy 1."""
    lv = list(TAG_PREAMBLE)
    cur = lv[-1]
    for b in list(bits) + [1]:
        cur ^= 1            # inversion at every bit boundary
        lv.append(cur)
        if not b:
            cur ^= 1        # mid-bit inversion for a 0
        lv.append(cur)
    return np.asarray(lv, dtype=np.float32)


def epc_frame(epc96: Sequence[int], pc: Optional[Sequence[int]] = None) -> List[int]:
    pc = list(pc) if pc is not None else [0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    body = pc + list(epc96)
    assert len(body) == 112
    return body + crc16(body)


def epc_for_id(tag_id: int, rng: Optional[np.random.Generator] = None) -> List[int]:
    """96-bit EPC whose last byte (frame bits 104..111) is `tag_id`."""
    head = (rng.integers(0, 2, 88).tolist() if rng is not None else [0] * 88)
    return head + [(tag_id >> i) & 1 for i in range(7, -1, -1)]


@dataclass
class SlotTruth:
    round: int
    slot: int
    n_tags: int
    rn16: Optional[List[int]]      # bits backscattered (single responder) or None
    epc: Optional[List[int]]       # 128 frame bits actually sent (after any corruption)
    tag_id: Optional[int]
    epc_valid: bool                # CRC of the sent frame is intact
    rn16_raw_start: int = -1       # raw-sample index where the RN16 preamble begins
    epc_raw_start: int = -1


# rfid_synth_slot of include/rfid_mi355x.h (56 bytes): the slot table the device-side synthesiser
# (rfid_synth_gen2) expands into samples
SLOT_DTYPE = np.dtype([("cmd", "u1"), ("q", "u1"), ("n_tags", "u1"), ("has_epc", "u1"), ("tag", "u1", (8,)),
                       ("rn16", "<u2", (8,)), ("ack", "<u2"), ("reserved_", "<i2"), ("rn16_off_raw", "<i4"),
                       ("epc_off_raw", "<i4"), ("epc", "<u4", (4,))])
assert SLOT_DTYPE.itemsize == 56
MAX_RESP, MAX_TAGS = 8, 16


def _pack16(bits: Sequence[int]) -> int:
    v = 0
    for b in bits:
        v = (v << 1) | int(b)
    return v


def _pack_frame(bits: Sequence[int]) -> List[int]:
    w = [0, 0, 0, 0]
    for j, b in enumerate(bits):
        if b:
            w[j >> 5] |= 1 << (j & 31)
    return w


@dataclass
class TracePlan:
    """What rfid_synth_gen2 needs to build the same (noise-free) trace on the device."""
    slots: np.ndarray              # SLOT_DTYPE records
    leak: complex
    hs: List[complex]              # backscatter coefficient per tag index
    tail_us: int
    n_raw: int                     # samples of the trace


@dataclass
class Trace:
    samples: Optional[np.ndarray]  # complex64 @ 2 Msps (None when only the plan was asked for)
    slots: List[SlotTruth] = field(default_factory=list)
    fixed_q: int = 0
    plan: Optional[TracePlan] = None


def make_trace(n_rounds: int = 5, fixed_q: int = 0, tag_ids: Sequence[int] = (0x27,),
               sigma: float = 0.002, seed: int = 1, leak: complex = 1.0 * np.exp(0.7j),
               h: complex = 0.10 * np.exp(2.1j), corrupt_rounds: Sequence[int] = (),
               t1_us: float = T1_US, tail_us: int = 200, noise: bool = True,
               t1_jitter_raw: int = 0, render: bool = True) -> Trace:
    """Build one RX trace of `n_rounds` inventory rounds with 2**fixed_q slots each.

    Each tag picks a slot uniformly per round; slots with exactly one tag carry an
    RN16 and an EPC reply, collided slots carry the sum of the RN16 replies only, and
    empty slots carry noise.  The reader ACKs every slot (the reference has no
    empty-slot detection: SURVEY.md section 3.3).  `corrupt_rounds` flips one EPC frame
    bit in the first occupied slot of those rounds (1-based) so the CRC fails.

    `render=False` skips the numpy rendering (samples = None) and only returns the ground truth and the
    slot table (`.plan`) from which rfid_synth_gen2 builds the identical noise-free trace in HBM -- the
    way the large configurations (10 000 rounds x 16 slots = 17.5 GB) are generated.
    """
    rng = np.random.default_rng(seed)
    n_slots = 1 << fixed_q
    tx_parts: List[np.ndarray] = []
    events = []   # (start_us_of_reply, levels, amplitude h_k)
    slots: List[SlotTruth] = []
    t = 0

    def emit(seg: np.ndarray):
        nonlocal t
        if render:
            tx_parts.append(seg)
        t += len(seg)

    table = np.zeros(n_rounds * n_slots, dtype=SLOT_DTYPE)
    n_tab = 0

    emit(np.ones(CW_ACK, np.float32))             # START: cw_ack (reader_impl.cc:218-224)
    hs = [h * np.exp(1j * 0.9 * k) * (1.0 - 0.1 * (k % 3)) for k in range(len(tag_ids))]
    epcs = [epc_for_id(tid, rng) for tid in tag_ids]
    for r in range(1, n_rounds + 1):
        picks = rng.integers(0, n_slots, len(tag_ids)) if n_slots > 1 else np.zeros(len(tag_ids), int)
        corrupted = False
        for s in range(n_slots):
            emit(query_cmd(fixed_q) if s == 0 else query_rep_cmd())
            who = [k for k in range(len(tag_ids)) if picks[k] == s]
            jit = int(rng.integers(-t1_jitter_raw, t1_jitter_raw + 1)) if t1_jitter_raw else 0
            reply_at = t + t1_us + jit / 2.0
            truth = SlotTruth(r, s + 1, len(who), None, None, None, False)
            rec = table[n_tab]
            n_tab += 1
            rec["cmd"] = 0 if s == 0 else 1
            rec["q"] = fixed_q
            if len(who) > MAX_RESP or len(tag_ids) > MAX_TAGS:
                raise ValueError("slot table: at most %d responders per slot and %d tags" % (MAX_RESP, MAX_TAGS))
            rec["n_tags"] = len(who)
            rec["rn16_off_raw"] = int(round(reply_at * 2)) - 2 * t
            rn = None
            for i_who, k in enumerate(who):
                bits = rng.integers(0, 2, 16).tolist()
                if render:
                    events.append((reply_at, fm0_levels(bits), hs[k]))
                rec["tag"][i_who] = k
                rec["rn16"][i_who] = _pack16(bits)
                rn = bits
            if len(who) == 1:
                truth.rn16 = rn
                truth.rn16_raw_start = int(round(reply_at * 2))
            emit(np.ones(CW_QUERY, np.float32))
            ack_bits = rn if rn is not None else rng.integers(0, 2, 16).tolist()
            rec["ack"] = _pack16(ack_bits)
            emit(ack_cmd(ack_bits))
            if len(who) == 1:
                k = who[0]
                frame = epc_frame(epcs[k])
                valid = True
                if (r in corrupt_rounds) and not corrupted:
                    frame = list(frame)
                    frame[40] ^= 1
                    valid = False
                    corrupted = True
                jit = int(rng.integers(-t1_jitter_raw, t1_jitter_raw + 1)) if t1_jitter_raw else 0
                reply_at = t + t1_us + jit / 2.0
                if render:
                    events.append((reply_at, fm0_levels(frame), hs[k]))
                rec["has_epc"] = 1
                rec["epc_off_raw"] = int(round(reply_at * 2)) - 2 * t
                rec["epc"] = _pack_frame(frame)
                truth.epc = list(frame)
                truth.tag_id = tag_ids[k]
                truth.epc_valid = valid
                truth.epc_raw_start = int(round(reply_at * 2))
            emit(np.ones(CW_ACK, np.float32))
            slots.append(truth)
    emit(np.ones(tail_us, np.float32))
    plan = TracePlan(slots=table[:n_tab], leak=complex(np.complex64(leak)), hs=[complex(np.complex64(v)) for v in hs],
                     tail_us=int(tail_us), n_raw=2 * t)
    if not render:
        return Trace(samples=None, slots=slots, fixed_q=fixed_q, plan=plan)

    tx = np.repeat(np.concatenate(tx_parts), 2)     # 1 Msps -> 2 Msps
    x = (np.complex64(leak) * tx).astype(np.complex64)
    for (start_us, levels, hk) in events:
        a = int(round(start_us * 2))
        wave = np.repeat(levels, HALF_BIT_RAW)
        b = min(a + len(wave), len(x))
        x[a:b] += (np.complex64(hk) * wave[: b - a] * tx[a:b]).astype(np.complex64)
    if noise and sigma > 0:
        n = rng.standard_normal((len(x), 2), dtype=np.float32)
        x += (np.float32(sigma) * (n[:, 0] + 1j * n[:, 1])).astype(np.complex64)
    return Trace(samples=np.ascontiguousarray(x, dtype=np.complex64), slots=slots, fixed_q=fixed_q, plan=plan)


def fst_like_trace(sigma: float = 0.002, seed: int = 7) -> Trace:
    """Stand-in for the reference's missing misc/data/file_source_test: 71 rounds,
    FIXED_Q=0, one tag with id 0x27, one EPC corrupted -> expected print_results:
    71 queries / round 72 / 70 correct / 1 unique / 'Tag ID : 27  Num of reads : 70'
    (README.md:48-53)."""
    return make_trace(n_rounds=71, fixed_q=0, tag_ids=(0x27,), sigma=sigma, seed=seed,
                      corrupt_rounds=(36,))


def add_noise_replicas(base: np.ndarray, n_rep: int, sigma: float, seed: int) -> np.ndarray:
    """[n_rep, L] replicas of a noise-free base trace, replica r seeded seed+r."""
    out = np.empty((n_rep, len(base)), dtype=np.complex64)
    for r in range(n_rep):
        rng = np.random.default_rng(seed + r)
        n = rng.standard_normal((len(base), 2), dtype=np.float32)
        out[r] = base + (np.float32(sigma) * (n[:, 0] + 1j * n[:, 1])).astype(np.complex64)
    return out
