"""Shared comparison of a HIP (or emulated-kernel) batch result with the oracle.

Bar: integer / index / bit outputs bit-exact; floating-point scores are compared bit-exact
too (the kernels keep the reference's binary32 operation order) -- the north star's 1e-5
relative tolerance is the fallback bound, asserted as well."""
import numpy as np

from rfid.context import unpack_bits

SCORE_RTOL = 1e-5


def _bits_equal(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def compare_trace(windows, results, scores, stats, o, exact_scores=True):
    """windows/results/scores: records of ONE trace ordered by seq; stats: its stats record;
    o: oracle.Result for the same trace (no termination inside)."""
    n = o.n_windows
    assert len(windows) == n, (len(windows), n)
    assert np.array_equal(windows["start"], o.open_idx)
    assert np.array_equal(windows["type"], o.dumps["type"])
    assert _bits_equal(windows["dc_re"], o.dc.real) and _bits_equal(windows["dc_im"], o.dc.imag), "dc_est"
    for i in range(n):
        d = o.dumps[i]
        r = results[i]
        assert r["type"] == d["type"] and r["index"] == d["index"], i
        assert r["n_bits"] == d["n_bits"]
        assert np.array_equal(unpack_bits(r["bits"], int(d["n_bits"])), d["bits"][: d["n_bits"]]), f"bits {i}"
        assert _bits_equal([r["h_re"], r["h_im"]], d["h_est"]), f"h_est {i}"
        if scores is not None:
            np.testing.assert_allclose(scores["corr"][i], d["corr"], rtol=SCORE_RTOL, atol=0)
            if exact_scores:
                assert _bits_equal(scores["corr"][i], d["corr"]), f"corr {i}"
        if d["type"] == 1:
            assert _bits_equal(r["T"], d["T"]), f"T {i}"
            assert r["crc_ok"] == d["crc_ok"], i
            if d["crc_ok"]:
                assert r["tag_id"] == d["tag_id"]
            if scores is not None:
                np.testing.assert_allclose(scores["energy"][i], d["energy"], rtol=SCORE_RTOL, atol=0)
                if exact_scores:
                    assert _bits_equal(scores["energy"][i], d["energy"]), f"energy {i}"
    if stats is not None:
        s = o.state
        assert stats["n_windows"] == n
        for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct", "n_unique_tags"):
            assert stats[k] == getattr(s, k), (k, stats[k], getattr(s, k))
        assert stats["status"] == s.status
        assert np.array_equal(stats["tag_reads"], np.array(s.tag_reads[:], dtype=np.int32))


def split_by_stream(windows, results, scores, n_streams):
    out = []
    for b in range(n_streams):
        m = windows["stream"] == b
        out.append((windows[m], results[m], scores[m] if scores is not None else None))
    return out


def compare_trace_fast(windows, results, stats, o):
    """compare_trace() without scores, vectorised over the windows (hundreds of thousands in the full-size
    configurations): every start, type, dc_est, sync index, h_est, T, decoded bit, CRC flag and tag id."""
    n = o.n_windows
    assert len(windows) == n and len(o.dumps) == n, (len(windows), n, len(o.dumps))
    d = o.dumps
    assert np.array_equal(windows["start"], o.open_idx) and np.array_equal(windows["type"], d["type"])
    assert _bits_equal(windows["dc_re"], o.dc.real) and _bits_equal(windows["dc_im"], o.dc.imag), "dc_est"
    assert np.array_equal(results["type"], d["type"]) and np.array_equal(results["index"], d["index"])
    assert np.array_equal(results["n_bits"], d["n_bits"])
    assert _bits_equal(results["h_re"], d["h_est"][:, 0]) and _bits_equal(results["h_im"], d["h_est"][:, 1]), "h_est"
    epc = d["type"] == 1
    assert _bits_equal(results["T"][epc], d["T"][epc]), "T"
    assert np.array_equal(results["crc_ok"][epc], d["crc_ok"][epc])
    ok = epc & (d["crc_ok"] == 1)
    assert np.array_equal(results["tag_id"][ok], d["tag_id"][ok])
    # bits: oracle keeps one byte per bit (frame order); the result packs bit j at word j>>5, bit j&31
    j = np.arange(128)
    got = ((results["bits"][:, j >> 5] >> (j & 31).astype(np.uint32)) & 1).astype(np.uint8)
    mask = j[None, :] < d["n_bits"][:, None]
    assert np.array_equal(got[mask], d["bits"][mask]), "decoded bits"
    if stats is not None:
        s = o.state
        assert stats["n_windows"] == n
        for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct", "n_unique_tags"):
            assert stats[k] == getattr(s, k), (k, stats[k], getattr(s, k))
        assert np.array_equal(stats["tag_reads"], np.array(s.tag_reads[:], dtype=np.int32))
