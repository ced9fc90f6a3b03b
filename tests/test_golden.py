"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
oracle) against: the oracle itself (drift guard), the emulated kernel source, and -- on the
GPU box -- the HIP path through the C-ABI."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _bits_eq(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def _check(g, windows, results, scores, stats, mf):
    from rfid.context import unpack_bits
    n = len(g["type"])
    assert len(windows) == n
    assert _bits_eq(mf.view(np.float32), g["mf"].view(np.float32)), "matched filter output"
    assert np.array_equal(windows["start"], g["open_idx"]) and np.array_equal(windows["type"], g["type"])
    assert _bits_eq(windows["dc_re"], g["dc"].real) and _bits_eq(windows["dc_im"], g["dc"].imag)
    for i in range(n):
        assert results["index"][i] == g["index"][i]
        nb = int(g["n_bits"][i])
        assert np.array_equal(unpack_bits(results["bits"][i], nb), g["bits"][i][:nb])
        assert _bits_eq([results["h_re"][i], results["h_im"][i]], g["h_est"][i])
        assert _bits_eq(scores["corr"][i], g["corr"][i])
        np.testing.assert_allclose(scores["corr"][i], g["corr"][i], rtol=1e-5)   # north-star tolerance
        if g["type"][i] == 1:
            assert _bits_eq(scores["energy"][i], g["energy"][i]) and _bits_eq(results["T"][i], g["T"][i])
            assert results["crc_ok"][i] == g["crc_ok"][i]
            if g["crc_ok"][i]:
                assert results["tag_id"][i] == g["tag_id"][i]
    got = [stats[k] for k in ("n_queries_sent", "cur_inventory_round", "cur_slot_number", "n_epc_correct",
                              "n_unique_tags", "status")]
    assert got == list(g["stats"])
    assert np.array_equal(stats["tag_reads"], g["tag_reads"])


def test_fixtures_present():
    assert len(FIXTURES) >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(oracle_mod, path):
    g = np.load(path)
    o = oracle_mod.run_trace(g["raw"], oracle_mod.config(fixed_q=int(g["fixed_q"])))
    assert np.array_equal(o.open_idx, g["open_idx"]) and np.array_equal(o.dumps["bits"], g["bits"])
    assert _bits_eq(o.dumps["corr"], g["corr"]) and _bits_eq(o.dumps["energy"], g["energy"])
    assert _bits_eq(oracle_mod.fir(g["raw"]).view(np.float32), g["mf"].view(np.float32))
    assert o.print_results().encode() == g["print_results"].tobytes()


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_emulated_kernels_reproduce_golden(emu_mod, path):
    g = np.load(path)
    r = emu_mod.batch_process(g["raw"][None, :], fixed_q=int(g["fixed_q"]), want_y=True)
    _check(g, r["windows"], r["results"], r["scores"], r["stats"][0], r["y"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_path_reproduces_golden(path):
    import torch
    import rfid
    g = np.load(path)
    raw = g["raw"]
    L = len(raw)
    stride = (L + 1) & ~1
    host = np.zeros((1, stride), dtype=np.complex64)
    host[0, :L] = raw
    dev = torch.from_numpy(host.view(np.float32)).to("cuda:0")
    ctx = rfid.Context(device=0, fixed_q=int(g["fixed_q"]))
    try:
        ctx.batch_plan(1, L)
        ctx.batch_process_ptr(dev.data_ptr(), stride, L, 0, want_scores=True)
        ctx.batch_sync()
        w, r, s = ctx.batch_windows(want_scores=True)
        _check(g, w, r, s, ctx.batch_stats()[0], ctx.batch_mf_output(0))
    finally:
        ctx.close()
