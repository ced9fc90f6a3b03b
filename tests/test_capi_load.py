"""The C-ABI library loads and exports every symbol include/rfid_mi355x.h declares.
(No compute calls here: those need the GPU and live in test_gpu_parity.py.)"""
import ctypes as C
import os
import re

import pytest


def _declared():
    import rfid
    text = open(rfid.capi.HEADER_PATH).read()
    return re.findall(r"RFID_API\s+[\w\s\*]+?\b(rfid_\w+)\s*\(", text)


def test_library_exists_and_exports_all_declared_symbols():
    import rfid
    names = _declared()
    assert len(names) >= 25, names
    lib = rfid.capi.load()          # raises if the .so is missing: there is no fallback
    for n in names:
        assert hasattr(lib, n), f"{n} not exported"
        assert n in rfid.capi.SIGNATURES, f"{n} has no ctypes signature"
    assert set(rfid.capi.SIGNATURES) == set(names)


def test_abi_struct_sizes_match_header_layout():
    import rfid
    assert C.sizeof(rfid.capi.Params) == 24
    assert C.sizeof(rfid.capi.ReaderState) == 4 * (11 + 256)
    assert rfid.capi.WINDOW_DTYPE.itemsize == 24 and rfid.capi.RESULT_DTYPE.itemsize == 48
    assert rfid.capi.SCORES_DTYPE.itemsize == 144 and rfid.capi.STATS_DTYPE.itemsize == 4 * (8 + 256)


def test_defaults_are_the_reference_constants():
    import rfid
    p = rfid.capi.default_params()
    # apps/reader.py:53-54,65,76; include/rfid/global_vars.h:72,76,100
    assert (p.sample_rate, p.decim, p.n_taps, p.fixed_q, p.max_num_queries, p.number_unique_tags) == \
        (400000, 5, 25, 0, 1000, 100)


def test_no_cpu_fallback_without_gpu():
    """Without a usable gfx950 device context creation must fail loudly."""
    import torch
    import rfid
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rfid.capi.RfidError) as e:
        rfid.Context(device=0)
    assert e.value.status == rfid.capi.ERR_NO_DEVICE


def test_unsupported_parameters_are_rejected():
    import rfid
    with pytest.raises(rfid.capi.RfidError) as e:
        rfid.Context(device=0, sample_rate=200000)
    assert e.value.status == rfid.capi.ERR_UNSUPPORTED


def test_product_never_imports_oracle_or_emulator():
    """The product package must not reference oracle/ or tests/wave_emu."""
    import rfid
    pkg = os.path.dirname(os.path.dirname(rfid.__file__))
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")) or f == "Makefile":
                text = open(os.path.join(root, f)).read()
                assert "librfid_oracle" not in text and "import oracle" not in text, f
                assert "from oracle" not in text and "orc_" not in text, f
                assert "emu_driver" not in text and "librfid_wave_emu" not in text, f


def test_cxx_block_adaptors_build_and_refuse_without_gpu(tmp_path):
    """cxx/rfid_blocks.hpp (gate / tag_decoder / reader / matched_filter adaptors with the reference's
    factory names and general_work signatures) compiles against the C-ABI header, and the offline
    flowgraph binary fails loudly -- exit code 3 -- when there is no gfx950 device."""
    import subprocess
    import torch
    import rfid
    cxx = os.path.join(rfid.capi.PKG_ROOT, "cxx")
    subprocess.check_call(["make", "-C", cxx], stdout=subprocess.DEVNULL)
    exe = os.path.join(rfid.capi.PKG_ROOT, "bin", "rfid_reader_offline")
    assert subprocess.run([exe], capture_output=True).returncode == 2            # usage
    assert subprocess.run([exe, str(tmp_path / "missing.bin")], capture_output=True).returncode == 2
    if not torch.cuda.is_available():
        trace = tmp_path / "t.bin"
        trace.write_bytes(b"\0" * 8 * 1000)
        r = subprocess.run([exe, str(trace)], capture_output=True, text=True)
        assert r.returncode == 3 and "no usable gfx950 device" in r.stderr
