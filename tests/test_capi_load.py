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
    """cxx/include/rfid/*.h + cxx/lib/rfid_blocks.cc (gr::rfid::gate / tag_decoder / reader with the reference's exact
    factories, the reader_state global, general_work signatures) compile against the C-ABI header into
    libgnuradio-rfid.so, and the offline flowgraph binary fails loudly -- exit code 3 -- when there is no gfx950 device."""
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


def test_cxx_block_api_is_the_references(tmp_path):
    """A translation unit written only against the reference's public block API -- gr::rfid::gate::make(int),
    tag_decoder::make(int), reader::make(int,int) + print_results(), `extern READER_STATE* reader_state`,
    `initialize_reader_state()`, sptr typedefs, `virtual public gr::block`, forecast/general_work signatures
    (gr-rfid/include/rfid/gate.h:51, tag_decoder.h:48, reader.h:42,51, global_vars.h:146-147) -- compiles and links
    against the shipped headers and libgnuradio-rfid.so.  (Construction needs a GPU: tests/test_gpu_round2.py.)"""
    import subprocess
    import rfid
    cxx = os.path.join(rfid.capi.PKG_ROOT, "cxx")
    subprocess.check_call(["make", "-C", cxx], stdout=subprocess.DEVNULL)
    src = tmp_path / "api_tu.cc"
    src.write_text(r"""
#include <rfid/gate.h>
#include <rfid/tag_decoder.h>
#include <rfid/reader.h>
#include <rfid/global_vars.h>
#include <type_traits>
using namespace gr::rfid;
static_assert(std::is_base_of<gr::block, gate>::value && std::is_base_of<gr::block, tag_decoder>::value &&
              std::is_base_of<gr::block, reader>::value, "blocks derive from gr::block");
static_assert(std::is_abstract<reader>::value, "reader::print_results is pure virtual");
gate::sptr (*f_gate)(int) = &gate::make;
tag_decoder::sptr (*f_dec)(int) = &tag_decoder::make;
reader::sptr (*f_reader)(int, int) = &reader::make;
void (reader::*f_print)() = &reader::print_results;
void (*f_init)() = &initialize_reader_state;
int (gr::block::*f_work)(int, gr_vector_int &, gr_vector_const_void_star &, gr_vector_void_star &) = &gr::block::general_work;
void (gr::block::*f_forecast)(int, gr_vector_int &) = &gr::block::forecast;
static_assert(FIXED_Q == 0 && MAX_NUM_QUERIES == 1000 && NUMBER_UNIQUE_TAGS == 100 && RN16_BITS == 17 && EPC_BITS == 129, "");
static_assert(SEND_ACK == 1 && IDLE == 3 && START == 5 && GATE_SEEK_EPC == 3 && DECODER_DECODE_EPC == 1 && TERMINATED == 1, "");
int main(int argc, char **) {
  READER_STATE **p = &reader_state;        // the global of include/rfid/global_vars.h:146
  if (argc > 100) {                        // never executed here: needs a GPU
    gate::sptr g = gate::make(400000);
    tag_decoder::sptr d = tag_decoder::make(400000);
    reader::sptr r = reader::make(400000, 1000000);
    r->print_results();
    return (*p)->reader_stats.n_epc_correct + (int)(*p)->reader_stats.tag_reads.size() + (*p)->n_samples_to_ungate;
  }
  return *p == nullptr ? 0 : 1;
}
""")
    exe = tmp_path / "api_tu"
    libdir = os.path.join(rfid.capi.PKG_ROOT, "lib")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(cxx, "minigr"), "-I", os.path.join(cxx, "include"),
                           "-I", os.path.join(rfid.capi.REPO_ROOT, "include"), str(src), "-o", str(exe), "-L", libdir,
                           "-lgnuradio-rfid", "-lrfid_mi355x", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.run([str(exe)]).returncode == 0
    syms = subprocess.run(["nm", "-DC", os.path.join(libdir, "libgnuradio-rfid.so")], capture_output=True, text=True).stdout
    for want in ("gr::rfid::gate::make(int)", "gr::rfid::tag_decoder::make(int)", "gr::rfid::reader::make(int, int)",
                 "gr::rfid::reader_state", "gr::rfid::initialize_reader_state()"):
        assert want in syms, want
