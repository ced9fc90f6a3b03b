// oracle/ref_harness.cc -- TEST INFRASTRUCTURE, own code.  Runs the REFERENCE's blocks (compiled from /root/reference, see
// oracle/Makefile `ref`) in the offline topology of apps/reader.py:101-112 (DEBUG = True):
//     file_source -> fir_filter_ccc(5, [1]*25) -> gate -> tag_decoder -> reader -> multiply_const(0) -> to_complex -> file_sink
//     gate -> file_sink (the gated samples),  matched filter -> file_sink
// and prints reader::print_results().  Only buildable where a real GNU Radio 3.7 is installed; nothing in this image
// can compile it, nothing in the product or in the default test run uses it (tests/test_reference_pin.py skips).
//
//   ref_harness TRACE_FILE OUT_PREFIX      -> OUT_PREFIX.mf, OUT_PREFIX.gate, OUT_PREFIX.reader (+ the report on stdout)
#include <gnuradio/blocks/file_sink.h>
#include <gnuradio/blocks/file_source.h>
#include <gnuradio/blocks/float_to_complex.h>
#include <gnuradio/blocks/multiply_const_ff.h>
#include <gnuradio/filter/fir_filter_ccc.h>
#include <gnuradio/top_block.h>
#include <rfid/gate.h>
#include <rfid/reader.h>
#include <rfid/tag_decoder.h>

#include <complex>
#include <iostream>
#include <string>
#include <vector>

int main(int argc, char **argv) {
  if (argc != 3) { std::cerr << "usage: ref_harness TRACE_FILE OUT_PREFIX\n"; return 2; }
  const std::string prefix = argv[2];
  // variables of apps/reader.py:52-65
  const double dac_rate = 1e6, adc_rate = 100e6 / 50;
  const int decim = 5;
  const std::vector<gr_complex> num_taps(25, gr_complex(1.0f, 0.0f));
  gr::top_block_sptr tb = gr::make_top_block("reader");
  // blocks of apps/reader.py:75-78, :102-109
  gr::filter::fir_filter_ccc::sptr matched_filter = gr::filter::fir_filter_ccc::make(decim, num_taps);
  gr::rfid::gate::sptr gate = gr::rfid::gate::make(int(adc_rate / decim));
  gr::rfid::tag_decoder::sptr tag_decoder = gr::rfid::tag_decoder::make(int(adc_rate / decim));
  gr::rfid::reader::sptr reader = gr::rfid::reader::make(int(adc_rate / decim), int(dac_rate));
  gr::blocks::multiply_const_ff::sptr amp = gr::blocks::multiply_const_ff::make(0.0f);
  gr::blocks::float_to_complex::sptr to_complex = gr::blocks::float_to_complex::make();
  gr::blocks::file_source::sptr file_source = gr::blocks::file_source::make(sizeof(gr_complex), argv[1], false);
  gr::blocks::file_sink::sptr sink_tx = gr::blocks::file_sink::make(sizeof(gr_complex), (prefix + ".reader").c_str());
  gr::blocks::file_sink::sptr sink_gate = gr::blocks::file_sink::make(sizeof(gr_complex), (prefix + ".gate").c_str());
  gr::blocks::file_sink::sptr sink_mf = gr::blocks::file_sink::make(sizeof(gr_complex), (prefix + ".mf").c_str());
  gr::blocks::file_sink::sptr sink_dec = gr::blocks::file_sink::make(sizeof(gr_complex), (prefix + ".decoder").c_str());
  sink_tx->set_unbuffered(true); sink_gate->set_unbuffered(true); sink_mf->set_unbuffered(true);
  tb->connect(file_source, 0, matched_filter, 0);
  tb->connect(matched_filter, 0, gate, 0);
  tb->connect(gate, 0, tag_decoder, 0);
  tb->connect(tag_decoder, 0, reader, 0);
  tb->connect(reader, 0, amp, 0);
  tb->connect(amp, 0, to_complex, 0);
  tb->connect(to_complex, 0, sink_tx, 0);
  tb->connect(gate, 0, sink_gate, 0);
  tb->connect(matched_filter, 0, sink_mf, 0);
  tb->connect(tag_decoder, 1, sink_dec, 0);   // (apps/reader.py:116: the decoder's second port must be connected)
  tb->run();   // (run it with GR_SCHEDULER=STS, as README.md:40 says: the blocks share reader_state without locks)
  reader->print_results();
  return 0;
}
