// rfid_reader_offline -- the offline (DEBUG = True) flowgraph of gr-rfid/apps/reader.py:101-112 in C++:
//   file_source(misc/data/file_source_test) -> matched_filter -> gate -> tag_decoder -> reader
// driven buffer by buffer through the block adaptors of rfid_blocks.hpp (one C-ABI call per
// general_work), then reader.print_results() (apps/reader.py:130; lib/reader_impl.cc:173-192).
//
//   rfid_reader_offline TRACE_FILE [--device N] [--chunk N] [--fixed-q Q] [--max-queries N] [--unique-tags N]
//                       [--tx-out FILE]   (the reader block's output, float32: apps/reader.py's file_sink_reader)
//
// TRACE_FILE: headerless little-endian interleaved float32 I,Q at 2 Msps (apps/reader.py:102).
// Exit codes: 0 ok, 2 usage / file error, 3 no gfx950 device (there is no CPU fallback), 4 other library error.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "rfid_blocks.hpp"

int main(int argc, char **argv) {
  const char *path = nullptr, *tx_path = nullptr;
  int device = 0, chunk = 8192;
  rfid_params p;
  rfid_params_default(&p);
  for (int i = 1; i < argc; ++i) {
    auto need = [&](const char *flag) -> int {
      if (i + 1 >= argc) { std::cerr << flag << " needs a value\n"; std::exit(2); }
      return std::atoi(argv[++i]);
    };
    if (!std::strcmp(argv[i], "--device")) device = need("--device");
    else if (!std::strcmp(argv[i], "--chunk")) chunk = need("--chunk");
    else if (!std::strcmp(argv[i], "--fixed-q")) p.fixed_q = need("--fixed-q");
    else if (!std::strcmp(argv[i], "--max-queries")) p.max_num_queries = need("--max-queries");
    else if (!std::strcmp(argv[i], "--unique-tags")) p.number_unique_tags = need("--unique-tags");
    else if (!std::strcmp(argv[i], "--tx-out")) { if (i + 1 >= argc) return 2; tx_path = argv[++i]; }
    else if (argv[i][0] == '-') { std::cerr << "unknown option " << argv[i] << "\n"; return 2; }
    else path = argv[i];
  }
  if (!path || chunk < 64) {
    std::cerr << "usage: rfid_reader_offline TRACE_FILE [--device N] [--chunk N] [--fixed-q Q] [--max-queries N] [--unique-tags N] [--tx-out FILE]\n";
    return 2;
  }
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::cerr << "cannot open " << path << "\n"; return 2; }
  const std::streamsize bytes = f.tellg();
  f.seekg(0);
  std::vector<gr_complex> samples((size_t)(bytes / (std::streamsize)sizeof(gr_complex)));
  if (!samples.empty() && !f.read(reinterpret_cast<char *>(samples.data()), (std::streamsize)(samples.size() * sizeof(gr_complex)))) {
    std::cerr << "short read on " << path << "\n";
    return 2;
  }

  try {
    // variables of apps/reader.py:52-65
    const double dac_rate = 1e6, adc_rate = 100e6 / 50;
    const int decim = 5;
    const std::vector<gr_complex> num_taps(25, gr_complex(1.0f, 0.0f));
    const int rate = (int)(adc_rate / decim);
    // construction order of apps/reader.py:75-78: the gate owns the shared reader state
    blocks::gate::sptr gate = blocks::gate::make(rate, device, &p);
    blocks::matched_filter::sptr mf = blocks::matched_filter::make(decim, num_taps, gate->context());
    blocks::tag_decoder::sptr dec = blocks::tag_decoder::make(rate, gate->context());
    blocks::reader::sptr reader = blocks::reader::make(rate, (int)dac_rate, gate->context());
    rfid_rt::sts_scheduler tb(mf, gate, dec, reader, chunk);
    tb.keep_tx(tx_path != nullptr);
    tb.run(samples.data(), samples.size());
    reader->print_results();
    if (tx_path) {
      std::ofstream o(tx_path, std::ios::binary);
      const std::vector<float> &tx = tb.tx_samples();
      o.write(reinterpret_cast<const char *>(tx.data()), (std::streamsize)(tx.size() * sizeof(float)));
      if (!o) { std::cerr << "cannot write " << tx_path << "\n"; return 2; }
    }
  } catch (const rfid_rt::error &e) {
    std::cerr << "rfid_reader_offline: " << e.what() << "\n";
    return e.status == RFID_ERR_NO_DEVICE ? 3 : 4;
  }
  return 0;
}
