// rfid_reader_offline -- the offline (DEBUG = True) flowgraph of gr-rfid/apps/reader.py:101-112 in C++, written
// against the reference's own block API (cxx/include/rfid/{gate,tag_decoder,reader,global_vars}.h = the classes,
// factories and the `reader_state` global of gr-rfid/include/rfid/*.h):
//   file_source(misc/data/file_source_test) -> matched_filter -> gate -> tag_decoder -> reader ; reader.print_results()
// Blocks are made in the order and with the arguments of apps/reader.py:75-78 -- gate::make(int),
// tag_decoder::make(int), reader::make(int,int), nothing else -- and driven buffer by buffer (one C-ABI call per
// general_work) by the single-threaded scheduler of rfid/mi355x.h.
//
//   rfid_reader_offline TRACE_FILE [--device N] [--chunk N] [--fixed-q Q] [--max-queries N] [--unique-tags N]
//                       [--tx-out FILE]    the reader block's output, float32 (apps/reader.py's file_sink_reader)
//                       [--mf-out FILE]    matched-filter output, complex64     (file_sink_matched_filter, :69)
//                       [--gate-out FILE]  gated samples, complex64            (file_sink_gate, :70)
//                       [--whole-chain N]  instead of block-by-block calls: rfid_stream_work, N raw samples per call
//                                          (matched filter -> gate -> tag_decoder in one submission, state on the device)
//                       [--host-fir]       apps/reader.py as it stands: the matched filter is NOT a block of this library (there
//                                          it is GNU Radio's filter.fir_filter_ccc, apps/reader.py:75) -- here a plain host loop
//                                          over the whole file, y[n] = sum_{k=0..24} x[5n-24+k], k ascending -- and only
//                                          gate, tag_decoder and reader are made; the gate is fed the filter's output
//                                          buffer by buffer
//                       [--scheduler sts|bounded]  sts (default): the single-threaded scheduler of rfid/mi355x.h, queues between the
//                                          blocks as long as they need to be, --chunk items read per turn; bounded: GNU Radio's
//                                          scheduling rules (bounded buffers of --buffer items, forecast, a block that could do nothing
//                                          left alone until new input arrives or its neighbour is done, stop() at the end and nothing
//                                          else: mi355x::bounded_flowgraph)
//                       [--buffer N]       items per buffer of the bounded scheduler (default 8192: 64 KB of gr_complex)
//                       [--time]           print the run's wall time and rate to stderr
//
// TRACE_FILE: headerless little-endian interleaved float32 I,Q at 2 Msps (apps/reader.py:102).
// Exit codes: 0 ok, 2 usage / file error, 3 no gfx950 device (there is no CPU fallback), 4 other library error.
#include <rfid/mi355x.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

namespace {
template <class T>
bool dump(const char *path, const std::vector<T> &v) {
  std::ofstream o(path, std::ios::binary);
  o.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
  return (bool)o;
}
}  // namespace

int main(int argc, char **argv) {
  const char *path = nullptr, *tx_path = nullptr, *mf_path = nullptr, *gate_path = nullptr;
  int device = 0, chunk = 8192;
  long whole_chain = 0;
  bool show_time = false, host_fir = false, bounded = false, pageable = false;
  int buffer_items = 8192;
  int fixed_q = gr::rfid::FIXED_Q, max_q = gr::rfid::MAX_NUM_QUERIES, uniq = gr::rfid::NUMBER_UNIQUE_TAGS;
  for (int i = 1; i < argc; ++i) {
    auto need = [&](const char *flag) -> const char * {
      if (i + 1 >= argc) { std::cerr << flag << " needs a value\n"; std::exit(2); }
      return argv[++i];
    };
    if (!std::strcmp(argv[i], "--device")) device = std::atoi(need("--device"));
    else if (!std::strcmp(argv[i], "--chunk")) chunk = std::atoi(need("--chunk"));
    else if (!std::strcmp(argv[i], "--fixed-q")) fixed_q = std::atoi(need("--fixed-q"));
    else if (!std::strcmp(argv[i], "--max-queries")) max_q = std::atoi(need("--max-queries"));
    else if (!std::strcmp(argv[i], "--unique-tags")) uniq = std::atoi(need("--unique-tags"));
    else if (!std::strcmp(argv[i], "--tx-out")) tx_path = need("--tx-out");
    else if (!std::strcmp(argv[i], "--mf-out")) mf_path = need("--mf-out");
    else if (!std::strcmp(argv[i], "--gate-out")) gate_path = need("--gate-out");
    else if (!std::strcmp(argv[i], "--whole-chain")) whole_chain = std::atol(need("--whole-chain"));
    else if (!std::strcmp(argv[i], "--time")) show_time = true;
    else if (!std::strcmp(argv[i], "--host-fir")) host_fir = true;
    else if (!std::strcmp(argv[i], "--pageable")) pageable = true;   // the trace in ordinary memory (what a GNU Radio buffer is)
    else if (!std::strcmp(argv[i], "--scheduler")) { const char *v = need("--scheduler"); bounded = !std::strcmp(v, "bounded"); if (!bounded && std::strcmp(v, "sts")) { std::cerr << "--scheduler sts|bounded\n"; return 2; } }
    else if (!std::strcmp(argv[i], "--buffer")) buffer_items = std::atoi(need("--buffer"));
    else if (argv[i][0] == '-') { std::cerr << "unknown option " << argv[i] << "\n"; return 2; }
    else path = argv[i];
  }
  if (!path || chunk < 64) {
    std::cerr << "usage: rfid_reader_offline TRACE_FILE [--device N] [--chunk N] [--fixed-q Q] [--max-queries N] "
                 "[--unique-tags N] [--tx-out FILE] [--mf-out FILE] [--gate-out FILE]\n";
    return 2;
  }
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::cerr << "cannot open " << path << "\n"; return 2; }
  const std::streamsize bytes = f.tellg();
  f.seekg(0);
  // the trace goes into page-locked memory (what a file source of this library's own would hand out): the blocks' input
  // then reaches the device without a staging copy.  Ordinary memory when that fails.
  struct host_buf {
    gr_complex *p = nullptr; size_t n = 0; bool locked = false;
    ~host_buf() { if (locked) rfid_host_free(p); else delete[] p; }
    gr_complex *data() { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
  } samples;
  samples.n = (size_t)(bytes / (std::streamsize)sizeof(gr_complex));
  samples.p = pageable ? nullptr : static_cast<gr_complex *>(rfid_host_alloc(samples.n * sizeof(gr_complex)));
  samples.locked = samples.p != nullptr;
  if (!samples.p) samples.p = new gr_complex[samples.n ? samples.n : 1];
  if (!samples.empty() && !f.read(reinterpret_cast<char *>(samples.data()), (std::streamsize)(samples.size() * sizeof(gr_complex)))) {
    std::cerr << "short read on " << path << "\n";
    return 2;
  }

  try {
    using namespace gr::rfid;
    mi355x::configure(device, fixed_q, max_q, uniq);   // what the reference fixes at compile time (global_vars.h:72,76,100)
    // variables of apps/reader.py:52-65
    const double dac_rate = 1e6, adc_rate = 100e6 / 50;
    const int decim = 5;
    const std::vector<gr_complex> num_taps(25, gr_complex(1.0f, 0.0f));
    // blocks of apps/reader.py:75-78, same order, same arguments (the gate owns the shared reader state)
    matched_filter::sptr mf;
    std::vector<gr_complex> y_host;
    double fir_secs = 0.0;
    if (host_fir) {
      // somebody else's filter (apps/reader.py:75: filter.fir_filter_ccc(decim, num_taps)): history of 24 zeros, one output
      // per complete group of 5 inputs, taps all one -- summed in tap order
      const auto f0 = std::chrono::steady_clock::now();
      const size_t n_out = samples.size() / (size_t)decim;
      y_host.resize(n_out);
      const gr_complex *x = samples.data();
      for (size_t n = 0; n < n_out; ++n) {
        float re = 0.0f, im = 0.0f;
        const long first = (long)(5 * n) - 24;
        for (long k = first < 0 ? -first : 0; k < 25; ++k) { re = re + x[first + k].real(); im = im + x[first + k].imag(); }
        y_host[n] = gr_complex(re, im);
      }
      fir_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - f0).count();
    } else {
      mf = matched_filter::make(decim, num_taps);   // (built before the gate, as at :75)
    }
    const auto c0 = std::chrono::steady_clock::now();
    gate::sptr gate_blk = gate::make(int(adc_rate / decim));
    const double make_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
    tag_decoder::sptr dec = tag_decoder::make(int(adc_rate / decim));
    reader::sptr reader_blk = reader::make(int(adc_rate / decim), int(dac_rate));
    mi355x::sts_flowgraph tb(mf, gate_blk, dec, reader_blk, chunk);
    mi355x::bounded_flowgraph tbb(mf, gate_blk, dec, reader_blk, bounded ? buffer_items : 0);
    if (!bounded) { if (mf) mf->minirt_set_buffers(0, 0); gate_blk->minirt_set_buffers(0, 0); dec->minirt_set_buffers(0, 0); reader_blk->minirt_set_buffers(0, 0); }
    tb.keep_tx(tx_path != nullptr); tbb.keep_tx(tx_path != nullptr);
    tb.keep_taps(mf_path != nullptr || gate_path != nullptr); tbb.keep_taps(mf_path != nullptr || gate_path != nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    long n_windows = 0;
    if (whole_chain > 0) {
      // the same blocks' stream, fed chunk by chunk through the whole-chain call of the C-ABI
      rfid_ctx *ctx = mi355x::current_context();
      if ((size_t)whole_chain > samples.size()) whole_chain = (long)samples.size();   // (no staging for more than the file holds)
      if (whole_chain < 200000) whole_chain = 200000;
      int st = rfid_stream_begin(ctx, whole_chain);
      if (st != RFID_OK) throw mi355x::error(st, std::string("rfid_stream_begin: ") + rfid_last_error(ctx));
      std::vector<rfid_stream_window> w(1 << 16);
      std::vector<rfid_decode_result> r(1 << 16);
      size_t pos = 0;
      bool flushed = false;
      while (!flushed) {
        const size_t n = (samples.size() - pos < (size_t)whole_chain) ? samples.size() - pos : (size_t)whole_chain;
        flushed = (pos + n >= samples.size());
        int64_t got = 0;
        st = rfid_stream_work(ctx, (const rfid_cf32 *)samples.data() + pos, (int64_t)n, flushed ? 1 : 0, w.data(), r.data(),
                              (int64_t)w.size(), &got);
        if (st == RFID_ERR_CAPACITY && got > (int64_t)w.size()) {
          w.resize((size_t)got * 2); r.resize((size_t)got * 2);
          st = rfid_stream_work(ctx, nullptr, 0, flushed ? 1 : 0, w.data(), r.data(), (int64_t)w.size(), &got);
        }
        if (st != RFID_OK) throw mi355x::error(st, std::string("rfid_stream_work: ") + rfid_last_error(ctx));
        n_windows += (long)got;
        pos += n;
      }
      rfid_stream_end(ctx);
    } else if (bounded) {
      if (host_fir) tbb.run(y_host.data(), y_host.size()); else tbb.run(samples.data(), samples.size());
      n_windows = tbb.windows_decoded();
      if (tbb.stalled()) { std::cerr << "rfid_reader_offline: the bounded scheduler stopped with input left and no block able to move\n"; return 4; }
    } else if (host_fir) {
      tb.run(y_host.data(), y_host.size());
      n_windows = tb.windows_decoded();
    } else {
      tb.run(samples.data(), samples.size());
      n_windows = tb.windows_decoded();
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    reader_blk->print_results();   // apps/reader.py:130
    if (show_time)
      std::cerr << "rfid_reader_offline: gate::make (the context) " << make_secs * 1e3 << " ms; " << samples.size() << " raw samples, " << n_windows << " windows in " << secs * 1e3
                << " ms = " << (double)samples.size() / secs / 1e6 << " Msamples/s ("
                << (whole_chain > 0 ? "whole-chain calls" : host_fir ? "block-by-block calls, gate / tag_decoder / reader only; the host's own FIR loop took "
                                                                      + std::to_string(fir_secs * 1e3) + " ms before that"
                                                                    : std::string("block-by-block calls")) << ")\n";
    if (getenv("RFID_PRINT_READER_STATE"))   // the reference's global, for tests of the mirror
      std::cout << "reader_state: n_queries_sent=" << reader_state->reader_stats.n_queries_sent
                << " n_epc_correct=" << reader_state->reader_stats.n_epc_correct
                << " unique=" << reader_state->reader_stats.tag_reads.size()
                << " gate_status=" << reader_state->gate_status << " windows=" << n_windows << "\n";
    if (tx_path && !dump(tx_path, bounded ? tbb.tx_samples() : tb.tx_samples())) { std::cerr << "cannot write " << tx_path << "\n"; return 2; }
    if (mf_path && !dump(mf_path, bounded ? tbb.tap_matched_filter() : tb.tap_matched_filter())) { std::cerr << "cannot write " << mf_path << "\n"; return 2; }
    if (gate_path && !dump(gate_path, bounded ? tbb.tap_gate() : tb.tap_gate())) { std::cerr << "cannot write " << gate_path << "\n"; return 2; }
  } catch (const gr::rfid::mi355x::error &e) {
    std::cerr << "rfid_reader_offline: " << e.what() << "\n";
    return e.status == RFID_ERR_NO_DEVICE ? 3 : 4;
  }
  return 0;
}
