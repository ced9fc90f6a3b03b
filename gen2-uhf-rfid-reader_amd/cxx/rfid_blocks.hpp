// rfid_blocks.hpp -- C++ host layer above the C-ABI (include/rfid_mi355x.h).
//
// The reference's receive blocks are compiled C++ (gr-rfid/lib/*_impl.cc) scheduled by GNU Radio
// 3.7, which is not available here.  This header mirrors their interface -- the same factory
// names, the same forecast()/general_work() signatures and the same consume_each()/produce()
// conventions -- on a minimal block base of our own (rfid_rt::block), so that
//   * the call-per-buffer form of the path can be driven from C++ exactly as the GNU Radio
//     scheduler drives it (rfid_rt::sts_scheduler = the README's GR_SCHEDULER=STS mode), and
//   * a maintainer with a real GNU Radio can copy the general_work() bodies into gr::block
//     subclasses unchanged (INTEGRATION.md shows that binding).
//
//   blocks::gate::make(int sample_rate)               <- gr::rfid::gate::make          include/rfid/gate.h:51
//   blocks::tag_decoder::make(int sample_rate)        <- gr::rfid::tag_decoder::make   include/rfid/tag_decoder.h:48
//   blocks::reader::make(int sample_rate, int dac)    <- gr::rfid::reader::make        include/rfid/reader.h:42,51
//   blocks::matched_filter::make(int decim, taps)     <- filter.fir_filter_ccc         apps/reader.py:65,75
//
// All sample arithmetic happens in the HIP kernels behind the C-ABI; there is no CPU path here.
#pragma once

#include <complex>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rfid_mi355x.h"

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace rfid_rt {

// what a gr::block exposes to a scheduler, reduced to what the receive path uses
class block {
 public:
  enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
  explicit block(const std::string &name) : d_name(name) {}
  virtual ~block() {}
  const std::string &name() const { return d_name; }
  virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required) = 0;
  virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items) = 0;
  // results of the last general_work(), as the scheduler reads them
  int consumed() const { return d_consumed; }
  int produced(int port) const { return port < (int)d_produced.size() ? d_produced[(size_t)port] : 0; }

 protected:
  void begin_work(int nports) { d_consumed = 0; d_produced.assign((size_t)nports, 0); }
  void consume_each(int n) { d_consumed = n; }
  void produce(int port, int n) { d_produced[(size_t)port] = n; }

 private:
  std::string d_name;
  int d_consumed = 0;
  std::vector<int> d_produced;
};

struct error : std::runtime_error {
  int status;
  error(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};

// One RX stream = one rfid_ctx shared by the blocks of a flowgraph (the reference shares the global
// `reader_state`, include/rfid/global_vars.h:146; the gate allocates it, lib/gate_impl.cc:67-69).
class stream_context {
 public:
  explicit stream_context(int device = 0, const rfid_params *params = nullptr) {
    rfid_params p;
    rfid_params_default(&p);
    if (params) p = *params;
    d_params = p;
    const int st = rfid_ctx_create(&p, device, &d_ctx);
    if (st != RFID_OK) throw error(st, std::string("rfid_ctx_create: ") + rfid_strerror(st));
  }
  ~stream_context() { if (d_ctx) rfid_ctx_destroy(d_ctx); }
  stream_context(const stream_context &) = delete;
  stream_context &operator=(const stream_context &) = delete;
  rfid_ctx *get() const { return d_ctx; }
  const rfid_params &params() const { return d_params; }
  void check(int st, const char *what) const {
    if (st != RFID_OK)
      throw error(st, std::string(what) + ": " + rfid_strerror(st) + " (" + rfid_last_error(d_ctx) + ")");
  }
  rfid_reader_state state() const {
    rfid_reader_state s;
    check(rfid_get_state(d_ctx, &s), "rfid_get_state");
    return s;
  }

 private:
  rfid_ctx *d_ctx = nullptr;
  rfid_params d_params;
};
typedef std::shared_ptr<stream_context> context_sptr;

}  // namespace rfid_rt

namespace blocks {

using rfid_rt::context_sptr;

// fir_filter_ccc(5, [1]*25) (apps/reader.py:65,75): complex in -> complex out, decimating
class matched_filter : public rfid_rt::block {
 public:
  typedef std::shared_ptr<matched_filter> sptr;
  static sptr make(int decim, const std::vector<gr_complex> &taps, context_sptr ctx) {
    if (decim != 5 || taps.size() != 25) throw rfid_rt::error(RFID_ERR_UNSUPPORTED, "only fir_filter_ccc(5,[1]*25) is built");
    for (const gr_complex &t : taps)
      if (t != gr_complex(1.0f, 0.0f)) throw rfid_rt::error(RFID_ERR_UNSUPPORTED, "only all-ones taps are built");
    return sptr(new matched_filter(ctx));
  }
  void forecast(int noutput_items, gr_vector_int &req) override { req.assign(1, noutput_items * 5 + 24); }
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    begin_work(1);
    int n_out = 0;
    d_ctx->check(rfid_mf_work(d_ctx->get(), (const rfid_cf32 *)input_items[0], ninput_items[0],
                              (rfid_cf32 *)output_items[0], noutput_items, &n_out), "rfid_mf_work");
    consume_each(ninput_items[0]);
    return n_out;
  }

 private:
  explicit matched_filter(context_sptr ctx) : block("matched_filter"), d_ctx(ctx) {}
  context_sptr d_ctx;
};

// gr::rfid::gate (include/rfid/gate.h:51; lib/gate_impl.cc:41-200): complex in -> complex out
class gate : public rfid_rt::block {
 public:
  typedef std::shared_ptr<gate> sptr;
  // Constructed first, as in apps/reader.py:76: it creates the stream's shared state.
  static sptr make(int sample_rate, int device = 0, const rfid_params *params = nullptr) {
    rfid_params p;
    rfid_params_default(&p);
    if (params) p = *params;
    p.sample_rate = sample_rate;
    return sptr(new gate(std::make_shared<rfid_rt::stream_context>(device, &p)));
  }
  context_sptr context() const { return d_ctx; }
  void forecast(int noutput_items, gr_vector_int &req) override { req.assign(1, noutput_items); }   // gate_impl.cc:79-83
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    begin_work(1);
    int consumed = 0, written = 0;
    int n_in = ninput_items[0] < noutput_items ? ninput_items[0] : noutput_items;   // n_items, gate_impl.cc:91
    d_ctx->check(rfid_gate_work(d_ctx->get(), (const rfid_cf32 *)input_items[0], n_in, (rfid_cf32 *)output_items[0],
                                noutput_items, &consumed, &written), "rfid_gate_work");
    consume_each(consumed);   // gate_impl.cc:198
    return written;           // gate_impl.cc:199
  }

 private:
  explicit gate(context_sptr ctx) : block("gate"), d_ctx(ctx) {}
  context_sptr d_ctx;
};

// gr::rfid::tag_decoder (include/rfid/tag_decoder.h:48; lib/tag_decoder_impl.cc:35-397):
// complex in -> {float (port 0), complex debug (port 1, never produced)}
class tag_decoder : public rfid_rt::block {
 public:
  typedef std::shared_ptr<tag_decoder> sptr;
  static sptr make(int sample_rate, context_sptr ctx) {
    if (sample_rate != ctx->params().sample_rate) throw rfid_rt::error(RFID_ERR_INVALID, "tag_decoder: sample_rate differs from the gate's");
    return sptr(new tag_decoder(ctx));
  }
  void forecast(int noutput_items, gr_vector_int &req) override { req.assign(1, noutput_items); }   // tag_decoder_impl.cc:72-76
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    begin_work(2);
    int consumed = 0, produced0 = 0;
    d_ctx->check(rfid_decoder_work(d_ctx->get(), (const rfid_cf32 *)input_items[0], ninput_items[0],
                                   (float *)output_items[0], noutput_items, &consumed, &produced0, &d_last, nullptr),
                 "rfid_decoder_work");
    d_last_valid = consumed > 0;
    produce(0, produced0);      // tag_decoder_impl.cc:266
    consume_each(consumed);     // :395
    return WORK_CALLED_PRODUCE; // :396
  }
  // details of the window decoded by the last call (not part of the reference's interface)
  bool last_result(rfid_decode_result *out) const { if (d_last_valid && out) *out = d_last; return d_last_valid; }

 private:
  explicit tag_decoder(context_sptr ctx) : block("tag_decoder"), d_ctx(ctx) {}
  context_sptr d_ctx;
  rfid_decode_result d_last{};
  bool d_last_valid = false;
};

// gr::rfid::reader (include/rfid/reader.h:42,51; lib/reader_impl.cc:43-380): float in -> float out: the Gen2 state
// transitions and the transmit waveform of the state at hand (Query / ACK / QueryRep / NAK / carrier).
class reader : public rfid_rt::block {
 public:
  typedef std::shared_ptr<reader> sptr;
  static sptr make(int sample_rate, int dac_rate, context_sptr ctx) { (void)sample_rate; return sptr(new reader(ctx, dac_rate)); }
  void forecast(int, gr_vector_int &req) override { req.assign(1, 0); }   // reader_impl.cc:194-198
  // the largest number of items one call writes (the scheduler's output buffer must hold it)
  int max_output() const { return rfid_reader_tx_max(d_dac_rate); }
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    begin_work(1);
    int consumed = 0, written = 0;
    float *out = output_items.empty() ? nullptr : (float *)output_items[0];
    const float *in = input_items.empty() ? nullptr : (const float *)input_items[0];
    if (out)
      d_ctx->check(rfid_reader_work_tx(d_ctx->get(), d_dac_rate, in, ninput_items[0], out, noutput_items, &consumed, &written),
                   "rfid_reader_work_tx");
    else   // no output buffer connected: transitions only
      d_ctx->check(rfid_reader_work(d_ctx->get(), ninput_items[0], &consumed), "rfid_reader_work");
    consume_each(consumed);   // reader_impl.cc:378
    return written;           // :379
  }
  void print_results() {      // reader_impl.cc:173-192
    std::vector<char> buf(1 << 15);
    int len = 0;
    d_ctx->check(rfid_print_results(d_ctx->get(), buf.data(), (int)buf.size(), &len), "rfid_print_results");
    fwrite(buf.data(), 1, (size_t)len, stdout);
  }
  int dac_rate() const { return d_dac_rate; }

 private:
  reader(context_sptr ctx, int dac_rate) : block("reader"), d_ctx(ctx), d_dac_rate(dac_rate) {}
  context_sptr d_ctx;
  int d_dac_rate;
};

}  // namespace blocks

namespace rfid_rt {

// Single-threaded scheduler for the DEBUG=True topology of apps/reader.py:101-112
//   file_source -> matched_filter -> gate -> tag_decoder -> reader
// (README.md:40: GR_SCHEDULER=STS).  Buffers between blocks are plain vectors; every block is
// called with what is available, as GNU Radio does, until nothing moves any more.
class sts_scheduler {
 public:
  sts_scheduler(blocks::matched_filter::sptr mf, blocks::gate::sptr g, blocks::tag_decoder::sptr d, blocks::reader::sptr r,
                int chunk = 8192)
      : d_mf(mf), d_gate(g), d_dec(d), d_reader(r), d_ctx(g->context()), d_chunk(chunk) {}

  long windows_decoded() const { return d_windows; }
  // everything the reader block wrote (apps/reader.py:110-112 connects it to a file_sink in DEBUG mode)
  const std::vector<float> &tx_samples() const { return d_tx; }
  void keep_tx(bool on) { d_keep_tx = on; }

  void run(const gr_complex *samples, size_t n) {
    reader_until_idle(0);   // START -> SEND_QUERY -> IDLE
    std::vector<gr_complex> gq, dq, mf_out((size_t)d_chunk + 8), gate_out((size_t)d_chunk);
    std::vector<float> bits(16);
    d_bits = &bits;
    d_txbuf.assign((size_t)d_reader->max_output(), 0.0f);
    size_t pos = 0;
    while (pos < n || !gq.empty()) {
      if (pos < n) {
        const size_t take = (n - pos < (size_t)d_chunk * 5) ? (n - pos) : (size_t)d_chunk * 5;
        gr_vector_int nin(1, (int)take);
        gr_vector_const_void_star in(1, samples + pos);
        gr_vector_void_star out(1, mf_out.data());
        const int produced = d_mf->general_work((int)mf_out.size(), nin, in, out);
        gq.insert(gq.end(), mf_out.begin(), mf_out.begin() + produced);
        pos += take;
      }
      while (!gq.empty()) {
        const int avail = (int)(gq.size() < (size_t)d_chunk ? gq.size() : (size_t)d_chunk);
        gr_vector_int nin(1, avail);
        gr_vector_const_void_star in(1, gq.data());
        gr_vector_void_star out(1, gate_out.data());
        const int written = d_gate->general_work(avail, nin, in, out);
        const int consumed = d_gate->consumed();
        gq.erase(gq.begin(), gq.begin() + consumed);
        dq.insert(dq.end(), gate_out.begin(), gate_out.begin() + written);
        for (;;) {
          gr_vector_int dn(1, (int)dq.size());
          gr_vector_const_void_star din(1, dq.data());
          gr_vector_void_star dout(2, nullptr);
          dout[0] = bits.data();
          d_dec->general_work((int)bits.size(), dn, din, dout);
          const int dcons = d_dec->consumed();
          if (dcons == 0) break;
          d_windows++;
          dq.erase(dq.begin(), dq.begin() + dcons);
          reader_until_idle(d_dec->produced(0));
        }
        if (consumed == 0) break;
      }
    }
  }

 private:
  void reader_until_idle(int q) {
    for (int it = 0; it < 8; ++it) {
      const int before = d_ctx->state().gen2_logic_status;
      if (before == RFID_IDLE) break;
      if (d_txbuf.empty()) d_txbuf.assign((size_t)d_reader->max_output(), 0.0f);
      gr_vector_int nin(1, q);
      gr_vector_const_void_star in(1, d_bits ? (const void *)d_bits->data() : nullptr);
      gr_vector_void_star out(1, d_txbuf.data());
      const int written = d_reader->general_work((int)d_txbuf.size(), nin, in, out);
      if (d_keep_tx) d_tx.insert(d_tx.end(), d_txbuf.begin(), d_txbuf.begin() + written);
      q = 0;
      if (d_ctx->state().gen2_logic_status == before) break;
    }
  }
  blocks::matched_filter::sptr d_mf;
  blocks::gate::sptr d_gate;
  blocks::tag_decoder::sptr d_dec;
  blocks::reader::sptr d_reader;
  context_sptr d_ctx;
  int d_chunk;
  long d_windows = 0;
  std::vector<float> d_tx, d_txbuf;
  const std::vector<float> *d_bits = nullptr;
  bool d_keep_tx = false;
};

}  // namespace rfid_rt
