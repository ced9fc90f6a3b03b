// rfid_blocks.cc -- gr::rfid::{gate, tag_decoder, reader}::make and the shared reader_state of the reference's
// block API (cxx/include/rfid/*.h), implemented on the MI355X C-ABI (include/rfid_mi355x.h): every general_work()
// is ONE C-ABI call that hands the scheduler's buffer to the HIP kernels and translates the returned counts into
// consume_each() / produce() / the return value exactly as the reference's blocks do
// (lib/gate_impl.cc:79-83,198-199; lib/tag_decoder_impl.cc:72-76,266,395-396; lib/reader_impl.cc:194-198,378-379).
// No sample arithmetic happens here and there is no CPU fallback: gate::make() throws without a gfx950 device.
// The matched filter switches the library's look-ahead on (rfid_lookahead_enable): its general_work() then runs the
// whole chain for the buffer on the device and the gate / tag_decoder calls are answered from what that left on the
// host -- same counts, same bytes, ~6 device round trips per inventory slot less (RFID_LOOKAHEAD=0: off).
// A gate WITHOUT a matched_filter block of this library in front of it -- apps/reader.py:75 as it stands instantiates
// GNU Radio's own filter.fir_filter_ccc -- switches the look-ahead on keyed on its own input
// (rfid_lookahead_enable_gate): gate -> tag_decoder over every buffer it is shown, in one submission.
#include <rfid/mi355x.h>
#ifndef GR_RFID_MINIRT   // a real GNU Radio: what a block may ask its runtime about its buffers (input_buffer_items, upstream_done below)
#include <gnuradio/block_detail.h>
#include <gnuradio/buffer.h>
#endif

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace gr {
namespace rfid {

READER_STATE *reader_state = nullptr;   // include/rfid/global_vars.h:146 (allocated by the gate, lib/gate_impl.cc:67-69)

namespace {

// one RX stream: the C-ABI context + the READER_STATE mirror the reference's API exposes
struct stream {
  rfid_ctx *ctx = nullptr;
  rfid_params params;
  READER_STATE mirror;
  bool has_filter = false;   // a matched_filter block is bound to this stream
  bool consume_ahead = false;   // the gate takes everything it is shown, the windows follow (rfid_lookahead_set_consume_ahead)
  // one context, four blocks: under a thread-per-block runtime their calls come from four threads -- serialised here (a context
  // is not re-entrant).  The ORDER of the calls is the flowgraph's business: the reference shares READER_STATE between its
  // blocks without any lock and recommends the single-threaded scheduler (README.md:40, GR_SCHEDULER=STS).
  std::recursive_mutex mu;
  ~stream() { if (ctx) rfid_ctx_destroy(ctx); if (reader_state == &mirror) reader_state = nullptr; }
  void check(int st, const char *what) const {
    if (st != RFID_OK)
      throw mi355x::error(st, std::string(what) + ": " + rfid_strerror(st) + " (" + rfid_last_error(ctx) + ")");
  }
  void refresh() {   // READER_STATE <- rfid_reader_state
    rfid_reader_state s;
    check(rfid_get_state(ctx, &s), "rfid_get_state");
    mirror.status = (STATUS)s.status;
    mirror.gen2_logic_status = (GEN2_LOGIC_STATUS)s.gen2_logic_status;
    mirror.gate_status = (GATE_STATUS)s.gate_status;
    mirror.decoder_status = (DECODER_STATUS)s.decoder_status;
    mirror.n_samples_to_ungate = s.n_samples_to_ungate;
    READER_STATS &r = mirror.reader_stats;
    r.n_queries_sent = s.n_queries_sent; r.cur_inventory_round = s.cur_inventory_round;
    r.cur_slot_number = s.cur_slot_number; r.max_slot_number = s.max_slot_number;
    r.max_inventory_round = MAX_INVENTORY_ROUND;
    if (r.n_epc_correct != s.n_epc_correct || s.n_epc_correct == 0) {   // (tag_reads only changes with a correctly decoded EPC)
      r.tag_reads.clear();
      for (int id = 0; id < 256; ++id)
        if (s.tag_reads[id]) r.tag_reads[id] = s.tag_reads[id];
    }
    r.n_epc_correct = s.n_epc_correct;
  }
};
typedef std::shared_ptr<stream> stream_sptr;

struct next_config { int device = 0; rfid_params p; bool set = false; } g_next;
// Binding rule = the reference's construction order (apps/reader.py:75-78), per thread: tag_decoder / reader bind at
// construction to the stream of the most recent gate; a matched_filter binds to it too unless that stream already
// has one -- then, or when no gate exists yet (apps/reader.py:75 builds the filter first), it waits for the next gate.
thread_local stream_sptr g_current;
// (weak: a filter that dies before its gate appears -- on whatever thread -- simply drops out)
thread_local std::vector<std::weak_ptr<stream_sptr>> g_pending_filters;

int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e && *e) ? atoi(e) : dflt;
}

// items the scheduler's buffer on a block's input / output side holds (0: unknown or not bounded)
#ifdef GR_RFID_MINIRT
int input_buffer_items(gr::block *b) { return b->minirt_input_capacity(); }
int output_buffer_items(gr::block *b) { return b->minirt_output_capacity(); }
#else
int input_buffer_items(gr::block *b) { return b->detail() ? b->detail()->input(0)->max_possible_items_available() : 0; }
int output_buffer_items(gr::block *b) { return b->detail() ? b->detail()->output(0)->bufsize() : 0; }
#endif
// the block's upstream neighbour has finished: what its input buffer holds is all there will be
#ifdef GR_RFID_MINIRT
bool upstream_done(const gr::block *b) { return b->minirt_input_done(); }
#else
bool upstream_done(const gr::block *b) { return b->detail() && b->detail()->input(0)->done(); }
#endif
// The look-ahead lets 65 536 decimated samples gather before a pass is submitted; a gate that can never be shown that many
// (its input buffer holds C items) CONSUMES AHEAD (rfid_lookahead_set_consume_ahead: it takes everything it is shown -- the
// device has it -- and hands out the windows when the passes have found them; -> true) or, with RFID_GATE_CONSUME_AHEAD=0, is
// told C / 4 (rfid_lookahead_set_scheduler: a gate call that can decide nothing decides at once, the pipeline drains)
bool tell_buffers(rfid_ctx *ctx, int gate_input_items) {
  if (gate_input_items <= 0) return false;
  if (env_int("RFID_GATE_CONSUME_AHEAD", 1) != 0 && rfid_lookahead_set_consume_ahead(ctx, 1) == RFID_OK) return true;
  (void)rfid_lookahead_set_scheduler(ctx, gate_input_items);
  return false;
}


stream_sptr current_or_throw(const char *who) {
  if (!g_current)
    throw mi355x::error(RFID_ERR_STATE, std::string(who) + ": construct gr::rfid::gate first -- it owns the shared reader "
                                        "state (lib/gate_impl.cc:67-69, apps/reader.py:76-78)");
  return g_current;
}

// ---- gate ------------------------------------------------------------------------------------------
class gate_impl : public gate {
 public:
  explicit gate_impl(int sample_rate)
      : gr::block("gate", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))) {
    stream_sptr s(new stream());
    rfid_params_default(&s->params);
    int device = env_int("RFID_DEVICE", 0);
    s->params.fixed_q = env_int("RFID_FIXED_Q", FIXED_Q);
    s->params.max_num_queries = env_int("RFID_MAX_NUM_QUERIES", MAX_NUM_QUERIES);
    s->params.number_unique_tags = env_int("RFID_NUMBER_UNIQUE_TAGS", NUMBER_UNIQUE_TAGS);
    if (g_next.set) { device = g_next.device; s->params = g_next.p; }
    s->params.sample_rate = sample_rate;
    const int st = rfid_ctx_create(&s->params, device, &s->ctx);
    if (st != RFID_OK)
      throw mi355x::error(st, std::string("gr::rfid::gate: rfid_ctx_create: ") + rfid_strerror(st) +
                                  " (the MI355X receive path has no CPU fallback)");
    d_stream = s;
    g_current = s;
    while (!g_pending_filters.empty()) {
      std::shared_ptr<stream_sptr> slot = g_pending_filters.front().lock();
      g_pending_filters.erase(g_pending_filters.begin());
      if (!slot) continue;             // that filter is gone
      *slot = s;
      s->has_filter = true;
      break;
    }
    gettimeofday(&s->mirror.reader_stats.start, nullptr);
    initialize_reader_state();   // lib/gate_impl.cc:69
  }
  void forecast(int noutput_items, gr_vector_int &ninput_items_required) override {
    int needs_input = 1;   // (consume-ahead: not while a window lies ready, or the input has ended and samples are undecided)
    std::lock_guard<std::recursive_mutex> lk(d_stream->mu);
    if (d_stream->consume_ahead) (void)rfid_gate_forecast(d_stream->ctx, upstream_done(this) ? 1 : 0, &needs_input);
    ninput_items_required[0] = needs_input ? noutput_items : 0;   // lib/gate_impl.cc:79-83
  }
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    std::lock_guard<std::recursive_mutex> lk(d_stream->mu);
    int n_items = std::min(ninput_items[0], noutput_items);   // lib/gate_impl.cc:91
    if (!d_started) {
      d_started = true;
      // no matched_filter block of this library feeds this gate (it would have switched the look-ahead on itself): the
      // filter is somebody else's, the look-ahead is keyed on the gate's input
      if (!d_stream->has_filter && env_int("RFID_LOOKAHEAD", 1) != 0) {
        d_stream->check(rfid_lookahead_enable_gate(d_stream->ctx, std::max(n_items, 16384)), "rfid_lookahead_enable_gate");
        d_stream->consume_ahead = tell_buffers(d_stream->ctx, input_buffer_items(this));
      }
    }
    if (d_stream->consume_ahead) {
      n_items = ninput_items[0];   // everything: what the gate writes does not depend on how much it is shown
      // the end of the input: nobody announces it, but a block can see it -- everything the device holds is decided now
      if (n_items == 0 && !d_told_end && upstream_done(this)) {
        d_told_end = true;
        d_stream->check(rfid_lookahead_flush(d_stream->ctx), "rfid_lookahead_flush");
      }
    }
    const GATE_STATUS before = d_stream->mirror.gate_status;
    int consumed = 0, written = 0;
    d_stream->check(rfid_gate_work(d_stream->ctx, (const rfid_cf32 *)input_items[0], n_items, (rfid_cf32 *)output_items[0],
                                   noutput_items, &consumed, &written), "rfid_gate_work");
    d_stream->refresh();
    if (written > 0) {   // magn_squared_samples side channel (lib/gate_impl.cc:171,175,186)
      std::vector<float> &m2 = d_stream->mirror.magn_squared_samples;
      if (before != GATE_OPEN) m2.clear();
      const size_t at = m2.size();
      m2.resize(at + (size_t)written);
      int got = 0;
      // formed on the device together with the gated samples (look-ahead); without the look-ahead the per-call gate path
      // returns the samples only and the mirror is filled here with the reference's own expression
      d_stream->check(rfid_gate_magn_squared(d_stream->ctx, m2.data() + at, written, &got), "rfid_gate_magn_squared");
      if (got != written) {
        const gr_complex *o = (const gr_complex *)output_items[0];
        for (int i = 0; i < written; ++i) m2[at + (size_t)i] = std::norm(o[i]);
      }
    }
    consume_each(consumed);   // lib/gate_impl.cc:198
    return written;           // :199
  }
  // The flowgraph has stopped (apps/reader.py:131).  Under a scheduler that went on calling the blocks until none could
  // move there is nothing left to do: every sample was consumed (the gate's second fruitless call decides what is held
  // back, see rfid_lookahead_enable).  A scheduler that gave up earlier leaves windows in the library that no call will
  // fetch any more: they are decided and accounted here (READER_STATE as the decoder / reader calls would have left it), so
  // that print_results() -- called after stop(), apps/reader.py:130-131 in either order -- counts them.
  bool stop() override {
    std::lock_guard<std::recursive_mutex> lk(d_stream->mu);
    if (d_started) {
      (void)rfid_lookahead_flush(d_stream->ctx);
      (void)rfid_lookahead_drain(d_stream->ctx);
      d_stream->refresh();
    }
    return true;
  }
  stream_sptr d_stream;
  bool d_started = false, d_told_end = false;
};

// ---- tag_decoder ------------------------------------------------------------------------------------
class tag_decoder_impl : public tag_decoder {
 public:
  explicit tag_decoder_impl(int sample_rate)
      : gr::block("tag_decoder", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                  gr::io_signature::makev(2, 2, std::vector<int>{(int)sizeof(float), (int)sizeof(gr_complex)})),   // lib/tag_decoder_impl.cc:39-41
        d_stream(current_or_throw("gr::rfid::tag_decoder")) {
    if (sample_rate != d_stream->params.sample_rate)
      throw mi355x::error(RFID_ERR_INVALID, "gr::rfid::tag_decoder: sample_rate differs from the gate's");
  }
  void forecast(int noutput_items, gr_vector_int &ninput_items_required) override {
    ninput_items_required[0] = noutput_items;   // lib/tag_decoder_impl.cc:72-76
  }
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    int consumed = 0, produced0 = 0;
    std::lock_guard<std::recursive_mutex> lk(d_stream->mu);
    d_stream->check(rfid_decoder_work(d_stream->ctx, (const rfid_cf32 *)input_items[0], ninput_items[0], (float *)output_items[0],
                                      noutput_items, &consumed, &produced0, nullptr, nullptr), "rfid_decoder_work");
    d_stream->refresh();
    if (produced0 > 0) produce(0, produced0);   // lib/tag_decoder_impl.cc:266 (port 1, the complex debug port, never produces)
    consume_each(consumed);                     // :395
    return WORK_CALLED_PRODUCE;                 // :396
  }
  stream_sptr d_stream;
};

// ---- reader -----------------------------------------------------------------------------------------
class reader_impl : public reader {
 public:
  reader_impl(int sample_rate, int dac_rate)
      : gr::block("reader", gr::io_signature::make(1, 1, sizeof(float)), gr::io_signature::make(1, 1, sizeof(float))),
        d_stream(current_or_throw("gr::rfid::reader")), d_dac_rate(dac_rate) {
    (void)sample_rate;
  }
  void forecast(int, gr_vector_int &ninput_items_required) override { ninput_items_required[0] = 0; }   // lib/reader_impl.cc:194-198
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    int consumed = 0, written = 0;
    std::lock_guard<std::recursive_mutex> lk(d_stream->mu);
    d_stream->check(rfid_reader_work_tx(d_stream->ctx, d_dac_rate, (const float *)input_items[0], ninput_items[0],
                                        (float *)output_items[0], noutput_items, &consumed, &written), "rfid_reader_work_tx");
    d_stream->refresh();
    consume_each(consumed);   // lib/reader_impl.cc:378
    return written;           // :379
  }
  void print_results() override {   // lib/reader_impl.cc:173-192
    std::vector<char> buf(1 << 15);
    int len = 0;
    std::lock_guard<std::recursive_mutex> lk(d_stream->mu);
    d_stream->check(rfid_print_results(d_stream->ctx, buf.data(), (int)buf.size(), &len), "rfid_print_results");
    fwrite(buf.data(), 1, (size_t)len, stdout);
    fflush(stdout);
  }
  stream_sptr d_stream;
  int d_dac_rate;
};

// ---- matched filter (replaces the third-party filter.fir_filter_ccc(5,[1]*25), apps/reader.py:65,75) -------
class matched_filter_impl : public matched_filter {
 public:
  matched_filter_impl(int decim, const std::vector<gr_complex> &taps)
      : gr::block("matched_filter", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))),
        d_slot(new stream_sptr()) {
    if (decim != 5 || taps.size() != 25) throw mi355x::error(RFID_ERR_UNSUPPORTED, "only fir_filter_ccc(5, [1]*25) is built");
    for (const gr_complex &t : taps)
      if (t != gr_complex(1.0f, 0.0f)) throw mi355x::error(RFID_ERR_UNSUPPORTED, "only all-ones taps are built");
    if (g_current && !g_current->has_filter) { *d_slot = g_current; g_current->has_filter = true; }
    else g_pending_filters.push_back(d_slot);
  }
  // outputs of the last call the library still holds (late outputs, see general_work)
  int held_back() const {
    int n = 0;
    if (d_late && *d_slot) {
      std::lock_guard<std::recursive_mutex> lk((**d_slot).mu);
      (void)rfid_mf_pending((**d_slot).ctx, &n);
    }
    return n;
  }
  // a decimator by 5 -- except while outputs are held back: they need no input to be handed out (so a runtime calls the block
  // once more when its source is done, block_executor.cc: a block whose forecast is met is called)
  void forecast(int noutput_items, gr_vector_int &ninput_items_required) override {
    ninput_items_required[0] = held_back() > 0 ? 0 : noutput_items * 5;
  }
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items) override {
    if (!*d_slot) throw mi355x::error(RFID_ERR_STATE, "gr::rfid::matched_filter: no gate constructed yet");
    stream &st = **d_slot;
    std::lock_guard<std::recursive_mutex> lk(st.mu);
    // a decimator consumes what its output has room for (sync_decimator: noutput * decim), however much the scheduler offers
    int n_in = std::min(ninput_items[0], 5 * noutput_items);
    if (!d_started) {
      d_started = true;
      // the look-ahead's staging is sized for a scheduler buffer of 64 k items (GNU Radio's default is 32 k complex items); a
      // larger call goes through in pieces inside rfid_mf_work, so a scheduler with bigger buffers cannot make it fail
      if (env_int("RFID_LOOKAHEAD", 1) != 0) {
        const int64_t cap = std::max<int64_t>(std::min<int64_t>(std::max<int64_t>(5 * (int64_t)noutput_items, ninput_items[0]), 5 * 262144), 5 * 8192) + 64;
        st.check(rfid_lookahead_enable(st.ctx, cap), "rfid_lookahead_enable");
        st.consume_ahead = tell_buffers(st.ctx, output_buffer_items(this));     // (this block's output buffer is the gate's input)
        // Late outputs (rfid_lookahead_set_late_outputs): a call returns the filter outputs of the call before it, which
        // the device finished while the scheduler ran the other blocks, instead of waiting ~28 us for its own
        // (RFID_MF_LATE_OUTPUTS=0: every call returns its own)
        if (env_int("RFID_MF_LATE_OUTPUTS", 1) != 0) {
          st.check(rfid_lookahead_set_late_outputs(st.ctx, 1), "rfid_lookahead_set_late_outputs");
          d_late = true;
          d_call_max = (int)std::min<int64_t>(((cap - 64) / 5) * 5, 0x7fffffff);
        }
      }
    }
    if (d_late) {
      if (n_in > d_call_max) n_in = d_call_max;
      int must = 0;                                 // (no room for what has to go first: that alone, in parts)
      (void)rfid_mf_must_fetch(st.ctx, &must);
      if (must > noutput_items) n_in = 0;
    }
    int n_out = 0;
    st.check(rfid_mf_work(st.ctx, (const rfid_cf32 *)input_items[0], n_in, (rfid_cf32 *)output_items[0],
                          noutput_items, &n_out), "rfid_mf_work");
    consume_each(n_in);
    return n_out;
  }
  bool stop() override {   // (see gate_impl::stop)
    if (d_started && *d_slot) { std::lock_guard<std::recursive_mutex> lk((**d_slot).mu); (void)rfid_lookahead_flush((**d_slot).ctx); }
    return true;
  }
  std::shared_ptr<stream_sptr> d_slot;   // bound now or by the next gate
  bool d_started = false, d_late = false;
  int d_call_max = 0;
};

}  // namespace

// ---- the reference's public symbols --------------------------------------------------------------------
void initialize_reader_state() {   // lib/global_vars.cc:34-54: state of the current stream back to its initial values
  stream_sptr s = current_or_throw("initialize_reader_state");
  s->check(rfid_ctx_reset(s->ctx), "rfid_ctx_reset");
  s->mirror.magn_squared_samples.clear();
  s->mirror.reader_stats.unique_tags_round.clear();
  s->refresh();
  reader_state = &s->mirror;
}

gate::sptr gate::make(int sample_rate) { return gnuradio::get_initial_sptr(new gate_impl(sample_rate)); }
tag_decoder::sptr tag_decoder::make(int sample_rate) { return gnuradio::get_initial_sptr(new tag_decoder_impl(sample_rate)); }
reader::sptr reader::make(int sample_rate, int dac_rate) { return gnuradio::get_initial_sptr(new reader_impl(sample_rate, dac_rate)); }
matched_filter::sptr matched_filter::make(int decim, const std::vector<gr_complex> &taps) {
  return gnuradio::get_initial_sptr(new matched_filter_impl(decim, taps));
}

namespace mi355x {

void configure(int device, int fixed_q, int max_num_queries, int number_unique_tags) {
  rfid_params_default(&g_next.p);
  g_next.device = device;
  g_next.p.fixed_q = fixed_q;
  g_next.p.max_num_queries = max_num_queries;
  g_next.p.number_unique_tags = number_unique_tags;
  g_next.set = true;
}

rfid_ctx *current_context() { return g_current ? g_current->ctx : nullptr; }

#ifdef GR_RFID_MINIRT
sts_flowgraph::sts_flowgraph(matched_filter::sptr mf, gate::sptr g, tag_decoder::sptr d, reader::sptr r, int chunk)
    : d_mf(mf), d_gate(g), d_dec(d), d_reader(r), d_chunk(chunk), d_bits(16, 0.0f) {
  d_txbuf.assign((size_t)rfid_reader_tx_max(1000000) * 4, 0.0f);
}

void sts_flowgraph::reader_until_idle(int n_items) {
  for (int it = 0; it < 8; ++it) {
    const int before = reader_state->gen2_logic_status;
    if (before == IDLE) break;
    gr_vector_int nin(1, n_items);
    gr_vector_const_void_star in(1, d_bits.data());
    gr_vector_void_star out(1, d_txbuf.data());
    d_reader->minirt_begin_work();
    const int written = d_reader->general_work((int)d_txbuf.size(), nin, in, out);
    if (d_keep_tx) d_tx.insert(d_tx.end(), d_txbuf.begin(), d_txbuf.begin() + written);
    n_items = 0;
    if (reader_state->gen2_logic_status == before) break;
  }
}

void sts_flowgraph::run(const gr_complex *samples, size_t n) {
  d_flushed = false; d_idle_calls = 0;   // (a flowgraph may be run again)
  // this flowgraph's own stream (not "the most recent gate of the thread": two flowgraphs may live on one thread)
  gate_impl *own_gate = dynamic_cast<gate_impl *>(d_gate.get());
  rfid_ctx *own_ctx = own_gate ? own_gate->d_stream->ctx : current_context();
  const size_t in_per_out = d_mf ? 5 : 1;   // without a matched_filter block `samples` are filter OUTPUTS (a filter of the caller's)
  reader_until_idle(0);   // START -> SEND_QUERY -> IDLE
  // the buffers between the blocks: read positions move, the data does not (it is dropped when a buffer has been read up)
  // (a gate fed by somebody else's filter is shown what it has not consumed yet AND what came in since, as a scheduler's
  // buffer does: up to two chunks)
  // ... unless the gate has just said that it can decide nothing on what it was shown: then it is shown everything that has
  // arrived (these queues are not bounded; mi355x::bounded_flowgraph is the scheduler with bounded ones)
  const size_t gate_view = d_mf ? (size_t)d_chunk : 2 * (size_t)d_chunk;
  bool gate_stalled = false;
  std::vector<gr_complex> gq, dq, mf_out((size_t)d_chunk + 8), gate_out(gate_view);
  size_t g_rd = 0, d_rd = 0;
  size_t pos = 0;
  gr_vector_int g_nin(1, 0), d_nin(1, 0);                       // (the per-call argument vectors, made once)
  gr_vector_const_void_star g_in(1, nullptr), d_in(1, nullptr);
  gr_vector_void_star g_out(1, nullptr), d_out(2, nullptr);
  // (a matched_filter block with late outputs holds the outputs of its last call: it is called until it has handed them out)
  auto mf_holds = [&]() -> bool {
    int held = 0;
    if (d_mf && own_ctx) (void)rfid_mf_pending(own_ctx, &held);
    return held > 0;
  };
  while (pos < n || g_rd < gq.size() || mf_holds()) {
    if (pos < n || mf_holds()) {
      const size_t take = (n - pos < (size_t)d_chunk * in_per_out) ? (n - pos) : (size_t)d_chunk * in_per_out;
      // (what has been read is dropped once it is most of the queue: the gate may leave tens of thousands of items standing
      // while a pass gathers, and moving them once per 8 192-item turn was a third of the run)
      if (g_rd > 0 && (g_rd == gq.size() || g_rd >= gq.size() / 2 + 65536)) { gq.erase(gq.begin(), gq.begin() + (long)g_rd); g_rd = 0; }
      if (d_mf) {
        gr_vector_int nin(1, (int)take);
        gr_vector_const_void_star in(1, samples + pos);
        gr_vector_void_star out(1, mf_out.data());
        d_mf->minirt_begin_work();
        const int produced = d_mf->general_work((int)mf_out.size(), nin, in, out);
        gq.insert(gq.end(), mf_out.begin(), mf_out.begin() + produced);
        if (d_keep_taps) d_tap_mf.insert(d_tap_mf.end(), mf_out.begin(), mf_out.begin() + produced);
        pos += (size_t)d_mf->minirt_consumed();
      } else {   // the upstream filter's output buffer
        gq.insert(gq.end(), samples + pos, samples + pos + take);
        if (d_keep_taps) d_tap_mf.insert(d_tap_mf.end(), samples + pos, samples + pos + take);
        pos += take;
      }
    }
    while (g_rd < gq.size()) {
      const size_t have = gq.size() - g_rd;
      // (at the end of the input the gate is shown everything that is left: the library's gate-keyed look-ahead only knows
      // the samples the gate was shown)
      const size_t show = (pos >= n || have < gate_view || gate_stalled) ? have : gate_view;
      if (gate_out.size() < show) gate_out.resize(show);
      const int avail = (int)show;
      g_nin[0] = avail; g_in[0] = gq.data() + g_rd; g_out[0] = gate_out.data();
      d_gate->minirt_begin_work();
      const int written = d_gate->general_work(avail, g_nin, g_in, g_out);
      const int consumed = d_gate->minirt_consumed();
      g_rd += (size_t)consumed;
      if (d_rd > 0 && d_rd == dq.size()) { dq.clear(); d_rd = 0; }
      dq.insert(dq.end(), gate_out.begin(), gate_out.begin() + written);
      if (d_keep_taps) d_tap_gate.insert(d_tap_gate.end(), gate_out.begin(), gate_out.begin() + written);
      for (;;) {
        d_nin[0] = (int)(dq.size() - d_rd); d_in[0] = dq.data() + d_rd; d_out[0] = d_bits.data();
        d_dec->minirt_begin_work();
        d_dec->general_work((int)d_bits.size(), d_nin, d_in, d_out);
        const int dcons = d_dec->minirt_consumed();
        if (dcons == 0) break;
        d_windows++;
        d_rd += (size_t)dcons;
        reader_until_idle(d_dec->minirt_produced(0));
      }
      gate_stalled = (consumed == 0 && written == 0);
      if (consumed == 0 && written == 0) {
        // the gate can decide nothing on what it has: more input first.  At the end of the input the library is told so
        // (with its look-ahead on, what it still holds back is decided then), and the gate asked again
        if (pos < n || mf_holds()) break;
        if (!d_flushed) {
          d_flushed = true;
          if (own_ctx) (void)rfid_lookahead_flush(own_ctx);
          continue;
        }
        if (++d_idle_calls > 4) { g_rd = gq.size(); break; }
      } else {
        d_idle_calls = 0;
        if (consumed == 0) break;
      }
    }
  }
}

// ---- GNU Radio's scheduling rules, one thread (see rfid/mi355x.h) -----------------------------------------------------
bounded_flowgraph::bounded_flowgraph(matched_filter::sptr mf, gate::sptr g, tag_decoder::sptr d, reader::sptr r, int buffer_items)
    : d_mf(mf), d_gate(g), d_dec(d), d_reader(r), d_cap(buffer_items < 2048 ? 2048 : buffer_items) {   // (a buffer holds an EPC window: 1 370 items)
  if (d_mf) d_mf->minirt_set_buffers(d_cap, d_cap);
  d_gate->minirt_set_buffers(d_cap, d_cap);
  d_dec->minirt_set_buffers(d_cap, d_cap);
  d_reader->minirt_set_buffers(d_cap, 0);
}

namespace {
struct edge_state { bool done = false; };   // the block that writes the buffer is done
// one block of the flowgraph as the executor sees it
struct node {
  gr::block *blk = nullptr;
  bool done = false, stalled = false, source_like = false;   // source_like: forecast() asks for nothing (the reader)
  size_t stalled_at = 0;
};
}  // namespace

void bounded_flowgraph::run(const gr_complex *samples, size_t n) {
  d_deadlock = false;
  const size_t C = (size_t)d_cap;
  std::vector<gr_complex> e_src, e_mf, e_gate;     // source -> (filter | gate), filter -> gate, gate -> decoder
  std::vector<float> e_bits;                       // decoder -> reader
  std::vector<float> txbuf((size_t)rfid_reader_tx_max(1000000) * 4, 0.0f);
  std::vector<gr_complex> outbuf(C + 8);
  std::vector<float> bitbuf(C + 8);
  size_t pos = 0;
  bool src_done = false;
  node nm, ng, nd, nr;
  nm.blk = d_mf.get(); ng.blk = d_gate.get(); nd.blk = d_dec.get(); nr.blk = d_reader.get(); nr.source_like = true;
  if (!d_mf) nm.done = true;
  for (gr::block *b : {nm.blk, ng.blk, nd.blk, nr.blk}) if (b) b->start();
  const bool trace = env_int("RFID_BOUNDED_TRACE", 0) != 0;   // every general_work call on stderr
  // runs one block once if the rules allow it; -> true when it was called and moved something
  auto turn = [&](node &nd_, size_t n_in_avail, bool in_done, size_t out_room, const void *in_ptr, void *out0, void *out1,
                  int &consumed, int &produced) -> bool {
    consumed = 0; produced = 0;
    if (nd_.done) return false;
    // (no input left and the upstream neighbour done: the block is done unless its forecast() asks for nothing -- below)
    if (out_room == 0) return false;                                                         // blocked on output
    int noutput = (int)out_room;
    gr_vector_int req(1, 0);
    nd_.blk->minirt_set_input_done(in_done);     // (what detail()->input(0)->done() tells a block)
    for (;;) {
      nd_.blk->forecast(noutput, req);
      if ((size_t)req[0] <= n_in_avail) break;
      if (noutput > 1) { noutput /= 2; continue; }          // "try again with half the output"
      if (in_done) nd_.done = true;                         // not enough input and no more coming
      return false;                                         // blocked on input
    }
    // a block that could do nothing is left alone until new input arrives or its upstream neighbour is done -- unless it asks
    // for no input: such a block is never blocked on input, a runtime calls it again at once (READY_NO_OUTPUT)
    if (nd_.stalled && !nd_.source_like && req[0] > 0 && n_in_avail == nd_.stalled_at && !in_done) return false;
    gr_vector_int nin(1, (int)n_in_avail);
    gr_vector_const_void_star in(1, in_ptr);
    gr_vector_void_star out(2, nullptr);
    out[0] = out0; out[1] = out1;
    nd_.blk->minirt_begin_work();
    const int ret = nd_.blk->general_work(noutput, nin, in, out);
    consumed = nd_.blk->minirt_consumed();
    produced = (ret == gr::block::WORK_CALLED_PRODUCE) ? nd_.blk->minirt_produced(0) : (ret > 0 ? ret : 0);
    if (trace) fprintf(stderr, "[bounded] %-14s in %6zu%s room %6zu noutput %6d -> consumed %6d produced %6d (gen2 %d gate %d)\n", nd_.blk->name().c_str(), n_in_avail,
                       in_done ? " (upstream done)" : "", out_room, noutput, consumed, produced, (int)reader_state->gen2_logic_status, (int)reader_state->gate_status);
    if (ret == gr::block::WORK_DONE) { nd_.done = true; return false; }
    if (consumed == 0 && produced == 0) {
      // nothing moved.  Its upstream neighbour was done when the call was made and the block asked for input: nothing ever
      // will, the block is done (it would be asked again and again until the flowgraph is stopped: the reference's tag_decoder
      // ends like that on an incomplete last window).  A block that asked for nothing (a gate waiting for the decoder / reader
      // to arm it) is simply called again in the next round.  Else it waits for new input (or for the neighbour to finish).
      if (in_done && req[0] > 0) { nd_.done = true; return false; }
      nd_.stalled = true; nd_.stalled_at = n_in_avail;
      return false;
    }
    nd_.stalled = false;
    return true;
  };
  // The reader asks for no input (forecast: 0, lib/reader_impl.cc:194-198): a runtime calls it whenever its output has room, i.e.
  // again and again until it has nothing to send -- its turn is that: until a call moves nothing (START -> SEND_QUERY -> IDLE
  // before the first sample arrives; ACK, then the carrier, behind an RN16)
  auto reader_turn = [&]() -> bool {
    bool any = false;
    for (int it = 0; it < 8; ++it) {
      int rc_ = 0, rp_ = 0;
      const bool m = turn(nr, e_bits.size(), nd.done, txbuf.size(), e_bits.data(), txbuf.data(), nullptr, rc_, rp_);
      if (rc_ > 0) e_bits.erase(e_bits.begin(), e_bits.begin() + rc_);
      if (rp_ > 0 && d_keep_tx) d_tx.insert(d_tx.end(), txbuf.begin(), txbuf.begin() + rp_);
      if (!m) break;
      any = true;
    }
    return any;
  };
  reader_turn();
  for (long round = 0;; ++round) {
    bool moved = false;
    // ---- file source ----
    if (!src_done) {
      std::vector<gr_complex> &e0 = e_src;
      const size_t room = C - e0.size();
      const size_t take = (n - pos < room) ? (n - pos) : room;
      if (take > 0) { e0.insert(e0.end(), samples + pos, samples + pos + take); pos += take; moved = true; }
      else if (pos >= n) { src_done = true; moved = true; }   // (a file source hands out its last items, the call after that says WORK_DONE)
    }
    int cons = 0, prod = 0;
    // ---- matched filter ----
    if (d_mf) {
      if (turn(nm, e_src.size(), src_done, C - e_mf.size(), e_src.data(), outbuf.data(), nullptr, cons, prod)) moved = true;
      if (cons > 0) e_src.erase(e_src.begin(), e_src.begin() + cons);
      if (prod > 0) { e_mf.insert(e_mf.end(), outbuf.begin(), outbuf.begin() + prod); if (d_keep_taps) d_tap_mf.insert(d_tap_mf.end(), outbuf.begin(), outbuf.begin() + prod); }
    }
    // ---- gate ----
    {
      std::vector<gr_complex> &gi = d_mf ? e_mf : e_src;
      const bool gi_done = d_mf ? nm.done : src_done;
      if (turn(ng, gi.size(), gi_done, C - e_gate.size(), gi.data(), outbuf.data(), nullptr, cons, prod)) moved = true;
      if (cons > 0) { if (!d_mf && d_keep_taps) d_tap_mf.insert(d_tap_mf.end(), gi.begin(), gi.begin() + cons); gi.erase(gi.begin(), gi.begin() + cons); }
      if (prod > 0) { e_gate.insert(e_gate.end(), outbuf.begin(), outbuf.begin() + prod); if (d_keep_taps) d_tap_gate.insert(d_tap_gate.end(), outbuf.begin(), outbuf.begin() + prod); }
    }
    // ---- tag_decoder ----
    {
      if (turn(nd, e_gate.size(), ng.done, C - e_bits.size(), e_gate.data(), bitbuf.data(), outbuf.data(), cons, prod)) moved = true;
      if (cons > 0) { e_gate.erase(e_gate.begin(), e_gate.begin() + cons); d_windows++; }
      if (prod > 0) e_bits.insert(e_bits.end(), bitbuf.begin(), bitbuf.begin() + prod);
    }
    // ---- reader (its output goes to a file sink that takes everything) ----
    if (reader_turn()) moved = true;
    const bool all_done = src_done && nm.done && ng.done && nd.done && nr.done;
    if (all_done) break;
    if (!moved) {
      // nobody could move: with input left somewhere that is a flowgraph that hangs under a real runtime
      if (!(e_src.empty() && e_mf.empty() && e_gate.empty() && e_bits.empty() && src_done)) d_deadlock = !(ng.done && nd.done);
      break;
    }
  }
  for (gr::block *b : {nm.blk, ng.blk, nd.blk, nr.blk}) if (b) b->stop();
}
#endif

}  // namespace mi355x
}  // namespace rfid
}  // namespace gr
