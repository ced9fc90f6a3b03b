// rfid/mi355x.h -- what the MI355X adaptor adds to the reference's block API (nothing here exists in gr-rfid):
//   * run-time configuration of what the reference fixes at compile time (FIXED_Q, MAX_NUM_QUERIES,
//     NUMBER_UNIQUE_TAGS: include/rfid/global_vars.h:72,76,100) and of the GPU the next gate::make() uses;
//   * gr::rfid::matched_filter, the HIP replacement of filter.fir_filter_ccc(5, [1]*25) (apps/reader.py:65,75);
//   * a single-threaded scheduler for the offline topology of apps/reader.py:101-112 (README.md:40, GR_SCHEDULER=STS)
//     where no GNU Radio runtime exists.
#ifndef INCLUDED_RFID_MI355X_H
#define INCLUDED_RFID_MI355X_H

#include <rfid/gate.h>
#include <rfid/global_vars.h>
#include <rfid/reader.h>
#include <rfid/tag_decoder.h>
#include <rfid_mi355x.h>

#include <stdexcept>
#include <string>

namespace gr {
namespace rfid {

class RFID_BLOCK_API matched_filter : virtual public gr::block {
 public:
#ifdef GR_RFID_MINIRT
  typedef std::shared_ptr<matched_filter> sptr;
#else
  typedef boost::shared_ptr<matched_filter> sptr;
#endif
  // filter.fir_filter_ccc(decim, taps); only (5, [1]*25) is built.  Binds to the stream of the most recent gate.
  static sptr make(int decim, const std::vector<gr_complex> &taps);
};

namespace mi355x {

struct RFID_BLOCK_API error : std::runtime_error {
  int status;
  error(int st, const std::string &what) : std::runtime_error(what), status(st) {}
};

// parameters of the stream the NEXT gate::make() creates (defaults: device 0 and the reference's constants;
// the environment variables RFID_DEVICE / RFID_FIXED_Q / RFID_MAX_NUM_QUERIES / RFID_NUMBER_UNIQUE_TAGS override
// the defaults for flowgraphs that cannot call this, e.g. an unchanged apps/reader.py)
RFID_BLOCK_API void configure(int device, int fixed_q = FIXED_Q, int max_num_queries = MAX_NUM_QUERIES,
                              int number_unique_tags = NUMBER_UNIQUE_TAGS);
// the C-ABI context behind the stream of the most recent gate (nullptr before the first gate::make())
RFID_BLOCK_API rfid_ctx *current_context();

#ifdef GR_RFID_MINIRT
// file_source -> matched_filter -> gate -> tag_decoder -> reader, one thread, every block called with what is
// available until nothing moves (the reference's README recommends GR_SCHEDULER=STS, README.md:40)
class RFID_BLOCK_API sts_flowgraph {
 public:
  // mf may be null: the flowgraph of apps/reader.py as it stands, whose matched filter is GNU Radio's own block -- run()
  // is then handed that filter's OUTPUT (400 ksps) and feeds the gate with it
  sts_flowgraph(matched_filter::sptr mf, gate::sptr g, tag_decoder::sptr d, reader::sptr r, int chunk = 8192);
  void run(const gr_complex *samples, size_t n);
  long windows_decoded() const { return d_windows; }
  void keep_tx(bool on) { d_keep_tx = on; }
  const std::vector<float> &tx_samples() const { return d_tx; }   // what the reader block wrote (file_sink_reader)
  // debug taps of apps/reader.py:67-72,114-118: matched-filter output and gated samples of the whole run
  void keep_taps(bool on) { d_keep_taps = on; }
  const std::vector<gr_complex> &tap_matched_filter() const { return d_tap_mf; }
  const std::vector<gr_complex> &tap_gate() const { return d_tap_gate; }

 private:
  void reader_until_idle(int n_items);
  matched_filter::sptr d_mf;
  gate::sptr d_gate;
  tag_decoder::sptr d_dec;
  reader::sptr d_reader;
  int d_chunk;
  long d_windows = 0;
  int d_idle_calls = 0;
  bool d_flushed = false;
  bool d_keep_tx = false, d_keep_taps = false;
  std::vector<float> d_tx, d_txbuf, d_bits;
  std::vector<gr_complex> d_tap_mf, d_tap_gate;
};

// The same flowgraph under GNU Radio's scheduling RULES (gnuradio-runtime/lib/block_executor.cc), one thread, blocks in turn:
// buffers between the blocks hold `buffer_items` items and no more; a block is called with what its input buffer holds and
// the room its output buffer has, noutput_items halved while forecast() asks for more input than there is; a block that
// neither consumed nor produced is not called again before new input has arrived or its upstream neighbour is done;
// a block whose neighbour is done and which can do nothing more is done itself; when all are, stop() is called on
// every block -- nobody tells anybody that the input has ended.  (What a real runtime does with the reference's blocks
// and apps/reader.py:120-131; used by the tests for the end of the input and small buffers.)
class RFID_BLOCK_API bounded_flowgraph {
 public:
  bounded_flowgraph(matched_filter::sptr mf, gate::sptr g, tag_decoder::sptr d, reader::sptr r, int buffer_items = 8192);
  void run(const gr_complex *samples, size_t n);
  long windows_decoded() const { return d_windows; }
  bool stalled() const { return d_deadlock; }            // the flowgraph stopped with input left and nobody able to move
  void keep_tx(bool on) { d_keep_tx = on; }
  const std::vector<float> &tx_samples() const { return d_tx; }
  void keep_taps(bool on) { d_keep_taps = on; }
  const std::vector<gr_complex> &tap_matched_filter() const { return d_tap_mf; }
  const std::vector<gr_complex> &tap_gate() const { return d_tap_gate; }

 private:
  matched_filter::sptr d_mf;
  gate::sptr d_gate;
  tag_decoder::sptr d_dec;
  reader::sptr d_reader;
  int d_cap;
  long d_windows = 0;
  bool d_deadlock = false, d_keep_tx = false, d_keep_taps = false;
  std::vector<float> d_tx;
  std::vector<gr_complex> d_tap_mf, d_tap_gate;
};
#endif

}  // namespace mi355x
}  // namespace rfid
}  // namespace gr
#endif
