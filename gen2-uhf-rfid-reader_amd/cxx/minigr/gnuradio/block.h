// minigr/gnuradio/block.h -- the part of GNU Radio 3.7's gr::block that the rfid blocks use, for building and
// running the gr::rfid adaptors where GNU Radio is not installed (this image).  With a real GNU Radio drop this
// directory from the include path: the adaptors (cxx/include/rfid/*.h, cxx/lib/rfid_blocks.cc) compile against
// <gnuradio/block.h> unchanged.  Own code (an interface subset, no GNU Radio source): names and signatures follow
// the public API the reference's blocks are written against (lib/gate_impl.h:46-55, lib/gate_impl.cc:41-44,79-83,198).
#pragma once
#define GR_RFID_MINIRT 1

#include <complex>
#include <memory>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {

class io_signature {
 public:
  typedef std::shared_ptr<io_signature> sptr;
  static sptr make(int min_streams, int max_streams, int sizeof_stream_item) {
    return sptr(new io_signature(min_streams, max_streams, std::vector<int>(1, sizeof_stream_item)));
  }
  static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizeof_stream_items) {
    return sptr(new io_signature(min_streams, max_streams, sizeof_stream_items));
  }
  int min_streams() const { return d_min; }
  int max_streams() const { return d_max; }
  int sizeof_stream_item(int i) const { return d_sizes[(size_t)i < d_sizes.size() ? (size_t)i : d_sizes.size() - 1]; }

 private:
  io_signature(int mn, int mx, const std::vector<int> &s) : d_min(mn), d_max(mx), d_sizes(s) {}
  int d_min, d_max;
  std::vector<int> d_sizes;
};

class block {
 public:
  enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
  virtual ~block() {}
  const std::string &name() const { return d_name; }
  io_signature::sptr input_signature() const { return d_in; }
  io_signature::sptr output_signature() const { return d_out; }
  virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required) {
    for (size_t i = 0; i < ninput_items_required.size(); ++i) ninput_items_required[i] = noutput_items;
  }
  virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items) = 0;
  // gr::block::start / stop (called by the runtime when the flowgraph starts / has stopped: gnuradio/block.h)
  virtual bool start() { return true; }
  virtual bool stop() { return true; }
  void consume_each(int how_many_items) { d_consumed = how_many_items; }
  void produce(int which_output, int how_many_items) {
    if ((size_t)which_output >= d_produced.size()) d_produced.resize((size_t)which_output + 1, 0);
    d_produced[(size_t)which_output] = how_many_items;
  }
  // what a scheduler reads back after general_work() (GNU Radio keeps the same numbers in the block detail)
  void minirt_begin_work() { d_consumed = 0; d_produced.assign(d_produced.size(), 0); }
  int minirt_consumed() const { return d_consumed; }
  int minirt_produced(int port) const { return (size_t)port < d_produced.size() ? d_produced[(size_t)port] : 0; }
  // what GNU Radio's block_detail tells a block about its buffers (detail()->input(0)->max_possible_items_available(),
  // detail()->output(0)->bufsize()): items, 0 = not bounded (the single-threaded scheduler's queues grow as needed)
  void minirt_set_buffers(int input_items, int output_items) { d_in_cap = input_items; d_out_cap = output_items; }
  // ... and about its upstream neighbour (detail()->input(0)->done()): it has finished, what is in the buffer is all there will be
  void minirt_set_input_done(bool done) { d_in_done = done; }
  bool minirt_input_done() const { return d_in_done; }
  int minirt_input_capacity() const { return d_in_cap; }
  int minirt_output_capacity() const { return d_out_cap; }

 protected:
  block() {}   // for virtual inheritance (class X : virtual public gr::block)
  block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out) {}

 private:
  std::string d_name;
  io_signature::sptr d_in, d_out;
  int d_consumed = 0;
  int d_in_cap = 0, d_out_cap = 0;
  bool d_in_done = false;
  std::vector<int> d_produced = std::vector<int>(2, 0);
};

}  // namespace gr

namespace gnuradio {
template <class T>
std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
}  // namespace gnuradio
