// rfid/reader.h -- gr::rfid::reader, the reference's public block API verbatim (gr-rfid/include/rfid/reader.h:38-53).
#ifndef INCLUDED_RFID_READER_H
#define INCLUDED_RFID_READER_H

#include <gnuradio/block.h>
#include <rfid/api.h>
#ifndef GR_RFID_MINIRT
#include <boost/shared_ptr.hpp>
#endif

namespace gr {
namespace rfid {

class RFID_BLOCK_API reader : virtual public gr::block {
 public:
#ifdef GR_RFID_MINIRT
  typedef std::shared_ptr<reader> sptr;
#else
  typedef boost::shared_ptr<reader> sptr;
#endif
  virtual void print_results() = 0;
  static sptr make(int sample_rate, int dac_rate);
};

}  // namespace rfid
}  // namespace gr
#endif
