// rfid/gate.h -- gr::rfid::gate, the reference's public block API verbatim (gr-rfid/include/rfid/gate.h:38-53):
// an abstract gr::block with a static factory; the implementation (cxx/lib/rfid_blocks.cc) hands every
// general_work() buffer through the C-ABI (include/rfid_mi355x.h) to the MI355X kernels.
#ifndef INCLUDED_RFID_GATE_H
#define INCLUDED_RFID_GATE_H

#include <gnuradio/block.h>
#include <rfid/api.h>
#ifndef GR_RFID_MINIRT
#include <boost/shared_ptr.hpp>
#endif

namespace gr {
namespace rfid {

class RFID_BLOCK_API gate : virtual public gr::block {
 public:
#ifdef GR_RFID_MINIRT
  typedef std::shared_ptr<gate> sptr;
#else
  typedef boost::shared_ptr<gate> sptr;
#endif
  static sptr make(int sample_rate);
};

}  // namespace rfid
}  // namespace gr
#endif
