// rfid/tag_decoder.h -- gr::rfid::tag_decoder, the reference's public block API verbatim (gr-rfid/include/rfid/tag_decoder.h:35-49):
// an abstract gr::block with a static factory; the implementation (cxx/lib/rfid_blocks.cc) hands every
// general_work() buffer through the C-ABI (include/rfid_mi355x.h) to the MI355X kernels.
#ifndef INCLUDED_RFID_TAG_DECODER_H
#define INCLUDED_RFID_TAG_DECODER_H

#include <gnuradio/block.h>
#include <rfid/api.h>
#ifndef GR_RFID_MINIRT
#include <boost/shared_ptr.hpp>
#endif

namespace gr {
namespace rfid {

class RFID_BLOCK_API tag_decoder : virtual public gr::block {
 public:
#ifdef GR_RFID_MINIRT
  typedef std::shared_ptr<tag_decoder> sptr;
#else
  typedef boost::shared_ptr<tag_decoder> sptr;
#endif
  static sptr make(int sample_rate);
};

}  // namespace rfid
}  // namespace gr
#endif
