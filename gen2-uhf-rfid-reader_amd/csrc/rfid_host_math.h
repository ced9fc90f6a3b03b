// rfid_host_math.h -- host-side derivations shared by the C-ABI library and the test
// emulator driver: the constructor arithmetic of the reference blocks, same types and order.
#pragma once
#include <cmath>

namespace rfidh {

// const float TAG_BIT_D = 1.0/T_READER_FREQ * pow(10,6)   (include/rfid/global_vars.h:110-111)
inline float tag_bit_d() { return (float)(1.0 / 40000 * pow(10, 6)); }

// n_samples_TAG_BIT of tag_decoder_impl (lib/tag_decoder_impl.cc:60)
inline float n_samples_tag_bit(int sample_rate) { return (float)(tag_bit_d() * sample_rate / pow(10, 6)); }

// the 20 half-period candidates of tag_detection_EPC (lib/tag_decoder_impl.cc:151-152,162)
inline void t_candidates(float *t_cand, int sample_rate) {
  const float nb = n_samples_tag_bit(sample_rate);
  const int number_steps = 20;
  const float min_val = (float)(nb / 2.0 - nb / 2.0 / 100);
  const float max_val = (float)(nb / 2.0 + nb / 2.0 / 100);
  for (int t = 0; t < number_steps; t++) t_cand[t] = min_val + t * (max_val - min_val) / (number_steps - 1);
}

}  // namespace rfidh
