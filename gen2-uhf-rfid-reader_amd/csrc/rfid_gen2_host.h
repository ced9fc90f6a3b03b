// rfid_gen2_host.h -- host side of the Gen2 trace synthesiser (rfid_synth_gen2): turns the public slot
// table (rfid_synth_slot, include/rfid_mi355x.h) into the records synth_gen2_kernel reads -- command bits with
// the CRC-5 appended, slot start offsets (prefix sum of the PIE-coded command lengths).  Included by
// rfid_capi.hip (the product) and by the kernel emulator of tests/wave_emu (test infrastructure).
#pragma once
#include <cstring>
#include <vector>

#include "rfid_kernels.hpp"
#include "rfid_mi355x.h"

namespace rfidh {
using namespace rfidk;
// Query bits (reader_impl.cc:131-146): 1000 | DR | M | TRext | Sel | Session | Target | Q | CRC-5 (:383-443)
inline uint32_t gen2_query_bits(int q) {
  uint32_t bits = 0;
  const int head[13] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int b : head) bits = (bits << 1) | (uint32_t)b;
  for (int i = 3; i >= 0; --i) bits = (bits << 1) | (uint32_t)((q >> i) & 1);
  unsigned reg = 0x09;   // preset 01001
  for (int i = 16; i >= 0; --i) {
    const unsigned fb = ((reg >> 4) & 1u) ^ ((bits >> i) & 1u);
    reg = (reg << 1) & 0x1Fu;
    if (fb) reg ^= 0x09;
  }
  return (bits << 5) | reg;   // 22 bits, first bit = bit 21
}
inline int gen2_pie_us(uint32_t bits, int n) {
  int ones = 0;
  for (int i = 0; i < n; ++i) ones += (int)((bits >> i) & 1u);
  return G2_DATA0 * n + (G2_DATA1 - G2_DATA0) * ones;
}
const int G2_FRAME_SYNC_US = G2_DELIM + G2_DATA0 + G2_RTCAL;   // 108
// fills dev (n_slots + 2 records: opening carrier, the slots, closing carrier); returns total raw samples or < 0
inline int64_t gen2_layout(const rfid_synth_gen2_params &p, const rfid_synth_slot *slots, int64_t n_slots,
                    std::vector<Gen2SlotDev> *dev) {
  if (p.n_tags < 0 || p.n_tags > G2_MAX_TAGS || p.tail_us < 0) return -1;
  int64_t t_us = 0;
  auto carrier = [&](int us) {
    if (dev) {
      Gen2SlotDev d;
      memset(&d, 0, sizeof(d));
      d.raw_start = 2 * t_us; d.kind = 2; d.cw_us = us;
      dev->push_back(d);
    }
    t_us += us;
  };
  carrier(G2_CW_ACK);   // START: cw_ack (reader_impl.cc:218-224)
  for (int64_t i = 0; i < n_slots; ++i) {
    const rfid_synth_slot &s = slots[i];
    if (s.cmd > 1 || s.q > 15 || s.n_tags > G2_MAX_RESP) return -1;
    Gen2SlotDev d;
    memset(&d, 0, sizeof(d));
    d.raw_start = 2 * t_us;
    d.kind = s.cmd;
    if (s.cmd == 0) { d.cmd_bits = gen2_query_bits(s.q); d.n_cmd_bits = 22; }
    else { d.cmd_bits = 0; d.n_cmd_bits = 4; }                         // QueryRep: 00 + session 00 (:106-111)
    d.ack_bits = (1u << 16) | (uint32_t)s.ack;                         // 01 + RN16 (:149-154)
    d.n_tags = s.n_tags; d.has_epc = (s.has_epc && s.n_tags >= 1) ? 1 : 0;
    d.rn16_off_raw = s.rn16_off_raw; d.epc_off_raw = s.epc_off_raw;
    for (int k = 0; k < G2_MAX_RESP; ++k) {
      if (k < s.n_tags && s.tag[k] >= p.n_tags) return -1;
      d.tag[k] = s.tag[k]; d.rn16[k] = s.rn16[k];
    }
    for (int k = 0; k < 4; ++k) d.epc[k] = s.epc[k];
    const int cmd_us = G2_FRAME_SYNC_US + (s.cmd == 0 ? G2_TRCAL : 0) + gen2_pie_us(d.cmd_bits, d.n_cmd_bits);
    const int ack_us = G2_FRAME_SYNC_US + gen2_pie_us(d.ack_bits, 18);
    // replies must lie inside the carrier that follows the command / the ACK
    if (s.n_tags && (s.rn16_off_raw < 0 || s.rn16_off_raw + G2_RN16_LV * G2_HALF_BIT_RAW > 2 * G2_CW_QUERY)) return -1;
    if (d.has_epc && (s.epc_off_raw < 0 || s.epc_off_raw + G2_EPC_LV * G2_HALF_BIT_RAW > 2 * G2_CW_ACK)) return -1;
    if (dev) dev->push_back(d);
    t_us += cmd_us + G2_CW_QUERY + ack_us + G2_CW_ACK;
  }
  carrier(p.tail_us);
  return 2 * t_us;
}
}  // namespace rfidh
