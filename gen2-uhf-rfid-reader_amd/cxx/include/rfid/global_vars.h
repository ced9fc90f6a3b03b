// rfid/global_vars.h -- the shared reader state and the protocol constants, as the reference's blocks and apps
// see them (gr-rfid/include/rfid/global_vars.h:31-67 types, :72-143 constants, :146-147 the global).
//
// MI355X adaptor: the state of an RX stream lives inside an rfid_ctx (include/rfid_mi355x.h) owned by the gate
// block; `reader_state` is kept for source compatibility as a read-only MIRROR of the stream of the most recently
// constructed gate, refreshed after every general_work() of a block of that stream.  FIXED_Q, MAX_NUM_QUERIES and
// NUMBER_UNIQUE_TAGS are compile-time constants in the reference; here they are the DEFAULTS of run-time
// parameters (gr::rfid::mi355x::configure() in rfid/mi355x.h, before gate::make()).
#ifndef INCLUDED_RFID_GLOBAL_VARS_H
#define INCLUDED_RFID_GLOBAL_VARS_H

#include <rfid/api.h>
#include <sys/time.h>

#include <map>
#include <vector>

namespace gr {
namespace rfid {

enum STATUS { RUNNING, TERMINATED };
enum GEN2_LOGIC_STATUS { SEND_QUERY, SEND_ACK, SEND_QUERY_REP, IDLE, SEND_CW, START, SEND_QUERY_ADJUST, SEND_NAK_QR,
                         SEND_NAK_Q, POWER_DOWN };
enum GATE_STATUS { GATE_OPEN, GATE_CLOSED, GATE_SEEK_RN16, GATE_SEEK_EPC };
enum DECODER_STATUS { DECODER_DECODE_RN16, DECODER_DECODE_EPC };

struct READER_STATS {
  int n_queries_sent;
  int cur_inventory_round;
  int cur_slot_number;
  int max_slot_number;
  int max_inventory_round;
  int n_epc_correct;
  std::vector<int> unique_tags_round;
  std::map<int, int> tag_reads;
  struct timeval start, end;
};

struct READER_STATE {
  STATUS status;
  GEN2_LOGIC_STATUS gen2_logic_status;
  GATE_STATUS gate_status;
  DECODER_STATUS decoder_status;
  READER_STATS reader_stats;
  std::vector<float> magn_squared_samples;   // |in - dc_est|^2 of the window being gated (lib/gate_impl.cc:175,186)
  int n_samples_to_ungate;
};

// ---- constants (values of gr-rfid/include/rfid/global_vars.h:72-143) -------------------------------
const int FIXED_Q = 0;
const int MAX_NUM_QUERIES = 1000;
const int MAX_INVENTORY_ROUND = 50;
const int NUMBER_UNIQUE_TAGS = 100;
const int NUM_PULSES_COMMAND = 5;
const int T1_D = 240, T2_D = 480, PW_D = 12, DELIM_D = 12, TRCAL_D = 200, RTCAL_D = 72;   // us
const int CW_D = 250, P_DOWN_D = 2000, RN16_D = 575, EPC_D = 3375;                          // us
const int TAG_PREAMBLE_BITS = 6, RN16_BITS = 17, EPC_BITS = 129, QUERY_LENGTH = 22;
const int T_READER_FREQ = 40000;
const float TAG_BIT_D = 1.0f / T_READER_FREQ * 1000000.0f;   // 25 us
const int TAG_PREAMBLE[] = {1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 1};
const float THRESH_FRACTION = 0.75f;
const int WIN_SIZE_D = 250;
const int DC_SIZE_D = 120;

// the global of the reference (include/rfid/global_vars.h:146-147)
extern RFID_BLOCK_API READER_STATE *reader_state;
extern RFID_BLOCK_API void initialize_reader_state();

}  // namespace rfid
}  // namespace gr
#endif
