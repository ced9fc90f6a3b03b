// rfid_capi.hip -- C-ABI of librfid_mi355x.so (see include/rfid_mi355x.h).
//
// Host side of the library: context, HBM workspace, kernel launches on the context's HIP
// stream, and the small integer bookkeeping of READER_STATE.  All sample arithmetic
// happens in the kernels of rfid_kernels.hpp; there is no CPU implementation of the path
// in this library -- without a gfx950 device every entry point fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <ctime>
#include <pthread.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <new>
#include <vector>

#include "rfid_host_math.h"
#include "rfid_kernels.hpp"
#include "rfid_mi355x.h"
#include "rfid_gen2_host.h"
// the launch list of the long-stream front end, on the stream named by the enclosing scope's `ls2_stream`
#define LS2_LAUNCH(kernel, gx, gy, block, args) \
  hipLaunchKernelGGL(rfidk::kernel, dim3((unsigned)(gx), (unsigned)(gy)), dim3((unsigned)(block)), 0, ls2_stream, args)
static thread_local hipStream_t ls2_stream = nullptr;
// (experiment knob front_lds_kb: dynamic LDS the first pass's workgroups ask for on top of their own -- fewer of its one-wave
// workgroups per CU, so the small launches of the pass before find wave slots beside it)
static thread_local unsigned ls2_front_lds = 0;
#define LS2_LAUNCH_FRONT(kernel, gx, gy, block, args) \
  hipLaunchKernelGGL(rfidk::kernel, dim3((unsigned)(gx), (unsigned)(gy)), dim3((unsigned)(block)), ls2_front_lds, ls2_stream, args)
// RFID_LA_PROFILE=1: where the look-ahead's time goes (printed when the context is destroyed)
static double g_la_t[20] = {0};
static long g_la_n[20] = {0};
static bool g_la_on = false;   // some context was created with RFID_LA_PROFILE=1 (the counters are a process-wide developer aid: not for several profiled contexts in different threads at once)
static inline void la_count(int k, double dt) { if (g_la_on) { g_la_t[k] += dt; g_la_n[k]++; } }
static inline double la_now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; }
struct LaTimer { int k; double t0; explicit LaTimer(int kk) : k(kk), t0(g_la_on ? la_now() : 0.0) {} ~LaTimer() { if (g_la_on) la_count(k, la_now() - t0); } };
#include "rfid_ls2_enqueue.hpp"

using namespace rfidk;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

}  // namespace

// Everything the environment can change, read ONCE when a context is created (rfid_ctx_create -> knobs_from_env) and kept
// here; rfid_ctx_set_knob changes a value of a living context (what the tests and the A/B scripts do).  One table
// (g_knob_table below) names them: INTEGRATION.md section 13 is that table.
struct RfidKnobs {
  int long_stream = 1;     // RFID_LONG_STREAM       0 never, 1 automatic (cost model), 2 whenever a trace can be cut
  int overlap = 1;         // RFID_OVERLAP           0 one stream only, 1 the next pass's first launches on the second stream, 2 ... and a
                           //                        second result set (decoder beside the next front end; fused front end, many traces)
  int ls_calibrate = 1;    // RFID_LS_CALIBRATE      0: keep the built-in cost model (profiled runs)
  int ls_debug = 0;        // RFID_LS_DEBUG          1: what the long-stream rounds did, on stderr (synchronises every pass)
  int la_profile = 0;      // RFID_LA_PROFILE        1: where the look-ahead's host time went, on stderr when the context is destroyed
  // ---- test hooks ----
  int front_unfused = 0;   // RFID_FRONT_UNFUSED     1: many traces through the stage kernels instead of front_end_fused_kernel
  int front_chunks = 1;    // RFID_FRONT_CHUNKS      2..16: the time-chunked stage kernels on two streams (round 1's overlap)
  int fsm_lanes_min = -1;  // RFID_LS2_FSM_LANES_MIN from how many possible units on the state machine runs one lane per unit (-1: 8192)
  int dc_rounds = -1;      // RFID_LS2_DC_ROUNDS     0..64: dc_est rounds a long-stream pass enqueues behind the first (-1: by the pass's size -- 0 / 3 / 10; what they leave, the finishing walk takes)
  int la_upload_kernel = 1;  // RFID_LA_UPLOAD_KERNEL  look-ahead: 1 a call's samples are fetched from page-locked memory by a launch, 0 by a transfer
  int front_lds_kb = -1;   // RFID_LS_FRONT_LDS_KB   0..64: extra LDS per workgroup of the long-stream first pass (caps its waves per CU); -1: 10 for long traces
};

struct rfid_ctx {
  rfid_params prm;
  RfidKnobs knobs;
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512];
  float t_cand[N_TCAND];

  // ---- READER_STATE (host) ----
  rfid_reader_state rs;

  // ---- streaming ----
  void *d_small = nullptr;        // one allocation behind the small buffers below
  GateState *d_gate1 = nullptr;   // gate state of the single streaming RX stream
  int *d_io = nullptr;            // [2]
  DevBuf s_in, s_out;
  DevBuf synth_tab;               // slot table of rfid_synth_gen2
  // long-stream front end (few long traces cut along time into concurrently processed pieces, rfid_ls2.hpp)
  DevBuf ls2_ws;                  // its work space (one allocation, carved up by ls2_layout)
  DevBuf ls2_ws_alt;              // a second one (rfid_batch_plan, where the device has the room): with it and a second matched-filter
                                  // output buffer the first launches of pass k + 1 -- the fused first pass: they touch the raw samples,
                                  // y and the work space only -- run on stream2 beside the rest of pass k; the two are used alternately
  bool ls2_mark_failed = false;   // (scratch of ls_enqueue's call-back)
  bool y_touched = false;         // work outside that protocol has used c->d_y on the main stream since the last such pass
  Ls2Ctl *ls2_host = nullptr;     // page-locked copy of the control block of the last pass (report) + consumed[0]
  Ls2Ctl *d_ls2_ctl = nullptr;    // the control block of the last pass that ran the front end (device), else nullptr
  int ls2_P = 0;                  // its nominal piece length
  int ls2_rounds[3] = {0, 0, 0};  // re-run rounds its launch list held per stage (avg_ampl, state machine, dc_est)
  bool ls2_generous = false;      // a pass ran out of rounds once: the launch lists hold the full number of rounds from then on
  int ls_mode = 1;                // 0 never, 1 automatic, 2 whenever a trace can be cut
  double ls_fixed_ms = 0.45, ls_ns_per_sample = 0.03, seq_ns_per_sample = 10.2;   // cost model of the automatic choice (ls_calibrate)
  bool ls_calibrated = false;     // the three numbers were measured on this device (or that was tried, or is not wanted)
  bool ls_measured = false;       // ls_calibrate really measured them (it leaves the built-in numbers when an allocation or a pass fails)
  bool plan_for_stream = false;   // the plan being made is a stream's own (rfid_stream_begin / look-ahead): no second filter-output buffer
  // whole-chain streaming (rfid_stream_*)
  struct StreamIO {
    bool open = false;
    bool failed = false;          // a call failed half-way: the stream has to be begun again
    bool ymode = false;           // the stream's samples are matched-filter OUTPUTS (400 ksps): the look-ahead keyed on the gate's
                                  // input, for a flowgraph whose filter is not this library's; d_buf then holds y, no filter runs
    int dec() const { return ymode ? 1 : DECIM; }          // stream samples per decimated sample
    int hist() const { return ymode ? 0 : 28; }            // stream samples kept in front of the held-back tail (SIO_HIST)
    int64_t max_chunk = 0, tail_max = 0;
    float2 *d_buf[2] = {nullptr, nullptr};
    rfid_cf32 *h_pin[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr};
    hipEvent_t ev_free[2] = {nullptr, nullptr};   // processing of the chunk in d_buf[i] is over (buffer reusable)
    int cur = 0;                  // buffer that receives the next upload
    bool pending = false;         // a chunk is uploaded (or uploading) into d_buf[pend_idx] and not processed yet
    int pend_idx = 0;
    int64_t pend_new = 0;         // its new samples (at offset tail_max)
    int64_t tail_len = 0;         // raw samples held back by the last processed chunk, placed right before tail_max in d_buf[pend_idx / cur]
                                  // (look-ahead: negative when the exact per-call scan has consumed into the samples behind tail_max)
    int64_t acc_new = 0;          // (look-ahead) samples uploaded behind tail_max of d_buf[cur] by the calls since the last pass was submitted
    hipEvent_t ev_hist = nullptr; // (look-ahead) the filter history in front of the other buffer's upload area is in place
    int64_t raw_base = 0;         // global raw index of the first held-back sample (multiple of 5)
    std::vector<rfid_stream_window> out_w;   // windows completed but not yet delivered (caller arrays too small)
    std::vector<rfid_decode_result> out_r;
    // a whole-chain pass that has been enqueued (sio_submit) and not looked at yet (sio_collect)
    struct Pass {
      bool active = false;
      int b = 0;                  // the buffer it works on
      int64_t n_have = 0, n_out = 0;
      bool flush = false;
      bool enq = false;           // the long-stream front end was enqueued
      bool small = false;         // a short pass: the sequential scan over [0, seq_end) was enqueued instead (no launch list)
      int64_t seq_end = 0;
      bool prefetched = false;    // (look-ahead) decoder + window packet were enqueued behind it, assuming it succeeds
      int n_hdr = 0, usual = 0;   //   ... with these packet sizes
    } pass;
    hipEvent_t ev_y = nullptr;    // (look-ahead) the filter outputs of the submitted pass have reached the host
  } sio;
  // look-ahead of the per-block calls (rfid_lookahead_enable): what rfid_mf_work's whole-chain pass left for the gate /
  // decoder calls that follow
  struct LookAhead {
    // the gated samples and their |.|^2 of one pass's windows, one after the other.  The blocks (and the blocks of filter
    // outputs below) are recycled: a fresh half-megabyte vector per call is an mmap + a page fault per 4 KB
    struct Blk { std::vector<rfid_cf32> g; std::vector<float> m; };
    std::vector<Blk *> pool;
    std::shared_ptr<Blk> take_blk() {
      Blk *b;
      if (pool.empty()) b = new Blk; else { b = pool.back(); pool.pop_back(); }
      return std::shared_ptr<Blk>(b, [this](Blk *p) { pool.push_back(p); });
    }
    struct Win {
      int64_t start = 0;              // global decimated position of the opening sample
      int type = 0, len = 0;
      rfid_decode_result res;
      // in[i] - dc_est over the window and |.|^2 of those: [off, off + len) of the block the whole-chain pass fetched
      std::shared_ptr<Blk> blk;
      size_t off = 0;
      rfid_cf32 first, last;          // (the decoder's input is recognised by them)
    };
    bool on = false, flushed = false;
    bool flush_req = false;           // (gate-keyed) the end of the input was announced: carried out by the first gate call that brings
                                      // nothing new and can decide nothing (the library only knows the samples the gate was shown)
    int64_t gate_pos = 0;             // decimated samples the gate calls have consumed
    int64_t up_end = 0;               // (gate-keyed) global position behind the last sample uploaded
    // matched-filter output handed out and not (all) consumed by the gate yet: one block per rfid_mf_work call
    struct YBlk { int64_t y0 = 0; std::vector<rfid_cf32> v; };
    std::deque<YBlk> yq;
    int64_t y_end = 0;                // global position behind the last block
    const rfid_cf32 *y_at(int64_t pos) const {   // nullptr: not held
      for (const YBlk &b : yq) if (pos >= b.y0 && pos < b.y0 + (int64_t)b.v.size()) return &b.v[(size_t)(pos - b.y0)];
      return nullptr;
    }
    std::vector<std::vector<rfid_cf32>> y_pool;
    void y_drop_before(int64_t pos) {
      while (!yq.empty() && yq.front().y0 + (int64_t)yq.front().v.size() <= pos) {
        if (y_pool.size() < 8) y_pool.push_back(std::move(yq.front().v));
        yq.pop_front();
      }
    }
    void y_push(int64_t y0, const rfid_cf32 *v, size_t n) {
      yq.emplace_back();
      if (!y_pool.empty()) { yq.back().v = std::move(y_pool.back()); y_pool.pop_back(); }
      yq.back().y0 = y0;
      yq.back().v.assign(v, v + n);
      y_end = y0 + (int64_t)n;
    }
    std::deque<Win> wins;             // windows the gate has not (completely) handed out yet
    int emitted = 0;                  // samples of wins.front() already handed out (gate open)
    std::deque<Win> dq;               // windows handed out by the gate, waiting for the decoder (results only)
    int stall = 0;                    // gate calls in a row without progress and without new input
    int64_t coalesce = 65536;         // decimated samples that gather before a pass is submitted (rfid_lookahead_set_coalesce)
    bool patient = true;              // a gate call that can decide nothing answers (0, 0) once per arrival of new samples (the scheduler's
                                      // queues grow as needed); false: it decides at once (bounded buffers, rfid_lookahead_set_scheduler)
    bool tail_tried = false;          // a pass has gone over everything the device holds since the last new sample (and left the rest)
    // rfid_lookahead_set_consume_ahead: the gate takes everything it is shown (the device has it all), the windows follow when the
    // passes have found them -- the gate's input buffer never fills, whatever its size
    bool consume_ahead = false;
    bool need_arm = false;            // a window was handed out completely: the next one waits for the decoder / reader to arm the gate
    bool exact_open = false;          // the exact per-call scan (la_exact_step) has left a window open: it goes on until the window closes
    int *h_flag = nullptr;            // page-locked word the device writes behind a call's filter outputs (mf_upload_kernel)
    int *d_done = nullptr;            // ... and its counter of workgroups through
    int flag_seq = 0;
    // rfid_lookahead_set_late_outputs: a rfid_mf_work call hands out the filter outputs of the call BEFORE it (they are long
    // there), its own are fetched by the next call: no call waits for the device
    bool late = false;
    // the sets of outputs held back, oldest first: at most LATE_SLOTS - 1 when a call arrives (it launches into the free slot)
    struct Held { int64_t y0 = 0; int n = 0, off = 0, seq = 0, slot = 0; bool ready = false; };   // position, count, handed out, flag value, slot of h_y, seen
    static const int LATE_SLOTS = 3;
    std::deque<Held> held;
    int held_total() const { int t = 0; for (const Held &h : held) t += h.n - h.off; return t; }
    bool soft_done = false;           // ... and the held-back samples went through the sequential scan since the last input
    std::vector<float> last_m2;       // |.|^2 of what the last gate call wrote
    // scratch of one whole-chain pass
    char *h_pack = nullptr;           // page-locked: one packet per pass -- count, window records, results, gated samples, |.|^2 (gated_windows_kernel)
    rfid_cf32 *h_y = nullptr;         // page-locked: filter outputs of the calls (two halves)
    size_t h_cap = 0, h_ycap = 0;
    int n_hdr = 48;                   // windows the packet is sized for (follows what the calls hold)
  } la;
  rfid_window *d_swin = nullptr;  // one window
  int *d_scount = nullptr;
  rfid_decode_result *d_sres = nullptr;
  rfid_scores *d_sscores = nullptr;
  // last MF_HIST raw samples seen: the 24-sample filter history plus the up to 4 samples that wait for
  // their decimation group of 5 to complete
  static const int MF_HIST = NTAPS - 1 + DECIM - 1;
  rfid_cf32 mf_hist[MF_HIST];
  int64_t mf_seen = 0;

  // ---- batch plan ----
  int B = 0;        // traces the next pass processes (rfid_batch_set_streams), <= B_plan
  int B_plan = 0;   // traces the workspace was planned for; 0 = no plan
  int64_t max_raw = 0, y_stride = 0;
  float2 *d_y = nullptr;
  float2 *y_view = nullptr;       // (ymode stream) where the current pass's decimated samples lie instead of d_y
  float2 *y() const { return y_view ? y_view : d_y; }
  GateState *d_gstate = nullptr;
  rfid_window *d_wtab = nullptr, *d_flat = nullptr;
  int wmax = 0, flat_cap = 0;
  int *d_wcount = nullptr, *d_flat_count = nullptr;
  int *d_ticket = nullptr;   // RN16 pack counters of the decoder launches (two, used alternately)
  int ticket_flip = 0;
  rfid_decode_result *d_res = nullptr;
  int *d_sum = nullptr;                          // plans of few, long traces: a one-word summary per result for the statistics kernel
  const rfid_decode_result *sum_of = nullptr;    // (rfid_kernels.hpp, stats_summary()); sum_of: the result table they were last written for
  rfid_scores *d_scores = nullptr;
  rfid_stream_stats *d_stats = nullptr;
  const int64_t *d_lens = nullptr;  // of the last rfid_batch_mf
  int64_t last_n_raw = 0;
  int decode_grid = 0;
  int n_cus = 256;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ev_valid[5] = {false, false, false, false, false};
  // overlapped front end: matched filter on `stream`, gate scan on `stream2`, time-chunked
  static const int MAX_CHUNKS = 16;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_mf[MAX_CHUNKS + 1], ev_gate[2 * MAX_CHUNKS], ev_pass = nullptr, ev_front_end = nullptr;
  // Decoder + statistics of pass k beside the front end of pass k + 1 (rfid_batch_process, fused front end): a second set
  // of everything the front end writes and the decoder reads (the matched-filter output and the window / result tables), used
  // alternately; the decoder and the statistics kernel of a pass run on stream2 behind that pass's front end.  Allocated by
  // rfid_batch_plan when the device has the room (RFID_OVERLAP=0: never).
  struct ResultSet {
    float2 *d_y = nullptr;
    rfid_window *d_wtab = nullptr, *d_flat = nullptr;
    int *d_wcount = nullptr, *d_flat_count = nullptr;
    rfid_decode_result *d_res = nullptr;
    rfid_stream_stats *d_stats = nullptr;
  } alt;
  bool alt_have = false;
  void *plan_blk = nullptr, *alt_blk = nullptr;   // the two allocations behind the plan's buffers (rfid_batch_plan)
  hipEvent_t ev_fe_done = nullptr, ev_tail_done[2] = {nullptr, nullptr};
  bool tail_recorded[2] = {false, false};   // ev_tail_done[i] has been recorded (set i's decoder / statistics were enqueued on stream2)
  int set_idx = 0;                          // which of the two sets c->d_* currently name
  // (long-stream passes: only the matched-filter output alternates -- the filter of pass k + 1 beside the front end of pass k)
  void *alt_y_blk = nullptr;                // a second matched-filter output buffer alone (plans too small for a whole second set)
  hipEvent_t ev_y_free[2] = {nullptr, nullptr};
  bool y_recorded[2] = {false, false};
  int y_idx = 0;
  hipStream_t tail_stream = nullptr;        // where rfid_batch_decode / rfid_batch_stats enqueue (c->stream, or stream2 in an overlapped pass)
  int n_chunks_last = 0;   // > 0 when the last pass used the overlapped path
  int fused_last = 0;      // 1 when the last rfid_batch_process pass used front_end_fused_kernel
  float front_ms = 0.0f;
};

namespace {

void sio_free(rfid_ctx *c);   // (whole-chain streaming, below)
int sio_process(rfid_ctx *c, int b, int64_t n_new, bool flush);
int sio_submit(rfid_ctx *c, int b, int64_t n_new, bool flush);
int sio_collect(rfid_ctx *c);
void la_free(rfid_ctx *c);    // (look-ahead of the per-block calls)
int la_mf_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_produced);
int la_gate_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_consumed, int *n_written);
int la_gate_swallow(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_consumed, int *n_written);
bool la_decoder_result(rfid_ctx *c, const rfid_cf32 *in, int wlen, int type, rfid_decode_result *r);

int fail(rfid_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
  if (c) {
    if (e != hipSuccess)
      snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
    else
      snprintf(c->err, sizeof(c->err), "%s", what);
  }
  return code;
}

struct KnobEntry { const char *name, *env; int RfidKnobs::*field; int lo, hi; };
const KnobEntry g_knob_table[] = {
  {"long_stream", "RFID_LONG_STREAM", &RfidKnobs::long_stream, 0, 2},
  {"overlap", "RFID_OVERLAP", &RfidKnobs::overlap, 0, 2},
  {"ls_calibrate", "RFID_LS_CALIBRATE", &RfidKnobs::ls_calibrate, 0, 1},
  {"ls_debug", "RFID_LS_DEBUG", &RfidKnobs::ls_debug, 0, 1},
  {"la_profile", "RFID_LA_PROFILE", &RfidKnobs::la_profile, 0, 1},
  {"front_unfused", "RFID_FRONT_UNFUSED", &RfidKnobs::front_unfused, 0, 1},
  {"front_chunks", "RFID_FRONT_CHUNKS", &RfidKnobs::front_chunks, 1, rfid_ctx::MAX_CHUNKS},
  {"fsm_lanes_min", "RFID_LS2_FSM_LANES_MIN", &RfidKnobs::fsm_lanes_min, -1, 1 << 30},
  {"dc_rounds", "RFID_LS2_DC_ROUNDS", &RfidKnobs::dc_rounds, -1, 64},
  {"la_upload_kernel", "RFID_LA_UPLOAD_KERNEL", &RfidKnobs::la_upload_kernel, 0, 1},
  {"front_lds_kb", "RFID_LS_FRONT_LDS_KB", &RfidKnobs::front_lds_kb, -1, 64},
};
int clamp_int(long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); }
// the environment, once per context: values out of range are clamped, anything that is not a number is ignored
void knobs_from_env(RfidKnobs &k) {
  for (const KnobEntry &e : g_knob_table) {
    const char *v = getenv(e.env);
    if (!v || !*v) continue;
    char *end = nullptr;
    const long x = strtol(v, &end, 10);
    if (end == v) continue;
    k.*(e.field) = clamp_int(x, e.lo, e.hi);
  }
}

#define HIPCHK(c, call)                                                     \
  do {                                                                      \
    hipError_t e__ = (call);                                                \
    if (e__ != hipSuccess) return fail((c), RFID_ERR_HIP, #call, e__);      \
  } while (0)

#define HIPCHK_T(c, call)                                                   \
  do {                                                                      \
    hipError_t e__ = (call);                                                \
    if (e__ != hipSuccess) { (c)->y_touched = true; return fail((c), RFID_ERR_HIP, #call, e__); } \
  } while (0)

int grow(rfid_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return RFID_OK;
  if (b.p) HIPCHK(c, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes < 4096 ? 4096 : bytes;
  HIPCHK(c, hipMalloc(&b.p, want));
  b.cap = want;
  return RFID_OK;
}

int grow_pinned(rfid_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return RFID_OK;
  if (b.p) HIPCHK(c, hipHostFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  const size_t want = bytes < 65536 ? 65536 : bytes;
  HIPCHK(c, hipHostMalloc(&b.p, want, hipHostMallocDefault));
  b.cap = want;
  return RFID_OK;
}

// derive the sample counts exactly as the block constructors do and check that they are
// the ones the kernels were written for (gate_impl.cc:48-53, tag_decoder_impl.cc:60)
bool params_supported(const rfid_params &p) {
  if (p.decim != DECIM || p.n_taps != NTAPS) return false;
  if (p.fixed_q < 0 || p.fixed_q > 15) return false;
  const float tag_bit_d = rfidh::tag_bit_d();
  const int t1 = (int)(240 * (p.sample_rate / pow(10, 6)));
  const int pw = (int)(12 * (p.sample_rate / pow(10, 6)));
  const int nb = (int)(tag_bit_d * (p.sample_rate / pow(10, 6)));
  const int win = (int)(250 * (p.sample_rate / pow(10, 6)));
  const int dc = (int)(120 * (p.sample_rate / pow(10, 6)));
  const float nbf = rfidh::n_samples_tag_bit(p.sample_rate);
  return t1 == T1_SAMPLES && pw / 2 == PW_HALF && nb == 10 && win == WIN_LEN && dc == DC_LEN &&
         nbf == 10.0f && (129 + 6) * nb + 2 * nb == EPC_WIN && (17 + 6) * nb + 2 * nb == RN16_WIN;
}

void compute_t_cand(float *t_cand, int sample_rate) { rfidh::t_candidates(t_cand, sample_rate); }

void init_reader_state(rfid_ctx *c) {  // global_vars.cc:34-54
  memset(&c->rs, 0, sizeof(c->rs));
  c->rs.status = RFID_RUNNING;
  c->rs.gen2_logic_status = RFID_START;
  c->rs.gate_status = RFID_GATE_SEEK_RN16;
  c->rs.decoder_status = RFID_DECODE_RN16;
  c->rs.max_slot_number = (int)pow(2, c->prm.fixed_q);
  c->rs.cur_inventory_round = 1;
  c->rs.cur_slot_number = 1;
}

void free_plan(rfid_ctx *c) {
  if (c->plan_blk) (void)hipFree(c->plan_blk);
  if (c->alt_blk) (void)hipFree(c->alt_blk);
  if (c->alt_y_blk) (void)hipFree(c->alt_y_blk);
  if (c->d_sum) (void)hipFree(c->d_sum);
  c->d_sum = nullptr; c->sum_of = nullptr;
  c->plan_blk = nullptr; c->alt_blk = nullptr; c->alt_y_blk = nullptr;
  c->y_recorded[0] = c->y_recorded[1] = false;
  c->y_idx = 0;
  c->y_touched = false;
  c->d_y = nullptr; c->d_gstate = nullptr; c->d_wtab = nullptr; c->d_flat = nullptr;
  c->d_wcount = nullptr; c->d_flat_count = nullptr; c->d_res = nullptr; c->d_scores = nullptr;
  c->d_stats = nullptr;
  c->alt = rfid_ctx::ResultSet();
  c->alt_have = false;
  c->tail_recorded[0] = c->tail_recorded[1] = false;
  c->set_idx = 0;
  c->B = 0;
  c->B_plan = 0;
}

// every pass that is not itself an overlapped one, and every look at the results, first lets the context's main stream wait
// for what overlapped passes left on stream2
int join_tails(rfid_ctx *c) {
  for (int i = 0; i < 2; ++i)
    if (c->tail_recorded[i]) {
      if (hipStreamWaitEvent(c->stream, c->ev_tail_done[i], 0) != hipSuccess) return RFID_ERR_HIP;
      c->tail_recorded[i] = false;
    }
  return RFID_OK;
}

// slot/round roll-over of tag_decoder_impl.cc:330-343 / :369-383 (and :269-288)
void next_slot(rfid_reader_state &rs) {
  if (rs.cur_slot_number > rs.max_slot_number) {
    rs.cur_slot_number = 1;
    rs.cur_inventory_round += 1;
    rs.gen2_logic_status = RFID_SEND_QUERY;
  } else {
    rs.gen2_logic_status = RFID_SEND_QUERY_REP;
  }
}

}  // namespace


// ======================================================================================
// long-stream front end (kernels: rfid_ls2.hpp; launch list: rfid_ls2_enqueue.hpp)
// ======================================================================================
namespace {
double ls_now_ms() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
}
// Automatic choice between the two front ends (mode 1): the fused front end runs up to 1024 traces side by side at
// seq_ns_per_sample per decimated sample of the LONGEST trace; the long-stream front end costs a fixed string of short
// launches plus ls_ns_per_sample per decimated sample of ALL traces.  The three numbers are measured on the device at
// hand when the context is created (rfid_ctx_create: ls_calibrate).
void ls_calibrate_lazily(rfid_ctx *c);
bool ls_pays_off(rfid_ctx *c, int B, int64_t n_dec) {
  for (int pass = 0; pass < 2; ++pass) {
    const double t_seq = 1e-6 * c->seq_ns_per_sample * (double)n_dec * (double)((B + 1023) / 1024);
    const double t_ls = c->ls_fixed_ms + 1e-6 * c->ls_ns_per_sample * (double)B * (double)n_dec;
    // The built-in numbers (one MI355X) decide the clear cases -- one long trace, a thousand traces.  Only a shape near
    // the crossover is worth the ~60 ms of measuring both front ends on the device at hand, once per device and process.
    if (pass == 0 && !c->ls_calibrated && t_ls > 0.33 * t_seq && t_ls < 3.0 * t_seq) { ls_calibrate_lazily(c); continue; }
    return t_ls < 0.9 * t_seq;
  }
  return false;
}
bool ls_applicable(rfid_ctx *c, int B, int64_t n_dec) {
  if (c->ls_mode == 0 || B > 4096) return false;
  if (ls2_geometry(B, n_dec).P == 0) return false;
  return c->ls_mode == 2 || ls_pays_off(c, B, n_dec);
}
// for the work-space reservation of rfid_batch_plan: could a pass of this shape take the long-stream front end?  (Never
// measures anything: a plan -- every rfid_stream_begin makes one -- must not cost a calibration.)
bool ls_may_apply(const rfid_ctx *c, int B, int64_t n_dec) {
  if (c->ls_mode == 0 || B > 4096) return false;
  if (ls2_geometry(B, n_dec).P == 0) return false;
  const double t_seq = 1e-6 * c->seq_ns_per_sample * (double)n_dec * (double)((B + 1023) / 1024);
  const double t_ls = c->ls_fixed_ms + 1e-6 * c->ls_ns_per_sample * (double)B * (double)n_dec;
  return c->ls_mode == 2 || t_ls < 3.0 * t_seq;
}
size_t ls_workspace_bytes(int B, int64_t n_dec, int64_t y_stride, int wmax) {
  const Ls2Geometry g = ls2_geometry(B, n_dec);
  return g.P ? ls2_layout(g, B, y_stride, wmax).total : 0;
}

int ls_calibrate(rfid_ctx *c);
void ls_note_last_pass(rfid_ctx *c);

struct LsOpts {
  bool carry = false;       // the trace starts from c->d_gstate[trace] (streaming) instead of the fresh gate, and the state
                            // after the last processed piece is written back there
  bool hold_last = false;   // leave each trace's last piece unprocessed (streaming: whatever follows the last idle cut
                            // waits for more samples)
  bool force = false;       // run even when no trace could be cut more than once
  bool ahead = false;       // (fused first pass) its launches on stream2 and in the work space the pass before did not use; the rest of the
                            // list on the main stream behind them
  // the fused first pass (ls2_front_kernel): the raw samples in HBM -- the matched filter runs inside the front end's first
  // launch and writes c->d_y; nullptr: c->d_y holds the filter's output already
  const void *raw = nullptr; int64_t raw_stride = 0;
};
// Enqueues one pass of the front end over c->d_y (n_dec decimated samples per trace, c->d_lens).  *enqueued = 0: not
// applicable here (traces too short, no work space) -- nothing was launched.  Whether the pass produced the window
// tables is known on the device only (Ls2Ctl::ok); the caller enqueues the sequential scan behind it with
// GateArgs::skip_if = &ctl->ok, or synchronises and looks at c->ls2_host.
// A look at the last pass's control block, if that pass is over (called before a pass enqueues anything): one that ran
// out of rounds -- not one that found no cut -- makes the following passes enqueue the full number of rounds.
void ls_note_last_pass(rfid_ctx *c) {
  if (!c->d_ls2_ctl || !c->ls2_host) return;
  if (hipStreamQuery(c->stream) == hipSuccess) {
    const Ls2Ctl &k = *c->ls2_host;
    if (!c->ls2_generous && k.fail == 0 && k.ok == 0 && k.n_pieces > 0) c->ls2_generous = true;
  }
  (void)hipGetLastError();
}
int ls_enqueue(rfid_ctx *c, int64_t n_dec, const LsOpts &opt, int *enqueued) {
  *enqueued = 0;
  c->d_ls2_ctl = nullptr;
  const Ls2Geometry geo = ls2_geometry(c->B, n_dec);
  if (geo.P == 0 || c->B > 65535) return RFID_OK;
  const Ls2Layout L = ls2_layout(geo, c->B, c->y_stride, c->wmax);
  const bool ahead = opt.ahead && opt.raw && c->ls2_ws.cap >= L.total && c->ls2_ws_alt.cap >= L.total;
  if (opt.ahead && !ahead) return RFID_OK;   // (the caller takes the pass without the second stream)
  if (L.total > c->ls2_ws.cap) {
    // (normally reserved by rfid_batch_plan; a pass that needs it all the same must not fail on a full device)
    if (c->ls2_ws.p) { (void)hipFree(c->ls2_ws.p); c->ls2_ws.p = nullptr; c->ls2_ws.cap = 0; }
    if (hipMalloc(&c->ls2_ws.p, L.total) != hipSuccess) { (void)hipGetLastError(); c->ls2_ws.p = nullptr; return RFID_OK; }
    c->ls2_ws.cap = L.total;
  }
  if (!c->ls2_host) {
    if (hipHostMalloc((void **)&c->ls2_host, sizeof(Ls2Ctl) + 64, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError(); c->ls2_host = nullptr; return RFID_OK;
    }
    memset(c->ls2_host, 0, sizeof(Ls2Ctl) + 64);
  }
  Ls2Args a;
  memset(&a, 0, sizeof(a));
  a.y = c->y(); a.y_stride = c->y_stride; a.lens = c->d_lens; a.n_dec = n_dec; a.n_streams = c->B;
  if (ahead) std::swap(c->ls2_ws, c->ls2_ws_alt);   // (alternating with the matched-filter output buffers: nothing below returns without a pass)
  ls2_bind(a, (char *)c->ls2_ws.p, L, geo);
  a.wtab = c->d_wtab; a.wmax = c->wmax; a.wcount = c->d_wcount; a.flat = c->d_flat; a.flat_count = c->d_flat_count; a.flat_cap = c->flat_cap;
  a.carry = opt.carry ? c->d_gstate : nullptr; a.carry_out = opt.carry ? c->d_gstate : nullptr;
  a.hold_last = opt.hold_last ? 1 : 0; a.force = opt.force ? 1 : 0;
  if (opt.raw) {
    a.fused = 1;
    a.raw = (const float2 *)opt.raw; a.raw_stride = opt.raw_stride;
    a.raw_vec_ok = ((opt.raw_stride & 1) == 0 && (((uintptr_t)opt.raw) & 15) == 0) ? 1 : 0;
    a.y_w = c->d_y;
  }
  ls2_stream = ahead ? c->stream2 : c->stream;
  // The first pass's one-wave workgroups would take every wave slot of the chip; launched beside the rest of the pass before
  // (re-run rounds, state machine, dc_est: a few waves each, one launch waiting for the other) they ask for 10 KB of LDS
  // they do not use -- 12 instead of 16 of them per CU, and the small launches find room at once: configs[2] back to back
  // 8.17 -> 7.6 ms, each pass waited for unchanged (profiles/r05/ls2_fused_front.txt, 7.).  Not for short streams, whose
  // first pass is a few rounds of waves and on the critical path (configs[3]: 10 183 pieces).
  {
    const int kb = (c->knobs.front_lds_kb >= 0) ? c->knobs.front_lds_kb : ((geo.NS >= 32768) ? 10 : 0);
    ls2_front_lds = (unsigned)kb * 1024u;
  }
  a.keep_flat_count = ahead ? 1 : 0;
  {   // test hook: from how many possible heads on the state machine takes its one-lane-per-unit form (default 8192)
    static const int lanes_min_default = ls2_fsm_lanes_min();
    ls2_fsm_lanes_min() = (c->knobs.fsm_lanes_min >= 0) ? c->knobs.fsm_lanes_min : lanes_min_default;
  }
  auto mark = [](void *p, int pt) {
    rfid_ctx *cc = (rfid_ctx *)p;
    if (pt == 2) {
      // the fused first pass is enqueued: the rest of the list goes to the main stream, behind it (and behind the pass before,
      // whose decoder reads the list counters that this pass's window assembly counts up from zero)
      if (ls2_stream == cc->stream2) {
        if (hipEventRecord(cc->ev_fe_done, cc->stream2) != hipSuccess || hipStreamWaitEvent(cc->stream, cc->ev_fe_done, 0) != hipSuccess ||
            hipMemsetAsync(cc->d_flat_count, 0, 2 * sizeof(int), cc->stream) != hipSuccess) cc->ls2_mark_failed = true;
        ls2_stream = cc->stream;
      }
      return;
    }
  };
  c->ls2_mark_failed = false;
  ls2_enqueue(a, true, c->ls2_rounds, c->ls2_generous, c->knobs.dc_rounds, ahead ? +mark : nullptr, c);
  ls2_stream = c->stream;
  HIPCHK(c, hipGetLastError());
  if (c->ls2_mark_failed) return fail(c, RFID_ERR_HIP, "long-stream front end: stream hand-over");
  // (the control block and, right behind it, consumed[0])
  HIPCHK(c, hipMemcpyAsync(c->ls2_host, a.ctl, sizeof(Ls2Ctl) + sizeof(int), hipMemcpyDeviceToHost, c->stream));
  c->d_ls2_ctl = a.ctl;
  c->ls2_P = geo.P;
  *enqueued = 1;
  if (c->knobs.ls_debug) {   // what the rounds did: per-round counts, margins against the shifts they had to cover
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const Ls2Ctl &k = *c->ls2_host;
    fprintf(stderr, "[ls2] pieces %d P %d fail %d ok %d | avg:", k.n_pieces, geo.P, k.fail, k.ok);
    for (int r = 0; r <= c->ls2_rounds[0]; ++r) fprintf(stderr, " %d/%d", k.avg_count[r], k.avg_list[r]);
    fprintf(stderr, " | fsm:");
    for (int r = 0; r <= c->ls2_rounds[1]; ++r) fprintf(stderr, " %d", k.fsm_count[r]);
    fprintf(stderr, " | dc:");
    for (int r = 0; r <= c->ls2_rounds[2]; ++r) fprintf(stderr, " %d", k.dc_count[r]);
    fprintf(stderr, " | units %d windows %d\n", k.n_units, k.n_windows);
    std::vector<Ls2AvgRun> ar((size_t)geo.NS);
    std::vector<int> aT((size_t)geo.NS);
    std::vector<Ls2Piece> pc((size_t)geo.NS);
    HIPCHK(c, hipMemcpy(ar.data(), a.arun, sizeof(Ls2AvgRun) * ar.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(aT.data(), a.aT, sizeof(int) * aT.size(), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(pc.data(), a.piece, sizeof(Ls2Piece) * pc.size(), hipMemcpyDeviceToHost));
    long hist_m[8] = {0}, hist_d[8] = {0};
    int shown = 0;
    {
      long hist_len[6] = {0}; int max_len = 0; long n_wide = 0, n_over = 0;
      for (int i = 0; i < geo.NS; ++i) {
        const int len = pc[(size_t)i].len;
        if (len <= 0) continue;
        if (len > max_len) max_len = len;
        const int q = (int)(4.0 * len / geo.P);   // quarters of the nominal length
        hist_len[q < 2 ? 0 : (q < 4 ? 1 : (q < 6 ? 2 : (q < 8 ? 3 : (q < 16 ? 4 : 5))))]++;
        if (ar[(size_t)i].wide & 2) n_wide++;
        if (ar[(size_t)i].wide & 4) n_over++;
      }
      fprintf(stderr, "[ls2] piece lengths in P (< 0.5, < 1, < 1.5, < 2, < 4, more): %ld %ld %ld %ld %ld %ld, longest %d; last run wide: %ld, exact end put in: %ld\n",
              hist_len[0], hist_len[1], hist_len[2], hist_len[3], hist_len[4], hist_len[5], max_len, n_wide, n_over);
    }
    for (int i = 0; i < geo.NS; ++i) {
      if (pc[(size_t)i].len <= 0) continue;
      uint32_t u; memcpy(&u, &ar[(size_t)i].s, 4);
      const long D = labs((long)aT[(size_t)i] - (long)(int)u);
      const int m = ar[(size_t)i].margin;
      int bm = 0; for (long v = m; v > 0 && bm < 7; v >>= 3) bm++;
      int bd = 0; for (long v = D; v > 0 && bd < 7; v >>= 3) bd++;
      hist_m[bm]++; hist_d[bd]++;
      if (shown < 12 && (i % 2500) == 1) { fprintf(stderr, "[ls2]   piece %d pos %d len %d s %.9g margin %d |D| %ld\n", i, pc[(size_t)i].pos0, pc[(size_t)i].len, ar[(size_t)i].s, m, D); shown++; }
    }
    fprintf(stderr, "[ls2] margin histogram (0, <8, <64, <512, <4096, <32768, <262144, more):");
    for (int b = 0; b < 8; ++b) fprintf(stderr, " %ld", hist_m[b]);
    fprintf(stderr, "\n[ls2] |D| histogram (same bins):");
    for (int b = 0; b < 8; ++b) fprintf(stderr, " %ld", hist_d[b]);
    fprintf(stderr, "\n");
    {   // dc_est: where the units' true starts lay in their windows (offset from the centre of the latest run), what is settled
      const size_t NHh = (size_t)c->B * (size_t)geo.max_bc;
      std::vector<int> dT(2 * NHh), dcen(2 * NHh), dstat(NHh);
      HIPCHK(c, hipMemcpy(dT.data(), a.dT, sizeof(int) * dT.size(), hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(dcen.data(), a.dcen, sizeof(int) * dcen.size(), hipMemcpyDeviceToHost));
      HIPCHK(c, hipMemcpy(dstat.data(), a.dstat, sizeof(int) * dstat.size(), hipMemcpyDeviceToHost));
      long units = 0, settled = 0, hd[2][8] = {{0}};
      for (size_t t = 0; t < NHh; ++t) {
        if (!(dstat[t] & 4)) continue;
        units++;
        if ((dstat[t] & 3) == 3) settled++;
        for (int q = 0; q < 2; ++q) {
          const long D = labs((long)dT[2 * t + q] - (long)dcen[2 * t + q]);
          int bd = 0; for (long v = D; v > 0 && bd < 7; v >>= 3) bd++;
          hd[q][bd]++;
        }
      }
      fprintf(stderr, "[ls2] dc_est: %ld units, %ld settled, finishing walk took %d, rounds used %d of %d enqueued\n", units, settled, k.dc_finished, k.dc_rounds, c->ls2_rounds[2] + 1);
      fprintf(stderr, "[ls2] finishing walk (trace 0): %d turns; units settled per turn (0, 1, 2-3, 4-7, 8-15, 16-31, 32-63, more): %d %d %d %d %d %d %d %d; first misses inside / outside twice the windows' reach: %d / %d\n",
              k.fin_turns, k.fin_reach[0], k.fin_reach[1], k.fin_reach[2], k.fin_reach[3], k.fin_reach[4], k.fin_reach[5], k.fin_reach[6], k.fin_reach[7], k.fin_far[0], k.fin_far[1]);
      for (int q = 0; q < 2; ++q) {
        fprintf(stderr, "[ls2] dc_est %s |true start - centre of the latest run| (0, <8, <64, <512, <4096, <32768, <262144, more):", q ? "im" : "re");
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %ld", hd[q][b]);
        fprintf(stderr, "\n");
      }
    }
  }
  return RFID_OK;
}
}  // namespace

namespace {
// The three numbers of ls_pays_off measured on the device at hand: a Gen2 trace of 64 slots (0.97 M raw samples) alone,
// and 16 noise replicas of one four times as long, are synthesised in HBM and run through both front ends (whole
// passes, decoder and statistics included on both sides).  ~30 ms, once per context; any failure leaves the defaults.
int ls_calibrate(rfid_ctx *c) {
  const int saved_mode = c->ls_mode;
  rfid_synth_gen2_params p;
  memset(&p, 0, sizeof(p));
  p.leak_re = 0.7648f; p.leak_im = 0.6442f; p.h_re[0] = 0.06f; p.h_im[0] = 0.03f; p.n_tags = 1; p.tail_us = 200;
  const int n_big = 256, B_big = 16;
  std::vector<rfid_synth_slot> slots((size_t)n_big);
  for (int i = 0; i < n_big; ++i) {
    rfid_synth_slot &s = slots[(size_t)i];
    memset(&s, 0, sizeof(s));
    s.cmd = (i == 0) ? 0 : 1; s.q = 0; s.n_tags = 1; s.has_epc = 1; s.tag[0] = 0;
    s.rn16[0] = (uint16_t)(0x5a5a ^ (i * 2654435761u)); s.ack = s.rn16[0];
    s.rn16_off_raw = 500; s.epc_off_raw = 500;
    for (int k = 0; k < 4; ++k) s.epc[k] = 0x3000f00fu * (uint32_t)(i + k + 1);
  }
  int64_t L[2] = {0, 0};
  if (rfid_synth_gen2_size(&p, slots.data(), n_big / 4, &L[0]) || rfid_synth_gen2_size(&p, slots.data(), n_big, &L[1])) return RFID_OK;
  void *d_base = nullptr, *d_many = nullptr;
  const int64_t stride = (L[1] + 1) & ~1LL;
  if (hipMalloc(&d_base, sizeof(float2) * (size_t)stride) != hipSuccess ||
      hipMalloc(&d_many, sizeof(float2) * (size_t)stride * B_big) != hipSuccess) {
    (void)hipGetLastError();
    if (d_base) (void)hipFree(d_base);
    return RFID_OK;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  double t_seq[2] = {0, 0}, t_ls[2] = {0, 0};
  for (int k = 0; k < 2 && ok; ++k) {
    int64_t n = 0;
    const int B = k ? B_big : 1;
    ok = rfid_synth_gen2(c, &p, slots.data(), k ? n_big : n_big / 4, d_base, stride, 0.0f, 77u, 0, &n) == RFID_OK && n == L[k];
    ok = ok && rfid_synth_replicas(c, d_base, L[k], d_many, stride, B, 0.003f, 78u, 0) == RFID_OK;
    ok = ok && rfid_batch_plan(c, B, L[k]) == RFID_OK;
    for (int mode = 0; mode <= 2 && ok; mode += 2) {
      c->ls_mode = mode;
      float best = 1e30f;
      for (int rep = 0; rep < 3 && ok; ++rep) {   // (the first pass warms up)
        ok = hipEventRecord(e0, c->stream) == hipSuccess && rfid_batch_process(c, d_many, stride, L[k], nullptr, 0) == RFID_OK &&
             hipEventRecord(e1, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        float ms = 0.0f;
        ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        if (rep > 0 && ms < best) best = ms;
      }
      if (mode == 2 && ok) ok = c->d_ls2_ctl != nullptr && c->ls2_host->ok != 0;   // (the front end must have taken the traces)
      (mode ? t_ls : t_seq)[k] = best;
    }
  }
  c->ls_mode = saved_mode;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipStreamSynchronize(c->stream);
  (void)hipFree(d_base);
  (void)hipFree(d_many);
  free_plan(c);
  c->d_ls2_ctl = nullptr;
  if (!ok) { (void)hipGetLastError(); return RFID_OK; }
  const double n0 = (double)(L[0] / DECIM), n1 = (double)(L[1] / DECIM), N1 = n1 * B_big;
  const double slope = (t_ls[1] - t_ls[0]) / (N1 - n0);          // ms per decimated sample of all traces
  const double fixed = t_ls[0] - slope * n0;
  if (slope > 0.0 && fixed > 0.0 && t_seq[1] > 0.0) {
    // long traces need more re-run rounds than these (their rounding drift grows with the square root of the length):
    // a quarter more on the slope
    c->ls_ns_per_sample = 1.25 * slope * 1e6;
    c->ls_fixed_ms = fixed;
    c->seq_ns_per_sample = t_seq[1] / n1 * 1e6;   // (sixteen traces side by side take the time of one)
    c->ls_measured = true;
  }
  c->ls_calibrated = true;
  if (c->knobs.ls_debug)
    fprintf(stderr, "[ls2] calibration: fused %.3f / %.3f ms, long-stream %.3f / %.3f ms for 1 x %.0f / %d x %.0f samples -> %.2f ns/sample of the longest trace vs %.3f ms + %.4f ns/sample of all\n",
            t_seq[0], t_seq[1], t_ls[0], t_ls[1], n0, B_big, n1, c->seq_ns_per_sample, c->ls_fixed_ms, c->ls_ns_per_sample);
  return RFID_OK;
}

// The cost model's numbers for the device of `c`, measured when a decision first needs them (ls_pays_off): on a context of
// their own -- the measurement plans and frees batches -- and kept per device for the life of the process.
// RFID_LS_CALIBRATE=0 keeps the built-in numbers (profiled runs: a fixed choice of the front end).
struct LsCalCache { bool done = false, ok = false; double fixed_ms = 0, ns_all = 0, ns_seq = 0; };
LsCalCache g_ls_cal[64];
pthread_mutex_t g_ls_cal_mu = PTHREAD_MUTEX_INITIALIZER;
void ls_calibrate_lazily(rfid_ctx *c) {
  c->ls_calibrated = true;                 // (whatever comes of it: tried once per context)
  if (!c->knobs.ls_calibrate || c->device < 0 || c->device >= 64) return;
  pthread_mutex_lock(&g_ls_cal_mu);
  LsCalCache &k = g_ls_cal[c->device];
  if (!k.done) {
    k.done = true;
    rfid_ctx *t = nullptr;
    if (rfid_ctx_create(&c->prm, c->device, &t) == RFID_OK && t) {
      t->ls_calibrated = true;             // (its own decisions use what it has)
      t->ls_mode = 1;
      if (ls_calibrate(t) == RFID_OK && t->ls_measured) {   // (a measurement that bailed out leaves nothing to cache: the built-in numbers stay)
        k.ok = true; k.fixed_ms = t->ls_fixed_ms; k.ns_all = t->ls_ns_per_sample; k.ns_seq = t->seq_ns_per_sample;
      }
      (void)rfid_ctx_destroy(t);
      (void)hipSetDevice(c->device);
    }
  }
  if (k.ok) { c->ls_fixed_ms = k.fixed_ms; c->ls_ns_per_sample = k.ns_all; c->seq_ns_per_sample = k.ns_seq; }
  pthread_mutex_unlock(&g_ls_cal_mu);
}
}  // namespace

extern "C" {

const char *rfid_version(void) { return "rfid_mi355x 0.1 (gfx950)"; }

const char *rfid_strerror(int s) {
  switch (s) {
    case RFID_OK: return "ok";
    case RFID_ERR_INVALID: return "invalid argument";
    case RFID_ERR_NO_DEVICE: return "no usable gfx950 device";
    case RFID_ERR_HIP: return "HIP runtime error";
    case RFID_ERR_UNSUPPORTED: return "unsupported parameters";
    case RFID_ERR_CAPACITY: return "buffer or workspace too small";
    case RFID_ERR_STATE: return "call not valid in this state";
    default: return "unknown status";
  }
}

const char *rfid_last_error(const rfid_ctx *ctx) { return ctx ? ctx->err : "null context"; }

int rfid_params_default(rfid_params *p) {
  if (!p) return RFID_ERR_INVALID;
  p->sample_rate = 400000;      // int(adc_rate/decim), apps/reader.py:76
  p->decim = DECIM;
  p->n_taps = NTAPS;
  p->fixed_q = 0;               // global_vars.h:72
  p->max_num_queries = 1000;    // global_vars.h:76
  p->number_unique_tags = 100;  // global_vars.h:100
  return RFID_OK;
}

int rfid_ctx_create(const rfid_params *p, int device, rfid_ctx **out) {
  if (!p || !out) return RFID_ERR_INVALID;
  *out = nullptr;
  if (!params_supported(*p)) return RFID_ERR_UNSUPPORTED;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return RFID_ERR_NO_DEVICE;
  if (device < 0 || device >= n_dev) return RFID_ERR_NO_DEVICE;
  // (the device's properties are looked up once per process: the query costs milliseconds)
  static pthread_mutex_t prop_mu = PTHREAD_MUTEX_INITIALIZER;
  static struct { bool known, gfx950; int cus; } dev_info[64];
  bool is_gfx950 = false;
  if (device < 64) {
    pthread_mutex_lock(&prop_mu);
    if (!dev_info[device].known) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        dev_info[device].known = true;
        dev_info[device].gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        dev_info[device].cus = prop.multiProcessorCount;
      }
    }
    is_gfx950 = dev_info[device].known && dev_info[device].gfx950;
    pthread_mutex_unlock(&prop_mu);
  }
  if (!is_gfx950) return RFID_ERR_NO_DEVICE;
  rfid_ctx *c = new (std::nothrow) rfid_ctx();
  if (c) c->n_cus = dev_info[device].cus;
  if (!c) return RFID_ERR_CAPACITY;
  c->prm = *p;
  c->device = device;
  c->err[0] = 0;
  compute_t_cand(c->t_cand, p->sample_rate);
  knobs_from_env(c->knobs);   // the one place where RFID_* variables are read
  if (c->knobs.la_profile) g_la_on = true;
  c->ls_mode = c->knobs.long_stream;
  init_reader_state(c);
  memset(c->mf_hist, 0, sizeof(c->mf_hist));
  int rc = RFID_OK;
  do {
    if (hipSetDevice(device) != hipSuccess) { rc = RFID_ERR_NO_DEVICE; break; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = RFID_ERR_HIP; break; }
    for (int i = 0; i < 5; ++i)
      if (hipEventCreate(&c->ev[i]) != hipSuccess) { rc = RFID_ERR_HIP; break; }
    if (rc) break;
    if (hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess) { rc = RFID_ERR_HIP; break; }
    for (int i = 0; i <= rfid_ctx::MAX_CHUNKS; ++i) c->ev_mf[i] = nullptr;
    for (int i = 0; i < 2 * rfid_ctx::MAX_CHUNKS; ++i) c->ev_gate[i] = nullptr;
    // (ev_mf / ev_gate: the time-chunked front end's, RFID_FRONT_CHUNKS -- made when that path first runs)
    if (!rc && (hipEventCreate(&c->ev_pass) != hipSuccess || hipEventCreate(&c->ev_front_end) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_fe_done, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_tail_done[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_tail_done[1], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_y_free[0], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&c->ev_y_free[1], hipEventDisableTiming) != hipSuccess))
      rc = RFID_ERR_HIP;
    if (rc) break;
    {
      // the context's small device buffers, carved out of one zeroed allocation (256-byte slots)
      char *blk = nullptr;
      const size_t slot = 256, n_slots = 4 + 1 + 1 + 1 + 1 + 1 + 1;   // GateState: 1 016 bytes = 4 slots
      if (hipMalloc((void **)&blk, slot * n_slots) != hipSuccess || hipMemset(blk, 0, slot * n_slots) != hipSuccess) { rc = RFID_ERR_HIP; c->d_small = blk; break; }
      static_assert(sizeof(GateState) <= 4 * 256 && sizeof(rfid_scores) <= 256 && sizeof(rfid_decode_result) <= 256, "slots");
      c->d_small = blk;
      c->d_gate1 = (GateState *)blk;
      c->d_io = (int *)(blk + 4 * slot);
      c->d_swin = (rfid_window *)(blk + 5 * slot);
      c->d_scount = (int *)(blk + 6 * slot);
      c->d_ticket = (int *)(blk + 7 * slot);
      c->d_sres = (rfid_decode_result *)(blk + 8 * slot);
      c->d_sscores = (rfid_scores *)(blk + 9 * slot);
    }
  } while (0);
  if (rc != RFID_OK) { rfid_ctx_destroy(c); return rc; }
  // (the cost model behind the automatic choice of the front end is measured on this device when a shape near the
  // crossover first asks for it: ls_calibrate_lazily)
  *out = c;
  return RFID_OK;
}

int rfid_ctx_destroy(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  if (c->knobs.la_profile && (g_la_n[0] || g_la_n[1]))
    fprintf(stderr, "[la] mf_work %ld calls %.2f ms (upload queued %.2f, the call's filter enqueued %.2f, wait for its outputs %.2f, passes submitted from here %.2f) | "
            "gate_work %ld calls %.2f ms | decoder_work %ld calls %.2f ms | reader_work_tx %ld calls %.2f ms | lookahead_enable %.2f ms | "
            "%ld passes: collected %.2f ms (waiting for the device %.2f, fetching windows %.2f), submitted %.2f ms (the pass's filter %.2f, front end's launch list %.2f, "
            "decoder %.2f, packet of results %.2f)\n",
            g_la_n[0], g_la_t[0], g_la_t[4], g_la_t[5], g_la_t[8], g_la_t[6], g_la_n[1], g_la_t[1], g_la_n[2], g_la_t[2], g_la_n[3], g_la_t[3], g_la_t[9],
            g_la_n[10], g_la_t[10], g_la_t[7], g_la_t[12], g_la_t[11], g_la_t[14], g_la_t[13], g_la_t[15], g_la_t[16]);
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  la_free(c);
  sio_free(c);
  free_plan(c);
  void *ptrs[] = {c->d_small, c->s_in.p, c->s_out.p, c->synth_tab.p, c->ls2_ws.p, c->ls2_ws_alt.p};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (c->ls2_host) (void)hipHostFree(c->ls2_host);
  for (int i = 0; i < 5; ++i)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream2) {
    (void)hipStreamSynchronize(c->stream2);
    for (int i = 0; i <= rfid_ctx::MAX_CHUNKS; ++i)
      if (c->ev_mf[i]) (void)hipEventDestroy(c->ev_mf[i]);
    for (int i = 0; i < 2 * rfid_ctx::MAX_CHUNKS; ++i)
      if (c->ev_gate[i]) (void)hipEventDestroy(c->ev_gate[i]);
    if (c->ev_pass) (void)hipEventDestroy(c->ev_pass);
    if (c->ev_front_end) (void)hipEventDestroy(c->ev_front_end);
    if (c->ev_fe_done) (void)hipEventDestroy(c->ev_fe_done);
    for (int i = 0; i < 2; ++i) {
      if (c->ev_tail_done[i]) (void)hipEventDestroy(c->ev_tail_done[i]);
      if (c->ev_y_free[i]) (void)hipEventDestroy(c->ev_y_free[i]);
    }
    (void)hipStreamDestroy(c->stream2);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return RFID_OK;
}

int rfid_ctx_reset(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  init_reader_state(c);
  memset(c->mf_hist, 0, sizeof(c->mf_hist));
  c->mf_seen = 0;
  if (c->la.on) { la_free(c); sio_free(c); }
  HIPCHK(c, hipMemsetAsync(c->d_gate1, 0, sizeof(GateState), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RFID_OK;
}

void *rfid_ctx_stream(rfid_ctx *c) { return c ? (void *)c->stream : nullptr; }

int rfid_get_state(const rfid_ctx *c, rfid_reader_state *out) {
  if (!c || !out) return RFID_ERR_INVALID;
  *out = c->rs;
  return RFID_OK;
}

int rfid_print_results(const rfid_ctx *c, char *buf, int cap, int *len) {  // reader_impl.cc:173-192
  if (!c || !buf || cap <= 0) return RFID_ERR_INVALID;
  const rfid_reader_state &rs = c->rs;
  int n = 0;
  n += snprintf(buf + n, cap - n, "\n --------------------------\n");
  n += snprintf(buf + n, cap - n, "| Number of queries/queryreps sent : %d\n", rs.n_queries_sent - 1);
  n += snprintf(buf + n, cap - n, "| Current Inventory round : %d\n", rs.cur_inventory_round);
  n += snprintf(buf + n, cap - n, " --------------------------\n");
  n += snprintf(buf + n, cap - n, "| Correctly decoded EPC : %d\n", rs.n_epc_correct);
  n += snprintf(buf + n, cap - n, "| Number of unique tags : %d\n", rs.n_unique_tags);
  for (int id = 0; id < 256 && n < cap - 64; id++)
    if (rs.tag_reads[id])
      n += snprintf(buf + n, cap - n, "| Tag ID : %x  Num of reads : %d\n", id, rs.tag_reads[id]);
  if (n < cap - 32) n += snprintf(buf + n, cap - n, " --------------------------\n");
  if (len) *len = n;
  return RFID_OK;
}

// ======================================================================================
// self test
// ======================================================================================
int rfid_selftest(rfid_ctx *c, int *n_failed) {
  if (!c) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  float hx[64], hn[64], hd[64];
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)((seed >> 8) & 0xFFFF) / 65536.0f; };
  float *d = nullptr;
  HIPCHK(c, hipMalloc((void **)&d, 9 * 64 * sizeof(float)));
  int bad = 0;
  // round 0: the magnitudes of the receive path; later rounds sweep the numerators over the binary
  // exponent range (2^-150 .. 2^124, zeros and denormals included) so that both paths of the
  // constant divisions (div_const) are compared with the host's correctly rounded quotient
  for (int round = 0; round < 48; ++round) {
    const float scale = (round == 0) ? 1.0f : ldexpf(1.0f, -150 + 6 * (round - 1));
    for (int i = 0; i < 64; ++i) {
      hx[i] = (rnd() - 0.5f) * 1e-3f * (float)(1 + (i % 7)) * ((round & 1) ? scale : 1.0f);   // (|x|: odd rounds sweep both operands)
      if (round > 0 && i == 9) hx[i] = 0.0f;
      hn[i] = (rnd() - 0.5f) * 37.0f * scale;
      if (round > 0 && i == 5) hn[i] = 0.0f;
      if (round > 0 && i == 6) hn[i] = -0.0f;
      hd[i] = (i % 3 == 0) ? 100.0f : ((i % 3 == 1) ? 48.0f : (rnd() + 0.01f) * 9.0f);
    }
    // carries for the in-order sums: plain, negative, next to a binade edge on either side (partial sums cross
    // it), zero, tiny, large; some rounds add exact half-ulp multiples (ties) -- the integer-scan form of the
    // sum must hand those steps to the chain and agree with the host's sequential sum everywhere
    const float carries[8] = {23.456789f, -19.12345f, 31.99999f, 16.000002f, 0.0f, 1e-30f, -15.99999f, 3.0e5f};
    const float carry = carries[round % 8];
    if (round % 8 == 5 || round % 16 == 8)
      for (int i = 0; i < 64; i += 3) hx[i] = ldexpf((float)((i % 7) - 3), -20) + ldexpf(1.0f, -20);
    HIPCHK(c, hipMemcpyAsync(d, hx, sizeof(hx), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 64, hn, sizeof(hn), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d + 128, hd, sizeof(hd), hipMemcpyHostToDevice, c->stream));
    SelfTestArgs a;
    a.x = d; a.num = d + 64; a.den = d + 128; a.carry = carry;
    a.chain_out = d + 192; a.div_out = d + 256; a.hyp_out = d + 320; a.shr_out = d + 384; a.scan_out = d + 448;
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    float out[5 * 64 + 1];
    HIPCHK(c, hipMemcpyAsync(out, d + 192, sizeof(out), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    volatile float acc = carry;
    for (int i = 0; i < 64; ++i) {
      acc = acc + hx[i];
      float e_chain = acc;
      volatile float q = hn[i] / hd[i];
      float e_div = q;
      float e_hyp = (float)sqrt((double)hn[i] * (double)hn[i] + (double)hx[i] * (double)hx[i]);
      float e_shr = (i == 0) ? 0.0f : hx[i - 1];
      if (memcmp(&e_chain, &out[i], 4)) bad++;
      if (memcmp(&e_div, &out[64 + i], 4)) bad++;
      if (memcmp(&e_hyp, &out[128 + i], 4)) bad++;
      if (memcmp(&e_shr, &out[192 + i], 4)) bad++;
      if (memcmp(&e_chain, &out[256 + i], 4)) bad++;   // the integer-scan form of the in-order sum (or its fallback)
    }
  }
  (void)hipFree(d);
  {
    // the wave scan of the long-stream front end's chain (DPP row shifts / broadcasts with a non-commutative operation)
    // against the same composition done one by one on the host
    int hin[128], hout[256];
    unsigned sd = 777u;
    for (int i = 0; i < 128; ++i) { sd = sd * 1664525u + 1013904223u; hin[i] = (int)(sd >> 9) - (1 << 22); }
    int *di = nullptr;
    HIPCHK(c, hipMalloc((void **)&di, sizeof(hin) + sizeof(hout)));
    HIPCHK(c, hipMemcpyAsync(di, hin, sizeof(hin), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(ls2_scan_selftest_kernel, dim3(1), dim3(64), 0, c->stream, (const int *)di, di + 128);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hout, di + 128, sizeof(hout), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(di);
    unsigned a0 = 0, a1 = 0;   // running composition, f then g: c[q] = f.c[q] + g.c[(q + f.c[q]) & 1]
    for (int l = 0; l < 64; ++l) {
      if (hout[4 * l + 2] != (int)a0 || hout[4 * l + 3] != (int)a1) bad++;
      const unsigned g0 = (unsigned)hin[2 * l], g1 = (unsigned)hin[2 * l + 1];
      const unsigned n0 = a0 + ((a0 & 1u) ? g1 : g0), n1 = a1 + (((1u + a1) & 1u) ? g1 : g0);
      a0 = n0; a1 = n1;
      if (hout[4 * l] != (int)a0 || hout[4 * l + 1] != (int)a1) bad++;
    }
  }
  {
    // the in-order sums of one step from two carries at once (rfid_ls2.hpp, chain_add_scan2): carries one ulp apart (a
    // piece's two variants), two apart and equal (what a tie leaves), addends with exact half-ulp multiples (ties) on
    // some lanes, next to a power of two (the sums leave the binade: the chains take over) -- against the host's
    // sequential sums, bit for bit
    float *dx = nullptr;
    HIPCHK(c, hipMalloc((void **)&dx, sizeof(float) * (64 + 129)));
    unsigned sd = 4242u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)(sd >> 8) * (1.0f / 16777216.0f); };
    int scanned_rounds = 0;
    for (int round = 0; round < 48; ++round) {
      const float bases[6] = {23.456789f, 0.7071f, -19.12345f, 16.000002f, 31.99999f, 3.0e5f};
      const float ca = bases[round % 6];
      uint32_t bits; memcpy(&bits, &ca, 4);
      const int du = (round / 6) % 4;                         // cb = ca + {1, 2, 0, 1} ulps (in the monotone integer image)
      uint32_t bb = bits + (uint32_t)((ca < 0.0f) ? -(int)((du == 3) ? 1 : ((du == 2) ? 0 : du + 1)) : (int)((du == 3) ? 1 : ((du == 2) ? 0 : du + 1)));
      float cb; memcpy(&cb, &bb, 4);
      const float u = ldexpf(1.0f, ilogbf(fabsf(ca)) - 23);   // ulp of the carries' binade
      float hx[64];
      for (int i = 0; i < 64; ++i) {
        hx[i] = (rnd() - 0.5f) * 200.0f * u * (float)(1 + i % 5);
        if ((round & 1) && i % 7 == 3) hx[i] = ((float)((int)(rnd() * 40.0f) - 20) + 0.5f) * u;      // ties
        if (round >= 24 && i % 11 == 5) hx[i] = ((float)((int)(rnd() * 9.0f) - 4) + 0.5f) * u;
      }
      HIPCHK(c, hipMemcpyAsync(dx, hx, sizeof(hx), hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(ls2_scan2_selftest_kernel, dim3(1), dim3(64), 0, c->stream, (const float *)dx, ca, cb, dx + 64);
      HIPCHK(c, hipGetLastError());
      float out[129];
      HIPCHK(c, hipMemcpyAsync(out, dx + 64, sizeof(out), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      volatile float sa = ca, sb = cb;
      for (int i = 0; i < 64; ++i) {
        sa = sa + hx[i]; sb = sb + hx[i];
        float ea = sa, eb = sb;
        if (memcmp(&ea, &out[i], 4)) bad++;
        if (memcmp(&eb, &out[64 + i], 4)) bad++;
      }
      if (out[128] != 0.0f) scanned_rounds++;
    }
    (void)hipFree(dx);
    if (scanned_rounds < 16) bad++;                           // (the shared scan must be what most of these rounds took)
  }
  if (n_failed) *n_failed = bad;
  if (bad) snprintf(c->err, sizeof(c->err), "selftest: %d primitive checks failed", bad);
  return RFID_OK;
}

// ======================================================================================
// (2) batched offline path
// ======================================================================================
int rfid_batch_plan(rfid_ctx *c, int n_streams, int64_t max_raw) {
  if (!c || n_streams <= 0 || max_raw <= 0) return RFID_ERR_INVALID;
  if (max_raw / DECIM > 0x7fffff00LL) return RFID_ERR_UNSUPPORTED;  // 32-bit sample indices per trace
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->sio.open) sio_free(c);   // a new plan replaces the work space an open stream runs on: the stream is closed
  free_plan(c);
  const int64_t n_dec = max_raw / DECIM;
  c->y_stride = (n_dec + 1) & ~1LL;  // even -> 16-byte aligned rows
  if (c->y_stride < 2) c->y_stride = 2;
  // closest two openings can be: RN16 window (250) + T1 (97 closed samples)
  int64_t wmax = n_dec / (RN16_WIN + T1_SAMPLES + 1) + 2;
  if (wmax * n_streams > 0x7fffffffLL) return RFID_ERR_UNSUPPORTED;
  c->wmax = (int)wmax;
  c->flat_cap = (int)(wmax * n_streams);
  c->max_raw = max_raw;
  // B (= "a plan exists") is set only after every allocation succeeded: a failed plan leaves the
  // context unplanned (free_plan), so later rfid_batch_* calls return RFID_ERR_STATE instead of
  // launching on null workspace pointers
  // one allocation per result set, carved up (256-byte aligned pieces): a plan -- every rfid_stream_begin and
  // rfid_lookahead_enable makes one -- costs one hipMalloc, not nine
  hipError_t e = hipSuccess;
  auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t sz_y = up256(sizeof(float2) * (size_t)c->y_stride * n_streams), sz_g = up256(sizeof(GateState) * (size_t)n_streams),
               sz_w = up256(sizeof(rfid_window) * (size_t)c->flat_cap), sz_f = up256(sizeof(rfid_window) * 2 * (size_t)c->flat_cap),
               sz_c = up256(sizeof(int) * (size_t)n_streams), sz_fc = 256, sz_r = up256(sizeof(rfid_decode_result) * (size_t)c->flat_cap),
               sz_sc = up256(sizeof(rfid_scores) * (size_t)c->flat_cap), sz_st = up256(sizeof(rfid_stream_stats) * (size_t)n_streams);
  const size_t sz_set = sz_y + sz_w + sz_f + sz_c + sz_fc + sz_r + sz_st;     // what a result set holds
  auto bind_set = [&](char *b, float2 *&y, rfid_window *&w, rfid_window *&f, int *&wc, int *&fc, rfid_decode_result *&r, rfid_stream_stats *&st) {
    y = (float2 *)b; b += sz_y; w = (rfid_window *)b; b += sz_w; f = (rfid_window *)b; b += sz_f; wc = (int *)b; b += sz_c;
    fc = (int *)b; b += sz_fc; r = (rfid_decode_result *)b; b += sz_r; st = (rfid_stream_stats *)b; b += sz_st;
    return b;
  };
  e = hipMalloc(&c->plan_blk, sz_set + sz_g + sz_sc);
  if (e == hipSuccess) {
    char *b = bind_set((char *)c->plan_blk, c->d_y, c->d_wtab, c->d_flat, c->d_wcount, c->d_flat_count, c->d_res, c->d_stats);
    c->d_gstate = (GateState *)b; b += sz_g;
    c->d_scores = (rfid_scores *)b;
    e = hipMemset(c->d_wcount, 0, sz_c + sz_fc);   // (the window counts and, right behind them, the two list counters)
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();   // clear the sticky allocation error
    free_plan(c);
    return fail(c, RFID_ERR_HIP, "rfid_batch_plan: workspace allocation", e);
  }
  c->B = c->B_plan = n_streams;
  if (c->wmax > 2048) {
    // few, long traces: the statistics kernel is one workgroup per trace, and 48-byte results through one CU are its whole
    // time (0.36 ms for configs[2]'s 320 000 windows); the decoder leaves the word it needs of each (doing without is fine)
    if (hipMalloc((void **)&c->d_sum, sizeof(int) * ((size_t)c->flat_cap + 4)) != hipSuccess) { (void)hipGetLastError(); c->d_sum = nullptr; }
  }
  {
    // the second result set (see rfid_ctx::ResultSet): opt-in (RFID_OVERLAP=2) and only where it is small beside what is
    // free.  Measured on configs[1] (profiles/r04/overlap.txt): the decoder's waves beside the next front end take their
    // instruction slots from it -- -1.5 % per pass on one box, +3 % on another (a decoder wave that gets to a CU first keeps
    // the front end's workgroup out until it is through) -- not a default.
    const size_t need = sizeof(float2) * (size_t)c->y_stride * n_streams + (sizeof(rfid_window) * 3 + sizeof(rfid_decode_result)) * (size_t)c->flat_cap +
                        sizeof(rfid_stream_stats) * (size_t)n_streams;
    size_t free_b = 0, total_b = 0;
    if (c->knobs.overlap >= 2 && n_streams >= 64 && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 8) {
      hipError_t e2 = hipMalloc(&c->alt_blk, sz_set);
      if (e2 == hipSuccess) {
        bind_set((char *)c->alt_blk, c->alt.d_y, c->alt.d_wtab, c->alt.d_flat, c->alt.d_wcount, c->alt.d_flat_count, c->alt.d_res, c->alt.d_stats);
        e2 = hipMemset(c->alt.d_wcount, 0, sz_c + sz_fc);
      }
      if (e2 == hipSuccess) c->alt_have = true;
      else {
        (void)hipGetLastError();
        if (c->alt_blk) (void)hipFree(c->alt_blk);
        c->alt_blk = nullptr;
        c->alt = rfid_ctx::ResultSet();
      }
    }
  }
  // the long-stream front end's work space, when this shape can take that path: reserved here so that a planned batch
  // does not meet an allocation in its passes (a pass that finds none falls back to the sequential scan)
  if (ls_may_apply(c, n_streams, n_dec)) {
    {
      // ... and a second matched-filter output buffer for the filter of the next pass (unless a whole second set exists, or
      // this is a stream's own plan: n_dec of a stream call is small and its passes do not overlap)
      size_t free_b = 0, total_b = 0;
      if (!c->alt_have && !c->plan_for_stream && c->knobs.overlap != 0 && sz_y >= ((size_t)16 << 20) && hipMemGetInfo(&free_b, &total_b) == hipSuccess && sz_y < free_b / 8) {
        if (hipMalloc(&c->alt_y_blk, sz_y) == hipSuccess) c->alt.d_y = (float2 *)c->alt_y_blk;
        else { (void)hipGetLastError(); c->alt_y_blk = nullptr; }
      }
    }
    const size_t need = ls_workspace_bytes(n_streams, n_dec, c->y_stride, c->wmax);
    if (need > c->ls2_ws.cap) {
      if (c->ls2_ws.p) (void)hipFree(c->ls2_ws.p);
      c->ls2_ws.p = nullptr; c->ls2_ws.cap = 0;
      if (hipMalloc(&c->ls2_ws.p, need) == hipSuccess) c->ls2_ws.cap = need;
      else { (void)hipGetLastError(); c->ls2_ws.p = nullptr; }
    }
    // ... and a second work space beside the second filter output buffer (the fused first pass of the next pass beside the
    // rest of this one, rfid_batch_process), where it is small beside what is free
    if (c->alt.d_y && !c->alt_have && (c->knobs.overlap != 0) && c->ls2_ws.p && need > c->ls2_ws_alt.cap) {
      size_t free_b = 0, total_b = 0;
      if (c->ls2_ws_alt.p) (void)hipFree(c->ls2_ws_alt.p);
      c->ls2_ws_alt.p = nullptr; c->ls2_ws_alt.cap = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need < free_b / 8) {
        if (hipMalloc(&c->ls2_ws_alt.p, need) == hipSuccess) c->ls2_ws_alt.cap = need;
        else { (void)hipGetLastError(); c->ls2_ws_alt.p = nullptr; }
      }
    }
  }
  // persistent decoders: the EPC kernel holds 18.6 KiB of LDS per single-wave workgroup -> 8 per CU
  c->decode_grid = c->n_cus * 8;
  for (int i = 0; i < 5; ++i) c->ev_valid[i] = false;
  return RFID_OK;
}

int rfid_batch_set_streams(rfid_ctx *c, int n_streams) {
  if (!c || n_streams <= 0) return RFID_ERR_INVALID;
  if (!c->B_plan) return RFID_ERR_STATE;
  if (n_streams > c->B_plan) return RFID_ERR_CAPACITY;
  c->B = n_streams;
  return RFID_OK;
}

static int batch_mf_on(rfid_ctx *c, hipStream_t stream, const void *d_raw, int64_t raw_stride, int64_t n_raw, const void *d_lens);
int rfid_batch_mf(rfid_ctx *c, const void *d_raw, int64_t raw_stride, int64_t n_raw, const void *d_lens) {
  if (!c || !d_raw || n_raw < 0 || raw_stride < n_raw) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  if (n_raw > c->max_raw) return RFID_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  c->y_touched = true;
  return batch_mf_on(c, c->stream, d_raw, raw_stride, n_raw, d_lens);
}
static int batch_mf_on(rfid_ctx *c, hipStream_t stream, const void *d_raw, int64_t raw_stride, int64_t n_raw, const void *d_lens) {
  c->d_lens = (const int64_t *)d_lens;
  c->last_n_raw = n_raw;
  MfArgs a;
  a.x = (const float2 *)d_raw; a.x_stride = raw_stride; a.n_raw = n_raw; a.lens = c->d_lens;
  a.n_out = n_raw / DECIM; a.in_off = -(NTAPS - 1);
  a.vec_ok = ((raw_stride & 1) == 0 && (((uintptr_t)d_raw) & 15) == 0) ? 1 : 0;
  a.y = c->d_y; a.y_stride = c->y_stride; a.tile0 = 0; a.stream0 = 0;
  c->n_chunks_last = 0;
  c->fused_last = 0;
  HIPCHK(c, hipEventRecord(c->ev[0], stream));
  const int64_t tiles = (a.n_out + MF_TILE - 1) / MF_TILE;
  for (int s0 = 0; s0 < c->B && tiles > 0; s0 += 65535) {   // gridDim.y limit
    a.stream0 = s0;
    const int ns = (c->B - s0 < 65535) ? (c->B - s0) : 65535;
    hipLaunchKernelGGL(mf_boxcar25_decim5_kernel, dim3((unsigned)tiles, (unsigned)ns), dim3(MF_THREADS), 0, stream, a);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventRecord(c->ev[1], stream));
  c->ev_valid[0] = c->ev_valid[1] = true;
  return RFID_OK;
}

static int rfid_batch_gate_impl(rfid_ctx *c, const int *skip_if) {
  if (!c) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  // fresh gate per trace (gate_impl ctor, gate_impl.cc:41-70): all-zero state; the kernel
  // arms n_samples_to_ungate for the first RN16 itself
  HIPCHK(c, hipMemsetAsync(c->d_gstate, 0, sizeof(GateState) * (size_t)c->B, c->stream));
  if (!skip_if) HIPCHK(c, hipMemsetAsync(c->d_flat_count, 0, 2 * sizeof(int), c->stream));   // (else: zeroed before the front end)
  GateArgs a = {};
  a.skip_if = skip_if;
  a.y = c->d_y; a.y_stride = c->y_stride; a.n_dec = c->last_n_raw / DECIM; a.lens = c->d_lens;
  a.pos0 = 0; a.chunk_len = a.n_dec;
  a.state = c->d_gstate; a.n_streams = c->B; a.wtab = c->d_wtab; a.wmax = c->wmax; a.wcount = c->d_wcount;
  a.flat = c->d_flat; a.flat_count = c->d_flat_count; a.flat_cap = c->flat_cap; a.mode = 0;
  a.gated = nullptr; a.gated_cap = 0; a.io = nullptr;
  if (!c->ev_valid[1]) { HIPCHK(c, hipEventRecord(c->ev[1], c->stream)); c->ev_valid[1] = true; }
  hipLaunchKernelGGL(gate_scan_kernel, dim3((unsigned)((c->B + GATE_STREAMS_PER_WG - 1) / GATE_STREAMS_PER_WG)), dim3(GATE_THREADS), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  c->ev_valid[2] = true;
  return RFID_OK;
}
int rfid_batch_gate(rfid_ctx *c) {
  if (c) { c->d_ls2_ctl = nullptr; c->y_touched = true; }
  return rfid_batch_gate_impl(c, nullptr);
}

int rfid_batch_decode(rfid_ctx *c, int want_scores) {
  if (!c) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  DecodeListArgs a;
  a.y = c->y(); a.y_stride = c->y_stride; a.cap = c->flat_cap; a.res = c->d_res;
  a.scores = want_scores ? c->d_scores : nullptr; a.wmax = c->wmax;
  a.sum = c->alt_have ? nullptr : c->d_sum;   // (one array: not with two result sets in flight)
  c->sum_of = a.sum ? c->d_res : nullptr;
  memcpy(a.t_cand, c->t_cand, sizeof(a.t_cand));
  hipStream_t ts = c->tail_stream ? c->tail_stream : c->stream;
  if (!c->tail_stream) { int rj = join_tails(c); if (rj) return rj; }
  if (!c->ev_valid[2]) { HIPCHK(c, hipEventRecord(c->ev[2], ts)); c->ev_valid[2] = true; }
  int grid = c->decode_grid;
  {   // (a small plan -- a stream call, a look-ahead pass -- holds few windows: no point in launching a chip's worth of waves)
    const int64_t most = ((int64_t)c->wmax * c->B + 2) / 3 + 1;
    if (grid > most) grid = (int)most;
  }
  if (grid < 1) grid = 1;
  // one launch: EPC windows 3 per wavefront, then RN16 windows 4 per wavefront drawn from a counter
  DecodeAllArgs d;
  d.epc = a; d.rn16 = a;
  d.epc.list = c->d_flat + c->flat_cap; d.epc.count = c->d_flat_count + 1;
  d.rn16.list = c->d_flat; d.rn16.count = c->d_flat_count;
  d.ticket = c->d_ticket + (c->ticket_flip & 1);        // (both zero after rfid_ctx_create; every launch zeroes the other one)
  d.ticket_next = c->d_ticket + ((c->ticket_flip & 1) ^ 1);
  c->ticket_flip ^= 1;
  hipLaunchKernelGGL(decode_all_kernel, dim3((unsigned)grid), dim3(64), 0, ts, d);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev[3], ts));
  c->ev_valid[3] = true;
  return RFID_OK;
}

int rfid_batch_stats(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  StatsArgs a;
  a.res = c->d_res; a.wcount = c->d_wcount; a.wmax = c->wmax; a.n_streams = c->B;
  a.sum = (c->d_sum && c->sum_of == c->d_res) ? c->d_sum : nullptr;   // (the summaries rfid_batch_decode left of THESE results)
  a.max_slot_number = (int)pow(2, c->prm.fixed_q);
  a.max_num_queries = c->prm.max_num_queries; a.number_unique_tags = c->prm.number_unique_tags;
  a.out = c->d_stats;
  hipStream_t ts = c->tail_stream ? c->tail_stream : c->stream;
  if (!c->tail_stream) { int rj = join_tails(c); if (rj) return rj; }
  if (!c->ev_valid[3]) { HIPCHK(c, hipEventRecord(c->ev[3], ts)); c->ev_valid[3] = true; }
  // one wave per trace; sixteen when a trace can hold thousands of windows (few long traces)
  hipLaunchKernelGGL(stream_stats_kernel, dim3((unsigned)c->B), dim3(c->wmax > 2048 ? 64 * STATS_MAX_WAVES : 64), 0, ts, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(c->ev[4], ts));
  c->ev_valid[4] = true;
  return RFID_OK;
}

// mf -> gate -> decode -> stats.  The matched filter (HBM-bound) and the gate scan (bound by the
// latency of its in-order sums, one wave per SIMD) are overlapped: the traces are cut along
// time into chunks; chunk k of the gate scan runs on a second stream as soon as chunk k of the
// matched filter is done, carrying the gate state from chunk to chunk (exactly the state a
// streaming call sequence would carry).
int rfid_batch_process(rfid_ctx *c, const void *d_raw, int64_t raw_stride, int64_t n_raw, const void *d_lens,
                       int want_scores) {
  if (!c || !d_raw || n_raw < 0 || raw_stride < n_raw) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  if (n_raw > c->max_raw) return RFID_ERR_CAPACITY;
  ls_note_last_pass(c);
  const int64_t n_out = n_raw / DECIM;
  const int64_t tiles = (n_out + MF_TILE - 1) / MF_TILE;
  // measured on MI355X (1024 traces): the overlap paid off while the gate scan took 4 ms (-5 %), but the
  // scan is slowed down by anything that shares its SIMDs; since it runs in 2.9 ms the plain sequence is
  // faster, so chunking is opt-in (RFID_FRONT_CHUNKS=8)
  const int nch = c->knobs.front_chunks;
  if (nch < 2 && ls_applicable(c, c->B, n_out)) {
    // few long traces: matched filter, then the gate scan as the long-stream front end -- every launch of it enqueued
    // here, the sequential scan behind them as the fallback that skips itself when the front end succeeded
    HIPCHK(c, hipSetDevice(c->device));
    { int rj = join_tails(c); if (rj) return rj; }
    int rc;
    // The fused first pass (round 5, the default for traces that start from the fresh gate): no matched-filter launch -- the
    // front end's first launch filters the raw samples itself (ls2_front_kernel), one sweep over the raw samples instead of the
    // filter's and three over its output.  When the front end gives up the sequential scan behind it needs all of y: a filter
    // launch and the scan are enqueued behind the list, both skipping themselves on Ls2Ctl::ok.
    if (raw_stride >= 2) {
      c->d_lens = (const int64_t *)d_lens;
      c->last_n_raw = n_raw;
      c->n_chunks_last = 0;
      // Passes enqueued back to back: with a second matched-filter output buffer and a second work space (rfid_batch_plan) the
      // fused first pass of THIS pass -- bound by the HBM, and touching nothing but the raw samples, y and its work space --
      // runs on stream2 beside the rest of the pass before (re-run rounds, state machine, dc_est, decoder: instruction- and
      // latency-bound launches that leave most of the HBM's bandwidth unused); the rest of this pass follows on the main
      // stream.  The buffers alternate; a buffer is written again only when the pass before last is through with it.
      // (an error anywhere below may leave the alternating buffers / work spaces swapped without a pass behind them: HIPCHK_T marks
      // the context so that the next pass waits for everything enqueued so far before its second stream starts, whichever
      // buffers it gets)
      const bool ahead = c->alt.d_y != nullptr && c->ls2_ws_alt.p != nullptr && !c->alt_have && (c->knobs.overlap != 0);
      if (ahead) {
        std::swap(c->d_y, c->alt.d_y);
        c->y_idx ^= 1;
        if (c->y_touched) {   // (something outside this protocol used a buffer on the main stream: wait for all of it)
          HIPCHK_T(c, hipEventRecord(c->ev_pass, c->stream));
          HIPCHK_T(c, hipStreamWaitEvent(c->stream2, c->ev_pass, 0));
          c->y_touched = false;
        }
        if (c->y_recorded[c->y_idx]) {
          HIPCHK_T(c, hipStreamWaitEvent(c->stream2, c->ev_y_free[c->y_idx], 0));
          c->y_recorded[c->y_idx] = false;
        }
      }
      HIPCHK_T(c, hipEventRecord(c->ev[0], c->stream));
      HIPCHK_T(c, hipEventRecord(c->ev[1], c->stream));   // mf_ms = 0: the filter runs inside the front end's first launch
      c->ev_valid[0] = c->ev_valid[1] = true;
      int enq = 0;
      LsOpts lo;
      lo.raw = d_raw; lo.raw_stride = raw_stride; lo.ahead = ahead;
      if ((rc = ls_enqueue(c, n_out, lo, &enq))) { c->y_touched = true; return rc; }
      if (!enq && ahead) {
        // (not applicable with the second stream after all: the buffers go back, the pass runs on the main stream alone)
        std::swap(c->d_y, c->alt.d_y); c->y_idx ^= 1;
        c->y_touched = true;
        lo.ahead = false;
        if ((rc = ls_enqueue(c, n_out, lo, &enq))) { c->y_touched = true; return rc; }
      }
      const bool ahead_now = ahead && lo.ahead;
      if (enq) {
        c->fused_last = 2;
        MfFallbackArgs f;
        f.m.x = (const float2 *)d_raw; f.m.x_stride = raw_stride; f.m.n_raw = n_raw; f.m.lens = c->d_lens;
        f.m.n_out = n_out; f.m.in_off = -(NTAPS - 1);
        f.m.vec_ok = ((raw_stride & 1) == 0 && (((uintptr_t)d_raw) & 15) == 0) ? 1 : 0;
        f.m.y = c->d_y; f.m.y_stride = c->y_stride; f.m.tile0 = 0; f.m.stream0 = 0;
        f.skip_if = &c->d_ls2_ctl->ok; f.n_tiles = tiles;
        const int64_t gx = (tiles < 4096) ? tiles : 4096;
        for (int s0 = 0; s0 < c->B && gx > 0; s0 += 65535) {
          f.m.stream0 = s0;
          const int ns = (c->B - s0 < 65535) ? (c->B - s0) : 65535;
          hipLaunchKernelGGL(mf_fallback_kernel, dim3((unsigned)gx, (unsigned)ns), dim3(MF_THREADS), 0, c->stream, f);
          HIPCHK_T(c, hipGetLastError());
        }
        if ((rc = rfid_batch_gate_impl(c, &c->d_ls2_ctl->ok)) || (rc = rfid_batch_decode(c, want_scores)) || (rc = rfid_batch_stats(c))) { c->y_touched = true; return rc; }
        if (ahead_now) {
          HIPCHK_T(c, hipEventRecord(c->ev_y_free[c->y_idx], c->stream));
          c->y_recorded[c->y_idx] = true;
        } else {
          c->y_touched = true;
        }
        return RFID_OK;
      }
      // (no work space: the plain sequence below)
    }
    // (rows of a single raw sample, or no work space for the fused first pass: the matched filter by itself, then the front end over
    // its output -- the list the streaming calls use)
    c->y_touched = true;
    if ((rc = batch_mf_on(c, c->stream, d_raw, raw_stride, n_raw, d_lens))) return rc;
    int enq = 0;
    LsOpts lo;
    if ((rc = ls_enqueue(c, n_out, lo, &enq))) return rc;
    if ((rc = rfid_batch_gate_impl(c, enq ? &c->d_ls2_ctl->ok : nullptr))) return rc;
    if ((rc = rfid_batch_decode(c, want_scores))) return rc;
    return rfid_batch_stats(c);
  }
  c->d_ls2_ctl = nullptr;   // (this pass does not run the long-stream front end)
  c->y_touched = true;
  if (nch < 2 && raw_stride >= 2 && !c->knobs.front_unfused) {
    // default: fused front end -- the gate's producer waves run the matched filter themselves
    // (one read of the raw samples, one write of y for the decoder, no second pass over y)
    HIPCHK(c, hipSetDevice(c->device));
    c->d_lens = (const int64_t *)d_lens;
    c->last_n_raw = n_raw;
    c->n_chunks_last = 0;
    const bool overlap = c->alt_have && !want_scores;
    if (overlap) {
      // this pass works on the set the pass before last used; its decoder and statistics (stream2) must be through with it
      std::swap(c->d_y, c->alt.d_y); std::swap(c->d_wtab, c->alt.d_wtab); std::swap(c->d_flat, c->alt.d_flat);
      std::swap(c->d_wcount, c->alt.d_wcount); std::swap(c->d_flat_count, c->alt.d_flat_count);
      std::swap(c->d_res, c->alt.d_res); std::swap(c->d_stats, c->alt.d_stats);
      c->set_idx ^= 1;
      if (c->tail_recorded[c->set_idx]) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_tail_done[c->set_idx], 0));
        c->tail_recorded[c->set_idx] = false;
      }
    } else {
      int rj = join_tails(c);
      if (rj) return rj;
    }
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));   // mf_ms = 0: the filter runs inside the gate launch
    HIPCHK(c, hipMemsetAsync(c->d_gstate, 0, sizeof(GateState) * (size_t)c->B, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_flat_count, 0, 2 * sizeof(int), c->stream));
    GateArgs g = {};
    g.y = c->d_y; g.y_w = c->d_y; g.y_stride = c->y_stride; g.n_dec = n_out; g.lens = c->d_lens;
    g.pos0 = 0; g.chunk_len = n_out;
    g.state = c->d_gstate; g.n_streams = c->B; g.wtab = c->d_wtab; g.wmax = c->wmax; g.wcount = c->d_wcount;
    g.flat = c->d_flat; g.flat_count = c->d_flat_count; g.flat_cap = c->flat_cap; g.mode = 0;
    g.raw = (const float2 *)d_raw; g.raw_stride = raw_stride; g.n_raw = n_raw;
    g.raw_vec_ok = ((raw_stride & 1) == 0 && (((uintptr_t)d_raw) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(front_end_fused_kernel, dim3((unsigned)((c->B + GATE_STREAMS_PER_WG - 1) / GATE_STREAMS_PER_WG)),
                       dim3(GATE_THREADS), 0, c->stream, g);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    c->ev_valid[0] = c->ev_valid[1] = c->ev_valid[2] = true;
    c->fused_last = 1;
    if (overlap) {
      // decoder + statistics of this pass on stream2, behind this front end; the next pass's front end does not wait for them
      HIPCHK(c, hipEventRecord(c->ev_fe_done, c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fe_done, 0));
      c->tail_stream = c->stream2;
      c->ev_valid[2] = false;           // (ev[2] is re-recorded on stream2: the decoder's time starts when it can start)
      int rc = rfid_batch_decode(c, 0);
      if (!rc) rc = rfid_batch_stats(c);
      c->tail_stream = nullptr;
      if (rc) return rc;
      HIPCHK(c, hipEventRecord(c->ev_tail_done[c->set_idx], c->stream2));
      c->tail_recorded[c->set_idx] = true;
      return RFID_OK;
    }
    int rc = rfid_batch_decode(c, want_scores);
    if (rc) return rc;
    return rfid_batch_stats(c);
  }
  c->fused_last = 0;
  if (tiles < 4 * (int64_t)nch || nch < 2) {   // plain sequence of the stage kernels
    int rc = rfid_batch_mf(c, d_raw, raw_stride, n_raw, d_lens);
    if (rc) return rc;
    if ((rc = rfid_batch_gate(c))) return rc;
    if ((rc = rfid_batch_decode(c, want_scores))) return rc;
    return rfid_batch_stats(c);
  }
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  if (!c->ev_mf[0]) {
    for (int i = 0; i <= rfid_ctx::MAX_CHUNKS; ++i) HIPCHK(c, hipEventCreate(&c->ev_mf[i]));
    for (int i = 0; i < 2 * rfid_ctx::MAX_CHUNKS; ++i) HIPCHK(c, hipEventCreate(&c->ev_gate[i]));
  }
  c->d_lens = (const int64_t *)d_lens;
  c->last_n_raw = n_raw;
  const int64_t tiles_per_chunk = (tiles + nch - 1) / nch;
  // the previous pass (decode/stats on `stream`) must be over before the gate state is reset
  HIPCHK(c, hipEventRecord(c->ev_pass, c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_pass, 0));
  HIPCHK(c, hipMemsetAsync(c->d_gstate, 0, sizeof(GateState) * (size_t)c->B, c->stream2));
  HIPCHK(c, hipMemsetAsync(c->d_flat_count, 0, 2 * sizeof(int), c->stream2));
  MfArgs m;
  m.x = (const float2 *)d_raw; m.x_stride = raw_stride; m.n_raw = n_raw; m.lens = c->d_lens;
  m.n_out = n_out; m.in_off = -(NTAPS - 1);
  m.vec_ok = ((raw_stride & 1) == 0 && (((uintptr_t)d_raw) & 15) == 0) ? 1 : 0;
  m.y = c->d_y; m.y_stride = c->y_stride; m.stream0 = 0;
  GateArgs g = {};
  g.y = c->d_y; g.y_stride = c->y_stride; g.n_dec = n_out; g.lens = c->d_lens;
  g.state = c->d_gstate; g.n_streams = c->B; g.wtab = c->d_wtab; g.wmax = c->wmax; g.wcount = c->d_wcount;
  g.flat = c->d_flat; g.flat_count = c->d_flat_count; g.flat_cap = c->flat_cap; g.mode = 0;
  g.gated = nullptr; g.gated_cap = 0; g.io = nullptr;
  HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
  HIPCHK(c, hipEventRecord(c->ev_mf[0], c->stream));
  int used = 0;
  for (int k = 0; k < nch; ++k) {
    const int64_t t0 = (int64_t)k * tiles_per_chunk;
    if (t0 >= tiles) break;
    const int64_t tn = (t0 + tiles_per_chunk <= tiles) ? tiles_per_chunk : (tiles - t0);
    m.tile0 = t0;
    for (int s0 = 0; s0 < c->B; s0 += 65535) {   // gridDim.y limit
      m.stream0 = s0;
      const int ns = (c->B - s0 < 65535) ? (c->B - s0) : 65535;
      hipLaunchKernelGGL(mf_boxcar25_decim5_kernel, dim3((unsigned)tn, (unsigned)ns), dim3(MF_THREADS), 0, c->stream, m);
      HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipEventRecord(c->ev_mf[k + 1], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_mf[k + 1], 0));
    g.pos0 = t0 * MF_TILE; g.chunk_len = tn * MF_TILE;
    HIPCHK(c, hipEventRecord(c->ev_gate[2 * k], c->stream2));
    hipLaunchKernelGGL(gate_scan_kernel, dim3((unsigned)((c->B + GATE_STREAMS_PER_WG - 1) / GATE_STREAMS_PER_WG)),
                       dim3(GATE_THREADS), 0, c->stream2, g);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_gate[2 * k + 1], c->stream2));
    used = k + 1;
  }
  c->n_chunks_last = used;
  HIPCHK(c, hipEventRecord(c->ev_front_end, c->stream2));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_front_end, 0));
  HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
  c->ev_valid[0] = true; c->ev_valid[1] = false; c->ev_valid[2] = true;
  int rc = rfid_batch_decode(c, want_scores);
  if (rc) return rc;
  return rfid_batch_stats(c);
}

int rfid_batch_set_long_stream(rfid_ctx *c, int mode) {
  if (!c || mode < 0 || mode > 2) return RFID_ERR_INVALID;
  c->ls_mode = mode;
  return RFID_OK;
}

int rfid_ctx_set_knob(rfid_ctx *c, const char *name, int value) {
  if (!c || !name) return RFID_ERR_INVALID;
  for (const KnobEntry &e : g_knob_table)
    if (strcmp(name, e.name) == 0) {
      if (value < e.lo || value > e.hi) return RFID_ERR_INVALID;
      c->knobs.*(e.field) = value;
      if (e.field == &RfidKnobs::long_stream) c->ls_mode = value;
      if (e.field == &RfidKnobs::la_profile && value) g_la_on = true;
      return RFID_OK;
    }
  return RFID_ERR_INVALID;
}
int rfid_ctx_get_knob(const rfid_ctx *c, const char *name, int *value) {
  if (!c || !name || !value) return RFID_ERR_INVALID;
  for (const KnobEntry &e : g_knob_table)
    if (strcmp(name, e.name) == 0) { *value = (e.field == &RfidKnobs::long_stream) ? c->ls_mode : c->knobs.*(e.field); return RFID_OK; }
  return RFID_ERR_INVALID;
}

int rfid_batch_ls_report(const rfid_ctx *c, rfid_ls_report *out) {
  if (!c || !out) return RFID_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  if (!c->d_ls2_ctl || !c->ls2_host) return RFID_OK;   // the last pass did not run the long-stream front end
  (void)hipSetDevice(c->device);
  if (hipStreamSynchronize(c->stream) != hipSuccess) return RFID_ERR_HIP;   // (the copy of the control block rides on the pass)
  const Ls2Ctl &k = *c->ls2_host;
  const int ra = c->ls2_rounds[0], rf = c->ls2_rounds[1], rd = c->ls2_rounds[2];
  out->pieces = k.n_pieces;
  out->units = k.n_units;
  out->chunk = c->ls2_P;
  out->avg_rounds = k.avg_rounds; out->avg_reruns = k.avg_reruns;
  out->fsm_rounds = k.fsm_rounds;
  out->dc_rounds = k.dc_rounds; out->dc_reruns = k.dc_reruns;
  for (int r = 0; r <= rf; ++r) out->cuts_dropped += k.fsm_count[r];
  out->windows = k.n_windows;
  out->dc_finished = k.dc_finished;
  out->verified = (k.ok != 0) ? 1 : 0;
  out->gave_up = out->verified ? 0 : (k.fail ? k.fail : (k.avg_count[ra] ? 2 : (k.fsm_count[rf] ? 3 : (k.dc_count[rd] ? 4 : 5))));   // (4: not reached since round 6 -- the finishing walk settles what the rounds leave)
  return RFID_OK;
}

int rfid_batch_sync(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RFID_OK;
}

int rfid_batch_timing_get(rfid_ctx *c, rfid_batch_timing *out) {
  if (!c || !out) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms[4] = {0, 0, 0, 0};
  float front = 0.0f;
  if (c->n_chunks_last > 0) {
    // overlapped front end: kernel times are the sums over the chunks on their own streams
    for (int k = 0; k < c->n_chunks_last; ++k) {
      float t = 0.0f;
      HIPCHK(c, hipEventElapsedTime(&t, c->ev_mf[k], c->ev_mf[k + 1]));
      ms[0] += t;
      HIPCHK(c, hipEventElapsedTime(&t, c->ev_gate[2 * k], c->ev_gate[2 * k + 1]));
      ms[1] += t;
    }
    HIPCHK(c, hipEventElapsedTime(&front, c->ev_mf[0], c->ev_front_end));
    for (int i = 2; i < 4; ++i)
      if (c->ev_valid[i] && c->ev_valid[i + 1]) HIPCHK(c, hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    out->total_ms = front + ms[2] + ms[3];
  } else {
    for (int i = 0; i < 4; ++i)
      if (c->ev_valid[i] && c->ev_valid[i + 1]) HIPCHK(c, hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));
    out->total_ms = ms[0] + ms[1] + ms[2] + ms[3];
  }
  out->mf_ms = ms[0]; out->gate_ms = ms[1]; out->decode_ms = ms[2]; out->stats_ms = ms[3];
  out->front_ms = (c->n_chunks_last > 0) ? front : (ms[0] + ms[1]);
  out->front_chunks = (c->n_chunks_last > 0) ? c->n_chunks_last : 1;
  out->decode_launches = 1;
  out->fused_front = c->fused_last;
  out->reserved_ = 0;
  return RFID_OK;
}

int rfid_batch_get_stats(rfid_ctx *c, rfid_stream_stats *out, int n_streams) {
  if (!c || !out || n_streams <= 0) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  if (n_streams > c->B) n_streams = c->B;
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  HIPCHK(c, hipMemcpyAsync(out, c->d_stats, sizeof(rfid_stream_stats) * (size_t)n_streams, hipMemcpyDeviceToHost,
                           c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RFID_OK;
}

int rfid_batch_get_windows(rfid_ctx *c, rfid_window *windows, rfid_decode_result *results, rfid_scores *scores,
                           int64_t cap, int64_t *n) {
  if (!c || !n) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  std::vector<int> wc((size_t)c->B);
  HIPCHK(c, hipMemcpyAsync(wc.data(), c->d_wcount, sizeof(int) * (size_t)c->B, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int64_t total = 0;
  for (int s = 0; s < c->B; ++s) {
    const int k = wc[(size_t)s];
    const int64_t room = cap - total;
    const int64_t take = (room <= 0) ? 0 : ((k < room) ? k : room);
    if (take > 0) {
      const size_t off = (size_t)s * (size_t)c->wmax;
      if (windows)
        HIPCHK(c, hipMemcpyAsync(windows + total, c->d_wtab + off, sizeof(rfid_window) * (size_t)take,
                                 hipMemcpyDeviceToHost, c->stream));
      if (results)
        HIPCHK(c, hipMemcpyAsync(results + total, c->d_res + off, sizeof(rfid_decode_result) * (size_t)take,
                                 hipMemcpyDeviceToHost, c->stream));
      if (scores)
        HIPCHK(c, hipMemcpyAsync(scores + total, c->d_scores + off, sizeof(rfid_scores) * (size_t)take,
                                 hipMemcpyDeviceToHost, c->stream));
    }
    total += k;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n = total;
  return RFID_OK;
}

int rfid_batch_device_ptrs(rfid_ctx *c, void **d_mf_out, int64_t *mf_stride, void **d_stats, void **d_flat_count) {
  if (!c) return RFID_ERR_INVALID;
  if (!c->B) return RFID_ERR_STATE;
  // (the buffers of the LAST pass; with the second result set in use they alternate from pass to pass)
  { int rj = join_tails(c); if (rj) return rj; }
  if (d_mf_out) *d_mf_out = c->d_y;
  if (mf_stride) *mf_stride = c->y_stride;
  if (d_stats) *d_stats = c->d_stats;
  if (d_flat_count) *d_flat_count = c->d_flat_count;
  return RFID_OK;
}

int rfid_batch_get_mf(rfid_ctx *c, int stream, rfid_cf32 *out, int64_t cap, int64_t *n) {
  if (!c || !out || !n) return RFID_ERR_INVALID;
  if (!c->B || stream < 0 || stream >= c->B) return RFID_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  int64_t n_raw = c->last_n_raw;
  if (c->d_lens) {
    int64_t l = 0;
    HIPCHK(c, hipMemcpyAsync(&l, c->d_lens + stream, sizeof(l), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (l < n_raw) n_raw = l;
    if (n_raw < 0) n_raw = 0;
  }
  int64_t k = n_raw / DECIM;
  *n = k;
  if (k > cap) k = cap;
  if (k > 0)
    HIPCHK(c, hipMemcpyAsync(out, c->d_y + (size_t)stream * (size_t)c->y_stride, sizeof(float2) * (size_t)k,
                             hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return RFID_OK;
}

int rfid_batch_get_gated(rfid_ctx *c, int stream, int seq, rfid_cf32 *out, int64_t cap, int64_t *n) {
  if (!c || !out || !n || seq < 0) return RFID_ERR_INVALID;
  if (!c->B || stream < 0 || stream >= c->B || seq >= c->wmax) return RFID_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  { int rj = join_tails(c); if (rj) return rj; }
  int wc = 0;
  rfid_window w;
  HIPCHK(c, hipMemcpyAsync(&wc, c->d_wcount + stream, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&w, c->d_wtab + (size_t)stream * (size_t)c->wmax + (size_t)seq, sizeof(w), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (seq >= wc) return fail(c, RFID_ERR_INVALID, "rfid_batch_get_gated: no such window");
  const int64_t len = w.type ? EPC_WIN : RN16_WIN;
  *n = len;
  const int64_t k = len < cap ? len : cap;
  if (k > 0) {
    HIPCHK(c, hipMemcpyAsync(out, c->d_y + (size_t)stream * (size_t)c->y_stride + (size_t)w.start, sizeof(float2) * (size_t)k,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // in[i] - dc_est: one binary32 subtraction per component, the same operation on any IEEE host (no sample arithmetic
    // beyond this debug tap's formatting happens on the host)
    for (int64_t i = 0; i < k; ++i) {
      volatile float re = out[i].re - w.dc_re, im = out[i].im - w.dc_im;
      out[i].re = re; out[i].im = im;
    }
  }
  return RFID_OK;
}

// ======================================================================================
// (3) synthetic workloads
// ======================================================================================
int rfid_synth_replicas(rfid_ctx *c, const void *d_base, int64_t n_raw, void *d_out, int64_t out_stride, int n_streams,
                        float sigma, uint64_t seed, int64_t first_replica) {
  if (!c || !d_base || !d_out || n_raw < 0 || out_stride < n_raw || n_streams < 0 || first_replica < 0) return RFID_ERR_INVALID;
  if (n_raw == 0 || n_streams == 0) return RFID_OK;
  HIPCHK(c, hipSetDevice(c->device));
  SynthArgs a;
  a.base = (const float2 *)d_base; a.out = (float2 *)d_out; a.n_raw = n_raw; a.out_stride = out_stride;
  a.first_replica = first_replica; a.sigma = sigma; a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32);
  const int64_t per_block = (int64_t)SYNTH_THREADS * SYNTH_PAIRS_PER_THREAD * 2;
  const int64_t blocks = (n_raw + per_block - 1) / per_block;
  for (int s0 = 0; s0 < n_streams; s0 += 65535) {   // grid.y limit
    const int ns = (n_streams - s0 < 65535) ? (n_streams - s0) : 65535;
    SynthArgs b = a;
    b.out = a.out + (int64_t)s0 * out_stride;
    b.first_replica = first_replica + s0;
    hipLaunchKernelGGL(synth_replicas_kernel, dim3((unsigned)blocks, (unsigned)ns), dim3(SYNTH_THREADS), 0, c->stream, b);
    HIPCHK(c, hipGetLastError());
  }
  return RFID_OK;
}


}  // extern "C"

extern "C" {

int rfid_synth_gen2_size(const rfid_synth_gen2_params *p, const rfid_synth_slot *slots, int64_t n_slots, int64_t *n_raw) {
  if (!p || n_slots < 0 || (n_slots > 0 && !slots) || !n_raw) return RFID_ERR_INVALID;
  const int64_t n = rfidh::gen2_layout(*p, slots, n_slots, nullptr);
  if (n < 0) return RFID_ERR_INVALID;
  *n_raw = n;
  return RFID_OK;
}

int rfid_synth_gen2(rfid_ctx *c, const rfid_synth_gen2_params *p, const rfid_synth_slot *slots, int64_t n_slots,
                    void *d_out, int64_t out_cap, float sigma, uint64_t seed, int64_t replica, int64_t *n_raw) {
  if (!c || !p || n_slots < 0 || (n_slots > 0 && !slots) || !d_out || replica < 0) return RFID_ERR_INVALID;
  if (((uintptr_t)d_out) & 15) return fail(c, RFID_ERR_INVALID, "rfid_synth_gen2: d_out must be 16-byte aligned");
  std::vector<Gen2SlotDev> dev;
  dev.reserve((size_t)n_slots + 2);
  const int64_t total = rfidh::gen2_layout(*p, slots, n_slots, &dev);
  if (total < 0) return fail(c, RFID_ERR_INVALID, "rfid_synth_gen2: bad slot table");
  if (total > out_cap) return RFID_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = grow(c, c->synth_tab, sizeof(Gen2SlotDev) * dev.size());
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->synth_tab.p, dev.data(), sizeof(Gen2SlotDev) * dev.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));   // `dev` is pageable host memory that dies with this call
  Gen2Args a;
  memset(&a, 0, sizeof(a));
  a.slots = (const Gen2SlotDev *)c->synth_tab.p; a.n_slots = (int64_t)dev.size();
  a.out = (float2 *)d_out; a.n_raw = total;
  a.leak_re = p->leak_re; a.leak_im = p->leak_im;
  for (int k = 0; k < G2_MAX_TAGS; ++k) { a.h_re[k] = p->h_re[k]; a.h_im[k] = p->h_im[k]; }
  a.sigma = sigma; a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32); a.replica = (uint64_t)replica;
  const int64_t n = (int64_t)dev.size();
  const unsigned gx = (unsigned)(n < 32768 ? n : 32768), gy = (unsigned)((n + gx - 1) / gx);
  hipLaunchKernelGGL(synth_gen2_kernel, dim3(gx, gy), dim3(G2_THREADS), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  if (n_raw) *n_raw = total;
  return RFID_OK;
}

}  // extern "C"

extern "C" {

// ======================================================================================
// (1) streaming per-block path (host buffers)
// ======================================================================================
int rfid_mf_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_produced) {
  if (!c || n_in < 0 || (n_in > 0 && !in) || !n_produced) return RFID_ERR_INVALID;
  *n_produced = 0;
  if (n_in == 0 && !(c->la.on && c->la.late && !c->la.held.empty())) return RFID_OK;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->la.on) {
    if (c->sio.ymode) return fail(c, RFID_ERR_STATE, "look-ahead: keyed on the gate's input (rfid_lookahead_enable_gate), rfid_mf_work has no part in it");
    // a call larger than the max_chunk_raw the look-ahead was sized for goes through in pieces (multiples of the decimation)
    const int64_t piece = (c->sio.max_chunk / DECIM) * DECIM;
    if (c->la.late) {   // (late outputs: one call, one set of outputs held back)
      if (n_in > piece) return fail(c, RFID_ERR_CAPACITY, "look-ahead with late outputs: a rfid_mf_work call takes at most the max_chunk_raw given to rfid_lookahead_enable");
      return la_mf_work(c, in, n_in, out, out_cap, n_produced);
    }
    int done = 0, made = 0;
    while (done < n_in) {
      const int take = (int)((n_in - done < piece) ? (n_in - done) : piece);
      int k = 0;
      const int rc = la_mf_work(c, in + done, take, out ? out + made : out, out_cap - made, &k);
      if (rc) { *n_produced = made; return rc; }
      done += take; made += k;
    }
    *n_produced = made;
    return RFID_OK;
  }
  // A decimating GNU Radio block produces output n once the whole group x[5n .. 5n+4] has arrived
  // (sync_decimator: noutput = ninput / decim), although y[n] only needs x[5n-24 .. 5n]: a stream of
  // N samples yields floor(N/5) outputs, the same as rfid_batch_mf / the fused front end.
  const int H = rfid_ctx::MF_HIST;
  const int64_t n_first = c->mf_seen / DECIM;                       // first output not yet produced
  const int n_out = (int)((c->mf_seen + n_in) / DECIM - n_first);
  // staging[0] is raw sample mf_seen - H; y[n_first] starts at raw 5 n_first - 24
  const int off = (int)(DECIM * n_first - (NTAPS - 1) - (c->mf_seen - H));   // 0..4
  if (n_out > out_cap || (n_out > 0 && !out)) return RFID_ERR_CAPACITY;
  if (n_out > 0) {
    int rc = grow(c, c->s_in, sizeof(float2) * (size_t)(H + n_in + 2));
    if (rc) return rc;
    if ((rc = grow(c, c->s_out, sizeof(float2) * (size_t)(n_out + 2)))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->s_in.p, c->mf_hist, sizeof(rfid_cf32) * H, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync((float2 *)c->s_in.p + H, in, sizeof(rfid_cf32) * (size_t)n_in, hipMemcpyHostToDevice,
                             c->stream));
    MfArgs a;
    a.x = (const float2 *)c->s_in.p; a.x_stride = H + n_in; a.n_raw = H + n_in; a.lens = nullptr;
    a.n_out = n_out; a.in_off = off; a.vec_ok = (off % 2 == 0) ? 1 : 0;
    a.y = (float2 *)c->s_out.p; a.y_stride = n_out; a.tile0 = 0; a.stream0 = 0;
    const int tiles = (n_out + MF_TILE - 1) / MF_TILE;
    hipLaunchKernelGGL(mf_boxcar25_decim5_kernel, dim3((unsigned)tiles, 1), dim3(MF_THREADS), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->s_out.p, sizeof(rfid_cf32) * (size_t)n_out, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  // roll the history
  if (n_in >= H) {
    memcpy(c->mf_hist, in + n_in - H, sizeof(rfid_cf32) * H);
  } else {
    memmove(c->mf_hist, c->mf_hist + n_in, sizeof(rfid_cf32) * (size_t)(H - n_in));
    memcpy(c->mf_hist + H - n_in, in, sizeof(rfid_cf32) * (size_t)n_in);
  }
  c->mf_seen += n_in;
  *n_produced = n_out;
  return RFID_OK;
}

int rfid_gate_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_consumed,
                   int *n_written) {
  if (!c || n_in < 0 || (n_in > 0 && !in) || !n_consumed || !n_written) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  rfid_reader_state &rs = c->rs;
  *n_consumed = n_in;
  *n_written = 0;
  // gate_impl.cc:101-109
  if ((rs.n_queries_sent > c->prm.max_num_queries || rs.n_unique_tags > c->prm.number_unique_tags) &&
      rs.status != RFID_TERMINATED)
    rs.status = RFID_TERMINATED;
  // gate_impl.cc:112-123: SEEK_* -> CLOSED, arm n_samples_to_ungate, n_samples = 0
  if (rs.gate_status == RFID_GATE_SEEK_EPC || rs.gate_status == RFID_GATE_SEEK_RN16) {
    const int type = (rs.gate_status == RFID_GATE_SEEK_EPC) ? 1 : 0;
    rs.gate_status = RFID_GATE_CLOSED;
    rs.n_samples_to_ungate = type ? EPC_WIN : RN16_WIN;
    c->la.need_arm = false;
    if (!c->la.on) {
      hipLaunchKernelGGL(gate_arm_kernel, dim3(1), dim3(64), 0, c->stream, c->d_gate1, rs.n_samples_to_ungate, type);
      HIPCHK(c, hipGetLastError());   // (stream-ordered before the scan below: no host round trip)
    }
  }
  if (c->la.on) {
    // (the SEEK_* -> CLOSED arming above touched the per-call gate state only, which the look-ahead does not use)
    if (rs.status != RFID_RUNNING) { c->la.gate_pos += n_in; c->la.last_m2.clear(); c->la.y_drop_before(c->la.gate_pos); return RFID_OK; }
    if (c->la.consume_ahead) {   // (a call without input hands out what the passes have found meanwhile)
      if (!out || out_cap < 1) return RFID_ERR_CAPACITY;
      return la_gate_swallow(c, in, n_in, out, out_cap, n_consumed, n_written);
    }
    if (n_in == 0) return RFID_OK;
    if (!out || out_cap < n_in) return RFID_ERR_CAPACITY;
    return la_gate_work(c, in, n_in, out, out_cap, n_consumed, n_written);
  }
  if (rs.status != RFID_RUNNING || n_in == 0) return RFID_OK;
  if (!out || out_cap < n_in) return RFID_ERR_CAPACITY;  // a call can emit up to n_in samples
  int rc = grow(c, c->s_in, sizeof(float2) * (size_t)(n_in + 2));
  if (rc) return rc;
  if ((rc = grow(c, c->s_out, sizeof(float2) * (size_t)(n_in + 2)))) return rc;
  HIPCHK(c, hipMemcpyAsync(c->s_in.p, in, sizeof(rfid_cf32) * (size_t)n_in, hipMemcpyHostToDevice, c->stream));
  GateArgs a = {};
  a.y = (const float2 *)c->s_in.p; a.y_stride = n_in; a.n_dec = n_in; a.lens = nullptr;
  a.pos0 = 0; a.chunk_len = n_in;
  a.state = c->d_gate1; a.n_streams = 1; a.wtab = nullptr; a.wmax = 0; a.wcount = nullptr;
  a.flat = nullptr; a.flat_count = nullptr; a.flat_cap = 0; a.mode = 1;
  a.gated = (float2 *)c->s_out.p; a.gated_cap = n_in; a.io = c->d_io;
  hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(GATE_THREADS), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  int io[2] = {0, 0};
  int open_now = 0;
  HIPCHK(c, hipMemcpyAsync(io, c->d_io, sizeof(io), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&open_now, &c->d_gate1->gate_open, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (io[1] > 0)
    HIPCHK(c, hipMemcpy(out, c->s_out.p, sizeof(rfid_cf32) * (size_t)io[1], hipMemcpyDeviceToHost));
  rs.gate_status = open_now ? RFID_GATE_OPEN : RFID_GATE_CLOSED;
  *n_consumed = io[0];
  *n_written = io[1];
  return RFID_OK;
}

int rfid_decoder_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, float *out_bits, int out_cap, int *n_consumed,
                      int *n_produced, rfid_decode_result *res_out, rfid_scores *scores_out) {
  LaTimer tm(2);
  if (!c || n_in < 0 || (n_in > 0 && !in) || !n_consumed || !n_produced) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  rfid_reader_state &rs = c->rs;
  *n_consumed = 0;
  *n_produced = 0;
  const int ung = rs.n_samples_to_ungate;
  // tag_decoder_impl.cc:223 / :291 -- act only on a complete window
  if (ung <= 0 || n_in < ung) return RFID_OK;
  const int type = rs.decoder_status;
  const int wlen = (type == RFID_DECODE_EPC) ? EPC_WIN : RN16_WIN;
  if (ung != wlen) return fail(c, RFID_ERR_STATE, "n_samples_to_ungate does not match decoder_status");
  if (type == RFID_DECODE_RN16 && (!out_bits || out_cap < 16)) return RFID_ERR_CAPACITY;
  rfid_decode_result r;
  rfid_scores sc;
  memset(&sc, 0, sizeof(sc));
  // look-ahead: the window the gate handed out is retired whenever a decoder call consumes it -- with scores wanted the
  // device decodes it again (the cached pass kept no scores), but the entry goes all the same
  rfid_decode_result r_la;
  const bool in_la = c->la.on && la_decoder_result(c, in, wlen, type, &r_la);
  const bool from_la = in_la && !scores_out;
  if (from_la) r = r_la;
  if (!from_la) {
  int rc = grow(c, c->s_in, sizeof(float2) * (size_t)(wlen + 2));
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->s_in.p, in, sizeof(rfid_cf32) * (size_t)wlen, hipMemcpyHostToDevice, c->stream));
  rfid_window w;
  w.stream = 0; w.seq = 0; w.start = 0; w.type = type; w.dc_re = 0.0f; w.dc_im = 0.0f;  // input is already DC-free
  const int one = 1;
  HIPCHK(c, hipMemcpyAsync(c->d_swin, &w, sizeof(w), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_scount, &one, sizeof(int), hipMemcpyHostToDevice, c->stream));
  DecodeArgs a;
  a.y = (const float2 *)c->s_in.p; a.y_stride = wlen; a.flat = c->d_swin; a.flat_count = c->d_scount;
  a.flat_cap = 1; a.res = c->d_sres; a.scores = c->d_sscores; a.wmax = 1;
  memcpy(a.t_cand, c->t_cand, sizeof(a.t_cand));
  hipLaunchKernelGGL(decode_windows_kernel, dim3(1), dim3(64), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(&r, c->d_sres, sizeof(r), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&sc, c->d_sscores, sizeof(sc), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (res_out) *res_out = r;
  if (scores_out) *scores_out = sc;
  if (type == RFID_DECODE_RN16) {
    // tag_decoder_impl.cc:256-268 (the else branch :269-288 is unreachable once n_in >= 250)
    for (int b = 0; b < 16; ++b) out_bits[b] = (float)((r.bits[0] >> b) & 1u);
    *n_produced = 16;
    rs.gen2_logic_status = RFID_SEND_ACK;
  } else {
    rs.cur_slot_number++;  // :295
    if (r.crc_ok) {        // :328-364
      next_slot(rs);
      rs.n_epc_correct += 1;
      const int id = r.tag_id & 255;
      if (rs.tag_reads[id] == 0) rs.n_unique_tags++;
      rs.tag_reads[id]++;
    } else {               // :366-387
      next_slot(rs);
    }
  }
  *n_consumed = ung;
  return RFID_OK;
}

int rfid_reader_work(rfid_ctx *c, int n_in, int *n_consumed) {  // reader_impl.cc:200-380
  if (!c) return RFID_ERR_INVALID;
  rfid_reader_state &rs = c->rs;
  if (n_consumed) *n_consumed = n_in;  // consume_each(ninput_items[0]) :378
  switch (rs.gen2_logic_status) {
    case RFID_START: rs.gen2_logic_status = RFID_SEND_QUERY; break;            // :218-224
    case RFID_POWER_DOWN: rs.gen2_logic_status = RFID_START; break;            // :226-231
    case RFID_SEND_NAK_QR: rs.gen2_logic_status = RFID_SEND_QUERY_REP; break;  // :233-240
    case RFID_SEND_NAK_Q: rs.gen2_logic_status = RFID_SEND_QUERY; break;       // :242-249
    case RFID_SEND_QUERY:                                                      // :251-288
      rs.n_queries_sent += 1;
      rs.decoder_status = RFID_DECODE_RN16;
      rs.gate_status = RFID_GATE_SEEK_RN16;
      rs.gen2_logic_status = RFID_IDLE;
      break;
    case RFID_SEND_ACK:                                                        // :290-320
      if (n_in == 16) {
        rs.decoder_status = RFID_DECODE_EPC;
        rs.gate_status = RFID_GATE_SEEK_EPC;
        rs.gen2_logic_status = RFID_SEND_CW;
      }
      break;
    case RFID_SEND_CW: rs.gen2_logic_status = RFID_IDLE; break;                // :322-327
    case RFID_SEND_QUERY_REP:                                                  // :329-344
    case RFID_SEND_QUERY_ADJUST:                                               // :346-372
      rs.decoder_status = RFID_DECODE_RN16;
      rs.gate_status = RFID_GATE_SEEK_RN16;
      rs.n_queries_sent += 1;
      rs.gen2_logic_status = RFID_IDLE;
      break;
    default: break;
  }
  return RFID_OK;
}

}  // extern "C"


#include "rfid_capi_stream.hpp"   // (1b) whole-chain streaming, (1c) the look-ahead of the per-block calls

// ---- reader TX waveform (reader_impl.cc:43-129 tables, :131-162 command bits, :383-443 CRC-5) ------------
namespace {
struct ReaderTx {
  int dac_rate = 0, fixed_q = -1;
  std::vector<float> data_0, data_1, cw, cw_ack, cw_query, delim, frame_sync, preamble, rtcal, trcal, query_bits,
      query_rep, nak, query_adjust_bits, p_down;
};
void append(std::vector<float> &dst, const std::vector<float> &src) { dst.insert(dst.end(), src.begin(), src.end()); }
void build_reader_tx(ReaderTx &t, int dac_rate, int fixed_q) {
  t = ReaderTx();
  t.dac_rate = dac_rate; t.fixed_q = fixed_q;
  const float sample_d = (float)(1.0 / dac_rate * pow(10, 6));
  const float n_data0_s = 2 * 12 / sample_d, n_data1_s = 4 * 12 / sample_d, n_pw_s = 12 / sample_d;   // PW_D = 12 us
  const float n_cw_s = 250 / sample_d, n_delim_s = 12 / sample_d, n_trcal_s = 200 / sample_d;
  const int n_cwquery_s = (int)((240 + 480 + 575) / sample_d);      // T1 + T2 + RN16
  const int n_cwack_s = (int)((3 * 240 + 480 + 3375) / sample_d);   // 3 T1 + T2 + EPC
  t.p_down.assign((size_t)(2000 / sample_d), 0.0f);
  t.cw_query.assign((size_t)n_cwquery_s, 1.0f);
  t.cw_ack.assign((size_t)n_cwack_s, 1.0f);
  t.data_0.assign((size_t)n_data0_s, 0.0f);
  t.data_1.assign((size_t)n_data1_s, 0.0f);
  t.cw.assign((size_t)n_cw_s, 1.0f);
  t.delim.assign((size_t)n_delim_s, 0.0f);
  t.rtcal.assign((size_t)(n_data0_s + n_data1_s), 0.0f);
  t.trcal.assign((size_t)n_trcal_s, 0.0f);
  std::fill_n(t.data_0.begin(), t.data_0.size() / 2, 1.0f);
  std::fill_n(t.data_1.begin(), 3 * t.data_1.size() / 4, 1.0f);
  std::fill_n(t.rtcal.begin(), (size_t)((float)t.rtcal.size() - n_pw_s), 1.0f);
  std::fill_n(t.trcal.begin(), (size_t)((float)t.trcal.size() - n_pw_s), 1.0f);
  append(t.preamble, t.delim); append(t.preamble, t.data_0); append(t.preamble, t.rtcal); append(t.preamble, t.trcal);
  append(t.frame_sync, t.delim); append(t.frame_sync, t.data_0); append(t.frame_sync, t.rtcal);
  append(t.query_rep, t.frame_sync);
  for (int i = 0; i < 4; ++i) append(t.query_rep, t.data_0);
  append(t.nak, t.frame_sync);
  append(t.nak, t.data_1); append(t.nak, t.data_1);
  for (int i = 0; i < 6; ++i) append(t.nak, t.data_0);
  // Query: 1000 DR M(2) TRext Sel(2) Session(2) Target Q(4) + CRC-5
  const int head[13] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int b : head) t.query_bits.push_back((float)b);
  for (int i = 3; i >= 0; --i) t.query_bits.push_back((float)((fixed_q >> i) & 1));
  unsigned reg = 0x09;   // 01001
  for (int i = 0; i < 17; ++i) {
    const unsigned fb = ((reg >> 4) & 1u) ^ (t.query_bits[(size_t)i] == 1.0f ? 1u : 0u);
    reg = (reg << 1) & 0x1Fu;
    if (fb) reg ^= 0x09;
  }
  for (int i = 4; i >= 0; --i) t.query_bits.push_back((float)((reg >> i) & 1u));
  const int qadj[9] = {1, 0, 0, 1, 0, 0, 0, 0, 0};   // QADJ_CODE, SESSION, Q_UPDN[1] (unchanged)
  for (int b : qadj) t.query_adjust_bits.push_back((float)b);
}
size_t pie_len(const ReaderTx &t, const std::vector<float> &bits) {
  size_t n = 0;
  for (float b : bits) n += (b == 1.0f) ? t.data_1.size() : t.data_0.size();
  return n;
}
void emit(float *out, int &w, const std::vector<float> &v) {
  if (!v.empty()) memcpy(out + w, v.data(), sizeof(float) * v.size());
  w += (int)v.size();
}
void emit_bits(const ReaderTx &t, float *out, int &w, const float *bits, int n) {
  for (int i = 0; i < n; ++i) emit(out, w, bits[i] == 1.0f ? t.data_1 : t.data_0);
}
}  // namespace

extern "C" {

int rfid_reader_tx_max(int dac_rate) {
  if (dac_rate <= 0) return 0;
  ReaderTx t;
  build_reader_tx(t, dac_rate, 15);
  // the longest outputs: Query (all-ones upper bound on its bits) and START / SEND_CW (cw_ack)
  const size_t q = t.preamble.size() + 22 * t.data_1.size() + t.cw_query.size();
  const size_t m = q > t.cw_ack.size() ? q : t.cw_ack.size();
  return (int)m;
}

int rfid_reader_work_tx(rfid_ctx *c, int dac_rate, const float *in_bits, int n_in, float *out, int out_cap, int *n_consumed,
                        int *n_written) {
  LaTimer tm(3);
  if (!c || dac_rate <= 0 || n_in < 0 || (!out && out_cap > 0) || out_cap < 0) return RFID_ERR_INVALID;
  static thread_local ReaderTx t;
  if (t.dac_rate != dac_rate || t.fixed_q != c->prm.fixed_q) build_reader_tx(t, dac_rate, c->prm.fixed_q);
  const rfid_reader_state &rs = c->rs;
  // how much the state at hand writes (reader_impl.cc:216-373)
  size_t need = 0;
  std::vector<float> ack_bits;
  switch (rs.gen2_logic_status) {
    case RFID_START: case RFID_SEND_CW: need = t.cw_ack.size(); break;
    case RFID_POWER_DOWN: need = t.p_down.size(); break;
    case RFID_SEND_NAK_QR: case RFID_SEND_NAK_Q: need = t.nak.size() + t.cw.size(); break;
    case RFID_SEND_QUERY: need = t.preamble.size() + pie_len(t, t.query_bits) + t.cw_query.size(); break;
    case RFID_SEND_ACK:
      if (n_in == 16) {
        if (!in_bits) return RFID_ERR_INVALID;
        ack_bits.push_back(0.0f); ack_bits.push_back(1.0f);   // ACK_CODE
        ack_bits.insert(ack_bits.end(), in_bits, in_bits + 16);
        need = t.frame_sync.size() + pie_len(t, ack_bits);
      }
      break;
    case RFID_SEND_QUERY_REP: need = t.query_rep.size() + t.cw_query.size(); break;
    case RFID_SEND_QUERY_ADJUST: need = t.frame_sync.size() + pie_len(t, t.query_adjust_bits) + t.cw_query.size(); break;
    default: break;
  }
  if (need > (size_t)out_cap) {
    snprintf(c->err, sizeof(c->err), "rfid_reader_work_tx: %zu floats needed, out_cap %d", need, out_cap);
    return RFID_ERR_CAPACITY;
  }
  int w = 0;
  switch (rs.gen2_logic_status) {
    case RFID_START: case RFID_SEND_CW: emit(out, w, t.cw_ack); break;
    case RFID_POWER_DOWN: emit(out, w, t.p_down); break;
    case RFID_SEND_NAK_QR: case RFID_SEND_NAK_Q: emit(out, w, t.nak); emit(out, w, t.cw); break;
    case RFID_SEND_QUERY:
      emit(out, w, t.preamble);
      emit_bits(t, out, w, t.query_bits.data(), (int)t.query_bits.size());
      emit(out, w, t.cw_query);
      break;
    case RFID_SEND_ACK:
      if (n_in == 16) { emit(out, w, t.frame_sync); emit_bits(t, out, w, ack_bits.data(), (int)ack_bits.size()); }
      break;
    case RFID_SEND_QUERY_REP: emit(out, w, t.query_rep); emit(out, w, t.cw_query); break;
    case RFID_SEND_QUERY_ADJUST:
      emit(out, w, t.frame_sync);
      emit_bits(t, out, w, t.query_adjust_bits.data(), (int)t.query_adjust_bits.size());
      emit(out, w, t.cw_query);
      break;
    default: break;
  }
  if (n_written) *n_written = w;
  return rfid_reader_work(c, n_in, n_consumed);   // the state transitions
}

}  // extern "C"
