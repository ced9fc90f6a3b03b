/*
 * rfid_mi355x.h -- C-ABI of the MI355X-native Gen2 RFID receive path
 *                  (matched filter -> gate -> tag_decoder), librfid_mi355x.so.
 *
 * This is the drop-in boundary for the hot path of nkargas/Gen2-UHF-RFID-Reader.
 * Every entry point names the reference interface it replaces (paths relative to
 * /root/reference/gr-rfid/).  Plain C: pointers and sizes only, no C++/torch types,
 * no exceptions; every function returns an rfid_status (0 = ok, <0 = error) and never
 * aborts.  There is NO CPU fallback: if no gfx950 device / HIP runtime is usable,
 * rfid_ctx_create fails with RFID_ERR_NO_DEVICE.
 *
 * Sample format everywhere: interleaved little-endian float32 I,Q (= gr_complex =
 * the reference's file_source format, misc/code/plot_signal.m:5-9).
 *
 * Two families of entry points:
 *   (1) per-block streaming calls on HOST buffers -- one per reference work():
 *         rfid_mf_work       <- filter.fir_filter_ccc(5,[1]*25)      apps/reader.py:65,75
 *         rfid_gate_work     <- gate_impl::general_work               lib/gate_impl.cc:85-200
 *         rfid_decoder_work  <- tag_decoder_impl::general_work        lib/tag_decoder_impl.cc:196-397
 *         rfid_reader_work_tx <- reader_impl::general_work (state transitions + transmit waveform)
 *         rfid_reader_work    <- its state transitions alone
 *                                                                     lib/reader_impl.cc:200-380
 *   (2) batched offline calls on DEVICE buffers (many independent traces per launch):
 *         rfid_batch_mf / rfid_batch_gate / rfid_batch_decode / rfid_batch_stats and the
 *         fused driver rfid_batch_process -- the same arithmetic, all windows of all
 *         traces decoded in one launch.
 *
 * Threading: one rfid_ctx = one GPU + one HIP stream.  Calls on a ctx must be
 * serialised by the caller; distinct contexts are independent (multi-GPU = one ctx,
 * one process, per device; no collective).
 */
#ifndef RFID_MI355X_H
#define RFID_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFID_API __attribute__((visibility("default")))
#define RFID_MI355X_ABI 5   /* see rfid_abi_version() */

typedef enum rfid_status {
  RFID_OK = 0,
  RFID_ERR_INVALID = -1,      /* bad argument */
  RFID_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime */
  RFID_ERR_HIP = -3,          /* a HIP call failed: see rfid_last_error() */
  RFID_ERR_UNSUPPORTED = -4,  /* parameter combination the kernels are not built for */
  RFID_ERR_CAPACITY = -5,     /* caller buffer or planned workspace too small */
  RFID_ERR_STATE = -6         /* call not valid in the current state (e.g. no plan) */
} rfid_status;

/* gr_complex */
typedef struct rfid_cf32 { float re, im; } rfid_cf32;

/* enums of include/rfid/global_vars.h:31-34, same numeric values */
enum { RFID_RUNNING = 0, RFID_TERMINATED = 1 };
enum { RFID_SEND_QUERY = 0, RFID_SEND_ACK, RFID_SEND_QUERY_REP, RFID_IDLE, RFID_SEND_CW, RFID_START,
       RFID_SEND_QUERY_ADJUST, RFID_SEND_NAK_QR, RFID_SEND_NAK_Q, RFID_POWER_DOWN };
enum { RFID_GATE_OPEN = 0, RFID_GATE_CLOSED, RFID_GATE_SEEK_RN16, RFID_GATE_SEEK_EPC };
enum { RFID_DECODE_RN16 = 0, RFID_DECODE_EPC = 1 };

/* Construction parameters.  Replaces the ctor arguments gate::make(int sample_rate)
 * (include/rfid/gate.h:51), tag_decoder::make(int sample_rate) (include/rfid/tag_decoder.h:48),
 * the flowgraph's decim / num_taps (apps/reader.py:54,65) and the compile-time constants
 * FIXED_Q, MAX_NUM_QUERIES, NUMBER_UNIQUE_TAGS (include/rfid/global_vars.h:72,76,100). */
typedef struct rfid_params {
  int32_t sample_rate;        /* rate after decimation; only 400000 is supported */
  int32_t decim;              /* 5  */
  int32_t n_taps;             /* 25, all-ones */
  int32_t fixed_q;            /* 0..15 */
  int32_t max_num_queries;    /* 1000 */
  int32_t number_unique_tags; /* 100 */
} rfid_params;

/* READER_STATE / READER_STATS view (include/rfid/global_vars.h:36-67) */
typedef struct rfid_reader_state {
  int32_t status, gen2_logic_status, gate_status, decoder_status;
  int32_t n_samples_to_ungate;
  int32_t n_queries_sent, cur_inventory_round, cur_slot_number, max_slot_number, n_epc_correct;
  int32_t n_unique_tags;
  int32_t tag_reads[256];     /* std::map<int,int> tag_reads, key = EPC bits 104..111 */
} rfid_reader_state;

/* one gate opening, produced by the batched gate scan */
typedef struct rfid_window {
  int32_t stream;   /* trace index in the batch */
  int32_t seq;      /* 0,1,2,... per trace; type = seq & 1 */
  int32_t start;    /* index (400 ksps domain) of the first gated sample */
  int32_t type;     /* RFID_DECODE_RN16 / RFID_DECODE_EPC */
  float dc_re, dc_im; /* gate_impl::dc_est at the opening (lib/gate_impl.cc:141,176) */
} rfid_window;

/* what tag_decoder computes for one window */
typedef struct rfid_decode_result {
  int32_t type;       /* RFID_DECODE_RN16 / RFID_DECODE_EPC */
  int32_t index;      /* value returned by tag_sync (lib/tag_decoder_impl.cc:107-108) */
  float h_re, h_im;   /* h_est (:103) */
  float T;            /* T_global (:166-169); 0 for RN16 */
  uint32_t bits[4];   /* decoded bits, bit j of the frame at bits[j>>5] bit (j&31) */
  int32_t n_bits;     /* 16 or 128 */
  int32_t crc_ok;     /* EPC: 1 if check_crc()==1 (:401-445), else 0 */
  int32_t tag_id;     /* EPC bits 104..111, MSB first (:348-352); -1 unless crc_ok */
} rfid_decode_result;

/* optional per-window scores for tolerance checks (correlation / energy values) */
typedef struct rfid_scores {
  float corr[15];     /* std::norm(corr2) per candidate offset (lib/tag_decoder_impl.cc:85-99) */
  float energy[20];   /* per half-period candidate (:157-164); zeros for RN16 */
  float pad_;
} rfid_scores;

/* per-trace statistics after the decode of a batch (READER_STATS per trace) */
typedef struct rfid_stream_stats {
  int32_t n_queries_sent, cur_inventory_round, cur_slot_number, n_epc_correct, n_unique_tags;
  int32_t n_windows;       /* complete windows the gate produced */
  int32_t n_windows_used;  /* windows consumed before TERMINATED (lib/gate_impl.cc:101-109) */
  int32_t status;          /* RFID_RUNNING / RFID_TERMINATED */
  int32_t tag_reads[256];
} rfid_stream_stats;

/* timing of the last rfid_batch_* pass, from HIP events on the ctx stream */
typedef struct rfid_batch_timing {
  float mf_ms, gate_ms, decode_ms, stats_ms; /* kernel time per pass (summed over the launches of a pass) */
  float total_ms;       /* wall time of the pass on the device (front end overlapped) */
  float front_ms;       /* wall time of matched filter + gate scan (they overlap on two streams) */
  int32_t front_chunks; /* launches of each of the two front-end kernels in the pass (1 = not chunked) */
  int32_t decode_launches; /* 2: one EPC launch, one RN16 launch */
  int32_t fused_front;  /* 1: rfid_batch_process ran the fused front end (matched filter inside the gate launch:
                           mf_ms = 0, gate_ms = the fused kernel); 2: the long-stream front end with the matched filter inside
                           its first launch (few, long traces: mf_ms = 0, gate_ms = its whole launch list) */
  int32_t reserved_;
} rfid_batch_timing;

/* what the long-stream front end did in the last rfid_batch_process pass (all zero when it was not used).  The front
 * end cuts each trace along time into pieces at idle points of the gate and processes all pieces at once: avg_ampl, then
 * the state machine -- from guessed start values whose runs are PROVEN to cover the true ones (or run again) -- then dc_est,
 * every unit from 64 neighbouring start values at once and the chain of their tables from the trace's exact start (round 6);
 * see csrc/rfid_ls2.hpp. */
typedef struct rfid_ls_report {
  int32_t pieces;           /* pieces the traces were cut into */
  int32_t units;            /* runs of pieces scanned in one go by the state-machine / dc_est passes (= pieces unless a cut
                             * turned out not to be idle: see cuts_dropped) */
  int32_t chunk;            /* nominal piece length, decimated samples */
  int32_t avg_rounds;       /* avg_ampl: chain rounds that had work (1 = every guess was proven at once) */
  int32_t avg_reruns;       /*   pieces run again from their predicted start because their first run did not cover it */
  int32_t fsm_rounds;       /* state machine: rounds */
  int32_t dc_rounds;        /* dc_est: rounds that had work (2: the first round's centres, the ring means, were off by the rounding
                             * drift; more: the sums hover at a binade edge and the settled prefix grew round by round) */
  int32_t dc_reruns;        /*   unit runs after the first round */
  int32_t verified;         /* 1: accepted -- every piece's latest run is exact or proven: the sequential scan, bit for bit */
  int32_t gave_up;          /* != 0: the sequential scan ran instead (1 no trace could be cut, 2 / 3: avg_ampl / state machine not
                             * settled within the round limit, 5: the first pass met a stretch of 32 nominal piece lengths
                             * without 128 carrier samples in a row).  dc_est never gives a pass up: see dc_finished */
  int32_t cuts_dropped;     /* cut points withdrawn because the state machine (or the dc ring) was not idle there */
  int32_t windows;          /* complete windows found */
  int32_t dc_finished;      /* dc_est units that were not settled when the enqueued rounds were used up and went through the
                             * finishing walk (one after the other from the proven value before them; the settled units kept
                             * their results): the partial fallback.  0: the rounds sufficed */
} rfid_ls_report;

typedef struct rfid_ctx rfid_ctx;

/* ---- lifetime ------------------------------------------------------------------------ */
RFID_API int rfid_params_default(rfid_params *p);
/* device = HIP device ordinal.  Fails (RFID_ERR_NO_DEVICE) when it is not a gfx950 GPU. */
RFID_API int rfid_ctx_create(const rfid_params *p, int device, rfid_ctx **out);
RFID_API int rfid_ctx_destroy(rfid_ctx *ctx);
/* resets gate/decoder/reader state as the block constructors + initialize_reader_state()
 * do (lib/gate_impl.cc:41-70, lib/global_vars.cc:34-54) */
RFID_API int rfid_ctx_reset(rfid_ctx *ctx);
RFID_API const char *rfid_strerror(int status);
RFID_API const char *rfid_last_error(const rfid_ctx *ctx);
RFID_API const char *rfid_version(void);
/* RFID_MI355X_ABI of the library that was loaded: it changes whenever a struct of this header changes size or layout
 * (4: rfid_ls_report has 13 fields since round 3; 5: its last field is dc_finished since round 6).  A caller built against another value must not pass structs. */
RFID_API int rfid_abi_version(void);
/* device self-test of the wave-level primitives the kernels rely on (DPP wave shift,
 * IEEE division, double sqrt).  0 = all good, >0 = number of failing checks. */
RFID_API int rfid_selftest(rfid_ctx *ctx, int *n_failed);

/* ---- (1) streaming, host buffers: one call per reference work() ----------------------- */
/* fir_filter_ccc(5,[1]*25): consumes all n_in samples, keeps the filter history and the decimation
 * phase inside ctx.  y[n] = sum_{k=0..24} x[5n-24+k], k ascending; output n is written by the call
 * that completes its decimation group x[5n .. 5n+4] (GNU Radio's sync_decimator: noutput =
 * ninput / decim), so a stream of N samples yields floor(N/5) outputs whatever the call sizes --
 * the same outputs as rfid_batch_mf / rfid_batch_process on the same samples. */
RFID_API int rfid_mf_work(rfid_ctx *ctx, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap,
                          int *n_produced);
/* gate_impl::general_work: scans in[0..n_in), writes gated, DC-removed samples to out and
 * stops right after a window closes (consume_each(i+1), lib/gate_impl.cc:189-194).
 * *n_consumed / *n_written are the values the block passes to consume_each() / returns. */
RFID_API int rfid_gate_work(rfid_ctx *ctx, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap,
                            int *n_consumed, int *n_written);
/* tag_decoder_impl::general_work: acts only when n_in >= n_samples_to_ungate.  RN16: writes 16
 * floats (0.0/1.0) to out_bits (port 0) and sets gen2_logic_status = SEND_ACK.  EPC: decodes,
 * checks CRC, updates the statistics.  *n_consumed as consume_each(); *n_produced = items
 * produced on port 0.  `res`/`scores` (nullable) receive the per-window details. */
RFID_API int rfid_decoder_work(rfid_ctx *ctx, const rfid_cf32 *in, int n_in, float *out_bits,
                               int out_cap, int *n_consumed, int *n_produced,
                               rfid_decode_result *res, rfid_scores *scores);
/* Look-ahead for the three per-block calls above.  Called block by block, every rfid_gate_work / rfid_decoder_work is
 * a host <-> device round trip for a few hundred samples (~6 per inventory slot: slower than one CPU core).  With the
 * look-ahead on, rfid_mf_work runs the WHOLE chain for the buffer it is handed (matched filter -> gate -> tag_decoder in
 * one submission, the machinery of rfid_stream_work, state carried on the device) and keeps the filter output, the
 * gate's windows -- their gated, DC-removed samples included -- and the decoder's results on the host; the gate / decoder
 * calls that follow are answered from there when their input is the data this library itself handed out (the gate's
 * input must be the matched filter's output, the decoder's input the gate's output: first and last sample of every
 * call are compared; anything else is RFID_ERR_STATE), with the same consume / produce counts, outputs and READER_STATE
 * transitions as without it (lib/gate_impl.cc:189-199, lib/tag_decoder_impl.cc:223,266,289,393-396).  The window types
 * are those of the reference's own control flow (RN16, EPC, RN16, ...: after an RN16 the reader always ACKs).
 * Since round 5 the calls GATHER: a rfid_mf_work call uploads and filters its own samples (its outputs are what it
 * returns) and a whole-chain pass over everything pending is submitted once 65 536 decimated samples have gathered
 * (rfid_lookahead_set_coalesce) -- a pass per 8 192-item scheduler buffer was a third of one CPU core's speed.  The pass
 * stays on the device while the next calls upload into the other buffer.
 * What lies behind the last idle point of the gate in the samples the passes have seen is not decided yet: a gate call
 * consumes up to there and no further.  A gate call that can decide nothing returns (0 consumed, 0 written) ONCE per
 * arrival of new samples -- the scheduler brings more; asked again without anything new (the input has paused or ended:
 * a scheduler calls a block again when its upstream neighbour is done), or shown 2 x the gathering threshold or more
 * (a bounded buffer must drain), it makes the device decide at once: the pass under way is waited for, a pass goes
 * over whatever is pending, and what even that leaves undecided goes through the exact per-call scan (the streaming
 * gate_scan_kernel from the carried state: gate_impl.cc:127-196 sample by sample, as without the look-ahead).  Every
 * sample is consumed and every window handed out whether or not rfid_lookahead_flush is ever called.
 * max_chunk_raw: the largest rfid_mf_work call to expect (a larger one goes through in pieces); the staging holds what
 * gathers + one such call.  Call before the first sample; rfid_ctx_reset switches it off.  Like rfid_stream_begin it runs
 * on a one-trace plan of its own: any rfid_batch_plan of the context is replaced.  (A reader that answers tags in
 * real time cannot wait for 164 ms of signal to gather: RFID_LOOKAHEAD=0 / no rfid_lookahead_enable is the path for that.) */
RFID_API int rfid_lookahead_enable(rfid_ctx *ctx, int64_t max_chunk_raw);
/* The same look-ahead keyed on the GATE's input, for a flowgraph whose matched filter is not this library's --
 * apps/reader.py:75 instantiates GNU Radio's own filter.fir_filter_ccc, so with that file unchanged the first buffer
 * this library sees is the gate's (apps/reader.py:76,106-107).  rfid_gate_work then uploads whatever part of its input
 * the device has not seen yet (a scheduler shows unconsumed samples again; the new ones lie behind them), runs
 * gate -> tag_decoder over it in one submission from the carried state, and answers this and the following gate /
 * decoder calls from the cache, exactly as above (same consume / produce counts, outputs and READER_STATE transitions,
 * same rules for what is undecided, rfid_lookahead_flush at the end of the input).  There is no restriction on where the
 * gate's input comes from; rfid_mf_work is not to be called on such a context.  max_items: decimated samples the device
 * takes per call at most (a larger call is taken in parts).  Call before the first sample.  The end of the input needs
 * no announcement (see above: the second fruitless call decides everything); rfid_lookahead_flush saves that one call. */
RFID_API int rfid_lookahead_enable_gate(rfid_ctx *ctx, int64_t max_items);
/* How many decimated samples gather before a whole-chain pass is submitted (default 65 536; clamped to [1 024, half the
 * staging]).  An adaptor that knows its scheduler's buffers tells the library here: with input buffers of C items the
 * gate can never be shown more than C, so it asks for C / 4 (a gate call shown 2 x this many items decides at once). */
RFID_API int rfid_lookahead_set_coalesce(rfid_ctx *ctx, int64_t items);
/* Late filter outputs (look-ahead keyed on rfid_mf_work only; off by default).  A rfid_mf_work call that returns its own
 * outputs waits for its samples' way through the device: ~17-28 us per call, a third of a block-by-block run at GNU Radio's
 * default buffers.  With late outputs on, a call stages and filters its samples as before but RETURNS OUTPUTS OF THE CALLS
 * BEFORE IT -- whatever the device has finished meanwhile, oldest first, as far as out_cap goes -- and holds its own back
 * (unless the device is through with them before the call returns):
 * *n_produced is what earlier calls made, not n_in / 5 of this one (it may be 0: the call has consumed its input all the
 * same).  A gr::block may do that (general_work consumes and produces what it says): the adaptor's forecast() asks for no
 * input while outputs are held back, so a scheduler calls the block again at the end of the input -- with n_in = 0: such a
 * call hands out what is held back (waiting for the device if need be: it always hands out something; also after
 * rfid_lookahead_flush) and does nothing else.  At most three sets of outputs are held back: a call that brings new
 * samples while three are held needs room for the rest of the oldest (RFID_ERR_CAPACITY otherwise, nothing consumed) -- an
 * adaptor whose output room is smaller than rfid_mf_pending() simply calls with n_in = 0 first.  A call takes at most the
 * max_chunk_raw given to rfid_lookahead_enable.  The gate / decoder calls see no difference: the gate is shown the
 * filter's outputs a call or two later, the passes run over what was uploaded. */
RFID_API int rfid_lookahead_set_late_outputs(rfid_ctx *ctx, int on);
/* filter outputs a late-outputs context holds back (0 otherwise) */
RFID_API int rfid_mf_pending(const rfid_ctx *ctx, int *n_outputs);
/* ... and how many of them the next call must have room for if it brings new samples (0 unless three sets are held back: then
 * what is left of the oldest) */
RFID_API int rfid_mf_must_fetch(const rfid_ctx *ctx, int *n_outputs);
/* What the adaptor knows about its scheduler: the buffer on the gate's input side holds gate_buffer_items items and no more
 * (GNU Radio: detail()->input(0)->max_possible_items_available(); 65 536-byte buffers = 8 192 items by default), or 0: the
 * queues between the blocks grow as needed (the single-threaded scheduler of rfid/mi355x.h).  Bounded: a quarter of the
 * buffer gathers before a pass, and a gate call that can decide nothing never answers (0, 0) -- it makes the device
 * decide at once (a block that moves nothing is, depending on the runtime, polled again at once or left alone for good:
 * neither helps anything gather).  Unbounded (the default): (0, 0) once per arrival of new samples, see above. */
RFID_API int rfid_lookahead_set_scheduler(rfid_ctx *ctx, int64_t gate_buffer_items);
/* Consume ahead (either keying; off by default; before the first gate call).  As described above, a gate call consumes up to
 * the point its windows are known and no further -- under a scheduler with small bounded buffers the buffer in front of
 * the gate is then full of samples the device already has, and nothing gathers.  With consume-ahead on, rfid_gate_work
 * CONSUMES EVERYTHING IT IS SHOWN (keyed on the gate: uploads it) and hands out the windows when the passes have found
 * them: the same windows, the same samples, in the same order, one window per call at most, the next one only after the
 * decoder / reader calls have armed the gate for it (gate_impl.cc:112-123) -- only *n_consumed no longer says where the
 * windows lie.  A call may bring no input (n_in = 0; out_cap >= 1): it hands out what has become known.  Nothing is forced in
 * this mode, so the adaptor must do two things a gr::block can do: (1) forecast() through rfid_gate_forecast -- the gate
 * needs no input while a window lies ready for it, or when its upstream neighbour is done and the device still holds
 * undecided samples; (2) when general_work is called with its upstream done and no input left, call rfid_lookahead_flush
 * (which then decides everything, whatever the keying) before rfid_gate_work.  rfid_lookahead_set_scheduler's bounded mode
 * is switched off by it (the gathering threshold goes back to 65 536). */
RFID_API int rfid_lookahead_set_consume_ahead(rfid_ctx *ctx, int on);
/* forecast() of a consume-ahead gate: *needs_input = 0 when a rfid_gate_work call can do something without input (see above),
 * 1 otherwise (always 1 without consume-ahead: the reference's forecast, gate_impl.cc:79-83).  upstream_done: the block that
 * feeds the gate has finished (GNU Radio: detail()->input(0)->done()). */
RFID_API int rfid_gate_forecast(const rfid_ctx *ctx, int upstream_done, int *needs_input);
/* windows the look-ahead holds: found by the passes and not yet (completely) handed out by rfid_gate_work /
 * handed out and waiting for their rfid_decoder_work call (a decoder call retires its window whether or not it asks for
 * scores).  Both stay small in a running flowgraph. */
RFID_API int rfid_lookahead_pending(const rfid_ctx *ctx, int *gate_windows, int *decoder_windows);
/* End of the input (a file source has run dry): everything still held back is decided now; the gate / decoder / reader
 * calls that follow hand it out.  rfid_mf_work fails with RFID_ERR_STATE afterwards.  No-op without look-ahead.
 * Optional since round 5: a flowgraph that never calls it loses nothing (the gate's second fruitless call decides what
 * is held back), it only pays that one extra call. */
RFID_API int rfid_lookahead_flush(rfid_ctx *ctx);
/* The flowgraph has stopped (gr::block::stop()) and no gate / decoder call will come any more: whatever the device still
 * holds is decided as at the end of a stream, and the windows nobody fetched are accounted in READER_STATE as the decoder /
 * reader calls would have accounted them, so that rfid_print_results counts them.  Under a scheduler that calls the blocks
 * until none can move there is nothing left to do here; it is the safety net for one that gives up earlier. */
RFID_API int rfid_lookahead_drain(rfid_ctx *ctx);
/* Page-locked host memory (nullptr on failure): samples handed to rfid_mf_work (look-ahead) / rfid_stream_work from such
 * memory -- or from any memory the caller page-locked himself -- go to the device without the staging copy. */
RFID_API void *rfid_host_alloc(size_t bytes);
RFID_API void rfid_host_free(void *p);
/* |out[i]|^2 of the samples the last rfid_gate_work wrote (READER_STATE::magn_squared_samples, lib/gate_impl.cc:171,175,186),
 * computed on the device with the reference's expression re*re + im*im.  Look-ahead mode only (else *n = 0). */
RFID_API int rfid_gate_magn_squared(rfid_ctx *ctx, float *out, int cap, int *n);
/* reader_impl::general_work, state transitions only (gate_status / decoder_status /
 * n_queries_sent) without the transmit waveform (see rfid_reader_work_tx).  n_in = float items
 * available on the reader's input (16 after an RN16). */
RFID_API int rfid_reader_work(rfid_ctx *ctx, int n_in, int *n_consumed);
/* reader_impl::general_work complete: the same state transitions AND the transmit waveform the block writes
 * to its output for the state it was in (lib/reader_impl.cc:43-129 tables, :200-380; float samples 0/1 at
 * dac_rate, e.g. preamble + 22 PIE-coded Query bits + 1295 us of carrier).  in_bits: the reader's float input
 * (the 16 RN16 bits when the state is SEND_ACK), n_in its item count.  *n_written = floats written to out
 * (what general_work returns); RFID_ERR_CAPACITY (state untouched) when out_cap is too small --
 * rfid_reader_tx_max(dac_rate) floats always suffice. */
RFID_API int rfid_reader_work_tx(rfid_ctx *ctx, int dac_rate, const float *in_bits, int n_in, float *out, int out_cap,
                                 int *n_consumed, int *n_written);
RFID_API int rfid_reader_tx_max(int dac_rate);
RFID_API int rfid_get_state(const rfid_ctx *ctx, rfid_reader_state *out);
/* reader_impl::print_results text (lib/reader_impl.cc:173-192) */
RFID_API int rfid_print_results(const rfid_ctx *ctx, char *buf, int cap, int *len);


/* ---- (1b) streaming, whole chain per call: raw chunk in -> decoded windows out ------------------------------------ */
/* The drop-in calls above cross the host / device boundary three times per buffer.  This family runs
 * matched filter -> gate -> tag_decoder for one RX stream in ONE submission per chunk, with all block state carried
 * on the device (filter history, gate state incl. both rings, window alternation, READER_STATE), and overlaps the
 * host -> HBM copy of chunk k+1 with the processing of chunk k (two pinned staging buffers, a copy stream).
 * The gate scan of a chunk uses the long-stream front end (units scanned concurrently, accepted only when bit-identical
 * to the sequential scan), so a single stream is not limited to one SIMD.  Results equal rfid_batch_process over the
 * concatenated stream, window for window. */
typedef struct rfid_stream_window {
  int64_t start;          /* index (400 ksps domain, from the start of the stream) of the first gated sample */
  int32_t type;           /* RFID_DECODE_RN16 / RFID_DECODE_EPC */
  int32_t reserved_;
  float dc_re, dc_im;     /* gate_impl::dc_est at the opening */
} rfid_stream_window;

/* Allocates the staging (2 pinned host buffers + 2 device buffers of max_chunk_raw samples, plus room for the samples a
 * call leaves unprocessed) and resets the stream (fresh blocks).  Replaces any batch plan of the context. */
RFID_API int rfid_stream_begin(rfid_ctx *ctx, int64_t max_chunk_raw);
/* the two pinned staging buffers (idx 0 / 1): filling them directly makes the upload a true asynchronous DMA; any other
 * host pointer passed to rfid_stream_work is first copied into one of them */
RFID_API int rfid_stream_staging(rfid_ctx *ctx, int idx, rfid_cf32 **host, int64_t *cap);
/* Hands over n_raw NEW samples (their upload starts at once) and processes the chunk handed over by the PREVIOUS call
 * while that upload runs; flush != 0 also processes the new samples and everything held back (end of stream).
 * A call holds back what follows the last idle point of the gate's state machine (the reference's blocks do the same
 * through consume_each(): lib/gate_impl.cc:189-199, lib/tag_decoder_impl.cc:223,291); held-back samples are processed
 * with a later call.  windows / results (cap entries each, in stream order) receive the windows completed by this call,
 * *n_out their number (RFID_ERR_CAPACITY if cap is too small: nothing is lost, call again with larger arrays and
 * n_raw = 0).  READER_STATE (rfid_get_state, rfid_print_results) advances as in the per-block calls.
 * raw may be page-locked host memory -- one of the staging buffers or the caller's own (hipHostMalloc, hipHostRegister,
 * a pinned torch tensor): it is then uploaded by DMA straight from there and must stay untouched until the NEXT call
 * has returned -- or ordinary host memory, which is first copied into the staging buffer of the call (free on return). */
RFID_API int rfid_stream_work(rfid_ctx *ctx, const rfid_cf32 *raw, int64_t n_raw, int flush, rfid_stream_window *windows,
                              rfid_decode_result *results, int64_t cap, int64_t *n_out);
RFID_API int rfid_stream_end(rfid_ctx *ctx);

/* ---- (2) batched offline, device buffers ---------------------------------------------- */
/* Plans workspace for n_streams traces of up to max_raw samples each (2 Msps domain).
 * Allocates, in HBM: matched-filter output [n_streams][max_raw/5], window tables, results.
 * On failure (e.g. RFID_ERR_HIP: out of memory) the context is left WITHOUT a plan: every later
 * rfid_batch_* call returns RFID_ERR_STATE until a plan succeeds. */
RFID_API int rfid_batch_plan(rfid_ctx *ctx, int n_streams, int64_t max_raw);
/* Number of traces (rows of d_raw, entries of d_lens) the following rfid_batch_* calls process:
 * 1 <= n_streams <= the planned count (rfid_batch_plan sets it to the planned count).  Lets one plan
 * serve batches of different sizes; results and statistics cover exactly these rows. */
RFID_API int rfid_batch_set_streams(rfid_ctx *ctx, int n_streams);
/* d_raw: device pointer to [n_streams][raw_stride] rfid_cf32; d_lens: device int64[n_streams]
 * valid sample counts per trace, or NULL when every trace has n_raw samples.  All launches
 * go to the ctx stream (a non-blocking stream of its own, see rfid_ctx_stream) and return without
 * synchronising: work the caller queued on OTHER streams that produces d_raw / d_lens must be
 * complete (or ordered before the ctx stream with an event) when these functions are called. */
RFID_API int rfid_batch_mf(rfid_ctx *ctx, const void *d_raw, int64_t raw_stride, int64_t n_raw,
                           const void *d_lens);
RFID_API int rfid_batch_gate(rfid_ctx *ctx);
RFID_API int rfid_batch_decode(rfid_ctx *ctx, int want_scores);
RFID_API int rfid_batch_stats(rfid_ctx *ctx);
/* mf -> gate -> decode -> stats on the ctx stream (asynchronous).  Matched filter and gate run as ONE launch
 * (fused front end: the raw samples are read from HBM once); results are identical to calling the four stage
 * functions above one after the other.  The matched-filter output stays available (rfid_batch_get_mf). */
RFID_API int rfid_batch_process(rfid_ctx *ctx, const void *d_raw, int64_t raw_stride, int64_t n_raw,
                                const void *d_lens, int want_scores);
/* Long-stream front end of rfid_batch_process: with few, long traces the gate scan (a sequential recurrence per
 * trace) leaves the chip idle, so each trace is cut along time at idle points of the gate's state machine and all
 * pieces are processed at once from guessed start values; a pass is accepted only when every piece's run is exact or
 * provably covers its true start value, i.e. the result IS the sequential scan (otherwise the sequential scan runs:
 * it is enqueued behind the front end and skips itself when that succeeded -- a pass has no host synchronisation).
 * mode 0: never, 1 (default): when it is expected to be faster than the fused front end -- few traces, long enough
 * (a cost estimate from the batch size and the trace length, calibrated on the device when the context is created),
 * 2: whenever a trace can be cut.  The environment variable RFID_LONG_STREAM overrides the default.
 * rfid_batch_ls_report synchronises with the pass. */
RFID_API int rfid_batch_set_long_stream(rfid_ctx *ctx, int mode);
RFID_API int rfid_batch_ls_report(const rfid_ctx *ctx, rfid_ls_report *out);
/* The switches the environment can set (INTEGRATION.md section 13 lists them: RFID_LONG_STREAM, RFID_OVERLAP,
 * RFID_LS_CALIBRATE, RFID_LS_DEBUG, RFID_LA_PROFILE, RFID_LA_UPLOAD_KERNEL, RFID_LS_FRONT_LDS_KB and the test hooks
 * RFID_FRONT_UNFUSED / RFID_FRONT_CHUNKS / RFID_LS2_FSM_LANES_MIN / RFID_LS2_DC_ROUNDS).  The environment is read ONCE,
 * by rfid_ctx_create; these two change / read a value of a living context by its lower-case name without the RFID_
 * prefix ("overlap", "front_chunks", ...; LS2_FSM_LANES_MIN is "fsm_lanes_min", LS2_DC_ROUNDS "dc_rounds").  RFID_ERR_INVALID: unknown name or a
 * value outside the knob's range.  A change that concerns buffers (overlap) takes effect with the next rfid_batch_plan. */
RFID_API int rfid_ctx_set_knob(rfid_ctx *ctx, const char *name, int value);
RFID_API int rfid_ctx_get_knob(const rfid_ctx *ctx, const char *name, int *value);
RFID_API int rfid_batch_sync(rfid_ctx *ctx);
/* synchronises, then reports per-kernel times of the last pass */
RFID_API int rfid_batch_timing_get(rfid_ctx *ctx, rfid_batch_timing *out);
/* host copies of the results of the last pass (synchronising).
 * windows/results/scores: up to cap entries, ordered by (stream, seq); *n = total count. */
RFID_API int rfid_batch_get_stats(rfid_ctx *ctx, rfid_stream_stats *out, int n_streams);
RFID_API int rfid_batch_get_windows(rfid_ctx *ctx, rfid_window *windows, rfid_decode_result *results,
                                    rfid_scores *scores, int64_t cap, int64_t *n);
/* device-side views for callers that keep everything in HBM */
RFID_API int rfid_batch_device_ptrs(rfid_ctx *ctx, void **d_mf_out, int64_t *mf_stride,
                                    void **d_stats, void **d_flat_count);
/* copies the matched-filter output of trace `stream` to host (debug tap, the
 * file_sink_matched_filter of apps/reader.py:69) */
RFID_API int rfid_batch_get_mf(rfid_ctx *ctx, int stream, rfid_cf32 *out, int64_t cap, int64_t *n);
/* copies the gated, DC-removed samples of window `seq` of trace `stream` to host -- in[i] - dc_est over the window,
 * exactly what the gate block writes to its output (lib/gate_impl.cc:176,187; debug tap = the file_sink_gate of
 * apps/reader.py:70).  *n = window length (250 / 1370). */
RFID_API int rfid_batch_get_gated(rfid_ctx *ctx, int stream, int seq, rfid_cf32 *out, int64_t cap, int64_t *n);
/* the HIP stream the ctx launches on (hipStream_t as void*) */
RFID_API void *rfid_ctx_stream(rfid_ctx *ctx);

/* ---- (3) synthetic workloads (SURVEY.md section 8 d: noise replicas made on the device) ---------- */
/* d_out[s][i] = d_base[i] + sigma * (N(0,1) + j N(0,1)) for s in [0, n_streams), i in [0, n_raw): replica
 * first_replica + s of the base trace.  The noise is a pure function of (seed, replica index, i)
 * (Philox4x32-10 + Box-Muller), so a batch may be generated in pieces and on any number of GPUs.
 * Asynchronous on the ctx stream.  Not part of the receive path: the workload generator of bench.py
 * and of the HBM-capacity configurations. */
RFID_API int rfid_synth_replicas(rfid_ctx *ctx, const void *d_base, int64_t n_raw, void *d_out, int64_t out_stride,
                                 int n_streams, float sigma, uint64_t seed, int64_t first_replica);


/* ---- (3b) Gen2 trace synthesiser ------------------------------------------ */
/* One inventory slot of a synthetic receive trace: what the reader sends (reader_impl.cc:84-162: Query with
 * CRC-5 for the first slot of a round, QueryRep otherwise, then the ACK) and what the tags answer. */
#define RFID_SYNTH_MAX_RESP 8
#define RFID_SYNTH_MAX_TAGS 16
typedef struct rfid_synth_slot {
  uint8_t cmd;            /* 0 Query, 1 QueryRep */
  uint8_t q;              /* Q field of the Query (FIXED_Q) */
  uint8_t n_tags;         /* tags answering in this slot: 0 empty, 1 single, >1 collision (<= 8) */
  uint8_t has_epc;        /* the (single) responder backscatters its EPC frame after the ACK */
  uint8_t tag[RFID_SYNTH_MAX_RESP];   /* index of each responder's backscatter coefficient */
  uint16_t rn16[RFID_SYNTH_MAX_RESP]; /* the RN16 each responder backscatters, bit 15 sent first */
  uint16_t ack;           /* RN16 the reader's ACK carries, bit 15 first (the reference ACKs every slot) */
  int16_t reserved_;
  int32_t rn16_off_raw;   /* start of the RN16 replies, raw samples (2 Msps) after the end of the command: 500 = 250 us */
  int32_t epc_off_raw;    /* start of the EPC reply after the end of the ACK */
  uint32_t epc[4];        /* 128 EPC frame bits (PC + EPC + CRC-16), frame bit j at epc[j >> 5] bit (j & 31) */
} rfid_synth_slot;        /* 56 bytes */

typedef struct rfid_synth_gen2_params {
  float leak_re, leak_im;                 /* carrier leakage L */
  float h_re[RFID_SYNTH_MAX_TAGS];        /* backscatter coefficient of tag k */
  float h_im[RFID_SYNTH_MAX_TAGS];
  int32_t n_tags;
  int32_t tail_us;                        /* carrier after the last slot */
} rfid_synth_gen2_params;

/* Samples (2 Msps) of the trace these slots make: 4575 us carrier, then per slot command + 1295 us carrier +
 * ACK + 4575 us carrier (reader_impl.cc:69-70,218-224), then tail_us of carrier. */
RFID_API int rfid_synth_gen2_size(const rfid_synth_gen2_params *p, const rfid_synth_slot *slots, int64_t n_slots,
                                  int64_t *n_raw);
/* Writes the trace  x = L tx + sum_k h_k level_k tx + sigma (N(0,1) + j N(0,1))  into d_out (device, 16-byte
 * aligned, room for out_cap samples), generated on the device from the slot table (host pointer; copied).
 * The noise is that of rfid_synth_replicas for replica index `replica` (so sigma = 0 here followed by
 * rfid_synth_replicas gives the same bytes).  Asynchronous on the ctx stream; *n_raw = samples written. */
RFID_API int rfid_synth_gen2(rfid_ctx *ctx, const rfid_synth_gen2_params *p, const rfid_synth_slot *slots,
                             int64_t n_slots, void *d_out, int64_t out_cap, float sigma, uint64_t seed,
                             int64_t replica, int64_t *n_raw);

#ifdef __cplusplus
}
#endif
#endif /* RFID_MI355X_H */
