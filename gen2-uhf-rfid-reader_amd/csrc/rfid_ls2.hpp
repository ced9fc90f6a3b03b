// rfid_ls2.hpp -- long-stream front end: the gate scan of gate_impl.cc:127-196 over FEW LONG traces, cut along time
// into pieces that are processed concurrently -- and still the sequential scan, bit for bit.
//
// The scan carries two binary32 recurrences (avg_ampl, dc_est: in-order sums over the whole past) and a small state
// machine.  They are separated here, because they depend on each other in one direction only:
//
//   avg_ampl  (amplitudes only)  ->  threshold votes  ->  state machine  ->  which samples are "closed", where
//   windows open  ->  dc_est (sums over closed samples)  ->  dc_est at every window opening.
//
//   1. pieces        ls_cut_kernel / ls2_pieces_kernel: cut points near a regular grid, where the state machine idles
//   2. avg_ampl      ls2_avg_kernel: every piece at once from a GUESSED start value s (the mean of the amplitude ring)
//                    and from s + 1 ulp, leaving |x| and the addends (|x| - ring)/100 in HBM, the threshold votes of
//                    every step, the two end values and a MARGIN (below);  ls2_avg_chain_kernel: one prefix scan per
//                    trace over the pieces' (end - start) pairs gives every piece's true start; pieces whose run does
//                    not provably cover it are run again from it (rare), until none is left
//   3. state machine ls2_fsm_kernel over the votes (scalar unit only), from the idle state at every cut;
//                    ls2_fsm_chain_kernel checks that each piece's start state IS its predecessor's end state (and
//                    that the dc ring is what a cut assumes) -- a piece that fails is appended to its predecessor
//   4. dc_est        ls2_dc_kernel / ls2_dc_chain_kernel: as 2., over the closed samples, two components
//   5. windows       ls2_seq_kernel / ls2_assemble_kernel: per-trace window tables + the decoder's lists
//
// No host decision lies between the launches: every kernel looks at the control block (Ls2Ctl) in HBM and returns at
// once when there is nothing to do, so one pass is ENQUEUE-ONLY (rfid_ls2_enqueue.hpp), the sequential gate scan
// behind it included -- it runs only when this front end gave up (GateArgs::skip_if).
//
// Why a run from the wrong start value can still be exact.  Let a run start from s and the true start be s + D, D an
// EVEN multiple of u0 = ulp(s).  If every partial sum S_j of the run lies in a binade not above that of s, then D is an
// even multiple of ulp(S_j) as well; if moreover S_j and S_j + D always lie in the SAME binade, each rounding
// RN(S_j + D + d_j) = RN(S_j + d_j) + D: same grid, same distance to it, same parity at ties-to-even.  By induction
// the whole trajectory is shifted by exactly D -- end value, dc_est at every opening -- and, for avg_ampl, no
// threshold vote changes as long as no |x| lies between the two thresholds.  The run records the smallest distance
// of any partial sum from a power of two and of any |x| from its threshold, in units of u0: the MARGIN.  |D| + 4 <=
// margin proves the shift (a few ulps of slack: variant B's trajectory may run 2 ulps off variant A's) -- for the values
// inside the piece; the chain of pieces is integer arithmetic on binary32 bit patterns, so a piece that ends in another
// binade than it starts in only counts when run from its exact start.  Odd D: the same from the run that started at s + 1 ulp (variant B).  A piece whose margin
// does not cover its D (it passes close to a power of two: a few per cent of the pieces) is simply run again from
// its predicted start; the prediction is then verified, not assumed: the procedure ends when every piece's latest
// run is exact (D = 0) or proven.  Nothing else is assumed about the data -- pathological input (partial sums
// hovering at a binade edge, no idle points) costs rounds and finally the sequential scan, never correctness.
#pragma once

namespace rfidk {

constexpr int LS2_FINE = 4;           // pieces per point of the idle-cut grid
constexpr int LS2_WBUCKET = 320;      // two gate openings are at least RN16_WIN + T1_SAMPLES = 346 samples apart:
                                      // window records live in a per-trace table indexed by start / LS2_WBUCKET
static_assert(LS2_WBUCKET <= RN16_WIN + T1_SAMPLES, "one window per bucket");
constexpr int LS2_AVG_ROUNDS = 11;    // re-run rounds after the first pass, per recurrence (a round without work costs two empty launches; configs[2]
                                      // settles in 5 rounds on y-given pieces and in 8 on the fused first pass's, profiles/r05/ls2_rounds.txt)
constexpr int LS2_FSM_ROUNDS = 3;
constexpr int LS2_DC_ROUNDS = 10;     // dc_est rounds behind the first that a long pass enqueues (rfid_ls2_enqueue.hpp: why not fewer)
constexpr int LS2_MAXR = 12;
constexpr int LS2_WIDE_BELOW = 64;    // a piece whose margin is below this is re-run from six neighbouring start values at once
constexpr int LS2_WIDE_LO = -2, LS2_WIDE_HI = 3;
constexpr int LS2_CHAIN_THREADS = 512;      // (1 024-thread workgroups need a CU with sixteen free wave slots at once: beside the next pass's matched filter a chain launch of 25 us then took 400 - 650 us)
constexpr int LS2_CHAIN_GMAX = 64;     // workgroups per trace of a chain launch, at most

struct Ls2Piece { int pos0, len; };   // len 0: slot not in use

constexpr int LS2_DC_MAXR = 64;   // dc_est rounds a pass can enqueue at most (a round without work: five empty launches)
struct Ls2Ctl {   // control block in HBM, zeroed before every pass
  int fail;                   // != 0: the front end gave up (1 no cut, 2 avg_ampl, 3 state machine rounds exhausted, 5 the fused first
                              // pass met a stretch without a rest point, 6 / 7 dc_est: table overflow / a trace without a first unit -- cannot happen)
  int ok;                     // 1: the window tables were produced (set last; the fallback scan skips itself on it)
  int n_pieces;               // pieces the traces were cut into
  int n_heads;                // ... of them at idle cuts (or a trace's start): where the state-machine / dc_est passes can start
  int avg_count[LS2_MAXR];    // what chain round r left to do: pieces on the re-run list + exact ends put in (the chain has to be redone)
  int avg_list[LS2_MAXR];     //   ... the length of the list
  int fsm_count[LS2_MAXR];    // pieces appended to their predecessor in chain round r
  int avg_reruns, fsm_reruns, dc_reruns;   // totals (report)
  int avg_rounds, fsm_rounds, dc_rounds;   // launches that had work (report)
  int n_units;                // units (runs of pieces scanned in one go) at the end
  int n_windows;              // complete windows
  int wb_clash;               // two openings in one bucket (cannot happen; checked all the same -> fail)
  int n_dc_pieces;            // dc_est runs of the first round (= n_units since round 6)
  int dc_open_alloc;          // places handed out in Ls2Args::dcand
  int dc_finished;            // units the finishing walk took (the rounds were used up: the partial fallback)
  int dc_count[LS2_DC_MAXR + 1];   // units not settled after chain round r
  int fin_turns;              // the finishing walk: turns (trace 0's), ...
  int fin_reach[8];           //   ... how many units a turn settled: 0, 1, 2-3, 4-7, 8-15, 16-31, 32-63, more
  int fin_far[2];             //   ... a turn's first miss: inside / outside twice the windows' reach (the components' larger distance)
};
struct Ls2AvgRun {   // a piece's latest run
  float s, eA, eB;   // start used, end from it, end from s + 1 ulp
  int margin;        // (ulps of s)
  int wide;          // bit 0 (chain -> run): run wide next time;  bit 1 (run -> chain): ew[] holds the ends from s - 2, s - 1, s + 2,
                     // s + 3 ulps;  bit 2 (chain): aover[] holds this piece's function with one of those exact ends put in
  float ew[4];
  int pad_[3];
};
struct Ls2Fsm {   // per slot
  int head;       // the piece starts a unit (scanned from the idle state, or from the trace's start state)
  int unit;       // slot of the head of the unit the piece belongs to
  int gen;        // generation of the state-machine launch that last covered the piece
  int rerun;      // (head) scan again in the next round
  int nwin;       // (head) complete windows opened in the unit
  int nepc;       //   ... of them EPC windows
  int last_end;   // (head) end of the unit's last window (INT_MIN: none)
  int u1;         // (head) end of the unit: the next head's start, or the end of what is processed of the trace
  int st[6];      // (head) start state used: n_samples, signal_state, num_pulses, gate_open, n_to_ungate, wtype
  int en[6];      // (head) state after the unit
};
struct Ls2Win {   // one gate opening
  int start;
  int tag;        // type | complete << 1 | gen << 8;  0: empty
  int slot;       // where dc_est at the opening lies for each of the unit's 64 candidate starts: Ls2Args::dcand[slot][64]
  int unit;       // ... that unit's idle-grid slot
  int pad_[4];
};

struct Ls2Aff { int c0, c1; int tstar; int pad_; };   // a piece's function with the exact end for the start `tstar` put in (the entry of tstar's parity)

struct Ls2Args {
  const float2 *y; int64_t y_stride;
  const int64_t *lens; int64_t n_dec; int n_streams;
  int P, max_b;                 // nominal piece length; slots per trace
  int Pc, max_bc;               // idle cuts are searched on a coarser grid: its step (= LS2_FINE * P), its points per trace
  int *cut;                     // [n_streams][max_bc] from ls_cut_kernel (point 0 unused): where the gate idles
  int *cutf;                    // [n_streams][max_b] from ls_cut_kernel: where avg_ampl is at rest (100 carrier samples before)
  Ls2Piece *piece;              // [NS]
  int *nextv, *prevv;           // [NS] next / previous slot in use of the same trace, -1 none
  uint64_t *votes;              // [n_streams][vstride][2]: below, above; bit b of word w = sample 64 w + b (zeroed before a pass)
  uint64_t *closed;             // [n_streams][cstride]
  int *openinfo;                // [n_streams][cstride]: lane | type << 8 of the step's opening, 0xff none
  int64_t vstride;              // votes: one word per absolute 64-sample block of the trace
  int64_t cstride;              // closed / openinfo: a unit's steps start at its first sample: (start >> 6) + k + its idle-grid index
  Ls2AvgRun *arun; int *aT;     // [NS]; aT = true start of the piece (monotone integer image)
  int *alist;                   // [2][NS]: the re-run list of chain round r at (r & 1)
  Ls2Aff *aover;                // [NS] chain scratch: a piece's function with an exact candidate end put in
  Ls2Fsm *fsm;                  // [NS]
  Ls2Win *wb; int64_t wb_stride;   // [n_streams][wb_stride]
  // dc_est (section 4): per idle-grid slot t = trace * max_bc + J (NH of them); lane = candidate start
  int *dT;                      // [NH][2] the unit's start value (re, im; ord images) from the latest chain: true when settled, else predicted
  int *dcen;                    // [NH][2] centre of the unit's latest run: candidate j started at centre + j - 32 ulps
  int *dtab;                    // [NH][2][64] dc_est behind the unit for each candidate
  int *dstat;                   // [NH] bit 0 / 1: re / im settled, bit 2: the slot holds a unit, bit 3: its latest run does not cover the start predicted for it (zeroed before a pass)
  uint64_t *dexm;               // [NH][2] which entries of the unit's table are there (a unit's run leaves all 64)
  int *dfront;                  // [n_streams] the trace's first idle-grid slot whose unit is not settled (after a chain; INT_MAX: none)
  float2 *dq;                   // [n_streams][y_stride] the dc_est increments of every closed step, formed once for the finishing walk (ls2_dcb_incr_kernel)
  int *fscr, *fbar;             // the finishing walk's scratch [n_streams][2][waves per trace][LS2_FIN_REC] and its meeting counters [n_streams] (zeroed before a pass)
  int *dmar;                    // [NH][2] how far from its centre a start may lie for the unit's end to be a plain shift of candidate 32's / 33's (ulps; 0: nowhere)
  int *dwbase;                  // [NH] the unit's first place in dcand
  float2 *dcand; int dcand_cap; // [dcand_cap][64] dc_est at a gate opening for each candidate
  // the chain's levels: nodes of 64 children (level 1: blocks of units, level 2: groups of blocks); per node the centre of its
  // first child, its table on that window, which entries are exact, whether it holds anything, and its entry value from the walk
  int dcb_bias;                 // test hook: ulps added to the first round's centres (the ring means), as the rounding drift of a long trace would
  int dcb_n1, dcb_n2, dcb_top;  // nodes per trace of level 1 / 2; the level the walk runs over (2 when a trace has more than 64 blocks)
  int *n1cen, *n1tab, *n1val, *n1ent, *n1mar; uint64_t *n1exm;
  int *n2cen, *n2tab, *n2val, *n2ent, *n2mar; uint64_t *n2exm;
  int *seq0;                    // [NS][2] complete windows of the trace before the piece: all, EPC
  int *flat_base;               // [n_streams][2] the trace's first place in the decoder's RN16 / EPC list
  rfid_window *wtab; int wmax; int *wcount;
  rfid_window *flat; int *flat_count; int flat_cap;
  Ls2Ctl *ctl;
  const GateState *carry;       // optional [n_streams]: the state a trace starts from (streaming); nullptr: the fresh gate
  GateState *carry_out;         // optional [n_streams]: the state after the last processed piece
  int hold_last;                // streaming: a trace's last piece stays unprocessed; consumed[s] = its first sample
  int force;                    // run even when no trace could be cut
  int *consumed;                // [n_streams]
  int round;
  int avg_rounds, fsm_rounds, dc_rounds;   // re-run rounds enqueued behind the first pass of each stage (<= LS2_*_ROUNDS)
  // the chain kernels run on several workgroups per trace: block b publishes the total of its slots, then waits for the
  // blocks before it
  int chain_g;                  // workgroups per trace of this launch
  int stamp;                    // this launch's number within the pass (the flags are zeroed before a pass)
  int *cflag;                   // [n_streams][LS2_CHAIN_GMAX]
  int *cagg;                    // [n_streams][LS2_CHAIN_GMAX][4]
  // fused first pass (ls2_front_kernel): the matched filter runs inside the first avg_ampl pass, from the raw 2 Msps samples;
  // y is written for the passes behind it and the decoder.  The avg_ampl pieces then lie on block boundaries the pass finds
  // itself, the units of the state machine / dc_est passes on idle cuts found afterwards (ls2_idle_cut_kernel): two piece
  // tables -- `piece / nextv / prevv` for avg_ampl, `upiece / unextv / uprevv` for the units (heads at the idle cuts, on the idle
  // grid's slots; the slots between them empty, or cut at the avg_ampl pieces' starts where dc_fine wants pieces inside the
  // units); the unit kernels get a copy of the arguments with the second table in the first one's place
  int fused;                    // 0: y is given (cut searches on y);  1: fused first pass, avg_ampl view;  2: ... the units' view
  const float2 *raw; int64_t raw_stride; int raw_vec_ok;
  float2 *y_w;                  // [n_streams][y_stride], written (the same memory as y)
  uint64_t *lowm;               // [n_streams][cstride]: per 64-sample block of the trace, the samples that are not carrier (the
                                // memory of `closed`, which the state machine only writes later)
  Ls2Piece *upiece; int *unextv, *uprevv;   // [NS]
  int keep_flat_count;          // 1: ls2_clear_kernel leaves the decoder's list counters alone (the caller zeroes them)
  // the fused first pass's look-back (ls2_front_kernel): what the slots before have found out about avg_ampl, for the first guess
  uint64_t *lb_fn;              // [NS] the slot's piece as a function of its start value (c0 | (c1 ^ 2^31) << 32); 0: not there yet
  uint64_t *lb_end;             // [NS] avg_ampl behind the slot's piece (integer image, != 0) | drift << 32; 0: not known (yet)
  int *lb_water;                // [n_streams] 1 + the highest slot of the trace whose lb_end is known (0: none yet)
};

// ---- small helpers -----------------------------------------------------------------------------------------------
// Passes enqueued back to back: the launches behind the first pass of a pass (chains, re-runs, state machine, dc_est) share the
// device with the NEXT pass's first pass; most of them are short strings of dependent instructions in few waves, and a wave
// that shares its SIMD with four or five waves of the big kernel gets every fifth issue slot.  LS2_TAIL_PRIO_N > 0: they
// raise their waves' priority (s_setprio) -- measured in profiles/r05/ls2_overlap.txt
RFID_DEVICE void ls2_tail_prio() {}   // (s_setprio 2 / 3 for these launches: measured, no gain -- profiles/r05/ls2_fused_front.txt)
RFID_DEVICE int ls2_ord(float f) {   // monotone integer image of a binary32 value: distance = ulps
  const uint32_t u = wv::f2u(f);
  return (u & 0x80000000u) ? -(int)(u & 0x7fffffffu) : (int)u;
}
RFID_DEVICE float ls2_from_ord(int k) { return wv::u2f((k < 0) ? (0x80000000u | (uint32_t)(-k)) : (uint32_t)k); }
RFID_DEVICE int ls2_trace_len(const Ls2Args &a, int s) {
  int64_t n = a.n_dec;
  if (a.lens) { int64_t r = a.lens[s]; if (r < 0) r = 0; r /= DECIM; if (r < n) n = r; }
  return (int)n;
}
// distance of v from the nearest power of two in ulps of the value whose bit pattern is `sb` (rounded down); 0 when v
// lies in a higher binade, has the other sign, or either is zero / denormal / tiny / not finite
RFID_DEVICE int ls2_margin(float v, uint32_t sb) {
  const uint32_t b = wv::f2u(v);
  const int e0 = (int)((sb >> 23) & 0xffu), e = (int)((b >> 23) & 0xffu);
  const int m = (int)(b & 0x7fffffu);
  const int up = 0x800000 - m;
  const int dist = (m < up) ? m : up;
  const int sh = e0 - e;
  const bool bad = (((b ^ sb) >> 31) != 0u) || sh < 0 || sh > 23 || e == 0 || e0 == 255 || e0 < 25;
  return bad ? 0 : (dist >> sh);
}
// the same for the two variants of one scanned step (chain_add_scan2 returned true: every partial sum lies in the binade of
// its carry, so the shift is wave-uniform): min over both, 0 when a carry is of the wrong kind
RFID_DEVICE int ls2_margin_scanned(float vA, float vB, uint32_t cinA, uint32_t cinB, uint32_t sbA, uint32_t sbB, bool e0_ok) {
  const int shA = (int)((sbA >> 23) & 0xffu) - (int)((cinA >> 23) & 0xffu), shB = (int)((sbB >> 23) & 0xffu) - (int)((cinB >> 23) & 0xffu);
  const bool good = e0_ok && (((cinA ^ sbA) | (cinB ^ sbB)) >> 31) == 0u && shA >= 0 && shA <= 23 && shB >= 0 && shB <= 23 &&
                    (cinA & 0x7f800000u) != 0u && (cinB & 0x7f800000u) != 0u;
  if (!good) return 0;
  const int mA = (int)(wv::f2u(vA) & 0x7fffffu), mB = (int)(wv::f2u(vB) & 0x7fffffu);
  const int uA = 0x800000 - mA, uB = 0x800000 - mB;
  const int dA = ((mA < uA) ? mA : uA) >> shA, dB = ((mB < uB) ? mB : uB) >> shB;
  return (dA < dB) ? dA : dB;
}
RFID_DEVICE bool ls2_e0_ok(uint32_t sbA, uint32_t sbB) {
  return ((sbA >> 23) & 0xffu) >= 25u && ((sbA >> 23) & 0xffu) != 255u && ((sbB >> 23) & 0xffu) >= 25u && ((sbB >> 23) & 0xffu) != 255u;
}
// The margin of the steps that run in the START's binade (nearly all of them: a piece starts at the carrier's level) without
// per-step arithmetic: every lane keeps the least and the greatest mantissa its partial sums took (two instructions per
// variant and step); the distance to the binade's ends is formed once per piece -- min over the samples of
// min(m, 2^23 - m) = min(least m, 2^23 - greatest m), so the margin is the same number as ls2_margin_scanned's.
struct Ls2MantRange { int lo, hi; };
RFID_DEVICE void ls2_range_init(Ls2MantRange &r) { r.lo = 0x800000; r.hi = 0; }
RFID_DEVICE void ls2_range_add(Ls2MantRange &r, float v) {
  const int m = (int)(wv::f2u(v) & 0x7fffffu);
  r.lo = (m < r.lo) ? m : r.lo;
  r.hi = (m > r.hi) ? m : r.hi;
}
RFID_DEVICE int ls2_range_margin(const Ls2MantRange &a, const Ls2MantRange &b) {   // (lane-local; no step recorded: 2^23)
  const int da = (a.lo < 0x800000 - a.hi) ? a.lo : (0x800000 - a.hi), db = (b.lo < 0x800000 - b.hi) ? b.lo : (0x800000 - b.hi);
  return (da < db) ? da : db;
}
// a scanned step whose two carries lie in the binade (and have the sign) of their variants' starts: ls2_margin_scanned's
// `good` holds and both shifts are 0
RFID_DEVICE bool ls2_in_start_binade(uint32_t cinA, uint32_t cinB, uint32_t sbA, uint32_t sbB, bool e0_ok) {
  return e0_ok && (((cinA ^ sbA) | (cinB ^ sbB)) & 0xff800000u) == 0u;
}
RFID_DEVICE int ls2_wave_min(int v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const int o = wv::shfl_xor(v, off); v = (o < v) ? o : v; }
  return v;
}

// The in-order sums of one step's addends from TWO carries in one go (variants A and B of a run): everything that
// depends on the addends and the binade only is shared (see chain_add_scan), the parities at ties and the prefix sums
// are per carry.  false: not both in one binade, or a partial sum left it -- the caller takes chain_add_auto twice.
RFID_DEVICE bool chain_add_scan2_dual(float ca, float cb2, float x, int lane, float &oa, float &ob) {
  const uint32_t ba = wv::f2u(ca), bb = wv::f2u(cb2);
  if (((ba ^ bb) & 0xff800000u) != 0u) return false;
  const uint32_t e_b = (ba >> 23) & 0xffu;
  const uint32_t sign = ba & 0x80000000u;
  const float scale = wv::u2f(((277u - e_b) & 0xffu) << 23);
  const float t = sign ? -(x * scale) : (x * scale);
  const float r = wv::rint_f(t);
  const float frac = t - r;
  const bool bad_t = !(__builtin_fabsf(t) < 4194304.0f);
  const bool tie = !bad_t && __builtin_fabsf(frac) == 0.5f;
  const int R = wv::f2i(bad_t ? 0.0f : r);
  int Ra = R, Rb = R;
  const uint64_t tiemask = wv::ballot(tie);
  if (tiemask != 0ull) {
    const int I = R - ((frac < 0.0f) ? 1 : 0);
    const uint64_t rodd = wv::ballot(!tie && (R & 1));
    const uint64_t iodd = wv::ballot(tie && (I & 1));
    const uint64_t Q = prefix_xor64(rodd);
    const uint64_t gen = tiemask & Q, X = gen | ~tiemask;
    const uint64_t suma = X + gen + (uint64_t)(ba & 1u), sumb = X + gen + (uint64_t)(bb & 1u);
    const uint64_t upa = (((Q << 1) ^ X ^ gen ^ suma) ^ iodd) & tiemask;
    const uint64_t upb = (((Q << 1) ^ X ^ gen ^ sumb) ^ iodd) & tiemask;
    Ra = tie ? (I + (int)((upa >> lane) & 1ull)) : R;
    Rb = tie ? (I + (int)((upb >> lane) & 1ull)) : R;
  }
  const uint32_t maga = (ba & 0x7fffffffu) + (uint32_t)wv::scan_add(Ra);
  const uint32_t magb = (bb & 0x7fffffffu) + (uint32_t)wv::scan_add(Rb);
  const bool bad_s = ((maga ^ ba) & 0x7f800000u) != 0u || (maga & 0x007fffffu) == 0u ||
                     ((magb ^ bb) & 0x7f800000u) != 0u || (magb & 0x007fffffu) == 0u;
  const bool bad_c = e_b < 23u || e_b > 254u;
  oa = wv::u2f(maga | sign);
  ob = wv::u2f(magb | sign);
  return wv::ballot(bad_t || bad_s || bad_c) == 0ull;
}
// The same at the price of ONE scan.  The two carries differ by dm units of the binade's grid (variant B starts one ulp above
// A); the integers R_j that the addends contribute do not depend on the carry except at ties, where only its parity
// counts -- so while dm is EVEN both variants take the same R_j and B's partial sums are A's + dm, lane for lane; and a
// step without a tie is the same integers whatever dm is.  Only a step that has a tie while dm is odd needs both scans --
// and it leaves dm even (A takes I + a, B takes I + 1 - a: dm becomes 2 - 2a), after which it stays even as long as the
// sums stay in one binade: a piece pays the second scan once.  (B in the binade with a non-zero mantissa: A is, and A's
// mantissa + dm lies in [1, 2^23 - 1].)
RFID_DEVICE bool chain_add_scan2(float ca, float cb2, float x, int lane, float &oa, float &ob) {
  const uint32_t ba = wv::f2u(ca), bb = wv::f2u(cb2);
  if (((ba ^ bb) & 0xff800000u) != 0u) return false;
  const int dm = (int)(bb & 0x7fffffffu) - (int)(ba & 0x7fffffffu);   // (wave-uniform)
  const uint32_t e_b = (ba >> 23) & 0xffu;
  const uint32_t sign = ba & 0x80000000u;
  const float scale = wv::u2f(((277u - e_b) & 0xffu) << 23);
  const float t = sign ? -(x * scale) : (x * scale);
  const float r = wv::rint_f(t);
  const float frac = t - r;
  const bool bad_t = !(__builtin_fabsf(t) < 4194304.0f);
  const bool half = __builtin_fabsf(frac) == 0.5f;
  const bool tie = !bad_t && half;
  int R = wv::f2i(bad_t ? 0.0f : r);
  const uint64_t badmask_t = wv::ballot(bad_t);
  const uint64_t tiemask = wv::ballot(half) & ~badmask_t;
  if (tiemask != 0ull) {
    if (dm & 1) return chain_add_scan2_dual(ca, cb2, x, lane, oa, ob);
    // (chain_add_scan's ties, from A's parity -- B's is the same)
    const int I = R - ((frac < 0.0f) ? 1 : 0);
    const uint64_t rodd = wv::ballot((R & 1) != 0) & ~tiemask;
    const uint64_t iodd = wv::ballot((I & 1) != 0) & tiemask;
    const uint64_t Q = prefix_xor64(rodd);
    const uint64_t gen = tiemask & Q, X = gen | ~tiemask;
    const uint64_t sum = X + gen + (uint64_t)(ba & 1u);
    const uint64_t W = X ^ gen ^ sum;
    const uint64_t par = (Q << 1) ^ W;
    const uint64_t up = (par ^ iodd) & tiemask;
    R = tie ? (I + (int)((up >> lane) & 1ull)) : R;
  }
  const uint32_t maga = (ba & 0x7fffffffu) + (uint32_t)wv::scan_add(R);
  const uint32_t m = maga & 0x007fffffu;
  const uint64_t badmask_s = wv::ballot(((maga ^ ba) & 0x7f800000u) != 0u) | wv::ballot(m == 0u) |
                             wv::ballot((uint32_t)((int)m + dm - 1) >= 0x007fffffu);
  const bool bad_c = e_b < 23u || e_b > 254u;
  oa = wv::u2f(maga | sign);
  ob = wv::u2f((uint32_t)((int)maga + dm) | sign);
  return (badmask_t | badmask_s) == 0ull && !bad_c;
}
RFID_DEVICE bool chain_add_auto2(float ca, float cb2, float x, int lane, float &oa, float &ob) {   // true: the shared scan applied
  if (__builtin_expect(chain_add_scan2(ca, cb2, x, lane, oa, ob), 1)) return true;
  oa = chain_add_auto(ca, x, lane);
  ob = chain_add_auto(cb2, x, lane);
  return false;
}

// ---- 0. what a pass expects zeroed: the control block, the chain flags, consumed[], the votes, the window buckets, the
//         counters of the decoder's lists -- one launch instead of six fills ---------------------------------------------
static_assert(sizeof(Ls2Ctl) == 512, "Ls2Ctl and consumed[] are fetched with one copy (rfid_ls2_enqueue.hpp lays them out back to back)");
RFID_KERNEL(256) void ls2_clear_kernel(Ls2Args a) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
  const int64_t B = a.n_streams;
  auto zero4 = [&](void *p, int64_t words) { uint32_t *q = (uint32_t *)p; for (int64_t i = tid; i < words; i += nth) q[i] = 0u; };
  auto zero8 = [&](void *p, int64_t words) { uint64_t *q = (uint64_t *)p; for (int64_t i = tid; i < words; i += nth) q[i] = 0ull; };
  zero4(a.ctl, (int64_t)(sizeof(Ls2Ctl) / 4));
  zero4(a.cflag, B * LS2_CHAIN_GMAX);
  zero4(a.consumed, B);
  zero4(a.dstat, B * a.max_bc);
  zero4(a.fbar, B);
  if (a.fused) { zero8(a.lb_fn, B * a.max_b); zero8(a.lb_end, B * a.max_b); zero4(a.lb_water, B); }
  if (!a.keep_flat_count) zero4(a.flat_count, 2);   // (a first pass that runs beside the pass before: its decoder still reads them)
  zero8(a.votes, 2 * B * a.vstride);
  zero8(a.wb, (int64_t)(sizeof(Ls2Win) / 8) * B * a.wb_stride);
}

// ---- 1. pieces -----------------------------------------------------------------------------------------------------
// Where a trace is cut.  avg_ampl only needs a place where it is at rest (100 carrier samples before: ls_cut_kernel with
// a short quiet zone, near every point of a grid of step P) -- short pieces keep every pass short.  The state machine
// and dc_est can only start where the gate idles: those cuts come from ls_cut_kernel with the long quiet zone, searched
// near every LS2_FINE-th grid point, and take the place of that grid point; the pieces that start there are "heads", the
// others are scanned through from the head before them.  One thread per slot (trace s, grid point j): the piece from the slot's boundary to the next one in use.
constexpr int LS2_REST = WIN_LEN + 28;   // carrier samples before a piece boundary: the amplitude ring holds carrier only, avg_ampl
                                         // sits at the carrier's level (in its binade) -- away from such points it is on the move
                                         // across binades and a run from a shifted start proves nothing
RFID_DEVICE int ls2_boundary(const Ls2Args &a, const int *cut, const int *cutf, int n, int j, bool &head) {   // -1: slot not in use
  const int J = j / LS2_FINE, q = j - J * LS2_FINE;
  const int c = (J > 0) ? cut[J] : 0;
  const bool found = (J == 0) ? (n > 0) : (c > 0 && c < n);
  head = false;
  if (q == 0 && found) { head = true; return c; }
  const int p = (j > 0 && cutf) ? cutf[j] : -1;
  if (p <= 0 || p >= n) return -1;
  if (found && p < c + a.P / 2) return -1;   // too close behind (or before) the idle cut that stands for this stretch's grid point
  return p;
}
RFID_KERNEL(256) void ls2_pieces_kernel(Ls2Args a) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  const int NS = a.n_streams * a.max_b;
  const int lane = wv::lane_id();
  bool used = false, is_head = false;
  if (i < NS) {
    const int s = i / a.max_b, j = i - s * a.max_b;
    const int n = ls2_trace_len(a, s);
    const int *cut = a.cut + (int64_t)s * a.max_bc;
    const int *cutf = a.cutf ? a.cutf + (int64_t)s * a.max_b : nullptr;
    bool head = false;
    const int p = ls2_boundary(a, cut, cutf, n, j, head);
    Ls2Piece pc; pc.pos0 = 0; pc.len = 0;
    int nx = -1, pv = -1;
    if (p >= 0) {
      int end = n;
      bool h2;
      for (int j2 = j + 1; j2 < a.max_b; ++j2) { const int q = ls2_boundary(a, cut, cutf, n, j2, h2); if (q >= 0) { end = q; nx = s * a.max_b + j2; break; } }
      for (int j2 = j - 1; j2 >= 0; --j2) if (ls2_boundary(a, cut, cutf, n, j2, h2) >= 0) { pv = s * a.max_b + j2; break; }
      pc.pos0 = p; pc.len = end - p;
      if (a.hold_last) {
        // streaming: everything from the trace's last idle cut on waits for more samples
        int last = 0;
        for (int J = a.max_bc - 1; J >= 1; --J) { const int c = cut[J]; if (c > 0 && c < n) { last = c; break; } }
        if (p >= last) pc.len = 0;
        if (j == 0 && a.consumed) a.consumed[s] = last;
      }
    } else if (j == 0 && a.consumed) {
      a.consumed[s] = 0;
    }
    used = pc.len > 0;
    is_head = used && head;
    a.piece[i] = pc;
    a.nextv[i] = nx;
    a.prevv[i] = pv;
    Ls2Fsm f;
    f.head = is_head ? 1 : 0; f.unit = i; f.gen = -1; f.rerun = 0; f.nwin = 0; f.nepc = 0; f.last_end = -2147483647 - 1; f.u1 = 0;
    for (int k = 0; k < 6; ++k) { f.st[k] = 0; f.en[k] = 0; }
    a.fsm[i] = f;
    a.seq0[2 * i] = 0; a.seq0[2 * i + 1] = 0;
  }
  const uint64_t m = wv::ballot(used), mh = wv::ballot(is_head);
  if (lane == 0 && m && a.fused != 2) wv::atomic_add(&a.ctl->n_pieces, wv::popc64(m));   // (the units' view: the avg_ampl pieces were counted)
  if (lane == 0 && mh) wv::atomic_add(&a.ctl->n_heads, wv::popc64(mh));
}

// nothing to gain (no trace has an idle cut) -> the sequential scan.  Known once ls2_pieces_kernel is through: the first
// launch behind it looks (every wave for itself) and leaves the verdict in Ls2Ctl::fail for the launches that follow
RFID_DEVICE bool ls2_nothing_to_gain(const Ls2Args &a) {
  const int nh = wv::uniform(a.ctl->n_heads);
  return nh <= 0 || (!a.force && nh <= a.n_streams);
}

// ---- 2. avg_ampl ---------------------------------------------------------------------------------------------------
// One wave per piece.  FIRST: |x| from the samples (and into HBM), start value guessed; later rounds: the pieces of the
// re-run list from their predicted start, |x| from HBM; the addends (|x| - ring)/100 are formed from |x| both times (until
// round 5 they were kept in HBM too: 4 B per sample written and read again for a subtraction and a three-instruction
// division).  Same arithmetic as the producer / consumer waves of the gate scan, value for value (gate_impl.cc:130-136).
template <bool FIRST>
RFID_DEVICE void ls2_avg_piece(const Ls2Args &a, const int i, const int lane) {
  const int s = i / a.max_b, j = i - s * a.max_b;
  const Ls2Piece pc = a.piece[i];
  const int p0 = wv::uniform(pc.pos0), n = wv::uniform(pc.len);
  if (n <= 0) return;
  // The piece is walked in the trace's own 64-sample blocks (lane L of step k = sample 64 (w0 + k) + L), so that the votes
  // of a sample land at the same bit whatever piece it belongs to; the first and last block are shared with the
  // neighbours (their lanes outside the piece add +0 to the sums and cast no vote).
  const int p1 = p0 + n, w0 = p0 >> 6, nsteps = ((p1 + 63) >> 6) - w0;
  const int n_total = ls2_trace_len(a, s);
  const int last_idx = n_total - 1;   // (n_total >= p1 > 0)
  const int64_t row = (int64_t)s * a.y_stride;
  const float2 *yr = a.y + row;
  uint64_t *votes = a.votes + 2 * ((int64_t)s * a.vstride + w0);
  const int base = 64 * w0;
  float sA;
  float a2 = 0.0f, a1 = 0.0f;   // amplitudes of the 128 samples before the step (the ring of gate_impl.cc:131 holds the last 100)
  // |x| of sample idx < base: from the samples (formed again wherever it is needed -- a record of it would be 4 bytes written
  // per sample for the few pieces that run twice), or (before the start of the trace) the carried ring / the fresh gate's zeros
  auto hist = [&](int idx) -> float {
    if (idx >= 0) { const float2 v = yr[idx]; return wv::hypot_f(v.x, v.y); }
    if (!a.carry || idx < -WIN_LEN) return 0.0f;
    const GateState *cs = a.carry + s;   // sample -k (k = 1..100): win[(win_index - k) mod 100] (win_index = the oldest = next written)
    return cs->win[(cs->win_index + idx + 2 * WIN_LEN) % WIN_LEN];
  };
  a2 = hist(base - 128 + lane);
  a1 = hist(base - 64 + lane);
  if (FIRST) {
    if (j == 0) {
      sA = a.carry ? wv::uniform(a.carry[s].avg_ampl) : 0.0f;   // the exact start of the trace
    } else {
      // first guess: the mean of the ring at the piece's first sample (what avg_ampl is up to its rounding drift)
      float part = hist(p0 - WIN_LEN + lane) + ((lane < WIN_LEN - 64) ? hist(p0 - WIN_LEN + 64 + lane) : 0.0f);
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) part += wv::shfl_xor(part, off);
      sA = wv::uniform(part) / WIN_LEN_F;
    }
  } else {
    sA = wv::uniform(a.arun[i].s);
  }
  // a piece that passes so close to a power of two that hardly any shift is provable is run from six neighbouring start
  // values at once (s - 2 .. s + 3 ulps): their ends are exact, not predicted, whatever the chain turns up among them
  const bool wide = !FIRST && (wv::uniform(a.arun[i].wide) & 1) != 0;
  float wv0 = 0.0f, wv1 = 0.0f, wv2 = 0.0f, wv3 = 0.0f;   // carries of s - 2, s - 1, s + 2, s + 3
  if (wide) {
    const int os = ls2_ord(sA);
    wv0 = ls2_from_ord(os - 2); wv1 = ls2_from_ord(os - 1); wv2 = ls2_from_ord(os + 2); wv3 = ls2_from_ord(os + 3);
  }
  const uint32_t sbA = wv::f2u(sA);
  const float sB = ls2_from_ord(ls2_ord(sA) + 1);
  const uint32_t sbB = wv::f2u(sB);
  float avA = sA, avB = sB;
  int marg = ls2_margin(sA, sbA);
  const bool e0_ok = ls2_e0_ok(sbA, sbB);
  Ls2MantRange rgA, rgB;              // steps in the start's binade: the mantissas' range and the least |x|-to-threshold distance,
  ls2_range_init(rgA); ls2_range_init(rgB);   // lane by lane; the margin they stand for is formed behind the loop
  int min_dv = 0x7fffffff;
  constexpr int AHEAD = 4;
  float2 ybuf[AHEAD];
#pragma unroll
  for (int u = 0; u < AHEAD; ++u) {
    const int idx = base + 64 * u + lane;
    // (loads past the end of the trace are clamped, not predicated: their lanes are not `valid` and add +0; an unconditional
    // load lands in the register the step reads it from -- no copies at the loop's end that would wait for all of them;
    // the step's lanes outside the piece too: they are the ring of the samples behind them)
    ybuf[u] = yr[(idx < n_total) ? idx : last_idx];
  }
  uint64_t my_lt = 0, my_gt = 0;   // lane (k & 63) keeps the votes of step k until 64 steps are stored together
  // one step.  The groups of AHEAD steps that are complete run without a condition around a step: every buffer is read and
  // loaded again in place, and the loop's only waits are counted ones for the oldest load.  (With the steps guarded one by one
  // the compiler kept the fresh loads in other registers and copied them at the loop's end -- behind a wait for ALL of them:
  // the read-ahead was one step deep, not four.)  The last, incomplete group reads what is left of the buffers.
  auto step = [&](const int k, float2 &yb, const bool reload) {
      {
        const int idx = base + 64 * k + lane;
        const bool valid = idx >= p0 && idx < p1;
        float amp, d;
        {
          const float2 v = yb;
          amp = wv::hypot_f(v.x, v.y);
          // (the buffer is loaded again once its value is used up: the load lands in the same register -- issued before
          // that, the compiler kept it elsewhere and copied it at the loop's end behind a wait for all loads in flight)
          if (reload) { const int nx = idx + 64 * AHEAD; yb = yr[(nx < n_total) ? nx : last_idx]; }
        }
        {
          // sample i - 100: lanes 0..35 take it from two steps back (lane + 28), lanes 36..63 from the previous step (lane - 36)
          const float o2 = wv::shfl(a2, (lane + 28) & 63), o1 = wv::shfl(a1, (lane - 36) & 63);
          const float old = (lane < 36) ? o2 : o1;
          const float nd = valid ? (amp - old) : 0.0f;
          d = div_const<WIN_LEN>(nd);
          a2 = a1; a1 = amp;
        }
        float vA, vB;
        const uint32_t cinA = wv::f2u(avA), cinB = wv::f2u(avB);
        const bool scanned = chain_add_auto2(avA, avB, d, lane, vA, vB);
        avA = wv::readlane(vA, 63);
        avB = wv::readlane(vB, 63);
        if (wide) {
          float t0, t1, t2, t3;
          chain_add_auto2(wv0, wv1, d, lane, t0, t1);
          chain_add_auto2(wv2, wv3, d, lane, t2, t3);
          wv0 = wv::readlane(t0, 63); wv1 = wv::readlane(t1, 63); wv2 = wv::readlane(t2, 63); wv3 = wv::readlane(t3, 63);
        }
        const float thresh = vA * THRESH_FRACTION;
        const uint64_t below = wv::ballot(valid && amp < thresh);
        const uint64_t above = wv::ballot(valid && amp > thresh);
        // margin: the partial sums of both variants against the powers of two, |x| against the threshold
        {
          const uint32_t tb = wv::f2u(thresh), ab = wv::f2u(amp);
          int dv = (int)ab - (int)tb;
          dv = (dv < 0) ? -dv : dv;
          if (scanned && ls2_in_start_binade(cinA, cinB, sbA, sbB, e0_ok)) {
            ls2_range_add(rgA, vA); ls2_range_add(rgB, vB);
            const int dvv = valid ? dv : 0x7fffffff;
            min_dv = (dvv < min_dv) ? dvv : min_dv;
          } else {
            int mm;
            if (scanned) {
              // every partial sum of the step lies in its carry's binade (chain_add_scan): one shift for all lanes
              mm = ls2_margin_scanned(vA, vB, cinA, cinB, sbA, sbB, e0_ok);
              if (mm > 0) {
                const int shA = (int)((sbA >> 23) & 0xffu) - (int)((cinA >> 23) & 0xffu);
                const int mV = valid ? ((dv - 3) >> (1 + shA)) : 0x7fffffff;
                mm = (mV < mm) ? mV : mm;
              }
            } else {
              const int mA = ls2_margin(vA, sbA), mB = ls2_margin(vB, sbB);
              const int sh = (int)((sbA >> 23) & 0xffu) - (int)((wv::f2u(vA) >> 23) & 0xffu);
              int mV = ((tb >> 31) != 0u || sh < 0 || sh > 23) ? 0 : (((dv - 3) >> 1) >> sh);
              mV = valid ? mV : 0x7fffffff;
              mm = (mA < mB) ? mA : mB;
              mm = (mV < mm) ? mV : mm;
            }
            marg = (mm < marg) ? mm : marg;
          }
        }
        // the votes: whole blocks are stored 64 at a time; the two blocks shared with the neighbouring pieces get this
        // piece's bits put in (a re-run replaces its own bits only)
        const bool shared = k == 0 || k == nsteps - 1;
        if (shared) {
          const uint64_t mine = wv::ballot(valid);
          if (lane == 0) {
            if (mine == ~0ull) { votes[2 * k] = below; votes[2 * k + 1] = above; }
            else {
              wv::atomic_and64(&votes[2 * k], ~mine); wv::atomic_or64(&votes[2 * k], below);
              wv::atomic_and64(&votes[2 * k + 1], ~mine); wv::atomic_or64(&votes[2 * k + 1], above);
            }
          }
        }
        if (lane == (k & 63)) { my_lt = below; my_gt = above; }
        if ((k & 63) == 63 || k == nsteps - 1) {
          const int k0 = k & ~63, kl = k0 + lane;
          if (kl <= k && kl != 0 && kl != nsteps - 1) { votes[2 * kl] = my_lt; votes[2 * kl + 1] = my_gt; }
        }
      }
  };
  int kb = 0;
  for (; kb + AHEAD <= nsteps; kb += AHEAD) {
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) step(kb + u, ybuf[u], true);
  }
#pragma unroll
  for (int u = 0; u < AHEAD - 1; ++u)
    if (kb + u < nsteps) step(kb + u, ybuf[u], false);
  {
    const int mb = ls2_range_margin(rgA, rgB);
    const int mv = (min_dv == 0x7fffffff) ? min_dv : ((min_dv - 3) >> 1);
    const int mf = (mv < mb) ? mv : mb;
    marg = (mf < marg) ? mf : marg;
  }
  marg = ls2_wave_min(marg);
  // the chain works on the integer image of binary32, where a shift by D ulps of the start is a shift by D at the end only
  // if both lie in one binade: else nothing but the exact start counts
  if ((((wv::f2u(avA) ^ sbA) | (wv::f2u(avB) ^ sbB)) & 0xff800000u) != 0u) marg = 0;
  if (lane == 0) {
    Ls2AvgRun r;
    r.s = sA; r.eA = avA; r.eB = avB; r.margin = marg;
    r.wide = wide ? 2 : 0;
    r.ew[0] = wv0; r.ew[1] = wv1; r.ew[2] = wv2; r.ew[3] = wv3;
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
    a.arun[i] = r;
  }
}

RFID_KERNEL(64) void ls2_avg_first_kernel(Ls2Args a) {
  const int lane = wv::lane_id();
  if (ls2_nothing_to_gain(a)) {
    if (blockIdx.x == 0 && lane == 0) a.ctl->fail = 1;
    return;
  }
  const int NS = a.n_streams * a.max_b;
  // workgroup b runs on XCD b mod 8: each XCD takes one contiguous eighth of the slots (grid = 8 * ceil(NS / 8)).  Slots in
  // their own order put every LS2_FINE-th slot -- the heads, whose pieces are the long ones (a rest point too close behind an
  // idle cut is dropped) -- on two of the eight XCDs, which then finish last; and neighbouring pieces share their
  // first / last vote block and the 128 samples of history, which now sit in one L2.
  const int per = (NS + 7) >> 3;
  const int b = (int)blockIdx.x, i = (b & 7) * per + (b >> 3);
  if ((b >> 3) < per && i < NS) ls2_avg_piece<true>(a, i, lane);
}
RFID_KERNEL(64) void ls2_avg_rerun_kernel(Ls2Args a) {
  ls2_tail_prio();
  if (wv::uniform(a.ctl->fail) != 0) return;
  const int NS = a.n_streams * a.max_b;
  const int cnt = wv::uniform(a.ctl->avg_list[a.round - 1]);
  const int lane = wv::lane_id();
  const int *list = a.alist + (int64_t)((a.round - 1) & 1) * NS;
  for (int r = (int)blockIdx.x; r < cnt; r += (int)gridDim.x) ls2_avg_piece<false>(a, wv::uniform(list[r]), lane);   // (grid = NS: one each)
}

// A piece's latest run as a function "true start -> true end", on the monotone integer image of binary32: T -> T + c[q],
// q = parity of T (the run from s serves the starts s + even, the run from s + 1 ulp the starts s + odd).  Functions
// of this form compose to the same form, so a trace's chain of pieces is ONE prefix scan.  (32-bit wrap-around
// arithmetic: the true values fit, so sums modulo 2^32 are the true sums.)
struct Ls2A32 { int c0, c1; };
RFID_DEVICE Ls2A32 ls2_comp32(const Ls2A32 f, const Ls2A32 g) {   // f first, then g
  Ls2A32 r;
  r.c0 = (int)((uint32_t)f.c0 + (uint32_t)((f.c0 & 1) ? g.c1 : g.c0));
  r.c1 = (int)((uint32_t)f.c1 + (uint32_t)(((1 + f.c1) & 1) ? g.c1 : g.c0));
  return r;
}
RFID_DEVICE Ls2A32 ls2_elem32(float s, float eA, float eB) {
  const uint32_t os = (uint32_t)ls2_ord(s);
  const int a0 = (int)((uint32_t)ls2_ord(eA) - os), a1 = (int)((uint32_t)ls2_ord(eB) - 1u - os);
  Ls2A32 r;
  if (os & 1u) { r.c0 = a1; r.c1 = a0; } else { r.c0 = a0; r.c1 = a1; }   // T even-distant from s -> a0, odd-distant -> a1
  return r;
}
RFID_DEVICE int ls2_apply32(const Ls2A32 f, int T) { return (int)((uint32_t)T + (uint32_t)((T & 1) ? f.c1 : f.c0)); }
// inclusive scan over the 64 lanes in lane order (lane L: the composition of lanes 0..L): four steps within the 16-lane
// rows, two across them (the identity (0, 0) where a lane has no source)
RFID_DEVICE Ls2A32 ls2_wave_incl(Ls2A32 v, int lane) {
  (void)lane;
  Ls2A32 o;
  o.c0 = wv::dpp_row_shr<1>(v.c0, 0); o.c1 = wv::dpp_row_shr<1>(v.c1, 0); v = ls2_comp32(o, v);
  o.c0 = wv::dpp_row_shr<2>(v.c0, 0); o.c1 = wv::dpp_row_shr<2>(v.c1, 0); v = ls2_comp32(o, v);
  o.c0 = wv::dpp_row_shr<4>(v.c0, 0); o.c1 = wv::dpp_row_shr<4>(v.c1, 0); v = ls2_comp32(o, v);
  o.c0 = wv::dpp_row_shr<8>(v.c0, 0); o.c1 = wv::dpp_row_shr<8>(v.c1, 0); v = ls2_comp32(o, v);
  o.c0 = wv::dpp_row_bcast15(v.c0, 0); o.c1 = wv::dpp_row_bcast15(v.c1, 0); v = ls2_comp32(o, v);
  o.c0 = wv::dpp_row_bcast31(v.c0, 0); o.c1 = wv::dpp_row_bcast31(v.c1, 0); v = ls2_comp32(o, v);
  return v;
}
RFID_DEVICE Ls2A32 ls2_wave_excl(const Ls2A32 incl, int lane) {
  (void)lane;
  Ls2A32 e;
  e.c0 = wv::dpp_wave_shr1(incl.c0, 0); e.c1 = wv::dpp_wave_shr1(incl.c1, 0);
  return e;
}
// ---- 2a. the FUSED first pass (round 5): matched filter + piece boundaries + avg_ampl in ONE sweep over the raw samples ----
// The matched filter is bound by the HBM (8 B in per raw sample), the first avg_ampl pass by its instruction streams; as two
// launches they ran one after the other (3.6 + 0.35 (cut searches over y) + 2.0 ms for configs[2]) and y was written, then read
// three times.  Here every wave filters the samples of its own piece (the filter wave's arithmetic of the fused front end,
// gate_fir_step: the 344 raw samples of a 64-output block through an LDS tile, 25 in-order adds per lane), writes y for the
// passes behind it and the decoder, and goes on with |x| and the sums out of registers: the loads of one wave's next blocks
// are in flight while the SIMD's other waves add.
//
// Without y there is no cut search before the pass, so the pieces lie where the pass itself finds the carrier at rest, by
// a rule that is a pure function of the three 64-sample blocks in front of a block boundary -- which is what makes two
// waves agree on it without talking: slot j's piece starts at the first block boundary b in [j P, j P + P/2] whose two
// preceding blocks hold carrier only ("quiet": no sample with |y|^2 < 0.7225 of the largest |y|^2 of its own and the
// preceding block -- 0.85 of the amplitude, as in ls_cut_body), the slot is not in use when there is none.  The wave of
// slot j filters three blocks in front of j P to find its start (nothing is written from there); the wave of the piece
// before runs on through every block up to that same boundary: it evaluates the same rule on the same values.  Every
// block's not-carrier mask goes to HBM (8 B per 64 samples) and ls2_idle_cut_kernel finds the units' idle cuts in them.
constexpr int LS2_FRONT_PRE = 3;         // blocks filtered in front of a slot's grid point: one for its maximum, two that must be quiet
constexpr int LS2_FRONT_AHEAD = 2;   // blocks of raw samples in flight per wave (12 VGPRs each)
constexpr int LS2_FRONT_GIVE_UP = 32;    // a piece that has found no rest point within this many nominal lengths gives the pass up (the sequential scan takes over)
constexpr float LS2_CARRIER_FRAC2 = 0.7225f;

RFID_DEVICE uint32_t ls2_wave_max_bits(uint32_t v) {   // (values below 2^31: bit patterns of non-negative binary32 numbers)
  int x = (int)v, o;
  o = wv::dpp_row_shr<1>(x, 0); x = (o > x) ? o : x;
  o = wv::dpp_row_shr<2>(x, 0); x = (o > x) ? o : x;
  o = wv::dpp_row_shr<4>(x, 0); x = (o > x) ? o : x;
  o = wv::dpp_row_shr<8>(x, 0); x = (o > x) ? o : x;
  o = wv::dpp_row_bcast15(x, 0); x = (o > x) ? o : x;
  o = wv::dpp_row_bcast31(x, 0); x = (o > x) ? o : x;
  return (uint32_t)wv::readlane(x, 63);
}

// The look-back of the fused first pass: a better first guess.  The mean of the amplitude ring is what avg_ampl is up to the
// rounding errors of all additions so far -- a random walk that is ~5 000 ulps off after 4e8 samples, which is why two
// thirds of configs[2]'s pieces had to be run a second time (a guess proves nothing beyond the distance of the nearest
// partial sum to a power of two, a few thousand ulps).  But that drift moves slowly: ~17 ulps per piece.  So every wave
// leaves behind what it has found out -- its piece as a function of the start value (Ls2A32) as soon as it is through,
// and, once the functions of the slots before it reach back to a slot that knows its start, avg_ampl behind its own
// piece and its drift (true start - ring mean) -- and a wave that begins takes the drift of the newest slot that knows
// it (Ls2Args::lb_water): off by the walk over the ~4 000 pieces in flight (~1 100 ulps), not by the whole past.  Exactly
// the single-pass scan with decoupled look-back -- but only for the GUESS: functions of pieces whose runs are not proven are
// off by an ulp or two, nothing here is trusted, the chain kernels behind this pass prove every start as before.
// (Workgroup b takes slot b: the dispatch order is the trace's order, a wave only ever waits for slots before its own.)
constexpr bool LS2_LB = true;
constexpr int LS2_LB_WINDOWS = 64;   // at most this many steps back (4 096 slots: more than are in flight)
// (one 8-byte word per slot and kind, written once with a device-coherent store and read with device-coherent loads: no
// flag beside the data, no release / acquire -- see wv::store_u64_agent)
RFID_DEVICE void ls2_lb_publish_fn(const Ls2Args &a, const int i, const Ls2A32 f, const int lane) {
  if (LS2_LB && lane == 0) wv::store_u64_agent(a.lb_fn + i, (uint64_t)(uint32_t)f.c0 | ((uint64_t)((uint32_t)f.c1 ^ 0x80000000u) << 32));
}
// avg_ampl (integer image) at the first sample of slot j >= 1 of the trace whose slot 0 is `base`: the functions of the slots
// before it, back to the nearest one that knows the value behind its piece.  64 slots per step, in the trace's order over
// the lanes.  NEVER waits: a slot on the way that has not published anything yet (its wave is still at work) ends the attempt
// (false) -- waiting would hold this wave's place on the device until the slowest of the waves before it is through, and the
// whole launch would run in lock step.  So only a wave that happens to finish behind everything in front of it comes to know
// its start; with pieces of 1 .. 1.5 nominal lengths that is every few hundredth, which is all the guesses need.

RFID_DEVICE bool ls2_lb_start(const Ls2Args &a, const int base, const int j, const int lane, int &t_start) {
  Ls2A32 acc; acc.c0 = 0; acc.c1 = 0;     // the slots behind the window (already accounted for)
  int k = j - 1;
  for (int it = 0; it < LS2_LB_WINDOWS; ++it, k -= 64) {
    const int slot = k - 63 + lane;
    const bool in = slot >= 0;
    const uint64_t we = in ? wv::load_u64_agent(a.lb_end + base + slot) : 0ull;
    const uint64_t wf = in ? wv::load_u64_agent(a.lb_fn + base + slot) : (1ull << 63);   // (before the trace: the identity)
    const uint64_t pm = wv::ballot(we != 0ull);
    const int q = pm ? (63 - (int)__builtin_clzll(pm)) : -1;      // the newest slot of the window that knows its end
    if (wv::ballot(lane > q && wf == 0ull) != 0ull) return false; // a slot in between is still at work
    Ls2A32 el; el.c0 = 0; el.c1 = 0;
    if (lane > q) { el.c0 = (int)(uint32_t)wf; el.c1 = (int)((uint32_t)(wf >> 32) ^ 0x80000000u); }
    const Ls2A32 inc = ls2_wave_incl(el, lane);
    Ls2A32 tot; tot.c0 = wv::readlane(inc.c0, 63); tot.c1 = wv::readlane(inc.c1, 63);
    if (q >= 0) {
      const int t_end = wv::readlane((int)(uint32_t)we, q);
      t_start = ls2_apply32(acc, ls2_apply32(tot, t_end));
      return true;
    }
    acc = ls2_comp32(tot, acc);
    if (k - 63 <= 0) return false;   // (slot 0 has not published its end: an empty or all-zero trace)
  }
  return false;
}
RFID_DEVICE void ls2_front_piece(const Ls2Args &a, const int i, const int lane, float4 *tile4) {
  const int s = i / a.max_b, j = i - s * a.max_b;
  const int n_total = ls2_trace_len(a, s);
  const int64_t G64 = (int64_t)j * a.P;
  Ls2Piece none; none.pos0 = 0; none.len = 0;
  Ls2A32 ident; ident.c0 = 0; ident.c1 = 0;
  if (G64 >= n_total) { if (lane == 0) { a.piece[i] = none; a.cutf[i] = -1; } ls2_lb_publish_fn(a, i, ident, lane); return; }
  const int kg = (int)(G64 >> 6);                                    // the block that starts at the slot's grid point
  const int klim = (int)((G64 + a.P / 2) >> 6);                      // the last block boundary this slot's piece may start at
  const int k0 = (j == 0) ? 0 : (kg - LS2_FRONT_PRE);                // (P >= 512: kg >= 8)
  const int64_t row = (int64_t)s * a.y_stride;
  float2 *yw = a.y_w + row;
  uint64_t *lowm = a.lowm + (int64_t)s * a.cstride;
  const float2 *xs = a.raw + (int64_t)s * a.raw_stride;
  const bool vec = a.raw_vec_ok != 0;
  const int64_t hi_idx = vec ? ((a.raw_stride - 2) & ~(int64_t)1) : (a.raw_stride - 2);
  const int64_t rbase = -(int64_t)(NTAPS - 1);                       // raw index of the window of output 0

  // ---- state of the search for block boundaries (all wave-uniform but a2 / a1) ----
  float a2 = 0.0f, a1 = 0.0f;          // |x| of the two blocks before the current one (the ring of gate_impl.cc:131 holds the last 100)
  uint32_t Mprev = 0u;                 // the previous block's largest |y|^2 (bit pattern)
  bool qprev = false, qcur = false;    // the two blocks before the current boundary: carrier only?
  bool running = (j == 0);             // inside the piece (the trace's first piece starts at sample 0, from the exact state)
  int kb = 0;                          // the piece's first block
  int jn = j + 1;                      // the next slot whose start range has not gone by
  int64_t Gn = G64 + a.P;
  const int64_t give_up64 = ((int64_t)LS2_FRONT_GIVE_UP * a.P) >> 6;
  const int give_up = (give_up64 > 0x3fffffff) ? 0x3fffffff : (int)give_up64;
  // ---- state of the piece's sums (ls2_avg_piece) ----
  float sA = 0.0f, sB, avA, avB;
  uint32_t sbA, sbB;
  int marg, min_dv = 0x7fffffff;
  bool e0_ok;
  Ls2MantRange rgA, rgB;
  uint64_t my_lt = 0, my_gt = 0;
  uint64_t *votes = a.votes + 2 * ((int64_t)s * a.vstride);
  auto begin_piece = [&](const int kfirst) {
    kb = kfirst;
    running = true;
    sB = ls2_from_ord(ls2_ord(sA) + 1);
    sbA = wv::f2u(sA); sbB = wv::f2u(sB);
    avA = sA; avB = sB;
    marg = ls2_margin(sA, sbA);
    e0_ok = ls2_e0_ok(sbA, sbB);
    ls2_range_init(rgA); ls2_range_init(rgB);
    votes += 2 * (int64_t)kfirst;
  };
  if (j == 0) begin_piece(0);          // sA = +0: the fresh gate (gate_impl.cc:45)
  int end = 0, rc = 0;                 // rc: 1 the piece has ended at `end`, 2 the slot is not in use, 3 given up

  GateRawRegs buf[LS2_FRONT_AHEAD];
#pragma unroll
  for (int u = 0; u < LS2_FRONT_AHEAD; ++u) {
    gate_load_raw(buf[u], xs, hi_idx, rbase + (int64_t)(k0 + u) * 64 * DECIM, lane, vec);
    wv::compiler_fence();
  }
  // the drift of the newest slot of this trace that knows it (the loads above are in flight meanwhile)
  int drift = 0, g_ord = 0;
  if (LS2_LB && j > 0) {
    const int w = wv::uniform(wv::load_coherent_i32(a.lb_water + s));
    if (w > 0 && w <= j) drift = wv::uniform((int)(uint32_t)(wv::load_u64_agent(a.lb_end + s * a.max_b + w - 1) >> 32));
    if (drift > (1 << 20) || drift < -(1 << 20)) drift = 0;
  }
  auto block = [&](const int k, GateRawRegs &rb) -> int {
    const float2 yv = gate_fir_step(rb, tile4, lane, k == 0);
    gate_load_raw(rb, xs, hi_idx, rbase + (int64_t)(k + LS2_FRONT_AHEAD) * 64 * DECIM, lane, vec);
    const int idx = 64 * k + lane;
    const bool valid = idx < n_total;
    const float amp = wv::hypot_f(yv.x, yv.y);
    // carrier or not: against the largest |y|^2 of this block and the one before (a NaN does not count as the largest)
    const float m2 = yv.x * yv.x + yv.y * yv.y;
    const uint32_t Mk = ls2_wave_max_bits((valid && m2 == m2) ? wv::f2u(m2) : 0u);
    const float theta = LS2_CARRIER_FRAC2 * wv::u2f((Mk > Mprev) ? Mk : Mprev);
    const uint64_t lowmask = wv::ballot(!valid) | wv::ballot(m2 < theta);
    if (running) {
      if (valid) yw[idx] = yv;
      if (lane == 0) lowm[k] = lowmask;
      // sample i - 100: lanes 0..35 take it from two blocks back (lane + 28), lanes 36..63 from the previous block (lane - 36)
      const float o2 = wv::shfl(a2, (lane + 28) & 63), o1 = wv::shfl(a1, (lane - 36) & 63);
      const float old = (lane < 36) ? o2 : o1;
      const float nd = valid ? (amp - old) : 0.0f;
      const float d = div_const<WIN_LEN>(nd);
      float vA, vB;
      const uint32_t cinA = wv::f2u(avA), cinB = wv::f2u(avB);
      const bool scanned = chain_add_auto2(avA, avB, d, lane, vA, vB);
      avA = wv::readlane(vA, 63);
      avB = wv::readlane(vB, 63);
      const float thresh = vA * THRESH_FRACTION;
      const uint64_t below = wv::ballot(valid && amp < thresh);
      const uint64_t above = wv::ballot(valid && amp > thresh);
      {   // margin: the partial sums of both variants against the powers of two, |x| against the threshold (ls2_avg_piece)
        const uint32_t tb = wv::f2u(thresh), ab = wv::f2u(amp);
        int dv = (int)ab - (int)tb;
        dv = (dv < 0) ? -dv : dv;
        if (scanned && ls2_in_start_binade(cinA, cinB, sbA, sbB, e0_ok)) {
          ls2_range_add(rgA, vA); ls2_range_add(rgB, vB);
          const int dvv = valid ? dv : 0x7fffffff;
          min_dv = (dvv < min_dv) ? dvv : min_dv;
        } else {
          int mm;
          if (scanned) {
            mm = ls2_margin_scanned(vA, vB, cinA, cinB, sbA, sbB, e0_ok);
            if (mm > 0) {
              const int shA = (int)((sbA >> 23) & 0xffu) - (int)((cinA >> 23) & 0xffu);
              const int mV = valid ? ((dv - 3) >> (1 + shA)) : 0x7fffffff;
              mm = (mV < mm) ? mV : mm;
            }
          } else {
            const int mA = ls2_margin(vA, sbA), mB = ls2_margin(vB, sbB);
            const int sh = (int)((sbA >> 23) & 0xffu) - (int)((wv::f2u(vA) >> 23) & 0xffu);
            int mV = ((tb >> 31) != 0u || sh < 0 || sh > 23) ? 0 : (((dv - 3) >> 1) >> sh);
            mV = valid ? mV : 0x7fffffff;
            mm = (mA < mB) ? mA : mB;
            mm = (mV < mm) ? mV : mm;
          }
          marg = (mm < marg) ? mm : marg;
        }
      }
      // the votes: lane (kk & 63) keeps those of block kb + kk until 64 blocks are stored together (every block belongs to
      // one piece only: plain stores)
      const int kk = k - kb;
      if (lane == (kk & 63)) { my_lt = below; my_gt = above; }
    }
    a2 = a1; a1 = amp;
    qprev = qcur; qcur = (lowmask == 0ull); Mprev = Mk;
    // ---- the boundary behind this block ----
    const int kn = k + 1;
    int r = 0;
    if (!running) {
      if ((int64_t)64 * kn >= n_total) r = 2;
      else if (kn >= kg && qcur && qprev) {
        // the piece starts here: the ring is the last 100 samples (36 of the block before the last, the last block), first
        // guess of avg_ampl = its mean
        float part = a1 + ((lane >= 64 - (WIN_LEN - 64)) ? a2 : 0.0f);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) part += wv::shfl_xor(part, off);
        sA = wv::uniform(part) / WIN_LEN_F;
        g_ord = ls2_ord(sA);
        // ... + the drift another slot has found (see ls2_lb_start), while that leaves the value an ordinary positive number
        if (drift != 0 && (wv::f2u(sA) >> 23) >= 25u && (wv::f2u(sA) >> 23) < 255u) {
          const float t = ls2_from_ord(g_ord + drift);
          if (((wv::f2u(t) ^ wv::f2u(sA)) >> 31) == 0u && (wv::f2u(t) >> 23) >= 25u && (wv::f2u(t) >> 23) < 255u) sA = t;
        }
        begin_piece(kn);
      } else if (kn >= klim) r = 2;
    } else {
      const int kk = k - kb;
      if ((int64_t)64 * kn >= n_total) { end = n_total; r = 1; }
      else {
        while ((int64_t)64 * kn > Gn + a.P / 2) { jn++; Gn += a.P; }
        if ((int64_t)64 * kn >= Gn && qcur && qprev) { end = 64 * kn; r = 1; }
        else if (kn - kb >= give_up) r = 3;
      }
      if ((kk & 63) == 63 || r == 1) {
        const int kl = (kk & ~63) + lane;
        if (kl <= kk) { votes[2 * kl] = my_lt; votes[2 * kl + 1] = my_gt; }
      }
    }
    return r;
  };
  for (int k = k0; rc == 0;) {
#pragma unroll
    for (int u = 0; u < LS2_FRONT_AHEAD; ++u) {
      rc = block(k, buf[u]);
      ++k;
      if (rc != 0) break;
    }
  }
  if (rc != 1) {
    if (lane == 0) {
      a.piece[i] = none; a.cutf[i] = -1;
      if (rc == 3) a.ctl->fail = 5;
    }
    ls2_lb_publish_fn(a, i, ident, lane);
    return;
  }
  {
    const int mb = ls2_range_margin(rgA, rgB);
    const int mv = (min_dv == 0x7fffffff) ? min_dv : ((min_dv - 3) >> 1);
    const int mf = (mv < mb) ? mv : mb;
    marg = (mf < marg) ? mf : marg;
  }
  marg = ls2_wave_min(marg);
  if ((((wv::f2u(avA) ^ sbA) | (wv::f2u(avB) ^ sbB)) & 0xff800000u) != 0u) marg = 0;   // (start and end in one binade: see ls2_avg_piece)
  if (lane == 0) {
    Ls2AvgRun r;
    r.s = sA; r.eA = avA; r.eB = avB; r.margin = marg;
    r.wide = 0;
    r.ew[0] = r.ew[1] = r.ew[2] = r.ew[3] = 0.0f;
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
    a.arun[i] = r;
    Ls2Piece pc; pc.pos0 = 64 * kb; pc.len = end - 64 * kb;
    a.piece[i] = pc;
    a.cutf[i] = 64 * kb;
  }
  // what the slots behind this one can use (see ls2_lb_start)
  if (LS2_LB) {
    const Ls2A32 f = ls2_elem32(sA, avA, avB);
    int t_start = ls2_ord(sA);
    bool known = true;
    if (j > 0) {
      ls2_lb_publish_fn(a, i, f, lane);
      known = ls2_lb_start(a, s * a.max_b, j, lane, t_start);
    }
    const int t_end = ls2_apply32(f, t_start);
    if (known && t_end != 0 && lane == 0) {
      wv::store_u64_agent(a.lb_end + i, (uint64_t)(uint32_t)t_end | ((uint64_t)(uint32_t)((j == 0) ? 0 : (t_start - g_ord)) << 32));
      if (j > 0) wv::atomic_max(a.lb_water + s, j + 1);
    }
  }
}
RFID_KERNEL(64) void ls2_front_kernel(Ls2Args a) {
  RFID_SHARED float4 tile4[64 * GATE_RAW_LD];
  const int lane = wv::lane_id();
  const int NS = a.n_streams * a.max_b;
  int i = (int)blockIdx.x;                             // (slots in dispatch order: the look-back looks at lower slots only)
  if (!LS2_LB) { const int per = (NS + 7) >> 3; i = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3); if (((int)blockIdx.x >> 3) >= per) return; }
  if (i < NS) ls2_front_piece(a, i, lane, tile4);
}
// the avg_ampl pieces' links (one thread per slot)
RFID_KERNEL(256) void ls2_link_kernel(Ls2Args a) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  const int NS = a.n_streams * a.max_b;
  const int lane = wv::lane_id();
  bool used = false;
  if (i < NS) {
    const int s = i / a.max_b, j = i - s * a.max_b;
    used = a.piece[i].len > 0;
    int nx = -1, pv = -1;
    if (used) {
      const int n = ls2_trace_len(a, s);
      for (int j2 = j + 1; j2 < a.max_b && (int64_t)j2 * a.P < n; ++j2) if (a.piece[s * a.max_b + j2].len > 0) { nx = s * a.max_b + j2; break; }
      for (int j2 = j - 1; j2 >= 0; --j2) if (a.piece[s * a.max_b + j2].len > 0) { pv = s * a.max_b + j2; break; }
    }
    a.nextv[i] = nx;
    a.prevv[i] = pv;
  }
  const uint64_t m = wv::ballot(used);
  if (lane == 0 && m) wv::atomic_add(&a.ctl->n_pieces, wv::popc64(m));
}
// the units' idle cuts from the blocks' not-carrier masks: for every point J >= 1 of the idle grid the first position p in
// [J Pc, J Pc + Pc/2) whose LS_QUIET preceding samples are all carrier, or -1 (what ls_cut_body finds in y; samples past the
// end of the trace are not carrier).  One thread per point: ~(LS_QUIET + Pc/2) / 64 words.
RFID_KERNEL(256) void ls2_idle_cut_kernel(Ls2Args a) {
  const int t = (int)(blockIdx.x * 256 + threadIdx.x);
  const int NH = a.n_streams * a.max_bc;
  if (t >= NH) return;
  const int s = t / a.max_bc, J = t - s * a.max_bc;
  if (J == 0) return;
  int *out = a.cut + (int64_t)s * a.max_bc + J;
  const int n = ls2_trace_len(a, s);
  const int64_t P64 = (int64_t)J * a.Pc;
  if (P64 >= n) { *out = -1; return; }
  const int Pn = (int)P64;
  int lo = Pn - LS_QUIET - 64;
  if (lo < 0) lo = 0;
  int64_t end64 = P64 + a.Pc / 2;
  const int end = (end64 > n) ? n : (int)end64;
  const uint64_t *lowm = a.lowm + (int64_t)s * a.cstride;
  int pall = lo & ~63;        // behind the last sample known not to be carrier (the look-back starts here)
  int found = -1;
  for (int w = lo >> 6; 64 * w < end; ++w) {
    const uint64_t m = lowm[w];
    if (m) pall = 64 * w + 64 - (int)__builtin_clzll(m);
    const int p = (Pn > pall + LS_QUIET) ? Pn : (pall + LS_QUIET);
    if (p <= 64 * (w + 1)) { found = (p < end) ? p : -1; break; }
  }
  *out = found;
}

// device self-test of the wave scan above (rfid_selftest): in[2 * lane], in[2 * lane + 1] -> inclusive scan, exclusive scan
RFID_KERNEL(64) void ls2_scan_selftest_kernel(const int *in, int *out) {
  const int lane = wv::lane_id();
  Ls2A32 v; v.c0 = in[2 * lane]; v.c1 = in[2 * lane + 1];
  const Ls2A32 i = ls2_wave_incl(v, lane), e = ls2_wave_excl(i, lane);
  out[4 * lane] = i.c0; out[4 * lane + 1] = i.c1; out[4 * lane + 2] = e.c0; out[4 * lane + 3] = e.c1;
}
// device self-test of the two-variant in-order sum (chain_add_auto2: one scan for both carries where that is provably the
// same arithmetic, two scans at a piece's first tie, the chains else): out[lane], out[64 + lane] = the sums from ca / cb,
// out[128] = 1 when the shared scan applied
RFID_KERNEL(64) void ls2_scan2_selftest_kernel(const float *x, float ca, float cb, float *out) {
  const int lane = wv::lane_id();
  float oa, ob;
  const bool scanned = chain_add_auto2(ca, cb, x[lane], lane, oa, ob);
  out[lane] = oa; out[64 + lane] = ob;
  if (lane == 0) out[128] = scanned ? 1.0f : 0.0f;
}
constexpr int LS2_CHAIN_WAVES = LS2_CHAIN_THREADS / 64;

// The chain of a trace's pieces: every piece's true start from the latest runs; what is not proven goes on the re-run
// list of this round.  chain_g workgroups per trace, each over a contiguous run of the trace's slots: sixteen waves, 64
// slots at a time (lane = slot: coalesced reads, a wave-level scan per 64).  A first sweep for the workgroup's total, which
// is published; the totals of the workgroups before it give its prefix; a second sweep for every piece's start.
// A run that was made from six neighbouring starts knows its end for each of them exactly: when the chain lands on one of
// those and the prediction used for it was off, the exact end is put into the piece's function (aover) and counted as
// work left -- the next round's chain then carries no error from it.
RFID_DEVICE bool ls2_wide_end(const Ls2AvgRun &ru, int64_t D, int &end_ord) {   // exact end for true start s + D, if known
  if (!(ru.wide & 2) || D < LS2_WIDE_LO || D > LS2_WIDE_HI) return false;
  const float e = (D == 0) ? ru.eA : (D == 1) ? ru.eB : (D == -2) ? ru.ew[0] : (D == -1) ? ru.ew[1] : (D == 2) ? ru.ew[2] : ru.ew[3];
  end_ord = ls2_ord(e);
  return true;
}
struct Ls2AvgRec { int len; float s, eA, eB; int margin, wide; };   // what the chain needs of a piece (fetched one chunk ahead)
RFID_DEVICE Ls2AvgRec ls2_avg_rec(const Ls2Args &a, int base, int j) {
  Ls2AvgRec r; r.len = 0; r.s = r.eA = r.eB = 0.0f; r.margin = 0; r.wide = 0;
  if (j < a.max_b) {
    r.len = a.piece[base + j].len;
    const Ls2AvgRun *ru = a.arun + base + j;
    r.s = ru->s; r.eA = ru->eA; r.eB = ru->eB; r.margin = ru->margin; r.wide = ru->wide;
  }
  return r;
}
// the slots of workgroup b of chain_g and of its wave `wave`, in chunks of 64
RFID_DEVICE void ls2_chain_range(int n_slots, int g, int b, int wave, int &c_lo, int &c_hi) {
  const int n_chunks = (n_slots + 63) >> 6, cpb = (n_chunks + g - 1) / g, cpw = (cpb + LS2_CHAIN_WAVES - 1) / LS2_CHAIN_WAVES;
  const int b_lo = b * cpb, b_hi = (b_lo + cpb < n_chunks) ? (b_lo + cpb) : n_chunks;
  c_lo = b_lo + wave * cpw;
  c_hi = (c_lo + cpw < b_hi) ? (c_lo + cpw) : b_hi;
  if (c_lo > c_hi) c_lo = c_hi;
}
// workgroup totals -> the composition of everything before this wave (NC scans at once; wagg: [NC][LS2_CHAIN_WAVES + 1] in LDS)
template <int NC>
RFID_DEVICE void ls2_chain_prefix(const Ls2Args &a, int s, int b, int wave, int lane, int tid, const Ls2A32 (&carry)[NC], Ls2A32 *wagg, Ls2A32 (&pre)[NC]) {
  if (lane == 0)
    for (int c = 0; c < NC; ++c) wagg[c * (LS2_CHAIN_WAVES + 1) + wave] = carry[c];
  wv::block_sync();
  if (tid == 0) {
    int *ag = a.cagg + ((int64_t)s * LS2_CHAIN_GMAX + b) * 4;
    for (int c = 0; c < NC; ++c) {
      Ls2A32 t; t.c0 = 0; t.c1 = 0;
      for (int w = 0; w < LS2_CHAIN_WAVES; ++w) t = ls2_comp32(t, wagg[c * (LS2_CHAIN_WAVES + 1) + w]);
      ag[2 * c] = t.c0; ag[2 * c + 1] = t.c1;
    }
    wv::publish(a.cflag + (int64_t)s * LS2_CHAIN_GMAX + b, a.stamp);
    Ls2A32 p[NC];
    for (int c = 0; c < NC; ++c) { p[c].c0 = 0; p[c].c1 = 0; }
    for (int b2 = 0; b2 < b; ++b2) {
      wv::await(a.cflag + (int64_t)s * LS2_CHAIN_GMAX + b2, a.stamp);
      const int *g2 = a.cagg + ((int64_t)s * LS2_CHAIN_GMAX + b2) * 4;
      for (int c = 0; c < NC; ++c) { Ls2A32 t; t.c0 = g2[2 * c]; t.c1 = g2[2 * c + 1]; p[c] = ls2_comp32(p[c], t); }
    }
    for (int c = 0; c < NC; ++c) wagg[c * (LS2_CHAIN_WAVES + 1) + LS2_CHAIN_WAVES] = p[c];
  }
  wv::block_sync();
  for (int c = 0; c < NC; ++c) {
    pre[c] = wagg[c * (LS2_CHAIN_WAVES + 1) + LS2_CHAIN_WAVES];
    for (int w = 0; w < wave; ++w) pre[c] = ls2_comp32(pre[c], wagg[c * (LS2_CHAIN_WAVES + 1) + w]);
  }
}
RFID_KERNEL(LS2_CHAIN_THREADS) void ls2_avg_chain_kernel(Ls2Args a) {
  ls2_tail_prio();
  RFID_SHARED Ls2A32 wagg[LS2_CHAIN_WAVES + 1];
  Ls2Ctl *ctl = a.ctl;
  const int r = a.round;
  const int tid = (int)threadIdx.x, b = (int)blockIdx.x, s = (int)blockIdx.y;
  if (ctl->fail != 0) return;
  if (a.fused && r == 0 && ls2_nothing_to_gain(a)) {   // (the fused first pass ran before the idle cuts were known; else ls2_avg_first_kernel looks)
    if (tid == 0 && b == 0 && s == 0) ctl->fail = 1;
    return;
  }
  if (r > 0 && ctl->avg_count[r - 1] == 0) return;   // settled in an earlier round (avg_count[r] stays 0)
  const int NS = a.n_streams * a.max_b;
  const int lane = wv::lane_id(), wave = wv::uniform(tid >> 6);
  const int base = s * a.max_b;
  if (a.piece[base].len <= 0) return;                 // an empty trace
  int c_lo, c_hi;
  ls2_chain_range(a.max_b, a.chain_g, b, wave, c_lo, c_hi);
  const int T0 = ls2_ord(a.arun[base].s);             // the trace's first piece starts from the exact value
  // ---- sweep 1: this wave's total ----
  Ls2A32 carry[1]; carry[0].c0 = 0; carry[0].c1 = 0;
  Ls2AvgRec nxt = ls2_avg_rec(a, base, 64 * c_lo + lane);
  for (int c = c_lo; c < c_hi; ++c) {
    const Ls2AvgRec ru = nxt;
    if (c + 1 < c_hi) nxt = ls2_avg_rec(a, base, 64 * (c + 1) + lane);
    Ls2A32 el; el.c0 = 0; el.c1 = 0;
    if (ru.len > 0) {
      if (ru.wide & 4) { const Ls2Aff o = a.aover[base + 64 * c + lane]; el.c0 = o.c0; el.c1 = o.c1; }
      else el = ls2_elem32(ru.s, ru.eA, ru.eB);
    }
    const Ls2A32 incl = ls2_wave_incl(el, lane);
    Ls2A32 tot; tot.c0 = wv::readlane(incl.c0, 63); tot.c1 = wv::readlane(incl.c1, 63);
    carry[0] = ls2_comp32(carry[0], tot);
  }
  Ls2A32 pre[1];
  ls2_chain_prefix<1>(a, s, b, wave, lane, tid, carry, wagg, pre);
  // ---- sweep 2: every piece's true (or predicted) start ----
  int n_rerun = 0, n_left = 0;
  Ls2A32 run = pre[0];
  nxt = ls2_avg_rec(a, base, 64 * c_lo + lane);
  for (int c = c_lo; c < c_hi; ++c) {
    const Ls2AvgRec ru = nxt;
    if (c + 1 < c_hi) nxt = ls2_avg_rec(a, base, 64 * (c + 1) + lane);
    const int i = base + 64 * c + lane;
    const bool in = ru.len > 0;
    Ls2A32 el; el.c0 = 0; el.c1 = 0;
    int tstar = 0;
    if (in) {
      if (ru.wide & 4) { const Ls2Aff o = a.aover[i]; el.c0 = o.c0; el.c1 = o.c1; tstar = o.tstar; }
      else el = ls2_elem32(ru.s, ru.eA, ru.eB);
    }
    const Ls2A32 incl = ls2_wave_incl(el, lane);
    const Ls2A32 upto = ls2_comp32(run, ls2_wave_excl(incl, lane));
    const int T = ls2_apply32(upto, T0);
    if (in) {
      const int64_t D = (int64_t)T - (int64_t)ls2_ord(ru.s);
      const int64_t aD = (D < 0) ? -D : D;
      a.aT[i] = T;
      int e_exact;
      bool hold = false;
      if (D != 0 && (ru.wide & 2) && D >= LS2_WIDE_LO && D <= LS2_WIDE_HI && ls2_wide_end(a.arun[i], D, e_exact)) {
        // the end is known exactly: was it what the chain used?  else the starts behind this piece are off -- the exact
        // end goes into the piece's function, the piece keeps its run, and the chain is redone in the next round (only
        // then is the piece run again from its true start: for its votes)
        const int want = (int)((uint32_t)e_exact - (uint32_t)T);
        const int cq = (T & 1) ? el.c1 : el.c0;
        if (cq != want) {
          Ls2Aff o; o.c0 = (T & 1) ? el.c0 : want; o.c1 = (T & 1) ? want : el.c1; o.tstar = T; o.pad_ = 0;
          a.aover[i] = o;
          a.arun[i].wide = ru.wide | 4;
          n_left++;
          hold = true;
        }
      }
      // A function with an exact end put in serves the start it was made for.  When a later chain lands on another start of
      // that parity -- the piece's own (D = 0: its end is eA), or one its margin covers -- the entry is not this start's: the
      // plain function comes back and the chain is redone.  (Found with an experiment that changed the order in which pieces
      // settle, profiles/r04/ls2_second_half.txt: a piece that settled with D = 0 kept the end of a start two ulps off, and
      // every start behind it was two ulps off, in a pass that reported success.  The full-size parity tests never met it.)
      const bool will_list = !hold && D != 0 && !(aD + 4 <= (int64_t)ru.margin);
      if ((ru.wide & 4) && T != tstar && !hold && !will_list) {
        a.arun[i].wide = ru.wide & ~4;
        n_left++;
      }
      // settled: the run started from the true value, or provably covers it (its votes included).  Anything else is run
      // again from the true (or predicted) start -- also a piece whose END is known exactly from a neighbouring start: its
      // votes are not.
      if (!hold && D != 0 && !(aD + 4 <= (int64_t)ru.margin)) {
        Ls2AvgRun *w = a.arun + i;
        w->s = ls2_from_ord(T);
        w->wide = (ru.margin < LS2_WIDE_BELOW || r >= 3) ? 1 : 0;
        const int k = wv::atomic_add(&ctl->avg_list[r], 1);
        a.alist[(int64_t)(r & 1) * NS + k] = i;
        n_rerun++;
      }
    }
    Ls2A32 tot; tot.c0 = wv::readlane(incl.c0, 63); tot.c1 = wv::readlane(incl.c1, 63);
    run = ls2_comp32(run, tot);
  }
  if (n_rerun) wv::atomic_add(&ctl->avg_reruns, n_rerun);
  if (n_rerun + n_left) wv::atomic_add(&ctl->avg_count[r], n_rerun + n_left);
  if (tid == 0 && s == 0 && b == 0) ctl->avg_rounds = r + 1;
}

// ---- 3. state machine ----------------------------------------------------------------------------------------------
constexpr int LS2_IDLE_N = GATE_N_SAT;   // the state at an idle cut: saturated count, POS_EDGE, no pulses, closed, next window an RN16

// one wave per unit (head slot): from the head's first sample to the next head, over the recorded votes; scalar work only.
// The unit's steps start at its first sample (step k = samples u0 + 64 k ..), the votes are kept per 64-sample block of
// the trace: every step's two masks are cut out of two neighbouring words.
RFID_KERNEL(64) void ls2_fsm_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (wv::uniform(ctl->fail) != 0) return;
  const int r = a.round;
  if (r == 0) { if (wv::uniform(ctl->avg_count[a.avg_rounds]) != 0) return; }
  else if (wv::uniform(ctl->fsm_count[r - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int NH = a.n_streams * a.max_bc;   // heads sit at every LS2_FINE-th slot only: one block per such slot (blocks indexed by
  for (int b = (int)blockIdx.x; b < NH; b += (int)gridDim.x) {   // slot would all land on two of the eight XCDs)
    const int i = (b / a.max_bc) * a.max_b + (b % a.max_bc) * LS2_FINE;
    if (wv::uniform(a.piece[i].len) <= 0) continue;
    Ls2Fsm *fh = a.fsm + i;
    if (wv::uniform(fh->head) == 0) continue;
    if (r > 0 && wv::uniform(fh->rerun) == 0) continue;
    const int s = i / a.max_b, J = (i - s * a.max_b) / LS2_FINE;
    const int n_total = ls2_trace_len(a, s);
    // the unit: this piece and the ones behind it up to the next head
    const int u0 = wv::uniform(a.piece[i].pos0);
    int u1 = u0 + wv::uniform(a.piece[i].len);
    for (int cur = i;;) {
      const int nx = wv::uniform(a.nextv[cur]);
      if (nx < 0 || wv::uniform(a.piece[nx].len) <= 0 || wv::uniform(a.fsm[nx].head) != 0) break;
      if (lane == 0) a.fsm[nx].unit = i;
      u1 = wv::uniform(a.piece[nx].pos0) + wv::uniform(a.piece[nx].len);
      cur = nx;
    }
    GateRegs g;
    g.avg_c = 0.0f; g.consumed = 0; g.stop = false;
    if (i == s * a.max_b) {   // the trace's first piece: the fresh gate, or the carried state
      if (a.carry) {
        const GateState *cs = a.carry + s;
        g.f_n = wv::uniform(cs->n_samples); g.f_state = wv::uniform(cs->signal_state); g.f_pulses = wv::uniform(cs->num_pulses);
        g.f_open = wv::uniform(cs->gate_open); g.f_ung = wv::uniform(cs->n_to_ungate); g.f_type = wv::uniform(cs->wtype);
      } else {
        g.f_n = 0; g.f_state = 0; g.f_pulses = 0; g.f_open = 0; g.f_ung = 0; g.f_type = 0;
      }
      if (g.f_ung == 0) g.f_ung = g.f_type ? EPC_WIN : RN16_WIN;   // fresh state: first window is an RN16
    } else {
      g.f_n = LS2_IDLE_N; g.f_state = 1; g.f_pulses = 0; g.f_open = 0; g.f_ung = RN16_WIN; g.f_type = 0;
    }
    if (lane == 0) {
      fh->st[0] = g.f_n; fh->st[1] = g.f_state; fh->st[2] = g.f_pulses; fh->st[3] = g.f_open; fh->st[4] = g.f_ung; fh->st[5] = g.f_type;
      fh->rerun = 0;
    }
    const int n = u1 - u0, off = u0 & 63;
    const uint64_t *votes = a.votes + 2 * ((int64_t)s * a.vstride + (u0 >> 6));
    const int64_t cbase = (int64_t)s * a.cstride + (u0 >> 6) + J;
    uint64_t *closed = a.closed + cbase;
    int *oinfo = a.openinfo + cbase;
    Ls2Win *wb = a.wb + (int64_t)s * a.wb_stride;
    const int nsteps = (n + 63) >> 6, nfull = n >> 6;
    int nwin = 0, nepc = 0, last_end = -2147483647 - 1;
    for (int k0 = 0; k0 < nsteps; k0 += 64) {
      // 64 steps at a time: lane L holds the votes of step k0 + L and collects what that step leaves behind
      const int nb = (nsteps - k0 < 64) ? (nsteps - k0) : 64;
      const bool in = lane < nb;
      uint64_t v_lt = in ? votes[2 * (k0 + lane)] : 0ull;        // (a unit's last word + 1 exists: vstride has the room)
      uint64_t v_gt = in ? votes[2 * (k0 + lane) + 1] : 0ull;
      if (off != 0) {
        // step L = bits off.. of word L and bits ..off of word L + 1
        const uint64_t x_lt = wv::uniform(votes[2 * (k0 + 64)]), x_gt = wv::uniform(votes[2 * (k0 + 64) + 1]);   // (word 64 of the block)
        uint64_t n_lt = ((uint64_t)(uint32_t)wv::shfl((int)(uint32_t)(v_lt >> 32), (lane + 1) & 63) << 32) | (uint32_t)wv::shfl((int)(uint32_t)v_lt, (lane + 1) & 63);
        uint64_t n_gt = ((uint64_t)(uint32_t)wv::shfl((int)(uint32_t)(v_gt >> 32), (lane + 1) & 63) << 32) | (uint32_t)wv::shfl((int)(uint32_t)v_gt, (lane + 1) & 63);
        if (lane == 63) { n_lt = x_lt; n_gt = x_gt; }
        v_lt = (v_lt >> off) | (n_lt << (64 - off));
        v_gt = (v_gt >> off) | (n_gt << (64 - off));
      }
      uint64_t my_closed = 0;
      int my_open = 0xff;
      // steps that cannot be skipped while the gate idles (closed, POS_EDGE, no command counted): any vote below the
      // threshold, the partial step at the end, the lanes past the block
      const uint64_t busy = wv::ballot(!(in && k0 + lane < nfull && v_lt == 0ull));
      for (int kk = 0; kk < nb;) {
        const int k = k0 + kk;
        if (!g.f_open && g.f_state == 1 && g.f_pulses <= NUM_PULSES_CMD) {
          // gate closed, nothing pending: every step without a sample below the threshold only counts samples
          const uint64_t m = busy >> kk;
          const int run = m ? wv::ffs64(m) : (64 - kk);
          if (run > 0) {
            if (lane >= kk && lane < kk + run) { my_closed = ~0ull; my_open = 0xff; }
            const int64_t fn = (int64_t)g.f_n + 64ll * run;
            g.f_n = (fn > GATE_N_SAT) ? GATE_N_SAT : (int)fn;
            kk += run;
            continue;
          }
        } else if (g.f_open) {
          // inside a window: the steps that lie wholly inside it
          const int rem = g.f_ung - g.f_n;
          int run = (rem > 64) ? ((rem - 65) / 64 + 1) : 0;
          const int room = nfull - k;
          run = (run < room) ? run : room;
          run = (run < nb - kk) ? run : (nb - kk);
          if (run > 0) {
            if (lane >= kk && lane < kk + run) { my_closed = 0ull; my_open = 0xff; }
            g.f_n += 64 * run;
            kk += run;
            continue;
          }
        }
        int nvalid = (n - 64 * k < 64) ? (n - 64 * k) : 64;
        const uint64_t vm = (nvalid >= 64) ? ~0ull : ((1ull << nvalid) - 1ull);   // (the last step's word holds the next unit's votes too)
        const uint64_t below = (((uint64_t)(uint32_t)wv::readlane((int)(uint32_t)(v_lt >> 32), kk) << 32) | (uint32_t)wv::readlane((int)(uint32_t)v_lt, kk)) & vm;
        const uint64_t above = (((uint64_t)(uint32_t)wv::readlane((int)(uint32_t)(v_gt >> 32), kk) << 32) | (uint32_t)wv::readlane((int)(uint32_t)v_gt, kk)) & vm;
        uint64_t closedmask, openmask;
        int open_lane, open_type;
        gate_fsm_step(0, g, below, above, 64 * k, nvalid, closedmask, openmask, open_lane, open_type);
        if (open_lane != 0xff) {   // gate_impl.cc:164-180
          const int start = u0 + 64 * k + open_lane;
          const int wlen = open_type ? EPC_WIN : RN16_WIN;
          const int complete = (start + wlen <= n_total) ? 1 : 0;   // only complete windows reach the decoder (:223,:291)
          if (lane == 0) {
            Ls2Win *w = wb + start / LS2_WBUCKET;
            if (w->tag != 0 && ((w->tag >> 8) == r + 1) && w->start != start) ctl->wb_clash = 1;   // (the table is cleared before every pass)
            w->start = start;
            w->tag = open_type | (complete << 1) | ((r + 1) << 8);
          }
          nwin += complete;
          nepc += complete & open_type;
          last_end = start + wlen;
        }
        if (lane == kk) { my_closed = closedmask; my_open = open_lane | (open_type << 8); }
        kk += 1;
      }
      if (in) { closed[k0 + lane] = my_closed; oinfo[k0 + lane] = my_open; }
    }
    if (lane == 0) {
      fh->unit = i; fh->gen = r + 1; fh->nwin = nwin; fh->nepc = nepc; fh->last_end = last_end; fh->u1 = u1;
      fh->en[0] = g.f_n; fh->en[1] = g.f_state; fh->en[2] = g.f_pulses; fh->en[3] = g.f_open; fh->en[4] = g.f_ung; fh->en[5] = g.f_type;
      if (r > 0) wv::atomic_add(&ctl->fsm_reruns, 1);
    }
  }
}

// The same with one LANE per unit (long passes).  The scalar unit is shared by a CU's four SIMDs and issues one instruction
// per cycle: with tens of thousands of units ls2_fsm_kernel is bound by exactly that (configs[2]: 338 M scalar instructions,
// 1.3 M per CU = the kernel's 0.62 ms).  Here the state machine of gate_fsm_step runs on the vector unit, 64 units per
// instruction; a lane walks its unit step by step -- the steps of an idle gate and the inside of a window are taken in
// runs, as above -- and fetches the vote words one step ahead.  No wave-level operation below the first line: lanes come and
// go as their units end.
constexpr int LS2_FSM_GROUP = 16;   // steps whose vote words are fetched together
constexpr int LS2_FSM_LANES = 16;   // units per wave: a wave's pace is its slowest lane's and the walk is bound by the latency of
                                      // its scattered loads, so fewer units per wave and more waves per CU
RFID_KERNEL(64) void ls2_fsm_lanes_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (ctl->fail != 0) return;
  const int r = a.round;
  if (r == 0) { if (ctl->avg_count[a.avg_rounds] != 0) return; }
  else if (ctl->fsm_count[r - 1] == 0) return;
  const int NH = a.n_streams * a.max_bc;
  if ((int)threadIdx.x >= LS2_FSM_LANES) return;
  const int b = (int)(blockIdx.x * LS2_FSM_LANES + threadIdx.x);
  if (b >= NH) return;
  const int i = (b / a.max_bc) * a.max_b + (b % a.max_bc) * LS2_FINE;
  if (a.piece[i].len <= 0) return;
  Ls2Fsm *fh = a.fsm + i;
  if (fh->head == 0) return;
  if (r > 0 && fh->rerun == 0) return;
  const int s = i / a.max_b, J = (i - s * a.max_b) / LS2_FINE;
  const int n_total = ls2_trace_len(a, s);
  const int u0 = a.piece[i].pos0;
  int u1 = u0 + a.piece[i].len;
  for (int cur = i;;) {
    const int nx = a.nextv[cur];
    if (nx < 0 || a.piece[nx].len <= 0 || a.fsm[nx].head != 0) break;
    a.fsm[nx].unit = i;
    u1 = a.piece[nx].pos0 + a.piece[nx].len;
    cur = nx;
  }
  GateRegs g;
  g.avg_c = 0.0f; g.consumed = 0; g.stop = false;
  if (i == s * a.max_b) {
    if (a.carry) {
      const GateState *cs = a.carry + s;
      g.f_n = cs->n_samples; g.f_state = cs->signal_state; g.f_pulses = cs->num_pulses;
      g.f_open = cs->gate_open; g.f_ung = cs->n_to_ungate; g.f_type = cs->wtype;
    } else {
      g.f_n = 0; g.f_state = 0; g.f_pulses = 0; g.f_open = 0; g.f_ung = 0; g.f_type = 0;
    }
    if (g.f_ung == 0) g.f_ung = g.f_type ? EPC_WIN : RN16_WIN;
  } else {
    g.f_n = LS2_IDLE_N; g.f_state = 1; g.f_pulses = 0; g.f_open = 0; g.f_ung = RN16_WIN; g.f_type = 0;
  }
  fh->st[0] = g.f_n; fh->st[1] = g.f_state; fh->st[2] = g.f_pulses; fh->st[3] = g.f_open; fh->st[4] = g.f_ung; fh->st[5] = g.f_type;
  fh->rerun = 0;
  const int n = u1 - u0, off = u0 & 63;
  const uint64_t *votes = a.votes + 2 * ((int64_t)s * a.vstride + (u0 >> 6));
  const int64_t cbase = (int64_t)s * a.cstride + (u0 >> 6) + J;
  uint64_t *closed = a.closed + cbase;
  int *oinfo = a.openinfo + cbase;
  Ls2Win *wb = a.wb + (int64_t)s * a.wb_stride;
  const int nsteps = (n + 63) >> 6, nfull = n >> 6;
  int nwin = 0, nepc = 0, last_end = -2147483647 - 1;
  // step k = bits off.. of word k and bits ..off of word k + 1 (a unit's last word + 1 exists: vstride has the room).  The
  // words of LS2_FSM_GROUP steps are fetched together -- one memory latency per group, not per step (a lane's loads are its
  // own: 64 cache lines per instruction, nothing hides them but other lanes' instructions)
  int k = 0;
  while (k < nsteps) {
    const int kg = k;
    uint64_t wl[LS2_FSM_GROUP + 1], wg[LS2_FSM_GROUP + 1];
#pragma unroll
    for (int u = 0; u <= LS2_FSM_GROUP; ++u) {
      const int idx = (kg + u < nsteps) ? (kg + u) : nsteps;
      wl[u] = votes[2 * idx]; wg[u] = votes[2 * idx + 1];
    }
#pragma unroll
    for (int u = 0; u < LS2_FSM_GROUP; ++u) {
      if (k != kg + u || k >= nsteps) continue;   // (a window took the steps up to k in one go, or the unit has ended)
      const uint64_t v_lt = off ? ((wl[u] >> off) | (wl[u + 1] << (64 - off))) : wl[u];
      const uint64_t v_gt = off ? ((wg[u] >> off) | (wg[u + 1] << (64 - off))) : wg[u];
      if (!g.f_open && g.f_state == 1 && g.f_pulses <= NUM_PULSES_CMD && k < nfull && v_lt == 0ull) {
        // gate closed, nothing pending, no sample below the threshold: the step only counts samples
        closed[k] = ~0ull; oinfo[k] = 0xff;
        const int fn = g.f_n + 64;
        g.f_n = (fn > GATE_N_SAT) ? GATE_N_SAT : fn;
        k += 1;
        continue;
      }
      if (g.f_open) {
        // inside a window: the steps that lie wholly inside it need no votes
        const int rem = g.f_ung - g.f_n;
        int run = (rem > 64) ? ((rem - 65) / 64 + 1) : 0;
        const int room = nfull - k;
        run = (run < room) ? run : room;
        if (run > 0) {
          for (int q = 0; q < run; ++q) { closed[k + q] = 0ull; oinfo[k + q] = 0xff; }
          g.f_n += 64 * run;
          k += run;
          continue;
        }
      }
      int nvalid = (n - 64 * k < 64) ? (n - 64 * k) : 64;
      const uint64_t vm = (nvalid >= 64) ? ~0ull : ((1ull << nvalid) - 1ull);   // (the last step's word holds the next unit's votes too)
      uint64_t closedmask, openmask;
      int open_lane, open_type;
      gate_fsm_step(0, g, v_lt & vm, v_gt & vm, 64 * k, nvalid, closedmask, openmask, open_lane, open_type);
      if (open_lane != 0xff) {   // gate_impl.cc:164-180
        const int start = u0 + 64 * k + open_lane;
        const int wlen = open_type ? EPC_WIN : RN16_WIN;
        const int complete = (start + wlen <= n_total) ? 1 : 0;
        Ls2Win *w = wb + start / LS2_WBUCKET;
        if (w->tag != 0 && ((w->tag >> 8) == r + 1) && w->start != start) ctl->wb_clash = 1;
        w->start = start;
        w->tag = open_type | (complete << 1) | ((r + 1) << 8);
        nwin += complete;
        nepc += complete & open_type;
        last_end = start + wlen;
      }
      closed[k] = closedmask; oinfo[k] = open_lane | (open_type << 8);
      k += 1;
    }
  }
  fh->unit = i; fh->gen = r + 1; fh->nwin = nwin; fh->nepc = nepc; fh->last_end = last_end; fh->u1 = u1;
  fh->en[0] = g.f_n; fh->en[1] = g.f_state; fh->en[2] = g.f_pulses; fh->en[3] = g.f_open; fh->en[4] = g.f_ung; fh->en[5] = g.f_type;
  if (r > 0) wv::atomic_add(&ctl->fsm_reruns, 1);
}

// one workgroup per trace: does every unit start from the state its predecessor ended in, with the dc ring a cut assumes
// (the 48 samples before it closed)?  A unit that does not is appended to its predecessor, which is scanned again.
RFID_KERNEL(256) void ls2_fsm_chain_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  const int r = a.round;
  if (ctl->fail != 0) return;
  if (r == 0) { if (ctl->avg_count[a.avg_rounds] != 0) return; }
  else if (ctl->fsm_count[r - 1] == 0) return;
  const int NH = a.n_streams * a.max_bc;
  const int lane = wv::lane_id();
  const int b = (int)(blockIdx.x * 256 + threadIdx.x);
  bool bad = false;
  if (b < NH && (b % a.max_bc) != 0) {   // (a trace's first piece has no predecessor)
    const int i = (b / a.max_bc) * a.max_b + (b % a.max_bc) * LS2_FINE;
    if (a.piece[i].len > 0 && a.fsm[i].head != 0) {
      const int p = a.prevv[i];
      if (p >= 0) {
        const int hp = a.fsm[p].unit;          // the unit the piece before this head belongs to
        const Ls2Fsm &fp = a.fsm[hp];
        const Ls2Fsm &fi = a.fsm[i];
        bool same = fp.last_end <= a.piece[i].pos0 - DC_LEN;
        for (int k = 0; k < 6; ++k) same = same && (fp.en[k] == fi.st[k]);
        if (!same) {
          a.fsm[i].head = 0;
          a.fsm[hp].rerun = 1;   // (if that unit is appended to ITS predecessor in this round, the flag is stale: the
          bad = true;            //  predecessor's unit is flagged by that very mismatch and scans through both)
        }
      }
    }
  }
  const uint64_t m = wv::ballot(bad);
  if (m && lane == 0) wv::atomic_add(&ctl->fsm_count[r], wv::popc64(m));
  if (b == 0) ctl->fsm_rounds = r + 1;
}

// ---- 4. dc_est -----------------------------------------------------------------------------------------------------
// dc_est += (x - dc_samples[dc_index]) / 48 over the closed samples (gate_impl.cc:139-143), two in-order binary32 sums (re, im).
// Its ring holds the last 48 CLOSED samples, so a run can start wherever the 48 samples before were all closed: at a unit's
// head (an idle cut).  What a run cannot know is the VALUE it starts from: the ring's mean is dc_est up to the rounding
// drift of all additions before (hundreds to thousands of ulps late in a long trace).
//
// Rounds 2 - 5 ran every unit from its guess and from one ulp above it, lane = sample, and proved a shifted start by a margin: the
// smallest distance of any partial sum from a power of two.  That proof collapses under noise.  dc_est is a moving average of
// the carrier's components; whenever one of them lies within a few standard deviations of the average's noise of a power of
// two -- SURVEY 8(d)'s own stress model does: 25 sin(0.7) = 16.105 with the average's noise at 0.05 (sigma = 0.03) and 0.10
// (0.06) -- the sums hover ACROSS the binade edge for the whole trace, every unit's margin is a few ulps, nothing is ever
// proven, and what a start value does to the end is no longer a shift: on the two sides of the edge the additions round on
// different grids, two trajectories D ulps apart come out D +- a few apart.  profiles/r06/noise_sweep.txt: at sigma = 0.03 and
// 0.06 every pass of configs[2] / configs[3] gave up in this stage and took the sequential scan (570x / 117x slower).
//
// So this stage no longer proves anything about shifted starts; it KNOWS.  Lane = candidate start: the 64 lanes of a unit's
// wave carry 64 neighbouring start values (centre - 32 .. centre + 31 ulps, per component), the step's 64 increments are
// formed lane = sample as before (gate_dc_incr: the reference's (x - ring) / 48, value for value), laid into LDS, and every
// lane adds all 64 of them in order to its own pair of sums -- 64 plain dependent v_pk_add_f32 out of broadcast LDS reads,
// about half the instructions of the two-variant scan with its tie and margin logic, and each lane's sum IS the
// reference's sum for that start: no binade argument, no ties, no margins.  A unit leaves its 64 ends per component (its
// TABLE) and dc_est at every gate opening for all 64 candidates.
//
// The chain.  A unit is the function "start value -> end value", known exactly on its 64-candidate window; outside the window
// it is continued as a shift from the nearest candidate (a guess, flagged inexact).  Tables compose by lookup -- lane j of
// (g after f) is g[f[j]], ONE wave shuffle when the table lies across the lanes -- so the trace's chain of units is evaluated
// in levels of 64: up (blocks of 64 units -> block tables -> groups of 64 blocks), a walk over the top level from the trace's
// exact start, down (every block's entry value -> every unit's start value).  A unit is SETTLED when everything before it is
// and its own start lies inside its window: its table entry, its openings' dc_est and the next unit's start are then
// exact.  The settled units are a prefix of the trace; everything behind the first unsettled unit is run again, centred on
// the start value the chain predicts for it (exact for that first unit, a guess behind it).  Away from binade edges the
// continuation IS exact, so the second round settles everything (the first round's centres -- the ring means -- are off by
// the drift); where the sums hover at an edge the prediction behind the frontier is off by a few ulps per unit, the window
// catches that for some thousand units, and the frontier advances by that much per round.  When the enqueued rounds are used
// up, ls2_dcb_finish_kernel takes what is left -- only the unsettled units, one after the other from the proven value at the
// frontier, each run from its exact start: the partial fallback (settled units keep their results; the pass costs its clean
// time plus the sequential time of the unsettled units, never the whole sequential gate scan).
constexpr int LS2_DCB_HALF = 32;      // candidate j (= lane) of a unit starts at its centre + j - 32 ulps
constexpr int LS2_DCB_DESCENTS = 1024;  // nodes a chain walk goes through child by child where their tables miss, per wave and launch
constexpr int LS2_DCB_AHEAD = 4096;    // re-run rounds look this many idle-grid slots behind a trace's frontier
#ifndef LS2_DCB_SNAPS_N      // (the test suite's emulator builds with 2: every unit with three gate openings then takes the in-between path)
#define LS2_DCB_SNAPS_N 16
#endif
constexpr int LS2_DCB_SNAPS = LS2_DCB_SNAPS_N;     // gate openings of a unit gathered in LDS before they are written (more: written in between)
constexpr int LS2_DCB_SLACK = 48;     // taken off a run's margin: the estimate of the partial sums is off by < 33 ulps, + the proof's own 4, + spare

RFID_DEVICE bool ls2_fsm_settled(const Ls2Args &a, const Ls2Ctl *ctl) {
  return wv::uniform(ctl->fail) == 0 && wv::uniform(ctl->avg_count[a.avg_rounds]) == 0 && wv::uniform(ctl->fsm_count[a.fsm_rounds]) == 0;
}
// idle-grid slot t = trace * max_bc + J  ->  piece slot
RFID_DEVICE int ls2_dcb_slot(const Ls2Args &a, const int t) { const int s = t / a.max_bc; return s * a.max_b + (t - s * a.max_bc) * LS2_FINE; }

// One unit (idle-grid slot t) from 64 neighbouring start values per component: lane j from centre + j - 32 ulps.
// have_centre: (cre, cim) is the centre (ord images); else the trace's exact start (its first unit) or the ring's mean.
// -> end_re / end_im: lane j's dc_est behind the unit (ord images); also left in a.dtab, the centre in a.dcen
// QUIET (the finishing walk's exploring runs: several waves on one unit, each with a window of its own): nothing is written but
// what the caller gets back -- the 64 ends, the centre actually used (cen_re / cen_im) and the two margins
// DQ (the finishing walk again): 0 the step's increments are formed here; 1 ONLY that -- they are left in Ls2Args::dq, nothing is summed
// (ls2_dcb_incr_kernel); 2 they are read from there: a unit's run is then a load, an LDS round trip and the 64 adds of a step
template <bool QUIET = false, bool NOMARGIN = false, int DQ = 0>   // NOMARGIN (the finishing walk): the margins are not looked at -- not formed, left 0
RFID_DEVICE void ls2_dcb_unit(const Ls2Args &a, const int t, const bool have_centre, int cre, int cim, const bool reserve, const int lane,
                              float2 *lds_dc, float2 *lds_tmp, float2 *lds_q, int &end_re, int &end_im,
                              int *q_cen = nullptr, int *q_mar = nullptr, float2 *lds_snap = nullptr, const int snap_cap = 0) {
  const int s = t / a.max_bc;
  const int i = ls2_dcb_slot(a, t);
  const int upos0 = wv::uniform(a.piece[i].pos0);
  const float2 *yrow = a.y + (int64_t)s * a.y_stride;
  GateBackRegs g;
  g.run_closed = 0; g.ring_stale = 0; g.prev_yv = make_float2(0.0f, 0.0f);
  g.win_seq = 0; g.n_complete = 0; g.written = 0; g.pos0 = upos0; g.strm = s;
  g.dc_index = 0; g.dcr_c = 0.0f; g.dci_c = 0.0f;
  wv::wave_sync();   // (the previous unit's LDS reads are over)
  if (i == s * a.max_b) {   // the trace's first unit: the fresh gate (all zero) or the carried state, exactly
    float sre = 0.0f, sim = 0.0f;
    if (a.carry) {
      const GateState *cs = a.carry + s;
      if (DQ != 2 && lane < DC_LEN) lds_dc[lane] = make_float2(cs->dcr_re[lane], cs->dcr_im[lane]);
      g.dc_index = wv::uniform(cs->dc_index);
      sre = wv::uniform(cs->dc_re); sim = wv::uniform(cs->dc_im);
    } else {
      if (DQ != 2 && lane < DC_LEN) lds_dc[lane] = make_float2(0.0f, 0.0f);
    }
    cre = ls2_ord(sre); cim = ls2_ord(sim);
  } else {
    // an idle cut: the ring holds the 48 samples before it; without a centre: their mean
    float2 v = make_float2(0.0f, 0.0f);
    if ((DQ != 2 || !have_centre) && lane < DC_LEN) { v = yrow[upos0 - DC_LEN + lane]; lds_dc[lane] = v; }
    if (!have_centre) {
      float pr = v.x, pi = v.y;   // (lanes >= 48 hold zeros)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { pr += wv::shfl_xor(pr, off); pi += wv::shfl_xor(pi, off); }
      cre = ls2_ord(wv::uniform(pr) / DC_LEN_F) + a.dcb_bias; cim = ls2_ord(wv::uniform(pi) / DC_LEN_F) - a.dcb_bias;
    }
  }
  wv::wave_sync();
  float2 acc = make_float2(ls2_from_ord(cre + lane - LS2_DCB_HALF), ls2_from_ord(cim + lane - LS2_DCB_HALF));
  // The margin of candidate 32's run (rounds 2 - 5 proved every shifted start with it; here it EXTENDS the table): while every
  // partial sum of the run stays at least m ulps of the start's binade away from every power of two (and in no binade above the
  // start's), a start D ulps off, |D| + slack <= m, D even, gives the same trajectory shifted by D -- rfid_ls2.hpp's header.  Odd D:
  // the same from candidate 33.  So outside the 64-candidate window the unit's end is still KNOWN wherever the margin reaches --
  // away from binade edges that is thousands of ulps, and the first round settles nearly everything; where the sums hover at an
  // edge the margin is nothing and only the window counts.  The partial sums are looked at lane = sample: candidate 32's value
  // before the step + a prefix sum of the step's increments -- an estimate of the in-order sums, off by at most the step's
  // accumulated rounding (LS2_DCB_SLACK covers it).
  const uint32_t sbr = wv::f2u(ls2_from_ord(cre)), sbi = wv::f2u(ls2_from_ord(cim));
  int mre = ls2_margin(ls2_from_ord(cre), sbr), mim = ls2_margin(ls2_from_ord(cim), sbi);
  const bool e0r_ok = ls2_e0_ok(sbr, sbr), e0i_ok = ls2_e0_ok(sbi, sbi);
  Ls2MantRange rgr, rgi;
  ls2_range_init(rgr); ls2_range_init(rgi);
  Ls2Win *wb = a.wb + (int64_t)s * a.wb_stride;
  // where the unit's gate openings go in a.dcand: reserved once (complete windows + the one a trace may end in)
  int wslot = -1;
  if (QUIET) {
  } else if (reserve) {
    const int want = wv::uniform(a.fsm[i].nwin) + 1;
    int got = 0;
    if (lane == 0) got = wv::atomic_add(&a.ctl->dc_open_alloc, want);
    wslot = wv::uniform(got);
    if (lane == 0) a.dwbase[t] = wslot;
    if (wslot + want > a.dcand_cap) { if (lane == 0) a.ctl->fail = 6; wslot = -1; }   // (cannot happen: the capacity is every window + a spare per unit)
  } else {
    wslot = wv::uniform(a.dwbase[t]);
    if (wslot + wv::uniform(a.fsm[i].nwin) + 1 > a.dcand_cap) wslot = -1;
  }
  {
    // the unit: from the head's first sample to the next head (the state-machine pass left its end, its closed samples and its
    // gate openings, step k = samples upos0 + 64 k ..)
    const int n = wv::uniform(a.fsm[i].u1) - upos0;
    const int64_t cbase = (int64_t)s * a.cstride + (upos0 >> 6) + (i - s * a.max_b) / LS2_FINE;
    const uint64_t *closed = a.closed + cbase;
    const int *oinfo = a.openinfo + cbase;
    const float2 *ys = ((DQ == 2) ? (const float2 *)(a.dq + (int64_t)s * a.y_stride) : yrow) + upos0;   // (DQ = 2: the step's increments instead of its samples)
    float2 *dqw = a.dq + (int64_t)s * a.y_stride + upos0;
    const int nsteps = (n + 63) >> 6;
    constexpr int AHEAD = 4;   // (8 in the finishing walk's one-wave workgroups: no difference)
    float2 buf[AHEAD];
    // (loads clamped, not predicated, and the complete groups of AHEAD steps without a condition around a step: see
    // ls2_avg_piece.  Samples past the unit's end are not closed -- nvalid -- whatever their value.)
    const int last_idx = n - 1;
    // Only the steps that hold closed samples are read: more than half of a Gen2 round lies inside the two reply windows, where
    // dc_est does not move (gate_impl.cc:139: the update sits in the closed branch).  Which steps those are is known 64 steps
    // ahead (nz_cur / nz_nxt: one bit per step of this and the next 64-step block); a step that is not needed loads the unit's
    // first samples again instead -- an unconditional load from a line that is in the cache: no branch around a load, the
    // read-ahead's registers and counted waits stay as they are.  (The samples of the step before a closed step are only looked
    // at when that step was closed itself: gate_dc_incr takes x[i-48] from them while run_closed >= 48.)
    uint64_t nz_cur = 0, nz_nxt = 0;
    auto nz_of = [&](const int k64) -> uint64_t {   // bit b: step k64 + b holds closed samples
      const bool in = k64 + lane < nsteps;
      const uint64_t mk = closed[in ? (k64 + lane) : (nsteps - 1)];
      return wv::ballot(in && mk != 0ull);
    };
    nz_cur = nz_of(0);
    nz_nxt = nz_of(64);
    auto load_step = [&](const int q, const int blk) -> float2 {   // the samples of step q (blk: the 64-step block the caller is in)
      const uint64_t nz = ((q >> 6) == blk) ? nz_cur : nz_nxt;
      const int b0 = ((nz >> (q & 63)) & 1ull) ? 64 * q : 0;
      const int idx = b0 + lane;
      return ys[(idx < n) ? idx : last_idx];
    };
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) buf[u] = load_step(u, 0);
    float2 before = make_float2(0.0f, 0.0f);   // the samples of the previous step
    uint64_t masks = 0;
    int oi_l = 0xff;
    int nopen = 0;
    const float4 *q4 = reinterpret_cast<const float4 *>(lds_q);
    // dc_est at the unit's gate openings: gathered in LDS, written behind the loop -- no global store among the loop's read-ahead
    // loads (where a load and a store may both be in flight the compiler waits for everything).  Measured: nothing.  A run of the
    // finishing walk in this (plain) form takes 126 us in a lone wave against 62 in the quiet form, the first dc_est round of
    // configs[2] 1.45 ms against 1.1, and neither the stores, nor the sums behind an opening, nor the in-between writes are it
    // (each taken out in turn: 125 - 128 us; profiles/r06/noise_sweep.txt) -- the first thing to find next.
    int nflushed = 0;
    int *snap_pos = reinterpret_cast<int *>(lds_snap + (int64_t)snap_cap * 64);
    auto flush_snaps = [&]() {
      for (int b = 0; b < nopen - nflushed; ++b) {
        const int idx = nflushed + b;
        if (wslot >= 0) a.dcand[(int64_t)(wslot + idx) * 64 + lane] = lds_snap[b * 64 + lane];
        if (lane == 0) { Ls2Win *w = wb + snap_pos[b] / LS2_WBUCKET; w->slot = (wslot >= 0) ? (wslot + idx) : 0; w->unit = t; }
      }
      nflushed = nopen;
    };
    auto put_snap = [&](const float2 v, const int at) {
      if (nopen - nflushed == snap_cap) { flush_snaps(); wv::drain_vm(); }   // (more openings than the buffer holds: hardly ever)
      lds_snap[(nopen - nflushed) * 64 + lane] = v;
      if (lane == 0) snap_pos[nopen - nflushed] = at;
      nopen++;
    };
    auto step = [&](const int k, float2 &yb, const bool reload) {
      if ((k & 63) == 0) {
        const bool in = k + lane < nsteps;
        const int kx = in ? (k + lane) : (nsteps - 1);
        const uint64_t mk = closed[kx];
        const int ok = oinfo[kx];
        masks = in ? mk : 0ull;
        oi_l = in ? ok : 0xff;
        if (k > 0) { nz_cur = nz_nxt; nz_nxt = nz_of(k + 64); }
      }
      const float2 yv = yb;
      if (reload) yb = load_step(k + AHEAD, k >> 6);
      const int kk = k & 63;
      const uint64_t closedmask = ((uint64_t)(uint32_t)wv::readlane((int)(uint32_t)(masks >> 32), kk) << 32) |
                                  (uint32_t)wv::readlane((int)(uint32_t)masks, kk);
      const int oi = wv::readlane(oi_l, kk);
      const int nvalid = (n - 64 * k < 64) ? (n - 64 * k) : 64;
      const int ol = oi & 0xff;
      if (closedmask != 0) {
        float tre, tim;
        if (DQ == 2) { const bool isc = ((closedmask >> lane) & 1ull) != 0ull; tre = isc ? yv.x : 0.0f; tim = isc ? yv.y : 0.0f; }   // (lanes past the unit's end loaded something else)
        else
        gate_dc_incr(g, closedmask, 0ull, nvalid, yv, lane, lds_dc, lds_tmp,
                     [&](float &qre, float &qim) {
                       // (x - x[i-48]) / 48 as the producer wave forms it: x[i-48] from the previous step's lanes 16..63
                       // or this step's lanes 0..15
                       const int src = (lane < DC_LEN) ? (lane + 64 - DC_LEN) : (lane - DC_LEN);
                       const float pre = wv::shfl(before.x, src), pim = wv::shfl(before.y, src);
                       const float cr2 = wv::shfl(yv.x, src), ci2 = wv::shfl(yv.y, src);
                       const float nr = yv.x - ((lane < DC_LEN) ? pre : cr2), ni = yv.y - ((lane < DC_LEN) ? pim : ci2);
                       if (__builtin_expect(wv::ballot(!(div_const_ok(nr) && div_const_ok(ni))) == 0, 1)) {
                         qre = div_const_fast<DC_LEN>(nr); qim = div_const_fast<DC_LEN>(ni);
                       } else {
                         qre = wv::fdiv(nr, DC_LEN_F); qim = wv::fdiv(ni, DC_LEN_F);
                       }
                     },
                     tre, tim);
        if (DQ == 1) {
          if (lane < nvalid) dqw[64 * k + lane] = make_float2(tre, tim);
          before = yv;
          return;
        }
        if (!NOMARGIN) {
          const float c32r = wv::readlane(acc.x, LS2_DCB_HALF), c32i = wv::readlane(acc.y, LS2_DCB_HALF);
          const float pr = c32r + wv::scan_add_f(tre), pi = c32i + wv::scan_add_f(tim);
          // (nearly every step stays in the start's binade: there the margin is the range of the mantissas, formed once behind the
          // loop -- ls2_range_margin, as in ls2_avg_piece)
          if (__builtin_expect(wv::ballot(((wv::f2u(pr) ^ sbr) & 0xff800000u) != 0u) == 0ull && e0r_ok, 1)) ls2_range_add(rgr, pr);
          else { const int m1 = ls2_margin(pr, sbr); mre = (m1 < mre) ? m1 : mre; }
          if (__builtin_expect(wv::ballot(((wv::f2u(pi) ^ sbi) & 0xff800000u) != 0u) == 0ull && e0i_ok, 1)) ls2_range_add(rgi, pi);
          else { const int m2 = ls2_margin(pi, sbi); mim = (m2 < mim) ? m2 : mim; }
        }
        // the step's 64 increments in sample order (samples that are not closed: +0), added in that order by every lane to
        // its own candidate
        wv::wave_sync();   // (the reads of the step before are over)
        lds_q[lane] = make_float2(tre, tim);
        wv::wave_sync();
        {
          if (!QUIET && __builtin_expect(ol != 0xff, 0)) {
            // a window opened at sample `ol` of this step: dc_est right behind that sample (the opening sample is still closed,
            // gate_impl.cc:164-180), for every candidate -- the sums as far as that sample, from the step's start, BESIDE the step's
            // own 64 below.  (One loop over the 64 samples with a snapshot at `ol` cost 6.5 us a step in a lone wave -- every
            // iteration a full LDS round trip; unrolled, 219 VGPRs.  A run of the finishing walk in the plain form took 129 us
            // against 65 in the quiet one.)
            float2 snap = acc;
            const int npair = (ol + 1) >> 1;
#pragma unroll 4
            for (int j = 0; j < npair; ++j) {
              const float4 qq = q4[j];
              wv::pk_add(snap, make_float2(qq.x, qq.y));
              wv::pk_add(snap, make_float2(qq.z, qq.w));
            }
            if ((ol + 1) & 1) wv::pk_add(snap, lds_q[ol]);
            put_snap(snap, upos0 + 64 * k + ol);
            wv::wave_sync();   // (and the reads below are reads of their own: kept in registers across both loops they cost the kernel half its waves)
          }
          // (unrolled all the way: the reads run ahead of the adds as far as the scheduler lets them)
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            const float4 qq = q4[j];
            wv::pk_add(acc, make_float2(qq.x, qq.y));
            wv::pk_add(acc, make_float2(qq.z, qq.w));
          }
        }
      } else {
        g.run_closed = 0;   // the step lies entirely inside a window: dc_est, the ring and its index do not move
        if (!QUIET && ol != 0xff) put_snap(acc, upos0 + 64 * k + ol);   // (an opening sample is closed itself: not reached)
      }
      before = yv;
    };
    int kb = 0;
    for (; kb + AHEAD <= nsteps; kb += AHEAD) {
#pragma unroll
      for (int u = 0; u < AHEAD; ++u) step(kb + u, buf[u], true);
    }
#pragma unroll
    for (int u = 0; u < AHEAD - 1; ++u)
      if (kb + u < nsteps) step(kb + u, buf[u], false);
    if (!QUIET) flush_snaps();
  }
  if (DQ == 1) return;
  end_re = ls2_ord(acc.x); end_im = ls2_ord(acc.y);
  if (!QUIET && lane == 0) { a.dexm[2 * t] = ~0ull; a.dexm[2 * t + 1] = ~0ull; }
  {
    { const int q = ls2_range_margin(rgr, rgr); mre = (q < mre) ? q : mre; }
    { const int q = ls2_range_margin(rgi, rgi); mim = (q < mim) ? q : mim; }
    mre = ls2_wave_min(mre) - LS2_DCB_SLACK; mim = ls2_wave_min(mim) - LS2_DCB_SLACK;
    if (NOMARGIN) { mre = 0; mim = 0; }
    // (the chain works on the integer image of binary32: a shift by D at the start is a shift by D at the end only if both lie in one binade)
    if (((wv::f2u(wv::readlane(acc.x, LS2_DCB_HALF)) ^ sbr) & 0xff800000u) != 0u || mre < 0) mre = 0;
    if (((wv::f2u(wv::readlane(acc.y, LS2_DCB_HALF)) ^ sbi) & 0xff800000u) != 0u || mim < 0) mim = 0;
    if (q_cen) { q_cen[0] = cre; q_cen[1] = cim; }
    if (QUIET) { q_mar[0] = mre; q_mar[1] = mim; return; }
    if (lane == 0) { a.dmar[2 * t] = mre; a.dmar[2 * t + 1] = mim; }
  }
  a.dtab[(int64_t)(2 * t) * 64 + lane] = end_re;
  a.dtab[(int64_t)(2 * t + 1) * 64 + lane] = end_im;
  if (lane == 0) { a.dcen[2 * t] = cre; a.dcen[2 * t + 1] = cim; }
}

// round 0: every unit from its guess; round r > 0: the units the last chain found not covered again, centred on the start it
// predicted.  (A first round from two starts only -- candidates 32 / 33 summed lane = sample by the two-carry scan, the unit's margin
// carrying them to any other start -- was tried for long passes: 2.34 ms for configs[2]'s first round against 1.5 ms in this form,
// whose 64 candidates cost ONE packed add per sample; profiles/r06/dcb_first_round.txt.)
RFID_DEVICE void ls2_dcb_run(const Ls2Args &a, float2 *lds_dc, float2 *lds_tmp, float2 *lds_q, float2 *lds_snap) {
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  const int r = a.round;
  if (r > 0 && wv::uniform(ctl->dc_count[r - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int NH = a.n_streams * a.max_bc;
  int n_run = 0;
  for (int t = (int)blockIdx.x; t < NH; t += (int)gridDim.x) {
    const int i = ls2_dcb_slot(a, t);
    if (r == 0) {
      if (wv::uniform(a.piece[i].len) <= 0 || wv::uniform(a.fsm[i].head) == 0 || wv::uniform(a.piece[i].pos0) >= wv::uniform(a.fsm[i].u1)) continue;
    } else {
      // again: the units whose latest run does not cover the start the chain predicts for them.  A unit that was run again twice and
      // is STILL not covered has sums that hover at a binade edge (no margin, and the prediction is off by more than the window
      // however often it is renewed): from then on only within LS2_DCB_AHEAD slots of the trace's frontier -- runs far behind it are
      // wasted there (bits 4 - 6 of dstat: how often the unit was run again)
      const int st = wv::uniform(a.dstat[t]);
      if (!(st & 4) || (st & 3) == 3 || !(st & 8)) continue;
      const int again = (st >> 4) & 7;
      if (again >= 2 && t > wv::uniform(a.dfront[t / a.max_bc]) + LS2_DCB_AHEAD) continue;
      if (lane == 0) a.dstat[t] = (st & ~0x70) | (((again < 7) ? again + 1 : 7) << 4);
    }
    int er, ei;
    ls2_dcb_unit(a, t, r > 0, (r > 0) ? wv::uniform(a.dT[2 * t]) : 0, (r > 0) ? wv::uniform(a.dT[2 * t + 1]) : 0, r == 0, lane, lds_dc, lds_tmp, lds_q, er, ei, nullptr, nullptr, lds_snap, LS2_DCB_SNAPS);
    if (r == 0 && lane == 0) a.dstat[t] = 4;
    n_run++;
  }
  if (r > 0 && n_run && lane == 0) wv::atomic_add(&ctl->dc_reruns, n_run);
}
// (97 VGPRs, four waves per SIMD.  Five, six and eight waves forced by launch bounds: configs[2]'s eleven launches 2.36 / 2.31 /
// 2.23 ms against 2.06 -- what the registers give the spills take; profiles/r06/dcb_first_round.txt)
RFID_KERNEL(64) void ls2_dcb_run_kernel(Ls2Args a) {
  ls2_tail_prio();
  RFID_SHARED float2 lds_dc[DC_LEN];
  RFID_SHARED float2 lds_tmp[64];
  RFID_SHARED float4 lds_q4[32];
  RFID_SHARED float2 lds_snap[LS2_DCB_SNAPS * 64 + LS2_DCB_SNAPS / 2];   // (dc_est at the unit's gate openings + where they are)
  ls2_dcb_run(a, lds_dc, lds_tmp, reinterpret_cast<float2 *>(lds_q4), lds_snap);
}

// ---- the chain of tables ----
// v -> the end of a node (unit, block, group) whose table lies across the lanes (lane j: the end for the start cen + j - 32):
// inside the window a lookup; outside, the nearest candidate's end shifted along (a guess: ex goes false).  exm: the
// candidates whose ends are themselves exact (a block's table entry is exact only if every lookup inside the block hit).
template <bool UNIFORM = false>   // UNIFORM: v is the same in every lane (a walk): the lookup is a v_readlane instead of a ds_bpermute round trip
RFID_DEVICE void ls2_dcb_apply(int &v, bool &ex, const int tab, const uint64_t exm, const int cen, const int mar) {
  const int D = (int)((uint32_t)v - (uint32_t)cen);
  const int o = D + LS2_DCB_HALF;
  const int aD = (D < 0) ? -D : D;
  const bool inw = o >= 0 && o < 64 && ((exm >> (o & 63)) & 1ull) != 0ull;   // inside the window, at an entry that is there and exact
  // else, inside the margin: candidate 32's end (33's for an odd distance) shifted along -- exact
  const bool far = !inw && D != (int)0x80000000 && aD <= mar;
  const int par = D & 1;
  const int oe = (o < 0) ? 0 : ((o > 63) ? 63 : o);      // (neither: a guess -- the nearest entry that is there, shifted along)
  const int og = (((exm >> oe) & 1ull) != 0ull) ? oe : (LS2_DCB_HALF + par);
  const int oc = inw ? o : (far ? (LS2_DCB_HALF + par) : og);
  const int e = UNIFORM ? wv::readlane(tab, wv::uniform(oc)) : wv::shfl(tab, oc);
  ex = ex && (inw || far) && (((exm >> oc) & 1ull) != 0ull);
  int step = o - oc;
  // A guess beyond the window's ends goes on with the table's own slope there, not with 1: on the integer image of binary32 a unit
  // that starts on one side of a binade edge and ends on the other maps neighbouring starts to ends 2 (or 1/2) apart, and where
  // the sums hover at an edge every other unit does -- guesses continued with slope 1 were off by as much as they were outside
  // (most first misses of the finishing walk at sigma = 0.06 lay more than 256 ulps off).  Only guesses: ex is false already.
  const int a0 = wv::readlane(tab, 0), a2 = wv::readlane(tab, 2), a61 = wv::readlane(tab, 61), a63 = wv::readlane(tab, 63);
  if (!inw && !far && (o < 0 || o > 63)) {
    const uint64_t need = (o < 0) ? 5ull : (5ull << 61);
    int sl = (o < 0) ? (int)((uint32_t)a2 - (uint32_t)a0) : (int)((uint32_t)a63 - (uint32_t)a61);   // the ends of starts two apart
    if ((exm & need) == need && oc == ((o < 0) ? 0 : 63) && sl >= 1 && sl <= 8 && sl != 2) {
      int far_by = (o < 0) ? o : (o - 63);
      far_by = (far_by < -(1 << 20)) ? -(1 << 20) : ((far_by > (1 << 20)) ? (1 << 20) : far_by);
      const int prod = far_by * sl;
      step = (prod >= 0) ? ((prod + 1) >> 1) : -((-prod + 1) >> 1);
    }
  }
  v = (int)((uint32_t)e + (uint32_t)step);
}
// is a start D ulps off the centre covered by the unit's latest run: an entry of its table that is there, or its margin
RFID_DEVICE bool ls2_dcb_covers(const int D, const uint64_t exm, const int mar) {
  const int o = D + LS2_DCB_HALF;
  return (o >= 0 && o < 64 && ((exm >> (o & 63)) & 1ull) != 0ull) || (D != (int)0x80000000 && ((D < 0) ? -D : D) <= mar);
}
RFID_DEVICE uint64_t ls2_readlane64(const uint64_t v, const int l) {
  return ((uint64_t)(uint32_t)wv::readlane((int)(uint32_t)(v >> 32), l) << 32) | (uint32_t)wv::readlane((int)(uint32_t)v, l);
}
// what a level's nodes are made of: level 1 = blocks of 64 units, level 2 = groups of 64 blocks
struct Ls2DcbKids { const int *cen; const int *tab; const uint64_t *exm; const int *val; const int *mar; int per_trace; int total; };
template <int L>
RFID_DEVICE Ls2DcbKids ls2_dcb_kids(const Ls2Args &a) {
  Ls2DcbKids k;
  if (L == 1) { k.cen = a.dcen; k.tab = a.dtab; k.exm = a.dexm; k.val = nullptr; k.mar = a.dmar; k.per_trace = a.max_bc; k.total = a.n_streams * a.max_bc; }
  else { k.cen = a.n1cen; k.tab = a.n1tab; k.exm = a.n1exm; k.val = a.n1val; k.mar = a.n1mar; k.per_trace = a.dcb_n1; k.total = a.n_streams * a.dcb_n1; }
  return k;
}
// A node's 64 children's tables (2 x 64 rows of 64 values, 32 KB in a row) into LDS with every load in flight at once.  The walks
// over a node's children used to fetch each child's two rows four children ahead: the tables were written by other CUs a launch
// ago, a load takes ~2 us, and 64 steps waited half a microsecond each -- 40 - 55 us per chain launch of configs[2], five launches
// per round.  (Rows of children that do not exist are loaded like the others and never looked at; rows past the array are not.)
struct alignas(16) Ls2Int4 { int x, y, z, w; };
constexpr int LS2_DCB_STAGE = 64 * 2 * 64;
RFID_DEVICE void ls2_dcb_stage(const Ls2DcbKids &kd, const int ch0, int *lds_tab, const int lane) {
  const Ls2Int4 *src = reinterpret_cast<const Ls2Int4 *>(kd.tab + (int64_t)(2 * ch0) * 64);
  const int n4 = (kd.total - ch0) * 32;   // (16-byte quarters of rows from the first child's on)
  Ls2Int4 *dst = reinterpret_cast<Ls2Int4 *>(lds_tab);
  wv::wave_sync();   // (the last node's reads are over)
#pragma unroll 8
  for (int q = 0; q < LS2_DCB_STAGE / 256; ++q) {
    const int idx = q * 64 + lane;
    dst[idx] = src[(idx < n4) ? idx : 0];
  }
  wv::wave_sync();
}
// A node's table misses (the entry value lies outside its window and its margin) where its CHILDREN, gone through one by one,
// may all be hit: the walk then descends -- (T, ex) through the children of node `node` of level L in order, a level-2 node's
// children (blocks) through their own tables first and through THEIR children where those miss too.  Wave-uniform values; the
// children's tables lie across the lanes.  `budget`: descents the caller still allows (sums that hover at a binade edge miss
// everywhere: a walk over every unit of a long trace in one wave is what the rounds are there to avoid).
template <int L>
RFID_DEVICE void ls2_dcb_through(const Ls2Args &a, const int node, int &Tre, int &Tim, bool &exr, bool &exi, const int lane, int &budget,
                                 int *lds_own, int *lds_kid) {   // (LDS for this node's children's tables / for a child's own descent)
  const Ls2DcbKids kd = ls2_dcb_kids<L>(a);
  const int nper = (L == 1) ? a.dcb_n1 : a.dcb_n2;
  const int s = node / nper, k = node - s * nper;
  const int ch0 = s * kd.per_trace + 64 * k;
  const bool in = 64 * k + lane < kd.per_trace;
  // the children's centres, margins and which of them exist: one load per lane; their tables through LDS (none of the loads
  // depends on the walk)
  int valid = 0, cre = 0, cim = 0, mre = 0, mim = 0;
  if (in) {
    valid = (L == 1) ? ((a.dstat[ch0 + lane] >> 2) & 1) : kd.val[ch0 + lane];
    if (valid) { cre = kd.cen[2 * (ch0 + lane)]; cim = kd.cen[2 * (ch0 + lane) + 1]; mre = kd.mar[2 * (ch0 + lane)]; mim = kd.mar[2 * (ch0 + lane) + 1]; }
  }
  uint64_t er_l = ~0ull, ei_l = ~0ull;
  if (kd.exm && in && valid) { er_l = kd.exm[2 * (ch0 + lane)]; ei_l = kd.exm[2 * (ch0 + lane) + 1]; }
  const uint64_t m = wv::ballot(valid != 0);
  if (m == 0ull) return;
  ls2_dcb_stage(kd, ch0, lds_own, lane);
  for (uint64_t rest = m; rest; rest &= rest - 1ull) {
    const int l = wv::ffs64(rest);
    const int t_re = lds_own[(2 * l) * 64 + lane], t_im = lds_own[(2 * l + 1) * 64 + lane];
    const uint64_t er = ls2_readlane64(er_l, l), ei = ls2_readlane64(ei_l, l);
    int T2r = Tre, T2i = Tim; bool e2r = exr, e2i = exi;
    ls2_dcb_apply<true>(T2r, e2r, t_re, er, wv::readlane(cre, l), wv::readlane(mre, l));
    ls2_dcb_apply<true>(T2i, e2i, t_im, ei, wv::readlane(cim, l), wv::readlane(mim, l));
    if (L == 2 && ((exr && !e2r) || (exi && !e2i)) && budget > 0) {
      budget--;
      T2r = Tre; T2i = Tim; e2r = exr; e2i = exi;
      ls2_dcb_through<1>(a, ch0 + l, T2r, T2i, e2r, e2i, lane, budget, lds_kid, nullptr);
    }
    Tre = T2r; Tim = T2i; exr = e2r; exi = e2i;
  }
}
// up: the 64 children of node `node` of level L composed in order -> the node's table (on the window of its first child)
template <int L>
RFID_DEVICE void ls2_dcb_up(const Ls2Args &a, const int node, const int lane, int *lds_tab) {
  const Ls2DcbKids kd = ls2_dcb_kids<L>(a);
  const int nper = (L == 1) ? a.dcb_n1 : a.dcb_n2;
  int *ocen = (L == 1) ? a.n1cen : a.n2cen; int *otab = (L == 1) ? a.n1tab : a.n2tab;
  uint64_t *oexm = (L == 1) ? a.n1exm : a.n2exm; int *oval = (L == 1) ? a.n1val : a.n2val; int *omar = (L == 1) ? a.n1mar : a.n2mar;
  const int s = node / nper, k = node - s * nper;
  const int ch0 = s * kd.per_trace + 64 * k;
  const bool in = 64 * k + lane < kd.per_trace;
  int valid = 0, cre = 0, cim = 0, mre = 0, mim = 0;
  if (in) {
    valid = (L == 1) ? ((a.dstat[ch0 + lane] >> 2) & 1) : kd.val[ch0 + lane];
    if (valid) { cre = kd.cen[2 * (ch0 + lane)]; cim = kd.cen[2 * (ch0 + lane) + 1]; mre = kd.mar[2 * (ch0 + lane)]; mim = kd.mar[2 * (ch0 + lane) + 1]; }
  }
  // (which entries of a child's table are there: with its centre, one load per lane -- a wave-uniform load per child inside the
  // walk is a full memory round trip per step that nothing can run ahead of: 50 us per chain launch of configs[2], 15 since)
  uint64_t er_l = ~0ull, ei_l = ~0ull;
  if (kd.exm && in && valid) { er_l = kd.exm[2 * (ch0 + lane)]; ei_l = kd.exm[2 * (ch0 + lane) + 1]; }
  const uint64_t m = wv::ballot(valid != 0);
  if (m == 0ull) { if (lane == 0) oval[node] = 0; return; }
  int l = wv::ffs64(m);
  // the node's window: round 0 around its first child's centre; later around the value the last chain found it entered at --
  // the first child's centre is a ring mean, off by the rounding drift, and an entry value outside the window passes exactly
  // only where EVERY child's margin covers it
  const int *pent = (L == 1) ? a.n1ent : a.n2ent;
  const bool recentre = a.round > 0 && wv::uniform(oval[node]) != 0;
  const int bre = recentre ? wv::uniform(pent[4 * node]) : wv::readlane(cre, l), bim = recentre ? wv::uniform(pent[4 * node + 1]) : wv::readlane(cim, l);
  int vre = bre + lane - LS2_DCB_HALF, vim = bim + lane - LS2_DCB_HALF;
  bool exr = true, exi = true;
  int nmr = 0x3fffffff, nmi = 0x3fffffff;   // the node's own margin: how far from ITS centre an entry value may lie (see below)
  auto dev = [&](const int v, const int c) -> int {   // how far candidates 32 / 33 of the node are from the child's centre when they reach it
    const int d0 = wv::readlane(v, LS2_DCB_HALF) - c, d1 = wv::readlane(v, LS2_DCB_HALF + 1) - c;
    const int a0 = (d0 < 0) ? -d0 : d0, a1 = (d1 < 0) ? -d1 : d1;
    return (a0 > a1) ? a0 : a1;
  };
  ls2_dcb_stage(kd, ch0, lds_tab, lane);
  for (uint64_t rest = m; rest; rest &= rest - 1ull) {
    {
      const int lc = wv::ffs64(rest);
      const int t_re = lds_tab[(2 * lc) * 64 + lane], t_im = lds_tab[(2 * lc + 1) * 64 + lane];
      const uint64_t e_re = ls2_readlane64(er_l, lc), e_im = ls2_readlane64(ei_l, lc);
      const int c_re = wv::readlane(cre, lc), c_im = wv::readlane(cim, lc), m_re = wv::readlane(mre, lc), m_im = wv::readlane(mim, lc);
      // an entry value D off the node's centre reaches this child D + dev off the child's: a plain shift all the way while that
      // stays inside every child's margin
      { const int q = m_re - dev(vre, c_re) - 1; nmr = (q < nmr) ? q : nmr; }
      { const int q = m_im - dev(vim, c_im) - 1; nmi = (q < nmi) ? q : nmi; }
      ls2_dcb_apply(vre, exr, t_re, e_re, c_re, m_re);
      ls2_dcb_apply(vim, exi, t_im, e_im, c_im, m_im);
    }
  }
  otab[(int64_t)(2 * node) * 64 + lane] = vre;
  otab[(int64_t)(2 * node + 1) * 64 + lane] = vim;
  const uint64_t mr = wv::ballot(exr), mi = wv::ballot(exi);
  const uint64_t both = 3ull << LS2_DCB_HALF;   // (the shift goes out from candidates 32 / 33: their own ends must be exact)
  if ((mr & both) != both || nmr < 0) nmr = 0;
  if ((mi & both) != both || nmi < 0) nmi = 0;
  if (lane == 0) {
    ocen[2 * node] = bre; ocen[2 * node + 1] = bim; oexm[2 * node] = mr; oexm[2 * node + 1] = mi; oval[node] = 1;
    omar[2 * node] = nmr; omar[2 * node + 1] = nmi;
  }
}
RFID_KERNEL(64) void ls2_dcb_up1_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  if (a.round > 0 && wv::uniform(ctl->dc_count[a.round - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int N = a.n_streams * a.dcb_n1;
  RFID_SHARED int lds_tab[LS2_DCB_STAGE];
  for (int node = (int)blockIdx.x; node < N; node += (int)gridDim.x) ls2_dcb_up<1>(a, node, lane, lds_tab);
}
RFID_KERNEL(64) void ls2_dcb_up2_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  if (a.round > 0 && wv::uniform(ctl->dc_count[a.round - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int N = a.n_streams * a.dcb_n2;
  RFID_SHARED int lds_tab[LS2_DCB_STAGE];
  for (int node = (int)blockIdx.x; node < N; node += (int)gridDim.x) ls2_dcb_up<2>(a, node, lane, lds_tab);
}
// the walk over a trace's top-level nodes from its exact start (the centre of its first unit): every node's entry value.
// One wave per trace; the values are wave-uniform.
RFID_KERNEL(64) void ls2_dcb_top_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  const int r = a.round;
  if (r > 0 && wv::uniform(ctl->dc_count[r - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int s = (int)blockIdx.x;
  const int t0 = s * a.max_bc;
  RFID_SHARED int lds_a[LS2_DCB_STAGE];   // (descents: a node's children's tables, and a child's own children's)
  RFID_SHARED int lds_b[LS2_DCB_STAGE];
  if (s == 0 && lane == 0) ctl->dc_rounds = r + 1;
  if (lane == 0) a.dfront[s] = 0x7fffffff;
  if (!(wv::uniform(a.dstat[t0]) & 4)) return;   // an empty trace
  const bool two = a.dcb_top == 2;
  const int nper = two ? a.dcb_n2 : a.dcb_n1;
  const int *cen = two ? a.n2cen : a.n1cen; const int *tab = two ? a.n2tab : a.n1tab;
  const uint64_t *exm = two ? a.n2exm : a.n1exm; const int *val = two ? a.n2val : a.n1val; const int *mar = two ? a.n2mar : a.n1mar;
  int *ent = two ? a.n2ent : a.n1ent;
  int Tre = wv::uniform(a.dcen[2 * t0]), Tim = wv::uniform(a.dcen[2 * t0 + 1]);
  bool exr = true, exi = true;
  int budget = LS2_DCB_DESCENTS;
  const int n0 = s * nper;
  int tr = tab[(int64_t)(2 * n0) * 64 + lane], ti = tab[(int64_t)(2 * n0 + 1) * 64 + lane];
  // (the nodes' centres, margins and masks 64 nodes at a time, one load per lane: a wave-uniform load inside the walk is a memory
  // round trip per node that nothing runs ahead of)
  int v_l = 0, cr_l = 0, ci_l = 0, mr_l = 0, mi_l = 0;
  uint64_t er_l = 0ull, ei_l = 0ull;
  for (int k = 0; k < nper; ++k) {
    const int node = n0 + k;
    if ((k & 63) == 0) {
      const int q = node + lane;
      const bool in = k + lane < nper;
      v_l = in ? val[q] : 0;
      cr_l = in ? cen[2 * q] : 0; ci_l = in ? cen[2 * q + 1] : 0; mr_l = in ? mar[2 * q] : 0; mi_l = in ? mar[2 * q + 1] : 0;
      er_l = in ? exm[2 * q] : 0ull; ei_l = in ? exm[2 * q + 1] : 0ull;
    }
    const int kl = k & 63;
    int ntr = 0, nti = 0;
    if (k + 1 < nper) { ntr = tab[(int64_t)(2 * (node + 1)) * 64 + lane]; nti = tab[(int64_t)(2 * (node + 1) + 1) * 64 + lane]; }
    if (wv::readlane(v_l, kl) != 0) {
      if (lane == 0) { ent[4 * node] = Tre; ent[4 * node + 1] = Tim; ent[4 * node + 2] = exr ? 1 : 0; ent[4 * node + 3] = exi ? 1 : 0; }
      int T2r = Tre, T2i = Tim; bool e2r = exr, e2i = exi;
      ls2_dcb_apply<true>(T2r, e2r, tr, ls2_readlane64(er_l, kl), wv::readlane(cr_l, kl), wv::readlane(mr_l, kl));
      ls2_dcb_apply<true>(T2i, e2i, ti, ls2_readlane64(ei_l, kl), wv::readlane(ci_l, kl), wv::readlane(mi_l, kl));
      if (((exr && !e2r) || (exi && !e2i)) && budget > 0) {
        // the node's table missed an exact entry value: through its children one by one (they may all be hit)
        budget--;
        T2r = Tre; T2i = Tim; e2r = exr; e2i = exi;
        if (two) ls2_dcb_through<2>(a, node, T2r, T2i, e2r, e2i, lane, budget, lds_a, lds_b);
        else ls2_dcb_through<1>(a, node, T2r, T2i, e2r, e2i, lane, budget, lds_a, nullptr);
      }
      Tre = T2r; Tim = T2i; exr = e2r; exi = e2i;
    }
    tr = ntr; ti = nti;
  }
}
// down: from a node's entry value to its children's.  Level 1: the children are the units -- their start values (a.dT),
// which of them are settled, how many are not (Ls2Ctl::dc_count[round]).
template <int L>
RFID_DEVICE void ls2_dcb_down(const Ls2Args &a, const int node, const int lane, int &n_uns, int &n_units, int &first_uns, int *lds_tab, int *lds_kid) {
  const Ls2DcbKids kd = ls2_dcb_kids<L>(a);
  const int nper = (L == 1) ? a.dcb_n1 : a.dcb_n2;
  const int *nval = (L == 1) ? a.n1val : a.n2val; const int *nent = (L == 1) ? a.n1ent : a.n2ent;
  if (wv::uniform(nval[node]) == 0) return;
  const int s = node / nper, k = node - s * nper;
  const int ch0 = s * kd.per_trace + 64 * k;
  const bool in = 64 * k + lane < kd.per_trace;
  int valid = 0, cre = 0, cim = 0, mre = 0, mim = 0, st_l = 0;
  if (in) {
    if (L == 1) st_l = a.dstat[ch0 + lane];
    valid = (L == 1) ? ((st_l >> 2) & 1) : kd.val[ch0 + lane];
    if (valid) { cre = kd.cen[2 * (ch0 + lane)]; cim = kd.cen[2 * (ch0 + lane) + 1]; mre = kd.mar[2 * (ch0 + lane)]; mim = kd.mar[2 * (ch0 + lane) + 1]; }
  }
  uint64_t er_l = ~0ull, ei_l = ~0ull;   // (one load per lane: see ls2_dcb_up)
  if (kd.exm && in && valid) { er_l = kd.exm[2 * (ch0 + lane)]; ei_l = kd.exm[2 * (ch0 + lane) + 1]; }
  const uint64_t m = wv::ballot(valid != 0);
  if (m == 0ull) return;
  int Tre = wv::uniform(nent[4 * node]), Tim = wv::uniform(nent[4 * node + 1]);
  bool exr = wv::uniform(nent[4 * node + 2]) != 0, exi = wv::uniform(nent[4 * node + 3]) != 0;
  int budget = 64;
  ls2_dcb_stage(kd, ch0, lds_tab, lane);
  for (uint64_t rest = m; rest; rest &= rest - 1ull) {
    {
      const int l = wv::ffs64(rest);
      const int t_re = lds_tab[(2 * l) * 64 + lane], t_im = lds_tab[(2 * l + 1) * 64 + lane];
      const uint64_t e_re = ls2_readlane64(er_l, l), e_im = ls2_readlane64(ei_l, l);
      const int c_re = wv::readlane(cre, l), c_im = wv::readlane(cim, l), m_re = wv::readlane(mre, l), m_im = wv::readlane(mim, l);
      const int c = ch0 + l;
      if (L == 1) {
        // settled: everything before is, and the unit's own start lies inside its window or its margin (its end, and dc_est at its
        // gate openings, are then known: a table entry, or candidate 32's / 33's shifted along)
        const int D_re = (int)((uint32_t)Tre - (uint32_t)c_re), D_im = (int)((uint32_t)Tim - (uint32_t)c_im);
        const bool k_re = ls2_dcb_covers(D_re, e_re, m_re), k_im = ls2_dcb_covers(D_im, e_im, m_im);
        const int bits = ((exr && k_re) ? 1 : 0) | ((exi && k_im) ? 2 : 0);
        // (bit 3: the unit's latest run does not cover the start predicted for it -- it is run again, centred on that; a unit whose
        // run does cover it only waits for the units before it)
        const int again = wv::readlane(st_l, l) & 0x70;
        if (lane == 0) { a.dT[2 * c] = Tre; a.dT[2 * c + 1] = Tim; a.dstat[c] = 4 | bits | ((k_re && k_im) ? 0 : 8) | again; }
        n_units++;
        if (bits != 3) { if (n_uns == 0) first_uns = c; n_uns++; }
      } else {
        if (lane == 0) { int *e = a.n1ent + 4 * c; e[0] = Tre; e[1] = Tim; e[2] = exr ? 1 : 0; e[3] = exi ? 1 : 0; }
      }
      {
        int T2r = Tre, T2i = Tim; bool e2r = exr, e2i = exi;
        ls2_dcb_apply<true>(T2r, e2r, t_re, e_re, c_re, m_re);
        ls2_dcb_apply<true>(T2i, e2i, t_im, e_im, c_im, m_im);
        if (L == 2 && ((exr && !e2r) || (exi && !e2i)) && budget > 0) {   // (the block's table missed: through its units one by one)
          budget--;
          T2r = Tre; T2i = Tim; e2r = exr; e2i = exi;
          ls2_dcb_through<1>(a, c, T2r, T2i, e2r, e2i, lane, budget, lds_kid, nullptr);
        }
        Tre = T2r; Tim = T2i; exr = e2r; exi = e2i;
      }
    }
  }
}
RFID_KERNEL(64) void ls2_dcb_down2_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  if (a.round > 0 && wv::uniform(ctl->dc_count[a.round - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int N = a.n_streams * a.dcb_n2;
  int u0 = 0, u1 = 0, u2 = 0;
  RFID_SHARED int lds_tab[LS2_DCB_STAGE];
  RFID_SHARED int lds_kid[LS2_DCB_STAGE];
  for (int node = (int)blockIdx.x; node < N; node += (int)gridDim.x) ls2_dcb_down<2>(a, node, lane, u0, u1, u2, lds_tab, lds_kid);
}
RFID_KERNEL(64) void ls2_dcb_down1_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  const int r = a.round;
  if (r > 0 && wv::uniform(ctl->dc_count[r - 1]) == 0) return;
  const int lane = wv::lane_id();
  const int N = a.n_streams * a.dcb_n1;
  int n_uns = 0, n_units = 0;
  RFID_SHARED int lds_tab[LS2_DCB_STAGE];
  for (int node = (int)blockIdx.x; node < N; node += (int)gridDim.x) {
    int nu = 0, first_uns = 0;
    ls2_dcb_down<1>(a, node, lane, nu, n_units, first_uns, lds_tab, nullptr);
    if (nu && lane == 0) wv::atomic_min(a.dfront + first_uns / a.max_bc, first_uns);   // the trace's frontier: its first unit that is not settled
    n_uns += nu;
  }
  if (lane == 0) {
    if (n_uns) wv::atomic_add(&ctl->dc_count[r], n_uns);
    if (r == 0 && n_units) { wv::atomic_add(&ctl->n_units, n_units); wv::atomic_add(&ctl->n_dc_pieces, n_units); }
  }
}
// The enqueued rounds are used up and units are still unsettled (sums that hover at a binade edge: what a start value does to
// a unit's end is then no shift, the window catches the chain's prediction for a few units only, and exactness can only walk
// along the trace).  The finishing walk: G waves per trace -- on the device G single-wave workgroups, each with a CU (nearly) to
// itself; WPB waves per workgroup is the same kernel for the test suite's emulator, which runs one workgroup at a time -- take
// turns.  Per turn wave w runs the w-th idle-grid slot behind the frontier: the first from its exact start, the others centred on
// the last chain's prediction moved along by what the frontier has turned out to be off; the waves meet (a device-wide barrier
// over one counter per trace); then EVERY wave goes through the turn's tables from the exact value -- the same arithmetic on the
// same data, so all of them know the new frontier without a second meeting: as far as each start lies inside its unit's window
// (or margin) the units are settled, the first miss is the next turn's first unit.  At least one unit per turn, up to G; as
// many waves take part as the last turn's reach suggests.  Settled units keep their results: the pass costs its clean time plus
// this walk over the unsettled units -- the partial fallback; the launches behind (window sequence numbers, assembly) find
// everything settled.  (The turn's tables go through a scratch area in HBM, two sets used alternately: a wave may be one
// turn ahead of the slowest, never two.)
//
// Wide windows (second half of round 6).  What limits a turn's reach is the window: behind a miss every prediction is off by what the
// missed unit's guess was off, plus a random walk of a few ulps per unit (at a binade edge two neighbouring trajectories do not stay
// neighbours), and the run of hits ends when that leaves +- 32 -- after ~16 units at sigma = 0.03, ~4 at 0.06.  The first passage of a
// random walk goes with the SQUARE of the distance: so a unit is explored by LS2_FIN_WIN waves at once, each with a 64-candidate window
// of its own, side by side (256 candidates, +- 128), in the quiet form of the unit's run (nothing written but the scratch record).
// A unit that the walk has settled is then run ONCE more in the plain form, centred on its now exact start -- that run leaves the
// table, dc_est at the gate openings and the records the assembly reads -- by a wave from the far end of the trace's waves while the
// others explore the next turn's units.
// The increments of every closed step of every unit that is not settled, once, for the finishing walk (nothing to do -- the usual
// case -- and the launch returns): what a unit's start value does NOT enter.  The walk's runs then read them instead of forming them
// turn after turn, window after window: a turn is as long as a lone wave's run, and forming the increments is two thirds of it
RFID_KERNEL(64) void ls2_dcb_incr_kernel(Ls2Args a) {
  ls2_tail_prio();
  RFID_SHARED float2 lds_dc[DC_LEN];
  RFID_SHARED float2 lds_tmp[64];
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  if (wv::uniform(ctl->dc_count[a.dc_rounds]) == 0) return;
  const int lane = wv::lane_id();
  const int NH = a.n_streams * a.max_bc;
  for (int t = (int)blockIdx.x; t < NH; t += (int)gridDim.x) {
    const int st = wv::uniform(a.dstat[t]);
    if (!(st & 4) || (st & 3) == 3) continue;
    int er, ei;
    ls2_dcb_unit<true, true, 1>(a, t, true, 0, 0, false, lane, lds_dc, lds_tmp, nullptr, er, ei);
  }
}
// the mean of the 48 samples in front of unit t (the dc ring's content at an idle cut: what round 0 centres a unit on) -> false for a
// trace's first unit (its start is the fresh gate's or the carried state's, exactly)
RFID_DEVICE bool ls2_dcb_ring_mean(const Ls2Args &a, const int t, const int lane, float &mre, float &mim) {
  const int s = t / a.max_bc;
  const int i = ls2_dcb_slot(a, t);
  if (i == s * a.max_b) return false;
  const int upos0 = wv::uniform(a.piece[i].pos0);
  const float2 *yrow = a.y + (int64_t)s * a.y_stride;
  float2 v = make_float2(0.0f, 0.0f);
  if (lane < DC_LEN) v = yrow[upos0 - DC_LEN + lane];
  float pr = v.x, pi = v.y;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { pr += wv::shfl_xor(pr, off); pi += wv::shfl_xor(pi, off); }
  mre = wv::uniform(pr) / DC_LEN_F; mim = wv::uniform(pi) / DC_LEN_F;
  return true;
}
constexpr int LS2_FIN_GMAX = 512;       // waves per trace, at most (256 / 512 / 1 024 measured: profiles/r06/noise_sweep.txt)
constexpr int LS2_FIN_TOTAL = 1024;     // ... and of all traces together: one-wave workgroups with 33 KB of LDS, four per CU -- they meet, so all of a trace's must be resident
constexpr int LS2_FIN_WIN_MAX = 8;      // ... at most
constexpr int LS2_FIN_WIN = 4;          // windows (waves) per explored unit when the trace has at least 16 waves
constexpr int LS2_FIN_AHEAD = 64;       // units explored per turn, at least (as far as the waves go)
constexpr int LS2_FIN_REC = 136;        // ints per wave and set: centre (2), margin (2), in use (1), pad (3), table (2 x 64)
constexpr int LS2_FIN_LIST = 256;       // units a turn can hold (the slots of the idle grid that hold one, in order)
constexpr int LS2_FIN_CHUNK = 64;       // scratch records the walk stages through LDS at a time
template <int WPB>
RFID_KERNEL(64 * WPB) void ls2_dcb_finish_kernel(Ls2Args a) {
  ls2_tail_prio();
  RFID_SHARED float2 lds_dc[WPB][DC_LEN];
  RFID_SHARED float2 lds_tmp[WPB][64];
  RFID_SHARED float4 lds_q4[WPB][32];
  RFID_SHARED int lds_rec[WPB][LS2_FIN_CHUNK * 128];
  RFID_SHARED int lds_ulist[WPB][LS2_FIN_LIST];
  Ls2Ctl *ctl = a.ctl;
  if (!ls2_fsm_settled(a, ctl)) return;
  if (wv::uniform(ctl->dc_count[a.dc_rounds]) == 0) return;
  const int tid = (int)threadIdx.x, lane = wv::lane_id(), wib = wv::uniform(tid >> 6);
  const int s = (int)blockIdx.y;
  const int G = (int)gridDim.x * WPB;                 // waves of this trace
  const int wid = (int)blockIdx.x * WPB + wib;        // this wave among them
  // windows per explored unit: LS2_FIN_WIN to LS2_FIN_WIN_MAX, one more pair after a turn that ended in a miss (the drift had left
  // the windows: wider ones reach further), one fewer after a turn without (more units per turn) -- configs[2] at sigma = 0.03 / 0.06
  // with 4 / 5 / 6 / 8 throughout: 220 / 276, 170 / 212, 160 / 193, 165 / 183 ms; configs[3] at 0.06: 5.8 / 6.0 / 6.4 / 7.5
  int M = (G >= 16) ? LS2_FIN_WIN : 1;
  const int t0 = s * a.max_bc;
  int *scr = a.fscr + (int64_t)s * 2 * G * LS2_FIN_REC;
  int *bar = a.fbar + s;
  int *ltab = lds_rec[wib];
  int *ulist = lds_ulist[wib];
  // ---- the frontier (every wave finds it for itself: the same data, the same answer) ----
  int first = -1;
  for (int k0 = 0; k0 < a.max_bc && first < 0; k0 += 64) {
    const int k = k0 + lane;
    const int st = (k < a.max_bc) ? a.dstat[t0 + k] : 0;
    const uint64_t m = wv::ballot((st & 4) != 0 && (st & 3) != 3);
    if (m) first = k0 + wv::ffs64(m);
  }
  if (first < 0) return;   // (every wave of the trace leaves: nobody waits at a barrier)
  // its exact start: the end of the settled unit before it (that unit's start is exact and inside its window or margin).  NOT the
  // start the chain wrote for it: that came through the tables of whole blocks, and a block's table may have missed where
  // every unit inside it, walked one by one, was hit
  int Tre = 0, Tim = 0;
  {
    int prev = -1;
    for (int k1 = first; k1 > 0 && prev < 0; k1 -= 64) {   // the slots k1 - 64 .. k1 - 1
      const int k = k1 - 64 + lane;
      const uint64_t m = wv::ballot(k >= 0 && (a.dstat[t0 + ((k >= 0) ? k : 0)] & 4) != 0);
      if (m) prev = k1 - 64 + (63 - (int)__builtin_clzll(m));
    }
    if (prev < 0) { Tre = wv::uniform(a.dcen[2 * (t0 + first)]); Tim = wv::uniform(a.dcen[2 * (t0 + first) + 1]); }   // (the trace's first unit: its centre is the exact start)
    else {
      const int tp = t0 + prev;
      Tre = wv::uniform(a.dT[2 * tp]); Tim = wv::uniform(a.dT[2 * tp + 1]);
      bool exr = true, exi = true;
      ls2_dcb_apply<true>(Tre, exr, a.dtab[(int64_t)(2 * tp) * 64 + lane], wv::uniform(a.dexm[2 * tp]), wv::uniform(a.dcen[2 * tp]), wv::uniform(a.dmar[2 * tp]));
      ls2_dcb_apply<true>(Tim, exi, a.dtab[(int64_t)(2 * tp + 1) * 64 + lane], wv::uniform(a.dexm[2 * tp + 1]), wv::uniform(a.dcen[2 * tp + 1]), wv::uniform(a.dmar[2 * tp + 1]));
      if (!(exr && exi)) { if (lane == 0) ctl->fail = 7; return; }   // (a settled unit's end is exact by definition; every wave sees the same)
    }
  }
  int pos = first;
  int Fre = Tre, Fim = Tim;          // the exact value entering slot `pos` (every wave carries it)
  int nact = (G / M < LS2_FIN_AHEAD) ? (G / M) : LS2_FIN_AHEAD;      // units explored in the next turn
  int fin_t = -1, fin_re = 0, fin_im = 0;                        // the settled unit this wave still has to run in the plain form
  int fixed = 0, turn = 0;
  auto run_final = [&]() {
    if (fin_t < 0) return;
    int er, ei;
    ls2_dcb_unit<false, true, 2>(a, fin_t, true, fin_re, fin_im, false, lane, lds_dc[wib], lds_tmp[wib], reinterpret_cast<float2 *>(lds_q4[wib]), er, ei, nullptr, nullptr, reinterpret_cast<float2 *>(ltab), 2 * LS2_DCB_SNAPS);
    if (lane == 0) { a.dT[2 * fin_t] = fin_re; a.dT[2 * fin_t + 1] = fin_im; a.dstat[fin_t] = 7; }
    fin_t = -1;
  };
  while (pos < a.max_bc) {
    int *set = scr + (int64_t)(turn & 1) * G * LS2_FIN_REC;
    const int uw = wid / M, mw = wid - uw * M;          // the unit (behind the frontier) and the window this wave explores
    const int pc = (M - 1) / 2;                         // the window centred on the prediction itself: ITS wave runs the plain form (see below)
    run_final();                       // (a unit the last turn settled: the others explore meanwhile)
    // the turn's units: the next `nact` slots from pos on that hold one (most slots of the idle grid do not: a turn over SLOTS
    // explored 24 units with 64 slots' waves), every wave for itself from the same flags
    int nu = 0;
    {
      const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      wv::wave_sync();   // (the last turn's reads of the list are over)
      for (int k0 = pos; k0 < a.max_bc && nu < nact; k0 += 256) {
        int st[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { const int kk = k0 + 64 * g + lane; st[g] = (kk < a.max_bc) ? a.dstat[t0 + kk] : 0; }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint64_t m = wv::ballot((st[g] & 4) != 0);
          const int at = nu + wv::popc64(m & lt);
          if ((st[g] & 4) != 0 && at < LS2_FIN_LIST) ulist[at] = k0 + 64 * g + lane;
          nu += wv::popc64(m);
        }
      }
      wv::wave_sync();
      if (nu > nact) nu = nact;
    }
    if (nu == 0) break;                // (no unit left behind the frontier: every wave sees the same)
    const int k = (uw < nu) ? ulist[uw] : 0;
    const int t = t0 + k;
    const bool on = uw < nu;
    // Where a unit is centred: dc_est is the mean of the last 48 closed samples plus the rounding of every addition so far -- a DRIFT
    // that moves by a dozen ulps per unit, whatever the sums do at a binade edge.  The exact value at the frontier gives the drift
    // there (value - ring mean of the frontier's unit); a unit k slots on is centred on ITS ring mean + that drift, off by ~12 sqrt k
    // ulps: inside +- 128 for the whole turn.  (The first version centred on the chain's predictions: where nothing settles by
    // rounds those are hundreds of ulps off, a guess through a missed table is off by a share of that, and the run of hits behind
    // the frontier grew by 4 - 16 units per turn.)
    float dr_re = 0.0f, dr_im = 0.0f;
    const int tq = t0 + ulist[0];      // the frontier's unit
    {
      float fr, fi;
      if (ls2_dcb_ring_mean(a, tq, lane, fr, fi)) { dr_re = ls2_from_ord(Fre) - fr; dr_im = ls2_from_ord(Fim) - fi; }
    }
    if (on) {
      int *rec = set + (int64_t)wid * LS2_FIN_REC;
      {
        // window mw of M: centred on C + 64 (mw - pc) -- C - 96 .. C + 159 in all.  The window around C itself is run in the PLAIN form
        // (table, dc_est at the gate openings, the records the assembly reads -- all left where the rounds leave them): a unit whose
        // start falls into it, as most do, is done with that; the others' windows in the quiet form
        int er, ei, qc[2] = {0, 0}, qm[2] = {0, 0};
        int Cre = Fre, Cim = Fim;
        if (t != tq) { float ur, ui; if (ls2_dcb_ring_mean(a, t, lane, ur, ui)) { Cre = ls2_ord(ur + dr_re); Cim = ls2_ord(ui + dr_im); } }
        const int off = 64 * (mw - pc);
        if (mw == pc) ls2_dcb_unit<false, true, 2>(a, t, true, Cre + off, Cim + off, false, lane, lds_dc[wib], lds_tmp[wib], reinterpret_cast<float2 *>(lds_q4[wib]), er, ei, qc, qm, reinterpret_cast<float2 *>(ltab), 2 * LS2_DCB_SNAPS);   // (the walk's LDS is free while the waves run)
        else ls2_dcb_unit<true, true, 2>(a, t, true, Cre + off, Cim + off, false, lane, lds_dc[wib], lds_tmp[wib], reinterpret_cast<float2 *>(lds_q4[wib]), er, ei, qc, qm);
        rec[8 + lane] = er; rec[72 + lane] = ei;
        if (lane == 0) { rec[0] = qc[0]; rec[1] = qc[1]; rec[2] = qm[0]; rec[3] = qm[1]; }
      }
    }
    // ---- the waves of the trace meet ----
    wv::block_sync();
    if (gridDim.x > 1) wv::grid_meet(bar, (int)gridDim.x * (turn + 1), tid);
    wv::block_sync();
    // ---- through the turn's tables from the exact value: as far as the starts are hit the units are settled; behind the first miss
    //      the walk goes on as a PREDICTION (the missed unit's end continued from its nearest candidate), which is where the
    //      next turn centres the units this turn has already run once -- a round of the chain within the turn's reach.
    //      The records go through LDS 64 at a time, every load in flight at once; their scalars one load per lane ----
    int Wre = Fre, Wim = Fim;         // the walk's value
    int miss = -1;                    // the turn's first slot whose unit was not hit
    bool ex = true;
    int nset = 0, nall = 0;           // units this turn has settled so far: those that need a run in the plain form, all
    int w = 0;
    const int upc = LS2_FIN_CHUNK / M;                 // units per staged chunk
    for (int w0 = 0; w0 < nu; w0 += upc) {
      const int nrec = ((nu - w0 < upc) ? (nu - w0) : upc) * M;
      const int *rb = set + (int64_t)(w0 * M) * LS2_FIN_REC;
      int c_re_l = 0, c_im_l = 0, m_re_l = 0, m_im_l = 0;
      if (lane < nrec) { const int *r = rb + (int64_t)lane * LS2_FIN_REC; c_re_l = r[0]; c_im_l = r[1]; m_re_l = r[2]; m_im_l = r[3]; }
      wv::wave_sync();   // (the last chunk's reads are over)
      {
        Ls2Int4 *dst = reinterpret_cast<Ls2Int4 *>(ltab);
        const int half = lane >> 5, q4 = lane & 31;   // (two records per load instruction: 32 lanes x 16 bytes = a record's two rows)
#pragma unroll 8
        for (int q = 0; q < LS2_FIN_CHUNK / 2; ++q) {
          const int r = 2 * q + half;
          const Ls2Int4 *src = reinterpret_cast<const Ls2Int4 *>(rb + (int64_t)((r < nrec) ? r : 0) * LS2_FIN_REC + 8);
          dst[r * 32 + q4] = src[q4];
        }
      }
      wv::wave_sync();
      for (int u = 0; u < upc && w0 + u < nu; ++u) {
        w = w0 + u;
        const int r0 = u * M;
        const int tw = t0 + ulist[w];
        // the window of the unit that holds the walk's value (else the outermost on that side: its margin may still reach, or a guess)
        const int C_re = wv::readlane(c_re_l, r0 + pc), C_im = wv::readlane(c_im_l, r0 + pc);
        int q_re = (((int)((uint32_t)Wre - (uint32_t)C_re) + 32) >> 6) + pc, q_im = (((int)((uint32_t)Wim - (uint32_t)C_im) + 32) >> 6) + pc;
        q_re = (q_re < 0) ? 0 : ((q_re > M - 1) ? M - 1 : q_re); q_im = (q_im < 0) ? 0 : ((q_im > M - 1) ? M - 1 : q_im);
        const int c_re = wv::readlane(c_re_l, r0 + q_re), c_im = wv::readlane(c_im_l, r0 + q_im);
        const int m_re = wv::readlane(m_re_l, r0 + q_re), m_im = wv::readlane(m_im_l, r0 + q_im);
        if (ex) {
          const int D_re = (int)((uint32_t)Wre - (uint32_t)c_re), D_im = (int)((uint32_t)Wim - (uint32_t)c_im);
          const bool k_re = ls2_dcb_covers(D_re, ~0ull, m_re), k_im = ls2_dcb_covers(D_im, ~0ull, m_im);
          if (k_re && k_im) {
            // settled: its exact start is known.  Inside the plain window: that run's results stand (the wave that ran it says so);
            // else wave G - 1 - (how many such before it this turn) runs the unit in the plain form next turn, centred on its start
            if (q_re == pc && q_im == pc && D_re >= -LS2_DCB_HALF && D_re < LS2_DCB_HALF && D_im >= -LS2_DCB_HALF && D_im < LS2_DCB_HALF) {
              if (wid == w * M + pc && lane == 0) { a.dT[2 * tw] = Wre; a.dT[2 * tw + 1] = Wim; a.dstat[tw] = 7; }
            } else {
              if (wid == G - 1 - nset) { fin_t = tw; fin_re = Wre; fin_im = Wim; }   // (nset < G / M: a wave is asked once per turn)
              nset++;
            }
            nall++;
            fixed++;
          } else {
            ex = false; miss = w;       // this unit's true start is known now: it is the next turn's first unit
            Fre = Wre; Fim = Wim;
            if (wid == 0 && s == 0 && lane == 0) {
              const int d1 = (D_re < 0) ? -D_re : D_re, d2 = (D_im < 0) ? -D_im : D_im;
              ctl->fin_far[(((d1 > d2) ? d1 : d2) < 64 * M) ? 0 : 1]++;
            }
          }
        }
        if (!ex) break;               // (behind the first miss nothing is known: the next turn centres those units anew)
        bool e1 = true, e2 = true;
        ls2_dcb_apply<true>(Wre, e1, ltab[((r0 + q_re) * 2) * 64 + lane], ~0ull, c_re, m_re);
        ls2_dcb_apply<true>(Wim, e2, ltab[((r0 + q_im) * 2 + 1) * 64 + lane], ~0ull, c_im, m_im);
      }
      if (!ex) break;
    }
    if (wid == 0 && s == 0 && lane == 0) {
      int b = 0; for (int v = nall; v > 0 && b < 7; v >>= 1) b++;
      ctl->fin_turns++; ctl->fin_reach[b]++;
    }
    if (ex) {
      // every unit of the turn was hit: the value behind the last one is exact -- the next turn's first start
      pos = ulist[nu - 1] + 1;
      Fre = Wre; Fim = Wim;
    } else {
      pos = ulist[miss];
    }
    // (units of the next turn: twice the reach, and never fewer than LS2_FIN_AHEAD -- the units behind the frontier are run again turn
    // after turn, each time centred on a better prediction, so that they are within a window's reach of the truth when the frontier
    // arrives; the waves that still have a settled unit to run in the plain form are left out when enough others remain)
    {
      int nw = 2 * (ex ? nu : miss) + 4;
      if (nw < LS2_FIN_AHEAD) nw = LS2_FIN_AHEAD;
      if (G >= 16) M = ex ? ((M > LS2_FIN_WIN) ? M - 1 : M) : ((M + 2 < LS2_FIN_WIN_MAX) ? M + 2 : LS2_FIN_WIN_MAX);
      const int amax = G / M;                           // units a turn can explore
      nw = (nw > amax) ? amax : nw;
      if (nw > LS2_FIN_LIST) nw = LS2_FIN_LIST;
      const int spare = (G - nset) / M;
      if (spare >= 1 && nw > spare) nw = spare;
      nact = nw;
    }
    turn++;
  }
  run_final();
  if (wid == 0 && lane == 0 && fixed) { wv::atomic_add(&ctl->dc_count[a.dc_rounds], -fixed); wv::atomic_add(&ctl->dc_finished, fixed); }
}

// ---- 5. windows ----------------------------------------------------------------------------------------------------
RFID_DEVICE bool ls2_all_settled(const Ls2Args &a, const Ls2Ctl *ctl) {
  return ctl->fail == 0 && ctl->avg_count[a.avg_rounds] == 0 && ctl->fsm_count[a.fsm_rounds] == 0 &&
         ctl->dc_count[a.dc_rounds] == 0 && ctl->wb_clash == 0;
}
// the number of complete windows before every unit (exclusive prefix sums: all, EPC), the trace's count, and the trace's
// places in the decoder's two lists.  Workgroups and waves over the slots as in the chain kernels (sums: the pair (v, v)
// composes by addition).
RFID_KERNEL(LS2_CHAIN_THREADS) void ls2_seq_kernel(Ls2Args a) {
  ls2_tail_prio();
  RFID_SHARED Ls2A32 wagg[2 * (LS2_CHAIN_WAVES + 1)];
  const Ls2Ctl *ctl = a.ctl;
  if (!ls2_all_settled(a, ctl)) return;
  const int tid = (int)threadIdx.x, b = (int)blockIdx.x, s = (int)blockIdx.y;
  const int lane = wv::lane_id(), wave = wv::uniform(tid >> 6);
  const int base = s * a.max_b;
  int c_lo, c_hi;
  ls2_chain_range(a.max_bc, a.chain_g, b, wave, c_lo, c_hi);
  int tw = 0, te = 0;
  for (int c = c_lo; c < c_hi; ++c) {
    const int J = 64 * c + lane, i = base + J * LS2_FINE;
    const bool in = J < a.max_bc && a.piece[i].len > 0 && a.fsm[i].head != 0;
    const int nw = in ? a.fsm[i].nwin : 0, ne = in ? a.fsm[i].nepc : 0;
    tw += wv::readlane(wv::scan_add(nw), 63);
    te += wv::readlane(wv::scan_add(ne), 63);
  }
  Ls2A32 carry[2], pre[2];
  carry[0].c0 = carry[0].c1 = tw; carry[1].c0 = carry[1].c1 = te;
  ls2_chain_prefix<2>(a, s, b, wave, lane, tid, carry, wagg, pre);
  int run = pre[0].c0, rune = pre[1].c0;
  for (int c = c_lo; c < c_hi; ++c) {
    const int J = 64 * c + lane, i = base + J * LS2_FINE;
    const bool in = J < a.max_bc && a.piece[i].len > 0 && a.fsm[i].head != 0;
    const int nw = in ? a.fsm[i].nwin : 0, ne = in ? a.fsm[i].nepc : 0;
    const int iw = wv::scan_add(nw), ie = wv::scan_add(ne);
    if (in) { a.seq0[2 * i] = run + iw - nw; a.seq0[2 * i + 1] = rune + ie - ne; }
    run += wv::readlane(iw, 63);
    rune += wv::readlane(ie, 63);
  }
  // the last workgroup's last wave ends on the trace's totals
  if (b == a.chain_g - 1 && wave == LS2_CHAIN_WAVES - 1 && lane == 0) {
    const int total = run, total_e = rune;
    a.wcount[s] = (total < a.wmax) ? total : a.wmax;
    a.flat_base[2 * s] = wv::atomic_add(a.flat_count + 0, total - total_e);
    a.flat_base[2 * s + 1] = wv::atomic_add(a.flat_count + 1, total_e);
    wv::atomic_add(&a.ctl->n_windows, total);
  }
}

// one wave per unit: its windows (in order) -> the trace's window table, dc_est shifted to the unit's true start, and
// the decoder's two lists
RFID_KERNEL(64) void ls2_assemble_kernel(Ls2Args a) {
  ls2_tail_prio();
  Ls2Ctl *ctl = a.ctl;
  if (!(wv::uniform(ctl->fail) == 0 && wv::uniform(ctl->avg_count[a.avg_rounds]) == 0 && wv::uniform(ctl->fsm_count[a.fsm_rounds]) == 0 &&
        wv::uniform(ctl->dc_count[a.dc_rounds]) == 0 && wv::uniform(ctl->wb_clash) == 0)) return;
  const int lane = wv::lane_id();
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (blockIdx.x == 0 && lane == 0) ctl->ok = 1;
  const int NH = a.n_streams * a.max_bc;
  for (int b = (int)blockIdx.x; b < NH; b += (int)gridDim.x) {
    const int i = (b / a.max_bc) * a.max_b + (b % a.max_bc) * LS2_FINE;
    if (wv::uniform(a.piece[i].len) <= 0) continue;
    const Ls2Fsm *f = a.fsm + i;
    if (wv::uniform(f->head) == 0 || wv::uniform(f->nwin) == 0) continue;
    const int pos0 = wv::uniform(a.piece[i].pos0), n = wv::uniform(f->u1) - pos0;
    const int s = i / a.max_b;
    const int gen = wv::uniform(f->gen);
    const Ls2Win *wb = a.wb + (int64_t)s * a.wb_stride;
    const int b0 = pos0 / LS2_WBUCKET, b1 = (pos0 + n - 1) / LS2_WBUCKET;
    int seq = wv::uniform(a.seq0[2 * i]), seq_e = wv::uniform(a.seq0[2 * i + 1]);
    const int fb0 = wv::uniform(a.flat_base[2 * s]), fb1 = wv::uniform(a.flat_base[2 * s + 1]);
    for (int bb = b0; bb <= b1; bb += 64) {
      const int b = bb + lane;
      Ls2Win w; w.start = 0; w.tag = 0; w.slot = 0; w.unit = 0; w.pad_[0] = w.pad_[1] = w.pad_[2] = w.pad_[3] = 0;
      if (b <= b1) w = wb[b];
      const bool on = b <= b1 && (w.tag >> 8) == gen && (w.tag & 2) != 0 && w.start >= pos0 && w.start < pos0 + n;
      const uint64_t m = wv::ballot(on);
      if (m == 0ull) continue;
      const int type = w.tag & 1;
      const uint64_t me = wv::ballot(on && type != 0);
      // dc_est at the opening: the unit's run left it for each of its 64 candidate starts; the chain says which one the true
      // start is (the unit is settled: inside the window), per component
      float dcr = 0.0f, dci = 0.0f;
      if (on) {
        float dc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int cen = a.dcen[2 * w.unit + c];
          const int D = a.dT[2 * w.unit + c] - cen;
          const bool inw = D >= -LS2_DCB_HALF && D < LS2_DCB_HALF && ((a.dexm[2 * w.unit + c] >> ((D + LS2_DCB_HALF) & 63)) & 1ull) != 0ull;
          // an entry that is there: that candidate's own value; else (inside the margin) candidate 32's / 33's run shifted by the even rest
          // of the distance, in ulps of the START's binade (the trajectory is shifted as a whole: rfid_ls2.hpp's header)
          const int par = D & 1;
          const float2 v = a.dcand[(int64_t)w.slot * 64 + (inw ? (D + LS2_DCB_HALF) : (LS2_DCB_HALF + par))];
          const uint32_t e0 = (wv::f2u(ls2_from_ord(cen)) >> 23) & 0xffu;
          const float u0 = (e0 >= 25u) ? wv::u2f((e0 - 23u) << 23) : 0.0f;
          dc[c] = (c ? v.y : v.x) + (inw ? 0.0f : (float)(D - par) * u0);
        }
        dcr = dc[0]; dci = dc[1];
      }
      rfid_window o;
      o.stream = s; o.seq = seq + wv::popc64(m & lt); o.start = w.start; o.type = type;
      o.dc_re = dcr;
      o.dc_im = dci;
      if (on && o.seq < a.wmax) {
        a.wtab[(int64_t)s * a.wmax + o.seq] = o;
        // its place in the decoder's list: the trace's base + the windows of its type before it in the trace
        const int before_e = seq_e + wv::popc64(me & lt);
        const int slotw = type ? (fb1 + before_e) : (fb0 + (o.seq - before_e));
        if (slotw < a.flat_cap) a.flat[(int64_t)type * a.flat_cap + slotw] = o;
      }
      seq += wv::popc64(m);
      seq_e += wv::popc64(me);
    }
  }
}

// streaming: the gate state after each trace's last processed piece -- it ends at an idle cut, so both rings are the
// preceding samples; avg_ampl / dc_est / the state machine from the chains.  One wave per trace.
RFID_KERNEL(64) void ls2_carry_kernel(Ls2Args a) {
  const Ls2Ctl *ctl = a.ctl;
  if (!a.carry_out) return;
  if (!(wv::uniform(ctl->fail) == 0 && wv::uniform(ctl->avg_count[a.avg_rounds]) == 0 && wv::uniform(ctl->fsm_count[a.fsm_rounds]) == 0 &&
        wv::uniform(ctl->dc_count[a.dc_rounds]) == 0 && wv::uniform(ctl->wb_clash) == 0)) return;
  const int lane = wv::lane_id();
  const int s = (int)blockIdx.x;
  const int base = s * a.max_b;
  int last = -1;   // the last piece in use
  for (int j = a.max_b - 1; j >= 0; --j) if (wv::uniform(a.piece[base + j].len) > 0) { last = base + j; break; }
  if (last < 0) return;
  const Ls2Piece pc = a.piece[last];
  const int end = wv::uniform(pc.pos0) + wv::uniform(pc.len);
  GateState *st = a.carry_out + s;
  const float2 *yrow = a.y + (int64_t)s * a.y_stride;
  // avg_ampl after the piece: its true start through its latest run
  const Ls2AvgRun ar = a.arun[last];
  const int Da = wv::uniform(a.aT[last]) - ls2_ord(ar.s);
  const float avg_end = ls2_from_ord(ls2_ord((Da & 1) ? ar.eB : ar.eA) + ((Da & 1) ? (Da - 1) : Da));
  const int h = wv::uniform(a.fsm[last].unit);
  // dc_est behind the last unit: its table at the candidate the chain found to be its true start
  const int tl = s * a.max_bc + (h - base) / LS2_FINE;
  float dc_end[2];
  for (int c = 0; c < 2; ++c) {
    const int D = wv::uniform(a.dT[2 * tl + c]) - wv::uniform(a.dcen[2 * tl + c]);
    const bool inw = D >= -LS2_DCB_HALF && D < LS2_DCB_HALF && ((wv::uniform(a.dexm[2 * tl + c]) >> ((D + LS2_DCB_HALF) & 63)) & 1ull) != 0ull;
    const int par = D & 1;
    const int e = wv::uniform(a.dtab[(int64_t)(2 * tl + c) * 64 + (inw ? (D + LS2_DCB_HALF) : (LS2_DCB_HALF + par))]);
    dc_end[c] = ls2_from_ord(inw ? e : (e + (D - par)));   // (a unit with a margin ends in the binade it starts in)
  }
  for (int k = lane; k < WIN_LEN; k += 64) {
    const int idx = end - WIN_LEN + k;
    float w = 0.0f;
    if (idx >= 0) { const float2 v = yrow[idx]; w = wv::hypot_f(v.x, v.y); }
    st->win[k] = w;
  }
  if (lane < DC_LEN) {
    const int idx = end - DC_LEN + lane;
    const float2 v = (idx >= 0) ? yrow[idx] : make_float2(0.0f, 0.0f);
    st->dcr_re[lane] = v.x; st->dcr_im[lane] = v.y;
  }
  if (lane == 0) {
    const Ls2Fsm &f = a.fsm[h];
    st->avg_ampl = avg_end; st->dc_re = dc_end[0]; st->dc_im = dc_end[1];
    st->n_samples = f.en[0]; st->signal_state = f.en[1]; st->num_pulses = f.en[2]; st->gate_open = f.en[3];
    st->n_to_ungate = f.en[4]; st->wtype = f.en[5];
    st->win_index = 0; st->dc_index = 0;
    st->win_seq = a.wcount[s];   // (a sequential scan that continues from here in the same call appends its windows)
  }
}

}  // namespace rfidk
