// rfid_ls2_enqueue.hpp -- the launch sequence of the long-stream front end (rfid_ls2.hpp): sizes, work space layout and the
// fixed list of launches of one pass.  There is no decision between the launches (the kernels look at Ls2Ctl themselves),
// so the same sequence serves the library (hipLaunchKernelGGL on the context's stream) and the host emulator of the
// test suite; the includer defines
//     LS2_LAUNCH(kernel, grid_x, grid_y, block, args)
// before including this file.
#pragma once
#include <stddef.h>
#ifndef LS2_FIN_WPB        // waves per workgroup of the dc_est finishing walk: 1 on the device (its workgroups meet device-wide); the
#define LS2_FIN_WPB 1      // test suite's emulator runs one workgroup at a time and defines 16 (one workgroup of sixteen waves per trace)
#endif
#ifndef LS2_LAUNCH_FRONT   // (the includer may give the first pass's launch a form of its own)
#define LS2_LAUNCH_FRONT LS2_LAUNCH
#endif
#include <stdint.h>

namespace rfidk {

constexpr int LS2_TARGET_PIECES = 131072;  // pieces per pass to aim for (all traces together)
constexpr int LS2_MIN_PIECE = 512;         // ... of at least this many decimated samples; idle cuts are searched every LS2_FINE pieces
                                           // (a cut needs LS_QUIET = 1615 idle samples before it)

struct Ls2Geometry {
  int P = 0, max_b = 0, NS = 0;            // piece length, slots per trace, slots
  int Pc = 0, max_bc = 0;                  // the idle-cut grid
  int64_t vstride = 0, cstride = 0, wb_stride = 0;
  int n1 = 0, n2 = 0;                      // dc_est chain: blocks of 64 idle-grid slots per trace, groups of 64 blocks
};
// P: nominal piece length for `n_streams` traces of (at most) n_dec decimated samples; 0 = the traces are too short to cut
inline Ls2Geometry ls2_geometry(int n_streams, int64_t n_dec, int min_piece = LS2_MIN_PIECE, int target = LS2_TARGET_PIECES) {
  Ls2Geometry g;
  if (n_streams <= 0 || n_dec < 2 * (int64_t)LS2_FINE * min_piece) return g;
  int64_t P = ((int64_t)n_streams * n_dec + target - 1) / target;
  if (P < min_piece) P = min_piece;
  P = (P + 63) & ~63LL;
  if (P * LS2_FINE > 0x3fffffff) return g;
  g.P = (int)P;
  g.Pc = (int)(P * LS2_FINE);
  g.max_bc = (int)(n_dec / g.Pc) + 1;
  g.max_b = g.max_bc * LS2_FINE;
  g.NS = n_streams * g.max_b;
  g.vstride = (n_dec >> 6) + 3;
  g.cstride = (n_dec >> 6) + g.max_bc + 3;
  g.wb_stride = n_dec / LS2_WBUCKET + 2;
  g.n1 = (g.max_bc + 63) / 64;
  g.n2 = (g.n1 + 63) / 64;
  return g;
}

// work space: one allocation, carved up here (offsets in bytes, 256-byte aligned)
struct Ls2Layout {
  size_t cut, cutf, piece, nextv, prevv, upiece, unextv, uprevv, lb_fn, lb_end, lb_water, votes, closed, openinfo, arun, aT, alist, aover, fsm, wb, dT, dcen, dtab, dstat, dexm, dfront, fscr, fbar, dq, dmar, dwbase, dcand, n1cen, n1tab, n1val, n1ent, n1mar, n1exm, n2cen, n2tab, n2val, n2ent, n2mar, n2exm, seq0, flat_base, cflag, cagg, ctl, consumed, total;
  int dcand_cap;
};
// wmax: complete windows a trace can hold (the caller's window table): sizes the dc_est stage's table of gate openings
// waves per trace of the dc_est finishing walk (its workgroups meet: rfid_ls2.hpp).  The test suite's emulator runs one workgroup at
// a time: one workgroup of LS2_FIN_WPB waves per trace there
inline int ls2_fin_waves(int B) {
  if (LS2_FIN_WPB > 1) return LS2_FIN_WPB;
  int g = LS2_FIN_TOTAL / (B < 1 ? 1 : B);
  if (g > LS2_FIN_GMAX) g = LS2_FIN_GMAX;
  return g < 1 ? 1 : g;
}
inline Ls2Layout ls2_layout(const Ls2Geometry &g, int n_streams, int64_t y_stride, int wmax) {
  Ls2Layout L;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t NS = (size_t)g.NS, B = (size_t)n_streams;
  L.cut = take(sizeof(int) * B * (size_t)g.max_bc);
  L.cutf = take(sizeof(int) * NS);
  L.piece = take(sizeof(Ls2Piece) * NS);
  L.nextv = take(sizeof(int) * NS);
  L.prevv = take(sizeof(int) * NS);
  L.upiece = take(sizeof(Ls2Piece) * NS);
  L.unextv = take(sizeof(int) * NS);
  L.uprevv = take(sizeof(int) * NS);
  L.lb_fn = take(sizeof(uint64_t) * NS);
  L.lb_end = take(sizeof(uint64_t) * NS);
  L.lb_water = take(sizeof(int) * B);
  L.votes = take(sizeof(uint64_t) * 2 * B * (size_t)g.vstride);
  L.closed = take(sizeof(uint64_t) * B * (size_t)g.cstride);
  L.openinfo = take(sizeof(int) * B * (size_t)g.cstride);
  L.arun = take(sizeof(Ls2AvgRun) * NS);
  L.aT = take(sizeof(int) * NS);
  L.alist = take(sizeof(int) * 2 * NS);
  L.aover = take(sizeof(Ls2Aff) * NS);
  L.fsm = take(sizeof(Ls2Fsm) * NS);
  L.wb = take(sizeof(Ls2Win) * B * (size_t)g.wb_stride);
  const size_t NH = B * (size_t)g.max_bc, N1 = B * (size_t)g.n1, N2 = B * (size_t)g.n2;
  L.dT = take(sizeof(int) * 2 * NH);
  L.dcen = take(sizeof(int) * 2 * NH);
  L.dtab = take(sizeof(int) * 2 * 64 * NH);
  L.dstat = take(sizeof(int) * NH);
  L.dmar = take(sizeof(int) * 2 * NH);
  L.dexm = take(sizeof(uint64_t) * 2 * NH);
  L.dfront = take(sizeof(int) * B);
  L.fscr = take(sizeof(int) * B * 2 * (size_t)ls2_fin_waves((int)B) * LS2_FIN_REC);
  L.fbar = take(sizeof(int) * B);
  L.dq = take(sizeof(float2) * B * (size_t)y_stride);
  L.dwbase = take(sizeof(int) * NH);
  L.dcand_cap = (int)(B * (size_t)(wmax > 0 ? wmax : 0) + NH + 8);   // every window + the one a trace may end in, per unit
  L.dcand = take(sizeof(float2) * 64 * (size_t)L.dcand_cap);
  L.n1cen = take(sizeof(int) * 2 * N1); L.n1tab = take(sizeof(int) * 2 * 64 * N1); L.n1val = take(sizeof(int) * N1);
  L.n1ent = take(sizeof(int) * 4 * N1); L.n1mar = take(sizeof(int) * 2 * N1); L.n1exm = take(sizeof(uint64_t) * 2 * N1);
  L.n2cen = take(sizeof(int) * 2 * N2); L.n2tab = take(sizeof(int) * 2 * 64 * N2); L.n2val = take(sizeof(int) * N2);
  L.n2ent = take(sizeof(int) * 4 * N2); L.n2mar = take(sizeof(int) * 2 * N2); L.n2exm = take(sizeof(uint64_t) * 2 * N2);
  L.seq0 = take(sizeof(int) * 2 * NS);
  L.flat_base = take(sizeof(int) * 2 * B);
  L.cflag = take(sizeof(int) * B * LS2_CHAIN_GMAX);
  L.cagg = take(sizeof(int) * 4 * B * LS2_CHAIN_GMAX);
  L.ctl = take(sizeof(Ls2Ctl));
  L.consumed = take(sizeof(int) * B);
  L.total = off;
  return L;
}
inline int &ls2_dcb_top_min();
inline int &ls2_dcb_bias();
inline void ls2_bind(Ls2Args &a, char *base, const Ls2Layout &L, const Ls2Geometry &g) {
  a.P = g.P; a.max_b = g.max_b; a.Pc = g.Pc; a.max_bc = g.max_bc; a.vstride = g.vstride; a.cstride = g.cstride; a.wb_stride = g.wb_stride;
  a.cut = (int *)(base + L.cut); a.cutf = (int *)(base + L.cutf); a.piece = (Ls2Piece *)(base + L.piece);
  a.nextv = (int *)(base + L.nextv); a.prevv = (int *)(base + L.prevv);
  a.upiece = (Ls2Piece *)(base + L.upiece); a.unextv = (int *)(base + L.unextv); a.uprevv = (int *)(base + L.uprevv);
  a.lb_fn = (uint64_t *)(base + L.lb_fn); a.lb_end = (uint64_t *)(base + L.lb_end); a.lb_water = (int *)(base + L.lb_water);
  a.lowm = (uint64_t *)(base + L.closed);   // (fused first pass: the blocks' not-carrier masks live there until the state machine runs)
  a.votes = (uint64_t *)(base + L.votes); a.closed = (uint64_t *)(base + L.closed); a.openinfo = (int *)(base + L.openinfo);
  a.arun = (Ls2AvgRun *)(base + L.arun); a.aT = (int *)(base + L.aT); a.alist = (int *)(base + L.alist); a.aover = (Ls2Aff *)(base + L.aover);
  a.fsm = (Ls2Fsm *)(base + L.fsm); a.wb = (Ls2Win *)(base + L.wb);
  a.dT = (int *)(base + L.dT); a.dcen = (int *)(base + L.dcen); a.dtab = (int *)(base + L.dtab); a.dstat = (int *)(base + L.dstat); a.dmar = (int *)(base + L.dmar); a.dexm = (uint64_t *)(base + L.dexm); a.dfront = (int *)(base + L.dfront); a.fscr = (int *)(base + L.fscr); a.fbar = (int *)(base + L.fbar); a.dq = (float2 *)(base + L.dq);
  a.dwbase = (int *)(base + L.dwbase); a.dcand = (float2 *)(base + L.dcand); a.dcand_cap = L.dcand_cap;
  a.dcb_n1 = g.n1; a.dcb_n2 = g.n2; a.dcb_top = (g.n1 > ls2_dcb_top_min()) ? 2 : 1; a.dcb_bias = ls2_dcb_bias();
  a.n1cen = (int *)(base + L.n1cen); a.n1tab = (int *)(base + L.n1tab); a.n1val = (int *)(base + L.n1val); a.n1ent = (int *)(base + L.n1ent); a.n1mar = (int *)(base + L.n1mar); a.n1exm = (uint64_t *)(base + L.n1exm);
  a.n2cen = (int *)(base + L.n2cen); a.n2tab = (int *)(base + L.n2tab); a.n2val = (int *)(base + L.n2val); a.n2ent = (int *)(base + L.n2ent); a.n2mar = (int *)(base + L.n2mar); a.n2exm = (uint64_t *)(base + L.n2exm);
  a.seq0 = (int *)(base + L.seq0); a.flat_base = (int *)(base + L.flat_base); a.cflag = (int *)(base + L.cflag); a.cagg = (int *)(base + L.cagg); a.ctl = (Ls2Ctl *)(base + L.ctl); a.consumed = (int *)(base + L.consumed);
}

inline int &ls2_fsm_lanes_min() { static int v = 8192; return v; }   // from this many possible heads on the state machine runs one lane per unit (tests: 0 / a huge number)
inline int &ls2_chain_slots() { static int v = 2048; return v; }   // slots per workgroup of a chain launch (tests shrink it)
inline int &ls2_dcb_top_min() { static int v = 64; return v; }   // the dc_est chain walks over groups of blocks when a trace has more blocks than this (tests: 0)
inline int &ls2_dcb_bias() { static int v = 0; return v; }   // (tests: Ls2Args::dcb_bias)

#ifdef LS2_LAUNCH
// One pass (its first launch zeroes Ls2Ctl, the chain flags, the votes, the window buckets and flat_count).  `a` complete but for
// `round`.  After it: wtab / wcount / flat lists + Ls2Ctl::ok = 1, or ok = 0 (the caller's fallback scan, enqueued
// behind with GateArgs::skip_if = &ctl->ok, then runs).
// mark / mark_arg: optional call-back (fused first pass only; point 2: behind the launches that touch nothing but the raw samples,
// y and the pass's own work space -- the library runs those on a second stream, beside the rest of the pass before, and changes
// streams here)
// dc_rounds: dc_est rounds to enqueue (< 0: by the pass's size; the library passes what the passes before needed)
inline void ls2_enqueue(Ls2Args a, bool search_cuts = true, int *rounds_out = nullptr, bool generous = false, int dc_rounds = -1,
                        void (*mark)(void *, int) = nullptr, void *mark_arg = nullptr) {   // (search_cuts = false: a.cut is given -- tests)
  const int NS = a.n_streams * a.max_b, NH = a.n_streams * a.max_bc;   // slots; slots that can be heads
  const int B = a.n_streams;
  const bool fused = a.fused != 0;
  a.fused = fused ? 1 : 0;
  // the fused first pass keeps two piece tables (Ls2Args::fused): the unit kernels get the arguments with the units' table
  // (its slots between the idle cuts: empty, or -- where the dc_est stage wants pieces, dc_fine -- cut at the avg_ampl pieces' starts)
  auto U = [fused](Ls2Args x) {
    if (fused) { x.piece = x.upiece; x.nextv = x.unextv; x.prevv = x.uprevv; x.cutf = nullptr; x.fused = 2; }
    return x;
  };
  {
    const int64_t words = 2 * (int64_t)B * a.vstride + 3 * (int64_t)B * a.wb_stride;
    int64_t g = (words + 4 * 256 - 1) / (4 * 256);   // (four words per thread)
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;
    LS2_LAUNCH(ls2_clear_kernel, (int)g, 1, 256, a);
  }
  if (!fused) {
    // idle cuts on the coarse grid (unless given: tests) and rest points on the fine one
    LsCutArgs ca, cf;
    ca.y = a.y; ca.y_stride = a.y_stride; ca.lens = a.lens; ca.n_dec = a.n_dec; ca.chunk = a.Pc; ca.limit = a.Pc / 2;
    ca.max_b = a.max_bc; ca.cut = a.cut; ca.quiet = LS_QUIET;
    cf = ca;
    cf.chunk = a.P; cf.limit = a.P / 2; cf.max_b = a.max_b; cf.cut = a.cutf; cf.quiet = LS2_REST;
    const bool coarse = a.max_bc > 1 && search_cuts;
    if (coarse && 2 * B <= 65535) {
      LsCut2Args c2;
      c2.a = ca; c2.b = cf; c2.n_streams = B;
      LS2_LAUNCH(ls_cut2_kernel, a.max_b - 1, 2 * B, 64, c2);
    } else {
      if (coarse) LS2_LAUNCH(ls_cut_kernel, a.max_bc - 1, B, 64, ca);
      LS2_LAUNCH(ls_cut_kernel, a.max_b - 1, B, 64, cf);
    }
  }
  a.round = 0;
  a.stamp = 0;
  // rounds enqueued behind each stage's first pass: a round without work is two empty launches (~10 us), which only a
  // short pass notices -- and short traces settle in few rounds (their rounding drift is small).  `generous`: the caller
  // saw a pass run out of rounds (the sequential scan took over): the full count from then on
  const bool small = NS < 32768 && !generous, tiny = NS < 1024 && !generous;
  a.avg_rounds = tiny ? 3 : (small ? 4 : LS2_AVG_ROUNDS); a.fsm_rounds = (tiny || small) ? 1 : LS2_FSM_ROUNDS;
  // dc_est rounds behind the first.  Away from binade edges the first round settles every unit whose margin covers the rounding
  // drift; the few others (a partial sum next to a power of two: ~5 %) are run again centred on the chain's prediction, which is
  // off by an ulp or two per such unit it came through -- the blocks' windows (+- 32) catch that for some thousand units per
  // round: one to three rounds with work for a stream of a few thousand units, three for configs[2]'s 32 000 at sigma = 0.002 and
  // five to six at sigma = 0.06 away from binade edges.  Three / ten are enqueued (a round without work is four / six launches that
  // return at once, 4 us each): the finishing walk takes what the rounds leave FROM THE FIRST UNSETTLED UNIT ON, at ~5 us a unit --
  // a handful of stragglers early in a long trace would cost more than every empty round together.  Sums that hover at a binade
  // edge do not settle by rounds at all (4, 6 or 10 rounds measured alike): there the walk is the way
  a.dc_rounds = (dc_rounds >= 0) ? dc_rounds : (tiny ? 0 : (small ? 3 : LS2_DC_ROUNDS));   // (a look-ahead pass holds a few dozen units: the first round and the walk)
  if (a.dc_rounds < 0) a.dc_rounds = 0;
  if (a.dc_rounds > LS2_DC_MAXR) a.dc_rounds = LS2_DC_MAXR;
  if (fused) {
    // matched filter + piece boundaries + the first avg_ampl pass in one sweep over the raw samples; then the pieces' links,
    // the idle cuts from the blocks' not-carrier masks (unless given: tests) and the units' table from them
    LS2_LAUNCH_FRONT(ls2_front_kernel, 8 * ((NS + 7) / 8), 1, 64, a);
    LS2_LAUNCH(ls2_link_kernel, (NS + 255) / 256, 1, 256, a);
    if (a.max_bc > 1 && search_cuts) LS2_LAUNCH(ls2_idle_cut_kernel, (NH + 255) / 256, 1, 256, a);
    LS2_LAUNCH(ls2_pieces_kernel, (NS + 255) / 256, 1, 256, U(a));
    if (mark) mark(mark_arg, 2);   // (everything up to here works on the raw samples and this pass's work space only)
  } else {
    LS2_LAUNCH(ls2_pieces_kernel, (NS + 255) / 256, 1, 256, a);
  }
  // re-run launches: one wave per list entry, the waves loop when the list is longer than the grid.  The grid used to be the
  // slot count: 130 000 workgroups that return at once cost 28 us per launch, and most of a pass's sixteen re-run launches
  // have little or nothing to do.  (A first avg_ampl round may hold most pieces: below 32 768 workgroups it ran slower.)
  auto rerun_grid = [&](int round, int most) { const int g = (round == 1) ? most : most / 2; return (NS < g) ? NS : g; };
  // workgroups per trace of the chain kernels: a few thousand slots each
  auto chain_g = [](int slots) { const int per = ls2_chain_slots(); int g = (slots + per - 1) / per; return g < 1 ? 1 : (g > LS2_CHAIN_GMAX ? LS2_CHAIN_GMAX : g); };
  const int g_avg = chain_g(a.max_b), g_seq = chain_g(a.max_bc);   // (pieces: any slot; units: idle-grid slots)
  if (!fused) LS2_LAUNCH(ls2_avg_first_kernel, 8 * ((NS + 7) / 8), 1, 64, a);   // (one wave per slot, an eighth of the slots per XCD)
  a.chain_g = g_avg; a.stamp++;
  LS2_LAUNCH(ls2_avg_chain_kernel, g_avg, B, LS2_CHAIN_THREADS, a);
  for (int r = 1; r <= a.avg_rounds; ++r) {
    a.round = r;
    LS2_LAUNCH(ls2_avg_rerun_kernel, rerun_grid(r, 32768), 1, 64, a);
    a.stamp++;
    LS2_LAUNCH(ls2_avg_chain_kernel, g_avg, B, LS2_CHAIN_THREADS, a);
  }
  for (int r = 0; r <= a.fsm_rounds; ++r) {
    a.round = r;
    if (NH >= ls2_fsm_lanes_min()) LS2_LAUNCH(ls2_fsm_lanes_kernel, (NH + LS2_FSM_LANES - 1) / LS2_FSM_LANES, 1, 64, U(a));
    else LS2_LAUNCH(ls2_fsm_kernel, NH, 1, 64, U(a));
    LS2_LAUNCH(ls2_fsm_chain_kernel, (NH + 255) / 256, 1, 256, U(a));
  }
  // dc_est: every unit from 64 neighbouring start values at once (lane = candidate), the chain of their tables in levels of
  // 64 (up, a walk over the top level, down); round r > 0 runs what is not settled again, centred on the chain's prediction
  {
    const int N1 = B * a.dcb_n1, N2 = B * a.dcb_n2;
    for (int r = 0; r <= a.dc_rounds; ++r) {
      a.round = r;
      // (late re-run rounds: the waves loop over the slots -- most of these launches have nothing to do, and 32 000 workgroups that
      // return at once cost 8 us against 4.  The first three keep a wave per slot: a round with work lasts as long as its longest
      // wave, and a unit's run is 50 - 100 us)
      LS2_LAUNCH(ls2_dcb_run_kernel, (r <= 3 || NH < 4096) ? NH : 4096, 1, 64, U(a));
      LS2_LAUNCH(ls2_dcb_up1_kernel, N1, 1, 64, U(a));
      if (a.dcb_top == 2) LS2_LAUNCH(ls2_dcb_up2_kernel, N2, 1, 64, U(a));
      LS2_LAUNCH(ls2_dcb_top_kernel, B, 1, 64, U(a));
      if (a.dcb_top == 2) LS2_LAUNCH(ls2_dcb_down2_kernel, N2, 1, 64, U(a));
      LS2_LAUNCH(ls2_dcb_down1_kernel, N1, 1, 64, U(a));
    }
    a.round = 0;
    {
      // the finishing walk: G waves per trace that meet once per turn -- all of them must be resident at once, so no more than the
      // device holds of them (ls2_fin_waves)
      LS2_LAUNCH(ls2_dcb_incr_kernel, NH, 1, 64, U(a));
      LS2_LAUNCH(ls2_dcb_finish_kernel<LS2_FIN_WPB>, ls2_fin_waves(B) / LS2_FIN_WPB, B, 64 * LS2_FIN_WPB, U(a));
    }
  }
  a.round = 0;
  a.stamp++;
  a.chain_g = g_seq;
  LS2_LAUNCH(ls2_seq_kernel, g_seq, B, LS2_CHAIN_THREADS, U(a));
  LS2_LAUNCH(ls2_assemble_kernel, NH, 1, 64, U(a));
  if (a.carry_out) LS2_LAUNCH(ls2_carry_kernel, B, 1, 64, U(a));
  if (rounds_out) { rounds_out[0] = a.avg_rounds; rounds_out[1] = a.fsm_rounds; rounds_out[2] = a.dc_rounds; }
}
#endif

}  // namespace rfidk
