// rfid_kernels.hpp -- hand-written CDNA4 (gfx950) kernels of the Gen2 receive path.
//
//   mf_boxcar25_decim5_kernel : fir_filter_ccc(5,[1]*25)            (apps/reader.py:65,75)
//   gate_scan_kernel          : gate_impl::general_work scan loop    (lib/gate_impl.cc:127-196)
//   decode_windows_kernel     : tag_sync + RN16 / EPC detection + CRC (lib/tag_decoder_impl.cc:78-193,401-445)
//   stream_stats_kernel       : READER_STATS bookkeeping              (lib/tag_decoder_impl.cc:267-388,
//                                                                      lib/reader_impl.cc:251-344,
//                                                                      lib/gate_impl.cc:101-109)
//
// All paths are relative to /root/reference/gr-rfid/.  None of these is a dense
// contraction: they are HBM-streaming / LDS-gather kernels, so no MFMA.  Arithmetic is
// binary32 in the reference's operation order (compile with -ffp-contract=off), which
// makes decoded bits, indices and scores bit-identical to the CPU path on finite input.
//
// Wave-level operations come from rfid_device_env.h (namespace wv).
#pragma once
#include <rfid_device_env.h>  // resolved through -I: csrc/ for hipcc (the product)
#include "rfid_mi355x.h"

namespace rfidk {

// ---- constants derived from sample_rate = 400 kHz (host verifies the derivation) --------
constexpr int DECIM = 5;           // apps/reader.py:54
constexpr int NTAPS = 25;          // apps/reader.py:65
constexpr int WIN_LEN = 100;       // gate_impl.cc:52  WIN_SIZE_D(250 us) * 0.4
constexpr int DC_LEN = 48;         // gate_impl.cc:53  DC_SIZE_D(120 us) * 0.4
constexpr int T1_SAMPLES = 96;     // gate_impl.cc:48  T1_D(240 us) * 0.4
constexpr int PW_HALF = 2;         // gate_impl.cc:157 n_samples_PW(4) / 2
// While the gate is closed, n_samples (gate_impl.cc:145-180) is only ever compared with PW/2 = 2 and T1 = 96, so the
// scan lets it saturate: every value >= N_SAT behaves like N_SAT.  That makes the state machine's state at an idle
// point a known constant (long-stream front end: the cuts lie where the count has saturated).
constexpr int GATE_N_SAT = 128;
static_assert(GATE_N_SAT > T1_SAMPLES && GATE_N_SAT > PW_HALF, "the saturated count must pass every comparison");
constexpr int NUM_PULSES_CMD = 5;  // global_vars.h:99
constexpr int RN16_WIN = 250;      // gate_impl.cc:121 (17+6)*10 + 2*10
constexpr int EPC_WIN = 1370;      // gate_impl.cc:115 (129+6)*10 + 2*10
constexpr int N_SYNC = 15;         // tag_decoder_impl.cc:85   i < 1.5 * 10
constexpr int N_TCAND = 20;        // tag_decoder_impl.cc:150  number_steps
constexpr float HALF_BIT = 5.0f;   // n_samples_TAG_BIT / 2
constexpr float WIN_LEN_F = 100.0f;
constexpr float DC_LEN_F = 48.0f;
constexpr float THRESH_FRACTION = 0.75f;  // global_vars.h:139

// =========================================================================================
// 1. matched filter: y[n] = sum_{k=0..24} x[5n + in_off + k], k ascending from (0,0).
//    Batch mode in_off = -24 (GNU Radio history: 24 zeros before the stream).
//    One workgroup = MF_TILE outputs of one trace; the 5*MF_TILE+20 raw samples it needs are
//    staged once through LDS with 16-byte coalesced loads; every lane then walks its own
//    25-sample window (lane stride 40 B -> conflict-free ds_read_b64).
// =========================================================================================
constexpr int MF_TILE = 512;
constexpr int MF_THREADS = 256;
constexpr int MF_RAW = MF_TILE * DECIM + (NTAPS - DECIM);  // 2580 raw samples per tile

struct MfArgs {
  const float2 *x;      // [n_streams][x_stride]
  int64_t x_stride;
  int64_t n_raw;        // valid raw samples per trace (when lens == nullptr)
  const int64_t *lens;  // optional per-trace valid raw sample counts
  int64_t n_out;        // outputs per trace (when lens == nullptr)
  int in_off;           // -24 in batch mode
  int vec_ok;           // rows 16-byte aligned and in_off even -> float4 loads
  float2 *y;            // [n_streams][y_stride]
  int64_t y_stride;
  int64_t tile0;        // first output tile of this launch (time-chunked launches)
  int stream0;          // first trace of this launch (gridDim.y <= 65535 traces per launch)
};

RFID_DEVICE void mf_tile(const MfArgs &a, const int b, const int64_t tile_idx, float4 *tile4) {
  float2 *tile = reinterpret_cast<float2 *>(tile4);
  const int tid = (int)threadIdx.x;
  int64_t n_raw = a.n_raw, n_out = a.n_out;
  if (a.lens) {
    n_raw = a.lens[b];
    if (n_raw > a.n_raw) n_raw = a.n_raw;
    if (n_raw < 0) n_raw = 0;
    n_out = n_raw / DECIM;
  }
  const int64_t n0 = tile_idx * MF_TILE;
  if (n0 >= n_out) return;
  const float2 *xs = a.x + (int64_t)b * a.x_stride;
  const int64_t r0 = n0 * DECIM + a.in_off;
  if (a.vec_ok) {
    for (int j = tid; j < MF_RAW / 2; j += MF_THREADS) {
      const int64_t r = r0 + 2 * j;
      float4 v;
      if (r >= 0 && r + 1 < n_raw) {
        v = *reinterpret_cast<const float4 *>(xs + r);
      } else {
        float2 lo = (r >= 0 && r < n_raw) ? xs[r] : make_float2(0.0f, 0.0f);
        float2 hi = (r + 1 >= 0 && r + 1 < n_raw) ? xs[r + 1] : make_float2(0.0f, 0.0f);
        v = make_float4(lo.x, lo.y, hi.x, hi.y);
      }
      tile4[j] = v;
    }
  } else {
    for (int j = tid; j < MF_RAW; j += MF_THREADS) {
      const int64_t r = r0 + j;
      tile[j] = (r >= 0 && r < n_raw) ? xs[r] : make_float2(0.0f, 0.0f);
    }
  }
  wv::block_sync();
  float2 *ys = a.y + (int64_t)b * a.y_stride;
#pragma unroll
  for (int rep = 0; rep < MF_TILE / MF_THREADS; ++rep) {
    const int o = tid + rep * MF_THREADS;
    const int64_t n = n0 + o;
    float re = 0.0f, im = 0.0f;
#pragma unroll
    for (int k = 0; k < NTAPS; ++k) {
      const float2 v = tile[DECIM * o + k];
      re = re + v.x;
      im = im + v.y;
    }
    if (n < n_out) ys[n] = make_float2(re, im);
  }
}
RFID_KERNEL(MF_THREADS) void mf_boxcar25_decim5_kernel(MfArgs a) {
  RFID_SHARED float4 tile4[MF_RAW / 2 + 2];
  mf_tile(a, (int)blockIdx.y + a.stream0, (int64_t)blockIdx.x + a.tile0, tile4);
}
// Look-ahead keyed on the gate: a call's new samples (the foreign filter's outputs, staged in page-locked memory) fetched over the
// bus by a launch -- cheaper for the calling thread than setting up a transfer of a few kilobytes (every thread's loads first)
struct UploadArgs { const float2 *src; float2 *dst; int n; };
RFID_KERNEL(256) void upload_kernel(UploadArgs a) {
  constexpr int PER = 4;
  const int base = (int)blockIdx.x * 256 * PER + (int)threadIdx.x;
  float2 v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = base + u * 256;
    v[u] = (i < a.n) ? a.src[i] : make_float2(0.0f, 0.0f);
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = base + u * 256;
    if (i < a.n) a.dst[i] = v[u];
  }
}
// Look-ahead, one rfid_mf_work call: the call's new raw samples come straight out of page-locked host memory -- no copy
// engine in front of the filter, whose hand-over to the kernel queue was half of the call's way through the device.  Every
// workgroup stages the window of its tile (the samples in front of the call from the device buffer, the new ones over the
// bus), puts the new ones where the passes expect them, filters (mf_tile's sums) and writes its outputs to page-locked
// memory; the last workgroup done tells the spinning host.
struct MfUploadArgs {
  const float2 *src;    // page-locked host memory: the call's n_new raw samples
  float2 *x;            // device: x[0 .. hist) the samples in front of the call (there), x[hist + i] <- src[i]
  int hist, n_new;
  int n_out, in_off;    // output n = sum x[5 n + in_off .. + 24], 0 <= in_off < 5
  float2 *y;            // page-locked host memory: the n_out outputs
  int *done;            // device counter of the workgroups through (0 between launches)
  int *flag, seq;       // page-locked word <- seq when everything is written
};
RFID_KERNEL(MF_THREADS) void mf_upload_kernel(MfUploadArgs a) {
  RFID_SHARED float4 tile4[MF_RAW / 2 + 2];
  float2 *tile = reinterpret_cast<float2 *>(tile4);
  const int tid = (int)threadIdx.x, t = (int)blockIdx.x, T = (int)gridDim.x;
  const int n_raw = a.hist + a.n_new;
  const int r0 = t * (MF_TILE * DECIM) + a.in_off;
  // the windows of neighbouring tiles share NTAPS - DECIM samples: a tile stores what its window has behind those (tile 0: all new ones)
  const int c_lo = (t == 0) ? a.hist : r0 + (NTAPS - DECIM);
  // (all of a thread's loads first, then its stores: a load behind a store to memory the compiler cannot tell apart from the
  // source waits for that store's data -- eleven trips over the bus one after the other, 19 us per launch instead of 6)
  constexpr int PER = (MF_RAW + MF_THREADS - 1) / MF_THREADS;
  float2 v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int j = tid + u * MF_THREADS, r = r0 + j;
    v[u] = make_float2(0.0f, 0.0f);
    if (j < MF_RAW && r < n_raw) v[u] = (r < a.hist) ? a.x[r] : a.src[r - a.hist];
  }
  float2 extra = make_float2(0.0f, 0.0f);
  const int r_extra = r0 + MF_RAW + tid;   // (behind the last window: the samples that wait for their group of five to complete)
  const bool has_extra = t == T - 1 && r_extra < n_raw;
  if (has_extra) extra = a.src[r_extra - a.hist];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int j = tid + u * MF_THREADS, r = r0 + j;
    if (j < MF_RAW) {
      tile[j] = v[u];
      if (r >= c_lo && r >= a.hist && r < n_raw) a.x[r] = v[u];
    }
  }
  if (has_extra) a.x[r_extra] = extra;
  wv::block_sync();
#pragma unroll
  for (int rep = 0; rep < MF_TILE / MF_THREADS; ++rep) {
    const int o = tid + rep * MF_THREADS;
    const int n = t * MF_TILE + o;
    float re = 0.0f, im = 0.0f;
#pragma unroll
    for (int k = 0; k < NTAPS; ++k) {
      const float2 v = tile[DECIM * o + k];
      re = re + v.x;
      im = im + v.y;
    }
    if (n < a.n_out) a.y[n] = make_float2(re, im);
  }
  wv::system_release_fence();
  wv::block_sync();
  if (tid == 0) {
    if (wv::atomic_add(a.done, 1) == T - 1) {
      *a.done = 0;
      wv::store_i32_system_release(a.flag, a.seq);
    }
  }
}
// The same behind a pass that normally needs no filter launch (the long-stream front end with its fused first pass filters
// inside ls2_front_kernel; when it gives the pass up, y is incomplete and the sequential scan behind it needs all of it):
// does nothing when *skip_if != 0 -- a fixed, small grid whose workgroups walk the tiles, so that the usual, skipped case
// costs a few microseconds whatever the length of the traces.
struct MfFallbackArgs { MfArgs m; const int *skip_if; int64_t n_tiles; };
RFID_KERNEL(MF_THREADS) void mf_fallback_kernel(MfFallbackArgs f) {
  RFID_SHARED float4 tile4[MF_RAW / 2 + 2];
  if (f.skip_if && *f.skip_if != 0) return;
  for (int64_t t = blockIdx.x; t < f.n_tiles; t += gridDim.x) {
    mf_tile(f.m, (int)blockIdx.y + f.m.stream0, t, tile4);
    wv::block_sync();   // (the tile is written again)
  }
}

// =========================================================================================
// 2. gate scan (+ fused matched filter).  64 decimated samples per step (lane L <-> sample
//    pos+L), four pipelined wavefronts per trace.  The reference's per-sample recurrences are
//    kept bit-exact:
//      * avg_ampl += (|x| - ring[w]) / 100     -> the increments are computed lane-parallel
//        (producer wave), then added IN ORDER in the consumer wave (lane L ends up holding the
//        value after sample L): as one integer prefix sum where that is provably the same
//        binary32 arithmetic, else by a 63-step DPP wave-shift chain (chain_add_auto);
//      * dc_est  += (x - dcring[d]) / 48       -> two such chains in the back wave, only over
//        "closed" samples;
//      * the edge / pulse-count / window state machine runs on the scalar unit of the consumer
//        wave over 64-bit vote masks, loop-free.
//    The filter wave feeds the samples: matched-filter output y (stage kernel) or the raw 2 Msps
//    samples, filtered on the fly (fused front end; y is then written for the decoder).
//    Output: one rfid_window {start, dc_est at opening, type} per gate opening.  Batch mode
//    re-arms the gate itself (type alternates RN16, EPC, ... : SURVEY.md section 3.3);
//    streaming mode stops right after a window closes, like gate_impl.cc:189-194.
// =========================================================================================
struct GateState {  // gate_impl members (gate_impl.h:36-44) + the READER_STATE fields the gate owns
  float avg_ampl;
  float dc_re, dc_im;
  int n_samples;
  int signal_state;  // 0 NEG_EDGE, 1 POS_EDGE
  int num_pulses;
  int gate_open;
  int n_to_ungate;
  int wtype;        // type of the window being sought / open: 0 RN16, 1 EPC
  int win_index;
  int dc_index;
  int win_seq;      // windows opened so far
  float win[WIN_LEN];
  float dcr_re[DC_LEN];
  float dcr_im[DC_LEN];
};

struct GateArgs {
  const float2 *y;      // [n_streams][y_stride] matched-filter output
  int64_t y_stride;
  int64_t n_dec;        // valid decimated samples per trace (when lens == nullptr)
  int64_t pos0;         // time-chunked launches: this launch scans samples [pos0, pos0 + chunk_len)
  int64_t chunk_len;    //   of every trace, carrying the gate state from the previous chunk
  const int64_t *lens;  // optional per-trace RAW lengths (n_dec = lens/5)
  GateState *state;     // [n_streams]
  int n_streams;
  rfid_window *wtab;    // [n_streams][wmax]
  int wmax;
  int *wcount;          // [n_streams] complete windows recorded
  rfid_window *flat;    // compact lists over all traces (arbitrary order) for the decoder:
  int *flat_count;      //   flat[0 .. flat_cap) RN16 windows, flat[flat_cap .. 2*flat_cap) EPC windows,
  int flat_cap;         //   flat_count[0] / flat_count[1] their device counters
  int mode;             // 0 batch (self re-arming), 1 streaming (stop after close)
  float2 *gated;        // streaming: gated, DC-removed samples
  int gated_cap;
  int *io;              // streaming: io[0] = consumed, io[1] = written
  // fused front end (front_end_fused_kernel): the filter wave computes the matched filter
  // itself from the raw 2 Msps samples and writes y (for the decoder) instead of reading it
  const float2 *raw;    // [n_streams][raw_stride]
  int64_t raw_stride;
  int64_t n_raw;        // valid raw samples per trace (when lens == nullptr)
  int raw_vec_ok;       // rows 16-byte aligned -> float4 loads
  float2 *y_w;          // [n_streams][y_stride], written
  // optional: the launch does nothing when *skip_if != 0 (the sequential scan enqueued behind the long-stream front end
  // as its fallback: it only runs when that front end gave up)
  const int *skip_if;
};

// x / C for the gate's two constant divisors (100: gate_impl.cc:131, 48: :141) in three instructions
// instead of the ~11 of a generic correctly rounded division:
//     q1 = x * RN(1/C);   q = fma(fma(-q1, C, x), RN(1/C), q1)
// equals RN(x / C) bit for bit for every finite |x| >= 2^-120 -- checked exhaustively over all
// 2^31 magnitudes on the host (tests/tools/divcheck.c) and sampled on the device by rfid_selftest().
// div_const_ok() admits |x| in [2^-100, inf) and +0; anything else (tiny, -0, inf, NaN) sends the
// whole wave through wv::fdiv.
RFID_DEVICE bool div_const_ok(float x) {
  const uint32_t u = wv::f2u(x);
  return ((u & 0x7fffffffu) - 0x0d800000u) < (0x7f800000u - 0x0d800000u) || u == 0u;
}
template <int C>
RFID_DEVICE float div_const_fast(float x) {
  constexpr float c = (float)C, rc = 1.0f / (float)C;
  const float q1 = x * rc;
  return wv::fma_f(wv::fma_f(-q1, c, x), rc, q1);
}
// the lanes whose operand div_const_fast must not see, as a vote mask.  (A vote over a conjunction of compares costs two
// extra vector instructions -- the compiler materialises the lane predicate and compares it again; votes over single
// compares, combined on the scalar unit, cost none.)
RFID_DEVICE uint64_t div_const_bad(float x) {
  const uint32_t u = wv::f2u(x);
  return wv::ballot(((u & 0x7fffffffu) - 0x0d800000u) >= (0x7f800000u - 0x0d800000u)) & wv::ballot(u != 0u);
}
template <int C>
RFID_DEVICE float div_const(float x) {   // wave-uniform choice of the path
  if (__builtin_expect(div_const_bad(x) == 0, 1)) return div_const_fast<C>(x);
  return wv::fdiv(x, (float)C);
}

// In-order sum: returns in lane L the value  (((carry + x_0) + x_1) + ...) + x_L.
RFID_DEVICE float chain_add(float carry, float x, int lane) {
  const float x0 = (lane == 0) ? (carry + x) : x;
  float p = x0;
#pragma unroll
  for (int s = 1; s < 64; ++s) p = wv::shr1(p) + x0;
  return p;
}

// The same in-order sum without the 63-deep chain.  While every partial sum stays in the binade of the carry
// s_0 = S_0 * u (u = its ulp, S_0 in [2^23, 2^24)), binary32 addition is integer arithmetic on the mantissa:
//     RN(S*u + d) = (S + RNE(d/u)) * u,
// and the rounding of d/u does not depend on S -- unless d/u lies exactly half-way between two integers
// I and I+1: ties-to-even then picks the one that makes the SUM even, i.e. adds I + ((S + I) & 1).  Such ties
// are frequent here: the addends are differences of binary32 samples (multiples of the sample's ulp) divided
// by 100 or 48, so d/u = K/100 hits x.5 once in a hundred samples.  They need only the PARITY of S, and the
// parity sequence is cheap: a tie leaves an even sum (parity 0), anything else toggles the parity by RNE(d/u)&1
// -- a prefix XOR with resets, done on the scalar unit over 64-bit vote masks (the "state before each sample"
// is a carry chain, generate = tie & Q, kill = tie & ~Q, as in the gate's edge state machine).  Then
//     s_{L+1} = as_float(as_int(s_0) + sum_{j<=L} R_j)
// is the reference's value bit for bit (one integer prefix sum, 6 DPP adds), PROVIDED that
//   (1) all |d_j/u| < 2^22 (exact conversion),
//   (2) every partial sum keeps the exponent of s_0 and a non-zero mantissa: then each exact sum lies strictly
//       inside the binade, where the spacing of binary32 numbers is u (a result that rounds onto a binade
//       edge, or crosses it, is rounded on a different grid -- not this arithmetic).
// Both are checked on all 64 lanes (one vote); a step that fails -- reader commands moving avg_ampl / dc_est
// across a power of two, the first samples of a trace -- takes the exact chain above.
// (carry < 0: the same on magnitudes with d negated -- rounding to nearest-even is symmetric.)
RFID_DEVICE uint64_t prefix_xor64(uint64_t x) {   // bit k = x_0 ^ ... ^ x_k
  x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32;
  return x;
}
RFID_DEVICE bool chain_add_scan(float carry, float x, int lane, float &out) {
  const uint32_t cb = wv::f2u(carry);
  const uint32_t e_b = (cb >> 23) & 0xffu;                        // biased exponent of the carry (wave-uniform)
  const uint32_t sign = cb & 0x80000000u;
  // 2^(23 - e) as a float: biased exponent 277 - e_b, representable for e_b in [23, 254]
  const float scale = wv::u2f(((277u - e_b) & 0xffu) << 23);
  const float t = sign ? -(x * scale) : (x * scale);              // d / u (exact: a power-of-two scaling)
  const float r = wv::rint_f(t);                                  // to nearest, ties to even
  const float frac = t - r;                                       // exact
  const bool bad_t = !(__builtin_fabsf(t) < 4194304.0f);
  const bool half = __builtin_fabsf(frac) == 0.5f;
  const bool tie = !bad_t && half;
  int R = wv::f2i(bad_t ? 0.0f : r);
  // (votes over single compares, combined on the scalar unit: see div_const_bad)
  const uint64_t badmask_t = wv::ballot(bad_t);
  const uint64_t tiemask = wv::ballot(half) & ~badmask_t;
  if (tiemask != 0ull) {
    // ties: d/u = I + 1/2 with I = floor(d/u) = r - (frac < 0); the sum takes I + ((S + I) & 1)
    const int I = R - ((frac < 0.0f) ? 1 : 0);
    const uint64_t rodd = wv::ballot((R & 1) != 0) & ~tiemask;    // parity toggles of the other samples
    const uint64_t iodd = wv::ballot((I & 1) != 0) & tiemask;
    const uint64_t Q = prefix_xor64(rodd);                        // toggles accumulated up to and including sample k
    // parity of S before sample k = (toggles up to k-1) ^ (toggles up to the last tie before k, or the carry's
    // parity if there is none): the second term is a fill-forward of Q from the tie positions = the carry
    // into bit k of a binary addition with generate = tie & Q, kill = tie & ~Q, carry-in = parity of S_0
    const uint64_t gen = tiemask & Q, X = gen | ~tiemask;
    const uint64_t sum = X + gen + (uint64_t)(cb & 1u);
    const uint64_t W = X ^ gen ^ sum;
    const uint64_t par = (Q << 1) ^ W;                            // bit k = parity of S_k
    const uint64_t up = (par ^ iodd) & tiemask;                   // ties that take I + 1
    R = tie ? (I + (int)((up >> lane) & 1ull)) : R;
  }
  const uint32_t mag = (cb & 0x7fffffffu) + (uint32_t)wv::scan_add(R);
  const uint64_t badmask_s = wv::ballot(((mag ^ cb) & 0x7f800000u) != 0u) | wv::ballot((mag & 0x007fffffu) == 0u);
  const bool bad_c = e_b < 23u || e_b > 254u;   // (wave-uniform)
  out = wv::u2f(mag | sign);
  return (badmask_t | badmask_s) == 0ull && !bad_c;
}

// in-order sum, scan when it is provably exact, else the chain
RFID_DEVICE float chain_add_auto(float carry, float x, int lane) {
  float v;
  if (__builtin_expect(chain_add_scan(carry, x, lane, v), 1)) return v;
  return chain_add(carry, x, lane);
}

RFID_DEVICE uint64_t lane_range(int lo, int hi) {  // bits [lo, hi), 0 <= lo <= hi <= 64
  const uint64_t up = (hi >= 64) ? ~0ull : ((1ull << hi) - 1ull);
  const uint64_t dn = (lo >= 64) ? ~0ull : ((1ull << lo) - 1ull);
  return up & ~dn;
}

// two chains (dc_est real / imaginary) advanced together: a dependent DPP add has ~12 cycles of latency but a
// single wave can issue one every ~4, so two chains cost little more than one
RFID_DEVICE void chain_add2(float cb, float xb, float cc, float xc, int lane, float &pb, float &pc) {
  const float b0 = (lane == 0) ? (cb + xb) : xb;
  const float c0 = (lane == 0) ? (cc + xc) : xc;
  pb = b0; pc = c0;
#pragma unroll
  for (int s = 1; s < 64; ++s) {
    pb = wv::shr1(pb) + b0;
    pc = wv::shr1(pc) + c0;
  }
}

// wave-uniform registers of one trace's state machine (consumer wave)
struct GateRegs {
  float avg_c;
  int f_n, f_state, f_pulses, f_open, f_ung, f_type;
  int consumed;
  bool stop;
};

// wave-uniform registers of the back wave: dc_est and the window records
struct GateBackRegs {
  float dcr_c, dci_c;
  int dc_index;
  int run_closed;   // closed samples seen back-to-back up to the current position (this call)
  int ring_stale;   // the dc ring in LDS is not maintained while whole steps are closed; rebuilt on demand from prev_yv
  float2 prev_yv;   // the samples of the last step with closed samples
  int win_seq, n_complete, written;
  int pos0, strm;   // first sample of this launch's chunk (or unit) within the trace; the trace index
};

// One step (64 decimated samples) on its way through the waves of a trace:
//   filter (yv) -> producer (amp, d, tre, tim) -> consumer (avg_ampl, threshold votes, state machine; b_*) ->
//   back (dc ring, dc_est, window records; frees the slot)
struct GateSlot {
  float amp[64];   // |x|                                   (gate_impl.cc:130)    producer -> consumer
  float d[64];     // (|x| - win_samples[win_index]) / 100   (gate_impl.cc:131)    producer -> consumer
  // consumer -> back (16-byte records, written by lane 0; the second one in streaming mode only)
  alignas(16) int b_word;                   //   bit 0 some sample of the step is "closed" (updates dc_est), bit 1 the scan stops after this
  int b_pad0_;                              //   step, bit 2 some sample lies inside an open window; bits 8..15 samples of the step that
  uint64_t b_closedmask;                    //   were consumed; bits 16..23 lane of the gate opening of the step (0xff: none), bit 24 its type
                                            //   b_closedmask: the closed samples (gate_impl.cc:139-143)
  uint64_t b_openmask;                      //   lanes inside an open window (streaming: their gated samples are emitted)
  int b_pad1_[2];
  float2 yv[64];   // the samples themselves
  float tre[64];   // dc_est increments of the step's samples: the filter wave's speculative (x - x[i-48]) / 48 -- exact
  float tim[64];   //   whenever the previous 48 samples were all "closed" (gate_impl.cc:141) -- else the back wave recomputes them
  int spec_bad;    // != 0: some difference of the step is outside the range of the 3-instruction constant division -- the
  int pad_[3];     //   speculative increments are not to be used (the back wave forms them from the ring, with wv::fdiv)
};

// ---- producer wave: everything that is lane-parallel ---------------------------------------
// speculative dc_est increments of one step, formed by the FILTER wave (the role with time to spare: it waits for a free
// slot a third of the time): (x_i - x_{i-48}) / 48 with x_{i-48} from the previous step (lanes 0..47 <- its lanes 16..63)
// or from this one (lanes 48..63 <- lanes 0..15).  Branch-free -- a branch here would make the compiler wait for ALL
// of the filter wave's raw-sample loads in flight: the constant division always takes its 3-instruction form, and a
// vote tells the back wave when that form was not valid for some lane (then it does not use these values).
RFID_DEVICE void gate_spec_dc(GateSlot &slot, float2 yv, float2 &prev_yv, int lane) {
  const int src = (lane < DC_LEN) ? (lane + 64 - DC_LEN) : (lane - DC_LEN);
  const float pre = wv::shfl(prev_yv.x, src), pim = wv::shfl(prev_yv.y, src);
  const float cre = wv::shfl(yv.x, src), cim = wv::shfl(yv.y, src);
  const float ore = (lane < DC_LEN) ? pre : cre, oim = (lane < DC_LEN) ? pim : cim;
  const float nr = yv.x - ore, ni = yv.y - oim;
  slot.tre[lane] = div_const_fast<DC_LEN>(nr);
  slot.tim[lane] = div_const_fast<DC_LEN>(ni);
  slot.spec_bad = ((div_const_bad(nr) | div_const_bad(ni)) != 0ull) ? 1 : 0;   // (every lane stores the same word)
  prev_yv = yv;
}

RFID_DEVICE void gate_produce(GateSlot &slot, float2 yv_in, int pos, int n, int lane,
                              float *lds_win, int &win_index) {
  const int nvalid = (n - pos < 64) ? (n - pos) : 64;
  const bool valid = lane < nvalid;
  const float2 yv = valid ? yv_in : make_float2(0.0f, 0.0f);
  const float amp = wv::hypot_f(yv.x, yv.y);
  int wi = win_index + lane;
  if (wi >= WIN_LEN) wi -= WIN_LEN;
  const float amp_old = lds_win[wi];
  const float nd = valid ? (amp - amp_old) : 0.0f;
  wv::wave_sync();
  if (valid) lds_win[wi] = amp;
  wv::wave_sync();
  win_index += nvalid;
  if (win_index >= WIN_LEN) win_index -= WIN_LEN;
  // the division by a constant (gate_impl.cc:131), one wave-uniform choice of the path
  const float d = div_const<WIN_LEN>(nd);
  slot.amp[lane] = amp;
  slot.d[lane] = d;
  // (slot.yv, the speculative dc_est increments: written by the filter wave; lanes past the end of the call hold zeros there)
}

// a gate opening at lane `ol` of the step that starts at `pos`: dc_est is the in-order sum at
// that lane (the opening sample itself is still a "closed" sample, gate_impl.cc:141-176)
RFID_DEVICE void gate_record_window(const GateArgs &a, GateBackRegs &g, int ol, int wtype, int pos, int n, int s,
                                    int lane, float dcr, float dci) {
  const float odr = wv::readlane(dcr, ol), odi = wv::readlane(dci, ol);
  const int wlen = wtype ? EPC_WIN : RN16_WIN;
  const int start = g.pos0 + pos + ol;
  if (a.mode == 0 && start + wlen <= n) {  // only complete windows reach the decoder (:223,:291)
    if (lane == 0 && g.win_seq < a.wmax) {
      rfid_window w;
      w.stream = g.strm; w.seq = g.win_seq; w.start = start; w.type = wtype;
      w.dc_re = odr; w.dc_im = odi;
      // (one store, not waited for.  The decoder's compact lists are filled from these records when the scan of the trace
      // is over -- gate_flat_lists: a list slot drawn per opening is a returning atomic in the back wave's path, ~5 000
      // cycles twice per inventory round when a thousand traces draw from the two counters at once)
      a.wtab[(int64_t)s * a.wmax + g.win_seq] = w;
    }
    g.n_complete++;
  }
  g.win_seq++;
}

// The windows [first, first + cnt) of this trace's row of the window table, just written by this wave, go into the
// decoder's two compact lists (by type; any order): one slot draw per type and trace.
RFID_DEVICE void gate_flat_lists(const GateArgs &a, int s, int first, int cnt, int lane) {
  if (!a.flat || cnt <= 0) return;
  wv::global_release();   // this wave's records have arrived
  const rfid_window *row = a.wtab + (int64_t)s * a.wmax;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int n1 = 0;
  for (int j0 = 0; j0 < cnt; j0 += 64) {
    const int j = j0 + lane;
    const int ty = (j < cnt) ? wv::load_coherent_i32(&row[first + j].type) : 0;
    n1 += wv::popc64(wv::ballot(ty != 0));
  }
  int base = 0;
  if (lane < 2) {
    const int want = lane ? n1 : (cnt - n1);
    if (want > 0) base = wv::atomic_add(a.flat_count + lane, want);
  }
  const int base0 = wv::readlane(base, 0), base1 = wv::readlane(base, 1);
  int done0 = 0, done1 = 0;
  for (int j0 = 0; j0 < cnt; j0 += 64) {
    const int j = j0 + lane;
    rfid_window w;
    w.type = 0;
    if (j < cnt) w = wv::load_coherent_window(&row[first + j]);
    const uint64_t m1 = wv::ballot(j < cnt && w.type != 0), m0 = wv::ballot(j < cnt && w.type == 0);
    if (j < cnt) {
      const int slotw = w.type ? (base1 + done1 + wv::popc64(m1 & lt)) : (base0 + done0 + wv::popc64(m0 & lt));
      if (slotw < a.flat_cap) a.flat[(int64_t)(w.type ? 1 : 0) * a.flat_cap + slotw] = w;
    }
    done0 += wv::popc64(m0); done1 += wv::popc64(m1);
  }
}

// One step (64 samples, the first `nvalid` of them valid) through the edge / pulse / window state machine of
// gate_impl.cc:145-195, on the scalar unit over the two threshold-vote masks.  Out: which samples are "closed" (update
// dc_est), which lie inside an open window, and the gate opening of the step if there is one.  Shared by the consumer
// wave of the gate scan and by the long-stream front end's state-machine pass (rfid_ls2.hpp).
RFID_DEVICE void gate_fsm_step(const int mode, GateRegs &g, const uint64_t below, const uint64_t above, const int pos, int &nvalid,
                               uint64_t &closedmask, uint64_t &openmask, int &open_lane, int &open_type) {
  const bool plain_open = (nvalid == 64) && g.f_open && (g.f_ung - g.f_n > 64);
  const bool plain_closed = (nvalid == 64) && !g.f_open && (g.f_state == 1) && (below == 0) &&
                            !((g.f_pulses > NUM_PULSES_CMD) && (T1_SAMPLES - g.f_n < 64));
  // what the back wave gets for this step
  closedmask = 0; openmask = 0;
  open_lane = 0xff; open_type = 0;   // (at most one opening per step: a window is longer than a step)

  // The two by far most frequent kinds of step are decided with a handful of scalar
  // instructions (the consumer wave is issue bound: every instruction costs ~4.5 cycles):
  //   (A) the whole step lies inside an open window that does not end in it;
  //   (B) gate closed, POS_EDGE, no sample below the threshold, no opening due.
  if (__builtin_expect(plain_open || plain_closed, 1)) {
    g.f_n += 64;
    closedmask = plain_closed ? ~0ull : 0ull;
    openmask = plain_closed ? 0ull : ~0ull;
  } else {
    // edge / pulse / window state machine on the scalar unit, event driven (gate_impl.cc:145-195)
    int f_n = g.f_n, f_state = g.f_state, f_pulses = g.f_pulses, f_open = g.f_open;
    int f_ung = g.f_ung, f_type = g.f_type;
    int p = 0;
    // (1) a window is open at the start of the step (gate_impl.cc:182-195)
    if (f_open) {
      int take = f_ung - f_n;
      if (take > nvalid) take = nvalid;
      if (take < 0) take = 0;
      openmask = lane_range(0, take);
      f_n += take;
      p = take;
      if (f_n >= f_ung) {  // gate_impl.cc:189-194
        f_open = 0;
        if (mode == 0) {  // decoder + reader ran; gate re-armed at the next sample (:112-123)
          f_n = 0;
          f_type ^= 1;
          f_ung = f_type ? EPC_WIN : RN16_WIN;
        } else {
          g.stop = true;
          g.consumed = pos + p;
          nvalid = p;
        }
      }
    }
    // (2) closed samples [p, nvalid)
    if (p < nvalid) {
      const uint64_t rem = lane_range(p, nvalid);
      // (2a) end of a reader command: the gate opens at the first sample with n_samples > T1 while
      //      POS_EDGE and num_pulses > 5 (gate_impl.cc:164-180) -- possible only before the first
      //      falling edge of the step (any edge restarts the 97-sample count)
      if (f_state == 1 && f_pulses > NUM_PULSES_CMD) {
        const int e = wv::ffs64(below & rem);
        int need = T1_SAMPLES - f_n;
        if (need < 0) need = 0;
        const int popen = p + need;
        if (popen < e && popen < nvalid) {
          closedmask = lane_range(p, popen + 1);
          openmask |= lane_range(popen, nvalid);     // the opening sample and everything after it
          open_lane = popen; open_type = f_type;
          f_open = 1;
          f_pulses = 0;
          f_n = nvalid - popen;                       // 1 for the opening sample + the rest of the step
          p = nvalid;
        }
      }
      // (2b) edge / pulse bookkeeping of a closed segment without opening, loop-free
      //      (gate_impl.cc:145-162).  The POS/NEG state after each sample is the type of the last
      //      threshold crossing: a carry chain with generate = above, kill = below.
      if (p < nvalid) {
        const int len = nvalid - p;
        const uint64_t m = lane_range(0, len);
        const uint64_t av = (above >> p) & m, bv = (below >> p) & m;
        const uint64_t pr = ~(av | bv) & m;                       // propagate: no crossing
        const uint64_t X = av | pr, Y = av;
        const uint64_t sum = X + Y + (uint64_t)(f_state & 1);
        const uint64_t s_before = (X ^ Y ^ sum) & m;              // carry INTO bit i = state before sample i
        const uint64_t F = bv & s_before;                         // falling edges (POS -> NEG)
        const uint64_t R = av & ~s_before;                        // rising edges  (NEG -> POS)
        const uint64_t marks = av | bv;
        if (marks) f_state = (int)((av >> (63 - __builtin_clzll(marks))) & 1ull);
        const uint64_t E = F | R;
        if (E == 0) {
          f_n += len;
        } else {
          if (R) {
            // a rising edge counts as a pulse if the low phase before it lasted more than PW/2 = 2
            // samples (n_samples > n_samples_PW/2), else the pulse count restarts
            uint64_t shortm = R & ((F << 1) | (F << 2));
            const int r1 = __builtin_ctzll(R);
            if ((F & lane_range(0, r1)) == 0 && !(f_n + r1 + 1 > PW_HALF)) shortm |= 1ull << r1;  // low phase began earlier
            if (shortm == 0) {
              f_pulses += wv::popc64(R);
            } else {
              const int hb = 63 - __builtin_clzll(shortm);
              f_pulses = wv::popc64(R & ~lane_range(0, hb + 1));
            }
          }
          f_n = len - 1 - (63 - __builtin_clzll(E));              // samples since the last edge
        }
        closedmask |= rem;
      }
    }
    g.f_n = f_n; g.f_state = f_state; g.f_pulses = f_pulses; g.f_open = f_open;
    g.f_ung = f_ung; g.f_type = f_type;
  }
  if (!g.f_open && g.f_n > GATE_N_SAT) g.f_n = GATE_N_SAT;
}

// ---- consumer wave: avg_ampl, the threshold votes, the scalar state machine -----------------------
// Step k goes through the edge / pulse / window state machine (on the scalar unit, over the vote masks); what the
// back wave needs -- which samples are "closed" (update dc_est), which lie inside a window, the gate openings --
// is left in the slot.

// the next step's slot, fetched one step early by the consumer
struct GateNext {
  int step;           // the step these values belong to (-1: none)
  int sq;             // the sequence word as read just before them: valid iff > step
  float amp, d;
};

RFID_DEVICE void gate_consume(const GateArgs &a, GateRegs &g, GateSlot *slot, const GateSlot *slot_next, GateNext &nx,
                              const int *seq, int k, int pos, int n, int lane) {
  float f_amp = 0.0f, f_d = 0.0f;
  {
    // Wait for step k and fetch it in ONE LDS round trip: the sequence word and the slot are read
    // back to back (a wave's LDS reads execute in order, and the producer wrote the slot
    // before it advanced the sequence word), and only then is the sequence word looked at.
    // Usually not even that: the previous step already fetched this slot (see below).
    if (!(nx.step == k && wv::uniform(nx.sq) > k)) {
      // (rare) wait for the producer, then fetch this step's slot the way the next one is fetched below: every value the
      // step works on comes out of wv::lds_prefetch, the compiler has no LDS read of its own to wait for where the two
      // paths meet (it would wait for ALL outstanding LDS operations there, the hand-over stores of the last step included)
      while (wv::lds_load(seq) <= k) wv::backoff();
      wv::lds_prefetch(seq, &slot->amp[lane], nx.sq, nx.amp, nx.d);
      wv::lds_prefetch_wait<0>(nx.sq, nx.amp, nx.d);
    }
    f_amp = nx.amp; f_d = nx.d;
    // fetch step k+1 now: the producer is normally more than one step ahead, and the reads
    // complete while this step is worked on (the sequence word tells the next call whether they count).  Not waited
    // for here: the caller's loop does that behind the hand-over of this step (wv::lds_prefetch_wait).
    static_assert(offsetof(GateSlot, d) - offsetof(GateSlot, amp) == 256, "ds_read2st64: amp and d 64 words apart");
    wv::lds_prefetch(seq, &slot_next->amp[lane], nx.sq, nx.amp, nx.d);
    nx.step = k + 1;
  }
  // avg_ampl after every sample (gate_impl.cc:130-134): the in-order sum in its integer-scan form where that is
  // provably exact, the 63-step chain otherwise; then the 0.75 avg threshold test (:136,147,155) as two votes
  int nvalid = (n - pos < 64) ? (n - pos) : 64;
  const float avg_in = g.avg_c;
  const float avg = chain_add_auto(g.avg_c, f_d, lane);
  g.avg_c = wv::readlane(avg, 63);   // the lanes past the end of the call add +0
  const float thresh = avg * THRESH_FRACTION;
  const uint64_t vmask = lane_range(0, nvalid);
  const uint64_t below = wv::ballot(f_amp < thresh) & vmask;
  const uint64_t above = wv::ballot(f_amp > thresh) & vmask;
  uint64_t closedmask, openmask;
  int open_lane, open_type;
  gate_fsm_step(a.mode, g, below, above, pos, nvalid, closedmask, openmask, open_lane, open_type);
  // streaming mode stopped inside the step: avg_ampl carries only over the samples actually consumed
  if (g.stop) g.avg_c = (nvalid > 0) ? wv::readlane(avg, nvalid - 1) : avg_in;
  // hand the step to the back wave (lane 0 writes after this wave's earlier LDS writes: in-order queue)
  if (a.mode != 0) wv::lds_store_rec(reinterpret_cast<int *>(&slot->b_openmask), 0, openmask, lane);
  wv::lds_store_rec(&slot->b_word, ((closedmask != 0) ? 1 : 0) | (g.stop ? 2 : 0) | ((openmask != 0) ? 4 : 0) | (nvalid << 8) |
                                       (open_lane << 16) | (open_type << 24), closedmask, lane);
}

// ---- back wave: the dc ring, dc_est and what hangs on it ------------------------------------------
// dc_est += (x - dc_samples[dc_index]) / 48 over the step's closed samples, in order (gate_impl.cc:139-143).  While
// whole steps are closed -- a reader command, the carrier between commands -- dc_samples[dc_index] is x[i-48] and
// the increments are the producer wave's; around the windows they are formed here from the ring.  Then two
// interleaved in-order sums, the window records (dc_est at the opening sample, gate_impl.cc:176) and, when
// streaming, the gated samples in[i] - dc_est (:176,187).
// the dc_est part of one step with closed samples: increments (from `spec` while whole steps are closed, else from
// the ring), ring upkeep, the two in-order sums; dcr / dci = dc_est after every sample of the step
// (first half: the increments of the step's closed samples and the ring upkeep; shared with the long-stream front end's
// dc_est pass, which sums them from two start values at once)
// -> true: the increments are the speculative ones (spec was called), false: they were formed here from the ring
template <class Spec>
RFID_DEVICE bool gate_dc_incr(GateBackRegs &g, uint64_t closedmask, uint64_t openmask, int nvalid, float2 yv, int lane,
                              float2 *lds_dc, float2 *lds_tmp, Spec spec, float &tre, float &tim, bool spec_ok = true) {
  const int cnt = wv::popc64(closedmask);
  bool used_spec = false;
  if (__builtin_expect(cnt == 64 && openmask == 0 && g.run_closed >= DC_LEN && spec_ok, 1)) {
    used_spec = true;
    // the 48 samples before every lane were closed too: dc_samples[dc_index] is x[i-48] and the speculative
    // increments are the reference's; the ring itself is left alone
    spec(tre, tim);
    g.dc_index += 64 - DC_LEN;            // (dc_index + 64) mod 48
    if (g.dc_index >= DC_LEN) g.dc_index -= DC_LEN;
    g.ring_stale = 1;
    if (g.run_closed < (1 << 28)) g.run_closed += 64;
  } else {
    if (g.ring_stale) {
      // the ring was not maintained: its content is the 48 samples before this step
      // (= lanes 16..63 of the previous closed step), oldest at dc_index
      if (lane >= 64 - DC_LEN) {
        int di = g.dc_index + (lane - (64 - DC_LEN));
        if (di >= DC_LEN) di -= DC_LEN;
        lds_dc[di] = g.prev_yv;
      }
      g.ring_stale = 0;
      wv::wave_sync();
    }
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool isclosed = ((closedmask >> lane) & 1ull) != 0;
    const int rank = wv::popc64(closedmask & lt);
    if (isclosed) lds_tmp[rank] = yv;
    wv::wave_sync();
    float2 old = make_float2(0.0f, 0.0f);
    if (isclosed) {
      if (rank < DC_LEN) {
        int di = g.dc_index + rank;
        if (di >= DC_LEN) di -= DC_LEN;
        old = lds_dc[di];
      } else {
        old = lds_tmp[rank - DC_LEN];
      }
    }
    const float nr = yv.x - old.x, ni = yv.y - old.y;
    float qr, qi;
    if (__builtin_expect((div_const_bad(nr) | div_const_bad(ni)) == 0, 1)) {
      qr = div_const_fast<DC_LEN>(nr); qi = div_const_fast<DC_LEN>(ni);
    } else {
      qr = wv::fdiv(nr, DC_LEN_F); qi = wv::fdiv(ni, DC_LEN_F);
    }
    tre = isclosed ? qr : 0.0f;
    tim = isclosed ? qi : 0.0f;
    wv::wave_sync();
    if (isclosed && rank >= cnt - DC_LEN) {
      int di = g.dc_index + rank;
      if (di >= DC_LEN) di -= DC_LEN;
      if (di >= DC_LEN) di -= DC_LEN;
      lds_dc[di] = yv;
    }
    wv::wave_sync();
    g.dc_index += cnt;
    if (g.dc_index >= DC_LEN) g.dc_index -= DC_LEN;
    if (g.dc_index >= DC_LEN) g.dc_index -= DC_LEN;
    // closed samples back-to-back up to the end of this step
    const uint64_t notclosed = lane_range(0, nvalid) & ~closedmask;
    if (notclosed == 0) {
      if (g.run_closed < (1 << 28)) g.run_closed += nvalid;
    } else {
      g.run_closed = nvalid - 1 - (63 - __builtin_clzll(notclosed));
    }
  }
  g.prev_yv = yv;
  return used_spec;
}
// The two in-order sums of a step when only their END values are wanted (no gate opening in the step, batch mode): the
// 64 addends of a component lie in LDS in sample order; every even lane adds up the real parts, every odd lane the
// imaginary parts, each by itself with 64 plain dependent adds out of registers -- half the vector instructions of the
// two 63-step DPP chains (chain_add2) and a third of their latency (a dependent v_add_f32 issues every ~4.5 cycles, a
// dependent DPP add every 12.5).  The arithmetic is the same left-to-right sum: (((carry + x_0) + x_1) + ...) + x_63.
RFID_DEVICE void chain_add2_ends(float &cr, float &ci, const float *re, const float *im, int lane) {
  const float4 *src = reinterpret_cast<const float4 *>((lane & 1) ? im : re);
  float4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = src[j];
  float sum = (lane & 1) ? ci : cr;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    sum = sum + v[j].x;
    sum = sum + v[j].y;
    sum = sum + v[j].z;
    sum = sum + v[j].w;
  }
  cr = wv::readlane(sum, 0);
  ci = wv::readlane(sum, 1);
}
template <bool SCAN, class Spec>
RFID_DEVICE void gate_dc_step(GateBackRegs &g, uint64_t closedmask, uint64_t openmask, int nvalid, float2 yv, int lane,
                              float2 *lds_dc, float2 *lds_tmp, Spec spec, float &dcr, float &dci, bool spec_ok = true) {
  float tre, tim;
  gate_dc_incr(g, closedmask, openmask, nvalid, yv, lane, lds_dc, lds_tmp, spec, tre, tim, spec_ok);
  if (SCAN) {
    // (ls_dc_kernel: one wave per unit, bound by the latency of its own sums.  In the back wave of the 4-wave
    // pipeline the two interleaved chains are faster: 2.45 vs 2.69 ms for the fused front end)
    dcr = chain_add_auto(g.dcr_c, tre, lane);
    dci = chain_add_auto(g.dci_c, tim, lane);
  } else {
    chain_add2(g.dcr_c, tre, g.dci_c, tim, lane, dcr, dci);
  }
  g.dcr_c = wv::readlane(dcr, 63);
  g.dci_c = wv::readlane(dci, 63);
}

RFID_DEVICE void gate_back(const GateArgs &a, GateBackRegs &g, const GateSlot *slot, int pos, int n_total, int row, int lane,
                           float2 *lds_dc, float2 *lds_tmp, bool &stop) {
  int word, w1 = 0;
  uint64_t closedmask, openmask;
  const int spec_bad_v = wv::lds_peek(&slot->spec_bad);   // (rides on the descriptor's round trip)
  wv::lds_load_rec(&slot->b_word, word, closedmask);
  const int flags = word & 7, nvalid = (word >> 8) & 0xff, open = (word >> 16) & 0x1ff;   // lane | type << 8
  // (batch mode only needs to know WHETHER samples lie inside a window: gate_dc_incr's test for a wholly closed step)
  if (a.mode != 0) wv::lds_load_rec(reinterpret_cast<const int *>(&slot->b_openmask), w1, openmask);
  else openmask = (flags & 4) ? 1ull : 0ull;
  const bool spec_ok = wv::uniform(spec_bad_v) == 0;
  float dcr, dci;
  const bool lanes_wanted = (open & 0xff) != 0xff || a.mode == 1;   // dc_est at a gate opening / under the gated samples
  if ((flags & 1) && !lanes_wanted) {
    // only dc_est after the step is wanted: sums out of registers (chain_add2_ends)
    float tre = 0.0f, tim = 0.0f;
    const bool spec = gate_dc_incr(g, closedmask, openmask, nvalid, slot->yv[lane], lane, lds_dc, lds_tmp,
                                   [&](float &, float &) {}, tre, tim, spec_ok);
    const float *re = slot->tre, *im = slot->tim;
    if (!spec) {
      float *t = reinterpret_cast<float *>(lds_tmp);   // (gate_dc_incr is through with it)
      t[lane] = tre; t[64 + lane] = tim;
      wv::wave_sync();
      re = t; im = t + 64;
    }
    chain_add2_ends(g.dcr_c, g.dci_c, re, im, lane);
    dcr = g.dcr_c; dci = g.dci_c;
  } else if (flags & 1) {
    gate_dc_step<false>(g, closedmask, openmask, nvalid, slot->yv[lane], lane, lds_dc, lds_tmp,
                 [&](float &tre, float &tim) { tre = slot->tre[lane]; tim = slot->tim[lane]; }, dcr, dci, spec_ok);
  } else {
    // the step lies entirely inside a window: dc_est, the ring and its index do not move
    g.run_closed = 0;
    dcr = g.dcr_c; dci = g.dci_c;
  }
  const int open_lane = open & 0xff;
  if (__builtin_expect(open_lane != 0xff, 0))   // rare: a window opened in this step
    gate_record_window(a, g, open_lane, (open >> 8) & 1, pos, n_total, row, lane, dcr, dci);
  if (__builtin_expect(a.mode == 1, 0)) {  // streaming: emit gated samples in[i] - dc_est (gate_impl.cc:176,187)
    if (openmask != 0) {
      const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
      const float2 yv = slot->yv[lane];
      const bool isopen = ((openmask >> lane) & 1ull) != 0;
      const int orank = wv::popc64(openmask & lt);
      if (isopen && g.written + orank < a.gated_cap)
        a.gated[g.written + orank] = make_float2(yv.x - dcr, yv.y - dci);
      g.written += wv::popc64(openmask);
    }
  }
  stop = (flags & 2) != 0;
}

// Workgroup = 16 waves = 4 traces.  Wave w serves trace w % 4 in role w / 4: 0 consumer (avg_ampl, state machine),
// 1 filter, 2 back (dc_est, window records), 3 producer (amplitude, ring difference, speculative dc increments).  A
// workgroup's waves are placed round-robin over the CU's 4 SIMDs, so the four waves of a trace share one SIMD (1024
// traces = the 1024 SIMDs of the device), and at that load the SIMD's vector ALU is what the trace is bound by: the
// four roles together issue ~300 vector instructions per step.  (Measured alternative: five roles per trace, the
// averaging in a wave of its own, two traces per workgroup -- 1.4x faster per trace while a CU holds two traces, 1.45x
// slower with the four it has to hold at 1024 traces.)  The waves of a trace talk through a 16-slot LDS ring with
// sequence counters (filter -> producer -> consumer -> back -> filter) -- no s_barrier, hence no coupling between
// the traces of a workgroup.
constexpr int GATE_RAW = 64 * DECIM + (NTAPS - DECIM);   // 344 raw samples feed 64 matched-filter outputs
constexpr int GATE_RAW4 = GATE_RAW / 2;                  // as float4 (2 samples each): 172
constexpr int GATE_RAW_LD = (GATE_RAW4 + 63) / 64;       // float4 loads per lane and step: 3
constexpr int GATE_RAW_DEPTH = 6;                        // steps of raw samples in flight per filter wave
constexpr int GATE_STREAMS_PER_WG = 4;
constexpr int GATE_ROLES = 4;            // consumer, filter, back, producer wave per trace
constexpr int GATE_THREADS = 64 * GATE_ROLES * GATE_STREAMS_PER_WG;
constexpr int GATE_WAVES_PER_SIMD = 4;   // one workgroup per CU: 16 waves, at most 128 VGPRs each
constexpr int GATE_SLOTS = 16;   // deep enough to ride out a reader command (a burst of ~20 slow consumer steps)
constexpr int GATE_PREFETCH = 4;    // steps (x64 samples) of matched-filter output held in registers

struct GateShared {          // per trace
  GateSlot slots[GATE_SLOTS];
  float win[WIN_LEN + 4];    // the producer wave's working copy of gate_impl::win_samples
  float2 dc[DC_LEN];         // gate_impl::dc_samples (back wave)
  alignas(16) float2 tmp[64];
  float4 rawtile[64 * GATE_RAW_LD];   // fused front end: the 344 raw samples one step's matched filter needs (+ padding)
  int fir_seq;               // steps whose samples are in the slot (filter wave)
  int prod_seq;              // steps the producer wave is through with
  int fsm_seq;               // steps the state machine is through with (consumer)
  int back_seq;              // steps finished by the back wave: their slots are free again
  int stop;                  // consumer -> the other waves: stop (streaming mode window close)
  int fsm_done;              // the consumer will not hand over any more steps (fsm_seq is final)
  int prod_done;             // the filter wave is through (fused front end: its y stores are visible device-wide)
};

// raw samples of one step (fused front end): float4 #(lane + 64 j) of the 172 the step needs
struct GateRawRegs {
  float4 v[GATE_RAW_LD];
};

// Loads are unconditional (a select or a lane-divergent branch on a loaded value would force an
// s_waitcnt right behind the load and drain the prefetch): the index is clamped into the row
// instead.  Samples below index 0 are zeroed when the step is consumed (gate_fir_step, first
// step only); samples at or above the trace length only feed outputs that do not exist.
struct GateTrue { static constexpr bool value = true; };
struct GateFalse { static constexpr bool value = false; };
template <bool INTERIOR = false>
RFID_DEVICE void gate_load_raw(GateRawRegs &r, const float2 *xs, int64_t hi_idx, int64_t r0, int lane, bool vec) {
  // r0 = raw index of the first sample of the step's window (even; -24 for the first step), wave-uniform and
  // 64 bits wide (a trace may hold more than 2^31 raw samples): the step's base pointer is scalar, the lane
  // offsets and their clamps are 32-bit
  const float2 *p = xs + r0;
  if (INTERIOR) {
    // the caller knows that the whole window lies inside the row and that rows are 16-byte aligned: scalar base +
    // a per-lane byte offset that does not change from step to step, no clamps
    const char *pb = reinterpret_cast<const char *>(p);
#pragma unroll
    for (int j = 0; j < GATE_RAW_LD; ++j) {
      int q = lane + 64 * j;
      q = (q < GATE_RAW4) ? q : (GATE_RAW4 - 1);
      r.v[j] = *reinterpret_cast<const float4 *>(pb + (uint32_t)(16 * q));
    }
    return;
  }
  const int lo = (r0 < 0) ? (int)(-r0) : 0;
  int64_t hrel = hi_idx - r0;
  hrel = (hrel > (1 << 30)) ? (1 << 30) : hrel;
  hrel = (hrel < -(1 << 30)) ? -(1 << 30) : hrel;
  const int hi = (int)hrel;
#pragma unroll
  for (int j = 0; j < GATE_RAW_LD; ++j) {
    int q = lane + 64 * j;
    q = (q < GATE_RAW4) ? q : (GATE_RAW4 - 1);
    int ri = 2 * q;
    ri = (ri < lo) ? lo : ri;
    ri = (ri > hi) ? hi : ri;
    if (vec) {
      r.v[j] = *reinterpret_cast<const float4 *>(p + ri);
    } else {
      const float2 lo2 = p[ri], hi2 = p[ri + 1];
      r.v[j] = make_float4(lo2.x, lo2.y, hi2.x, hi2.y);
    }
  }
}

// matched filter of one step inside the filter wave: y[n] = sum_{k=0..24} x[5n-24+k], k ascending
// (the arithmetic of mf_boxcar25_decim5_kernel), for n = first output of the step + lane
RFID_DEVICE float2 gate_fir_step(const GateRawRegs &r, float4 *tile4, int lane, bool first) {
  wv::wave_sync();   // the previous step's reads of the tile are done
#pragma unroll
  for (int j = 0; j < GATE_RAW_LD; ++j) {
    const int q = lane + 64 * j;
    float4 v = r.v[j];
    if (j == 0 && first && q < (NTAPS - 1) / 2) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // x[-24..-1] = 0
    tile4[q] = v;   // (entries >= GATE_RAW4 are padding: no lane-dependent branch around a pending load)
  }
  wv::wave_sync();
  const float2 *tile = reinterpret_cast<const float2 *>(tile4);
  static_assert(NTAPS == 25, "lds_sum25_in_order");
  return wv::lds_sum25_in_order(&tile[DECIM * lane]);   // single ds_read_b64s, never ds_read2_b64 (half the LDS rate)
}

template <bool FUSED>
RFID_DEVICE void gate_scan_body(const GateArgs &a) {
  RFID_SHARED GateShared sh_all[GATE_STREAMS_PER_WG];
  const int lane = wv::lane_id();
  const int wave = wv::uniform((int)(threadIdx.x >> 6));
  const int role = wave / GATE_STREAMS_PER_WG;           // 0 consumer, 1 filter, 2 back, 3 producer
  const int sl = wave % GATE_STREAMS_PER_WG;
  const int s = (int)blockIdx.x * GATE_STREAMS_PER_WG + sl;
  GateShared &sh = sh_all[sl];
  if (lane == 0 && role == 0) {
    sh.fir_seq = 0; sh.prod_seq = 0; sh.fsm_seq = 0; sh.back_seq = 0; sh.stop = 0; sh.fsm_done = 0; sh.prod_done = 0;
  }
  wv::block_sync();   // once, before any hand-off
  if (s >= a.n_streams) return;
  if (a.skip_if && wv::uniform(*a.skip_if) != 0) return;
  const int strm = s, row = s;
  const int64_t pos0 = a.pos0, chunk_len = a.chunk_len;
  GateState *st = a.state + row;
  const float2 *ys = a.y + (int64_t)strm * a.y_stride + pos0;
  int64_t n64 = a.n_dec;
  if (a.lens) {
    int64_t r = a.lens[strm];
    if (r < 0) r = 0;
    n64 = r / DECIM;
    if (n64 > a.n_dec) n64 = a.n_dec;
  }
  const int n_total = wv::uniform((int)n64);          // valid samples of the whole trace
  int64_t nl = n64 - pos0;                            // ... of this launch's chunk
  if (nl > chunk_len) nl = chunk_len;
  if (nl < 0) nl = 0;
  const int n = wv::uniform((int)nl);
  const int nsteps = (n + 63) >> 6;
  const int win_index0 = wv::uniform(st->win_index);

  if (role == 1) {
    // ================= filter wave: samples -> slot.yv =========================================
    // fused front end: raw samples in, matched filter here, y written for the decoder;
    // stage kernels / streaming: y in
    bool stopped = false;
    float2 prev_yv = make_float2(0.0f, 0.0f);   // the samples of the previous step (x[i-48] of the speculative dc_est increments)
    if (FUSED) {
      const float2 *xs = a.raw + (int64_t)strm * a.raw_stride;
      const bool vec = a.raw_vec_ok != 0;
      // last index a (two-sample) load may start at: inside the row's stride (rows are contiguous)
      const int64_t hi_idx = vec ? ((a.raw_stride - 2) & ~(int64_t)1) : (a.raw_stride - 2);
      float2 *yw = a.y_w + (int64_t)strm * a.y_stride + pos0;
      const int64_t rbase = pos0 * DECIM - (NTAPS - 1);   // raw index of the window of output pos0
      // Rolling prefetch: the raw samples of the next GATE_RAW_DEPTH steps are in flight at all times
      // (a register set is reloaded right after its step went to LDS) -- with 1024 filter waves
      // on the device HBM needs that many bytes outstanding to stream.  The main loop runs over
      // whole groups of GATE_RAW_DEPTH full steps with no conditional step inside: s_waitcnt vmcnt
      // counts in order, and any control-flow path on which a step is skipped makes the compiler
      // wait for the youngest load instead of the oldest.  (Batch mode only: nothing stops the scan.)
      const int nfull = n >> 6;                       // steps with all 64 samples
      const int ngroups = nfull / GATE_RAW_DEPTH;
      const bool at_start = pos0 == 0;
      int back_seen = 0;   // sh.back_seq as last read: it only grows, so the slot ring is polled only when it looks full
      if (ngroups > 0) {
        GateRawRegs buf[GATE_RAW_DEPTH];
#pragma unroll
        for (int u = 0; u < GATE_RAW_DEPTH; ++u) {
          gate_load_raw(buf[u], xs, hi_idx, rbase + (int64_t)u * 64 * DECIM, lane, vec);
          wv::compiler_fence();                       // keep the issue order: step 0 first
        }
        // groups whose reloads (the steps one group further on) lie wholly inside the row take the clamp-free loader
        // -- two copies of the loop, not a branch inside it (see above)
        int g_fast = 0;
        if (vec) {
          const int64_t j_max = (hi_idx - 2 * (GATE_RAW4 - 1) - rbase) / (64 * DECIM);   // last step with an interior window
          int64_t gf = (j_max + 1) / GATE_RAW_DEPTH - 1;
          gf = (gf < 0) ? 0 : gf;
          g_fast = (gf > ngroups) ? ngroups : (int)gf;
        }
        auto group = [&](int grp, auto interior) {
#pragma unroll
          for (int u = 0; u < GATE_RAW_DEPTH; ++u) {
            const int k = grp * GATE_RAW_DEPTH + u;
            while (k - back_seen >= GATE_SLOTS) { back_seen = wv::lds_load(&sh.back_seq); if (k - back_seen >= GATE_SLOTS) wv::backoff(); }
            const float2 yv = gate_fir_step(buf[u], sh.rawtile, lane, u == 0 && grp == 0 && at_start);
            gate_load_raw<decltype(interior)::value>(buf[u], xs, hi_idx, rbase + (int64_t)(k + GATE_RAW_DEPTH) * 64 * DECIM, lane, vec);
            yw[64 * k + lane] = yv;
            sh.slots[k % GATE_SLOTS].yv[lane] = yv;
            gate_spec_dc(sh.slots[k % GATE_SLOTS], yv, prev_yv, lane);
            wv::lds_store(&sh.fir_seq, k + 1, lane);   // after the slot's data (in-order LDS queue)
          }
        };
        int grp = 0;
        for (; grp < g_fast; ++grp) group(grp, GateTrue());
        for (; grp < ngroups; ++grp) group(grp, GateFalse());
      }
      // the last few steps (fewer than a group, the partial step included): load, then filter
      for (int k = ngroups * GATE_RAW_DEPTH; k < nsteps; ++k) {
        while (k - back_seen >= GATE_SLOTS) { back_seen = wv::lds_load(&sh.back_seq); if (k - back_seen >= GATE_SLOTS) wv::backoff(); }
        GateRawRegs r;
        gate_load_raw(r, xs, hi_idx, rbase + (int64_t)k * 64 * DECIM, lane, vec);
        float2 yv = gate_fir_step(r, sh.rawtile, lane, k == 0 && at_start);
        if (64 * k + lane < n) yw[64 * k + lane] = yv;
        else yv = make_float2(0.0f, 0.0f);
        sh.slots[k % GATE_SLOTS].yv[lane] = yv;
        gate_spec_dc(sh.slots[k % GATE_SLOTS], yv, prev_yv, lane);
        wv::lds_store(&sh.fir_seq, k + 1, lane);
      }
      wv::global_release();                       // the consumer's write-back re-reads y
    } else {
      float2 cur[GATE_PREFETCH], nxt[GATE_PREFETCH];
#pragma unroll
      for (int u = 0; u < GATE_PREFETCH; ++u) {
        const int i = 64 * u + lane;
        cur[u] = (i < n) ? ys[i] : make_float2(0.0f, 0.0f);
      }
      for (int base = 0; base < nsteps && !stopped; base += GATE_PREFETCH) {
#pragma unroll
        for (int u = 0; u < GATE_PREFETCH; ++u) {
          const int i = 64 * (base + GATE_PREFETCH + u) + lane;
          nxt[u] = (i < n) ? ys[i] : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int u = 0; u < GATE_PREFETCH; ++u) {
          const int k = base + u;
          if (k < nsteps && !stopped) {
            // wait for a free slot (the back wave is at most GATE_SLOTS steps behind)
            while (!stopped && k - wv::lds_load(&sh.back_seq) >= GATE_SLOTS) {
              stopped = wv::lds_load(&sh.stop) != 0;
              wv::backoff();
            }
            if (!stopped) {
              sh.slots[k % GATE_SLOTS].yv[lane] = cur[u];
              gate_spec_dc(sh.slots[k % GATE_SLOTS], cur[u], prev_yv, lane);
              wv::lds_store(&sh.fir_seq, k + 1, lane);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < GATE_PREFETCH; ++u) cur[u] = nxt[u];
      }
    }
    wv::lds_store(&sh.prod_done, 1, lane);
  } else if (role == 3) {
    // ================= producer wave: amplitude ring, (|x| - old)/100, speculative dc increments =================
    for (int j = lane; j < WIN_LEN; j += 64) sh.win[j] = st->win[j];
    int win_index = win_index0;
    wv::wave_sync();
    bool stopped = false;
    int fir_seen = 0;    // sh.fir_seq as last read (it only grows)
    for (int k = 0; k < nsteps && !stopped; ++k) {
      while (!stopped && fir_seen <= k) {
        fir_seen = wv::lds_load(&sh.fir_seq);
        if (fir_seen <= k) {
          stopped = wv::lds_load(&sh.stop) != 0;
          wv::backoff();
        }
      }
      if (!stopped) {
        GateSlot &slot = sh.slots[k % GATE_SLOTS];
        gate_produce(slot, slot.yv[lane], 64 * k, n, lane, sh.win, win_index);
        wv::lds_store(&sh.prod_seq, k + 1, lane);   // after the slot's data (in-order LDS queue)
      }
    }
  } else if (role == 2) {
    // ================= back wave: dc_est, window records, gated output; frees the slots ===========================
    float2 *lds_dc = sh.dc, *lds_tmp = sh.tmp;
    if (lane < DC_LEN) lds_dc[lane] = make_float2(st->dcr_re[lane], st->dcr_im[lane]);
    GateBackRegs g;
    g.dcr_c = wv::uniform(st->dc_re); g.dci_c = wv::uniform(st->dc_im);
    g.dc_index = wv::uniform(st->dc_index);
    g.run_closed = 0; g.ring_stale = 0;
    g.prev_yv = make_float2(0.0f, 0.0f);
    g.win_seq = wv::uniform(st->win_seq);
    g.n_complete = 0; g.written = 0;
    g.pos0 = (int)pos0; g.strm = strm;
    wv::wave_sync();
    bool stop = false;
    int fsm_seen = 0;    // sh.fsm_seq as last read (it only grows)
    for (int k = 0; k < nsteps && !stop; ++k) {
      bool have = true;
      while (fsm_seen <= k) {
        fsm_seen = wv::lds_load(&sh.fsm_seq);
        if (fsm_seen > k) break;
        if (wv::lds_load(&sh.fsm_done) != 0) {
          fsm_seen = wv::lds_load(&sh.fsm_seq);
          if (fsm_seen <= k) { have = false; break; }
        } else {
          wv::backoff();
        }
      }
      if (!have) break;
      gate_back(a, g, &sh.slots[k % GATE_SLOTS], 64 * k, n_total, row, lane, lds_dc, lds_tmp, stop);
      wv::lds_store(&sh.back_seq, k + 1, lane);   // slot k free again
    }
    if (g.ring_stale) {
      // materialise the dc ring: the last 48 closed samples (lanes 16..63 of the last closed step), oldest at dc_index
      if (lane >= 64 - DC_LEN) {
        int di = g.dc_index + (lane - (64 - DC_LEN));
        if (di >= DC_LEN) di -= DC_LEN;
        lds_dc[di] = g.prev_yv;
      }
      wv::wave_sync();
    }
    if (lane < DC_LEN) { st->dcr_re[lane] = lds_dc[lane].x; st->dcr_im[lane] = lds_dc[lane].y; }
    if (lane == 0) {
      st->dc_re = g.dcr_c; st->dc_im = g.dci_c; st->win_seq = g.win_seq; st->dc_index = g.dc_index;
      if (a.mode != 0) a.io[1] = g.written;
    }
    if (a.mode == 0) {
      const int before = (pos0 > 0) ? wv::uniform(a.wcount[row]) : 0;   // windows recorded by earlier chunks
      int tot = before + g.n_complete;
      tot = (tot < a.wmax) ? tot : a.wmax;
      if (lane == 0) a.wcount[row] = tot;
      gate_flat_lists(a, row, before, tot - before, lane);
    }
  } else {
    // ================= consumer: the edge / pulse / window state machine ============================================
    wv::set_priority_high();
    GateRegs g;
    g.avg_c = wv::uniform(st->avg_ampl);
    g.f_n = wv::uniform(st->n_samples); g.f_state = wv::uniform(st->signal_state);
    g.f_pulses = wv::uniform(st->num_pulses); g.f_open = wv::uniform(st->gate_open);
    g.f_ung = wv::uniform(st->n_to_ungate); g.f_type = wv::uniform(st->wtype);
    if (g.f_ung == 0) g.f_ung = g.f_type ? EPC_WIN : RN16_WIN;  // fresh state: first window is an RN16
    g.consumed = n; g.stop = false;
    GateNext nx;
    nx.step = -1; nx.sq = 0; nx.amp = nx.d = 0.0f;
    for (int k = 0; k < nsteps && !g.stop; ++k) {
      // (waits until step k went through the producer wave)
      gate_consume(a, g, &sh.slots[k % GATE_SLOTS], &sh.slots[(k + 1) % GATE_SLOTS], nx, &sh.prod_seq, k, 64 * k, n, lane);
      wv::lds_store(&sh.fsm_seq, k + 1, lane);   // step k handed to the back wave
      wv::lds_prefetch_wait<2>(nx.sq, nx.amp, nx.d);   // the slot of step k + 1, fetched when this step began: two LDS stores since
    }
    wv::lds_store(&sh.fsm_done, 1, lane);
    if (g.stop) wv::lds_store(&sh.stop, 1, lane);

    // ---- write state back ----------------------------------------------------------------
    // wait for the filter wave (fused front end: its y stores being visible)
    while (wv::lds_load(&sh.prod_done) == 0) wv::backoff();
    // amplitude ring: the last min(100, consumed) samples of this call overwrite their slots
    // (the producer wave read st->win before it produced step 0, i.e. long before this point --
    // except for an empty call, which writes nothing here)
    {
      const int c = g.consumed;
      const int first = (c > WIN_LEN) ? (c - WIN_LEN) : 0;
      for (int i = first + lane; i < c; i += 64) {
        const float2 v = FUSED ? wv::load_coherent(&ys[i]) : ys[i];
        st->win[(win_index0 + i) % WIN_LEN] = wv::hypot_f(v.x, v.y);
      }
    }
    if (lane == 0) {
      st->avg_ampl = g.avg_c;
      st->n_samples = g.f_n; st->signal_state = g.f_state; st->num_pulses = g.f_pulses;
      st->gate_open = g.f_open; st->n_to_ungate = g.f_ung; st->wtype = g.f_type;
      st->win_index = (win_index0 + g.consumed) % WIN_LEN;
      if (a.mode != 0) a.io[0] = g.consumed;
    }
  }
}

RFID_KERNEL_OCC(GATE_THREADS, GATE_WAVES_PER_SIMD) void gate_scan_kernel(GateArgs a) { gate_scan_body<false>(a); }
// fused front end: matched filter (in the filter waves) + gate scan in one launch; reads the raw
// 2 Msps samples once, writes y for the decoder
RFID_KERNEL_OCC(GATE_THREADS, GATE_WAVES_PER_SIMD) void front_end_fused_kernel(GateArgs a) { gate_scan_body<true>(a); }

// =========================================================================================
// 2b. Long-stream front end (rfid_ls2.hpp): where ONE long trace (or a few) can be cut along time.  Pieces are cut
//     where the gate's state machine is in its idle state: closed, POS_EDGE, no pulses counted, the last 48 samples
//     closed (then the two rings hold exactly the preceding samples) -- found here from the data.
// =========================================================================================
constexpr int LS_QUIET = EPC_WIN + T1_SAMPLES + 1 + WIN_LEN + DC_LEN;   // 1615: a window opened by the last command has closed,
                                                                        // both rings have refilled and n_samples has saturated since
static_assert(GATE_N_SAT < WIN_LEN + DC_LEN - 1, "n_samples must have saturated at every cut (147 samples after a window at least)");

struct LsCutArgs {
  const float2 *y;
  int64_t y_stride;
  const int64_t *lens;     // optional per-trace RAW lengths
  int64_t n_dec;
  int chunk;               // nominal unit length (decimated samples)
  int limit;               // search at most this far past the nominal boundary
  int max_b;               // boundaries per trace (row length of cut)
  int quiet;               // carrier samples required right before a cut (LS_QUIET: the gate idles; ~WIN_LEN: avg_ampl is at rest)
  int *cut;                // [n_streams][max_b]: cut position for nominal boundary j (j >= 1), or -1
};

// one wave per (trace, nominal boundary j): the first position p >= j*chunk whose a.quiet preceding samples all have
// |y|^2 >= 0.72 of the largest |y|^2 seen in the look-back region (carrier, no reader command), or -1
RFID_DEVICE void ls_cut_body(const LsCutArgs &a, const int s, const int j) {
  const int lane = wv::lane_id();
  int64_t n64 = a.n_dec;
  if (a.lens) { int64_t r = a.lens[s]; if (r < 0) r = 0; n64 = r / DECIM; if (n64 > a.n_dec) n64 = a.n_dec; }
  const int n = (int)n64;
  const int P = j * a.chunk;
  int *out = a.cut + (int64_t)s * a.max_b + j;
  if (P >= n) { if (lane == 0) *out = -1; return; }
  const float2 *ys = a.y + (int64_t)s * a.y_stride;
  int lo = P - a.quiet - 64;
  if (lo < 0) lo = 0;
  float ref = 0.0f;
  for (int base = lo; base < P; base += 64) {
    const int i = base + lane;
    float m2 = 0.0f;
    if (i < P) { const float2 v = ys[i]; m2 = v.x * v.x + v.y * v.y; }
    ref = (m2 > ref) ? m2 : ref;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const float o = wv::shfl_xor(ref, off); ref = (o > ref) ? o : ref; }
  // (0.85 of the largest amplitude: above the gate's own 0.75 avg_ampl test, so that the first, still high samples of
  // a reader command's falling ramp do not pass for carrier -- the state machine would already be at NEG_EDGE there)
  const float theta = 0.7225f * ref;
  int run = 0, found = -1;
  int end = P + a.limit;
  if (end > n) end = n;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int base = lo; base < end && found < 0; base += 64) {
    const int i = base + lane;
    bool low = false;
    if (i < n) { const float2 v = ys[i]; low = (v.x * v.x + v.y * v.y) < theta; } else low = true;
    const uint64_t lowmask = wv::ballot(low);
    const uint64_t below = lowmask & lt;
    const int q = below ? (lane - 1 - (63 - __builtin_clzll(below))) : (run + lane);   // quiet samples right before i
    const uint64_t hit = wv::ballot(i >= P && i < end && q >= a.quiet);
    if (hit) found = base + wv::ffs64(hit);
    run = lowmask ? __builtin_clzll(lowmask) : (run + 64);   // quiet samples at the end of the step
  }
  if (lane == 0) *out = found;
}
RFID_KERNEL(64) void ls_cut_kernel(LsCutArgs a) { ls_cut_body(a, (int)blockIdx.y, (int)blockIdx.x + 1); }
// two searches in one launch (the long-stream front end looks for idle cuts on a coarse grid and for rest points on a
// fine one): blockIdx.y < n_streams: a, else b
struct LsCut2Args { LsCutArgs a, b; int n_streams; };
RFID_KERNEL(64) void ls_cut2_kernel(LsCut2Args p) {
  const int y = (int)blockIdx.y, j = (int)blockIdx.x + 1;
  if (y < p.n_streams) { if (j < p.a.max_b) ls_cut_body(p.a, y, j); }
  else if (j < p.b.max_b) ls_cut_body(p.b, y - p.n_streams, j);
}

// look-ahead of the per-block calls: ONE packet per whole-chain pass for the host -- the window count, the first n_hdr window
// records and results, and the gated, DC-removed samples of the windows, one after the other (in[i] - dc_est,
// gate_impl.cc:176,187) with their squared magnitudes (:171,175,186: std::norm = re*re + im*im).  The first `usual`
// samples of both lie together right behind the header (what a call usually holds: one copy fetches it all), the rest
// behind them.  One workgroup per window; a window's place = the lengths of the windows before it.
struct GatedPack {
  const rfid_window *wtab; const int *wcount; const rfid_decode_result *res; int wmax;   // wmax: windows to pack at most
  const float2 *y;
  char *pack;              // page-locked host memory, written over the bus: [hdr | g[usual] | m[usual]] (what does not fit is left out:
  int n_hdr, usual;        //  the host sees it from the records and asks again with the right sizes)
  const int *only_if;      // optional: pack nothing (count 0) unless *only_if != 0
};
constexpr int GATED_HDR = 64;   // bytes in front of the window records (the count)
RFID_DEVICE size_t gated_pack_hdr_bytes(int n_hdr) { return (size_t)GATED_HDR + (sizeof(rfid_window) + sizeof(rfid_decode_result)) * (size_t)n_hdr; }
RFID_KERNEL(256) void gated_windows_kernel(GatedPack a) {
  int n = *a.wcount;
  if (a.only_if && *a.only_if == 0) n = 0;
  const int b = (int)blockIdx.x;
  char *pk = a.pack;
  rfid_window *hw = reinterpret_cast<rfid_window *>(pk + GATED_HDR);
  rfid_decode_result *hr = reinterpret_cast<rfid_decode_result *>(hw + a.n_hdr);
  if (b == 0 && threadIdx.x == 0) *reinterpret_cast<int *>(pk) = n;
  if (n > a.wmax) n = a.wmax;
  if (b >= n) return;
  if (b < a.n_hdr && threadIdx.x == 0) { hw[b] = a.wtab[b]; hr[b] = a.res[b]; }
  float2 *g_lo = reinterpret_cast<float2 *>(pk + gated_pack_hdr_bytes(a.n_hdr));
  float *m_lo = reinterpret_cast<float *>(g_lo + a.usual);
  int off = 0;
  for (int k = 0; k < b; ++k) off += a.wtab[k].type ? EPC_WIN : RN16_WIN;
  const rfid_window w = a.wtab[b];
  const int len = w.type ? EPC_WIN : RN16_WIN;
  for (int i = (int)threadIdx.x; i < len; i += 256) {
    const float2 v = a.y[w.start + i];
    const float re = v.x - w.dc_re, im = v.y - w.dc_im;
    const int k = off + i;
    if (k < a.usual) { g_lo[k] = make_float2(re, im); m_lo[k] = re * re + im * im; }
  }
}

// gate_impl.cc:112-123 for the streaming gate: SEEK_* -> CLOSED arms the next window (one launch, no host round trip)
RFID_KERNEL(64) void gate_arm_kernel(GateState *st, int n_to_ungate, int wtype) {
  if (threadIdx.x == 0) { st->n_samples = 0; st->n_to_ungate = n_to_ungate; st->wtype = wtype; }
}

// =========================================================================================
// 3. tag_decoder: one wavefront per window, persistent over the compact window list.
//    The window (250 or 1370 complex samples, 2 000 / 10 960 B) is read from HBM exactly
//    once with coalesced loads, DC-removed on the fly and staged in LDS together with its
//    squared magnitude; every later access (preamble correlation, 20x256 energy gathers,
//    2x128 half-bit gathers) hits LDS.  Reductions are wave shuffles.
// =========================================================================================
struct DecodeArgs {
  const float2 *y;
  int64_t y_stride;
  const rfid_window *flat;
  const int *flat_count;   // device counter; the kernel clamps it to flat_cap
  int flat_cap;
  rfid_decode_result *res;  // [n_streams][wmax], indexed stream*wmax + seq
  rfid_scores *scores;      // same indexing, nullable
  int wmax;
  float t_cand[N_TCAND];    // half-period candidates, computed on the host exactly as
                            // tag_decoder_impl.cc:151-152,162
};

// CRC-16/CCITT (init 0xFFFF, poly 0x1021, MSB first, complemented) is GF(2)-linear:
// register = K ^ XOR_{set message bits j} C[j].  C and K are compile-time tables.
struct Crc16Table {
  unsigned short c[112];
  unsigned short k;
  constexpr Crc16Table() : c{}, k(0) {
    for (int j = 0; j < 112; ++j) {
      unsigned reg = 0;
      for (int i = 0; i < 112; ++i) {
        const unsigned bit = (i == j) ? 1u : 0u;
        const unsigned msb = (reg >> 15) & 1u;
        reg = (reg << 1) & 0xFFFFu;
        if (msb ^ bit) reg ^= 0x1021u;
      }
      c[j] = (unsigned short)reg;
    }
    unsigned reg = 0xFFFFu;
    for (int i = 0; i < 112; ++i) {
      const unsigned msb = (reg >> 15) & 1u;
      reg = (reg << 1) & 0xFFFFu;
      if (msb) reg ^= 0x1021u;
    }
    k = (unsigned short)reg;
  }
};
__device__ const Crc16Table g_crc16 = Crc16Table();

template <int LEN>
RFID_DEVICE void stage_window(const float2 *src, float dcr, float dci, float2 *s, float *m2,
                              int lane) {
  constexpr int IT = (LEN + 63) / 64;
  float2 v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int j = lane + 64 * k;
    v[k] = (j < LEN) ? src[j] : make_float2(0.0f, 0.0f);
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int j = lane + 64 * k;
    if (j < LEN) {
      // gate output in[i] - dc_est and its std::norm (gate_impl.cc:175-176,186-187)
      const float re = v[k].x - dcr, im = v[k].y - dci;
      s[j] = make_float2(re, im);
      m2[j] = re * re + im * im;
    }
  }
}

// first-maximum argmax over lanes [0, nl): returns lane index; `strict_from_zero` reproduces
// tag_sync's  `if (corr > max)` scan that starts from max = 0 (index 0 when nothing exceeds 0);
// otherwise std::max_element semantics.
RFID_DEVICE int wave_first_argmax(float v, int nl, int lane, bool strict_from_zero) {
  float key = (lane < nl && v == v) ? v : -1.0f;  // scores are >= 0; NaN never wins
  float m = key;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float o = wv::shfl_xor(m, off);
    m = (o > m) ? o : m;
  }
  if (strict_from_zero && !(m > 0.0f)) return 0;
  const uint64_t hit = wv::ballot(lane < nl && key == m);
  return wv::ffs64(hit) & 63;
}

RFID_KERNEL(64) void decode_windows_kernel(DecodeArgs a) {
  RFID_SHARED float2 s[EPC_WIN + 2];
  RFID_SHARED float m2[EPC_WIN + 2];
  const int lane = wv::lane_id();
  int total = wv::uniform(*a.flat_count);
  if (total > a.flat_cap) total = a.flat_cap;

  for (int w = (int)blockIdx.x; w < total; w += (int)gridDim.x) {
    const rfid_window wd = a.flat[w];
    const int type = wv::uniform(wd.type);
    const float dcr = wv::uniform(wd.dc_re), dci = wv::uniform(wd.dc_im);
    const float2 *src = a.y + (int64_t)wv::uniform(wd.stream) * a.y_stride + wv::uniform(wd.start);
    if (type == RFID_DECODE_EPC) stage_window<EPC_WIN>(src, dcr, dci, s, m2, lane);
    else stage_window<RN16_WIN>(src, dcr, dci, s, m2, lane);
    wv::wave_sync();

    // ---- tag_sync (tag_decoder_impl.cc:78-109): 15 offsets x 6 non-zero preamble taps -----
    float cre = 0.0f, cim = 0.0f;
    if (lane < N_SYNC) {
      const int taps[6] = {0, 5, 15, 30, 50, 55};  // j*5 for TAG_PREAMBLE[j] == 1
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float2 v = s[lane + taps[k]];
        cre = cre + v.x;
        cim = cim + v.y;
      }
    }
    const float corr = cre * cre + cim * cim;
    const int m_idx = wave_first_argmax(corr, N_SYNC, lane, true);
    const float hre = wv::fdiv(wv::shfl(cre, m_idx), 6.0f);  // :103
    const float him = wv::fdiv(wv::shfl(cim, m_idx), 6.0f);
    const int index = m_idx + 65;                             // :107  m + 6*10 + 5
    const float findex = (float)index;
    const float nhim = -him;                                  // conj(h_est)

    rfid_decode_result r;
    r.type = type; r.index = index; r.h_re = hre; r.h_im = him; r.T = 0.0f;
    r.bits[0] = r.bits[1] = r.bits[2] = r.bits[3] = 0u;
    r.crc_ok = 0; r.tag_id = -1;
    float energy = 0.0f;

    if (type == RFID_DECODE_RN16) {
      // 32 half-bit samples at index + 5q (:237-253), 16 decisions (:114-142)
      bool cur = false;
      if (lane < 16) {
        const float2 p = s[index + 10 * lane], q = s[index + 10 * lane + 5];
        const float res = (p.x - q.x) * hre - (p.y - q.y) * nhim;
        cur = res > 0.0f;
      }
      const uint64_t c = wv::ballot(cur) & 0xFFFFull;
      const uint64_t bits = (c ^ ((c << 1) | 1ull)) & 0xFFFFull;  // bit = (cur != prev), prev0 = +1
      r.bits[0] = (uint32_t)bits;
      r.n_bits = 16;
    } else {
      // ---- half-period search (:150-166): lane t sums 256 squared magnitudes in order ----
      if (lane < N_TCAND) {
        const float Tt = a.t_cand[lane];
#pragma unroll 8
        for (int i = 0; i < 256; ++i) {
          const float pos = (float)i * Tt + findex;
          energy = energy + m2[wv::f2i(pos)];
        }
      }
      const int t_idx = wave_first_argmax(energy, N_TCAND, lane, false);
      const float T = a.t_cand[t_idx];
      r.T = T;
      // ---- 128 bit decisions (:171-190), lanes take j and j+64 ---------------------------
      const float T2 = 2.0f * T;
      bool cur0, cur1;
      {
        const int j = lane;
        const float2 p = s[wv::f2i((float)j * T2 + findex)];
        const float2 q = s[wv::f2i(((float)(j * 2) * T + T) + findex)];
        cur0 = ((p.x - q.x) * hre - (p.y - q.y) * nhim) > 0.0f;
      }
      {
        const int j = lane + 64;
        const float2 p = s[wv::f2i((float)j * T2 + findex)];
        const float2 q = s[wv::f2i(((float)(j * 2) * T + T) + findex)];
        cur1 = ((p.x - q.x) * hre - (p.y - q.y) * nhim) > 0.0f;
      }
      const uint64_t c0 = wv::ballot(cur0), c1 = wv::ballot(cur1);
      const uint64_t b0 = c0 ^ ((c0 << 1) | 1ull);
      const uint64_t b1 = c1 ^ ((c1 << 1) | (c0 >> 63));
      r.bits[0] = (uint32_t)b0; r.bits[1] = (uint32_t)(b0 >> 32);
      r.bits[2] = (uint32_t)b1; r.bits[3] = (uint32_t)(b1 >> 32);
      r.n_bits = 128;
      // ---- CRC-16 over frame bits 0..111 vs bits 112..127 (:401-445) ---------------------
      unsigned x = 0;
      if ((b0 >> lane) & 1ull) x ^= g_crc16.c[lane];
      if (lane < 48 && ((b1 >> lane) & 1ull)) x ^= g_crc16.c[lane + 64];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) x ^= wv::shfl_xor(x, off);
      const unsigned crc = (~(x ^ g_crc16.k)) & 0xFFFFu;
      unsigned rcvd = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) rcvd |= (unsigned)((b1 >> (48 + i)) & 1ull) << (15 - i);
      r.crc_ok = (crc == rcvd) ? 1 : 0;
      if (r.crc_ok) {
        int id = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) id |= (int)((b1 >> (40 + i)) & 1ull) << (7 - i);  // bits 104..111
        r.tag_id = id;
      }
    }

    const int64_t slot = (int64_t)wv::uniform(wd.stream) * a.wmax + wv::uniform(wd.seq);
    if (lane == 0) a.res[slot] = r;
    if (a.scores) {
      rfid_scores *sc = a.scores + slot;
      if (lane < N_SYNC) sc->corr[lane] = corr;
      if (lane < N_TCAND) sc->energy[lane] = energy;
    }
    wv::wave_sync();
  }
}

// -----------------------------------------------------------------------------------------
// 3b. Batched decoders, one launch per window type (the gate writes two compact lists).
//
// decode_epc3_kernel: THREE EPC windows per wavefront.  The cost of an EPC window is the
// half-period search: 20 candidates x 256 strictly sequential gathers.  One window keeps only
// 20 of 64 lanes busy, three keep 60.  Only what is gathered often is staged in LDS -- the
// squared magnitudes (5.5 KB / window) and the first 72 samples for the preamble correlation;
// the 256 half-bit samples of the final decision are re-gathered from global memory (they
// were streamed through this CU's L2 microseconds earlier).  18.6 KB LDS per wave -> 8 waves
// per CU, each with up to two windows of loads in flight.
// -----------------------------------------------------------------------------------------
struct DecodeListArgs {
  const float2 *y;
  int64_t y_stride;
  const rfid_window *list;  // compact list of one window type
  const int *count;         // device counter; clamped to cap
  int cap;
  rfid_decode_result *res;  // [n_streams][wmax]
  rfid_scores *scores;      // nullable
  int wmax;
  float t_cand[N_TCAND];
  int *sum;                 // nullable, [n_streams][wmax]: what stream_stats_kernel needs of a result in one word (stats_summary())
};
// type | crc_ok << 1 | (tag_id & 255) << 2: all the statistics kernel reads of a 48-byte result.  With few, long traces that
// kernel is one workgroup walking hundreds of thousands of results -- through one CU's 64 bytes per cycle
RFID_DEVICE int stats_summary(const rfid_decode_result &r) { return (r.type & 1) | ((r.crc_ok & 1) << 1) | ((r.tag_id & 255) << 2); }

constexpr int EPC_PACK = 3;
constexpr int EPC_GRP = N_TCAND;            // 20 lanes per window
constexpr int EPC_M2_STRIDE = EPC_WIN + 17; // 1387 = 11 mod 32: the three windows of a pack gather from disjoint LDS banks (1376 = 0 mod 32 made every gather a 2-way conflict)
constexpr int SYNC_KEEP = 72;               // samples 0..69 are touched by tag_sync

template <int K0>
RFID_DEVICE void epc_load_regs(const float2 *src, float2 (&v)[(EPC_WIN + 63) / 64], int lane, bool on) {
#pragma unroll
  for (int k = 0; k < (EPC_WIN + 63) / 64; ++k) {
    const int j = lane + 64 * k;
    v[k] = (on && j < EPC_WIN) ? src[j] : make_float2(0.0f, 0.0f);
  }
}

RFID_DEVICE void epc_stage_regs(const float2 (&v)[(EPC_WIN + 63) / 64], float dcr, float dci, float *m2,
                                float2 *sy, int lane) {
#pragma unroll
  for (int k = 0; k < (EPC_WIN + 63) / 64; ++k) {
    const int j = lane + 64 * k;
    if (j < EPC_WIN) {
      const float re = v[k].x - dcr, im = v[k].y - dci;   // gate output in[i] - dc_est
      m2[j] = re * re + im * im;                          // std::norm   (gate_impl.cc:175,186)
      if (k == 0 || (k == 1 && j < SYNC_KEEP)) sy[j] = make_float2(re, im);
    }
  }
}

struct EpcPackDesc {   // wave-uniform description of one pack of (up to) three EPC windows
  const float2 *src[EPC_PACK];
  float dcr[EPC_PACK], dci[EPC_PACK];
  int64_t slot[EPC_PACK];
  bool on[EPC_PACK];
};

// descriptor fetch is split in two so that the (vector-memory) loads can be issued two packs
// ahead and turned into wave-uniform values only when they have long arrived
RFID_DEVICE void epc_fetch_raw(const DecodeListArgs &a, int pk, int total, rfid_window (&raw)[EPC_PACK]) {
  const int w0 = pk * EPC_PACK;
#pragma unroll
  for (int q = 0; q < EPC_PACK; ++q) {
    const int w = w0 + q;
    raw[q] = a.list[(w < total) ? w : ((w0 < total) ? w0 : 0)];
  }
}

RFID_DEVICE void epc_uniformize(const DecodeListArgs &a, int pk, int total, const rfid_window (&raw)[EPC_PACK],
                                EpcPackDesc &d) {
  const int w0 = pk * EPC_PACK;
#pragma unroll
  for (int q = 0; q < EPC_PACK; ++q) {
    d.on[q] = (w0 + q) < total;
    const int stream = wv::uniform(raw[q].stream);
    d.src[q] = a.y + (int64_t)stream * a.y_stride + wv::uniform(raw[q].start);
    d.dcr[q] = wv::uniform(raw[q].dc_re);
    d.dci[q] = wv::uniform(raw[q].dc_im);
    d.slot[q] = (int64_t)stream * a.wmax + wv::uniform(raw[q].seq);
  }
}

RFID_DEVICE void decode_epc3_body(const DecodeListArgs &a) {
  RFID_SHARED float m2[EPC_PACK * EPC_M2_STRIDE];   // |x|^2 of the 3 windows; later reused as complex samples
  RFID_SHARED float2 sy[EPC_PACK * SYNC_KEEP];
  RFID_SHARED float sc0[64];
  RFID_SHARED float sc1[64];
  RFID_SHARED float sc2[64];
  const int lane = wv::lane_id();
  int total = wv::uniform(*a.count);
  if (total > a.cap) total = a.cap;
  const int g = lane / EPC_GRP;          // window of this lane within the pack (3 = idle lanes 60..63)
  const int t = lane - g * EPC_GRP;      // candidate / offset index
  const bool lane_on = g < EPC_PACK;
  const int n_packs = (total + EPC_PACK - 1) / EPC_PACK;
  const int stride = (int)gridDim.x;
  constexpr int NV = (EPC_WIN + 63) / 64;
  // per-lane constants fetched once (no vector-memory load inside the search/decision phases)
  const float my_Tt = a.t_cand[(t < N_TCAND) ? t : 0];
  const unsigned crc_lo = g_crc16.c[lane];
  const unsigned crc_hi = (lane < 48) ? g_crc16.c[lane + 64] : 0u;
  const unsigned crc_k = g_crc16.k;

  int pk = (int)blockIdx.x;
  if (pk >= n_packs) return;
  EpcPackDesc cur;
  rfid_window rawN[EPC_PACK];
  epc_fetch_raw(a, pk, total, rawN);

  while (pk < n_packs) {
    const int pk_next = pk + stride;
    epc_uniformize(a, pk, total, rawN, cur);
    // ---- the pack's windows are read from HBM exactly once, into registers --------------------
    float2 v0[NV], v1[NV], v2[NV];
    epc_load_regs<0>(cur.src[0], v0, lane, cur.on[0]);
    epc_load_regs<1>(cur.src[1], v1, lane, cur.on[1]);
    epc_load_regs<2>(cur.src[2], v2, lane, cur.on[2]);
    if (pk_next < n_packs) epc_fetch_raw(a, pk_next, total, rawN);   // descriptors one pack ahead
    // ---- stage |x|^2 and the sync samples -----------------------------------------------------
    epc_stage_regs(v0, cur.dcr[0], cur.dci[0], m2, sy, lane);
    epc_stage_regs(v1, cur.dcr[1], cur.dci[1], m2 + EPC_M2_STRIDE, sy + SYNC_KEEP, lane);
    epc_stage_regs(v2, cur.dcr[2], cur.dci[2], m2 + 2 * EPC_M2_STRIDE, sy + 2 * SYNC_KEEP, lane);
    wv::wave_sync();

    // ---- tag_sync per window: lane (g,t<15) sums the 6 non-zero preamble taps (:78-99) --------
    float cre = 0.0f, cim = 0.0f;
    if (lane_on && t < N_SYNC) {
      const float2 *sg = sy + g * SYNC_KEEP;
      const int taps[6] = {0, 5, 15, 30, 50, 55};
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float2 v = sg[t + taps[k]];
        cre = cre + v.x;
        cim = cim + v.y;
      }
    }
    const float corr = cre * cre + cim * cim;
    sc0[lane] = corr; sc1[lane] = cre; sc2[lane] = cim;
    wv::wave_sync();
    int m_idx = 0;
    float hre = 0.0f, him = 0.0f;
    if (lane_on) {
      float best = 0.0f;                       // the reference's scan: if (corr > max) ... from max = 0
      const int gb = g * EPC_GRP;
#pragma unroll
      for (int j = 0; j < N_SYNC; ++j) {
        const float v = sc0[gb + j];
        if (v > best) { best = v; m_idx = j; }
      }
      hre = wv::fdiv(sc1[gb + m_idx], 6.0f);   // :103
      him = wv::fdiv(sc2[gb + m_idx], 6.0f);
    }
    const int index = m_idx + 65;              // :107
    const float findex = (float)index;
    wv::wave_sync();

    // ---- half-period search (:150-166): lane (g,t) sums 256 squared magnitudes in order -------
    float energy = 0.0f;
    if (lane_on) {
      const float Tt = my_Tt;
      const float *mg = m2 + g * EPC_M2_STRIDE;
#pragma unroll 16
      for (int i = 0; i < 256; ++i) {
        const float pos = (float)i * Tt + findex;
        energy = energy + mg[wv::f2i(pos)];
      }
    }
    sc0[lane] = energy;
    wv::wave_sync();
    int t_idx = 0;
    if (lane_on) {
      const int gb = g * EPC_GRP;
      float best = sc0[gb];                    // std::max_element: first largest (:165)
#pragma unroll
      for (int j = 1; j < N_TCAND; ++j) {
        const float v = sc0[gb + j];
        if (best < v) { best = v; t_idx = j; }
      }
    }
    const float T_lane = wv::shfl(my_Tt, (lane_on ? g * EPC_GRP : 0) + t_idx);   // t_cand[t_idx]
    wv::wave_sync();   // all three searches done: the |x|^2 area is free

    // ---- per window: rewrite it as complex samples into LDS, gather the 2 x 128 half-bit
    //      samples from there (:171-190), CRC-16 (:401-445), result ---------------------------
    float2 *cs = reinterpret_cast<float2 *>(m2);
#pragma unroll
    for (int q = 0; q < EPC_PACK; ++q) {
      if (!cur.on[q]) continue;
      const float dcr = cur.dcr[q], dci = cur.dci[q];
      const float2 (&vq)[NV] = (q == 0) ? v0 : ((q == 1) ? v1 : v2);
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int j = lane + 64 * k;
        if (j < EPC_WIN) cs[j] = make_float2(vq[k].x - dcr, vq[k].y - dci);   // in[i] - dc_est
      }
      wv::wave_sync();
      const int lead = q * EPC_GRP;            // any lane of the group holds the window's values
      const float T = wv::readlane(T_lane, lead);
      const float fidx = wv::readlane(findex, lead);
      const float h_re = wv::readlane(hre, lead), h_im = wv::readlane(him, lead);
      const float nhim = -h_im, T2 = 2.0f * T;
      const int j0 = lane, j1 = lane + 64;
      const float2 pa = cs[wv::f2i((float)j0 * T2 + fidx)];
      const float2 qa = cs[wv::f2i(((float)(j0 * 2) * T + T) + fidx)];
      const float2 pb = cs[wv::f2i((float)j1 * T2 + fidx)];
      const float2 qb = cs[wv::f2i(((float)(j1 * 2) * T + T) + fidx)];
      const bool cur0 = ((pa.x - qa.x) * h_re - (pa.y - qa.y) * nhim) > 0.0f;
      const bool cur1 = ((pb.x - qb.x) * h_re - (pb.y - qb.y) * nhim) > 0.0f;
      const uint64_t c0 = wv::ballot(cur0), c1 = wv::ballot(cur1);
      const uint64_t b0 = c0 ^ ((c0 << 1) | 1ull);
      const uint64_t b1 = c1 ^ ((c1 << 1) | (c0 >> 63));
      unsigned x = 0;
      if ((b0 >> lane) & 1ull) x ^= crc_lo;
      if (lane < 48 && ((b1 >> lane) & 1ull)) x ^= crc_hi;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) x ^= wv::shfl_xor(x, off);
      const unsigned crc = (~(x ^ crc_k)) & 0xFFFFu;
      unsigned rcvd = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) rcvd |= (unsigned)((b1 >> (48 + i)) & 1ull) << (15 - i);
      rfid_decode_result r;
      r.type = RFID_DECODE_EPC; r.index = wv::f2i(fidx); r.h_re = h_re; r.h_im = h_im; r.T = T;
      r.bits[0] = (uint32_t)b0; r.bits[1] = (uint32_t)(b0 >> 32);
      r.bits[2] = (uint32_t)b1; r.bits[3] = (uint32_t)(b1 >> 32);
      r.n_bits = 128;
      r.crc_ok = (crc == rcvd) ? 1 : 0;
      r.tag_id = -1;
      if (r.crc_ok) {
        int id = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) id |= (int)((b1 >> (40 + i)) & 1ull) << (7 - i);
        r.tag_id = id;
      }
      if (lane == 0) { a.res[cur.slot[q]] = r; if (a.sum) a.sum[cur.slot[q]] = stats_summary(r); }
      if (a.scores) {
        rfid_scores *sc = a.scores + cur.slot[q];
        if (g == q && t < N_SYNC) sc->corr[t] = corr;
        if (g == q) sc->energy[t] = energy;
      }
      wv::wave_sync();   // the next window overwrites the complex-sample area
    }
    pk = pk_next;
  }
}

RFID_KERNEL(64) void decode_epc3_kernel(DecodeListArgs a) { decode_epc3_body(a); }

// decode_rn16x4_kernel: FOUR RN16 windows per wavefront, one 16-lane row each, no LDS.  An RN16
// decode touches 6x15 preamble taps and 32 half-bit samples; the lanes gather them straight
// from global memory (8-byte loads, neighbouring lanes -> neighbouring addresses).
constexpr int RN16_PACK = 4;

RFID_DEVICE float row_max16(float v) {   // maximum over the 16-lane row of the calling lane
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    const float o = wv::shfl_xor(v, off);
    v = (o > v) ? o : v;
  }
  return v;
}

// ticket: nullptr = packs by workgroup index; else packs are drawn from this counter (workgroups that come out of the EPC
// windows at different times take what is left)
RFID_DEVICE void decode_rn16x4_body(const DecodeListArgs &a, int *ticket) {
  const int lane = wv::lane_id();
  int total = wv::uniform(*a.count);
  if (total > a.cap) total = a.cap;
  const int row = lane >> 4, t = lane & 15;
  const int n_packs = (total + RN16_PACK - 1) / RN16_PACK;
  constexpr int TAKE = 8;   // packs per draw
  for (int p0 = ticket ? wv::uniform((lane == 0) ? wv::atomic_add(ticket, TAKE) : 0) : (int)blockIdx.x; p0 < n_packs;
       p0 = ticket ? wv::uniform((lane == 0) ? wv::atomic_add(ticket, TAKE) : 0) : (p0 + (int)gridDim.x))
  for (int pk = p0; pk < n_packs && pk < (ticket ? p0 + TAKE : p0 + 1); ++pk) {
    const int w = pk * RN16_PACK + row;
    const bool on = w < total;
    const rfid_window wd = a.list[on ? w : (total - 1)];
    const float2 *src = a.y + (int64_t)wd.stream * a.y_stride + wd.start;
    const float dcr = wd.dc_re, dci = wd.dc_im;
    // ---- tag_sync: offset t (0..14), 6 taps, in order ---------------------------------------
    float cre = 0.0f, cim = 0.0f;
    if (t < N_SYNC) {
      const int taps[6] = {0, 5, 15, 30, 50, 55};
      float2 v[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = src[t + taps[k]];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        cre = cre + (v[k].x - dcr);
        cim = cim + (v[k].y - dci);
      }
    }
    const float corr = cre * cre + cim * cim;
    const float key = (t < N_SYNC && corr == corr) ? corr : -1.0f;
    const float mx = row_max16(key);
    const uint64_t hit = wv::ballot(t < N_SYNC && key == mx);
    const unsigned rowhit = (unsigned)((hit >> (row * 16)) & 0xFFFFull);
    const int m_idx = (mx > 0.0f && rowhit) ? (wv::ffs64((uint64_t)rowhit) & 15) : 0;   // first maximum > 0, else 0
    const float hre = wv::fdiv(wv::shfl(cre, row * 16 + m_idx), 6.0f);
    const float him = wv::fdiv(wv::shfl(cim, row * 16 + m_idx), 6.0f);
    const int index = m_idx + 65;
    // ---- 32 half-bit samples at index + 5q (:237-253), 16 decisions (:114-142) ----------------
    const float2 p = src[index + 10 * t], q = src[index + 10 * t + 5];
    const float res = ((p.x - dcr) - (q.x - dcr)) * hre - ((p.y - dci) - (q.y - dci)) * (-him);
    const uint64_t call = wv::ballot(res > 0.0f);
    const uint64_t c = (call >> (row * 16)) & 0xFFFFull;
    const uint64_t bits = (c ^ ((c << 1) | 1ull)) & 0xFFFFull;
    if (on) {
      const int64_t slot = (int64_t)wd.stream * a.wmax + wd.seq;
      if (t == 0) {
        rfid_decode_result r;
        r.type = RFID_DECODE_RN16; r.index = index; r.h_re = hre; r.h_im = him; r.T = 0.0f;
        r.bits[0] = (uint32_t)bits; r.bits[1] = r.bits[2] = r.bits[3] = 0u;
        r.n_bits = 16; r.crc_ok = 0; r.tag_id = -1;
        a.res[slot] = r;
        if (a.sum) a.sum[slot] = stats_summary(r);
      }
      if (a.scores) {
        rfid_scores *sc = a.scores + slot;
        if (t < N_SYNC) sc->corr[t] = corr;
        sc->energy[t] = 0.0f;
        if (t < 4) sc->energy[16 + t] = 0.0f;
      }
    }
  }
}
RFID_KERNEL(64) void decode_rn16x4_kernel(DecodeListArgs a) { decode_rn16x4_body(a, nullptr); }

// tag_decoder in ONE launch: every persistent wave first takes its share of the EPC windows (three per wave and pass,
// LDS-staged), then draws RN16 windows (four per wave, no LDS) from a common counter -- the waves that come out of the EPC
// windows early do most of them, so the short RN16 pass fills the EPC pass's ragged end instead of paying a launch and
// a ramp of its own.
struct DecodeAllArgs {
  DecodeListArgs epc, rn16;
  int *ticket;              // zero at the start of the launch
  int *ticket_next;         // the next launch's counter: zeroed here (the launches alternate between two counters, so no
                            // fill is needed between them)
};
RFID_KERNEL(64) void decode_all_kernel(DecodeAllArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *a.ticket_next = 0;
  decode_epc3_body(a.epc);
  decode_rn16x4_body(a.rn16, a.ticket);
}

// =========================================================================================
// 4. per-trace statistics: replays the decoded windows of each trace in order through the
//    reader/decoder bookkeeping, including the TERMINATED cut-off.  One wavefront per trace.
// =========================================================================================
struct StatsArgs {
  const rfid_decode_result *res;  // [n_streams][wmax]
  const int *wcount;              // [n_streams]
  int wmax;
  int n_streams;
  int max_slot_number;            // 2^FIXED_Q (global_vars.cc:47)
  int max_num_queries;
  int number_unique_tags;
  rfid_stream_stats *out;         // [n_streams]
  const int *sum;                 // nullable: the results' one-word summaries (DecodeListArgs::sum), read instead of `res`
};

// One workgroup per trace (one wavefront, or sixteen when traces hold many windows: the host picks), 64 windows per
// wave and step.  The replay is sequential only through the TERMINATED cut-off (gate_impl.cc:101-109: checked before
// every window against n_queries_sent and the number of distinct tag ids read so far); both are monotone counts, so the
// cut-off index is found first -- n_queries_sent passes its limit right after the max_num_queries-th EPC window, the
// distinct-id count after the first read of the (number_unique_tags + 1)-th new id -- and everything before it is plain
// counting, each wave over its own contiguous share of the windows (a trace may hold hundreds of thousands: a one-lane
// replay cost 40 ms for the 320 000 windows of a 10 000-round inventory, one wave 4 ms).
constexpr int STATS_MAX_WAVES = 16;
constexpr int STATS_UNROLL = 4;
RFID_KERNEL(64 * STATS_MAX_WAVES) void stream_stats_kernel(StatsArgs a) {
  RFID_SHARED int hist[256];
  RFID_SHARED int first[256];      // index of the first CRC-verified read of each tag id
  RFID_SHARED int epc_cnt[STATS_MAX_WAVES];
  RFID_SHARED int sh_kq, sh_nepc, sh_nok;
  const int lane = wv::lane_id();
  const int wave = wv::uniform((int)(threadIdx.x >> 6)), nwv = (int)(blockDim.x >> 6);
  const int s = (int)blockIdx.x;
  if (s >= a.n_streams) return;
  rfid_stream_stats *o = a.out + s;
  for (int i = (int)threadIdx.x; i < 256; i += (int)blockDim.x) { hist[i] = 0; first[i] = 0x7fffffff; }
  if (threadIdx.x == 0) { sh_kq = 0x7fffffff; sh_nepc = 0; sh_nok = 0; }
  wv::block_sync();
  const int nw = wv::uniform(a.wcount[s]);
  const rfid_decode_result *rs = a.res + (int64_t)s * a.wmax;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int seg = (((nw + nwv - 1) / nwv) + 63) & ~63;            // windows per wave
  const int k0 = wave * seg, k1 = (k0 + seg < nw) ? (k0 + seg) : nw;
  // ---- pass 1: first reads of every id; EPC windows per wave ----
  // (four 64-window batches per turn, their loads in flight together: with few long traces a wave walks tens of
  // thousands of 48-byte records and is bound by the latency of one batch after the other)
  int epc_mine = 0;
  const int *sm = a.sum ? (a.sum + (int64_t)s * a.wmax) : nullptr;
  const bool sm_vec = sm && (((uintptr_t)sm) & 15u) == 0u;
  // four summaries per lane and load (16 bytes: a wave's load is 1 KB in one piece), zeros from `limit` on
  auto fetch4 = [&](const int k, const int limit, int (&w)[4]) {
    if (sm_vec && k + 3 < limit) wv::load4_i32(sm + k, w[0], w[1], w[2], w[3]);
    else { for (int j = 0; j < 4; ++j) w[j] = (k + j < limit) ? sm[k + j] : 0; }
  };
  if (sm) {
    for (int base = k0; base < k1; base += 256 * STATS_UNROLL) {
      int w[STATS_UNROLL][4];
#pragma unroll
      for (int u = 0; u < STATS_UNROLL; ++u) fetch4(base + 256 * u + 4 * lane, k1, w[u]);
#pragma unroll
      for (int u = 0; u < STATS_UNROLL; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((w[u][j] & 3) == 3) wv::atomic_min(&first[(w[u][j] >> 2) & 255], base + 256 * u + 4 * lane + j);
          epc_mine += wv::popc64(wv::ballot((w[u][j] & 1) != 0));
        }
    }
  }
  for (int base = k0; base < k1 && !sm; base += 64 * STATS_UNROLL) {
    int v[STATS_UNROLL];
#pragma unroll
    for (int u = 0; u < STATS_UNROLL; ++u) {
      const int k = base + 64 * u + lane;
      v[u] = 0;
      if (k < k1) {
        const rfid_decode_result &r = rs[k];
        v[u] = (r.type & 1) | ((r.crc_ok & 1) << 1) | ((r.tag_id & 255) << 2);
      }
    }
#pragma unroll
    for (int u = 0; u < STATS_UNROLL; ++u) {
      if ((v[u] & 3) == 3) wv::atomic_min(&first[(v[u] >> 2) & 255], base + 64 * u + lane);
      epc_mine += wv::popc64(wv::ballot((v[u] & 1) != 0));
    }
  }
  if (lane == 0) epc_cnt[wave] = epc_mine;
  wv::block_sync();
  // n_queries_sent = 1 + EPC windows so far (reader_impl.cc:259,335); it exceeds the limit once max_num_queries EPC
  // windows have been processed: k_q = the index right after the max_num_queries-th EPC window -- found by the wave
  // whose share holds it
  {
    int before = 0;
    for (int j = 0; j < wave; ++j) before += epc_cnt[j];
    const int need = a.max_num_queries - before;                   // that many more EPC windows, counted from k0
    if (a.max_num_queries <= 0) {
      if (threadIdx.x == 0) sh_kq = 0;
    } else if (need > 0 && need <= epc_mine) {
      int seen = 0;
      for (int base = k0; base < k1; base += 64) {
        const int k = base + lane;
        const bool is_epc = (k < k1) && ((sm ? sm[k] : rs[k].type) & 1);
        const uint64_t epc = wv::ballot(is_epc);
        const int c = wv::popc64(epc);
        if (seen + c >= need) {
          const uint64_t hit = wv::ballot(((epc >> lane) & 1ull) && wv::popc64(epc & lt) == need - seen - 1);
          if (lane == 0) sh_kq = base + wv::ffs64(hit) + 1;
          break;
        }
        seen += c;
      }
    }
  }
  wv::block_sync();
  const int k_q = sh_kq;
  // the (number_unique_tags + 1)-th smallest first-read index: the distinct-id count exceeds the limit right after it
  // (every wave works it out for itself: 256 ids)
  int k_u = 0x7fffffff;
  {
    int cand = 0x7fffffff;
    for (int q = 0; q < 4; ++q) {
      const int id = lane + 64 * q;
      const int f = first[id];
      if (f == 0x7fffffff) continue;
      int rank = 0;                                            // ids first read earlier than this one
      for (int j = 0; j < 256; ++j) rank += (first[j] < f) ? 1 : 0;
      if (rank == a.number_unique_tags && f < cand) cand = f;  // (first-read indices are distinct: one id per window)
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const int ot = wv::shfl_xor(cand, off); cand = (ot < cand) ? ot : cand; }
    cand = wv::uniform(cand);
    if (cand != 0x7fffffff) k_u = cand + 1;
  }
  int k_term = (k_q < k_u) ? k_q : k_u;
  const bool cut = k_term < nw;
  if (!cut) k_term = nw;
  // ---- pass 2: counts over the windows before the cut-off ----
  int n_epc = 0, n_ok = 0;
  const int k1t = (k1 < k_term) ? k1 : k_term;
  if (sm) {
    for (int base = k0; base < k1t; base += 256 * STATS_UNROLL) {
      int w[STATS_UNROLL][4];
#pragma unroll
      for (int u = 0; u < STATS_UNROLL; ++u) fetch4(base + 256 * u + 4 * lane, k1t, w[u]);
#pragma unroll
      for (int u = 0; u < STATS_UNROLL; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((w[u][j] & 3) == 3) wv::atomic_add(&hist[(w[u][j] >> 2) & 255], 1);
          n_epc += wv::popc64(wv::ballot((w[u][j] & 1) != 0));
          n_ok += wv::popc64(wv::ballot((w[u][j] & 3) == 3));
        }
    }
  }
  for (int base = k0; base < k1t && !sm; base += 64 * STATS_UNROLL) {
    int v[STATS_UNROLL];
#pragma unroll
    for (int u = 0; u < STATS_UNROLL; ++u) {
      const int k = base + 64 * u + lane;
      v[u] = 0;
      if (k < k1t) {
        const rfid_decode_result &r = rs[k];
        v[u] = (r.type & 1) | ((r.crc_ok & 1) << 1) | ((r.tag_id & 255) << 2);
      }
    }
#pragma unroll
    for (int u = 0; u < STATS_UNROLL; ++u) {
      if ((v[u] & 3) == 3) wv::atomic_add(&hist[(v[u] >> 2) & 255], 1);   // tag_decoder_impl.cc:356-364
      n_epc += wv::popc64(wv::ballot((v[u] & 1) != 0));
      n_ok += wv::popc64(wv::ballot((v[u] & 3) == 3));                     // :346
    }
  }
  if (lane == 0) { wv::atomic_add(&sh_nepc, n_epc); wv::atomic_add(&sh_nok, n_ok); }
  wv::block_sync();
  if (wave != 0) return;
  n_epc = sh_nepc; n_ok = sh_nok;
  int uniq = 0;
  for (int i = lane; i < 256; i += 64) {
    o->tag_reads[i] = hist[i];
    uniq += hist[i] ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) uniq += wv::shfl_xor(uniq, off);
  if (lane == 0) {
    const int n_queries = 1 + n_epc;  // START -> SEND_QUERY before the first sample (reader_impl.cc:218-260), +1 per EPC window (:259/:335)
    // slot / round roll-over per EPC window (tag_decoder_impl.cc:295,330-343,369-383): slot++ ; past max_slot_number -> 1, round++
    const int round = 1 + n_epc / a.max_slot_number, slot = 1 + n_epc % a.max_slot_number;
    int status = cut ? RFID_TERMINATED : RFID_RUNNING;
    if (status == RFID_RUNNING && (n_queries > a.max_num_queries || uniq > a.number_unique_tags)) status = RFID_TERMINATED;
    o->n_queries_sent = n_queries;
    o->cur_inventory_round = round;
    o->cur_slot_number = slot;
    o->n_epc_correct = n_ok;
    o->n_unique_tags = uniq;
    o->n_windows = nw;
    o->n_windows_used = k_term;
    o->status = status;
  }
}

// =========================================================================================
// 5. self-test of the primitives the exactness argument rests on
// =========================================================================================
struct SelfTestArgs {
  const float *x;    // [64]
  const float *num;  // [64]
  const float *den;  // [64]
  float carry;
  float *chain_out;  // [64]
  float *div_out;    // [64]
  float *hyp_out;    // [64] hypot(num, x)
  float *shr_out;    // [64]
  float *scan_out;   // [65] chain_add_scan (or its fallback); [64] = 1 when the scan was provably exact
};

RFID_KERNEL(64) void selftest_kernel(SelfTestArgs a) {
  const int lane = wv::lane_id();
  a.chain_out[lane] = chain_add(a.carry, a.x[lane], lane);
  if (a.scan_out) {
    float v = 0.0f;
    const bool ok = chain_add_scan(a.carry, a.x[lane], lane, v);
    a.scan_out[lane] = ok ? v : chain_add(a.carry, a.x[lane], lane);
    if (lane == 0) a.scan_out[64] = ok ? 1.0f : 0.0f;
  }
  {
    const float num = a.num[lane], den = a.den[lane];
    a.div_out[lane] = (den == WIN_LEN_F) ? div_const<WIN_LEN>(num) : ((den == DC_LEN_F) ? div_const<DC_LEN>(num) : wv::fdiv(num, den));
  }
  a.hyp_out[lane] = wv::hypot_f(a.num[lane], a.x[lane]);
  a.shr_out[lane] = wv::shr1(a.x[lane]);
}

// =========================================================================================
// 6. synthetic replicas (SURVEY.md section 8 d, configs 1/3/5: noise replicas of a base trace made
//    on the device).  out[s][i] = base[i] + sigma * (N(0,1) + j N(0,1)); the noise of replica r is a
//    pure function of (seed, r, i): Philox4x32-10 keyed by the seed, counter = (r, i / 2), two
//    Box-Muller pairs per call -> two complex samples.  Nothing here is on the receive path and
//    nothing needs to be bit-exact against the reference: this is the workload generator.
// =========================================================================================
struct SynthArgs {
  const float2 *base;   // [n_raw]
  float2 *out;          // [n_streams][out_stride]
  int64_t n_raw, out_stride;
  int64_t first_replica;   // replica index of out row 0 (so that a batch can be generated in pieces)
  float sigma;
  uint32_t key0, key1;  // seed
};
constexpr int SYNTH_THREADS = 256;
constexpr int SYNTH_PAIRS_PER_THREAD = 4;   // 4 x 2 complex samples per thread

RFID_DEVICE void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                               uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

RFID_DEVICE void box_muller(uint32_t a, uint32_t b, float &n0, float &n1) {
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);            // [0, 1)
  const float r = wv::sqrt_fast(-2.0f * wv::log_fast(u1));
  const float th = 6.28318530717958647692f * u2;
  n0 = r * wv::cos_fast(th);
  n1 = r * wv::sin_fast(th);
}

RFID_KERNEL(SYNTH_THREADS) void synth_replicas_kernel(SynthArgs a) {
  const int64_t s = (int64_t)blockIdx.y;
  const int64_t pair0 = ((int64_t)blockIdx.x * SYNTH_THREADS + threadIdx.x) * SYNTH_PAIRS_PER_THREAD;
  const uint64_t rep = (uint64_t)(a.first_replica + s);
  float2 *row = a.out + s * a.out_stride;
#pragma unroll
  for (int q = 0; q < SYNTH_PAIRS_PER_THREAD; ++q) {
    const int64_t pair = pair0 + q;
    const int64_t i = 2 * pair;
    if (i >= a.n_raw) break;
    uint32_t rnd[4];
    philox4x32_10((uint32_t)pair, (uint32_t)((uint64_t)pair >> 32), (uint32_t)rep, (uint32_t)(rep >> 32), a.key0, a.key1, rnd);
    float g0, g1, g2, g3;
    box_muller(rnd[0], rnd[1], g0, g1);
    box_muller(rnd[2], rnd[3], g2, g3);
    const float2 b0 = a.base[i];
    row[i] = make_float2(b0.x + a.sigma * g0, b0.y + a.sigma * g1);
    if (i + 1 < a.n_raw) {
      const float2 b1 = a.base[i + 1];
      row[i + 1] = make_float2(b1.x + a.sigma * g2, b1.y + a.sigma * g3);
    }
  }
}


// =========================================================================================
// 7. Gen2 trace synthesiser: the receive trace of a whole inventory run written
//    straight into HBM from a small slot table -- the reader's PIE envelope as reader_impl transmits it
//    (tables lib/reader_impl.cc:84-125 at the 1 us DAC resolution, command bits :131-162 with the CRC-5
//    of :383-443 appended by the host), the tags' FM0 backscatter (preamble global_vars.h:136) and
//    Philox noise:   x = L*tx + sum_k h_k * level_k * tx + sigma * (N(0,1) + j N(0,1))
//    at 2 Msps (two samples per microsecond).  One workgroup per slot
//        [Query | QueryRep] [1295 us CW: RN16 replies] [ACK] [4575 us CW: EPC reply]
//    (a carrier-only pseudo slot opens and closes the trace).  The noise-free part is bit-identical to
//    the numpy generator rfid/synth.py (same operation order: leak first, then the responders in table
//    order); the noise is that of synth_replicas_kernel -- a pure function of (seed, replica, sample).
//    Workload generator: nothing here is on the receive path.
// =========================================================================================
constexpr int G2_MAX_RESP = 8;          // responders per slot (collisions)
constexpr int G2_MAX_TAGS = 16;         // distinct backscatter coefficients
constexpr int G2_THREADS = 256;
constexpr int G2_PW = 12, G2_DELIM = 12, G2_DATA0 = 24, G2_DATA1 = 48, G2_RTCAL = 72, G2_TRCAL = 200;   // us
constexpr int G2_CW_QUERY = 240 + 480 + (17 + 6) * 25;      // 1295 us  (reader_impl.cc:69)
constexpr int G2_CW_ACK = 3 * 240 + 480 + (129 + 6) * 25;   // 4575 us  (reader_impl.cc:70)
constexpr int G2_HALF_BIT_RAW = 25;     // 12.5 us at 2 Msps
constexpr int G2_RN16_LV = 12 + 2 * 17; // half-bit levels of an RN16 reply (preamble + 16 bits + dummy 1)
constexpr int G2_EPC_LV = 12 + 2 * 129;
constexpr int G2_ENV_MAX = 1408;        // us: Query = 308 + 22 * 48 at most

struct Gen2SlotDev {     // one slot as the kernel reads it (the public rfid_synth_slot + what the host derived)
  int64_t raw_start;     // raw index of the slot's first sample
  int32_t kind;          // 0 Query, 1 QueryRep, 2 carrier only (cw_us long)
  int32_t cw_us;         // kind 2: length
  uint32_t cmd_bits;     // command bits after the frame sync / preamble, first bit = bit (n_cmd_bits - 1)
  int32_t n_cmd_bits;    // 22 (Query incl. CRC-5) / 4 (QueryRep)
  uint32_t ack_bits;     // 18 bits: 01 + RN16
  int32_t n_tags;        // responders
  int32_t has_epc;
  int32_t rn16_off_raw;  // reply start, raw samples after the end of the command
  int32_t epc_off_raw;   // EPC reply start, raw samples after the end of the ACK
  uint8_t tag[G2_MAX_RESP];
  uint16_t rn16[G2_MAX_RESP];   // bit 15 = first bit sent
  uint32_t epc[4];       // frame bit j at epc[j >> 5] bit (j & 31)
};

struct Gen2Args {
  const Gen2SlotDev *slots;
  int64_t n_slots;
  float2 *out;
  int64_t n_raw;         // samples of the whole trace (= capacity check done on the host)
  float leak_re, leak_im;
  float h_re[G2_MAX_TAGS], h_im[G2_MAX_TAGS];
  float sigma;
  uint32_t key0, key1;
  uint64_t replica;
};

RFID_DEVICE int g2_popc(uint32_t v) { return wv::popc64((uint64_t)v); }

// PIE envelope of: [delim data_0 RTcal (TRcal)] + n bits, one entry per microsecond; returns its length
RFID_DEVICE int g2_build_env(unsigned char *env, int *sym, bool with_trcal, uint32_t bits, int n_bits, int tid) {
  // sym[2k] = start, sym[2k+1] = high time of symbol k
  int n_sym = 0;
  if (tid == 0) {
    int t = 0, k = 0;
    sym[2 * k] = t; sym[2 * k + 1] = 0; t += G2_DELIM; k++;                                   // delimiter: low
    sym[2 * k] = t; sym[2 * k + 1] = G2_DATA0 / 2; t += G2_DATA0; k++;                        // data_0
    sym[2 * k] = t; sym[2 * k + 1] = G2_RTCAL - G2_PW; t += G2_RTCAL; k++;                    // RTcal
    if (with_trcal) { sym[2 * k] = t; sym[2 * k + 1] = G2_TRCAL - G2_PW; t += G2_TRCAL; k++; } // TRcal
    for (int b = n_bits - 1; b >= 0; --b) {
      const bool one = ((bits >> b) & 1u) != 0;
      sym[2 * k] = t; sym[2 * k + 1] = one ? (3 * G2_DATA1 / 4) : (G2_DATA0 / 2);
      t += one ? G2_DATA1 : G2_DATA0; k++;
    }
    sym[2 * k] = t;   // end
    sym[63] = k;
  }
  wv::block_sync();
  n_sym = sym[63];
  for (int k = 0; k < n_sym; ++k) {
    const int st = sym[2 * k], len = sym[2 * k + 2] - st, hi = sym[2 * k + 1];
    if (tid < len) env[st + tid] = (tid < hi) ? 1 : 0;
  }
  const int total = sym[2 * n_sym];
  wv::block_sync();
  return total;
}

// FM0 half-bit levels of preamble + n bits + dummy 1 (rfid/synth.py fm0_levels): the first half of bit m
// is the parity of the ones before it, the second half differs iff the bit is 0
RFID_DEVICE void g2_build_levels(unsigned char *lv, const uint32_t *words, int n_bits, bool msb16, int tid) {
  const unsigned char pre[12] = {1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 1};   // global_vars.h:136
  if (tid < 12) lv[tid] = pre[tid];
  if (tid <= n_bits) {
    const int m = tid;
    int ones = 0, bit = 1;   // the dummy bit is a 1
    if (msb16) {
      const uint32_t w = words[0] & 0xFFFFu;
      ones = g2_popc(m ? (w >> (16 - m)) : 0u);
      if (m < n_bits) bit = (int)((w >> (15 - m)) & 1u);
    } else {
      for (int q = 0; q < (m >> 5); ++q) ones += g2_popc(words[q]);
      if (m & 31) ones += g2_popc(words[m >> 5] & ((1u << (m & 31)) - 1u));
      if (m < n_bits) bit = (int)((words[m >> 5] >> (m & 31)) & 1u);
    }
    const int first = ones & 1;
    lv[12 + 2 * m] = (unsigned char)first;
    lv[12 + 2 * m + 1] = (unsigned char)(first ^ (bit ? 0 : 1));
  }
}

RFID_KERNEL(G2_THREADS) void synth_gen2_kernel(Gen2Args a) {
  RFID_SHARED unsigned char env_cmd[G2_ENV_MAX];
  RFID_SHARED unsigned char env_ack[G2_ENV_MAX];
  RFID_SHARED unsigned char lv_rn[G2_MAX_RESP][G2_RN16_LV + 2];
  RFID_SHARED unsigned char lv_epc[G2_EPC_LV + 2];
  RFID_SHARED int sym[64];
  RFID_SHARED Gen2SlotDev sl;
  const int tid = (int)threadIdx.x;
  const int64_t si = (int64_t)blockIdx.x + (int64_t)blockIdx.y * (int64_t)gridDim.x;
  if (si >= a.n_slots) return;
  if (tid == 0) sl = a.slots[si];
  wv::block_sync();
  int cmd_us = 0, ack_us = 0, total_us;
  if (sl.kind == 2) {
    total_us = sl.cw_us;
  } else {
    cmd_us = g2_build_env(env_cmd, sym, sl.kind == 0, sl.cmd_bits, sl.n_cmd_bits, tid);
    ack_us = g2_build_env(env_ack, sym, false, sl.ack_bits, 18, tid);
    for (int r = 0; r < sl.n_tags; ++r) {
      const uint32_t w = sl.rn16[r];
      g2_build_levels(lv_rn[r], &w, 16, true, tid);
    }
    if (sl.has_epc) g2_build_levels(lv_epc, sl.epc, 128, false, tid);
    wv::block_sync();
    total_us = cmd_us + G2_CW_QUERY + ack_us + G2_CW_ACK;
  }
  const int ack_at = cmd_us + G2_CW_QUERY;            // us
  const int rn_at = 2 * cmd_us + sl.rn16_off_raw;     // raw
  const int epc_at = 2 * (ack_at + ack_us) + sl.epc_off_raw;
  float2 *out = a.out + sl.raw_start;
  for (int us = tid; us < total_us; us += G2_THREADS) {
    // carrier envelope of this microsecond (both raw samples)
    float tx = 1.0f;
    if (sl.kind != 2) {
      if (us < cmd_us) tx = (float)env_cmd[us];
      else if (us >= ack_at && us < ack_at + ack_us) tx = (float)env_ack[us - ack_at];
    }
    float4 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 2 * us + h;
      // numpy's complex64 * float32: (ar*br - ai*0, ar*0 + ai*br)
      float re = a.leak_re * tx - a.leak_im * 0.0f;
      float im = a.leak_re * 0.0f + a.leak_im * tx;
      if (sl.kind != 2) {
        const int d = r - rn_at;
        if (d >= 0 && d < G2_RN16_LV * G2_HALF_BIT_RAW) {
          const int j = d / G2_HALF_BIT_RAW;
          for (int q = 0; q < sl.n_tags; ++q) {
            const float lvl = (float)lv_rn[q][j];
            const int t = sl.tag[q];
            re = re + (a.h_re[t] * lvl) * tx;
            im = im + (a.h_im[t] * lvl) * tx;
          }
        }
        const int e = r - epc_at;
        if (sl.has_epc && e >= 0 && e < G2_EPC_LV * G2_HALF_BIT_RAW) {
          const float lvl = (float)lv_epc[e / G2_HALF_BIT_RAW];
          const int t = sl.tag[0];
          re = re + (a.h_re[t] * lvl) * tx;
          im = im + (a.h_im[t] * lvl) * tx;
        }
      }
      if (h == 0) { o.x = re; o.y = im; } else { o.z = re; o.w = im; }
    }
    if (a.sigma != 0.0f) {
      const int64_t pair = (sl.raw_start >> 1) + us;     // raw_start is even
      uint32_t rnd[4];
      philox4x32_10((uint32_t)pair, (uint32_t)((uint64_t)pair >> 32), (uint32_t)a.replica, (uint32_t)(a.replica >> 32),
                    a.key0, a.key1, rnd);
      float g0, g1, g2, g3;
      box_muller(rnd[0], rnd[1], g0, g1);
      box_muller(rnd[2], rnd[3], g2, g3);
      o.x = o.x + a.sigma * g0; o.y = o.y + a.sigma * g1;
      o.z = o.z + a.sigma * g2; o.w = o.w + a.sigma * g3;
    }
    *reinterpret_cast<float4 *>(out + 2 * us) = o;
  }
}

}  // namespace rfidk

#include "rfid_ls2.hpp"   // long-stream front end (few long traces cut along time), built on the pieces above
