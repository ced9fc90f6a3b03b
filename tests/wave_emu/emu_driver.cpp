// tests/wave_emu/emu_driver.cpp -- TEST INFRASTRUCTURE ONLY (see rfid_device_env.h here).
//
// Runs the unmodified kernel source gen2-uhf-rfid-reader_amd/csrc/rfid_kernels.hpp on a
// lock-step 64-lane host emulator so the kernel logic (indices, state machine, ring
// handling, reductions) can be compared with the oracle in the GPU-less CI container.
// Built by tests/wave_emu/build.py into tests/wave_emu/librfid_wave_emu.so; nothing in the
// product imports or links it.
#include <stdio.h>
#include <stdlib.h>

#include <functional>
#include <vector>

#include <rfid_device_env.h>
#include "rfid_host_math.h"
#include "rfid_kernels.hpp"
#include "rfid_gen2_host.h"
#define LS2_DCB_SNAPS_N 2   // (a unit's gate openings go through LDS two at a time here: the in-between writes are exercised)
#define LS2_FIN_WPB 16   // (one workgroup at a time here: the finishing walk's waves of a trace share ONE workgroup)
#define LS2_LAUNCH(kernel, gx, gy, block, args) \
  emu::launch(emu::Idx3{(unsigned)(gx), (unsigned)(gy), 1}, emu::Idx3{(unsigned)(block), 1, 1}, [&]() { rfidk::kernel(args); })
#include "rfid_ls2_enqueue.hpp"

namespace emu {

Block *g_blk = nullptr;
Fiber *g_cur = nullptr;
Idx3 g_block_idx, g_grid_dim, g_block_dim;
ucontext_t g_sched;
static const std::function<void()> *g_body = nullptr;

void yield() { swapcontext(&g_cur->ctx, &g_sched); }

static void fiber_entry() {
  (*g_body)();
  g_cur->done = true;
  swapcontext(&g_cur->ctx, &g_sched);
}

void launch(Idx3 grid, Idx3 block, const std::function<void()> &body) {
  g_grid_dim = grid;
  g_block_dim = block;
  g_body = &body;
  const int nthreads = (int)(block.x * block.y * block.z);
  const int nwaves = (nthreads + 63) / 64;
  Block B;
  B.nthreads = nthreads;
  B.fibers.resize((size_t)nthreads);
  for (auto &f : B.fibers) f.stack.resize(256 * 1024);
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_block_idx = Idx3{bx, by, bz};
        B.xbuf.assign((size_t)nwaves * 2 * 64, 0);
        B.wave_arrived.assign((size_t)nwaves, 0);
        B.wave_gen.assign((size_t)nwaves, 0);
        B.block_arrived = 0;
        B.block_gen = 0;
        g_blk = &B;
        for (int t = 0; t < nthreads; ++t) {
          Fiber &f = B.fibers[(size_t)t];
          f.tid = Idx3{(unsigned)t, 0, 0};
          f.done = false;
          f.wave_calls = 0;
          f.block_calls = 0;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack.data();
          f.ctx.uc_stack.ss_size = f.stack.size();
          f.ctx.uc_link = &g_sched;
          makecontext(&f.ctx, fiber_entry, 0);
        }
        int remaining = nthreads;
        long sweeps = 0;
        while (remaining > 0) {
          // watchdog (a kernel that never finishes under lock-step emulation): report where the waves stand
          if (++sweeps == 3000000L && getenv("RFID_EMU_WATCHDOG")) {
            for (int w = 0; w < nwaves; ++w) {
              const Fiber &f0 = B.fibers[(size_t)w * 64];
              fprintf(stderr, "[emu watchdog] wave %d: lane0 done=%d wave_calls=%llu arrived=%d gen=%llu", w, (int)f0.done,
                      (unsigned long long)f0.wave_calls, B.wave_arrived[(size_t)w], (unsigned long long)B.wave_gen[(size_t)w]);
              for (int l = 1; l < 64 && w * 64 + l < nthreads; ++l)
                if (B.fibers[(size_t)w * 64 + l].wave_calls != f0.wave_calls || B.fibers[(size_t)w * 64 + l].done != f0.done) {
                  fprintf(stderr, " | lane %d: done=%d wave_calls=%llu", l, (int)B.fibers[(size_t)w * 64 + l].done,
                          (unsigned long long)B.fibers[(size_t)w * 64 + l].wave_calls);
                  break;
                }
              fprintf(stderr, "\n");
            }
            abort();
          }
          remaining = 0;
          for (int t = 0; t < nthreads; ++t) {
            Fiber &f = B.fibers[(size_t)t];
            if (f.done) continue;
            g_cur = &f;
            swapcontext(&g_sched, &f.ctx);
            if (!f.done) remaining++;
          }
        }
      }
  g_blk = nullptr;
  g_cur = nullptr;
}

}  // namespace emu

using namespace rfidk;

extern "C" {

// Batched pipeline mf -> gate -> decode -> stats, mirroring rfid_batch_process().
// raw: [B][stride] complex64.  Outputs ordered by (stream, seq).
int emu_batch_process(const float *raw, int B, long stride, long n_raw, const int64_t *lens, int fixed_q,
                      int max_num_queries, int number_unique_tags, rfid_window *windows,
                      rfid_decode_result *results, rfid_scores *scores, long cap, long *n_windows,
                      rfid_stream_stats *stats, float *y_out, long gate_chunk) {
  const long n_dec = n_raw / DECIM;
  long y_stride = (n_dec + 1) & ~1L;
  if (y_stride < 2) y_stride = 2;
  std::vector<float4> ybuf((size_t)(y_stride * B / 2 + 2));
  float2 *y = reinterpret_cast<float2 *>(ybuf.data());
  const int wmax = (int)(n_dec / (RN16_WIN + T1_SAMPLES + 1) + 2);
  const int flat_cap = wmax * B;
  std::vector<GateState> gstate((size_t)B);
  memset(gstate.data(), 0, sizeof(GateState) * (size_t)B);
  std::vector<rfid_window> wtab((size_t)flat_cap), flat((size_t)flat_cap * 2);
  std::vector<int> wcount((size_t)B, 0);
  int flat_count[2] = {0, 0};
  std::vector<rfid_decode_result> res((size_t)flat_cap);
  std::vector<rfid_scores> sc((size_t)flat_cap);
  memset(sc.data(), 0, sizeof(rfid_scores) * (size_t)flat_cap);

  MfArgs ma;
  ma.x = reinterpret_cast<const float2 *>(raw); ma.x_stride = stride; ma.n_raw = n_raw; ma.lens = lens;
  ma.n_out = n_dec; ma.in_off = -(NTAPS - 1);
  ma.vec_ok = ((stride & 1) == 0 && (((uintptr_t)raw) & 15) == 0) ? 1 : 0;
  ma.y = y; ma.y_stride = y_stride; ma.tile0 = 0; ma.stream0 = 0;
  const long tiles = (n_dec + MF_TILE - 1) / MF_TILE;
  const bool fused = gate_chunk < 0;   // gate_chunk -1: the fused front end, as rfid_batch_process() runs by default
  if (tiles > 0 && !fused)
    emu::launch(emu::Idx3{(unsigned)tiles, (unsigned)B, 1}, emu::Idx3{MF_THREADS, 1, 1},
                [&]() { mf_boxcar25_decim5_kernel(ma); });

  GateArgs ga = {};
  ga.y = y; ga.y_stride = y_stride; ga.n_dec = n_dec; ga.lens = lens; ga.state = gstate.data(); ga.n_streams = B;
  ga.wtab = wtab.data(); ga.wmax = wmax; ga.wcount = wcount.data(); ga.flat = flat.data();
  ga.flat_count = flat_count; ga.flat_cap = flat_cap; ga.mode = 0; ga.gated = nullptr; ga.gated_cap = 0;
  ga.io = nullptr;
  if (fused) {
    ga.pos0 = 0; ga.chunk_len = n_dec; ga.y_w = y;
    ga.raw = reinterpret_cast<const float2 *>(raw); ga.raw_stride = stride; ga.n_raw = n_raw; ga.raw_vec_ok = ma.vec_ok;
    emu::launch(emu::Idx3{(unsigned)((B + GATE_STREAMS_PER_WG - 1) / GATE_STREAMS_PER_WG), 1, 1},
                emu::Idx3{GATE_THREADS, 1, 1}, [&]() { front_end_fused_kernel(ga); });
  } else {
    // time-chunked launches with carried state, as rfid_batch_process() issues them
    const long chunk = (gate_chunk > 0) ? gate_chunk : (n_dec > 0 ? n_dec : 1);
    for (long p0 = 0; p0 == 0 || p0 < n_dec; p0 += chunk) {
      ga.pos0 = p0; ga.chunk_len = chunk;
      emu::launch(emu::Idx3{(unsigned)((B + GATE_STREAMS_PER_WG - 1) / GATE_STREAMS_PER_WG), 1, 1},
                  emu::Idx3{GATE_THREADS, 1, 1}, [&]() { gate_scan_kernel(ga); });
    }
  }
  if (y_out)
    for (int b = 0; b < B; ++b) memcpy(y_out + 2 * (size_t)b * n_dec, y + (size_t)b * y_stride, sizeof(float2) * (size_t)n_dec);

  DecodeListArgs da;
  da.y = y; da.y_stride = y_stride; da.cap = flat_cap; da.res = res.data(); da.scores = sc.data(); da.wmax = wmax;
  da.sum = nullptr;   // (the statistics kernel reads the results themselves here; emu_ls2_process runs it on the one-word summaries)
  rfidh::t_candidates(da.t_cand, 400000);
  {   // the one-launch tag_decoder, as rfid_batch_decode launches it (2 persistent waves here)
    DecodeAllArgs all;
    int ticket = 0, ticket_next = 0;
    all.epc = da; all.rn16 = da; all.ticket = &ticket; all.ticket_next = &ticket_next;
    all.epc.list = flat.data() + flat_cap; all.epc.count = &flat_count[1];
    all.rn16.list = flat.data(); all.rn16.count = &flat_count[0];
    emu::launch(emu::Idx3{2, 1, 1}, emu::Idx3{64, 1, 1}, [&]() { decode_all_kernel(all); });
  }

  StatsArgs sa;
  sa.res = res.data(); sa.wcount = wcount.data(); sa.wmax = wmax; sa.n_streams = B;
  sa.max_slot_number = 1 << fixed_q; sa.max_num_queries = max_num_queries;
  sa.number_unique_tags = number_unique_tags; sa.out = stats; sa.sum = nullptr;
  emu::launch(emu::Idx3{(unsigned)B, 1, 1}, emu::Idx3{256, 1, 1}, [&]() { stream_stats_kernel(sa); });   // four waves share a trace's windows

  long total = 0;
  for (int s = 0; s < B; ++s) {
    for (int k = 0; k < wcount[(size_t)s]; ++k) {
      if (total < cap) {
        const size_t off = (size_t)s * (size_t)wmax + (size_t)k;
        if (windows) windows[total] = wtab[off];
        if (results) results[total] = res[off];
        if (scores) scores[total] = sc[off];
      }
      total++;
    }
  }
  *n_windows = total;
  return 0;
}

// The long-stream front end (rfid_ls2.hpp) in place of the sequential gate scan: mf -> pieces / avg_ampl / state machine /
// dc_est / windows (the launch list of rfid_ls2_enqueue.hpp, as the library enqueues it) -> the sequential scan as the
// skipped-or-not fallback -> decode -> stats.  min_piece / target shrink the pieces so that small traces are cut many
// times.  cuts: test hook -- the idle-cut search is replaced by these positions (trace 0).  ctl_out: the Ls2Ctl block as ints.  carry / hold_last = the streaming form (state_blob: GateState of trace 0 in
// and out, consumed[0] = first unprocessed sample).
int emu_ls2_process(const float *raw, int B, long stride, long n_raw, const int64_t *lens, int fixed_q,
                    int max_num_queries, int number_unique_tags, rfid_window *windows,
                    rfid_decode_result *results, rfid_scores *scores, long cap, long *n_windows,
                    rfid_stream_stats *stats, int min_piece, int target, int *ctl_out, int ctl_cap,
                    void *state_blob, int hold_last, int *consumed_out, int *pieces_out, int pieces_cap,
                    const int *cuts, int n_cuts, int y_skip, int generous, int dc_rounds, int fused) {
  const long n_dec_all = n_raw / DECIM;
  const long n_dec = n_dec_all - y_skip;   // (y_skip: leading outputs that only exist to give the filter its history)
  long y_stride = (n_dec_all + 1) & ~1L;
  if (y_stride < 2) y_stride = 2;
  std::vector<float4> ybuf((size_t)(y_stride * B / 2 + 2));
  float2 *y0 = reinterpret_cast<float2 *>(ybuf.data());
  float2 *y = y0 + y_skip;
  const int wmax = (int)(n_dec / (RN16_WIN + T1_SAMPLES + 1) + 2);
  const int flat_cap = wmax * B;
  std::vector<GateState> gstate((size_t)B);
  memset(gstate.data(), 0, sizeof(GateState) * (size_t)B);
  if (state_blob) memcpy(gstate.data(), state_blob, sizeof(GateState));
  std::vector<rfid_window> wtab((size_t)flat_cap), flat((size_t)flat_cap * 2);
  std::vector<int> wcount((size_t)B, 0);
  int flat_count[2] = {0, 0};
  std::vector<rfid_decode_result> res((size_t)flat_cap);
  std::vector<rfid_scores> sc((size_t)flat_cap);
  memset(sc.data(), 0, sizeof(rfid_scores) * (size_t)flat_cap);

  MfArgs ma;
  ma.x = reinterpret_cast<const float2 *>(raw); ma.x_stride = stride; ma.n_raw = n_raw; ma.lens = lens;
  ma.n_out = n_dec_all; ma.in_off = -(NTAPS - 1);
  ma.vec_ok = ((stride & 1) == 0 && (((uintptr_t)raw) & 15) == 0) ? 1 : 0;
  ma.y = y0; ma.y_stride = y_stride; ma.tile0 = 0; ma.stream0 = 0;
  const long tiles = (n_dec_all + MF_TILE - 1) / MF_TILE;
  // fused: the long-stream front end's first pass runs the matched filter itself (ls2_front_kernel), as rfid_batch_process does
  // for fresh traces; y is only filtered here when that pass is not taken or gives up
  auto run_mf = [&]() {
    if (tiles > 0)
      emu::launch(emu::Idx3{(unsigned)tiles, (unsigned)B, 1}, emu::Idx3{MF_THREADS, 1, 1}, [&]() { mf_boxcar25_decim5_kernel(ma); });
  };
  if (fused && (state_blob || hold_last || y_skip)) return -1;
  if (!fused) run_mf();

  const Ls2Geometry geo = ls2_geometry(B, n_dec, min_piece, target);
  std::vector<char> ws;
  Ls2Ctl ctl_host;
  memset(&ctl_host, 0, sizeof(ctl_host));
  int ok = 0;
  if (geo.P > 0) {
    const Ls2Layout L = ls2_layout(geo, B, y_stride, wmax);
    ws.assign(L.total + 256, 0);
    char *base = ws.data() + (256 - ((uintptr_t)ws.data() & 255));
    Ls2Args a;
    memset(&a, 0, sizeof(a));
    a.y = y; a.y_stride = y_stride; a.lens = lens; a.n_dec = n_dec; a.n_streams = B;
    ls2_bind(a, base, L, geo);
    a.wtab = wtab.data(); a.wmax = wmax; a.wcount = wcount.data(); a.flat = flat.data(); a.flat_count = flat_count; a.flat_cap = flat_cap;
    a.carry = state_blob ? gstate.data() : nullptr; a.carry_out = state_blob ? gstate.data() : nullptr;
    a.hold_last = hold_last; a.force = state_blob ? 1 : 0;
    if (fused) {
      a.fused = 1; a.raw = reinterpret_cast<const float2 *>(raw); a.raw_stride = stride; a.raw_vec_ok = ma.vec_ok; a.y_w = y;
    }
    if (cuts) {   // test hook: cut trace 0 at the given positions (ascending) instead of searching idle points
      for (int i = 0; i < B * geo.max_bc; ++i) a.cut[i] = -1;
      for (int k = 0; k < n_cuts; ++k) {   // (a cut stands for the grid point it lies behind, less than half a step away)
        const int J = cuts[k] / geo.Pc;
        if (J >= 1 && J < geo.max_bc && cuts[k] - J * geo.Pc < geo.Pc / 2) a.cut[J] = cuts[k];
      }
    }
    ls2_enqueue(a, cuts == nullptr, nullptr, generous != 0, dc_rounds);
    ctl_host = *a.ctl;
    ok = a.ctl->ok;
    if (consumed_out) consumed_out[0] = a.consumed[0];
    if (pieces_out) {   // per piece in use: trace, pos0, len, true start of avg_ampl / dc_est (bit patterns), unit head
      int k = 0;
      for (int i = 0; i < geo.NS && k < pieces_cap; ++i) {
        if (a.piece[i].len <= 0) continue;
        int *o = pieces_out + 8 * k++;
        o[0] = i / geo.max_b; o[1] = a.piece[i].pos0; o[2] = a.piece[i].len;
        float f = ls2_from_ord(a.aT[i]); memcpy(&o[3], &f, 4);
        if (fused) { o[4] = o[5] = 0; o[6] = -1; o[7] = i; continue; }   // (the avg_ampl pieces are not the units' pieces there)
        const int h = a.fsm[i].unit;
        const int tu = (h / geo.max_b) * geo.max_bc + (h % geo.max_b) / LS2_FINE;   // the unit's idle-grid slot: its dc_est start
        f = ls2_from_ord(a.dT[2 * tu]); memcpy(&o[4], &f, 4);
        f = ls2_from_ord(a.dT[2 * tu + 1]); memcpy(&o[5], &f, 4);
        o[6] = h; o[7] = i;
      }
      if (k < pieces_cap) pieces_out[8 * k] = -1;
    }
  }
  if (ctl_out) memcpy(ctl_out, &ctl_host, sizeof(int) * (size_t)((int)(sizeof(Ls2Ctl) / 4) < ctl_cap ? (int)(sizeof(Ls2Ctl) / 4) : ctl_cap));
  if (fused && geo.P == 0) run_mf();
  if (!ok && !hold_last) {   // the fallback the library enqueues behind the front end (GateArgs::skip_if)
    if (fused && geo.P > 0) run_mf();   // (the fused first pass may have given up half-way: the library filters again, MfArgs::skip_if)
    if (state_blob) gstate[0].win_seq = 0;   // (a call's windows are numbered from 0, as the front end numbers them)
    GateArgs ga = {};
    ga.y = y; ga.y_stride = y_stride; ga.n_dec = n_dec; ga.lens = lens; ga.state = gstate.data(); ga.n_streams = B;
    ga.wtab = wtab.data(); ga.wmax = wmax; ga.wcount = wcount.data(); ga.flat = flat.data();
    ga.flat_count = flat_count; ga.flat_cap = flat_cap; ga.mode = 0; ga.pos0 = 0; ga.chunk_len = n_dec;
    emu::launch(emu::Idx3{(unsigned)((B + GATE_STREAMS_PER_WG - 1) / GATE_STREAMS_PER_WG), 1, 1},
                emu::Idx3{GATE_THREADS, 1, 1}, [&]() { gate_scan_kernel(ga); });
  }
  if (state_blob) memcpy(state_blob, gstate.data(), sizeof(GateState));

  std::vector<float4> sums4((size_t)flat_cap / 4 + 2);   // (16-byte aligned: the statistics kernel's 16-byte loads)
  struct { int *p; int *data() { return p; } } sums = {reinterpret_cast<int *>(sums4.data())};
  DecodeListArgs da;
  da.y = y; da.y_stride = y_stride; da.cap = flat_cap; da.res = res.data(); da.scores = sc.data(); da.wmax = wmax;
  da.sum = sums.data();
  rfidh::t_candidates(da.t_cand, 400000);
  da.list = flat.data() + flat_cap; da.count = &flat_count[1];
  emu::launch(emu::Idx3{2, 1, 1}, emu::Idx3{64, 1, 1}, [&]() { decode_epc3_kernel(da); });
  da.list = flat.data(); da.count = &flat_count[0];
  emu::launch(emu::Idx3{2, 1, 1}, emu::Idx3{64, 1, 1}, [&]() { decode_rn16x4_kernel(da); });

  StatsArgs sa;
  sa.res = res.data(); sa.wcount = wcount.data(); sa.wmax = wmax; sa.n_streams = B;
  sa.max_slot_number = 1 << fixed_q; sa.max_num_queries = max_num_queries;
  sa.number_unique_tags = number_unique_tags; sa.out = stats; sa.sum = sums.data();
  emu::launch(emu::Idx3{(unsigned)B, 1, 1}, emu::Idx3{256, 1, 1}, [&]() { stream_stats_kernel(sa); });

  long total = 0;
  for (int s = 0; s < B; ++s) {
    for (int k = 0; k < wcount[(size_t)s]; ++k) {
      if (total < cap) {
        const size_t off = (size_t)s * (size_t)wmax + (size_t)k;
        if (windows) windows[total] = wtab[off];
        if (results) results[total] = res[off];
        if (scores) scores[total] = sc[off];
      }
      total++;
    }
  }
  *n_windows = total;
  return ok;
}
int emu_ls2_ctl_words(void) { return (int)(sizeof(Ls2Ctl) / 4); }
// slots per workgroup of the chain launches (the library: 4096): small values make the emulated traces span several workgroups
void emu_ls2_chain_slots(int n) { ls2_chain_slots() = n; }
void emu_ls2_dcb_top_min(int n) { ls2_dcb_top_min() = n; }
void emu_ls2_dcb_bias(int n) { ls2_dcb_bias() = n; }
// from how many possible heads on the state machine takes its one-lane-per-unit form (the library: 8192)
void emu_ls2_fsm_lanes_min(int n) { ls2_fsm_lanes_min() = n; }

// gate_scan_kernel in streaming mode (mode 1) on one call's worth of samples.
// seek_type: -1 none, 0 SEEK_RN16, 1 SEEK_EPC applied before the scan (gate_impl.cc:112-123).
int emu_gate_stream(void *state_blob, const float *in, int n_in, int seek_type, float *out, int *consumed,
                    int *written, int *gate_open) {
  GateState *st = reinterpret_cast<GateState *>(state_blob);
  if (seek_type >= 0) {
    st->n_samples = 0;
    st->wtype = seek_type;
    st->n_to_ungate = seek_type ? EPC_WIN : RN16_WIN;
  }
  int io[2] = {n_in, 0};
  if (n_in > 0) {
    GateArgs ga = {};
    ga.y = reinterpret_cast<const float2 *>(in); ga.y_stride = n_in; ga.n_dec = n_in; ga.lens = nullptr;
    ga.pos0 = 0; ga.chunk_len = n_in; ga.state = st; ga.n_streams = 1; ga.wtab = nullptr; ga.wmax = 0; ga.wcount = nullptr; ga.flat = nullptr;
    ga.flat_count = nullptr; ga.flat_cap = 0; ga.mode = 1; ga.gated = reinterpret_cast<float2 *>(out);
    ga.gated_cap = n_in; ga.io = io;
    emu::launch(emu::Idx3{1, 1, 1}, emu::Idx3{GATE_THREADS, 1, 1}, [&]() { gate_scan_kernel(ga); });
  }
  *consumed = io[0];
  *written = io[1];
  *gate_open = st->gate_open;
  return 0;
}

int emu_gate_state_size(void) { return (int)sizeof(GateState); }

// matched filter in streaming form (staging = 24 history samples + new samples)
int emu_mf_stream(const float *staging, int n_staging, int in_off, int n_out, float *out) {
  if (n_out <= 0) return 0;
  std::vector<float4> ybuf((size_t)(n_out / 2 + 2));
  MfArgs ma;
  ma.x = reinterpret_cast<const float2 *>(staging); ma.x_stride = n_staging; ma.n_raw = n_staging; ma.lens = nullptr;
  ma.n_out = n_out; ma.in_off = in_off;
  ma.vec_ok = (in_off % 2 == 0 && (((uintptr_t)staging) & 15) == 0) ? 1 : 0;
  ma.y = reinterpret_cast<float2 *>(ybuf.data()); ma.y_stride = n_out; ma.tile0 = 0; ma.stream0 = 0;
  const int tiles = (n_out + MF_TILE - 1) / MF_TILE;
  emu::launch(emu::Idx3{(unsigned)tiles, 1, 1}, emu::Idx3{MF_THREADS, 1, 1}, [&]() { mf_boxcar25_decim5_kernel(ma); });
  memcpy(out, ybuf.data(), sizeof(float2) * (size_t)n_out);
  return 0;
}

// one window through decode_windows_kernel (input already DC-free, as the decoder block sees it)
int emu_decode_one(const float *win, int type, rfid_decode_result *res, rfid_scores *scores) {
  const int wlen = type ? EPC_WIN : RN16_WIN;
  rfid_window w;
  w.stream = 0; w.seq = 0; w.start = 0; w.type = type; w.dc_re = 0.0f; w.dc_im = 0.0f;
  int one = 1;
  DecodeArgs da;
  da.y = reinterpret_cast<const float2 *>(win); da.y_stride = wlen; da.flat = &w; da.flat_count = &one;
  da.flat_cap = 1; da.res = res; da.scores = scores; da.wmax = 1;
  rfidh::t_candidates(da.t_cand, 400000);
  memset(scores, 0, sizeof(*scores));
  emu::launch(emu::Idx3{1, 1, 1}, emu::Idx3{64, 1, 1}, [&]() { decode_windows_kernel(da); });
  return 0;
}

// primitives self-test kernel
int emu_selftest(const float *x, const float *num, const float *den, float carry, float *chain_out,
                 float *div_out, float *hyp_out, float *shr_out) {
  SelfTestArgs a;
  a.x = x; a.num = num; a.den = den; a.carry = carry; a.chain_out = chain_out; a.div_out = div_out;
  a.hyp_out = hyp_out; a.shr_out = shr_out; a.scan_out = nullptr;
  emu::launch(emu::Idx3{1, 1, 1}, emu::Idx3{64, 1, 1}, [&]() { selftest_kernel(a); });
  return 0;
}

// in-order sum: the integer-scan form against the chain; scan_out[64] = 1 when the scan was provably exact
int emu_chain_scan(const float *x, float carry, float *chain_out, float *scan_out) {
  std::vector<float> z(64, 1.0f), o(64 * 3);
  SelfTestArgs a;
  a.x = x; a.num = z.data(); a.den = z.data(); a.carry = carry; a.chain_out = chain_out; a.div_out = o.data();
  a.hyp_out = o.data() + 64; a.shr_out = o.data() + 128; a.scan_out = scan_out;
  emu::launch(emu::Idx3{1, 1, 1}, emu::Idx3{64, 1, 1}, [&]() { selftest_kernel(a); });
  return 0;
}

// chain_add_auto2 (long-stream front end: the in-order sums from two carries at once) on one step; scanned[0] = 1 when the
// shared integer-scan form applied
int emu_chain_scan2(const float *x, float ca, float cb, float *out_a, float *out_b, int *scanned) {
  struct Args { const float *x; float ca, cb; float *oa, *ob; int *sc; } a = {x, ca, cb, out_a, out_b, scanned};
  emu::launch(emu::Idx3{1, 1, 1}, emu::Idx3{64, 1, 1}, [&]() {
    const int lane = wv::lane_id();
    float va, vb;
    const bool ok = chain_add_scan2(a.ca, a.cb, a.x[lane], lane, va, vb);
    if (lane == 0) a.sc[0] = ok ? 1 : 0;
    chain_add_auto2(a.ca, a.cb, a.x[lane], lane, va, vb);
    a.oa[lane] = va; a.ob[lane] = vb;
  });
  return 0;
}

// synthetic-replica generator (workload generator, not on the receive path)
int emu_synth_replicas(const float *base, long n_raw, float *out, long out_stride, int n_streams, float sigma,
                       unsigned long long seed, long first_replica) {
  if (n_raw <= 0 || n_streams <= 0) return 0;
  SynthArgs a;
  a.base = reinterpret_cast<const float2 *>(base); a.out = reinterpret_cast<float2 *>(out); a.n_raw = n_raw;
  a.out_stride = out_stride; a.first_replica = first_replica; a.sigma = sigma;
  a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32);
  const long per_block = (long)SYNTH_THREADS * SYNTH_PAIRS_PER_THREAD * 2;
  const long blocks = (n_raw + per_block - 1) / per_block;
  emu::launch(emu::Idx3{(unsigned)blocks, (unsigned)n_streams, 1}, emu::Idx3{SYNTH_THREADS, 1, 1},
              [&]() { synth_replicas_kernel(a); });
  return 0;
}

// Gen2 trace synthesiser (workload generator): the same host layout as rfid_synth_gen2, the kernel emulated
long emu_synth_gen2(const rfid_synth_gen2_params *p, const rfid_synth_slot *slots, long n_slots, float *out, long out_cap,
                    float sigma, unsigned long long seed, long replica) {
  std::vector<Gen2SlotDev> dev;
  const int64_t total = rfidh::gen2_layout(*p, slots, n_slots, &dev);
  if (total < 0 || total > out_cap) return -1;
  Gen2Args a;
  memset(&a, 0, sizeof(a));
  a.slots = dev.data(); a.n_slots = (int64_t)dev.size(); a.out = reinterpret_cast<float2 *>(out); a.n_raw = total;
  a.leak_re = p->leak_re; a.leak_im = p->leak_im;
  for (int k = 0; k < G2_MAX_TAGS; ++k) { a.h_re[k] = p->h_re[k]; a.h_im[k] = p->h_im[k]; }
  a.sigma = sigma; a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32); a.replica = (uint64_t)replica;
  emu::launch(emu::Idx3{(unsigned)dev.size(), 1, 1}, emu::Idx3{G2_THREADS, 1, 1}, [&]() { synth_gen2_kernel(a); });
  return (long)total;
}

void emu_philox4x32_10(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
  uint32_t o[4];
  philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], o);
  for (int i = 0; i < 4; ++i) out[i] = o[i];
}

}  // extern "C"
