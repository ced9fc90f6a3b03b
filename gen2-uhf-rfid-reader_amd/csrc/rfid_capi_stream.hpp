// rfid_capi_stream.hpp -- part of rfid_capi.hip (included there, once): the HOST protocols that sit between a scheduler's per-block calls
// and the device's whole-chain passes --
//   (1b) whole-chain streaming: rfid_stream_begin / _work / _end (sio_*: one submission per chunk, every block's state carried on
//        the device, the upload of call k + 1 beside the processing of call k);
//   (1c) the look-ahead of the per-block calls, keyed on the filter's or on the gate's input (la_*: the calls gather, passes are
//        submitted and collected, late filter outputs, the gate consuming ahead, the end of the input, rfid_lookahead_*).
// No kernel lives here.  tests/test_capi_protocol.py runs all of it on the CPU (this source, a stand-in HIP runtime, the kernels on the
// wave emulator).
// ======================================================================================
// (1b) whole-chain streaming
// ======================================================================================
namespace {
const int64_t SIO_SMALL_DEC = 32768;   // decimated samples below which a pass takes the sequential scan (sio_submit)
// (with the look-ahead's 65 536-sample passes too: the sequential scan over such a pass -- one launch, 0.65 ms -- instead of the
// front end's list -- ~50 launches, 0.25 ms of device, 0.1 ms of the host's time to enqueue -- was measured: the scheduler's
// calls then wait for the device, 393 -> 310 Msamples/s at 8 192-item buffers, profiles/r05/drop_in_path.txt)
const int64_t SIO_SMALL_DEC_LA = 32768;
const int SIO_HIST = 28;   // raw samples kept before the held-back tail: 24 of filter history + the decimation group
                           // (= rfid_ctx::StreamIO::hist() of a raw stream)

void sio_free(rfid_ctx *c) {
  rfid_ctx::StreamIO &io = c->sio;
  if (io.open && c->stream) (void)hipStreamSynchronize(c->stream);   // (a submitted pass may still be running on these buffers)
  for (int i = 0; i < 2; ++i) {
    if (io.d_buf[i]) (void)hipFree(io.d_buf[i]);
    if (io.h_pin[i]) (void)hipHostFree(io.h_pin[i]);
    if (io.ev_up[i]) (void)hipEventDestroy(io.ev_up[i]);
    if (io.ev_free[i]) (void)hipEventDestroy(io.ev_free[i]);
    io.d_buf[i] = nullptr; io.h_pin[i] = nullptr; io.ev_up[i] = nullptr; io.ev_free[i] = nullptr;
  }
  if (io.copy_stream) (void)hipStreamDestroy(io.copy_stream);
  io.copy_stream = nullptr;
  if (io.ev_y) (void)hipEventDestroy(io.ev_y);
  io.ev_y = nullptr;
  if (io.ev_hist) (void)hipEventDestroy(io.ev_hist);
  io.ev_hist = nullptr;
  io.acc_new = 0;
  io.pass.active = false;
  io.open = false;
  io.failed = false;
  io.ymode = false;
  c->y_view = nullptr;
}

// READER_STATE bookkeeping for one decoded window, as the blocks do it call by call (tag_decoder_impl.cc:267-388,
// reader_impl.cc:251-344) incl. the TERMINATED cut-off that gate_impl.cc:101-109 applies before the next window
bool sio_account(rfid_ctx *c, const rfid_decode_result &r) {
  rfid_reader_state &rs = c->rs;
  if (rs.n_queries_sent > c->prm.max_num_queries || rs.n_unique_tags > c->prm.number_unique_tags) rs.status = RFID_TERMINATED;
  if (rs.status != RFID_RUNNING) return false;
  if (r.type == RFID_DECODE_EPC) {
    rs.cur_slot_number++;
    if (rs.cur_slot_number > rs.max_slot_number) { rs.cur_slot_number = 1; rs.cur_inventory_round += 1; }
    if (r.crc_ok) {
      rs.n_epc_correct += 1;
      const int id = r.tag_id & 255;
      if (rs.tag_reads[id] == 0) rs.n_unique_tags++;
      rs.tag_reads[id]++;
    }
    rs.n_queries_sent += 1;
    rs.decoder_status = RFID_DECODE_RN16; rs.gate_status = RFID_GATE_SEEK_RN16;
  } else {
    rs.decoder_status = RFID_DECODE_EPC; rs.gate_status = RFID_GATE_SEEK_EPC;
  }
  rs.gen2_logic_status = RFID_IDLE;
  return true;
}

// processes the chunk that sits in d_buf[b]: [SIO_HIST history | tail_len held back | n_new new] ending at tail_max + n_new
// (look-ahead) the packet for the host: window count, window records, results, gated samples and their |.|^2 -- sized for
// what a call usually holds and fetched with ONE copy
int sio_enqueue_packet(rfid_ctx *c, int n_hdr, int usual, const int *only_if) {
  const size_t hdr = GATED_HDR + (sizeof(rfid_window) + sizeof(rfid_decode_result)) * (size_t)n_hdr;
  const size_t first = hdr + (sizeof(float2) + sizeof(float)) * (size_t)usual;
  // (with room to spare: the window count creeps up and down from call to call, and every re-allocation synchronises the device)
  if (first > c->la.h_cap) {
    if (c->la.h_pack) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->la.h_pack); }
    c->la.h_pack = nullptr; c->la.h_cap = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->la.h_pack, first + first / 2, hipHostMallocDefault));
    c->la.h_cap = first + first / 2;
  }
  // the kernel writes the packet into page-locked host memory itself (half a megabyte over the bus at the end of a pass): a
  // device-side packet + hipMemcpyAsync cost the submitting call ~90 us of host time per pass
  GatedPack gp;
  gp.wtab = c->d_wtab; gp.wcount = c->d_wcount; gp.res = c->d_res; gp.wmax = n_hdr; gp.y = c->y();
  gp.pack = c->la.h_pack; gp.n_hdr = n_hdr; gp.usual = usual;
  gp.only_if = only_if;
  hipLaunchKernelGGL(gated_windows_kernel, dim3((unsigned)n_hdr), dim3(256), 0, c->stream, gp);
  HIPCHK(c, hipGetLastError());
  return RFID_OK;
}

// One whole-chain pass over the raw samples of buffer b (the held-back ones in front of the n_new new ones), in two
// halves.  sio_submit enqueues everything that needs no decision of the host: matched filter, (look-ahead: the copy of
// the filter outputs the current call hands out, io.ev_y behind it,) the long-stream front end from the carried gate
// state and -- look-ahead only, on the assumption that the front end succeeds -- the decoder and the packet of results.
// sio_collect waits for the pass, runs the sequential scan where the front end left something over, fetches the
// results, and moves what was not processed in front of the other buffer.  rfid_stream_work does both in one call; the
// look-ahead of the per-block calls collects a pass when the next rfid_mf_work call arrives, so that the device works
// on a call's samples while the scheduler hands out the windows of the call before.
int sio_submit(rfid_ctx *c, int b, int64_t n_new, bool flush) {
  LaTimer tm_submit(11);
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::StreamIO::Pass &ps = io.pass;
  ps = rfid_ctx::StreamIO::Pass();
  ps.b = b; ps.flush = flush;
  ps.n_have = io.tail_len + n_new;          // raw samples available beyond the history
  ps.n_out = ps.n_have / io.dec();
  const int64_t n_have = ps.n_have, n_out = ps.n_out;
  float2 *data = io.d_buf[b] + (io.tail_max - io.tail_len);   // first held-back (or new) sample
  if (io.ymode) {
    // the stream's samples ARE the matched filter's outputs (gate-keyed look-ahead): the pass reads them where they lie
    c->y_view = data;
    c->d_lens = nullptr;
    c->last_n_raw = DECIM * n_out;
  } else if (n_out > 0) {
    // ---- matched filter over everything available: y[n] = sum x[5n - 24 .. 5n], history in front of `data` ----
    MfArgs a;
    a.x = data - SIO_HIST; a.x_stride = SIO_HIST + n_have; a.n_raw = SIO_HIST + n_have; a.lens = nullptr;
    a.n_out = n_out; a.in_off = SIO_HIST - (NTAPS - 1);
    a.vec_ok = ((((uintptr_t)a.x) & 15) == 0) ? 1 : 0;
    a.y = c->d_y; a.y_stride = c->y_stride; a.tile0 = 0; a.stream0 = 0;
    const int64_t tiles = (n_out + MF_TILE - 1) / MF_TILE;
    hipLaunchKernelGGL(mf_boxcar25_decim5_kernel, dim3((unsigned)tiles, 1), dim3(MF_THREADS), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    c->d_lens = nullptr;
    c->last_n_raw = n_have;
  }
  if (c->la.on) HIPCHK(c, hipEventRecord(io.ev_y, c->stream));
  la_count(14, la_now() - tm_submit.t0);
  if (n_out > 0 && n_out < (c->la.on ? SIO_SMALL_DEC_LA : SIO_SMALL_DEC)) {
    // ---- a short pass (a scheduler's 8 k-item buffer, a small file): the long-stream front end is a string of ~45 launches
    //      that one trace of this length does not repay -- the sequential scan (one launch, ~10 ns per sample) goes over it
    //      from the carried state, up to one EPC window before the end of what is there (a window that opens before that
    //      point is complete, rfid_stream_work's rule for what the front end cannot take); decoder and packet right behind ----
    ps.small = true;
    ps.seq_end = flush ? n_out : (n_out - EPC_WIN);
    if (ps.seq_end > 0) {
      HIPCHK(c, hipMemsetAsync(c->d_flat_count, 0, 2 * sizeof(int), c->stream));
      HIPCHK(c, hipMemsetAsync(&c->d_gstate->win_seq, 0, sizeof(int), c->stream));   // windows are numbered per call
      HIPCHK(c, hipMemsetAsync(c->d_wcount, 0, sizeof(int), c->stream));
      c->d_ls2_ctl = nullptr;
      GateArgs g = {};
      g.y = c->y(); g.y_stride = c->y_stride; g.n_dec = n_out; g.lens = nullptr; g.pos0 = 0; g.chunk_len = ps.seq_end;
      g.state = c->d_gstate; g.n_streams = 1; g.wtab = c->d_wtab; g.wmax = c->wmax; g.wcount = c->d_wcount;
      g.flat = c->d_flat; g.flat_count = c->d_flat_count; g.flat_cap = c->flat_cap; g.mode = 0;
      hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(GATE_THREADS), 0, c->stream, g);
      HIPCHK(c, hipGetLastError());
      c->ev_valid[2] = false;
      int rc = rfid_batch_decode(c, 0);
      if (rc) return rc;
      if (c->la.on) {
        ps.n_hdr = c->la.n_hdr; ps.usual = (ps.n_hdr / 2 + 1) * (EPC_WIN + RN16_WIN);
        if ((rc = sio_enqueue_packet(c, ps.n_hdr, ps.usual, nullptr))) return rc;
      }
      ps.prefetched = true;   // (decoded -- and, with the look-ahead, packed -- behind the scan)
    }
  } else if (n_out > 0) {
    // ---- gate: the pieces up to the last idle cut, from the carried state (the long-stream front end) ----
    LsOpts opt;
    opt.carry = true; opt.hold_last = !flush; opt.force = true;
    int enq = 0;
    const double t_ls0 = la_now();
    int rc = ls_enqueue(c, n_out, opt, &enq);
    la_count(13, la_now() - t_ls0);
    if (rc) return rc;
    ps.enq = enq != 0;
    if (ps.enq && c->la.on) {
      // look-ahead: nearly every pass ends with the front end's tables -- decode them and pack the results right behind
      // it (a pass that ends otherwise is decoded and packed again by sio_collect)
      c->ev_valid[2] = false;
      const double t_d0 = la_now();
      if ((rc = rfid_batch_decode(c, 0))) return rc;
      const double t_d1 = la_now();
      la_count(15, t_d1 - t_d0);
      ps.n_hdr = c->la.n_hdr; ps.usual = (ps.n_hdr / 2 + 1) * (EPC_WIN + RN16_WIN);   // (the types alternate)
      if ((rc = sio_enqueue_packet(c, ps.n_hdr, ps.usual, &c->d_ls2_ctl->ok))) return rc;   // (packed only if the front end made the tables)
      la_count(16, la_now() - t_d1);
      ps.prefetched = true;
    }
  }
  ps.active = true;
  return RFID_OK;
}

int sio_collect(rfid_ctx *c) {
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::StreamIO::Pass &ps = io.pass;
  if (!ps.active) return RFID_OK;
  LaTimer tm_collect(10);
  ps.active = false;
  const int b = ps.b;
  const bool flush = ps.flush;
  const int64_t n_have = ps.n_have, n_out = ps.n_out;
  float2 *data = io.d_buf[b] + (io.tail_max - io.tail_len);
  int64_t consumed = 0;                                 // decimated samples processed
  int n_windows = 0;
  {
    const double t_s0 = la_now();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    la_count(7, la_now() - t_s0);
  }
  ls_note_last_pass(c);
  if (n_out > 0) {
    bool ok = false;
    if (ps.small) {
      ok = ps.seq_end > 0;
      consumed = ok ? ps.seq_end : 0;
    } else if (ps.enq) {
      ok = c->ls2_host->ok != 0;
      if (ok) consumed = flush ? n_out : *(const int *)((const char *)c->ls2_host + sizeof(Ls2Ctl));
      if (ok && consumed <= 0) ok = false;
    } else {
      HIPCHK(c, hipMemsetAsync(c->d_flat_count, 0, 2 * sizeof(int), c->stream));
    }
    // What the front end could not take -- no idle cut in what is available (a silent or noise-only stretch, not a Gen2
    // trace), rounds exhausted, or so much behind the last cut that it would not fit the hold-back area -- goes through
    // the plain sequential scan from the carried state, up to EPC_WIN samples before the end of what is available: a
    // window that opens there is complete within these samples, so its record is written (only complete windows are
    // recorded, as the decoder would only ever see those: tag_decoder_impl.cc:223,291) and the next call resumes
    // inside it, the gate open.  At the end of the stream: everything.
    if (!ok) consumed = 0;
    const int64_t seq_end = flush ? n_out : (n_out - EPC_WIN);
    const bool tail_too_long = io.dec() * (n_out - consumed) + (n_have - io.dec() * n_out) + io.hist() > io.tail_max;
    bool prefetched = ps.prefetched && ok;   // the packet behind the pass holds the front end's windows
    if (!ps.small && (!ok || tail_too_long) && seq_end > consumed) {
      if (!ok) {
        HIPCHK(c, hipMemsetAsync(c->d_flat_count, 0, 2 * sizeof(int), c->stream));
        HIPCHK(c, hipMemsetAsync(&c->d_gstate->win_seq, 0, sizeof(int), c->stream));   // windows are numbered per call
        HIPCHK(c, hipMemsetAsync(c->d_wcount, 0, sizeof(int), c->stream));
      }
      GateArgs g = {};
      g.y = c->y(); g.y_stride = c->y_stride; g.n_dec = n_out; g.lens = nullptr; g.pos0 = consumed; g.chunk_len = seq_end - consumed;
      g.state = c->d_gstate; g.n_streams = 1; g.wtab = c->d_wtab; g.wmax = c->wmax; g.wcount = c->d_wcount;
      g.flat = c->d_flat; g.flat_count = c->d_flat_count; g.flat_cap = c->flat_cap; g.mode = 0;
      hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(GATE_THREADS), 0, c->stream, g);
      HIPCHK(c, hipGetLastError());
      consumed = seq_end;
      ok = true;
      prefetched = false;   // (more windows than the packet behind the pass knows: decode and pack again)
    }
    if (ok) {
      // ---- decode what the gate found (unless that is done), fetch it ----
      int rc;
      if (!prefetched) {
        c->ev_valid[2] = false;
        if ((rc = rfid_batch_decode(c, 0))) return rc;
      }
      int wc = 0;
      std::vector<rfid_window> w;
      std::vector<rfid_decode_result> r;
      std::shared_ptr<rfid_ctx::LookAhead::Blk> blk;
      LaTimer tm_fetch(12);
      if (c->la.on) {
        // a pass with more windows (or more gated samples) than the packet was sized for is fetched again with the right sizes
        int n_hdr = prefetched ? ps.n_hdr : c->la.n_hdr;
        int usual = prefetched ? ps.usual : (n_hdr / 2 + 1) * (EPC_WIN + RN16_WIN);   // (the types alternate)
        for (int attempt = 0; attempt < 2; ++attempt) {
          const size_t hdr = GATED_HDR + (sizeof(rfid_window) + sizeof(rfid_decode_result)) * (size_t)n_hdr;
          if (!(attempt == 0 && prefetched)) {
            if ((rc = sio_enqueue_packet(c, n_hdr, usual, nullptr))) return rc;
            HIPCHK(c, hipStreamSynchronize(c->stream));
          }
          wc = *(const int *)c->la.h_pack;
          if (wc > c->wmax) wc = c->wmax;
          const rfid_window *hw = (const rfid_window *)(c->la.h_pack + GATED_HDR);
          const rfid_decode_result *hr = (const rfid_decode_result *)(hw + n_hdr);
          size_t need = 0;
          for (int i = 0; i < wc && i < n_hdr; ++i) need += hw[i].type ? EPC_WIN : RN16_WIN;
          if (wc <= n_hdr && need <= (size_t)usual) {
            if (wc + wc / 4 + 8 > c->la.n_hdr) c->la.n_hdr = wc + wc / 4 + 8;   // (the next calls hold about as many)
            w.assign(hw, hw + wc); r.assign(hr, hr + wc);
            const rfid_cf32 *g = (const rfid_cf32 *)(c->la.h_pack + hdr);
            const float *m = (const float *)(g + usual);
            blk = c->la.take_blk();
            blk->g.assign(g, g + need);
            blk->m.assign(m, m + need);
            break;
          }
          if (attempt == 1) return fail(c, RFID_ERR_CAPACITY, "look-ahead: window packet");
          n_hdr = wc + 1; usual = n_hdr * EPC_WIN;   // (rare: more windows than the calls so far held -- all of them, all in the first part)
          c->la.n_hdr = wc + wc / 4 + 8;
        }
      } else {
        HIPCHK(c, hipMemcpyAsync(&wc, c->d_wcount, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
      }
      n_windows = wc;
      if (wc > 0) {
        if (!c->la.on) {
          w.assign((size_t)wc, rfid_window());
          r.assign((size_t)wc, rfid_decode_result());
          HIPCHK(c, hipMemcpyAsync(w.data(), c->d_wtab, sizeof(rfid_window) * (size_t)wc, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipMemcpyAsync(r.data(), c->d_res, sizeof(rfid_decode_result) * (size_t)wc, hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        const int64_t n0 = io.raw_base / io.dec();
        size_t goff = 0;
        for (int i = 0; i < wc; ++i) {
          const rfid_window &wi = w[(size_t)i];
          const int len = wi.type ? EPC_WIN : RN16_WIN;
          if (c->la.on) {
            // (the READER_STATE bookkeeping is done by the gate / decoder / reader calls that consume this window)
            c->la.wins.emplace_back();
            rfid_ctx::LookAhead::Win &q = c->la.wins.back();
            q.start = n0 + wi.start; q.type = wi.type; q.len = len; q.res = r[(size_t)i];
            q.blk = blk; q.off = goff;
            q.first = blk->g[goff]; q.last = blk->g[goff + (size_t)len - 1];
            goff += (size_t)len;
            continue;
          }
          if (!sio_account(c, r[(size_t)i])) break;   // TERMINATED: the gate swallows the rest (gate_impl.cc:125,198)
          rfid_stream_window sw;
          sw.start = n0 + wi.start; sw.type = wi.type; sw.reserved_ = 0;
          sw.dc_re = wi.dc_re; sw.dc_im = wi.dc_im;
          io.out_w.push_back(sw);
          io.out_r.push_back(r[(size_t)i]);
        }
      }
    }
  }
  // ---- what was not processed moves in front of the other buffer's upload area, history included ----
  const int64_t left = n_have - io.dec() * consumed;
  if (left + io.hist() > io.tail_max)   // (cannot happen: the sequential scan above leaves EPC_WIN samples at most)
    return fail(c, RFID_ERR_CAPACITY, "rfid_stream_work: hold-back capacity exceeded");
  const int o = b ^ 1;
  if (left + io.hist() > 0)
    HIPCHK(c, hipMemcpyAsync(io.d_buf[o] + (io.tail_max - left - io.hist()), data + io.dec() * consumed - io.hist(),
                             sizeof(float2) * (size_t)(left + io.hist()), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipEventRecord(io.ev_free[b], c->stream));
  io.tail_len = left;
  io.raw_base += io.dec() * consumed;
  (void)n_windows;
  return RFID_OK;
}

int sio_process(rfid_ctx *c, int b, int64_t n_new, bool flush) {
  int rc = sio_submit(c, b, n_new, flush);
  if (rc) return rc;
  return sio_collect(c);
}
}  // namespace

extern "C" {

}  // extern "C"
namespace {
// rfid_stream_begin; ymode: the stream's samples are matched-filter outputs (max_chunk_raw counts those)
int sio_begin(rfid_ctx *c, int64_t max_chunk_raw, bool ymode) {
  const int dec = ymode ? 1 : DECIM, hist = ymode ? 0 : SIO_HIST;
  if (!c || max_chunk_raw < dec * 4 * LS2_MIN_PIECE) return RFID_ERR_INVALID;   // a chunk must hold a few pieces
  HIPCHK(c, hipSetDevice(c->device));
  rfid_ctx::StreamIO &io = c->sio;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  sio_free(c);
  // held back per call: at most a few (stretched) pieces of the largest call's grid -- more goes through the sequential scan
  int64_t piece = (max_chunk_raw / dec + LS2_TARGET_PIECES - 1) / LS2_TARGET_PIECES;
  if (piece < LS2_MIN_PIECE) piece = LS2_MIN_PIECE;
  // (three steps of the idle-cut grid when calls are long: an inventory round has one idle stretch, a grid point may miss
  // it; what does not fit goes through the sequential scan, which a short call can afford and a long one cannot)
  const int64_t hold = (max_chunk_raw >= dec * 48 * piece) ? 3 * LS2_FINE * piece : 6 * piece;
  io.tail_max = ((dec * hold + dec * (int64_t)EPC_WIN + hist + 63) & ~63LL);
  io.max_chunk = max_chunk_raw;
  // (the plan's sizes follow the decimated sample count: a ymode stream of N samples is planned like 5 N raw ones)
  c->plan_for_stream = true;     // (a stream's passes do not overlap one another's matched filters: no second output buffer for it)
  int rc = rfid_batch_plan(c, 1, (DECIM / dec) * (io.tail_max + max_chunk_raw));
  c->plan_for_stream = false;
  if (rc) return rc;
  io.ymode = ymode;
  for (int i = 0; i < 2; ++i) {
    if (hipMalloc((void **)&io.d_buf[i], sizeof(float2) * (size_t)(io.tail_max + max_chunk_raw)) != hipSuccess ||
        hipHostMalloc((void **)&io.h_pin[i], sizeof(rfid_cf32) * (size_t)max_chunk_raw, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&io.ev_up[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&io.ev_free[i], hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      sio_free(c);
      return fail(c, RFID_ERR_HIP, "rfid_stream_begin: staging allocation");
    }
    if (hipMemsetAsync(io.d_buf[i], 0, sizeof(float2) * (size_t)io.tail_max, c->stream) != hipSuccess) { sio_free(c); return RFID_ERR_HIP; }
  }
  if (hipStreamCreateWithFlags(&io.copy_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&io.ev_y, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&io.ev_hist, hipEventDisableTiming) != hipSuccess) { sio_free(c); return RFID_ERR_HIP; }
  // fresh blocks: gate_impl ctor state (zeros), READER_STATE after START -> SEND_QUERY (reader_impl.cc:218-288)
  init_reader_state(c);
  c->rs.n_queries_sent = 1;
  c->rs.gen2_logic_status = RFID_IDLE;
  HIPCHK(c, hipMemsetAsync(c->d_gstate, 0, sizeof(GateState), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  io.cur = 0; io.pending = false; io.pend_new = 0; io.tail_len = 0; io.raw_base = 0; io.acc_new = 0;
  io.out_w.clear(); io.out_r.clear();
  io.open = true;
  return RFID_OK;
}
}  // namespace
extern "C" {
int rfid_stream_begin(rfid_ctx *c, int64_t max_chunk_raw) { return sio_begin(c, max_chunk_raw, false); }

int rfid_stream_staging(rfid_ctx *c, int idx, rfid_cf32 **host, int64_t *cap) {
  if (!c || idx < 0 || idx > 1 || !host) return RFID_ERR_INVALID;
  if (!c->sio.open) return RFID_ERR_STATE;
  *host = c->sio.h_pin[idx];
  if (cap) *cap = c->sio.max_chunk;
  return RFID_OK;
}

int rfid_stream_work(rfid_ctx *c, const rfid_cf32 *raw, int64_t n_raw, int flush, rfid_stream_window *windows,
                     rfid_decode_result *results, int64_t cap, int64_t *n_out) {
  if (!c || n_raw < 0 || (n_raw > 0 && !raw) || !n_out || cap < 0) return RFID_ERR_INVALID;
  rfid_ctx::StreamIO &io = c->sio;
  if (!io.open || io.failed) return RFID_ERR_STATE;
  if (n_raw > io.max_chunk) return RFID_ERR_CAPACITY;
  HIPCHK(c, hipSetDevice(c->device));
  *n_out = 0;
  // ---- 1. start the upload of the new samples ----
  int up_idx = -1;
  if (n_raw > 0) {
    up_idx = io.cur;
    const rfid_cf32 *src = raw;
    bool pinned = (raw == io.h_pin[0] || raw == io.h_pin[1]);
    if (!pinned) {   // page-locked memory of the caller's own (hipHostMalloc / hipHostRegister, a torch pinned tensor)?
      hipPointerAttribute_t attr;
      if (hipPointerGetAttributes(&attr, raw) == hipSuccess) pinned = (attr.type == hipMemoryTypeHost);
      else (void)hipGetLastError();
    }
    if (!pinned) {   // ordinary host memory: through the pinned buffer of this slot (its previous upload is long over:
                     // the chunk it carried was processed one call ago -- waited for all the same)
      HIPCHK(c, hipEventSynchronize(io.ev_up[up_idx]));
      memcpy(io.h_pin[up_idx], raw, sizeof(rfid_cf32) * (size_t)n_raw);
      src = io.h_pin[up_idx];
    }
    HIPCHK(c, hipStreamWaitEvent(io.copy_stream, io.ev_free[up_idx], 0));   // the chunk last processed from this buffer is done
    HIPCHK(c, hipMemcpyAsync(io.d_buf[up_idx] + io.tail_max, src, sizeof(rfid_cf32) * (size_t)n_raw, hipMemcpyHostToDevice,
                             io.copy_stream));
    HIPCHK(c, hipEventRecord(io.ev_up[up_idx], io.copy_stream));
    io.cur ^= 1;
  }
  // ---- 2. process the chunk of the previous call while that upload runs ----
  if (io.pending) {
    HIPCHK(c, hipStreamWaitEvent(c->stream, io.ev_up[io.pend_idx], 0));
    int rc = sio_process(c, io.pend_idx, io.pend_new, false);
    if (rc) { io.failed = true; return rc; }
    io.pending = false;
  }
  if (n_raw > 0) { io.pending = true; io.pend_idx = up_idx; io.pend_new = n_raw; }
  // ---- 3. end of stream: the new chunk too, and whatever is still held back ----
  if (flush) {
    if (io.pending) {
      HIPCHK(c, hipStreamWaitEvent(c->stream, io.ev_up[io.pend_idx], 0));
      int rc = sio_process(c, io.pend_idx, io.pend_new, true);
      if (rc) { io.failed = true; return rc; }
      io.pending = false;
    } else if (io.tail_len > 0) {
      int rc = sio_process(c, io.cur, 0, true);   // the held-back samples sit in front of the next upload area
      if (rc) { io.failed = true; return rc; }
    }
  }
  // ---- 4. deliver ----
  const int64_t have = (int64_t)io.out_w.size();
  if (have > cap || (have > 0 && (!windows || !results))) { *n_out = have; return RFID_ERR_CAPACITY; }
  if (have > 0) {
    memcpy(windows, io.out_w.data(), sizeof(rfid_stream_window) * (size_t)have);
    memcpy(results, io.out_r.data(), sizeof(rfid_decode_result) * (size_t)have);
  }
  io.out_w.clear(); io.out_r.clear();
  *n_out = have;
  return RFID_OK;
}

int rfid_stream_end(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->sio.copy_stream) (void)hipStreamSynchronize(c->sio.copy_stream);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  sio_free(c);
  return RFID_OK;
}

}  // extern "C"

// ======================================================================================
// (1c) look-ahead of the per-block calls (rfid_lookahead_enable)
// ======================================================================================
namespace {
void la_free(rfid_ctx *c) {
  rfid_ctx::LookAhead &la = c->la;
  if (la.on && c->stream) (void)hipStreamSynchronize(c->stream);   // (a submitted pass may still be copying into these buffers)
  if (la.h_pack) (void)hipHostFree(la.h_pack);
  if (la.h_y) (void)hipHostFree(la.h_y);
  if (la.h_flag) (void)hipHostFree(la.h_flag);
  if (la.d_done) (void)hipFree(la.d_done);
  la.wins.clear(); la.dq.clear();   // (their blocks go back to the pool, which is emptied next)
  for (rfid_ctx::LookAhead::Blk *b : la.pool) delete b;
  la.pool.clear();
  la = rfid_ctx::LookAhead();
}

// ---- round 5: the calls gather, the passes are big ---------------------------------------------------------------------
// Until round 4 every rfid_mf_work (gate-keyed: rfid_gate_work) call ran a whole-chain pass over its own buffer: at GNU
// Radio's default of 8 192 items per buffer that is a launch list and a packet fetch per 20 ms of signal -- 167 Msamples/s,
// a third of one CPU core.  Now a call only UPLOADS its new samples behind what the device already holds (and, keyed on
// the filter, filters them: its outputs are what the call returns) on the copy stream; a pass over everything pending
// is submitted once LookAhead::coalesce decimated samples have gathered (65 536 unless the adaptor knows the scheduler's
// buffers to be smaller, rfid_lookahead_set_coalesce), and runs on the main stream while the next calls upload into the
// other buffer.  The gate's answers come from the passes as before; what changes is when a gate call that can decide
// nothing makes the device decide: at once when it is shown 2 x coalesce items or more (a bounded buffer must drain),
// and at the latest when it is asked a second time without anything new having arrived (the input has paused, or
// ended: a scheduler calls a block again when its upstream neighbour is done) -- then the pending pass is collected,
// a pass goes over what is still pending, and what even that leaves undecided (the stretch behind the last idle cut)
// goes through the EXACT per-call scan (la_exact_step: the streaming form of gate_scan_kernel, the reference's loop sample
// by sample from the carried state).  So a gate call returns (0, 0) at most once per arrival of new samples, every
// sample is consumed whether or not anybody calls rfid_lookahead_flush, and the end of the input needs no announcement.

// uploads n new stream samples (raw ones, or filter outputs when keyed on the gate) behind what is pending in d_buf[cur]
int la_submit_pending(rfid_ctx *c);
// staged != nullptr: no transfer is queued -- *staged is where the samples lie in page-locked memory (the caller's kernel reads
// them from there and puts them into d_buf itself, and records ev_up behind it)
int la_append(rfid_ctx *c, const rfid_cf32 *src, int64_t n, const rfid_cf32 **staged = nullptr) {
  rfid_ctx::StreamIO &io = c->sio;
  if (io.acc_new + n > io.max_chunk) {   // no room behind what is pending: that goes into a pass first
    const int rc = la_submit_pending(c);
    if (rc) return rc;
  }
  const int up = io.cur;
  if (io.acc_new == 0) {
    HIPCHK(c, hipEventSynchronize(io.ev_free[up]));                       // the pass that last ran on this buffer has been collected
    HIPCHK(c, hipStreamWaitEvent(io.copy_stream, io.ev_hist, 0));         // ... and the history in front of its upload area is in place
  }
  // page-locked memory of the caller's (rfid_host_alloc, hipHostMalloc / hipHostRegister)?  then no staging copy -- but only where the
  // call does not return before the device has read the samples: a call that takes `staged` waits for its own filter outputs,
  // which lie behind the upload.  A gate-keyed call, one with late outputs and one that produces no output return at once, and
  // the scheduler may reuse its buffer.  Asked of the runtime every time (a device-side read of a pageable address is a fault, not
  // a slow copy: no remembered verdict), and the kernel gets the DEVICE's address of the range (a registered range may have another)
  bool pinned = false;
  if (!io.ymode && !c->la.late && staged) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost && attr.devicePointer) {
      const void *last = (const void *)(src + n - 1);
      hipPointerAttribute_t attr2;   // (the whole range, not just its first sample)
      if (hipPointerGetAttributes(&attr2, last) == hipSuccess && attr2.type == hipMemoryTypeHost && attr2.devicePointer &&
          (const char *)attr2.devicePointer - (const char *)attr.devicePointer == (const char *)last - (const char *)src) {
        pinned = true;
        src = (const rfid_cf32 *)attr.devicePointer;
      } else (void)hipGetLastError();
    } else (void)hipGetLastError();
  }
  if (!pinned) { memcpy(io.h_pin[up] + io.acc_new, src, sizeof(rfid_cf32) * (size_t)n); src = io.h_pin[up] + io.acc_new; }
  if (staged) *staged = src;
  else {
    if (c->knobs.la_upload_kernel && n <= (1 << 22)) {   // (a launch that reads the page-locked samples over the bus: less host time than a transfer)
      UploadArgs ua;
      ua.src = (const float2 *)src; ua.dst = io.d_buf[up] + io.tail_max + io.acc_new; ua.n = (int)n;
      hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, io.copy_stream, ua);
      HIPCHK(c, hipGetLastError());
    } else {
      HIPCHK(c, hipMemcpyAsync(io.d_buf[up] + io.tail_max + io.acc_new, src, sizeof(rfid_cf32) * (size_t)n, hipMemcpyHostToDevice, io.copy_stream));
    }
    HIPCHK(c, hipEventRecord(io.ev_up[up], io.copy_stream));
  }
  io.acc_new += n;
  return RFID_OK;
}
// a pass over everything the device holds and has not decided: the held-back stretch + what the calls have uploaded since
int la_submit_pending(rfid_ctx *c) {
  rfid_ctx::StreamIO &io = c->sio;
  if (io.pass.active) {                     // (its held-back stretch is what this pass starts with)
    const int rc = sio_collect(c);
    if (rc) { io.failed = true; return rc; }
  }
  const int up = io.cur;
  if (io.acc_new > 0) HIPCHK(c, hipStreamWaitEvent(c->stream, io.ev_up[up], 0));
  // the filter history of the calls that follow goes in front of the other buffer's upload area right away (sio_collect
  // copies the held-back stretch there later: the same samples), so that they need not wait for this pass
  if (io.hist() > 0)
    HIPCHK(c, hipMemcpyAsync(io.d_buf[up ^ 1] + (io.tail_max - io.hist()), io.d_buf[up] + (io.tail_max + io.acc_new - io.hist()),
                             sizeof(float2) * (size_t)io.hist(), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipEventRecord(io.ev_hist, c->stream));
  const int rc = sio_submit(c, up, io.acc_new, false);
  if (rc) { io.failed = true; return rc; }
  io.acc_new = 0;
  io.cur ^= 1;
  return RFID_OK;
}

// enough has gathered for a pass -- and the pass before is through (or there is none), or the staging is nearly full
bool la_should_submit(rfid_ctx *c) {
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  const int64_t have = io.acc_new / io.dec();
  if (la.flushed || la.exact_open || have < la.coalesce) return false;
  if (!io.pass.active) return true;
  if (hipStreamQuery(c->stream) == hipSuccess) return true;
  (void)hipGetLastError();   // (hipErrorNotReady is no error)
  return have >= 4 * la.coalesce || io.acc_new + io.dec() * 16384 > io.max_chunk;
}

bool same_sample(const rfid_cf32 &a, const rfid_cf32 &b) { return memcmp(&a, &b, sizeof(a)) == 0; }

// rfid_mf_work with the look-ahead on: the call's samples go to the device, its filter outputs come back -- with late
// outputs (rfid_lookahead_set_late_outputs) one call later
namespace {
// spins until the device has written `seq` behind the filter outputs in h_y
int la_wait_flag(rfid_ctx *c, int seq) {
  volatile int *fl = c->la.h_flag;
  long spins = 0;
  while (*fl - seq < 0) {            // (sequence numbers only grow: a later call's flag covers this one's too)
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    __asm__ __volatile__("" ::: "memory");
#endif
    if (++spins > 2000000L) {        // (~ tens of ms: something is wrong or very slow -- wait the ordinary way)
      HIPCHK(c, hipStreamSynchronize(c->sio.copy_stream));
      break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return RFID_OK;
}
}  // namespace
int la_mf_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_produced) {
  LaTimer tm(0);
  typedef rfid_ctx::LookAhead::Held Held;
  const int SLOTS = rfid_ctx::LookAhead::LATE_SLOTS;
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  const bool deliver_only = la.late && n_in == 0;   // (what is held back is handed out after the end of the stream too)
  if (!io.open || io.failed || (la.flushed && !deliver_only))
    return fail(c, RFID_ERR_STATE, "look-ahead: the stream has ended (rfid_ctx_reset starts a new one)");
  if (n_in > io.max_chunk) return fail(c, RFID_ERR_CAPACITY, "look-ahead: rfid_mf_work call larger than the max_chunk_raw given to rfid_lookahead_enable");
  const int64_t n_first = c->mf_seen / DECIM;
  const int n_out = (int)((c->mf_seen + n_in) / DECIM - n_first);
  // (h_y is made once, every slot for the largest call the stream takes -- n_in <= max_chunk was checked above: it never grows, so
  // no call has to hand out everything held back first, and rfid_mf_must_fetch's answer is all there is to it)
  const bool grow = n_out > 0 && la.h_y == nullptr;
  // Late outputs: up to SLOTS - 1 sets are held back when a call arrives; it launches its own filter into the free slot first
  // and then hands out whatever the device has finished -- so a call's outputs have two calls' time to get through the device
  // (one call's time was about what they need: the calls still waited ~10 us each).  What MUST be handed out before the launch:
  // the oldest set when all other slots are taken.
  int must = 0;
  if (la.late && n_out > 0) {
    if (grow) must = la.held_total();
    else if ((int)la.held.size() >= SLOTS) must = la.held.front().n - la.held.front().off;
  }
  if (la.late) {
    if (must > out_cap)
      return fail(c, RFID_ERR_CAPACITY, "look-ahead: rfid_mf_work with new samples while the outputs held back do not fit (rfid_mf_pending: fetch them first, n_in = 0)");
    if (!la.held.empty() && !out) return RFID_ERR_CAPACITY;
  } else if (n_out > out_cap || (n_out > 0 && !out)) return RFID_ERR_CAPACITY;
  int rc = RFID_OK;
  const rfid_cf32 *staged = nullptr;   // with outputs to make: the filter's launch fetches the samples itself (mf_upload_kernel)
  if (n_in > 0) {
    rc = la_append(c, in, n_in, n_out > 0 ? &staged : nullptr);
    if (rc) { io.failed = true; return rc; }
  }
  la_count(4, la_now() - tm.t0);   // (samples staged)
  int give = 0;
  // hands out (part of) the oldest set; wait: for the device if it is not through with it yet.  -> false: nothing handed out
  auto hand_out_oldest = [&](bool wait, int &err) -> bool {
    err = RFID_OK;
    if (la.held.empty() || give >= out_cap) return false;
    Held &h = la.held.front();
    const rfid_cf32 *from = la.h_y + (size_t)h.slot * (la.h_ycap / (size_t)SLOTS);
    if (!h.ready) {
      if (!wait && *(volatile int *)la.h_flag - h.seq < 0) return false;
      const double t_y0 = la_now();
      err = la_wait_flag(c, h.seq);
      la_count(8, la_now() - t_y0);
      if (err) return false;
      h.ready = true;
      la.y_push(h.y0, from, (size_t)h.n);
    }
    int k = h.n - h.off;
    if (k > out_cap - give) k = out_cap - give;
    memcpy(out + give, from + h.off, sizeof(rfid_cf32) * (size_t)k);
    give += k;
    h.off += k;
    if (h.off == h.n) la.held.pop_front();
    return true;
  };
  while (must > 0) {                     // (must <= out_cap: these fit)
    const int before = give;
    if (!hand_out_oldest(true, rc)) { if (rc) return rc; break; }
    must -= give - before;
  }
  if (n_out > 0) {
    // y[n] = sum x[5n - 24 .. 5n] for this call's outputs: the matched filter over the new samples, whose history lies in
    // front of them in the buffer, on the copy stream (the pass before may still be at work on the main stream)
    const double t_sp = la_now();
    if (!la.h_flag) {
      HIPCHK(c, hipHostMalloc((void **)&la.h_flag, 64, hipHostMallocDefault));
      *la.h_flag = 0;
      HIPCHK(c, hipMalloc((void **)&la.d_done, 64));
      HIPCHK(c, hipMemsetAsync(la.d_done, 0, 64, io.copy_stream));
    }
    if (grow) {   // SLOTS parts: this call's outputs and those of the calls before (late outputs)
      if (la.h_y) { HIPCHK(c, hipStreamSynchronize(io.copy_stream)); (void)hipHostFree(la.h_y); }
      la.h_y = nullptr; la.h_ycap = 0;
      const size_t want = (size_t)(io.max_chunk / DECIM + 2);
      HIPCHK(c, hipHostMalloc((void **)&la.h_y, sizeof(rfid_cf32) * want * (size_t)SLOTS, hipHostMallocDefault));
      la.h_ycap = want * (size_t)SLOTS;
    }
    int slot = 0;
    if (la.late) {
      bool used[8] = {false, false, false, false, false, false, false, false};
      for (const Held &h : la.held) used[h.slot] = true;
      while (slot < SLOTS && used[slot]) ++slot;
      if (slot >= SLOTS) return fail(c, RFID_ERR_STATE, "look-ahead: no free slot for the filter outputs");
    }
    rfid_cf32 *y_here = la.h_y + (size_t)slot * (la.h_ycap / (size_t)SLOTS);
    // ONE launch: the samples come out of page-locked memory, go into the buffer the passes read and through the filter;
    // the outputs go straight into page-locked host memory (the device writes it over the bus: no copy to set up), a word
    // behind them says they are there, and the host spins on that word (an event's wake-up costs more than the filter).
    // (Until the middle of round 5: a transfer, the filter's launch and a launch for the word -- the hand-over from the copy
    // engine to the kernel queue alone was ~15 of the ~40 us a call's samples took through the device.)
    MfUploadArgs a;
    a.src = (const float2 *)staged;
    a.x = io.d_buf[io.cur] + io.tail_max + (io.acc_new - n_in) - SIO_HIST;     // (raw sample mf_seen - 28)
    a.hist = SIO_HIST; a.n_new = n_in;
    a.n_out = n_out; a.in_off = (int)(DECIM * n_first - c->mf_seen) + SIO_HIST - (NTAPS - 1);   // 0 .. 4
    a.y = (float2 *)y_here;
    a.done = la.d_done; a.flag = la.h_flag;
    const int seq = ++la.flag_seq;
    a.seq = seq;
    const int tiles = (n_out + MF_TILE - 1) / MF_TILE;
    hipLaunchKernelGGL(mf_upload_kernel, dim3((unsigned)tiles), dim3(MF_THREADS), 0, io.copy_stream, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(io.ev_up[io.cur], io.copy_stream));
    la_count(5, la_now() - t_sp);
    if (la.late) {
      Held h;
      h.y0 = n_first; h.n = n_out; h.off = 0; h.seq = seq; h.slot = slot; h.ready = false;
      la.held.push_back(h);
    } else {
      const double t_y0 = la_now();
      rc = la_wait_flag(c, seq);
      if (rc) return rc;
      la_count(8, la_now() - t_y0);
      memcpy(out, y_here, sizeof(rfid_cf32) * (size_t)n_out);
      la.y_push(n_first, y_here, (size_t)n_out);
      give = n_out;
    }
  }
  if (la.late) {
    // whatever the device has finished, oldest first, as far as the room goes; a call that brought nothing and has handed out
    // nothing yet waits for the oldest set (a scheduler that calls a block must see it move)
    for (;;) {
      const bool wait = n_in == 0 && give == 0;
      if (!hand_out_oldest(wait, rc)) { if (rc) return rc; break; }
    }
  }
  c->mf_seen += n_in;
  if (n_in > 0 || give > 0) la.stall = 0;   // (outputs handed out late are new input for the gate: it may answer (0, 0) once more)
  if (n_in > 0) la.tail_tried = false;
  if (la_should_submit(c)) {
    const double t_c0 = la_now();
    rc = la_submit_pending(c);
    la_count(6, la_now() - t_c0);
    if (rc) return rc;
  }
  *n_produced = give;
  return RFID_OK;
}

// The exact per-call scan over what the device holds undecided (the held-back stretch and what has been uploaded behind
// it), at most the n_in samples the gate was shown: gate_scan_kernel in its streaming form from the carried gate state --
// gate_impl.cc:127-196 sample by sample, stopping behind a window that closes (:189-194) -- the gated samples fetched, the
// stream's base moved behind what was consumed.  This is what rfid_gate_work does without the look-ahead, on the
// look-ahead's buffers and state.
int la_exact_step(rfid_ctx *c, int n_in, rfid_cf32 *out, int *consumed, int *written, bool *open_after) {
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  rfid_reader_state &rs = c->rs;
  *consumed = 0; *written = 0; *open_after = false;
  const int b = io.cur;
  if (io.acc_new > 0) HIPCHK(c, hipStreamWaitEvent(c->stream, io.ev_up[b], 0));
  // The carried state stands at the passes' frontier.  The gate itself may be further on: a pass that stopped inside a window
  // (the sequential scan goes up to one EPC window before the end of what it has; a window that opens before that point is
  // complete and on record) has left the gate OPEN there, and the gate call that handed the window out has consumed up to its
  // end.  Those samples are scanned first with their output thrown away: the window runs out exactly at the gate's position.
  const int64_t skip = la.gate_pos - io.raw_base / io.dec();
  if (skip < 0) return fail(c, RFID_ERR_STATE, "look-ahead: the gate is behind what the passes have decided");
  const int64_t n_avail = (io.tail_len + io.acc_new) / io.dec() - skip;
  const int n_use = (int)((n_avail < n_in) ? n_avail : n_in);
  if (n_use <= 0) return RFID_OK;
  const int n_scan = (int)skip + n_use;
  float2 *data = io.d_buf[b] + (io.tail_max - io.tail_len);
  const float2 *ysrc = data;
  if (!io.ymode) {
    MfArgs m;
    m.x = data - SIO_HIST; m.x_stride = SIO_HIST + (int64_t)DECIM * n_scan; m.n_raw = m.x_stride; m.lens = nullptr;
    m.n_out = n_scan; m.in_off = SIO_HIST - (NTAPS - 1);
    m.vec_ok = ((((uintptr_t)m.x) & 15) == 0) ? 1 : 0;
    m.y = c->d_y; m.y_stride = c->y_stride; m.tile0 = 0; m.stream0 = 0;
    hipLaunchKernelGGL(mf_boxcar25_decim5_kernel, dim3((unsigned)((n_scan + MF_TILE - 1) / MF_TILE), 1), dim3(MF_THREADS), 0, c->stream, m);
    HIPCHK(c, hipGetLastError());
    ysrc = c->d_y;
  }
  int rc = grow(c, c->s_out, sizeof(float2) * (size_t)(n_scan + 2));
  if (rc) return rc;
  GateArgs a = {};
  a.lens = nullptr; a.pos0 = 0; a.state = c->d_gstate; a.n_streams = 1; a.mode = 1;
  a.gated = (float2 *)c->s_out.p; a.gated_cap = n_scan; a.io = c->d_io;
  int iov[2] = {0, 0}, open_now = 0;
  if (skip > 0) {
    a.y = ysrc; a.y_stride = skip; a.n_dec = skip; a.chunk_len = skip;
    hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(GATE_THREADS), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(iov, c->d_io, sizeof(iov), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&open_now, &c->d_gstate->gate_open, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (iov[0] != (int)skip || open_now) return fail(c, RFID_ERR_STATE, "look-ahead: the window the passes left open does not end where the gate stands");
    // (the decoder and the reader have run since that window was handed out: the gate is armed for what they ask for now)
    hipLaunchKernelGGL(gate_arm_kernel, dim3(1), dim3(64), 0, c->stream, c->d_gstate, rs.n_samples_to_ungate, (rs.n_samples_to_ungate == EPC_WIN) ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    io.raw_base += (int64_t)io.dec() * skip;
    io.tail_len -= (int64_t)io.dec() * skip;
  }
  a.y = ysrc + skip; a.y_stride = n_use; a.n_dec = n_use; a.chunk_len = n_use;
  hipLaunchKernelGGL(gate_scan_kernel, dim3(1), dim3(GATE_THREADS), 0, c->stream, a);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(iov, c->d_io, sizeof(iov), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&open_now, &c->d_gstate->gate_open, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (iov[1] > 0) HIPCHK(c, hipMemcpy(out, c->s_out.p, sizeof(rfid_cf32) * (size_t)iov[1], hipMemcpyDeviceToHost));
  if (iov[1] > 0 && !open_now) {
    // a window has closed: the decoder and the reader run next and arm the gate for the other kind (gate_impl.cc:112-123:
    // n_samples = 0, the window's length) -- here at once, as the batch form of the scan does at the window's last sample
    const int next_type = (rs.n_samples_to_ungate == EPC_WIN) ? 0 : 1;
    hipLaunchKernelGGL(gate_arm_kernel, dim3(1), dim3(64), 0, c->stream, c->d_gstate, next_type ? EPC_WIN : RN16_WIN, next_type);
    HIPCHK(c, hipGetLastError());
  }
  io.raw_base += (int64_t)io.dec() * iov[0];
  io.tail_len -= (int64_t)io.dec() * iov[0];      // (may go below zero: into the samples behind tail_max)
  la.exact_open = open_now != 0;
  *consumed = iov[0]; *written = iov[1]; *open_after = open_now != 0;
  return RFID_OK;
}

// rfid_gate_work with the look-ahead on (gate_impl.cc:127-199 answered from the windows the whole-chain passes found)
int la_gate_work(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_consumed, int *n_written) {
  LaTimer tm(1);
  (void)out_cap;
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  rfid_reader_state &rs = c->rs;
  *n_consumed = 0; *n_written = 0;
  la.last_m2.clear();
  const int64_t p = la.gate_pos;
  if (!io.ymode) {
    // the input must be the matched filter's output at the gate's position
    const rfid_cf32 *y_first = la.y_at(p), *y_last = la.y_at(p + n_in - 1);
    if (!y_first || !y_last || !same_sample(in[0], *y_first) || !same_sample(in[n_in - 1], *y_last))
      return fail(c, RFID_ERR_STATE, "look-ahead: rfid_gate_work was not handed the matched filter's output at the gate's position");
  } else if (p + n_in > la.up_end) {
    // gate-keyed: the filter is somebody else's; whatever of this call's input the device has not seen yet (the scheduler
    // shows unconsumed samples again, the new ones come behind them) is uploaded -- at most max_chunk samples per call,
    // the rest when it is shown again
    if (!io.open || io.failed || la.flushed) return fail(c, RFID_ERR_STATE, "look-ahead: the stream has ended (rfid_ctx_reset starts a new one)");
    if (la.up_end < p) return fail(c, RFID_ERR_STATE, "look-ahead: the gate's input skipped samples");
    int64_t n_new = p + n_in - la.up_end;
    if (n_new > io.max_chunk) n_new = io.max_chunk;
    int rc = la_append(c, in + (la.up_end - p), n_new);
    if (rc) { io.failed = true; return rc; }
    la.up_end += n_new;
    la.stall = 0;
    la.tail_tried = false;
    if (la_should_submit(c) && (rc = la_submit_pending(c))) return rc;
  }
  for (int attempt = 0; attempt < 8; ++attempt) {
    const int64_t frontier = io.raw_base / io.dec();   // the gate's doing is known for the samples before this position
    int consumed = 0, written = 0;
    bool open_after = false;
    while (!la.wins.empty() && la.wins.front().start + la.wins.front().len <= p && la.emitted == 0) la.wins.pop_front();   // (never: windows are consumed in order)
    if (!la.wins.empty() && (la.emitted > 0 || la.wins.front().start < p + n_in)) {
      rfid_ctx::LookAhead::Win &w = la.wins.front();
      if (la.emitted == 0 && w.len != rs.n_samples_to_ungate)
        return fail(c, RFID_ERR_STATE, "look-ahead: the window the gate was armed for is not the next one of the RN16 / EPC alternation");
      const int64_t from = w.start + la.emitted;           // first sample of the window still to hand out (>= p)
      const int64_t upto = (w.start + w.len < p + n_in) ? (w.start + w.len) : (p + n_in);
      written = (int)(upto - from);
      memcpy(out, w.blk->g.data() + w.off + la.emitted, sizeof(rfid_cf32) * (size_t)written);
      la.last_m2.assign(w.blk->m.begin() + (long)(w.off + la.emitted), w.blk->m.begin() + (long)(w.off + la.emitted + written));
      la.emitted += written;
      if (la.emitted == w.len) {                           // gate_impl.cc:189-194: closed, consume_each(i + 1)
        consumed = (int)(w.start + w.len - p);
        la.dq.push_back(std::move(w));
        la.dq.back().blk.reset();                          // (the decoder call is recognised by the window's first and last sample)
        la.wins.pop_front();
        la.emitted = 0;
      } else {
        consumed = n_in;
        open_after = true;
      }
    } else if (!la.exact_open) {
      // no opening in [p, p + n_in) as far as the gate's doing is known
      const int64_t lim = (p + n_in < frontier) ? (p + n_in) : frontier;
      consumed = (lim > p) ? (int)(lim - p) : 0;
    }
    if (consumed == 0 && written == 0 && !la.flushed) {
      // Nothing can be decided from what the passes have found so far.  The first such answer since new samples arrived is
      // (0, 0): the scheduler brings more (and the pass that is under way goes on undisturbed).  Asked again without
      // anything new -- the input has paused or ended -- or shown as much as a bounded buffer can hold, the device decides
      // now: the pass under way is waited for, then a pass goes over whatever is pending, then the exact scan takes the rest.
      if (attempt == 0) ++la.stall;
      // (a pass that has finished meanwhile is looked at whoever asks: its windows are what the gate hands out next)
      if (!la.exact_open && io.pass.active && hipStreamQuery(c->stream) == hipSuccess) {
        const int rc = sio_collect(c);
        if (rc) { io.failed = true; return rc; }
        continue;
      }
      (void)hipGetLastError();   // (hipErrorNotReady is no error)
      const bool force = !la.patient || la.stall >= 2 || la.flush_req || n_in >= 2 * la.coalesce || la.exact_open;
      if (!force) break;
      if (!la.exact_open && io.pass.active) {
        const int rc = sio_collect(c);
        if (rc) { io.failed = true; return rc; }
        continue;
      }
      if (!la.exact_open && (io.acc_new > 0 || (!la.tail_tried && io.tail_len / io.dec() > 0))) {
        la.tail_tried = true;
        int rc = la_submit_pending(c);
        if (!rc) rc = sio_collect(c);
        if (rc) { io.failed = true; return rc; }
        continue;
      }
      const int rc = la_exact_step(c, n_in, out, &consumed, &written, &open_after);
      if (rc) { io.failed = true; return rc; }
      if (consumed == 0 && written == 0) break;
    }
    la.stall = 0;
    // keep the state the window's last sample leaves: the dc ring etc. live on the device; here only what the blocks share
    rs.gate_status = open_after ? RFID_GATE_OPEN : RFID_GATE_CLOSED;
    la.gate_pos += consumed;
    la.y_drop_before(la.gate_pos);
    *n_consumed = consumed;
    *n_written = written;
    return RFID_OK;
  }
  return RFID_OK;
}

// rfid_gate_work with the look-ahead on and rfid_lookahead_set_consume_ahead: the gate CONSUMES everything it is shown -- the
// device has the samples (the filter call uploaded them; keyed on the gate: this call does) -- and hands out the windows
// when the passes have found them, one window per call at most and the next one only once the decoder / reader calls have
// armed the gate for it (gate_impl.cc:112-123: the order of the reference's single-threaded flowgraph).  What the gate
// writes is what it writes without this -- the same windows, the same samples --; what changes is that consuming does not
// wait for deciding, so the scheduler's buffer in front of the gate never fills and the passes gather 65 536 samples
// whatever its size.  Nothing is forced here: the end of the input is told by the adaptor, which can see it
// (rfid_gate_forecast, rfid_lookahead_flush).
int la_gate_swallow(rfid_ctx *c, const rfid_cf32 *in, int n_in, rfid_cf32 *out, int out_cap, int *n_consumed, int *n_written) {
  LaTimer tm(1);
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  rfid_reader_state &rs = c->rs;
  *n_consumed = 0; *n_written = 0;
  la.last_m2.clear();
  const int64_t p = la.gate_pos;
  int consumed = 0;
  if (n_in > 0) {
    if (!io.ymode) {
      // the input must be the matched filter's output at the gate's position
      const rfid_cf32 *y_first = la.y_at(p), *y_last = la.y_at(p + n_in - 1);
      if (!y_first || !y_last || !same_sample(in[0], *y_first) || !same_sample(in[n_in - 1], *y_last))
        return fail(c, RFID_ERR_STATE, "look-ahead: rfid_gate_work was not handed the matched filter's output at the gate's position");
      consumed = n_in;
    } else {
      // gate-keyed: everything shown is new (what was shown before was consumed); at most max_chunk samples per call
      if (!io.open || io.failed || la.flushed) return fail(c, RFID_ERR_STATE, "look-ahead: the stream has ended (rfid_ctx_reset starts a new one)");
      if (la.up_end != p) return fail(c, RFID_ERR_STATE, "look-ahead: the gate's input skipped samples");
      const int64_t take = (n_in < io.max_chunk) ? n_in : io.max_chunk;
      int rc = la_append(c, in, take);
      if (rc) { io.failed = true; return rc; }
      la.up_end += take;
      consumed = (int)take;
      if (la_should_submit(c) && (rc = la_submit_pending(c))) return rc;
    }
  }
  // a pass that has finished meanwhile: its windows are what the gate hands out next
  if (io.pass.active && hipStreamQuery(c->stream) == hipSuccess) {
    const int rc = sio_collect(c);
    if (rc) { io.failed = true; return rc; }
  }
  (void)hipGetLastError();   // (hipErrorNotReady is no error)
  int written = 0;
  if (!la.wins.empty()) {
    rfid_ctx::LookAhead::Win &w = la.wins.front();
    bool go = la.emitted > 0;
    if (!go && !la.need_arm) {
      if (w.len != rs.n_samples_to_ungate)
        return fail(c, RFID_ERR_STATE, "look-ahead: the window the gate was armed for is not the next one of the RN16 / EPC alternation");
      go = true;
    }
    if (go) {
      written = w.len - la.emitted;
      if (written > out_cap) written = out_cap;
      memcpy(out, w.blk->g.data() + w.off + la.emitted, sizeof(rfid_cf32) * (size_t)written);
      la.last_m2.assign(w.blk->m.begin() + (long)(w.off + la.emitted), w.blk->m.begin() + (long)(w.off + la.emitted + written));
      la.emitted += written;
      if (la.emitted == w.len) {                           // gate_impl.cc:189-194: closed
        la.dq.push_back(std::move(w));
        la.dq.back().blk.reset();                          // (the decoder call is recognised by the window's first and last sample)
        la.wins.pop_front();
        la.emitted = 0;
        la.need_arm = true;
        rs.gate_status = RFID_GATE_CLOSED;
      } else {
        rs.gate_status = RFID_GATE_OPEN;
      }
    }
  }
  la.gate_pos += consumed;
  la.y_drop_before(la.gate_pos);
  *n_consumed = consumed;
  *n_written = written;
  return RFID_OK;
}

// rfid_decoder_work with the look-ahead on: the result of the window the gate handed out, when `in` is that window
bool la_decoder_result(rfid_ctx *c, const rfid_cf32 *in, int wlen, int type, rfid_decode_result *r) {
  rfid_ctx::LookAhead &la = c->la;
  // the windows are decoded in the order the gate handed them out: the call's window is the oldest entry, or -- if a
  // window was skipped by the caller -- a later one, and what lies before it is stale
  size_t hit = la.dq.size();
  for (size_t i = 0; i < la.dq.size(); ++i) {
    const rfid_ctx::LookAhead::Win &w = la.dq[i];
    if (w.len == wlen && w.type == type && same_sample(in[0], w.first) && same_sample(in[wlen - 1], w.last)) { hit = i; break; }
  }
  if (hit == la.dq.size()) {
    while (la.dq.size() > 64) la.dq.pop_front();   // (a caller that decodes something else altogether: the queue stays bounded)
    return false;
  }
  *r = la.dq[hit].res;
  la.dq.erase(la.dq.begin(), la.dq.begin() + (long)hit + 1);
  return true;
}
}  // namespace

extern "C" {

int rfid_lookahead_flush(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  if (!la.on || la.flushed) return RFID_OK;   // (without look-ahead nothing is held back)
  if (!io.open || io.failed) return fail(c, RFID_ERR_STATE, "rfid_lookahead_flush: the stream has failed or was closed");
  if ((io.ymode && !la.consume_ahead) || la.exact_open) {
    // keyed on the gate: the library has seen only what the gate was shown; the next gate call that brings nothing new and
    // can decide nothing makes the device decide everything it holds (a scheduler shows a block everything its buffer
    // holds).  (Since round 5 such a call does that anyway the second time it is asked: the announcement saves one call.)
    la.flush_req = true;
    return RFID_OK;
  }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = RFID_OK;
  if (io.acc_new > 0) rc = la_submit_pending(c);
  if (!rc) rc = sio_collect(c);
  if (!rc && io.tail_len > 0) {
    rc = sio_process(c, io.cur, 0, true);   // rfid_stream_work's flush: everything available is decided now
  }
  if (rc) { io.failed = true; return rc; }
  la.flushed = true;
  la.stall = 0;
  return RFID_OK;
}

// room for what gathers before a pass (LookAhead::coalesce) and the largest call behind it
// (the staging holds LA_GATHER_MAX x the threshold: while the pass before is still at work the calls go on gathering -- a pass's
// cost is mostly its launch list, so a device that is behind gets fewer, bigger passes instead of a host that waits for it)
static const int64_t LA_COALESCE_DEFAULT = 65536, LA_CALL_ROOM = 16384, LA_GATHER_MAX = 4;

int rfid_lookahead_enable(rfid_ctx *c, int64_t max_chunk_raw) {
  LaTimer tm(9);
  if (!c || max_chunk_raw < 1) return RFID_ERR_INVALID;
  if (c->mf_seen != 0) return fail(c, RFID_ERR_STATE, "rfid_lookahead_enable: the stream has started");
  la_free(c);
  int64_t cap = max_chunk_raw + DECIM * LA_GATHER_MAX * LA_COALESCE_DEFAULT;
  if (cap < DECIM * (LA_GATHER_MAX * LA_COALESCE_DEFAULT + LA_CALL_ROOM)) cap = DECIM * (LA_GATHER_MAX * LA_COALESCE_DEFAULT + LA_CALL_ROOM);
  const rfid_reader_state keep = c->rs;            // rfid_stream_begin sets the whole-chain form's READER_STATE; these calls keep theirs
  int rc = rfid_stream_begin(c, cap);
  c->rs = keep;
  if (rc) return rc;
  c->la.on = true;
  c->la.coalesce = LA_COALESCE_DEFAULT;
  return RFID_OK;
}

int rfid_lookahead_enable_gate(rfid_ctx *c, int64_t max_items) {
  LaTimer tm(9);
  if (!c || max_items < 1) return RFID_ERR_INVALID;
  if (c->mf_seen != 0 || c->la.gate_pos != 0) return fail(c, RFID_ERR_STATE, "rfid_lookahead_enable_gate: the stream has started");
  la_free(c);
  int64_t cap = max_items + LA_GATHER_MAX * LA_COALESCE_DEFAULT;
  if (cap < LA_GATHER_MAX * LA_COALESCE_DEFAULT + LA_CALL_ROOM) cap = LA_GATHER_MAX * LA_COALESCE_DEFAULT + LA_CALL_ROOM;
  const rfid_reader_state keep = c->rs;
  int rc = sio_begin(c, cap, true);
  c->rs = keep;
  if (rc) return rc;
  c->la.on = true;
  c->la.coalesce = LA_COALESCE_DEFAULT;
  return RFID_OK;
}

int rfid_lookahead_drain(rfid_ctx *c) {
  if (!c) return RFID_ERR_INVALID;
  rfid_ctx::StreamIO &io = c->sio;
  rfid_ctx::LookAhead &la = c->la;
  if (!la.on) return RFID_OK;
  HIPCHK(c, hipSetDevice(c->device));
  int rc = RFID_OK;
  if (io.open && !io.failed && !la.flushed) {
    // everything the device holds is decided as at the end of a stream (a window the exact scan has left open is incomplete:
    // it never reaches the decoder, tag_decoder_impl.cc:223,291)
    if (io.acc_new > 0) rc = la_submit_pending(c);
    if (!rc) rc = sio_collect(c);
    if (!rc && !la.exact_open && io.tail_len > 0) {
        rc = sio_process(c, io.cur, 0, true);
    }
    if (rc) { io.failed = true; return rc; }
  }
  // the windows no call will fetch any more: READER_STATE as the decoder / reader calls would have left it
  for (const rfid_ctx::LookAhead::Win &w : la.dq) if (!sio_account(c, w.res)) break;
  for (const rfid_ctx::LookAhead::Win &w : la.wins) if (!sio_account(c, w.res)) break;
  la.dq.clear(); la.wins.clear(); la.emitted = 0;
  la.flushed = true;
  return RFID_OK;
}

int rfid_lookahead_set_scheduler(rfid_ctx *c, int64_t gate_buffer_items) {
  if (!c || gate_buffer_items < 0) return RFID_ERR_INVALID;
  if (!c->la.on) return fail(c, RFID_ERR_STATE, "rfid_lookahead_set_scheduler: the look-ahead is not on");
  if (gate_buffer_items == 0) { c->la.patient = true; c->la.coalesce = LA_COALESCE_DEFAULT; return RFID_OK; }
  c->la.patient = false;
  return rfid_lookahead_set_coalesce(c, gate_buffer_items / 4);
}

int rfid_lookahead_set_consume_ahead(rfid_ctx *c, int on) {
  if (!c) return RFID_ERR_INVALID;
  if (!c->la.on) return fail(c, RFID_ERR_STATE, "rfid_lookahead_set_consume_ahead: the look-ahead is not on");
  if (c->la.gate_pos != 0 || c->la.up_end != 0) return fail(c, RFID_ERR_STATE, "rfid_lookahead_set_consume_ahead: before the first gate call");
  c->la.consume_ahead = on != 0;
  if (on) { c->la.patient = true; c->la.coalesce = LA_COALESCE_DEFAULT; }   // (the gathering does not depend on the scheduler's buffers then)
  return RFID_OK;
}

int rfid_gate_forecast(const rfid_ctx *c, int upstream_done, int *needs_input) {
  if (!c || !needs_input) return RFID_ERR_INVALID;
  *needs_input = 1;
  const rfid_ctx::LookAhead &la = c->la;
  if (!la.on || !la.consume_ahead) return RFID_OK;
  if (c->rs.status != RFID_RUNNING) return RFID_OK;        // (terminated: the gate swallows its input, gate_impl.cc:125,198)
  if (!la.wins.empty()) {
    // a window lies ready: handed out when the gate is armed for it (or being armed by the call: SEEK_*), at once when it is open
    const int st = c->rs.gate_status;
    if (la.emitted > 0 || !la.need_arm || st == RFID_GATE_SEEK_EPC || st == RFID_GATE_SEEK_RN16 || upstream_done) *needs_input = 0;
    return RFID_OK;
  }
  // the input has ended and the device still holds samples nobody has decided about: a call (without input) is told so and decides
  if (upstream_done && !la.flushed && c->sio.open && !c->sio.failed &&
      (c->sio.acc_new > 0 || c->sio.pass.active || c->sio.tail_len > 0)) *needs_input = 0;
  return RFID_OK;
}

int rfid_lookahead_set_late_outputs(rfid_ctx *c, int on) {
  if (!c) return RFID_ERR_INVALID;
  if (!c->la.on || c->sio.ymode) return fail(c, RFID_ERR_STATE, "rfid_lookahead_set_late_outputs: needs the look-ahead keyed on rfid_mf_work");
  if (!c->la.held.empty()) return fail(c, RFID_ERR_STATE, "rfid_lookahead_set_late_outputs: outputs are held back (fetch them first)");
  c->la.late = on != 0;
  return RFID_OK;
}

int rfid_mf_pending(const rfid_ctx *c, int *n_outputs) {
  if (!c || !n_outputs) return RFID_ERR_INVALID;
  *n_outputs = c->la.on ? c->la.held_total() : 0;
  return RFID_OK;
}

int rfid_mf_must_fetch(const rfid_ctx *c, int *n_outputs) {
  if (!c || !n_outputs) return RFID_ERR_INVALID;
  *n_outputs = 0;
  if (c->la.on && (int)c->la.held.size() >= rfid_ctx::LookAhead::LATE_SLOTS) *n_outputs = c->la.held.front().n - c->la.held.front().off;
  return RFID_OK;
}

int rfid_lookahead_set_coalesce(rfid_ctx *c, int64_t items) {
  if (!c || items < 1) return RFID_ERR_INVALID;
  if (!c->la.on) return fail(c, RFID_ERR_STATE, "rfid_lookahead_set_coalesce: the look-ahead is not on");
  const int64_t most = c->sio.max_chunk / c->sio.dec() / 2;
  if (items < 1024) items = 1024;
  if (items > most) items = most;
  c->la.coalesce = items;
  return RFID_OK;
}

int rfid_lookahead_pending(const rfid_ctx *c, int *gate_windows, int *decoder_windows) {
  if (!c) return RFID_ERR_INVALID;
  if (gate_windows) *gate_windows = (int)c->la.wins.size();
  if (decoder_windows) *decoder_windows = (int)c->la.dq.size();
  return RFID_OK;
}

int rfid_abi_version(void) { return RFID_MI355X_ABI; }

void *rfid_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void rfid_host_free(void *p) { if (p) (void)hipHostFree(p); }

int rfid_gate_magn_squared(rfid_ctx *c, float *out, int cap, int *n) {
  if (!c || !n || cap < 0 || (cap > 0 && !out)) return RFID_ERR_INVALID;
  const int k = (int)c->la.last_m2.size();
  *n = k;
  if (k > cap) return RFID_ERR_CAPACITY;
  if (k > 0) memcpy(out, c->la.last_m2.data(), sizeof(float) * (size_t)k);
  return RFID_OK;
}

}  // extern "C"

