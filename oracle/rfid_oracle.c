/*
 * rfid_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See rfid_oracle.h for scope, usage rules and the "parity unpinned" statement.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/gr-rfid/).  Arithmetic types and operation order follow the
 * reference expression by expression; build with -O2/-O3 -ffp-contract=off and
 * no -march / -ffast-math.
 */
#include "rfid_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- constants: include/rfid/global_vars.h:72-143 ---------------------------------- */
#define T1_D 240
#define PW_D 12
#define NUM_PULSES_COMMAND 5
#define TAG_PREAMBLE_BITS 6
#define RN16_BITS 17
#define EPC_BITS 129
#define T_READER_FREQ 40000 /* const int T_READER_FREQ = 40e3 */
static const int TAG_PREAMBLE[12] = {1, 1, 0, 1, 0, 0, 1, 0, 0, 0, 1, 1}; /* :136 */
static const float THRESH_FRACTION = 0.75f;                                /* :139 */
#define WIN_SIZE_D 250
#define DC_SIZE_D 120

/* const float TAG_BIT_D = 1.0/T_READER_FREQ * pow(10,6)   (global_vars.h:111) */
static float tag_bit_d(void) { return (float)(1.0 / T_READER_FREQ * pow(10, 6)); }

static orc_cf cf(float re, float im) { orc_cf z; z.re = re; z.im = im; return z; }
static orc_cf cadd(orc_cf a, orc_cf b) { return cf(a.re + b.re, a.im + b.im); }
static orc_cf csub(orc_cf a, orc_cf b) { return cf(a.re - b.re, a.im - b.im); }
/* std::complex<float> / std::complex<float>(c,0): libgcc __divsc3 reduces to the
 * correctly rounded elementwise quotient (checked in tests/test_toolchain_semantics.py) */
static orc_cf cdiv_real(orc_cf a, float c) { return cf(a.re / c, a.im / c); }
/* std::norm(std::complex<float>) in libstdc++ 11: x*x + y*y */
static float cnorm(orc_cf a) { return a.re * a.re + a.im * a.im; }
/* std::abs(std::complex<float>) -> cabsf -> glibc 2.35 hypotf, which equals this
 * double-precision formula bit for bit (same test) */
static float cabs_f(orc_cf a) {
  return (float)sqrt((double)a.re * (double)a.re + (double)a.im * (double)a.im);
}

void orc_default_config(orc_config *cfg) {
  cfg->fixed_q = 0;
  cfg->max_num_queries = 1000;
  cfg->number_unique_tags = 100;
}

/* lib/global_vars.cc:34-54 */
void orc_initialize_reader_state(orc_reader_state *rs, const orc_config *cfg) {
  memset(rs, 0, sizeof(*rs));
  rs->cfg = *cfg;
  rs->n_queries_sent = 0;
  rs->n_epc_correct = 0;
  rs->status = ORC_RUNNING;
  rs->gen2_logic_status = ORC_START;
  rs->gate_status = ORC_GATE_SEEK_RN16;
  rs->decoder_status = ORC_DECODE_RN16;
  rs->max_slot_number = (int)pow(2, cfg->fixed_q);
  rs->cur_inventory_round = 1;
  rs->cur_slot_number = 1;
}

/* ---- a1: matched filter, apps/reader.py:65,75 (third-party arithmetic; canonical
 * order defined by this repo: k ascending from a zero accumulator) ------------------- */
long orc_fir_boxcar25_decim5(const orc_cf *x, long n_in, orc_cf *y) {
  long n_out = n_in / 5;
  for (long n = 0; n < n_out; n++) {
    orc_cf acc = cf(0.0f, 0.0f);
    for (int k = 0; k < 25; k++) {
      long idx = 5 * n - 24 + k;
      if (idx >= 0) acc = cadd(acc, x[idx]);
      else acc = cadd(acc, cf(0.0f, 0.0f));
    }
    y[n] = acc;
  }
  return n_out;
}

long orc_fir_stream(const orc_cf *x, long n_in, orc_cf *y, orc_cf hist[24], int *phase) {
  /* Outputs fall on raw indices that are multiples of 5 counted from stream start;
   * *phase = (raw samples consumed so far) mod 5.  An output is produced at the
   * raw sample whose global index g satisfies g % 5 == 0. */
  long n_out = 0;
  for (long i = 0; i < n_in; i++) {
    if (*phase == 0) {
      orc_cf acc = cf(0.0f, 0.0f);
      for (int k = 0; k < 25; k++) {
        long idx = i - 24 + k;
        orc_cf v = (idx >= 0) ? x[idx] : hist[24 + idx];
        acc = cadd(acc, v);
      }
      y[n_out++] = acc;
    }
    *phase = (*phase + 1) % 5;
  }
  /* roll history */
  if (n_in >= 24) {
    memcpy(hist, x + n_in - 24, 24 * sizeof(orc_cf));
  } else if (n_in > 0) {
    memmove(hist, hist + n_in, (24 - n_in) * sizeof(orc_cf));
    memcpy(hist + 24 - n_in, x, n_in * sizeof(orc_cf));
  }
  return n_out;
}

/* ---- a2: gate_impl ctor, lib/gate_impl.cc:41-70 ------------------------------------ */
void orc_gate_init(orc_gate *g, int sample_rate) {
  memset(g, 0, sizeof(*g));
  g->n_samples = 0; g->win_index = 0; g->dc_index = 0; g->num_pulses = 0;
  g->signal_state = 0; g->avg_ampl = 0; g->dc_est = cf(0, 0);
  /* int = int * (int / double) -> double -> truncated            (:48-50) */
  g->n_samples_T1 = (int)(T1_D * (sample_rate / pow(10, 6)));
  g->n_samples_PW = (int)(PW_D * (sample_rate / pow(10, 6)));
  g->n_samples_TAG_BIT = (int)(tag_bit_d() * (sample_rate / pow(10, 6)));
  g->win_length = (int)(WIN_SIZE_D * (sample_rate / pow(10, 6)));  /* :52 */
  g->dc_length = (int)(DC_SIZE_D * (sample_rate / pow(10, 6)));    /* :53 */
}

/* ---- a3: gate_impl::general_work, lib/gate_impl.cc:85-200 -------------------------- */
int orc_gate_work(orc_gate *g, orc_reader_state *rs, const orc_cf *in, int n_items, orc_cf *out,
                  int *consumed) {
  int number_samples_consumed = n_items;
  float sample_ampl = 0;
  int written = 0;

  /* :101-109 termination */
  if ((rs->n_queries_sent > rs->cfg.max_num_queries ||
       rs->n_unique_tags > rs->cfg.number_unique_tags) && rs->status != ORC_TERMINATED) {
    rs->status = ORC_TERMINATED;
  }
  /* :112-123 */
  if (rs->gate_status == ORC_GATE_SEEK_EPC) {
    rs->gate_status = ORC_GATE_CLOSED;
    rs->n_samples_to_ungate = (EPC_BITS + TAG_PREAMBLE_BITS) * g->n_samples_TAG_BIT + 2 * g->n_samples_TAG_BIT;
    g->n_samples = 0;
  } else if (rs->gate_status == ORC_GATE_SEEK_RN16) {
    rs->gate_status = ORC_GATE_CLOSED;
    rs->n_samples_to_ungate = (RN16_BITS + TAG_PREAMBLE_BITS) * g->n_samples_TAG_BIT + 2 * g->n_samples_TAG_BIT;
    g->n_samples = 0;
  }

  if (rs->status == ORC_RUNNING) {
    for (int i = 0; i < n_items; i++) {
      /* :130-133 moving average of |x| */
      sample_ampl = cabs_f(in[i]);
      g->avg_ampl = g->avg_ampl + (sample_ampl - g->win_samples[g->win_index]) / g->win_length;
      g->win_samples[g->win_index] = sample_ampl;
      g->win_index = (g->win_index + 1) % g->win_length;
      /* :136 */
      g->sample_thresh = g->avg_ampl * THRESH_FRACTION;

      if (!(rs->gate_status == ORC_GATE_OPEN)) {
        /* :141-143 dc tracking */
        g->dc_est = cadd(g->dc_est, cdiv_real(csub(in[i], g->dc_samples[g->dc_index]), (float)g->dc_length));
        g->dc_samples[g->dc_index] = in[i];
        g->dc_index = (g->dc_index + 1) % g->dc_length;

        g->n_samples++;
        if (sample_ampl < g->sample_thresh && g->signal_state == 1) {        /* :148-152 */
          g->n_samples = 0;
          g->signal_state = 0;
        } else if (sample_ampl > g->sample_thresh && g->signal_state == 0) { /* :154-162 */
          g->signal_state = 1;
          if (g->n_samples > g->n_samples_PW / 2) g->num_pulses++;
          else g->num_pulses = 0;
          g->n_samples = 0;
        }
        if (g->n_samples > g->n_samples_T1 && g->signal_state == 1 &&
            g->num_pulses > NUM_PULSES_COMMAND) {                            /* :164-180 */
          rs->gate_status = ORC_GATE_OPEN;
          rs->n_magn = 0;
          orc_cf s = csub(in[i], g->dc_est);
          if (rs->n_magn < ORC_MAX_MAGN) rs->magn_squared[rs->n_magn++] = cnorm(s);
          out[written] = s;
          written++;
          g->num_pulses = 0;
          g->n_samples = 1;
        }
      } else {                                                               /* :182-195 */
        g->n_samples++;
        orc_cf s = csub(in[i], g->dc_est);
        if (rs->n_magn < ORC_MAX_MAGN) rs->magn_squared[rs->n_magn++] = cnorm(s);
        out[written] = s;
        written++;
        if (g->n_samples >= rs->n_samples_to_ungate) {
          rs->gate_status = ORC_GATE_CLOSED;
          number_samples_consumed = i + 1;
          break;
        }
      }
    }
  }
  *consumed = number_samples_consumed;
  return written;
}

/* ---- a4: tag_decoder_impl ctor, lib/tag_decoder_impl.cc:50-62 ---------------------- */
void orc_decoder_init(orc_decoder *d, int sample_rate) {
  memset(d, 0, sizeof(*d));
  /* float = float * int / double -> double -> float   (:60) */
  d->n_samples_TAG_BIT = (float)(tag_bit_d() * sample_rate / pow(10, 6));
}

/* ---- a5: tag_sync, lib/tag_decoder_impl.cc:78-109 ---------------------------------- */
int orc_tag_sync(orc_decoder *d, const orc_cf *in, int size, float corr_out[15]) {
  (void)size;
  int max_index = 0;
  float max = 0, corr;
  const float nb = d->n_samples_TAG_BIT;
  int n_i = 0;
  for (int i = 0; i < 1.5 * nb; i++) {
    orc_cf corr2 = cf(0, 0);
    for (int j = 0; j < 2 * TAG_PREAMBLE_BITS; j++) {
      orc_cf s = in[(int)(i + j * nb / 2)];
      float c = (float)TAG_PREAMBLE[j];
      /* complex * complex(c,0): (a*c - b*0, a*0 + b*c) */
      orc_cf p = cf(s.re * c - s.im * 0.0f, s.re * 0.0f + s.im * c);
      corr2 = cadd(corr2, p);
    }
    corr = cnorm(corr2);
    if (corr_out && n_i < 15) corr_out[n_i] = corr;
    n_i++;
    if (corr > max) { max = corr; max_index = i; }
  }
  /* :103 */
  orc_cf h = in[max_index];
  h = cadd(h, in[(int)(max_index + nb / 2)]);
  h = cadd(h, in[(int)(max_index + 3 * nb / 2)]);
  h = cadd(h, in[(int)(max_index + 6 * nb / 2)]);
  h = cadd(h, in[(int)(max_index + 10 * nb / 2)]);
  h = cadd(h, in[(int)(max_index + 11 * nb / 2)]);
  d->h_est = cdiv_real(h, 6.0f);
  /* :107 */
  max_index = (int)(max_index + TAG_PREAMBLE_BITS * nb + nb / 2);
  return max_index;
}

/* Re((a-b)*conj(h)) as GCC evaluates complex*complex: ar*cr - ai*ci, c = conj(h) */
static float diff_proj(orc_cf a, orc_cf b, orc_cf h) {
  orc_cf dlt = csub(a, b);
  float cr = h.re, ci = -h.im;
  return dlt.re * cr - dlt.im * ci;
}

/* FM0 differential decision shared by :125-139 and :176-190 */
static float fm0_decide(float result, int *prev) {
  float bit;
  if (result > 0) { bit = (*prev == 1) ? 0.0f : 1.0f; *prev = 1; }
  else            { bit = (*prev == -1) ? 0.0f : 1.0f; *prev = -1; }
  return bit;
}

/* ---- a7: tag_detection_RN16, lib/tag_decoder_impl.cc:114-142 ----------------------- */
int orc_detect_rn16(const orc_decoder *d, const orc_cf *s, int n_s, float *bits) {
  int prev = 1, n = 0;
  for (int j = 0; j < n_s / 2; j++) {
    float result = diff_proj(s[2 * j], s[2 * j + 1], d->h_est);
    bits[n++] = fm0_decide(result, &prev);
  }
  return n;
}

/* ---- a8: tag_detection_EPC, lib/tag_decoder_impl.cc:145-193 ------------------------ */
int orc_detect_epc(orc_decoder *d, const orc_reader_state *rs, const orc_cf *in, int index,
                   float *bits, float energy_out[20]) {
  int prev = 1;
  const int number_steps = 20;
  const float nb = d->n_samples_TAG_BIT;
  /* float <- double expression (:151) */
  float min_val = (float)(nb / 2.0 - nb / 2.0 / 100);
  float max_val = (float)(nb / 2.0 + nb / 2.0 / 100);
  float energy[20];
  for (int t = 0; t < number_steps; t++) {
    energy[t] = 0.0f;
    for (int i = 0; i < 256; i++) {
      energy[t] += rs->magn_squared[(int)(i * (min_val + t * (max_val - min_val) / (number_steps - 1)) + index)];
    }
  }
  int index_T = 0; /* std::max_element: first largest (:165) */
  for (int t = 1; t < number_steps; t++) if (energy[index_T] < energy[t]) index_T = t;
  float T = min_val + index_T * (max_val - min_val) / (number_steps - 1);
  d->T_global = T;
  if (energy_out) memcpy(energy_out, energy, sizeof(energy));
  for (int j = 0; j < 128; j++) {
    orc_cf a = in[(int)(j * (2 * T) + index)];
    orc_cf b = in[(int)(j * 2 * T + T + index)];
    bits[j] = fm0_decide(diff_proj(a, b, d->h_est), &prev);
  }
  return 128;
}

/* ---- a9: check_crc, lib/tag_decoder_impl.cc:401-445 -------------------------------- */
unsigned orc_crc16_bytes(const unsigned char *data, int n) {
  unsigned short crc_16 = 0xFFFF;
  for (int i = 0; i < n; i++) {
    crc_16 ^= (unsigned short)(data[i] << 8);
    for (int j = 0; j < 8; j++) {
      if (crc_16 & 0x8000) { crc_16 <<= 1; crc_16 ^= 0x1021; }
      else crc_16 <<= 1;
    }
  }
  crc_16 = (unsigned short)~crc_16;
  return crc_16;
}

int orc_check_crc(const char *bits, int num_bits) {
  unsigned char data[64];
  int num_bytes = num_bits / 8;
  for (int i = 0; i < num_bytes; i++) {
    int mask = 0x80;
    data[i] = 0;
    for (int j = 0; j < 8; j++) {
      if (bits[i * 8 + j] == '1') data[i] = (unsigned char)(data[i] | mask);
      mask >>= 1;
    }
  }
  unsigned short rcvd = (unsigned short)((data[num_bytes - 2] << 8) + data[num_bytes - 1]);
  unsigned short crc = (unsigned short)orc_crc16_bytes(data, num_bytes - 2);
  return (rcvd != crc) ? -1 : 1;
}

/* slot/round roll-over shared by :269-288, :330-343, :369-383 */
static void next_slot(orc_reader_state *rs, int log_unique) {
  if (rs->cur_slot_number > rs->max_slot_number) {
    rs->cur_slot_number = 1;
    if (log_unique && rs->n_rounds_logged < ORC_MAX_ROUNDS_LOG)
      rs->unique_tags_round[rs->n_rounds_logged++] = rs->n_unique_tags;
    rs->cur_inventory_round += 1;
    rs->gen2_logic_status = ORC_SEND_QUERY;
  } else {
    rs->gen2_logic_status = ORC_SEND_QUERY_REP;
  }
}

/* ---- a10: tag_decoder_impl::general_work, lib/tag_decoder_impl.cc:196-397 ---------- */
int orc_decoder_work(orc_decoder *d, orc_reader_state *rs, const orc_cf *in, int ninput,
                     float *out0, int *consumed, orc_decode_dump *dump) {
  int written = 0;
  *consumed = 0;
  const float nb = d->n_samples_TAG_BIT;
  if (rs->decoder_status == ORC_DECODE_RN16 && ninput >= rs->n_samples_to_ungate) {
    float corr[15];
    int RN16_index = orc_tag_sync(d, in, ninput, corr);
    orc_cf samples[2 * (RN16_BITS - 1)];
    int number_of_half_bits = 0;
    for (float j = RN16_index; j < ninput; j += nb / 2) {   /* :237-253 */
      int k = (int)round(j);
      samples[number_of_half_bits++] = in[k];
      if (number_of_half_bits == 2 * (RN16_BITS - 1)) break;
    }
    if (dump) {
      memset(dump, 0, sizeof(*dump));
      dump->type = ORC_DECODE_RN16; dump->index = RN16_index; dump->h_est = d->h_est;
      memcpy(dump->corr, corr, sizeof(corr));
    }
    if (number_of_half_bits == 2 * (RN16_BITS - 1)) {       /* :256-268 */
      float bits[RN16_BITS - 1];
      int nbits = orc_detect_rn16(d, samples, number_of_half_bits, bits);
      for (int b = 0; b < nbits; b++) out0[written++] = bits[b];
      if (dump) { dump->n_bits = nbits; for (int b = 0; b < nbits; b++) dump->bits[b] = (unsigned char)bits[b]; }
      rs->gen2_logic_status = ORC_SEND_ACK;
    } else {                                                /* :269-288 */
      rs->cur_slot_number++;
      next_slot(rs, 1);
    }
    *consumed = rs->n_samples_to_ungate;
  } else if (rs->decoder_status == ORC_DECODE_EPC && ninput >= rs->n_samples_to_ungate) {
    rs->cur_slot_number++;                                  /* :295 */
    float corr[15], energy[20], bits[128];
    int EPC_index = orc_tag_sync(d, in, ninput, corr);
    int nbits = orc_detect_epc(d, rs, in, EPC_index, bits, energy);
    if (dump) {
      memset(dump, 0, sizeof(*dump));
      dump->type = ORC_DECODE_EPC; dump->index = EPC_index; dump->h_est = d->h_est;
      memcpy(dump->corr, corr, sizeof(corr)); memcpy(dump->energy, energy, sizeof(energy));
      dump->T = d->T_global; dump->n_bits = nbits;
      for (int b = 0; b < nbits; b++) dump->bits[b] = (unsigned char)bits[b];
    }
    if (nbits == EPC_BITS - 1) {
      for (int i = 0; i < 128; i++) d->char_bits[i] = (bits[i] == 0) ? '0' : '1'; /* :320-326 */
      if (orc_check_crc(d->char_bits, 128) == 1) {
        next_slot(rs, 1);                                   /* :330-343 */
        rs->n_epc_correct += 1;
        int result = 0;
        for (int i = 0; i < 8; i++) result += (int)(pow(2, 7 - i) * bits[104 + i]); /* :348-352 */
        if (rs->tag_reads[result] == 0) rs->n_unique_tags++;
        rs->tag_reads[result]++;
        if (dump) { dump->crc_ok = 1; dump->tag_id = result; }
      } else {
        next_slot(rs, 0);                                   /* :369-383 */
      }
    }
    *consumed = rs->n_samples_to_ungate;
  }
  return written;
}

/* ---- reader_impl::general_work, state transitions only: lib/reader_impl.cc:216-372 - */
void orc_reader_work(orc_reader_state *rs, int ninput_items) {
  switch (rs->gen2_logic_status) {
    case ORC_START:                                         /* :218-224 */
      rs->gen2_logic_status = ORC_SEND_QUERY; break;
    case ORC_POWER_DOWN:
      rs->gen2_logic_status = ORC_START; break;
    case ORC_SEND_NAK_QR:
      rs->gen2_logic_status = ORC_SEND_QUERY_REP; break;
    case ORC_SEND_NAK_Q:
      rs->gen2_logic_status = ORC_SEND_QUERY; break;
    case ORC_SEND_QUERY:                                    /* :251-288 */
      rs->n_queries_sent += 1;
      rs->decoder_status = ORC_DECODE_RN16;
      rs->gate_status = ORC_GATE_SEEK_RN16;
      rs->gen2_logic_status = ORC_IDLE; break;
    case ORC_SEND_ACK:                                      /* :290-320 */
      if (ninput_items == RN16_BITS - 1) {
        rs->decoder_status = ORC_DECODE_EPC;
        rs->gate_status = ORC_GATE_SEEK_EPC;
        rs->gen2_logic_status = ORC_SEND_CW;
      }
      break;
    case ORC_SEND_CW:                                       /* :322-327 */
      rs->gen2_logic_status = ORC_IDLE; break;
    case ORC_SEND_QUERY_REP:                                /* :329-344 */
      rs->decoder_status = ORC_DECODE_RN16;
      rs->gate_status = ORC_GATE_SEEK_RN16;
      rs->n_queries_sent += 1;
      rs->gen2_logic_status = ORC_IDLE; break;
    case ORC_SEND_QUERY_ADJUST:                             /* :346-372 */
      rs->decoder_status = ORC_DECODE_RN16;
      rs->gate_status = ORC_GATE_SEEK_RN16;
      rs->n_queries_sent += 1;
      rs->gen2_logic_status = ORC_IDLE; break;
    default: break;
  }
}

/* ---- reader TX waveform -------------------------------------------------------------------------
 * Durations in us (include/rfid/global_vars.h:87-96): CW_D 250, P_DOWN_D 2000, T1_D 240, T2_D 480, PW_D 12,
 * DELIM_D 12, TRCAL_D 200; RN16_D = (17+6)*25 = 575, EPC_D = (129+6)*25 = 3375 (:107-109). */
void orc_reader_tx_init(orc_reader_tx *t, int dac_rate, int fixed_q) {
  const float sample_d = (float)(1.0 / dac_rate * pow(10, 6));            /* reader_impl.cc:51 */
  /* float members (reader_impl.h:35); each use truncates where the reference does */
  const float n_data0_s = 2 * 12 / sample_d, n_data1_s = 4 * 12 / sample_d, n_pw_s = 12 / sample_d;   /* :55-57 */
  const float n_cw_s = 250 / sample_d, n_delim_s = 12 / sample_d, n_trcal_s = 200 / sample_d;          /* :58-60 */
  t->n_data0 = (int)(size_t)n_data0_s;                                     /* data_0.resize(n_data0_s) :84 */
  t->n_data1 = (int)(size_t)n_data1_s;
  t->n_pw = (int)n_pw_s;
  t->n_cw = (int)(size_t)n_cw_s;
  t->n_delim = (int)(size_t)n_delim_s;
  t->n_trcal = (int)(size_t)n_trcal_s;
  t->n_rtcal = (int)(size_t)(n_data0_s + n_data1_s);                       /* rtcal.resize(n_data0_s + n_data1_s) :88 */
  t->n_rtcal_hi = (int)(size_t)((float)(size_t)t->n_rtcal - n_pw_s);       /* fill_n(rtcal.size() - n_pw_s) :95 */
  t->n_trcal_hi = (int)(size_t)((float)(size_t)t->n_trcal - n_pw_s);       /* :96 */
  t->n_cwquery = (int)((240 + 480 + 575) / sample_d);                      /* int members :69-71 */
  t->n_cwack = (int)((3 * 240 + 480 + 3375) / sample_d);
  t->n_pdown = (int)(2000 / sample_d);
  t->fixed_q = fixed_q;
  /* gen_query_bits :131-146: 1000 | DR 0 | M 00 | TRext 0 | Sel 00 | Session 00 | Target 0 | Q(4) | CRC-5 */
  float *q = t->query_bits;
  int n = 0;
  q[n++] = 1; q[n++] = 0; q[n++] = 0; q[n++] = 0;
  for (int i = 0; i < 9; i++) q[n++] = 0;
  for (int i = 3; i >= 0; i--) q[n++] = (float)((fixed_q >> i) & 1);       /* Q_VALUE[FIXED_Q], global_vars.h:79-85 */
  /* crc_append :383-443: 5-bit register preset 01001 (crc[4]..crc[0]), feedback = crc[4] xor bit into
   * positions 0 and 3, appended crc[4] first */
  int crc[5] = {1, 0, 0, 1, 0};
  for (int i = 0; i < 17; i++) {
    const int f = crc[4] ^ (q[i] == 1.0f);
    const int t4 = crc[3], t3 = crc[2] ^ f, t2 = crc[1], t1 = crc[0], t0 = f;
    crc[0] = t0; crc[1] = t1; crc[2] = t2; crc[3] = t3; crc[4] = t4;
  }
  for (int i = 4; i >= 0; i--) q[n++] = (float)crc[i];
}

static int tx_fill(float *out, int w, float v, int n) { for (int i = 0; i < n; i++) out[w + i] = v; return w + n; }
static int tx_data0(const orc_reader_tx *t, float *out, int w) {            /* half high, half low :89 */
  w = tx_fill(out, w, 1.0f, t->n_data0 / 2); return tx_fill(out, w, 0.0f, t->n_data0 - t->n_data0 / 2);
}
static int tx_data1(const orc_reader_tx *t, float *out, int w) {            /* 3/4 high :90 */
  w = tx_fill(out, w, 1.0f, 3 * t->n_data1 / 4); return tx_fill(out, w, 0.0f, t->n_data1 - 3 * t->n_data1 / 4);
}
static int tx_frame_sync(const orc_reader_tx *t, float *out, int w) {       /* delim, data_0, rtcal :101-104 */
  w = tx_fill(out, w, 0.0f, t->n_delim);
  w = tx_data0(t, out, w);
  w = tx_fill(out, w, 1.0f, t->n_rtcal_hi);                                 /* :88,95 */
  return tx_fill(out, w, 0.0f, t->n_rtcal - t->n_rtcal_hi);
}
static int tx_bits(const orc_reader_tx *t, float *out, int w, const float *bits, int n) {
  for (int i = 0; i < n; i++) w = (bits[i] == 1.0f) ? tx_data1(t, out, w) : tx_data0(t, out, w);
  return w;
}

int orc_reader_work_tx(const orc_reader_tx *t, orc_reader_state *rs, const float *in, int ninput_items, float *out) {
  int w = 0;
  switch (rs->gen2_logic_status) {
    case ORC_START: w = tx_fill(out, w, 1.0f, t->n_cwack); break;                       /* :218-224 */
    case ORC_POWER_DOWN: w = tx_fill(out, w, 0.0f, t->n_pdown); break;                  /* :226-231 */
    case ORC_SEND_NAK_QR:
    case ORC_SEND_NAK_Q: {                                                              /* :233-249; nak :114-123 */
      const float nak_bits[8] = {1, 1, 0, 0, 0, 0, 0, 0};
      w = tx_frame_sync(t, out, w);
      w = tx_bits(t, out, w, nak_bits, 8);
      w = tx_fill(out, w, 1.0f, t->n_cw);
      break;
    }
    case ORC_SEND_QUERY:                                                                /* :251-288 */
      w = tx_frame_sync(t, out, w);                                                     /* preamble = frame_sync + trcal :95-99 */
      w = tx_fill(out, w, 1.0f, t->n_trcal_hi);
      w = tx_fill(out, w, 0.0f, t->n_trcal - t->n_trcal_hi);
      w = tx_bits(t, out, w, t->query_bits, 22);
      w = tx_fill(out, w, 1.0f, t->n_cwquery);
      break;
    case ORC_SEND_ACK:                                                                  /* :290-320 */
      if (ninput_items == RN16_BITS - 1) {
        const float ack_code[2] = {0, 1};
        w = tx_frame_sync(t, out, w);
        w = tx_bits(t, out, w, ack_code, 2);
        w = tx_bits(t, out, w, in, 16);
      }
      break;
    case ORC_SEND_CW: w = tx_fill(out, w, 1.0f, t->n_cwack); break;                     /* :322-327 */
    case ORC_SEND_QUERY_REP: {                                                          /* :329-344; query_rep :106-111 */
      const float z4[4] = {0, 0, 0, 0};
      w = tx_frame_sync(t, out, w);
      w = tx_bits(t, out, w, z4, 4);
      w = tx_fill(out, w, 1.0f, t->n_cwquery);
      break;
    }
    case ORC_SEND_QUERY_ADJUST: {                                                       /* :346-372; bits :155-161 */
      const float qa[9] = {1, 0, 0, 1, 0, 0, 0, 0, 0};                                  /* QADJ_CODE, SESSION, Q_UPDN[1] */
      w = tx_frame_sync(t, out, w);
      w = tx_bits(t, out, w, qa, 9);
      w = tx_fill(out, w, 1.0f, t->n_cwquery);
      break;
    }
    default: break;
  }
  orc_reader_work(rs, ninput_items);
  return w;
}

/* ---- reader_impl::print_results, lib/reader_impl.cc:173-192 ------------------------ */
int orc_print_results(const orc_reader_state *rs, char *buf, int cap) {
  int n = 0;
  n += snprintf(buf + n, cap - n, "\n --------------------------\n");
  n += snprintf(buf + n, cap - n, "| Number of queries/queryreps sent : %d\n", rs->n_queries_sent - 1);
  n += snprintf(buf + n, cap - n, "| Current Inventory round : %d\n", rs->cur_inventory_round);
  n += snprintf(buf + n, cap - n, " --------------------------\n");
  n += snprintf(buf + n, cap - n, "| Correctly decoded EPC : %d\n", rs->n_epc_correct);
  n += snprintf(buf + n, cap - n, "| Number of unique tags : %d\n", rs->n_unique_tags);
  for (int id = 0; id < 256; id++) {
    if (rs->tag_reads[id] && n < cap - 64)
      n += snprintf(buf + n, cap - n, "| Tag ID : %x  Num of reads : %d\n", id, rs->tag_reads[id]);
  }
  n += snprintf(buf + n, cap - n, " --------------------------\n");
  return n;
}

/* ---- single-threaded-scheduler harness ---------------------------------------------
 * Order per SURVEY.md section 3.3: the gate breaks at window close, then decoder and reader
 * run to quiescence before the gate sees the next sample. */
static void reader_until_idle(orc_reader_state *rs, int *reader_q) {
  for (int guard = 0; guard < 8; guard++) {
    int before = rs->gen2_logic_status;
    if (before == ORC_IDLE) break;
    int nin = *reader_q;
    orc_reader_work(rs, nin);
    *reader_q = 0; /* consume_each(ninput_items[0]) :378 */
    if (rs->gen2_logic_status == before) break;
  }
}

/* The harness as a resumable object: decimated samples may be fed in pieces of any size (the gate, decoder
 * and reader state, the decoder's input queue and the open-window bookkeeping carry over), so a trace far
 * larger than host memory can be checked chunk by chunk.  open_idx values are global decimated indices. */
struct orc_stream {
  orc_config cfg;
  orc_gate g;
  orc_decoder d;
  orc_reader_state rs;
  int reader_q;
  int chunk;
  orc_cf *gout, *dq;
  int dq_n;
  long n_windows, pos0;       /* windows so far; global index of the next decimated sample */
  long cur_open;
  orc_cf cur_dc;
  /* raw side (orc_stream_feed_raw): the last 28 raw samples and how many were seen */
  orc_cf hist[28];
  long raw_seen;
};

orc_stream *orc_stream_new(const orc_config *cfg, int chunk) {
  orc_stream *s = (orc_stream *)calloc(1, sizeof(orc_stream));
  s->cfg = *cfg;
  orc_gate_init(&s->g, 400000);          /* gate first: it owns reader_state (gate_impl.cc:69) */
  orc_initialize_reader_state(&s->rs, cfg);
  orc_decoder_init(&s->d, 400000);
  s->reader_q = 0;
  reader_until_idle(&s->rs, &s->reader_q);   /* START -> SEND_QUERY -> IDLE */
  s->chunk = (chunk < 1) ? 4096 : chunk;
  s->gout = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)s->chunk);
  s->dq = (orc_cf *)malloc(sizeof(orc_cf) * 8192);
  s->cur_open = -1;
  s->cur_dc = cf(0, 0);
  return s;
}

void orc_stream_free(orc_stream *s) {
  if (!s) return;
  free(s->gout);
  free(s->dq);
  free(s);
}

void orc_stream_state(const orc_stream *s, orc_reader_state *rs) { *rs = s->rs; }
long orc_stream_windows(const orc_stream *s) { return s->n_windows; }

/* feeds n_dec decimated samples; dumps / open_idx / dc_at_open (nullable) receive the windows completed in THIS
 * call (at most max_dumps of them are stored); returns how many were completed */
long orc_stream_feed(orc_stream *s, const orc_cf *y, long n_dec, orc_decode_dump *dumps, long *open_idx,
                     orc_cf *dc_at_open, long max_dumps) {
  orc_reader_state *rs = &s->rs;
  long pos = 0, n_new = 0;
  while (pos < n_dec) {
    int n_items = (int)((n_dec - pos < s->chunk) ? (n_dec - pos) : s->chunk);
    int consumed = 0;
    int written = orc_gate_work(&s->g, rs, y + pos, n_items, s->gout, &consumed);
    /* track open index: the gate emits contiguous samples; the first emitted sample of
     * a window sits (written-1) samples before the last emitted one.  When the window
     * closes in this call the last emitted sample is y[pos+consumed-1]. */
    if (written > 0) {
      if (s->dq_n == 0) {
        long last = (rs->gate_status == ORC_GATE_OPEN) ? (pos + n_items - 1) : (pos + consumed - 1);
        s->cur_open = s->pos0 + last - (written - 1);
        s->cur_dc = s->g.dc_est;
      }
      if (s->dq_n + written <= 8192) { memcpy(s->dq + s->dq_n, s->gout, sizeof(orc_cf) * (size_t)written); s->dq_n += written; }
    }
    pos += consumed;
    /* decoder + reader to quiescence */
    for (;;) {
      float out0[32];
      int dcons = 0;
      orc_decode_dump tmp;
      int produced = orc_decoder_work(&s->d, rs, s->dq, s->dq_n, out0, &dcons, &tmp);
      if (dcons == 0) break;
      if (n_new < max_dumps) {
        if (dumps) dumps[n_new] = tmp;
        if (open_idx) open_idx[n_new] = s->cur_open;
        if (dc_at_open) dc_at_open[n_new] = s->cur_dc;
      }
      n_new++;
      s->n_windows++;
      memmove(s->dq, s->dq + dcons, sizeof(orc_cf) * (size_t)(s->dq_n - dcons));
      s->dq_n -= dcons;
      s->reader_q += produced;
      reader_until_idle(rs, &s->reader_q);
    }
  }
  s->pos0 += n_dec;
  return n_new;
}

/* the same from raw 2 Msps samples: matched filter first (a1), emitting y[n] once its decimation group
 * x[5n..5n+4] is complete, so that any split of a trace yields the floor(N/5) outputs of the one-shot filter */
long orc_stream_feed_raw(orc_stream *s, const orc_cf *x, long n_raw, orc_decode_dump *dumps, long *open_idx,
                         orc_cf *dc_at_open, long max_dumps) {
  const long n_first = s->raw_seen / 5, n_last = (s->raw_seen + n_raw) / 5;
  const long n_out = n_last - n_first;
  orc_cf *y = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)(n_out > 0 ? n_out : 1));
  for (long n = n_first; n < n_last; n++) {
    orc_cf acc = cf(0.0f, 0.0f);
    for (int k = 0; k < 25; k++) {
      const long g = 5 * n - 24 + k;              /* global raw index */
      const long rel = g - s->raw_seen;           /* relative to this call's x[0] */
      orc_cf v;
      if (g < 0) v = cf(0.0f, 0.0f);
      else if (rel >= 0) v = x[rel];
      else v = s->hist[28 + rel];
      acc = cadd(acc, v);
    }
    y[n - n_first] = acc;
  }
  if (n_raw >= 28) memcpy(s->hist, x + n_raw - 28, 28 * sizeof(orc_cf));
  else if (n_raw > 0) {
    memmove(s->hist, s->hist + n_raw, (size_t)(28 - n_raw) * sizeof(orc_cf));
    memcpy(s->hist + 28 - n_raw, x, (size_t)n_raw * sizeof(orc_cf));
  }
  s->raw_seen += n_raw;
  const long r = orc_stream_feed(s, y, n_out, dumps, open_idx, dc_at_open, max_dumps);
  free(y);
  return r;
}

long orc_run_decimated(const orc_config *cfg, const orc_cf *y, long n_dec, int chunk,
                       orc_reader_state *rs, orc_decode_dump *dumps, long *open_idx,
                       orc_cf *dc_at_open, long max_dumps) {
  orc_stream *s = orc_stream_new(cfg, chunk);
  const long n = orc_stream_feed(s, y, n_dec, dumps, open_idx, dc_at_open, max_dumps);
  *rs = s->rs;
  orc_stream_free(s);
  return n;
}

long orc_run_trace(const orc_config *cfg, const orc_cf *raw, long n_raw, int chunk,
                   orc_reader_state *rs, orc_decode_dump *dumps, long *open_idx,
                   orc_cf *dc_at_open, long max_dumps) {
  long n_dec = n_raw / 5;
  orc_cf *y = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)(n_dec > 0 ? n_dec : 1));
  orc_fir_boxcar25_decim5(raw, n_raw, y);
  long r = orc_run_decimated(cfg, y, n_dec, chunk, rs, dumps, open_idx, dc_at_open, max_dumps);
  free(y);
  return r;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* cpu_baseline leg: times the full chain (FIR, then gate+decoder harness) `reps`
 * times on one thread.  secs[0]=FIR, secs[1]=gate+decoder+reader, secs[2]=total. */
long orc_time_trace(const orc_config *cfg, const orc_cf *raw, long n_raw, int reps,
                    double secs[3], orc_reader_state *rs_out) {
  long n_dec = n_raw / 5;
  orc_cf *y = (orc_cf *)malloc(sizeof(orc_cf) * (size_t)(n_dec > 0 ? n_dec : 1));
  secs[0] = secs[1] = secs[2] = 0.0;
  long nw = 0;
  for (int r = 0; r < reps; r++) {
    double t0 = now_s();
    orc_fir_boxcar25_decim5(raw, n_raw, y);
    double t1 = now_s();
    nw = orc_run_decimated(cfg, y, n_dec, 4096, rs_out, NULL, NULL, NULL, 0);
    double t2 = now_s();
    secs[0] += t1 - t0; secs[1] += t2 - t1; secs[2] += t2 - t0;
  }
  free(y);
  return nw;
}

/* cpu_baseline leg, all host cores: `nthreads` independent copies of the single-thread run above
 * (the reference is single-threaded per stream; independent streams are the only parallelism it has).
 * Returns the wall time of the slowest thread in *wall_s; every thread decodes the same trace `reps`
 * times.  n_epc_out (nullable) = EPC decodes of one pass. */
#include <pthread.h>
typedef struct {
  const orc_config *cfg; const orc_cf *raw; long n_raw; int reps; double secs[3]; orc_reader_state rs; long nw;
} orc_mt_job;
static void *orc_mt_main(void *p) {
  orc_mt_job *j = (orc_mt_job *)p;
  j->nw = orc_time_trace(j->cfg, j->raw, j->n_raw, j->reps, j->secs, &j->rs);
  return NULL;
}
long orc_time_trace_mt(const orc_config *cfg, const orc_cf *raw, long n_raw, int reps, int nthreads,
                       double *wall_s, int *n_epc_out) {
  if (nthreads < 1) nthreads = 1;
  orc_mt_job *jobs = (orc_mt_job *)calloc((size_t)nthreads, sizeof(orc_mt_job));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  double t0 = now_s();
  for (int i = 0; i < nthreads; i++) {
    jobs[i].cfg = cfg; jobs[i].raw = raw; jobs[i].n_raw = n_raw; jobs[i].reps = reps;
    pthread_create(&th[i], NULL, orc_mt_main, &jobs[i]);
  }
  for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
  *wall_s = now_s() - t0;
  long nw = jobs[0].nw;
  if (n_epc_out) *n_epc_out = jobs[0].rs.n_epc_correct;
  free(jobs); free(th);
  return nw;
}
