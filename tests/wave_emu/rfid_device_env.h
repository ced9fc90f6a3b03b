// tests/wave_emu/rfid_device_env.h -- HOST EMULATION of the gfx950 device environment.
//
// TEST INFRASTRUCTURE ONLY.  This header shadows gen2-uhf-rfid-reader_amd/csrc/
// rfid_device_env.h when tests/wave_emu/emu_driver.cpp is compiled with g++, so that the
// unmodified kernel source (rfid_kernels.hpp) can be executed in the GPU-less CI container:
// every workgroup runs as a set of cooperative fibers (ucontext) that advance in lock step
// at each wave-level operation.  It is never compiled into, linked with or loaded by
// librfid_mi355x.so and is not a fallback of any kind: the product has no CPU path.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "rfid_mi355x.h"
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <vector>

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

namespace emu {

struct Idx3 { unsigned x, y, z; };

struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  Idx3 tid;
  bool done = false;
  uint64_t wave_calls = 0;   // number of wave collectives this fiber has entered
  uint64_t block_calls = 0;  // number of block barriers this fiber has entered
};

struct Block {
  std::vector<Fiber> fibers;
  int nthreads = 0;
  // per wave: exchange buffers (double buffered), arrival counts, released generation
  std::vector<uint64_t> xbuf;      // [nwaves][2][64]
  std::vector<int> wave_arrived;   // [nwaves]
  std::vector<uint64_t> wave_gen;  // [nwaves]
  int block_arrived = 0;
  uint64_t block_gen = 0;
};

extern Block *g_blk;
extern Fiber *g_cur;
extern Idx3 g_block_idx, g_grid_dim, g_block_dim;
extern ucontext_t g_sched;

void yield();
void launch(Idx3 grid, Idx3 block, const std::function<void()> &body);

static inline int lanes_in_wave(int wave) {
  int rem = g_blk->nthreads - wave * 64;
  return rem > 64 ? 64 : rem;
}

// all live lanes of the calling fiber's wave meet here
static inline void wave_barrier() {
  const int wave = (int)(g_cur->tid.x / 64);
  const uint64_t my = g_cur->wave_calls++;
  if (++g_blk->wave_arrived[wave] == lanes_in_wave(wave)) {
    g_blk->wave_arrived[wave] = 0;
    g_blk->wave_gen[wave] = my + 1;
  } else {
    while (g_blk->wave_gen[wave] <= my) yield();
  }
}

// deposit a 64-bit value, meet, and get the wave's 64 deposited values
static inline const uint64_t *exchange(uint64_t v) {
  const int wave = (int)(g_cur->tid.x / 64), lane = (int)(g_cur->tid.x % 64);
  uint64_t *buf = &g_blk->xbuf[((size_t)wave * 2 + (g_cur->wave_calls & 1)) * 64];
  if (lane == 0) for (int i = lanes_in_wave(wave); i < 64; ++i) buf[i] = 0;
  buf[lane] = v;
  wave_barrier();
  return buf;
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

}  // namespace emu

#define threadIdx (emu::g_cur->tid)
#define blockIdx (emu::g_block_idx)
#define gridDim (emu::g_grid_dim)
#define blockDim (emu::g_block_dim)

#define RFID_KERNEL(threads) static inline
#define RFID_KERNEL_OCC(threads, waves) static inline
#define RFID_DEVICE static inline
#define RFID_SHARED static
#define __device__

namespace wv {

static inline int lane_id() { return (int)(threadIdx.x & 63u); }

static inline uint64_t ballot(bool p) {
  const uint64_t *b = emu::exchange(p ? 1 : 0);
  uint64_t m = 0;
  for (int i = 0; i < 64; ++i) if (b[i]) m |= 1ull << i;
  return m;
}
static inline float shfl(float v, int src) { return emu::u2f((uint32_t)emu::exchange(emu::f2u(v))[src & 63]); }
static inline int shfl(int v, int src) { return (int)(uint32_t)emu::exchange((uint32_t)v)[src & 63]; }
static inline float shfl_xor(float v, int m) { return emu::u2f((uint32_t)emu::exchange(emu::f2u(v))[(lane_id() ^ m) & 63]); }
static inline int shfl_xor(int v, int m) { return (int)(uint32_t)emu::exchange((uint32_t)v)[(lane_id() ^ m) & 63]; }
static inline unsigned shfl_xor(unsigned v, int m) { return (unsigned)emu::exchange(v)[(lane_id() ^ m) & 63]; }
static inline float shr1(float v) {
  const uint64_t *b = emu::exchange(emu::f2u(v));
  const int l = lane_id();
  return (l == 0) ? 0.0f : emu::u2f((uint32_t)b[l - 1]);
}
static inline int scan_add(int v) {
  const uint64_t *b = emu::exchange((uint32_t)v);
  int acc = 0;
  for (int i = 0; i <= lane_id(); ++i) acc += (int)(uint32_t)b[i];
  return acc;
}
static inline float scan_add_f(float v) {   // (an estimate on the device too: any order of the additions will do)
  const uint64_t *b = emu::exchange(emu::f2u(v));
  float acc = 0.0f;
  for (int i = 0; i <= lane_id(); ++i) acc += emu::u2f((uint32_t)b[i]);
  return acc;
}
template <int N> static inline int dpp_row_shr(int v, int fill) {
  const uint64_t *b = emu::exchange((uint32_t)v);
  const int l = lane_id();
  return ((l & 15) >= N) ? (int)(uint32_t)b[l - N] : fill;
}
static inline int dpp_row_bcast15(int v, int fill) {
  const uint64_t *b = emu::exchange((uint32_t)v);
  const int l = lane_id(), row = l >> 4;
  return (row == 1 || row == 3) ? (int)(uint32_t)b[16 * row - 1] : fill;
}
static inline int dpp_row_bcast31(int v, int fill) {
  const uint64_t *b = emu::exchange((uint32_t)v);
  return (lane_id() >= 32) ? (int)(uint32_t)b[31] : fill;
}
static inline int dpp_wave_shr1(int v, int fill) {
  const uint64_t *b = emu::exchange((uint32_t)v);
  const int l = lane_id();
  return (l > 0) ? (int)(uint32_t)b[l - 1] : fill;
}
static inline float rint_f(float v) { return nearbyintf(v); }   // default rounding mode: to nearest, ties to even
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float readlane(float v, int k) { return emu::u2f((uint32_t)emu::exchange(emu::f2u(v))[k & 63]); }
static inline int readlane(int v, int k) { return (int)(uint32_t)emu::exchange((uint32_t)v)[k & 63]; }
// readfirstlane: lane 0's value (all lanes are live at every call site in the kernels)
static inline int uniform(int v) { return (int)(uint32_t)emu::exchange((uint32_t)v)[0]; }
static inline float uniform(float v) { return emu::u2f((uint32_t)emu::exchange(emu::f2u(v))[0]); }
static inline uint64_t uniform(uint64_t v) { return emu::exchange(v)[0]; }

static inline int popc64(uint64_t m) { return __builtin_popcountll(m); }
static inline int ffs64(uint64_t m) { return m ? __builtin_ctzll(m) : 64; }
static inline float fdiv(float a, float b) { volatile float q = a / b; return q; }
static inline float log_fast(float x) { return logf(x); }
static inline float sin_fast(float x) { return sinf(x); }
static inline float cos_fast(float x) { return cosf(x); }
static inline float sqrt_fast(float x) { return sqrtf(x); }
static inline float fma_f(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline void compiler_fence() {}
static inline void lds_wait() {}
static inline void drain_vm() {}
static inline float hypot_f(float x, float y) {
  return (float)sqrt((double)x * (double)x + (double)y * (double)y);
}
static inline int f2i(float v) { return (int)v; }

static inline void block_sync() {
  emu::Block *B = emu::g_blk;
  const uint64_t my = emu::g_cur->block_calls++;
  int live = 0;
  for (auto &f : B->fibers) if (!f.done) live++;
  if (++B->block_arrived >= live) {
    B->block_arrived = 0;
    B->block_gen = my + 1;
  } else {
    while (B->block_gen <= my) emu::yield();
  }
}
static inline void wave_sync() { emu::wave_barrier(); }
// the device versions end in v_readfirstlane: every lane gets LANE 0's read (lanes are fibers here and would
// otherwise sample a mailbox word at different times and diverge)
static inline int lds_load(const int *p) { return (int)(uint32_t)emu::exchange((uint32_t)*(const volatile int *)p)[0]; }
static inline int lds_peek(const int *p) { return *(const volatile int *)p; }
static inline void lds_store(int *p, int v, int lane) { emu::wave_barrier(); if (lane == 0) *(volatile int *)p = v; }
static inline uint64_t lds_load64(const uint64_t *p) { return emu::exchange(*(const volatile uint64_t *)p)[0]; }
static inline void lds_store_rec(int *p, int w0, uint64_t m, int lane) {
  emu::wave_barrier();
  if (lane == 0) {
    volatile uint32_t *q = (volatile uint32_t *)p;
    q[0] = (uint32_t)w0; q[1] = 0u; q[2] = (uint32_t)m; q[3] = (uint32_t)(m >> 32);
  }
}
static inline void lds_load_rec(const int *p, int &w0, uint64_t &m) {
  w0 = lds_load(p); m = lds_load64((const uint64_t *)(p + 2));
}
static inline void lds_prefetch(const int *seq, const float *pair, int &sq, float &a, float &b) {
  sq = *(const volatile int *)seq; a = *(const volatile float *)pair; b = *(const volatile float *)(pair + 64);
}
template <int N> static inline void lds_prefetch_wait(int &, float &, float &) {}
static inline float2 lds_sum25_in_order(const float2 *p) {
  float re = 0.0f, im = 0.0f;
  for (int k = 0; k < 25; ++k) { re = re + p[k].x; im = im + p[k].y; }
  return make_float2(re, im);
}
static inline void lds_store_desc(int *p, int flags, int nvalid, uint64_t m0, uint64_t m1, int info, int lane) {
  emu::wave_barrier();
  if (lane == 0) {
    volatile uint32_t *q = (volatile uint32_t *)p;
    q[0] = (uint32_t)flags; q[1] = (uint32_t)nvalid; q[2] = (uint32_t)m0; q[3] = (uint32_t)(m0 >> 32);
    q[4] = (uint32_t)m1; q[5] = (uint32_t)(m1 >> 32); q[6] = (uint32_t)info; q[7] = 0u;
  }
}
static inline void lds_load_desc(const int *p, int &flags, int &nvalid, uint64_t &m0, uint64_t &m1, int &info) {
  flags = lds_load(p); nvalid = lds_load(p + 1);
  m0 = lds_load64((const uint64_t *)(p + 2)); m1 = lds_load64((const uint64_t *)(p + 4));
  info = lds_load(p + 6);
}
static inline void pk_add(float2 &acc, const float2 q) {
  volatile float x = acc.x + q.x, y = acc.y + q.y;   // (two binary32 additions, each rounded by itself)
  acc.x = x; acc.y = y;
}
static inline void set_priority_high() {}
template <int P> static inline void set_priority() {}
static inline void backoff() { emu::yield(); }
static inline void load4_i32(const int *p, int &a, int &b, int &c, int &d) { a = p[0]; b = p[1]; c = p[2]; d = p[3]; }
static inline int atomic_add(int *p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomic_min(int *p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomic_max(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
// (the emulator runs one workgroup at a time: a launch whose workgroups meet has exactly one here)
static inline void grid_meet(int *, int, int) { fprintf(stderr, "[emu] grid_meet with more than one workgroup\n"); abort(); }
static inline uint64_t load_u64_agent(const uint64_t *p) { return *(const volatile uint64_t *)p; }
static inline void store_u64_agent(uint64_t *p, uint64_t v) { *(volatile uint64_t *)p = v; }
static inline void system_release_fence() {}
static inline void store_i32_system_release(int *p, int v) { *(volatile int *)p = v; }
static inline void publish(int *flag, int v) { *(volatile int *)flag = v; }
// (the emulator runs the workgroups of a launch one after the other in index order: a flag of a lower block is set by now)
static inline void await(const int *flag, int v) { if (*(const volatile int *)flag != v) { fprintf(stderr, "[emu] await on a flag that was never published\n"); abort(); } }
static inline void atomic_or64(uint64_t *p, uint64_t v) { *p |= v; }
static inline void atomic_and64(uint64_t *p, uint64_t v) { *p &= v; }
static inline void global_release() {}
static inline float2 load_coherent(const float2 *p) { return *p; }
static inline int load_coherent_i32(const int *p) { return *p; }
static inline rfid_window load_coherent_window(const rfid_window *p) { return *p; }

}  // namespace wv
