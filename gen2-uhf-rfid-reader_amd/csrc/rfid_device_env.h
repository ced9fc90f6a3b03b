// rfid_device_env.h -- gfx950 device environment for rfid_kernels.hpp.
//
// Thin named wrappers over the CDNA4 wave-level operations the kernels use, so the
// kernel source reads in domain terms.  Wavefront = 64 lanes, hard-coded.
//
// (tests/wave_emu/ carries a same-named header that maps these operations onto a
// lock-step 64-lane host emulator so that the kernel logic can be exercised in the
// GPU-less CI container.  That emulator is test infrastructure: it is never built
// into, linked with or loaded by librfid_mi355x.so.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rfid_mi355x.h"

#define RFID_KERNEL(threads) __global__ __launch_bounds__(threads)
// ... with at least `waves` waves per SIMD resident (caps the VGPR budget: 512 / waves)
#define RFID_KERNEL_OCC(threads, waves) __global__ __launch_bounds__(threads, waves)
#define RFID_DEVICE __device__ __forceinline__
#define RFID_SHARED __shared__

namespace wv {

RFID_DEVICE int lane_id() { return (int)(threadIdx.x & 63u); }

// 64-bit vote mask (bit L = predicate of lane L)
RFID_DEVICE uint64_t ballot(bool p) { return __ballot(p); }

RFID_DEVICE float shfl(float v, int src) { return __shfl(v, src, 64); }
RFID_DEVICE int shfl(int v, int src) { return __shfl(v, src, 64); }
RFID_DEVICE float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
RFID_DEVICE int shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
RFID_DEVICE unsigned shfl_xor(unsigned v, int m) { return (unsigned)__shfl_xor((int)v, m, 64); }

// lane L receives lane L-1's value, lane 0 receives 0.0f: DPP wave_shr:1 (fuses into
// v_add_f32_dpp when followed by an add).
RFID_DEVICE float shr1(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}

// inclusive prefix sum of a 32-bit integer over the 64 lanes (lane L gets v_0 + ... + v_L): four row_shr steps
// scan each 16-lane row, row_bcast:15 / row_bcast:31 carry the row totals across (6 DPP adds, no LDS)
RFID_DEVICE int scan_add(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2 and 3
  return v;
}
// the same for binary32 values (a tree of additions: NOT the in-order sum -- for estimates only)
RFID_DEVICE float scan_add_f(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
  return v;
}
// building blocks of a wave scan with any associative operation (lane order kept): the value N lanes down within the 16-lane
// row, the last lane of the previous row (rows 1 and 3), lane 31 (rows 2 and 3), the lane below; lanes without a source get `fill`
template <int N> RFID_DEVICE int dpp_row_shr(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x110 + N, 0xf, 0xf, false); }
RFID_DEVICE int dpp_row_bcast15(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x142, 0xa, 0xf, false); }
RFID_DEVICE int dpp_row_bcast31(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x143, 0xc, 0xf, false); }
RFID_DEVICE int dpp_wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
RFID_DEVICE float rint_f(float v) { return __builtin_rintf(v); }     // v_rndne_f32: to nearest, ties to even
RFID_DEVICE float u2f(uint32_t u) { return __uint_as_float(u); }

// value of lane `k` (k wave-uniform) broadcast to the wave
RFID_DEVICE float readlane(float v, int k) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k));
}
RFID_DEVICE int readlane(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
RFID_DEVICE int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
RFID_DEVICE float uniform(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
RFID_DEVICE uint64_t uniform(uint64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

RFID_DEVICE int popc64(uint64_t m) { return __popcll(m); }
// index of lowest set bit, 64 if none
RFID_DEVICE int ffs64(uint64_t m) { return m ? (__ffsll((long long)m) - 1) : 64; }

// IEEE-754 correctly rounded binary32 quotient
RFID_DEVICE float fdiv(float a, float b) { return __fdiv_rn(a, b); }
RFID_DEVICE float fma_f(float a, float b, float c) { return __builtin_fmaf(a, b, c); }   // one v_fma_f32, single rounding
RFID_DEVICE uint32_t f2u(float x) { return __float_as_uint(x); }
// all LDS reads issued so far have returned (and none issued later starts before): keeps a batch of
// independent ds_reads in flight together instead of the compiler's just-in-time interleaving
RFID_DEVICE void compiler_fence() { asm volatile("" ::: "memory"); }
RFID_DEVICE void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// everything this wave has in flight to or from memory is through (behind a rare store inside a loop of read-ahead loads: the waits
// behind it need not be waits for everything then)
RFID_DEVICE void drain_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// glibc-2.35 hypotf: (float)sqrt((double)x*x + (double)y*y), double sqrt correctly rounded.  Both products are
// exact in binary64, so the fma rounds once exactly like the sum.  The square root is the device library's
// correctly rounded sequence (v_rsq_f64 + two coupled Newton steps + two residual corrections) without its range
// scaling: the sum of two squared binary32 numbers is 0, inf, nan or lies in [2^-298, 2^129), where none is needed.
RFID_DEVICE float hypot_f(float x, float y) {
  const double xd = (double)x, yd = (double)y;
  const double s = __builtin_fma(xd, xd, yd * yd);
  const double y0 = __builtin_amdgcn_rsq(s);
  double g = s * y0, h = y0 * 0.5;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, s);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, s);
  g = __builtin_fma(d, h, g);
  return (float)(__builtin_amdgcn_class(s, 0x260) ? s : g);   // +-0 and +inf are their own square roots
}
// fast transcendental functions for the synthetic-replica generator only (never on the receive path)
RFID_DEVICE float log_fast(float x) { return __logf(x); }
RFID_DEVICE float sin_fast(float x) { return __sinf(x); }
RFID_DEVICE float cos_fast(float x) { return __cosf(x); }
RFID_DEVICE float sqrt_fast(float x) { return __fsqrt_rn(x); }
// truncation toward zero of a binary32 value, as (int) in C
RFID_DEVICE int f2i(float v) { return (int)v; }

RFID_DEVICE void block_sync() { __syncthreads(); }
// LDS hand-off inside ONE wavefront (single-wave workgroups): the LDS queue of a wave is
// in order, so a later ds_read sees an earlier ds_write of any lane; only the compiler must
// not reorder them.  Unlike __syncthreads() this does not drain outstanding global loads
// (s_waitcnt vmcnt(0)), which would serialise the register prefetch of the gate scan.
RFID_DEVICE void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// four consecutive ints from a 16-byte aligned address in one load
RFID_DEVICE void load4_i32(const int *p, int &a, int &b, int &c, int &d) {
  const int4 q = *reinterpret_cast<const int4 *>(p);
  a = q.x; b = q.y; c = q.z; d = q.w;
}
RFID_DEVICE int atomic_add(int *p, int v) { return atomicAdd(p, v); }
RFID_DEVICE int atomic_min(int *p, int v) { return atomicMin(p, v); }
// hand-off between workgroups of one launch (those with lower block index are dispatched first): the publisher's earlier
// global stores are visible to whoever has seen the flag
RFID_DEVICE void publish(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
RFID_DEVICE void await(const int *flag, int v) {
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != v) __builtin_amdgcn_s_sleep(2);
}
RFID_DEVICE int atomic_max(int *p, int v) { return atomicMax(p, v); }
// the workgroups of one launch meet: everything a workgroup has written before is visible to all of them behind it.  `target` =
// arrivals so far expected in the counter (it only counts up: workgroups x meetings).  ALL workgroups concerned must be resident at
// once -- the caller launches no more of them than the device holds whatever else is running (a few hundred single-wave workgroups).
RFID_DEVICE void grid_meet(int *counter, int target, int tid) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  if (tid == 0) {
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// one 8-byte word handed from one workgroup of a launch to another: device-coherent, and NOTHING else is ordered by it (no
// release / acquire: on this part those write back / invalidate the XCD's whole L2 -- once per wave of a 130 000-wave launch
// that made the launch 3x slower).  Whatever belongs together has to sit in the one word.
RFID_DEVICE uint64_t load_u64_agent(const uint64_t *p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
RFID_DEVICE void store_u64_agent(uint64_t *p, uint64_t v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a word in page-locked host memory, behind everything this wave has written (system scope)
RFID_DEVICE void system_release_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }   // this thread's stores, out to host memory
RFID_DEVICE void store_i32_system_release(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
RFID_DEVICE void atomic_or64(uint64_t *p, uint64_t v) { atomicOr(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
RFID_DEVICE void atomic_and64(uint64_t *p, uint64_t v) { atomicAnd(reinterpret_cast<unsigned long long *>(p), (unsigned long long)v); }
// this wave's global stores are visible device-wide when this returns
RFID_DEVICE void global_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
// global load that bypasses the per-CU vector cache (device-coherent): for data another wave of
// the workgroup stored earlier in this launch
RFID_DEVICE float2 load_coherent(const float2 *p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)(u & 0xffffffffu)), __uint_as_float((unsigned)(u >> 32)));
}
RFID_DEVICE int load_coherent_i32(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a 24-byte window record another lane of this wave stored earlier in the launch (three 8-byte device-coherent loads)
RFID_DEVICE rfid_window load_coherent_window(const rfid_window *p) {
  const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
  unsigned long long u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = __hip_atomic_load(q + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  rfid_window w;
  __builtin_memcpy(&w, u, sizeof(w));
  return w;
}
// LDS mailbox words shared by two waves of one workgroup: volatile accesses, in-order LDS
// queue per wave; the store is issued by one lane after the wave's earlier LDS writes.
#define RFID_LDS_AS __attribute__((address_space(3)))
RFID_DEVICE int lds_load(const int *p) {
  // explicit LDS address space: through a generic pointer a volatile access becomes a
  // system-coherent flat load (flat_load_dword sc0 sc1, ~1 us) instead of a ds_read_b32
  const volatile RFID_LDS_AS int *q = (const volatile RFID_LDS_AS int *)p;
  return __builtin_amdgcn_readfirstlane(*q);
}
// the same read without waiting for it: the value stays in a VGPR until wv::uniform() looks at it
RFID_DEVICE int lds_peek(const int *p) {
  const volatile RFID_LDS_AS int *q = (const volatile RFID_LDS_AS int *)p;
  return *q;
}
typedef uint32_t rfid_u32x4 __attribute__((ext_vector_type(4)));
// step descriptor between two waves: 32 bytes at a 16-byte aligned LDS address, written by lane 0 with two 16-byte
// stores (after the wave's earlier LDS writes), read back with two 16-byte loads
RFID_DEVICE void lds_store_desc(int *p, int flags, int nvalid, uint64_t m0, uint64_t m1, int info, int lane) {
  volatile RFID_LDS_AS rfid_u32x4 *q = (volatile RFID_LDS_AS rfid_u32x4 *)p;
  asm volatile("" ::: "memory");
  if (lane == 0) {
    rfid_u32x4 a, b;
    a.x = (uint32_t)flags; a.y = (uint32_t)nvalid; a.z = (uint32_t)m0; a.w = (uint32_t)(m0 >> 32);
    b.x = (uint32_t)m1; b.y = (uint32_t)(m1 >> 32); b.z = (uint32_t)info; b.w = 0u;
    q[0] = a;
    q[1] = b;
  }
  asm volatile("" ::: "memory");
}
RFID_DEVICE void lds_load_desc(const int *p, int &flags, int &nvalid, uint64_t &m0, uint64_t &m1, int &info) {
  const volatile RFID_LDS_AS rfid_u32x4 *q = (const volatile RFID_LDS_AS rfid_u32x4 *)p;
  const rfid_u32x4 a = q[0], b = q[1];
  flags = __builtin_amdgcn_readfirstlane((int)a.x);
  nvalid = __builtin_amdgcn_readfirstlane((int)a.y);
  m0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a.z);
  m1 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)b.y) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)b.x);
  info = __builtin_amdgcn_readfirstlane((int)b.z);
}
// one 16-byte record {w0, 0, mask} at a 16-byte aligned LDS address: written by lane 0 with one store (after the wave's
// earlier LDS writes), read back with one load
RFID_DEVICE void lds_store_rec(int *p, int w0, uint64_t m, int lane) {
  volatile RFID_LDS_AS rfid_u32x4 *q = (volatile RFID_LDS_AS rfid_u32x4 *)p;
  asm volatile("" ::: "memory");
  if (lane == 0) {
    rfid_u32x4 a;
    a.x = (uint32_t)w0; a.y = 0u; a.z = (uint32_t)m; a.w = (uint32_t)(m >> 32);
    q[0] = a;
  }
  asm volatile("" ::: "memory");
}
RFID_DEVICE void lds_load_rec(const int *p, int &w0, uint64_t &m) {
  const volatile RFID_LDS_AS rfid_u32x4 *q = (const volatile RFID_LDS_AS rfid_u32x4 *)p;
  const rfid_u32x4 a = q[0];
  w0 = __builtin_amdgcn_readfirstlane((int)a.x);
  m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)a.w) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)a.z);
}
// LDS reads issued now and used later, outside the compiler's wait bookkeeping (which, across the branches of a loop body,
// falls back to waiting for ALL outstanding LDS operations -- a full round trip right behind the reads): one word at
// `seq`, and lane-wise the two floats at `pair` and 256 bytes behind it.  The three values are valid only behind
// lds_prefetch_wait<N>(), N = the LDS operations this wave issues AT LEAST on every path between the two calls (the
// LDS queue of a wave completes in order).
typedef float rfid_f32x2 __attribute__((ext_vector_type(2)));
RFID_DEVICE void lds_prefetch(const int *seq, const float *pair, int &sq, float &a, float &b) {
  const uint32_t o_seq = (uint32_t)(uintptr_t)(const RFID_LDS_AS int *)seq;
  const uint32_t o_pair = (uint32_t)(uintptr_t)(const RFID_LDS_AS float *)pair;
  rfid_f32x2 v;
  asm volatile("ds_read_b32 %0, %2\n\tds_read2st64_b32 %1, %3 offset1:1" : "=&v"(sq), "=&v"(v) : "v"(o_seq), "v"(o_pair) : "memory");
  a = v.x; b = v.y;
}
// N consecutive float2 at an LDS address, each with a ds_read_b64 of its own, all in flight together, then one wait.
// (Left to the compiler, neighbouring reads are merged into ds_read2_b64 -- which the LDS serves at HALF the rate of two
// ds_read_b64: 8 cycles per instruction against 2 x 2, MI355X_MICROARCH.md "LDS" -- and the matched filter's 25 reads per
// lane are the largest single item on the LDS of a CU that four traces share.)
template <int K, int N, int OFF>
struct LdsSeqReader {   // v[K .. N) <- the float2 at base + 8 (OFF + K ..)
  static RFID_DEVICE void issue(uint32_t base, rfid_f32x2 *v) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=&v"(v[K]) : "v"(base), "n"(8 * (OFF + K)) : "memory");
    LdsSeqReader<K + 1, N, OFF>::issue(base, v);
  }
};
template <int N, int OFF>
struct LdsSeqReader<N, N, OFF> { static RFID_DEVICE void issue(uint32_t, rfid_f32x2 *) {} };
// (((0 + x_0) + x_1) + ...) + x_24 per component over the 25 float2 at p -- the matched filter's in-order sum --
// with the reads in three batches (9 + 8 + 8), two of them in flight at any time: a wait is tied to the first value of
// its batch, the sum's dependence chain orders every later use behind it.
RFID_DEVICE float2 lds_sum25_in_order(const float2 *p) {
  const uint32_t base = (uint32_t)(uintptr_t)(const RFID_LDS_AS float2 *)p;
  rfid_f32x2 a[9], b[8], c[8];
  LdsSeqReader<0, 9, 0>::issue(base, a);
  LdsSeqReader<0, 8, 9>::issue(base, b);
  asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a[0]) : : "memory");
  rfid_f32x2 acc = {0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < 9; ++k) acc = acc + a[k];
  LdsSeqReader<0, 8, 17>::issue(base, c);
  asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(b[0]), "+v"(acc) : : "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = acc + b[k];
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(acc) : : "memory");
#pragma unroll
  for (int k = 0; k < 8; ++k) acc = acc + c[k];
  return make_float2(acc.x, acc.y);
}
template <int N>
RFID_DEVICE void lds_prefetch_wait(int &sq, float &a, float &b) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(sq), "+v"(a), "+v"(b) : "n"(N) : "memory");
}
RFID_DEVICE void lds_store(int *p, int v, int lane) {
  volatile RFID_LDS_AS int *q = (volatile RFID_LDS_AS int *)p;
  asm volatile("" ::: "memory");
  if (lane == 0) *q = v;
  asm volatile("" ::: "memory");
}
// acc += q on both components at once (v_pk_add_f32: two IEEE binary32 additions, each rounded by itself)
RFID_DEVICE void pk_add(float2 &acc, const float2 q) {
  rfid_f32x2 a = {acc.x, acc.y};
  const rfid_f32x2 b = {q.x, q.y};
  a = a + b;
  acc.x = a.x; acc.y = a.y;
}
RFID_DEVICE void set_priority_high() { __builtin_amdgcn_s_setprio(3); }
template <int P> RFID_DEVICE void set_priority() { __builtin_amdgcn_s_setprio(P); }   // 0 (default) .. 3
RFID_DEVICE void backoff() { __builtin_amdgcn_s_sleep(1); }

}  // namespace wv
