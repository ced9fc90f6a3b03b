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
RFID_DEVICE int atomic_add(int *p, int v) { return atomicAdd(p, v); }
RFID_DEVICE int atomic_min(int *p, int v) { return atomicMin(p, v); }
// hand-off between workgroups of one launch (those with lower block index are dispatched first): the publisher's earlier
// global stores are visible to whoever has seen the flag
RFID_DEVICE void publish(int *flag, int v) { __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
RFID_DEVICE void await(const int *flag, int v) {
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != v) __builtin_amdgcn_s_sleep(2);
}
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
RFID_DEVICE void lds_store(int *p, int v, int lane) {
  volatile RFID_LDS_AS int *q = (volatile RFID_LDS_AS int *)p;
  asm volatile("" ::: "memory");
  if (lane == 0) *q = v;
  asm volatile("" ::: "memory");
}
RFID_DEVICE void set_priority_high() { __builtin_amdgcn_s_setprio(3); }
RFID_DEVICE void backoff() { __builtin_amdgcn_s_sleep(1); }

}  // namespace wv
