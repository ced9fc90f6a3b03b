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
// glibc-2.35 hypotf: (float)sqrt((double)x*x + (double)y*y), double sqrt correctly rounded
RFID_DEVICE float hypot_f(float x, float y) {
  double s = (double)x * (double)x + (double)y * (double)y;
  return (float)__dsqrt_rn(s);
}
// truncation toward zero of a binary32 value, as (int) in C
RFID_DEVICE int f2i(float v) { return (int)v; }

RFID_DEVICE void block_sync() { __syncthreads(); }
RFID_DEVICE int atomic_add(int *p, int v) { return atomicAdd(p, v); }

}  // namespace wv
