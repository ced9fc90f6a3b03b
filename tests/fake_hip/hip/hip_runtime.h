// tests/fake_hip/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY: a stand-in for the HIP runtime API, for ONE purpose: to compile the
// HOST side of the C-ABI library (gen2-uhf-rfid-reader_amd/csrc/rfid_capi.hip -- contexts, plans, the streaming and look-ahead
// protocols: ~3 000 lines that otherwise only ever run on a GPU box) with g++ in the GPU-less CI container and drive it from
// `pytest -m "not gpu"`.  "Device memory" is host memory, a "launch" runs the UNMODIFIED kernel source on the suite's lock-step wave
// emulator (tests/wave_emu), streams execute in order.  tests/fake_hip/build_capi_emu.py links this into tests/fake_hip/librfid_capi_emu.so,
// which only tests load (tests/test_capi_protocol.py binds it with ctypes themselves); the product library is built by hipcc from the
// same source and has no CPU path -- nothing under gen2-uhf-rfid-reader_amd/ knows this directory exists.
//
// A stream here is a queue of closures.  FAKE_HIP_LAG = n (environment, read per process; default 0): work enqueued on a stream
// becomes runnable only n runtime-API calls later -- the host then meets passes that are still "running" (hipStreamQuery ==
// hipErrorNotReady, flag words not written yet), as it does on a device; a wait (hipStreamSynchronize, hipEventSynchronize, a
// synchronous copy) and fakehip::idle() (what a host spin loop calls) run what is due.  n = 0: every call finishes at once.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <deque>
#include <functional>
#include <map>
#include <vector>

#include <rfid_device_env.h>   // (the emulator's: tests/wave_emu)

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; size_t totalGlobalMem; };
struct hipPointerAttribute_t { hipMemoryType type; void *devicePointer; void *hostPointer; int device; };

namespace fakehip {
struct Event;
struct Op { std::function<void()> fn; uint64_t due; Event *wait; uint64_t wait_gen; };
struct Stream { std::deque<Op> q; };
struct Event { uint64_t gen = 0, done_gen = 0; double t_ms = 0.0; };   // gen: records enqueued; done_gen: records executed
struct State {
  uint64_t tick = 0; int lag = 0; hipError_t last = hipSuccess;
  std::vector<Stream *> streams; std::map<const char *, size_t> pinned;
  State() { const char *v = getenv("FAKE_HIP_LAG"); lag = v ? atoi(v) : 0; }
};
inline State &st() { static State s; return s; }
inline double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; }
// runs what is runnable on `s` (everything when force); an op behind an event record that has not executed stays put
inline bool run_stream(Stream *s, bool force) {
  bool any = false;
  while (!s->q.empty()) {
    Op &o = s->q.front();
    if (!force && o.due > st().tick) break;
    if (o.wait && o.wait->done_gen < o.wait_gen) break;     // (the stream that records it goes first)
    std::function<void()> fn = std::move(o.fn);
    s->q.pop_front();
    fn();
    any = true;
  }
  return any;
}
inline void pump(bool force) { for (bool again = true; again;) { again = false; for (Stream *s : st().streams) again = run_stream(s, force) || again; } }
inline void api_call() { st().tick++; pump(false); }
inline void idle() { api_call(); }    // (a host loop that spins on a word the device writes)
inline void drain_stream(Stream *s) {   // everything enqueued on s so far, and whatever it waits for
  while (!s->q.empty()) {
    if (!run_stream(s, true)) { Op &o = s->q.front(); if (o.wait && o.wait->done_gen < o.wait_gen) pump(true); else break; }
  }
}
inline void enqueue(Stream *s, std::function<void()> fn, Event *wait = nullptr, uint64_t wait_gen = 0) {
  Op o; o.fn = std::move(fn); o.due = st().tick + (uint64_t)st().lag; o.wait = wait; o.wait_gen = wait_gen;
  s->q.push_back(std::move(o));
  if (st().lag == 0) pump(false);
}
inline emu::Idx3 idx3(const dim3 &d) { return emu::Idx3{d.x, d.y, d.z}; }
}  // namespace fakehip

typedef fakehip::Stream *hipStream_t;
typedef fakehip::Event *hipEvent_t;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                                        \
  do {                                                                                                                      \
    const dim3 g__ = (grid), b__ = (block);                                                                                 \
    fakehip::api_call();                                                                                                    \
    fakehip::enqueue((stream), [=]() { emu::launch(fakehip::idx3(g__), fakehip::idx3(b__), [&]() { kernel(__VA_ARGS__); }); }); \
  } while (0)

inline hipError_t hipGetLastError() { hipError_t e = fakehip::st().last; fakehip::st().last = hipSuccess; return e; }
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorNotReady ? "not ready" : "fake hip error"); }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)16 << 30; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)2 << 30; *t = (size_t)16 << 30; return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { fakehip::api_call(); *p = nullptr; if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory; return hipSuccess; }
inline hipError_t hipFree(void *p) { fakehip::pump(true); free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { fakehip::api_call(); *p = nullptr; if (posix_memalign(p, 256, n ? n : 256)) return hipErrorOutOfMemory; fakehip::st().pinned[(const char *)*p] = n; return hipSuccess; }
inline hipError_t hipHostFree(void *p) { fakehip::pump(true); fakehip::st().pinned.erase((const char *)p); free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void *p, size_t n, unsigned) { fakehip::st().pinned[(const char *)p] = n; return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
  for (auto &kv : fakehip::st().pinned)
    if ((const char *)p >= kv.first && (const char *)p < kv.first + kv.second) { a->type = hipMemoryTypeHost; a->devicePointer = (void *)p; a->hostPointer = (void *)p; a->device = 0; return hipSuccess; }
  a->type = hipMemoryTypeUnregistered; a->devicePointer = nullptr; a->hostPointer = (void *)p; a->device = 0;
  return hipErrorInvalidValue;     // (as the runtime answers for pageable memory)
}
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new fakehip::Stream; fakehip::st().streams.push_back(*s); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) {
  fakehip::drain_stream(s);
  auto &v = fakehip::st().streams;
  for (size_t i = 0; i < v.size(); ++i) if (v[i] == s) { v.erase(v.begin() + (long)i); break; }
  delete s; return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t s) { fakehip::api_call(); fakehip::drain_stream(s); return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t s) { fakehip::api_call(); return s->q.empty() ? hipSuccess : hipErrorNotReady; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new fakehip::Event; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { fakehip::pump(true); delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
  fakehip::api_call();
  const uint64_t g = ++e->gen;
  fakehip::enqueue(s, [e, g]() { e->done_gen = g; e->t_ms = fakehip::now_ms(); });
  return hipSuccess;
}
inline hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) {
  fakehip::api_call();
  fakehip::enqueue(s, []() {}, e, e->gen);     // (a no-op that cannot run before the record it names has)
  return hipSuccess;
}
inline hipError_t hipEventSynchronize(hipEvent_t e) { fakehip::api_call(); while (e->done_gen < e->gen) fakehip::pump(true); return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t_ms - a->t_ms); if (*ms < 0.0f) *ms = 0.0f; return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st_) {
  fakehip::api_call();
  fakehip::enqueue(st_, [=]() { memmove(d, s, n); });
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st_) { fakehip::api_call(); fakehip::enqueue(st_, [=]() { memset(d, v, n); }); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { fakehip::api_call(); fakehip::pump(true); memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { fakehip::api_call(); fakehip::pump(true); memset(d, v, n); return hipSuccess; }
