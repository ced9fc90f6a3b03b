# where the fixed costs of a fresh process go: runtime initialisation, context creation, first launch (code object load), staging
import ctypes as C, os, sys, time
sys.path.insert(0, "gen2-uhf-rfid-reader_amd"); sys.path.insert(0, ".")
t0 = time.perf_counter()
lib = C.CDLL(os.path.join("gen2-uhf-rfid-reader_amd", "lib", "librfid_mi355x.so"))
t1 = time.perf_counter()
lib.rfid_host_alloc.restype = C.c_void_p; lib.rfid_host_alloc.argtypes = [C.c_size_t]
p = lib.rfid_host_alloc(1 << 20)
t2 = time.perf_counter()
class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sample_rate", "decim", "n_taps", "fixed_q", "max_num_queries", "number_unique_tags")]
prm = Params(); lib.rfid_params_default(C.byref(prm))
h = C.c_void_p()
rc = lib.rfid_ctx_create(C.byref(prm), 0, C.byref(h))
t3 = time.perf_counter()
h2 = C.c_void_p()
rc2 = lib.rfid_ctx_create(C.byref(prm), 0, C.byref(h2))
t4 = time.perf_counter()
n = C.c_int(0)
lib.rfid_selftest(h, C.byref(n))
t5 = time.perf_counter()
lib.rfid_selftest(h, C.byref(n))
t6 = time.perf_counter()
lib.rfid_stream_begin.argtypes = [C.c_void_p, C.c_int64]
lib.rfid_stream_begin(h, 200000)
t7 = time.perf_counter()
lib.rfid_stream_begin(h2, 200000)
t8 = time.perf_counter()
print("dlopen %.1f ms | first HIP call (rfid_host_alloc 1 MB: runtime + device initialisation) %.1f ms | rfid_ctx_create #1 %.1f ms (rc %d) | #2 %.2f ms | "
      "first kernel launch (rfid_selftest: code object load) %.1f ms | second %.2f ms | rfid_stream_begin(200000) #1 %.1f ms | on the other context %.1f ms"
      % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), rc, 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t6 - t5), 1e3 * (t7 - t6), 1e3 * (t8 - t7)))
