"""Builds tests/fake_hip/librfid_capi_emu.so (TEST INFRASTRUCTURE): the HOST side of the C-ABI library -- csrc/rfid_capi.hip, unmodified --
compiled with g++ against tests/fake_hip/hip/hip_runtime.h (a stand-in runtime whose launches run the unmodified kernel source on the
wave emulator of tests/wave_emu).  Only tests load it; the product library is hipcc's build of the same source and has no CPU path."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gen2-uhf-rfid-reader_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "wave_emu")
OUT = os.path.join(HERE, "librfid_capi_emu.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(EMU, "emu_driver.cpp"), os.path.join(EMU, "rfid_device_env.h"),
            os.path.join(ROOT, "include", "rfid_mi355x.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in srcs):
        return OUT
    cmd = ["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fno-strict-aliasing", "-pthread",
           "-DLS2_FIN_WPB=16",
           "-I", HERE,                       # <hip/hip_runtime.h> = the stand-in
           "-I", EMU,                        # the emulator's rfid_device_env.h shadows the HIP one
           "-I", os.path.join(ROOT, "include"), "-iquote", EMU, "-I", CSRC,
           "-o", OUT, "-x", "c++", os.path.join(CSRC, "rfid_capi.hip"), os.path.join(EMU, "emu_driver.cpp")]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
