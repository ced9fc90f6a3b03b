"""Builds tests/wave_emu/librfid_wave_emu.so (TEST INFRASTRUCTURE: host emulation of the
kernels for the GPU-less CI container; never part of the product)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gen2-uhf-rfid-reader_amd", "csrc")
OUT = os.path.join(HERE, "librfid_wave_emu.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, "emu_driver.cpp"), os.path.join(HERE, "rfid_device_env.h"),
            os.path.join(CSRC, "rfid_kernels.hpp"), os.path.join(CSRC, "rfid_ls2.hpp"), os.path.join(CSRC, "rfid_ls2_enqueue.hpp"),
            os.path.join(CSRC, "rfid_host_math.h"),
            os.path.join(CSRC, "rfid_gen2_host.h"),
            os.path.join(ROOT, "include", "rfid_mi355x.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(s) <= os.path.getmtime(OUT) for s in srcs):
        return OUT
    cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fno-strict-aliasing",
           "-I", HERE,                       # the emulator's rfid_device_env.h shadows the HIP one
           "-I", os.path.join(ROOT, "include"), "-iquote", HERE,
           "-o", OUT, os.path.join(HERE, "emu_driver.cpp"), "-I", CSRC]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
