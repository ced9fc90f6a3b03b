"""The C/C++ library semantics the reference's compiled code leans on, checked on THIS
toolchain (g++ / libstdc++ / glibc): the oracle restates them as explicit formulas.
  std::abs(complex<float>)            == (float)sqrt((double)re*re + (double)im*im)   (glibc hypotf)
  complex<float> / complex<float>(c,0) == elementwise correctly rounded quotient       (libgcc __divsc3)
  std::norm(complex<float>)           == re*re + im*im                                 (libstdc++ 11)
  Re((a+bi) * conj(h))                == a*hr - b*(-hi)
(reference call sites: gate_impl.cc:130,141,175; tag_decoder_impl.cc:94,103,123)"""
import os
import subprocess
import textwrap


def test_libstdcxx_glibc_semantics(tmp_path):
    src = tmp_path / "sem.cc"
    src.write_text(textwrap.dedent(r"""
        #include <complex>
        #include <cstdio>
        #include <cmath>
        #include <cstring>
        #include <random>
        int main() {
          std::mt19937_64 g(1);
          std::uniform_real_distribution<float> u(-30.f, 30.f);
          long bad = 0;
          for (long i = 0; i < 3000000; i++) {
            float x = u(g), y = u(g);
            if (i % 3 == 0) { x *= 1e-3f; y *= 1e-4f; }
            if (i % 7 == 0) { x *= 1e-20f; }
            std::complex<float> z(x, y);
            float a = std::abs(z);
            float b = (float)std::sqrt((double)x * (double)x + (double)y * (double)y);
            if (std::memcmp(&a, &b, 4)) bad++;
            std::complex<float> q = z / std::complex<float>(48, 0);
            if (q.real() != x / 48.0f || q.imag() != y / 48.0f) bad++;
            std::complex<float> q6 = z / std::complex<float>(6, 0);
            if (q6.real() != x / 6.0f || q6.imag() != y / 6.0f) bad++;
            if (std::norm(z) != x * x + y * y) bad++;
            std::complex<float> h(y, x * 0.5f);
            float r = std::real(z * std::conj(h));
            float r2 = x * y - y * (-(x * 0.5f));
            if (r != r2) bad++;
            if ((float)hypotf(x, y) != b) bad++;
          }
          std::printf("%ld\n", bad);
          return bad != 0;
        }
    """))
    exe = tmp_path / "sem"
    subprocess.check_call(["g++", "-O3", "-DNDEBUG", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().strip()
    assert out == "0"


def test_constant_division_sequence_is_correctly_rounded(tmp_path):
    """div_const_fast (rfid_kernels.hpp): q1 = x*RN(1/C), q = fma(fma(-q1, C, x), RN(1/C), q1) equals the
    correctly rounded x / C for C = 100 and 48 whenever |x| >= 2^-100 (the range div_const_ok admits).
    Sampled here (every 64th magnitude + every magnitude around the admission boundary and near
    FLT_MAX); tests/tools/divcheck.c without an argument runs all 2^31 magnitudes (about a minute:
    mismatches exist only below 2^-122)."""
    exe = tmp_path / "divcheck"
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "divcheck.c")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), src, "-lm"])
    out = subprocess.check_output([str(exe), "64"]).decode().strip().splitlines()
    assert len(out) == 2
    for line in out:
        kv = dict(f.split("=") for f in line.split()[0:] if "=" in f)
        assert int(kv["largest_bad_bits"], 16) < 0x0d800000, line      # nothing admitted by div_const_ok fails
        assert int(kv["smallest_large_bad_bits"], 16) == 0x7f800000, line
