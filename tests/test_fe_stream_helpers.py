"""The integer / packed helpers of am_fe_stream.h (round 4: reciprocal-multiply divisions, 24-bit multiplies, packed pairs) against
the plain expressions they replace, over the whole domain they are stated for.  CPU only: the host twins the emulated kernels run."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_fe_stream_helpers(tmp_path):
    exe = str(tmp_path / "fe_stream_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unknown-pragmas",
                           "-Wno-unused-function", "-I", os.path.join(HERE, "emu"), "-I", os.path.join(ROOT, "gr-air-modes_amd", "csrc"),
                           "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(HERE, "helpers", "fe_stream_check.cc"),
                           os.path.join(HERE, "emu", "hipemu.cc")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
