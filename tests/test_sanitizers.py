"""CPU hygiene: the host half of the C ABI and the oracle's C sources built under
-fsanitize=address,undefined and driven over truncated / bit-flipped / garbage streams, hostile headers, every writer capacity
around the true size and threaded piece placement (tests/san/san_driver.cpp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_abi_and_oracle_clean_under_asan_ubsan(tmp_path):
    exe = tmp_path / "san_driver"
    objs = []
    for src in ("oracle/icsp_oracle.c", "oracle/icsp_oracle_dec.c"):
        o = tmp_path / (os.path.basename(src) + ".o")
        subprocess.check_call(["gcc", "-c", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-fno-omit-frame-pointer", "-o", str(o), os.path.join(ROOT, src)])
        objs.append(str(o))
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include"), "-o", str(exe), os.path.join(ROOT, "tests", "san", "san_driver.cpp"),
                           os.path.join(ROOT, "icspcodec_amd", "csrc", "icsp_bitstream.cpp"), *objs, "-pthread", "-lm"])
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "sanitizer driver: ok" in out and "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
