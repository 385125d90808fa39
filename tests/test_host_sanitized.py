"""The host entry points of libqplace.so under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU code involved):
tools/host_sanitize.py builds csrc/qplace_host.cpp with g++ -fsanitize=address,undefined and runs the host ABI tests
against that build in a child interpreter; any sanitizer report aborts the child."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.slow
def test_host_abi_under_address_and_ub_sanitizers():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no sanitizer runtime")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_sanitize.py"), "tests/test_host_abi.py",
                          "-k", "not test_bench_starts_its_own_ranks"],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:]
    assert " passed" in out.stdout and "ERROR: AddressSanitizer" not in out.stdout and "runtime error" not in out.stdout
