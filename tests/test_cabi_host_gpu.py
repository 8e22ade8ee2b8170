"""The C ABI driven by a plain C99 host (tests/cabi_host/wiski_cabi_host.c: no Python, no torch in the process): absorb -> posterior mean solve ->
predictive mean by the fused gather and by the stored-rows product (both entries), each checked inside the program against the C oracle compiled into
it.  This is the boundary a C / C++ caller of the reference's hot path would bind."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _binary():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge

    return ge.build_c_host()


def test_header_is_c99_and_the_c_host_builds():
    """include/wiski.h parses as C99 (gcc -fsyntax-only) and the C host links against libwiski_hip.so -- runs without a GPU."""
    subprocess.check_call(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "wiski.h")])
    exe = _binary()
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_plain_c_host_reproduces_the_oracle():
    exe = _binary()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "PARITY OK" in res.stdout
