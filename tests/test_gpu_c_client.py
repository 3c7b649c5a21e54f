"""-m gpu: the C ABI from a plain C client (examples/fm_receivers.c): compiled with gcc against
include/webradio_amd.h, linked with nothing but the shared library, run as its own process."""
import os
import pytest

import _proc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c99_client_hears_its_tones(tmp_path):
    exe = str(tmp_path / "fm_receivers")
    lib = os.path.join(ROOT, "webradio_amd", "lib")
    _proc.run(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "fm_receivers.c"), "-L" + lib, "-lwebradio_amd", "-lm",
                           "-Wl,-rpath," + lib, "-o", exe])
    out = _proc.output([exe], timeout=120).decode()
    assert out.strip().endswith("ok"), out
    assert out.count("hears") == 8 and "not its own" not in out
