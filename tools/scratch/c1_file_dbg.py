import ctypes as C, sys, os, numpy as np, subprocess, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from webradio_amd import synth
ROOT = os.getcwd()
lib = os.path.join(ROOT, "tests", "cxx", "libwr_host_pipeline.so")
g = np.load("tests/golden/reference_c1.npz"); c1 = synth.C1; n = int(g["block_frames"])
d = tempfile.mkdtemp(); path = os.path.join(d, "capture.bin"); g["u8"].tofile(path)
RUN = r'''
import ctypes as C, sys, numpy as np
lib, path, out = sys.argv[1], sys.argv[2], sys.argv[3]
p = [int(v) for v in sys.argv[4:]]
import torch
L = C.CDLL(lib, mode=C.RTLD_GLOBAL)
fp = C.POINTER(C.c_float)
L.wr_host_run_file.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, fp, C.c_size_t, C.POINTER(C.c_size_t)]
audio = np.zeros(1 << 16, np.float32); n = C.c_size_t()
rc = L.wr_host_run_file(path.encode(), *p, audio.ctypes.data_as(fp), audio.size, C.byref(n))
L.wr_host_stream_blocks.restype = L.wr_host_stream_launches.restype = C.c_ulonglong
np.savez(out, rc=rc, audio=audio[:n.value], sb=int(L.wr_host_stream_blocks()), sl=int(L.wr_host_stream_launches()))
'''
def run(env, fe):
    out = os.path.join(d, "o.npz")
    args = [c1["input_rate"], n, 4, c1["if_hz"], 1, c1["chan_passband"], c1["chan_rate"], c1["audio_passband"], c1["audio_rate"], fe]
    subprocess.run([sys.executable, "-c", RUN, lib, path, out] + [str(a) for a in args], env=dict(os.environ, WEBRADIO_PIN_MIN_BYTES="0", **env), check=True, stderr=subprocess.PIPE if "Q" in env else None)
    r = np.load(out); return r["audio"].copy(), int(r["rc"]), int(r["sl"]), int(r["sb"])
for fe in (1, 0):
    plain, rc, _, _ = run({"WEBRADIO_QUIET": "1"}, fe)
    for extra in ({}, {"WR_TEST_NO_U8_SPECTRUM": "1"}):
        a, rc, sl, sb = run(dict({"WEBRADIO_QUIET": "1", "WEBRADIO_STREAM": "2"}, **extra), fe)
        per = plain.size // 4
        print("frontend", fe, extra, "rc", rc, "launches", sl, "blocks", sb, "per-block max diff", [float(np.abs(a[i*per:(i+1)*per] - plain[i*per:(i+1)*per]).max()) for i in range(4)])
