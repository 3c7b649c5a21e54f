#!/bin/bash
# the drop-in path for 20 000 blocks (13 minutes of a 100 Msps stream) per mode: rate and the process's peak memory
cd $(dirname $0)/../../tests/cxx
python3 - <<'PY'
import json, os, resource, subprocess
for src in ("u8", "f32"):
    for late in ("0", "1"):
        before = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss
        r = subprocess.run(["./host_bench", "256", "20000", "4000000", src], env=dict(os.environ, WEBRADIO_QUIET="1", WEBRADIO_AUDIO_LATE=late),
                           capture_output=True, text=True, timeout=600)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print("%-4s %-40s rc=%d %.3f ms per block over %d blocks, peak rss of the children so far %d MB" % (
            src, d["audio"][:40], r.returncode, d["ms_per_block"], d["blocks"], resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss // 1024))
PY
