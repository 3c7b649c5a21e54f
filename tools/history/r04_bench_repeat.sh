#!/bin/bash
# the driver's bench invocation N times on one lease: exit code, wall time, value and whether every section is there
N=${1:-8}
mkdir -p gpurun_out
: > gpurun_out/r04_bench_repeat.txt
for i in $(seq 1 $N); do
  t0=$(date +%s)
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b.out 2> /tmp/b.err
  rc=$?
  python - "$i" "$rc" "$(( $(date +%s) - t0 ))" <<'PY' | tee -a gpurun_out/r04_bench_repeat.txt
import json, sys
i, rc, secs = sys.argv[1:4]
try:
    lines = [l for l in open("/tmp/b.out") if l.startswith("{")]
    j = json.loads(lines[-1])
    hf = j["secondary"]["host_fed"]["runs"]
    print("run %s rc=%s %ss lines=%d value=%.0f frac=%.4f one=%.0f c3=%.4f c1=%.0f host_fed=%d/%d ok cpu=%s %.2f (port %.2f)" % (
        i, rc, secs, len(lines), j["value"], j["roofline"]["frac"], j["value_one_block_per_launch"], j["secondary"]["c3"]["roofline"]["frac"],
        j["secondary"]["c1"]["value"], sum("error" not in r for r in hf), len(hf), j["cpu_baseline"]["kind"], j["cpu_baseline"]["value"],
        j["cpu_baseline"].get("port", {}).get("value", 0)))
except Exception as e:
    print("run %s rc=%s %ss BROKEN: %r" % (i, rc, secs, e))
    print(open("/tmp/b.err").read()[-1500:])
PY
done
