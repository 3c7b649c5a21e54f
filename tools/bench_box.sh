#!/bin/bash
# One fresh box's answer to the driver's invocation, kept as a file of its own (gpurun merges files, it does not append):
#   gpurun --timeout 600 -- 'bash tools/bench_box.sh'   ->   gpurun_out/${ROUND:-r06}_box_<UTC time>.json
# tools/bench_boxes_table.py turns the collected files into profiles/r05_bench_boxes.txt.
mkdir -p gpurun_out
stamp=$(date -u +%Y%m%dT%H%M%SZ)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${ROUND:-r06}_box_$stamp.json 2> gpurun_out/${ROUND:-r06}_box_$stamp.err
echo "rc=$? $(cut -c1-200 gpurun_out/${ROUND:-r06}_box_$stamp.json)"
