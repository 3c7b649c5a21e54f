#!/bin/bash
# r06 first lease: the new streaming pins, smoke, the pin report, a baseline bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reference_pin.py tests/test_gpu_stream.py -x -q -m gpu --durations=10 > gpurun_out/r06a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06a_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06a_smoke.log 2>&1
echo "smoke rc=$?" >> gpurun_out/r06a_smoke.log
timeout 600 python tools/reference_pin_report.py > gpurun_out/r06a_reference_pin.txt 2>gpurun_out/r06a_reference_pin.err
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06a_bench20.json 2>gpurun_out/r06a_bench20.err
tail -5 gpurun_out/r06a_pytest.log; tail -3 gpurun_out/r06a_smoke.log; tail -30 gpurun_out/r06a_reference_pin.txt; head -c 1500 gpurun_out/r06a_bench20.json
