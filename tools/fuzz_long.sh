#!/bin/bash
# The seeded fuzz tests with many more seeds than the gate runs (tests/test_gpu_fuzz.py, test_gpu_f4.py: random tuner
# configurations, spectrum streams, block kernels, ring streams, blocks per launch, post-stage runs -- every one against the
# oracle or the sequential pass).  Leaves gpurun_out/<round>_fuzz_long.txt: the command, pytest's count per test function, its
# summary line.   gpurun --timeout 2400 -- 'bash tools/fuzz_long.sh 300'
seeds=${1:-300}
mkdir -p gpurun_out
out=gpurun_out/${2:-r05}_fuzz_long.txt
{
echo "WR_FUZZ_SEEDS=$seeds python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_f4.py tests/test_gpu_stage.py tests/test_gpu_spectrum.py tests/test_gpu_stream.py -q -p no:cacheprovider   ($(date -u +%FT%TZ), head $(cat .gate_head 2>/dev/null))"
WR_FUZZ_SEEDS=$seeds timeout 2200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_f4.py tests/test_gpu_stage.py tests/test_gpu_spectrum.py tests/test_gpu_stream.py -q -p no:cacheprovider -rA 2>&1 | grep -E "^(PASSED|FAILED|ERROR)|passed|failed" | sed -E 's/\[.*//' | sort | uniq -c
} > $out 2>&1
tail -5 $out
