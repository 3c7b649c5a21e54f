run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --no-secondary --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"; }
run A=1
run A=1
run WR_STREAM_RUN=4
run WR_STREAM_RUN=8
run WR_STREAM_RUN=10
run WR_STREAM_RUN=3
run WR_STREAM_NPOST=500
run WR_STREAM_NPOST=128
run WR_STREAM_NPOST=384
run WR_STREAM_DBG=1
echo "== 20 steps"; python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"
python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"
