cd tests/cxx
for st in 1 0; do
  echo "u8 late stream=$st"; WEBRADIO_QUIET=1 WEBRADIO_AUDIO_LATE=1 WEBRADIO_STREAM=$st timeout 120 ./host_bench 256 100 4000000 u8 2>&1 | tail -2
done
echo "u8 on time stream=2"; WEBRADIO_QUIET=1 WEBRADIO_STREAM=2 timeout 120 ./host_bench 256 100 4000000 u8 2>&1 | tail -2
echo "u8 on time stream=1"; WEBRADIO_QUIET=1 WEBRADIO_STREAM=1 timeout 120 ./host_bench 256 100 4000000 u8 2>&1 | tail -2
cd ../..
for e in 0 1; do echo "WR_STREAM_EXT=$e"; for i in 1 2 3; do WR_STREAM_EXT=$e python bench.py --steps 200 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; done; done
timeout 800 python -m pytest tests/test_gpu_host.py -x -q -m gpu 2>&1 | tail -6
