cd tests/cxx
for late in 0 1; do for st in 1 0; do
  echo "late=$late stream=$st"; WEBRADIO_QUIET=1 WEBRADIO_AUDIO_LATE=$late WEBRADIO_STREAM=$st timeout 120 ./host_bench 256 100 4000000 dev 2>&1 | tail -2
done; done
WEBRADIO_QUIET=1 WEBRADIO_AUDIO_LATE=1 timeout 120 ./host_bench 256 100 4000000 u8 2>&1 | tail -1
WEBRADIO_QUIET=1 WEBRADIO_STREAM=1 timeout 60 ./host_bench 64 6 400000 dev 2>&1 | tail -1
WEBRADIO_QUIET=1 WEBRADIO_STREAM=0 timeout 60 ./host_bench 64 6 400000 dev 2>&1 | tail -1
