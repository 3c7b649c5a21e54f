cd tests/cxx
for rep in 1 2; do
for src in u8 f32; do
  for D in 0 1; do
    for late in 0 1; do
    echo -n "$src WR_RING_DIRECT=$D late=$late: "
    env WEBRADIO_QUIET=1 WR_RING_DIRECT=$D WEBRADIO_AUDIO_LATE=$late ./host_bench 256 300 4000000 $src 2>&1 | grep -o '"ms_per_block": [0-9.]*, .*abs_sum": [0-9.]*'
    done
  done
done
done
