#!/bin/bash
# Same box, alternating runs on lds/liblds_prev.so (a copy of the previous build, kept by hand) and the current library: the sampler step
# at B = 16 (bench.py) and, with "b1" as the first argument, one utterance in latency mode as well (tools/host_enqueue_time.py prints its wall time).
set -o pipefail
L=latent-diffusion-speech_amd/lds
cp $L/liblds.so $L/liblds_new.so
B="python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-profile"
one() {
  $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1 B=16 ms_per_step', round(d['ms_per_step'],2))"
  if [ "$2" = "b1" ]; then python tools/host_enqueue_time.py 1 --latency 2>/dev/null | tail -1 | sed "s/^/$1 B=1 latency mode: /"; fi
}
for r in 1 2; do
  cp $L/liblds_new.so $L/liblds.so && one new $1
  cp $L/liblds_prev.so $L/liblds.so && one prev $1
done
cp $L/liblds_new.so $L/liblds.so
