set -o pipefail
L=latent-diffusion-speech_amd/lds
cp $L/liblds.so $L/liblds_new.so
B="python bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-profile"
for r in 1 2; do
  cp $L/liblds_new.so $L/liblds.so && $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('new ', d['ms_per_step'])"
  cp $L/liblds_prev.so $L/liblds.so && $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prev', d['ms_per_step'])"
done
cp $L/liblds_new.so $L/liblds.so
