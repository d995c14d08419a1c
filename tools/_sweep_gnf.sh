run() { echo "== $*"; timeout -k 10 200 python tools/diag_gnf_repro.py --Cm 256 "$@" 2>&1 | grep -v "^/opt" | grep -vE "^call [0-9]+ rep|lanes|stats"; }
for r in 65536 0 65536 0; do
run --T 2114 --B 1 --tile-batch 1 --rule $r --reps 200 --calls 1
run --T 1000 --B 3 --tile-batch 3 --rule $r --reps 200 --calls 1
run --T 4000 --B 1 --tile-batch 1 --rule $r --reps 100 --calls 1
done
