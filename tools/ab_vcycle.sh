#!/bin/bash
# same-box A/B of two library builds (tree vs MMGP_LIBDIR=$1) on both headline V-cycles; development aid
old=${1:?directory of the other build}
for rep in 1 2; do
  for lib in tree "$old"; do
    for cfg in "--nside 1000 --levels 5 --polydeg 4 --cycles 20" "--dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10"; do
      if [ "$lib" = tree ]; then out=$(python bench_vcycle.py $cfg 2>/dev/null | tail -1); else out=$(MMGP_LIBDIR=$lib python bench_vcycle.py $cfg 2>/dev/null | tail -1); fi
      echo "$lib rep $rep: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["workload"][:12], round(d["device_ms_per_vcycle"],3), "ms")')"
    done
  done
done
