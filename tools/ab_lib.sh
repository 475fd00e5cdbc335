#!/bin/bash
# Same-box A/B of two builds of the library: bash tools/ab_lib.sh <old.so> <new.so> [bench args...]
old=$1; new=$2; shift 2
args=${*:---steps 60 --warmup 5 --no-cpu-baseline}
for rep in 1 2 3; do for lib in "$old" "$new"; do
  SFM_HIP_LIBRARY=$PWD/$lib python bench.py $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done; done
