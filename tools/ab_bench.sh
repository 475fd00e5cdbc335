#!/bin/bash
# A/B two builds of libsfm_hip.so on the SAME box: gpurun_ab/{old,new}.so are swapped into the package
# and bench.py runs for each, twice, interleaved (clock / thermal drift shows up as A-A differences).
out=gpurun_out/ab; mkdir -p $out
lib=structure-from-motion_amd/libsfm_hip.so
cp $lib /tmp/keep.so
for rep in ${AB_REPS:-1 2}; do
  for v in ${AB_VARIANTS:-old new}; do
    cp gpurun_ab/$v.so $lib
    timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $out/$v.$rep.log 2>&1 || exit 1
    python3 - $out/$v.$rep.log $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/it %.4f" % d["ms_per_step"], {k: round(v*1e3,1) for k,v in d["kernel_ms"].items()})
PY
  done
done
cp /tmp/keep.so $lib
