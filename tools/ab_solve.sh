#!/bin/bash
# Same-box A/B of the reduced solve: identity rows + dp = X y (default) against the block-row back substitution
# (bench.py --debug 512), C3 and the C4 share, two repeats.
for rep in 1 2; do
for dbg in 0 512; do
  python bench.py --steps 40 --warmup 3 --no-cpu-baseline --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 debug', $dbg, round(d['value'],1), round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
  python bench.py --config C4 --pts 12500 --steps 20 --warmup 2 --no-cpu-baseline --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4s debug', $dbg, round(d['value'],1), round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})"
done; done
