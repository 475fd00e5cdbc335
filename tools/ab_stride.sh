#!/bin/bash
# Same-box A/B of the bench's own hipEvent sampling: every launch / every 4th / one launch of the timed region.
for rep in 1 2 3; do for st in 1 4 1000; do
python bench.py --steps 40 --warmup 3 --no-cpu-baseline --timing-stride $st 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stride', $st, round(d['value'],1), round(d['ms_per_step']*1e3,2), d['roofline']['launches'], round(d['roofline']['avg_launch_ms']*1e3,2))"
done; done
