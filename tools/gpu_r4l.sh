#!/bin/bash
mkdir -p gpurun_out/r4l
for sl in 1024 512 256; do echo "slice $sl"; SFM_PNP_SPLIT_SLICE=$sl timeout -k 10 200 python tools/time_pnp_stages.py 2>/dev/null | grep -E "^n +(3000|3336|5000|9601)" | tee -a gpurun_out/r4l/time_pnp_slice_$sl.txt; done
