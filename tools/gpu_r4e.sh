mkdir -p gpurun_out/r4e
for d in 1024 2048 4096 3072 7168; do SFM_HIP_LIBRARY=$PWD/gpurun_ab/ablate.so timeout -k 10 120 python tools/ablate_linearize.py $d 2>&1 | grep "^debug" | tee -a gpurun_out/r4e/ablate_linearize.txt; done
