mkdir -p gpurun_out/r3j
python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread 2>&1 | tail -3 | tee gpurun_out/r3j/pytest.txt
python tools/time_small.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3j/time_small.txt | grep "6x1260\|8x2000"
