#!/bin/bash
out=gpurun_out/flow2; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 200 --timeout-method=thread -k "data_flow" > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/steps.log; tail -3 $out/pytest.log
