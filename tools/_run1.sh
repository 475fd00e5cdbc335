bash tools/gpu_guard.sh r3guard2
