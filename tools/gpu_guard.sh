#!/bin/bash
# ONE pass of the GPU suite with SFM_POOL_GUARD=1: every device buffer of the library ends at the end of its own
# virtual-memory mapping with an unmapped granule behind it, so an out-of-bounds READ faults at the access.
#   bash tools/gpu_guard.sh <tag> [pytest targets]
tag=${1:-guard}
shift
targets=${*:-tests}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SFM_TRACE_EXAMPLES="$out/examples.log"
echo "== guard-mode probe" | tee -a "$out/steps.log"
SFM_POOL_GUARD=1 timeout -k 10 120 python - > "$out/probe.log" 2>&1 <<'PY'
import importlib, sys
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); n = sfm.native; n.init(0)
print("pool mode, tail slack, guard allocations:", n.pool_mode(1000), n.pool_mode(4096), n.pool_mode(1 << 20))
sc = sfm.scenes.make_scene(6, 300, 0.6, seed=5); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
import time; t0 = time.perf_counter()
with n.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
    prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); prob.get_state()
print("one small problem under guard mode: %.3f s" % (time.perf_counter() - t0), n.pool_mode(1000))
PY
rc=$?; echo "   rc=$rc" | tee -a "$out/steps.log"; cat "$out/probe.log" | grep -v amdgpu.ids
if [ $rc -ne 0 ]; then echo "guard mode unavailable; stopping" | tee -a "$out/steps.log"; exit 0; fi
echo "== pytest -m gpu under SFM_POOL_GUARD=1" | tee -a "$out/steps.log"
SFM_POOL_GUARD=1 timeout -k 10 1000 python -m pytest $targets -m gpu --maxfail=3 --timeout 300 --timeout-method=thread --capture=sys -v > "$out/pytest_gpu_guard.log" 2>&1
rc=$?; echo "   rc=$rc" | tee -a "$out/steps.log"
tail -5 "$out/pytest_gpu_guard.log"
exit 0
