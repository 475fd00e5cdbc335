#!/bin/bash
mkdir -p gpurun_out/r4o
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py tests/test_gpu_append.py -m gpu -q -x --timeout 280 > gpurun_out/r4o/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r4o/pytest.log
for p8 in 0 1 auto; do
  if [ $p8 = auto ]; then unset SFM_ROWS_PITCH8; else export SFM_ROWS_PITCH8=$p8; fi
  timeout -k 10 200 python tools/time_schur.py 2>/dev/null | grep "mode 3" | sed "s/^/pitch8=$p8 /" | tee -a gpurun_out/r4o/time_schur_rows_pitch.txt
done
unset SFM_ROWS_PITCH8
python - <<'PY' 2>/dev/null | tee -a gpurun_out/r4o/time_schur_rows_pitch.txt
import importlib, sys, os
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
for name, args in (("100x10000@0.25", (100, 10000, 0.25)), ("80x10000@0.4", (80, 10000, 0.4)), ("120x8000@0.1", (120, 8000, 0.1)), ("40x6000@0.5", (40, 6000, 0.5))):
    sc = sfm.scenes.make_scene(*args, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    for p8 in ("0", "1"):
        os.environ["SFM_ROWS_PITCH8"] = p8
        # the pitch is read once per process: report through a child
        import subprocess
        code = ("import importlib,sys;sys.path.insert(0,'.');sfm=importlib.import_module('structure-from-motion_amd');n=sfm.native;n.init(0);"
                "sc=sfm.scenes.make_scene(%d,%d,%r,seed=0);uvn=sfm.geometry.normalise_pixels(sc.uv_pix,sc.intrinsic);"
                "p=n.BaProblem(sc.n_cams,sc.pt_ptr,sc.cam_idx,uvn);p.set_option(n.OPT_SCHUR,n.SCHUR_ROWS);p.set_option(n.OPT_TIMING,1<<n.K_SCHUR);"
                "p.set_state(sc.cams_init,sc.pts_init);p.iterate(5.0,3);p.reset_timing();p.iterate(5.0,10);ms,k=p.kernel_time(n.K_SCHUR);print(round(1e3*ms/k,1))" % args)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ)).stdout.strip().splitlines()
        print(name, "rows kernel us, pitch8 =", p8, ":", out[-1] if out else "?")
PY
