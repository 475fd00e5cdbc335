"""Proof that SFM_POOL_REDZONE=1 catches an out-of-bounds device write: pokes 8 bytes past the reduced buffer of a
problem and closes it -- the run must END IN AN ABORT with the pool's message (run by hand on the GPU box:
SFM_POOL_REDZONE=1 python tools/redzone_selfcheck.py)."""
import ctypes, importlib, sys, os
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
print("redzone active:", native.pool_redzone_active(), flush=True)
sc = sfm.scenes.make_scene(20, 300, 0.5, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
prob = native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn)
ptr, n, _ = prob.reduced_buffer()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
print("poking 8 bytes past the reduced buffer", flush=True)
hip.hipMemset(ctypes.c_void_p(ptr + 8 * n + 64), 0, 8)
hip.hipDeviceSynchronize()
prob.close()
print("closed without abort (detector did NOT fire)", flush=True)
