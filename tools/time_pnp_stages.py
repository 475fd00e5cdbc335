"""Where the per-view PnP of the incremental loop spends its time (host-buffer entry points, as the drop-in calls them):
the six-point RANSAC (300 hypotheses) and the nonlinear refinement (lambda = 5, 300 iterations) by view size."""
import importlib, sys, time, random
import numpy as np
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
import bench
for n in (200, 400, 600, 1000, 1112, 2224, 3000, 3336, 5000, 9601):
    off, uvp, xs, ks, r0, c0 = bench.pnp_batch(sfm, 1, n, seed=3)
    random.seed(1)
    samples = [random.sample(range(n), 6) for _ in range(300)]
    def t(fn, reps=20):
        fn(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        return (time.perf_counter() - t0) / reps * 1e3
    ransac = t(lambda: native.pnp_linear_ransac(uvp, xs, ks[0], samples, 8.0))
    its = {k: t(lambda k=k: native.pnp_nonlinear(uvp, xs, ks[0], r0[0], c0[0], 5.0, k)) for k in (1, 100, 300)}
    print("n %5d  ransac %.3f ms   nonlinear 1 / 100 / 300 iterations %.3f / %.3f / %.3f ms  -> %.2f us per iteration" % (
        n, ransac, its[1], its[100], its[300], (its[300] - its[100]) / 200 * 1e3), flush=True)
