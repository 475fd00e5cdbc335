"""Data-flow solve (sfm_ba_flow.h) against the column-step launches (SFM_OPT_DEBUG bit 1024) and the oracle on scenes of 9-70
cameras; with `stamps` as the first argument: the chain's phase stamps at C3."""
import importlib, json, os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "oracle"))
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)

def run(sc, uvn, dbg, iters, lam=5.0):
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(native.OPT_DEBUG, dbg)
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(lam, iters)
        return prob.get_state()

if len(sys.argv) > 1 and sys.argv[1] == "stamps":
    sc = sfm.scenes.make_config("C3", seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(native.OPT_DEBUG, 8 | (int(sys.argv[2]) if len(sys.argv) > 2 else 0))
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(5.0, 3); native.synchronize()      # the stamps of the third solve (the first runs the 61 KB kernel from a cold instruction cache)
        raw = prob.debug_stamps(1024).astype(np.int64)
    el, pr = raw[:128].reshape(16, 8), raw[512:640].reshape(16, 8)
    t0 = el[0, 0]
    for j in range(11):
        print("step %2d  start %6d  elimination beside the preparation %5d  L[j+1][j] %4d  D_j+1 %4d | preparation: starts %6d, done %6d, L[j+1][j] announced %6d (rel. to step start)" % (
            j, el[j, 0] - t0, el[j, 1] - el[j, 0], el[j, 3] - el[j, 2], el[j, 4] - el[j, 3], pr[j, 5] - el[j, 0], pr[j, 6] - el[j, 0], pr[j, 7] - el[j, 0]))
    pd = raw[768:896].reshape(16, 8)
    for j in range(10):
        print("prep %2d (rel. to step start): seen %6d  regs->LDS %6d  sync %6d  d'+trsm %6d  sync %6d  done %6d" % (
            j, pr[j, 5] - el[j, 0], pd[j, 0] - el[j, 0], pd[j, 1] - el[j, 0], pd[j, 3] - el[j, 0], pd[j, 4] - el[j, 0], pr[j, 6] - el[j, 0]))
    print("chain total ticks", el[10, 2] - t0, " kernel entry -> D_0 in LDS, first step starts:", t0 - el[0, 5])
    rt = raw[896:1024].reshape(16, 8)      # chain, constant 100 MHz clock (s_memrealtime): [0] step start, [1] preparation starts, [2] row in LDS, [3] W_j announced, [4] L[j+1][j] announced
    cl = raw[640:768].reshape(16, 8)       # closer of row i, same clock: [0] start, [6] last term of the sums there, [1] sums done, [5] W seen, [2] L[i-2][k] seen, [3] L formed, [4] hand-over announced
    t1 = raw[128:256].reshape(16, 8)       # task L[i][i-4], same clock: [0] start, [1] sum done, [2] W seen, [3] announced
    r0 = rt[0, 0]
    if t1[0, 0]:      # deferred reduce: the way to D_0 (us since the chain workgroup's entry)
        k0 = el[0, 6]; u0 = lambda v: (int(v) - int(k0)) / 100.0
        print("deferred reduce, us since the chain's kernel entry: camera sums (0, part 0) start %.2f, rows summed %.2f, published %.2f | S(1,0) rows 0-7: start %.2f, slabs summed %.2f, camera sums seen %.2f, published %.2f | chain: D_0's slabs summed %.2f, camera sums seen %.2f, D_0 in LDS (thread 0) %.2f, first step starts %.2f" % (
            u0(t1[0, 0]), u0(t1[0, 1]), u0(t1[0, 2]), u0(t1[1, 0]), u0(t1[1, 1]), u0(t1[1, 2]), u0(t1[1, 3]), u0(el[0, 7]), u0(el[1, 5]), u0(el[1, 6]), u0(r0)))
    us = lambda v: (int(v) - int(r0)) / 100.0
    print("cross-workgroup timeline, us on the 100 MHz clock since the chain's first step")
    for i in range(4, 11):
        k = i - 3
        print("  row %2d | chain: W_%d announced %6.2f, W_%d announced %6.2f, L[%d][%d] announced %6.2f | task L[%d][%d]: sum done %6.2f, W_%d seen %6.2f, announced %6.2f | closer: last term there %6.2f, sums done %6.2f, W_%d seen %6.2f, L seen %6.2f, L formed %6.2f, hand-over announced %6.2f | chain step %d: start %6.2f, preparation starts %6.2f, row in LDS %6.2f" % (
            i, k - 1, us(rt[k - 1, 3]), k, us(rt[k, 3]), k + 1, k, us(rt[k, 4]), i, k - 1, us(t1[i, 1]), k - 1, us(t1[i, 2]), us(t1[i, 3]),
            us(cl[i, 6]), us(cl[i, 1]), k, us(cl[i, 5]), us(cl[i, 2]), us(cl[i, 3]), us(cl[i, 4]), i - 1, us(rt[i - 1, 0]), us(rt[i - 1, 1]), us(rt[i - 1, 2])))
    sys.exit(0)

oracle = importlib.import_module("sfm_oracle")
worst = 0.0
for (V, N, vis, seed) in ((9, 300, 0.8, 1), (10, 400, 0.6, 2), (14, 500, 0.5, 3), (19, 600, 0.5, 4), (28, 600, 0.4, 5), (37, 700, 0.4, 6), (50, 900, 0.6, 7), (64, 900, 0.3, 8), (73, 900, 0.3, 9), (74, 900, 0.3, 10), (120, 1500, 0.2, 11), (200, 2500, 0.15, 12), (237, 2500, 0.15, 13)):
    sc = sfm.scenes.make_scene(V, N, vis, seed=seed); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    t0 = time.time()
    c_new, p_new = run(sc, uvn, 0, 2)
    c_old, p_old = run(sc, uvn, 1024, 2)
    c_own, p_own = run(sc, uvn, 16384, 2)      # the reduce as its own launch
    c_dense, _ = run(sc, uvn, 0, 2) if V > 140 else (c_new, None)
    oc, op = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    r_old = max(float(np.max(np.abs(c_new - c_old)) / np.max(np.abs(c_old))), float(np.max(np.abs(c_new - c_own)) / np.max(np.abs(c_own))))
    r_or = float(np.max(np.abs(c_new - oc)) / np.max(np.abs(oc)))
    r_pt = float(np.max(np.abs(p_new - op)) / np.max(np.abs(op)))
    worst = max(worst, r_or, r_pt)
    print("V=%d nbk=%d: flow vs column steps / own reduce launch %.2e, vs oracle cams %.2e pts %.2e  (%.1f s)" % (V, (7 * V + 31) // 32, r_old, r_or, r_pt, time.time() - t0), flush=True)
print("worst", worst)
sys.exit(0 if worst < 1e-9 else 1)
