/* sfm_hip.h — C-ABI of libsfm_hip.so: the MI355X (gfx950) back end for the nonlinear-refinement
 * hot path of willSapgreen/structure-from-motion.
 *
 * The reference is pure Python and has no FFI layer; its boundary for this path is the public
 * method surface of three classes.  Every entry point below names the reference interface it
 * replaces (file:line relative to the reference repo).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions (identical to the reference): float64 everywhere; rot = R (3x3 row-major),
 * loc = C (camera centre); world->camera p = R^T (X - C); quaternion [qw,qx,qy,qz]; a camera
 * parameter block is 7 doubles [Cx,Cy,Cz,qw,qx,qy,qz] (ba_processor.py:285-288); 3D points are
 * homogeneous columns; 2D points are pixel columns [u,v,1].
 *
 * Unless a function says "device", all pointers are HOST memory borrowed for the duration of the
 * call; outputs are caller-allocated.  Calls are blocking (they synchronise the library stream)
 * except sfm_ba_iterate / sfm_ba_linearize_reduce / sfm_ba_solve_update, which only enqueue.
 * No exception crosses the boundary: every function returns SFM_OK (0) or a negative status;
 * sfm_last_error() returns a message for the calling thread's last failure.
 */
#ifndef SFM_HIP_H
#define SFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes -------------------------------------------------------------------------- */
#define SFM_OK               0
#define SFM_E_SHAPE         -1   /* bad sizes -> ValueError (campose_processor.py:353-357, triangulation_processor.py:65-74) */
#define SFM_E_BAD_ROTATION  -2   /* verify_rotation_mat failed -> ValueError (utils.py:43-45, 93-95) */
#define SFM_E_QW_ZERO       -3   /* |qw| < 1e-6 -> ValueError (utils.py:49-51) */
#define SFM_E_SQRT_DOMAIN   -4   /* 1 + tr R < 0 -> math.sqrt ValueError (utils.py:47) */
#define SFM_E_HIP           -5   /* HIP runtime error */
#define SFM_E_NO_DEVICE     -6   /* no gfx950 device visible */
#define SFM_E_HANDLE        -7   /* null / destroyed problem handle */
#define SFM_E_RCCL          -9   /* the RCCL library could not be loaded, or one of its calls failed (sfm_comm_*, sfm_ba_set_comm) */
#define SFM_E_RANK          -8   /* rank-2 projection of a fundamental / essential matrix is not rank 2 -> ValueError (epipolar_processor.py:187-190, 90-93) */

/* ---- quirk bits; the reference's behaviour = SFM_QUIRKS_REFERENCE ------ */
#define SFM_Q1_PNP_ROW_OVERLAP  1  /* campose_processor.py:404-405: rows stored at [pt:pt+2] */
#define SFM_Q2_LOC_JAC_SIGN     2  /* campose_processor.py:802-804: sign of the v-row of d/dC */
#define SFM_QUIRKS_REFERENCE    3
/* Q13 (not a selectable bit: it is not a property of the arithmetic but of the host's LAPACK): the six-point DLT of the
 * linear PnP (campose_processor.py:565-633) negates loc together with rot when det(rot) < 0 (campose:629-631).  rot and
 * the null vector it comes from flip sign together, loc = rot @ -cam_mat[:, 3] / s does not -- so whether that branch
 * is taken, and with it whether the returned centre is C or -C, depends on the arbitrary sign LAPACK gives the
 * last right-singular vector.  Measured on the reference's own PnP fixture (tests/golden/g5_pnp.npz, 300 seeded
 * hypotheses) and on the captured per-view chains (tests/golden/g10_incremental_*.npz): the branch fires for about half
 * of the hypotheses, each of which then usually scores next to nothing -- so the reference's winner is the first
 * best hypothesis AMONG THOSE ITS LAPACK DID NOT RUIN, typically not the first best hypothesis.
 * The device returns the sign-invariant centre C for every hypothesis (sfm_pnp_linear_ransac: the sane RANSAC).
 * Reproducing the reference needs the host's LAPACK, so it is split (round 4): sfm_pnp_ransac_evaluate returns every
 * hypothesis' pose and its inlier counts under (R, C) AND under (R, -C); the Python drop-in asks NumPy -- the very
 * library the reference would have asked -- for the branch decision of the few hypotheses that can win
 * (structure-from-motion_amd/q13.py), picks the reference's winner and fetches its inlier mask with
 * sfm_pnp_inlier_mask.  tests/test_gpu_linear_and_incremental.py asserts the per-hypothesis facts on the reference's
 * fixture, tests/test_gpu_chain_golden.py the winners, inlier lists and RNG stream of whole per-view chains. */
#define SFM_Q13_PNP_LOC_SIGN_UNDEFINED 0

/* ---- Schur-product algorithm selection (sfm_ba_set_option SFM_OPT_SCHUR) ---------------------- */
#define SFM_SCHUR_AUTO    0  /* the cheapest of the three below by cost models fitted on MI355X */
#define SFM_SCHUR_PAIRS   1  /* sparse product over 18 x 18-camera LDS tiles: only camera pairs that share a point (small scenes) */
#define SFM_SCHUR_MFMA    2  /* dense v_mfma_f64_16x16x4 SYRK over the materialised, zero-filled Z (LDS-DMA staged; high visibility) */
#define SFM_SCHUR_ROWS    3  /* sparse product over LDS row panels: one observation owns its camera's block row of its point's
                              * contribution (many cameras at low visibility: BASELINE config 4) */

#define SFM_OPT_SCHUR        1
#define SFM_OPT_DEBUG        3  /* profiling ablations of the Schur kernel (1 no MFMA, 4 no staging DMA: results are wrong when set; 8 = record clock stamps; 16 = keep ba_backsub and ba_linearize as separate launches, 64 = block column steps even for P <= 56 (no single-launch small-system solve), 256 = single-launch solve up to P = 64 instead of 56, 512 = block-row back substitution instead of dp = L^-T y with the inverse carried through the column steps, 128 = never pick the row-panel sparse product, 1024 = the column steps of the reduced solve as separate launches (ba_chol_step) where the single data-flow launch would run (9 to 237 cameras), 2048 = dp = X y and the camera update as their own launch (ba_inv_apply) behind the data-flow launch, 4096 = its tasks dealt by workgroup index instead of taken by ticket (A/B only: needs every workgroup of the launch resident): results unchanged; the environment variable SFM_FLOW_SOLVE=0 selects the column-step launches (bit 1024) for every handle of the process; 8192 = test of the data-flow launch's bounded waits: one hand-over is never announced, every wait gives up after 20 000 polls and the solve reports SFM_E_HIP; 16384 = the split-K reduce of the dense product always as its own launch (sfm_ba_iterate on one GPU otherwise leaves it to the first tasks of the data-flow launch: same sums in a fixed order) */
#define SFM_OPT_DETERMINISTIC 4 /* 1: fixed summation order everywhere -- one wave per ba_linearize workgroup (ordered LDS accumulation), the
                                 * atomic-free dense Schur product, a single-writer split-K / camera-accumulator reduce.  Two runs from the
                                 * same state then agree bit for bit (the default path agrees to ~1e-13).  Needs the dense product to fit
                                 * and V <= 234; slower (C3: see DESIGN.md). */
#define SFM_OPT_GRAPH         5 /* 1: sfm_ba_iterate captures the steady-state iteration body (fused linearise, Schur product, reduce,
                                 * reduced solve) as a hipGraph -- one per camera-slot parity -- and replays it for every iteration
                                 * after the first; same kernels, same arguments, same results.  Off by default: on this stack the
                                 * kernels of an iteration already run back to back from eager launches. */
#define SFM_OPT_TIMING_STRIDE 6 /* bracket a timed kernel class only every value-th time it runs (default 1): an event pair costs two
                                 * ~6 us stream bubbles on this stack, so a measurement that must not disturb what it measures samples */
#define SFM_OPT_TIMING       2  /* bitmask (1 << SFM_K_x): bracket those kernel classes with hipEvents */

/* ---- items of sfm_ba_info -------------------------------------------------------------------------- */
#define SFM_INFO_SCHUR_KERNEL  1  /* SFM_SCHUR_PAIRS / SFM_SCHUR_MFMA / SFM_SCHUR_ROWS: the product kernel the next iteration launches
                                   * (asking builds the row-panel product's work split if that is the candidate, so the answer is
                                   * the kernel that will run, not the one that was hoped for).  NOT a pure query: when it has
                                   * to build that split it completes a deferred back substitution (as sfm_ba_flush does) and
                                   * enqueues kernels on the problem's stream -- do not ask between sfm_ba_linearize_reduce and
                                   * sfm_ba_solve_update of one iteration, nor inside a stream capture */
#define SFM_INFO_UPLOAD_BYTES  2  /* host -> device bytes moved on behalf of this handle since sfm_ba_create */
#define SFM_INFO_N_CAMS        3
#define SFM_INFO_N_PTS         4
#define SFM_INFO_N_OBS         5
#define SFM_INFO_MAX_TRACK     6  /* longest track (observations of one point) */
#define SFM_INFO_GRAPH_REPLAYS 7  /* iterations sfm_ba_iterate carried out as hipGraph replays (SFM_OPT_GRAPH) */
#define SFM_INFO_REDUCE_IN_SOLVE 8 /* 1 if the last iteration sfm_ba_iterate enqueued left the split-K reduce of the dense product to the
                                   * data-flow solve's launch (no ba_schur_reduce launch: one GPU, 37 to 237 cameras, not deterministic), else 0 */

/* ---- kernel ids for sfm_ba_kernel_time ------------------------------------------------------- */
#define SFM_K_PREP       0
#define SFM_K_LINEARIZE  1
#define SFM_K_SCHUR      2
#define SFM_K_SOLVE      3
#define SFM_K_BACKSUB    4
#define SFM_K_REDUCE     5  /* ba_schur_reduce: split-K slabs + camera accumulators -> [S | rhs] */
#define SFM_K_COUNT      6

/* ---- library / device ------------------------------------------------------------------------ */
int sfm_version(void);
/* Select the HIP device this process drives (one process per GPU) and create the library stream. */
int sfm_init(int device);
int sfm_shutdown(void);
/* Default stream of the library (hipStream_t as void*): the host-pointer entry points run on it and a new BA
 * problem starts on it.  NULL = the library's own stream (created by sfm_init, alive until sfm_shutdown).
 * A resident BA problem keeps the stream it was given (sfm_ba_set_stream), whatever this is set to later. */
int sfm_set_stream(void* hip_stream);
int sfm_synchronize(void);
/* Diagnostic: 1 when the process runs with SFM_POOL_REDZONE=1 -- every device buffer of the library then sits between
 * two 4 KB zones of 0xA5 that are checked (after a device-wide synchronise) whenever the buffer goes back to the pool;
 * a kernel that wrote outside its buffer aborts the process with a message.  For test runs only. */
int sfm_pool_redzone_active(void);
/* Diagnostic: pool mode bits (1 = SFM_POOL_REDZONE, 2 = SFM_POOL_GUARD: every device buffer is its own virtual-memory
 * mapping that ends where the buffer ends, followed by a reserved, unmapped granule -- an out-of-bounds READ past the end
 * faults at the access; one test pass on the GPU box runs under it).  *tail_slack = mapped bytes behind a probe buffer of
 * probe_bytes (guard mode: < 16; -1 before sfm_init), *guard_allocs = buffers handed out in guard mode so far. */
int sfm_pool_mode(int64_t probe_bytes, int64_t* tail_slack, int64_t* guard_allocs);
const char* sfm_last_error(void);

/* ---- unit-level helpers (parity hooks; batched) ------------------------------------------------ */
/* utils.convert_quaternion_to_rotation (utils.py:64-97).  status[i] = SFM_OK / SFM_E_BAD_ROTATION. */
int sfm_quat_to_rot(int n, const double* q /*[n][4]*/, double* R /*[n][9]*/, int* status /*[n]*/);
/* utils.convert_rotation_to_quaternion (utils.py:28-60). */
int sfm_rot_to_quat(int n, const double* R /*[n][9]*/, double* q /*[n][4]*/, int* status /*[n]*/);
/* CamposeProcessor.construct_jacobian_matrix(rot, loc, pt_3d) -> (2,7) (campose_processor.py:462-482). */
int sfm_jac_cam(int n, const double* R /*[n][9]*/, const double* C /*[n][3]*/, const double* X /*[n][4]*/,
                int quirks, double* Jp /*[n][2][7]*/, int* status /*[n]*/);
/* TriangulationProcessor.construct_jacobian_matrix(tri_3d_pt, projs, num_views) -> (2*num_views,3)
 * (triangulation_processor.py:237-271). */
int sfm_jac_pt(int n, int n_views, const double* projs /*[n][n_views][3][4]*/, const double* X /*[n][4]*/,
               double* Jx /*[n][2*n_views][3]*/);

/* ---- TriangulationProcessor.nonlinear_triangulate (triangulation_processor.py:160-234) --------- */
/* One thread per point, all iterations in registers.  Row W of X is carried through unchanged. */
int sfm_tri_nonlinear(int m, int n_views, const double* projs /*[n_views][3][4]*/,
                      const double* uv /*[n_views][2][m]*/, const double* X_in /*[4][m]*/,
                      double lambda, int iters, double* X_out /*[4][m]*/);

/* ---- TriangulationProcessor.linear_triangulate (triangulation_processor.py:91-157) ------------------ */
/* DLT: per point the null vector of the (2 n_views x 4) matrix of rows u P[2,:] - P[0,:], v P[2,:] - P[1,:],
 * divided by its W (streaming Givens QR + one-sided Jacobi SVD per thread). */
int sfm_tri_linear(int m, int n_views, const double* projs /*[n_views][3][4]*/, const double* uv /*[n_views][2][m]*/,
                   double* X_out /*[4][m]*/);
/* TriangulationProcessor.triangulate (triangulation_processor.py:31-88): linear then nonlinear, the
 * initial points never leave the device. */
int sfm_triangulate(int m, int n_views, const double* projs /*[n_views][3][4]*/, const double* uv /*[n_views][2][m]*/,
                    double lambda, int iters, double* X_out /*[4][m]*/);

/* ---- CamposeProcessor.nonlinear_estimate_cam_pose_pnp (campose_processor.py:308-459) ----------- */
int sfm_pnp_nonlinear(int n, const double* uv_pix /*[3][n]*/, const double* X /*[4][n]*/,
                      const double K[9], const double R0[9], const double C0[3],
                      double lambda, int iters, int quirks, double R_out[9], double C_out[3]);
/* Independent views in one launch (one workgroup per view); view v owns columns
 * [offsets[v], offsets[v+1]) of uv_pix / X.  status[v] per view. */
int sfm_pnp_nonlinear_batch(int n_views, const int* offsets /*[n_views+1]*/, int total,
                            const double* uv_pix /*[3][total]*/, const double* X /*[4][total]*/,
                            const double* K /*[n_views][9]*/, const double* R0 /*[n_views][9]*/,
                            const double* C0 /*[n_views][3]*/, double lambda, int iters, int quirks,
                            double* R_out /*[n_views][9]*/, double* C_out /*[n_views][3]*/,
                            int* status /*[n_views]*/);

/* ---- DEVICE-pointer, stream-ordered forms of the calls above ---------------------------------------------- */
/* Same kernels, same layouts, but every pointer is DEVICE memory, the work is only ENQUEUED on `hip_stream`
 * (hipStream_t as void*; NULL = the library stream) and the call returns without synchronising: the per-view loop of
 * the reference (PnP at ba_processor.py:191, triangulate at :246, BA at :267) can chain them on one stream around a
 * resident BA problem.  d_status of the PnP form is read by the caller after its own synchronisation
 * (SFM_OK / SFM_E_BAD_ROTATION / SFM_E_QW_ZERO / SFM_E_SQRT_DOMAIN per view).  d_X_out may equal d_X_in. */
int sfm_tri_nonlinear_dev(int m, int n_views, const double* d_projs /*[n_views][3][4]*/, const double* d_uv /*[n_views][2][m]*/,
                          const double* d_X_in /*[4][m]*/, double lambda, int iters, double* d_X_out /*[4][m]*/, void* hip_stream);
int sfm_tri_linear_dev(int m, int n_views, const double* d_projs, const double* d_uv, double* d_X_out /*[4][m]*/, void* hip_stream);
int sfm_triangulate_dev(int m, int n_views, const double* d_projs, const double* d_uv, double lambda, int iters,
                        double* d_X_out /*[4][m]*/, void* hip_stream);
int sfm_pnp_nonlinear_batch_dev(int n_views, const int* d_offsets /*[n_views+1]*/, int total, const double* d_uv_pix /*[3][total]*/,
                                const double* d_X /*[4][total]*/, const double* d_K /*[n_views][9]*/, const double* d_R0 /*[n_views][9]*/,
                                const double* d_C0 /*[n_views][3]*/, double lambda, int iters, int quirks,
                                double* d_R_out /*[n_views][9]*/, double* d_C_out /*[n_views][3]*/, int* d_status /*[n_views]*/,
                                int max_view_points /* size of the largest view (the offsets are on the device); 0 = unknown.
                                                       Views of up to 1024 points and larger ones run on different kernels, each
                                                       launched over the whole batch; a truthful value <= 1024 saves the second
                                                       launch.  A view's result depends on its own size only, never on the batch. */,
                                void* hip_stream);
/* X_out[4][n] = (px, py, pz, 1)[index[i]]: the 2D-3D association of ba_processor.py:184-188 (np.take of tri_pts) for points
 * that already live on the device (sfm_ba_points_ptr), so that the per-view PnP uploads keys and indices only. */
int sfm_gather_points_dev(int n, const int* d_index /*[n]*/, const double* d_px, const double* d_py, const double* d_pz,
                          double* d_X_out /*[4][n]*/, void* hip_stream);

/* ---- CamposeProcessor.linear_estimate_cam_pose_pnp (campose_processor.py:249-305, 485-633) ----------- */
/* RANSAC over 6-point DLT hypotheses.  The caller draws the n_hyp six-point samples (the reference uses
 * Python's global `random.sample`, campose:531, so the host keeps the RNG stream); the device solves every
 * hypothesis (12x12 null vector + 3x3 polar factor), scores all points against `threshold` in pixels and
 * returns the FIRST hypothesis with the largest inlier count, its inlier mask and count.  If no hypothesis
 * has an inlier the reference's initial identity pose is returned with *best_hypothesis = -1. */
int sfm_pnp_linear_ransac(int n, const double* uv_pix /*[3][n]*/, const double* X /*[4][n]*/, const double K[9],
                          int n_hyp, const int* samples /*[n_hyp][6]*/, double threshold,
                          double R_out[9], double C_out[3], int* inlier_mask /*[n]*/, int* n_inliers,
                          int* best_hypothesis);

/* The same hypotheses, for a caller that reproduces quirk Q13 (above): pose (R, C) of every six-point sample, its inlier
 * count under (R, C) and under (R, -C) -- what the reference scores when its det(rot) < 0 branch fired (campose:629-631). */
int sfm_pnp_ransac_evaluate(int n, const double* uv_pix /*[3][n]*/, const double* X /*[4][n]*/, const double K[9],
                            int n_hyp, const int* samples /*[n_hyp][6]*/, double threshold,
                            double* R_out /*[n_hyp][9]*/, double* C_out /*[n_hyp][3]*/, int* counts /*[n_hyp]*/,
                            int* counts_neg /*[n_hyp]*/);
/* Inlier mask and count of ONE pose: pixel reprojection error of every point against `threshold` (campose:544-554). */
int sfm_pnp_inlier_mask(int n, const double* uv_pix /*[3][n]*/, const double* X /*[4][n]*/, const double K[9],
                        const double R[9], const double C[3], double threshold, int* inlier_mask /*[n]*/, int* n_inliers);

/* CamposeProcessor.estimate_cam_pose_pnp (campose_processor.py:192-246) as two calls around the host's choice of the winner,
 * with the view's keys and points RESIDENT on the device in between: sfm_pnp_ransac_begin = sfm_pnp_ransac_evaluate + a
 * session; sfm_pnp_ransac_finish(session, R, C of the chosen hypothesis, ...) = the inlier mask of that pose, the inlier
 * columns compacted on the device in ascending order (campose:236-237) and `iters` nonlinear iterations on them
 * (campose:239) -- the result of sfm_pnp_inlier_mask followed by sfm_pnp_nonlinear on the gathered columns, bit for bit,
 * with one upload instead of three.  finish releases the session (also when it fails); sfm_pnp_session_destroy releases
 * one that is never finished. */
typedef struct sfm_pnp_session sfm_pnp_session;
int sfm_pnp_ransac_begin(int n, const double* uv_pix /*[3][n]*/, const double* X /*[4][n]*/, const double K[9],
                         int n_hyp, const int* samples /*[n_hyp][6]*/, double threshold,
                         double* R_out /*[n_hyp][9]*/, double* C_out /*[n_hyp][3]*/, int* counts /*[n_hyp]*/,
                         int* counts_neg /*[n_hyp]*/, sfm_pnp_session** out);
int sfm_pnp_ransac_finish(sfm_pnp_session* session, const double R[9], const double C[3], double threshold,
                          double lambda, int iters, int quirks, int* inlier_mask /*[n]*/, int* n_inliers,
                          double R_out[9], double C_out[3]);
int sfm_pnp_session_destroy(sfm_pnp_session* session);

/* Parity hook: every hypothesis of the RANSAC above -- pose and inlier count of each six-point sample
 * (campose_processor.py:524-560 loop body, 565-633). */
int sfm_pnp_six_point_hypotheses(int n, const double* uv_pix /*[3][n]*/, const double* X /*[4][n]*/, const double K[9],
                                 int n_hyp, const int* samples /*[n_hyp][6]*/, double threshold,
                                 double* R_out /*[n_hyp][9]*/, double* C_out /*[n_hyp][3]*/, int* counts /*[n_hyp]*/);

/* ---- Two-view initialisation ------------------------------------------------- */
/* EpipolarProcessor.determine_fundamental_mat (epipolar_processor.py:22-57 = __normalize 97-137,
 * __estimate_ransac 196-247 over __estimate_eight_pts 140-193, __denormalize 251-267).  left/right are rows 0-1
 * of the matched KeyPt arrays.  The caller draws the n_hyp eight-index samples (Python's `random.sample`,
 * epipolar:225); the device normalises the points, solves every hypothesis (8x9 null vector, rank-2 projection,
 * / f[2][2]), scores |x_r^T F x_l| < threshold on the normalised pairs and returns the FIRST hypothesis with
 * the largest inlier count, de-normalised and divided by F[2][2].  n == 8: the single estimate from all eight
 * pairs, every pair an inlier (epipolar:219-221; samples/n_hyp ignored).  n < 8: SFM_E_SHAPE.  A hypothesis whose
 * rank-2 projection has rank != 2: SFM_E_RANK (the reference raises ValueError, epipolar:187-190).  No hypothesis
 * with an inlier: *best_hypothesis = -1, mask all zero, F = NaN (the reference divides its zero matrix by
 * F[2][2] = 0, epipolar:216, 266). */
int sfm_fundamental_ransac(int n, const double* left /*[2][n]*/, const double* right /*[2][n]*/, int n_hyp,
                           const int* samples /*[n_hyp][8]*/, double threshold, double F_out[9],
                           int* inlier_mask /*[n]*/, int* n_inliers, int* best_hypothesis);
/* Parity hook: EpipolarProcessor.__estimate_eight_pts (epipolar_processor.py:140-193) for n_hyp samples of the
 * given (already normalised) pairs [n][4] = (x_l, y_l, x_r, y_r).  status[h] = SFM_OK / SFM_E_RANK. */
int sfm_fundamental_eight_point(int n, const double* pairs /*[n][4]*/, int n_hyp, const int* samples /*[n_hyp][8]*/,
                                double* F_out /*[n_hyp][9]*/, int* status /*[n_hyp]*/);
/* EpipolarProcessor.extract_essential_mat (epipolar_processor.py:60-95): E = U diag(1,1,0) V^T of
 * K_right^T F K_left, divided by E[2][2]. */
int sfm_essential_from_fundamental(const double F[9], const double K_left[9], const double K_right[9], double E_out[9]);
/* CamposeProcessor.extract_cam_pose_from_essential_mat (campose_processor.py:29-100): the two rotations
 * (R_out[0], R_out[1], camera->world convention of the reference's return value) and the centre c1 (c2 = -c1).
 * The reference's ORDER of (r1, r2) and SIGN of c1 follow LAPACK's singular-vector signs; the set of four
 * candidates {r1, r2} x {c1, -c1} is what is defined, and what callers (ba_processor.py:81-97) consume. */
int sfm_pose_candidates(const double E[9], double R_out[18], double C1_out[3]);
/* CamposeProcessor.evalulate_cam_pose_cheirality / disambiguate_cam_pose_four (campose_processor.py:102-189)
 * for k candidate (projection, point set) pairs against the reference projection P1: mask[c][i] = both depths
 * (third rows of P1 X, P2_c X) positive, counts[c] = their number, *best = FIRST candidate with the strictly
 * largest count starting from 0 (so 0 if no candidate has a valid point). */
int sfm_cheirality(int k, int n, const double P1[12], const double* P2 /*[k][12]*/, const double* X /*[k][4][n]*/,
                   int* mask /*[k][n]*/, int* counts /*[k]*/, int* best);

/* ---- BaProcessor.__execute_bundle_adjustment (ba_processor.py:274-439) ------------------------- */
/* Observations are sorted by (point, camera) — the reference's loop order (ba_processor.py:304-306)
 * — and given as a CSR over points: observation o in [pt_ptr[p], pt_ptr[p+1]) belongs to point p and
 * camera cam_idx[o]; uv_norm[0][o], uv_norm[1][o] = inv(K)[u,v,1]^T / z (ba_processor.py:339-342). */
int sfm_ba_solve(int V, int N, int64_t M, const int* pt_ptr /*[N+1]*/, const int* cam_idx /*[M]*/,
                 const double* uv_norm /*[2][M]*/, double* cams /*[V][7] in/out*/,
                 double* pts /*[3][N] in/out*/, double lambda, int iters, int quirks);

/* Device-resident problem: upload once, iterate many times (what bench.py times). */
typedef struct sfm_ba_problem sfm_ba_problem;
int sfm_ba_create(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx,
                  const double* uv_norm, sfm_ba_problem** out);
int sfm_ba_destroy(sfm_ba_problem* p);
int sfm_ba_set_option(sfm_ba_problem* p, int option, int value);
/* Run every copy and kernel of THIS problem on an existing HIP stream (e.g. a torch stream, so that an RCCL
 * all-reduce issued under it orders with the kernels).  Each problem has its own stream: several problems in
 * one process do not interfere.  NULL = the library's own stream.  Synchronises the previous stream. */
int sfm_ba_set_stream(sfm_ba_problem* p, void* hip_stream);
/* Facts about a resident problem (SFM_INFO_*). */
int sfm_ba_info(sfm_ba_problem* p, int what, int64_t* value);
int sfm_ba_set_state(sfm_ba_problem* p, const double* cams /*[V][7]*/, const double* pts /*[3][N]*/);
/* Replace only the cameras / only the points [first, first + count) of the resident state: the per-view BA call
 * of the reference (ba_processor.py:267) re-reads poses and points that did not change since the previous
 * call's write-back; the drop-in uploads what did. */
int sfm_ba_set_cameras(sfm_ba_problem* p, const double* cams /*[V][7]*/);
int sfm_ba_set_points(sfm_ba_problem* p, int first, int count, const double* pts /*[3][count]*/);
/* Enqueue `iters` damped Gauss-Newton iterations (ba_processor.py:297-406) on the library stream. */
int sfm_ba_iterate(sfm_ba_problem* p, double lambda, int iters, int quirks);
/* Per-iteration statistics without a state download (the `stats` of SURVEY.md section 8(b)): cost[i] = sum over this
 * problem's observations of |b - f|^2 in normalised image coordinates (the quantity ba_processor.py:376 minimises)
 * at the linearisation point of iteration i, for the iterations run since the state was last uploaded (set_state /
 * set_cameras / set_points / append start a new history; at most 256 iterations are kept).  sqrt(cost / M) is the RMS
 * residual; in a sharded run every rank reports its own observations.  Synchronises. */
int sfm_ba_get_stats(sfm_ba_problem* p, double* cost /*[max_iters]*/, int max_iters, int* n_iters);
/* Synchronise, copy the state back, and report the first device-side failure (bad rotation ...). */
int sfm_ba_get_state(sfm_ba_problem* p, double* cams, double* pts);
/* sfm_ba_get_state plus R(q) of every camera (what ba_processor.py:412 computes from the refined quaternions with
 * convert_quaternion_to_rotation, validated: a failing camera is reported exactly as by sfm_ba_get_state). */
int sfm_ba_get_state_rot(sfm_ba_problem* p, double* cams /*[V][7]*/, double* pts /*[3][N]*/, double* rots /*[V][9]*/);
/* q <- convert_rotation_to_quaternion(R(q)) for the resident cameras [first, first + count): the round trip through the
 * rotation matrix that the reference performs between two BA calls (view.rot = R(q) at ba_processor.py:412-413, q = q(view.rot)
 * at ba:285-288) done on the device, for cameras the caller did not change since the last write-back: the per-view BA call then
 * uploads nothing for them.  Enqueues only. */
int sfm_ba_rederive_quaternions(sfm_ba_problem* p, int first, int count);
/* Grow a resident problem in place — the incremental pipeline registers a view, triangulates new points and
 * re-runs global BA (ba_processor.py:137-267).  n_new_cams cameras (indices V..)
 * and n_new_pts points (indices N..) are appended with their initial state; n_new_obs observations of ANY
 * (camera, point) pair not yet present are merged into the (point, camera)-sorted list.  The existing keys,
 * cameras and points never leave the device and ONLY the new cameras, points and observations are uploaded
 * (SFM_INFO_UPLOAD_BYTES grows by 56 n_new_cams + 24 n_new_pts + 24 n_new_obs): the merge itself runs on the
 * device (count / scan / bucket / merge kernels), every workspace is re-planned for the new size.  The handle,
 * its options, stream and current state survive; an externally bound reduced buffer survives when no camera
 * was added, otherwise the library's own buffer takes over and the caller binds a new one (its size follows
 * V).  On error the problem is unchanged.  Blocking. */
int sfm_ba_append(sfm_ba_problem* p, int n_new_cams, const double* cams /*[n_new_cams][7]*/, int n_new_pts,
                  const double* pts /*[3][n_new_pts]*/, int64_t n_new_obs, const int* obs_cam /*[n_new_obs]*/,
                  const int* obs_pt /*[n_new_obs]*/, const double* uv_norm /*[2][n_new_obs]*/);
/* Accumulated device time of one kernel class since the last reset (needs SFM_OPT_TIMING = 1);
 * synchronises.  *launches may be NULL. */
int sfm_ba_kernel_time(sfm_ba_problem* p, int kernel_id, double* total_ms, int* launches);
int sfm_ba_reset_timing(sfm_ba_problem* p);
/* Calibration of those brackets: the average hipEvent-to-hipEvent time around a kernel that does nothing, on the
 * problem's stream, between other launches (n samples).  A bracket reads the kernel's duration PLUS this (the events'
 * own stream bubbles); bench.py subtracts it, less the ~1.5 us an empty kernel itself takes.  Synchronises. */
int sfm_ba_event_overhead(sfm_ba_problem* p, int n, double* avg_ms);
/* Diagnostic: shader-clock stamps written by instrumented kernels when SFM_OPT_DEBUG has bit 8 set. */
int sfm_ba_debug_stamps(sfm_ba_problem* p, unsigned long long* out, int n);
/* Diagnostic (host only, needs no device): the task table of the data-flow reduced solve for nbk = ceil(7 V / 32) block columns,
 * in the order its workgroups take it: {type, row, column, sort key} per task, type 0 = a block of L, 1 = the last library-side
 * block of a row + its two hand-over blocks, 2 = the hand-over block (i, i-1), 3 = a block of the rhs row, 4 = a block of an
 * identity row.  Returns the number of tasks (0 outside 2 .. 52 block columns); fills at most `capacity` of them. */
int sfm_ba_flow_tasks(int nbk, int* out, int capacity);
/* The same for n_cams cameras with the reduce deferred into the launch (sfm_ba_iterate on one GPU, dense product): type 5 = the
 * accumulators of camera `row`, part `column` of 4; type 6 = block (row, column) of S, rows 8 q .. 8 q + 7 with q = key & 3. */
int sfm_ba_flow_tasks_deferred(int n_cams, int* out, int capacity);

/* Multi-GPU split of one iteration (points sharded by rank, cameras replicated):
 *   sfm_ba_linearize_reduce : this rank's partial reduced system [S (P x P, P = 7V padded to
 *                             sfm_ba_reduced_ld) | rhs (P)] -> the reduced buffer (device)
 *   <caller all-reduces (SUM, double) the reduced buffer across ranks, e.g. RCCL via torch.distributed>
 *   sfm_ba_solve_update     : + lambda I, factor, solve, update cameras, back-substitute own points
 * sfm_ba_iterate == these two back to back on one rank. */
int sfm_ba_linearize_reduce(sfm_ba_problem* p, double lambda, int quirks);
int sfm_ba_solve_update(sfm_ba_problem* p, double lambda, int quirks);
/* sfm_ba_solve_update may leave the back substitution of the points (ba_processor.py:405-406) to the next
 * sfm_ba_linearize_reduce, whose launch then does both in one pass over the observations.  Every entry point that
 * reads or replaces the state completes it first; a loop that only times linearize_reduce / solve_update calls ends
 * with sfm_ba_flush so that its last iteration is complete as well.  Enqueues only. */
int sfm_ba_flush(sfm_ba_problem* p);
/* DEVICE pointers of the resident points (SoA, n_pts doubles each) and the stream the problem runs on: the current
 * state after everything enqueued so far (a deferred back substitution is enqueued first).  Valid until the next
 * sfm_ba_append / sfm_ba_destroy. */
int sfm_ba_points_ptr(sfm_ba_problem* p, void** d_px, void** d_py, void** d_pz, int* n_pts);
int sfm_ba_stream(sfm_ba_problem* p, void** hip_stream);
/* DEVICE pointer + element count of the contiguous reduced buffer (doubles).  It holds [S | rhs] between sfm_ba_linearize_reduce and
 * sfm_ba_solve_update (what the caller all-reduces); the factorisation overwrites it, and sfm_ba_iterate on one GPU may never form S
 * in it at all (SFM_INFO_REDUCE_IN_SOLVE) -- use sfm_ba_reduced_system to look at S. */
int sfm_ba_reduced_buffer(sfm_ba_problem* p, void** device_ptr, int64_t* n_doubles, int* ld);
/* Bind an externally owned DEVICE buffer (e.g. a torch tensor) as the reduced buffer. */
int sfm_ba_bind_reduced_buffer(sfm_ba_problem* p, void* device_ptr, int64_t n_doubles);

/* ---- the exchange step inside the library ("library owns its RCCL communicators") --------------
 * One process per GPU.  Rank 0 calls sfm_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the
 * application has (MPI, a file, torch.distributed ...); every rank then calls sfm_comm_create (ncclCommInitRank on the
 * device of sfm_init) and attaches the communicator to its shard's problem.  From then on sfm_ba_iterate IS the sharded
 * loop: per iteration linearise + partial reduce, ncclAllReduce(SUM, double) of the packed [S | rhs] buffer on the
 * problem's stream, replicated atomic-free solve, back substitution of the rank's own points -- K iterations enqueued by
 * ONE call, no host code between them, a plain C program included.  RCCL is loaded with dlopen on first use (librccl.so.1
 * / librccl.so: a process that already carries torch's copy gets that one); the library has no link-time dependency on it.
 * The reference has no distributed code; the algebra is ba_processor.py:376-406 (sums over points). */
typedef struct sfm_comm sfm_comm;
/* SFM_OK when RCCL can be loaded in this process (so that all ranks can agree on the library path BEFORE any of them
 * enters ncclCommInitRank), SFM_E_RCCL otherwise. */
int sfm_comm_available(void);
int sfm_comm_unique_id(char id_out[128]);
int sfm_comm_create(int world_size, int rank, const char id[128], sfm_comm** out);
int sfm_comm_destroy(sfm_comm* comm);
/* Attach (or, with NULL, detach) a communicator: the problem's iterations then all-reduce the reduced buffer that is
 * bound at that moment (the library's own, or a caller's: sfm_ba_bind_reduced_buffer). */
int sfm_ba_set_comm(sfm_ba_problem* p, sfm_comm* comm);

/* Parity hooks: per-observation terms and the reduced system at the given state (one linearisation,
 * ba_processor.py:317-382).  S is (7V x 7V) row-major (both triangles filled), rhs (7V). */
int sfm_ba_residual_jacobian(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx,
                             const double* uv_norm, const double* cams, const double* pts, int quirks,
                             double* r /*[M][2]*/, double* Jp /*[M][2][7]*/, double* Jx /*[M][2][3]*/);
int sfm_ba_reduced_system(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx,
                          const double* uv_norm, const double* cams, const double* pts,
                          double lambda, int quirks, int schur_mode,
                          double* S /*[7V][7V]*/, double* rhs /*[7V]*/);

#ifdef __cplusplus
}
#endif
#endif /* SFM_HIP_H */
