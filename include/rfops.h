/*
 * rfops.h -- C ABI of librfops.so: RFNet's point-cloud operator hot path on MI355X (gfx950).
 *
 * This is the drop-in boundary.  Each entry point replaces one of the reference's
 * C++-mangled "Launcher" functions that its TensorFlow OpKernels call (the native ABI
 * under tf_ops/ and pc_distance/); the reference interface each
 * one replaces is cited as file:line.  Conventions, identical to the reference's:
 *
 *   - all pointers are DEVICE pointers (HIP), row-major contiguous; xyz tensors are
 *     (b, npts, 3) float32, index tensors int32;
 *   - the caller owns every buffer (outputs, gradients, scratch); the library never
 *     allocates or frees device memory, and never synchronises the stream;
 *   - gradient outputs are zero-filled by the call itself (the reference's ops do the
 *     cudaMemset inside the launcher / OpKernel) -- by a kernel, never hipMemset*: every call can
 *     be captured into a HIP graph and replayed;
 *   - kernels are stateless and re-entrant; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream).  The reference launches
 *     on the legacy default stream with no error checking; here every call returns a status.
 *     The library keeps no state between calls and reads no environment variables; the one
 *     exception is the opt-in measurement hook at the end of this file (rf_profile_*), which is
 *     process-global, thread-safe and off by default;
 *   - every call that launches work first checks that the calling thread's current HIP device
 *     is a gfx950 (the only code objects in the library) and returns RF_ENODEVICE otherwise;
 *   - `workspace` buffers and sorted-set handles (rf_nn_sort) must be 16-byte aligned -- the kernels read
 *     them with 16-byte vector loads at 256-byte-multiple offsets; any hipMalloc pointer is.  A misaligned
 *     one is RF_EINVAL (culled Chamfer paths, approx_match, earth_mover), not a fault inside a kernel.
 *     Tensor arguments need their element's natural alignment (4 bytes), except rf_point_affine's
 *     (16 bytes: rows of c % 4 == 0 floats).
 *
 * Status codes: 0 = success; > 0 = the hipError_t of the failing HIP call;
 *               < 0 = RF_EINVAL-style argument errors below.
 *
 * No torch / TensorFlow types appear here; bind with ctypes, cgo, JNI ... as needed
 * (INTEGRATION.md shows the reference-side stub).
 */
#ifndef RFOPS_H_
#define RFOPS_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_OK 0
#define RF_EINVAL (-1)     /* negative size, NULL pointer with non-zero size, bad attribute, misaligned workspace */
#define RF_EWORKSPACE (-2) /* workspace smaller than rf_*_workspace_bytes() says            */
#define RF_ENODEVICE (-3)  /* no gfx950 device / code object could not be loaded            */

typedef void *rf_stream_t; /* hipStream_t */

const char *rf_version(void);
const char *rf_status_string(int status);
/* RF_OK when the calling thread's current HIP device is a gfx950, RF_ENODEVICE otherwise
 * (no device visible, or another architecture).  Every launching entry point makes this check. */
int rf_device_check(void);

/* ---------------------------------------------------------------- Chamfer (tf_ops/CD) --- */
/* Replaces NmDistanceKernelLauncher(b,n,xyz,m,xyz2,result,result_i,result2,result2_i)
 * (tf_ops/CD/tf_nndistance.cpp:168, tf_nndistance_g.cu:127-130; same op duplicated under
 * pc_distance/).  dist1[i,j] = min_k |xyz2[i,k]-xyz1[i,j]|^2, idx1 = lowest argmin;
 * dist2/idx2 symmetric.  `workspace`: per-split partial minima (dense sweep) or the sorted clouds
 * and their boxes (culled sweep); rf_nn_distance_workspace_bytes sizes it for the sweep that
 * rf_nn_distance picks for this shape. */
size_t rf_nn_distance_workspace_bytes(int b, int n, int m);
int rf_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                   int *idx1, float *dist2, int *idx2, void *workspace, size_t workspace_bytes,
                   rf_stream_t stream);

/* Same call with the sweep pinned: RF_NN_DENSE evaluates all b*n*m pairs (nn_distance.hip),
 * RF_NN_CULLED sorts both clouds along a space-filling curve and skips blocks of candidates whose
 * bounding box is strictly farther than every query's current minimum (nn_pruned.hip; n, m <=
 * 65536) -- identical outputs, bit for bit, ties included.  RF_NN_AUTO (what rf_nn_distance
 * uses) picks by size.  `stats` (host pointer to 32 counters, or NULL; filled by the culled sweep
 * only): per direction d at [4d..4d+3] {waves, superblock steps, most steps of one wave, 16-candidate
 * block scans}, [8+d] most block scans of one wave, [12+d] directed pairs evaluated (summed by the kernel), [14+d] directed pairs per
 * counted block scan of the last wave of direction d to report (1024: 64 queries x 16 candidates; 16 where a query has four lanes of
 * its own, nn_pruned.hip sweep_tile16; a launch may mix both), [16..31] phase time stamps of the sort; a non-NULL
 * pointer synchronises the stream. */
#define RF_NN_AUTO 0
#define RF_NN_DENSE 1
#define RF_NN_CULLED 2
size_t rf_nn_distance_mode_workspace_bytes(int b, int n, int m, int mode);
int rf_nn_distance_mode(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                        int *idx1, float *dist2, int *idx2, void *workspace, size_t workspace_bytes,
                        rf_stream_t stream, int mode, unsigned long long *stats);

/* Replaces NmDistanceGradKernelLauncher (tf_nndistance.cpp:208, tf_nndistance_g.cu:151-156).
 * grad_xyz1 (b,n,3) and grad_xyz2 (b,m,3) are zero-filled here, then accumulated. */
int rf_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                        const float *grad_dist1, const int *idx1, const float *grad_dist2,
                        const int *idx2, float *grad_xyz1, float *grad_xyz2, rf_stream_t stream);

/* ---- one direction, sorted-cloud handles, one-call step (Chamfer, continued) ------------- */
/* The same op with only the direction(s) a caller uses.  The reference's glue calls
 * nn_distance(xyz1, xyz2) and then drops outputs: merge_layer keeps idx2 alone
 * (vv_recon.py:134-135), fidelity_loss dist1 (:386-390), zero_groupnear dist2 (:415-419);
 * NmDistanceKernelLauncher (tf_nndistance_g.cu:127-130) is two independent kernel launches, one per
 * direction, so a TF-side op with a "directions" attribute maps onto this entry.  want1: dist1/idx1
 * (nearest neighbour of every xyz1 point in xyz2); want2: dist2/idx2.  Outputs of a direction that
 * is not wanted may be NULL and are not written.  Bit-identical to rf_nn_distance's. */
size_t rf_nn_distance_dir_workspace_bytes(int b, int n, int m, int want1, int want2);
int rf_nn_distance_dir(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                       int *idx1, float *dist2, int *idx2, void *workspace, size_t workspace_bytes,
                       rf_stream_t stream, int want1, int want2);

/* A cloud that takes part in several Chamfers (the model Chamfers `pointcloud` 3x and `gt` 5x per
 * training step, vv_recon.py:213,225,238 and :484-498) can be put in space-filling-curve order
 * ONCE.  `sorted` is a caller-owned device buffer of rf_nn_sort_bytes(b, n) bytes whose layout is a
 * pure function of (b, n) (records, original indices and block boxes in key order): the library
 * keeps no state, the buffer IS the handle, valid for as long as the caller keeps it and xyz is
 * unchanged.  rf_nn_distance_sorted runs the culled exact sweep on two such buffers; a direction
 * whose outputs are NULL is skipped.  n, m <= 65536.  Same results as rf_nn_distance. */
size_t rf_nn_sort_bytes(int b, int n);
int rf_nn_sort(int b, int n, const float *xyz, void *sorted, size_t sorted_bytes, rf_stream_t stream);
int rf_nn_distance_sorted(int b, int n, int m, const void *sorted1, const void *sorted2, float *dist1,
                          int *idx1, float *dist2, int *idx2, rf_stream_t stream);

/* NnDistance followed by NnDistanceGrad on its own indices (what one training step of the
 * reference's Chamfer bench does, tf_ops/CD/tf_nndistance.py:35-61) in ONE call on caller-owned
 * buffers: one FFI crossing per step, nothing allocated.  Equivalent to rf_nn_distance +
 * rf_nn_distance_grad(..., grad_dist1, idx1, grad_dist2, idx2, ...): forward outputs bit-identical, gradients
 * within the backward's tolerance (fp32 add order of a scatter).  On shapes that take the culled sweep the
 * forward also leaves, in the workspace and in sorted query order, every query's winner position and own
 * gradient term, and the backward runs in sorted index space (nnp_grad_sorted_kernel): the
 * workspace is larger than rf_nn_distance's there (rf_chamfer_step_workspace_bytes says by how much). */
size_t rf_chamfer_step_workspace_bytes(int b, int n, int m);
int rf_chamfer_step(int b, int n, int m, const float *xyz1, const float *xyz2,
                    const float *grad_dist1, const float *grad_dist2, float *dist1, int *idx1,
                    float *dist2, int *idx2, float *grad_xyz1, float *grad_xyz2, void *workspace,
                    size_t workspace_bytes, rf_stream_t stream);

/* ----------------------------------------------------------- EMD (pc_distance) ---------- */
/* Replaces approxmatchLauncher(b,n,m,xyz1,xyz2,match,temp) (pc_distance/tf_approxmatch.cpp:141,
 * tf_approxmatch.cu:180-182).  xyz1 (b,n,3) "dataset", xyz2 (b,m,3) "query" (b <= 65535 for
 * every entry point of this section: the batch is a grid dimension); match is
 * (b,m,n) (tf_approxmatch.cpp:164).  The reference's `temp` (b,2(n+m)) becomes `workspace`
 * (larger: it keeps the per-level ratio vectors so that match is written once). */
size_t rf_approxmatch_workspace_bytes(int b, int n, int m, int nlevels /* 0 = reference's 10 */);
int rf_approxmatch(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                   void *workspace, size_t workspace_bytes, rf_stream_t stream);
/* Same with an explicit annealing schedule (host array of `nlevels` values, each the
 * multiplier of d^2 inside exp(); the reference's is {-4^7..-4^-1, 0}, tf_approxmatch.cu:21-25). */
int rf_approxmatch_levels(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                          const float *levels_host, int nlevels, void *workspace,
                          size_t workspace_bytes, rf_stream_t stream);

/* The same op with its route pinned.  The reference's kernel loops over the samples of a batch independently
 * (tf_approxmatch.cu:13), so a sample's match does not depend on the batch it is called in.  RF_EMD_AUTO (what rf_approxmatch /
 * rf_approxmatch_levels / rf_earth_mover pass) picks launch shapes and routes by the size of the WHOLE batch: from 6e7 pairs on
 * the sharp levels' sweeps take their rows in spatial order and skip columns, and the column segments of a sweep follow b --
 * every route within the op's tolerances, but the bits of sample i may differ between a call on the batch and a call on that
 * sample alone.  RF_EMD_SWEPT pins the route: every level a dense sweep over the clouds in the caller's order, launch shapes
 * those of a batch of one -- sample i's match (and rf_earth_mover_mode's cost) is bit-identical whatever the batch around it or
 * the shard it lands in (what a loss compared across differently sharded runs wants; C4 costs ~1.5x the time).
 * levels_host == NULL with nlevels == 0: the reference schedule. */
#define RF_EMD_AUTO 0
#define RF_EMD_SWEPT 1
size_t rf_approxmatch_mode_workspace_bytes(int b, int n, int m, int nlevels /* 0 = reference's 10 */, int mode);
int rf_approxmatch_mode(int b, int n, int m, const float *xyz1, const float *xyz2, float *match,
                        const float *levels_host, int nlevels, void *workspace, size_t workspace_bytes,
                        rf_stream_t stream, int mode);

/* Replaces matchcostLauncher (tf_approxmatch.cpp:142, tf_approxmatch.cu:226-228): cost (b). */
size_t rf_matchcost_workspace_bytes(int b, int n, int m);
int rf_matchcost(int b, int n, int m, const float *xyz1, const float *xyz2, const float *match,
                 float *cost, void *workspace, size_t workspace_bytes, rf_stream_t stream);

/* Replaces matchcostgradLauncher (tf_approxmatch.cpp:143, tf_approxmatch.cu:292-295):
 * grad1 (b,n,3), grad2 (b,m,3), fully overwritten. */
int rf_matchcost_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                      const float *match, float *grad1, float *grad2, rf_stream_t stream);

/* ------------------------------------------------------- sampling (tf_ops/sampling) ------ */
/* Replaces farthestpointsamplingLauncher(b,n,m,inp,temp,out) (tf_sampling.cpp:94,
 * tf_sampling_g.cu:203-205).  inp (b,n,3); out (b,m) int32; temp: caller scratch of
 * b*n floats, used only when n exceeds the register-resident limit (may be NULL otherwise,
 * see rf_farthestpointsampling_temp_floats). */
size_t rf_farthestpointsampling_temp_floats(int b, int n);
int rf_farthestpointsampling(int b, int n, int m, const float *inp, float *temp, int *out,
                             rf_stream_t stream);

/* The op with caller scratch of a stated size -- the entry point a binding should prefer: for clouds of 4097..16384 points from
 * 256 samples on (2049..4096 points: from 512) it sorts the cloud into the workspace and samples over the sorted cloud (rf_farthestpointsampling_sorted
 * below: the same indices, iterations a quarter shorter), otherwise it is rf_farthestpointsampling with `workspace` as `temp`.
 * workspace: rf_farthestpointsampling_workspace_bytes(b, n, m) bytes, 16-byte aligned (may be NULL when that is 0). */
size_t rf_farthestpointsampling_workspace_bytes(int b, int n, int m);
int rf_farthestpointsampling_ws(int b, int n, int m, const float *inp, void *workspace, size_t workspace_bytes, int *out,
                                rf_stream_t stream);

/* The same op (same indices, bit for bit) over the spatially sorted cloud: the call sorts the cloud into `workspace`
 * (rf_farthestpointsampling_sorted_workspace_bytes(b, n) bytes, 16-byte aligned) and a new sample then only re-scans the
 * regions it can still change (sampling.hip fps_sorted_kernel).  1024 < n <= 16384, m <= the kernel's capacity (1024 x 2, 4, 8 or 16 points: the power of two that holds n).  form: reserved (0).
 * new_xyz (b, m, 3), may be NULL: the samples' coordinates. */
size_t rf_farthestpointsampling_sorted_workspace_bytes(int b, int n);
int rf_farthestpointsampling_sorted(int b, int n, int m, int form, const float *inp, void *workspace,
                                    size_t workspace_bytes, int *out, float *new_xyz, rf_stream_t stream);

/* Replaces gatherpointLauncher (tf_sampling.cpp:125, tf_sampling_g.cu:206-208). */
int rf_gatherpoint(int b, int n, int m, const float *inp, const int *idx, float *out,
                   rf_stream_t stream);
/* Replaces scatteraddpointLauncher + the cudaMemset before it (tf_sampling.cpp:150,174). */
int rf_scatteraddpoint(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                       rf_stream_t stream);

/* ------------------------------------------------------- grouping (tf_ops/grouping) ------ */
/* Replaces queryBallPointLauncher(b,n,m,radius*,nsample,xyz1,xyz2,idx,pts_cnt)
 * (tf_grouping.cpp:67, tf_grouping_g.cu:125-128).  xyz1 (b,n,3) dataset, xyz2 (b,m,3)
 * queries; idx (b,m,nsample), pts_cnt (b,m).  The reference passes `radius` as a device
 * pointer to one float (an op input tensor); here it is passed by value.  Rows with an
 * empty ball are left untouched, as in the reference. */
int rf_queryballpoint(int b, int n, int m, float radius, int nsample, const float *xyz1,
                      const float *xyz2, int *idx, int *pts_cnt, rf_stream_t stream);
/* The reference's exact signature: `radius_dev` is a DEVICE pointer to one float, the op's input
 * tensor (tf_grouping.cpp:18,93-95; queryBallPointLauncher(b,n,m,radius*,...), :67).  A TF-side
 * binder passes the tensor's buffer straight through: no D2H copy, no stream synchronisation.
 * Same results as rf_queryballpoint for the same radius value. */
int rf_queryballpoint_dev(int b, int n, int m, const float *radius_dev, int nsample,
                          const float *xyz1, const float *xyz2, int *idx, int *pts_cnt,
                          rf_stream_t stream);

/* The same op (same idx and pts_cnt, bit for bit) for datasets of 64 .. 65536 points and nsample <= 64, with
 * caller scratch: the dataset is put in sort-tile-recursive order once (or comes as an rf_nn_sort handle in
 * `sorted1`, NULL otherwise) and every query tests only the 64-record blocks whose box lies within the radius,
 * then keeps the nsample lowest ORIGINAL indices in ascending order (tf_grouping_g.cu:18-33); queries whose ball
 * reaches a large share of the cloud, non-finite queries and clouds with a non-finite point walk the cloud in index
 * order inside the same launch.  radius_dev: NULL (use `radius`) or the reference's device scalar (then `radius` is
 * ignored).  RF_EINVAL outside that domain: use rf_queryballpoint there.  A TF-side binder allocates the scratch
 * with allocate_temp, as tf_sampling.cpp:115 does for FPS. */
size_t rf_queryballpoint_boxes_workspace_bytes(int b, int n);
int rf_queryballpoint_boxes(int b, int n, int m, float radius, const float *radius_dev, int nsample,
                            const float *xyz1, const float *xyz2, const void *sorted1, int *idx, int *pts_cnt,
                            void *workspace, size_t workspace_bytes, rf_stream_t stream);

/* The set-abstraction chain of BASELINE.json configs[2] as ONE call on caller buffers:
 *     fps_idx = farthest_point_sample(npoint, xyz)          (tf_sampling_g.cu:105-170)
 *     new_xyz = gather_point(xyz, fps_idx)                  (tf_sampling_g.cu:172-181)
 *     idx, pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)   (tf_grouping_g.cu:3-36)
 *     grouped_xyz = group_point(xyz, idx)                   (tf_grouping_g.cu:40-57)
 * with the results of the four separate entry points, bit for bit (rows of empty balls -- impossible for radius > 1e-20,
 * every sample lies in its own ball -- are written as index 0 instead of being left untouched, because the grouping reads
 * them).  xyz (b,n,3) with 64 <= n <= 65536, nsample <= 64; fps_idx (b,npoint) int32, new_xyz (b,npoint,3), idx
 * (b,npoint,nsample) int32, pts_cnt (b,npoint) int32, grouped_xyz (b,npoint,nsample,3).  radius_dev: NULL or the reference's
 * device scalar.  Two launches on `stream` (FPS writes new_xyz itself, the ball query writes grouped_xyz itself) plus the
 * dataset's sort, which runs on `aux_stream` beside FPS when one is given (NULL: on `stream`, before FPS); the call creates
 * and releases the two events that order the streams, keeps no state and is graph-capturable.  A TF-side op for a fused
 * "SampleAndGroup" would bind this with its four outputs from allocate_output and the scratch from allocate_temp. */
size_t rf_sample_and_group_workspace_bytes(int b, int n);
int rf_sample_and_group(int b, int n, int npoint, float radius, const float *radius_dev, int nsample, const float *xyz,
                        int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz, void *workspace,
                        size_t workspace_bytes, rf_stream_t stream, rf_stream_t aux_stream);

/* Replaces groupPointLauncher / groupPointGradLauncher (tf_grouping.cpp:146,177,208).
 * points (b,n,c); idx (b,m,nsample); out / grad_out (b,m,nsample,c); grad_points (b,n,c)
 * zero-filled here. */
int rf_grouppoint(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                  float *out, rf_stream_t stream);
int rf_grouppoint_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                       const int *idx, float *grad_points, rf_stream_t stream);
/* The gradient with caller scratch -- the entry point a binding should prefer.  The reference's form (one atomicAdd per element
 * into the zeroed tensor, tf_grouping_g.cu:61-78) is bound by the L2's float-atomic rate on this chip; given
 * rf_grouppoint_grad_workspace_bytes(b, n, c, m, nsample) bytes of 16-byte aligned scratch (0: the shape stays on the atomics, pass
 * NULL) the slots are counting-sorted by destination row and every row of grad_points is written once, from sums in double:
 * no zero fill, no atomics on memory, the same values to fp32 rounding whatever the order (scatter_rows.hip).  Slots whose index
 * is outside [0, n) add to no row.  With workspace == NULL or too small: rf_grouppoint_grad. */
size_t rf_grouppoint_grad_workspace_bytes(int b, int n, int c, int m, int nsample);
int rf_grouppoint_grad_ws(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                          float *grad_points, void *workspace, size_t workspace_bytes, rf_stream_t stream);

/* -------------------------------------------------- interpolation (tf_ops/interpolation) - */
/* Replace threenn_cpu / threeinterpolate_cpu / threeinterpolate_grad_cpu
 * (tf_interpolate.cpp:60-153; CPU-only ops in the reference).  xyz1 (b,n,3) unknown,
 * xyz2 (b,m,3) known; dist/idx (b,n,3).  points (b,m,c), weight (b,n,3), out (b,n,c);
 * grad_points (b,m,c) zero-filled here. */
int rf_threenn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
               rf_stream_t stream);
/* The same op over spatially sorted copies of the two sets (the Chamfer sweep's sort, rf_nn_sort): a wave of 64
 * neighbouring unknown points visits only the candidate blocks whose box can still hold one of its three nearest.
 * dist / idx bit-identical to rf_threenn (ties included: the three smallest by (distance, index), which is what
 * the scan's strict '<' insertion in index order yields, tf_interpolate.cpp:78-93).  1 <= n, m <= 65536,
 * b <= 65535 (else RF_EINVAL: use rf_threenn).  sorted1 / sorted2: rf_nn_sort handles of xyz1 / xyz2 or NULL
 * (sorted here, into the workspace).  workspace: rf_threenn_boxes_workspace_bytes(b, n, m) bytes, 16-byte
 * aligned (0 = outside the domain).  Pays from about 1e8 pairs per call on (the sort is ~25 us). */
size_t rf_threenn_boxes_workspace_bytes(int b, int n, int m);
int rf_threenn_boxes(int b, int n, int m, const float *xyz1, const float *xyz2, const void *sorted1,
                     const void *sorted2, float *dist, int *idx, void *workspace, size_t workspace_bytes,
                     rf_stream_t stream);
int rf_threeinterpolate(int b, int m, int c, int n, const float *points, const int *idx,
                        const float *weight, float *out, rf_stream_t stream);
int rf_threeinterpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                             const float *weight, float *grad_points, rf_stream_t stream);
/* ... with caller scratch: where a sample's known points do not fit the LDS tile of the in-kernel form (more than 2048 at 8
 * channels per slice) the 3 n slots are counting-sorted by known point and every row of grad_points is written once
 * (scatter_rows.hip, as rf_grouppoint_grad_ws).  rf_threeinterpolate_grad_workspace_bytes = 0: no scratch needed, pass NULL. */
size_t rf_threeinterpolate_grad_workspace_bytes(int b, int n, int c, int m);
int rf_threeinterpolate_grad_ws(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                const float *weight, float *grad_points, void *workspace, size_t workspace_bytes,
                                rf_stream_t stream);

/* ------------------------------------------- rest of the import surface ("next" row f3) --- */
/* Replaces AuctionMatchLauncher(b,n,xyz1,xyz2,matchl,matchr,cost) (tf_ops/emd/tf_auctionmatch.cpp:25,
 * tf_auctionmatch_g.cu:292-294).  xyz1, xyz2 (b,n,3); matchl, matchr (b,n) int32: matchr[j] = the
 * xyz1 point assigned to xyz2 point j, matchl its inverse.  `workspace` is the reference's temp
 * cost matrix (b,n,n) floats.  Defined for n < 1024 or n in {1024, 2048, 4096}
 * (rf_auctionmatch_supported); the reference additionally caps n at 4096. */
int rf_auctionmatch_supported(int n);
size_t rf_auctionmatch_workspace_bytes(int b, int n);
int rf_auctionmatch(int b, int n, const float *xyz1, const float *xyz2, int *matchl, int *matchr,
                    void *workspace, size_t workspace_bytes, rf_stream_t stream);

/* Replaces selectionSortLauncher(b,n,m,k,dist,outi,out) (tf_ops/grouping/tf_grouping.cpp:112,
 * tf_grouping_g.cu:129-132).  dist (b,m,n); outi (b,m,n) int32, out (b,m,n): per row a partial
 * selection sort, the first k entries are the k smallest in ascending order.  n <= 16384. */
int rf_selectionsort(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                     rf_stream_t stream);

/* Replaces probsampleLauncher(b,n,m,inp_p,inp_r,temp,out) (tf_ops/sampling/tf_sampling.cpp:65,
 * tf_sampling_g.cu:198-201): inp_p (b,n) weights, inp_r (b,m) uniforms in [0,1), temp (b,n) floats
 * (receives the cumulative sums), out (b,m) int32 inverse-CDF indices. */
int rf_probsample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                  rf_stream_t stream);

/* ------------------------------------------------------- loss glue as a fused op (row f1) --- */
/* The reference's `earth_mover` (vv_recon.py:392-399) is approx_match -> match_cost, with
 * MatchCostGrad behind it (pc_distance/tf_approxmatch.py:44-50; ApproxMatch itself is NoGradient,
 * so match is a constant in the backward).  rf_earth_mover computes the same cost (b) -- and,
 * when grad1/grad2 are non-NULL (both or neither), the same MatchCostGrad outputs grad1 (b,n,3),
 * grad2 (b,m,3) -- straight from the per-level ratio vectors, without materialising the
 * (b,m,n) match tensor (512 MiB at 32x2048x2048; 1 GiB per sample at 16384^2).  No reference
 * launcher corresponds to it: a TF-side maintainer would register it as one new op replacing
 * the three-op chain.  Reference 10-level schedule only.  The cost is the chain's within rel 1e-5 (north_star's bar), not its
 * bits: the cost-only form (grad1 == grad2 == NULL) sums a sample's columns in the order of their last live level and takes
 * sqrt(d2) from v_sqrt_f32 (1 ulp) -- a function of the sample alone, so RF_EMD_SWEPT's batch independence holds. */
size_t rf_earth_mover_workspace_bytes(int b, int n, int m);
int rf_earth_mover(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost,
                   float *grad1, float *grad2, void *workspace, size_t workspace_bytes,
                   rf_stream_t stream);
/* ... with the route pinned as rf_approxmatch_mode pins it (RF_EMD_SWEPT: cost[i] bit-identical whatever the batch). */
size_t rf_earth_mover_mode_workspace_bytes(int b, int n, int m, int mode);
int rf_earth_mover_mode(int b, int n, int m, const float *xyz1, const float *xyz2, float *cost,
                        float *grad1, float *grad2, void *workspace, size_t workspace_bytes,
                        rf_stream_t stream, int mode);

/* `chamfer_big` / `fidelity_loss` (vv_recon.py:381-390) are reduce_mean(sqrt(dist)) over the
 * nn_distance outputs.  rf_chamfer_loss returns the per-sample means loss (b, 2):
 * loss[i][0] = mean_j sqrt(dist1[i][j]), loss[i][1] = mean_k sqrt(dist2[i][k]) (0 for a direction
 * that is not computed) -- the batch means of the reference are the means of these columns -- next
 * to the nn_distance outputs of the computed directions (a direction whose dist/idx pointers are
 * NULL is skipped: fidelity_loss needs direction 1 only).  sorted1 / sorted2: optional rf_nn_sort
 * handles of xyz1 / xyz2 (NULL: sorted internally when the culled sweep is used).
 * rf_chamfer_loss_grad is its backward: NnDistanceGrad with
 * grad_dist_d[i][j] = grad_loss[i][d] / npts_d * 0.5 / sqrt(dist_d[i][j]) formed inside the scatter
 * kernel (no intermediate tensors); grad_xyz1 (b,n,3) / grad_xyz2 (b,m,3) fully overwritten. */
size_t rf_chamfer_loss_workspace_bytes(int b, int n, int m, int want1, int want2, int have_sorted1,
                                       int have_sorted2);
int rf_chamfer_loss(int b, int n, int m, const float *xyz1, const float *xyz2, const void *sorted1,
                    const void *sorted2, float *loss, float *dist1, int *idx1, float *dist2, int *idx2,
                    void *workspace, size_t workspace_bytes, rf_stream_t stream);
int rf_chamfer_loss_grad(int b, int n, int m, const float *xyz1, const float *xyz2, const float *dist1,
                         const int *idx1, const float *dist2, const int *idx2, const float *grad_loss,
                         float *grad_xyz1, float *grad_xyz2, rf_stream_t stream);

/* `merge_layer` (vv_recon.py:132-139): idx2 of nn_distance(rawpts, newpts), the winner gathered
 * (group_point with nsample = 1), and every new point pulled towards it:
 *   refined = newpts + exp(-|g - newpts|^2 / (1e-8 + decfactor^2)) * (g - newpts).
 * rawpts (b,n,3), newpts (b,m,3), decfactor_dev: DEVICE pointer to the one-element variable
 * (vv_recon.py:211), sorted_raw: optional rf_nn_sort handle of rawpts; refined (b,m,3), idx2 (b,m)
 * (kept for the backward).  rf_merge_layer_grad: grad_newpts (b,m,3), grad_dec (b) per-sample
 * partial derivatives wrt decfactor (their sum is the variable's gradient), grad_raw (b,n,3) or
 * NULL (rawpts is the network input in the model). */
size_t rf_merge_layer_workspace_bytes(int b, int n, int m, int have_sorted_raw);
int rf_merge_layer(int b, int n, int m, const float *rawpts, const float *newpts, const void *sorted_raw,
                   const float *decfactor_dev, float *refined, int *idx2, void *workspace,
                   size_t workspace_bytes, rf_stream_t stream);
int rf_merge_layer_grad(int b, int n, int m, const float *rawpts, const float *newpts,
                        const float *decfactor_dev, const int *idx2, const float *grad_refined,
                        float *grad_newpts, float *grad_dec, float *grad_raw, rf_stream_t stream);

/* ------------------------------------------------ model graph helper (row f2) ------------ */
/* The elementwise tail of RFNet's per-point dense layers in one pass.  The reference's conv2d
 * (vv_recon.py:47-65: conv + bias_add + activation) is mostly applied to
 * tf.concat([per-point features, tf.tile(global code word)]) (:101,127,144,148,280,288,299,317,343);
 * computed without the concatenation that is
 *     out[i,j,:] = act( y[i,j,:] + sum_{k<kp} p[i,j,k] w[k,:] + r[i,:] )
 * y (b,n,c): the GEMM of the wide per-point inputs, or NULL; p (b,n,kp), w (kp,c): a narrow
 * per-point input (the coordinates: kp = 3; kp <= 16) applied on the fly, or kp = 0; r: bias + code
 * word term, (b,c) with r_per_sample != 0, (c) otherwise; act: 0 none, 1 relu, 2 tanh.
 * c % 4 == 0, c <= 1024 (rf_point_affine_supported).  out may alias y. */
/* tf.reduce_max(axis=1) of a (b, n, c) feature tensor -> (b, c) (global_mlp / encode_cell /
 * recover_cell / init_move_layer / refine_layer, vv_recon.py:90,107,129,151,286).  c % 4 == 0,
 * c <= 1024.  Two launches (strips, fold); exact and deterministic (max in any order).  NaNs are
 * ignored (fmaxf), which the graph never produces after a ReLU. */
size_t rf_maxpool_points_workspace_bytes(int b, int n, int c);
int rf_maxpool_points(int b, int n, int c, const float *x, float *out, void *workspace,
                      size_t workspace_bytes, rf_stream_t stream);

/* The same with the point index of each maximum (idx (b, c) int32, the lowest index among ties): the
 * forward of the pooling when a backward follows (training: tf.gradients of reduce_max routes the
 * gradient to the arg-max rows). */
size_t rf_maxpool_points_idx_workspace_bytes(int b, int n, int c);
int rf_maxpool_points_idx(int b, int n, int c, const float *x, float *out, int *idx, void *workspace,
                          size_t workspace_bytes, rf_stream_t stream);

/* Backward of a layer tail (training step of the graph; conv2d's bias_add + activation,
 * vv_recon.py:47-65, differentiated):  g[i,j,:] = grad[i,j,:] * act'(out[i,j,:])  and
 * sums[i,:] = sum_j g[i,j,:]  (the bias gradient is the sum of `sums` over i; the gradient of a
 * per-sample row r of rf_point_affine is `sums` itself) in ONE pass over grad/out.  act: 0 none, 1 relu
 * (out > 0), 2 tanh (1 - out^2), 3 leaky relu with slope 0.2 (sign of out); `out` is the layer's
 * OUTPUT (NULL with act 0).  g may alias grad; g NULL (or act 0 with g == grad) writes nothing.
 * c % 4 == 0, c <= 1024.  Deterministic (fixed summation order). */
size_t rf_act_grad_colsum_workspace_bytes(int b, int n, int c);
int rf_act_grad_colsum(int b, int n, int c, const float *grad, const float *out, int act, float *g,
                       float *sums, void *workspace, size_t workspace_bytes, rf_stream_t stream);

int rf_point_affine_supported(int c, int kp);
int rf_point_affine(int b, int n, int c, const float *y, const float *p, int kp, const float *w,
                    const float *r, int r_per_sample, int act, float *out, rf_stream_t stream);

/* ------------------------------------------------------------------ runtime diagnostic --- */
/* hipMemsetAsync(p, 0, bytes) on `stream` -- the ONE place this library issues a memset, and only for
 * this purpose: on ROCm 7 with the graph "packet capture" on (the default; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
 * before the runtime starts turns it off) a memset node of a small buffer captured into a HIP graph writes
 * garbage on later launches of the graph -- harmless for this library, whose zero fills are kernels, fatal
 * for anything else in the same graph that does (PyTorch's reduction kernels clear their semaphores that
 * way).  A host that captures graphs captures this call in a small test graph first and checks the buffer
 * after every replay (rfnet_amd/_host.py:graph_replay_ok). */
int rf_probe_memset_async(void *p, size_t bytes, rf_stream_t stream);

/* y[i] = v_exp_f32(x[i]) for i < count, the instruction every EMD weight exp(level * d2) goes through
 * (pc_distance/tf_approxmatch.cu:49,77,110,146 use __expf = ex2.approx(x * log2e)).  The skipping sweeps of the sharp
 * levels leave out pairs whose argument is <= -160 on the ground that the instruction returns EXACTLY +0 there; this entry
 * point lets a host (and tests/test_gpu_emd.py) check that on the device it runs on. */
int rf_probe_exp2(const float *x, float *y, int count, rf_stream_t stream);

/* ------------------------------------------------------------------ measurement hooks --- */
/* When enabled, every kernel launch made by this library is bracketed by hipEvents recorded
 * on the launch stream.  rf_profile_collect() waits for them and returns the per-kernel sums
 * since the last collect.  Used by bench.py for roofline.achieved; off by default.
 * This is the library's only process-global state: one switch and one list of pending event
 * pairs for the whole process, shared by all threads and streams (both calls are thread-safe;
 * a collect() concurrent with launches simply leaves those launches for the next collect). */
void rf_profile_enable(int on);
/* Fills up to `cap` entries; returns the number of distinct kernel names seen.  names[i] is a
 * pointer to a static string; ms[i] the summed duration in milliseconds; launches[i] the count. */
int rf_profile_collect(const char **names, double *ms, long *launches, int cap);

#ifdef __cplusplus
}
#endif
#endif /* RFOPS_H_ */
