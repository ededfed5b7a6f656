// sample_group.hip -- BASELINE.json configs[2] as ONE call: farthest_point_sample -> gather_point -> query_ball_point ->
// group_point on caller buffers (the chain of tf_ops/sampling/tf_sampling_g.cu:105-181 and tf_ops/grouping/tf_grouping_g.cu:3-57
// that a PointNet++-style set-abstraction level runs), one FFI crossing, graph-capturable.
//
// MI355X shape of the chain.  The four ops are four dependent launches in the reference; three of them are under 15 us at C3
// and pay more in launch gaps and host work than in kernel time.  Here:
//   * FPS writes the samples' coordinates itself (the winner's coordinates are in scalar registers every iteration): no
//     gather launch;
//   * the ball query runs over the sorted dataset's boxes and writes the grouped coordinates with the index rows (the row is
//     in LDS): no group launch for xyz;
//   * the dataset's sort serves BOTH ops where FPS runs over the sorted cloud (rfi::fps_sorted_pays); elsewhere it
//     does not depend on FPS: given a second stream it runs BESIDE FPS and costs the chain nothing.
// So the pass is two launches on the caller's stream and one on the auxiliary stream.  Results are those of the four
// separate ops, bit for bit (tests/test_gpu_sample_group.py).
#include "common.hpp"
#include "group_internal.hpp"
#include "nn_pruned.hpp"

namespace {
size_t align256(size_t v) { return (v + 255) / 256 * 256; }
}  // namespace

extern "C" {

size_t rf_sample_and_group_workspace_bytes(int b, int n) {
    if (b <= 0 || n <= 0) return 0;
    const size_t sorted = rf_queryballpoint_boxes_workspace_bytes(b, n);
    if (!sorted) return 0;
    return align256(sizeof(float) * rf_farthestpointsampling_temp_floats(b, n)) + sorted;
}

int rf_sample_and_group(int b, int n, int npoint, float radius, const float *radius_dev, int nsample, const float *xyz,
                        int *fps_idx, float *new_xyz, int *idx, int *pts_cnt, float *grouped_xyz, void *workspace,
                        size_t workspace_bytes, rf_stream_t stream, rf_stream_t aux_stream) {
    if (b < 0 || n < 0 || npoint < 0 || nsample <= 0) return RF_EINVAL;
    if ((long)b * npoint == 0) return RF_OK;
    if (n < 64 || nsample > 64 || b > 65535 || !rfp::pruned_supported(b, n, n)) return RF_EINVAL;
    if (!xyz || !fps_idx || !new_xyz || !idx || !pts_cnt || !grouped_xyz || !workspace || !rf::aligned16(workspace))
        return RF_EINVAL;
    if (workspace_bytes < rf_sample_and_group_workspace_bytes(b, n)) return RF_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream, aux = (hipStream_t)aux_stream;
    char *w = (char *)workspace;
    const size_t temp_floats = rf_farthestpointsampling_temp_floats(b, n);
    float *temp = temp_floats ? (float *)w : nullptr;
    w += align256(sizeof(float) * temp_floats);
    const rfp::Sorted so = rfp::sorted_view(b, n, w);
    const int nn[1] = {n};
    const float *src[1] = {xyz};
    if (rfi::fps_sorted_pays(n, npoint)) {
        // FPS itself runs over the sorted cloud (fps_sorted_kernel: a third shorter iterations): one sort serves both ops, on
        // the caller's stream -- nothing is left to run beside FPS
        if (int e = rfp::sort_sets(b, 1, nn, src, &so, s, nullptr)) return e;
        if (int e = rfi::fps_sorted(b, n, npoint, xyz, so, fps_idx, new_xyz, s)) return e;
        return rfi::ball_boxes(b, n, npoint, radius, radius_dev, nsample, xyz, new_xyz, so, idx, pts_cnt, grouped_xyz, 1, s);
    }
    // The sort beside FPS on the auxiliary stream: aux waits for the caller's stream (the inputs are ready there), the caller's
    // stream waits for the sort before the ball query.  The events live for this call only and are destroyed on EVERY exit (an
    // event still pending when it is destroyed is released by the runtime once it has completed); once the sort has been
    // queued on aux, the caller's stream joins it on every exit too -- an error return never leaves aux running ahead unjoined.
    struct Fork {
        hipEvent_t fork = nullptr, join = nullptr;
        hipStream_t s = nullptr, aux = nullptr;
        bool queued = false;  // work sits on aux that s has not joined yet
        int join_now() {
            if (!queued) return RF_OK;
            queued = false;
            RF_HIP(hipEventRecord(join, aux));
            RF_HIP(hipStreamWaitEvent(s, join, 0));
            return RF_OK;
        }
        ~Fork() {
            (void)join_now();
            if (fork) (void)hipEventDestroy(fork);
            if (join) (void)hipEventDestroy(join);
        }
    } fk;
    if (aux && aux != s) {
        fk.s = s, fk.aux = aux;
        RF_HIP(hipEventCreateWithFlags(&fk.fork, hipEventDisableTiming));
        RF_HIP(hipEventCreateWithFlags(&fk.join, hipEventDisableTiming));
        RF_HIP(hipEventRecord(fk.fork, s));
        RF_HIP(hipStreamWaitEvent(aux, fk.fork, 0));
        fk.queued = true;
        if (int e = rfp::sort_sets(b, 1, nn, src, &so, aux, nullptr)) return e;
        RF_HIP(hipEventRecord(fk.join, aux));  // (recorded NOW, behind the sort: FPS below does not wait for it)
    } else {
        if (int e = rfp::sort_sets(b, 1, nn, src, &so, s, nullptr)) return e;
    }
    if (int e = rfi::fps(b, n, npoint, xyz, temp, fps_idx, new_xyz, s)) return e;
    if (fk.queued) {
        fk.queued = false;
        RF_HIP(hipStreamWaitEvent(s, fk.join, 0));
    }
    return rfi::ball_boxes(b, n, npoint, radius, radius_dev, nsample, xyz, new_xyz, so, idx, pts_cnt, grouped_xyz, 1, s);
}

}  // extern "C"
