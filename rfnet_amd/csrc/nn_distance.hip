// nn_distance.hip -- Chamfer nearest-neighbour distance, forward and backward, for gfx950.
//
// Replaces NmDistanceKernel / NmDistanceGradKernel (tf_ops/CD/tf_nndistance_g.cu:4-156).
// Results: bit-identical to oracle/rfops_oracle.c (d2 = fma(dz,dz,fma(dx,dx,dy*dy)),
// differences "other - own", lowest index wins ties).
//
// MI355X design (not the reference's 32x16 blocks of one-thread-per-point):
//   * the sweep is fp32-VALU bound; every instruction in the pair loop
//     counts.  The reference spends 9 VALU ops per pair (3 sub, mul, 2 fma, cmp, 2 select).
//     Here the argmin bookkeeping is taken out of the pair loop: per own point only the
//     running MIN VALUE over a chunk of CH candidates is kept (v_min3_f32: half an op per
//     pair), one compare+select per CHUNK remembers which chunk lowered the minimum, and
//     the winning chunk (CH candidates) is re-scanned once at the end for the first index
//     whose distance equals the minimum bit-for-bit.  6.5 + 3/CH ops per pair.
//   * each thread owns R query points in registers; candidates are staged as float4 in LDS
//     and read with one broadcast ds_read_b128 per candidate per wave (R pairs per read).
//   * both directions run in ONE launch (no tail between the two sweeps) and the candidate
//     range is split over workgroups so that >= 4 workgroups per CU exist even for
//     B=32 x 2048 queries; per-split partial (min, argmin) go to the workspace and a small
//     merge kernel combines them in split order with strict '<' (deterministic, no atomics).
#include "common.hpp"

namespace {

constexpr int TPB = 256;    // threads per workgroup (4 waves: one per SIMD)
constexpr int R = 4;        // own (query) points per thread
constexpr int CH = 16;      // candidates per chunk (argmin granularity in the sweep)
constexpr int TILE = 1024;  // candidates per LDS tile: 16 KiB as float4

struct Dir {
    const float *own;    // (b, nq, 3)
    const float *other;  // (b, nc, 3)
    float *out_dist;     // partial or final (see nsplit)
    int *out_idx;
    int nq, nc;
    int qblocks;  // ceil(nq / (TPB*R))
    int nsplit;   // candidate splits
    int span;     // candidates per split, multiple of CH
};

struct Args {
    Dir d[2];
    int b;
    int nblk0;  // workgroups of direction 0
};

__global__ __launch_bounds__(TPB) void nn_sweep_kernel(Args a) {
    __shared__ float4 tile[TILE];
    int bid = blockIdx.x;
    const int which = bid >= a.nblk0;
    if (which) bid -= a.nblk0;
    const Dir &D = a.d[which];
    const int split = bid % D.nsplit;
    const int qb = (bid / D.nsplit) % D.qblocks;
    const int bi = bid / (D.nsplit * D.qblocks);
    const int tid = threadIdx.x;

    const float *own = D.own + (size_t)bi * D.nq * 3;
    const float *oth = D.other + (size_t)bi * D.nc * 3;

    float ax[R], ay[R], az[R], best[R];
    int bchunk[R];
    const int c_begin = split * D.span;
    const int c_end = min(D.nc, c_begin + D.span);
#pragma unroll
    for (int r = 0; r < R; r++) {
        int j = (qb * R + r) * TPB + tid;
        int jj = min(j, D.nq - 1);
        ax[r] = own[jj * 3 + 0];
        ay[r] = own[jj * 3 + 1];
        az[r] = own[jj * 3 + 2];
        best[r] = INFINITY;
        bchunk[r] = c_begin / CH;
    }

    for (int t0 = c_begin; t0 < c_end; t0 += TILE) {
        const int tcount = min(TILE, c_end - t0);
        const int tpad = (tcount + CH - 1) / CH * CH;
        __syncthreads();
        for (int k = tid; k < tpad; k += TPB) {
            float4 v;
            if (k < tcount) {
                const float *p = oth + (size_t)(t0 + k) * 3;
                v = make_float4(p[0], p[1], p[2], 0.f);
            } else {
                v = make_float4(INFINITY, 0.f, 0.f, 0.f);  // d2 = +inf: never the minimum
            }
            tile[k] = v;
        }
        __syncthreads();
        for (int c = 0; c < tpad; c += CH) {
            float cm[R];
#pragma unroll
            for (int r = 0; r < R; r++) cm[r] = INFINITY;
#pragma unroll
            for (int u = 0; u < CH; u += 2) {
                const float4 p = tile[c + u];
                const float4 q = tile[c + u + 1];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    float d0 = rf::d2_fma(p.x - ax[r], p.y - ay[r], p.z - az[r]);
                    float d1 = rf::d2_fma(q.x - ax[r], q.y - ay[r], q.z - az[r]);
                    cm[r] = fminf(fminf(cm[r], d0), d1);
                }
            }
            const int chunk = (t0 + c) / CH;
#pragma unroll
            for (int r = 0; r < R; r++) {
                if (cm[r] < best[r]) {
                    best[r] = cm[r];
                    bchunk[r] = chunk;
                }
            }
        }
    }

    // Resolve the argmin: first index inside the winning chunk whose d2 equals the minimum.
    // (same instruction sequence => same bits.)  If every distance was +inf the chunk is the
    // first one and index c_begin is returned, like the reference's unconditional k==0.
    const size_t obase = ((size_t)split * a.b + bi) * D.nq;
#pragma unroll
    for (int r = 0; r < R; r++) {
        int j = (qb * R + r) * TPB + tid;
        if (j >= D.nq) continue;
        int k0 = bchunk[r] * CH;
        int besti = k0;
#pragma unroll
        for (int u = CH - 1; u >= 0; u--) {
            int k = k0 + u;
            if (k < c_end) {
                const float *p = oth + (size_t)k * 3;
                float d = rf::d2_fma(p[0] - ax[r], p[1] - ay[r], p[2] - az[r]);
                if (d == best[r]) besti = k;
            }
        }
        float bd = best[r];
        D.out_dist[obase + j] = bd;
        D.out_idx[obase + j] = besti;
    }
}

// Combine the per-split partials in split order; strict '<' keeps the lowest index.
__global__ void nn_merge_kernel(const float *pd, const int *pi, float *dist, int *idx, int nsplit,
                                long total) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    float best = pd[g];
    int besti = pi[g];
    for (int s = 1; s < nsplit; s++) {
        float d = pd[(size_t)s * total + g];
        if (d < best) {
            best = d;
            besti = pi[(size_t)s * total + g];
        }
    }
    dist[g] = best;
    idx[g] = besti;
}

// Backward: one thread per (direction, batch, point).  g = gd+gd; v = (a-b)*g rounded alone;
// own-side contributions are unique per point (plain adds after the zero fill would also do,
// but the other direction scatters into the same array, so both use atomics like the reference,
// tf_nndistance_g.cu:142-147).
__global__ void nn_grad_kernel(int b, int n, int m, const float *xyz1, const float *xyz2,
                               const float *gd1, const int *idx1, const float *gd2,
                               const int *idx2, float *g1, float *g2) {
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long t1 = (long)b * n, t2 = (long)b * m;
    const float *A, *B, *gd;
    const int *ix;
    float *GA, *GB;
    long j;
    int na, nb;
    if (g < t1) {
        A = xyz1; B = xyz2; gd = gd1; ix = idx1; GA = g1; GB = g2; j = g; na = n; nb = m;
    } else if (g < t1 + t2) {
        A = xyz2; B = xyz1; gd = gd2; ix = idx2; GA = g2; GB = g1; j = g - t1; na = m; nb = n;
    } else {
        return;
    }
    long bi = j / na;
    int k = ix[j];
    const float *pa = A + j * 3;
    const float *pb = B + (bi * nb + k) * 3;
    float *ga = GA + j * 3;
    float *gb = GB + (bi * nb + k) * 3;
    float gg = gd[j] + gd[j];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = (pa[c] - pb[c]) * gg;
        atomicAdd(ga + c, v);
        atomicAdd(gb + c, -v);
    }
}

struct Plan {
    int qblocks[2], nsplit[2], span[2];
};

Plan make_plan(int b, int n, int m) {
    Plan p;
    const int nq[2] = {n, m}, nc[2] = {m, n};
    for (int d = 0; d < 2; d++) {
        p.qblocks[d] = rf::ceil_div(nq[d], TPB * R);
        long base = (long)b * p.qblocks[d];
        // aim for >= 1024 workgroups per direction (4 per CU), spans of at least 256 candidates
        int want = (int)((1024 + base - 1) / (base > 0 ? base : 1));
        int maxs = nc[d] / 256 > 0 ? nc[d] / 256 : 1;
        int s = want < 1 ? 1 : (want > maxs ? maxs : want);
        int span = rf::ceil_div(rf::ceil_div(nc[d], s), CH) * CH;
        if (span < CH) span = CH;
        p.span[d] = span;
        p.nsplit[d] = nc[d] > 0 ? rf::ceil_div(nc[d], span) : 1;
    }
    return p;
}

}  // namespace

extern "C" {

size_t rf_nn_distance_workspace_bytes(int b, int n, int m) {
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    Plan p = make_plan(b, n, m);
    size_t e0 = p.nsplit[0] > 1 ? (size_t)p.nsplit[0] * b * n : 0;
    size_t e1 = p.nsplit[1] > 1 ? (size_t)p.nsplit[1] * b * m : 0;
    return (e0 + e1) * 8;
}

int rf_nn_distance(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                   int *idx1, float *dist2, int *idx2, void *workspace, size_t workspace_bytes,
                   rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    if (b == 0 || (n == 0 && m == 0)) return RF_OK;
    if (n == 0 || m == 0) return RF_EINVAL;  // a nearest neighbour in an empty set is undefined
    if (!xyz1 || !xyz2 || !dist1 || !idx1 || !dist2 || !idx2) return RF_EINVAL;
    if (workspace_bytes < rf_nn_distance_workspace_bytes(b, n, m)) return RF_EWORKSPACE;
    if (workspace_bytes && !workspace) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    Plan p = make_plan(b, n, m);

    Args a;
    a.b = b;
    size_t e0 = p.nsplit[0] > 1 ? (size_t)p.nsplit[0] * b * n : 0;
    size_t e1 = p.nsplit[1] > 1 ? (size_t)p.nsplit[1] * b * m : 0;
    float *w = (float *)workspace;
    float *pd0 = w, *pd1 = w + 2 * e0;
    int *pi0 = (int *)(w + e0), *pi1 = (int *)(w + 2 * e0 + e1);

    a.d[0] = Dir{xyz1, xyz2, e0 ? pd0 : dist1, e0 ? pi0 : idx1, n, m, p.qblocks[0], p.nsplit[0], p.span[0]};
    a.d[1] = Dir{xyz2, xyz1, e1 ? pd1 : dist2, e1 ? pi1 : idx2, m, n, p.qblocks[1], p.nsplit[1], p.span[1]};
    a.nblk0 = b * p.qblocks[0] * p.nsplit[0];
    int nblk1 = b * p.qblocks[1] * p.nsplit[1];
    RF_LAUNCH("nn_sweep", nn_sweep_kernel, dim3(a.nblk0 + nblk1), dim3(TPB), 0, s, a);
    if (e0) {
        long tot = (long)b * n;
        RF_LAUNCH("nn_merge", nn_merge_kernel, dim3(rf::ceil_div(tot, 256)), dim3(256), 0, s, pd0, pi0,
                  dist1, idx1, p.nsplit[0], tot);
    }
    if (e1) {
        long tot = (long)b * m;
        RF_LAUNCH("nn_merge", nn_merge_kernel, dim3(rf::ceil_div(tot, 256)), dim3(256), 0, s, pd1, pi1,
                  dist2, idx2, p.nsplit[1], tot);
    }
    return RF_OK;
}

int rf_nn_distance_grad(int b, int n, int m, const float *xyz1, const float *xyz2,
                        const float *grad_dist1, const int *idx1, const float *grad_dist2,
                        const int *idx2, float *grad_xyz1, float *grad_xyz2, rf_stream_t stream) {
    if (b < 0 || n < 0 || m < 0) return RF_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((size_t)b * n) RF_HIP(hipMemsetAsync(grad_xyz1, 0, sizeof(float) * 3 * (size_t)b * n, s));
    if ((size_t)b * m) RF_HIP(hipMemsetAsync(grad_xyz2, 0, sizeof(float) * 3 * (size_t)b * m, s));
    long total = (long)b * n + (long)b * m;
    if (total == 0 || n == 0 || m == 0) return RF_OK;
    RF_LAUNCH("nn_grad", nn_grad_kernel, dim3(rf::ceil_div(total, 256)), dim3(256), 0, s, b, n, m, xyz1,
              xyz2, grad_dist1, idx1, grad_dist2, idx2, grad_xyz1, grad_xyz2);
    return RF_OK;
}

}  // extern "C"
