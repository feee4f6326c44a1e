/* CPU oracle -- plain-C restatement of the reference's compiled (numba-jit) loops.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the HIP
 * path at sizes where the pure-Python restatement (oracle/ref_numpy.py) is too
 * slow, and the `cpu_baseline` leg of bench.py.  Never linked into or called by
 * the product library.  Each function cites the reference file:line it follows
 * (paths relative to /root/reference).  Pinned against the reference's own
 * outputs through tests/golden/ref_*.npz (tests/test_oracle_golden.py).
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC; no -ffast-math: the cell
 * arithmetic must stay IEEE float64).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* load_data.py:593-641 (_points_to_voxel_reverse_kernel) + :695-771.
 * points [n,F] f32; range[6], vsize[3] f64; outputs sized for max_voxels.
 * coors rows are (z,y,x).  cell_to_pillar is caller-provided scratch of
 * nz*ny*nx int32.  Returns the number of pillars. */
int oracle_points_to_voxel(const float* points, int n, int F, const double* range,
                           const double* vsize, int max_points, int max_voxels,
                           float* voxels, int32_t* coors, int32_t* num_points,
                           int32_t* cell_to_pillar) {
    int grid[3];
    for (int j = 0; j < 3; ++j) {
        double g = (range[3 + j] - range[j]) / vsize[j];
        grid[j] = (int)nearbyint(g); /* np.round: half-to-even under the default rounding mode */
    }
    const int nx = grid[0], ny = grid[1], nz = grid[2];
    const long ncell = (long)nx * ny * nz;
    for (long c = 0; c < ncell; ++c) cell_to_pillar[c] = -1;
    memset(num_points, 0, sizeof(int32_t) * (size_t)max_voxels);
    memset(voxels, 0, sizeof(float) * (size_t)max_voxels * max_points * F);
    memset(coors, 0, sizeof(int32_t) * (size_t)max_voxels * 3);
    int n_pillars = 0;
    for (int i = 0; i < n; ++i) {
        int c3[3];
        int ok = 1;
        for (int j = 0; j < 3; ++j) {
            double c = floor(((double)points[(long)i * F + j] - range[j]) / vsize[j]);
            if (c < 0 || c >= grid[j]) { ok = 0; break; }
            c3[j] = (int)c;
        }
        if (!ok) continue;
        long lin = ((long)c3[2] * ny + c3[1]) * nx + c3[0];
        int pid = cell_to_pillar[lin];
        if (pid == -1) {
            pid = n_pillars;
            if (n_pillars >= max_voxels) break;
            n_pillars++;
            cell_to_pillar[lin] = pid;
            coors[pid * 3 + 0] = c3[2];
            coors[pid * 3 + 1] = c3[1];
            coors[pid * 3 + 2] = c3[0];
        }
        int k = num_points[pid];
        if (k < max_points) {
            memcpy(voxels + ((long)pid * max_points + k) * F, points + (long)i * F, sizeof(float) * F);
            num_points[pid] = k + 1;
        }
    }
    return n_pillars;
}

/* load_data.py:586-591 + :3054-3055 + :558-584 + :3070.
 * coors [P,3] (z,y,x); cells [A,4] = clamped (x0,y0,x1,y1) per anchor (computed
 * on the host exactly as fused_get_anchors_area does).  dense is scratch
 * [ny*nx] f32.  mask[a] = area > threshold. */
void oracle_anchor_mask(const int32_t* coors, int P, int ny, int nx, const int32_t* cells, int A,
                        float threshold, float* dense, uint8_t* mask) {
    memset(dense, 0, sizeof(float) * (size_t)ny * nx);
    for (int p = 0; p < P; ++p) dense[(long)coors[p * 3 + 1] * nx + coors[p * 3 + 2]] += 1.0f;
    for (int y = 1; y < ny; ++y)
        for (int x = 0; x < nx; ++x) dense[(long)y * nx + x] += dense[(long)(y - 1) * nx + x];
    for (int y = 0; y < ny; ++y)
        for (int x = 1; x < nx; ++x) dense[(long)y * nx + x] += dense[(long)y * nx + x - 1];
    for (int a = 0; a < A; ++a) {
        int x0 = cells[a * 4 + 0], y0 = cells[a * 4 + 1], x1 = cells[a * 4 + 2], y1 = cells[a * 4 + 3];
        float area = dense[(long)y1 * nx + x1] - dense[(long)y1 * nx + x0] - dense[(long)y0 * nx + x1] +
                     dense[(long)y0 * nx + x0];
        mask[a] = area > threshold;
    }
}

/* libraries/eval_helper_functions.py:553-564 (iou_device): f32 differences,
 * then float64 (numba types f32 + int64 literal as f64). */
static double oracle_iou(const float* a, const float* b) {
    float left = a[0] > b[0] ? a[0] : b[0];
    float right = a[2] < b[2] ? a[2] : b[2];
    float top = a[1] > b[1] ? a[1] : b[1];
    float bottom = a[3] < b[3] ? a[3] : b[3];
    double w = (double)(float)(right - left) + 1.0;
    double h = (double)(float)(bottom - top) + 1.0;
    if (w < 0.0) w = 0.0;
    if (h < 0.0) h = 0.0;
    double inter = w * h;
    double sa = ((double)(float)(a[2] - a[0]) + 1.0) * ((double)(float)(a[3] - a[1]) + 1.0);
    double sb = ((double)(float)(b[2] - b[0]) + 1.0) * ((double)(float)(b[3] - b[1]) + 1.0);
    return inter / (sa + sb - inter);
}

/* libraries/eval_helper_functions.py:567-598 (nms_kernel) + :529-546
 * (nms_postprocess).  boxes [n,5] f32 sorted by score descending.  keep gets
 * the kept positions (indices into the sorted order); returns their count. */
int oracle_nms_sorted(const float* boxes, int n, float thresh, int32_t* keep) {
    int cb = (n + 63) / 64;
    uint64_t* mask = (uint64_t*)calloc((size_t)n * cb + 1, sizeof(uint64_t));
    uint64_t* remv = (uint64_t*)calloc((size_t)cb + 1, sizeof(uint64_t));
    double thr = (double)thresh;
    for (int i = 0; i < n; ++i)
        for (int blk = 0; blk < cb; ++blk) {
            int lo = blk * 64, hi = n < lo + 64 ? n : lo + 64;
            int start = (blk == i / 64) ? (i % 64) + 1 : 0;
            uint64_t t = 0;
            for (int k = start; k < hi - lo; ++k)
                if (oracle_iou(boxes + (long)i * 5, boxes + (long)(lo + k) * 5) > thr) t |= 1ull << k;
            mask[(long)i * cb + blk] = t;
        }
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        int blk = i / 64, bit = i % 64;
        if (!((remv[blk] >> bit) & 1ull)) {
            keep[nk++] = i;
            for (int j = blk; j < cb; ++j) remv[j] |= mask[(long)i * cb + j];
        }
    }
    free(mask);
    free(remv);
    return nk;
}
