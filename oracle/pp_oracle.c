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

/* ------------------------------------------------------------------------------------------
 * Rotated-box IoU of the KITTI-style evaluator.
 * Follows second/core/non_max_suppression/nms_gpu.py: rbbox_to_corners :361-385,
 * point_in_quadrilateral :316-333, line_segment_intersection :240-275,
 * quadrilateral_intersection :336-358, sort_vertex_in_convex_polygon :197-234, area :187-194,
 * devRotateIoUEval :564-576, rotate_iou_kernel_eval :579-615 (note the kernel passes the QUERY
 * box first).  All arithmetic is float32, as the float32 arrays of the reference make it;
 * cos / sin / sqrt are evaluated in double and rounded (math.cos on a float32 scalar).
 * ------------------------------------------------------------------------------------------ */
static void riou_corners(const float* r, float* c) {
    const float a_cos = (float)cos((double)r[4]), a_sin = (float)sin((double)r[4]);
    const float cx = r[0], cy = r[1], xd = r[2], yd = r[3];
    const float px[4] = {-xd / 2, -xd / 2, xd / 2, xd / 2};
    const float py[4] = {-yd / 2, yd / 2, yd / 2, -yd / 2};
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * px[i] + a_sin * py[i] + cx;
        c[2 * i + 1] = -a_sin * px[i] + a_cos * py[i] + cy;
    }
}

static int riou_point_in_quad(float x, float y, const float* c) {
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1];
    const float ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = x - c[0], ap1 = y - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}

static int riou_seg_intersection(const float* p1, const float* p2, int i, int j, float* t) {
    const float A0 = p1[2 * i], A1 = p1[2 * i + 1];
    const float B0 = p1[2 * ((i + 1) % 4)], B1 = p1[2 * ((i + 1) % 4) + 1];
    const float C0 = p2[2 * j], C1 = p2[2 * j + 1];
    const float D0 = p2[2 * ((j + 1) % 4)], D1 = p2[2 * ((j + 1) % 4) + 1];
    const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
    const int acd = DA1 * CA0 > CA1 * DA0;
    const int bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
    if (acd != bcd) {
        const int abc = CA1 * BA0 > BA1 * CA0;
        const int abd = DA1 * BA0 > BA1 * DA0;
        if (abc != abd) {
            const float DC0 = D0 - C0, DC1 = D1 - C1;
            const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
            const float DH = BA1 * DC0 - BA0 * DC1;
            const float Dx = ABBA * DC0 - BA0 * CDDC, Dy = ABBA * DC1 - BA1 * CDDC;
            t[0] = Dx / DH;
            t[1] = Dy / DH;
            return 1;
        }
    }
    return 0;
}

static float riou_inter(const float* r1, const float* r2) {
    float c1[8], c2[8], ip[16 + 64], t[2];   /* 16 used by the reference; slack keeps a >8-point case in bounds */
    int n = 0;
    riou_corners(r1, c1);
    riou_corners(r2, c2);
    for (int i = 0; i < 4; ++i) {
        if (riou_point_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { ip[2 * n] = c1[2 * i]; ip[2 * n + 1] = c1[2 * i + 1]; ++n; }
        if (riou_point_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { ip[2 * n] = c2[2 * i]; ip[2 * n + 1] = c2[2 * i + 1]; ++n; }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            if (riou_seg_intersection(c1, c2, i, j, t)) { ip[2 * n] = t[0]; ip[2 * n + 1] = t[1]; ++n; }
    if (n > 0) {   /* sort_vertex_in_convex_polygon */
        float cx = 0.f, cy = 0.f, vs[16 + 32];
        for (int i = 0; i < n; ++i) { cx += ip[2 * i]; cy += ip[2 * i + 1]; }
        cx /= (float)n;
        cy /= (float)n;
        for (int i = 0; i < n; ++i) {
            float v0 = ip[2 * i] - cx, v1 = ip[2 * i + 1] - cy;
            const float d = (float)sqrt((double)(v0 * v0 + v1 * v1));
            v0 = v0 / d;
            v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i] = v0;
        }
        for (int i = 1; i < n; ++i) {
            if (vs[i - 1] > vs[i]) {
                const float temp = vs[i], tx = ip[2 * i], ty = ip[2 * i + 1];
                int j = i;
                while (j > 0 && vs[j - 1] > temp) {
                    vs[j] = vs[j - 1];
                    ip[2 * j] = ip[2 * j - 2];
                    ip[2 * j + 1] = ip[2 * j - 1];
                    --j;
                }
                vs[j] = temp;
                ip[2 * j] = tx;
                ip[2 * j + 1] = ty;
            }
        }
    }
    float area = 0.f;
    for (int i = 0; i < n - 2; ++i) {
        const float* a = ip;
        const float* b = ip + 2 * i + 2;
        const float* c = ip + 2 * i + 4;
        area += fabsf(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / 2.0f);
    }
    return area;
}

/* out[n*K + k] = devRotateIoUEval(query[k], boxes[n], criterion)   (nms_gpu.py:611-613) */
void oracle_rotate_iou_eval(const float* boxes, int64_t N, const float* qboxes, int64_t K, int criterion,
                            float* out) {
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) {
            const float* r1 = qboxes + 5 * k;
            const float* r2 = boxes + 5 * n;
            const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
            const float ai = riou_inter(r1, r2);
            float v;
            if (criterion == -1) v = ai / (a1 + a2 - ai);
            else if (criterion == 0) v = ai / a1;
            else if (criterion == 1) v = ai / a2;
            else v = ai;
            out[n * K + k] = v;
        }
}
