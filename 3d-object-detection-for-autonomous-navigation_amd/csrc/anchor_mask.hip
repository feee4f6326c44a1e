// Per-frame anchor mask: pillar occupancy -> 2-D inclusive prefix sum -> 4 lookups per anchor.
//
// Replaces, per frame (reference load_data.py:3043-3072):
//   sparse_sum_for_anchors_mask (:586-591)  occupancy[y][x] += 1 per pillar (both z-cells count)
//   .cumsum(0).cumsum(1)        (:3054-3055)
//   fused_get_anchors_area      (:558-584)  area = D - B - C + A at the anchor's clamped cells
//   anchors_mask = area > anchor_area_threshold (:3070)
// The anchor -> cell mapping (rbbox2d_to_near_bbox + float64 floor + clamp) is
// static and precomputed on the host (anchors.py); the occupancy comes for free
// from the voxeliser's cell -> pillar map.  Counts are kept in int32 (the
// reference's float32 counts are exact integers).  HBM/L2-bound integer work:
// 4*nz bytes read + 4 bytes written per cell, 16 B + 4 gathers + 1 B per anchor.
#include <stdlib.h>

#include "pp_common.h"
#include "anchor_mask_dev.h"

__global__ __launch_bounds__(256) void k_occ_rowscan(const int* __restrict__ cellmap, int rows, int nz, int ny,
                                                     int nx, int* __restrict__ integ) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / ny, y = row - b * ny;
    const size_t plane = (size_t)ny * nx;
    const int* map = cellmap + (size_t)b * nz * plane + (size_t)y * nx;
    int* out = integ + (size_t)b * plane + (size_t)y * nx;
    int carry = 0;
    for (int x0 = 0; x0 < nx; x0 += 64) {
        const int x = x0 + lane;
        int v = 0;
        if (x < nx)
            for (int z = 0; z < nz; ++z) v += (map[(size_t)z * plane + x] >= 0) ? 1 : 0;
        const int incl = wave_inclusive_scan(v);
        if (x < nx) out[x] = carry + incl;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
}

__global__ __launch_bounds__(256) void k_colscan(int* __restrict__ integ, int batch, int ny, int nx) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= batch * nx) return;
    const int b = t / nx, x = t - b * nx;
    int* col = integ + (size_t)b * ny * nx + x;
    int run = 0;
    int y = 0;
    for (; y + 8 <= ny; y += 8) {
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = col[(size_t)(y + q) * nx];
#pragma unroll
        for (int q = 0; q < 8; ++q) { run += v[q]; col[(size_t)(y + q) * nx] = run; }
    }
    for (; y < ny; ++y) { run += col[(size_t)y * nx]; col[(size_t)y * nx] = run; }
}

__global__ __launch_bounds__(256) void k_anchor_lookup(const int* __restrict__ integ,
                                                       const int* __restrict__ cells, int64_t A, int ny, int nx,
                                                       float threshold, uint8_t* __restrict__ mask) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const int b = blockIdx.y;
    const int4 c = reinterpret_cast<const int4*>(cells)[a];  // x0 y0 x1 y1
    const int* I = integ + (size_t)b * ny * nx;
    const int area = I[(size_t)c.w * nx + c.z] - I[(size_t)c.w * nx + c.x] - I[(size_t)c.y * nx + c.z] +
                     I[(size_t)c.y * nx + c.x];
    mask[(size_t)b * A + a] = ((float)area > threshold) ? 1 : 0;
}

// The same three steps for one frame in ONE workgroup when the BEV grid fits in LDS (the shipped 80 x 64 grid: 20 KB):
// occupancy -> LDS, row scan (a wave per row), column scan (a thread per column), the anchors' four lookups out of LDS.
// One launch instead of three: on a single frame (the reference's production mode) the three kernels were 12 us of
// dependent launches for a few microseconds of work.
__global__ __launch_bounds__(1024) void k_anchor_mask_frame(const int* __restrict__ cellmap, int nz, int ny, int nx,
                                                            const int* __restrict__ cells, int64_t A, float threshold,
                                                            uint8_t* __restrict__ mask) {
    __shared__ int sI[AM_MAX_CELLS];
    const int b = blockIdx.x;
    anchor_mask_frame_block<1024>(cellmap + (size_t)b * nz * ny * nx, nz, ny, nx, cells, 0, A, threshold, mask + (size_t)b * A, sI);
}

// The same mask straight from the occupancy bitmap (grids with one z-cell: a set bit IS the count the reference adds,
// load_data.py:586-591): area = pillars with y0 < y <= y1 and x0 < x <= x1 = popcounts of the rows' bit ranges.  One
// launch instead of row scan + column scan + lookup (51 us at the KITTI-shaped B = 32), and no integral image.
__global__ __launch_bounds__(256) void k_anchor_lookup_bits(const unsigned long long* __restrict__ occbits,
                                                            const int* __restrict__ cells, int64_t A, int ny, int w64,
                                                            float threshold, uint8_t* __restrict__ mask) {
    const int64_t a = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const int b = blockIdx.y;
    const int4 c = reinterpret_cast<const int4*>(cells)[a];  // x0 y0 x1 y1 (clamped to the grid on the host)
    const unsigned long long* rows = occbits + (size_t)b * ny * w64;
    // columns x0 + 1 .. x1 live in bits lo = x0 + 2 .. hi = x1 + 1
    const int lo = c.x + 2, hi = c.z + 1;
    int area = 0;
    if (hi >= lo) {
        const int w0 = lo >> 6, w1 = hi >> 6;
        for (int y = c.y + 1; y <= c.w; ++y) {
            const unsigned long long* r = rows + (size_t)y * w64;
            for (int w = w0; w <= w1; ++w) {
                unsigned long long m = r[w];
                if (w == w0) m &= ~0ull << (lo & 63);
                if (w == w1) m &= ~0ull >> (63 - (hi & 63));
                area += __popcll(m);
            }
        }
    }
    mask[(size_t)b * A + a] = ((float)area > threshold) ? 1 : 0;
}

void launch_anchor_mask_bits(const unsigned long long* occbits, int batch, int ny, int nx, const int* cells, int64_t A,
                             float threshold, uint8_t* mask, hipStream_t s) {
    if (batch <= 0) return;
    PP_LAUNCH("k_anchor_lookup_bits", k_anchor_lookup_bits, dim3((unsigned)((A + 255) / 256), batch), dim3(256), 0, s, occbits,
              cells, A, ny, occ_words(nx), threshold, mask);
}

void launch_anchor_mask(const int* cellmap, int batch, int nz, int ny, int nx, const int* cells, int64_t A,
                        float threshold, int* integ, uint8_t* mask, hipStream_t s) {
    if (batch <= 0) return;
    static int fused = -1;      // PP_ANCHOR_MASK_FUSED=0: the three-kernel path everywhere (A/B measurements)
    if (fused < 0) { const char* e = getenv("PP_ANCHOR_MASK_FUSED"); fused = (e && e[0] == '0') ? 0 : 1; }
    // one workgroup per frame whenever the grid fits the LDS image; larger grids keep the three chip-wide kernels
    static int maxb = -1;       // PP_ANCHOR_MASK_FUSED_MAXB: largest batch of the one-workgroup-per-frame kernel
    if (maxb < 0) { const char* e = getenv("PP_ANCHOR_MASK_FUSED_MAXB"); maxb = e ? atoi(e) : (1 << 30); }   // (8 until round 3: at B=64 one launch of 64 workgroups instead of three chip-wide ones is 0-1 % more frames/s)
    if (fused && ny * (nx | 1) <= AM_MAX_CELLS && batch <= maxb) {
        PP_LAUNCH("k_anchor_mask_frame", k_anchor_mask_frame, dim3(batch), dim3(1024), 0, s, cellmap, nz, ny, nx, cells, A,
                  threshold, mask);
        return;
    }
    const int rows = batch * ny;
    PP_LAUNCH("k_occ_rowscan", k_occ_rowscan, dim3((rows + 3) / 4), dim3(256), 0, s, cellmap, rows, nz, ny, nx, integ);
    PP_LAUNCH("k_colscan", k_colscan, dim3((batch * nx + 255) / 256), dim3(256), 0, s, integ, batch, ny, nx);
    PP_LAUNCH("k_anchor_lookup", k_anchor_lookup, dim3((unsigned)((A + 255) / 256), batch), dim3(256), 0, s, integ, cells, A,
                       ny, nx, threshold, mask);
}
