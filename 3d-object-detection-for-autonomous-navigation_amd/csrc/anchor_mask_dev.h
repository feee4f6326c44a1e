// One frame's anchor mask inside ONE workgroup (device code shared by k_anchor_mask_frame and the extra workgroup of
// the PFN launch): occupancy -> LDS, inclusive scan along x (a wave per row), inclusive scan along y (a wave per
// column, lanes = rows: shuffles instead of a serial walk), then the anchors' four lookups out of LDS.
// Replaces, per frame, load_data.py:586-591, :3054-3055, :558-584, :3070 (see anchor_mask.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pp_common.h"

#define AM_MAX_CELLS 8192      // int32 cells of the (row-padded) BEV grid image in LDS (32 KB)
#define AM_PFN_BLOCKS 4         // workgroups of the PFN launch that share a frame's anchors

// NT threads (a multiple of 64); sI: AM_MAX_CELLS ints of LDS; ny * (nx | 1) <= AM_MAX_CELLS.  The LDS image has an
// ODD row stride (nx | 1): the scan along y reads a column with lanes = rows, and an even stride would put the 64
// rows on two banks.  Anchors [a_begin, a_end) are looked up (the PFN launch splits a frame's anchors over several
// workgroups, each with its own copy of the integral image).
template <int NT>
__device__ __forceinline__ void anchor_mask_frame_block(const int* __restrict__ map, int nz, int ny, int nx,
                                                        const int* __restrict__ cells, int64_t a_begin, int64_t a_end,
                                                        float threshold, uint8_t* __restrict__ mask, int* __restrict__ sI) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64;
    const int plane = ny * nx, ls = nx | 1;
    // occupancy: the block is a handful of memory round trips long, so every load of a phase is issued before the
    // first one is used (16-byte loads of 4 cells, up to 8 in flight per thread)
    if ((plane & 3) == 0 && nz <= 2) {
        const int n4 = plane >> 2;
        for (int i0 = 0; i0 < n4; i0 += 4 * NT) {
            int4 v[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int z = 0; z < 2; ++z) {
                    const int i = i0 + k * NT + tid;
                    v[k][z] = (i < n4 && z < nz) ? reinterpret_cast<const int4*>(map + (size_t)z * plane)[i] : make_int4(-1, -1, -1, -1);
                }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * NT + tid;
                if (i < n4) {
                    const int c[4] = {(v[k][0].x >= 0) + (v[k][1].x >= 0), (v[k][0].y >= 0) + (v[k][1].y >= 0),
                                      (v[k][0].z >= 0) + (v[k][1].z >= 0), (v[k][0].w >= 0) + (v[k][1].w >= 0)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cell = 4 * i + j, y = cell / nx;
                        sI[y * ls + (cell - y * nx)] = c[j];
                    }
                }
            }
        }
    } else {
        for (int i = tid; i < plane; i += NT) {
            int v = 0;
            for (int z = 0; z < nz; ++z) v += (map[(size_t)z * plane + i] >= 0) ? 1 : 0;
            const int y = i / nx;
            sI[y * ls + (i - y * nx)] = v;
        }
    }
    __syncthreads();
    for (int y = wave; y < ny; y += NW) {                // inclusive scan along x
        int carry = 0;
        for (int x0 = 0; x0 < nx; x0 += 64) {
            const int x = x0 + lane;
            const int incl = wave_inclusive_scan((x < nx) ? sI[y * ls + x] : 0);
            if (x < nx) sI[y * ls + x] = carry + incl;
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    __syncthreads();
    for (int x = wave; x < nx; x += NW) {                // inclusive scan along y: lanes = rows
        int carry = 0;
        for (int y0 = 0; y0 < ny; y0 += 64) {
            const int y = y0 + lane;
            const int incl = wave_inclusive_scan((y < ny) ? sI[y * ls + x] : 0);
            if (y < ny) sI[y * ls + x] = carry + incl;
            carry += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    __syncthreads();
    for (int64_t a0 = a_begin; a0 < a_end; a0 += 4 * NT) {     // four anchors per thread and round trip
        int4 c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t a = a0 + k * NT + tid;
            c[k] = (a < a_end) ? reinterpret_cast<const int4*>(cells)[a] : make_int4(0, 0, 0, 0);  // x0 y0 x1 y1
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t a = a0 + k * NT + tid;
            const int area = sI[c[k].w * ls + c[k].z] - sI[c[k].w * ls + c[k].x] - sI[c[k].y * ls + c[k].z] + sI[c[k].y * ls + c[k].x];
            if (a < a_end) mask[a] = ((float)area > threshold) ? 1 : 0;
        }
    }
}
