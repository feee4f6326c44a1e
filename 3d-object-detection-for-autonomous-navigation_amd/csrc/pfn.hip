// Fused PillarFeatureNet + PointPillarsScatter, canvas-cell centric.
//
// Replaces PillarFeatureNet.call (reference model/pointpillars.py:128-225) and
// PointPillarsScatter.call (:285-341), and the NCHW->NHWC transpose the RPN does
// first (model/voxelnet.py:697):
//   mean over the pillar's points; features [raw F | xyz - mean | xy - pillar centre];
//   rows >= num_points masked to zero; Dense(Fa->C, no bias) -> BN(eps 1e-3) -> ReLU;
//   max over ALL T rows (a pillar with fewer than T points also maxes with the
//   padded-row constant ReLU(beta - gamma*mean/sqrt(var+eps)));
//   scatter to canvas[y*nx + x], pillars of different z-cells that share (y, x)
//   are SUMMED (tf.scatter_nd semantics).
// BatchNorm is folded into the dense kernel / bias at weight-finalise time.
//
// Mapping: every canvas cell is written exactly once -- zeros when no pillar
// maps to it, otherwise the sum over its z-cells -- so there is no memset, no
// atomics and the result is bit-reproducible.  One wavefront owns 16
// consecutive cells of a frame: lanes own channels (C/64 per lane; the Fa x C
// weights live in VGPRs), the pillar's points are loaded one per lane
// (coalesced through the CSR index list) and broadcast with v_readlane, the
// per-pillar max is a running register max.  HBM-bound: reads 4F bytes per
// point + the cell map, writes 4C bytes per canvas cell.
//
// Arithmetic: the dense layer is linear in the 8 decorated features, so per
// pillar the terms that do not depend on the point are folded into a constant:
//   sum_k W_k f_k = (W_x + W_cx + W_px) x' + (W_y + W_cy + W_py) y' + (W_z + W_cz) z [+ W_i i] + K
//   x' = x - pillar_centre_x (the reference's f_center, small), K = b + W_x cx + W_y cy
//        - W_cx (mean_x - cx) - W_cy (mean_y - cy) - W_cz mean_z
// (3-4 FMAs per point and channel instead of 8-9; working in pillar-local x', y'
// keeps every product small, so the rounding error stays below the reference's
// own evaluation of the raw-coordinate terms), and max_j ReLU(a_j) = ReLU(max_j a_j).
#include <stdlib.h>

#include "pp_common.h"
#include "anchor_mask_dev.h"

template <int N>
struct FVec { float v[N]; };

__device__ __forceinline__ float bcast(float x, int srclane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), srclane));
}

// v_max_f32 without the canonicalising self-max fmaxf() puts in front of it (the operands here are never NaN:
// the voxeliser rejects non-finite points and the weights are finite)
__device__ __forceinline__ float raw_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

// cells per wavefront: pillars hold 1..T points, so a tile's cost varies by 100x; small tiles keep
// the heaviest wave short (the kernel ends with its slowest wave)
#define PFN_CW 4
// DIST: voxel_feature_extractor.with_distance (model/pointpillars.py:185-188): a ninth / tenth input feature,
// the Euclidean norm of the point's raw xyz (tf.norm, float32), after the cluster and centre offsets
template <int CPL, int F, bool PADDED, bool DIST = false>
__global__ __launch_bounds__(256) void k_pfn_canvas(PfnParams p) {
    constexpr int FA = F + 5 + (DIST ? 1 : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int ncanvas = p.ny * p.nx;
    const int cell0 = (blockIdx.x * 4 + wave) * PFN_CW;
    if (cell0 >= ncanvas) return;
    const int ncells = min(PFN_CW, ncanvas - cell0);
    const int C = p.C;
    const int ch0 = lane * CPL;
    const bool ch_ok = ch0 < C;

    // weights for this lane's channels
    float w[FA][CPL], bias[CPL];
#pragma unroll
    for (int k = 0; k < FA; ++k)
#pragma unroll
        for (int q = 0; q < CPL; ++q) w[k][q] = ch_ok ? p.w[k * C + ch0 + q] : 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) bias[q] = ch_ok ? p.bias[ch0 + q] : 0.f;
    // folded per-coordinate weights: raw + cluster-offset + centre-offset columns
    float wsx[CPL], wsy[CPL], wsz[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        wsx[q] = (w[0][q] + w[F + 0][q]) + w[F + 3][q];
        wsy[q] = (w[1][q] + w[F + 1][q]) + w[F + 4][q];
        wsz[q] = w[2][q] + w[F + 2][q];
    }

    const int* map = p.cellmap + (size_t)b * p.nz * ncanvas;
    unsigned occ = 0;
    for (int z = 0; z < p.nz; ++z) {
        const int v = (lane < ncells) ? map[(size_t)z * ncanvas + cell0 + lane] : -1;
        occ |= (unsigned)(__ballot(v >= 0) & ((1ull << PFN_CW) - 1ull));
    }

    float* cbase = p.canvas + ((size_t)b * ncanvas + cell0) * C;
    // zero-fill the cells no pillar maps to (16-byte stores, one cell per C/4 lanes)
    {
        const int nq = ncells * C / 4;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = lane; q < nq; q += 64) {
            const int c = (q * 4) / C;
            if (!((occ >> c) & 1u)) reinterpret_cast<float4*>(cbase)[q] = z4;
        }
    }

    const int n0 = PADDED ? 0 : p.offsets[b];
    const int* ps = PADDED ? nullptr : p.pillar_start + (size_t)b * (p.max_voxels + 1);
    const int T = p.T;

    while (occ) {
        const int c = __builtin_ctz(occ);
        occ &= occ - 1;
        const int cell = cell0 + c;
        const int yi = cell / p.nx, xi = cell - yi * p.nx;
        // pillar centre: float(idx) * v + offset, two float32 roundings (model/pointpillars.py:156-171)
        const float cxf = __fadd_rn(__fmul_rn((float)xi, p.vx), p.x_off);
        const float cyf = __fadd_rn(__fmul_rn((float)yi, p.vy), p.y_off);
        float acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = 0.f;

        for (int z = 0; z < p.nz; ++z) {
            // wave-uniform (every lane loads the same word); make that visible to the compiler
            const int pid = __builtin_amdgcn_readfirstlane(map[(size_t)z * ncanvas + cell]);
            if (pid < 0) continue;
            int n, nsum;
            const float* src;           // rows of this pillar (padded tensor, or its run of the pillar-sorted points)
            if (PADDED) {
                n = __builtin_amdgcn_readfirstlane(min(max(p.num_points[pid], 0), T));
                nsum = T;               // the reference sums all T rows (model/pointpillars.py:143)
                src = p.voxels + (size_t)pid * T * F;
            } else {
                const int start = __builtin_amdgcn_readfirstlane(ps[pid]);
                n = __builtin_amdgcn_readfirstlane(min(ps[pid + 1] - start, T));
                nsum = n;
                src = p.pts_sorted + (size_t)(n0 + start) * F;
            }
            // ---- mean over the pillar ----
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int j0 = 0; j0 < nsum; j0 += 64) {
                const int j = j0 + lane;
                if (j < nsum) {
                    const float* q = src + (size_t)j * F;
                    sx += q[0]; sy += q[1]; sz += q[2];
                }
            }
            sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
            const float fn = (float)n;
            const float mx = sx / fn, my = sy / fn, mz = sz / fn;

            // ---- per pillar constant, then per point 3-4 FMAs + max per channel ----
            const float dxm = mx - cxf, dym = my - cyf;
            float kp[CPL], m[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                float k0 = bias[q];
                k0 = fmaf(cxf, w[0][q], k0);
                k0 = fmaf(cyf, w[1][q], k0);
                k0 = fmaf(-dxm, w[F + 0][q], k0);
                k0 = fmaf(-dym, w[F + 1][q], k0);
                k0 = fmaf(-mz, w[F + 2][q], k0);
                kp[q] = k0;
                m[q] = -3.0e38f;
            }
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                float pt[F];
#pragma unroll
                for (int f = 0; f < F; ++f) pt[f] = 0.f;
                if (j < n) {
                    const float* q = src + (size_t)j * F;
#pragma unroll
                    for (int f = 0; f < F; ++f) pt[f] = q[f];
                }
                float pdist = 0.f;
                if (DIST) pdist = __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(pt[0], pt[0]), __fmul_rn(pt[1], pt[1])), __fmul_rn(pt[2], pt[2])));
                pt[0] = pt[0] - cxf;   // pillar-local coordinates (the reference's f_center features)
                pt[1] = pt[1] - cyf;
                const int cnt = min(64, n - j0);
                for (int jj = 0; jj < cnt; ++jj) {
                    float ft[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) ft[f] = bcast(pt[f], jj);
                    const float fd = DIST ? bcast(pdist, jj) : 0.f;
#pragma unroll
                    for (int q = 0; q < CPL; ++q) {
                        float o = kp[q];
                        o = fmaf(ft[0], wsx[q], o);
                        o = fmaf(ft[1], wsy[q], o);
                        o = fmaf(ft[2], wsz[q], o);
                        if (F > 3) o = fmaf(ft[F - 1], w[F - 1][q], o);
                        if (DIST) o = fmaf(fd, w[FA - 1][q], o);
                        m[q] = fmaxf(m[q], o);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) m[q] = fmaxf(m[q], 0.f);   // ReLU after the max (monotone)
            if (n < T) {  // zero-padded rows: Dense(0) = 0 -> BN shift -> ReLU
#pragma unroll
                for (int q = 0; q < CPL; ++q) m[q] = fmaxf(m[q], fmaxf(bias[q], 0.f));
            }
            if (p.feat_out != nullptr && ch_ok) {
                const size_t row = PADDED ? (size_t)pid : ((size_t)b * p.max_voxels + pid);
#pragma unroll
                for (int q = 0; q < CPL; ++q) p.feat_out[row * C + ch0 + q] = m[q];
            }
#pragma unroll
            for (int q = 0; q < CPL; ++q) acc[q] += m[q];
        }
        if (ch_ok) {
            float* dst = cbase + (size_t)c * C + ch0;
            if constexpr (CPL == 4) {
                *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else if constexpr (CPL == 2) {
                *reinterpret_cast<float2*>(dst) = make_float2(acc[0], acc[1]);
            } else {
                dst[0] = acc[0];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Fused path (CSR source), second generation: memory round trips hoisted out of the per-pillar
// loop, one pass over the points.  A wavefront owns PFN2_CW consecutive canvas cells = up to
// PFN2_CW * nz pillar slots (cell-major).  (0) one lane per slot reads the cell map and the slot's
// CSR range -- two dependent loads for ALL slots at once -- and derives its pillar centre; (1) a
// wave prefix sum lays the slots' points out as one stream; (2) the stream is read 64 points at a
// time (one point per lane: CSR index, then the point; the next batch is prefetched while the
// current one is used); (3) the wave walks the stream in order with wave-uniform control flow:
// pillar-local coordinates are broadcast with v_readlane, every lane keeps the running max of
//     a_j = x'_j (Wx+Wcx+Wpx) + y'_j (Wy+Wcy+Wpy) + z_j (Wz+Wcz) [+ i_j Wi]
// for its channels and the running coordinate sums; since the per-pillar constant K (bias, centre
// and mean terms) does not depend on j, max_j (K + a_j) = K + max_j a_j, so the mean is only needed
// when the slot ends: no second pass over the points.  Slot / cell boundaries finalise (ReLU, pad
// constant, sum over z) and write the 4C-byte canvas row -- zeros for cells no pillar maps to.
#ifndef PFN2_CW
#define PFN2_CW 8
#endif
// PIL (sparse canvas, round 4): PILLAR-centric ownership.  On a KITTI-shaped grid 2.7 % of the cells hold a pillar, and
// the cell-centric launch above spent its time dispatching ~27 000 waves per frame of which nearly all read one line
// of the cell map and left (199 us for 184 k pillars at B = 32).  Here a wave owns PFN2_CW consecutive PILLARS of the
// voxeliser's list: lane c reads pillar c's cell; a pillar that is the lowest occupied z-cell of its (y, x) column
// takes the whole column (its z-neighbours come from the cell map and are summed in z order, exactly as above), any
// other pillar's slots stay empty; from the slot metadata on the kernel is the same stream walk.  Cells without a
// pillar are never visited, let alone written.
template <int CPL, int F, bool PIL = false>
__global__ __launch_bounds__(256) void k_pfn_canvas2(PfnParams p) {
    constexpr int FA = F + 5;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int ncanvas = p.ny * p.nx;
    // the frame's anchor mask rides in this launch: its workgroups come FIRST in dispatch order (they are a few serial
    // phases long; at the end of the grid they were the launch's tail: +25 us at B = 64)
    const int amb = (p.am_mask != nullptr) ? AM_PFN_BLOCKS : 0;
    if ((int)blockIdx.x < amb) {
        // the LDS image is DYNAMIC shared memory, sized by the launcher (zero without the mask): a static array would
        // cap every PFN workgroup of every launch at 160 KB / 32 KB = 5 per CU (measured: B = 64 73 -> 90 us)
        extern __shared__ int sI[];
        const int part = (int)blockIdx.x;                                       // this workgroup's share of the anchors
        const int64_t per = (p.am_A + AM_PFN_BLOCKS - 1) / AM_PFN_BLOCKS;
        const int64_t a0 = part * per, a1 = (a0 + per < p.am_A) ? a0 + per : p.am_A;
        anchor_mask_frame_block<256>(p.cellmap + (size_t)b * p.nz * ncanvas, p.nz, p.ny, p.nx, p.am_cells, a0, a1,
                                     p.am_threshold, p.am_mask + (size_t)b * p.am_A, sI);
        return;
    }
    // a wave's PFN2_CW cells are spread over the map (cell = wave id + c * waves per frame), not contiguous:
    // points cluster (a person is a few dozen neighbouring cells with up to T points each), and a wave that owned
    // eight neighbouring crowded cells ran 30x longer than the average one -- the kernel's duration was its tail
    // (sparse canvas: contiguous cells instead -- on a mostly empty grid whole waves then have nothing to do)
    // cell stride: waves per frame (NW * PFN2_CW >= ncanvas; the anchor-mask workgroup does not count) or 1
    const int bx = (int)blockIdx.x - amb;
    const int NW = p.sparse ? 1 : ((int)gridDim.x - amb) * 4;
    const int wid = (PIL || p.sparse) ? (bx * 4 + wave) * PFN2_CW : bx * 4 + wave;   // first cell (PIL: first pillar)
    const int np_frame = PIL ? p.npillars[b] : 0;
    if (PIL ? (wid >= np_frame) : (wid >= ncanvas)) return;
    const int ncells = PIL ? min(PFN2_CW, np_frame - wid) : min(PFN2_CW, (ncanvas - wid + NW - 1) / NW);
    const int nz = p.nz;
    const int NS = ncells * nz;                       // slots of this wave (<= 64, checked by the launcher)
    const int C = p.C, T = p.T;
    const int ch0 = lane * CPL;
    const bool ch_ok = ch0 < C;

    // ---- (0) slot metadata, one lane per slot ----
    const int n0 = p.offsets[b];
    const int* ps = p.pillar_start + (size_t)b * (p.max_voxels + 1);
    int pid = -1, start = 0, cnt = 0, slot_cell = 0;
    float slot_cx = 0.f, slot_cy = 0.f;
    [[maybe_unused]] int col_xy = 0;                  // PIL: lane c = the (y, x) cell of the wave's pillar c
    if constexpr (PIL) {
        int z0 = 0;
        bool primary = false;
        if (lane < ncells) {
            const int cz = p.pillar_cell[(size_t)b * p.max_voxels + wid + lane];
            z0 = cz / ncanvas;
            col_xy = cz - z0 * ncanvas;
            primary = true;                           // ... unless a lower z-cell of the column holds a pillar too
            for (int z = 0; z < z0; ++z) primary = primary & (p.cellmap[((size_t)b * nz + z) * ncanvas + col_xy] < 0);
            if (p.occbits != nullptr) {               // the column's bit of the frame's occupancy bitmap (bit x + 1 of row y)
                const int yy = col_xy / p.nx, xx = col_xy - yy * p.nx;
                atomicOr(p.occbits + ((size_t)b * p.ny + yy) * occ_words(p.nx) + ((xx + 1) >> 6), 1ull << ((xx + 1) & 63));
            }
        }
        if (lane < NS) {
            const int c = lane / nz, z = lane - c * nz;
            slot_cell = c;
            const int cxy = __shfl(col_xy, c), zc = __shfl(z0, c);
            const bool prim = __shfl((int)primary, c) != 0;
            if (prim && z == zc) pid = wid + c;
            else if (prim && z > zc) pid = p.cellmap[((size_t)b * nz + z) * ncanvas + cxy];
        }
    } else {
        if (lane < NS) {
            const int c = lane / nz, z = lane - c * nz;
            slot_cell = c;
            pid = p.cellmap[((size_t)b * nz + z) * ncanvas + wid + c * NW];
        }
    }
    // sparse canvas: most waves of a mostly empty grid have nothing to write -- leave before any other work
    if (p.sparse && __ballot(pid >= 0) == 0ull) return;
    if (lane < NS) {
        const int c = slot_cell;
        if (pid >= 0) {
            start = ps[pid];
            cnt = min(ps[pid + 1] - start, T);
        }
        const int cell = PIL ? __shfl(col_xy, c) : wid + c * NW;
        const int yi = cell / p.nx, xi = cell - yi * p.nx;
        slot_cx = __fadd_rn(__fmul_rn((float)xi, p.vx), p.x_off);      // model/pointpillars.py:156-171
        slot_cy = __fadd_rn(__fmul_rn((float)yi, p.vy), p.y_off);
    }
    // weights of this lane's channels (independent of the loads above)
    float w[FA][CPL], bias[CPL];
#pragma unroll
    for (int k = 0; k < FA; ++k)
#pragma unroll
        for (int q = 0; q < CPL; ++q) w[k][q] = ch_ok ? p.w[k * C + ch0 + q] : 0.f;
#pragma unroll
    for (int q = 0; q < CPL; ++q) bias[q] = ch_ok ? p.bias[ch0 + q] : 0.f;
    float wsx[CPL], wsy[CPL], wsz[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        wsx[q] = (w[0][q] + w[F + 0][q]) + w[F + 3][q];
        wsy[q] = (w[1][q] + w[F + 1][q]) + w[F + 4][q];
        wsz[q] = w[2][q] + w[F + 2][q];
    }

    // ---- (1) stream layout: inclusive prefix of cnt over the slot lanes ----
    const int incl = wave_inclusive_scan(cnt);
    const int excl = incl - cnt;
    const int tot = __builtin_amdgcn_readlane(incl, 63);
    if (p.sparse && tot == 0) return;                 // sparse canvas: nothing to write for cells without pillars
    const float* src = p.pts_sorted + (size_t)n0 * F;   // this frame's pillar-sorted points: a slot is one contiguous run

    // one batch = stream positions base..base+63, one per lane: slot, coordinates
    struct Batch { int slot; float x, y, z, i; };
    auto load_batch = [&](int base) -> Batch {
        Batch r;
        r.slot = -1; r.x = r.y = r.z = r.i = 0.f;
        const int g = base + lane;
        int sl = 0;
        for (int s = 0; s < NS; ++s) sl += (g >= __builtin_amdgcn_readlane(incl, s)) ? 1 : 0;
        sl = min(sl, NS - 1);
        const int st = __shfl(start, sl), ex = __shfl(excl, sl);
        const float pcx = __shfl(slot_cx, sl), pcy = __shfl(slot_cy, sl);
        if (g < tot) {
            r.slot = sl;
            const float* q = src + (size_t)(st + (g - ex)) * F;
            r.x = q[0] - pcx;      // pillar-local coordinates (the reference's f_center features), one
            r.y = q[1] - pcy;      // subtraction per point instead of one per point and lane
            r.z = q[2];
            if (F > 3) r.i = q[F - 1];
        }
        return r;
    };

    // ---- (3) walk the stream ----
    float* cbase = p.canvas + ((size_t)b * ncanvas + (PIL ? 0 : wid)) * C;
    const size_t cstep = (size_t)NW * C;              // floats between two cells of this wave (PIL: unused)
    float acc[CPL], m[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) { acc[q] = 0.f; m[q] = -3.0e38f; }
    int cur_slot = -1, cur_cell = 0;
    float cxf = 0.f, cyf = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;      // wave-uniform values (same in every lane)
    bool cell_dirty = false;                          // a pillar of the current cell has been added (wave-uniform)
    auto write_cell = [&](int c) {
        // sparse canvas: cells without a pillar are not written at all (the first layer looks the cell up in
        // the cell map and reads zeros); writing the zeros of an almost empty grid is most of the traffic
        if (ch_ok && (cell_dirty || !p.sparse)) {
            float* dst = PIL ? cbase + (size_t)__builtin_amdgcn_readlane(col_xy, c) * C + ch0 : cbase + (size_t)c * cstep + ch0;
            if constexpr (CPL == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            else if constexpr (CPL == 2) *reinterpret_cast<float2*>(dst) = make_float2(acc[0], acc[1]);
            else dst[0] = acc[0];
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
        cell_dirty = false;
    };
    auto finish_slot = [&](int s) {
        const int n = __builtin_amdgcn_readlane(cnt, s);
        // means of the RAW coordinates from the sums of the pillar-local ones (exactly what the reference's
        // f_cluster / f_center differences need: mean_x - centre_x = mean of x').  One v_rcp_f32 (1 ulp) and three
        // multiplies instead of three IEEE divisions: the division sequences were ~40 of the ~90 instructions a
        // pillar costs, and a pillar has 3.2 points on average (the result moves by <= 2 ulp of the mean)
        const float rn = __builtin_amdgcn_rcpf((float)n);
        const float dxm = sx * rn, dym = sy * rn, mz = sz * rn;
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            float k0 = bias[q];
            k0 = fmaf(cxf, w[0][q], k0);
            k0 = fmaf(cyf, w[1][q], k0);
            k0 = fmaf(-dxm, w[F + 0][q], k0);
            k0 = fmaf(-dym, w[F + 1][q], k0);
            k0 = fmaf(-mz, w[F + 2][q], k0);
            m[q] = fmaxf(k0 + m[q], 0.f);                                 // ReLU after the max (monotone)
        }
        if (n < T) {                                                     // zero-padded rows: ReLU(folded bias)
#pragma unroll
            for (int q = 0; q < CPL; ++q) m[q] = fmaxf(m[q], fmaxf(bias[q], 0.f));
        }
        if (p.feat_out != nullptr && ch_ok) {
            const size_t row = (size_t)b * p.max_voxels + __builtin_amdgcn_readlane(pid, s);
#pragma unroll
            for (int q = 0; q < CPL; ++q) p.feat_out[row * C + ch0 + q] = m[q];
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) { acc[q] += m[q]; m[q] = -3.0e38f; }
        cell_dirty = true;
    };
    auto begin_slot = [&](int s) {
        const int c = __builtin_amdgcn_readlane(slot_cell, s);
        while (cur_cell < c) { write_cell(cur_cell); ++cur_cell; }
        cxf = bcast(slot_cx, s);
        cyf = bcast(slot_cy, s);
        sx = sy = sz = 0.f;
        cur_slot = s;
    };
    Batch cur = load_batch(0);
    int left = 0;                                     // points of the current slot still to come (wave-uniform)
    int next_slot = 0;
    for (int base = 0; base < tot; base += 64) {
        Batch nxt = cur;
        if (base + 64 < tot) nxt = load_batch(base + 64);
        const int nb = min(64, tot - base);
        for (int jj = 0; jj < nb; ++jj) {
            if (left == 0) {                          // slot boundary: the stream is slot-major, skip empty slots
                if (cur_slot >= 0) finish_slot(cur_slot);
                while ((left = __builtin_amdgcn_readlane(cnt, next_slot)) == 0) ++next_slot;
                begin_slot(next_slot);
                ++next_slot;
            }
            --left;
            const float fx = bcast(cur.x, jj), fy = bcast(cur.y, jj), fz = bcast(cur.z, jj);
            const float fi = (F > 3) ? bcast(cur.i, jj) : 0.f;
            sx += fx; sy += fy; sz += fz;
            if constexpr (CPL == 2) {                 // two channels per lane: packed fp32 math
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                const f32x2 vx = {fx, fx}, vy = {fy, fy}, vz = {fz, fz};
                const f32x2 wx2 = {wsx[0], wsx[1]}, wy2 = {wsy[0], wsy[1]}, wz2 = {wsz[0], wsz[1]};
                f32x2 o = vx * wx2;
                o = __builtin_elementwise_fma(vy, wy2, o);
                o = __builtin_elementwise_fma(vz, wz2, o);
                if (F > 3) {
                    const f32x2 vi = {fi, fi}, wi2 = {w[F - 1][0], w[F - 1][1]};
                    o = __builtin_elementwise_fma(vi, wi2, o);
                }
                m[0] = raw_max(m[0], o.x);
                m[1] = raw_max(m[1], o.y);
            } else {
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    float o = fx * wsx[q];
                    o = fmaf(fy, wsy[q], o);
                    o = fmaf(fz, wsz[q], o);
                    if (F > 3) o = fmaf(fi, w[F - 1][q], o);
                    m[q] = raw_max(m[q], o);
                }
            }
        }
        cur = nxt;
    }
    if (cur_slot >= 0) finish_slot(cur_slot);
    while (cur_cell < ncells) { write_cell(cur_cell); ++cur_cell; }
}

// PP_PFN_KERNEL=1 selects the first-generation kernel for the fused path (A/B timing)
static bool pfn_first_generation();
// the extra anchor-mask workgroups ride in the second-generation CSR kernel only (dense canvas, grid fits the LDS image)
bool pfn_can_carry_anchor_mask(const PfnParams& p, bool padded_source) {
    return !padded_source && p.feat_out == nullptr && PFN2_CW * p.nz <= 64 && !pfn_first_generation() && !p.sparse &&
           p.ny * (p.nx | 1) <= AM_MAX_CELLS && !p.with_distance;
}
static bool pfn_first_generation() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("PP_PFN_KERNEL");
        v = (e && e[0] == '1') ? 1 : 0;
    }
    return v == 1;
}

static bool pfn_cell_centric_sparse() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("PP_PFN_SPARSE_CELLS"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// does launch_pfn run the pillar-centric kernel (the one that sets the occupancy bitmap) for these parameters?
static bool pfn_pillar_centric(const PfnParams& p, bool padded) {
    return !padded && !p.with_distance && p.sparse && p.pillar_cell != nullptr && p.npillars != nullptr && PFN2_CW * p.nz <= 64 &&
           !pfn_first_generation() && !pfn_cell_centric_sparse();
}
bool pfn_writes_occbits(const PfnParams& p, bool padded) { return pfn_pillar_centric(p, padded); }

template <int CPL, int F>
static void launch_pfn_t(const PfnParams& p, bool padded, hipStream_t s) {
    const int ncanvas = p.ny * p.nx;
    dim3 grid((ncanvas + 4 * PFN_CW - 1) / (4 * PFN_CW), p.batch);
    if (p.with_distance) {   // the generic kernel carries the distance feature (not a shipped configuration)
        if (padded) PP_LAUNCH("k_pfn_canvas", (k_pfn_canvas<CPL, F, true, true>), grid, dim3(256), 0, s, p);
        else PP_LAUNCH("k_pfn_canvas", (k_pfn_canvas<CPL, F, false, true>), grid, dim3(256), 0, s, p);
    } else if (padded) {
        PP_LAUNCH("k_pfn_canvas", (k_pfn_canvas<CPL, F, true>), grid, dim3(256), 0, s, p);
    } else if (pfn_pillar_centric(p, padded)) {
        // sparse canvas: a wave per PFN2_CW pillars of the voxeliser's list (PP_PFN_SPARSE_CELLS=1: the cell-centric walk)
        dim3 gridp((p.max_voxels + 4 * PFN2_CW - 1) / (4 * PFN2_CW), p.batch);
        PP_LAUNCH("k_pfn_canvas2", (k_pfn_canvas2<CPL, F, true>), gridp, dim3(256), 0, s, p);
    } else if (PFN2_CW * p.nz <= 64 && !pfn_first_generation()) {
        dim3 grid2((ncanvas + 4 * PFN2_CW - 1) / (4 * PFN2_CW) + (p.am_mask != nullptr ? AM_PFN_BLOCKS : 0), p.batch);
        const size_t lds = (p.am_mask != nullptr) ? (size_t)p.ny * (p.nx | 1) * sizeof(int) : 0;
        PP_LAUNCH("k_pfn_canvas2", (k_pfn_canvas2<CPL, F>), grid2, dim3(256), lds, s, p);
    } else {
        PP_LAUNCH("k_pfn_canvas", (k_pfn_canvas<CPL, F, false>), grid, dim3(256), 0, s, p);
    }
}

int launch_pfn(const PfnParams& p, bool padded, hipStream_t s) {
    if (p.batch <= 0) return 0;
    const int C = p.C;
    int cpl;
    if (C <= 64) cpl = 1;
    else if (C == 128) cpl = 2;
    else if (C == 256) cpl = 4;
    else return PP_ERR_UNSUPPORTED;
    if (C % 4 != 0) return PP_ERR_UNSUPPORTED;
    if (p.F != 3 && p.F != 4) return PP_ERR_UNSUPPORTED;
#define PFN_CASE(CPLV, FV) \
    if (cpl == CPLV && p.F == FV) { launch_pfn_t<CPLV, FV>(p, padded, s); return 0; }
    PFN_CASE(1, 3) PFN_CASE(1, 4) PFN_CASE(2, 3) PFN_CASE(2, 4) PFN_CASE(4, 3) PFN_CASE(4, 4)
#undef PFN_CASE
    return PP_ERR_UNSUPPORTED;
}
