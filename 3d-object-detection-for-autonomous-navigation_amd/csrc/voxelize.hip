// Pillar voxelisation with the reference's exact sequential semantics, in parallel.
//
// Replaces points_to_voxel / _points_to_voxel_reverse_kernel
// (reference load_data.py:695-771, :593-641): a scalar loop over the points in
// input order in which (1) the pillar id of a cell is the order in which the
// cell is first hit, (2) the slot of a point in its pillar is its arrival order,
// cut at max_points, (3) the scan BREAKS when pillar number max_voxels+1 would
// be opened, dropping every later point, and (4) the cell index is
// floor((p - min) / voxel) evaluated in float64.
//
// Parallel restatement (bit-exact, deterministic -- no result depends on the
// order in which atomics land):
//   k_cell_first   one thread per point: float64 cell -> linear cell id (or -1);
//                  first[cell] = min(point index) by atomicMin.
//   k_voxel_frame  one 1024-thread workgroup per frame:
//     A. flag = "this point is the first of its cell"; pillar id = exclusive
//        prefix count of flags (ballot + popcount per wave, running carry);
//        the break point is the flagged point whose prefix == max_voxels.
//     B. stable compaction of the surviving points (cell valid, index < break)
//        into (key = pillar id, value = point index).
//     C. stable LSD radix sort by pillar id (<= 8 bits per pass; per-wave digit
//        histograms in LDS; in-wave rank by digit-bit ballots), so points end
//        up grouped by pillar in arrival order: the CSR layout the PFN reads.
//     D. pillar_start[] from the key boundaries, sorted point indices.
//   k_sort_points  one thread per surviving point: the pillar-sorted copy of the points (the CSR
//                  payload the PFN streams).
// The padded [P,T,F] tensor of the reference is NOT materialised on the fused
// path; k_voxel_expand produces it for the compat / parity entry point.
//
// Memory-bound integer work: coalesced 4-byte loads over the point axis, all
// scratch L2-resident (a 16k-point frame moves ~1 MB).
#include "pp_common.h"

#define VT 1024
#define VWAVES 16
#define VL_PPT 16               // points per thread on the LDS path
#define VL_CAP (VT * VL_PPT)    // frames up to this many points take the LDS path

__device__ __forceinline__ unsigned long long lanemask_lt() {
    unsigned lane = threadIdx.x & 63u;
    return (lane == 0) ? 0ull : (~0ull >> (64u - lane));
}

__global__ __launch_bounds__(256) void k_cell_first(const float* __restrict__ pts,
                                                    const int* __restrict__ offsets, int F, VoxGeom g,
                                                    int* __restrict__ cell, int* __restrict__ first,
                                                    int* __restrict__ cellmap, const PpFeed* __restrict__ feed,
                                                    float* __restrict__ pts_dst, int* __restrict__ offsets_dst,
                                                    unsigned long long* __restrict__ occbits, int occ_n) {
    const int b = blockIdx.y;
    if (feed != nullptr) {   // zero-copy feed: host-resident offsets / points, device copies for the later kernels
        offsets = feed->offsets;
        pts = feed->src;
    }
    // this frame's cell -> pillar map is cleared here (-1 = empty; k_voxel_frame, next in the stream, is the
    // first to write it): one launch less than a separate memset node
    // (16-byte stores when the frame's map is 16-byte aligned, ncell % 4 == 0.  Measured on the KITTI-shaped 214 272-cell
    // map: no change -- the kernel's 40 us per 32 frames there are the device-scope atomicMin of the global first-point
    // table, which grids too large for the LDS tables still need)
    if (cellmap != nullptr) {
        int* cm = cellmap + (size_t)b * g.ncell;
        if ((g.ncell & 3) == 0) {
            const int4 m1 = make_int4(-1, -1, -1, -1);
            for (int e = blockIdx.x * 256 + threadIdx.x; e < (g.ncell >> 2); e += gridDim.x * 256) reinterpret_cast<int4*>(cm)[e] = m1;
        } else {
            for (int e = blockIdx.x * 256 + threadIdx.x; e < g.ncell; e += gridDim.x * 256) cm[e] = -1;
        }
    }
    // ... and so is its occupancy bitmap (the pillar-centric PFN launch sets the bits)
    if (occbits != nullptr)
        for (int e = blockIdx.x * 256 + threadIdx.x; e < occ_n; e += gridDim.x * 256) occbits[(size_t)b * occ_n + e] = 0ull;
    const int n0 = offsets[b];
    const int n1 = offsets[b + 1];
    const int n = n1 - n0;
    if (feed != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        offsets_dst[b] = n0;
        if (b == (int)gridDim.y - 1) offsets_dst[b + 1] = n1;
    }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float* ps_ = pts + (size_t)(n0 + i) * F;
    float p[3] = {ps_[0], ps_[1], ps_[2]};      // one trip over the host link per coordinate (zero-copy feed)
    if (feed != nullptr) {
        float* q = pts_dst + (size_t)(n0 + i) * F;
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
        for (int j = 3; j < F; ++j) q[j] = ps_[j];
    }
    int c3[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // float32 point promoted to float64, minus float64 range, IEEE float64 divide, floor
        double c = floor(((double)p[j] - g.lo[j]) / g.vs[j]);
        // reference: reject if c < 0 or c >= grid.  A NaN coordinate passes that test in
        // the reference and then indexes out of bounds (undefined); here it is rejected.
        if (!(c >= 0.0 && c < (double)g.grid[j])) ok = false;
        c3[j] = ok ? (int)c : 0;
    }
    int lin = -1;
    if (ok) {
        lin = (c3[2] * g.grid[1] + c3[1]) * g.grid[0] + c3[0];  // (z, y, x)
        if (first != nullptr) atomicMin(&first[(size_t)b * g.ncell + lin], i);   // NULL: k_voxel_frame finds the minima in LDS
    }
    cell[n0 + i] = lin;
}

// exclusive scan of one int per thread over a 1024-thread block
__device__ __forceinline__ int block_excl_scan(int v, int* s_tmp, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = wave_inclusive_scan(v);
    if (lane == 63) s_tmp[wave] = x;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VWAVES; ++w) {
        int t = s_tmp[w];
        if (w < wave) woff += t;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return woff + x - v;
}

int voxel_sort_passes(int max_voxels) {
    int bits = 1;
    while ((1 << bits) < max_voxels) ++bits;
    return (bits + 7) / 8;
}

// BITS: digit width of the radix passes as a compile-time constant (the ballot loops unroll), 0 = run time.
// PPT: points per thread of the register / LDS path -- 16 (frames up to 16 384 points; the upper half of the
// 128 KB sort buffer then carries the first-point / pillar-id tables of grids up to 16 384 cells) or 32 (frames up
// to 32 768 points, KITTI-sized clouds: the whole buffer is sort space, the cell tables stay in global memory).
template <int BITS, int PPT>
__global__ __launch_bounds__(VT) void k_voxel_frame(
    const int* __restrict__ offsets, const int* __restrict__ cell, const int* __restrict__ first,
    int* __restrict__ cellmap, unsigned* keyA, unsigned* idxA, unsigned* keyB, unsigned* idxB,
    int* __restrict__ pillar_start, int* __restrict__ pillar_cell, int* __restrict__ npillars,
    int* __restrict__ nvalid_out, int ncell, int max_voxels, int npass, int bits_rt) {
    const int bits = (BITS > 0) ? BITS : bits_rt;
    __shared__ int s_tot[VWAVES];
    __shared__ int s_carry;
    __shared__ int s_break;
    __shared__ int s_tmp[VWAVES];
    __shared__ int s_hist[256 * VWAVES];
    constexpr int CAP = VT * PPT;                    // points of a frame on the register / LDS path
    constexpr bool TABLES = (PPT == 16);             // cell tables fit beside the sort space
    __shared__ unsigned s_sort[2 * VL_CAP];          // 128 KB: PPT 16: [sort | cell tables], PPT 32: sort

    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = offsets[b];
    const int n = offsets[b + 1] - n0;
    const int* fcell = cell + n0;
    const int* ffirst = (first != nullptr) ? first + (size_t)b * ncell : nullptr;
    int* fmap = cellmap + (size_t)b * ncell;
    unsigned* kA = keyA + n0;
    unsigned* vA = idxA + n0;
    unsigned* kB = keyB + n0;
    unsigned* vB = idxB + n0;
    const unsigned long long lt = lanemask_lt();

    if (tid == 0) { s_carry = 0; s_break = 0x7fffffff; }
    __syncthreads();

    // ---- fast path: a frame of up to VL_CAP points is processed out of registers and LDS.  Every thread
    // owns PPT consecutive points (all of its loads are issued at once: one memory round trip per phase
    // instead of one per 1024 points), pillar ids / compaction offsets come from one block-wide scan each,
    // and the (pillar id, point index) pairs are packed into one 32-bit word and radix-sorted in LDS. ----
    int ib = 1;
    while ((1 << ib) < n) ++ib;                 // bits of a point index
#ifdef PP_VOX_STAMPS   // diagnostic build: phase times of frame 0 (100 MHz ticks), printed by thread 0
    long long vst[24];
    int vsn = 0;
#define V_STAMP() { if (vsn < 24) vst[vsn++] = wall_clock64(); }
#else
#define V_STAMP() {}
#endif
    V_STAMP()
    if (n <= CAP && npass * bits + ib <= 32 && (first != nullptr || (TABLES && ncell <= VL_CAP))) {
        const int ppt = (n + VT - 1) / VT;      // <= PPT
        const int i0 = tid * ppt;
        int c[PPT];
        if (ppt == PPT && ((n0 | n) & 3) == 0 && i0 + PPT <= n) {   // full frame: 16-byte loads (wave-uniform but for the tail)
#pragma unroll
            for (int k = 0; k < PPT; k += 4) {
                const int4 v = *reinterpret_cast<const int4*>(fcell + i0 + k);
                c[k] = v.x; c[k + 1] = v.y; c[k + 2] = v.z; c[k + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) c[k] = (k < ppt && i0 + k < n) ? fcell[i0 + k] : -1;
        }
        int f[PPT];
        if (first != nullptr) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) f[k] = (c[k] >= 0) ? ffirst[c[k]] : -1;
        } else {
            // first point index of every cell by LDS atomics (the sort buffers are not in use yet): no global
            // atomics, whose same-address traffic on crowded cells serialises across the chip
            // (upper half of the sort buffers: it stays intact until the first sort pass scatters into it, and
            // carries the cell -> pillar id map after the first-point indices have been consumed)
            int* s_first = reinterpret_cast<int*>(s_sort) + VL_CAP;
            for (int e = tid; e < ncell; e += VT) s_first[e] = 0x7fffffff;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PPT; ++k) if (c[k] >= 0) atomicMin(&s_first[c[k]], i0 + k);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PPT; ++k) f[k] = (c[k] >= 0) ? s_first[c[k]] : -1;
            __syncthreads();   // s_sort is written below
        }
        V_STAMP()   // 1: cells loaded, first-of-cell known
        int* const s_map = reinterpret_cast<int*>(s_sort) + VL_CAP;   // cell -> pillar id (LDS copy of fmap)
        unsigned flags = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) if (c[k] >= 0 && f[k] == i0 + k) flags |= 1u << k;
        int totp;
        const int basep = block_excl_scan(__popc(flags), s_tmp, totp);
        {
            int r = 0;
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if ((flags >> k) & 1u) {
                    const int pid = basep + r++;
                    if (pid < max_voxels) {
                        fmap[c[k]] = pid;
                        if (first == nullptr) s_map[c[k]] = pid;
                        pillar_cell[(size_t)b * max_voxels + pid] = c[k];
                    } else if (pid == max_voxels) {
                        s_break = i0 + k;       // exactly one point has this prefix
                    }
                }
        }
        __syncthreads();                        // cell map of this frame + break point visible to the block
        V_STAMP()   // 2: pillar ids assigned, cell map written
        const int P = min(totp, max_voxels);
        const int ibreak = s_break;
        unsigned vmask = 0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) if (c[k] >= 0 && i0 + k < ibreak) vmask |= 1u << k;
        int key[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) key[k] = ((vmask >> k) & 1u) ? (first == nullptr ? s_map[c[k]] : fmap[c[k]]) : 0;
        int nv;
        const int basev = block_excl_scan(__popc(vmask), s_tmp, nv);
        {
            int r = 0;
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if ((vmask >> k) & 1u) s_sort[basev + r++] = ((unsigned)key[k] << ib) | (unsigned)(i0 + k);
        }
        __syncthreads();
        V_STAMP()   // 3: keys fetched, compaction done
        // stable LSD radix sort of the packed words by pillar id, LDS to LDS
        // in place: a pass holds every element in registers between its read and its scatter (barriers in
        // between), so source and destination are the same buffer
        unsigned* const sk = s_sort;
        unsigned* const dk = s_sort;
        const int NB = 1 << bits;
        const unsigned dmask = (unsigned)NB - 1u;
        const int chunk = ((nv + VWAVES * 64 - 1) / (VWAVES * 64)) * 64;
        const int wbeg = min(wave * chunk, nv), wend = min(wbeg + chunk, nv);
        // A wave sorts its contiguous chunk (<= PPT steps of 64 elements, held in registers for the pass).
        // Histogram step: the lanes that share a digit find each other by digit-bit ballots; each reads the
        // wave's running count of that digit (= how many earlier elements of the chunk carry it) and one of
        // them adds the group size -- conflict-free LDS operations that execute in issue order, so the 16
        // steps pipeline without a wait.  After the block-wide scan of the (digit, wave) counts the target
        // position of every element is known: no serial chain in the scatter.
        for (int pass = 0; pass < npass; ++pass) {
            const int shift = ib + pass * bits;
            for (int e = tid; e < NB * VWAVES; e += VT) s_hist[e] = 0;
            unsigned ev[PPT];
#pragma unroll
            for (int t = 0; t < PPT; ++t) {
                const int j = wbeg + t * 64 + lane;
                ev[t] = (j < wend) ? sk[j] : 0u;
            }
            __syncthreads();
            V_STAMP()   // pass: elements read, histogram cleared
            // (the peer set of a lane is kept as two 32-bit words and narrowed by one xnor + and per word and digit
            // bit: m = 0 / -1 from the lane's bit, peers &= ~(ballot ^ m); inactive lanes are outside the initial
            // mask and stay outside.  Count reads and group-leader adds are relaxed workgroup-scope LDS operations
            // (ds_read / ds_add in issue order); the counts read are consumed after the loop, so no step waits.)
            int rank[PPT], rbase[PPT];
            const unsigned lt_lo = (unsigned)lt, lt_hi = (unsigned)(lt >> 32);
#pragma unroll
            for (int t = 0; t < PPT; ++t) {
                rank[t] = 0;
                rbase[t] = 0;
                if (wbeg + t * 64 < wend) {                     // wave-uniform
                    const bool act = wbeg + t * 64 + lane < wend;
                    const unsigned d = (ev[t] >> shift) & dmask;
                    const unsigned long long am = __builtin_amdgcn_ballot_w64(act);
                    unsigned plo = (unsigned)am, phi = (unsigned)(am >> 32);
                    if (BITS > 0) {
#pragma unroll
                        for (int bit = 0; bit < BITS; ++bit) {
                            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)d, bit, 1);      // 0 / -1
                            const unsigned long long bb = __builtin_amdgcn_ballot_w64(__builtin_amdgcn_ubfe(d, bit, 1) != 0u);
                            plo &= ~((unsigned)bb ^ m);
                            phi &= ~((unsigned)(bb >> 32) ^ m);
                        }
                    } else {
                        for (int bit = 0; bit < bits; ++bit) {
                            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)d, bit, 1);      // 0 / -1
                            const unsigned long long bb = __builtin_amdgcn_ballot_w64(__builtin_amdgcn_ubfe(d, bit, 1) != 0u);
                            plo &= ~((unsigned)bb ^ m);
                            phi &= ~((unsigned)(bb >> 32) ^ m);
                        }
                    }
                    if (act) {
                        int* hp = &s_hist[wave * NB + d];   // [wave][digit]: the lanes of a wave spread over the banks
                        rbase[t] = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        const unsigned blo = plo & lt_lo, bhi = phi & lt_hi;
                        rank[t] = __popc(blo) + __popc(bhi);
                        if ((blo | bhi) == 0u)                  // lowest lane of the group
                            __hip_atomic_fetch_add(hp, __popc(plo) + __popc(phi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < PPT; ++t) rank[t] += rbase[t];
            __syncthreads();
            V_STAMP()   // pass: in-wave ranks + digit counts
            {
                const int E = NB * VWAVES;
                const int per = (E + VT - 1) / VT;
                const int e0 = tid * per;
                int local = 0;
                // scan order is digit-major, wave-minor (entry e = digit * VWAVES + wave), storage is [wave][digit]
                for (int q = 0; q < per; ++q)
                    if (e0 + q < E) local += s_hist[((e0 + q) % VWAVES) * NB + (e0 + q) / VWAVES];
                int total;
                int run = block_excl_scan(local, s_tmp, total);
                for (int q = 0; q < per; ++q)
                    if (e0 + q < E) {
                        int* hp = &s_hist[((e0 + q) % VWAVES) * NB + (e0 + q) / VWAVES];
                        const int t = *hp;
                        *hp = run;
                        run += t;
                    }
            }
            __syncthreads();
            V_STAMP()   // pass: block scan of the counts
#pragma unroll
            for (int t = 0; t < PPT; ++t) {
                if (wbeg + t * 64 + lane < wend) {
                    const unsigned d = (ev[t] >> shift) & dmask;
                    dk[s_hist[wave * NB + d] + rank[t]] = ev[t];
                }
            }
            __syncthreads();
            V_STAMP()   // 4, 5: sort passes
        }
        // sorted point indices + CSR row starts
        int* ps = pillar_start + (size_t)b * (max_voxels + 1);
        const unsigned imask = (1u << ib) - 1u;
        // (the pillar-sorted copy of the points -- the CSR payload the PFN streams with contiguous loads -- is
        // gathered by k_sort_points, next in the stream, on the whole chip: inside this one workgroup the 16 K
        // gathers of a frame are bound by a single CU's address path, 15 us of the frame's 60)
        unsigned* fin = ((npass & 1) ? idxB : idxA) + n0;     // sorted point indices (voxel_sort_passes)
#pragma unroll
        for (int t = 0; t < PPT; ++t) {
            const int j = tid + t * VT;
            if (j < nv) {
                const unsigned v = sk[j];
                fin[j] = v & imask;
                const unsigned k = v >> ib;
                if (j == 0 || (sk[j - 1] >> ib) != k) ps[k] = j;
            }
        }
        if (tid == 0) {
            ps[P] = nv;
            npillars[b] = P;
            nvalid_out[b] = nv;
        }
#ifdef PP_VOX_STAMPS
        __syncthreads();
        V_STAMP()
        if (tid == 0 && b == 0) {
            printf("vox n=%d nv=%d:", n, nv);
            for (int q = 1; q < vsn; ++q) printf(" %d", (int)(vst[q] - vst[q - 1]));
            printf("\n");
        }
#endif
        return;
    }

    // ---- A: pillar ids in first-appearance order, break point ----
    for (int base = 0; base < n; base += VT) {
        const int i = base + tid;
        const int c = (i < n) ? fcell[i] : -1;
        const bool flag = (c >= 0) && (ffirst[c] == i);
        const unsigned long long bal = __ballot(flag);
        if (lane == 0) s_tot[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tile_tot = 0;
#pragma unroll
        for (int w = 0; w < VWAVES; ++w) {
            int t = s_tot[w];
            if (w < wave) woff += t;
            tile_tot += t;
        }
        const int pid = s_carry + woff + __popcll(bal & lt);
        if (flag) {
            if (pid < max_voxels) {
                fmap[c] = pid;
                pillar_cell[(size_t)b * max_voxels + pid] = c;
            } else if (pid == max_voxels) {
                s_break = i;  // exactly one point has this prefix
            }
        }
        __syncthreads();
        if (tid == 0) s_carry += tile_tot;
    }
    __syncthreads();
    const int P = min(s_carry, max_voxels);
    const int ibreak = s_break;
    __syncthreads();

    // ---- B: stable compaction of surviving points -> (pillar id, point index) ----
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += VT) {
        const int i = base + tid;
        const int c = (i < n) ? fcell[i] : -1;
        const bool valid = (c >= 0) && (i < ibreak);
        const unsigned long long bal = __ballot(valid);
        if (lane == 0) s_tot[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, tile_tot = 0;
#pragma unroll
        for (int w = 0; w < VWAVES; ++w) {
            int t = s_tot[w];
            if (w < wave) woff += t;
            tile_tot += t;
        }
        if (valid) {
            const int pos = s_carry + woff + __popcll(bal & lt);
            kA[pos] = (unsigned)fmap[c];
            vA[pos] = (unsigned)i;
        }
        __syncthreads();
        if (tid == 0) s_carry += tile_tot;
    }
    __syncthreads();
    const int nv = s_carry;
    __syncthreads();

    // ---- C: stable LSD radix sort by pillar id ----
    unsigned* sk = kA; unsigned* sv = vA; unsigned* dk = kB; unsigned* dv = vB;
    const int NB = 1 << bits;
    const unsigned dmask = (unsigned)NB - 1u;
    const int chunk = ((nv + VWAVES * 64 - 1) / (VWAVES * 64)) * 64;  // per-wave contiguous range
    const int wbeg = min(wave * chunk, nv);
    const int wend = min(wbeg + chunk, nv);
    volatile int* vhist = s_hist;
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = pass * bits;
        for (int e = tid; e < NB * VWAVES; e += VT) s_hist[e] = 0;
        __syncthreads();
        for (int t0 = wbeg; t0 < wend; t0 += 64) {
            const int j = t0 + lane;
            if (j < wend) atomicAdd(&s_hist[wave * NB + ((sk[j] >> shift) & dmask)], 1);   // [wave][digit]: banks by digit
        }
        __syncthreads();
        {   // exclusive scan over (digit major, wave minor): entry e = digit * VWAVES + wave, stored at [wave][digit]
            const int E = NB * VWAVES;
            const int per = (E + VT - 1) / VT;
            const int e0 = tid * per;
            int local = 0;
            for (int q = 0; q < per; ++q)
                if (e0 + q < E) local += s_hist[((e0 + q) % VWAVES) * NB + (e0 + q) / VWAVES];
            int total;
            int run = block_excl_scan(local, s_tmp, total);
            for (int q = 0; q < per; ++q)
                if (e0 + q < E) {
                    int* hp = &s_hist[((e0 + q) % VWAVES) * NB + (e0 + q) / VWAVES];
                    const int t = *hp;
                    *hp = run;
                    run += t;
                }
        }
        __syncthreads();
        for (int t0 = wbeg; t0 < wend; t0 += 64) {
            const int j = t0 + lane;
            const bool act = j < wend;
            const unsigned k = act ? sk[j] : 0u;
            const unsigned v = act ? sv[j] : 0u;
            const unsigned d = (k >> shift) & dmask;
            unsigned long long peers = __ballot(act);
            for (int bit = 0; bit < bits; ++bit) {
                const bool one = (d >> bit) & 1u;
                const unsigned long long bb = __ballot(act && one);
                peers &= one ? bb : ~bb;
            }
            if (act) {
                const int basep = vhist[wave * NB + d];
                const int pos = basep + __popcll(peers & lt);
                dk[pos] = k;
                dv[pos] = v;
                if (lane == 63 - __clzll(peers)) vhist[wave * NB + d] = basep + __popcll(peers);
            }
        }
        __syncthreads();
        unsigned* t;
        t = sk; sk = dk; dk = t;
        t = sv; sv = dv; dv = t;
    }

    // ---- D: CSR row starts from the key boundaries ----
    int* ps = pillar_start + (size_t)b * (max_voxels + 1);
    for (int j = tid; j < nv; j += VT) {
        const unsigned k = sk[j];
        if (j == 0 || sk[j - 1] != k) ps[k] = j;
    }
    if (tid == 0) {
        ps[P] = nv;
        npillars[b] = P;
        nvalid_out[b] = nv;
    }
}

// pillar-sorted copy of the points: pts_sorted[n0 + j] = pts[n0 + idx[n0 + j]] for the nvalid[b] surviving points of
// every frame (one thread per point, the whole chip; 16-byte accesses for F == 4, one 4-byte access per feature else)
__global__ __launch_bounds__(256) void k_sort_points(const float* __restrict__ pts, const int* __restrict__ offsets,
                                                     const unsigned* __restrict__ idx, const int* __restrict__ nvalid,
                                                     int F, float* __restrict__ pts_sorted, int batch, int nchunk) {
    // consecutive workgroup ids go to consecutive XCDs: all chunks of a frame on ONE XCD (frame = 8 * group + xcd), so a
    // frame's 16-K random 12-byte reads hit lines that XCD's L2 already holds instead of fetching them once per XCD
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    const int chunk = r % nchunk, b = (r / nchunk) * 8 + xcd;
    if (b >= batch) return;
    const int j = chunk * 256 + threadIdx.x;
    if (j >= nvalid[b]) return;
    const int n0 = offsets[b];
    const unsigned src = idx[n0 + j];
    const float* fp = pts + (size_t)(n0 + src) * F;
    float* fs = pts_sorted + (size_t)(n0 + j) * F;
    if (F == 4) {
        *reinterpret_cast<float4*>(fs) = *reinterpret_cast<const float4*>(fp);
    } else {
        const float x = fp[0], y = fp[1], z = fp[2];
        fs[0] = x; fs[1] = y; fs[2] = z;
    }
}

// compat: the reference's padded outputs for ONE frame (load_data.py:757-771)
__global__ __launch_bounds__(64) void k_voxel_expand(const float* __restrict__ pts_sorted,
                                                     const int* __restrict__ offsets,
                                                     const unsigned* __restrict__ sorted_idx,
                                                     const int* __restrict__ pillar_start,
                                                     const int* __restrict__ pillar_cell,
                                                     const int* __restrict__ npillars, int frame, int F, int T,
                                                     int max_voxels, int ny, int nx, float* __restrict__ voxels,
                                                     int* __restrict__ coors, int* __restrict__ num_points) {
    const int p = blockIdx.x;
    if (p >= npillars[frame]) return;
    const int n0 = offsets[frame];
    const int* ps = pillar_start + (size_t)frame * (max_voxels + 1);
    const int start = ps[p];
    const int cnt = min(ps[p + 1] - start, T);
    (void)sorted_idx;
    for (int e = threadIdx.x; e < T * F; e += 64) {
        const int s = e / F, f = e - s * F;
        float v = 0.f;
        if (s < cnt) v = pts_sorted[(size_t)(n0 + start + s) * F + f];
        voxels[((size_t)p * T) * F + e] = v;
    }
    if (threadIdx.x == 0) {
        const int c = pillar_cell[(size_t)frame * max_voxels + p];
        const int x = c % nx, y = (c / nx) % ny, z = c / (nx * ny);
        coors[p * 3 + 0] = z;
        coors[p * 3 + 1] = y;
        coors[p * 3 + 2] = x;
        num_points[p] = cnt;
    }
}

// compat: cell -> pillar map from batched coors [P,4] (b z y x); validated on the host
__global__ __launch_bounds__(256) void k_build_cellmap(const int* __restrict__ coors4, int64_t P, int ncell,
                                                       int ny, int nx, int* __restrict__ cellmap) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int b = coors4[p * 4 + 0], z = coors4[p * 4 + 1], y = coors4[p * 4 + 2], x = coors4[p * 4 + 3];
    cellmap[(size_t)b * ncell + ((size_t)z * ny + y) * nx + x] = (int)p;
}

// true when every frame of the batch takes k_voxel_frame's LDS path and its cell table fits LDS: the global
// first-index table (memset + atomics) is then not needed at all
bool voxel_first_in_lds(int max_n, int ncell, int max_voxels) {
    int kb = 1, ib = 1;
    while ((1 << kb) < max_voxels) ++kb;
    while ((1 << ib) < max_n) ++ib;
    const int npass = (kb + 7) / 8;
    const int bits = (kb + npass - 1) / npass;
    return max_n <= VL_CAP && ncell <= VL_CAP && npass * bits + ib <= 32;
}

void launch_cell_first(const float* pts, const int* offsets, int batch, int max_n, int F, const VoxGeom& g,
                       int* cell, int* first, int* cellmap, const PpFeed* feed, float* pts_dst, int* offsets_dst,
                       hipStream_t s, unsigned long long* occbits, int occ_n) {
    if (batch <= 0) return;
    dim3 grid(max_n > 0 ? (max_n + 255) / 256 : 1, batch);   // at least one block per frame: it clears the cell map
    PP_LAUNCH("k_cell_first", k_cell_first, grid, dim3(256), 0, s, pts, offsets, F, g, cell, first, cellmap, feed, pts_dst,
              offsets_dst, occbits, occ_n);
}

void launch_voxel_frame(const int* offsets, const int* cell, const int* first, int* cellmap, unsigned* keyA,
                        unsigned* idxA, unsigned* keyB, unsigned* idxB, int* pillar_start, int* pillar_cell,
                        int* npillars, int* nvalid, int batch, int max_n, int ncell, int max_voxels, hipStream_t s) {
    if (batch <= 0) return;
    int kb = 1;
    while ((1 << kb) < max_voxels) ++kb;
    const int npass = (kb + 7) / 8;
    const int bits = (kb + npass - 1) / npass;
    // frames of more than 16 384 points (KITTI-sized clouds) run the 32-points-per-thread instantiation
    const bool big = max_n > VL_CAP;
#define VOX_LAUNCH(B_, P_)                                                                                            \
    PP_LAUNCH("k_voxel_frame", (k_voxel_frame<B_, P_>), dim3(batch), dim3(VT), 0, s, offsets, cell, first, cellmap, keyA, \
              idxA, keyB, idxB, pillar_start, pillar_cell, npillars, nvalid, ncell, max_voxels, npass, bits)
    if (bits == 7) {   // 8193..16384 pillars (the shipped configuration): unrolled digit loops
        if (big) VOX_LAUNCH(7, 32); else VOX_LAUNCH(7, 16);
    } else {
        if (big) VOX_LAUNCH(0, 32); else VOX_LAUNCH(0, 16);
    }
#undef VOX_LAUNCH
}

void launch_sort_points(const float* pts, const int* offsets, const unsigned* sorted_idx, const int* nvalid, int batch,
                        int max_n, int F, float* pts_sorted, hipStream_t s) {
    if (batch <= 0 || max_n <= 0) return;
    const int nchunk = (max_n + 255) / 256;
    PP_LAUNCH("k_sort_points", k_sort_points, dim3(8 * nchunk * ((batch + 7) / 8)), dim3(256), 0, s, pts, offsets,
              sorted_idx, nvalid, F, pts_sorted, batch, nchunk);
}

void launch_voxel_expand(const float* pts, const int* offsets, const unsigned* sorted_idx, const int* pillar_start,
                         const int* pillar_cell, const int* npillars, int frame, int F, int T, int max_voxels,
                         int ny, int nx, float* voxels, int* coors, int* num_points, hipStream_t s) {
    PP_LAUNCH("k_voxel_expand", k_voxel_expand, dim3(max_voxels), dim3(64), 0, s, pts, offsets, sorted_idx, pillar_start,
                       pillar_cell, npillars, frame, F, T, max_voxels, ny, nx, voxels, coors, num_points);
}

void launch_build_cellmap(const int* coors4, int64_t P, int ncell, int ny, int nx, int* cellmap, hipStream_t s) {
    if (P <= 0) return;
    PP_LAUNCH("k_build_cellmap", k_build_cellmap, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, coors4, P, ncell, ny,
                       nx, cellmap);
}
