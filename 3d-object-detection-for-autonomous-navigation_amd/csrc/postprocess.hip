// Detection post-process: mask -> top-100 -> decode -> stand-up AABB -> NMS -> flip -> camera.
//
// Replaces VoxelNet.predict for one frame per workgroup (reference
// model/voxelnet.py:1105-1379) and the functions it calls:
//   anchors_mask gather, argmax(dir), sigmoid           model/voxelnet.py:1120-1150
//   top-100 by score (np.argpartition)                  model/voxelnet.py:1207-1214
//   second_box_decode                                   libraries/eval_helper_functions.py:388-461
//   center_to_corner_box2d + corner_to_standup_nd_jit   load_data.py:1525-1593, :1330-1340
//   nms -> nms_gpu -> nms_kernel / iou_device / nms_postprocess
//                                                       libraries/eval_helper_functions.py:463-598
//   direction flip, box_lidar_to_camera                 model/voxelnet.py:1305-1310; eval_helper_functions.py:728-740
// The reference pulls the head maps to the host and bounces the <=100 boxes to
// the GPU and back for a numba NMS kernel (1.4-2.2 ms per call); here the head
// maps never leave HBM and the whole tail is one launch.
//
// Selection is done on the logit (sigmoid is monotone), with an order-preserving
// integer key and the anchor index as tie-break (lower index first), i.e. a
// deterministic refinement of np.argpartition / argsort whose tie order is
// implementation-defined.  NMS IoU follows iou_device exactly: AABB with `+1`
// on widths (pixel convention applied to metres), differences in float32, the
// rest in float64, strict `>` threshold.
#include <type_traits>

#include "pp_common.h"

#define PT 1024
#define KMAX 128   // >= the reference's hard-coded top-100
#define CCAP 12288 // candidate keys kept in LDS (96 KB); more candidates -> re-read the head map per pass

__device__ __forceinline__ unsigned long long comp_key(float logit, unsigned a) {
    unsigned u = __float_as_uint(logit);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(~a);
}

__global__ __launch_bounds__(PT) void k_postprocess(PostParams p) {
    __shared__ int s_hist[256];
    __shared__ int s_cum[256];
    __shared__ int s_wsum[4];
    __shared__ unsigned long long s_ckey[CCAP];   // compacted candidate keys (order irrelevant: keys are unique)
    __shared__ int s_ncand;
    __shared__ unsigned long long s_prefix;
    __shared__ int s_need, s_shift, s_done, s_cnt;
    __shared__ unsigned long long s_key[KMAX];
    __shared__ int s_order[KMAX];
    __shared__ float s_box[KMAX][7];
    __shared__ float s_aabb[KMAX][4];
    __shared__ float s_score[KMAX];
    __shared__ int s_dir[KMAX];
    __shared__ int s_label[KMAX];
    __shared__ int s_anchor[KMAX];
    __shared__ unsigned long long s_mask[KMAX][2];
    __shared__ int s_keep[KMAX];
    __shared__ int s_nkeep;
    __shared__ int s_bad;      // a non-finite class logit anywhere in the frame, or a non-finite value in a selected row

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
#ifdef PP_POST_STAMPS   // diagnostic build: phase times of frame 0 (100 MHz ticks), printed by thread 0
    long long pst[24];
    int psn = 0;
#define P_STAMP() { if (psn < 24) pst[psn++] = wall_clock64(); }
#else
#define P_STAMP() {}
#endif
    P_STAMP()
    const long long A = p.A;
    const int napl = p.napl;
    const int ncls = p.ncls;
    const int nb = napl * 7, nc = napl * ncls;
    // fused head map row of pixel px: [box napl*7 | cls napl*ncls | dir napl*2 | pad]; anchor a = px*napl + r,
    // class logit k of that anchor at column nb + r*ncls + k (the reference reshapes cls_preds to
    // [B, -1, num_class], model/voxelnet.py:1090).  More than one class: the anchor's score is its largest
    // class score and its label the argmax (model/voxelnet.py:1183-1185; sigmoid is monotone, so the
    // selection runs on the largest logit)
    const float* head = p.head + (size_t)b * (A / napl) * PP_HEAD_COLS;
    const uint8_t* msk = p.mask + (size_t)b * A;
    // candidate scan source: the compact class-logit plane the last deconv left (napl*ncls floats per pixel,
    // consecutive anchors = consecutive words) or, without it, the cls columns of the 128-byte head rows
    const float* cplane = (p.cls != nullptr) ? p.cls + (size_t)b * A * ncls : nullptr;
    // Non-finite head outputs (the split-precision GEMMs carry an operand as two float16 pieces: an activation beyond
    // +-65504 becomes inf - inf = NaN in the product; the ReLUs of the backbone keep a NaN, relu_keep_nan) must not
    // turn into boxes: every class logit of the frame passes through here, masked or not, and the frame is flagged.
    bool bad = false;
    auto cls_of = [&](long long a) -> float {
        const long long px = a / napl;
        const float* q = (cplane != nullptr) ? cplane + a * ncls
                                             : head + px * PP_HEAD_COLS + nb + (int)(a - px * napl) * ncls;
        float m = q[0];
        bad |= !pp_finite(m);
        for (int k = 1; k < ncls; ++k) { bad |= !pp_finite(q[k]); m = fmaxf(m, q[k]); }
        return m;
    };
    const float thr = p.score_thr;
    const int KTOP = 100;  // model/voxelnet.py:1207 (hard-coded)

    auto is_cand = [&](long long a, float& logit) -> bool {
        if (msk[a] != 1) return false;
        logit = cls_of(a);
        if (thr > 0.f) {
            const float sc = 1.f / (1.f + expf(-logit));
            if (!(sc >= thr)) return false;
        }
        return true;
    };

    // ---- candidates -> LDS once (the head map is read a single time; the select passes run on LDS) ----
    if (tid == 0) { s_cnt = 0; s_ncand = 0; s_bad = 0; }
    __syncthreads();
    // One pass over the frame's anchors: emit(anchor, logit) for every candidate (mask byte 1, score >= threshold).
    //   * compact class-logit plane and A % 16 == 0 (every shipped grid): a thread takes 16 consecutive anchors -- one
    //     16-byte load of their mask bytes, 4 * ncls 16-byte loads of their logits, all issued before the first use: 9
    //     loads per 16 anchors on the KITTI-shaped head (two classes) instead of 48 (the scan of 107 136 anchors by one
    //     workgroup was 110 us per launch of 32 frames, load-issue bound);
    //   * otherwise 16 strided anchors per thread at a time (the logit loads do not wait for the mask bytes).
    auto scan_candidates = [&](auto&& emit) {
        if (cplane != nullptr && (A & 15) == 0 && ncls <= 4) {
            const uint4* m16 = reinterpret_cast<const uint4*>(msk);
            const float4* l4 = reinterpret_cast<const float4*>(cplane);
            auto fast = [&](auto NC) {                                // the class count as a compile-time constant: the
                constexpr int nc = decltype(NC)::value;               // 16 * nc logits of a thread stay in registers
                for (long long c = tid; c < (A >> 4); c += PT) {
                    const uint4 mk4 = m16[c];
                    float4 lg[4 * nc];
#pragma unroll
                    for (int j = 0; j < 4 * nc; ++j) lg[j] = l4[c * (4 * nc) + j];
                    const unsigned mw[4] = {mk4.x, mk4.y, mk4.z, mk4.w};
                    float lf[16 * nc];
#pragma unroll
                    for (int j = 0; j < 4 * nc; ++j) { lf[4 * j] = lg[j].x; lf[4 * j + 1] = lg[j].y; lf[4 * j + 2] = lg[j].z; lf[4 * j + 3] = lg[j].w; }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float m = lf[i * nc];
                        bad |= !pp_finite(m);
#pragma unroll
                        for (int k = 1; k < nc; ++k) { const float v = lf[i * nc + k]; bad |= !pp_finite(v); m = fmaxf(m, v); }
                        if (((mw[i >> 2] >> (8 * (i & 3))) & 0xffu) != 1u) continue;
                        if (thr > 0.f) {
                            const float sc = 1.f / (1.f + expf(-m));
                            if (!(sc >= thr)) continue;
                        }
                        emit((unsigned)(c * 16 + i), m);
                    }
                }
            };
            if (ncls == 1) fast(std::integral_constant<int, 1>{});
            else if (ncls == 2) fast(std::integral_constant<int, 2>{});
            else if (ncls == 3) fast(std::integral_constant<int, 3>{});
            else fast(std::integral_constant<int, 4>{});
            return;
        }
        for (long long ab = 0; ab < A; ab += 16 * PT) {
            const long long a0 = ab + tid;
            uint8_t mk[16];
            float lgs[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long long a = a0 + (long long)k * PT;
                mk[k] = (a < A) ? msk[a] : (uint8_t)0;
                lgs[k] = (a < A) ? cls_of(a) : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (mk[k] != 1) continue;
                if (thr > 0.f) {
                    const float sc = 1.f / (1.f + expf(-lgs[k]));
                    if (!(sc >= thr)) continue;
                }
                emit((unsigned)(a0 + (long long)k * PT), lgs[k]);
            }
        }
    };
    scan_candidates([&](unsigned a, float lg) {
        const int pos = atomicAdd(&s_ncand, 1);
        if (pos < CCAP) s_ckey[pos] = comp_key(lg, a);
    });
    if (bad) s_bad = 1;
    __syncthreads();
    P_STAMP()   // candidates gathered
    int ncand = s_ncand;
    bool in_lds = ncand <= CCAP;

    // ---- radix select of the KTOP largest composite keys (8 bits per pass, MSB first) ----
    // over the `n` keys in LDS (lds) or over the frame's candidates re-read from the head map
    auto run_select = [&](bool lds, int n) {
      if (tid == 0) { s_prefix = 0ull; s_need = KTOP; s_shift = 56; s_done = 0; }
      __syncthreads();
      for (int pass = 0; pass < 8; ++pass) {
        if (tid < 256) s_hist[tid] = 0;
        __syncthreads();
        if (s_done) break;
        const int shift = 56 - 8 * pass;
        const unsigned long long prefix = s_prefix;
        if (lds) {
            for (int i = tid; i < n; i += PT) {
                const unsigned long long key = s_ckey[i];
                if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&s_hist[(int)((key >> shift) & 255ull)], 1);
            }
        } else {
            for (long long a = tid; a < A; a += PT) {
                float lg;
                if (!is_cand(a, lg)) continue;
                const unsigned long long key = comp_key(lg, (unsigned)a);
                if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&s_hist[(int)((key >> shift) & 255ull)], 1);
            }
        }
        __syncthreads();
        // digit of the need-th largest key: suffix sums over the 256 bins by 256 threads (bin 255 - tid)
        if (tid < 256) {
            const int lane_ = tid & 63, wv = tid >> 6;
            const int hv = s_hist[255 - tid];
            const int incl = wave_inclusive_scan(hv);
            if (lane_ == 63) s_wsum[wv] = incl;
            s_cum[tid] = incl;   // within-wave inclusive sums; the wave offsets are added by thread 0 below
        }
        __syncthreads();
        if (tid == 0) {
            int need = s_need, cum = 0, digit = 0;
            bool found = false;
            int woff = 0;
            for (int wv = 0; wv < 4 && !found; ++wv) {
                if (woff + s_wsum[wv] >= need) {   // the bin is inside this wave's 64: binary search on its inclusive sums
                    int lo = 0, hi = 63;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (woff + s_cum[wv * 64 + mid] >= need) hi = mid; else lo = mid + 1;
                    }
                    const int t = wv * 64 + lo;
                    digit = 255 - t;
                    cum = woff + s_cum[t] - s_hist[digit];
                    need -= cum;
                    found = true;
                }
                woff += s_wsum[wv];
            }
            if (!found) {
                // fewer candidates than requested (only possible in pass 0): take them all
                s_prefix = 0ull; s_shift = 0; s_done = 1;
            } else {
                s_prefix = (prefix << 8) | (unsigned long long)digit;
                s_shift = shift;
                s_need = need;
                if (s_hist[digit] == need) s_done = 1;  // the whole bucket is taken
            }
        }
        __syncthreads();
        P_STAMP()   // one select pass
      }
      __syncthreads();
    };
    if (in_lds) {
        run_select(true, ncand);
    } else {
        // More candidates than LDS holds (large grids: 107 k anchors on the KITTI-shaped map): the 100th largest of the
        // FIRST CCAP candidates is a lower bound of the 100th largest of all of them, so one more read of the head map
        // keeps only the keys at or above it -- a few hundred for candidates in no particular order -- and the select
        // runs on those in LDS (round 3 re-read the head map in every one of up to nine passes: 92 us per frame).
        run_select(true, CCAP);
        const unsigned long long floor_key = s_prefix << s_shift;     // every key of that top set is >= this
        __syncthreads();
        if (tid == 0) s_ncand = 0;
        __syncthreads();
        scan_candidates([&](unsigned a, float lg) {
            const unsigned long long key = comp_key(lg, a);
            if (key >= floor_key) {
                const int pos = atomicAdd(&s_ncand, 1);
                if (pos < CCAP) s_ckey[pos] = key;
            }
        });
        __syncthreads();
        const int kept = s_ncand;
        if (kept <= CCAP) { in_lds = true; ncand = kept; run_select(true, kept); }
        else run_select(false, 0);                                    // (adversarial order: the old way)
    }
    __syncthreads();
    P_STAMP()
    {
        const int shift = s_shift;
        const unsigned long long prefix = s_prefix;
        if (in_lds) {
            for (int i = tid; i < ncand; i += PT) {
                const unsigned long long key = s_ckey[i];
                if ((key >> shift) >= prefix) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < KMAX) s_key[pos] = key;
                }
            }
        } else {
            for (long long a = tid; a < A; a += PT) {
                float lg;
                if (!is_cand(a, lg)) continue;
                const unsigned long long key = comp_key(lg, (unsigned)a);
                if ((key >> shift) >= prefix) {
                    const int pos = atomicAdd(&s_cnt, 1);
                    if (pos < KMAX) s_key[pos] = key;
                }
            }
        }
    }
    __syncthreads();
    P_STAMP()   // selected keys collected
    const int K = min(s_cnt, KTOP);
    // ---- order by descending key (rank by counting; keys are unique) ----
    if (tid < K) {
        const unsigned long long me = s_key[tid];
        int rank = 0;
        for (int j = 0; j < K; ++j) rank += (s_key[j] > me) ? 1 : 0;
        s_order[rank] = tid;
    }
    __syncthreads();
    P_STAMP()   // ordered

    // ---- decode + stand-up AABB (float32, the reference's operation order) ----
    if (tid < K) {
        const unsigned long long key = s_key[s_order[tid]];
        const unsigned a = ~(unsigned)(key & 0xffffffffull);
        const unsigned apx = a / (unsigned)napl, ar = a - apx * (unsigned)napl;
        const float* hrow = head + (size_t)apx * PP_HEAD_COLS;
        const float* e = hrow + ar * 7;
        const float* an = p.anchors + (size_t)a * 7;
        const float xa = an[0], ya = an[1], wa = an[3], la = an[4], ha = an[5], ra = an[6];
        const float za = __fadd_rn(an[2], __fdiv_rn(ha, 2.f));
        const float diag = __fsqrt_rn(__fadd_rn(__fmul_rn(la, la), __fmul_rn(wa, wa)));
        const float xg = __fadd_rn(__fmul_rn(e[0], diag), xa);
        const float yg = __fadd_rn(__fmul_rn(e[1], diag), ya);
        float zg = __fadd_rn(__fmul_rn(e[2], ha), za);
        const float lg = __fmul_rn(expf(e[4]), la);
        const float wg = __fmul_rn(expf(e[3]), wa);
        const float hg = __fmul_rn(expf(e[5]), ha);
        const float rg = __fadd_rn(e[6], ra);
        zg = __fsub_rn(zg, __fdiv_rn(hg, 2.f));
        s_box[tid][0] = xg; s_box[tid][1] = yg; s_box[tid][2] = zg;
        s_box[tid][3] = wg; s_box[tid][4] = lg; s_box[tid][5] = hg; s_box[tid][6] = rg;
        float lgt = hrow[nb + ar * ncls];
        int lab = 0;
        for (int k = 1; k < ncls; ++k) {
            const float v = hrow[nb + ar * ncls + k];
            if (v > lgt) { lgt = v; lab = k; }   // argmax: first maximum
        }
        s_label[tid] = lab;
        s_score[tid] = 1.f / (1.f + expf(-lgt));
        const float* d = hrow + nb + nc + ar * 2;
        s_dir[tid] = (p.use_dir && d[1] > d[0]) ? 1 : 0;  // np.argmax: first maximum
        s_anchor[tid] = (int)a;
        {
            bool rb = !pp_finite(d[0]) || !pp_finite(d[1]);
#pragma unroll
            for (int q = 0; q < 7; ++q) rb |= !pp_finite(e[q]);
            if (rb) s_bad = 1;      // (the class logits were checked by the scan)
        }
        // corners (-,-),(-,+),(+,+),(+,-) * dims, rotate by [[c,-s],[s,c]], + centre; min/max
        const float sn = sinf(rg), cs = cosf(rg);
        const float hx = __fmul_rn(wg, 0.5f), hy = __fmul_rn(lg, 0.5f);
        const float cxs[4] = {-hx, -hx, hx, hx};
        const float cys[4] = {-hy, hy, hy, -hy};
        float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float rx = __fadd_rn(__fadd_rn(__fmul_rn(cxs[q], cs), __fmul_rn(cys[q], sn)), xg);
            const float ry = __fadd_rn(__fadd_rn(__fmul_rn(cxs[q], -sn), __fmul_rn(cys[q], cs)), yg);
            if (q == 0) { x0 = x1 = rx; y0 = y1 = ry; }
            else { x0 = fminf(x0, rx); x1 = fmaxf(x1, rx); y0 = fminf(y0, ry); y1 = fmaxf(y1, ry); }
        }
        s_aabb[tid][0] = x0; s_aabb[tid][1] = y0; s_aabb[tid][2] = x1; s_aabb[tid][3] = y1;
    }
    __syncthreads();
    P_STAMP()   // decoded

    // ---- NMS over the first min(K, pre_max) boxes (already score-descending) ----
    // suppression masks: 8 threads share a row i and take every 8th j > i (the float64 division keeps
    // iou_device's arithmetic); their partial masks meet in LDS
    const int n = min(K, p.pre_max);
    if (tid < 2 * KMAX) reinterpret_cast<unsigned long long*>(s_mask)[tid] = 0ull;
    __syncthreads();
    {
        const int row = tid >> 3, sub = tid & 7;
        if (row < n) {
            unsigned long long m0 = 0ull, m1 = 0ull;
            const float ax0 = s_aabb[row][0], ay0 = s_aabb[row][1], ax1 = s_aabb[row][2], ay1 = s_aabb[row][3];
            const double sa = ((double)(ax1 - ax0) + 1.0) * ((double)(ay1 - ay0) + 1.0);
            const double dthr = (double)p.iou_thr;
            for (int j = row + 1 + sub; j < n; j += 8) {
                const float bx0 = s_aabb[j][0], by0 = s_aabb[j][1], bx1 = s_aabb[j][2], by1 = s_aabb[j][3];
                const float left = fmaxf(ax0, bx0), right = fminf(ax1, bx1);
                const float top = fmaxf(ay0, by0), bottom = fminf(ay1, by1);
                const float dw = right - left, dh = bottom - top;   // float32 differences
                const double w = fmax((double)dw + 1.0, 0.0);
                const double hh = fmax((double)dh + 1.0, 0.0);
                const double inter = w * hh;
                const double sb = ((double)(bx1 - bx0) + 1.0) * ((double)(by1 - by0) + 1.0);
                const double iou = inter / (sa + sb - inter);
                if (iou > dthr) {
                    if (j < 64) m0 |= 1ull << j; else m1 |= 1ull << (j - 64);
                }
            }
            if (m0) atomicOr(&s_mask[row][0], m0);
            if (m1) atomicOr(&s_mask[row][1], m1);
        }
    }
    __syncthreads();
    P_STAMP()   // pair masks
    // greedy sweep by wave 0: lane l holds the mask rows l and l + 64 in registers; the kept boxes are walked
    // with find-first-set over "not yet visited and not removed" (scalar), a kept row's mask comes by readlane
    if (tid < 64) {
        const unsigned long long a0 = (lane < n) ? s_mask[lane][0] : 0ull, a1 = (lane < n) ? s_mask[lane][1] : 0ull;
        const unsigned long long b0 = (lane + 64 < n) ? s_mask[lane + 64][0] : 0ull;
        const unsigned long long b1 = (lane + 64 < n) ? s_mask[lane + 64][1] : 0ull;
        auto rdl = [](unsigned long long v, int l) -> unsigned long long {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
            return ((unsigned long long)hi << 32) | lo;
        };
        unsigned long long todo0 = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
        unsigned long long todo1 = (n <= 64) ? 0ull : ((n - 64 >= 64) ? ~0ull : ((1ull << (n - 64)) - 1ull));
        unsigned long long r0 = 0ull, r1 = 0ull;
        int nk = 0, k0 = 0, k1 = 0;
        while (nk < p.post_max) {
            const unsigned long long c0 = todo0 & ~r0, c1 = todo1 & ~r1;
            if ((c0 | c1) == 0ull) break;
            int i = c0 ? __builtin_ctzll(c0) : 64 + __builtin_ctzll(c1);
            i = __builtin_amdgcn_readfirstlane(i);
            if (lane == (nk & 63)) { if (nk < 64) k0 = i; else k1 = i; }   // lane l remembers kept boxes l and l + 64
            ++nk;
            if (i < 64) {
                todo0 &= ~((2ull << i) - 1ull);       // i = 63: 2 << 63 wraps to 0, minus 1 = all ones
                r0 |= rdl(a0, i); r1 |= rdl(a1, i);
            } else {
                todo0 = 0ull;
                todo1 &= ~((2ull << (i - 64)) - 1ull);
                r0 |= rdl(b0, i - 64); r1 |= rdl(b1, i - 64);
            }
        }
        if (lane < nk) s_keep[lane] = k0;
        if (lane + 64 < nk) s_keep[lane + 64] = k1;
        if (lane == 0) {
            s_nkeep = nk;
            const int flagged = nk | (s_bad ? PP_NDETS_NONFINITE : 0);   // pp_get_detections / pp_predict: PP_ERR_NUMERIC
            p.n_dets[b] = flagged;
            if (p.n_dets_host != nullptr) p.n_dets_host[b] = flagged;
        }
    }
    __syncthreads();
    P_STAMP()   // sweep

    // ---- direction flip + lidar -> camera, in keep (descending score) order ----
    const int nk = s_nkeep;
    if (tid < nk) {
        const int i = s_keep[tid];
        pp_detection d;
        float r = s_box[i][6];
        // model/voxelnet.py:1297-1310: the flip exists only with use_direction_classifier
        const bool opp = p.use_dir && ((r > 0.f) != (s_dir[i] == 1));
        if (p.use_dir) r = (float)((double)r + (opp ? 3.141592653589793 : 0.0));  // f32 += f64 (numpy in-place add)
        const float* M = p.calib + (size_t)b * 16;
        const double x = s_box[i][0], y = s_box[i][1], z = s_box[i][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            d.box3d_camera[q] = x * (double)M[q * 4 + 0] + y * (double)M[q * 4 + 1] + z * (double)M[q * 4 + 2] +
                                (double)M[q * 4 + 3];
        d.box3d_camera[3] = (double)s_box[i][4];  // l
        d.box3d_camera[4] = (double)s_box[i][5];  // h
        d.box3d_camera[5] = (double)s_box[i][3];  // w
        d.box3d_camera[6] = (double)r;
#pragma unroll
        for (int q = 0; q < 6; ++q) d.box3d_lidar[q] = s_box[i][q];
        d.box3d_lidar[6] = r;
        d.score = s_score[i];
        d.label = s_label[i];
        d.dir_label = s_dir[i];
        d.anchor_index = s_anchor[i];
        d.reserved = 0;
        p.dets[(size_t)b * p.post_max + tid] = d;
        if (p.dets_host != nullptr) p.dets_host[(size_t)b * p.post_max + tid] = d;
    }
#ifdef PP_POST_STAMPS
    __syncthreads();
    P_STAMP()
    if (tid == 0 && b == 0) {
        printf("post ncand=%d K=%d nk=%d:", ncand, K, nk);
        for (int q = 1; q < psn; ++q) printf(" %d", (int)(pst[q] - pst[q - 1]));
        printf("\n");
    }
#endif
}

void launch_postprocess(const PostParams& p, hipStream_t s) {
    if (p.batch <= 0) return;
    PP_LAUNCH("k_postprocess", k_postprocess, dim3(p.batch), dim3(PT), 0, s, p);
}
