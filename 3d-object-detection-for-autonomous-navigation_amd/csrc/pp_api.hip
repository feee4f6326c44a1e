// C-ABI of libpp_hip.so (include/pp_hip.h): engine lifetime, weight folding, device
// workspaces, the stage pipelines and the measurement hooks.  Host side only; the
// kernels live in voxelize.hip / pfn.hip / anchor_mask.hip / backbone.hip /
// postprocess.hip.  Everything runs on one HIP stream owned by the handle.
#include <math.h>
#include <algorithm>
#include <mutex>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unordered_map>

#include "pp_common.h"
#include "train.h"

extern int g_num_cus;   // backbone.hip: CU count for persistent launches

namespace {

std::string g_create_error;

struct KTime { const char* name; int ev; };

}  // namespace

thread_local PpProf g_pp_prof = {nullptr, nullptr};

struct pp_engine {
    pp_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    VoxGeom geom;
    int nx = 0, ny = 0, nz = 0, ncell = 0;
    int head_h = 0, head_w = 0, napl = 0, ncls = 1;
    bool use_dir = true, with_dist = false;
    int64_t A = 0;
    int C = 0, F = 0, T = 0, FA = 0, CC = 0;
    int B = 0, NMAX = 0;

    std::map<std::string, std::vector<float>> hw;
    std::map<std::string, std::vector<int64_t>> hshape;
    bool weights_ready = false, anchors_ready = false;
    std::vector<void*> allocs;
    std::vector<void*> wallocs;   // weight buffers: released and re-made by every pp_finalize_weights

    // raw points and frame offsets are double-buffered: an upload fills the buffer the previous pass is NOT
    // reading (pp_upload_points_async on the copy stream, so the copy of batch k+1 runs beside the kernels of
    // batch k); d_points / d_offsets point at the buffer the next pp_detect_async consumes
    float* d_points_buf[2] = {nullptr, nullptr};
    int* d_offsets_buf[2] = {nullptr, nullptr};
    int in_buf = 0;                       // index of d_points / d_offsets (and of the voxeliser products: vox[])
    // The voxeliser's products exist twice as well (round 4): pp_upload_points_async voxelises batch k+1 right behind
    // its copy, on the copy stream, while batch k's PFN .. post-process read the other set -- the single-workgroup-
    // per-frame voxeliser (36 us on 64 of 256 CUs at B = 64) is then off the pass's chain of dependent launches.
    // The d_* members below always point at set in_buf.
    struct VoxSet {
        float* points_sorted = nullptr;
        int *cellmap = nullptr, *pstart = nullptr, *pcell = nullptr, *npillars = nullptr, *nvalid = nullptr;
        unsigned long long* occbits = nullptr;
        bool occ_cleared = false;         // its k_cell_first cleared the occupancy bitmap (consumed by run_pfn)
    } vox[2];
    int cache_budget_mb = 256;            // run_backbone's frame sub-ranges (pp_set_cache_budget; 0 = off)
    bool vox_ahead = false;               // the resident batch was voxelised at upload time (pp_detect_async skips it)
    bool prevox_issued = false;           // a voxeliser launch is (or was) queued on the copy stream: a main-stream one waits for ev_up
    int results_buf = 0;                  // set the last pp_detect_async read (pp_fetch_intermediates)
    hipStream_t copy_stream = nullptr;   // the device's shared upload stream (not owned by the handle)
    hipEvent_t ev_up = nullptr;           // recorded on the copy stream behind an asynchronous upload
    hipEvent_t ev_tgt = nullptr;          // ... behind the labels / regression targets of a training step
    bool up_pending = false;              // the next pp_detect_async must wait for ev_up
    hipEvent_t ev_read[2] = {nullptr, nullptr};   // recorded on the main stream behind the pass that read buffer i
    float* d_points = nullptr;
    float* d_points_sorted = nullptr;   // pillar-sorted copy left by k_sort_points (what the PFN streams)
    int* d_offsets = nullptr;
    int* d_cell = nullptr;
    int* d_first = nullptr;
    int* d_cellmap = nullptr;
    unsigned *d_keyA = nullptr, *d_idxA = nullptr, *d_keyB = nullptr, *d_idxB = nullptr;
    int* d_pstart = nullptr;
    int* d_pcell = nullptr;
    int* d_npillars = nullptr;
    int* d_nvalid = nullptr;
    float *d_pfn_w = nullptr, *d_pfn_b = nullptr;
    float* d_canvas = nullptr;
    float* d_act[2] = {nullptr, nullptr};
    float* d_concat = nullptr;
    float* d_head = nullptr;      // fused head map [B][H'*W'][PP_HEAD_COLS]
    float* d_cls = nullptr;       // compact class-logit plane [B][H'*W'][napl*ncls] (last fused-head deconv -> post-process)
    bool cls_plane_live = false;  // the last forward pass wrote d_cls (fused path with the uniform deconv kernels)
    bool fuse_heads = false;      // heads computed in the deconv epilogues (no concat buffer, no head launch)
    bool sparse_canvas = false;   // PFN writes occupied cells only; layer 0 consults the cell map (pp_finalize_weights)
    int* d_loss_labels = nullptr;      // training-side buffers (pp_head_loss), allocated on first use
    float* d_loss_regt = nullptr;
    int* d_loss_npos = nullptr;
    double* d_loss_partials = nullptr;
    float* d_loss_out = nullptr;
    float* h_train_losses = nullptr;      // page-locked [8]: the losses of a step launched by pp_train_step_async
    bool train_pending = false;           // ... which pp_train_step_wait has not collected yet
    float* d_head_grad = nullptr;
    int* d_integ = nullptr;
    unsigned long long* d_occbits = nullptr;   // [B][ny][occ_words(nx)] occupancy bitmap of the sparse-canvas passes
    bool occbits_live = false;                 // this pass's PFN launch wrote it (the pillar-centric kernel)
    bool ablate_vox_done = false;              // PP_ABLATE_STAGES bit 1 (timing experiments)
    uint8_t* d_mask = nullptr;
    float* d_anchors = nullptr;
    int* d_cells = nullptr;
    float* d_calib = nullptr;
    pp_detection* d_dets = nullptr;
    int* d_ndets = nullptr;
    pp_detection* h_dets = nullptr;  // pinned
    int* h_ndets = nullptr;          // pinned
    std::vector<LayerDesc> layers;
    std::vector<std::string> layer_tags;  // "<kernel symbol>:<layer>" for the profiler (for batch tag_batch)
    int tag_batch = -1;

    // compat scratch (grow-only)
    float* d_voxels = nullptr; size_t cap_voxels = 0;
    int* d_numpts = nullptr;   size_t cap_numpts = 0;
    int* d_coors = nullptr;    size_t cap_coors = 0;
    float* d_feat = nullptr;   size_t cap_feat = 0;

    int cur_batch = 0, cur_max_n = 0;
    int results_batch = 0;        // frames of the last enqueued pp_detect_async (0: no results to fetch)
    // frame offsets travel through a small pinned ring (a pageable source would be staged synchronously and a
    // single pinned buffer could be rewritten while its copy is still queued); a slot is reused only after the
    // event recorded behind its copy has passed
    static constexpr int OFF_RING = 4;
    int* h_off_ring = nullptr;    // pinned [OFF_RING][B + 1]
    hipEvent_t off_ev[OFF_RING] = {nullptr, nullptr, nullptr, nullptr};
    int off_slot = 0;
    hipEvent_t ev_in = nullptr;   // orders the engine's stream behind a producer stream (pp_upload_points_device)
    // zero-copy feed of small batches (pp_upload_points_async, batch <= PP_ZC_MAX_BATCH): one page-locked descriptor
    // per input buffer, read by k_cell_first; no copy-engine transfer, no events
    PpFeed* h_feed[2] = {nullptr, nullptr};
    const PpFeed* d_feed[2] = {nullptr, nullptr};
    bool zc = false;              // the uploaded batch is fed that way

    // training step (train.hip): shapes, flat layout and device buffers, set up by the first pp_train_* call
    struct TrainState {
        TrainShape shape;
        std::vector<TrainEntry> layout;
        int64_t n_params = 0, n_state = 0;
        TrainCtx cx;
        bool buffers = false;
        // the ~250 launches of a step replay as one hipGraph while nothing they depend on changes; ONE GRAPH PER INPUT
        // BUFFER: every upload flips the handle's input buffer (the kernels' point / offset pointers), so a single
        // graph would be re-captured on every optimizer step
        struct Graph {
            hipGraphExec_t exec = nullptr;     // voxelise + forward
            hipGraphExec_t exec_bwd = nullptr; // loss + backward (launched behind the target upload's event)
            int batch = -1, bucket = -1, zc = 0;
            const void *params = nullptr, *grads = nullptr, *state = nullptr;
            pp_loss_config loss;
        } graph[2];
        int last_batch = 0;    // frames of the last step (pp_train_fetch_decisions)
        int graph_state = 0;   // -1: capture failed once, plain launches from then on
        int n_captures = 0, n_replays = 0;   // pp_train_graph_stats
    };
    TrainState* train = nullptr;
    bool mask_in_pfn = false;      // the last run_pfn also computed the anchor mask (few frames)
    int f32_fallback_layers = 0;   // layers whose folded weights do not fit float16 pieces (pp_finalize_weights)
    bool force_f32 = false;        // pp_set_gemm_precision(PP_PREC_F32): no layer gets split weights

    int prof = 0;
    // pp_detect_async as one hipGraph launch (captured on first use per (batch, max points per frame))
    struct GraphSlot { hipGraphExec_t exec = nullptr; int batch = -1, bucket = -1, buf = -1, zc = 0, vox = 0; unsigned long long used = 0; };
    GraphSlot graphs[8];          // small LRU keyed by (batch, point-count bucket, input buffer)
    unsigned long long graph_tick = 0;
    int graph_state = 0;          // 0: try, -1: capture failed once (use plain launches)
    std::vector<hipEvent_t> events;
    std::vector<KTime> ktimes;
    int ev_used = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace {

// PP_CU_PARTITION=n (measurement switch, default 0 = off): n compute units of every XCD are reserved for the upload /
// voxeliser stream, the handles' own streams get the other 32 - n, and the persistent grids are sized for those
static int cu_partition() {
    static int v = -1;
    if (v < 0) { const char* s = getenv("PP_CU_PARTITION"); v = s ? atoi(s) : 0; if (v < 0 || v > 16) v = 0; }
    return v;
}

// process-wide upload stream of a device (created on first use, lives as long as the process)
hipStream_t device_copy_stream(int device) {
    static std::mutex mu;
    static std::map<int, hipStream_t> streams;
    std::lock_guard<std::mutex> lock(mu);
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    const int part = cu_partition();
    if (part > 0) {   // experiment (PP_CU_PARTITION): the upload + voxeliser stream on `part` CUs of every XCD, the handles' on the rest
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 8 * part; ++i) mask[i >> 5] |= 1u << (i & 31);
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    streams[device] = s;
    return s;
}

int fail(pp_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(e, call)                                                                              \
    do {                                                                                             \
        hipError_t _st = (call);                                                                     \
        if (_st != hipSuccess)                                                                       \
            return fail(e, PP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_st), __FILE__, __LINE__); \
    } while (0)

template <typename Tp>
int dalloc(pp_engine* e, Tp** p, size_t count) {
    void* q = nullptr;
    size_t bytes = count * sizeof(Tp);
    if (bytes == 0) bytes = sizeof(Tp);
    HIPCHK(e, hipMalloc(&q, bytes));
    e->allocs.push_back(q);
    *p = (Tp*)q;
    return PP_OK;
}

template <typename Tp>
int dgrow(pp_engine* e, Tp** p, size_t* cap, size_t count) {
    if (count <= *cap && *p) return PP_OK;
    if (*p) { (void)hipStreamSynchronize(e->stream); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    void* q = nullptr;
    HIPCHK(e, hipMalloc(&q, (count ? count : 1) * sizeof(Tp)));
    *p = (Tp*)q;
    *cap = count;
    return PP_OK;
}

// ---- profiling: an event pair around each kernel launch ----
void prof_reset(pp_engine* e) { e->ktimes.clear(); e->ev_used = 0; }

int prof_event(pp_engine* e) {
    if (e->ev_used == (int)e->events.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return -1;
        e->events.push_back(ev);
    }
    return e->ev_used++;
}

// Stage scope: the kernel launches inside record their own start / stop events (PP_LAUNCH) under `name`
// (NULL: under each launch site's kernel name); non-kernel work (a memset) is bracketed with plain event records.
}  // namespace

bool pp_prof_events(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    pp_engine* e = g_pp_prof.e;
    if (e == nullptr || e->prof <= 0) return false;
    const int e0 = prof_event(e), e1 = prof_event(e);
    if (e0 < 0 || e1 < 0) return false;
    e->ktimes.push_back({g_pp_prof.tag ? g_pp_prof.tag : name, e0});
    *start = e->events[e0];
    *stop = e->events[e1];
    return true;
}

namespace {

struct ProfScope {
    pp_engine* e;
    int e1 = -1;
    PpProf saved;          // scopes nest (pp_train_step wraps the voxeliser's named scopes)
    ProfScope(pp_engine* en, const char* name, bool bracket = false) : e(en), saved(g_pp_prof) {
        if (e->prof <= 0) return;
        if (bracket) {
            int e0 = prof_event(e);
            e1 = prof_event(e);
            if (e0 < 0 || e1 < 0) { e1 = -1; return; }
            (void)hipEventRecord(e->events[e0], e->stream);
            e->ktimes.push_back({name, e0});
        } else {
            g_pp_prof.e = e;
            g_pp_prof.tag = name;
        }
    }
    ~ProfScope() {
        if (e1 >= 0) (void)hipEventRecord(e->events[e1], e->stream);
        g_pp_prof = saved;
    }
};

int bits_ok(const pp_config& c) {
    for (int j = 0; j < 3; ++j)
        if (!(c.voxel_size[j] > 0.0) || !(c.pc_range[3 + j] > c.pc_range[j])) return 0;
    return 1;
}

const char* kBn[4] = {"gamma", "beta", "moving_mean", "moving_variance"};

const std::vector<float>* getw(pp_engine* e, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = e->hw.find(name);
    if (it == e->hw.end()) { e->err = "missing weight tensor '" + name + "'"; return nullptr; }
    const auto& sh = e->hshape[name];
    std::vector<int64_t> want(shape);
    if (sh != want) {
        std::string s = "weight '" + name + "' has shape [";
        for (auto v : sh) s += std::to_string(v) + ",";
        s += "] expected [";
        for (auto v : want) s += std::to_string(v) + ",";
        s += "]";
        e->err = s;
        return nullptr;
    }
    return &it->second;
}

// BatchNorm inference folded to y = x*scale + shift (eps 1e-3: model/pointpillars.py:109; Keras default for the RPN)
bool bn_fold(pp_engine* e, const std::string& prefix, int c, std::vector<float>& scale, std::vector<float>& shift) {
    const std::vector<float>* p[4];
    for (int i = 0; i < 4; ++i) {
        p[i] = getw(e, prefix + "/" + kBn[i], {c});
        if (!p[i]) return false;
    }
    scale.resize(c);
    shift.resize(c);
    for (int i = 0; i < c; ++i) {
        const float inv = (*p[0])[i] / sqrtf((*p[3])[i] + 1e-3f);
        scale[i] = inv;
        shift[i] = (*p[1])[i] - (*p[2])[i] * inv;
    }
    return true;
}

int upload(pp_engine* e, float** d, const std::vector<float>& h) {
    void* q = nullptr;
    HIPCHK(e, hipMalloc(&q, (h.size() ? h.size() : 1) * sizeof(float)));
    e->wallocs.push_back(q);   // released by the next pp_finalize_weights / pp_destroy
    *d = (float*)q;
    HIPCHK(e, hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return PP_OK;
}

// Split-precision operand for the bf16 matrix pipe: every float32 weight w becomes three bfloat16
// pieces hi + mid + lo (round-to-nearest-even each, 24 mantissa bits in total, exact for finite w),
// laid out [cin / 16][piece][n_total][16] so that one K-chunk's tile of one piece is contiguous
// (what k_sep_u / the deconv kernel stage into LDS).  Returned as raw 16-bit words packed in floats.
static inline uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
std::vector<float> split_weights_bf16x3(const std::vector<float>& wt, int n_total, int cin) {
    std::vector<uint16_t> out((size_t)n_total * cin * PP_NPIECE);
    const int nch = cin / 16;
    for (int n = 0; n < n_total; ++n)
        for (int c = 0; c < cin; ++c) {
            const float w = wt[(size_t)n * cin + c];
#if PP_SPLIT_MODE == 0
            const uint16_t hi = bf16_rne(w);
            const float r1 = w - bf16_to_f32(hi);
            const uint16_t mid = bf16_rne(r1);
            const float r2 = r1 - bf16_to_f32(mid);
            const uint16_t lo = bf16_rne(r2);
#else   // two float16 pieces (round-to-nearest-even conversions), the third slot of the layout stays zero
            const _Float16 hf = (_Float16)w;
            const _Float16 mf = (_Float16)(w - (float)hf);
            uint16_t hi, mid;
            memcpy(&hi, &hf, 2);
            memcpy(&mid, &mf, 2);
            const uint16_t lo = 0;
#endif
            const int kc = c / 16, cc = c % 16;
            const uint16_t pcs[3] = {hi, mid, lo};
            for (int p = 0; p < PP_NPIECE; ++p) out[(((size_t)kc * PP_NPIECE + p) * n_total + n) * 16 + cc] = pcs[p];
        }
    (void)nch;
    std::vector<float> packed(out.size() / 2);
    memcpy(packed.data(), out.data(), out.size() * 2);
    return packed;
}

// May these BN-folded weights go to the two-float16-piece operand layout?  A piece is a float16: |w| must stay below
// its largest finite value (with headroom for the rounding of hi); a layer that fails runs on the float32 matrix
// instruction instead (its d_wt16 stays NULL and the launchers pick the PREC = 0 instantiations).  Small weights need
// no guard: below 2^-3 the mid piece is a float16 subnormal, so a weight carries an ABSOLUTE error of at most 2^-25,
// which is what bounds the error of a dot product whose other terms are O(1).
static bool f16_pair_range_ok(const std::vector<float>& wt) {
#if PP_SPLIT_MODE == 0
    (void)wt;
    return true;          // bfloat16 pieces have float32's exponent range
#else
    for (float w : wt)
        if (!(fabsf(w) < 32768.f)) return false;      // also catches NaN / inf
    return true;
#endif
}

// canvas -> host (debug taps).  With the sparse canvas the cells without a pillar were never written: they are
// zeroed here from the cell map, so the caller sees the dense pseudo-image of the reference.
static int fetch_canvas(pp_engine* e, float* canvas, int batch, int set) {
    const size_t plane = (size_t)e->ny * e->nx;
    HIPCHK(e, hipMemcpyAsync(canvas, e->d_canvas, (size_t)batch * plane * e->C * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (!e->sparse_canvas) return PP_OK;
    std::vector<int> cm((size_t)batch * e->nz * plane);
    HIPCHK(e, hipMemcpy(cm.data(), e->vox[set].cellmap, cm.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b)
        for (size_t c = 0; c < plane; ++c) {
            bool occ = false;
            for (int z = 0; z < e->nz; ++z) occ = occ || cm[((size_t)b * e->nz + z) * plane + c] >= 0;
            if (!occ) memset(canvas + ((size_t)b * plane + c) * e->C, 0, (size_t)e->C * sizeof(float));
        }
    return PP_OK;
}

// the voxeliser products the d_* members name: set i (follows in_buf)
static void use_vox_set(pp_engine* e, int i) {
    const pp_engine::VoxSet& v = e->vox[i];
    e->d_points_sorted = v.points_sorted; e->d_cellmap = v.cellmap; e->d_pstart = v.pstart; e->d_pcell = v.pcell;
    e->d_npillars = v.npillars; e->d_nvalid = v.nvalid; e->d_occbits = v.occbits;
}

// ---- stage pipelines (all enqueue on e->stream) ----
const unsigned* sorted_idx(pp_engine* e);
int run_voxelize(pp_engine* e, int batch, int max_n, hipStream_t vs = nullptr) {
    if (vs == nullptr) {
        vs = e->stream;
        // the voxeliser's scratch (cells, keys, indices) exists once: a launch here must not overtake one that
        // pp_upload_points_async queued on the copy stream
        if (e->prevox_issued) { HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_up, 0)); e->prevox_issued = false; }
    }
    const bool lds_first = voxel_first_in_lds(max_n, e->ncell, e->cfg.max_voxels);
    int* d_first = lds_first ? nullptr : e->d_first;
    if (!lds_first) {
        ProfScope ps(e, "memset_first", true);
        HIPCHK(e, hipMemsetAsync(e->d_first, 0x7f, (size_t)batch * e->ncell * sizeof(int), vs));
    }
    {
        ProfScope ps(e, "k_cell_first");   // also clears the cell map
        launch_cell_first(e->d_points, e->d_offsets, batch, max_n, e->F, e->geom, e->d_cell, d_first, e->d_cellmap,
                          e->zc ? e->d_feed[e->in_buf] : nullptr, e->d_points, e->d_offsets, vs,
                          e->sparse_canvas ? e->d_occbits : nullptr, e->ny * occ_words(e->nx));
        e->vox[e->in_buf].occ_cleared = e->sparse_canvas;
    }
    {
        ProfScope ps(e, "k_voxel_frame");
        launch_voxel_frame(e->d_offsets, e->d_cell, d_first, e->d_cellmap, e->d_keyA, e->d_idxA, e->d_keyB,
                           e->d_idxB, e->d_pstart, e->d_pcell, e->d_npillars, e->d_nvalid, batch, max_n, e->ncell,
                           e->cfg.max_voxels, vs);
    }
    {
        ProfScope ps(e, "k_sort_points");
        launch_sort_points(e->d_points, e->d_offsets, sorted_idx(e), e->d_nvalid, batch, max_n, e->F,
                           e->d_points_sorted, vs);
    }
    HIPCHK(e, hipGetLastError());
    return PP_OK;
}

const unsigned* sorted_idx(pp_engine* e) {
    return (voxel_sort_passes(e->cfg.max_voxels) % 2 == 0) ? e->d_idxA : e->d_idxB;
}

// largest batch whose anchor masks ride in the PFN launch (PP_MASK_IN_PFN=0: never, =n: up to n frames).  Few frames
// only: the 32 KB LDS image the extra workgroups declare caps EVERY workgroup of the launch at 5 per CU, and a full
// chip of PFN workgroups lives on occupancy (B = 64: 73 -> 99 us with the masks inside, against 14 us for the three
// mask kernels by themselves); on one frame the launch is 14.7 us instead of 12.2 + 8.8.
static int anchor_mask_in_pfn_max_batch() {
    static int v = -1;
    if (v < 0) { const char* s = getenv("PP_MASK_IN_PFN"); v = s ? atoi(s) : 8; }
    return v;
}

int run_pfn(pp_engine* e, int batch, bool padded, float* feat_out, bool with_mask = false) {
    PfnParams p;
    memset(&p, 0, sizeof(p));
    p.batch = batch; p.nz = e->nz; p.ny = e->ny; p.nx = e->nx; p.C = e->C; p.F = e->F; p.T = e->T;
    p.max_voxels = e->cfg.max_voxels;
    p.vx = (float)e->cfg.voxel_size[0];
    p.vy = (float)e->cfg.voxel_size[1];
    // Python-float64 arithmetic, then a float32 constant (model/pointpillars.py:121-124)
    p.x_off = (float)(e->cfg.voxel_size[0] / 2 + e->cfg.pc_range[0]);
    p.y_off = (float)(e->cfg.voxel_size[1] / 2 + e->cfg.pc_range[1]);
    p.w = e->d_pfn_w; p.bias = e->d_pfn_b; p.cellmap = e->d_cellmap;
    p.pts_sorted = e->d_points_sorted; p.offsets = e->d_offsets; p.pillar_start = e->d_pstart;
    p.pillar_cell = e->d_pcell; p.npillars = e->d_npillars;
    p.voxels = e->d_voxels; p.num_points = e->d_numpts;
    p.canvas = e->d_canvas; p.feat_out = feat_out;
    p.sparse = e->sparse_canvas ? 1 : 0;
    p.with_distance = e->with_dist ? 1 : 0;
    // the bitmap is only as good as its clearing: the fused path's k_cell_first does it (run_voxelize); the stage entry
    // points build the cell map from the caller's coordinates and keep the cell-map lookups
    p.occbits = (e->sparse_canvas && !padded && e->vox[e->in_buf].occ_cleared) ? e->d_occbits : nullptr;
    e->occbits_live = p.occbits != nullptr && pfn_writes_occbits(p, padded);
    if (!e->occbits_live) p.occbits = nullptr;
    e->vox[e->in_buf].occ_cleared = false;
    e->mask_in_pfn = false;
    if (with_mask && batch <= anchor_mask_in_pfn_max_batch() && pfn_can_carry_anchor_mask(p, padded)) {
        // the anchor mask (needs the cell map only, read by the post-process only) rides in this launch
        p.am_cells = e->d_cells; p.am_A = e->A; p.am_threshold = e->cfg.anchor_area_threshold; p.am_mask = e->d_mask;
        e->mask_in_pfn = true;
    }
    ProfScope ps(e, "k_pfn_canvas:pfn+scatter");
    int st = launch_pfn(p, padded, e->stream);
    if (st) return fail(e, st, "PFN: unsupported C=%d / F=%d", e->C, e->F);
    HIPCHK(e, hipGetLastError());
    return PP_OK;
}

int run_anchor_mask(pp_engine* e, int batch) {
    ProfScope ps(e, nullptr);   // three kernels, each under its own name
    static int bits = -1;       // PP_ANCHOR_MASK_BITS=0: the integral-image kernels on the sparse-canvas path too
    if (bits < 0) { const char* s_ = getenv("PP_ANCHOR_MASK_BITS"); bits = (s_ && s_[0] == '0') ? 0 : 1; }
    if (bits && e->occbits_live && e->nz == 1) {   // one z-cell: a bit of the occupancy bitmap is the pillar count
        launch_anchor_mask_bits(e->d_occbits, batch, e->ny, e->nx, e->d_cells, e->A, e->cfg.anchor_area_threshold, e->d_mask,
                                e->stream);
        HIPCHK(e, hipGetLastError());
        return PP_OK;
    }
    launch_anchor_mask(e->d_cellmap, batch, e->nz, e->ny, e->nx, e->d_cells, e->A, e->cfg.anchor_area_threshold,
                       e->d_integ, e->d_mask, e->stream);
    HIPCHK(e, hipGetLastError());
    return PP_OK;
}

void refresh_tags(pp_engine* e, int batch) {
    if (e->tag_batch == batch) return;
    for (size_t i = 0; i < e->layers.size(); ++i)
        e->layer_tags[i] = layer_kernel_name(e->layers[i], batch) + ":" + e->layers[i].name;
    e->tag_batch = batch;
}

int run_backbone(pp_engine* e, int batch) {
    refresh_tags(e, batch);
    e->cls_plane_live = false;
    for (const LayerDesc& L : e->layers) if (layer_writes_cls_plane(L)) e->cls_plane_live = true;
    // Frame sub-ranges for the layers whose maps do not fit the last-level cache (round 4).  A run of consecutive
    // separable layers of a block is walked sub-batch by sub-batch -- L1(s0) L2(s0) L3(s0), L1(s1) ... -- so that what a
    // layer reads is what the layer before has just written for the same frames and still sits in the 256 MB cache:
    // on the KITTI-shaped B = 32 maps (439 MB in block1, 219 MB in block2) a layer's loads are a third of its time, on
    // cfg-A's 84-168 MB maps they are free already.  Same kernels, same buffers, same pixel
    // numbering: a launch walks a sub-range of the batch's tiles (launch_layer's frame0).  The budget (input + output map
    // of a launch) is the handle's cache_budget_mb: pp_set_cache_budget, default 256 = the whole cache for ONE handle in
    // flight; callers that keep several handles in flight set 0 -- their working sets evict each other and the extra
    // launches only cost (measured: DESIGN section 4.6).
    const int sub_on = e->cache_budget_mb > 0 ? 1 : 0;
    const long sub_mb = e->cache_budget_mb;
    size_t i = 0;
    while (i < e->layers.size()) {
        // the run [i, j) of separable layers, and the sub-batch it is walked in (1 = the whole batch)
        size_t j = i;
        int sub = batch;
        if (sub_on && e->layers[i].kind == LAYER_SEP) {
            double per_frame = 0.0;          // bytes of the largest (input + output) map pair of the run, per frame
            while (j < e->layers.size() && e->layers[j].kind == LAYER_SEP) {
                const LayerDesc& l = e->layers[j];
                const double in_b = (j == 0 && l.d_occ != nullptr) ? 0.0 : 4.0 * l.in_h * l.in_w * l.cin;   // (the sparse canvas is not read as a map)
                per_frame = std::max(per_frame, in_b + 4.0 * l.out_h * l.out_w * l.cout);
                ++j;
            }
            const double budget = (double)sub_mb * 1048576.0;
            while (sub > 1 && per_frame * sub > budget && sub % 2 == 0) sub /= 2;
            // every launch must stay a full-chip k_sep_u launch whose sub-range starts on a tile boundary
            for (bool ok = false; sub < batch && !ok; ) {
                ok = true;
                for (size_t k = i; k < j && ok; ++k)
                    for (int f0 = 0; f0 < batch && ok; f0 += sub)
                        ok = launch_layer_subrange_ok(e->layers[k], f0, std::min(sub, batch - f0), batch) &&
                             (long long)sub * e->layers[k].out_h * e->layers[k].out_w >= 1024ll * 128;
                if (!ok) sub *= 2;
            }
            if (sub >= batch) sub = batch;
        } else {
            j = i + 1;
        }
        for (int f0 = 0; f0 < batch; f0 += sub) {
            const int nb = std::min(sub, batch - f0);
            for (size_t k = i; k < j; ++k) {
                LayerDesc L = e->layers[k];
                // sparse first layer: the occupancy bitmap when this pass's PFN launch left one, else the cell map
                if (k == 0 && L.d_occ != nullptr) { L.d_occ = e->d_cellmap; L.d_occbits = e->occbits_live ? e->d_occbits : nullptr; }
                ProfScope ps(e, e->layer_tags[k].c_str());
                int st = launch_layer(L, nb, e->d_head, e->stream, 0, f0);
                if (st) return fail(e, st, "layer %s: unsupported shape (cin=%d cout=%d)", L.name, L.cin, L.cout);
            }
        }
        i = j;
    }
    HIPCHK(e, hipGetLastError());
    return PP_OK;
}

// to_host: the kernel also fills the page-locked result buffers (the fused path: no copy nodes behind it)
int run_post(pp_engine* e, int batch, bool to_host = false) {
    PostParams p;
    p.batch = batch; p.A = e->A; p.pre_max = e->cfg.nms_pre_max_size; p.post_max = e->cfg.nms_post_max_size;
    p.score_thr = e->cfg.nms_score_threshold; p.iou_thr = e->cfg.nms_iou_threshold;
    p.head = e->d_head; p.cls = e->cls_plane_live ? e->d_cls : nullptr; p.napl = e->napl; p.ncls = e->ncls; p.use_dir = e->use_dir ? 1 : 0; p.mask = e->d_mask; p.anchors = e->d_anchors;
    p.calib = e->d_calib; p.dets = e->d_dets; p.n_dets = e->d_ndets;
    p.dets_host = to_host ? e->h_dets : nullptr; p.n_dets_host = to_host ? e->h_ndets : nullptr;
    ProfScope ps(e, "k_postprocess");
    launch_postprocess(p, e->stream);
    HIPCHK(e, hipGetLastError());
    return PP_OK;
}

// The stage entry points (pp_points_to_voxel, pp_anchor_mask, pp_forward_voxels, pp_predict) reuse the fused
// path's device buffers (points, cell map, canvas, head map, mask, detections): after one of them the resident
// frames and the last results of the fused path are gone, and the calls that would read them say so.
void stage_call_done(pp_engine* e) {
    e->cur_batch = 0;
    e->cur_max_n = 0;
    e->results_batch = 0;
}

int check_batch(pp_engine* e, int batch) {
    if (batch < 1 || batch > e->B) return fail(e, PP_ERR_ARG, "batch %d outside [1, max_batch=%d]", batch, e->B);
    return PP_OK;
}

// Validates the frame offsets, flips to the other input buffer and queues the offsets' copy on `stream`
// (the main stream, or the copy stream for the asynchronous upload -- which first waits until the pass that
// last read that buffer has finished).
int set_offsets(pp_engine* e, const int32_t* off, int batch, hipStream_t stream) {
    if (!off) return fail(e, PP_ERR_ARG, "frame_offsets is NULL");
    if (off[0] != 0) return fail(e, PP_ERR_ARG, "frame_offsets[0] must be 0");
    int max_n = 0;
    for (int b = 0; b < batch; ++b) {
        const int n = off[b + 1] - off[b];
        if (n < 0) return fail(e, PP_ERR_ARG, "frame_offsets not monotone at frame %d", b);
        if (n > e->NMAX) return fail(e, PP_ERR_ARG, "frame %d has %d points > max_points_per_frame=%d", b, n, e->NMAX);
        if (n > max_n) max_n = n;
    }
    e->zc = false;                                     // inputs arrive by copy: the first kernel reads device memory
    const int slot = e->off_slot;
    e->off_slot = (slot + 1) % pp_engine::OFF_RING;
    HIPCHK(e, hipEventSynchronize(e->off_ev[slot]));   // the copy that last used this slot has been consumed
    int* ring = e->h_off_ring + (size_t)slot * (e->B + 1);
    memcpy(ring, off, (size_t)(batch + 1) * sizeof(int));
    e->cur_batch = batch;
    e->cur_max_n = max_n;
    const int nb = e->in_buf ^ 1;
    e->in_buf = nb;
    e->d_points = e->d_points_buf[nb];
    e->d_offsets = e->d_offsets_buf[nb];
    use_vox_set(e, nb);
    e->vox_ahead = false;
    HIPCHK(e, hipStreamWaitEvent(stream, e->ev_read[nb], 0));   // (a no-op on the main stream, which is ordered anyway)
    HIPCHK(e, hipMemcpyAsync(e->d_offsets, ring, (batch + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
    HIPCHK(e, hipEventRecord(e->off_ev[slot], stream));
    return PP_OK;
}

// fused head map [pixels][PP_HEAD_COLS] <-> the reference's three NHWC head tensors
int fetch_heads(pp_engine* e, int batch, float* box, float* cls, float* dir) {
    const size_t px = (size_t)batch * e->head_h * e->head_w;
    const int nb = e->napl * 7, nc = e->napl * e->ncls, nd = e->use_dir ? e->napl * 2 : 0;
    std::vector<float> h(px * PP_HEAD_COLS);
    HIPCHK(e, hipMemcpyAsync(h.data(), e->d_head, h.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (size_t p = 0; p < px; ++p) {
        const float* r = h.data() + p * PP_HEAD_COLS;
        if (box) memcpy(box + p * nb, r, nb * sizeof(float));
        if (cls) memcpy(cls + p * nc, r + nb, nc * sizeof(float));
        if (dir && nd) memcpy(dir + p * nd, r + nb + nc, nd * sizeof(float));
    }
    return PP_OK;
}

int upload_heads(pp_engine* e, int batch, const float* box, const float* cls, const float* dir) {
    const size_t px = (size_t)batch * e->head_h * e->head_w;
    const int nb = e->napl * 7, nc = e->napl * e->ncls, nd = e->use_dir ? e->napl * 2 : 0;
    std::vector<float> h(px * PP_HEAD_COLS, 0.f);
    for (size_t p = 0; p < px; ++p) {
        float* r = h.data() + p * PP_HEAD_COLS;
        memcpy(r, box + p * nb, nb * sizeof(float));
        memcpy(r + nb, cls + p * nc, nc * sizeof(float));
        if (nd) memcpy(r + nb + nc, dir + p * nd, nd * sizeof(float));
    }
    HIPCHK(e, hipMemcpyAsync(e->d_head, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));   // h is a local
    e->cls_plane_live = false;                    // the compact plane no longer mirrors the head map
    return PP_OK;
}

void calib_matrix(const float* rect, const float* trv, float* M) {
    // r_rect @ velo2cam in float32 (libraries/eval_helper_functions.py:732)
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += rect[i * 4 + k] * trv[k * 4 + j];
            M[i * 4 + j] = s;
        }
}

}  // namespace

extern "C" {

int pp_abi_version(void) { return PP_ABI_VERSION; }

const char* pp_last_error(pp_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pp_create(const pp_config* cfg, int device, pp_handle* out) {
    if (!cfg || !out) return fail(nullptr, PP_ERR_ARG, "pp_create: NULL argument");
    *out = nullptr;
    if (!bits_ok(*cfg)) return fail(nullptr, PP_ERR_ARG, "pp_create: bad range / voxel size");
    if (cfg->max_points < 1 || cfg->max_voxels < 1 || cfg->max_batch < 1 || cfg->max_points_per_frame < 1)
        return fail(nullptr, PP_ERR_ARG, "pp_create: max_points / max_voxels / max_batch / max_points_per_frame must be >= 1");
    if (cfg->num_point_features != 3 && cfg->num_point_features != 4)
        return fail(nullptr, PP_ERR_UNSUPPORTED, "pp_create: num_point_features must be 3 or 4");
    if (cfg->num_class < 1) return fail(nullptr, PP_ERR_ARG, "pp_create: num_class must be >= 1");
    if (cfg->num_anchor_per_loc < 1 ||
        cfg->num_anchor_per_loc * (7 + cfg->num_class + (cfg->use_direction_classifier ? 2 : 0)) > PP_HEAD_COLS)
        return fail(nullptr, PP_ERR_UNSUPPORTED, "pp_create: num_anchor_per_loc * (7 + num_class + 2) must fit the %d-column head row", PP_HEAD_COLS);
    if (cfg->nms_post_max_size < 1 || cfg->nms_pre_max_size < 1)
        return fail(nullptr, PP_ERR_ARG, "pp_create: nms sizes must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PP_ERR_HIP, "pp_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, PP_ERR_ARG, "pp_create: device %d not in [0,%d)", device, ndev);
    pp_engine* e = new pp_engine();
    e->cfg = *cfg;
    e->device = device;
    {   // PP_SUBBATCH=0 / PP_SUBBATCH_MB=n: the default of pp_set_cache_budget for handles of this process
        const char* s0 = getenv("PP_SUBBATCH");
        const char* s1 = getenv("PP_SUBBATCH_MB");
        if (s1) e->cache_budget_mb = std::max(0, atoi(s1));
        if (s0 && s0[0] == '0') e->cache_budget_mb = 0;
    }
    hipError_t st = hipSetDevice(device);
    if (st == hipSuccess) {
        const int part = cu_partition();
        if (part > 0) {
            uint32_t mask[8];
            for (int w = 0; w < 8; ++w) mask[w] = 0xffffffffu;
            for (int i = 0; i < 8 * part; ++i) mask[i >> 5] &= ~(1u << (i & 31));
            st = hipExtStreamCreateWithCUMask(&e->stream, 8, mask);
        } else {
            st = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        }
    }
    if (st != hipSuccess) {
        fail(nullptr, PP_ERR_HIP, "pp_create: %s", hipGetErrorString(st));
        delete e;
        return PP_ERR_HIP;
    }
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0)
            g_num_cus = ncu - 8 * cu_partition();
    }
    for (int j = 0; j < 3; ++j) {
        e->geom.lo[j] = cfg->pc_range[j];
        e->geom.vs[j] = cfg->voxel_size[j];
        e->geom.grid[j] = (int)nearbyint((cfg->pc_range[3 + j] - cfg->pc_range[j]) / cfg->voxel_size[j]);
    }
    e->nx = e->geom.grid[0]; e->ny = e->geom.grid[1]; e->nz = e->geom.grid[2];
    e->ncell = e->nx * e->ny * e->nz;
    e->geom.ncell = e->ncell;
    e->C = cfg->pfn_filters; e->F = cfg->num_point_features; e->T = cfg->max_points;
    e->ncls = cfg->num_class; e->use_dir = cfg->use_direction_classifier != 0; e->with_dist = cfg->with_distance != 0;
    e->FA = e->F + 5 + (e->with_dist ? 1 : 0);
    e->B = cfg->max_batch; e->NMAX = cfg->max_points_per_frame;
    e->napl = cfg->num_anchor_per_loc;
    const int osf = cfg->layer_strides[0] / cfg->upsample_strides[0];
    if (osf < 1 || e->nx < 1 || e->ny < 1 || e->nz < 1) {
        fail(nullptr, PP_ERR_ARG, "pp_create: degenerate grid");
        delete e;
        return PP_ERR_ARG;
    }
    e->head_h = e->ny / osf; e->head_w = e->nx / osf;
    e->A = (int64_t)e->head_h * e->head_w * e->napl;
    e->CC = cfg->num_upsample_filters[0] + cfg->num_upsample_filters[1] + cfg->num_upsample_filters[2];

    // layer table + map sizes
    int st2 = PP_OK;
    {
        int h = e->ny, w = e->nx, cin = e->C;
        size_t act_max = 1;
        int co_off = 0;
        static const char* sep_names[3][8] = {
            {"block1.0", "block1.1", "block1.2", "block1.3", "block1.4", "block1.5", "block1.6", "block1.7"},
            {"block2.0", "block2.1", "block2.2", "block2.3", "block2.4", "block2.5", "block2.6", "block2.7"},
            {"block3.0", "block3.1", "block3.2", "block3.3", "block3.4", "block3.5", "block3.6", "block3.7"}};
        static const char* dec_names[3] = {"deconv1", "deconv2", "deconv3"};
        for (int b = 0; b < 3 && st2 == PP_OK; ++b) {
            if (cfg->layer_nums[b] < 0 || cfg->layer_nums[b] > 7) { st2 = fail(nullptr, PP_ERR_UNSUPPORTED, "layer_nums[%d] must be 0..7", b); break; }
            for (int j = 0; j <= cfg->layer_nums[b]; ++j) {
                LayerDesc L;
                memset(&L, 0, sizeof(L));
                L.kind = LAYER_SEP; L.cin = cin; L.cout = cfg->num_filters[b];
                L.stride = (j == 0) ? cfg->layer_strides[b] : 1;
                L.in_h = h; L.in_w = w;
                L.out_h = (h + 2 - 3) / L.stride + 1; L.out_w = (w + 2 - 3) / L.stride + 1;
                L.n_total = L.cout; L.ld_out = L.cout; L.co_off = 0; L.name = sep_names[b][j];
                e->layers.push_back(L);
                h = L.out_h; w = L.out_w; cin = L.cout;
                act_max = std::max(act_max, (size_t)e->B * h * w * cin);
            }
            LayerDesc D;
            memset(&D, 0, sizeof(D));
            D.kind = LAYER_DECONV; D.cin = cin; D.cout = cfg->num_upsample_filters[b]; D.k = cfg->upsample_strides[b];
            D.in_h = h; D.in_w = w; D.out_h = h * D.k; D.out_w = w * D.k;
            D.n_total = D.k * D.k * D.cout; D.ld_out = e->CC; D.co_off = co_off; D.name = dec_names[b];
            if (D.out_h != e->head_h || D.out_w != e->head_w)
                st2 = fail(nullptr, PP_ERR_SHAPE, "deconv%d output %dx%d != head map %dx%d", b + 1, D.out_h, D.out_w, e->head_h, e->head_w);
            co_off += D.cout;
            e->layers.push_back(D);
        }
        if (st2 == PP_OK) {
            e->fuse_heads = true;
            for (const LayerDesc& L : e->layers)
                if (L.kind == LAYER_DECONV && !deconv_can_fuse_heads(L)) e->fuse_heads = false;
            if (getenv("PP_NO_HEAD_FUSION")) e->fuse_heads = false;   // A/B switch for measurements
        }
        if (st2 == PP_OK && e->fuse_heads) {
            int nd = 0;
            for (LayerDesc& L : e->layers)
                if (L.kind == LAYER_DECONV) L.head_mode = (nd++ == 0) ? 1 : 2;
        }
        if (st2 == PP_OK && !e->fuse_heads) {
            LayerDesc H;
            memset(&H, 0, sizeof(H));
            H.kind = LAYER_HEAD; H.cin = e->CC; H.cout = 32; H.in_h = e->head_h; H.in_w = e->head_w;
            H.out_h = e->head_h; H.out_w = e->head_w; H.n_total = 32; H.name = "heads";
            e->layers.push_back(H);
        }
        if (st2 == PP_OK) {
            pp_engine* q = e;
            const size_t BN = (size_t)e->B * e->NMAX;
            const size_t BMV = (size_t)e->B * cfg->max_voxels;
            const size_t HW = (size_t)e->head_h * e->head_w;
            auto A1 = [&](int s) { if (st2 == PP_OK) st2 = s; };
            A1(dalloc(q, &e->d_points_buf[0], BN * e->F));
            A1(dalloc(q, &e->d_points_buf[1], BN * e->F));
            A1(dalloc(q, &e->d_offsets_buf[0], (size_t)e->B + 1));
            A1(dalloc(q, &e->d_offsets_buf[1], (size_t)e->B + 1));
            e->d_points = e->d_points_buf[0];
            e->d_offsets = e->d_offsets_buf[0];
            A1(dalloc(q, &e->d_points_sorted, BN * e->F));
            A1(dalloc(q, &e->d_cell, BN));
            A1(dalloc(q, &e->d_first, (size_t)e->B * e->ncell));
            A1(dalloc(q, &e->d_cellmap, (size_t)e->B * e->ncell));
            A1(dalloc(q, &e->d_keyA, BN)); A1(dalloc(q, &e->d_idxA, BN));
            A1(dalloc(q, &e->d_keyB, BN)); A1(dalloc(q, &e->d_idxB, BN));
            A1(dalloc(q, &e->d_pstart, (size_t)e->B * (cfg->max_voxels + 1)));
            A1(dalloc(q, &e->d_pcell, BMV));
            A1(dalloc(q, &e->d_npillars, (size_t)e->B));
            A1(dalloc(q, &e->d_nvalid, (size_t)e->B));
            // activation buffers carry a zeroed PP_ZPAD_FLOATS header (see backbone.hip producers)
            auto APAD = [&](float** p, size_t count) {
                float* raw = nullptr;
                A1(dalloc(q, &raw, count + PP_ZPAD_FLOATS));
                if (st2 == PP_OK && hipMemset(raw, 0, PP_ZPAD_FLOATS * sizeof(float)) != hipSuccess) st2 = PP_ERR_HIP;
                *p = raw ? raw + PP_ZPAD_FLOATS : nullptr;
            };
            APAD(&e->d_canvas, (size_t)e->B * e->ny * e->nx * e->C);
            APAD(&e->d_act[0], act_max);
            APAD(&e->d_act[1], act_max);
            if (!e->fuse_heads) APAD(&e->d_concat, (size_t)e->B * HW * e->CC);
            A1(dalloc(q, &e->d_head, (size_t)e->B * HW * PP_HEAD_COLS));
            A1(dalloc(q, &e->d_cls, (size_t)e->B * HW * e->napl * e->ncls));
            A1(dalloc(q, &e->d_integ, (size_t)e->B * e->ny * e->nx));
            A1(dalloc(q, &e->d_occbits, (size_t)e->B * e->ny * occ_words(e->nx)));
            {   // voxeliser products, set 0 = the buffers above, set 1 = a second copy (see pp_engine::vox)
                pp_engine::VoxSet& v0 = e->vox[0];
                v0.points_sorted = e->d_points_sorted; v0.cellmap = e->d_cellmap; v0.pstart = e->d_pstart; v0.pcell = e->d_pcell;
                v0.npillars = e->d_npillars; v0.nvalid = e->d_nvalid; v0.occbits = e->d_occbits;
                pp_engine::VoxSet& v1 = e->vox[1];
                A1(dalloc(q, &v1.points_sorted, BN * e->F));
                A1(dalloc(q, &v1.cellmap, (size_t)e->B * e->ncell));
                A1(dalloc(q, &v1.pstart, (size_t)e->B * (cfg->max_voxels + 1)));
                A1(dalloc(q, &v1.pcell, BMV));
                A1(dalloc(q, &v1.npillars, (size_t)e->B));
                A1(dalloc(q, &v1.nvalid, (size_t)e->B));
                A1(dalloc(q, &v1.occbits, (size_t)e->B * e->ny * occ_words(e->nx)));
            }
            A1(dalloc(q, &e->d_mask, (size_t)e->B * e->A));
            A1(dalloc(q, &e->d_anchors, (size_t)e->A * 7));
            A1(dalloc(q, &e->d_cells, (size_t)e->A * 4));
            A1(dalloc(q, &e->d_calib, (size_t)e->B * 16));
            A1(dalloc(q, &e->d_dets, (size_t)e->B * cfg->nms_post_max_size));
            A1(dalloc(q, &e->d_ndets, (size_t)e->B));
            if (st2 == PP_OK && hipHostMalloc((void**)&e->h_dets, (size_t)e->B * cfg->nms_post_max_size * sizeof(pp_detection)) != hipSuccess) st2 = PP_ERR_HIP;
            if (st2 == PP_OK && hipHostMalloc((void**)&e->h_ndets, (size_t)e->B * sizeof(int)) != hipSuccess) st2 = PP_ERR_HIP;
            if (st2 == PP_OK && (hipEventCreate(&e->t0) != hipSuccess || hipEventCreate(&e->t1) != hipSuccess)) st2 = PP_ERR_HIP;
            if (st2 == PP_OK && hipHostMalloc((void**)&e->h_off_ring, (size_t)pp_engine::OFF_RING * (e->B + 1) * sizeof(int)) != hipSuccess) st2 = PP_ERR_HIP;
            for (int i = 0; i < pp_engine::OFF_RING && st2 == PP_OK; ++i)
                if (hipEventCreateWithFlags(&e->off_ev[i], hipEventDisableTiming) != hipSuccess) st2 = PP_ERR_HIP;
            if (st2 == PP_OK && hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming) != hipSuccess) st2 = PP_ERR_HIP;
            for (int i = 0; i < 2 && st2 == PP_OK; ++i) {
                void* dp = nullptr;
                if (hipHostMalloc((void**)&e->h_feed[i], sizeof(PpFeed) + (size_t)(e->B + 1) * sizeof(int)) != hipSuccess ||
                    hipHostGetDevicePointer(&dp, e->h_feed[i], 0) != hipSuccess) st2 = PP_ERR_HIP;
                e->d_feed[i] = (const PpFeed*)dp;
            }
            if (st2 == PP_OK && hipEventCreateWithFlags(&e->ev_up, hipEventDisableTiming) != hipSuccess) st2 = PP_ERR_HIP;
            if (st2 == PP_OK && hipEventCreateWithFlags(&e->ev_tgt, hipEventDisableTiming) != hipSuccess) st2 = PP_ERR_HIP;
            for (int i = 0; i < 2 && st2 == PP_OK; ++i)
                if (hipEventCreateWithFlags(&e->ev_read[i], hipEventDisableTiming) != hipSuccess) st2 = PP_ERR_HIP;
            if (st2 == PP_OK) {
                // one upload stream per device, shared by every handle: the host link is one resource, and the
                // runtime multiplexes streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, default 4) --
                // a private copy stream per handle made the handles' compute streams share queues and serialise
                e->copy_stream = device_copy_stream(device);
                if (e->copy_stream == nullptr) st2 = PP_ERR_HIP;
            }
            if (st2 == PP_OK) {
                // identity calibration until pp_set_calib
                std::vector<float> I((size_t)e->B * 16, 0.f);
                for (int b = 0; b < e->B; ++b) for (int d = 0; d < 4; ++d) I[(size_t)b * 16 + d * 5] = 1.f;
                if (hipMemcpy(e->d_calib, I.data(), I.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) st2 = PP_ERR_HIP;
            }
            if (st2 != PP_OK && g_create_error.empty()) g_create_error = e->err.empty() ? "pp_create: allocation failed" : e->err;
            if (st2 != PP_OK && !e->err.empty()) g_create_error = e->err;
        }
        // wire activation buffers: ping-pong inside the blocks, deconvs into the concat buffer
        if (st2 == PP_OK) {
            e->layer_tags.assign(e->layers.size(), std::string());
            const float* cur = e->d_canvas;
            int pp = 0;
            if (e->fuse_heads && !getenv("PP_NO_CLS_PLANE")) {   // the last branch finishes the head sums
                LayerDesc* last = nullptr;
                for (LayerDesc& L : e->layers) if (L.kind == LAYER_DECONV) last = &L;
                if (last && last->head_mode == 2) {
                    last->d_cls_plane = e->d_cls;
                    last->cls_col0 = e->napl * 7;
                    last->cls_ncol = e->napl * e->ncls;
                }
            }
            for (LayerDesc& L : e->layers) {
                if (L.kind == LAYER_SEP) { L.in = cur; L.out = e->d_act[pp]; cur = L.out; pp ^= 1; }
                else if (L.kind == LAYER_DECONV) { L.in = cur; L.out = e->fuse_heads ? nullptr : e->d_concat; }
                else { L.in = e->d_concat; L.out = nullptr; }
            }
        }
    }
    if (st2 != PP_OK) { pp_destroy(e); return st2; }
    *out = e;
    return PP_OK;
}

static void graph_invalidate(pp_engine* e);

int pp_destroy(pp_handle e) {
    if (!e) return PP_OK;
    (void)hipSetDevice(e->device);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    graph_invalidate(e);
    for (void* p : e->allocs) (void)hipFree(p);
    for (void* p : e->wallocs) (void)hipFree(p);
    if (e->train) for (auto& tg : e->train->graph) {
        if (tg.exec) (void)hipGraphExecDestroy(tg.exec);
        if (tg.exec_bwd) (void)hipGraphExecDestroy(tg.exec_bwd);
    }
    delete e->train;
    if (e->h_off_ring) (void)hipHostFree(e->h_off_ring);
    for (hipEvent_t ev : e->off_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    for (PpFeed* f : e->h_feed) if (f) (void)hipHostFree(f);
    if (e->ev_up) (void)hipEventDestroy(e->ev_up);
    if (e->ev_tgt) (void)hipEventDestroy(e->ev_tgt);
    if (e->h_train_losses) (void)hipHostFree(e->h_train_losses);
    for (hipEvent_t ev : e->ev_read) if (ev) (void)hipEventDestroy(ev);
    if (e->d_voxels) (void)hipFree(e->d_voxels);
    if (e->d_numpts) (void)hipFree(e->d_numpts);
    if (e->d_coors) (void)hipFree(e->d_coors);
    if (e->d_feat) (void)hipFree(e->d_feat);
    if (e->h_dets) (void)hipHostFree(e->h_dets);
    if (e->h_ndets) (void)hipHostFree(e->h_ndets);
    for (hipEvent_t ev : e->events) (void)hipEventDestroy(ev);
    if (e->t0) (void)hipEventDestroy(e->t0);
    if (e->t1) (void)hipEventDestroy(e->t1);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return PP_OK;
}

int pp_set_weight(pp_handle e, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!e) return PP_ERR_ARG;
    if (!name || !data || !shape || ndim < 1 || ndim > 4) return fail(e, PP_ERR_ARG, "pp_set_weight: bad argument");
    size_t n = 1;
    std::vector<int64_t> sh(shape, shape + ndim);
    for (auto v : sh) { if (v < 1) return fail(e, PP_ERR_SHAPE, "pp_set_weight(%s): non-positive dim", name); n *= (size_t)v; }
    e->hw[name].assign(data, data + n);
    e->hshape[name] = sh;
    e->weights_ready = false;
    return PP_OK;
}

int pp_finalize_weights(pp_handle e) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    graph_invalidate(e);          // waits for the stream; the captured graphs hold the old weight pointers
    e->weights_ready = false;
    for (void* p : e->wallocs) (void)hipFree(p);   // the previous weight set (the stream is idle)
    e->wallocs.clear();
    std::vector<float> sc, sh;
    // PFN: dense [Fa,C] * scale, bias = shift
    {
        const auto* k = getw(e, "pfn/dense/kernel", {e->FA, e->C});
        if (!k || !bn_fold(e, "pfn/bn", e->C, sc, sh)) return PP_ERR_SHAPE;
        std::vector<float> w((size_t)e->FA * e->C);
        for (int f = 0; f < e->FA; ++f)
            for (int c = 0; c < e->C; ++c) w[(size_t)f * e->C + c] = (*k)[(size_t)f * e->C + c] * sc[c];
        int st = upload(e, &e->d_pfn_w, w); if (st) return st;
        st = upload(e, &e->d_pfn_b, sh); if (st) return st;
    }
    // the three head kernels as one [PP_HEAD_COLS][CC] matrix (rows: box | cls | dir | zero pad) + bias
    std::vector<float> headw((size_t)PP_HEAD_COLS * e->CC, 0.f), headb(PP_HEAD_COLS, 0.f);
    {
        const int nb = e->napl * 7, nc = e->napl * e->ncls, nd = e->use_dir ? e->napl * 2 : 0, CC = e->CC;
        const auto* kb = getw(e, "rpn/conv_box/kernel", {1, 1, CC, nb});
        const auto* bb = getw(e, "rpn/conv_box/bias", {nb});
        const auto* kc = getw(e, "rpn/conv_cls/kernel", {1, 1, CC, nc});
        const auto* bc = getw(e, "rpn/conv_cls/bias", {nc});
        if (!kb || !bb || !kc || !bc) return PP_ERR_SHAPE;
        for (int ci = 0; ci < CC; ++ci) {
            for (int o = 0; o < nb; ++o) headw[(size_t)o * CC + ci] = (*kb)[(size_t)ci * nb + o];
            for (int o = 0; o < nc; ++o) headw[(size_t)(nb + o) * CC + ci] = (*kc)[(size_t)ci * nc + o];
        }
        for (int o = 0; o < nb; ++o) headb[o] = (*bb)[o];
        for (int o = 0; o < nc; ++o) headb[nb + o] = (*bc)[o];
        if (e->use_dir) {   // model/voxelnet.py:690: the direction head exists only with use_direction_classifier
            const auto* kd = getw(e, "rpn/conv_dir_cls/kernel", {1, 1, CC, nd});
            const auto* bd = getw(e, "rpn/conv_dir_cls/bias", {nd});
            if (!kd || !bd) return PP_ERR_SHAPE;
            for (int ci = 0; ci < CC; ++ci)
                for (int o = 0; o < nd; ++o) headw[(size_t)(nb + nc + o) * CC + ci] = (*kd)[(size_t)ci * nd + o];
            for (int o = 0; o < nd; ++o) headb[nb + nc + o] = (*bd)[o];
        }
    }
    int bi = 0, li = 0;
    e->f32_fallback_layers = 0;
    for (LayerDesc& L : e->layers) {
        if (L.kind == LAYER_SEP) {
            const std::string pre = "rpn/block" + std::to_string(bi + 1) + "/" + std::to_string(li);
            const auto* dw = getw(e, pre + "/depthwise_kernel", {3, 3, L.cin, 1});
            const auto* pw = getw(e, pre + "/pointwise_kernel", {1, 1, L.cin, L.cout});
            if (!dw || !pw || !bn_fold(e, pre + "/bn", L.cout, sc, sh)) return PP_ERR_SHAPE;
            std::vector<float> wt((size_t)L.cout * L.cin);
            for (int ci = 0; ci < L.cin; ++ci)
                for (int co = 0; co < L.cout; ++co) wt[(size_t)co * L.cin + ci] = (*pw)[(size_t)ci * L.cout + co] * sc[co];
            int st = upload(e, &L.d_dw, *dw); if (st) return st;
            st = upload(e, &L.d_wt, wt); if (st) return st;
            L.d_wt16 = nullptr;
            if (!e->force_f32 && L.cin % 16 == 0 && f16_pair_range_ok(wt)) { st = upload(e, &L.d_wt16, split_weights_bf16x3(wt, L.n_total, L.cin)); if (st) return st; }
            else ++e->f32_fallback_layers;
            st = upload(e, &L.d_bias, sh); if (st) return st;
            ++li;
        } else if (L.kind == LAYER_DECONV) {
            const std::string pre = "rpn/deconv" + std::to_string(bi + 1);
            const auto* k = getw(e, pre + "/kernel", {L.k, L.k, L.cout, L.cin});
            if (!k || !bn_fold(e, pre + "/bn", L.cout, sc, sh)) return PP_ERR_SHAPE;
            std::vector<float> wt(k->size());
            for (size_t n = 0; n < (size_t)L.n_total; ++n) {
                const int co = (int)(n % L.cout);
                for (int ci = 0; ci < L.cin; ++ci) wt[n * L.cin + ci] = (*k)[n * L.cin + ci] * sc[co];
            }
            int st = upload(e, &L.d_wt, wt); if (st) return st;
            L.d_wt16 = nullptr;
            L.d_head_wt16 = nullptr;
            if (!e->force_f32 && L.cin % 16 == 0 && f16_pair_range_ok(wt)) { st = upload(e, &L.d_wt16, split_weights_bf16x3(wt, L.n_total, L.cin)); if (st) return st; }
            else ++e->f32_fallback_layers;
            st = upload(e, &L.d_bias, sh); if (st) return st;
            if (L.head_mode != 0) {   // this branch's [PP_HEAD_COLS][cout] slice of the head matrix
                std::vector<float> hw((size_t)PP_HEAD_COLS * L.cout);
                for (int o = 0; o < PP_HEAD_COLS; ++o)
                    for (int c = 0; c < L.cout; ++c) hw[(size_t)o * L.cout + c] = headw[(size_t)o * e->CC + L.co_off + c];
                st = upload(e, &L.d_head_wt, hw); if (st) return st;
                if (!e->force_f32 && L.cout % 32 == 0 && f16_pair_range_ok(hw)) {
                    // k_deconv_u feeds the head GEMM from its accumulator registers: slot (h, j) of 16-channel group
                    // (n, g) holds channel n*32 + (j&3) + 8*(2g + (j>>2)) + 4h (the 32x32 MFMA result layout); the
                    // head kernels get the same order of k
                    std::vector<float> hwp(hw.size());
                    for (int o = 0; o < PP_HEAD_COLS; ++o)
                        for (int c = 0; c < L.cout; ++c) {
                            const int n = c / 32, g = (c % 32) / 16, sl = c % 16, hh = sl / 8, j = sl % 8;
                            const int src = n * 32 + (j & 3) + 8 * (2 * g + (j >> 2)) + 4 * hh;
                            hwp[(size_t)o * L.cout + c] = hw[(size_t)o * L.cout + src];
                        }
                    st = upload(e, &L.d_head_wt16, split_weights_bf16x3(hwp, PP_HEAD_COLS, L.cout)); if (st) return st;
                }
                st = upload(e, &L.d_head_bias, headb); if (st) return st;
            }
            ++bi; li = 0;
        } else {
            int st = upload(e, &L.d_wt, headw); if (st) return st;
            st = upload(e, &L.d_bias, headb); if (st) return st;
        }
    }
    // sparse canvas: on a large, mostly empty BEV grid (KITTI-shaped: 214 k cells, <= 12 k pillars) writing and
    // re-reading the zeros of the pseudo-image is most of the PFN's and the first layer's traffic.  The PFN then
    // writes only the cells that hold a pillar and the first layer looks every window position up in the cell
    // map.  Needs the kernels that know the lookup (sparse_input_supported); PP_DENSE_CANVAS=1 turns it off.
    {
        const char* env = getenv("PP_DENSE_CANVAS");
        const long long cells = (long long)e->ny * e->nx;
        LayerDesc& L0 = e->layers[0];
        const bool force = env && env[0] == '0';     // measurement switch: sparse wherever the kernels support it
        e->sparse_canvas = !(env && env[0] == '1') && (force || (cells >= 32768 && 4ll * e->cfg.max_voxels <= cells)) &&
                           L0.in == e->d_canvas && sparse_input_supported(L0, e->B);
        L0.d_occ = e->sparse_canvas ? e->d_cellmap : nullptr;
        L0.occ_nz = e->nz;
    }
    e->tag_batch = -1;            // which kernel runs a layer may depend on its weights (float16 range fallback)
    e->weights_ready = true;
    return PP_OK;
}

int pp_set_anchors(pp_handle e, const float* anchors, const int32_t* cells, int64_t num_anchors) {
    if (e) graph_invalidate(e);
    if (!e) return PP_ERR_ARG;
    if (!anchors || !cells) return fail(e, PP_ERR_ARG, "pp_set_anchors: NULL argument");
    if (num_anchors != e->A) return fail(e, PP_ERR_SHAPE, "pp_set_anchors: got %lld anchors, config needs %lld", (long long)num_anchors, (long long)e->A);
    for (int64_t a = 0; a < num_anchors; ++a) {
        const int32_t* c = cells + a * 4;
        if (c[0] < 0 || c[1] < 0 || c[2] >= e->nx || c[3] >= e->ny || c[0] >= e->nx || c[1] >= e->ny || c[2] < 0 || c[3] < 0)
            return fail(e, PP_ERR_ARG, "pp_set_anchors: anchor %lld cells out of the %dx%d grid", (long long)a, e->nx, e->ny);
    }
    (void)hipSetDevice(e->device);
    HIPCHK(e, hipMemcpy(e->d_anchors, anchors, (size_t)num_anchors * 7 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(e->d_cells, cells, (size_t)num_anchors * 4 * sizeof(int), hipMemcpyHostToDevice));
    e->anchors_ready = true;
    return PP_OK;
}

int pp_upload_points(pp_handle e, const float* points, const int32_t* frame_offsets, int32_t batch) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    st = set_offsets(e, frame_offsets, batch, e->stream); if (st) return st;
    e->up_pending = false;
    const size_t n = (size_t)frame_offsets[batch];
    if (n && !points) return fail(e, PP_ERR_ARG, "pp_upload_points: points is NULL");
    if (n) HIPCHK(e, hipMemcpyAsync(e->d_points, points, n * e->F * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));  // the host buffers may be pageable / reused by the caller
    return PP_OK;
}

#ifndef PP_ZC_MAX_BATCH
#define PP_ZC_MAX_BATCH 4
#endif
// live pp_host_alloc blocks (base -> bytes) whose device mapping is the identity; pp_host_free removes its entry
// BEFORE the memory goes back to the runtime, so a later lookup of a recycled address cannot hit
static std::mutex g_pinned_mu;
static std::map<uintptr_t, size_t> g_pinned;
static bool pinned_block_holds(const void* p, size_t bytes) {
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> lk(g_pinned_mu);
    auto it = g_pinned.upper_bound(a);
    if (it == g_pinned.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}
static bool zero_copy_enabled() {
    static int v = -1;
    if (v < 0) { const char* s = getenv("PP_NO_ZERO_COPY"); v = (s && s[0] == '1') ? 0 : 1; }
    return v != 0;
}

// Small batch, page-locked points: nothing is copied and no HIP call is made besides one event query -- the
// descriptor of the other input buffer gets the points' device address and the offsets, and the next pass's first
// kernel reads both over the host link (a 16 K-point frame is 196 KB: ~4 us of it) while it writes the device copies.
static int feed_zero_copy(pp_engine* e, const float* points_pinned, const int32_t* off, int batch) {
    if (!off) return fail(e, PP_ERR_ARG, "frame_offsets is NULL");
    if (off[0] != 0) return fail(e, PP_ERR_ARG, "frame_offsets[0] must be 0");
    int max_n = 0;
    for (int b = 0; b < batch; ++b) {
        const int n = off[b + 1] - off[b];
        if (n < 0) return fail(e, PP_ERR_ARG, "frame_offsets not monotone at frame %d", b);
        if (n > e->NMAX) return fail(e, PP_ERR_ARG, "frame %d has %d points > max_points_per_frame=%d", b, n, e->NMAX);
        if (n > max_n) max_n = n;
    }
    const void* dev = nullptr;
    if (off[batch] > 0) {
        // Device address of the caller's buffer.  No address is remembered per handle (a freed buffer's address can
        // come back as pageable memory): a range inside a LIVE pp_host_alloc block is its own device address
        // (hipHostMalloc under unified addressing) -- one lookup in the library's registry, which pp_host_free
        // updates -- and anything else is asked of the runtime on every call (pageable memory fails there and the
        // caller falls back to the copy path).
        const size_t bytes = (size_t)off[batch] * e->F * sizeof(float);
        if (pinned_block_holds(points_pinned, bytes)) dev = points_pinned;
        else {
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, (void*)points_pinned, 0) != hipSuccess || dp == nullptr) {
                (void)hipGetLastError();
                return PP_ERR_UNSUPPORTED;
            }
            dev = dp;
        }
    }
    const int nb = e->in_buf ^ 1;
    // the pass that last read this buffer's descriptor (two uploads ago) must be through
    HIPCHK(e, hipEventSynchronize(e->ev_read[nb]));
    PpFeed* f = e->h_feed[nb];
    f->src = (const float*)dev;
    memcpy(f->offsets, off, (size_t)(batch + 1) * sizeof(int));
    __atomic_thread_fence(__ATOMIC_RELEASE);
    e->in_buf = nb;
    e->d_points = e->d_points_buf[nb];
    e->d_offsets = e->d_offsets_buf[nb];
    use_vox_set(e, nb);
    e->vox_ahead = false;
    e->cur_batch = batch;
    e->cur_max_n = max_n;
    e->up_pending = false;
    e->zc = true;
    return PP_OK;
}

int pp_upload_points_async(pp_handle e, const float* points_pinned, const int32_t* frame_offsets, int32_t batch) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    if (frame_offsets && batch >= 1 && frame_offsets[batch] > 0 && !points_pinned)
        return fail(e, PP_ERR_ARG, "pp_upload_points_async: points is NULL");
    if (frame_offsets && batch >= 1 && batch <= PP_ZC_MAX_BATCH && zero_copy_enabled()) {
        st = feed_zero_copy(e, points_pinned, frame_offsets, batch);
        if (st != PP_ERR_UNSUPPORTED) return st;      // (not device-mapped memory: the copy below still works)
    }
    // on the copy stream, into the input buffer the running pass is not reading: no wait here, and the DMA runs
    // beside this handle's own kernels; pp_detect_async orders itself behind ev_up
    st = set_offsets(e, frame_offsets, batch, e->copy_stream); if (st) return st;
    const size_t n = (size_t)frame_offsets[batch];
    if (n) HIPCHK(e, hipMemcpyAsync(e->d_points, points_pinned, n * e->F * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
    // Voxelise right here, behind the copy and beside the pass in flight (PP_PREVOX=0: inside pp_detect_async as before).
    // Not while per-launch times are collected (their events belong to the main stream's pass) and not on a handle that
    // trains (pp_train_step voxelises inside its own graphs).
    static int prevox = -1;
    if (prevox < 0) { const char* s_ = getenv("PP_PREVOX"); prevox = (s_ && s_[0] == '0') ? 0 : 1; }
    if (prevox && e->prof <= 0 && e->train == nullptr) {
        if ((st = run_voxelize(e, batch, e->cur_max_n, e->copy_stream))) return st;
        e->vox_ahead = true;
        e->prevox_issued = true;
    }
    HIPCHK(e, hipEventRecord(e->ev_up, e->copy_stream));
    e->up_pending = true;
    return PP_OK;
}

int pp_host_alloc(int64_t bytes, void** out) {
    if (!out || bytes < 0) return fail(nullptr, PP_ERR_ARG, "pp_host_alloc: bad argument");
    *out = nullptr;
    if (hipHostMalloc(out, (size_t)(bytes ? bytes : 1)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, PP_ERR_HIP, "pp_host_alloc: hipHostMalloc(%lld) failed", (long long)bytes);
    }
    void* dp = nullptr;   // registered for the zero-copy feed only when the device sees the block at the same address
    if (hipHostGetDevicePointer(&dp, *out, 0) == hipSuccess && dp == *out) {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned[(uintptr_t)*out] = (size_t)(bytes ? bytes : 1);
    } else {
        (void)hipGetLastError();
    }
    return PP_OK;
}

int pp_host_free(void* p) {
    if (p) {
        std::lock_guard<std::mutex> lk(g_pinned_mu);
        g_pinned.erase((uintptr_t)p);
    }
    if (p && hipHostFree(p) != hipSuccess) return fail(nullptr, PP_ERR_HIP, "pp_host_free: hipHostFree failed");
    return PP_OK;
}

int pp_upload_points_device(pp_handle e, const void* points_dev, const int32_t* frame_offsets, int32_t batch,
                            void* producer_stream) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    st = set_offsets(e, frame_offsets, batch, e->stream); if (st) return st;
    e->up_pending = false;
    const size_t n = (size_t)frame_offsets[batch];
    if (n && !points_dev) return fail(e, PP_ERR_ARG, "pp_upload_points_device: points is NULL");
    if (producer_stream != nullptr) {
        // the copy must not start before the work queued on the producer's stream has written the points
        HIPCHK(e, hipEventRecord(e->ev_in, (hipStream_t)producer_stream));
        HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_in, 0));
    }
    if (n) HIPCHK(e, hipMemcpyAsync(e->d_points, points_dev, n * e->F * sizeof(float), hipMemcpyDeviceToDevice, e->stream));
    return PP_OK;
}

int pp_current_batch(pp_handle e, int32_t* uploaded, int32_t* results) {
    if (!e) return PP_ERR_ARG;
    if (uploaded) *uploaded = e->cur_batch;
    if (results) *results = e->results_batch;
    return PP_OK;
}

int pp_set_calib(pp_handle e, const float* rect, const float* trv2c, int32_t batch) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    if (!rect || !trv2c) return fail(e, PP_ERR_ARG, "pp_set_calib: NULL argument");
    std::vector<float> M((size_t)batch * 16);
    for (int b = 0; b < batch; ++b) calib_matrix(rect + (size_t)b * 16, trv2c + (size_t)b * 16, M.data() + (size_t)b * 16);
    HIPCHK(e, hipMemcpy(e->d_calib, M.data(), M.size() * sizeof(float), hipMemcpyHostToDevice));
    return PP_OK;
}

static void graph_invalidate(pp_engine* e) {
    // a replay of one of these graphs may still be running on the stream: its kernarg / node storage must
    // outlive it
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto& g : e->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        g = pp_engine::GraphSlot();
    }
}

// The only launch parameters that depend on the frames' point counts are k_cell_first's grid and the
// LDS-vs-global choice of the voxeliser, both functions of max(points per frame): a graph is keyed by that
// maximum rounded up to 4096 (the kernels bound-check every frame against its own count), so batches of
// similar size share one graph.
static int graph_bucket(const pp_engine* e, int max_n) {
    int b = ((max_n + 4095) / 4096) * 4096;
    if (b < 4096) b = 4096;
    return b < e->cfg.max_points_per_frame ? b : e->cfg.max_points_per_frame;
}

// the whole fused pipeline of one batch, enqueued on e->stream (plain launches or under stream capture)
static int enqueue_detect(pp_engine* e, int B, int max_n) {
    int st;
    // PP_ABLATE_STAGES (timing experiments only, WRONG results): bit 1 = the single-workgroup-per-frame voxeliser
    // kernel is left out after the handle's first pass (later passes reuse its products), bit 2 = no post-process.
    // What the step gains without them is what splitting them over more workgroups could gain at most.
    static int abl = -1;
    if (abl < 0) { const char* s_ = getenv("PP_ABLATE_STAGES"); abl = s_ ? atoi(s_) : 0; }
    if (!e->vox_ahead && !((abl & 1) && e->ablate_vox_done)) {
        if ((st = run_voxelize(e, B, max_n))) return st;
        e->ablate_vox_done = true;
    }
    if ((st = run_pfn(e, B, false, nullptr, true))) return st;
    if (!e->mask_in_pfn && (st = run_anchor_mask(e, B))) return st;
    if ((st = run_backbone(e, B))) return st;
    // the post-process stores its few kept detections per frame straight into the page-locked result buffers (PP_POST_COPY=1:
    // the two device-to-host copy nodes of rounds 1-3 instead)
    static int post_copy = -1;
    if (post_copy < 0) { const char* s = getenv("PP_POST_COPY"); post_copy = (s && s[0] == '1') ? 1 : 0; }
    if (!(abl & 2) && (st = run_post(e, B, post_copy == 0))) return st;
    if (post_copy) {
        HIPCHK(e, hipMemcpyAsync(e->h_dets, e->d_dets, (size_t)B * e->cfg.nms_post_max_size * sizeof(pp_detection), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->h_ndets, e->d_ndets, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    }
    return PP_OK;
}

static bool graphs_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* s = getenv("PP_NO_GRAPH");
        v = (s && s[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

int pp_detect_async(pp_handle e) {
    if (!e) return PP_ERR_ARG;
    if (!e->weights_ready) return fail(e, PP_ERR_STATE, "pp_detect_async: weights not finalised");
    if (!e->anchors_ready) return fail(e, PP_ERR_STATE, "pp_detect_async: anchors not set");
    if (e->cur_batch < 1) return fail(e, PP_ERR_STATE, "pp_detect_async: no frames uploaded");
    (void)hipSetDevice(e->device);
    const int B = e->cur_batch;
    prof_reset(e);
    if (e->up_pending || e->prevox_issued) {   // the frames were uploaded (and voxelised) on the copy stream
        HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_up, 0));     // (outside any capture: the event is not part of the graph)
        e->up_pending = false;
        e->prevox_issued = false;
    }
    // ~35 launches per batch replay as ONE graph launch: every kernel argument is a device pointer or a
    // per-(batch, max points) constant, so the captured graph is reusable until either changes (profiling
    // needs the per-launch events and uses plain launches)
    if (e->prof <= 0 && e->graph_state == 0 && graphs_enabled()) {
        const int bucket = graph_bucket(e, e->cur_max_n);
        pp_engine::GraphSlot* slot = nullptr;
        pp_engine::GraphSlot* lru = &e->graphs[0];
        for (auto& g : e->graphs) {
            if (g.exec && g.batch == B && g.bucket == bucket && g.buf == e->in_buf && g.zc == (e->zc ? 1 : 0) &&
                g.vox == (e->vox_ahead ? 1 : 0)) slot = &g;
            if (g.used < lru->used) lru = &g;
        }
        if (slot == nullptr) {
            slot = lru;
            if (slot->exec) {
                // LRU eviction: the evicted graph may still be replaying (pp_detect_async does not wait)
                HIPCHK(e, hipStreamSynchronize(e->stream));
                (void)hipGraphExecDestroy(slot->exec);
            }
            *slot = pp_engine::GraphSlot();
            hipGraph_t g = nullptr;
            bool ok = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            int st = ok ? enqueue_detect(e, B, bucket) : PP_ERR_HIP;
            if (ok && hipStreamEndCapture(e->stream, &g) != hipSuccess) { ok = false; g = nullptr; }
            if (ok && st == PP_OK && g != nullptr && hipGraphInstantiate(&slot->exec, g, nullptr, nullptr, 0) == hipSuccess) {
                slot->batch = B;
                slot->bucket = bucket;
                slot->buf = e->in_buf;
                slot->zc = e->zc ? 1 : 0;
                slot->vox = e->vox_ahead ? 1 : 0;
            } else {
                slot->exec = nullptr;
                e->graph_state = -1;           // fall back to plain launches for the life of the handle
                (void)hipGetLastError();
            }
            if (g) (void)hipGraphDestroy(g);
        }
        if (slot->exec != nullptr) {
            slot->used = ++e->graph_tick;
            HIPCHK(e, hipGraphLaunch(slot->exec, e->stream));
            HIPCHK(e, hipEventRecord(e->ev_read[e->in_buf], e->stream));
            e->results_batch = B;
            e->results_buf = e->in_buf;
            return PP_OK;
        }
    }
    int st = enqueue_detect(e, B, e->cur_max_n);
    if (st == PP_OK) {
        HIPCHK(e, hipEventRecord(e->ev_read[e->in_buf], e->stream));
        e->results_batch = B;
        e->results_buf = e->in_buf;
    }
    return st;
}

int pp_sync(pp_handle e) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return PP_OK;
}

// k_postprocess flags a frame whose head maps hold a non-finite value in bit PP_NDETS_NONFINITE of its count
static int check_numeric(pp_engine* e, const int* n_dets, int B, const char* who) {
    int bad = 0, first = -1;
    for (int b = 0; b < B; ++b)
        if (n_dets[b] & PP_NDETS_NONFINITE) { if (first < 0) first = b; ++bad; }
    if (!bad) return PP_OK;
    return fail(e, PP_ERR_NUMERIC, "%s: non-finite head outputs in %d of %d frames (first: frame %d); %s", who, bad, B, first,
                e->force_f32 ? "the network overflows float32 on these inputs"
                             : "an activation left the range of the float16 operand pieces (|x| < 65504): "
                               "pp_set_gemm_precision(h, PP_PREC_F32) and run the frames again");
}

int pp_set_gemm_precision(pp_handle e, int32_t precision) {
    if (!e) return PP_ERR_ARG;
    if (precision != PP_PREC_SPLIT_F16 && precision != PP_PREC_F32) return fail(e, PP_ERR_ARG, "pp_set_gemm_precision: unknown precision %d", precision);
    const bool f32 = precision == PP_PREC_F32;
    if (f32 == e->force_f32) return PP_OK;
    if (e->train_pending) return fail(e, PP_ERR_STATE, "pp_set_gemm_precision: a training step is in flight");
    e->force_f32 = f32;
    if (!e->weights_ready) return PP_OK;
    return pp_finalize_weights(e);       // waits for the stream, drops the graphs, rebuilds the device weights
}

int pp_set_cache_budget(pp_handle e, int32_t megabytes) {
    if (!e) return PP_ERR_ARG;
    if (megabytes < 0) return fail(e, PP_ERR_ARG, "pp_set_cache_budget: negative budget");
    if (megabytes == e->cache_budget_mb) return PP_OK;
    (void)hipSetDevice(e->device);
    graph_invalidate(e);          // the captured passes hold the old launch plan
    e->cache_budget_mb = megabytes;
    return PP_OK;
}

int pp_get_gemm_precision(pp_handle e, int32_t* precision) {
    if (!e || !precision) return PP_ERR_ARG;
    *precision = e->force_f32 ? PP_PREC_F32 : PP_PREC_SPLIT_F16;
    return PP_OK;
}

int pp_get_detections(pp_handle e, pp_detection* dets, int32_t* n_dets) {
    if (!e) return PP_ERR_ARG;
    if (!dets || !n_dets) return fail(e, PP_ERR_ARG, "pp_get_detections: NULL argument");
    const int B = e->results_batch;
    if (B < 1) return fail(e, PP_ERR_STATE, "pp_get_detections: no pp_detect_async results on this handle (or a stage call has reused the buffers)");
    (void)hipSetDevice(e->device);
    HIPCHK(e, hipStreamSynchronize(e->stream));   // immediate after pp_sync; never hands out a half-written buffer
    // the kept detections of every frame, zeros behind them (the kernel writes only what it keeps)
    const size_t pm = (size_t)e->cfg.nms_post_max_size;
    if (int st = check_numeric(e, e->h_ndets, B, "pp_get_detections")) return st;
    for (int b = 0; b < B; ++b) {
        size_t n = (size_t)std::max(0, std::min(e->h_ndets[b], (int)pm));
        memcpy(dets + (size_t)b * pm, e->h_dets + (size_t)b * pm, n * sizeof(pp_detection));
        memset(dets + (size_t)b * pm + n, 0, (pm - n) * sizeof(pp_detection));
    }
    memcpy(n_dets, e->h_ndets, (size_t)B * sizeof(int));
    return PP_OK;
}

int pp_detect(pp_handle e, const float* points, const int32_t* frame_offsets, int32_t batch, const float* rect,
              const float* trv2c, pp_detection* dets, int32_t* n_dets) {
    int st;
    if ((st = pp_upload_points(e, points, frame_offsets, batch))) return st;
    if (rect && trv2c && (st = pp_set_calib(e, rect, trv2c, batch))) return st;
    if ((st = pp_detect_async(e))) return st;
    if ((st = pp_sync(e))) return st;
    return pp_get_detections(e, dets, n_dets);
}

int pp_points_to_voxel(pp_handle e, const float* points, int64_t n, float* voxels, int32_t* coors,
                       int32_t* num_points, int32_t* n_pillars) {
    if (!e) return PP_ERR_ARG;
    if (!voxels || !coors || !num_points || !n_pillars) return fail(e, PP_ERR_ARG, "pp_points_to_voxel: NULL output");
    if (n < 0 || n > e->NMAX) return fail(e, PP_ERR_ARG, "pp_points_to_voxel: n=%lld outside [0, max_points_per_frame=%d]", (long long)n, e->NMAX);
    (void)hipSetDevice(e->device);
    const int32_t off[2] = {0, (int32_t)n};
    int st = pp_upload_points(e, points, off, 1); if (st) return st;
    prof_reset(e);
    if ((st = run_voxelize(e, 1, (int)n))) return st;
    const size_t MV = (size_t)e->cfg.max_voxels;
    if ((st = dgrow(e, &e->d_voxels, &e->cap_voxels, MV * e->T * e->F))) return st;
    if ((st = dgrow(e, &e->d_numpts, &e->cap_numpts, MV))) return st;
    if ((st = dgrow(e, &e->d_coors, &e->cap_coors, MV * 4))) return st;
    launch_voxel_expand(e->d_points_sorted, e->d_offsets, sorted_idx(e), e->d_pstart, e->d_pcell, e->d_npillars, 0, e->F,
                        e->T, e->cfg.max_voxels, e->ny, e->nx, e->d_voxels, e->d_coors, e->d_numpts, e->stream);
    HIPCHK(e, hipGetLastError());
    int P = 0;
    HIPCHK(e, hipMemcpyAsync(&P, e->d_npillars, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (P < 0 || P > e->cfg.max_voxels) return fail(e, PP_ERR_HIP, "pp_points_to_voxel: device returned %d pillars", P);
    *n_pillars = P;
    if (P) {
        HIPCHK(e, hipMemcpy(voxels, e->d_voxels, (size_t)P * e->T * e->F * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(e, hipMemcpy(coors, e->d_coors, (size_t)P * 3 * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(e, hipMemcpy(num_points, e->d_numpts, (size_t)P * sizeof(int), hipMemcpyDeviceToHost));
    }
    stage_call_done(e);
    return PP_OK;
}

static int upload_coors(pp_engine* e, const int32_t* coors, int64_t P, int32_t batch, const char* who) {
    for (int64_t p = 0; p < P; ++p) {
        const int32_t* c = coors + p * 4;
        if (c[0] < 0 || c[0] >= batch || c[1] < 0 || c[1] >= e->nz || c[2] < 0 || c[2] >= e->ny || c[3] < 0 || c[3] >= e->nx)
            return fail(e, PP_ERR_ARG, "%s: coors[%lld] = (%d,%d,%d,%d) outside batch=%d / grid z%d y%d x%d", who,
                        (long long)p, c[0], c[1], c[2], c[3], batch, e->nz, e->ny, e->nx);
    }
    int st = dgrow(e, &e->d_coors, &e->cap_coors, (size_t)P * 4); if (st) return st;
    if (P) HIPCHK(e, hipMemcpyAsync(e->d_coors, coors, (size_t)P * 4 * sizeof(int), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemsetAsync(e->d_cellmap, 0xff, (size_t)batch * e->ncell * sizeof(int), e->stream));
    launch_build_cellmap(e->d_coors, P, e->ncell, e->ny, e->nx, e->d_cellmap, e->stream);
    HIPCHK(e, hipGetLastError());
    return PP_OK;
}

int pp_anchor_mask(pp_handle e, const int32_t* coors, int64_t num_pillars, int32_t batch, uint8_t* mask) {
    if (!e) return PP_ERR_ARG;
    if (!e->anchors_ready) return fail(e, PP_ERR_STATE, "pp_anchor_mask: anchors not set");
    if ((num_pillars && !coors) || !mask || num_pillars < 0) return fail(e, PP_ERR_ARG, "pp_anchor_mask: bad argument");
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    if ((st = upload_coors(e, coors, num_pillars, batch, "pp_anchor_mask"))) return st;
    prof_reset(e);
    if ((st = run_anchor_mask(e, batch))) return st;
    HIPCHK(e, hipMemcpyAsync(mask, e->d_mask, (size_t)batch * e->A, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    stage_call_done(e);
    return PP_OK;
}

int pp_forward_voxels(pp_handle e, const float* voxels, const int32_t* num_points, const int32_t* coors,
                      int64_t P, int32_t batch, float* box_preds, float* cls_preds, float* dir_cls_preds,
                      float* pillar_features, float* canvas) {
    if (!e) return PP_ERR_ARG;
    if (!e->weights_ready) return fail(e, PP_ERR_STATE, "pp_forward_voxels: weights not finalised");
    if (P < 0 || (P && (!voxels || !num_points || !coors)) || !box_preds || !cls_preds || (e->use_dir && !dir_cls_preds))
        return fail(e, PP_ERR_ARG, "pp_forward_voxels: NULL argument");
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    for (int64_t p = 0; p < P; ++p)
        if (num_points[p] < 1 || num_points[p] > e->T)
            return fail(e, PP_ERR_ARG, "pp_forward_voxels: num_points[%lld]=%d outside [1,%d]", (long long)p, num_points[p], e->T);
    if ((st = upload_coors(e, coors, P, batch, "pp_forward_voxels"))) return st;
    if ((st = dgrow(e, &e->d_voxels, &e->cap_voxels, (size_t)P * e->T * e->F))) return st;
    if ((st = dgrow(e, &e->d_numpts, &e->cap_numpts, (size_t)P))) return st;
    if (pillar_features && (st = dgrow(e, &e->d_feat, &e->cap_feat, (size_t)P * e->C))) return st;
    if (P) {
        HIPCHK(e, hipMemcpyAsync(e->d_voxels, voxels, (size_t)P * e->T * e->F * sizeof(float), hipMemcpyHostToDevice, e->stream));
        HIPCHK(e, hipMemcpyAsync(e->d_numpts, num_points, (size_t)P * sizeof(int), hipMemcpyHostToDevice, e->stream));
    }
    prof_reset(e);
    if ((st = run_pfn(e, batch, true, pillar_features ? e->d_feat : nullptr))) return st;
    if ((st = run_backbone(e, batch))) return st;
    if ((st = fetch_heads(e, batch, box_preds, cls_preds, dir_cls_preds))) return st;
    if (pillar_features && P)
        HIPCHK(e, hipMemcpyAsync(pillar_features, e->d_feat, (size_t)P * e->C * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (canvas && (st = fetch_canvas(e, canvas, batch, e->in_buf))) return st;
    stage_call_done(e);
    return PP_OK;
}

int pp_predict(pp_handle e, const float* box_preds, const float* cls_preds, const float* dir_cls_preds,
               const uint8_t* anchors_mask, const float* rect, const float* trv2c, int32_t batch,
               pp_detection* dets, int32_t* n_dets) {
    if (!e) return PP_ERR_ARG;
    if (!e->anchors_ready) return fail(e, PP_ERR_STATE, "pp_predict: anchors not set");
    if (!box_preds || !cls_preds || (e->use_dir && !dir_cls_preds) || !anchors_mask || !rect || !trv2c || !dets || !n_dets)
        return fail(e, PP_ERR_ARG, "pp_predict: NULL argument");
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    if ((st = pp_set_calib(e, rect, trv2c, batch))) return st;
    if ((st = upload_heads(e, batch, box_preds, cls_preds, dir_cls_preds))) return st;
    HIPCHK(e, hipMemcpyAsync(e->d_mask, anchors_mask, (size_t)batch * e->A, hipMemcpyHostToDevice, e->stream));
    prof_reset(e);
    if ((st = run_post(e, batch))) return st;
    HIPCHK(e, hipMemcpyAsync(dets, e->d_dets, (size_t)batch * e->cfg.nms_post_max_size * sizeof(pp_detection), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(n_dets, e->d_ndets, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    stage_call_done(e);
    // non-finite predictions: the reference's predict() would hand NaN boxes on (np.argpartition over NaN scores);
    // this one says so instead (documented deviation)
    if ((st = check_numeric(e, n_dets, batch, "pp_predict"))) {
        for (int b = 0; b < batch; ++b) n_dets[b] = 0;
        return st;
    }
    return PP_OK;
}

int pp_fetch_intermediates(pp_handle e, int32_t* n_pillars, int32_t* coors, int32_t* num_points,
                           uint8_t* anchors_mask, float* box_preds, float* cls_preds, float* dir_cls_preds,
                           float* canvas) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    const int B = e->results_batch;
    if (B < 1) return fail(e, PP_ERR_STATE, "pp_fetch_intermediates: no fused-path pass to tap (run pp_detect / pp_detect_async first)");
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const int MV = e->cfg.max_voxels;
    std::vector<int> np(B);
    const pp_engine::VoxSet& vs = e->vox[e->results_buf];      // the set that pass read (a later upload has flipped the d_* members)
    HIPCHK(e, hipMemcpy(np.data(), vs.npillars, B * sizeof(int), hipMemcpyDeviceToHost));
    if (n_pillars) memcpy(n_pillars, np.data(), B * sizeof(int));
    if (coors || num_points) {
        std::vector<int> pcell((size_t)B * MV), pstart((size_t)B * (MV + 1));
        HIPCHK(e, hipMemcpy(pcell.data(), vs.pcell, pcell.size() * sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(e, hipMemcpy(pstart.data(), vs.pstart, pstart.size() * sizeof(int), hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b)
            for (int p = 0; p < np[b]; ++p) {
                const size_t r = (size_t)b * MV + p;
                if (coors) {
                    const int c = pcell[r];
                    coors[r * 3 + 0] = c / (e->nx * e->ny);
                    coors[r * 3 + 1] = (c / e->nx) % e->ny;
                    coors[r * 3 + 2] = c % e->nx;
                }
                if (num_points) {
                    const int cnt = pstart[(size_t)b * (MV + 1) + p + 1] - pstart[(size_t)b * (MV + 1) + p];
                    num_points[r] = cnt < e->T ? cnt : e->T;
                }
            }
    }
    if (anchors_mask) HIPCHK(e, hipMemcpy(anchors_mask, e->d_mask, (size_t)B * e->A, hipMemcpyDeviceToHost));
    if (box_preds || cls_preds || dir_cls_preds) {
        int st = fetch_heads(e, B, box_preds, cls_preds, dir_cls_preds);
        if (st) return st;
    }
    if (canvas) { int stc = fetch_canvas(e, canvas, B, e->results_buf); if (stc) return stc; }
    return PP_OK;
}

int pp_set_profiling(pp_handle e, int32_t level) {
    if (!e) return PP_ERR_ARG;
    e->prof = level > 0 ? 1 : 0;
    return PP_OK;
}

int pp_get_kernel_times(pp_handle e, int32_t capacity, const char** names, float* ms, int32_t* count) {
    if (!e) return PP_ERR_ARG;
    if (!count) return fail(e, PP_ERR_ARG, "pp_get_kernel_times: NULL count");
    (void)hipSetDevice(e->device);
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const int n = (int)e->ktimes.size();
    *count = n;
    for (int i = 0; i < n && i < capacity; ++i) {
        float t = 0.f;
        HIPCHK(e, hipEventElapsedTime(&t, e->events[e->ktimes[i].ev], e->events[e->ktimes[i].ev + 1]));
        if (names) names[i] = e->ktimes[i].name;
        if (ms) ms[i] = t;
    }
    return PP_OK;
}

int pp_timer_start(pp_handle e) {
    if (!e) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    HIPCHK(e, hipEventRecord(e->t0, e->stream));
    return PP_OK;
}

int pp_timer_stop(pp_handle e, float* elapsed_ms) {
    if (!e) return PP_ERR_ARG;
    if (!elapsed_ms) return fail(e, PP_ERR_ARG, "pp_timer_stop: NULL argument");
    (void)hipSetDevice(e->device);
    HIPCHK(e, hipEventRecord(e->t1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->t1));
    HIPCHK(e, hipEventElapsedTime(elapsed_ms, e->t0, e->t1));
    return PP_OK;
}

int pp_layer_count(pp_handle e, int32_t* count) {
    if (!e || !count) return PP_ERR_ARG;
    *count = (int32_t)e->layers.size();
    return PP_OK;
}

const char* pp_layer_tag(pp_handle e, int32_t layer) {
    if (!e || layer < 0 || layer >= (int)e->layer_tags.size()) return "";
    refresh_tags(e, e->cur_batch > 0 ? e->cur_batch : e->B);
    return e->layer_tags[layer].c_str();
}

extern long long* g_stamps;

int pp_bench_layer(pp_handle e, int32_t layer, int32_t batch, int32_t reps, int32_t ablate, float* avg_ms) {
    if (!e) return PP_ERR_ARG;
    if (!avg_ms || reps < 1 || layer < 0 || layer >= (int)e->layers.size()) return fail(e, PP_ERR_ARG, "pp_bench_layer: bad argument");
    if (!e->weights_ready) return fail(e, PP_ERR_STATE, "pp_bench_layer: weights not finalised");
    (void)hipSetDevice(e->device);
    int st = check_batch(e, batch); if (st) return st;
    const LayerDesc& L = e->layers[layer];
    if ((ablate & 64) && !g_stamps) HIPCHK(e, hipMalloc((void**)&g_stamps, (4096 * 8 + 64 * 64) * sizeof(long long)));
    for (int i = 0; i < 2; ++i)
        if ((st = launch_layer(L, batch, e->d_head, e->stream, ablate))) return fail(e, st, "pp_bench_layer: unsupported layer");
    HIPCHK(e, hipEventRecord(e->t0, e->stream));
    for (int i = 0; i < reps; ++i) launch_layer(L, batch, e->d_head, e->stream, ablate);
    HIPCHK(e, hipEventRecord(e->t1, e->stream));
    HIPCHK(e, hipEventSynchronize(e->t1));
    HIPCHK(e, hipGetLastError());
    float ms = 0.f;
    HIPCHK(e, hipEventElapsedTime(&ms, e->t0, e->t1));
    *avg_ms = ms / reps;
    if (ablate & 64) {   // in-kernel stamps of one extra launch -> stderr (tuning aid)
        const size_t n = 4096 * 8 + 64 * 64;
        std::vector<long long> hs(n, 0);
        HIPCHK(e, hipMemsetAsync(g_stamps, 0, n * sizeof(long long), e->stream));
        launch_layer(L, batch, e->d_head, e->stream, ablate);
        HIPCHK(e, hipMemcpyAsync(hs.data(), g_stamps, n * sizeof(long long), hipMemcpyDeviceToHost, e->stream));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (const char* path = getenv("PP_STAMPS_OUT")) {   // raw dump for offline analysis
            if (FILE* f = fopen(path, "wb")) { fwrite(hs.data(), sizeof(long long), n, f); fclose(f); }
        }
        for (int blk : {0, 1, 8, 63}) {
            for (int role = 0; role < 2; ++role) {
                const long long* st = hs.data() + ((size_t)blk * 2 + role) * 160;
                const long long t0 = hs[((size_t)blk * 2 + 0) * 160];
                fprintf(stderr, "blk %2d %s:", blk, role ? "prod" : "cons");
                for (int i = 0; i < 20; ++i) {
                    if (!st[i * 4 + 3] && !st[i * 4]) continue;
                    fprintf(stderr, " [%d: %lld %lld %lld %lld]", i, st[i * 4] ? st[i * 4] - t0 : -1, st[i * 4 + 1] ? st[i * 4 + 1] - t0 : -1,
                            st[i * 4 + 2] ? st[i * 4 + 2] - t0 : -1, st[i * 4 + 3] ? st[i * 4 + 3] - t0 : -1);
                }
                if (!role) fprintf(stderr, " end=%lld", st[39 * 4 + 1] ? st[39 * 4 + 1] - t0 : -1);
                fprintf(stderr, "\n");
            }
        }
    }
    return PP_OK;
}

// ---- AP-evaluator overlaps (stateless: host buffers in, host buffers out) ----
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
int riou_common(int device, const float* boxes, int64_t n, const float* qboxes, int64_t k, int32_t criterion,
                DevBuf& d_out, const char* who) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, PP_ERR_HIP, "%s: no HIP device available (this library has no CPU fallback)", who);
    if (device < 0 || device >= ndev) return fail(nullptr, PP_ERR_ARG, "%s: device %d not in [0,%d)", who, device, ndev);
    if (n < 0 || k < 0 || (n > 0 && !boxes) || (k > 0 && !qboxes)) return fail(nullptr, PP_ERR_ARG, "%s: bad argument", who);
    if (criterion < -1 || criterion > 2) return fail(nullptr, PP_ERR_ARG, "%s: criterion %d not in {-1,0,1,2}", who, criterion);
    if (n > 200000) return fail(nullptr, PP_ERR_ARG, "%s: at most 200000 boxes per call (got %lld)", who, (long long)n);
#define RCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(nullptr, PP_ERR_HIP, "%s: %s", who, hipGetErrorString(e_)); } while (0)
    RCHK(hipSetDevice(device));
    DevBuf d_b, d_q, d_bc, d_qc;
    RCHK(d_b.alloc(sizeof(float) * 5 * n)); RCHK(d_q.alloc(sizeof(float) * 5 * k));
    RCHK(d_bc.alloc(sizeof(float) * 9 * n)); RCHK(d_qc.alloc(sizeof(float) * 9 * k));
    RCHK(d_out.alloc(sizeof(float) * n * k));
    if (n == 0 || k == 0) return PP_OK;
    RCHK(hipMemcpy(d_b.p, boxes, sizeof(float) * 5 * n, hipMemcpyHostToDevice));
    RCHK(hipMemcpy(d_q.p, qboxes, sizeof(float) * 5 * k, hipMemcpyHostToDevice));
    launch_riou_corners((const float*)d_b.p, n, (float*)d_bc.p, nullptr);
    launch_riou_corners((const float*)d_q.p, k, (float*)d_qc.p, nullptr);
    launch_riou_pairs((const float*)d_bc.p, n, (const float*)d_qc.p, k, criterion, (float*)d_out.p, nullptr);
    RCHK(hipGetLastError());
    return PP_OK;
}
}  // namespace

int pp_rotate_iou_eval(int device, const float* boxes, int64_t n, const float* query_boxes, int64_t k,
                       int32_t criterion, float* out) {
    DevBuf d_out;
    int st = riou_common(device, boxes, n, query_boxes, k, criterion, d_out, "pp_rotate_iou_eval");
    if (st || n == 0 || k == 0) return st;
    if (!out) return fail(nullptr, PP_ERR_ARG, "pp_rotate_iou_eval: out is null");
    const char* who = "pp_rotate_iou_eval";
    RCHK(hipMemcpy(out, d_out.p, sizeof(float) * n * k, hipMemcpyDeviceToHost));
    return PP_OK;
}

int pp_d3_box_overlap(int device, const double* boxes, int64_t n, const double* query_boxes, int64_t k,
                      int32_t criterion, double* out) {
    const char* who = "pp_d3_box_overlap";
    if (n < 0 || k < 0 || (n > 0 && !boxes) || (k > 0 && !query_boxes)) return fail(nullptr, PP_ERR_ARG, "%s: bad argument", who);
    // BEV rectangles [x, z, l, w, ry] in float32, raw intersection area (criterion 2), like eval.py:160-161
    std::vector<float> b5((size_t)n * 5), q5((size_t)k * 5);
    const int sel[5] = {0, 2, 3, 5, 6};
    for (int64_t i = 0; i < n; ++i) for (int j = 0; j < 5; ++j) b5[i * 5 + j] = (float)boxes[i * 7 + sel[j]];
    for (int64_t i = 0; i < k; ++i) for (int j = 0; j < 5; ++j) q5[i * 5 + j] = (float)query_boxes[i * 7 + sel[j]];
    DevBuf d_rinc;
    int st = riou_common(device, b5.data(), n, q5.data(), k, 2, d_rinc, who);
    if (st || n == 0 || k == 0) return st;
    if (!out) return fail(nullptr, PP_ERR_ARG, "%s: out is null", who);
    if (criterion < -1 || criterion > 2) return fail(nullptr, PP_ERR_ARG, "%s: criterion %d not in {-1,0,1,2}", who, criterion);
    DevBuf d_b, d_q, d_o;
    RCHK(d_b.alloc(sizeof(double) * 7 * n)); RCHK(d_q.alloc(sizeof(double) * 7 * k)); RCHK(d_o.alloc(sizeof(double) * n * k));
    RCHK(hipMemcpy(d_b.p, boxes, sizeof(double) * 7 * n, hipMemcpyHostToDevice));
    RCHK(hipMemcpy(d_q.p, query_boxes, sizeof(double) * 7 * k, hipMemcpyHostToDevice));
    launch_d3_finish((const double*)d_b.p, n, (const double*)d_q.p, k, criterion, (const float*)d_rinc.p, (double*)d_o.p, nullptr);
    RCHK(hipGetLastError());
    RCHK(hipMemcpy(out, d_o.p, sizeof(double) * n * k, hipMemcpyDeviceToHost));
#undef RCHK
    return PP_OK;
}

// ---- training step ----
namespace {

int ensure_loss_buffers(pp_engine* e) {
    if (e->d_head_grad) return PP_OK;
    const size_t npx = (size_t)e->head_h * e->head_w;
    int st;
    if ((st = dalloc(e, &e->d_loss_labels, (size_t)e->B * e->A))) return st;
    if ((st = dalloc(e, &e->d_loss_regt, (size_t)e->B * e->A * 7))) return st;
    if ((st = dalloc(e, &e->d_loss_npos, (size_t)e->B))) return st;
    if ((st = dalloc(e, &e->d_loss_partials, (size_t)e->B * loss_blocks((int)npx) * 5))) return st;
    if ((st = dalloc(e, &e->d_loss_out, (size_t)8))) return st;
    if ((st = dalloc(e, &e->d_head_grad, (size_t)e->B * npx * PP_HEAD_COLS))) return st;
    return PP_OK;
}

void fill_loss_params(pp_engine* e, const pp_loss_config* lc, int batch, LossParams& p) {
    memset(&p, 0, sizeof(p));
    p.batch = batch; p.A = e->A; p.npx = e->head_h * e->head_w; p.napl = e->napl; p.ncls = e->ncls;
    p.head = e->d_head; p.labels = e->d_loss_labels; p.reg_targets = e->d_loss_regt; p.anchors = e->d_anchors;
    p.npos = e->d_loss_npos; p.partials = e->d_loss_partials; p.losses = e->d_loss_out;
    p.alpha = lc->alpha; p.gamma = lc->gamma; p.sigma = lc->sigma;
    for (int i = 0; i < 7; ++i) p.code_weight[i] = lc->code_weight[i];
    p.pos_cls_weight = lc->pos_class_weight; p.neg_cls_weight = lc->neg_class_weight;
    p.cls_weight = lc->classification_weight; p.loc_weight = lc->localization_weight; p.dir_weight = lc->direction_loss_weight;
    p.norm_by_num_positives = lc->norm_by_num_positives; p.encode_rad_error_by_sin = lc->encode_rad_error_by_sin;
    p.use_direction = lc->use_direction_classifier;
}

int train_state(pp_engine* e) {
    if (e->train) return PP_OK;
    auto* t = new pp_engine::TrainState();
    TrainShape& s = t->shape;
    s.nx = e->nx; s.ny = e->ny; s.nz = e->nz; s.C = e->C; s.F = e->F; s.FA = e->FA; s.T = e->T;
    s.max_voxels = e->cfg.max_voxels; s.with_dist = e->with_dist ? 1 : 0;
    s.vx = (float)e->cfg.voxel_size[0]; s.vy = (float)e->cfg.voxel_size[1];
    s.x_off = (float)(e->cfg.voxel_size[0] / 2 + e->cfg.pc_range[0]);
    s.y_off = (float)(e->cfg.voxel_size[1] / 2 + e->cfg.pc_range[1]);
    s.head_h = e->head_h; s.head_w = e->head_w; s.napl = e->napl; s.ncls = e->ncls; s.use_dir = e->use_dir ? 1 : 0;
    s.CC = e->CC;
    s.layers = e->layers;
    t->layout = train_layout(s, &t->n_params, &t->n_state);
    e->train = t;
    return PP_OK;
}

int train_buffers(pp_engine* e) {
    pp_engine::TrainState* t = e->train;
    if (t->buffers) return PP_OK;
    const TrainShape& s = t->shape;
    TrainCtx& cx = t->cx;
    const size_t B = (size_t)e->B, HW = (size_t)s.head_h * s.head_w;
    int st = PP_OK;
    auto A1 = [&](int r) { if (st == PP_OK) st = r; };
    A1(dalloc(e, &cx.pfn_feat, B * s.max_voxels * s.C));
    A1(dalloc(e, &cx.pfn_arg, B * s.max_voxels * s.C));
    A1(dalloc(e, &cx.pfn_stats, (size_t)2 * s.C));
    A1(dalloc(e, &cx.pfn_sums, (size_t)2 * s.C));
    A1(dalloc(e, &cx.pfn_nrows, (size_t)1));
    A1(dalloc(e, &cx.pfn_prefix, B + 1));
    A1(dalloc(e, &cx.pfn_rec, B * s.max_voxels * 3));
    // maps the fused forward kernel reads through a 3x3 window carry a PP_ZPAD_FLOATS header in front, the padding of
    // the convolution: NaN-filled for the pre-BatchNorm maps (relu(NaN * sc + sh) evaluates to 0 on the vector unit,
    // launch_sep_train), zero-filled for the tensors read as they are (canvas, block-final activations)
    auto dalloc_hdr = [&](float** p, size_t count, int fill = 0xff) -> int {
        float* raw = nullptr;
        int r = dalloc(e, &raw, count + PP_ZPAD_FLOATS);
        if (r == PP_OK && hipMemset(raw, fill, PP_ZPAD_FLOATS * sizeof(float)) != hipSuccess) r = PP_ERR_HIP;
        *p = raw ? raw + PP_ZPAD_FLOATS : nullptr;
        return r;
    };
    A1(dalloc_hdr(&cx.canvas, B * s.ny * s.nx * s.C, 0));
    A1(dalloc(e, &cx.dcanvas, B * s.ny * s.nx * s.C));
    size_t max_z = 1, max_d = 1;
    long pw16_words = 0;
    cx.lbuf.assign(s.layers.size(), TrainLayerBuf{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0});
    for (size_t i = 0; i < s.layers.size(); ++i) {
        const LayerDesc& l = s.layers[i];
        TrainLayerBuf& tb = cx.lbuf[i];
        if (l.kind == LAYER_SEP) {
            const size_t rows = B * l.out_h * l.out_w;
            A1(dalloc(e, &tb.D, rows * l.cin)); A1(dalloc_hdr(&tb.Z, rows * l.cout));
            // the activation tensor only where it is read as one: the layer in front of a transposed convolution
            if (i + 1 < s.layers.size() && s.layers[i + 1].kind == LAYER_DECONV) A1(dalloc_hdr(&tb.A, rows * l.cout, 0));
            tb.pw16_off = pw16_words;
            pw16_words += (long)2 * l.cin * l.cout;
            A1(dalloc(e, &tb.dA, rows * l.cout));
            max_z = std::max(max_z, rows * l.cout); max_d = std::max(max_d, rows * l.cin);
        } else if (l.kind == LAYER_DECONV) {
            const size_t n = B * l.in_h * l.in_w * l.k * l.k * l.cout;
            A1(dalloc(e, &tb.Z, n));
            max_z = std::max(max_z, n);
            tb.pw16_off = pw16_words;                       // the kernel as a GEMM operand [cin][k * k * cout], two float16 pieces
            pw16_words += (long)2 * l.cin * l.k * l.k * l.cout;
        } else {
            continue;
        }
        A1(dalloc(e, &tb.stats, (size_t)2 * l.cout)); A1(dalloc(e, &tb.sums, (size_t)2 * l.cout));
        A1(dalloc(e, &tb.coef, (size_t)l.cout));
    }
    A1(dalloc(e, &cx.cat, B * HW * s.CC)); A1(dalloc(e, &cx.dcat, B * HW * s.CC));
    A1(dalloc(e, &cx.head_w, (size_t)s.CC * PP_HEAD_COLS)); A1(dalloc(e, &cx.head_b, (size_t)PP_HEAD_COLS));
    A1(dalloc(e, &cx.dhead_w, (size_t)s.CC * PP_HEAD_COLS)); A1(dalloc(e, &cx.dhead_b, (size_t)2 * PP_HEAD_COLS));
    A1(dalloc(e, &cx.dZ, max_z)); A1(dalloc(e, &cx.dD, max_d));
    A1(dalloc(e, &cx.part, train_part_floats(s)));
    {   // one [2][N] row per 64-row tile of the largest forward product (per 32 rows for the fused separable launches)
        size_t need = 1;
        for (const LayerDesc& l : s.layers) {
            if (l.kind == LAYER_SEP) need = std::max(need, (B * l.out_h * l.out_w + 127) / 128 * 4 * 2 * (size_t)l.cout);
            else if (l.kind == LAYER_DECONV) need = std::max(need, ((B * l.in_h * l.in_w + 63) / 64 + 8) * 2 * (size_t)l.k * l.k * l.cout);
        }
        A1(dalloc(e, &cx.stat_part, need));
        cx.stat_part_floats = (long)need;
    }
    A1(dalloc(e, &cx.pw16, (size_t)std::max<long>(pw16_words, 8)));
    A1(dalloc(e, &cx.head_w16, (size_t)2 * s.CC * PP_HEAD_COLS));
    // split-K partial tiles + the regions of the step's deferred reductions (every weight gradient keeps its
    // partials until the end of the step): 64 MB at the reference's batch, 16 MB more per frame beyond 4
    // (round 4: capped -- the deferred regions are bounded by the SHAPES, not the batch: a weight-gradient product keeps at
    // most ~1 024 partial tiles of 64 x 64 floats, a depthwise layer 512 rows of 11 * cin, about two dozen of each per
    // step; an engine created for 512 frames used to take 8.6 GB here.  Past the cap the products split less and the
    // depthwise backward falls back to the shared scratch, train.hip)
    cx.gemm_part_floats = std::min<long>(std::max<long>(16l << 20, (long)B * (4l << 20)), 192l << 20);
    A1(dalloc(e, &cx.gemm_part, (size_t)cx.gemm_part_floats));
    if (st == PP_OK) st = ensure_loss_buffers(e);
    if (st == PP_OK) t->buffers = true;
    return st;
}

}  // namespace

int pp_head_loss(pp_handle e, const int32_t* labels, const float* reg_targets, int32_t batch,
                 const pp_loss_config* lc, float* losses, float* head_grad) {
    if (!e) return PP_ERR_ARG;
    if (!labels || !reg_targets || !lc || !losses) return fail(e, PP_ERR_ARG, "pp_head_loss: null argument");
    if (!e->anchors_ready) return fail(e, PP_ERR_STATE, "pp_head_loss: anchors not set");
    int st = check_batch(e, batch);
    if (st) return st;
    if (!(lc->sigma > 0.f)) return fail(e, PP_ERR_ARG, "pp_head_loss: sigma must be positive");
    if ((lc->use_direction_classifier != 0) != e->use_dir)
        return fail(e, PP_ERR_ARG, "pp_head_loss: loss config and engine disagree on use_direction_classifier");
    (void)hipSetDevice(e->device);
    const size_t npx = (size_t)e->head_h * e->head_w;
    if ((st = ensure_loss_buffers(e))) return st;
    HIPCHK(e, hipMemcpyAsync(e->d_loss_labels, labels, (size_t)batch * e->A * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->d_loss_regt, reg_targets, (size_t)batch * e->A * 7 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    LossParams p;
    fill_loss_params(e, lc, batch, p);
    p.head_grad = head_grad ? e->d_head_grad : nullptr;
    {
        ProfScope ps(e, "k_loss_pixels:loss+grad", true);
        if ((st = launch_head_loss(p, e->stream))) return fail(e, st, "pp_head_loss: %d anchors per pixel x %d classes not supported", e->napl, e->ncls);
    }
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(losses, e->d_loss_out, 8 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (head_grad)
        HIPCHK(e, hipMemcpyAsync(head_grad, e->d_head_grad, (size_t)batch * npx * PP_HEAD_COLS * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return PP_OK;
}

int pp_train_layout(pp_handle e, int32_t* n_entries, int64_t* n_param_floats, int64_t* n_state_floats) {
    if (!e) return PP_ERR_ARG;
    int st = train_state(e); if (st) return st;
    if (n_entries) *n_entries = (int32_t)e->train->layout.size();
    if (n_param_floats) *n_param_floats = e->train->n_params;
    if (n_state_floats) *n_state_floats = e->train->n_state;
    return PP_OK;
}

int pp_train_layout_entry(pp_handle e, int32_t i, const char** name, int64_t* offset, int64_t* size, int32_t* is_state) {
    if (!e) return PP_ERR_ARG;
    int st = train_state(e); if (st) return st;
    if (i < 0 || i >= (int)e->train->layout.size()) return fail(e, PP_ERR_ARG, "pp_train_layout_entry: index %d out of range", i);
    const TrainEntry& t = e->train->layout[i];
    if (name) *name = t.name.c_str();
    if (offset) *offset = t.offset;
    if (size) *size = t.size;
    if (is_state) *is_state = t.is_state;
    return PP_OK;
}

int pp_train_step_async(pp_handle e, const float* params_dev, float* grads_dev, float* state_dev, const int32_t* labels,
                        const float* reg_targets, int32_t batch, const pp_loss_config* lc) {
    if (!e) return PP_ERR_ARG;
    if (!params_dev || !grads_dev || !state_dev || !labels || !reg_targets || !lc)
        return fail(e, PP_ERR_ARG, "pp_train_step: null argument");
    if (e->train_pending) return fail(e, PP_ERR_STATE, "pp_train_step_async: the step before has not been waited for");
    if (!e->anchors_ready) return fail(e, PP_ERR_STATE, "pp_train_step: anchors not set");
    if (e->cur_batch < 1 || e->cur_batch != batch)
        return fail(e, PP_ERR_STATE, "pp_train_step: %d frames are resident, batch is %d (upload the frames first)", e->cur_batch, batch);
    if ((lc->use_direction_classifier != 0) != e->use_dir)
        return fail(e, PP_ERR_ARG, "pp_train_step: loss config and engine disagree on use_direction_classifier");
    if (!(lc->sigma > 0.f)) return fail(e, PP_ERR_ARG, "pp_train_step: sigma must be positive");
    (void)hipSetDevice(e->device);
    int st = train_state(e); if (st) return st;
    if ((st = train_buffers(e))) return st;
    if (e->up_pending || e->prevox_issued) {   // (prevox_issued: the handle served pp_detect_async passes before it trained)
        HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_up, 0));
        e->up_pending = false;
        e->prevox_issued = false;
    }
    prof_reset(e);
    // labels and regression targets travel on the copy stream (behind the points, if their upload is still queued
    // there) while voxeliser and forward pass run: the loss kernel is the first reader, the second half of the step
    // waits for ev_tgt.  (The previous step has been synchronised before it returned: nobody still reads the buffers.)
    // Issued AFTER the first half has been launched: a pageable source makes hipMemcpyAsync block the host, and the
    // GPU should be busy with the forward pass by then.
    auto upload_targets = [&]() -> int {
        HIPCHK(e, hipMemcpyAsync(e->d_loss_labels, labels, (size_t)batch * e->A * sizeof(int32_t), hipMemcpyHostToDevice, e->copy_stream));
        HIPCHK(e, hipMemcpyAsync(e->d_loss_regt, reg_targets, (size_t)batch * e->A * 7 * sizeof(float), hipMemcpyHostToDevice, e->copy_stream));
        HIPCHK(e, hipEventRecord(e->ev_tgt, e->copy_stream));
        HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_tgt, 0));
        return PP_OK;
    };
    pp_engine::TrainState* t = e->train;
    TrainCtx& cx = t->cx;
    cx.stream = e->stream;
    cx.pts_sorted = e->d_points_sorted; cx.offsets = e->d_offsets; cx.pillar_start = e->d_pstart; cx.pillar_cell = e->d_pcell;
    cx.npillars = e->d_npillars; cx.cellmap = e->d_cellmap;
    cx.head = e->d_head; cx.dhead = e->d_head_grad;
    LossParams lp;
    fill_loss_params(e, lc, batch, lp);
    // the step in two halves: 1 = voxelise + forward, 2 = loss + backward
    auto enqueue = [&](int max_n, int phase) -> int {
        if (phase & 1) {
            int r = run_voxelize(e, batch, max_n);
            if (r) return r;
        }
        return train_step(cx, t->shape, t->layout, params_dev, grads_dev, state_dev, batch, lp, phase);
    };
    bool launched = false;
    if (e->prof <= 0 && t->graph_state == 0 && graphs_enabled()) {
        const int bucket = graph_bucket(e, e->cur_max_n);
        pp_engine::TrainState::Graph& tg = t->graph[e->in_buf & 1];
        const bool hit = tg.exec != nullptr && tg.exec_bwd != nullptr && tg.batch == batch && tg.bucket == bucket &&
                         tg.zc == (e->zc ? 1 : 0) && tg.params == params_dev && tg.grads == grads_dev &&
                         tg.state == state_dev && memcmp(&tg.loss, lc, sizeof(pp_loss_config)) == 0;
        if (!hit) {
            if (tg.exec || tg.exec_bwd) {
                HIPCHK(e, hipStreamSynchronize(e->stream));
                if (tg.exec) (void)hipGraphExecDestroy(tg.exec);
                if (tg.exec_bwd) (void)hipGraphExecDestroy(tg.exec_bwd);
                tg.exec = tg.exec_bwd = nullptr;
            }
            bool all_ok = true;
            for (int phase = 1; phase <= 2 && all_ok; ++phase) {
                hipGraph_t g = nullptr;
                bool ok = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                st = ok ? enqueue(bucket, phase) : PP_ERR_HIP;
                if (ok && hipStreamEndCapture(e->stream, &g) != hipSuccess) { ok = false; g = nullptr; }
                if (ok && st == PP_ERR_UNSUPPORTED) {
                    if (g) (void)hipGraphDestroy(g);
                    if (tg.exec) { (void)hipGraphExecDestroy(tg.exec); tg.exec = nullptr; }
                    return fail(e, st, "pp_train_step: configuration not supported by the training kernels");
                }
                hipGraphExec_t* slot = (phase == 1) ? &tg.exec : &tg.exec_bwd;
                if (!(ok && st == PP_OK && g != nullptr && hipGraphInstantiate(slot, g, nullptr, nullptr, 0) == hipSuccess)) {
                    *slot = nullptr;
                    all_ok = false;
                }
                if (g) (void)hipGraphDestroy(g);
            }
            if (all_ok) {
                tg.batch = batch; tg.bucket = bucket; tg.zc = e->zc ? 1 : 0;
                tg.params = params_dev; tg.grads = grads_dev; tg.state = state_dev; tg.loss = *lc;
                ++t->n_captures;
            } else {
                if (tg.exec) (void)hipGraphExecDestroy(tg.exec);
                if (tg.exec_bwd) (void)hipGraphExecDestroy(tg.exec_bwd);
                tg.exec = tg.exec_bwd = nullptr;
                t->graph_state = -1;
                (void)hipGetLastError();
            }
        }
        if (tg.exec != nullptr && tg.exec_bwd != nullptr) {
            HIPCHK(e, hipGraphLaunch(tg.exec, e->stream));
            if ((st = upload_targets())) return st;
            HIPCHK(e, hipGraphLaunch(tg.exec_bwd, e->stream));
            ++t->n_replays;
            launched = true;
            st = PP_OK;
        }
    }
    if (!launched) {
        ProfScope ps(e, nullptr);
        st = enqueue(e->cur_max_n, 1);
        if (st == PP_OK) st = upload_targets();
        if (st == PP_OK) st = enqueue(e->cur_max_n, 2);
    }
    if (st) return fail(e, st, "pp_train_step: configuration not supported by the training kernels");
    HIPCHK(e, hipGetLastError());
    if (!e->h_train_losses && hipHostMalloc((void**)&e->h_train_losses, 8 * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(e, PP_ERR_HIP, "pp_train_step: hipHostMalloc failed");
    }
    HIPCHK(e, hipMemcpyAsync(e->h_train_losses, e->d_loss_out, 8 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    // this input buffer (and its zero-copy descriptor) is free again once the step is through: the NEXT batch may be
    // uploaded into the other one while this step runs (pp_upload_points_async between _async and _wait)
    HIPCHK(e, hipEventRecord(e->ev_read[e->in_buf], e->stream));
    e->results_batch = 0;          // the head map now holds training-mode outputs, not detections
    e->cls_plane_live = false;
    e->train_pending = true;
    t->last_batch = batch;
    return PP_OK;
}

int pp_train_step_wait(pp_handle e, float* losses) {
    if (!e) return PP_ERR_ARG;
    if (!losses) return fail(e, PP_ERR_ARG, "pp_train_step_wait: losses is NULL");
    if (!e->train_pending) return fail(e, PP_ERR_STATE, "pp_train_step_wait: no step in flight");
    (void)hipSetDevice(e->device);
    e->train_pending = false;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    memcpy(losses, e->h_train_losses, 8 * sizeof(float));
    return PP_OK;
}

int pp_train_step(pp_handle e, const float* params_dev, float* grads_dev, float* state_dev, const int32_t* labels,
                  const float* reg_targets, int32_t batch, const pp_loss_config* lc, float* losses) {
    if (!e) return PP_ERR_ARG;
    if (!losses) return fail(e, PP_ERR_ARG, "pp_train_step: null argument");
    int st = pp_train_step_async(e, params_dev, grads_dev, state_dev, labels, reg_targets, batch, lc);
    if (st) return st;
    return pp_train_step_wait(e, losses);
}

int pp_stream(pp_handle e, void** stream) {
    if (!e) return PP_ERR_ARG;
    if (!stream) return fail(e, PP_ERR_ARG, "pp_stream: stream is NULL");
    *stream = (void*)e->stream;
    return PP_OK;
}

int pp_train_fetch_decisions(pp_handle e, int32_t layer, uint8_t* relu_mask, int64_t capacity, int64_t* count) {
    if (!e) return PP_ERR_ARG;
    if (!count) return fail(e, PP_ERR_ARG, "pp_train_fetch_decisions: count is NULL");
    if (!e->train || !e->train->buffers || e->train->last_batch < 1)
        return fail(e, PP_ERR_STATE, "pp_train_fetch_decisions: no training step to tap");
    if (e->train_pending) return fail(e, PP_ERR_STATE, "pp_train_fetch_decisions: the step has not been waited for");
    (void)hipSetDevice(e->device);
    pp_engine::TrainState* t = e->train;
    const size_t B = (size_t)t->last_batch;
    if (layer < 0) {      // the PFN: winning row per (pillar slot, channel), int32 stored as 4 bytes each
        const int64_t n = (int64_t)B * t->shape.max_voxels * t->shape.C;
        *count = n;
        if (!relu_mask) return PP_OK;
        if (capacity < n * 4) return fail(e, PP_ERR_ARG, "pp_train_fetch_decisions: %lld bytes needed", (long long)(n * 4));
        HIPCHK(e, hipStreamSynchronize(e->stream));
        HIPCHK(e, hipMemcpy(relu_mask, t->cx.pfn_arg, (size_t)n * 4, hipMemcpyDeviceToHost));
        return PP_OK;
    }
    int k = -1;
    for (size_t i = 0; i < t->shape.layers.size(); ++i) {
        const LayerDesc& l = t->shape.layers[i];
        if (l.kind != LAYER_SEP && l.kind != LAYER_DECONV) continue;
        if (++k != layer) continue;
        const int64_t n = (l.kind == LAYER_SEP) ? (int64_t)B * l.out_h * l.out_w * l.cout
                                                : (int64_t)B * l.in_h * l.in_w * l.k * l.k * l.cout;
        *count = n;
        if (!relu_mask) return PP_OK;
        if (capacity < n) return fail(e, PP_ERR_ARG, "pp_train_fetch_decisions: %lld bytes needed", (long long)n);
        unsigned char* d = nullptr;
        HIPCHK(e, hipMalloc(&d, (size_t)n));
        launch_relu_mask(t->cx.lbuf[i].Z, t->cx.lbuf[i].coef, (long)n, l.cout, d, e->stream);
        hipError_t he = hipStreamSynchronize(e->stream);
        if (he == hipSuccess) he = hipMemcpy(relu_mask, d, (size_t)n, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        if (he != hipSuccess) return fail(e, PP_ERR_HIP, "pp_train_fetch_decisions: %s", hipGetErrorString(he));
        return PP_OK;
    }
    return fail(e, PP_ERR_ARG, "pp_train_fetch_decisions: layer %d out of range", layer);
}

int pp_train_graph_stats(pp_handle e, int32_t* captures, int32_t* replays) {
    if (!e) return PP_ERR_ARG;
    if (captures) *captures = e->train ? e->train->n_captures : 0;
    if (replays) *replays = e->train ? e->train->n_replays : 0;
    return PP_OK;
}

int pp_adamw_step_device(int device, void* stream, float* params, const float* grads, float* m, float* v,
                         int64_t n, float lr_t, float beta1, float beta2, float epsilon, float weight_decay) {
    if (n < 0 || (n > 0 && (!params || !grads || !m || !v))) return fail(nullptr, PP_ERR_ARG, "pp_adamw_step_device: bad argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, PP_ERR_HIP, "pp_adamw_step_device: hipSetDevice(%d) failed", device);
    launch_adamw(params, grads, m, v, n, lr_t, beta1, beta2, epsilon, weight_decay, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return fail(nullptr, PP_ERR_HIP, "pp_adamw_step_device: launch failed");
    return PP_OK;
}

int pp_device_mem_free(pp_handle e, int64_t* free_bytes) {
    if (!e || !free_bytes) return PP_ERR_ARG;
    (void)hipSetDevice(e->device);
    size_t fr = 0, tot = 0;
    HIPCHK(e, hipMemGetInfo(&fr, &tot));
    *free_bytes = (int64_t)fr;
    return PP_OK;
}

// measurement helper: device-to-device copy rate on the handle's stream (the bench states the HBM figure it
// measured in the run beside the spec constant its roofline fractions use)
int pp_device_copy_bench(pp_handle e, int64_t bytes, int32_t reps, float* gbytes_per_s) {
    if (!e || !gbytes_per_s || bytes <= 0 || reps <= 0) return fail(e, PP_ERR_ARG, "pp_device_copy_bench: bad argument");
    (void)hipSetDevice(e->device);
    void *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, (size_t)bytes) != hipSuccess || hipMalloc(&dst, (size_t)bytes) != hipSuccess) {
        if (src) (void)hipFree(src);
        (void)hipGetLastError();
        return fail(e, PP_ERR_HIP, "pp_device_copy_bench: hipMalloc(2 x %lld) failed", (long long)bytes);
    }
    int st = PP_OK;
    float ms = 0.f;
    if (hipMemsetAsync(src, 1, (size_t)bytes, e->stream) != hipSuccess ||
        hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, e->stream) != hipSuccess ||     // warm-up
        hipEventRecord(e->t0, e->stream) != hipSuccess) st = PP_ERR_HIP;
    for (int i = 0; i < reps && st == PP_OK; ++i)
        if (hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, e->stream) != hipSuccess) st = PP_ERR_HIP;
    if (st == PP_OK && (hipEventRecord(e->t1, e->stream) != hipSuccess || hipEventSynchronize(e->t1) != hipSuccess ||
                        hipEventElapsedTime(&ms, e->t0, e->t1) != hipSuccess)) st = PP_ERR_HIP;
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(src);
    (void)hipFree(dst);
    if (st != PP_OK) { (void)hipGetLastError(); return fail(e, st, "pp_device_copy_bench: copy failed"); }
    *gbytes_per_s = (float)(2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);   // read + write
    return PP_OK;
}

int pp_device_info(pp_handle e, char* name, int32_t name_capacity, int32_t* compute_units, int64_t* hbm_bytes) {
    if (!e) return PP_ERR_ARG;
    hipDeviceProp_t prop;
    HIPCHK(e, hipGetDeviceProperties(&prop, e->device));
    if (name && name_capacity > 0) {
        snprintf(name, (size_t)name_capacity, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return PP_OK;
}

}  // extern "C"
