// rtdm_api.hip -- the C ABI declared in include/rtdm.h: handle management, parameter validation
// (the checks cv::StereoBM::compute performs), geometry, staging copies and
// the launch sequence  K1 prefilter -> K2 search -> K3 left-right check -> K4 speckle filter.
// There is no CPU fallback anywhere in this file: every entry point needs a HIP device.
#include "../../include/rtdm.h"
#include "rtdm_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

using namespace rtdm;

static thread_local std::string g_hip_err;

#define HIPC(expr)                                                                               \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);                       \
            return RTDM_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

struct StageEvent { hipEvent_t a, b; int stage; int frames; };

// Host entry points hand the caller's (possibly page-locked) planes to async copies: whatever way they leave, nothing may
// still be in flight from / to those planes.  Success paths synchronise themselves and disarm the guard.
struct DrainOnError {
    hipStream_t s;
    bool armed = true;
    ~DrainOnError()
    {
        if (!armed) return;
        const std::string first = g_hip_err;
        (void)hipStreamSynchronize(s); (void)hipGetLastError();
        g_hip_err = first;
    }
};

// Shape of a batched search launch, compared field by field (tune_strips).
struct TuneKey {
    int W, H, n, ncols, nrows, fuse;
    bool operator==(const TuneKey& o) const { return W == o.W && H == o.H && n == o.n && ncols == o.ncols && nrows == o.nrows && fuse == o.fuse; }
};
struct TuneEntry { TuneKey key; int strips; };   // strips: 0 = seen once, not measured yet; -1 = measuring failed

// A lane = one HIP stream plus its slice of the per-handle workspace.  A batch is cut into pieces that
// alternate between two lanes, so that the latency-bound row kernels (left-right check, speckle
// filter) of one piece run beside the VALU-bound SAD search of the other.
struct Lane {
    hipStream_t stream;
    hipEvent_t done;
    hipStream_t side;              // the border-column search runs here, next to the tile search (RTDM_BORDER_ASYNC)
    hipEvent_t fork, join;
    hipStream_t back;              // two-lane mode: left-right check and speckle filter of the lane's piece run here
    hipEvent_t mid;                // ... after this event (the piece's search is complete)
    uint8_t *dLp, *dRp;
    int32_t *dCost, *dLabel, *dSize, *dRowCnt;
    uint32_t* dRuns;
    int16_t* dHead;
    int16_t* dOut;                 // internal disparity plane of the lane (rows of Ws elements)
};

struct rtdm_bm {
    rtdm_bm_params p;
    int maxW, maxH, maxB, device;
    int roi1[4], roi2[4];
    hipStream_t stream;
    Lane lane[2];
    int nlanes, laneB;             // frames per lane piece
    hipEvent_t evIn;
    hipEvent_t evBand[4];          // rtdm_bm_compute: one per band of the result on its way back
    hipStream_t sIn, sOut;         // rtdm_bm_compute_batch: copies in / out beside the compute stream
    hipEvent_t evH2D[2], evComp[2], evD2H[2];   // ... one set per half of the staging planes
    size_t ppitch;                 // pitch of the internal 8-bit planes
    uint8_t *dLp, *dRp;            // prefiltered planes   [maxB][maxH][ppitch]
    uint8_t *dInL, *dInR;          // staging for the host entry points
    int16_t* dOut;                 //                      [maxB][maxH][maxW]
    std::vector<TuneEntry> tuned;  // measured strip counts per work shape, least recently used first (<= 16 entries)
    long tune_shapes, tune_launches;   // rtdm_bm_get_tuner_stats
    int32_t *dCost, *dLabel, *dSize, *dRowCnt;
    uint32_t* dRuns;
    int16_t* dHead;
    uint8_t* dMask;                // staging for rtdm_bm_compute_depth: mask plane + reduction scratch
    void* dDepth;
    uint8_t* hStage;               // page-locked staging for the single-frame host entry point
    size_t hStageBytes;
    bool profiling;
    std::vector<StageEvent> pending;
    double stage_ms[RTDM_NUM_STAGES];
    long stage_launches[RTDM_NUM_STAGES], stage_frames[RTDM_NUM_STAGES];
    std::string variant;
};

struct rtdm_morph {
    int W, H, maxB, device;
    hipStream_t stream;
    uint8_t *hIn, *hOut;           // page-locked host buffers handed to the application
    uint8_t *dIn, *dOut, *dT0, *dT1;
};

extern "C" {

const char* rtdm_strerror(int s)
{
    switch (s) {
        case RTDM_OK: return "ok";
        case RTDM_ERR_BAD_PARAM: return "invalid StereoBM parameter";
        case RTDM_ERR_BAD_SIZE: return "invalid frame size or pitch";
        case RTDM_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case RTDM_ERR_HIP: return "HIP runtime error";
        case RTDM_ERR_NOMEM: return "out of memory";
        case RTDM_ERR_UNSUPPORTED: return "configuration not supported by this build";
        case RTDM_ERR_NULL: return "null pointer";
        default: return "unknown rtdm status";
    }
}

const char* rtdm_last_hip_error(void) { return g_hip_err.c_str(); }
int rtdm_abi_version(void) { return RTDM_ABI_VERSION; }

int rtdm_device_count(int* count)
{
    if (!count) return RTDM_ERR_NULL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return n > 0 ? RTDM_OK : RTDM_ERR_NO_DEVICE;
}

void rtdm_bm_default_params(rtdm_bm_params* p, int numDisparities)
{
    if (!p) return;
    p->preFilterCap = 31; p->blockSize = 13; p->minDisparity = 0; p->numDisparities = numDisparities;
    p->textureThreshold = 10; p->uniquenessRatio = 10; p->speckleWindowSize = 100; p->speckleRange = 32;
    p->disp12MaxDiff = 1; p->legacy_right_clamp = 0;
}

static int validate_params(const rtdm_bm_params& p)
{
    if (p.preFilterCap < 1 || p.preFilterCap > 63) return RTDM_ERR_BAD_PARAM;
    if (p.blockSize < 5 || p.blockSize > 255 || (p.blockSize & 1) == 0) return RTDM_ERR_BAD_PARAM;
    if (p.numDisparities <= 0 || p.numDisparities % 16 != 0) return RTDM_ERR_BAD_PARAM;
    if (p.textureThreshold < 0 || p.uniquenessRatio < 0) return RTDM_ERR_BAD_PARAM;
    // the x16 fixed-point output is 16 bits wide: (minDisparity - 1) * 16 .. (minDisparity + numDisparities) * 16 must fit
    if (p.minDisparity < -2047 || (long)p.minDisparity + p.numDisparities > 2047) return RTDM_ERR_BAD_PARAM;
    if (p.legacy_right_clamp != 0 && p.legacy_right_clamp != 1) return RTDM_ERR_BAD_PARAM;
    return RTDM_OK;
}

static int use_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RTDM_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return RTDM_ERR_NO_DEVICE;
    HIPC(hipSetDevice(device));
    return RTDM_OK;
}

int rtdm_bm_create(const rtdm_bm_params* params, int max_width, int max_height, int max_batch,
                   int device, rtdm_bm** out)
{
    if (!params || !out) return RTDM_ERR_NULL;
    *out = nullptr;
    int rc = validate_params(*params);
    if (rc) return rc;
    if (max_width <= 0 || max_height <= 0 || max_batch <= 0 || max_width > 32767 || max_height > 32767)
        return RTDM_ERR_BAD_SIZE;
    if ((long)max_batch * max_width * max_height >= (1L << 31)) return RTDM_ERR_BAD_SIZE;
    if (max_width > 4096) return RTDM_ERR_UNSUPPORTED;   // row kernels keep whole rows in LDS
    rc = use_device(device);
    if (rc) return rc;
    rtdm_bm* bm = new (std::nothrow) rtdm_bm();
    if (!bm) return RTDM_ERR_NOMEM;
    bm->p = *params; bm->maxW = max_width; bm->maxH = max_height; bm->maxB = max_batch; bm->device = device;
    for (int i = 0; i < 4; ++i) bm->roi1[i] = bm->roi2[i] = 0;
    bm->profiling = false;
    bm->tune_shapes = bm->tune_launches = 0;
    for (int i = 0; i < RTDM_NUM_STAGES; ++i) { bm->stage_ms[i] = 0; bm->stage_launches[i] = 0; bm->stage_frames[i] = 0; }
    bm->ppitch = ((size_t)max_width + 63) & ~(size_t)63;
    const size_t plane = bm->ppitch * max_height * (size_t)max_batch;
    // per-pixel workspace and the internal disparity plane use rows of Ws = max_width rounded up to 8 elements
    const size_t px = (size_t)((max_width + 7) & ~7) * max_height * max_batch;
    hipError_t e = hipStreamCreateWithFlags(&bm->stream, hipStreamNonBlocking);
    // + slack: the search kernels stage whole dwords of whole tiles and may read past the last row's end
    // (one allocation, the right planes behind the left ones: k_search_ring addresses both from the left plane's rows)
    const size_t plane_al = (plane + 1024 + 255) & ~(size_t)255;
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dLp, 2 * plane_al);
    if (e == hipSuccess) bm->dRp = bm->dLp + plane_al;
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dInL, plane);
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dInR, plane);
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dOut, px * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dCost, px * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dLabel, px * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dSize, px * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dRuns, px * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dHead, px * sizeof(int16_t));
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dMask, (size_t)max_width * max_height);
    if (e == hipSuccess) e = hipMalloc(&bm->dDepth, depth_scratch_bytes(RTDM_MAX_REGIONS, max_height));
    bm->hStageBytes = 2 * bm->ppitch * (size_t)max_height + (size_t)((max_width + 7) & ~7) * max_height * (sizeof(int16_t) + 1) + 1024;
    if (e == hipSuccess) e = hipHostMalloc((void**)&bm->hStage, bm->hStageBytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&bm->dRowCnt, (size_t)max_batch * max_height * sizeof(int32_t));
    if (e != hipSuccess) {
        g_hip_err = std::string("rtdm_bm_create: ") + hipGetErrorString(e);
        rtdm_bm_destroy(bm);
        return e == hipErrorOutOfMemory ? RTDM_ERR_NOMEM : RTDM_ERR_HIP;
    }
    {   // lanes: slices of the workspace (lane 1 starts laneB frames in)
        const char* env = getenv("RTDM_LANES");
        // opt-in.  Measured in round 2 with the ring kernel (profiles/r02_two_lane_pipeline_ab.txt): the row kernels of piece k
        // under the search of piece k+1 give 51.8 k pairs/s with 2 pieces and 51.2 k with 4 against 52.1 k on one lane -- a
        // search wave loses as many issue slots to a co-resident row-kernel wave as the overlap hides; with the front stream at
        // high and the back streams at low priority the row kernels starve (left-right check 1.7 -> 4.5 ms) and the search
        // still slows by 8 %: 47.7 k.
        bm->nlanes = (max_batch >= 2 && env && atoi(env) == 2) ? 2 : 1;
        bm->laneB = bm->nlanes == 2 ? (max_batch + 1) / 2 : max_batch;
        HIPC(hipEventCreateWithFlags(&bm->evIn, hipEventDisableTiming));
        for (auto& e : bm->evBand) HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPC(hipStreamCreateWithFlags(&bm->sIn, hipStreamNonBlocking));
        HIPC(hipStreamCreateWithFlags(&bm->sOut, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPC(hipEventCreateWithFlags(&bm->evH2D[k], hipEventDisableTiming));
            HIPC(hipEventCreateWithFlags(&bm->evComp[k], hipEventDisableTiming));
            HIPC(hipEventCreateWithFlags(&bm->evD2H[k], hipEventDisableTiming));
        }
        for (int k = 0; k < bm->nlanes; ++k) {
            Lane& ln = bm->lane[k];
            const size_t fo = (size_t)k * bm->laneB;                       // first frame of the slice
            const size_t po = fo * ((max_width + 7) & ~7) * max_height;     // in workspace pixels
            HIPC(hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking));
            HIPC(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
            HIPC(hipStreamCreateWithFlags(&ln.side, hipStreamNonBlocking));
            HIPC(hipEventCreateWithFlags(&ln.fork, hipEventDisableTiming));
            HIPC(hipEventCreateWithFlags(&ln.join, hipEventDisableTiming));
            HIPC(hipStreamCreateWithFlags(&ln.back, hipStreamNonBlocking));
            HIPC(hipEventCreateWithFlags(&ln.mid, hipEventDisableTiming));
            ln.dLp = bm->dLp + fo * bm->ppitch * max_height; ln.dRp = bm->dRp + fo * bm->ppitch * max_height;
            ln.dCost = bm->dCost + po; ln.dLabel = bm->dLabel + po; ln.dSize = bm->dSize + po;
            ln.dRuns = bm->dRuns + po; ln.dHead = bm->dHead + po; ln.dOut = bm->dOut + po; ln.dRowCnt = bm->dRowCnt + fo * max_height;
        }
    }
    *out = bm;
    return RTDM_OK;
}

void rtdm_bm_destroy(rtdm_bm* bm)
{
    if (!bm) return;
    (void)hipSetDevice(bm->device);
    if (bm->stream) (void)hipStreamSynchronize(bm->stream);
    for (int k = 0; k < bm->nlanes; ++k) {
        if (bm->lane[k].stream) { (void)hipStreamSynchronize(bm->lane[k].stream); (void)hipStreamDestroy(bm->lane[k].stream); }
        if (bm->lane[k].done) (void)hipEventDestroy(bm->lane[k].done);
        if (bm->lane[k].side) { (void)hipStreamSynchronize(bm->lane[k].side); (void)hipStreamDestroy(bm->lane[k].side); }
        if (bm->lane[k].fork) (void)hipEventDestroy(bm->lane[k].fork);
        if (bm->lane[k].join) (void)hipEventDestroy(bm->lane[k].join);
        if (bm->lane[k].back) { (void)hipStreamSynchronize(bm->lane[k].back); (void)hipStreamDestroy(bm->lane[k].back); }
        if (bm->lane[k].mid) (void)hipEventDestroy(bm->lane[k].mid);
    }
    if (bm->evIn) (void)hipEventDestroy(bm->evIn);
    for (auto& e : bm->evBand) if (e) (void)hipEventDestroy(e);
    if (bm->sIn) { (void)hipStreamSynchronize(bm->sIn); (void)hipStreamDestroy(bm->sIn); }
    if (bm->sOut) { (void)hipStreamSynchronize(bm->sOut); (void)hipStreamDestroy(bm->sOut); }
    for (int k = 0; k < 2; ++k) {
        if (bm->evH2D[k]) (void)hipEventDestroy(bm->evH2D[k]);
        if (bm->evComp[k]) (void)hipEventDestroy(bm->evComp[k]);
        if (bm->evD2H[k]) (void)hipEventDestroy(bm->evD2H[k]);
    }
    for (auto& ev : bm->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    void* bufs[] = {bm->dLp, bm->dInL, bm->dInR, bm->dOut, bm->dCost, bm->dLabel, bm->dSize, bm->dRuns, bm->dRowCnt, bm->dHead, bm->dMask, bm->dDepth};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (bm->hStage) (void)hipHostFree(bm->hStage);
    if (bm->stream) (void)hipStreamDestroy(bm->stream);
    delete bm;
}

int rtdm_bm_set_roi(rtdm_bm* bm, int which, int x, int y, int width, int height)
{
    if (!bm) return RTDM_ERR_NULL;
    if (which != 1 && which != 2) return RTDM_ERR_BAD_PARAM;
    int* r = which == 1 ? bm->roi1 : bm->roi2;
    r[0] = x; r[1] = y; r[2] = width; r[3] = height;
    return RTDM_OK;
}

int rtdm_bm_get_params(const rtdm_bm* bm, rtdm_bm_params* out)
{
    if (!bm || !out) return RTDM_ERR_NULL;
    *out = bm->p;
    return RTDM_OK;
}

// offsets, valid rectangle (getValidDisparityROI, clipped to the image and
// to rows that have a full window).  Returns false when the whole frame is FILTERED.
static bool make_geom(const rtdm_bm* bm, int W, int H, BMGeom* g)
{
    const rtdm_bm_params& p = bm->p;
    g->W = W; g->H = H; g->Ws = (W + 7) & ~7; g->D = p.numDisparities; g->minD = p.minDisparity;
    g->w = p.blockSize; g->r = p.blockSize / 2;
    g->cap = p.preFilterCap; g->tex = p.textureThreshold; g->uniq = p.uniquenessRatio;
    g->lofs = std::max(g->D - 1 + g->minD, 0);
    g->rofs = -std::min(g->D - 1 + g->minD, 0);
    g->width1 = W - g->rofs - g->D + 1;
    g->filtered = (g->minD - 1) * 16;
    g->want_cost = p.disp12MaxDiff >= 0;
    g->cost16 = 2L * p.preFilterCap * p.blockSize * p.blockSize < 65536;
    g->mask_cols = p.disp12MaxDiff < 0;
    g->legacy = p.legacy_right_clamp;
    int r1[4] = {0, 0, W, H}, r2[4] = {0, 0, W, H};
    if (bm->roi1[2] > 0 && bm->roi1[3] > 0) std::copy(bm->roi1, bm->roi1 + 4, r1);
    if (bm->roi2[2] > 0 && bm->roi2[3] > 0) std::copy(bm->roi2, bm->roi2 + 4, r2);
    const int maxD = g->minD + g->D - 1;
    int xmin = std::max(r1[0], r2[0] + maxD) + g->r;
    int xmax = std::min(r1[0] + r1[2], r2[0] + r2[2] - g->minD) - g->r;
    int ymin = std::max(r1[1], r2[1]) + g->r;
    int ymax = std::min(r1[1] + r1[3], r2[1] + r2[3]) - g->r;
    xmin = std::max(xmin, 0); xmax = std::min(xmax, W);
    ymin = std::max(ymin, g->r); ymax = std::min(ymax, H - g->r);
    g->vx0 = xmin; g->vx1 = xmax; g->vy0 = ymin; g->vy1 = ymax;
    // Columns worth searching (estimator.cpp:54 shrinks the rectangle every frame with setROI1):
    // validateDisparity lets column x vote for x - round(d/16) in [x - maxD - 1, x - minD + 1] and
    // reads the votes of x - floor/ceil(d/16), so a valid column depends on searched columns at
    // most D + |minD| + 2 away.  Everything else is masked to FILTERED anyway.
    const int reach = (p.disp12MaxDiff >= 0) ? g->D + std::abs(g->minD) + 2 : 0;
    g->cx0 = std::max(g->lofs, xmin - reach);
    g->cx1 = std::min(std::min(W, g->lofs + g->width1), xmax + reach);
    if (g->lofs >= W || g->rofs >= W || g->width1 < 1) return false;
    return xmax > xmin && ymax > ymin;
}

static void stage_begin(rtdm_bm* bm, int stage, int frames, hipStream_t s, StageEvent* ev)
{
    if (!bm->profiling) return;
    ev->stage = stage; ev->frames = frames;
    (void)hipEventCreate(&ev->a); (void)hipEventCreate(&ev->b);
    (void)hipEventRecord(ev->a, s);
}
static void stage_end(rtdm_bm* bm, hipStream_t s, StageEvent* ev)
{
    if (!bm->profiling) return;
    (void)hipEventRecord(ev->b, s);
    bm->pending.push_back(*ev);
}

static int chunk_front(rtdm_bm* bm, const Lane& ln, int n, Plane8 L, Plane8 R, int W, int H, Plane16W disp, hipStream_t s,
                       BMGeom* gout, bool* anyout);
static int chunk_back(rtdm_bm* bm, const Lane& ln, int n, int W, int H, Plane16W disp, const BMGeom& g, hipStream_t s);

// Row strips per frame for the fast search of a batch, chosen by measurement and remembered in the handle: the model
// (fast_strips_model) is right on average, but neighbouring strip counts differ by up to 5 % through scheduling effects it
// cannot see (profiles/r01_xcd_mapping_sweep.txt).  The search only writes its own outputs, so timing it a few times on
// the caller's data is harmless.  The key is the SHAPE of the work (frame size, batch, searched columns and rows as
// counts, not positions): a caller that moves a same-sized ROI around (estimator.cpp:53-54) keeps its entry.  A shape is
// measured the SECOND time it is seen -- a caller whose ROI changes size every call never pays the ~30 extra launches and
// the stream synchronisation -- and the table is a 16-entry LRU.  Small batches keep the model.  RTDM_AUTOTUNE=0: off.
static int tune_strips(rtdm_bm* bm, const Lane& ln, Plane8 Lpr, Plane8 Rpr, Plane16W disp, const BMGeom& g, int n, hipStream_t s, bool fuse, bool ring)
{
    const auto launch = [&](int c) {
        if (ring) launch_search_ring(Lpr, Rpr, disp, ln.dCost, g, n, s, c, fuse);
        else launch_search_fast(Lpr, Rpr, disp, ln.dCost, g, n, s, fuse, c);
    };
    static const bool enabled = env_int("RTDM_AUTOTUNE", 1) != 0 && getenv("RTDM_FAST_WGS") == nullptr;
    if (!enabled || n < 16) return 0;
    TuneKey key{g.W, g.H, n, g.cx1 - g.cx0, g.vy1 - g.vy0, (fuse ? 1 : 0) | (ring ? ring_lanes_per_pixel(g) : 0)};
    for (size_t i = 0; i < bm->tuned.size(); ++i) {
        if (!(bm->tuned[i].key == key)) continue;
        TuneEntry e = bm->tuned[i];
        bm->tuned.erase(bm->tuned.begin() + (long)i);              // most recently used goes to the back
        if (e.strips == 0) {
            e.strips = -1;                                          // being measured: a failure below leaves the model in charge
            ++bm->tune_shapes;
            const int model = ring ? ring_strips_model(g, n) : fast_strips_model(g, n), cap = (g.vy1 - g.vy0 + 15) / 16;
            int best = model;
            float best_ms = 1e30f;
            hipEvent_t a, b;
            if (hipEventCreate(&a) == hipSuccess) {
                if (hipEventCreate(&b) == hipSuccess) {
                    static const float f[] = {0.6f, 0.7f, 0.8f, 0.9f, 1.0f, 1.1f, 1.2f, 1.35f, 1.5f, 1.75f};
                    int seen[10], nseen = 0;
                    for (float fk : f) {
                        const int c = std::max(1, std::min(cap, (int)(model * fk + 0.5f)));
                        bool dup = false;
                        for (int k = 0; k < nseen; ++k) dup |= seen[k] == c;
                        if (dup) continue;
                        seen[nseen++] = c;
                        launch(c);          // warm
                        bm->tune_launches += 3;
                        float ms = 1e30f;
                        for (int rep = 0; rep < 2; ++rep) {
                            (void)hipEventRecord(a, s);
                            launch(c);
                            (void)hipEventRecord(b, s);
                            float t = 0.f;
                            if (hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&t, a, b) == hipSuccess) ms = std::min(ms, t);
                        }
                        if (ms < best_ms) { best_ms = ms; best = c; }
                    }
                    e.strips = best;
                    (void)hipEventDestroy(b);
                }
                (void)hipEventDestroy(a);
            }
        }
        bm->tuned.push_back(e);
        return e.strips > 0 ? e.strips : 0;
    }
    if (bm->tuned.size() >= 16) bm->tuned.erase(bm->tuned.begin());  // least recently used
    bm->tuned.push_back(TuneEntry{key, 0});                           // first sighting: remember, keep the model
    return 0;
}

// One chunk (n <= maxB) of device-resident frames, enqueued on `s`.  The row kernels move 8 columns per 128-bit
// access and let a ragged last chunk spill into the row padding, so they need 16-byte aligned rows of at least
// W rounded up to 8 elements whose padding is ours to write.  A caller's plane qualifies only if W % 8 == 0 and it
// is aligned; anything else (the reference's own crops are 233, 534 and 934 columns wide) runs on the lane's
// internal plane and is copied out at the end.
// `back` (two-lane mode): the stream the chunk's row kernels run on, behind the event ln.mid recorded after the search;
// nullptr = everything on `s`.
static int run_chunk(rtdm_bm* bm, const Lane& ln, int n, Plane8 L, Plane8 R, int W, int H, Plane16W out, hipStream_t s,
                     hipStream_t back = nullptr)
{
    const size_t Ws = (size_t)((W + 7) & ~7);
    // (the lane's own plane, or a frame-aligned part of it: rows of Ws elements whose padding is ours)
    const bool internal = out.pitch_e == Ws && out.base >= ln.dOut && out.base < ln.dOut + (size_t)bm->laneB * Ws * (size_t)H &&
                          (size_t)(out.base - ln.dOut) % (Ws * (size_t)H) == 0;
    const bool direct = internal || ((W & 7) == 0 && (((size_t)out.base | (out.pitch_e * 2) | (out.frame_e * 2)) & 15) == 0);
    const Plane16W disp = direct ? out : Plane16W{ln.dOut, Ws, Ws * (size_t)H};
    BMGeom g;
    bool any = false;
    int rc = chunk_front(bm, ln, n, L, R, W, H, disp, s, &g, &any);
    if (rc) return rc;
    hipStream_t b = s;
    if (back) {
        HIPC(hipEventRecord(ln.mid, s));
        HIPC(hipStreamWaitEvent(back, ln.mid, 0));
        b = back;
    }
    if (any) { rc = chunk_back(bm, ln, n, W, H, disp, g, b); if (rc) return rc; }
    if (!direct) launch_copy16(disp, out, W, H, n, b);
    HIPC(hipGetLastError());
    return RTDM_OK;
}

// Fill + prefilter + SAD search of a chunk (VALU bound); *any = false: the whole frame is FILTERED, nothing follows.
static int chunk_front(rtdm_bm* bm, const Lane& ln, int n, Plane8 L, Plane8 R, int W, int H, Plane16W disp, hipStream_t s,
                       BMGeom* gout, bool* anyout)
{
    const rtdm_bm_params& p = bm->p;
    BMGeom& g = *gout;
    const bool any = make_geom(bm, W, H, &g);
    *anyout = any;
    if (!any) { launch_fill16(disp, 0, W, 0, H, n, g.filtered, s); return RTDM_OK; }
    // the search kernels write columns [cx0, cx1) of the valid rows; everything else is FILTERED
    // (+ the speckle filter's run counts = 0); the fill rides in the prefilter's launch
    const FillJob fill{disp, g.cx0, g.cx1, g.vy0, g.vy1, g.filtered, (p.speckleRange >= 0 && p.speckleWindowSize > 0) ? ln.dRowCnt : nullptr};
    StageEvent ev;
    const bool fast = fast_search_supported(g);
    bool u16 = false;
    if (!generic_search_supported(g, &u16)) return RTDM_ERR_UNSUPPORTED;
    {
        const bool ring = fast && ring_search_supported(g);
        const int lpp = ring ? ring_lanes_per_pixel(g) : 0;
        bm->variant = ring ? (lpp == 16 ? "fast_ring16_qsad" : lpp == 8 ? "fast_ring8_qsad" : lpp == 4 ? "fast_ring4_qsad" : "fast_ring_qsad") : fast ? "fast_qsad" : (u16 ? "generic_u16" : "generic_u32");
        Plane8W Lp{ln.dLp, bm->ppitch, bm->ppitch * (size_t)H}, Rp{ln.dRp, bm->ppitch, bm->ppitch * (size_t)H};
        stage_begin(bm, RTDM_STAGE_PREFILTER, n, s, &ev);
        launch_prefilter(L, R, Lp, Rp, W, H, p.preFilterCap, n, s, &fill);
        stage_end(bm, s, &ev);
        Plane8 Lpr{ln.dLp, Lp.pitch, Lp.frame}, Rpr{ln.dRp, Rp.pitch, Rp.frame};
        stage_begin(bm, RTDM_STAGE_SEARCH, n, s, &ev);
        if (fast) {
            int lx0, lx1, rx0, rx1;
            fast_border_ranges(g, &lx0, &lx1, &rx0, &rx1);
            static const bool separate = getenv("RTDM_SEPARATE_BORDER") != nullptr;   // A/B switch
            // Batches: the border columns run as a kernel of their own on a side stream, concurrently with the tile kernel.
            // Its waves need 37-72 VGPRs and fit NEXT to the four tile waves of a SIMD, whereas inside the tile kernel's
            // grid a border workgroup takes a tile workgroup's slot for the length of its latency-bound walk
            // (search -2 %).  Single frames keep the fused launch (one kernel less).  RTDM_BORDER_ASYNC=0: always fused.
            static const bool async_border = [] { const char* e = getenv("RTDM_BORDER_ASYNC"); return !e || atoi(e) != 0; }();
            static const int side_min = env_int("RTDM_BORDER_SIDE_MIN", 16);   // (A/B: smallest batch whose border columns get the side stream)
            const bool side = async_border && border_search_supported(g) && n >= side_min;
            // (the 3.x clamp exists in the stand-alone border kernel only: the fused forms keep their register budget)
            const bool fuse = border_search_supported(g) && !separate && !side && !g.legacy;   // border workgroups inside the tile kernel's grid
            // (measured, if at all, before the side stream forks: nothing else runs beside the timed launches)
            const int strips = tune_strips(bm, ln, Lpr, Rpr, disp, g, n, s, fuse, ring);
            if (side) {
                HIPC(hipEventRecord(ln.fork, s));
                HIPC(hipStreamWaitEvent(ln.side, ln.fork, 0));
#ifndef RTDM_DEBUG_SKIP_BORDER   // (timing-only variant build: what the border columns cost the search stage; outputs are wrong)
                launch_search_border(Lpr, Rpr, disp, ln.dCost, g, n, ln.side, lx0, lx1, rx0, rx1);
#endif
                HIPC(hipEventRecord(ln.join, ln.side));
            }
            bool fused = fuse;
            if (ring) fused = launch_search_ring(Lpr, Rpr, disp, ln.dCost, g, n, s, strips, fuse);
            else launch_search_fast(Lpr, Rpr, disp, ln.dCost, g, n, s, fuse, strips);
            if (side) {
                HIPC(hipStreamWaitEvent(s, ln.join, 0));
            } else if (fused) {
            } else if (border_search_supported(g)) {
                launch_search_border(Lpr, Rpr, disp, ln.dCost, g, n, s, lx0, lx1, rx0, rx1);
            } else {
                launch_search_generic(Lpr, Rpr, disp, ln.dCost, g, n, s, lx0, lx1);
                launch_search_generic(Lpr, Rpr, disp, ln.dCost, g, n, s, rx0, rx1);
            }
        } else {
            launch_search_generic(Lpr, Rpr, disp, ln.dCost, g, n, s, g.cx0 - g.lofs, g.cx1 - g.lofs);
        }
        stage_end(bm, s, &ev);
    }
    HIPC(hipGetLastError());
    return RTDM_OK;
}

// Left-right check + speckle filter of a chunk, in place on `disp` (latency bound).
static int chunk_back(rtdm_bm* bm, const Lane& ln, int n, int W, int H, Plane16W disp, const BMGeom& g, hipStream_t s)
{
    const rtdm_bm_params& p = bm->p;
    StageEvent ev;
    const bool speckle = p.speckleRange >= 0 && p.speckleWindowSize > 0;
    const bool lr = p.disp12MaxDiff >= 0;
    int compact_rows = 0;                // > 0: k_lrcheck_vec wrote per-chunk head records and merged blocks of that many rows
    if (lr) {
        stage_begin(bm, RTDM_STAGE_LRCHECK, n, s, &ev);
        if (speckle) compact_rows = launch_lrcheck(disp, ln.dCost, g, p.disp12MaxDiff, n, s, ln.dLabel, ln.dSize, ln.dRuns, ln.dRowCnt, ln.dHead,
                                                   p.speckleRange);
        else         launch_lrcheck(disp, ln.dCost, g, p.disp12MaxDiff, n, s);
        stage_end(bm, s, &ev);
    }
    if (speckle) {
        stage_begin(bm, RTDM_STAGE_SPECKLE, n, s, &ev);
        launch_speckle(disp, ln.dLabel, ln.dSize, ln.dRuns, ln.dRowCnt, ln.dHead, W, g.Ws, H, n, g.filtered, p.speckleWindowSize,
                       p.speckleRange, lr, !lr ? 1 : compact_rows > 0 ? compact_rows : lrcheck_rows_per_block(), g.vy0, g.vy1, s, compact_rows > 0);
        stage_end(bm, s, &ev);
    }
    HIPC(hipGetLastError());
    return RTDM_OK;
}

// host memory the GPU can DMA from / to directly (hipHostMalloc, hipHostRegister)?
static bool page_locked(const void* q)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc memory: an error, not a fault
    return a.type == hipMemoryTypeHost;
}

static int check_frame(const rtdm_bm* bm, int W, int H)
{
    if (W <= 0 || H <= 0 || W > bm->maxW || H > bm->maxH) return RTDM_ERR_BAD_SIZE;
    if (bm->p.blockSize >= std::min(W, H)) return RTDM_ERR_BAD_PARAM;   // cv::StereoBM::compute's check
    return RTDM_OK;
}

int rtdm_bm_compute_device(rtdm_bm* bm, int n, const uint8_t* d_left, const uint8_t* d_right,
                           size_t pitch, size_t frame_stride, int width, int height,
                           int16_t* d_disp, size_t disp_pitch, size_t disp_frame_stride, void* hip_stream)
{
    if (!bm || !d_left || !d_right || !d_disp) return RTDM_ERR_NULL;
    if (n <= 0) return RTDM_ERR_BAD_SIZE;
    int rc = check_frame(bm, width, height);
    if (rc) return rc;
    if (pitch < (size_t)width || disp_pitch < (size_t)width * 2 || (disp_pitch & 1) || (disp_frame_stride & 1))
        return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(bm->device));
    hipStream_t s = (hipStream_t)hip_stream;          // NULL = the HIP null stream (what torch's default stream is)
    const bool split = bm->nlanes == 2 && n >= 2;
    if (!split) {
        for (int i0 = 0; i0 < n; i0 += bm->laneB) {
            const int m = std::min(bm->laneB, n - i0);
            Plane8 L{d_left + (size_t)i0 * frame_stride, pitch, frame_stride};
            Plane8 R{d_right + (size_t)i0 * frame_stride, pitch, frame_stride};
            Plane16W O{d_disp + (size_t)i0 * (disp_frame_stride / 2), disp_pitch / 2, disp_frame_stride / 2};
            rc = run_chunk(bm, bm->lane[0], m, L, R, width, height, O, s);
            if (rc) return rc;
        }
        return RTDM_OK;
    }
    // Two lanes: the batch is cut into pieces whose searches run back to back on ONE front stream (two searches side by
    // side would only share the VALUs), while the row kernels of piece k (latency bound, <= 40 VGPRs: their waves fit beside
    // the two 232-VGPR search waves of a SIMD) run on the lane's back stream under the search of piece k+1.  The pieces
    // alternate between the two workspace slices; a slice is reused once its previous piece's row kernels are done.
    // RTDM_PIECES = pieces per call.
    hipStream_t front = bm->lane[0].stream;
    HIPC(hipEventRecord(bm->evIn, s));
    HIPC(hipStreamWaitEvent(front, bm->evIn, 0));
    static const int npieces = std::max(2, env_int("RTDM_PIECES", 4));
    const int piece = std::max(1, std::min(bm->laneB, (n + npieces - 1) / npieces));
    bool used[2] = {false, false};
    int k = 0;
    for (int i0 = 0; i0 < n; i0 += piece, k ^= 1) {
        const int m = std::min(piece, n - i0);
        Lane& ln = bm->lane[k];
        if (used[k]) HIPC(hipStreamWaitEvent(front, ln.done, 0));
        Plane8 L{d_left + (size_t)i0 * frame_stride, pitch, frame_stride};
        Plane8 R{d_right + (size_t)i0 * frame_stride, pitch, frame_stride};
        Plane16W O{d_disp + (size_t)i0 * (disp_frame_stride / 2), disp_pitch / 2, disp_frame_stride / 2};
        rc = run_chunk(bm, ln, m, L, R, width, height, O, front, ln.back);
        if (rc) return rc;
        HIPC(hipEventRecord(ln.done, ln.back));
        used[k] = true;
    }
    for (int q = 0; q < 2; ++q)
        if (used[q]) HIPC(hipStreamWaitEvent(s, bm->lane[q].done, 0));
    return RTDM_OK;
}

static int compute_batch_enqueue(rtdm_bm* bm, int n, const uint8_t* left, const uint8_t* right, size_t pitch, size_t frame_stride,
                                 int width, int height, int16_t* disp, size_t disp_pitch, size_t disp_frame_stride);

int rtdm_bm_compute_batch(rtdm_bm* bm, int n, const uint8_t* left, const uint8_t* right,
                          size_t pitch, size_t frame_stride, int width, int height,
                          int16_t* disp, size_t disp_pitch, size_t disp_frame_stride)
{
    if (!bm || !left || !right || !disp) return RTDM_ERR_NULL;
    if (n <= 0) return RTDM_ERR_BAD_SIZE;
    int rc = check_frame(bm, width, height);
    if (rc) return rc;
    if (pitch < (size_t)width || disp_pitch < (size_t)width * 2) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(bm->device));
    rc = compute_batch_enqueue(bm, n, left, right, pitch, frame_stride, width, height, disp, disp_pitch, disp_frame_stride);
    if (rc) {
        // an error exit must not leave DMA in flight from / to the caller's buffers: it is free to release them once we return
        const std::string first = g_hip_err;
        (void)hipStreamSynchronize(bm->sIn); (void)hipStreamSynchronize(bm->stream); (void)hipStreamSynchronize(bm->sOut);
        (void)hipGetLastError();
        g_hip_err = first;
    }
    return rc;
}

static int compute_batch_enqueue(rtdm_bm* bm, int n, const uint8_t* left, const uint8_t* right, size_t pitch, size_t frame_stride,
                                 int width, int height, int16_t* disp, size_t disp_pitch, size_t disp_frame_stride)
{
    int rc = RTDM_OK;
    hipStream_t s = bm->stream;
    const size_t dpitch = bm->ppitch, dframe = bm->ppitch * (size_t)height;
    const size_t Ws = (size_t)((width + 7) & ~7);                      // the lane's internal plane (see run_chunk)
    const size_t opitch = Ws * 2, oframe = opitch * (size_t)height;
    // Three streams, two halves of the staging planes: while chunk k is computed, chunk k+1 comes in over PCIe and chunk
    // k-1 goes out (both directions of the bus at once).  It pays for page-locked caller memory (hipHostMalloc /
    // hipHostRegister: the copies are true DMA); pageable frames are staged by the runtime inside the copy call.
    // Measured (tools/host_batch_rate.py, 256 x 720p): page-locked 11.6 k -> 21.1 k pairs/s (78 GB/s over PCIe, both ways),
    // pageable 11.2 k -> 10.5 k: so only for page-locked callers.  RTDM_BATCH_PIPELINE = 0 never, 2 always.
    const int half = std::max(1, bm->laneB / 2);
    static const int pipe_mode = [] { const char* e = getenv("RTDM_BATCH_PIPELINE"); return e ? atoi(e) : 1; }();
    const bool two = bm->laneB >= 2 && (pipe_mode == 2 || (pipe_mode == 1 && page_locked(left) && page_locked(right) && page_locked(disp)));
    const int chunk = two ? half : bm->laneB;
    int k = 0;
    for (int i0 = 0; i0 < n; i0 += chunk, ++k) {
        const int m = std::min(chunk, n - i0), b = two ? (k & 1) : 0;
        const size_t fo = (size_t)b * (size_t)half;                    // first staging frame of this half
        hipStream_t si = two ? bm->sIn : s, so = two ? bm->sOut : s;
        if (two && k >= 2) HIPC(hipStreamWaitEvent(si, bm->evComp[b], 0));      // the half's previous chunk has been consumed
        uint8_t *dl = bm->dInL + fo * dframe, *dr = bm->dInR + fo * dframe;
        if (pitch == dpitch && frame_stride == dframe) {               // the caller's frames have the staging layout: one copy per image
            // (up to the last pixel of the last row: a view that starts at x > 0 of its parent plane ends before the row's pitch does)
            const size_t nbytes = (size_t)m * dframe - (dpitch - (size_t)width);
            HIPC(hipMemcpyAsync(dl, left + (size_t)i0 * frame_stride, nbytes, hipMemcpyHostToDevice, si));
            HIPC(hipMemcpyAsync(dr, right + (size_t)i0 * frame_stride, nbytes, hipMemcpyHostToDevice, si));
        } else {
            for (int i = 0; i < m; ++i) {
                HIPC(hipMemcpy2DAsync(dl + i * dframe, dpitch, left + (size_t)(i0 + i) * frame_stride, pitch, width, height, hipMemcpyHostToDevice, si));
                HIPC(hipMemcpy2DAsync(dr + i * dframe, dpitch, right + (size_t)(i0 + i) * frame_stride, pitch, width, height, hipMemcpyHostToDevice, si));
            }
        }
        if (two) {
            HIPC(hipEventRecord(bm->evH2D[b], si));
            HIPC(hipStreamWaitEvent(s, bm->evH2D[b], 0));
            if (k >= 2) HIPC(hipStreamWaitEvent(s, bm->evD2H[b], 0));   // ... and its previous result has left
        }
        Plane8 L{dl, dpitch, dframe}, R{dr, dpitch, dframe};
        int16_t* dout = bm->dOut + fo * Ws * (size_t)height;
        Plane16W O{dout, Ws, Ws * (size_t)height};
        rc = run_chunk(bm, bm->lane[0], m, L, R, width, height, O, s);
        if (rc) return rc;
        if (two) { HIPC(hipEventRecord(bm->evComp[b], s)); HIPC(hipStreamWaitEvent(so, bm->evComp[b], 0)); }
        // one linear copy only when the internal rows carry no padding: with width < Ws it would write the pad columns
        // [width, Ws) of the caller's rows, which are not part of the view
        if ((size_t)width == Ws && disp_pitch == opitch && disp_frame_stride == oframe) {
            HIPC(hipMemcpyAsync((uint8_t*)disp + (size_t)i0 * disp_frame_stride, dout, (size_t)m * oframe, hipMemcpyDeviceToHost, so));
        } else {
            for (int i = 0; i < m; ++i)
                HIPC(hipMemcpy2DAsync((uint8_t*)disp + (size_t)(i0 + i) * disp_frame_stride, disp_pitch,
                                      (uint8_t*)dout + i * oframe, opitch, (size_t)width * 2, height, hipMemcpyDeviceToHost, so));
        }
        if (two) HIPC(hipEventRecord(bm->evD2H[b], so));
        else HIPC(hipStreamSynchronize(s));                            // staging buffers are reused by the next chunk
    }
    if (two) { HIPC(hipStreamSynchronize(bm->sOut)); HIPC(hipStreamSynchronize(s)); HIPC(hipStreamSynchronize(bm->sIn)); }
    return RTDM_OK;
}

int rtdm_bm_compute(rtdm_bm* bm, const uint8_t* left, size_t left_pitch, const uint8_t* right,
                    size_t right_pitch, int width, int height, int16_t* disp, size_t disp_pitch)
{
    if (!bm || !left || !right || !disp) return RTDM_ERR_NULL;
    int rc = check_frame(bm, width, height);
    if (rc) return rc;
    if (left_pitch < (size_t)width || right_pitch < (size_t)width || disp_pitch < (size_t)width * 2)
        return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(bm->device));
    hipStream_t s = bm->stream;
    DrainOnError drain{s};
    const size_t dpitch = bm->ppitch, dframe = bm->ppitch * (size_t)height;
    // The caller's Mats are pageable ROI views (estimator.cpp:33,36): the rows are gathered into the page-locked staging
    // area on the host and go over in linear async copies -- in two BANDS of rows per direction, so that the host gathers
    // band 1 while band 0 is on the bus, and scatters band 0 of the result while band 1 arrives.  720p pair, host to host:
    // 0.363 ms with one copy per plane and row-by-row gathers, 0.347 with one band, 0.325 with two, 0.375 with four (every
    // further async copy costs more in the runtime than its overlap hides; RTDM_HOST_BANDS=1..4).  Rows that are contiguous
    // in the caller's plane move as one memcpy.
    const size_t Wsd = (size_t)((width + 7) & ~7);
    if (page_locked(left) && page_locked(right) && page_locked(disp)) {
        // the caller's planes are page-locked: DMA straight from and to them, no gathers on the host
        HIPC(hipMemcpy2DAsync(bm->dInL, dpitch, left, left_pitch, (size_t)width, height, hipMemcpyHostToDevice, s));
        HIPC(hipMemcpy2DAsync(bm->dInR, dpitch, right, right_pitch, (size_t)width, height, hipMemcpyHostToDevice, s));
        Plane8 Ld{bm->dInL, dpitch, dframe}, Rd{bm->dInR, dpitch, dframe};
        Plane16W Od{bm->dOut, Wsd, Wsd * (size_t)height};
        rc = run_chunk(bm, bm->lane[0], 1, Ld, Rd, width, height, Od, s);
        if (rc) return rc;
        HIPC(hipMemcpy2DAsync(disp, disp_pitch, bm->dOut, Wsd * sizeof(int16_t), (size_t)width * sizeof(int16_t), height, hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        drain.armed = false;
        return RTDM_OK;
    }
    uint8_t* hL = bm->hStage;
    uint8_t* hR = hL + dframe;
    int16_t* hD = (int16_t*)(bm->hStage + 2 * bm->ppitch * (size_t)bm->maxH);
    static const int bands = [] { const char* e = getenv("RTDM_HOST_BANDS"); return e ? std::max(1, std::min(4, atoi(e))) : 2; }();
    const int nb = height >= 256 ? bands : 1, bh = (height + nb - 1) / nb;
    const auto gather = [&](uint8_t* dst, const uint8_t* src, size_t spitch, int y0, int y1) {
        if (spitch == dpitch) { memcpy(dst + (size_t)y0 * dpitch, src + (size_t)y0 * spitch, (size_t)(y1 - y0 - 1) * dpitch + (size_t)width); return; }
        for (int y = y0; y < y1; ++y) memcpy(dst + (size_t)y * dpitch, src + (size_t)y * spitch, (size_t)width);
    };
    for (int b = 0; b < nb; ++b) {
        const int y0 = b * bh, y1 = std::min(height, y0 + bh);
        if (y0 >= y1) break;
        gather(hL, left, left_pitch, y0, y1);
        HIPC(hipMemcpyAsync(bm->dInL + (size_t)y0 * dpitch, hL + (size_t)y0 * dpitch, (size_t)(y1 - y0) * dpitch, hipMemcpyHostToDevice, s));
        gather(hR, right, right_pitch, y0, y1);
        HIPC(hipMemcpyAsync(bm->dInR + (size_t)y0 * dpitch, hR + (size_t)y0 * dpitch, (size_t)(y1 - y0) * dpitch, hipMemcpyHostToDevice, s));
    }
    Plane8 L{bm->dInL, dpitch, dframe}, R{bm->dInR, dpitch, dframe};
    const size_t Ws = (size_t)((width + 7) & ~7);                      // the lane's internal plane (see run_chunk)
    Plane16W O{bm->dOut, Ws, Ws * (size_t)height};
    rc = run_chunk(bm, bm->lane[0], 1, L, R, width, height, O, s);
    if (rc) return rc;
    int nbo = 0;
    for (int b = 0; b < nb; ++b, ++nbo) {
        const int y0 = b * bh, y1 = std::min(height, y0 + bh);
        if (y0 >= y1) break;
        HIPC(hipMemcpyAsync(hD + (size_t)y0 * Ws, bm->dOut + (size_t)y0 * Ws, (size_t)(y1 - y0) * Ws * sizeof(int16_t), hipMemcpyDeviceToHost, s));
        HIPC(hipEventRecord(bm->evBand[b], s));
    }
    for (int b = 0; b < nbo; ++b) {
        const int y0 = b * bh, y1 = std::min(height, y0 + bh);
        HIPC(hipEventSynchronize(bm->evBand[b]));
        if ((size_t)width == Ws && disp_pitch == Ws * sizeof(int16_t)) {   // no pad columns between the rows: one copy per band
            memcpy((uint8_t*)disp + (size_t)y0 * disp_pitch, hD + (size_t)y0 * Ws, (size_t)(y1 - y0 - 1) * disp_pitch + (size_t)width * sizeof(int16_t));
        } else {
            for (int y = y0; y < y1; ++y)
                memcpy((uint8_t*)disp + (size_t)y * disp_pitch, hD + (size_t)y * Ws, (size_t)width * sizeof(int16_t));
        }
    }
    drain.armed = false;
    return RTDM_OK;
}

int rtdm_bm_synchronize(rtdm_bm* bm)
{
    if (!bm) return RTDM_ERR_NULL;
    HIPC(hipSetDevice(bm->device));
    HIPC(hipStreamSynchronize(bm->stream));
    return RTDM_OK;
}

int rtdm_bm_set_profiling(rtdm_bm* bm, int enabled)
{
    if (!bm) return RTDM_ERR_NULL;
    bm->profiling = enabled != 0;
    return RTDM_OK;
}

static int drain_events(rtdm_bm* bm)
{
    for (auto& ev : bm->pending) {
        HIPC(hipEventSynchronize(ev.b));
        float ms = 0.f;
        HIPC(hipEventElapsedTime(&ms, ev.a, ev.b));
        bm->stage_ms[ev.stage] += ms;
        bm->stage_launches[ev.stage] += 1;
        bm->stage_frames[ev.stage] += ev.frames;
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b);
    }
    bm->pending.clear();
    return RTDM_OK;
}

int rtdm_bm_get_stage_time(rtdm_bm* bm, int stage, double* total_ms, long* launches, long* frames)
{
    if (!bm) return RTDM_ERR_NULL;
    if (stage < 0 || stage >= RTDM_NUM_STAGES) return RTDM_ERR_BAD_PARAM;
    HIPC(hipSetDevice(bm->device));
    int rc = drain_events(bm);
    if (rc) return rc;
    if (total_ms) *total_ms = bm->stage_ms[stage];
    if (launches) *launches = bm->stage_launches[stage];
    if (frames) *frames = bm->stage_frames[stage];
    return RTDM_OK;
}

int rtdm_bm_reset_stage_times(rtdm_bm* bm)
{
    if (!bm) return RTDM_ERR_NULL;
    HIPC(hipSetDevice(bm->device));
    int rc = drain_events(bm);
    if (rc) return rc;
    for (int i = 0; i < RTDM_NUM_STAGES; ++i) { bm->stage_ms[i] = 0; bm->stage_launches[i] = 0; bm->stage_frames[i] = 0; }
    return RTDM_OK;
}

const char* rtdm_bm_search_variant(const rtdm_bm* bm) { return bm ? bm->variant.c_str() : ""; }
int rtdm_bm_get_tuner_stats(const rtdm_bm* bm, long* shapes_measured, long* timing_launches)
{
    if (!bm) return RTDM_ERR_NULL;
    if (shapes_measured) *shapes_measured = bm->tune_shapes;
    if (timing_launches) *timing_launches = bm->tune_launches;
    return RTDM_OK;
}
void rtdm_debug_search_kernel(int mode) { ring_set_mode(mode); }

// ---- VideoFilterDevice ---------------------------------------------------------------------
int rtdm_morph_create(int width, int height, int max_batch, int device, rtdm_morph** out)
{
    if (!out) return RTDM_ERR_NULL;
    *out = nullptr;
    if (width <= 0 || height <= 0 || max_batch <= 0) return RTDM_ERR_BAD_SIZE;
    int rc = use_device(device);
    if (rc) return rc;
    rtdm_morph* mf = new (std::nothrow) rtdm_morph();
    if (!mf) return RTDM_ERR_NOMEM;
    mf->W = width; mf->H = height; mf->maxB = max_batch; mf->device = device;
    const size_t px = (size_t)width * height, all = px * max_batch;
    hipError_t e = hipStreamCreateWithFlags(&mf->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void**)&mf->hIn, px, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&mf->hOut, px, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&mf->dIn, all);
    if (e == hipSuccess) e = hipMalloc((void**)&mf->dOut, all);
    if (e == hipSuccess) e = hipMalloc((void**)&mf->dT0, all);
    if (e == hipSuccess) e = hipMalloc((void**)&mf->dT1, all);
    if (e != hipSuccess) {
        g_hip_err = std::string("rtdm_morph_create: ") + hipGetErrorString(e);
        rtdm_morph_destroy(mf);
        return e == hipErrorOutOfMemory ? RTDM_ERR_NOMEM : RTDM_ERR_HIP;
    }
    *out = mf;
    return RTDM_OK;
}

void rtdm_morph_destroy(rtdm_morph* mf)
{
    if (!mf) return;
    (void)hipSetDevice(mf->device);
    if (mf->stream) (void)hipStreamSynchronize(mf->stream);
    if (mf->hIn) (void)hipHostFree(mf->hIn);
    if (mf->hOut) (void)hipHostFree(mf->hOut);
    void* bufs[] = {mf->dIn, mf->dOut, mf->dT0, mf->dT1};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (mf->stream) (void)hipStreamDestroy(mf->stream);
    delete mf;
}

uint8_t* rtdm_morph_in_buffer(rtdm_morph* mf) { return mf ? mf->hIn : nullptr; }
uint8_t* rtdm_morph_out_buffer(rtdm_morph* mf) { return mf ? mf->hOut : nullptr; }

int rtdm_morph_run_device(rtdm_morph* mf, int n, const uint8_t* d_in, size_t in_pitch, size_t in_frame_stride,
                          uint8_t* d_out, size_t out_pitch, size_t out_frame_stride, int width, int height,
                          void* hip_stream)
{
    if (!mf || !d_in || !d_out) return RTDM_ERR_NULL;
    if (n <= 0 || width <= 0 || height <= 0 || (size_t)width * height > (size_t)mf->W * mf->H) return RTDM_ERR_BAD_SIZE;
    if (in_pitch < (size_t)width || out_pitch < (size_t)width) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(mf->device));
    hipStream_t s = (hipStream_t)hip_stream;          // NULL = the HIP null stream
    for (int i0 = 0; i0 < n; i0 += mf->maxB) {
        const int m = std::min(mf->maxB, n - i0);
        Plane8 in{d_in + (size_t)i0 * in_frame_stride, in_pitch, in_frame_stride};
        Plane8W out{d_out + (size_t)i0 * out_frame_stride, out_pitch, out_frame_stride};
        launch_morph_open_close(in, out, mf->dT0, mf->dT1, width, height, m, s);
    }
    HIPC(hipGetLastError());
    return RTDM_OK;
}

int rtdm_morph_run(rtdm_morph* mf, const uint8_t* in, size_t in_pitch, uint8_t* out, size_t out_pitch,
                   int width, int height)
{
    if (!mf || !in || !out) return RTDM_ERR_NULL;
    if (width <= 0 || height <= 0 || (size_t)width * height > (size_t)mf->W * mf->H) return RTDM_ERR_BAD_SIZE;
    if (in_pitch < (size_t)width || out_pitch < (size_t)width) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(mf->device));
    hipStream_t s = mf->stream;
    HIPC(hipMemcpy2DAsync(mf->dIn, width, in, in_pitch, width, height, hipMemcpyHostToDevice, s));
    int rc = rtdm_morph_run_device(mf, 1, mf->dIn, width, (size_t)width * height, mf->dOut, width,
                                   (size_t)width * height, width, height, s);
    if (rc) return rc;
    HIPC(hipMemcpy2DAsync(out, out_pitch, mf->dOut, width, width, height, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    return RTDM_OK;
}

// ---- the step after the matcher ---------------------------------------------------------------
static int check_regions(const rtdm_region* regions, int n, int W, int H, int* flat, int* maxh)
{
    if (n < 0 || n > RTDM_MAX_REGIONS || (n > 0 && !regions)) return RTDM_ERR_BAD_SIZE;
    *maxh = 1;
    for (int i = 0; i < n; ++i) {
        const rtdm_region& r = regions[i];
        if (r.x < 0 || r.y < 0 || r.width < 0 || r.height < 0 || r.x + r.width > W || r.y + r.height > H) return RTDM_ERR_BAD_SIZE;
        flat[4 * i] = r.x; flat[4 * i + 1] = r.y; flat[4 * i + 2] = r.width; flat[4 * i + 3] = r.height;
        *maxh = std::max(*maxh, r.height);
    }
    return RTDM_OK;
}

int rtdm_depth_stats_device(int device, const int16_t* d_disp, size_t disp_pitch, int width, int height, const double* Q,
                            const uint8_t* d_mask, size_t mask_pitch, const rtdm_region* regions, int nregions,
                            double calibration_unit, double* mean_cm, int* counts, void* hip_stream)
{
    if (!d_disp || !Q || !d_mask || !mean_cm || !counts) return RTDM_ERR_NULL;
    if (width <= 0 || height <= 0 || disp_pitch < (size_t)width * 2 || (disp_pitch & 1) || mask_pitch < (size_t)width) return RTDM_ERR_BAD_SIZE;
    int flat[4 * RTDM_MAX_REGIONS], maxh = 1;
    int rc = check_regions(regions, nregions, width, height, flat, &maxh);
    if (rc) return rc;
    rc = use_device(device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    void* scratch = nullptr;
    HIPC(hipMalloc(&scratch, depth_scratch_bytes(std::max(nregions, 1), maxh)));
    DepthQ q; std::copy(Q, Q + 16, q.q);
    launch_depth_stats(d_disp, disp_pitch / 2, width, height, q, d_mask, mask_pitch, flat, nregions, maxh, calibration_unit,
                       scratch, mean_cm, counts, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    HIPC(e);
    return RTDM_OK;
}

int rtdm_bm_compute_depth(rtdm_bm* bm, const uint8_t* left, size_t left_pitch, const uint8_t* right, size_t right_pitch,
                          int width, int height, const double* Q, const uint8_t* mask, size_t mask_pitch,
                          const rtdm_region* regions, int nregions, double calibration_unit,
                          double* mean_cm, int* counts, int16_t* disp, size_t disp_pitch)
{
    if (!bm || !left || !right || !Q || !mask || !mean_cm || !counts) return RTDM_ERR_NULL;
    int rc = check_frame(bm, width, height);
    if (rc) return rc;
    if (left_pitch < (size_t)width || right_pitch < (size_t)width || mask_pitch < (size_t)width) return RTDM_ERR_BAD_SIZE;
    if (disp && disp_pitch < (size_t)width * 2) return RTDM_ERR_BAD_SIZE;
    int flat[4 * RTDM_MAX_REGIONS], maxh = 1;
    rc = check_regions(regions, nregions, width, height, flat, &maxh);
    if (rc) return rc;
    HIPC(hipSetDevice(bm->device));
    hipStream_t s = bm->stream;
    const size_t dpitch = bm->ppitch, dframe = bm->ppitch * (size_t)height;
    uint8_t* hL = bm->hStage;
    uint8_t* hR = hL + dframe;
    int16_t* hD = (int16_t*)(bm->hStage + 2 * bm->ppitch * (size_t)bm->maxH);
    uint8_t* hM = (uint8_t*)(hD + (size_t)((bm->maxW + 7) & ~7) * bm->maxH);
    for (int y = 0; y < height; ++y) {
        memcpy(hL + (size_t)y * dpitch, left + (size_t)y * left_pitch, (size_t)width);
        memcpy(hR + (size_t)y * dpitch, right + (size_t)y * right_pitch, (size_t)width);
        memcpy(hM + (size_t)y * width, mask + (size_t)y * mask_pitch, (size_t)width);
    }
    HIPC(hipMemcpyAsync(bm->dInL, hL, dframe, hipMemcpyHostToDevice, s));
    HIPC(hipMemcpyAsync(bm->dInR, hR, dframe, hipMemcpyHostToDevice, s));
    HIPC(hipMemcpyAsync(bm->dMask, hM, (size_t)width * height, hipMemcpyHostToDevice, s));
    Plane8 L{bm->dInL, dpitch, dframe}, R{bm->dInR, dpitch, dframe};
    const size_t Ws = (size_t)((width + 7) & ~7);                      // the lane's internal plane (see run_chunk)
    Plane16W O{bm->dOut, Ws, Ws * (size_t)height};
    rc = run_chunk(bm, bm->lane[0], 1, L, R, width, height, O, s);
    if (rc) return rc;
    DepthQ q; std::copy(Q, Q + 16, q.q);
    launch_depth_stats(bm->dOut, Ws, width, height, q, bm->dMask, (size_t)width, flat, nregions, bm->maxH,
                       calibration_unit, bm->dDepth, mean_cm, counts, s);
    if (disp) HIPC(hipMemcpyAsync(hD, bm->dOut, Ws * height * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s));
    if (disp)
        for (int y = 0; y < height; ++y)
            memcpy((uint8_t*)disp + (size_t)y * disp_pitch, hD + (size_t)y * Ws, (size_t)width * sizeof(int16_t));
    return RTDM_OK;
}

// ---- SWSemiGlobalMatcher counterpart -------------------------------------------------------
struct rtdm_sgm {
    rtdm_sgm_params p;
    int maxW, maxH, maxB, device;
    hipStream_t stream;
    uint8_t *dInL, *dInR;
    int16_t* dOut;
    SGMBuffers b;
    int cost_limit;                // > 0 (windows > 17 at P2 = 2400): a block cost above it would wrap the library's 16-bit path costs
    int32_t* hOvf;                 // page-locked copy of b.ovf
    uint32_t sweep_epoch;          // launches of k_sgm_sweep (tags of its edge ring)
    int sweep_cap[36];             // workgroups the device holds at once, per instantiation (0 = not asked yet)
    bool sweep_reported;           // a give-up of the sweep has been returned to the caller
};

// The row-synchronous sweep waits on its neighbours with a bound; a pass that gave up has produced garbage and has said so in a
// page-locked flag.  The call that finds the flag returns an error ONCE; from then on the handle runs one pass per direction.
static int sgm_sweep_check(rtdm_sgm* sg)
{
    if (!sg->b.abortf || !*sg->b.abortf || sg->sweep_reported) return RTDM_OK;
    sg->sweep_reported = true;
    g_hip_err = "StereoSGBM: a row-synchronous sweep gave up waiting for a neighbouring strip; the output of that call is invalid "
                "(this handle runs one pass per direction from now on)";
    return RTDM_ERR_HIP;
}

// windows whose block cost + P2 can pass 32767: the frame is refused if it does (what is not restated is the wrap-around)
static int sgm_overflow_check(rtdm_sgm* sg, hipStream_t s)
{
    if (!sg->cost_limit) return RTDM_OK;
    HIPC(hipMemcpyAsync(sg->hOvf, sg->b.ovf, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    if (!*sg->hOvf) return RTDM_OK;
    HIPC(hipMemsetAsync(sg->b.ovf, 0, sizeof(int32_t), s));
    g_hip_err = "StereoSGBM: a block cost + P2 exceeds 32767 in this frame (the library's 16-bit costs would wrap)";
    return RTDM_ERR_UNSUPPORTED;
}

void rtdm_sgm_default_params(rtdm_sgm_params* p, int numDisparities, int blockSize)
{
    if (!p) return;
    p->blockSize = blockSize; p->minDisparity = 0; p->numDisparities = numDisparities; p->P1 = 600; p->P2 = 2400;
    p->uniquenessRatio = 10; p->speckleWindowSize = 100; p->speckleRange = 32; p->disp12MaxDiff = 1; p->paths = 8;
}

int rtdm_sgm_create(const rtdm_sgm_params* params, int max_width, int max_height, int max_batch, int device, rtdm_sgm** out)
{
    if (!params || !out) return RTDM_ERR_NULL;
    *out = nullptr;
    rtdm_sgm_params p = *params;
    if (p.numDisparities <= 0 || p.numDisparities % 16 != 0 || p.blockSize < 1) return RTDM_ERR_BAD_PARAM;
    if (p.uniquenessRatio > 100) return RTDM_ERR_BAD_PARAM;
    if (p.paths != 5 && p.paths != 8) return RTDM_ERR_BAD_PARAM;
    // cv::StereoSGBM never checks the parity of blockSize: its window is SADWindowSize / 2 either side, an even size runs as
    // the next odd one (sgbm-sw.cpp:15 hands the caller's blockSize straight through)
    p.blockSize = p.blockSize / 2 * 2 + 1;
    // what cv::StereoSGBM does with out-of-range knobs (oracle/sgm_oracle.c R6, R9, R12): it coerces them
    if (p.P1 <= 0) p.P1 = 2;
    p.P2 = std::max(p.P2 > 0 ? p.P2 : 5, p.P1 + 1);
    if (p.uniquenessRatio < 0) p.uniquenessRatio = 10;
    if (p.disp12MaxDiff <= 0) p.disp12MaxDiff = 1;         // the library's left-right check cannot be switched off
    if (max_width <= 0 || max_height <= 0 || max_batch <= 0) return RTDM_ERR_BAD_SIZE;
    if (p.numDisparities > 256 || max_width > 4096) return RTDM_ERR_UNSUPPORTED;
    // 16-bit costs: a path cost is at most block cost + P2 (pixel cost <= 30 + 63); above 32767 the library's short
    // arithmetic wraps, which is not restated: windows that CAN get there (> 17 at P2 = 2400) run with a check of the block
    // costs and refuse the frame that does (RTDM_ERR_UNSUPPORTED from the compute call; it takes nearly every pixel of a
    // window at the maximum pixel cost)
    if (p.blockSize > 255 || p.P2 > 32000) return RTDM_ERR_UNSUPPORTED;
    const int cost_limit = 93L * p.blockSize * p.blockSize + p.P2 > 32767 ? 32767 - p.P2 : 0;
    int rc = use_device(device);
    if (rc) return rc;
    rtdm_sgm* sg = new (std::nothrow) rtdm_sgm();
    if (!sg) return RTDM_ERR_NOMEM;
    sg->p = p; sg->maxW = max_width; sg->maxH = max_height; sg->maxB = max_batch; sg->device = device;
    sg->cost_limit = cost_limit;
    const size_t px = (size_t)max_width * max_height * max_batch;
    const size_t vol = px * p.numDisparities;
    hipError_t e = hipStreamCreateWithFlags(&sg->stream, hipStreamNonBlocking);
    void** ptrs[] = {(void**)&sg->dInL, (void**)&sg->dInR, (void**)&sg->dOut, (void**)&sg->b.gl, (void**)&sg->b.gr,
                     (void**)&sg->b.pix, (void**)&sg->b.C, (void**)&sg->b.S, (void**)&sg->b.label, (void**)&sg->b.size,
                     (void**)&sg->b.runs, (void**)&sg->b.rowcnt, (void**)&sg->b.headmap};
    const size_t sizes[] = {px, px, px * 2, px * 8, px * 8, vol, vol * 2, vol * 2, px * 4, px * 4, px * 4,
                            (size_t)max_batch * max_height * 4, px * 2};
    for (int i = 0; i < 13 && e == hipSuccess; ++i) e = hipMalloc(ptrs[i], sizes[i]);
    if (e == hipSuccess) e = hipMalloc((void**)&sg->b.ovf, sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(sg->b.ovf, 0, sizeof(int32_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&sg->hOvf, sizeof(int32_t), hipHostMallocDefault);
    sg->b.ring_words = sgm_ring_words(max_width, p.numDisparities, max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&sg->b.ring, sg->b.ring_words * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(sg->b.ring, 0, sg->b.ring_words * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void**)&sg->b.abortf, sizeof(int32_t), hipHostMallocMapped);
    if (e == hipSuccess && hipMalloc((void**)&sg->b.S2, vol * 2) != hipSuccess) { (void)hipGetLastError(); sg->b.S2 = nullptr; }   // (optional: without it the horizontal passes run one after the other)
    if (e == hipSuccess) *sg->b.abortf = 0;
    sg->b.epoch = &sg->sweep_epoch; sg->b.sweep_cap = sg->sweep_cap;
    if (e == hipSuccess) e = hipEventCreateWithFlags((hipEvent_t*)&sg->b.ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags((hipEvent_t*)&sg->b.ev_out, hipEventDisableTiming);
    if (e != hipSuccess) {
        g_hip_err = std::string("rtdm_sgm_create: ") + hipGetErrorString(e);
        rtdm_sgm_destroy(sg);
        return e == hipErrorOutOfMemory ? RTDM_ERR_NOMEM : RTDM_ERR_HIP;
    }
    *out = sg;
    return RTDM_OK;
}

void rtdm_sgm_destroy(rtdm_sgm* sg)
{
    if (!sg) return;
    (void)hipSetDevice(sg->device);
    if (sg->stream) (void)hipStreamSynchronize(sg->stream);
    void* bufs[] = {sg->dInL, sg->dInR, sg->dOut, sg->b.gl, sg->b.gr, sg->b.pix, sg->b.C, sg->b.S, sg->b.label, sg->b.size,
                    sg->b.runs, sg->b.rowcnt, sg->b.headmap};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (sg->b.ovf) (void)hipFree(sg->b.ovf);
    if (sg->b.ring) (void)hipFree(sg->b.ring);
    if (sg->b.S2) (void)hipFree(sg->b.S2);
    if (sg->b.abortf) (void)hipHostFree(sg->b.abortf);
    if (sg->b.ev_in) (void)hipEventDestroy((hipEvent_t)sg->b.ev_in);
    if (sg->b.ev_out) (void)hipEventDestroy((hipEvent_t)sg->b.ev_out);
    if (sg->hOvf) (void)hipHostFree(sg->hOvf);
    if (sg->stream) (void)hipStreamDestroy(sg->stream);
    delete sg;
}

static int sgm_chunk(rtdm_sgm* sg, int n, Plane8 L, Plane8 R, int W, int H, Plane16W disp, hipStream_t s)
{
    const rtdm_sgm_params& p = sg->p;
    SGMGeom g;
    g.W = W; g.H = H; g.D = p.numDisparities; g.minD = p.minDisparity;
    g.x0 = std::max(g.minD + g.D, 0);
    g.W1 = (W + std::min(g.minD, 0)) - g.x0;
    if (g.W1 <= 0) { launch_fill16(disp, 0, W, 0, H, n, (g.minD - 1) * 16, s); return RTDM_OK; }
    launch_sgm(L, R, disp, g, sg->b, p.blockSize, p.P1, p.P2, p.uniquenessRatio, p.disp12MaxDiff, p.speckleWindowSize,
               p.speckleRange, p.paths, n, s, sg->cost_limit);
    HIPC(hipGetLastError());
    return RTDM_OK;
}

int rtdm_sgm_compute_device(rtdm_sgm* sg, int n, const uint8_t* d_left, const uint8_t* d_right, size_t pitch,
                            size_t frame_stride, int width, int height, int16_t* d_disp, size_t disp_pitch,
                            size_t disp_frame_stride, void* hip_stream)
{
    if (!sg || !d_left || !d_right || !d_disp) return RTDM_ERR_NULL;
    if (n <= 0 || width <= 0 || height <= 0 || width > sg->maxW || height > sg->maxH) return RTDM_ERR_BAD_SIZE;
    if (pitch < (size_t)width || disp_pitch < (size_t)width * 2 || (disp_pitch & 1) || (disp_frame_stride & 1)) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(sg->device));
    hipStream_t s = (hipStream_t)hip_stream;          // NULL = the HIP null stream
    for (int i0 = 0; i0 < n; i0 += sg->maxB) {
        const int m = std::min(sg->maxB, n - i0);
        Plane8 L{d_left + (size_t)i0 * frame_stride, pitch, frame_stride}, R{d_right + (size_t)i0 * frame_stride, pitch, frame_stride};
        Plane16W O{d_disp + (size_t)i0 * (disp_frame_stride / 2), disp_pitch / 2, disp_frame_stride / 2};
        int rc = sgm_chunk(sg, m, L, R, width, height, O, s);
        if (rc) return rc;
    }
    int rc = sgm_overflow_check(sg, s);                // (windows > 17 only: this call then synchronises the stream)
    return rc ? rc : sgm_sweep_check(sg);              // (asynchronous call: a give-up of this call's sweep shows in the next call)
}

int rtdm_sgm_compute(rtdm_sgm* sg, const uint8_t* left, size_t left_pitch, const uint8_t* right, size_t right_pitch,
                     int width, int height, int16_t* disp, size_t disp_pitch)
{
    if (!sg || !left || !right || !disp) return RTDM_ERR_NULL;
    if (width <= 0 || height <= 0 || width > sg->maxW || height > sg->maxH) return RTDM_ERR_BAD_SIZE;
    if (left_pitch < (size_t)width || right_pitch < (size_t)width || disp_pitch < (size_t)width * 2) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(sg->device));
    hipStream_t s = sg->stream;
    HIPC(hipMemcpy2DAsync(sg->dInL, width, left, left_pitch, width, height, hipMemcpyHostToDevice, s));
    HIPC(hipMemcpy2DAsync(sg->dInR, width, right, right_pitch, width, height, hipMemcpyHostToDevice, s));
    Plane8 L{sg->dInL, (size_t)width, (size_t)width * height}, R{sg->dInR, (size_t)width, (size_t)width * height};
    Plane16W O{sg->dOut, (size_t)width, (size_t)width * height};
    int rc = sgm_chunk(sg, 1, L, R, width, height, O, s);
    if (rc) return rc;
    HIPC(hipMemcpy2DAsync(disp, disp_pitch, sg->dOut, (size_t)width * 2, (size_t)width * 2, height, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    rc = sgm_overflow_check(sg, s);
    return rc ? rc : sgm_sweep_check(sg);
}

int rtdm_sgm_get_pass_stats(const rtdm_sgm* sg, long* sweeps, int* gave_up)
{
    if (!sg) return RTDM_ERR_NULL;
    if (sweeps) *sweeps = (long)sg->sweep_epoch;
    if (gave_up) *gave_up = sg->b.abortf && *sg->b.abortf ? 1 : 0;
    return RTDM_OK;
}

// ---- rectification in front of the matcher (estimator.cpp:29-39) ----------------------------------------------
struct rtdm_rectify {
    int W, H, rx, ry, rw, rh, maxB, device;
    int16_t* dMap1[2];            // roi part of the maps: rh x rw x 2
    uint16_t* dMap2[2];           // rh x rw
    uint8_t* dRgb[2];             // staging for host frames: H x W x 3 (+ padding)
    uint8_t* dOut;                // staging for host outputs: rh x rw x 3
    uint8_t* dGray[2];            // rectified gray pair for the chained matcher call: rh x pitch
    size_t gpitch;
    uint8_t* hStage;              // pinned: 2 RGB frames in, rh x rw x 3 out
    hipStream_t stream;
};

void rtdm_rectify_destroy(rtdm_rectify* rc)
{
    if (!rc) return;
    (void)hipSetDevice(rc->device);
    if (rc->stream) (void)hipStreamSynchronize(rc->stream);
    void* bufs[] = {rc->dMap1[0], rc->dMap1[1], rc->dMap2[0], rc->dMap2[1], rc->dRgb[0], rc->dRgb[1], rc->dOut, rc->dGray[0], rc->dGray[1]};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (rc->hStage) (void)hipHostFree(rc->hStage);
    if (rc->stream) (void)hipStreamDestroy(rc->stream);
    delete rc;
}

int rtdm_rectify_create(const int16_t* map1_left, const uint16_t* map2_left, const int16_t* map1_right,
                        const uint16_t* map2_right, int width, int height, int roi_x, int roi_y, int roi_width,
                        int roi_height, int max_batch, int device, rtdm_rectify** out)
{
    if (!map1_left || !map2_left || !map1_right || !map2_right || !out) return RTDM_ERR_NULL;
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 32767 || height > 32767 || max_batch <= 0) return RTDM_ERR_BAD_SIZE;
    if (roi_x < 0 || roi_y < 0 || roi_width <= 0 || roi_height <= 0 || roi_x + roi_width > width || roi_y + roi_height > height)
        return RTDM_ERR_BAD_SIZE;
    int st = use_device(device);
    if (st) return st;
    rtdm_rectify* rc = new (std::nothrow) rtdm_rectify();
    if (!rc) return RTDM_ERR_NOMEM;
    rc->W = width; rc->H = height; rc->rx = roi_x; rc->ry = roi_y; rc->rw = roi_width; rc->rh = roi_height;
    rc->maxB = max_batch; rc->device = device;
    rc->gpitch = ((size_t)roi_width + 63) & ~(size_t)63;
    const size_t npx = (size_t)roi_width * roi_height, fbytes = (size_t)width * height * 3;
    hipError_t e = hipStreamCreateWithFlags(&rc->stream, hipStreamNonBlocking);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) {
        e = hipMalloc((void**)&rc->dMap1[k], npx * 4);
        if (e == hipSuccess) e = hipMalloc((void**)&rc->dMap2[k], npx * 2);
        if (e == hipSuccess) e = hipMalloc((void**)&rc->dRgb[k], fbytes + 16);
        if (e == hipSuccess) e = hipMalloc((void**)&rc->dGray[k], rc->gpitch * roi_height * (size_t)max_batch);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&rc->dOut, npx * 3);
    if (e == hipSuccess) e = hipHostMalloc((void**)&rc->hStage, 2 * fbytes + npx * 3 + 128 * (size_t)roi_height, hipHostMallocDefault);
    if (e == hipSuccess) {
        // crop the maps on the host (through the pinned area), one linear copy each
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            const int16_t* m1 = k ? map1_right : map1_left;
            const uint16_t* m2 = k ? map2_right : map2_left;
            int16_t* h1 = (int16_t*)rc->hStage;
            uint16_t* h2 = (uint16_t*)(rc->hStage + npx * 4);
            for (int y = 0; y < roi_height; ++y) {
                memcpy(h1 + (size_t)y * roi_width * 2, m1 + ((size_t)(roi_y + y) * width + roi_x) * 2, (size_t)roi_width * 4);
                memcpy(h2 + (size_t)y * roi_width, m2 + (size_t)(roi_y + y) * width + roi_x, (size_t)roi_width * 2);
            }
            e = hipMemcpy(rc->dMap1[k], h1, npx * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(rc->dMap2[k], h2, npx * 2, hipMemcpyHostToDevice);
        }
    }
    if (e != hipSuccess) {
        g_hip_err = std::string("rtdm_rectify_create: ") + hipGetErrorString(e);
        rtdm_rectify_destroy(rc);
        return e == hipErrorOutOfMemory ? RTDM_ERR_NOMEM : RTDM_ERR_HIP;
    }
    *out = rc;
    return RTDM_OK;
}

// host RGB frames -> dRgb[0/1] through the pinned staging area
static int rectify_upload(rtdm_rectify* rc, const uint8_t* a, size_t apitch, const uint8_t* b, size_t bpitch, hipStream_t s)
{
    const size_t row = (size_t)rc->W * 3, fbytes = row * rc->H;
    const uint8_t* src[2] = {a, b};
    const size_t pit[2] = {apitch, bpitch};
    for (int k = 0; k < 2; ++k) {
        if (!src[k]) continue;
        uint8_t* h = rc->hStage + k * fbytes;
        if (pit[k] == row) memcpy(h, src[k], fbytes);
        else for (int y = 0; y < rc->H; ++y) memcpy(h + (size_t)y * row, src[k] + (size_t)y * pit[k], row);
        HIPC(hipMemcpyAsync(rc->dRgb[k], h, fbytes, hipMemcpyHostToDevice, s));
    }
    return RTDM_OK;
}

static void rectify_gray_launch(rtdm_rectify* rc, const uint8_t* dl, const uint8_t* dr, int n, Plane8W ol, Plane8W orr, hipStream_t s)
{
    const size_t row = (size_t)rc->W * 3, fbytes = row * rc->H;
    RectifySrc L{dl, row, fbytes}, R{dr, row, fbytes};
    launch_rectify_gray(L, R, rc->dMap1[0], rc->dMap2[0], rc->dMap1[1], rc->dMap2[1], rc->W, rc->H, rc->rw, rc->rh, ol, orr, n, s);
}

int rtdm_rectify_gray(rtdm_rectify* rc, const uint8_t* rgb_left, size_t left_pitch, const uint8_t* rgb_right,
                      size_t right_pitch, uint8_t* left_rect, size_t left_rect_pitch, uint8_t* right_rect,
                      size_t right_rect_pitch)
{
    if (!rc || !rgb_left || !rgb_right || !left_rect || !right_rect) return RTDM_ERR_NULL;
    const size_t row = (size_t)rc->W * 3;
    if (left_pitch < row || right_pitch < row || left_rect_pitch < (size_t)rc->rw || right_rect_pitch < (size_t)rc->rw) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(rc->device));
    hipStream_t s = rc->stream;
    int st = rectify_upload(rc, rgb_left, left_pitch, rgb_right, right_pitch, s);
    if (st) return st;
    const size_t gframe = rc->gpitch * rc->rh;
    rectify_gray_launch(rc, rc->dRgb[0], rc->dRgb[1], 1, Plane8W{rc->dGray[0], rc->gpitch, gframe}, Plane8W{rc->dGray[1], rc->gpitch, gframe}, s);
    HIPC(hipGetLastError());
    // back through the page-locked area (2-D copies into pageable memory take the slow path)
    uint8_t* hl = rc->hStage + 2 * (size_t)rc->W * rc->H * 3;
    uint8_t* hr = hl + gframe;
    HIPC(hipMemcpyAsync(hl, rc->dGray[0], gframe, hipMemcpyDeviceToHost, s));
    HIPC(hipMemcpyAsync(hr, rc->dGray[1], gframe, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    for (int y = 0; y < rc->rh; ++y) {
        memcpy(left_rect + (size_t)y * left_rect_pitch, hl + (size_t)y * rc->gpitch, (size_t)rc->rw);
        memcpy(right_rect + (size_t)y * right_rect_pitch, hr + (size_t)y * rc->gpitch, (size_t)rc->rw);
    }
    return RTDM_OK;
}

int rtdm_rectify_rgb(rtdm_rectify* rc, int which, const uint8_t* rgb, size_t pitch, uint8_t* out, size_t out_pitch)
{
    if (!rc || !rgb || !out) return RTDM_ERR_NULL;
    const size_t row = (size_t)rc->W * 3, fbytes = row * rc->H;
    if ((which != 0 && which != 1) || pitch < row || out_pitch < (size_t)rc->rw * 3) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(rc->device));
    hipStream_t s = rc->stream;
    int st = rectify_upload(rc, rgb, pitch, nullptr, 0, s);
    if (st) return st;
    launch_rectify_rgb(RectifySrc{rc->dRgb[0], row, fbytes}, rc->dMap1[which], rc->dMap2[which], rc->W, rc->H, rc->rw, rc->rh,
                       Plane8W{rc->dOut, (size_t)rc->rw * 3, (size_t)rc->rw * 3 * rc->rh}, 1, s);
    HIPC(hipGetLastError());
    uint8_t* ho = rc->hStage + 2 * fbytes;
    HIPC(hipMemcpyAsync(ho, rc->dOut, (size_t)rc->rw * 3 * rc->rh, hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    for (int y = 0; y < rc->rh; ++y) memcpy(out + (size_t)y * out_pitch, ho + (size_t)y * rc->rw * 3, (size_t)rc->rw * 3);
    return RTDM_OK;
}

int rtdm_rectify_gray_device(rtdm_rectify* rc, int n, const uint8_t* d_rgb_left, const uint8_t* d_rgb_right,
                             uint8_t* d_left_rect, uint8_t* d_right_rect, void* hip_stream)
{
    if (!rc || !d_rgb_left || !d_rgb_right || !d_left_rect || !d_right_rect) return RTDM_ERR_NULL;
    if (n <= 0) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(rc->device));
    const size_t oframe = (size_t)rc->rw * rc->rh;
    rectify_gray_launch(rc, d_rgb_left, d_rgb_right, n, Plane8W{d_left_rect, (size_t)rc->rw, oframe},
                        Plane8W{d_right_rect, (size_t)rc->rw, oframe}, (hipStream_t)hip_stream);
    HIPC(hipGetLastError());
    return RTDM_OK;
}

// gray + remap + crop of n device frames into the rectifier's planes, then the matcher, chunk by chunk
static int rgb_chunks(rtdm_bm* bm, rtdm_rectify* rc, int n, const uint8_t* dl, const uint8_t* dr, Plane16W out, hipStream_t s)
{
    const size_t fbytes = (size_t)rc->W * rc->H * 3, gframe = rc->gpitch * rc->rh;
    const int chunk = std::min(bm->laneB, rc->maxB);
    for (int i0 = 0; i0 < n; i0 += chunk) {
        const int m = std::min(chunk, n - i0);
        rectify_gray_launch(rc, dl + (size_t)i0 * fbytes, dr + (size_t)i0 * fbytes, m,
                            Plane8W{rc->dGray[0], rc->gpitch, gframe}, Plane8W{rc->dGray[1], rc->gpitch, gframe}, s);
        HIPC(hipGetLastError());
        const int st = run_chunk(bm, bm->lane[0], m, Plane8{rc->dGray[0], rc->gpitch, gframe}, Plane8{rc->dGray[1], rc->gpitch, gframe},
                                 rc->rw, rc->rh, Plane16W{out.base + (size_t)i0 * out.frame_e, out.pitch_e, out.frame_e}, s);
        if (st) return st;
    }
    return RTDM_OK;
}

int rtdm_bm_compute_rgb_device(rtdm_bm* bm, rtdm_rectify* rc, int n, const uint8_t* d_rgb_left,
                               const uint8_t* d_rgb_right, int16_t* d_disp, void* hip_stream)
{
    if (!bm || !rc || !d_rgb_left || !d_rgb_right || !d_disp) return RTDM_ERR_NULL;
    if (n <= 0) return RTDM_ERR_BAD_SIZE;
    if (bm->device != rc->device) return RTDM_ERR_BAD_PARAM;
    int st = check_frame(bm, rc->rw, rc->rh);
    if (st) return st;
    HIPC(hipSetDevice(bm->device));
    const size_t oframe = (size_t)rc->rw * rc->rh;
    return rgb_chunks(bm, rc, n, d_rgb_left, d_rgb_right, Plane16W{d_disp, (size_t)rc->rw, oframe}, (hipStream_t)hip_stream);
}

int rtdm_bm_compute_rgb(rtdm_bm* bm, rtdm_rectify* rc, const uint8_t* rgb_left, size_t left_pitch,
                        const uint8_t* rgb_right, size_t right_pitch, int16_t* disp, size_t disp_pitch)
{
    if (!bm || !rc || !rgb_left || !rgb_right || !disp) return RTDM_ERR_NULL;
    if (bm->device != rc->device) return RTDM_ERR_BAD_PARAM;
    int st = check_frame(bm, rc->rw, rc->rh);
    if (st) return st;
    const size_t row = (size_t)rc->W * 3;
    if (left_pitch < row || right_pitch < row || disp_pitch < (size_t)rc->rw * 2) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(bm->device));
    hipStream_t s = bm->stream;
    st = rectify_upload(rc, rgb_left, left_pitch, rgb_right, right_pitch, s);
    if (st) return st;
    const size_t Ws = (size_t)((rc->rw + 7) & ~7);                     // the lane's internal plane (see run_chunk)
    st = rgb_chunks(bm, rc, 1, rc->dRgb[0], rc->dRgb[1], Plane16W{bm->dOut, Ws, Ws * (size_t)rc->rh}, s);
    if (st) return st;
    int16_t* hD = (int16_t*)(bm->hStage + 2 * bm->ppitch * (size_t)bm->maxH);
    HIPC(hipMemcpyAsync(hD, bm->dOut, Ws * rc->rh * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    for (int y = 0; y < rc->rh; ++y)
        memcpy((uint8_t*)disp + (size_t)y * disp_pitch, hD + (size_t)y * Ws, (size_t)rc->rw * sizeof(int16_t));
    return RTDM_OK;
}

// ---- object detection -> ROI (estimator.cpp:40-53) and the whole per-frame chain ---------------------------------
struct rtdm_objects {
    int W, H, device, maxRec;
    uint8_t *dRgb, *dMaskIn, *dMaskOut, *dT0, *dT1;
    void* dScratch;
    int* hRec;                    // pinned: [0] = count, then the first HEAD records
    std::vector<int> all;         // host copy of every record when there are more than HEAD
    hipStream_t stream;
};
static const int OBJ_HEAD = 512;  // records fetched together with the count

void rtdm_objects_destroy(rtdm_objects* ob)
{
    if (!ob) return;
    (void)hipSetDevice(ob->device);
    if (ob->stream) (void)hipStreamSynchronize(ob->stream);
    void* bufs[] = {ob->dRgb, ob->dMaskIn, ob->dMaskOut, ob->dT0, ob->dT1, ob->dScratch};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (ob->hRec) (void)hipHostFree(ob->hRec);
    if (ob->stream) (void)hipStreamDestroy(ob->stream);
    delete ob;
}

int rtdm_objects_create(int width, int height, int device, rtdm_objects** out)
{
    if (!out) return RTDM_ERR_NULL;
    *out = nullptr;
    if (width <= 0 || height <= 0 || width > 32767 || height > 32767 || (long)width * height >= (1L << 30)) return RTDM_ERR_BAD_SIZE;
    if (width > 8192) return RTDM_ERR_UNSUPPORTED;      // the component labelling keeps one row of ints in LDS
    int st = use_device(device);
    if (st) return st;
    rtdm_objects* ob = new (std::nothrow) rtdm_objects();
    if (!ob) return RTDM_ERR_NOMEM;
    ob->W = width; ob->H = height; ob->device = device;
    ob->maxRec = ((width + 1) / 2 + 1) * ((height + 1) / 2 + 1);          // more 8-connected components cannot exist
    const size_t px = (size_t)width * height;
    hipError_t e = hipStreamCreateWithFlags(&ob->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&ob->dRgb, px * 3 + 16);
    if (e == hipSuccess) e = hipMalloc((void**)&ob->dMaskIn, px);
    if (e == hipSuccess) e = hipMalloc((void**)&ob->dMaskOut, px);
    if (e == hipSuccess) e = hipMalloc((void**)&ob->dT0, px);
    if (e == hipSuccess) e = hipMalloc((void**)&ob->dT1, px);
    if (e == hipSuccess) e = hipMalloc(&ob->dScratch, cc_scratch_bytes(width, height, ob->maxRec));
    if (e == hipSuccess) e = hipHostMalloc((void**)&ob->hRec, (size_t)(1 + 6 * OBJ_HEAD) * sizeof(int) + px * 3, hipHostMallocDefault);
    if (e != hipSuccess) {
        g_hip_err = std::string("rtdm_objects_create: ") + hipGetErrorString(e);
        rtdm_objects_destroy(ob);
        return e == hipErrorOutOfMemory ? RTDM_ERR_NOMEM : RTDM_ERR_HIP;
    }
    *out = ob;
    return RTDM_OK;
}

// dRgb (crop, pitch 3W) -> mask -> morphology -> component boxes; synchronises `s` to read the boxes back.
static int objects_run(rtdm_objects* ob, const rtdm_hsv_range* range, int min_area, int zero_border, rtdm_region* boxes,
                       int max_boxes, int* nboxes, rtdm_region* roi, hipStream_t s)
{
    const int W = ob->W, H = ob->H;
    launch_hsv_inrange(ob->dRgb, (size_t)W * 3, W, H, range->low, range->high, ob->dMaskIn, (size_t)W, s);
    launch_morph_open_close(Plane8{ob->dMaskIn, (size_t)W, (size_t)W * H}, Plane8W{ob->dMaskOut, (size_t)W, (size_t)W * H},
                            ob->dT0, ob->dT1, W, H, 1, s);
    int *dCount = nullptr, *dRec = nullptr;
    launch_cc_boxes(ob->dMaskOut, (size_t)W, W, H, zero_border, ob->dScratch, ob->maxRec, &dCount, &dRec, s);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(ob->hRec, dCount, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPC(hipMemcpyAsync(ob->hRec + 1, dRec, (size_t)6 * std::min(OBJ_HEAD, ob->maxRec) * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPC(hipStreamSynchronize(s));
    const int nrec = std::min(ob->hRec[0], ob->maxRec);
    const int* rec = ob->hRec + 1;
    if (nrec > OBJ_HEAD) {
        ob->all.resize((size_t)6 * nrec);
        HIPC(hipMemcpy(ob->all.data(), dRec, (size_t)6 * nrec * sizeof(int), hipMemcpyDeviceToHost));
        rec = ob->all.data();
    }
    // reverse discovery order = descending first-pixel index; external components with area >= min_area only
    std::vector<const int*> keep;
    for (int i = 0; i < nrec; ++i) {
        const int* r = rec + 6 * i;
        if (r[5] && r[3] * r[4] >= min_area) keep.push_back(r);
    }
    std::sort(keep.begin(), keep.end(), [](const int* a, const int* b) { return a[0] > b[0]; });
    int max_x = -1000000, max_y = -1000000, min_x = 1000000, min_y = 1000000;
    for (size_t i = 0; i < keep.size(); ++i) {
        const int* r = keep[i];
        if ((int)i < max_boxes && boxes) { boxes[i].x = r[1]; boxes[i].y = r[2]; boxes[i].width = r[3]; boxes[i].height = r[4]; }
        min_x = std::min(min_x, r[1]); min_y = std::min(min_y, r[2]);
        max_x = std::max(max_x, r[1] + r[3]); max_y = std::max(max_y, r[2] + r[4]);
    }
    *nboxes = (int)keep.size();
    if (roi) { roi->x = min_x; roi->y = min_y; roi->width = max_x - min_x; roi->height = max_y - min_y; }
    return RTDM_OK;
}

int rtdm_objects_detect(rtdm_objects* ob, const uint8_t* rgb, size_t pitch, const rtdm_hsv_range* range, int min_area,
                        int zero_border, uint8_t* mask_out, size_t mask_pitch, rtdm_region* boxes, int max_boxes,
                        int* nboxes, rtdm_region* roi)
{
    if (!ob || !rgb || !range || !nboxes || (max_boxes > 0 && !boxes)) return RTDM_ERR_NULL;
    const size_t row = (size_t)ob->W * 3;
    if (pitch < row || max_boxes < 0 || (mask_out && mask_pitch < (size_t)ob->W)) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(ob->device));
    hipStream_t s = ob->stream;
    uint8_t* h = (uint8_t*)(ob->hRec + 1 + 6 * OBJ_HEAD);
    for (int y = 0; y < ob->H; ++y) memcpy(h + (size_t)y * row, rgb + (size_t)y * pitch, row);
    HIPC(hipMemcpyAsync(ob->dRgb, h, row * ob->H, hipMemcpyHostToDevice, s));
    int st = objects_run(ob, range, min_area, zero_border, boxes, max_boxes, nboxes, roi, s);
    if (st) return st;
    if (mask_out) {                                     // through the page-locked area (the upload is long done)
        HIPC(hipMemcpyAsync(h, ob->dMaskOut, (size_t)ob->W * ob->H, hipMemcpyDeviceToHost, s));
        HIPC(hipStreamSynchronize(s));
        for (int y = 0; y < ob->H; ++y) memcpy(mask_out + (size_t)y * mask_pitch, h + (size_t)y * ob->W, (size_t)ob->W);
    }
    return RTDM_OK;
}

int rtdm_estimate_frame(rtdm_bm* bm, rtdm_rectify* rc, rtdm_objects* ob, const uint8_t* rgb_left, size_t left_pitch,
                        const uint8_t* rgb_right, size_t right_pitch, const double* Q, const rtdm_hsv_range* range,
                        int min_area, int zero_border, double calibration_unit, rtdm_region* boxes, double* mean_cm,
                        int* counts, int max_boxes, int* nboxes, int16_t* disp, size_t disp_pitch)
{
    if (!bm || !rc || !ob || !rgb_left || !rgb_right || !Q || !range || !boxes || !mean_cm || !counts || !nboxes) return RTDM_ERR_NULL;
    if (bm->device != rc->device || bm->device != ob->device) return RTDM_ERR_BAD_PARAM;
    if (ob->W != rc->rw || ob->H != rc->rh || max_boxes <= 0) return RTDM_ERR_BAD_SIZE;
    int st = check_frame(bm, rc->rw, rc->rh);
    if (st) return st;
    const int W = rc->rw, H = rc->rh;
    const size_t row = (size_t)rc->W * 3, fbytes = row * rc->H;
    if (left_pitch < row || right_pitch < row || (disp && disp_pitch < (size_t)W * 2)) return RTDM_ERR_BAD_SIZE;
    HIPC(hipSetDevice(bm->device));
    hipStream_t s = bm->stream;
    st = rectify_upload(rc, rgb_left, left_pitch, rgb_right, right_pitch, s);
    if (st) return st;
    // estimator.cpp:29-39: both gray crops and the colour crop of the left frame
    const size_t gframe = rc->gpitch * H;
    rectify_gray_launch(rc, rc->dRgb[0], rc->dRgb[1], 1, Plane8W{rc->dGray[0], rc->gpitch, gframe}, Plane8W{rc->dGray[1], rc->gpitch, gframe}, s);
    launch_rectify_rgb(RectifySrc{rc->dRgb[0], row, fbytes}, rc->dMap1[0], rc->dMap2[0], rc->W, rc->H, W, H,
                       Plane8W{ob->dRgb, (size_t)W * 3, (size_t)W * 3 * H}, 1, s);
    // estimator.cpp:40-53
    rtdm_region roi;
    st = objects_run(ob, range, min_area, zero_border, boxes, max_boxes, nboxes, &roi, s);
    if (st) return st;
    if (*nboxes == 0) return RTDM_OK;                     // estimator.cpp:48: nothing to measure in this frame
    const int nreg = std::min(std::min(*nboxes, max_boxes), RTDM_MAX_REGIONS);
    // estimator.cpp:54-56
    st = rtdm_bm_set_roi(bm, 1, roi.x, roi.y, roi.width, roi.height);
    if (st) return st;
    const size_t Ws = (size_t)((W + 7) & ~7);
    st = run_chunk(bm, bm->lane[0], 1, Plane8{rc->dGray[0], rc->gpitch, gframe}, Plane8{rc->dGray[1], rc->gpitch, gframe}, W, H,
                   Plane16W{bm->dOut, Ws, Ws * (size_t)H}, s);
    if (st) return st;
    // estimator.cpp:75-77
    int flat[4 * RTDM_MAX_REGIONS], maxh = 1;
    st = check_regions(boxes, nreg, W, H, flat, &maxh);
    if (st) return st;
    DepthQ q; std::copy(Q, Q + 16, q.q);
    launch_depth_stats(bm->dOut, Ws, W, H, q, ob->dMaskOut, (size_t)W, flat, nreg, bm->maxH, calibration_unit, bm->dDepth,
                       mean_cm, counts, s);
    int16_t* hD = (int16_t*)(bm->hStage + 2 * bm->ppitch * (size_t)bm->maxH);
    if (disp) HIPC(hipMemcpyAsync(hD, bm->dOut, Ws * H * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(s));
    if (disp)
        for (int y = 0; y < H; ++y) memcpy((uint8_t*)disp + (size_t)y * disp_pitch, hD + (size_t)y * Ws, (size_t)W * sizeof(int16_t));
    return RTDM_OK;
}

// ---- synthetic stream ------------------------------------------------------------------------
int rtdm_synth_pairs_device(uint64_t seed, int first_frame, int n, int width, int height, int numDisparities,
                            uint8_t* d_left, uint8_t* d_right, size_t pitch, size_t frame_stride, int device,
                            void* hip_stream)
{
    if (!d_left || !d_right) return RTDM_ERR_NULL;
    if (n <= 0 || width <= 0 || height <= 0 || pitch < (size_t)width) return RTDM_ERR_BAD_SIZE;
    int rc = use_device(device);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    void* scratch = nullptr;
    HIPC(hipMalloc(&scratch, synth_scratch_bytes(n)));
    Plane8W L{d_left, pitch, frame_stride}, R{d_right, pitch, frame_stride};
    launch_synth(seed, first_frame, n, width, height, numDisparities, L, R, scratch, s);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    HIPC(e);
    return RTDM_OK;
}

}  // extern "C"
