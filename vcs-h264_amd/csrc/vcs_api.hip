// vcs_api.hip -- the C ABI of libvcs_hip.so (include/vcs_hip.h): contexts, HBM
// buffers, launch chains, timing.  Host C++ + HIP runtime only; no PyTorch.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include <utility>

#include "../../include/vcs_hip.h"
#include "dct_kernels.hpp"
#include "intra_kernels.hpp"
#include "me_fast.hpp"
#include "me_fast8.hpp"
#include "me_roll.hpp"
#include "me_group.hpp"
#include "me_group8.hpp"
#include "me_bits.hpp"
#include "me_bits8.hpp"
#include "me_kernels.hpp"
#include "me_ext.hpp"

using namespace vcs;

namespace {

thread_local std::string g_err;

struct EvPair {
    hipEvent_t a = nullptr, b = nullptr;
    int slot = 0;
    bool attached = false;  // events ride on one kernel's dispatch packet (hipExtLaunchKernelGGL)
};

}  // namespace

struct vcs_group;
struct PlaneSlab;

struct AsyncSlot {  // one frame in flight through vcs_encode_p_async
    uint8_t *d_cur = nullptr;
    int16_t *d_mv = nullptr, *h_mv = nullptr;  // N x 2 vectors + the static count behind them
    hipEvent_t ev_in = nullptr, ev_kern = nullptr, ev_out = nullptr;
    bool busy = false;
};

struct vcs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Geom g{};
    int64_t thresh64 = 2000;
    int coeff_mode = VCS_COEFF_F64;
    int search = VCS_SEARCH_GRID, cost = VCS_COST_WRAP;  // vcs_set_search (extensions)
    size_t F = 0, Fs = 0;  // frame bytes, padded frame stride
    size_t me_lds = 0;     // dynamic LDS of the generic ME kernel
    bool ref_valid = false;
    bool me_fast_ok = false;
    bool me_fast8_ok = false;
    uint32_t *d_sums = nullptr;  // k_me16: per-reference sum(r) table
    unsigned long long *d_clk = nullptr;  // k_me16g: longest workgroup lifetime in shader cycles (timing on)
    bool sums_valid = false;
    // k_me16b (me_bits.hpp): the reference cut into bit-plane chunks (per reference frame), the
    // current blocks as bit planes + {static-test sum, sum(c)} per block (scratch, per launch)
    uint32_t *d_rtab = nullptr, *d_cpl = nullptr;   // d_rtab: the chunk table in use (one of tabs[].tab)
    uint2 *d_binfo = nullptr;
    int cpl_frames = 0;      // frames d_cpl / d_binfo hold
    bool rtab_valid = false;
    bool bits_unusable = false;   // the scratch could not be allocated: the shared-window kernels serve this context
    // Two chunk tables, each tied to the reference buffer it was cut from.  The second exists for the
    // SHADOW BUILD: once the caller has declared the alternate reference buffer complete
    // (vcs_ref_next_ready), the next GOP's table is cut on a side stream beside this GOP's search kernel
    // (vector-issue bound, ~15 % of its issue slots idle and a drain phase at its end) instead of as a
    // serial 28 us kernel in front of the next GOP; vcs_ref_swap / the group's buffer exchange then finds
    // it `fresh`.
    struct RefTable {
        uint32_t *tab = nullptr;
        const uint8_t *for_ref = nullptr;   // reference buffer the table was cut from
        bool fresh = false;                 // cut from content the search has not used yet (shadow build)
        bool pending = false;               // ev_done has to be waited for before the table is read
        hipEvent_t ev_done = nullptr;
    } tabs[2];
    int l3_mode = -1;               // vcs_set_l3_prefetch: -1 = the environment's choice (default: inside the search launch)
    bool next_ready = false;        // the alternate reference buffer holds the next GOP's I frame
    hipEvent_t ev_next = nullptr;   // ... as of this event (recorded on the stream that wrote it)
    hipStream_t s_side = nullptr;
    hipEvent_t ev_side_go = nullptr;
    int cus = 256;         // compute units of the device
    // single-frame buffers
    uint8_t *d_ref = nullptr, *d_cur = nullptr, *d_a = nullptr, *d_b = nullptr;
    uint8_t *d_ref_alt = nullptr;  // second reference buffer (vcs_ref_swap), allocated on demand
    int16_t *d_mv = nullptr;
    int32_t *d_coords = nullptr;
    unsigned long long *d_cost = nullptr;
    uint32_t *d_ns = nullptr;
    void *d_planes = nullptr;
    double *d_intra = nullptr;  // intra prediction outputs: 4 planes + modes, float64
    // vcs_intra_batch_*: n I frames resident -- per frame three uint8 planes in (ib_in) and, float64, the luma4
    // pair, the luma16 pair, the four chroma planes, the transform of the luma4 residual and the three mode maps
    uint8_t *ib_in = nullptr;
    double *ib_out = nullptr;
    int ib_frames = 0;
    double *d_DQ = nullptr;  // D[64] + Q[192] + 1/Q[192]
    double hDQ[448];
    bool fastdiv = false;    // Q table admits the 3-FMA exact division
    // GOP buffers
    int slots = 0;
    bool gop_dec = false;
    uint8_t *g_cur = nullptr, *g_dec = nullptr;
    int16_t *g_mv = nullptr;
    uint32_t *g_ns = nullptr;
    void *g_planes = nullptr;
    hipEvent_t xev = nullptr;  // cross-stream ordering with a foreign (torch / RCCL) stream
    vcs_group *group = nullptr;  // multi-GPU group this context is a shard of (vcs_multi.hpp)
    hipEvent_t ev_me_done = nullptr;  // group members: the last GOP's search kernel has finished
    bool me_done_rec = false;
    // pipelined single-frame feeder (vcs_async_reserve)
    hipStream_t s_in = nullptr, s_out = nullptr;
    std::vector<AsyncSlot> aslots;
    PlaneSlab *slab = nullptr;   // where the next resident planes handle is carved from
    // CU-split pipelining of the GOP path (VCS_SPLIT_CUS = c > 0): the HBM-bound encoder tail
    // of GOP k runs on c CUs (c / 8 per XCD) beside the vector-issue-bound search of GOP k+1
    // on the other 256 - c; two CU-masked streams, ordered by events
    int split_cus = 0;
    hipStream_t s_me = nullptr, s_tail = nullptr;
    hipEvent_t ev_main = nullptr, ev_me = nullptr, ev_tail = nullptr;
    bool split_pending = false;  // work on s_me / s_tail that c->stream has not been joined with
    int tail_first = 0, tail_n = 0;  // slots the last tail launch reads and writes
    // timing
    bool timing = false;
    std::vector<EvPair> pending;
    std::vector<EvPair> pool;
    double acc_ms[VCS_T_COUNT] = {0, 0, 0, 0, 0};
    uint64_t launches[VCS_T_COUNT] = {0, 0, 0, 0, 0};
    bool pf_in_search = false;  // in: the caller wants the frames prefetched; out: the search kernel does it
    bool t_no_count = false;   // a later sub-batch of one vcs_gop_encode call: time it, do not count it
    std::string err;
};

namespace {

int fail(vcs_ctx *c, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    g_err = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                              \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail((c), VCS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

size_t coeff_size(int mode) { return mode == VCS_COEFF_F64 ? 8 : mode == VCS_COEFF_F32 ? 4 : 2; }
size_t planes_bytes(const vcs_ctx *c) { return (size_t)3 * c->g.H * c->g.W * coeff_size(c->coeff_mode); }
size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// JPEG tables of DCTcompressor.py:11-27 (data; [1][5] of QY is 48 there).
const int kQY[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  48,  60,  55,
                     14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                     18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                     49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const int kQC[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                     24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

int py_round_div3(int bs) {  // Python round(bs / 3): half to even (motion.py:132)
    return (int)nearbyint((double)bs / 3.0);
}

// LDS demand of k_me_generic for the current geometry; 0 if it cannot fit.
size_t me_generic_lds(const Geom &g, int *wpitch_out, int ext = 0) {
    // ext: extra rows / columns behind the reference's window (the three-step search of
    // me_ext.hpp parks y - win .. y + win + bs)
    const int rowb = 3 * g.bs, cpitch = (rowb + 3) & ~3;
    const int wpitch = ((3 * (2 * g.win + ext) + 3 + 3) & ~3) + 8;
    size_t lds = round_up((size_t)g.bs * cpitch, 16) + (size_t)(2 * g.win + ext) * wpitch + 16;
    if (wpitch_out) *wpitch_out = wpitch;
    return lds;
}

int refresh_me_geometry(vcs_ctx *c) {
    Geom &g = c->g;
    if (g.win < 1 || g.step < 1) return fail(c, VCS_E_ARG, "win and step must be >= 1");
    size_t lds = me_generic_lds(g, &g.wpitch, c->search == VCS_SEARCH_TSS ? g.bs : 0);
    if (lds > 160 * 1024)
        return fail(c, VCS_E_ARG, "search window (bs=%d win=%d) needs %zu B of LDS, > 160 KiB",
                    g.bs, g.win, lds);
    c->me_lds = lds;
    g.thresh = c->thresh64 < 0 ? 0u
               : c->thresh64 > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)c->thresh64;
    g.never_static = c->thresh64 < 0;
    // VCS_ME_GENERIC=1 forces the generic kernel (A/B measurements, debugging)
    const char *force = getenv("VCS_ME_GENERIC");
    const bool reference_search = c->search == VCS_SEARCH_GRID && c->cost == VCS_COST_WRAP;
    c->me_fast_ok = reference_search && me_fast_supported(g) && !(force && force[0] == '1');
    c->me_fast8_ok = reference_search && me_fast8_supported(g) && !(force && force[0] == '1');
    return VCS_OK;
}

// Fills 1/Q behind Q and decides whether the 3-FMA division is exact for it:
// every Q within [2^-60, 2^60] (positive) and no all-ones significand (Markstein).
void refresh_q(vcs_ctx *c) {
    bool ok = true;
    for (int i = 0; i < 192; i++) {
        const double q = c->hDQ[64 + i];
        c->hDQ[256 + i] = 1.0 / q;
        uint64_t bits;
        memcpy(&bits, &q, 8);
        if (!(q >= 0x1p-60 && q <= 0x1p60) || (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull)
            ok = false;
    }
    // Default when the table admits it; VCS_FAST_DIV=0 forces the IEEE division sequence
    // (tests run both).  k_fdct is co-limited by vector issue once its stores are
    // non-temporal: 4 operations per quotient instead of 11 + a quarter-rate v_rcp_f64.
    const char *e = getenv("VCS_FAST_DIV");
    c->fastdiv = ok && !(e && e[0] == '0');
}

// ---- timing brackets -----------------------------------------------------
// Two forms.  Recorded: hipEventRecord before and after whatever is launched in between
// (each record is a packet of its own on the queue, ~5 us).  Attached (attach = true): the
// caller passes ev->a / ev->b to hipExtLaunchKernelGGL, which takes the start and end
// timestamps of that one kernel's dispatch packet -- no extra packets, and exactly the
// kernel's duration.  With timing off both are no-ops and ev->a / ev->b stay null.
int t_begin(vcs_ctx *c, int slot, EvPair *ev, bool attach = false) {
    if (!c->t_no_count) c->launches[slot]++;
    ev->a = ev->b = nullptr;
    ev->attached = attach;
    if (!c->timing) return VCS_OK;
    if (!c->pool.empty()) {
        *ev = c->pool.back();
        c->pool.pop_back();
    } else {
        HIP_TRY(c, hipEventCreate(&ev->a));
        HIP_TRY(c, hipEventCreate(&ev->b));
    }
    ev->slot = slot;
    ev->attached = attach;  // (the pooled pair carried its previous use's flag)
    if (!attach) HIP_TRY(c, hipEventRecord(ev->a, c->stream));
    return VCS_OK;
}
int t_end(vcs_ctx *c, EvPair *ev) {
    HIP_TRY(c, hipGetLastError());
    if (!c->timing || !ev->a) return VCS_OK;
    if (!ev->attached) HIP_TRY(c, hipEventRecord(ev->b, c->stream));
    c->pending.push_back(*ev);
    return VCS_OK;
}

// ---- launches --------------------------------------------------------------
template <int BS_T>
int launch_me_generic_t(vcs_ctx *c, const uint8_t *ref, const uint8_t *curs, size_t stride,
                        int n, int16_t *mv, uint32_t *ns, unsigned long long *cost) {
    auto kern = k_me_generic<BS_T>;
    if (c->me_lds > 48 * 1024)
        HIP_TRY(c, hipFuncSetAttribute((const void *)kern,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->me_lds));
    hipLaunchKernelGGL(kern, dim3(c->g.N, n), dim3(64), c->me_lds, c->stream, ref, curs, stride,
                       c->g, mv, ns, cost);
    return VCS_OK;
}

// Scratch of the bit-plane search for n frames (grown on demand; allocation synchronises, so it
// happens on the first launch of a given size only).
static size_t bits_table_bytes(const vcs_ctx *c) {
    return c->g.bs == 8 ? me_bits8_rtab_bytes(c->g) : me_bits_rtab_bytes(c->g);   // (a context has one block size: the 8x8 and 16x16 forms never mix)
}
// On an allocation failure the bit-plane path is marked unusable for this context and the caller falls
// through to the shared-window kernels, which need no scratch (VCS_TEST_BITS_NOMEM=1 forces that path).
static int ensure_bits(vcs_ctx *c, int n) {
    static const bool test_nomem = [] { const char *e = getenv("VCS_TEST_BITS_NOMEM"); return e && e[0] == '1'; }();
    const bool b8 = c->g.bs == 8;
    auto give_up = [&] {
        (void)hipGetLastError();
        c->bits_unusable = true;
        return VCS_E_HIP;
    };
    if (test_nomem) return give_up();
    if (!c->tabs[0].tab && hipMalloc(&c->tabs[0].tab, bits_table_bytes(c)) != hipSuccess) return give_up();
    if (n > c->cpl_frames) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(c->d_cpl);
        hipFree(c->d_binfo);
        c->d_cpl = nullptr;
        c->d_binfo = nullptr;
        c->cpl_frames = 0;
        if (hipMalloc(&c->d_cpl, b8 ? me_bits8_cpl_bytes(c->g, n) : me_bits_cpl_bytes(c->g, n)) != hipSuccess ||
            hipMalloc(&c->d_binfo, b8 ? me_bits8_binfo_bytes(c->g, n) : me_bits_binfo_bytes(c->g, n)) != hipSuccess) {
            hipFree(c->d_cpl);
            c->d_cpl = nullptr;
            return give_up();
        }
        c->cpl_frames = n;
    }
    return VCS_OK;
}

static void bits_cut_table(vcs_ctx *c, hipStream_t st, const uint8_t *ref, uint32_t *tab) {
    if (c->g.bs == 8)
        launch_bits_ref8(c->g, st, ref, c->Fs + me_fast_ref_slack(c->g), tab, tab + me_bits8_table_words(c->g));
    else
        launch_bits_ref(c->g, st, ref, c->Fs + me_fast_ref_slack(c->g), tab, 0, c->g.nbx, tab + me_bits_table_words(c->g));
}

// The chunk table of `ref` (one band), whose content is new since the table in use was cut: the one cut in
// the shadow of the previous GOP's search if there is one, else cut now, in front of the search.
static int bits_table_for(vcs_ctx *c, const uint8_t *ref) {
    int i = c->tabs[0].for_ref == ref && c->tabs[0].fresh ? 0 : c->tabs[1].for_ref == ref && c->tabs[1].fresh ? 1 : -1;
    if (i < 0) {   // cut it now; into the table that is not waiting for its turn
        i = c->tabs[0].fresh ? 1 : 0;
        vcs_ctx::RefTable &t = c->tabs[i];
        if (!t.tab) HIP_TRY(c, hipMalloc(&t.tab, bits_table_bytes(c)));
        if (t.pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, t.ev_done, 0));
        t.pending = false;
        bits_cut_table(c, c->stream, ref, t.tab);
        HIP_TRY(c, hipGetLastError());
        t.for_ref = ref;
    }
    vcs_ctx::RefTable &t = c->tabs[i];
    if (t.pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, t.ev_done, 0));
    t.pending = false;
    t.fresh = false;
    c->d_rtab = t.tab;
    c->rtab_valid = true;
    return VCS_OK;
}

// The shadow build: the next GOP's table, from the alternate reference buffer, on a side stream that
// starts where the main stream is now (right in front of the search kernel).  Failures (no memory for the
// second table) just leave the table to be cut in front of the next search.
static void bits_shadow_build(vcs_ctx *c) {
    static const bool off = [] { const char *e = getenv("VCS_BITS_SHADOW"); return e && e[0] == '0'; }();
    if (off || !c->next_ready || !c->d_ref_alt) return;
    c->next_ready = false;
    const int j = c->d_rtab == c->tabs[0].tab ? 1 : 0;
    vcs_ctx::RefTable &t = c->tabs[j];
    if (!t.tab && hipMalloc(&t.tab, bits_table_bytes(c)) != hipSuccess) { (void)hipGetLastError(); t.tab = nullptr; return; }
    if (!c->s_side) {
        if (hipStreamCreateWithFlags(&c->s_side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_side_go, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
    }
    if (!t.ev_done && hipEventCreateWithFlags(&t.ev_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return; }
    hipEventRecord(c->ev_side_go, c->stream);
    hipStreamWaitEvent(c->s_side, c->ev_side_go, 0);
    if (c->ev_next) hipStreamWaitEvent(c->s_side, c->ev_next, 0);   // the I frame's own arrival (a collective's stream)
    bits_cut_table(c, c->s_side, c->d_ref_alt, t.tab);
    hipEventRecord(t.ev_done, c->s_side);
    (void)hipGetLastError();
    t.for_ref = c->d_ref_alt;
    t.fresh = true;
    t.pending = true;
}

// The content of reference buffer `buf` is about to change / has changed outside the swap protocol: a
// table cut from it in the shadow of a search is stale (and, if still being cut, is waited for first).
static int bits_tables_stale(vcs_ctx *c, const uint8_t *buf) {
    for (auto &t : c->tabs)
        if (t.for_ref == buf) {
            if (t.pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, t.ev_done, 0));
            t.pending = false;
            t.fresh = false;
            t.for_ref = nullptr;
        }
    return VCS_OK;
}

// Motion search for n frames (frame f at curs + f*stride) against `ref`.
int launch_me(vcs_ctx *c, const uint8_t *ref, const uint8_t *curs, size_t stride, int n,
              int16_t *mv, uint32_t *ns, unsigned long long *cost, bool force_generic = false) {
    if (c->g.N == 0 || n == 0) return VCS_OK;
    bool bits = c->me_fast_ok && !force_generic && !c->bits_unusable && me_bits_worthwhile(c->g, n);
    // the per-frame static-block counters start at zero; the rolling kernel's primer does it
    bool bits8 = c->me_fast8_ok && !force_generic && !c->bits_unusable && me_bits8_worthwhile(c->g, n);
    if ((bits || bits8) && ensure_bits(c, n) != VCS_OK) {
        if (!c->bits_unusable) return VCS_E_HIP;   // (a failure other than the allocation: c->err has it)
        bits = bits8 = false;
    }
    const bool roll = !bits && c->me_fast_ok && !force_generic && me16r_worthwhile(c->g, c->cus, n);
    if (!roll && !bits && !bits8) HIP_TRY(c, hipMemsetAsync(ns, 0, sizeof(uint32_t) * n, c->stream));
    // per-reference sum(r) table, once per reference frame (all P frames of a GOP share it);
    // not part of VCS_T_ME, which times the search kernel alone
    // (the one-band bit-plane search makes the 16x16 table itself, from k_bits_ref's row-pair sums)
    const bool fold_sums = bits && !c->sums_valid && me_bits_bands(c->g) == 1;
    const bool fold_sums8 = bits8 && !c->sums_valid;   // the same at 8x8 (k_bits_ref8's chunk sums)
    if (fold_sums || fold_sums8) c->rtab_valid = false;
    if (!c->sums_valid && !fold_sums && !fold_sums8 && !force_generic && (c->me_fast_ok || c->me_fast8_ok)) {
        if (c->me_fast_ok) launch_me_fast_sums(c->g, c->stream, c->cus, ref, c->d_sums);
        else launch_me_fast8_sums(c->g, c->stream, c->cus, ref, c->d_sums);
        HIP_TRY(c, hipGetLastError());
        c->sums_valid = true;
        c->rtab_valid = false;
    }
    if (bits) {
        int rc;
        const int nb = me_bits_bands(c->g), bc = me_bits_band_cols(c->g);
        // one band: the table of a new reference first (k_bits_cur adds its sums up); several: per launch, below
        if (nb == 1) {
            if ((!c->rtab_valid || c->tabs[c->d_rtab == c->tabs[1].tab ? 1 : 0].for_ref != ref) &&
                (rc = bits_table_for(c, ref)))
                return rc;
        } else {
            c->d_rtab = c->tabs[0].tab;
        }
        uint32_t *r2 = c->d_rtab + me_bits_table_words(c->g);   // row-pair sums behind the chunk table
        EvPair evp;
        if ((rc = t_begin(c, VCS_T_ME_PREP, &evp, /*attach=*/true))) return rc;
        uint32_t *ns_part = (uint32_t *)(c->d_binfo + (size_t)n * c->g.N);   // behind the n * N block records
        launch_bits_cur(c->g, c->stream, curs, stride, ref, n, c->d_cpl, c->d_binfo, ns_part,
                        fold_sums ? r2 : nullptr, c->d_sums, evp.a, evp.b);
        if ((rc = t_end(c, &evp))) return rc;
        if (fold_sums) c->sums_valid = true;
        // The caller's Infinity Cache prefetch of the frames (vcs_gop_encode) rides at the END of the
        // (last) search launch (prefetch workgroups dispatched behind the search's own, me_bits.hpp).
        // Measured alternatives, 29 x 1080p: k_bits_cur's own reads of the frames are gone from the
        // cache when the tail starts (tail 0.37 instead of 0.29 ms: the search moves 300 MB of planes
        // and chunks in between); so is a k_l3_prefetch started beside the search on a second stream;
        // LDS-direct loads issued by every search wave on its way out: tail 0.35, search + 0.016 ms;
        // a separate pass between search and tail: its own 0.04 ms.
        static const bool pf_tail_wgs = [] { const char *e = getenv("VCS_BITS_PF"); return !(e && e[0] == '0'); }();
        const bool pf = c->pf_in_search && pf_tail_wgs && (size_t)n * stride < 0xFFFFFFFFull;
        c->pf_in_search = pf;
        // The chunk table covers one band of block columns (the whole picture up to 1 GiB of table: 1080p
        // 124 MB, 4K 497 MB): with one band it is built once per reference frame, like the sum table; with
        // several (larger frames, VCS_BITS_BAND_KB) it is rebuilt band by band in every launch.
        const bool nc = c->t_no_count;
        if (nb == 1) bits_shadow_build(c);   // the next GOP's table, beside this search
        for (int band = 0; band < nb && !rc; band++) {
            const int a0 = band * bc, na = std::min(bc, c->g.nbx - a0);
            if (nb > 1) {
                launch_bits_ref(c->g, c->stream, ref, c->Fs + me_fast_ref_slack(c->g), c->d_rtab, a0, na);
                HIP_TRY(c, hipGetLastError());
            }
            EvPair ev;
            c->t_no_count = nc || band > 0;   // timing: one "launch" per call, however many bands
            if ((rc = t_begin(c, VCS_T_ME, &ev, /*attach=*/true))) break;
            launch_bits_search(c->g, c->stream, c->d_rtab, c->d_cpl, c->d_binfo, c->d_sums, n, mv, ns_part, ns, cost,
                               c->timing ? c->d_clk : nullptr, a0, na, pf && band == nb - 1 ? curs : nullptr,
                               (size_t)n * stride, ev.a, ev.b);
            rc = t_end(c, &ev);
        }
        c->t_no_count = nc;
        if (nb > 1) c->rtab_valid = false;
        return rc;
    }
    if (bits8) {   // the 8x8 bit-plane search (me_bits8.hpp): chunk table per reference, planes, search
        int rc;
        if ((!c->rtab_valid || c->tabs[c->d_rtab == c->tabs[1].tab ? 1 : 0].for_ref != ref) &&
            (rc = bits_table_for(c, ref)))
            return rc;
        uint32_t *r2 = c->d_rtab + me_bits8_table_words(c->g);   // chunk sums behind the chunk table
        uint32_t *ns_part = (uint32_t *)(c->d_binfo + (size_t)n * c->g.N);
        EvPair evp, ev;
        if ((rc = t_begin(c, VCS_T_ME_PREP, &evp, /*attach=*/true))) return rc;
        launch_bits_cur8(c->g, c->stream, curs, stride, ref, n, c->d_cpl, c->d_binfo, ns_part,
                         fold_sums8 ? r2 : nullptr, c->d_sums, evp.a, evp.b);
        if ((rc = t_end(c, &evp))) return rc;
        if (fold_sums8) c->sums_valid = true;
        static const bool pf_tail_wgs = [] { const char *e = getenv("VCS_BITS_PF"); return !(e && e[0] == '0'); }();
        const bool pf = c->pf_in_search && pf_tail_wgs && (size_t)n * stride < 0xFFFFFFFFull;
        c->pf_in_search = pf;
        bits_shadow_build(c);   // the next GOP's table, beside this search
        if ((rc = t_begin(c, VCS_T_ME, &ev, /*attach=*/true))) return rc;
        launch_bits_search8(c->g, c->stream, c->d_rtab, c->d_cpl, c->d_binfo, c->d_sums, n, mv, ns_part, ns, cost,
                            pf ? curs : nullptr, (size_t)n * stride, ev.a, ev.b);
        return t_end(c, &ev);
    }
    const bool fast16 = c->me_fast_ok && !force_generic;
    const bool group8 = c->me_fast8_ok && !force_generic && me8g_worthwhile(c->g, c->cus, n);
    bool pf = c->pf_in_search;   // prefetch request: only the shared-window kernels can serve it
    c->pf_in_search = false;
    EvPair ev;
    int rc = t_begin(c, VCS_T_ME, &ev, /*attach=*/fast16 || group8);  // the search kernel alone
    if (rc) return rc;
    if (roll) launch_me_roll_primer(c->stream, c->cus, ns, n);
    if (fast16) {
        rc = roll ? (me16g_worthwhile(c->g, c->cus, n)
                         ? launch_me_group(c->g, c->stream, c->cus, ref, curs, stride, n, c->d_sums, mv,
                                           ns, cost, c->timing ? c->d_clk : nullptr, &pf, ev.a, ev.b)
                         : launch_me_roll(c->g, c->stream, c->cus, ref, curs, stride, n, c->d_sums, mv,
                                          ns, cost, ev.a, ev.b))
                  : launch_me_fast(c->g, c->stream, c->cus, ref, curs, stride, n, c->d_sums, mv, ns, cost,
                                   ev.a, ev.b);
        if (rc) return fail(c, VCS_E_HIP, "fast ME launch failed");
        c->pf_in_search = pf && roll && me16g_worthwhile(c->g, c->cus, n);
    } else if (group8) {
        rc = launch_me_group8(c->g, c->stream, c->cus, ref, curs, stride, n, c->d_sums, mv, ns, cost, &pf, ev.a, ev.b);
        if (rc) return fail(c, VCS_E_HIP, "fast ME launch failed");
        c->pf_in_search = pf;
    } else if (c->me_fast8_ok && !force_generic) {
        rc = launch_me_fast8(c->g, c->stream, c->cus, ref, curs, stride, n, c->d_sums, mv, ns, cost);
        if (rc) return fail(c, VCS_E_HIP, "fast ME launch failed");
    } else if (c->search != VCS_SEARCH_GRID || c->cost != VCS_COST_WRAP) {
        if (c->me_lds > 48 * 1024) {
#define VCS_EXT_ATTR(S, C)                                                                      \
    HIP_TRY(c, hipFuncSetAttribute((const void *)k_me_ext<S, C>,                                \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->me_lds))
            VCS_EXT_ATTR(kSearchGrid, kCostSad); VCS_EXT_ATTR(kSearchTss, kCostWrap); VCS_EXT_ATTR(kSearchTss, kCostSad);
#undef VCS_EXT_ATTR
        }
#define VCS_EXT(S, C)                                                                           \
    hipLaunchKernelGGL((k_me_ext<S, C>), dim3(c->g.N, n), dim3(64), c->me_lds, c->stream, ref, curs, \
                       stride, c->g, mv, ns, cost)
        if (c->search == VCS_SEARCH_TSS) { if (c->cost == VCS_COST_SAD) VCS_EXT(kSearchTss, kCostSad); else VCS_EXT(kSearchTss, kCostWrap); }
        else VCS_EXT(kSearchGrid, kCostSad);
#undef VCS_EXT
    } else {
        switch (c->g.bs) {
            case 16: rc = launch_me_generic_t<16>(c, ref, curs, stride, n, mv, ns, cost); break;
            case 8: rc = launch_me_generic_t<8>(c, ref, curs, stride, n, mv, ns, cost); break;
            case 4: rc = launch_me_generic_t<4>(c, ref, curs, stride, n, mv, ns, cost); break;
            default: rc = launch_me_generic_t<0>(c, ref, curs, stride, n, mv, ns, cost); break;
        }
        if (rc) return rc;
    }
    return t_end(c, &ev);
}

// strip -> workgroup mapping of k_fdct / k_idct (dct_kernels.hpp, tail_unit)
static int tail_map() {
    static const int m = [] {
        const char *e = getenv("VCS_TAIL_MAP");
        // Raster by default.  The band-per-XCD mapping (2) wins when the same GOP is re-encoded
        // and its frames sit in the Infinity Cache (k_fdct 0.30 vs 0.33 ms); on a stream of GOPs
        // (inputs from HBM) raster wins, 0.41-0.43 vs 0.45-0.48 ms, three runs.
        const int v = e ? atoi(e) : 0;
        return v < 0 || v > 2 ? 0 : v;
    }();
    return m;
}

template <int MODE>
int launch_fdct(vcs_ctx *c, int tslot, const uint8_t *imgs, size_t stride, const uint8_t *ref,
                const int16_t *mv, int n, void *planes, size_t pstride_elems) {
    if (n == 0) return VCS_OK;
    const Geom &g = c->g;
    const int map = tail_map();
    const dim3 grid = tail_grid_dim(g, n, map, kFdctWaves), block(64 * kFdctWaves);
    EvPair ev;
    int rc = t_begin(c, tslot, &ev, /*attach=*/true);
    if (rc) return rc;
#define VCS_FDCT(CM, FD, T)                                                                      \
    hipExtLaunchKernelGGL((k_fdct<MODE, CM, FD>), grid, block, 0, c->stream, ev.a, ev.b, 0, imgs, \
                          stride, ref, mv, g, c->d_DQ, (T *)planes, pstride_elems, n, map)
    switch (c->coeff_mode) {
        case VCS_COEFF_F64:
            if (c->fastdiv) VCS_FDCT(0, true, double); else VCS_FDCT(0, false, double);
            break;
        case VCS_COEFF_F32:
            if (c->fastdiv) VCS_FDCT(1, true, float); else VCS_FDCT(1, false, float);
            break;
        default:
            if (c->fastdiv) VCS_FDCT(2, true, int16_t); else VCS_FDCT(2, false, int16_t);
            break;
    }
#undef VCS_FDCT
    return t_end(c, &ev);
}

template <int MODE>
int launch_idct(vcs_ctx *c, int tslot, const void *planes, size_t pstride_elems,
                const uint8_t *ref, const int16_t *mv, int n, uint8_t *outs, size_t ostride) {
    if (n == 0) return VCS_OK;
    const Geom &g = c->g;
    // k_idct addresses a frame's planes with 32-bit byte offsets (buffer loads)
    if ((unsigned long long)3 * g.H * g.W * 8 > 0xFFFFFFF0ull)
        return fail(c, VCS_E_ARG, "%dx%d: the decode path takes frames up to 178 Mpixel", g.W, g.H);
    const int map = tail_map();
    const dim3 grid = tail_grid_dim(g, n, map, kIdctWaves), block(64 * kIdctWaves);
    EvPair ev;
    int rc = t_begin(c, tslot, &ev, /*attach=*/true);
    if (rc) return rc;
    const bool one = MODE == 1 && g.bs % 8 == 0;   // a lane's 8 pixels share one motion block
#define VCS_IDCT(CM, T, ONE)                                                                        \
    hipExtLaunchKernelGGL((k_idct<MODE, CM, ONE>), grid, block, 0, c->stream, ev.a, ev.b, 0,        \
                          (const T *)planes, pstride_elems, ref, mv, g, c->d_DQ, outs, ostride, n, map)
    switch (c->coeff_mode) {
        case VCS_COEFF_F64:
            if (one) VCS_IDCT(0, double, MODE == 1); else VCS_IDCT(0, double, false);
            break;
        case VCS_COEFF_F32:
            if (one) VCS_IDCT(1, float, MODE == 1); else VCS_IDCT(1, float, false);
            break;
        default:
            if (one) VCS_IDCT(2, int16_t, MODE == 1); else VCS_IDCT(2, int16_t, false);
            break;
    }
#undef VCS_IDCT
    return t_end(c, &ev);
}

int launch_mc(vcs_ctx *c, const int16_t *mv, const int32_t *coords, int n, uint8_t *pred) {
    HIP_TRY(c, hipMemsetAsync(pred, 0, c->F, c->stream));  // motion.py:45-46
    if (n == 0) return VCS_OK;
    EvPair ev;
    int rc = t_begin(c, VCS_T_AUX, &ev);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mc, dim3(n), dim3(64), 0, c->stream, c->d_ref, mv, coords, c->g, pred);
    return t_end(c, &ev);
}

int launch_addsub(vcs_ctx *c, const uint8_t *a, const uint8_t *b, int sign, size_t n,
                  uint8_t *out) {
    if (n == 0) return VCS_OK;
    size_t nv = n >> 4;
    int blocks = (int)((nv + 255) / 256);
    blocks = blocks < 1 ? 1 : blocks > 4096 ? 4096 : blocks;
    EvPair ev;
    int rc = t_begin(c, VCS_T_AUX, &ev);
    if (rc) return rc;
    if (sign < 0)
        hipLaunchKernelGGL((k_bytes_addsub<-1>), dim3(blocks), dim3(256), 0, c->stream, a, b, n, out);
    else
        hipLaunchKernelGGL((k_bytes_addsub<1>), dim3(blocks), dim3(256), 0, c->stream, a, b, n, out);
    return t_end(c, &ev);
}

int ensure_planes(vcs_ctx *c) {
    if (c->d_planes) return VCS_OK;
    // sized for the widest mode so a later vcs_set_params(coeff_mode) is free
    HIP_TRY(c, hipMalloc(&c->d_planes, (size_t)3 * c->g.H * c->g.W * 8 + 64));
    return VCS_OK;
}

int need_dct_geometry(vcs_ctx *c) {
    if (c->g.H % 8 || c->g.W % 8)
        return fail(c, VCS_E_ARG,
                    "DCT path needs H and W to be multiples of 8 (got %dx%d): the reference "
                    "resamples such frames with cv2.resize (DCTcompressor.py:52-53), which is "
                    "not reproduced", c->g.H, c->g.W);
    return VCS_OK;
}

// Host-side bounds check of motion vectors (+ optional coords): a faulting
// kernel can reset the whole GPU, so nothing unchecked reaches k_mc / k_idct.
int check_mv(vcs_ctx *c, const int16_t *mv, const int32_t *coords, int n) {
    const Geom &g = c->g;
    for (int k = 0; k < n; k++) {
        int x = coords ? coords[2 * k] : (k % g.nbx) * g.bs;
        int y = coords ? coords[2 * k + 1] : (k / g.nbx) * g.bs;
        int sx = x + mv[2 * k], sy = y + mv[2 * k + 1];
        if (x < 0 || y < 0 || x + g.bs > g.W || y + g.bs > g.H || sx < 0 || sy < 0 ||
            sx + g.bs > g.W || sy + g.bs > g.H)
            return fail(c, VCS_E_RANGE, "block %d at (%d,%d) with mv (%d,%d) leaves the frame", k,
                        x, y, mv[2 * k], mv[2 * k + 1]);
    }
    return VCS_OK;
}

int use(vcs_ctx *c, bool join = true) {
    if (!c) return fail(nullptr, VCS_E_ARG, "null context");
    HIP_TRY(c, hipSetDevice(c->device));
    if (join && c->split_pending) {  // everything else is ordered behind the split streams' work
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_me, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_tail, 0));
        c->split_pending = false;
    }
    return VCS_OK;
}

// Two CU-masked streams: bit i of the mask is CU i / 8 of XCD i % 8 (tools/cu_mask_probe.hip),
// so the first c bits are c / 8 CUs on every XCD.
int ensure_split(vcs_ctx *c) {
    if (c->s_me) return VCS_OK;
    const int words = (c->cus + 31) / 32;
    std::vector<uint32_t> m_tail(words, 0u), m_me(words, 0u);
    const char *sh = getenv("VCS_SPLIT_SHARED");  // diagnostic: both streams on every CU
    for (int i = 0; i < c->cus; i++) {
        if (sh && sh[0] == '1') { m_tail[i / 32] |= 1u << (i % 32); m_me[i / 32] |= 1u << (i % 32); }
        else (i < c->split_cus ? m_tail : m_me)[i / 32] |= 1u << (i % 32);
    }
    HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->s_me, (uint32_t)words, m_me.data()));
    HIP_TRY(c, hipExtStreamCreateWithCUMask(&c->s_tail, (uint32_t)words, m_tail.data()));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_me, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    return VCS_OK;
}

void free_gop(vcs_ctx *c) {
    hipFree(c->g_cur); hipFree(c->g_dec); hipFree(c->g_mv); hipFree(c->g_ns); hipFree(c->g_planes);
    c->g_cur = c->g_dec = nullptr; c->g_mv = nullptr; c->g_ns = nullptr; c->g_planes = nullptr;
    c->slots = 0;
}

void free_async(vcs_ctx *c) {
    for (auto &s : c->aslots) {
        hipFree(s.d_cur); hipFree(s.d_mv);
        if (s.h_mv) hipHostFree(s.h_mv);
        if (s.ev_in) hipEventDestroy(s.ev_in);
        if (s.ev_kern) hipEventDestroy(s.ev_kern);
        if (s.ev_out) hipEventDestroy(s.ev_out);
    }
    c->aslots.clear();
    if (c->s_in) hipStreamDestroy(c->s_in);
    if (c->s_out) hipStreamDestroy(c->s_out);
    c->s_in = c->s_out = nullptr;
}

}  // namespace

// ---- HBM-resident coefficient planes: handles carved out of slabs -----------------
// One hipMalloc per frame (49.8 MB at 1080p) costs more than the frame's kernels and
// synchronises the device; handles are therefore carved out of slabs of up to 16 frames.
// A slab is freed when its last handle is (handles may outlive their context).
struct PlaneSlab {
    void *base = nullptr;
    size_t each = 0;
    int cap = 0, used = 0, live = 0;
    bool current = false;  // some context still carves from it
};
struct vcs_planes {
    void *d = nullptr;
    size_t bytes = 0;
    int coeff_mode = 0;
    int H = 0, W = 0;
    PlaneSlab *slab = nullptr;
};

namespace {

void slab_release(PlaneSlab *sl) {
    if (!sl || sl->live > 0) return;
    if (sl->current) { sl->used = 0; return; }  // empty and still current: start over
    hipFree(sl->base);
    delete sl;
}

int planes_alloc(vcs_ctx *c, vcs_planes **out) {
    const size_t bytes = planes_bytes(c), each = round_up(bytes + 64, 256);
    PlaneSlab *sl = c->slab;
    if (sl && (sl->each != each || sl->used == sl->cap)) {
        sl->current = false;
        slab_release(sl);
        sl = c->slab = nullptr;
    }
    if (!sl) {
        sl = new (std::nothrow) PlaneSlab();
        if (!sl) return fail(c, VCS_E_NOMEM, "out of host memory");
        const size_t want = ((size_t)1 << 30) / each;
        sl->cap = (int)(want < 1 ? 1 : want > 16 ? 16 : want);
        sl->each = each;
        hipError_t e = hipMalloc(&sl->base, each * sl->cap);
        if (e != hipSuccess && sl->cap > 1) {  // memory is tight: one frame at a time
            sl->cap = 1;
            e = hipMalloc(&sl->base, each);
        }
        if (e != hipSuccess) {
            delete sl;
            return fail(c, VCS_E_NOMEM, "hipMalloc(planes): %s", hipGetErrorString(e));
        }
        sl->current = true;
        c->slab = sl;
    }
    vcs_planes *h = new (std::nothrow) vcs_planes();
    if (!h) return fail(c, VCS_E_NOMEM, "out of host memory");
    h->d = (char *)sl->base + (size_t)sl->used * each;
    h->bytes = bytes;
    h->coeff_mode = c->coeff_mode;
    h->H = c->g.H;
    h->W = c->g.W;
    h->slab = sl;
    sl->used++;
    sl->live++;
    *out = h;
    return VCS_OK;
}

}  // namespace

// ==========================================================================
extern "C" {

int vcs_group_destroy(vcs_group *g);
int vcs_store_free(vcs_ctx *c, vcs_planes *h);

int vcs_abi_version(void) { return VCS_ABI_VERSION; }

#ifndef VCS_BUILD_ID
#define VCS_BUILD_ID "unknown"
#endif
const char *vcs_build_id(void) { return VCS_BUILD_ID; }

int vcs_experiment_build(void) {
#ifdef VCS_EXPERIMENT_BUILD
    return 1;
#else
    return 0;
#endif
}

int vcs_host_alloc(size_t bytes, void **ptr) {
    if (!ptr || bytes == 0) return fail(nullptr, VCS_E_ARG, "null pointer or zero size");
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        *ptr = nullptr;
        return fail(nullptr, e == hipErrorOutOfMemory ? VCS_E_NOMEM : VCS_E_HIP, "hipHostMalloc(%zu): %s",
                    bytes, hipGetErrorString(e));
    }
    return VCS_OK;
}

int vcs_host_free(void *ptr) {
    if (!ptr) return VCS_OK;
    hipError_t e = hipHostFree(ptr);
    if (e != hipSuccess) return fail(nullptr, VCS_E_HIP, "hipHostFree: %s", hipGetErrorString(e));
    return VCS_OK;
}

int vcs_device_count(int *n) {
    if (!n) return fail(nullptr, VCS_E_ARG, "null pointer");
    int k = 0;
    hipError_t e = hipGetDeviceCount(&k);
    if (e != hipSuccess) {
        *n = 0;
        return fail(nullptr, VCS_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *n = k;
    return VCS_OK;
}

const char *vcs_last_error(const vcs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int vcs_dct_matrix(double *D) {
    if (!D) return fail(nullptr, VCS_E_ARG, "null pointer");
    memcpy(D, kDct8Host, sizeof(kDct8Host));  // DCTcompressor.py:124-133 (dct_kernels.hpp)
    return VCS_OK;
}

int vcs_make_q(double qf, double *Q) {
    if (!Q) return fail(nullptr, VCS_E_ARG, "null pointer");
    double scale;
    if (qf < 50 && qf > 1) scale = 50 / qf;      // DCTcompressor.py:30-31
    else if (qf < 100) scale = (100 - qf) / 50;  // :32-33
    else return fail(nullptr, VCS_E_ARG, "Invalid quality setting, must be between 1 and 99.");
    for (int i = 0; i < 64; i++) {
        double y = nearbyint(kQY[i] * scale), cc = nearbyint(kQC[i] * scale);  // np.round
        y = y < 1 ? 1 : y > 255 ? 255 : y;                                     // np.clip
        cc = cc < 1 ? 1 : cc > 255 ? 255 : cc;
        Q[i] = y;
        Q[64 + i] = cc;
        Q[128 + i] = cc;
    }
    return VCS_OK;
}

int vcs_ctx_create(int device, int H, int W, int me_bs, int dct_bs, vcs_ctx **out) {
    if (!out) return fail(nullptr, VCS_E_ARG, "null out pointer");
    *out = nullptr;
    if (H < 1 || W < 1 || H > 16384 || W > 16384)
        return fail(nullptr, VCS_E_ARG, "bad frame size %dx%d", H, W);
    if (me_bs < 2 || me_bs > 128)
        return fail(nullptr, VCS_E_ARG, "motion block size %d unsupported (2..128); note "
                    "round(1/3) == 0 makes the reference raise for block size 1", me_bs);
    if (dct_bs != 8)
        return fail(nullptr, VCS_E_ARG, "DCT block size must be 8: the reference's Q tables are "
                    "8x8 and np.true_divide raises for any other size (DCTcompressor.py:71)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return fail(nullptr, VCS_E_HIP, "no usable HIP device (%s); libvcs_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= ndev)
        return fail(nullptr, VCS_E_ARG, "device %d out of range (have %d)", device, ndev);
    vcs_ctx *c = new (std::nothrow) vcs_ctx();
    if (!c) return fail(nullptr, VCS_E_NOMEM, "out of host memory");
    c->device = device;
    Geom &g = c->g;
    g.H = H; g.W = W; g.pitch = 3 * W; g.bs = me_bs;
    g.nbx = W / me_bs; g.nby = H / me_bs; g.N = g.nbx * g.nby;
    g.bs_magic = magic_u32((uint32_t)me_bs);
    g.gx_magic = magic_u32((uint32_t)((W + 63) >> 6));
    {   // the tail kernels divide strip < groups_x * H / 8 by groups_x with one multiply-high (div_magic): exact
        // while the product stays below 2^32 (W = 65,536 with H = 32,768 is the first frame that does not)
        const unsigned long long gx = (unsigned long long)((W + 63) >> 6);
        if (gx * gx * (unsigned long long)(H >> 3) >= (1ull << 32)) {
            delete c;
            return fail(nullptr, VCS_E_ARG, "frame %d x %d is too large for the tail kernels' strip indexing", W, H);
        }
    }
    g.win = 2 * me_bs;               // motion.py:18
    g.step = py_round_div3(me_bs);   // motion.py:132
    c->thresh64 = 2000;              // motion.py:8
    c->F = (size_t)3 * H * W;
    c->Fs = round_up(c->F + 64, 256);  // slack: kernels may over-read a few bytes
    int rc = refresh_me_geometry(c);
    if (rc) { g_err = c->err; delete c; return rc; }
#define CREATE_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e2 = (expr);                                                            \
        if (e2 != hipSuccess) {                                                            \
            fail(nullptr, e2 == hipErrorOutOfMemory ? VCS_E_NOMEM : VCS_E_HIP, "%s failed: %s", \
                 #expr, hipGetErrorString(e2));                                            \
            vcs_ctx_destroy(c);                                                            \
            return e2 == hipErrorOutOfMemory ? VCS_E_NOMEM : VCS_E_HIP;                    \
        }                                                                                  \
    } while (0)
    CREATE_TRY(hipSetDevice(device));
    CREATE_TRY(hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device));
    if (c->cus < 1) c->cus = 1;
    {
        const char *e = getenv("VCS_SPLIT_CUS");
        const int v = e ? atoi(e) : 0;
        c->split_cus = (v >= 8 && v <= c->cus - 8) ? v / 8 * 8 : 0;
    }
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&c->xev, hipEventDisableTiming));
    // the reference carries slack rows: k_me16 fetches its window 64 rows deep
    // without clamping at the bottom edge (the extra rows are never used)
    CREATE_TRY(hipMalloc(&c->d_ref, c->Fs + me_fast_ref_slack(g)));
    CREATE_TRY(hipMemset(c->d_ref, 0, c->Fs + me_fast_ref_slack(g)));
    CREATE_TRY(hipMalloc(&c->d_sums, me_fast_sums_bytes(g)));
    CREATE_TRY(hipMalloc(&c->d_clk, 128));
    CREATE_TRY(hipMemset(c->d_clk, 0, 128));
    CREATE_TRY(hipMalloc(&c->d_cur, c->Fs));
    CREATE_TRY(hipMalloc(&c->d_a, c->Fs));
    CREATE_TRY(hipMalloc(&c->d_b, c->Fs));
    CREATE_TRY(hipMemset(c->d_ref, 0, c->Fs));
    CREATE_TRY(hipMemset(c->d_cur, 0, c->Fs));
    size_t nmv = (size_t)(g.N > 0 ? g.N : 1);
    CREATE_TRY(hipMalloc(&c->d_mv, nmv * 2 * sizeof(int16_t)));
    CREATE_TRY(hipMalloc(&c->d_coords, nmv * 2 * sizeof(int32_t)));
    CREATE_TRY(hipMalloc(&c->d_cost, nmv * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(&c->d_ns, sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(&c->d_DQ, sizeof(double) * 448));
#undef CREATE_TRY
    vcs_dct_matrix(c->hDQ);
    vcs_make_q(50.0, c->hDQ + 64);  // DCTcompressor.py:29
    refresh_q(c);
    hipError_t e3 = hipMemcpy(c->d_DQ, c->hDQ, sizeof(double) * 448, hipMemcpyHostToDevice);
    if (e3 != hipSuccess) {
        fail(nullptr, VCS_E_HIP, "hipMemcpy(DQ): %s", hipGetErrorString(e3));
        vcs_ctx_destroy(c);
        return VCS_E_HIP;
    }
    *out = c;
    return VCS_OK;
}

int vcs_ctx_destroy(vcs_ctx *c) {
    if (!c) return VCS_OK;
    if (c->group) vcs_group_destroy(c->group);  // detaches every member context
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->s_out) hipStreamSynchronize(c->s_out);
    free_async(c);
    if (c->s_me) { hipStreamSynchronize(c->s_me); hipStreamDestroy(c->s_me); }
    if (c->s_tail) { hipStreamSynchronize(c->s_tail); hipStreamDestroy(c->s_tail); }
    if (c->ev_me_done) hipEventDestroy(c->ev_me_done);
    if (c->ev_main) hipEventDestroy(c->ev_main);
    if (c->ev_me) hipEventDestroy(c->ev_me);
    if (c->ev_tail) hipEventDestroy(c->ev_tail);
    if (c->slab) {
        c->slab->current = false;
        slab_release(c->slab);
        c->slab = nullptr;
    }
    for (auto &p : c->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto &p : c->pool) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    free_gop(c);
    hipFree(c->d_ref); hipFree(c->d_ref_alt); hipFree(c->d_cur); hipFree(c->d_a); hipFree(c->d_b); hipFree(c->d_mv);
    hipFree(c->d_coords); hipFree(c->d_cost); hipFree(c->d_ns); hipFree(c->d_planes);
    hipFree(c->d_DQ);
    hipFree(c->d_sums);
    for (auto &t : c->tabs) { hipFree(t.tab); if (t.ev_done) hipEventDestroy(t.ev_done); }
    hipFree(c->d_cpl); hipFree(c->d_binfo);
    if (c->ev_side_go) hipEventDestroy(c->ev_side_go);
    if (c->ev_next) hipEventDestroy(c->ev_next);
    if (c->s_side) hipStreamDestroy(c->s_side);
    hipFree(c->d_clk);
    hipFree(c->d_intra);
    hipFree(c->ib_in); hipFree(c->ib_out);
    if (c->xev) hipEventDestroy(c->xev);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return VCS_OK;
}

int vcs_set_params(vcs_ctx *c, int win, int step, int64_t sim_thresh, const double *Q,
                   int coeff_mode) {
    int rc = use(c);
    if (rc) return rc;
    if (coeff_mode == VCS_KEEP) coeff_mode = c->coeff_mode;
    if (coeff_mode < VCS_COEFF_F64 || coeff_mode > VCS_COEFF_I16)
        return fail(c, VCS_E_ARG, "bad coeff_mode %d", coeff_mode);
    if (Q)
        for (int i = 0; i < 192; i++)
            if (!(Q[i] == Q[i]) || Q[i] == 0.0 || fabs(Q[i]) > 1e300)
                return fail(c, VCS_E_ARG, "Q[%d] must be finite and non-zero", i);
    Geom saved = c->g;
    int64_t saved_t = c->thresh64;
    if (win > 0) c->g.win = win;
    if (step > 0) c->g.step = step;
    if (sim_thresh != VCS_KEEP_THRESH) c->thresh64 = sim_thresh;
    rc = refresh_me_geometry(c);
    if (rc) {
        c->g = saved;
        c->thresh64 = saved_t;
        refresh_me_geometry(c);
        return rc;
    }
    if (coeff_mode != c->coeff_mode && c->slots)
        return fail(c, VCS_E_STATE, "coeff_mode cannot change while GOP slots are reserved");
    c->coeff_mode = coeff_mode;
    if (Q) {
        memcpy(c->hDQ + 64, Q, sizeof(double) * 192);
        refresh_q(c);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(c->d_DQ, c->hDQ, sizeof(double) * 448, hipMemcpyHostToDevice));
    }
    return VCS_OK;
}

int vcs_set_search(vcs_ctx *c, int search, int cost) {
    int rc = use(c);
    if (rc) return rc;
    if (search == VCS_KEEP) search = c->search;
    if (cost == VCS_KEEP) cost = c->cost;
    if (search < VCS_SEARCH_GRID || search > VCS_SEARCH_TSS || cost < VCS_COST_WRAP || cost > VCS_COST_SAD)
        return fail(c, VCS_E_ARG, "bad search / cost selector (%d, %d)", search, cost);
    const int s0 = c->search, c0 = c->cost;
    c->search = search;
    c->cost = cost;
    rc = refresh_me_geometry(c);
    if (rc) {
        c->search = s0;
        c->cost = c0;
        refresh_me_geometry(c);
    }
    return rc;
}

int vcs_set_ref(vcs_ctx *c, const uint8_t *ref) {
    int rc = use(c);
    if (rc) return rc;
    if (!ref) return fail(c, VCS_E_ARG, "null reference frame");
    if ((rc = bits_tables_stale(c, c->d_ref))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_ref, ref, c->F, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ref_valid = true;
    c->sums_valid = false;
    return VCS_OK;
}

int vcs_me(vcs_ctx *c, const uint8_t *cur, int16_t *mv_out, uint32_t *n_static,
           uint64_t *cost_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!cur || !mv_out) return fail(c, VCS_E_ARG, "null pointer");
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    HIP_TRY(c, hipMemcpyAsync(c->d_cur, cur, c->F, hipMemcpyHostToDevice, c->stream));
    rc = launch_me(c, c->d_ref, c->d_cur, c->Fs, 1, c->d_mv, c->d_ns, cost_out ? c->d_cost : nullptr);
    if (rc) return rc;
    size_t N = c->g.N;
    if (N) HIP_TRY(c, hipMemcpyAsync(mv_out, c->d_mv, N * 4, hipMemcpyDeviceToHost, c->stream));
    uint32_t ns = 0;
    if (N) HIP_TRY(c, hipMemcpyAsync(&ns, c->d_ns, 4, hipMemcpyDeviceToHost, c->stream));
    if (cost_out && N)
        HIP_TRY(c, hipMemcpyAsync(cost_out, c->d_cost, N * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_static) *n_static = ns;
    return VCS_OK;
}

int vcs_mc(vcs_ctx *c, const int16_t *mv, const int32_t *coords, int n, uint8_t *pred_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!pred_out || (n > 0 && !mv)) return fail(c, VCS_E_ARG, "null pointer");
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    if (n < 0 || (!coords && n != c->g.N) || (coords && n > c->g.N))
        return fail(c, VCS_E_ARG, "n = %d does not match the block grid (N = %d)", n, c->g.N);
    rc = check_mv(c, mv, coords, n);
    if (rc) return rc;
    if (n) HIP_TRY(c, hipMemcpyAsync(c->d_mv, mv, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    if (coords && n)
        HIP_TRY(c, hipMemcpyAsync(c->d_coords, coords, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    rc = launch_mc(c, c->d_mv, coords ? c->d_coords : nullptr, n, c->d_a);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(pred_out, c->d_a, c->F, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_residual(vcs_ctx *c, const uint8_t *a, const uint8_t *b, int sign, size_t n,
                 uint8_t *out) {
    int rc = use(c);
    if (rc) return rc;
    if (!a || !b || !out) return fail(c, VCS_E_ARG, "null pointer");
    if (n > c->F) return fail(c, VCS_E_ARG, "n = %zu exceeds the frame size %zu", n, c->F);
    HIP_TRY(c, hipMemcpyAsync(c->d_a, a, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_b, b, n, hipMemcpyHostToDevice, c->stream));
    rc = launch_addsub(c, c->d_a, c->d_b, sign, n, c->d_cur);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->d_cur, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_compress(vcs_ctx *c, const uint8_t *bgr, void *planes_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!bgr || !planes_out) return fail(c, VCS_E_ARG, "null pointer");
    if ((rc = need_dct_geometry(c)) || (rc = ensure_planes(c))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_a, bgr, c->F, hipMemcpyHostToDevice, c->stream));
    rc = launch_fdct<0>(c, VCS_T_AUX, c->d_a, c->Fs, nullptr, nullptr, 1, c->d_planes, 0);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(planes_out, c->d_planes, planes_bytes(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_decompress(vcs_ctx *c, const void *planes_in, uint8_t *bgr_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!planes_in || !bgr_out) return fail(c, VCS_E_ARG, "null pointer");
    if ((rc = need_dct_geometry(c)) || (rc = ensure_planes(c))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_planes, planes_in, planes_bytes(c), hipMemcpyHostToDevice, c->stream));
    rc = launch_idct<0>(c, VCS_T_AUX, c->d_planes, 0, nullptr, nullptr, 1, c->d_a, c->Fs);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(bgr_out, c->d_a, c->F, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_encode_p(vcs_ctx *c, const uint8_t *cur, int16_t *mv_out, void *planes_out,
                 uint8_t *res_out, uint32_t *n_static) {
    int rc = use(c);
    if (rc) return rc;
    if (!cur || !mv_out) return fail(c, VCS_E_ARG, "null pointer");
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    if (planes_out && ((rc = need_dct_geometry(c)) || (rc = ensure_planes(c)))) return rc;
    size_t N = c->g.N;
    HIP_TRY(c, hipMemcpyAsync(c->d_cur, cur, c->F, hipMemcpyHostToDevice, c->stream));
    rc = launch_me(c, c->d_ref, c->d_cur, c->Fs, 1, c->d_mv, c->d_ns, nullptr);
    if (rc) return rc;
    if (planes_out) {
        rc = N ? launch_fdct<1>(c, VCS_T_ENC_TAIL, c->d_cur, c->Fs, c->d_ref, c->d_mv, 1, c->d_planes, 0)
               : launch_fdct<0>(c, VCS_T_ENC_TAIL, c->d_cur, c->Fs, nullptr, nullptr, 1, c->d_planes, 0);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(planes_out, c->d_planes, planes_bytes(c), hipMemcpyDeviceToHost, c->stream));
    }
    if (res_out) {  // with_DCT = False (encoder.py:64-66): the raw uint8 residual
        rc = launch_mc(c, c->d_mv, nullptr, (int)N, c->d_a);
        if (!rc) rc = launch_addsub(c, c->d_cur, c->d_a, -1, c->F, c->d_b);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpyAsync(res_out, c->d_b, c->F, hipMemcpyDeviceToHost, c->stream));
    }
    uint32_t ns = 0;
    if (N) {
        HIP_TRY(c, hipMemcpyAsync(mv_out, c->d_mv, N * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&ns, c->d_ns, 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (n_static) *n_static = ns;
    return VCS_OK;
}

int vcs_decode_p(vcs_ctx *c, const int16_t *mv, const void *planes, uint8_t *bgr_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!bgr_out || (c->g.N > 0 && !mv)) return fail(c, VCS_E_ARG, "null pointer");
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    rc = check_mv(c, mv, nullptr, c->g.N);
    if (rc) return rc;
    size_t N = c->g.N;
    if (N) HIP_TRY(c, hipMemcpyAsync(c->d_mv, mv, N * 4, hipMemcpyHostToDevice, c->stream));
    if (planes) {
        if ((rc = need_dct_geometry(c)) || (rc = ensure_planes(c))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->d_planes, planes, planes_bytes(c), hipMemcpyHostToDevice, c->stream));
        rc = N ? launch_idct<1>(c, VCS_T_DEC, c->d_planes, 0, c->d_ref, c->d_mv, 1, c->d_a, c->Fs)
               : launch_idct<0>(c, VCS_T_DEC, c->d_planes, 0, nullptr, nullptr, 1, c->d_a, c->Fs);
    } else {
        rc = launch_mc(c, c->d_mv, nullptr, (int)N, c->d_a);
    }
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(bgr_out, c->d_a, c->F, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

// ---- HBM-resident coefficient planes (next row N3) ----------------------------
// The float64 planes of a P frame are 8x the frame (49.8 MB at 1080p) and
// dominate the PCIe time of the single-frame API.  These entry points leave them
// in HBM behind an opaque handle: the encoder returns only the motion vectors,
// the decoder consumes the handle directly, and the planes are downloaded only
// if somebody actually looks at them.
int vcs_store_encode_p(vcs_ctx *c, const uint8_t *cur, int16_t *mv_out, uint32_t *n_static,
                       vcs_planes **handle_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!cur || !mv_out || !handle_out) return fail(c, VCS_E_ARG, "null pointer");
    *handle_out = nullptr;
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    if ((rc = need_dct_geometry(c))) return rc;
    vcs_planes *h = nullptr;
    if ((rc = planes_alloc(c, &h))) return rc;
    const size_t N = c->g.N;
    rc = VCS_OK;
    hipError_t e2 = hipMemcpyAsync(c->d_cur, cur, c->F, hipMemcpyHostToDevice, c->stream);
    if (e2 != hipSuccess) rc = fail(c, VCS_E_HIP, "upload: %s", hipGetErrorString(e2));
    if (!rc) rc = launch_me(c, c->d_ref, c->d_cur, c->Fs, 1, c->d_mv, c->d_ns, nullptr);
    if (!rc)
        rc = N ? launch_fdct<1>(c, VCS_T_ENC_TAIL, c->d_cur, c->Fs, c->d_ref, c->d_mv, 1, h->d, 0)
               : launch_fdct<0>(c, VCS_T_ENC_TAIL, c->d_cur, c->Fs, nullptr, nullptr, 1, h->d, 0);
    uint32_t ns = 0;
    if (!rc && N) {
        e2 = hipMemcpyAsync(mv_out, c->d_mv, N * 4, hipMemcpyDeviceToHost, c->stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(&ns, c->d_ns, 4, hipMemcpyDeviceToHost, c->stream);
        if (e2 != hipSuccess) rc = fail(c, VCS_E_HIP, "download: %s", hipGetErrorString(e2));
    }
    e2 = hipStreamSynchronize(c->stream);
    if (!rc && e2 != hipSuccess) rc = fail(c, VCS_E_HIP, "sync: %s", hipGetErrorString(e2));
    if (rc) {
        vcs_store_free(c, h);
        return rc;
    }
    if (n_static) *n_static = ns;
    *handle_out = h;
    return VCS_OK;
}

static int check_handle(vcs_ctx *c, const vcs_planes *h) {
    if (!h || !h->d) return fail(c, VCS_E_ARG, "null planes handle");
    if (h->H != c->g.H || h->W != c->g.W || h->coeff_mode != c->coeff_mode)
        return fail(c, VCS_E_ARG, "planes handle does not match this context (size / coeff_mode)");
    return VCS_OK;
}

int vcs_store_planes(vcs_ctx *c, const vcs_planes *h, void *planes_out) {
    int rc = use(c);
    if (rc || (rc = check_handle(c, h))) return rc;
    if (!planes_out) return fail(c, VCS_E_ARG, "null pointer");
    HIP_TRY(c, hipMemcpyAsync(planes_out, h->d, h->bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_store_decode_p(vcs_ctx *c, const int16_t *mv, const vcs_planes *h, uint8_t *bgr_out) {
    int rc = use(c);
    if (rc || (rc = check_handle(c, h))) return rc;
    if (!bgr_out || (c->g.N > 0 && !mv)) return fail(c, VCS_E_ARG, "null pointer");
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    if ((rc = check_mv(c, mv, nullptr, c->g.N))) return rc;
    const size_t N = c->g.N;
    if (N) HIP_TRY(c, hipMemcpyAsync(c->d_mv, mv, N * 4, hipMemcpyHostToDevice, c->stream));
    rc = N ? launch_idct<1>(c, VCS_T_DEC, h->d, 0, c->d_ref, c->d_mv, 1, c->d_a, c->Fs)
           : launch_idct<0>(c, VCS_T_DEC, h->d, 0, nullptr, nullptr, 1, c->d_a, c->Fs);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(bgr_out, c->d_a, c->F, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_store_free(vcs_ctx *c, vcs_planes *h) {
    if (!h) return VCS_OK;
    if (c) {
        hipSetDevice(c->device);
        if (c->stream) hipStreamSynchronize(c->stream);
    }
    PlaneSlab *sl = h->slab;
    delete h;
    if (sl) {
        sl->live--;
        slab_release(sl);
    }
    return VCS_OK;
}

// ---- GOP path ------------------------------------------------------------
int vcs_gop_reserve(vcs_ctx *c, int n_slots, int with_decode) {
    int rc = use(c);
    if (rc) return rc;
    if (n_slots < 1 || n_slots > 4096) return fail(c, VCS_E_ARG, "bad slot count %d", n_slots);
    if ((rc = need_dct_geometry(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_gop(c);
    size_t nmv = (size_t)(c->g.N > 0 ? c->g.N : 1);
    hipError_t e = hipMalloc(&c->g_cur, c->Fs * n_slots);
    if (e == hipSuccess) e = hipMalloc(&c->g_mv, nmv * 4 * n_slots);
    if (e == hipSuccess) e = hipMalloc(&c->g_ns, sizeof(uint32_t) * n_slots);
    if (e == hipSuccess) e = hipMalloc(&c->g_planes, planes_bytes(c) * n_slots + 64);
    if (e == hipSuccess && with_decode) e = hipMalloc(&c->g_dec, c->Fs * n_slots);
    if (e == hipSuccess) e = hipMemset(c->g_cur, 0, c->Fs * n_slots);
    if (e == hipSuccess) e = hipMemset(c->g_mv, 0, nmv * 4 * n_slots);
    if (e == hipSuccess) e = hipMemset(c->g_ns, 0, sizeof(uint32_t) * n_slots);
    if (e != hipSuccess) {
        free_gop(c);
        return fail(c, e == hipErrorOutOfMemory ? VCS_E_NOMEM : VCS_E_HIP, "GOP reserve: %s",
                    hipGetErrorString(e));
    }
    c->slots = n_slots;
    c->gop_dec = with_decode != 0;
    return VCS_OK;
}

static int slot_range(vcs_ctx *c, int first, int n) {
    if (first < 0 || n < 0 || first + n > c->slots)
        return fail(c, VCS_E_ARG, "slots [%d, %d) outside the %d reserved", first, first + n, c->slots);
    return VCS_OK;
}

int vcs_gop_upload(vcs_ctx *c, int first, int n, const uint8_t *frames) {
    int rc = use(c);
    if (rc || (rc = slot_range(c, first, n))) return rc;
    if (!frames) return fail(c, VCS_E_ARG, "null pointer");
    // host frames are packed at F, HBM slots sit at the padded stride Fs
    HIP_TRY(c, hipMemcpy2DAsync(c->g_cur + (size_t)first * c->Fs, c->Fs, frames, c->F, c->F, n,
                                hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

// The tail kernel reads the GOP's frames 0.6 ms after the search kernel did, and finds none
// of them in the 256 MB Infinity Cache: the search's scattered 48-byte row pieces do not
// leave the lines there, a stream of full-line reads does.  k_l3_prefetch is that stream, run
// between the two kernels: 0.04 ms of reading buys 0.09 ms of tail (0.42 -> 0.33 ms; HBM then
// serves the tail's 1.44 GB of plane writes without interleaved reads).  Running it BESIDE the
// search instead (own stream; own slice per search workgroup on its way out) was measured too:
// the search, which lives on short dependent loads, stretches by 0.04-0.1 ms.  DESIGN.md 5.
__global__ __launch_bounds__(256) void k_l3_prefetch(const uint4 *__restrict__ p, size_t n16,
                                                      uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const uint4 v = p[i];
        acc |= v.x & v.y & v.z & v.w;
    }
    if (acc == 0xFFFFFFFEu) *sink = acc;  // keeps the loads; 16 bytes of 0xFF..FE in a row do not occur... and would be harmless
}

inline bool l3_prefetch_in_search() {   // VCS_L3_PREFETCH=2: always the separate pass
    static const bool v = [] {
        const char *e = getenv("VCS_L3_PREFETCH");
        return !(e && e[0] == '2');
    }();
    return v;
}

// frames the 256 MB cache holds beside the reference (VCS_L3_PREFETCH_MAX_KB: tests force sub-batches)
inline size_t l3_prefetch_max_bytes() {
    static const size_t v = [] {
        const char *e = getenv("VCS_L3_PREFETCH_MAX_KB");
        const long long kb = e ? atoll(e) : 0;
        return kb > 0 ? (size_t)kb << 10 : (size_t)208 << 20;
    }();
    return v;
}

inline bool l3_prefetch_on() {
    static const bool v = [] {
        const char *e = getenv("VCS_L3_PREFETCH");
        return !(e && e[0] == '0');
    }();
    return v;
}

int vcs_gop_encode(vcs_ctx *c, int first, int n) {
    const bool split = c && c->split_cus > 0 && c->g.N > 0 && n > 0;
    int rc = use(c, /*join=*/!split);
    if (rc || (rc = slot_range(c, first, n))) return rc;
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    size_t N = c->g.N;
    const uint8_t *curs = c->g_cur + (size_t)first * c->Fs;
    int16_t *mv = c->g_mv + (size_t)first * N * 2;
    size_t pe = (size_t)3 * c->g.H * c->g.W;
    void *planes = (char *)c->g_planes + (size_t)first * planes_bytes(c);
    if (split) {
        // search on the big CU partition, tail on the small one; the NEXT call's search may
        // start while this call's tail is still running (different slots) -- see vcs_ctx
        if ((rc = ensure_split(c))) return rc;
        hipStream_t main = c->stream;
        const int cus = c->cus;
        HIP_TRY(c, hipEventRecord(c->ev_main, main));          // uploads, set_ref, ... so far
        HIP_TRY(c, hipStreamWaitEvent(c->s_me, c->ev_main, 0));
        const bool overlap = c->split_pending && first < c->tail_first + c->tail_n && c->tail_first < first + n;
        if (overlap) HIP_TRY(c, hipStreamWaitEvent(c->s_me, c->ev_tail, 0));  // same slots: in sequence
        c->stream = c->s_me;
        { const char *sh = getenv("VCS_SPLIT_SHARED"); c->cus = (sh && sh[0] == '1') ? cus : cus - c->split_cus; }
        rc = launch_me(c, c->d_ref, curs, c->Fs, n, mv, c->g_ns + first, nullptr);
        c->cus = cus;
        if (!rc) {
            hipEventRecord(c->ev_me, c->s_me);
            hipStreamWaitEvent(c->s_tail, c->ev_me, 0);
            c->stream = c->s_tail;
            rc = launch_fdct<1>(c, VCS_T_ENC_TAIL, curs, c->Fs, c->d_ref, mv, n, planes, pe);
            hipEventRecord(c->ev_tail, c->s_tail);
        }
        c->stream = main;
        c->split_pending = true;
        c->tail_first = first;
        c->tail_n = n;
        return rc;
    }
    // Infinity Cache prefetch (k_l3_prefetch above): for GOP-sized batches (a single frame was
    // uploaded a moment ago).  A GOP whose frames do not fit beside the reference is encoded in
    // sub-batches that do (search, prefetch, tail per sub-batch; 8 frames at 4K) -- otherwise what
    // is read first would be gone again when the tail gets there.
    size_t chunk = (size_t)n;
    const bool l3_on = c->l3_mode < 0 ? l3_prefetch_on() : c->l3_mode != 0;
    const bool l3_in_search = c->l3_mode < 0 ? l3_prefetch_in_search() : c->l3_mode == 1;
    const bool pf = l3_on && n >= 2 && N > 0;
    if (pf && (size_t)n * c->Fs > l3_prefetch_max_bytes()) {
        chunk = l3_prefetch_max_bytes() / c->Fs;
        if (chunk >= 4) chunk &= ~(size_t)3;          // whole 4-frame groups of the search kernel
        if (chunk < 2) chunk = (size_t)n;             // frames too large to hold two: one batch, no prefetch
    }
    const bool pf_on = pf && chunk * c->Fs <= l3_prefetch_max_bytes();
    // The bit-plane search wants the whole batch (its per-launch tables are repaid by many frames):
    // when it takes it, only the tail goes in cache-sized sub-batches (with the separate prefetch pass)
    const bool whole_me = chunk < (size_t)n && ((c->me_fast_ok && me_bits_worthwhile(c->g, n)) ||
                                                 (c->me_fast8_ok && me_bits8_worthwhile(c->g, n)));
    if (whole_me) {
        c->pf_in_search = false;
        if ((rc = launch_me(c, c->d_ref, curs, c->Fs, n, mv, c->g_ns + first, nullptr))) return rc;
        if (c->ev_me_done) {   // the group's collectives start behind the search, beside the tail
            HIP_TRY(c, hipEventRecord(c->ev_me_done, c->stream));
            c->me_done_rec = true;
        }
    }
    for (size_t f0 = 0; f0 < (size_t)n; f0 += chunk) {
        const int m = (int)std::min(chunk, (size_t)n - f0);
        const uint8_t *cc = curs + f0 * c->Fs;
        int16_t *mvc = mv + f0 * N * 2;
        c->t_no_count = f0 > 0;                      // timing: one "launch" per call, whatever the batching
        c->pf_in_search = !whole_me && pf_on && m >= 2 && l3_in_search;
        rc = whole_me ? VCS_OK : launch_me(c, c->d_ref, cc, c->Fs, m, mvc, c->g_ns + first + f0, nullptr);
        const bool pf_done = c->pf_in_search;
        c->pf_in_search = false;
        if (!rc && !whole_me && f0 + m == (size_t)n && c->ev_me_done) {  // the group's collectives start behind the search, beside the tail
            if (hipEventRecord(c->ev_me_done, c->stream) != hipSuccess) rc = fail(c, VCS_E_HIP, "hipEventRecord failed");
            c->me_done_rec = true;
        }
        if (!rc && pf_on && m >= 2 && !pf_done)
            hipLaunchKernelGGL(k_l3_prefetch, dim3(2048), dim3(256), 0, c->stream, (const uint4 *)cc,
                               (size_t)m * c->Fs / 16, (uint32_t *)c->d_clk + 30);
        void *pl = (char *)planes + f0 * planes_bytes(c);
        if (!rc)
            rc = N == 0  // no full motion block: the prediction is all zero (motion.py:45-46)
                     ? launch_fdct<0>(c, VCS_T_ENC_TAIL, cc, c->Fs, nullptr, nullptr, m, pl, pe)
                     : launch_fdct<1>(c, VCS_T_ENC_TAIL, cc, c->Fs, c->d_ref, mvc, m, pl, pe);
        c->t_no_count = false;
        if (rc) return rc;
    }
    return VCS_OK;
}

int vcs_gop_decode(vcs_ctx *c, int first, int n) {
    int rc = use(c);
    if (rc || (rc = slot_range(c, first, n))) return rc;
    if (!c->gop_dec) return fail(c, VCS_E_STATE, "GOP slots were reserved without decode buffers");
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    size_t N = c->g.N, pe = (size_t)3 * c->g.H * c->g.W;
    const void *planes = (const char *)c->g_planes + (size_t)first * planes_bytes(c);
    if (N == 0)
        return launch_idct<0>(c, VCS_T_DEC, planes, pe, nullptr, nullptr, n,
                              c->g_dec + (size_t)first * c->Fs, c->Fs);
    return launch_idct<1>(c, VCS_T_DEC, planes, pe, c->d_ref, c->g_mv + (size_t)first * N * 2, n,
                          c->g_dec + (size_t)first * c->Fs, c->Fs);
}

int vcs_gop_get_mv(vcs_ctx *c, int first, int n, int16_t *mv_out, uint32_t *ns_out) {
    int rc = use(c);
    if (rc || (rc = slot_range(c, first, n))) return rc;
    size_t N = c->g.N;
    if (mv_out && N && n)
        HIP_TRY(c, hipMemcpyAsync(mv_out, c->g_mv + (size_t)first * N * 2, (size_t)n * N * 4,
                                  hipMemcpyDeviceToHost, c->stream));
    if (ns_out && n)
        HIP_TRY(c, hipMemcpyAsync(ns_out, c->g_ns + first, sizeof(uint32_t) * n,
                                  hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_gop_get_planes(vcs_ctx *c, int slot, void *planes_out) {
    int rc = use(c);
    if (rc || (rc = slot_range(c, slot, 1))) return rc;
    if (!planes_out) return fail(c, VCS_E_ARG, "null pointer");
    HIP_TRY(c, hipMemcpyAsync(planes_out, (const char *)c->g_planes + (size_t)slot * planes_bytes(c),
                              planes_bytes(c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_gop_get_decoded(vcs_ctx *c, int slot, uint8_t *bgr_out) {
    int rc = use(c);
    if (rc || (rc = slot_range(c, slot, 1))) return rc;
    if (!bgr_out) return fail(c, VCS_E_ARG, "null pointer");
    if (!c->gop_dec) return fail(c, VCS_E_STATE, "GOP slots were reserved without decode buffers");
    HIP_TRY(c, hipMemcpyAsync(bgr_out, c->g_dec + (size_t)slot * c->Fs, c->F, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_set_l3_prefetch(vcs_ctx *c, int mode) {
    int rc = use(c);
    if (rc) return rc;
    if (mode < -1 || mode > 2) return fail(c, VCS_E_ARG, "l3 prefetch mode %d (want -1, 0, 1 or 2)", mode);
    c->l3_mode = mode;
    return VCS_OK;
}

int vcs_sync(vcs_ctx *c) {
    int rc = use(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

// same size and slack as d_ref (k_me16's unclamped window rows), zeroed
static int ensure_ref_alt(vcs_ctx *c) {
    if (c->d_ref_alt) return VCS_OK;
    const size_t bytes = c->Fs + me_fast_ref_slack(c->g);
    HIP_TRY(c, hipMalloc(&c->d_ref_alt, bytes));
    HIP_TRY(c, hipMemsetAsync(c->d_ref_alt, 0, bytes, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_dev_ptr(vcs_ctx *c, int which, void **ptr, size_t *bytes) {
    int rc = use(c);
    if (rc) return rc;
    if (!ptr) return fail(c, VCS_E_ARG, "null pointer");
    size_t nmv = (size_t)(c->g.N > 0 ? c->g.N : 1), b = 0;
    void *p = nullptr;
    switch (which) {
        case 0: p = c->d_ref; b = c->F; break;
        case 1: p = c->g_mv; b = nmv * 4 * c->slots; break;
        case 2: p = c->g_ns; b = sizeof(uint32_t) * c->slots; break;
        case 3:
            if ((rc = ensure_ref_alt(c))) return rc;
            p = c->d_ref_alt; b = c->F;
            break;
        default: return fail(c, VCS_E_ARG, "bad buffer id %d", which);
    }
    if (!p) return fail(c, VCS_E_STATE, "buffer %d not allocated (vcs_gop_reserve first)", which);
    *ptr = p;
    if (bytes) *bytes = b;
    return VCS_OK;
}

int vcs_ref_written(vcs_ctx *c) {
    int rc = use(c);
    if (rc || (rc = bits_tables_stale(c, c->d_ref))) return rc;
    c->ref_valid = true;
    c->sums_valid = false;
    return VCS_OK;
}

int vcs_ref_swap(vcs_ctx *c) {
    int rc = use(c);
    if (rc || (rc = ensure_ref_alt(c))) return rc;
    std::swap(c->d_ref, c->d_ref_alt);
    c->ref_valid = true;
    c->sums_valid = false;
    c->next_ready = false;
    return VCS_OK;
}

int vcs_ref_next_ready(vcs_ctx *c, void *written_on_stream) {
    int rc = use(c);
    if (rc || (rc = ensure_ref_alt(c))) return rc;
    if (written_on_stream && (hipStream_t)written_on_stream != c->stream) {
        if (!c->ev_next) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_next, hipEventDisableTiming));
        HIP_TRY(c, hipEventRecord(c->ev_next, (hipStream_t)written_on_stream));
    } else if (c->ev_next) {
        hipEventDestroy(c->ev_next);   // written in the context's own stream order: nothing else to wait for
        c->ev_next = nullptr;
    }
    c->next_ready = true;
    return VCS_OK;
}

int vcs_set_ref_next(vcs_ctx *c, const uint8_t *ref) {
    int rc = use(c);
    if (rc || (rc = ensure_ref_alt(c))) return rc;
    if (!ref) return fail(c, VCS_E_ARG, "null reference frame");
    // (a shadow build still reading the buffer's previous content -- the caller skipped a swap -- first)
    if ((rc = bits_tables_stale(c, c->d_ref_alt))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_ref_alt, ref, c->F, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // the host buffer is the caller's again
    return vcs_ref_next_ready(c, nullptr);
}

// ---- intra prediction -----------------------------------------
static int ensure_intra(vcs_ctx *c) {
    if (c->d_intra) return VCS_OK;
    const size_t hw = (size_t)c->g.H * c->g.W;
    HIP_TRY(c, hipMalloc(&c->d_intra, sizeof(double) * (4 * hw + hw / 16 + 64)));
    return VCS_OK;
}

static int intra_luma(vcs_ctx *c, int bs, const uint8_t *Y, double *res, double *pred,
                      double *modes) {
    int rc = use(c);
    if (rc) return rc;
    if (!Y || !res || !pred || !modes) return fail(c, VCS_E_ARG, "null pointer");
    const int H = c->g.H, W = c->g.W;
    if (H % bs || W % bs || W < 8)
        return fail(c, VCS_E_ARG, "intra luma %dx%d needs plane sizes that are multiples of %d "
                    "(the reference resizes to multiples of 16 first, intraframe.py:324)", bs, bs, bs);
    if ((rc = ensure_intra(c))) return rc;
    const size_t hw = (size_t)H * W, nm = (size_t)(H / bs) * (W / bs);
    double *dres = c->d_intra, *dpred = c->d_intra + hw, *dmodes = c->d_intra + 4 * hw;
    HIP_TRY(c, hipMemcpyAsync(c->d_a, Y, hw, hipMemcpyHostToDevice, c->stream));
    EvPair ev;
    if ((rc = t_begin(c, VCS_T_AUX, &ev))) return rc;
    if (bs == 4)
        hipLaunchKernelGGL(k_intra_luma4, dim3((unsigned)((nm + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_a, H, W, dres, dpred, dmodes);
    else
        hipLaunchKernelGGL(k_intra_luma16, dim3((unsigned)nm), dim3(64), 0, c->stream, c->d_a, H, W,
                           dres, dpred, dmodes);
    if ((rc = t_end(c, &ev))) return rc;
    HIP_TRY(c, hipMemcpyAsync(res, dres, hw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(pred, dpred, hw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(modes, dmodes, nm * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_intra_luma4x4(vcs_ctx *c, const uint8_t *Y, double *res, double *pred, double *modes) {
    return intra_luma(c, 4, Y, res, pred, modes);
}
int vcs_intra_luma16x16(vcs_ctx *c, const uint8_t *Y, double *res, double *pred, double *modes) {
    return intra_luma(c, 16, Y, res, pred, modes);
}

int vcs_intra_chroma8x8(vcs_ctx *c, const uint8_t *Cr, const uint8_t *Cb, double *crres,
                        double *crpred, double *cbres, double *cbpred, double *modes) {
    int rc = use(c);
    if (rc) return rc;
    if (!Cr || !Cb || !crres || !crpred || !cbres || !cbpred || !modes)
        return fail(c, VCS_E_ARG, "null pointer");
    const int H = c->g.H, W = c->g.W;
    if (H % 8 || W % 8) return fail(c, VCS_E_ARG, "intra chroma 8x8 needs multiples of 8");
    if ((rc = ensure_intra(c))) return rc;
    const size_t hw = (size_t)H * W, nm = (size_t)(H / 8) * (W / 8);
    double *d0 = c->d_intra, *dmodes = c->d_intra + 4 * hw;
    HIP_TRY(c, hipMemcpyAsync(c->d_a, Cr, hw, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_a + hw, Cb, hw, hipMemcpyHostToDevice, c->stream));
    EvPair ev;
    if ((rc = t_begin(c, VCS_T_AUX, &ev))) return rc;
    hipLaunchKernelGGL(k_intra_chroma8, dim3(W / 8), dim3(64), 0, c->stream, c->d_a, c->d_a + hw, H,
                       W, d0, d0 + hw, d0 + 2 * hw, d0 + 3 * hw, dmodes);
    if ((rc = t_end(c, &ev))) return rc;
    double *outs[4] = {crres, crpred, cbres, cbpred};
    for (int k = 0; k < 4; k++)
        HIP_TRY(c, hipMemcpyAsync(outs[k], d0 + k * hw, hw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(modes, dmodes, nm * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_intra_transform4x4(vcs_ctx *c, const double *res, double *out) {
    int rc = use(c);
    if (rc) return rc;
    if (!res || !out) return fail(c, VCS_E_ARG, "null pointer");
    const int H = c->g.H, W = c->g.W;
    if (H % 4 || W % 4) return fail(c, VCS_E_ARG, "intra 4x4 transform needs multiples of 4");
    if ((rc = ensure_intra(c))) return rc;
    const size_t hw = (size_t)H * W, nb = hw / 16;
    double *din = c->d_intra, *dout = c->d_intra + hw;
    HIP_TRY(c, hipMemcpyAsync(din, res, hw * 8, hipMemcpyHostToDevice, c->stream));
    EvPair ev;
    if ((rc = t_begin(c, VCS_T_AUX, &ev))) return rc;
    hipLaunchKernelGGL(k_intra_transform4, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream,
                       din, H, W, dout);
    if ((rc = t_end(c, &ev))) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, dout, hw * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

// ---- intra prediction over a batch of I frames ----------------------------------------------
// Layout of one frame's outputs in ib_out (float64 elements, hw = H * W): luma4 res, pred | luma16 res, pred |
// chroma crres, crpred, cbres, cbpred | transform4(luma4 res) | modes: luma4 (hw/16), luma16 (hw/256), chroma (hw/64)
namespace {
struct IbLayout { size_t hw, m4, m16, m8, per_frame; };
IbLayout ib_layout(const vcs_ctx *c) {
    IbLayout l;
    l.hw = (size_t)c->g.H * c->g.W;
    l.m4 = l.hw / 16; l.m16 = l.hw / 256; l.m8 = l.hw / 64;
    l.per_frame = 9 * l.hw + round_up(l.m4 + l.m16 + l.m8, 16);
    return l;
}
}  // namespace

int vcs_intra_batch_reserve(vcs_ctx *c, int n_frames) {
    int rc = use(c);
    if (rc) return rc;
    if (n_frames < 1) return fail(c, VCS_E_ARG, "n_frames must be >= 1");
    if (c->g.H % 16 || c->g.W % 16)
        return fail(c, VCS_E_ARG, "the intra batch needs plane sizes that are multiples of 16 "
                    "(the reference resizes to multiples of 16 first, intraframe.py:324)");
    if (n_frames <= c->ib_frames) return VCS_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipFree(c->ib_in); hipFree(c->ib_out);
    c->ib_in = nullptr; c->ib_out = nullptr; c->ib_frames = 0;
    const IbLayout l = ib_layout(c);
    HIP_TRY(c, hipMalloc(&c->ib_in, (size_t)n_frames * 3 * l.hw + 256));
    HIP_TRY(c, hipMalloc(&c->ib_out, (size_t)n_frames * l.per_frame * sizeof(double)));
    c->ib_frames = n_frames;
    return VCS_OK;
}

static int ib_range(vcs_ctx *c, int first, int n) {
    if (!c->ib_in) return fail(c, VCS_E_STATE, "no intra batch reserved (vcs_intra_batch_reserve)");
    if (first < 0 || n < 1 || first + n > c->ib_frames)
        return fail(c, VCS_E_ARG, "frames [%d, %d) outside the %d reserved", first, first + n, c->ib_frames);
    return VCS_OK;
}

int vcs_intra_batch_upload(vcs_ctx *c, int first, int n, const uint8_t *Y, const uint8_t *Cr, const uint8_t *Cb) {
    int rc = use(c);
    if (rc || (rc = ib_range(c, first, n))) return rc;
    if (!Y || !Cr || !Cb) return fail(c, VCS_E_ARG, "null pointer");
    const IbLayout l = ib_layout(c);
    // per frame [Y][Cr][Cb]; the caller's arrays are n x H x W each
    const uint8_t *src[3] = {Y, Cr, Cb};
    for (int k = 0; k < 3; k++)
        HIP_TRY(c, hipMemcpy2DAsync(c->ib_in + ((size_t)first * 3 + k) * l.hw, 3 * l.hw, src[k], l.hw, l.hw, (size_t)n,
                                    hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_intra_batch_run(vcs_ctx *c, int first, int n, int what) {
    int rc = use(c);
    if (rc || (rc = ib_range(c, first, n))) return rc;
    if (what < 1 || what > 15) return fail(c, VCS_E_ARG, "what = %d: a mask of 1 (luma 4x4), 2 (luma 16x16), 4 (chroma 8x8), 8 (4x4 transform)", what);
    const IbLayout l = ib_layout(c);
    const int H = c->g.H, W = c->g.W;
    const uint8_t *in = c->ib_in + (size_t)first * 3 * l.hw;
    double *out = c->ib_out + (size_t)first * l.per_frame, *modes = out + 9 * l.hw;
    const size_t is = 3 * l.hw, os = l.per_frame;
    EvPair ev;
    if ((rc = t_begin(c, VCS_T_AUX, &ev))) return rc;
    if (what & 1)
        hipLaunchKernelGGL(k_intra_luma4, dim3((unsigned)((l.m4 + 255) / 256), n), dim3(256), 0, c->stream, in, H, W,
                           out, out + l.hw, modes, is, os, os);
    if (what & 2)
        hipLaunchKernelGGL(k_intra_luma16, dim3((unsigned)l.m16, n), dim3(64), 0, c->stream, in, H, W, out + 2 * l.hw,
                           out + 3 * l.hw, modes + l.m4, is, os, os);
    if (what & 4)   // serial down each block column (intraframe.py:266), but the columns of ALL frames at once
        hipLaunchKernelGGL(k_intra_chroma8, dim3(W / 8, n), dim3(64), 0, c->stream, in + l.hw, in + 2 * l.hw, H, W,
                           out + 4 * l.hw, out + 5 * l.hw, out + 6 * l.hw, out + 7 * l.hw, modes + l.m4 + l.m16, is, os, os);
    if (what & 8) {
        if (!(what & 1)) return fail(c, VCS_E_ARG, "the 4x4 transform (8) runs on the luma 4x4 residual (1) of the same call");
        hipLaunchKernelGGL(k_intra_transform4, dim3((unsigned)((l.m4 + 255) / 256), n), dim3(256), 0, c->stream, out, H, W,
                           out + 8 * l.hw, os, os);
    }
    return t_end(c, &ev);
}

int vcs_intra_batch_get(vcs_ctx *c, int frame, int which, double *p0, double *p1, double *p2, double *p3, double *modes) {
    int rc = use(c);
    if (rc || (rc = ib_range(c, frame, 1))) return rc;
    const IbLayout l = ib_layout(c);
    const double *out = c->ib_out + (size_t)frame * l.per_frame, *m = out + 9 * l.hw;
    const double *src[4] = {nullptr, nullptr, nullptr, nullptr};
    const double *msrc = nullptr;
    size_t mn = 0;
    switch (which) {
        case 0: src[0] = out; src[1] = out + l.hw; msrc = m; mn = l.m4; break;                                  // luma4: res, pred
        case 1: src[0] = out + 2 * l.hw; src[1] = out + 3 * l.hw; msrc = m + l.m4; mn = l.m16; break;           // luma16
        case 2: for (int k = 0; k < 4; k++) src[k] = out + (4 + k) * l.hw; msrc = m + l.m4 + l.m16; mn = l.m8; break;   // chroma
        case 3: src[0] = out + 8 * l.hw; break;                                                                  // transform
        default: return fail(c, VCS_E_ARG, "which = %d (0 luma 4x4, 1 luma 16x16, 2 chroma 8x8, 3 transform)", which);
    }
    double *dst[4] = {p0, p1, p2, p3};
    for (int k = 0; k < 4; k++)
        if (src[k] && dst[k]) HIP_TRY(c, hipMemcpyAsync(dst[k], src[k], l.hw * 8, hipMemcpyDeviceToHost, c->stream));
    if (msrc && modes) HIP_TRY(c, hipMemcpyAsync(modes, msrc, mn * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_chroma420(vcs_ctx *c, const uint8_t *bgr, uint8_t *Y, uint8_t *cr_s, uint8_t *cb_s) {
    int rc = use(c);
    if (rc) return rc;
    if (!bgr || !cr_s || !cb_s) return fail(c, VCS_E_ARG, "null pointer");
    const int H = c->g.H, W = c->g.W, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
    const size_t nc = (size_t)Hc * Wc, hw = (size_t)H * W;
    // d_a: the frame; d_b: Y plane (H*W) then the two sample planes (2 * nc <= 2 * F / 3 ... fits F)
    uint8_t *dY = c->d_b, *dcr = c->d_b + hw, *dcb = dcr + nc;
    HIP_TRY(c, hipMemcpyAsync(c->d_a, bgr, c->F, hipMemcpyHostToDevice, c->stream));
    EvPair ev;
    if ((rc = t_begin(c, VCS_T_AUX, &ev))) return rc;
    hipLaunchKernelGGL(k_chroma420, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, c->stream, c->d_a, H,
                       W, Y ? dY : (uint8_t *)nullptr, dcr, dcb);
    if ((rc = t_end(c, &ev))) return rc;
    if (Y) HIP_TRY(c, hipMemcpyAsync(Y, dY, hw, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cr_s, dcr, nc, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cb_s, dcb, nc, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return VCS_OK;
}

int vcs_stream_wait(vcs_ctx *c, void *foreign_stream) {
    int rc = use(c);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->xev, (hipStream_t)foreign_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->xev, 0));
    return VCS_OK;
}

int vcs_stream_signal(vcs_ctx *c, void *foreign_stream) {
    int rc = use(c);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->xev, c->stream));
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)foreign_stream, c->xev, 0));
    return VCS_OK;
}

int vcs_stream_handle(vcs_ctx *c, void **stream) {
    int rc = use(c);
    if (rc) return rc;
    if (!stream) return fail(c, VCS_E_ARG, "null pointer");
    *stream = (void *)c->stream;
    return VCS_OK;
}

// ---- measurement -----------------------------------------------------------
int vcs_timing_enable(vcs_ctx *c, int on) {
    int rc = use(c);
    if (rc) return rc;
    c->timing = on != 0;
    return VCS_OK;
}

static int drain_events(vcs_ctx *c) {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto &p : c->pending) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, p.a, p.b));
        c->acc_ms[p.slot] += ms;
        c->pool.push_back(p);
    }
    c->pending.clear();
    return VCS_OK;
}

int vcs_timing_reset(vcs_ctx *c) {
    int rc = use(c);
    if (rc || (rc = drain_events(c))) return rc;
    for (int i = 0; i < VCS_T_COUNT; i++) { c->acc_ms[i] = 0; c->launches[i] = 0; }
    return VCS_OK;
}

int vcs_timing(vcs_ctx *c, double *ms_out, uint64_t *launches_out, int n) {
    int rc = use(c);
    if (rc || (rc = drain_events(c))) return rc;
    for (int i = 0; i < n && i < VCS_T_COUNT; i++) {
        if (ms_out) ms_out[i] = c->acc_ms[i];
        if (launches_out) launches_out[i] = c->launches[i];
    }
    return VCS_OK;
}

int vcs_me_cycles(vcs_ctx *c, uint64_t *cycles, int reset) {
    int rc = use(c);
    if (rc) return rc;
    if (!cycles) return fail(c, VCS_E_ARG, "null pointer");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long v = 0;
    HIP_TRY(c, hipMemcpy(&v, c->d_clk, 8, hipMemcpyDeviceToHost));
    *cycles = v;
    if (reset) HIP_TRY(c, hipMemset(c->d_clk, 0, 8));
    return VCS_OK;
}

int vcs_me_clock(vcs_ctx *c, double *ghz, int reset) {
    int rc = use(c);
    if (rc) return rc;
    if (!ghz) return fail(c, VCS_E_ARG, "null pointer");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long v[2] = {0, 0};
    HIP_TRY(c, hipMemcpy(v, c->d_clk + 12, 16, hipMemcpyDeviceToHost));
    *ghz = v[1] ? (double)v[0] / (double)v[1] * 0.1 : 0.0;   // s_memrealtime ticks at 100 MHz
    if (reset) HIP_TRY(c, hipMemset(c->d_clk + 12, 0, 16));
    return VCS_OK;
}

#ifdef VCS_EXP_PHASES   // experiment builds only (tools/gpu_phases.py)
extern "C" int vcs_me_phases(vcs_ctx *c, uint64_t *out9) {
    int rc = use(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->g.bs == 8) {
        static unsigned long long zero[16] = {0};
        HIP_TRY(c, hipMemcpyFromSymbol(out9, HIP_SYMBOL(vcs::g_me8_phases), 80));
        HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(vcs::g_me8_phases), zero, 128));
        return VCS_OK;
    }
    HIP_TRY(c, hipMemcpy(out9, c->d_clk, 80, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemset(c->d_clk, 0, 80));
    return VCS_OK;
}
#endif

#ifdef VCS_EXP_PHASES
extern "C" int vcs_me16b_phases(vcs_ctx *c, uint64_t *out8) {
    int rc = use(c);
    if (rc) return rc;
    static unsigned long long zero[8] = {0};
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out8, HIP_SYMBOL(vcs::g_me16b_phases), 64));
    HIP_TRY(c, hipMemcpyToSymbol(HIP_SYMBOL(vcs::g_me16b_phases), zero, 64));
    return VCS_OK;
}
extern "C" int vcs_me_wg_times(vcs_ctx *c, uint64_t *out, int n) {
    int rc = use(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(vcs::g_me16_wg_t), (size_t)n * 16));
    return VCS_OK;
}
extern "C" int vcs_me_wg_hw(vcs_ctx *c, uint64_t *out, int n) {
    int rc = use(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(vcs::g_me16_hw), (size_t)n * 64));
    return VCS_OK;
}
#endif

int vcs_measure_copy_bw(vcs_ctx *c, size_t bytes, int iters, double *gbps_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!gbps_out || bytes < 1024 || iters < 1) return fail(c, VCS_E_ARG, "bad arguments");
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0, e1;
    HIP_TRY(c, hipMalloc(&a, bytes));
    hipError_t e = hipMalloc(&b, bytes);
    if (e != hipSuccess) { hipFree(a); return fail(c, VCS_E_NOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    hipMemsetAsync(a, 1, bytes, c->stream);
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);  // warm
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters; i++) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, c->stream);
    hipEventRecord(e1, c->stream);
    e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a); hipFree(b);
    if (e != hipSuccess) return fail(c, VCS_E_HIP, "copy benchmark: %s", hipGetErrorString(e));
    *gbps_out = 2.0 * (double)bytes * iters / (ms * 1e-3) / 1e9;  // read + write
    return VCS_OK;
}

int vcs_measure_fill_bw(vcs_ctx *c, size_t bytes, int iters, double *gbps_out) {
    int rc = use(c);
    if (rc) return rc;
    if (!gbps_out || bytes < 1024 || iters < 1) return fail(c, VCS_E_ARG, "bad arguments");
    void *a = nullptr;
    hipEvent_t e0, e1;
    HIP_TRY(c, hipMalloc(&a, bytes));
    hipMemsetAsync(a, 1, bytes, c->stream);  // warm
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, c->stream);
    for (int i = 0; i < iters; i++) hipMemsetAsync(a, i, bytes, c->stream);
    hipEventRecord(e1, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(a);
    if (e != hipSuccess) return fail(c, VCS_E_HIP, "fill benchmark: %s", hipGetErrorString(e));
    *gbps_out = (double)bytes * iters / (ms * 1e-3) / 1e9;  // write only
    return VCS_OK;
}

}  // extern "C"

#include "vcs_multi.hpp"
