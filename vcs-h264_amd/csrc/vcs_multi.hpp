// vcs_multi.hpp -- part of vcs_api.hip (included at its end, inside the same translation
// unit): the multi-GPU GOP encoder with RCCL inside the library (SURVEY 8b vcs_encode_gop,
// 8e) and the pipelined single-frame feeder.
//
// Multi-GPU: the P frames of a GOP depend only on the raw I frame (encoder.py:51-52), so
// they are dealt to the shards in contiguous chunks.  The only exchanges of the path are
//   1. ncclBroadcast of the I frame from shard 0 straight into every context's resident
//      reference buffer, and
//   2. ncclAllGather of the motion vectors (+ static-block counts).
// Coefficient planes never leave the GPU that produced them.  Two ways to drive it:
//   * single process, one context per device: vcs_group_create (ncclCommInitAll) -- what
//     vcs_encode_gop and `bench.py --gpus N` use;
//   * one process per GPU: vcs_group_create_rank (ncclCommInitRank) -- `bench.py` under
//     torch.distributed.run; the launcher only hands out ranks, no PyTorch in the process.
// Collectives run on their own stream per context, ordered against the kernels' stream by
// events only: the I frame of GOP k+1 is broadcast into the context's ALTERNATE reference
// buffer while GOP k is encoded, GOP k's vectors are copied aside (1 MB) behind its kernels
// and gathered from there while GOP k+1 is encoded.  Every shard enqueues the collectives in
// the same order.  When several contexts sit on ONE device (RCCL refuses duplicate devices)
// the same schedule runs with device-to-device copies instead: that is how the sharding and
// gather layout are tested on a one-GPU box.
#pragma once
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

// One host thread per local shard for the per-GOP enqueue (single-process groups): a shard's enqueue -- five
// kernel launches, two copies, the event traffic -- costs ~0.06 ms of host time, and in sequence from one
// thread eight shards cost as much host time as the GPUs need for the step.  Shard 0 is enqueued by the
// calling thread, the others by workers that live as long as the group.  (One process per GPU -- what
// torch.distributed.run starts -- has one shard per process and never gets here.)  VCS_GROUP_THREADS=0:
// everything from the calling thread.
struct GroupPool {
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::function<int(int)> job;
    uint64_t gen = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rc;

    void start(int n) {
        rc.assign(n, 0);
        for (int i = 1; i < n; i++)
            workers.emplace_back([this, i] {
                uint64_t seen = 0;
                for (;;) {
                    std::function<int(int)> j;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_go.wait(lk, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        j = job;
                    }
                    const int r = j(i);
                    {
                        std::lock_guard<std::mutex> lk(m);
                        rc[i] = r;
                        if (--pending == 0) cv_done.notify_one();
                    }
                }
            });
    }
    // job(i) for every shard i; returns the first non-zero result (by shard index)
    int run(int n, const std::function<int(int)> &j) {
        if (workers.empty()) {
            for (int i = 0; i < n; i++) {
                const int r = j(i);
                if (r) return r;
            }
            return 0;
        }
        {
            std::lock_guard<std::mutex> lk(m);
            job = j;
            pending = (int)workers.size();
            gen++;
        }
        cv_go.notify_all();
        rc[0] = j(0);
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
        for (int i = 0; i < n; i++)
            if (rc[i]) return rc[i];
        return 0;
    }
    ~GroupPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_go.notify_all();
        for (auto &t : workers) t.join();
    }
};

struct vcs_group {
    struct Per {
        vcs_ctx *c = nullptr;
        ncclComm_t comm = nullptr;
        hipStream_t cs = nullptr;   // collectives' stream (== c->stream in serial mode)
        bool own_cs = false;
        hipEvent_t ev_ref[2] = {nullptr, nullptr};     // broadcast into reference buffer b done (cs)
        hipEvent_t ev_kern[2] = {nullptr, nullptr};    // kernels of the GOP on buffer b + stage copy done
        hipEvent_t ev_gather[2] = {nullptr, nullptr};  // gather out of stage[b] done (cs)
        bool kern_rec[2] = {false, false}, gather_rec[2] = {false, false};
        uint8_t *refbuf[2] = {nullptr, nullptr};
        uint8_t *stage[2] = {nullptr, nullptr};        // [cmax * N * 4 bytes of vectors][cmax * 4 of counts]
        uint8_t *all[2] = {nullptr, nullptr};          // world * chunk bytes
        int first = 0, count = 0;                      // this shard's P frames
    };
    std::vector<Per> per;   // local shards, in global order
    int world = 1;          // shards over all processes
    int rank0 = 0;          // global index of per[0]
    bool rccl = false;      // RCCL transport; false: same-device copies (single process only)
    bool overlap = true;
    bool with_decode = false;
    int n_p = -1, cmax = 0;
    int sets = 1, set = 0;  // resident copies of the GOP's slots per shard; the one in use
    size_t chunk = 0;       // bytes one shard contributes to the gather
    uint64_t gop = 0;       // GOPs encoded so far
    uint64_t gop_prefetched = 0;  // GOPs whose I frame has been broadcast (or enqueued)
    int gather_pending = -1;      // stage buffer whose all-gather is still to be enqueued (-1: none)
    uint8_t *d_iframe = nullptr;  // on shard 0's device: the staged I frame of the next GOP
    hipEvent_t ev_iframe = nullptr;
    bool iframe_valid = false;
    uint8_t *d_word = nullptr;    // per[0]'s device: 64 bytes for barrier / max reductions and the parameter-hash check
    std::vector<uint8_t> host;    // D2H staging of the gathered vectors
    GroupPool *pool = nullptr;    // one enqueue thread per local shard beyond the first (single-process groups)
    std::string err;
};

namespace {

thread_local std::string g_gerr;

int gfail(vcs_group *g, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    g_gerr = buf;
    g_err = buf;
    return code;
}

#define G_HIP(g, expr)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess)                                                            \
            return gfail((g), VCS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define G_NCCL(g, expr)                                                                    \
    do {                                                                                   \
        ncclResult_t r_ = (expr);                                                          \
        if (r_ != ncclSuccess)                                                             \
            return gfail((g), VCS_E_COMM, "%s failed: %s", #expr, ncclGetErrorString(r_)); \
    } while (0)

void shard_of(int n, int world, int rank, int *first, int *count) {
    const int base = n / world, extra = n % world;
    *count = base + (rank < extra ? 1 : 0);
    *first = rank * base + (rank < extra ? rank : extra);
}

// VCS_GROUP_SERIAL=1: collectives on the kernels' stream, strictly in sequence; =0: overlapped
// on their own stream.  Default: overlapped when there is somebody to exchange with; a
// one-shard group's "collectives" are local copies with nothing to hide behind the kernels,
// and the cross-stream waits cost more than they save (measured: +4 % overlapped, +1.5-2 %
// in sequence, against the plain single-context path).
bool group_overlap(int world) {
    const char *e = getenv("VCS_GROUP_SERIAL");
    if (e && e[0] == '1') return false;
    if (e && e[0] == '0') return true;
    return world > 1;
}

int group_init_common(vcs_group *g) {
    g->overlap = group_overlap(g->world);
    for (auto &p : g->per) {
        vcs_ctx *c = p.c;
        G_HIP(g, hipSetDevice(c->device));
        int rc = ensure_ref_alt(c);
        if (rc) return gfail(g, rc, "%s", c->err.c_str());
        p.refbuf[0] = c->d_ref;
        p.refbuf[1] = c->d_ref_alt;
        if (g->overlap) {
            // highest priority: the collectives' few workgroups must not queue behind the tail
            // kernel's 117 k (VCS_GROUP_PRIO=0: default priority)
            int lo = 0, hi = 0;
            G_HIP(g, hipDeviceGetStreamPriorityRange(&lo, &hi));
            const char *pe = getenv("VCS_GROUP_PRIO");
            G_HIP(g, hipStreamCreateWithPriority(&p.cs, hipStreamNonBlocking, (pe && pe[0] == '0') ? 0 : hi));
            p.own_cs = true;
            if (!c->ev_me_done) G_HIP(g, hipEventCreateWithFlags(&c->ev_me_done, hipEventDisableTiming));
        } else {
            p.cs = c->stream;
        }
        for (int b = 0; b < 2; b++) {
            G_HIP(g, hipEventCreateWithFlags(&p.ev_ref[b], hipEventDisableTiming));
            G_HIP(g, hipEventCreateWithFlags(&p.ev_kern[b], hipEventDisableTiming));
            G_HIP(g, hipEventCreateWithFlags(&p.ev_gather[b], hipEventDisableTiming));
        }
    }
    vcs_ctx *c0 = g->per[0].c;
    G_HIP(g, hipSetDevice(c0->device));
    G_HIP(g, hipMalloc(&g->d_word, 64));
    G_HIP(g, hipMemset(g->d_word, 0, 64));
    if (g->rank0 == 0) {
        G_HIP(g, hipMalloc(&g->d_iframe, c0->Fs));
        G_HIP(g, hipMemset(g->d_iframe, 0, c0->Fs));
        G_HIP(g, hipEventCreateWithFlags(&g->ev_iframe, hipEventDisableTiming));
    }
    return VCS_OK;
}

// Everything that decides a shard's results: geometry, search parameters, threshold, search / cost
// selectors, coefficient mode and the three quantiser tables (FNV-1a).
uint64_t group_param_hash(const vcs_ctx *c) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    };
    const int v[9] = {c->g.H, c->g.W, c->g.bs, c->g.win, c->g.step, c->coeff_mode, c->search, c->cost, 0};
    mix(v, sizeof v);
    mix(&c->thresh64, sizeof c->thresh64);
    mix(c->hDQ + 64, sizeof(double) * 192);
    return h;
}

int group_check_same_geometry(vcs_group *g, vcs_ctx *const *ctxs, int n) {
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return gfail(g, VCS_E_ARG, "context %d is null", i);
        const Geom &a = ctxs[0]->g, &b = ctxs[i]->g;
        if (a.H != b.H || a.W != b.W || a.bs != b.bs || a.win != b.win || a.step != b.step ||
            ctxs[0]->coeff_mode != ctxs[i]->coeff_mode)
            return gfail(g, VCS_E_ARG, "context %d differs from context 0 in geometry / parameters", i);
        if (group_param_hash(ctxs[0]) != group_param_hash(ctxs[i]))
            return gfail(g, VCS_E_ARG, "context %d differs from context 0 in threshold / search / cost / Q tables", i);
        for (int j = 0; j < i; j++)
            if (ctxs[j] == ctxs[i]) return gfail(g, VCS_E_ARG, "context %d listed twice", i);
    }
    return VCS_OK;
}

void group_free_buffers(vcs_group *g) {
    for (auto &p : g->per) {
        hipSetDevice(p.c->device);
        for (int b = 0; b < 2; b++) {
            hipFree(p.stage[b]); hipFree(p.all[b]);
            p.stage[b] = p.all[b] = nullptr;
            p.kern_rec[b] = p.gather_rec[b] = false;
        }
    }
}

int group_sync_all(vcs_group *g) {
    for (auto &p : g->per) {
        G_HIP(g, hipSetDevice(p.c->device));
        G_HIP(g, hipStreamSynchronize(p.c->stream));
        if (p.own_cs) G_HIP(g, hipStreamSynchronize(p.cs));
    }
    return VCS_OK;
}

// Broadcast of the staged I frame into reference buffer b of every local shard (async).
int group_bcast(vcs_group *g, int b) {
    const size_t F = g->per[0].c->F;
    if (g->rccl) {
        // every stream first waits until the GOP that last read buffer b has finished
        for (auto &p : g->per) {
            G_HIP(g, hipSetDevice(p.c->device));
            if (p.kern_rec[b] && p.cs != p.c->stream) G_HIP(g, hipStreamWaitEvent(p.cs, p.ev_kern[b], 0));
            // ... and start behind the search kernel of the GOP being encoded: the copy kernels of
            // the collective then run beside the HBM-bound tail instead of taking CUs from the
            // search, whose persistent workgroups are dealt to a full chip
            if (p.c->me_done_rec && p.cs != p.c->stream) G_HIP(g, hipStreamWaitEvent(p.cs, p.c->ev_me_done, 0));
        }
        G_NCCL(g, ncclGroupStart());
        for (size_t i = 0; i < g->per.size(); i++) {
            auto &p = g->per[i];
            const bool root = g->rank0 + (int)i == 0;
            hipSetDevice(p.c->device);
            ncclResult_t r = ncclBroadcast(root ? (const void *)g->d_iframe : (const void *)p.refbuf[b],
                                           p.refbuf[b], F, ncclUint8, 0, p.comm, p.cs);
            if (r != ncclSuccess) {
                ncclGroupEnd();
                return gfail(g, VCS_E_COMM, "ncclBroadcast failed: %s", ncclGetErrorString(r));
            }
        }
        G_NCCL(g, ncclGroupEnd());
    } else {
        // same-device transport: plain copies out of shard 0's staging buffer
        for (auto &p : g->per) {
            G_HIP(g, hipSetDevice(p.c->device));
            if (p.kern_rec[b] && p.cs != p.c->stream) G_HIP(g, hipStreamWaitEvent(p.cs, p.ev_kern[b], 0));
            if (p.c->me_done_rec && p.cs != p.c->stream) G_HIP(g, hipStreamWaitEvent(p.cs, p.c->ev_me_done, 0));
            G_HIP(g, hipStreamWaitEvent(p.cs, g->ev_iframe, 0));
            G_HIP(g, hipMemcpyAsync(p.refbuf[b], g->d_iframe, F, hipMemcpyDeviceToDevice, p.cs));
        }
    }
    for (auto &p : g->per) {
        G_HIP(g, hipSetDevice(p.c->device));
        G_HIP(g, hipEventRecord(p.ev_ref[b], p.cs));
    }
    return VCS_OK;
}

// Enqueues the pending all-gather (see vcs_group_encode).
int group_flush_gather(vcs_group *g) {
    if (g->gather_pending < 0) return VCS_OK;
    const int b = g->gather_pending;
    g->gather_pending = -1;
    if (g->rccl) {
        for (auto &p : g->per) {
            G_HIP(g, hipSetDevice(p.c->device));
            if (p.cs != p.c->stream) G_HIP(g, hipStreamWaitEvent(p.cs, p.ev_kern[b], 0));
        }
        G_NCCL(g, ncclGroupStart());
        for (auto &p : g->per) {
            hipSetDevice(p.c->device);
            ncclResult_t r = ncclAllGather(p.stage[b], p.all[b], g->chunk, ncclUint8, p.comm, p.cs);
            if (r != ncclSuccess) {
                ncclGroupEnd();
                return gfail(g, VCS_E_COMM, "ncclAllGather failed: %s", ncclGetErrorString(r));
            }
        }
        G_NCCL(g, ncclGroupEnd());
    } else {
        for (size_t i = 0; i < g->per.size(); i++) {
            auto &p = g->per[i];
            G_HIP(g, hipSetDevice(p.c->device));
            if (p.cs != p.c->stream) G_HIP(g, hipStreamWaitEvent(p.cs, p.ev_kern[b], 0));
            for (auto &q : g->per)
                G_HIP(g, hipMemcpyAsync(q.all[b] + (size_t)(g->rank0 + i) * g->chunk, p.stage[b], g->chunk,
                                        hipMemcpyDeviceToDevice, p.cs));
        }
    }
    for (auto &p : g->per) {
        G_HIP(g, hipSetDevice(p.c->device));
        G_HIP(g, hipEventRecord(p.ev_gather[b], p.cs));
        p.gather_rec[b] = true;
    }
    return VCS_OK;
}

}  // namespace

extern "C" {

int vcs_group_layout_sets(vcs_group *g, int n_p_frames, int with_decode, int n_sets);

int vcs_shard_frames(int n_frames, int world, int rank, int *first, int *count) {
    if (!first || !count || world < 1 || rank < 0 || rank >= world || n_frames < 0)
        return gfail(nullptr, VCS_E_ARG, "bad shard request (n = %d, world = %d, rank = %d)", n_frames,
                     world, rank);
    shard_of(n_frames, world, rank, first, count);
    return VCS_OK;
}

const char *vcs_group_last_error(const vcs_group *g) { return g ? g->err.c_str() : g_gerr.c_str(); }

int vcs_comm_unique_id(void *id, size_t bytes) {
    if (!id || bytes < NCCL_UNIQUE_ID_BYTES)
        return gfail(nullptr, VCS_E_ARG, "unique id buffer must hold %d bytes", NCCL_UNIQUE_ID_BYTES);
    ncclUniqueId u;
    G_NCCL(nullptr, ncclGetUniqueId(&u));
    memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
    return VCS_OK;
}

int vcs_group_destroy(vcs_group *g) {
    if (!g) return VCS_OK;
    group_flush_gather(g);   // (keeps the ranks' collective sequences equal up to the end)
    group_sync_all(g);
    group_free_buffers(g);
    for (auto &p : g->per) {
        hipSetDevice(p.c->device);
        if (p.comm) ncclCommDestroy(p.comm);
        for (int b = 0; b < 2; b++) {
            if (p.ev_ref[b]) hipEventDestroy(p.ev_ref[b]);
            if (p.ev_kern[b]) hipEventDestroy(p.ev_kern[b]);
            if (p.ev_gather[b]) hipEventDestroy(p.ev_gather[b]);
        }
        if (p.own_cs && p.cs) hipStreamDestroy(p.cs);
        // the contexts go back to a plain single-GPU state: refbuf[0/1] are their two buffers
        p.c->group = nullptr;
        if (p.c->ev_me_done) { hipEventDestroy(p.c->ev_me_done); p.c->ev_me_done = nullptr; p.c->me_done_rec = false; }
    }
    if (!g->per.empty()) hipSetDevice(g->per[0].c->device);
    hipFree(g->d_iframe);
    hipFree(g->d_word);
    if (g->ev_iframe) hipEventDestroy(g->ev_iframe);
    delete g->pool;
    delete g;
    return VCS_OK;
}

int vcs_group_create(vcs_ctx *const *ctxs, int n_ctx, vcs_group **out) {
    if (!out) return gfail(nullptr, VCS_E_ARG, "null out pointer");
    *out = nullptr;
    if (!ctxs || n_ctx < 1 || n_ctx > 64) return gfail(nullptr, VCS_E_ARG, "bad context list");
    int rc = group_check_same_geometry(nullptr, ctxs, n_ctx);
    if (rc) return rc;
    for (int i = 0; i < n_ctx; i++)
        if (ctxs[i]->group) return gfail(nullptr, VCS_E_STATE, "context %d already belongs to a group", i);
    vcs_group *g = new (std::nothrow) vcs_group();
    if (!g) return gfail(nullptr, VCS_E_NOMEM, "out of host memory");
    g->world = n_ctx;
    g->rank0 = 0;
    g->per.resize(n_ctx);
    bool distinct = true;
    for (int i = 0; i < n_ctx; i++) {
        g->per[i].c = ctxs[i];
        for (int j = 0; j < i; j++) distinct = distinct && ctxs[i]->device != ctxs[j]->device;
    }
    // RCCL needs one device per rank; contexts that share a device exchange by copies.
    // VCS_GROUP_TRANSPORT=copy forces the copy transport (diagnostics).
    const char *tr = getenv("VCS_GROUP_TRANSPORT");
    g->rccl = distinct && !(tr && !strcmp(tr, "copy"));
    if (!distinct) {
        for (int i = 1; i < n_ctx; i++)
            if (ctxs[i]->device != ctxs[0]->device) {
                delete g;
                return gfail(nullptr, VCS_E_ARG, "contexts must sit on pairwise distinct devices "
                             "(RCCL) or all on one device (copy transport)");
            }
    }
    rc = group_init_common(g);
    if (!rc && g->rccl) {
        std::vector<int> devs(n_ctx);
        std::vector<ncclComm_t> comms(n_ctx);
        for (int i = 0; i < n_ctx; i++) devs[i] = ctxs[i]->device;
        ncclResult_t r = ncclCommInitAll(comms.data(), n_ctx, devs.data());
        if (r != ncclSuccess)
            rc = gfail(g, VCS_E_COMM, "ncclCommInitAll(%d devices) failed: %s", n_ctx, ncclGetErrorString(r));
        else
            for (int i = 0; i < n_ctx; i++) g->per[i].comm = comms[i];
    }
    if (rc) {
        std::string keep = g->err;
        vcs_group_destroy(g);
        g_gerr = g_err = keep;
        return rc;
    }
    for (int i = 0; i < n_ctx; i++) ctxs[i]->group = g;
    *out = g;
    return VCS_OK;
}

int vcs_group_create_rank(vcs_ctx *ctx, const void *unique_id, int world, int rank, vcs_group **out) {
    if (!out) return gfail(nullptr, VCS_E_ARG, "null out pointer");
    *out = nullptr;
    if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world)
        return gfail(nullptr, VCS_E_ARG, "bad arguments (world = %d, rank = %d)", world, rank);
    if (ctx->group) return gfail(nullptr, VCS_E_STATE, "context already belongs to a group");
    vcs_group *g = new (std::nothrow) vcs_group();
    if (!g) return gfail(nullptr, VCS_E_NOMEM, "out of host memory");
    g->world = world;
    g->rank0 = rank;
    g->per.resize(1);
    g->per[0].c = ctx;
    g->rccl = true;
    int rc = group_init_common(g);
    if (!rc) {
        ncclUniqueId u;
        memcpy(&u, unique_id, NCCL_UNIQUE_ID_BYTES);
        hipSetDevice(ctx->device);
        ncclResult_t r = ncclCommInitRank(&g->per[0].comm, world, u, rank);
        if (r != ncclSuccess)
            rc = gfail(g, VCS_E_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world,
                       ncclGetErrorString(r));
    }
    if (rc) {
        std::string keep = g->err;
        vcs_group_destroy(g);
        g_gerr = g_err = keep;
        return rc;
    }
    ctx->group = g;
    *out = g;
    return VCS_OK;
}

int vcs_group_info(vcs_group *g, int *world, int *rank0, int *n_local, int *transport) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (world) *world = g->world;
    if (rank0) *rank0 = g->rank0;
    if (n_local) *n_local = (int)g->per.size();
    if (transport) *transport = g->rccl ? (g->overlap ? 1 : 2) : (g->overlap ? 3 : 4);
    return VCS_OK;
}

int vcs_group_shard(vcs_group *g, int local_idx, int *first, int *count) {
    if (!g || local_idx < 0 || local_idx >= (int)g->per.size()) return gfail(g, VCS_E_ARG, "bad shard index");
    if (g->n_p < 0) return gfail(g, VCS_E_STATE, "no GOP layout yet (vcs_group_layout)");
    if (first) *first = g->per[local_idx].first;
    if (count) *count = g->per[local_idx].count;
    return VCS_OK;
}

int vcs_group_layout(vcs_group *g, int n_p_frames, int with_decode) {
    return vcs_group_layout_sets(g, n_p_frames, with_decode, 1);
}

int vcs_group_use_set(vcs_group *g, int set) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (set < 0 || set >= g->sets) return gfail(g, VCS_E_ARG, "set %d outside the %d laid out", set, g->sets);
    g->set = set;
    return VCS_OK;
}

int vcs_group_layout_sets(vcs_group *g, int n_p_frames, int with_decode, int n_sets) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (n_sets < 1 || n_sets > 64) return gfail(g, VCS_E_ARG, "bad set count %d", n_sets);
    { int rc0 = group_flush_gather(g); if (rc0) return rc0; }
    if (n_p_frames < 0 || n_p_frames > 4096 * g->world) return gfail(g, VCS_E_ARG, "bad frame count %d", n_p_frames);
    int rc = group_sync_all(g);
    if (rc) return rc;
    group_free_buffers(g);
    // differently configured shards would encode silently: the parameter block of every shard
    // (every process's too: max and min of its hash through the communicator) must be the same
    {
        std::vector<vcs_ctx *> cs;
        for (auto &p : g->per) cs.push_back(p.c);
        if ((rc = group_check_same_geometry(g, cs.data(), (int)cs.size()))) return rc;
        if (g->rccl && g->world != (int)g->per.size()) {
            auto &p0 = g->per[0];
            const uint64_t h = group_param_hash(p0.c);
            uint64_t pair[2] = {h, ~h}, red[2] = {0, 0};
            G_HIP(g, hipSetDevice(p0.c->device));
            G_HIP(g, hipMemcpy(g->d_word, pair, 16, hipMemcpyHostToDevice));
            G_NCCL(g, ncclAllReduce(g->d_word, g->d_word + 16, 2, ncclUint64, ncclMax, p0.comm, p0.cs));
            G_HIP(g, hipStreamSynchronize(p0.cs));
            G_HIP(g, hipMemcpy(red, g->d_word + 16, 16, hipMemcpyDeviceToHost));
            if (red[0] != h || red[1] != ~h)   // max(h) and max(~h) = ~min(h): all equal iff both are mine
                return gfail(g, VCS_E_ARG, "the shards' parameters differ across processes (geometry, search "
                                           "window / stride / threshold, search / cost selectors, coefficient mode or Q tables)");
        }
    }
    g->n_p = n_p_frames;
    g->sets = n_sets;
    g->set = 0;
    g->with_decode = with_decode != 0;
    g->cmax = (n_p_frames + g->world - 1) / g->world;
    const size_t N = (size_t)(g->per[0].c->g.N > 0 ? g->per[0].c->g.N : 1);
    g->chunk = round_up((size_t)(g->cmax > 0 ? g->cmax : 1) * (N * 4 + 4), 256);
    for (size_t i = 0; i < g->per.size(); i++) {
        auto &p = g->per[i];
        shard_of(n_p_frames, g->world, g->rank0 + (int)i, &p.first, &p.count);
        G_HIP(g, hipSetDevice(p.c->device));
        if (p.count * n_sets > p.c->slots || (with_decode && !p.c->gop_dec)) {
            rc = vcs_gop_reserve(p.c, p.count > 0 ? p.count * n_sets : 1, with_decode);
            if (rc) return gfail(g, rc, "shard %d: %s", g->rank0 + (int)i, p.c->err.c_str());
        }
        for (int b = 0; b < 2; b++) {
            G_HIP(g, hipMalloc(&p.stage[b], g->chunk));
            G_HIP(g, hipMalloc(&p.all[b], g->chunk * g->world));
            G_HIP(g, hipMemset(p.stage[b], 0, g->chunk));
            G_HIP(g, hipMemset(p.all[b], 0, g->chunk * g->world));
        }
    }
    g->host.resize(g->chunk * g->world);
    return VCS_OK;
}

int vcs_group_upload(vcs_group *g, int first_p, int n, const uint8_t *frames) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (g->n_p < 0) return gfail(g, VCS_E_STATE, "no GOP layout yet (vcs_group_layout)");
    if (!frames || first_p < 0 || n < 0 || first_p + n > g->n_p)
        return gfail(g, VCS_E_ARG, "P frames [%d, %d) outside the GOP's %d", first_p, first_p + n, g->n_p);
    const size_t F = g->per[0].c->F;
    for (auto &p : g->per) {
        const int lo = std::max(first_p, p.first), hi = std::min(first_p + n, p.first + p.count);
        if (lo >= hi) continue;
        int rc = vcs_gop_upload(p.c, g->set * p.count + lo - p.first, hi - lo,
                                frames + (size_t)(lo - first_p) * F);
        if (rc) return gfail(g, rc, "%s", p.c->err.c_str());
    }
    return VCS_OK;
}

int vcs_group_set_ref(vcs_group *g, const uint8_t *ref_bgr) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (g->rank0 != 0) {
        // only the process that holds shard 0 feeds the I frame; the others just keep their
        // sequence of collectives in step with it (see the end of this function)
        if (g->gop_prefetched > g->gop) g->gop_prefetched = g->gop;
        return VCS_OK;
    }
    if (!ref_bgr) return gfail(g, VCS_E_ARG, "null reference frame");
    auto &p0 = g->per[0];
    G_HIP(g, hipSetDevice(p0.c->device));
    if (!g->rccl)  // copies out of d_iframe by the other shards' streams must have finished
        for (auto &p : g->per)
            for (int b = 0; b < 2; b++)
                if (g->gop_prefetched) G_HIP(g, hipStreamWaitEvent(p0.cs, p.ev_ref[b], 0));
    G_HIP(g, hipMemcpyAsync(g->d_iframe, ref_bgr, p0.c->F, hipMemcpyHostToDevice, p0.cs));
    G_HIP(g, hipEventRecord(g->ev_iframe, p0.cs));
    G_HIP(g, hipStreamSynchronize(p0.cs));  // the host buffer is the caller's again
    g->iframe_valid = true;
    // a broadcast already enqueued for the next GOP carried the PREVIOUS I frame: this one
    // supersedes it (the next vcs_group_encode / _prefetch_ref broadcasts again, behind it on
    // the same stream, into the same buffer)
    if (g->gop_prefetched > g->gop) g->gop_prefetched = g->gop;
    return VCS_OK;
}

int vcs_group_prefetch_ref(vcs_group *g) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (g->rank0 == 0 && !g->iframe_valid) return gfail(g, VCS_E_STATE, "no I frame staged (vcs_group_set_ref)");
    if (g->gop_prefetched > g->gop) return group_flush_gather(g);  // already on its way
    int rc = group_bcast(g, (int)(g->gop_prefetched & 1));
    if (rc) return rc;
    g->gop_prefetched++;
    return group_flush_gather(g);   // behind the broadcast on the collectives' stream
}

int vcs_group_encode(vcs_group *g) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (g->n_p < 0) return gfail(g, VCS_E_STATE, "no GOP layout yet (vcs_group_layout)");
    int rc;
    if ((rc = group_flush_gather(g))) return rc;   // the previous GOP's vectors
    if (g->gop_prefetched <= g->gop && (rc = vcs_group_prefetch_ref(g))) return rc;
    const int b = (int)(g->gop & 1);
    // one shard's enqueue; errors land in the shard's context (c->err), the caller reports the first
    auto shard = [g, b](int i) -> int {
        vcs_group::Per &p = g->per[i];
        vcs_ctx *c = p.c;
        c->err.clear();
        HIP_TRY(c, hipSetDevice(c->device));
        if (p.cs != c->stream) {
            HIP_TRY(c, hipStreamWaitEvent(c->stream, p.ev_ref[b], 0));  // the I frame has arrived
            if (p.gather_rec[b]) HIP_TRY(c, hipStreamWaitEvent(c->stream, p.ev_gather[b], 0));  // stage[b] is free
        }
        c->d_ref = p.refbuf[b];
        c->d_ref_alt = p.refbuf[b ^ 1];
        c->ref_valid = true;
        c->sums_valid = false;
        const size_t N = (size_t)c->g.N;
        if (p.count > 0) {
            const size_t s0 = (size_t)g->set * p.count;  // first slot of the set in use
            int rc = vcs_gop_encode(c, (int)s0, p.count);
            if (rc || (rc = use(c))) return rc;          // (use: joins CU-split streams, if any)
            if (N) HIP_TRY(c, hipMemcpyAsync(p.stage[b], c->g_mv + s0 * N * 2, (size_t)p.count * N * 4,
                                             hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(p.stage[b] + (size_t)g->cmax * (N ? N : 1) * 4, c->g_ns + s0, (size_t)p.count * 4,
                                      hipMemcpyDeviceToDevice, c->stream));
        }
        HIP_TRY(c, hipEventRecord(p.ev_kern[b], c->stream));
        p.kern_rec[b] = true;
        return VCS_OK;
    };
    const int n_local = (int)g->per.size();
    if (n_local > 1 && !g->pool) {
        static const bool threads = [] { const char *e = getenv("VCS_GROUP_THREADS"); return !(e && e[0] == '0'); }();
        g->pool = new (std::nothrow) GroupPool();
        if (g->pool && threads) g->pool->start(n_local);
    }
    if (g->pool) {
        rc = g->pool->run(n_local, shard);
    } else {
        rc = VCS_OK;
        for (int i = 0; i < n_local && !rc; i++) rc = shard(i);
    }
    if (rc) {
        for (int i = 0; i < n_local; i++)
            if (!g->per[i].c->err.empty()) return gfail(g, rc, "shard %d: %s", g->rank0 + i, g->per[i].c->err.c_str());
        return gfail(g, rc, "a shard's enqueue failed");
    }
    // The all-gather of this GOP's vectors is enqueued LAZILY, by the next group call: a
    // vcs_group_prefetch_ref that follows puts the next I frame's broadcast IN FRONT of it on
    // the collectives' stream (the broadcast runs beside this GOP's tail kernel; the gather
    // has to wait for the tail anyway).  Every process makes the same calls in the same order.
    g->gather_pending = b;
    g->gop++;
    return VCS_OK;
}

int vcs_group_decode(vcs_group *g) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (g->n_p < 0 || !g->with_decode) return gfail(g, VCS_E_STATE, "layout was made without decode buffers");
    for (auto &p : g->per) {
        if (p.count <= 0) continue;
        int rc = vcs_gop_decode(p.c, g->set * p.count, p.count);
        if (rc) return gfail(g, rc, "%s", p.c->err.c_str());
    }
    return VCS_OK;
}

int vcs_group_sync(vcs_group *g) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    int rc = group_flush_gather(g);
    return rc ? rc : group_sync_all(g);
}

int vcs_group_get_mv(vcs_group *g, int16_t *mv_out, uint32_t *ns_out) {
    if (!g) return gfail(nullptr, VCS_E_ARG, "null group");
    if (g->gop == 0) return gfail(g, VCS_E_STATE, "nothing encoded yet");
    int rc = group_flush_gather(g);
    if (rc || (rc = group_sync_all(g))) return rc;
    const int b = (int)((g->gop - 1) & 1);
    auto &p0 = g->per[0];
    const size_t N = (size_t)p0.c->g.N, Nn = N ? N : 1;
    G_HIP(g, hipSetDevice(p0.c->device));
    G_HIP(g, hipMemcpy(g->host.data(), p0.all[b], g->chunk * g->world, hipMemcpyDeviceToHost));
    for (int r = 0; r < g->world; r++) {
        int first, count;
        shard_of(g->n_p, g->world, r, &first, &count);
        const uint8_t *src = g->host.data() + (size_t)r * g->chunk;
        if (mv_out && N) memcpy(mv_out + (size_t)first * N * 2, src, (size_t)count * N * 4);
        if (ns_out) memcpy(ns_out + first, src + (size_t)g->cmax * Nn * 4, (size_t)count * 4);
    }
    return VCS_OK;
}

// max over all shards' processes of *v (bench.py's max-over-ranks step time); also a barrier.
int vcs_group_max_f64(vcs_group *g, double *v) {
    if (!g || !v) return gfail(g, VCS_E_ARG, "null pointer");
    int rc = group_flush_gather(g);
    if (rc || (rc = group_sync_all(g))) return rc;
    if (!g->rccl || g->world == (int)g->per.size()) return VCS_OK;  // one process: nothing to reduce
    auto &p0 = g->per[0];
    G_HIP(g, hipSetDevice(p0.c->device));
    G_HIP(g, hipMemcpy(g->d_word, v, 8, hipMemcpyHostToDevice));
    G_NCCL(g, ncclAllReduce(g->d_word, g->d_word + 8, 1, ncclDouble, ncclMax, p0.comm, p0.cs));
    G_HIP(g, hipStreamSynchronize(p0.cs));
    G_HIP(g, hipMemcpy(v, g->d_word + 8, 8, hipMemcpyDeviceToHost));
    return VCS_OK;
}

int vcs_group_barrier(vcs_group *g) {
    double z = 0.0;
    return vcs_group_max_f64(g, &z);
}

// the whole GOP in one call.  frames: n_frames x F, frame 0 is the I frame,
// frames 1.. are its P frames (encoder.py:25,51-52).  mv_out: (n_frames - 1) x N x [dx, dy];
// planes_out (nullable): (n_frames - 1) x 3 x H x W coefficients.  The group (communicators,
// streams) is created on first use and kept on the contexts until one of them is destroyed.
int vcs_encode_gop(vcs_ctx *const *ctxs, int n_ctx, const uint8_t *frames, int n_frames,
                   int16_t *mv_out, void *planes_out) {
    if (!ctxs || n_ctx < 1 || !ctxs[0]) return gfail(nullptr, VCS_E_ARG, "bad context list");
    if (!frames || n_frames < 1 || !mv_out) return gfail(nullptr, VCS_E_ARG, "null pointer / no frames");
    vcs_group *g = ctxs[0]->group;
    bool match = g && (int)g->per.size() == n_ctx && g->world == n_ctx;
    for (int i = 0; match && i < n_ctx; i++) match = g->per[i].c == ctxs[i];
    int rc;
    if (!match) {
        if (g) return gfail(nullptr, VCS_E_STATE, "context 0 belongs to a different group");
        if ((rc = vcs_group_create(ctxs, n_ctx, &g))) return rc;
    }
    const size_t F = ctxs[0]->F;
    const int n_p = n_frames - 1;
    if (g->n_p != n_p && (rc = vcs_group_layout(g, n_p, 0))) return rc;
    if ((rc = vcs_group_set_ref(g, frames))) return rc;
    if (n_p && (rc = vcs_group_upload(g, 0, n_p, frames + F))) return rc;
    if ((rc = vcs_group_encode(g))) return rc;
    if ((rc = vcs_group_get_mv(g, mv_out, nullptr))) return rc;
    if (planes_out) {
        const size_t pb = planes_bytes(ctxs[0]);
        for (auto &p : g->per)
            for (int k = 0; k < p.count; k++)
                if ((rc = vcs_gop_get_planes(p.c, g->set * p.count + k, (char *)planes_out + (size_t)(p.first + k) * pb)))
                    return gfail(g, rc, "%s", p.c->err.c_str());
    }
    return VCS_OK;
}

// ==========================================================================
// Pipelined single-frame feeder: Encoder.encode_frame per captured frame
// (main.py:34-43, encoder.py:21-36) without a host synchronisation per frame.  K staging
// slots; frame k's upload runs on a copy stream while frame k-1's kernels run on the
// context's stream and frame k-2's vectors are copied back on a third.
// ==========================================================================
int vcs_async_reserve(vcs_ctx *c, int n_slots) {
    int rc = use(c);
    if (rc) return rc;
    if (n_slots < 1 || n_slots > 64) return fail(c, VCS_E_ARG, "bad slot count %d", n_slots);
    if ((rc = need_dct_geometry(c))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_async(c);
    HIP_TRY(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
    const size_t nmv = (size_t)(c->g.N > 0 ? c->g.N : 1);
    c->aslots.resize(n_slots);
    for (auto &s : c->aslots) {
        HIP_TRY(c, hipMalloc(&s.d_cur, c->Fs));
        HIP_TRY(c, hipMemset(s.d_cur, 0, c->Fs));
        HIP_TRY(c, hipMalloc(&s.d_mv, nmv * 4 + 16));
        HIP_TRY(c, hipHostMalloc(&s.h_mv, nmv * 4 + 16, hipHostMallocDefault));
        HIP_TRY(c, hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&s.ev_kern, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming));
    }
    return VCS_OK;
}

int vcs_encode_p_async(vcs_ctx *c, int slot, const uint8_t *cur, vcs_planes **handle_out) {
    int rc = use(c);
    if (rc) return rc;
    if (slot < 0 || slot >= (int)c->aslots.size())
        return fail(c, VCS_E_ARG, "slot %d outside the %zu reserved (vcs_async_reserve)", slot, c->aslots.size());
    if (!cur || !handle_out) return fail(c, VCS_E_ARG, "null pointer");
    *handle_out = nullptr;
    if (!c->ref_valid) return fail(c, VCS_E_STATE, "no reference frame set (vcs_set_ref)");
    auto &s = c->aslots[slot];
    if (s.busy) return fail(c, VCS_E_STATE, "slot %d is still in flight (vcs_wait first)", slot);
    vcs_planes *h = nullptr;
    if ((rc = planes_alloc(c, &h))) return rc;
    const size_t N = c->g.N;
    hipError_t e = hipMemcpyAsync(s.d_cur, cur, c->F, hipMemcpyHostToDevice, c->s_in);
    if (e == hipSuccess) e = hipEventRecord(s.ev_in, c->s_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, s.ev_in, 0);
    if (e != hipSuccess) rc = fail(c, VCS_E_HIP, "upload: %s", hipGetErrorString(e));
    uint32_t *d_ns = (uint32_t *)(s.d_mv + (N ? N : 1) * 2);  // the count rides behind the vectors
    if (!rc) rc = launch_me(c, c->d_ref, s.d_cur, c->Fs, 1, s.d_mv, d_ns, nullptr);
    if (!rc)
        rc = N ? launch_fdct<1>(c, VCS_T_ENC_TAIL, s.d_cur, c->Fs, c->d_ref, s.d_mv, 1, h->d, 0)
               : launch_fdct<0>(c, VCS_T_ENC_TAIL, s.d_cur, c->Fs, nullptr, nullptr, 1, h->d, 0);
    if (!rc) {
        e = hipEventRecord(s.ev_kern, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->s_out, s.ev_kern, 0);
        if (e == hipSuccess)
            e = hipMemcpyAsync(s.h_mv, s.d_mv, (N ? N : 1) * 4 + 4, hipMemcpyDeviceToHost, c->s_out);
        if (e == hipSuccess) e = hipEventRecord(s.ev_out, c->s_out);
        // (the next upload into this slot's d_cur cannot overtake these kernels: the slot is
        // only handed out again after vcs_wait, i.e. after ev_out, which follows ev_kern)
        if (e != hipSuccess) rc = fail(c, VCS_E_HIP, "enqueue: %s", hipGetErrorString(e));
    }
    if (rc) {
        hipStreamSynchronize(c->stream);
        vcs_store_free(c, h);
        return rc;
    }
    s.busy = true;
    *handle_out = h;
    return VCS_OK;
}

int vcs_wait(vcs_ctx *c, int slot, int16_t *mv_out, uint32_t *n_static) {
    int rc = use(c);
    if (rc) return rc;
    if (slot < 0 || slot >= (int)c->aslots.size()) return fail(c, VCS_E_ARG, "bad slot %d", slot);
    auto &s = c->aslots[slot];
    if (!s.busy) return fail(c, VCS_E_STATE, "slot %d has nothing in flight", slot);
    s.busy = false;   // whatever happens, the slot is handed back
    HIP_TRY(c, hipEventSynchronize(s.ev_out));
    const size_t N = c->g.N;
    if (mv_out && N) memcpy(mv_out, s.h_mv, N * 4);
    if (n_static) memcpy(n_static, s.h_mv + (N ? N : 1) * 2, 4);
    return VCS_OK;
}

}  // extern "C"
