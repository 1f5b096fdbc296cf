/*
 * vcs_hip.h -- C ABI of libvcs_hip.so: the MI355X (gfx950) implementation of
 * the inter-frame encode hot path of miatang13/VCS-h264.
 *
 * The reference is pure Python and has no FFI layer; its boundary for this
 * path is the class API used by InterframeCompression/main.py, encoder.py and
 * decoder.py.  The drop-in modules in vcs-h264_amd/InterframeCompression/
 * keep that class API and forward the work to the entry points below through
 * ctypes.  Each entry point names the reference interface it replaces
 * (file:line under /root/reference/InterframeCompression/).
 *
 * Conventions
 *   - every function returns 0 on success or a negative VCS_E* code; nothing
 *     throws across the boundary; vcs_last_error() gives the message.
 *   - frames are H x W x 3 uint8, BGR interleaved, C-contiguous (what
 *     cv2.VideoCapture.read returns, main.py:35).  F = 3*H*W bytes.
 *   - N = (H / me_bs) * (W / me_bs) motion blocks in raster order
 *     (motion.py:74-98); motion vectors are N x [dx, dy] int16.
 *   - coefficient planes are 3 planar H x W arrays [Y, Cr, Cb]
 *     (DCTcompressor.py:73), element type per `coeff_mode`.
 *   - host pointers are caller-owned; a context serialises its own calls
 *     (one HIP stream per context); distinct contexts may be used from
 *     distinct threads.  There is NO CPU fallback: without a usable GPU
 *     vcs_ctx_create fails.
 */
#ifndef VCS_HIP_H
#define VCS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCS_ABI_VERSION 2

enum {
    VCS_OK = 0,
    VCS_E_ARG = -1,     /* bad argument / unsupported geometry              */
    VCS_E_HIP = -2,     /* a HIP runtime call failed                        */
    VCS_E_NOMEM = -3,   /* allocation failed                                */
    VCS_E_STATE = -4,   /* call sequence error (e.g. no reference set)      */
    VCS_E_RANGE = -5,   /* a motion vector / block would read out of frame  */
    VCS_E_COMM = -6     /* an RCCL call failed (multi-GPU group)             */
};

enum {                  /* coefficient plane element type                   */
    VCS_COEFF_F64 = 0,  /* float64, un-rounded: DCTcompressor.py:71 as is   */
    VCS_COEFF_F32 = 1,  /* the same value rounded once to float32           */
    VCS_COEFF_I16 = 2   /* np.round(value) as int16 (DCTCompression/dct.py:179) */
};

enum {                  /* indices into vcs_timing()                        */
    VCS_T_ME = 0,       /* motion search kernel                             */
    VCS_T_ENC_TAIL = 1, /* MC + residual + colour + DCT + quantise kernel   */
    VCS_T_DEC = 2,      /* dequantise + IDCT + colour + MC-add kernel       */
    VCS_T_AUX = 3,      /* stand-alone mc / residual / compress kernels     */
    VCS_T_ME_PREP = 4,  /* bit-plane search: current blocks -> planes + static test (k_bits_cur) */
    VCS_T_COUNT = 5
};

typedef struct vcs_ctx vcs_ctx;

/* ---- library / device -------------------------------------------------- */
int vcs_abi_version(void);
/* Identity of the build: the first 16 hex digits of sha256 over the kernel and
 * ABI sources (csrc/ Makefile order) and the extra compiler flags, stamped by
 * the Makefile.  profiles/hbm_traffic.json carries the id of the library its
 * counters were collected with; bench.py reports `traffic` only when the two
 * agree (no reference counterpart: measurement plumbing).                    */
const char *vcs_build_id(void);
/* 1 if the library was built with any VCS_EXP_* experiment switch (timing
 * probes, variants that compute wrong results on purpose), else 0.           */
int vcs_experiment_build(void);
int vcs_device_count(int *n);
/* Page-locked host memory for frames and planes (hipHostMalloc): buffers handed to the
 * entry points below are copied by DMA at PCIe rate when they live in such memory and
 * through the runtime's staging path otherwise (a 1080p frame: 0.12 vs 0.5 ms).  The
 * frame feeder of SURVEY 8f N3: the producer (video decoder) writes into these.        */
int vcs_host_alloc(size_t bytes, void **ptr);
int vcs_host_free(void *ptr);
/* Message of the last failure on `ctx` (or, with ctx == NULL, of the last
 * failed vcs_ctx_create / vcs_device_count on this thread).                 */
const char *vcs_last_error(const vcs_ctx *ctx);

/* ---- context ----------------------------------------------------------- */
/* MotionProcessor.__init__ (motion.py:15-18) + DCTCompressor.__init__
 * (DCTcompressor.py:43-46).  me_bs: motion block size (any >= 2);
 * dct_bs must be 8 (the reference's Q tables are 8x8).  Defaults installed:
 * win = 2*me_bs (motion.py:18), step = round(me_bs/3) (motion.py:132),
 * sim_thresh = 2000 (motion.py:8), Q from QF = 50 (DCTcompressor.py:29-38),
 * coeff_mode = VCS_COEFF_F64.                                               */
int vcs_ctx_create(int device, int H, int W, int me_bs, int dct_bs, vcs_ctx **out);
int vcs_ctx_destroy(vcs_ctx *ctx);

/* Override the search / quantiser parameters.  win, step <= 0, sim_thresh ==
 * VCS_KEEP_THRESH, Q3x64 == NULL and coeff_mode == VCS_KEEP keep the current
 * value (so a caller that only pushes Q never resets motion.py:8's threshold).
 * sim_thresh < 0 (other than the sentinel): no block is ever static.
 * Q3x64: 3 x 8 x 8 float64, row-major, every entry finite and != 0 (the `Q`
 * list of DCTcompressor.py:36-38).                                          */
#define VCS_KEEP (-1)
#define VCS_KEEP_THRESH INT64_MIN
int vcs_set_params(vcs_ctx *ctx, int win, int step, int64_t sim_thresh,
                   const double *Q3x64, int coeff_mode);
/* Searches and costs that are NOT in the reference (default: the reference's
 * own, and only those run on the specialised kernels).  VCS_SEARCH_TSS: the classic three-
 * step / logarithmic search from the co-located position (steps 16, 8, 4, 2, 1 for +-32;
 * a stage moves to the cheapest valid neighbour only if it is strictly cheaper than the
 * centre, ties between neighbours in raster order).  VCS_COST_SAD: sum |r - c| instead of
 * the reference's wrap-around sum (motion.py:146).  The static early-out is unchanged.  No
 * reference output exists for these modes (parity unpinned; oracle: vcso_me_ext).  VCS_KEEP
 * keeps a selector.                                                                       */
enum { VCS_SEARCH_GRID = 0, VCS_SEARCH_TSS = 1 };
enum { VCS_COST_WRAP = 0, VCS_COST_SAD = 1 };
int vcs_set_search(vcs_ctx *ctx, int search, int cost);
/* Fills Q3x64 by the reference's QF rule (DCTcompressor.py:29-38).          */
int vcs_make_q(double qf, double *Q3x64);
/* The 8x8 DCT-II matrix of DCTcompressor.py:124-133, as the kernels use it. */
int vcs_dct_matrix(double *D8x8);

/* ---- single-frame entry points (host buffers) -------------------------- */
/* Encoder._process_I_frame (encoder.py:41-43): the raw I frame becomes the
 * reference of its GOP; copied to HBM once and kept resident.               */
int vcs_set_ref(vcs_ctx *ctx, const uint8_t *ref_bgr);

/* MotionProcessor.process_motion_prediction (motion.py:20-36, incl.
 * _find_match :100-154 and _get_motion_vector :156-161).
 * n_static (nullable): blocks that took the static early-out (:113).
 * cost_out (nullable, N x uint64): winning cost per block (debug/parity).   */
int vcs_me(vcs_ctx *ctx, const uint8_t *cur_bgr, int16_t *mv_out,
           uint32_t *n_static, uint64_t *cost_out);

/* MotionProcessor.reconstruct_from_motion_vectors (motion.py:42-69).
 * coords: n x [x, y] int32, or NULL for the raster grid of motion.py:74-98
 * (then n must be N).  Every source and destination block is bounds-checked
 * on the host first (VCS_E_RANGE).  pred_out: F bytes, zero outside blocks. */
int vcs_mc(vcs_ctx *ctx, const int16_t *mv, const int32_t *coords, int n,
           uint8_t *pred_out);

/* MotionProcessor.get_residuals (motion.py:38-40): a - b, uint8 wrap.
 * sign = +1 gives a + b (decoder.py:57,60).  n bytes each.                  */
int vcs_residual(vcs_ctx *ctx, const uint8_t *a, const uint8_t *b, int sign,
                 size_t n, uint8_t *out);

/* DCTCompressor.compress (DCTcompressor.py:49-74).  H and W must be
 * multiples of 8 (cv2.resize at :52-53 is then the identity).               */
int vcs_compress(vcs_ctx *ctx, const uint8_t *bgr, void *planes_out);
/* DCTCompressor.decompress (DCTcompressor.py:76-93).                        */
int vcs_decompress(vcs_ctx *ctx, const void *planes_in, uint8_t *bgr_out);

/* Encoder._process_P_frame (encoder.py:49-70), fused: one upload, ME kernel,
 * then one MC + residual + colour + DCT + quantise kernel.  planes_out /
 * res_out nullable (res_out: the uint8 residual, i.e. with_DCT = False).    */
int vcs_encode_p(vcs_ctx *ctx, const uint8_t *cur_bgr, int16_t *mv_out,
                 void *planes_out, uint8_t *res_out, uint32_t *n_static);
/* Decoder._reconstruct_P_frame + _fully_reconstruct (decoder.py:52-69).
 * planes == NULL: motion compensation only (with_residuals = False).        */
int vcs_decode_p(vcs_ctx *ctx, const int16_t *mv, const void *planes,
                 uint8_t *bgr_out);

/* ---- HBM-resident coefficient planes (next row N3) ---------------------- */
/* Encoder._process_P_frame / Decoder._reconstruct_P_frame with the planes of
 * Frame.r (encoder.py:66, decoder.py:55) left in HBM behind an opaque handle:
 * only the motion vectors (and, on decode, the frame) cross PCIe; the planes
 * are downloaded on demand.  A handle belongs to the device of its context
 * and must be freed with vcs_store_free.                                    */
typedef struct vcs_planes vcs_planes;
int vcs_store_encode_p(vcs_ctx *ctx, const uint8_t *cur_bgr, int16_t *mv_out,
                       uint32_t *n_static, vcs_planes **handle_out);
int vcs_store_planes(vcs_ctx *ctx, const vcs_planes *h, void *planes_out);
int vcs_store_decode_p(vcs_ctx *ctx, const int16_t *mv, const vcs_planes *h,
                       uint8_t *bgr_out);
int vcs_store_free(vcs_ctx *ctx, vcs_planes *h);

/* ---- pipelined single-frame feeder (next row N3) -------------------------- */
/* main.py:34-43 calls Encoder.encode_frame once per captured frame.  These entry
 * points do what vcs_store_encode_p does without a host synchronisation per
 * frame: K staging slots, the upload of frame k on a copy stream beside the
 * kernels of frame k-1 and the vector download of frame k-2.
 *   vcs_async_reserve   K slots (>= 2 to overlap anything).
 *   vcs_encode_p_async  enqueue one P frame in a free slot; the planes handle is
 *                       valid at once (its contents when the slot completes).
 *                       `cur_bgr` is read until the upload has run: page-locked
 *                       buffers (vcs_host_alloc) must stay untouched until
 *                       vcs_wait(slot); pageable ones are staged by the runtime
 *                       before the call returns.
 *   vcs_wait            block until the slot's frame is done; its motion vectors
 *                       and static count.  The slot is free again afterwards.
 * vcs_set_ref between frames is allowed (it waits for the frames in flight).   */
int vcs_async_reserve(vcs_ctx *ctx, int n_slots);
int vcs_encode_p_async(vcs_ctx *ctx, int slot, const uint8_t *cur_bgr, vcs_planes **handle_out);
int vcs_wait(vcs_ctx *ctx, int slot, int16_t *mv_out, uint32_t *n_static);

/* ---- multi-GPU GOP encode, RCCL inside the library ------- */
/* Every P frame depends only on the raw I frame of its GOP (encoder.py:51-52):
 * the P frames are dealt to the shards (one context per GPU) in contiguous
 * chunks, chunk sizes differ by at most one, earlier shards take the larger
 * ones (vcs_shard_frames).  Exchanges: ncclBroadcast of the I frame from shard
 * 0 into every context's resident reference buffer, ncclAllGather of motion
 * vectors and static counts.  Coefficient planes stay on their GPU.
 *
 * vcs_encode_gop: the whole GOP in one call, single process, one context per
 * device.  frames: n_frames x F, frame 0 = the I frame, frames 1.. = its P
 * frames in order (encoder.py:25).  mv_out: (n_frames-1) x N x [dx,dy] int16.
 * planes_out (nullable): (n_frames-1) x 3 x H x W coefficients (downloaded from
 * the GPU that holds them).  The group behind it (communicators, streams) is
 * created on the first call and lives until one of the contexts is destroyed.
 * Contexts on pairwise distinct devices use RCCL; contexts that all share ONE
 * device (RCCL refuses duplicate devices) run the same schedule with
 * device-to-device copies -- the sharding / gather logic on a one-GPU box.    */
int vcs_shard_frames(int n_frames, int world, int rank, int *first, int *count);
int vcs_encode_gop(vcs_ctx *const *ctxs, int n_ctx, const uint8_t *frames, int n_frames,
                   int16_t *mv_out, void *planes_out);

/* The pieces of vcs_encode_gop, for callers that keep GOPs resident, pipeline
 * them (bench.py) or run one process per GPU.
 *   vcs_group_create       single process: context i is shard i of n_ctx.
 *   vcs_comm_unique_id + vcs_group_create_rank
 *                          one process per GPU (ncclCommInitRank): rank 0 makes
 *                          the 128-byte id, every rank passes the same bytes.
 *   vcs_group_layout       shard table for a GOP of n_p_frames P frames; reserves
 *                          the contexts' GOP slots (vcs_gop_reserve).
 *   vcs_group_upload       P frames [first_p, first_p+n) of the GOP from host
 *                          memory; each local shard takes its part of the range.
 *   vcs_group_set_ref      stages the next GOP's I frame on shard 0's GPU (no-op in
 *                          processes that do not hold shard 0).  It supersedes a
 *                          prefetch that was enqueued before it.  With one process
 *                          per GPU EVERY rank makes the same sequence of set_ref /
 *                          prefetch_ref / encode calls (ref_bgr may be NULL off
 *                          shard 0): the calls carry collectives.
 *   vcs_group_prefetch_ref enqueues the broadcast of the staged I frame into the
 *                          contexts' ALTERNATE reference buffers, on the
 *                          collectives' stream -- beside the kernels of the GOP
 *                          being encoded.  Optional: vcs_group_encode does it
 *                          itself when nothing was prefetched.
 *   vcs_group_encode       asynchronous: makes the prefetched buffer current,
 *                          motion search + encoder tail over the shard's frames,
 *                          copies the vectors aside and all-gathers them on the
 *                          collectives' stream.
 *   vcs_group_get_mv       waits, then all P frames' vectors / static counts in
 *                          GOP order (from the gathered copy on local shard 0).
 *   vcs_group_max_f64      max over all processes of *v (and a barrier).
 * Planes / decoded frames of local shard i: vcs_gop_get_planes(ctx_i, slot) with
 * slot = P-frame index - first (vcs_group_shard).
 * VCS_GROUP_SERIAL=1: collectives on the kernels' stream, strictly in sequence. */
typedef struct vcs_group vcs_group;
#define VCS_UNIQUE_ID_BYTES 128
int vcs_comm_unique_id(void *id, size_t bytes);
int vcs_group_create(vcs_ctx *const *ctxs, int n_ctx, vcs_group **out);
int vcs_group_create_rank(vcs_ctx *ctx, const void *unique_id, int world, int rank,
                          vcs_group **out);
int vcs_group_destroy(vcs_group *g);
const char *vcs_group_last_error(const vcs_group *g);
/* transport: 1 = RCCL overlapped, 2 = RCCL serial, 3 = same-device copies overlapped, 4 = serial */
int vcs_group_info(vcs_group *g, int *world, int *rank0, int *n_local, int *transport);
int vcs_group_layout(vcs_group *g, int n_p_frames, int with_decode);
/* n_sets resident copies of the GOP's slots per shard (set k of a shard with `count`
 * frames = context slots [k*count, (k+1)*count)); vcs_group_use_set picks the one that
 * upload / encode / decode work on (bench.py rotates through them so that consecutive
 * steps find nothing of their input in the Infinity Cache).                        */
int vcs_group_layout_sets(vcs_group *g, int n_p_frames, int with_decode, int n_sets);
int vcs_group_use_set(vcs_group *g, int set);
int vcs_group_shard(vcs_group *g, int local_idx, int *first, int *count);
int vcs_group_upload(vcs_group *g, int first_p, int n, const uint8_t *frames);
int vcs_group_set_ref(vcs_group *g, const uint8_t *ref_bgr);
int vcs_group_prefetch_ref(vcs_group *g);
int vcs_group_encode(vcs_group *g);
int vcs_group_decode(vcs_group *g);
int vcs_group_sync(vcs_group *g);
int vcs_group_get_mv(vcs_group *g, int16_t *mv_out, uint32_t *n_static_out);
int vcs_group_max_f64(vcs_group *g, double *v);
int vcs_group_barrier(vcs_group *g);

/* ---- batched, HBM-resident GOP path ------------------------------------ */
/* All P frames of a GOP depend only on the raw I frame (encoder.py:51-52), so
 * a whole GOP is one launch chain.  Slots 0..n-1 hold P frames; outputs stay
 * in HBM until downloaded.  vcs_gop_encode / vcs_gop_decode are asynchronous
 * on the context's stream; vcs_sync() waits.                                */
int vcs_gop_reserve(vcs_ctx *ctx, int n_slots, int with_decode);
int vcs_gop_upload(vcs_ctx *ctx, int first_slot, int n, const uint8_t *frames);
int vcs_gop_encode(vcs_ctx *ctx, int first_slot, int n);
int vcs_gop_decode(vcs_ctx *ctx, int first_slot, int n);
int vcs_gop_get_mv(vcs_ctx *ctx, int first_slot, int n, int16_t *mv_out,
                   uint32_t *n_static_out /* n entries, nullable */);
int vcs_gop_get_planes(vcs_ctx *ctx, int slot, void *planes_out);
int vcs_gop_get_decoded(vcs_ctx *ctx, int slot, uint8_t *bgr_out);
int vcs_sync(vcs_ctx *ctx);
/* How vcs_gop_encode brings the batch's frames into the Infinity Cache for the
 * encoder tail (the search kernels read them as scattered pieces that do not
 * stay there): 1 = by the last workgroups of the search launch (default),
 * 2 = a separate pass between search and tail, 0 = not at all, -1 = back to
 * the default / VCS_L3_PREFETCH.  Scheduling only -- results never change;
 * bench.py uses 0 to time the search kernel without its prefetch workgroups.
 * (No reference counterpart: encoder.py:58-66 runs on the CPU.)               */
int vcs_set_l3_prefetch(vcs_ctx *ctx, int mode);

/* Device-side plumbing for the multi-GPU host layer (RCCL broadcast of the I
 * frame, gather of motion vectors): raw HBM pointers of the context's
 * resident buffers, valid until the context is destroyed / re-reserved.
 * which: 0 = reference frame (F bytes), 1 = GOP motion vectors
 * (n_slots*N*2 int16), 2 = GOP static counts (n_slots uint32), 3 = the
 * alternate reference buffer (F bytes; allocated by the first request).      */
int vcs_dev_ptr(vcs_ctx *ctx, int which, void **ptr, size_t *bytes);
/* Marks the resident reference as valid after it was written through
 * vcs_dev_ptr(ctx, 0, ...) by a collective.                                 */
int vcs_ref_written(vcs_ctx *ctx);
/* Exchanges the current and the alternate reference buffer (pointer swap on
 * the host, no GPU work) and marks the new current one as written: the next
 * GOP's I frame can be broadcast into the alternate buffer while this GOP is
 * still being encoded against the current one.  Work already enqueued keeps
 * the buffer it was launched with.  After a swap, vcs_dev_ptr(0 / 3) return
 * the exchanged pointers.                                                    */
int vcs_ref_swap(vcs_ctx *ctx);
/* Declares the ALTERNATE reference buffer complete: it holds the next GOP's I
 * frame (encoder.py:41-43: the frame a later vcs_ref_swap makes current), as
 * of the work enqueued so far on `written_on_stream` (a hipStream_t; null =
 * the context's own stream order, e.g. after a copy the caller enqueued behind
 * vcs_stream_wait).  From then on the per-reference tables of the bit-plane
 * search (what k_bits_ref makes) are cut on a side stream BESIDE the search
 * kernel of the GOP encoded next, instead of in front of the following GOP's
 * search; the swap finds them ready.  Purely a scheduling hint: results are the
 * same with or without it.                                                   */
int vcs_ref_next_ready(vcs_ctx *ctx, void *written_on_stream);
/* vcs_set_ref into the alternate buffer + vcs_ref_next_ready: the host-side
 * form for a caller that knows its next I frame early (main.py:34-43 reads
 * frames in order: frame n + gop_length is known while GOP n is encoded).     */
int vcs_set_ref_next(vcs_ctx *ctx, const uint8_t *ref_bgr);

/* ---- intra prediction (next row N1; BASELINE config 5) ------------------ */
/* IntraframeCompression/intraframe.py: luma4x4 (:24-151), luma16x16 (:153-225)
 * and chroma8x8 (:228-317) with the predictors of intramodes.py.  Planes are
 * H x W uint8 with H, W the context's frame size (multiples of 4 / 16 / 8);
 * outputs are float64 like the reference's: residual, prediction (H x W) and
 * the mode map ((H/bs) x (W/bs)).                                            */
int vcs_intra_luma4x4(vcs_ctx *ctx, const uint8_t *Y, double *res, double *pred, double *modes);
int vcs_intra_luma16x16(vcs_ctx *ctx, const uint8_t *Y, double *res, double *pred, double *modes);
int vcs_intra_chroma8x8(vcs_ctx *ctx, const uint8_t *Cr, const uint8_t *Cb, double *crres,
                        double *crpred, double *cbres, double *cbpred, double *modes);

/* The 4x4 integer core transform the reference keeps commented out behind its intra
 * prediction (intraframe.py:14-17, :352-358): out = np.round(C . X . C^T) per 4x4 block
 * of a float64 H x W residual plane (H, W multiples of 4).  No running upstream code
 * pins it; exact for integer-valued input.                                      */
int vcs_intra_transform4x4(vcs_ctx *ctx, const double *res, double *out);
/* The same four kernels over a BATCH of I frames resident in HBM (the I frames of
 * many GOPs: intraframe.py processes one image per run, :321-350).  A single
 * 1080p frame cannot fill the chip -- chroma8x8 is serial down each of its 240
 * block columns because Cb is predicted from the residual above (:266) --, n
 * frames give n times the independent columns.  reserve: device buffers for
 * n_frames (planes must be multiples of 16); upload: Y / Cr / Cb as n x H x W
 * uint8 each; run: `what` is a mask of 1 = luma4x4, 2 = luma16x16, 4 =
 * chroma8x8, 8 = the 4x4 transform of the luma4x4 residual (needs 1), one launch
 * per kernel over frames [first, first + n), asynchronous; get: one frame's
 * results (`which`: 0 luma4x4 -> res, pred, modes; 1 luma16x16 -> the same;
 * 2 chroma8x8 -> crres, crpred, cbres, cbpred, modes; 3 transform -> out),
 * null pointers are skipped.  Results are identical to the single-frame calls. */
int vcs_intra_batch_reserve(vcs_ctx *ctx, int n_frames);
int vcs_intra_batch_upload(vcs_ctx *ctx, int first, int n, const uint8_t *Y, const uint8_t *Cr, const uint8_t *Cb);
int vcs_intra_batch_run(vcs_ctx *ctx, int first, int n, int what);
int vcs_intra_batch_get(vcs_ctx *ctx, int frame, int which, double *p0, double *p1, double *p2, double *p3,
                        double *modes);

/* 4:2:0 chroma subsampling (next row N4; reference ChromaSubsampling/chroma.py:8-21, a
 * run-at-import script upstream): BGR -> YCrCb, cv2.boxFilter(ksize=(2,2)) over Cr and Cb,
 * samples at even rows / columns.  Y (nullable): H x W; cr_s / cb_s: ((H+1)/2) x ((W+1)/2).
 * OpenCV's box filter is third-party arithmetic restated here: parity unpinned.          */
int vcs_chroma420(vcs_ctx *ctx, const uint8_t *bgr, uint8_t *Y, uint8_t *cr_s, uint8_t *cb_s);

/* GPU-side ordering against a foreign HIP stream (the stream torch / RCCL
 * collectives run on), without host synchronisation:
 *   vcs_stream_wait:   work enqueued on the context's stream after this call
 *                      waits for everything enqueued on `foreign_stream` so far;
 *   vcs_stream_signal: work enqueued on `foreign_stream` after this call waits
 *                      for everything enqueued on the context's stream so far.
 * `foreign_stream` is a hipStream_t (NULL = the legacy default stream).        */
int vcs_stream_wait(vcs_ctx *ctx, void *foreign_stream);
int vcs_stream_signal(vcs_ctx *ctx, void *foreign_stream);
/* The context's own hipStream_t, for callers that order against it with their own
 * events or enqueue device-to-device copies on it (bench.py wraps it in a
 * torch.cuda.ExternalStream to overlap RCCL collectives with the kernels).  The
 * stream stays owned by the context and dies with it.                          */
int vcs_stream_handle(vcs_ctx *ctx, void **stream);

/* ---- measurement ------------------------------------------------------- */
/* With timing on, every kernel launch is bracketed by HIP events on the
 * context's stream; vcs_timing() synchronises and returns, per VCS_T_* slot,
 * the accumulated kernel milliseconds and launch counts since the last
 * vcs_timing_reset().                                                       */
int vcs_timing_enable(vcs_ctx *ctx, int on);
int vcs_timing_reset(vcs_ctx *ctx);
int vcs_timing(vcs_ctx *ctx, double *ms_out, uint64_t *launches_out, int n);
/* With timing on, the shared-window search kernel (k_me16g) records the longest lifetime of
 * any of its workgroups in shader-clock cycles (s_memtime), maximum over the launches since
 * the last reset: divided by that launch's duration it is the clock the chip held under
 * this load (bench.py prices the kernel's vector-issue utilisation with it).            */
int vcs_me_cycles(vcs_ctx *ctx, uint64_t *cycles, int reset);
/* With timing on, the bit-plane search kernel (k_me16b) sums, over its workgroups, the shader-clock
 * cycles (s_memtime) and the 100 MHz ticks (s_memrealtime) each of them lived: *ghz is their quotient,
 * the clock the chip held under this load since the last reset (0 if the kernel has not run).        */
int vcs_me_clock(vcs_ctx *ctx, double *ghz, int reset);
/* Bytes/s of a device-to-device copy of `bytes` (the practical HBM roof).   */
int vcs_measure_copy_bw(vcs_ctx *ctx, size_t bytes, int iters, double *gbps_out);
/* Bytes/s of a device memset of `bytes` (the practical HBM write-only roof).  */
int vcs_measure_fill_bw(vcs_ctx *ctx, size_t bytes, int iters, double *gbps_out);

#ifdef __cplusplus
}
#endif
#endif /* VCS_HIP_H */
