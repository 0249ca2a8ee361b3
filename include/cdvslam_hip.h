/*
 * cdvslam_hip.h -- C ABI of libcdvslam_hip.so: the MI355X (gfx950) implementation of CDV-SLAM's
 * per-frame update hot path (altcorr -> projective_ops -> fastba).
 *
 * Drop-in boundary: these entry points are what the reference's three pybind11 torch extensions
 * would bind for this path (reference paths relative to the reference root):
 *   cuda_corr          cdvslam/altcorr/correlation.cpp:57-63   (forward / patchify_forward)
 *   cuda_ba            cdvslam/fastba/ba.cpp:183-188           (forward / neighbors / reproject)
 *   lietorch_backends  cdvslam/lietorch/src/lietorch.cpp:286-316 (expm, logm, inv, mul, adj, adjT,
 *                                                                act, act4, as_matrix)
 * plus fused forms the reference composes in Python (SLAM.corr, slam.py:316-323; pops.transform,
 * projective_ops.py:53-113).  See INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes; no torch types.  All pointers are device memory of the
 *     current HIP device unless the parameter name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every call only ENQUEUES
 *     work on that stream; no call synchronises or allocates (hipGraph-capturable), except
 *     cdv_graph_read_meta_host / cdv_ba_status, which are explicit device->host read-backs.  Only KERNELS are enqueued
 *     (no hipMemsetAsync: a captured memset is not replayed like the eager one on ROCm 7.2), and nothing of the host but
 *     pointers, sizes and the bundle adjustment's hand-off tokens is fixed into a launch when a call is enqueued -- the
 *     hand-off words are re-armed by each iteration's first launch, so replays do not meet their own tags, and the table
 *     build's generation is a device word.  A captured sequence of the static entry points is valid for the SHAPES it was
 *     captured with (launch geometry follows the number of edges) and may be replayed over other edge lists of that shape;
 *     the *_dyn entry points take even the sizes from the device, so a captured frame pair is good for a whole stream.
 *   - return value: 0 = success, < 0 = error (CDV_ERR_*); cdv_last_error() gives the message.
 *     Never calls exit() (the reference does, block_e.cu:20-27, ba.cpp:151-152).
 *   - index tensors are int64 (torch.long) exactly as the reference passes them.
 *   - float layouts are the reference's contiguous torch layouts unless stated otherwise.
 */
#ifndef CDVSLAM_HIP_H
#define CDVSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDV_OK 0
#define CDV_ERR_HIP (-1)         /* a HIP runtime call failed */
#define CDV_ERR_ARG (-2)         /* invalid argument / unsupported shape */
#define CDV_ERR_WORKSPACE (-3)   /* workspace too small */
#define CDV_ERR_UNSUPPORTED (-4) /* group / dtype / configuration not implemented */
/* returned by cdv_ba_status only (the enqueueing calls cannot know): what went wrong inside the last bundle adjustment */
#define CDV_ERR_BA_NOT_SPD (-5)  /* a Cholesky pivot was not positive: the update is garbage / NaN (the reference ignores
                                    cholesky_ex's info, ba_cuda.cu:576,590) */
#define CDV_ERR_BA_OVERFLOW (-6) /* more unique patches than U_max: the update was SKIPPED */
#define CDV_ERR_BA_HANDOFF (-7)  /* an in-launch hand-off timed out: the update was NOT applied */
#define CDV_ERR_GRAPH_RANGE (-8) /* the patch-graph index had reported a patch-id range beyond k_range: update skipped */

#define CDV_F16 0
#define CDV_F32 1
#define CDV_F64 2

const char* cdv_last_error(void);
/* "gfx950" + build info; also a cheap symbol to probe that the library loaded */
const char* cdv_version(void);
/* Workspaces (cdv_graph_*, cdv_ba_forward) are initialised by the library the first time it sees their address and kept
 * consistent between calls.  Call this for an address that was freed and may have been written by someone else before
 * it is used as a workspace again (allocators hand addresses back); the next call re-initialises it. */
void cdv_workspace_forget(const void* ws);
/* The explicit form: whoever allocates a workspace initialises it, and nothing depends on what the library remembers about
 * an address.  cdv_graph_workspace_init zeroes on `stream` what the index builds keep zero between calls and records the
 * layout (a bound correlation stream is dropped); cdv_ba_workspace_init makes the next cdv_ba_forward on the workspace
 * initialise it (status counters stay bound).  cdv_workspace_forget(ws) stays as the old spelling of "treat as new at
 * the next use". */
int cdv_graph_workspace_init(void* graph_ws, size_t ws_bytes, int64_t E_max, int64_t k_range, void* stream);
int cdv_ba_workspace_init(void* ba_ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * altcorr  (replaces cuda_corr.forward / patchify_forward)
 * ---------------------------------------------------------------------------------------------- */

/*
 * cuda_corr.forward(fmap1, fmap2, coords, ii, jj, radius)  -- correlation.cpp:35-42,
 * correlation_kernel.cu:82-136 (kernel) + :213-232 (bilinear blend + permute), one pyramid level.
 *   fmap1  [N1][C][P][P]     patch feature tiles, channel-planar (reference layout)
 *   fmap2  [N2][C][H2][W2]   frame feature maps, channel-planar (reference layout)
 *   coords [M][2][P][P] f32 ; us[M] index into fmap1 ; vs[M] index into fmap2
 *   out    [M][D-1 (x)][D-1 (y)][P][P], D = 2*radius+2 -- the logical layout of the tensor the
 *          reference returns (its permute(0,1,3,2,4,5)), stored contiguously.
 * dtype CDV_F16: f16 in/out, f32 accumulate (the reference accumulates in f16; tolerance in
 * DESIGN.md).  dtype CDV_F32: f32 throughout.  Any C, P, radius.  Generic (non-MFMA) kernel.
 */
int cdv_corr_fwd(const void* fmap1, const void* fmap2, const float* coords, const int64_t* us,
                 const int64_t* vs, void* out, int64_t M, int64_t N1, int64_t N2, int C, int P, int H2, int W2,
                 int radius, int dtype, void* stream);

/*
 * HBM layout of the feature rings the fused correlation gathers from -- "padded channels-last":
 *     [slot][H + 2*CDV_FMAP_PADY][W + 2*CDV_FMAP_PADX][C] f16, zero margins
 * (one pixel = C contiguous halves; the margins make every window load in-bounds and supply the zeros of
 * the reference's out-of-bounds rule, correlation_kernel.cu:122).  The buffer must be zero-initialised
 * once (cdv_fmap_padded_elems() halves); the kernels below only write the interior.
 */
#define CDV_FMAP_PADX 16
#define CDV_FMAP_PADY 12
size_t cdv_fmap_padded_elems(int64_t slots, int C, int H, int W);

/*
 * Channel-planar [N][C][H][W] f16 -> padded channels-last.  `first`/`count` select a slot range so a
 * ring buffer can be refreshed one frame at a time (slam.py:679-682 writes one slot).
 */
int cdv_fmap_to_nhwc(const void* src_nchw, void* dst_nhwc, int64_t N, int C, int H, int W, int64_t first,
                     int64_t count, void* stream);

/*
 * Keep a padded channels-last shadow in step with a planar ring that SOMEBODY ELSE writes (the unchanged reference writes
 * fmap1_[:, n % mem] = fmap with torch ops, slam.py:679-682, and hands the whole ring to cuda_corr.forward): pass 1
 * fingerprints every slot of the planar ring (16 position-keyed 64-bit sums per slot: one read of the ring), pass 2
 * converts only the slots whose fingerprint differs from the previous sync's -- normally one -- instead of all N.
 *   ws: cdv_fmap_sync_workspace_bytes(N) bytes, zeroed once by the caller (zero = "no fingerprint yet": the first sync
 *   converts every slot); parity: 0, 1, 0, 1, ... on successive calls for one (ring, shadow) pair.  The int32 behind the
 *   two fingerprint arrays counts converted slots (diagnostics).  A changed slot goes undetected only if all of its 16
 *   64-bit sums collide.
 */
size_t cdv_fmap_sync_workspace_bytes(int64_t N);
int cdv_fmap_sync_nhwc(const void* src_nchw, void* dst_nhwc, int64_t N, int C, int H, int W, void* ws, int parity,
                       void* stream);

/*
 * What an unchanged slam.py needs in step in front of SLAM.corr (slam.py:316-323) -- the shadows of BOTH pyramid levels
 * (slam.py:679-682 writes one slot of each per frame) and the pixel-major shadow of the patch tiles (slam.py:676) -- in TWO
 * launches instead of the five of 2 x cdv_fmap_sync_nhwc + cdv_gmap_to_pixel_major: launch 1 fingerprints every slot of every
 * ring given, launch 2 converts the slots that differ and, next to them, all Ng tiles.  Per ring exactly what
 * cdv_fmap_sync_nhwc does with the same (ws, parity); the caller flips each ring's parity after the call as it does there.
 *   rings [n_rings], n_rings 0 .. 2 (read during the call); gmap_planar [Ng][C][3][3] f16 -> gmap_pm [Ng][9][C], or NULL.
 */
typedef struct cdv_shadow_ring {
  const void* src_nchw;   /* [N][C][H][W] f16, written by somebody else */
  void* dst_nhwc;         /* its padded channels-last shadow (cdv_fmap_padded_elems) */
  void* ws;               /* cdv_fmap_sync_workspace_bytes(N) */
  int64_t N;
  int32_t C, H, W, parity;
} cdv_shadow_ring;
int cdv_shadows_sync(const cdv_shadow_ring* rings, int n_rings, const void* gmap_planar, void* gmap_pm, int64_t Ng, int C_tiles,
                     void* stream);

/*
 * Per-frame ingest of one new feature frame (slam.py:681-682): fmap [C][H][W] f16 (planar) is
 * written into ring slot `slot` of the padded channels-last ring fmap1_nhwc (H x W interior) and its
 * 4x4 average pool into fmap2_nhwc (H/4 x W/4 interior); optionally also into the planar rings the
 * reference keeps (fmap1_nchw / fmap2_nchw may be NULL).
 */
int cdv_fmap_ingest(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, void* fmap1_nchw, void* fmap2_nchw,
                    int slot, int C, int H, int W, void* stream);

/*
 * Patch feature tiles in "pixel-major" layout [Ng][9][C] f16: the 3x3 patch pixels outermost, the C channels of a
 * pixel contiguous -- the MFMA operand of the fused correlation is then ONE 16-byte load per lane instead of eight
 * strided 2-byte gathers from the reference's [Ng][C][3][3] (gmap_, slam.py:71,250-251).
 *   cdv_gmap_to_pixel_major: converts tiles first .. first+count-1 of a planar array (a whole ring, or the M tiles
 *   patchify just produced for the new frame, net_cdv.py:355-374).
 *   cdv_frame_ingest: cdv_fmap_ingest plus that conversion for the new frame's tiles, in the SAME launch
 *   (gmap_planar / gmap_pm may be NULL: then identical to cdv_fmap_ingest).
 */
int cdv_gmap_to_pixel_major(const void* gmap_planar, void* gmap_pm, int64_t Ng, int C, int64_t first, int64_t count,
                            void* stream);
int cdv_frame_ingest(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, void* fmap1_nchw, void* fmap2_nchw,
                     int slot, int C, int H, int W, const void* gmap_planar, void* gmap_pm, int64_t Ng,
                     int64_t gmap_first, int64_t gmap_count, void* stream);

/*
 * Fused multi-level correlation: SLAM.corr (slam.py:316-323) = two cuda_corr.forward calls +
 * torch.stack(..., -1).view(1, E, -1), in ONE launch on MFMA.
 *   gmap        [Ng][C][3][3] f16 planar (view of gmap_, slam.py:250-251), or -- gmap_pixel_major != 0 -- the
 *               [Ng][9][C] layout of cdv_gmap_to_pixel_major
 *   fmapL_nhwc  padded channels-last rings (layout above) with interior H_L x W_L, L = 0 .. nlev-1 ;
 *               coords are divided by scale[L] (1 and 4 in SLAM.corr)
 *   coords      [E][2][3][3] f32
 *   kk, jj      raw graph indices; the kernel applies ii1 = kk % kmod, jj1 = jj % jmod
 *               (slam.py:319-320; pass kmod = jmod = 0 for "no modulus")
 *   order       optional [E] int32 processing order (a permutation of the edge ids), NULL = natural
 *               order.  Output rows are always indexed by edge id.
 *   out         [E][7 (x)][7 (y)][3][3][nlev] f16  == corr.view(E, 882) for nlev = 2
 * Requirements: radius 3, P 3, C % 8 == 0, C <= 128, nlev in {1,2}, pyramid scales powers of two, every ring and the
 * tile array below 4 GB, Ng and slots below 2^31.  An index outside its ring (after the modulus) gives a zero row.
 * C <= 32 (CDV-SLAM: 24) runs the tuned kernel (workgroups dealt to the XCDs in contiguous eighths of the edge list);
 * wider features (DPVO: 128) a simpler one.
 */
int cdv_corr_fused(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const float* coords,
                   const int64_t* kk, const int64_t* jj, const int32_t* order, void* out, int64_t E, int64_t Ng,
                   int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0, float scale1, int nlev,
                   int64_t kmod, int64_t jmod, int gmap_pixel_major, void* stream);

/*
 * The reference's two-call sequence (slam.py:321-322: cuda_corr.forward on pyramid[0] with coords, then on pyramid[1] with
 * coords / 4, stacked by slam.py:323) served by one two-level launch and a check:
 *   cdv_corr_fused_split   = cdv_corr_fused with the levels kept apart: out2 [E][2][442] halves (each level a contiguous
 *                            run of 441, rows of 884 bytes), so that either level can be handed out as a tensor view;
 *   cdv_corr_level_checked = the SECOND call, given that the first already produced both levels into out2: every edge whose
 *                            coords equal coords_ref * ref_mul bit for bit keeps what is there, any other edge is
 *                            recomputed from coords into out2[e][level].  ~5 us when all match instead of a second
 *                            correlation.  C <= 32 only.
 */
int cdv_corr_fused_split(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const float* coords,
                         const int64_t* kk, const int64_t* jj, const int32_t* order, void* out2, int64_t E, int64_t Ng,
                         int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0, float scale1, int64_t kmod,
                         int64_t jmod, int gmap_pixel_major, void* stream);
int cdv_corr_level_checked(const void* gmap, const void* fmap_nhwc, const float* coords, const float* coords_ref,
                           float ref_mul, const int64_t* kk, const int64_t* jj, void* out2, int level, int64_t E,
                           int64_t Ng, int64_t slots, int C, int H, int W, float scale, int64_t kmod, int64_t jmod,
                           int gmap_pixel_major, void* stream);
/* cdv_corr_level_checked writing into the interleaved two-level result of cdv_corr_fused ([E][441][2] halves, exactly what
 * SLAM.corr's torch.stack([corr1, corr2], -1) holds, slam.py:323): element t of edge e, level `level`, at
 * out[e * 882 + 2 t + level].  Lets a drop-in hand out both per-level results as views of ONE buffer. */
int cdv_corr_level_checked_interleaved(const void* gmap, const void* fmap_nhwc, const float* coords, const float* coords_ref,
                                       float ref_mul, const int64_t* kk, const int64_t* jj, void* out, int level, int64_t E,
                                       int64_t Ng, int64_t slots, int C, int H, int W, float scale, int64_t kmod, int64_t jmod,
                                       int gmap_pixel_major, void* stream);

/* cuda_corr.patchify_forward(net, coords, radius) -- correlation.cpp:49-52, kernel :16-47.
 *   net [B][C][H][W] (f16 or f32), coords [B][M][2] f32 -> patches [B][M][C][D][D], zero when OOB */
int cdv_patchify_fwd(const void* net, const float* coords, void* patches, int B, int64_t M, int C, int H, int W,
                     int radius, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * projective ops  (replaces the ~15 launches of pops.transform, projective_ops.py:53-113)
 * ---------------------------------------------------------------------------------------------- */

#define CDV_TF_LAYOUT_EPP2 0 /* coords [E][P][P][2]  (what pops.transform returns)            */
#define CDV_TF_LAYOUT_E2PP 1 /* coords [E][2][P][P]  (SLAM.reproject's permute+contiguous)    */
#define CDV_TF_TONLY 2       /* flag: rotation of Gij replaced by identity (projective_ops.py:62) */

/*
 * pops.transform(poses, patches, intrinsics, ii, jj, kk[, valid][, jacobian][, tonly]) in f32,
 * lietorch semantics (quaternions re-normalised on every load, so3.h:30-37; d = 1/max(Z, 0.1)).
 *   poses [n][7], patches [m][3][P][P], intrinsics [n][4] (per-frame), ii/jj/kk [E]
 *   coords out (layout by flags) ; optional (NULL to skip):
 *     validpx [E][P][P]  (X1.z > 0.2)                      projective_ops.py:110-111
 *     valid [E], Ji [E][2][6], Jj [E][2][6], Jz [E][2]     projective_ops.py:71-108
 */
int cdv_transform(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                  const int64_t* jj, const int64_t* kk, int64_t E, int P, int flags, float* coords, float* validpx,
                  float* valid, float* Ji, float* Jj, float* Jz, void* stream);

/* cuda_ba.reproject (ba.cpp:50-57, ba_cuda.cu:408-458): intrinsics row 0, no depth clamp, no
 * quaternion normalisation -> coords [E][2][P][P] */
int cdv_fastba_reproject(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                         const int64_t* jj, const int64_t* kk, int64_t E, int P, float* coords, void* stream);

/* ------------------------------------------------------------------------------------------------
 * patch-graph index (device-side replacement of torch::_unique + the CPU loops of
 * fastba.neighbors, ba.cpp:59-97; shared by neighbors, BA and the correlation order)
 * ---------------------------------------------------------------------------------------------- */

/* bytes of device workspace for a graph of up to E_max edges whose patch ids span at most k_range
 * values (kmax - kmin + 1 <= k_range) */
size_t cdv_graph_workspace_bytes(int64_t E_max, int64_t k_range);

/*
 * Build the index for edge lists (jj = target frame, kk = patch id), [E] int64 -- the arguments of
 * cuda_ba.neighbors(kk, jj).  Contents (device side, used by cdv_neighbors / cdv_ba_forward):
 *   kx [U] sorted unique patch ids, ku [E] inverse index  == torch::_unique(kk, true, true)
 *   patch CSR: for each unique patch its edges ordered by (jj, edge id)
 * (E_max, k_range) must be the values the workspace was sized with.  A patch-id range larger than
 * k_range sets an error word readable via cdv_graph_read_meta_host (consumers then do nothing).
 */
int cdv_graph_build(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes, int64_t E_max,
                    int64_t k_range, void* stream);

/* meta_host[8] <- {U, 0, kmin, kmax, jmin, jmax, error, E}; synchronises `stream`. */
/* cdv_graph_build that also writes fastba.neighbors' result (ba.cpp:59-97) for the same (kk, jj) straight into
 * ix / jx [E] int64 -- the sweep that orders a patch's edges in time sees each edge's predecessor and successor
 * anyway, so the update path needs no separate neighbors launch.  ix == jx == NULL: same as cdv_graph_build. */
int cdv_graph_build_neighbors(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                              int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx, void* stream);

/* cdv_graph_build_neighbors given the source frames ii [E] as well (what cdv_update_prologue does): the per-patch edge
 * records then carry (edge id, ii, jj), so cdv_ba_forward walks them with one 16-byte load per edge instead of a dependent
 * chain of index loads.  ii == NULL: identical to cdv_graph_build_neighbors (the BA then reads ii itself). */
int cdv_graph_build_edges(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                          int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx, void* stream);

int cdv_graph_read_meta_host(const void* ws, int64_t* meta_host, void* stream);

/* The same index as a patch TABLE, in two launches instead of four and with no scan between them: a patch's slot is
 * id mod table_capacity, so nothing has to be counted before records can be placed.  table_capacity (<= min(k_range,
 * 65536)) must be at least the number of ids between the oldest and the newest patch that has an edge -- for a SLAM
 * object (REMOVAL_WINDOW + 2) * PATCHES_PER_FRAME (slam.py:453-458 drops edges of older patches; ids are renumbered when
 * a frame is dropped, :425-427) -- so that no two live ids share a slot; two ids in one slot put the index into its error
 * state, as do a negative id and a patch with more than 128 edges (neighbors all -1, cdv_ba_forward skipped with
 * CDV_ERR_GRAPH_RANGE).
 * Launch 1 puts every edge's record {edge, ii, jj, kk} into its patch's slot in arrival order; launch 2 sorts each slot
 * into (jj, edge id) order -- the order std::stable_sort by jj gives on an ascending index list, ba.cpp:84-86 -- and writes
 * fastba.neighbors from it (ix / jx as in cdv_graph_build_neighbors, bit-exact), the correlation's processing order and,
 * when a coordinate source is bound (cdv_graph_bind_corr_stream), its packed input stream.  A patch with more than 32
 * edges keeps its first 32 records in the table and all of them in an overflow segment.
 * What it does NOT produce is torch::_unique's (kx, ku): no ranks exist (cdv_graph_get_unique refuses; cdv_graph_build* is
 * still there for that and for more than 32 free poses).  cdv_ba_forward accepts either index for 1 <= N <= 32 (U_max >=
 * table_capacity): it works through the slots, a slot without a patch contributes nothing and is not retracted.
 *   cdv_graph_table_offsets: byte offsets of the table's arrays inside the workspace (tools, tests). */
int cdv_graph_build_table(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                          int64_t E_max, int64_t k_range, int64_t table_capacity, int64_t* ix, int64_t* jx, void* stream);
int cdv_graph_table_offsets(int64_t E_max, int64_t k_range, int64_t* out7);

/* A processing order for cdv_corr_fused that the index build produces on the side: the edge ids [E] (int32) grouped by target
 * frame (jj mod 32), so that the share of the list one XCD works through touches ~3 frames' feature maps instead of ~16
 * (the reference launches one thread block per edge in list order, correlation_kernel.cu:82-136; results do not depend on
 * the order).  Device pointer into the workspace, valid after any cdv_graph_build* / cdv_update_prologue on it until the
 * next build; NULL if the workspace holds no index. */
const int32_t* cdv_graph_corr_order(const void* graph_ws);

/* The correlation's inputs as ONE packed stream in that processing order, written by the index build as well: record p
 * (24 x 4 bytes) = the 18 reprojected coordinates of edge order[p], its edge id, and its ring indices kk % kmod, jj % jmod
 * (slam.py:319-320; 0xFFFFFFFF when outside [0, Ng) / [0, slots)).  A correlation wave then needs one memory round trip
 * (its record) before it can request its windows, instead of order[] -> coords / kk / jj.  Replaces nothing in the
 * reference (which reads coords, ii, jj per thread, correlation_kernel.cu:93-113); same values, other placement.
 *   cdv_graph_bind_corr_stream  tells a workspace where the coordinates of the NEXT builds live (coords [E][2][3][3] f32,
 *                               written earlier on the same stream -- cdv_update_prologue does) and the ring sizes;
 *                               coords == NULL unbinds.  Kept across builds, dropped by cdv_workspace_forget.
 *   cdv_graph_corr_records      device pointer of the stream (valid like cdv_graph_corr_order), NULL without an index
 *   cdv_corr_fused_stream       cdv_corr_fused (two levels, C <= 32) reading that stream */
int cdv_graph_bind_corr_stream(void* graph_ws, const float* coords, int64_t kmod, int64_t jmod, int64_t Ng, int64_t slots,
                               float scale0);
const uint32_t* cdv_graph_corr_records(const void* graph_ws);
int cdv_corr_fused_stream(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const void* records, void* out,
                          int64_t E, int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1, float scale0,
                          float scale1, int gmap_pixel_major, void* stream);

/* copy out torch::_unique results (kx needs U from cdv_graph_read_meta_host to size it) */
int cdv_graph_get_unique(const void* ws, int64_t* kx, int64_t kx_capacity, int64_t* ku, int64_t E, void* stream);

/* cuda_ba.neighbors(kk, jj) (ba.cpp:59-97) from a built graph: ix/jx [E] int64, -1 = none */
int cdv_neighbors(const void* ws, int64_t E, int64_t* ix, int64_t* jx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fastba  (replaces cuda_ba.forward, ba.cpp:31-45, ba_cuda.cu:462-611)
 * ---------------------------------------------------------------------------------------------- */

/* bytes of device workspace for up to E_max edges, U_max unique patches and N_max <= 1024 free poses (the dense E,
 * 6 N_max x round_up(U_max, 64) floats, dominates: 2.4 GB at N = 1024, U = 100k; for N_max > 32 the workspace also holds
 * the frame-pair index of the call's edges, ~250 bytes per edge, which is where E_max counts) */
size_t cdv_ba_workspace_bytes(int64_t E_max, int64_t U_max, int N_max);

/*
 * In-place bundle adjustment: `iterations` Gauss-Newton steps with Schur complement over patch
 * inverse depths; poses[t0:t1] and patches[kx] are updated in place, nothing is returned
 * (ba_cuda.cu:462-611; t1 == t0 is the structure-only branch :550-560).  The reference's eff_impl flag only
 * selects a block-sparse storage of E (block_e.cu:38-300) for the same S, y, dX, dZ; there is no such flag here:
 * E stays dense in HBM and the library picks the solver by N.
 *   poses [*][7] f32, patches [*][3][P][P] f32, intrinsics [*][4] (only row 0 is read, :253-259),
 *   target/weight [E][2] f32, lmbda: DEVICE pointer to 1 float, ii/jj/kk [E] int64.
 *   graph_ws: a workspace on which cdv_graph_build(jj, kk, E) has been enqueued on `stream`.
 *   ba_ws / U_max: workspace of cdv_ba_workspace_bytes(E, U_max, N) bytes; U_max bounds the number of
 *             unique patches (exceeding it sets info word 1 and skips the update).  The library zeroes
 *             the accumulators inside it the first time it sees (ba_ws, U_max, N) and keeps them zero
 *             between calls; do not write into it.
 *   dbg (optional, NULL): receives iteration-0 values, with n = 6N and Us = round_up(U_max, 64):
 *             [S n*n (damped) | y n | dX n | dZ Us | C Us | u Us | E n*Us]
 *             (N > 32: only the lower triangle of S is accumulated)
 * Every path sums with one owner and a fixed order per entry and no float atomic reaches HBM.  Results are bitwise
 * reproducible for patch graphs as slam.py builds them -- one source frame per patch, every (patch, target frame) edge once --
 * on all three paths, and for ANY edge list on the global path.  On the N <= 32 paths a patch with two source frames or a
 * duplicated edge sums part of its E column with LDS float atomics (arrival order): correct, reproducible to rounding only.
 * The solve -> retract hand-off inside a launch is all-or-nothing (one verdict word per launch): a lost hand-off sets the
 * hand-off status word and leaves poses and patches exactly as they were.
 * 1 <= N = t1 - t0 <= 10 (the optimisation window): two launches per iteration; <= 32: three, a single-workgroup LDS Cholesky;
 * <= 1024 (global BA, slam.py:460-478): patch / frame-pair / pose / tile owners (the pair index is built inside the call),
 * blocked multi-workgroup Cholesky.  More: CDV_ERR_UNSUPPORTED.
 * Failures inside the launches (not positive definite, U_max exceeded, ...) are reported by cdv_ba_status.
 */
int cdv_ba_forward(float* poses, float* patches, const float* intrinsics, const float* target, const float* weight,
                   const float* lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, int P,
                   int t0, int t1, int iterations, const void* graph_ws, void* ba_ws, size_t ba_ws_bytes,
                   int64_t U_max, float* dbg, void* stream);

/*
 * Status of the last cdv_ba_forward on a workspace.  The reference ignores cholesky_ex's info (ba_cuda.cu:576,590: NaNs
 * propagate silently) and exits the process on its own capacity errors (block_e.cu:20-27); here every failure is a word:
 *   info_host[4] <- { cholesky: 0 or 1 + the pose block whose pivot was not positive,
 *                     overflow: 1 = more unique patches than U_max, the update was skipped,
 *                     hand-off: 1 = an in-launch hand-off timed out, the update was not applied,
 *                     graph:    1 = the graph index was in its range-error state, the update was skipped }
 * The words are reset by the next cdv_ba_forward on the workspace (a workspace recovers by itself).  Synchronises
 * `stream`.  Returns CDV_OK or the CDV_ERR_BA_* / CDV_ERR_GRAPH_RANGE code of the first non-zero word.
 */
int cdv_ba_status(const void* ba_ws, int32_t* info_host, void* stream);

/*
 * Non-blocking alternative: bind four event counters {not positive definite, overflow, hand-off, graph} to a workspace.
 * `counters` must be memory both the host and the device can address (hipHostMalloc / a pinned torch tensor); the kernels
 * add 1 (system-scope load + store) whenever the event happens in an iteration and never reset them, so a host thread can
 * watch them without synchronising any stream.  NULL unbinds.
 */
int cdv_ba_bind_status_counters(void* ba_ws, int32_t* counters);

/*
 * The PPF argument of cuda_ba.forward (fastba/ba.cpp:31-45: patches per frame) as a hint on a workspace: when graph_ws holds a
 * patch table whose capacity is a multiple of it (a frame's patches are then consecutive slots) and it is a multiple of 4, the
 * 10 < N <= 32 path cuts its workgroups' patch ranges per frame -- no workgroup then mixes two source frames (+9 % on
 * BASELINE configs[4]).  Affects summation order only.  0 forgets the hint.
 */
int cdv_ba_set_patches_per_frame(void* ba_ws, int patches_per_frame);

/*
 * Fault injection for the in-launch hand-offs (tests only; process-wide, read by the following cdv_ba_forward calls; no
 * counterpart in the reference, whose kernels hand nothing over inside a launch):  0 off;  1 the solver of the N <= 32
 * paths stalls before it commits / the global path's back substitution withholds one block -- the waiting workgroups
 * give up, the status word `hand-off` is set and NOTHING of the iteration is applied;  2 the solver stalls after its
 * commit -- the waiting workgroups lose patience, learn that the solution is coming, wait on, and the update is applied;
 * 3 (global path) a diagonal block of the one-launch factorisation never raises its flag -- as 1: nothing is applied.
 */
int cdv_ba_test_handoff(int mode);

/*
 * Test hook, host only (no counterpart in the reference): the work item behind ticket `ticket` of the one-launch Cholesky
 * factorisation of the global path (csrc/ba_factor.hip) for a reduced system of nb 64-column blocks.  out[0] = 0 the chain
 * workgroup / 1 a block (r, c) that is solved against L(c, c) / 2 the two pre-accumulated blocks of block row c; out[1] = c;
 * out[2] = r (nb: the right-hand side's block).  Returns the number of tickets.  tests/test_factor_order.py checks with it that
 * every item only needs items with smaller tickets.
 */
int cdv_ba_factor_ticket(int ticket, int nb, int32_t* out);


/* altcorr.patchify(net, coords, radius, mode) -- correlation.py:51-71 -- in one launch: the gather of
 * patchify_forward plus the blend the reference composes from four slice products.
 *   mode 1 'bilinear': out [B][M][C][2r+1][2r+1] FLOAT32 (the reference multiplies float32 offsets into the tile);
 *   mode 2 'upperleft': out [B][M][C][1][1] in the map's dtype. */
int cdv_patchify_blend(const void* net, const float* coords, void* out, int B, int64_t M, int C, int H, int W,
                       int radius, int mode, int dtype, void* stream);

/*
 * The altcorr.patchify calls of one new frame (net_cdv.py:355-374: imap, gmap, colour and patch tiles, all cut at the
 * same M patch centres) in ONE launch.  Job j computes altcorr.patchify(net_j, (coords + (ox, oy)) * (sx, sy), radius,
 * mode) -- the reference scales the centres with the same two float operations before each call -- for batch size 1:
 * out_j [M][C][d][d], d = 2 radius + 1 (mode 1, FLOAT32) or 1 (mode 2, the map's dtype).  coords [M][2] f32 (x, y);
 * jobs is a HOST array of up to CDV_MAX_PATCHIFY_JOBS descriptors.
 */
#define CDV_MAX_PATCHIFY_JOBS 8
typedef struct {
  const void* net;   /* [C][H][W] */
  void* out;
  int C, H, W, radius, mode, dtype;
  float sx, sy, ox, oy;
} cdv_patchify_job;
int cdv_patchify_multi(const cdv_patchify_job* jobs, int n_jobs, const float* coords, int64_t M, void* stream);

/*
 * PatchGraph.edges_loop (patchgraph.py:71-97), the device part: the mean flow magnitude (pops.flow_mag with beta, patch
 * CENTRES only, patchgraph.py:86) of the M patches of source frame f reprojected into target frame j, for every pair
 * j in [j0, j0 + nj), f in [f0, f0 + nf) -- over the valid centres, +inf when not more than 0.75 M of them are valid
 * (patchgraph.py:87-90).  One launch instead of the reference's flatmeshgrid + three transforms + two reductions over
 * nj * nf * M candidate edges.  ix [n_patches] int64 (frame of a patch); out [nj][nf] f32.  The greedy selection that
 * follows (reduce_edges, a sequential numba loop in the reference) stays on the host: cdv_slam_amd/loop.py.
 */
int cdv_loop_flow(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix, int M, int P,
                  int j0, int nj, int f0, int nf, float beta, float* flow_out, void* stream);

/* pops.flow_mag(poses, patches, intrinsics, ii, jj, kk, beta) -- projective_ops.py:120-130, the keyframe test's motion
 * measure (slam.py:399-406): three reprojections per edge in one launch.
 *   flow [E][P][P] f32 ; valid [E][P][P] uint8 (X_ij.z > 0.2) */
int cdv_flow_mag(const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                 const int64_t* jj, const int64_t* kk, int64_t E, int P, float beta, float* flow, uint8_t* valid,
                 void* stream);

/* pops.point_cloud(poses, patches, intrinsics, ix) -- projective_ops.py:115-117 (slam.py:524-526): world points
 * P_ix^-1 * iproj(patch) of M patches, patches [M][3][P][P], ix [M] -> points [M][P][P][4] */
int cdv_point_cloud(const float* poses, const float* patches, const float* intrinsics, const int64_t* ix, int64_t M,
                    int P, float* points, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Start of an update in one launch
 * ---------------------------------------------------------------------------------------------- */

/* cdv_update_prologue with the table index (cdv_graph_build_table): launch 1 = ring / tile ingest next to the table's fill
 * pass, launch 2 = slot sort + neighbors next to the per-edge work -- the reprojection (coords [1,E,2,3,3], the layout
 * SLAM.reproject returns, slam.py:325-329), the correlation's processing order and its packed input stream (ring sizes as
 * bound with cdv_graph_bind_corr_stream; its coords argument is not used here: the coordinates go from registers into
 * both places).  Two launches for everything in front of the correlation (the ranked variant: four).  coords 16-byte
 * aligned; P = 3. */
int cdv_update_prologue_table(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int slot, int C, int H, int W,
                              const void* gmap_planar, void* gmap_pm, int64_t Ng, int64_t gmap_first, int64_t gmap_count,
                              const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                              const int64_t* jj, const int64_t* kk, int64_t E, float* coords, void* graph_ws,
                              size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t table_capacity, int64_t* ix,
                              int64_t* jx, void* stream);

/*
 * cdv_frame_ingest + cdv_transform (P = 3, coordinates only) + the first launch of
 * cdv_graph_build_neighbors SIDE BY SIDE in one grid, then the rest of the index build.  The three have no mutual
 * dependency at the start of SLAM.update (slam.py:480-526: ring writes :676-682, reproject :325-329, the edge
 * lists of this update) and each is a few microseconds of latency-bound work, so one launch costs the longest of
 * them instead of their sum.  Arguments as in the three functions (tf_flags: CDV_TF_LAYOUT_* / CDV_TF_TONLY);
 * gmap_planar / gmap_pm and ix / jx may be NULL as there.
 */
int cdv_update_prologue(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int slot, int C, int H, int W,
                        const void* gmap_planar, void* gmap_pm, int64_t Ng, int64_t gmap_first, int64_t gmap_count,
                        const float* poses, const float* patches, const float* intrinsics, const int64_t* ii,
                        const int64_t* jj, const int64_t* kk, int64_t E, int tf_flags, float* coords, void* graph_ws,
                        size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge bookkeeping of the patch graph on the device (the steps either side of the update path)
 * Edge arrays ii / jj / kk (int64), target / weight ([E][2] f32) and the per-edge hidden state live in
 * fixed-capacity device buffers; nothing is reallocated per call.
 * ---------------------------------------------------------------------------------------------- */

/* append_factors(*edges_forw()) + append_factors(*edges_back()) of the frame that just arrived (slam.py:331-337,
 * 528-541, call site :707-709): written at ii/jj/kk[E0 ...]; ix [patches] = frame of every patch; n = number of
 * frames including the new one, M = patches per frame, r = PATCH_LIFETIME.  *added_host = number of new edges. */
int cdv_edges_frame(int64_t* ii, int64_t* jj, int64_t* kk, const int64_t* ix, int64_t E0, int64_t capacity, int n,
                    int M, int r, int64_t* added_host, void* stream);
/* append_factors(new_k, new_j) for arbitrary (loop-closure) edges: kk <- new_k, jj <- new_j, ii <- ix[new_k] */
int cdv_edges_append(int64_t* ii, int64_t* jj, int64_t* kk, const int64_t* ix, const int64_t* new_k,
                     const int64_t* new_j, int64_t E0, int64_t count, int64_t capacity, void* stream);
/* remove_factors(mask, store) (slam.py:339-354) as a stable stream compaction -- the order boolean-mask indexing
 * gives, bit-exact.  remove [E] uint8 (1 = drop).  Kept edges go, compacted, to the *_out buffers (twin buffers of
 * the same capacity); dropped ones are appended to the inactive arrays at index r0 when ii_r != NULL (store = True).
 * target / weight / net (+ their outputs) may be NULL; net rows are net_bytes wide (multiple of 4).
 * ws: cdv_edges_workspace_bytes(capacity) bytes, zeroed once by the caller.  counts_host (optional) receives
 * {kept, removed} by an async copy: valid after the stream is synchronised. */
size_t cdv_edges_workspace_bytes(int64_t capacity);
int cdv_edges_remove(const uint8_t* remove, int64_t E, void* ws, const int64_t* ii, const int64_t* jj, const int64_t* kk,
                     const float* target, const float* weight, const void* net, int net_bytes, int64_t* ii_out,
                     int64_t* jj_out, int64_t* kk_out, float* target_out, float* weight_out, void* net_out,
                     int64_t* ii_r, int64_t* jj_r, int64_t* kk_r, float* target_r, float* weight_r, int64_t r0,
                     int32_t* counts_host, void* stream);
/* keyframe(): index shift after frame k is dropped (slam.py:425-427): kk[ii > k] -= M; ii[ii > k] -= 1; jj[jj > k] -= 1 */
int cdv_edges_keyframe_shift(int64_t* ii, int64_t* jj, int64_t* kk, int64_t E, int k, int M, void* stream);

/* keyframe(): the frame buffers after frame k is dropped (slam.py:431-441): for i = k .. n - 2, in this order,
 * buf[slot(i)] = buf[slot(i + 1)] for every per-frame buffer -- tstamps_, colors_, poses_, patches_, intrinsics_
 * (modulus 0: slot(i) = i) and the rings imap_, gmap_ (modulus pmem), fmap1_, fmap2_ (modulus mem: slot(i) = i % modulus).
 * One launch for all of them (the reference: nine tensor copies per shifted frame).  bufs is a HOST array of up to
 * CDV_MAX_FRAME_BUFS descriptors (device base pointer, 4-byte aligned; bytes per slot, a multiple of 4). */
#define CDV_MAX_FRAME_BUFS 16
typedef struct {
  void* base;
  int64_t slot_bytes;
  int32_t modulus;
  int32_t reserved;
} cdv_frame_buf;
int cdv_frames_keyframe_shift(const cdv_frame_buf* bufs, int n_bufs, int k, int n, void* stream);

/*
 * ---- a frame stream whose sizes live on the DEVICE ------------------------------------------------------------------
 * The reference decides per frame, on the host, whether frame n - KEYFRAME_INDEX leaves the graph (two .item() read-backs,
 * cdvslam/slam.py:399-413), and every size that follows -- the number of keyframes n, the number of edges E after the
 * mask-indexing of remove_factors (slam.py:339-354) -- is a host integer again.  Here the decision and the sizes can stay
 * on the device: a "dynamic block" of CDV_DYN_WORDS int32 words holds them, the *_dyn entry points read their sizes from
 * it (their size ARGUMENTS are then upper bounds that only dimension the launches), and the three entry points that
 * change a size -- cdv_stream_frame_begin, and the two compactions inside cdv_stream_keyframe -- read one block and write
 * the next (the caller hands in a small ring of blocks: nothing is updated in place under a running launch).  No host
 * synchronisation anywhere in a frame; the whole frame is a fixed sequence of launches (hipGraph-capturable).
 */
#define CDV_DYN_WORDS 16
enum {
  CDV_DYN_N = 0,      /* keyframes in the graph (slam.py: self.n) */
  CDV_DYN_E = 1,      /* active edges */
  CDV_DYN_EINAC = 2,  /* inactive (stored) edges */
  CDV_DYN_DROP = 3,   /* outcome of the last keyframe test: 1 = frame n - KEYFRAME_INDEX was dropped */
  CDV_DYN_T0 = 4,     /* first free pose of the update: max(1, n - OPTIMIZATION_WINDOW) (slam.py:512-513) */
  CDV_DYN_NFREE = 5,  /* n - t0 */
  CDV_DYN_FRAME = 6,  /* frames seen so far */
  CDV_DYN_ERR = 7     /* != 0: a capacity was exceeded (edges / inactive edges / frame buffer); the stream stops changing */
};

/* cdv_update_prologue_table (above) with the sizes on the device: dyn[CDV_DYN_E] edges (E_bound dimensions the launches), the
 * new frame goes to ring slot (n - 1) % mem and tiles [((n - 1) % pmem) * tiles_per_frame, + tiles_per_frame), n = dyn[CDV_DYN_N].
 * No neighbor lists are written (cdv_neighbors serves them from the index). */
int cdv_update_prologue_table_dyn(const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int mem, int pmem, int C, int H, int W,
                                  const void* gmap_planar, void* gmap_pm, int64_t Ng, int64_t tiles_per_frame, const float* poses,
                                  const float* patches, const float* intrinsics, const int64_t* ii, const int64_t* jj,
                                  const int64_t* kk, int64_t E_bound, const int32_t* dyn, float* coords, void* graph_ws,
                                  size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t table_capacity, void* stream);

/* cdv_corr_fused_stream with dyn[CDV_DYN_E] edges; out must hold E_bound rows. */
int cdv_corr_fused_stream_dyn(const void* gmap, const void* fmap0_nhwc, const void* fmap1_nhwc, const void* records, void* out,
                              int64_t E_bound, const int32_t* dyn, int64_t Ng, int64_t slots, int C, int H0, int W0, int H1, int W1,
                              float scale0, float scale1, int gmap_pixel_major, void* stream);

/* cdv_ba_forward over the window [dyn[CDV_DYN_T0], + dyn[CDV_DYN_NFREE]) (slam.py:512-513), NFREE <= N_max <= 32 (N_max <= 10:
 * the window kernels; 11 .. 32: the three-launch path of the wider OPTIMIZATION_WINDOW configurations, default_cdvo++.yaml);
 * graph_ws must hold a patch table; E_bound / U_max size the workspace (cdv_ba_workspace_bytes(E_bound, U_max, N_max)). */
int cdv_ba_forward_dyn(float* poses, float* patches, const float* intrinsics, const float* target, const float* weight,
                       const float* lmbda, const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E_bound, int P,
                       int N_max, const int32_t* dyn, int iterations, const void* graph_ws, void* ba_ws, size_t ba_ws_bytes,
                       int64_t U_max, void* stream);

/* Workspace of the stream entry points below (zero-initialise once). */
size_t cdv_stream_workspace_bytes(int64_t edge_capacity, int M);

/*
 * A frame arrives (slam.py:697-709; the state write of :676-696 when the frame's patch centres are given): dyn_out <- dyn_in
 * with n + 1, E + the frame's forward / backward edges (slam.py:528-541), the window of the coming update; the edges are
 * appended at E (ii = ix[kk], their target / weight rows zeroed, slam.py:331-337).  cx / cy / d [M] (optional): the new
 * frame's patch centres and inverse depths -> patches_[n] (3 x 3 grids), the pose guess poses_[n] = poses_[n - 1] moved by
 * pose_step along x, and the frame's patch tiles gmap_[n % pmem] = altcorr.patchify(fmap, centres, 1, 'bilinear') in half.
 */
int cdv_stream_frame_begin(const int32_t* dyn_in, int32_t* dyn_out, int64_t* ii, int64_t* jj, int64_t* kk, float* target,
                           float* weight, const int64_t* ix, int64_t edge_capacity, int M, int patch_lifetime, int opt_window,
                           int frames_capacity, const float* cx, const float* cy, const float* d, const void* fmap_chw,
                           void* gmap_planar, float* poses, float* patches, int C, int H, int W, int pmem, float pose_step,
                           void* ws, void* stream);

/* Stand-in for the update operator in a stream without networks: target = reprojected patch centre + gain * tanh(corr[:, 0:2]),
 * weight = sigmoid(corr[:, 2:4]) for the dyn[CDV_DYN_E] edges (coords [E][2][3][3] f32, corr rows of corr_pitch halves). */
int cdv_stream_operator_stub(const int32_t* dyn, const float* coords, const void* corr, int corr_pitch, float* target,
                             float* weight, float gain, int64_t E_bound, void* stream);

/* slam.py:524-526 restricted to what can have moved: points [*][3] <- world point of the centre pixel of every patch of the
 * last window_frames keyframes. */
int cdv_stream_points(const int32_t* dyn, const float* poses, const float* patches, const float* intrinsics, const int64_t* ix,
                      int M, int window_frames, float* points, void* stream);

/*
 * SLAM.keyframe() (slam.py:408-458) and the point cloud of slam.py:524-526, 3 x 3 patches, three launches, no read-back: mean
 * flow_mag of the edges between the frames either side of k = n - keyframe_index (slam.py:399-413, beta 0.5) -> decision on the
 * device (force -1: drop k when the mean is under keyframe_thresh and n > keyframe_index + 2; 0 / 1: the caller decides) -> ONE
 * stable compaction src -> dst that does both removals of keyframe(): k's edges go (:423), the indices above k are shifted
 * (:425-427), the edges whose source frame left the removal window (judged on the shifted indices, n - 1) are stored as
 * inactive edges (:453-458); next to it the frame buffers `bufs` are shifted (:431-441, as cdv_frames_keyframe_shift).
 * dyn_in -> dyn_out (n - drop, the edges left, the inactive edges).  points (optional): as cdv_stream_points, same launch as the
 * flow statistic.  mirror_host (optional, pinned): receives (frames << 32 | edges) so that the host can size the next launches
 * without synchronising.
 */
int cdv_stream_keyframe(const int32_t* dyn_in, int32_t* dyn_out, const float* poses, const float* patches, const float* intrinsics,
                        const int64_t* ix, const int64_t* ii_src, const int64_t* jj_src, const int64_t* kk_src,
                        const float* target_src, const float* weight_src, int64_t* ii_dst, int64_t* jj_dst, int64_t* kk_dst,
                        float* target_dst, float* weight_dst, int64_t* ii_inac, int64_t* jj_inac, int64_t* kk_inac,
                        float* target_inac, float* weight_inac, int64_t inactive_capacity, int64_t edge_capacity, int64_t E_bound,
                        int M, int keyframe_index, int removal_window, float keyframe_thresh, int force, const cdv_frame_buf* bufs,
                        int n_bufs, float* points, int64_t* mirror_host, void* ws, void* stream);

/* device pointer to {flow statistic, decision} of the last cdv_stream_keyframe on the workspace (tests). */
const float* cdv_stream_motion(void* ws, int64_t edge_capacity, int M);

/*
 * One frame of the stream as ONE call: everything SLAM.__call__ does for an initialised system around the (stubbed) networks
 * -- cdv_stream_frame_begin, cdv_update_prologue_table_dyn, cdv_corr_fused_stream_dyn, cdv_stream_operator_stub,
 * cdv_ba_forward_dyn (2 iterations), cdv_stream_keyframe (with the point cloud); for the first 7 frames only the state write
 * and the ring ingest, as slam.py:711 waits for 8 frames -- enqueued on `stream`, 12 launches, no synchronisation.  The
 * descriptor names the buffers once; `slot` and `frames` are host state the call advances; the launches are sized from
 * the pinned word `mirror_host` (cdv_stream_keyframe).  force: -1 the reference's keyframe test, 0 / 1 the caller's decision.
 */
typedef struct {
  int32_t M, C, H, W, mem, pmem, frames_capacity, patch_lifetime, removal_window, opt_window, keyframe_index, n_bufs;
  float keyframe_thresh, gain, pose_step;
  int32_t slot, frames, cur;         /* host state: current dynamic block, frames begun, which twin holds the edge lists */
  int32_t ring_blocks;               /* dynamic blocks in `dyn`: 0 = 8.  2: a frame ends in the block it started from, so TWO frames
                                        (the edge lists are back in their first twin then) are a closed sequence of launches */
  int32_t fixed_bound;               /* != 0: every launch is sized for edge_capacity instead of the bound read from mirror_host --
                                        together with ring_blocks = 2 a pair of frames can be captured as a hipGraph and replayed */
  int64_t edge_capacity, inactive_capacity, table_capacity, graph_E_max, graph_k_range;
  size_t graph_ws_bytes, ba_ws_bytes;
  float *poses, *patches, *intrinsics, *points;
  const int64_t* ix;
  void *fmap1_nhwc, *fmap2_nhwc, *gmap_planar, *gmap_pm;
  int64_t *ii[2], *jj[2], *kk[2];    /* the edge lists and their twin: [cur] is current, every keyframe() compacts into the other */
  float *target[2], *weight[2];
  int64_t *ii_inac, *jj_inac, *kk_inac;
  float *target_inac, *weight_inac;
  float* coords;                     /* [edge_capacity][2][3][3] */
  void* corr_out;                    /* [edge_capacity][882] f16 */
  const float* lmbda;
  int32_t* dyn;                      /* ring of 8 dynamic blocks (8 x CDV_DYN_WORDS int32, zero-initialised) */
  void *ws, *graph_ws, *ba_ws;
  int64_t* mirror_host;              /* pinned, zero-initialised */
  cdv_frame_buf bufs[CDV_MAX_FRAME_BUFS];
} cdv_stream_desc;
int cdv_stream_frame(cdv_stream_desc* desc, const void* fmap_chw, const float* cx, const float* cy, const float* depth, int force,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * lietorch forward ops  (replaces lietorch_backends.{expm,logm,inv,mul,adj,adjT,act,act4,as_matrix})
 * group ids as the reference: SO3 = 1, SE3 = 3 (lietorch.cpp:286-316, groups.py:236-290).
 * op: 0 exp, 1 log, 2 inv, 3 mul, 4 adj, 5 adjT, 6 act, 7 act4, 8 as_matrix.  Flat [n][dim] rows.
 * ---------------------------------------------------------------------------------------------- */
int cdv_lie_op(int group, int op, int dtype, int64_t n, const void* x, const void* y, void* z, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDVSLAM_HIP_H */
