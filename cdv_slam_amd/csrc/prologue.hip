// prologue.hip -- the three independent kernels that open an update, in ONE launch.
//
// SLAM.update (cdvslam/slam.py:480-526) starts with work that has no mutual dependency: the new frame's feature maps
// and patch tiles go into the rings (slam.py:676-682), the patches are reprojected (slam.py:325-329), and the
// patch-graph index of this update's edge lists is started (ba_cuda.cu:476-478 / ba.cpp:59-97).  Each is a few
// microseconds of latency-bound work; as separate launches they cost the sum of their latencies, side by side in one
// grid they cost the longest one.  Workgroups [0, n_hist) histogram the patch ids, the next n_ing ingest, the rest
// reproject; then the rest of the index build (scan, fill, segment sort + neighbors) follows.
#include "cdv_common.h"
#include "cdv_graph.h"
#include "cdv_parts.h"

namespace {

__global__ __launch_bounds__(256) void update_prologue_kernel(cdv::IngestArgs ing, int n_ing, cdv::TfArgs tf, int n_tf,
                                                              cdv::HistArgs hist, int n_hist) {
  // the histogram workgroups first: theirs is the longest dependent chain (atomics -> drain -> arrival count -> the last
  // one scans), so they get the head start of the dispatch order
  int b = (int)blockIdx.x;
  if (b < n_hist) {
    cdv::graph_hist_body(hist, b, n_hist, 256, (int)threadIdx.x);
    return;
  }
  b -= n_hist;
  if (b < n_ing) {
    cdv::ingest_body(ing, b, 256, (int)threadIdx.x);
    return;
  }
  b -= n_ing;
  cdv::transform_body<3>(tf, (int64_t)b * 256 + threadIdx.x);
}

// the same opening with the table index: ring / tile ingest next to the table's fill pass (nothing else has to wait for
// a scan any more); the reprojection moves into the index's second launch, where the processing order is known and the
// correlation's packed input stream can be written from the registers that hold the coordinates
__global__ __launch_bounds__(256) void update_prologue_table_kernel(cdv::IngestArgs ing, int n_ing, cdv::TFillArgs fill,
                                                                    int n_fill) {
  int b = (int)blockIdx.x;
  if (b < n_fill) {
    cdv::graph_tfill_body(fill, b, n_fill, 256, (int)threadIdx.x);
    return;
  }
  cdv::ingest_body(ing, b - n_fill, 256, (int)threadIdx.x);
}

}  // namespace

static int prologue_table_impl(
    const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int slot, int C, int H, int W, const void* gmap_planar,
    void* gmap_pm, int64_t Ng, int64_t gmap_first, int64_t gmap_count, const float* poses, const float* patches,
    const float* intrinsics, const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, float* coords,
    void* graph_ws, size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t table_capacity, int64_t* ix, int64_t* jx,
    void* stream, const int32_t* dyn, int dyn_mem, int dyn_pmem) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_update_prologue_table: C must be a multiple of 8");
  CDV_REQUIRE(H % 4 == 0 && W % 4 == 0, CDV_ERR_ARG, "cdv_update_prologue_table: H and W must be multiples of 4");
  CDV_REQUIRE(slot >= 0, CDV_ERR_ARG, "cdv_update_prologue_table: slot");
  CDV_REQUIRE(fmap_chw && fmap1_nhwc && fmap2_nhwc && coords && poses && patches && intrinsics && ii, CDV_ERR_ARG,
              "cdv_update_prologue_table: NULL buffer");
  CDV_REQUIRE(((uintptr_t)coords & 15) == 0, CDV_ERR_ARG, "cdv_update_prologue_table: coords must be 16-byte aligned");
  const bool do_g = gmap_planar != nullptr && gmap_pm != nullptr && gmap_count > 0;
  CDV_REQUIRE(!do_g || (gmap_first >= 0 && gmap_first + gmap_count <= Ng), CDV_ERR_ARG, "cdv_update_prologue_table: tile range");
  cdv::TFillArgs fill;
  int n_fill = 0;
  const int rc = cdv_graph_table_prepare(ii, jj, kk, E, graph_ws, graph_ws_bytes, E_max, k_range, table_capacity, ix, jx, stream,
                                         &fill, &n_fill, dyn);
  if (rc != CDV_OK) return rc;
  const int fblocks = cdv_div_up((int64_t)(H / 4) * (W / 4) * (C / 8) * 16, 256);
  const int gblocks = do_g ? cdv_div_up(gmap_count * 9 * (C / 8), 256) : 0;
  const cdv::IngestArgs ing{(const _Float16*)fmap_chw, (_Float16*)fmap1_nhwc, (_Float16*)fmap2_nhwc, nullptr, nullptr,
                            slot, C, H, W, (const _Float16*)gmap_planar, (_Float16*)gmap_pm, gmap_first, gmap_count,
                            fblocks, gblocks, dyn, dyn_mem, dyn_pmem};
  const int n_ing = fblocks + gblocks;
  hipLaunchKernelGGL(update_prologue_table_kernel, dim3(n_fill + n_ing), dim3(256), 0, (hipStream_t)stream, ing, n_ing, fill,
                     n_fill);
  CDV_LAUNCH_CHECK();
  return cdv_graph_table_finish(fill, n_fill, graph_ws, E_max, k_range, ix, jx, poses, patches, intrinsics, coords, true, stream);
}

extern "C" int cdv_update_prologue_table(
    const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int slot, int C, int H, int W, const void* gmap_planar,
    void* gmap_pm, int64_t Ng, int64_t gmap_first, int64_t gmap_count, const float* poses, const float* patches,
    const float* intrinsics, const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, float* coords,
    void* graph_ws, size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t table_capacity, int64_t* ix, int64_t* jx,
    void* stream) {
  return prologue_table_impl(fmap_chw, fmap1_nhwc, fmap2_nhwc, slot, C, H, W, gmap_planar, gmap_pm, Ng, gmap_first, gmap_count,
                             poses, patches, intrinsics, ii, jj, kk, E, coords, graph_ws, graph_ws_bytes, E_max, k_range,
                             table_capacity, ix, jx, stream, nullptr, 0, 0);
}

// the same with the sizes on the device (include/cdvslam_hip.h "a frame stream whose sizes live on the DEVICE"): the number of
// edges is dyn[CDV_DYN_E] (E_bound only dimensions the launches), the newest keyframe n - 1 = dyn[CDV_DYN_N] - 1 names the
// ring slot ((n - 1) % mem) and the tiles ([((n - 1) % pmem) * tiles_per_frame, + tiles_per_frame)) the new frame goes to
extern "C" int cdv_update_prologue_table_dyn(
    const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int mem, int pmem, int C, int H, int W, const void* gmap_planar,
    void* gmap_pm, int64_t Ng, int64_t tiles_per_frame, const float* poses, const float* patches, const float* intrinsics,
    const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E_bound, const int32_t* dyn, float* coords, void* graph_ws,
    size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t table_capacity, void* stream) {
  CDV_REQUIRE(dyn != nullptr && mem >= 1 && pmem >= 1 && tiles_per_frame >= 1 && (int64_t)pmem * tiles_per_frame <= Ng, CDV_ERR_ARG,
              "cdv_update_prologue_table_dyn: dyn / ring sizes");
  CDV_REQUIRE(E_bound >= 1, CDV_ERR_ARG, "cdv_update_prologue_table_dyn: E_bound must be >= 1");
  return prologue_table_impl(fmap_chw, fmap1_nhwc, fmap2_nhwc, 0, C, H, W, gmap_planar, gmap_pm, Ng, 0, tiles_per_frame, poses,
                             patches, intrinsics, ii, jj, kk, E_bound, coords, graph_ws, graph_ws_bytes, E_max, k_range,
                             table_capacity, nullptr, nullptr, stream, dyn, mem, pmem);
}

extern "C" int cdv_update_prologue(
    // cdv_frame_ingest
    const void* fmap_chw, void* fmap1_nhwc, void* fmap2_nhwc, int slot, int C, int H, int W, const void* gmap_planar,
    void* gmap_pm, int64_t Ng, int64_t gmap_first, int64_t gmap_count,
    // cdv_transform (P = 3, coords only)
    const float* poses, const float* patches, const float* intrinsics, const int64_t* ii, const int64_t* jj,
    const int64_t* kk, int64_t E, int tf_flags, float* coords,
    // cdv_graph_build_neighbors
    void* graph_ws, size_t graph_ws_bytes, int64_t E_max, int64_t k_range, int64_t* ix, int64_t* jx, void* stream) {
  CDV_REQUIRE(C % 8 == 0 && C > 0, CDV_ERR_ARG, "cdv_update_prologue: C must be a multiple of 8");
  CDV_REQUIRE(H % 4 == 0 && W % 4 == 0, CDV_ERR_ARG, "cdv_update_prologue: H and W must be multiples of 4");
  CDV_REQUIRE(slot >= 0, CDV_ERR_ARG, "cdv_update_prologue: slot");
  CDV_REQUIRE(fmap_chw && fmap1_nhwc && fmap2_nhwc && coords, CDV_ERR_ARG, "cdv_update_prologue: NULL buffer");
  const bool do_g = gmap_planar != nullptr && gmap_pm != nullptr && gmap_count > 0;
  CDV_REQUIRE(!do_g || (gmap_first >= 0 && gmap_first + gmap_count <= Ng), CDV_ERR_ARG, "cdv_update_prologue: tile range");
  cdv::HistArgs hist;
  int n_hist = 0;
  const int rc = cdv_graph_prepare(jj, kk, E, graph_ws, graph_ws_bytes, E_max, k_range, ix, jx, stream, &hist, &n_hist);
  if (rc != CDV_OK) return rc;
  const int fblocks = cdv_div_up((int64_t)(H / 4) * (W / 4) * (C / 8) * 16, 256);
  const int gblocks = do_g ? cdv_div_up(gmap_count * 9 * (C / 8), 256) : 0;
  const cdv::IngestArgs ing{(const _Float16*)fmap_chw, (_Float16*)fmap1_nhwc, (_Float16*)fmap2_nhwc, nullptr, nullptr,
                            slot, C, H, W, (const _Float16*)gmap_planar, (_Float16*)gmap_pm, gmap_first, gmap_count,
                            fblocks, gblocks};
  const cdv::TfArgs tf{poses, patches, intrinsics, ii, jj, kk, E, tf_flags, coords, nullptr, nullptr, nullptr, nullptr,
                       nullptr};
  const int n_ing = fblocks + gblocks, n_tf = cdv_div_up(E, 256);
  hipLaunchKernelGGL(update_prologue_kernel, dim3(n_ing + n_tf + n_hist), dim3(256), 0, (hipStream_t)stream, ing, n_ing,
                     tf, n_tf, hist, n_hist);
  CDV_LAUNCH_CHECK();
  return cdv_graph_finish(ii, jj, kk, E, graph_ws, E_max, k_range, n_hist, ix, jx, stream);
}
