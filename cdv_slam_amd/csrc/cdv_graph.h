// cdv_graph.h -- device workspace layout of the patch-graph index (internal).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace cdv {

// meta words (int32), written on device
enum {
  GM_U = 0,        // number of unique patches
  GM_KMIN = 2,     // published by the scan kernel
  GM_KMAX = 3,
  GM_JMIN = 4,
  GM_JMAX = 5,
  GM_ERROR = 6,    // != 0: the patch-id range exceeded the workspace capacity (index contents undefined)
  GM_E = 7,
  GM_KRANGE = 8,   // kmax - kmin + 1 of the LAST successful build (0 if none): the range to re-zero
  GM_HAS_II = 9,   // != 0: the CSR records carry the source frame of every edge (the build was given ii)
  GM_STAGE = 16,   // arrival counter of the histogram launch (its last workgroup does the scan); zero between builds
  // ---- table build (cdv_graph_build_table / cdv_update_prologue_table; no scan, no ranks: a patch's slot is id mod R) ----
  GM_MODE = 20,    // 1: the index in the workspace is a patch TABLE, 0: the ranked CSR index above
  GM_GEN = 21,     // generation (build counter) of the table build in the workspace.  The counter lives HERE, on the device:
                   // a fill launch takes its generation from GM_GENNEXT and publishes it, the sort launch behind it advances
                   // GM_GENNEXT -- no host value is frozen into a launch, so a captured hipGraph may be replayed for ever
  GM_TCAP = 23,    // capacity R (slots) of the table build in the workspace
  GM_GENNEXT = 24, // generation the NEXT fill launch will take (written by the sort launch; only its parity matters)
  GM_PRECN = 26,   // records handed out of the overflow CSR (patches with more than ELL_SLOTS edges)
  GM_OVFN = 28,    // [2], by build parity: edges that did not fit their patch's ELL_SLOTS table slots
  GM_TERR = 30,    // [2], by build parity: != 0: that build is in its error state (negative id, two live ids in one slot, a
                   // patch with more than TAB_MAX_DEG edges).  The sort launch of build g clears the word of build g + 1.
  GM_WORDS = 64
};

constexpr int GRAPH_MAX_BLOCKS = 1024;   // workgroups of the per-edge kernels (one min/max staging slot each)
// "chunk-slot" copy of the edge records for the window bundle adjustment: record of the t-th edge (in (jj, edge id)
// order) of the patch with unique rank u at [(u / 16) * ELL_SLOTS + t][u % 16] -- addressable without the CSR offsets, so
// the BA's first memory round trip already fetches records.  Kept for t < ELL_SLOTS and u < 16 * ELL_CHUNKS (beyond: CSR).
constexpr int ELL_SLOTS = 32;
constexpr int ELL_CHUNKS = 4096;
constexpr int TAB_MAX_DEG = 128;   // table build: edges of one patch the sort launch handles (beyond: error state)
constexpr int64_t TAB_CAP_MAX = 1 << 16;   // largest table capacity (slots) a workspace is laid out for
// processing order of the fused correlation: edges grouped by target frame (jj mod ORD_BINS), so that the eighth of the
// list an XCD works through touches three frames' feature maps instead of sixteen (counting sort riding the index build:
// per-block counts from the histogram launch, positions and scatter in the fill launch)
constexpr int ORD_BINS = 32;
// packed input stream of the fused correlation, one record per edge in PROCESSING order (position p of `order`):
// words [0, 18) the edge's reprojected coordinates (x[9], y[9]), [18] edge id, [19] patch-ring index, [20] frame-ring
// index (both 0xFFFFFFFF when an index is outside its ring), [21] / [22] the extremes of floor(x / scale0) / floor(y / scale0)
// over the nine pixels as (max << 16) | (min & 0xffff), [23] reserved.  96 bytes: a record never straddles more
// than two 64-byte lines, and the correlation reads it with one vector and one scalar load.
constexpr int CORR_REC_WORDS = 24;

// what the index build needs to write that stream (bound to a workspace with cdv_graph_bind_corr_stream)
struct CorrStream {
  const float* coords;   // [E][2][3][3] f32, written earlier on the same stream (the prologue's reprojection)
  uint32_t kmod, jmod, kmagic, jmagic, Ng, slots;
  float inv_scale0;      // 1 / scale of pyramid level 0 (slam.py:321: 1)
};

struct GraphLayout {
  int64_t E_max, k_range;
  size_t meta, stage, khist, kcount, kcursor, krank, koff_u, kx, ku, pcsr_tmp, pcsr, prec, pell, nprev, nnext, ocnt, order, crec,
      tcur, town, tdeg, tplo, tkid, tlive, ttab, tovf, tprec, total;
  int64_t tab_chunks;
  int64_t ell_chunks;
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline GraphLayout graph_layout(int64_t E_max, int64_t k_range) {
  GraphLayout L;
  L.E_max = E_max; L.k_range = k_range;
  const int64_t U_max = k_range < E_max ? k_range : E_max;
  size_t o = 0;
  L.meta = o;     o = align256(o + sizeof(int32_t) * GM_WORDS);
  L.stage = o;    o = align256(o + sizeof(int32_t) * 4 * GRAPH_MAX_BLOCKS);   // per-block (kmin, kmax, jmin, jmax) of the build in flight
  L.khist = o;    o = align256(o + sizeof(int32_t) * (size_t)(k_range + 1));   // histogram over id mod k_range (zero between builds)
  L.kcount = o;   o = align256(o + sizeof(int32_t) * (size_t)(k_range + 1));   // dense CSR offsets by id - kmin
  L.kcursor = o;  o = align256(o + sizeof(int32_t) * (size_t)(k_range + 1));
  L.krank = o;    o = align256(o + sizeof(int32_t) * (size_t)k_range);
  L.koff_u = o;   o = align256(o + sizeof(int32_t) * (size_t)(U_max + 1));
  L.kx = o;       o = align256(o + sizeof(int64_t) * (size_t)U_max);
  L.ku = o;       o = align256(o + sizeof(int32_t) * (size_t)E_max);
  L.pcsr_tmp = o; o = align256(o + sizeof(int32_t) * 4 * (size_t)E_max);   // unsorted per-patch lists, {edge, jj, id - kmin, 0} per entry
  L.pcsr = o;     o = align256(o + sizeof(int32_t) * (size_t)E_max);
  L.prec = o;     o = align256(o + sizeof(int32_t) * 4 * (size_t)E_max);   // CSR records {edge, ii, jj, 0} in pcsr order
  L.ell_chunks = (U_max + 15) / 16 < ELL_CHUNKS ? (U_max + 15) / 16 : ELL_CHUNKS;
  L.pell = o;     o = align256(o + sizeof(int32_t) * 4 * 16 * (size_t)ELL_SLOTS * (size_t)L.ell_chunks);
  L.nprev = o;    o = align256(o + sizeof(int32_t) * (size_t)E_max);   // neighbors: previous / next edge of the same patch in time
  L.nnext = o;    o = align256(o + sizeof(int32_t) * (size_t)E_max);
  L.ocnt = o;     o = align256(o + sizeof(int32_t) * ORD_BINS * GRAPH_MAX_BLOCKS);   // [block][bin] edges of the block per target bin
  L.order = o;    o = align256(o + sizeof(int32_t) * (size_t)E_max);
  L.crec = o;     o = align256(o + sizeof(uint32_t) * CORR_REC_WORDS * (size_t)E_max);
  // patch table: slot = patch id mod R, R <= the table capacity the workspace was sized for = min(k_range, TAB_CAP_MAX).
  // Records in the chunk-slot layout of `pell` with the slot in place of the unique rank; tcur = fill cursors (zero
  // between builds), tdeg / tplo / tkid = degree, overflow-CSR offset and patch id per slot, tlive = live slots per
  // workgroup of the sort launch
  const int64_t tcap = k_range < TAB_CAP_MAX ? k_range : TAB_CAP_MAX;
  L.tab_chunks = (tcap + 15) / 16;
  L.tcur = o;     o = align256(o + sizeof(int32_t) * (size_t)(tcap + 16));
  L.town = o;     o = align256(o + sizeof(uint64_t) * (size_t)(tcap + 16));
  L.tdeg = o;     o = align256(o + sizeof(int32_t) * (size_t)(tcap + 16));
  L.tplo = o;     o = align256(o + sizeof(int32_t) * (size_t)(tcap + 16));
  L.tkid = o;     o = align256(o + sizeof(int32_t) * (size_t)(tcap + 16));
  L.tlive = o;    o = align256(o + sizeof(int32_t) * (size_t)(tcap / 8 + 2));
  L.ttab = o;     o = align256(o + sizeof(int32_t) * 4 * 16 * (size_t)ELL_SLOTS * (size_t)L.tab_chunks);
  L.tovf = o;     o = align256(o + sizeof(int32_t) * 4 * (size_t)E_max);
  L.tprec = o;    o = align256(o + sizeof(int32_t) * 4 * ((size_t)E_max + 1));   // [0]: a record that is always valid
  L.total = o;
  return L;
}

struct GraphView {
  int32_t* meta;
  int32_t *stage, *khist, *kcount, *kcursor, *krank, *koff_u, *ku, *pcsr_tmp, *pcsr, *prec, *pell, *nprev, *nnext, *ocnt, *order;
  uint32_t* crec;
  int32_t *tcur, *tdeg, *tplo, *tkid, *tlive, *ttab, *tovf, *tprec;
  unsigned long long* town;
  int64_t* kx;
};

static inline GraphView graph_view(void* ws, const GraphLayout& L) {
  char* b = (char*)ws;
  GraphView v;
  v.meta = (int32_t*)(b + L.meta);
  v.stage = (int32_t*)(b + L.stage);
  v.khist = (int32_t*)(b + L.khist);
  v.kcount = (int32_t*)(b + L.kcount);
  v.kcursor = (int32_t*)(b + L.kcursor);
  v.krank = (int32_t*)(b + L.krank);
  v.koff_u = (int32_t*)(b + L.koff_u);
  v.kx = (int64_t*)(b + L.kx);
  v.ku = (int32_t*)(b + L.ku);
  v.pcsr_tmp = (int32_t*)(b + L.pcsr_tmp);
  v.pcsr = (int32_t*)(b + L.pcsr);
  v.prec = (int32_t*)(b + L.prec);
  v.pell = (int32_t*)(b + L.pell);
  v.nprev = (int32_t*)(b + L.nprev);
  v.nnext = (int32_t*)(b + L.nnext);
  v.ocnt = (int32_t*)(b + L.ocnt);
  v.order = (int32_t*)(b + L.order);
  v.crec = (uint32_t*)(b + L.crec);
  v.tcur = (int32_t*)(b + L.tcur);
  v.tdeg = (int32_t*)(b + L.tdeg);
  v.tplo = (int32_t*)(b + L.tplo);
  v.town = (unsigned long long*)(b + L.town);
  v.tkid = (int32_t*)(b + L.tkid);
  v.tlive = (int32_t*)(b + L.tlive);
  v.ttab = (int32_t*)(b + L.ttab);
  v.tovf = (int32_t*)(b + L.tovf);
  v.tprec = (int32_t*)(b + L.tprec);
  return v;
}

}  // namespace cdv

// host-side registry: workspace pointer -> layout (filled by cdv_graph_build)
bool cdv_graph_lookup(const void* ws, cdv::GraphLayout* out);

void cdv_graph_forget(const void* ws);

// builds on this workspace skip the correlation's processing order (an index no correlation walks; forgotten with the workspace)
void cdv_graph_no_corr_order(const void* ws);

// was the last build on this workspace given the source frames ii (they are then part of the edge records)?
bool cdv_graph_has_ii(const void* ws);

// is the index in the workspace a patch table (cdv_graph_build_table)?
bool cdv_graph_is_table(const void* ws);
// ... and its capacity in slots (0: not a table)
int64_t cdv_graph_table_capacity(const void* ws);

// is the index a usable one: the error state of the ranked build, or of the table build (a negative id, two ids in one
// slot, a patch with more edges than the sort launch serves)
namespace cdv {
__device__ __forceinline__ int graph_error(const int32_t* __restrict__ meta) {
  const int e = meta[GM_ERROR], t0 = meta[GM_TERR], t1 = meta[GM_TERR + 1], g = meta[GM_GEN];
  return meta[GM_MODE] ? ((g & 1) ? t1 : t0) : e;
}
// the same for a caller that knows which form the index has (a kernel argument): every word is read unconditionally, so
// the answer costs ONE memory round trip, not two dependent ones, at the head of a latency-bound kernel
__device__ __forceinline__ int graph_error_of(const int32_t* __restrict__ meta, bool table) {
  const int e = meta[GM_ERROR], t0 = meta[GM_TERR], t1 = meta[GM_TERR + 1], g = meta[GM_GEN];
  return table ? ((g & 1) ? t1 : t0) : e;
}
}  // namespace cdv

// the table build in two halves (graph.hip), for cdv_update_prologue_table
namespace cdv { struct TFillArgs; }
int cdv_graph_table_prepare(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes,
                            int64_t E_max, int64_t k_range, int64_t tab_cap, int64_t* ix, int64_t* jx, void* stream,
                            cdv::TFillArgs* fill, int* fill_blocks, const int32_t* dyn = nullptr);
int cdv_graph_table_finish(const cdv::TFillArgs& fill, int fill_blocks, void* ws, int64_t E_max, int64_t k_range, int64_t* ix,
                           int64_t* jx, const float* poses, const float* patches, const float* intr, float* coords_out,
                           bool with_stream, void* stream);

// the index build in two halves (graph.hip), for cdv_update_prologue
namespace cdv { struct HistArgs; }
int cdv_graph_prepare(const int64_t* jj, const int64_t* kk, int64_t E, void* ws, size_t ws_bytes, int64_t E_max,
                      int64_t k_range, int64_t* ix, int64_t* jx, void* stream, cdv::HistArgs* hist, int* hist_blocks);
// ii (optional, may be NULL): the source frame of every edge, copied into the CSR records for the bundle adjustment
int cdv_graph_finish(const int64_t* ii, const int64_t* jj, const int64_t* kk, int64_t E, void* ws, int64_t E_max,
                     int64_t k_range, int hist_blocks, int64_t* ix, int64_t* jx, void* stream);
