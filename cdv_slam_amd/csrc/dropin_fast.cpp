// dropin_fast.cpp -- the per-call bookkeeping of the drop-in modules in compiled code (CPython extension, no device code).
//
// The reference binds its kernels with a pybind11 / torch extension (cdvslam/altcorr/correlation.cpp:57-63,
// cdvslam/fastba/ba.cpp:183-188): a call costs a few microseconds of host time there.  The drop-in modules of this package
// (cdv_slam_amd/dropin/*.py -> ops.py) serve the same names over the C ABI of libcdvslam_hip.so and keep, per call, what
// makes an unchanged slam.py fast: shadows of the planar rings and tiles re-synchronised when their version counters move,
// the pairing of the two per-level cuda_corr.forward calls into one launch, one patch-graph index per set of edge tensors
// shared by neighbors() and BA().  In Python that bookkeeping is 15-55 us per call (profiles/r5_dropin_host.log); this file
// is the SAME bookkeeping for the steady state only -- the call sequence of slam.py:316-329,486-515 on tensors that look
// exactly like those of the update before -- at 2-4 us per call.  Everything else returns None and the Python code in
// ops.py (the authority on semantics: learning the pairing, allocation, growth, warnings, fallbacks, errors) serves the call.
//
// State is handed over ("armed") by ops.py after it has served a complete pair / a complete neighbors + BA itself, and
// taken back ("disarmed") before any Python code touches the shared shadows or workspaces again.  Identity of a tensor =
// the memory it views + the version counter of that memory (ops._ident): the reference hands over fresh views and fresh
// torch.cat results at every update, never the same Python objects.  Every tensor whose identity is remembered is HELD, so
// an address the allocator recycles cannot impersonate it.
//
// The C-ABI entry points are bound by address from Python (fast.bind: the library _lib.load() opened, CDV_LIB included),
// so this module links against nothing but torch / Python.
#include <torch/extension.h>

#include <cstdint>
#include <string>
#include <vector>

namespace py = pybind11;
using at::Tensor;

namespace {

struct ShadowRing {   // cdv_shadow_ring (include/cdvslam_hip.h)
  const void* src_nchw; void* dst_nhwc; void* ws;
  int64_t N;
  int32_t C, H, W, parity;
};

struct Abi {   // include/cdvslam_hip.h, by address
  int (*shadows_sync)(const ShadowRing*, int, const void*, void*, int64_t, int, void*) = nullptr;
  int (*corr_fused)(const void*, const void*, const void*, const float*, const int64_t*, const int64_t*, const int32_t*, void*,
                    int64_t, int64_t, int64_t, int, int, int, int, int, float, float, int, int64_t, int64_t, int, void*) = nullptr;
  int (*corr_level_checked_interleaved)(const void*, const void*, const float*, const float*, float, const int64_t*,
                                        const int64_t*, void*, int, int64_t, int64_t, int64_t, int, int, int, float, int64_t,
                                        int64_t, int, void*) = nullptr;
  int (*graph_build_table)(const int64_t*, const int64_t*, const int64_t*, int64_t, void*, size_t, int64_t, int64_t, int64_t,
                           int64_t*, int64_t*, void*) = nullptr;
  size_t (*ba_workspace_bytes)(int64_t, int64_t, int) = nullptr;
  int (*ba_forward)(float*, float*, const float*, const float*, const float*, const float*, const int64_t*, const int64_t*,
                    const int64_t*, int64_t, int, int, int, int, const void*, void*, size_t, int64_t, float*, void*) = nullptr;
  int (*transform)(const float*, const float*, const float*, const int64_t*, const int64_t*, const int64_t*, int64_t, int, int,
                   float*, float*, float*, float*, float*, float*, void*) = nullptr;
  bool bound = false;
} abi;

template <typename F>
void take(const py::dict& d, const char* name, F& f) {
  if (!d.contains(name)) throw std::runtime_error(std::string("dropin_fast.bind: missing ") + name);
  f = reinterpret_cast<F>(d[name].cast<uintptr_t>());
}

void bind(const py::dict& d) {
  take(d, "cdv_shadows_sync", abi.shadows_sync);
  take(d, "cdv_corr_fused", abi.corr_fused);
  take(d, "cdv_corr_level_checked_interleaved", abi.corr_level_checked_interleaved);
  take(d, "cdv_graph_build_table", abi.graph_build_table);
  take(d, "cdv_ba_workspace_bytes", abi.ba_workspace_bytes);
  take(d, "cdv_ba_forward", abi.ba_forward);
  take(d, "cdv_transform", abi.transform);
  abi.bound = true;
}

struct Ident {   // a contiguous index tensor: the memory, its version, its length (dtype / contiguity are checked per call)
  const void* ptr = nullptr;
  int64_t version = -1, numel = -1;
  bool operator==(const Ident& o) const { return ptr == o.ptr && version == o.version && numel == o.numel; }
};
inline Ident ident_of(const Tensor& t) { return Ident{t.data_ptr(), (int64_t)t._version(), t.numel()}; }

inline bool on_dev(const Tensor& t, int dev) { return t.is_cuda() && t.get_device() == dev; }
inline bool index_like(const Tensor& t, int dev, int64_t E) {
  return on_dev(t, dev) && t.scalar_type() == at::kLong && t.is_contiguous() && t.numel() == E;
}
inline bool f32_rows(const Tensor& t, int dev) { return on_dev(t, dev) && t.scalar_type() == at::kFloat && t.is_contiguous(); }

// ------------------------------------------------------------------------------------------------------------------
// cuda_corr.forward: the pair of per-level calls (ops._LevelPairing, armed state only)
// ------------------------------------------------------------------------------------------------------------------
struct Ring {   // a planar feature ring of the caller [1, N, C, H, W] f16 and its padded channels-last shadow (ops.NhwcCache)
  Tensor src, shadow, ws;
  const void* ptr = nullptr;
  std::vector<int64_t> sizes;
  int64_t version = -1, N = 0;
  int parity = 0, C = 0, H = 0, W = 0;
  bool is(const Tensor& t) const {
    return t.data_ptr() == ptr && t.scalar_type() == at::kHalf && t.is_contiguous() && t.sizes() == at::IntArrayRef(sizes);
  }
};

struct Pair {
  bool armed = false;
  int64_t token = 0;
  int dev = -1, ratio = 0;
  Ring A, B;
  Tensor tiles_src, tiles_pm;          // the caller's planar tiles [1, Ng, C, 3, 3] (latest view, held) and their pixel-major shadow
  const void* tiles_ptr = nullptr;
  std::vector<int64_t> tiles_sizes;
  int64_t tiles_version = -1, Ng = 0;
  // the first call of a pair that computed both levels, waiting for its second
  bool pending = false;
  Tensor p_coords, p_buf, p_ii, p_jj;
  Ident p_iid, p_jid;
  int64_t p_E = 0, p_tiles_version = -1;
} pr;

void ring_from(Ring& r, const py::dict& d) {
  r.src = d["src"].cast<Tensor>();
  r.shadow = d["shadow"].cast<Tensor>();
  r.ws = d["ws"].cast<Tensor>();
  r.version = d["version"].cast<int64_t>();
  r.parity = d["parity"].cast<int>();
  r.ptr = r.src.data_ptr();
  r.sizes = r.src.sizes().vec();
  TORCH_CHECK(r.src.dim() == 5 && r.src.size(0) == 1 && r.src.is_contiguous() && r.src.scalar_type() == at::kHalf,
              "dropin_fast: a ring must be a contiguous [1, N, C, H, W] half tensor");
  r.N = r.src.size(1); r.C = (int)r.src.size(2); r.H = (int)r.src.size(3); r.W = (int)r.src.size(4);
}

int64_t next_token = 1;

int64_t arm_pair(const py::dict& d) {
  TORCH_CHECK(abi.bound, "dropin_fast: bind() first");
  Pair p;
  ring_from(p.A, d["A"].cast<py::dict>());
  ring_from(p.B, d["B"].cast<py::dict>());
  p.ratio = d["ratio"].cast<int>();
  p.tiles_src = d["tiles_src"].cast<Tensor>();
  p.tiles_pm = d["tiles_pm"].cast<Tensor>();
  p.tiles_version = d["tiles_version"].cast<int64_t>();
  TORCH_CHECK(p.tiles_src.dim() == 5 && p.tiles_src.size(0) == 1 && p.tiles_src.size(3) == 3 && p.tiles_src.size(4) == 3 &&
                  p.tiles_src.is_contiguous() && p.tiles_src.scalar_type() == at::kHalf,
              "dropin_fast: tiles must be a contiguous [1, Ng, C, 3, 3] half tensor");
  p.tiles_ptr = p.tiles_src.data_ptr();
  p.tiles_sizes = p.tiles_src.sizes().vec();
  p.Ng = p.tiles_src.size(1);
  TORCH_CHECK(p.A.C == p.B.C && p.A.C == p.tiles_src.size(2) && p.A.N == p.B.N && p.B.H * p.ratio == p.A.H &&
                  p.B.W * p.ratio == p.A.W && p.ratio >= 2, "dropin_fast: the two rings are not a pyramid");
  p.dev = p.A.src.get_device();
  p.armed = true;
  p.token = next_token++;
  pr = std::move(p);
  return pr.token;
}

py::object disarm_pair() {
  if (!pr.armed) return py::none();
  py::dict out;
  out["A_version"] = pr.A.version; out["A_parity"] = pr.A.parity;
  out["B_version"] = pr.B.version; out["B_parity"] = pr.B.parity;
  out["tiles_src"] = pr.tiles_src; out["tiles_version"] = pr.tiles_version;
  pr = Pair();
  return std::move(out);
}

void drop_pending() {
  pr.pending = false;
  pr.p_coords = Tensor(); pr.p_buf = Tensor(); pr.p_ii = Tensor(); pr.p_jj = Tensor();
}

inline ShadowRing shadow_job(const Ring& r) {
  return ShadowRing{r.ptr, r.shadow.data_ptr(), r.ws.data_ptr(), r.N, r.C, r.H, r.W, r.parity};
}

// -> None (not served: ops.corr_forward goes on) | int (a C-ABI error code) | (view, buffer, level, what was re-synchronised)
py::object corr(int64_t token, const Tensor& fmap1, const Tensor& fmap2, const Tensor& coords, const Tensor& ii, const Tensor& jj,
                int64_t radius, uintptr_t stream_) {
  const bool had = pr.pending;
  pr.pending = false;                                   // whatever this call is, a speculative level only serves the NEXT call
  if (!pr.armed || token != pr.token || radius != 3) return py::none();
  if (fmap1.data_ptr() != pr.tiles_ptr || fmap1.scalar_type() != at::kHalf || !fmap1.is_contiguous() ||
      fmap1.sizes() != at::IntArrayRef(pr.tiles_sizes))
    return py::none();
  if (!f32_rows(coords, pr.dev) || coords.dim() != 5 || coords.size(0) != 1 || coords.size(2) != 2 || coords.size(3) != 3 ||
      coords.size(4) != 3)
    return py::none();
  const int64_t E = coords.size(1);
  if (E <= 0 || !index_like(ii, pr.dev, E) || !index_like(jj, pr.dev, E)) return py::none();
  void* stream = reinterpret_cast<void*>(stream_);
  const int64_t tv = (int64_t)fmap1._version();
  const int C = pr.A.C;
  if (had && pr.B.is(fmap2)) {
    // the second call of the pair: same tiles, same index tensors, unmodified since the first; ring B as it was synchronised
    if (E != pr.p_E || tv != pr.p_tiles_version || !(ident_of(ii) == pr.p_iid) || !(ident_of(jj) == pr.p_jid) ||
        (int64_t)fmap2._version() != pr.B.version)
      return py::none();
    const int rc = abi.corr_level_checked_interleaved(fmap1.data_ptr(), pr.B.shadow.data_ptr(), coords.data_ptr<float>(),
                                                      pr.p_coords.data_ptr<float>(), 1.0f / (float)pr.ratio, ii.data_ptr<int64_t>(),
                                                      jj.data_ptr<int64_t>(), pr.p_buf.data_ptr(), 1, E, pr.Ng, pr.B.N, C, pr.B.H,
                                                      pr.B.W, 1.0f, 0, 0, 0, stream);
    if (rc != 0) return py::int_(rc);
    Tensor buf = pr.p_buf;
    drop_pending();
    return py::make_tuple(buf.select(-1, 1), buf, 1, 0);
  }
  if (!pr.A.is(fmap2)) return py::none();
  // the first call: bring the shadows in step (only what somebody wrote since), both levels in one launch
  int what = 0, rc = 0;
  const int64_t va = (int64_t)fmap2._version(), vb = (int64_t)pr.B.src._version();
  if (va != pr.A.version) what |= 1;
  if (vb != pr.B.version) what |= 2;
  if (tv != pr.tiles_version) what |= 4;
  if (what) {                                            // whatever has to be brought in step: one call, two launches
    ShadowRing jobs[2];
    int n = 0;
    if (what & 1) jobs[n++] = shadow_job(pr.A);
    if (what & 2) jobs[n++] = shadow_job(pr.B);
    rc = abi.shadows_sync(jobs, n, (what & 4) ? fmap1.data_ptr() : nullptr, pr.tiles_pm.data_ptr(), pr.Ng, C, stream);
    if (rc != 0) return py::int_(rc);
    if (what & 1) { pr.A.parity ^= 1; pr.A.version = va; }
    if (what & 2) { pr.B.parity ^= 1; pr.B.version = vb; }
    if (what & 4) pr.tiles_version = tv;
  }
  pr.tiles_src = fmap1;                                  // the view in hand pins the storage the version speaks of
  Tensor buf = at::empty({1, E, 7, 7, 3, 3, 2}, fmap1.options());
  rc = abi.corr_fused(pr.tiles_pm.data_ptr(), pr.A.shadow.data_ptr(), pr.B.shadow.data_ptr(), coords.data_ptr<float>(),
                      ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), nullptr, buf.data_ptr(), E, pr.Ng, pr.A.N, C, pr.A.H, pr.A.W,
                      pr.B.H, pr.B.W, 1.0f, (float)pr.ratio, 2, 0, 0, 1, stream);
  if (rc != 0) return py::int_(rc);
  pr.pending = true;
  pr.p_coords = coords; pr.p_buf = buf; pr.p_ii = ii; pr.p_jj = jj;
  pr.p_iid = ident_of(ii); pr.p_jid = ident_of(jj);
  pr.p_E = E; pr.p_tiles_version = tv;
  return py::make_tuple(buf.select(-1, 0), buf, 0, what);
}

// ------------------------------------------------------------------------------------------------------------------
// cuda_ba.neighbors / cuda_ba.forward on the per-device table index (ops.graph_for / ops.ba_forward, armed state only)
// ------------------------------------------------------------------------------------------------------------------
struct Graph {
  bool armed = false;
  int64_t token = 0;
  int dev = -1;
  Tensor ws, ba_ws;
  int64_t ws_bytes = 0, E_cap = 0, k_range = 0, tab_cap = 0, ppf = 0;
  bool has_key = false, has_ii = false, has_nbr = false;
  Tensor kjj, kkk, kii, ix, jx;
  Ident jid, kid, iid;
  const volatile int32_t* events = nullptr;            // four pinned counters of this index's EventBlock
  int32_t seen[4] = {0, 0, 0, 0};
  int64_t need_E = -1, need_U = -1, need_N = -1;        // the last workspace size asked of the library
  size_t need = 0;
} gr;

int64_t arm_graph(const py::dict& d) {
  TORCH_CHECK(abi.bound, "dropin_fast: bind() first");
  Graph g;
  g.ws = d["ws"].cast<Tensor>();
  g.ba_ws = d["ba_ws"].cast<Tensor>();
  g.ws_bytes = d["ws_bytes"].cast<int64_t>();
  g.E_cap = d["E_cap"].cast<int64_t>();
  g.k_range = d["k_range"].cast<int64_t>();
  g.tab_cap = d["table_capacity"].cast<int64_t>();
  g.ppf = d["ppf"].cast<int64_t>();
  g.events = reinterpret_cast<const volatile int32_t*>(d["events_ptr"].cast<uintptr_t>());
  TORCH_CHECK(g.events != nullptr && g.tab_cap >= 1 && g.ws.is_cuda() && g.ba_ws.is_cuda(), "dropin_fast.arm_graph: bad state");
  const auto seen = d["events_seen"].cast<std::vector<int64_t>>();     // what Python has already looked at (and warned about)
  TORCH_CHECK(seen.size() == 4, "dropin_fast.arm_graph: events_seen");
  for (int i = 0; i < 4; i++) g.seen[i] = (int32_t)seen[i];
  g.dev = g.ws.get_device();
  if (!d["jj"].is_none()) {
    g.kjj = d["jj"].cast<Tensor>(); g.kkk = d["kk"].cast<Tensor>();
    g.jid = Ident{g.kjj.data_ptr(), d["jj_version"].cast<int64_t>(), g.kjj.numel()};
    g.kid = Ident{g.kkk.data_ptr(), d["kk_version"].cast<int64_t>(), g.kkk.numel()};
    g.has_key = true;
    if (!d["ii"].is_none()) {
      g.kii = d["ii"].cast<Tensor>();
      g.iid = Ident{g.kii.data_ptr(), d["ii_version"].cast<int64_t>(), g.kii.numel()};
      g.has_ii = true;
    }
    if (!d["ix"].is_none()) { g.ix = d["ix"].cast<Tensor>(); g.jx = d["jx"].cast<Tensor>(); g.has_nbr = true; }
  }
  g.armed = true;
  g.token = next_token++;
  gr = std::move(g);
  return gr.token;
}

// -> None (nothing was armed) | what the workspace holds now: the tensors the index was built from, their versions at the
// build, the neighbors it produced (ops._disarm_graph writes it back into the GraphIndex)
py::object disarm_graph() {
  if (!gr.armed) return py::none();
  py::dict out;
  out["has_key"] = gr.has_key;
  if (gr.has_key) {
    out["jj"] = gr.kjj; out["kk"] = gr.kkk;
    out["jj_version"] = gr.jid.version; out["kk_version"] = gr.kid.version;
    if (gr.has_ii) { out["ii"] = gr.kii; out["ii_version"] = gr.iid.version; } else { out["ii"] = py::none(); }
    if (gr.has_nbr) { out["ix"] = gr.ix; out["jx"] = gr.jx; } else { out["ix"] = py::none(); }
  }
  gr = Graph();
  return std::move(out);
}

inline bool events_moved() {
  bool moved = false;
  for (int i = 0; i < 4; i++) moved |= gr.events[i] != gr.seen[i];
  return moved;
}

// -> None | int (error) | (ix, jx, builds enqueued)
py::object neighbors(int64_t token, const Tensor& kk, const Tensor& jj, uintptr_t stream_) {
  if (!gr.armed || token != gr.token) return py::none();
  const int64_t E = kk.numel();
  if (E <= 0 || E > gr.E_cap || !index_like(kk, gr.dev, E) || !index_like(jj, gr.dev, E) || events_moved()) return py::none();
  const Ident jid = ident_of(jj), kid = ident_of(kk);
  if (gr.has_key && jid == gr.jid && kid == gr.kid) {
    if (!gr.has_nbr) return py::none();
    return py::make_tuple(gr.ix, gr.jx, 0);
  }
  Tensor ix = at::empty({E}, kk.options()), jx = at::empty({E}, kk.options());
  const int rc = abi.graph_build_table(nullptr, jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(), E, gr.ws.data_ptr(),
                                       (size_t)gr.ws_bytes, gr.E_cap, gr.k_range, gr.tab_cap, ix.data_ptr<int64_t>(),
                                       jx.data_ptr<int64_t>(), reinterpret_cast<void*>(stream_));
  if (rc != 0) { gr.has_key = false; return py::int_(rc); }
  gr.kjj = jj; gr.kkk = kk; gr.kii = Tensor();
  gr.jid = jid; gr.kid = kid;
  gr.has_key = true; gr.has_ii = false; gr.has_nbr = true;
  gr.ix = ix; gr.jx = jx;
  return py::make_tuple(ix, jx, 1);
}

// -> None | int (error) | (builds enqueued,)
py::object ba(int64_t token, const Tensor& poses, const Tensor& patches, const Tensor& intrinsics, const Tensor& target,
              const Tensor& weight, const Tensor& lmbda, const Tensor& ii, const Tensor& jj, const Tensor& kk, int64_t PPF, int64_t t0,
              int64_t t1, int64_t iterations, uintptr_t stream_) {
  if (!gr.armed || token != gr.token) return py::none();
  const int64_t N = t1 - t0, E = kk.numel();
  if (N < 1 || N > 32 || E <= 0 || E > gr.E_cap || iterations < 0 || (PPF > 0 ? PPF : 0) != gr.ppf) return py::none();
  const int dev = gr.dev;
  if (!f32_rows(poses, dev) || !f32_rows(patches, dev) || !f32_rows(intrinsics, dev) || !f32_rows(target, dev) ||
      !f32_rows(weight, dev) || !f32_rows(lmbda, dev) || lmbda.numel() < 1 || patches.dim() < 3)
    return py::none();
  if (!index_like(ii, dev, E) || !index_like(jj, dev, E) || !index_like(kk, dev, E) || target.numel() != 2 * E ||
      weight.numel() != 2 * E || events_moved())
    return py::none();
  const int64_t P = patches.size(-1);
  if (P < 1 || patches.size(-2) != P) return py::none();
  int64_t U_max = std::min<int64_t>(E, patches.numel() / (3 * P * P));
  U_max = std::max<int64_t>(U_max, gr.tab_cap);         // the slab kernels work through every slot of the table
  if (gr.need_E != E || gr.need_U != U_max || gr.need_N != N) {
    gr.need = abi.ba_workspace_bytes(E, U_max, (int)N);
    gr.need_E = E; gr.need_U = U_max; gr.need_N = N;
  }
  if ((size_t)gr.ba_ws.numel() < gr.need) return py::none();      // growth is Python's business
  void* stream = reinterpret_cast<void*>(stream_);
  int built = 0;
  const Ident jid = ident_of(jj), kid = ident_of(kk);
  bool current = gr.has_key && jid == gr.jid && kid == gr.kid;
  if (current && gr.has_ii) current = ident_of(ii) == gr.iid;     // an index built WITHOUT ii serves any ii (read per edge)
  if (!current) {
    const int rc = abi.graph_build_table(ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(), E, gr.ws.data_ptr(),
                                         (size_t)gr.ws_bytes, gr.E_cap, gr.k_range, gr.tab_cap, nullptr, nullptr, stream);
    if (rc != 0) { gr.has_key = false; return py::int_(rc); }
    gr.kjj = jj; gr.kkk = kk; gr.kii = ii;
    gr.jid = jid; gr.kid = kid; gr.iid = ident_of(ii);
    gr.has_key = true; gr.has_ii = true; gr.has_nbr = false;
    gr.ix = Tensor(); gr.jx = Tensor();
    built = 1;
  }
  const int rc = abi.ba_forward(poses.data_ptr<float>(), patches.data_ptr<float>(), intrinsics.data_ptr<float>(),
                                target.data_ptr<float>(), weight.data_ptr<float>(), lmbda.data_ptr<float>(), ii.data_ptr<int64_t>(),
                                jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(), E, (int)P, (int)t0, (int)t1, (int)iterations,
                                gr.ws.data_ptr(), gr.ba_ws.data_ptr(), (size_t)gr.ba_ws.numel(), U_max, nullptr, stream);
  if (rc != 0) return py::int_(rc);
  return py::make_tuple(built);
}

// ------------------------------------------------------------------------------------------------------------------
// pops.transform, coordinates only (projective_ops.py:53-113 / SLAM.reproject slam.py:325-329): stateless
// ------------------------------------------------------------------------------------------------------------------
// -> None | int (error) | coords [1, E, 2, P, P]
py::object transform(const Tensor& poses, const Tensor& patches, const Tensor& intrinsics, const Tensor& ii, const Tensor& jj,
                     const Tensor& kk, uintptr_t stream_) {
  if (!abi.bound || !poses.is_cuda()) return py::none();
  const int dev = poses.get_device();
  const int64_t E = ii.numel();
  if (!f32_rows(poses, dev) || poses.dim() != 3 || poses.size(0) != 1 || !f32_rows(patches, dev) || patches.dim() != 5 ||
      !f32_rows(intrinsics, dev) || !index_like(ii, dev, E) || !index_like(jj, dev, E) || !index_like(kk, dev, E))
    return py::none();
  const int64_t P = patches.size(-1);
  Tensor coords = at::empty({1, E, 2, P, P}, poses.options());
  const int rc = abi.transform(poses.data_ptr<float>(), patches.data_ptr<float>(), intrinsics.data_ptr<float>(),
                               ii.data_ptr<int64_t>(), jj.data_ptr<int64_t>(), kk.data_ptr<int64_t>(), E, (int)P, 1, coords.data_ptr<float>(),
                               nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<void*>(stream_));
  if (rc != 0) return py::int_(rc);
  return py::cast(coords);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled steady-state bookkeeping of the cdv_slam_amd drop-in modules (see dropin_fast.cpp)";
  m.def("bind", &bind);
  m.def("arm_pair", &arm_pair);
  m.def("disarm_pair", &disarm_pair);
  m.def("drop_pending", &drop_pending);
  m.def("corr", &corr);
  m.def("arm_graph", &arm_graph);
  m.def("disarm_graph", &disarm_graph);
  m.def("neighbors", &neighbors);
  m.def("ba", &ba);
  m.def("transform", &transform);
}
