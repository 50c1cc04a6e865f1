// pt_query.hip -- exact k-nearest-neighbour search on the cell grid, for gfx950 (MI355X).
//
// Replaces the reference's query loop
//     K_neighbor_search search(tree, vertices[...], K);  for (it = search.begin(); ...)
// (reference src/pointsTransfer.cpp:462-479; CGAL Orthogonal_k_neighbor_search, eps = 0, results ascending)
// with one launch over all targets.  Metric: reference src/Distance.h:6-11, evaluated in fp64 as
// (dx*dx + dy*dy) + dz*dz with contraction off -- the same 3 mul + 2 add the reference's flags produce.
// Cell pruning is Distance::min_distance_to_rectangle (reference src/Distance.h:27-57) applied to cell
// boxes; ring termination is the same bound applied to the faces of the box already scanned.
//
// Three kernels, all exact (a third, knn_wave_kernel -- ONE WAVE PER TARGET, for dense neighbourhoods of clouds with strong density
// contrast, k > 16 leftovers of the tile kernel and surfaces -- sits between the two halves):
//   knn_tile_kernel  (second half of this file) fp32 clouds, unbounded queries, k <= 32 -- the throughput path.  One
//                    workgroup per 8^3-cell block stages the 10^3-cell region around it in LDS and ranks it with a DPP quad
//                    per target: fp32 bound -> queue -> exact fp64 re-rank.  What ring 1 cannot settle goes to a todo list.
//   knn_kernel       (first half) everything else: fp64 clouds, radius-bounded multi-GPU queries, the todo list.
//                    8 lanes per target, 8 targets per wave64, 32 per 256-thread workgroup:
//   - ring 1 (the 3x3x3 cells around the target) is 9 x-rows of 3 cells.  The group's lanes look the rows'
//     cell ranges up in parallel (one latency for all of them), then the rows are processed centre first;
//     a row's surviving cells are flattened into one index space so that the 8 lanes always read 8 consecutive
//     candidates (128-B lines of 16-B records), a whole row's records are requested in one batch, and the next
//     row's batch is already in flight while the current one is ranked;
//   - the running top-k lives in registers, distributed over the group's lanes (lane L holds ranks
//     [L*KPL, (L+1)*KPL), KPL = ceil(k/8)), ordered by the total order (d2, original index);
//   - a candidate is offered with one group ballot, accepted through one ballot bit of the lane that holds rank
//     k-1, and inserted as a one-position shift across lanes done with DPP row operations (no LDS traffic);
//   - rings >= 2 (needed by the few targets whose k-th neighbour is farther than one cell) use a plain
//     row-by-row walk.
// Control flow is uniform inside a group (all lanes of a group / quad take every branch together), so cross-lane
// operations never see an inactive partner; different groups of a wave diverge freely.
#include "pt_internal.h"
#include <type_traits>

namespace {

constexpr int WG = 256;
constexpr int GL = 8;   // lanes per target
constexpr int PT_RING_LIMIT = 8;   // least number of rings walked shell by shell before the group kernel sweeps the blocks instead

__device__ inline bool key_lt(double ad, uint32_t ai, double bd, uint32_t bi) { return ad < bd || (ad == bd && ai < bi); }

template <class Rec>
__device__ inline double dist2(const double (&q)[3], const Rec& r) {
#pragma clang fp contract(off)
  const double dx = q[0] - (double)r.x;
  const double dy = q[1] - (double)r.y;
  const double dz = q[2] - (double)r.z;
  return (dx * dx + dy * dy) + dz * dz;     // reference src/Distance.h:10, left to right, unfused
}

// ---- DPP helpers: data movement inside the 8-lane group without touching LDS ---------------------------------
template <int CTRL>
__device__ inline uint32_t dpp_u32(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ inline double dpp_f64(double v) {
  const uint32_t lo = dpp_u32<CTRL>((uint32_t)__double2loint(v)), hi = dpp_u32<CTRL>((uint32_t)__double2hiint(v));
  return __hiloint2double((int)hi, (int)lo);
}
constexpr int DPP_SHR1 = 0x111;          // row_shr:1      lane i <- lane i-1
constexpr int DPP_QUAD3 = 0xFF;          // quad_perm [3,3,3,3]
constexpr int DPP_HMIRROR = 0x141;       // row_half_mirror: lane i <- lane 7-i of the same 8 lanes
// broadcast lane 7 (resp. lane 0) of every 8-lane group to the whole group; L = lane index inside the group
__device__ inline uint32_t bcast7(uint32_t v, int L) { const uint32_t a = dpp_u32<DPP_QUAD3>(v), b = dpp_u32<DPP_HMIRROR>(a); return L < 4 ? b : a; }
__device__ inline double bcast7(double v, int L) { const double a = dpp_f64<DPP_QUAD3>(v), b = dpp_f64<DPP_HMIRROR>(a); return L < 4 ? b : a; }

// ---- the running top list of one target, spread over the 8 lanes of its group ------------------------------------
template <int KPL>
struct TopList {
  double ld[KPL];
  uint32_t li[KPL];
  double lim_d, bnd_d;     // acceptance limit = min(entry of rank k-1, caller's bound); index NOIDX when it is a bare bound
  uint32_t lim_i;
  int L, hl, hr;
  uint32_t notfirst;       // 0 for lane 0 of the group, 1 otherwise
  bool fullk;              // k == 8*KPL: the rank k-1 entry is the last entry of lane 7
#ifdef PT_VISITS
  uint32_t nv;             // instrumented build (tools/probe_visits.py): offers made (8 records each)
#endif

  __device__ void init(int lane_in_group, int k, double bound) {
#pragma unroll
    for (int j = 0; j < KPL; ++j) { ld[j] = INFINITY; li[j] = PT_NOIDX_U; }
    L = lane_in_group;
    notfirst = lane_in_group != 0 ? 1u : 0u;
    hl = (k - 1) / KPL;
    hr = (k - 1) % KPL;
    fullk = (k == GL * KPL);
    bnd_d = bound;
    lim_d = bound;
    lim_i = PT_NOIDX_U;
#ifdef PT_VISITS
    nv = 0;
#endif
  }
  // cheap pre-test against the cached limit (may be stale, i.e. too permissive -- never too strict)
  __device__ bool may_accept(double d, uint32_t i) const { return key_lt(d, i, lim_d, lim_i); }

  // Try to insert (xd, xi), known by every lane of the group.  The exact acceptance test is the comparison with the
  // entry of rank k-1, which lives in lane hl: its verdict reaches the group through one ballot bit, so the k-th
  // entry itself never has to be broadcast.  Returns whether the list changed.
  __device__ bool try_insert(double xd, uint32_t xi, int gshift) {
    bool cj[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) cj[j] = key_lt(xd, xi, ld[j], li[j]);
    bool csel = cj[0];
#pragma unroll
    for (int j = 1; j < KPL; ++j) if (hr == j) csel = cj[j];
    const bool acc = ((__ballot(csel) >> (gshift + hl)) & 1ull) != 0ull;   // group-uniform
    if (!acc) return false;
    // one-position shift: the lane below hands over its last entry if the new key sorts before it.
    // (every cross-lane move is executed by ALL lanes of the group: never under a lane-dependent branch, or the
    //  source lane may be masked off; lane 0's incoming value is discarded arithmetically instead)
    const double pd = dpp_f64<DPP_SHR1>(ld[KPL - 1]);
    const uint32_t pi = dpp_u32<DPP_SHR1>(li[KPL - 1]);
    const bool pc = (dpp_u32<DPP_SHR1>(cj[KPL - 1] ? 1u : 0u) & notfirst) != 0u;
#pragma unroll
    for (int j = KPL - 1; j >= 1; --j) {
      if (cj[j - 1]) { ld[j] = ld[j - 1]; li[j] = li[j - 1]; }
      else if (cj[j]) { ld[j] = xd; li[j] = xi; }
    }
    if (pc) { ld[0] = pd; li[0] = pi; }
    else if (cj[0]) { ld[0] = xd; li[0] = xi; }
    return true;
  }

  // re-read the limit after insertions: the entry of rank k-1, unless the caller's bound is tighter
  __device__ void refresh_limit() {
    double kd;
    uint32_t ki;
    if (fullk) {
      kd = bcast7(ld[KPL - 1], L);
      ki = bcast7(li[KPL - 1], L);
    } else {
      kd = ld[0];
      ki = li[0];
#pragma unroll
      for (int j = 1; j < KPL; ++j) if (hr == j) { kd = ld[j]; ki = li[j]; }
      kd = __shfl(kd, hl, GL);
      ki = __shfl(ki, hl, GL);
    }
    if (key_lt(kd, ki, bnd_d, PT_NOIDX_U)) { lim_d = kd; lim_i = ki; }
    else { lim_d = bnd_d; lim_i = PT_NOIDX_U; }
  }

  // offer one candidate per lane (d = +inf / id = NOIDX for lanes without one)
  __device__ void offer(double d, uint32_t id, int gshift) {
#ifdef PT_VISITS
    ++nv;
#endif
    const bool pass = may_accept(d, id) && !(d > bnd_d);
    uint32_t mask = (uint32_t)(__ballot(pass) >> gshift) & 0xFFu;
    if (mask) {
      bool changed = false;
      do {
        const int t = __ffs(mask) - 1;
        mask &= mask - 1;
        const double xd = __shfl(d, t, GL);
        const uint32_t xi = __shfl(id, t, GL);
        changed |= try_insert(xd, xi, gshift);
      } while (mask);
      if (changed) refresh_limit();
    }
  }
};

// geometry of one target relative to the grid
struct TargetGeom {
  double q[3], u[3];
  int c[3];
  double h2;
  // distance (cell units, >= 0) from the target to the cell interval [lo, hi] along axis a, minus the slack
  __device__ double gap(int a, int lo, int hi) const {
    const double g = fmax((double)lo - u[a], u[a] - (double)(hi + 1)) - PT_CELL_EPS;
    return fmax(g, 0.0);
  }
};

__device__ inline uint32_t cell_key(const GridParams& gp, int x, int y, int z) {
  return (pt_block_id(gp.mdim, x, y, z) << 9) + pt_local_cell(x, y, z);
}

// =====================================================================================================================
// Search over the REFINED grid (pt_refine.hip), used by the group kernel (knn_kernel<.., HIER = true>): a cell that carries a node
// is not scanned end to end but descended into.  Inside a node the 64 rows of sub-cells are tested against the current bound
// eight at a time (one lane each), the surviving rows are cut to the sub-cells the bound still reaches, leaf sub-cells are scanned
// as before and sub-cells that are nodes themselves are descended into the same way (PT_REFINE_DEPTH levels).  The sub-cell that
// holds the target is visited FIRST on every level, so the bound is tight before the neighbours are looked at; it is skipped when
// the sweep over the rows comes by, so no point is ever offered twice.  Exact for the same reason the group kernel is: a box is
// skipped only if Distance::min_distance_to_rectangle (reference src/Distance.h:27-57) of it exceeds the current k-th distance.
template <class Rec, int KPL>
struct HierScan {
  const GridParams& gp;
  const Rec* __restrict__ src;
  const uint32_t* __restrict__ nodes;
  const TargetGeom& T;
  TopList<KPL>& top;
  int gshift;

  __device__ void range(uint32_t s, uint32_t e) {
    for (uint32_t base = s; base < e; base += GL) {
      const uint32_t p = base + (uint32_t)top.L;
      double d = INFINITY;
      uint32_t id = PT_NOIDX_U;
      if (p < e) { const Rec r = src[p]; d = dist2(T.q, r); id = r.id; }
      top.offer(d, id, gshift);
    }
  }
  // squared distance (cell units) from the target to the interval [lo, hi] on axis a, under-estimated by the slack
  __device__ double gap2(int a, double lo, double hi) const {
    const double g = fmax(fmax(lo - T.u[a], T.u[a] - hi) - PT_CELL_EPS, 0.0);
    return g * g;
  }
  template <int DEPTH>
  __device__ void node(uint32_t nid) {
    const uint32_t* __restrict__ N = nodes + (size_t)(nid - 1u) * PT_NODE_WORDS;
    const double* hd = reinterpret_cast<const double*>(N);
    const double ox = hd[0], oy = hd[1], oz = hd[2], inv = hd[3], w = hd[4];        // w = 1 / inv: sub-cell side in cell units (a power of 1/8)
    // the sub-cell the target falls in, if it is inside this node's box
    const double rx = (T.u[0] - ox) * inv, ry = (T.u[1] - oy) * inv, rz = (T.u[2] - oz) * inv;
    const bool inside = rx >= 0.0 && rx < 8.0 && ry >= 0.0 && ry < 8.0 && rz >= 0.0 && rz < 8.0;
    const uint32_t own = inside ? (uint32_t)(((int)rz << 6) | ((int)ry << 3) | (int)rx) : 0xFFFFFFFFu;
    // per-axis gaps of the eight slabs of sub-cells, one per lane: every box test below is two or three shuffles and adds
    const double fl = (double)top.L;
    const double gxl = gap2(0, ox + fl * w, ox + (fl + 1.0) * w), gyl = gap2(1, oy + fl * w, oy + (fl + 1.0) * w), gzl = gap2(2, oz + fl * w, oz + (fl + 1.0) * w);
    // the rows (sy, sz) that can hold anything under the bound as it is now -- geometry only, no memory touched; lane L tests
    // the rows with sy = L, one sz per step
    uint32_t live_lo = 0, live_hi = 0;                     // bit sz * 8 + sy, group-uniform
#pragma unroll
    for (int sz = 0; sz < 8; ++sz) {
      const bool ok = !((gyl + __shfl(gzl, sz, GL)) * T.h2 > top.lim_d);
      const uint32_t m8 = (uint32_t)((__ballot(ok) >> gshift) & 0xFFull);
      if (sz < 4) live_lo |= m8 << (8 * sz); else live_hi |= m8 << (8 * (sz - 4));
    }
    live_lo &= N[PT_NODE_ROWMASK];                         // ... and are not empty (the node's row mask, next to its header)
    live_hi &= N[PT_NODE_ROWMASK + 1];
    // Sweep: first the target's own sub-cell alone (so that the bound is tight before anything else is looked at), then the live
    // rows of eight sub-cells; the own sub-cell is skipped when its row comes by.  One code path serves both, so that the scan and
    // the descent are instantiated once per level.  A row's nine starts and eight child links are fetched by the eight lanes in
    // ONE go (a single memory latency per row) and handed round by shuffles.
    bool first = inside;
    while (first || (live_lo | live_hi)) {                  // group-uniform
      int r2, xa, xb;
      if (first) { r2 = (int)(own >> 3); xa = xb = (int)(own & 7u); }
      else {
        if (live_lo) { r2 = __ffs((int)live_lo) - 1; live_lo &= live_lo - 1; }
        else { r2 = 32 + __ffs((int)live_hi) - 1; live_hi &= live_hi - 1; }
        const double t2 = __shfl(gyl, r2 & 7, GL) + __shfl(gzl, r2 >> 3, GL);
        if (t2 * T.h2 > top.lim_d) continue;                // the bound may have tightened since the ballots
        xa = 0; xb = 7;
        while (xa <= xb && (__shfl(gxl, xa, GL) + t2) * T.h2 > top.lim_d) ++xa;
        while (xb >= xa && (__shfl(gxl, xb, GL) + t2) * T.h2 > top.lim_d) --xb;
        if (xa > xb) continue;
      }
      const bool sweep = !first;
      first = false;
      const uint32_t stl = N[PT_NODE_START + r2 * 8 + top.L], end8 = N[PT_NODE_START + r2 * 8 + 8];
      uint32_t chl = 0;
      if constexpr (DEPTH + 1 < PT_REFINE_DEPTH) chl = N[PT_NODE_CHILD + r2 * 8 + top.L];
      // leaf sub-cells next to each other are one contiguous run of records, scanned in one go; a sub-cell that is a node, the
      // own sub-cell (already done) and the end of the row cut the run
      uint32_t run_s = 0, run_e = 0;
      for (int x = xa; x <= xb + 1; ++x) {
        const uint32_t sub = (uint32_t)(r2 * 8 + x);
        uint32_t child = 0;
        bool cut = x > xb || (sweep && sub == own);
        if constexpr (DEPTH + 1 < PT_REFINE_DEPTH) {
          if (!cut) {
            child = (uint32_t)__shfl((int)chl, x, GL);
            if (child & PT_LEAF_TRUNC) child = 0u;          // a leaf of identical points with its lowest indices in front (pt_common.h): scanned whole here, which is exact too
            cut = child != 0u;
          }
        }
        if (!cut) {
          if (run_e == run_s) run_s = (uint32_t)__shfl((int)stl, x, GL);
          run_e = x < 7 ? (uint32_t)__shfl((int)stl, x + 1, GL) : end8;
          continue;
        }
        if (run_e > run_s) range(run_s, run_e);
        run_s = run_e = 0;
        if constexpr (DEPTH + 1 < PT_REFINE_DEPTH) { if (child) node<DEPTH + 1>(child); }
      }
    }
  }
};

// Heavy cells met by the group kernel (HIER builds) are not scanned on the spot but remembered -- their key, in the group's slice of
// an LDS list -- and descended into at ONE place of the kernel (the descent is three levels of inlined code: one copy is enough).
constexpr int PEND_CAP = 32;
struct Pending {
  uint32_t* slot;          // this group's PEND_CAP words of LDS
  uint32_t n;              // group-uniform
  uint32_t thr;            // cells with more points than this may carry a node (0xFFFFFFFF: the grid has none)
  __device__ bool heavy(uint32_t s, uint32_t e) const { return e - s > thr; }
  __device__ bool push(uint32_t key, int lane) {           // false: list full, the caller scans the cell linearly (exact, only slower)
    if (n >= (uint32_t)PEND_CAP) return false;
    if (lane == 0) slot[n] = key;
    ++n;
    return true;
  }
};

// generic walk of cells [xa, xb] x {y} x {z} (inside the grid): prune by the box lower bound, then scan block by block
template <class Rec, int KPL>
__device__ void scan_row_generic(const GridParams& gp, const Rec* __restrict__ src, const uint32_t* __restrict__ cs, const TargetGeom& T,
                                 TopList<KPL>& top, int gshift, int xa, int xb, int y, int z, Pending* pend = nullptr) {
  const double gy = T.gap(1, y, y), gz = T.gap(2, z, z);
  const double s2 = gy * gy + gz * gz;
  if (s2 * T.h2 > top.lim_d) return;
  while (xa < xb) { const double g = T.gap(0, xa, xa); if ((g * g + s2) * T.h2 > top.lim_d) ++xa; else break; }
  while (xb > xa) { const double g = T.gap(0, xb, xb); if ((g * g + s2) * T.h2 > top.lim_d) --xb; else break; }
  { const double g = T.gap(0, xa, xb); if ((g * g + s2) * T.h2 > top.lim_d) return; }
  for (int bx = xa >> 3; bx <= (xb >> 3); ++bx) {
    const int pa = max(xa, bx << 3), pb = min(xb, (bx << 3) + 7);
    const uint32_t key = cell_key(gp, pa, y, z);
    uint32_t s = cs[key], e = cs[key + (uint32_t)(pb - pa) + 1u];
    if (pend && pend->heavy(s, e)) {
      // a run that may hold heavy cells: those are set aside for the descent, the light ones in between are scanned here
      const uint32_t e_all = e;
      e = s;
      for (int i = 0; i <= pb - pa; ++i) {
        const uint32_t s1 = cs[key + (uint32_t)i], e1 = cs[key + (uint32_t)i + 1u];
        const bool defer = pend->heavy(s1, e1) && pend->push(key + (uint32_t)i, top.L);
        if (!defer) { e = e1; continue; }
        for (uint32_t base = s; base < e; base += GL) {
          const uint32_t p = base + (uint32_t)top.L;
          double d = INFINITY;
          uint32_t id = PT_NOIDX_U;
          if (p < e) { const Rec r = src[p]; d = dist2(T.q, r); id = r.id; }
          top.offer(d, id, gshift);
        }       // (the light run collected so far)
        s = e = e1;
      }
      (void)e_all;
    }
    for (uint32_t base = s; base < e; base += GL) {
      const uint32_t p = base + (uint32_t)top.L;
      double d = INFINITY;
      uint32_t id = PT_NOIDX_U;
      if (p < e) { const Rec r = src[p]; d = dist2(T.q, r); id = r.id; }
      top.offer(d, id, gshift);
    }
  }
}

// (dy,dz)+1 of the 9 rows of ring 1, packed 2 bits each, centre row first, then faces, then edges:
// dy = 0,-1,1,0,0,-1,1,-1,1 ; dz = 0,0,0,-1,1,-1,-1,1,1
constexpr uint32_t ROW_OY = 139617u, ROW_OZ = 164373u;
__device__ inline int row_dy(int r) { return (int)((ROW_OY >> (2 * r)) & 3u) - 1; }
__device__ inline int row_dz(int r) { return (int)((ROW_OZ >> (2 * r)) & 3u) - 1; }

template <class Rec> struct Batch { static constexpr int N = sizeof(Rec) == 16 ? 4 : 2; };   // steps requested at once

// the cells of one row that survive pruning, flattened: virtual position v -> record index
struct RowPlan {
  uint32_t a0, a1, a2;     // first record of the three cells
  uint32_t n0, n01, T;     // prefix sums of the surviving cells' sizes: n0, n0+n1, n0+n1+n2
  __device__ uint32_t addr(uint32_t v) const { return v < n0 ? a0 + v : (v < n01 ? a1 + (v - n0) : a2 + (v - n01)); }
};

// HIER: the grid carries refined cells (pt_refine.hip: cell_node / nodes / node_thr).  Cells with more than node_thr points are then
// left out of the flat scans, remembered in the group's pending list and descended into (HierScan) at the head of the ring loop.
// heavy / heavy_n / wave_min: targets whose 27 nearest cells hold at least wave_min points are not answered here but listed (their
// position in the sorted target array) for the wave kernel below -- one wave per target pays off where the scans are long.
// heavy: one byte per target position, zeroed by the caller; 1 = wave kernel, 2 = its descending variant (a refined cell among the 27).
struct HierArgs { const uint32_t* cell_node; const uint32_t* nodes; uint32_t thr; uint8_t* heavy; uint32_t wave_min; };
// attribute blend fused into the wave kernel (attr == null: none): the table, its length, the mode and the two outputs
struct WaveBlend { const Attr* attr; uint32_t n_attr; int mode; float* rgb_out; float* nrm_out; };
template <class Rec, int KPL, bool HIER>
__global__ __launch_bounds__(WG, HIER ? (KPL == 4 ? 3 : 4) : 1) void knn_kernel(GridParams gp, const Rec* __restrict__ src, const uint32_t* __restrict__ cs,
                                                 const Rec* __restrict__ tgt, uint32_t m, int k, const double* __restrict__ bound2,
                                                 uint32_t* __restrict__ out_idx, double* __restrict__ out_d2,
                                                 const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n, HierArgs ha) {
  constexpr int NB = Batch<Rec>::N;
  __shared__ uint32_t pend_lds[HIER ? (WG / GL) * PEND_CAP : 1];
  const uint32_t gid = (blockIdx.x * WG + threadIdx.x) / GL;
  if (gid >= (list ? *list_n : m)) return;    // whole groups leave together
  const int L = threadIdx.x & (GL - 1);
  const int gshift = (threadIdx.x & 63) & ~(GL - 1);
  const Rec tr = tgt[list ? list[gid] : gid];  // `list`: positions (in the sorted target array) left over by the tile kernel
#ifdef PT_VISITS
  const unsigned long long pt_t0 = wall_clock64();
#endif
  TargetGeom T;
  T.q[0] = (double)tr.x; T.q[1] = (double)tr.y; T.q[2] = (double)tr.z;
  T.h2 = gp.h * gp.h;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    T.u[a] = (T.q[a] - gp.bbmin[a]) * gp.inv_h;
    T.c[a] = (int)fmin(fmax(T.u[a], 0.0), (double)(gp.dim[a] - 1));
  }
  TopList<KPL> top;
  const double bnd0 = bound2 ? bound2[tr.id] : INFINITY;
  if (bnd0 < 0.0) {            // a negative bound: this target wants nothing from this cloud (pt_stream_query's "not this chunk") -- group-uniform
    const size_t row0 = (size_t)tr.id * (size_t)k;
    for (int e = L; e < k; e += GL) { out_idx[row0 + e] = PT_NOIDX_U; if (out_d2) out_d2[row0 + e] = INFINITY; }
    return;
  }
  top.init(L, k, bnd0);
  const int c0 = T.c[0], c1 = T.c[1], c2 = T.c[2];
  Pending pend{&pend_lds[HIER ? (threadIdx.x / GL) * PEND_CAP : 0], 0u, HIER ? ha.thr : 0xFFFFFFFFu};
  Pending* const pp = HIER ? &pend : nullptr;

  // ---- ring 1, phase A: cell ranges of the 9 rows.  Every lane looks up the centre row (row 0); lane L also
  //      looks up row L+1.  12 independent loads per lane, one memory latency for the whole neighbourhood.
  uint32_t cS[3], cE[3], mS[3], mE[3];          // centre row / my row: [start, end) of cells x = c0-1, c0, c0+1
  {
    const int my = L + 1;
    const int y = c1 + row_dy(my), z = c2 + row_dz(my);
    const bool rowok = y >= 0 && y < gp.dim[1] && z >= 0 && z < gp.dim[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = c0 - 1 + j;
      const bool xok = x >= 0 && x < gp.dim[0];
      cS[j] = cE[j] = mS[j] = mE[j] = 0;
      if (xok) {
        const uint32_t kc = cell_key(gp, x, c1, c2);
        cS[j] = cs[kc]; cE[j] = cs[kc + 1];
        if (rowok) {
          const uint32_t km = cell_key(gp, x, y, z);
          mS[j] = cs[km]; mE[j] = cs[km + 1];
        }
      }
    }
  }

  if (ha.heavy) {                                           // group-uniform
    uint32_t pop = (mE[0] - mS[0]) + (mE[1] - mS[1]) + (mE[2] - mS[2]);
    uint32_t big = max(max(mE[0] - mS[0], mE[1] - mS[1]), mE[2] - mS[2]);
    pop += (uint32_t)__shfl_xor((int)pop, 1, GL); pop += (uint32_t)__shfl_xor((int)pop, 2, GL); pop += (uint32_t)__shfl_xor((int)pop, 4, GL);
    big = max(big, (uint32_t)__shfl_xor((int)big, 1, GL)); big = max(big, (uint32_t)__shfl_xor((int)big, 2, GL)); big = max(big, (uint32_t)__shfl_xor((int)big, 4, GL));
    pop += (cE[0] - cS[0]) + (cE[1] - cS[1]) + (cE[2] - cS[2]);
    big = max(big, max(max(cE[0] - cS[0], cE[1] - cS[1]), cE[2] - cS[2]));
    if (pop >= ha.wave_min) {
      // marked by position in the sorted target array (2: a refined cell among the 27 -- those need the descending variant of the
      // wave kernel, which runs at half the occupancy); the marks are compacted IN ORDER afterwards, so that the wave kernel meets
      // the targets cell by cell and neighbours share what they read through L2
      if (L == 0) ha.heavy[list ? list[gid] : gid] = (HIER && big > ha.thr) ? 2u : 1u;
      return;                                               // whole groups leave together
    }
  }

  // plan of row r under the current limit: which cells survive, where their records are
  auto make_plan = [&](int r) -> RowPlan {
    RowPlan P;
    P.a0 = P.a1 = P.a2 = 0; P.n0 = P.n01 = P.T = 0;
    const int y = c1 + row_dy(r), z = c2 + row_dz(r);
    if (y < 0 || y >= gp.dim[1] || z < 0 || z >= gp.dim[2]) return P;
    const double gy = T.gap(1, y, y), gz = T.gap(2, z, z);
    const double s2 = gy * gy + gz * gz;
    if (s2 * T.h2 > top.lim_d) return P;
    uint32_t S[3], E[3];
    if (r == 0) {
#pragma unroll
      for (int j = 0; j < 3; ++j) { S[j] = cS[j]; E[j] = cE[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 3; ++j) { S[j] = __shfl(mS[j], r - 1, GL); E[j] = __shfl(mE[j], r - 1, GL); }
    }
    uint32_t n[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int x = c0 - 1 + j;
      const double g = T.gap(0, x, x);
      n[j] = ((g * g + s2) * T.h2 > top.lim_d) ? 0u : (E[j] - S[j]);   // cells outside the grid have S == E == 0
    }
    if constexpr (HIER) {
      // heavy cells leave the flat plan for the pending list; the middle cell first, so that in row 0 (planned first) the target's
      // own cell heads the list and its descent tightens the bound for all the others
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int j = jj == 0 ? 1 : (jj == 1 ? 0 : 2);
        if (n[j] && pend.heavy(S[j], E[j]) && pend.push(cell_key(gp, c0 - 1 + j, y, z), L)) n[j] = 0u;
      }
    }
    P.a0 = S[0]; P.a1 = S[1]; P.a2 = S[2];
    P.n0 = n[0]; P.n01 = n[0] + n[1]; P.T = P.n01 + n[2];
    return P;
  };
  auto request = [&](const RowPlan& P, Rec (&R)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t v = b * GL + L;
      if (v < P.T) R[b] = src[P.addr(v)];
    }
  };
  auto rank_batch = [&](const RowPlan& P, const Rec (&R)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if ((uint32_t)(b * GL) < P.T) {            // group-uniform
        const uint32_t v = b * GL + L;
        double d = INFINITY;
        uint32_t id = PT_NOIDX_U;
        if (v < P.T) { d = dist2(T.q, R[b]); id = R[b].id; }
        top.offer(d, id, gshift);
      }
    }
    for (uint32_t vb = NB * GL; vb < P.T; vb += GL) {   // rows longer than one batch (dense cells)
      const uint32_t v = vb + L;
      double d = INFINITY;
      uint32_t id = PT_NOIDX_U;
      if (v < P.T) { const Rec r = src[P.addr(v)]; d = dist2(T.q, r); id = r.id; }
      top.offer(d, id, gshift);
    }
  };

  // ---- ring 1, phase B: rows in centre-first order, the next row's records in flight while this one is ranked
  {
    Rec Rn[NB];
    RowPlan Pn = make_plan(0);
    request(Pn, Rn);
#pragma unroll 1
    for (int r = 0; r < 9; ++r) {
      Rec Rc[NB];
      const RowPlan Pc = Pn;
#pragma unroll
      for (int b = 0; b < NB; ++b) Rc[b] = Rn[b];
      if (r + 1 < 9) {
        Pn = make_plan(r + 1);        // planned under the limit as it is now: conservative, never wrong
        request(Pn, Rn);
      }
      rank_batch(Pc, Rc);
    }
  }

  // ---- rings >= 2: only while something outside the scanned box can still beat the limit ---------------------------
  const int ring_limit = max(PT_RING_LIMIT, (int)cbrtf(0.07f * (float)gp.nblocks));
  for (int r = 1;; ++r) {
    if constexpr (HIER) {
      // the heavy cells of the ring just scanned (ring 1 on the first pass): descended into here, the ONLY place -- before the
      // termination test, which therefore sees the bound they leave
      if (pend.n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // lane 0's list entries, for the whole group
        __builtin_amdgcn_wave_barrier();
        HierScan<Rec, KPL> H{gp, src, ha.nodes, T, top, gshift};
        for (uint32_t i = 0; i < pend.n; ++i) {
          const uint32_t key = pend.slot[i];
          int x, y, z;
          pt_decode_cell(gp, key, x, y, z);
          const double gx = T.gap(0, x, x), gy = T.gap(1, y, y), gz = T.gap(2, z, z);
          if ((gx * gx + gy * gy + gz * gz) * T.h2 > top.lim_d) continue;      // the bound has tightened since the cell was set aside
          const uint32_t nid = ha.cell_node[key];
          if (nid) H.template node<0>(nid); else H.range(cs[key], cs[key + 1]);   // (no node: the table was full when the cell asked)
        }
        pend.n = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");        // the next ring's entries stay behind these reads
        __builtin_amdgcn_wave_barrier();
      }
    }
    // every unscanned point lies beyond one of the box faces that still has cells behind it
    bool covered = true;
    double dout = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int lo = T.c[a] - r, hi = T.c[a] + r;
      if (lo > 0) { covered = false; dout = fmin(dout, T.u[a] - (double)lo); }
      if (hi < gp.dim[a] - 1) { covered = false; dout = fmin(dout, (double)(hi + 1) - T.u[a]); }
    }
    if (covered) break;
    dout = fmax(dout - PT_CELL_EPS, 0.0);
    if (dout * dout * T.h2 > top.lim_d) break;
    if (r >= ring_limit) {
      // Far from the points (a stray target, a gap in the cloud): walking ever larger, mostly empty shells costs O(r^2)
      // per ring.  Sweep the BLOCKS instead -- skip the empty ones, prune the others by their box, scan what is left --
      // starting the list again so that no point is offered twice.  O(blocks) per such target, exact like the walk; taken
      // once the walk has cost about as much as the sweep will (ring_limit^3 ~ blocks / 14).
      top.init(L, k, bound2 ? bound2[tr.id] : INFINITY);
      const uint32_t nb = (uint32_t)gp.nblocks;
      for (uint32_t b0 = 0; b0 < nb; b0 += GL) {
        const uint32_t b = b0 + (uint32_t)L;             // lane L looks at block b0 + L
        uint32_t bs_ = 0, be_ = 0;
        if (b < nb) { bs_ = cs[(size_t)b * PT_BLOCK_CELLS]; be_ = cs[((size_t)b + 1) * PT_BLOCK_CELLS]; }
        bool want = be_ > bs_;
        if (want) {
          const uint32_t macro = b >> 9, m9 = b & 511u;
          const int bx = (int)(macro % (uint32_t)gp.mdim[0]) * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u));
          const int by = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]) * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u));
          const int bz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1])) * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u));
          const double gx = T.gap(0, bx * 8, bx * 8 + 7), gy = T.gap(1, by * 8, by * 8 + 7), gz = T.gap(2, bz * 8, bz * 8 + 7);
          want = !((gx * gx + gy * gy + gz * gz) * T.h2 > top.lim_d);
        }
        uint32_t mask = (uint32_t)((__ballot(want) >> gshift) & 0xFFull);      // the group's eight verdicts
        while (mask) {                                    // group-uniform
          const int j = __ffs((int)mask) - 1;
          mask &= mask - 1;
          const uint32_t s0 = (uint32_t)__shfl(bs_, gshift + j), e0 = (uint32_t)__shfl(be_, gshift + j);
          for (uint32_t base = s0; base < e0; base += GL) {
            const uint32_t p = base + (uint32_t)top.L;
            double d = INFINITY;
            uint32_t id = PT_NOIDX_U;
            if (p < e0) { const Rec r = src[p]; d = dist2(T.q, r); id = r.id; }
            top.offer(d, id, gshift);
          }
        }
      }
      break;
    }
    const int rr = r + 1;                      // scan the shell box(rr) \ box(rr-1)
    const int x0 = max(c0 - rr, 0), x1 = min(c0 + rr, gp.dim[0] - 1);
    const int y0 = max(c1 - rr, 0), y1 = min(c1 + rr, gp.dim[1] - 1);
    const int z0 = max(c2 - rr, 0), z1 = min(c2 + rr, gp.dim[2] - 1);
    for (int z = z0; z <= z1; ++z)
      for (int y = y0; y <= y1; ++y) {
        const bool shell = (z == c2 - rr) || (z == c2 + rr) || (y == c1 - rr) || (y == c1 + rr);
        if (shell) scan_row_generic<Rec, KPL>(gp, src, cs, T, top, gshift, x0, x1, y, z, pp);
        else {
          if (c0 - rr >= 0) scan_row_generic<Rec, KPL>(gp, src, cs, T, top, gshift, c0 - rr, c0 - rr, y, z, pp);
          if (c0 + rr <= gp.dim[0] - 1) scan_row_generic<Rec, KPL>(gp, src, cs, T, top, gshift, c0 + rr, c0 + rr, y, z, pp);
        }
      }
  }

  const size_t row = (size_t)tr.id * (size_t)k;
#pragma unroll
  for (int j = 0; j < KPL; ++j) {
    const int e = L * KPL + j;
    if (e < k) {
      out_idx[row + e] = top.li[j];
      if (out_d2) out_d2[row + e] = top.ld[j];
    }
  }
#ifdef PT_VISITS
  __builtin_amdgcn_wave_barrier();
  if (out_d2 && L == GL - 1) {                                              // (results are garbage in these columns)
    out_d2[row + k - 1] = (double)top.nv * GL;
    if (k >= 4) { out_d2[row + k - 2] = (double)(wall_clock64() - pt_t0); out_d2[row + k - 3] = (double)pt_t0; out_d2[row + k - 4] = (double)(blockIdx.x * 4u + threadIdx.x / 64u); }
  }
#endif
}

// =====================================================================================================================
// Wave kernel: ONE WAVE (64 lanes) PER TARGET -- the targets of dense neighbourhoods (clouds with strong density contrast).
//
// Why a third kernel: the group kernel keeps eight targets per wave in lockstep, and in a dense cell every step of eight records
// ends in the insertion path for SOME group (k ln(n / k) insertions per target, ~100 VALU instructions each at k = 32, seven
// groups idle meanwhile): measured on the clustered generator it looks at 1e11 records/s whatever the index offers.  Here the
// whole wave serves one target: 64 records per step with wave-uniform control flow; the k best live one entry per lane -- an
// unsorted pool whose k-th smallest key, found by pivoting, is the scalar limit (see WaveScan: THE LIST) -- and are sorted once, at
// the end.  Cells, shells, blocks and the rows of refined nodes are looked up 64 at a time, one per lane.
// Same order, same bounds, same results as the group kernel (exact); k <= 32 (PT_MAX_K).
constexpr int WV_RING_MAX = 31;          // shells are walked up to this ring at most (then the blocks are swept)
constexpr uint32_t WV_RUN = 16;          // consecutive workgroups (64 targets) that share an XCD

// key_lt without short-circuit evaluation: no branches around the comparisons (the compiler turns `a < b || (a == b && i < j)` on
// per-lane values into three exec-masked blocks)
__device__ inline bool key_lt_flat(double ad, uint32_t ai, double bd, uint32_t bi) {
  const bool lt = ad < bd, eq = ad == bd, il = ai < bi;
  return lt | (eq & il);
}
__device__ inline double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline uint32_t readlane_u32(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
// lane l of v := the wave-uniform value x (no builtin for it in this compiler).  M0 is free in the kernels that use this (no LDS-DMA,
// no GWS): the compiler's warning about the clobber is silenced.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ inline uint32_t writelane_u32(uint32_t v, uint32_t x, int l) {
  asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(x), "s"(l) : "m0");     // (one SGPR per VOP3: the lane goes through M0)
  return v;
}
#pragma clang diagnostic pop

// fp32 pre-filter distance of the tile and wave kernels (fp32 records, fp32 targets): 3 sub, 1 mul, 2 fma on exact inputs, all
// terms >= 0 -- relative error < 2^-21.  Never a result: what passes is evaluated again in fp64, unfused.
__device__ inline float dist2_f32(float qx, float qy, float qz, const RecF& r) {
  const float dx = qx - r.x, dy = qy - r.y, dz = qz - r.z;
  return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
}
template <class Rec> struct IsRecF { static constexpr bool value = false; };
template <> struct IsRecF<RecF> { static constexpr bool value = true; };

// __ballot(bool) goes through an integer compare: the compiler materialises the predicate (v_cndmask 0 / 1) and compares it with zero
// again -- two VALU instructions per ballot in a kernel bound by VALU issue.  The builtin takes the condition's mask as it is.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

template <class Rec>
struct WaveScan {
  const Rec* __restrict__ src;
  const uint32_t* __restrict__ nodes;
  double q[3], u[3], h2;       // the target, its position in cell units, squared cell side: wave-uniform
  double ld;                   // my entry of the pool of the k best (+inf, NOIDX: none); after finish(): lane i holds rank i
  uint32_t li;
  double lim_d, bnd_d;         // acceptance limit = min(entry of rank k-1, caller's bound): wave-uniform
  uint32_t lim_i;
  // fp32 clouds: a step whose 64 records are all beyond the limit ALREADY IN FP32 (most steps of a long scan) skips the fp64 metric
  // and everything after it.  d32 <= d (1 + 2^-21) (dist2_f32), so a record with d <= lim_d has d32 <= lim32 := lim_d (1 + 2^-20)
  // rounded to float (nearest: 2^-24 at most the wrong way), plus a slack for fp32 underflow; +inf stays +inf.
#ifdef PT_NOPRE32
  static constexpr bool PRE32 = false;                   // (A/B builds: tools/sweep_pend.sh)
#else
  static constexpr bool PRE32 = IsRecF<Rec>::value;
#endif
  float qf[3], lim32;
  __device__ __forceinline__ void set_lim32() { lim32 = (float)(lim_d * 1.00000095367431640625) + 1e-30f; }
  int k, lane;
  // candidates set aside: this wave's 64 slots in LDS and how many are taken (wave-uniform); see offer()
  uint4* pend;                 // slot: (d2 low word, d2 high word, index, -)
  uint32_t npend;
#ifdef PT_VISITS
  uint32_t nv = 0, nn = 0, nmerge = 0;     // instrumented build: steps of 64 records, nodes entered, selections (sort-merges until round 4)
#endif

  // THE LIST (round 4, second form).  Rounds 2 - 4 kept the k best SORTED across the lanes and paid a 27-stage bitonic sort-merge (361 VALU
  // instructions) for every 16 - 48 candidates, 4.5 of them per sheet target of config 5 -- 46 % of a kernel that is bound by VALU issue (32
  // more fp32 instructions per step of 64 records cost their full 3 cycles each: tools/ab_c5.sh, -DPT_WABLATE).  Nothing needs the order
  // before the end: a scan needs the k-th smallest key, as its limit, and somewhere to keep the k best.  So the lanes hold an UNSORTED POOL
  // (an empty lane: +inf, NOIDX), candidates collect in the wave's 64 LDS slots as before, and a flush
  //   places them into free lanes (lane j, the r-th free one, reads slot r: one prefix count, two LDS reads),
  //   selects the k-th smallest key of the pool by pivoting (a lane's key against all: one ballot and a count per probe, ~ 8 probes of ~ 6
  //   VALU + scalar work on random data; every decision is scalar), makes it the limit and empties the lanes beyond it:
  // ~ 70 VALU instructions instead of 361.  One 21-stage sort of the pool at the very end puts rank i into lane i for the output.
  uint32_t npool;              // entries in the pool (wave-uniform)
  __device__ void reset() { ld = INFINITY; li = PT_NOIDX_U; lim_d = bnd_d; lim_i = PT_NOIDX_U; npend = 0; npool = 0; set_lim32(); }
  // one step's 64 records: fp32 clouds look at the fp32 distance first
  __device__ __forceinline__ void step(const Rec& r, bool have) {
    if constexpr (PRE32) {
      const bool near32 = have & (dist2_f32(qf[0], qf[1], qf[2], r) <= lim32);      // (no short circuit: a branch around six instructions costs more than they do)
      if (!ballot64(near32)) return;                        // wave-uniform
    }
    offer(dist2(q, r), r.id, have);                         // (lanes without a record computed on whatever record their registers held: `have` keeps them out)
  }
  // Many candidates at once (the first steps of a target: with fewer than k points seen every record is one): sort the 64 candidate
  // slots across the lanes (bitonic, 21 exchange stages), take the 64 smallest of list and candidates (list[i] against candidate
  // [63 - i]) and sort that bitonic sequence (6 stages) -- ~500 instructions whatever the number of candidates, against ~35 for
  // each one-by-one insertion.  Keys are distinct (ids) except the empty slots (+inf, NOIDX), whose order does not matter.
  // The exchanges never touch the LDS: partner lane ^ 1, ^ 2 by DPP quad permutes, ^ 4 by two bank-masked row shifts, ^ 8 by a row
  // rotation, ^ 16 and ^ 32 by gfx950's v_permlane16_swap / v_permlane32_swap (both copies of the value go in; each lane picks the
  // one that holds its partner's).  With ds_bpermute every one of the 27 stages was an LDS round trip.
  template <int J>
  __device__ __forceinline__ uint32_t xor_lane(uint32_t x) const {
    if constexpr (J == 1) return dpp_u32<0xB1>(x);                                       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_u32<0x4E>(x);                                  // quad_perm [2,3,0,1]
    else if constexpr (J == 4) {
      const int t = __builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xF, 0x5, false);  // row_shl:4 into lanes 0-3, 8-11 of a row
      return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x114, 0xF, 0xA, false);    // row_shr:4 into lanes 4-7, 12-15
    } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xF, 0xF, false);   // row_ror:8
    else if constexpr (J == 16) { const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false); return (lane & 16) ? r[0] : r[1]; }
    else { const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false); return (lane & 32) ? r[0] : r[1]; }
  }
  template <int J>
  __device__ __forceinline__ void exchange(double& xd, uint32_t& xi, bool keep_min) const {
    const uint32_t plo = xor_lane<J>((uint32_t)__double2loint(xd)), phi = xor_lane<J>((uint32_t)__double2hiint(xd)), pi = xor_lane<J>(xi);
    const double pd = __hiloint2double((int)phi, (int)plo);
    // keep_min: take the partner's if it is smaller; else take it unless it is smaller (equal keys -- two empty slots -- swap to no effect)
    if (key_lt_flat(pd, pi, xd, xi) == keep_min) { xd = pd; xi = pi; }
  }
  // the exchange stages of one bitonic block size k2 (partners ^ k2/2 ... ^ 1); k2 is a constant wherever this is used
  __device__ __forceinline__ void stages(double& xd, uint32_t& xi, int k2, int l) const {
    const bool up = (l & k2) == 0;
    if (k2 > 32) exchange<32>(xd, xi, ((l & 32) == 0) == up);          // wave-uniform tests
    if (k2 > 16) exchange<16>(xd, xi, ((l & 16) == 0) == up);
    if (k2 > 8) exchange<8>(xd, xi, ((l & 8) == 0) == up);
    if (k2 > 4) exchange<4>(xd, xi, ((l & 4) == 0) == up);
    if (k2 > 2) exchange<2>(xd, xi, ((l & 2) == 0) == up);
    exchange<1>(xd, xi, ((l & 1) == 0) == up);
  }
#ifndef PT_PEND_FLUSH
#define PT_PEND_FLUSH 24
#endif
  static constexpr int PEND_FLUSH = PT_PEND_FLUSH;
  // The flush is ONE function in the code object (as the sort-merge was), values in and values out -- nothing of the scan's state goes through
  // memory: inlined at every place a scan may flush (65 of them in the descending variant) it pushed other members out of line, and a member
  // called as a function takes `this`, i.e. the whole scan state moves to scratch memory (26 -> 84 ms for that launch at config 5's shape).
  //   place:  slots [done, done + take) -> the first `take` free lanes (lane j, the r-th free one, reads slot r)
  //   select: the k-th smallest key of the pool by pivoting -- the lowest / the highest lane in question by turns (records arrive in memory
  //           order, not by distance; a pool that happens to be sorted one way round still halves every other probe); every decision is
  //           scalar, the set in question shrinks with every probe; the key found is the limit, the lanes beyond it are emptied.
  // Arguments arrive in VGPRs: the wave-uniform ones are said to be uniform, or the loops are compiled for divergent lanes.
  struct Pool { double ld; uint32_t li; uint32_t npool; double lim_d; uint32_t lim_i; uint32_t nsel; };
  __device__ __attribute__((noinline)) static Pool flush_core(double ld_, uint32_t li_, uint32_t npool_, uint32_t n_, uint32_t k_, double lim_d_, uint32_t lim_i_,
                                                              const uint4* pend_) {
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_), k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k_);
    uint32_t npool = (uint32_t)__builtin_amdgcn_readfirstlane((int)npool_), nsel = 0;
    double lim_d = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(lim_d_)), __builtin_amdgcn_readfirstlane(__double2loint(lim_d_)));
    uint32_t lim_i = (uint32_t)__builtin_amdgcn_readfirstlane((int)lim_i_);
    const uint4* pend = reinterpret_cast<const uint4*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)pend_ >> 32)) << 32) |
                                                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint64_t)pend_));
    uint32_t done = 0;
    while (done < n) {                                      // wave-uniform; one trip unless more arrived than there are free lanes
      const uint32_t take = min(n - done, 64u - npool);
      {
        const bool fre = li_ == PT_NOIDX_U && ld_ == INFINITY;
        const unsigned long long F = ballot64(fre);
        const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(F >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)F, 0u));
        if (fre && r < take) { const uint4 v = pend[done + r]; ld_ = __hiloint2double((int)v.y, (int)v.x); li_ = v.z; }
        npool += take;
      }
      done += take;
      if (npool >= k) {
        ++nsel;
        unsigned long long A = ballot64(!(li_ == PT_NOIDX_U && ld_ == INFINITY));      // the lanes still in question
        uint32_t need = k;                                                             // rank sought among them
        double td = INFINITY;
        uint32_t ti = PT_NOIDX_U;
        bool found = false;
        int flip = 0;
        while (!found) {                                    // wave-uniform: A and need are scalars; A shrinks with every probe
          const int p = flip ? 63 - __builtin_clzll(A) : __ffsll((long long)A) - 1;
          flip ^= 1;
          const double pd = readlane_f64(ld_, p);
          const uint32_t pi = readlane_u32(li_, p);
          const unsigned long long L = ballot64(key_lt_flat(ld_, li_, pd, pi)) & A;
          const uint32_t cl = (uint32_t)__popcll(L);
          if (need <= cl) A = L;
          else if (need == cl + 1u) { td = pd; ti = pi; found = true; }
          else { need -= cl + 1u; A &= ~L; A &= ~(1ull << p); }
        }
        if (key_lt_flat(td, ti, ld_, li_)) { ld_ = INFINITY; li_ = PT_NOIDX_U; }       // beyond the k-th: out
        npool = k;
        lim_d = td; lim_i = ti;                             // (<= the caller's bound: nothing beyond it was ever offered)
      }
    }
    return Pool{ld_, li_, npool, lim_d, lim_i, nsel};
  }
  __device__ __forceinline__ void flush() {
    if (!npend) return;                                     // wave-uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the slots were written by this wave's own lanes: LDS keeps a wave's order
    __builtin_amdgcn_wave_barrier();
    const Pool r = flush_core(ld, li, npool, npend, (uint32_t)k, lim_d, lim_i, pend);
    npend = 0;
    ld = r.ld; li = r.li;
    npool = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.npool);
    lim_d = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(r.lim_d)), __builtin_amdgcn_readfirstlane(__double2loint(r.lim_d)));
    lim_i = (uint32_t)__builtin_amdgcn_readfirstlane((int)r.lim_i);
    set_lim32();
#ifdef PT_VISITS
    nmerge += (uint32_t)__builtin_amdgcn_readfirstlane((int)r.nsel);
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // ... and the next round's writes stay behind these reads
    __builtin_amdgcn_wave_barrier();
  }
  // one candidate per lane (d = +inf for lanes without one)
  __device__ __forceinline__ void offer(double d, uint32_t id, bool have) {
    const bool pass = have & key_lt_flat(d, id, lim_d, lim_i) & !(d > bnd_d);
    const unsigned long long mask = ballot64(pass);
    if (!mask) return;                                      // wave-uniform (as every branch below)
    const uint32_t c = (uint32_t)__popcll(mask);
    if (npend + c > 64u) flush();                           // (no room in the slots: the pool takes what is there first)
    const uint32_t slot = npend + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (pass) pend[slot] = make_uint4((uint32_t)__double2loint(d), (uint32_t)__double2hiint(d), id, 0u);      // one 16-byte LDS write
    npend += c;
    if (npend >= (uint32_t)PEND_FLUSH) flush();
  }
  // the end of a search: rank i into lane i (the empty lanes sort last)
  // (the pool's <= k entries are gathered in the low lanes first -- through the slots, free by now -- so that the sort spans 8, 16 or 32 lanes:
  //  6, 10 or 15 exchange stages instead of the 21 that 64 lanes take)
  __device__ __forceinline__ void finish() {                // (forced: called as a function it takes `this`, and the whole scan state moves to scratch memory)
    flush();
    const bool has = !(li == PT_NOIDX_U && ld == INFINITY);
    const unsigned long long M = ballot64(has);
    const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
    if (has) pend[r] = make_uint4((uint32_t)__double2loint(ld), (uint32_t)__double2hiint(ld), li, 0u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ld = INFINITY; li = PT_NOIDX_U;
    if ((uint32_t)lane < (uint32_t)__popcll(M)) { const uint4 v = pend[lane]; ld = __hiloint2double((int)v.y, (int)v.x); li = v.z; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    stages(ld, li, 2, lane); stages(ld, li, 4, lane); stages(ld, li, 8, lane);
    if (k > 8) stages(ld, li, 16, lane);                    // wave-uniform
    if (k > 16) stages(ld, li, 32, lane);
    if (k > 32) stages(ld, li, 64, lane);
  }
  // WPF steps of loads are in flight while a step is ranked: with one, every step of 64 records cost a full memory latency (60 us
  // per target at 25 - 35 steps, measured: the steps' arithmetic is ~0.15 us)
#ifndef PT_WPF
#define PT_WPF 1
#endif
  static constexpr int WPF = PT_WPF;
  // Two register sets take turns (a is ranked while b's load is in flight and the other way round): with one set and a copy per step the
  // compiler waits for a load right after issuing it, to move its words into the set the ranking reads.  For the same reason a 16-byte record
  // travels as ONE four-word value (four consecutive registers, the load's own destination) until the step takes it apart: as a struct of four
  // scalars its index word was given a register elsewhere, and the move into it waited for the load.
  using Vec = typename std::conditional<IsRecF<Rec>::value, float4, Rec>::type;
  __device__ __forceinline__ static Vec loadv(const Rec* p) {
    if constexpr (IsRecF<Rec>::value) return *reinterpret_cast<const float4*>(p); else return *p;
  }
  __device__ __forceinline__ void stepv(const Vec& v, bool have) {
    if constexpr (IsRecF<Rec>::value) { RecF r; r.x = v.x; r.y = v.y; r.z = v.z; r.id = __float_as_uint(v.w); step(r, have); }
    else step(v, have);
  }
  __device__ __forceinline__ void range(uint32_t s, uint32_t e) {
    // (every load is UNCONDITIONAL, its index clamped to the last record: a load under `if (p < e)` merges with the old value behind it, and the
    //  copy that merge needs waits for the load on the spot -- the prefetch gone; lanes beyond the end rank a record twice and `have` discards it)
    if (s >= e) return;                                     // wave-uniform
    const uint32_t last = e - 1u;
    Vec a = loadv(src + min(s + (uint32_t)lane, last)), b;
    for (uint32_t base = s; base < e; base += 128u) {       // wave-uniform trip count, no early exit
      const uint32_t p = base + (uint32_t)lane;
      b = loadv(src + min(p + 64u, last));
#ifdef PT_VISITS
      ++nv;
#endif
      stepv(a, p < e);
      if (base + 64u < e) {                                 // wave-uniform
        a = loadv(src + min(p + 128u, last));
#ifdef PT_VISITS
        ++nv;
#endif
        stepv(b, p + 64u < e);
      }
    }
  }
  // Up to 64 runs of records as ONE stream: lane j brings its run's first record S and length C (0: none); virtual record v of the
  // stream is record v - P[j] of the run j whose prefix interval holds v.  64 records per step whatever the runs' lengths, and the
  // next step's loads are in flight while this one is ranked -- a run costs no memory latency of its own (cell by cell, the 27
  // cells of ring 1 cost 27: 40 us per target, measured).  No pruning inside the stream: the caller decides the runs beforehand.
  __device__ __forceinline__ void stream(uint32_t S, uint32_t C) {
    const uint32_t pin = wave_incl_scan(C), pex = pin - C;
    const uint32_t T = readlane_u32(pin, 63);
    if (!T) return;                                         // wave-uniform
    // my cursor: the run my current virtual record is in -- its interval [c_lo, c_hi) of the stream and its first record.  When a
    // lane leaves its run, ALL lanes search the prefix sums again: a binary search of six shuffles, unrolled, with no loop around it
    // (a `while any lane must advance` loop was turned by the compiler into one that lanes leave one by one, and a shuffle reads
    // nothing from a lane that has left).
    uint32_t c_lo = 0, c_hi = 0, c_S = 0;
    auto locate = [&](uint32_t v) -> uint32_t {             // address of virtual record v (any value for v >= T)
      const bool out = v < T && v >= c_hi;
      if (ballot64(out) != 0ull) {                          // wave-uniform
        int sg = 0;                                         // number of runs that end at or before v
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
          const uint32_t pe = (uint32_t)__shfl((int)pin, sg + step - 1);
          sg += v >= pe ? step : 0;
        }
        sg = min(sg, 63);
        const uint32_t nS = (uint32_t)__shfl((int)S, sg), nlo = (uint32_t)__shfl((int)pex, sg), nhi = (uint32_t)__shfl((int)pin, sg);
        if (out) { c_S = nS; c_lo = nlo; c_hi = nhi; }
      }
      return c_S + (v - c_lo);
    };
    const uint32_t last = T - 1u;                           // two register sets taking turns and unconditional loads, as in range()
    Vec a = loadv(src + locate(min((uint32_t)lane, last))), b;
    for (uint32_t base = 0; base < T; base += 128u) {       // wave-uniform trip count, no early exit
      const uint32_t v = base + (uint32_t)lane;
      b = loadv(src + locate(min(v + 64u, last)));
#ifdef PT_VISITS
      ++nv;
#endif
      stepv(a, v < T);
      if (base + 64u < T) {                                 // wave-uniform
        a = loadv(src + locate(min(v + 128u, last)));
#ifdef PT_VISITS
        ++nv;
#endif
        stepv(b, v + 64u < T);
      }
    }
  }
  __device__ double gap2(int a, double lo, double hi) const {
    const double g = fmax(fmax(lo - u[a], u[a] - hi) - PT_CELL_EPS, 0.0);
    return g * g;
  }
  // refined cell (pt_refine.hip): the 64 rows of sub-cells are tested one per lane, the target's own sub-cell goes first.  Little is
  // kept across a descent into a child (three levels of this are inlined into one another): the node's address, the rows still to
  // visit, the children of the current row -- the header is read again (scalar loads) whenever a row needs its geometry.
  template <int DEPTH>
  __device__ void node(uint32_t nid) {
    const uint32_t* N = nodes + (size_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)nid) - 1u) * PT_NODE_WORDS;     // (wave-uniform: scalar loads)
    uint32_t own = 0xFFFFFFFFu;
    unsigned long long live;
#ifdef PT_VISITS
    ++nn;
#endif
    flush();                                                // the rows are chosen by the limit
    {
      const double* hd = reinterpret_cast<const double*>(N);
      const double ox = hd[0], oy = hd[1], oz = hd[2], inv = hd[3], w = hd[4];
      const double rx = (u[0] - ox) * inv, ry = (u[1] - oy) * inv, rz = (u[2] - oz) * inv;
      if (rx >= 0.0 && rx < 8.0 && ry >= 0.0 && ry < 8.0 && rz >= 0.0 && rz < 8.0) own = (uint32_t)(((int)rz << 6) | ((int)ry << 3) | (int)rx);
      const double fy = (double)(lane & 7), fz = (double)(lane >> 3);          // my row (sy, sz) = (lane & 7, lane >> 3)
      const double s2 = gap2(1, oy + fy * w, oy + (fy + 1.0) * w) + gap2(2, oz + fz * w, oz + (fz + 1.0) * w);
      live = ballot64(!(s2 * h2 > lim_d)) & ((unsigned long long)N[PT_NODE_ROWMASK] | ((unsigned long long)N[PT_NODE_ROWMASK + 1] << 32));
    }
    bool first = own != 0xFFFFFFFFu;
    while (first || live) {                                 // wave-uniform
      int r2, xa, xb;
      if (first) { r2 = (int)(own >> 3); xa = xb = (int)(own & 7u); }
      else {
        r2 = __ffsll((long long)live) - 1;
        live &= live - 1;
        uint32_t again = 0;
        asm volatile("" : "+s"(again));                     // (read the header again rather than keep it across the descents)
        const double* hd = reinterpret_cast<const double*>(N + again);
        const double ox = hd[0], oy = hd[1], oz = hd[2], w = hd[4];
        const double fy = (double)(r2 & 7), fz = (double)(r2 >> 3);
        const double t2 = gap2(1, oy + fy * w, oy + (fy + 1.0) * w) + gap2(2, oz + fz * w, oz + (fz + 1.0) * w);
        if (t2 * h2 > lim_d) continue;                      // the limit has moved since the ballot
        xa = 0; xb = 7;
        while (xa <= xb && (gap2(0, ox + (double)xa * w, ox + (double)(xa + 1) * w) + t2) * h2 > lim_d) ++xa;
        while (xb >= xa && (gap2(0, ox + (double)xb * w, ox + (double)(xb + 1) * w) + t2) * h2 > lim_d) --xb;
        if (xa > xb) continue;
      }
      const bool sweep = !first;
      first = false;
      uint32_t stl = 0, chl = 0;                            // lane x: start of sub-cell x of the row (x = 8: its end) and its child
      if (lane < 9) stl = N[PT_NODE_START + r2 * 8 + lane];
      if (lane < 8) chl = N[PT_NODE_CHILD + r2 * 8 + lane];  // (the last level has no children, but its leaves may carry the identical-points tag)
      // leaf sub-cells next to each other are one contiguous run of records, scanned in one go; a sub-cell that is a node, the own
      // sub-cell (already done) and the end of the row cut the run.  The row's leaves first, then its children one by one.
      uint32_t kids = 0, run_s = 0, run_e = 0;
      for (int x = xa; x <= xb + 1; ++x) {
        bool cut = x > xb || (sweep && (uint32_t)(r2 * 8 + x) == own);
        uint32_t front = 0;                                   // > 0: a leaf of identical points, this many of them (the lowest indices) are all a search needs
        if (!cut) {
          const uint32_t ch = readlane_u32(chl, x);
          if (ch & PT_LEAF_TRUNC) { front = ch & ~PT_LEAF_TRUNC; cut = true; }
          else if (ch != 0u) { kids |= 1u << x; cut = true; }
        }
        if (!cut) {
          if (run_e == run_s) run_s = readlane_u32(stl, x);
          run_e = readlane_u32(stl, x + 1);
          continue;
        }
        if (run_e > run_s) range(run_s, run_e);
        run_s = run_e = 0;
        if (front) { const uint32_t fs = readlane_u32(stl, x); range(fs, fs + front); }
      }
      if constexpr (DEPTH + 1 < PT_REFINE_DEPTH) {
        while (kids) {
          const int x = __ffs((int)kids) - 1;
          kids &= kids - 1;
          node<DEPTH + 1>(readlane_u32(chl, x));
        }
      }
    }
  }
};

#ifndef PT_WV_MINW_H
#define PT_WV_MINW_H 6
#endif
#ifndef PT_WV_MINW
#define PT_WV_MINW 8
#endif
template <class Rec, bool HIER>
__global__ __launch_bounds__(WG, HIER ? PT_WV_MINW_H : PT_WV_MINW) void knn_wave_kernel(GridParams gp, const Rec* __restrict__ src, const uint32_t* __restrict__ cs, const Rec* __restrict__ tgt,
                                                      uint32_t m, int k, const double* __restrict__ bound2, uint32_t* __restrict__ out_idx,
                                                      double* __restrict__ out_d2, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n,
                                                      HierArgs ha, WaveBlend wb) {
  // Consecutive workgroups go to different XCDs (8 of them, each with its own L2): hand the list out in runs of WV_RUN workgroups
  // per XCD, so that the targets of neighbouring cells -- which read the same 27 cells -- meet in one L2, while all XCDs still
  // advance through the list together (one contiguous eighth per XCD: the dense parts of the cloud end up on a few XCDs, 1.6 x slower).
  const uint32_t count = list ? *list_n : m;
  const uint32_t j = blockIdx.x >> 3, wgl = ((j / WV_RUN) * 8u + (blockIdx.x & 7u)) * WV_RUN + j % WV_RUN;
  const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wgl * 4u + (threadIdx.x >> 6)));       // wave-uniform by construction: said so, the target and everything derived from it live in SGPRs
  if (wid >= count) return;                                 // whole waves leave together
  const int lane = threadIdx.x & 63;
  const Rec tr = tgt[list ? list[wid] : wid];
#ifdef PT_VISITS
  const unsigned long long pt_t0 = wall_clock64();
  unsigned long long pt_ph[4] = {pt_t0, pt_t0, pt_t0, pt_t0};      // cells known / own cell done / ring-1 stream done / search done
#endif
  __shared__ uint4 pend[WG / 64][64];
  WaveScan<Rec> W;
  W.src = src; W.nodes = ha.nodes; W.k = k; W.lane = lane;
  W.pend = pend[threadIdx.x >> 6];
  W.q[0] = (double)tr.x; W.q[1] = (double)tr.y; W.q[2] = (double)tr.z;
  W.qf[0] = (float)tr.x; W.qf[1] = (float)tr.y; W.qf[2] = (float)tr.z;      // (used by fp32 clouds only, whose targets are fp32 too)
  W.h2 = gp.h * gp.h;
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    W.u[a] = (W.q[a] - gp.bbmin[a]) * gp.inv_h;
    c[a] = (int)fmin(fmax(W.u[a], 0.0), (double)(gp.dim[a] - 1));
  }
  W.bnd_d = bound2 ? bound2[tr.id] : INFINITY;
  if (W.bnd_d < 0.0) {         // (as in knn_kernel: nothing wanted from this cloud; wave-uniform)
    if (lane < k) { const size_t row0 = (size_t)tr.id * (size_t)k; out_idx[row0 + lane] = PT_NOIDX_U; if (out_d2) out_d2[row0 + lane] = INFINITY; }
    return;
  }
  W.reset();
  auto cgap = [&](int a, int lo, int hi) -> double {        // as TargetGeom::gap
    return fmax(fmax((double)lo - W.u[a], W.u[a] - (double)(hi + 1)) - PT_CELL_EPS, 0.0);
  };
  const int ring_limit = min(WV_RING_MAX, max(PT_RING_LIMIT, (int)cbrtf(0.07f * (float)gp.nblocks)));
  // One loop serves ring 1 (27 cells, the target's own first, then its row, then the rest centre-first) and every further shell
  // (64 of its cells per step), so that the scan and the descent exist once in the code.
  int rr = 1, st = -1, nst = 0;
  for (;;) {
    int x = 0, y = 0, z = 0;
    bool valid;
    if (st < 0) {                                           // ring 1: lane i < 27 -> row i / 3 (centre-first), cell 0, -1, +1 of it
      const int r = lane / 3, j = lane - 3 * r;
      valid = lane < 27;
      x = c[0] + (j == 0 ? 0 : (j == 1 ? -1 : 1)); y = c[1] + row_dy(valid ? r : 0); z = c[2] + row_dz(valid ? r : 0);
    } else {
      // cell i of the shell of ring rr (side^3 - (side - 2)^3 of them), 64 per step: the two full planes dz = -rr, +rr row by row, then
      // for every plane in between its perimeter -- row dy = -rr, row dy = +rr, column dx = -rr, column dx = +rr
      const uint32_t side = 2u * (uint32_t)rr + 1u, in = side - 2u, plane = side * side, per = 4u * side - 4u;
      const uint32_t i = (uint32_t)st * 64u + (uint32_t)lane;
      valid = i < 2u * plane + in * per;
      int dx, dy, dz;
      if (i < 2u * plane) {
        const uint32_t j = i < plane ? i : i - plane, row = j / side;
        dz = i < plane ? -rr : rr; dy = (int)row - rr; dx = (int)(j - row * side) - rr;
      } else {
        const uint32_t j = i - 2u * plane, pz = j / per, q = j - pz * per;
        dz = -rr + 1 + (int)pz;
        if (q < 2u * side) { dy = q < side ? -rr : rr; dx = (int)(q < side ? q : q - side) - rr; }
        else { const uint32_t t = q - 2u * side; dx = t < in ? -rr : rr; dy = -rr + 1 + (int)(t < in ? t : t - in); }
      }
      x = c[0] + dx; y = c[1] + dy; z = c[2] + dz;
    }
    valid = valid && x >= 0 && x < gp.dim[0] && y >= 0 && y < gp.dim[1] && z >= 0 && z < gp.dim[2];
    uint32_t key = 0, S = 0, E = 0;
    double g2 = 0.0;
    if (valid) {
      key = cell_key(gp, x, y, z);
      S = cs[key]; E = cs[key + 1];
      const double gx = cgap(0, x, x), gy = cgap(1, y, y), gz = cgap(2, z, z);
      g2 = gx * gx + gy * gy + gz * gz;
    }
#ifdef PT_VISITS
    if (st < 0) { asm volatile("" ::"v"(S), "v"(E)); pt_ph[0] = wall_clock64(); }      // the 27 cells' table entries are here
#endif
    // refined cells are descended into (one by one: the target's own first); everything else of this step is ONE stream, after the
    // own cell on the first step so that the bound it leaves decides which of the other 26 are read at all
    uint32_t nid = 0;
    if constexpr (HIER) { if (E - S > ha.thr) nid = ha.cell_node[key]; }       // (S == E == 0 for lanes without a cell)
    if (st < 0) {
      const uint32_t s0 = readlane_u32(S, 0), e0 = readlane_u32(E, 0), n0 = readlane_u32(nid, 0);
      if (n0) W.template node<0>(n0);
      else W.range(s0, e0);
      if (lane == 0) { S = E = 0; nid = 0; }
      W.flush();                                            // the limit the own cell leaves decides which of the other 26 are read
#ifdef PT_VISITS
      asm volatile("" ::"v"(W.ld)); pt_ph[1] = wall_clock64();
#endif
    }
    const bool on = E > S && !(g2 * W.h2 > W.lim_d);
    W.stream(S, on && !nid ? E - S : 0u);
#ifdef PT_VISITS
    if (st < 0) { asm volatile("" ::"v"(W.ld)); pt_ph[2] = wall_clock64(); }
#endif
    if constexpr (HIER) {
      unsigned long long want = ballot64(on && nid);
      while (want) {                                        // wave-uniform
        const int i = __ffsll((long long)want) - 1;
        want &= want - 1;
        if (readlane_f64(g2, i) * W.h2 > W.lim_d) continue; // the limit has moved since the ballot
        W.template node<0>(readlane_u32(nid, i));
      }
    }
    if (st >= 0 && ++st < nst) continue;
    // ring rr is complete: every unscanned point lies beyond one of the box faces that still has cells behind it
    W.flush();
    bool covered = true;
    double dout = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int lo = c[a] - rr, hi = c[a] + rr;
      if (lo > 0) { covered = false; dout = fmin(dout, W.u[a] - (double)lo); }
      if (hi < gp.dim[a] - 1) { covered = false; dout = fmin(dout, (double)(hi + 1) - W.u[a]); }
    }
    if (covered) break;
    dout = fmax(dout - PT_CELL_EPS, 0.0);
    if (dout * dout * W.h2 > W.lim_d) break;
    if (rr >= ring_limit) {
      // far from the points: sweep the BLOCKS (skip the empty ones, prune by box, scan the rest), the list started again so that
      // no point is offered twice -- as the group kernel does, 64 blocks per step
      W.reset();
      const uint32_t nb = (uint32_t)gp.nblocks;
      for (uint32_t b0 = 0; b0 < nb; b0 += 64u) {
        const uint32_t b = b0 + (uint32_t)lane;
        uint32_t bs_ = 0, be_ = 0;
        double bg2 = 0.0;
        if (b < nb) { bs_ = cs[(size_t)b * PT_BLOCK_CELLS]; be_ = cs[((size_t)b + 1) * PT_BLOCK_CELLS]; }
        if (be_ > bs_) {
          const uint32_t macro = b >> 9, m9 = b & 511u;
          const int bx = (int)(macro % (uint32_t)gp.mdim[0]) * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u));
          const int by = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]) * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u));
          const int bz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1])) * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u));
          const double gx = cgap(0, bx * 8, bx * 8 + 7), gy = cgap(1, by * 8, by * 8 + 7), gz = cgap(2, bz * 8, bz * 8 + 7);
          bg2 = gx * gx + gy * gy + gz * gz;
        }
        W.stream(bs_, be_ > bs_ && !(bg2 * W.h2 > W.lim_d) ? be_ - bs_ : 0u);
      }
      break;
    }
    ++rr;
    st = 0;
    { const int side = 2 * rr + 1; nst = (side * side * side - (side - 2) * (side - 2) * (side - 2) + 63) / 64; }
  }
  W.finish();                                               // (the block sweep ends with candidates set aside; the pool is sorted for the output)
#ifdef PT_VISITS
  asm volatile("" ::"v"(W.ld)); pt_ph[3] = wall_clock64();
#endif
  if (lane < k) {
    const size_t row = (size_t)tr.id * (size_t)k;
    out_idx[row + lane] = W.li;
    if (out_d2) out_d2[row + lane] = W.ld;
  }
  if (wb.attr) {
    // the blend of pt_attr.hip's blend_one, one neighbour per lane: a single gather instruction per target, whose latency hides
    // behind the other waves' ranking (as a kernel of its own the 1.6e9 gathers of 50 M targets at k = 32 take 50 ms)
    double w = 0.0, a0 = 0.0, a1 = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
    if (lane < k && W.li != PT_NOIDX_U && W.li < wb.n_attr) {
      w = wb.mode == 1 ? __builtin_amdgcn_rcp(W.ld + 1e-12) : 1.0;      // (v_rcp_f64 / v_rsq_f64, as in the tile kernel's epilogue: no division expanded into FMAs in this kernel)
      const Attr a = pt_gather_attr(wb.attr, W.li);
      a0 = w * (double)(a.rgba & 0xFFu); a1 = w * (double)((a.rgba >> 8) & 0xFFu); a2 = w * (double)((a.rgba >> 16) & 0xFFu);
      b0 = w * (double)a.nx; b1 = w * (double)a.ny; b2 = w * (double)a.nz;
    }
#pragma unroll 1
    for (int o = 32; o > 0; o >>= 1) {
      w += __shfl_xor(w, o); a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o);
      b0 += __shfl_xor(b0, o); b1 += __shfl_xor(b1, o); b2 += __shfl_xor(b2, o);
    }
    if (lane == 0) {
      if (w > 0.0) {
        const double iw = __builtin_amdgcn_rcp(w);
        a0 *= iw; a1 *= iw; a2 *= iw; b0 *= iw; b1 *= iw; b2 *= iw;
        const double l2 = (b0 * b0 + b1 * b1) + b2 * b2;
        if (l2 >= 1e-24) { const double il = __builtin_amdgcn_rsq(l2); b0 *= il; b1 *= il; b2 *= il; }
      }
      const size_t t3 = 3 * (size_t)tr.id;
      if (wb.rgb_out) { wb.rgb_out[t3] = (float)a0; wb.rgb_out[t3 + 1] = (float)a1; wb.rgb_out[t3 + 2] = (float)a2; }
      if (wb.nrm_out) { wb.nrm_out[t3] = (float)b0; wb.nrm_out[t3 + 1] = (float)b1; wb.nrm_out[t3 + 2] = (float)b2; }
    }
  }
#ifdef PT_VISITS
  __builtin_amdgcn_wave_barrier();
  if (out_d2 && lane == 0 && k >= 4) {                      // (results are garbage in these columns)
    const size_t row = (size_t)tr.id * (size_t)k;
    out_d2[row + k - 1] = (double)W.nv * 64.0; out_d2[row + k - 2] = (double)(wall_clock64() - pt_t0); out_d2[row + k - 3] = (double)pt_t0;
    out_d2[row + k - 4] = -(double)(W.nn + 1u);            // negative: a wave-kernel row, and how many nodes it entered (+1)
    if (k >= 12) {                                          // phase times (tools/probe_wave_visits.py) and the number of sort-merges
      out_d2[row + k - 5] = (double)(pt_ph[0] - pt_t0); out_d2[row + k - 6] = (double)(pt_ph[1] - pt_ph[0]); out_d2[row + k - 7] = (double)(pt_ph[2] - pt_ph[1]);
      out_d2[row + k - 8] = (double)(pt_ph[3] - pt_ph[2]); out_d2[row + k - 9] = (double)(wall_clock64() - pt_ph[3]); out_d2[row + k - 10] = (double)W.nmerge;
    }
  }
#endif
}

// =====================================================================================================================
// Tile kernel: one 8x8x8-cell block per workgroup, candidates staged in LDS, FOUR LANES (a DPP quad) PER TARGET.
//
// Why a second kernel: the 8-lanes-per-target kernel above is VALU-issue-bound -- every step pays fp64 ranking and
// cross-lane insertion with 1/8 of the wave doing useful work.  Targets of one block share their 3x3x3 neighbourhoods,
// so the 10x10x10-cell region around the block is staged ONCE into LDS and ranked from there:
//   stage   one thread per region ROW builds the cell table (a row's cells x = 1..8 are eight consecutive keys of one
//           block); rows go HBM -> LDS by LDS-DMA with wave-uniform addresses, the 200 halo cells through registers;
//   pass 1  the lanes of a quad walk the 2x2x2 cells nearest to the target interleaved (lane q: records q, q+4, ...)
//           in fp32 and keep the K smallest VALUES only (v_med3 chain, no payload); one bitonic DPP merge + a max of
//           mins gives the quad's k-th smallest -> a proven upper bound on the exact k-th squared distance (any set of
//           >= k candidates bounds it; see `kth_bound32`);
//   pass 2  ring 1 under that bound, rows and end cells pruned in fp32; what is within the bound is appended, branch-
//           free, to the lane's own queue segment (no atomics);
//   pass 3  exact fp64 metric on the queued candidates, ranked by all-pairs counting through DPP quad broadcasts
//           (distance only; ranks that do not add up reveal equal distances and the quad recounts under (d2, index));
//           each survivor is written straight to its final slot.
// Targets that ring 1 cannot settle (k-th neighbour farther than the region guarantees, more candidates under the bound
// than the queue holds, region larger than the LDS budget) are appended to `todo` and finished by the group kernel.
// fp32 records only (the fp32 pre-filter needs exact fp32 inputs).
constexpr int TILE_R = 10, TILE_CELLS = TILE_R * TILE_R * TILE_R;
// Geometries: LARGE = 768 threads, 8448 staged records (132 KB, one workgroup per CU) for rho ~ 6-8;
//             SMALL = 512 threads, 4400 / 3888 staged records (two 80-KB workgroups per CU: one stages while the other ranks);
//             WIDE  = 512 threads, 8960 staged records, 64-entry queue, one per CU: k in 25..32.
// queue entries per quad (CAP: room for the k survivors plus whatever else the fp32 bound lets through) and per lane
// (LCAP: every lane of the quad appends to its own segment, so no atomics and no counters in LDS)
// WIDE: k in (24, 32] -- a longer queue for pass 3 (512-thread workgroups: the registers of 12 waves would not hold it)
template <int K, bool WIDE> struct TileQ { static constexpr int CAP = K == 8 ? 24 : (K == 16 ? 40 : (WIDE ? 64 : 48)), LCAP = K == 8 ? 8 : 16; };

// LDS read of one staged record as ONE ds_read_b128 (4 LDS cycles per wave-instruction).  Without the empty asm the
// compiler drops the unused id and emits ds_read_b96, which costs 8 (MI355X_MICROARCH.md, LDS table).
__device__ inline RecF lds_rec(const RecF* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  asm volatile("" ::"v"(v.w));
  RecF r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.id = __float_as_uint(v.w);
  return r;
}
// two staged records, both reads issued before either is waited for
__device__ inline void lds_rec2(const RecF* p, const RecF* q, RecF& a, RecF& b) {
  const float4 u = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(q);
  asm volatile("" ::"v"(u.w), "v"(v.w));
  a.x = u.x; a.y = u.y; a.z = u.z; a.id = __float_as_uint(u.w);
  b.x = v.x; b.y = v.y; b.z = v.z; b.id = __float_as_uint(v.w);
}
// d32 is computed from exact fp32 inputs with 3 sub, 1 mul, 2 fma: relative error < 2^-21 (all terms >= 0).
// If b = k-th smallest d32 of a candidate set, then k candidates have exact d2 <= b*(1+2^-21), so the exact k-th d2
// D_k <= b*(1+2^-21), and every candidate with exact d2 <= D_k has d32 <= b*(1+2^-21)^2 < b*(1+2^-18).
__device__ inline float kth_bound32(float b) { return b * 1.0000038146972656f + 1e-30f; }   // 1 + 2^-18, + denormal slack

template <int CTRL>
__device__ inline float dpp_f32(float v) { return __uint_as_float(dpp_u32<CTRL>(__float_as_uint(v))); }
constexpr int DPP_QP_1032 = 0xB1;    // quad_perm [1,0,3,2]: partner lane ^ 1
constexpr int DPP_QP_2301 = 0x4E;    // quad_perm [2,3,0,1]: partner lane ^ 2
constexpr int DPP_QP_0000 = 0x00, DPP_QP_1111 = 0x55, DPP_QP_2222 = 0xAA, DPP_QP_3333 = 0xFF;

// merge my ascending list with the partner lane's: afterwards both lanes hold the K smallest of the 2K values, ascending
template <int K, int CTRL>
__device__ inline void quad_merge_sorted(float (&l)[K]) {
#pragma unroll
  for (int j = 0; j < K / 2; ++j) {                    // bitonic: lowest K of the union, in place (pairs j, K-1-j)
    const float a = l[j], b = l[K - 1 - j];
    const float pa = dpp_f32<CTRL>(b), pb = dpp_f32<CTRL>(a);
    l[j] = fminf(a, pa); l[K - 1 - j] = fminf(b, pb);
  }
#pragma unroll
  for (int d = K / 2; d >= 1; d >>= 1) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if ((j & d) == 0) { const float lo = fminf(l[j], l[j + d]), hi = fmaxf(l[j], l[j + d]); l[j] = lo; l[j + d] = hi; }
    }
  }
}

// one LDS-DMA wave-instruction: active lane L copies 16 bytes from its own `g` to `lbase + L` (lbase wave-uniform)
__device__ inline void glds16(const uint4* g, uint4* lbase) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lbase, 16, 0, 0);
}

// attribute blend fused into the tile kernel (BLEND): the table, its length, the mode and the two outputs
struct TileBlend { const Attr* attr; uint32_t n_attr; int mode; float* rgb_out; float* nrm_out; };
// Second chance for blocks whose region is over this geometry's LDS budget but within the large geometry's: their ids go
// to `retry` (retry != null), and a second launch (blocks != null: blockIdx.x indexes that list) takes them.
struct TileBlocks { const uint32_t* blocks; uint32_t* retry; uint32_t* retry_n; uint32_t retry_cap; const double* bound; };
// BND (fp32 clouds, no fused blend): every target brings a radius bound[id] -- the k-th squared distance it already has from another
// part of the cloud (the chunks of a streamed source, pt_stream_query) -- and only points with d2 <= bound matter.  The bound joins
// pass 1's own (whichever is smaller prunes pass 2), settles targets whose k-th neighbour lies beyond ring 1 when the bound does not,
// and lets targets with fewer than k points in reach finish with a short list (the rest NOIDX / +inf, as a bounded query returns).
// fp64 clouds (DBL): the LDS image and the two fp32 passes work on fp32-ROUNDED coordinates (the build's shadow records,
// whose id field is the sorted position), under a bound widened by the rounding; pass 3 fetches the exact 32-byte
// records of the queued candidates by position.  src / tgt: the exact records; e_src: largest |coordinate| rounding
// error of a source point (2^-24 * largest |coordinate| of the cloud's bounding box).
struct TileDouble { const RecD* src; const RecD* tgt; float e_src; };

// KC: length of pass 1's per-lane value chain (<= K).  The merges and the ranking run at width K (a power of two); a chain of KC
// entries leaves l32[KC..K) at +inf, which is all a query with k <= KC needs: the reference's K = 20 runs the K = 32 body with
// a 24-deep chain (three quarters of pass 1's per-candidate work).
template <int K, int TILE_CAP, int TWG, bool WIDE = false, bool BLEND = false, bool DBL = false, int KC = K, bool BND = false>
__global__ __launch_bounds__(TWG, TILE_CAP > 5000 ? 1 : (TWG == 384 ? 3 : 4)) void knn_tile_kernel(GridParams gp, const RecF* __restrict__ src, const uint32_t* __restrict__ cs,
                                                        const RecF* __restrict__ tgt, const uint32_t* __restrict__ tblock_start, int k,
                                                        uint32_t* __restrict__ out_idx, double* __restrict__ out_d2,
                                                        uint32_t* __restrict__ todo, uint32_t* __restrict__ todo_n, TileBlend bl, TileBlocks tb,
                                                        TileDouble dd) {
  constexpr int NW = TWG / 64;
  constexpr int TILE_QUADS = TWG / 4;
  constexpr int TILE_QCAP = TileQ<K, WIDE>::CAP, TILE_LCAP = TileQ<K, WIDE>::LCAP;
  __shared__ __attribute__((aligned(16))) RecF lrec[TILE_CAP];
  __shared__ uint16_t lstart[TILE_CELLS + 8];
  __shared__ __attribute__((aligned(16))) uint16_t queue[TWG * (TILE_LCAP + 1)];          // doubles as rowdesc[] during staging
  __shared__ uint32_t wsum[NW];
  __shared__ uint32_t ptotal;
  // per region row: global starts of its left halo cell / its run of 8 cells / its right halo cell, and the LDS offsets
  // of cells 0, 1, 9 and of the next row (two 16-bit pairs)
  uint4* rowdesc = reinterpret_cast<uint4*>(queue);
  uint32_t* rowdesc_e = reinterpret_cast<uint32_t*>(queue) + 4 * TILE_R * TILE_R;
  static_assert(sizeof(queue) >= 5 * TILE_R * TILE_R * sizeof(uint32_t), "rowdesc aliases the queue");
  static_assert(TILE_CAP < 65536, "LDS offsets are 16-bit");

  uint32_t b = blockIdx.x;
  if (tb.blocks) b = tb.blocks[blockIdx.x];
  const uint32_t ts = tblock_start[b], te = tblock_start[b + 1];   // (waited for only after the cell-table loads below are out)
  // block id -> cell origin of the block
  const uint32_t macro = b >> 9, m9 = b & 511u;
  const int bx = (int)(macro % (uint32_t)gp.mdim[0]) * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u));
  const int by = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]) * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u));
  const int bz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1])) * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u));
#if defined(PT_ABLATE) && PT_ABLATE == 4
  // timing-only build: every workgroup stages the region of one of 512 HOT blocks (one macro block in the middle of the grid:
  // L2 / Infinity-Cache resident) and its targets are shifted into that block -- same ranking work, no HBM traffic for staging
  const uint32_t hmacro = (uint32_t)(((gp.mdim[2] / 2) * gp.mdim[1] + gp.mdim[1] / 2) * gp.mdim[0] + gp.mdim[0] / 2), hm9 = b & 511u;
  const int hbx = (int)(hmacro % (uint32_t)gp.mdim[0]) * 8 + (int)((hm9 & 1u) | ((hm9 >> 2) & 2u) | ((hm9 >> 4) & 4u));
  const int hby = (int)((hmacro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]) * 8 + (int)(((hm9 >> 1) & 1u) | ((hm9 >> 3) & 2u) | ((hm9 >> 5) & 4u));
  const int hbz = (int)(hmacro / (uint32_t)(gp.mdim[0] * gp.mdim[1])) * 8 + (int)(((hm9 >> 2) & 1u) | ((hm9 >> 4) & 2u) | ((hm9 >> 6) & 4u));
  const double hshift[3] = {(double)((hbx - bx) * 8) * gp.h, (double)((hby - by) * 8) * gp.h, (double)((hbz - bz) * 8) * gp.h};
  const int ox = hbx * 8 - 1, oy = hby * 8 - 1, oz = hbz * 8 - 1;
#else
  const int ox = bx * 8 - 1, oy = by * 8 - 1, oz = bz * 8 - 1;          // cell coordinates of region cell (0,0,0)
#endif

  // ---- A: region cell table (global start + LDS offset of each of the 1000 cells).  One thread per region ROW (y, z):
  //         cells x = 1..8 of a row are eight consecutive keys of one block, so a row needs three key computations
  //         (left halo cell, the run, right halo cell) and 13 table words.  Waves 0 and 1 do this; the rest go to the barrier.
  constexpr int NROWS = TILE_R * TILE_R;
  if (threadIdx.x < 128) {
    const int row = threadIdx.x;
    uint32_t g[TILE_R], cnt[TILE_R], sum = 0;
#pragma unroll
    for (int i = 0; i < TILE_R; ++i) { g[i] = 0; cnt[i] = 0; }
    if (row < NROWS) {
      const int y = oy + row % TILE_R, z = oz + row / TILE_R;
      if (y >= 0 && y < gp.dim[1] && z >= 0 && z < gp.dim[2]) {
        const uint32_t km = cell_key(gp, ox + 1, y, z);              // cells ox+1 .. ox+8: keys km .. km+7 (32-byte aligned)
        const uint4 m0 = *reinterpret_cast<const uint4*>(cs + km), m1 = *reinterpret_cast<const uint4*>(cs + km + 4);
        const uint32_t m8 = cs[km + 8];
        uint32_t l0 = 0, l1 = 0, r0 = 0, r1 = 0;
        if (ox >= 0) { const uint32_t kl = cell_key(gp, ox, y, z); l0 = cs[kl]; l1 = cs[kl + 1]; }
        if (ox + 9 < gp.dim[0]) { const uint32_t kr = cell_key(gp, ox + 9, y, z); r0 = cs[kr]; r1 = cs[kr + 1]; }
        g[0] = l0; g[1] = m0.x; g[2] = m0.y; g[3] = m0.z; g[4] = m0.w; g[5] = m1.x; g[6] = m1.y; g[7] = m1.z; g[8] = m1.w; g[9] = r0;
        cnt[0] = l1 - l0; cnt[9] = r1 - r0;
        cnt[1] = m0.y - m0.x; cnt[2] = m0.z - m0.y; cnt[3] = m0.w - m0.z; cnt[4] = m1.x - m0.w;
        cnt[5] = m1.y - m1.x; cnt[6] = m1.z - m1.y; cnt[7] = m1.w - m1.z; cnt[8] = m8 - m1.w;
      }
#pragma unroll
      for (int i = 0; i < TILE_R; ++i) sum += cnt[i];
    }
    // INVARIANT: ts and te are loaded from tblock_start[b] with b a function of blockIdx.x only, so they are the same in
    // every lane of every wave of the workgroup: either ALL waves return here (and in the else branch below) or none does,
    // and every wave that stays executes exactly one s_barrier in its branch -- the table waves the one between their scan
    // halves, the other waves the one in the else branch -- before all of them meet again at the __syncthreads() below.
    // (The test sits here rather than at the top so that the cell-table loads are in flight while ts / te arrive.)
    if (ts == te) return;                               // no targets in this block
    const uint32_t incl = wave_incl_scan(sum);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();                       // all waves: waves >= 2 execute the matching s_barrier in the else branch
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const uint32_t w0 = wsum[0], w1 = wsum[1];
    uint32_t ex = (threadIdx.x >= 64 ? w0 : 0u) + incl - sum;
    if (row < NROWS) {
      uint32_t lo[TILE_R + 1];
#pragma unroll
      for (int i = 0; i < TILE_R; ++i) {
        lo[i] = ex < 65535u ? ex : 65535u;
        lstart[row * TILE_R + i] = (uint16_t)lo[i];
        ex += cnt[i];
      }
      lo[TILE_R] = ex < 65535u ? ex : 65535u;
      rowdesc[row] = make_uint4(g[0], g[1], g[9], lo[0] | (lo[1] << 16));
      rowdesc_e[row] = lo[9] | (lo[TILE_R] << 16);
    }
    if (threadIdx.x == 0) { const uint32_t tot = w0 + w1; lstart[TILE_CELLS] = (uint16_t)(tot < 65535u ? tot : 65535u); ptotal = tot; }
  } else {
    if (ts == te) return;                               // same workgroup-uniform test as above
    __builtin_amdgcn_s_barrier();                       // pairs with the barrier between the two table waves' scan halves
  }
  __syncthreads();
  const uint32_t P = ptotal;
  if (P > (uint32_t)TILE_CAP) {                     // denser than the LDS budget
    bool again = false;
    if constexpr (!WIDE) again = tb.retry && P <= tb.retry_cap;
    if (again) {                                    // ... but not than the large geometry's: that launch takes the block
      if constexpr (!WIDE) { if (threadIdx.x == 0) tb.retry[atomicAdd(tb.retry_n, 1u)] = b; }
    } else {                                        // the group kernel takes the whole tile
      for (uint32_t t = ts + threadIdx.x; t < te; t += TWG) todo[atomicAdd(todo_n, 1u)] = t;
    }
    return;
  }
  // the first round's target of this quad: requested here so that it arrives during the staging (loaded where it is first used,
  // every round began with a memory latency that nothing else of the wave could cover)
  RecF tr_first;
  tr_first.x = tr_first.y = tr_first.z = 0.f; tr_first.id = 0;
  if constexpr (!DBL) { if (ts + (threadIdx.x >> 2) < te) tr_first = tgt[ts + (threadIdx.x >> 2)]; }
  // ---- B: stage the region, HBM -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + lane * 16, per-lane
  //         source address; no staging registers).  Cells x = 1..8 of a region row are one contiguous run in HBM and in
  //         LDS: two DMA instructions per row (<= 128 records) with wave-uniform (scalar) addresses; the 200 halo cells
  //         (x = 0 and 9) follow through registers.  Every load of the tile is in flight before the first wait.
  {
    const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(src);     // records move as raw 16-byte words
    uint4* l4 = reinterpret_cast<uint4*>(lrec);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int RPW = (NROWS + NW - 1) / NW;                      // rows per wave (9 or 13)
    bool long_rows = false;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      const int rr = w + i * NW;
      if (rr < NROWS) {                                             // wave-uniform
        const uint32_t g1 = __builtin_amdgcn_readfirstlane(rowdesc[rr].y);
        const uint32_t p01 = __builtin_amdgcn_readfirstlane(rowdesc[rr].w), p9 = __builtin_amdgcn_readfirstlane(rowdesc_e[rr]);
        const uint32_t la1 = p01 >> 16, nm = (p9 & 0xFFFFu) - la1;
        if ((uint32_t)lane < nm) glds16(src4 + g1 + lane, l4 + la1);
        if ((uint32_t)lane + 64u < nm) glds16(src4 + g1 + 64 + lane, l4 + la1 + 64u);
        long_rows |= nm > 128u;
      }
    }
    // halo cells: 16 records would waste a 64-lane DMA each, so 8-lane groups move them through registers
    const int g8 = threadIdx.x >> 3, l8 = threadIdx.x & 7;
    constexpr int NHALO = 2 * NROWS;
    constexpr int HC = (NHALO + TWG / 8 - 1) / (TWG / 8);           // halo cells per 8-lane group
    uint4 h0[HC], h1[HC];
    uint32_t hla[HC], hlen[HC];
    bool long_cells = false;
#pragma unroll
    for (int i = 0; i < HC; ++i) {
      const int hcr = g8 + i * (TWG / 8);
      const int hc = hcr < NHALO ? hcr : NHALO - 1;
      const int c = (hc >> 1) * TILE_R + ((hc & 1) ? TILE_R - 1 : 0);
      hla[i] = lstart[c];
      hlen[i] = hcr < NHALO ? lstart[c + 1] - hla[i] : 0u;
      const uint32_t ga = (hc & 1) ? rowdesc[hc >> 1].z : rowdesc[hc >> 1].x;
      h0[i] = make_uint4(0, 0, 0, 0); h1[i] = make_uint4(0, 0, 0, 0);
      if ((uint32_t)l8 < hlen[i]) h0[i] = src4[ga + l8];
      if ((uint32_t)l8 + 8u < hlen[i]) h1[i] = src4[ga + l8 + 8u];
      long_cells |= hlen[i] > 16u;
    }
#pragma unroll
    for (int i = 0; i < HC; ++i) {
      if ((uint32_t)l8 < hlen[i]) l4[hla[i] + l8] = h0[i];
      if ((uint32_t)l8 + 8u < hlen[i]) l4[hla[i] + l8 + 8u] = h1[i];
    }
    if (long_rows) {                                                // very dense rows: the rest synchronously
      for (int rr = w; rr < NROWS; rr += NW) {
        const uint32_t g1 = rowdesc[rr].y, la1 = rowdesc[rr].w >> 16, nm = (rowdesc_e[rr] & 0xFFFFu) - la1;
        for (uint32_t p = lane + 128u; p < nm; p += 64) l4[la1 + p] = src4[g1 + p];
      }
    }
    if (long_cells) {
      for (int hc = g8; hc < NHALO; hc += TWG / 8) {
        const int c = (hc >> 1) * TILE_R + ((hc & 1) ? TILE_R - 1 : 0);
        const uint32_t a0 = lstart[c], n0 = lstart[c + 1] - a0, ga = (hc & 1) ? rowdesc[hc >> 1].z : rowdesc[hc >> 1].x;
        for (uint32_t p = l8 + 16u; p < n0; p += 8) l4[a0 + p] = src4[ga + p];
      }
    }
  }
  __syncthreads();                                  // rowdesc is dead from here on: the queue takes its place
#if defined(PT_ABLATE) && PT_ABLATE == 1
  if (lrec[threadIdx.x % (P ? P : 1u)].id == 0xFFFFFFFEu) out_idx[0] = 1;   // keeps the staging alive
  return;                                           // timing-only build: staging cost alone (results are garbage)
#endif

  // ---- C: four lanes per target, 192 targets per round ----------------------------------------------------------------
  const double h2 = gp.h * gp.h;
  const int quad = threadIdx.x >> 2, ql = threadIdx.x & 3;
  for (uint32_t base = ts; base < te; base += TILE_QUADS) {
    const uint32_t t = base + quad;
    const bool active = t < te;                                        // whole quads are active or not
    RecF tr;
    tr.x = tr.y = tr.z = 0.f; tr.id = 0;
    double q[3] = {0.0, 0.0, 0.0};
    if constexpr (DBL) {
      if (active) { const RecD td = dd.tgt[t]; q[0] = td.x; q[1] = td.y; q[2] = td.z; tr.x = (float)td.x; tr.y = (float)td.y; tr.z = (float)td.z; tr.id = td.id; }
    } else {
      if (base == ts) tr = tr_first;                    // (workgroup-uniform test)
      else if (active) tr = tgt[t];
#if defined(PT_ABLATE) && PT_ABLATE == 4
      tr.x = (float)((double)tr.x + hshift[0]); tr.y = (float)((double)tr.y + hshift[1]); tr.z = (float)((double)tr.z + hshift[2]);
#endif
      q[0] = (double)tr.x; q[1] = (double)tr.y; q[2] = (double)tr.z;
    }
    double u[3];
    int cc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      u[a] = (q[a] - gp.bbmin[a]) * gp.inv_h;
      cc[a] = (int)fmin(fmax(u[a], 0.0), (double)(gp.dim[a] - 1));
    }
    const int rx = active ? cc[0] - ox : 1, ry = active ? cc[1] - oy : 1, rz = active ? cc[2] - oz : 1;   // in [1, 8]
    const int cbase = (rz * TILE_R + ry) * TILE_R + (rx - 1);          // region cell left of the target's cell
    // The four lanes of a quad walk every run of records together, lane ql taking records ql, ql+4, ... of it: the quad
    // reads 64 contiguous bytes per step, every lane sees a quarter of every cell (even shares), and the trip counts are
    // the same for the whole quad.

    // ---- pass 1: the K smallest fp32 distances (values only) of the 2x2x2 cells nearest to the target -- on each axis
    //      the target's cell and its neighbour on the target's side.  Any candidate set with >= k members gives a valid
    //      bound; this one holds most of the k nearest at a quarter of ring 1's reads.  (Fewer than k points there: the
    //      bound is +inf, the queue overflows and the target goes to the todo list.)  Per-lane lists shorter than K would
    //      still be valid but loosen the bound: measured 3.7 % of the targets overflow the queue with 3K/4, 17 % with K/2. ----
    const int sx = (u[0] - (double)cc[0]) >= 0.5 ? 0 : -1, sy = (u[1] - (double)cc[1]) >= 0.5 ? 0 : -1, sz = (u[2] - (double)cc[2]) >= 0.5 ? 0 : -1;
    double bnd = INFINITY;
    if constexpr (BND) {
      static_assert(!BLEND, "the bounded variant is built without the fused blend");
      if (active) bnd = tb.bound[tr.id];
    }
    const bool scan1 = active && !(BND && bnd < 0.0);      // (a negative bound -- "nothing from this cloud" -- skips pass 1 too; pass 2 prunes itself)
    float l32[K];
#pragma unroll
    for (int j = 0; j < K; ++j) l32[j] = INFINITY;
    auto push1 = [&](float x) {
      float prev = l32[0];
      l32[0] = fminf(x, prev);
#pragma unroll
      for (int j = 1; j < KC; ++j) { const float cur = l32[j]; l32[j] = __builtin_amdgcn_fmed3f(x, prev, cur); prev = cur; }
    };
    {
      uint32_t ps[4], pe[4];
#pragma unroll
      for (int o = 0; o < 4; ++o) {                    // all eight table reads in flight together
        const int c = cbase + ((sz + (o >> 1)) * TILE_R + (sy + (o & 1))) * TILE_R + 1 + sx;
        ps[o] = (uint32_t)lstart[c] + ql;
        pe[o] = scan1 ? (uint32_t)lstart[c + 2] : 0u;
      }
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        uint32_t p = ps[o];
        const uint32_t e = pe[o];
        for (; p + 4 < e; p += 8) {
          RecF a, b;
          lds_rec2(&lrec[p], &lrec[p + 4], a, b);
          push1(dist2_f32(tr.x, tr.y, tr.z, a)); push1(dist2_f32(tr.x, tr.y, tr.z, b));
        }
        if (p < e) push1(dist2_f32(tr.x, tr.y, tr.z, lds_rec(&lrec[p])));
      }
    }
    // the quad's K smallest: two bitonic merges through DPP (all lanes of the wave take part: no divergence here)
    quad_merge_sorted<K, DPP_QP_1032>(l32);
    float kv;
    if (k == K) {                                      // only the K-th smallest is wanted: the largest of the bitonic lower half
      kv = fminf(l32[0], dpp_f32<DPP_QP_2301>(l32[K - 1]));
#pragma unroll
      for (int j = 1; j < K; ++j) kv = fmaxf(kv, fminf(l32[j], dpp_f32<DPP_QP_2301>(l32[K - 1 - j])));
    } else {
      quad_merge_sorted<K, DPP_QP_2301>(l32);
      kv = l32[0];
#pragma unroll
      for (int j = 1; j < K; ++j) kv = (j == k - 1) ? l32[j] : kv;
    }
    float thr = kth_bound32(kv);
    if constexpr (BND) {
      // a candidate with exact d2 <= bnd has d32 <= bnd (1 + 2^-21): the bound rounded UP to fp32, times 1 + 2^-20
      thr = bnd < 0.0 ? -1.f : fminf(thr, __double2float_ru(bnd) * 1.000001f + 1e-30f);
    }
    if constexpr (DBL) {
      // Rounded coordinates move every difference by at most E per axis (source + target rounding), i.e. every distance
      // by at most sqrt(3) E: k candidates lie within sqrt(kv') + sqrt(3) E of the target, so the true top k do, and
      // their rounded distances are within another sqrt(3) E.  (1.0000005 covers sqrtf's rounding.)  A caller's bound (BND) is on
      // EXACT distances: what it lets through has a rounded distance within ONE sqrt(3) E of it, so the same widening covers it.
      if (!BND || thr >= 0.f) {
        const float e_t = 5.9604645e-8f * fmaxf(fmaxf(fabsf(tr.x), fabsf(tr.y)), fabsf(tr.z));
        const float r = sqrtf(thr) * 1.0000005f + 3.4642f * (dd.e_src + e_t) * 1.000001f;
        thr = r * r * 1.0000039f + 1e-30f;
      }
    }
#if defined(PT_ABLATE) && PT_ABLATE == 2
    if (thr >= 0.f) continue;                       // timing-only build: staging + pass 1 (results are garbage)
#endif

    // ---- pass 2: scan ring 1 under the bound; what is within it goes to this lane's own queue segment.  Rows and their end
    //      cells are pruned in fp32 on the target's position inside its cell, with gaps UNDER-estimated by a slack far above
    //      the rounding of the products, so nothing that could hold a candidate within the bound is skipped. ----
    const float fx = (float)(u[0] - (double)cc[0]), fy = (float)(u[1] - (double)cc[1]), fz = (float)(u[2] - (double)cc[2]);
    const float h2f = (float)h2;
    auto gapf = [&](float f, int d) -> float {         // distance (in cells) from offset f in the centre cell to cell d = -1, 0, +1
      const float g = fmaxf((float)d - f, f - (float)(d + 1)) - 4e-6f * (1.f + fabsf(f));
      return fmaxf(g, 0.f);
    };
    const float g2x[3] = {gapf(fx, -1) * gapf(fx, -1), 0.f, gapf(fx, 1) * gapf(fx, 1)};
    const float g2y[3] = {gapf(fy, -1) * gapf(fy, -1), gapf(fy, 0) * gapf(fy, 0), gapf(fy, 1) * gapf(fy, 1)};
    const float g2z[3] = {gapf(fz, -1) * gapf(fz, -1), gapf(fz, 0) * gapf(fz, 0), gapf(fz, 1) * gapf(fz, 1)};
    // branch-free append: the position is always stored at the segment's next slot and the count only moves when the
    // candidate is within the bound (a rejected one is overwritten by its successor); slot TILE_LCAP takes the spill.
    uint16_t* myq = &queue[threadIdx.x * (TILE_LCAP + 1)];
    uint32_t nmine = 0;
    auto push2 = [&](float x, uint32_t p) {
      myq[nmine < (uint32_t)TILE_LCAP ? nmine : (uint32_t)TILE_LCAP] = (uint16_t)p;
      nmine += (x <= thr) ? 1u : 0u;
    };
    {
      uint32_t qs[9], qe[9];
#pragma unroll
      for (int r = 0; r < 9; ++r) {                    // the runs of all nine rows first: their table reads overlap
        const int dy = r % 3 - 1, dz = r / 3 - 1;
        const int c = cbase + (dz * TILE_R + dy) * TILE_R;
        const float s2 = g2y[dy + 1] + g2z[dz + 1];
        const bool row_on = active && !(s2 * h2f > thr);
        const bool lo_on = !((g2x[0] + s2) * h2f > thr), hi_on = !((g2x[2] + s2) * h2f > thr);
        qe[r] = row_on ? (uint32_t)lstart[hi_on ? c + 3 : c + 2] : 0u;
        qs[r] = (uint32_t)lstart[lo_on ? c : c + 1] + ql;
      }
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        uint32_t p = qs[r];
        const uint32_t e = qe[r];
        for (; p + 4 < e; p += 8) {
          RecF a, b;
          lds_rec2(&lrec[p], &lrec[p + 4], a, b);
          push2(dist2_f32(tr.x, tr.y, tr.z, a), p); push2(dist2_f32(tr.x, tr.y, tr.z, b), p + 4);
        }
        if (p < e) push2(dist2_f32(tr.x, tr.y, tr.z, lds_rec(&lrec[p])), p);
      }
    }
    // A quad's segments are written and read by lanes of ONE wave: the LDS executes a wave's operations in issue order and
    // the scans above have reconverged, so no workgroup barrier is needed -- only a compiler fence.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- pass 3: exact metric, ranking by all-pairs counting inside the quad ----
    const uint32_t n0 = dpp_u32<DPP_QP_0000>(nmine), n1 = dpp_u32<DPP_QP_1111>(nmine), n2 = dpp_u32<DPP_QP_2222>(nmine),
                   n3 = dpp_u32<DPP_QP_3333>(nmine);
    const uint32_t p1 = n0, p2 = n0 + n1, p3 = p2 + n2, nq = p3 + n3;
    const bool overflow = nq > (uint32_t)TILE_QCAP || n0 > (uint32_t)TILE_LCAP || n1 > (uint32_t)TILE_LCAP || n2 > (uint32_t)TILE_LCAP ||
                          n3 > (uint32_t)TILE_LCAP;                                 // quad-uniform by construction
#if defined(PT_ABLATE) && PT_ABLATE == 3
    if (nq < 1000u) continue;                          // timing-only build: staging + passes 1, 2
#endif
    double od[TILE_QCAP / 4];
    uint32_t oi[TILE_QCAP / 4];
    int rk[TILE_QCAP / 4];
#pragma unroll
    for (int j = 0; j < TILE_QCAP / 4; ++j) {          // my entries of the concatenated segments: ql, ql+4, ...
      const uint32_t e = (uint32_t)(4 * j + ql);
      od[j] = INFINITY; oi[j] = PT_NOIDX_U; rk[j] = 0;
      if (e < nq && !overflow) {
        const uint32_t seg = (uint32_t)(e >= p1) + (uint32_t)(e >= p2) + (uint32_t)(e >= p3);
        const uint32_t off = e - (seg == 0 ? 0u : (seg == 1 ? p1 : (seg == 2 ? p2 : p3)));
        const RecF r = lrec[queue[((threadIdx.x & ~3u) + seg) * (TILE_LCAP + 1) + off]];
        if constexpr (DBL) { const RecD rd = dd.src[r.id]; od[j] = dist2(q, rd); oi[j] = rd.id; }     // r.id: sorted position of the exact record
        else { od[j] = dist2(q, r); oi[j] = r.id; }
      }
    }
    // Ranking counts, for each of my entries, the queue entries with a smaller distance.  Equal distances (rare) leave
    // two entries with the same count: the ranks then do not add up to 0 + 1 + ... + (nq-1) and the quad redoes the count
    // under the full order (d2, index).
    auto rank_all = [&](auto lt) {
#pragma unroll
      for (int j = 0; j < TILE_QCAP / 4; ++j) {        // round j: the four lanes' j-th entries visit every lane
        if ((uint32_t)(4 * j) < nq) {                  // quad-uniform
          const double b0 = dpp_f64<DPP_QP_0000>(od[j]), b1 = dpp_f64<DPP_QP_1111>(od[j]), b2 = dpp_f64<DPP_QP_2222>(od[j]),
                       b3 = dpp_f64<DPP_QP_3333>(od[j]);
          const uint32_t i0 = dpp_u32<DPP_QP_0000>(oi[j]), i1 = dpp_u32<DPP_QP_1111>(oi[j]), i2 = dpp_u32<DPP_QP_2222>(oi[j]),
                         i3 = dpp_u32<DPP_QP_3333>(oi[j]);
#pragma unroll
          for (int m = 0; m < TILE_QCAP / 4; ++m) {
            if ((uint32_t)(4 * m) < nq)                // quad-uniform: slots beyond the queue hold +inf and rank nowhere
              rk[m] += (int)lt(b0, i0, od[m], oi[m]) + (int)lt(b1, i1, od[m], oi[m]) + (int)lt(b2, i2, od[m], oi[m]) +
                       (int)lt(b3, i3, od[m], oi[m]);
          }
        }
      }
    };
    if constexpr (K > 16 && (WIDE || (BLEND && DBL))) {   // (register budget of the wide and of the fp64 + blend variants: one ranking body only)
      rank_all([](double ad, uint32_t ai, double bd, uint32_t bi) { return key_lt(ad, ai, bd, bi); });
    } else {
      rank_all([](double ad, uint32_t, double bd, uint32_t) { return ad < bd; });
      int rs = 0;
#pragma unroll
      for (int j = 0; j < TILE_QCAP / 4; ++j) rs += (oi[j] != PT_NOIDX_U) ? rk[j] : 0;
      rs += (int)dpp_u32<DPP_QP_1032>((uint32_t)rs);
      rs += (int)dpp_u32<DPP_QP_2301>((uint32_t)rs);
      const uint32_t nv = overflow ? 0u : nq;
      if ((uint32_t)rs != nv * (nv - 1u) / 2u) {       // quad-uniform
#pragma unroll
        for (int j = 0; j < TILE_QCAP / 4; ++j) rk[j] = 0;
        rank_all([](double ad, uint32_t ai, double bd, uint32_t bi) { return key_lt(ad, ai, bd, bi); });
      }
    }
    // exact k-th squared distance of ring 1 (rank k-1), known to one lane -> quad minimum
    double kd = INFINITY;
#pragma unroll
    for (int j = 0; j < TILE_QCAP / 4; ++j) if (oi[j] != PT_NOIDX_U && rk[j] == k - 1) kd = od[j];
    kd = fmin(kd, dpp_f64<DPP_QP_1032>(kd));
    kd = fmin(kd, dpp_f64<DPP_QP_2301>(kd));
    bool covered = true;
    double dout = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int lo = cc[a] - 1, hi = cc[a] + 1;
      if (lo > 0) { covered = false; dout = fmin(dout, u[a] - (double)lo); }
      if (hi < gp.dim[a] - 1) { covered = false; dout = fmin(dout, (double)(hi + 1) - u[a]); }
    }
    dout = fmax(dout - PT_CELL_EPS, 0.0);
    const bool done = !overflow && (covered || dout * dout * h2 > (BND ? fmin(kd, bnd) : kd));
    if (active) {
      if (done) {
        const size_t row = (size_t)tr.id * (size_t)k;
        // (BLEND) the k neighbours' attribute records are gathered right here, GB of a lane's gathers issued before anything waits
        // for one, the first group before the result stores (the memory counter is in order: a load behind a store waits for it).
        // Gather, wait, accumulate per neighbour -- the first form of this -- cost a lane six random-access latencies in a row.
        // GB: all six entries at k <= 8, four at a time beyond (registers); the fp64 + blend variants have none left and keep
        // gathering one by one.  The loads are unconditional -- an entry that is not among the k reads record 0, one cached line
        // for the whole chip -- because behind a branch each the compiler still put a full wait between them.
        constexpr int NE = TILE_QCAP / 4;
        constexpr int GB = (BLEND && !DBL) ? (NE <= 6 ? NE : 4) : 1;
        auto store_results = [&]() {
#pragma unroll
          for (int j = 0; j < NE; ++j)
            if (oi[j] != PT_NOIDX_U && rk[j] < k) { out_idx[row + rk[j]] = oi[j]; if (out_d2) out_d2[row + rk[j]] = od[j]; }
          for (uint32_t sl = nq + ql; sl < (uint32_t)k; sl += 4) { out_idx[row + sl] = PT_NOIDX_U; if (out_d2) out_d2[row + sl] = INFINITY; }
        };
        if constexpr (!BLEND) store_results();
        if constexpr (BLEND) {
          // blended as pt_attr.hip's blend_kernel does: fp64 sums, then one normalisation
          double ws = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0, n0 = 0.0, n1 = 0.0, n2 = 0.0;
#pragma unroll
          for (int g0 = 0; g0 < NE; g0 += GB) {
            Attr at[GB];
            if constexpr (GB > 1) {
#pragma unroll
              for (int q = 0; q < GB; ++q) {
                const int j = g0 + q < NE ? g0 + q : NE - 1;
                at[q] = pt_gather_attr(bl.attr, (g0 + q < NE && oi[j] != PT_NOIDX_U && rk[j] < k && oi[j] < bl.n_attr) ? oi[j] : 0u);
              }
            }
            if (g0 == 0) store_results();
#pragma unroll
            for (int q = 0; q < GB; ++q) {
              const int j = g0 + q < NE ? g0 + q : NE - 1;
              if (g0 + q < NE && oi[j] != PT_NOIDX_U && rk[j] < k && oi[j] < bl.n_attr) {
                const double w = (bl.mode == 1) ? 1.0 / (od[j] + 1e-12) : 1.0;
                Attr a;
                if constexpr (GB > 1) a = at[q]; else a = pt_gather_attr(bl.attr, oi[j]);
                ws += w;
                c0 += w * (double)(a.rgba & 0xFFu); c1 += w * (double)((a.rgba >> 8) & 0xFFu); c2 += w * (double)((a.rgba >> 16) & 0xFFu);
                n0 += w * (double)a.nx; n1 += w * (double)a.ny; n2 += w * (double)a.nz;
              }
            }
          }
          auto quad_sum = [](double v) { v += dpp_f64<DPP_QP_1032>(v); v += dpp_f64<DPP_QP_2301>(v); return v; };
          ws = quad_sum(ws); c0 = quad_sum(c0); c1 = quad_sum(c1); c2 = quad_sum(c2); n0 = quad_sum(n0); n1 = quad_sum(n1); n2 = quad_sum(n2);
          if (ws > 0.0) {
            // The sums above are fp64 (normals may cancel); the finishing touches use the hardware reciprocal and
            // reciprocal square root (v_rcp_f64 / v_rsq_f64, ~2^-23 relative: two orders inside the 1e-5 bar) instead of
            // four fp64 divisions and a square root, which were a third of this epilogue's instructions.
            const double iw = __builtin_amdgcn_rcp(ws);
            c0 *= iw; c1 *= iw; c2 *= iw;
            const double l2 = n0 * n0 + n1 * n1 + n2 * n2;
            const double sc = (l2 * iw * iw >= 1e-24) ? __builtin_amdgcn_rsq(l2) : iw;      // |mean normal| >= 1e-12: renormalise
            n0 *= sc; n1 *= sc; n2 *= sc;
          }
          float* o = (ql == 0) ? bl.rgb_out : bl.nrm_out;
          if (ql < 2 && o) {
            o[3 * (size_t)tr.id] = (float)(ql == 0 ? c0 : n0); o[3 * (size_t)tr.id + 1] = (float)(ql == 0 ? c1 : n1);
            o[3 * (size_t)tr.id + 2] = (float)(ql == 0 ? c2 : n2);
          }
        }
      } else if (ql == 0) {
        todo[atomicAdd(todo_n, 1u)] = t;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // next round's segment writes stay behind this round's reads
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- G-way merge of candidate lists under (d2, idx): one thread per (target, list slot) --------------
__global__ __launch_bounds__(WG) void merge_kernel(const uint32_t* __restrict__ idx_lists, const double* __restrict__ d2_lists, int g,
                                                   uint32_t m, int k, uint32_t* __restrict__ idx_out, double* __restrict__ d2_out) {
  const int per = g * k;
  const uint64_t gt = (uint64_t)blockIdx.x * WG + threadIdx.x;
  const uint32_t t = (uint32_t)(gt / (uint64_t)per);
  if (t >= m) return;
  const int slot = (int)(gt % (uint64_t)per);
  const int mg = slot / k, mj = slot % k;
  const size_t mo = ((size_t)mg * m + t) * (size_t)k + (size_t)mj;
  const uint32_t myi = idx_lists[mo];
  const double myd = d2_lists[mo];
  int rank = 0, cnt = 0;
  for (int r = 0; r < g; ++r)
    for (int j = 0; j < k; ++j) {
      const size_t o = ((size_t)r * m + t) * (size_t)k + (size_t)j;
      const uint32_t oi = idx_lists[o];
      if (oi == PT_NOIDX_U) continue;
      ++cnt;
      const double od = d2_lists[o];
      if (key_lt(od, oi, myd, myi) || (od == myd && oi == myi && (r * k + j) < slot)) ++rank;
    }
  if (myi != PT_NOIDX_U && rank < k) { idx_out[(size_t)t * k + rank] = myi; d2_out[(size_t)t * k + rank] = myd; }
  if (slot < k && slot >= cnt) { idx_out[(size_t)t * k + slot] = PT_NOIDX_U; d2_out[(size_t)t * k + slot] = INFINITY; }
}

// ---- running merge of a streamed source (pt_stream_query): the k best so far (64-bit ids) with the k of the chunk just searched
// (32-bit chunk-local ids + the chunk's first id), both ascending under (d2, id); one thread per target, out-of-place
__global__ __launch_bounds__(WG) void merge_stream_kernel(const unsigned long long* __restrict__ bi, const double* __restrict__ bd,
                                                          const uint32_t* __restrict__ ci, const double* __restrict__ cd, unsigned long long base,
                                                          uint32_t m, int k, unsigned long long* __restrict__ oi, double* __restrict__ od) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  const size_t row = (size_t)t * (size_t)k;
  int a = 0, b = 0;
  for (int o = 0; o < k; ++o) {
    const bool ha = a < k && bi[row + a] != ~0ull, hb = b < k && ci[row + b] != PT_NOIDX_U;
    unsigned long long ia = ~0ull, ib = ~0ull;
    double da = INFINITY, db = INFINITY;
    if (ha) { ia = bi[row + a]; da = bd[row + a]; }
    if (hb) { ib = base + (unsigned long long)ci[row + b]; db = cd[row + b]; }
    const bool take_a = ha && (!hb || da < db || (da == db && ia < ib));
    if (take_a) { oi[row + o] = ia; od[row + o] = da; ++a; }
    else if (hb) { oi[row + o] = ib; od[row + o] = db; ++b; }
    else { oi[row + o] = ~0ull; od[row + o] = INFINITY; }
  }
}

// ---- which other slabs can still hold one of a target's k nearest (reference Distance.h:27-57 on slab boxes)
template <class T>
__global__ __launch_bounds__(WG) void slab_need_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                                                       const double* __restrict__ d2, uint32_t m, int k, int axis,
                                                       const double* __restrict__ bounds, int g, int my_slab, uint8_t* __restrict__ need) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  const double c = (double)(axis == 0 ? x[t] : (axis == 1 ? y[t] : z[t]));
  const double kth = d2[(size_t)t * k + (k - 1)];
  for (int s = 0; s < g; ++s) {
    uint8_t v = 0;
    if (s != my_slab) {
      const double lo = bounds[s], hi = bounds[s + 1];
      const double gapd = c < lo ? lo - c : (c >= hi ? c - hi : 0.0);
      v = (gapd * gapd * (1.0 - 1e-12) <= kth) ? 1 : 0;     // `<=`: an equal-distance lower-index point would win the tie
    }
    need[(size_t)s * m + t] = v;
  }
}

// slab_need + compaction in one pass: the targets that need another slab leave as request packets
// {x, y, z, current k-th d2, bitmask of the slabs asked} (5 doubles) with their row in `sel`; *count is the number written
// (order unspecified: whoever reserves a slot first).  The bitmask is exact in a double for g <= 52.
template <class T>
__global__ __launch_bounds__(WG) void request_pack_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                                                          const double* __restrict__ d2, uint32_t m, int k, int axis,
                                                          const double* __restrict__ bounds, int g, int my_slab, uint32_t* __restrict__ count,
                                                          uint32_t* __restrict__ sel, double* __restrict__ pkt) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  if (t >= m) return;
  const double c = (double)(axis == 0 ? x[t] : (axis == 1 ? y[t] : z[t]));
  const double kth = d2[(size_t)t * k + (k - 1)];
  uint64_t mask = 0;
  for (int s = 0; s < g; ++s) {
    if (s == my_slab) continue;
    const double lo = bounds[s], hi = bounds[s + 1];
    const double gapd = c < lo ? lo - c : (c >= hi ? c - hi : 0.0);
    if (gapd * gapd * (1.0 - 1e-12) <= kth) mask |= 1ull << s;       // same test as slab_need_kernel
  }
  if (!mask) return;
  const uint32_t pos = atomicAdd(count, 1u);                         // (folded into one atomic per wave)
  sel[pos] = t;
  double* o = pkt + (size_t)pos * 5;
  o[0] = (double)x[t]; o[1] = (double)y[t]; o[2] = (double)z[t]; o[3] = kth; o[4] = (double)mask;
}

}  // namespace

template <class T>
void pt_launch_request_pack(const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g,
                            int my_slab, uint32_t* count, uint32_t* sel, double* pkt, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(request_pack_kernel<T>, dim3((m + WG - 1) / WG), dim3(WG), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, my_slab, count, sel, pkt);
}
template void pt_launch_request_pack<float>(const float*, const float*, const float*, const double*, uint32_t, int, int, const double*, int, int,
                                            uint32_t*, uint32_t*, double*, hipStream_t);
template void pt_launch_request_pack<double>(const double*, const double*, const double*, const double*, uint32_t, int, int, const double*, int,
                                             int, uint32_t*, uint32_t*, double*, hipStream_t);

template <class Rec>
void pt_launch_knn(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const Rec* tgt, uint32_t m, int k, const double* bound2,
                   uint32_t* out_idx, double* out_d2, const uint32_t* list, const uint32_t* list_n, hipStream_t s, uint8_t* heavy, uint32_t wave_min) {
  if (!m) return;
  const uint32_t nwg = (uint32_t)(((uint64_t)m * GL + WG - 1) / WG);
  const HierArgs ha{nullptr, nullptr, 0xFFFFFFFFu, heavy, wave_min};
  if (k <= 8)
    hipLaunchKernelGGL((knn_kernel<Rec, 1, false>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2, list, list_n, ha);
  else if (k <= 16)
    hipLaunchKernelGGL((knn_kernel<Rec, 2, false>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2, list, list_n, ha);
  else
    hipLaunchKernelGGL((knn_kernel<Rec, 4, false>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2, list, list_n, ha);
}
template void pt_launch_knn<RecF>(const GridParams&, const RecF*, const uint32_t*, const RecF*, uint32_t, int, const double*, uint32_t*, double*,
                                  const uint32_t*, const uint32_t*, hipStream_t, uint8_t*, uint32_t);
template void pt_launch_knn<RecD>(const GridParams&, const RecD*, const uint32_t*, const RecD*, uint32_t, int, const double*, uint32_t*, double*,
                                  const uint32_t*, const uint32_t*, hipStream_t, uint8_t*, uint32_t);

template <class Rec>
void pt_launch_knn_hier(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const uint32_t* cell_node, const uint32_t* nodes, uint32_t node_thr,
                        const Rec* tgt, uint32_t m, int k, const double* bound2, uint32_t* out_idx, double* out_d2, const uint32_t* list,
                        const uint32_t* list_n, hipStream_t s, uint8_t* heavy, uint32_t wave_min) {
  if (!m) return;
  const uint32_t nwg = (uint32_t)(((uint64_t)m * GL + WG - 1) / WG);
  const HierArgs ha{cell_node, nodes, node_thr, heavy, wave_min};
  if (k <= 8)
    hipLaunchKernelGGL((knn_kernel<Rec, 1, true>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2, list, list_n, ha);
  else if (k <= 16)
    hipLaunchKernelGGL((knn_kernel<Rec, 2, true>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2, list, list_n, ha);
  else
    hipLaunchKernelGGL((knn_kernel<Rec, 4, true>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, m, k, bound2, out_idx, out_d2, list, list_n, ha);
}
template void pt_launch_knn_hier<RecF>(const GridParams&, const RecF*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, const RecF*, uint32_t, int, const double*,
                                       uint32_t*, double*, const uint32_t*, const uint32_t*, hipStream_t, uint8_t*, uint32_t);
template void pt_launch_knn_hier<RecD>(const GridParams&, const RecD*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, const RecD*, uint32_t, int, const double*,
                                       uint32_t*, double*, const uint32_t*, const uint32_t*, hipStream_t, uint8_t*, uint32_t);

// ---- the marks of the group kernel, compacted in order: positions with mark 1 to list1, with mark 2 to list2 ----------------------
constexpr int CP_ITEMS = 8, CP_TILE = WG * CP_ITEMS;
__global__ __launch_bounds__(WG) void mark_count_kernel(const uint8_t* __restrict__ mark, uint32_t m, uint32_t* __restrict__ c1, uint32_t* __restrict__ c2) {
  __shared__ uint32_t wsum[4];
  uint32_t a = 0, b = 0;
#pragma unroll
  for (int i = 0; i < CP_ITEMS; ++i) {
    const uint32_t t = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS + i;
    const uint32_t v = t < m ? mark[t] : 0u;
    a += v == 1u; b += v == 2u;
  }
  uint32_t ta, tb;
  block_excl_scan(a, wsum, ta);
  __syncthreads();
  block_excl_scan(b, wsum, tb);
  if (threadIdx.x == 0) { c1[blockIdx.x] = ta; c2[blockIdx.x] = tb; }
}
__global__ __launch_bounds__(WG) void mark_write_kernel(const uint8_t* __restrict__ mark, uint32_t m, const uint32_t* __restrict__ o1, const uint32_t* __restrict__ o2,
                                                        uint32_t* __restrict__ list1, uint32_t* __restrict__ list2) {
  __shared__ uint32_t wsum[4];
  uint32_t v[CP_ITEMS], a = 0, b = 0;
#pragma unroll
  for (int i = 0; i < CP_ITEMS; ++i) {
    const uint32_t t = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS + i;
    v[i] = t < m ? mark[t] : 0u;
    a += v[i] == 1u; b += v[i] == 2u;
  }
  uint32_t ta, tb;
  uint32_t ea = o1[blockIdx.x] + block_excl_scan(a, wsum, ta);
  __syncthreads();
  uint32_t eb = o2[blockIdx.x] + block_excl_scan(b, wsum, tb);
#pragma unroll
  for (int i = 0; i < CP_ITEMS; ++i) {
    const uint32_t t = blockIdx.x * CP_TILE + threadIdx.x * CP_ITEMS + i;
    if (v[i] == 1u) list1[ea++] = t;
    else if (v[i] == 2u) list2[eb++] = t;
  }
}
template <class Rec>
__global__ __launch_bounds__(WG) void mark_near_kernel(GridParams gp, const Rec* __restrict__ tgt, const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n,
                                                       uint32_t m, const uint8_t* __restrict__ near, uint8_t* __restrict__ mark) {
  const uint32_t i = blockIdx.x * WG + threadIdx.x;
  if (i >= (list ? *list_n : m)) return;
  const uint32_t pos = list ? list[i] : i;
  const Rec r = tgt[pos];
  const double q[3] = {(double)r.x, (double)r.y, (double)r.z};
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) c[a] = (int)fmin(fmax((q[a] - gp.bbmin[a]) * gp.inv_h, 0.0), (double)(gp.dim[a] - 1));     // as the search kernels place the target
  mark[pos] = near[cell_key(gp, c[0], c[1], c[2])] ? 2u : 1u;
}
template <class Rec>
void pt_launch_mark_near(const GridParams& gp, const Rec* tgt, const uint32_t* list, const uint32_t* list_n, uint32_t m, const uint8_t* near, uint8_t* mark, hipStream_t s) {
  if (m) hipLaunchKernelGGL(mark_near_kernel<Rec>, dim3((m + WG - 1) / WG), dim3(WG), 0, s, gp, tgt, list, list_n, m, near, mark);
}
template void pt_launch_mark_near<RecF>(const GridParams&, const RecF*, const uint32_t*, const uint32_t*, uint32_t, const uint8_t*, uint8_t*, hipStream_t);
template void pt_launch_mark_near<RecD>(const GridParams&, const RecD*, const uint32_t*, const uint32_t*, uint32_t, const uint8_t*, uint8_t*, hipStream_t);
// counts -> cnt[nt + 1] each (exclusive offsets, the totals in the last entry); scratch: 2 * (nt + 1) + scan scratch words
void pt_launch_mark_count(const uint8_t* mark, uint32_t m, uint32_t* off1, uint32_t* off2, uint32_t* scan_tmp, hipStream_t s) {
  const uint32_t nt = (m + CP_TILE - 1) / CP_TILE;
  (void)hipMemsetAsync(off1 + nt, 0, 4, s); (void)hipMemsetAsync(off2 + nt, 0, 4, s);
  if (nt) hipLaunchKernelGGL(mark_count_kernel, dim3(nt), dim3(WG), 0, s, mark, m, off1, off2);
  pt_launch_scan_u32(off1, off1, nt + 1, scan_tmp, s);
  pt_launch_scan_u32(off2, off2, nt + 1, scan_tmp, s);
}
void pt_launch_mark_write(const uint8_t* mark, uint32_t m, const uint32_t* off1, const uint32_t* off2, uint32_t* list1, uint32_t* list2, hipStream_t s) {
  const uint32_t nt = (m + CP_TILE - 1) / CP_TILE;
  if (nt) hipLaunchKernelGGL(mark_write_kernel, dim3(nt), dim3(WG), 0, s, mark, m, off1, off2, list1, list2);
}
uint32_t pt_mark_tiles(uint32_t m) { return (m + CP_TILE - 1) / CP_TILE; }

// ---- the blocks that hold targets, as a list (round 4): what the tile kernel is launched over on clouds that leave most of their grid
// empty -- a surface in a fine grid has one block in a dozen occupied, and an empty block's workgroup still costs its launch and two loads
__global__ __launch_bounds__(WG) void tblock_list_kernel(const uint32_t* __restrict__ tblock_start, uint32_t nblocks, uint32_t* __restrict__ list, uint32_t* count) {
  const uint32_t b = blockIdx.x * WG + threadIdx.x;
  const bool has = b < nblocks && tblock_start[b + 1] > tblock_start[b];
  const unsigned long long mask = __ballot(has);
  if (!mask) return;                                        // wave-uniform
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(count, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, 0);
  if (has) list[base + (uint32_t)__popcll(mask & ((1ull << (threadIdx.x & 63)) - 1ull))] = b;      // (block order kept inside a wave: neighbours in the list are neighbours in the grid)
}
void pt_launch_tblock_list(const uint32_t* tblock_start, uint32_t nblocks, uint32_t* list, uint32_t* count, hipStream_t s) {
  (void)hipMemsetAsync(count, 0, 4, s);
  if (nblocks) hipLaunchKernelGGL(tblock_list_kernel, dim3((nblocks + WG - 1) / WG), dim3(WG), 0, s, tblock_start, nblocks, list, count);
}

// wave kernel over a list of `count` target positions (list == nullptr: all m targets); cell_node / nodes may be null (no refined cells)
template <class Rec>
void pt_launch_knn_wave(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const uint32_t* cell_node, const uint32_t* nodes, uint32_t node_thr,
                        const Rec* tgt, uint32_t count, int k, const double* bound2, uint32_t* out_idx, double* out_d2, const uint32_t* list,
                        const uint32_t* list_n, hipStream_t s, const Attr* attr, uint32_t n_attr, int blend_mode, float* rgb_out, float* nrm_out) {
  if (!count) return;
  const WaveBlend wb{attr, n_attr, blend_mode, rgb_out, nrm_out};
  const uint32_t nwg = (((count + 3u) / 4u + 8u * WV_RUN - 1u) / (8u * WV_RUN)) * 8u * WV_RUN;      // whole rounds of 8 XCDs x WV_RUN workgroups (the kernel's mapping)
  const HierArgs ha{cell_node, nodes, node_thr, nullptr, 0u};
  if (nodes) hipLaunchKernelGGL((knn_wave_kernel<Rec, true>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, count, k, bound2, out_idx, out_d2, list, list_n, ha, wb);
  else hipLaunchKernelGGL((knn_wave_kernel<Rec, false>), dim3(nwg), dim3(WG), 0, s, gp, src, cell_start, tgt, count, k, bound2, out_idx, out_d2, list, list_n, ha, wb);
}
template void pt_launch_knn_wave<RecF>(const GridParams&, const RecF*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, const RecF*, uint32_t, int, const double*,
                                       uint32_t*, double*, const uint32_t*, const uint32_t*, hipStream_t, const Attr*, uint32_t, int, float*, float*);
template void pt_launch_knn_wave<RecD>(const GridParams&, const RecD*, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, const RecD*, uint32_t, int, const double*,
                                       uint32_t*, double*, const uint32_t*, const uint32_t*, hipStream_t, const Attr*, uint32_t, int, float*, float*);

// tile kernel over all blocks; targets it cannot settle are appended to todo[*todo_n] (todo_n zeroed by the caller).
// geometry 1 = the two-workgroups-per-CU geometry (regions of <= PT_TILE_CAP_SMALL_* records), 0 = large.  With `attr` the
// neighbours' attributes are blended in the same pass (rgb_out / nrm_out rows of the settled targets only).  `retry`:
// blocks over the small budget but within the large one are listed there instead of going to `todo`; `blocks`: run over
// such a list (nblocks_listed entries) instead of every block.  src_exact / tgt_exact (fp64 clouds): `src` is then the fp32
// shadow of the sorted records and `tgt` is unused.
void pt_launch_knn_tile(const GridParams& gp, const RecF* src, const uint32_t* cell_start, const RecF* tgt, const uint32_t* tblock_start, int k,
                        uint32_t* out_idx, double* out_d2, uint32_t* todo, uint32_t* todo_n, int geometry, const Attr* attr, uint32_t n_attr, int mode,
                        float* rgb_out, float* nrm_out, const uint32_t* blocks, uint32_t nblocks_listed, uint32_t* retry, uint32_t* retry_n,
                        const RecD* src_exact, const RecD* tgt_exact, float e_src, hipStream_t s, const double* bound) {
  const uint32_t nb = blocks ? nblocks_listed : (uint32_t)gp.nblocks;
  if (!nb) return;
  const TileBlend bl{attr, n_attr, mode, rgb_out, nrm_out};
  const TileBlocks tbk{blocks, retry, retry_n, (uint32_t)PT_TILE_CAP_LARGE, bound};
  if (bound) {                                       // bounded variant: no fused blend (pt_api.hip only sends those here); round 4: fp64 clouds and k in 25..32 too
#define PT_TILE_LAUNCHB1(KK, CAP, TH, WD, DB, KCH)                                                                                                     \
  hipLaunchKernelGGL((knn_tile_kernel<KK, CAP, TH, WD, false, DB, KCH, true>), dim3(nb), dim3(TH), 0, s, gp, src, cell_start, tgt, tblock_start, k, \
                     out_idx, out_d2, todo, todo_n, bl, tbk, TileDouble{src_exact, tgt_exact, e_src})
#define PT_TILE_LAUNCHB(KK, CAP, TH, KCH)                             \
  do {                                                                \
    if (src_exact) PT_TILE_LAUNCHB1(KK, CAP, TH, false, true, KCH);   \
    else PT_TILE_LAUNCHB1(KK, CAP, TH, false, false, KCH);            \
  } while (0)
    if (k > 24) {
      if (src_exact) PT_TILE_LAUNCHB1(32, PT_TILE_CAP_WIDE, 512, true, true, 32);
      else PT_TILE_LAUNCHB1(32, PT_TILE_CAP_WIDE, 512, true, false, 32);
    } else if (geometry == 1 && k <= 16) {
      if (k <= 8) PT_TILE_LAUNCHB(8, PT_TILE_CAP_SMALL_8, 512, 8);
      else PT_TILE_LAUNCHB(16, PT_TILE_CAP_SMALL_16, 512, 16);
    } else {
      if (k <= 8) PT_TILE_LAUNCHB(8, PT_TILE_CAP_LARGE, 768, 8);
      else if (k <= 16) PT_TILE_LAUNCHB(16, PT_TILE_CAP_LARGE, 768, 16);
      else if (k <= 20) PT_TILE_LAUNCHB(32, PT_TILE_CAP_LARGE, 768, 20);
      else PT_TILE_LAUNCHB(32, PT_TILE_CAP_LARGE, 768, 24);
    }
#undef PT_TILE_LAUNCHB
#undef PT_TILE_LAUNCHB1
    return;
  }
  const TileDouble dd{src_exact, tgt_exact, e_src};
#define PT_TILE_LAUNCH1(KK, CAP, TH, WD, BL, DB, KCH)                                                                                             \
  hipLaunchKernelGGL((knn_tile_kernel<KK, CAP, TH, WD, BL, DB, KCH>), dim3(nb), dim3(TH), 0, s, gp, src, cell_start, tgt, tblock_start, k, out_idx, \
                     out_d2, todo, todo_n, bl, tbk, dd)
#define PT_TILE_LAUNCHC(KK, CAP, TH, WD, KCH)                             \
  do {                                                                    \
    if (src_exact) {                                                      \
      if (attr) PT_TILE_LAUNCH1(KK, CAP, TH, WD, true, true, KCH);        \
      else PT_TILE_LAUNCH1(KK, CAP, TH, WD, false, true, KCH);            \
    } else {                                                              \
      if (attr) PT_TILE_LAUNCH1(KK, CAP, TH, WD, true, false, KCH);       \
      else PT_TILE_LAUNCH1(KK, CAP, TH, WD, false, false, KCH);           \
    }                                                                     \
  } while (0)
#define PT_TILE_LAUNCH(KK, CAP, TH, WD) PT_TILE_LAUNCHC(KK, CAP, TH, WD, KK)
  const int small = geometry == 1;
  if (k > 24) PT_TILE_LAUNCH(32, PT_TILE_CAP_WIDE, 512, true);      // wide queue, 512 threads, one workgroup per CU
  else if (geometry == 4 && k > 16) {
    // MEDIUM (round 4): the K = 32 body on 384 threads and a 3888-record region -- 62 KB of LDS, six waves of <= 170 VGPRs: TWO workgroups
    // per CU where the 768-thread geometry has one.  For clouds whose regions are small because most of their cells are empty (surfaces)
    if (k <= 20) PT_TILE_LAUNCHC(32, PT_TILE_CAP_SMALL_16, 384, false, 20);
    else PT_TILE_LAUNCHC(32, PT_TILE_CAP_SMALL_16, 384, false, 24);
  }
  else if (small && k <= 16) {       // (K = 32 needs more registers than two workgroups per CU leave: large geometry only)
    if (k <= 8) PT_TILE_LAUNCH(8, PT_TILE_CAP_SMALL_8, 512, false);
    else PT_TILE_LAUNCH(16, PT_TILE_CAP_SMALL_16, 512, false);
  } else {
    if (k <= 8) PT_TILE_LAUNCH(8, PT_TILE_CAP_LARGE, 768, false);
    else if (k <= 16) PT_TILE_LAUNCH(16, PT_TILE_CAP_LARGE, 768, false);
    else if (k <= 20) PT_TILE_LAUNCHC(32, PT_TILE_CAP_LARGE, 768, false, 20);       // the reference's K = 20 (src/pointsTransfer.cpp:128): a chain of exactly 20
    // (round 3: the same body on 1024 threads -- 16 waves per CU, 128 VGPRs with 56 bytes of spills, 7680-record region -- measured 6.77 ms
    //  against 6.74 at 100M / 10M: more waves of one workgroup do not shorten its latency chain)
    else if (k <= 24) PT_TILE_LAUNCHC(32, PT_TILE_CAP_LARGE, 768, false, 24);
    else PT_TILE_LAUNCH(32, PT_TILE_CAP_LARGE, 768, false);
  }
#undef PT_TILE_LAUNCH
#undef PT_TILE_LAUNCHC
#undef PT_TILE_LAUNCH1
}

// pt_stream_query, once per chunk and sweep: the bound every target brings to this chunk's search, and how many targets bring one that
// reaches the chunk's bounding box at all (box distance: Distance::min_distance_to_rectangle of the reference, src/Distance.h:27-57;
// `<=` because an equal distance could still enter the list).
//   forward sweep, chunk c:   a target searched in an earlier chunk (first[t] < c) brings its current k-th squared distance (+inf while
//                             its list is short); one that has not been searched yet and lies INSIDE the chunk's box -- or within
//                             `margin` of it, a few point spacings: the distance over which its neighbours may well be in this chunk and
//                             an unbounded search from outside costs a ring or two -- is searched unbounded from now on (first[t] = c);
//                             one that lies farther outside and has no list yet is DEFERRED (-1: the
//                             kernels return an empty list for it) -- searching it from outside, unbounded, is the slow path of every
//                             kernel, and the chunk that holds its neighbourhood is still to come;
//   backward sweep, chunk c:  exactly the deferred pairs, first[t] > c, now with a bound.  Every (target, chunk) pair is searched once,
//                             under a bound that is an upper bound of the target's final k-th distance, so the merged lists are the
//                             resident search's.
template <class T>
__global__ __launch_bounds__(WG) void stream_sweep_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ z,
                                                          const unsigned long long* __restrict__ bi, const double* __restrict__ bd, uint32_t m, int k,
                                                          uint32_t c, int backward, uint32_t* __restrict__ first, double lx, double ly, double lz, double hx,
                                                          double hy, double hz, double margin2, double* __restrict__ bound, uint32_t* count) {
  const uint32_t t = blockIdx.x * WG + threadIdx.x;
  bool reach = false;
  if (t < m) {
    const double q[3] = {(double)x[t], (double)y[t], (double)z[t]}, lo[3] = {lx, ly, lz}, hi[3] = {hx, hy, hz};
    double d = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { const double g = q[a] < lo[a] ? lo[a] - q[a] : (q[a] > hi[a] ? q[a] - hi[a] : 0.0); d += g * g; }
    const size_t last = (size_t)t * (size_t)k + (size_t)(k - 1);
    const double kth = bi[last] != ~0ull ? bd[last] : INFINITY;
    const uint32_t f = first[t];
    double b;
    if (d != d) b = -1.0;                                       // a NaN coordinate: no neighbours anywhere, the row stays empty
    else if (backward) b = f > c ? kth : -1.0;
    else if (f < c) b = kth;
    else if (d <= margin2) { b = INFINITY; first[t] = c; }      // inside the box, or within a few point spacings of it
    else b = -1.0;
    bound[t] = b;
    reach = b >= 0.0 && !(d > b);
  }
  const uint32_t n = (uint32_t)__popcll(__ballot(reach));
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(count, n);
}
template <class T>
void pt_launch_stream_sweep(const T* xyz_planar, const unsigned long long* best_idx, const double* best_d2, uint32_t m, int k, uint32_t chunk, int backward,
                            uint32_t* first, const double lo[3], const double hi[3], double margin, double* bound, uint32_t* count, hipStream_t s) {
  if (m) hipLaunchKernelGGL(stream_sweep_kernel<T>, dim3((m + WG - 1) / WG), dim3(WG), 0, s, xyz_planar, xyz_planar + m, xyz_planar + 2 * (size_t)m, best_idx, best_d2,
                            m, k, chunk, backward, first, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], margin * margin, bound, count);
}
template void pt_launch_stream_sweep<float>(const float*, const unsigned long long*, const double*, uint32_t, int, uint32_t, int, uint32_t*, const double*, const double*,
                                            double, double*, uint32_t*, hipStream_t);
template void pt_launch_stream_sweep<double>(const double*, const unsigned long long*, const double*, uint32_t, int, uint32_t, int, uint32_t*, const double*, const double*,
                                             double, double*, uint32_t*, hipStream_t);
void pt_launch_merge_stream(const unsigned long long* best_idx, const double* best_d2, const uint32_t* chunk_idx, const double* chunk_d2,
                           unsigned long long base, uint32_t m, int k, unsigned long long* out_idx, double* out_d2, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(merge_stream_kernel, dim3((m + WG - 1) / WG), dim3(WG), 0, s, best_idx, best_d2, chunk_idx, chunk_d2, base, m, k, out_idx, out_d2);
}

void pt_launch_merge(const uint32_t* idx_lists, const double* d2_lists, int g, uint32_t m, int k, uint32_t* idx_out, double* d2_out,
                     hipStream_t s) {
  if (!m) return;
  const uint64_t threads = (uint64_t)m * (uint64_t)(g * k);
  hipLaunchKernelGGL(merge_kernel, dim3((uint32_t)((threads + WG - 1) / WG)), dim3(WG), 0, s, idx_lists, d2_lists, g, m, k, idx_out, d2_out);
}

template <class T>
void pt_launch_slab_need(const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g,
                         int my_slab, uint8_t* need, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(slab_need_kernel<T>, dim3((m + WG - 1) / WG), dim3(WG), 0, s, x, y, z, d2, m, k, axis, bounds_dev, g, my_slab, need);
}
template void pt_launch_slab_need<float>(const float*, const float*, const float*, const double*, uint32_t, int, int, const double*, int, int,
                                         uint8_t*, hipStream_t);
template void pt_launch_slab_need<double>(const double*, const double*, const double*, const double*, uint32_t, int, int, const double*, int,
                                          int, uint8_t*, hipStream_t);
