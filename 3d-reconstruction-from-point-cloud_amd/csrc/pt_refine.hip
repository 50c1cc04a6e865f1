// pt_refine.hip -- adaptive refinement of the cell grid for clouds with strong density contrast (gfx950 / MI355X).
//
// The reference's kd-tree (CGAL Kd_tree behind `Tree tree(points.begin(), points.end())`, src/pointsTransfer.cpp:259) adapts to
// the local density by construction; a dense uniform grid does not: in a cloud of surfaces and clusters (BASELINE config 5)
// single cells hold 10^3..10^5 points while most are empty, and every query near them scans whole cells.  This file puts an
// 8 x 8 x 8 sub-grid inside every HEAVY cell (more than `threshold` points), recursively (up to PT_REFINE_DEPTH levels):
//   * the records of a heavy cell are counting-sorted by sub-cell IN PLACE inside the cell's own range of the sorted array --
//     the level-0 table (cell_start) stays valid, so the tile kernel and every level-0 scan work unchanged;
//   * every refined cell ("node") gets a 513-entry table of sub-cell starts and a 512-entry table of child nodes;
//   * membership: sub = clamp(floor((u - node origin) * cells-per-unit), 0, 7) per axis with u = (p - bbmin) / h, the position
//     in level-0 cell units; origins and scales are dyadic rationals, exact in fp64, and the query computes the very same
//     expression, so build and search agree on every point.
// The search over this structure (HierScan, used by the group kernel knn_kernel<.., HIER = true>) is in pt_query.hip.
#include "pt_internal.h"

namespace {

constexpr int RW = 512;            // refine workgroup: one thread per sub-cell in the scan
constexpr int RITEMS = 12;         // records a thread keeps in registers: nodes up to 6144 points are read once

__device__ inline void node_header(uint32_t* node, double ox, double oy, double oz, double inv, uint32_t s, uint32_t e) {
  double* h = reinterpret_cast<double*>(node);
  h[0] = ox; h[1] = oy; h[2] = oz; h[3] = inv; h[4] = 1.0 / inv;      // (a power of 1/8: exact)
  node[PT_NODE_START] = s;                 // provisional: the refine kernel overwrites the start table
  node[PT_NODE_START + 512] = e;
}

// level 0: every cell with more than `threshold` points becomes a node
// (near, optional, one zeroed byte per cell: the 27 cells around every node are flagged -- a target whose own cell carries the flag
//  has a refined cell among its 27 nearest, which is all the search needs to know to pick the kernel variant for it)
__global__ __launch_bounds__(256) void heavy_cells_kernel(GridParams gp, const uint32_t* __restrict__ cs, uint32_t ncells, uint32_t threshold,
                                                          uint32_t* __restrict__ cell_node, uint32_t* __restrict__ node_count, uint32_t node_cap,
                                                          uint32_t* __restrict__ nodes, uint8_t* __restrict__ near) {
  const uint32_t c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncells) return;
  const uint32_t s = cs[c], e = cs[c + 1];
  uint32_t id = 0;
  if (e - s > threshold) {
    const uint32_t slot = atomicAdd(node_count, 1u);
    if (slot < node_cap) {
      int cx, cy, cz;
      pt_decode_cell(gp, c, cx, cy, cz);
      node_header(nodes + (size_t)slot * PT_NODE_WORDS, (double)cx, (double)cy, (double)cz, 8.0, s, e);
      id = slot + 1;
      if (near) {
        for (int dz = -1; dz <= 1; ++dz)
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int x = cx + dx, y = cy + dy, z = cz + dz;
              if (x >= 0 && x < gp.dim[0] && y >= 0 && y < gp.dim[1] && z >= 0 && z < gp.dim[2])
                near[(pt_block_id(gp.mdim, x, y, z) << 9) + pt_local_cell(x, y, z)] = 1;
            }
      }
    }
  }
  cell_node[c] = id;
}

// deeper levels: heavy sub-cells of the nodes [n0, n1) become nodes themselves
__global__ __launch_bounds__(RW) void heavy_subcells_kernel(uint32_t n0, uint32_t threshold, uint32_t* __restrict__ node_count, uint32_t node_cap,
                                                            uint32_t* __restrict__ nodes) {
  uint32_t* N = nodes + (size_t)(n0 + blockIdx.x) * PT_NODE_WORDS;
  const uint32_t sub = threadIdx.x;
  const uint32_t s = N[PT_NODE_START + sub], e = N[PT_NODE_START + sub + 1];
  uint32_t id = 0;
  if (e - s > threshold) {
    const uint32_t slot = atomicAdd(node_count, 1u);
    if (slot < node_cap) {
      const double* h = reinterpret_cast<const double*>(N);
      const double inv = h[3], w = h[4];
      node_header(nodes + (size_t)slot * PT_NODE_WORDS, h[0] + (double)(sub & 7u) * w, h[1] + (double)((sub >> 3) & 7u) * w,
                  h[2] + (double)(sub >> 6) * w, inv * 8.0, s, e);
      id = slot + 1;
    }
  }
  N[PT_NODE_CHILD + sub] = id;
}

template <class Rec>
__device__ inline uint32_t subcell_of(const GridParams& gp, const Rec& r, double ox, double oy, double oz, double inv) {
  // raw (unclamped) position in level-0 cell units, as pt_cell_axis computes it before clamping; the sub-cell index is clamped
  const double ux = ((double)r.x - gp.bbmin[0]) * gp.inv_h, uy = ((double)r.y - gp.bbmin[1]) * gp.inv_h, uz = ((double)r.z - gp.bbmin[2]) * gp.inv_h;
  const int sx = (int)fmin(fmax(floor((ux - ox) * inv), 0.0), 7.0), sy = (int)fmin(fmax(floor((uy - oy) * inv), 0.0), 7.0),
            sz = (int)fmin(fmax(floor((uz - oz) * inv), 0.0), 7.0);
  return (uint32_t)((sz << 6) | (sy << 3) | sx);
}

// one workgroup per node: counting sort of the node's records by sub-cell, in place; writes the node's start table
template <class Rec>
__global__ __launch_bounds__(RW) void refine_nodes_kernel(GridParams gp, Rec* __restrict__ rec, Rec* __restrict__ tmp, uint32_t n0, uint32_t* __restrict__ nodes) {
  __shared__ uint32_t cnt[512];
  __shared__ uint32_t wsum[RW / 64];
  uint32_t* N = nodes + (size_t)(n0 + blockIdx.x) * PT_NODE_WORDS;
  const double* h = reinterpret_cast<const double*>(N);
  const double ox = h[0], oy = h[1], oz = h[2], inv = h[3];
  const uint32_t s = N[PT_NODE_START], e = N[PT_NODE_START + 512];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const bool in_regs = (e - s) <= (uint32_t)(RW * RITEMS);
  Rec r[RITEMS];
  uint32_t sc[RITEMS];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < RITEMS; ++j) {
      const uint32_t i = s + j * RW + threadIdx.x;
      if (i < e) { r[j] = rec[i]; sc[j] = subcell_of(gp, r[j], ox, oy, oz, inv); atomicAdd(&cnt[sc[j]], 1u); }
    }
  } else {
    for (uint32_t i = s + threadIdx.x; i < e; i += RW) atomicAdd(&cnt[subcell_of(gp, rec[i], ox, oy, oz, inv)], 1u);
  }
  __syncthreads();
  const uint32_t c0 = cnt[threadIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(c0);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t off = 0;
#pragma unroll
  for (int i = 0; i < RW / 64; ++i) if (i < w) off += wsum[i];
  const uint32_t ex = off + incl - c0;
  __syncthreads();
  cnt[threadIdx.x] = ex;                                    // cursor, relative to s
  N[PT_NODE_START + threadIdx.x] = s + ex;                  // (entry 512 = e is already there)
  {                                                         // which rows of eight sub-cells hold anything: what a search looks at first
    const unsigned long long nz = __ballot(c0 != 0u);       // wave w covers sub-cells 64 w .. 64 w + 63 = rows 8 w .. 8 w + 7
    if (lane == 0) {
      uint32_t rows8 = 0;
#pragma unroll
      for (int r = 0; r < 8; ++r) rows8 |= ((nz >> (8 * r)) & 0xFFull) ? (1u << r) : 0u;
      reinterpret_cast<unsigned char*>(N + PT_NODE_ROWMASK)[w] = (unsigned char)rows8;
    }
  }
  __syncthreads();
  if (in_regs) {                                            // everything is in registers: the range can be rewritten in place
#pragma unroll
    for (int j = 0; j < RITEMS; ++j) {
      const uint32_t i = s + j * RW + threadIdx.x;
      if (i < e) rec[s + atomicAdd(&cnt[sc[j]], 1u)] = r[j];
    }
  } else {                                                  // big node: through the scratch array, then back (one workgroup owns the range)
    for (uint32_t i = s + threadIdx.x; i < e; i += RW) {
      const Rec v = rec[i];
      tmp[s + atomicAdd(&cnt[subcell_of(gp, v, ox, oy, oz, inv)], 1u)] = v;
    }
    __threadfence();
    __syncthreads();
    for (uint32_t i = s + threadIdx.x; i < e; i += RW) rec[i] = tmp[i];
  }
}

// ---- runs of identical points (pt_common.h, PT_DUP_KEEP) -------------------------------------------------------------------------
// One workgroup per node, one WAVE per leaf sub-cell with more than PT_DUP_KEEP points (sub-cells w, w + 4, ... for wave w).  The wave
// streams the leaf once: are all positions the first one's?  which are the 64 lowest indices? -- a sorted list, one entry per lane,
// merged with the 64 indices of a step by a bitonic sort-merge only when one of them beats the current PT_DUP_KEEP-th (after the first
// steps almost none does: the expected number of merges is ~ keep * ln(L / keep) / 64).  A pure leaf is then rewritten through the
// scratch array: the keep lowest indices in ascending order, then everybody else; its child link gets the tag.
__device__ inline uint32_t bitonic_sort64_u32(uint32_t v, int lane) {           // ascending across the 64 lanes
#pragma unroll
  for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j >= 1; j >>= 1) {
      const uint32_t p = (uint32_t)__shfl_xor((int)v, j);
      const bool keep_min = ((lane & j) == 0) == ((lane & k2) == 0);
      v = keep_min ? min(v, p) : max(v, p);
    }
  }
  return v;
}
template <class Rec>
__global__ __launch_bounds__(256) void dedup_leaves_kernel(Rec* __restrict__ rec, Rec* __restrict__ tmp, uint32_t n0, uint32_t* __restrict__ nodes) {
  uint32_t* N = nodes + (size_t)(n0 + blockIdx.x) * PT_NODE_WORDS;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Every loop of this kernel has a wave-uniform trip count and no `continue` / `break`: the body is full of cross-lane operations, and
  // a first version that handed the leaves out through an LDS counter (`for (;;) { ... if (small) continue; ... }`) was compiled into a
  // loop whose back-edge re-read the counter's value from a lane that had not fetched it -- sub-cell 0 for ever.
  for (uint32_t it = 0; it < 128u; ++it) {
    const uint32_t sub = it * 4u + wave;                    // sub-cells wave, wave + 4, ...
    const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)N[PT_NODE_START + sub]), e = (uint32_t)__builtin_amdgcn_readfirstlane((int)N[PT_NODE_START + sub + 1]);
    const uint32_t link = (uint32_t)__builtin_amdgcn_readfirstlane((int)N[PT_NODE_CHILD + sub]);
    if (e - s > (uint32_t)PT_DUP_KEEP && link == 0u) {      // (link != 0: a node of the next level -- its own leaves are looked at there)
      const Rec p0 = rec[s];
      bool same = true;
      uint32_t list = 0xFFFFFFFFu;                          // lane i: the i-th lowest index so far
      for (uint32_t base = s; base < e; base += 64u) {
        const uint32_t i = base + (uint32_t)lane;
        uint32_t c = 0xFFFFFFFFu;
        if (i < e) { const Rec r = rec[i]; same = same && r.x == p0.x && r.y == p0.y && r.z == p0.z; c = r.id; }
        const uint32_t lim = (uint32_t)__shfl((int)list, PT_DUP_KEEP - 1);
        if (__ballot(c < lim) != 0ull) {                    // wave-uniform: somebody beats the keep-th lowest
          c = bitonic_sort64_u32(c, lane);
          const uint32_t rv = (uint32_t)__shfl((int)c, 63 - lane);
          list = min(list, rv);                             // the 64 lowest of list and candidates: a bitonic sequence
#pragma unroll
          for (int j = 32; j >= 1; j >>= 1) {
            const uint32_t p = (uint32_t)__shfl_xor((int)list, j);
            list = ((lane & j) == 0) ? min(list, p) : max(list, p);
          }
        }
      }
      if (__ballot(!same) == 0ull) {                        // one position only (several: the leaf is left as it is)
        const uint32_t pivot = (uint32_t)__shfl((int)list, PT_DUP_KEEP - 1);        // the keep-th lowest index (indices are distinct)
        if (lane < PT_DUP_KEEP) { Rec o = p0; o.id = list; tmp[s + (uint32_t)lane] = o; }
        uint32_t w = s + (uint32_t)PT_DUP_KEEP;
        for (uint32_t base = s; base < e; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          Rec r = p0;
          bool rest = false;
          if (i < e) { r = rec[i]; rest = r.id > pivot; }
          const unsigned long long m = __ballot(rest);
          if (rest) tmp[w + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = r;
          w += (uint32_t)__popcll(m);
        }
        __threadfence();                                    // (tmp is read back by other lanes of this wave below)
        for (uint32_t base = s; base < e; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          if (i < e) rec[i] = tmp[i];
        }
        if (lane == 0) N[PT_NODE_CHILD + sub] = PT_LEAF_TRUNC | (uint32_t)PT_DUP_KEEP;
      }
    }
  }
}

// how many leaves carry the tag, and how many points sit behind their fronts (pt_stats)
__global__ __launch_bounds__(RW) void count_tagged_leaves_kernel(uint32_t n0, const uint32_t* __restrict__ nodes, uint32_t* __restrict__ stats) {
  const uint32_t* N = nodes + (size_t)(n0 + blockIdx.x) * PT_NODE_WORDS;
  const uint32_t link = N[PT_NODE_CHILD + threadIdx.x];
  const bool tagged = (link & PT_LEAF_TRUNC) != 0u;
  uint32_t behind = tagged ? N[PT_NODE_START + threadIdx.x + 1] - N[PT_NODE_START + threadIdx.x] - (link & ~PT_LEAF_TRUNC) : 0u;
  const uint32_t cnt = (uint32_t)__popcll(__ballot(tagged));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) behind += __shfl_xor(behind, o);
  if ((threadIdx.x & 63) == 0 && cnt) { atomicAdd(&stats[0], cnt); atomicAdd(&stats[1], behind); }
}

// fp64 clouds: the fp32 shadow of the sorted records (id = sorted position) after refinement has moved records inside their cells
__global__ __launch_bounds__(256) void reshadow_kernel(const RecD* __restrict__ rec, uint32_t n, RecF* __restrict__ shadow) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const RecD v = rec[i];
  RecF o;
  o.x = (float)v.x; o.y = (float)v.y; o.z = (float)v.z; o.id = i;
  shadow[i] = o;
}

}  // namespace

void pt_launch_heavy_cells(const GridParams& gp, const uint32_t* cs, uint32_t ncells, uint32_t threshold, uint32_t* cell_node, uint32_t* node_count,
                           uint32_t node_cap, uint32_t* nodes, uint8_t* near, hipStream_t s) {
  if (!ncells) return;
  hipLaunchKernelGGL(heavy_cells_kernel, dim3((ncells + 255) / 256), dim3(256), 0, s, gp, cs, ncells, threshold, cell_node, node_count, node_cap, nodes, near);
}
void pt_launch_heavy_subcells(uint32_t n0, uint32_t n1, uint32_t threshold, uint32_t* node_count, uint32_t node_cap, uint32_t* nodes, hipStream_t s) {
  if (n1 <= n0) return;
  hipLaunchKernelGGL(heavy_subcells_kernel, dim3(n1 - n0), dim3(RW), 0, s, n0, threshold, node_count, node_cap, nodes);
}
template <class Rec>
void pt_launch_refine_nodes(const GridParams& gp, Rec* rec, Rec* tmp, uint32_t n0, uint32_t n1, uint32_t* nodes, hipStream_t s) {
  if (n1 <= n0) return;
  hipLaunchKernelGGL(refine_nodes_kernel<Rec>, dim3(n1 - n0), dim3(RW), 0, s, gp, rec, tmp, n0, nodes);
}
template void pt_launch_refine_nodes<RecF>(const GridParams&, RecF*, RecF*, uint32_t, uint32_t, uint32_t*, hipStream_t);
template void pt_launch_refine_nodes<RecD>(const GridParams&, RecD*, RecD*, uint32_t, uint32_t, uint32_t*, hipStream_t);
template <class Rec>
void pt_launch_dedup_leaves(Rec* rec, Rec* tmp, uint32_t n0, uint32_t n1, uint32_t* nodes, uint32_t* stats2, hipStream_t s) {
  if (n1 <= n0) return;
  hipLaunchKernelGGL(dedup_leaves_kernel<Rec>, dim3(n1 - n0), dim3(256), 0, s, rec, tmp, n0, nodes);
  if (stats2) hipLaunchKernelGGL(count_tagged_leaves_kernel, dim3(n1 - n0), dim3(RW), 0, s, n0, nodes, stats2);
}
template void pt_launch_dedup_leaves<RecF>(RecF*, RecF*, uint32_t, uint32_t, uint32_t*, uint32_t*, hipStream_t);
template void pt_launch_dedup_leaves<RecD>(RecD*, RecD*, uint32_t, uint32_t, uint32_t*, uint32_t*, hipStream_t);
void pt_launch_reshadow(const RecD* rec, uint32_t n, RecF* shadow, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(reshadow_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rec, n, shadow);
}
