// pt_common.h -- types shared by the gfx950 kernels and the C-ABI glue (libpt_hip.so).
//
// Data layout in HBM:
//   input cloud      planar xyz  x[n] y[n] z[n]      (f32, or f64 for reference AoS input)
//   attribute table  16 B / point {rgba8, nx, ny, nz} indexed by ORIGINAL index (one gather / neighbour)
//   sorted records   16 B {x,y,z,id} (f32) or 32 B (f64): one dwordx4 / 2x dwordx4 load per lane
//   cell_start       u32[ncells+1]
//   cell key         = macro (row-major over 64^3-cell macro blocks) << 18
//                    | Morton3(block within macro, 3 bits / axis)     <<  9
//                    | row-major cell within the 8x8x8 block (lz<<6 | ly<<3 | lx)
//                    Morton order keeps spatially close 8^3-cell blocks close in HBM / L2; row-major
//                    inside a block keeps a run of x-adjacent cells one contiguous 128-B-granular
//                    range, which is what the query's 8-lane groups load coalesced.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_NOIDX_U 0xFFFFFFFFu
#define PT_BLOCK_CELLS 512          // 8 x 8 x 8 cells per block
#define PT_MACRO_BLOCKS 512         // 8 x 8 x 8 blocks per macro block
#define PT_MAXBINS 1024             // LDS histogram bins per partition pass (macro blocks, or blocks when <= 1024)
#define PT_MAX_MACROS 8192          // macro blocks of the largest grid: beyond PT_MAXBINS of them the sort takes a pass more (groups of macro blocks first)
#define PT_CELL_EPS 1e-9            // slack (in cell units) on every cell-box bound: cell membership is
                                    // computed in fp64 with ~1e-12 cell units of rounding at most

// refined ("heavy") cells, pt_refine.hip: one node = header {origin x, y, z in level-0 cell units, sub-cells per cell unit, sub-cell
// side in cell units} as five doubles, the 64-bit mask of the non-empty rows (bit sz * 8 + sy) as two words, 513 absolute starts of its
// 8 x 8 x 8 sub-cells (sub = sz << 6 | sy << 3 | sx), 512 child node ids (id + 1, 0 = leaf)
#define PT_NODE_ROWMASK 10
#define PT_NODE_START 12
#define PT_NODE_CHILD (PT_NODE_START + 513)
#define PT_NODE_WORDS 1040
#define PT_REFINE_DEPTH 3           // levels below the grid: sub-cells of 1/8, 1/64, 1/512 of a cell side
// Runs of IDENTICAL points (round 4): a leaf sub-cell of a node whose points all share one position -- what a quantised cloud (fp16
// coordinates, scanner lattices) leaves in its dense places: clumps of up to 10^5 duplicates -- keeps its PT_DUP_KEEP lowest original
// indices, sorted, at the front of its range, and the leaf's child link says so: PT_LEAF_TRUNC | length of that front.  Under the
// total order (d2, index) no other point of the leaf can be among the k <= PT_DUP_KEEP nearest of any target, so a search that reads
// the front only is exact; the leaf's range still holds every one of its points (a flat scan of the cell stays valid).
#define PT_DUP_KEEP 32
#define PT_LEAF_TRUNC 0x80000000u

struct RecF { float x, y, z; uint32_t id; };                     // 16 B
struct RecD { double x, y, z; uint32_t id; uint32_t pad; };      // 32 B
struct Attr { uint32_t rgba; float nx, ny, nz; };                // 16 B
#if defined(__HIPCC__)
// one attribute record of a random point: a 16-byte non-temporal load (the table is gathered once per neighbour with no
// reuse worth a cache line: measured 9.6 vs 10.2 ms for the 400 M gathers of C4's blend)
__device__ inline Attr pt_gather_attr(const Attr* __restrict__ table, uint32_t id) {
  typedef uint32_t pt_u4 __attribute__((ext_vector_type(4)));
  const pt_u4 v = __builtin_nontemporal_load(reinterpret_cast<const pt_u4*>(table) + id);
  Attr a;
  a.rgba = v.x; a.nx = __uint_as_float(v.y); a.ny = __uint_as_float(v.z); a.nz = __uint_as_float(v.w);
  return a;
}
#endif

struct GridParams {
  double bbmin[3];
  double inv_h, h;
  int dim[3];        // occupied cells per axis (points are clamped into [0, dim))
  int mdim[3];       // macro blocks (64 cells) per axis = ceil(dim / 64)
  int nblocks;       // mdim[0]*mdim[1]*mdim[2]*512   (padded)
};

// ---- Morton code of 3-bit block coordinates inside a macro block ----------------------------
__host__ __device__ inline uint32_t pt_spread3(uint32_t v) {   // 3 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4);
}
__host__ __device__ inline uint32_t pt_morton9(uint32_t bx, uint32_t by, uint32_t bz) {
  return pt_spread3(bx) | (pt_spread3(by) << 1) | (pt_spread3(bz) << 2);
}
// dense block id of the block holding cell (cx,cy,cz)
__host__ __device__ inline uint32_t pt_block_id(const int* mdim, int cx, int cy, int cz) {
  const uint32_t macro = ((uint32_t)(cz >> 6) * (uint32_t)mdim[1] + (uint32_t)(cy >> 6)) * (uint32_t)mdim[0] + (uint32_t)(cx >> 6);
  return (macro << 9) | pt_morton9((uint32_t)(cx >> 3) & 7u, (uint32_t)(cy >> 3) & 7u, (uint32_t)(cz >> 3) & 7u);
}
// cell key -> cell coordinates (inverse of pt_block_id << 9 | pt_local_cell)
__host__ __device__ inline void pt_decode_cell(const GridParams& gp, uint32_t key, int& cx, int& cy, int& cz) {
  const uint32_t blk = key >> 9, local = key & 511u, macro = blk >> 9, m9 = blk & 511u;
  const int bx = (int)(macro % (uint32_t)gp.mdim[0]) * 8 + (int)((m9 & 1u) | ((m9 >> 2) & 2u) | ((m9 >> 4) & 4u));
  const int by = (int)((macro / (uint32_t)gp.mdim[0]) % (uint32_t)gp.mdim[1]) * 8 + (int)(((m9 >> 1) & 1u) | ((m9 >> 3) & 2u) | ((m9 >> 5) & 4u));
  const int bz = (int)(macro / (uint32_t)(gp.mdim[0] * gp.mdim[1])) * 8 + (int)(((m9 >> 2) & 1u) | ((m9 >> 4) & 2u) | ((m9 >> 6) & 4u));
  cx = bx * 8 + (int)(local & 7u); cy = by * 8 + (int)((local >> 3) & 7u); cz = bz * 8 + (int)(local >> 6);
}
__host__ __device__ inline uint32_t pt_local_cell(int cx, int cy, int cz) {
  return (uint32_t)(((cz & 7) << 6) | ((cy & 7) << 3) | (cx & 7));
}

#ifdef __HIPCC__
// cell coordinate along one axis: clamp(floor((p - bbmin) * inv_h), 0, dim-1), all in fp64
__device__ inline int pt_cell_axis(double p, double bbmin, double inv_h, int dim) {
  double u = (p - bbmin) * inv_h;
  u = fmin(fmax(u, 0.0), (double)(dim - 1));
  return (int)u;   // u >= 0: truncation == floor
}
template <class Rec>
__device__ inline void pt_cell_of(const GridParams& gp, const Rec& r, int& cx, int& cy, int& cz) {
  cx = pt_cell_axis((double)r.x, gp.bbmin[0], gp.inv_h, gp.dim[0]);
  cy = pt_cell_axis((double)r.y, gp.bbmin[1], gp.inv_h, gp.dim[1]);
  cz = pt_cell_axis((double)r.z, gp.bbmin[2], gp.inv_h, gp.dim[2]);
}

// ---- workgroup scan helpers (256-thread workgroups) ---------------------------------------------------
__device__ inline uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}
// exclusive scan of one value per thread over a workgroup of NW waves; wsum: LDS scratch of NW words
template <int NW>
__device__ inline uint32_t block_excl_scan_n(uint32_t v, uint32_t* wsum, uint32_t& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(v);
  __syncthreads();
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  uint32_t off = 0;
  total = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const uint32_t ws = wsum[i];
    if (i < w) off += ws;
    total += ws;
  }
  return off + incl - v;
}
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* wsum, uint32_t& total) { return block_excl_scan_n<4>(v, wsum, total); }

#endif
