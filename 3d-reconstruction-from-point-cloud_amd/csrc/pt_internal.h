// pt_internal.h -- host-side launch interface between the C-ABI glue (pt_api.hip) and the kernel
// translation units (pt_grid.hip, pt_query.hip, pt_attr.hip).  Everything takes the stream to launch on;
// nothing here allocates or synchronises.
#pragma once
#include "pt_common.h"

// tables used by one sort (source cloud or target set)
struct SortTables {
  uint32_t* counts1;     // [PT_MAXBINS+1]     pass-1 histogram (macro blocks, or blocks if 1-level)
  uint32_t* start1;      // [PT_MAXBINS+1]     exclusive scan of counts1 (+ total)
  uint32_t* cursor1;     // [PT_MAXBINS]
  uint32_t* tile_first1; // [2]                one segment: {0, ntiles}
  uint32_t* seg_start1;  // [2]                {0, n}
  uint32_t* tile_first2; // [PT_MAXBINS+1]     tiles per pass-1 segment, scanned
  uint32_t* block_count; // [nblocks+1]
  uint32_t* block_start; // [nblocks+1]
  uint32_t* cursor2;     // [nblocks]
  uint32_t* scan_tmp;    // [>= nblocks/2048 + 2]
  uint32_t* chunk_hist;  // [nchunks][nbins1]   per-chunk pass-1 histograms, then per-chunk bin cursors
  uint32_t* chunk_gsum;  // [ceil(nchunks/64)][nbins1]
  uint32_t* occupied;    // optional [nblocks]: non-empty cells of every block | points of its fullest cell << 10, written by finalize (may alias block_count)
  RecF* shadow32;        // optional [npoints] (fp64 clouds): fp32-rounded copy of the sorted records, id = sorted position (tile kernel's LDS image)
  uint16_t* bid;         // [npoints] (two-level sorts): block-in-macro of every record as pass 1 placed it -- what pass 2's histogram reads
  uint32_t* countsM;     // three-level sorts (more than PT_MAXBINS macro blocks) only: [macros+1] histogram of the macro blocks, its scan,
  uint32_t* startM;      //   the scatter cursors and the tile table of the last partition pass (segments = macro blocks)
  uint32_t* cursorM;
  uint32_t* tile_firstM;
  // pooled pass 1 (two-level sorts of big clouds; pool_records == 0: the exact, histogram-first pass 1): bin regions sized from a
  // sample with slack between them, space taken in blocks, sentinel padding that pass 2 drops (pt_grid.hip, scatter_pool_kernel)
  uint32_t* pool_est;    // [PT_MAXBINS]  sample counts
  uint32_t* pool_limit;  // [PT_MAXBINS]  end of every bin's region
  uint32_t* pool_flag;   // [4] {a region overflowed, points sampled, records incl. sentinels after pass 1, -}
  int occ_lo[3], occ_hi[3];  // pooled pass 2: the cells [occ_lo, occ_hi) per axis the cloud is expected to occupy (a grid laid out from a sampled
                         // bounding box carries empty padding cells around it; the whole grid otherwise)
  uint32_t* rstart;      // [nblocks+1] pooled pass 2: where every block's region starts in the pass-2 output
  uint64_t pool2_records;// capacity (records) of the pass-2 output when pass 2 is pooled (0: exact pass 2), scratch area included
  uint64_t pool_records; // capacity (records) of the pass-1 output and of `bid`, scratch area of one tile at its end included
  uint32_t pool_nwg;     // persistent workgroups of the pooled pass 1 (one per CU)
  hipError_t* status;    // optional: receives the first HIP error of the sort's launch path (hipSuccess otherwise)
  hipEvent_t* ev;        // optional [6]: start, after hist1, scatter1, hist2+scan, scatter2, finalize (null = no timing)
};

// ---- pt_grid.hip ------------------------------------------------------------------------------
// bbox of planar xyz (T = float/double); out6 = orderable-u64 encoded {min xyz, max xyz}; init first.
void pt_launch_bbox_init(uint64_t* out6, hipStream_t s);
template <class T> void pt_launch_bbox(const T* x, const T* y, const T* z, uint32_t n, uint64_t* out6, hipStream_t s);
template <class T> void pt_launch_bbox_sample(const T* x, const T* y, const T* z, uint32_t n, uint32_t stride, uint64_t* out6, hipStream_t s);
double pt_bbox_decode(uint64_t enc);

// Sort n points into cell order.  Planar input (x,y,z[,gidx]); `out_final` and `tmp` are record buffers of n entries.
// With do_finalize the result is sorted by cell key and cell_start (u32[nblocks*512+1]) is written when non-null;
// without it the records are only grouped by 8^3-cell block (tb.block_start delimits the groups) -- all the tile
// kernel needs of the TARGETS.  Returns the buffer that holds the result.  Rec = RecF (T=float) or RecD (T=double).
template <class T, class Rec>
const Rec* pt_launch_grid_sort(const GridParams& gp, const T* x, const T* y, const T* z, const uint32_t* gidx, uint32_t n,
                               Rec* out_final, Rec* tmp, uint32_t* cell_start, const SortTables& tb, bool do_finalize, hipStream_t s,
                               uint64_t* bbox6_verify = nullptr);   // two-level sorts: pass 1's histogram also reduces the exact bbox into it
// does a 1/64 sample of the cloud sit in the grid's blocks the way a uniform cloud would (pooled pass 2 on the FIRST build)?  scratch:
// pt_uniform_probe_acc_offset(nblocks) + 8 + 16 nblocks + 2 nblocks / 512 words (one bit per cell the sample has seen; two words per macro block); *flag (device) = 1 when some block holds far more sample points than its macro block's
// count predicts; the three 64-bit words at scratch + offset: chi-square sum over the blocks (x 1024, fixed point), how many blocks it is over,
// an estimate of the number of cells the cloud occupies on this grid (x 16), and how many sample points fell into the block of their wave's
// first point (64 consecutive points: a cloud stored in spatial order gives itself away)
inline uint32_t pt_uniform_probe_acc_offset(uint32_t nblocks) { return (nblocks + nblocks / PT_MACRO_BLOCKS + 4u) & ~3u; }      // (a multiple of four words: 16-byte reads of what follows)
template <class T>
void pt_launch_uniform_probe(const GridParams& gp, const T* x, const T* y, const T* z, uint32_t n, const int occ_lo[3], const int occ_hi[3],
                             uint32_t* scratch, uint32_t* flag, hipStream_t s);
void pt_launch_sum_u32(const uint32_t* v, uint32_t n, uint32_t* out, hipStream_t s);   // finalize's block words: out[0] += non-empty cells, out[1] = max(fullest cell)
int pt_sort_tile_points(size_t rec_size);
// capacity (records) the pooled pass 1 needs for n points in nbins bins with nwg persistent workgroups: the regions' worst case
// for ANY distribution of the sample over the bins, plus the scratch area; 0 = the cloud is not pooled (too small, too large)
uint64_t pt_sort_pool_records(uint32_t n, uint32_t nbins, uint32_t nwg, size_t rec_size);
uint64_t pt_sort_pool2_records(uint64_t n_in, uint32_t nblocks, size_t rec_size);
// grids of more than PT_MAXBINS macro blocks: pass 1 partitions by GROUPS of 2^shift macro blocks (at most PT_MAXBINS groups)
inline int pt_sort_group_shift(uint32_t nblocks) {
  const uint32_t nm = nblocks / PT_MACRO_BLOCKS;
  int sh = 0;
  while (((nm + (1u << sh) - 1u) >> sh) > (uint32_t)PT_MAXBINS) ++sh;
  return sh;
}
int pt_sort_chunk_tiles(uint32_t n, size_t rec_size);
uint32_t pt_sort_num_chunks(uint32_t n, size_t rec_size);

// generic exclusive scan of u32 (n <= 2048*2048*... see pt_grid.hip); out may alias in
void pt_launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* tmp, hipStream_t s);

// ---- pt_query.hip -----------------------------------------------------------------------------
// k-NN of m sorted target records against the sorted source records.  bound2 (may be null) is indexed by
// the target's id.  Results go to out_idx/out_d2 at row `id` (k entries per row).
template <class Rec>
void pt_launch_knn(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const Rec* tgt, uint32_t m, int k,
                   const double* bound2, uint32_t* out_idx, double* out_d2, const uint32_t* list, const uint32_t* list_n, hipStream_t s,
                   uint8_t* heavy = nullptr, uint32_t wave_min = 0);
// (heavy != null, one zeroed byte per target position: targets whose 27 nearest cells hold >= wave_min points are MARKED instead of
//  being answered -- 1, or 2 when one of those cells is refined; pt_launch_mark_count / _write turn the marks into ordered lists)
void pt_launch_mark_count(const uint8_t* mark, uint32_t m, uint32_t* off1, uint32_t* off2, uint32_t* scan_tmp, hipStream_t s);   // off*: [tiles + 1], totals last
void pt_launch_mark_write(const uint8_t* mark, uint32_t m, const uint32_t* off1, const uint32_t* off2, uint32_t* list1, uint32_t* list2, hipStream_t s);
uint32_t pt_mark_tiles(uint32_t m);
// the marks without the group kernel: every listed target (list == null: all m) gets 1, or 2 when its own cell carries the `near` flag
template <class Rec>
void pt_launch_mark_near(const GridParams& gp, const Rec* tgt, const uint32_t* list, const uint32_t* list_n, uint32_t m, const uint8_t* near, uint8_t* mark, hipStream_t s);
// quad-per-target LDS tile kernel (fp32 records, k <= 32); leftovers go to todo[*todo_n] and are finished by pt_launch_knn(list=todo)
// staged-region capacities (records) of the tile kernel's geometries: what is left of 80 KB (two workgroups per CU) or
// 160 KB (one) after the per-lane queue segments and the cell table
// the blocks with at least one target (tblock_start: the target sort's block table), appended to list[*count] (count zeroed here)
void pt_launch_tblock_list(const uint32_t* tblock_start, uint32_t nblocks, uint32_t* list, uint32_t* count, hipStream_t s);
constexpr int PT_TILE_CAP_SMALL_8 = 4400, PT_TILE_CAP_SMALL_16 = 3888, PT_TILE_CAP_LARGE = 8448, PT_TILE_CAP_WIDE = 8960;
constexpr int PT_TILE_MAX_K = 32;      // beyond this the group kernel answers everything
void pt_launch_knn_tile(const GridParams& gp, const RecF* src, const uint32_t* cell_start, const RecF* tgt, const uint32_t* tblock_start,
                        int k, uint32_t* out_idx, double* out_d2, uint32_t* todo, uint32_t* todo_n, int geometry, const Attr* attr, uint32_t n_attr,
                        int mode, float* rgb_out, float* nrm_out, const uint32_t* blocks, uint32_t nblocks_listed, uint32_t* retry,
                        uint32_t* retry_n, const RecD* src_exact, const RecD* tgt_exact, float e_src, hipStream_t s, const double* bound = nullptr);
template <class T>
void pt_launch_request_pack(const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g,
                            int my_slab, uint32_t* count, uint32_t* sel, double* pkt, hipStream_t s);
void pt_launch_merge(const uint32_t* idx_lists, const double* d2_lists, int g, uint32_t m, int k, uint32_t* idx_out,
                     double* d2_out, hipStream_t s);
// streamed sources: merge the running best lists (u64 ids) with one chunk's lists (chunk-local u32 ids + base), out of place
template <class T>
void pt_launch_stream_sweep(const T* xyz_planar, const unsigned long long* best_idx, const double* best_d2, uint32_t m, int k, uint32_t chunk, int backward,
                            uint32_t* first, const double lo[3], const double hi[3], double margin, double* bound, uint32_t* count, hipStream_t s);
void pt_launch_merge_stream(const unsigned long long* best_idx, const double* best_d2, const uint32_t* chunk_idx, const double* chunk_d2,
                            unsigned long long base, uint32_t m, int k, unsigned long long* out_idx, double* out_d2, hipStream_t s);
template <class T>
void pt_launch_slab_need(const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis,
                         const double* bounds_dev, int g, int my_slab, uint8_t* need, hipStream_t s);

// ---- pt_attr.hip ------------------------------------------------------------------------------
// SURVEY.md Appendix C generator.  Writes planar xyz (T) for indices [0,n_total) whose `axis` coordinate is
// in [lo,hi) to x/y/z/gidx, appending through *counter (device u32, zeroed by the caller).
template <class T>
void pt_launch_synth_xyz(uint64_t seed, uint64_t stream, uint32_t n_total, int axis, double lo, double hi, T* x, T* y, T* z,
                         uint32_t* gidx, uint32_t* counter, uint32_t capacity, int round_f16, int dist, uint64_t src_total,
                         uint64_t tgt_total, hipStream_t s, uint32_t* wg_cnt = nullptr, const uint32_t* wg_off = nullptr);      // wg_cnt / wg_off: slabs in index order (pt_attr.hip)
void pt_launch_half_to_float(const void* in_half, float* out, uint64_t count, hipStream_t s);
void pt_launch_float_to_double(const float* in, double* out, uint64_t count, hipStream_t s);
void pt_launch_synth_attr(uint64_t seed, uint32_t n_total, Attr* attr, hipStream_t s, const uint32_t* gidx = nullptr);      // gidx: n_total records, of the points gidx[j]
// reference AoS records (80-B stride, device copy) -> planar f64 xyz + attribute table
void pt_launch_aos_split(const void* aos, uint32_t n, double* x, double* y, double* z, Attr* attr, hipStream_t s);
void pt_launch_pack_attr(const uint8_t* rgb, const float* nrm, uint32_t n, Attr* attr, hipStream_t s);
void pt_launch_blend_weighted(const uint32_t* idx, const double* w, uint32_t m, int k, const Attr* attr, uint32_t n_attr, float* rgb_out,
                              float* nrm_out, hipStream_t s);
template <class Rec>
void pt_launch_blend_list(const uint32_t* list, const uint32_t* list_n, uint32_t m_max, const Rec* tgt, const uint32_t* idx, const double* d2, int k,
                          int mode, const Attr* attr, uint32_t n_attr, float* rgb_out, float* nrm_out, hipStream_t s);
void pt_launch_blend(const uint32_t* idx, const double* d2, uint32_t m, int k, int mode, const Attr* attr, uint32_t n_attr,
                     float* rgb_out, float* nrm_out, hipStream_t s);
template <class T>
void pt_launch_pca(const uint32_t* idx, uint32_t m, int k, const T* x, const T* y, const T* z, uint32_t n, const Attr* attr,
                   float* nrm_out, hipStream_t s);
// fp32 clouds: 32-byte {position, attributes} records by original index (built once per cloud) and the PCA pass over them
void pt_launch_pack_posattr(const float* x, const float* y, const float* z, const Attr* attr, uint32_t n, void* out, hipStream_t s);
void pt_launch_pca_posattr(const uint32_t* idx, uint32_t m, int k, const void* posattr, uint32_t n, int has_attr, float* nrm_out, hipStream_t s);
void pt_launch_iota(uint32_t* p, uint32_t n, hipStream_t s);

// ---- pt_bake.hip ------------------------------------------------------------------------------
// per-face texture bake (reference src/pointsTransfer.cpp:466-581, :66-107): every covered pixel of the R x R atlas does an
// atomicMax on keys[] (zeroed by the caller) with {face * 256 + triangle + 1, BGRA}; resolve keeps the BGRA of the last triangle.
// sx/sy/sz: planar source xyz by ORIGINAL index; verts_aos: the mesh's 80-byte reference records (device copy);
// nbr: the mesh vertices' neighbour lists [nv][k] (original indices).
template <class T>
void pt_launch_bake_faces(const T* sx, const T* sy, const T* sz, const Attr* attr, uint32_t n, const void* verts_aos, uint32_t nv, const int32_t* faces,
                          uint32_t nf, const uint32_t* nbr, int k, int R, unsigned long long* keys, hipStream_t s);
void pt_launch_bake_resolve(const unsigned long long* keys, uint32_t* bgra, size_t npix, hipStream_t s);
// edge padding (reference :593-611): out = tex + (dilate(tex, ksize x ksize) & ~alpha), saturating; tmp: R*R words of scratch
void pt_launch_dilate_pad(const uint32_t* tex, uint32_t* tmp, uint32_t* out, int R, int ksize, hipStream_t s);

// ---- pt_exchange.hip ------------------------------------------------------------------------------
// requests of the targets that need another slab: fill = false counts them per destination slab (counts[g], zeroed by the caller),
// fill = true writes packets {x, y, z, k-th d2} at off[s] + cursor[s]++ (cursor zeroed by the caller) and the row they came from
template <class T>
void pt_launch_xreq(bool fill, const T* x, const T* y, const T* z, const double* d2, uint32_t m, int k, int axis, const double* bounds_dev, int g, int me,
                    uint32_t* counts, const uint32_t* off, uint32_t* cursor, double* req, uint32_t* req_row, hipStream_t s);
template <class T>
void pt_launch_xunpack(const double* rreq, uint32_t r, T* xyz_planar, double* bound, hipStream_t s);
// merge bucket answers (bi, bd)[cnt][k] into rows[e]'s lists in place; flags[row] = 1 for every row touched
void pt_launch_xmerge(const uint32_t* rows, uint32_t cnt, const uint32_t* bi, const double* bd, int k, uint32_t* idx, double* d2, uint8_t* flags, hipStream_t s);
void pt_launch_xflag_rows(const uint8_t* flags, uint32_t m, uint32_t* rows, uint32_t* count, hipStream_t s);
// slabs with LOCAL ids in their records (ascending gidx) and their own points' attribute records only (pt_exchange.hip)
void pt_launch_ids_to_global(uint32_t* idx, size_t count, const uint32_t* gidx, hipStream_t s);                       // idx[i] = gidx[idx[i]] (NOIDX stays)
void pt_launch_ids_to_local(const uint32_t* in, size_t count, const uint32_t* gidx, uint32_t n, uint32_t* out, hipStream_t s);      // binary search; NOIDX for another slab's points
void pt_launch_xgather_attr(const uint32_t* ids, size_t count, const uint32_t* gidx, uint32_t n, const Attr* attr, Attr* out, hipStream_t s);
void pt_launch_xmerge_attr(const uint32_t* rows, uint32_t cnt, const uint32_t* bi, const double* bd, const Attr* ba, int k, uint32_t* idx, double* d2, Attr* rattr,
                           uint8_t* flags, const uint32_t* gidx, uint32_t n, const Attr* attr, hipStream_t s);
void pt_launch_blend_rows_attr(const uint32_t* rows, const uint32_t* rows_n, uint32_t m_max, const uint32_t* idx, const double* d2, const Attr* rattr, int k, int mode,
                               float* rgb_out, float* nrm_out, hipStream_t s);
void pt_launch_ascending(const uint32_t* gidx, uint32_t n, uint32_t* flag, hipStream_t s);                             // *flag |= 1 unless gidx is strictly ascending
// blend of the listed rows (row ids, not sorted positions) from their idx / d2 lists
void pt_launch_blend_rows(const uint32_t* rows, const uint32_t* rows_n, uint32_t m_max, const uint32_t* idx, const double* d2, int k, int mode,
                          const Attr* attr, uint32_t n_attr, float* rgb_out, float* nrm_out, hipStream_t s);

// ---- pt_refine.hip ------------------------------------------------------------------------------
// cells with more than `threshold` points become nodes (cell_node[c] = node id + 1, else 0; *node_count counts them, also past node_cap)
void pt_launch_heavy_cells(const GridParams& gp, const uint32_t* cs, uint32_t ncells, uint32_t threshold, uint32_t* cell_node, uint32_t* node_count,
                           uint32_t node_cap, uint32_t* nodes, uint8_t* near_or_null, hipStream_t s);   // near: one zeroed byte per cell, set for the 27 cells around every node
// the same one level down for the nodes [n0, n1) (their tables must be built); threshold 0xFFFFFFFF just clears the child tables
void pt_launch_heavy_subcells(uint32_t n0, uint32_t n1, uint32_t threshold, uint32_t* node_count, uint32_t node_cap, uint32_t* nodes, hipStream_t s);
// counting sort of the records of the nodes [n0, n1) by sub-cell, in place (tmp: scratch of the same size as rec), + their start tables
template <class Rec>
void pt_launch_refine_nodes(const GridParams& gp, Rec* rec, Rec* tmp, uint32_t n0, uint32_t n1, uint32_t* nodes, hipStream_t s);
// leaves of the nodes [n0, n1) that hold more than PT_DUP_KEEP copies of ONE position: the PT_DUP_KEEP lowest indices to the front, the
// leaf's child link tagged (pt_common.h); tmp: scratch as large as rec; stats2 (device, may be null): += {leaves tagged, points behind a front}
template <class Rec>
void pt_launch_dedup_leaves(Rec* rec, Rec* tmp, uint32_t n0, uint32_t n1, uint32_t* nodes, uint32_t* stats2, hipStream_t s);
void pt_launch_reshadow(const RecD* rec, uint32_t n, RecF* shadow, hipStream_t s);
// k-NN over the refined grid (group kernel with hierarchical cell scans): same contract as pt_launch_knn
template <class Rec>
void pt_launch_knn_hier(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const uint32_t* cell_node, const uint32_t* nodes, uint32_t node_thr,
                        const Rec* tgt, uint32_t m, int k, const double* bound2, uint32_t* out_idx, double* out_d2, const uint32_t* list,
                        const uint32_t* list_n, hipStream_t s, uint8_t* heavy = nullptr, uint32_t wave_min = 0);
// one wave per target (dense neighbourhoods, k <= 64): the targets at positions list[0 .. count) of the sorted target array, or the
// first `count` targets when list is null (list_n, if given, holds the live count on the device); cell_node / nodes may be null
template <class Rec>
void pt_launch_knn_wave(const GridParams& gp, const Rec* src, const uint32_t* cell_start, const uint32_t* cell_node, const uint32_t* nodes, uint32_t node_thr,
                        const Rec* tgt, uint32_t count, int k, const double* bound2, uint32_t* out_idx, double* out_d2, const uint32_t* list,
                        const uint32_t* list_n, hipStream_t s, const Attr* attr = nullptr, uint32_t n_attr = 0, int blend_mode = 0, float* rgb_out = nullptr,
                        float* nrm_out = nullptr);   // attr != null: the neighbours' attributes are blended in the same launch (rows of these targets only)
