/*
 * pt_api.h -- C ABI of libpt_hip.so, the MI355X (gfx950) implementation of the
 * point-based detail-transfer hot path of
 * horizon-research/3D-Reconstruction-From-Point-Cloud.
 *
 * The reference has no FFI / plugin layer: its hot path is inline
 * code in main().  Each entry point below names the reference lines it replaces;
 * INTEGRATION.md shows the patch a maintainer applies to src/pointsTransfer.cpp.
 *
 * Conventions
 *   - every function returns 0 on success, a negative pt_status otherwise; nothing
 *     throws across the boundary; pt_last_error() gives the text of the last failure;
 *   - the caller owns every buffer; nothing passed in is retained after return unless
 *     stated ("_dev" functions read device pointers during the call only);
 *   - one context per calling thread and per GPU (one process per GPU for multi-GPU);
 *   - neighbours of a target are returned ascending under the total order
 *     (d2 as IEEE double, original index as uint32) with
 *       d2 = (dx*dx + dy*dy) + dz*dz,  dx = (double)t.x - (double)p.x,  no FMA
 *     -- bit-for-bit src/Distance.h:6-11 as the reference's Release flags compile it;
 *   - `idx` values are positions in the caller's ORIGINAL cloud order (or the global
 *     indices given to pt_build_soa_indexed); missing neighbours (k > N) are
 *     PT_NOIDX with d2 = +inf;
 *   - planar xyz: `xyz` points at 3*n elements, x[0..n) then y[0..n) then z[0..n).
 *   - there is NO CPU fallback: every compute entry point fails with PT_ERR_HIP when no
 *     gfx950 device is usable.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_NOIDX 0xFFFFFFFFu
#define PT_MAX_K 32

typedef enum {
  PT_OK = 0,
  PT_ERR_ARG = -1,      /* bad argument (null pointer, k out of range, n too large ...) */
  PT_ERR_HIP = -2,      /* HIP runtime error / no device */
  PT_ERR_STATE = -3,    /* call order (query before build ...) */
  PT_ERR_NOMEM = -4,    /* device allocation failed */
  PT_ERR_UNSUPPORTED = -5
} pt_status;

typedef enum { PT_F32 = 0, PT_F16 = 1, PT_F64 = 2 } pt_xyz_type;
typedef enum { PT_DIST_UNIFORM = 0, PT_DIST_CLUSTERED = 1 } pt_synth_dist;
typedef enum { PT_BLEND_MEAN = 0, PT_BLEND_INV_D2 = 1 } pt_blend_mode;

/* The reference's record (src/Point.h:1-6), sizeof == 80, offsets 0/24/48/64/72.
 * include/Point.h carries the C++ struct with the reference's member functions;
 * this POD twin is what crosses the C boundary. */
typedef struct pt_point {
  double ver[3];
  double normal[3];
  int32_t color[3];
  int32_t _pad;
  double U;
  double V;
} pt_point;

typedef struct pt_ctx pt_ctx;

/* per-phase device timings (HIP events on the context's stream) and byte accounting */
typedef struct pt_stats_t {
  uint64_t n_source, n_target; int32_t k, _pad;
  double ms_build;          /* grid build over the source cloud (last build) */
  double ms_sort_targets;   /* target binning (last query) */
  double ms_query;          /* k-NN kernel (last query) */
  double ms_blend;          /* attribute gather + blend (last blend) */
  double ms_pca;            /* PCA normals (last pca) */
  uint64_t bytes_alg_build; /* N*(2s+4) */
  uint64_t bytes_alg_query; /* N*s + M*s + M*k*16 + M*(4k+24) */
  int32_t grid_dim[3];      /* cells per axis */
  int32_t n_levels;         /* partition passes used by the build (1 or 2) + finalize */
  double cell_size;
  uint64_t n_cells;
  uint64_t device_bytes;    /* bytes currently allocated by the context */
  /* per-kernel device times of the last build / query (HIP events on the launch stream), ms:
   * [0] bbox reduce + readback, [1] pass-1 histogram, [2] pass-1 scatter, [3] pass-2 histogram + block scan,
   * [4] pass-2 scatter, [5] finalize (cell sort), [6] target sort (all passes), [7] k-NN kernel */
  double ms_kernel[8];
  uint64_t n_leftover;      /* targets of the last query that the tile kernel handed to the group kernel */
  double rho_occupied;      /* points per NON-EMPTY cell of the last build (0 when adaptive is off) */
  int32_t n_refine;         /* how many times the last build refined its cell size */
  int32_t bbox_guess;       /* last build: 0 the bounding box came from a pass of its own; 1 the grid was laid out from a sampled
                             * box and pass 1 verified it (big clouds); -1 the sampled box was too small and the build was redone */
  double ms_bake;           /* texture bake (+ edge padding) of the last pt_bake_texture, device time */
  uint32_t n_nodes;         /* last build: refined ("heavy") cells and sub-cells that carry an 8x8x8 sub-grid (0: none needed) */
  int32_t refine_levels;    /* ... and how many levels deep (<= 3) */
  uint32_t max_cell_points; /* points of the fullest grid cell of the last build (adaptive builds) */
  uint32_t n_wave;          /* last query: targets answered by the one-wave-per-target kernel (dense neighbourhoods) */
  int32_t pass1_pooled;     /* last build: 1 pass 1 took its bin regions from a sample (no histogram pass: big clouds, two-level sort);
                             * -1 a bin outgrew its sampled region and the build was redone with the exact pass 1; 0 exact pass 1 */
  int32_t stream_skipped;   /* last pt_stream_query: (chunk, forward sweep) steps it did not search because no target's bound reached the chunk's box */
  int32_t stream_revisited; /* ... and chunks its backward sweep brought back for targets that lay outside them before they had a list */
  int32_t pass2_pooled;     /* last build: 1 pass 2 took its block regions from the macro counts (no pass-2 histogram: a rebuild of a resident cloud the
                             * previous build found uniform, or -- round 4 -- a first build whose 1/64 sample said so); -1 a block outgrew its region and
                             * the build was redone exactly; 0 exact pass 2 */
  int32_t uniform_probe;    /* last build: 1 a 1/64 sample taken before the sort found the cloud uniform (pooled pass 2 on a FIRST build), -1 it did
                             * not, 0 not asked (small cloud, one- or three-level grid, or a previous build of this cloud already knew) */
  int32_t dup_leaves;       /* last build: leaves of refined cells found to hold ONE position more than 32 times (quantised clouds): a search reads their
                             * 32 lowest indices only */
  int32_t presort_refine;   /* last build: refinements of the cell size decided from the sample BEFORE the first sort (0 .. 3; first builds of clouds the
                             * sample found non-uniform); negative: the sort's own count overruled them (a cloud stored in spatial order) and the grid was
                             * laid out again */
  int32_t n_sorts;          /* last build: full sorts of the cloud it ran (1; more when the cell size was refined after a count, a sampled box or a
                             * pooled pass had to be redone) */
  int32_t ordered_input;    /* last build: 1 = the sample found the cloud stored in spatial order (of 64 consecutive points most share a block): regions and
                             * cell size are then not taken from a sample */
} pt_stats_t;

/* ---- context ------------------------------------------------------------------------ */
/* device_ids[0] is the GPU this context runs on (one process per GPU); n_devices must be 1. */
int  pt_ctx_create(pt_ctx** out, const int* device_ids, int n_devices);
void pt_ctx_destroy(pt_ctx*);
/* Plumbing: run all work of this context on the caller's hipStream_t (e.g. torch's current stream), so that
 * kernels queue behind whatever produced the device buffers handed in.  NULL = HIP's default stream;
 * pt_set_param(ctx, "own_stream", 1) returns to the context's own (non-blocking) stream. */
int  pt_set_stream(pt_ctx*, void* hip_stream);
/* Tunables: "k_hint" (the k later queries will use: picks the cell density before a build; default 8), "rho" (points
 * per grid cell, set directly; default 4), "sync" (1 = every call blocks until
 * the GPU is done, default 1; 0 = _dev calls only enqueue), "adaptive" (1 = refine the cell
 * size when non-empty cells hold far more than rho points, default; needs one host read-back per build), "tile" (0 = group kernel only, 1 = tile kernel +
 * group kernel for its leftovers with the geometry chosen from the cell density (default), 2 / 3 = force the small /
 * large tile geometry), "guess_min_points" (clouds at least this large lay their grid out from the bounding box of a
 * sample and verify it during the first partition pass instead of spending a pass on the exact box; default 8 Mi),
 * "pool_min_points" (clouds at least this large, on the two-level sort, size the bins of the first partition pass from a sample
 * instead of a histogram pass over the whole cloud -- with slack, and a flag that sends the build back to the exact histogram when a
 * bin outgrows its estimate; default 32 Mi, 0 = never), "pool2" (1, default: a rebuild of a resident cloud that the previous build
 * found uniform -- no refinement, occupied cells at rho -- also sizes the blocks of the second partition pass from the macro counts
 * instead of a histogram pass, verified the same way; 0 = never),
 * "refine_threshold" (grid cells holding more points than this get an 8x8x8 sub-grid, recursively up to three levels, which
 * searches descend into instead of scanning the cell end to end -- clouds with strong density contrast; default 8192, 0 = never),
 * "wave_min" (on such clouds a target whose 27 nearest cells hold at least this many points is answered by a whole wave instead of
 * an 8-lane group; default 1 = every target of such a cloud, 0 = never; "wave_force" = 1 applies that split to every cloud -- a testing hook), "refine_macros"
 * (the finest grid the refinement of the cell size may ask for, in 64^3-cell macro blocks: default 1024, at most 8192 -- grids beyond
 * 1024 macro blocks, which a cloud of more than ~1e9 points gets anyway, cost the sort one more partition pass),
 * "refine_cells_per_point" (that refinement also stops at this many grid cells per point; default 2), "grid_hint" (1,
 * default: pt_rebuild of the same resident cloud starts from the cell size the previous build ended with -- checked against the
 * occupancy it finds -- instead of searching for it again; 0: every build searches from scratch), "stream_bounds" (1, default:
 * pt_stream_query searches every chunk under the bounds the targets bring and defers / skips what is out of reach; 0: every target,
 * unbounded, in every chunk -- round 2's behaviour, kept as a measurement switch).
 * Round 4: "forget" (1: the next build of the resident cloud decides everything a FIRST build decides -- sampled bounding box, pooled
 * passes, the uniformity sample, the cell size: what bench.py times as its cold step), "dup_runs" (1, default: leaves of refined cells
 * that hold ONE position more than 32 times keep their 32 lowest indices in front and searches read that front only; 0 = off),
 * "tile_sparse" (the tile kernel over a list of the blocks that hold targets: 0 never, 1 always, 2 = on clouds that leave most of
 * their grid empty, default), "tile_contrast" (1: on clouds with strong density contrast the tile kernel runs first, k <= 24, and the
 * wave kernel takes what it leaves; 0, default: a wave per target -- measured faster), "local_ids" (1: the next slab build -- ascending
 * global indices, or a slab pt_build_synth generates -- keeps positions in its records and its own attribute records only; see
 * pt_set_attributes_local), "presort_refine" (1, default: the first build of a big cloud the sample finds non-uniform refines its cell size
 * from the sample's bound on the points per occupied cell, before the first sort; 0: after it, from the sort's count -- round 3's behaviour). */
int  pt_set_param(pt_ctx*, const char* name, double value);
const char* pt_last_error(pt_ctx*);
int  pt_stats(pt_ctx*, pt_stats_t* out);
int  pt_synchronize(pt_ctx*);

/* ---- build: replaces `Tree tree(points.begin(), points.end())`, pointsTransfer.cpp:259 ---- */
/* AoS reference records on the host (80-B stride); coordinates are kept as double on the GPU. */
int  pt_build_aos(pt_ctx*, const pt_point* cloud, uint64_t n);
/* Planar xyz of `xyz_type` (+ optional interleaved rgb u8[n][3] and normals f32[n][3]); host or
 * device memory according to on_device. PT_F16 clouds stay fp16 in the resident input (6 bytes per
 * point) and are widened -- exactly -- as the build reads them; answers are those of the fp32 cloud
 * holding the same values. */
/* Slabs that keep their OWN points' attribute records only (round 4): after pt_set_param("local_ids", 1), a
 * pt_build_soa_indexed whose global indices are STRICTLY ASCENDING (PT_ERR_ARG otherwise) sorts each point's position in the slab's
 * arrays into the records -- positions order like indices, so results are unchanged -- and pt_set_attributes_local uploads the records
 * of exactly those n points in that order (rgb [n][3], nrm [n][3]): 16 n bytes per GPU instead of 16 N.  Finished neighbour lists carry
 * global indices as ever.  pt_exchange_merge_* then sends every candidate's record with it and blends the completed rows from what
 * arrived; every rank of a job must run the same mode.  pt_blend_dev on such a context blends the entries that belong to this slab
 * (lists the exchange completed are blended by the exchange). */
int  pt_set_attributes_local(pt_ctx*, const uint8_t* rgb, const float* nrm, int on_device);
int  pt_build_soa(pt_ctx*, const void* xyz, int xyz_type, const uint8_t* rgb, const float* nrm,
                  uint64_t n, int on_device);
/* Same, for one spatial slab of a larger cloud: gidx[i] is the point's index in the whole cloud
 * (what queries return); attributes stay indexed by that global index, see pt_set_attributes. */
int  pt_build_soa_indexed(pt_ctx*, const void* xyz, int xyz_type, const uint32_t* gidx, uint64_t n,
                          int on_device);
/* Attribute table indexed by global index (the whole cloud's, on every GPU). */
int  pt_set_attributes(pt_ctx*, const uint8_t* rgb, const float* nrm, uint64_t n_total, int on_device);
/* The same table filled piecewise from HOST memory: records [first, first + count) of a table of n_total (the first call, or a
 * change of n_total, allocates it zeroed).  For hosts that hold the attributes in pieces -- the ranks of `pointsTransfer --gpus N`
 * each parse 1/N of the file and read the other pieces from the rendezvous directory -- so that no rank ever assembles the whole
 * table in host memory. */
int  pt_set_attributes_range(pt_ctx*, uint64_t first, uint64_t count, const uint8_t* rgb, const float* nrm, uint64_t n_total);
/* SURVEY.md Appendix C generator, on the device.  Keeps the points whose coordinate along
 * `slab_axis` lies in [slab_lo, slab_hi) (pass -inf/+inf, or slab_axis < 0, for the whole cloud);
 * indices stay global; the attribute table is generated for all n_total points. */
int  pt_build_synth(pt_ctx*, uint64_t n_total, uint64_t seed, int dist, int xyz_type,
                    int slab_axis, double slab_lo, double slab_hi);
/* Re-run the grid build over the resident source cloud (what a bench step times). */
int  pt_rebuild(pt_ctx*);
uint64_t pt_num_source(pt_ctx*);        /* points resident in this context (slab-local) */

/* ---- query: replaces the K_neighbor_search loop, pointsTransfer.cpp:462-479 ---------------- */
int  pt_query_aos(pt_ctx*, const pt_point* targets, uint64_t m, int k, uint32_t* idx, double* d2_or_null);
int  pt_query_soa(pt_ctx*, const void* xyz, int xyz_type, uint64_t m, int k, int on_device,
                  uint32_t* idx, double* d2_or_null);
/* Generate m targets on the device (stream 1 of the generator) into the context; query them with
 * pt_query_resident.  tgt_lo/hi restrict to targets whose slab_axis coordinate is in [lo,hi).  The clustered
 * distribution derives targets from the sources (strided subsample + jitter): call it after a pt_build_synth with
 * the same seed. */
int  pt_targets_synth(pt_ctx*, uint64_t m_total, uint64_t seed, int dist, int xyz_type,
                      int slab_axis, double slab_lo, double slab_hi);
/* Make the caller's own targets resident (planar xyz, host or device; or host AoS Point records = mesh vertices), so that
 * pt_query_resident / pt_query_blend_resident work on them.  Same type rule as the queries: the targets' type must be the
 * cloud's (fp16 is widened to fp32; AoS records are fp64 like a cloud built with pt_build_aos). */
int  pt_targets_soa(pt_ctx*, const void* xyz, int xyz_type, uint64_t m, int on_device);
int  pt_targets_aos(pt_ctx*, const pt_point* targets, uint64_t m);
uint64_t pt_num_targets(pt_ctx*);
/* Query the resident targets; idx/d2 are DEVICE buffers of m*k entries (d2 may be NULL). */
int  pt_query_resident(pt_ctx*, int k, uint32_t* idx_dev, double* d2_dev_or_null);
/* pt_query_resident and pt_blend_dev in ONE pass over the resident targets (north_star's "find the k nearest
 * and blend colour/normal onto the vertex"): where the LDS tile kernel answers (fp32 clouds, k <= 32) it gathers
 * the k attribute records of a target as soon as it has ranked them, so the gathers overlap the ranking of other
 * targets instead of forming a pass of their own.  Same outputs as the two calls; the blend sums the same fp64
 * terms in a different order (within the 1e-5 tolerance of the path, not bit-identical to pt_blend_dev).
 * rgb_out_dev / nrm_out_dev: device float[m*3], either may be NULL. */
int  pt_query_blend_resident(pt_ctx*, int k, int blend_mode, uint32_t* idx_dev, double* d2_dev_or_null,
                             float* rgb_out_dev, float* nrm_out_dev);
/* Global index (position in the whole target set) of each resident target, device u32[m]. */
/* The same for a C++ host that holds no device memory (the CLI's --synthetic): idx / d2 / rgb / nrm come back to HOST memory
 * ([m][k], [m][k], [m][3], [m][3], m = pt_num_targets; d2 / rgb / nrm may be NULL, blend_mode < 0 skips the blend). */
int  pt_query_resident_host(pt_ctx*, int k, int blend_mode, uint32_t* idx_out, double* d2_out_or_null, float* rgb_out_or_null, float* nrm_out_or_null);
int  pt_resident_target_ids(pt_ctx*, uint32_t* ids_dev);
/* Planar xyz (f32 or f64 as generated) of the resident targets, copied to a device buffer. */
int  pt_resident_target_xyz(pt_ctx*, void* xyz_dev);
/* Planar xyz of the resident SOURCE cloud as it is kept (f32, f64, or f16 for clouds built from PT_F16), copied to a device buffer of
 * 3 * pt_num_source values; *xyz_type_out (may be null) receives the element type.  Tests and probes read generated clouds with it. */
int  pt_resident_source_xyz(pt_ctx*, void* xyz_dev, int* xyz_type_out);

/* ---- blend: the only blend arithmetic of the reference is pointsTransfer.cpp:95-97 --------- */
/* host buffers in / out */
int  pt_blend(pt_ctx*, const uint32_t* idx, const double* d2_or_null, uint64_t m, int k, int mode,
              float* rgb_out, float* nrm_out);
/* device buffers in / out */
int  pt_blend_dev(pt_ctx*, const uint32_t* idx_dev, const double* d2_dev_or_null, uint64_t m, int k,
                  int mode, float* rgb_out_dev, float* nrm_out_dev);
/* The reference's own mix formula (src/pointsTransfer.cpp:95-97: `float c = bc0*c0 + bc1*c1 + bc2*c2`, double weights
 * times int colours summed left to right in double, stored to a float; :100-102 assign it to an unsigned char) for k
 * terms with CALLER-given weights w[m*k] -- with k = 3 and barycentric weights it is bit-for-bit that expression.  No
 * normalisation; entries with idx = PT_NOIDX contribute nothing; normals get the same arithmetic. */
int  pt_blend_weighted(pt_ctx*, const uint32_t* idx, const double* w, uint64_t m, int k, float* rgb_out, float* nrm_out);
int  pt_blend_weighted_dev(pt_ctx*, const uint32_t* idx_dev, const double* w_dev, uint64_t m, int k,
                           float* rgb_out_dev, float* nrm_out_dev);
/* PCA normal of the k neighbours (BASELINE config 3); needs the whole cloud resident (no slabs). */
int  pt_pca_normals(pt_ctx*, const uint32_t* idx, uint64_t m, int k, float* nrm_out);
int  pt_pca_normals_dev(pt_ctx*, const uint32_t* idx_dev, uint64_t m, int k, float* nrm_out_dev);

/* ---- multi-GPU merge ---------------------------------------------------------- */
/* G-way merge of candidate lists under (d2, idx): lists are [g][m][k] device arrays. */
int  pt_merge_candidates_dev(pt_ctx*, const uint32_t* idx_lists_dev, const double* d2_lists_dev, int g,
                             uint64_t m, int k, uint32_t* idx_out_dev, double* d2_out_dev);
/* For each target t and each slab s != my_slab, need[s*m + t] = 1 iff slab s can still hold one of
 * t's k nearest: dist2(t, slab interval) <= d2[t][k-1], or the list is not full.  This is
 * Distance::min_distance_to_rectangle (src/Distance.h:27-57) applied to slab boxes.
 * slab_bounds: G+1 ascending doubles on the host. */
int  pt_slab_need_dev(pt_ctx*, const void* tgt_xyz_dev, int xyz_type, const double* d2_dev, uint64_t m, int k,
                      int slab_axis, const double* slab_bounds, int g, int my_slab, uint8_t* need_dev);
/* pt_slab_need_dev and the selection of the targets that need another slab in one pass: those targets leave as request
 * packets pkt_out[c][5] = {x, y, z, current k-th d2, bitmask of the slabs to ask (exact in a double: g <= 52)} with
 * their rows in sel_out[c]; *count_out = c (host).  Both outputs must hold m entries; the order is unspecified. */
int  pt_pack_requests_dev(pt_ctx*, const void* tgt_xyz_dev, int xyz_type, const double* d2_dev, uint64_t m, int k,
                          int slab_axis, const double* slab_bounds, int g, int my_slab, uint32_t* sel_out_dev,
                          double* pkt_out_dev, uint32_t* count_out);
/* Bounded query for foreign targets: like pt_query_soa(on_device=1) but each target starts from the
 * radius bound2[t] (its current k-th squared distance; +inf = unbounded): only points with
 * d2 <= bound2[t] are returned. */
int  pt_query_bounded_dev(pt_ctx*, const void* xyz_dev, int xyz_type, const double* bound2_dev, uint64_t m,
                          int k, uint32_t* idx_dev, double* d2_dev);

/* ---- native slab exchange: RCCL over xGMI behind the C ABI, one rank per process and GPU ---------------
 * After every rank has searched its HOME targets in its own slab (pt_query_*), pt_exchange_merge_dev completes the lists:
 *   1. the targets whose k-th distance reaches another slab (Distance::min_distance_to_rectangle, src/Distance.h:27-57, on
 *      the slab boxes) are counted per destination; ONE all-gather ships the G x G count matrix (its read-back is the exchange's only host wait);
 *   2. grouped ncclSend / ncclRecv carry 32-byte request packets {x, y, z, k-th d2} owner to owner (all links at once);
 *   3. every rank answers what it received with a radius-bounded search of its slab;
 *   4. the k candidates per request travel back the same way and are merged under the total order (d2, index);
 *   5. optionally the rows that were completed get their blend redone (blend_mode >= 0 and rgb / nrm outputs given).
 * pt_comm_unique_id (any one rank) creates the 128-byte RCCL id the ranks share out of band; pt_comm_init joins the
 * communicator on the context's device -- call it before the heavy GPU work of the process.  bounds: world + 1 ascending
 * slab bounds along `axis` (first / last may be -inf / +inf).  librccl is loaded at pt_comm_* time (dlopen), not at link time.
 * pt_exchange_merge_local runs the same phases for G contexts of ONE process with device copies as transport (G logical
 * slabs on one GPU: what tests use where only one GPU exists). */
#define PT_COMM_ID_BYTES 128
typedef struct pt_exchange_stats_t {
  uint64_t crossing;        /* request packets this rank sent (a target needing two slabs counts twice) */
  uint64_t answered;        /* request packets this rank answered */
  uint64_t bytes_sent, bytes_received;   /* requests + answers, this rank */
  double ms;                /* device time of the whole exchange on this rank's stream (HIP events) */
} pt_exchange_stats_t;
int  pt_comm_unique_id(void* id_out);
int  pt_comm_init(pt_ctx*, int world, int rank, const void* id);
int  pt_comm_destroy(pt_ctx*);
/* Error-path teardown: ncclCommAbort instead of ncclCommDestroy -- never waits for peers (which may be waiting for this rank).
 * pt_comm_destroy and pt_ctx_destroy take this path by themselves once an RCCL call or any phase of an exchange has failed on
 * the context; a host that gives up for reasons of its own (a failed build, a bad input file) calls it before exiting. */
int  pt_comm_abort(pt_ctx*);
int  pt_exchange_merge_dev(pt_ctx*, const void* tgt_xyz_dev, int xyz_type, uint64_t m, int k, int slab_axis, const double* slab_bounds,
                           uint32_t* idx_dev, double* d2_dev, int blend_mode, float* rgb_dev, float* nrm_dev, pt_exchange_stats_t* stats_or_null);
/* One rank's whole query for a C++ host that holds no device memory: the home targets (planar host xyz) are searched in this
 * rank's slab with the blend fused in, completed by pt_exchange_merge_dev (world > 1), and idx / d2 / rgb / nrm come back to
 * host memory ([m][k], [m][k], [m][3], [m][3]; rgb / nrm may be NULL, blend_mode < 0 skips the blend). */
int  pt_query_exchange_blend(pt_ctx*, const void* tgt_xyz, int xyz_type, uint64_t m, int k, int slab_axis, const double* slab_bounds, int blend_mode,
                             uint32_t* idx_out, double* d2_out, float* rgb_out, float* nrm_out, pt_exchange_stats_t* stats_or_null);
int  pt_exchange_merge_local(pt_ctx* const* ctxs, int g, const void* const* tgt_xyz_dev, int xyz_type, const uint64_t* m, int k, int slab_axis,
                             const double* slab_bounds, uint32_t* const* idx_dev, double* const* d2_dev, int blend_mode, float* const* rgb_dev,
                             float* const* nrm_dev);

/* ---- streamed upload of a planar cloud: what a file reader feeds while it is still parsing ---------
 * Replaces the copy of the points into the tree, `Tree tree(points.begin(), points.end())` (src/pointsTransfer.cpp:259),
 * for callers that hold x[] y[] z[] (+ rgb, normals) instead of 80-byte records: 39 bytes per point cross PCIe instead of 80.
 *   pt_host_alloc / pt_host_free   page-locked host memory (what makes the copies below asynchronous)
 *   pt_upload_begin                reserve a cloud of n points of xyz_type (PT_F32 / PT_F64), with or without attributes
 *   pt_upload_range                enqueue records [first, first + count): x, y, z point at `count` coordinates each,
 *                                  rgb at count * 3 bytes, nrm at count * 3 floats (both may be NULL when begun without
 *                                  attributes).  Thread-safe: parser threads call it as their ranges complete.  The memory
 *                                  must stay valid until pt_upload_end returns.
 *   pt_upload_end                  wait for the copies, pack the attribute table and build the grid (as pt_build_soa). */
void* pt_host_alloc(uint64_t bytes);
void  pt_host_free(void*);
int   pt_upload_begin(pt_ctx*, uint64_t n, int xyz_type, int with_attributes);
int   pt_upload_range(pt_ctx*, uint64_t first, uint64_t count, const void* x, const void* y, const void* z,
                      const uint8_t* rgb, const float* nrm);
int   pt_upload_end(pt_ctx*);

/* ---- out-of-core source (reference README.md:3 "billions of points") ------------------------------------
 * Searches the RESIDENT targets (pt_targets_*) in a cloud that stays in host memory: the cloud is cut into chunks of
 * `chunk_points` consecutive points, every chunk is uploaded (the next one while the current one is being searched: keep
 * `xyz` in page-locked memory, pt_host_alloc, for that overlap), gridded and searched like a resident cloud, and the k best
 * of the chunk are merged into the running k best under the same total order (d2, index).  Indices are 64-bit -- a streamed
 * cloud may hold more than 2^32 points -- and are `first_id` + the point's position in `xyz`.  The result is bit-identical to
 * a resident search of the whole cloud.  Every target brings a bound to every chunk: its current k-th squared distance once it has a
 * list; nothing (an unbounded search) in the first chunk whose bounding box contains it; and "not now" for chunks it lies outside of
 * before it has a list -- those (target, chunk) pairs are taken up by a second, backward sweep, under a bound by then.  A chunk that
 * no target's bound reaches is not searched (forward sweep) or not even uploaded again (backward sweep): a cloud stored in spatial
 * order -- what scanners and tiled exports deliver -- costs each target its own neighbourhood's chunks, a cloud in random order costs
 * what it did before (every chunk covers everything: no pair is ever deferred, the backward sweep uploads nothing).
 * xyz: planar, n points of xyz_type (PT_F32 / PT_F64; the targets' type);
 * idx64_out / d2_out: host, [m][k].  Afterwards NO source cloud is resident in the context (the chunks lived in the stage
 * buffers): a later pt_query_* needs a pt_build_* first and fails with PT_ERR_STATE otherwise. */
int  pt_stream_query(pt_ctx*, const void* xyz, int xyz_type, uint64_t n, uint64_t chunk_points, uint64_t first_id, int k,
                     uint64_t* idx64_out, double* d2_out);

/* ---- texture bake: the consumer of the neighbour lists -------------------------------------
 * pt_bake_texture replaces the body of the reference's face loop after the search and its rasteriser
 * (src/pointsTransfer.cpp:466-581 and draw_triangle :66-107): per face, the union of its three corners' neighbour
 * lists, projection into the face plane, the in-triangle filter, a Delaunay triangulation of corners + interior
 * points, and barycentric rasterisation of every sub-triangle into a resolution x resolution BGRA atlas addressed
 * (resolution - j, i) as the reference does.  pad_ksize > 0 additionally applies the reference's edge padding
 * (:593-611: ksize x ksize dilate, ~alpha mask, saturating add; the reference uses 25) before the atlas is copied out.
 * The source cloud must be resident (pt_build_*); `mesh_vertices` are the reference's records (ver, color, U, V are
 * read), `faces` holds 3 vertex indices per face, `nbr_idx` is the [nv][k] index matrix a pt_query_* call returned
 * for those vertices.  All host memory; bgra_out receives resolution * resolution * 4 bytes (B, G, R, A).
 * Where the reference's result is decided by CGAL / OpenCV internals or by undefined behaviour the result is defined
 * by this build (oracle/pt_oracle.c states the definition; INTEGRATION.md lists the points). */
int  pt_bake_texture(pt_ctx*, const pt_point* mesh_vertices, uint64_t nv, const int32_t* faces, uint64_t nf,
                     const uint32_t* nbr_idx, int k, int resolution, int pad_ksize, uint8_t* bgra_out);
/* The edge padding alone (reference :593-611) on a host BGRA image. */
int  pt_texture_pad(pt_ctx*, const uint8_t* bgra_in, int resolution, int ksize, uint8_t* bgra_out);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
