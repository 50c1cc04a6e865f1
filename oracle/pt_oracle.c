/*
 * pt_oracle.c -- CPU ORACLE for the point-based detail-transfer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package's csrc/,
 * the C-ABI library, the pointsTransfer CLI) may link, load or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / reported baseline.
 *
 * PARITY STATUS: "parity unpinned" for the k-NN search itself.  The reference
 * (horizon-research/3D-Reconstruction-From-Point-Cloud) delegates the search to
 * CGAL (Kd_tree / Orthogonal_k_neighbor_search; bare find_package(CGAL),
 * src/CMakeLists.txt:12 -- un-vendored, version unpinned, absent from this
 * image) and ships no tests, golden vectors or sample data (.gitignore:3-4).
 * What IS pinned: the Point record layout (oracle/_ref/point_layout, compiled
 * from the reference's own src/Point.h) and the closed-form metric/bound
 * formulas of src/Distance.h, restated below with their file:line.
 *
 * Everything here is plain C, double arithmetic, compiled with
 * -ffp-contract=off so that d2 = (dx*dx + dy*dy) + dz*dz is 3 mul + 2 add,
 * individually rounded -- what the reference's Release flags (-O3 -DNDEBUG, no
 * -march; src/CMakeLists.txt:7-9) produce for src/Distance.h:6-11.
 *
 * Array layout everywhere: planar xyz, i.e. xyz[0..n) = x, xyz[n..2n) = y,
 * xyz[2n..3n) = z.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PTO_NOIDX 0xFFFFFFFFu

/* ------------------------------------------------------------------------- */
/* Synthetic generator -- index-addressable SplitMix64 */
/* ------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline uint64_t stream_key(uint64_t seed, uint64_t stream) {
  return splitmix64(seed ^ (stream << 56));
}
static inline uint64_t hash4(uint64_t key, uint64_t i, uint64_t c) {
  return splitmix64(key + 4ull * i + c);
}
static inline float u24(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }

uint64_t pto_splitmix64(uint64_t z) { return splitmix64(z); }

/* uniform xyz in [0,1)^3 on the 2^-24 lattice; stream 0 = source, 1 = target */
void pto_synth_xyz_f32(uint64_t seed, uint64_t stream, uint64_t i0, uint64_t n, float* xyz /* planar [3][n] */) {
  const uint64_t key = stream_key(seed, stream);
  for (uint64_t j = 0; j < n; ++j) {
    const uint64_t i = i0 + j;
    xyz[j] = u24(hash4(key, i, 0));
    xyz[n + j] = u24(hash4(key, i, 1));
    xyz[2 * n + j] = u24(hash4(key, i, 2));
  }
}
/* ---- clustered distribution (BASELINE config 5; SURVEY.md Appendix C, frozen here) ------------------------------
 * 70 % thin planar patches (64 of them), 25 % Gaussian-ish blobs (256, sigma in [1e-3, 2e-2]), 5 % uniform; targets are a
 * strided subsample of the sources plus 5e-4 jitter.  Only +, -, * on floats and exact int->float conversions, each
 * individually rounded (-ffp-contract=off), so the HIP generator produces the same bits. */
static inline float gauss4(uint64_t h) {          /* Irwin-Hall(4), unit variance */
  const uint64_t a = splitmix64(h), b = splitmix64(a), c = splitmix64(b), d = splitmix64(c);
  return ((u24(a) + u24(b)) + (u24(c) + u24(d)) - 2.0f) * 1.7320508f;
}
static inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 0.99999994f ? 0.99999994f : v); }
static void clustered_source(uint64_t seed, uint64_t i, float* out) {
  const uint64_t k0 = stream_key(seed, 0), k4 = stream_key(seed, 4);
  const uint64_t sel = hash4(k4, i, 0);
  const uint32_t t = (uint32_t)(sel % 100u);
  const float ua = u24(hash4(k0, i, 0)), ub = u24(hash4(k0, i, 1));
  const uint64_t hc = hash4(k0, i, 2);
  if (t < 70u) {
    const uint64_t p = (sel >> 8) % 64u, kp = stream_key(seed, 5);
    const float g = gauss4(hc);
    for (int c = 0; c < 3; ++c) {
      const float o = u24(hash4(kp, p, (uint64_t)c));
      const float e1 = (2.0f * u24(hash4(kp, 64u + p, (uint64_t)c)) - 1.0f) * 0.3f;
      const float e2 = (2.0f * u24(hash4(kp, 128u + p, (uint64_t)c)) - 1.0f) * 0.3f;
      const float nn = 2.0f * u24(hash4(kp, 192u + p, (uint64_t)c)) - 1.0f;
      out[c] = clamp01(((o + ua * e1) + ub * e2) + (1e-4f * g) * nn);
    }
  } else if (t < 95u) {
    const uint64_t q = (sel >> 8) % 256u, kq = stream_key(seed, 6);
    const float sigma = 1e-3f + 1.9e-2f * u24(hash4(kq, q, 3));
    for (int c = 0; c < 3; ++c) out[c] = clamp01(u24(hash4(kq, q, (uint64_t)c)) + sigma * gauss4(splitmix64(hc + (uint64_t)c)));
  } else {
    out[0] = ua; out[1] = ub; out[2] = u24(hc);
  }
}
/* dist 0 = uniform (as pto_synth_xyz_f32), 1 = clustered; stream 0 = sources, 1 = targets (needs n_total, m_total) */
void pto_synth_xyz_dist(uint64_t seed, uint64_t stream, int dist, uint64_t i0, uint64_t n, uint64_t n_total, uint64_t m_total,
                        float* xyz /* planar [3][n] */) {
  if (dist == 0) { pto_synth_xyz_f32(seed, stream, i0, n, xyz); return; }
  const uint64_t k1 = stream_key(seed, 1);
  const uint64_t step = (m_total && n_total / m_total) ? n_total / m_total : 1;
  for (uint64_t j = 0; j < n; ++j) {
    float p[3];
    if (stream == 0) clustered_source(seed, i0 + j, p);
    else {
      const uint64_t t = i0 + j;
      clustered_source(seed, (t * step) % (n_total ? n_total : 1), p);
      for (int c = 0; c < 3; ++c) p[c] = clamp01(p[c] + 5e-4f * gauss4(splitmix64(hash4(k1, t, (uint64_t)c))));
    }
    xyz[j] = p[0]; xyz[n + j] = p[1]; xyz[2 * n + j] = p[2];
  }
}

/* Uniform generator, streamed and FILTERED: of the points i in [0, n_total) of (seed, stream) keep those inside any of
 * `nbox` axis-aligned boxes [lo, hi) (lo / hi: [nbox][3]) -- what lets a test check a billion-point GPU answer against
 * brute force without holding the cloud: every point is generated (same bits as pto_synth_xyz_f32), only a few thousand
 * are kept.  x is hashed first and tested against the boxes' x-ranges, so most points cost one hash.  OpenMP over index
 * chunks; the kept points come out in ascending index order.  Returns how many there were (may exceed `cap`: only the
 * first cap are written), or -1 on allocation failure. */
int64_t pto_synth_filter_boxes(uint64_t seed, uint64_t stream, uint64_t n_total, int nbox, const float* lo, const float* hi,
                               uint64_t cap, float* out_xyz /* [cap][3] interleaved */, uint32_t* out_idx, int32_t* out_box) {
  const uint64_t key = stream_key(seed, stream);
  const uint64_t chunk = 1ull << 22;
  const uint64_t nchunks = (n_total + chunk - 1) / chunk;
  uint64_t* cnt = (uint64_t*)calloc(nchunks + 1, sizeof(uint64_t));
  if (!cnt) return -1;
  for (int pass = 0; pass < 2; ++pass) {           /* pass 0 counts per chunk, pass 1 writes at the scanned offsets */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < (int64_t)nchunks; ++c) {
      const uint64_t i0 = (uint64_t)c * chunk, i1 = i0 + chunk < n_total ? i0 + chunk : n_total;
      uint64_t w = cnt[c], found = 0;
      for (uint64_t i = i0; i < i1; ++i) {
        const float x = u24(hash4(key, i, 0));
        int hit = 0;
        for (int b = 0; b < nbox; ++b) hit |= (x >= lo[3 * b] && x < hi[3 * b]);
        if (!hit) continue;
        const float y = u24(hash4(key, i, 1)), z = u24(hash4(key, i, 2));
        for (int b = 0; b < nbox; ++b) {
          if (x >= lo[3 * b] && x < hi[3 * b] && y >= lo[3 * b + 1] && y < hi[3 * b + 1] && z >= lo[3 * b + 2] && z < hi[3 * b + 2]) {
            if (pass == 1 && w < cap) { out_xyz[3 * w] = x; out_xyz[3 * w + 1] = y; out_xyz[3 * w + 2] = z; out_idx[w] = (uint32_t)i; out_box[w] = b; }
            ++w; ++found;
            break;                                  /* boxes may overlap: the first one claims the point */
          }
        }
      }
      if (pass == 0) cnt[c] = found;
    }
    if (pass == 0) {
      uint64_t run = 0;
      for (uint64_t c = 0; c <= nchunks; ++c) { const uint64_t t = c < nchunks ? cnt[c] : 0; cnt[c] = run; run += t; }
    }
  }
  const int64_t total = (int64_t)cnt[nchunks];
  free(cnt);
  return total;
}

/* colour = bytes 0,1,2 of h(seed,2,i,0); rgb is interleaved [n][3] */
void pto_synth_rgb(uint64_t seed, uint64_t i0, uint64_t n, uint8_t* rgb) {
  const uint64_t key = stream_key(seed, 2);
  for (uint64_t j = 0; j < n; ++j) {
    const uint64_t h = hash4(key, i0 + j, 0);
    rgb[3 * j + 0] = (uint8_t)(h & 0xFF);
    rgb[3 * j + 1] = (uint8_t)((h >> 8) & 0xFF);
    rgb[3 * j + 2] = (uint8_t)((h >> 16) & 0xFF);
  }
}
/* unit normal, f32: normalize(2*u24-1); (0,0,1) if |n| < 1e-12; nrm interleaved [n][3] */
void pto_synth_nrm(uint64_t seed, uint64_t i0, uint64_t n, float* nrm) {
  const uint64_t key = stream_key(seed, 3);
  for (uint64_t j = 0; j < n; ++j) {
    const uint64_t i = i0 + j;
    float a = 2.0f * u24(hash4(key, i, 0)) - 1.0f;
    float b = 2.0f * u24(hash4(key, i, 1)) - 1.0f;
    float c = 2.0f * u24(hash4(key, i, 2)) - 1.0f;
    float len = sqrtf((a * a + b * b) + c * c);
    if (len < 1e-12f) { a = 0.f; b = 0.f; c = 1.f; }
    else { a = a / len; b = b / len; c = c / len; }
    nrm[3 * j + 0] = a; nrm[3 * j + 1] = b; nrm[3 * j + 2] = c;
  }
}

/* ------------------------------------------------------------------------- */
/* Distance functor restatement (reference src/Distance.h)                    */
/* ------------------------------------------------------------------------- */
/* src/Distance.h:6-11 -- squared Euclidean, (dx*dx + dy*dy) + dz*dz */
static inline double td3(double px, double py, double pz, double qx, double qy, double qz) {
  const double dx = px - qx, dy = py - qy, dz = pz - qz;
  return dx * dx + dy * dy + dz * dz;
}
double pto_transformed_distance(const double* p, const double* q) {
  return td3(p[0], p[1], p[2], q[0], q[1], q[2]);
}
/* src/Distance.h:27-57 -- point -> AABB squared lower bound, per-axis offsets in dists
 * (the 3-arg overload; the 2-arg overload at :13-25 has a typo at :20 and is dead code
 * on the search path -- SURVEY.md 3.2 -- so the CORRECT formula is restated). */
double pto_min_distance_to_rectangle(const double* p, const double* lo, const double* hi, double* dists) {
  double distance = 0.0;
  for (int a = 0; a < 3; ++a) {
    const double h = p[a];
    if (h < lo[a]) { const double d = lo[a] - h; if (dists) dists[a] = d; distance += d * d; }
    if (h > hi[a]) { const double d = h - hi[a]; if (dists) dists[a] = d; distance += d * d; }
  }
  return distance;
}
/* src/Distance.h:60-90 -- point -> AABB squared upper bound */
double pto_max_distance_to_rectangle(const double* p, const double* lo, const double* hi, double* dists) {
  double d[3];
  for (int a = 0; a < 3; ++a) {
    const double h = p[a];
    d[a] = (h >= (lo[a] + hi[a]) / 2.0) ? (h - lo[a]) : (hi[a] - h);
    if (dists) dists[a] = d[a];
  }
  return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}
/* src/Distance.h:92-95 */
double pto_new_distance(double dist, double old_off, double new_off) {
  return dist + new_off * new_off - old_off * old_off;
}
/* src/Distance.h:97,99 */
double pto_transformed_distance_scalar(double d) { return d * d; }
double pto_inverse_of_transformed_distance(double d) { return sqrt(d); }

/* ------------------------------------------------------------------------- */
/* Total order on candidates: (d2 as IEEE double, original index) ascending   */
/* (a refinement of CGAL's unspecified tie order)       */
/* ------------------------------------------------------------------------- */
typedef struct { double d; uint32_t i; } cand_t;
static inline int cand_lt(double ad, uint32_t ai, double bd, uint32_t bi) {
  return (ad < bd) || (ad == bd && ai < bi);
}
/* bounded max-heap of k under the total order; heap[0] is the current worst */
static inline void heap_sift_down(cand_t* h, int n, int i) {
  for (;;) {
    int l = 2 * i + 1, r = l + 1, m = i;
    if (l < n && cand_lt(h[m].d, h[m].i, h[l].d, h[l].i)) m = l;
    if (r < n && cand_lt(h[m].d, h[m].i, h[r].d, h[r].i)) m = r;
    if (m == i) return;
    cand_t t = h[i]; h[i] = h[m]; h[m] = t; i = m;
  }
}
static inline void heap_sift_up(cand_t* h, int i) {
  while (i > 0) {
    int p = (i - 1) / 2;
    if (!cand_lt(h[p].d, h[p].i, h[i].d, h[i].i)) return;
    cand_t t = h[i]; h[i] = h[p]; h[p] = t; i = p;
  }
}
static inline void heap_offer(cand_t* h, int* n, int k, double d, uint32_t idx) {
  if (*n < k) { h[*n].d = d; h[*n].i = idx; heap_sift_up(h, (*n)++); }
  else if (cand_lt(d, idx, h[0].d, h[0].i)) { h[0].d = d; h[0].i = idx; heap_sift_down(h, k, 0); }
}
static int cand_cmp(const void* a, const void* b) {
  const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
  if (cand_lt(x->d, x->i, y->d, y->i)) return -1;
  if (cand_lt(y->d, y->i, x->d, x->i)) return 1;
  return 0;
}
static void heap_emit(cand_t* h, int n, int k, uint32_t* idx, double* d2) {
  qsort(h, (size_t)n, sizeof(cand_t), cand_cmp);   /* results ascending: pointsTransfer.cpp:475 iterates sorted */
  for (int j = 0; j < k; ++j) {
    idx[j] = j < n ? h[j].i : PTO_NOIDX;
    if (d2) d2[j] = j < n ? h[j].d : INFINITY;
  }
}

/* ------------------------------------------------------------------------- */
/* (i) brute-force O(N*M) k-NN: the ground truth                              */
/*     query semantics: pointsTransfer.cpp:474-478 (exact K nearest, eps 0)   */
/* ------------------------------------------------------------------------- */
int pto_knn_bruteforce(const double* src, uint64_t n, const uint32_t* gidx /* may be NULL */,
                       const double* tgt, uint64_t m, int k, uint32_t* idx, double* d2, int nthreads) {
  if (k <= 0) return -1;
  const double *sx = src, *sy = src + n, *sz = src + 2 * n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    cand_t* h = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t t = 0; t < (int64_t)m; ++t) {
      const double qx = tgt[t], qy = tgt[m + t], qz = tgt[2 * m + t];
      int cnt = 0;
      for (uint64_t i = 0; i < n; ++i) {
        const double d = td3(qx, qy, qz, sx[i], sy[i], sz[i]);
        heap_offer(h, &cnt, k, d, gidx ? gidx[i] : (uint32_t)i);
      }
      heap_emit(h, cnt, k, idx + (size_t)t * k, d2 ? d2 + (size_t)t * k : NULL);
    }
    free(h);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* (ii) kd-tree restatement of the reference's search = timed CPU baseline    */
/*   tree:  pointsTransfer.cpp:37-40,259  (Kd_tree, Sliding_midpoint, bucket  */
/*          size 10 -- SURVEY.md Appendix B [upstream])                        */
/*   query: pointsTransfer.cpp:474-478 + SURVEY.md 3.2: root bound            */
/*          Distance.h:27-57, incremental bound Distance.h:92-95, leaf metric */
/*          Distance.h:6-11.  Far-subtree pruning uses `new_rd > worst` (not   */
/*          CGAL's `>=`) so that equal-distance, lower-index points are found, */
/*          and takes the incremental bound with a 1e-10 relative margin: the  */
/*          running sum rd + new^2 - old^2 is rounded at every level and can   */
/*          end an ulp ABOVE the exact bound, which on lattice data (exact     */
/*          ties) pruned subtrees holding a tied, lower-index point (found by  */
/*          tests/test_gpu_stress.py against the brute force (i); CGAL's own   */
/*          search has the same rounding, its tie order is unspecified).  The  */
/*          result is exactly the (d2, idx) total order of (i).               */
/* ------------------------------------------------------------------------- */
#define PTO_BUCKET 10
typedef struct {
  int32_t left, right;     /* children (internal) or -1 */
  int32_t cutdim;          /* -1 for leaf */
  uint32_t begin, end;     /* leaf: point range */
  double cutval, lower_high, upper_low;
} kdnode_t;
typedef struct {
  uint64_t n;
  double* pts;             /* AoS [n][3], permuted into leaf order */
  uint32_t* ids;           /* original (global) index per permuted point */
  kdnode_t* nodes; int32_t n_nodes, cap_nodes;
  double lo[3], hi[3];     /* root bounding box */
} kdtree_t;

static int32_t kd_new_node(kdtree_t* t) {
  if (t->n_nodes == t->cap_nodes) {
    t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
    t->nodes = (kdnode_t*)realloc(t->nodes, sizeof(kdnode_t) * (size_t)t->cap_nodes);
  }
  return t->n_nodes++;
}
static void kd_swap(kdtree_t* t, uint64_t a, uint64_t b) {
  if (a == b) return;
  double tmp[3]; memcpy(tmp, t->pts + 3 * a, 24); memcpy(t->pts + 3 * a, t->pts + 3 * b, 24); memcpy(t->pts + 3 * b, tmp, 24);
  uint32_t ti = t->ids[a]; t->ids[a] = t->ids[b]; t->ids[b] = ti;
}
/* node rectangle [lo,hi]; points [b,e) */
static int32_t kd_build_rec(kdtree_t* t, uint64_t b, uint64_t e, double* lo, double* hi) {
  const int32_t me = kd_new_node(t);
  if (e - b <= PTO_BUCKET) {
    kdnode_t* nd = &t->nodes[me]; nd->left = nd->right = -1; nd->cutdim = -1; nd->begin = (uint32_t)b; nd->end = (uint32_t)e;
    return me;
  }
  /* tight box of the points */
  double tlo[3] = {INFINITY, INFINITY, INFINITY}, thi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint64_t i = b; i < e; ++i) for (int a = 0; a < 3; ++a) {
    const double v = t->pts[3 * i + a]; if (v < tlo[a]) tlo[a] = v; if (v > thi[a]) thi[a] = v; }
  /* sliding midpoint: cut the longest side of the node rectangle at its midpoint ... */
  int cd = 0; double span = hi[0] - lo[0];
  for (int a = 1; a < 3; ++a) if (hi[a] - lo[a] > span) { span = hi[a] - lo[a]; cd = a; }
  if (!(thi[cd] > tlo[cd])) {            /* all points equal along cd: use the longest TIGHT side */
    cd = 0; span = thi[0] - tlo[0];
    for (int a = 1; a < 3; ++a) if (thi[a] - tlo[a] > span) { span = thi[a] - tlo[a]; cd = a; }
  }
  if (!(thi[cd] > tlo[cd])) {            /* all points identical: oversized leaf */
    kdnode_t* nd = &t->nodes[me]; nd->left = nd->right = -1; nd->cutdim = -1; nd->begin = (uint32_t)b; nd->end = (uint32_t)e;
    return me;
  }
  double cv = (lo[cd] + hi[cd]) / 2.0;
  /* ... and slide it to the nearest point if one side would be empty */
  if (cv >= thi[cd]) cv = thi[cd]; else if (cv < tlo[cd]) cv = tlo[cd];
  /* partition: lower = {p[cd] < cv}, upper = {p[cd] >= cv}; if cv == max then lower = {p[cd] < cv} is
   * non-empty because tlo < thi; if lower would be empty (cv == tlo) move points equal to cv down */
  uint64_t i = b, j = e;
  while (i < j) { if (t->pts[3 * i + cd] < cv) ++i; else kd_swap(t, i, --j); }
  if (i == b) { /* nothing strictly below: put the points equal to the minimum in the lower child */
    j = e;
    while (i < j) { if (t->pts[3 * i + cd] <= cv) ++i; else kd_swap(t, i, --j); }
  }
  const uint64_t mid = i;
  double lh = -INFINITY, ul = INFINITY;
  for (uint64_t q = b; q < mid; ++q) if (t->pts[3 * q + cd] > lh) lh = t->pts[3 * q + cd];
  for (uint64_t q = mid; q < e; ++q) if (t->pts[3 * q + cd] < ul) ul = t->pts[3 * q + cd];
  double save = hi[cd]; hi[cd] = cv;
  const int32_t l = kd_build_rec(t, b, mid, lo, hi);
  hi[cd] = save; save = lo[cd]; lo[cd] = cv;
  const int32_t r = kd_build_rec(t, mid, e, lo, hi);
  lo[cd] = save;
  kdnode_t* nd = &t->nodes[me];
  nd->left = l; nd->right = r; nd->cutdim = cd; nd->cutval = cv; nd->lower_high = lh; nd->upper_low = ul; nd->begin = nd->end = 0;
  return me;
}

void* pto_kdtree_build(const double* src, uint64_t n, const uint32_t* gidx /* may be NULL */) {
  kdtree_t* t = (kdtree_t*)calloc(1, sizeof(kdtree_t));
  t->n = n;
  t->pts = (double*)malloc(sizeof(double) * 3 * (n ? n : 1));
  t->ids = (uint32_t*)malloc(sizeof(uint32_t) * (n ? n : 1));
  for (int a = 0; a < 3; ++a) { t->lo[a] = INFINITY; t->hi[a] = -INFINITY; }
  for (uint64_t i = 0; i < n; ++i) {
    for (int a = 0; a < 3; ++a) {
      const double v = src[(uint64_t)a * n + i]; t->pts[3 * i + a] = v;
      if (v < t->lo[a]) t->lo[a] = v; if (v > t->hi[a]) t->hi[a] = v;
    }
    t->ids[i] = gidx ? gidx[i] : (uint32_t)i;
  }
  if (n) { double lo[3], hi[3]; memcpy(lo, t->lo, 24); memcpy(hi, t->hi, 24); kd_build_rec(t, 0, n, lo, hi); }
  return t;
}
void pto_kdtree_free(void* h) {
  kdtree_t* t = (kdtree_t*)h; if (!t) return;
  free(t->pts); free(t->ids); free(t->nodes); free(t);
}

typedef struct { const kdtree_t* t; double q[3]; double dists[3]; cand_t* heap; int cnt, k; } kdsearch_t;

static void kd_search_rec(kdsearch_t* s, int32_t ni, double rd) {
  const kdnode_t* nd = &s->t->nodes[ni];
  if (nd->cutdim < 0) {
    for (uint32_t i = nd->begin; i < nd->end; ++i) {
      const double* p = s->t->pts + 3 * (uint64_t)i;
      const double d = td3(s->q[0], s->q[1], s->q[2], p[0], p[1], p[2]);      /* Distance.h:6-11 */
      heap_offer(s->heap, &s->cnt, s->k, d, s->t->ids[i]);
    }
    return;
  }
  const int cd = nd->cutdim;
  const double val = s->q[cd];
  const double diff1 = val - nd->upper_low, diff2 = val - nd->lower_high;
  int32_t nearc, farc; double new_off;
  if (diff1 + diff2 < 0) { nearc = nd->left; farc = nd->right; new_off = diff1; }
  else { nearc = nd->right; farc = nd->left; new_off = diff2; }
  kd_search_rec(s, nearc, rd);
  const double old_off = s->dists[cd];
  const double new_rd = rd + new_off * new_off - old_off * old_off;           /* Distance.h:92-95 */
  if (s->cnt < s->k || !(new_rd * (1.0 - 1e-10) > s->heap[0].d)) {            /* visit on <= (with rounding margin): total order */
    s->dists[cd] = new_off;
    kd_search_rec(s, farc, new_rd);
    s->dists[cd] = old_off;
  }
}

int pto_kdtree_query(void* h, const double* tgt, uint64_t m, int k, uint32_t* idx, double* d2, int nthreads) {
  const kdtree_t* t = (const kdtree_t*)h;
  if (!t || k <= 0) return -1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
  {
    kdsearch_t s; s.t = t; s.k = k; s.heap = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (int64_t q = 0; q < (int64_t)m; ++q) {
      s.q[0] = tgt[q]; s.q[1] = tgt[m + q]; s.q[2] = tgt[2 * m + q];
      s.cnt = 0;
      if (t->n) {
        s.dists[0] = s.dists[1] = s.dists[2] = 0.0;
        const double rd = pto_min_distance_to_rectangle(s.q, t->lo, t->hi, s.dists);  /* Distance.h:27-57 */
        kd_search_rec(&s, 0, rd);
      }
      heap_emit(s.heap, s.cnt, k, idx + (size_t)q * k, d2 ? d2 + (size_t)q * k : NULL);
    }
    free(s.heap);
  }
  return 0;
}
int pto_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* G-way candidate merge under (d2, idx) -- the multi-GPU merge's reference   */
/* lists: [g][m][k] ; entries with idx == PTO_NOIDX are empty                 */
/* ------------------------------------------------------------------------- */
int pto_merge_candidates(const uint32_t* idx_lists, const double* d2_lists, int g, uint64_t m, int k,
                         uint32_t* idx, double* d2) {
  cand_t* buf = (cand_t*)malloc(sizeof(cand_t) * (size_t)g * (size_t)k);
  for (uint64_t t = 0; t < m; ++t) {
    int c = 0;
    for (int r = 0; r < g; ++r) for (int j = 0; j < k; ++j) {
      const size_t o = ((size_t)r * m + t) * (size_t)k + (size_t)j;
      if (idx_lists[o] != PTO_NOIDX) { buf[c].d = d2_lists[o]; buf[c].i = idx_lists[o]; ++c; }
    }
    qsort(buf, (size_t)c, sizeof(cand_t), cand_cmp);
    /* the same point can only come from one slab, but be defensive: drop exact duplicates */
    int w = 0;
    for (int j = 0; j < c; ++j) if (w == 0 || buf[j].i != buf[w - 1].i || buf[j].d != buf[w - 1].d) buf[w++] = buf[j];
    for (int j = 0; j < k; ++j) { idx[t * k + j] = j < w ? buf[j].i : PTO_NOIDX; d2[t * k + j] = j < w ? buf[j].d : INFINITY; }
  }
  free(buf);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Attribute blend (BUILD-DEFINED: the reference has no per-vertex k-blend;    */
/* its only blend arithmetic is the 3-point barycentric colour mix at          */
/* pointsTransfer.cpp:95-97 -- weights * colours summed as double products     */
/* into float).  Same shape here: out = sum_j w_j * a_j, k neighbours.         */
/*   mode 0: w_j = 1/k_eff (uniform mean)                                      */
/*   mode 1: w_j = (1/(d2_j + 1e-12)) / sum (inverse squared distance)         */
/* colours: float 0..255 (not truncated); normals: blended then renormalised   */
/* (left as is if length < 1e-12).  rgb: u8 [n][3]; nrm: f32 [n][3].           */
/* ------------------------------------------------------------------------- */
int pto_blend(const uint32_t* idx, const double* d2, uint64_t m, int k, int mode,
              const uint8_t* rgb, const float* nrm, float* rgb_out, float* nrm_out) {
  for (uint64_t t = 0; t < m; ++t) {
    double w[64]; double wsum = 0.0; int ke = 0;
    if (k > 64) return -1;
    for (int j = 0; j < k; ++j) {
      if (idx[t * k + j] == PTO_NOIDX) { w[j] = 0.0; continue; }
      ++ke;
      w[j] = (mode == 1) ? 1.0 / (d2[t * k + j] + 1e-12) : 1.0;
      wsum += w[j];
    }
    double c[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
    for (int j = 0; j < k; ++j) {
      const uint32_t id = idx[t * k + j];
      if (id == PTO_NOIDX) continue;
      const double wj = w[j] / wsum;
      for (int a = 0; a < 3; ++a) {
        if (rgb) c[a] += wj * (double)rgb[3 * (size_t)id + a];
        if (nrm) nn[a] += wj * (double)nrm[3 * (size_t)id + a];
      }
    }
    if (rgb_out) for (int a = 0; a < 3; ++a) rgb_out[3 * t + a] = ke ? (float)c[a] : 0.f;
    if (nrm_out) {
      const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
      if (len >= 1e-12) { nn[0] /= len; nn[1] /= len; nn[2] /= len; }
      for (int a = 0; a < 3; ++a) nrm_out[3 * t + a] = ke ? (float)nn[a] : 0.f;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* The reference's own blend, restated: pointsTransfer.cpp:95-97               */
/*   float r = bc0 * c0.r() + bc1 * c1.r() + bc2 * c2.r();                     */
/* (double weights times int colours, summed left to right in double, stored  */
/* to a float; :100-102 then assign that float to an unsigned char).  Here for  */
/* k terms with caller-given weights w[m][k]; no normalisation; entries with   */
/* idx == PTO_NOIDX are skipped.  Normals get the same arithmetic (an           */
/* extension: the reference mixes colours only).                                */
/* ------------------------------------------------------------------------- */
int pto_blend_weighted(const uint32_t* idx, const double* w, uint64_t m, int k, const uint8_t* rgb, const float* nrm,
                       float* rgb_out, float* nrm_out) {
  for (uint64_t t = 0; t < m; ++t) {
    double c[3] = {0, 0, 0}, nn[3] = {0, 0, 0};
    int first = 1;
    for (int j = 0; j < k; ++j) {
      const uint32_t id = idx[t * k + j];
      if (id == PTO_NOIDX) continue;
      const double wj = w[t * k + j];
      for (int a = 0; a < 3; ++a) {
        const double pc = rgb ? wj * (double)rgb[3 * (size_t)id + a] : 0.0, pn = nrm ? wj * (double)nrm[3 * (size_t)id + a] : 0.0;
        c[a] = first ? pc : c[a] + pc;            /* p0 + p1 + p2: the first term is not added to a zero */
        nn[a] = first ? pn : nn[a] + pn;
      }
      first = 0;
    }
    if (rgb_out) for (int a = 0; a < 3; ++a) rgb_out[3 * t + a] = (float)c[a];
    if (nrm_out) for (int a = 0; a < 3; ++a) nrm_out[3 * t + a] = (float)nn[a];
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* PCA normal from the k neighbours (BUILD-DEFINED, BASELINE config 3; no      */
/* reference counterpart): eigenvector of the smallest eigenvalue of the       */
/* neighbours' 3x3 covariance (double, Jacobi), sign-oriented so that          */
/* dot(n, mean of the neighbours' stored normals) >= 0 (or +z if nrm is NULL). */
/* `planarity` (may be NULL) receives lambda_min / (lambda_0+lambda_1+lambda_2) */
/* so tests can skip ill-conditioned neighbourhoods.                           */
/* ------------------------------------------------------------------------- */
static void jacobi3(double a[3][3], double v[3][3], double ev[3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) v[i][j] = (i == j);
  for (int sweep = 0; sweep < 64; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    if (off <= 1e-20 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;   /* converged far below fp64 resolution */
    for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
      if (fabs(a[p][q]) < 1e-300) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
      const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
      for (int r = 0; r < 3; ++r) { const double arp = a[r][p], arq = a[r][q]; a[r][p] = c * arp - s * arq; a[r][q] = s * arp + c * arq; }
      for (int r = 0; r < 3; ++r) { const double apr = a[p][r], aqr = a[q][r]; a[p][r] = c * apr - s * aqr; a[q][r] = s * apr + c * aqr; }
      for (int r = 0; r < 3; ++r) { const double vrp = v[r][p], vrq = v[r][q]; v[r][p] = c * vrp - s * vrq; v[r][q] = s * vrp + c * vrq; }
    }
  }
  for (int i = 0; i < 3; ++i) ev[i] = a[i][i];
}
int pto_pca_normals(const uint32_t* idx, uint64_t m, int k, const double* src, uint64_t n,
                    const float* nrm /* may be NULL */, float* nrm_out, double* planarity /* may be NULL */) {
  const double *sx = src, *sy = src + n, *sz = src + 2 * n;
  for (uint64_t t = 0; t < m; ++t) {
    double mu[3] = {0, 0, 0}, mn[3] = {0, 0, 0}; int ke = 0;
    for (int j = 0; j < k; ++j) {
      const uint32_t id = idx[t * k + j]; if (id == PTO_NOIDX) continue;
      mu[0] += sx[id]; mu[1] += sy[id]; mu[2] += sz[id]; ++ke;
      if (nrm) for (int a = 0; a < 3; ++a) mn[a] += (double)nrm[3 * (size_t)id + a];
    }
    if (ke < 3) { nrm_out[3 * t] = 0.f; nrm_out[3 * t + 1] = 0.f; nrm_out[3 * t + 2] = 1.f; if (planarity) planarity[t] = 1.0; continue; }
    for (int a = 0; a < 3; ++a) mu[a] /= ke;
    double cv[3][3] = {{0}};
    for (int j = 0; j < k; ++j) {
      const uint32_t id = idx[t * k + j]; if (id == PTO_NOIDX) continue;
      const double d[3] = {sx[id] - mu[0], sy[id] - mu[1], sz[id] - mu[2]};
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cv[a][b] += d[a] * d[b];
    }
    double v[3][3], ev[3]; jacobi3(cv, v, ev);
    int mi = 0; for (int a = 1; a < 3; ++a) if (ev[a] < ev[mi]) mi = a;
    double nn[3] = {v[0][mi], v[1][mi], v[2][mi]};
    const double len = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
    for (int a = 0; a < 3; ++a) nn[a] /= len;
    const double ref = nrm ? (nn[0] * mn[0] + nn[1] * mn[1] + nn[2] * mn[2]) : nn[2];
    if (ref < 0) for (int a = 0; a < 3; ++a) nn[a] = -nn[a];
    for (int a = 0; a < 3; ++a) nrm_out[3 * t + a] = (float)nn[a];
    if (planarity) { const double s = ev[0] + ev[1] + ev[2]; planarity[t] = s > 0 ? ev[mi] / s : 1.0; }
  }
  return 0;
}

/* ========================================================================= */
/* Per-face texture bake and edge padding (f3): CPU restatement  */
/* of reference src/pointsTransfer.cpp:462-581 (face loop), :66-107              */
/* (draw_triangle) and :593-615 (dilate / alpha mask / add).                      */
/*                                                                               */
/* PARITY: unpinned.  The reference delegates the plane frame (Plane_3::to_2d),   */
/* the barycentric coordinates and the per-face Delaunay triangulation to CGAL   */
/* and the dilation to OpenCV; neither library is in this image and the           */
/* reference holds no fixture for them.  What the reference's own lines fix is     */
/* restated literally (the data flow, the inside test bc >= 0, the UV              */
/* interpolation :569-574, the rasteriser's pixel loop, colour mix and            */
/* (resolution - j, i) addressing :68-103, the 25x25 dilate + ~alpha mask + add    */
/* :593-611); where it leans on library behaviour or on undefined behaviour the    */
/* BUILD defines the result, and this file is that definition:                     */
/*  - the 3 corners' neighbour lists are united and de-duplicated by ORIGINAL       */
/*    INDEX, ascending (the reference's std::set with a non-strict-weak            */
/*    comparator, Point.h:94-102, dedupes only partially and is not reproducible);  */
/*  - plane frame: orthonormal (e1 along corner0->corner1, e2 = n x e1), origin at  */
/*    corner 0; any affine frame gives the same barycentrics, and an orthonormal    */
/*    one makes the Delaunay triangulation the true in-plane one;                    */
/*  - a neighbour whose 2-D image coincides with an earlier point's is dropped;      */
/*  - Delaunay = every non-degenerate triple (i<j<k) whose circumcircle holds no     */
/*    other point strictly inside, predicates in plain fp64 evaluated on the          */
/*    index-SORTED tuple with the permutation's sign applied (so the two diagonals    */
/*    of a quad never both fail); co-circular points yield overlapping triangles,      */
/*    which is harmless because of the next rule;                                    */
/*  - a pixel covered by several triangles takes the LAST one in (face, triangle)     */
/*    order -- what the reference's single-threaded loop produces;                   */
/*  - writes outside the texture (row resolution - j for j = 0, negative indices;     */
/*    undefined behaviour in the reference) are skipped; the texture starts as zeros   */
/*    (the reference's cv::Mat is uninitialised); colours are clamped to 0..255.       */
/* All arithmetic fp64, every operation individually rounded (-ffp-contract=off).  */
/* ========================================================================= */
typedef struct { double x, y; } v2;
static inline double cross2(double ax, double ay, double bx, double by) { return ax * by - ay * bx; }
/* barycentric coordinates of X in (v0,v1,v2), A = cross(v1-v0, v2-v0) != 0 */
static inline void bary2(v2 X, v2 v0, v2 v1, v2 v2_, double A, double* b) {
  b[0] = cross2(v1.x - X.x, v1.y - X.y, v2_.x - X.x, v2_.y - X.y) / A;
  b[1] = cross2(v2_.x - X.x, v2_.y - X.y, v0.x - X.x, v0.y - X.y) / A;
  b[2] = (1.0 - b[0]) - b[1];
}
static inline int finite_d(double v) { return v == v && v - v == 0.0; }
/* reference draw_triangle (:66-107) on a zero-initialised BGRA image; see the rules above */
static void pto_draw_triangle(const double* U, const double* V, const uint8_t (*col)[3], int R, uint8_t* bgra) {
  v2 p = {U[0] * R, V[0] * R}, q = {U[1] * R, V[1] * R}, r = {U[2] * R, V[2] * R};
  if (!(finite_d(p.x) && finite_d(p.y) && finite_d(q.x) && finite_d(q.y) && finite_d(r.x) && finite_d(r.y))) return;
  const double A = cross2(q.x - p.x, q.y - p.y, r.x - p.x, r.y - p.y);
  if (!(A != 0.0) || !finite_d(A)) return;
  double xmin = fmin(p.x, fmin(q.x, r.x)), xmax = fmax(p.x, fmax(q.x, r.x));
  double ymin = fmin(p.y, fmin(q.y, r.y)), ymax = fmax(p.y, fmax(q.y, r.y));
  double fi0 = floor(xmin), fi1 = floor(xmax), fj0 = floor(ymin), fj1 = floor(ymax);
  /* only pixels that land inside the texture: col i in [0, R), row R - j in [0, R) <=> j in [1, R] */
  const int i0 = (int)fmax(fi0, 0.0), i1 = (int)fmin(fi1, (double)(R - 1));
  const int j0 = (int)fmax(fj0, 1.0), j1 = (int)fmin(fj1, (double)R);
  for (int i = i0; i <= i1; ++i)
    for (int j = j0; j <= j1; ++j) {
      const int x = i >= R ? R - 1 : i, y = j >= R ? R - 1 : j;          /* :80-82 */
      v2 X = {(double)x, (double)y};
      double b[3];
      bary2(X, p, q, r, A, b);
      if (b[0] >= 0 && b[1] >= 0 && b[2] >= 0) {
        uint8_t* px = bgra + ((size_t)(R - j) * (size_t)R + (size_t)i) * 4;
        for (int c = 0; c < 3; ++c) {                                     /* :95-97: double products, summed, stored to a float */
          const float f = (float)((b[0] * (double)col[0][c] + b[1] * (double)col[1][c]) + b[2] * (double)col[2][c]);
          const float g = f < 0.f ? 0.f : (f > 255.f ? 255.f : f);
          px[2 - c] = (uint8_t)g;                                         /* B, G, R <- colour b, g, r (:100-102), truncated */
        }
        px[3] = 255;
      }
    }
}
#define PTO_BAKE_MAXPTS 99            /* 3 corners + 3 * 32 neighbours */
static int cmp_u32(const void* a, const void* b) { const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b; return x < y ? -1 : (x > y); }
/* lifted 4-point determinant on the index-sorted tuple (w<x<y<z): > 0 iff z is inside the circle through w, x, y when those are
 * counter-clockwise */
static inline double incircle_sorted(const v2* P, int w, int x, int y, int z) {
  const double adx = P[w].x - P[z].x, ady = P[w].y - P[z].y, bdx = P[x].x - P[z].x, bdy = P[x].y - P[z].y, cdx = P[y].x - P[z].x, cdy = P[y].y - P[z].y;
  const double al = adx * adx + ady * ady, bl = bdx * bdx + bdy * bdy, cl = cdx * cdx + cdy * cdy;
  return (al * (bdx * cdy - bdy * cdx) - bl * (adx * cdy - ady * cdx)) + cl * (adx * bdy - ady * bdx);
}
/* is point l strictly inside the circumcircle of the triangle (i<j<k) whose orientation sign is `os` (+1 ccw, -1 cw)? */
static inline int in_circumcircle(const v2* P, int i, int j, int k, int l, int os) {
  double d; int par;
  if (l > k) { d = incircle_sorted(P, i, j, k, l); par = 1; }
  else if (l > j) { d = incircle_sorted(P, i, j, l, k); par = -1; }
  else if (l > i) { d = incircle_sorted(P, i, l, j, k); par = 1; }
  else { d = incircle_sorted(P, l, i, j, k); par = -1; }
  return (double)(os * par) * d > 0.0;
}
int pto_bake_texture(const double* src_xyz, const uint8_t* src_rgb, uint64_t n, const double* vert_xyz, const double* vert_uv,
                     const uint8_t* vert_rgb, uint64_t nv, const int32_t* faces, uint64_t nf, const uint32_t* nbr_idx, int k,
                     int R, uint8_t* bgra) {
  if (k < 1 || k > 32 || R < 1) return -1;
  for (uint64_t f = 0; f < nf; ++f) {
    const int32_t* fv = faces + 3 * f;
    if (fv[0] < 0 || fv[1] < 0 || fv[2] < 0 || (uint64_t)fv[0] >= nv || (uint64_t)fv[1] >= nv || (uint64_t)fv[2] >= nv) continue;   /* malformed face: skipped */
    double cu[3], cv[3], c3[3][3];
    uint8_t ccol[3][3];
    for (int c = 0; c < 3; ++c) {
      cu[c] = vert_uv[2 * (size_t)fv[c]]; cv[c] = vert_uv[2 * (size_t)fv[c] + 1];
      for (int a = 0; a < 3; ++a) { c3[c][a] = vert_xyz[(size_t)a * nv + (size_t)fv[c]]; ccol[c][a] = vert_rgb[3 * (size_t)fv[c] + a]; }
    }
    /* union of the three corners' neighbour lists by original index, ascending (:470-479) */
    uint32_t ids[96]; int nid = 0;
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < k; ++j) { const uint32_t id = nbr_idx[(size_t)fv[c] * k + j]; if (id != PTO_NOIDX && id < n) ids[nid++] = id; }
    qsort(ids, (size_t)nid, sizeof(uint32_t), cmp_u32);
    { int w = 0; for (int j = 0; j < nid; ++j) if (j == 0 || ids[j] != ids[j - 1]) ids[w++] = ids[j]; nid = w; }
    /* plane frame (:485-494): origin corner 0, e1 along corner0 -> corner1, e2 = n x e1, both unit */
    const double ax = c3[1][0] - c3[0][0], ay = c3[1][1] - c3[0][1], az = c3[1][2] - c3[0][2];
    const double bx = c3[2][0] - c3[0][0], by = c3[2][1] - c3[0][1], bz = c3[2][2] - c3[0][2];
    const double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    const double la = sqrt((ax * ax + ay * ay) + az * az);
    const double e1x = ax / la, e1y = ay / la, e1z = az / la;
    const double tx = ny * e1z - nz * e1y, ty = nz * e1x - nx * e1z, tz = nx * e1y - ny * e1x;
    const double lt = sqrt((tx * tx + ty * ty) + tz * tz);
    const double e2x = tx / lt, e2y = ty / lt, e2z = tz / lt;
    v2 P[PTO_BAKE_MAXPTS];
    double PU[PTO_BAKE_MAXPTS], PV[PTO_BAKE_MAXPTS];
    uint8_t PC[PTO_BAKE_MAXPTS][3];
    int np = 3;
    const int frame_ok = la > 0.0 && lt > 0.0 && finite_d(la) && finite_d(lt);
    P[0].x = 0.0; P[0].y = 0.0;
    P[1].x = (ax * e1x + ay * e1y) + az * e1z; P[1].y = (ax * e2x + ay * e2y) + az * e2z;
    P[2].x = (bx * e1x + by * e1y) + bz * e1z; P[2].y = (bx * e2x + by * e2y) + bz * e2z;
    for (int c = 0; c < 3; ++c) { PU[c] = cu[c]; PV[c] = cv[c]; memcpy(PC[c], ccol[c], 3); }
    const double A = frame_ok ? cross2(P[1].x - P[0].x, P[1].y - P[0].y, P[2].x - P[0].x, P[2].y - P[0].y) : 0.0;
    if (frame_ok && A != 0.0 && finite_d(A)) {
      for (int j = 0; j < nid; ++j) {                                   /* :505-537: project, barycentrics, keep what is inside */
        const size_t id = ids[j];
        const double dx = src_xyz[id] - c3[0][0], dy = src_xyz[n + id] - c3[0][1], dz = src_xyz[2 * n + id] - c3[0][2];
        v2 X = {(dx * e1x + dy * e1y) + dz * e1z, (dx * e2x + dy * e2y) + dz * e2z};
        double b[3];
        bary2(X, P[0], P[1], P[2], A, b);
        if (!(b[0] >= 0 && b[1] >= 0 && b[2] >= 0)) continue;
        int dup = 0;
        for (int q = 0; q < np; ++q) dup |= (P[q].x == X.x && P[q].y == X.y);
        if (dup) continue;
        P[np] = X;
        PU[np] = (b[0] * cu[0] + b[1] * cu[1]) + b[2] * cu[2];          /* :571-572 */
        PV[np] = (b[0] * cv[0] + b[1] * cv[1]) + b[2] * cv[2];
        PC[np][0] = src_rgb[3 * id]; PC[np][1] = src_rgb[3 * id + 1]; PC[np][2] = src_rgb[3 * id + 2];
        ++np;
      }
    }
    if (np == 3) {                                                       /* :540-544 */
      pto_draw_triangle(PU, PV, (const uint8_t(*)[3])PC, R, bgra);
      continue;
    }
    int ntri = 0;                                                        /* at most 255 triangles per face (2 np - 5 <= 193 unless many points are co-circular) */
    for (int i = 0; i < np - 2; ++i)                                     /* :546-581 with the build's Delaunay definition */
      for (int j = i + 1; j < np - 1; ++j)
        for (int kk = j + 1; kk < np; ++kk) {
          const double o = cross2(P[j].x - P[i].x, P[j].y - P[i].y, P[kk].x - P[i].x, P[kk].y - P[i].y);
          if (!(o != 0.0)) continue;
          const int os = o > 0.0 ? 1 : -1;
          int empty = 1;
          for (int l = 0; l < np && empty; ++l) if (l != i && l != j && l != kk && in_circumcircle(P, i, j, kk, l, os)) empty = 0;
          if (!empty || ntri >= 255) continue;
          ++ntri;
          const double tu[3] = {PU[i], PU[j], PU[kk]}, tv[3] = {PV[i], PV[j], PV[kk]};
          uint8_t tc[3][3];
          memcpy(tc[0], PC[i], 3); memcpy(tc[1], PC[j], 3); memcpy(tc[2], PC[kk], 3);
          pto_draw_triangle(tu, tv, (const uint8_t(*)[3])tc, R, bgra);
        }
  }
  return 0;
}
/* reference :593-611: dilate(texture, 25x25 rect), edges = dilated & ~alpha (all four channels), padded = texture + edges
 * (saturating).  OpenCV's dilate ignores what lies outside the image (border value = the channel minimum). */
int pto_dilate_pad(const uint8_t* in, int R, int ksize, uint8_t* out) {
  if (ksize < 1 || !(ksize & 1)) return -1;
  const int h = ksize / 2;
  uint8_t* tmp = (uint8_t*)malloc((size_t)R * R * 4);
  if (!tmp) return -1;
#pragma omp parallel for
  for (int y = 0; y < R; ++y)
    for (int x = 0; x < R; ++x)
      for (int c = 0; c < 4; ++c) {
        uint8_t m = 0;
        for (int d = -h; d <= h; ++d) { const int xx = x + d; if (xx >= 0 && xx < R) { const uint8_t v = in[((size_t)y * R + xx) * 4 + c]; if (v > m) m = v; } }
        tmp[((size_t)y * R + x) * 4 + c] = m;
      }
#pragma omp parallel for
  for (int y = 0; y < R; ++y)
    for (int x = 0; x < R; ++x) {
      const uint8_t mask = (uint8_t)~in[((size_t)y * R + x) * 4 + 3];
      for (int c = 0; c < 4; ++c) {
        uint8_t m = 0;
        for (int d = -h; d <= h; ++d) { const int yy = y + d; if (yy >= 0 && yy < R) { const uint8_t v = tmp[((size_t)yy * R + x) * 4 + c]; if (v > m) m = v; } }
        const int s = (int)in[((size_t)y * R + x) * 4 + c] + (int)(m & mask);
        out[((size_t)y * R + x) * 4 + c] = (uint8_t)(s > 255 ? 255 : s);
      }
    }
  free(tmp);
  return 0;
}
