/*
 * vdb_oracle.c — CPU ORACLE for the cosine top-k path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (vectordb_amd/) never falls back to it: it fails
 * loudly when libvdb_hip.so is missing.
 *
 * PARITY UNPINNED: the reference (billenewman4/VectorDB) holds no golden
 * vectors, known-answer tests or fixtures for this path, its
 * stored vectors are stripped from the checkout and its arithmetic lives in
 * un-vendored third-party code (chromadb >= 0.4.6 -> chroma-hnswlib, no
 * lockfile).  This file restates the *published* algorithm of that dependency
 * for cosine space and the exact brute force BASELINE.json names as the
 * correctness standard:
 *   - chroma-hnswlib cosine space = inner-product space over L2-normalised
 *     vectors, distance = 1 - <q^, x^>; results ascending by distance.
 *     Reference call sites: src/vectordb.py:156-160 ("hnsw:space":"cosine"),
 *     src/vectordb.py:259-264 (query), src/vectordb.py:277 (sim = 1 - d).
 *   - exact top-k: all rows scored, k smallest distances, ties by row index.
 * Storage is fp32, accumulation fp64.
 *
 * The library exports the same vdb_* C ABI as include/vdb.h ("dev" pointers
 * are plain host pointers here) plus ora_* helpers, so the Python face and the
 * world_size-2 gloo tests can run without a GPU *as tests*.
 */
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/vdb.h"

/* ------------------------------------------------------------------ */
/* synthetic generator — bit-identical to csrc/synth.h  */
/* ------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline uint64_t rowkey(uint64_t seed, uint64_t row) {
  return mix64(seed ^ (row * 0xD6E8FEB86659FD93ULL));
}
static inline float uval(uint64_t key, uint32_t col) {
  uint64_t h = mix64(key + (uint64_t)col);
  int32_t t = (int32_t)(h >> 40); /* 24 bits */
  return (float)(t - 8388608) * (1.0f / 8388608.0f);
}
static inline float raw_base(uint64_t seed, int kind, int n_clusters, float noise, int64_t row,
                             uint32_t col) {
  float u = uval(rowkey(seed, (uint64_t)row), col);
  if (kind == 1) {
    int64_t c = row % (int64_t)(n_clusters > 0 ? n_clusters : 1);
    float cen = uval(rowkey(seed ^ 0xC1A55E5ULL, (uint64_t)c), col);
    return fmaf(noise, u, cen);
  }
  return u;
}
static inline float raw_value(const vdb_synth_spec* s, int64_t row, uint32_t col) {
  if (s->kind == 2) {
    uint64_t key = rowkey(s->seed, (uint64_t)row);
    int64_t src = (int64_t)(mix64(key ^ 0x51C0FFEEULL) % (uint64_t)(s->base_n > 0 ? s->base_n : 1));
    float base = raw_base(s->base_seed, s->base_kind, s->base_n_clusters, s->base_noise, src, col);
    return fmaf(s->noise, uval(key, col), base);
  }
  return raw_base(s->seed, s->kind, s->n_clusters, s->noise, row, col);
}

void ora_synth_rows(float* out, int64_t row0, int64_t n, int dim, const vdb_synth_spec* spec) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r)
    for (int c = 0; c < dim; ++c) out[r * dim + c] = raw_value(spec, row0 + r, (uint32_t)c);
}

/* source corpus row of a kind-2 query row (so tests know the planted neighbour) */
int64_t ora_synth_query_src(const vdb_synth_spec* s, int64_t row) {
  uint64_t key = rowkey(s->seed, (uint64_t)row);
  return (int64_t)(mix64(key ^ 0x51C0FFEEULL) % (uint64_t)(s->base_n > 0 ? s->base_n : 1));
}

/* ------------------------------------------------------------------ */
/* numerics                                                            */
/* ------------------------------------------------------------------ */
/* L2 normalisation as cosine space requires (chroma-hnswlib normalises with
 * 1/(sqrt(sum)+1e-30)); sum and scale in fp64, result rounded once to fp32. */
void ora_normalize_rows(const float* in, float* out, int64_t n, int dim) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    double ss = 0.0;
    for (int c = 0; c < dim; ++c) ss += (double)in[r * dim + c] * (double)in[r * dim + c];
    double inv = 1.0 / (sqrt(ss) + 1e-30);
    for (int c = 0; c < dim; ++c) out[r * dim + c] = (float)((double)in[r * dim + c] * inv);
  }
}

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN stays NaN */
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
void ora_round_bf16(const float* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = bf16_to_f32(f32_to_bf16_rne(in[i]));
}

/* candidate ordering: higher score first, then lower index */
typedef struct { double s; int64_t i; } cand_t;
static inline int better(double s, int64_t i, const cand_t* c) {
  return (s > c->s) || (s == c->s && i < c->i);
}
/* sorted insertion list of capacity k (cnt entries valid) */
static inline void list_push(cand_t* L, int* cnt, int k, double s, int64_t i) {
  if (s != s) return; /* NaN scores are never returned */
  if (*cnt == k && !better(s, i, &L[k - 1])) return;
  int p = (*cnt < k) ? (*cnt)++ : k - 1;
  while (p > 0 && better(s, i, &L[p - 1])) { L[p] = L[p - 1]; --p; }
  L[p].s = s; L[p].i = i;
}

/* Exact top-k of one normalised fp32 query against rows (fp32, or bf16 bits when
 * rows16 != NULL); fp64 accumulation; optional excluded row; optional eligibility bitmap
 * (bit r of allow set = row r may be returned). Returns count. */
static int scan_one(const float* rows, const uint16_t* rows16, int64_t n, int dim, const float* qn,
                    int k, int64_t exclude, const uint32_t* allow, cand_t* out) {
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
  if (n < 4096) nthreads = 1;
#endif
  cand_t* lists = (cand_t*)malloc(sizeof(cand_t) * (size_t)k * (size_t)nthreads);
  int* cnts = (int*)calloc((size_t)nthreads, sizeof(int));
#pragma omp parallel num_threads(nthreads)
  {
    int t = 0, T = 1;
#ifdef _OPENMP
    t = omp_get_thread_num(); T = omp_get_num_threads();
#endif
    int64_t lo = n * t / T, hi = n * (t + 1) / T;
    cand_t* L = lists + (size_t)t * k;
    int cnt = 0;
    for (int64_t r = lo; r < hi; ++r) {
      if (r == exclude) continue;
      if (allow && !((allow[r >> 5] >> (r & 31)) & 1u)) continue;
      double acc = 0.0;
      if (rows16) {
        const uint16_t* x = rows16 + r * dim;
        for (int c = 0; c < dim; ++c) acc += (double)qn[c] * (double)bf16_to_f32(x[c]);
      } else {
        const float* x = rows + r * dim;
        for (int c = 0; c < dim; ++c) acc += (double)qn[c] * (double)x[c];
      }
      list_push(L, &cnt, k, acc, r);
    }
    cnts[t] = cnt;
  }
  int cnt = 0;
  for (int t = 0; t < nthreads; ++t)
    for (int j = 0; j < cnts[t]; ++j) list_push(out, &cnt, k, lists[(size_t)t * k + j].s, lists[(size_t)t * k + j].i);
  free(lists); free(cnts);
  return cnt;
}

/* Stand-alone exact search over already-normalised fp32 rows (tests call this
 * directly).  queries are raw (normalised here).  out: b x k, -1/+inf padded. */
void ora_search(const float* rows_norm, int64_t n, int dim, const float* queries, int b, int k,
                int64_t* out_idx, float* out_dist, double* out_score64) {
  float* qn = (float*)malloc(sizeof(float) * (size_t)dim);
  cand_t* L = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
  for (int q = 0; q < b; ++q) {
    ora_normalize_rows(queries + (size_t)q * dim, qn, 1, dim);
    int cnt = scan_one(rows_norm, NULL, n, dim, qn, k, -1, NULL, L);
    for (int j = 0; j < k; ++j) {
      out_idx[(size_t)q * k + j] = j < cnt ? L[j].i : -1;
      out_dist[(size_t)q * k + j] = j < cnt ? (float)(1.0 - L[j].s) : INFINITY;
      if (out_score64) out_score64[(size_t)q * k + j] = j < cnt ? L[j].s : -INFINITY;
    }
  }
  free(qn); free(L);
}

/* ------------------------------------------------------------------ */
/* the vdb_* ABI on the CPU                                            */
/* ------------------------------------------------------------------ */
struct vdb {
  int dim, dtype, keep_f32;
  int64_t n, cap, row_offset;
  float* rows;       /* normalised fp32 (dtype f32, or bf16 with keep_f32) */
  uint16_t* rows16;  /* bf16 bits (dtype bf16)                             */
  double last_bytes;
  int n_shards;       /* > 0: made by vdb_create_sharded (a brute force has nothing to shard: one store, same answers) */
  int reserved;       /* vdb_reserve has been called (a sharded handle insists on it before rows are added) */
  float* split_q[2];  /* queries of a pending two-phase search, per slot */
  int split_b[2], split_k[2], split_kp[2];
  char err[256];
};
static char g_err[256];

static int fail(vdb_t* h, int code, const char* msg) {
  snprintf(h ? h->err : g_err, 256, "%s", msg);
  return code;
}

int vdb_create(vdb_t** out, int dim, int dtype, int keep_f32, int device_id) {
  (void)device_id;
  if (!out) return fail(NULL, VDB_ERR_ARG, "vdb_create: out is NULL");
  if (dim <= 0 || dim > 32768) return fail(NULL, VDB_ERR_ARG, "vdb_create: dim must be in [1,32768]");
  if (dtype != VDB_F32 && dtype != VDB_BF16) return fail(NULL, VDB_ERR_ARG, "vdb_create: dtype must be 0 (f32) or 1 (bf16)");
  vdb_t* h = (vdb_t*)calloc(1, sizeof(vdb_t));
  if (!h) return fail(NULL, VDB_ERR_MEMORY, "vdb_create: out of memory");
  h->dim = dim; h->dtype = dtype; h->keep_f32 = (dtype == VDB_F32) ? 1 : (keep_f32 != 0);
  *out = h;
  return VDB_OK;
}
/* The oracle's answer does not depend on how the engine shards its rows: the sharded handle is the plain exact
 * store (that IS the contract of vdb_create_sharded in include/vdb.h); only its calling rules are mirrored. */
int vdb_create_sharded(vdb_t** out, int dim, int dtype, int keep_f32, const int* device_ids, int n_devices) {
  if (!out) return fail(NULL, VDB_ERR_ARG, "vdb_create_sharded: out is NULL");
  if (!device_ids || n_devices < 1 || n_devices > 64) return fail(NULL, VDB_ERR_ARG, "vdb_create_sharded: need 1..64 device ids");
  int rc = vdb_create(out, dim, dtype, keep_f32, 0);
  if (rc) return rc;
  (*out)->n_shards = n_devices;
  return VDB_OK;
}
void vdb_destroy(vdb_t* h) {
  if (!h) return;
  free(h->rows); free(h->rows16); free(h->split_q[0]); free(h->split_q[1]); free(h);
}
const char* vdb_last_error(vdb_t* h) { return h ? h->err : g_err; }

int vdb_reserve(vdb_t* h, int64_t n_rows) {
  if (!h) return VDB_ERR_ARG;
  h->reserved = 1;
  if (n_rows <= h->cap) return VDB_OK;
  if (h->keep_f32) {
    float* p = (float*)realloc(h->rows, sizeof(float) * (size_t)n_rows * h->dim);
    if (!p) return fail(h, VDB_ERR_MEMORY, "vdb_reserve: out of memory");
    h->rows = p;
  }
  if (h->dtype == VDB_BF16) {
    uint16_t* p = (uint16_t*)realloc(h->rows16, sizeof(uint16_t) * (size_t)n_rows * h->dim);
    if (!p) return fail(h, VDB_ERR_MEMORY, "vdb_reserve: out of memory");
    h->rows16 = p;
  }
  h->cap = n_rows;
  return VDB_OK;
}
/* the twin's sharded handle is the plain store: re-partitioning moves nothing, and there are no workspaces to warm */
int vdb_rebalance(vdb_t* h, int64_t total_rows) {
  if (!h) return VDB_ERR_ARG;
  if (total_rows < 0) return fail(h, VDB_ERR_ARG, "rebalance: total_rows must be >= 0");
  return vdb_reserve(h, total_rows > h->n ? total_rows : h->n);
}
int vdb_warm(vdb_t* h, int max_batch, int k, int kp) {
  (void)k; (void)kp;
  if (!h) return VDB_ERR_ARG;
  if (max_batch < 1 || max_batch > (1 << 20)) return fail(h, VDB_ERR_ARG, "warm: max_batch must be in [1, 1048576]");
  return VDB_OK;
}
int vdb_set_row_offset(vdb_t* h, int64_t off) {
  if (!h) return VDB_ERR_ARG;
  h->row_offset = off;
  return VDB_OK;
}

int vdb_add(vdb_t* h, const float* rows, int64_t n) {
  if (!h) return VDB_ERR_ARG;
  if (n < 0 || (n > 0 && !rows)) return fail(h, VDB_ERR_ARG, "vdb_add: bad rows/n");
  if (n == 0) return VDB_OK;
  if (h->n_shards && !h->reserved) return fail(h, VDB_ERR_STATE, "vdb_add: call vdb_reserve(total rows) on a sharded handle first (it fixes the row partition)");
  if (h->n + n > h->cap) {
    int64_t want = h->cap + h->cap / 2;
    if (want < h->n + n) want = h->n + n;
    int rc = vdb_reserve(h, want);
    if (rc) return rc;
  }
  float* tmp = (float*)malloc(sizeof(float) * (size_t)h->dim);
  for (int64_t r = 0; r < n; ++r) {
    ora_normalize_rows(rows + r * h->dim, tmp, 1, h->dim);
    if (h->keep_f32) memcpy(h->rows + (h->n + r) * h->dim, tmp, sizeof(float) * (size_t)h->dim);
    if (h->dtype == VDB_BF16)
      for (int c = 0; c < h->dim; ++c) h->rows16[(h->n + r) * h->dim + c] = f32_to_bf16_rne(tmp[c]);
  }
  free(tmp);
  h->n += n;
  return VDB_OK;
}
int vdb_add_dev(vdb_t* h, const float* rows_dev, int64_t n) { return vdb_add(h, rows_dev, n); }
int vdb_count(vdb_t* h, int64_t* n) {
  if (!h || !n) return VDB_ERR_ARG;
  *n = h->n;
  return VDB_OK;
}

static int check_search_args(vdb_t* h, const void* q, int b, int k, int kp, const void* oi, const void* od) {
  if (b < 0 || (b > 0 && (!q || !oi || !od))) return fail(h, VDB_ERR_ARG, "search: NULL buffer");
  if (k < 1 || k > VDB_MAX_K) return fail(h, VDB_ERR_ARG, "search: k must be in [1,1024]");
  if (kp != 0 && (kp < k || kp > VDB_MAX_K)) return fail(h, VDB_ERR_ARG, "search: rerank_kprime must be 0 or in [k,1024]");
  if (kp != 0 && h->dtype != VDB_BF16) return fail(h, VDB_ERR_STATE, "search: rerank_kprime needs a bf16 store");
  return VDB_OK;
}

/* one query through the store: scan (fp32 or bf16 rows) [+ fp32 rerank] */
static void search_one(vdb_t* h, const float* qn, int k, int kp, int64_t exclude, const uint32_t* allow, int64_t* oi, float* od) {
  int kk = kp ? kp : k;
  cand_t* L = (cand_t*)malloc(sizeof(cand_t) * (size_t)kk);
  int cnt = scan_one(h->dtype == VDB_BF16 ? NULL : h->rows, h->dtype == VDB_BF16 ? h->rows16 : NULL,
                     h->n, h->dim, qn, kk, exclude, allow, L);
  if (kp) { /* exact re-score of the k' candidates against the fp32 copy — or, on a store without one, against the stored
               bf16 rows themselves (the scan's own scores again: the answer is the exact top-k of <fp32 q, bf16 row>) */
    cand_t* R = (cand_t*)malloc(sizeof(cand_t) * (size_t)kk);
    int rc = 0;
    for (int j = 0; j < cnt; ++j) {
      double acc = 0.0;
      if (h->keep_f32) {
        const float* x = h->rows + L[j].i * h->dim;
        for (int c = 0; c < h->dim; ++c) acc += (double)qn[c] * (double)x[c];
      } else {
        const uint16_t* x = h->rows16 + L[j].i * h->dim;
        for (int c = 0; c < h->dim; ++c) acc += (double)qn[c] * (double)bf16_to_f32(x[c]);
      }
      list_push(R, &rc, kk, acc, L[j].i);
    }
    memcpy(L, R, sizeof(cand_t) * (size_t)rc);
    cnt = rc;
    free(R);
  }
  for (int j = 0; j < k; ++j) {
    oi[j] = j < cnt ? L[j].i + h->row_offset : -1;
    od[j] = j < cnt ? (float)(1.0 - L[j].s) : INFINITY;
  }
  free(L);
}

static int search_all(vdb_t* h, const float* queries, int b, int k, int kp, const uint32_t* allow, int64_t* out_idx,
                      float* out_dist) {
  int rc = check_search_args(h, queries, b, k, kp, out_idx, out_dist);
  if (rc) return rc;
  float* qn = (float*)malloc(sizeof(float) * (size_t)h->dim);
  for (int q = 0; q < b; ++q) {
    ora_normalize_rows(queries + (size_t)q * h->dim, qn, 1, h->dim);
    search_one(h, qn, k, kp, -1, allow, out_idx + (size_t)q * k, out_dist + (size_t)q * k);
  }
  free(qn);
  h->last_bytes = (double)h->n * h->dim * (h->dtype == VDB_BF16 ? 2 : 4);
  return VDB_OK;
}
int vdb_search(vdb_t* h, const float* queries, int b, int k, int kp, int64_t* out_idx, float* out_dist) {
  if (!h) return VDB_ERR_ARG;
  return search_all(h, queries, b, k, kp, NULL, out_idx, out_dist);
}
int vdb_search_fetch(vdb_t* h, const float* queries, int b, int k, int kp, int64_t* out_idx, float* out_dist, float* out_rows) {
  if (!h) return VDB_ERR_ARG;
  if (b > 0 && !out_rows) return fail(h, VDB_ERR_ARG, "search_fetch: NULL rows buffer");
  int rc = search_all(h, queries, b, k, kp, NULL, out_idx, out_dist);
  if (rc) return rc;
  for (int64_t j = 0; j < (int64_t)b * k; ++j) {
    float* o = out_rows + (size_t)j * h->dim;
    int64_t r = out_idx[j] < 0 ? -1 : out_idx[j] - h->row_offset;
    for (int c = 0; c < h->dim; ++c)
      o[c] = r < 0 ? 0.f : h->keep_f32 ? h->rows[r * h->dim + c] : bf16_to_f32(h->rows16[r * h->dim + c]);
  }
  return VDB_OK;
}
int vdb_search_dev(vdb_t* h, const float* q, int b, int k, int kp, int64_t* oi, float* od, void* stream) {
  (void)stream;
  return vdb_search(h, q, b, k, kp, oi, od);
}
int vdb_search_dev_noalloc(vdb_t* h, const float* q, int b, int k, int kp, int64_t* oi, float* od, void* stream) {
  if (!h) return VDB_ERR_ARG;
  if (b > 8) return fail(h, VDB_ERR_ARG, "search_dev_noalloc: at most 8 queries per call (the fixed scratch array holds no more)");
  if ((kp ? kp : k) > 64) return fail(h, VDB_ERR_ARG, "search_dev_noalloc: k (k' when reranking) must be <= 64");
  if (h->dim > 4096) return fail(h, VDB_ERR_ARG, "search_dev_noalloc: dim must be <= 4096");
  return vdb_search_dev(h, q, b, k, kp, oi, od, stream);
}
/* two-phase form: the CPU twin keeps the queries in phase 1 and searches in phase 2 */
int vdb_search_scan_dev(vdb_t* h, const float* q, int b, int k, int kp, int slot, void* stream) {
  (void)stream;
  if (!h) return VDB_ERR_ARG;
  if (slot != 0 && slot != 1) return fail(h, VDB_ERR_ARG, "search_scan: slot must be 0 or 1");
  int rc = check_search_args(h, q, b, k, kp, h, h);
  if (rc) return rc;
  if (b < 1 || b > 64 || h->n == 0) return fail(h, VDB_ERR_STATE, "search_scan: this batch does not split: use vdb_search_dev");
  free(h->split_q[slot]);
  h->split_q[slot] = (float*)malloc(sizeof(float) * (size_t)b * h->dim);
  memcpy(h->split_q[slot], q, sizeof(float) * (size_t)b * h->dim);
  h->split_b[slot] = b; h->split_k[slot] = k; h->split_kp[slot] = kp;
  return VDB_OK;
}
int vdb_search_finish_dev(vdb_t* h, int slot, int64_t* oi, float* od, void* stream) {
  (void)stream;
  if (!h) return VDB_ERR_ARG;
  if (slot != 0 && slot != 1) return fail(h, VDB_ERR_ARG, "search_finish: slot must be 0 or 1");
  if (!h->split_q[slot]) return fail(h, VDB_ERR_STATE, "search_finish: no scan is pending in this slot");
  int rc = search_all(h, h->split_q[slot], h->split_b[slot], h->split_k[slot], h->split_kp[slot], NULL, oi, od);
  free(h->split_q[slot]);
  h->split_q[slot] = NULL;
  return rc;
}
/* where-filtered query: only rows whose bit is set are eligible */
int vdb_search_masked(vdb_t* h, const float* queries, int b, int k, int kp, const uint32_t* allow_bits, int64_t n_bits,
                      int64_t* out_idx, float* out_dist) {
  if (!h) return VDB_ERR_ARG;
  if (!allow_bits && h->n > 0) return fail(h, VDB_ERR_ARG, "search_masked: NULL bitmap");
  if (n_bits != h->n) return fail(h, VDB_ERR_ARG, "search_masked: the bitmap must hold one bit per stored row");
  return search_all(h, queries, b, k, kp, allow_bits, out_idx, out_dist);
}
int vdb_search_masked_dev(vdb_t* h, const float* q, int b, int k, int kp, const uint32_t* allow_bits, int64_t n_bits,
                          int64_t* oi, float* od, void* stream) {
  (void)stream;
  return vdb_search_masked(h, q, b, k, kp, allow_bits, n_bits, oi, od);
}

int vdb_merge_topk_dev(vdb_t* h, const float* cd, int64_t dstride, const int64_t* ci, int64_t istride,
                       int g, int b, int k, int64_t* oi, float* od, void* stream) {
  (void)stream;
  if (!h) return VDB_ERR_ARG;
  if (g < 1 || b < 0 || k < 1 || k > VDB_MAX_K) return fail(h, VDB_ERR_ARG, "merge: bad g/b/k");
  if (b > 0 && (!cd || !ci || !oi || !od)) return fail(h, VDB_ERR_ARG, "merge: NULL buffer");
  cand_t* L = (cand_t*)malloc(sizeof(cand_t) * (size_t)k);
  for (int q = 0; q < b; ++q) {
    int cnt = 0;
    for (int s = 0; s < g; ++s)
      for (int j = 0; j < k; ++j) {
        size_t od_ = (size_t)s * (size_t)(dstride ? dstride : (int64_t)b * k) + (size_t)q * k + j;
        size_t oi_ = (size_t)s * (size_t)(istride ? istride : (int64_t)b * k) + (size_t)q * k + j;
        if (ci[oi_] < 0) continue;
        /* merge on distance ascending == score descending; keep fp32 distance bits */
        list_push(L, &cnt, k, -(double)cd[od_], ci[oi_]);
      }
    for (int j = 0; j < k; ++j) {
      oi[(size_t)q * k + j] = j < cnt ? L[j].i : -1;
      od[(size_t)q * k + j] = j < cnt ? (float)(-L[j].s) : INFINITY;
    }
  }
  free(L);
  return VDB_OK;
}

int vdb_selfjoin(vdb_t* h, int64_t q0, int64_t qn_, int k, int exclude_self, int kp,
                 int64_t* out_idx, float* out_dist) {
  if (!h) return VDB_ERR_ARG;
  if (q0 < 0 || qn_ < 0 || q0 + qn_ > h->n) return fail(h, VDB_ERR_ARG, "selfjoin: query range outside the store");
  int rc = check_search_args(h, h, (int)(qn_ > 0), k, kp, out_idx, out_dist);
  if (rc) return rc;
  float* qv = (float*)malloc(sizeof(float) * (size_t)h->dim);
  for (int64_t q = 0; q < qn_; ++q) {
    int64_t r = q0 + q;
    if (h->keep_f32) memcpy(qv, h->rows + r * h->dim, sizeof(float) * (size_t)h->dim);
    else for (int c = 0; c < h->dim; ++c) qv[c] = bf16_to_f32(h->rows16[r * h->dim + c]);
    search_one(h, qv, k, kp, exclude_self ? r : -1, NULL, out_idx + (size_t)q * k, out_dist + (size_t)q * k);
  }
  free(qv);
  return VDB_OK;
}

int vdb_fetch(vdb_t* h, const int64_t* idx, int64_t n, float* out) {
  if (!h) return VDB_ERR_ARG;
  if (n < 0 || (n > 0 && (!idx || !out))) return fail(h, VDB_ERR_ARG, "fetch: NULL buffer");
  for (int64_t j = 0; j < n; ++j) {
    if (idx[j] < 0 || idx[j] >= h->n) return fail(h, VDB_ERR_ARG, "fetch: index out of range");
    if (h->keep_f32) memcpy(out + j * h->dim, h->rows + idx[j] * h->dim, sizeof(float) * (size_t)h->dim);
    else for (int c = 0; c < h->dim; ++c) out[j * h->dim + c] = bf16_to_f32(h->rows16[idx[j] * h->dim + c]);
  }
  return VDB_OK;
}

/* flat store file, shared with libvdb_hip.so: 32-byte header + payload */
typedef struct { char magic[8]; int32_t dim; int32_t stored_dtype; int64_t n; int64_t row_offset; } vdb_file_header;

int vdb_save(vdb_t* h, const char* dir) {
  if (!h || !dir) return VDB_ERR_ARG;
  mkdir(dir, 0777);
  char path[4096], tmp[4200];
  snprintf(path, sizeof path, "%s/vectors.vdb", dir);
  snprintf(tmp, sizeof tmp, "%s.tmp", path);
  FILE* f = fopen(tmp, "wb");
  if (!f) return fail(h, VDB_ERR_IO, "save: cannot open file for writing");
  vdb_file_header hd;
  memset(&hd, 0, sizeof hd);
  memcpy(hd.magic, "VDBFLAT1", 8);
  hd.dim = h->dim; hd.stored_dtype = h->keep_f32 ? VDB_F32 : VDB_BF16; hd.n = h->n; hd.row_offset = h->row_offset;
  int ok = fwrite(&hd, sizeof hd, 1, f) == 1;
  size_t cnt = (size_t)h->n * h->dim;
  if (ok && cnt) ok = h->keep_f32 ? fwrite(h->rows, 4, cnt, f) == cnt : fwrite(h->rows16, 2, cnt, f) == cnt;
  fclose(f);
  if (ok && rename(tmp, path) != 0) ok = 0;
  if (!ok) remove(tmp);
  return ok ? VDB_OK : fail(h, VDB_ERR_IO, "save: short write");
}

int vdb_load(vdb_t** out, const char* dir, int dtype, int keep_f32, int device_id) {
  if (!out || !dir) return fail(NULL, VDB_ERR_ARG, "load: NULL argument");
  char path[4096];
  snprintf(path, sizeof path, "%s/vectors.vdb", dir);
  FILE* f = fopen(path, "rb");
  if (!f) return fail(NULL, VDB_ERR_IO, "load: cannot open vectors.vdb");
  vdb_file_header hd;
  if (fread(&hd, sizeof hd, 1, f) != 1 || memcmp(hd.magic, "VDBFLAT1", 8) != 0) { fclose(f); return fail(NULL, VDB_ERR_IO, "load: bad header"); }
  if (hd.dim <= 0 || hd.dim > 32768 || (hd.stored_dtype != VDB_F32 && hd.stored_dtype != VDB_BF16) || hd.n < 0) {
    fclose(f);
    return fail(NULL, VDB_ERR_IO, "load: bad header (dim, dtype or row count out of range)");
  }
  {
    struct stat sb;
    int64_t want = (int64_t)sizeof hd + hd.n * (int64_t)hd.dim * (hd.stored_dtype == VDB_F32 ? 4 : 2);
    if (fstat(fileno(f), &sb) != 0 || (int64_t)sb.st_size != want) {
      fclose(f);
      return fail(NULL, VDB_ERR_IO, "load: vectors.vdb is truncated or its header disagrees with its length");
    }
  }
  if (hd.stored_dtype == VDB_BF16 && (dtype == VDB_F32 || keep_f32)) { fclose(f); return fail(NULL, VDB_ERR_STATE, "load: file holds bf16 rows only; cannot rebuild fp32 rows"); }
  vdb_t* h;
  int rc = vdb_create(&h, hd.dim, dtype, keep_f32, device_id);
  if (rc) { fclose(f); return rc; }
  h->row_offset = hd.row_offset;
  rc = vdb_reserve(h, hd.n > 0 ? hd.n : 1);
  if (rc) { snprintf(g_err, 256, "%s", h->err); vdb_destroy(h); fclose(f); return rc; }
  size_t cnt = (size_t)hd.n * hd.dim;
  int ok = 1;
  if (hd.stored_dtype == VDB_F32) {
    /* stored rows are already normalised: copy verbatim, derive bf16 */
    float* buf = h->keep_f32 ? h->rows : (float*)malloc(sizeof(float) * (cnt ? cnt : 1));
    ok = cnt == 0 || fread(buf, 4, cnt, f) == cnt;
    if (ok && h->dtype == VDB_BF16) for (size_t i = 0; i < cnt; ++i) h->rows16[i] = f32_to_bf16_rne(buf[i]);
    if (!h->keep_f32) free(buf);
  } else {
    ok = cnt == 0 || fread(h->rows16, 2, cnt, f) == cnt;
  }
  fclose(f);
  if (!ok) { vdb_destroy(h); return fail(NULL, VDB_ERR_IO, "load: short read"); }
  h->n = hd.n;
  *out = h;
  return VDB_OK;
}

int vdb_load_sharded(vdb_t** out, const char* dir, int dtype, int keep_f32, const int* device_ids, int n_devices) {
  if (!device_ids || n_devices < 1 || n_devices > 64) return fail(NULL, VDB_ERR_ARG, "vdb_load_sharded: need 1..64 device ids");
  int rc = vdb_load(out, dir, dtype, keep_f32, 0);
  if (rc) return rc;
  (*out)->n_shards = n_devices; (*out)->reserved = 1;
  return VDB_OK;
}

int vdb_last_timing(vdb_t* h, float* scan_ms, float* merge_ms, double* bytes) {
  if (!h) return VDB_ERR_ARG;
  if (scan_ms) *scan_ms = 0.f;
  if (merge_ms) *merge_ms = 0.f;
  if (bytes) *bytes = h->last_bytes;
  return VDB_OK;
}
int vdb_get_stat(vdb_t* h, const char* name, int64_t* value) {
  if (!h || !name || !value) return VDB_ERR_ARG;
  if (!strcmp(name, "rows")) { *value = h->n; return VDB_OK; }
  if (!strcmp(name, "capacity_rows")) { *value = h->cap; return VDB_OK; }
  if (!strcmp(name, "n_shards")) { *value = h->n_shards ? h->n_shards : 1; return VDB_OK; }
  if (!strcmp(name, "last_path") || !strcmp(name, "scratch_bytes") || !strcmp(name, "p2p_direct_shards") || !strcmp(name, "workspace_allocs") || !strcmp(name, "last_fused")) { *value = 0; return VDB_OK; }
  if (!strcmp(name, "last_corpus_passes")) { *value = 1; return VDB_OK; }
  if (!strcmp(name, "mfma_last_overflow") || !strcmp(name, "mfma_last_hits") || !strcmp(name, "mfma_last_skipped_tiles")) { *value = 0; return VDB_OK; }
  return fail(h, VDB_ERR_ARG, "get_stat: unknown stat");
}
int vdb_timing_stats(vdb_t* h, int64_t* n, double* a, double* b, int reset) {
  (void)reset;
  if (!h) return VDB_ERR_ARG;
  if (n) *n = 0;
  if (a) *a = 0.0;
  if (b) *b = 0.0;
  return VDB_OK;
}
int vdb_set_option(vdb_t* h, const char* name, int64_t value) {
  (void)value;
  if (!h || !name) return VDB_ERR_ARG;
  if (!strcmp(name, "mfma_variant") || !strcmp(name, "mfma_cost_model") || !strcmp(name, "mfma_force_fallback") || !strcmp(name, "mfma_append_cap") || !strcmp(name, "mfma_ablate") || !strcmp(name, "timing_ring") || !strcmp(name, "timing_every") || !strcmp(name, "mfma_split_x1000") || !strcmp(name, "mfma_skip_sample") || !strcmp(name, "scan_blocks_per_cu") || !strcmp(name, "scan_nt_loads") || !strcmp(name, "scan_group_launch") || !strcmp(name, "alloc_f32_first") || !strcmp(name, "stream_xcd_slices") || !strcmp(name, "scan_xcd_slices") || !strcmp(name, "scan_super_rounds") || !strcmp(name, "batch_mfma_min_b") || !strcmp(name, "stream_nt_loads")) return VDB_OK;
  return fail(h, VDB_ERR_ARG, "set_option: unknown option");
}

int vdb_synth_rows_dev(vdb_t* h, float* out, int64_t row0, int64_t n, const vdb_synth_spec* spec, void* stream) {
  (void)stream;
  if (!h || !spec || (n > 0 && !out) || n < 0) return VDB_ERR_ARG;
  ora_synth_rows(out, row0, n, h->dim, spec);
  return VDB_OK;
}
int vdb_add_synth(vdb_t* h, int64_t row0, int64_t n, const vdb_synth_spec* spec) {
  if (!h || !spec || n < 0) return VDB_ERR_ARG;
  const int64_t chunk = 4096;
  float* buf = (float*)malloc(sizeof(float) * (size_t)chunk * h->dim);
  for (int64_t r = 0; r < n; r += chunk) {
    int64_t m = n - r < chunk ? n - r : chunk;
    ora_synth_rows(buf, row0 + r, m, h->dim, spec);
    int rc = vdb_add(h, buf, m);
    if (rc) { free(buf); return rc; }
  }
  free(buf);
  return VDB_OK;
}

int ora_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
