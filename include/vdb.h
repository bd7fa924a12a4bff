/*
 * vdb.h — C ABI of the MI355X-native exact cosine top-k engine (libvdb_hip.so).
 *
 * The reference (billenewman4/VectorDB, a pure-Python project) has no FFI of
 * its own: its hot path leaves the repository through the chromadb Python
 * object boundary.  Each entry point below therefore cites the reference
 * *Python call site* whose work it replaces (paths relative to the reference
 * checkout).  The Python host layer (vectordb_amd/) binds these symbols with
 * ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, opaque handle, every function returns 0 on success and a
 *     negative vdb_status on failure; vdb_last_error() gives the message.
 *   - caller allocates every output buffer; no callbacks; no exceptions.
 *   - one host thread per handle at a time; different handles may be driven
 *     from different threads.
 *   - "host" pointers are ordinary CPU memory, "dev" pointers are device
 *     memory on the handle's GPU (e.g. tensor.data_ptr()).
 *   - row index = insertion order (0-based) + row_offset (shard offset).
 *   - scores are cosine similarities of L2-normalised vectors; distances are
 *     1 - cosine (fp32), ascending; ties are broken by ascending row index.
 */
#ifndef VDB_H
#define VDB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdb vdb_t;

enum vdb_status {
  VDB_OK = 0,
  VDB_ERR_ARG = -1,      /* invalid argument                        */
  VDB_ERR_DEVICE = -2,   /* HIP runtime error / no GPU              */
  VDB_ERR_MEMORY = -3,   /* allocation failed                       */
  VDB_ERR_IO = -4,       /* save/load failure                       */
  VDB_ERR_STATE = -5     /* call not valid in the handle's state    */
};

enum vdb_dtype { VDB_F32 = 0, VDB_BF16 = 1 };

#define VDB_MAX_K 1024 /* largest k (or rerank k') one call may ask for */

/* Counter-based synthetic row generator: identical bits on
 * host (oracle) and device, so 150 GB corpora never cross PCIe.
 *   kind 0  isotropic : x = u(seed,row,col), u uniform in [-1,1)
 *   kind 1  clustered : x = u(seed^C, row % n_clusters, col) + noise*u(seed,row,col)
 *   kind 2  query     : x = base_row(src) + noise*u(seed,row,col), where
 *                       src = hash(seed,row) % base_n and base_row is the raw
 *                       (un-normalised) row `src` of the corpus described by
 *                       base_seed/base_kind/base_n_clusters/base_noise. */
typedef struct vdb_synth_spec {
  uint64_t seed;
  int32_t kind;
  int32_t n_clusters;
  float noise;
  int32_t base_kind;
  uint64_t base_seed;
  int64_t base_n;
  int32_t base_n_clusters;
  float base_noise;
} vdb_synth_spec;

/* ---- lifecycle --------------------------------------------------------- */

/* Replaces chromadb.PersistentClient(...).get_or_create_collection(name,
 * metadata={"hnsw:space":"cosine"}) as the owner of the vector segment
 * (src/vectordb.py:149,156-160).  dtype: storage of the scanned copy.
 * keep_f32_for_rerank: with VDB_BF16, also keep fp32 rows for the rerank (6 instead of 2 bytes per element). */
int vdb_create(vdb_t** out, int dim, int dtype, int keep_f32_for_rerank, int device_id);

/* The same store ROW-SHARDED over the GPUs of one node, inside one process and one host thread (SURVEY.md 8b/8e;
 * the reference is single-process, src/vectordb.py:259-264 — there is nothing to cite beyond the same call sites as
 * vdb_create).  The handle routes every call to one child store per entry of device_ids (an id may repeat: several
 * shards on one GPU).  vdb_reserve(h, total_rows) must come first: it fixes the partition — shard s owns global rows
 * [s*per, (s+1)*per), per = ceil(total/n_devices) rounded up to 256, the last shard also any rows beyond the total —
 * and row indices stay insertion order, so a sharded handle returns bit for bit what a single store of the same
 * rows returns.  A search runs every shard's scan (+ shard-local fp32 rerank) concurrently on its own device and
 * stream; each shard's finish kernel stores its k results straight into shard 0's gather buffer (over xGMI when peer
 * access is available, a peer copy otherwise) and one merge kernel on shard 0 ranks the n_devices lists.  No host
 * synchronisation inside vdb_search_dev; queries and outputs of the *_dev calls live on shard 0's device.
 * Shard 0 (it lives on the callers' device) runs on the CALLER's stream, the others on streams of their own behind one
 * event each way; the host spends ~16 us per further shard on those runtime calls (profiles/r03_sharded_issue_cost.json).
 * Served on a sharded handle: reserve, rebalance, warm, add, add_dev (rows on shard 0's device; each span goes to its
 * shard, over a peer copy where that is another GPU), add_synth, count, search, search_fetch, search_dev, search_masked
 * and search_masked_dev (the bitmap covers the GLOBAL rows; a shard's first row is a multiple of 256, so its slice is
 * word-aligned), selfjoin (blocks of rows-as-queries cut at shard borders, searched like any batch), fetch, save,
 * last_timing, timing_stats, set_option (applied to every shard), get_stat.  Not served (VDB_ERR_STATE):
 * set_row_offset, the two-phase pair, search_dev_noalloc, merge_topk_dev. */
int vdb_create_sharded(vdb_t** out, int dim, int dtype, int keep_f32_for_rerank, const int* device_ids, int n_devices);

/* Replaces client.delete_collection(name) for the vector segment
 * (src/vectordb.py:180). */
void vdb_destroy(vdb_t* h);

/* Message of the last failing call on this handle (owned by the library).
 * With h == NULL: the message of the last failing vdb_create/vdb_load. */
const char* vdb_last_error(vdb_t* h);

/* Pre-size the store for n_rows rows (avoids regrowth copies).  No reference
 * counterpart (hnswlib's max_elements resize is internal to chromadb). */
int vdb_reserve(vdb_t* h, int64_t n_rows);

/* Re-partition a row-sharded handle for total_rows rows (no counterpart in the reference: chromadb's segment lives in
 * one process' memory).  The stored rows move between the shards' GPUs as they are (device to device, verbatim bits),
 * so a store that outgrew the partition its first vdb_reserve fixed — the last shard takes every row beyond it — is
 * balanced again; searches return what they returned before.  Needs room for a second copy of each shard while it
 * runs.  On a single store it is vdb_reserve.  The Chroma-shaped face calls it as a collection grows (doubling). */
int vdb_rebalance(vdb_t* h, int64_t total_rows);

/* Size every workspace that searches of up to max_batch queries with this k / rerank_kprime need on the store as it is
 * now (one dummy search per kernel family on generated queries, results discarded), so that no later search of those
 * shapes grows a workspace: a growth drains the device, frees and allocates inside the caller's search.  No counterpart
 * in the reference.  Blocks until done. */
int vdb_warm(vdb_t* h, int max_batch, int k, int rerank_kprime);

/* Global index of this shard's row 0 (multi-GPU row sharding). */
int vdb_set_row_offset(vdb_t* h, int64_t row_offset);

/* ---- add -------------------------------------------- */

/* Replaces collection.add(embeddings=...) (src/vectordb.py:215-219): append n
 * rows (host, row-major n x dim fp32).  Rows are L2-normalised on the device
 * (cosine space) and stored fp32 and/or bf16. */
int vdb_add(vdb_t* h, const float* rows_host, int64_t n);

/* Same, rows already resident in device memory (embedder output, generator); on a sharded handle: on shard 0's device. */
int vdb_add_dev(vdb_t* h, const float* rows_dev, int64_t n);

/* Replaces collection.count() (tests/test_vectordb.py:101). */
int vdb_count(vdb_t* h, int64_t* n);

/* ---- query --------------------------------- */

/* Replaces collection.query(query_embeddings=, n_results=k)
 * (src/vectordb.py:259-264; analysis/bidirectional_similarity.py:110-114;
 * analysis/run_bidirectional_test.py:139-143;
 * analysis/generate_best_usda_matches.py:196-200) and the top-k selection
 * inside chroma-hnswlib.  queries: host, b x dim fp32 (not required to be
 * normalised).  rerank_kprime: 0 = no rerank; otherwise (bf16 store) the scan
 * keeps k' >= k candidates that are re-scored in fp32 against the fp32 copy
 * (the GPU analogue of src/vectordb.py:282-314) — or, on a bf16 store created
 * WITHOUT the fp32 copy (2 bytes per element: 100M x 768 rows fit one GPU),
 * against the stored bf16 rows with the fp32 query: on the GEMV scan the answer
 * is then that of rerank_kprime = 0 (the exact top-k of <fp32 query, bf16 row>);
 * batches may take the batched MFMA kernels instead of ceil(b/4) scan passes, which
 * choose the k' candidates from bf16-rounded QUERIES — the same answer within the
 * k' candidate limit described below (near-duplicate clusters larger than k').  Outputs: b x k,
 * padded with idx -1 / dist +inf when fewer than k rows exist.
 * Precision: an fp32 store, and a bf16 store with a rerank, return fp32 cosines of the stored fp32 rows whatever the
 * batch size.  A bf16 store searched WITHOUT a rerank returns the exact top-k of <fp32 query, bf16 row> scores; it
 * always runs on the GEMV scan (the batched MFMA kernels round the queries to bf16 as well, which only a rerank
 * repairs), so its result does not depend on the batch size either.
 * Limit of the rerank: the k' candidates are chosen by bf16 scores (resolution 2^-9 relative), so a true top-k row can
 * be missed when MORE than k' rows lie within that resolution of the k-th best — clusters of near-duplicates larger
 * than k', or thousands of unit vectors in two or three dimensions.  The returned rows, their fp32 distances and their
 * order are still exact for what is returned, and the k-th distance is within bf16 resolution of the true one; on
 * embedding-like data k' = 4k gives recall 1.0 at 10M rows (tests/test_gpu_fullsize.py).  Where that is not enough: a
 * larger k' (up to 1024) or an fp32 store. */
int vdb_search(vdb_t* h, const float* queries_host, int b, int k, int rerank_kprime,
               int64_t* out_idx_host, float* out_dist_host);

/* vdb_search that also returns the STORED rows of the hits: replaces collection.query(..., include=[..., "embeddings"])
 * as the reference's hot call makes it (src/vectordb.py:259-264 asks for 30 hits with their embeddings and re-scores
 * them, :282-308).  out_rows_host: b x k x dim fp32, the normalised fp32 row of every hit (zeros behind an idx of -1),
 * gathered on the device behind the search and returned in the same copy — one round trip instead of a search and a
 * vdb_fetch. */
int vdb_search_fetch(vdb_t* h, const float* queries_host, int b, int k, int rerank_kprime,
                     int64_t* out_idx_host, float* out_dist_host, float* out_rows_host);

/* Same with device-resident queries and outputs, asynchronous on `stream`
 * (a hipStream_t; NULL = the default (null) stream, as everywhere in HIP).  The
 * caller synchronises.  All *_dev calls of one handle must be issued on one
 * stream at a time (they share the handle's workspaces). */
int vdb_search_dev(vdb_t* h, const float* queries_dev, int b, int k, int rerank_kprime,
                   int64_t* out_idx_dev, float* out_dist_dev, void* stream);

/* vdb_search_dev for callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation.  NO COUNTERPART IN THE REFERENCE.  Same inputs and outputs as vdb_search_dev.  Results:
 * identical (bit for bit) whenever vdb_search_dev runs the GEMV scan itself; where vdb_search_dev would take the batched
 * MFMA kernels (bf16 store with a rerank, from 3 queries on a large store), equal up to the k' candidate limit — this
 * entry point picks its k' candidates with fp32 queries, the batched kernels with bf16-rounded ones, which differ only
 * inside near-duplicate clusters larger than k' (tests/test_zz_no_alloc.py pins both statements).  Here
 * the search always runs on the GEMV scan and its workspaces (normalised queries, partial top-k' lists) come out of a
 * fixed array in device memory that the library holds per GPU (2.3 MiB, "scratch_bytes" in vdb_get_stat) instead of
 * the handle's grown-on-demand buffers: the call makes no hipMalloc/hipFree and creates no event.  Limits (beyond
 * them: VDB_ERR_ARG, nothing is launched): b <= 8 queries, k (k' when reranking) <= 64, dim <= 4096.  All calls on
 * one GPU — from any handle — share that GPU's array: calls that overlap in time on one GPU race; issue them on one
 * stream, or order them yourself.  No timing events are recorded (vdb_last_timing keeps the previous search's). */
int vdb_search_dev_noalloc(vdb_t* h, const float* queries_dev, int b, int k, int rerank_kprime,
                           int64_t* out_idx_dev, float* out_dist_dev, void* stream);

/* vdb_search_dev in two phases, for callers that pipeline batches (the multi-GPU exchange of SURVEY 8e):
 * phase 1 normalises the queries and scans the corpus (the HBM-bound part), phase 2 merges the partial lists,
 * reranks and ranks.  The phases may run on different streams (the caller orders phase 2 after phase 1 with an
 * event), and `slot` (0 or 1) names one of two workspace sets, so that phase 2 of batch i overlaps phase 1 of
 * batch i+1.  A slot holds one pending scan.  Only batches that run as ONE scan round split: 1..64 queries on
 * the GEMV scan; otherwise vdb_search_scan_dev returns VDB_ERR_STATE and does nothing (use vdb_search_dev).
 * Same results as vdb_search_dev.  No counterpart in the reference (its queries are synchronous, one at a time).
 * While a scan is pending in either slot, every other search entry point of the handle (they share slot 0's
 * workspaces), vdb_selfjoin and the "timing_ring" option return VDB_ERR_STATE; a second scan into a slot that is
 * still pending does too. */
int vdb_search_scan_dev(vdb_t* h, const float* queries_dev, int b, int k, int rerank_kprime, int slot, void* stream);
int vdb_search_finish_dev(vdb_t* h, int slot, int64_t* out_idx_dev, float* out_dist_dev, void* stream);

/* Replaces collection.query(..., where=where_filter) (src/vectordb.py:230 and
 * :259-264: find_similar_products passes its where_filter through to Chroma,
 * which pre-filters by metadata and searches the matching rows only).  The host
 * evaluates the metadata predicate into a bitmap, one bit per stored row of THIS
 * handle in insertion order (bit r = word r/32, bit r%32; the row offset does
 * not shift it): set = the row may be returned.  n_bits must equal vdb_count.
 * The scan skips ineligible rows before top-k selection (and does not read
 * 4..16-row tiles without an eligible row at all), so the result is the exact
 * top-k of the eligible rows, -1/+inf padded when fewer than k are eligible.
 * Runs on the GEMV scan for every batch size. */
int vdb_search_masked(vdb_t* h, const float* queries_host, int b, int k, int rerank_kprime,
                      const uint32_t* allow_bits_host, int64_t n_bits,
                      int64_t* out_idx_host, float* out_dist_host);

/* Same with device-resident queries, bitmap and outputs, asynchronous on `stream`. */
int vdb_search_masked_dev(vdb_t* h, const float* queries_dev, int b, int k, int rerank_kprime,
                          const uint32_t* allow_bits_dev, int64_t n_bits,
                          int64_t* out_idx_dev, float* out_dist_dev, void* stream);

/* Multi-GPU merge (new in the build): after an all-gather of the
 * per-shard results, cand_* hold g lists per query laid out [g][b][k]
 * (ascending distance, idx -1 = empty); shard s starts at element
 * s*dist_shard_stride / s*idx_shard_stride (0 = dense, b*k), so one packed
 * all-gather buffer can be merged in place.  Writes the b x k global top-k. */
int vdb_merge_topk_dev(vdb_t* h, const float* cand_dist_dev, int64_t dist_shard_stride,
                       const int64_t* cand_idx_dev, int64_t idx_shard_stride, int g, int b, int k,
                       int64_t* out_idx_dev, float* out_dist_dev, void* stream);

/* All-pairs self-join over stored rows [q_row0, q_row0+q_n) as queries
 * (the loop of analysis/bidirectional_similarity.py:246-271, which issues one
 * query per product).  exclude_self drops the query's own row
 * (bidirectional_similarity.py:161 skips self matches).  Outputs host,
 * q_n x k. */
int vdb_selfjoin(vdb_t* h, int64_t q_row0, int64_t q_n, int k, int exclude_self,
                 int rerank_kprime, int64_t* out_idx_host, float* out_dist_host);

/* Replaces collection.get(ids=, include=["embeddings"]) and the
 * include=["embeddings"] part of query (src/vectordb.py:262,270;
 * analysis/generate_best_usda_matches.py:185-188): copy the stored
 * (normalised, fp32) rows with the given *local* indices to host. */
int vdb_fetch(vdb_t* h, const int64_t* idx_host, int64_t n, float* out_rows_host);

/* ---- persistence ------------------------------------ */

/* Replace the PersistentClient directory (src/config.py:43,
 * src/vectordb.py:149): flat file `<dir>/vectors.vdb`. */
int vdb_save(vdb_t* h, const char* dir);
int vdb_load(vdb_t** out, const char* dir, int dtype, int keep_f32_for_rerank, int device_id);
/* Reopen a saved store row-sharded (vdb_create_sharded's partition of the file's row count).  The file format is
 * the same whichever kind of handle wrote it; vdb_save writes to a temporary name and renames it into place, and
 * vdb_load* reject a header whose row count, dimension or dtype is out of range or disagrees with the file's
 * length (VDB_ERR_IO). */
int vdb_load_sharded(vdb_t** out, const char* dir, int dtype, int keep_f32_for_rerank, const int* device_ids, int n_devices);

/* ---- measurement ------------------------------------------------------- */

/* hipEvent timing of the last search on this handle: scan kernel(s) time,
 * merge+rerank time, and the algorithmic bytes the scan streamed
 * (n * dim * sizeof(T) per corpus pass).  Synchronises the handle's events. */
int vdb_last_timing(vdb_t* h, float* scan_ms, float* merge_ms, double* bytes_streamed);

/* Sums over the searches recorded in the timing ring (option "timing_ring" = ring
 * capacity, default 1; at most that many most-recent searches are summed).
 * Lets a benchmark read the scan kernel's average duration over its whole timed
 * region without synchronising between steps.  reset != 0 restarts the count. */
int vdb_timing_stats(vdb_t* h, int64_t* n_searches, double* scan_ms_sum, double* merge_ms_sum, int reset);

/* Tuning knobs: "timing_ring", "timing_every" (record the timing events on every Nth search only; default 1), "scan_blocks_per_cu", "scan_nt_loads", "scan_xcd_slices" and "stream_xcd_slices" (default 1: the blocks of one XCD take adjacent tiles of each grid-wide step; 0 = block-by-block round-robin; same results), "scan_super_rounds" (default 1 = 8 tiles per wave and super-round; 2..64 set that number: the one-launch scan's block groups take their tiles as slices of super-rounds, so the grid walks the corpus as one window; 0 = one contiguous range per group; same results), "alloc_f32_first" (default 0: a bf16 store's scanned rows are allocated before its fp32 rerank copy; set before the first reserve/add), "scan_group_launch" (default 1: a GEMV-path batch sends all its whole groups of four queries in ONE 2-D launch; 0 = one launch per group; same results), "batch_mfma_min_b" (default 2),
 * "mfma_cost_model" (default 1: a batch takes the MFMA path only where it is faster than ceil(b/4) GEMV passes;
 * 0: whenever it is supported), "mfma_variant" (4: 256x128 tiles only; 5: + 256x256 tiles; 6, default: + the streaming
 * kernel for <= 64 queries), "stream_nt_loads", "mfma_append_cap", "mfma_split_x1000" (where the sample pass sits between the dense sample and
 * the full pass: 0, the default, = 500, the geometric mean, for the tiled kernels and 600 for the streaming kernel, both measured
 * best; or 100..900), "mfma_skip_sample" (default 1: the last of two append passes leaves out the tiles the sample pass
 * multiplied), "mfma_force_fallback" (test hook: the exact list kernel serves the batch), "mfma_ablate" (diagnostic builds of the
 * pass kernels; bits 1, 2, 4, 8 give INVALID results — timing only; bit 128 = one append flush per hit).
 * Round 3: "fuse_small_batch" (0: never the one-launch search; 1, default: where it wins — one query always, two or four
 * from 4M rows on; 2: wherever the shape allows), "mfma_dense_tiles" (256-row tiles of the dense sample that gives the first
 * threshold; 0, default: 64, or 128 from 6M rows on), "mfma_one_pass_span" (default 24: stores up to that many times the
 * dense sample are searched in one append pass).  On a sharded handle only: "shard_exchange_rccl" (1: the shards' k results
 * are exchanged by ONE RCCL all-gather — loaded on demand, one rank per GPU, VDB_ERR_STATE when two shards share a device —
 * and merged on shard 0, the exchange SURVEY 8e names; 0, default: P2P stores into shard 0's gather buffer),
 * "shard_force_remote" (test hook: every shard but the first takes the no-peer-access path), "shard_own_stream_for_shard0"
 * (diagnostic: the round-2 form).
 * Unknown names return VDB_ERR_ARG. */
int vdb_set_option(vdb_t* h, const char* name, int64_t value);

/* Read-only counters/diagnostics (synchronises the device): "mfma_last_overflow" (1 when the
 * last batched search overflowed its candidate buffers and ran the exact list kernel),
 * "rows", "capacity_rows", "n_shards", "rows_per_shard" and "p2p_direct_shards" (sharded handles), "scratch_bytes",
 * "last_path" (kernel family of the last search: 0 = GEMV scan, 4 / 5 = tiled MFMA kernels, 6 = streaming MFMA kernel),
 * "last_corpus_passes" (corpus passes of the last search: the roofline's algorithmic bytes = passes x rows x row bytes),
 * "mfma_last_hits", "mfma_last_skipped_tiles" (tiles the final pass of the last
 * batched search left out because the sample pass had multiplied them); measured ceilings of this device (a few ms
 * each): "probe_hbm_read_mbps" (streams the store's own rows, needs >= 64 MiB of them) and
 * "probe_mfma16_bf16_gflops" / "probe_mfma_bf16_gflops" (bare v_mfma_f32_16x16x32_bf16 / 32x32x16 loops on random operands).
 * Unknown names return VDB_ERR_ARG. */
int vdb_get_stat(vdb_t* h, const char* name, int64_t* value);

/* ---- synthetic data (bench/test utility; no reference counterpart) ----- */

/* Write raw (un-normalised) synthetic rows [row0,row0+n) into out_dev
 * (n x dim fp32, device memory on the handle's GPU). */
int vdb_synth_rows_dev(vdb_t* h, float* out_dev, int64_t row0, int64_t n,
                       const vdb_synth_spec* spec, void* stream);

/* Generate and append rows [row0,row0+n) chunk by chunk on the device. */
int vdb_add_synth(vdb_t* h, int64_t row0, int64_t n, const vdb_synth_spec* spec);

#ifdef __cplusplus
}
#endif
#endif /* VDB_H */
