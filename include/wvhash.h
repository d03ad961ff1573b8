/*
 * wvhash.h -- C ABI of libwvhash.so, the MI355X (gfx950) implementation of the
 * wavelet-hashing retrieval hot path of ArseneAmoya/image-retrieval-wavelet.
 *
 * The reference is pure Python; its FFI boundary for this path is "Python calls into a native
 * wheel" (PyWavelets for the transform, ATen / faiss for the ranking, ATen for the attention
 * head).  Each entry point below names the reference call site it replaces (file:line under
 * /root/reference) -- the ctypes binding a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *   - Every function returns 0 (WV_OK) or a negative errno-style code; nothing throws across
 *     the ABI.  wv_last_error() returns a per-thread message for the last failure.
 *   - All pointers are DEVICE pointers unless the name says host.  The caller owns every
 *     buffer; the library allocates nothing.  Scratch comes from the caller, sized by the
 *     matching *_workspace_bytes() query.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue
 *     work: no hidden synchronisation, no host<->device copies.
 *   - Thread-safe and re-entrant; no global mutable state except the per-thread error string.
 *   - One process per GPU; the current HIP device of the calling thread is used.
 */
#ifndef WVHASH_H
#define WVHASH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WV_OK 0
#define WV_EINVAL (-22)  /* bad argument (message says which) */
#define WV_ENOMEM (-12)  /* workspace too small */
#define WV_ENOTSUP (-95) /* shape outside what the kernels implement */
#define WV_EHIP (-5)     /* HIP runtime error at launch */

/* element types of image / coefficient buffers */
#define WV_DT_U8 0
#define WV_DT_F32 1
#define WV_DT_BF16 2

/* image memory layouts */
#define WV_LAYOUT_NCHW 0 /* [B][C][H][W]  (torch, after ToTensor-style permute) */
#define WV_LAYOUT_NHWC 1 /* [B][H][W][C]  (PIL / numpy, what np.array(img) yields) */

const char *wv_last_error(void);
/* 5 = this header.  History: 5 added WV_METRIC_L2_SQUARED, wv_rank_scores[_cpu] and the host twins wv_knn_float_cpu,
 * wv_band_attn_pool_cpu, wv_hash_tail_cpu; 4 added the host twins of the ranking side (wv_pack_bits_cpu ... wv_hit_prefix_cpu); 2 added wv_head_params.q_proj, the host twins, wv_swt2d_forward_ex, the two-step shard entry
 * points and wv_map_at_k_ld; 3 added wv_head_params.prepared / wv_band_attn_prepare (one-launch head front), the ranking + AP
 * entry points (wv_hamming_map_at_k, wv_rank_labels_prepare), wv_hamming_shard_prefix / wv_topk_merge_cum_need and the
 * relevance-string pair of the sharded mAP (wv_hamming_shard_relbits, wv_merge_relbits_map).  A struct gaining a field bumps it. */
int wv_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Stationary wavelet transform.
 * Replaces pywt.swt2(channel, wavelet, level)[0] + np.stack + the /255 scaling of
 *   main/transforms/custom_transforms.py:145-157 (BaseWaveletTransform.__call__) and
 *   :163-166 (SWTTransform._apply_wavelet), for a whole batch at once.
 * in : [B,C,H,W] or [B,H,W,C]; u8 values are divided by 255.0f, f32 values are used as is.
 * out: [B][C][4][H][W], band order cA, cH, cV, cD of level `level` (coarsest) only.
 * H and W must be multiples of 2^level (PyWavelets raises otherwise -> WV_EINVAL).
 * dec_lo / dec_hi: HOST pointers to `flen` decomposition taps (PyWavelets order).
 * workspace: needed only for shapes the tiled kernel does not cover (query below; may be 0).
 * ------------------------------------------------------------------------------------------ */
size_t wv_swt2d_workspace_bytes(int B, int C, int H, int W, int level, int flen);
int wv_swt2d_forward(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int B,
                     int C, int H, int W, int level, const float *dec_lo, const float *dec_hi,
                     int flen, void *workspace, size_t workspace_bytes, void *stream);

/* Same transform with a choice of output layout.
 *   WV_BANDS_INNER: out [B][C][4][H][W] -- the reference's tensor (custom_transforms.py:155-157, batched).
 *   WV_BANDS_OUTER: out [4][B'][C][H][W], band_stride = B'*C*H*W elements between the bands of one plane
 *                   (B' >= B: a chunk of a larger batch may be written in place).  Every band is then one
 *                   contiguous NCHW batch: exactly what SharedDinoHashing.forward builds with
 *                   x.permute(2,0,1,3,4).contiguous().view(4B,3,H,W) (multi_dino_attention.py:818) and what
 *                   MultiDinoHashing indexes per backbone (:745) -- written directly, without that second copy of
 *                   the sub-band tensor.  Returns WV_ENOTSUP for shapes only the tiled/generic kernels cover. */
#define WV_BANDS_INNER 0
#define WV_BANDS_OUTER 1
int wv_swt2d_forward_ex(const void *in, int in_dtype, int in_layout, void *out, int out_dtype, int out_layout,
                        int64_t band_stride, int B, int C, int H, int W, int level, const float *dec_lo,
                        const float *dec_hi, int flen, void *workspace, size_t workspace_bytes, void *stream);

/* RawStackTransform (custom_transforms.py:172-188): `copies` identical planes per channel.
 * out: [B][C][copies][H][W]. */
int wv_rawstack_forward(const void *in, int in_dtype, int in_layout, void *out, int out_dtype,
                        int B, int C, int H, int W, int copies, void *stream);

/* ------------------------------------------------------------------------------------------
 * Host twins of the three transform entry points ("_cpu twins taking host pointers").
 * The reference calls its transform inside forked DataLoader worker processes, one image at a time
 *   (main/datasets/flikr_coco.py:59-60 -> custom_transforms.py:145-157); a forked worker cannot use the parent's
 *   GPU context, so with an unchanged transform YAML and num_workers > 0 the plugin's __call__ runs these.
 * in / out are HOST pointers; out is float32 [B][C][4|copies][H'][W'] like the device entry points.
 * No HIP call, no thread started, no global state: safe after fork().  Float32 arithmetic in the order of the
 * device kernels (bit-identical to wv_swt2d_forward on the shapes its sliding kernel covers).
 * ------------------------------------------------------------------------------------------ */
int wv_swt2d_forward_cpu(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W,
                         int level, const float *dec_lo, const float *dec_hi, int flen);
int wv_rawstack_forward_cpu(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W,
                            int copies);
int wv_dwt2d_forward_cpu(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W,
                         int level, const float *dec_lo, const float *dec_hi, int flen);

/* ------------------------------------------------------------------------------------------
 * Host twins of the ranking-side entry points (BASELINE config c0 is the CPU plumbing case).
 * The reference's calculator runs on CPU tensors (main/engine/accuracy_calculator.py:279-349 with self.device = cpu,
 *   main/engine/evaluate.py:76-81); CustomCalculator(device='cpu') -- explicit, never a silent fallback -- runs these.
 * All pointers are HOST pointers; argument meaning as for the device entry point of the same name (no stream, no
 * workspace).  Same results bit for bit: packed words, counts, distances and the (distance, row) order are integers; the
 * average precision sums its fp32 quotients in wv_map_at_k's order.  No HIP call, no thread, no global state.
 *   wv_pack_bits_cpu     <- wv_pack_bits        wv_bit_counts_cpu   <- wv_bit_counts     wv_hamming_dist_cpu <- wv_hamming_dist
 *   wv_hamming_topk_cpu  <- wv_hamming_topk (stable counting sort; dist may be NULL)
 *   wv_map_at_k_cpu      <- wv_map_at_k_ld      wv_hit_prefix_cpu   <- wv_hit_prefix
 * ------------------------------------------------------------------------------------------ */
int wv_pack_bits_cpu(const float *src, int64_t ld_src, uint64_t *packed, int64_t rows, int nbits, int mode, int32_t *bad_flag);
int wv_bit_counts_cpu(const uint64_t *packed, int64_t rows, int nbits, uint32_t *counts);
int wv_hamming_dist_cpu(const uint64_t *q, const uint64_t *db, uint8_t *dist, int64_t ld_dist, int Q, int64_t N, int words);
int wv_hamming_topk_cpu(const uint64_t *q, const uint64_t *db, int32_t *idx, uint8_t *dist, int Q, int64_t N, int nbits, int k,
                        int64_t idx_offset);
int wv_map_at_k_cpu(const int32_t *idx, int64_t ld, int Q, int k, const uint64_t *qlab, const uint64_t *dblab, int lwords,
                    float *ap, int32_t *nrel);
int wv_hit_prefix_cpu(const int32_t *idx, int Q, int k, const uint64_t *qlab, const uint64_t *dblab, int lwords, uint32_t *hits);

/* Decimated multi-level 2-D DWT (DWTTransform, custom_transforms.py:191-205 -> pywt.wavedec2, mode
 * 'symmetric'): the four bands (cA, cH, cV, cD) of the coarsest level.
 * out: float32 [B][C][4][Hn][Wn] with Hn = wv_dwt_out_len(H, flen, level) (each level: floor((n + flen - 1) / 2)). */
int wv_dwt_out_len(int n, int flen, int level);
size_t wv_dwt2d_workspace_bytes(int B, int C, int H, int W, int level, int flen);
int wv_dwt2d_forward(const void *in, int in_dtype, int in_layout, float *out, int B, int C, int H, int W,
                     int level, const float *dec_lo, const float *dec_hi, int flen, void *workspace,
                     size_t workspace_bytes, void *stream);

/* One level of the legacy lifting-scheme DWT behind CustomTransform (custom_transforms.py:14-55,90-117;
 * wavelets/haar.py:21-86, wavelets/cdf_97.py:33-133): basis 0 = haar, 1 = cdf 9/7, zero-padded lifting steps,
 * 2-D scales (1/2, 1, 1, sqrt 2).  in: float32 [planes][H][W], H and W even (the caller pads like
 * HaarLifting / Cdf97Lifting do);  ll: [planes][H/2][W/2];  hi: [planes][3][H/2][W/2] = (LH, HL, HH).
 * Bit-identical to the reference's float32 torch ops (every product and sum rounded separately, same order). */
size_t wv_lifting2d_workspace_bytes(int64_t planes, int H, int W);
int wv_lifting2d_forward(const float *in, int64_t planes, int H, int W, int basis, float *ll, float *hi,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Bit packing of +-1 hash codes and multi-hot labels.
 * Codes come out of torch.sign(logits) (multi_dino_attention.py:833) as fp32 in {-1,+1};
 * the reference keeps them as fp32 rows and multiplies (accuracy_calculator.py:183-186).
 * packed[row][w] bit j  =  src[row][64*w + j] > 0.      words = ceil(nbits / 64).
 * mode 0 (codes): *bad_flag |= 1 if any value is not exactly +1 or -1 (sign(0)=0, NaN ...)
 * mode 1 (labels): *bad_flag |= 1 if any value is negative or NaN (then "q.r > 0" is no longer
 *                  "shares a tag" and label_comparison_fn :31-37 cannot be evaluated on bits).
 * bad_flag: device int32, caller zero-initialises; may be NULL.
 * ------------------------------------------------------------------------------------------ */
int wv_pack_bits(const float *src, int64_t ld_src, uint64_t *packed, int64_t rows, int nbits,
                 int mode, int32_t *bad_flag, void *stream);

/* Per-bit population counts over the rows (per_bit_balance, accuracy_calculator.py:188-194:
 * (reference > 0).float().mean(0) = counts / rows).  counts: device uint32[nbits], zeroed here. */
int wv_bit_counts(const uint64_t *packed, int64_t rows, int nbits, uint32_t *counts, void *stream);

/* ------------------------------------------------------------------------------------------
 * Hamming distances.  Replaces calc_hamming_dist (accuracy_calculator.py:183-186),
 * 0.5 * (B - q @ r.T), for +-1 codes, where it is an exact small integer.
 * dist[qi * ld_dist + n] = popcount(q[qi] ^ db[n]),  uint8 (nbits <= 255).
 * ------------------------------------------------------------------------------------------ */
int wv_hamming_dist(const uint64_t *q, const uint64_t *db, uint8_t *dist, int64_t ld_dist, int Q,
                    int64_t N, int words, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused Hamming distance + ranking: the k nearest database codes of every query in ascending
 * (distance, database index) order -- the canonical tie-break of torch.argsort(stable=True).
 * Replaces  hamm = calc_hamming_dist(...); indices = torch.argsort(hamm)[:topk]
 *   (accuracy_calculator.py:219-223) and get_knn_torch's  q @ r.T + torch.topk(largest=True)
 *   (get_knn.py:63-66; IP score = nbits - 2 * dist), and faiss IndexFlatIP.search (:35-52).
 * idx : int32 [Q][k], values = local row + idx_offset (idx_offset = first global row of this
 *       shard when the database is row-sharded across GPUs).
 * dist: uint8 [Q][k].        Requires 1 <= k <= N, nbits <= 128.
 * ------------------------------------------------------------------------------------------ */
size_t wv_hamming_topk_workspace_bytes(int Q, int64_t N, int words, int k);
int wv_hamming_topk(const uint64_t *q, const uint64_t *db, int32_t *idx, uint8_t *dist, int Q,
                    int64_t N, int nbits, int k, int64_t idx_offset, void *workspace,
                    size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Prepared database.  The reference rebuilds a faiss index on every call (index.add, get_knn.py:54);
 * here the per-database work is explicit: wv_db_prepare lays the packed codes out once in the two
 * images the kernels read with fully coalesced loads (distance tile image + ranking column image),
 * and the *_prepared entry points skip the per-call re-layout.  Results are identical to the
 * plain entry points.  `prepared` is caller-owned device memory of wv_db_prepared_bytes() bytes.
 * ------------------------------------------------------------------------------------------ */
size_t wv_db_prepared_bytes(int64_t N, int words);
int wv_db_prepare(const uint64_t *db, int64_t N, int words, void *prepared, size_t prepared_bytes,
                  void *stream);
int wv_hamming_dist_prepared(const uint64_t *q, const void *prepared, uint8_t *dist, int64_t ld_dist,
                             int Q, int64_t N, int words, void *stream);
int wv_hamming_topk_prepared(const uint64_t *q, const void *prepared, int32_t *idx, uint8_t *dist, int Q,
                             int64_t N, int nbits, int k, int64_t idx_offset, void *stream);

/* Same ranking with every option: `prepared` (or NULL) instead of / beside `db`, and `cum` (or NULL):
 * uint32 [Q][nbits + 2], cum[q][b] = number of database rows of THIS call with distance < b (cum[q][nbits+1] = N).
 * A row-sharded search all-reduces `cum` to learn each query's global k-th distance, and then exchanges only
 * the list prefixes that can matter (wvhash/parallel.py) -- the role faiss' host-side shard merge plays at
 * get_knn.py:41-44. */
int wv_hamming_topk_ex(const uint64_t *q, const uint64_t *db, const void *prepared, int32_t *idx, uint8_t *dist,
                       uint32_t *cum, int Q, int64_t N, int nbits, int k, int64_t idx_offset, void *workspace,
                       size_t workspace_bytes, void *stream);

/* mAP@k of every query straight from the codes -- what calculate_maphashing (accuracy_calculator.py:183-231) returns -- without
 * writing the ranked lists: the windowed ranking kernel builds the list in LDS as wv_hamming_topk does (ascending distance,
 * ties by ascending row) and evaluates it there against a relevance bitmap of the query made from the rows' label words
 * (lwords = 1 or 2 64-bit multi-hot words per row, i.e. up to 128 classes: label_comparison_fn, :31-37).  ap float32 [Q] and nrel int32 [Q] (or NULL) are exactly
 * what wv_map_at_k returns for wv_hamming_topk's list (same summation order; bit-identical at 256 threads per query).
 *   prepared         wv_db_prepare's blob of the database codes
 *   prepared_labels  wv_rank_labels_prepare's class-major bit matrix of the rows' label words
 *                    (wv_rank_labels_prepared_bytes(N) bytes; 0 = N is outside the windowed kernel)
 * Returns WV_ENOTSUP for shapes outside the fused kernel (more than 32,768 rows, k > 32,639 -- the list is built in LDS with
 * 16-bit cells --, lists that do not fit LDS beside the label bitmap, ...; mAP@ALL with k = N is inside for N <= 32,639): the caller then runs
 * wv_hamming_topk + wv_map_at_k. */
size_t wv_rank_labels_prepared_bytes(int64_t N, int lwords);
int wv_rank_labels_prepare(const uint64_t *dblab, int64_t N, int lwords, void *prepared_labels, size_t prepared_bytes, void *stream);
int wv_hamming_map_at_k(const uint64_t *q, const void *prepared, const void *prepared_labels, const uint64_t *qlab, int lwords,
                        int Q, int64_t N, int nbits, int k, float *ap, int32_t *nrel, void *stream);

/* The two steps of a row-sharded search (wvhash/parallel.py; the role of faiss' shard search + host merge at
 * get_knn.py:41-44) on one shard of at most 32,768 rows:
 *   wv_hamming_hist        only the cumulative distance histogram of every query, cum uint32 [Q][nbits + 2] as above -- no
 *                          list is built.  All-reduced over the shards it gives every rank the global k-th distance of every
 *                          query, hence how long a list prefix each shard has to contribute (about k / shards + ties).
 *   wv_hamming_topk_rows16 the k nearest rows of the shard per query in (distance, row) order as 16-bit LOCAL row numbers,
 *                          uint16 [Q][k] -- the wire format wv_topk_merge_cum reads -- with k = that prefix length.
 * Returns WV_ENOTSUP for shards outside the windowed kernel's range (the caller then uses wv_hamming_topk_ex). */
int wv_hamming_hist(const uint64_t *q, const uint64_t *db, const void *prepared, uint32_t *cum, int Q, int64_t N, int nbits,
                    void *workspace, size_t workspace_bytes, void *stream);
int wv_hamming_topk_rows16(const uint64_t *q, const uint64_t *db, const void *prepared, uint16_t *rows, int Q, int64_t N,
                           int nbits, int k, void *workspace, size_t workspace_bytes, void *stream);
/* Both in one pass, for a steady stream of query batches whose prefix length is known from earlier batches (wvhash/parallel.py
 * with send_hint): the shard's k nearest rows per query as 16-bit local row numbers AND its complete cumulative histograms.
 * No all-reduce is needed before the lists can be built; whether k was enough is checked by the receiver of the lists
 * (wv_topk_merge_cum_need). */
int wv_hamming_shard_prefix(const uint64_t *q, const uint64_t *db, const void *prepared, uint16_t *rows, uint32_t *cum, int Q,
                            int64_t N, int nbits, int k, void *workspace, size_t workspace_bytes, void *stream);

/* Merge of G per-shard top-k lists (gathered with one all-gather) into the global top-k.
 * Replaces the host-side shard merge inside faiss.index_cpu_to_all_gpus(shards=True)
 *   (get_knn.py:41-44).  Lists must come from contiguous row shards in rank order, so that
 * ascending global index inside a distance bucket = (shard, position) order.
 * idx_in/dist_in: [G][Q][kin];  idx_out/dist_out: [Q][k], k <= G*kin. nbits <= 128.
 * Shorter lists are padded with dist = nbits + 1 (idx = -1): such entries rank after every real one. */
int wv_topk_merge(const int32_t *idx_in, const uint8_t *dist_in, int G, int Q, int kin,
                  int32_t *idx_out, uint8_t *dist_out, int k, int nbits, void *stream);

/* Compact merge for the sharded search: every shard sends, per query, its cumulative distance histogram
 * (cum, uint32 [G][Q][nbits + 2], the `cum` output of wv_hamming_topk_ex: rows with distance < b) instead of a
 * distance row, and its list as 16-bit LOCAL row numbers (idx_local uint16 [G][Q][kin], shard_rows <= 65536;
 * shard g holds global rows [g * shard_rows, ...)).  A sorted list is fully described by its histogram.
 * Output as wv_topk_merge.  Replaces the host merge of faiss' sharded index (get_knn.py:41-44) with 2.2x fewer
 * exchanged bytes than int32 indices + uint8 distances. */
int wv_topk_merge_cum(const uint16_t *idx_local, const uint32_t *cum, int G, int Q, int kin, int64_t shard_rows,
                      int32_t *idx_out, uint8_t *dist_out, int k, int nbits, void *stream);
/* The same merge; additionally need_out[0] = max(need_out[0], the longest prefix any shard had to contribute for a query of
 * this launch) -- from the unclamped histograms: the merged lists are exact iff that value is <= kin.  The caller zeroes
 * need_out beforehand and looks at it whenever it synchronises anyway. */
int wv_topk_merge_cum_need(const uint16_t *idx_local, const uint32_t *cum, int G, int Q, int kin, int64_t shard_rows,
                           int32_t *idx_out, uint8_t *dist_out, int k, int nbits, int32_t *need_out, void *stream);

/* Sharded mAP@k without lists on the wire (wvhash/parallel.py: sharded_hamming_map_at_k).  calculate_maphashing needs, of every
 * list entry, only whether it is relevant: a shard therefore sends per query the RELEVANCE STRING of its k nearest rows (bit p
 * = its p-th nearest row shares a label with the query; 1 bit per entry instead of a 16-bit row number) and its cumulative
 * histogram; the receiver interleaves the strings of the G shards bin by bin (global order = distance, shard, position) and
 * evaluates the merged string exactly as wv_map_at_k evaluates a list.
 *   wv_hamming_shard_relbits  relbits uint64 [Q][ceil(k / 64)], cum uint32 [Q][nbits + 2]; prepared / prepared_labels: the
 *                             shard's wv_db_prepare and wv_rank_labels_prepare blobs; qlab: label word of every query
 *   wv_merge_relbits_map      relbits [G][Q][ceil(kin / 64)], cum [G][Q][nbits + 2] -> ap float32 [Q], nrel int32 [Q] (or NULL),
 *                             need_out as in wv_topk_merge_cum_need
 * relbits_ld / cum_ld: row pitches in uint64 / uint32 units (0 = tight): string and histogram of a (query, shard) may lie side
 * by side in ONE wire buffer, so that a single all_to_all moves both.
 * WV_ENOTSUP outside the windowed kernel's range (shards of more than 32,768 rows, k > 32,639, wider labels); any k inside it:
 * a shard can send the relevance string of its whole ranking (mAP@ALL). */
int wv_hamming_shard_relbits(const uint64_t *q, const void *prepared, const void *prepared_labels, const uint64_t *qlab,
                             int lwords, uint64_t *relbits, int64_t relbits_ld, uint32_t *cum, int64_t cum_ld, int Q, int64_t N,
                             int nbits, int k, void *stream);
int wv_merge_relbits_map(const uint64_t *relbits, int64_t relbits_ld, const uint32_t *cum, int64_t cum_ld, int G, int Q, int kin,
                         int k, int nbits, float *ap, int32_t *nrel, int32_t *need_out, void *stream);

/* Ranking from a stored distance matrix row (same order as wv_hamming_topk). */
int wv_rank_from_dist(const uint8_t *dist_matrix, int64_t ld_dist, int Q, int64_t N, int nbits,
                      int32_t *idx, uint8_t *dist, int k, void *stream);

/* ------------------------------------------------------------------------------------------
 * Average precision of each query over its ranked list.
 * Replaces the per-query body of calculate_maphashing (accuracy_calculator.py:216-229):
 *   gnd = labels share a tag; tgnd = gnd[indices][:topk]; AP = mean_j(j / rank_j) over hits.
 * idx: int32 [Q][k] database rows (as written by wv_hamming_topk); entries < 0 are skipped.
 * qlab [Q][lwords], dblab [N][lwords]: labels packed by wv_pack_bits(mode 1).
 * ap: float32 [Q] (0 when the query has no hit in its list); nrel: int32 [Q] = hits.
 * mAP = sum(ap) / Q  (the reference divides by all queries, :231).
 * ------------------------------------------------------------------------------------------ */
int wv_map_at_k(const int32_t *idx, int Q, int k, const uint64_t *qlab, const uint64_t *dblab,
                int lwords, float *ap, int32_t *nrel, void *stream);

/* The same over the first k entries of longer lists (row pitch ld >= k): mAP@k for several k from ONE ranking at the
 * largest k -- evaluate_multi_k (main/engine/evaluate.py:172-245) re-ranks once per k in the reference. */
int wv_map_at_k_ld(const int32_t *idx, int64_t ld, int Q, int k, const uint64_t *qlab, const uint64_t *dblab,
                   int lwords, float *ap, int32_t *nrel, void *stream);

/* Running hit counts along each ranked list: hits[q][p] = relevant entries among idx[q][0..p] (uint32 [Q][k]).
 * The ratios of these counts are the secondary retrieval diagnostics of accuracy_calculator.py:131-181
 * (RetrievalRPrecision, RetrievalPrecision(top_k=1), RetrievalPrecisionRecallCurve) and the full-gallery
 * precision/recall curves of calculate_pr_rc_hashing (:235-273).  Same relevance rule as wv_map_at_k. */
int wv_hit_prefix(const int32_t *idx, int Q, int k, const uint64_t *qlab, const uint64_t *dblab,
                  int lwords, uint32_t *hits, void *stream);

/* ------------------------------------------------------------------------------------------
 * Real-valued k-NN (non-binary embeddings).  Replaces get_knn_torch (get_knn.py:60-71):
 *   metric 0: scores = q @ r.T,           top-k largest  (hamming / cosine branch)
 *   metric 1: d = torch.cdist(q, r, p=2), top-k smallest (true L2)
 *   metric 2: the same neighbours with faiss IndexFlatL2's SQUARED distances (get_knn.py:38-39,55)
 * L2 rows are ranked on the squared distance |q|^2 + |r|^2 - 2 q.r (clamped at 0) -- what faiss ranks on, and
 * torch.cdist's order up to the ties the rounding of the root creates; metric 1 takes the root of the k results.
 * Ties are broken by ascending database index.  idx int32 [Q][k], val float32 [Q][k].  Any D >= 1 (rows 16-byte aligned when D % 4 == 0), N <= 2^26,
 * q and db 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
#define WV_METRIC_IP 0
#define WV_METRIC_L2 1
#define WV_METRIC_L2_SQUARED 2
size_t wv_knn_float_workspace_bytes(int Q, int64_t N, int D, int k);
int wv_knn_float(const float *q, const float *db, int Q, int64_t N, int D, int metric, int k,
                 int32_t *idx, float *val, void *workspace, size_t workspace_bytes, void *stream);
/* Host twin (HOST pointers, no stream, no workspace; csrc/host_knn.cpp): the same indices and values bit for bit --
 * v_mfma_f32_32x32x2_f32 is an fmaf chain (k = 0 before k = 1; tools/mfma_order_test.hip), the twin walks k in the order
 * the kernel feeds it, forms the squared norms in the kernel's lane / butterfly order and ranks on the same keys. */
int wv_knn_float_cpu(const float *q, const float *db, int Q, int64_t N, int D, int metric, int k, int32_t *idx, float *val);

/* The ranking stage of wv_knn_float on scores the caller made: per row of S [Q][N] (dense) the k best columns, smallest
 * first (WV_RANK_DESCENDING: largest first), ties by ascending column -- idx int32 [Q][k] column numbers, val float32
 * [Q][k] their scores (WV_RANK_SQRT: the root of them, for squared distances).  It is the merge of per-shard k-NN lists
 * that faiss' sharded index does on the host (get_knn.py:41-44): the shards' (value, global row) lists laid side by side in
 * shard order rank to exactly the unsharded result (wvhash.parallel.sharded_knn_float).  Host twin: same arguments, host
 * pointers, no workspace / stream. */
#define WV_RANK_DESCENDING 1
#define WV_RANK_SQRT 2
size_t wv_rank_scores_workspace_bytes(int Q, int64_t N, int k);
int wv_rank_scores(const float *S, int Q, int64_t N, int k, int flags, int32_t *idx, float *val, void *workspace,
                   size_t workspace_bytes, void *stream);
int wv_rank_scores_cpu(const float *S, int Q, int64_t N, int k, int flags, int32_t *idx, float *val);

/* ------------------------------------------------------------------------------------------
 * Band-attention pooling head + hashing tail (eval mode).
 * Replaces CrossAttentionBottleneckHeadAdvanced.forward (multi_dino_attention.py:1111-1141;
 * same core in ...Head :1030-1062, ...Pooled :568-599, ...Decoupled :448-481) and
 * SharedDinoHashing's tail hash_fc -> BatchNorm1d(eval) -> sign (:829-833).
 * All weights row-major fp32 exactly as in the module's state_dict.
 * ------------------------------------------------------------------------------------------ */
typedef struct wv_head_params {
    int embed_dim;   /* E (multiple of 32; 384 for ViT-S) */
    int num_heads;   /* E % num_heads == 0 */
    int num_queries; /* Nq */
    int num_tokens;  /* S = 4 band tokens */
    int pool_mean;   /* 0: concat read-out (Nq*E -> E), 1: mean read-out (E -> E) */
    const float *q_eff;      /* [Nq][E] effective query tokens (after optional normalise*scale) */
    const float *in_proj_w;  /* [3E][E] */
    const float *in_proj_b;  /* [3E] */
    const float *attn_out_w; /* [E][E] */
    const float *attn_out_b; /* [E] */
    const float *norm1_w, *norm1_b; /* [E] */
    const float *mlp0_w;     /* [4E][E] */
    const float *mlp0_b;     /* [4E] */
    const float *mlp2_w;     /* [E][4E] */
    const float *mlp2_b;     /* [E] */
    const float *out_w;      /* [E][Nq*E] or [E][E] */
    const float *out_b;      /* [E] */
    const float *norm2_w, *norm2_b; /* [E] */
    float ln_eps;
    /* optional: [Nq][E] projected queries  q_eff @ Wq^T + bq  made once by wv_band_attn_qproj (the query tokens are
     * parameters: in eval mode their projection does not change from call to call); NULL = computed in every call */
    const float *q_proj;
    /* optional: the blob wv_band_attn_prepare wrote (projected queries + the weights of every product up to the MLP
     * output in MFMA-fragment order, with the K projection folded into the queries).  When set, and the configuration
     * has a fused kernel (wv_band_attn_prepared_bytes != 0), everything before the read-out product runs as ONE
     * launch (csrc/head_front.hip) and q_proj is not needed.  It is a snapshot of in_proj_w, q_eff, attn_out_w, mlp0_w,
     * mlp2_w: make it again whenever one of them changes.  NULL = separate launches (any configuration). */
    const void *prepared;
} wv_head_params;

size_t wv_band_attn_pool_workspace_bytes(const wv_head_params *p, int B);
/* q_proj_out float32 [Nq][E] = q_eff @ in_proj_w[0:E]^T + in_proj_b[0:E] (the Q third of nn.MultiheadAttention's packed
 * in-projection, multi_dino_attention.py:1081,1128) */
int wv_band_attn_qproj(const wv_head_params *p, float *q_proj_out, void *stream);
/* Size of the prepared blob for this configuration; 0 = no fused kernel (E = 384, 4 band tokens and 4 or 8 queries
 * have one), use the separate-launch path.  wv_band_attn_prepare fills `prepared_out` (device memory of that size) on
 * `stream`; p->q_proj and p->prepared are ignored by it.  The weights are parameters (multi_dino_attention.py:1064-1109):
 * in eval mode the blob is made once per parameter update, like q_proj. */
size_t wv_band_attn_prepared_bytes(const wv_head_params *p);
int wv_band_attn_prepare(const wv_head_params *p, void *prepared_out, void *stream);
/* feats: [S][B][E] (band-major, the layout cls_tokens.chunk(4) has at :824) -> out [B][E] */
int wv_band_attn_pool(const wv_head_params *p, const float *feats, int B, float *out,
                      void *workspace, size_t workspace_bytes, void *stream);

/* logits = fused @ hash_w.T (+ hash_b); BN(eval); codes = sign(logits) as fp32 and bit-packed.
 * Any of logits_out / codes_out / packed_out may be NULL.  bn_* NULL = no BatchNorm. */
int wv_hash_tail(const float *fused, int B, int E, const float *hash_w, const float *hash_b,
                 const float *bn_w, const float *bn_b, const float *bn_mean, const float *bn_var,
                 float bn_eps, int nbits, float *logits_out, float *codes_out,
                 uint64_t *packed_out, void *stream);

/* Host twins of the two entry points above (HOST pointers everywhere, incl. the members of wv_head_params; q_proj and
 * prepared are ignored; no stream, no workspace; csrc/host_head.cpp) for a model whose tensors live on the host.  fp32 with
 * a machine-independent summation order (eight interleaved fmaf chains, one fixed tree); they agree with the kernels to
 * fp32 rounding (both are held to the reference-made golden vectors with the same tolerance), the codes wherever the
 * logit is not within that of zero. */
int wv_band_attn_pool_cpu(const wv_head_params *p, const float *feats, int B, float *out);
int wv_hash_tail_cpu(const float *fused, int B, int E, const float *hash_w, const float *hash_b, const float *bn_w,
                     const float *bn_b, const float *bn_mean, const float *bn_var, float bn_eps, int nbits,
                     float *logits_out, float *codes_out, uint64_t *packed_out);

#ifdef __cplusplus
}
#endif
#endif /* WVHASH_H */
