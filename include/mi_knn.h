/* mi_knn.h -- C ABI of the MI355X (gfx950) feature-bank similarity library (libmi_knn.so).
 *
 * Drop-in boundary for ONE path of sailist/image-classification: the weighted-kNN monitor
 * (models/knn.py) and the MoCo queue (models/memory_bank.py, track_ssl/moco.py).  The reference
 * is pure Python on torch; it has no FFI of its own.  What a maintainer binds from Python is
 * shown in INTEGRATION.md (ctypes, as `image-classification_amd/_lib.py` does).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless it says "host";
 *   - plain pointers and sizes only, no torch types; the library never allocates or frees
 *     caller memory: outputs and workspaces are caller-allocated (sizes from the *_bytes calls);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, nothing synchronises the host;
 *   - every entry point returns 0 (MI_OK) or a negative MI_ERR_* code; the message of the last
 *     error on the calling thread is available from mi_last_error();
 *   - re-entrant, with three documented exceptions.  The entry points keep no state between calls beyond the
 *     thread-local error string, so calls from several host threads / on several streams are independent, EXCEPT:
 *       (1) mi_knn_topk_static works in ONE per-GPU static scratch array: it is not re-entrant per GPU -- calls
 *           that overlap on one GPU (two streams, two threads) race on that array; keep them on one stream;
 *       (2) mi_profile_enable / mi_profile_collect / mi_profile_clock are process-wide DEBUG switches behind
 *           mutexes: while a stamp buffer is registered EVERY filter launch of the process runs the stamped
 *           kernel instantiation, while profiling is on every launch is bracketed by events;
 *       (3) the first call of a kernel that needs more than the default dynamic LDS raises that kernel's limit
 *           once per process (std::call_once), not per call.
 */
#ifndef MI_KNN_H
#define MI_KNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 8

enum { MI_OK = 0, MI_ERR_ARG = -1, MI_ERR_HIP = -2, MI_ERR_WORKSPACE = -3, MI_ERR_UNSUPPORTED = -4 };
enum { MI_F32 = 0, MI_BF16 = 1 };            /* compute dtype of the similarity GEMM            */
enum { MI_MAX_K = 1024, MI_MAX_CLASSES = 65536 };

typedef void *mi_stream_t;

int mi_abi_version(void);
const char *mi_last_error(void);

/* Optional per-launch timing for benchmarks (no reference counterpart).  While enabled, every
 * similarity-kernel launch (kind MI_PROF_SIM; `work` = algorithmic flops 2*B*cols*D of that
 * launch) and every select launch (MI_PROF_SELECT; work = bytes) is bracketed by HIP events on the
 * stream it is launched on.  mi_profile_collect waits for the recorded launches of `kind`, returns
 * their summed duration (ms), count and work, and forgets them.  Process-wide debug state behind
 * a mutex; leave it off in production. */
enum { MI_PROF_SIM = 0, MI_PROF_SELECT = 1, MI_PROF_MOCO_FWD = 2, MI_PROF_MOCO_BWD = 3 };
int mi_profile_enable(int on);
int mi_profile_collect(int kind, double *total_ms, int64_t *launches, double *total_work);
/* Diagnostic: the shader clock the chip sustains inside the similarity kernel (MI355X lowers it under MFMA load).
 * While a buffer is registered (device memory, `entries` pairs of uint64, 16-byte aligned), every filter launch of
 * mi_knn_topk runs a stamped instantiation of the same kernel; workgroup b writes (shader cycles, 100 MHz ticks) of
 * its lifetime to pair b % entries.  clock = cycles / ticks x 100 MHz (median over workgroups).  NULL switches it
 * off.  Never on in a timed run: it is a separate build of the kernel.  Process-wide debug state behind a mutex. */
int mi_profile_clock(void *stamps, int64_t entries);

/* Debug / test switch (no reference counterpart), process-wide like mi_profile_*: mi_knn_topk starts its filter with a
 * SPECULATIVE first threshold per query row and redoes, exactly, any row the speculation failed for.  With
 * modulus > 0 every query row whose index is a multiple of it is treated as failed, so that the fallback runs; the results
 * must not change.  0 switches it off. */
int mi_debug_force_fallback(int modulus);

/* Leading dimension (in ELEMENTS of `dtype`) the similarity kernels require for a D-wide row:
 * D rounded up so that a row is a whole number of 128-byte K-steps (32 f32 / 64 bf16). */
int64_t mi_knn_ld(int64_t D, int dtype);

/* Pack float32 rows into the compute layout: dst[r][0..D) = (dtype) src(r, .), dst[r][D..ld_dst) = 0.
 * transposed == 0: src is [rows, D] with leading dimension ld_src  (query matrix; [N, D] bank).
 * transposed == 1: src is [D, rows] with leading dimension ld_src  (the reference's bank layout:
 *                  track_ssl/ssltrainer.py:179 hands knn_predict a physically transposed [D, N]).
 * Replaces: the layout/dtype handling torch.mm does implicitly at models/knn.py:6. */
int mi_pack_rows(const float *src, int64_t ld_src, int transposed, int64_t rows, int64_t D,
                 int dtype, void *dst, int64_t ld_dst, mi_stream_t stream);

/* Bank construction without the reference's three extra passes.  Replaces track_ssl/ssltrainer.py:176-179
 * (per memo batch `F.normalize(feature, dim=1)`, then `cat(...).t().contiguous()`): the b rows of one batch
 * are L2-normalised (v / max(||v||_2, eps), eps = 1e-12 in F.normalize) and written, in the compute dtype
 * and zero padded, to rows [row0, row0 + b) of a pre-allocated packed bank [n_rows, ld]
 * (ld == mi_knn_ld(D, dtype)) -- directly the layout mi_knn_topk reads, and directly into a rank's row shard.
 * src is [b, D] float32 with leading dimension ld_src. */
int mi_bank_append_normalized(const float *src, int64_t ld_src, int64_t b, int64_t D, int dtype,
                              void *bank, int64_t ld, int64_t n_rows, int64_t row0, float eps, mi_stream_t stream);

/* Recommended workspace (bytes) for mi_knn_topk: fixed per-row state plus a candidate buffer of
 * up to 8 GiB.  mi_knn_topk sizes its filter stages by the workspace it is actually given: the
 * candidate buffer holds the worst case of one stage (every similarity of the stage passing), so a
 * larger workspace means fewer, larger stages; the minimum is the fixed part + B_pad*2048*8 bytes
 * (B_pad = B rounded up to 256).  The result never depends on the workspace size. */
size_t mi_knn_topk_workspace_bytes(int64_t B, int64_t N, int k);
/* The same for a candidate buffer of up to `candidate_bytes` instead of 8 GiB (never more than one stage over the
 * whole bank needs): on a 288 GB part a caller with memory to spare buys fewer threshold refreshes with it. */
size_t mi_knn_topk_workspace_bytes_for_budget(int64_t B, int64_t N, int k, size_t candidate_bytes);

/* Fused similarity GEMM + streaming per-row top-k.  Replaces models/knn.py:6 (torch.mm) and
 * models/knn.py:8 (topk) without materialising the [B, N] similarity matrix.
 *   q    [B, ldq]  packed queries  (dtype, K-contiguous, ldq == mi_knn_ld(D, dtype), 16-B aligned)
 *   bank [N, ldb]  packed bank rows (same dtype; one row per bank entry = a row shard of the bank)
 *   out_val [B, k] float32 similarities, sorted descending; ties -> lower bank index first
 *   out_idx [B, k] int64 = idx_base + local bank row (idx_base = first global row of this shard)
 * Rows with fewer than k finite similarities are padded with (-inf, -1).  NaN similarities are
 * never selected.  Requires 1 <= k <= min(N, MI_MAX_K).
 * fp32 mode accumulates each dot product as ONE k-ascending fused-multiply-add chain (bit-exact
 * against oracle/knn_chain.c); bf16 mode uses bf16 MFMA with float32 accumulation. */
int mi_knn_topk(const void *q, int64_t ldq, const void *bank, int64_t ldb, int dtype,
                int64_t B, int64_t N, int64_t D, int k, int64_t idx_base,
                float *out_val, int64_t *out_idx,
                void *workspace, size_t workspace_bytes, mi_stream_t stream);

/* mi_knn_topk for callers that cannot allocate device memory.  No counterpart in the reference (pure torch,
 * which allocates freely): it is for code under HIP graph capture, tight serving loops and harnesses that
 * forbid hipMalloc -- they cannot provide the workspace mi_knn_topk asks for.  Same arguments, same kernels
 * and bit-identical results; the workspace is a fixed MI_KNN_STATIC_SCRATCH_BYTES array inside the library's
 * code object (`static __device__`, one copy per GPU, held until the process ends).  The call allocates and
 * frees nothing (no hipMalloc/hipFree, no allocating library routine); the first call on a device resolves
 * that device's address of the array with hipGetSymbolAddress (make that first call outside a graph capture).
 * Limit: the smallest workspace of the shape must fit the array -- roughly 16.4 KiB + 8 k bytes per query row,
 * rows rounded up to 256 (k = 200: B <= 3584) -- otherwise MI_ERR_WORKSPACE; N is not limited by it.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * serialise them.  Every word that must start at zero is cleared at the top of each call. */
#define MI_KNN_STATIC_SCRATCH_BYTES (64u << 20)
int mi_knn_topk_static(const void *q, int64_t ldq, const void *bank, int64_t ldb, int dtype,
                       int64_t B, int64_t N, int64_t D, int k, int64_t idx_base,
                       float *out_val, int64_t *out_idx, mi_stream_t stream);

/* Deterministic merge of per-shard candidates (the step after the RCCL all-gather; no reference
 * counterpart -- the reference kNN is single-device).
 *   cand_val [G, B, kc] float32, cand_idx [G, B, kc] int64 (global rows; < 0 = padding)
 *   -> out_val/out_idx [B, k], total order (value desc, global index asc).  G*kc <= 16384,
 *      every valid index < 2^32. */
int mi_knn_merge(const float *cand_val, const int64_t *cand_idx, int G, int64_t B, int kc, int k,
                 float *out_val, int64_t *out_idx, mi_stream_t stream);

/* k-th largest value of a 1-D float32 bank.  Replaces track_semi/chmatch.py:252-253 and :294-295
 * (`thresh, _ = mem['fine'].topk(int(k)); thresh[-1]`, a 50 000-score bank, k up to 0.8 n): MSB-first radix
 * select, no sort, result = one of the inputs bit for bit (NaN orders above +inf, as torch.topk).
 * out: device float32[1].  1 <= k <= n < 2^31. */
int mi_kth_largest(const float *x, int64_t n, int64_t k, float *out, mi_stream_t stream);

/* Label gather + exp(s/T) weights + per-class vote + full class ranking.
 * Replaces models/knn.py:10-20 (gather, div/exp, one-hot scatter, weighted sum, argsort).
 *   val/idx     [B, k] from mi_knn_topk / mi_knn_merge (idx < 0 entries are skipped)
 *   labels      [n_labels] int64 class ids of the bank rows that idx addresses
 *   out_scores  [B, classes] float32 or NULL
 *   out_rank    [B, classes] int64: classes by score descending, ties -> lower class id
 * Votes of one class are summed over neighbours in ascending neighbour rank (deterministic).
 * 1 <= classes <= MI_MAX_CLASSES, 1 <= k <= MI_MAX_K. */
int mi_knn_vote(const float *val, const int64_t *idx, const int64_t *labels, int64_t n_labels,
                int64_t B, int k, int classes, float knn_t,
                float *out_scores, int64_t *out_rank, mi_stream_t stream);

/* Truncating ring-buffer enqueue.  Replaces models/memory_bank.py:65-72 (MemoryBank.push):
 * writes w = min(b, K - ptr) rows of `value` at queue[ptr .. ptr+w) and stores the new pointer
 * (ptr + w) % K to *new_ptr (HOST pointer, may be NULL).  Overflow rows are dropped, exactly as
 * the reference does.  row_bytes = bytes of one queue row (any dtype). */
int mi_queue_push(void *queue, int64_t K, int64_t row_bytes, int64_t ptr,
                  const void *value, int64_t b, int64_t *new_ptr, mi_stream_t stream);

/* WRAPPING ring-buffer enqueue.  Replaces models/module_utils.py:73-85 (MemoryBank.push and
 * LongTensorMemoryBank.push:125-136, the queues of supcontrast/ft/infomin/cmc/swav/byol): value row r goes
 * to queue row (cursor + r) % K -- the overflow wraps to the front instead of being dropped -- and the new
 * cursor (cursor + b) % K is stored to *new_cursor (HOST pointer, may be NULL).  A batch longer than the queue is
 * accepted exactly where the reference's two slice assignments fit (b <= 2K - cursor: `data[cursor:] = item[:right]`,
 * then `data[:left] = item[right:]` -- rows written by both keep the later item row); b > 2K - cursor is an error, as
 * the reference's second assignment raises (it has written the first slice by then; this call writes nothing). */
int mi_queue_push_wrap(void *queue, int64_t K, int64_t row_bytes, int64_t cursor,
                       const void *value, int64_t b, int64_t *new_cursor, mi_stream_t stream);

/* Index-addressed row store.  Replaces models/memory_bank.py:32 (StorageBank.scatter:
 * `self[name][index] = value`): bank[index[i]] = value[i] for i < b; rows of row_bytes bytes.
 * Negative indices count from the end (torch semantics); out-of-range indices are skipped. */
int mi_bank_scatter(void *bank, int64_t K, int64_t row_bytes, const int64_t *index,
                    const void *value, int64_t b, mi_stream_t stream);

/* MoCo InfoNCE over a negative queue, forward.  Replaces the arithmetic behind
 * track_ssl/moco.py:152-156 (lumo contrastive_loss2 with query_neg=False, key_neg=False,
 * norm=False; third-party source absent: parity unpinned, MoCo-paper form):
 *   l_pos[i] = q_i . key_i ; l_neg = q @ queue^T ; logits = [l_pos, l_neg] / T
 *   lse[i] = logsumexp(logits_i) ; loss = mean_i(lse[i] - l_pos[i] / T)
 * q, key [b, D] float32 (leading dim D), queue [K, D] float32, D % 32 == 0.
 * Outputs: lse [b], lpos [b], loss [1] (all device float32).  The logits are never written. */
size_t mi_moco_workspace_bytes(int64_t b, int64_t K, int64_t D);
int mi_moco_fwd(const float *q, const float *key, const float *queue, int64_t b, int64_t K,
                int64_t D, float temperature, float *lse, float *lpos, float *loss,
                void *workspace, size_t workspace_bytes, mi_stream_t stream);

/* Backward.  At the reference's default call site key and queue are detached (moco.py:152-154) and only dq is
 * wanted; its symmetric branch (moco.py:144-148, `query=feat_key, key=feat_query`) carries the gradient through
 * `key` instead.  Either output may be NULL (not both):
 *   dq[i]   = grad_loss / (b T) * ( sum_n softmax_i[n] queue[n] + (softmax_i[pos] - 1) key_i )
 *   dkey[i] = grad_loss / (b T) * (softmax_i[pos] - 1) q_i          (needs only lse / lpos: no pass over the queue)
 * The probabilities are recomputed tile by tile from q, queue and lse (never stored).
 * grad_loss: device pointer to the upstream scalar gradient.  dq needs D % 64 == 0, D <= 256 and the workspace
 * (mi_moco_workspace_bytes covers forward and backward); with dq == NULL queue/workspace may be NULL and any D
 * the forward accepts is fine.  q, key, dq, dkey: 16-byte aligned.
 * The reference pushes into the queue BEFORE backward (moco.py:159-172) and therefore clones the
 * whole queue every step (moco.py:153).  Here the caller may pass the live, already pushed-into
 * queue plus the rows the push overwrote: queue rows [patch_lo, patch_lo + patch_n) are read from
 * patch_rows [patch_n, D] instead (patch_n == 0: no patch, patch_rows may be NULL). */
int mi_moco_bwd(const float *q, const float *key, const float *queue, int64_t b, int64_t K,
                int64_t D, float temperature, const float *lse, const float *lpos,
                const float *grad_loss, float *dq, float *dkey,
                const float *patch_rows, int64_t patch_lo, int64_t patch_n,
                void *workspace, size_t workspace_bytes, mi_stream_t stream);

/* CoMatch soft-kNN pseudo-label smoothing.  Replaces track_semi/comatch.py:187-189:
 *   A = exp(feature @ bank_feature^T / T) ; A = A / A.sum(1) ; probs_out = alpha*probs_in + (1-alpha) * A @ bank_prob
 * (plain exp without max subtraction, as the reference) without writing the [b, K] matrix A.
 * feature [b, D], bank_feature [K, D], bank_prob [K, C], probs_in/probs_out [b, C], all float32 row-major.
 * D % 64 == 0, 64 <= D <= 256, 1 <= C <= 255 (MI_ERR_UNSUPPORTED otherwise; the reference's feature_dim is 64). */
size_t mi_soft_knn_workspace_bytes(int64_t b, int64_t K, int64_t D, int64_t C);
int mi_soft_knn(const float *feature, const float *bank_feature, const float *bank_prob, const float *probs_in,
                int64_t b, int64_t K, int64_t D, int64_t C, float temperature, float alpha, float *probs_out,
                void *workspace, size_t workspace_bytes, mi_stream_t stream);

/* mi_queue_push that first copies the rows it overwrites to stash[0 .. w) (w = rows written):
 * the companion of mi_moco_bwd's patch.  Same truncation rule and pointer update
 * (models/memory_bank.py:65-72). */
int mi_queue_push_stash(void *queue, int64_t K, int64_t row_bytes, int64_t ptr,
                        const void *value, int64_t b, void *stash, int64_t *new_ptr, mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_KNN_H */
