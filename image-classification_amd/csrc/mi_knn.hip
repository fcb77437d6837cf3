// mi_knn.hip -- C ABI (include/mi_knn.h) and host-side stage scheduling for gfx950.
//
// Path: models/knn.py:4-21 (similarity GEMM, top-k, vote, ranking),
// models/memory_bank.py:58-72 (truncating ring push), track_ssl/moco.py:152-156 (queue logits).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/mi_knn.h"
#include "moco_kernel.h"
#include "select_kernel.h"
#include "sim_kernel.h"
#include "sim_walk_kernel.h"
#include "vote_kernel.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_CHECK(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(MI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                              \
    } while (0)

// ---- optional per-launch timing (bench.py's roofline leg).  Off by default; when on, every
// similarity-kernel launch is bracketed by HIP events on ITS OWN stream.  Debug facility: guarded
// by a mutex, not meant for concurrent callers.
struct ProfRec { hipEvent_t e0, e1; int kind; double work; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof;

struct ProfScope {
    bool on = false;
    ProfRec r{};
    hipStream_t s;
    ProfScope(int kind, double work, hipStream_t stream) : s(stream)
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (!g_prof_on) return;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        r.kind = kind;
        r.work = work;
        on = hipEventRecord(r.e0, s) == hipSuccess;
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(r.e1, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.push_back(r);
    }
};

inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// A kernel that needs more dynamic LDS than the default limit has that limit raised ONCE per device (bit d of `done`:
// device ordinal d), not on every call: hipFuncSetAttribute is a driver call on the launch path otherwise.
template <typename Kernel>
hipError_t ensure_dynamic_lds(Kernel kernel, size_t bytes, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev >= 0 && dev < 64 ? (uint64_t)1 << dev : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}

// --------------------------------------------------------------------------- small kernels
// pack: float32 source -> compute dtype, K-contiguous rows, zero padded to ld_dst.
template <typename OutT, bool TRANSPOSED>
__global__ __launch_bounds__(256) void pack_kernel(const float *__restrict__ src, long long ld_src,
                                                   long long rows, int D, OutT *__restrict__ dst,
                                                   long long ld_dst)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const long long r0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    if constexpr (!TRANSPOSED) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long r = r0 + ty + i * 8;
            const int c = c0 + tx;
            if (r < rows && c < ld_dst) dst[r * ld_dst + c] = (OutT)(c < D ? src[r * ld_src + c] : 0.f);
        }
    } else {
        // src[c][r]: read with r contiguous, write with c contiguous
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + ty + i * 8;
            const long long r = r0 + tx;
            tile[ty + i * 8][tx] = (c < D && r < rows) ? src[(long long)c * ld_src + r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long r = r0 + ty + i * 8;
            const int c = c0 + tx;
            if (r < rows && c < ld_dst) dst[r * ld_dst + c] = (OutT)tile[tx][ty + i * 8];
        }
    }
}

// pack, fast path: untransposed rows, D % 8 == 0, 16-byte aligned source rows.  One thread moves 8
// consecutive elements (two 16-byte loads; one 16-byte bf16 store or two float stores); HBM-bound.
template <typename OutT>
__global__ __launch_bounds__(256) void pack_rows_vec_kernel(const float *__restrict__ src, long long ld_src,
                                                            long long rows, int D, OutT *__restrict__ dst,
                                                            long long ld_dst)
{
    const long long per_row = ld_dst >> 3;
    const long long total = rows * per_row;
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
        const long long r = e / per_row;
        const int c = (int)(e - r * per_row) << 3;
        typedef __attribute__((ext_vector_type(4))) float f4;
        f4 a = f4{0.f, 0.f, 0.f, 0.f}, b = a;
        if (c < D) {
            const f4 *sp = (const f4 *)(src + r * ld_src + c);
            a = __builtin_nontemporal_load(sp);
            b = __builtin_nontemporal_load(sp + 1);
        }
        if constexpr (sizeof(OutT) == 2) {
            typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
            bf16x8_t o;
            o[0] = (__bf16)a.x; o[1] = (__bf16)a.y; o[2] = (__bf16)a.z; o[3] = (__bf16)a.w;
            o[4] = (__bf16)b.x; o[5] = (__bf16)b.y; o[6] = (__bf16)b.z; o[7] = (__bf16)b.w;
            *(bf16x8_t *)(dst + r * ld_dst + c) = o;
        } else {
            f4 *dp = (f4 *)(dst + r * ld_dst + c);
            dp[0] = a;
            dp[1] = b;
        }
    }
}

// bank construction (ssltrainer.py:172-179): L2-normalise each row and write it straight into the packed
// [N, ld] bank in the compute dtype.  One wave per row; the row is read twice (the second read hits L2).
template <typename OutT>
__global__ __launch_bounds__(256) void append_normalized_kernel(const float *__restrict__ src, long long ld_src,
                                                                long long b, int D, OutT *__restrict__ bank,
                                                                long long ld, long long row0, float eps)
{
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= b) return;
    const float *x = src + r * ld_src;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) ss = fmaf(x[c], x[c], ss);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d, 64);
    const float denom = fmaxf(sqrtf(ss), eps);                      // F.normalize: v / max(||v||_2, eps)
    OutT *o = bank + (row0 + r) * ld;
    for (int c = lane; c < (int)ld; c += 64) o[c] = (OutT)(c < D ? x[c] / denom : 0.f);
}

// Vector path: D % 4 == 0, D <= 2048, 16-byte aligned rows.  The row is read ONCE (8 x float4 per lane stay in
// registers between the norm and the scaled store) and written with 8-/16-byte stores: 2.6 -> ~4.5 TB/s.
template <typename OutT>
__global__ __launch_bounds__(256) void append_normalized_vec_kernel(const float *__restrict__ src, long long ld_src,
                                                                    long long b, int D, OutT *__restrict__ bank,
                                                                    long long ld, long long row0, float eps)
{
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= b) return;
    const float4 *x = (const float4 *)(src + r * ld_src);
    const int nv = D >> 2, nvo = (int)(ld >> 2);
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (j * 64 + lane) < nv ? x[j * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, ss))));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d, 64);
    const float denom = fmaxf(sqrtf(ss), eps);                      // F.normalize: v / max(||v||_2, eps)
    OutT *o = bank + (row0 + r) * ld;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = j * 64 + lane;
        if (c < nvo) {                                              // columns D .. ld are zero padding
            const bool in = c < nv;                                 // padding stays exactly 0 (even for 0 / 0 rows with eps = 0)
            const float a0 = in ? v[j].x / denom : 0.f, a1 = in ? v[j].y / denom : 0.f;
            const float a2 = in ? v[j].z / denom : 0.f, a3 = in ? v[j].w / denom : 0.f;
            if constexpr (sizeof(OutT) == 4) {
                ((float4 *)o)[c] = make_float4(a0, a1, a2, a3);
            } else {
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
                bf16x4_t w = {(__bf16)a0, (__bf16)a1, (__bf16)a2, (__bf16)a3};
                ((bf16x4_t *)o)[c] = w;
            }
        }
    }
}

// fb_*: state of the speculation fallback (null / 0 when the pass does not speculate): the failed-row counter, the safe
// thresholds, and the compact rows' tile masks and key counts, all of which must start at zero / -inf
__global__ void init_state_kernel(float *tau, int *topk_n, unsigned *big_cnt, unsigned *tmask, int mask_words, int B, int B_pad,
                                  float *tau0, unsigned *fail_cnt, unsigned *fb_tmask, int fb_mask_words, int *fb_topk_n, int fb_rows)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B_pad)
        for (int j = 0; j < mask_words; ++j) tmask[(long long)i * mask_words + j] = 0u;
    if (i < 2 * mi::BIG_SUBLISTS) big_cnt[i * 32] = 0u;
    if (i < B_pad) {
        tau[i] = i < B ? -INFINITY : INFINITY;
        topk_n[i] = 0;
        if (tau0) tau0[i] = -INFINITY;
    }
    if (fail_cnt && i == 0) *fail_cnt = 0u;
    if (i < fb_rows) {
        fb_topk_n[i] = 0;
        for (int j = 0; j < fb_mask_words; ++j) fb_tmask[(long long)i * fb_mask_words + j] = 0u;
    }
}

// ring-buffer enqueue body: copy `bytes` (16-byte vector path when aligned)
__global__ void push_kernel(char *__restrict__ dst, const char *__restrict__ src, long long bytes, int vec)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const long long n = bytes >> 4;
        for (; i < n; i += stride) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
    } else {
        for (; i < bytes; i += stride) dst[i] = src[i];
    }
}

// wrapping ring-buffer enqueue (models/module_utils.py:73-85): item row r -> queue row (cursor + r) % K; when the
// batch is longer than the queue (b <= 2K - cursor) the rows both slice assignments write keep the later item row
__global__ void push_wrap_kernel(char *__restrict__ queue, long long K, long long row_bytes, long long cursor,
                                 const char *__restrict__ value, long long b, int vec)
{
    const long long per = vec ? row_bytes >> 4 : row_bytes;
    const long long total = b * per;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i / per, c = i - r * per;
        long long dst = cursor + r;
        if (dst >= K) dst -= K;                       // second slice assignment: data[:left] = item[right:]
        else if (dst < b - (K - cursor)) continue;    // b > K: the second assignment rewrites this row (later item rows win)
        if (vec) ((uint4 *)(queue + dst * row_bytes))[c] = ((const uint4 *)(value + r * row_bytes))[c];
        else queue[dst * row_bytes + c] = value[r * row_bytes + c];
    }
}

// StorageBank.scatter body: bank[index[i]] = value[i] (rows of row_bytes); out-of-range indices are skipped
__global__ void scatter_rows_kernel(char *__restrict__ bank, long long K, long long row_bytes,
                                    const long long *__restrict__ index, const char *__restrict__ value,
                                    long long b, int vec)
{
    const long long per = vec ? row_bytes >> 4 : row_bytes;
    const long long total = b * per;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long long r = i / per, c = i - r * per;
        long long dst = index[r];
        if (dst < 0) dst += K;                       // torch-style negative index
        if (dst < 0 || dst >= K) continue;
        if (vec) ((uint4 *)(bank + dst * row_bytes))[c] = ((const uint4 *)(value + r * row_bytes))[c];
        else bank[dst * row_bytes + c] = value[r * row_bytes + c];
    }
}

// MoCo forward, step 2: fold the per-tile (max, sumexp) partials and the positive logit, and -- in the
// workgroup that finishes last -- the mean over the rows: one launch instead of two (a dependent launch costs
// ~2-10 us of dispatch gap at this size, the whole forward is ~50 us).
// 256 threads = 16 rows x 16 groups; group g folds tiles g, g+16, ... of its row (16 rows of one
// tile = one coalesced 128-byte read) and columns g*D/16.. of q.key, then the groups meet in LDS.
// Hand-off of the per-block sums (MI355X_MICROARCH.md, inter-workgroup visibility, counter form): lane 0 stores its
// block's sum write-through (sc1), waits for the store, then adds to the arrival counter (agent scope, acq_rel); the block
// whose add returns the last ticket reads every block sum with sc1 loads, adds them in block order (deterministic)
// and resets the counter.  The counter is zeroed by the preceding similarity kernel (plain store, kernel boundary).
__global__ __launch_bounds__(256) void moco_combine_kernel(const float *q, const float *key, int b, int D,
                                                          const float2 *part, int nparts, int lse_ld,
                                                          float inv_t, float *lse, float *lpos,
                                                          float *block_sum, unsigned *arrived, float *loss)
{
    __shared__ float s_m[16][17], s_s[16][17], s_d[16][17];
    __shared__ float s_loss[16];
    const int r = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int row = blockIdx.x * 16 + r;
    float m = -INFINITY, sum = 0.f, d = 0.f;
    if (row < b) {
        for (int i = g; i < nparts; i += 16) {
            const float2 ps = part[(long long)i * lse_ld + row];
            if (ps.y > 0.f) {
                const float mn = fmaxf(m, ps.x);
                sum = sum * expf(m - mn) + ps.y * expf(ps.x - mn);
                m = mn;
            }
        }
        for (int c = g; c < D; c += 16) d = fmaf(q[(long long)row * D + c], key[(long long)row * D + c], d);
    }
    s_m[r][g] = m;
    s_s[r][g] = sum;
    s_d[r][g] = d;
    __syncthreads();
    if (g == 0) {
        float rl = 0.f;
        if (row < b) {
            float dd = 0.f;
            for (int i = 0; i < 16; ++i) dd += s_d[r][i];
            const float lp = dd * inv_t;
            float mm = lp;
            for (int i = 0; i < 16; ++i) mm = fmaxf(mm, s_m[r][i]);
            float ss = expf(lp - mm);
            for (int i = 0; i < 16; ++i)
                if (s_s[r][i] > 0.f) ss += s_s[r][i] * expf(s_m[r][i] - mm);
            const float l = mm + logf(ss);
            lse[row] = l;
            lpos[row] = dd;
            rl = l - lp;
        }
        s_loss[r] = rl;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bs = 0.f;
        for (int i = 0; i < 16; ++i) bs += s_loss[i];
        __hip_atomic_store(block_sum + blockIdx.x, bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // sc1: write-through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the store has left this CU before the ticket is drawn
        // acq_rel at agent scope: the arrival RELEASES this block's sum and, in the block that draws the last ticket,
        // ACQUIRES every other block's -- the ordering the HIP memory model asks for; the sc1 store / sc1 loads and the
        // drain above are the hardware form the guide prescribes (cdna_hip_programming.md Guideline 16, pitfall 12: keep both)
        const unsigned ticket = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == gridDim.x - 1) {
            float tot = 0.f;
            for (unsigned i = 0; i < gridDim.x; ++i)
                tot += __hip_atomic_load(block_sum + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 loads, block order
            loss[0] = tot / (float)b;
            __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// --------------------------------------------------------------------------- top-k plan
struct Plan {
    int B_pad, qtiles;
    int s0_cols;        // dense bootstrap stage
    int min_cols;       // smallest useful filter stage (enough workgroups to fill the chip)
    int cap;            // candidate slots per row (>= every stage's columns: overflow impossible)
    int launch_cols;    // columns of one GEMM launch (<= cap): a stage is one or more launches, then ONE refresh
    size_t off_tau, off_topkn, off_big, off_topk, off_mask, off_cnt2, off_cand, fixed, total;
    int mask_words;
    // speculation (see spec_rank below): state of the exact fallback.  fb_rows = 0: the pass does not speculate.
    int spec_rank;      // rank of the speculative first threshold in the sample (== k: none)
    int fb_rows;        // compact rows one fallback round redoes (multiple of 256): fb_rows * N_pad slots fit the candidate buffer
    int fb_rounds;      // ceil(B_pad / fb_rows): every row can be redone, whatever fails
    int fb_ntiles, fb_mask_words;
    size_t off_tau0, off_fail, off_fb_topk, off_fb_topkn, off_fb_mask, off_fb_cnt2, total_spec;
};

constexpr int S0_COLS = 2048;           // dense bootstrap columns
constexpr int BIG_CNT_BYTES = 2 * mi::BIG_SUBLISTS * 128;      // one 128-byte line per counter
// rows (row >> 2) % 64 == s go to sub-list s: at most 4 * ceil(B_pad / 256) of them
inline int big_stride(int B_pad) { return 4 * (int)ceil_div(B_pad, 4 * mi::BIG_SUBLISTS); }
constexpr int STAGE_QUANTUM = 1024;                // filter stages are whole multiples of 4 bank tiles
constexpr int64_t FIT_BYTES = (int64_t)128 << 20;             // bank slice of one launch: stays in the 256 MiB Infinity Cache
constexpr int64_t CAND_BUDGET_BYTES = (int64_t)8 << 30;       // default size of the candidate buffer

// Speculative first threshold.  The safe first threshold -- the k-th largest of the S sampled similarities of a row --
// lets ~k/S of all elements through (10 % at k = 200, S = 2048) until the first stages have raised it, and those
// candidate-dense ramp stages are what a small bank (CIFAR: all of it) and a small row shard pay most for.  The row's
// FINAL k-th similarity sits near rank m = k S / N of the sample, far above rank k.  So the filter starts with the
// r-th largest sample value instead, r << k chosen such that, for a strided sample of an exchangeable bank, fewer than
// k of the N bank rows reach it with probability < 1e-9 per row: the count above the r-th order statistic of S
// uniform draws is N x Gamma(r) / S, and P(Gamma(r) < m) <= m^r e^-m / r! / (1 - m / (r + 1)).  The threshold still
// only rises (the bag's k-th key takes over once it is higher).  Exactness does not rest on the estimate: a row that
// ends the pass with fewer than k keys is redone from its safe threshold (fallback rounds in knn_topk_impl) -- whatever
// the bank looks like, the result is the exact top-k; only the time is statistical.
int spec_rank(int64_t N, int k, int S)
{
    const double m = (double)k * (double)S / (double)N;
    for (int r = 4; r < k; ++r) {
        if ((double)r + 1.0 <= 2.0 * m) continue;                          // keep the series bound tight: m / (r + 1) < 1/2
        const double logp = r * log(m) - m - lgamma((double)r + 1.0) - log(1.0 - m / ((double)r + 1.0));
        if (logp < log(1e-9)) return r;
    }
    return k;
}

// Workspace layout.  cand_bytes < 0: the default candidate budget; otherwise what the caller gave.
Plan make_plan(int64_t B, int64_t N, int k, int64_t cand_bytes, int64_t bank_row_bytes = 0)
{
    Plan pl{};
    pl.B_pad = (int)round_up(B, mi::TILE);
    pl.qtiles = pl.B_pad / mi::TILE;
    pl.s0_cols = (int)(N < S0_COLS ? N : S0_COLS);
    pl.min_cols = (int)round_up(ceil_div(768, pl.qtiles) * mi::TILE, STAGE_QUANTUM);
    // per stage column: 8 bytes of candidate slot per query row + the per-(tile, row) u16 counts (B_pad / 128)
    // (+ one bit per (tile, row) in the tile masks: B_pad / 2048; rounded up into the 1024-byte slack)
    const int64_t row_bytes = (int64_t)pl.B_pad * 8 + pl.B_pad / 128 + (pl.B_pad + 2047) / 2048;
    int64_t cap = ((cand_bytes < 0 ? CAND_BUDGET_BYTES : cand_bytes) - 1024 - 4 * pl.B_pad) / row_bytes / STAGE_QUANTUM * STAGE_QUANTUM;
    int64_t useful = round_up(N > S0_COLS ? N : S0_COLS, STAGE_QUANTUM);        // one stage for the whole bank
    if (useful < S0_COLS) useful = S0_COLS;
    if (cap > useful) cap = useful;
    // Keep one stage's bank slice (+ the query chunk) resident in the 256 MiB Infinity Cache: the
    // slice is re-read once per group of 8 query tiles, and the K loop is latency sensitive (a
    // 268 MB slice ran 12 % slower than a 134 MB one at C4).
    int64_t launch = cap;
    if (bank_row_bytes > 0) {
        int64_t fit = (FIT_BYTES / bank_row_bytes) / STAGE_QUANTUM * STAGE_QUANTUM;
        if (fit < 8 * STAGE_QUANTUM) fit = 8 * STAGE_QUANTUM;
        if (launch > fit) launch = fit;
    }
    if (cap < S0_COLS) cap = cand_bytes < 0 ? S0_COLS : 0;      // 0: the caller's workspace is too small
    if (launch < S0_COLS) launch = cap;
    pl.cap = (int)cap;
    pl.launch_cols = (int)launch;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (size_t)round_up((int64_t)(off + bytes), 256); return o; };
    pl.off_tau = take((size_t)pl.B_pad * 4);
    pl.off_topkn = take((size_t)pl.B_pad * 4);
    pl.off_big = take((size_t)BIG_CNT_BYTES + (size_t)mi::BIG_SUBLISTS * big_stride(pl.B_pad) * 4);   // counters + row sub-lists
    pl.off_topk = take((size_t)pl.B_pad * k * 8);
    pl.fixed = off;
    pl.mask_words = (int)ceil_div(pl.cap / mi::TILE, 32);
    pl.off_mask = take((size_t)pl.B_pad * pl.mask_words * 4);
    pl.off_cnt2 = take((size_t)(pl.cap / mi::TILE) * pl.B_pad * 2);
    pl.off_cand = take((size_t)pl.B_pad * pl.cap * 8);
    pl.total = off;
    // speculation: worth it when the sample is a strict subset of the bank, the refresh is the register-resident one
    // (k <= 256) and at least one 256-row query tile can be redone against the WHOLE bank in one stage
    pl.spec_rank = k;
    pl.fb_rows = pl.fb_rounds = 0;
    pl.total_spec = pl.total;
    const int64_t n_pad = round_up(N, mi::TILE);
    if (N > S0_COLS && k <= mi::WSEL_MAX_K && pl.cap > 0) {
        const int r = spec_rank(N, k, S0_COLS);
        int64_t fb = (int64_t)pl.B_pad * pl.cap / n_pad / mi::TILE * mi::TILE;
        if (fb > pl.B_pad) fb = pl.B_pad;
        if (r < k && fb >= mi::TILE && ceil_div(pl.B_pad, fb) <= 16) {
            pl.spec_rank = r;
            pl.fb_rows = (int)fb;
            pl.fb_rounds = (int)ceil_div(pl.B_pad, fb);
            pl.fb_ntiles = (int)(n_pad / mi::TILE);
            pl.fb_mask_words = (int)ceil_div(pl.fb_ntiles, 32);
            pl.off_tau0 = take((size_t)pl.B_pad * 4);
            pl.off_fail = take(256 + (size_t)pl.B_pad * 4);                 // counter (own line) + failed rows
            pl.off_fb_topk = take((size_t)pl.fb_rows * k * 8);
            pl.off_fb_topkn = take((size_t)pl.fb_rows * 4);
            pl.off_fb_mask = take((size_t)pl.fb_rows * pl.fb_mask_words * 4);
            pl.off_fb_cnt2 = take((size_t)pl.fb_ntiles * pl.fb_rows * 2);
            pl.total_spec = off;
        }
    }
    return pl;
}

// Filter stage schedule.  A stage filters with the threshold tau known at its start (the k-th best of the
// `seen` columns before it), so about k * ln(1 + n / seen) elements per row pass a stage of n columns -- the
// total over a run does not depend on how it is cut into stages, the number of threshold refreshes does.
//   ramp: n = 3 * seen (growth factor 4: small launches run well below the large ones' rate, so the ramp is kept
//         short at the price of ~3k instead of ~k candidates in its stages), up to one cache-fit launch;
//   then: n = seen / 4, at least one launch: ~k ln 1.25 = 0.22 k candidates per row and stage (within the
//         wave-per-row merge's 64) and O(log N) refreshes instead of one per launch;
//   always <= cap: the candidate buffer holds the worst case of one stage (every element passes).
// With the speculative first threshold (spec_rank) the ramp is not needed: that threshold already passes only
// ~spec_rank * n / S elements of a stage, so the first stage runs straight to one cache-fit launch -- a bank that fits one
// (C1: 50 000 rows) is ONE filter launch and ONE refresh instead of three of each (C1 bf16 0.55 -> 0.53 ms, one box; the
// 1.28 M-row bank and its 1/2 ... 1/8 shards: +-0.1 %).
constexpr int LATE_GROWTH_DIV = 4;
constexpr int RAMP_SAFE = 3, RAMP_SPEC = 63;
int64_t next_stage_cols(const Plan &pl, int64_t seen, int64_t left, int ramp)      // left: bank rows not filtered yet
{
    int64_t n = ramp * seen / STAGE_QUANTUM * STAGE_QUANTUM;
    if (n < pl.min_cols) n = pl.min_cols;
    if (n > pl.launch_cols) {
        n = seen / LATE_GROWTH_DIV / pl.launch_cols * pl.launch_cols;      // whole launches
        if (n < pl.launch_cols) n = pl.launch_cols;
    }
    if (n > pl.cap) n = pl.cap;
    if (n > left) n = left;
    return n;
}

// mi_debug_force_fallback: rows with row % modulus == 0 are reported as failed speculations (tests of the fallback path)
std::atomic<int> g_force_mod{0};

// mi_profile_clock: when a stamp buffer is registered, filter launches run the STAMP instantiation of the kernel
std::mutex g_stamp_mu;
unsigned long long *g_stamp_buf = nullptr;
int g_stamp_cap = 0;

template <int DT, int MODE, bool FB = false>
hipError_t launch_sim(const mi::SimParams &p_in, hipStream_t s)
{
    // 32 consecutive same-XCD workgroups form one patch of (1 << qsh) query tiles x (32 >> qsh) bank
    // tiles.  The patch is as tall as the launch has query tiles (at most 8): a launch with few
    // query tiles (MoCo: one) must not pad patches with workgroups that only exit -- they still
    // queue for a CU with 129 KiB of free LDS and delayed the real ones by ~100 us at C2.
    mi::SimParams p = p_in;
    int qsh = 3;
    while (qsh > 0 && (1 << (qsh - 1)) >= p.qtiles) --qsh;
    p.qsh = qsh;
    p.pq = (int)ceil_div(p.qtiles, 1 << qsh);
    p.pn = (int)ceil_div(p.ntiles, 32 >> qsh);
    const int patches = p.pq * p.pn;
    const int grid = (int)ceil_div(patches, 8) * 256;
    if constexpr (FB) {
        hipLaunchKernelGGL((mi::sim_kernel<DT, MODE, false, true>), dim3(grid), dim3(mi::SIM_THREADS), 0, s, p);
        return hipGetLastError();
    } else if constexpr (MODE == mi::MODE_FILTER) {
        std::lock_guard<std::mutex> lk(g_stamp_mu);
        if (g_stamp_buf) {
            p.stamps = g_stamp_buf;
            p.stamp_cap = g_stamp_cap;
            hipLaunchKernelGGL((mi::sim_kernel<DT, MODE, true>), dim3(grid), dim3(mi::SIM_THREADS), 0, s, p);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((mi::sim_kernel<DT, MODE>), dim3(grid), dim3(mi::SIM_THREADS), 0, s, p);
    return hipGetLastError();
}

// Filter launch for short rows (<= 2 K-steps: D <= 128 bf16, D <= 64 f32): sim_walk_kernel.h.  One workgroup per CU
// (130 KiB of LDS), so the bank tiles of the launch are cut into runs such that qtiles x runs fills whole rounds of CUs.
int compute_units()
{
    static std::atomic<int> cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

hipError_t launch_walk(int dtype, const mi::SimParams &p_in, bool compact, hipStream_t s)
{
    mi::SimParams p = p_in;
    p.walk_compact = compact ? 1 : 0;
    const int cus = compute_units();
    // tiles per run: the run length that minimises rounds x (run + ~1.5 tiles of per-workgroup set-up); a compacted run's
    // count has to fit cnt2's 16 bits
    double best = 1e30;
    const int tp_max = p.ntiles < 255 ? p.ntiles : 255;
    int best_tp = tp_max;
    for (int tp = tp_max; tp >= 1; --tp) {
        const int64_t nwg = (int64_t)p.qtiles * ceil_div(p.ntiles, tp);
        const double cost = (double)ceil_div(nwg, cus) * (tp + 1.5);
        if (cost < best) { best = cost; best_tp = tp; }
    }
    p.walk_tps = best_tp;
    p.walk_nsplit = (int)ceil_div(p.ntiles, best_tp);
    const int64_t nwg = (int64_t)p.qtiles * p.walk_nsplit;
    const dim3 grid((unsigned)(ceil_div(nwg, 8) * 8));
    // the query tile (ksteps x 32 KiB) stays resident in LDS.  (The variant that streams both operands, for rows of
    // 257 .. 512 bytes, measured 3 % SLOWER than sim_kernel.h at f32 D = 128 and is not instantiated.)
    if (dtype == MI_BF16) hipLaunchKernelGGL((mi::sim_walk_kernel<mi::DT_BF16>), grid, dim3(mi::SIM_THREADS), 0, s, p);
    else hipLaunchKernelGGL((mi::sim_walk_kernel<mi::DT_F32>), grid, dim3(mi::SIM_THREADS), 0, s, p);
    return hipGetLastError();
}
constexpr int WALK_MAX_KSTEPS = 2;      // rows of at most 256 bytes: bf16 D <= 128, f32 D <= 64

template <int MODE>
hipError_t launch_sim_dt(int dtype, const mi::SimParams &p, hipStream_t s)
{
    return dtype == MI_BF16 ? launch_sim<mi::DT_BF16, MODE>(p, s) : launch_sim<mi::DT_F32, MODE>(p, s);
}

int elt_bytes(int dtype) { return dtype == MI_BF16 ? 2 : 4; }

}  // namespace

// =========================================================================== C ABI
extern "C" {

int mi_abi_version(void) { return MI_ABI_VERSION; }
const char *mi_last_error(void) { return g_err; }

int mi_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    if (!g_prof_on) {
        for (auto &r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        g_prof.clear();
    }
    return MI_OK;
}

int mi_profile_collect(int kind, double *total_ms, int64_t *launches, double *total_work)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0.0, work = 0.0;
    int64_t n = 0;
    std::vector<ProfRec> keep;
    for (auto &r : g_prof) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        float t = 0.f;
        HIP_CHECK(hipEventSynchronize(r.e1));
        HIP_CHECK(hipEventElapsedTime(&t, r.e0, r.e1));
        ms += t;
        work += r.work;
        ++n;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.swap(keep);
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (total_work) *total_work = work;
    return MI_OK;
}

int mi_debug_force_fallback(int modulus)
{
    if (modulus < 0) return fail(MI_ERR_ARG, "mi_debug_force_fallback: modulus must be >= 0");
    g_force_mod.store(modulus);
    return MI_OK;
}

int mi_profile_clock(void *stamps, int64_t entries)
{
    if (stamps && (entries < 1 || entries > INT32_MAX || ((uintptr_t)stamps & 15))) return fail(MI_ERR_ARG, "mi_profile_clock: bad buffer");
    std::lock_guard<std::mutex> lk(g_stamp_mu);
    g_stamp_buf = (unsigned long long *)stamps;
    g_stamp_cap = stamps ? (int)entries : 0;
    return MI_OK;
}

int64_t mi_knn_ld(int64_t D, int dtype)
{
    const int per = mi::KSTEP_BYTES / elt_bytes(dtype);
    return round_up(D, per);
}

int mi_pack_rows(const float *src, int64_t ld_src, int transposed, int64_t rows, int64_t D,
                 int dtype, void *dst, int64_t ld_dst, mi_stream_t stream)
{
    if (!src || !dst || rows <= 0 || D <= 0) return fail(MI_ERR_ARG, "mi_pack_rows: null pointer or empty shape");
    if (dtype != MI_F32 && dtype != MI_BF16) return fail(MI_ERR_ARG, "mi_pack_rows: bad dtype %d", dtype);
    if (ld_dst < D || ld_src < (transposed ? rows : D)) return fail(MI_ERR_ARG, "mi_pack_rows: leading dimension too small");
    if (D > INT32_MAX / 2 || ld_dst > INT32_MAX / 2) return fail(MI_ERR_ARG, "mi_pack_rows: D too large");
    hipStream_t s = (hipStream_t)stream;
    if (rows > (int64_t)INT32_MAX * 16 || ld_dst > 65535 * 32) return fail(MI_ERR_ARG, "mi_pack_rows: shape too large");
    if (!transposed && D % 8 == 0 && ld_dst % 8 == 0 && ld_src % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const long long items = rows * (ld_dst / 8);
        const unsigned vgrid = (unsigned)(ceil_div(items, 256) < 256 * 64 ? ceil_div(items, 256) : 256 * 64);
        if (dtype == MI_BF16) hipLaunchKernelGGL((pack_rows_vec_kernel<__bf16>), dim3(vgrid), dim3(256), 0, s, src, ld_src, rows, (int)D, (__bf16 *)dst, ld_dst);
        else hipLaunchKernelGGL((pack_rows_vec_kernel<float>), dim3(vgrid), dim3(256), 0, s, src, ld_src, rows, (int)D, (float *)dst, ld_dst);
        HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
    dim3 grid((unsigned)ceil_div(rows, 32), (unsigned)ceil_div(ld_dst, 32));
    if (dtype == MI_BF16) {
        if (transposed) hipLaunchKernelGGL((pack_kernel<__bf16, true>), grid, dim3(256), 0, s, src, ld_src, rows, (int)D, (__bf16 *)dst, ld_dst);
        else hipLaunchKernelGGL((pack_kernel<__bf16, false>), grid, dim3(256), 0, s, src, ld_src, rows, (int)D, (__bf16 *)dst, ld_dst);
    } else {
        if (transposed) hipLaunchKernelGGL((pack_kernel<float, true>), grid, dim3(256), 0, s, src, ld_src, rows, (int)D, (float *)dst, ld_dst);
        else hipLaunchKernelGGL((pack_kernel<float, false>), grid, dim3(256), 0, s, src, ld_src, rows, (int)D, (float *)dst, ld_dst);
    }
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_bank_append_normalized(const float *src, int64_t ld_src, int64_t b, int64_t D, int dtype,
                              void *bank, int64_t ld, int64_t n_rows, int64_t row0, float eps, mi_stream_t stream)
{
    if (!src || !bank) return fail(MI_ERR_ARG, "mi_bank_append_normalized: null pointer");
    if (dtype != MI_F32 && dtype != MI_BF16) return fail(MI_ERR_ARG, "mi_bank_append_normalized: bad dtype %d", dtype);
    if (b < 0 || D < 1 || ld_src < D || D > INT32_MAX / 2) return fail(MI_ERR_ARG, "mi_bank_append_normalized: bad shape b=%lld D=%lld ld_src=%lld", (long long)b, (long long)D, (long long)ld_src);
    if (ld != mi_knn_ld(D, dtype)) return fail(MI_ERR_ARG, "mi_bank_append_normalized: ld must equal mi_knn_ld(D)=%lld (got %lld)", (long long)mi_knn_ld(D, dtype), (long long)ld);
    if (row0 < 0 || row0 + b > n_rows) return fail(MI_ERR_ARG, "mi_bank_append_normalized: rows [%lld, %lld) do not fit a bank of %lld rows", (long long)row0, (long long)(row0 + b), (long long)n_rows);
    if (!(eps >= 0.f)) return fail(MI_ERR_ARG, "mi_bank_append_normalized: eps must be non-negative");
    if (b == 0) return MI_OK;
    const dim3 grid((unsigned)ceil_div(b, 4));
    const bool vec = D % 4 == 0 && ld <= 2048 && ld_src % 4 == 0 && (((uintptr_t)src | (uintptr_t)bank) & 15) == 0;   // ld % 32 == 0 already
    if (vec) {
        if (dtype == MI_BF16) hipLaunchKernelGGL((append_normalized_vec_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, b, (int)D, (__bf16 *)bank, ld, row0, eps);
        else hipLaunchKernelGGL((append_normalized_vec_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, b, (int)D, (float *)bank, ld, row0, eps);
    } else if (dtype == MI_BF16) hipLaunchKernelGGL((append_normalized_kernel<__bf16>), grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, b, (int)D, (__bf16 *)bank, ld, row0, eps);
    else hipLaunchKernelGGL((append_normalized_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, b, (int)D, (float *)bank, ld, row0, eps);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

size_t mi_knn_topk_workspace_bytes(int64_t B, int64_t N, int k)
{
    return mi_knn_topk_workspace_bytes_for_budget(B, N, k, (size_t)CAND_BUDGET_BYTES);
}

size_t mi_knn_topk_workspace_bytes_for_budget(int64_t B, int64_t N, int k, size_t candidate_bytes)
{
    if (B <= 0 || N <= 0 || k <= 0 || k > MI_MAX_K || B > (1 << 30) || N > INT32_MAX - 2 * mi::TILE) return 0;
    const int64_t budget = candidate_bytes > ((size_t)1 << 46) ? ((int64_t)1 << 46) : (int64_t)candidate_bytes;
    Plan pl = make_plan(B, N, k, budget);
    if (pl.cap == 0) pl = make_plan(B, N, k, -1);             // below the minimum: the default (it covers the minimum)
    return pl.total_spec + 2048 + 4 * (size_t)round_up(B, mi::TILE);   // + the alignment slack make_plan reserves when it sizes from a byte count
}

}  // extern "C"

namespace {
// Body of mi_knn_topk and mi_knn_topk_static: the two differ only in where the workspace lives.
int knn_topk_impl(const void *q, int64_t ldq, const void *bank, int64_t ldb, int dtype,
                  int64_t B, int64_t N, int64_t D, int k, int64_t idx_base,
                  float *out_val, int64_t *out_idx,
                  void *workspace, size_t workspace_bytes, mi_stream_t stream)
{
    if (!q || !bank || !out_val || !out_idx || !workspace) return fail(MI_ERR_ARG, "mi_knn_topk: null pointer");
    if (dtype != MI_F32 && dtype != MI_BF16) return fail(MI_ERR_ARG, "mi_knn_topk: bad dtype %d", dtype);
    if (B <= 0 || N <= 0 || D <= 0) return fail(MI_ERR_ARG, "mi_knn_topk: empty shape B=%lld N=%lld D=%lld", (long long)B, (long long)N, (long long)D);
    if (B > (1 << 30) || N > INT32_MAX - 2 * mi::TILE) return fail(MI_ERR_UNSUPPORTED, "mi_knn_topk: B or N too large for one shard");
    if (k < 1 || k > MI_MAX_K) return fail(MI_ERR_UNSUPPORTED, "mi_knn_topk: k=%d outside [1, %d]", k, MI_MAX_K);
    if (k > N) return fail(MI_ERR_ARG, "mi_knn_topk: selected index k out of range (k=%d > N=%lld)", k, (long long)N);
    const int64_t ld = mi_knn_ld(D, dtype);
    if (ldq != ld || ldb != ld) return fail(MI_ERR_ARG, "mi_knn_topk: ldq/ldb must equal mi_knn_ld(D)=%lld (got %lld, %lld)", (long long)ld, (long long)ldq, (long long)ldb);
    if (((uintptr_t)q | (uintptr_t)bank | (uintptr_t)workspace) & 15) return fail(MI_ERR_ARG, "mi_knn_topk: q, bank and workspace must be 16-byte aligned");
    // the stage size follows the workspace the caller actually gave (more memory = fewer, larger stages)
    const size_t fixed = make_plan(B, N, k, -1).fixed;
    Plan pl = make_plan(B, N, k, workspace_bytes > fixed ? (int64_t)(workspace_bytes - fixed) : 0, ld * elt_bytes(dtype));
    // speculation is opportunistic: used when the state of its fallback fits the workspace as well (the result never
    // depends on it); a workspace sized by mi_knn_topk_workspace_bytes* always has the room
    bool spec = pl.cap > 0 && pl.fb_rows > 0 && pl.total_spec <= workspace_bytes;
    if (!spec && pl.cap > 0 && pl.fb_rows > 0) {
        const size_t extra = pl.total_spec - pl.total + 4096;
        if (workspace_bytes > fixed + extra) {
            const Plan p2 = make_plan(B, N, k, (int64_t)(workspace_bytes - fixed - extra), ld * elt_bytes(dtype));
            if (p2.cap > 0 && p2.fb_rows > 0 && p2.total_spec <= workspace_bytes) { pl = p2; spec = true; }
        }
    }
    if (pl.cap == 0 || workspace_bytes < pl.total)
        return fail(MI_ERR_WORKSPACE, "mi_knn_topk: workspace %zu is below the minimum %zu (recommended: mi_knn_topk_workspace_bytes)",
                    workspace_bytes, fixed + 2048 + 4 * (size_t)pl.B_pad + (size_t)S0_COLS * ((size_t)pl.B_pad * 8 + pl.B_pad / 128 + 1));

    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    float *tau = (float *)(ws + pl.off_tau);
    unsigned short *cnt2 = (unsigned short *)(ws + pl.off_cnt2);
    unsigned *tmask = (unsigned *)(ws + pl.off_mask);
    int *topk_n = (int *)(ws + pl.off_topkn);
    unsigned long long *topk = (unsigned long long *)(ws + pl.off_topk);
    uint2 *cand = (uint2 *)(ws + pl.off_cand);

    unsigned *big_cnt = (unsigned *)(ws + pl.off_big);
    int *big_rows = (int *)(ws + pl.off_big + BIG_CNT_BYTES);
    float *tau0 = spec ? (float *)(ws + pl.off_tau0) : nullptr;
    unsigned *fail_cnt = spec ? (unsigned *)(ws + pl.off_fail) : nullptr;
    int *fail_rows = spec ? (int *)(ws + pl.off_fail + 256) : nullptr;
    unsigned *fb_mask = spec ? (unsigned *)(ws + pl.off_fb_mask) : nullptr;
    int *fb_topkn = spec ? (int *)(ws + pl.off_fb_topkn) : nullptr;
    hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)ceil_div(pl.B_pad, 256)), dim3(256), 0, s, tau, topk_n, big_cnt, tmask, pl.mask_words, (int)B, pl.B_pad,
                       tau0, fail_cnt, fb_mask, spec ? pl.fb_mask_words : 0, fb_topkn, spec ? pl.fb_rows : 0);
    HIP_CHECK(hipGetLastError());

    mi::SimParams sp{};
    sp.q = (const char *)q;
    sp.x = (const char *)bank;
    sp.ldq_bytes = ldq * elt_bytes(dtype);
    sp.ldx_bytes = ldb * elt_bytes(dtype);
    sp.B = (int)B;
    sp.N = (int)N;
    sp.ksteps = (int)(ld * elt_bytes(dtype) / mi::KSTEP_BYTES);
    sp.qtiles = pl.qtiles;
    sp.tau = tau;
    sp.cnt2 = cnt2;
    sp.tmask = tmask;
    sp.mask_words = pl.mask_words;
    sp.bpad = pl.B_pad;
    sp.cand = cand;
    sp.cap = pl.cap;

    mi::SelectParams sel{};
    sel.cand = cand;
    sel.cnt2 = cnt2;
    sel.tmask = tmask;
    sel.mask_words = pl.mask_words;
    sel.bpad = pl.B_pad;
    sel.cap = pl.cap;
    sel.B = B;
    sel.topk = topk;
    sel.topk_n = topk_n;
    sel.tau = tau;
    sel.k = k;
    sel.idx_base = idx_base;
    sel.tau0 = tau0;
    sel.spec_rank = spec ? pl.spec_rank : k;
    sel.fail_cnt = fail_cnt;
    sel.fail_rows = fail_rows;
    sel.force_mod = spec ? g_force_mod.load() : 0;

    // stage 0: dense bootstrap.  A bank no larger than the bootstrap is finished by it.  Otherwise the bootstrap
    // runs over a STRIDED SAMPLE of the bank (every floor(N / s0)-th row: whatever order the bank is in -- an
    // ImageFolder memo loader hands it over class by class, ssltrainer.py:226-230 -- the sample sees every part
    // of it) and only its k-th largest value is kept: a lower bound of the row's final k-th similarity, i.e. a
    // valid first threshold.  Its keys are dropped, so every bank row, the sampled ones included, passes through
    // the filter exactly once and the kept keys stay distinct.
    const bool sampled = N > pl.s0_cols;
    sp.col0 = 0;
    sp.ncols = pl.s0_cols;
    sp.ntiles = (int)ceil_div(pl.s0_cols, mi::TILE);
    sp.row_stride = sampled ? (int)(N / pl.s0_cols) : 1;
    {
        ProfScope ps(MI_PROF_SIM, 2.0 * (double)B * sp.ncols * (double)D, s);
        HIP_CHECK((launch_sim_dt<mi::MODE_DENSE>(dtype, sp, s)));
    }
    int64_t col = sampled ? 0 : pl.s0_cols;          // bank rows filtered (or finished) so far
    sel.dense_n = pl.s0_cols;
    sel.dense_col0 = 0;
    if (col >= N) { sel.out_val = out_val; sel.out_idx = (long long *)out_idx; }
    {
        ProfScope ps(MI_PROF_SELECT, (double)B * sel.dense_n * 4.0, s);
        if (sampled) hipLaunchKernelGGL((mi::sample_tau_kernel<S0_COLS / 64>), dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, sel);
        else if (k <= mi::WSEL_MAX_K) hipLaunchKernelGGL((mi::wave_select_kernel<mi::SRC_DENSE>), dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, s, sel);
        else hipLaunchKernelGGL((mi::select_kernel<mi::SRC_DENSE>), dim3((unsigned)B), dim3(mi::SEL_THREADS), 0, s, sel);
        HIP_CHECK(hipGetLastError());
    }

    // filter stages
    sel.big_rows = big_rows;
    sel.big_cnt = big_cnt;
    sel.big_stride = big_stride(pl.B_pad);
    for (int stage = 0; col < N; ++stage) {
        sel.parity = stage & 1;
        const int64_t ncols = next_stage_cols(pl, col + (sampled ? pl.s0_cols : 0), N - col,      // the sample counts as "seen"
                                              spec ? RAMP_SPEC : RAMP_SAFE);
        sel.ntiles = (int)ceil_div(ncols, mi::TILE);
        for (int64_t off = 0; off < ncols; off += pl.launch_cols) {          // launch_cols is a whole number of tiles
            const int64_t lc = ncols - off < pl.launch_cols ? ncols - off : pl.launch_cols;
            sp.col0 = (int)(col + off);
            sp.ncols = (int)lc;
            sp.ntiles = (int)ceil_div(lc, mi::TILE);
            sp.nt_base = (int)(off / mi::TILE);
            ProfScope ps(MI_PROF_SIM, 2.0 * (double)B * (double)lc * (double)D, s);
            if (sp.ksteps <= WALK_MAX_KSTEPS && sp.cap < (1 << 24)) HIP_CHECK(launch_walk(dtype, sp, k <= mi::WSEL_MAX_K, s));   // (24-bit segment offsets)
            else HIP_CHECK((launch_sim_dt<mi::MODE_FILTER>(dtype, sp, s)));
        }
        col += ncols;
        if (col >= N) { sel.out_val = out_val; sel.out_idx = (long long *)out_idx; }
        {
            // threshold refresh.  k <= 256: one wave per row, selection in registers (bag_refresh_kernel).
            // k > 256: wave-per-row counting merge for rows with few candidates, workgroup select for the rest.
            ProfScope ps(MI_PROF_SELECT, (double)B * k * 16.0, s);
            const dim3 mgrid((unsigned)ceil_div(B, 4));
            if (k <= mi::WSEL_MAX_K) {
                switch ((k + 63) / 64) {
                case 1: hipLaunchKernelGGL((mi::bag_refresh_kernel<1>), mgrid, dim3(256), 0, s, sel); break;
                case 2: hipLaunchKernelGGL((mi::bag_refresh_kernel<2>), mgrid, dim3(256), 0, s, sel); break;
                case 3: hipLaunchKernelGGL((mi::bag_refresh_kernel<3>), mgrid, dim3(256), 0, s, sel); break;
                default: hipLaunchKernelGGL((mi::bag_refresh_kernel<4>), mgrid, dim3(256), 0, s, sel); break;
                }
                HIP_CHECK(hipGetLastError());
            } else {
                sel.skip_small = 1;
                if (k <= 512) hipLaunchKernelGGL((mi::merge_kernel<8>), mgrid, dim3(256), 0, s, sel);
                else hipLaunchKernelGGL((mi::merge_kernel<16>), mgrid, dim3(256), 0, s, sel);
                HIP_CHECK(hipGetLastError());
                hipLaunchKernelGGL((mi::select_kernel<mi::SRC_LIST>), dim3(2048), dim3(mi::SEL_THREADS), 0, s, sel);
                HIP_CHECK(hipGetLastError());
            }
        }
    }
    if (!spec) return MI_OK;

    // Fallback of the speculative first threshold (spec_rank above).  The final refresh listed the rows that ended with
    // fewer than k keys; each round redoes up to fb_rows of them EXACTLY: one filter stage over the whole bank from the
    // row's safe sample threshold (the candidate buffer holds fb_rows x N slots: nothing can overflow), then one
    // refresh that writes the row's final top-k.  Nothing has failed in the normal case, and both launches of a round
    // are empty: every workgroup reads the counter and exits.  No host synchronisation either way.
    const int64_t n_pad = round_up(N, mi::TILE);
    mi::SimParams fp = sp;
    fp.B = pl.fb_rows;
    fp.qtiles = pl.fb_rows / mi::TILE;
    fp.cnt2 = (unsigned short *)(ws + pl.off_fb_cnt2);
    fp.tmask = fb_mask;
    fp.mask_words = pl.fb_mask_words;
    fp.bpad = pl.fb_rows;
    fp.cap = (int)n_pad;
    fp.fb_rows = fail_rows;
    fp.fb_cnt = fail_cnt;
    fp.fb_tau0 = tau0;
    mi::SelectParams fs = sel;
    fs.cnt2 = fp.cnt2;
    fs.tmask = fb_mask;
    fs.mask_words = pl.fb_mask_words;
    fs.bpad = pl.fb_rows;
    fs.cap = (int)n_pad;
    fs.ntiles = pl.fb_ntiles;
    fs.B = pl.fb_rows;
    fs.topk = (unsigned long long *)(ws + pl.off_fb_topk);
    fs.topk_n = fb_topkn;
    fs.out_val = out_val;
    fs.out_idx = (long long *)out_idx;
    fs.tau0 = nullptr;
    fs.fail_cnt = nullptr;              // a redone row is final
    fs.force_mod = 0;
    fs.fb_rows = fail_rows;
    fs.fb_cnt = fail_cnt;
    for (int round = 0; round < pl.fb_rounds; ++round) {
        fp.fb_base = fs.fb_base = round * pl.fb_rows;
        // ONE launch over the whole bank (no cache-fit slicing for a path that is empty but for pathological banks:
        // an empty launch costs ~2 us, and a sliced round would be dozens of them)
        fp.col0 = 0;
        fp.ncols = (int)N;
        fp.ntiles = pl.fb_ntiles;
        fp.nt_base = 0;
        if (dtype == MI_BF16) HIP_CHECK((launch_sim<mi::DT_BF16, mi::MODE_FILTER, true>(fp, s)));
        else HIP_CHECK((launch_sim<mi::DT_F32, mi::MODE_FILTER, true>(fp, s)));
        const dim3 fgrid((unsigned)ceil_div(pl.fb_rows, 4));
        switch ((k + 63) / 64) {
        case 1: hipLaunchKernelGGL((mi::bag_refresh_kernel<1>), fgrid, dim3(256), 0, s, fs); break;
        case 2: hipLaunchKernelGGL((mi::bag_refresh_kernel<2>), fgrid, dim3(256), 0, s, fs); break;
        case 3: hipLaunchKernelGGL((mi::bag_refresh_kernel<3>), fgrid, dim3(256), 0, s, fs); break;
        default: hipLaunchKernelGGL((mi::bag_refresh_kernel<4>), fgrid, dim3(256), 0, s, fs); break;
        }
        HIP_CHECK(hipGetLastError());
    }
    return MI_OK;
}

// ---- scratch of mi_knn_topk_static: device memory that belongs to the code object, one copy per GPU.
// In host code the array's NAME is only a handle for hipGetSymbolAddress; the address kernels and
// memsets may use is the one that call returns, resolved once per device and kept here.
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)MI_KNN_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[MAX_DEVICES] = {};

int static_scratch(void **out)
{
    int dev = -1;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(MI_ERR_UNSUPPORTED, "mi_knn_topk_static: device ordinal %d outside [0, %d)", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));          // this device's copy of the array
        if (!ptr) return fail(MI_ERR_HIP, "mi_knn_topk_static: hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return MI_OK;
}
}  // namespace

extern "C" {

int mi_knn_topk(const void *q, int64_t ldq, const void *bank, int64_t ldb, int dtype,
                int64_t B, int64_t N, int64_t D, int k, int64_t idx_base,
                float *out_val, int64_t *out_idx,
                void *workspace, size_t workspace_bytes, mi_stream_t stream)
{
    return knn_topk_impl(q, ldq, bank, ldb, dtype, B, N, D, k, idx_base, out_val, out_idx, workspace, workspace_bytes, stream);
}

int mi_knn_topk_static(const void *q, int64_t ldq, const void *bank, int64_t ldb, int dtype,
                       int64_t B, int64_t N, int64_t D, int k, int64_t idx_base,
                       float *out_val, int64_t *out_idx, mi_stream_t stream)
{
    if (B <= 0 || N <= 0 || k < 1 || k > MI_MAX_K || B > (1 << 30) || N > INT32_MAX - 2 * mi::TILE)
        return fail(MI_ERR_ARG, "mi_knn_topk_static: bad shape B=%lld N=%lld k=%d", (long long)B, (long long)N, k);
    // the smallest workspace mi_knn_topk accepts for this shape must fit the fixed scratch array
    const size_t fixed = make_plan(B, N, k, -1).fixed;
    const Plan pl = make_plan(B, N, k, STATIC_SCRATCH_BYTES > fixed ? (int64_t)(STATIC_SCRATCH_BYTES - fixed) : 0);
    if (pl.cap == 0 || pl.total > STATIC_SCRATCH_BYTES)
        return fail(MI_ERR_WORKSPACE, "mi_knn_topk_static: B=%lld queries with k=%d need more than the %zu-byte static scratch "
                    "(about 16.4 KiB + 8 k bytes per query row, rows rounded up to 256); use mi_knn_topk with a caller workspace",
                    (long long)B, k, STATIC_SCRATCH_BYTES);
    void *ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != MI_OK) return rc;
    // every word the kernels expect to start at zero is cleared by init_state_kernel at the top of the pass,
    // so the previous call's contents (this GPU's array keeps them) never leak into this one
    return knn_topk_impl(q, ldq, bank, ldb, dtype, B, N, D, k, idx_base, out_val, out_idx, ws, STATIC_SCRATCH_BYTES, stream);
}

int mi_knn_merge(const float *cand_val, const int64_t *cand_idx, int G, int64_t B, int kc, int k,
                 float *out_val, int64_t *out_idx, mi_stream_t stream)
{
    if (!cand_val || !cand_idx || !out_val || !out_idx) return fail(MI_ERR_ARG, "mi_knn_merge: null pointer");
    if (G < 1 || B < 1 || kc < 1 || k < 1 || k > MI_MAX_K) return fail(MI_ERR_ARG, "mi_knn_merge: bad shape G=%d B=%lld kc=%d k=%d", G, (long long)B, kc, k);
    if ((int64_t)G * kc > 16384) return fail(MI_ERR_UNSUPPORTED, "mi_knn_merge: G*kc=%lld > 16384", (long long)G * kc);
    if (k > (int64_t)G * kc) return fail(MI_ERR_ARG, "mi_knn_merge: k=%d exceeds the %lld candidates per row", k, (long long)G * kc);
    mi::SelectParams sel{};
    sel.g_val = cand_val;
    sel.g_idx = (const long long *)cand_idx;
    sel.G = G;
    sel.kc = kc;
    sel.B = B;
    sel.k = k;
    sel.out_val = out_val;
    sel.out_idx = (long long *)out_idx;
    if (k <= mi::WSEL_MAX_K && (int64_t)G * kc <= 64 * mi::MERGE_WAVE_SLOTS) {    // every configuration of the sharded monitor (k = 200, G <= 8)
        if ((int64_t)G * kc <= 64 * mi::MERGE_WAVE_SLOTS_SMALL)
            hipLaunchKernelGGL((mi::merge_wave_kernel<mi::MERGE_WAVE_SLOTS_SMALL, 4>), dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, sel);
        else
            hipLaunchKernelGGL((mi::merge_wave_kernel<mi::MERGE_WAVE_SLOTS, 3>), dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, (hipStream_t)stream, sel);
    } else
        hipLaunchKernelGGL((mi::select_kernel<mi::SRC_GATHER>), dim3((unsigned)B), dim3(mi::SEL_THREADS), 0, (hipStream_t)stream, sel);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_kth_largest(const float *x, int64_t n, int64_t k, float *out, mi_stream_t stream)
{
    if (!x || !out) return fail(MI_ERR_ARG, "mi_kth_largest: null pointer");
    if (n < 1 || n > INT32_MAX) return fail(MI_ERR_ARG, "mi_kth_largest: n=%lld outside [1, 2^31)", (long long)n);
    if (k < 1 || k > n) return fail(MI_ERR_ARG, "mi_kth_largest: selected index k out of range (k=%lld, n=%lld)", (long long)k, (long long)n);
    if (n <= mi::KTH_REG_MAX_N && ((uintptr_t)x & 15u) == 0)      // 16-byte loads
        hipLaunchKernelGGL(mi::kth_largest_reg_kernel, dim3(1), dim3(mi::KTH_THREADS), 0, (hipStream_t)stream, x, (int)n, (int)k, out);
    else
        hipLaunchKernelGGL(mi::kth_largest_kernel, dim3(1), dim3(mi::KTH_THREADS), 0, (hipStream_t)stream, x, (long long)n, (long long)k, out);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_knn_vote(const float *val, const int64_t *idx, const int64_t *labels, int64_t n_labels,
                int64_t B, int k, int classes, float knn_t,
                float *out_scores, int64_t *out_rank, mi_stream_t stream)
{
    if (!val || !idx || !labels || !out_rank) return fail(MI_ERR_ARG, "mi_knn_vote: null pointer");
    if (B < 1 || n_labels < 1) return fail(MI_ERR_ARG, "mi_knn_vote: empty shape");
    if (k < 1 || k > MI_MAX_K) return fail(MI_ERR_UNSUPPORTED, "mi_knn_vote: k=%d outside [1, %d]", k, MI_MAX_K);
    if (classes < 1 || classes > MI_MAX_CLASSES) return fail(MI_ERR_UNSUPPORTED, "mi_knn_vote: classes=%d outside [1, %d]", classes, MI_MAX_CLASSES);
    if (!(knn_t > 0.f)) return fail(MI_ERR_ARG, "mi_knn_vote: temperature must be positive");
    mi::VoteParams vp{B, val, (const long long *)idx, (const long long *)labels, n_labels, k, classes, knn_t, out_scores, (long long *)out_rank};
    if (k <= 256 && classes <= mi::VOTE_WAVE_MAX_C) {
        // wave-per-row LDS histogram (every configuration of the reference: k = 200, 10 / 100 / 1000 classes)
        const size_t lds = mi::vote_wave_lds_bytes(classes);
        hipLaunchKernelGGL(mi::vote_wave_kernel, dim3((unsigned)ceil_div(B, 4)), dim3(256), lds, (hipStream_t)stream, vp);
        HIP_CHECK(hipGetLastError());
        return MI_OK;
    }
    const size_t dyn = (size_t)round_up((int64_t)classes * 2, 16);
    if (dyn > 48 * 1024) {
        static std::atomic<uint64_t> done{0};
        HIP_CHECK(ensure_dynamic_lds(mi::vote_kernel, (size_t)MI_MAX_CLASSES * 2, done));      // the largest it can ask for
    }
    hipLaunchKernelGGL(mi::vote_kernel, dim3((unsigned)B), dim3(mi::VOTE_THREADS), dyn, (hipStream_t)stream, vp);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_queue_push(void *queue, int64_t K, int64_t row_bytes, int64_t ptr,
                  const void *value, int64_t b, int64_t *new_ptr, mi_stream_t stream)
{
    if (!queue || !value) return fail(MI_ERR_ARG, "mi_queue_push: null pointer");
    if (K < 1 || row_bytes < 1 || b < 0 || ptr < 0 || ptr >= K) return fail(MI_ERR_ARG, "mi_queue_push: bad shape K=%lld row_bytes=%lld b=%lld ptr=%lld", (long long)K, (long long)row_bytes, (long long)b, (long long)ptr);
    const int64_t w = b < K - ptr ? b : K - ptr;       // truncate: overflow rows are dropped (memory_bank.py:68-70)
    if (w > 0) {
        char *dst = (char *)queue + ptr * row_bytes;
        const long long bytes = w * row_bytes;
        const int vec = ((((uintptr_t)dst | (uintptr_t)value) & 15) == 0 && (bytes & 15) == 0) ? 1 : 0;
        const long long items = vec ? bytes >> 4 : bytes;
        int grid = (int)ceil_div(items, 256);
        if (grid > 2048) grid = 2048;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(push_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, (const char *)value, bytes, vec);
        HIP_CHECK(hipGetLastError());
    }
    if (new_ptr) *new_ptr = (ptr + w) % K;
    return MI_OK;
}

int mi_queue_push_wrap(void *queue, int64_t K, int64_t row_bytes, int64_t cursor,
                       const void *value, int64_t b, int64_t *new_cursor, mi_stream_t stream)
{
    if (!queue || !value) return fail(MI_ERR_ARG, "mi_queue_push_wrap: null pointer");
    if (K < 1 || row_bytes < 1 || b < 0 || cursor < 0 || cursor >= K) return fail(MI_ERR_ARG, "mi_queue_push_wrap: bad shape K=%lld row_bytes=%lld b=%lld cursor=%lld", (long long)K, (long long)row_bytes, (long long)b, (long long)cursor);
    if (b - (K - cursor) > K) return fail(MI_ERR_ARG, "mi_queue_push_wrap: shape mismatch: %lld rows from cursor %lld do not fit a queue of %lld (the reference's second slice assignment, data[:left] = item[right:] with left > K, fails the same way)", (long long)b, (long long)cursor, (long long)K);
    if (b > 0) {
        const int vec = ((((uintptr_t)queue | (uintptr_t)value) & 15) == 0 && (row_bytes & 15) == 0) ? 1 : 0;
        const long long items = b * (vec ? row_bytes >> 4 : row_bytes);
        int grid = (int)ceil_div(items, 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(push_wrap_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (char *)queue, (long long)K,
                           (long long)row_bytes, (long long)cursor, (const char *)value, (long long)b, vec);
        HIP_CHECK(hipGetLastError());
    }
    if (new_cursor) *new_cursor = (cursor + b) % K;
    return MI_OK;
}

int mi_bank_scatter(void *bank, int64_t K, int64_t row_bytes, const int64_t *index,
                    const void *value, int64_t b, mi_stream_t stream)
{
    if (!bank || !index || !value) return fail(MI_ERR_ARG, "mi_bank_scatter: null pointer");
    if (K < 1 || row_bytes < 1 || b < 0) return fail(MI_ERR_ARG, "mi_bank_scatter: bad shape");
    if (b == 0) return MI_OK;
    const int vec = ((((uintptr_t)bank | (uintptr_t)value) & 15) == 0 && (row_bytes & 15) == 0) ? 1 : 0;
    const long long items = b * (vec ? row_bytes >> 4 : row_bytes);
    int grid = (int)ceil_div(items, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (char *)bank, (long long)K,
                       (long long)row_bytes, (const long long *)index, (const char *)value, (long long)b, vec);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ------------------------------------------------------------------ MoCo queue logits
namespace {
constexpr int MOCO_BWD_WGS = 512;       // workgroups of the backward (two waves per SIMD)
struct MocoPlan { int b_pad, ntiles, lse_ld; size_t off_part, off_rowloss, fwd_total; int nslices, rows_per_slice; size_t bwd_total, total; };
MocoPlan moco_plan(int64_t b, int64_t K, int64_t D)
{
    MocoPlan m{};
    m.b_pad = (int)round_up(b, mi::TILE);
    m.ntiles = (int)ceil_div(K, mi::TILE);
    m.lse_ld = m.b_pad;
    m.off_part = 0;
    m.off_rowloss = (size_t)round_up((int64_t)m.ntiles * m.lse_ld * 8, 256);      // the arrival counter (own line) + per-block loss sums
    m.fwd_total = m.off_rowloss + 256 + (size_t)round_up((int64_t)ceil_div(b, 16) * 4, 256);
    // backward: ~MOCO_BWD_WGS workgroups = (64-row query tiles) x (queue slices); one [b, D] partial per slice
    const int64_t qtiles = ceil_div(b, mi::MB_ROWS);
    const int64_t ktiles = ceil_div(K, mi::MB_ROWS);
    int64_t ns = ceil_div(MOCO_BWD_WGS, qtiles);
    if (ns > ktiles) ns = ktiles;
    if (ns < 1) ns = 1;
    m.rows_per_slice = (int)(ceil_div(ktiles, ns) * mi::MB_ROWS);
    m.nslices = (int)ceil_div(K, m.rows_per_slice);
    m.bwd_total = (size_t)round_up((int64_t)m.nslices * b * D * 4, 256);
    m.total = m.fwd_total > m.bwd_total ? m.fwd_total : m.bwd_total;
    return m;
}
}  // namespace

size_t mi_moco_workspace_bytes(int64_t b, int64_t K, int64_t D)
{
    if (b <= 0 || K <= 0 || D <= 0) return 0;
    return moco_plan(b, K, D).total;
}

int mi_moco_fwd(const float *q, const float *key, const float *queue, int64_t b, int64_t K,
                int64_t D, float temperature, float *lse, float *lpos, float *loss,
                void *workspace, size_t workspace_bytes, mi_stream_t stream)
{
    if (!q || !key || !queue || !lse || !lpos || !loss || !workspace) return fail(MI_ERR_ARG, "mi_moco_fwd: null pointer");
    if (b < 1 || K < 1 || D < 1 || b > (1 << 24) || K > INT32_MAX - 2 * mi::TILE) return fail(MI_ERR_ARG, "mi_moco_fwd: bad shape");
    if (D % 32) return fail(MI_ERR_UNSUPPORTED, "mi_moco_fwd: D=%lld must be a multiple of 32", (long long)D);
    if (((uintptr_t)q | (uintptr_t)queue | (uintptr_t)workspace) & 15) return fail(MI_ERR_ARG, "mi_moco_fwd: q, queue, workspace must be 16-byte aligned");
    if (!(temperature > 0.f)) return fail(MI_ERR_ARG, "mi_moco_fwd: temperature must be positive");
    const MocoPlan mp = moco_plan(b, K, D);
    if (workspace_bytes < mp.fwd_total) return fail(MI_ERR_WORKSPACE, "mi_moco_fwd: workspace %zu < required %zu", workspace_bytes, mp.fwd_total);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    mi::SimParams sp{};
    sp.q = (const char *)q;
    sp.x = (const char *)queue;
    sp.ldq_bytes = sp.ldx_bytes = D * 4;
    sp.B = (int)b;
    sp.N = (int)K;
    sp.ksteps = (int)(D * 4 / mi::KSTEP_BYTES);
    sp.col0 = 0;
    sp.ncols = (int)K;
    sp.qtiles = mp.b_pad / mi::TILE;
    sp.ntiles = mp.ntiles;
    sp.inv_t = 1.0f / temperature;
    sp.lse_part = (float2 *)(ws + mp.off_part);
    sp.lse_ld = mp.lse_ld;
    const int nblocks = (int)ceil_div(b, 16);
    unsigned *arrived = (unsigned *)(ws + mp.off_rowloss);                 // own 256-byte line
    float *block_sum = (float *)(ws + mp.off_rowloss + 256);
    sp.zero_word = arrived;
    ProfScope ps(MI_PROF_MOCO_FWD, 2.0 * (double)b * (double)(K + 1) * (double)D, s);
    HIP_CHECK((launch_sim<mi::DT_F32, mi::MODE_LSE>(sp, s)));
    hipLaunchKernelGGL(moco_combine_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, q, key, (int)b, (int)D, sp.lse_part, mp.ntiles, mp.lse_ld, sp.inv_t, lse, lpos, block_sum, arrived, loss);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_moco_bwd(const float *q, const float *key, const float *queue, int64_t b, int64_t K,
                int64_t D, float temperature, const float *lse, const float *lpos,
                const float *grad_loss, float *dq, float *dkey,
                const float *patch_rows, int64_t patch_lo, int64_t patch_n,
                void *workspace, size_t workspace_bytes, mi_stream_t stream)
{
    if (!q || !key || !lse || !lpos || !grad_loss) return fail(MI_ERR_ARG, "mi_moco_bwd: null pointer");
    if (!dq && !dkey) return fail(MI_ERR_ARG, "mi_moco_bwd: dq and dkey are both null");
    if (b < 1 || K < 1 || D < 1 || b > (1 << 24) || K > INT32_MAX - 2 * mi::TILE) return fail(MI_ERR_ARG, "mi_moco_bwd: bad shape");
    if (!(temperature > 0.f)) return fail(MI_ERR_ARG, "mi_moco_bwd: temperature must be positive");
    if (D % 4) return fail(MI_ERR_UNSUPPORTED, "mi_moco_bwd: D=%lld must be a multiple of 4 (of 64 for dq)", (long long)D);
    if (((uintptr_t)q | (uintptr_t)key | (uintptr_t)dq | (uintptr_t)dkey) & 15) return fail(MI_ERR_ARG, "mi_moco_bwd: q, key, dq, dkey must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float inv_t = 1.0f / temperature;
    int nslices = 0;
    ProfScope ps(MI_PROF_MOCO_BWD, (dq ? 4.0 * (double)b * (double)K * (double)D : 0.0), s);
    if (dq) {
        if (!queue || !workspace) return fail(MI_ERR_ARG, "mi_moco_bwd: null pointer");
        if (D % 64 || D > 256) return fail(MI_ERR_UNSUPPORTED, "mi_moco_bwd: D=%lld must be a multiple of 64 and at most 256", (long long)D);
        if (patch_n < 0 || patch_lo < 0 || patch_lo + patch_n > K || (patch_n > 0 && !patch_rows)) return fail(MI_ERR_ARG, "mi_moco_bwd: bad patch range [%lld, +%lld) for K=%lld", (long long)patch_lo, (long long)patch_n, (long long)K);
        if (((uintptr_t)q | (uintptr_t)queue | (uintptr_t)workspace | (uintptr_t)patch_rows | (uintptr_t)dq) & 15) return fail(MI_ERR_ARG, "mi_moco_bwd: q, queue, patch_rows, dq, workspace must be 16-byte aligned");
        const MocoPlan mp = moco_plan(b, K, D);
        if (workspace_bytes < mp.bwd_total) return fail(MI_ERR_WORKSPACE, "mi_moco_bwd: workspace %zu < required %zu", workspace_bytes, mp.bwd_total);
        mi::AttendParams bp{};
        bp.q = q;
        bp.bank = queue;
        bp.lse = lse;
        bp.patch = patch_n > 0 ? patch_rows : nullptr;
        bp.patch_lo = (int)patch_lo;
        bp.patch_n = (int)patch_n;
        bp.b = (int)b;
        bp.K = (int)K;
        bp.inv_t = inv_t;
        bp.part = (float *)workspace;
        bp.rows_per_slice = mp.rows_per_slice;
        nslices = mp.nslices;
        const dim3 grid((unsigned)ceil_div(b, mi::MB_ROWS), (unsigned)mp.nslices);
        const size_t lds = mi::attend_lds_bytes((int)D, (int)D, false);
#define MI_LAUNCH_BWD(D64)                                                                                       \
        do {                                                                                                    \
            static std::atomic<uint64_t> done{0};      /* lds is a constant of the instantiation */                \
            HIP_CHECK(ensure_dynamic_lds(mi::attend_kernel<D64, D64, false>, lds, done));                       \
            hipLaunchKernelGGL((mi::attend_kernel<D64, D64, false>), grid, dim3(mi::MB_THREADS), lds, s, bp);   \
        } while (0)
        switch ((int)(D / 64)) {
        case 1: MI_LAUNCH_BWD(1); break;
        case 2: MI_LAUNCH_BWD(2); break;
        case 3: MI_LAUNCH_BWD(3); break;
        default: MI_LAUNCH_BWD(4); break;
        }
#undef MI_LAUNCH_BWD
        HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(mi::moco_bwd_reduce_kernel, dim3((unsigned)ceil_div(b * D / 4, 16)), dim3(256), 0, s,
                       (const float *)workspace, nslices, (int)b, (int)D, q, key, lse, lpos, grad_loss, inv_t, dq, dkey);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

// ------------------------------------------------------------------ CoMatch soft-kNN smoothing
extern "C++" {
namespace {
constexpr int SOFT_WGS = 256;
struct SoftPlan { int V, nslices, rows_per_slice; size_t total; };
SoftPlan soft_plan(int64_t b, int64_t K, int64_t D, int64_t C)
{
    SoftPlan sp{};
    sp.V = (int)round_up(C + 1, 64);                       // + the row-sum column
    // attend_kernel<3, 3, true> and <4, 3, true> do not fit the register file (124 / 240 bytes of scratch per lane at
    // 256 VGPR + 256 AGPR): 128 <= C <= 190 with D >= 192 runs the 256-column instantiation, which does (zero columns)
    if (sp.V == 192 && D >= 192) sp.V = 256;
    const int64_t qtiles = ceil_div(b, mi::MB_ROWS), ktiles = ceil_div(K, mi::MB_ROWS);
    int64_t ns = ceil_div(SOFT_WGS, qtiles);
    if (ns > ktiles) ns = ktiles;
    if (ns < 1) ns = 1;
    sp.rows_per_slice = (int)(ceil_div(ktiles, ns) * mi::MB_ROWS);
    sp.nslices = (int)ceil_div(K, sp.rows_per_slice);
    sp.total = (size_t)round_up((int64_t)sp.nslices * b * sp.V * 4, 256);
    return sp;
}

template <int D64, int V64>
hipError_t launch_soft(const mi::AttendParams &ap, dim3 grid, size_t lds, hipStream_t s)
{
    static std::atomic<uint64_t> done{0};              // lds is a constant of the instantiation
    hipError_t e = ensure_dynamic_lds(mi::attend_kernel<D64, V64, true>, lds, done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((mi::attend_kernel<D64, V64, true>), grid, dim3(mi::MB_THREADS), lds, s, ap);
    return hipGetLastError();
}
template <int D64>
hipError_t launch_soft_v(int v64, const mi::AttendParams &ap, dim3 grid, size_t lds, hipStream_t s)
{
    switch (v64) {
    case 1: return launch_soft<D64, 1>(ap, grid, lds, s);
    case 2: return launch_soft<D64, 2>(ap, grid, lds, s);
    case 3:
        if constexpr (D64 < 3) return launch_soft<D64, 3>(ap, grid, lds, s);
        else return hipErrorInvalidValue;              // never planned (soft_plan pads V to 256): those two instantiations spill
    default: return launch_soft<D64, 4>(ap, grid, lds, s);
    }
}
}  // namespace
}  // extern "C++"

size_t mi_soft_knn_workspace_bytes(int64_t b, int64_t K, int64_t D, int64_t C)
{
    if (b <= 0 || K <= 0 || D <= 0 || C <= 0 || C > 255) return 0;
    return soft_plan(b, K, D, C).total;
}

int mi_soft_knn(const float *feature, const float *bank_feature, const float *bank_prob, const float *probs_in,
                int64_t b, int64_t K, int64_t D, int64_t C, float temperature, float alpha, float *probs_out,
                void *workspace, size_t workspace_bytes, mi_stream_t stream)
{
    if (!feature || !bank_feature || !bank_prob || !probs_in || !probs_out || !workspace) return fail(MI_ERR_ARG, "mi_soft_knn: null pointer");
    if (b < 1 || K < 1 || b > (1 << 24) || K > INT32_MAX - 2 * mi::TILE) return fail(MI_ERR_ARG, "mi_soft_knn: bad shape b=%lld K=%lld", (long long)b, (long long)K);
    if (D % 64 || D < 64 || D > 256) return fail(MI_ERR_UNSUPPORTED, "mi_soft_knn: D=%lld must be a multiple of 64 in [64, 256]", (long long)D);
    if (C < 1 || C > 255) return fail(MI_ERR_UNSUPPORTED, "mi_soft_knn: C=%lld outside [1, 255]", (long long)C);
    if (!(temperature > 0.f)) return fail(MI_ERR_ARG, "mi_soft_knn: temperature must be positive");
    if (((uintptr_t)feature | (uintptr_t)bank_feature | (uintptr_t)workspace) & 15) return fail(MI_ERR_ARG, "mi_soft_knn: feature, bank_feature, workspace must be 16-byte aligned");
    const SoftPlan sp = soft_plan(b, K, D, C);
    if (workspace_bytes < sp.total) return fail(MI_ERR_WORKSPACE, "mi_soft_knn: workspace %zu < required %zu", workspace_bytes, sp.total);
    const size_t lds = mi::attend_lds_bytes((int)D, sp.V, true);
    if (lds > 160 * 1024) return fail(MI_ERR_UNSUPPORTED, "mi_soft_knn: D=%lld with C=%lld needs %zu bytes of LDS (> 160 KiB)", (long long)D, (long long)C, lds);
    hipStream_t s = (hipStream_t)stream;
    mi::AttendParams ap{};
    ap.q = feature;
    ap.bank = bank_feature;
    ap.vals = bank_prob;
    ap.C = (int)C;
    ap.b = (int)b;
    ap.K = (int)K;
    ap.inv_t = 1.0f / temperature;
    ap.part = (float *)workspace;
    ap.rows_per_slice = sp.rows_per_slice;
    const dim3 grid((unsigned)ceil_div(b, mi::MB_ROWS), (unsigned)sp.nslices);
    const int v64 = sp.V / 64;
    hipError_t e;
    switch ((int)(D / 64)) {
    case 1: e = launch_soft_v<1>(v64, ap, grid, lds, s); break;
    case 2: e = launch_soft_v<2>(v64, ap, grid, lds, s); break;
    case 3: e = launch_soft_v<3>(v64, ap, grid, lds, s); break;
    default: e = launch_soft_v<4>(v64, ap, grid, lds, s); break;
    }
    HIP_CHECK(e);
    hipLaunchKernelGGL(mi::soft_knn_reduce_kernel, dim3((unsigned)b), dim3(256), 0, s,
                       (const float *)workspace, sp.nslices, (int)b, sp.V, (int)C, probs_in, alpha, probs_out);
    HIP_CHECK(hipGetLastError());
    return MI_OK;
}

int mi_queue_push_stash(void *queue, int64_t K, int64_t row_bytes, int64_t ptr,
                        const void *value, int64_t b, void *stash, int64_t *new_ptr, mi_stream_t stream)
{
    if (!queue || !value || !stash) return fail(MI_ERR_ARG, "mi_queue_push_stash: null pointer");
    if (K < 1 || row_bytes < 1 || b < 0 || ptr < 0 || ptr >= K) return fail(MI_ERR_ARG, "mi_queue_push_stash: bad shape K=%lld row_bytes=%lld b=%lld ptr=%lld", (long long)K, (long long)row_bytes, (long long)b, (long long)ptr);
    const int64_t w = b < K - ptr ? b : K - ptr;       // truncating, as mi_queue_push
    if (w > 0) {
        char *dst = (char *)queue + ptr * row_bytes;
        const long long bytes = w * row_bytes;
        const bool vec = ((((uintptr_t)dst | (uintptr_t)value | (uintptr_t)stash) & 15) == 0 && (bytes & 15) == 0);
        const long long items = vec ? bytes >> 4 : bytes;
        int grid = (int)ceil_div(items, 256);
        if (grid > 2048) grid = 2048;
        if (vec) hipLaunchKernelGGL(mi::push_stash_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (uint4 *)dst, (const uint4 *)value, (uint4 *)stash, items);
        else hipLaunchKernelGGL(mi::push_stash_bytes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst, (const char *)value, (char *)stash, items);
        HIP_CHECK(hipGetLastError());
    }
    if (new_ptr) *new_ptr = (ptr + w) % K;
    return MI_OK;
}

}  // extern "C"
