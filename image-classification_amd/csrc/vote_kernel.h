// vote_kernel.h -- label gather, exp(s/T) weights, class vote and full class ranking
// (models/knn.py:10-20) for one query row per 256-thread workgroup.
//
// The reference builds a dense [B*k, C] one-hot matrix (800 MB at 10k x 200 x 100) and reduces
// it; here the k neighbour labels and weights live in LDS.  At most k classes receive a vote, so:
//   1. every neighbour j sums, in ascending neighbour rank, the weights of all neighbours that
//      share its label (k x k LDS broadcast reads) -> deterministic class totals;
//   2. the first neighbour of each class is that class's representative; representatives are
//      sorted by (total descending, class ascending) with an LDS bitonic sort (<= 1024 keys);
//   3. the ranking row is [voted classes in that order] ++ [all other classes ascending], the
//      second part placed by a block-wide prefix count over an LDS class->representative map.
// Ties (in particular every zero-score class) go to the lower class id: the reference's argsort
// leaves that order unspecified.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "select_kernel.h"

namespace mi {

constexpr int VOTE_THREADS = 256;
constexpr int VOTE_MAX_K = 1024;

struct VoteParams {
    long long B;
    const float *val;            // [B, k]
    const long long *idx;        // [B, k]
    const long long *labels;     // [n_labels]
    long long n_labels;
    int k, classes;
    float knn_t;
    float *out_scores;           // [B, classes] or nullptr
    long long *out_rank;         // [B, classes]
};

__global__ __launch_bounds__(VOTE_THREADS) void vote_kernel(const VoteParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];   // class -> representative+1 (u16)
    unsigned short *slot = (unsigned short *)dyn;
    __shared__ float s_w[VOTE_MAX_K];
    __shared__ float s_tot[VOTE_MAX_K];
    __shared__ int s_lbl[VOTE_MAX_K];
    __shared__ unsigned long long s_keys[VOTE_MAX_K];
    __shared__ int s_wave[VOTE_THREADS / 64];
    __shared__ int s_npos;

    const long long row = blockIdx.x;
    const int tid = threadIdx.x, k = p.k, C = p.classes;

    // (knn.py:10-11) gather labels, w = exp(sim / T): division first, then exp, as the reference
    for (int j = tid; j < k; j += VOTE_THREADS) {
        const long long i = p.idx[row * k + j];
        int l = -1;
        float w = 0.f;
        if (i >= 0 && i < p.n_labels) {
            const long long ll = p.labels[i];
            if (ll >= 0 && ll < C) { l = (int)ll; w = expf(p.val[row * k + j] / p.knn_t); }
        }
        s_lbl[j] = l;
        s_w[j] = w;
    }
    for (int c = tid; c < C; c += VOTE_THREADS) slot[c] = 0;
    if (tid == 0) s_npos = 0;
    int k2 = 64;
    while (k2 < k) k2 <<= 1;
    __syncthreads();

    // (knn.py:14-18) class totals, summed in ascending neighbour rank
    for (int j = tid; j < k2; j += VOTE_THREADS) {
        unsigned long long key = 0ull;
        if (j < k) {
            const int l = s_lbl[j];
            if (l >= 0) {
                float s = 0.f;
                bool first = true;
                for (int jj = 0; jj < k; ++jj) {
                    if (s_lbl[jj] == l) {
                        s += s_w[jj];
                        if (jj < j) first = false;
                    }
                }
                if (first) {
                    s_tot[j] = s;
                    slot[l] = (unsigned short)(j + 1);
                    key = ((unsigned long long)orderable_f32(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)l);
                    atomicAdd(&s_npos, 1);
                }
            }
        }
        s_keys[j] = key;
    }
    __syncthreads();

    // (knn.py:20) ranking: voted classes by (score desc, class asc) ...
    bitonic_sort_desc(s_keys, k2, tid);
    const int npos = s_npos;
    long long *rank = p.out_rank + row * C;
    for (int i = tid; i < npos; i += VOTE_THREADS) rank[i] = (long long)(0xFFFFFFFFu - (unsigned)s_keys[i]);

    // ... followed by every class without a vote, ascending
    int running = npos;
    const int lane = tid & 63, wv = tid >> 6;
    for (int c0 = 0; c0 < C; c0 += VOTE_THREADS) {
        const int c = c0 + tid;
        const bool unvoted = c < C && slot[c] == 0;
        const unsigned long long mask = __ballot(unvoted);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wv] = __popcll(mask);
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int i = 0; i < VOTE_THREADS / 64; ++i) {
            const int n = s_wave[i];
            if (i < wv) off += n;
            total += n;
        }
        if (unvoted) rank[running + off + before] = c;
        if (p.out_scores && c < C) {
            const int sl = slot[c];
            p.out_scores[row * C + c] = sl ? s_tot[sl - 1] : 0.f;
        }
        running += total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-per-row vote (k <= 256, classes <= VOTE_WAVE_MAX_C): the LDS class histogram of models/knn.py:14-18 without
// its k x k label matching and without workgroup barriers.  One wave owns a query row; everything between the
// loads and the stores happens in ~6 KiB of wave-private LDS:
//   1. neighbour j -> key (label << 8 | j), sorted ascending (wave-private bitonic sort): neighbours of one class
//      become one run, in ascending neighbour rank inside the run;
//   2. the first element of each run adds up its run's weights in that order (deterministic: the sum of a class
//      does not depend on scheduling) -> the class's histogram bin; bins are compacted by ballot;
//   3. bins sorted by (total descending, class ascending) -> the voted prefix of the ranking; the classes without a
//      vote follow in ascending order, placed by ballot prefix over an LDS class -> bin map (which also serves the
//      optional dense score row).
// HBM traffic is the algorithmic minimum (k * 12 B in, classes * 8 (+4) B out per row); 16+ rows are in flight per CU.
constexpr int VOTE_WAVE_MAX_C = 2048;          // 4 waves x (5 KiB + 2 B per class) stay below 48 KiB of dynamic LDS

template <typename T, bool DESC>
__device__ __forceinline__ void wave_bitonic_sort(T *keys, int n, int lane)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool first_big = ((lo & size) == 0) == DESC;     // the larger key belongs at `lo`
                const T a = keys[lo], b = keys[hi];
                if ((a < b) == first_big) { keys[lo] = b; keys[hi] = a; }
            }
            wave_step();
        }
    }
}

inline size_t vote_wave_lds_bytes(int classes)
{
    const size_t per_wave = 256 * 4 /*w*/ + 256 * 4 /*label keys*/ + 256 * 4 /*bin totals*/ + 256 * 8 /*bin keys*/
                          + (size_t)((classes * 2 + 15) / 16 * 16);
    return 4 * per_wave;
}

__global__ __launch_bounds__(256) void vote_wave_kernel(const VoteParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wv;
    if (row >= p.B) return;
    const int k = p.k, C = p.classes;
    const size_t per_wave = 256 * 4 * 3 + 256 * 8 + (size_t)((C * 2 + 15) / 16 * 16);
    unsigned char *base = dyn + wv * per_wave;
    unsigned long long *s_bin = (unsigned long long *)base;            // [256] (total, class) keys of the bins
    float *s_w = (float *)(base + 2048);                               // [256] weight of neighbour j
    unsigned *s_lk = (unsigned *)(base + 3072);                        // [256] label << 8 | j
    float *s_tot = (float *)(base + 4096);                             // [256] bin totals (by bin number before the sort)
    unsigned short *slot = (unsigned short *)(base + 5120);            // [C] class -> bin + 1 (0: no vote)

    int n2 = 64;
    while (n2 < k) n2 <<= 1;
    // (knn.py:10-11) gather labels, w = exp(sim / T): division first, then exp, as the reference
    for (int j = lane; j < n2; j += 64) {
        unsigned key = 0xFFFFFFFFu;
        float w = 0.f;
        if (j < k) {
            const long long i = p.idx[row * k + j];
            if (i >= 0 && i < p.n_labels) {
                const long long ll = p.labels[i];
                if (ll >= 0 && ll < C) { key = ((unsigned)ll << 8) | (unsigned)j; w = expf(p.val[row * k + j] / p.knn_t); }
            }
        }
        s_lk[j] = key;
        s_w[j] = w;
    }
    for (int c = lane; c < C; c += 64) slot[c] = 0;
    wave_step();
    wave_bitonic_sort<unsigned, false>(s_lk, n2, lane);                // runs of equal label, neighbour rank ascending

    // (knn.py:14-18) one histogram bin per run, summed in ascending neighbour rank
    int nbins = 0;
    for (int t0 = 0; t0 < n2; t0 += 64) {
        const int t = t0 + lane;
        const unsigned key = s_lk[t];
        const unsigned l = key >> 8;
        const bool head = key != 0xFFFFFFFFu && (t == 0 || (s_lk[t - 1] >> 8) != l);
        float sum = 0.f;
        if (head)
            for (int u = t; u < n2 && (s_lk[u] >> 8) == l; ++u) sum += s_w[s_lk[u] & 255u];
        const unsigned long long bal = __ballot(head);
        if (head) {
            const int b = nbins + __popcll(bal & ((1ull << lane) - 1ull));
            s_tot[b] = sum;
            slot[l] = (unsigned short)(b + 1);
            s_bin[b] = ((unsigned long long)orderable_f32(sum) << 32) | (unsigned long long)(0xFFFFFFFFu - l);
        }
        nbins += __popcll(bal);
    }
    int m2 = 64;
    while (m2 < nbins) m2 <<= 1;
    wave_step();
    for (int i = nbins + lane; i < m2; i += 64) s_bin[i] = 0ull;
    wave_step();
    // (knn.py:20) ranking: voted classes by (total desc, class asc) ...
    wave_bitonic_sort<unsigned long long, true>(s_bin, m2, lane);
    long long *rank = p.out_rank + row * C;
    for (int i = lane; i < nbins; i += 64) rank[i] = (long long)(0xFFFFFFFFu - (unsigned)s_bin[i]);
    // ... followed by every class without a vote, ascending; the dense score row on the way
    int running = nbins;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        const int sl = c < C ? (int)slot[c] : 1;
        const bool unvoted = sl == 0;
        const unsigned long long bal = __ballot(unvoted);
        if (unvoted) rank[running + __popcll(bal & ((1ull << lane) - 1ull))] = c;
        if (p.out_scores && c < C) p.out_scores[row * C + c] = sl ? s_tot[sl - 1] : 0.f;
        running += __popcll(bal);
    }
}

}  // namespace mi
