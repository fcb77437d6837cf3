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

}  // namespace mi
