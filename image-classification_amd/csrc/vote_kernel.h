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
constexpr int VOTE_WAVE_MAX_C = 2048;          // 4 waves x (3 KiB + 2 B per class) stay well below 48 KiB of dynamic LDS

inline size_t vote_wave_lds_bytes(int classes)
{
    const size_t per_wave = 256 * 4 /*w*/ + 256 * 4 /*label keys*/ + 256 * 4 /*bin totals*/ + (size_t)((classes * 2 + 15) / 16 * 16);
    return 4 * per_wave;
}

__global__ __launch_bounds__(256) void vote_wave_kernel(const VoteParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wv;
    if (row >= p.B) return;
    const int k = p.k, C = p.classes;
    const size_t per_wave = 256 * 4 * 3 + (size_t)((C * 2 + 15) / 16 * 16);
    unsigned char *base = dyn + wv * per_wave;
    float *s_w = (float *)base;                                        // [256] weight of neighbour j
    unsigned *s_lk = (unsigned *)(base + 1024);                        // [256] sorted label << 8 | j
    float *s_tot = (float *)(base + 2048);                             // [256] bin totals (by bin number before the sort)
    unsigned short *slot = (unsigned short *)(base + 3072);            // [C] class -> bin + 1 (0: no vote)

    // (knn.py:10-11) gather labels, w = exp(sim / T): division first, then exp, as the reference.
    // All index loads first, then all label loads, then the values: one round trip each, not four.
    long long bi[4];
    float sv[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const int j = s2 * 64 + lane;
        bi[s2] = j < k ? p.idx[row * k + j] : -1ll;
        sv[s2] = j < k ? p.val[row * k + j] : 0.f;
    }
    long long bl[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) bl[s2] = (bi[s2] >= 0 && bi[s2] < p.n_labels) ? p.labels[bi[s2]] : -1ll;
    for (int c = lane; c < C; c += 64) slot[c] = 0;
    unsigned lk[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const int j = s2 * 64 + lane;
        const bool ok = bl[s2] >= 0 && bl[s2] < C;
        lk[s2] = ok ? (((unsigned)bl[s2] << 8) | (unsigned)j) : 0xFFFFFFFFu;
        s_w[j] = ok ? expf(sv[s2] / p.knn_t) : 0.f;
    }
    wave_sort256<unsigned, false>(lk, lane);                           // runs of equal label, neighbour rank ascending
    *(uint4 *)(s_lk + 4 * lane) = make_uint4(lk[0], lk[1], lk[2], lk[3]);
    wave_step();

    // (knn.py:14-18) one histogram bin per run, summed in ascending neighbour rank
    unsigned long long bk[4];
    int nbins = 0;
    const unsigned prev_lane_last = lane ? s_lk[4 * lane - 1] : 0xFFFFFFFFu;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
        const int t = 4 * lane + s2;
        const unsigned key = lk[s2];
        const unsigned l = key >> 8;
        const unsigned before = s2 ? lk[s2 ? s2 - 1 : 0] : prev_lane_last;
        const bool head = key != 0xFFFFFFFFu && (t == 0 || (before >> 8) != l);
        float sum = 0.f;
        if (head)
            for (int u = t; u < 256 && (s_lk[u] >> 8) == l; ++u) sum += s_w[s_lk[u] & 255u];
        const unsigned long long bal = __ballot(head);
        bk[s2] = 0ull;
        if (head) {
            const int b = nbins + __popcll(bal & ((1ull << lane) - 1ull));
            s_tot[b] = sum;
            slot[l] = (unsigned short)(b + 1);
            bk[s2] = ((unsigned long long)orderable_f32(sum) << 32) | (unsigned long long)(0xFFFFFFFFu - l);
        }
        nbins += __popcll(bal);
    }
    // (knn.py:20) ranking: voted classes by (total desc, class asc): empty slots (0) sort behind every bin ...
    wave_sort256<unsigned long long, true>(bk, lane);
    long long *rank = p.out_rank + row * C;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2)
        if (4 * lane + s2 < nbins) rank[4 * lane + s2] = (long long)(0xFFFFFFFFu - (unsigned)bk[s2]);
    wave_step();
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
