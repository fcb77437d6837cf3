// select_kernel.h -- per-row top-k maintenance, merge and finalisation (models/knn.py:8).
//
// Total order used everywhere: (similarity descending, bank row ascending).  It is encoded in one
// 64-bit key so that a plain unsigned sort implements it:
//     key = orderable(value + 0.0f) << 32 | (0xFFFFFFFF - row)         (key 0 = empty slot)
// orderable() maps float bits monotonically to unsigned; "+ 0.0f" folds -0.0 into +0.0 so equal
// similarities compare equal; NaN similarities map to key 0 and are never selected.
//
// One 256-thread workgroup per query row keeps the row's running top-k (sorted, in HBM) and folds
// new candidates in through an LDS bitonic sort of at most SORT_P keys per pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

constexpr int SORT_P = 2048;          // LDS sort capacity (keys); k <= SORT_P / 2
constexpr int SEL_THREADS = 256;
constexpr int SRC_LIST = 0, SRC_DENSE = 1, SRC_GATHER = 2;

__device__ __forceinline__ unsigned orderable_f32(float v)
{
    const unsigned u = __float_as_uint(v + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(unsigned o)
{
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ unsigned long long make_key(float v, unsigned row)
{
    if (v != v) return 0ull;
    return ((unsigned long long)orderable_f32(v) << 32) | (unsigned long long)(0xFFFFFFFFu - row);
}
__device__ __forceinline__ float key_value(unsigned long long k) { return from_orderable((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned key_row(unsigned long long k) { return 0xFFFFFFFFu - (unsigned)k; }

// Descending bitonic sort of n (power of two, <= SORT_P) keys in LDS by one 256-thread workgroup.
__device__ __forceinline__ void bitonic_sort_desc(unsigned long long *keys, int n, int tid)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += SEL_THREADS) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
}

struct SelectParams {
    // candidate sources
    const uint2 *cand;          // SRC_LIST : [B_pad, cap] (value bits, row);  SRC_DENSE: float [B_pad][2*cap]
    unsigned *cnt;              // SRC_LIST : [B_pad] candidate counts (reset to 0 here)
    int cap;
    int dense_n, dense_col0;    // SRC_DENSE: n values per row, bank row of value 0
    const float *g_val;         // SRC_GATHER: [G, B, kc]
    const long long *g_idx;     // SRC_GATHER: [G, B, kc] (global rows, < 0 = padding)
    int G, kc;
    long long B;
    // running state
    unsigned long long *topk;   // [B_pad, k] keys sorted descending
    int *topk_n;                // [B_pad] valid keys
    float *tau;                 // [B_pad] threshold = k-th value once k keys are held, else -inf
    int k;
    // final output (written when out_val != nullptr)
    float *out_val;             // [B, k]
    long long *out_idx;         // [B, k]
    long long idx_base;
};

template <int SRC>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const SelectParams p)
{
    __shared__ unsigned long long keys[SORT_P];
    __shared__ int s_valid;
    const int row = blockIdx.x;
    const int tid = threadIdx.x;

    int n_new;
    if constexpr (SRC == SRC_LIST) {
        const unsigned c = p.cnt[row];
        n_new = c < (unsigned)p.cap ? (int)c : p.cap;
    } else if constexpr (SRC == SRC_DENSE) {
        n_new = p.dense_n;
    } else {
        n_new = p.G * p.kc;
    }
    const bool finalize = p.out_val != nullptr;
    if (n_new == 0 && !finalize) return;

    int n_cur = 0;
    if constexpr (SRC != SRC_GATHER) {
        n_cur = p.topk_n[row];
        for (int i = tid; i < n_cur; i += SEL_THREADS) keys[i] = p.topk[(long long)row * p.k + i];
    }
    int pos = 0;
    while (pos < n_new) {
        const int room = SORT_P - n_cur;
        const int m = (n_new - pos) < room ? (n_new - pos) : room;
        for (int i = tid; i < m; i += SEL_THREADS) {
            const int e = pos + i;
            unsigned long long key;
            if constexpr (SRC == SRC_LIST) {
                const uint2 c = p.cand[(long long)row * p.cap + e];
                key = make_key(__uint_as_float(c.x), c.y);
            } else if constexpr (SRC == SRC_DENSE) {
                const float *d = (const float *)p.cand + (long long)row * 2 * p.cap;
                key = make_key(d[e], (unsigned)(p.dense_col0 + e));
            } else {
                const int g = e / p.kc, j = e - g * p.kc;
                const long long off = ((long long)g * p.B + row) * p.kc + j;
                const long long gi = p.g_idx[off];
                key = gi < 0 ? 0ull : make_key(p.g_val[off], (unsigned)gi);
            }
            keys[n_cur + i] = key;
        }
        const int tot = n_cur + m;
        int n2 = 64;
        while (n2 < tot) n2 <<= 1;
        for (int i = tot + tid; i < n2; i += SEL_THREADS) keys[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc(keys, n2, tid);
        // keep the best k; empty (0) keys sort last, count how many of the kept ones are real
        const int keep = tot < p.k ? tot : p.k;
        if (tid == 0) s_valid = 0;
        __syncthreads();
        int mine = 0;
        for (int i = tid; i < keep; i += SEL_THREADS) mine += keys[i] != 0ull;
        if (mine) atomicAdd(&s_valid, mine);
        __syncthreads();
        n_cur = s_valid;
        pos += m;
        __syncthreads();
    }

    __syncthreads();
    if constexpr (SRC != SRC_GATHER) {
        for (int i = tid; i < n_cur; i += SEL_THREADS) p.topk[(long long)row * p.k + i] = keys[i];
        if (tid == 0) {
            p.topk_n[row] = n_cur;
            p.tau[row] = (n_cur >= p.k) ? key_value(keys[p.k - 1]) : -INFINITY;
            if constexpr (SRC == SRC_LIST) p.cnt[row] = 0u;
        }
    }
    if (finalize) {
        for (int i = tid; i < p.k; i += SEL_THREADS) {
            const bool ok = i < n_cur;
            p.out_val[(long long)row * p.k + i] = ok ? key_value(keys[i]) : -INFINITY;
            p.out_idx[(long long)row * p.k + i] =
                ok ? (SRC == SRC_GATHER ? (long long)key_row(keys[i]) : p.idx_base + (long long)key_row(keys[i])) : -1ll;
        }
    }
}

}  // namespace mi
