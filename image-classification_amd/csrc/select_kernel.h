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

constexpr int BIG_SUBLISTS = 64;
constexpr int MERGE_MAX_NEW = 64;     // rows with <= 64 new candidates take the wave-per-row merge path

struct SelectParams {
    // candidate sources
    const uint2 *cand;          // SRC_LIST : [B_pad, cap] (value bits, row), bank tile t of the stage owns slots
                                //            [256 t, 256 t + cnt2[t][row]) of a row;  SRC_DENSE: float [B_pad][2*cap]
    const unsigned short *cnt2; // SRC_LIST : [ntiles, bpad] candidates per (bank tile of the stage, query row)
    unsigned *tmask;            // SRC_LIST : [B_pad, mask_words] bit t set <=> cnt2[t][row] is valid and > 0 (cleared here)
    int ntiles, bpad, mask_words;
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
    int skip_small;             // SRC_LIST: rows with <= MERGE_MAX_NEW candidates were merged by merge_kernel
    // SRC_LIST work list: merge_kernel appends the rows it leaves to select_kernel (> MERGE_MAX_NEW
    // candidates) to big_rows and counts them in big_cnt[parity]; select_kernel<SRC_LIST> then runs a
    // fixed grid over that list instead of one (mostly idle) workgroup per query row.
    int *big_rows;              // [BIG_SUBLISTS][big_stride]
    unsigned *big_cnt;          // [2][BIG_SUBLISTS] counters, one per 128-byte line, alternating by stage;
                                // merge_kernel zeroes the other set
    int big_stride;             // >= rows one sub-list can receive
    int parity;
    // speculative first threshold (mi_knn.hip, "speculation"): sample_tau_kernel writes the SAFE threshold (k-th largest
    // of the sample) to tau0 and the speculative one (spec_rank-th largest, spec_rank < k) to tau; a row that ends the
    // pass with fewer than k keys although its sample held k real values FAILED and is appended to fail_rows.
    float *tau0;                // [B_pad] or nullptr (no speculation)
    int spec_rank;
    unsigned *fail_cnt;         // one counter, zeroed at the start of the pass
    int *fail_rows;             // [B_pad]
    int force_mod;              // debug (mi_debug_force_fallback): rows with row % force_mod == 0 are reported failed
    // fallback launch: row = COMPACT row c of the failed-row list; rows c >= *fb_cnt - fb_base do nothing, outputs go to
    // query row fb_rows[fb_base + c]
    const int *fb_rows;
    const unsigned *fb_cnt;
    int fb_base;
};

// LDS of one select workgroup.
constexpr int SORT_OUT = 1024;        // >= MI_MAX_K: the kept keys are sorted here
struct SelectSmem {
    unsigned long long keys[SORT_P];  // current top-k + one batch of new candidates (unsorted)
    unsigned long long out[SORT_OUT]; // the kept keys, compacted, then sorted
    unsigned hist[256];
    unsigned long long red[2 * (SEL_THREADS / 64)];
    unsigned long long kth;
    int wsum[SEL_THREADS / 64];
    int bin, rank, bin_count, count;
};

__device__ __forceinline__ unsigned long long wave_and64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v &= __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v |= __shfl_xor(v, d, 64);
    return v;
}

// The rank-th largest (1-based) of the NON-ZERO keys in keys[0, tot) (at least `rank` of them, all distinct):
// MSB-first radix select, 8 bits per pass, starting at the highest bit in which the keys differ at all (so
// the exponent bits that similarities share cost no pass and no histogram hot spot) and stopping as soon as
// the selected bin holds one key.  Every thread returns the same value.  A full sort of 2048 keys took
// ~80 us per row; this plus sorting only the kept keys takes ~1/5 of that.
__device__ __forceinline__ unsigned long long radix_select_kth(SelectSmem &sm, int tot, int rank, int tid)
{
    unsigned long long a = ~0ull, o = 0ull;
    for (int i = tid; i < tot; i += SEL_THREADS) {
        const unsigned long long key = sm.keys[i];
        if (key) { a &= key; o |= key; }
    }
    a = wave_and64(a);
    o = wave_or64(o);
    if ((tid & 63) == 0) { sm.red[2 * (tid >> 6)] = a; sm.red[2 * (tid >> 6) + 1] = o; }
    if (tid == 0) sm.rank = rank;
    __syncthreads();
    a = ~0ull; o = 0ull;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) { a &= sm.red[2 * w]; o |= sm.red[2 * w + 1]; }
    const unsigned long long var = a ^ o;
    if (var == 0ull) return a;                                   // a single distinct key
    const int hb = 63 - __clzll((long long)var);
    unsigned long long pmask = hb == 63 ? 0ull : ~((2ull << hb) - 1ull);     // bits every key shares
    unsigned long long prefix = a & pmask;
    for (int top = hb; top >= 0; top -= 8) {
        const int sh = top >= 7 ? top - 7 : 0;
        const unsigned dmask = top >= 7 ? 255u : ((1u << (top + 1)) - 1u);
        sm.hist[tid] = 0u;                                       // SEL_THREADS == 256 bins
        __syncthreads();
        for (int i = tid; i < tot; i += SEL_THREADS) {
            const unsigned long long key = sm.keys[i];
            if (key && (key & pmask) == prefix) atomicAdd(&sm.hist[(unsigned)(key >> sh) & dmask], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins 4l .. 4l+3; suffix sums from the top bin down
            const unsigned h0 = sm.hist[4 * tid], h1 = sm.hist[4 * tid + 1], h2 = sm.hist[4 * tid + 2], h3 = sm.hist[4 * tid + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned incl = mine;                                // sum over lanes >= this one
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned t = __shfl_down(incl, d, 64);
                if (tid + d < 64) incl += t;
            }
            const unsigned above = incl - mine;                  // keys in bins of higher lanes
            const unsigned r = (unsigned)sm.rank;
            if (above < r && r <= incl) {                        // the rank-th largest falls into this lane's bins
                unsigned acc = above;
                int b = 4 * tid + 3;
                unsigned hb_cnt = h3;
                if (acc + h3 < r) { acc += h3; b = 4 * tid + 2; hb_cnt = h2;
                    if (acc + h2 < r) { acc += h2; b = 4 * tid + 1; hb_cnt = h1;
                        if (acc + h1 < r) { acc += h1; b = 4 * tid; hb_cnt = h0; } } }
                sm.bin = b;
                sm.rank = (int)(r - acc);
                sm.bin_count = (int)hb_cnt;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)(unsigned)sm.bin << sh;
        pmask |= (unsigned long long)dmask << sh;
        if (sh == 0) break;
        if (sm.bin_count == 1) {                                 // one key left under this prefix: it is the answer
            for (int i = tid; i < tot; i += SEL_THREADS) {
                const unsigned long long key = sm.keys[i];
                if (key && (key & pmask) == prefix) sm.kth = key;
            }
            __syncthreads();
            return sm.kth;
        }
    }
    return prefix;
}

template <int SRC>
__device__ __forceinline__ void select_row(const SelectParams &p, const int row, SelectSmem &sm)
{
    const int tid = threadIdx.x;
    unsigned long long *keys = sm.keys;

    int n_new = 0;                                    // SRC_LIST: tiles still to consume (the counts are per tile)
    if constexpr (SRC == SRC_LIST) n_new = p.ntiles;
    else if constexpr (SRC == SRC_DENSE) n_new = p.dense_n;
    else n_new = p.G * p.kc;
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
        int m, step;                                  // m keys appended, `pos` advances by step
        int nz_mine = 0;
        if constexpr (SRC == SRC_LIST) {
            // take the longest run of tiles from `pos` (one tile per thread) whose candidates fit the room:
            // block-wide inclusive scan of the per-tile counts (a tile holds <= 256, room >= 1024: progress)
            const int t = pos + tid;
            int c = 0;
            if (t < n_new && ((p.tmask[(long long)row * p.mask_words + (t >> 5)] >> (t & 31)) & 1u))
                c = (int)p.cnt2[(long long)t * p.bpad + row];
            int incl = c;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if ((tid & 63) >= d) incl += o;
            }
            if ((tid & 63) == 63) sm.wsum[tid >> 6] = incl;
            if (tid == 0) { sm.count = 0; sm.rank = 0; }
            __syncthreads();
            for (int w = 0; w < (tid >> 6); ++w) incl += sm.wsum[w];
            const bool fits = t < n_new && incl <= room;
            if (fits) {
                atomicAdd(&sm.rank, 1);                              // tiles taken (a prefix: incl is monotone)
                atomicMax(&sm.count, incl);                          // keys taken
                const uint2 *seg = p.cand + (long long)row * p.cap + (long long)t * 256;
                for (int e = 0; e < c; ++e) {
                    const uint2 cd = seg[e];
                    const unsigned long long key = make_key(__uint_as_float(cd.x), cd.y);
                    keys[n_cur + incl - c + e] = key;
                    nz_mine += key != 0ull;
                }
            }
            __syncthreads();
            m = sm.count;
            step = sm.rank;
            __syncthreads();
        } else {
            m = (n_new - pos) < room ? (n_new - pos) : room;
            step = m;
            for (int i = tid; i < m; i += SEL_THREADS) {
                const int e = pos + i;
                unsigned long long key;
                if constexpr (SRC == SRC_DENSE) {
                    const float *d = (const float *)p.cand + (long long)row * 2 * p.cap;
                    key = make_key(d[e], (unsigned)(p.dense_col0 + e));
                } else {
                    const int g = e / p.kc, j = e - g * p.kc;
                    const long long off = ((long long)g * p.B + row) * p.kc + j;
                    const long long gi = p.g_idx[off];
                    key = gi < 0 ? 0ull : make_key(p.g_val[off], (unsigned)gi);
                }
                keys[n_cur + i] = key;
                nz_mine += key != 0ull;
            }
        }
        if (tid == 0) sm.count = 0;
        __syncthreads();
        // real (non-zero) keys in this batch; the kept keys of earlier passes are all real
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz_mine += __shfl_xor(nz_mine, d, 64);
        if ((tid & 63) == 0 && nz_mine) atomicAdd(&sm.count, nz_mine);
        __syncthreads();
        const int tot = n_cur + m;
        const int nz = n_cur + sm.count;
        const int keep = nz < p.k ? nz : p.k;
        __syncthreads();                                          // sm.count is reused below
        unsigned long long kth = 1ull;                            // nz <= k: keep every real key
        if (nz > keep) kth = radix_select_kth(sm, tot, keep, tid);
        if (tid == 0) sm.count = 0;
        __syncthreads();
        for (int i = tid; i < tot; i += SEL_THREADS) {
            const unsigned long long key = keys[i];
            if (key >= kth) {                                     // exactly `keep` keys: they are distinct
                const int slot = atomicAdd(&sm.count, 1);
                if (slot < SORT_OUT) sm.out[slot] = key;
            }
        }
        int n2 = 64;
        while (n2 < keep) n2 <<= 1;
        __syncthreads();
        for (int i = keep + tid; i < n2; i += SEL_THREADS) sm.out[i] = 0ull;
        __syncthreads();
        bitonic_sort_desc(sm.out, n2, tid);
        for (int i = tid; i < keep; i += SEL_THREADS) keys[i] = sm.out[i];
        n_cur = keep;
        pos += step;
        __syncthreads();
    }
    __syncthreads();

    if constexpr (SRC == SRC_LIST) {
        for (int i = tid; i < p.mask_words; i += SEL_THREADS) p.tmask[(long long)row * p.mask_words + i] = 0u;
    }
    if constexpr (SRC != SRC_GATHER) {
        for (int i = tid; i < n_cur; i += SEL_THREADS) p.topk[(long long)row * p.k + i] = keys[i];
        if (tid == 0) {
            p.topk_n[row] = n_cur;
            // tau only ever rises: it stays at its initial -inf (or at the bootstrap sample's threshold) until k keys are held
            if (n_cur >= p.k) p.tau[row] = key_value(keys[p.k - 1]);
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

template <int SRC>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const SelectParams p)
{
    __shared__ SelectSmem sm;
    if constexpr (SRC == SRC_LIST) {
        if (p.big_rows) {
            // workgroup b serves sub-list b % BIG_SUBLISTS, every (gridDim / BIG_SUBLISTS)-th entry
            const unsigned sub = blockIdx.x & (BIG_SUBLISTS - 1), per = gridDim.x / BIG_SUBLISTS;
            const unsigned n = p.big_cnt[(p.parity * BIG_SUBLISTS + sub) * 32];
            for (unsigned i = blockIdx.x / BIG_SUBLISTS; i < n; i += per) {
                select_row<SRC>(p, p.big_rows[(long long)sub * p.big_stride + i], sm);
                __syncthreads();                                   // the LDS is reused by the next row
            }
            return;
        }
    }
    select_row<SRC>(p, blockIdx.x, sm);
}

// ---------------------------------------------------------------------------------------------
// Wave-per-row select (k <= 256): the workgroup-per-row kernel above spends its time in barriers
// (~60 per row) with three of its four waves idle in the scans; here ONE wave owns a row, its keys
// sit in a wave-private LDS block, and nothing waits for another wave.  LDS operations of one wave
// complete in order, so only the compiler has to be kept from moving accesses across the steps.
constexpr int WSEL_CAP = 1024;               // keys of one row in LDS: the kept <= 256 + one batch of candidates
constexpr int WSEL_MAX_K = 256;

struct WaveSmem {
    unsigned long long keys[WSEL_CAP];
    unsigned hist[256];
};

__device__ __forceinline__ void wave_step() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// rank-th largest (1-based) of the non-zero keys in sm.keys[0, tot): as radix_select_kth, by one wave.
__device__ __forceinline__ unsigned long long wave_radix_select_kth(WaveSmem &sm, int tot, int rank, int lane)
{
    unsigned long long a = ~0ull, o = 0ull;
    for (int i = lane; i < tot; i += 64) {
        const unsigned long long key = sm.keys[i];
        if (key) { a &= key; o |= key; }
    }
    a = wave_and64(a);
    o = wave_or64(o);
    const unsigned long long var = a ^ o;
    if (var == 0ull) return a;                                   // a single distinct key
    const int hb = 63 - __clzll((long long)var);
    unsigned long long pmask = hb == 63 ? 0ull : ~((2ull << hb) - 1ull);
    unsigned long long prefix = a & pmask;
    unsigned r = (unsigned)rank;
    for (int top = hb; top >= 0; top -= 8) {
        const int sh = top >= 7 ? top - 7 : 0;
        const unsigned dmask = top >= 7 ? 255u : ((1u << (top + 1)) - 1u);
        *(uint4 *)&sm.hist[4 * lane] = make_uint4(0u, 0u, 0u, 0u);
        wave_step();
        for (int i = lane; i < tot; i += 64) {
            const unsigned long long key = sm.keys[i];
            if (key && (key & pmask) == prefix) atomicAdd(&sm.hist[(unsigned)(key >> sh) & dmask], 1u);
        }
        wave_step();
        const uint4 h = *(const uint4 *)&sm.hist[4 * lane];      // lane l owns bins 4l .. 4l+3
        const unsigned mine = h.x + h.y + h.z + h.w;
        unsigned incl = mine;                                    // sum over lanes >= this one
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_down(incl, d, 64);
            if (lane + d < 64) incl += t;
        }
        const unsigned above = incl - mine;
        const bool hit = above < r && r <= incl;                 // exactly one lane
        unsigned acc = above, hcnt = h.w;
        int b = 4 * lane + 3;
        if (acc + h.w < r) { acc += h.w; b = 4 * lane + 2; hcnt = h.z;
            if (acc + h.z < r) { acc += h.z; b = 4 * lane + 1; hcnt = h.y;
                if (acc + h.y < r) { acc += h.y; b = 4 * lane; hcnt = h.x; } } }
        const int src = __ffsll((long long)__ballot(hit)) - 1;
        const unsigned bin = (unsigned)__shfl(b, src, 64);
        const unsigned bin_count = __shfl(hcnt, src, 64);
        r = __shfl(r - acc, src, 64);
        prefix |= (unsigned long long)bin << sh;
        pmask |= (unsigned long long)dmask << sh;
        if (sh == 0) break;
        if (bin_count == 1u) {                                   // one key left under this prefix: it is the answer
            unsigned long long found = 0ull;
            for (int i = lane; i < tot; i += 64) {
                const unsigned long long key = sm.keys[i];
                if (key && (key & pmask) == prefix) found = key;
            }
            return wave_or64(found);
        }
        wave_step();
    }
    return prefix;
}

// Descending bitonic sort of n (power of two, 64 <= n <= WSEL_CAP) keys in a wave-private LDS block.
__device__ __forceinline__ void wave_bitonic_sort_desc(unsigned long long *keys, int n, int lane)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = lane; t < (n >> 1); t += 64) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            wave_step();
        }
    }
}

template <int SRC>
__device__ __forceinline__ void wave_select_row(const SelectParams &p, const long long row, WaveSmem &sm, const int lane)
{
    static_assert(SRC == SRC_LIST || SRC == SRC_DENSE, "wave select: candidate lists or dense rows");
    unsigned long long *keys = sm.keys;
    const int n_new = SRC == SRC_LIST ? p.ntiles : p.dense_n;      // SRC_LIST: tiles to consume (counts are per tile)
    const bool finalize = p.out_val != nullptr;
    if (n_new == 0 && !finalize) return;
    int n_cur = p.topk_n[row];
    {
        // every global load of a batch is issued before the first LDS store waits for one (k <= 256: 4 per lane)
        unsigned long long pk[WSEL_MAX_K / 64];
#pragma unroll
        for (int j = 0; j < WSEL_MAX_K / 64; ++j) pk[j] = (j * 64 + lane) < n_cur ? p.topk[row * p.k + j * 64 + lane] : 0ull;
#pragma unroll
        for (int j = 0; j < WSEL_MAX_K / 64; ++j)
            if ((j * 64 + lane) < n_cur) keys[j * 64 + lane] = pk[j];
    }
    bool touched = false;
    bool have_kth = false;                                         // SRC_DENSE: last_kth = smallest of the k kept keys
    unsigned long long last_kth = 0ull;
    int pos = 0;
    while (pos < n_new) {
        const int room = WSEL_CAP - n_cur;
        int m, step, nz = 0;
        if constexpr (SRC == SRC_LIST) {
            // Runs of 64 tiles (one per lane) are appended while their candidates fit the room; the select below
            // runs once per filled batch, not once per run: a stage of several launches has thousands of tiles,
            // most of them empty for a given row (a tile holds <= 256, room >= 768: progress).
            const unsigned *tm = p.tmask + row * p.mask_words;
            m = 0;
            step = 0;
            for (;;) {
                const int p0 = pos + step;
                if (p0 >= n_new) break;
                const int t = p0 + lane;
                int c = 0;
                if (t < n_new && ((tm[t >> 5] >> (t & 31)) & 1u)) c = (int)p.cnt2[(long long)t * p.bpad + row];
                int incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                const bool fits = t < n_new && incl <= room - m;
                const int took = __popcll(__ballot(fits));          // a prefix of the lanes: incl is monotone
                if (took == 0) break;
                if (fits && c) {
                    // 8 candidates (4 x 16 bytes) in flight per lane; slot c of an odd count is read (inside the
                    // segment) and ignored
                    const uint4 *seg = (const uint4 *)(p.cand + row * p.cap + (long long)t * 256);
                    unsigned long long *dst = keys + n_cur + m + incl - c;
                    for (int e0 = 0; e0 < c; e0 += 8) {
                        uint4 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (e0 + 2 * j < c) v[j] = seg[(e0 >> 1) + j];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int e = e0 + 2 * j;
                            if (e < c) {
                                const unsigned long long key = make_key(__uint_as_float(v[j].x), v[j].y);
                                dst[e] = key;
                                nz += key != 0ull;
                            }
                            if (e + 1 < c) {
                                const unsigned long long key = make_key(__uint_as_float(v[j].z), v[j].w);
                                dst[e + 1] = key;
                                nz += key != 0ull;
                            }
                        }
                    }
                }
                m += __shfl(incl, took - 1, 64);
                step += took;
                if (p0 + took < n_new && took < 64) break;          // the room is used up: select, then go on from here
            }
        } else if (have_kth) {
            // k keys are held and their smallest is known: of the next values only those above it can matter
            // (in a random row as many as are held already), so they are filtered on the way into LDS -- 256
            // values per round while a round's worst case (all 256 pass) still fits the room
            const float *d = (const float *)p.cand + row * 2 * p.cap + pos;
            const int left = n_new - pos;
            int appended = 0, consumed = 0;
            while (consumed < left && room - appended >= 256) {
                const int i = consumed + 4 * lane;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < left) v = *(const float4 *)(d + i);
                const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned long long key = (i + u < left) ? make_key(f[u], (unsigned)(p.dense_col0 + pos + i + u)) : 0ull;
                    const bool pass = key > last_kth;
                    const unsigned long long bal = __ballot(pass);
                    if (pass) keys[n_cur + appended + __popcll(bal & ((1ull << lane) - 1ull))] = key;
                    appended += __popcll(bal);
                }
                consumed += 256;
            }
            m = appended;
            step = consumed < left ? consumed : left;
            nz = lane == 0 ? appended : 0;
        } else {
            m = (n_new - pos) < room ? (n_new - pos) : (room & ~3);     // pos stays a multiple of 4: 16-byte loads
            step = m;
            const float *d = (const float *)p.cand + row * 2 * p.cap + pos;
            for (int i0 = 0; i0 < m; i0 += 1024) {                       // 4 x 16 bytes in flight per lane
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 4 * (j * 64 + lane);
                    if (i < m) v[j] = *(const float4 *)(d + i);          // the row is padded to a multiple of 1024 floats
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + 4 * (j * 64 + lane);
                    const float f[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i + u < m) {
                            const unsigned long long key = make_key(f[u], (unsigned)(p.dense_col0 + pos + i + u));
                            keys[n_cur + i + u] = key;
                            nz += key != 0ull;
                        }
                    }
                }
            }
        }
        if (m == 0) { pos += step; continue; }                     // no candidate in the tiles consumed: nothing to select
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nz += __shfl_xor(nz, d, 64);
        wave_step();
        const int tot = n_cur + m;
        nz += n_cur;                                               // the kept keys of earlier batches are all real
        const int keep = nz < p.k ? nz : p.k;
        unsigned long long kth = 1ull;                             // nz <= k: keep every real key
        if (nz > keep) kth = wave_radix_select_kth(sm, tot, keep, lane);
        if (SRC == SRC_DENSE && nz > keep) { have_kth = true; last_kth = kth; }      // keep == k here
        wave_step();
        // ordered in-place compaction of the `keep` keys >= kth (they are distinct): a chunk is read before it
        // is written, and writes never reach a chunk not yet read
        int base = 0;
        for (int c0 = 0; c0 < tot; c0 += 64) {
            const int i = c0 + lane;
            const unsigned long long key = i < tot ? keys[i] : 0ull;
            const bool take = key >= kth;
            const unsigned long long bal = __ballot(take);
            wave_step();
            if (take) keys[base + __popcll(bal & ((1ull << lane) - 1ull))] = key;
            base += __popcll(bal);
        }
        wave_step();
        n_cur = keep;
        pos += step;
        touched = true;
    }
    if (touched) {
        int n2 = 64;
        while (n2 < n_cur) n2 <<= 1;
        for (int i = n_cur + lane; i < n2; i += 64) keys[i] = 0ull;
        wave_step();
        wave_bitonic_sort_desc(keys, n2, lane);
    }
    if constexpr (SRC == SRC_LIST) {
        for (int i = lane; i < p.mask_words; i += 64) p.tmask[row * p.mask_words + i] = 0u;
    }
    for (int i = lane; i < n_cur; i += 64) p.topk[row * p.k + i] = keys[i];
    if (lane == 0) {
        p.topk_n[row] = n_cur;
        // tau only ever rises: it stays at its initial -inf (or at the bootstrap sample's threshold) until k keys are held
        if (n_cur >= p.k) p.tau[row] = key_value(keys[p.k - 1]);
    }
    if (finalize) {
        for (int i = lane; i < p.k; i += 64) {
            const bool ok = i < n_cur;
            p.out_val[row * p.k + i] = ok ? key_value(keys[i]) : -INFINITY;
            p.out_idx[row * p.k + i] = ok ? p.idx_base + (long long)key_row(keys[i]) : -1ll;
        }
    }
    wave_step();                                                   // the LDS block is reused by this wave's next row
}

// SRC_DENSE: wave w of the grid serves query row w.  SRC_LIST: the rows merge_kernel left in the 64 work lists;
// wave w serves list w % 64, every (waves / 64)-th entry.
template <int SRC>
__global__ __launch_bounds__(256) void wave_select_kernel(const SelectParams p)
{
    __shared__ WaveSmem sm[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned gw = blockIdx.x * 4 + w;
    if constexpr (SRC == SRC_LIST) {
        const unsigned sub = gw & (BIG_SUBLISTS - 1), per = gridDim.x * 4 / BIG_SUBLISTS;
        const unsigned n = p.big_cnt[(p.parity * BIG_SUBLISTS + sub) * 32];
        for (unsigned i = gw / BIG_SUBLISTS; i < n; i += per)
            wave_select_row<SRC>(p, (long long)p.big_rows[(long long)sub * p.big_stride + i], sm[w], lane);
    } else {
        if ((long long)gw < p.B) wave_select_row<SRC>(p, (long long)gw, sm[w], lane);
    }
}

// Wave-per-row merge: the common case after the first few stages is a handful of new candidates
// per row.  One wave holds the row's sorted top-k in registers (KPL keys per lane, key i on lane
// i % 64) and its <= 64 new candidates one per lane, and computes every key's position in the
// merged order by counting -- no LDS, no barrier, no sort:
//   new key c   -> position = #prev > c + #new > c
//   prev key i  -> position = i + #new > prev_i
// Keys whose position is unchanged are not rewritten.  Rows with more than MERGE_MAX_NEW candidates
// are left to select_kernel<SRC_LIST> (skip_small = 1), which is launched right after.
template <int KPL>
__global__ __launch_bounds__(256) void merge_kernel(const SelectParams p)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B) return;
    __shared__ uint2 stage[4][MERGE_MAX_NEW];
    const bool finalize = p.out_val != nullptr;
    if (blockIdx.x == 0 && threadIdx.x < BIG_SUBLISTS && p.big_cnt)
        p.big_cnt[((p.parity ^ 1) * BIG_SUBLISTS + threadIdx.x) * 32] = 0u;                  // the next stage's counters
    // this row's candidates: cnt2[t][row] of them in the segment of every bank tile t of the stage
    const unsigned *tm = p.tmask + row * p.mask_words;
    int mine = 0;
    for (int t = lane; t < p.ntiles; t += 64)
        if ((tm[t >> 5] >> (t & 31)) & 1u) mine += (int)p.cnt2[(long long)t * p.bpad + row];
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    const unsigned c = (unsigned)__shfl(incl, 63, 64);
    if (c > (unsigned)MERGE_MAX_NEW) {
        // one returning atomic per row on one word saturates at ~88 per us (16384 rows: 187 us);
        // the rows are spread over 64 sub-lists instead, each with its own counter line
        if (lane == 0 && p.big_rows) {
            const unsigned sub = (unsigned)(row >> 2) & (BIG_SUBLISTS - 1);
            const unsigned slot = atomicAdd(p.big_cnt + (p.parity * BIG_SUBLISTS + sub) * 32, 1u);
            p.big_rows[(long long)sub * p.big_stride + slot] = (int)row;
        }
        return;
    }
    if (c == 0 && !finalize) return;
    const int n_new = (int)c;
    const int k = p.k;
    const int n_cur = p.topk_n[row];
    unsigned long long *topk = p.topk + row * k;

    unsigned long long pk[KPL];
    int shift[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i = j * 64 + lane;
        pk[j] = i < n_cur ? topk[i] : 0ull;
        shift[j] = 0;
    }
    // gather the <= 64 candidates, one per lane, through a wave-private LDS patch (any order)
    uint2 *st = stage[threadIdx.x >> 6];
    if (mine) {
        int o = incl - mine;
        for (int t = lane; t < p.ntiles; t += 64) {
            if (!((tm[t >> 5] >> (t & 31)) & 1u)) continue;
            const int n = (int)p.cnt2[(long long)t * p.bpad + row];
            const uint2 *seg = p.cand + row * p.cap + (long long)t * 256;
            for (int e = 0; e < n; ++e) st[o++] = seg[e];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // same wave: LDS ops complete in order
    if (c) {                                                           // consumed: the next stage starts from a clear mask
        for (int i = lane; i < p.mask_words; i += 64) p.tmask[row * p.mask_words + i] = 0u;
    }
    unsigned long long ck = 0ull;
    if (lane < n_new) {
        const uint2 cd = st[lane];
        ck = make_key(__uint_as_float(cd.x), cd.y);
    }
    const int n_valid_new = __popcll(__ballot(ck != 0ull));
    int my_pos = 0;
    for (int l = 0; l < n_new; ++l) {
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)ck, l);
        const unsigned hi = __builtin_amdgcn_readlane((unsigned)(ck >> 32), l);
        const unsigned long long cl = ((unsigned long long)hi << 32) | lo;     // wave-uniform
        if (cl == 0ull) continue;
        int above = __popcll(__ballot(ck > cl));                                 // #new > c_l
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const bool gt = pk[j] > cl;                                          // empty slots (0) never are
            above += __popcll(__ballot(gt));                                     // #prev > c_l
            shift[j] += (pk[j] != 0ull && !gt) ? 1 : 0;                          // c_l outranks prev_i
        }
        if (lane == l) my_pos = above;
    }
    const int n_out = (n_cur + n_valid_new) < k ? (n_cur + n_valid_new) : k;
    // scatter to the merged positions (all reads above are complete: the stores depend on them)
    if (ck != 0ull && my_pos < k) {
        topk[my_pos] = ck;
        if (my_pos == k - 1) p.tau[row] = key_value(ck);
    }
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i = j * 64 + lane;
        const int np = i + shift[j];
        if (pk[j] != 0ull && shift[j] > 0 && np < k) {
            topk[np] = pk[j];
            if (np == k - 1) p.tau[row] = key_value(pk[j]);
        }
    }
    if (lane == 0) p.topk_n[row] = n_out;
    if (finalize) {
        // positions are final; emit (value, index) for every slot of this row from the merged keys
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            const int i = j * 64 + lane;
            const int np = i + shift[j];
            if (pk[j] != 0ull && np < k) {
                p.out_val[row * k + np] = key_value(pk[j]);
                p.out_idx[row * k + np] = p.idx_base + (long long)key_row(pk[j]);
            }
        }
        if (ck != 0ull && my_pos < k) {
            p.out_val[row * k + my_pos] = key_value(ck);
            p.out_idx[row * k + my_pos] = p.idx_base + (long long)key_row(ck);
        }
        for (int i = n_out + lane; i < k; i += 64) {
            p.out_val[row * k + i] = -INFINITY;
            p.out_idx[row * k + i] = -1ll;
        }
    }
}

// Bitonic sort of 256 keys held by one wave in REGISTERS, 4 per lane (sorted position i = 4 * lane + s): the two
// innermost strides of every merge step are register-to-register inside a lane, the others fetch the partner lane's
// key with one cross-lane read each (lane ^ 1, 2: DPP quad_perm; lane ^ 4 .. 32: ds_bpermute, which uses the LDS
// crossbar but no LDS memory).  36 compare-exchange layers, fully unrolled; no LDS traffic, no barrier.
template <typename T> __device__ __forceinline__ T lane_xor(T v, int m);
template <> __device__ __forceinline__ unsigned lane_xor<unsigned>(unsigned v, int m)
{
    if (m == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    if (m == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    return (unsigned)__shfl_xor((int)v, m, 64);
}
template <> __device__ __forceinline__ unsigned long long lane_xor<unsigned long long>(unsigned long long v, int m)
{
    const unsigned lo = lane_xor<unsigned>((unsigned)v, m), hi = lane_xor<unsigned>((unsigned)(v >> 32), m);
    return ((unsigned long long)hi << 32) | lo;
}

template <typename T, bool DESC>
__device__ __forceinline__ void wave_sort256(T (&k)[4], int lane)
{
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 4) {
                const int m = stride >> 2;
                const bool lower = (lane & m) == 0;
                const bool up = (size == 256) || ((lane & (size >> 2)) == 0);       // this block sorts towards the final order
                const bool keep_first = (lower == up);                                // keep the key that comes first in the final order
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    const T o = lane_xor<T>(k[s2], m);
                    const bool o_first = DESC ? (o > k[s2]) : (o < k[s2]);            // the partner's key comes first
                    k[s2] = (o_first == keep_first) ? o : k[s2];
                }
            } else {
#pragma unroll
                for (int lo = 0; lo < 4; ++lo) {
                    if (lo & stride) continue;
                    const int hi = lo | stride;
                    const bool up = (size >= 4) ? ((size == 256) || ((lane & (size >> 2)) == 0)) : ((lo & size) == 0);
                    const T a = k[lo], b = k[hi];
                    const bool b_first = DESC ? (b > a) : (b < a);
                    const bool swap = (b_first == up);
                    k[lo] = swap ? b : a;
                    k[hi] = swap ? a : b;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Register-resident threshold refresh (k <= 256): ONE wave folds ONE query row's new candidates into
// the row's kept keys and raises its threshold -- the stage refresh of the filter scheme.
//
// The kept keys are a BAG (unsorted, compact: slots [0, topk_n) of the row): between stages only the
// threshold matters, the order is needed once, at the end.  A refresh is therefore a SELECTION, not a
// sort, and it runs in registers.  The kernel is latency-, not bandwidth-bound (a row moves ~4 KB), so
// it is built for many rows in flight (<= 64 VGPRs: 8 waves per SIMD, 4 KiB of LDS per wave) and a short
// chain of dependent memory round trips:
//   * the row's candidates are located by a per-tile exclusive prefix of the stage's counts (cnt2, through
//     the row's tile bitmask) in a wave-private LDS window; candidate j finds its (tile, slot) by binary
//     search and every lane loads its candidates straight into registers, all loads of a batch in flight
//     together; the kept keys are requested before that chain starts;
//   * the k-th largest of kept + new keys comes from an MSB-first BINARY radix select on wave ballots: one
//     v_and + v_cmp + s_bcnt1 per register slot and key bit, only over the bits in which the keys differ at
//     all, stopping when one key is left (about log2(keys) + a few steps); keys that drop out are zeroed in
//     a working copy, so no activity mask is carried; no LDS histogram, no atomics, no barrier;
//   * the survivors (exactly k: keys are distinct) are compacted through 2 KiB of wave-private LDS back into
//     the bag; the final stage sorts the bag once, in registers (wave_sort256), and emits (value, index).
// This replaces the counting merge + LDS radix-select/compact/sort pair for k <= 256; k > 256 keeps the
// workgroup-per-row kernels above.
constexpr int BAG_WIN = 512;           // bank tiles per prefix window (2 KiB of LDS per wave)
constexpr int BAG_TPL = BAG_WIN / 64;  // consecutive tiles per lane
constexpr int BAG_NSLOT = 4;           // new candidates per lane and batch (256 per batch)

struct BagSmem {
    unsigned pre[BAG_WIN];             // exclusive prefix of the window's per-tile candidate counts
    unsigned long long bag[256];       // survivors of a batch, compacted; sorted in place by the final stage
};

// wave-uniform values the compiler cannot prove uniform (results of cross-lane ops, per-lane loads of one
// address): moved to scalar registers so that the loops and branches on them are scalar
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v, unsigned old = 0u)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false);
}
// AND / OR over the 64 lanes (result wave-uniform, in scalar registers): xor-butterfly inside each 16-lane
// DPP row (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the four rows by readlane
__device__ __forceinline__ unsigned wave_and_u32(unsigned v)
{
    v &= dpp_u32<0xB1>(v); v &= dpp_u32<0x4E>(v); v &= dpp_u32<0x141>(v); v &= dpp_u32<0x140>(v);
    return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) & __builtin_amdgcn_readlane((int)v, 16) &
                      __builtin_amdgcn_readlane((int)v, 32) & __builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v)
{
    v |= dpp_u32<0xB1>(v); v |= dpp_u32<0x4E>(v); v |= dpp_u32<0x141>(v); v |= dpp_u32<0x140>(v);
    return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) | __builtin_amdgcn_readlane((int)v, 16) |
                      __builtin_amdgcn_readlane((int)v, 32) | __builtin_amdgcn_readlane((int)v, 48));
}
// inclusive prefix sum over the 64 lanes + the wave total (uniform): row_shr 1/2/4/8 inside the DPP rows
// (lanes shifted in from outside the row contribute `old` = 0), then the row totals by readlane
__device__ __forceinline__ int wave_scan_incl(int v, int lane, int &total)
{
    unsigned x = (unsigned)v;
    x += dpp_u32<0x111>(x); x += dpp_u32<0x112>(x); x += dpp_u32<0x114>(x); x += dpp_u32<0x118>(x);
    const int t0 = __builtin_amdgcn_readlane((int)x, 15), t1 = __builtin_amdgcn_readlane((int)x, 31);
    const int t2 = __builtin_amdgcn_readlane((int)x, 47), t3 = __builtin_amdgcn_readlane((int)x, 63);
    const int r = lane >> 4;
    total = t0 + t1 + t2 + t3;
    return (int)x + (r > 0 ? t0 : 0) + (r > 1 ? t1 : 0) + (r > 2 ? t2 : 0);
}

// One phase of the MSB-first binary radix select over the 32-bit words w[NK] of one wave (0 = not a candidate):
// narrows the candidates to those that share the rank-th largest word's bits, over the bits set in `var` (the bits
// in which the words differ at all), until one candidate is left or the bits run out.  Words that drop out are
// zeroed.  r: rank among the candidates (1-based), n_act: candidates; both are updated.  Per slot and bit:
// v_bfe_i32 + v_cmp (+ s_bcnt1) to count, v_and to narrow.
template <int NK>
__device__ __forceinline__ void radix_narrow_u32(unsigned (&w)[NK], unsigned var, int &r, int &n_act)
{
    if (n_act <= 1 || !var) return;
    for (int bit = 31 - __builtin_clz(var); bit >= 0 && n_act > 1; --bit) {
        if (!(var & (1u << bit))) continue;                     // every word agrees on this bit (scalar branch)
        int cnt1 = 0;
        if constexpr (NK <= 8) {
            int mb[NK];                                         // -1 where the bit is set, else 0
#pragma unroll
            for (int i = 0; i < NK; ++i) { mb[i] = __builtin_amdgcn_sbfe((int)w[i], bit, 1); cnt1 += __popcll(__ballot(mb[i] != 0)); }
            if (cnt1 >= r) {                                    // the rank-th largest has this bit set: zeros drop out
                n_act = cnt1;
#pragma unroll
                for (int i = 0; i < NK; ++i) w[i] &= (unsigned)mb[i];
            } else {                                            // it has the bit clear: ones drop out
                r -= cnt1; n_act -= cnt1;
#pragma unroll
                for (int i = 0; i < NK; ++i) w[i] &= ~(unsigned)mb[i];
            }
        } else {                                                // many slots: recompute the bit instead of keeping NK masks
#pragma unroll
            for (int i = 0; i < NK; ++i) cnt1 += __popcll(__ballot(__builtin_amdgcn_sbfe((int)w[i], bit, 1) != 0));
            if (cnt1 >= r) {
                n_act = cnt1;
#pragma unroll
                for (int i = 0; i < NK; ++i) w[i] &= (unsigned)__builtin_amdgcn_sbfe((int)w[i], bit, 1);
            } else {
                r -= cnt1; n_act -= cnt1;
#pragma unroll
                for (int i = 0; i < NK; ++i) w[i] &= ~(unsigned)__builtin_amdgcn_sbfe((int)w[i], bit, 1);
            }
        }
    }
}

// rank-th largest (1-based) of the non-zero keys held in registers by one wave (NK slots per lane: kept keys and
// new ones alike).  Needs nz >= rank >= 1 non-zero keys, all distinct.  Every lane returns the same key.
template <int NK>
__device__ __forceinline__ unsigned long long bag_select_kth(const unsigned long long (&key)[NK], int nz, int rank)
{
    unsigned alo = ~0u, ahi = ~0u, olo = 0u, ohi = 0u;
    unsigned w[NK];                                             // working copy: the key word under test, 0 = dropped out
#pragma unroll
    for (int i = 0; i < NK; ++i) {
        const unsigned lo = (unsigned)key[i], hi = (unsigned)(key[i] >> 32);
        w[i] = hi;                                              // hi = orderable(value) is never 0 for a real key
        if (key[i]) { alo &= lo; ahi &= hi; olo |= lo; ohi |= hi; }
    }
    alo = wave_and_u32(alo); ahi = wave_and_u32(ahi);
    olo = wave_or_u32(olo); ohi = wave_or_u32(ohi);
    int r = rank, n_act = nz;
    radix_narrow_u32<NK>(w, ahi ^ ohi, r, n_act);               // by value ...
    if (n_act > 1) {                                            // ... equal values: go on through the row word
#pragma unroll
        for (int i = 0; i < NK; ++i) w[i] = w[i] ? (unsigned)key[i] : 0u;      // 0xFFFFFFFF - row is never 0 (row < 2^31)
        radix_narrow_u32<NK>(w, alo ^ olo, r, n_act);
    }
    // one key is left (keys are distinct): the slot whose working word is still non-zero
    unsigned flo = 0u, fhi = 0u;
#pragma unroll
    for (int i = 0; i < NK; ++i)
        if (w[i]) { flo = (unsigned)key[i]; fhi = (unsigned)(key[i] >> 32); }
    const unsigned long long has = __ballot((flo | fhi) != 0u);
    const int src = has ? __builtin_ctzll(has) : 0;
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)fhi, src) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)flo, src);
}

// sum over the 64 lanes (result wave-uniform): xor-butterfly inside each 16-lane DPP row, then the four rows by readlane
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
    v += dpp_u32<0xB1>(v); v += dpp_u32<0x4E>(v); v += dpp_u32<0x141>(v); v += dpp_u32<0x140>(v);
    return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16) +
                      __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
}

// 32 x 32 bit-matrix transpose in registers (the five-stage butterfly): afterwards word i holds bit (31 - i) of every
// input word, input word c at bit position (31 - c).
__device__ __forceinline__ void bit_transpose32(unsigned (&a)[32])
{
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int j = 16 >> st;
        const unsigned m = st == 0 ? 0x0000FFFFu : st == 1 ? 0x00FF00FFu : st == 2 ? 0x0F0F0F0Fu : st == 3 ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k & j) continue;
            const unsigned t = (a[k] ^ (a[k + j] >> j)) & m;
            a[k] ^= t;
            a[k + j] ^= t << j;
        }
    }
}

// First threshold from the bootstrap SAMPLE (mi_knn_topk, sampled mode): tau[row] = k-th largest of the row's 2048
// dense similarities (duplicates count, NaNs do not), left at -inf when fewer than k are real.  One wave per row, 32
// values per lane in registers; nothing else is kept -- the sampled bank rows go through the filter like every other row.
// The MSB-first binary radix select runs on BIT-SLICED keys: the lane's 32 order-preserving words are transposed once
// (32 x 32 bits, ~480 VALU), after which word b holds bit b of all 32 slots, the set of slots still in the race is one
// 32-bit mask per lane, and a radix step is v_and + v_bcnt + a wave sum + one mask update (~17 instructions) whatever the
// slot count -- against v_bfe + v_cmp + s_bcnt1 + v_and PER SLOT and step of the ballot form (~190 for 32 slots).
template <int NV>
__global__ __launch_bounds__(256) void sample_tau_kernel(const SelectParams p)
{
    static_assert(NV == 32, "one 32 x 32 bit matrix per lane");
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B) return;
    const float4 *d = (const float4 *)((const float *)p.cand + row * 2 * p.cap);      // dense row: NV * 64 floats
    unsigned w[NV];
    unsigned a = ~0u, o = 0u, act = 0u;
#pragma unroll
    for (int j = 0; j < NV / 4; ++j) {
        const float4 v = d[j * 64 + lane];
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned x = (f[u] != f[u]) ? 0u : orderable_f32(f[u]);             // NaN: not a candidate
            w[4 * j + u] = x;
            if (x) { a &= x; o |= x; act |= 0x80000000u >> (4 * j + u); }             // slot c lives at bit 31 - c
        }
    }
    int n_act = (int)wave_sum_u32((unsigned)__popc(act));
    if (n_act < p.k) return;                                    // tau (and tau0) stay -inf: the row is never speculated on
    a = wave_and_u32(a);
    o = wave_or_u32(o);
    const unsigned var = a ^ o;                                 // the bits in which the words differ at all
    bit_transpose32(w);                                         // w[31 - b] = bit b of every slot
    // rank-th largest word (duplicates count) among the slots in `act0`
    auto select = [&](int rank) {
        unsigned in = act;
        int r = rank;
        unsigned kth = a & ~var;                                // the bits every word agrees on
#pragma unroll
        for (int b = 31; b >= 0; --b) {
            if (!(var & (1u << b))) continue;                   // wave-uniform: scalar branch
            const unsigned ones = in & w[31 - b];               // slots in the race that have bit b set
            const int cnt1 = (int)wave_sum_u32((unsigned)__popc(ones));
            if (cnt1 >= r) {                                    // the r-th largest has the bit set: the zeros drop out
                in = ones;
                kth |= 1u << b;
            } else {                                            // it has the bit clear: the ones drop out
                r -= cnt1;
                in &= ~w[31 - b];
            }
        }
        return kth;
    };
    const unsigned safe = select(p.k);
    const unsigned spec = (p.tau0 && p.spec_rank < p.k) ? select(p.spec_rank) : safe;
    if (lane == 0) {
        p.tau[row] = from_orderable(spec);                      // what the filter stages start with
        if (p.tau0) p.tau0[row] = from_orderable(safe);         // what a failed row is redone with
        p.topk_n[row] = 0;
    }
}

template <int KPL>
__global__ __launch_bounds__(256, 6) void bag_refresh_kernel(const SelectParams p)
{
    __shared__ BagSmem smem[4];
    constexpr int NK = KPL + BAG_NSLOT;                         // key[0, KPL): kept bag; key[KPL, NK): new candidates
    constexpr int BATCH = BAG_NSLOT * 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wv;
    if (row >= p.B) return;
    if (p.fb_rows && row >= (long long)*p.fb_cnt - p.fb_base) return;      // fallback launch: compact rows in use only
    BagSmem &sm = smem[wv];
    const bool finalize = p.out_val != nullptr;
    const int k = p.k;
    unsigned *tm = p.tmask + row * p.mask_words;
    const unsigned long long ltmask = (1ull << lane) - 1ull;

    unsigned long long key[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) key[i] = 0ull;
    bool have_bag = false, any = false;
    int n_cur = 0, fill = 0, nz_new = 0;                        // fill: batch positions taken; nz_new: real keys among them
    unsigned long long kth_last = 0ull;
    auto load_bag = [&]() {                                     // deferred: rows without candidates never read it
        n_cur = uniform_i32(p.topk_n[row]);
#pragma unroll
        for (int j = 0; j < KPL; ++j) key[j] = (j * 64 + lane) < n_cur ? p.topk[row * k + j * 64 + lane] : 0ull;
        have_bag = true;
    };
    // fold the batch in registers into the bag: select the k-th largest of kept + new, compact the survivors
    auto fold = [&]() {
        const int nz = n_cur + nz_new;
        const int keep = nz < k ? nz : k;
        unsigned long long kth = 1ull;                          // fewer than k keys: keep every real one
        if (nz >= k) { kth = bag_select_kth<NK>(key, nz, k); kth_last = kth; }
        int base = 0;
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const bool take = key[i] >= kth && key[i] != 0ull;
            const unsigned long long bal = __ballot(take);
            const int at = base + __popcll(bal & ltmask);
            if (take && at < k) sm.bag[at] = key[i];            // exactly `keep` survivors (distinct keys); `< k` guards LDS regardless
            base += __popcll(bal);
        }
        wave_step();
        n_cur = keep;
#pragma unroll
        for (int j = 0; j < KPL; ++j) key[j] = (j * 64 + lane) < n_cur ? sm.bag[j * 64 + lane] : 0ull;
#pragma unroll
        for (int i = KPL; i < NK; ++i) key[i] = 0ull;
        wave_step();
        fill = 0;
        nz_new = 0;
        any = true;
    };

    for (int w0 = 0; w0 < p.ntiles; w0 += BAG_WIN) {
        // this lane's BAG_TPL consecutive tiles of the window: one 8-bit piece of the row's tile mask
        const int t0 = w0 + lane * BAG_TPL;
        unsigned bits = 0u;
        if (t0 < p.ntiles) bits = (tm[t0 >> 5] >> (t0 & 31)) & ((1u << BAG_TPL) - 1u);
        if (__ballot(bits != 0u) == 0ull) continue;             // no candidate of this row in the window
        if (!have_bag) load_bag();                              // in flight beside the count / candidate loads below
        int cnt[BAG_TPL];
        int mine = 0;
#pragma unroll
        for (int i = 0; i < BAG_TPL; ++i) {
            int c = 0;
            if ((bits >> i) & 1u) c = (int)p.cnt2[(long long)(t0 + i) * p.bpad + row];     // t0 + i < ntiles: the bit is set
            cnt[i] = c;
            mine += c;
        }
        int cw;                                                 // candidates of the row in this window
        const int incl = wave_scan_incl(mine, lane, cw);
        if (cw == 0) continue;
        {
            unsigned run = (unsigned)(incl - mine);
#pragma unroll
            for (int i = 0; i < BAG_TPL; ++i) {
                sm.pre[lane * BAG_TPL + i] = run;
                run += (unsigned)cnt[i];
            }
        }
        wave_step();
        for (int j0 = 0; j0 < cw;) {
            // batch position q = s * 64 + lane takes candidate j0 + q - fill of the window (positions < fill are taken)
            const int take_n = (cw - j0) < (BATCH - fill) ? (cw - j0) : (BATCH - fill);
            int pos[BAG_NSLOT];
#pragma unroll
            for (int s2 = 0; s2 < BAG_NSLOT; ++s2) pos[s2] = 0;
            // candidate j -> tile = largest tau with pre[tau] <= j, slot j - pre[tau]
#pragma unroll
            for (int step = BAG_WIN / 2; step > 0; step >>= 1) {
#pragma unroll
                for (int s2 = 0; s2 < BAG_NSLOT; ++s2) {
                    const int q = s2 * 64 + lane - fill;
                    if (q >= 0 && q < take_n && sm.pre[pos[s2] + step] <= (unsigned)(j0 + q)) pos[s2] += step;
                }
            }
            uint2 cd[BAG_NSLOT];
#pragma unroll
            for (int s2 = 0; s2 < BAG_NSLOT; ++s2) {
                const int q = s2 * 64 + lane - fill;
                cd[s2] = make_uint2(0u, 0xFFFFFFFFu);
                if (q >= 0 && q < take_n)
                    cd[s2] = p.cand[row * p.cap + (long long)(w0 + pos[s2]) * 256 + (j0 + q - (int)sm.pre[pos[s2]])];
            }
#pragma unroll
            for (int s2 = 0; s2 < BAG_NSLOT; ++s2) {
                const int q = s2 * 64 + lane - fill;
                if (q >= 0 && q < take_n) key[KPL + s2] = make_key(__uint_as_float(cd[s2].x), cd[s2].y);
                nz_new += __popcll(__ballot(q >= 0 && q < take_n && key[KPL + s2] != 0ull));
            }
            fill += take_n;
            j0 += take_n;
            if (fill == BATCH) fold();
        }
        wave_step();                                            // sm.pre is rewritten by the next window
    }
    if (fill > 0) fold();

    if (any) {
        for (int i = lane; i < p.mask_words; i += 64) tm[i] = 0u;           // consumed: the next stage starts clear
        if (!finalize) {
#pragma unroll
            for (int j = 0; j < KPL; ++j)
                if ((j * 64 + lane) < n_cur) p.topk[row * k + j * 64 + lane] = key[j];
            if (lane == 0) {
                p.topk_n[row] = n_cur;
                // tau only ever rises; under speculation it starts ABOVE the k-th key of the first stages
                if (kth_last && key_value(kth_last) > p.tau[row]) p.tau[row] = key_value(kth_last);
            }
        }
    }
    if (finalize) {
        if (!have_bag) load_bag();
        // speculation check: every key the filter let through is >= the speculative threshold, so the k-th of k such keys
        // bounds everything that was filtered out -- the result is exact.  With fewer than k keys it is not known to be:
        // the row is redone with its safe threshold (rows whose sample held fewer than k real values were never
        // speculated on: tau0 = -inf).
        if (p.fail_cnt && lane == 0) {
            const bool failed = n_cur < k && p.tau0[row] > -INFINITY;
            if (failed || (p.force_mod > 0 && row % p.force_mod == 0)) p.fail_rows[atomicAdd(p.fail_cnt, 1u)] = (int)row;
        }
        const long long orow = p.fb_rows ? (long long)p.fb_rows[p.fb_base + row] : row;
        // the one sort of the row: the bag's keys through the register sorting network (empty slots, 0, go last)
        unsigned long long k4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k4[j] = j < KPL ? key[j < KPL ? j : 0] : 0ull;
        wave_sort256<unsigned long long, true>(k4, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * lane + j;                                     // sorted position
            if (i < k) {
                const bool ok = i < n_cur;
                p.out_val[orow * k + i] = ok ? key_value(k4[j]) : -INFINITY;
                p.out_idx[orow * k + i] = ok ? p.idx_base + (long long)key_row(k4[j]) : -1ll;
            }
        }
    }
}

// Cross-shard merge for k <= 256 and at most 2048 candidates per query (8 shards x k = 200: 1600): ONE wave per query row,
// the row's G x kc candidates in registers (32 key slots per lane), the k-th largest key by the bit-sliced radix select
// of sample_tau_kernel -- on the value word, and on the row word among equal values -- then the k survivors compacted
// through 2 KiB of wave-private LDS, sorted once by the register network and written out.  Replaces the
// workgroup-per-row LDS radix select + bitonic sort (select_kernel<SRC_GATHER>: 41 us / 1.07 TB/s per 2048 rows at G = 8),
// which stays for larger k or more candidates.  Keys are distinct (global rows are); ties in value go to the lower row.
// Two instantiations: 26 slots (<= 1664 candidates: 8 shards x k = 200) at four waves per SIMD, 32 slots at three -- the
// kernel is latency bound, and the 32-slot form does not fit 128 registers without spilling.
constexpr int MERGE_WAVE_SLOTS = 32, MERGE_WAVE_SLOTS_SMALL = 26;
template <int SLOTS, int MIN_WGS>
__global__ __launch_bounds__(256, MIN_WGS) void merge_wave_kernel(const SelectParams p)
{
    static_assert(SLOTS <= 32, "one 32-bit slot mask per lane");
    __shared__ unsigned long long bag_all[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wv;
    if (row >= p.B) return;
    unsigned long long *bag = bag_all[wv];
    const int k = p.k, total = p.G * p.kc;
    unsigned hi[SLOTS], lo[SLOTS], w[32];                                  // w: the bit-sliced working copy (32 x 32 transpose)
    unsigned a = ~0u, o = 0u, act = 0u;
    // candidate j = s2 * 64 + lane is entry c = j % kc of shard g = j / kc: element ((g * B + row) * kc + c) of the
    // [G, B, kc] blocks.  (c, element) are carried from slot to slot -- one division per lane instead of one per slot
    int c = lane % p.kc;
    long long e = ((long long)(lane / p.kc) * p.B + row) * p.kc + c;
    const int dc = 64 % p.kc;
    const long long de = (long long)(64 / p.kc) * p.B * p.kc + dc, wrap = (long long)(p.B - 1) * p.kc;
#pragma unroll
    for (int s2 = 0; s2 < SLOTS; ++s2) {
        const int j = s2 * 64 + lane;
        unsigned long long key = 0ull;
        if (j < total) {
            const long long gi = p.g_idx[e];
            if (gi >= 0) key = make_key(p.g_val[e], (unsigned)gi);          // padding (-1) and NaN: not a candidate
        }
        c += dc;
        e += de;
        if (c >= p.kc) { c -= p.kc; e += wrap; }
        hi[s2] = (unsigned)(key >> 32);
        lo[s2] = (unsigned)key;
        if (key) { a &= hi[s2]; o |= hi[s2]; act |= 0x80000000u >> s2; }      // slot c lives at bit 31 - c
    }
#pragma unroll
    for (int s2 = 0; s2 < 32; ++s2) w[s2] = s2 < SLOTS ? hi[s2 < SLOTS ? s2 : 0] : 0u;
    const int nz = (int)wave_sum_u32((unsigned)__popc(act));
    const int keep = nz < k ? nz : k;
    unsigned long long kth = 1ull;                                          // fewer than k real keys: keep every one
    if (nz >= k) {
        a = wave_and_u32(a);
        o = wave_or_u32(o);
        // rank-th largest word among the slots in `in` (MSB first, over the bits in `var`); leaves the slots that hold it
        auto select = [&](unsigned var, unsigned common, unsigned &in, int &r) {
            unsigned res = common & ~var;
#pragma unroll
            for (int b = 31; b >= 0; --b) {
                if (!(var & (1u << b))) continue;
                const unsigned ones = in & w[31 - b];
                const int cnt1 = (int)wave_sum_u32((unsigned)__popc(ones));
                if (cnt1 >= r) { in = ones; res |= 1u << b; }
                else { r -= cnt1; in &= ~w[31 - b]; }
            }
            return res;
        };
        bit_transpose32(w);
        unsigned in = act;
        int r = k;
        const unsigned kth_hi = select(a ^ o, a, in, r);                     // by value ...
        // ... then, among the slots that hold exactly that value, by the row word (0xFFFFFFFF - row: lower row = larger)
        unsigned al = ~0u, ol = 0u;
#pragma unroll
        for (int s2 = 0; s2 < 32; ++s2) {
            w[s2] = s2 < SLOTS ? lo[s2 < SLOTS ? s2 : 0] : 0u;
            if (s2 < SLOTS && (in & (0x80000000u >> s2))) { al &= w[s2]; ol |= w[s2]; }
        }
        al = wave_and_u32(al);
        ol = wave_or_u32(ol);
        unsigned kth_lo = al;                                               // one slot left: its row word
        if (al != ol) {                                                     // several rows share the k-th value
            bit_transpose32(w);
            kth_lo = select(al ^ ol, al, in, r);
        }
        kth = ((unsigned long long)kth_hi << 32) | kth_lo;
    }
    // compact the survivors (exactly `keep`: keys are distinct), sort them, write them out
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int s2 = 0; s2 < SLOTS; ++s2) {
        if (s2 * 64 < total) {                                              // slots beyond the candidates hold nothing (uniform)
            const unsigned long long key = ((unsigned long long)hi[s2] << 32) | lo[s2];
            const bool take = key >= kth && key != 0ull;
            const unsigned long long bal = __ballot(take);
            const int at = base + __popcll(bal & ltmask);
            if (take && at < 256) bag[at] = key;
            base += __popcll(bal);
        }
    }
    wave_step();
    unsigned long long k4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k4[j] = (j * 64 + lane) < keep ? bag[j * 64 + lane] : 0ull;
    wave_sort256<unsigned long long, true>(k4, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 4 * lane + j;
        if (i < k) {
            const bool ok = i < keep;
            p.out_val[row * k + i] = ok ? key_value(k4[j]) : -INFINITY;
            p.out_idx[row * k + i] = ok ? (long long)key_row(k4[j]) : -1ll;
        }
    }
}

// k-th largest value of a 1-D float bank (track_semi/chmatch.py:252,294: `bank.topk(k)[0][-1]`, k up to 0.8 n):
// MSB-first radix select over the order-preserving key of every element, 4 passes of 8 bits, one workgroup
// (the bank is 50 000 floats: 200 KB, L2 resident; the kernel is latency-, not bandwidth-bound).  Selection
// only: the result is one of the inputs, bit for bit.  NaNs order above +inf, as torch.topk does.
constexpr int KTH_THREADS = 1024;
__global__ __launch_bounds__(KTH_THREADS) void kth_largest_kernel(const float *__restrict__ x, long long n, long long k,
                                                                 float *__restrict__ out)
{
    __shared__ unsigned hist[KTH_THREADS / 64][256];          // one histogram per wave: fewer LDS atomic collisions
    __shared__ unsigned s_prefix;
    __shared__ long long s_rank;
    const int tid = threadIdx.x, wv = tid >> 6;
    if (tid == 0) { s_prefix = 0u; s_rank = k; }
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < (KTH_THREADS / 64) * 256; i += KTH_THREADS) (&hist[0][0])[i] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (long long i = tid; i < n; i += KTH_THREADS) {
            const float v = x[i];
            const unsigned key = (v != v) ? 0xFFFFFFFFu : orderable_f32(v);
            if ((key & mask) == prefix) atomicAdd(&hist[wv][(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 256) {                                       // fold the per-wave histograms
            unsigned c = 0u;
            for (int w2 = 0; w2 < KTH_THREADS / 64; ++w2) c += hist[w2][tid];
            hist[0][tid] = c;
        }
        __syncthreads();
        if (tid == 0) {
            long long rank = s_rank, cum = 0;
            int b = 255;
            for (; b > 0; --b) {
                if (cum + hist[0][b] >= rank) break;
                cum += hist[0][b];
            }
            s_rank = rank - cum;
            s_prefix = prefix | ((unsigned)b << shift);
        }
        mask |= 0xFFu << shift;
        __syncthreads();
    }
    if (tid == 0) out[0] = s_prefix == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : from_orderable(s_prefix);
}

// The same selection for a bank of at most 65 536 values (the CHMatch queues hold 50 000) with the bank IN REGISTERS: one
// workgroup of 1024 threads, 64 consecutive values per thread, read once, then an MSB-first binary radix select on bit-sliced
// keys as sample_tau_kernel's -- two 32 x 32 bit transposes per thread, after which a radix step is two v_and + two v_bcnt,
// a wave sum, one LDS word per wave and ONE workgroup barrier -- only over the bits in which the keys differ at all.  No
// LDS atomics (the 8-bit histogram of kth_largest_kernel serialises on them when the values share their high bits, as
// confidences in [0, 1] do: 68-75 us for 50 000 values), no second pass over memory.
constexpr int KTH_REG_MAX_N = 64 * KTH_THREADS;
__global__ __launch_bounds__(KTH_THREADS) void kth_largest_reg_kernel(const float *__restrict__ x, int n, int k, float *__restrict__ out)
{
    __shared__ unsigned s_part[2][KTH_THREADS / 64];
    __shared__ unsigned s_ao[2][KTH_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned w0[32], w1[32];                                   // slots 0..31 and 32..63: element tid * 64 + slot
    unsigned in0 = 0u, in1 = 0u, a = ~0u, o = 0u;
    // A thread owns 64 CONSECUTIVE values (256 bytes: sixteen 16-byte loads off one base address with immediate offsets --
    // a slot-major mapping needs 64 separate 64-bit addresses and spills).  One half at a time (load, convert, transpose).
    const int base = tid * 64;
#define MI_KTH_LOAD_HALF(WW, IN, C0)                                                                                  \
    _Pragma("unroll") for (int q = 0; q < 8; ++q) {                                                                 \
        const int i0 = base + (C0) + q * 4;                                                                         \
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
        if (i0 + 3 < n) v = *(const float4 *)(x + i0);                                                              \
        else {                                                      /* the one thread that holds the end of the bank */ \
            if (i0 < n) v.x = x[i0];                                                                                \
            if (i0 + 1 < n) v.y = x[i0 + 1];                                                                        \
            if (i0 + 2 < n) v.z = x[i0 + 2];                                                                        \
        }                                                                                                           \
        const float f[4] = {v.x, v.y, v.z, v.w};                                                                    \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                             \
            unsigned key = 0u;                                                                                      \
            if (i0 + u < n) {                                                                                       \
                key = (f[u] != f[u]) ? 0xFFFFFFFFu : orderable_f32(f[u]);   /* NaNs order above +inf, as torch.topk does */ \
                a &= key;                                                                                           \
                o |= key;                                                                                           \
                IN |= 0x80000000u >> (q * 4 + u);                   /* slot c lives at bit 31 - c */                \
            }                                                                                                       \
            WW[q * 4 + u] = key;                                                                                     \
        }                                                                                                           \
    }                                                                                                               \
    bit_transpose32(WW);                                            /* w[31 - b] = bit b of every slot */
    MI_KTH_LOAD_HALF(w0, in0, 0)
    // the first half is DONE here (opaque copies): left to itself the compiler sinks both transposes behind the reduction
    // below and interleaves them, which needs more than the 128 registers of a 1024-thread workgroup
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(w0[i]));
    MI_KTH_LOAD_HALF(w1, in1, 32)
#pragma unroll
    for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(w1[i]));
#undef MI_KTH_LOAD_HALF
    // the bits in which the keys differ at all
    a = wave_and_u32(a);
    o = wave_or_u32(o);
    if (lane == 0) { s_ao[0][wv] = a; s_ao[1][wv] = o; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KTH_THREADS / 64; ++i) { a &= s_ao[0][i]; o |= s_ao[1][i]; }
    const unsigned var = (unsigned)__builtin_amdgcn_readfirstlane((int)(a ^ o));
    unsigned res = (unsigned)__builtin_amdgcn_readfirstlane((int)a) & ~var;
    int r = k, step = 0;
#pragma unroll
    for (int b = 31; b >= 0; --b) {
        if (!(var & (1u << b))) continue;                      // workgroup-uniform: scalar branch
        const unsigned ones0 = in0 & w0[31 - b], ones1 = in1 & w1[31 - b];
        const unsigned c = wave_sum_u32((unsigned)(__popc(ones0) + __popc(ones1)));
        unsigned *part = s_part[step & 1];                     // two buffers, one barrier per step: a wave two steps ahead has passed the barrier every reader of this buffer arrives at first
        ++step;
        if (lane == 0) part[wv] = c;
        __syncthreads();
        int cnt1 = 0;
#pragma unroll
        for (int i = 0; i < KTH_THREADS / 64; ++i) cnt1 += (int)part[i];
        cnt1 = __builtin_amdgcn_readfirstlane(cnt1);
        if (cnt1 >= r) {                                       // the r-th largest has the bit set: the zeros drop out
            in0 = ones0;
            in1 = ones1;
            res |= 1u << b;
        } else {                                               // it has the bit clear: the ones drop out
            r -= cnt1;
            in0 ^= ones0;                                      // = in & ~w, without a second copy of w (the compiler kept 64 of them, and spilled)
            in1 ^= ones1;
        }
    }
    if (tid == 0) out[0] = res == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : from_orderable(res);
}


}  // namespace mi
