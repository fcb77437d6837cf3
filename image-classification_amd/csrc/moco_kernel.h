// moco_kernel.h -- "attend over a feature bank" kernels: exp-weighted sums of bank rows per query that never
// write the [b, K] weight matrix.  Two users:
//
//  * MoCo queue InfoNCE, backward w.r.t. the query (track_ssl/moco.py:152-156,172):
//        p[i][n] = exp(q_i . queue_n / T - lse_i)                     (softmax over [pos, queue])
//        dq_i    = g / (b T) * ( sum_n p[i][n] queue_n + (p_pos[i] - 1) key_i )
//    The queue may have been overwritten between forward and backward (MemoryBank.push precedes backward at
//    moco.py:159-172).  Instead of the reference's full clone() per step the caller stashes only the
//    overwritten rows: rows [patch_lo, patch_lo + patch_n) are read from `patch`.
//
//  * CoMatch soft-kNN pseudo-label smoothing (track_semi/comatch.py:187-189):
//        A[i][n] = exp(f_i . bank_n / T) ;  probs_i = alpha p_i + (1 - alpha) sum_n A[i][n] prob_n / sum_n A[i][n]
//    (plain exp, no max subtraction -- as the reference; the row sum rides as one extra value column of ones).
//
// A workgroup owns 64 query rows and one slice of the bank, recomputes S = q . tile^T for 64 bank rows at a
// time with fp32 MFMA, turns it into weights in registers, transposes them through a wave-private LDS patch
// and accumulates weights . values into its [64, V] partial (fp32 MFMA again).  Slices are summed by a second
// kernel in slice order, so the result does not depend on scheduling.
//
// MFMA operand maps (v_mfma_f32_16x16x4_f32): A lane -> (row = lane & 15, k = lane >> 4),
// B lane -> (col = lane & 15, k = lane >> 4), C/D lane -> (col = lane & 15, row = 4 * (lane >> 4) + reg).
// Both GEMMs feed A and B with the SAME permutation of k (a lane's float4 holds 4 k values used
// by 4 consecutive MFMAs), which keeps every LDS read a conflict-free 128-bit access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

constexpr int MB_ROWS = 64;        // query rows and bank rows per tile
constexpr int MB_THREADS = 256;    // 4 waves, wave w owns query rows 16w .. 16w+15
constexpr int MB_PLD = 68;         // leading dimension of the wave-private weight patch

struct AttendParams {
    const float *q;            // [b, D]
    const float *bank;         // [K, D]
    const float *lse;          // [b] (MoCo) or nullptr (soft-kNN: weights are exp(S / T))
    const float *patch;        // [patch_n, D] pre-push contents of bank rows [patch_lo, patch_lo + patch_n), or nullptr
    int patch_lo, patch_n;
    const float *vals;         // soft-kNN: [K, C] values; column C of the LDS image is 1 (row sum)
    int C;
    int b, K;
    float inv_t;
    float *part;               // [nslices, b, V]   V = D (MoCo) or V64 * 64 (soft-kNN)
    int rows_per_slice;        // multiple of MB_ROWS
};

inline size_t attend_lds_bytes(int D, int V, bool separate_vals)
{
    // bank tile + (soft-kNN) value tile + one weight patch per wave; the query tile lives in registers
    return (size_t)(MB_ROWS * (D + 4) + (separate_vals ? MB_ROWS * (V + 4) : 0) + 4 * 16 * MB_PLD) * 4;
}

template <int D64, int V64, bool SOFT>      // D / 64, V / 64 (values); !SOFT: the values ARE the bank rows (V64 == D64)
__global__ __launch_bounds__(MB_THREADS) void attend_kernel(const AttendParams p)
{
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    constexpr int D = D64 * 64, LD = D + 4;
    constexpr int V = V64 * 64, LV = V + 4;
    constexpr int LOADS = MB_ROWS * D / 4 / MB_THREADS;      // float4 per thread and feature tile = D / 16
    constexpr int VLOADS = SOFT ? MB_ROWS * V / MB_THREADS : 1;   // floats per thread and value tile = V / 4
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *sx = lds;                                  // [64][LD] bank tile
    float *sv = SOFT ? sx + MB_ROWS * LD : sx;        // [64][LV] value tile (soft-kNN) / the bank tile itself
    float *sp = sv + (SOFT ? MB_ROWS * LV : MB_ROWS * LD);   // [4][16][MB_PLD] weights, one patch per wave
    constexpr int SLD = SOFT ? LV : LD;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int q0 = blockIdx.x * MB_ROWS;
    const int n_begin = blockIdx.y * p.rows_per_slice;
    const int n_end = (n_begin + p.rows_per_slice) < p.K ? (n_begin + p.rows_per_slice) : p.K;

    auto tile_row_ptr = [&](int n) -> const float * {
        if (n >= p.patch_lo && n < p.patch_lo + p.patch_n) return p.patch + (long long)(n - p.patch_lo) * D;
        return p.bank + (long long)n * D;
    };
    float4 pre[LOADS];
    float vpre[VLOADS];
    auto prefetch = [&](int n0) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int e = tid + i * MB_THREADS, r = e / (D / 4), c = e % (D / 4);
            const int n = n0 + r;
            pre[i] = n < n_end ? *(const float4 *)(tile_row_ptr(n) + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (SOFT) {
            // value tile: element e -> (row e / V, column e % V); C is arbitrary, so scalar loads; column C = 1
#pragma unroll
            for (int i = 0; i < VLOADS; ++i) {
                const int e = tid + i * MB_THREADS, r = e / V, c = e % V;
                const int n = n0 + r;
                float v = 0.f;
                if (n < n_end) v = c < p.C ? p.vals[(long long)n * p.C + c] : (c == p.C ? 1.f : 0.f);
                vpre[i] = v;
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int e = tid + i * MB_THREADS, r = e / (D / 4), c = e % (D / 4);
            *(float4 *)(sx + r * LD + c * 4) = pre[i];
        }
        if constexpr (SOFT) {
#pragma unroll
            for (int i = 0; i < VLOADS; ++i) {
                const int e = tid + i * MB_THREADS, r = e / V, c = e % V;
                sv[r * LV + c] = vpre[i];
            }
        }
    };

    // query operand -> registers, once: lane (lr, lq) of wave w holds row 16w + lr, columns 16g + 4lq .. + 3 of every
    // 16-column group g (the A operand of the first GEMM; D / 4 registers).  Rows past b are zero (MoCo gives them
    // lse = +inf below: their weight is exactly 0; soft-kNN rows past b are simply not written out).  Keeping it out of
    // LDS (33 KiB at D = 128) lets three workgroups share a CU instead of one.
    float4 aq[D / 16];
    {
        const int row = q0 + 16 * w + lr;
#pragma unroll
        for (int g = 0; g < D / 16; ++g)
            aq[g] = row < p.b ? *(const float4 *)(p.q + (long long)row * D + 16 * g + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float lse_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q0 + 16 * w + 4 * lq + r;
        lse_r[r] = SOFT ? 0.f : (row < p.b ? p.lse[row] : INFINITY);
    }
    f32x4 acc[V64 * 4];
#pragma unroll
    for (int i = 0; i < V64 * 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    float *spw = sp + w * 16 * MB_PLD;
    if (n_begin < n_end) prefetch(n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += MB_ROWS) {
        __syncthreads();                       // everyone is done reading the previous tile
        commit();
        __syncthreads();
        if (n0 + MB_ROWS < n_end) prefetch(n0 + MB_ROWS);     // in flight during both GEMMs

        // S[16 x 64] = q rows (16w ..) . tile rows^T.  The bank operand of column group g + 1 is read while the MFMAs of
        // group g run (two register sets, one read per four MFMAs, pinned: hipcc otherwise sinks every read to just
        // before its first use and the wave waits out the LDS latency 8 times per tile).  The four accumulators take
        // turns: a v_mfma_f32_16x16x4_f32 issues every 32 cycles but its result is ready after 40.
        f32x4 s[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) s[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 x0[4], x1[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) x0[nb] = *(const float4 *)(sx + (16 * nb + lr) * LD + 4 * lq);
#pragma unroll
        for (int g = 0; g < D / 16; ++g) {
            const float4 a = aq[g];
            float4 *xc = (g & 1) ? x1 : x0, *xn = (g & 1) ? x0 : x1;
            if (g + 1 < D / 16) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) xn[nb] = *(const float4 *)(sx + (16 * nb + lr) * LD + 16 * (g + 1) + 4 * lq);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) s[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, xc[nb].x, s[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) s[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, xc[nb].y, s[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) s[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, xc[nb].z, s[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) s[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, xc[nb].w, s[nb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int g = 0; g + 1 < D / 16; ++g) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        // weights = exp(S / T - lse), columns past the slice end are 0; C layout -> LDS patch [row][n]
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            const bool ok = (n0 + 16 * nb + lr) < n_end;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float wgt;
                if constexpr (SOFT) wgt = expf(s[nb][r] * p.inv_t);
                else wgt = __expf(s[nb][r] * p.inv_t - lse_r[r]);
                spw[(4 * lq + r) * MB_PLD + 16 * nb + lr] = ok ? wgt : 0.f;
            }
        }
        // the patch is wave private and a wave's LDS operations complete in order: no workgroup barrier between its
        // stores above and its loads below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // out[16 x V] += W[16 x 64] . values[64 x V]; block c*4+j holds columns 64c + 4*(lane&15) + j.  Operands of
        // step kk + 1 are read under the MFMAs of step kk (pinned as above).
        float wa[2];
        float4 v0[V64], v1[V64];
        wa[0] = spw[lr * MB_PLD + lq];
#pragma unroll
        for (int c = 0; c < V64; ++c) v0[c] = *(const float4 *)(sv + lq * SLD + 64 * c + 4 * lr);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float a = wa[kk & 1];
            float4 *vc = (kk & 1) ? v1 : v0, *vn = (kk & 1) ? v0 : v1;
            if (kk < 15) {
                wa[(kk + 1) & 1] = spw[lr * MB_PLD + 4 * (kk + 1) + lq];
#pragma unroll
                for (int c = 0; c < V64; ++c) vn[c] = *(const float4 *)(sv + (4 * (kk + 1) + lq) * SLD + 64 * c + 4 * lr);
            }
#pragma unroll
            for (int c = 0; c < V64; ++c) {
                acc[c * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vc[c].x, acc[c * 4 + 0], 0, 0, 0);
                acc[c * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vc[c].y, acc[c * 4 + 1], 0, 0, 0);
                acc[c * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vc[c].z, acc[c * 4 + 2], 0, 0, 0);
                acc[c * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vc[c].w, acc[c * 4 + 3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + V64, 0);
#pragma unroll
        for (int kk = 0; kk < 15; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
            for (int c = 0; c < V64; ++c) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (c + 1 < V64) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * V64, 0);
    }
    // partial of this slice
    float *part = p.part + (long long)blockIdx.y * p.b * V;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q0 + 16 * w + 4 * lq + r;
        if (row < p.b) {
#pragma unroll
            for (int c = 0; c < V64; ++c)
                *(float4 *)(part + (long long)row * V + 64 * c + 4 * lr) =
                    make_float4(acc[c * 4 + 0][r], acc[c * 4 + 1][r], acc[c * 4 + 2][r], acc[c * 4 + 3][r]);
        }
    }
}

// dq[i][d]   = g * inv_t / b * ( sum_s part[s][i][d] + (p_pos_i - 1) * key[i][d] )      p_pos_i = exp(lpos_i * inv_t - lse_i)
// dkey[i][d] = g * inv_t / b * (p_pos_i - 1) * q[i][d]            (the positive logit q_i . key_i is the only term with key)
// Either output may be null (dq == nullptr: `part` is not read).
// 256 threads = 16 float4 columns x 16 slice groups: group g adds slices g, g + 16, ... (all its loads in flight
// together), the 16 group sums meet in LDS and are added in group order -- a fixed order, so the result does not depend
// on timing.  (One thread per element walking all slices read the partials at 0.5 TB/s.)
constexpr int RED_GROUPS = 16;
__global__ __launch_bounds__(256) void moco_bwd_reduce_kernel(const float *part, int nslices, int b, int D,
                                                              const float *q, const float *key, const float *lse,
                                                              const float *lpos, const float *grad_loss, float inv_t,
                                                              float *dq, float *dkey)
{
    __shared__ float4 sm[RED_GROUPS][16];
    const long long n4 = (long long)b * D / 4;                 // D % 64 == 0
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long long e4 = (long long)blockIdx.x * 16 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (dq && e4 < n4) {
        const float4 *p4 = (const float4 *)part + e4;
#pragma unroll 8
        for (int sl = grp; sl < nslices; sl += RED_GROUPS) {
            const float4 v = p4[(long long)sl * n4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sm[grp][col] = s;
    __syncthreads();
    if (grp != 0 || e4 >= n4) return;
#pragma unroll
    for (int g = 1; g < RED_GROUPS; ++g) {
        const float4 v = sm[g][col];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long long e = e4 * 4;
    const int row = (int)(e / D);
    const float ppos = __expf(lpos[row] * inv_t - lse[row]);
    const float scale = grad_loss[0] * inv_t / (float)b;
    if (dq) {
        const float4 kv = *(const float4 *)(key + e);
        *(float4 *)(dq + e) = make_float4(scale * (s.x + (ppos - 1.f) * kv.x), scale * (s.y + (ppos - 1.f) * kv.y),
                                          scale * (s.z + (ppos - 1.f) * kv.z), scale * (s.w + (ppos - 1.f) * kv.w));
    }
    if (dkey) {
        const float4 qv = *(const float4 *)(q + e);
        *(float4 *)(dkey + e) = make_float4(scale * (ppos - 1.f) * qv.x, scale * (ppos - 1.f) * qv.y,
                                            scale * (ppos - 1.f) * qv.z, scale * (ppos - 1.f) * qv.w);
    }
}

// probs_out[i][c] = alpha * probs_in[i][c] + (1 - alpha) * (sum_s part[s][i][c]) / (sum_s part[s][i][C])
// One workgroup per row: V / 4 float4 columns x (1024 / V) slice groups; group g adds slices g, g + groups, ... with all its
// loads in flight, the group sums meet in LDS and are added in group order (fixed order: deterministic).
__global__ __launch_bounds__(256) void soft_knn_reduce_kernel(const float *part, int nslices, int b, int V, int C,
                                                              const float *probs_in, float alpha, float *probs_out)
{
    __shared__ float4 sm[256];
    __shared__ float rowsum[256];
    const int row = blockIdx.x;
    const int cols4 = V >> 2;                                  // 16, 32, 48 or 64
    const int groups = 256 / cols4;                            // 16, 8, 5 or 4 (threads past groups * cols4 idle)
    const int col = threadIdx.x % cols4, grp = threadIdx.x / cols4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (grp < groups) {
        const float4 *p4 = (const float4 *)(part + (long long)row * V) + col;
        const long long stride4 = (long long)b * V / 4;
#pragma unroll 4
        for (int sl = grp; sl < nslices; sl += groups) {
            const float4 v = p4[(long long)sl * stride4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0) {
        for (int g = 1; g < groups; ++g) {
            const float4 v = sm[g * cols4 + col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        rowsum[4 * col + 0] = s.x; rowsum[4 * col + 1] = s.y; rowsum[4 * col + 2] = s.z; rowsum[4 * col + 3] = s.w;
    }
    __syncthreads();
    const float den = rowsum[C];
    for (int c = threadIdx.x; c < C; c += 256) {
        const long long e = (long long)row * C + c;
        probs_out[e] = alpha * probs_in[e] + (1.f - alpha) * (rowsum[c] / den);
    }
}

// Ring-buffer enqueue that first saves the rows it overwrites (16-byte units).
__global__ void push_stash_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint4 *__restrict__ stash, long long n16)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        stash[i] = dst[i];
        dst[i] = src[i];
    }
}
__global__ void push_stash_bytes_kernel(char *__restrict__ dst, const char *__restrict__ src, char *__restrict__ stash, long long n)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        stash[i] = dst[i];
        dst[i] = src[i];
    }
}

}  // namespace mi
