// sim_kernel.h -- fused similarity GEMM tile + streaming top-k filter for gfx950 (CDNA4).
//
// One workgroup (512 threads = 8 waves, one per CU) computes a 256 x 256 tile of
//     sim[b][n] = sum_k q[b][k] * bank[n][k]                       (models/knn.py:6)
// with MFMA and never writes it to HBM.  What leaves the tile depends on MODE:
//   MODE_DENSE  : bootstrap stage -- the tile is stored densely (values only) into the per-row
//                 candidate buffer, the select kernel turns it into the first top-k + threshold;
//   MODE_FILTER : every element is compared with its row's threshold tau[row] (a lower bound of
//                 the row's final k-th largest similarity); survivors (rare) are appended to the
//                 row's candidate list in HBM through one atomic slot counter per row;
//   MODE_LSE    : MoCo logits -- per-row (max, sum exp) partials of sim / T over the tile's columns.
//
// Data layout: both operands are K-contiguous rows ([rows, ld] elements, a row = whole 128-byte
// K-steps: 64 bf16 or 32 f32).  Per K-step each operand tile is 256 rows x 128 B = 32 KiB, brought
// in by LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows per wave-instruction), double buffered
// (2 x 64 KiB).  LDS image: row r at r*128, its 16-byte chunk c stored at chunk slot
// c ^ ((r >> 1) & 7): the DMA destination stays lane-linear and the swizzle is applied to the
// per-lane SOURCE address and to the MFMA operand reads (conflict-free ds_read_b128 for the
// 16x16x32 bf16 operand map; 2-way for the f32 ds_read_b32).
//
// Wave tiling: 2 (M) x 4 (N) waves, 128 x 64 outputs per wave = 8 x 4 MFMA blocks of 16 x 16,
// 128 accumulator registers per lane.  bf16: v_mfma_f32_16x16x32_bf16; f32: v_mfma_f32_16x16x4_f32,
// which accumulates each output as one k-ascending fmaf chain (bit-exact vs oracle/knn_chain.c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mi {

constexpr int DT_F32 = 0, DT_BF16 = 1;
constexpr int MODE_DENSE = 0, MODE_FILTER = 1, MODE_LSE = 2;

constexpr int TILE = 256;                 // block tile edge (queries x bank rows)
constexpr int KSTEP_BYTES = 128;          // bytes of one row per K-step
constexpr int OPERAND_TILE_BYTES = TILE * KSTEP_BYTES;       // 32 KiB
constexpr int STAGE_BYTES = 2 * OPERAND_TILE_BYTES;          // A + B
constexpr int SIM_LDS_BYTES = 2 * STAGE_BYTES;               // double buffered: 128 KiB
constexpr int SIM_THREADS = 512;
constexpr int PATCH_Q = 8, PATCH_N = 4;   // tiles of one XCD patch (8 query tiles x 4 bank tiles)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct SimParams {
    const char *q;            // [B, ldq] packed queries
    const char *x;            // [N, ldx] packed bank rows
    long long ldq_bytes, ldx_bytes;
    int B, N;                 // valid rows of q / x
    int ksteps;               // padded row bytes / 128
    int col0, ncols;          // this stage covers bank rows [col0, col0 + ncols)
    int qtiles, ntiles;       // tiles in this launch
    int pq, pn;               // patches (groups of 8 x 4 tiles)
    // MODE_DENSE / MODE_FILTER
    const float *tau;         // [B_pad]  (+inf on padding rows)
    unsigned *cnt;            // [B_pad]  per-row candidate count
    uint2 *cand;              // [B_pad, cap] (value bits, bank row); dense mode: float[B_pad][2*cap]
    int cap;
    // MODE_LSE
    float inv_t;
    float2 *lse_part;         // [B_pad, ntiles * 4] (max, sumexp) per wave column strip
    int lse_ld;
};

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))

template <int DT, int MODE>
__global__ __launch_bounds__(SIM_THREADS, 2) void sim_kernel(const SimParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[SIM_LDS_BYTES];

    // ---- block -> tile.  Blocks b and b+8 share an XCD (round-robin dispatch), an XCD holds 32
    // resident blocks (one per CU: 128 KiB LDS), so 32 consecutive same-XCD blocks form one patch
    // of 8 query tiles x 4 bank tiles that share operands through that XCD's L2.  Speed only.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, seq = bid >> 3;
    const int patch = (seq >> 5) * 8 + xcd, inner = seq & 31;
    if (patch >= p.pq * p.pn) return;
    const int qt = (patch / p.pn) * PATCH_Q + (inner & 7);
    const int nt = (patch % p.pn) * PATCH_N + (inner >> 3);
    if (qt >= p.qtiles || nt >= p.ntiles) return;
    const int q0 = qt * TILE;
    const int n0 = p.col0 + nt * TILE;
    const int col_end = p.col0 + p.ncols;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;

    // ---- LDS-DMA source addresses.  Wave w stages row groups g = 4w .. 4w+3 (8 rows each) of both
    // operand tiles; lane -> (row r = 8g + lane/8, LDS chunk slot lane%8), source chunk = slot ^ swz(r).
    const char *a_src[4];
    const char *b_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (w * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int qa = q0 + r; qa = qa < p.B ? qa : p.B - 1;             // clamp: padding rows re-read a valid row
        int xb = n0 + r; xb = xb < p.N ? xb : p.N - 1;
        a_src[i] = p.q + (long long)qa * p.ldq_bytes + chunk * 16;
        b_src[i] = p.x + (long long)xb * p.ldx_bytes + chunk * 16;
    }
    auto stage = [&](int t, int buf) {
        char *base = smem + buf * STAGE_BYTES + w * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((const AS1 void *)(a_src[i] + (long long)t * KSTEP_BYTES),
                                             (AS3 void *)(base + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const AS1 void *)(b_src[i] + (long long)t * KSTEP_BYTES),
                                             (AS3 void *)(base + OPERAND_TILE_BYTES + i * 1024), 16, 0, 0);
        }
    };

    // ---- MFMA operand read offsets.  Lane reads row (lane & 15) of a 16-row block; K chunk
    // (lane >> 4) [+4 for the second half of the K-step]; swizzle term depends on the lane only.
    const int lr = lane & 15, lq = lane >> 4;
    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wm * 128 + lr) * KSTEP_BYTES;
    const int b_row_off = OPERAND_TILE_BYTES + (wn * 64 + lr) * KSTEP_BYTES;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    __syncthreads();
    for (int t = 0; t < p.ksteps; ++t) {
        if (t + 1 < p.ksteps) stage(t + 1, (t + 1) & 1);
        const char *lds = smem + (t & 1) * STAGE_BYTES;
        if constexpr (DT == DT_BF16) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int coff = ((lq ^ sw) ^ (s << 2)) * 16;
                bf16x8 af[8], bf[4];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    af[i] = *(const bf16x8 *)(lds + a_row_off + i * 16 * KSTEP_BYTES + coff);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bf[j] = *(const bf16x8 *)(lds + b_row_off + j * 16 * KSTEP_BYTES + coff);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int coff = ((kk ^ sw) * 16) + lq * 4;
                float af[8], bf[4];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    af[i] = *(const float *)(lds + a_row_off + i * 16 * KSTEP_BYTES + coff);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    bf[j] = *(const float *)(lds + b_row_off + j * 16 * KSTEP_BYTES + coff);
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D map of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg.
    const int row_base = q0 + wm * 128 + lq * 4;          // + i*16 + reg
    const int col_base = n0 + wn * 64 + lr;               // + j*16
    if constexpr (MODE == MODE_DENSE) {
        float *dense = (float *)p.cand;
        const long long ld = 2ll * p.cap;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_base + i * 16 + r;
                if (row < p.B) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int col = col_base + j * 16;
                        if (col < col_end) dense[row * ld + (col - p.col0)] = acc[i][j][r];
                    }
                }
            }
    } else if constexpr (MODE == MODE_FILTER) {
        // tau is padded to a multiple of 256 rows (+inf on padding): no row bound needed here.
        f32x4 tau[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tau[i] = *(const f32x4 *)(p.tau + row_base + i * 16);
        bool colok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) colok[j] = (col_base + j * 16) < col_end;
        bool any = false;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) any |= (colok[j] && acc[i][j][r] >= tau[i][r]);
        if (__any(any)) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[i][j][r];
                        const int row = row_base + i * 16 + r;
                        if (colok[j] && v >= tau[i][r] && row < p.B) {
                            const unsigned slot = atomicAdd(p.cnt + row, 1u);
                            if (slot < (unsigned)p.cap)
                                p.cand[(long long)row * p.cap + slot] =
                                    make_uint2(__float_as_uint(v), (unsigned)(col_base + j * 16));
                        }
                    }
        }
    } else {   // MODE_LSE: per-row (max, sum exp) of sim * inv_t over this wave's 64 columns
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col_base + j * 16 < col_end) m = fmaxf(m, acc[i][j][r] * p.inv_t);
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col_base + j * 16 < col_end) s += expf(acc[i][j][r] * p.inv_t - m);
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) s += __shfl_xor(s, d, 64);
                const int row = row_base + i * 16 + r;
                if (lr == 0 && row < p.B)
                    p.lse_part[(long long)row * p.lse_ld + nt * 4 + wn] = make_float2(m, s);
            }
    }
}

}  // namespace mi
