// sim_kernel.h -- fused similarity GEMM tile + streaming top-k filter for gfx950 (CDNA4).
//
// One workgroup (512 threads = 8 waves, one per CU) computes a 256 x 256 tile of
//     sim[b][n] = sum_k q[b][k] * bank[n][k]                       (models/knn.py:6)
// with MFMA and never writes it to HBM.  What leaves the tile depends on MODE:
//   MODE_DENSE  : bootstrap stage -- the tile is stored densely (values only) into the per-row
//                 candidate buffer, the select kernel turns it into the first threshold (a strided
//                 sample of the bank: dense column c = bank row c * row_stride) or, for a bank no
//                 larger than the bootstrap, into the final top-k;
//   MODE_FILTER : every element is compared with its row's threshold tau[row] (a lower bound of
//                 the row's final k-th largest similarity); survivors (rare) go straight to the
//                 (row, bank tile) segment of the row's candidate buffer in HBM -- slot from an LDS
//                 counter, per-tile counts and a per-row tile bitmask tell the refresh kernels
//                 where to look, no returning global atomic;
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

#include <type_traits>

namespace mi {

constexpr int DT_F32 = 0, DT_BF16 = 1;
constexpr int MODE_DENSE = 0, MODE_FILTER = 1, MODE_LSE = 2;

constexpr int TILE = 256;                 // block tile edge (queries x bank rows)
constexpr int KSTEP_BYTES = 128;          // bytes of one row per K-step
constexpr int OPERAND_TILE_BYTES = TILE * KSTEP_BYTES;       // 32 KiB
constexpr int STAGE_BYTES = 2 * OPERAND_TILE_BYTES;          // A + B
constexpr int SIM_LDS_BYTES = 2 * STAGE_BYTES;               // double buffered: 128 KiB
constexpr int SIM_EXTRA_LDS = 2048 + 256;                   // thresholds + candidate counters of the tile's 256 rows, prefetch sink
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
    int col0, ncols;          // this launch covers bank rows [col0, col0 + ncols)
    int nt_base;              // MODE_FILTER: stage tile of the launch's tile 0 (a stage may take several launches)
    int row_stride;           // MODE_DENSE: dense column c is bank row c * row_stride (strided sample of the bank)
    int qtiles, ntiles;       // tiles in this launch
    int pq, pn, qsh;          // patches of (1 << qsh) query tiles x (32 >> qsh) bank tiles (set by the launcher)
    // MODE_DENSE / MODE_FILTER
    const float *tau;         // [B_pad]  (+inf on padding rows)
    unsigned short *cnt2;     // MODE_FILTER: [stage tiles, B_pad] candidates of (bank tile, query row)
    unsigned *tmask;          // MODE_FILTER: [B_pad, mask_words] bit t set: bank tile t holds candidates of the row
    int bpad, mask_words;
    uint2 *cand;              // [B_pad, cap] (value bits, bank row): bank tile t of the stage owns slots
                              // [256 t, 256 t + 256) of every row; dense mode: float[B_pad][2*cap]
    int cap;
    // MODE_LSE
    float inv_t;
    float2 *lse_part;         // [ntiles, lse_ld] (max, sumexp) of every row over one tile; lse_ld = padded rows
    int lse_ld;
    unsigned *zero_word;      // MODE_LSE: a word the NEXT kernel on the stream expects to be 0 (its arrival counter)
    // sim_walk_kernel.h only: workgroup v of qtiles * walk_nsplit walks bank tiles [split * walk_tps, (split + 1) * walk_tps)
    int walk_nsplit, walk_tps;
    // ... and, when set, hands over the survivors of a row's WHOLE RUN as one list: slots counted on from the run's first
    // tile's segment (the segments of a run are contiguous), one cnt2 entry (<= 255 tiles x 256 fits its 16 bits) and one
    // mask bit per (row, run).  Only for the consumer that takes a list of any length: bag_refresh_kernel (k <= 256).
    int walk_compact;
    // FB instantiation only (the exact fallback of a speculative first threshold, mi_knn.hip): the launch's query rows
    // are the COMPACT rows c = 0 .. n - 1 of the failed-row list, n = *fb_cnt - fb_base; compact row c is query row
    // fb_rows[fb_base + c] and filters with that row's SAFE sample threshold fb_tau0[...]; workgroups beyond n exit at once
    const int *fb_rows;
    const unsigned *fb_cnt;
    int fb_base;
    const float *fb_tau0;
    // STAMP instantiation only (diagnostic, mi_profile_clock): per workgroup (shader cycles, 100 MHz ticks) of its lifetime
    unsigned long long *stamps;
    int stamp_cap;            // entries (pairs) in `stamps`; workgroup b writes pair b % stamp_cap
};

// all-reduce over the 16 lanes of a DPP row (= one C-tile row of the 16x16 MFMA), no LDS traffic:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    return fmaxf(v, dpp_mov<0x140>(v));
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    return v + dpp_mov<0x140>(v);
}

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
// L2 prefetch schedule (measured): touch the lines of K-step t + PF_DIST after MFMA group PF_POS of step t
constexpr int PF_DIST = 2, PF_POS = 4;

template <int DT, int MODE, bool STAMP = false, bool FB = false>
__global__ __launch_bounds__(SIM_THREADS, 2) void sim_kernel(const SimParams p)
{
    static_assert(!FB || MODE == MODE_FILTER, "the fallback instantiation is a filter launch");
    int fb_n = 0;                                    // FB: compact rows in use
    if constexpr (FB) {
        fb_n = (int)*p.fb_cnt - p.fb_base;
        if (fb_n <= 0) return;                       // nothing failed (the normal case): the launch is empty
    }
    __shared__ __attribute__((aligned(16))) char smem[SIM_LDS_BYTES + SIM_EXTRA_LDS];
    // Diagnostic build of the same kernel (STAMP): the in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz.
    // The stamps go to a buffer of their own; no output depends on them.  The production instantiation has none.
    unsigned long long stamp_c0 = 0ull, stamp_r0 = 0ull;
    if constexpr (STAMP) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }

    // ---- block -> tile.  Blocks b and b+8 share an XCD (round-robin dispatch), an XCD holds 32
    // resident blocks (one per CU: 128 KiB LDS), so 32 consecutive same-XCD blocks form one patch
    // of 8 query tiles x 4 bank tiles that share operands through that XCD's L2.  Speed only.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, seq = bid >> 3;
    const int patch = (seq >> 5) * 8 + xcd, inner = seq & 31;
    if (patch >= p.pq * p.pn) return;
    const int qt = ((patch / p.pn) << p.qsh) + (inner & ((1 << p.qsh) - 1));
    const int nt = (patch % p.pn) * (32 >> p.qsh) + (inner >> p.qsh);
    if (qt >= p.qtiles || nt >= p.ntiles) return;
    if constexpr (FB) {
        if (qt * TILE >= fb_n) return;               // no failed row in this query tile
    }
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
        if constexpr (FB) qa = p.fb_rows[p.fb_base + (qa < fb_n ? qa : 0)];      // compact row -> query row
        int xb = n0 + r;
        if constexpr (MODE == MODE_DENSE) {                          // padding columns re-read the last sampled row
            xb = xb < col_end ? xb : col_end - 1;
            xb *= p.row_stride;                                      // < N: (s0 - 1) * floor(N / s0)
        }
        xb = xb < p.N ? xb : p.N - 1;
        a_src[i] = p.q + (long long)qa * p.ldq_bytes + chunk * 16;
        b_src[i] = p.x + (long long)xb * p.ldx_bytes + chunk * 16;
    }
    // The DMA is issued from inline asm on purpose: with the global_load_lds BUILTIN in the loop,
    // hipcc (ROCm 7.2) turns every counted LDS wait of the loop into lgkmcnt(0) and the operand-read
    // pipelining below is lost.  Hidden from the compiler, the loads are counted by hand: one
    // `s_waitcnt vmcnt(0)` in front of the K-step's barrier.  M0 (LDS destination base) is saved
    // and restored inside the statement.  One call stages row group 4w+i of A and of B (2 KiB).
    auto stage_pair = [&](int i, int t, int buf) {
        const unsigned base = (unsigned)(size_t)(AS3 char *)(smem + buf * STAGE_BYTES + w * 4096 + i * 1024);
        const long long koff = (long long)t * KSTEP_BYTES;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\t"
            "s_mov_b32 m0, %[la]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a], off\n\t"
            "s_mov_b32 m0, %[lb]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[b], off\n\t"
            "s_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [la] "s"(base), [lb] "s"(base + OPERAND_TILE_BYTES), [a] "v"(a_src[i] + koff), [b] "v"(b_src[i] + koff)
            : "memory");
    };
    // vmcnt counts in issue order: with a prefetch issued after the K-step's DMA, vmcnt(1) waits for the DMA only
    auto stage_wait = [&](auto pf_tag) {
        if constexpr (decltype(pf_tag)::value) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // The first operand stage goes out before anything else is set up: the prefetch pointers, the operand read
    // offsets and the staging of the thresholds below run while it is in flight.
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_pair(i, 0, 0);

    // ---- L2 prefetch.  The 32 workgroups of an XCD patch stream the same 8 + 4 operand tiles, so 19 % of the
    // DMA requests -- the first touch of every line in this XCD -- miss the L2 and take the Infinity-Cache
    // latency, which a two-buffer pipeline cannot hide (one K-step of lead).  Each workgroup therefore touches
    // its share of the lines of the K-step AFTER next (8 x 4 patch: A tile 4 sharers x 64 rows, B tile 8 sharers
    // x 32 rows = 96 lines, 12 per wave; narrower patches share less and touch more) with a dword load nobody waits
    // for: the DMA one step later hits.
    const char *pf_src;
    {
        const int la = TILE >> (5 - p.qsh), lb = TILE >> p.qsh;        // A / B rows this workgroup touches
        const int per = (la + lb + 7) >> 3;                            // lines per wave (12 .. 33)
        int idx = w * per + (lane % per);
        idx = idx < la + lb ? idx : la + lb - 1;
        if (idx < la) {
            int r = q0 + la * (inner >> p.qsh) + idx; r = r < p.B ? r : p.B - 1;
            if constexpr (FB) r = p.fb_rows[p.fb_base + (r < fb_n ? r : 0)];
            pf_src = p.q + (long long)r * p.ldq_bytes;
        } else {
            int r = n0 + lb * (inner & ((1 << p.qsh) - 1)) + (idx - la);
            if constexpr (MODE == MODE_DENSE) {
                r = r < col_end ? r : col_end - 1;
                r *= p.row_stride;
            }
            r = r < p.N ? r : p.N - 1;
            pf_src = p.x + (long long)r * p.ldx_bytes;
        }
    }
    // The touch is an LDS-DMA dword load into 256 spare bytes of LDS that nothing reads: no destination register
    // exists that the compiler could reuse while the load is still in flight.
    const unsigned pf_lds = (unsigned)(size_t)(AS3 char *)(smem + SIM_LDS_BYTES + 2048);
    auto prefetch = [&](int t) {
        t = t < p.ksteps ? t : p.ksteps - 1;                // never past the end of a row
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dword %[a], off\n\ts_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [l] "s"(pf_lds), [a] "v"(pf_src + (long long)t * KSTEP_BYTES)
            : "memory");
    };

    // ---- MFMA operand read offsets.  Lane reads row (lane & 15) of a 16-row block; K chunk
    // (lane >> 4) [+4 for the second half of the K-step]; swizzle term depends on the lane only.
    const int lr = lane & 15, lq = lane >> 4;
    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wm * 128 + lr) * KSTEP_BYTES;
    const int b_row_off = OPERAND_TILE_BYTES + (wn * 64 + lr) * KSTEP_BYTES;

    if constexpr (MODE == MODE_FILTER) {
        // thresholds of this tile's 256 query rows -> spare LDS (read back in the epilogue)
        // (the K loop's barriers publish them long before the epilogue reads them)
        if (tid < TILE) {
            float t;
            if constexpr (FB) t = (q0 + tid) < fb_n ? p.fb_tau0[p.fb_rows[p.fb_base + q0 + tid]] : INFINITY;
            else t = p.tau[q0 + tid];
            ((float *)(smem + SIM_LDS_BYTES))[tid] = t;
            ((unsigned *)(smem + SIM_LDS_BYTES + 1024))[tid] = 0u;
        }
    }

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define MI_RD(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define MI_MM(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
    // One K-step: 64 MFMAs per wave in 8 groups of 8.  The operand reads are software pipelined
    // (the reads of group g+1 are issued before the MFMAs of group g, two register sets) and the
    // next K-step's DMA is spread over the first four groups (one A + one B piece each) instead of
    // being issued as a burst: a burst keeps BOTH waves of a SIMD out of the matrix pipe at once.
    auto kstep = [&](int t, auto dma_tag, auto pf_tag) {
        constexpr bool DMA = decltype(dma_tag)::value;
        constexpr bool PF = decltype(pf_tag)::value;
        const char *lds = smem + (t & 1) * STAGE_BYTES;
        const int nbuf = (t + 1) & 1;
        if constexpr (DT == DT_BF16) {
            const int coff0 = (lq ^ sw) * 16, coff1 = coff0 ^ 64;
            const char *la = lds + a_row_off, *lb = lds + b_row_off;
            bf16x8 b0[4], b1[4], a0[2], a1[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) b0[j] = *(const bf16x8 *)(lb + j * 2048 + coff0);
            a0[0] = *(const bf16x8 *)(la + coff0);
            a0[1] = *(const bf16x8 *)(la + 2048 + coff0);
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {          // pr = s * 4 + pair
                const int s = pr >> 2, pp = pr & 3;
                bf16x8 *ac = (pr & 1) ? a1 : a0, *an = (pr & 1) ? a0 : a1;
                bf16x8 *bc = s ? b1 : b0;
                if constexpr (DMA) {
                    if (pr < 4) stage_pair(pr, t + 1, nbuf);
                }
                if constexpr (PF) {
                    if (pr == PF_POS) prefetch(t + PF_DIST);
                }
                if (pr < 7) {
                    const int ns = (pr + 1) >> 2, np = (pr + 1) & 3;
                    const int co = ns ? coff1 : coff0;
                    an[0] = *(const bf16x8 *)(la + (2 * np) * 2048 + co);
                    an[1] = *(const bf16x8 *)(la + (2 * np + 1) * 2048 + co);
                    if (pr == 3) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) b1[j] = *(const bf16x8 *)(lb + j * 2048 + coff1);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // boustrophedon over the (A block, B block) grid: consecutive MFMAs share one operand, also
                        // across groups (18 % fewer operand changes; +0.8 ... +1.8 % at C4: the clock rises, not the issue rate;
                        // the f32 loop below gains nothing from it)
                        const int jj = u ? 3 - j : j;
                        acc[2 * pp + u][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[u], bc[jj], acc[2 * pp + u][jj], 0, 0, 0);
                    }
            }
            // pin that interleave (hipcc otherwise sinks every read to just before its first use)
            MI_RD(6);
            MI_RD(2); MI_MM(8);
            MI_RD(2); MI_MM(8);
            MI_RD(2); MI_MM(8);
            MI_RD(6); MI_MM(8);
            MI_RD(2); MI_MM(8);
            MI_RD(2); MI_MM(8);
            MI_RD(2); MI_MM(8);
            MI_MM(8);
        } else {
            // f32: 8 sub-steps of 4 k each; v_mfma_f32_16x16x4_f32 chains k ascending (bit-exact order)
            const char *la = lds + a_row_off, *lb = lds + b_row_off;
            float a0[8], a1[8], b0[4], b1[4];
            {
                const int co = (sw * 16) + lq * 4;          // kk = 0
#pragma unroll
                for (int i = 0; i < 8; ++i) a0[i] = *(const float *)(la + i * 2048 + co);
#pragma unroll
                for (int j = 0; j < 4; ++j) b0[j] = *(const float *)(lb + j * 2048 + co);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float *ac = (kk & 1) ? a1 : a0, *an = (kk & 1) ? a0 : a1;
                float *bc = (kk & 1) ? b1 : b0, *bn = (kk & 1) ? b0 : b1;
                if constexpr (DMA) {
                    if (kk < 4) stage_pair(kk, t + 1, nbuf);
                }
                if constexpr (PF) {
                    if (kk == PF_POS) prefetch(t + PF_DIST);
                }
                if (kk < 7) {
                    const int co = (((kk + 1) ^ sw) * 16) + lq * 4;
#pragma unroll
                    for (int i = 0; i < 8; ++i) an[i] = *(const float *)(la + i * 2048 + co);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bn[j] = *(const float *)(lb + j * 2048 + co);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[i], bc[j], acc[i][j], 0, 0, 0);
            }
            // hipcc pairs the 12 dword reads of a sub-step into 6 ds_read2st64_b32; one of them after every 5 MFMAs stays
            // inside the 32-cycle MFMA's issue shadow (issued as a burst they hold the wave's issue for ~100 cycles per
            // two sub-steps: C3 f32 125 -> 132 TFLOP/s)
            MI_RD(6);
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5);
                MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5);
                MI_MM(2);
            }
            MI_MM(32);
        }
        stage_wait(pf_tag);
        __syncthreads();
    };
#undef MI_RD
#undef MI_MM

    stage_wait(std::false_type{});                   // the first operand stage (issued at the top) and the thresholds
    __syncthreads();
    for (int t = 0; t + 2 < p.ksteps; ++t) kstep(t, std::true_type{}, std::true_type{});
    if (p.ksteps > 1) kstep(p.ksteps - 2, std::true_type{}, std::false_type{});
    kstep(p.ksteps - 1, std::false_type{}, std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the last touches have landed before LDS is reused

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
        // Survivors go straight to HBM with NO global atomic: bank tile nt owns slots [256 nt, 256 nt + 256) of
        // every row's candidate list (cap >= the stage's columns, so the segment always exists and can never
        // overflow), an LDS atomic hands out the slot inside the segment, and the tile ends by storing its 256
        // per-row counts as one contiguous block.  (History: returning GLOBAL atomics per candidate serialised
        // a wave, -19 %; an LDS-staged list flushed through one returning global atomic per row still kept
        // the workgroup -- and its 129 KiB of LDS -- alive for the atomic's round trip at the end of every tile.)
        unsigned *lcnt = (unsigned *)(smem + SIM_LDS_BYTES + 1024);   // [256] zeroed at kernel start
        const float *ltau = (const float *)(smem + SIM_LDS_BYTES);    // [256] staged at kernel start
        f32x4 tau[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tau[i] = *(const f32x4 *)(ltau + wm * 128 + lq * 4 + i * 16);
        bool colok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) colok[j] = (col_base + j * 16) < col_end;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = tau[i][r];
                // one v_max3 + v_max + v_cmp per row instead of four masked compares: the column-validity
                // test only matters in the (rare) slow path below.  NaNs never pass: max ignores them.
                const float m = fmaxf(fmaxf(acc[i][0][r], acc[i][1][r]), fmaxf(acc[i][2][r], acc[i][3][r]));
                if (__any(m >= t)) {
                    const int rl = wm * 128 + lq * 4 + i * 16 + r;        // row within the tile
                    if (q0 + rl < p.B) {
                        // What a survivor costs is its STORE INSTRUCTION (the CU's store path takes ~10-18 cycles per
                        // instruction however few lanes it carries: tools/ablation/c1_epilogue_ab.sh), so the first survivor
                        // of every lane goes out in ONE instruction per row group instead of one per column block; a lane
                        // with more than one survivor (rare below ~5 % of the elements passing) stores the rest behind it.
                        // One slot request (returning LDS atomic) per lane.
                        const bool h0 = colok[0] && acc[i][0][r] >= t, h1 = colok[1] && acc[i][1][r] >= t;
                        const bool h2 = colok[2] && acc[i][2][r] >= t, h3 = colok[3] && acc[i][3][r] >= t;
                        const unsigned c = (h0 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
                        if (c) {
                            uint2 *seg = p.cand + (long long)(q0 + rl) * p.cap + (long long)(p.nt_base + nt) * TILE;
                            const int j1 = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
                            const float v1 = h0 ? acc[i][0][r] : h1 ? acc[i][1][r] : h2 ? acc[i][2][r] : acc[i][3][r];
                            unsigned slot = atomicAdd(lcnt + rl, c);
                            seg[slot] = make_uint2(__float_as_uint(v1), (unsigned)(col_base + j1 * 16));
                            if (c > 1u) {
#pragma unroll
                                for (int j = 1; j < 4; ++j) {
                                    const bool hj = j == 1 ? h1 : j == 2 ? h2 : h3;
                                    if (hj && j != j1) seg[++slot] = make_uint2(__float_as_uint(acc[i][j][r]), (unsigned)(col_base + j * 16));
                                }
                            }
                        }
                    }
                }
            }
        __syncthreads();
        if (tid < TILE) {
            const unsigned n = lcnt[tid];
            const int st = p.nt_base + nt;                                           // tile of the stage
            p.cnt2[(long long)st * p.bpad + q0 + tid] = (unsigned short)n;           // one contiguous block per tile
            // which tiles to look at, per row: a no-return atomic (nothing waits for it)
            if (n) atomicOr(p.tmask + (long long)(q0 + tid) * p.mask_words + (st >> 5), 1u << (st & 31));
        }
        if constexpr (STAMP) {
            if (tid == 0) {
                const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
                unsigned long long *o = p.stamps + 2ll * (bid % p.stamp_cap);
                o[0] = c1 - stamp_c0;
                o[1] = r1 - stamp_r0;
            }
        }
    } else {   // MODE_LSE: per-row (max, sum exp) of sim * inv_t over the tile's 256 columns
        if (bid == 0 && tid == 0 && p.zero_word) *p.zero_word = 0u;    // plain store: zero for the following kernel's atomics
        // Each wave reduces its 64 columns, the 4 column strips of a row meet in LDS (the operand
        // buffers are free after the K loop's last barrier) and the tile writes ONE contiguous
        // [256] float2 block.  (Per-strip 8-byte stores scattered over rows shared 128-byte lines
        // between workgroups of different XCDs: 262k partial-line writes cost 150 us at C2.)
        float2 *lpart = (float2 *)smem;                               // [256 rows][4 strips]
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col_base + j * 16 < col_end) m = fmaxf(m, acc[i][j][r] * p.inv_t);
                m = row16_max(m);
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (col_base + j * 16 < col_end) s += __expf(acc[i][j][r] * p.inv_t - m);     // v_exp_f32: ~1e-7 relative, as in the backward
                s = row16_sum(s);
                if (lr == 0) lpart[(wm * 128 + lq * 4 + i * 16 + r) * 4 + wn] = make_float2(m, s);
            }
        __syncthreads();
        if (tid < TILE) {
            float2 a[4];
            float m = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] = lpart[tid * 4 + j];
                m = fmaxf(m, a[j].x);
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (a[j].y > 0.f) s += a[j].y * expf(a[j].x - m);
            p.lse_part[(long long)nt * p.lse_ld + q0 + tid] = make_float2(m, s);     // lse_ld >= padded rows
        }
    }
}

}  // namespace mi
