// sim_walk_kernel.h -- the filter stage of the similarity kernel for SHORT rows (at most 256 bytes: D <= 128 bf16 -- the
// reference's own feature_dim = 128, track_ssl/moco.py:38 -- or D <= 64 f32), gfx950.
//
// sim_kernel.h computes one 256 x 256 tile per workgroup.  With 2 .. 4 K-steps per tile the tile's fixed costs -- set-up,
// the first operand stage that nothing overlaps, the hand-over of the CU to the next workgroup -- are as long as its K loop
// (C1, D = 128: a tile with no survivor at all took 10.6 k cycles for 4.1 k cycles of bf16 MFMA).  Here ONE workgroup keeps
// its 256-row query tile and WALKS a run of bank tiles:
//   * the operand stream never stops at a tile boundary: the K-step that ends tile j stages the first K-step of tile j + 1,
//     so there is no prologue and no seam per tile, only the filter epilogue;
//   * the query tile is staged ONCE and stays resident in LDS (2 x 32 KiB); only bank K-steps are streamed (double buffered,
//     2 x 32 KiB): half the LDS-DMA traffic of a tile.  (A variant that streams both operands, for rows of 257 .. 512 bytes,
//     measured 3 % SLOWER than sim_kernel.h at f32 D = 128 and was dropped: DESIGN.md 5.3.)
// Everything that defines a result is unchanged: the same LDS image and swizzle, the same MFMA sequence per output element
// (f32: one k-ascending chain), the same threshold filter and candidate hand-off (sim_kernel.h, MODE_FILTER) -- a similarity
// does not depend on which kernel computed it (tests: fp32 bit-exact sweeps over D = 32 .. 160).
//
// Work split: workgroup v of qtiles x nsplit handles query tile v % qtiles and bank tiles [split * tps, (split + 1) * tps) of
// the launch.  Workgroups b, b + 8, ... share an XCD (round-robin dispatch) and are given consecutive v, so the workgroups
// of one XCD walk the SAME bank tiles side by side and share them through that XCD's L2.  Speed only.
#pragma once
#include "sim_kernel.h"

namespace mi {

template <int DT>
__global__ __launch_bounds__(SIM_THREADS, 2) void sim_walk_kernel(const SimParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[SIM_LDS_BYTES + SIM_EXTRA_LDS];
    const int bid = blockIdx.x;
    const int nwg = p.qtiles * p.walk_nsplit;
    const int per_xcd = (nwg + 7) >> 3;
    const int v = (bid & 7) * per_xcd + (bid >> 3);
    if ((bid >> 3) >= per_xcd || v >= nwg) return;
    const int split = v / p.qtiles, qt = v - split * p.qtiles;
    const int nt_lo = split * p.walk_tps;
    const int nt_hi = (nt_lo + p.walk_tps) < p.ntiles ? (nt_lo + p.walk_tps) : p.ntiles;
    if (nt_lo >= nt_hi) return;
    const int q0 = qt * TILE;
    const int col_end = p.col0 + p.ncols;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;

    // LDS regions (bytes from smem): query K-step t at t * 32 KiB (t < 2), bank buffer b at 64 KiB + b * 32 KiB
    auto a_off = [&](int t) { return t * OPERAND_TILE_BYTES; };
    auto b_off = [&](int buf) { return 2 * OPERAND_TILE_BYTES + buf * OPERAND_TILE_BYTES; };

    // ---- LDS-DMA sources (as sim_kernel.h): wave w stages row groups 4w .. 4w+3 (8 rows each) of an operand tile; lane ->
    // (row r = 8g + lane/8, LDS chunk slot lane%8), source chunk = slot ^ swz(r).  The addresses are recomputed per piece
    // (a dozen VALU instructions against the 1024 cycles of a K-step's MFMAs) instead of being kept in registers: the
    // walk needs them for a moving bank tile, and the 128 accumulators leave no room for eight more pointers.
    auto src_of = [&](const char *base, long long ld_bytes, int row0, int limit, int i, int st) -> const char * {
        const int r = (w * 4 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        int g = row0 + r;
        g = g < limit ? g : limit - 1;                              // clamp: padding rows re-read a valid row
        return base + (long long)g * ld_bytes + chunk * 16 + st * KSTEP_BYTES;
    };
    // one LDS-DMA piece (1 KiB = 8 rows x 128 B per wave instruction); inline asm for the reason given in sim_kernel.h
    // (counted by hand: one `s_waitcnt vmcnt(0)` in front of the K-step's barrier)
    auto dma = [&](const char *g, int lds_off) {
        const unsigned dst = (unsigned)(size_t)(AS3 char *)(smem + lds_off);
        unsigned keep;
        asm volatile(
            "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[l]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[a], off\n\ts_mov_b32 m0, %[keep]"
            : [keep] "=&s"(keep)
            : [l] "s"(dst), [a] "v"(g)
            : "memory");
    };
    // stage piece i of K-step (bank tile s_nt, K-step st) into buffer `buf`
    int s_nt = nt_lo, s_t = 0;                                       // cursor: the next K-step to stage
    auto stage_piece = [&](int i, int buf) {
        dma(src_of(p.x, p.ldx_bytes, p.col0 + s_nt * TILE, p.N, i, s_t), b_off(buf) + w * 4096 + i * 1024);
    };

    // ---- prologue (once per workgroup): the resident query tile, the first bank K-step, thresholds, counters
    for (int t = 0; t < p.ksteps; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) dma(src_of(p.q, p.ldq_bytes, q0, p.B, i, t), a_off(t) + w * 4096 + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece(i, 0);
    auto advance = [&]() {                                           // cursor -> the K-step after the one just staged
        if (++s_t == p.ksteps) {
            s_t = 0;
            if (s_nt + 1 < nt_hi) ++s_nt;                            // past the end: the last tile is re-staged and never read
        }
    };
    advance();

    const int lr = lane & 15, lq = lane >> 4;
    const int sw = (lane >> 1) & 7;
    const int a_row_off = (wm * 128 + lr) * KSTEP_BYTES;
    const int b_row_off = (wn * 64 + lr) * KSTEP_BYTES;
    if (tid < TILE) {
        // rows past B (padding of the last query tile) get a NaN threshold: nothing compares >= NaN, so the epilogue needs no row test
        ((float *)(smem + SIM_LDS_BYTES))[tid] = (q0 + tid) < p.B ? p.tau[q0 + tid] : __builtin_nanf("");
        ((unsigned *)(smem + SIM_LDS_BYTES + 1024))[tid] = 0u;
    }

    f32x4 acc[8][4];

#define MI_RD(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define MI_MM(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
    // One K-step (the body of sim_kernel.h's): 64 MFMAs per wave in 8 groups of 8, operand reads software pipelined, the
    // NEXT K-step's DMA (cursor s_t, buffer nbuf) spread over the first four groups.
    auto kstep = [&](const char *la_base, const char *lb_base, int nbuf) {
        const char *la = la_base + a_row_off, *lb = lb_base + b_row_off;
        if constexpr (DT == DT_BF16) {
            const int coff0 = (lq ^ sw) * 16, coff1 = coff0 ^ 64;
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
                if (pr < 4) stage_piece(pr, nbuf);
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
                        const int jj = u ? 3 - j : j;          // boustrophedon, as sim_kernel.h
                        acc[2 * pp + u][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[u], bc[jj], acc[2 * pp + u][jj], 0, 0, 0);
                    }
            }
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
                if (kk < 4) stage_piece(kk, nbuf);
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
            MI_RD(6);
#pragma unroll
            for (int kk = 0; kk < 7; ++kk) {
                MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5);
                MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5); MI_RD(1); MI_MM(5);
                MI_MM(2);
            }
            MI_MM(32);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
#undef MI_RD
#undef MI_MM

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the resident query tile, the first bank K-step; thresholds
    __syncthreads();

    unsigned *lcnt = (unsigned *)(smem + SIM_LDS_BYTES + 1024);   // [256] survivors per row: of the tile, or of the run so far (walk_compact)
    const float *ltau = (const float *)(smem + SIM_LDS_BYTES);    // [256]
    int buf = 0;
    for (int nt = nt_lo; nt < nt_hi; ++nt) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < p.ksteps; ++t) {
            kstep(smem + a_off(t), smem + b_off(buf), buf ^ 1);
            advance();
            buf ^= 1;
        }
        // ---- filter epilogue of tile nt (sim_kernel.h, MODE_FILTER): C/D map of the 16x16 MFMA: column = lane & 15,
        // row = (lane >> 4) * 4 + reg
        const int n0 = p.col0 + nt * TILE;
        const int col_base = n0 + wn * 64 + lr;               // + j*16
        // opaque copy of the query tile's first row: keeps the compiler from hoisting the 32 per-row 64-bit segment
        // offsets of the slow path out of the tile loop (it did, and spilled them: 220 bytes of scratch per lane)
        int q0e = q0;
        asm volatile("" : "+s"(q0e));
        int row_l = wm * 128 + lq * 4;                        // same for the lane's first row within the tile (LDS counter addresses)
        asm volatile("" : "+v"(row_l));
        uint2 *seg_base = p.cand + (long long)q0e * p.cap + (long long)(p.nt_base + (p.walk_compact ? nt_lo : nt)) * TILE;
        bool colok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) colok[j] = (col_base + j * 16) < col_end;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 tau_i = *(const f32x4 *)(ltau + row_l + i * 16);     // thresholds of this block's 4 rows
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float t = tau_i[r];
                // four compares whose masks the slow path uses again, OR-ed on the scalar side -- not max3 + max + compare (fmaxf
                // also costs a canonicalising v_max x, x per operand pair): the epilogue is VALU-issue bound at D = 128
                // (1190 VALU instructions per wave and tile beside 128 MFMAs: tools/pmc_c1.sh).  NaNs never pass.
                const bool g0 = acc[i][0][r] >= t, g1 = acc[i][1][r] >= t, g2 = acc[i][2][r] >= t, g3 = acc[i][3][r] >= t;
                if (__any(g0 || g1 || g2 || g3)) {
                    const int rl = row_l + i * 16 + r;                    // row within the tile (a padding row's threshold is NaN: never here)
                    {
                        // What a survivor costs is its STORE INSTRUCTION (the CU's store path takes ~10-18 cycles per
                        // instruction however few lanes it carries), so the first survivor of every lane goes out in ONE
                        // instruction per row group instead of one per column block; a lane with more than one survivor
                        // (rare below ~5 % of the elements passing) stores the rest behind it.  One slot request per lane.
                        const bool h0 = colok[0] && g0, h1 = colok[1] && g1, h2 = colok[2] && g2, h3 = colok[3] && g3;
                        const unsigned c = (h0 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
                        if (c) {
                            // segment of (row, tile or run): a wave-uniform base + a 32-bit element offset (rl < 256, cap < 2^24:
                            // the host checks) -- a 24-bit multiply and one 64-bit add per row group instead of a 64-bit multiply
                            const unsigned seg_off = __umul24((unsigned)rl, (unsigned)p.cap);
                            const int j1 = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
                            uint2 *seg = seg_base + seg_off;
                            unsigned slot = atomicAdd(lcnt + rl, c);
                            const float v1 = h0 ? acc[i][0][r] : h1 ? acc[i][1][r] : h2 ? acc[i][2][r] : acc[i][3][r];
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
        }
        // publish the counts: per tile, or once for the run (walk_compact: the counters run on through the tiles, whose
        // K-steps' barriers order the epilogues' atomics; the refresh then reads a row's survivors of the run as ONE list
        // instead of one short segment per tile -- at C1 four entries each)
        if (!p.walk_compact || nt + 1 == nt_hi) {
            __syncthreads();
            if (tid < TILE) {
                const unsigned n = lcnt[tid];
                lcnt[tid] = 0u;                                                      // for the next tile (its K-steps' barriers publish it)
                const int st = p.nt_base + (p.walk_compact ? nt_lo : nt);            // tile of the stage
                p.cnt2[(long long)st * p.bpad + q0e + tid] = (unsigned short)n;
                if (n) atomicOr(p.tmask + (long long)(q0e + tid) * p.mask_words + (st >> 5), 1u << (st & 31));
            }
        }
    }
}

}  // namespace mi
