/* CPU oracle, C part.  TEST INFRASTRUCTURE ONLY -- never linked into the product library.
 *
 * Restates models/knn.py:6 (sim = feature @ feature_bank) with a *defined* float32
 * accumulation order so that the HIP fp32 kernel can be compared bit for bit:
 *
 *     sim[b][n] = fmaf(q[b][D-1], x[n][D-1], ... fmaf(q[b][1], x[n][1], fmaf(q[b][0], x[n][0], 0)))
 *
 * i.e. one k-ascending fused-multiply-add chain per output, which is exactly what a chain of
 * gfx950 v_mfma_f32_16x16x4_f32 instructions computes (MI355X guide: "bit-for-bit a k-ordered
 * f32 fmaf chain").  The reference itself (torch.mm -> MKL sgemm) uses a different, unspecified
 * order; tests/test_oracle_golden.py bounds the difference between the two on the golden vectors.
 *
 * Pinned against: tests/golden/knn_*.npz (similarities produced by the reference's torch.mm).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ROWS 8

/* q: [B, ldq] row-major, x: [N, ldx] row-major (bank rows), out: [B, N]. */
void oracle_sim_chain_f32(const float *q, int64_t ldq, const float *x, int64_t ldx,
                          int64_t B, int64_t N, int64_t D, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t n0 = 0; n0 < N; n0 += ROWS) {
        int64_t nr = (N - n0 < ROWS) ? (N - n0) : ROWS;
        for (int64_t b = 0; b < B; ++b) {
            const float *qb = q + b * ldq;
            float acc[ROWS];
            for (int j = 0; j < ROWS; ++j) acc[j] = 0.0f;
            for (int64_t k = 0; k < D; ++k) {
                float a = qb[k];
                for (int64_t j = 0; j < nr; ++j)
                    acc[j] = fmaf(a, x[(n0 + j) * ldx + k], acc[j]);
            }
            for (int64_t j = 0; j < nr; ++j) out[b * N + n0 + j] = acc[j];
        }
    }
}

static inline float bf16_to_f32(uint16_t h)
{
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* bf16 inputs (bit patterns), products exact, accumulated in double, rounded once to float.
 * This is the "ideal" value of the bf16 MFMA path (whose internal add order is hardware
 * defined); the GPU result must lie within a few float32 ulps of sum|q.x| of it. */
void oracle_sim_bf16_f64(const uint16_t *q, int64_t ldq, const uint16_t *x, int64_t ldx,
                         int64_t B, int64_t N, int64_t D, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        for (int64_t b = 0; b < B; ++b) {
            double acc = 0.0;
            for (int64_t k = 0; k < D; ++k)
                acc += (double)bf16_to_f32(q[b * ldq + k]) * (double)bf16_to_f32(x[n * ldx + k]);
            out[b * N + n] = (float)acc;
        }
    }
}

/* Per-row top-k with the build's total order (value descending, index ascending), O(N*k)
 * insertion -- used at full bank sizes where a full argsort per row is too slow.
 * sim: [N] one row; out_val/out_idx: [k]. NaN entries are skipped. */
void oracle_topk_row(const float *sim, int64_t N, int k, float *out_val, int64_t *out_idx)
{
    int cnt = 0;
    for (int64_t n = 0; n < N; ++n) {
        float v = sim[n];
        if (v != v) continue;
        if (cnt == k && !(v > out_val[k - 1])) continue;   /* equal value, higher index: loses */
        int p = (cnt < k) ? cnt : k - 1;
        while (p > 0 && out_val[p - 1] < v) {
            out_val[p] = out_val[p - 1];
            out_idx[p] = out_idx[p - 1];
            --p;
        }
        out_val[p] = v;
        out_idx[p] = n;
        if (cnt < k) ++cnt;
    }
    for (int p = cnt; p < k; ++p) { out_val[p] = -INFINITY; out_idx[p] = -1; }
}
