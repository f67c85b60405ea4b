/*
 * CPU oracle for exact inner-product search (test infrastructure only; nothing in the
 * product package links or loads this file).
 *
 * Restates what the reference asks of its index at src/pipeline/retrieval/dense.py:82-84:
 *   scores, indices = index.search(q.astype(float32), top_k)
 * for L2-normalised embeddings (dense.py:63), i.e. exact inner product, best first.  The
 * arithmetic of the reference lives in faiss (un-pinned, not vendored, absent from this
 * image), so this file DEFINES it -- "parity unpinned" against faiss itself:
 *
 *   score(q, c) = s_d  where  s_0 = +0.0f,  s_{j+1} = fmaf(c[j], q[j], s_j)   (j = 0..d-1)
 *
 * one fp32 fused multiply-add per dimension, in dimension order, and results ordered by
 * (score descending, id ascending).  Rows whose score is NaN are never returned.  When
 * fewer than k rows qualify the tail is filled with id -1 and score -FLT_MAX, the faiss
 * convention that dense.py:118-125 tolerates.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define QB 32 /* queries handled together so the FMA chains vectorise across queries */

static inline int better(float s, int64_t id, float s2, int64_t id2) {
    return (s > s2) || (s == s2 && id < id2);
}

/* Insert (s, id) into a best-first list of length k. */
static inline void push(float* ls, int64_t* li, int k, float s, int64_t id) {
    if (!(s == s)) return; /* NaN */
    if (!better(s, id, ls[k - 1], li[k - 1])) return;
    int p = k - 1;
    while (p > 0 && better(s, id, ls[p - 1], li[p - 1])) {
        ls[p] = ls[p - 1];
        li[p] = li[p - 1];
        --p;
    }
    ls[p] = s;
    li[p] = id;
}

/* Scores of every corpus row against every query: out[n * nq + q].  For small checks. */
int oracle_dense_scores(const float* Q, int nq, const float* C, int64_t N, int d, float* out) {
    for (int64_t n = 0; n < N; ++n)
        for (int q = 0; q < nq; ++q) {
            float s = 0.0f;
            for (int j = 0; j < d; ++j) s = fmaf(C[n * d + j], Q[(int64_t)q * d + j], s);
            out[n * nq + q] = s;
        }
    return 0;
}

int oracle_dense_topk(const float* Q, int nq, const float* C, int64_t N, int d, int k,
                      float* out_s, int64_t* out_i, int64_t id_base) {
    if (nq < 0 || N < 0 || d <= 0 || k <= 0) return -1;
    for (int64_t t = 0; t < (int64_t)nq * k; ++t) {
        out_s[t] = -FLT_MAX;
        out_i[t] = -1;
    }
    /* sentinel ids: INT64_MAX loses every tie against a real id */
    float* Qt = (float*)malloc(sizeof(float) * (size_t)d * QB);
    if (!Qt) return -2;
    for (int q0 = 0; q0 < nq; q0 += QB) {
        int nb = nq - q0 < QB ? nq - q0 : QB;
        memset(Qt, 0, sizeof(float) * (size_t)d * QB);
        for (int q = 0; q < nb; ++q)
            for (int j = 0; j < d; ++j) Qt[(size_t)j * QB + q] = Q[(int64_t)(q0 + q) * d + j];
#pragma omp parallel
        {
            float* ls = (float*)malloc(sizeof(float) * (size_t)QB * k);
            int64_t* li = (int64_t*)malloc(sizeof(int64_t) * (size_t)QB * k);
            for (int t = 0; t < QB * k; ++t) {
                ls[t] = -INFINITY;
                li[t] = INT64_MAX;
            }
#pragma omp for schedule(static)
            for (int64_t n = 0; n < N; ++n) {
                float acc[QB];
                for (int q = 0; q < QB; ++q) acc[q] = 0.0f;
                const float* c = C + n * d;
                for (int j = 0; j < d; ++j) {
                    const float cj = c[j];
                    const float* qr = Qt + (size_t)j * QB;
                    for (int q = 0; q < QB; ++q) acc[q] = fmaf(cj, qr[q], acc[q]);
                }
                for (int q = 0; q < nb; ++q) push(ls + q * k, li + q * k, k, acc[q], id_base + n);
            }
#pragma omp critical
            {
                for (int q = 0; q < nb; ++q) {
                    float* gs = out_s + (int64_t)(q0 + q) * k;
                    int64_t* gi = out_i + (int64_t)(q0 + q) * k;
                    for (int t = 0; t < k; ++t) {
                        if (li[q * k + t] == INT64_MAX) break;
                        /* global list uses (-FLT_MAX, -1) fill: treat a fill slot as worst */
                        float s = ls[q * k + t];
                        int64_t id = li[q * k + t];
                        int p = k;
                        while (p > 0 && (gi[p - 1] < 0 || better(s, id, gs[p - 1], gi[p - 1]))) --p;
                        if (p < k) {
                            for (int u = k - 1; u > p; --u) {
                                gs[u] = gs[u - 1];
                                gi[u] = gi[u - 1];
                            }
                            gs[p] = s;
                            gi[p] = id;
                        }
                    }
                }
            }
            free(ls);
            free(li);
        }
    }
    free(Qt);
    return 0;
}
