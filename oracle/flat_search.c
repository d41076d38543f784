/* CPU ORACLE (plain C) for the dense flat search -- TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY UNPINNED: the reference holds no golden vectors for this path; the
 * arithmetic lives in Milvus-Lite 2.4.7 (FLAT, metric L2) / pgvector v0.8.0 (`<=>`), neither of
 * which is vendored under /root/reference nor installable here.
 *
 * Restates: reference call site server/RAGHelper.py:497-499 (dense retriever) ->
 *   3P Milvus `col.search(..., limit=k)` on a FLAT index == exhaustive scan of every stored
 *   row, one fp32 dot product / squared distance per (query,row), keep the best k.
 * Tie rule: order by (-score, row).
 *
 * fp32 dot in k order with fmaf (what a scalar CPU build of a FLAT scan does); rows split
 * across OpenMP threads, per-thread top-k merged at the end.  Built by oracle/Makefile into
 * oracle/_build/liboracle_flat.so.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float s; int64_t r; } cand_t;

static inline int better(float s, int64_t r, const cand_t *c) {
    return (s > c->s) || (s == c->s && r < c->r);
}

/* insert into a sorted (best first) list of length k */
static inline void insert(cand_t *top, int k, float s, int64_t r) {
    if (!better(s, r, &top[k - 1])) return;
    int i = k - 1;
    while (i > 0 && better(s, r, &top[i - 1])) { top[i] = top[i - 1]; --i; }
    top[i].s = s; top[i].r = r;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* metric: 0 = inner product, 1 = cosine, 2 = negative squared L2 (larger = better everywhere).
 * alive: optional byte mask (NULL = all rows live).  out_s/out_r are [nq*k]; missing = (-inf,-1). */
int oracle_flat_search(const float *q, int64_t nq, const float *x, int64_t n, int d, int k,
                       int metric, const uint8_t *alive, float *out_s, int64_t *out_r) {
    if (!q || !x || !out_s || !out_r || k <= 0 || d <= 0) return -1;
    int nt = oracle_num_threads();
    cand_t *all = (cand_t *)malloc(sizeof(cand_t) * (size_t)nt * (size_t)nq * (size_t)k);
    if (!all) return -3;
    for (size_t i = 0; i < (size_t)nt * nq * k; ++i) { all[i].s = -INFINITY; all[i].r = INT64_MAX; }

#pragma omp parallel num_threads(nt)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
#else
        int t = 0;
#endif
        int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        for (int64_t qi = 0; qi < nq; ++qi) {
            const float *qv = q + qi * d;
            cand_t *top = all + ((size_t)t * nq + qi) * k;
            float qn = 0.f;
            for (int j = 0; j < d; ++j) qn = fmaf(qv[j], qv[j], qn);
            for (int64_t r = lo; r < hi; ++r) {
                if (alive && !alive[r]) continue;
                const float *xv = x + r * d;
                float dot = 0.f, xn = 0.f;
                for (int j = 0; j < d; ++j) dot = fmaf(qv[j], xv[j], dot);
                float s = dot;
                if (metric != 0) {
                    for (int j = 0; j < d; ++j) xn = fmaf(xv[j], xv[j], xn);
                    if (metric == 1) s = dot / (sqrtf(qn) * sqrtf(xn));
                    else s = -(qn - 2.f * dot + xn);
                }
                if (s != s) continue; /* NaN never ranks */
                insert(top, k, s, r);
            }
        }
    }
    for (int64_t qi = 0; qi < nq; ++qi) {
        cand_t *dst = all + (size_t)qi * k; /* thread 0's list doubles as the merge target */
        for (int t = 1; t < nt; ++t) {
            cand_t *src = all + ((size_t)t * nq + qi) * k;
            for (int i = 0; i < k; ++i)
                if (src[i].r != INT64_MAX) insert(dst, k, src[i].s, src[i].r);
        }
        for (int i = 0; i < k; ++i) {
            out_s[qi * k + i] = dst[i].s;
            out_r[qi * k + i] = dst[i].r == INT64_MAX ? -1 : dst[i].r;
        }
    }
    free(all);
    return 0;
}
