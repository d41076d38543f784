/* flat_search.c -- the C-ABI of librmu.so from plain C (no Python, no torch): build an index, search, merge two shards.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/flat_search.c -Lragmeup_amd/lib -lrmu -Wl,-rpath,$PWD/ragmeup_amd/lib -lm -o flat_search
 *   ./flat_search            (needs an MI355X; the library has no CPU fallback)
 *
 * This is the binding any host language with a C FFI (cgo, JNI, N-API, ctypes) writes against include/rmu.h; the
 * reference itself reaches the same calls through the LangChain classes in ragmeup_amd/ (INTEGRATION.md). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rmu.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != RMU_OK) {                                                     \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rmu_last_error()); \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (float)(*s >> 8) / 16777216.0f - 0.5f; }

int main(void) {
    enum { N = 100000, D = 384, NQ = 4, K = 10 };
    unsigned seed = 1234u;
    float* x = (float*)malloc(sizeof(float) * N * D);
    float* q = (float*)malloc(sizeof(float) * NQ * D);
    float scores[NQ * K], s2[2 * NQ * K], ms[NQ * K];
    int64_t rows[NQ * K], r2[2 * NQ * K], mr[NQ * K];
    if (!x || !q) return 1;
    for (long i = 0; i < (long)N * D; ++i) x[i] = frand(&seed);
    for (int i = 0; i < NQ * D; ++i) q[i] = x[(long)(i / D) * 777 * D + i % D] + 0.05f * frand(&seed);   /* near rows 0, 777, ... */

    printf("%s\n", rmu_version());
    CHECK(rmu_init(0));
    rmu_index_t* idx = NULL;
    int64_t first = -1, n = 0;
    CHECK(rmu_index_create(&idx, D, RMU_METRIC_COSINE, N));
    CHECK(rmu_index_add(idx, x, N, /*is_device=*/0, &first));
    CHECK(rmu_index_size(idx, &n));
    CHECK(rmu_index_search(idx, q, NQ, K, /*flags=*/0, /*row_base=*/0, scores, rows, /*hip_stream=*/0));
    for (int i = 0; i < NQ; ++i) printf("query %d: best row %lld (cosine %.4f), expected %d\n", i, (long long)rows[i * K], scores[i * K], i * 777);

    /* two shards of the same corpus + one merge: what the 8-GPU path does after its all-gather */
    rmu_index_t *a = NULL, *b = NULL;
    CHECK(rmu_index_create(&a, D, RMU_METRIC_COSINE, N / 2));
    CHECK(rmu_index_create(&b, D, RMU_METRIC_COSINE, N / 2));
    CHECK(rmu_index_add(a, x, N / 2, 0, &first));
    CHECK(rmu_index_add(b, x + (long)(N / 2) * D, N / 2, 0, &first));
    CHECK(rmu_index_search(a, q, NQ, K, 0, 0, s2, r2, 0));
    CHECK(rmu_index_search(b, q, NQ, K, 0, N / 2, s2 + NQ * K, r2 + NQ * K, 0));
    CHECK(rmu_topk_merge(s2, r2, 2, NQ, K, 0, ms, mr, 0));
    int same = 1;
    for (int i = 0; i < NQ * K; ++i) same &= (mr[i] == rows[i]) && (fabsf(ms[i] - scores[i]) <= 1e-6f);
    printf("sharded search + merge equals the single-index search: %s\n", same ? "yes" : "NO");

    rmu_index_free(a); rmu_index_free(b); rmu_index_free(idx);
    free(x); free(q);
    return same ? 0 : 2;
}
