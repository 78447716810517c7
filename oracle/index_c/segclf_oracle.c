/* ORACLE - TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C index-form restatement of the reference's SegmentClassifier forward
 * (reference gnn/model.py:140-156; index semantics gnn/graph.py:128-135; padded
 * segments gnn/trainSegmentClassifier.py:83-93).  It is the CPU checker for the
 * HIP path at sizes where the dense formulation does not fit, and one leg of
 * bench.py's cpu_baseline.  Pinned against outputs of the reference itself through
 * tests/golden/ (see oracle/__init__.py).  Nothing in the product links this file.
 *
 *   H0_n   = [ tanh(Win X_n + bin) | X_n ]                                  model.py:144-146
 *   e_j    = sigmoid(W2 tanh(W1 [H_src(j) | H_dst(j)] + b1) + b2)           model.py:71-73,45-49
 *   mi_n   = sum_{j: dst(j)=n} e_j H_src(j);  mo_n = sum_{j: src(j)=n} e_j H_dst(j)   :116-119
 *   H'_n   = tanh(W4 tanh(W3 [mi_n | mo_n | H_n] + b3) + b4);  H_n <- [H'_n | X_n]     :120,125,154
 *   output = e after one more edge pass                                      model.py:156
 *
 * Sums run in ascending segment id per hit (fixed order), in `real` = float or double.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int build_csr(const int *key, long n_seg, long n_hits, long **ptr_out, int **eid_out)
{
    long *ptr = (long *)calloc((size_t)n_hits + 1, sizeof(long));
    long *fill = (long *)malloc(((size_t)n_hits + 1) * sizeof(long));
    int *eid = (int *)malloc((size_t)(n_seg > 0 ? n_seg : 1) * sizeof(int));
    if (!ptr || !fill || !eid) { free(ptr); free(fill); free(eid); return -1; }
    for (long j = 0; j < n_seg; ++j)
        if (key[j] >= 0) ptr[key[j] + 1]++;
    for (long n = 0; n < n_hits; ++n) ptr[n + 1] += ptr[n];
    memcpy(fill, ptr, ((size_t)n_hits + 1) * sizeof(long));
    for (long j = 0; j < n_seg; ++j)
        if (key[j] >= 0) eid[fill[key[j]]++] = (int)j;
    free(fill);
    *ptr_out = ptr;
    *eid_out = eid;
    return 0;
}

#define DEFINE_ORACLE(NAME, real, TANH, EXP)                                                      \
static void NAME##_edge(const real *H, int C, const int *src, const int *dst, long n_seg,         \
                        const float *W1, const float *b1, const float *W2, const float *b2,       \
                        int D, real *e)                                                           \
{                                                                                                 \
    _Pragma("omp parallel for schedule(static)")                                                  \
    for (long j = 0; j < n_seg; ++j) {                                                            \
        const real *hs = src[j] >= 0 ? H + (size_t)src[j] * C : NULL;                             \
        const real *hd = dst[j] >= 0 ? H + (size_t)dst[j] * C : NULL;                             \
        real z = (real)b2[0];                                                                     \
        for (int d = 0; d < D; ++d) {                                                             \
            const float *w = W1 + (size_t)d * 2 * C;                                              \
            real acc = (real)b1[d];                                                               \
            if (hs) for (int k = 0; k < C; ++k) acc += (real)w[k] * hs[k];                        \
            if (hd) for (int k = 0; k < C; ++k) acc += (real)w[C + k] * hd[k];                    \
            z += (real)W2[d] * TANH(acc);                                                         \
        }                                                                                         \
        e[j] = (real)1 / ((real)1 + EXP(-z));                                                     \
    }                                                                                             \
}                                                                                                 \
                                                                                                  \
static void NAME##_node(const real *H, int C, int F, const real *e, const int *src,               \
                        const int *dst, const long *in_ptr, const int *in_eid,                    \
                        const long *out_ptr, const int *out_eid, long n_hits,                     \
                        const float *W3, const float *b3, const float *W4, const float *b4,       \
                        int D, real *Hn)                                                          \
{                                                                                                 \
    _Pragma("omp parallel")                                                                       \
    {                                                                                             \
        real *M = (real *)malloc(((size_t)3 * C + D) * sizeof(real));                             \
        real *q = M + 3 * C;                                                                      \
        _Pragma("omp for schedule(static)")                                                       \
        for (long n = 0; n < n_hits; ++n) {                                                       \
            for (int k = 0; k < 2 * C; ++k) M[k] = 0;                                             \
            for (long p = in_ptr[n]; p < in_ptr[n + 1]; ++p) {                                    \
                int j = in_eid[p];                                                                \
                const real *h = H + (size_t)src[j] * C;                                           \
                for (int k = 0; k < C; ++k) M[k] += e[j] * h[k];                                  \
            }                                                                                     \
            for (long p = out_ptr[n]; p < out_ptr[n + 1]; ++p) {                                  \
                int j = out_eid[p];                                                               \
                const real *h = H + (size_t)dst[j] * C;                                           \
                for (int k = 0; k < C; ++k) M[C + k] += e[j] * h[k];                              \
            }                                                                                     \
            for (int k = 0; k < C; ++k) M[2 * C + k] = H[(size_t)n * C + k];                      \
            for (int d = 0; d < D; ++d) {                                                         \
                real acc = (real)b3[d];                                                           \
                for (int k = 0; k < 3 * C; ++k) acc += (real)W3[(size_t)d * 3 * C + k] * M[k];    \
                q[d] = TANH(acc);                                                                 \
            }                                                                                     \
            for (int d = 0; d < D; ++d) {                                                         \
                real acc = (real)b4[d];                                                           \
                for (int k = 0; k < D; ++k) acc += (real)W4[(size_t)d * D + k] * q[k];            \
                Hn[(size_t)n * C + d] = TANH(acc);                                                \
            }                                                                                     \
            for (int k = 0; k < F; ++k) Hn[(size_t)n * C + D + k] = H[(size_t)n * C + D + k];     \
        }                                                                                         \
        free(M);                                                                                  \
    }                                                                                             \
}                                                                                                 \
                                                                                                  \
static int NAME(const float *X, long n_hits, int F, const int *src, const int *dst, long n_seg,   \
                const float *Win, const float *bin, const float *W1, const float *b1,             \
                const float *W2, const float *b2, const float *W3, const float *b3,               \
                const float *W4, const float *b4, int D, int n_iters,                             \
                float *e_out, float *e_trace, float *H_trace)                                     \
{                                                                                                 \
    const int C = D + F;                                                                          \
    long *in_ptr = NULL, *out_ptr = NULL;                                                         \
    int *in_eid = NULL, *out_eid = NULL;                                                          \
    real *H = (real *)malloc((size_t)(n_hits > 0 ? n_hits : 1) * C * sizeof(real));               \
    real *Hn = (real *)malloc((size_t)(n_hits > 0 ? n_hits : 1) * C * sizeof(real));              \
    real *e = (real *)malloc((size_t)(n_seg > 0 ? n_seg : 1) * sizeof(real));                     \
    int rc = -1;                                                                                  \
    if (!H || !Hn || !e) goto done;                                                               \
    if (build_csr(dst, n_seg, n_hits, &in_ptr, &in_eid)) goto done;                               \
    if (build_csr(src, n_seg, n_hits, &out_ptr, &out_eid)) goto done;                             \
    _Pragma("omp parallel for schedule(static)")                                                  \
    for (long n = 0; n < n_hits; ++n) {                                                           \
        for (int d = 0; d < D; ++d) {                                                             \
            real acc = (real)bin[d];                                                              \
            for (int k = 0; k < F; ++k) acc += (real)Win[d * F + k] * (real)X[(size_t)n * F + k]; \
            H[(size_t)n * C + d] = TANH(acc);                                                     \
        }                                                                                         \
        for (int k = 0; k < F; ++k) H[(size_t)n * C + D + k] = (real)X[(size_t)n * F + k];        \
    }                                                                                             \
    for (int t = 0; t <= n_iters; ++t) {                                                          \
        if (H_trace)                                                                              \
            for (size_t i = 0; i < (size_t)n_hits * C; ++i)                                       \
                H_trace[(size_t)t * n_hits * C + i] = (float)H[i];                                \
        NAME##_edge(H, C, src, dst, n_seg, W1, b1, W2, b2, D, e);                                 \
        if (e_trace)                                                                              \
            for (long j = 0; j < n_seg; ++j) e_trace[(size_t)t * n_seg + j] = (float)e[j];        \
        if (t == n_iters) break;                                                                  \
        NAME##_node(H, C, F, e, src, dst, in_ptr, in_eid, out_ptr, out_eid, n_hits,               \
                    W3, b3, W4, b4, D, Hn);                                                       \
        { real *tmp = H; H = Hn; Hn = tmp; }                                                      \
    }                                                                                             \
    for (long j = 0; j < n_seg; ++j) e_out[j] = (float)e[j];                                      \
    rc = 0;                                                                                       \
done:                                                                                             \
    free(H); free(Hn); free(e); free(in_ptr); free(in_eid); free(out_ptr); free(out_eid);         \
    return rc;                                                                                    \
}

DEFINE_ORACLE(fwd_f32, float, tanhf, expf)
DEFINE_ORACLE(fwd_f64, double, tanh, exp)

/* Returns 0 on success, -1 on allocation failure, -2 on bad arguments. */
int segclf_oracle_forward(const float *X, long n_hits, int F, const int *src, const int *dst,
                          long n_seg, const float *Win, const float *bin, const float *W1,
                          const float *b1, const float *W2, const float *b2, const float *W3,
                          const float *b3, const float *W4, const float *b4, int D, int n_iters,
                          float *e_out, float *e_trace, float *H_trace, int n_threads,
                          int use_f64)
{
    if (n_hits < 0 || n_seg < 0 || F <= 0 || D <= 0 || n_iters < 0) return -2;
    for (long j = 0; j < n_seg; ++j) {
        if ((src[j] < 0) != (dst[j] < 0)) return -2;
        if (src[j] >= n_hits || dst[j] >= n_hits) return -2;
    }
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
    return use_f64 ? fwd_f64(X, n_hits, F, src, dst, n_seg, Win, bin, W1, b1, W2, b2, W3, b3,
                             W4, b4, D, n_iters, e_out, e_trace, H_trace)
                   : fwd_f32(X, n_hits, F, src, dst, n_seg, Win, bin, W1, b1, W2, b2, W3, b3,
                             W4, b4, D, n_iters, e_out, e_trace, H_trace);
}

int segclf_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
