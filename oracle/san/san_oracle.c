/* Sanitizer harness for the CPU oracle (test infrastructure, like oracle/dvm_oracle.c itself): every entry point of
 * libdvm_oracle on small random inputs with ragged sizes, built with -fsanitize=address,undefined and run by
 * tests/test_sanitizers.py.  A heap overflow, use-after-free, signed overflow or misaligned access aborts the run. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void dvo_rownorm2(const float *x, int rows, int K, float *out);
void dvo_cdist(const float *a, const float *b, int n, int m, int K, int exact, float *out);
void dvo_argmin_exact(const float *f1, const float *f2, int n, int m, int d, int32_t *T, float *dmin);
void dvo_softcorr(const float *f1, const float *f2, int n, int m, int d, float neg_alpha, int topk, float *val, int32_t *idx,
                  float *smax, float *ssum);
void dvo_knn_cdist(const float *x, const float *y, int n, int m, int C, int k, int32_t *idx);
void dvo_knn_neg(const float *a, const float *b, int n, int m, int C, int k, int32_t *idx);
void dvo_apply(const float *val, const int32_t *idx, const float *V, int n, int k, int C, float *out);
void dvo_fps(const float *xyz, int N, int npoint, int start, int32_t *out);
void dvo_dg_build(const float *xyz, int N, int start, int32_t *nodes, int32_t *ring, int32_t *infl, float *dists, float *weights,
                  double *sigma);
void dvo_rot6d(const float *def9, int n, float *R, float *T);
void dvo_dg_warp_arap(const float *xyz, int N, const int32_t *nodes_idx, const int32_t *ring, const int32_t *infl, const float *weights,
                      const float *R, const float *T, float *warped, float *arap_out, float *sr_out);
void dvo_pair_direction(const float *feat1, const float *feat2, const float *verts1, const float *verts2, int N, int M, float negalpha,
                        int fps_start, const float *cw, float cb, const float *W0, const float *b0, const float *W1, const float *b1,
                        const float *W2, const float *b2, const float *W3, const float *b3, int with_map, float *warped, float *verts12,
                        int32_t *T12, float *losses);
void dvo_chamfer(const float *a, const float *b, int n, int m, float *d1, float *d2, int32_t *i1, int32_t *i2);
void dvo_linear(const float *x, const float *w, int M, int K, int Co, const float *bias, const float *res, const float *alpha,
                const float *beta, float slope, float *y);
int dvo_gemm_kblocks(int K, int *starts, int max);
float dvo_aten_sum(const float *x, int n);

static unsigned st = 777u;
static float rnd(void) {
    st = st * 1664525u + 1013904223u;
    return ((st >> 8) / 16777216.0f - 0.5f) * 2.f;
}
static float *randv(size_t n) {
    float *p = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) p[i] = rnd();
    return p;
}

int main(void) {
    const int sizes[][2] = {{1, 1}, {7, 5}, {33, 64}, {65, 31}, {130, 97}};
    for (unsigned s = 0; s < sizeof(sizes) / sizeof(sizes[0]); ++s) {
        const int n = sizes[s][0], m = sizes[s][1], d = 128, k = m < 10 ? m : 10;
        float *f1 = randv((size_t)n * d), *f2 = randv((size_t)m * d), *v1 = randv((size_t)n * 3), *v2 = randv((size_t)m * 3);
        float *out = (float *)malloc((size_t)n * m * sizeof(float));
        float *nr = (float *)malloc((size_t)n * sizeof(float));
        dvo_rownorm2(f1, n, d, nr);
        dvo_cdist(f1, f2, n, m, d, 0, out);
        dvo_cdist(f1, f2, n, m, d, 1, out);
        int32_t *T = (int32_t *)malloc((size_t)n * sizeof(int32_t));
        float *dm = (float *)malloc((size_t)n * sizeof(float));
        dvo_argmin_exact(f1, f2, n, m, d, T, dm);
        float *val = (float *)malloc((size_t)n * k * sizeof(float)), *sm = (float *)malloc(n * sizeof(float)),
              *ss = (float *)malloc(n * sizeof(float));
        int32_t *idx = (int32_t *)malloc((size_t)n * k * sizeof(int32_t));
        dvo_softcorr(f1, f2, n, m, d, -37.5f, k, val, idx, sm, ss);
        float *ap = (float *)malloc((size_t)n * 3 * sizeof(float));
        dvo_apply(val, idx, v2, n, k, 3, ap);
        int32_t *kn = (int32_t *)malloc((size_t)n * k * sizeof(int32_t));
        dvo_knn_cdist(v1, v2, n, m, 3, k, kn);
        dvo_knn_neg(f1, f2, n, m, d, k, kn);
        float *d1 = (float *)malloc(n * sizeof(float)), *d2 = (float *)malloc(m * sizeof(float));
        int32_t *i1 = (int32_t *)malloc(n * sizeof(int32_t)), *i2 = (int32_t *)malloc(m * sizeof(int32_t));
        dvo_chamfer(v1, v2, n, m, d1, d2, i1, i2);
        if (n >= 20) {
            const int Nn = n / 2;
            int32_t *nodes = (int32_t *)malloc(Nn * sizeof(int32_t)), *ring = (int32_t *)malloc((size_t)Nn * 9 * sizeof(int32_t)),
                    *infl = (int32_t *)malloc((size_t)n * 3 * sizeof(int32_t));
            float *dd = (float *)malloc((size_t)n * 3 * sizeof(float)), *ww = (float *)malloc((size_t)n * 3 * sizeof(float));
            double sigma;
            dvo_fps(v1, n, Nn, n - 1, nodes);
            dvo_dg_build(v1, n, 0, nodes, ring, infl, dd, ww, &sigma);
            float *d9 = randv((size_t)Nn * 9), *R = (float *)malloc((size_t)Nn * 9 * sizeof(float)), *Tt = (float *)malloc((size_t)Nn * 3 * sizeof(float));
            dvo_rot6d(d9, Nn, R, Tt);
            float *wp = (float *)malloc((size_t)n * 3 * sizeof(float)), arap, sr;
            dvo_dg_warp_arap(v1, n, nodes, ring, infl, ww, R, Tt, wp, &arap, &sr);
            if (m >= 20) { /* the whole fused direction (graph, soft correspondence, Deformer MLP, warp, Chamfer, map term) */
                float *cw = randv(10), *W0 = randv(512 * 262), *b0 = randv(512), *W1 = randv(256 * 512), *b1 = randv(256), *W2 = randv(128 * 256),
                      *b2 = randv(128), *W3 = randv(9 * 128), *b3 = randv(9);
                float *v12 = (float *)malloc((size_t)n * 3 * sizeof(float)), losses[6];
                int32_t *T12 = (int32_t *)malloc(n * sizeof(int32_t));
                dvo_pair_direction(f1, f2, v1, v2, n, m, -50.f, 3, cw, 0.1f, W0, b0, W1, b1, W2, b2, W3, b3, 1, wp, v12, T12, losses);
                free(cw), free(W0), free(b0), free(W1), free(b1), free(W2), free(b2), free(W3), free(b3), free(v12), free(T12);
            }
            free(nodes), free(ring), free(infl), free(dd), free(ww), free(d9), free(R), free(Tt), free(wp);
        }
        free(f1), free(f2), free(v1), free(v2), free(out), free(nr), free(T), free(dm), free(val), free(sm), free(ss), free(idx), free(ap),
            free(kn), free(d1), free(d2), free(i1), free(i2);
    }
    /* the K-blocked chain at every block rule, ragged M / Co, all epilogue combinations */
    const int Ks[] = {1, 4, 63, 384, 385, 769, 1152, 1157};
    for (unsigned s = 0; s < sizeof(Ks) / sizeof(Ks[0]); ++s) {
        const int K = Ks[s], M = 5 + (int)s, Co = 3 + 2 * (int)s;
        int ks[34];
        const int nb = dvo_gemm_kblocks(K, ks, 33);
        if (ks[0] != 0 || ks[nb] != K) return 2;
        float *x = randv((size_t)M * K), *w = randv((size_t)Co * K), *b = randv(Co), *r = randv((size_t)M * Co), *a = randv(Co), *be = randv(Co);
        float *y = (float *)malloc((size_t)M * Co * sizeof(float));
        dvo_linear(x, w, M, K, Co, NULL, NULL, NULL, NULL, 1.f, y);
        dvo_linear(x, w, M, K, Co, b, r, a, be, 0.2f, y);
        dvo_linear(x, w, M, K, Co, b, NULL, a, be, 0.f, y);
        (void)dvo_aten_sum(x, M * K);
        free(x), free(w), free(b), free(r), free(a), free(be), free(y);
    }
    printf("san_oracle: ok\n");
    return 0;
}
