/* CPU port of the ELBO train step in C + OpenMP -- TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/ and bench.py's cpu_baseline leg load this (through ctypes); the product path
 * (vae_training_amd) never does.  PARITY UNPINNED, like oracle/elbo_oracle.py (the reference ships no
 * tests or fixtures and JAX/Flax are not importable here): this file is checked against that float64
 * NumPy oracle in tests/test_oracle_c.py.
 *
 * It restates, in float32 and sample by sample, /root/reference/networks.py:
 *   :26-44  FullyConnectedNetwork.apply  (Dense + relu between layers)          -> mlp_forward
 *   :61-84  VAE.apply                    (encoder, reparameterisation, decoder(s), decoder noise)
 *   :94-98  the loss                     (Dkl + Gaussian reconstruction, batch mean)
 *   :99     jax.value_and_grad           -> the hand-derived backward of SURVEY.md 8a row a5
 *   :100    flax.optim.Adam.apply_gradient (ASSUMED-FROM-API: beta1 .9, beta2 .999, eps 1e-8)
 * Parallelisation: OpenMP over samples with one private gradient accumulator per thread, summed in
 * thread order (deterministic for a fixed thread count).  Flat parameter layout = include/vaek.h.
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC oracle/elbo_ref.c -o oracle/libelbo_ref.so -lm   (oracle/Makefile)
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define MAXL 9      /* layers per network: up to 8 hidden + output */
#define MAXW 1024   /* widest layer supported by the per-thread scratch */

typedef struct {
    int D, L;
    int n_enc, enc[8];
    int n_dec, dec[8];
    int sigmoid, tdv;
    float eps_cli;
} elbo_ref_cfg;

typedef struct { int n; int in[MAXL], out[MAXL]; long w_off[MAXL]; } net_t;

static long build_net(net_t* n, int fan_in, const int* hidden, int n_hidden, int last, long off) {
    n->n = n_hidden + 1;
    int k = fan_in;
    for (int i = 0; i <= n_hidden; ++i) {
        const int o = i < n_hidden ? hidden[i] : last;
        n->in[i] = k; n->out[i] = o; n->w_off[i] = off;
        off += (long)k * o + o;
        k = o;
    }
    return off;
}

long elbo_ref_param_count(const elbo_ref_cfg* c) {
    net_t t;
    long off = build_net(&t, c->D, c->enc, c->n_enc, c->L, 0);
    off = build_net(&t, c->L, c->dec, c->n_dec, c->D, off);
    if (c->sigmoid) off = build_net(&t, c->L, c->dec, c->n_dec, c->D, off);
    return off + c->L + (c->tdv ? 1 : 0);
}

/* acts[i] = input of layer i (acts[0] = x); returns pointer to the last Dense output (no relu after it) */
static const float* mlp_forward(const net_t* n, const float* p, const float* x, float acts[][MAXW], float* out) {
    const float* h = x;
    for (int i = 0; i < n->n; ++i) {
        const float* w = p + n->w_off[i];
        const float* b = w + (long)n->in[i] * n->out[i];
        float* y = (i + 1 < n->n) ? acts[i + 1] : out;
        for (int o = 0; o < n->out[i]; ++o) y[o] = b[o];
        for (int k = 0; k < n->in[i]; ++k) {
            const float hk = h[k];
            const float* wr = w + (long)k * n->out[i];
            for (int o = 0; o < n->out[i]; ++o) y[o] += hk * wr[o];
        }
        if (i + 1 < n->n) for (int o = 0; o < n->out[i]; ++o) y[o] = y[o] > 0.f ? y[o] : 0.f;   /* utils.py:29-30 */
        h = y;
    }
    return out;
}

/* d_out: gradient w.r.t. the last Dense output; accumulates dW|db into g; writes d(input) to dx if non-NULL */
static void mlp_backward(const net_t* n, const float* p, float* g, const float* x, float acts[][MAXW], const float* d_out,
                         float* dx, float* tmp_a, float* tmp_b) {
    const float* d = d_out;
    for (int i = n->n - 1; i >= 0; --i) {
        const float* w = p + n->w_off[i];
        float* gw = g + n->w_off[i];
        float* gb = gw + (long)n->in[i] * n->out[i];
        const float* h = i == 0 ? x : acts[i];
        for (int o = 0; o < n->out[i]; ++o) gb[o] += d[o];
        float* dn = (i == 0) ? dx : (d == tmp_a ? tmp_b : tmp_a);
        for (int k = 0; k < n->in[i]; ++k) {
            const float hk = h[k];
            float* gr = gw + (long)k * n->out[i];
            const float* wr = w + (long)k * n->out[i];
            float acc = 0.f;
            for (int o = 0; o < n->out[i]; ++o) { gr[o] += hk * d[o]; acc += d[o] * wr[o]; }
            if (dn) dn[k] = (i > 0 && !(hk > 0.f)) ? 0.f : acc;      /* relu'(pre) == [post > 0] */
        }
        d = dn;
    }
}

/* Returns the loss.  grads (P + 4 floats, may be NULL): gradient, then {loss, mean Dkl, mean mse, 0}.
 * apply != 0: Adam update of params/m/v with step t (1-based).  batch_total <= 0 -> B. */
float elbo_ref_step(const elbo_ref_cfg* c, float* params, float* m, float* v, int t, const float* x, const float* z1,
                    const float* z2, int B, long batch_total, float lr, float* grads, int apply, int nthreads) {
    net_t enc, dec, sig;
    long off = build_net(&enc, c->D, c->enc, c->n_enc, c->L, 0);
    off = build_net(&dec, c->L, c->dec, c->n_dec, c->D, off);
    if (c->sigmoid) off = build_net(&sig, c->L, c->dec, c->n_dec, c->D, off);
    const long off_epsp = off, off_eps = c->tdv ? off + c->L : -1, P = off + c->L + (c->tdv ? 1 : 0);
    const int D = c->D, L = c->L;
    const float Bt = (float)(batch_total > 0 ? batch_total : B);
    const float eps = c->tdv ? params[off_eps] * c->eps_cli : c->eps_cli;
    const float inv_var = expf(-eps), sigma = expf(0.5f * eps), inv_bt = 1.f / Bt;
    float sdev[MAXW];
    for (int l = 0; l < L; ++l) sdev[l] = expf(0.5f * params[off_epsp + l]);              /* :73 */
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    float* acc = (float*)calloc((size_t)nthreads * (P + 4), sizeof(float));
    if (!acc) return NAN;
#pragma omp parallel num_threads(nthreads)
    {
        float* g = acc + (size_t)omp_get_thread_num() * (P + 4);
        float (*ea)[MAXW] = malloc(sizeof(float[MAXL][MAXW]));
        float (*da)[MAXW] = malloc(sizeof(float[MAXL][MAXW]));
        float (*sa)[MAXW] = malloc(sizeof(float[MAXL][MAXW]));
        float mu[MAXW], smp[MAXW], y[MAXW], ys[MAXW], dy[MAXW], dys[MAXW], ds[MAXW], ds2[MAXW], ta[MAXW], tb[MAXW];
        double s_mse = 0, s_musq = 0, s_deps = 0;
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            const float* xb = x + (long)b * D; const float* z1b = z1 + (long)b * L; const float* z2b = z2 + (long)b * D;
            mlp_forward(&enc, params, xb, ea, mu);                                       /* :67-68 */
            for (int l = 0; l < L; ++l) { smp[l] = mu[l] + sdev[l] * z1b[l]; s_musq += (double)mu[l] * mu[l]; }
            mlp_forward(&dec, params, smp, da, y);                                       /* :77 / :80 */
            if (c->sigmoid) mlp_forward(&sig, params, smp, sa, ys);                      /* :76 */
            for (int d = 0; d < D; ++d) {
                float sg = 0.f, xh = y[d] + sigma * z2b[d];                              /* :81-83 */
                if (c->sigmoid) { sg = 1.f / (1.f + expf(-ys[d])); xh += sg; }
                const float r = xh - xb[d];
                const float q = r * r * inv_var;
                s_mse += 0.5 * q;                                                        /* :95-96, constant part added below */
                s_deps += -0.5 * q + 0.5 * sigma * z2b[d] * r * inv_var;
                dy[d] = r * inv_var * inv_bt;
                if (c->sigmoid) dys[d] = dy[d] * sg * (1.f - sg);
            }
            mlp_backward(&dec, params, g, smp, da, dy, ds, ta, tb);
            if (c->sigmoid) { mlp_backward(&sig, params, g, smp, sa, dys, ds2, ta, tb); for (int l = 0; l < L; ++l) ds[l] += ds2[l]; }
            for (int l = 0; l < L; ++l) { g[off_epsp + l] += ds[l] * z1b[l]; ds[l] += mu[l] * inv_bt; }   /* reparam + KL */
            mlp_backward(&enc, params, g, xb, ea, ds, NULL, ta, tb);
        }
        g[P + 0] = (float)s_mse; g[P + 1] = (float)s_musq; g[P + 2] = (float)s_deps;
        free(ea); free(da); free(sa);
    }
    /* fixed-order sum over threads, closed-form terms (same algebra as the kernels' finalize) */
    float* G = grads ? grads : acc;            /* reuse thread 0's row when the caller wants no gradient */
    double smse = 0, smusq = 0, sdeps = 0;
    for (long i = 0; i < P; ++i) { double s = 0; for (int th = 0; th < nthreads; ++th) s += acc[(size_t)th * (P + 4) + i]; G[i] = (float)s; }
    for (int th = 0; th < nthreads; ++th) { const float* a = acc + (size_t)th * (P + 4); smse += a[P]; smusq += a[P + 1]; sdeps += a[P + 2]; }
    double klc = 0;
    for (int l = 0; l < L; ++l) {
        const float lv = params[off_epsp + l];
        klc += 1.0 + lv - exp(lv);
        G[off_epsp + l] = 0.5f * expf(0.5f * lv) * G[off_epsp + l] - 0.5f * (1.f - expf(lv)) * ((float)B * inv_bt);
    }
    if (c->tdv) G[off_eps] = c->eps_cli * (float)((sdeps + 0.5 * B * D) * inv_bt);      /* eps = param * eps_cli, :71 */
    const float dkl = (float)((0.5 * smusq - 0.5 * B * klc) * inv_bt);
    const float mse = (float)((smse + 0.5 * B * D * (1.8378770664093453 + eps)) * inv_bt);
    const float loss = dkl + mse;
    G[P] = loss; G[P + 1] = dkl; G[P + 2] = mse; G[P + 3] = 0.f;
    if (apply) {
        const float bc1 = (float)(1.0 - pow(0.9, t)), bc2 = (float)(1.0 - pow(0.999, t));
        for (long i = 0; i < P; ++i) {
            m[i] = 0.9f * m[i] + (float)(1.0 - 0.9) * G[i];
            v[i] = 0.999f * v[i] + (float)(1.0 - 0.999) * G[i] * G[i];
            params[i] -= lr * (m[i] / bc1) / (sqrtf(v[i] / bc2) + 1e-8f);
        }
    }
    free(acc);
    return loss;
}
