/*
 * lccrf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lccrf_oracle.h).
 *
 * Scalar restatement of the reference's CPU dense-CRF path.  Build with
 *   gcc -O2 -std=c11 -ffp-contract=off   (no -march, no -ffast-math)
 * so that every a*b+c is two roundings, exactly like the reference's SSE2
 * build (CMakeLists.txt:10-11 has no -march/-mfma).
 *
 * "ref:" comments give the reference file:line being restated, relative to
 * /root/reference/Thirdparty/DenseCRF/include/ unless a directory is given.
 */
#include "lccrf_oracle.h"

#include <limits.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* hash table: key -> dense id in first-insertion order                      */
/* ref: permutohedral_cpu.h:66-167 (HashTableCPU)                            */
/* ------------------------------------------------------------------------ */
typedef struct {
    size_t d, filled, cap;
    int16_t *keys;   /* [filled*d] */
    size_t keys_cap; /* in keys    */
    int *slot;       /* [cap], -1 = empty */
} orc_hash;

/* ref: permutohedral_cpu.h:104-111.  k[i] is sign-extended into size_t. */
static size_t orc_hash_of(const int16_t *k, size_t d)
{
    size_t r = 0;
    for (size_t i = 0; i < d; i++) {
        r += (size_t)(long)k[i];
        r *= 1664525u;
    }
    return r;
}

static int orc_hash_setup(orc_hash *h, int d, int n_elements)
{
    h->d = (size_t)d;
    h->filled = 0;
    h->cap = 2 * (size_t)n_elements;          /* ref: :114 */
    if (h->cap < 8) h->cap = 8;               /* the reference would spin on cap==0 */
    h->keys_cap = h->cap / 2 + 10;
    h->keys = (int16_t *)malloc(h->keys_cap * h->d * sizeof(int16_t));
    h->slot = (int *)malloc(h->cap * sizeof(int));
    if (!h->keys || !h->slot) return -1;
    memset(h->slot, -1, h->cap * sizeof(int));
    return 0;
}

static void orc_hash_release(orc_hash *h)
{
    free(h->keys);
    free(h->slot);
    h->keys = NULL;
    h->slot = NULL;
}

/* ref: :79-102.  Doubling keeps ids; only the probe table is rebuilt. */
static int orc_hash_grow(orc_hash *h)
{
    size_t new_cap = h->cap * 2;
    int *ns = (int *)malloc(new_cap * sizeof(int));
    int16_t *nk = (int16_t *)realloc(h->keys, (new_cap / 2 + 10) * h->d * sizeof(int16_t));
    if (!ns || !nk) { free(ns); return -1; }
    h->keys = nk;
    h->keys_cap = new_cap / 2 + 10;
    memset(ns, -1, new_cap * sizeof(int));
    for (size_t i = 0; i < h->cap; i++) {
        int e = h->slot[i];
        if (e < 0) continue;
        size_t p = orc_hash_of(h->keys + (size_t)e * h->d, h->d) % new_cap;
        while (ns[p] >= 0) p = (p + 1 == new_cap) ? 0 : p + 1;
        ns[p] = e;
    }
    free(h->slot);
    h->slot = ns;
    h->cap = new_cap;
    return 0;
}

/* ref: :134-161.  Returns the id, -1 if absent and !create, -2 on OOM. */
static int orc_hash_find(orc_hash *h, const int16_t *k, int create)
{
    if (2 * h->filled >= h->cap)              /* ref: :135, also on lookups */
        if (orc_hash_grow(h)) return -2;
    size_t p = orc_hash_of(k, h->d) % h->cap;
    for (;;) {
        int e = h->slot[p];
        if (e == -1) {
            if (!create) return -1;
            memcpy(h->keys + h->filled * h->d, k, h->d * sizeof(int16_t));
            h->slot[p] = (int)h->filled;
            return (int)h->filled++;
        }
        if (memcmp(h->keys + (size_t)e * h->d, k, h->d * sizeof(int16_t)) == 0)
            return e;
        p = (p + 1 == h->cap) ? 0 : p + 1;
    }
}

/* ------------------------------------------------------------------------ */
/* lattice construction                                                      */
/* ref: permutohedral_cpu.h:241-424 (the SSE init; what g++ compiles on      */
/* x86-64 because __SSE__ is predefined and __SSE4_1__ is not, :37-52)        */
/* ------------------------------------------------------------------------ */

/* _mm_cvtps_epi32 under MXCSR round-to-nearest(-even), then back to float.
 * ref: :288-291,319.  Out-of-range / NaN give INT_MIN. */
static float orc_round_half_even(float v)
{
    int r;
    if (!(v >= -2147483648.0f && v < 2147483648.0f))
        r = INT_MIN;
    else
        r = (int)lrintf(v);   /* default FE_TONEAREST = ties to even */
    return (float)r;
}

/* float -> short as x86 does it for in-range values (cvttss2si, low 16 bits) */
static int16_t orc_to_short(float f)
{
    int i;
    if (!(f >= -2147483648.0f && f < 2147483648.0f))
        i = INT_MIN;
    else
        i = (int)f;
    return (int16_t)(uint16_t)((unsigned)i & 0xffffu);
}

#define ORC_MAX_D 30

int orc_lattice_init(orc_lattice *lat, const float *feature, int d, int N)
{
    memset(lat, 0, sizeof(*lat));
    if (d < 1 || d > ORC_MAX_D || N < 0) return -1;
    const int D1 = d + 1;
    const int Npad = (N + 3) & ~3;             /* blocks of four, ref: :248,294 */
    lat->N = N;
    lat->Npad = Npad;
    lat->d = d;

    lat->offset = (int *)calloc((size_t)(Npad > 0 ? Npad : 1) * D1, sizeof(int));
    lat->bary = (float *)calloc((size_t)(Npad > 0 ? Npad : 1) * D1, sizeof(float));
    if (!lat->offset || !lat->bary) { orc_lattice_free(lat); return -2; }

    orc_hash ht;
    if (orc_hash_setup(&ht, d, N)) { orc_lattice_free(lat); return -2; }   /* ref: :246 */

    /* constants, ref: :249-250,282-285 (quirk Q4) */
    const float inv_dp1 = 1.0f / (float)D1;
    const float dp1 = (float)D1;
    const float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (double)D1);
    float scale[ORC_MAX_D];
    for (int i = 0; i < d; i++)
        scale[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * (double)inv_std_dev);

    /* canonical simplex, ref: :274-279 */
    int16_t canon[(ORC_MAX_D + 1) * (ORC_MAX_D + 1)];
    for (int i = 0; i <= d; i++) {
        for (int j = 0; j <= d - i; j++) canon[i * D1 + j] = (int16_t)i;
        for (int j = d - i + 1; j <= d; j++) canon[i * D1 + j] = (int16_t)(i - D1);
    }

    float f[ORC_MAX_D], el[ORC_MAX_D + 1], rem0[ORC_MAX_D + 1], rank[ORC_MAX_D + 1];
    float b[ORC_MAX_D + 2];
    int16_t key[ORC_MAX_D + 1];

    for (int n = 0; n < Npad; n++) {
        /* lanes past N are fed feature 0.0 and still hashed, ref: :299 (quirk Q1) */
        for (int j = 0; j < d; j++)
            f[j] = n < N ? feature[(size_t)n * d + j] : 0.0f;

        /* elevate, ref: :304-310 */
        float sm = 0.0f;
        for (int j = d; j > 0; j--) {
            float cf = f[j - 1] * scale[j - 1];
            el[j] = sm - (float)j * cf;
            sm += cf;
        }
        el[0] = sm;

        /* nearest remainder-0 point, ref: :313-323 */
        float sum = 0.0f;
        for (int i = 0; i <= d; i++) {
            float v = orc_round_half_even(inv_dp1 * el[i]);
            rem0[i] = v * dp1;
            sum += v;
        }

        /* rank by strict float '<', ref: :326-336 (quirk Q3) */
        for (int i = 0; i <= d; i++) rank[i] = 0.0f;
        for (int i = 0; i < d; i++) {
            float di = el[i] - rem0[i];
            for (int j = i + 1; j <= d; j++) {
                float dj = el[j] - rem0[j];
                float c = (di < dj) ? 1.0f : 0.0f;
                rank[i] += c;
                rank[j] += 1.0f - c;
            }
        }

        /* bring off-plane points back, ref: :339-345 */
        for (int i = 0; i <= d; i++) {
            rank[i] += sum;
            float add = (rank[i] < 0.0f) ? dp1 : 0.0f;
            float sub = (rank[i] >= dp1) ? dp1 : 0.0f;
            float adj = add - sub;
            rank[i] += adj;
            rem0[i] += adj;
        }

        /* barycentric weights, ref: :348-366 */
        for (int i = 0; i <= d + 1; i++) b[i] = 0.0f;
        for (int i = 0; i <= d; i++) {
            float v = (el[i] - rem0[i]) * inv_dp1;
            int p = (int)((float)d - rank[i]);
            b[p] += v;
            b[p + 1] -= v;
        }
        b[0] += 1.0f + b[d + 1];

        /* the d+1 enclosing vertices, ref: :371-377 */
        for (int rem = 0; rem <= d; rem++) {
            for (int i = 0; i < d; i++)
                key[i] = orc_to_short(rem0[i] + (float)canon[rem * D1 + (int)rank[i]]);
            int id = orc_hash_find(&ht, key, 1);
            if (id < 0) { orc_hash_release(&ht); orc_lattice_free(lat); return -2; }
            lat->offset[(size_t)n * D1 + rem] = id;
            lat->bary[(size_t)n * D1 + rem] = b[rem];
        }
    }

    const int V = (int)ht.filled;              /* ref: :398 */
    lat->V = V;
    lat->keys = (int16_t *)malloc(((size_t)V * d + 1) * sizeof(int16_t));
    lat->nbr = (int *)malloc(((size_t)D1 * V * 2 + 1) * sizeof(int));
    if (!lat->keys || !lat->nbr) { orc_hash_release(&ht); orc_lattice_free(lat); return -2; }
    memcpy(lat->keys, ht.keys, (size_t)V * d * sizeof(int16_t));

    /* blur neighbours along each of the d+1 axes, ref: :408-421 */
    int16_t n1[ORC_MAX_D + 1], n2[ORC_MAX_D + 1];
    for (int j = 0; j <= d; j++) {
        for (int i = 0; i < V; i++) {
            const int16_t *k = lat->keys + (size_t)i * d;
            for (int t = 0; t < d; t++) {
                n1[t] = (int16_t)(k[t] - 1);
                n2[t] = (int16_t)(k[t] + 1);
            }
            if (j < d) {   /* for j == d the reference writes past the compared prefix */
                n1[j] = (int16_t)(k[j] + d);
                n2[j] = (int16_t)(k[j] - d);
            }
            int a = orc_hash_find(&ht, n1, 0);
            int c = orc_hash_find(&ht, n2, 0);
            lat->nbr[((size_t)j * V + i) * 2 + 0] = a;
            lat->nbr[((size_t)j * V + i) * 2 + 1] = c;
        }
    }
    orc_hash_release(&ht);
    return 0;
}

void orc_lattice_free(orc_lattice *lat)
{
    free(lat->offset);
    free(lat->bary);
    free(lat->nbr);
    free(lat->keys);
    memset(lat, 0, sizeof(*lat));
}

/* ------------------------------------------------------------------------ */
/* splat / blur / slice                                                      */
/* ref: permutohedral_cpu.h:634-699 (compute(float*,...), SSE variant).      */
/* The reference pads value_size to a multiple of 4 lanes; the padding lanes */
/* only ever hold zeros, so they are not materialised here.                  */
/* in == out is allowed (splat finishes before slice writes), as the         */
/* reference relies on at pairwise3d.h:24.                                   */
/* ------------------------------------------------------------------------ */
void orc_lattice_compute(const orc_lattice *lat, float *out, const float *in, int vs)
{
    const int D1 = lat->d + 1, V = lat->V, N = lat->N;
    const size_t n_val = (size_t)(V + 2) * vs;
    float *val = (float *)calloc(n_val, sizeof(float));     /* slot 0 = "absent" */
    float *nxt = (float *)calloc(n_val, sizeof(float));

    /* splat in (point, corner) order, ref: :653-661 (quirk Q6) */
    for (int i = 0; i < N; i++) {
        const float *x = in + (size_t)i * vs;
        for (int j = 0; j < D1; j++) {
            size_t o = (size_t)(lat->offset[(size_t)i * D1 + j] + 1) * vs;
            float w = lat->bary[(size_t)i * D1 + j];
            for (int k = 0; k < vs; k++) val[o + k] += w * x[k];
        }
    }

    /* d+1 Jacobi blur passes, ref: :663-679 */
    for (int j = 0; j < D1; j++) {
        for (int i = 0; i < V; i++) {
            size_t a = (size_t)(lat->nbr[((size_t)j * V + i) * 2 + 0] + 1) * vs;
            size_t c = (size_t)(lat->nbr[((size_t)j * V + i) * 2 + 1] + 1) * vs;
            size_t o = (size_t)(i + 1) * vs;
            for (int k = 0; k < vs; k++)
                nxt[o + k] = val[o + k] + 0.5f * (val[a + k] + val[c + k]);
        }
        float *t = val; val = nxt; nxt = t;
    }

    /* ref: :681 */
    const float alpha = 1.0f / (1.0f + powf(2.0f, (float)(-lat->d)));

    /* slice, ref: :684-694; weight is (bary*alpha) first, then times value */
    for (int i = 0; i < N; i++) {
        float *y = out + (size_t)i * vs;
        for (int k = 0; k < vs; k++) y[k] = 0.0f;
        for (int j = 0; j < D1; j++) {
            size_t o = (size_t)(lat->offset[(size_t)i * D1 + j] + 1) * vs;
            float w = lat->bary[(size_t)i * D1 + j] * alpha;
            for (int k = 0; k < vs; k++) y[k] += w * val[o + k];
        }
    }
    free(val);
    free(nxt);
}

/* ------------------------------------------------------------------------ */
/* softmax with the reference's polynomial exp                               */
/* ref: densecrf3d.h:51-98                                                   */
/* ------------------------------------------------------------------------ */
static float orc_very_fast_exp(float x)   /* ref: densecrf3d.h:51-54 */
{
    return 1 - x * (0.9999999995f - x * (0.4999999206f - x * (0.1666653019f - x * (0.0416573475f
             - x * (0.0083013598f - x * (0.0013298820f - x * (0.0001413161f)))))));
}

float orc_fast_exp(float x)               /* ref: densecrf3d.h:55-67 (quirk Q5) */
{
    int less_zero = 1;
    if (x < 0) { less_zero = 0; x = -x; }
    if (x > 20) return 0;
    int mult = 0;
    while ((double)x > 0.69 * 2 * 2 * 2) { mult += 3; x /= 8.0f; }
    while ((double)x > 0.69 * 2 * 2)     { mult += 2; x /= 4.0f; }
    while ((double)x > 0.69)             { mult += 1; x /= 2.0f; }
    x = orc_very_fast_exp(x);
    while (mult) { mult--; x = x * x; }
    return less_zero ? 1 / x : x;
}

void orc_exp_and_normalize(float *out, const float *in, int N, int L, float scale, float relax)
{                                         /* ref: densecrf3d.h:70-98 */
    float *Vv = (float *)malloc((size_t)(L > 0 ? L : 1) * sizeof(float));
    for (int i = 0; i < N; i++) {
        const float *b = in + (size_t)i * L;
        float mx = scale * b[0];
        for (int j = 1; j < L; j++)
            if (mx < scale * b[j]) mx = scale * b[j];
        float tt = 0;
        for (int j = 0; j < L; j++) {
            Vv[j] = orc_fast_exp(scale * b[j] - mx);
            tt += Vv[j];
        }
        for (int j = 0; j < L; j++) Vv[j] /= tt;
        float *a = out + (size_t)i * L;
        for (int j = 0; j < L; j++) {
            if (relax == 1) a[j] = Vv[j];
            else a[j] = (1 - relax) * a[j] + relax * Vv[j];
        }
    }
    free(Vv);
}

/* ------------------------------------------------------------------------ */
/* CRF object                                                                */
/* ------------------------------------------------------------------------ */
orc_crf *orc_crf_create(int N, int L)     /* ref: densecrf3d.h:23-28 */
{
    if (N < 0 || L < 1) return NULL;
    orc_crf *c = (orc_crf *)calloc(1, sizeof(orc_crf));
    if (!c) return NULL;
    c->N = N;
    c->L = L;
    size_t n = (size_t)N * L + 1;
    c->unary = (float *)calloc(n, sizeof(float));
    c->current = (float *)calloc(n, sizeof(float));
    c->next = (float *)calloc(n, sizeof(float));
    c->tmp = (float *)calloc(n, sizeof(float));
    return c;
}

void orc_crf_destroy(orc_crf *c)          /* ref: densecrf3d.h:30-36, densecrf_base.h:41-45 */
{
    if (!c) return;
    for (int k = 0; k < c->K; k++) {
        orc_lattice_free(&c->pw[k]->lat);
        free(c->pw[k]->norm);
        free(c->pw[k]);
    }
    free(c->unary); free(c->current); free(c->next); free(c->tmp); free(c->map);
    free(c);
}

void orc_crf_set_unary(orc_crf *c, const float *unary)   /* ref: densecrf3d.h:41-43 */
{
    memcpy(c->unary, unary, sizeof(float) * (size_t)c->N * c->L);
}

/* ref: densecrf3d.h:107-130.  At the reference's call site (src/Tracking.cc:1921,
 * `using namespace std` in scope via include/Tracking.h:55) log(float) resolves
 * to the float overload, hence logf. */
void orc_crf_set_unary_from_label(orc_crf *c, const int16_t *label, const float *conf)
{
    const int L = c->L;
    float u_energy = -logf(1.0f / L);
    float *n_e = (float *)malloc(sizeof(float) * L);
    float *p_e = (float *)malloc(sizeof(float) * L);
    for (int i = 0; i < L; i++) {
        n_e[i] = -logf((1.0f - conf[i]) / (L - 1));
        p_e[i] = -logf(conf[i]);
    }
    for (int i = 0; i < c->N; i++) {
        int t = label[i];
        float *u = c->unary + (size_t)i * L;
        if (t == -1) {
            for (int m = 0; m < L; m++) u[m] = u_energy;
        } else {
            for (int m = 0; m < L; m++) u[m] = n_e[t];
            u[t] = p_e[t];
        }
    }
    free(n_e);
    free(p_e);
}

/* ref: pairwise3d.h:20-28 (ctor: lattice + normalisation) */
int orc_crf_add_pairwise(orc_crf *c, const float *features, int d, float w)
{
    if (c->K >= ORC_MAX_KERNELS) return -1;
    orc_pairwise *p = (orc_pairwise *)calloc(1, sizeof(orc_pairwise));
    if (!p) return -2;
    int rc = orc_lattice_init(&p->lat, features, d, c->N);
    if (rc) { free(p); return rc; }
    p->w = w;
    p->norm = (float *)malloc(sizeof(float) * (size_t)(c->N + 1));
    for (int i = 0; i < c->N; i++) p->norm[i] = 1.0f;
    orc_lattice_compute(&p->lat, p->norm, p->norm, 1);
    for (int i = 0; i < c->N; i++) p->norm[i] = 1.0f / (p->norm[i] + 1e-20f);
    c->pw[c->K++] = p;
    return 0;
}

void orc_crf_start_inference(orc_crf *c)  /* ref: densecrf_base.h:78-80 */
{
    orc_exp_and_normalize(c->current, c->unary, c->N, c->L, -1.0f, 1.0f);
}

void orc_crf_step_inference(orc_crf *c, float relax)   /* ref: densecrf_base.h:82-91 */
{
    const size_t n = (size_t)c->N * c->L;
    for (size_t i = 0; i < n; i++) c->next[i] = -c->unary[i];   /* densecrf3d.h:154-158 */
    for (int k = 0; k < c->K; k++) {                             /* pairwise3d.h:73-78 */
        const orc_pairwise *p = c->pw[k];
        orc_lattice_compute(&p->lat, c->tmp, c->current, c->L);
        size_t t = 0;
        for (int i = 0; i < c->N; i++)
            for (int j = 0; j < c->L; j++, t++)
                c->next[t] += p->w * p->norm[i] * c->tmp[t];
    }
    orc_exp_and_normalize(c->current, c->next, c->N, c->L, 1.0f, relax);
}

/* PairwisePotential::apply, ref: densecrf_base.h:18, pairwise3d.h:73-78: out += w * norm * compute(in) */
void orc_pairwise_apply(const orc_crf *c, int k, float *out, const float *in)
{
    const orc_pairwise *p = c->pw[k];
    float *tmp = (float *)malloc(sizeof(float) * ((size_t)c->N * c->L + 1));
    orc_lattice_compute(&p->lat, tmp, in, c->L);
    size_t t = 0;
    for (int i = 0; i < c->N; i++)
        for (int j = 0; j < c->L; j++, t++)
            out[t] += p->w * p->norm[i] * tmp[t];
    free(tmp);
}

void orc_crf_build_map(orc_crf *c)        /* ref: densecrf3d.h:136-151 */
{
    if (!c->map) c->map = (int16_t *)malloc(sizeof(int16_t) * (size_t)(c->N + 1));
    for (int i = 0; i < c->N; i++) {
        const float *p = c->current + (size_t)i * c->L;
        float mx = p[0];
        int16_t imx = 0;
        for (int m = 1; m < c->L; m++)
            if (mx < p[m]) { mx = p[m]; imx = (int16_t)m; }
        c->map[i] = imx;
    }
}

void orc_crf_inference(orc_crf *c, int n_iter, int with_map, float relax)
{                                         /* ref: densecrf_base.h:65-73 */
    orc_crf_start_inference(c);
    for (int it = 0; it < n_iter; it++) orc_crf_step_inference(c, relax);
    if (with_map) orc_crf_build_map(c);
}

/* ------------------------------------------------------------------------ */
/* feature assembly                                                          */
/* ------------------------------------------------------------------------ */
void orc_appearance_features(int N, const float *vobserv, const float *verror,
                             float sd_observ, float sd_error, float *out)
{                                         /* ref: pairwise3d.h:37-48 */
    for (int i = 0; i < N; i++) {
        out[2 * i + 0] = vobserv[i] / sd_observ;
        out[2 * i + 1] = verror[i] / sd_error;
    }
}

void orc_smooth_features(int N, const float *xy, float sd2d, float *out)
{                                         /* ref: pairwise3d.h:51-71 (2-D branch) */
    for (int i = 0; i < N; i++) {
        out[2 * i + 0] = xy[2 * i + 0] / sd2d;
        out[2 * i + 1] = xy[2 * i + 1] / sd2d;
    }
}

void orc_image_features(int W, int H, float posdev, const uint8_t *rgb, int C,
                        float featuredev, float *out)
{                                         /* ref: pairwise_cpu.h:33-51 (FromImage) */
    const int F = 2 + (rgb ? C : 0);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t idx = (size_t)y * W + x;
            out[idx * F + 0] = (float)x / posdev;
            out[idx * F + 1] = (float)y / posdev;
            for (int i = 2; i < F; i++)
                out[idx * F + i] = (float)rgb[idx * C + (i - 2)] / featuredev;
        }
}

/* ------------------------------------------------------------------------ */
/* SLAM-side unary builder (PARITY UNPINNED, see header)                     */
/* ------------------------------------------------------------------------ */
void orc_default_params(orc_crf_params *p)   /* ref: Examples/RGB-D/TUM3.yaml:78-101 */
{
    p->w1 = 10.0f; p->w2 = 30.0f;
    p->u_alpha = 1.7f; p->stdev_alpha = 0.6f;
    p->u_beta = 5.4f; p->stdev_beta = 1.5f;
    p->u_gamma = 0.3f; p->stdev_gamma = 0.2f;
    p->point3d_stdev = 0.5f; p->point2d_stdev = 18.0f;
    p->u_depth = 2.75f; p->pth = 0.8f; p->confidence = 0.7f;
}

void orc_rough_classify(int N, const float *vobservs, const float *verrors,
                        const float *vdepths, const double *match_prob,
                        const orc_crf_params *p, int16_t *label)
{                                         /* ref: src/Tracking.cc:1961-2013 */
    float observ_sigma2 = p->stdev_beta * p->stdev_beta;
    float rpjerror_sigma2 = p->stdev_alpha * p->stdev_alpha;
    float depth_sigma2 = p->point3d_stdev * p->point3d_stdev;
    for (int i = 0; i < N; i++) {
        float k1 = (vobservs[i] - p->u_beta) * (vobservs[i] - p->u_beta) / (2 * observ_sigma2);
        float k2 = (verrors[i] - p->u_alpha) * (verrors[i] - p->u_alpha) / (2 * rpjerror_sigma2);
        float k3 = (vdepths[i] - p->u_depth) * (vdepths[i] - p->u_depth) / (2 * depth_sigma2);
        float p1 = expf(-k1), p2 = expf(-k2), p3 = expf(-k3);
        if (!match_prob) {
            label[i] = (p1 + p2 + p3 <= p->pth) ? 0 : 1;                 /* :1996 */
        } else {
            double p4 = match_prob[i];
            label[i] = ((double)(p1 + p2 + p3) + p4 <= (double)p->pth + 0.2) ? 0 : 1;  /* :2004 */
        }
    }
}

void orc_map_point_err_observ(int n_obs, const float *Xw, const float *poses,
                              const float *intr, const float *bounds,
                              const double *kp, int *observs, float *error, float *depth)
{                                         /* ref: src/Tracking.cc:1803-1839 */
    *observs = n_obs;
    if (n_obs == 0) return;
    float err = *error, dep = *depth;     /* accumulators come in zeroed, :1854-1855 */
    for (int o = 0; o < n_obs; o++) {
        const float *P = poses + (size_t)o * 12;
        float xc3[3];
        for (int r = 0; r < 3; r++) {
            /* `Rcw * x3Dw + tcw` is one cv::gemm(A,B,1,C,1).  Third-party dependency, absent from
             * /root/reference: OpenCV 3.x (CMakeLists.txt:31), modules/core/src/matmul.cpp, the
             * small-matrix path taken when 2 <= len <= 4 (here CV_32F, len == 3 == d_size.height,
             * d_size.width == 1): t = a0*b0 + a1*b1 + a2*b2 in float, d = (float)(t*alpha + c*beta)
             * with double alpha = beta = 1. */
            float t = P[r * 4 + 0] * Xw[0] + P[r * 4 + 1] * Xw[1] + P[r * 4 + 2] * Xw[2];
            xc3[r] = (float)((double)t * 1.0 + (double)P[r * 4 + 3] * 1.0);
        }
        float xc = xc3[0], yc = xc3[1];
        float invzc = (float)(1.0 / (double)xc3[2]);                     /* :1821 */
        if (invzc < 0) continue;
        float u = intr[o * 4 + 0] * xc * invzc + intr[o * 4 + 2];
        float v = intr[o * 4 + 1] * yc * invzc + intr[o * 4 + 3];
        if (u < bounds[o * 4 + 0] || u > bounds[o * 4 + 1] ||
            v < bounds[o * 4 + 2] || v > bounds[o * 4 + 3])
            continue;
        double dx = (double)u - kp[o * 2 + 0], dy = (double)v - kp[o * 2 + 1];
        float e = (float)sqrt(dx * dx + dy * dy);                        /* :1833 */
        err += e;
        dep += xc3[2];
    }
    *error = err / (float)n_obs;          /* divides by ALL observations, :1837-1838 */
    *depth = dep / (float)n_obs;
}

/* Whole-frame form of the two functions above, mirroring lccrf_unary_build's arguments
 * (include/lccrf.h section 3).  label -1 marks a point without observations, which the
 * reference drops at src/Tracking.cc:1858. */
void orc_unary_build(int n_points, const float *Xw, const int32_t *obs_ptr, const int32_t *obs_kf,
                     const double *obs_kp, const float *kf_pose, const float *kf_intr, const float *kf_bounds,
                     const double *match_prob, const orc_crf_params *p, float *observs, float *error,
                     float *depth, int16_t *label)
{
    for (int i = 0; i < n_points; i++) {
        const int o0 = obs_ptr[i], n = obs_ptr[i + 1] - obs_ptr[i];
        float *poses = (float *)malloc(sizeof(float) * 12 * (size_t)(n > 0 ? n : 1));
        float *intr = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
        float *bnd = (float *)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
        for (int o = 0; o < n; o++) {
            const int kf = obs_kf[o0 + o];
            memcpy(poses + 12 * o, kf_pose + 12 * (size_t)kf, sizeof(float) * 12);
            memcpy(intr + 4 * o, kf_intr + 4 * (size_t)kf, sizeof(float) * 4);
            memcpy(bnd + 4 * o, kf_bounds + 4 * (size_t)kf, sizeof(float) * 4);
        }
        int nobs = 0;
        float e = 0, d = 0;
        orc_map_point_err_observ(n, Xw + 3 * (size_t)i, poses, intr, bnd, obs_kp + 2 * (size_t)o0, &nobs, &e, &d);
        observs[i] = (float)nobs;
        error[i] = e;
        depth[i] = d;
        const double mp = match_prob ? match_prob[i] : 0.0;
        orc_rough_classify(1, &observs[i], &error[i], &depth[i], match_prob ? &mp : NULL, p, &label[i]);
        if (n == 0) label[i] = -1;
        free(poses); free(intr); free(bnd);
    }
}

/* Tracking::BfMatch, ref: src/Tracking.cc:1747-1766.  cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) is a
 * third-party dependency absent from /root/reference (OpenCV 3.x): restated from its published
 * behaviour -- cv::batchDistance scans the train set in index order and inserts a candidate into the
 * ascending top-K list only where it is strictly smaller, so ties keep the lower train index first. */
void orc_bf_match(int n_query, const uint8_t *desc_query, int n_train, const uint8_t *desc_train, double ratio,
                  int32_t *train_of_query, int32_t *n_matches)
{
    int nm = 0;
    for (int q = 0; q < n_query; q++) {
        int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
        for (int t = 0; t < n_train; t++) {
            int d = 0;
            for (int b = 0; b < 32; b++) d += __builtin_popcount((unsigned)(desc_query[32 * (size_t)q + b] ^ desc_train[32 * (size_t)t + b]));
            if (d < d1) {                                 /* insert before the first strictly greater element */
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = t; }
                else { d1 = d; i1 = t; }
            }
        }
        (void)i1;
        int m = -1;
        if (n_train >= 2 && (double)(float)d0 < (double)(float)d1 * ratio) m = i0;    /* :1755, match.size() == 2 */
        train_of_query[q] = m;
        nm += m >= 0;
    }
    if (n_matches) *n_matches = nm;
}

/* ======================================================================== */
/* Optimizer::PoseOptimization (src/Optimizer.cc:239-450): the step right    */
/* after the CRF (src/Tracking.cc:1002).  PARITY UNPINNED: the reference     */
/* runs it on g2o (Thirdparty/g2o, needs Eigen 3, absent in this image) and  */
/* holds no test or fixture for it.  This restates, in double precision and  */
/* operation by operation,                                                   */
/*   - the two pose-only edges, types_six_dof_expmap.{h,cpp}:153-196,266-364 */
/*     (the stereo projection's `const float invz = 1.0f/z` included),       */
/*   - RobustKernelHuber, robust_kernel_impl.cpp:78-91, and the first-order  */
/*     robustified quadratic form, base_unary_edge.hpp:43-72,                */
/*   - OptimizationAlgorithmLevenberg::solve, optimization_algorithm_        */
/*     levenberg.cpp:33-150 (tau 1e-5, gain ratio, 10 trials, Raul's stop),  */
/*   - SE3Quat::exp / operator* / map, se3quat.h:104-110,214-256 with        */
/*     Eigen's published quaternion <-> matrix conversions,                  */
/*   - the 4 x 10 schedule with chi2 re-classification, Optimizer.cc:366-440.*/
/* What is NOT reproduced bit for bit: Eigen's pivoted LDLT of the 6x6       */
/* system (a plain LDL^T here).  Checked by known answers (Jacobians against */
/* finite differences, exact recovery of a known pose) -- see tests.         */
/* ======================================================================== */
typedef struct { double w, x, y, z; } pq_t;

static void pq_normalize(pq_t *q)                     /* se3quat.h:280-285 */
{
    if (q->w < 0) { q->w = -q->w; q->x = -q->x; q->y = -q->y; q->z = -q->z; }
    const double n = sqrt(q->x * q->x + q->y * q->y + q->z * q->z + q->w * q->w);
    q->w /= n; q->x /= n; q->y /= n; q->z /= n;
}

static pq_t pq_from_matrix(const double m[3][3])      /* Eigen::Quaterniond(Matrix3d) */
{
    pq_t q;
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[2][1] - m[1][2]) * t;
        q.y = (m[0][2] - m[2][0]) * t;
        q.z = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double v[3];
        t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[k][j] - m[j][k]) * t;
        v[j] = (m[j][i] + m[i][j]) * t;
        v[k] = (m[k][i] + m[i][k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

static void pq_rotate(const pq_t *q, const double v[3], double out[3])   /* Eigen quaternion * vector */
{
    double uv[3] = {q->y * v[2] - q->z * v[1], q->z * v[0] - q->x * v[2], q->x * v[1] - q->y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q->w * uv[0] + (q->y * uv[2] - q->z * uv[1]);
    out[1] = v[1] + q->w * uv[1] + (q->z * uv[0] - q->x * uv[2]);
    out[2] = v[2] + q->w * uv[2] + (q->x * uv[1] - q->y * uv[0]);
}

static pq_t pq_mul(const pq_t *a, const pq_t *b)      /* Eigen quaternion product */
{
    pq_t r;
    r.w = a->w * b->w - a->x * b->x - a->y * b->y - a->z * b->z;
    r.x = a->w * b->x + a->x * b->w + a->y * b->z - a->z * b->y;
    r.y = a->w * b->y + a->y * b->w + a->z * b->x - a->x * b->z;
    r.z = a->w * b->z + a->z * b->w + a->x * b->y - a->y * b->x;
    return r;
}

/* estimate <- SE3Quat::exp(update) * estimate     (VertexSE3Expmap::oplusImpl, se3quat.h:214-256,104-110) */
static void pose_oplus(const double upd[6], pq_t *q, double t[3])
{
    const double om[3] = {upd[0], upd[1], upd[2]}, up[3] = {upd[3], upd[4], upd[5]};
    const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
    const double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
    double O2[3][3], R[3][3], V[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
    if (theta < 0.00001) {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { R[i][j] = ((i == j) ? 1.0 : 0.0) + O[i][j] + O2[i][j]; V[i][j] = R[i][j]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                const double I = (i == j) ? 1.0 : 0.0;
                R[i][j] = I + a * O[i][j] + b * O2[i][j];
                V[i][j] = I + b * O[i][j] + c * O2[i][j];
            }
    }
    pq_t dq = pq_from_matrix(R);
    pq_normalize(&dq);
    const double dt[3] = {V[0][0] * up[0] + V[0][1] * up[1] + V[0][2] * up[2], V[1][0] * up[0] + V[1][1] * up[1] + V[1][2] * up[2],
                          V[2][0] * up[0] + V[2][1] * up[1] + V[2][2] * up[2]};
    double rt[3];
    pq_rotate(&dq, t, rt);
    t[0] = dt[0] + rt[0]; t[1] = dt[1] + rt[1]; t[2] = dt[2] + rt[2];
    *q = pq_mul(&dq, q);
    pq_normalize(q);
}

typedef struct {
    int n;
    const float *Xw, *kp, *ur, *is2;
    double fx, fy, cx, cy, bf;
} pose_prob;

/* _error of edge i at pose (q,t) and chi2 = e^T (invSigma2 I) e; types_six_dof_expmap.h:153-157,184-188 */
static double pose_edge_error(const pose_prob *p, int i, const pq_t *q, const double t[3], double e[3], double pc[3])
{
    const double X[3] = {p->Xw[3 * i], p->Xw[3 * i + 1], p->Xw[3 * i + 2]};
    pq_rotate(q, X, pc);
    pc[0] += t[0]; pc[1] += t[1]; pc[2] += t[2];
    const double w = p->is2[i];
    if (p->ur[i] < 0) {                                   /* mono: project2d in double, cpp:290-296 */
        e[0] = (double)p->kp[2 * i] - ((pc[0] / pc[2]) * p->fx + p->cx);
        e[1] = (double)p->kp[2 * i + 1] - ((pc[1] / pc[2]) * p->fy + p->cy);
        e[2] = 0.0;
        return e[0] * (w * e[0]) + e[1] * (w * e[1]);
    }
    const float invz = 1.0f / (float)pc[2];              /* cpp:299-306: `const float invz = 1.0f/trans_xyz[2]` */
    const double u = pc[0] * invz * p->fx + p->cx;
    e[0] = (double)p->kp[2 * i] - u;
    e[1] = (double)p->kp[2 * i + 1] - (pc[1] * invz * p->fy + p->cy);
    e[2] = (double)p->ur[i] - (u - p->bf * invz);
    return e[0] * (w * e[0]) + e[1] * (w * e[1]) + e[2] * (w * e[2]);
}

static void huber(double e, double delta, double rho[3])   /* robust_kernel_impl.cpp:78-91 */
{
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else { const double s = sqrt(e); rho[0] = 2 * s * delta - dsqr; rho[1] = delta / s; rho[2] = -0.5 * rho[1] / e; }
}

/* Jacobian rows of edge i at camera-frame point pc; cpp:266-288,335-364 */
static int pose_edge_jacobian(const pose_prob *p, int i, const double pc[3], double J[3][6])
{
    const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz;
    J[0][0] = x * y * invz_2 * p->fx;          J[0][1] = -(1 + (x * x * invz_2)) * p->fx; J[0][2] = y * invz * p->fx;
    J[0][3] = -invz * p->fx;                   J[0][4] = 0;                               J[0][5] = x * invz_2 * p->fx;
    J[1][0] = (1 + y * y * invz_2) * p->fy;    J[1][1] = -x * y * invz_2 * p->fy;         J[1][2] = -x * invz * p->fy;
    J[1][3] = 0;                               J[1][4] = -invz * p->fy;                   J[1][5] = y * invz_2 * p->fy;
    if (p->ur[i] < 0) return 2;
    J[2][0] = J[0][0] - p->bf * y * invz_2;    J[2][1] = J[0][1] + p->bf * x * invz_2;    J[2][2] = J[0][2];
    J[2][3] = J[0][3];                         J[2][4] = 0;                               J[2][5] = J[0][5] - p->bf * invz_2;
    return 3;
}

/* solve (H + lambda I) x = b for a symmetric 6x6 H by LDL^T; 0 if not positive definite */
static int solve6(const double H[6][6], double lambda, const double b[6], double x[6])
{
    double L[6][6], D[6];
    for (int j = 0; j < 6; j++) {
        double d = H[j][j] + lambda;
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k] * D[k];
        if (!(d > 0)) return 0;
        D[j] = d;
        for (int i = j + 1; i < 6; i++) {
            double s = H[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k] * D[k];
            L[i][j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k] * y[k]; y[i] = s; }
    for (int i = 5; i >= 0; i--) { double s = y[i] / D[i]; for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k]; x[i] = s; }
    return 1;
}

/* computeActiveErrors + activeRobustChi2 (sparse_optimizer.cpp:61-114); edge_chi2[i] keeps what e->chi2() would return */
static double pose_active_chi2(const pose_prob *p, const uint8_t *level1, const uint8_t *valid, int robust, double dMono,
                               double dStereo, const pq_t *q, const double t[3], double *edge_chi2)
{
    double chi = 0.0, e[3], pc[3], rho[3];
    for (int i = 0; i < p->n; i++) {
        if (!valid[i] || level1[i]) continue;
        const double c = pose_edge_error(p, i, q, t, e, pc);
        edge_chi2[i] = c;
        if (robust) { huber(c, p->ur[i] < 0 ? dMono : dStereo, rho); chi += rho[0]; }
        else chi += c;
    }
    return chi;
}

int orc_pose_optimization(int n, const float *Xw, const float *kp, const float *u_right, const float *inv_sigma2,
                          const uint8_t *valid, const float *K4, float bf, const float *Tcw_in, float *Tcw_out,
                          uint8_t *outlier, int *n_initial)
{
    pose_prob P = {n, Xw, kp, u_right, inv_sigma2, K4[0], K4[1], K4[2], K4[3], bf};
    int n_init = 0;
    uint8_t *level1 = (uint8_t *)calloc((size_t)n + 1, 1);
    double *edge_chi2 = (double *)calloc((size_t)n + 1, sizeof(double));   /* chi2 of every edge's stored _error */
    for (int i = 0; i < n; i++) {
        if (valid[i]) { n_init++; outlier[i] = 0; }     /* Optimizer.cc:283-284 */
    }
    if (n_initial) *n_initial = n_init;
    memcpy(Tcw_out, Tcw_in, 16 * sizeof(float));
    if (n_init < 3) { free(level1); free(edge_chi2); return 0; }          /* Optimizer.cc:361-362 */
    double R0[3][3], t0[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R0[i][j] = Tcw_in[4 * i + j]; t0[i] = Tcw_in[4 * i + 3]; }
    pq_t q0 = pq_from_matrix(R0);                         /* Converter::toSE3Quat, Converter.cc:37-47 */
    pq_normalize(&q0);
    /* deltaMono / deltaStereo are `const float sqrt(5.991)` / `sqrt(7.815)` (Optimizer.cc:274-275) */
    const double dMono = (double)(float)sqrt(5.991), dStereo = (double)(float)sqrt(7.815);
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    pq_t q = q0;
    double t[3] = {t0[0], t0[1], t0[2]};
    int nBad = 0;
    for (int it = 0; it < 4; it++) {
        const int robust = it < 3;                        /* the kernel is dropped after the third round, :406-407 */
        q = q0; t[0] = t0[0]; t[1] = t0[1]; t[2] = t0[2]; /* vSE3->setEstimate(toSE3Quat(pFrame->mTcw)), :374 */
        double lambda = 0, ni = 2;
        int nBadLM = 0;
        for (int iter = 0; iter < 10; iter++) {           /* optimizer.optimize(10) */
            double H[6][6] = {{0}}, b[6] = {0}, currentChi = 0;
            for (int i = 0; i < n; i++) {
                if (!valid[i] || level1[i]) continue;
                double e[3], pc[3], J[3][6], rho[3] = {0, 1, 0};
                const double c = pose_edge_error(&P, i, &q, t, e, pc);
                edge_chi2[i] = c;
                if (robust) { huber(c, u_right[i] < 0 ? dMono : dStereo, rho); currentChi += rho[0]; }
                else currentChi += c;
                const int D = pose_edge_jacobian(&P, i, pc, J);
                const double w = inv_sigma2[i];
                for (int a = 0; a < 6; a++) {
                    double s = 0;
                    for (int r = 0; r < D; r++) s += J[r][a] * (w * e[r]);
                    b[a] -= rho[1] * s;                   /* base_unary_edge.hpp:62 */
                    for (int c2 = 0; c2 < 6; c2++) {
                        double h = 0;
                        for (int r = 0; r < D; r++) h += J[r][a] * ((rho[1] * w) * J[r][c2]);
                        H[a][c2] += h;                    /* :63 */
                    }
                }
            }
            const double iniChi = currentChi;
            if (iter == 0) {                              /* computeLambdaInit, levenberg.cpp:153-166 */
                double mx = 0;
                for (int j = 0; j < 6; j++) mx = fmax(fabs(H[j][j]), mx);
                lambda = 1e-5 * mx; ni = 2; nBadLM = 0;
            }
            double rho_gain = 0;
            int qmax = 0;
            do {
                pq_t qb = q;
                double tb[3] = {t[0], t[1], t[2]}, x[6] = {0, 0, 0, 0, 0, 0};
                const int ok2 = solve6(H, lambda, b, x);
                pose_oplus(x, &q, t);
                double tempChi = pose_active_chi2(&P, level1, valid, robust, dMono, dStereo, &q, t, edge_chi2);
                if (!ok2) tempChi = DBL_MAX;
                rho_gain = currentChi - tempChi;
                double scale = 0;
                for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
                scale += 1e-3;
                rho_gain /= scale;
                if (rho_gain > 0 && isfinite(tempChi)) {
                    double alpha = 1. - pow(2 * rho_gain - 1, 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                } else {
                    lambda *= ni; ni *= 2;
                    q = qb; t[0] = tb[0]; t[1] = tb[1]; t[2] = tb[2];
                }
                qmax++;
            } while (rho_gain < 0 && qmax < 10);
            if (qmax == 10 || rho_gain == 0) break;      /* Terminate */
            if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++; else nBadLM = 0;
            if (nBadLM >= 3) break;
        }
        nBad = 0;                                         /* Optimizer.cc:378-432 */
        for (int i = 0; i < n; i++) {
            if (!valid[i]) continue;
            double e[3], pc[3];
            /* an edge that sat the round out (flagged last time) gets a fresh computeError(); the others still hold the
             * _error of the LAST computeActiveErrors -- the last LM trial, even if that trial was rejected and the
             * estimate popped (Optimizer.cc:385-390: `if(pFrame->mvbOutlier[idx]) e->computeError();`) */
            if (outlier[i]) edge_chi2[i] = pose_edge_error(&P, i, &q, t, e, pc);
            const float chi2 = (float)edge_chi2[i];                           /* `const float chi2 = e->chi2()` */
            if (chi2 > (u_right[i] < 0 ? chi2Mono : chi2Stereo)) { outlier[i] = 1; level1[i] = 1; nBad++; }
            else { outlier[i] = 0; level1[i] = 0; }
        }
        if (n_init < 10) break;                           /* optimizer.edges().size() < 10, :434-435 */
    }
    /* to_homogeneous_matrix + Converter::toCvMat (float), Optimizer.cc:439-443 */
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z, twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x,
                 txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    const double R[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Tcw_out[4 * i + j] = (float)R[i][j]; Tcw_out[4 * i + 3] = (float)t[i]; }
    Tcw_out[12] = Tcw_out[13] = Tcw_out[14] = 0.0f; Tcw_out[15] = 1.0f;
    free(level1);
    free(edge_chi2);
    return n_init - nBad;
}
