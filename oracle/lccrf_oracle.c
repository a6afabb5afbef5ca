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
