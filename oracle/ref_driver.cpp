/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C-ABI shim around the REFERENCE's own header-only DenseCRF library, compiled
 * from the sources where they lie (/root/reference/Thirdparty/DenseCRF/include)
 * by oracle/Makefile into oracle/_ref/liblccrf_ref.so.  No reference source is
 * copied: this file only #includes the headers and instantiates their templates.
 *
 * It is used (a) to prove oracle/lccrf_oracle.c bit-identical to the reference,
 * (b) to generate the golden vectors in tests/golden/ (tests/golden/make_golden.py)
 * and (c) optionally as the "reference" CPU baseline in bench.py.
 *
 * What is and is not built from the reference:
 *   - densecrf_base.h, densecrf3d.h (DenseCRF3D<M>, the class src/Tracking.cc:1920
 *     uses), permutohedral_cpu.h, pairwise_cpu.h: need only libc, built unmodified.
 *     (densecrf_cpu.h redefines fast_exp, so the image example that uses it lives
 *     in its own translation unit, ref_driver_image.cpp.)
 *   - pairwise3d.h needs <opencv2/core/core.hpp> (absent in this image) for two
 *     POD point types, so it is NOT built.  Its ctor and apply() are textually the
 *     same as pairwise_cpu.h's PottsPotentialCPU (pairwise3d.h:20-28,73-78 vs
 *     pairwise_cpu.h:15-23,53-57); its two factories only divide features by a
 *     stdev (pairwise3d.h:37-71) and that division is done by the caller here.
 *
 * The translation unit mirrors the reference's call site: src/Tracking.cc sees
 * `using namespace std;` (include/Tracking.h:55) before it includes densecrf3d.h,
 * so the unqualified log(float) at densecrf3d.h:109-114 binds to the float overload.
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

using namespace std;

#include "densecrf3d.h"
#include "pairwise_cpu.h"

namespace {

using DenseCRF::PermutohedralLatticeCPU;

/* read-only window onto the lattice's protected members */
struct LatProbe : PermutohedralLatticeCPU {
    static int V(const PermutohedralLatticeCPU &l) { return static_cast<const LatProbe &>(l).M_; }
    static const int *offset(const PermutohedralLatticeCPU &l) { return static_cast<const LatProbe &>(l).offset_; }
    static const float *bary(const PermutohedralLatticeCPU &l) { return static_cast<const LatProbe &>(l).barycentric_; }
    static const int *nbr(const PermutohedralLatticeCPU &l) {
        return reinterpret_cast<const int *>(static_cast<const LatProbe &>(l).blur_neighbors_);
    }
};

template <int M, int F>
struct PotProbe : DenseCRF::PottsPotentialCPU<M, F> {
    PotProbe(const float *f, int N, float w) : DenseCRF::PottsPotentialCPU<M, F>(f, N, w) {}
    const float *norm() const { return this->norm_; }
    const PermutohedralLatticeCPU &lat() const { return this->lattice_; }
};

struct KernelView {
    const DenseCRF::PairwisePotential *pot;   /* owned by the CRF (densecrf_base.h:54) */
    int d, V;
    const float *norm;
    const int *offset;
    const float *bary;
    const int *nbr;
};

struct Handle {
    DenseCRF::DenseCRF *crf = nullptr;
    void (*build_map)(DenseCRF::DenseCRF *) = nullptr;
    int N = 0, L = 0;
    std::vector<KernelView> kernels;
};

/* buildMap() is protected (densecrf3d.h:17); this subclass only forwards to it */
template <int M>
struct CrfProbe : DenseCRF::DenseCRF3D<M> {
    explicit CrfProbe(int N) : DenseCRF::DenseCRF3D<M>(N) {}
    static void call_build_map(DenseCRF::DenseCRF *c) { static_cast<CrfProbe<M> *>(c)->buildMap(); }
};

template <int M>
void make_crf(Handle *h, int N)
{
    h->crf = new CrfProbe<M>(N);
    h->build_map = &CrfProbe<M>::call_build_map;
}

template <int M, int F>
int add_kernel(Handle *h, const float *feat, float w)
{
    auto *p = new PotProbe<M, F>(feat, h->N, w);
    KernelView kv;
    kv.pot = p;
    kv.d = F;
    kv.V = LatProbe::V(p->lat());
    kv.norm = p->norm();
    kv.offset = LatProbe::offset(p->lat());
    kv.bary = LatProbe::bary(p->lat());
    kv.nbr = LatProbe::nbr(p->lat());
    h->kernels.push_back(kv);
    h->crf->addPairwiseEnergy(p);   /* ownership moves to the CRF, densecrf_base.h:54 */
    return 0;
}

template <int M>
int add_kernel_d(Handle *h, const float *feat, int d, float w)
{
    switch (d) {
    case 1: return add_kernel<M, 1>(h, feat, w);
    case 2: return add_kernel<M, 2>(h, feat, w);
    case 3: return add_kernel<M, 3>(h, feat, w);
    case 4: return add_kernel<M, 4>(h, feat, w);
    case 5: return add_kernel<M, 5>(h, feat, w);
    case 6: return add_kernel<M, 6>(h, feat, w);
    default: return -1;
    }
}

}  // namespace

extern "C" {

void *ref_crf_create(int N, int L)
{
    Handle *h = new Handle;
    h->N = N;
    h->L = L;
    switch (L) {
    case 2: make_crf<2>(h, N); break;
    case 3: make_crf<3>(h, N); break;
    case 4: make_crf<4>(h, N); break;
    case 21: make_crf<21>(h, N); break;
    default: delete h; return nullptr;
    }
    return h;
}

void ref_crf_destroy(void *hv)
{
    Handle *h = static_cast<Handle *>(hv);
    if (!h) return;
    delete h->crf;
    delete h;
}

void ref_crf_set_unary(void *hv, const float *unary)
{
    static_cast<Handle *>(hv)->crf->setUnaryEnergy(unary);
}

void ref_crf_set_unary_from_label(void *hv, const int16_t *label, float *conf)
{
    static_cast<Handle *>(hv)->crf->setUnaryEnergyFromLabel(label, conf);
}

int ref_crf_add_pairwise(void *hv, const float *feat, int d, float w)
{
    Handle *h = static_cast<Handle *>(hv);
    switch (h->L) {
    case 2: return add_kernel_d<2>(h, feat, d, w);
    case 3: return add_kernel_d<3>(h, feat, d, w);
    case 4: return add_kernel_d<4>(h, feat, d, w);
    case 21: return add_kernel_d<21>(h, feat, d, w);
    default: return -1;
    }
}

void ref_crf_start_inference(void *hv) { static_cast<Handle *>(hv)->crf->startInference(); }
void ref_crf_step_inference(void *hv, float relax) { static_cast<Handle *>(hv)->crf->stepInference(relax); }
void ref_crf_inference(void *hv, int n_iter, int with_map, float relax)
{
    static_cast<Handle *>(hv)->crf->inference(n_iter, with_map != 0, relax);
}

void ref_crf_build_map(void *hv)
{
    Handle *h = static_cast<Handle *>(hv);
    h->build_map(h->crf);
}

const float *ref_crf_probability(void *hv) { return static_cast<Handle *>(hv)->crf->getProbability(); }
const int16_t *ref_crf_map(void *hv) { return static_cast<Handle *>(hv)->crf->getMap(); }

int ref_kernel_count(void *hv) { return (int)static_cast<Handle *>(hv)->kernels.size(); }
int ref_kernel_V(void *hv, int k) { return static_cast<Handle *>(hv)->kernels[k].V; }
const float *ref_kernel_norm(void *hv, int k) { return static_cast<Handle *>(hv)->kernels[k].norm; }
const int *ref_kernel_offset(void *hv, int k) { return static_cast<Handle *>(hv)->kernels[k].offset; }
const float *ref_kernel_bary(void *hv, int k) { return static_cast<Handle *>(hv)->kernels[k].bary; }
const int *ref_kernel_nbr(void *hv, int k) { return static_cast<Handle *>(hv)->kernels[k].nbr; }

/* PairwisePotential::apply (densecrf_base.h:18, pairwise_cpu.h:53-57 == pairwise3d.h:73-78): out += w * norm * compute(in) */
void ref_kernel_apply(void *hv, int k, float *out, const float *in)
{
    Handle *h = static_cast<Handle *>(hv);
    std::vector<float> tmp((size_t)h->N * h->L + 16);
    h->kernels[k].pot->apply(out, in, tmp.data());
}

/* bare lattice filter: out = compute(in) for one feature set (permutohedral_cpu.h:241,634) */
int ref_lattice_filter(const float *feat, int d, int N, const float *in, float *out, int value_size)
{
    PermutohedralLatticeCPU lat;
    lat.init(feat, d, N);
    lat.compute(out, in, value_size);
    return LatProbe::V(lat);
}

float ref_fast_exp(float x) { return DenseCRF::fast_exp(x); }

}  // extern "C"
