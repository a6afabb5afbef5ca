// lccrf_densecrf.hpp -- header-only C++ mirror of the reference's DenseCRF operator classes
// on top of the C-ABI (lccrf.h).  C++14, no dependencies beyond the standard library.
//
// Purpose: the reference's single call site, src/Tracking.cc:1919-1930,
//
//     DenseCRF3D<M> crf(N);
//     crf.setUnaryEnergyFromLabel(init_label, mConf);
//     crf.addPairwiseEnergy(PottsPotential3D<M,2>::appearanceKernel(N, mW1, vobservs, verrors, mObservStdev, mRpjErrorStdev));
//     crf.addPairwiseEnergy(PottsPotential3D<M,2>::smoothKernel(N, mW2, vpoints, vcorrd2d, mPoint3dStdev, mPoint2dStdev));
//     crf.inference(5, true);
//     short *res_label = crf.getMap();
//
// compiles against this header by changing the two type names to DenseCRFHIP / PottsPotentialHIP
// (see INTEGRATION.md).  Method names, argument meaning, defaults and ownership follow
//   densecrf_base.h:22-92   (DenseCRF)            densecrf3d.h:13-49   (DenseCRF3D<M>)
//   densecrf_base.h:12-19   (PairwisePotential)   pairwise3d.h:13-79   (PottsPotential3D<M,F>)
//
// The adapters DERIVE from the reference's two abstract classes.  With the reference tree on the include
// path (-I Thirdparty/DenseCRF/include) those are the reference's own densecrf_base.h, so
//   * a DenseCRF* / PairwisePotential* held by reference code can point at a DenseCRFHIP / PottsPotentialHIP;
//   * PottsPotentialHIP::apply() is the reference's pure virtual (densecrf_base.h:18) on the GPU -- it can be
//     added to the reference's CPU DenseCRF3D, or called on its own;
//   * the reference's (or a user's) own PairwisePotential subclasses can be added to a DenseCRFHIP: the
//     mean-field step then runs as densecrf_base.h:82-91 writes it -- stepInit, every potential's apply on the
//     host arrays, expAndNormalize -- with this library's share of each step on the GPU (lccrf_step_init,
//     lccrf_pairwise_apply, lccrf_exp_and_normalize) and the foreign potentials where their authors put them.
// Without the reference tree (this repository's own tests; -DLCCRF_NO_REFERENCE_BASES forces it) the same two
// shapes are declared below.  When every potential is ours, inference() is ONE kernel launch per frame.
//
// Differences, all deliberate:
//   * errors are loud: any non-zero lccrf status throws std::runtime_error (the reference has
//     no error path at all); there is no CPU fallback behind these classes;
//   * the unary energies live on the device: the base's unary_ stays null;
//   * getMap() is refreshed by inference(n, true) / buildMap(); getProbability() by this class's own (const)
//     getter.  DenseCRF::getProbability() is NOT virtual in the reference (densecrf_base.h:75): through a DenseCRF*
//     it returns the host buffer current_ as it stands.  An object that is going to be used that way keeps that
//     buffer fresh itself -- inference() / startInference() / stepInference() end with a download of Q (16 KB at
//     N = 2000) when syncThroughBase(true) is set; CreateDenseCRF<M>() sets it, because a base pointer is all its
//     caller holds.  Objects used by their own type (the call site of Tracking.cc:1920-1930) skip the download;
//   * a potential is a feature carrier until it is added or applied: the lattice is built on the GPU then
//     (the reference builds it in the potential's constructor).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "lccrf.h"

#if !defined(LCCRF_NO_REFERENCE_BASES) && defined(__has_include)
#if __has_include("densecrf_base.h")
#include "densecrf_base.h"
#define LCCRF_HAVE_REFERENCE_BASES 1
#endif
#endif

#ifndef LCCRF_HAVE_REFERENCE_BASES
namespace DenseCRF {
// The two abstract shapes of densecrf_base.h (:7-9, :12-19, :22-92), for builds without the reference tree.
enum Device { CPU, GPU };

class PairwisePotential {
protected:
    int N_;
public:
    explicit PairwisePotential(int N) : N_(N) {}
    virtual ~PairwisePotential() = default;
    virtual void apply(float *out_values, const float *in_values, float *tmp) const = 0;
};

class DenseCRF {
protected:
    int N_;
    float *unary_ = nullptr, *current_ = nullptr, *next_ = nullptr, *tmp_ = nullptr;
    short *map_ = nullptr;
    std::vector<PairwisePotential *> pairwise_;
    virtual void expAndNormalize(float *out, const float *in, float scale = 1.0, float relax = 1.0) = 0;
    virtual void buildMap() = 0;
    virtual void stepInit() = 0;
public:
    explicit DenseCRF(int N) : N_(N) {}
    virtual ~DenseCRF()
    {
        for (PairwisePotential *p : pairwise_) delete p;          // the CRF owns its terms
    }
    void addPairwiseEnergy(PairwisePotential *potential) { pairwise_.push_back(potential); }
    virtual void setUnaryEnergy(const float *unary) = 0;
    virtual void setUnaryEnergyFromLabel(const short *label, float *confidences) = 0;
    virtual void setUnaryEnergyFromLabel(const short *label, float confidence = 0.5) = 0;
    virtual void startInference() { expAndNormalize(current_, unary_, -1); }
    virtual void stepInference(float relax = 1.0)
    {
        stepInit();
        for (PairwisePotential *p : pairwise_) p->apply(next_, current_, tmp_);
        expAndNormalize(current_, next_, 1.0, relax);
    }
    virtual void inference(int n_iterations, bool with_map = false, float relax = 1.0)
    {
        startInference();
        for (int it = 0; it < n_iterations; ++it) stepInference(relax);
        if (with_map) buildMap();
    }
    short *getMap() const { return map_; }
    float *getProbability() const { return current_; }
};
}  // namespace DenseCRF
#endif

namespace DenseCRF {

inline void lccrf_check(int rc, const char *what)
{
    if (rc != LCCRF_OK)
        throw std::runtime_error(std::string(what) + ": lccrf status " + std::to_string(rc) + ": " + lccrf_last_error());
}

// What a DenseCRFHIP needs to know about a potential that runs on this library's kernels.
class HipPotential {
public:
    virtual ~HipPotential() = default;
    virtual int dims() const = 0;
    virtual float weight() const = 0;
    virtual const float *features() const = 0;      // [N][dims], already divided by the stdevs
    virtual void bind(lccrf_handle crf, int kernel) const = 0;   // from now on apply() is term `kernel` of that CRF
};
using PairwisePotentialHIP = HipPotential;          // (round-1 name)

// PottsPotential3D<M,F>, pairwise3d.h:13-79
template <int M, int F>
class PottsPotentialHIP : public PairwisePotential, public HipPotential {
protected:
    float w_;
    std::vector<float> feat_;
    mutable lccrf_handle h_ = nullptr;              // the CRF this term belongs to, or a private one-term CRF (own_)
    mutable int k_ = -1;
    mutable bool own_ = false;
    int device_id_ = 0;                             // where a potential that is applied on its own builds its private CRF
public:
    // pairwise3d.h:20 -- features are [N][F] AoS
    PottsPotentialHIP(const float *features, int N, float w, int device_id = 0)
        : PairwisePotential(N), w_(w), feat_(features, features + (size_t)N * F), device_id_(device_id) {}
    // (the factories hand their feature array over instead of having it copied: one 16 KB pass less per kernel and frame)
    PottsPotentialHIP(std::vector<float> &&features, int N, float w, int device_id = 0)
        : PairwisePotential(N), w_(w), feat_(std::move(features)), device_id_(device_id) {}
    PottsPotentialHIP(const PottsPotentialHIP &) = delete;
    ~PottsPotentialHIP() override
    {
        if (own_) lccrf_destroy(h_);
    }

    // GPU of the private one-term CRF behind a stand-alone apply(); irrelevant once the potential belongs to a DenseCRFHIP
    void setDevice(int device_id)
    {
        if (own_ && device_id != device_id_) {
            lccrf_destroy(h_);
            h_ = nullptr;
            own_ = false;
            k_ = -1;
        }
        device_id_ = device_id;
    }
    int device() const { return device_id_; }
    int dims() const override { return F; }
    float weight() const override { return w_; }
    const float *features() const override { return feat_.data(); }
    void bind(lccrf_handle crf, int kernel) const override
    {
        if (own_) lccrf_destroy(h_);
        own_ = false;
        h_ = crf;
        k_ = kernel;
    }

    // densecrf_base.h:18 / pairwise3d.h:73-78: out_values += w * norm * compute(in_values), both [N][M] on the host.
    // `tmp` is the reference's scratch for compute(); ours lives on the device.
    void apply(float *out_values, const float *in_values, float * /*tmp*/) const override
    {
        if (!h_) {                                   // not part of a DenseCRFHIP (yet): a private one-term CRF carries the lattice
            lccrf_check(lccrf_create(&h_, device_id_, N_, M), "lccrf_create");
            own_ = true;
            k_ = 0;
            lccrf_check(lccrf_add_pairwise(h_, feat_.data(), F, w_), "lccrf_add_pairwise");
        }
        lccrf_check(lccrf_pairwise_apply(h_, k_, out_values, in_values), "lccrf_pairwise_apply");
    }

    // pairwise3d.h:37-48
    template <class T = float>
    static PottsPotentialHIP<M, F> *appearanceKernel(int N, float weight, std::vector<float> &vobserv,
                                                     std::vector<float> &verror, float posdev1, float posdev2)
    {
        static_assert(F == 2, "the reference's appearance kernel has two features");
        std::vector<float> all((size_t)F * N);
        for (int idx = 0; idx < N; ++idx) {
            all[(size_t)idx * F + 0] = vobserv[idx] / posdev1;
            all[(size_t)idx * F + 1] = verror[idx] / posdev2;
        }
        return new PottsPotentialHIP<M, F>(std::move(all), N, weight);
    }

    // pairwise3d.h:51-71 -- only the 2-D branch is live in the reference; points3d is accepted
    // and ignored exactly as there.  P3 / P2 are any types with .x/.y (cv::Point3f / cv::Point2f).
    template <class P3, class P2>
    static PottsPotentialHIP<M, F> *smoothKernel(int N, float weight, std::vector<P3> & /*points3d*/,
                                                 std::vector<P2> &points2d, float /*posdev1*/, float posdev2)
    {
        static_assert(F == 2, "the reference's smoothness kernel has two features");
        std::vector<float> all((size_t)F * N);
        for (int idx = 0; idx < N; ++idx) {
            all[(size_t)idx * F + 0] = points2d[idx].x / posdev2;
            all[(size_t)idx * F + 1] = points2d[idx].y / posdev2;
        }
        return new PottsPotentialHIP<M, F>(std::move(all), N, weight);
    }
};

// ---- DenseCRF3D<M> ---------------------------------------------------------------------
template <int M>
class DenseCRFHIP : public DenseCRF {
protected:
    lccrf_handle h_ = nullptr;
    size_t adopted_ = 0;                  // pairwise_[0, adopted_) have been looked at
    int n_hip_ = 0;                       // ... of which this many are terms of the handle
    bool mixed_ = false;                  // some potential is not ours: host-array stepping (densecrf_base.h:82-91)
    bool base_sync_ = false;              // keep current_ fresh for readers that only hold a DenseCRF* (non-virtual getProbability)
    mutable std::vector<short> map_buf_;
    mutable std::vector<float> cur_buf_, next_buf_, tmp_buf_;

    // Potentials may also arrive through the base class's non-virtual addPairwiseEnergy (a DenseCRF* in reference
    // code): whatever is new in pairwise_ is taken in before anything runs.
    void adopt()
    {
        for (; adopted_ < pairwise_.size(); ++adopted_) {
            const HipPotential *hp = dynamic_cast<const HipPotential *>(pairwise_[adopted_]);
            if (hp) {
                lccrf_check(lccrf_add_pairwise(h_, hp->features(), hp->dims(), hp->weight()), "lccrf_add_pairwise");
                hp->bind(h_, n_hip_++);
            } else {
                mixed_ = true;
            }
        }
    }

    // DenseCRF's protected virtuals (densecrf_base.h:34-36), each one kernel on the GPU
    void expAndNormalize(float *out, const float *in, float scale = 1.0, float relax = 1.0) override   // densecrf3d.h:70-98
    {
        lccrf_check(lccrf_exp_and_normalize(h_, out, in, scale, relax), "lccrf_exp_and_normalize");
    }
    void stepInit() override                                                   // densecrf3d.h:154-158
    {
        lccrf_check(lccrf_step_init(h_, next_), "lccrf_step_init");
    }
    void buildMap() override                                                   // densecrf3d.h:136-151
    {
        if (mixed_) {
            lccrf_check(lccrf_map_of(h_, current_, map_), "lccrf_map_of");
        } else {
            lccrf_check(lccrf_build_map(h_), "lccrf_build_map");
            lccrf_check(lccrf_get_map(h_, map_), "lccrf_get_map");
        }
    }

public:
    explicit DenseCRFHIP(int N, int device_id = 0)                              // densecrf3d.h:23
        : DenseCRF(N), map_buf_((size_t)N + 1), cur_buf_((size_t)N * M + 1), next_buf_((size_t)N * M + 1), tmp_buf_((size_t)N * M + 1)
    {
        lccrf_check(lccrf_create(&h_, device_id, N, M), "lccrf_create");
        current_ = cur_buf_.data();       // the buffers the base class's getters hand out
        next_ = next_buf_.data();
        tmp_ = tmp_buf_.data();
        map_ = map_buf_.data();
    }
    ~DenseCRFHIP() override                                                    // densecrf3d.h:30; ~DenseCRF deletes the potentials
    {
        lccrf_destroy(h_);
        current_ = next_ = tmp_ = nullptr;
        map_ = nullptr;
    }
    DenseCRFHIP(DenseCRFHIP &) = delete;

    // densecrf_base.h:54 -- ownership of the potential moves to the CRF
    void addPairwiseEnergy(PairwisePotential *potential)
    {
        DenseCRF::addPairwiseEnergy(potential);
        adopt();
    }

    void setUnaryEnergy(const float *unary) override { lccrf_check(lccrf_set_unary(h_, unary), "lccrf_set_unary"); }

    void setUnaryEnergyFromLabel(const short *label, float *confidences) override    // densecrf3d.h:107
    {
        lccrf_check(lccrf_set_unary_from_label(h_, label, confidences), "lccrf_set_unary_from_label");
    }
    void setUnaryEnergyFromLabel(const short *label, float confidence = 0.5) override   // densecrf3d.h:100
    {
        float c[M];
        for (int i = 0; i < M; ++i) c[i] = confidence;
        setUnaryEnergyFromLabel(label, c);
    }

    void inference(int n_iterations, bool with_map = false, float relax = 1.0) override   // densecrf_base.h:65
    {
        adopt();
        if (mixed_) {
            DenseCRF::inference(n_iterations, with_map, relax);
            return;
        }
        lccrf_check(lccrf_inference(h_, n_iterations, with_map ? 1 : 0, relax), "lccrf_inference");
        if (with_map) lccrf_check(lccrf_get_map(h_, map_), "lccrf_get_map");
        if (base_sync_) syncProbability();
    }
    void startInference() override                                             // densecrf_base.h:78
    {
        adopt();
        if (mixed_) {
            lccrf_check(lccrf_step_init(h_, next_), "lccrf_step_init");       // next = -unary (the unaries live on the device) ...
            expAndNormalize(current_, next_, 1.0, 1.0);                         // ... so softmax(-unary) is expAndNormalize(next, scale 1)
        } else {
            lccrf_check(lccrf_start_inference(h_), "lccrf_start_inference");
            if (base_sync_) syncProbability();
        }
    }
    void stepInference(float relax = 1.0) override                             // densecrf_base.h:82
    {
        adopt();
        if (mixed_) {
            DenseCRF::stepInference(relax);
        } else {
            lccrf_check(lccrf_step_inference(h_, relax), "lccrf_step_inference");
            if (base_sync_) syncProbability();
        }
    }

    // densecrf_base.h:74-75.  Pointers into object-owned host buffers, valid until destruction.
    short *getMap() const { return map_; }
    float *getProbability() const
    {
        syncProbability();
        return current_;
    }
    // refresh the host copy of Q behind getProbability() (a no-op in mixed mode, where Q lives in current_ anyway)
    void syncProbability() const
    {
        if (!mixed_) lccrf_check(lccrf_get_probability(h_, current_), "lccrf_get_probability");
    }

    int latticeSize(int kernel)
    {
        adopt();
        int V = 0;
        lccrf_check(lccrf_get_lattice_size(h_, kernel, &V), "lccrf_get_lattice_size");
        return V;
    }
    // lccrf_set_option (include/lccrf.h), e.g. setOption(LCCRF_OPT_SINGLE_WORKGROUP, 1) for a tracker on a shared GPU
    void setOption(int option, int value) { lccrf_check(lccrf_set_option(h_, option, value), "lccrf_set_option"); }
    // see the header comment: true = every inference call ends with Q in the buffer DenseCRF::getProbability() returns
    void syncThroughBase(bool on) { base_sync_ = on; }
    bool syncsThroughBase() const { return base_sync_; }
    bool mixed() const { return mixed_; }
    lccrf_handle handle() const { return h_; }
};

// DenseCRF::Create<M>(N) is an empty stub in the reference (densecrf_base.h:47-51) next to its Device enum (:7-9);
// this is the factory it points at: the GPU implementation behind the abstract type.
template <int M>
inline DenseCRF *CreateDenseCRF(int N, Device device = GPU, int device_id = 0)
{
    if (device != GPU) throw std::runtime_error("CreateDenseCRF: this library has no CPU implementation (use the reference's DenseCRF3D)");
    DenseCRFHIP<M> *crf = new DenseCRFHIP<M>(N, device_id);
    crf->syncThroughBase(true);           // the caller holds a DenseCRF*: its getProbability() must see every result (densecrf_base.h:75)
    return crf;
}

}  // namespace DenseCRF
