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
// Differences, all deliberate:
//   * errors are loud: any non-zero lccrf status throws std::runtime_error (the reference has
//     no error path at all); there is no CPU fallback behind these classes;
//   * getMap()/getProbability() return pointers into object-owned HOST buffers that are
//     refreshed on call (the reference returns its live internal arrays);
//   * a potential is a feature carrier until it is added: the lattice is built on the GPU when
//     the CRF takes ownership (the reference builds it in the potential's constructor).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "lccrf.h"

namespace DenseCRF {

inline void lccrf_check(int rc, const char *what)
{
    if (rc != LCCRF_OK)
        throw std::runtime_error(std::string(what) + ": lccrf status " + std::to_string(rc) + ": " + lccrf_last_error());
}

// ---- PairwisePotential ----------------------------------------------------------------
class PairwisePotentialHIP {
protected:
    int N_;
public:
    explicit PairwisePotentialHIP(int N) : N_(N) {}
    virtual ~PairwisePotentialHIP() = default;
    virtual int dims() const = 0;
    virtual float weight() const = 0;
    virtual const float *features() const = 0;      // [N][dims], already divided by the stdevs
};

// PottsPotential3D<M,F>, pairwise3d.h:13-79
template <int M, int F>
class PottsPotentialHIP : public PairwisePotentialHIP {
protected:
    float w_;
    std::vector<float> feat_;
public:
    // pairwise3d.h:20 -- features are [N][F] AoS
    PottsPotentialHIP(const float *features, int N, float w)
        : PairwisePotentialHIP(N), w_(w), feat_(features, features + (size_t)N * F) {}
    PottsPotentialHIP(const PottsPotentialHIP &) = delete;

    int dims() const override { return F; }
    float weight() const override { return w_; }
    const float *features() const override { return feat_.data(); }

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
        return new PottsPotentialHIP<M, F>(all.data(), N, weight);
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
        return new PottsPotentialHIP<M, F>(all.data(), N, weight);
    }
};

// ---- DenseCRF3D<M> ---------------------------------------------------------------------
template <int M>
class DenseCRFHIP {
protected:
    int N_;
    lccrf_handle h_;
    std::vector<short> map_;
    std::vector<float> prob_;
public:
    explicit DenseCRFHIP(int N, int device_id = 0) : N_(N), h_(nullptr)       // densecrf3d.h:23
    {
        lccrf_check(lccrf_create(&h_, device_id, N, M), "lccrf_create");
    }
    ~DenseCRFHIP() { lccrf_destroy(h_); }                                     // densecrf3d.h:30
    DenseCRFHIP(DenseCRFHIP &) = delete;

    // densecrf_base.h:54 -- ownership of the potential moves to the CRF
    void addPairwiseEnergy(PairwisePotentialHIP *potential)
    {
        const int rc = lccrf_add_pairwise(h_, potential->features(), potential->dims(), potential->weight());
        delete potential;
        lccrf_check(rc, "lccrf_add_pairwise");
    }

    void setUnaryEnergy(const float *unary) { lccrf_check(lccrf_set_unary(h_, unary), "lccrf_set_unary"); }

    void setUnaryEnergyFromLabel(const short *label, float *confidences)      // densecrf3d.h:107
    {
        lccrf_check(lccrf_set_unary_from_label(h_, label, confidences), "lccrf_set_unary_from_label");
    }
    void setUnaryEnergyFromLabel(const short *label, float confidence = 0.5)  // densecrf3d.h:100
    {
        float c[M];
        for (int i = 0; i < M; ++i) c[i] = confidence;
        setUnaryEnergyFromLabel(label, c);
    }

    void inference(int n_iterations, bool with_map = false, float relax = 1.0)   // densecrf_base.h:65
    {
        lccrf_check(lccrf_inference(h_, n_iterations, with_map ? 1 : 0, relax), "lccrf_inference");
    }
    void startInference() { lccrf_check(lccrf_start_inference(h_), "lccrf_start_inference"); }
    void stepInference(float relax = 1.0) { lccrf_check(lccrf_step_inference(h_, relax), "lccrf_step_inference"); }
    void buildMap() { lccrf_check(lccrf_build_map(h_), "lccrf_build_map"); }

    short *getMap()                                                            // densecrf_base.h:74
    {
        map_.resize((size_t)N_ + 1);
        lccrf_check(lccrf_get_map(h_, map_.data()), "lccrf_get_map");
        return map_.data();
    }
    float *getProbability()                                                    // densecrf_base.h:75
    {
        prob_.resize((size_t)N_ * M + 1);
        lccrf_check(lccrf_get_probability(h_, prob_.data()), "lccrf_get_probability");
        return prob_.data();
    }

    int latticeSize(int kernel)
    {
        int V = 0;
        lccrf_check(lccrf_get_lattice_size(h_, kernel, &V), "lccrf_get_lattice_size");
        return V;
    }
    lccrf_handle handle() const { return h_; }
};

}  // namespace DenseCRF
