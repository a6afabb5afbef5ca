// call_site_test.cpp -- the reference's call site (src/Tracking.cc:1919-1930) written against
// the drop-in adapter include/lccrf_densecrf.hpp, checked against the oracle's C API.
//
//   call_site_test <inputs.bin> [repetitions]     repetitions > 0: also time the call site (median, us)
//                                   inputs: int32 N, float obs[N], float err[N], float xy[2N], int16 label[N]
//
// Exit 0 and print "CALL-SITE OK ..." when labels are identical and Q is bit-identical.
// Exit 3 when the library reports that no GPU is usable (the adapter throws -- no fallback).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lccrf_densecrf.hpp"
#include "../../oracle/lccrf_oracle.h"      // the CHECKER; tests may link it, the product never does

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };

using namespace DenseCRF;
using namespace std;

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    int N = 0;
    if (fread(&N, 4, 1, fp) != 1) return 2;
    vector<float> vobservs(N), verrors(N);
    vector<Point2f> vcorrd2d(N);
    vector<Point3f> vpoints(N);
    vector<short> init_label(N);
    if (fread(vobservs.data(), 4, N, fp) != (size_t)N || fread(verrors.data(), 4, N, fp) != (size_t)N ||
        fread(vcorrd2d.data(), 8, N, fp) != (size_t)N || fread(init_label.data(), 2, N, fp) != (size_t)N)
        return 2;
    fclose(fp);

    // TUM3.yaml:78-101
    const float mConf = 0.7f, mW1 = 10.0f, mW2 = 30.0f, mObservStdev = 1.5f, mRpjErrorStdev = 0.6f,
                mPoint3dStdev = 0.5f, mPoint2dStdev = 18.0f;

    try {
        // ---- Tracking.cc:1919-1930 with the two type names swapped --------------------------
        const int M = 2;
        DenseCRFHIP<M> crf(N);
        crf.setUnaryEnergyFromLabel(init_label.data(), mConf);

        auto *appearancePairwise = PottsPotentialHIP<M, 2>::appearanceKernel(N, mW1, vobservs, verrors, mObservStdev, mRpjErrorStdev);
        crf.addPairwiseEnergy(appearancePairwise);

        auto *smoothnessPairwise = PottsPotentialHIP<M, 2>::smoothKernel(N, mW2, vpoints, vcorrd2d, mPoint3dStdev, mPoint2dStdev);
        crf.addPairwiseEnergy(smoothnessPairwise);

        crf.inference(5, true);
        short *res_label = crf.getMap();
        // ---------------------------------------------------------------------------------------

        // the same through the oracle
        orc_crf *o = orc_crf_create(N, M);
        float conf[M] = {mConf, mConf};
        orc_crf_set_unary_from_label(o, init_label.data(), conf);
        vector<float> f((size_t)2 * N + 1);
        orc_appearance_features(N, vobservs.data(), verrors.data(), mObservStdev, mRpjErrorStdev, f.data());
        orc_crf_add_pairwise(o, f.data(), 2, mW1);
        orc_smooth_features(N, &vcorrd2d[0].x, mPoint2dStdev, f.data());
        orc_crf_add_pairwise(o, f.data(), 2, mW2);
        orc_crf_inference(o, 5, 1, 1.0f);

        int bad_label = 0, bad_q = 0, n_dyn = 0;
        const float *q = crf.getProbability();
        for (int i = 0; i < N; ++i) {
            bad_label += res_label[i] != o->map[i];
            n_dyn += res_label[i] == 0;
        }
        bad_q = memcmp(q, o->current, sizeof(float) * (size_t)N * M) != 0;
        printf("%s N=%d dynamic=%d V=(%d,%d) label_mismatches=%d q_bit_identical=%d\n",
               (bad_label || bad_q) ? "CALL-SITE MISMATCH" : "CALL-SITE OK", N, n_dyn, crf.latticeSize(0),
               crf.latticeSize(1), bad_label, !bad_q);
        orc_crf_destroy(o);
        const int reps = argc > 2 ? atoi(argv[2]) : 0;
        if (reps > 0 && !(bad_label || bad_q)) {            // latency of the call site as a C++ caller sees it
            vector<double> us;
            for (int r = 0; r < reps + 5; ++r) {
                const auto t0 = chrono::steady_clock::now();
                {
                    DenseCRFHIP<M> c2(N);
                    c2.setUnaryEnergyFromLabel(init_label.data(), mConf);
                    c2.addPairwiseEnergy(PottsPotentialHIP<M, 2>::appearanceKernel(N, mW1, vobservs, verrors, mObservStdev, mRpjErrorStdev));
                    c2.addPairwiseEnergy(PottsPotentialHIP<M, 2>::smoothKernel(N, mW2, vpoints, vcorrd2d, mPoint3dStdev, mPoint2dStdev));
                    c2.inference(5, true);
                    volatile short first = c2.getMap()[0];
                    (void)first;
                }
                const auto t1 = chrono::steady_clock::now();
                if (r >= 5) us.push_back(chrono::duration<double, micro>(t1 - t0).count());
            }
            sort(us.begin(), us.end());
            printf("CALL-SITE LATENCY N=%d median %.1f us  min %.1f us  (%d frames, construct .. getMap .. destroy)\n", N,
                   us[us.size() / 2], us[0], reps);
        }
        return (bad_label || bad_q) ? 1 : 0;
    } catch (const std::exception &e) {
        printf("EXCEPTION: %s\n", e.what());
        return strstr(e.what(), "no HIP device") ? 3 : 4;
    }
}
