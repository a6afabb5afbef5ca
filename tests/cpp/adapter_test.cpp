// adapter_test.cpp -- the adapter as a subclass of the reference's abstract bases (include/lccrf_densecrf.hpp):
//   1. used through DenseCRF* / PairwisePotential* base pointers (potentials added with the BASE's addPairwiseEnergy),
//      created by the factory the reference's Device enum / Create<M> stub point at;
//   2. PottsPotentialHIP::apply -- the reference's pure virtual -- on its own, before the potential joins any CRF;
//   3. a user-defined CPU PairwisePotential mixed with two of ours: the mean-field step of densecrf_base.h:82-91 with
//      this library's share on the GPU;
// each checked bit for bit against the oracle's C API (the CHECKER; tests may link it, the product never does).
//   adapter_test <inputs.bin>      inputs: int32 N, float obs[N], float err[N], float xy[2N], int16 label[N]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lccrf_densecrf.hpp"
#include "../../oracle/lccrf_oracle.h"

using namespace DenseCRF;
using namespace std;

// somebody else's pairwise term: out += c * in  (any subclass of the abstract base will do)
struct ScaledIdentity : PairwisePotential {
    float c;
    int L;
    ScaledIdentity(int N, int L_, float c_) : PairwisePotential(N), c(c_), L(L_) {}
    void apply(float *out, const float *in, float *) const override
    {
        for (int i = 0; i < N_ * L; ++i) out[i] += c * in[i];
    }
};

static int same(const float *a, const float *b, size_t n) { return memcmp(a, b, n * sizeof(float)) == 0; }

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    int N = 0;
    if (fread(&N, 4, 1, fp) != 1) return 2;
    vector<float> obs(N), err(N), xy(2 * (size_t)N);
    vector<short> label(N);
    if (fread(obs.data(), 4, N, fp) != (size_t)N || fread(err.data(), 4, N, fp) != (size_t)N ||
        fread(xy.data(), 8, N, fp) != (size_t)N || fread(label.data(), 2, N, fp) != (size_t)N)
        return 2;
    fclose(fp);
    const int M = 2;
    const float conf = 0.7f, w1 = 10.0f, w2 = 30.0f;
    vector<float> fa(2 * (size_t)N + 1), fs(2 * (size_t)N + 1);
    orc_appearance_features(N, obs.data(), err.data(), 1.5f, 0.6f, fa.data());
    orc_smooth_features(N, xy.data(), 18.0f, fs.data());

    // the oracle's CRF with both kernels
    orc_crf *o = orc_crf_create(N, M);
    float cf[M] = {conf, conf};
    orc_crf_set_unary_from_label(o, label.data(), cf);
    orc_crf_add_pairwise(o, fa.data(), 2, w1);
    orc_crf_add_pairwise(o, fs.data(), 2, w2);
    int bad = 0;
    try {
        // ---- 1. through base pointers ---------------------------------------------------------------
        {
            DenseCRF::DenseCRF *crf = CreateDenseCRF<M>(N, GPU);
            crf->setUnaryEnergyFromLabel(label.data(), conf);
            PairwisePotential *p1 = new PottsPotentialHIP<M, 2>(fa.data(), N, w1);
            PairwisePotential *p2 = new PottsPotentialHIP<M, 2>(fs.data(), N, w2);
            crf->addPairwiseEnergy(p1);              // the base class's non-virtual member
            crf->addPairwiseEnergy(p2);
            crf->inference(5, true);
            orc_crf_inference(o, 5, 1, 1.0f);
            const short *m = crf->getMap();          // the base class's getter: refreshed by inference(.., true)
            for (int i = 0; i < N; ++i) bad += m[i] != o->map[i];
            // densecrf_base.h:75 is non-virtual: a factory-made object keeps the base's buffer fresh by itself
            bad += !same(crf->getProbability(), o->current, (size_t)N * M);
            bad += !static_cast<const DenseCRFHIP<M> *>(crf)->syncsThroughBase();
            crf->startInference();                   // stepwise through the base pointer: Q visible after every call
            crf->stepInference();
            {
                orc_crf *o2 = orc_crf_create(N, M);
                orc_crf_set_unary_from_label(o2, label.data(), cf);
                orc_crf_add_pairwise(o2, fa.data(), 2, w1);
                orc_crf_add_pairwise(o2, fs.data(), 2, w2);
                orc_crf_inference(o2, 1, 0, 1.0f);
                bad += !same(crf->getProbability(), o2->current, (size_t)N * M);
                orc_crf_destroy(o2);
            }
            crf->inference(5, true);
            const DenseCRFHIP<M> &cref = *static_cast<DenseCRFHIP<M> *>(crf);
            bad += !same(cref.getProbability(), o->current, (size_t)N * M);    // const getters
            bad += cref.getMap()[0] != o->map[0];
            // a tracker on a shared GPU opts out of the two-workgroup form through the adapter (lccrf_set_option): same bits
            static_cast<DenseCRFHIP<M> *>(crf)->setOption(LCCRF_OPT_SINGLE_WORKGROUP, 1);
            crf->inference(5, true);
            bad += !same(cref.getProbability(), o->current, (size_t)N * M);
            for (int i = 0; i < N; ++i) bad += cref.getMap()[i] != o->map[i];
            delete crf;                              // deletes both potentials
            printf("base-pointer use: %s\n", bad ? "MISMATCH" : "ok");
        }
        // ---- 2. apply() on its own --------------------------------------------------------------------
        vector<float> in((size_t)N * M), out((size_t)N * M), ref((size_t)N * M), tmp((size_t)N * M);
        for (size_t i = 0; i < in.size(); ++i) {
            in[i] = (float)((i * 2654435761u) % 1000) / 1000.0f;
            out[i] = ref[i] = (float)((i * 40503u) % 97) / 10.0f - 4.0f;
        }
        {
            PottsPotentialHIP<M, 2> pot(fs.data(), N, w2, /*device_id=*/0);
            pot.setDevice(0);
            const PairwisePotential &base = pot;
            base.apply(out.data(), in.data(), tmp.data());
            orc_pairwise_apply(o, 1, ref.data(), in.data());
            const int b2 = !same(out.data(), ref.data(), out.size());
            bad += b2;
            printf("stand-alone apply: %s\n", b2 ? "MISMATCH" : "ok");
        }
        // ---- 3. a foreign potential between two of ours ------------------------------------------------
        {
            DenseCRFHIP<M> crf(N);
            crf.setUnaryEnergyFromLabel(label.data(), conf);
            crf.addPairwiseEnergy(new PottsPotentialHIP<M, 2>(fa.data(), N, w1));
            crf.addPairwiseEnergy(new ScaledIdentity(N, M, 0.25f));
            crf.addPairwiseEnergy(new PottsPotentialHIP<M, 2>(fs.data(), N, w2));
            crf.inference(3, true, 0.75f);
            // the same three steps by hand on the oracle (densecrf_base.h:78-91)
            vector<float> cur((size_t)N * M), nxt((size_t)N * M);
            orc_exp_and_normalize(cur.data(), o->unary, N, M, -1.0f, 1.0f);
            for (int it = 0; it < 3; ++it) {
                for (size_t i = 0; i < nxt.size(); ++i) nxt[i] = -o->unary[i];
                orc_pairwise_apply(o, 0, nxt.data(), cur.data());
                for (size_t i = 0; i < nxt.size(); ++i) nxt[i] += 0.25f * cur[i];
                orc_pairwise_apply(o, 1, nxt.data(), cur.data());
                orc_exp_and_normalize(cur.data(), nxt.data(), N, M, 1.0f, 0.75f);
            }
            int b3 = !crf.mixed() || !same(crf.getProbability(), cur.data(), cur.size());
            for (int i = 0; i < N; ++i) b3 += crf.getMap()[i] != (cur[2 * i] < cur[2 * i + 1] ? 1 : 0);
            bad += b3;
            printf("mixed potentials: %s\n", b3 ? "MISMATCH" : "ok");
        }
    } catch (const std::exception &e) {
        printf("EXCEPTION %s\n", e.what());
        return strstr(e.what(), "no HIP device") ? 3 : 4;
    }
    orc_crf_destroy(o);
    printf("%s N=%d\n", bad ? "ADAPTER MISMATCH" : "ADAPTER OK", N);
    return bad ? 1 : 0;
}
