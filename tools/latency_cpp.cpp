// latency_cpp.cpp -- single-frame latency of the CRF call site as the tracker sees it: a C++ caller, one frame at a
// time, host buffers in and out (src/Tracking.cc:1919-1930 with the two type names swapped, include/lccrf_densecrf.hpp).
// Product code only (no oracle): bench.py compiles and runs it for the `single_frame_latency_us` record.
//
//   latency_cpp <inputs.bin> <repetitions>
//       inputs: int32 n_frames, int32 N, then per frame float obs[N], float err[N], float xy[2N], int16 label[N]
//   prints one JSON object: median / p90 of the whole call site and of its parts (microseconds)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lccrf_densecrf.hpp"

struct Point2f { float x, y; };
struct Point3f { float x, y, z; };
using namespace DenseCRF;
using Clock = std::chrono::steady_clock;

static double us(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double pct(std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    int F = 0, N = 0;
    if (fread(&F, 4, 1, fp) != 1 || fread(&N, 4, 1, fp) != 1 || F < 1) return 2;
    std::vector<std::vector<float>> obs(F), err(F);
    std::vector<std::vector<Point2f>> xy(F);
    std::vector<std::vector<short>> lab(F);
    for (int f = 0; f < F; ++f) {
        obs[f].resize(N); err[f].resize(N); xy[f].resize(N); lab[f].resize(N);
        if (fread(obs[f].data(), 4, N, fp) != (size_t)N || fread(err[f].data(), 4, N, fp) != (size_t)N ||
            fread(xy[f].data(), 8, N, fp) != (size_t)N || fread(lab[f].data(), 2, N, fp) != (size_t)N) return 2;
    }
    fclose(fp);
    const int reps = atoi(argv[2]);
    const float mConf = 0.7f, mW1 = 10.0f, mW2 = 30.0f, mObservStdev = 1.5f, mRpjErrorStdev = 0.6f, mPoint3dStdev = 0.5f,
                mPoint2dStdev = 18.0f;                                                  // TUM3.yaml:78-101
    std::vector<Point3f> vpoints(N);
    std::vector<double> total, t_setup, t_launch, t_wait, t_destroy;
    long dyn = 0;
    try {
        const int M = 2;
        for (int r = 0; r < reps + 10; ++r) {
            const int f = r % F;
            const auto t0 = Clock::now();
            Clock::time_point t1, t2, t3;
            {
                DenseCRFHIP<M> crf(N);                                                  // Tracking.cc:1920
                crf.setUnaryEnergyFromLabel(lab[f].data(), mConf);
                crf.addPairwiseEnergy(PottsPotentialHIP<M, 2>::appearanceKernel(N, mW1, obs[f], err[f], mObservStdev, mRpjErrorStdev));
                crf.addPairwiseEnergy(PottsPotentialHIP<M, 2>::smoothKernel(N, mW2, vpoints, xy[f], mPoint3dStdev, mPoint2dStdev));
                t1 = Clock::now();
                // inference(5, true) = launch + wait for the labels; timed apart through the C-ABI the adapter itself uses
                lccrf_check(lccrf_inference(crf.handle(), 5, 1, 1.0f), "lccrf_inference");
                t2 = Clock::now();
                lccrf_check(lccrf_get_map(crf.handle(), crf.getMap()), "lccrf_get_map");
                t3 = Clock::now();
                dyn += crf.getMap()[0] == 0;
            }
            const auto t4 = Clock::now();
            if (r >= 10) {
                total.push_back(us(t0, t4)); t_setup.push_back(us(t0, t1)); t_launch.push_back(us(t1, t2));
                t_wait.push_back(us(t2, t3)); t_destroy.push_back(us(t3, t4));
            }
        }
    } catch (const std::exception &e) {
        printf("{\"error\": \"%s\"}\n", e.what());
        return strstr(e.what(), "no HIP device") ? 3 : 4;
    }
    printf("{\"n_points\": %d, \"reps\": %d, \"median\": %.2f, \"p90\": %.2f, \"min\": %.2f, \"setup_create_unary_kernels\": %.2f, "
           "\"launch_inference_call\": %.2f, \"wait_getMap\": %.2f, \"destroy\": %.2f, \"dyn0\": %ld}\n",
           N, reps, pct(total, 0.5), pct(total, 0.9), pct(total, 0.0), pct(t_setup, 0.5), pct(t_launch, 0.5), pct(t_wait, 0.5),
           pct(t_destroy, 0.5), dyn);
    return 0;
}
