/*
 * ref_driver_image.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Second translation unit of oracle/_ref/liblccrf_ref.so (see ref_driver.cpp):
 * the reference's 2-D image classes DenseCRFCPU<M> / PottsPotentialCPU<M,F>::FromImage,
 * compiled from the headers where they lie.  Kept apart from ref_driver.cpp because
 * densecrf_cpu.h and densecrf3d.h both define DenseCRF::fast_exp.
 *
 * Like examples/example_cpu.cpp (whose `using namespace std` comes AFTER its
 * includes, example_cpu.cpp:1-10) this TU has no using-directive ahead of the
 * headers.  example_cpu.cpp itself is not built: its main() declares an unused
 * std::vector<cv::Point3f> (example_cpu.cpp:54) and OpenCV is absent here.
 */
#include <cstdint>
#include <cstring>
#include <cmath>
#include <algorithm>

#include "densecrf_cpu.h"
#include "pairwise_cpu.h"

extern "C" {

/*
 * The flow of the reference's example program (examples/example_cpu.cpp:79-103) on
 * caller-supplied pixels and labels: DenseCRFCPU<21>, FromImage 2-D (posdev 3, w 3)
 * and 5-D (posdev 60, featuredev 20, w 10) kernels, inference(10, true).
 */
int ref_example_image(int W, int H, const unsigned char *rgb, const int16_t *label,
                      float gt_prob, int n_iter, int16_t *map_out)
{
    const int M = 21;
    DenseCRF::DenseCRFCPU<M> crf(W * H);
    crf.setUnaryEnergyFromLabel(label, gt_prob);
    crf.addPairwiseEnergy(DenseCRF::PottsPotentialCPU<M, 2>::FromImage<>(W, H, 3.0, 3.0));
    crf.addPairwiseEnergy(DenseCRF::PottsPotentialCPU<M, 5>::FromImage<unsigned char>(W, H, 10.0, 60.0, rgb, 20.0));
    crf.inference(n_iter, true);
    std::memcpy(map_out, crf.getMap(), sizeof(int16_t) * (size_t)W * H);
    return 0;
}

}  // extern "C"
