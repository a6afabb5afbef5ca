// image_demo_test.cpp -- the reference library's own demo (Thirdparty/DenseCRF/examples/example_cpu.cpp:79-103) re-typed against the
// adapter: DenseCRF3D<21> -> DenseCRFHIP<21>, the image potentials -> PottsPotentialHIP<21, 2> / <21, 5> on features formed exactly
// as PairwisePotential's image constructors form them (pairwise_cpu.h:33-51).  The Python test compares the labels written here with
// the reference's known answer (res1_cpu.ppm).
//   image_demo_test <in.bin> <out.bin>     in: int32 W, H; uint8 rgb[W*H*3]; int16 anno[W*H]     out: int16 map[W*H]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lccrf_densecrf.hpp"

using namespace DenseCRF;
using namespace std;

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *fp = fopen(argv[1], "rb");
    if (!fp) return 2;
    int W = 0, H = 0;
    if (fread(&W, 4, 1, fp) != 1 || fread(&H, 4, 1, fp) != 1) return 2;
    const int N = W * H, M = 21;
    vector<unsigned char> im((size_t)N * 3);
    vector<short> anno(N);
    if (fread(im.data(), 1, im.size(), fp) != im.size() || fread(anno.data(), 2, N, fp) != (size_t)N) return 2;
    fclose(fp);
    vector<float> smooth((size_t)N * 2), app((size_t)N * 5);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            smooth[i * 2 + 0] = (float)x / 3.0f;          // addPairwiseGaussian(3, 3, 3)
            smooth[i * 2 + 1] = (float)y / 3.0f;
            app[i * 5 + 0] = (float)x / 60.0f;            // addPairwiseBilateral(60, 60, 20, 20, 20, im, 10)
            app[i * 5 + 1] = (float)y / 60.0f;
            for (int c = 0; c < 3; ++c) app[i * 5 + 2 + c] = (float)im[i * 3 + c] / 20.0f;
        }
    try {
        DenseCRFHIP<M> crf(N);
        crf.setUnaryEnergyFromLabel(anno.data(), 0.5f);   // GT_PROB
        crf.addPairwiseEnergy(new PottsPotentialHIP<M, 2>(smooth.data(), N, 3.0f));
        crf.addPairwiseEnergy(new PottsPotentialHIP<M, 5>(app.data(), N, 10.0f));
        crf.inference(10, true);
        const short *map = crf.getMap();
        FILE *fo = fopen(argv[2], "wb");
        if (!fo || fwrite(map, 2, N, fo) != (size_t)N) return 2;
        fclose(fo);
    } catch (const std::exception &e) {
        fprintf(stderr, "image_demo_test: %s\n", e.what());
        return 1;
    }
    printf("IMAGE DEMO OK\n");
    return 0;
}
