// Fill most of the free device memory with a pattern and release it, so that a later process that
// reads device memory it never wrote sees the pattern instead of zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
int main(int argc, char **argv)
{
    int pat = argc > 1 ? (int)strtoul(argv[1], nullptr, 16) : 0xff;
    size_t fr = 0, tot = 0;
    hipMemGetInfo(&fr, &tot);
    std::vector<void *> bufs;
    const size_t chunk = (size_t)1 << 30;
    size_t got = 0;
    while (got + 2 * chunk < fr && bufs.size() < 200) {
        void *p = nullptr;
        if (hipMalloc(&p, chunk) != hipSuccess) break;
        hipMemset(p, pat, chunk);
        bufs.push_back(p);
        got += chunk;
    }
    hipDeviceSynchronize();
    for (void *p : bufs) hipFree(p);
    printf("poisoned %zu GiB of device memory with byte %02x\n", got >> 30, pat & 0xff);
    return 0;
}
