// Test-infrastructure driver (NOT product code): links the reference's own
// thirdparty/sobol/sobol.cpp, compiled where it lies under /root/reference, and
// writes the generator-matrix columns reachable with a 32-bit sample index as
// OUTPUTS of sobol::sample(): column c of dimension d = sample(1ull << c, d, 0)
// (thirdparty/sobol/sobol.h:40-53).  Output: 1024 x 32 little-endian u32.
#include <sobol/sobol.h>
#include <cstdio>
#include <cstdint>

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: sobol_dump out.bin\n"); return 2; }
    FILE *f = std::fopen(argv[1], "wb");
    if (!f) return 1;
    for (unsigned d = 0; d < sobol::Matrices::num_dimensions; ++d) {
        for (unsigned c = 0; c < 32; ++c) {
            uint32_t v = sobol::sample(1ull << c, d, 0u);
            std::fwrite(&v, 4, 1, f);
        }
    }
    std::fclose(f);
    return 0;
}
