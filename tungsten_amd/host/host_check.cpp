// Hardening driver for the loader (`make asan`): loads and flattens every scene file named on the command line and reports, one line per file,
// "ok" or "refused: <message>".  Built with -fsanitize=address,undefined it turns a parser that reads past a truncated or corrupt input
// (own inflate / PNG / RGBE / .wo3 / instance-file readers, host/scene.cpp) into a hard failure; tests/test_host.py feeds it damaged copies of
// the vendored assets.  It never touches the GPU.
#include "scene.hpp"

#include <cstdio>
#include <exception>

int main(int argc, char **argv)
{
    int refused = 0;
    for (int i = 1; i < argc; ++i) {
        try {
            tungsten_amd::Scene scene = tungsten_amd::Scene::load(argv[i], "");
            scene.flatten();
            std::printf("%s: ok (%u triangles)\n", argv[i], scene.desc().n_tris);
        } catch (const std::exception &e) {
            std::printf("%s: refused: %s\n", argv[i], e.what());
            ++refused;
        }
    }
    std::printf("%d of %d refused\n", refused, argc - 1);
    return 0;
}
