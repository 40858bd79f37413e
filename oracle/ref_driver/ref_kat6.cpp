// Known-answer driver (test infrastructure, built by oracle/Makefile.ref from the reference's OWN thirdparty/stbi/stb_image.c where it lies):
// what stbi_load(..., 4) -- the call behind ImageIO::loadStbi, io/ImageIO.cpp:489-490 -- returns for the .bmp / .tga / .jpg files of
// assets/next/textures: width, height and every RGBA byte, as JSON on stdout.  tests/test_ref_kat6.py compares the host loader's decoders with it.
#include <stbi/stb_image.h>
#include <cstdio>
#include <cstring>

int main(int argc, char **argv)
{
    std::printf("{\n");
    for (int i = 1; i < argc; ++i) {
        int w = 0, h = 0, comp = 0;
        unsigned char *px = stbi_load(argv[i], &w, &h, &comp, 4);
        const char *name = std::strrchr(argv[i], '/');
        name = name ? name + 1 : argv[i];
        std::printf("  \"%s\": ", name);
        if (!px) {
            std::printf("{\"error\": \"%s\"}", stbi_failure_reason());
        } else {
            std::printf("{\"w\": %d, \"h\": %d, \"components_in_file\": %d, \"rgba\": \"", w, h, comp);
            for (int k = 0; k < w*h*4; ++k) std::printf("%02x", px[k]);
            std::printf("\"}");
            stbi_image_free(px);
        }
        std::printf("%s\n", i + 1 < argc ? "," : "");
    }
    std::printf("}\n");
    return 0;
}
