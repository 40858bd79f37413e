// `tungsten_amd` -- the StandaloneRenderer loop of the reference CLI (src/tungsten/Shared.hpp:191-337)
// on top of the MI355X integrator:  load scene -> flatten -> pass loop -> outputs.
//   tungsten_amd [--spp N] [-s seed] [-o out.png] [-e out.pfm] [-d outdir] [--gpus a,b,..] [--pools K] scene.json...
// --pools K (default 2): path pools per GPU, each a context on its own HIP stream; the image tiles are dealt over all
// pools, which fill each other's kernel tails and host round trips (+10 % measured, same image bit for bit).
#include "integrator.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>

using namespace tungsten_amd;

int main(int argc, char **argv)
{
    uint32_t spp = 0, seed = 0xBA5EBA11u; // default seed: Shared.hpp:246-248
    std::string outPng, outPfm, outDir;
    std::vector<int> gpus = {0};
    int pools = 2;
    std::vector<std::string> scenes;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--spp") spp = uint32_t(std::atoi(next().c_str()));
        else if (a == "-s" || a == "--seed") seed = uint32_t(std::atoi(next().c_str())); // note: atoi, as the reference
        else if (a == "-o") outPng = next();
        else if (a == "-e") outPfm = next();
        else if (a == "-d") outDir = next();
        else if (a == "-t" || a == "--threads") next(); // host threads: meaningless here, accepted for compatibility
        else if (a == "--gpus") { gpus.clear(); std::stringstream ss(next()); std::string t; while (std::getline(ss, t, ',')) gpus.push_back(std::atoi(t.c_str())); }
        else if (a == "--pools") pools = std::max(1, std::atoi(next().c_str()));
        else if (a == "-h" || a == "--help") { std::printf("usage: tungsten_amd [--spp N] [-s seed] [-o png] [-e pfm] [-d dir] [--gpus 0,1,..] [--pools K] scene.json...\n"); return 0; }
        else scenes.push_back(a);
    }
    if (scenes.empty()) { std::fprintf(stderr, "No input files\n"); return 1; }
    int failures = 0;
    for (const std::string &path : scenes) {
        try {
            std::printf("Loading scene '%s'...\n", path.c_str());
            Scene scene = Scene::load(path);
            if (spp) scene.setSpp(spp);
            if (!outPng.empty()) scene.renderer.outputFile = outPng;
            if (!outPfm.empty()) scene.renderer.hdrOutputFile = outPfm;
            if (!outDir.empty()) {
                if (!scene.renderer.outputFile.empty()) scene.renderer.outputFile = outDir + "/" + scene.renderer.outputFile;
                if (!scene.renderer.hdrOutputFile.empty()) scene.renderer.hdrOutputFile = outDir + "/" + scene.renderer.hdrOutputFile;
            }
            scene.flatten();
            std::vector<int> contexts; // one entry per pool: GPU g appears `pools` times
            for (int g : gpus) for (int k = 0; k < pools; ++k) contexts.push_back(g);
            PathTraceIntegrator integrator(contexts);
            integrator.prepareForRender(scene, seed);
            std::printf("Starting render...\n");
            auto t0 = std::chrono::steady_clock::now();
            while (!integrator.done()) { // Shared.hpp:283-311
                integrator.startRender([]() {});
                integrator.waitForCompletion();
                std::printf("Completed %u/%u spp\n", integrator.currentSpp(), scene.renderer.spp);
            }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("Finished render. Render time %.3fs  (%.1f Msamples/s)\n", dt, double(scene.resX)*scene.resY*scene.renderer.spp/dt/1e6);
            integrator.saveOutputs();
        } catch (const std::exception &e) { // Shared.hpp:325-328: log and move on to the next scene
            std::fprintf(stderr, "Scene '%s' failed: %s\n", path.c_str(), e.what());
            ++failures;
        }
    }
    return failures ? 1 : 0;
}
