// `tungsten_amd` -- the StandaloneRenderer loop of the reference CLI (src/tungsten/Shared.hpp:191-337)
// on top of the MI355X integrator:  load scene -> flatten -> pass loop -> outputs.
//   tungsten_amd [--spp N] [-s seed] [-o out.png] [-e out.pfm] [-d outdir] [-i indir] [-c interval] [--timeout t] [-r] [--accept-foreign-resume-file]
//                [--gpus a,b,..] [--pools K] scene.json...
// -c / --timeout / -r / -i and the resume file are the reference's (Shared.hpp:134-145,262-319, Integrator.cpp:108-162).
// --pools K (default 2): path pools per GPU, each a context on its own HIP stream; the image tiles are dealt over all
// pools, which fill each other's kernel tails and host round trips (+10 % measured, same image bit for bit).
#include "integrator.hpp"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sstream>

using namespace tungsten_amd;

// Durations of -c / --timeout (and of "checkpoint_interval" / "timeout" in a scene file): a sum of <number><unit> terms such as "90", "1.5h",
// "2m 30s" or "500ms".  Units are recognised by their first letters in either case; a number without a unit counts minutes -- the reference's
// convention (io/StringUtils.cpp:14-48), kept so that existing command lines mean the same.  Returns seconds.
static double parseDuration(const std::string &text)
{
    struct Unit { const char *letters; double seconds; };
    static const Unit kUnits[] = {{"ms", 1e-3}, {"s", 1.0}, {"m", 60.0}, {"h", 3600.0}, {"d", 86400.0}}; // "ms" has to be tried before "m"
    auto startsWith = [](const char *at, const char *letters) {
        for (; *letters; ++letters, ++at)
            if (std::tolower(static_cast<unsigned char>(*at)) != *letters) return false;
        return true;
    };
    double seconds = 0.0;
    const char *at = text.c_str();
    for (;;) {
        char *afterNumber;
        const double value = std::strtod(at, &afterNumber);
        if (afterNumber == at) return seconds; // no further term
        at = afterNumber;
        while (std::isspace(static_cast<unsigned char>(*at))) ++at;
        double scale = 60.0;
        for (const Unit &u : kUnits)
            if (startsWith(at, u.letters)) { scale = u.seconds; break; }
        seconds += value*scale;
        while (std::isalpha(static_cast<unsigned char>(*at))) ++at; // the rest of the unit's name ("min", "hours", ...)
    }
}

int main(int argc, char **argv)
{
    uint32_t spp = 0, seed = 0xBA5EBA11u; // default seed: Shared.hpp:246-248
    std::string outPng, outPfm, outDir, inDir, checkpointArg, timeoutArg;
    bool restart = false, haveCheckpoint = false, haveTimeout = false, acceptForeignResume = false;
    std::vector<int> gpus = {0};
    int pools = 2;
    std::vector<std::string> scenes;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--spp") spp = uint32_t(std::atoi(next().c_str()));
        else if (a == "-s" || a == "--seed") seed = uint32_t(std::atoi(next().c_str())); // note: atoi, as the reference
        else if (a == "-o" || a == "--output-file") outPng = next();
        else if (a == "-e" || a == "--hdr-output-file") outPfm = next();
        else if (a == "-d" || a == "--output-directory") outDir = next();
        else if (a == "-i" || a == "--input-directory") inDir = next();
        else if (a == "-c" || a == "--checkpoint") { checkpointArg = next(); haveCheckpoint = true; }
        else if (a == "--timeout") { timeoutArg = next(); haveTimeout = true; }
        else if (a == "-r" || a == "--restart") restart = true;
        else if (a == "--accept-foreign-resume-file") acceptForeignResume = true; // a state file whose scene hash is not this loader's (the reference's)
        else if (a == "-t" || a == "--threads") next(); // host threads: meaningless here, accepted for compatibility
        else if (a == "--gpus") { gpus.clear(); std::stringstream ss(next()); std::string t; while (std::getline(ss, t, ',')) gpus.push_back(std::atoi(t.c_str())); }
        else if (a == "--pools") pools = std::max(1, std::atoi(next().c_str()));
        else if (a == "-h" || a == "--help") {
            std::printf("usage: tungsten_amd [--spp N] [-s seed] [-o png] [-e pfm] [-d outdir] [-i indir] [-c checkpoint-interval] [--timeout duration] [-r] [--accept-foreign-resume-file]\n"
                        "                    [--gpus 0,1,..] [--pools K] scene.json...\n");
            return 0;
        }
        else scenes.push_back(a);
    }
    if (scenes.empty()) { std::fprintf(stderr, "No input files\n"); return 1; }
    int failures = 0;
    for (const std::string &path : scenes) {
        try {
            std::printf("Loading scene '%s'...\n", path.c_str());
            Scene scene = Scene::load(path, inDir); // -i: resources are looked up there instead of next to the scene file (Shared.hpp:205)
            if (spp) scene.setSpp(spp);
            if (!outPng.empty()) scene.renderer.outputFile = outPng;
            if (!outPfm.empty()) scene.renderer.hdrOutputFile = outPfm;
            if (!outDir.empty()) { // RendererSettings::setOutputDirectory: output, hdr output and resume files
                if (!scene.renderer.outputFile.empty()) scene.renderer.outputFile = outDir + "/" + scene.renderer.outputFile;
                if (!scene.renderer.hdrOutputFile.empty()) scene.renderer.hdrOutputFile = outDir + "/" + scene.renderer.hdrOutputFile;
                if (!scene.renderer.resumeRenderFile.empty()) scene.renderer.resumeRenderFile = outDir + "/" + scene.renderer.resumeRenderFile;
                for (OutputBufferSettings &b : scene.renderer.outputs) { // OutputBufferSettings::setOutputDirectory (RendererSettings.hpp:119-120)
                    if (!b.ldrOutputFile.empty()) b.ldrOutputFile = outDir + "/" + b.ldrOutputFile;
                    if (!b.hdrOutputFile.empty()) b.hdrOutputFile = outDir + "/" + b.hdrOutputFile;
                }
            }
            scene.flatten();
            std::vector<int> contexts; // one entry per pool: GPU g appears `pools` times
            for (int g : gpus) for (int k = 0; k < pools; ++k) contexts.push_back(g);
            PathTraceIntegrator integrator(contexts);
            integrator.prepareForRender(scene, seed);
            const double checkpointInterval = parseDuration(haveCheckpoint ? checkpointArg : scene.renderer.checkpointInterval); // Shared.hpp:262-265
            const double timeout = parseDuration(haveTimeout ? timeoutArg : scene.renderer.timeout);
            bool resume = scene.renderer.enableResumeRender;
            if (resume && !restart) { // Shared.hpp:267-278
                std::printf("Trying to resume render from saved state... \n");
                bool hashMismatch = false;
                if (integrator.resumeRender(acceptForeignResume, &hashMismatch)) {
                    if (hashMismatch) std::printf("Warning: the resume file carries another scene hash (written by the reference renderer, or for another scene text); accepted on request\n");
                    std::printf("Resume successful\n");
                } else {
                    std::printf("Resume unsuccessful. Starting from 0 spp\n");
                }
            }
            std::printf("Starting render...\n");
            auto t0 = std::chrono::steady_clock::now(), tCheckpoint = t0;
            const uint32_t startSpp = integrator.currentSpp();
            while (!integrator.done()) { // Shared.hpp:283-311
                integrator.startRender([]() {});
                integrator.waitForCompletion();
                std::printf("Completed %u/%u spp\n", integrator.currentSpp(), scene.renderer.spp);
                auto now = std::chrono::steady_clock::now();
                if (timeout > 0.0 && std::chrono::duration<double>(now - t0).count() > timeout) break;
                if (checkpointInterval > 0.0 && std::chrono::duration<double>(now - tCheckpoint).count() > checkpointInterval) {
                    std::printf("Saving checkpoint after %.1fs\n", std::chrono::duration<double>(now - t0).count());
                    tCheckpoint = now;
                    integrator.saveCheckpoint();
                    if (resume) integrator.saveRenderResumeData();
                }
            }
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("Finished render. Render time %.3fs  (%.1f Msamples/s)\n", dt, double(scene.resX)*scene.resY*(integrator.currentSpp() - startSpp)/dt/1e6);
            integrator.saveOutputs();
            if (resume) integrator.saveRenderResumeData(); // Shared.hpp:318-319
        } catch (const std::exception &e) { // Shared.hpp:325-328: log and move on to the next scene
            std::fprintf(stderr, "Scene '%s' failed: %s\n", path.c_str(), e.what());
            ++failures;
        }
    }
    return failures ? 1 : 0;
}
