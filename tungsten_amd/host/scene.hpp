// Host-side scene: loads the reference's JSON scene format and resources and
// flattens them into the POD description the C ABI takes (tgpu_scene_desc).
// Mirrors Scene::load -> fromJson -> loadResources -> makeTraceable
// (src/core/io/Scene.cpp:236-253,281-306,373-391) and the prepareForRender
// bakes of TraceableScene (src/core/renderer/TraceableScene.hpp:57-137).
#pragma once
#include "../../include/tungsten_amd.h"
#include "json.hpp"
#include "vecmath.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace tungsten_amd {

// RendererSettings (src/core/renderer/RendererSettings.hpp:34-76)
// cameras/OutputBufferSettings.hpp:13-50: one entry of renderer "output_buffers"
struct OutputBufferSettings {
    int type = TGPU_OUTPUT_COLOR; // "color" | "depth" | "normal" | "albedo" | "visibility"
    std::string ldrOutputFile, hdrOutputFile;
    bool twoBufferVariance = false, sampleVariance = false;
};

struct RendererSettings {
    std::string outputFile = "TungstenRender.png";
    std::string hdrOutputFile;
    std::string resumeRenderFile = "TungstenRenderState.dat";
    bool overwriteOutputFiles = true;
    bool useAdaptiveSampling = true;
    bool enableResumeRender = false;
    bool useSceneBvh = true;
    bool useSobol = true;
    uint32_t spp = 32;
    uint32_t sppStep = 16;
    std::string checkpointInterval = "0";
    std::string timeout = "0";
    std::vector<OutputBufferSettings> outputs; // "output_buffers" (RendererSettings.hpp:69-74)
};

struct MeshData {
    std::string name;
    std::vector<float> verts;      // nVerts*8 : pos3 normal3 uv2 (.wo3 layout, MeshIO.cpp:12-28)
    std::vector<uint32_t> tris;    // nTris*4  : v0 v1 v2 material
    Mat4 transform;
    bool smooth = false;
    std::vector<int> bsdfs;
    int intMedium = -1, extMedium = -1; // Primitive::_intMedium / _extMedium
};

class Scene {
public:
    // Scene::load (Scene.cpp:378-391). Throws JsonLoadException / std::runtime_error.
    // inputDirectory: where relative resource paths are resolved (the CLI's -i; default: the scene file's directory, Shared.hpp:205)
    static Scene load(const std::string &jsonPath, const std::string &inputDirectory = std::string());
    // The same from the scene text itself (what Scene::toJson of the reference writes); relative resource paths are resolved in resourceDirectory
    static Scene loadText(const std::string &jsonText, const std::string &resourceDirectory);

    // CLI-style overrides applied before flatten() (Shared.hpp:221-248)
    void setResolution(uint32_t w, uint32_t h);
    void setSpp(uint32_t spp) { renderer.spp = spp; }

    // Bakes everything (prepareForRender of camera/primitives/textures) and fills desc().
    void flatten();
    const tgpu_scene_desc &desc() const { return _desc; }

    RendererSettings renderer;
    std::string tonemap = "gamma";
    std::string sceneDir;
    std::string sceneJsonText;

    // camera (Camera.cpp:44-68, PinholeCamera.cpp:28-43)
    uint32_t resX = 1000, resY = 563;
    float fovDeg = 60.0f;
    Mat4 camTransform;
    std::string filterName = "tent";

    tgpu_settings settings{};

    std::vector<MeshData> meshes;

    // Optional location of the Sobol' matrix asset; default: $TUNGSTEN_AMD_ASSETS or <repo>/assets
    static std::string assetDir();

private:
    int fetchBsdf(const Json &v);
    int fetchTexture(const Json &v, bool scalarAverage);
    int fetchBitmap(const std::string &path, bool scalarAverage, bool gammaCorrect, bool linear, bool clamp, float scale);
    int parseBsdf(const Json &v);
    void parsePrimitive(const Json &v);
    void parseCamera(const Json &v);

    struct QuadSrc { Mat4 transform; int bsdf; int emissionTex; int intMedium = -1, extMedium = -1; };
    struct CubeSrc { Mat4 transform; int bsdf; int emissionTex; int intMedium = -1, extMedium = -1; };
    int fetchMedium(const Json &v);
    int parseMedium(const Json &v);
    std::vector<std::pair<std::string, int>> _namedMedia;
    std::vector<tgpu_medium> _media;
    std::vector<int16_t> _triMedia;
    std::vector<uint32_t> _meshFirstTri; // n_meshes + 1 offsets into the soup
    std::vector<float> _meshBounds;      // n_meshes*6
    int _cameraMedium = -1;
    // "instances" primitive (primitives/Instance.cpp:60-93): masters are private primitives of the set
    struct MasterSrc { int kind; MeshData mesh; QuadSrc quad; CubeSrc cube; };
    struct InstanceSrc { Mat4 transform; std::vector<MasterSrc> masters; std::vector<float> pos, rot; std::vector<uint8_t> id; };
    struct EnvSrc { bool present = false; Mat4 transform; int emissionTex = -1; bool sample = true; };
    enum PrimKind { PrimMesh, PrimQuad, PrimCube, PrimEnv, PrimSphere, PrimDisk };
    struct SphereSrc { Mat4 transform; int bsdf; int emissionTex; int intMedium = -1, extMedium = -1; };
    struct DiskSrc { Mat4 transform; int bsdf; int emissionTex; float coneAngle = 90.0f; int intMedium = -1, extMedium = -1; };
    std::vector<SphereSrc> _sphereSrc;
    std::vector<DiskSrc> _diskSrc;
    std::vector<tgpu_sphere> _spheres;
    std::vector<tgpu_disk> _disks;
    struct PrimRef { PrimKind kind; int index; };

    std::vector<std::pair<std::string, int>> _namedBsdfs;
    std::vector<tgpu_bsdf> _bsdfs;
    std::vector<tgpu_texture> _textures;
    std::vector<tgpu_bitmap> _bitmaps;
    std::vector<std::vector<float>> _bitmapTexels;
    std::vector<std::string> _bitmapPaths;
    std::vector<QuadSrc> _quadSrc;
    std::vector<CubeSrc> _cubeSrc;
    std::vector<InstanceSrc> _instanceSrc;
    // flattened instance sets (pointers into these go into _desc)
    struct InstanceFlat { std::vector<tgpu_master> masters; std::vector<float> pos, rot; std::vector<uint8_t> id; };
    std::vector<InstanceFlat> _instanceFlat;
    std::vector<tgpu_instances> _instanceSets;
    void bakeMesh(const MeshData &m, float *boundsMin, float *boundsMax);
    tgpu_quad bakeQuad(const QuadSrc &q) const;
    tgpu_cube bakeCube(const CubeSrc &c) const;
    MeshData parseMesh(const Json &v, const Mat4 &transform, int emissionTex);
    EnvSrc _envSrc;
    std::vector<PrimRef> _primOrder;
    Json _bsdfArray;

    // flattened storage
    std::vector<float> _triPos, _triNormal, _triUv;
    std::vector<int32_t> _triBsdf;
    std::vector<uint8_t> _triSmooth;
    std::vector<tgpu_quad> _quads;
    std::vector<tgpu_cube> _cubes;
    std::vector<tgpu_light> _lights;
    std::vector<float> _envPdf, _envCdf, _envMarginalPdf, _envMarginalCdf;
    std::vector<uint32_t> _sobol;
    tgpu_scene_desc _desc{};
};

// The image-plane shard of rank `rank` of `world`: 16x16 tiles diced and seeded
// exactly as PathTraceIntegrator::diceTiles (PathTraceIntegrator.cpp:27-42) does
// from `seed` (prepareForRender :184-201), tile t -> rank t % world.
struct Shard {
    std::vector<uint32_t> pixelIndex;
    std::vector<uint32_t> tileSeed;
    tgpu_shard_desc desc() const
    {
        return tgpu_shard_desc{uint32_t(pixelIndex.size()), pixelIndex.data(), tileSeed.data()};
    }
};
Shard makeShard(uint32_t resX, uint32_t resY, uint32_t seed, uint32_t rank, uint32_t world);
// the seeds diceTiles draws for the 16x16 tiles, row-major (PathTraceIntegrator.cpp:27-42)
std::vector<uint32_t> tileSeeds(uint32_t resX, uint32_t resY, uint32_t seed);

// UniformSampler / hash32 restated for seeding (UniformSampler.hpp:40-52, MathUtil.hpp:120-128)
uint32_t hash32(uint32_t x);
struct Pcg32 {
    uint64_t state, sequence;
    explicit Pcg32(uint64_t seed = 0xBA5EBA11ull, uint64_t seq = 0) : state(seed), sequence(seq) {}
    uint32_t nextI()
    {
        uint64_t old = state;
        state = old*6364136223846793005ULL + (sequence | 1);
        uint32_t xorShifted = uint32_t(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = uint32_t(old >> 59u);
        return (xorShifted >> rot) | (xorShifted << (uint32_t(-int32_t(rot)) & 31));
    }
    float next1D()
    {
        uint32_t i = nextI();
        uint32_t bits = (i >> 9u) | 0x3F800000u;
        float f;
        __builtin_memcpy(&f, &bits, 4);
        return f - 1.0f;
    }
};

// Image / mesh I/O used by the host (formats: SURVEY.md Appendix B)
bool loadWo3(const std::string &path, std::vector<float> &verts, std::vector<uint32_t> &tris);
bool loadRadianceHdr(const std::string &path, int &w, int &h, std::vector<float> &rgb);
bool savePfm(const std::string &path, const float *img, int w, int h, int channels = 3);
bool loadPfm(const std::string &path, int &w, int &h, std::vector<float> &rgb);
bool savePng(const std::string &path, const uint8_t *rgb, int w, int h);
// 8-bit, non-interlaced PNG -> RGBA8 (what lodepng_decode_memory(..., LCT_RGBA, 8) returns for such files, io/ImageIO.cpp:386-407)
bool loadPng(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba);
bool loadBmp(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba, std::string &why);
bool loadTga(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba, std::string &why);
bool loadJpeg(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba, std::string &why);
// Camera::tonemap + LDR quantisation (Tonemap.hpp:25-48, Integrator.cpp:56-80)
void tonemapToLdr(const std::string &op, const float *rgb, int w, int h, std::vector<uint8_t> &ldr);

} // namespace tungsten_amd
