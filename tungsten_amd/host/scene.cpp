#include "scene.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace tungsten_amd {

static const float PI = 3.1415926536f; // Angle.hpp:8

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
static std::string readFile(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("Unable to open file '" + path + "'");
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

static std::string dirName(const std::string &path)
{
    size_t p = path.find_last_of('/');
    return p == std::string::npos ? std::string(".") : path.substr(0, p);
}

static bool endsWith(const std::string &s, const std::string &suffix)
{
    if (s.size() < suffix.size()) return false;
    std::string tail = s.substr(s.size() - suffix.size());
    std::transform(tail.begin(), tail.end(), tail.begin(), [](unsigned char c) { return char(std::tolower(c)); });
    return tail == suffix;
}

uint32_t hash32(uint32_t x) // MathUtil.hpp:120-128
{
    x = ~x + (x << 15);
    x = x ^ (x >> 12);
    x = x + (x << 2);
    x = x ^ (x >> 4);
    x = x*2057;
    x = x ^ (x >> 16);
    return x;
}

std::string Scene::assetDir()
{
    if (const char *env = std::getenv("TUNGSTEN_AMD_ASSETS")) return env;
#ifdef TUNGSTEN_AMD_DEFAULT_ASSETS
    return TUNGSTEN_AMD_DEFAULT_ASSETS;
#else
    return "assets";
#endif
}

// ---------------------------------------------------------------------------
// transform parsing.  Behaviour = the reference's fromJson(Mat4f) (JsonPtr.cpp:91-186): a 16-number array, or an object
// {position, look_at | x_axis / y_axis / z_axis / up, scale, rotation}.  The arithmetic order is the reference's (the camera and every
// primitive frame come out of here, bit for bit); the structure is this loader's: a frame of three axes, a bit mask of the axes the file
// gives, and one table saying in which order the axes are made orthonormal.
// ---------------------------------------------------------------------------
namespace {

struct AxisFrame {
    Vec3 axis[3] = {Vec3(1.0f, 0.0f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), Vec3(0.0f, 0.0f, 1.0f)};
    unsigned given = 0; // bit k: axis k comes from the file

    bool read(const Json &v, const char *key, int k)
    {
        float t[3];
        if (!v.getVec(key, t, 3)) return false;
        axis[k] = Vec3(t[0], t[1], t[2]);
        given |= 1u << k;
        return true;
    }

    // Orthonormal frame from (first, second, third): `first` keeps its direction, `second` loses its component along it, `third` along both.
    // An axis that degenerates (squared length under 1e-5) is replaced: the second by a perpendicular of the first built from the
    // coordinate axis the first is less aligned with, the third by first x second.
    void orthonormalise(int first, int second, int third)
    {
        Vec3 &a = axis[first], &b = axis[second], &c = axis[third];
        a.normalize();
        b -= a*a.dot(b);
        if (b.lengthSq() < 1e-5) {
            Vec3 helper = std::abs(a.x) > std::abs(a.y) ? Vec3(0.0f, 1.0f, 0.0f) : Vec3(1.0f, 0.0f, 0.0f);
            b = a.cross(helper).normalized();
        } else {
            b.normalize();
        }
        c -= a*a.dot(c);
        c -= b*b.dot(c);
        if (c.lengthSq() < 1e-5) c = a.cross(b);
        else c.normalize();
    }
};

// index = mask of given axes (x = 1, y = 2, z = 4) -> the order (first, second, third) in which they are orthonormalised: given axes lead,
// z before y before x among equals, except that a lone x is followed by z (JsonPtr.cpp:134-144)
constexpr int kAxisOrder[8][3] = {
    {2, 1, 0}, // nothing given
    {0, 2, 1}, // x
    {1, 2, 0}, // y
    {1, 0, 2}, // x y
    {2, 1, 0}, // z
    {2, 0, 1}, // x z
    {2, 1, 0}, // y z
    {2, 1, 0}, // x y z
};

Mat4 parseTransform(const Json &v)
{
    if (v.isArray()) {
        if (v.size() != 16) throw JsonLoadException("Trying to parse a matrix, but this array has the wrong size");
        Mat4 m;
        for (int i = 0; i < 16; ++i) m.a[i] = v[i].asFloat();
        return m;
    }
    if (!v.isObject()) throw JsonLoadException("Parameter has wrong type: Expecting a matrix value here");

    float p[3] = {0.0f, 0.0f, 0.0f};
    v.getVec("position", p, 3);
    const Vec3 origin(p[0], p[1], p[2]);

    AxisFrame f;
    if (f.read(v, "look_at", 2)) f.axis[2] = f.axis[2] - origin; // the view direction; an explicit z_axis below overrides it
    f.read(v, "up", 1);
    f.read(v, "x_axis", 0);
    f.read(v, "y_axis", 1);
    f.read(v, "z_axis", 2);

    const int *order = kAxisOrder[f.given];
    f.orthonormalise(order[0], order[1], order[2]);
    if (f.axis[0].cross(f.axis[1]).dot(f.axis[2]) < 0.0f) { // left-handed: flip the first axis the file did not give (z if it gave all three)
        int flip = 2;
        for (int k = 0; k < 2; ++k)
            if (!(f.given & (1u << k))) { flip = k; break; }
        f.axis[flip] = -f.axis[flip];
    }
    float t[3];
    if (v.getVec("scale", t, 3))
        for (int k = 0; k < 3; ++k) f.axis[k] *= t[k];
    if (v.getVec("rotation", t, 3)) {
        const Mat4 turn = Mat4::rotYXZ(Vec3(t[0], t[1], t[2]));
        for (int k = 0; k < 3; ++k) f.axis[k] = turn*f.axis[k];
    }
    const Vec3 &x = f.axis[0], &y = f.axis[1], &z = f.axis[2];
    return Mat4(x[0], y[0], z[0], origin[0], x[1], y[1], z[1], origin[1], x[2], y[2], z[2], origin[2], 0.0f, 0.0f, 0.0f, 1.0f);
}

} // namespace

// ---------------------------------------------------------------------------
// resources
// ---------------------------------------------------------------------------
bool loadWo3(const std::string &path, std::vector<float> &verts, std::vector<uint32_t> &tris)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    // counts come from the file: nothing is allocated before they have been checked against the file's size
    std::fseek(f, 0, SEEK_END);
    const uint64_t fileBytes = uint64_t(std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    uint64_t nv = 0, nt = 0;
    bool ok = std::fread(&nv, 8, 1, f) == 1;
    if (ok && nv < (1ull << 32) && 16 + nv*32 <= fileBytes) {
        verts.resize(size_t(nv)*8);
        ok = nv == 0 || std::fread(verts.data(), 32, size_t(nv), f) == size_t(nv);
    } else {
        ok = false;
    }
    ok = ok && std::fread(&nt, 8, 1, f) == 1;
    if (ok && nt < (1ull << 32) && 16 + nv*32 + nt*16 <= fileBytes) {
        tris.resize(size_t(nt)*4);
        ok = nt == 0 || std::fread(tris.data(), 16, size_t(nt), f) == size_t(nt);
    } else {
        ok = false;
    }
    std::fclose(f);
    return ok;
}

// Radiance RGBE (.hdr) decode, same arithmetic as the reference's vendored
// stb_image (thirdparty/stbi/stb_image.c:5588-5613): f = ldexp(1, e-136), c*f.
static void rgbeToFloat(const uint8_t *rgbe, float *out)
{
    if (rgbe[3] != 0) {
        float f1 = float(std::ldexp(1.0f, int(rgbe[3]) - (128 + 8)));
        out[0] = rgbe[0]*f1;
        out[1] = rgbe[1]*f1;
        out[2] = rgbe[2]*f1;
    } else {
        out[0] = out[1] = out[2] = 0.0f;
    }
}

bool loadRadianceHdr(const std::string &path, int &w, int &h, std::vector<float> &rgb)
{
    std::string data;
    try {
        data = readFile(path);
    } catch (...) {
        return false;
    }
    size_t pos = 0;
    auto readLine = [&](std::string &line) {
        line.clear();
        while (pos < data.size() && data[pos] != '\n') line += data[pos++];
        if (pos < data.size()) ++pos;
    };
    std::string line;
    readLine(line);
    if (line != "#?RADIANCE" && line != "#?RGBE") return false;
    bool valid = false;
    while (true) {
        readLine(line);
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") valid = true;
        if (pos >= data.size()) return false;
    }
    if (!valid) return false;
    readLine(line);
    if (std::sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2) return false;
    if (w <= 0 || h <= 0) return false;
    const uint8_t *p = reinterpret_cast<const uint8_t *>(data.data());
    size_t n = data.size();
    // the resolution comes from the file: a scanline costs at least its 4-byte header (run-length form) or 4 bytes per pixel (flat form),
    // so a header that promises more rows or pixels than the file can hold is refused before anything is allocated
    const bool flat = w < 8 || w >= 32768;
    const size_t pixels = size_t(w)*size_t(h), available = n - pos;
    if (flat ? pixels > available/4 : size_t(h) > available/4) return false;
    rgb.resize(pixels*3);
    if (flat) {
        // flat data
        for (size_t i = 0; i < pixels; ++i) {
            if (pos + 4 > n) return false;
            rgbeToFloat(p + pos, &rgb[size_t(i)*3]);
            pos += 4;
        }
        return true;
    }
    std::vector<uint8_t> scan(size_t(w)*4);
    for (int j = 0; j < h; ++j) {
        if (pos + 4 > n) return false;
        uint8_t c1 = p[pos], c2 = p[pos + 1], len = p[pos + 2];
        if (c1 != 2 || c2 != 2 || (len & 0x80)) {
            // not RLE: the whole image is flat (stb falls back for the full image from this pixel on
            // only when j == 0)
            if (j != 0 || pixels > (n - pos)/4) return false;
            for (size_t i = 0; i < pixels; ++i) {
                if (pos + 4 > n) return false;
                rgbeToFloat(p + pos, &rgb[size_t(i)*3]);
                pos += 4;
            }
            return true;
        }
        int slen = (int(len) << 8) | p[pos + 3];
        pos += 4;
        if (slen != w) return false;
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                if (pos >= n) return false;
                uint8_t count = p[pos++];
                if (count > 128) {
                    if (pos >= n) return false;
                    uint8_t value = p[pos++];
                    count -= 128;
                    if (i + count > w) return false;
                    for (int z = 0; z < count; ++z) scan[size_t(i++)*4 + k] = value;
                } else {
                    if (i + count > w || pos + count > n) return false;
                    for (int z = 0; z < count; ++z) scan[size_t(i++)*4 + k] = p[pos++];
                }
            }
        }
        for (int i = 0; i < w; ++i) rgbeToFloat(&scan[size_t(i)*4], &rgb[(size_t(j)*w + i)*3]);
    }
    return true;
}

bool savePfm(const std::string &path, const float *img, int w, int h, int channels) // ImageIO.cpp:528-545
{
    if (channels != 1 && channels != 3) return false;
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "%s\n%d %d\n-1\n", channels == 1 ? "Pf" : "PF", w, h);
    for (int y = h - 1; y >= 0; --y) std::fwrite(img + size_t(y)*w*channels, sizeof(float), size_t(w)*channels, f);
    std::fclose(f);
    return true;
}

bool loadPfm(const std::string &path, int &w, int &h, std::vector<float> &rgb)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    float scale = 0.0f;
    if (std::fscanf(f, "%2s %d %d %f", magic, &w, &h, &scale) != 4 || std::strcmp(magic, "PF") != 0 || scale >= 0.0f) {
        std::fclose(f);
        return false;
    }
    std::fgetc(f);
    const long dataStart = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const uint64_t dataBytes = uint64_t(std::ftell(f) - dataStart);
    std::fseek(f, dataStart, SEEK_SET);
    if (w <= 0 || h <= 0 || uint64_t(w)*uint64_t(h) > dataBytes/12) { // the header promises more pixels than the file holds
        std::fclose(f);
        return false;
    }
    rgb.resize(size_t(w)*h*3);
    bool ok = true;
    for (int y = h - 1; y >= 0 && ok; --y) ok = std::fread(&rgb[size_t(y)*w*3], sizeof(float), size_t(w)*3, f) == size_t(w)*3;
    std::fclose(f);
    return ok;
}

// ---------------------------------------------------------------------------
// PNG decoding (8-bit, non-interlaced: what texture files are; the reference goes through lodepng_decode_memory(..., LCT_RGBA, 8),
// io/ImageIO.cpp:386-407): zlib inflate (stored, fixed and dynamic Huffman blocks) + the five scanline filters -> RGBA8
// ---------------------------------------------------------------------------
namespace {
struct BitReader {
    const uint8_t *p;
    size_t n, pos = 0;
    uint32_t bitBuf = 0;
    int bitCnt = 0;
    bool fail = false;
    uint32_t bits(int k)
    {
        while (bitCnt < k) {
            if (pos >= n) { fail = true; return 0; }
            bitBuf |= uint32_t(p[pos++]) << bitCnt;
            bitCnt += 8;
        }
        uint32_t v = bitBuf & ((k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1u));
        bitBuf >>= k;
        bitCnt -= k;
        return v;
    }
    void alignByte() { bitBuf = 0; bitCnt = 0; }
};
struct Huffman { // canonical code, decoded bit by bit (texture files are small)
    uint16_t count[16] = {0}, symbol[320] = {0};
    void build(const uint8_t *lengths, int n)
    {
        for (int i = 0; i < 16; ++i) count[i] = 0;
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; ++i) offs[i + 1] = uint16_t(offs[i] + count[i]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
    }
    int decode(BitReader &br) const
    {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; ++len) {
            code |= int(br.bits(1));
            if (br.fail) return -1;
            int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};
bool inflateZlib(const std::vector<uint8_t> &in, std::vector<uint8_t> &out)
{
    if (in.size() < 6) return false;
    BitReader br{in.data() + 2, in.size() - 2}; // skip the zlib header (CMF, FLG); the Adler checksum at the end is not verified
    static const uint16_t lenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t lenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t distBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t distExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    bool last = false;
    while (!last) {
        last = br.bits(1) != 0;
        const uint32_t type = br.bits(2);
        if (br.fail) return false;
        if (type == 0) {
            br.alignByte();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8);
            br.pos += 4;
            if (br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
            continue;
        }
        if (type == 3) return false;
        Huffman lit, dist;
        uint8_t lengths[320];
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lengths[i] = 8;
            for (int i = 144; i < 256; ++i) lengths[i] = 9;
            for (int i = 256; i < 280; ++i) lengths[i] = 7;
            for (int i = 280; i < 288; ++i) lengths[i] = 8;
            lit.build(lengths, 288);
            for (int i = 0; i < 30; ++i) lengths[i] = 5;
            dist.build(lengths, 30);
        } else {
            const int nlen = int(br.bits(5)) + 257, ndist = int(br.bits(5)) + 1, ncode = int(br.bits(4)) + 4;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t cl[19] = {0};
            for (int i = 0; i < ncode; ++i) cl[order[i]] = uint8_t(br.bits(3));
            Huffman codeLen;
            codeLen.build(cl, 19);
            int i = 0;
            while (i < nlen + ndist) {
                int sym = codeLen.decode(br);
                if (sym < 0 || br.fail) return false;
                if (sym < 16) { lengths[i++] = uint8_t(sym); continue; }
                int rep = 0;
                uint8_t val = 0;
                if (sym == 16) { if (i == 0) return false; val = lengths[i - 1]; rep = 3 + int(br.bits(2)); }
                else if (sym == 17) rep = 3 + int(br.bits(3));
                else rep = 11 + int(br.bits(7));
                if (i + rep > nlen + ndist) return false;
                while (rep--) lengths[i++] = val;
            }
            lit.build(lengths, nlen);
            dist.build(lengths + nlen, ndist);
        }
        while (true) {
            int sym = lit.decode(br);
            if (sym < 0 || br.fail) return false;
            if (sym < 256) { out.push_back(uint8_t(sym)); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) return false;
            const uint32_t len = lenBase[sym] + br.bits(lenExtra[sym]);
            const int ds = dist.decode(br);
            if (ds < 0 || ds >= 30) return false;
            const uint32_t d = distBase[ds] + br.bits(distExtra[ds]);
            if (br.fail || d > out.size()) return false;
            const size_t from = out.size() - d;
            for (uint32_t k = 0; k < len; ++k) out.push_back(out[from + k]);
        }
    }
    return true;
}
} // namespace

// ---- .bmp / .tga: the files the reference hands to stb_image (io/ImageIO.cpp:386-407,489-490: stbi_load_from_callbacks(..., 4)) ------------------
// Both decode to RGBA8, top row first.  Covered: what image tools write -- BMP with 4- or 8-bit palettes, 24 bits, 32 bits in B,G,R,A order, bottom-up
// or top-down, uncompressed; TGA true-colour (24 / 32 bits) and grey (8 bits), raw or run-length encoded, either row order.  The rarer forms
// (16-bit pixels, custom bit masks, RLE bitmaps, colour-mapped TGAs) are refused by name rather than decoded approximately.
namespace {
struct ByteReader {
    const std::string &data;
    size_t pos = 0;
    bool ok = true;
    explicit ByteReader(const std::string &d) : data(d) {}
    uint32_t u8() { if (pos >= data.size()) { ok = false; return 0; } return uint8_t(data[pos++]); }
    uint32_t u16() { uint32_t a = u8(); return a | (u8() << 8); }
    uint32_t u32() { uint32_t a = u16(); return a | (u16() << 16); }
    void skip(size_t n) { if (n > data.size() - pos) { ok = false; pos = data.size(); } else pos += n; }
};
}

bool loadBmp(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba, std::string &why)
{
    std::string data;
    try { data = readFile(path); } catch (...) { why = "cannot be read"; return false; }
    ByteReader in(data);
    if (in.u8() != 'B' || in.u8() != 'M') { why = "is not a BMP file"; return false; }
    in.skip(8);
    const uint32_t offset = in.u32(), headerSize = in.u32();
    int32_t width, height;
    uint32_t bpp, compression = 0;
    if (headerSize == 12) {
        width = int32_t(in.u16()); height = int32_t(in.u16());
        in.skip(2);
        bpp = in.u16();
    } else if (headerSize == 40 || headerSize == 56 || headerSize == 108 || headerSize == 124) {
        width = int32_t(in.u32()); height = int32_t(in.u32());
        in.skip(2);
        bpp = in.u16();
        compression = in.u32();
        in.skip(20);
        uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
        if (headerSize == 40 && compression == 3) { mr = in.u32(); mg = in.u32(); mb = in.u32(); }
        else if (headerSize > 40) { mr = in.u32(); mg = in.u32(); mb = in.u32(); ma = in.u32(); in.skip(headerSize - 56); }
        if (compression == 1 || compression == 2) { why = "is run-length encoded"; return false; }
        if (compression == 3 && !(bpp == 32 && mr == 0x00FF0000u && mg == 0x0000FF00u && mb == 0x000000FFu && (ma == 0xFF000000u || ma == 0))) { why = "uses custom bit masks"; return false; }
    } else { why = "has an unknown header"; return false; }
    if (!in.ok || width <= 0 || height == 0) { why = "is truncated or has no pixels"; return false; }
    const bool bottomUp = height > 0;
    w = width;
    h = height < 0 ? -height : height;
    if (bpp != 4 && bpp != 8 && bpp != 24 && bpp != 32) { why = "has " + std::to_string(bpp) + " bits per pixel"; return false; }
    const size_t rowBytes = ((size_t(w)*bpp + 31)/32)*4;
    if (size_t(h) > (data.size() - std::min(data.size(), size_t(offset)))/std::max<size_t>(rowBytes, 1)) { why = "is truncated"; return false; }
    uint8_t palette[256][3] = {};
    if (bpp <= 8) {
        const size_t entry = headerSize == 12 ? 3 : 4, count = (offset - 14 - headerSize)/entry;
        if (count == 0 || count > 256) { why = "has a corrupt palette"; return false; }
        for (size_t i = 0; i < count; ++i) {
            palette[i][2] = uint8_t(in.u8()); palette[i][1] = uint8_t(in.u8()); palette[i][0] = uint8_t(in.u8());
            if (entry == 4) in.u8();
        }
    }
    rgba.assign(size_t(w)*h*4, 255);
    bool anyAlpha = false;
    for (int row = 0; row < h; ++row) {
        const size_t at = size_t(offset) + size_t(row)*rowBytes;
        uint8_t *out = &rgba[size_t(bottomUp ? h - 1 - row : row)*w*4];
        for (int x = 0; x < w; ++x) {
            const uint8_t *px = reinterpret_cast<const uint8_t *>(data.data()) + at;
            if (bpp == 24) { out[x*4] = px[x*3 + 2]; out[x*4 + 1] = px[x*3 + 1]; out[x*4 + 2] = px[x*3]; }
            else if (bpp == 32) { out[x*4] = px[x*4 + 2]; out[x*4 + 1] = px[x*4 + 1]; out[x*4 + 2] = px[x*4]; out[x*4 + 3] = px[x*4 + 3]; anyAlpha = anyAlpha || px[x*4 + 3]; }
            else {
                const uint32_t index = bpp == 8 ? px[x] : ((x & 1) ? (px[x/2] & 15u) : (px[x/2] >> 4));
                out[x*4] = palette[index][0]; out[x*4 + 1] = palette[index][1]; out[x*4 + 2] = palette[index][2];
            }
        }
    }
    if (bpp == 32 && !anyAlpha) for (size_t i = 3; i < rgba.size(); i += 4) rgba[i] = 255; // an all-zero alpha channel means "no alpha"
    return true;
}

bool loadTga(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba, std::string &why)
{
    std::string data;
    try { data = readFile(path); } catch (...) { why = "cannot be read"; return false; }
    ByteReader in(data);
    const uint32_t idLength = in.u8(), colourMapped = in.u8();
    uint32_t type = in.u8();
    in.skip(5);            // colour map specification
    in.skip(4);            // x / y origin
    w = int(in.u16());
    h = int(in.u16());
    const uint32_t bpp = in.u8(), descriptor = in.u8();
    if (!in.ok || w <= 0 || h <= 0) { why = "is truncated or has no pixels"; return false; }
    if (colourMapped) { why = "is colour-mapped"; return false; }
    const bool rle = type >= 8;
    if (rle) type -= 8;
    if (!((type == 2 && (bpp == 24 || bpp == 32)) || (type == 3 && bpp == 8))) { why = "has pixel format " + std::to_string(type) + "/" + std::to_string(bpp); return false; }
    const size_t bytes = bpp/8;
    in.skip(idLength);
    if (size_t(w)*h > (data.size() - in.pos)*(rle ? 128 : 1)/ (rle ? 1 : bytes)) { why = "is truncated"; return false; }
    const bool topDown = (descriptor >> 5) & 1u;
    rgba.assign(size_t(w)*h*4, 255);
    uint8_t px[4] = {0, 0, 0, 255};
    uint32_t run = 0;
    bool repeating = false;
    for (size_t i = 0; i < size_t(w)*h; ++i) {
        bool read = true;
        if (rle) {
            if (run == 0) { const uint32_t cmd = in.u8(); run = 1 + (cmd & 127u); repeating = (cmd >> 7) != 0; }
            else if (repeating) read = false;
            --run;
        }
        if (read) for (size_t k = 0; k < bytes; ++k) px[k] = uint8_t(in.u8());
        if (!in.ok) { why = "is truncated"; return false; }
        const size_t row = i/size_t(w), x = i % size_t(w);
        uint8_t *out = &rgba[((topDown ? row : size_t(h) - 1 - row)*w + x)*4];
        if (bytes == 1) { out[0] = out[1] = out[2] = px[0]; }
        else { out[0] = px[2]; out[1] = px[1]; out[2] = px[0]; if (bytes == 4) out[3] = px[3]; }
    }
    return true;
}

// ---- .jpg: baseline (sequential, Huffman, 8-bit) JPEG, decoded to the bytes the reference gets from stb_image --------------------------------------
// The reference build has no libjpeg, so ImageIO hands .jpg files to stb_image too (io/ImageIO.cpp:340-377,489-490).  A JPEG's pixels depend on the
// decoder's arithmetic, so this one keeps stb_image's (thirdparty/stbi/stb_image.c; its SSE2 kernels are documented bit-identical to the scalar forms
// restated here): the integer IDCT derived from jidctint with 12-bit constants and two extra bits between the passes (:1404-1502), chroma upsampling by
// the 3:1 "JFIF-centred" filters across block boundaries (:2343-2415) driven by the line0 / line1 stepping of load_jpeg_image (:2823-2850), and the
// reduced-precision YCbCr -> RGB (:2577-2601).  Entropy decoding is the JPEG specification's (canonical Huffman codes, receive-and-extend, restart
// intervals; for progressive files the spectral-selection / successive-approximation scans of annex G with stb's coefficient handling, :1247-1393,
// dequantised and transformed once at the end); a marker inside the entropy-coded data ends the bit supply with zero bits, as in stb.
namespace {
class JpegDecoder {
public:
    explicit JpegDecoder(const std::string &bytes) : _d(bytes) {}

    bool decode(int &w, int &h, std::vector<uint8_t> &rgba, std::string &why)
    {
        try {
            parse();
            assemble(rgba);
            w = _width;
            h = _height;
            return true;
        } catch (const char *msg) {
            why = msg;
            return false;
        }
    }

private:
    struct Huffman {
        uint8_t size[257] = {};
        uint16_t code[256] = {};
        uint8_t value[256] = {};
        int count = 0;
        bool defined = false;
    };
    struct Component {
        int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
        uint32_t dcPred = 0; // two's complement, modulo 2^32: a damaged file can drive the predictor anywhere, and only its low 16 bits are kept
        int x = 0, y = 0, w2 = 0, h2 = 0;
        std::vector<uint8_t> data;
        std::vector<short> coeff; // progressive files: the coefficients of every block (w2/8 x h2/8 blocks), refined scan by scan
    };

    uint32_t byte() { if (_pos >= _d.size()) throw "is truncated"; return uint8_t(_d[_pos++]); }
    uint32_t be16() { uint32_t a = byte(); return (a << 8) | byte(); }
    uint32_t nextMarker()
    {
        if (_pendingMarker != 0xFF) { uint32_t m = _pendingMarker; _pendingMarker = 0xFF; return m; }
        uint32_t x = byte();
        if (x != 0xFF) return 0xFF; // "no marker"
        while (x == 0xFF) x = byte();
        return x;
    }

    // ---- entropy-coded segment: bits, MSB first; a marker stops the supply (zeros from then on) ----
    void resetScan()
    {
        _bitBuffer = 0; _bitCount = 0; _noMore = false;
        for (Component &c : _comp) c.dcPred = 0;
        _pendingMarker = 0xFF;
        _todo = _restartInterval ? _restartInterval : 0x7fffffff;
        _eobRun = 0;
    }
    void fill()
    {
        do {
            uint32_t b = 0;
            if (!_noMore) {
                b = _pos < _d.size() ? uint8_t(_d[_pos++]) : 0;
                if (b == 0xFF) {
                    const uint32_t c = _pos < _d.size() ? uint8_t(_d[_pos++]) : 0;
                    if (c != 0) { _pendingMarker = c; _noMore = true; return; }
                }
            }
            _bitBuffer |= b << (24 - _bitCount);
            _bitCount += 8;
        } while (_bitCount <= 24);
    }
    int bits(int n)
    {
        if (n == 0) return 0;
        if (_bitCount < n) fill();
        const uint32_t v = _bitBuffer >> (32 - n);
        _bitBuffer <<= n;
        _bitCount -= n;
        return int(v);
    }
    int symbol(const Huffman &h)
    {
        if (!h.defined) throw "uses an undefined Huffman table";
        if (_bitCount < 16) fill();
        uint32_t code = 0;
        int k = 0;
        for (int len = 1; len <= 16; ++len) {
            code = (code << 1) | (_bitBuffer >> 31);
            _bitBuffer <<= 1;
            --_bitCount;
            while (k < h.count && h.size[k] == len) {
                if (h.code[k] == code) return h.value[k];
                ++k;
            }
        }
        throw "has a corrupt Huffman code";
    }
    int receiveExtend(int n) // F.2.2.1: the n-bit value, negative range when its first bit is clear
    {
        if (n == 0) return 0;
        const int v = bits(n);
        return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
    }

    void decodeBlock(Component &c, short *block)
    {
        static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        const uint8_t *q = _quant[c.tq];
        std::memset(block, 0, 64*sizeof(short));
        const int t = symbol(_dc[c.hd]);
        if (t > 15) throw "has a corrupt DC code";
        c.dcPred += uint32_t(receiveExtend(t));
        block[0] = short(c.dcPred*uint32_t(q[0]));
        for (int k = 1; k < 64;) {
            const int rs = symbol(_ac[c.ha]), run = rs >> 4, size = rs & 15;
            if (size == 0) {
                if (rs != 0xF0) break; // end of block
                k += 16;
            } else {
                k += run;
                if (k > 63) throw "has a corrupt coefficient run";
                const int pos = zigzag[k++];
                block[pos] = short(receiveExtend(size)*q[pos]); // the table is stored by position, not in zigzag order
            }
        }
    }

    // progressive scans (annex G): a DC scan sets or refines block[0], an AC scan a band [specStart, specEnd] of one component
    void progressiveDc(Component &c, short *block)
    {
        if (_specEnd != 0) throw "mixes DC and AC coefficients in a progressive scan";
        if (_succHigh == 0) {
            std::memset(block, 0, 64*sizeof(short));
            const int t = symbol(_dc[c.hd]);
            if (t > 15) throw "has a corrupt DC code";
            c.dcPred += uint32_t(receiveExtend(t));
            block[0] = short(c.dcPred << _succLow);
        } else if (bits(1)) {
            block[0] += short(1 << _succLow);
        }
    }
    void refineBit(short &coefficient, short bit) // one correction bit for a coefficient that is already non-zero
    {
        if (bits(1) && (coefficient & bit) == 0) coefficient = short(coefficient > 0 ? coefficient + bit : coefficient - bit);
    }
    void progressiveAc(Component &c, short *block)
    {
        static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        if (_specStart == 0) throw "mixes DC and AC coefficients in a progressive scan";
        const Huffman &h = _ac[c.ha];
        if (_succHigh == 0) { // first pass over the band
            if (_eobRun) { --_eobRun; return; }
            int k = _specStart;
            do {
                const int rs = symbol(h), run = rs >> 4, size = rs & 15;
                if (size == 0) {
                    if (run < 15) { _eobRun = (1 << run) + (run ? bits(run) : 0) - 1; break; }
                    k += 16;
                } else {
                    k += run;
                    if (k > 63) throw "has a corrupt coefficient run";
                    block[zigzag[k++]] = short(receiveExtend(size)*(1 << _succLow));
                }
            } while (k <= _specEnd);
            return;
        }
        const short bit = short(1 << _succLow); // refinement pass
        if (_eobRun) {
            --_eobRun;
            for (int k = _specStart; k <= _specEnd; ++k) {
                short &p = block[zigzag[k]];
                if (p != 0) refineBit(p, bit);
            }
            return;
        }
        int k = _specStart;
        do {
            const int rs = symbol(h);
            int run = rs >> 4, value = rs & 15;
            if (value == 0) {
                if (run < 15) { _eobRun = (1 << run) - 1 + (run ? bits(run) : 0); run = 64; } // the rest of the band: corrections only
            } else {
                if (value != 1) throw "has a corrupt refinement code";
                value = bits(1) ? bit : -bit;
            }
            while (k <= _specEnd) { // skip `run` zero coefficients, correcting the non-zero ones on the way; then place the new one
                short &p = block[zigzag[k++]];
                if (p != 0) refineBit(p, bit);
                else if (run == 0) { p = short(value); break; }
                else --run;
            }
        } while (k <= _specEnd);
    }
    void finishProgressive() // every block: dequantise (16-bit product, as stb's in-place multiply), then the transform
    {
        for (Component &c : _comp) {
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3, coeffW = c.w2/8;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    short *block = &c.coeff[(size_t(j)*coeffW + i)*64];
                    for (int k = 0; k < 64; ++k) block[k] = short(block[k]*_quant[c.tq][k]);
                    idct(&c.data[size_t(c.w2)*j*8 + size_t(i)*8], c.w2, block);
                }
        }
    }

    // The 8x8 inverse DCT with stb_image's numbers (stbi__idct_block, :1404-1502: the jidctint factorisation with 12-bit constants, columns first with two
    // guard bits kept, then rows with the +128 level shift folded into the rounding term).  One dimension splits into the half fed by the even inputs
    // and the half fed by the odd ones: output k is their sum, output 7 - k their difference.  Integer arithmetic throughout, so only the constants
    // (rounded as stb rounds them, the negative ones included) and the two rounding terms decide the bytes -- tests/test_ref_kat6.py compares them all.
    static int fixed12(double x) { return int(x*4096 + 0.5); }
    static void oneDimension(const int in[8], int evenHalf[4], int oddHalf[4])
    {
        const int rotation = (in[2] + in[6])*fixed12(0.5411961f);
        const int outer = rotation + in[2]*fixed12(0.765366865f), inner = rotation + in[6]*fixed12(-1.847759065f);
        const int sum04 = (in[0] + in[4])*4096, diff04 = (in[0] - in[4])*4096;
        evenHalf[0] = sum04 + outer; evenHalf[1] = diff04 + inner; evenHalf[2] = diff04 - inner; evenHalf[3] = sum04 - outer;
        const int pair73 = in[7] + in[3], pair51 = in[5] + in[1], pair71 = in[7] + in[1], pair53 = in[5] + in[3];
        const int common = (pair73 + pair51)*fixed12(1.175875602f);
        const int via71 = common + pair71*fixed12(-0.899976223f), via53 = common + pair53*fixed12(-2.562915447f);
        const int via73 = pair73*fixed12(-1.961570560f), via51 = pair51*fixed12(-0.390180644f);
        oddHalf[0] = in[1]*fixed12(1.501321110f) + via71 + via51;
        oddHalf[1] = in[3]*fixed12(3.072711026f) + via53 + via73;
        oddHalf[2] = in[5]*fixed12(2.053119869f) + via53 + via51;
        oddHalf[3] = in[7]*fixed12(0.298631336f) + via71 + via73;
    }
    static uint8_t clampByte(int x) { return uint8_t(x < 0 ? 0 : (x > 255 ? 255 : x)); }
    static void idct(uint8_t *out, int stride, const short *coeff)
    {
        int columns[64]; // after the column pass, scaled by 4
        for (int x = 0; x < 8; ++x) {
            int in[8];
            bool onlyDc = true;
            for (int k = 0; k < 8; ++k) { in[k] = coeff[x + 8*k]; onlyDc = onlyDc && (k == 0 || in[k] == 0); }
            if (onlyDc) { // stb's shortcut: a column of one value
                for (int k = 0; k < 8; ++k) columns[x + 8*k] = in[0]*4;
                continue;
            }
            int even[4], odd[4];
            oneDimension(in, even, odd);
            for (int k = 0; k < 4; ++k) {
                columns[x + 8*k] = (even[k] + 512 + odd[k]) >> 10;
                columns[x + 8*(7 - k)] = (even[k] + 512 - odd[k]) >> 10;
            }
        }
        const int rounding = 65536 + (128 << 17); // half an output unit plus the level shift
        for (int y = 0; y < 8; ++y) {
            int even[4], odd[4];
            oneDimension(columns + 8*y, even, odd);
            uint8_t *row = out + size_t(y)*stride;
            for (int k = 0; k < 4; ++k) {
                row[k] = clampByte((even[k] + rounding + odd[k]) >> 17);
                row[7 - k] = clampByte((even[k] + rounding - odd[k]) >> 17);
            }
        }
    }

    // ---- segments ----
    void parse()
    {
        if (nextMarker() != 0xD8) throw "is not a JPEG file";
        bool haveFrame = false;
        for (;;) {
            uint32_t m = nextMarker();
            while (m == 0xFF) { if (_pos >= _d.size()) throw "ends before its image data"; m = nextMarker(); }
            if (m == 0xD9) break;                                        // EOI
            if (m == 0xC0 || m == 0xC1 || m == 0xC2) { _progressive = m == 0xC2; frameHeader(); haveFrame = true; continue; }
            if (m == 0xDA) {                                             // SOS
                if (!haveFrame) throw "has a scan before its frame header";
                scan();
                if (_pendingMarker == 0xFF) {                            // padding up to the next marker
                    while (_pos < _d.size()) {
                        const uint32_t x = byte();
                        if (x == 0xFF) { _pendingMarker = _pos < _d.size() ? byte() : 0xD9; break; }
                        if (x != 0) throw "has junk before a marker";
                    }
                    if (_pendingMarker == 0xFF) break;                   // the file ends without EOI: what was decoded stands
                }
                continue;
            }
            if (m == 0xDD) { if (be16() != 4) throw "has a corrupt DRI segment"; _restartInterval = int(be16()); continue; }
            if (m == 0xDB) { quantTables(); continue; }
            if (m == 0xC4) { huffmanTables(); continue; }
            if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) { const uint32_t n = be16(); if (n < 2 || n - 2 > _d.size() - _pos) throw "is truncated"; _pos += n - 2; continue; }
            throw "uses a JPEG feature outside the baseline process";
        }
        if (!haveFrame) throw "has no frame header";
        if (_progressive) finishProgressive();
    }
    void quantTables()
    {
        static const uint8_t zigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
        int left = int(be16()) - 2;
        while (left > 0) {
            const uint32_t q = byte();
            if (q >> 4) throw "has 16-bit quantisation tables";
            if ((q & 15) > 3) throw "has a corrupt DQT segment";
            for (int i = 0; i < 64; ++i) _quant[q & 15][zigzag[i]] = uint8_t(byte());
            left -= 65;
        }
        if (left != 0) throw "has a corrupt DQT segment";
    }
    void huffmanTables()
    {
        int left = int(be16()) - 2;
        while (left > 0) {
            const uint32_t q = byte();
            const int tc = int(q >> 4), th = int(q & 15);
            if (tc > 1 || th > 3) throw "has a corrupt DHT segment";
            int counts[16], n = 0;
            for (int i = 0; i < 16; ++i) { counts[i] = int(byte()); n += counts[i]; }
            if (n > 256) throw "has a corrupt DHT segment";
            Huffman &h = tc ? _ac[th] : _dc[th];
            h = Huffman();
            int k = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < counts[i]; ++j) h.size[k++] = uint8_t(i + 1);
            uint32_t code = 0;
            k = 0;
            for (int len = 1; len <= 16; ++len) {          // canonical codes, annex C
                while (k < n && h.size[k] == len) h.code[k++] = uint16_t(code++);
                if (code > (1u << len)) throw "has corrupt Huffman code lengths";
                code <<= 1;
            }
            for (int i = 0; i < n; ++i) h.value[i] = uint8_t(byte());
            h.count = n;
            h.defined = true;
            left -= 17 + n;
        }
        if (left != 0) throw "has a corrupt DHT segment";
    }
    void frameHeader()
    {
        const uint32_t len = be16();
        if (len < 11) throw "has a corrupt frame header";
        if (byte() != 8) throw "is not an 8-bit JPEG";
        _height = int(be16());
        _width = int(be16());
        if (_height == 0 || _width == 0) throw "has no pixels";
        const int n = int(byte());
        if (n != 1 && n != 3) throw "has neither one nor three components";
        if (len != uint32_t(8 + 3*n)) throw "has a corrupt frame header";
        if ((1 << 30)/_width/n < _height) throw "is too large";
        _comp.assign(size_t(n), Component());
        _rgbIds = 0;
        for (int i = 0; i < n; ++i) {
            Component &c = _comp[size_t(i)];
            c.id = int(byte());
            if (c.id != i + 1 && c.id != i) {
                if (c.id != "RGB"[i]) throw "has unusual component identifiers";
                ++_rgbIds;
            }
            const uint32_t q = byte();
            c.h = int(q >> 4); c.v = int(q & 15);
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4) throw "has corrupt sampling factors";
            c.tq = int(byte());
            if (c.tq > 3) throw "has a corrupt frame header";
        }
        _hMax = _vMax = 1;
        for (const Component &c : _comp) { _hMax = std::max(_hMax, c.h); _vMax = std::max(_vMax, c.v); }
        _mcuX = (_width + _hMax*8 - 1)/(_hMax*8);
        _mcuY = (_height + _vMax*8 - 1)/(_vMax*8);
        for (Component &c : _comp) {
            c.x = (_width*c.h + _hMax - 1)/_hMax;
            c.y = (_height*c.v + _vMax - 1)/_vMax;
            c.w2 = _mcuX*c.h*8;
            c.h2 = _mcuY*c.v*8;
            c.data.assign(size_t(c.w2)*c.h2, 0);
            if (_progressive) c.coeff.assign(size_t(c.w2/8)*(c.h2/8)*64, 0);
        }
    }
    void scan()
    {
        const uint32_t len = be16();
        const int n = int(byte());
        if (n < 1 || n > int(_comp.size())) throw "has a corrupt scan header";
        if (len != uint32_t(6 + 2*n)) throw "has a corrupt scan header";
        int order[4];
        for (int i = 0; i < n; ++i) {
            const int id = int(byte());
            const uint32_t q = byte();
            int which = 0;
            while (which < int(_comp.size()) && _comp[size_t(which)].id != id) ++which;
            if (which == int(_comp.size())) throw "names an unknown component in a scan";
            _comp[size_t(which)].hd = int(q >> 4);
            _comp[size_t(which)].ha = int(q & 15);
            if (_comp[size_t(which)].hd > 3 || _comp[size_t(which)].ha > 3) throw "has a corrupt scan header";
            order[i] = which;
        }
        _specStart = int(byte());
        _specEnd = int(byte());
        const uint32_t approx = byte();
        _succHigh = int(approx >> 4);
        _succLow = int(approx & 15);
        if (_progressive) {
            if (_specStart > 63 || _specEnd > 63 || _specStart > _specEnd || _succHigh > 13 || _succLow > 13) throw "has a corrupt scan header";
        } else if (_specStart != 0 || approx != 0) throw "has a corrupt scan header";
        resetScan();
        short block[64];
        auto restartDue = [&]() { // true: the scan goes on; false: it ends here (no restart marker where one is due)
            if (--_todo > 0) return true;
            if (_bitCount < 24) fill();
            if (!(_pendingMarker >= 0xD0 && _pendingMarker <= 0xD7)) return false;
            resetScan();
            return true;
        };
        if (n == 1) { // one component: its blocks in raster order, without MCU padding
            Component &c = _comp[size_t(order[0])];
            const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
            for (int j = 0; j < bh; ++j)
                for (int i = 0; i < bw; ++i) {
                    if (_progressive) {
                        short *stored = &c.coeff[(size_t(j)*(c.w2/8) + i)*64];
                        if (_specStart == 0) progressiveDc(c, stored); else progressiveAc(c, stored);
                    } else {
                        decodeBlock(c, block);
                        idct(&c.data[size_t(c.w2)*j*8 + size_t(i)*8], c.w2, block);
                    }
                    if (!restartDue()) return;
                }
            return;
        }
        for (int j = 0; j < _mcuY; ++j)
            for (int i = 0; i < _mcuX; ++i) {
                for (int k = 0; k < n; ++k) {
                    Component &c = _comp[size_t(order[k])];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x) {
                            if (_progressive) { // interleaved progressive scans carry DC coefficients only
                                progressiveDc(c, &c.coeff[(size_t(j*c.v + y)*(c.w2/8) + size_t(i*c.h + x))*64]);
                            } else {
                                decodeBlock(c, block);
                                idct(&c.data[size_t(c.w2)*((j*c.v + y)*8) + size_t(i*c.h + x)*8], c.w2, block);
                            }
                        }
                }
                if (!restartDue()) return;
            }
    }

    // ---- planes -> RGBA: load_jpeg_image's row loop ----
    void assemble(std::vector<uint8_t> &rgba)
    {
        const size_t n = _comp.size();
        struct Resampler { int hs, vs, wLores, ystep, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
        std::vector<Resampler> rs(n);
        for (size_t k = 0; k < n; ++k) {
            Resampler &r = rs[k];
            r.hs = _hMax/_comp[k].h;
            r.vs = _vMax/_comp[k].v;
            r.ystep = r.vs >> 1;
            r.wLores = (_width + r.hs - 1)/r.hs;
            r.ypos = 0;
            r.line0 = r.line1 = _comp[k].data.data();
            r.buf.assign(size_t(r.wLores)*r.hs + 8, 0);
        }
        rgba.assign(size_t(_width)*_height*4, 255);
        std::vector<const uint8_t *> row(n);
        for (int j = 0; j < _height; ++j) {
            for (size_t k = 0; k < n; ++k) {
                Resampler &r = rs[k];
                const bool bottom = r.ystep >= (r.vs >> 1);
                const uint8_t *nearRow = bottom ? r.line1 : r.line0, *farRow = bottom ? r.line0 : r.line1;
                row[k] = resample(r.buf.data(), nearRow, farRow, r.wLores, r.hs, r.vs);
                if (++r.ystep >= r.vs) {
                    r.ystep = 0;
                    r.line0 = r.line1;
                    if (++r.ypos < _comp[k].y) r.line1 += _comp[k].w2;
                }
            }
            uint8_t *out = &rgba[size_t(j)*_width*4];
            if (n == 1) {
                for (int i = 0; i < _width; ++i) out[i*4] = out[i*4 + 1] = out[i*4 + 2] = row[0][i];
            } else if (_rgbIds == 3) {
                for (int i = 0; i < _width; ++i) { out[i*4] = row[0][i]; out[i*4 + 1] = row[1][i]; out[i*4 + 2] = row[2][i]; }
            } else {
                // stbi__YCbCr_to_RGB_row: 20 fractional bits, constants rounded to 12 and shifted up, the Cb term of green truncated to 16 bits
                const int crR = int(1.40200f*4096.0f + 0.5f) << 8, crG = -(int(0.71414f*4096.0f + 0.5f) << 8), cbG = -(int(0.34414f*4096.0f + 0.5f) << 8),
                          cbB = int(1.77200f*4096.0f + 0.5f) << 8;
                for (int i = 0; i < _width; ++i) {
                    const int yFixed = (int(row[0][i]) << 20) + (1 << 19), cr = int(row[2][i]) - 128, cb = int(row[1][i]) - 128;
                    int r = yFixed + cr*crR;
                    int g = yFixed + cr*crG + int(uint32_t(cb*cbG) & 0xffff0000u);
                    int b = yFixed + cb*cbB;
                    r >>= 20; g >>= 20; b >>= 20;
                    out[i*4] = clampByte(r); out[i*4 + 1] = clampByte(g); out[i*4 + 2] = clampByte(b);
                }
            }
        }
    }
    static const uint8_t *resample(uint8_t *out, const uint8_t *nearRow, const uint8_t *farRow, int w, int hs, int vs)
    {
        if (hs == 1 && vs == 1) return nearRow;
        if (hs == 1 && vs == 2) { // two rows for one: 3:1 towards the nearer source row
            for (int i = 0; i < w; ++i) out[i] = uint8_t((3*nearRow[i] + farRow[i] + 2) >> 2);
            return out;
        }
        if (hs == 2 && vs == 1) { // two samples for one along the row: each is 3:1 between its source sample and the neighbour on its side, the ends stand
            if (w == 1) { out[0] = out[1] = nearRow[0]; return out; }
            for (int i = 0; i < w; ++i) {
                const int centre = 3*nearRow[i] + 2;
                out[2*i] = i == 0 ? nearRow[0] : uint8_t((centre + nearRow[i - 1]) >> 2);
                out[2*i + 1] = i == w - 1 ? nearRow[w - 1] : uint8_t((centre + nearRow[i + 1]) >> 2);
            }
            out[2*(w - 1)] = uint8_t((3*nearRow[w - 2] + nearRow[w - 1] + 2) >> 2); // stb weights the last pair the other way round (stb_image.c:2382); kept, the bytes are the contract
            return out;
        }
        if (hs == 2 && vs == 2) { // 2x2 for one: the vertical 3:1 blend of the two source rows first (kept at 4x), then the horizontal one on those sums
            auto column = [&](int i) { return 3*nearRow[i] + farRow[i]; };
            for (int i = 0; i < w; ++i) {
                const int here = column(i);
                out[2*i] = i == 0 ? uint8_t((here + 2) >> 2) : uint8_t((3*here + column(i - 1) + 8) >> 4);
                out[2*i + 1] = i == w - 1 ? uint8_t((here + 2) >> 2) : uint8_t((3*here + column(i + 1) + 8) >> 4);
            }
            return out;
        }
        for (int i = 0; i < w; ++i) for (int j = 0; j < hs; ++j) out[i*hs + j] = nearRow[i]; // other ratios: nearest
        return out;
    }

    const std::string &_d;
    size_t _pos = 0;
    uint32_t _pendingMarker = 0xFF, _bitBuffer = 0;
    int _bitCount = 0, _todo = 0, _restartInterval = 0, _eobRun = 0, _specStart = 0, _specEnd = 63, _succHigh = 0, _succLow = 0;
    bool _noMore = false, _progressive = false;
    int _width = 0, _height = 0, _hMax = 1, _vMax = 1, _mcuX = 0, _mcuY = 0, _rgbIds = 0;
    uint8_t _quant[4][64] = {};
    Huffman _dc[4], _ac[4];
    std::vector<Component> _comp;
};
}

bool loadJpeg(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba, std::string &why)
{
    std::string data;
    try { data = readFile(path); } catch (...) { why = "cannot be read"; return false; }
    JpegDecoder decoder(data);
    return decoder.decode(w, h, rgba, why);
}

bool loadPng(const std::string &path, int &w, int &h, std::vector<uint8_t> &rgba)
{
    std::vector<uint8_t> file;
    {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) return false;
        uint8_t buf[65536];
        size_t k;
        while ((k = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + k);
        std::fclose(f);
    }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return (uint32_t(file[o]) << 24) | (uint32_t(file[o + 1]) << 16) | (uint32_t(file[o + 2]) << 8) | uint32_t(file[o + 3]); };
    int depth = 0, colorType = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    for (size_t o = 8; o + 12 <= file.size();) {
        const uint32_t len = be32(o);
        const std::string tag(reinterpret_cast<const char *>(&file[o + 4]), 4);
        if (o + 12 + len > file.size()) return false;
        const uint8_t *d = &file[o + 8];
        if (tag == "IHDR" && len >= 13) {
            w = int(be32(o + 8)); h = int(be32(o + 12));
            depth = d[8]; colorType = d[9]; interlace = d[12];
        } else if (tag == "PLTE") palette.assign(d, d + len);
        else if (tag == "tRNS") trns.assign(d, d + len);
        else if (tag == "IDAT") idat.insert(idat.end(), d, d + len);
        else if (tag == "IEND") break;
        o += 12 + len;
    }
    if (w <= 0 || h <= 0 || depth != 8 || interlace != 0) return false; // 8-bit, non-interlaced files only
    const int channels = colorType == 0 ? 1 : colorType == 2 ? 3 : colorType == 3 ? 1 : colorType == 4 ? 2 : colorType == 6 ? 4 : 0;
    if (!channels) return false;
    std::vector<uint8_t> raw;
    if (!inflateZlib(idat, raw)) return false;
    const size_t stride = size_t(w)*channels;
    if (raw.size() < (stride + 1)*size_t(h)) return false;
    std::vector<uint8_t> img(stride*h), zero(stride, 0);
    for (int y = 0; y < h; ++y) { // PNG specification, section 9: filter types 0..4 on bytes, bpp = channels
        const uint8_t *src = &raw[(stride + 1)*size_t(y)];
        const int filter = src[0];
        uint8_t *cur = &img[stride*y];
        const uint8_t *up = y ? &img[stride*(y - 1)] : zero.data();
        for (size_t x = 0; x < stride; ++x) {
            const int a = x >= size_t(channels) ? cur[x - channels] : 0, b = up[x], c = x >= size_t(channels) ? up[x - channels] : 0;
            int v = src[1 + x];
            if (filter == 1) v += a;
            else if (filter == 2) v += b;
            else if (filter == 3) v += (a + b) >> 1;
            else if (filter == 4) {
                const int p0 = a + b - c, pa = std::abs(p0 - a), pb = std::abs(p0 - b), pc = std::abs(p0 - c);
                v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (filter != 0) return false;
            cur[x] = uint8_t(v);
        }
    }
    rgba.resize(size_t(w)*h*4);
    for (size_t i = 0; i < size_t(w)*h; ++i) {
        uint8_t r, g, b, a = 255;
        const uint8_t *p = &img[i*channels];
        if (colorType == 0) { r = g = b = p[0]; }
        else if (colorType == 2) { r = p[0]; g = p[1]; b = p[2]; }
        else if (colorType == 3) {
            if (size_t(p[0])*3 + 2 >= palette.size()) return false;
            r = palette[p[0]*3]; g = palette[p[0]*3 + 1]; b = palette[p[0]*3 + 2];
            if (p[0] < trns.size()) a = trns[p[0]];
        } else if (colorType == 4) { r = g = b = p[0]; a = p[1]; }
        else { r = p[0]; g = p[1]; b = p[2]; a = p[3]; }
        rgba[i*4] = r; rgba[i*4 + 1] = g; rgba[i*4 + 2] = b; rgba[i*4 + 3] = a;
    }
    return true;
}

// PNG with stored (uncompressed) deflate blocks: enough for an output writer.
bool savePng(const std::string &path, const uint8_t *rgb, int w, int h)
{
    static uint32_t crcTable[256];
    static bool init = false;
    if (!init) {
        for (uint32_t n = 0; n < 256; ++n) {
            uint32_t c = n;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            crcTable[n] = c;
        }
        init = true;
    }
    auto crc = [&](const std::vector<uint8_t> &buf, size_t from) {
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = from; i < buf.size(); ++i) c = crcTable[(c ^ buf[i]) & 0xFF] ^ (c >> 8);
        return c ^ 0xFFFFFFFFu;
    };
    auto put32 = [](std::vector<uint8_t> &b, uint32_t v) {
        b.push_back(uint8_t(v >> 24)); b.push_back(uint8_t(v >> 16)); b.push_back(uint8_t(v >> 8)); b.push_back(uint8_t(v));
    };
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    auto chunk = [&](const char *tag, const std::vector<uint8_t> &payload) {
        put32(out, uint32_t(payload.size()));
        size_t from = out.size();
        out.insert(out.end(), tag, tag + 4);
        out.insert(out.end(), payload.begin(), payload.end());
        put32(out, crc(out, from));
    };
    std::vector<uint8_t> ihdr;
    put32(ihdr, uint32_t(w)); put32(ihdr, uint32_t(h));
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr);

    std::vector<uint8_t> raw;
    raw.reserve(size_t(h)*(size_t(w)*3 + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + size_t(y)*w*3, rgb + size_t(y + 1)*w*3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    size_t pos = 0;
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521; b = (b + a) % 65521; }
    while (pos < raw.size() || raw.empty()) {
        size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n >= raw.size() ? 1 : 0);
        z.push_back(uint8_t(n & 0xFF)); z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n & 0xFF)); z.push_back(uint8_t((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
        if (raw.empty()) break;
    }
    put32(z, (b << 16) | a);
    chunk("IDAT", z);
    chunk("IEND", {});

    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return true;
}

void tonemapToLdr(const std::string &op, const float *rgb, int w, int h, std::vector<uint8_t> &ldr)
{
    ldr.resize(size_t(w)*h*3);
    for (size_t i = 0; i < size_t(w)*h*3; ++i) {
        float c = rgb[i] > 0.0f ? rgb[i] : 0.0f; // Camera::tonemap clamps below at 0 (Camera.hpp:158-161)
        float t;
        if (op == "linear") {
            t = c;
        } else if (op == "gamma") {
            t = std::pow(c, 1.0f/2.2f);
        } else if (op == "reinhard") {
            t = std::pow(c/(c + 1.0f), 1.0f/2.2f);
        } else if (op == "filmic") {
            float x = std::max(0.0f, c - 0.004f);
            t = (x*(6.2f*x + 0.5f))/(x*(6.2f*x + 1.7f) + 0.06f);
        } else { // pbrt
            t = c < 0.0031308f ? 12.92f*c : 1.055f*std::pow(c, 1.0f/2.4f) - 0.055f;
        }
        int v = int(t*255.0f);
        ldr[i] = uint8_t(std::min(std::max(v, 0), 255));
    }
}

// ---------------------------------------------------------------------------
// JSON -> scene objects
// ---------------------------------------------------------------------------
// One bitmap texture, expanded to float RGB texels exactly as BitmapTexture::getRgb/getScalar would return them
// (textures/BitmapTexture.cpp:140-156), so the device lookup needs no texel types:
//  * HDR files (.hdr, .pfm) stay float; a scalar request averages (r + g + b)/3 in float (io/ImageIO.cpp:173-187,298-365);
//  * LDR files (.png) are RGBA8: an RGB request runs R, G, B through the integer gamma table when gamma_correct is set
//    (ImageIO.cpp:26-43,507-519: entry i = int(255*(i/255)^2.2)), a scalar request takes the integer (r + g + b)/3 of the
//    uncorrected bytes (:520-523); either way a texel is float(byte)*(1/255) (BitmapTexture.cpp:15-23,140-146).
// A scalar texture is stored as (s, s, s): operator[] returns Vec3f(scalar) (BitmapTexture.cpp:323-351).
int Scene::fetchBitmap(const std::string &path, bool scalarAverage, bool gammaCorrect, bool linear, bool clamp, float scale)
{
    std::string full = path.size() && path[0] == '/' ? path : sceneDir + "/" + path;
    char keyTail[96];
    std::snprintf(keyTail, sizeof(keyTail), "|%d%d%d%d|%a", int(scalarAverage), int(gammaCorrect), int(linear), int(clamp), double(scale));
    const std::string key = full + keyTail;
    for (size_t i = 0; i < _bitmapPaths.size(); ++i)
        if (_bitmapPaths[i] == key) return int(i);

    int w = 0, h = 0;
    std::vector<float> texels;
    const bool isHdr = endsWith(full, ".hdr") || endsWith(full, ".pfm");
    if (isHdr) {
        const bool ok = endsWith(full, ".hdr") ? loadRadianceHdr(full, w, h, texels) : loadPfm(full, w, h, texels);
        if (!ok) throw std::runtime_error("Unable to load texture at '" + full + "'");
        if (scalarAverage)
            for (size_t i = 0; i < size_t(w)*h; ++i) {
                const float s = (texels[i*3] + texels[i*3 + 1] + texels[i*3 + 2])/3;
                texels[i*3] = texels[i*3 + 1] = texels[i*3 + 2] = s;
            }
    } else if (endsWith(full, ".png") || endsWith(full, ".bmp") || endsWith(full, ".tga") || endsWith(full, ".jpg") || endsWith(full, ".jpeg")) {
        std::vector<uint8_t> rgba;
        std::string why;
        if (endsWith(full, ".png")) {
            if (!loadPng(full, w, h, rgba))
                throw std::runtime_error("Unable to load texture at '" + full + "' (8-bit non-interlaced PNG expected)");
        } else if (!(endsWith(full, ".bmp") ? loadBmp(full, w, h, rgba, why) : endsWith(full, ".tga") ? loadTga(full, w, h, rgba, why) : loadJpeg(full, w, h, rgba, why))) {
            throw std::runtime_error("Unable to load texture at '" + full + "': the file " + why);
        }
        uint8_t gamma[256];
        for (int i = 0; i < 256; ++i) gamma[i] = gammaCorrect ? uint8_t(int(255.0*std::pow(i/255.0, 2.2))) : uint8_t(i);
        texels.resize(size_t(w)*h*3);
        for (size_t i = 0; i < size_t(w)*h; ++i) {
            if (scalarAverage) {
                const int s = (int(rgba[i*4]) + int(rgba[i*4 + 1]) + int(rgba[i*4 + 2]))/3;
                texels[i*3] = texels[i*3 + 1] = texels[i*3 + 2] = float(uint8_t(s))*(1.0f/255.0f);
            } else {
                for (int k = 0; k < 3; ++k) texels[i*3 + k] = float(gamma[rgba[i*4 + k]])*(1.0f/255.0f);
            }
        }
    } else {
        throw std::runtime_error("Unable to load texture at '" + full + "' (.hdr, .pfm, .png, .bmp, .tga and baseline .jpg bitmaps are in scope)");
    }
    tgpu_bitmap bm{};
    bm.w = w; bm.h = h; bm.clamp = clamp ? 1 : 0; bm.linear = linear ? 1 : 0; bm.scale = scale; bm.texels = nullptr;
    { // BitmapTexture::init (:183-207): _avg += getRgb(x, y)/float(_w*_h), rows top to bottom (average() = _scale*_avg is applied by the consumers)
        float avg[3] = {0.0f, 0.0f, 0.0f};
        const float n = float(w*h);
        for (size_t i = 0; i < size_t(w)*h; ++i)
            for (int k = 0; k < 3; ++k) avg[k] += texels[i*3 + k]/n;
        for (int k = 0; k < 3; ++k) bm.average[k] = avg[k];
    }
    _bitmaps.push_back(bm);
    _bitmapTexels.push_back(std::move(texels));
    _bitmapPaths.push_back(key);
    return int(_bitmaps.size()) - 1;
}

int Scene::fetchTexture(const Json &v, bool scalarAverage) // Scene.cpp:127-151
{
    tgpu_texture t{};
    t.bitmap = -1;
    if (v.isString()) {
        // TextureCache.hpp:32-33: gammaCorrect = true, linear = true, clamp = false
        t.type = TGPU_TEX_BITMAP;
        t.bitmap = fetchBitmap(v.asString(), scalarAverage, true, true, false, 1.0f);
    } else if (v.isNumber()) {
        t.type = TGPU_TEX_CONSTANT;
        t.value[0] = t.value[1] = t.value[2] = v.asFloat();
    } else if (v.isArray()) {
        t.type = TGPU_TEX_CONSTANT;
        v.toVec(t.value, 3);
    } else if (v.isObject()) {
        std::string type = v["type"].asString();
        if (type == "checker") { // CheckerTexture.cpp:15-45
            t.type = TGPU_TEX_CHECKER;
            t.on_color[0] = t.on_color[1] = t.on_color[2] = 0.8f;
            t.off_color[0] = t.off_color[1] = t.off_color[2] = 0.2f;
            t.res_u = t.res_v = 20;
            v.getVec("on_color", t.on_color, 3);
            v.getVec("off_color", t.off_color, 3);
            v.getField("res_u", t.res_u);
            v.getField("res_v", t.res_v);
        } else if (type == "constant") {
            t.type = TGPU_TEX_CONSTANT;
            v.getVec("value", t.value, 3);
        } else if (type == "bitmap") { // BitmapTexture::fromJson (textures/BitmapTexture.cpp:211-219)
            bool gammaCorrect = true, linear = true, clamp = false;
            float scale = 1.0f;
            v.getField("gamma_correct", gammaCorrect);
            v.getField("interpolate", linear);
            v.getField("clamp", clamp);
            v.getField("scale", scale);
            if (!v["file"]) throw std::runtime_error("bitmap texture without a \"file\"");
            t.type = TGPU_TEX_BITMAP;
            t.bitmap = fetchBitmap(v["file"].asString(), scalarAverage, gammaCorrect, linear, clamp, scale);
        } else {
            throw std::runtime_error("Texture type '" + type + "' is outside the hot-path scope");
        }
    } else {
        throw JsonLoadException("Type mismatch: Expecting a texture here");
    }
    _textures.push_back(t);
    return int(_textures.size()) - 1;
}

// Cu from ComplexIorData.hpp:26 is the only conductor the benchmark configs use;
// a few more common ones are listed for convenience.
static bool lookupComplexIor(const std::string &name, float *eta, float *k)
{
    struct Entry { const char *name; float eta[3]; float k[3]; };
    static const Entry table[] = {
        // measured complex IORs at R/G/B, the values of bsdfs/ComplexIorData.hpp:18-26 (data)
        {"Ag", {0.1552646489f, 0.1167232965f, 0.1383806959f}, {4.8283433224f, 3.1222459278f, 2.1469504455f}},
        {"Al", {1.6574599595f, 0.8803689579f, 0.5212287346f}, {9.2238691996f, 6.2695232477f, 4.8370012281f}},
        {"Au", {0.1431189557f, 0.3749570432f, 1.4424785571f}, {3.9831604247f, 2.3857207478f, 1.6032152899f}},
        {"Cu", {0.2004376970f, 0.9240334304f, 1.1022119527f}, {3.9129485033f, 2.4528477015f, 2.1421879552f}},
    };
    for (const Entry &e : table) {
        if (name == e.name) {
            for (int i = 0; i < 3; ++i) { eta[i] = e.eta[i]; k[i] = e.k[i]; }
            return true;
        }
    }
    return false;
}

int Scene::parseBsdf(const Json &v)
{
    tgpu_bsdf b{};
    std::string type = v["type"].asString();
    // Bsdf::fromJson (Bsdf.cpp): albedo default = constant 1
    if (v["albedo"])
        b.albedo_tex = fetchTexture(v["albedo"], false);
    else {
        Json one; one.type = Json::Number; one.num = 1.0;
        b.albedo_tex = fetchTexture(one, false);
    }
    if (v["bump"]) throw std::runtime_error("bump maps are outside the hot-path scope");
    b.roughness_tex = -1;
    b.substrate = -1;
    b.aux_bsdf = -1;
    b.aux_tex = -1;
    if (type == "lambert") {
        b.type = TGPU_BSDF_LAMBERT;
    } else if (type == "null") {
        b.type = TGPU_BSDF_NULL;
    } else if (type == "forward") {
        b.type = TGPU_BSDF_FORWARD;
    } else if (type == "plastic" || type == "rough_plastic") { // PlasticBsdf.cpp:19-33,189-195, RoughPlasticBsdf.cpp:19-41,252-258
        b.type = type == "plastic" ? TGPU_BSDF_PLASTIC : TGPU_BSDF_ROUGH_PLASTIC;
        b.ior = 1.5f;
        b.thickness = 1.0f;
        b.sigma_a[0] = b.sigma_a[1] = b.sigma_a[2] = 0.0f;
        v.getField("ior", b.ior);
        v.getField("thickness", b.thickness);
        v.getVec("sigma_a", b.sigma_a, 3);
        for (int i = 0; i < 3; ++i) b.scaled_sigma_a[i] = b.thickness*b.sigma_a[i];
        {
            float sum = b.scaled_sigma_a[0]; sum += b.scaled_sigma_a[1]; sum += b.scaled_sigma_a[2];
            b.avg_transmittance = std::exp(-2.0f*(sum*(1.0f/3.0f)));
        }
        { // Fresnel::computeDiffuseFresnel(ior, 1000000) (Fresnel.hpp:141-153) with dielectricReflectance (:75-92)
            auto fresnel = [](float eta, float cosThetaI) {
                if (cosThetaI < 0.0f) { eta = 1.0f/eta; cosThetaI = -cosThetaI; }
                float sinThetaTSq = eta*eta*(1.0f - cosThetaI*cosThetaI);
                if (sinThetaTSq > 1.0f) return 1.0f;
                float cosThetaT = std::sqrt(std::max(1.0f - sinThetaTSq, 0.0f));
                float Rs = (eta*cosThetaI - cosThetaT)/(eta*cosThetaI + cosThetaT);
                float Rp = (eta*cosThetaT - cosThetaI)/(eta*cosThetaT + cosThetaI);
                return (Rs*Rs + Rp*Rp)*0.5f;
            };
            const int sampleCount = 1000000;
            double diffuseFresnel = 0.0;
            float fb = fresnel(b.ior, 0.0f);
            for (int i = 1; i <= sampleCount; ++i) {
                float cosThetaSq = float(i)/sampleCount;
                float fa = fresnel(b.ior, std::min(std::sqrt(cosThetaSq), 1.0f));
                diffuseFresnel += double(fa + fb)*(0.5/sampleCount);
                fb = fa;
            }
            b.diffuse_fresnel = float(diffuseFresnel);
        }
        if (type == "rough_plastic") {
            std::string dist = "ggx";
            v.getField("distribution", dist);
            if (dist == "beckmann") b.distribution = TGPU_DIST_BECKMANN;
            else if (dist == "phong") b.distribution = TGPU_DIST_PHONG;
            else if (dist == "ggx") b.distribution = TGPU_DIST_GGX;
            else throw JsonLoadException("Unknown microfacet distribution '" + dist + "'");
            if (v["roughness"])
                b.roughness_tex = fetchTexture(v["roughness"], true);
            else {
                Json r; r.type = Json::Number; r.num = 0.02; // RoughPlasticBsdf.cpp:26
                b.roughness_tex = fetchTexture(r, true);
            }
            const tgpu_texture &at = _textures[size_t(b.albedo_tex)];
            float avg[3]; // Texture::average(): ConstantTexture = _value, BitmapTexture = _scale*_avg (BitmapTexture.cpp:281-284)
            if (at.type == TGPU_TEX_CONSTANT) for (int k = 0; k < 3; ++k) avg[k] = at.value[k];
            else if (at.type == TGPU_TEX_BITMAP) for (int k = 0; k < 3; ++k) avg[k] = _bitmaps[size_t(at.bitmap)].scale*_bitmaps[size_t(at.bitmap)].average[k];
            else throw std::runtime_error("rough_plastic with a procedural albedo is outside the hot-path scope");
            float sum = avg[0]; sum += avg[1]; sum += avg[2]; // Vec::avg()
            b.substrate_weight = sum*(1.0f/3.0f);
        }
    } else if (type == "oren_nayar") { // OrenNayarBsdf.cpp:14-31 (roughness default 1)
        b.type = TGPU_BSDF_OREN_NAYAR;
        if (v["roughness"])
            b.roughness_tex = fetchTexture(v["roughness"], true);
        else {
            Json r; r.type = Json::Number; r.num = 1.0;
            b.roughness_tex = fetchTexture(r, true);
        }
    } else if (type == "phong") { // PhongBsdf.cpp:16-30
        b.type = TGPU_BSDF_PHONG;
        b.exponent = 64.0f;
        b.diffuse_ratio = 0.2f;
        v.getField("exponent", b.exponent);
        v.getField("diffuse_ratio", b.diffuse_ratio);
    } else if (type == "mixed") { // MixedBsdf.cpp:47-58
        b.type = TGPU_BSDF_MIXED;
        if (!v["bsdf0"] || !v["bsdf1"]) throw JsonLoadException("mixed: bsdf0 and bsdf1 are required");
        b.substrate = fetchBsdf(v["bsdf0"]);
        b.aux_bsdf = fetchBsdf(v["bsdf1"]);
        if (v["ratio"])
            b.aux_tex = fetchTexture(v["ratio"], true);
        else {
            Json r; r.type = Json::Number; r.num = 0.5;
            b.aux_tex = fetchTexture(r, true);
        }
    } else if (type == "rough_coat") { // RoughCoatBsdf.cpp:14-33,300-305
        b.type = TGPU_BSDF_ROUGH_COAT;
        b.ior = 1.3f;
        b.thickness = 1.0f;
        b.sigma_a[0] = b.sigma_a[1] = b.sigma_a[2] = 0.0f;
        v.getField("ior", b.ior);
        v.getField("thickness", b.thickness);
        v.getVec("sigma_a", b.sigma_a, 3);
        std::string dist = "ggx";
        v.getField("distribution", dist);
        if (dist == "beckmann") b.distribution = TGPU_DIST_BECKMANN;
        else if (dist == "phong") b.distribution = TGPU_DIST_PHONG;
        else if (dist == "ggx") b.distribution = TGPU_DIST_GGX;
        else throw JsonLoadException("Unknown microfacet distribution '" + dist + "'");
        if (v["roughness"])
            b.roughness_tex = fetchTexture(v["roughness"], true);
        else {
            Json r; r.type = Json::Number; r.num = 0.02;
            b.roughness_tex = fetchTexture(r, true);
        }
        if (!v["substrate"]) throw std::runtime_error("rough_coat without substrate (the default rough_conductor) is outside the hot-path scope");
        b.substrate = fetchBsdf(v["substrate"]);
        for (int i = 0; i < 3; ++i) b.scaled_sigma_a[i] = b.thickness*b.sigma_a[i];
        {
            float sum = b.scaled_sigma_a[0]; sum += b.scaled_sigma_a[1]; sum += b.scaled_sigma_a[2];
            b.avg_transmittance = std::exp(-2.0f*(sum*(1.0f/3.0f)));
        }
    } else if (type == "transparency") { // TransparencyBsdf.cpp:12-32
        b.type = TGPU_BSDF_TRANSPARENCY;
        if (!v["base"]) throw std::runtime_error("transparency without base is outside the hot-path scope");
        b.substrate = fetchBsdf(v["base"]);
        if (v["alpha"])
            b.aux_tex = fetchTexture(v["alpha"], true);
        else {
            Json r; r.type = Json::Number; r.num = 1.0;
            b.aux_tex = fetchTexture(r, true);
        }
    } else if (type == "thinsheet") { // ThinSheetBsdf.cpp:16-38
        b.type = TGPU_BSDF_THINSHEET;
        b.ior = 1.5f;
        b.sigma_a[0] = b.sigma_a[1] = b.sigma_a[2] = 0.0f;
        v.getField("ior", b.ior);
        v.getVec("sigma_a", b.sigma_a, 3);
        bool interference = false;
        v.getField("enable_interference", interference);
        if (interference) throw std::runtime_error("thinsheet with enable_interference is outside the hot-path scope");
        if (v["thickness"])
            b.aux_tex = fetchTexture(v["thickness"], true);
        else {
            Json r; r.type = Json::Number; r.num = 0.5;
            b.aux_tex = fetchTexture(r, true);
        }
    } else if (type == "mirror") {
        b.type = TGPU_BSDF_MIRROR;
    } else if (type == "dielectric") { // DielectricBsdf.cpp:34-40
        b.type = TGPU_BSDF_DIELECTRIC;
        b.ior = 1.5f;
        v.getField("ior", b.ior);
        bool enableT = true;
        v.getField("enable_refraction", enableT);
        b.enable_refraction = enableT ? 1 : 0;
    } else if (type == "conductor") { // ConductorBsdf.cpp:14-38 (default material Cu)
        b.type = TGPU_BSDF_CONDUCTOR;
        std::string material;
        const float defEta[3] = {0.200438f, 0.924033f, 1.10221f}, defK[3] = {3.91295f, 2.45285f, 2.14219f};
        for (int i = 0; i < 3; ++i) { b.eta[i] = defEta[i]; b.k[i] = defK[i]; }
        v.getVec("eta", b.eta, 3);
        v.getVec("k", b.k, 3);
        if (v.getField("material", material) && !lookupComplexIor(material, b.eta, b.k))
            throw JsonLoadException("Unable to find material with name '" + material + "'");
    } else if (type == "rough_conductor") { // RoughConductorBsdf.cpp:14-45
        b.type = TGPU_BSDF_ROUGH_CONDUCTOR;
        std::string material;
        std::string dist = "ggx";
        const float defEta[3] = {0.200438f, 0.924033f, 1.10221f}, defK[3] = {3.91295f, 2.45285f, 2.14219f};
        for (int i = 0; i < 3; ++i) { b.eta[i] = defEta[i]; b.k[i] = defK[i]; }
        v.getVec("eta", b.eta, 3);
        v.getVec("k", b.k, 3);
        v.getField("distribution", dist);
        if (v.getField("material", material) && !lookupComplexIor(material, b.eta, b.k))
            throw JsonLoadException("Unable to find material with name '" + material + "'");
        if (dist == "beckmann") b.distribution = TGPU_DIST_BECKMANN;
        else if (dist == "phong") b.distribution = TGPU_DIST_PHONG;
        else if (dist == "ggx") b.distribution = TGPU_DIST_GGX;
        else throw JsonLoadException("Unknown microfacet distribution '" + dist + "'");
        if (v["roughness"])
            b.roughness_tex = fetchTexture(v["roughness"], true);
        else {
            Json r; r.type = Json::Number; r.num = 0.1;
            b.roughness_tex = fetchTexture(r, true);
        }
    } else if (type == "smooth_coat") { // SmoothCoatBsdf.cpp:19-39
        b.type = TGPU_BSDF_SMOOTH_COAT;
        b.ior = 1.3f;
        b.thickness = 1.0f;
        b.sigma_a[0] = b.sigma_a[1] = b.sigma_a[2] = 0.0f;
        v.getField("ior", b.ior);
        v.getField("thickness", b.thickness);
        v.getVec("sigma_a", b.sigma_a, 3);
        if (!v["substrate"]) throw std::runtime_error("smooth_coat without substrate is outside the hot-path scope");
        b.substrate = fetchBsdf(v["substrate"]);
        if (_bsdfs[size_t(b.substrate)].type == TGPU_BSDF_SMOOTH_COAT || _bsdfs[size_t(b.substrate)].type == TGPU_BSDF_NULL)
            throw std::runtime_error("smooth_coat substrate must be lambert or rough_conductor in this scope");
        for (int i = 0; i < 3; ++i) b.scaled_sigma_a[i] = b.thickness*b.sigma_a[i]; // SmoothCoatBsdf::prepareForRender
        {
            float sum = b.scaled_sigma_a[0]; sum += b.scaled_sigma_a[1]; sum += b.scaled_sigma_a[2];
            float avg = sum*(1.0f/3.0f);
            b.avg_transmittance = std::exp(-2.0f*avg);
        }
    } else if (type == "rough_dielectric") { // RoughDielectricBsdf.cpp:19-42
        b.type = TGPU_BSDF_ROUGH_DIELECTRIC;
        b.ior = 1.5f;
        v.getField("ior", b.ior);
        std::string dist = "ggx";
        v.getField("distribution", dist);
        if (dist == "beckmann") b.distribution = TGPU_DIST_BECKMANN;
        else if (dist == "phong") b.distribution = TGPU_DIST_PHONG;
        else if (dist == "ggx") b.distribution = TGPU_DIST_GGX;
        else throw JsonLoadException("Unknown microfacet distribution '" + dist + "'");
        bool enableT = true;
        v.getField("enable_refraction", enableT);
        b.enable_refraction = enableT ? 1 : 0;
        if (v["roughness"])
            b.roughness_tex = fetchTexture(v["roughness"], true);
        else {
            Json r; r.type = Json::Number; r.num = 0.1;
            b.roughness_tex = fetchTexture(r, true);
        }
    } else {
        throw std::runtime_error("BSDF type '" + type + "' is outside the hot-path scope (SURVEY.md section 2 row 8)");
    }
    _bsdfs.push_back(b);
    return int(_bsdfs.size()) - 1;
}

int Scene::parseMedium(const Json &v) // HomogeneousMedium.cpp:19-48, Medium.cpp:19-29
{
    std::string type = v["type"].asString();
    if (type != "homogeneous") throw std::runtime_error("Medium type '" + type + "' is outside the hot-path scope (SURVEY.md section 8 a26: homogeneous only)");
    int phaseType = 0;
    float phaseG = 0.0f;
    if (const Json &ph = v["phase_function"]) { // PhaseFunctionFactory.cpp
        std::string pt = ph.isObject() ? ph["type"].asString() : ph.asString();
        if (pt == "henyey_greenstein") {
            phaseType = 1;
            if (ph.isObject()) ph.getField("g", phaseG);
        } else if (pt != "isotropic") {
            throw std::runtime_error("phase function '" + pt + "' is outside the hot-path scope");
        }
    }
    // TransmittanceFactory.cpp:14-24; one model: type + up to three parameters
    struct Model { int type = TGPU_TRANS_EXPONENTIAL; float param[3] = {0, 0, 0}; };
    auto parseModel = [](const Json &tr, bool allowInterpolated, auto &self, Model *childA, Model *childB, float *ratio) -> Model {
        Model m;
        std::string tt = tr.isObject() ? tr["type"].asString() : tr.asString();
        if (tt == "exponential") {
            m.type = TGPU_TRANS_EXPONENTIAL;
        } else if (tt == "linear" || tt == "quadratic") { // LinearTransmittance.cpp:10-20, QuadraticTransmittance.cpp:10-20
            m.type = tt == "linear" ? TGPU_TRANS_LINEAR : TGPU_TRANS_QUADRATIC;
            m.param[0] = 1.0f;
            if (tr.isObject()) tr.getField("max_t", m.param[0]);
        } else if (tt == "double_exponential") { // DoubleExponentialTransmittance.cpp:13-24
            m.type = TGPU_TRANS_DOUBLE_EXPONENTIAL;
            m.param[0] = 0.5f;
            m.param[1] = 10.0f;
            if (tr.isObject()) { tr.getField("sigma_a", m.param[0]); tr.getField("sigma_b", m.param[1]); }
        } else if (tt == "pulse") { // PulseTransmittance.cpp:13-27
            m.type = TGPU_TRANS_PULSE;
            m.param[0] = 0.0f;
            m.param[1] = 1.0f;
            int numPulses = 4;
            if (tr.isObject()) { tr.getField("min", m.param[0]); tr.getField("max", m.param[1]); tr.getField("num_pulses", numPulses); }
            m.param[2] = float(numPulses);
        } else if (tt == "erlang") { // ErlangTransmittance.cpp:13-22
            m.type = TGPU_TRANS_ERLANG;
            m.param[0] = 5.0f;
            if (tr.isObject()) tr.getField("rate", m.param[0]);
        } else if (tt == "davis") { // DavisTransmittance.cpp:10-21: alpha has to be > 1, clamped with a warning
            m.type = TGPU_TRANS_DAVIS;
            m.param[0] = 1.1f;
            if (tr.isObject()) tr.getField("alpha", m.param[0]);
            if (m.param[0] < 1 + 1e-6f) {
                std::printf("Warning: alpha parameter of Davis transmittance has to be > 1. Clamping the current value (%g)\n", double(m.param[0]));
                m.param[0] = 1 + 1e-6f;
            }
        } else if (tt == "davis_weinstein") { // DavisWeinsteinTransmittance.cpp:8-24
            m.type = TGPU_TRANS_DAVIS_WEINSTEIN;
            m.param[0] = 0.75f;
            m.param[1] = 1.0f;
            if (tr.isObject()) { tr.getField("h", m.param[0]); tr.getField("c", m.param[1]); }
            if (m.param[0] < 0.5f || m.param[0] > 1.0f)
                std::printf("Warning: Valid range of Davis Weinstein transmittance is [0.5, 1.0]. Clamping current value (%g) to within range\n", double(m.param[0]));
            m.param[0] = std::min(std::max(m.param[0], 0.5f), 1.0f);
        } else if (tt == "interpolated" && allowInterpolated) { // InterpolatedTransmittance.cpp:17-30: linear and erlang by default
            m.type = TGPU_TRANS_INTERPOLATED;
            childA->type = TGPU_TRANS_LINEAR; childA->param[0] = 1.0f;
            childB->type = TGPU_TRANS_ERLANG; childB->param[0] = 5.0f;
            *ratio = 0.5f;
            if (tr.isObject()) {
                if (const Json &a = tr["tr_a"]) *childA = self(a, false, self, nullptr, nullptr, nullptr);
                if (const Json &b = tr["tr_b"]) *childB = self(b, false, self, nullptr, nullptr, nullptr);
                tr.getField("ratio", *ratio);
            }
        } else if (tt == "interpolated") {
            throw std::runtime_error("an interpolated transmittance inside an interpolated transmittance is outside the hot-path scope");
        } else {
            throw std::runtime_error("transmittance '" + tt + "' is outside the hot-path scope");
        }
        return m;
    };
    Model model, childA, childB;
    float trRatio = 0.0f;
    if (const Json &tr = v["transmittance"]) model = parseModel(tr, true, parseModel, &childA, &childB, &trRatio);
    float sa[3] = {0, 0, 0}, ss[3] = {0, 0, 0}, density = 1.0f;
    v.getVec("sigma_a", sa, 3);
    v.getVec("sigma_s", ss, 3);
    v.getField("density", density);
    tgpu_medium m{};
    m.max_bounce = 1024;
    v.getField("max_bounces", m.max_bounce);
    for (int i = 0; i < 3; ++i) {
        m.sigma_a[i] = sa[i]*density;
        m.sigma_s[i] = ss[i]*density;
        m.sigma_t[i] = m.sigma_a[i] + m.sigma_s[i];
    }
    m.absorption_only = m.sigma_s[0] == 0.0f && m.sigma_s[1] == 0.0f && m.sigma_s[2] == 0.0f;
    m.phase = phaseType;
    m.phase_g = phaseG;
    m.transmittance = model.type;
    m.tr_a_type = childA.type;
    m.tr_b_type = childB.type;
    m.tr_ratio = trRatio;
    for (int i = 0; i < 3; ++i) { m.param[i] = model.param[i]; m.tr_a_param[i] = childA.param[i]; m.tr_b_param[i] = childB.param[i]; }
    _media.push_back(m);
    return int(_media.size()) - 1;
}

int Scene::fetchMedium(const Json &v) // Scene.cpp:105-108
{
    if (v.isString()) {
        for (auto &kv : _namedMedia)
            if (kv.first == v.asString()) return kv.second;
        throw JsonLoadException("Unable to find an object with name '" + v.asString() + "'");
    } else if (v.isObject()) {
        return parseMedium(v);
    }
    throw JsonLoadException("Type mismatch: Expecting either an object or an object reference here");
}

int Scene::fetchBsdf(const Json &v) // Scene.cpp:83-93,120-125
{
    if (v.isString()) {
        for (auto &kv : _namedBsdfs)
            if (kv.first == v.asString()) return kv.second;
        throw JsonLoadException("Unable to find an object with name '" + v.asString() + "'");
    } else if (v.isObject()) {
        return parseBsdf(v);
    }
    throw JsonLoadException("Type mismatch: Expecting either an object or an object reference here");
}

MeshData Scene::parseMesh(const Json &v, const Mat4 &transform, int emissionTex)
{
    if (emissionTex >= 0) throw std::runtime_error("emissive meshes are outside the hot-path scope");
    MeshData m;
    v.getField("name", m.name);
    m.transform = transform;
    v.getField("smooth", m.smooth);
    bool recompute = false;
    v.getField("recompute_normals", recompute);
    if (recompute) throw std::runtime_error("recompute_normals is outside the hot-path scope");
    const Json &bsdf = v["bsdf"];
    if (bsdf.isArray())
        for (size_t i = 0; i < bsdf.size(); ++i) m.bsdfs.push_back(fetchBsdf(bsdf[i]));
    else if (bsdf)
        m.bsdfs.push_back(fetchBsdf(bsdf));
    else {
        Json d = Json::parse("{\"type\":\"lambert\",\"albedo\":1}"); // Primitive::_defaultBsdf
        m.bsdfs.push_back(parseBsdf(d));
    }
    std::string file;
    if (v.getField("file", file)) {
        std::string full = file.size() && file[0] == '/' ? file : sceneDir + "/" + file;
        if (!endsWith(full, ".wo3") || !loadWo3(full, m.verts, m.tris))
            throw std::runtime_error("Unable to load triangle mesh at " + full);
    }
    return m;
}

// ---- QuaternionF (math/Quaternion.hpp), stored w x y z ----
namespace {
struct Quat {
    float v[4];
    static Quat fromMatrix(const Mat4 &a) // Quaternion.hpp:113-151
    {
        float trace = a(0, 0) + a(1, 1) + a(2, 2);
        if (trace > 0.0f) {
            float s = 0.5f/std::sqrt(trace + 1.0f);
            return Quat{{0.25f/s, (a(2, 1) - a(1, 2))*s, (a(0, 2) - a(2, 0))*s, (a(1, 0) - a(0, 1))*s}};
        } else if (a(0, 0) > a(1, 1) && a(0, 0) > a(2, 2)) {
            float s = 2.0f*std::sqrt(1.0f + a(0, 0) - a(1, 1) - a(2, 2));
            return Quat{{(a(2, 1) - a(1, 2))/s, 0.25f*s, (a(0, 1) + a(1, 0))/s, (a(0, 2) + a(2, 0))/s}};
        } else if (a(1, 1) > a(2, 2)) {
            float s = 2.0f*std::sqrt(1.0f + a(1, 1) - a(0, 0) - a(2, 2));
            return Quat{{(a(0, 2) - a(2, 0))/s, (a(0, 1) + a(1, 0))/s, 0.25f*s, (a(1, 2) + a(2, 1))/s}};
        } else {
            float s = 2.0f*std::sqrt(1.0f + a(2, 2) - a(0, 0) - a(1, 1));
            return Quat{{(a(1, 0) - a(0, 1))/s, (a(0, 2) + a(2, 0))/s, (a(1, 2) + a(2, 1))/s, 0.25f*s}};
        }
    }
    static Quat fromAxisAngle(float theta, const Vec3 &u) // Quaternion.hpp:32-40
    {
        float cosTheta = std::cos(theta/2.0f), sinTheta = std::sin(theta/2.0f);
        return Quat{{cosTheta, u.x*sinTheta, u.y*sinTheta, u.z*sinTheta}};
    }
    Quat operator*(const Quat &o) const // :66-73
    {
        return Quat{{v[0]*o.v[0] - v[1]*o.v[1] - v[2]*o.v[2] - v[3]*o.v[3],
                     v[0]*o.v[1] + v[1]*o.v[0] + v[2]*o.v[3] - v[3]*o.v[2],
                     v[0]*o.v[2] - v[1]*o.v[3] + v[2]*o.v[0] + v[3]*o.v[1],
                     v[0]*o.v[3] + v[1]*o.v[2] - v[2]*o.v[1] + v[3]*o.v[0]}};
    }
};

// Instance files (Instance.cpp:129-232): u32 count, u32 compression flags, Box3f position bounds, per-instance
// records (lossless: pos3 + rotation vector3; lossy: 3 x u32), then count master ids (u8)
bool loadInstanceFile(const std::string &path, std::vector<float> &pos, std::vector<Quat> &rot, std::vector<uint8_t> &id)
{
    std::string data = readFile(path);
    if (data.size() < 32) return false;
    const char *p = data.data(), *end = p + data.size();
    auto rd = [&](void *dst, size_t n) { if (p + n > end) return false; std::memcpy(dst, p, n); p += n; return true; };
    uint32_t count, compressed;
    float bounds[6];
    if (!rd(&count, 4) || !rd(&compressed, 4) || !rd(bounds, 24)) return false;
    if (size_t(count)*((compressed & 1u) ? 12u : 24u) + count > size_t(end - p)) return false; // more instances than the file can hold
    pos.resize(size_t(count)*3);
    rot.resize(count);
    id.resize(count);
    const float TWO_PI_ = 6.283185307f; // Angle.hpp
    for (uint32_t i = 0; i < count; ++i) {
        if (compressed & 1u) { // loadLossyInstance :131-160
            uint32_t a, b, c;
            if (!rd(&a, 4) || !rd(&b, 4) || !rd(&c, 4)) return false;
            uint32_t mask = (1u << 21) - 1u;
            uint32_t xyz[3] = {a >> 11, ((a << 10) | (b >> 22)) & mask, (b >> 1) & mask};
            uint32_t r = c & 255u, axisX = (c >> 8) & 4095u, axisY = (c >> 20) & 4095u;
            float axisXf = (axisX/float(1 << 12))*2.0f - 1.0f;
            float axisYf = (axisY/float(1 << 12))*2.0f - 1.0f;
            float rotW = TWO_PI_*r/(1 << 8);
            float zz = 1 - axisXf*axisXf - axisYf*axisYf;
            Vec3 w(axisXf, axisYf, std::sqrt(zz > 0.0f ? zz : 0.0f));
            for (int k = 0; k < 3; ++k) {
                float t = float(xyz[k])/float(1 << 21);
                pos[size_t(i)*3 + k] = bounds[k]*(1.0f - t) + bounds[3 + k]*t;
            }
            rot[i] = Quat::fromAxisAngle(rotW, w);
        } else { // loadLosslessInstance :161-170
            float rec[6];
            if (!rd(rec, 24)) return false;
            for (int k = 0; k < 3; ++k) pos[size_t(i)*3 + k] = rec[k];
            Vec3 w(rec[3], rec[4], rec[5]);
            float angle = w.length();
            w = angle > 0 ? w/angle : Vec3(0.0f, 1.0f, 0.0f);
            rot[i] = Quat::fromAxisAngle(angle, w);
        }
    }
    return rd(id.data(), count);
}
} // namespace

void Scene::parsePrimitive(const Json &v)
{
    std::string type = v["type"].asString();
    Mat4 transform;
    if (v["transform"]) transform = parseTransform(v["transform"]);
    if (v["power"] && type != "quad") throw std::runtime_error("'power' emitters other than quads are outside the hot-path scope");
    int intMedium = -1, extMedium = -1;
    if (v["int_medium"]) intMedium = fetchMedium(v["int_medium"]);
    if (v["ext_medium"]) extMedium = fetchMedium(v["ext_medium"]);
    int emissionTex = -1;
    if (v["emission"]) emissionTex = fetchTexture(v["emission"], false);
    if (v["power"]) { // Primitive::prepareForRender (Primitive.cpp:111-117): _emission = power*powerToRadianceFactor()
        int powerTex = fetchTexture(v["power"], false);
        tgpu_texture t = _textures[size_t(powerTex)];
        if (t.type != TGPU_TEX_CONSTANT) throw std::runtime_error("textured 'power' is outside the hot-path scope");
        Vec3 edge0 = transform.transformVector(Vec3(1.0f, 0.0f, 0.0f)), edge1 = transform.transformVector(Vec3(0.0f, 0.0f, 1.0f));
        float area = edge1.cross(edge0).length();           // Quad.cpp:298-316
        float factor = (1.0f/3.1415926536f)*(1.0f/area);    // Quad::powerToRadianceFactor (:50-53): INV_PI*_invArea
        for (int i = 0; i < 3; ++i) t.value[i] *= factor;   // ConstantTexture::scaleValues
        _textures.push_back(t);
        emissionTex = int(_textures.size()) - 1;
    }

    if (type == "mesh") {
        meshes.push_back(parseMesh(v, transform, emissionTex));
        meshes.back().intMedium = intMedium;
        meshes.back().extMedium = extMedium;
        _primOrder.push_back({PrimMesh, int(meshes.size()) - 1});
    } else if (type == "quad") {
        QuadSrc q{transform, fetchBsdf(v["bsdf"]), emissionTex, intMedium, extMedium};
        _quadSrc.push_back(q);
        _primOrder.push_back({PrimQuad, int(_quadSrc.size()) - 1});
    } else if (type == "cube") {
        CubeSrc c{transform, fetchBsdf(v["bsdf"]), emissionTex, intMedium, extMedium};
        _cubeSrc.push_back(c);
        _primOrder.push_back({PrimCube, int(_cubeSrc.size()) - 1});
    } else if (type == "sphere") {
        SphereSrc sp{transform, fetchBsdf(v["bsdf"]), emissionTex, intMedium, extMedium};
        _sphereSrc.push_back(sp);
        _primOrder.push_back({PrimSphere, int(_sphereSrc.size()) - 1});
    } else if (type == "disk") {
        DiskSrc dk{transform, fetchBsdf(v["bsdf"]), emissionTex, 90.0f, intMedium, extMedium};
        v.getField("cone_angle", dk.coneAngle);
        _diskSrc.push_back(dk);
        _primOrder.push_back({PrimDisk, int(_diskSrc.size()) - 1});
    } else if (type == "instances") { // Instance::fromJson (Instance.cpp:60-93)
        if (emissionTex >= 0) throw std::runtime_error("emissive 'instances' primitives are outside the hot-path scope");
        if (intMedium >= 0 || extMedium >= 0) throw std::runtime_error("media on 'instances' primitives are outside the hot-path scope");
        if (v["instancesB"]) throw std::runtime_error("'instancesB'/'ratio' blending is outside the hot-path scope");
        InstanceSrc src;
        src.transform = transform;
        const Json &masters = v["masters"];
        for (size_t i = 0; masters && i < masters.size(); ++i) {
            const Json &m = masters[i];
            std::string mt = m["type"].asString();
            Mat4 mtf;
            if (m["transform"]) mtf = parseTransform(m["transform"]);
            if (m["int_medium"] || m["ext_medium"]) throw std::runtime_error("media on instance masters are outside the hot-path scope");
            MasterSrc ms{};
            if (mt == "mesh") { ms.kind = TGPU_MASTER_MESH; ms.mesh = parseMesh(m, mtf, -1); }
            else if (mt == "quad") { ms.kind = TGPU_MASTER_QUAD; ms.quad = QuadSrc{mtf, fetchBsdf(m["bsdf"]), -1, -1, -1}; }
            else if (mt == "cube") { ms.kind = TGPU_MASTER_CUBE; ms.cube = CubeSrc{mtf, fetchBsdf(m["bsdf"]), -1, -1, -1}; }
            else throw std::runtime_error("instance master of type '" + mt + "' is outside the hot-path scope");
            src.masters.push_back(std::move(ms));
        }
        if (src.masters.empty() || src.masters.size() > 256) throw std::runtime_error("'instances' needs 1..256 masters");
        std::vector<Quat> rot;
        const Json &inst = v["instances"] ? v["instances"] : v["instancesA"];
        if (inst.isString()) {
            std::string file = inst.asString();
            std::string full = file.size() && file[0] == '/' ? file : sceneDir + "/" + file;
            if (!loadInstanceFile(full, src.pos, rot, src.id)) throw std::runtime_error("Unable to load instances at " + full);
        } else if (inst) {
            for (size_t i = 0; i < inst.size(); ++i) {
                int id = 0;
                inst[i].getField("id", id);
                Mat4 tf;
                if (inst[i]["transform"]) tf = parseTransform(inst[i]["transform"]);
                Vec3 t = tf.translation();
                src.pos.push_back(t.x); src.pos.push_back(t.y); src.pos.push_back(t.z);
                rot.push_back(Quat::fromMatrix(tf.extractRotation()));
                src.id.push_back(uint8_t(id));
            }
        }
        for (uint8_t id : src.id)
            if (id >= src.masters.size()) throw std::runtime_error("'instances': master id out of range");
        for (const Quat &q : rot) for (int k = 0; k < 4; ++k) src.rot.push_back(q.v[k]);
        _instanceSrc.push_back(std::move(src));
    } else if (type == "infinite_sphere") {
        if (_envSrc.present) throw std::runtime_error("more than one infinite_sphere is outside the hot-path scope");
        if (intMedium >= 0 || extMedium >= 0) throw std::runtime_error("media on infinite_sphere are outside the hot-path scope");
        _envSrc.present = true;
        _envSrc.transform = transform;
        _envSrc.emissionTex = emissionTex;
        _envSrc.sample = true;
        v.getField("sample", _envSrc.sample);
        _primOrder.push_back({PrimEnv, 0});
    } else {
        throw std::runtime_error("Primitive type '" + type + "' is outside the hot-path scope (SURVEY.md section 2 row 6)");
    }
}

void Scene::parseCamera(const Json &v)
{
    std::string type = "pinhole";
    v.getField("type", type);
    if (type != "pinhole") throw std::runtime_error("Camera type '" + type + "' is outside the hot-path scope");
    v.getField("tonemap", tonemap);
    float res[2] = {float(resX), float(resY)};
    if (v.getVec("resolution", res, 2)) {
        resX = uint32_t(res[0]);
        resY = uint32_t(res[1]);
    }
    if (v["medium"]) _cameraMedium = fetchMedium(v["medium"]);
    v.getField("reconstruction_filter", filterName);
    if (v["transform"]) camTransform = parseTransform(v["transform"]);
    v.getField("fov", fovDeg);
}

Scene Scene::load(const std::string &jsonPath, const std::string &inputDirectory)
{
    return loadText(readFile(jsonPath), inputDirectory.empty() ? dirName(jsonPath) : inputDirectory);
}

Scene Scene::loadText(const std::string &jsonText, const std::string &resourceDirectory)
{
    Scene s;
    s.sceneDir = resourceDirectory;
    s.sceneJsonText = jsonText;
    Json root = Json::parse(s.sceneJsonText);

    const Json &media = root["media"];
    for (size_t i = 0; media && i < media.size(); ++i) {
        int id = s.parseMedium(media[i]);
        std::string name;
        if (media[i].getField("name", name)) s._namedMedia.emplace_back(name, id);
    }
    const Json &bsdfs = root["bsdfs"];
    for (size_t i = 0; bsdfs && i < bsdfs.size(); ++i) {
        int id = s.parseBsdf(bsdfs[i]);
        std::string name;
        if (bsdfs[i].getField("name", name)) s._namedBsdfs.emplace_back(name, id);
    }
    const Json &prims = root["primitives"];
    for (size_t i = 0; prims && i < prims.size(); ++i) s.parsePrimitive(prims[i]);
    if (root["camera"]) s.parseCamera(root["camera"]);

    // PathTracerSettings / TraceSettings defaults
    s.settings.min_bounces = 0;
    s.settings.max_bounces = 64;
    s.settings.enable_light_sampling = 1;
    s.settings.enable_two_sided_shading = 1;
    s.settings.enable_consistency_checks = 0;
    s.settings.enable_volume_light_sampling = 1;
    if (const Json &it = root["integrator"]) {
        std::string type = "path_tracer";
        it.getField("type", type);
        if (type != "path_tracer")
            throw std::runtime_error("Integrator '" + type + "' is out of scope: this core implements path_tracer only");
        bool b;
        it.getField("min_bounces", s.settings.min_bounces);
        it.getField("max_bounces", s.settings.max_bounces);
        if (it.getField("enable_light_sampling", b)) s.settings.enable_light_sampling = b;
        if (it.getField("enable_two_sided_shading", b)) s.settings.enable_two_sided_shading = b;
        if (it.getField("enable_consistency_checks", b)) s.settings.enable_consistency_checks = b;
        s.settings.enable_volume_light_sampling = 1;
        if (it.getField("enable_volume_light_sampling", b)) s.settings.enable_volume_light_sampling = b;
        bool flag = true;
        if (it.getField("include_surfaces", flag) && !flag) throw std::runtime_error("include_surfaces=false is outside the hot-path scope");
        flag = true;
        if (it.getField("low_order_scattering", flag) && !flag) throw std::runtime_error("low_order_scattering=false is outside the hot-path scope");
    }
    if (const Json &r = root["renderer"]) {
        r.getField("output_file", s.renderer.outputFile);
        r.getField("hdr_output_file", s.renderer.hdrOutputFile);
        r.getField("resume_render_file", s.renderer.resumeRenderFile);
        r.getField("overwrite_output_files", s.renderer.overwriteOutputFiles);
        r.getField("adaptive_sampling", s.renderer.useAdaptiveSampling);
        r.getField("enable_resume_render", s.renderer.enableResumeRender);
        r.getField("stratified_sampler", s.renderer.useSobol);
        r.getField("scene_bvh", s.renderer.useSceneBvh);
        int v;
        if (r.getField("spp", v)) s.renderer.spp = uint32_t(v);
        if (r.getField("spp_step", v)) s.renderer.sppStep = uint32_t(v);
        if (r["checkpoint_interval"].isString()) s.renderer.checkpointInterval = r["checkpoint_interval"].asString();
        if (r["timeout"].isString()) s.renderer.timeout = r["timeout"].asString();
        if (const Json &outputs = r["output_buffers"]) { // RendererSettings.hpp:69-74, OutputBufferSettings.cpp:8-32
            for (size_t i = 0; i < outputs.size(); ++i) {
                const Json &o = outputs[i];
                OutputBufferSettings b;
                std::string type;
                if (!o.getField("type", type)) throw std::runtime_error("output_buffers: entry without \"type\"");
                static const char *names[TGPU_OUTPUT_COUNT] = {"color", "depth", "normal", "albedo", "visibility"};
                b.type = -1;
                for (int t = 0; t < TGPU_OUTPUT_COUNT; ++t)
                    if (type == names[t]) b.type = t;
                if (b.type < 0) throw std::runtime_error("Invalid output buffer type: '" + type + "'");
                o.getField("ldr_output_file", b.ldrOutputFile);
                o.getField("hdr_output_file", b.hdrOutputFile);
                o.getField("two_buffer_variance", b.twoBufferVariance);
                o.getField("sample_variance", b.sampleVariance);
                s.renderer.outputs.push_back(b);
            }
        }
    }
    return s;
}

void Scene::setResolution(uint32_t w, uint32_t h)
{
    resX = w;
    resY = h;
}

// ---------------------------------------------------------------------------
// flatten: prepareForRender bakes
// ---------------------------------------------------------------------------
static void storeMat3(const Mat4 &m, float *dst)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dst[i*3 + j] = m(i, j);
}

static float filterEval(const std::string &name, float x) // ReconstructionFilter.hpp:187-209
{
    if (name == "tent") return 1.0f - std::abs(x);
    if (name == "gaussian") {
        const float Alpha = 2.0f;
        return std::max(std::exp(-Alpha*x*x) - std::exp(-Alpha*4.0f), 0.0f);
    }
    x = std::abs(x);
    if (name == "mitchell_netravali") {
        const float B = 1.0f/3.0f, C = 1.0f/3.0f;
        if (x < 1.0f)
            return 1.0f/6.0f*((12.0f - 9.0f*B - 6.0f*C)*x*x*x + (-18.0f + 12.0f*B + 6.0f*C)*x*x + (6.0f - 2.0f*B));
        else if (x < 2.0f)
            return 1.0f/6.0f*((-B - 6.0f*C)*x*x*x + (6.0f*B + 30.0f*C)*x*x + (-12.0f*B - 48.0f*C)*x + (8.0f*B + 24.0f*C));
        return 0.0f;
    }
    if (name == "catmull_rom") {
        if (x < 1.0f) return 1.0f/6.0f*((12.0f - 3.0f)*x*x*x + (-18.0f + 3.0f)*x*x + 6.0f);
        else if (x < 2.0f) return 1.0f/6.0f*(-3.0f*x*x*x + 15.0f*x*x - 24.0f*x + 12.0f);
        return 0.0f;
    }
    if (name == "lanczos") {
        if (x == 0.0f) return 1.0f;
        else if (x < 2.0f) return std::sin(PI*x)*std::sin(PI*x/2.0f)/(PI*PI*x*x/2.0f);
        return 0.0f;
    }
    throw JsonLoadException("Unknown reconstruction filter '" + name + "'");
}

void Scene::bakeMesh(const MeshData &m, float *boundsMin, float *boundsMax)
{
    size_t nv = m.verts.size()/8, nt = m.tris.size()/4;
    Mat4 normalTform = m.transform.toNormalMatrix();
    std::vector<Vec3> p(nv), n(nv);
    if (boundsMin) for (int k = 0; k < 3; ++k) { boundsMin[k] = std::numeric_limits<float>::max(); boundsMax[k] = std::numeric_limits<float>::lowest(); }
    for (size_t i = 0; i < nv; ++i) {
        const float *v = &m.verts[i*8];
        p[i] = m.transform*Vec3(v[0], v[1], v[2]);
        n[i] = normalTform.transformVector(Vec3(v[3], v[4], v[5]));
        if (boundsMin) // TriangleMesh::computeBounds :233-239 (every vertex, referenced or not)
            for (int k = 0; k < 3; ++k) { boundsMin[k] = std::min(boundsMin[k], p[i][k]); boundsMax[k] = std::max(boundsMax[k], p[i][k]); }
    }
    for (size_t i = 0; i < nt; ++i) {
        const uint32_t *tri = &m.tris[i*4];
        int material = int(int32_t(tri[3]));
        material = std::min(std::max(material, 0), int(m.bsdfs.size()) - 1);
        for (int k = 0; k < 3; ++k) {
            uint32_t vi = tri[k];
            if (vi >= nv) throw std::runtime_error("mesh '" + m.name + "' has an out-of-range vertex index");
            _triPos.push_back(p[vi].x); _triPos.push_back(p[vi].y); _triPos.push_back(p[vi].z);
            _triNormal.push_back(n[vi].x); _triNormal.push_back(n[vi].y); _triNormal.push_back(n[vi].z);
            _triUv.push_back(m.verts[size_t(vi)*8 + 6]); _triUv.push_back(m.verts[size_t(vi)*8 + 7]);
        }
        _triBsdf.push_back(m.bsdfs[size_t(material)]);
        _triSmooth.push_back(m.smooth ? 1 : 0);
        _triMedia.push_back(int16_t(m.intMedium));
        _triMedia.push_back(int16_t(m.extMedium));
    }
}

tgpu_quad Scene::bakeQuad(const QuadSrc &q) const
{
    tgpu_quad o{};
    Vec3 base = q.transform*Vec3(0.0f);
    Vec3 edge0 = q.transform.transformVector(Vec3(1.0f, 0.0f, 0.0f));
    Vec3 edge1 = q.transform.transformVector(Vec3(0.0f, 0.0f, 1.0f));
    base -= edge0*0.5f;
    base -= edge1*0.5f;
    Vec3 n = edge1.cross(edge0);
    float area = n.length();
    n /= area;
    Vec3 tg = edge0.normalized(), bt = edge1.normalized();
    for (int i = 0; i < 3; ++i) {
        o.base[i] = base[i]; o.edge0[i] = edge0[i]; o.edge1[i] = edge1[i];
        o.normal[i] = n[i]; o.tangent[i] = tg[i]; o.bitangent[i] = bt[i];
    }
    o.area = area;
    o.inv_uv_sq[0] = 1.0f/edge0.lengthSq();
    o.inv_uv_sq[1] = 1.0f/edge1.lengthSq();
    o.bsdf = q.bsdf;
    o.emission_tex = q.emissionTex;
    o.int_medium = q.intMedium;
    o.ext_medium = q.extMedium;
    return o;
}

tgpu_cube Scene::bakeCube(const CubeSrc &c) const
{
    tgpu_cube o{};
    Vec3 p = c.transform*Vec3(0.0f);
    Vec3 sc = c.transform.extractScale()*Vec3(0.5f);
    Mat4 rot = c.transform.extractRotation();
    Mat4 inv = rot.transpose();
    storeMat3(rot, o.rot);
    storeMat3(inv, o.inv_rot);
    for (int i = 0; i < 3; ++i) { o.pos[i] = p[i]; o.scale[i] = sc[i]; }
    o.bsdf = c.bsdf;
    o.emission_tex = c.emissionTex;
    o.int_medium = c.intMedium;
    o.ext_medium = c.extMedium;
    return o;
}

void Scene::flatten()
{
    const int RFILTER_RESOLUTION = 31;
    _desc = tgpu_scene_desc{};
    _desc.abi_version = TGPU_ABI_VERSION;

    // ---- camera: Camera::fromJson flips the right axis (Camera.cpp:64), then precompute ----
    tgpu_camera &cam = _desc.camera;
    Mat4 t = camTransform;
    Vec3 pos = t.translation();
    t.setRight(-t.right());
    cam.pos[0] = pos.x; cam.pos[1] = pos.y; cam.pos[2] = pos.z;
    storeMat3(t, cam.rot);
    cam.res_x = resX;
    cam.res_y = resY;
    cam.ratio = resY/float(resX);
    cam.pixel_size_x = 1.0f/resX;
    float fovRad = fovDeg*(PI/180.0f);
    cam.plane_dist = 1.0f/std::tan(fovRad*0.5f);
    if (filterName == "dirac") {
        cam.filter_type = TGPU_FILTER_DIRAC;
    } else if (filterName == "box") {
        cam.filter_type = TGPU_FILTER_BOX;
    } else { // ReconstructionFilter.cpp:34-58
        cam.filter_type = TGPU_FILTER_TABULATED;
        float width = filterName == "tent" ? 1.0f : 2.0f;
        cam.filter_bin_size = width/RFILTER_RESOLUTION;
        float filter[RFILTER_RESOLUTION + 1];
        float filterSum = 0.0f;
        for (int i = 0; i < RFILTER_RESOLUTION; ++i) {
            filter[i] = filterEval(filterName, (i*width)/RFILTER_RESOLUTION);
            filterSum += filter[i];
        }
        cam.filter_cdf[0] = 0.0f;
        for (int i = 1; i < RFILTER_RESOLUTION; ++i) cam.filter_cdf[i] = cam.filter_cdf[i - 1] + filter[i - 1]/filterSum;
        cam.filter_cdf[RFILTER_RESOLUTION] = 1.0f;
    }

    _desc.settings = settings;
    _desc.settings.use_sobol = renderer.useSobol ? 1 : 0;
    _desc.settings.adaptive_sampling = renderer.useAdaptiveSampling ? 1 : 0;
    for (tgpu_output_buffer &b : _desc.output_buffers) b = tgpu_output_buffer{0, 0, 0};
    _desc.output_buffers[TGPU_OUTPUT_COLOR].enabled = 1;
    for (const OutputBufferSettings &b : renderer.outputs) // Camera::requestOutputBuffers (Camera.cpp:148-161): a later entry of a type replaces an earlier one
        _desc.output_buffers[b.type] = tgpu_output_buffer{1, b.twoBufferVariance ? 1 : 0, b.sampleVariance ? 1 : 0};

    // ---- meshes: world-space bake (TriangleMesh.cpp:524-552) ----
    _triPos.clear(); _triNormal.clear(); _triUv.clear(); _triBsdf.clear(); _triSmooth.clear(); _triMedia.clear();
    _meshFirstTri.assign(1, 0u);
    _meshBounds.clear();
    for (const MeshData &m : meshes) {
        float lo[3], hi[3];
        bakeMesh(m, lo, hi);
        _meshFirstTri.push_back(uint32_t(_triBsdf.size()));
        _meshBounds.insert(_meshBounds.end(), lo, lo + 3);
        _meshBounds.insert(_meshBounds.end(), hi, hi + 3);
    }
    _desc.n_tris = uint32_t(_triBsdf.size());
    _desc.n_meshes = uint32_t(meshes.size());

    // ---- quads (Quad.cpp:298-316), cubes (Cube.cpp:353-370) ----
    _quads.clear();
    for (const QuadSrc &q : _quadSrc) _quads.push_back(bakeQuad(q));
    _desc.n_quads = uint32_t(_quads.size());
    _cubes.clear();
    for (const CubeSrc &c : _cubeSrc) _cubes.push_back(bakeCube(c));
    _desc.n_cubes = uint32_t(_cubes.size());

    // ---- spheres (Sphere.cpp:285-295), disks (Disk.cpp:315-327) ----
    _spheres.clear();
    for (const SphereSrc &sp : _sphereSrc) {
        tgpu_sphere o{};
        Vec3 p = sp.transform*Vec3(0.0f);
        Vec3 sc3 = sp.transform.extractScale()*Vec3(1.0f);
        float radius = sc3.x; if (sc3.y > radius) radius = sc3.y; if (sc3.z > radius) radius = sc3.z; // Vec::max()
        Mat4 rot = sp.transform.extractRotation();
        Mat4 inv = rot.transpose();
        storeMat3(rot, o.rot);
        storeMat3(inv, o.inv_rot);
        for (int i = 0; i < 3; ++i) o.pos[i] = p[i];
        o.radius = radius;
        o.bsdf = sp.bsdf; o.emission_tex = sp.emissionTex; o.int_medium = sp.intMedium; o.ext_medium = sp.extMedium;
        _spheres.push_back(o);
    }
    _desc.n_spheres = uint32_t(_spheres.size());
    _desc.spheres = _spheres.data();
    _disks.clear();
    for (const DiskSrc &dk : _diskSrc) {
        tgpu_disk o{};
        Vec3 c = dk.transform*Vec3(0.0f);
        Vec3 sc3 = dk.transform.extractScale()*Vec3(1.0f, 0.0f, 1.0f);
        float r = sc3.x; if (sc3.y > r) r = sc3.y; if (sc3.z > r) r = sc3.z;
        Vec3 n = dk.transform.transformVector(Vec3(0.0f, 1.0f, 0.0f)).normalized();
        // TangentFrame(n) (math/TangentFrame.hpp:22-31, Duff et al.)
        float sign = std::copysign(1.0f, n.z);
        const float a = -1.0f/(sign + n.z);
        const float b = n.x*n.y*a;
        Vec3 tangent(1.0f + sign*n.x*n.x*a, sign*b, -sign*n.x);
        Vec3 bitangent(b, sign + n.y*n.y*a, -n.y);
        float coneRad = dk.coneAngle*(PI/180.0f); // Angle::degToRad (Angle.hpp:26-29)
        Vec3 coneBase = c - n/std::sin(coneRad);
        for (int i = 0; i < 3; ++i) { o.center[i] = c[i]; o.n[i] = n[i]; o.tangent[i] = tangent[i]; o.bitangent[i] = bitangent[i]; o.cone_base[i] = coneBase[i]; }
        o.r = r;
        o.cos_apex = std::cos(coneRad);
        o.bsdf = dk.bsdf; o.emission_tex = dk.emissionTex; o.int_medium = dk.intMedium; o.ext_medium = dk.extMedium;
        _disks.push_back(o);
    }
    _desc.n_disks = uint32_t(_disks.size());
    _desc.disks = _disks.data();

    // ---- "instances": masters go to the tail of the geometry arrays; Instance::prepareForRender :392-403 ----
    _instanceFlat.clear();
    _instanceSets.clear();
    for (const InstanceSrc &src : _instanceSrc) {
        _instanceFlat.emplace_back();
        InstanceFlat &f = _instanceFlat.back();
        for (const MasterSrc &ms : src.masters) {
            tgpu_master m{};
            m.kind = ms.kind;
            if (ms.kind == TGPU_MASTER_MESH) {
                m.first = uint32_t(_triBsdf.size());
                bakeMesh(ms.mesh, m.bounds_min, m.bounds_max);
                m.count = uint32_t(_triBsdf.size()) - m.first;
                if (m.count == 0) throw std::runtime_error("instance master mesh has no triangles");
            } else if (ms.kind == TGPU_MASTER_QUAD) {
                m.first = uint32_t(_quads.size());
                m.count = 1;
                tgpu_quad q = bakeQuad(ms.quad);
                _quads.push_back(q);
                Vec3 base(q.base[0], q.base[1], q.base[2]), e0(q.edge0[0], q.edge0[1], q.edge0[2]), e1(q.edge1[0], q.edge1[1], q.edge1[2]);
                Vec3 corners[4] = {base, base + e0, base + e1, base + e0 + e1}; // Quad::bounds :281-289
                for (int k = 0; k < 3; ++k) { m.bounds_min[k] = std::numeric_limits<float>::max(); m.bounds_max[k] = std::numeric_limits<float>::lowest(); }
                for (const Vec3 &c : corners)
                    for (int k = 0; k < 3; ++k) { m.bounds_min[k] = std::min(m.bounds_min[k], c[k]); m.bounds_max[k] = std::max(m.bounds_max[k], c[k]); }
            } else {
                m.first = uint32_t(_cubes.size());
                m.count = 1;
                tgpu_cube c = bakeCube(ms.cube);
                _cubes.push_back(c);
                for (int k = 0; k < 3; ++k) { m.bounds_min[k] = std::numeric_limits<float>::max(); m.bounds_max[k] = std::numeric_limits<float>::lowest(); }
                for (int i = 0; i < 8; ++i) { // Cube::bounds :333-344 (_rot*v is the point transform of a rotation matrix)
                    Vec3 v((i & 1 ? c.scale[0] : -c.scale[0]), (i & 2 ? c.scale[1] : -c.scale[1]), (i & 4 ? c.scale[2] : -c.scale[2]));
                    Vec3 r(c.rot[0]*v.x + c.rot[1]*v.y + c.rot[2]*v.z + 0.0f, c.rot[3]*v.x + c.rot[4]*v.y + c.rot[5]*v.z + 0.0f,
                           c.rot[6]*v.x + c.rot[7]*v.y + c.rot[8]*v.z + 0.0f);
                    for (int k = 0; k < 3; ++k) {
                        float w = c.pos[k] + r[k];
                        m.bounds_min[k] = std::min(m.bounds_min[k], w);
                        m.bounds_max[k] = std::max(m.bounds_max[k], w);
                    }
                }
            }
            f.masters.push_back(m);
        }
        Quat own = Quat::fromMatrix(src.transform.extractRotation());
        size_t n = src.id.size();
        f.pos.resize(n*3); f.rot.resize(n*4); f.id = src.id;
        for (size_t i = 0; i < n; ++i) {
            Vec3 p = src.transform*Vec3(src.pos[i*3], src.pos[i*3 + 1], src.pos[i*3 + 2]);
            Quat q = own*Quat{{src.rot[i*4], src.rot[i*4 + 1], src.rot[i*4 + 2], src.rot[i*4 + 3]}};
            for (int k = 0; k < 3; ++k) f.pos[i*3 + k] = p[k];
            for (int k = 0; k < 4; ++k) f.rot[i*4 + k] = q.v[k];
        }
    }
    for (InstanceFlat &f : _instanceFlat) {
        tgpu_instances is{};
        is.n_masters = uint32_t(f.masters.size());
        is.masters = f.masters.data();
        is.n_instances = uint32_t(f.id.size());
        is.pos = f.pos.data();
        is.rot = f.rot.data();
        is.master_id = f.id.data();
        _instanceSets.push_back(is);
    }
    _desc.n_instance_sets = uint32_t(_instanceSets.size());
    _desc.instance_sets = _instanceSets.data();
    _desc.n_master_tris = uint32_t(_triBsdf.size()) - _desc.n_tris;
    _desc.n_master_quads = uint32_t(_quads.size()) - _desc.n_quads;
    _desc.n_master_cubes = uint32_t(_cubes.size()) - _desc.n_cubes;
    _desc.tri_pos = _triPos.data();
    _desc.mesh_first_tri = _meshFirstTri.data();
    _desc.mesh_bounds = _meshBounds.data();
    _desc.tri_normal = _triNormal.data();
    _desc.tri_uv = _triUv.data();
    _desc.tri_bsdf = _triBsdf.data();
    _desc.tri_smooth = _triSmooth.data();
    _desc.n_media = uint32_t(_media.size());
    _desc.media = _media.data();
    _desc.tri_media = _media.empty() ? nullptr : _triMedia.data();
    cam.medium = _cameraMedium;
    _desc.quads = _quads.data();
    _desc.cubes = _cubes.data();

    // ---- lights in primitive order (TraceableScene.hpp:86-102) ----
    _lights.clear();
    int lightCount = 0;
    for (const PrimRef &pr : _primOrder) {
        if (pr.kind == PrimQuad && _quadSrc[size_t(pr.index)].emissionTex >= 0) {
            _lights.push_back({TGPU_LIGHT_QUAD, pr.index});
            ++lightCount;
        } else if (pr.kind == PrimSphere && _sphereSrc[size_t(pr.index)].emissionTex >= 0) {
            _lights.push_back({TGPU_LIGHT_SPHERE, pr.index});
            ++lightCount;
        } else if (pr.kind == PrimDisk && _diskSrc[size_t(pr.index)].emissionTex >= 0) {
            _lights.push_back({TGPU_LIGHT_DISK, pr.index});
            ++lightCount;
        } else if (pr.kind == PrimCube && _cubeSrc[size_t(pr.index)].emissionTex >= 0) {
            throw std::runtime_error("emissive cubes are outside the hot-path scope");
        } else if (pr.kind == PrimEnv && _envSrc.emissionTex >= 0) {
            ++lightCount;
            if (_envSrc.sample) _lights.push_back({TGPU_LIGHT_ENV, 0});
        }
    }
    EnvSrc env = _envSrc;
    if (env.present && env.emissionTex < 0) env.present = false; // non-emissive sphere never contributes
    if (lightCount == 0) { // default white environment (TraceableScene.hpp:97-102)
        Json one; one.type = Json::Number; one.num = 1.0;
        env = EnvSrc{};
        env.present = true;
        env.sample = true;
        env.emissionTex = fetchTexture(one, false);
        _lights.push_back({TGPU_LIGHT_ENV, 0});
    }
    _desc.n_lights = uint32_t(_lights.size());
    _desc.lights = _lights.data();

    // ---- environment (InfiniteSphere.cpp:280-286, BitmapTexture.cpp:400-431, Distribution2D.hpp:18-64) ----
    tgpu_env &e = _desc.env;
    e = tgpu_env{};
    e.present = env.present ? 1 : 0;
    if (env.present) {
        e.samplable = env.sample ? 1 : 0;
        e.emission_tex = env.emissionTex;
        Mat4 rot = env.transform.extractRotation();
        Mat4 inv = rot.transpose();
        storeMat3(rot, e.rot);
        storeMat3(inv, e.inv_rot);
        const tgpu_texture &tex = _textures[size_t(env.emissionTex)];
        if (tex.type == TGPU_TEX_CHECKER) throw std::runtime_error("checker emission on infinite_sphere is outside the hot-path scope");
        if (tex.type == TGPU_TEX_BITMAP && env.sample) {
            const tgpu_bitmap &bm = _bitmaps[size_t(tex.bitmap)];
            const std::vector<float> &px = _bitmapTexels[size_t(tex.bitmap)];
            int w = bm.w, h = bm.h;
            std::vector<float> weights(size_t(w)*h);
            for (int y = 0, idx = 0; y < h; ++y) {
                float rowWeight = 1.0f;
                rowWeight *= std::sin((y*PI)/h);
                for (int x = 0; x < w; ++x, ++idx) {
                    const float *c = &px[size_t(idx)*3];
                    float m = c[0]; if (c[1] > m) m = c[1]; if (c[2] > m) m = c[2];
                    weights[size_t(idx)] = m*rowWeight;
                }
            }
            auto W = [&](int x, int y) -> float & { return weights[size_t(x) + size_t(y)*w]; };
            for (int y = 0; y < h; ++y) {
                for (int x = 0; x < w - 1; ++x) W(x, y) = std::max(W(x, y), W(x + 1, y));
                if (!bm.clamp) W(0, y) = W(w - 1, y) = std::max(W(w - 1, y), W(0, y));
                for (int x = w - 1; x > 0; --x) W(x, y) = std::max(W(x, y), W(x - 1, y));
            }
            for (int x = 0; x < w; ++x) {
                for (int y = 0; y < h - 1; ++y) W(x, y) = std::max(W(x, y), W(x, y + 1));
                if (!bm.clamp) W(x, 0) = W(x, h - 1) = std::max(W(x, 0), W(x, h - 1));
                for (int y = h - 1; y > 0; --y) W(x, y) = std::max(W(x, y), W(x, y - 1));
            }
            // Distribution2D ctor
            _envPdf = std::move(weights);
            _envCdf.assign(_envPdf.size() + size_t(h), 0.0f);
            _envMarginalPdf.assign(size_t(h), 0.0f);
            _envMarginalCdf.assign(size_t(h) + 1, 0.0f);
            _envMarginalCdf[0] = 0.0f;
            for (int y = 0; y < h; ++y) {
                size_t idxP = size_t(y)*w, idxC = size_t(y)*(w + 1);
                _envCdf[idxC] = 0.0f;
                for (int x = 0; x < w; ++x, ++idxP, ++idxC) {
                    _envMarginalPdf[size_t(y)] += _envPdf[idxP];
                    _envCdf[idxC + 1] = _envCdf[idxC] + _envPdf[idxP];
                }
                _envMarginalCdf[size_t(y) + 1] = _envMarginalCdf[size_t(y)] + _envMarginalPdf[size_t(y)];
            }
            for (int y = 0; y < h; ++y) {
                size_t idxP = size_t(y)*w, idxC = size_t(y)*(w + 1);
                size_t idxTail = idxC + size_t(w);
                float rowWeight = _envCdf[idxTail];
                if (rowWeight < 1e-4f) {
                    for (int x = 0; x < w; ++x, ++idxP, ++idxC) {
                        _envPdf[idxP] = 1.0f/w;
                        _envCdf[idxC] = x/float(w);
                    }
                } else {
                    for (int x = 0; x < w; ++x, ++idxP, ++idxC) {
                        _envPdf[idxP] /= rowWeight;
                        _envCdf[idxC] /= rowWeight;
                    }
                }
                _envCdf[idxTail] = 1.0f;
            }
            float totalWeight = _envMarginalCdf.back();
            for (float &p : _envMarginalPdf) p /= totalWeight;
            for (float &c : _envMarginalCdf) c /= totalWeight;
            _envMarginalCdf.back() = 1.0f;
            e.dist_w = w;
            e.dist_h = h;
            e.pdf = _envPdf.data();
            e.cdf = _envCdf.data();
            e.marginal_pdf = _envMarginalPdf.data();
            e.marginal_cdf = _envMarginalCdf.data();
        }
    }

    for (size_t i = 0; i < _bitmaps.size(); ++i) _bitmaps[i].texels = _bitmapTexels[i].data();
    _desc.n_bsdfs = uint32_t(_bsdfs.size());
    _desc.bsdfs = _bsdfs.data();
    _desc.n_textures = uint32_t(_textures.size());
    _desc.textures = _textures.data();
    _desc.n_bitmaps = uint32_t(_bitmaps.size());
    _desc.bitmaps = _bitmaps.data();

    // ---- Sobol' matrices asset ----
    if (_sobol.empty()) {
        std::string path = assetDir() + "/sobol_matrices_1024x32.bin";
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("Unable to open Sobol' matrix asset '" + path + "' (set TUNGSTEN_AMD_ASSETS)");
        _sobol.resize(1024*32);
        size_t got = std::fread(_sobol.data(), 4, _sobol.size(), f);
        std::fclose(f);
        if (got != _sobol.size()) throw std::runtime_error("Sobol' matrix asset '" + path + "' is truncated");
    }
    _desc.sobol_matrices = _sobol.data();
}

// ---------------------------------------------------------------------------
// shard
// ---------------------------------------------------------------------------
std::vector<uint32_t> tileSeeds(uint32_t resX, uint32_t resY, uint32_t seed)
{
    std::vector<uint32_t> seeds;
    Pcg32 sampler(hash32(seed));
    for (uint32_t y = 0; y < resY; y += 16)
        for (uint32_t x = 0; x < resX; x += 16) seeds.push_back(hash32(sampler.nextI()));
    return seeds;
}

Shard makeShard(uint32_t resX, uint32_t resY, uint32_t seed, uint32_t rank, uint32_t world)
{
    const uint32_t TileSize = 16;
    Shard s;
    Pcg32 sampler(hash32(seed)); // PathTraceIntegrator.cpp:187
    uint32_t tileId = 0;
    for (uint32_t y = 0; y < resY; y += TileSize) {
        for (uint32_t x = 0; x < resX; x += TileSize, ++tileId) {
            uint32_t tileSeed = hash32(sampler.nextI()); // :36-38, drawn for every tile in row-major order
            if (world > 1 && tileId % world != rank) continue;
            uint32_t tw = std::min(TileSize, resX - x), th = std::min(TileSize, resY - y);
            // 8x8 blocks inside the tile keep one wavefront's 64 primary rays spatially compact
            for (uint32_t by = 0; by < th; by += 8)
                for (uint32_t bx = 0; bx < tw; bx += 8)
                    for (uint32_t py = by; py < std::min(by + 8, th); ++py)
                        for (uint32_t px = bx; px < std::min(bx + 8, tw); ++px) {
                            s.pixelIndex.push_back((x + px) + (y + py)*resX);
                            s.tileSeed.push_back(tileSeed);
                        }
        }
    }
    return s;
}

} // namespace tungsten_amd
