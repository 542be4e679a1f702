// GemScene.h — reader for the reference's scene directories: scene.json + .gem meshes +
// textures, producing the host-side scene description (SceneData) that either feeds the
// device core or is written out as an .rtxs dump.
//
// Formats and rules followed:
//   scene.json       GEMLoader::GEMScene::load / parseInstance, RTBase/GEMLoader.h:685-738: every
//                    top-level value that is not an array is a scene property, every array is
//                    a list of instances; inside an instance "filename" and "world" (16 floats,
//                    row-major) are structural, everything else is a material property.
//                    Values are kept as text (numbers as std::to_string(float)), :459-480.
//   .gem             GEMModelLoader::load, RTBase/GEMLoader.h:344-365: u32 magic 4058972161,
//                    u32 animated, u32 meshes; per mesh u32 properties x (string, string),
//                    u32 vertices x 44 bytes (position, normal, tangent, u, v), u32 indices
//                    x u32; strings are i32 length + bytes (:270-281).
//   materials        loadInstance, RTBase/SceneLoader.h:205-294, with its quirks: the checks are
//                    independent ifs; "dielectric" with roughness < 0.001 is Glass; an emission
//                    property makes the BSDF a light; a coating (integer-parsed
//                    "coatingThickness" > 0) builds a LayeredBSDF that the scene never sees,
//                    because only the base BSDF was pushed to the material list.
//   vertices         p = world.mulPoint(p), normal = (world^-1)^T.mulVec(normal).normalize(),
//                    triangles of zero area dropped, :298-334 (rtx_host_transform_instance).
//   camera           loadScene, RTBase/SceneLoader.h:342-360.
//   textures         Texture::load, RTBase/Imaging.h:39-78: ".hdr" in the name -> float RGB,
//                    otherwise 8-bit channels / 255; one Texture per file name (:193-203).
//                    Decoders here: Radiance RGBE (flat and run-length scanlines) and PNG
//                    (8/16-bit, grey/RGB/palette/alpha, non-interlaced; zlib inflates).
#pragma once

#include <zlib.h>

#include <cctype>
#include <cfloat>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>

#include "RayTracerHost.h"

namespace gem {

// ---- JSON ------------------------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false; float num = 0; std::string str; std::vector<Json> arr; std::map<std::string, Json> obj;
    // GEMJson::asStr: what a property value becomes
    std::string text() const
    {
        if (kind == Bool) return std::to_string(b);
        if (kind == Num) return std::to_string(num);
        if (kind == Str) return str;
        return "";
    }
};

class JsonReader {
    const std::string& s; size_t at = 0;
    void ws() { while (at < s.size() && std::isspace((unsigned char)s[at])) at++; }
    char look() const { return at < s.size() ? s[at] : 0; }
    void need(char c) { ws(); if (look() != c) fail(std::string("expected '") + c + "'"); at++; }
    [[noreturn]] void fail(const std::string& why) const { throw std::runtime_error("scene.json: " + why + " at byte " + std::to_string(at)); }
    Json string()
    {
        Json j; j.kind = Json::Str; at++;
        while (at < s.size() && s[at] != '"') j.str.push_back(s[at++]);     // the format has no escapes
        if (at >= s.size()) fail("unterminated string");
        at++; return j;
    }
    Json number()
    {
        size_t b = at;
        if (look() == '-') at++;
        while (std::isdigit((unsigned char)look())) at++;
        if (look() == '.') { at++; while (std::isdigit((unsigned char)look())) at++; }
        if (look() == 'e' || look() == 'E') { at++; if (look() == '+' || look() == '-') at++; while (std::isdigit((unsigned char)look())) at++; }
        Json j; j.kind = Json::Num;
        try { j.num = std::stof(s.substr(b, at - b)); } catch (...) { fail("bad number"); }
        return j;
    }
    Json value()
    {
        ws();
        const char c = look();
        if (c == '"') return string();
        if (c == '-' || std::isdigit((unsigned char)c)) return number();
        if (c == '[') {
            Json j; j.kind = Json::Arr; at++; ws();
            if (look() == ']') { at++; return j; }
            for (;;) { j.arr.push_back(value()); ws(); if (look() == ',') { at++; continue; } need(']'); return j; }
        }
        if (c == '{') {
            Json j; j.kind = Json::Obj; at++; ws();
            if (look() == '}') { at++; return j; }
            for (;;) {
                ws(); if (look() != '"') fail("expected a key");
                std::string key = string().str; need(':');
                j.obj[key] = value(); ws();
                if (look() == ',') { at++; continue; }
                need('}'); return j;
            }
        }
        if (s.compare(at, 4, "true") == 0) { at += 4; Json j; j.kind = Json::Bool; j.b = true; return j; }
        if (s.compare(at, 5, "false") == 0) { at += 5; Json j; j.kind = Json::Bool; j.b = false; return j; }
        if (s.compare(at, 4, "null") == 0) { at += 4; return Json(); }
        fail("unexpected character");
    }
public:
    explicit JsonReader(const std::string& text) : s(text) {}
    Json parse() { return value(); }
};

// ---- properties (GEMProperty, RTBase/GEMLoader.h:40-115) ---------------------------------------
struct Properties {
    std::vector<std::pair<std::string, std::string>> items;
    std::string text(const std::string& name) const { for (auto& p : items) if (p.first == name) return p.second; return ""; }
    float number(const std::string& name, float dflt) const { try { return std::stof(text(name)); } catch (...) { return dflt; } }
    int integer(const std::string& name, int dflt) const { try { return std::stoi(text(name)); } catch (...) { return dflt; } }
    void vector3(const std::string& name, float& x, float& y, float& z) const
    {
        std::stringstream ss(text(name)); std::string word; std::vector<float> v;
        while (std::getline(ss, word, ' ')) { float f; try { f = std::stof(word); } catch (...) { f = 0; } v.push_back(f); }
        while (v.size() < 3) v.push_back(0);
        x = v[0]; y = v[1]; z = v[2];
    }
};

// ---- .gem meshes ---------------------------------------------------------------------------------
struct StaticVertex { float position[3], normal[3], tangent[3], u, v; };
static_assert(sizeof(StaticVertex) == 44, "on-disk vertex record");
struct Mesh { Properties material; std::vector<StaticVertex> vertices; std::vector<uint32_t> indices; };
constexpr uint32_t kGemMagic = 4058972161u;

inline std::vector<Mesh> readModel(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open mesh " + path);
    auto u32 = [&]() { uint32_t v = 0; f.read(reinterpret_cast<char*>(&v), 4); if (!f) throw std::runtime_error(path + ": truncated"); return v; };
    auto str = [&]() { int32_t l = (int32_t)u32(); if (l < 0 || l > (1 << 24)) throw std::runtime_error(path + ": bad string length"); std::string s((size_t)l, '\0'); f.read(&s[0], l); if (!f) throw std::runtime_error(path + ": truncated"); return std::string(s.c_str()); };
    if (u32() != kGemMagic) throw std::runtime_error(path + " is not a GE Model File");
    if (u32() != 0) throw std::runtime_error(path + ": animated models carry no static vertices and cannot be rendered");
    const uint32_t n = u32();
    std::vector<Mesh> meshes(n);
    for (Mesh& m : meshes) {
        const uint32_t np = u32();
        for (uint32_t i = 0; i < np; i++) { std::string k = str(); std::string v = str(); m.material.items.emplace_back(k, v); }
        m.vertices.resize(u32());
        f.read(reinterpret_cast<char*>(m.vertices.data()), (std::streamsize)(m.vertices.size() * sizeof(StaticVertex)));
        m.indices.resize(u32());
        f.read(reinterpret_cast<char*>(m.indices.data()), (std::streamsize)(m.indices.size() * 4));
        if (!f) throw std::runtime_error(path + ": truncated");
    }
    return meshes;
}

// ---- images ----------------------------------------------------------------------------------------
struct Image { int width = 0, height = 0, channels = 0; std::vector<float> f; std::vector<unsigned char> u8; bool isFloat = false; };

inline bool readHdr(const std::string& path, Image& img)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string line; bool ok = false;
    std::getline(f, line);
    if (line.rfind("#?RADIANCE", 0) != 0 && line.rfind("#?RGBE", 0) != 0) return false;
    while (std::getline(f, line)) { if (line.empty() || line == "\r") break; if (line.rfind("FORMAT=32-bit_rle_rgbe", 0) == 0) ok = true; }
    if (!ok) return false;
    std::getline(f, line);
    int w = 0, h = 0;
    if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w < 1 || h < 1) return false;
    img.width = w; img.height = h; img.channels = 3; img.isFloat = true; img.f.assign((size_t)w * h * 3, 0.0f);
    std::vector<unsigned char> row((size_t)w * 4);
    auto decode = [&](int y) {
        for (int x = 0; x < w; x++) {
            const unsigned char* p = &row[(size_t)x * 4]; float* o = &img.f[((size_t)y * w + x) * 3];
            if (p[3]) { const float s = ldexpf(1.0f, (int)p[3] - (128 + 8)); o[0] = p[0] * s; o[1] = p[1] * s; o[2] = p[2] * s; }
        }
    };
    for (int y = 0; y < h; y++) {
        unsigned char head[4];
        f.read(reinterpret_cast<char*>(head), 4);
        if (!f) return false;
        const bool rle = w >= 8 && w < 32768 && head[0] == 2 && head[1] == 2 && !(head[2] & 0x80) && ((head[2] << 8) | head[3]) == w;
        if (!rle) {
            // flat pixels: this scanline (and by convention all others) is w x 4 bytes
            std::memcpy(row.data(), head, 4);
            f.read(reinterpret_cast<char*>(row.data() + 4), (std::streamsize)(w * 4 - 4));
            if (!f) return false;
        } else {
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    int count = f.get(); if (count < 0) return false;
                    if (count > 128) { count -= 128; int v = f.get(); if (v < 0 || x + count > w) return false; while (count--) row[(size_t)(x++) * 4 + c] = (unsigned char)v; }
                    else { if (count == 0 || x + count > w) return false; while (count--) { int v = f.get(); if (v < 0) return false; row[(size_t)(x++) * 4 + c] = (unsigned char)v; } }
                }
            }
        }
        decode(y);
    }
    return true;
}

inline bool readPng(const std::string& path, Image& img)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::vector<unsigned char> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) return false;
    auto be32 = [&](size_t o) { return ((uint32_t)d[o] << 24) | ((uint32_t)d[o + 1] << 16) | ((uint32_t)d[o + 2] << 8) | d[o + 3]; };
    uint32_t w = 0, h = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, palette, trns;
    for (size_t o = 8; o + 12 <= d.size();) {
        const uint32_t len = be32(o); const std::string type(reinterpret_cast<const char*>(&d[o + 4]), 4);
        if (o + 12 + len > d.size()) return false;
        const unsigned char* p = &d[o + 8];
        if (type == "IHDR") { w = be32(o + 8); h = be32(o + 12); depth = p[8]; ctype = p[9]; interlace = p[12]; }
        else if (type == "PLTE") palette.assign(p, p + len);
        else if (type == "tRNS") trns.assign(p, p + len);
        else if (type == "IDAT") idat.insert(idat.end(), p, p + len);
        else if (type == "IEND") break;
        o += 12 + len;
    }
    if (!w || !h || interlace || (depth != 8 && depth != 16 && !(ctype == 3 || ctype == 0))) return false;
    const int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples) return false;
    const size_t bpp = std::max<size_t>(1, (size_t)samples * depth / 8), stride = ((size_t)w * samples * depth + 7) / 8;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf rawLen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) return false;
    // undo the scanline filters
    std::vector<unsigned char> pix(stride * h);
    for (uint32_t y = 0; y < h; y++) {
        const unsigned char* in = &raw[(stride + 1) * y]; unsigned char* out = &pix[stride * y];
        const unsigned char* up = y ? &pix[stride * (y - 1)] : nullptr;
        const int ft = in[0];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) return false;
            out[x] = (unsigned char)(in[x + 1] + pred);
        }
    }
    // to 8-bit samples, channel counts as the usual decoders report them (grey 1, grey+alpha 2, RGB 3, RGBA 4)
    const bool pal = ctype == 3;
    const int ch = pal ? (trns.empty() ? 3 : 4) : samples;
    img.width = (int)w; img.height = (int)h; img.channels = ch; img.isFloat = false; img.u8.resize((size_t)w * h * ch);
    for (uint32_t y = 0; y < h; y++) for (uint32_t x = 0; x < w; x++) {
        unsigned char* o = &img.u8[((size_t)y * w + x) * ch]; const unsigned char* row = &pix[stride * y];
        if (pal || (ctype == 0 && depth < 8)) {
            const int per = 8 / depth, shift = (per - 1 - (int)(x % per)) * depth; const int v = (row[x / per] >> shift) & ((1 << depth) - 1);
            if (pal) { if ((size_t)v * 3 + 2 >= palette.size()) return false; o[0] = palette[v * 3]; o[1] = palette[v * 3 + 1]; o[2] = palette[v * 3 + 2]; if (ch == 4) o[3] = (size_t)v < trns.size() ? trns[v] : 255; }
            else o[0] = (unsigned char)(v * 255 / ((1 << depth) - 1));
        } else for (int c = 0; c < samples; c++) o[c] = depth == 8 ? row[(size_t)x * samples + c] : row[((size_t)x * samples + c) * 2];   // 16-bit: high byte
    }
    return true;
}

// Texture::load, RTBase/Imaging.h:39-78
inline void loadTexture(const std::string& filename, Texture& t)
{
    Image img;
    if (filename.find(".hdr") != std::string::npos) {
        if (!readHdr(filename, img)) { std::cerr << "warning: cannot read " << filename << ", using the 1x1 white default\n"; t.loadDefault(); return; }
        t.width = img.width; t.height = img.height; t.texels.resize((size_t)img.width * img.height);
        for (size_t i = 0; i < t.texels.size(); i++) t.texels[i] = Colour(img.f[i * 3], img.f[i * 3 + 1], img.f[i * 3 + 2]);
        return;
    }
    if (!readPng(filename, img)) { std::cerr << "warning: cannot read " << filename << " (PNG and Radiance HDR are decoded here), using the 1x1 white default\n"; t.loadDefault(); return; }
    const int c = img.channels; const size_t n = (size_t)img.width * img.height, last = img.u8.size() - 1;
    t.width = img.width; t.height = img.height; t.texels.resize(n);
    // the reference reads bytes i*channels, +1, +2 whatever the channel count is (:67)
    for (size_t i = 0; i < n; i++)
        t.texels[i] = Colour(img.u8[std::min(i * c, last)] / 255.0f, img.u8[std::min(i * c + 1, last)] / 255.0f, img.u8[std::min(i * c + 2, last)] / 255.0f);
}

// ---- the assembled scene -----------------------------------------------------------------------------
struct SceneData {
    Vec3 from, to, up; float fov = 45.0f; int width = 1920, height = 1080, flipX = 0;
    std::vector<Texture*> textures; std::vector<BSDF*> materials; std::vector<Triangle> triangles;
    Texture* env = nullptr; Colour background;
    std::map<std::string, Texture*> textureManager;
    std::vector<BSDF*> unusedCoatings;      // LayeredBSDF objects the loader creates and the scene never references
};

inline Texture* texture(SceneData& sd, const std::string& filename)      // loadTexture, RTBase/SceneLoader.h:193-203
{
    auto it = sd.textureManager.find(filename);
    if (it != sd.textureManager.end()) return it->second;
    Texture* t = new Texture(); loadTexture(filename, *t);
    sd.textureManager[filename] = t; sd.textures.push_back(t);
    return t;
}

inline void loadInstance(const std::string& dir, SceneData& sd, const std::string& meshFile, const float world[16], const Properties& mp)
{
    std::vector<Mesh> meshes = readModel(dir + "/" + meshFile);
    BSDF* material = nullptr;
    const std::string bsdf = mp.text("bsdf");
    auto refl = [&]() { return texture(sd, dir + "/" + mp.text("reflectance")); };
    if (bsdf == "diffuse") { material = new DiffuseBSDF(refl()); sd.materials.push_back(material); }
    if (bsdf == "orennayar") { material = new OrenNayarBSDF(refl(), mp.number("alpha", 1.0f)); sd.materials.push_back(material); }
    if (bsdf == "glass") { material = new GlassBSDF(refl(), mp.number("intIOR", 1.33f), mp.number("extIOR", 1.0f)); sd.materials.push_back(material); }
    if (bsdf == "mirror") { material = new MirrorBSDF(refl()); sd.materials.push_back(material); }
    if (bsdf == "plastic") { material = new PlasticBSDF(refl(), mp.number("intIOR", 1.33f), mp.number("extIOR", 1.0f), mp.number("roughness", 1.0f)); sd.materials.push_back(material); }
    if (bsdf == "dielectric") {
        const float i = mp.number("intIOR", 1.33f), e = mp.number("extIOR", 1.0f), r = mp.number("roughness", 1.0f);
        if (r < 0.001f) material = new GlassBSDF(refl(), i, e); else material = new DielectricBSDF(refl(), i, e, r);
        sd.materials.push_back(material);
    }
    if (bsdf == "conductor") {
        Colour eta, k; mp.vector3("eta", eta.r, eta.g, eta.b); mp.vector3("k", k.r, k.g, k.b);
        material = new ConductorBSDF(refl(), eta, k, mp.number("roughness", 1.0f)); sd.materials.push_back(material);
    }
    if (mp.text("emission") != "") {
        if (!material) throw std::runtime_error(meshFile + ": emission on an instance without a known bsdf");
        Colour e; mp.vector3("emission", e.r, e.g, e.b); material->addLight(e);
    }
    if (mp.integer("coatingThickness", 0) > 0) {
        if (!material) throw std::runtime_error(meshFile + ": coating on an instance without a known bsdf");
        Colour sa; mp.vector3("coatingSigmaA", sa.r, sa.g, sa.b);
        sd.unusedCoatings.push_back(new LayeredBSDF(material, sa, mp.number("coatingThickness", 0.0f), mp.number("coatingIntIOR", 1.33f), mp.number("coatingExtIOR", 1.0f)));
    }
    if (!material) { std::cout << "Error in loading" << std::endl; return; }
    const unsigned materialIndex = (unsigned)sd.materials.size() - 1;
    std::vector<Vertex> vertices; std::vector<unsigned> indices;
    for (const Mesh& m : meshes) {
        const size_t base = vertices.size(); std::vector<float> p(m.vertices.size() * 3), n(m.vertices.size() * 3);
        for (size_t i = 0; i < m.vertices.size(); i++) for (int k = 0; k < 3; k++) { p[i * 3 + k] = m.vertices[i].position[k]; n[i * 3 + k] = m.vertices[i].normal[k]; }
        rtxCheck(rtx_host_transform_instance(world, m.vertices.size(), p.data(), n.data()), "loadInstance");
        for (size_t i = 0; i < m.vertices.size(); i++) {
            Vertex v; v.p = Vec3(p[i * 3], p[i * 3 + 1], p[i * 3 + 2]); v.normal = Vec3(n[i * 3], n[i * 3 + 1], n[i * 3 + 2]); v.u = m.vertices[i].u; v.v = m.vertices[i].v;
            vertices.push_back(v);
        }
        (void)base;
        const unsigned offset = (unsigned)indices.size();     // the reference offsets by the INDEX count so far (:321-325)
        for (uint32_t ix : m.indices) indices.push_back(offset + ix);
    }
    for (size_t i = 0; i + 2 < indices.size(); i += 3) {
        if (indices[i] >= vertices.size() || indices[i + 1] >= vertices.size() || indices[i + 2] >= vertices.size()) throw std::runtime_error(meshFile + ": vertex index out of range");
        Triangle t; t.init(vertices[indices[i]], vertices[indices[i + 1]], vertices[indices[i + 2]], materialIndex);
        if (t.area > 0) sd.triangles.push_back(t);
    }
}

// loadScene, RTBase/SceneLoader.h:337-389, up to (not including) Scene::init / build
inline void loadGemScene(const std::string& dir, SceneData& sd)
{
    std::ifstream f(dir + "/scene.json");
    if (!f) throw std::runtime_error("cannot open " + dir + "/scene.json");
    std::stringstream buf; buf << f.rdbuf();
    const std::string content = buf.str();
    Json root = JsonReader(content).parse();
    if (root.kind != Json::Obj) throw std::runtime_error("scene.json: top level must be an object");
    Properties sp;
    for (auto& item : root.obj) if (item.second.kind != Json::Arr) sp.items.emplace_back(item.first, item.second.text());
    sd.width = sp.integer("width", 1920); sd.height = sp.integer("height", 1080); sd.fov = sp.number("fov", 45.0f);
    sp.vector3("from", sd.from.x, sd.from.y, sd.from.z); sp.vector3("to", sd.to.x, sd.to.y, sd.to.z); sp.vector3("up", sd.up.x, sd.up.y, sd.up.z);
    sd.flipX = sp.integer("flipX", 0) == 1 ? 1 : 0;
    for (auto& item : root.obj) {
        if (item.second.kind != Json::Arr) continue;
        for (const Json& inst : item.second.arr) {
            std::string meshFile; float world[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; Properties mp;
            for (auto& kv : inst.obj) {
                if (kv.first == "filename") meshFile = kv.second.text();
                else if (kv.first == "world") { if (kv.second.arr.size() < 16) throw std::runtime_error("scene.json: world needs 16 numbers"); for (int i = 0; i < 16; i++) world[i] = kv.second.arr[i].num; }
                else mp.items.emplace_back(kv.first, kv.second.text());
            }
            loadInstance(dir, sd, meshFile, world, mp);
        }
    }
    if (sp.text("envmap") != "") sd.env = texture(sd, dir + "/" + sp.text("envmap"));
    else sd.background = Colour(0.0f, 0.0f, 0.0f);
}

// What this reader sees in a scene directory, condensed into eight numbers (FNV-1a hashes and
// counts) for comparison with the same digest taken through the reference's own GEMLoader.h:
// [0] scene properties, [1] hash of their (name, value) in iteration order, [2] instances,
// [3] hash of (mesh file, 16 world floats, material properties), [4] vertices, [5] indices,
// [6] hash of the 44-byte vertex records and the indices, [7] hash of the mesh-level properties.
inline void digest(const std::string& dir, uint64_t out[8])
{
    struct Fnv {
        uint64_t h = 1469598103934665603ull;
        void bytes(const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } }
        void str(const std::string& s) { uint32_t n = (uint32_t)s.size(); bytes(&n, 4); bytes(s.data(), s.size()); }
    } props, inst, geo, mprops;
    std::ifstream f(dir + "/scene.json");
    if (!f) throw std::runtime_error("cannot open " + dir + "/scene.json");
    std::stringstream buf; buf << f.rdbuf();
    const std::string content = buf.str();
    Json root = JsonReader(content).parse();
    uint64_t nprops = 0, ninst = 0, nv = 0, ni = 0;
    for (auto& item : root.obj) if (item.second.kind != Json::Arr) { props.str(item.first); props.str(item.second.text()); nprops++; }
    for (auto& item : root.obj) {
        if (item.second.kind != Json::Arr) continue;
        for (const Json& in : item.second.arr) {
            std::string meshFile; float world[16] = {0}; Properties mp;
            for (auto& kv : in.obj) {
                if (kv.first == "filename") meshFile = kv.second.text();
                else if (kv.first == "world") { for (int i = 0; i < 16 && i < (int)kv.second.arr.size(); i++) world[i] = kv.second.arr[i].num; }
                else mp.items.emplace_back(kv.first, kv.second.text());
            }
            inst.str(meshFile); inst.bytes(world, 64);
            for (auto& p : mp.items) { inst.str(p.first); inst.str(p.second); }
            ninst++;
            for (const Mesh& m : readModel(dir + "/" + meshFile)) {
                for (auto& p : m.material.items) { mprops.str(p.first); mprops.str(p.second); }
                nv += m.vertices.size(); ni += m.indices.size();
                if (!m.vertices.empty()) geo.bytes(m.vertices.data(), m.vertices.size() * sizeof(StaticVertex));
                if (!m.indices.empty()) geo.bytes(m.indices.data(), m.indices.size() * 4);
            }
        }
    }
    out[0] = nprops; out[1] = props.h; out[2] = ninst; out[3] = inst.h; out[4] = nv; out[5] = ni; out[6] = geo.h; out[7] = mprops.h;
}

} // namespace gem

// loadScene, RTBase/SceneLoader.h:337-389, with the reference's signature (plus the device to
// render on): reads the scene directory, sets up the camera and the viewer's fly-through
// camera, builds the scene on the device. The caller deletes the scene (RTBase/Main.cpp:226).
inline Scene* loadScene(std::string sceneName, RTCamera& viewcamera, int device = 0, gem::SceneData* keep = nullptr)
{
    gem::SceneData* sd = keep ? keep : new gem::SceneData();      // textures and BSDFs live as long as the scene (the reference leaks them too)
    gem::loadGemScene(sceneName, *sd);
    Scene* scene = new Scene(device);
    scene->camera.lookAt(sd->from, sd->to, sd->up, sd->fov, sd->width, sd->height, sd->flipX);
    viewcamera.from = sd->from; viewcamera.to = sd->to; viewcamera.up = sd->up; viewcamera.camera = &scene->camera;
    Background bg; bg.colour = sd->background; bg.env = sd->env;
    scene->init(sd->triangles, sd->materials, bg);
    Vec3 mn(FLT_MAX, FLT_MAX, FLT_MAX), mx(-FLT_MAX, -FLT_MAX, -FLT_MAX);      // Scene::init's bounds, Scene.h:151-160
    for (const Triangle& t : sd->triangles) for (const Vertex& v : t.vertices) {
        mn = Vec3(std::min(mn.x, v.p.x), std::min(mn.y, v.p.y), std::min(mn.z, v.p.z));
        mx = Vec3(std::max(mx.x, v.p.x), std::max(mx.y, v.p.y), std::max(mx.z, v.p.z));
    }
    viewcamera.movespeed = (mx - mn).length() * 0.05f;             // :385
    scene->build();
    return scene;
}
