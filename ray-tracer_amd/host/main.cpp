// main.cpp — headless Linux driver with the call sequence of the reference's interactive
// main (RTBase/Main.cpp:71-228): create settings, load the scene, init the ray tracer, one
// renderMT() per progressive pass with the five console statistics, save the HDR when
// totalSPP is reached. Flags of the reference: -scene, -outputFilename, -SPP
// (RTBase/Main.cpp:104-115); added: -alg pt|lt|ir, -bounces N, -mis 0|1, -adaptive 0|1,
// -seed N, -png file, -dumpFilm file (raw float sums), -device N, -quiet 1.
//
// Scenes are ".rtxs" dumps (see SceneArrays.save in the Python package): the reference's own
// GEM assets are not available offline.
#include <cfloat>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <unordered_map>

#include "RayTracerHost.h"
#include "png_writer.h"

SETTINGS createSettings()   // RTBase/Main.cpp:11-36, denoise off (no OIDN on this platform)
{
    SETTINGS s;
    s.algorithm = AL_PATH_TRACE; s.drawMode = DM_ALGORITHM; s.toneMap = TM_LINEAR; s.filter = FT_BOX;
    s.canHitLight = true; s.debug = false; s.denoise = false;
    s.useMultithreading = true; s.useMis = true;
    s.adaptiveSampling = true; s.initSPP = 10; s.totalSPP = 500;
    s.numThreads = 20; s.maxBounces = 5; s.vplRaysPerTile = 1;
    return s;
}

template <class T> static void rd(std::ifstream& f, T* p, size_t n) { f.read(reinterpret_cast<char*>(p), (std::streamsize)(n * sizeof(T))); if (!f) throw std::runtime_error("scene file truncated"); }

struct Loaded { std::vector<Texture*> textures; Texture* env = nullptr; };

// .rtxs: magic "RTXS", u32 version, camera (from, to, up, fov: 10 floats, width, height: 2 u32),
// u32 nTex {u32 w, h, float rgb[w*h*3]}, u32 nMat {rtx_material}, u64 nTri,
// positions, normals, uvs, material index, u32 envW, envH, env rgb, background[3]
static Scene* loadScene(const std::string& path, int device, Loaded& keep)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open scene " + path);
    char magic[4]; uint32_t version; rd(f, magic, 4); rd(f, &version, 1);
    if (std::memcmp(magic, "RTXS", 4) != 0 || version != 1) throw std::runtime_error("not an .rtxs scene");
    float cam[10]; uint32_t wh[2]; rd(f, cam, 10); rd(f, wh, 2);
    Scene* scene = new Scene(device);
    scene->camera.lookAt(Vec3(cam[0], cam[1], cam[2]), Vec3(cam[3], cam[4], cam[5]), Vec3(cam[6], cam[7], cam[8]), cam[9], (int)wh[0], (int)wh[1]);
    uint32_t nTex; rd(f, &nTex, 1);
    for (uint32_t i = 0; i < nTex; i++) {
        uint32_t d[2]; rd(f, d, 2); std::vector<float> px((size_t)d[0] * d[1] * 3); rd(f, px.data(), px.size());
        Texture* t = new Texture(); t->fromPixels((int)d[0], (int)d[1], px.data()); keep.textures.push_back(t);
    }
    uint32_t nMat; rd(f, &nMat, 1);
    std::vector<BSDF*> mats;
    for (uint32_t i = 0; i < nMat; i++) {
        rtx_material m; rd(f, &m, 1);
        Texture* t = keep.textures.at((size_t)m.texture); BSDF* b = nullptr;
        switch (m.type) {   // the factory of RTBase/SceneLoader.h:212-272
        case RTX_BSDF_DIFFUSE: b = new DiffuseBSDF(t); break;
        case RTX_BSDF_MIRROR: b = new MirrorBSDF(t); break;
        case RTX_BSDF_CONDUCTOR: b = new ConductorBSDF(t, Colour(m.eta[0], m.eta[1], m.eta[2]), Colour(m.k[0], m.k[1], m.k[2]), m.roughness); break;
        case RTX_BSDF_GLASS: b = new GlassBSDF(t, m.int_ior, m.ext_ior); break;
        case RTX_BSDF_PLASTIC: b = new PlasticBSDF(t, m.int_ior, m.ext_ior, m.roughness); break;
        default: throw std::runtime_error("unknown bsdf in scene");
        }
        if (m.emission[0] != 0 || m.emission[1] != 0 || m.emission[2] != 0) b->addLight(Colour(m.emission[0], m.emission[1], m.emission[2]));
        mats.push_back(b);
    }
    uint64_t nTri; rd(f, &nTri, 1);
    std::vector<float> pos(nTri * 9), nrm(nTri * 9), uv(nTri * 6); std::vector<uint32_t> mi(nTri);
    rd(f, pos.data(), pos.size()); rd(f, nrm.data(), nrm.size()); rd(f, uv.data(), uv.size()); rd(f, mi.data(), mi.size());
    std::vector<Triangle> tris;
    for (uint64_t i = 0; i < nTri; i++) {
        Vertex v[3];
        for (int k = 0; k < 3; k++) {
            v[k].p = Vec3(pos[i * 9 + k * 3], pos[i * 9 + k * 3 + 1], pos[i * 9 + k * 3 + 2]);
            v[k].normal = Vec3(nrm[i * 9 + k * 3], nrm[i * 9 + k * 3 + 1], nrm[i * 9 + k * 3 + 2]);
            v[k].u = uv[i * 6 + k * 2]; v[k].v = uv[i * 6 + k * 2 + 1];
        }
        Triangle t; t.init(v[0], v[1], v[2], mi[i]);
        if (t.area > 0) tris.push_back(t);   // RTBase/SceneLoader.h:330
    }
    uint32_t env[2]; rd(f, env, 2);
    Background bg;
    if (env[0] && env[1]) { std::vector<float> px((size_t)env[0] * env[1] * 3); rd(f, px.data(), px.size()); keep.env = new Texture(); keep.env->fromPixels((int)env[0], (int)env[1], px.data()); bg.env = keep.env; }
    float b3[3]; rd(f, b3, 3); bg.colour = Colour(b3[0], b3[1], b3[2]);
    scene->init(tris, mats, bg);
    scene->build();
    return scene;
}

int main(int argc, char* argv[])
{
    SETTINGS settings = createSettings();
    std::string sceneName, filename = "render_GI.hdr", png, dump; int device = 0; bool quiet = false;
    std::unordered_map<std::string, std::string> args;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (!a.empty() && a[0] == '-') { if (i + 1 < argc) args[a] = argv[++i]; else std::cerr << "Error: Missing value for argument '" << a << "'\n"; }
        else std::cerr << "Warning: Ignoring unexpected argument '" << a << "'\n";
    }
    for (const auto& p : args) {
        if (p.first == "-scene") sceneName = p.second;
        else if (p.first == "-outputFilename") filename = p.second;
        else if (p.first == "-SPP") settings.totalSPP = (unsigned)std::stoi(p.second);
        else if (p.first == "-alg") settings.algorithm = p.second == "lt" ? AL_LIGHT_TRACE : p.second == "ir" ? AL_INSTANT_RADIOSITY : AL_PATH_TRACE;
        else if (p.first == "-bounces") settings.maxBounces = (unsigned)std::stoi(p.second);
        else if (p.first == "-mis") settings.useMis = std::stoi(p.second) != 0;
        else if (p.first == "-adaptive") settings.adaptiveSampling = std::stoi(p.second) != 0;
        else if (p.first == "-initSPP") settings.initSPP = (unsigned)std::stoi(p.second);
        else if (p.first == "-seed") settings.seed = (unsigned)std::stoul(p.second);
        else if (p.first == "-png") png = p.second;
        else if (p.first == "-dumpFilm") dump = p.second;
        else if (p.first == "-device") device = std::stoi(p.second);
        else if (p.first == "-quiet") quiet = std::stoi(p.second) != 0;
        else std::cerr << "Warning: unknown flag '" << p.first << "'\n";
    }
    if (sceneName.empty()) { std::cerr << "usage: rt_headless -scene file.rtxs [-outputFilename out.hdr] [-SPP n] ...\n"; return 2; }
    try {
        Loaded keep;
        std::cout << "Loading scene: " << sceneName << std::endl;
        Scene* scene = loadScene(sceneName, device, keep);
        GamesEngineeringBase::Window canvas;
        canvas.create((unsigned int)scene->camera.width, (unsigned int)scene->camera.height, "Tracer", 1.0f);
        RayTracer rt;
        rt.init(scene, &canvas, settings);
        GamesEngineeringBase::Timer timer; float totalTime = 0;
        bool completed = false;
        while (!completed) {   // the body of the reference's frame loop, RTBase/Main.cpp:166-215
            timer.reset();
            if (settings.useMultithreading) rt.renderMT(); else rt.render();
            float t = timer.dt();
            totalTime += t;
            float progress = rt.getSPP() * 100 / settings.totalSPP;
            if (!quiet) {
                std::cout << "Progress   : " << progress << "%\n" << "Samples    : " << rt.getSPP() << "\n" << "Time       : " << t << "\n"
                          << "FPS        : " << (t > 0 ? 1.0f / t : FLT_MAX) << "\n" << "Total time : " << std::roundf(totalTime) << " sec\n";
            }
            if ((int)settings.totalSPP <= rt.getSPP() && settings.drawMode == DM_ALGORITHM) { completed = true; rt.saveHDR(filename); }
        }
        rt.draw();
        if (!png.empty()) rt.savePNG(png);
        if (!dump.empty()) {
            std::ofstream o(dump, std::ios::binary);
            uint32_t hdr[3] = {rt.film->width, rt.film->height, (uint32_t)rt.film->SPP};
            o.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
            o.write(reinterpret_cast<const char*>(rt.film->film), (std::streamsize)((size_t)rt.film->width * rt.film->height * sizeof(Colour)));
            o.write(reinterpret_cast<const char*>(rt.tileSamples.data()), (std::streamsize)(rt.tileSamples.size() * 4));
        }
        std::cout << "Done: " << rt.getSPP() << " spp in " << totalTime << " s (" << rt.getSPP() / totalTime << " spp/s)" << std::endl;
        delete scene;
    } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
    return 0;
}
