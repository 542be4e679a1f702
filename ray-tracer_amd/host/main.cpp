// main.cpp — headless Linux driver with the call sequence of the reference's interactive
// main (RTBase/Main.cpp:71-228): create settings, load the scene, init the ray tracer, one
// renderMT() per progressive pass with the five console statistics, save the HDR when
// totalSPP is reached. Flags of the reference: -scene, -outputFilename, -SPP
// (RTBase/Main.cpp:104-115); added: -alg pt|lt|ir, -bounces N, -mis 0|1, -adaptive 0|1,
// -seed N, -png file, -dumpFilm file (raw float sums), -dumpScene file, -digestScene dir, -digestImage file, -orbit degrees (yaw the view first), -batch N (passes per renderMT call), -device N, -quiet 1,
// -gpus N -shard tiles|spp: one process per GPU (forked before anything touches a GPU; GPU r renders tiles or samples r mod N),
// films combined by an RCCL all-reduce through the core's C ABI; rank 0 writes the outputs.
//
// -scene takes what the reference takes, a directory with scene.json + .gem meshes + textures
// (GemScene.h), or an ".rtxs" dump (see SceneArrays.save in the Python package). -dumpScene file
// writes the loaded scene as an .rtxs dump and exits without touching a GPU.
#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <unordered_map>

#include <sys/wait.h>
#include <signal.h>
#include <unistd.h>

#include "RayTracerHost.h"
#include "GemScene.h"
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
template <class T> static void wr(std::ofstream& f, const T* p, size_t n) { f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T))); }

// .rtxs: magic "RTXS", u32 version (1, or 2 = a u32 flipX follows the film size), camera (from, to, up,
// fov: 10 floats, width, height: 2 u32), u32 nTex {u32 w, h, float rgb[w*h*3]}, u32 nMat {rtx_material},
// u64 nTri, positions, normals, uvs, material index, u32 envW, envH, env rgb, background[3]
static void loadRtxs(const std::string& path, gem::SceneData& sd)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open scene " + path);
    char magic[4]; uint32_t version; rd(f, magic, 4); rd(f, &version, 1);
    if (std::memcmp(magic, "RTXS", 4) != 0 || (version != 1 && version != 2)) throw std::runtime_error("not an .rtxs scene");
    float cam[10]; uint32_t wh[2]; rd(f, cam, 10); rd(f, wh, 2);
    if (version == 2) { uint32_t fx; rd(f, &fx, 1); sd.flipX = fx ? 1 : 0; }
    sd.from = Vec3(cam[0], cam[1], cam[2]); sd.to = Vec3(cam[3], cam[4], cam[5]); sd.up = Vec3(cam[6], cam[7], cam[8]); sd.fov = cam[9];
    sd.width = (int)wh[0]; sd.height = (int)wh[1];
    uint32_t nTex; rd(f, &nTex, 1);
    for (uint32_t i = 0; i < nTex; i++) {
        uint32_t d[2]; rd(f, d, 2); std::vector<float> px((size_t)d[0] * d[1] * 3); rd(f, px.data(), px.size());
        Texture* t = new Texture(); t->fromPixels((int)d[0], (int)d[1], px.data()); sd.textures.push_back(t);
    }
    uint32_t nMat; rd(f, &nMat, 1);
    for (uint32_t i = 0; i < nMat; i++) {
        rtx_material m; rd(f, &m, 1);
        Texture* t = sd.textures.at((size_t)m.texture); BSDF* b = nullptr;
        switch (m.type & ~RTX_BSDF_LAYERED) {   // the factory of RTBase/SceneLoader.h:212-272
        case RTX_BSDF_DIFFUSE: b = new DiffuseBSDF(t); break;
        case RTX_BSDF_MIRROR: b = new MirrorBSDF(t); break;
        case RTX_BSDF_CONDUCTOR: b = new ConductorBSDF(t, Colour(m.eta[0], m.eta[1], m.eta[2]), Colour(m.k[0], m.k[1], m.k[2]), m.roughness); break;
        case RTX_BSDF_GLASS: b = new GlassBSDF(t, m.int_ior, m.ext_ior); break;
        case RTX_BSDF_PLASTIC: b = new PlasticBSDF(t, m.int_ior, m.ext_ior, m.roughness); break;
        case RTX_BSDF_DIELECTRIC: b = new DielectricBSDF(t, m.int_ior, m.ext_ior, m.roughness); break;
        case RTX_BSDF_ORENNAYAR: b = new OrenNayarBSDF(t, m.roughness); break;
        default: throw std::runtime_error("unknown bsdf in scene");
        }
        if (m.type & RTX_BSDF_LAYERED) b = new LayeredBSDF(b, Colour(0, 0, 0), 0.0f, 1.33f, 1.0f);
        if (m.emission[0] != 0 || m.emission[1] != 0 || m.emission[2] != 0) b->addLight(Colour(m.emission[0], m.emission[1], m.emission[2]));
        sd.materials.push_back(b);
    }
    uint64_t nTri; rd(f, &nTri, 1);
    std::vector<float> pos(nTri * 9), nrm(nTri * 9), uv(nTri * 6); std::vector<uint32_t> mi(nTri);
    rd(f, pos.data(), pos.size()); rd(f, nrm.data(), nrm.size()); rd(f, uv.data(), uv.size()); rd(f, mi.data(), mi.size());
    for (uint64_t i = 0; i < nTri; i++) {
        Vertex v[3];
        for (int k = 0; k < 3; k++) {
            v[k].p = Vec3(pos[i * 9 + k * 3], pos[i * 9 + k * 3 + 1], pos[i * 9 + k * 3 + 2]);
            v[k].normal = Vec3(nrm[i * 9 + k * 3], nrm[i * 9 + k * 3 + 1], nrm[i * 9 + k * 3 + 2]);
            v[k].u = uv[i * 6 + k * 2]; v[k].v = uv[i * 6 + k * 2 + 1];
        }
        Triangle t; t.init(v[0], v[1], v[2], mi[i]);
        if (t.area > 0) sd.triangles.push_back(t);   // RTBase/SceneLoader.h:330
    }
    uint32_t env[2]; rd(f, env, 2);
    if (env[0] && env[1]) { std::vector<float> px((size_t)env[0] * env[1] * 3); rd(f, px.data(), px.size()); sd.env = new Texture(); sd.env->fromPixels((int)env[0], (int)env[1], px.data()); }
    float b3[3]; rd(f, b3, 3); sd.background = Colour(b3[0], b3[1], b3[2]);
}

// What Scene::build would upload, as an .rtxs file: lets a loaded GEM scene be inspected (and
// re-rendered) without the scene directory, and needs no GPU.
static void dumpRtxs(const std::string& path, const gem::SceneData& sd)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write " + path);
    const uint32_t version = 2; f.write("RTXS", 4); wr(f, &version, 1);
    const float cam[10] = {sd.from.x, sd.from.y, sd.from.z, sd.to.x, sd.to.y, sd.to.z, sd.up.x, sd.up.y, sd.up.z, sd.fov};
    const uint32_t wh[2] = {(uint32_t)sd.width, (uint32_t)sd.height}, fx = (uint32_t)sd.flipX;
    wr(f, cam, 10); wr(f, wh, 2); wr(f, &fx, 1);
    std::vector<Texture*> texList; std::vector<rtx_material> mats;
    for (BSDF* b : sd.materials) {
        rtx_material m = b->describe();
        size_t t = 0; while (t < texList.size() && texList[t] != b->albedo) t++;
        if (t == texList.size()) texList.push_back(b->albedo);
        m.texture = (int32_t)t; mats.push_back(m);
    }
    const uint32_t nTex = (uint32_t)texList.size(); wr(f, &nTex, 1);
    for (Texture* t : texList) { const uint32_t d[2] = {(uint32_t)t->width, (uint32_t)t->height}; wr(f, d, 2); wr(f, &t->texels[0].r, t->texels.size() * 3); }
    const uint32_t nMat = (uint32_t)mats.size(); wr(f, &nMat, 1); wr(f, mats.data(), mats.size());
    const uint64_t nTri = sd.triangles.size(); wr(f, &nTri, 1);
    for (int what = 0; what < 4; what++) for (const Triangle& t : sd.triangles) {
        if (what == 3) { const uint32_t m = t.materialIndex; wr(f, &m, 1); continue; }
        for (int k = 0; k < 3; k++) {
            const Vertex& v = t.vertices[k];
            if (what == 0) { const float p[3] = {v.p.x, v.p.y, v.p.z}; wr(f, p, 3); }
            if (what == 1) { const float n[3] = {v.normal.x, v.normal.y, v.normal.z}; wr(f, n, 3); }
            if (what == 2) { const float u[2] = {v.u, v.v}; wr(f, u, 2); }
        }
    }
    const uint32_t env[2] = {sd.env ? (uint32_t)sd.env->width : 0u, sd.env ? (uint32_t)sd.env->height : 0u}; wr(f, env, 2);
    if (sd.env) wr(f, &sd.env->texels[0].r, sd.env->texels.size() * 3);
    const float b3[3] = {sd.background.r, sd.background.g, sd.background.b}; wr(f, b3, 3);
}

// loadScene, RTBase/SceneLoader.h:337-389: a directory with scene.json is a GEM scene, anything else an .rtxs dump
static void loadSceneData(const std::string& path, gem::SceneData& sd)
{
    std::ifstream probe(path + "/scene.json");
    if (probe) gem::loadGemScene(path, sd); else loadRtxs(path, sd);
}

static Scene* buildScene(const gem::SceneData& sd, int device)
{
    Scene* scene = new Scene(device);
    scene->camera.lookAt(sd.from, sd.to, sd.up, sd.fov, sd.width, sd.height, sd.flipX);
    Background bg; bg.colour = sd.background; bg.env = sd.env;
    scene->init(sd.triangles, sd.materials, bg);
    scene->build();
    return scene;
}

int main(int argc, char* argv[])
{
    SETTINGS settings = createSettings();
    std::string sceneName, filename = "render_GI.hdr", png, dump, dumpScene, digestDir, digestImage, shard = "tiles"; int device = 0, batch = 1, gpus = 1; bool quiet = false; float orbitDeg = 0.0f;
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
        else if (p.first == "-dumpScene") dumpScene = p.second;
        else if (p.first == "-digestScene") digestDir = p.second;
        else if (p.first == "-digestImage") digestImage = p.second;
        else if (p.first == "-device") device = std::stoi(p.second);
        else if (p.first == "-quiet") quiet = std::stoi(p.second) != 0;
        else if (p.first == "-orbit") orbitDeg = std::stof(p.second);
        else if (p.first == "-batch") batch = std::max(1, std::stoi(p.second));
        else if (p.first == "-gpus") gpus = std::max(1, std::stoi(p.second));
        else if (p.first == "-shard") shard = p.second;
        else std::cerr << "Warning: unknown flag '" << p.first << "'\n";
    }
    if (!digestImage.empty()) {   // width, height, channels and FNV-1a hash of the decoded pixels of one image file
        gem::Image img; const bool hdr = digestImage.find(".hdr") != std::string::npos;
        if (!(hdr ? gem::readHdr(digestImage, img) : gem::readPng(digestImage, img))) { std::cerr << "error: cannot decode " << digestImage << std::endl; return 1; }
        uint64_t h = 1469598103934665603ull;
        const unsigned char* b = hdr ? (const unsigned char*)img.f.data() : img.u8.data();
        const size_t nb = hdr ? img.f.size() * sizeof(float) : img.u8.size();
        for (size_t i = 0; i < nb; i++) { h ^= b[i]; h *= 1099511628211ull; }
        std::cout << img.width << " " << img.height << " " << img.channels << " " << h << "\n";
        return 0;
    }
    if (!digestDir.empty()) {   // what the scene-directory reader sees, as eight numbers (see gem::digest); no GPU involved
        try { uint64_t d[8]; gem::digest(digestDir, d); for (int i = 0; i < 8; i++) std::cout << d[i] << (i < 7 ? " " : "\n"); return 0; }
        catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
    }
    if (sceneName.empty()) { std::cerr << "usage: rt_headless -scene <scene directory | file.rtxs> [-outputFilename out.hdr] [-SPP n] ...\n"; return 2; }
    if (shard != "tiles" && shard != "spp") { std::cerr << "error: -shard takes tiles or spp\n"; return 2; }
    // -gpus N: one process per GPU. The ranks are forked here, before this process has made any GPU call; rank 0 creates the
    // RCCL id once its own GPU is up and sends it to the others through pipes that exist before the fork.
    int rank = 0; std::vector<int> idPipes;        // rank 0: write ends towards ranks 1..N-1; other ranks: their read end
    if (gpus > 1) {
        std::vector<std::array<int, 2>> pipes((size_t)gpus);
        for (int r = 1; r < gpus; r++) if (pipe(pipes[(size_t)r].data()) != 0) { std::perror("pipe"); return 1; }
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < gpus && !child; r++) {
            pid_t pid = fork();
            if (pid < 0) { std::perror("fork"); return 1; }
            if (pid == 0) { child = true; rank = r; } else kids.push_back(pid);
        }
        if (!child) {
            for (int r = 1; r < gpus; r++) { close(pipes[(size_t)r][0]); close(pipes[(size_t)r][1]); }
            // whichever rank ends first: if it failed, the others would wait for it in a collective for ever, so they are ended too
            int worst = 0; size_t left = kids.size();
            while (left > 0) {
                int st = 0; const pid_t done = waitpid(-1, &st, 0);
                if (done < 0) break;
                left--;
                const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 1;
                if (code != 0 && worst == 0) { worst = code; for (pid_t k : kids) if (k != done) kill(k, SIGTERM); }
            }
            return worst;
        }
        for (int r = 1; r < gpus; r++) {
            if (rank == 0) { close(pipes[(size_t)r][0]); idPipes.push_back(pipes[(size_t)r][1]); }
            else if (r == rank) { close(pipes[(size_t)r][1]); idPipes.push_back(pipes[(size_t)r][0]); }
            else { close(pipes[(size_t)r][0]); close(pipes[(size_t)r][1]); }
        }
        device += rank;
        if (rank != 0) quiet = true;
    }
    try {
        gem::SceneData data;
        std::cout << "Loading scene: " << sceneName << std::endl;
        loadSceneData(sceneName, data);
        if (!dumpScene.empty()) { dumpRtxs(dumpScene, data); std::cout << "Wrote " << dumpScene << " (" << data.triangles.size() << " triangles, " << data.materials.size() << " materials)" << std::endl; return 0; }
        Scene* scene = buildScene(data, device);
        RTCamera viewcamera; viewcamera.from = data.from; viewcamera.to = data.to; viewcamera.up = data.up; viewcamera.camera = &scene->camera;
        // -orbit deg: what holding an arrow key does in the viewer (SceneLoader.h:159-173 -> yaw -> updateCamera), before the render starts
        if (orbitDeg != 0.0f) { viewcamera.yaw(orbitDeg); viewcamera.updateCamera(); }
        GamesEngineeringBase::Window canvas;
        canvas.create((unsigned int)scene->camera.width, (unsigned int)scene->camera.height, "Tracer", 1.0f);
        RayTracer rt;
        rt.shardMode = shard == "spp" || settings.algorithm == AL_LIGHT_TRACE ? 2 : 1;   // light-tracing splats land anywhere: samples, not tiles
        rt.shardRank = rank; rt.shardWorld = gpus;
        rt.init(scene, &canvas, settings);
        GamesEngineeringBase::Timer timer; float totalTime = 0;
        bool completed = false;
        if (gpus > 1) {
            // the whole render in one call: the ranks meet only at initSPP (adaptive sampling) and for the final film sum
            char id[128];
            if (rank == 0) { rtxCheck(rtx_rccl_unique_id(id), "rtx_rccl_unique_id"); for (int fd : idPipes) if (write(fd, id, sizeof id) != (ssize_t)sizeof id) throw std::runtime_error("cannot send the RCCL id"); }
            else if (read(idPipes.at(0), id, sizeof id) != (ssize_t)sizeof id) throw std::runtime_error("cannot receive the RCCL id");
            void* comm = nullptr; rtx_collective coll;
            rtxCheck(rtx_rccl_comm_create(scene->ctx, &comm, gpus, rank, id), "rtx_rccl_comm_create");
            rtxCheck(rtx_collective_rccl(&coll, comm), "rtx_collective_rccl");
            timer.reset();
            rt.renderSharded(&coll);
            totalTime = timer.dt();
            rtxCheck(rtx_rccl_comm_destroy(comm), "rtx_rccl_comm_destroy");
            completed = true;
            if (rank == 0) rt.saveHDR(filename);
        }
        while (!completed) {   // the body of the reference's frame loop, RTBase/Main.cpp:166-215
            timer.reset();
            if (batch > 1) rt.renderMT(std::min<int>(batch, std::max<int>(1, (int)settings.totalSPP - rt.getSPP())));
            else if (settings.useMultithreading) rt.renderMT(); else rt.render();
            float t = timer.dt();
            totalTime += t;
            float progress = rt.getSPP() * 100 / settings.totalSPP;
            if (!quiet) {
                std::cout << "Progress   : " << progress << "%\n" << "Samples    : " << rt.getSPP() << "\n" << "Time       : " << t << "\n"
                          << "FPS        : " << (t > 0 ? 1.0f / t : FLT_MAX) << "\n" << "Total time : " << std::roundf(totalTime) << " sec\n";
            }
            if ((int)settings.totalSPP <= rt.getSPP() && settings.drawMode == DM_ALGORITHM) { completed = true; rt.saveHDR(filename); }
        }
        if (rank != 0) { delete scene; return 0; }      // every rank holds the complete film; rank 0 writes it
        rt.draw();
        if (!png.empty()) rt.savePNG(png);
        if (!dump.empty()) {
            std::ofstream o(dump, std::ios::binary);
            uint32_t hdr[3] = {rt.film->width, rt.film->height, (uint32_t)rt.film->SPP};
            o.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
            o.write(reinterpret_cast<const char*>(rt.film->film), (std::streamsize)((size_t)rt.film->width * rt.film->height * sizeof(Colour)));
            o.write(reinterpret_cast<const char*>(rt.tileSamples.data()), (std::streamsize)(rt.tileSamples.size() * 4));
        }
        std::cout << "Done: " << rt.getSPP() << " spp in " << totalTime << " s (" << rt.getSPP() / totalTime << " spp/s)" << (gpus > 1 ? " on " + std::to_string(gpus) + " GPUs, split by " + shard : std::string()) << std::endl;
        delete scene;
    } catch (const std::exception& e) { std::cerr << "error: " << e.what() << std::endl; return 1; }
    return 0;
}
