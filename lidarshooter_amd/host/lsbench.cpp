// lsbench.cpp -- a C++ caller of the C ABI (include/lidarshooter_hip.h) with no Python and no PyTorch in
// the process: sensor JSON and STL meshes through the host mirror (LidarDevice, loadPolygonFileSTL), the
// meshes uploaded to HBM once, then frames streamed the way MeshProjector.cpp:446-464 drives a tracer --
// updateGeometry for every mesh, commitScene, traceScene -- with the device-resident update variant.
// Prints one JSON line.  It is the harness SURVEY.md 8b calls "lsbench"; tests/test_gpu_parity.py runs it
// on the XT-32 scene and checks the reference's known answer (1781 points).
//
//   lsbench --config sensor.json [--mesh name=file.stl]... [--syn V H]  [--grid NX NY]
//           [--frames K] [--warmup W] [--pipeline 0|1|2] [--graph 0|1] [--engine 0|1|2]
//           [--ranks N [--group sharded|interleaved] [--group-flags F]]
//   --ranks N    : one process per GPU (forked before anything touches a GPU), frames through include/lidarshooter_group.h:
//                  azimuth shards + one ncclAllGather of hit-record slots per frame, or whole frames interleaved over the ranks
//   --syn V H    : replace the sensor's raster by V channels (+15 .. -25 deg) x H azimuths (0 .. 360 deg)
//   --grid NX NY : add a synthetic ground of NX x NY cells (2 triangles each) over [-50, 50]^2 m
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lidarshooter_group.h"
#include "../../include/lidarshooter_hip.h"
#include "HostTypes.hpp"
#include "LidarDevice.hpp"
#include "Sha256.hpp"

namespace {

struct DeviceMesh {
    std::string name;
    void* d_verts = nullptr;
    void* d_tris = nullptr;
    uint32_t stride = 12, n_verts = 0, n_tris = 0;
};

#define HIP_OK(x)                                                                                  \
    do {                                                                                           \
        const hipError_t e_ = (x);                                                                 \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)

int upload(DeviceMesh& m, const void* verts, size_t vbytes, const uint32_t* tris, size_t tbytes)
{
    HIP_OK(hipMalloc(&m.d_verts, vbytes));
    HIP_OK(hipMalloc(&m.d_tris, tbytes));
    HIP_OK(hipMemcpy(m.d_verts, verts, vbytes, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(m.d_tris, tris, tbytes, hipMemcpyHostToDevice));
    return 0;
}

}  // namespace

namespace {
std::string config, group_mode = "sharded";
std::vector<std::pair<std::string, std::string>> mesh_files, raw_files;
int synV = 0, synH = 0, gridX = 0, gridY = 0, frames = 200, warmup = 50, pipeline = 0, engine = 0, ranks = 0, group_flags = 0, frame_graph = 0;
int run(int rank, int world, const std::string& id_path, int result_fd);
}  // namespace

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { std::fprintf(stderr, "%s: missing value\n", a.c_str()); std::exit(2); } };
        if (a == "--config") { need(1); config = argv[++i]; }
        else if (a == "--mesh") { need(1); const std::string v = argv[++i]; const auto eq = v.find('='); mesh_files.push_back({v.substr(0, eq), v.substr(eq + 1)}); }
        else if (a == "--mesh-raw") { need(1); const std::string v = argv[++i]; const auto eq = v.find('='); raw_files.push_back({v.substr(0, eq), v.substr(eq + 1)}); }
        else if (a == "--syn") { need(2); synV = std::atoi(argv[++i]); synH = std::atoi(argv[++i]); }
        else if (a == "--grid") { need(2); gridX = std::atoi(argv[++i]); gridY = std::atoi(argv[++i]); }
        else if (a == "--frames") { need(1); frames = std::atoi(argv[++i]); }
        else if (a == "--warmup") { need(1); warmup = std::atoi(argv[++i]); }
        else if (a == "--pipeline") { need(1); pipeline = std::atoi(argv[++i]); }
        else if (a == "--engine") { need(1); engine = std::atoi(argv[++i]); }
        else if (a == "--ranks") { need(1); ranks = std::atoi(argv[++i]); }
        else if (a == "--group") { need(1); group_mode = argv[++i]; }
        else if (a == "--group-flags") { need(1); group_flags = std::atoi(argv[++i]); }   // LS_GROUP_FLAG_*: 1 one communicator (round 3's path), 2 no graph
        else if (a == "--graph") { need(1); frame_graph = std::atoi(argv[++i]); }         // single process: LS_OPT_FRAME_GRAPH
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (config.empty()) { std::fprintf(stderr, "--config is required\n"); return 2; }
    if (ranks <= 0) return run(0, 0, "", -1);
    if (group_mode != "sharded" && group_mode != "interleaved") { std::fprintf(stderr, "--group sharded|interleaved\n"); return 2; }

    // ---- one process per rank, forked before anything here has touched a GPU; results come back through pipes
    char id_path[] = "/tmp/lsbench_id_XXXXXX";
    const int idfd = mkstemp(id_path);
    if (idfd >= 0) close(idfd);
    unlink(id_path);   // rank 0 creates it (renamed into place) once it holds the RCCL id
    std::vector<pid_t> pids(ranks);
    std::vector<int> fds(ranks);
    for (int r = 0; r < ranks; ++r) {
        int pfd[2];
        if (pipe(pfd) != 0) { std::perror("pipe"); return 2; }
        const pid_t pid = fork();
        if (pid < 0) { std::perror("fork"); return 2; }
        if (pid == 0) {
            close(pfd[0]);
            const int rc = run(r, ranks, id_path, pfd[1]);
            close(pfd[1]);
            _exit(rc);
        }
        close(pfd[1]);
        pids[r] = pid;
        fds[r] = pfd[0];
    }
    double worst = 0.0, enq = 0.0;
    long ginfo[6] = {0, 0, 0, 0, 0, 0};   // rank 0: RCCL version, ncclCommCount, communicators, per-set mode, frame-graph state, graphs captured
    unsigned rays = 0, points = 0;
    unsigned long long tris = 0;
    char sha[80] = "";
    int rc = 0;
    for (int r = 0; r < ranks; ++r) {
        std::string text;
        char buf[512];
        for (ssize_t n; (n = read(fds[r], buf, sizeof(buf))) > 0;) text.append(buf, static_cast<size_t>(n));
        close(fds[r]);
        int status = 0;
        waitpid(pids[r], &status, 0);
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { std::fprintf(stderr, "rank %d failed\n", r); rc = 2; continue; }
        double el = 0, eq = 0;
        unsigned ry = 0, pt = 0, owns_last = 0;
        unsigned long long tt = 0;
        char h[80] = "";
        long gi[6] = {0, 0, 0, 0, 0, 0};
        if (std::sscanf(text.c_str(), "%lf %lf %u %llu %u %u %79s %ld %ld %ld %ld %ld %ld", &el, &eq, &ry, &tt, &pt, &owns_last, h, &gi[0], &gi[1],
                        &gi[2], &gi[3], &gi[4], &gi[5]) < 6) { rc = 2; continue; }
        if (r == 0) std::memcpy(ginfo, gi, sizeof(gi));
        worst = std::max(worst, el);
        enq = std::max(enq, eq);
        rays = ry;
        tris = tt;
        if (owns_last) { points = pt; std::snprintf(sha, sizeof(sha), "%s", h); }
    }
    unlink(id_path);
    if (rc) return rc;
    const unsigned whole = group_mode == "sharded" ? rays : rays;   // rays of the whole frame either way (reported by the ranks)
    std::printf("{\"harness\": \"lsbench\", \"ranks\": %d, \"group\": \"%s\", \"rays_per_frame\": %u, \"triangles\": %llu, \"frames\": %d, "
                "\"us_per_frame\": %.3f, \"frames_per_s\": %.1f, \"mrays_per_s\": %.1f, \"host_enqueue_us_per_frame\": %.3f, "
                "\"points_last_frame\": %u, \"points_sha256\": \"%s\", \"group_flags\": %d, \"rccl\": {\"version\": %ld, \"comm_ranks\": %ld, "
                "\"communicators\": %ld}, \"per_set_streams\": %ld, \"frame_graph_state\": %ld, \"frame_graphs_captured\": %ld}\n",
                ranks, group_mode.c_str(), whole, tris, frames, worst / frames * 1e6, frames / worst,
                static_cast<double>(whole) * frames / worst / 1e6, enq / frames * 1e6, points, sha, group_flags, ginfo[0], ginfo[1], ginfo[2],
                ginfo[3], ginfo[4], ginfo[5]);
    return 0;
}

namespace {

// one rank (world == 0: the plain single-process harness)
int run(int rank, int world, const std::string& id_path, int result_fd)
{
    int device = 0;
    if (world > 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { std::fprintf(stderr, "no HIP device\n"); return 2; }
        device = ndev >= world ? rank : 0;
        HIP_OK(hipSetDevice(device));
    }

    // ---- sensor: LidarDevice.cpp:482-633 through the host mirror; --syn swaps the raster, keeps the pose
    lidarshooter::LidarDevice::Ptr dev;
    try { dev = lidarshooter::LidarDevice::create(config); }
    catch (const std::exception& e) { std::fprintf(stderr, "sensor config: %s\n", e.what()); return 2; }
    ls_sensor_desc sd = dev->sensorDesc();
    std::vector<float> syn_vertical;
    if (synV > 0 && synH > 1) {
        syn_vertical.resize(synV);
        // lidarshooter_amd/synth.py syn_vertical: float64 arithmetic, then one rounding to float32 (bit-identical channel angles)
        for (int v = 0; v < synV; ++v) syn_vertical[v] = static_cast<float>(15.0 - static_cast<double>(v) * (40.0 / static_cast<double>(synV > 1 ? synV - 1 : 1)));
        sd.vertical_deg = syn_vertical.data();
        sd.n_vertical = static_cast<uint32_t>(synV);
        sd.h_begin = 0.0f;
        sd.h_end = 360.0f;
        sd.h_count = static_cast<uint32_t>(synH);
    }
    ls_tracer* tr = nullptr;
    if (ls_tracer_create(&sd, device, &tr) != LS_OK) { std::fprintf(stderr, "ls_tracer_create failed (no MI355X?)\n"); return 2; }
    ls_tracer_set_option(tr, LS_OPT_ENGINE, engine);

    // ---- meshes: STL files (pcl::io::loadPolygonFileSTL semantics) and / or a synthetic ground, resident in HBM
    std::vector<DeviceMesh> meshes;
    for (const auto& mf : mesh_files) {
        lidarshooter::PolygonMesh pm;
        if (lidarshooter::loadPolygonFileSTL(mf.second, pm) <= 0) { std::fprintf(stderr, "cannot load %s\n", mf.second.c_str()); return 2; }
        std::vector<uint32_t> idx;
        idx.reserve(pm.polygons.size() * 3);
        for (const auto& p : pm.polygons) { idx.push_back(p.vertices[0]); idx.push_back(p.vertices[1]); idx.push_back(p.vertices[2]); }
        DeviceMesh m;
        m.name = mf.first;
        m.stride = pm.cloud.point_step;
        m.n_verts = pm.cloud.width * pm.cloud.height;
        m.n_tris = static_cast<uint32_t>(pm.polygons.size());
        if (upload(m, pm.cloud.data.data(), pm.cloud.data.size(), idx.data(), idx.size() * 4)) return 2;
        meshes.push_back(m);
    }
    // raw dumps written by tools/dump_mesh.py: "LSMESH1\0", uint32 n_verts, uint32 n_tris, float32 xyz[n_verts], uint32 idx[3 n_tris]
    // -- BASELINE.md's synthetic meshes bit for bit as lidarshooter_amd/synth.py makes them (numpy's generator and its
    // vectorised sin / cos are not reproducible from C++; --grid below is a look-alike with its own noise)
    for (const auto& rf : raw_files) {
        FILE* f = std::fopen(rf.second.c_str(), "rb");
        char magic[8];
        uint32_t nv = 0, nt = 0;
        if (!f || std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "LSMESH1", 8) != 0 || std::fread(&nv, 4, 1, f) != 1 ||
            std::fread(&nt, 4, 1, f) != 1) {
            std::fprintf(stderr, "cannot read raw mesh %s\n", rf.second.c_str());
            return 2;
        }
        std::vector<float> v(static_cast<size_t>(nv) * 3);
        std::vector<uint32_t> t(static_cast<size_t>(nt) * 3);
        const bool ok = std::fread(v.data(), 4, v.size(), f) == v.size() && std::fread(t.data(), 4, t.size(), f) == t.size();
        std::fclose(f);
        if (!ok) { std::fprintf(stderr, "raw mesh %s is truncated\n", rf.second.c_str()); return 2; }
        DeviceMesh m;
        m.name = rf.first;
        m.n_verts = nv;
        m.n_tris = nt;
        if (upload(m, v.data(), v.size() * 4, t.data(), t.size() * 4)) return 2;
        meshes.push_back(m);
    }
    if (gridX > 0 && gridY > 0) {
        const int nx = gridX + 1, ny = gridY + 1;
        std::vector<float> v(static_cast<size_t>(nx) * ny * 3);
        uint32_t lcg = 20240u;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const float x = -50.0f + 100.0f * static_cast<float>(i) / static_cast<float>(gridX);
                const float y = -50.0f + 100.0f * static_cast<float>(j) / static_cast<float>(gridY);
                lcg = lcg * 1664525u + 1013904223u;
                const float noise = (static_cast<float>(lcg >> 8) / 16777216.0f - 0.5f) * 0.02f;
                float* p = &v[(static_cast<size_t>(j) * nx + i) * 3];
                p[0] = x; p[1] = y; p[2] = 0.25f * std::sin(0.35f * x) * std::cos(0.27f * y) + noise;
            }
        std::vector<uint32_t> t;
        t.reserve(static_cast<size_t>(gridX) * gridY * 6);
        for (int j = 0; j < gridY; ++j)
            for (int i = 0; i < gridX; ++i) {
                const uint32_t a = static_cast<uint32_t>(j * nx + i), b = a + 1, c = a + nx, d = c + 1;
                t.insert(t.end(), {a, b, d, a, d, c});
            }
        DeviceMesh m;
        m.name = "grid";
        m.n_verts = static_cast<uint32_t>(v.size() / 3);
        m.n_tris = static_cast<uint32_t>(t.size() / 3);
        if (upload(m, v.data(), v.size() * 4, t.data(), t.size() * 4)) return 2;
        meshes.push_back(m);
    }
    if (meshes.empty()) { std::fprintf(stderr, "no meshes\n"); return 2; }
    uint64_t total_tris = 0;
    for (const auto& m : meshes) {
        if (ls_add_geometry(tr, m.name.c_str(), LS_GEOMETRY_TYPE_TRIANGLE, static_cast<int>(m.n_verts), static_cast<int>(m.n_tris)) < 0) {
            std::fprintf(stderr, "ls_add_geometry: %s\n", ls_last_error(tr));
            return 2;
        }
        total_tris += m.n_tris;
    }

    static const float kIdentity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    if (world > 0) {
        // ---- frames through the group: every rank restates every mesh's pose, commits, and hands the frame to the group
        const uint32_t whole_rays = ls_total_rays(tr);
        ls_tracer_set_option(tr, LS_OPT_PIPELINE, pipeline);
        for (const auto& m : meshes)
            if (ls_update_geometry_device_shared(tr, m.name.c_str(), kIdentity, m.d_verts, m.stride, static_cast<const uint32_t*>(m.d_tris)) < 0) return 2;
        uint8_t id[LS_GROUP_ID_BYTES] = {0};
        const bool sharded = group_mode == "sharded";
        if (sharded) {   // the RCCL id: rank 0 makes it, the others read it from the file
            if (rank == 0) {
                if (ls_group_unique_id(id) != LS_OK) { std::fprintf(stderr, "ls_group_unique_id failed (librccl?)\n"); return 2; }
                const std::string tmp = id_path + ".tmp";
                FILE* f = std::fopen(tmp.c_str(), "wb");
                if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 2;
                std::fclose(f);
                std::rename(tmp.c_str(), id_path.c_str());
            } else {
                FILE* f = nullptr;
                for (int tries = 0; tries < 3000 && !(f = std::fopen(id_path.c_str(), "rb")); ++tries) usleep(10000);
                if (!f || std::fread(id, 1, sizeof(id), f) != sizeof(id)) { std::fprintf(stderr, "rank %d: no RCCL id\n", rank); return 2; }
                std::fclose(f);
            }
        }
        ls_group* g = nullptr;
        if (ls_group_create_opts(id, static_cast<uint32_t>(world), static_cast<uint32_t>(rank), sharded ? LS_GROUP_SHARDED : LS_GROUP_INTERLEAVED,
                                 static_cast<uint32_t>(group_flags), tr, &g) != LS_OK) {
            std::fprintf(stderr, "rank %d: ls_group_create failed\n", rank);
            return 2;
        }
        auto gframe = [&](uint32_t i) -> int {
            for (const auto& m : meshes)
                if (ls_update_geometry_transform(tr, m.name.c_str(), kIdentity) < 0) return -2;
            if (ls_commit_scene(tr) < -1) return -2;
            return ls_group_trace(g, i) < -1 ? -2 : 0;
        };
        for (int i = 0; i < warmup; ++i)
            if (gframe(static_cast<uint32_t>(i))) { std::fprintf(stderr, "frame: %s\n", ls_group_last_error(g)); return 2; }
        ls_group_synchronize(g);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < frames; ++i)
            if (gframe(static_cast<uint32_t>(i))) { std::fprintf(stderr, "frame: %s\n", ls_group_last_error(g)); return 2; }
        const double enqueue_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ls_group_synchronize(g);
        const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const uint32_t last = static_cast<uint32_t>(frames - 1);
        const bool owns_last = sharded ? rank == 0 : ls_group_owns_frame(g, last) != 0;
        long n = 0;
        std::string sha = "-";
        if (owns_last) {
            std::vector<uint8_t> pts(static_cast<size_t>(whole_rays) * 32);
            std::vector<ls_hit> hits(whole_rays);
            n = ls_group_download_cloud(g, last, pts.data(), hits.data(), whole_rays);
            if (n < 0) { std::fprintf(stderr, "download: %s\n", ls_group_last_error(g)); return 2; }
            // a sharded group's cloud is sector-major (rank 0's records, then rank 1's ...): hashed in ray order, the order of
            // the one-GPU cloud and of the oracle's
            std::vector<uint32_t> order(static_cast<size_t>(n));
            for (long i = 0; i < n; ++i) order[static_cast<size_t>(i)] = static_cast<uint32_t>(i);
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hits[a].ray < hits[b].ray; });
            std::vector<uint8_t> sorted(static_cast<size_t>(n) * 32);
            for (long i = 0; i < n; ++i) std::memcpy(&sorted[static_cast<size_t>(i) * 32], &pts[static_cast<size_t>(order[static_cast<size_t>(i)]) * 32], 32);
            sha = lidarshooter::sha256Hex(sorted.data(), sorted.size());
        }
        char line[384];
        const int len = std::snprintf(line, sizeof(line), "%.9f %.9f %u %llu %ld %d %s %ld %ld %ld %ld %ld %ld\n", elapsed, enqueue_s, whole_rays,
                                      static_cast<unsigned long long>(total_tris), n, owns_last ? 1 : 0, sha.c_str(),
                                      ls_group_info(g, LS_GROUP_INFO_RCCL_VERSION), ls_group_info(g, LS_GROUP_INFO_COMM_RANKS),
                                      ls_group_info(g, LS_GROUP_INFO_COMMUNICATORS), ls_group_info(g, LS_GROUP_INFO_PER_SET),
                                      ls_group_info(g, LS_GROUP_INFO_FRAME_GRAPH), ls_get_info(tr, LS_INFO_FRAME_GRAPH_CAPTURES));
        if (result_fd >= 0 && write(result_fd, line, static_cast<size_t>(len)) != len) return 2;
        ls_group_destroy(g);
        ls_tracer_destroy(tr);
        return 0;
    }

    // ---- outputs: three caller-owned sets (frames in flight rotate over them)
    const uint32_t rays = ls_total_rays(tr);
    struct Out { void *points = nullptr, *hits = nullptr; uint32_t* n = nullptr; } out[3];
    for (auto& o : out) {
        HIP_OK(hipMalloc(&o.points, static_cast<size_t>(rays) * 32));
        HIP_OK(hipMalloc(&o.hits, static_cast<size_t>(rays) * 16));
        HIP_OK(hipMalloc(reinterpret_cast<void**>(&o.n), 4));
    }
    ls_tracer_set_option(tr, LS_OPT_PIPELINE, pipeline);
    ls_tracer_set_option(tr, LS_OPT_FRAME_GRAPH, frame_graph);
    // the meshes are handed over once (in place, in HBM); after that every frame restates every mesh's pose
    // (MeshProjector.cpp:448-461 calls updateGeometry for every mesh, every frame): the unchanged-mesh update
    for (const auto& m : meshes)
        if (ls_update_geometry_device_shared(tr, m.name.c_str(), kIdentity, m.d_verts, m.stride, static_cast<const uint32_t*>(m.d_tris)) < 0) {
            std::fprintf(stderr, "ls_update_geometry_device_shared: %s\n", ls_last_error(tr));
            return 2;
        }
    auto frame = [&](uint32_t i) -> int {
        for (const auto& m : meshes)
            if (ls_update_geometry_transform(tr, m.name.c_str(), kIdentity) < 0) return -2;
        if (ls_commit_scene(tr) < -1) return -2;
        const Out& o = out[i % 3];
        if (ls_tracer_set_output_buffers(tr, o.points, o.hits, o.n, rays) < 0) return -2;
        ls_frame f;
        return ls_trace_scene_async(tr, i, &f) < -1 ? -2 : 0;
    };
    for (int i = 0; i < warmup; ++i)
        if (frame(static_cast<uint32_t>(i))) { std::fprintf(stderr, "frame: %s\n", ls_last_error(tr)); return 2; }
    ls_tracer_synchronize(tr);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < frames; ++i)
        if (frame(static_cast<uint32_t>(i))) { std::fprintf(stderr, "frame: %s\n", ls_last_error(tr)); return 2; }
    const double enqueue_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ls_tracer_synchronize(tr);
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint32_t n_points = 0;
    const Out& last = out[(frames - 1) % 3];
    HIP_OK(hipMemcpy(&n_points, last.n, 4, hipMemcpyDeviceToHost));
    // the last frame's cloud and hit records, hashed: the test compares them with the oracle's bytes
    std::vector<uint8_t> h_points(static_cast<size_t>(n_points) * 32), h_hits(static_cast<size_t>(n_points) * 16);
    if (n_points) {
        HIP_OK(hipMemcpy(h_points.data(), last.points, h_points.size(), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(h_hits.data(), last.hits, h_hits.size(), hipMemcpyDeviceToHost));
    }
    std::printf("{\"harness\": \"lsbench\", \"rays_per_frame\": %u, \"triangles\": %llu, \"frames\": %d, \"pipeline\": %d, \"frame_graph\": %d, \"frame_graph_replays\": %ld, \"frame_graph_patches\": %ld, "
                "\"us_per_frame\": %.3f, \"frames_per_s\": %.1f, \"mrays_per_s\": %.1f, \"host_enqueue_us_per_frame\": %.3f, "
                "\"points_last_frame\": %u, \"points_sha256\": \"%s\", \"hits_sha256\": \"%s\"}\n",
                rays, static_cast<unsigned long long>(total_tris), frames, pipeline, frame_graph, ls_get_info(tr, LS_INFO_FRAME_GRAPH_REPLAYS),
                ls_get_info(tr, LS_INFO_FRAME_GRAPH_PATCHES), elapsed / frames * 1e6, frames / elapsed,
                static_cast<double>(rays) * frames / elapsed / 1e6, enqueue_s / frames * 1e6, n_points,
                lidarshooter::sha256Hex(h_points.data(), h_points.size()).c_str(), lidarshooter::sha256Hex(h_hits.data(), h_hits.size()).c_str());
    ls_tracer_destroy(tr);
    return 0;
}

}  // namespace
