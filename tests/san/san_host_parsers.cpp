// san_host_parsers.cpp -- the host mirror's parsers under AddressSanitizer / UBSan (CPU build only): the tolerant JSON
// reader (host/Json.hpp; the reference uses jsoncpp with comments allowed, LidarDevice.cpp:485-493) over the shipped sensor
// configs, transform files and trajectory.json AND over every truncation and a few hundred mutations of them -- each must
// end in a value or in an exception, never in a read past the text --, LidarDevice::initialize over the shipped configs
// (LidarDevice.cpp:482-633), the trajectory player (host/Trajectory.hpp), the STL ingest over the shipped meshes and
// truncations of them.  usage: san_host_parsers <tests/golden/data>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../lidarshooter_amd/host/HostTypes.hpp"
#include "../../lidarshooter_amd/host/Json.hpp"
#include "../../lidarshooter_amd/host/LidarDevice.hpp"
#include "../../lidarshooter_amd/host/Trajectory.hpp"

using namespace lidarshooter;

static std::string slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

int main(int argc, char **argv)
{
    const std::string data = argc > 1 ? argv[1] : "tests/golden/data";
    const std::vector<std::string> jsons = {"config/hesai-pandar-XT-32-lidar_0000.json", "config/hesai-pandar-XT-32-lidar_0001.json",
                                            "config/transform-lidar_0000.json", "config/transform-lidar_0001.json", "config/trajectory.json"};
    long parsed = 0, refused = 0;
    for (const std::string &rel : jsons) {
        const std::string text = slurp(data + "/" + rel);
        if (text.empty()) { std::printf("cannot read %s\n", rel.c_str()); return 2; }
        try { (void)json::parse(text); ++parsed; } catch (const std::exception &e) { std::printf("%s: %s\n", rel.c_str(), e.what()); return 1; }
        for (size_t cut = 0; cut < text.size(); cut += (text.size() > 4000 ? 7 : 1)) {
            try { (void)json::parse(text.substr(0, cut)); ++parsed; } catch (const std::exception &) { ++refused; }
        }
        unsigned seed = 99u;
        for (int m = 0; m < 400; ++m) {
            std::string t = text;
            for (int k = 0; k < 1 + m % 4; ++k) {
                seed = seed * 1664525u + 1013904223u;
                const size_t at = (seed >> 8) % t.size();
                seed = seed * 1664525u + 1013904223u;
                static const char junk[] = "{}[]\",:\\/u0eE-+.tfn \n\x01\xff";
                t[at] = junk[(seed >> 8) % (sizeof(junk) - 1)];
            }
            try { (void)json::parse(t); ++parsed; } catch (const std::exception &) { ++refused; }
        }
    }
    std::printf("json: %ld texts parsed, %ld refused with an exception\n", parsed, refused);
    for (const char *uid : {"0000", "0001"}) {
        auto dev = LidarDevice::create(data + "/config/hesai-pandar-XT-32-lidar_" + uid + ".json");
        if (dev->getTotalRays() != 4800u) { std::printf("lidar_%s: %u rays\n", uid, dev->getTotalRays()); return 1; }   // LidarDevice_test.cpp:58
        float d[3];
        dev->rayDirection(31, 149, d);
        PointCloud2 msg;
        dev->initMessage(msg, 3);
        if (msg.point_step != 32u || msg.fields.size() != 5u) return 1;   // LidarDevice_test.cpp:61-76
    }
    try { (void)LidarDevice::create(data + "/config/does-not-exist.json"); std::printf("a missing config was accepted\n"); return 1; } catch (const std::exception &) {}
    const auto poses = Trajectory::load(data + "/config/trajectory.json").play(0.1f);
    std::printf("trajectory: %zu poses\n", poses.size());
    if (poses.empty()) return 1;
    for (const char *name : {"ground", "ben"}) {
        const std::string path = data + "/mesh/" + name + ".stl";
        PolygonMesh mesh;
        if (loadPolygonFileSTL(path, mesh) <= 0) { std::printf("cannot load %s\n", path.c_str()); return 1; }
        std::printf("%s.stl: %u vertices, %zu triangles\n", name, mesh.cloud.width * mesh.cloud.height, mesh.polygons.size());
        const std::string bytes = slurp(path);
        for (size_t cut : {size_t(0), size_t(10), size_t(83), size_t(84), size_t(85), size_t(133), size_t(134), bytes.size() / 2, bytes.size() - 1}) {
            const std::string tmp = "/tmp/ls_san_truncated.stl";
            { std::ofstream o(tmp, std::ios::binary); o.write(bytes.data(), (std::streamsize)std::min(cut, bytes.size())); }
            PolygonMesh part;
            (void)loadPolygonFileSTL(tmp, part);   // a short file yields a short (or no) mesh: never a read past the data
        }
    }
    std::printf("ok\n");
    return 0;
}
