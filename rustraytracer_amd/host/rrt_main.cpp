// Command-line renderer with the reference's interface (src/main.rs:18-50):
//   rrt -s <scene.toml> -r <render.toml> -o <image.png>
// reads the reference's TOML configs, renders on the HIP backend through the C ABI of include/rrt.h
// and saves an 8-bit RGB image in the format the output extension names.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>

#include "png_writer.hpp"
#include "scene_host.hpp"

namespace {

const char* kUsage =
    "A path tracer for MI355X with the command line of RustRayTracer.\n\n"
    "Usage: rrt --scene-config <SCENE_CONFIG> --render-config <RENDER_CONFIG> --output <OUTPUT>\n\n"
    "Options:\n"
    "  -s, --scene-config <SCENE_CONFIG>    scene description (TOML)\n"
    "  -r, --render-config <RENDER_CONFIG>  tracer / image / sampler / post-processing settings (TOML)\n"
    "  -o, --output <OUTPUT>                image file to write (.png or .ppm)\n"
    "      --device <N>                     GPU to render on [default: 0]\n"
    "      --devices <A,B,...>              render on several GPUs: the sample indices of every pixel are split\n"
    "                                       evenly, the radiance sums are added on the host\n"
    "      --seed <N>                       sampler seed (overrides `seed` of the render config)\n"
    "      --precision <f32|f64>            arithmetic of the kernels [default: f32]\n"
    "      --accel <auto|linear|lbvh|sah|sah_gpu>  closest-hit structure [default: auto]\n"
    "      --pipeline <auto|fused|wavefront> kernel organisation [default: auto]\n"
    "      --next-event                     sample the lights at every Lambertian / Phong vertex (next-event estimation:\n"
    "                                       same image in the mean, less noise with small lights; not the reference's estimator)\n"
    "      --exact-hits                     f32 only: repeat in f64 every closest-hit query that f32 cannot decide within its\n"
    "                                       rounding-error bounds (slower; fused pipeline)\n"
    "      --stats-json <FILE>              write timings and counters of the run as JSON\n"
    "      --dump-scene <FILE>              write the flattened scene (binary) and exit without rendering\n"
    "  -h, --help                           Print help\n"
    "  -V, --version                        Print version\n";

struct Args {
    std::string scene, render, output, stats_json, dump_scene;
    int device = 0;
    std::vector<int> devices;
    bool has_seed = false;
    unsigned long long seed = 0;
    uint32_t precision = RRT_PRECISION_F32, accel = RRT_ACCEL_AUTO, pipeline = RRT_PIPELINE_AUTO;
    uint32_t flags = 0;
};

[[noreturn]] void usage_error(const std::string& what) {
    std::fprintf(stderr, "error: %s\n\n%s", what.c_str(), kUsage);
    std::exit(2);
}

uint32_t choice(const std::string& opt, const std::string& v, const char* const* names, int n) {
    for (int k = 0; k < n; ++k)
        if (v == names[k]) return (uint32_t)k;
    usage_error("invalid value '" + v + "' for '" + opt + "'");
}

Args parse_args(int argc, char** argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i], val;
        bool has_val = false;
        size_t eq = arg.find('=');
        if (arg.compare(0, 2, "--") == 0 && eq != std::string::npos) {
            val = arg.substr(eq + 1);
            arg = arg.substr(0, eq);
            has_val = true;
        }
        auto value = [&]() -> std::string {
            if (has_val) return val;
            if (i + 1 >= argc) usage_error("a value is required for '" + arg + "' but none was supplied");
            return argv[++i];
        };
        if (arg == "-h" || arg == "--help") {
            std::fputs(kUsage, stdout);
            std::exit(0);
        } else if (arg == "-V" || arg == "--version") {
            std::puts("rrt 0.1.0 (HIP backend for gfx950)");
            std::exit(0);
        } else if (arg == "-s" || arg == "--scene-config") {
            a.scene = value();
        } else if (arg == "-r" || arg == "--render-config") {
            a.render = value();
        } else if (arg == "-o" || arg == "--output") {
            a.output = value();
        } else if (arg == "--device") {
            a.device = std::atoi(value().c_str());
        } else if (arg == "--devices") {
            std::string list = value();
            size_t pos = 0;
            while (pos <= list.size()) {
                size_t comma = list.find(',', pos);
                std::string item = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                if (item.empty() || item.find_first_not_of("0123456789") != std::string::npos)
                    usage_error("invalid value '" + list + "' for '--devices'");
                a.devices.push_back(std::atoi(item.c_str()));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
        } else if (arg == "--seed") {
            a.seed = std::strtoull(value().c_str(), nullptr, 10);
            a.has_seed = true;
        } else if (arg == "--precision") {
            static const char* const n[] = {"f32", "f64"};
            a.precision = choice(arg, value(), n, 2);
        } else if (arg == "--accel") {
            static const char* const n[] = {"auto", "linear", "lbvh", "sah", "sah_gpu"};
            a.accel = choice(arg, value(), n, 5);
        } else if (arg == "--pipeline") {
            static const char* const n[] = {"auto", "fused", "wavefront"};
            a.pipeline = choice(arg, value(), n, 3);
        } else if (arg == "--next-event") {
            a.flags |= RRT_RENDER_NEE;
        } else if (arg == "--exact-hits") {
            a.flags |= RRT_RENDER_EXACT_HITS;
        } else if (arg == "--stats-json") {
            a.stats_json = value();
        } else if (arg == "--dump-scene") {
            a.dump_scene = value();
        } else if (arg == "--self-test-image") {  // developer switch: a fixed 37x21 gradient through the image writer
            std::string out = value();
            std::vector<uint8_t> img(37 * 21 * 3);
            for (int y = 0; y < 21; ++y)
                for (int x = 0; x < 37; ++x)
                    for (int c = 0; c < 3; ++c) img[(size_t)(y * 37 + x) * 3 + c] = (uint8_t)((x * 7 + y * 13 + c * 101) & 0xff);
            try {
                host::save_image(out, img.data(), 37, 21);
            } catch (const std::exception& e) {
                std::fprintf(stderr, "error: %s\n", e.what());
                std::exit(101);
            }
            std::exit(0);
        } else {
            usage_error("unexpected argument '" + arg + "' found");
        }
    }
    if (a.scene.empty()) usage_error("the following required arguments were not provided: --scene-config <SCENE_CONFIG>");
    if (a.dump_scene.empty()) {
        if (a.render.empty()) usage_error("the following required arguments were not provided: --render-config <RENDER_CONFIG>");
        if (a.output.empty()) usage_error("the following required arguments were not provided: --output <OUTPUT>");
    }
    return a;
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void dump_scene(const host::SceneData& sc, const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    const uint32_t head[4] = {0x53545252u /* "RRTS" */, (uint32_t)sc.prims.size(), (uint32_t)sc.materials.size(), 0};
    std::fwrite(head, sizeof(head), 1, f);
    std::fwrite(&sc.camera, sizeof(sc.camera), 1, f);
    std::fwrite(sc.prims.data(), sizeof(rrt_prim), sc.prims.size(), f);
    std::fwrite(sc.materials.data(), sizeof(rrt_material), sc.materials.size(), f);
    if (std::fclose(f) != 0) throw std::runtime_error("short write to " + path);
}

#define RRT_CALL(expr)                                                         \
    do {                                                                       \
        if ((expr) != RRT_OK) throw std::runtime_error(std::string(#expr) + ": " + rrt_last_error()); \
    } while (0)

int run(const Args& a) {
    const auto t_start = std::chrono::steady_clock::now();
    host::SceneData sc = host::load_scene(a.scene);
    if (!a.dump_scene.empty()) {
        dump_scene(sc, a.dump_scene);
        return 0;
    }
    host::RenderSettings rs = host::load_render_config(a.render);
    rs.render.precision = a.precision;
    rs.render.accel = a.accel;
    rs.render.pipeline = a.pipeline;
    rs.render.flags = a.flags;
    if (a.has_seed) rs.render.seed = a.seed;
    const double t_load = seconds_since(t_start);

    std::vector<int> devices = a.devices.empty() ? std::vector<int>{a.device} : a.devices;
    const size_t n_dev = devices.size();
    const size_t n_values = (size_t)rs.render.width * rs.render.height * 3;
    std::vector<uint8_t> image(n_values);
    rrt_stats stats;
    std::memset(&stats, 0, sizeof(stats));
    const auto t_render = std::chrono::steady_clock::now();

    // One host thread per GPU, each with its own scene and backend handle (handles are not shared
    // between threads).  GPU g traces sample indices [begin + g n / G, begin + (g + 1) n / G) of every
    // pixel with the same seed -- the stratified permutations are keyed on (seed, pixel, dimension), so
    // the shards together visit each stratum once -- into a host accumulator of its own.
    const uint64_t s_begin = rs.render.sample_begin, s_count = rs.render.sample_end - rs.render.sample_begin;
    std::vector<std::vector<float>> sums(n_dev);
    std::vector<rrt_stats> part(n_dev);
    std::vector<std::string> errors(n_dev);
    auto worker = [&](size_t g) {
        rrt_render_desc d = rs.render;
        d.sample_begin = s_begin + s_count * g / n_dev;
        d.sample_end = s_begin + s_count * (g + 1) / n_dev;
        std::memset(&part[g], 0, sizeof(rrt_stats));
        if (d.sample_end == d.sample_begin) return;  // more GPUs than samples
        sums[g].assign(n_values, 0.f);
        rrt_scene_desc desc = sc.desc();
        rrt_scene* scene = nullptr;
        rrt_backend* backend = nullptr;
        if (rrt_scene_create(&desc, &scene) != RRT_OK || rrt_backend_create("hip", devices[g], &backend) != RRT_OK ||
            rrt_render_accumulate(backend, scene, &d, sums[g].data(), &part[g]) != RRT_OK)
            errors[g] = rrt_last_error();  // thread-local in the library
        rrt_backend_destroy(backend);
        rrt_scene_destroy(scene);
    };
    if (n_dev == 1) {
        worker(0);
    } else {
        std::vector<std::thread> threads;
        for (size_t g = 0; g < n_dev; ++g) threads.emplace_back(worker, g);
        for (std::thread& t : threads) t.join();
    }
    for (size_t g = 0; g < n_dev; ++g)
        if (!errors[g].empty()) throw std::runtime_error("GPU " + std::to_string(devices[g]) + ": " + errors[g]);
    std::vector<float> total(n_values, 0.f);
    for (size_t g = 0; g < n_dev; ++g) {
        if (sums[g].empty()) continue;
        for (size_t i = 0; i < n_values; ++i) total[i] += sums[g][i];
        stats.samples += part[g].samples;
        stats.rays += part[g].rays;
        stats.prim_tests += part[g].prim_tests;
        stats.node_tests += part[g].node_tests;
        stats.nonfinite += part[g].nonfinite;
        stats.exact_redos += part[g].exact_redos;
        stats.stack_overflows += part[g].stack_overflows;
        stats.kernel_ms = std::max(stats.kernel_ms, part[g].kernel_ms);
        stats.tree_build_ms = std::max(stats.tree_build_ms, part[g].tree_build_ms);
        stats.accel_used = part[g].accel_used;  // (the same on every GPU: one scene, one description)
        stats.pipeline_used = part[g].pipeline_used;
        stats.tree_depth = part[g].tree_depth;
    }
    {   // mean -> tone map -> gamma -> white balance -> u8 on the first GPU; the divisor is samples_per_pixel
        // even where RandomSampler traced one sample more (src/sampler.rs:48-55, src/renderer.rs:128)
        rrt_backend* backend = nullptr;
        RRT_CALL(rrt_backend_create("hip", devices[0], &backend));
        int rc = rrt_postprocess(backend, total.data(), rs.render.spp, &rs.post, rs.render.width, rs.render.height, image.data());
        std::string err = rc == RRT_OK ? "" : rrt_last_error();
        rrt_backend_destroy(backend);
        if (rc != RRT_OK) throw std::runtime_error("rrt_postprocess: " + err);
    }
    const double render_s = seconds_since(t_render);
    host::save_image(a.output, image.data(), rs.render.width, rs.render.height);
    const double total_s = seconds_since(t_start);
    std::fprintf(stderr, "rendered %ux%u, %llu samples, %llu rays in %.3f s (kernels %.3f s): %.1f Msamples/s; image saved to %s\n",
                 rs.render.width, rs.render.height, (unsigned long long)stats.samples, (unsigned long long)stats.rays,
                 render_s, stats.kernel_ms / 1e3, stats.samples / render_s / 1e6, a.output.c_str());
    if (!a.stats_json.empty()) {
        FILE* f = std::fopen(a.stats_json.c_str(), "w");
        if (!f) throw std::runtime_error("cannot write " + a.stats_json);
        std::fprintf(f,
                     "{\"n_gpus\": %zu, \"width\": %u, \"height\": %u, \"spp\": %u, \"n_prims\": %zu, \"n_objects\": %zu, \"samples\": %llu, "
                     "\"rays\": %llu, \"prim_tests\": %llu, \"node_tests\": %llu, \"nonfinite\": %llu, \"load_s\": %.6f, "
                     "\"render_s\": %.6f, \"kernel_s\": %.6f, \"total_s\": %.6f, \"accel_used\": %u, \"pipeline_used\": %u, "
                     "\"tree_depth\": %u, \"tree_build_ms\": %.3f, \"exact_redos\": %llu, \"stack_overflows\": %llu}\n",
                     n_dev, rs.render.width, rs.render.height, rs.render.spp, sc.prims.size(), sc.materials.size(),
                     (unsigned long long)stats.samples, (unsigned long long)stats.rays, (unsigned long long)stats.prim_tests,
                     (unsigned long long)stats.node_tests, (unsigned long long)stats.nonfinite, t_load, render_s,
                     stats.kernel_ms / 1e3, total_s, stats.accel_used, stats.pipeline_used, stats.tree_depth, stats.tree_build_ms,
                     (unsigned long long)stats.exact_redos, (unsigned long long)stats.stack_overflows);
        std::fclose(f);
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    Args a = parse_args(argc, argv);
    try {
        return run(a);
    } catch (const std::exception& e) {
        // the reference panics (exit status 101) on every one of these conditions
        std::fprintf(stderr, "error: %s\n", e.what());
        return 101;
    }
}
