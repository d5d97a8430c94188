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
    "      --checkpoint <FILE>              render the sample indices in portions and keep the radiance sums and the next\n"
    "                                       index in FILE after each; a run that finds FILE (same scene, image, sampler,\n"
    "                                       seed) goes on where it stopped (the reference writes its image once, at the end)\n"
    "      --checkpoint-every <N>           sample indices per portion [default: an eighth of the render]\n"
    "      --stats-json <FILE>              write timings and counters of the run as JSON\n"
    "      --dump-scene <FILE>              write the flattened scene (binary) and exit without rendering\n"
    "  -h, --help                           Print help\n"
    "  -V, --version                        Print version\n";

struct Args {
    std::string scene, render, output, stats_json, dump_scene, checkpoint;
    unsigned long long checkpoint_every = 0;
    long stop_after_portions = -1;  // developer switch: leave with status 3 after this many portions, as an interrupted run would
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
        } else if (arg == "--checkpoint") {
            a.checkpoint = value();
        } else if (arg == "--checkpoint-every") {
            a.checkpoint_every = std::strtoull(value().c_str(), nullptr, 10);
            if (a.checkpoint_every == 0) usage_error("invalid value for '--checkpoint-every'");
        } else if (arg == "--stop-after-portions") {
            a.stop_after_portions = std::atol(value().c_str());
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
    if (a.checkpoint.empty() && (a.checkpoint_every || a.stop_after_portions >= 0)) usage_error("'--checkpoint-every' needs '--checkpoint <FILE>'");
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
    // (an empty scene has no array to write: fwrite's pointer must not be null even for a count of zero)
    if (!sc.prims.empty()) std::fwrite(sc.prims.data(), sizeof(rrt_prim), sc.prims.size(), f);
    if (!sc.materials.empty()) std::fwrite(sc.materials.data(), sizeof(rrt_material), sc.materials.size(), f);
    if (std::fclose(f) != 0) throw std::runtime_error("short write to " + path);
}

// ---------------------------------------------------------------- checkpoint file
// Header, then width * height * 3 f32 radiance sums.  A file is taken up only by the render it came from: same scene
// (a hash of the flattened primitives, materials and camera), image, sampler, depth limits, seed, estimator and arithmetic.
struct CheckpointHeader {
    uint32_t magic, version;       // "RRTC", 1
    uint32_t width, height, sampler_type, spp, x_strata, y_strata, min_depth, max_depth, precision, flags;
    uint64_t seed, sample_begin, sample_end;
    uint64_t scene_hash;
    uint64_t next_sample;          // the sums hold the sample indices [sample_begin, next_sample) of every pixel
    uint64_t samples, rays, prim_tests, node_tests, nonfinite;  // counters of what was traced so far
    double kernel_ms;
};
const uint32_t kCheckpointMagic = 0x43545252u;  // "RRTC"

uint64_t fnv1a(const void* data, size_t bytes, uint64_t h) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < bytes; ++i) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

CheckpointHeader checkpoint_identity(const host::SceneData& sc, const rrt_render_desc& d) {
    CheckpointHeader h;
    std::memset(&h, 0, sizeof(h));
    h.magic = kCheckpointMagic;
    h.version = 1;
    h.width = d.width; h.height = d.height; h.sampler_type = d.sampler_type; h.spp = d.spp;
    h.x_strata = d.x_strata; h.y_strata = d.y_strata; h.min_depth = d.min_depth; h.max_depth = d.max_depth;
    h.precision = d.precision; h.flags = d.flags;
    h.seed = d.seed; h.sample_begin = d.sample_begin; h.sample_end = d.sample_end;
    uint64_t x = 0xcbf29ce484222325ull;
    x = fnv1a(&sc.camera, sizeof(sc.camera), x);
    x = fnv1a(sc.prims.data(), sizeof(rrt_prim) * sc.prims.size(), x);
    x = fnv1a(sc.materials.data(), sizeof(rrt_material) * sc.materials.size(), x);
    h.scene_hash = x;
    h.next_sample = d.sample_begin;
    return h;
}

// true: `h` and `sums` hold the file's state.  false: no such file.  Throws on a file of another render or a damaged one.
bool checkpoint_load(const std::string& path, const CheckpointHeader& want, CheckpointHeader& h, std::vector<float>& sums) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool got = std::fread(&h, sizeof(h), 1, f) == 1;
    bool same = got && h.magic == want.magic && h.version == want.version;
    if (same) {
        CheckpointHeader a = h, b = want;  // everything but the progress must agree
        a.next_sample = b.next_sample = 0;
        a.samples = a.rays = a.prim_tests = a.node_tests = a.nonfinite = 0;
        a.kernel_ms = 0;
        same = std::memcmp(&a, &b, sizeof(a)) == 0;
    }
    if (!same) {
        std::fclose(f);
        throw std::runtime_error(path + " is not a checkpoint of this render (another scene, image, sampler, seed or precision); remove it to start over");
    }
    const bool whole = std::fread(sums.data(), sizeof(float), sums.size(), f) == sums.size() && std::fgetc(f) == EOF;
    std::fclose(f);
    if (!whole || h.next_sample < h.sample_begin || h.next_sample > h.sample_end)
        throw std::runtime_error(path + " is damaged (wrong length or progress); remove it to start over");
    return true;
}

// Written beside the file and renamed over it: an interrupted write leaves the previous state.
void checkpoint_save(const std::string& path, const CheckpointHeader& h, const std::vector<float>& sums) {
    const std::string tmp = path + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + tmp);
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(sums.data(), sizeof(float), sums.size(), f) == sums.size();
    if (std::fclose(f) != 0 || !ok) throw std::runtime_error("short write to " + tmp);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot rename " + tmp + " to " + path);
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
    // between threads).  Of a portion [lo, hi) of sample indices GPU g traces [lo + g n / G, lo + (g + 1) n / G) of every
    // pixel with the same seed -- the stratified permutations are keyed on (seed, pixel, dimension), so
    // the shards together visit each stratum once -- into a host accumulator of its own.  Without --checkpoint the
    // whole render is one portion.
    struct DeviceContext {
        rrt_scene* scene = nullptr;
        rrt_backend* backend = nullptr;
        std::vector<float> sums;
        rrt_stats part;
        std::string error;
    };
    std::vector<DeviceContext> ctx(n_dev);
    auto worker = [&](size_t g, uint64_t lo, uint64_t hi) {
        DeviceContext& c = ctx[g];
        rrt_render_desc d = rs.render;
        const uint64_t n = hi - lo;
        d.sample_begin = lo + n * g / n_dev;
        d.sample_end = lo + n * (g + 1) / n_dev;
        std::memset(&c.part, 0, sizeof(rrt_stats));
        if (d.sample_end == d.sample_begin) return;  // more GPUs than samples
        c.sums.assign(n_values, 0.f);
        if (!c.scene) {
            rrt_scene_desc desc = sc.desc();
            if (rrt_scene_create(&desc, &c.scene) != RRT_OK || rrt_backend_create("hip", devices[g], &c.backend) != RRT_OK) {
                c.error = rrt_last_error();  // thread-local in the library
                return;
            }
        }
        if (rrt_render_accumulate(c.backend, c.scene, &d, c.sums.data(), &c.part) != RRT_OK) c.error = rrt_last_error();
    };
    auto release = [&]() {
        for (DeviceContext& c : ctx) {
            rrt_backend_destroy(c.backend);
            rrt_scene_destroy(c.scene);
            c.backend = nullptr;
            c.scene = nullptr;
        }
    };
    std::vector<float> total(n_values, 0.f);
    CheckpointHeader progress = checkpoint_identity(sc, rs.render);
    if (!a.checkpoint.empty()) {
        CheckpointHeader found;
        if (checkpoint_load(a.checkpoint, progress, found, total)) {
            progress = found;
            std::fprintf(stderr, "resuming %s at sample index %llu of [%llu, %llu)\n", a.checkpoint.c_str(), (unsigned long long)progress.next_sample,
                         (unsigned long long)progress.sample_begin, (unsigned long long)progress.sample_end);
        }
    }
    const uint64_t s_end = rs.render.sample_end, s_count = s_end - rs.render.sample_begin;
    uint64_t portion = s_count ? s_count : 1;
    if (!a.checkpoint.empty()) portion = a.checkpoint_every ? a.checkpoint_every : std::max<uint64_t>(1, (s_count + 7) / 8);
    long portions_done = 0;
    bool interrupted = false;
    for (uint64_t lo = progress.next_sample; lo < s_end; lo += portion) {
        if (a.stop_after_portions >= 0 && portions_done >= a.stop_after_portions) {
            interrupted = true;
            break;
        }
        const uint64_t hi = std::min(s_end, lo + portion);
        if (n_dev == 1) {
            worker(0, lo, hi);
        } else {
            std::vector<std::thread> threads;
            for (size_t g = 0; g < n_dev; ++g) threads.emplace_back(worker, g, lo, hi);
            for (std::thread& t : threads) t.join();
        }
        for (size_t g = 0; g < n_dev; ++g)
            if (!ctx[g].error.empty()) {
                const std::string what = "GPU " + std::to_string(devices[g]) + ": " + ctx[g].error;
                release();
                throw std::runtime_error(what);
            }
        double kernel_ms = 0;
        for (size_t g = 0; g < n_dev; ++g) {
            const DeviceContext& c = ctx[g];
            if (c.part.samples == 0 && c.part.rays == 0 && c.part.kernel_ms == 0) continue;  // no share of this portion
            for (size_t i = 0; i < n_values; ++i) total[i] += c.sums[i];
            progress.samples += c.part.samples;
            progress.rays += c.part.rays;
            progress.prim_tests += c.part.prim_tests;
            progress.node_tests += c.part.node_tests;
            progress.nonfinite += c.part.nonfinite;
            stats.exact_redos += c.part.exact_redos;
            stats.stack_overflows += c.part.stack_overflows;
            kernel_ms = std::max(kernel_ms, c.part.kernel_ms);
            stats.tree_build_ms = std::max(stats.tree_build_ms, c.part.tree_build_ms);
            stats.accel_used = c.part.accel_used;  // (the same on every GPU: one scene, one description)
            stats.pipeline_used = c.part.pipeline_used;
            stats.tree_depth = c.part.tree_depth;
        }
        progress.kernel_ms += kernel_ms;
        progress.next_sample = hi;
        portions_done += 1;
        if (!a.checkpoint.empty()) checkpoint_save(a.checkpoint, progress, total);
    }
    release();
    stats.samples = progress.samples;
    stats.rays = progress.rays;
    stats.prim_tests = progress.prim_tests;
    stats.node_tests = progress.node_tests;
    stats.nonfinite = progress.nonfinite;
    stats.kernel_ms = progress.kernel_ms;
    if (interrupted) {
        std::fprintf(stderr, "stopped after %ld portion(s) at sample index %llu; state in %s\n", portions_done,
                     (unsigned long long)progress.next_sample, a.checkpoint.c_str());
        return 3;
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
