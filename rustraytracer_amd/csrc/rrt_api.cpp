// C ABI of the HIP backend (include/rrt.h): argument checking, host-side scene
// preparation, device memory, kernel launches.  There is no CPU compute path in
// this library: every entry point that traces rays runs kernels on a gfx950
// device or fails with an error code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/rrt.h"
#include "rrt_launch.hpp"
#include "rrt_lbvh.hpp"
#include "rrt_sah.hpp"
#include "rrt_types.hpp"
#include "rrt_wavefront.hpp"

// Staging memory of rrt_intersect_static: the rays of one call and their results.  One copy per GPU the
// library runs on, held until the process ends.  In host code the NAME of a __device__ array is a host-side stand-in,
// not a device address: every use below goes through the pointer hipGetSymbolAddress returned for the device.
struct StaticRays {
    double origins[3 * RRT_INTERSECT_STATIC_MAX_RAYS];
    double dirs[3 * RRT_INTERSECT_STATIC_MAX_RAYS];
    double t[RRT_INTERSECT_STATIC_MAX_RAYS];
    int32_t object_id[RRT_INTERSECT_STATIC_MAX_RAYS];
    int32_t prim_id[RRT_INTERSECT_STATIC_MAX_RAYS];
};
static __device__ __align__(256) unsigned char scratch[sizeof(StaticRays)];

namespace {

thread_local std::string g_last_error;

// Device address of `scratch` on each GPU (every GPU holds its own copy at its own address), resolved once per
// device by rrt_backend_create with that device current.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
StaticRays* g_scratch_ptr[kMaxDevices] = {};

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(RRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ------------------------------------------------------------------ f64 vector helpers
struct D3 {
    double x, y, z;
};
D3 d3(const double* p) { return D3{p[0], p[1], p[2]}; }
D3 operator-(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
D3 operator+(D3 a, D3 b) { return D3{a.x + b.x, a.y + b.y, a.z + b.z}; }
D3 operator*(D3 a, double s) { return D3{a.x * s, a.y * s, a.z * s}; }
double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
D3 cross(D3 a, D3 b) { return D3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
D3 normalize(D3 a) { return a * (1.0 / std::sqrt(dot(a, a))); }  // cgmath: v * (1 / |v|)

// ------------------------------------------------------------------ scene preparation
// Plane-form records (see rrt_types.hpp / rrt_device.hpp) in f64.
void prepare_prim(const rrt_prim& p, rrt::PrimRec<double>& rec, rrt::ShadeRec<double>& sh) {
    std::memset(&rec, 0, sizeof(rec));
    std::memset(&sh, 0, sizeof(sh));
    rec.type = (int32_t)p.type;
    sh.type = (int32_t)p.type;
    sh.object_id = (int32_t)p.object_id;
    sh.prim_id = (int32_t)p.prim_id;
    const double* v = p.v;
    if (p.type == RRT_PRIM_SPHERE) {
        rec.f[0] = v[0]; rec.f[1] = v[1]; rec.f[2] = v[2];
        rec.f[3] = v[3];
        rec.f[4] = v[3] * v[3];
        sh.nx = v[0]; sh.ny = v[1]; sh.nz = v[2]; sh.w = v[3];
        return;
    }
    if (p.type == RRT_PRIM_PLANE) {
        D3 point = d3(v), n = d3(v + 3);
        rec.f[0] = n.x; rec.f[1] = n.y; rec.f[2] = n.z;
        rec.f[3] = dot(n, point);
        sh.nx = n.x; sh.ny = n.y; sh.nz = n.z;
        return;
    }
    const bool quad = p.type == RRT_PRIM_QUAD;
    D3 v0 = d3(v), v1 = d3(v + 3), v2 = d3(v + 6);
    D3 vl = quad ? d3(v + 9) : v2;  // the corner that spans the second edge from v0
    D3 e01 = v1 - v0, e03 = vl - v0;
    D3 n = cross(e01, e03);
    auto dual = [](D3 ea, D3 eb, D3 nn, D3 corner, double* out4) {
        // coefficient of (P - corner) on ea in the basis (ea, eb) of the plane with normal nn
        D3 k = cross(eb, nn);
        double s = 1.0 / dot(ea, k);
        out4[0] = k.x * s; out4[1] = k.y * s; out4[2] = k.z * s;
        out4[3] = -(out4[0] * corner.x + out4[1] * corner.y + out4[2] * corner.z);
    };
    rec.f[0] = n.x; rec.f[1] = n.y; rec.f[2] = n.z;
    rec.f[3] = dot(n, v0);
    dual(e01, e03, n, v0, &rec.f[4]);
    dual(e03, e01, n, v0, &rec.f[8]);
    if (quad) {
        D3 v3 = vl;
        // v0 + v2 == v1 + v3 (to rounding): a parallelogram, whose second-corner coefficients are
        // 1 - alpha and 1 - beta; its record stores alpha - 1/2 and beta - 1/2 and nothing more
        D3 gap = (v0 + v2) - (v1 + v3);
        double scale = std::sqrt(dot(e01, e01)) + std::sqrt(dot(e03, e03));
        if (std::sqrt(dot(gap, gap)) <= 1e-12 * scale) {
            rec.type = rrt::PRIM_PGRAM;
            rec.f[7] -= 0.5;
            rec.f[11] -= 0.5;
            D3 sn0 = normalize(cross(v1 - v0, v2 - v0));
            sh.nx = sn0.x; sh.ny = sn0.y; sh.nz = sn0.z;
            return;
        }
        D3 e23 = v3 - v2, e21 = v1 - v2;
        D3 n2 = cross(e23, e21);
        dual(e23, e21, n2, v2, &rec.f[12]);
        dual(e21, e23, n2, v2, &rec.f[16]);
        rec.f[20] = n2.x; rec.f[21] = n2.y; rec.f[22] = n2.z;
        rec.f[23] = dot(n2, v2);
    }
    // the normal the reference reports for a hit: normalize((v1-v0) x (v2-v0))
    // (src/shapes/triangle.rs:78-80, quadrilateral.rs:180-182, mesh.rs:49-51,70-72)
    D3 sn = normalize(cross(v1 - v0, v2 - v0));
    sh.nx = sn.x; sh.ny = sn.y; sh.nz = sn.z;
}

void prepare_material(const rrt_material& m, rrt::MatRec<double>& out) {
    std::memset(&out, 0, sizeof(out));
    out.type = (int32_t)m.type;
    out.r = m.rgb[0]; out.g = m.rgb[1]; out.b = m.rgb[2];
    out.param = m.param;
    double mag = std::sqrt(m.rgb[0] * m.rgb[0] + m.rgb[1] * m.rgb[1] + m.rgb[2] * m.rgb[2]);
    out.emitter = (m.type == RRT_MAT_EMISSIVE && mag > 1e-6) ? 1 : 0;
    if (m.type != RRT_MAT_EMISSIVE && m.type != RRT_MAT_LAMBERTIAN && m.type != RRT_MAT_PHONG) {
        out.r = out.g = out.b = 0.0;
    }
    if (m.type == RRT_MAT_DIELECTRIC) {
        // eta = eta_i / eta_t and 1 / eta^2 for a ray that enters (1 / ior) and one that leaves (ior), as the
        // reference forms them (src/material.rs:193-284): the kernels select instead of dividing per bounce
        const double eta_in = 1.0 * (1.0 / m.param), eta_out = m.param * (1.0 / 1.0);
        out.r = eta_in;
        out.g = 1.0 / (eta_in * eta_in);
        out.b = 1.0 / (eta_out * eta_out);
    }
}

template <typename R>
rrt::PrimRec<R> narrow(const rrt::PrimRec<double>& a) {
    rrt::PrimRec<R> o;
    for (int i = 0; i < 24; ++i) o.f[i] = (R)a.f[i];
    o.type = a.type;
    return o;
}
template <typename R>
rrt::ShadeRec<R> narrow(const rrt::ShadeRec<double>& a) {
    rrt::ShadeRec<R> o;
    o.nx = (R)a.nx; o.ny = (R)a.ny; o.nz = (R)a.nz; o.w = (R)a.w;
    o.object_id = a.object_id; o.prim_id = a.prim_id; o.type = a.type; o.pad = 0;
    return o;
}
template <typename R>
rrt::MatRec<R> narrow(const rrt::MatRec<double>& a) {
    rrt::MatRec<R> o;
    o.r = (R)a.r; o.g = (R)a.g; o.b = (R)a.b; o.param = (R)a.param;
    o.type = a.type; o.emitter = a.emitter; o.pad[0] = o.pad[1] = 0;
    return o;
}

struct DeviceScene {
    int device = -1;
    std::vector<void*> allocs;
    rrt::SceneView<float> v32{};
    rrt::SceneView<double> v64{};
    rrt::LightView<float> l32{};
    rrt::LightView<double> l64{};
    const rrt::ErrRec* err32 = nullptr;  // per-primitive magnitudes for the error bounds of tracked f32 queries, scan order
    // One tree per builder (rrt_accel): [0] the cluster tree built on the GPU, [1] the surface-area tree built on the
    // host, [2] the surface-area tree built on the GPU.  v32 / v64 themselves never point at a tree: every call makes its own copy of the view and puts the tree
    // it asked for into that copy (with_tree), so two host threads that use one scene with different builders never
    // see each other's choice.
    struct Tree {
        bool built = false;
        int depth = 0;           // levels of wide nodes
        double build_ms = 0.0;   // host wall time of the build
        const rrt::WideNode* nodes = nullptr;
        const rrt::LeafRec<float>* leaves32 = nullptr;
        const rrt::LeafRec<double>* leaves64 = nullptr;
        const int32_t* bvh_prim = nullptr;
        const int32_t* linear_prim = nullptr;
        int32_t n_bvh_prims = 0, n_linear_prims = 0;
    } tree[3];
    int n_unbounded_lights = 0;  // emissive planes: next-event estimation cannot sample them
    ~DeviceScene() {
        if (device >= 0) {
            int cur = 0;
            (void)hipGetDevice(&cur);
            (void)hipSetDevice(device);
            for (void* p : allocs) (void)hipFree(p);
            (void)hipSetDevice(cur);
        }
    }
};

}  // namespace

struct rrt_scene {
    rrt_camera camera;
    std::vector<rrt_prim> prims;
    std::vector<rrt_material> materials;
    std::map<int, DeviceScene*> dev;
    // Guards `dev` and the first-use builds (device tables, trees): host threads may share one scene, each with a
    // backend of its own; what they build on first use is built once, by whoever comes first.
    std::mutex mu;
};

struct rrt_backend {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    unsigned long long* d_counters = nullptr;  // COUNTER_ROWS x COUNTER_COUNT + 1 (the last one is the work counter)
    // wavefront pipeline: queues and control block, allocated on first use and kept
    void* wf_mem = nullptr;
    uint32_t wf_capacity = 0;
    bool wf_has_shadow = false;  // the shadow-ray queue of next-event estimation is part of wf_mem
    size_t wf_elem = 0;          // bytes per real number in the queues: 4 (f32, the product path) or 8
};

namespace {

// fn(thread, begin, end) over [0, count) in contiguous shares on up to 16 host threads (one for small counts); a thread
// that cannot be had (a container's pid limit) leaves its share to the calling thread.  Nothing thrown inside fn may
// escape a thread: the loops run through it allocate nothing.
int share_threads(size_t count) {
    return count < 20000 ? 1 : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
}
void run_in_shares(size_t count, int n_threads, const std::function<void(int, size_t, size_t)>& fn) {
    std::vector<std::thread> pool;
    const size_t per = (count + (size_t)n_threads - 1) / (size_t)n_threads;
    int started = 1;
    try {
        for (; started < n_threads; ++started) {
            const size_t b = std::min(count, per * (size_t)started), e = std::min(count, b + per);
            pool.emplace_back(fn, started, b, e);
        }
    } catch (const std::system_error&) {
    }
    fn(0, 0, std::min(count, per));
    for (std::thread& t : pool) t.join();
    if (started < n_threads) fn(started, std::min(count, per * (size_t)started), count);  // what the missing threads would have done
}

template <typename T>
int upload(DeviceScene* ds, const std::vector<T>& host, const T** out, hipStream_t stream) {
    *out = nullptr;
    if (host.empty()) return RRT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, host.size() * sizeof(T)));
    ds->allocs.push_back(p);
    HIP_TRY(hipMemcpyAsync(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, stream));
    *out = (const T*)p;
    return RRT_OK;
}

// Area lights for next-event estimation: every primitive of an emitting object, in scan order.  A parallelogram is
// one light, any other quad the two triangles the reference's test covers it with ((v0, v1, v3) and (v2, v3, v1),
// src/shapes/quadrilateral.rs:56-117).  Emissive planes have no area to sample: counted in n_unbounded.
struct HostLights {
    std::vector<rrt::LightRec<double>> recs;
    std::vector<float> cdf;
    int n_unbounded = 0;
};

void build_lights(const rrt_scene* sc, HostLights& out) {
    std::vector<double> weight, area;
    auto planar = [&](int type, D3 v0, D3 e1, D3 e2, const rrt_material& m, size_t prim) {
        rrt::LightRec<double> l;
        std::memset(&l, 0, sizeof(l));
        l.type = type;
        l.prim = (int32_t)prim;
        D3 c = cross(e1, e2);
        const double len = std::sqrt(dot(c, c));
        const double a = type == rrt::LIGHT_TRIANGLE ? 0.5 * len : len;
        if (!(a > 0.0)) return;  // a degenerate face emits nothing
        D3 n = c * (1.0 / len);
        l.v0[0] = v0.x; l.v0[1] = v0.y; l.v0[2] = v0.z;
        l.e1[0] = e1.x; l.e1[1] = e1.y; l.e1[2] = e1.z;
        l.e2[0] = e2.x; l.e2[1] = e2.y; l.e2[2] = e2.z;
        l.n[0] = n.x; l.n[1] = n.y; l.n[2] = n.z;
        for (int k = 0; k < 3; ++k) l.le[k] = m.rgb[k];
        out.recs.push_back(l);
        area.push_back(a);
        weight.push_back(a * std::max(m.rgb[0], std::max(m.rgb[1], m.rgb[2])));
    };
    for (size_t pi = 0; pi < sc->prims.size(); ++pi) {
        const rrt_prim& p = sc->prims[pi];
        const rrt_material& m = sc->materials[p.object_id];
        const double mag = std::sqrt(m.rgb[0] * m.rgb[0] + m.rgb[1] * m.rgb[1] + m.rgb[2] * m.rgb[2]);
        if (m.type != RRT_MAT_EMISSIVE || !(mag > 1e-6)) continue;
        const double* v = p.v;
        if (p.type == RRT_PRIM_PLANE) {
            out.n_unbounded += 1;
        } else if (p.type == RRT_PRIM_SPHERE) {
            rrt::LightRec<double> l;
            std::memset(&l, 0, sizeof(l));
            l.type = rrt::LIGHT_SPHERE;
            l.prim = (int32_t)pi;
            l.v0[0] = v[0]; l.v0[1] = v[1]; l.v0[2] = v[2];
            l.e1[0] = v[3];
            for (int k = 0; k < 3; ++k) l.le[k] = m.rgb[k];
            const double a = 4.0 * 3.14159265358979323846 * v[3] * v[3];
            if (!(a > 0.0)) continue;
            out.recs.push_back(l);
            area.push_back(a);
            weight.push_back(a * std::max(m.rgb[0], std::max(m.rgb[1], m.rgb[2])));
        } else if (p.type == RRT_PRIM_TRIANGLE) {
            planar(rrt::LIGHT_TRIANGLE, d3(v), d3(v + 3) - d3(v), d3(v + 6) - d3(v), m, pi);
        } else {
            D3 v0 = d3(v), v1 = d3(v + 3), v2 = d3(v + 6), v3 = d3(v + 9);
            D3 gap = (v0 + v2) - (v1 + v3);
            const double scale = std::sqrt(dot(v1 - v0, v1 - v0)) + std::sqrt(dot(v3 - v0, v3 - v0));
            if (std::sqrt(dot(gap, gap)) <= 1e-12 * scale) {
                planar(rrt::LIGHT_PGRAM, v0, v1 - v0, v3 - v0, m, pi);
            } else {
                planar(rrt::LIGHT_TRIANGLE, v0, v1 - v0, v3 - v0, m, pi);
                planar(rrt::LIGHT_TRIANGLE, v2, v3 - v2, v1 - v2, m, pi);
            }
        }
    }
    double total = 0.0;
    for (double w : weight) total += w;
    double run = 0.0;
    for (size_t i = 0; i < out.recs.size(); ++i) {
        const double prob = total > 0.0 ? weight[i] / total : 1.0 / (double)out.recs.size();
        run += prob;
        out.recs[i].inv_pdf = prob > 0.0 ? area[i] / prob : 0.0;
        out.cdf.push_back(i + 1 == out.recs.size() ? 1.0f : (float)run);
    }
}

template <typename R>
rrt::LightRec<R> narrow(const rrt::LightRec<double>& a) {
    rrt::LightRec<R> o;
    for (int k = 0; k < 3; ++k) {
        o.v0[k] = (R)a.v0[k]; o.e1[k] = (R)a.e1[k]; o.e2[k] = (R)a.e2[k]; o.n[k] = (R)a.n[k]; o.le[k] = (R)a.le[k];
    }
    o.inv_pdf = (R)a.inv_pdf;
    o.type = a.type;
    o.prim = a.prim;
    return o;
}

template <typename R>
int fill_view(DeviceScene* ds, const rrt_scene* sc, rrt::SceneView<R>& view, rrt::LightView<R>& lview,
              const std::vector<rrt::PrimRec<double>>& prims, const std::vector<rrt::ShadeRec<double>>& shade,
              const std::vector<rrt::MatRec<double>>& mats, hipStream_t stream) {
    const auto t_view = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (getenv("RRT_DEBUG_BUILD_TIMES")) fprintf(stderr, "[rrt]   fill_view<%zu>: %s at %.1f ms\n", sizeof(R), what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_view).count()); };
    std::vector<rrt::PrimRec<R>> p(prims.size());
    std::vector<rrt::ShadeRec<R>> s(shade.size());
    std::vector<rrt::MatRec<R>> m(mats.size());
    const int n_threads = share_threads(prims.size());
    run_in_shares(prims.size(), n_threads, [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            p[i] = narrow<R>(prims[i]);
            s[i] = narrow<R>(shade[i]);
        }
    });
    for (size_t i = 0; i < mats.size(); ++i) m[i] = narrow<R>(mats[i]);
    lap("narrowed");
    std::memset(&view, 0, sizeof(view));
    for (int k = 0; k < 3; ++k) {
        view.cam_o[k] = (R)sc->camera.origin[k];
        view.cam_ll[k] = (R)(sc->camera.lower_left_corner[k] - sc->camera.origin[k]);
        view.cam_h[k] = (R)sc->camera.horizontal[k];
        view.cam_v[k] = (R)sc->camera.vertical[k];
    }
    int rc;
    if (sizeof(R) == 4) {  // magnitudes for the error bounds of tracked f32 queries (rrt_device.hpp, HitTrack), rounded up
        std::vector<rrt::ErrRec> err(prims.size());
        auto up = [](double x) { return std::nextafterf((float)x, INFINITY) * 1.000001f; };
        run_in_shares(prims.size(), n_threads, [&](int, size_t b0, size_t e0) {
          for (size_t i = b0; i < e0; ++i) {
            const double* f = prims[i].f;
            auto sum3 = [&](int k) { return std::fabs(f[k]) + std::fabs(f[k + 1]) + std::fabs(f[k + 2]); };
            rrt::ErrRec e;
            std::memset(&e, 0, sizeof(e));
            if (prims[i].type == rrt::PRIM_SPHERE) {
                e.e[0] = up(std::fmax(std::fmax(std::fabs(f[0]), std::fabs(f[1])), std::fabs(f[2])));
            } else {
                e.e[0] = up(sum3(0));
                e.e[1] = up(std::fabs(f[3]));
                if (prims[i].type != rrt::PRIM_PLANE) {
                    e.e[2] = up(std::fmax(sum3(4), sum3(8)));
                    e.e[3] = up(std::fmax(std::fabs(f[7]), std::fabs(f[11])) + 1.0);
                }
                if (prims[i].type == rrt::PRIM_QUAD) {
                    e.e[4] = up(sum3(20));
                    e.e[5] = up(std::fabs(f[23]));
                    e.e[6] = up(std::fmax(sum3(12), sum3(16)));
                    e.e[7] = up(std::fmax(std::fabs(f[15]), std::fabs(f[19])) + 1.0);
                }
            }
            err[i] = e;
          }
        });
        lap("magnitudes");
        if ((rc = upload(ds, err, &ds->err32, stream)) != RRT_OK) return rc;
        lap("magnitudes uploaded");
    }
    if ((rc = upload(ds, p, &view.prims, stream)) != RRT_OK) return rc;
    if ((rc = upload(ds, s, &view.shade, stream)) != RRT_OK) return rc;
    if ((rc = upload(ds, m, &view.mats, stream)) != RRT_OK) return rc;
    view.n_prims = (int32_t)prims.size();
    view.n_objects = (int32_t)mats.size();
    lap("records uploaded");
    std::vector<int32_t> runs;  // maximal runs of equal record type, in scan order
    for (size_t i = 0; i < prims.size(); ++i) {
        if (i == 0 || prims[i].type != prims[i - 1].type) {
            runs.push_back(prims[i].type);
            runs.push_back((int32_t)i);
            runs.push_back(0);
            runs.push_back(0);
        }
        runs[runs.size() - 2] += 1;
    }
    // parallelogram and triangle runs, and sphere and plane runs of more than one, two records side by side
    std::vector<rrt::PairRec<R>> pairs;
    {
        size_t n_pairs = 0;
        for (size_t r = 0; r < runs.size(); r += 4) {
            const bool round = runs[r] == rrt::PRIM_SPHERE || runs[r] == rrt::PRIM_PLANE;
            if (runs[r] == rrt::PRIM_QUAD || (round && runs[r + 2] < 2)) continue;
            runs[r + 3] = (int32_t)n_pairs;
            n_pairs += (size_t)(runs[r + 2] + 1) / 2;
        }
        pairs.resize(n_pairs);
        for (size_t r = 0; r < runs.size(); r += 4) {
            const bool round = runs[r] == rrt::PRIM_SPHERE || runs[r] == rrt::PRIM_PLANE;
            if (runs[r] == rrt::PRIM_QUAD || (round && runs[r + 2] < 2)) continue;
            const int type = runs[r], first = runs[r + 1], count = runs[r + 2];
            rrt::PairRec<R>* out = pairs.data() + runs[r + 3];
            const size_t in_run = (size_t)(count + 1) / 2;
            run_in_shares(in_run, share_threads(in_run), [&](int, size_t b, size_t e) {
                for (size_t q = b; q < e; ++q) {
                    const int k = (int)(2 * q);
                    rrt::PairRec<R> pr;
                    std::memset(&pr, 0, sizeof(pr));
                    // the pad of an odd run must be missed by every ray: n = 0 does that for the plane forms, a sphere
                    // needs radius^2 < 0 (with radius^2 = 0 a ray through its centre would hit it)
                    if (type == rrt::PRIM_SPHERE) pr.g[4][1] = (R)-1;
                    for (int h = 0; h < 2 && k + h < count; ++h)
                        for (int j = 0; j < 12; ++j) pr.g[j][h] = p[(size_t)first + (size_t)(k + h)].f[j];
                    out[q] = pr;
                }
            });
        }
    }
    lap("pairs made");
    if ((rc = upload(ds, runs, &view.runs, stream)) != RRT_OK) return rc;
    if ((rc = upload(ds, pairs, &view.pairs, stream)) != RRT_OK) return rc;
    view.n_runs = (int32_t)(runs.size() / 4);
    HostLights hl;
    build_lights(sc, hl);
    std::vector<rrt::LightRec<R>> lights(hl.recs.size());
    for (size_t i = 0; i < hl.recs.size(); ++i) lights[i] = narrow<R>(hl.recs[i]);
    std::memset(&lview, 0, sizeof(lview));
    if ((rc = upload(ds, lights, &lview.lights, stream)) != RRT_OK) return rc;
    if ((rc = upload(ds, hl.cdf, &lview.cdf, stream)) != RRT_OK) return rc;
    lview.n = (int32_t)lights.size();
    ds->n_unbounded_lights = hl.n_unbounded;
    HIP_TRY(hipStreamSynchronize(stream));  // the staging vectors die here
    lap("all uploaded");
    return RRT_OK;
}

// Device copy of a scene for the backend's GPU, built on first use.
int device_scene(rrt_backend* be, const rrt_scene* sc_const, DeviceScene** out) {
    rrt_scene* sc = const_cast<rrt_scene*>(sc_const);
    std::lock_guard<std::mutex> lock(sc->mu);
    auto it = sc->dev.find(be->device);
    if (it != sc->dev.end()) {
        *out = it->second;
        return RRT_OK;
    }
    DeviceScene* ds = new (std::nothrow) DeviceScene();
    if (!ds) return fail(RRT_ERR_NOMEM, "out of host memory");
    ds->device = be->device;
    static const bool print_times = getenv("RRT_DEBUG_BUILD_TIMES") != nullptr;  // developer switch: phases to stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&t_begin]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<rrt::PrimRec<double>> prims(sc->prims.size());
    std::vector<rrt::ShadeRec<double>> shade(sc->prims.size());
    std::vector<rrt::MatRec<double>> mats(sc->materials.size());
    run_in_shares(sc->prims.size(), share_threads(sc->prims.size()), [&](int, size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) prepare_prim(sc->prims[i], prims[i], shade[i]);
    });
    for (size_t i = 0; i < sc->materials.size(); ++i) prepare_material(sc->materials[i], mats[i]);
    if (print_times) fprintf(stderr, "[rrt] scene tables: records of %zu primitives at %.1f ms\n", sc->prims.size(), since());
    int rc = fill_view<float>(ds, sc, ds->v32, ds->l32, prims, shade, mats, be->stream);
    if (print_times) fprintf(stderr, "[rrt] scene tables: f32 view at %.1f ms\n", since());
    if (rc == RRT_OK) rc = fill_view<double>(ds, sc, ds->v64, ds->l64, prims, shade, mats, be->stream);
    if (print_times) fprintf(stderr, "[rrt] scene tables: f64 view at %.1f ms\n", since());
    if (rc != RRT_OK) {
        delete ds;
        return rc;
    }
    sc->dev[be->device] = ds;
    *out = ds;
    return RRT_OK;
}

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

// Scenes with more primitives than this get the surface-area tree under RRT_ACCEL_AUTO, and tree scenes with more
// primitives than kAutoWavefrontThreshold the wavefront pipeline under RRT_PIPELINE_AUTO.  Measured on
// triangle meshes at 1024x1024 @ 256 spp (tools/gpu_crossover.py), Grays/s:
//   primitives               26     50     92    184    382    758   1484   5854  23718  99858  399426
//   scan, fused            42.4   27.3   16.3    8.7    4.3    2.2    1.1    0.3
//   cluster tree, fused    25.7   22.4   19.4   16.9   14.6   12.9   11.2    9.3    7.8    6.0    4.8
//   cluster tree, wavefront 8.1    8.0    7.9    7.8    7.6    7.3    7.1    6.8    6.4    6.1    6.1
//   area tree, fused       25.9   22.0   19.6   16.9   15.0   12.8   12.2    9.9    8.2    6.5    5.4
//   area tree, wavefront    8.1    8.1    8.0    7.8    7.6    7.4    7.4    7.0    6.6    6.4    6.4   (1 M triangles: 6.2 against 3.7 fused)
const size_t kAutoBvhThreshold = 72;
const size_t kAutoWavefrontThreshold = 150000;
// One primitive per leaf.  A traversal step costs one 64-byte fetch whether it reads a node or a primitive, and with
// four child boxes per node a second primitive in a leaf is tested in vain more often than a node step is saved
// (million-triangle scene, node + primitive steps per ray, surface-area tree: 1 -> 13.3 + 3.2, 2 -> 11.9 + 5.4, 3 -> 11.6 + 7.2).
const int kBvhLeafMax = 1;

float round_down(double x) {
    float f = (float)x;
    return (double)f > x ? std::nextafterf(f, -INFINITY) : f;
}
float round_up(double x) {
    float f = (float)x;
    return (double)f < x ? std::nextafterf(f, INFINITY) : f;
}

// A copy of a scene view that points at one of the scene's trees (kind < 0: none, the linear scan).
rrt::SceneView<float> with_tree(const rrt_backend* be, const DeviceScene* ds, const rrt::SceneView<float>& base, int kind) {
    rrt::SceneView<float> v = base;
    v.stack_overflow = be->d_counters + rrt::COUNTER_STACK_OVERFLOW;
    if (kind < 0) return v;
    const DeviceScene::Tree& t = ds->tree[kind];
    v.nodes = t.nodes; v.leaves = t.leaves32; v.bvh_prim = t.bvh_prim; v.linear_prim = t.linear_prim;
    v.n_bvh_prims = t.n_bvh_prims; v.n_linear_prims = t.n_linear_prims;
    return v;
}
rrt::SceneView<double> with_tree(const rrt_backend* be, const DeviceScene* ds, const rrt::SceneView<double>& base, int kind) {
    rrt::SceneView<double> v = base;
    v.stack_overflow = be->d_counters + rrt::COUNTER_STACK_OVERFLOW;
    if (kind < 0) return v;
    const DeviceScene::Tree& t = ds->tree[kind];
    v.nodes = t.nodes; v.leaves = t.leaves64; v.bvh_prim = t.bvh_prim; v.linear_prim = t.linear_prim;
    v.n_bvh_prims = t.n_bvh_prims; v.n_linear_prims = t.n_linear_prims;
    return v;
}

// Builds a tree of a device scene on first use (shared by the f32 and f64 views: boxes are f32).
// kind 0: clustering along the Morton order, built on the GPU (rrt_lbvh.hip, rrt_ploc.hip); kind 1: binned surface-area
// heuristic, built by the host's threads (rrt_sah.hpp).  Both end in the same 4-wide quantised nodes and leaf records.
// A traversal holds at most three stack entries per level of wide nodes, and the kernels' stacks hold
// kTreeStackEntries: a tree deeper than that (the cost function can peel primitives off one by one on adversarial
// input; the clustering makes one level per round) is rebuilt with median cuts, which bound the depth by log2 n.
const int kTreeStackEntries = rrt::BVH_STACK_ENTRIES;
int ensure_bvh(rrt_backend* be, const rrt_scene* sc, DeviceScene* ds, int kind, bool* built_now) {
    std::lock_guard<std::mutex> lock(const_cast<rrt_scene*>(sc)->mu);
    if (built_now) *built_now = !ds->tree[kind].built;
    if (ds->tree[kind].built) return RRT_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<int32_t> bounded, unbounded;  // scan-order indices
    std::vector<float> lo, hi;
    double slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto bounds = [](const rrt_prim& p, double blo[3], double bhi[3]) {
        if (p.type == RRT_PRIM_SPHERE) {
            for (int k = 0; k < 3; ++k) { blo[k] = p.v[k] - std::fabs(p.v[3]); bhi[k] = p.v[k] + std::fabs(p.v[3]); }
        } else {
            const int nv = p.type == RRT_PRIM_TRIANGLE ? 3 : 4;
            for (int k = 0; k < 3; ++k) {
                blo[k] = bhi[k] = p.v[k];
                for (int j = 1; j < nv; ++j) {
                    blo[k] = std::fmin(blo[k], p.v[3 * j + k]);
                    bhi[k] = std::fmax(bhi[k], p.v[3 * j + k]);
                }
            }
        }
    };
    // Which primitives stay out of the tree and are scanned beside it: planes (unbounded), and quads that are bent.
    // A bent quad reports its hit on the plane of (v0, v1, v3) while the corner at v2 is judged on the plane of
    // (v2, v3, v1) (src/shapes/quadrilateral.rs:90-114); for a grazing ray the reported point can lie well outside the
    // box of the four corners, so no box makes the tree return what the scan returns.  Bent by up to 1e-6 of its
    // extent (corners rounded to f32 in a PLY file) a quad stays in the tree: the padding of its box covers rays down
    // to half a degree of grazing.
    auto outside_tree = [](const rrt_prim& p) {
        if (p.type == RRT_PRIM_PLANE) return true;
        if (p.type != RRT_PRIM_QUAD) return false;
        D3 v0 = d3(p.v), v1 = d3(p.v + 3), v2 = d3(p.v + 6), v3 = d3(p.v + 9);
        D3 n = cross(v1 - v0, v3 - v0);
        double len = std::sqrt(dot(n, n));
        double ext = std::sqrt(std::fmax(std::fmax(dot(v1 - v0, v1 - v0), dot(v2 - v0, v2 - v0)), dot(v3 - v0, v3 - v0)));
        if (!(len > 0.0) || !(ext > 0.0)) return false;  // degenerate: every ray misses it
        return std::fabs(dot(n, v2 - v0)) / len > 1e-6 * ext;
    };
    // Largest coordinate magnitude a traversal works with: the scene's bounded primitives and the
    // camera (every other ray starts on a surface).  The slab tests are exact up to a few f32 ulps
    // of the distance between ray origin and box, and the f32 primitive tests accept points a few
    // 1e-7 (relative to that scale) outside the exact primitive; the boxes are padded for both.
    // (three passes over the primitives: which are in the tree and the scale; their indices; their boxes.  The first
    // and the last run on the host's threads: alone they took 40 of the 80 ms of a million-triangle build.)
    static const bool print_times = getenv("RRT_DEBUG_BUILD_TIMES") != nullptr;  // developer switch: phases to stderr
    auto since = [&t_begin]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const size_t n_all = sc->prims.size();
    const int n_threads = share_threads(n_all);
    auto in_shares = [n_threads](size_t count, const std::function<void(int, size_t, size_t)>& fn) { run_in_shares(count, n_threads, fn); };
    std::vector<uint8_t> outside(n_all);
    double scale = 0;
    try {
        std::vector<double> share_scale((size_t)n_threads, 0.0);
        in_shares(n_all, [&](int t, size_t b, size_t e) {
            double m = 0;
            for (size_t g = b; g < e; ++g) {
                const rrt_prim& p = sc->prims[g];
                outside[g] = outside_tree(p) ? 1 : 0;
                if (outside[g]) continue;
                double blo[3], bhi[3];
                bounds(p, blo, bhi);
                for (int k = 0; k < 3; ++k) m = std::fmax(m, std::fmax(std::fabs(blo[k]), std::fabs(bhi[k])));
            }
            share_scale[(size_t)t] = std::fmax(share_scale[(size_t)t], m);
        });
        for (int k = 0; k < 3; ++k) scale = std::fmax(scale, std::fabs(sc->camera.origin[k]));
        for (double m : share_scale) scale = std::fmax(scale, m);
        for (size_t g = 0; g < n_all; ++g) (outside[g] ? unbounded : bounded).push_back((int32_t)g);
        lo.resize(3 * bounded.size());
        hi.resize(3 * bounded.size());
        std::vector<double> share_box((size_t)n_threads * 6);
        for (int t = 0; t < n_threads; ++t)
            for (int k = 0; k < 3; ++k) { share_box[(size_t)t * 6 + k] = INFINITY; share_box[(size_t)t * 6 + 3 + k] = -INFINITY; }
        in_shares(bounded.size(), [&](int t, size_t b, size_t e) {
            double* box = share_box.data() + (size_t)t * 6;
            for (size_t i = b; i < e; ++i) {
                double blo[3], bhi[3];
                bounds(sc->prims[(size_t)bounded[i]], blo, bhi);
                double ext = std::fmax(std::fmax(bhi[0] - blo[0], bhi[1] - blo[1]), bhi[2] - blo[2]);
                double pad = 1e-4 * ext + 4e-6 * scale + 1e-30;
                for (int k = 0; k < 3; ++k) {
                    lo[3 * i + k] = round_down(blo[k] - pad);
                    hi[3 * i + k] = round_up(bhi[k] + pad);
                    box[k] = std::fmin(box[k], blo[k]);
                    box[3 + k] = std::fmax(box[3 + k], bhi[k]);
                }
            }
        });
        for (int t = 0; t < n_threads; ++t)
            for (int k = 0; k < 3; ++k) {
                slo[k] = std::fmin(slo[k], share_box[(size_t)t * 6 + k]);
                shi[k] = std::fmax(shi[k], share_box[(size_t)t * 6 + 3 + k]);
            }
    } catch (const std::bad_alloc&) {
        return fail(RRT_ERR_NOMEM, "out of host memory while preparing the tree of %zu primitives", n_all);
    }
    if (print_times) fprintf(stderr, "[rrt] tree build: boxes of %zu primitives on %d thread(s) at %.1f ms\n", n_all, n_threads, since());
    if (bounded.size() > (size_t)rrt::BVH_MAX_PRIMS)
        return fail(RRT_ERR_UNSUPPORTED, "the LBVH addresses at most %d bounded primitives", rrt::BVH_MAX_PRIMS);
    const int n = (int)bounded.size();
    int rc;
    const int32_t* d_lin = nullptr;
    if ((rc = upload(ds, unbounded, &d_lin, be->stream)) != RRT_OK) return rc;
    const int32_t* d_slot = nullptr;
    const rrt::WideNode* d_nodes = nullptr;
    int depth = 0;
    if (n == 1) {
        if ((rc = upload(ds, bounded, &d_slot, be->stream)) != RRT_OK) return rc;
    } else if (n >= 2) {
        void *nodes = nullptr, *slots = nullptr;
        HIP_TRY(hipMalloc(&nodes, sizeof(rrt::WideNode) * (size_t)(n - 1)));
        ds->allocs.push_back(nodes);
        HIP_TRY(hipMalloc(&slots, sizeof(int32_t) * (size_t)n));
        ds->allocs.push_back(slots);
        std::vector<int32_t> map;  // sorted slot -> index among the bounded primitives
        // The host builder: threads and vectors; nothing may unwind through the C ABI.
        auto host_tree = [&](bool median) -> int {
            try {
                std::vector<rrt::BvhNode> binary;
                int threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
                try {
                    rrt::SahBuilder(lo.data(), hi.data(), n, kBvhLeafMax).build(binary, map, threads, median);
                } catch (const std::system_error&) {  // no more threads to be had (a container's pid limit): alone, then
                    rrt::SahBuilder(lo.data(), hi.data(), n, kBvhLeafMax).build(binary, map, 1, median);
                }
                HIP_TRY(rrt::bvh_widen_host_tree(binary.data(), (int)binary.size(), (rrt::WideNode*)nodes, be->stream, &depth));
            } catch (const std::bad_alloc&) {
                return fail(RRT_ERR_NOMEM, "out of host memory while building the tree of %d primitives", n);
            } catch (const std::exception& e) {
                return fail(RRT_ERR_HIP, "the tree build failed: %s", e.what());
            }
            return RRT_OK;
        };
        bool on_host = kind == 1;
        if (kind == 1) {
            if ((rc = host_tree(false)) != RRT_OK) return rc;
        } else if (kind == 2) {
            // the builder's fourth cut coordinate, as SahBuilder::build computes it (so that both bin the same numbers)
            std::vector<float> size((size_t)n);
            in_shares((size_t)n, [&](int, size_t b0, size_t e0) {
                for (size_t i = b0; i < e0; ++i) {
                    rrt::SahBox b;
                    b.clear();
                    b.grow(lo.data() + 3 * i, hi.data() + 3 * i);
                    size[i] = std::log2(std::max(b.half_area(), 1e-30f));
                }
            });
            if (print_times) fprintf(stderr, "[rrt] tree build: cut coordinates at %.1f ms\n", since());
            DevBuf dlo, dhi, dsize;
            HIP_TRY(dlo.alloc(sizeof(float) * 3 * n));
            HIP_TRY(dhi.alloc(sizeof(float) * 3 * n));
            HIP_TRY(dsize.alloc(sizeof(float) * n));
            HIP_TRY(hipMemcpyAsync(dlo.p, lo.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice, be->stream));
            HIP_TRY(hipMemcpyAsync(dhi.p, hi.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice, be->stream));
            HIP_TRY(hipMemcpyAsync(dsize.p, size.data(), sizeof(float) * n, hipMemcpyHostToDevice, be->stream));
            hipError_t built = rrt::gsah_tree((const float*)dlo.p, (const float*)dhi.p, (const float*)dsize.p, n, (rrt::WideNode*)nodes, (int*)slots, be->stream, &depth);
            static const bool debug_fail = getenv("RRT_DEBUG_GSAH_FAIL") != nullptr;  // tests only: take the path below
            if (debug_fail) built = hipErrorUnknown;
            if (built == hipErrorOutOfMemory) HIP_TRY(built);
            if (built != hipSuccess) {
                // The level-by-level build gives up on a tree thousands of levels deep (ranges that lose one primitive per
                // cut): the host builder makes the same tree, and the depth check below then replaces it by median cuts.
                (void)hipGetLastError();
                if ((rc = host_tree(false)) != RRT_OK) return rc;
                on_host = true;
            }
        } else {
            DevBuf dlo, dhi;
            HIP_TRY(dlo.alloc(sizeof(float) * 3 * n));
            HIP_TRY(dhi.alloc(sizeof(float) * 3 * n));
            HIP_TRY(hipMemcpyAsync(dlo.p, lo.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice, be->stream));
            HIP_TRY(hipMemcpyAsync(dhi.p, hi.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice, be->stream));
            float flo[3], fhi[3];
            for (int k = 0; k < 3; ++k) { flo[k] = round_down(slo[k]); fhi[k] = round_up(shi[k]); }
            HIP_TRY(rrt::lbvh_build((const float*)dlo.p, (const float*)dhi.p, n, flo, fhi, (rrt::WideNode*)nodes, (int*)slots, be->stream, &depth));
        }
        // RRT_DEBUG_MAX_TREE_DEPTH: tests only (a small limit sends an ordinary scene through the fallback)
        static const int debug_depth = [] { const char* e = getenv("RRT_DEBUG_MAX_TREE_DEPTH"); return e ? atoi(e) : 0; }();
        if (3 * depth > kTreeStackEntries || (debug_depth > 0 && depth > debug_depth)) {  // too deep for the traversal stacks: median cuts instead
            if ((rc = host_tree(true)) != RRT_OK) return rc;
            on_host = true;
            if (3 * depth > kTreeStackEntries)
                return fail(RRT_ERR_UNSUPPORTED, "the tree of %d primitives is %d levels deep; the traversal stacks hold %d", n, depth, kTreeStackEntries / 3);
        }
        if (print_times) { (void)hipStreamSynchronize(be->stream); fprintf(stderr, "[rrt] tree build: tree and wide nodes at %.1f ms\n", since()); }
        if (!on_host) {
            map.resize((size_t)n);
            HIP_TRY(hipMemcpy(map.data(), slots, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
        }
        // -> scan-order index
        for (int i = 0; i < n; ++i) {
            if (map[i] < 0 || map[i] >= n) return fail(RRT_ERR_HIP, "the tree build produced an invalid slot table");
            map[i] = bounded[(size_t)map[i]];
        }
        HIP_TRY(hipMemcpy(slots, map.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
        d_nodes = (const rrt::WideNode*)nodes;
        d_slot = (const int32_t*)slots;
    }
    DeviceScene::Tree& tree = ds->tree[kind];
    if (n >= 1) {  // leaf records in sorted-slot order, both precisions
        void *l32 = nullptr, *l64 = nullptr;
        HIP_TRY(hipMalloc(&l32, sizeof(rrt::LeafRec<float>) * (size_t)n));
        ds->allocs.push_back(l32);
        HIP_TRY(hipMalloc(&l64, sizeof(rrt::LeafRec<double>) * (size_t)n));
        ds->allocs.push_back(l64);
        HIP_TRY(rrt::lbvh_pack_leaves(ds->v32.prims, d_slot, n, (rrt::LeafRec<float>*)l32, be->stream));
        HIP_TRY(rrt::lbvh_pack_leaves(ds->v64.prims, d_slot, n, (rrt::LeafRec<double>*)l64, be->stream));
        tree.leaves32 = (const rrt::LeafRec<float>*)l32;
        tree.leaves64 = (const rrt::LeafRec<double>*)l64;
    }
    HIP_TRY(hipStreamSynchronize(be->stream));
    if (print_times) fprintf(stderr, "[rrt] tree build: leaf records at %.1f ms\n", since());
    tree.nodes = d_nodes;
    tree.bvh_prim = d_slot;
    tree.linear_prim = d_lin;
    tree.n_bvh_prims = n;
    tree.n_linear_prims = (int32_t)unbounded.size();
    tree.depth = depth;
    tree.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    tree.built = true;
    return RRT_OK;
}

// Which tree an rrt_accel value means: -1 none (linear scan), 0 cluster tree (GPU), 1 surface-area tree (host), 2 surface-area tree (GPU).
int tree_kind(const rrt_scene* sc, uint32_t accel) {
    if (accel == RRT_ACCEL_LBVH) return 0;
    if (accel == RRT_ACCEL_SAH) return 1;
    if (accel == RRT_ACCEL_SAH_GPU) return 2;
    return accel == RRT_ACCEL_AUTO && sc->prims.size() > kAutoBvhThreshold ? 2 : -1;
}
const uint32_t kAccelOfKind[3] = {RRT_ACCEL_LBVH, RRT_ACCEL_SAH, RRT_ACCEL_SAH_GPU};

// AUTO: the surface-area tree for scenes too large for the wave-uniform scan; LBVH, SAH: that tree; LINEAR: none.
// *kind = the tree the call walks (-1: none); *built_now = whether this call built it.
int resolve_accel(rrt_backend* be, const rrt_scene* sc, DeviceScene* ds, uint32_t accel, int* kind, bool* built_now = nullptr) {
    *kind = tree_kind(sc, accel);
    if (built_now) *built_now = false;
    return *kind >= 0 ? ensure_bvh(be, sc, ds, *kind, built_now) : RRT_OK;
}

// How a caller's buffer is used.  Kernels write in place only into plain device memory of the backend's own GPU:
// the render kernels add with unsafe FP atomics, which are not guaranteed to take effect on managed or fine-grained
// memory that may live on the host.  Managed memory and plain host memory are staged (both are host-readable);
// device memory of ANOTHER GPU is refused.
enum class PtrKind { Host, Device, OtherDevice };
PtrKind classify_pointer(const rrt_backend* be, const void* p) {
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory is reported as an error; clear it
        return PtrKind::Host;
    }
    if (attr.type == hipMemoryTypeDevice) return attr.device == be->device ? PtrKind::Device : PtrKind::OtherDevice;
    return PtrKind::Host;  // pinned host, managed, unregistered
}

int check_backend(rrt_backend* be) {
    if (!be) return fail(RRT_ERR_INVALID, "backend is NULL");
    HIP_TRY(hipSetDevice(be->device));
    return RRT_OK;
}

int closest_hit_call(rrt_backend* be, const rrt_scene* scene, uint32_t precision, uint32_t accel, bool primary,
                     size_t n, const double* a, const double* b, int32_t* object_id, int32_t* prim_id, double* t) {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!scene) return fail(RRT_ERR_INVALID, "scene is NULL");
    if (precision > RRT_PRECISION_F64) return fail(RRT_ERR_INVALID, "unknown precision %u", precision);
    if (accel > RRT_ACCEL_SAH_GPU) return fail(RRT_ERR_INVALID, "unknown accel %u", accel);
    if (n == 0) return RRT_OK;
    if (!a || !b || !object_id || !prim_id || !t) return fail(RRT_ERR_INVALID, "NULL array argument");
    if (scene->prims.empty()) {
        for (size_t i = 0; i < n; ++i) {
            object_id[i] = -1;
            prim_id[i] = -1;
            t[i] = INFINITY;
        }
        return RRT_OK;
    }
    DeviceScene* ds = nullptr;
    if ((rc = device_scene(be, scene, &ds)) != RRT_OK) return rc;
    int kind = -1;
    if ((rc = resolve_accel(be, scene, ds, accel, &kind)) != RRT_OK) return rc;
    const bool use_bvh = kind >= 0;
    const size_t per = primary ? 1 : 3;
    DevBuf da, db, doid, dpid, dt;
    HIP_TRY(da.alloc(n * per * sizeof(double)));
    HIP_TRY(db.alloc(n * per * sizeof(double)));
    HIP_TRY(doid.alloc(n * sizeof(int32_t)));
    HIP_TRY(dpid.alloc(n * sizeof(int32_t)));
    HIP_TRY(dt.alloc(n * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(da.p, a, n * per * sizeof(double), hipMemcpyHostToDevice, be->stream));
    HIP_TRY(hipMemcpyAsync(db.p, b, n * per * sizeof(double), hipMemcpyHostToDevice, be->stream));
    if (precision == RRT_PRECISION_F64)
        HIP_TRY(rrt::launch_intersect(with_tree(be, ds, ds->v64, kind), use_bvh, primary, n, (const double*)da.p, (const double*)db.p,
                                      (int32_t*)doid.p, (int32_t*)dpid.p, (double*)dt.p, be->stream));
    else
        HIP_TRY(rrt::launch_intersect(with_tree(be, ds, ds->v32, kind), with_tree(be, ds, ds->v64, kind), ds->err32, use_bvh, primary, n, (const double*)da.p, (const double*)db.p,
                                      (int32_t*)doid.p, (int32_t*)dpid.p, (double*)dt.p, be->stream));
    HIP_TRY(hipMemcpyAsync(object_id, doid.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipMemcpyAsync(prim_id, dpid.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipMemcpyAsync(t, dt.p, n * sizeof(double), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipStreamSynchronize(be->stream));
    return RRT_OK;
}

uint32_t next_pow2_mask(uint32_t l) {
    uint32_t w = l - 1;
    w |= w >> 1; w |= w >> 2; w |= w >> 4; w |= w >> 8; w |= w >> 16;
    return w;
}

int validate_render_desc(const rrt_render_desc* d) {
    if (!d) return fail(RRT_ERR_INVALID, "render desc is NULL");
    if (d->width == 0 || d->height == 0) return fail(RRT_ERR_INVALID, "image size %ux%u", d->width, d->height);
    if ((uint64_t)d->width * d->height > (1ull << 28)) return fail(RRT_ERR_INVALID, "image too large");
    if (d->sampler_type > RRT_SAMPLER_STRATIFIED) return fail(RRT_ERR_INVALID, "unknown sampler type %u", d->sampler_type);
    if (d->spp == 0) return fail(RRT_ERR_INVALID, "samples_per_pixel must be positive");
    if (d->sampler_type == RRT_SAMPLER_STRATIFIED) {
        // assert_eq!(x_strata * y_strata, samples_per_pixel), src/sampler.rs:87-91
        if ((uint64_t)d->x_strata * d->y_strata != d->spp)
            return fail(RRT_ERR_INVALID, "x_strata * y_strata != samples_per_pixel");
        if (d->sample_end > d->spp)
            return fail(RRT_ERR_INVALID, "stratified sample index range ends at %llu > spp %u",
                        (unsigned long long)d->sample_end, d->spp);
    }
    if (d->sample_end <= d->sample_begin) return fail(RRT_ERR_INVALID, "empty sample range");
    if (d->sample_end > 0x7fffffffull) return fail(RRT_ERR_INVALID, "sample index beyond 2^31");
    if (d->precision > RRT_PRECISION_F64) return fail(RRT_ERR_INVALID, "unknown precision %u", d->precision);
    if (d->accel > RRT_ACCEL_SAH_GPU) return fail(RRT_ERR_INVALID, "unknown accel %u", d->accel);
    if (d->pipeline > RRT_PIPELINE_WAVEFRONT) return fail(RRT_ERR_INVALID, "unknown pipeline %u", d->pipeline);
    if (d->flags & ~(uint32_t)(RRT_RENDER_NEE | RRT_RENDER_EXACT_HITS)) return fail(RRT_ERR_INVALID, "unknown render flags 0x%x", d->flags);
    if ((d->flags & RRT_RENDER_EXACT_HITS) && d->pipeline == RRT_PIPELINE_WAVEFRONT)
        return fail(RRT_ERR_UNSUPPORTED, "RRT_RENDER_EXACT_HITS runs in the fused pipeline only");
    if ((d->flags & RRT_RENDER_EXACT_HITS) && (d->flags & RRT_RENDER_NEE))
        return fail(RRT_ERR_UNSUPPORTED, "RRT_RENDER_EXACT_HITS and RRT_RENDER_NEE cannot be combined");
    return RRT_OK;
}

// Kernel-side view of a validated render description.
// lanes: paths the device traces at once in the fused kernel (blocks x 256)
rrt::RenderParams make_render_params(const rrt_render_desc* desc, uint64_t lanes) {
    const size_t n_pixels = (size_t)desc->width * desc->height;
    rrt::RenderParams rp;
    std::memset(&rp, 0, sizeof(rp));
    rp.width = desc->width;
    rp.height = desc->height;
    rp.sampler_type = desc->sampler_type;
    rp.spp = desc->spp;
    rp.x_strata = desc->x_strata ? desc->x_strata : 1;
    rp.y_strata = desc->y_strata ? desc->y_strata : 1;
    rp.x_shift = 0xffffffffu;
    if ((rp.x_strata & (rp.x_strata - 1)) == 0) {
        rp.x_shift = 0;
        while ((1u << rp.x_shift) < rp.x_strata) rp.x_shift++;
    }
    rp.perm_mask = next_pow2_mask(desc->spp);
    rp.flags = ((desc->flags & RRT_RENDER_NEE) ? rrt::RENDER_FLAG_NEE : 0u) | ((desc->flags & RRT_RENDER_EXACT_HITS) ? rrt::RENDER_FLAG_EXACT : 0u);
    rp.min_depth = desc->min_depth;
    rp.max_depth = desc->max_depth;
    rp.sample_begin = (uint32_t)desc->sample_begin;
    rp.n_samples = (uint32_t)(desc->sample_end - desc->sample_begin);
    // samples per work item: enough items to balance the tail, few enough to keep
    // the per-item atomics and the 32-bit item index negligible / valid
    // 128 (96-160 measure the same at 16384 spp; 32 costs 2.3 % in bookkeeping, 2048 costs 15 %: four items per lane).
    // A call with few samples per lane -- one GPU's share of a frame split over eight -- takes 64, so that every lane
    // still gets ~15 items or more: 2025 spp of the Cornell box 18.8 -> 19.3 Gsamples/s, 4096 spp 20.3 -> 20.6.
    uint32_t target = 128;
    if ((uint64_t)((rp.n_samples + 127) / 128) * n_pixels < 24 * lanes) target = 64;
    if (const char* e = getenv("RRT_DEBUG_CHUNK")) target = (uint32_t)atoi(e) > 0 ? (uint32_t)atoi(e) : 128;  // experiments
    uint32_t chunk = rp.n_samples < target ? rp.n_samples : target;
    while ((uint64_t)((rp.n_samples + chunk - 1) / chunk) * n_pixels >= (1ull << 31)) chunk *= 2;
    // The last samples of every pixel go out in items a quarter as long (chunk >> small_shift), about one and a half big
    // items' worth per lane (tail_halves / 2): the kernel ends when the slowest lane finishes its last item, and a launch
    // is over sooner when those are short.  Cornell box, one GPU's share of the frame: 2048 spp 25.19 -> 24.73 ms,
    // 1024 spp 13.19 -> 12.67 ms, 16384 spp 190.9 -> 190.4 ms (halves: 24.83, eighths: 24.80, two items' worth: 24.80).
    // (It only pays since the item bookkeeping is cheap: with 32-bit divisions in it the same change lost time.)
    static const int small_shift = [] { const char* e = getenv("RRT_DEBUG_SMALL_SHIFT"); return e ? atoi(e) : 2; }();  // experiments
    static const int tail_halves = [] { const char* e = getenv("RRT_DEBUG_TAIL_ITEMS"); return e ? atoi(e) : 3; }();
    // (never shorter than eight samples: every 64 items cost a wave one atomic on the work counter, and with items of one
    //  or two samples -- a preview at 4 or 9 spp -- the kernel waited for that counter: 500x500 @ 4 spp 0.32 -> 0.15 ms,
    //  1920x1080 @ 4 spp 1.84 -> 0.63 ms)
    uint32_t small = (chunk >> small_shift) >= 2 ? chunk >> small_shift : 1;
    if (small < 8u) small = chunk < 8u ? chunk : 8u;
    uint64_t tail_samples = n_pixels ? ((uint64_t)tail_halves * chunk * lanes / 2 + n_pixels - 1) / n_pixels : 0;  // per pixel
    if (small_shift == 0) tail_samples = 0;
    if (tail_samples > rp.n_samples) tail_samples = rp.n_samples;
    uint32_t big_chunks = (uint32_t)((rp.n_samples - tail_samples + (tail_samples ? 0 : chunk - 1)) / chunk);
    uint32_t rest = rp.n_samples > big_chunks * chunk ? rp.n_samples - big_chunks * chunk : 0;
    uint32_t small_chunks = (rest + small - 1) / small;
    if ((uint64_t)(big_chunks + small_chunks) * n_pixels >= (1ull << 31)) {
        big_chunks = (rp.n_samples + chunk - 1) / chunk;
        small_chunks = 0;
    }
    rp.chunk = chunk;
    rp.small_chunk = small;
    rp.n_big_chunks = big_chunks;
    rp.n_items = (uint64_t)(big_chunks + small_chunks) * n_pixels;
    auto magic = [](uint32_t d, uint32_t& m, uint32_t& sh) {
        if (d <= 1) { m = 0; sh = 0xffffffffu; return; }
        uint32_t L = 0;
        while ((1ull << L) < d) ++L;                       // ceil(log2 d) >= 1
        m = (uint32_t)((((unsigned long long)1 << (31 + L)) + d - 1) / d);  // ceil(2^(31 + L) / d) <= 2^32 - 1 ... 2^31
        sh = L - 1;
    };
    magic((uint32_t)n_pixels, rp.div_pixels_m, rp.div_pixels_s);
    magic(rp.width, rp.div_width_m, rp.div_width_s);
    rp.inv_width = 1.0f / (float)rp.width;
    rp.inv_height = 1.0f / (float)rp.height;
    rp.inv_spp = 1.0f / (float)rp.spp;
    rp.inv_x_strata = 1.0f / (float)rp.x_strata;
    rp.inv_y_strata = 1.0f / (float)rp.y_strata;
    rp.seed_lo = (uint32_t)desc->seed;
    rp.seed_hi = (uint32_t)(desc->seed >> 32);
    return rp;
}

// ------------------------------------------------------------------ wavefront pipeline
// Paths per queue (56 bytes per path and queue).  Every trip costs one or two kernel launches with
// their ramp-up and tail, so both pipelines want many rays per lane and trip.  Linear-scan scenes: 16 Mi paths
// measured best (Cornell box through the one-kernel-per-trip pipeline: 7.8 Gsamples/s at 1 Mi, 10.4 at 4 Mi,
// 11.2 at 16 Mi).  Tree scenes: the persistent traversal kernel ends every trip with waves that drain their last
// block of 64 rays at a fifth of their lanes, ~100 loop trips out of the ~700 a wave makes per 16 Mi rays; with 64 Mi
// paths (7.5 GB of the 288 GB) that tail is a quarter as large: 5.15 -> 5.60 Grays/s on the million-triangle scene
// (128 Mi and 256 Mi: 5.62, 5.63 then).  With the traversal kernel a third faster since, the fixed costs of a trip weigh
// more: 64 Mi -> 7.50, 128 Mi -> 7.60, 256 Mi -> 7.53 Grays/s; 128 Mi paths are 15 GB of the 288 GB.  A device with less
// free memory gets smaller queues.
const uint32_t kWfCapacityLinear = 1u << 24;
const uint32_t kWfCapacityTree = 1u << 27;
const uint32_t kWfCapacityFloor = 1u << 20;

size_t wavefront_bytes(size_t n, bool shadow, size_t elem) {
    const size_t per_queue = n * (12 * elem + sizeof(uint2));  // two 4-vectors, two 2-vectors, the packed sampler words
    const size_t cursors = (n / 64) * sizeof(rrt::WfWaveCursor);
    const size_t rows = (n / rrt::WF_HIT_BLOCK) * rrt::COUNTER_COUNT * sizeof(unsigned long long);
    return (shadow ? 3 : 2) * per_queue + cursors + rows + 256 + 18 * 256;  // every array is padded to 256 bytes
}

// The arrays of the queues inside the backend's one allocation.
template <typename R>
rrt::WfParams<R> wavefront_params(const rrt_backend* be) {
    typedef typename rrt::WfElem<R>::V4 V4;
    typedef typename rrt::WfElem<R>::V2 V2;
    const size_t n = be->wf_capacity;
    char* p = (char*)be->wf_mem;
    auto take = [&p](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    // capacity is a multiple of 1024, so every array below is a multiple of 256 bytes long except
    // the control block at the end, which the spare 256 bytes cover
    rrt::WfParams<R> wp;
    std::memset(&wp, 0, sizeof(wp));
    auto carve = [&](rrt::WfQueue<R>& q) {
        q.o_dx = (V4*)take(n * sizeof(V4));
        q.dyz = (V2*)take(n * sizeof(V2));
        q.hit = (V2*)take(n * sizeof(V2));
        q.beta_pix = (V4*)take(n * sizeof(V4));
        q.smp = (uint2*)take(n * sizeof(uint2));
    };
    carve(wp.q[0]);
    carve(wp.q[1]);
    if (be->wf_has_shadow) carve(wp.sq);
    wp.cursors = (rrt::WfWaveCursor*)take((n / 64) * sizeof(rrt::WfWaveCursor));
    wp.block_counters = (unsigned long long*)take((n / rrt::WF_HIT_BLOCK) * rrt::COUNTER_COUNT * sizeof(unsigned long long));
    wp.ctl = (rrt::WfControl*)take(sizeof(rrt::WfControl));
    wp.capacity = be->wf_capacity;
    return wp;
}

// shadow: with the queue of shadow rays (next-event estimation through the tree)
int ensure_wavefront_buffers(rrt_backend* be, uint32_t capacity, bool shadow, size_t elem) {
    if (be->wf_mem && be->wf_capacity == capacity && be->wf_elem == elem && (be->wf_has_shadow || !shadow)) return RRT_OK;
    if (be->wf_mem) {
        HIP_TRY(hipStreamSynchronize(be->stream));
        HIP_TRY(hipFree(be->wf_mem));
        be->wf_mem = nullptr;
        be->wf_capacity = 0;
        be->wf_has_shadow = false;
        be->wf_elem = 0;
    }
    HIP_TRY(hipMalloc(&be->wf_mem, wavefront_bytes(capacity, shadow, elem)));
    be->wf_capacity = capacity;
    be->wf_has_shadow = shadow;
    be->wf_elem = elem;
    return RRT_OK;
}

// Runs the trips of the wavefront pipeline until every sample of the call is traced; fills
// `counters` like the fused kernel does.
template <typename R>
int render_wavefront(rrt_backend* be, const rrt::SceneView<R>& sv, const rrt::LightView<R>& lights, bool use_bvh, rrt::RenderParams rp,
                     unsigned long long* counters) {
    const bool nee = (rp.flags & rrt::RENDER_FLAG_NEE) != 0;
    const bool shadow_queue = nee && use_bvh;  // (the one-kernel-per-trip pipeline of scan scenes traces shadow rays on the spot)
    if (rp.max_depth > rrt::WF_MAX_DEPTH)
        return fail(RRT_ERR_UNSUPPORTED, "the wavefront pipeline packs the path depth into 9 bits: max_depth %u > %u",
                    rp.max_depth, rrt::WF_MAX_DEPTH);
    const uint64_t n_pixels = (uint64_t)rp.width * rp.height;
    const uint64_t total = n_pixels * rp.n_samples;
    uint32_t capacity = use_bvh ? kWfCapacityTree : kWfCapacityLinear;
    if (const char* e = getenv("RRT_DEBUG_WF_CAPACITY")) {  // experiments and tests
        long v = atol(e);
        if (v >= 64 && v <= (1l << 28)) capacity = (uint32_t)v;
    }
    const uint32_t block = (uint32_t)rrt::WF_SHADE_BLOCK;
    if ((uint64_t)capacity > total) capacity = (uint32_t)total;
    {   // the queues must leave the caller at least half of what is free (counting the queues this backend already holds)
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (be->wf_mem) free_b += wavefront_bytes(be->wf_capacity, be->wf_has_shadow, be->wf_elem);
        while (capacity > kWfCapacityFloor && wavefront_bytes(capacity, shadow_queue, sizeof(R)) > free_b / 2) capacity /= 2;
    }
    capacity = (capacity + block - 1) / block * block;
    int rc = ensure_wavefront_buffers(be, capacity, shadow_queue, sizeof(R));
    if (rc != RRT_OK) return rc;
    rrt::WfParams<R> wp = wavefront_params<R>(be);
    wp.stride_s = (uint32_t)(capacity / n_pixels);
    wp.stride_p = (uint32_t)(capacity % n_pixels);
    const size_t rows_bytes = (size_t)(capacity / rrt::WF_HIT_BLOCK) * rrt::COUNTER_COUNT * sizeof(unsigned long long);
    HIP_TRY(hipMemsetAsync(wp.block_counters, 0, rows_bytes, be->stream));
    HIP_TRY(rrt::wf_launch_init(rp, wp, be->stream));

    // A trip starts at most `capacity` samples, so the first total / capacity trips cannot be the
    // last; after that the live count is read back every few trips (an idle trip costs two
    // near-empty launches).
    const uint64_t sure = total / capacity;
    const uint64_t check_every = 4;
    rrt::WfControl ctl;
    if (!use_bvh) {
        // linear-scan scenes: one kernel per trip (closest hit fused into the shade kernel); the last trip's
        // output count is count[(trip + 1) % 3]
        for (uint64_t trip = 0;; ++trip) {
            HIP_TRY(rrt::wf_launch_shade_scan(sv, lights, rp, wp, trip, be->stream));
            if (trip >= sure && (trip - sure) % check_every == 0) {
                HIP_TRY(hipMemcpyAsync(&ctl, wp.ctl, sizeof(ctl), hipMemcpyDeviceToHost, be->stream));
                HIP_TRY(hipStreamSynchronize(be->stream));
                if (ctl.count[(trip + 1) % 3] == 0) break;
            }
        }
    } else {
        int parity = 0;
        for (uint64_t trip = 0;; ++trip) {
            HIP_TRY(rrt::wf_launch_shade(sv, lights, rp, wp, parity, be->stream));
            parity ^= 1;
            HIP_TRY(rrt::wf_launch_closest_hit(sv, use_bvh, wp, parity, false, be->n_cu, be->stream));
            if (shadow_queue) {  // the shadow rays that shade kernel queued: their own closest-hit launch, then what they carry
                HIP_TRY(rrt::wf_launch_closest_hit(sv, use_bvh, wp, parity, true, be->n_cu, be->stream));
                HIP_TRY(rrt::wf_launch_shadow_resolve(rp, wp, parity, be->stream));
            }
            if (trip >= sure && (trip - sure) % check_every == 0) {
                HIP_TRY(hipMemcpyAsync(&ctl, wp.ctl, sizeof(ctl), hipMemcpyDeviceToHost, be->stream));
                HIP_TRY(hipStreamSynchronize(be->stream));
                if (ctl.count[parity] == 0) break;
            }
        }
    }
    std::vector<unsigned long long> rows(rows_bytes / sizeof(unsigned long long));
    HIP_TRY(hipMemcpyAsync(rows.data(), wp.block_counters, rows_bytes, hipMemcpyDeviceToHost, be->stream));
    unsigned long long dropped = 0;  // (the tree walks report through the view's own counter word, not through the rows)
    HIP_TRY(hipMemcpyAsync(&dropped, be->d_counters + rrt::COUNTER_STACK_OVERFLOW, sizeof(dropped), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipStreamSynchronize(be->stream));
    for (int k = 0; k < rrt::COUNTER_COUNT; ++k) counters[k] = 0;
    for (size_t i = 0; i < rows.size(); ++i) counters[i % rrt::COUNTER_COUNT] += rows[i];
    counters[rrt::COUNTER_STACK_OVERFLOW] += dropped;
    counters[rrt::COUNTER_RAYS] += ctl.rays;  // (the rows hold the shadow rays the one-kernel-per-trip pipeline traced on the spot)
    if (!use_bvh) counters[rrt::COUNTER_PRIM_TESTS] = counters[rrt::COUNTER_RAYS] * (unsigned long long)sv.n_prims;
    return RRT_OK;
}

// What an exception caught at the C ABI turns into (called from a catch block).
int exception_status() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        return fail(RRT_ERR_NOMEM, "out of host memory");
    } catch (const std::exception& e) {
        return fail(RRT_ERR_HIP, "internal error: %s", e.what());
    } catch (...) {
        return fail(RRT_ERR_HIP, "internal error");
    }
}

}  // namespace

extern "C" {

const char* rrt_last_error(void) { return g_last_error.c_str(); }

int rrt_camera_perspective(const double look_from[3], const double look_at[3], const double vup[3],
                           double vfov_deg, double aspect, rrt_camera* out) try {
    if (!look_from || !look_at || !vup || !out) return fail(RRT_ERR_INVALID, "NULL argument");
    const double PI = 3.14159265358979323846264338327950288;
    double theta = vfov_deg * PI / 180.0;
    double half_height = std::tan(theta / 2.0);
    double half_width = aspect * half_height;
    D3 from = d3(look_from);
    D3 w = normalize(from - d3(look_at));
    D3 u = normalize(cross(d3(vup), w));
    D3 v = cross(w, u);
    D3 ll = from - u * half_width - v * half_height - w;
    D3 h = u * (2.0 * half_width), ve = v * (2.0 * half_height);
    out->origin[0] = from.x; out->origin[1] = from.y; out->origin[2] = from.z;
    out->lower_left_corner[0] = ll.x; out->lower_left_corner[1] = ll.y; out->lower_left_corner[2] = ll.z;
    out->horizontal[0] = h.x; out->horizontal[1] = h.y; out->horizontal[2] = h.z;
    out->vertical[0] = ve.x; out->vertical[1] = ve.y; out->vertical[2] = ve.z;
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_scene_create(const rrt_scene_desc* desc, rrt_scene** out) try {
    if (!desc || !out) return fail(RRT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (desc->n_prims && !desc->prims) return fail(RRT_ERR_INVALID, "prims is NULL");
    if (desc->n_objects && !desc->materials) return fail(RRT_ERR_INVALID, "materials is NULL");
    if (desc->n_prims > 0x7fffffffu) return fail(RRT_ERR_INVALID, "too many primitives");
    for (uint32_t i = 0; i < desc->n_prims; ++i) {
        const rrt_prim& p = desc->prims[i];
        if (p.type > RRT_PRIM_QUAD) return fail(RRT_ERR_INVALID, "prim %u: unknown type %u", i, p.type);
        if (p.object_id >= desc->n_objects)
            return fail(RRT_ERR_INVALID, "prim %u: object_id %u out of range", i, p.object_id);
        static const int used[4] = {4, 6, 9, 12};
        for (int k = 0; k < used[p.type]; ++k)
            if (!std::isfinite(p.v[k])) return fail(RRT_ERR_INVALID, "prim %u: non-finite coordinate", i);
    }
    for (uint32_t i = 0; i < desc->n_objects; ++i)
        if (desc->materials[i].type > RRT_MAT_DIELECTRIC)
            return fail(RRT_ERR_INVALID, "object %u: unknown material type %u", i, desc->materials[i].type);
    rrt_scene* sc = new (std::nothrow) rrt_scene();
    if (!sc) return fail(RRT_ERR_NOMEM, "out of host memory");
    sc->camera = desc->camera;
    sc->prims.assign(desc->prims, desc->prims + desc->n_prims);
    sc->materials.assign(desc->materials, desc->materials + desc->n_objects);
    *out = sc;
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

void rrt_scene_destroy(rrt_scene* scene) {
    if (!scene) return;
    for (auto& kv : scene->dev) delete kv.second;
    delete scene;
}

int rrt_backend_create(const char* name, int device_id, rrt_backend** out) try {
    if (!name || !out) return fail(RRT_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (std::strcmp(name, "hip") != 0)
        return fail(RRT_ERR_UNSUPPORTED, "unknown backend \"%s\" (this library only has \"hip\")", name);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(RRT_ERR_NO_DEVICE, "no HIP device is visible");
    }
    if (device_id < 0 || device_id >= count)
        return fail(RRT_ERR_NO_DEVICE, "device %d out of range (%d visible)", device_id, count);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RRT_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only", device_id,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));
    rrt_backend* be = new (std::nothrow) rrt_backend();
    if (!be) return fail(RRT_ERR_NOMEM, "out of host memory");
    be->device = device_id;
    be->n_cu = prop.multiProcessorCount;
    hipError_t e = hipSuccess;
    if (device_id < kMaxDevices) {  // (the device is current: hipSetDevice above)
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_id]) {
            void* ptr = nullptr;
            e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
            if (e == hipSuccess) g_scratch_ptr[device_id] = (StaticRays*)ptr;
        }
    }
    if (e == hipSuccess) e = hipEventCreate(&be->ev0);
    if (e == hipSuccess) e = hipEventCreate(&be->ev1);
    if (e == hipSuccess) e = hipMalloc((void**)&be->d_counters, sizeof(unsigned long long) * (rrt::COUNTER_ROWS * rrt::COUNTER_COUNT + 1));
    if (e != hipSuccess) {
        rrt_backend_destroy(be);
        return fail(RRT_ERR_HIP, "backend set-up failed: %s", hipGetErrorString(e));
    }
    *out = be;
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

void rrt_backend_destroy(rrt_backend* be) {
    if (!be) return;
    (void)hipSetDevice(be->device);
    if (be->ev0) (void)hipEventDestroy(be->ev0);
    if (be->ev1) (void)hipEventDestroy(be->ev1);
    if (be->d_counters) (void)hipFree(be->d_counters);
    if (be->wf_mem) (void)hipFree(be->wf_mem);
    delete be;
}

int rrt_backend_set_stream(rrt_backend* be, void* hip_stream) try {
    if (!be) return fail(RRT_ERR_INVALID, "backend is NULL");
    be->stream = (hipStream_t)hip_stream;
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_intersect(rrt_backend* be, const rrt_scene* scene, uint32_t precision, uint32_t accel, size_t n,
                  const double* origins, const double* dirs, int32_t* object_id, int32_t* prim_id, double* t) try {
    return closest_hit_call(be, scene, precision, accel, false, n, origins, dirs, object_id, prim_id, t);
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_trace_primary(rrt_backend* be, const rrt_scene* scene, uint32_t precision, uint32_t accel, size_t n,
                      const double* u, const double* v, int32_t* object_id, int32_t* prim_id, double* t) try {
    return closest_hit_call(be, scene, precision, accel, true, n, u, v, object_id, prim_id, t);
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_intersect_static(rrt_backend* be, const rrt_scene* scene, uint32_t precision, uint32_t accel, size_t n,
                         const double* origins, const double* dirs, int32_t* object_id, int32_t* prim_id, double* t) try {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!scene) return fail(RRT_ERR_INVALID, "scene is NULL");
    if (precision > RRT_PRECISION_F64) return fail(RRT_ERR_INVALID, "unknown precision %u", precision);
    if (accel > RRT_ACCEL_SAH_GPU) return fail(RRT_ERR_INVALID, "unknown accel %u", accel);
    if (n > (size_t)RRT_INTERSECT_STATIC_MAX_RAYS)
        return fail(RRT_ERR_INVALID, "rrt_intersect_static takes at most %d rays per call, got %zu (split the batch, or use rrt_intersect)",
                    RRT_INTERSECT_STATIC_MAX_RAYS, n);
    if (n == 0) return RRT_OK;
    if (!origins || !dirs || !object_id || !prim_id || !t) return fail(RRT_ERR_INVALID, "NULL array argument");
    if (scene->prims.empty()) {
        for (size_t i = 0; i < n; ++i) {
            object_id[i] = -1;
            prim_id[i] = -1;
            t[i] = INFINITY;
        }
        return RRT_OK;
    }
    // nothing may be allocated here: the scene's tables (and its tree, if this query walks one) must already be on
    // the device, which any earlier rrt_intersect / rrt_trace_primary / rrt_render_accumulate call of the same kind does
    auto it = scene->dev.find(be->device);
    if (it == scene->dev.end())
        return fail(RRT_ERR_INVALID, "rrt_intersect_static: the scene is not resident on device %d yet; make one "
                                     "rrt_intersect call with the same scene and accel first", be->device);
    DeviceScene* ds = it->second;
    const int kind = tree_kind(scene, accel);
    const bool use_bvh = kind >= 0;
    if (use_bvh && !ds->tree[kind].built)
        return fail(RRT_ERR_INVALID, "rrt_intersect_static: the scene's tree is not built on device %d yet; make one "
                                     "rrt_intersect call with the same scene and accel first", be->device);
    StaticRays* sp = be->device < kMaxDevices ? g_scratch_ptr[be->device] : nullptr;
    if (!sp) return fail(RRT_ERR_UNSUPPORTED, "no static scratch on device %d", be->device);
    // every byte a kernel reads (3n + 3n doubles) is written here first and every result the caller gets was
    // written by this call's kernel, so what an earlier call left in the array never shows
    HIP_TRY(hipMemcpyAsync(sp->origins, origins, n * 3 * sizeof(double), hipMemcpyHostToDevice, be->stream));
    HIP_TRY(hipMemcpyAsync(sp->dirs, dirs, n * 3 * sizeof(double), hipMemcpyHostToDevice, be->stream));
    if (precision == RRT_PRECISION_F64)
        HIP_TRY(rrt::launch_intersect(with_tree(be, ds, ds->v64, kind), use_bvh, false, n, sp->origins, sp->dirs, sp->object_id, sp->prim_id, sp->t, be->stream));
    else
        HIP_TRY(rrt::launch_intersect(with_tree(be, ds, ds->v32, kind), with_tree(be, ds, ds->v64, kind), ds->err32, use_bvh, false, n, sp->origins, sp->dirs, sp->object_id, sp->prim_id, sp->t, be->stream));
    HIP_TRY(hipMemcpyAsync(object_id, sp->object_id, n * sizeof(int32_t), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipMemcpyAsync(prim_id, sp->prim_id, n * sizeof(int32_t), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipMemcpyAsync(t, sp->t, n * sizeof(double), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipStreamSynchronize(be->stream));
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_render_accumulate(rrt_backend* be, const rrt_scene* scene, const rrt_render_desc* desc,
                          float* rgb_sum, rrt_stats* stats) try {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!scene) return fail(RRT_ERR_INVALID, "scene is NULL");
    if (!rgb_sum) return fail(RRT_ERR_INVALID, "rgb_sum is NULL");
    if ((rc = validate_render_desc(desc)) != RRT_OK) return rc;
    DeviceScene* ds = nullptr;
    if ((rc = device_scene(be, scene, &ds)) != RRT_OK) return rc;
    int kind = -1;
    bool built_now = false;
    if ((rc = resolve_accel(be, scene, ds, desc->accel, &kind, &built_now)) != RRT_OK) return rc;
    const bool use_bvh = kind >= 0;
    // AUTO: tree traversal wants many waves in flight, which the register-lean closest-hit kernel of
    // the wavefront pipeline has; a scan over a few primitives is fastest with the path in registers
    const bool nee = (desc->flags & RRT_RENDER_EXACT_HITS) != 0;  // a flag that only the fused kernel serves
    if ((desc->flags & RRT_RENDER_NEE) && ds->n_unbounded_lights > 0)
        return fail(RRT_ERR_UNSUPPORTED, "next-event estimation cannot sample the scene's %d emissive plane(s)", ds->n_unbounded_lights);
    const bool wavefront = desc->pipeline == RRT_PIPELINE_WAVEFRONT ||
                           (desc->pipeline == RRT_PIPELINE_AUTO && !nee && use_bvh && scene->prims.size() > kAutoWavefrontThreshold &&
                            desc->precision == RRT_PRECISION_F32 && desc->max_depth <= rrt::WF_MAX_DEPTH);

    const size_t n_pixels = (size_t)desc->width * desc->height;
    const PtrKind sum_kind = classify_pointer(be, rgb_sum);
    if (sum_kind == PtrKind::OtherDevice) return fail(RRT_ERR_INVALID, "rgb_sum is device memory of another GPU than the backend's (%d)", be->device);
    const bool on_device = sum_kind == PtrKind::Device;
    DevBuf staging;
    float* d_accum = rgb_sum;
    if (!on_device) {
        HIP_TRY(staging.alloc(n_pixels * 3 * sizeof(float)));
        d_accum = (float*)staging.p;
        HIP_TRY(hipMemsetAsync(d_accum, 0, n_pixels * 3 * sizeof(float), be->stream));
    }

    rrt::RenderParams rp = make_render_params(desc, (uint64_t)be->n_cu * 8 * 256);
    rp.accum = d_accum;
    rp.counters = be->d_counters;
    rp.work_counter = be->d_counters + rrt::COUNTER_ROWS * rrt::COUNTER_COUNT;

    HIP_TRY(hipMemsetAsync(be->d_counters, 0, sizeof(unsigned long long) * (rrt::COUNTER_ROWS * rrt::COUNTER_COUNT + 1), be->stream));
    // 8 blocks of 4 waves per CU is the most a CU holds; fewer lanes than items are never needed
    uint64_t max_blocks = (uint64_t)be->n_cu * 8;
    uint64_t want = (rp.n_items + 255) / 256;
    int n_blocks = (int)(want < max_blocks ? want : max_blocks);
    if (n_blocks < 1) n_blocks = 1;

    unsigned long long counters[rrt::COUNTER_COUNT + 1];
    unsigned long long counter_rows[rrt::COUNTER_ROWS * rrt::COUNTER_COUNT];
    HIP_TRY(hipEventRecord(be->ev0, be->stream));
    if (desc->max_depth > 0 && wavefront) {
        // (f64: the same pipeline in the reference's arithmetic type, only when asked for by name)
        if (desc->precision == RRT_PRECISION_F64)
            rc = render_wavefront<double>(be, with_tree(be, ds, ds->v64, kind), ds->l64, use_bvh, rp, counters);
        else
            rc = render_wavefront<float>(be, with_tree(be, ds, ds->v32, kind), ds->l32, use_bvh, rp, counters);
        if (rc != RRT_OK) return rc;
    } else if (desc->max_depth > 0) {
        if (desc->precision == RRT_PRECISION_F64)
            HIP_TRY(rrt::launch_render_fused(with_tree(be, ds, ds->v64, kind), ds->l64, use_bvh, rp, n_blocks, be->stream));
        else
            HIP_TRY(rrt::launch_render_fused(with_tree(be, ds, ds->v32, kind), with_tree(be, ds, ds->v64, kind), ds->err32, ds->l32, use_bvh, rp, n_blocks, be->stream));
    }
    HIP_TRY(hipEventRecord(be->ev1, be->stream));

    const bool fused_counters = !wavefront || desc->max_depth == 0;
    if (fused_counters)
        HIP_TRY(hipMemcpyAsync(counter_rows, be->d_counters, sizeof(counter_rows), hipMemcpyDeviceToHost, be->stream));
    std::vector<float> host_tmp;
    if (!on_device) {
        host_tmp.resize(n_pixels * 3);
        HIP_TRY(hipMemcpyAsync(host_tmp.data(), d_accum, n_pixels * 3 * sizeof(float), hipMemcpyDeviceToHost,
                               be->stream));
    }
    HIP_TRY(hipStreamSynchronize(be->stream));
    if (fused_counters) {
        for (int k = 0; k < rrt::COUNTER_COUNT; ++k) counters[k] = 0;
        for (int r = 0; r < rrt::COUNTER_ROWS; ++r)
            for (int k = 0; k < rrt::COUNTER_COUNT; ++k) counters[k] += counter_rows[r * rrt::COUNTER_COUNT + k];
    }
    if (!on_device)
        for (size_t i = 0; i < n_pixels * 3; ++i) rgb_sum[i] += host_tmp[i];
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, be->ev0, be->ev1));
        stats->kernel_ms = ms;
        stats->samples = desc->max_depth > 0 ? counters[rrt::COUNTER_SAMPLES] : (uint64_t)rp.n_samples * n_pixels;
        stats->rays = counters[rrt::COUNTER_RAYS];
        stats->prim_tests = counters[rrt::COUNTER_PRIM_TESTS];
        stats->node_tests = counters[rrt::COUNTER_NODE_TESTS];
        for (int k = 0; k < 5; ++k) stats->bounces[k] = counters[rrt::COUNTER_BOUNCE0 + k];
        stats->nonfinite = counters[rrt::COUNTER_NONFINITE];
        stats->accel_used = kind < 0 ? (uint32_t)RRT_ACCEL_LINEAR : kAccelOfKind[kind];
        stats->pipeline_used = wavefront ? RRT_PIPELINE_WAVEFRONT : RRT_PIPELINE_FUSED;
        stats->tree_depth = kind >= 0 ? (uint32_t)ds->tree[kind].depth : 0u;
        stats->tree_built = built_now ? 1u : 0u;
        stats->tree_build_ms = built_now ? ds->tree[kind].build_ms : 0.0;
        stats->stack_overflows = counters[rrt::COUNTER_STACK_OVERFLOW];
        stats->exact_redos = counters[rrt::COUNTER_EXACT_REDOS];
    }
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_sampler_values(rrt_backend* be, const rrt_render_desc* desc, uint32_t pixel, float* out_host) try {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!out_host) return fail(RRT_ERR_INVALID, "out is NULL");
    if ((rc = validate_render_desc(desc)) != RRT_OK) return rc;
    if (pixel >= desc->width * desc->height) return fail(RRT_ERR_INVALID, "pixel %u outside the image", pixel);
    rrt::RenderParams rp = make_render_params(desc, (uint64_t)be->n_cu * 8 * 256);
    DevBuf buf;
    const size_t bytes = (size_t)rp.n_samples * 8 * sizeof(float);
    HIP_TRY(buf.alloc(bytes));
    HIP_TRY(rrt::launch_sampler_values(rp, pixel, (float*)buf.p, be->stream));
    HIP_TRY(hipMemcpyAsync(out_host, buf.p, bytes, hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipStreamSynchronize(be->stream));
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_material_probe(rrt_backend* be, const rrt_material* material, uint32_t precision, size_t n, const double* d_in,
                       const double* normal, const double* u, double* d_out, double* weight, int32_t* ok) try {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!material) return fail(RRT_ERR_INVALID, "material is NULL");
    if (material->type > RRT_MAT_DIELECTRIC) return fail(RRT_ERR_INVALID, "unknown material type %u", material->type);
    if (precision > RRT_PRECISION_F64) return fail(RRT_ERR_INVALID, "unknown precision %u", precision);
    if (n == 0) return RRT_OK;
    if (!d_in || !normal || !u || !d_out || !weight || !ok) return fail(RRT_ERR_INVALID, "NULL argument");
    rrt::MatRec<double> m;
    prepare_material(*material, m);
    // one staging buffer: three inputs, two outputs of 3 doubles per sample, then the flags
    DevBuf buf;
    const size_t vec = n * 3 * sizeof(double);
    HIP_TRY(buf.alloc(5 * vec + n * sizeof(int32_t)));
    double* d = (double*)buf.p;
    double *dd_in = d, *dd_n = d + 3 * n, *dd_u = d + 6 * n, *dd_out = d + 9 * n, *dd_w = d + 12 * n;
    int32_t* d_ok = (int32_t*)(d + 15 * n);
    HIP_TRY(hipMemcpyAsync(dd_in, d_in, vec, hipMemcpyHostToDevice, be->stream));
    HIP_TRY(hipMemcpyAsync(dd_n, normal, vec, hipMemcpyHostToDevice, be->stream));
    HIP_TRY(hipMemcpyAsync(dd_u, u, vec, hipMemcpyHostToDevice, be->stream));
    if (precision == RRT_PRECISION_F64)
        HIP_TRY(rrt::launch_material_probe(m, n, dd_in, dd_n, dd_u, dd_out, dd_w, d_ok, be->stream));
    else
        HIP_TRY(rrt::launch_material_probe(narrow<float>(m), n, dd_in, dd_n, dd_u, dd_out, dd_w, d_ok, be->stream));
    HIP_TRY(hipMemcpyAsync(d_out, dd_out, vec, hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipMemcpyAsync(weight, dd_w, vec, hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipMemcpyAsync(ok, d_ok, n * sizeof(int32_t), hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipStreamSynchronize(be->stream));
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_postprocess(rrt_backend* be, const float* rgb_sum, uint64_t spp_divisor, const rrt_post_desc* post,
                    uint32_t width, uint32_t height, uint8_t* rgb8) try {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!rgb_sum || !post || !rgb8) return fail(RRT_ERR_INVALID, "NULL argument");
    if (spp_divisor == 0) return fail(RRT_ERR_INVALID, "spp divisor is zero");
    if (post->tone_mapping > 1) return fail(RRT_ERR_INVALID, "unknown tone mapping %u", post->tone_mapping);
    const size_t n = (size_t)width * height * 3;
    if (n == 0) return RRT_OK;
    DevBuf in_stage, out_stage;
    const float* d_in = rgb_sum;
    uint8_t* d_out = rgb8;
    const PtrKind in_kind = classify_pointer(be, rgb_sum), out_kind = classify_pointer(be, rgb8);
    if (in_kind == PtrKind::OtherDevice || out_kind == PtrKind::OtherDevice)
        return fail(RRT_ERR_INVALID, "rgb_sum / rgb8 is device memory of another GPU than the backend's (%d)", be->device);
    const bool in_dev = in_kind == PtrKind::Device, out_dev = out_kind == PtrKind::Device;
    if (!in_dev) {
        HIP_TRY(in_stage.alloc(n * sizeof(float)));
        HIP_TRY(hipMemcpyAsync(in_stage.p, rgb_sum, n * sizeof(float), hipMemcpyHostToDevice, be->stream));
        d_in = (const float*)in_stage.p;
    }
    if (!out_dev) {
        HIP_TRY(out_stage.alloc(n));
        d_out = (uint8_t*)out_stage.p;
    }
    rrt::PostParams pp;
    pp.tone_mapping = (int32_t)post->tone_mapping;
    pp.gamma = post->gamma_correction ? 1 : 0;
    pp.has_wb = post->has_white_balance ? 1 : 0;
    pp.pad = 0;
    for (int k = 0; k < 3; ++k) pp.wb[k] = post->white_balance[k];
    HIP_TRY(rrt::launch_postprocess(d_in, 1.0 / (double)spp_divisor, pp, n, d_out, be->stream));
    if (!out_dev) HIP_TRY(hipMemcpyAsync(rgb8, d_out, n, hipMemcpyDeviceToHost, be->stream));
    HIP_TRY(hipStreamSynchronize(be->stream));
    return RRT_OK;
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

int rrt_render(rrt_backend* be, const rrt_scene* scene, const rrt_render_desc* desc, const rrt_post_desc* post,
               uint8_t* rgb8_host, rrt_stats* stats) try {
    int rc = check_backend(be);
    if (rc != RRT_OK) return rc;
    if (!desc || !post || !rgb8_host) return fail(RRT_ERR_INVALID, "NULL argument");
    rrt_render_desc d = *desc;
    // all samples of the pixel loop: Random starts spp + 1 samples (src/sampler.rs:48-55)
    d.sample_begin = 0;
    d.sample_end = (uint64_t)d.spp + (d.sampler_type == RRT_SAMPLER_RANDOM ? 1 : 0);
    const size_t n = (size_t)d.width * d.height * 3;
    DevBuf accum;
    HIP_TRY(accum.alloc(n * sizeof(float)));
    HIP_TRY(hipMemsetAsync(accum.p, 0, n * sizeof(float), be->stream));
    if ((rc = rrt_render_accumulate(be, scene, &d, (float*)accum.p, stats)) != RRT_OK) return rc;
    return rrt_postprocess(be, (const float*)accum.p, d.spp, post, d.width, d.height, rgb8_host);
} catch (...) {
    return exception_status();  // nothing may unwind through the C ABI
}

}  // extern "C"
