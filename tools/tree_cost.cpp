// Expected traversal steps of the trees the two builders make, on the CPU, from a file of primitive boxes
// (tools/export_boxes.py):   g++ -O2 -std=c++17 -pthread -Irustraytracer_amd/csrc tools/tree_cost.cpp -o /tmp/tree_cost
//                            /tmp/tree_cost BOXES.f32 [sah|cluster] [threads]
// "sah" runs the host builder itself (rrt_sah.hpp); "cluster" is a serial model of the GPU builder (rrt_ploc.hip: same
// Morton codes, radius and merge rule).  The measure: a ray enters a box with a probability proportional to its area, so
//   node steps = 1 + sum over the wide nodes' internal children of area / root area,   primitive steps = the same over leaves.
// On the million-triangle benchmark scene it predicts the node steps the kernels count within 2 %.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rrt_sah.hpp"

using rrt::BvhNode;

static double half_area(const float* l, const float* h) {
    const double x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2];
    return x * y + y * z + z * x;
}

// the wide tree the widening kernel forms from a marked binary tree (rrt_lbvh.hip, widen_kernel)
static void report(const char* name, const std::vector<BvhNode>& nodes, double root_area, double seconds) {
    double inner = 0, leaf = 0;
    size_t wide = 0, slots = 0;
    std::vector<int> todo{0};
    while (!todo.empty()) {
        const int id = todo.back();
        todo.pop_back();
        ++wide;
        struct E { const float *l, *h; int32_t link; } c[4];
        int k = 2;
        c[0] = {nodes[id].lo0, nodes[id].hi0, nodes[id].c0};
        c[1] = {nodes[id].lo1, nodes[id].hi1, nodes[id].c1};
        auto open = [&](int at) { const BvhNode& N = nodes[c[at].link]; c[at] = {N.lo0, N.hi0, N.c0}; c[k] = {N.lo1, N.hi1, N.c1}; return k++; };
        const int32_t shape = nodes[id].pad0;
        int l1 = -1, r1 = -1;
        if ((shape & rrt::SAH_OPEN_LEFT) && c[0].link >= 0) l1 = open(0);
        if ((shape & rrt::SAH_OPEN_RIGHT) && c[1].link >= 0 && k < 4) r1 = open(1);
        if ((shape & rrt::SAH_OPEN_LEFT_0) && l1 >= 0 && c[0].link >= 0 && k < 4) open(0);
        if ((shape & rrt::SAH_OPEN_LEFT_1) && l1 >= 0 && c[l1].link >= 0 && k < 4) open(l1);
        if ((shape & rrt::SAH_OPEN_RIGHT_0) && r1 >= 0 && c[1].link >= 0 && k < 4) open(1);
        if ((shape & rrt::SAH_OPEN_RIGHT_1) && r1 >= 0 && c[r1].link >= 0 && k < 4) open(r1);
        slots += (size_t)k;
        for (int j = 0; j < k; ++j) {
            if (c[j].link >= 0) { inner += half_area(c[j].l, c[j].h); todo.push_back(c[j].link); }
            else leaf += half_area(c[j].l, c[j].h) * (((~c[j].link) >> rrt::BVH_SLOT_BITS) + 1);
        }
    }
    std::printf("%s: %zu binary nodes, %zu wide nodes (%.2f children each), expected node steps %.2f, primitive steps %.2f, built in %.2f s\n",
                name, nodes.size(), wide, (double)slots / (double)wide, 1 + inner / root_area, leaf / root_area, seconds);
}

// The same measure for WIDTH-wide nodes over the same binary tree: a wide node = a binary node with its internal children
// opened largest box first until it has WIDTH children (the greedy rule; for 4-wide nodes the optimal choice of
// sah_mark_node is 3-4 % better than it).
static void report_wide(const char* name, const std::vector<BvhNode>& nodes, double root_area, int width) {
    double inner = 0, leaf = 0;
    size_t wide = 0, slots = 0;
    std::vector<int> todo{0};
    struct E { const float *l, *h; int32_t link; };
    std::vector<E> c;
    while (!todo.empty()) {
        const int id = todo.back();
        todo.pop_back();
        ++wide;
        c.clear();
        c.push_back({nodes[id].lo0, nodes[id].hi0, nodes[id].c0});
        c.push_back({nodes[id].lo1, nodes[id].hi1, nodes[id].c1});
        while ((int)c.size() < width) {
            int best = -1;
            double best_a = -1;
            for (size_t j = 0; j < c.size(); ++j)
                if (c[j].link >= 0 && half_area(c[j].l, c[j].h) > best_a) { best_a = half_area(c[j].l, c[j].h); best = (int)j; }
            if (best < 0) break;
            const BvhNode& N = nodes[c[(size_t)best].link];
            c[(size_t)best] = {N.lo0, N.hi0, N.c0};
            c.push_back({N.lo1, N.hi1, N.c1});
        }
        slots += c.size();
        for (const E& e : c) {
            if (e.link >= 0) { inner += half_area(e.l, e.h); todo.push_back(e.link); }
            else leaf += half_area(e.l, e.h) * (((~e.link) >> rrt::BVH_SLOT_BITS) + 1);
        }
    }
    std::printf("%s, %d-wide (largest child first): %zu wide nodes (%.2f children each), expected node steps %.2f, primitive steps %.2f\n",
                name, width, wide, (double)slots / (double)wide, 1 + inner / root_area, leaf / root_area);
}

static unsigned long long expand21(unsigned long long x) {
    x &= 0x1fffffull; x = (x | x << 32) & 0x1f00000000ffffull; x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full; x = (x | x << 4) & 0x10c30c30c30c30c3ull; x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: tree_cost BOXES.f32 [sah|cluster] [threads]\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::printf("cannot open %s\n", argv[1]); return 1; }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const int n = (int)(bytes / 24);
    std::vector<float> raw((size_t)n * 6), lo((size_t)n * 3), hi((size_t)n * 3);
    if (n < 2 || std::fread(raw.data(), 4, raw.size(), f) != raw.size()) return 1;
    std::fclose(f);
    float slo[3] = {3e38f, 3e38f, 3e38f}, shi[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[3 * i + a] = raw[6 * i + a];
            hi[3 * i + a] = raw[6 * i + 3 + a];
            slo[a] = std::min(slo[a], lo[3 * i + a]);
            shi[a] = std::max(shi[a], hi[3 * i + a]);
        }
    const double root_area = half_area(slo, shi);
    const bool cluster = argc > 2 && !std::strcmp(argv[2], "cluster");
    const int threads = argc > 3 ? std::atoi(argv[3]) : 8;
    const auto t0 = std::chrono::steady_clock::now();
    auto seconds = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (!cluster) {
        std::vector<BvhNode> nodes;
        std::vector<int32_t> order;
        rrt::SahBuilder(lo.data(), hi.data(), n, 1).build(nodes, order, threads);
        report("surface-area builder", nodes, root_area, seconds());
        report_wide("surface-area builder", nodes, root_area, 4);
        report_wide("surface-area builder", nodes, root_area, 6);
        report_wide("surface-area builder", nodes, root_area, 8);
        return 0;
    }
    constexpr int RADIUS = 16;
    std::vector<std::pair<unsigned long long, int>> keys((size_t)n);
    for (int i = 0; i < n; ++i) {
        unsigned long long k = 0;
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * (lo[3 * i + a] + hi[3 * i + a]);
            const float inv = shi[a] > slo[a] ? 1.f / (shi[a] - slo[a]) : 0.f;
            const float v = std::min(std::max((c - slo[a]) * inv * 2097152.f, 0.f), 2097151.f);
            k |= expand21((unsigned long long)v) << (2 - a);
        }
        keys[(size_t)i] = {k, i};
    }
    std::sort(keys.begin(), keys.end());
    struct C { float lo[3], hi[3]; int32_t link; };
    auto united = [](const C& a, const C& b) {
        const float dx = std::max(a.hi[0], b.hi[0]) - std::min(a.lo[0], b.lo[0]), dy = std::max(a.hi[1], b.hi[1]) - std::min(a.lo[1], b.lo[1]),
                    dz = std::max(a.hi[2], b.hi[2]) - std::min(a.lo[2], b.lo[2]);
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<C> cur((size_t)n), nxt((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int p = keys[(size_t)i].second;
        for (int a = 0; a < 3; ++a) { cur[(size_t)i].lo[a] = lo[3 * p + a]; cur[(size_t)i].hi[a] = hi[3 * p + a]; }
        cur[(size_t)i].link = ~i;
    }
    std::vector<BvhNode> nodes((size_t)n - 1);
    std::vector<int> nearest((size_t)n);
    int m = n, next_id = n - 2, rounds = 0;
    bool by_position = false;
    while (m > 1) {
        for (int i = 0; i < m; ++i) {
            if (by_position) { nearest[(size_t)i] = (i ^ 1) < m ? (i ^ 1) : -1; continue; }
            float best = 3e38f;
            int bj = -1;
            for (int d = -RADIUS; d <= RADIUS; ++d) {
                const int j = i + d;
                if (!d || j < 0 || j >= m) continue;
                const float a = united(cur[(size_t)i], cur[(size_t)j]);
                if (a < best) { best = a; bj = j; }
            }
            nearest[(size_t)i] = bj;
        }
        int out = 0, merged = 0;
        for (int i = 0; i < m; ++i) {
            const int j = nearest[(size_t)i];
            const bool mutual = j >= 0 && nearest[(size_t)j] == i;
            if (mutual && j < i) continue;
            C me = cur[(size_t)i];
            if (mutual) {
                const C& o = cur[(size_t)j];
                BvhNode N;
                for (int a = 0; a < 3; ++a) { N.lo0[a] = me.lo[a]; N.hi0[a] = me.hi[a]; N.lo1[a] = o.lo[a]; N.hi1[a] = o.hi[a]; }
                N.c0 = me.link; N.c1 = o.link; N.pad0 = N.pad1 = 0;
                nodes[(size_t)next_id] = N;
                for (int a = 0; a < 3; ++a) { me.lo[a] = std::min(me.lo[a], o.lo[a]); me.hi[a] = std::max(me.hi[a], o.hi[a]); }
                me.link = next_id--;
                ++merged;
            }
            nxt[(size_t)out++] = me;
        }
        by_position = !by_position && m > 64 && merged < m / 32;
        m = out;
        std::swap(cur, nxt);
        ++rounds;
    }
    rrt::SahBuilder::mark_wide_nodes(nodes);
    std::printf("%d rounds; ", rounds);
    report("clustering builder (serial model)", nodes, root_area, seconds());
    return 0;
}
