// Host-side top-down builder of the binary tree under the wide nodes: binned surface-area heuristic.
// Produces what the radix-tree kernels of rrt_lbvh.hip produce (BvhNode records holding the boxes of
// both children, leaves as slot ranges, node 0 the root), so the widening and packing kernels take
// either tree; in addition it chooses, for the whole tree at once, which nodes every wide node absorbs.
// The RRT_SAH_* macros are hooks for experiments (bins per coordinate, exact-sweep range, cost weight,
// number of cut coordinates); the defaults are what measurements on a million-triangle scene chose.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "rrt_types.hpp"

namespace rrt {

struct SahBox {
    float lo[3], hi[3];
    void clear() { for (int a = 0; a < 3; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }
    void grow(const float* l, const float* h) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); } }
    void grow(const SahBox& b) { grow(b.lo, b.hi); }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

#if defined(__HIPCC__)
#define RRT_SAH_HD __host__ __device__
#else
#define RRT_SAH_HD
#endif

// Which binary nodes every 4-wide node absorbs, chosen for the whole tree at once.  A ray pays one step for every
// wide node it enters, i.e. for every child box of a wide node that is itself a wide node; with
//   f(v) = expected steps below v when v is the root of a wide node (box areas as entry probabilities),
// f(v) = min over the ways to absorb up to two internal descendants connected to v of
//        the sum, over the internal children the wide node is left with, of (their box area + their f).
// This is the step for node k: f of its children and grandchildren must be known (any bottom-up order does);
// the choice goes into BvhNode::pad0 as SAH_OPEN_* bits.  Opening the largest child first, as the widening kernel
// does on its own, ends 4 % above this on a million-triangle mesh.
RRT_SAH_HD inline void sah_mark_node(BvhNode* nodes, float* f, size_t k) {
#if defined(__clang__)
#pragma clang fp contract(off)  // host and device must price a node alike: every product and sum rounded
#endif
    BvhNode& N = nodes[k];
    auto area = [](const BvhNode& P, int w) {
        const float* l = w ? P.lo1 : P.lo0;
        const float* h = w ? P.hi1 : P.hi0;
        const float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto link = [](const BvhNode& P, int w) { return w ? P.c1 : P.c0; };
    auto G = [&](const BvhNode& P, int w) { const int32_t c = link(P, w); return c >= 0 ? area(P, w) + f[(size_t)c] : 0.f; };  // a leaf child costs the same wherever it hangs
    float best = G(N, 0) + G(N, 1);
    int32_t shape = 0;
    auto consider = [&](float cost, int32_t bits) { if (cost < best) { best = cost; shape = bits; } };
    for (int w = 0; w < 2; ++w) {
        const int32_t c = link(N, w);
        if (c < 0) continue;
        const BvhNode& C = nodes[(size_t)c];
        const float other = G(N, 1 - w);
        consider(other + G(C, 0) + G(C, 1), SAH_OPEN_LEFT << w);
        for (int x = 0; x < 2; ++x) {
            const int32_t d = link(C, x);
            if (d < 0) continue;
            const BvhNode& D = nodes[(size_t)d];
            consider(other + G(C, 1 - x) + G(D, 0) + G(D, 1), (SAH_OPEN_LEFT << w) | (SAH_OPEN_LEFT_0 << (2 * w + x)));
        }
    }
    if (N.c0 >= 0 && N.c1 >= 0) {
        const BvhNode &L = nodes[(size_t)N.c0], &R = nodes[(size_t)N.c1];
        consider(G(L, 0) + G(L, 1) + G(R, 0) + G(R, 1), SAH_OPEN_LEFT | SAH_OPEN_RIGHT);
    }
    f[k] = best;
    N.pad0 = shape;
}

class SahBuilder {
public:
#ifndef RRT_SAH_BINS
#define RRT_SAH_BINS 16
#endif
    static constexpr int BINS = RRT_SAH_BINS;
    SahBuilder(const float* lo, const float* hi, int n, int leaf_max) : lo_(lo), hi_(hi), n_(n), leaf_max_(leaf_max) {}

    // nodes: at most n - 1 records, node 0 the root; prim_of_slot: n entries.  Needs n >= 2.
    // The top of the tree is split on the calling thread down to ranges of at most `grain` primitives; those
    // subtrees are independent jobs for `threads` threads.  Ranges, cuts and the order of the records depend
    // on the input alone, not on the number of threads.
    // median = true: every range is cut in the middle of its order along its longest coordinate instead (no cost
    // function): a tree of depth ceil(log2 n), for inputs on which the cost function peels primitives off one by one
    // and the tree would be deeper than a traversal stack holds.
    void build(std::vector<BvhNode>& nodes, std::vector<int32_t>& prim_of_slot, int threads, bool median = false) {
        median_ = median;
        prim_of_slot.resize((size_t)n_);
        for (int i = 0; i < n_; ++i) prim_of_slot[(size_t)i] = i;
        idx_ = prim_of_slot.data();
        size_.resize((size_t)n_);
        for (int i = 0; i < n_; ++i) {
            SahBox b;
            b.clear();
            b.grow(lo_ + 3 * (size_t)i, hi_ + 3 * (size_t)i);
            size_[(size_t)i] = std::log2(std::max(b.half_area(), 1e-30f));
        }
        nodes.clear();
        nodes.reserve((size_t)n_);
        nodes.emplace_back();
        top_threads_ = std::max(1, threads);
        const int grain = std::max(4096, n_ / 64);
        std::vector<Job> jobs;
        at_top_ = true;  // (only this thread is at work: large ranges may use all the threads and the shared scratch)
        if (n_ <= grain) jobs.push_back({0, 0, n_});
        else expand(nodes, 0, 0, n_, grain, &jobs);
        at_top_ = false;
        std::vector<std::vector<BvhNode>> local(jobs.size());
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t j; (j = next.fetch_add(1)) < jobs.size();) {
                local[j].reserve((size_t)jobs[j].count);
                local[j].emplace_back();
                expand(local[j], 0, jobs[j].first, jobs[j].count, 0, nullptr);
            }
        };
        threads = std::max(1, std::min(threads, (int)jobs.size()));
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(work);
        work();
        for (std::thread& t : pool) t.join();
        for (size_t j = 0; j < jobs.size(); ++j) {
            const int32_t base = (int32_t)nodes.size();  // local record k >= 1 becomes record base + k - 1
            for (size_t k = 0; k < local[j].size(); ++k) {
                BvhNode N = local[j][k];
                if (N.c0 >= 0) N.c0 += base - 1;
                if (N.c1 >= 0) N.c1 += base - 1;
                if (k == 0) nodes[(size_t)jobs[j].id] = N;
                else nodes.push_back(N);
            }
        }
        mark_wide_nodes(nodes);
        order_top(nodes, 4096);
    }

    // Renumbers the nodes so that the `top` most probably entered ones (largest boxes first, from the root down) come
    // first, in that order; the others keep their relative order behind them.  Children still have larger numbers than
    // their parents.  The traversal kernels keep the first wide nodes of the array in LDS, and the hot top of the tree
    // shares cache lines.
    static void order_top(std::vector<BvhNode>& nodes, size_t top) {
        const size_t m = nodes.size();
        if (m < 3) return;
        top = std::min(top, m);
        std::vector<int32_t> new_id(m, -1);
        auto area = [](const BvhNode& N, int w) {
            const float* l = w ? N.lo1 : N.lo0;
            const float* h = w ? N.hi1 : N.hi0;
            const float dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
            return dx * dy + dy * dz + dz * dx;
        };
        std::vector<std::pair<float, int32_t>> heap;  // (box area, node), largest area on top; ties: the lower number
        auto less = [](const std::pair<float, int32_t>& a, const std::pair<float, int32_t>& b) {
            return a.first < b.first || (a.first == b.first && a.second > b.second);
        };
        heap.push_back({3.0e38f, 0});
        size_t next = 0;
        while (next < top && !heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const int32_t v = heap.back().second;
            heap.pop_back();
            new_id[(size_t)v] = (int32_t)next++;
            const BvhNode& N = nodes[(size_t)v];
            if (N.c0 >= 0) { heap.push_back({area(N, 0), N.c0}); std::push_heap(heap.begin(), heap.end(), less); }
            if (N.c1 >= 0) { heap.push_back({area(N, 1), N.c1}); std::push_heap(heap.begin(), heap.end(), less); }
        }
        for (size_t i = 0; i < m; ++i)
            if (new_id[i] < 0) new_id[i] = (int32_t)next++;
        std::vector<BvhNode> out(m);
        for (size_t i = 0; i < m; ++i) {
            BvhNode N = nodes[i];
            if (N.c0 >= 0) N.c0 = new_id[(size_t)N.c0];
            if (N.c1 >= 0) N.c1 = new_id[(size_t)N.c1];
            out[(size_t)new_id[i]] = N;
        }
        nodes.swap(out);
    }

    // Marks every node of a tree whose children have larger indices than their parents (see sah_mark_node).
    static void mark_wide_nodes(std::vector<BvhNode>& nodes) {
        std::vector<float> f(nodes.size(), 0.f);
        for (size_t k = nodes.size(); k-- > 0;) sah_mark_node(nodes.data(), f.data(), k);
    }

private:
#ifndef RRT_SAH_WEIGHT
#define RRT_SAH_WEIGHT(n) ((float)(n))
#endif
    static float weight(int n) { return RRT_SAH_WEIGHT(n); }
    const float* lo_;
    const float* hi_;
    int n_, leaf_max_;
    int32_t* idx_ = nullptr;
    struct Job { int id, first, count; };
    // Cuts are tried along four coordinates of a primitive: the three of its box centre and, as a fourth, the
    // logarithm of its box area.  A cut along the fourth one takes the few large primitives of a range (the walls
    // of a room around a mesh) apart from the many small ones; cuts by position cannot, because a large box
    // keeps its side of the cut as large as the whole range for level after level.
#ifndef RRT_SAH_AXES
#define RRT_SAH_AXES 4
#endif
    static constexpr int AXES = RRT_SAH_AXES;
    std::vector<float> size_;  // log2 of the half area of every primitive's box
    std::vector<int32_t> temp_;  // scratch of the top-of-tree partition
    float coord(size_t p, int a) const { return a < 3 ? 0.5f * (lo_[3 * p + a] + hi_[3 * p + a]) : size_[p]; }
    int top_threads_ = 1;
    bool at_top_ = false;  // the calling thread is splitting the top of the tree; the jobs have not started
    bool median_ = false;
    struct Extent {
        float lo[AXES], hi[AXES];
        void clear() { for (int a = 0; a < AXES; ++a) { lo[a] = 3.0e38f; hi[a] = -3.0e38f; } }
    };
    struct Bins {
        SahBox box[AXES][BINS];
        int n[AXES][BINS];
        void clear() { for (int a = 0; a < AXES; ++a) for (int b = 0; b < BINS; ++b) { box[a][b].clear(); n[a][b] = 0; } }
    };
    void extent(int first, int count, Extent& e) const {
        for (int i = first; i < first + count; ++i) {
            const size_t p = (size_t)idx_[i];
            for (int a = 0; a < AXES; ++a) {
                const float c = coord(p, a);
                e.lo[a] = std::min(e.lo[a], c);
                e.hi[a] = std::max(e.hi[a], c);
            }
        }
    }
    void fill_bins(int first, int count, const float* clo, const float* scale, Bins& bins) const {
        for (int i = first; i < first + count; ++i) {
            const size_t p = (size_t)idx_[i];
            for (int a = 0; a < AXES; ++a) {
                int b = (int)((coord(p, a) - clo[a]) * scale[a]);
                b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                bins.box[a][b].grow(lo_ + 3 * p, hi_ + 3 * p);
                bins.n[a][b] += 1;
            }
        }
    }
    // fn(chunk, first, count) over `chunks` equal parts of a range, on as many threads
    template <typename F>
    static void run_chunks(int first, int count, int chunks, F fn) {
        if (chunks <= 1) { fn(0, first, count); return; }
        std::vector<std::thread> pool;
        for (int c = 0; c < chunks; ++c) {
            const int f = first + (int)((long long)count * c / chunks), l = first + (int)((long long)count * (c + 1) / chunks);
            if (c + 1 < chunks) pool.emplace_back(fn, c, f, l - f);
            else fn(c, f, l - f);
        }
        for (std::thread& t : pool) t.join();
    }

    SahBox bounds_of(int first, int count) const {
        SahBox b;
        b.clear();
        for (int i = first; i < first + count; ++i) b.grow(lo_ + 3 * (size_t)idx_[i], hi_ + 3 * (size_t)idx_[i]);
        return b;
    }

    // Small ranges: every cut of the order along every axis is tried (bins are too coarse to keep the
    // two triangles of a mesh cell together).
#ifndef RRT_SAH_SWEEP
#define RRT_SAH_SWEEP 32
#endif
    static constexpr int SWEEP_MAX = RRT_SAH_SWEEP;
    int partition_sweep(int first, int count, SahBox& bl, SahBox& br) {
        thread_local std::vector<std::pair<float, int32_t>> key_v;
        thread_local std::vector<float> area_v;
        if ((int)key_v.size() < count) { key_v.resize((size_t)count); area_v.resize((size_t)count); }
        std::pair<float, int32_t>* key = key_v.data();
        float* right_area = area_v.data();
        float best_cost = 3.0e38f;
        int best_nl = count / 2, best_axis = -1;
        auto sort_along = [&](int a) {
            for (int i = 0; i < count; ++i) {
                const int32_t p = idx_[first + i];
                key[i] = {coord((size_t)p, a), p};
            }
            std::sort(key, key + count);
        };
        for (int a = 0; a < AXES; ++a) {
            sort_along(a);
            SahBox acc;
            acc.clear();
            for (int i = count - 1; i > 0; --i) {
                acc.grow(lo_ + 3 * (size_t)key[i].second, hi_ + 3 * (size_t)key[i].second);
                right_area[i] = acc.half_area();
            }
            acc.clear();
            for (int i = 0; i < count - 1; ++i) {
                acc.grow(lo_ + 3 * (size_t)key[i].second, hi_ + 3 * (size_t)key[i].second);
                const int nl = i + 1, nr = count - nl;
                // leaves hold up to leaf_max_ primitives: a part of that size costs one box, whatever it holds
                const float cost = acc.half_area() * (nl <= leaf_max_ ? 1.f : weight(nl)) + right_area[i + 1] * (nr <= leaf_max_ ? 1.f : weight(nr));
                if (cost < best_cost) {
                    best_cost = cost;
                    best_nl = nl;
                    best_axis = a;
                }
            }
        }
        if (best_axis >= 0) {
            if (best_axis != AXES - 1) sort_along(best_axis);
            for (int k = 0; k < count; ++k) idx_[first + k] = key[k].second;
        }
        bl = bounds_of(first, best_nl);
        br = bounds_of(first + best_nl, count - best_nl);
        return best_nl;
    }

    // Where to cut [first, first + count): returns the size of the left part (1 .. count - 1) and the boxes of the parts.
    int partition(int first, int count, SahBox& bl, SahBox& br) {
        if (median_) {
            Extent e;
            e.clear();
            extent(first, count, e);
            int a = 0;
            for (int k = 1; k < 3; ++k)
                if (e.hi[k] - e.lo[k] > e.hi[a] - e.lo[a]) a = k;
            std::nth_element(idx_ + first, idx_ + first + count / 2, idx_ + first + count, [&](int32_t x, int32_t y) {
                const float cx = coord((size_t)x, a), cy = coord((size_t)y, a);
                return cx < cy || (cx == cy && x < y);
            });
            bl = bounds_of(first, count / 2);
            br = bounds_of(first + count / 2, count - count / 2);
            return count / 2;
        }
        if (count <= SWEEP_MAX) return partition_sweep(first, count, bl, br);
        Extent ext;
        ext.clear();
        // Large ranges at the top of the tree are cut with all the threads.  Never inside a job: the jobs run side by side
        // (above 64 x 65536 primitives their ranges are this large too) and the scratch of the gather is one per builder.
        // (The tree does not depend on which way a range was partitioned: nothing below reads the order inside a part.)
        const int chunks = (count >= 65536 && at_top_) ? top_threads_ : 1;
        std::vector<Extent> ext_part((size_t)chunks);
        run_chunks(first, count, chunks, [&](int c, int f, int n) { ext_part[(size_t)c].clear(); extent(f, n, ext_part[(size_t)c]); });
        for (const Extent& e : ext_part)
            for (int a = 0; a < AXES; ++a) { ext.lo[a] = std::min(ext.lo[a], e.lo[a]); ext.hi[a] = std::max(ext.hi[a], e.hi[a]); }
        const float* clo = ext.lo;
        const float* chi = ext.hi;
        float best_cost = 3.0e38f;
        int best_axis = -1, best_bin = 0;
        float best_scale = 0.f;
        float scale[AXES];
        for (int a = 0; a < AXES; ++a) scale[a] = chi[a] > clo[a] ? (float)BINS * (1.f - 1e-6f) / (chi[a] - clo[a]) : 0.f;
        std::vector<Bins> bins_part((size_t)chunks);
        run_chunks(first, count, chunks, [&](int c, int f, int n) { bins_part[(size_t)c].clear(); fill_bins(f, n, clo, scale, bins_part[(size_t)c]); });
        Bins& bins = bins_part[0];
        for (int c = 1; c < chunks; ++c)
            for (int a = 0; a < AXES; ++a)
                for (int k = 0; k < BINS; ++k)
                    if (bins_part[(size_t)c].n[a][k]) { bins.box[a][k].grow(bins_part[(size_t)c].box[a][k]); bins.n[a][k] += bins_part[(size_t)c].n[a][k]; }
        auto& bin_box = bins.box;
        auto& bin_n = bins.n;
        for (int a = 0; a < AXES; ++a) {
            if (!(scale[a] > 0.f)) continue;
            float right_area[BINS];
            SahBox acc;
            acc.clear();
            for (int b = BINS - 1; b > 0; --b) {
                if (bin_n[a][b]) acc.grow(bin_box[a][b]);
                right_area[b] = acc.half_area();
            }
            acc.clear();
            int nl = 0;
            for (int b = 0; b < BINS - 1; ++b) {
                if (bin_n[a][b]) acc.grow(bin_box[a][b]);
                nl += bin_n[a][b];
                const int nr = count - nl;
                if (nl == 0 || nr == 0) continue;
                const float cost = acc.half_area() * weight(nl) + right_area[b + 1] * weight(nr);
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; best_scale = scale[a]; }
            }
        }
        auto halves = [&]() {  // all centres and sizes coincide: any cut is as good
            bl = bounds_of(first, count / 2);
            br = bounds_of(first + count / 2, count - count / 2);
            return count / 2;
        };
        if (best_axis < 0) return halves();
        const int a = best_axis;
        auto goes_left = [&](int32_t p) {
            int b = (int)((coord((size_t)p, a) - clo[a]) * best_scale);
            b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
            return b <= best_bin;
        };
        int nl;
        if (chunks <= 1) {
            nl = (int)(std::partition(idx_ + first, idx_ + first + count, goes_left) - (idx_ + first));
        } else {
            // a large range in PARTS pieces (a fixed number: the order of the result must not depend on the thread
            // count), every piece partitioned in place, then the left parts and the right parts gathered through temp_
            constexpr int PARTS = 16;
            int begin[PARTS + 1], left[PARTS];
            for (int c = 0; c <= PARTS; ++c) begin[c] = first + (int)((long long)count * c / PARTS);
            run_chunks(0, PARTS, std::min(PARTS, top_threads_), [&](int, int c0, int nc) {
                for (int c = c0; c < c0 + nc; ++c) left[c] = (int)(std::partition(idx_ + begin[c], idx_ + begin[c + 1], goes_left) - (idx_ + begin[c]));
            });
            int left_at[PARTS], right_at[PARTS];
            nl = 0;
            for (int c = 0; c < PARTS; ++c) { left_at[c] = nl; nl += left[c]; }
            int nr = 0;
            for (int c = 0; c < PARTS; ++c) { right_at[c] = nl + nr; nr += begin[c + 1] - begin[c] - left[c]; }
            if ((int)temp_.size() < count) temp_.resize((size_t)count);
            run_chunks(0, PARTS, std::min(PARTS, top_threads_), [&](int, int c0, int nc) {
                for (int c = c0; c < c0 + nc; ++c) {
                    std::copy(idx_ + begin[c], idx_ + begin[c] + left[c], temp_.data() + left_at[c]);
                    std::copy(idx_ + begin[c] + left[c], idx_ + begin[c + 1], temp_.data() + right_at[c]);
                }
            });
            run_chunks(0, count, top_threads_, [&](int, int f, int n) { std::copy(temp_.data() + f, temp_.data() + f + n, idx_ + first + f); });
        }
        if (nl <= 0 || nl >= count) return halves();
        bl.clear();
        br.clear();
        for (int b = 0; b < BINS; ++b)
            if (bin_n[a][b]) (b <= best_bin ? bl : br).grow(bin_box[a][b]);  // the bins already hold the boxes of both parts
        return nl;
    }

    // Builds the subtree of out[root] over [first0, first0 + count0).  With `jobs`, ranges of at most `grain`
    // primitives get their record in `out` but are left for later (one job each).
    void expand(std::vector<BvhNode>& out, int root, int first0, int count0, int grain, std::vector<Job>* jobs) {
        std::vector<Job> todo;
        todo.push_back({root, first0, count0});
        auto link = [&](int first, int count) -> int32_t {
            if (count <= leaf_max_) return ~(first | ((count - 1) << BVH_SLOT_BITS));
            const int id = (int)out.size();
            out.emplace_back();
            if (jobs && count <= grain) jobs->push_back({id, first, count});
            else todo.push_back({id, first, count});
            return id;
        };
        while (!todo.empty()) {
            const Job j = todo.back();
            todo.pop_back();
            SahBox bl, br;
            const int nl = partition(j.first, j.count, bl, br);
            const int32_t c0 = link(j.first, nl), c1 = link(j.first + nl, j.count - nl);
            BvhNode& N = out[(size_t)j.id];
            for (int a = 0; a < 3; ++a) { N.lo0[a] = bl.lo[a]; N.hi0[a] = bl.hi[a]; N.lo1[a] = br.lo[a]; N.hi1[a] = br.hi[a]; }
            N.c0 = c0;
            N.c1 = c1;
            N.pad0 = N.pad1 = 0;
        }
    }
};

}  // namespace rrt
