// pt_bvh.cpp -- binned-SAH BVH2 builder producing the 64-byte node / 48-byte triangle layout of pt_types.h.
#include "pt_bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#ifdef __linux__
#include <sched.h>
#endif

#ifndef PT_SAH_BINS
#define PT_SAH_BINS 64   // 16 -> 64 bins: 1.5 % fewer node visits on C4, 552 -> 541 ms (profiles/r03_notes.md)
#endif
#ifndef PT_SAH_LEAFCOUNT
#define PT_SAH_LEAFCOUNT 0
#endif
#ifndef PT_SAH_SWEEP
#define PT_SAH_SWEEP 0   // > 0: nodes of at most this many triangles are split by the exact sweep instead of bins (512: 0.2 % fewer node visits, 2x the build time)
#endif

namespace {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int a = 0; a < 3; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; } }
    void grow(const float* p) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); } }
    float half_area() const
    {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (!(dx >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

// A small fork-join pool for the build: run(n, fn) calls fn(i) for i in [0, n) on the pool's threads and the caller.
class Pool {
public:
    explicit Pool(int threads)
    {
        for (int t = 1; t < threads; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++gen_;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size() + 1; }
    template <class F>
    void run(int n, F&& fn)
    {
        if (n <= 0) return;
        if (workers_.empty() || n == 1) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        std::function<void(int)> f = fn;
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &f;
            n_ = n;
            next_.store(0);
            left_ = (int)workers_.size();
            ++gen_;
        }
        cv_.notify_all();
        for (int i; (i = next_.fetch_add(1)) < n;) f(i);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop()
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)>* f;
            int n;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                f = fn_;
                n = n_;
            }
            for (int i; (i = next_.fetch_add(1)) < n;) (*f)(i);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--left_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, left_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

// Threads the build may use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a container 16 of its 256 cores),
// at most 32; PT_BUILD_THREADS overrides.
int build_threads()
{
    if (const char* e = getenv("PT_BUILD_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) return v > 64 ? 64 : v;
    }
    int n = (int)std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const int q = (int)((atof(quota) + period / 2) / period);
            if (q >= 1 && q < n) n = q;
        }
        fclose(f);
    }
#endif
    return n < 1 ? 1 : (n > 32 ? 32 : n);
}

} // namespace

// fn(begin, end) over [0, n) in one contiguous share per thread (pt_api.cpp: flattening and re-ordering of the per-triangle records)
void pt_parallel_ranges(size_t n, const std::function<void(size_t, size_t)>& fn)
{
    const int T = n >= 65536 ? build_threads() : 1;
    if (T == 1) { fn(0, n); return; }
    Pool pool(T);
    pool.run(T, [&](int t) { fn(n * (size_t)t / (size_t)T, n * (size_t)(t + 1) / (size_t)T); });
}

namespace {

#ifndef PT_BUILD_PAR_MIN
#define PT_BUILD_PAR_MIN 16384 // nodes with more triangles are split with all threads; smaller subtrees are tasks of their own
#endif

// Binned-SAH builder.  The tree, its node numbering (depth-first pre-order) and the triangle order do not depend on the number of
// threads: nodes of more than PT_BUILD_PAR_MIN triangles are split one after the other with every thread binning a share of the
// triangles (bins are min / max / counts: order independent) and taking part in a stable partition, the subtrees below them are
// independent tasks that build into arrays of their own, and a last pass lays top nodes and subtrees out in the order the
// recursion would have created them.  (Against the single-threaded builder of rounds 1-3: the same nodes; inside a leaf the
// triangles may stand in another order - the top splits used std::partition - which no result depends on: closest hits break ties
// by triangle id.)
struct Builder {
    const float* pos;
    std::vector<Box> tb;        // per-triangle bounds
    std::vector<float> cen;     // per-triangle centroid * 3
    std::vector<int32_t> order; // permutation being partitioned
    std::vector<int32_t> scratch; // the parallel partition's second buffer
    int leaf_size, max_depth;
    float pad;
    Pool* pool = nullptr;

    struct Sub { // a subtree with local node indices
        std::vector<PtNode> nodes;
        int depth = 0, max_leaf = 0;
    };

    static int levels_needed(int n, int leaf)
    {
        int l = 0;
        long cap = leaf;
        while (cap < n) { cap *= 2; ++l; }
        return l; // internal levels a balanced split needs
    }

    void median_split(int lo, int hi, int axis, int mid)
    {
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [&](int32_t a, int32_t b) {
            float ca = cen[(size_t)a * 3 + axis], cb = cen[(size_t)b * 3 + axis];
            return ca < cb || (ca == cb && a < b);
        });
    }

    // fn(chunk_lo, chunk_hi, chunk_index) over [lo, hi) - on the pool when `par`
    template <class F>
    void chunks(bool par, int lo, int hi, int* n_chunks, F&& fn)
    {
        const int T = par ? pool->size() : 1;
        *n_chunks = T;
        if (T == 1) { fn(lo, hi, 0); return; }
        const long n = hi - lo;
        pool->run(T, [&](int t) { fn(lo + (int)(n * t / T), lo + (int)(n * (t + 1) / T), t); });
    }

    // Splits order[lo, hi) (more than leaf_size triangles): returns mid and the boxes of the two sides.
    int split(int lo, int hi, int depth, bool& balanced, Box& l, Box& r, bool par)
    {
        const int n = hi - lo;
        constexpr int MAXT = 64;
        int nt = 1;
        Box cb;
        cb.reset();
        {
            Box part[MAXT];
            chunks(par, lo, hi, &nt, [&](int a, int b, int t) {
                Box c;
                c.reset();
                for (int i = a; i < b; ++i) c.grow(&cen[(size_t)order[i] * 3]);
                part[t] = c;
            });
            for (int t = 0; t < nt; ++t) cb.grow(part[t]);
        }
        const int remaining = max_depth - depth; // internal levels still available below this node (this one included)
        if (!balanced && levels_needed(n, leaf_size) >= remaining) balanced = true;

        int mid = -1;
        // (PT_SAH_LEAFCOUNT = 1 counts leaves, ceil(n / leaf_size), instead of triangles - a leaf step requests the records of a whole
        // leaf together - but fuller leaves cost more triangle tests than the saved node steps are worth: C4 +19 % tests, 541 -> 549 ms)
        auto leaves_of = [&](int c) { return PT_SAH_LEAFCOUNT ? (float)((c + leaf_size - 1) / leaf_size) : (float)c; };
        bool have_boxes = false;
        if (!balanced && n <= PT_SAH_SWEEP) {
            // small node (most nodes are): the exact sweep - every split position along every axis, triangles sorted by centroid
            float best = INFINITY;
            int best_axis = -1, best_pos = -1;
            std::vector<int32_t> tmp((size_t)n);
            std::vector<float> suffix((size_t)n + 1);
            for (int axis = 0; axis < 3; ++axis) {
                if (!(cb.mx[axis] - cb.mn[axis] > 0.0f)) continue;
                std::copy(order.begin() + lo, order.begin() + hi, tmp.begin());
                std::sort(tmp.begin(), tmp.end(), [&](int32_t a, int32_t b) {
                    const float ca = cen[(size_t)a * 3 + axis], cbv = cen[(size_t)b * 3 + axis];
                    return ca < cbv || (ca == cbv && a < b);
                });
                Box acc;
                acc.reset();
                for (int i = n - 1; i > 0; --i) { acc.grow(tb[tmp[(size_t)i]]); suffix[(size_t)i] = acc.half_area(); }
                acc.reset();
                for (int i = 0; i < n - 1; ++i) {
                    acc.grow(tb[tmp[(size_t)i]]);
                    const float cost = acc.half_area() * leaves_of(i + 1) + suffix[(size_t)i + 1] * leaves_of(n - i - 1);
                    if (cost < best) { best = cost; best_axis = axis; best_pos = i + 1; }
                }
            }
            if (best_axis >= 0) {
                std::sort(order.begin() + lo, order.begin() + hi, [&](int32_t a, int32_t b) {
                    const float ca = cen[(size_t)a * 3 + best_axis], cbv = cen[(size_t)b * 3 + best_axis];
                    return ca < cbv || (ca == cbv && a < b);
                });
                mid = lo + best_pos;
            }
        } else if (!balanced) {
            constexpr int NB = PT_SAH_BINS;
            struct Bins { Box bb[3][NB]; int cnt[3][NB]; };
            float lo_c[3], scale[3];
            bool use[3];
            for (int axis = 0; axis < 3; ++axis) {
                const float ext = cb.mx[axis] - cb.mn[axis];
                use[axis] = ext > 0.0f;
                lo_c[axis] = cb.mn[axis];
                scale[axis] = use[axis] ? (float)NB / ext : 0.0f;
            }
            // one pass over the triangles fills the bins of all three axes; with `par` every thread bins a share and the shares are
            // merged (boxes by min / max, counts by addition: the result does not depend on the split)
            std::vector<Bins> part((size_t)(par ? pool->size() : 1));
            chunks(par, lo, hi, &nt, [&](int a, int b, int t) {
                Bins& B = part[(size_t)t];
                for (int axis = 0; axis < 3; ++axis)
                    for (int k = 0; k < NB; ++k) { B.bb[axis][k].reset(); B.cnt[axis][k] = 0; }
                for (int i = a; i < b; ++i) {
                    const int id = order[i];
                    const Box& t_box = tb[id];
                    for (int axis = 0; axis < 3; ++axis) {
                        if (!use[axis]) continue;
                        int k = (int)((cen[(size_t)id * 3 + axis] - lo_c[axis]) * scale[axis]);
                        k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                        B.bb[axis][k].grow(t_box);
                        B.cnt[axis][k]++;
                    }
                }
            });
            Bins& B = part[0];
            for (int t = 1; t < nt; ++t)
                for (int axis = 0; axis < 3; ++axis)
                    for (int k = 0; k < NB; ++k) { B.bb[axis][k].grow(part[(size_t)t].bb[axis][k]); B.cnt[axis][k] += part[(size_t)t].cnt[axis][k]; }
            float best = INFINITY;
            int best_axis = -1, best_bin = -1;
            for (int axis = 0; axis < 3; ++axis) {
                if (!use[axis]) continue;
                float ra[NB];
                int rc[NB];
                Box acc;
                acc.reset();
                int c = 0;
                for (int k = NB - 1; k > 0; --k) { acc.grow(B.bb[axis][k]); c += B.cnt[axis][k]; ra[k] = acc.half_area(); rc[k] = c; }
                acc.reset();
                c = 0;
                for (int k = 0; k < NB - 1; ++k) {
                    acc.grow(B.bb[axis][k]);
                    c += B.cnt[axis][k];
                    if (c == 0 || rc[k + 1] == 0) continue;
                    float cost = acc.half_area() * leaves_of(c) + ra[k + 1] * leaves_of(rc[k + 1]);
                    if (cost < best) { best = cost; best_axis = axis; best_bin = k; }
                }
            }
            if (best_axis >= 0) {
                const float lc = lo_c[best_axis], sc = scale[best_axis];
                auto goes_left = [&](int32_t id) {
                    int k = (int)((cen[(size_t)id * 3 + best_axis] - lc) * sc);
                    k = k < 0 ? 0 : (k >= NB ? NB - 1 : k);
                    return k <= best_bin;
                };
                if (par) {
                    // all threads: a STABLE partition through `scratch` (each thread counts its share's left side, a prefix sum places the
                    // shares, every thread writes its share) - the only order that does not depend on how the range was shared out
                    int n_left[MAXT], start_l[MAXT], start_r[MAXT];
                    chunks(true, lo, hi, &nt, [&](int a, int b, int t) {
                        int c = 0;
                        for (int i = a; i < b; ++i) c += goes_left(order[i]) ? 1 : 0;
                        n_left[t] = c;
                    });
                    int total_left = 0;
                    for (int t = 0; t < nt; ++t) total_left += n_left[t];
                    int run_l = 0, run_r = total_left;
                    for (int t = 0; t < nt; ++t) {
                        const int a = lo + (int)((long)n * t / nt), b = lo + (int)((long)n * (t + 1) / nt);
                        start_l[t] = run_l; start_r[t] = run_r;
                        run_l += n_left[t]; run_r += (b - a) - n_left[t];
                    }
                    chunks(true, lo, hi, &nt, [&](int a, int b, int t) {
                        int32_t* dl = scratch.data() + start_l[t];
                        int32_t* dr = scratch.data() + start_r[t];
                        for (int i = a; i < b; ++i) {
                            const int32_t id = order[i];
                            if (goes_left(id)) *dl++ = id; else *dr++ = id;
                        }
                    });
                    chunks(true, lo, hi, &nt, [&](int a, int b, int) { std::memcpy(order.data() + a, scratch.data() + (a - lo), (size_t)(b - a) * 4); });
                    mid = lo + total_left;
                } else {
                    auto it = std::partition(order.begin() + lo, order.begin() + hi, goes_left);
                    mid = (int)(it - order.begin());
                }
                if (mid == lo || mid == hi) mid = -1;
                else { // the sides' boxes are unions of the bins (every triangle is in exactly one)
                    l.reset();
                    r.reset();
                    for (int k = 0; k <= best_bin; ++k) l.grow(B.bb[best_axis][k]);
                    for (int k = best_bin + 1; k < NB; ++k) r.grow(B.bb[best_axis][k]);
                    have_boxes = true;
                }
            }
        }
        if (mid < 0) { // balanced / degenerate: object median along the widest centroid axis
            int axis = 0;
            float e = cb.mx[0] - cb.mn[0];
            if (cb.mx[1] - cb.mn[1] > e) { axis = 1; e = cb.mx[1] - cb.mn[1]; }
            if (cb.mx[2] - cb.mn[2] > e) axis = 2;
            mid = lo + n / 2;
            median_split(lo, hi, axis, mid);
        }
        if (!have_boxes) {
            l.reset();
            r.reset();
            for (int i = lo; i < mid; ++i) l.grow(tb[order[i]]);
            for (int i = mid; i < hi; ++i) r.grow(tb[order[i]]);
        }
        return mid;
    }

    void store(PtNode& nd, const Box& l, const Box& r, int32_t lc, int32_t rc) const
    {
        for (int a = 0; a < 3; ++a) {
            nd.lo[a][0] = l.mn[a] - pad; nd.hi[a][0] = l.mx[a] + pad;
            nd.lo[a][1] = r.mn[a] - pad; nd.hi[a][1] = r.mx[a] + pad;
        }
        nd.left = lc;
        nd.right = rc;
        nd.pad[0] = nd.pad[1] = 0;
    }

    // serial recursion into `dst` (local indices); depth = number of internal nodes on the path including the one created here
    int32_t build(Sub& dst, int lo, int hi, int depth, bool balanced)
    {
        const int n = hi - lo;
        if (n <= leaf_size) {
            dst.max_leaf = std::max(dst.max_leaf, n);
            return ~((lo << 3) | n);
        }
        Box l, r;
        const int mid = split(lo, hi, depth, balanced, l, r, false);
        const int idx = (int)dst.nodes.size();
        dst.nodes.emplace_back();
        dst.depth = std::max(dst.depth, depth);
        const int32_t lc = build(dst, lo, mid, depth + 1, balanced);
        const int32_t rc = build(dst, mid, hi, depth + 1, balanced);
        store(dst.nodes[(size_t)idx], l, r, lc, rc);
        return idx;
    }

    // ---- the top of the tree: nodes split with all threads; what hangs below them is a task ----
    struct Ref { int kind; int32_t v; }; // 0: leaf code, 1: top node, 2: task
    struct Top { Box l, r; Ref lc, rc; int depth; };
    struct Task { int lo, hi, depth; bool balanced; Sub sub; int32_t base = 0; };
    std::vector<Top> top;
    std::vector<Task> tasks;
    int top_max_leaf = 0;

    Ref build_top(int lo, int hi, int depth, bool balanced)
    {
        const int n = hi - lo;
        if (n <= leaf_size) {
            top_max_leaf = std::max(top_max_leaf, n);
            return Ref{0, (int32_t)~((lo << 3) | n)};
        }
        if (n <= PT_BUILD_PAR_MIN) {
            tasks.emplace_back();
            Task& t = tasks.back();
            t.lo = lo; t.hi = hi; t.depth = depth; t.balanced = balanced;
            return Ref{2, (int32_t)tasks.size() - 1};
        }
        Box l, r;
        const int mid = split(lo, hi, depth, balanced, l, r, true);
        const int idx = (int)top.size();
        top.emplace_back();
        const Ref lc = build_top(lo, mid, depth + 1, balanced);
        const Ref rc = build_top(mid, hi, depth + 1, balanced);
        Top& t = top[(size_t)idx];
        t.l = l; t.r = r; t.lc = lc; t.rc = rc; t.depth = depth;
        return Ref{1, idx};
    }

    void run(int n_tris, PtBvh* out)
    {
        scratch.resize((size_t)n_tris);
        const Ref root = build_top(0, n_tris, 1, false);
        pool->run((int)tasks.size(), [&](int k) {
            Task& t = tasks[(size_t)k];
            t.sub.nodes.reserve((size_t)(t.hi - t.lo) / 3 + 16);
            (void)build(t.sub, t.lo, t.hi, t.depth, t.balanced); // more than leaf_size triangles: its root is local node 0
        });
        // final indices in the order the serial recursion creates nodes: a node, its left subtree, its right subtree
        std::vector<int32_t> final_of(top.size(), -1);
        int32_t next = 0;
        {
            std::vector<Ref> st{root};
            while (!st.empty()) {
                const Ref r = st.back();
                st.pop_back();
                if (r.kind == 1) {
                    final_of[(size_t)r.v] = next++;
                    st.push_back(top[(size_t)r.v].rc); // left first
                    st.push_back(top[(size_t)r.v].lc);
                } else if (r.kind == 2) {
                    tasks[(size_t)r.v].base = next;
                    next += (int32_t)tasks[(size_t)r.v].sub.nodes.size();
                }
            }
        }
        auto resolve = [&](const Ref& r) { return r.kind == 0 ? r.v : (r.kind == 1 ? final_of[(size_t)r.v] : tasks[(size_t)r.v].base); };
        out->nodes.resize((size_t)next);
        out->depth = 0;
        out->max_leaf = top_max_leaf;
        for (size_t i = 0; i < top.size(); ++i) {
            const Top& t = top[i];
            store(out->nodes[(size_t)final_of[i]], t.l, t.r, resolve(t.lc), resolve(t.rc));
            out->depth = std::max(out->depth, t.depth);
        }
        pool->run((int)tasks.size(), [&](int k) {
            const Task& t = tasks[(size_t)k];
            PtNode* dst = out->nodes.data() + t.base;
            for (size_t i = 0; i < t.sub.nodes.size(); ++i) {
                PtNode nd = t.sub.nodes[i];
                if (nd.left >= 0) nd.left += t.base;
                if (nd.right >= 0) nd.right += t.base;
                dst[i] = nd;
            }
        });
        for (const Task& t : tasks) {
            out->depth = std::max(out->depth, t.sub.depth);
            out->max_leaf = std::max(out->max_leaf, t.sub.max_leaf);
        }
        out->root = resolve(root);
    }
};

// Tree rotations (Kensler 2008) after the top-down build: at every internal node, bottom-up, one of its children may trade places
// with a grandchild on the other side if that makes the other child's box smaller - the set of leaves below the node, hence its own
// box, does not change.  Leaves stay what they are (ranges of the triangle array), only the topology above them moves.
#ifndef PT_BVH_ROTATIONS
#define PT_BVH_ROTATIONS 0 // passes over the tree (0: off)
#endif
static void rotate_tree(PtBvh* b, int max_depth, int passes)
{
    if (b->root < 0) return;
    const std::vector<PtNode> saved = b->nodes;
    const int saved_depth = b->depth;
    auto area2 = [](const float lo[3], const float hi[3]) {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<int32_t> post;
    for (int pass = 0; pass < passes; ++pass) {
        post.clear();
        std::vector<int32_t> st{b->root};
        while (!st.empty()) { // reverse pre-order = children before parents when walked backwards
            const int32_t i = st.back();
            st.pop_back();
            post.push_back(i);
            const PtNode& nd = b->nodes[(size_t)i];
            if (nd.left >= 0) st.push_back(nd.left);
            if (nd.right >= 0) st.push_back(nd.right);
        }
        long n_rot = 0;
        for (size_t k = post.size(); k-- > 0;) {
            PtNode& N = b->nodes[(size_t)post[k]];
            int best_s = -1, best_g = -1;
            float best_gain = 0.0f;
            for (int s = 0; s < 2; ++s) {
                const int32_t c = s ? N.right : N.left;
                if (c < 0) continue;
                const PtNode& C = b->nodes[(size_t)c];
                float clo[3], chi[3], olo[3], ohi[3];
                for (int a = 0; a < 3; ++a) { clo[a] = N.lo[a][s]; chi[a] = N.hi[a][s]; olo[a] = N.lo[a][1 - s]; ohi[a] = N.hi[a][1 - s]; }
                const float a_c = area2(clo, chi);
                for (int g = 0; g < 2; ++g) { // grandchild g of C goes up, the other child of N comes down next to C's child 1 - g
                    float nlo[3], nhi[3];
                    for (int a = 0; a < 3; ++a) { nlo[a] = std::min(olo[a], C.lo[a][1 - g]); nhi[a] = std::max(ohi[a], C.hi[a][1 - g]); }
                    const float gain = a_c - area2(nlo, nhi);
                    if (gain > best_gain) { best_gain = gain; best_s = s; best_g = g; }
                }
            }
            if (best_s < 0) continue;
            const int s = best_s, g = best_g;
            const int32_t c = s ? N.right : N.left;
            PtNode& C = b->nodes[(size_t)c];
            int32_t& n_other = s ? N.left : N.right;
            int32_t& c_up = g ? C.right : C.left;
            // boxes: the one that goes up, the one that comes down
            float ulo[3], uhi[3], dlo[3], dhi[3];
            for (int a = 0; a < 3; ++a) { ulo[a] = C.lo[a][g]; uhi[a] = C.hi[a][g]; dlo[a] = N.lo[a][1 - s]; dhi[a] = N.hi[a][1 - s]; }
            std::swap(n_other, c_up);
            for (int a = 0; a < 3; ++a) {
                C.lo[a][g] = dlo[a]; C.hi[a][g] = dhi[a];
                N.lo[a][1 - s] = ulo[a]; N.hi[a][1 - s] = uhi[a];
                N.lo[a][s] = std::min(C.lo[a][0], C.lo[a][1]); N.hi[a][s] = std::max(C.hi[a][0], C.hi[a][1]);
            }
            ++n_rot;
        }
        if (n_rot == 0) break;
    }
    // depth of the rotated tree; keep the original if the stack bound would be exceeded
    int depth = 0;
    std::vector<std::pair<int32_t, int>> st{{b->root, 1}};
    while (!st.empty()) {
        const auto [i, d] = st.back();
        st.pop_back();
        depth = std::max(depth, d);
        const PtNode& nd = b->nodes[(size_t)i];
        if (nd.left >= 0) st.push_back({nd.left, d + 1});
        if (nd.right >= 0) st.push_back({nd.right, d + 1});
    }
    if (depth > max_depth) { b->nodes = saved; b->depth = saved_depth; }
    else b->depth = depth;
}

} // namespace

// ---- memory layout passes (the tree itself is untouched: same boxes, same topology, same leaf contents) -----------------------
// L2 misses are served at one 128-byte line per request whatever part of the line is used (measured: tools/gather_rec.hip,
// profiles/r02_gather_ceilings.md), so records that are read together should share lines.

// Sibling pairs: the two children of a node that are internal nodes get adjacent records, the pair aligned to 128 bytes
// (index 2k, 2k + 1); a pair is followed by the pairs of the left subtree, then those of the right subtree.  A ray that
// visits both children pays for one line; a single internal child leaves a 64-byte hole.
static void layout_sibling_pairs(PtBvh* b)
{
    const size_t n = b->nodes.size();
    if (b->root < 0 || n == 0) return;
    std::vector<int32_t> new_of(n, -1);
    std::vector<int32_t> stack;
    int32_t next = 2; // root alone in the first line
    new_of[(size_t)b->root] = 0;
    stack.push_back(b->root);
    while (!stack.empty()) {
        const int32_t i = stack.back();
        stack.pop_back();
        const PtNode& nd = b->nodes[(size_t)i];
        if (nd.left >= 0 || nd.right >= 0) {
            if (nd.left >= 0) new_of[(size_t)nd.left] = next;
            if (nd.right >= 0) new_of[(size_t)nd.right] = next + 1;
            next += 2;
            if (nd.right >= 0) stack.push_back(nd.right); // left subtree first
            if (nd.left >= 0) stack.push_back(nd.left);
        }
    }
    std::vector<PtNode> out((size_t)next);
    std::memset(out.data(), 0, out.size() * sizeof(PtNode));
    for (size_t i = 0; i < out.size(); ++i) { // holes: empty boxes, never referenced
        for (int a = 0; a < 3; ++a) { out[i].lo[a][0] = out[i].lo[a][1] = INFINITY; out[i].hi[a][0] = out[i].hi[a][1] = -INFINITY; }
        out[i].left = out[i].right = -1;
    }
    for (size_t i = 0; i < n; ++i) {
        if (new_of[i] < 0) continue;
        PtNode nd = b->nodes[i];
        if (nd.left >= 0) nd.left = new_of[(size_t)nd.left];
        if (nd.right >= 0) nd.right = new_of[(size_t)nd.right];
        out[(size_t)new_of[i]] = nd;
    }
    b->nodes.swap(out);
    b->root = 0;
}

// Leaf alignment: every leaf starts at a triangle slot that is a multiple of `align` (4 slots = 192 bytes: a leaf of up to four
// 48-byte records then covers exactly two 128-byte lines instead of two or three).  Padding slots hold a never-hit record.
static void layout_align_leaves(PtBvh* b, int align)
{
    if (align <= 1 || b->tris.empty()) return;
    // leaves partition the slot range in order of their first slot
    struct LeafRef { uint32_t first, count; int32_t* ref; };
    std::vector<LeafRef> leaves;
    auto note = [&](int32_t* ref) {
        if (*ref < -1) { const uint32_t code = ~(uint32_t)*ref; leaves.push_back({code >> 3, code & 7u, ref}); }
    };
    for (PtNode& nd : b->nodes) { note(&nd.left); note(&nd.right); }
    note(&b->root);
    std::sort(leaves.begin(), leaves.end(), [](const LeafRef& x, const LeafRef& y) { return x.first < y.first; });
    std::vector<PtTri> out;
    out.reserve(b->tris.size() + leaves.size() * (size_t)(align - 1));
    PtTri dummy;
    std::memset(&dummy, 0, sizeof(dummy));
    dummy.id = 0x7fffffff;
    dummy.material = -1;
    for (const LeafRef& lf : leaves) {
        while (out.size() % (size_t)align) out.push_back(dummy);
        const uint32_t nf = (uint32_t)out.size();
        for (uint32_t k = 0; k < lf.count; ++k) out.push_back(b->tris[lf.first + k]);
        *lf.ref = (int32_t)~((nf << 3) | lf.count);
    }
    b->tris.swap(out);
}

// Quad nodes by surface area (round 4): starting from the two children of a binary node, the internal slot with the largest box is
// replaced by its two children until four slots are used (as pt_bvh_collapse8 does for eight).  Against the fixed two-level collapse
// below: no slot stays empty next to a leaf child while another slot could still be opened, and a large child is opened in preference
// to a small one - fewer quad nodes on a ray's way (expected visits = sum over quad nodes of area(slot box) / area(root)).
static void collapse4_by_area(const PtBvh& b, std::vector<PtNode4>* out, int32_t* root4, int* depth4)
{
    struct Slot { float lo[3], hi[3]; int32_t ref; };
    auto area = [](const Slot& s) {
        const float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto child_slot = [&](const PtNode& nd, int side) {
        Slot s;
        for (int a = 0; a < 3; ++a) { s.lo[a] = nd.lo[a][side]; s.hi[a] = nd.hi[a][side]; }
        s.ref = side ? nd.right : nd.left;
        return s;
    };
    struct Item { int32_t node2; int32_t idx4; int depth; };
    std::vector<Item> todo;
    out->emplace_back();
    todo.push_back({b.root, 0, 1});
    *root4 = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        *depth4 = std::max(*depth4, it.depth);
        Slot slots[4];
        int n = 0;
        {
            const PtNode& nd = b.nodes[(size_t)it.node2];
            for (int side = 0; side < 2; ++side)
                if ((side ? nd.right : nd.left) != -1) slots[n++] = child_slot(nd, side);
        }
        while (n < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (int k = 0; k < n; ++k)
                if (slots[k].ref >= 0) {
                    const float ar = area(slots[k]);
                    if (ar > best_area) { best_area = ar; best = k; }
                }
            if (best < 0) break;
            const PtNode& cn = b.nodes[(size_t)slots[best].ref];
            // the two children stay next to each other (the kernel pushes a slot's pair partner last: siblings are usually next nearest)
            for (int k = n; k > best + 1; --k) slots[k] = slots[k - 1];
            slots[best] = child_slot(cn, 0);
            slots[best + 1] = child_slot(cn, 1);
            ++n;
        }
        PtNode4 q;
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 4; ++k) q.lo[a][k] = q.hi[a][k] = INFINITY; // never hit (empty slot)
        for (int k = 0; k < 4; ++k) { q.child[k] = -1; q.pad[k] = 0; }
        for (int k = 0; k < n; ++k) {
            for (int a = 0; a < 3; ++a) { q.lo[a][k] = slots[k].lo[a]; q.hi[a][k] = slots[k].hi[a]; }
            if (slots[k].ref >= 0) {
                const int32_t idx = (int32_t)out->size();
                out->emplace_back();
                q.child[k] = idx;
                todo.push_back({slots[k].ref, idx, it.depth + 1});
            } else {
                q.child[k] = slots[k].ref;
            }
        }
        (*out)[(size_t)it.idx4] = q;
    }
}

// expected quad-node visits of a long random ray that crosses the root box: sum over the quad nodes of area(box of the slot that refers to
// the node) / area(root box); leaf visits likewise (diagnostics: pt_debug_quad_info)
void pt_bvh_quad_cost(const std::vector<PtNode4>& nodes4, int32_t root4, double* node_visits, double* leaf_visits)
{
    *node_visits = *leaf_visits = 0.0;
    if (root4 < 0 || nodes4.empty()) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const PtNode4& r = nodes4[(size_t)root4];
    for (int k = 0; k < 4; ++k)
        if (r.child[k] != -1)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], r.lo[a][k]); hi[a] = std::max(hi[a], r.hi[a][k]); }
    const double ra = (double)(hi[0] - lo[0]) * (hi[1] - lo[1]) + (double)(hi[1] - lo[1]) * (hi[2] - lo[2]) + (double)(hi[2] - lo[2]) * (hi[0] - lo[0]);
    if (!(ra > 0.0)) return;
    double nv = 1.0, lv = 0.0;
    for (const PtNode4& q : nodes4)
        for (int k = 0; k < 4; ++k) {
            if (q.child[k] == -1) continue;
            const double dx = q.hi[0][k] - q.lo[0][k], dy = q.hi[1][k] - q.lo[1][k], dz = q.hi[2][k] - q.lo[2][k];
            const double ar = (dx * dy + dy * dz + dz * dx) / ra;
            if (q.child[k] >= 0) nv += ar; else lv += ar;
        }
    *node_visits = nv;
    *leaf_visits = lv;
}

#ifndef PT_COLLAPSE4_BY_AREA
#define PT_COLLAPSE4_BY_AREA 0 // measured (profiles/r04_notes.md): 2-10 % fewer quad steps, no time gained (C5 +1.5 %): off
#endif
void pt_bvh_collapse4(const PtBvh& b, std::vector<PtNode4>* out, int32_t* root4, int* depth4)
{
    out->clear();
    *root4 = b.root;
    *depth4 = 0;
    if (b.root < 0) return; // empty scene or a leaf as root: no quad nodes
    {
        const char* e = getenv("PT_COLLAPSE4_BY_AREA"); // A/B switch (tools/): 0 = the fixed two-level collapse of rounds 2-3
        if (e ? e[0] != '0' : PT_COLLAPSE4_BY_AREA) { collapse4_by_area(b, out, root4, depth4); return; }
    }
    struct Item { int32_t node2; int32_t idx4; int depth; };
    std::vector<Item> todo;
    out->emplace_back();
    todo.push_back({b.root, 0, 1});
    *root4 = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        *depth4 = std::max(*depth4, it.depth);
        const PtNode& nd = b.nodes[(size_t)it.node2];
        PtNode4 q;
        for (int a = 0; a < 3; ++a)
            for (int k = 0; k < 4; ++k) q.lo[a][k] = q.hi[a][k] = INFINITY; // never hit (empty slot)
        for (int k = 0; k < 4; ++k) { q.child[k] = -1; q.pad[k] = 0; }
        for (int side = 0; side < 2; ++side) {
            const int32_t c = side ? nd.right : nd.left;
            if (c >= 0) { // internal child: its two children take the slots of this side
                const PtNode& cn = b.nodes[(size_t)c];
                for (int s2 = 0; s2 < 2; ++s2) {
                    const int slot = side * 2 + s2;
                    const int32_t gc = s2 ? cn.right : cn.left;
                    for (int a = 0; a < 3; ++a) { q.lo[a][slot] = cn.lo[a][s2]; q.hi[a][slot] = cn.hi[a][s2]; }
                    if (gc >= 0) {
                        const int32_t idx = (int32_t)out->size();
                        out->emplace_back();
                        q.child[slot] = idx;
                        todo.push_back({gc, idx, it.depth + 1});
                    } else {
                        q.child[slot] = gc;
                    }
                }
            } else if (c < -1) { // leaf child: keeps its own box
                const int slot = side * 2;
                for (int a = 0; a < 3; ++a) { q.lo[a][slot] = nd.lo[a][side]; q.hi[a][slot] = nd.hi[a][side]; }
                q.child[slot] = c;
            }
        }
        (*out)[(size_t)it.idx4] = q;
    }
}

void pt_bvh_collapse8(const PtBvh& b, int wide_leaves, std::vector<PtNode8>* out, int32_t* root8, int* depth8)
{
    out->clear();
    *root8 = b.root;
    *depth8 = 0;
    if (b.root < 0) return; // empty scene or a leaf as root: no oct nodes
    struct Slot { float lo[3], hi[3]; int32_t ref; };
    auto area = [](const Slot& s) {
        const float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    // Wide leaves: in the group walk lane k tests triangle k, so a subtree of <= 7 triangles that lie next to each other in leaf
    // order is ONE leaf step (its binary levels are simply not walked).  sub[i] = {first slot, triangle count} of node i's subtree,
    // count 0 if its triangles are not contiguous (leaf_align padding) or too many.
    struct Range { uint32_t first, count; };
    auto leaf_range = [](int32_t ref) { const uint32_t code = ~(uint32_t)ref; return Range{code >> 3, code & 7u}; };
    std::vector<Range> sub(b.nodes.size(), Range{0, 0});
    {
        std::vector<int32_t> order; // children before parents
        order.reserve(b.nodes.size());
        std::vector<int32_t> st{b.root};
        while (!st.empty()) {
            const int32_t i = st.back();
            st.pop_back();
            order.push_back(i);
            if (b.nodes[(size_t)i].left >= 0) st.push_back(b.nodes[(size_t)i].left);
            if (b.nodes[(size_t)i].right >= 0) st.push_back(b.nodes[(size_t)i].right);
        }
        for (size_t k = order.size(); k-- > 0;) {
            const PtNode& nd = b.nodes[(size_t)order[k]];
            const Range l = nd.left >= 0 ? sub[(size_t)nd.left] : (nd.left < -1 ? leaf_range(nd.left) : Range{0, 0});
            const Range r = nd.right >= 0 ? sub[(size_t)nd.right] : (nd.right < -1 ? leaf_range(nd.right) : Range{0, 0});
            Range m{0, 0};
            if (l.count && r.count && l.count + r.count <= 7u) {
                if (l.first + l.count == r.first) m = Range{l.first, l.count + r.count};
                else if (r.first + r.count == l.first) m = Range{r.first, l.count + r.count};
            }
            sub[(size_t)order[k]] = m;
        }
    }
    if (wide_leaves && sub[(size_t)b.root].count) { // the whole scene is one wide leaf
        *root8 = (int32_t)~((sub[(size_t)b.root].first << 3) | sub[(size_t)b.root].count);
        return;
    }
    auto child_slot = [&](const PtNode& nd, int side) {
        Slot s;
        for (int a = 0; a < 3; ++a) { s.lo[a] = nd.lo[a][side]; s.hi[a] = nd.hi[a][side]; }
        s.ref = side ? nd.right : nd.left;
        if (wide_leaves && s.ref >= 0 && sub[(size_t)s.ref].count) s.ref = (int32_t)~((sub[(size_t)s.ref].first << 3) | sub[(size_t)s.ref].count);
        return s;
    };
    struct Item { int32_t node2; int32_t idx8; int depth; };
    std::vector<Item> todo;
    out->emplace_back();
    todo.push_back({b.root, 0, 1});
    *root8 = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        *depth8 = std::max(*depth8, it.depth);
        Slot slots[8];
        int n = 0;
        {
            const PtNode& nd = b.nodes[(size_t)it.node2];
            for (int side = 0; side < 2; ++side)
                if ((side ? nd.right : nd.left) != -1) slots[n++] = child_slot(nd, side);
        }
        while (n < 8) {
            int best = -1;
            float best_area = -1.0f;
            for (int k = 0; k < n; ++k)
                if (slots[k].ref >= 0) {
                    const float ar = area(slots[k]);
                    if (ar > best_area) { best_area = ar; best = k; }
                }
            if (best < 0) break;
            const PtNode& cn = b.nodes[(size_t)slots[best].ref];
            slots[best] = child_slot(cn, 0); // children keep their relative order: left in place, right appended
            slots[n++] = child_slot(cn, 1);
        }
        PtNode8 q;
        for (int k = 0; k < 8; ++k) {
            for (int a = 0; a < 3; ++a) q.c[k].lo[a] = q.c[k].hi[a] = INFINITY; // never hit (empty slot)
            q.c[k].ref = -1;
            q.c[k].pad = 0;
        }
        for (int k = 0; k < n; ++k) {
            for (int a = 0; a < 3; ++a) { q.c[k].lo[a] = slots[k].lo[a]; q.c[k].hi[a] = slots[k].hi[a]; }
            if (slots[k].ref >= 0) {
                const int32_t idx = (int32_t)out->size();
                out->emplace_back();
                q.c[k].ref = idx;
                todo.push_back({slots[k].ref, idx, it.depth + 1});
            } else {
                q.c[k].ref = slots[k].ref;
            }
        }
        (*out)[(size_t)it.idx8] = q;
    }
}

void pt_bvh_layout(PtBvh* b, int sibling_pairs, int leaf_align)
{
    if (sibling_pairs) layout_sibling_pairs(b);
    layout_align_leaves(b, leaf_align);
}

void pt_bvh_build(const float* positions, int32_t n_tris, int leaf_size, int max_depth, PtBvh* out)
{
    out->nodes.clear();
    out->tris.clear();
    out->root = -1;
    out->depth = 0;
    out->max_leaf = 0;
    out->pad = 0.0f;
    if (n_tris <= 0) return;
    leaf_size = std::max(1, std::min(7, leaf_size));
    max_depth = std::max(2, std::min((int)PT_MAX_STACK, max_depth));
    Pool pool(n_tris > 4 * PT_BUILD_PAR_MIN ? build_threads() : 1); // small scenes: one thread, no pool
    Builder b;
    b.pos = positions;
    b.pool = &pool;
    b.leaf_size = leaf_size;
    b.max_depth = max_depth;
    b.tb.resize(n_tris);
    b.cen.resize((size_t)n_tris * 3);
    b.order.resize(n_tris);
    const int T = pool.size();
    std::vector<Box> part((size_t)T);
    pool.run(T, [&](int t) {
        Box all;
        all.reset();
        for (int i = (int)((long)n_tris * t / T); i < (int)((long)n_tris * (t + 1) / T); ++i) {
            const float* p = positions + (size_t)i * 9;
            Box tbx;
            tbx.reset();
            tbx.grow(p); tbx.grow(p + 3); tbx.grow(p + 6);
            b.tb[i] = tbx;
            for (int a = 0; a < 3; ++a) b.cen[(size_t)i * 3 + a] = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
            b.order[i] = i;
            all.grow(tbx);
        }
        part[(size_t)t] = all;
    });
    Box all;
    all.reset();
    for (const Box& p : part) all.grow(p);
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        ext = std::max(ext, all.mx[a] - all.mn[a]);
        ext = std::max(ext, std::max(std::fabs(all.mn[a]), std::fabs(all.mx[a])));
    }
    // Spatial padding: makes the slab test conservative with respect to every hit the Moeller-Trumbore test can
    // report (its geometric error is orders of magnitude below 1e-5 * extent), so closest-hit is topology-independent.
    b.pad = ext * 1e-5f;
    out->pad = b.pad;
    b.run(n_tris, out);
#if PT_BVH_ROTATIONS > 0
    rotate_tree(out, max_depth, PT_BVH_ROTATIONS);
#endif
    out->tris.resize(n_tris);
    pool.run(T, [&](int t) {
        for (int i = (int)((long)n_tris * t / T); i < (int)((long)n_tris * (t + 1) / T); ++i) {
            const int id = b.order[i];
            PtTri& tr = out->tris[i];
            std::memcpy(tr.p0, positions + (size_t)id * 9, 36);
            tr.id = id;
            tr.material = -1; // filled in by the caller that owns the shading records (pt_api.cpp)
            tr.pad = 0;
        }
    });
}

bool pt_bvh_from_hierarchy(const float* positions, int32_t n_tris, const int32_t* child, const float* box, const int32_t* count, const uint32_t* order, int32_t root,
                           int leaf_size, int max_depth, PtBvh* out)
{
    out->nodes.clear();
    out->tris.clear();
    out->root = -1;
    out->depth = 0;
    out->max_leaf = 0;
    out->pad = 0.0f;
    if (n_tris <= 0) return true;
    leaf_size = std::max(1, std::min(7, leaf_size));
    max_depth = std::max(2, std::min((int)PT_MAX_STACK, max_depth));
    float ext = 0.0f;
    {
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < (size_t)n_tris * 3; ++i)
            for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], positions[i * 3 + a]); mx[a] = std::max(mx[a], positions[i * 3 + a]); }
        for (int a = 0; a < 3; ++a) { ext = std::max(ext, mx[a] - mn[a]); ext = std::max(ext, std::max(std::fabs(mn[a]), std::fabs(mx[a]))); }
    }
    const float pad = ext * 1e-5f; // as pt_bvh_build
    out->pad = pad;
    out->tris.reserve((size_t)n_tris);
    out->nodes.reserve((size_t)n_tris / 2 + 16);
    std::vector<int32_t> walk;
    bool too_deep = false;
    // returns the child reference of hierarchy node p; depth = internal nodes on the path including one created here
    auto emit = [&](auto&& self, int32_t p, int depth) -> int32_t {
        const int cnt = count[p];
        if (cnt <= leaf_size) { // leaf: the triangles of the subtree, depth first
            const uint32_t first = (uint32_t)out->tris.size();
            walk.assign(1, p);
            while (!walk.empty()) {
                const int32_t q = walk.back();
                walk.pop_back();
                if (q < n_tris) {
                    const uint32_t id = order[q];
                    PtTri t;
                    std::memcpy(t.p0, positions + (size_t)id * 9, 36);
                    t.id = (int32_t)id;
                    t.material = -1;
                    t.pad = 0;
                    out->tris.push_back(t);
                } else {
                    walk.push_back(child[2 * (size_t)q + 1]);
                    walk.push_back(child[2 * (size_t)q]);
                }
            }
            out->max_leaf = std::max(out->max_leaf, cnt);
            return (int32_t)~((first << 3) | (uint32_t)cnt);
        }
        if (depth > max_depth) { too_deep = true; return -1; }
        const int32_t idx = (int32_t)out->nodes.size();
        out->nodes.emplace_back();
        out->depth = std::max(out->depth, depth);
        const int32_t l = child[2 * (size_t)p], r = child[2 * (size_t)p + 1];
        const int32_t lc = self(self, l, depth + 1);
        const int32_t rc = too_deep ? -1 : self(self, r, depth + 1);
        PtNode& nd = out->nodes[(size_t)idx];
        for (int a = 0; a < 3; ++a) {
            nd.lo[a][0] = box[6 * (size_t)l + a] - pad; nd.hi[a][0] = box[6 * (size_t)l + 3 + a] + pad;
            nd.lo[a][1] = box[6 * (size_t)r + a] - pad; nd.hi[a][1] = box[6 * (size_t)r + 3 + a] + pad;
        }
        nd.left = lc;
        nd.right = rc;
        nd.pad[0] = nd.pad[1] = 0;
        return idx;
    };
    out->root = emit(emit, root, 1);
    return !too_deep && (int32_t)out->tris.size() == n_tris;
}

// ---- host mirror of the kernel traversal (validation of the builder; same arithmetic as pt_kernel.hip) ----

namespace {
inline float hfma(float a, float b, float c) { return std::fma(a, b, c); }
inline float hmin(float a, float b) { return (b != b || a < b) ? a : b; }
inline float hmax(float a, float b) { return (b != b || a > b) ? a : b; }
struct hv3 { float x, y, z; };
inline hv3 hsub(hv3 a, hv3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float hdot(hv3 a, hv3 b) { return hfma(a.z, b.z, hfma(a.y, b.y, a.x * b.x)); }
inline hv3 hcross(hv3 a, hv3 b) { return {hfma(a.y, b.z, -(a.z * b.y)), hfma(a.z, b.x, -(a.x * b.z)), hfma(a.x, b.y, -(a.y * b.x))}; }
inline bool hbox(const float* mn, const float* mx, hv3 o, hv3 inv, float tmin, float tbest, float* tn_out)
{
    float t0x = (mn[0] - o.x) * inv.x, t1x = (mx[0] - o.x) * inv.x;
    float t0y = (mn[1] - o.y) * inv.y, t1y = (mx[1] - o.y) * inv.y;
    float t0z = (mn[2] - o.z) * inv.z, t1z = (mx[2] - o.z) * inv.z;
    float tn = hmax(hmax(hmin(t0x, t1x), hmin(t0y, t1y)), hmax(hmin(t0z, t1z), tmin));
    float tf = hmin(hmin(hmax(t0x, t1x), hmax(t0y, t1y)), hmin(hmax(t0z, t1z), tbest));
    *tn_out = tn;
    return tn <= tf * 1.0000004f;
}
} // namespace

bool pt_bvh_closest_hit_host(const PtBvh& bvh, const float org[3], const float dir[3], float tmin, float tmax, float* t_out, float* u_out,
                             float* v_out, int32_t* prim)
{
    hv3 o{org[0], org[1], org[2]}, d{dir[0], dir[1], dir[2]};
    hv3 inv{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    float bt = tmax, bu = 0.0f, bv = 0.0f;
    int32_t bid = 0x7fffffff;
    int32_t stack[PT_MAX_STACK + 8];
    int sp = 0;
    int32_t cur = bvh.root;
    for (;;) {
        if (cur >= 0) {
            const PtNode& nd = bvh.nodes[cur];
            float tl, tr;
            const float lmin[3] = {nd.lo[0][0], nd.lo[1][0], nd.lo[2][0]}, lmax[3] = {nd.hi[0][0], nd.hi[1][0], nd.hi[2][0]};
            const float rmin[3] = {nd.lo[0][1], nd.lo[1][1], nd.lo[2][1]}, rmax[3] = {nd.hi[0][1], nd.hi[1][1], nd.hi[2][1]};
            bool hl = hbox(lmin, lmax, o, inv, tmin, bt, &tl);
            bool hr = hbox(rmin, rmax, o, inv, tmin, bt, &tr);
            if (hl && hr) {
                bool swap = tr < tl;
                stack[sp++] = swap ? nd.left : nd.right;
                cur = swap ? nd.right : nd.left;
                continue;
            } else if (hl) { cur = nd.left; continue; }
            else if (hr) { cur = nd.right; continue; }
        } else if (cur != -1) {
            uint32_t code = ~(uint32_t)cur;
            int first = (int)(code >> 3), count = (int)(code & 7u);
            for (int i = 0; i < count; ++i) {
                const PtTri& tr = bvh.tris[first + i];
                hv3 p0{tr.p0[0], tr.p0[1], tr.p0[2]}, p1{tr.p1[0], tr.p1[1], tr.p1[2]}, p2{tr.p2[0], tr.p2[1], tr.p2[2]};
                hv3 e1 = hsub(p1, p0), e2 = hsub(p2, p0);
                hv3 pv = hcross(d, e2);
                float det = hdot(e1, pv);
                float idet = 1.0f / det;
                hv3 tv = hsub(o, p0);
                float u = hdot(tv, pv) * idet;
                hv3 qv = hcross(tv, e1);
                float v = hdot(d, qv) * idet;
                float t = hdot(e2, qv) * idet;
                if (u >= 0.0f && v >= 0.0f && u + v <= 1.0f && t > tmin && (t < bt || (t == bt && tr.id < bid))) {
                    bt = t; bu = u; bv = v; bid = tr.id;
                }
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    *t_out = bt; *u_out = bu; *v_out = bv;
    *prim = bid == 0x7fffffff ? -1 : bid;
    return bid != 0x7fffffff;
}
