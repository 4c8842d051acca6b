// ORBextractor::DistributeOctTree (SF/src/ORBextractor.cc:529-753, ExtractorNode::DivideNode :454-510, compareNodes :512-527) on the
// device: one workgroup per (image, pyramid level), all of them in one launch.  The reference walks a std::list of nodes and, per
// node, partitions a std::vector of keypoints -- sequential code whose RESULT ORDER (the list order at the end) defines the keypoint
// indices downstream.  Here every round of the walk is level-synchronous:
//   * every key looks up its node and computes its quadrant; ONE prefix sum along the key array over the packed quadrant indicators
//     (3 x 21 bits in a 64-bit word, the fourth quadrant is the remainder) gives both the populations of the children (difference of
//     the prefix at the node's two ends) and the place of every key inside its child (difference to the prefix at the node's first
//     key): keys keep their relative order, as the push_back loop of DivideNode does;
//   * a prefix sum over the nodes in processing order gives every child its place in the new list -- children are pushed to the
//     FRONT in the order UL, UR, BL, BR while the walk goes on, so the new list is [children of the last divided node, BR..UL | ... |
//     children of the first divided node | the nodes that were not divided, in their old order].
// The closing phase of the reference ("size + 3 * nToExpand > N": divide the largest nodes first until N nodes exist) is the same
// round with the nodes to divide taken from the back of a list sorted by (population, UL.x).  That sort is std::sort in the reference:
// not stable, so which of two nodes with equal population and equal UL.x comes first depends on libstdc++'s introsort -- restated
// here move for move (median-of-three partition, depth limit 2 log2 n with the heap-sort fallback, final insertion sort with the
// threshold 16), run by one lane on a few hundred records in LDS.
// The candidate with the largest response of every final node is emitted in list order (:731-752).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "launch.hpp"
#pragma clang fp contract(off)
#include <stdint.h>

#include "intro_sort.hpp"
#include "orb_device.hpp"

namespace tc2li {

namespace {

struct QNode { int16_t ulx, uly, brx, bry; int32_t begin, count; };  // 16 bytes
struct QRec { uint32_t key; int32_t node; };                        // vSizeAndPointerToNode entry: key = population << 12 | UL.x (compareNodes' order)
constexpr int kSortCap = 1536;                                        // records sorted in LDS when they do not fit the register form
constexpr int kSortRegLanes = 128;                                    // register form: 64 lanes x 2 records
constexpr unsigned long long kF21 = (1ull << 21) - 1;

__device__ __forceinline__ int q_cx(uint32_t c) { return (int)((c >> 8) & 0xfff); }
__device__ __forceinline__ int q_cy(uint32_t c) { return (int)(c >> 20); }
__device__ __forceinline__ int q_cr(uint32_t c) { return (int)(c & 0xff); }
__device__ __forceinline__ int q_quadrant(const QNode& p, uint32_t c) {
    const int mx = p.ulx + (int)ceilf((float)(p.brx - p.ulx) / 2), my = p.uly + (int)ceilf((float)(p.bry - p.uly) / 2);
    return (q_cx(c) < mx ? 0 : 1) + (q_cy(c) < my ? 0 : 2);
}

// ---- libstdc++ std::sort(first, last, comp) on QRec, comp = (population, UL.x) ascending ----------------------------------------
// The algorithm is scalar; what it costs is the latency of every dependent array access.  Two homes for the array:
//   ArrMem: LDS (or global) memory, run by one lane -- ~100 cycles per access;
//   ArrReg: up to 128 records, record i in lane i % 64 of register i / 64 of ONE wavefront whose lanes all execute the same scalar
//           program: a read is a v_readlane with a scalar lane index, a write a compare-and-select on the lane id -- a few cycles.
struct ArrMem {
    QRec* p;
    __device__ __forceinline__ QRec get(int i) const { return p[i]; }
    __device__ __forceinline__ void set(int i, const QRec& r) const { p[i] = r; }
};
struct ArrReg {
    uint32_t &k0, &k1;
    int32_t &n0, &n1;
    __device__ __forceinline__ QRec get(int i) const {
        const int lane = __builtin_amdgcn_readfirstlane(i & 63);
        const bool hi = __builtin_amdgcn_readfirstlane(i >> 6) != 0;
        const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)k0, lane), b = (uint32_t)__builtin_amdgcn_readlane((int)k1, lane);
        const int32_t c = __builtin_amdgcn_readlane(n0, lane), d = __builtin_amdgcn_readlane(n1, lane);
        return QRec{hi ? b : a, hi ? d : c};
    }
    __device__ __forceinline__ void set(int i, const QRec& q) const {
        const int lane = __builtin_amdgcn_readfirstlane(i & 63);
        const bool hi = __builtin_amdgcn_readfirstlane(i >> 6) != 0;
        const bool me = (int)(threadIdx.x & 63) == lane;
        k0 = me && !hi ? q.key : k0; k1 = me && hi ? q.key : k1;
        n0 = me && !hi ? q.node : n0; n1 = me && hi ? q.node : n1;
    }
};
__device__ __forceinline__ bool rec_less(const QRec& a, const QRec& b) { return a.key < b.key; }
template <class A> __device__ __forceinline__ void rec_swap(const A& v, int a, int b) { const QRec t = v.get(a), u = v.get(b); v.set(a, u); v.set(b, t); }
template <class A> __device__ void std_push_heap(const A& f, int o, int hole, int top, QRec value) {
    int parent = (hole - 1) / 2;
    while (hole > top && rec_less(f.get(o + parent), value)) { f.set(o + hole, f.get(o + parent)); hole = parent; parent = (hole - 1) / 2; }
    f.set(o + hole, value);
}
template <class A> __device__ void std_adjust_heap(const A& f, int o, int hole, int len, QRec value) {
    const int top = hole;
    int second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (rec_less(f.get(o + second), f.get(o + second - 1))) second--;
        f.set(o + hole, f.get(o + second));
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        f.set(o + hole, f.get(o + second - 1));
        hole = second - 1;
    }
    std_push_heap(f, o, hole, top, value);
}
template <class A> __device__ void std_heap_sort(const A& f, int o, int len) {  // __partial_sort(first, last, last): __make_heap + __sort_heap
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; --parent) {
            const QRec v = f.get(o + parent);
            std_adjust_heap(f, o, parent, len, v);
            if (parent == 0) break;
        }
    }
    for (int last = len; last > 1;) {
        --last;
        const QRec v = f.get(o + last);
        f.set(o + last, f.get(o));
        std_adjust_heap(f, o, 0, last, v);
    }
}
template <class A> __device__ void std_unguarded_linear_insert(const A& v, int last) {
    const QRec val = v.get(last);
    int next = last - 1;
    for (;;) {
        const QRec nx = v.get(next);
        if (!rec_less(val, nx)) break;
        v.set(last, nx); last = next; --next;
    }
    v.set(last, val);
}
template <class A> __device__ void std_insertion_sort(const A& v, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const QRec val = v.get(i);
        if (rec_less(val, v.get(first))) {
            for (int k = i; k > first; --k) v.set(k, v.get(k - 1));
            v.set(first, val);
        } else {
            std_unguarded_linear_insert(v, i);
        }
    }
}
// stack: 3 x 64 ints of LDS for the right-hand ranges of __introsort_loop (its recursion)
template <class A> __device__ void std_sort(const A& v, int n, int* stack) {
    if (n <= 0) return;
    int depth0 = 0;
    for (int t = n; t > 1; t >>= 1) ++depth0;
    depth0 *= 2;
    int sp = 0;
    stack[0] = 0; stack[1] = n; stack[2] = depth0; ++sp;
    while (sp > 0) {
        --sp;
        int first = stack[3 * sp], last = stack[3 * sp + 1], depth = stack[3 * sp + 2];
        while (last - first > 16) {
            if (depth == 0) { std_heap_sort(v, first, last - first); break; }
            --depth;
            // __unguarded_partition_pivot: the median of (first + 1, mid, last - 1) goes to first
            const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
            const QRec va = v.get(a), vb = v.get(b), vc = v.get(c);
            if (rec_less(va, vb)) {
                if (rec_less(vb, vc)) rec_swap(v, first, b);
                else if (rec_less(va, vc)) rec_swap(v, first, c);
                else rec_swap(v, first, a);
            } else if (rec_less(va, vc)) rec_swap(v, first, a);
            else if (rec_less(vb, vc)) rec_swap(v, first, c);
            else rec_swap(v, first, b);
            int lo = first + 1, hi = last;
            const QRec pivot = v.get(first);  // *first does not move during the partition
            for (;;) {
                while (rec_less(v.get(lo), pivot)) ++lo;
                --hi;
                while (rec_less(pivot, v.get(hi))) --hi;
                if (!(lo < hi)) break;
                rec_swap(v, lo, hi);
                ++lo;
            }
            // the reference recurses into [cut, last) and loops on [first, cut): disjoint ranges, any processing order gives the same array.
            // The depth budget bounds the pending ranges to 2 log2 n < 64
            stack[3 * sp] = lo; stack[3 * sp + 1] = last; stack[3 * sp + 2] = depth; ++sp;
            last = lo;
        }
    }
    if (n > 16) {  // __final_insertion_sort
        std_insertion_sort(v, 0, 16);
        for (int i = 16; i != n; ++i) std_unguarded_linear_insert(v, i);
    } else {
        std_insertion_sort(v, 0, n);
    }
}

// Exclusive prefix sums over the block; every thread of the block calls them the same number of times.
template <int T, typename V>
__device__ __forceinline__ V block_excl_scan(V v, V* s_wave, V& total) {
    const int lane = threadIdx.x & 63, wave = wave_in_block();
    V incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const V t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    V base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < T / 64; ++k) { const V w = s_wave[k]; base += k < wave ? w : (V)0; tot += w; }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

struct ScratchLayout {
    size_t scanq, keys_a, keys_b, nodeof_a, nodeof_b, kq, nodes_a, nodes_b, cnt, cidx, procpos, order, newpos, rec_a, rec_b, total;
    __host__ __device__ ScratchLayout(int K, int MN) {
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
        scanq = take((size_t)(K + 1) * 8);
        keys_a = take((size_t)K * 4); keys_b = take((size_t)K * 4);
        nodeof_a = take((size_t)K * 4); nodeof_b = take((size_t)K * 4);
        kq = take((size_t)K);
        nodes_a = take((size_t)MN * sizeof(QNode)); nodes_b = take((size_t)MN * sizeof(QNode));
        cnt = take((size_t)MN * 16); cidx = take((size_t)MN * 16);
        procpos = take((size_t)MN * 4); order = take((size_t)MN * 4); newpos = take((size_t)MN * 4);
        rec_a = take((size_t)MN * sizeof(QRec)); rec_b = take((size_t)MN * sizeof(QRec));
        total = o;
    }
};

// ---- the sorted form's work space (k_quadtree_sorted below) ----
constexpr int kPathDepth = 12;      // 12-bit coordinates: after 12 halvings a node is one pixel wide and high and every key goes to child 0
constexpr int kQuadClasses = 3;       // LDS classes of the sorted form: a job runs in the first class it fits
__host__ __device__ constexpr int quad_class_threads(int c) { return c == 0 ? 256 : c == 1 ? 512 : 1024; }
__host__ __device__ constexpr size_t quad_class_lds(int c) { return (c == 0 ? 38 : c == 1 ? 76 : 156) * (size_t)1024; }  // dynamic LDS per workgroup: four, two, one per CU
struct SNode { int16_t ulx, uly, brx, bry; int32_t begin, cd; };  // cd = population | depth << 24
__device__ __forceinline__ int sn_count(const SNode& n) { return n.cd & 0xffffff; }
__device__ __forceinline__ int sn_depth(const SNode& n) { return (int)((uint32_t)n.cd >> 24); }

struct SortedLayout {
    size_t codes, idx, codes_b, idx_b, counters, nodes_a, nodes_b, cnt, cidx, procpos, order, newpos, rec_a, rec_b, s_key, s_idx, s_u16, s_flag, total;
    __host__ __device__ SortedLayout(int n, int MN, int T) {
        size_t o = 0;
        auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 15) & ~(size_t)15; return at; };
        codes = take((size_t)n * 4); idx = take((size_t)n * 2);
        const size_t b = o;
        codes_b = take((size_t)n * 4); idx_b = take((size_t)n * 2); counters = take((size_t)16 * T * 2);
        const size_t end_sort = o;
        o = b;  // after the sort its second buffer and the counters are free
        nodes_a = take((size_t)MN * sizeof(SNode)); nodes_b = take((size_t)MN * sizeof(SNode));
        cnt = take((size_t)MN * 16); cidx = take((size_t)MN * 16);
        procpos = take((size_t)MN * 4); order = take((size_t)MN * 4); newpos = take((size_t)MN * 4);
        rec_a = take((size_t)MN * sizeof(QRec)); rec_b = take((size_t)MN * sizeof(QRec));
        // the closing sort (intro_sort.hpp): keys, payload, seven 16-bit index arrays of MN + 2 entries, flags
        s_key = take((size_t)MN * 4); s_idx = take((size_t)MN * 4); s_u16 = take((size_t)7 * (MN + 2) * 2); s_flag = take((size_t)MN);
        total = o > end_sort ? o : end_sort;
    }
};
// Jobs are stored (image, level) with `nlevels` levels per image, and the hardware deals consecutive workgroups to the eight XCDs in
// turn: with eight levels every level's jobs -- the large level 0 ones, too -- would meet on ONE XCD (measured: the level-0 class ran
// four rounds on 32 CUs while 224 idled).  Workgroup b takes image b % M, level b / M: neighbours are the same level of different images.
__device__ __forceinline__ int quad_job_of_block(int b, int n_blocks, int nlevels) {
    if (nlevels <= 1 || n_blocks % nlevels) return b;
    const int M = n_blocks / nlevels;
    return (b % M) * nlevels + b / M;
}
constexpr int kQuadMaxRun = 15;  // keys per lane in the sort (they stay in registers during a pass); the LDS classes hold fewer anyway
// Does the job fit class c?  Never: more than 256 columns, key indices beyond 16 bits, more than kQuadMaxRun keys per lane
__host__ __device__ inline bool quad_fits(int c, int ncand, int MN, int n_ini) {
    const int T = quad_class_threads(c);
    if (ncand > 65535 || n_ini > 256 || (((ncand + T - 1) / T) | 1) > kQuadMaxRun) return false;
    return SortedLayout(ncand, MN, T).total <= quad_class_lds(c);
}
// the class a job runs in; kQuadClasses: none (k_quadtree takes it)
__host__ __device__ inline int quad_class_of(int ncand, int MN, int n_ini, int first_class) {
    for (int c = first_class; c < kQuadClasses; ++c) if (quad_fits(c, ncand, MN, n_ini)) return c;
    return kQuadClasses;
}

}  // namespace

template <int T>
__global__ __launch_bounds__(T) void k_quadtree(const QuadJob* __restrict__ jobs, const uint32_t* __restrict__ dense, const int32_t* __restrict__ level_counts,
                                               uint8_t* __restrict__ scratch, uint32_t* __restrict__ picked, int32_t* __restrict__ picked_count,
                                               int32_t* __restrict__ status, int after_sorted /* 0: all jobs; else 1 + the first class that ran: only the jobs the sorted form left */, int nlevels) {
    const QuadJob J = jobs[quad_job_of_block(blockIdx.x, gridDim.x, nlevels)];
    const int tid = threadIdx.x;
    const uint32_t* cand = dense + J.cand_off;
    const int ncand = level_counts[J.count_idx];
    uint32_t* out = picked + J.out_off;
    int32_t* const out_count = picked_count + J.count_idx;
    __shared__ int s_wave[T / 64];
    __shared__ unsigned long long s_wave64[T / 64];
    __shared__ int s_cut;
    __shared__ int s_stack[3 * 64];
    __shared__ QRec s_rec[kSortCap];
    const int K = J.max_keys, MN = J.max_nodes, N = J.n_target;
    if (after_sorted) {  // the trivial outcomes and the jobs that fit were k_quadtree_sorted's
        const int n_ini0 = (int)roundf((float)(J.max_x - J.min_x) / (float)(J.max_y - J.min_y));
        if (ncand <= 0 || ncand > K || n_ini0 <= 0 || n_ini0 > MN || quad_class_of(ncand, MN, n_ini0, after_sorted - 1) < kQuadClasses) return;
    }
    if (ncand <= 0 || ncand > K) {
        if (tid == 0) { *out_count = 0; if (ncand > K) atomicMax(status, 1); }
        return;
    }
    const ScratchLayout lay(K, MN);
    uint8_t* base = scratch + J.scratch_off;
    unsigned long long* scanq = reinterpret_cast<unsigned long long*>(base + lay.scanq);
    int32_t* keys = reinterpret_cast<int32_t*>(base + lay.keys_a);
    int32_t* keys2 = reinterpret_cast<int32_t*>(base + lay.keys_b);
    int32_t* nodeof = reinterpret_cast<int32_t*>(base + lay.nodeof_a);
    int32_t* nodeof2 = reinterpret_cast<int32_t*>(base + lay.nodeof_b);
    uint8_t* kq = base + lay.kq;
    QNode* nodes = reinterpret_cast<QNode*>(base + lay.nodes_a);
    QNode* nodes2 = reinterpret_cast<QNode*>(base + lay.nodes_b);
    int32_t* cnt = reinterpret_cast<int32_t*>(base + lay.cnt);          // [MN][4] populations of the children of a divided node
    int32_t* cidx = reinterpret_cast<int32_t*>(base + lay.cidx);        // [MN][4] place of the child in the new list
    int32_t* procpos = reinterpret_cast<int32_t*>(base + lay.procpos);  // [MN] place of the node in this round's processing order, -1: not divided
    int32_t* order = reinterpret_cast<int32_t*>(base + lay.order);      // [MN] the processing order
    int32_t* newpos = reinterpret_cast<int32_t*>(base + lay.newpos);    // [MN] children pushed before this node's / new place of a node that stays
    QRec* recs = reinterpret_cast<QRec*>(base + lay.rec_a);             // the multi-key children of the last round, in creation order
    QRec* recs2 = reinterpret_cast<QRec*>(base + lay.rec_b);

    // ---- initial nodes (:533-560): nIni columns, the keys dealt to them in candidate order; empty columns are erased ----
    const int W = J.max_x - J.min_x, H = J.max_y - J.min_y;
    const int n_ini = (int)roundf((float)W / (float)H);
    if (n_ini <= 0 || n_ini > MN) {  // the reference would divide by zero for n_ini == 0
        if (tid == 0) { *out_count = 0; if (n_ini > MN) atomicMax(status, 2); }
        return;
    }
    const float hX = (float)W / (float)n_ini;
    for (int k = tid; k < ncand; k += T) {
        size_t b = (size_t)((float)q_cx(cand[k]) / hX);
        if (b >= (size_t)n_ini) b = (size_t)n_ini - 1;  // unreachable for in-range candidates
        nodeof2[k] = (int)b;
    }
    __syncthreads();
    for (int s0 = 0; s0 < n_ini; s0 += 3) {  // three columns per pass
        unsigned long long carry = 0;
        for (int c0 = 0; c0 < ncand; c0 += T) {
            const int k = c0 + tid;
            const int sl = k < ncand ? nodeof2[k] - s0 : -1;
            const bool mine = sl >= 0 && sl < 3;
            unsigned long long tot;
            const unsigned long long ex = carry + block_excl_scan<T>(mine ? 1ull << (21 * sl) : 0ull, s_wave64, tot);
            if (mine) keys2[k] = (int)((ex >> (21 * sl)) & kF21);  // rank inside the column
            carry += tot;
        }
        if (tid < 3 && s0 + tid < n_ini) cnt[s0 + tid] = (int)((carry >> (21 * tid)) & kF21);
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0, n = 0;
        for (int i = 0; i < n_ini; ++i) {
            const int c = cnt[i];
            procpos[i] = c > 0 ? n : -1;  // column -> node
            cidx[i] = acc;                // first key of the column
            if (c > 0) nodes[n++] = QNode{(int16_t)(int)(hX * (float)i), 0, (int16_t)(int)(hX * (float)(i + 1)), (int16_t)H, acc, c};
            acc += c;
        }
        s_cut = n;
    }
    __syncthreads();
    for (int k = tid; k < ncand; k += T) {
        const int sl = nodeof2[k], pos = cidx[sl] + keys2[k];
        keys[pos] = k;
        nodeof[pos] = procpos[sl];
    }
    int size = s_cut, phase = 0, n_rec = 0;
    __syncthreads();

    for (int guard = 0; guard < 100000 && phase != 2; ++guard) {
        // ---- 1. the nodes to divide and their order: every multi-key node in list order, or (closing phase) the sorted records from the back ----
        int n_proc;
        if (phase == 0) {
            int carry = 0;
            for (int i0 = 0; i0 < size; i0 += T) {
                const int i = i0 + tid;
                const int f = i < size && nodes[i].count > 1 ? 1 : 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(f, s_wave, tot);
                if (i < size) procpos[i] = f ? ex : -1;
                if (f) order[ex] = i;
                carry += tot;
            }
            n_proc = carry;
        } else {
            n_proc = n_rec;
            for (int i = tid; i < size; i += T) procpos[i] = -1;
            __syncthreads();
            for (int j = tid; j < n_proc; j += T) { const int nd = recs[n_rec - 1 - j].node; order[j] = nd; procpos[nd] = j; }
        }
        __syncthreads();
        // ---- 2. quadrant of every key of those nodes; prefix sums of the indicators along the key array ----
        // four consecutive keys per lane: one block-wide scan (two barriers) per 4 T keys, vector loads / stores
        {
            unsigned long long carry = 0;
            for (int c0 = 0; c0 < ncand; c0 += 4 * T) {
                const int k0 = c0 + 4 * tid;
                int nd4[4] = {0, 0, 0, 0}, key4[4] = {0, 0, 0, 0};
                const bool full = k0 + 3 < ncand;
                if (full) {
                    const int4 a = *reinterpret_cast<const int4*>(nodeof + k0), bkey = *reinterpret_cast<const int4*>(keys + k0);
                    nd4[0] = a.x; nd4[1] = a.y; nd4[2] = a.z; nd4[3] = a.w;
                    key4[0] = bkey.x; key4[1] = bkey.y; key4[2] = bkey.z; key4[3] = bkey.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k0 + j < ncand) { nd4[j] = nodeof[k0 + j]; key4[j] = keys[k0 + j]; }
                }
                unsigned long long v[4] = {0, 0, 0, 0};
                uint32_t q4 = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (k0 + j < ncand && procpos[nd4[j]] >= 0) {
                        const int q = q_quadrant(nodes[nd4[j]], cand[key4[j]]);
                        q4 |= (uint32_t)q << (8 * j);
                        if (q < 3) v[j] = 1ull << (21 * q);
                    }
                unsigned long long tot;
                unsigned long long run = carry + block_excl_scan<T>(v[0] + v[1] + v[2] + v[3], s_wave64, tot);
                if (full) {
                    *reinterpret_cast<uint32_t*>(kq + k0) = q4;
                    ulonglong2 lo, hi;
                    lo.x = run; lo.y = run + v[0]; hi.x = lo.y + v[1]; hi.y = hi.x + v[2];
                    *reinterpret_cast<ulonglong2*>(scanq + k0) = lo;
                    *reinterpret_cast<ulonglong2*>(scanq + k0 + 2) = hi;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k0 + j < ncand) { kq[k0 + j] = (uint8_t)(q4 >> (8 * j)); scanq[k0 + j] = run; run += v[j]; }
                }
                carry += tot;
            }
            if (tid == 0) scanq[ncand] = carry;
        }
        __syncthreads();
        // ---- 3. populations of the children ----
        for (int j = tid; j < n_proc; j += T) {
            const int nd = order[j];
            const QNode p = nodes[nd];
            const unsigned long long d = scanq[p.begin + p.count] - scanq[p.begin];
            const int c0 = (int)(d & kF21), c1 = (int)((d >> 21) & kF21), c2 = (int)((d >> 42) & kF21);
            cnt[4 * nd] = c0; cnt[4 * nd + 1] = c1; cnt[4 * nd + 2] = c2; cnt[4 * nd + 3] = p.count - c0 - c1 - c2;
        }
        if (tid == 0) s_cut = n_proc;
        __syncthreads();
        // ---- 4. closing phase: the walk stops as soon as the list holds N nodes (:700-701); a division adds children - 1 ----
        if (phase == 1) {
            int carry = 0;
            for (int j0 = 0; j0 < n_proc; j0 += T) {
                const int j = j0 + tid;
                int grow = 0;
                if (j < n_proc) { for (int q = 0; q < 4; ++q) grow += cnt[4 * order[j] + q] > 0; grow -= 1; }
                int tot;
                const int ex = carry + block_excl_scan<T>(grow, s_wave, tot);
                if (j < n_proc && size + ex + grow >= N) atomicMin(&s_cut, j + 1);
                carry += tot;
            }
            __syncthreads();
            const int m = s_cut;
            for (int j = m + tid; j < n_proc; j += T) procpos[order[j]] = -1;  // not reached
            n_proc = m;
            __syncthreads();
        }
        // ---- 5. places in the new list ----
        int n_children;
        {
            int carry = 0;
            for (int j0 = 0; j0 < n_proc; j0 += T) {
                const int j = j0 + tid;
                int ch = 0;
                if (j < n_proc) for (int q = 0; q < 4; ++q) ch += cnt[4 * order[j] + q] > 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(ch, s_wave, tot);
                if (j < n_proc) newpos[order[j]] = ex;
                carry += tot;
            }
            n_children = carry;
        }
        int new_size;
        {
            int carry = 0;
            for (int i0 = 0; i0 < size; i0 += T) {
                const int i = i0 + tid;
                const int f = i < size && procpos[i] < 0 ? 1 : 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(f, s_wave, tot);
                if (f) newpos[i] = n_children + ex;
                carry += tot;
            }
            new_size = n_children + carry;
        }
        if (new_size > MN) { if (tid == 0) { atomicMax(status, 3); *out_count = 0; } return; }
        for (int i = tid; i < size; i += T) {
            const QNode p = nodes[i];
            if (procpos[i] < 0) { nodes2[newpos[i]] = p; continue; }
            const int mx = p.ulx + (int)ceilf((float)(p.brx - p.ulx) / 2), my = p.uly + (int)ceilf((float)(p.bry - p.uly) / 2);
            int pushed = newpos[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cnt[4 * i + q];
                if (c == 0) { cidx[4 * i + q] = -1; continue; }
                const int at = n_children - 1 - pushed;  // push_front: a later push lies nearer to the front
                nodes2[at] = QNode{(int16_t)((q & 1) ? mx : p.ulx), (int16_t)((q & 2) ? my : p.uly), (int16_t)((q & 1) ? p.brx : mx),
                                   (int16_t)((q & 2) ? p.bry : my), 0, c};
                cidx[4 * i + q] = at;
                ++pushed;
            }
        }
        __syncthreads();
        {  // first key of every new node
            int carry = 0;
            for (int i0 = 0; i0 < new_size; i0 += T) {
                const int i = i0 + tid;
                const int c = i < new_size ? nodes2[i].count : 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(c, s_wave, tot);
                if (i < new_size) nodes2[i].begin = ex;
                carry += tot;
            }
        }
        __syncthreads();
        // ---- 6. keys to their new nodes ----
        for (int k = tid; k < ncand; k += T) {
            const int nd = nodeof[k];
            const QNode p = nodes[nd];
            int nn, rank;
            if (procpos[nd] < 0) {
                nn = newpos[nd];
                rank = k - p.begin;
            } else {
                const int q = kq[k];
                const unsigned long long d = scanq[k] - scanq[p.begin];
                const int r0 = (int)(d & kF21), r1 = (int)((d >> 21) & kF21), r2 = (int)((d >> 42) & kF21);
                rank = q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : (k - p.begin) - r0 - r1 - r2;
                nn = cidx[4 * nd + q];
            }
            const int pos = nodes2[nn].begin + rank;
            keys2[pos] = keys[k];
            nodeof2[pos] = nn;
        }
        // ---- 7. the multi-key children of this round, in creation order ----
        int n_expand;
        {
            int carry = 0;
            for (int j0 = 0; j0 < n_proc; j0 += T) {
                const int j = j0 + tid;
                int m = 0;
                if (j < n_proc) for (int q = 0; q < 4; ++q) m += cnt[4 * order[j] + q] > 1;
                int tot;
                const int ex = carry + block_excl_scan<T>(m, s_wave, tot);
                if (j < n_proc) {
                    int at = ex;
                    const int nd = order[j];
                    for (int q = 0; q < 4; ++q) {
                        const int c = cnt[4 * nd + q];
                        if (c > 1) { const int nn = cidx[4 * nd + q]; recs2[at++] = QRec{((uint32_t)c << 12) | (uint32_t)nodes2[nn].ulx, nn}; }
                    }
                }
                carry += tot;
            }
            n_expand = carry;
        }
        __syncthreads();
        // ---- 8. what comes next (:651-728) ----
        int next_phase;
        if (new_size >= N || new_size == size) next_phase = 2;
        else if (phase == 1 || new_size + n_expand * 3 > N) next_phase = 1;
        else next_phase = 0;
        if (next_phase == 1) {  // sort(vPrevSizeAndPointerToNode.begin(), vPrevSizeAndPointerToNode.end(), compareNodes)
            if (n_expand <= kSortRegLanes) {
                if (tid < 64) {  // wavefront 0, all lanes: the array lives in its registers
                    const QRec pad{0xffffffffu, 0};
                    const QRec q0 = tid < n_expand ? recs2[tid] : pad, q1 = 64 + tid < n_expand ? recs2[64 + tid] : pad;
                    uint32_t k0 = q0.key, k1 = q1.key;
                    int32_t n0 = q0.node, n1 = q1.node;
                    std_sort(ArrReg{k0, k1, n0, n1}, n_expand, s_stack);
                    if (tid < n_expand) recs2[tid] = QRec{k0, n0};
                    if (64 + tid < n_expand) recs2[64 + tid] = QRec{k1, n1};
                }
            } else if (n_expand <= kSortCap) {
                for (int j = tid; j < n_expand; j += T) s_rec[j] = recs2[j];
                __syncthreads();
                if (tid == 0) std_sort(ArrMem{s_rec}, n_expand, s_stack);
                __syncthreads();
                for (int j = tid; j < n_expand; j += T) recs2[j] = s_rec[j];
            } else if (tid == 0) {
                std_sort(ArrMem{recs2}, n_expand, s_stack);
            }
        }
        { int32_t* t = keys; keys = keys2; keys2 = t; t = nodeof; nodeof = nodeof2; nodeof2 = t; }
        { QNode* t = nodes; nodes = nodes2; nodes2 = t; }
        { QRec* t = recs; recs = recs2; recs2 = t; }
        size = new_size; phase = next_phase; n_rec = n_expand;
        __syncthreads();
    }
    // ---- the best candidate of every node, in list order (:731-752) ----
    if (size > J.out_cap) { if (tid == 0) { atomicMax(status, 4); *out_count = 0; } return; }
    for (int i = tid; i < size; i += T) {
        const QNode n = nodes[i];
        int best = keys[n.begin], best_r = q_cr(cand[best]);
        for (int k = 1; k < n.count; ++k) {
            const int key = keys[n.begin + k];
            const int r = q_cr(cand[key]);
            if (r > best_r) { best = key; best_r = r; }
        }
        out[i] = cand[best];
    }
    if (tid == 0) *out_count = size;
}

// ---- the same walk over keys sorted ONCE by their path (the default since round 3) ----
// The rectangle of a node is a function of its column and of the quadrants taken on the way down -- not of the data: DivideNode halves
// with ceil() whatever the node holds.  So every candidate's whole path (column, then kPathDepth quadrant digits) is computed up
// front and the keys are sorted by it once (LSD radix, 4 bits per pass; a lane keeps its run of <= 15 keys, its 16 digit counts and
// its 16 running places in registers during a pass, LDS sees one counter matrix and the scatter).  From then on a node of ANY depth is
// a contiguous range [begin, begin + count) of the sorted array, and a round of the walk touches nodes only: the populations of a
// node's children are three binary searches for the digit boundaries inside its range (no pass over the keys, no key moves), the list
// bookkeeping is the prefix sums of the form above.  The order of the keys INSIDE a node is not the reference's (candidate order) but
// the order of their deeper digits; the only place the reference's order shows is "the first of the largest responses" at the end,
// decided here by the key index.  Codes, indices and all node arrays live in LDS (the node arrays take the place of the sort's
// second buffer).  The form above moved ~25 bytes per candidate and round through L2 / HBM (45x the algorithmic bytes on the PMC
// counters, 0.47 ms per 128 KITTI images alone) and a third of a job's time was ONE lane replaying std::sort for the closing phase;
// this one reads a candidate twice, replays the sort with the whole workgroup (intro_sort.hpp) and takes 0.26 ms in three launches
// (LDS classes of 38 / 76 / 156 KB, i.e. 4 / 2 / 1 workgroups per CU: a job runs in the first class it fits; what fits none -- more
// than ~12 k candidates, or more than ~1000 nodes -- is left to k_quadtree.  Same results: tests/test_quadtree_gpu.py runs both).
// One job of the sorted form by the calling workgroup (all of its lanes; every return is taken by all of them).  s_quad: the class's
// dynamic LDS block.  first_class >= 0: the job is checked against the class (one workgroup per job, k_quadtree_sorted); < 0: the caller
// knows it belongs here (k_quadtree_sorted_list).
template <int CLASS>
__device__ __forceinline__ void quad_sorted_job(const QuadJob& J, const uint32_t* __restrict__ dense, const int32_t* __restrict__ level_counts,
                                                uint32_t* __restrict__ picked, int32_t* __restrict__ picked_count, int32_t* __restrict__ status, int first_class,
                                                uint8_t* s_quad) {
    constexpr int T = quad_class_threads(CLASS);
    const int tid = threadIdx.x;
    const uint32_t* cand = dense + J.cand_off;
    const int ncand = level_counts[J.count_idx];
    uint32_t* out = picked + J.out_off;
    int32_t* const out_count = picked_count + J.count_idx;
    __shared__ int s_wave[T / 64];
    __shared__ int s_cut;
    __shared__ int s_stack[3 * 64];
    __shared__ int s_tot[2][T / 64], s_base[2][T / 64 + 1], s_any;
    const int K = J.max_keys, MN = J.max_nodes, N = J.n_target;
    const int W = J.max_x - J.min_x, H = J.max_y - J.min_y;
    const int n_ini = (int)roundf((float)W / (float)H);
    if (ncand <= 0 || ncand > K || n_ini <= 0 || n_ini > MN) {  // the trivial outcomes belong to the first class (first_class < 0: to the list's maker)
        if (CLASS == first_class && tid == 0) {
            *out_count = 0;
            if (ncand > K) atomicMax(status, 1);
            else if (ncand > 0 && n_ini > MN) atomicMax(status, 2);
        }
        return;
    }
    if (first_class >= 0 && quad_class_of(ncand, MN, n_ini, first_class) != CLASS) return;  // another launch's job
    const SortedLayout lay(ncand, MN, T);
    uint32_t* const codes = reinterpret_cast<uint32_t*>(s_quad + lay.codes);
    uint16_t* const idxs = reinterpret_cast<uint16_t*>(s_quad + lay.idx);
    SNode* nodes = reinterpret_cast<SNode*>(s_quad + lay.nodes_a);
    SNode* nodes2 = reinterpret_cast<SNode*>(s_quad + lay.nodes_b);
    int32_t* const cnt = reinterpret_cast<int32_t*>(s_quad + lay.cnt);          // [MN][4] populations of the children of a divided node
    int32_t* const cidx = reinterpret_cast<int32_t*>(s_quad + lay.cidx);        // [MN][4] place of the child in the new list
    int32_t* const procpos = reinterpret_cast<int32_t*>(s_quad + lay.procpos);  // [MN] place of the node in this round's processing order, -1: not divided
    int32_t* const order = reinterpret_cast<int32_t*>(s_quad + lay.order);      // [MN] the processing order
    int32_t* const newpos = reinterpret_cast<int32_t*>(s_quad + lay.newpos);    // [MN] children pushed before this node's / new place of a node that stays
    QRec* recs = reinterpret_cast<QRec*>(s_quad + lay.rec_a);                   // the multi-key children of the last round, in creation order
    QRec* recs2 = reinterpret_cast<QRec*>(s_quad + lay.rec_b);

    // ---- every key's path: column (:533-560), then the quadrants of DivideNode (:454-510) down to one pixel ----
    const float hX = (float)W / (float)n_ini;
    int col_bits = 0;
    while ((1 << col_bits) < n_ini) ++col_bits;
    const int passes = (2 * kPathDepth + col_bits + 3) / 4;
    {
        uint32_t* src_c = (passes & 1) ? reinterpret_cast<uint32_t*>(s_quad + lay.codes_b) : codes;
        uint16_t* src_i = (passes & 1) ? reinterpret_cast<uint16_t*>(s_quad + lay.idx_b) : idxs;
        uint32_t* dst_c = (passes & 1) ? codes : reinterpret_cast<uint32_t*>(s_quad + lay.codes_b);
        uint16_t* dst_i = (passes & 1) ? idxs : reinterpret_cast<uint16_t*>(s_quad + lay.idx_b);
        uint16_t* const ctr = reinterpret_cast<uint16_t*>(s_quad + lay.counters);  // [16][T]: digit-major, lane-minor = the order of a stable pass
        for (int kb = tid; kb < ncand; kb += 4 * T) {  // four candidates per lane and trip: their loads are in flight together
            uint32_t c4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c4[u] = kb + u * T < ncand ? cand[kb + u * T] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u * T;
                if (k >= ncand) continue;
                const uint32_t c = c4[u];
                const int x = q_cx(c), y = q_cy(c);
                size_t b = (size_t)((float)x / hX);
                if (b >= (size_t)n_ini) b = (size_t)n_ini - 1;  // unreachable for in-range candidates
                int ulx = (int16_t)(int)(hX * (float)b), brx = (int16_t)(int)(hX * (float)(b + 1)), uly = 0, bry = (int16_t)H;
                uint32_t code = (uint32_t)b;
#pragma unroll
                for (int d = 0; d < kPathDepth; ++d) {
                    const int mx = ulx + ((brx - ulx + 1) >> 1), my = uly + ((bry - uly + 1) >> 1);  // ceil(w / 2) of DivideNode, w >= 0
                    const bool right = !(x < mx), low = !(y < my);
                    code = (code << 2) | (right ? 1u : 0u) | (low ? 2u : 0u);
                    if (right) ulx = mx; else brx = mx;
                    if (low) uly = my; else bry = my;
                }
                src_c[k] = code;
                src_i[k] = (uint16_t)k;
            }
        }
        __syncthreads();
        const int seg = ((ncand + T - 1) / T) | 1;  // odd: the lanes' runs start in different banks
        const int k0 = min(tid * seg, ncand), k1 = min(k0 + seg, ncand);
        __syncthreads();
        for (int p = 0; p < passes; ++p) {
            const int sh = 4 * p;
            const uint32_t* __restrict__ sc = src_c;
            const uint16_t* __restrict__ si = src_i;
            uint32_t* __restrict__ dc = dst_c;
            uint16_t* __restrict__ di = dst_i;
            // the lane's run of keys in registers for the whole pass; its digit counts likewise: 16 x 8 bits
            uint32_t kc[kQuadMaxRun];
            uint16_t ki[kQuadMaxRun];
#pragma unroll
            for (int j = 0; j < kQuadMaxRun; ++j) {
                const bool in = k0 + j < k1;
                kc[j] = in ? sc[k0 + j] : 0u;
                ki[j] = in ? si[k0 + j] : (uint16_t)0;
            }
            unsigned long long c_lo = 0, c_hi = 0;
#pragma unroll
            for (int j = 0; j < kQuadMaxRun; ++j) {
                const uint32_t d = (kc[j] >> sh) & 15;
                const unsigned long long one = k0 + j < k1 ? 1ull << (8 * (d & 7)) : 0ull;
                c_lo += d < 8 ? one : 0ull;
                c_hi += d < 8 ? 0ull : one;
            }
#pragma unroll
            for (int d = 0; d < 16; ++d) ctr[d * T + tid] = (uint16_t)(((d < 8 ? c_lo : c_hi) >> (8 * (d & 7))) & 255);
            __syncthreads();
            {
                uint16_t* mine = ctr + 16 * tid;
                int v[16], sum = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) { v[j] = mine[j]; sum += v[j]; }
                int tot;
                int run = block_excl_scan<T>(sum, s_wave, tot);
#pragma unroll
                for (int j = 0; j < 16; ++j) { mine[j] = (uint16_t)run; run += v[j]; }
            }
            __syncthreads();
            // the lane's 16 running places, 16 bits each, in four registers
            unsigned long long o0 = 0, o1 = 0, o2 = 0, o3 = 0;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const unsigned long long v = (unsigned long long)ctr[d * T + tid] << (16 * (d & 3));
                if (d < 4) o0 |= v; else if (d < 8) o1 |= v; else if (d < 12) o2 |= v; else o3 |= v;
            }
#pragma unroll
            for (int j = 0; j < kQuadMaxRun; ++j) {
                if (k0 + j >= k1) continue;
                const uint32_t code = kc[j];
                const uint32_t d = (code >> sh) & 15, w = d >> 2, s16 = 16 * (d & 3);
                const unsigned long long reg = w == 0 ? o0 : w == 1 ? o1 : w == 2 ? o2 : o3;
                const int pos = (int)((reg >> s16) & 0xffff);
                const unsigned long long inc = 1ull << s16;
                o0 += w == 0 ? inc : 0ull; o1 += w == 1 ? inc : 0ull; o2 += w == 2 ? inc : 0ull; o3 += w == 3 ? inc : 0ull;
                dc[pos] = code;
                di[pos] = ki[j];
            }
            __syncthreads();
            { uint32_t* t = src_c; src_c = dst_c; dst_c = t; }
            { uint16_t* t = src_i; src_i = dst_i; dst_i = t; }
        }
    }
    // ---- initial nodes: the non-empty columns in order ----
    for (int i = tid; i <= n_ini; i += T) {  // cidx[i]: first key of column i
        const uint32_t want = (uint32_t)i << (2 * kPathDepth);
        int lo = 0, hi = ncand;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (codes[mid] < want) lo = mid + 1; else hi = mid; }
        cidx[i] = i == n_ini ? ncand : lo;
    }
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int i = 0; i < n_ini; ++i) {
            const int c = cidx[i + 1] - cidx[i];
            if (c > 0) nodes[n++] = SNode{(int16_t)(int)(hX * (float)i), 0, (int16_t)(int)(hX * (float)(i + 1)), (int16_t)H, cidx[i], c};
        }
        s_cut = n;
    }
    __syncthreads();
    int size = s_cut, phase = 0, n_rec = 0;
    __syncthreads();

    for (int guard = 0; guard < 100000 && phase != 2; ++guard) {
        // ---- 1. the nodes to divide and their order: every multi-key node in list order, or (closing phase) the sorted records from the back ----
        int n_proc;
        if (phase == 0) {
            int carry = 0;
            for (int i0 = 0; i0 < size; i0 += T) {
                const int i = i0 + tid;
                const int f = i < size && sn_count(nodes[i]) > 1 ? 1 : 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(f, s_wave, tot);
                if (i < size) procpos[i] = f ? ex : -1;
                if (f) order[ex] = i;
                carry += tot;
            }
            n_proc = carry;
        } else {
            n_proc = n_rec;
            for (int i = tid; i < size; i += T) procpos[i] = -1;
            __syncthreads();
            for (int j = tid; j < n_proc; j += T) { const int nd = recs[n_rec - 1 - j].node; order[j] = nd; procpos[nd] = j; }
        }
        __syncthreads();
        // ---- 2. populations of the children: the boundaries of the next digit inside the node's range (four lanes per node) ----
        for (int j0 = 0; j0 < n_proc; j0 += T / 4) {
            const int j = j0 + (tid >> 2), q = tid & 3;
            int bound = 0, nd = 0;
            SNode p{};
            if (j < n_proc) {
                nd = order[j];
                p = nodes[nd];
                const int depth = sn_depth(p), count = sn_count(p);
                if (q == 0) bound = p.begin;
                else if (depth >= kPathDepth) bound = p.begin + count;  // one pixel: every key goes to the first child
                else {
                    const int sh = 2 * (kPathDepth - 1 - depth);
                    const uint32_t want = ((codes[p.begin] >> (sh + 2)) << 2) | (uint32_t)q;
                    int lo = p.begin, hi = p.begin + count;
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((codes[mid] >> sh) < want) lo = mid + 1; else hi = mid; }
                    bound = lo;
                }
            }
            const int next = __shfl_down(bound, 1, 4);
            if (j < n_proc) cnt[4 * nd + q] = (q == 3 ? p.begin + sn_count(p) : next) - bound;
        }
        if (tid == 0) s_cut = n_proc;
        __syncthreads();
        // ---- 3. closing phase: the walk stops as soon as the list holds N nodes (:700-701); a division adds children - 1 ----
        if (phase == 1) {
            int carry = 0;
            for (int j0 = 0; j0 < n_proc; j0 += T) {
                const int j = j0 + tid;
                int grow = 0;
                if (j < n_proc) { for (int q = 0; q < 4; ++q) grow += cnt[4 * order[j] + q] > 0; grow -= 1; }
                int tot;
                const int ex = carry + block_excl_scan<T>(grow, s_wave, tot);
                if (j < n_proc && size + ex + grow >= N) atomicMin(&s_cut, j + 1);
                carry += tot;
            }
            __syncthreads();
            const int m = s_cut;
            for (int j = m + tid; j < n_proc; j += T) procpos[order[j]] = -1;  // not reached
            n_proc = m;
            __syncthreads();
        }
        // ---- 4. places in the new list ----
        int n_children;
        {
            int carry = 0;
            for (int j0 = 0; j0 < n_proc; j0 += T) {
                const int j = j0 + tid;
                int ch = 0;
                if (j < n_proc) for (int q = 0; q < 4; ++q) ch += cnt[4 * order[j] + q] > 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(ch, s_wave, tot);
                if (j < n_proc) newpos[order[j]] = ex;
                carry += tot;
            }
            n_children = carry;
        }
        int new_size;
        {
            int carry = 0;
            for (int i0 = 0; i0 < size; i0 += T) {
                const int i = i0 + tid;
                const int f = i < size && procpos[i] < 0 ? 1 : 0;
                int tot;
                const int ex = carry + block_excl_scan<T>(f, s_wave, tot);
                if (f) newpos[i] = n_children + ex;
                carry += tot;
            }
            new_size = n_children + carry;
        }
        if (new_size > MN) { if (tid == 0) { atomicMax(status, 3); *out_count = 0; } return; }
        for (int i = tid; i < size; i += T) {
            const SNode p = nodes[i];
            if (procpos[i] < 0) { nodes2[newpos[i]] = p; continue; }
            const int mx = p.ulx + (int)ceilf((float)(p.brx - p.ulx) / 2), my = p.uly + (int)ceilf((float)(p.bry - p.uly) / 2);
            const int depth1 = min(sn_depth(p) + 1, 64);
            int pushed = newpos[i], first = p.begin;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cnt[4 * i + q];
                if (c == 0) { cidx[4 * i + q] = -1; continue; }
                const int at = n_children - 1 - pushed;  // push_front: a later push lies nearer to the front
                nodes2[at] = SNode{(int16_t)((q & 1) ? mx : p.ulx), (int16_t)((q & 2) ? my : p.uly), (int16_t)((q & 1) ? p.brx : mx),
                                   (int16_t)((q & 2) ? p.bry : my), first, c | depth1 << 24};
                cidx[4 * i + q] = at;
                first += c;
                ++pushed;
            }
        }
        __syncthreads();
        // ---- 5. the multi-key children of this round, in creation order ----
        int n_expand;
        {
            int carry = 0;
            for (int j0 = 0; j0 < n_proc; j0 += T) {
                const int j = j0 + tid;
                int m = 0;
                if (j < n_proc) for (int q = 0; q < 4; ++q) m += cnt[4 * order[j] + q] > 1;
                int tot;
                const int ex = carry + block_excl_scan<T>(m, s_wave, tot);
                if (j < n_proc) {
                    int at = ex;
                    const int nd = order[j];
                    for (int q = 0; q < 4; ++q) {
                        const int c = cnt[4 * nd + q];
                        if (c > 1) { const int nn = cidx[4 * nd + q]; recs2[at++] = QRec{((uint32_t)c << 12) | (uint32_t)nodes2[nn].ulx, nn}; }
                    }
                }
                carry += tot;
            }
            n_expand = carry;
        }
        __syncthreads();
        // ---- 6. what comes next (:651-728) ----
        int next_phase;
        if (new_size >= N || new_size == size) next_phase = 2;
        else if (phase == 1 || new_size + n_expand * 3 > N) next_phase = 1;
        else next_phase = 0;
        bool sorted_into_recs = false;
        if (next_phase == 1) {  // sort(vPrevSizeAndPointerToNode.begin(), vPrevSizeAndPointerToNode.end(), compareNodes)
            // std::sort's moves replayed by the whole workgroup (intro_sort.hpp); one lane doing them one by one took 120 us of a job's 190
            uint32_t* const skey = reinterpret_cast<uint32_t*>(s_quad + lay.s_key);
            int* const sidx = reinterpret_cast<int*>(s_quad + lay.s_idx);
            unsigned short* const u16 = reinterpret_cast<unsigned short*>(s_quad + lay.s_u16);
            unsigned short *const sf = u16, *const sl = u16 + (MN + 2), *const cl = u16 + 2 * (MN + 2), *const cr = u16 + 3 * (MN + 2),
                                 *const lp = u16 + 4 * (MN + 2), *const rp = u16 + 5 * (MN + 2), *const cut = u16 + 6 * (MN + 2);
            uint8_t* const flag = s_quad + lay.s_flag;
            for (int j = tid; j < n_expand; j += T) { skey[j] = recs2[j].key; sidx[j] = j; sf[j] = 0; sl[j] = (unsigned short)n_expand; }
            const int depth = n_expand > 1 ? 2 * (31 - __clz(n_expand)) : 0;
            const bool ok = sort_levels<T, unsigned short, uint32_t>(skey, sidx, sf, sl, cl, cr, lp, rp, cut, flag, nullptr, n_expand, depth, 16, s_tot, s_base, &s_any);
            __syncthreads();
            if (ok) {
                sort_final<T, unsigned short, uint32_t>(skey, sidx, sf, sl, n_expand, newpos);  // newpos is free here: the sorted order as record numbers
                __syncthreads();
                for (int j = tid; j < n_expand; j += T) recs[j] = recs2[newpos[j]];
                sorted_into_recs = true;
            } else if (tid == 0) {
                std_sort(ArrMem{recs2}, n_expand, s_stack);  // the depth limit was reached (std::sort heap-sorts that range): one lane, move by move
            }
        }
        { SNode* t = nodes; nodes = nodes2; nodes2 = t; }
        if (!sorted_into_recs) { QRec* t = recs; recs = recs2; recs2 = t; }
        size = new_size; phase = next_phase; n_rec = n_expand;
        __syncthreads();
    }
    // ---- the best candidate of every node, in list order (:731-752): the node's keys stand in candidate order ----
    if (size > J.out_cap) { if (tid == 0) { atomicMax(status, 4); *out_count = 0; } return; }
    // the sorted order inside a node is the order of the deeper digits, so "the first of the largest responses" is decided by the key index
    for (int kb = tid; kb < ncand; kb += 4 * T) {  // the paths have done their work: the candidates take their place
        uint32_t c4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) c4[u] = kb + u * T < ncand ? cand[idxs[kb + u * T]] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) if (kb + u * T < ncand) codes[kb + u * T] = c4[u];
    }
    __syncthreads();
    for (int i = tid; i < size; i += T) {
        const SNode n = nodes[i];
        const int count = sn_count(n);
        uint32_t best = codes[n.begin];
        int best_k = idxs[n.begin];
        for (int k = 1; k < count; ++k) {
            const uint32_t c = codes[n.begin + k];
            const int key = idxs[n.begin + k];
            if (q_cr(c) > q_cr(best) || (q_cr(c) == q_cr(best) && key < best_k)) { best = c; best_k = key; }
        }
        out[i] = best;
    }
    if (tid == 0) *out_count = size;
}


template <int CLASS>
__global__ __launch_bounds__(quad_class_threads(CLASS)) void k_quadtree_sorted(const QuadJob* __restrict__ jobs, const uint32_t* __restrict__ dense,
                                                                               const int32_t* __restrict__ level_counts, uint32_t* __restrict__ picked,
                                                                               int32_t* __restrict__ picked_count, int32_t* __restrict__ status, int nlevels, int first_class) {
    extern __shared__ __align__(16) uint8_t s_quad[];
    const QuadJob J = jobs[quad_job_of_block(blockIdx.x, gridDim.x, nlevels)];
    quad_sorted_job<CLASS>(J, dense, level_counts, picked, picked_count, status, first_class, s_quad);
}

// ---- batches: the jobs of a class from a list, by as many workgroups as the class's LDS block lets the GPU hold ----
// One workgroup per job and launch (above) asks the dispatcher for the class's LDS block -- 76 or 156 KB, half or all of a CU's -- for EVERY
// job of the batch, although a job of another class only looks at its candidate count and leaves: beside the other stages' kernels each of
// those 4096 workgroups waits for a CU with that much LDS free (k_quadtree_sorted<2>: 0.35 ms alone, 2 ms in the loop).  Here a first
// kernel sorts the jobs into per-class lists (and settles the trivial outcomes), and every class runs as a resident set of workgroups that
// take the list's jobs one after the other.
// lists: [kQuadClasses + 1][n_jobs] job numbers; counters: [0 .. kQuadClasses] list lengths, [kQuadClasses + 1 ..] the classes' next entries
__global__ __launch_bounds__(256) void k_quadtree_classify(const QuadJob* __restrict__ jobs, int n_jobs, const int32_t* __restrict__ level_counts,
                                                           int32_t* __restrict__ picked_count, int32_t* __restrict__ status, int32_t* __restrict__ lists,
                                                           int32_t* __restrict__ counters) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_jobs) return;
    const QuadJob J = jobs[j];
    const int ncand = level_counts[J.count_idx];
    const int W = J.max_x - J.min_x, H = J.max_y - J.min_y;
    const int n_ini = (int)roundf((float)W / (float)H);
    if (ncand <= 0 || ncand > J.max_keys || n_ini <= 0 || n_ini > J.max_nodes) {
        picked_count[J.count_idx] = 0;
        if (ncand > J.max_keys) atomicMax(status, 1);
        else if (ncand > 0 && n_ini > J.max_nodes) atomicMax(status, 2);
        return;
    }
    const int c = quad_class_of(ncand, J.max_nodes, n_ini, 0);
    lists[(size_t)c * n_jobs + atomicAdd(&counters[c], 1)] = j;
}
template <int CLASS>
__global__ __launch_bounds__(quad_class_threads(CLASS)) void k_quadtree_sorted_list(const QuadJob* __restrict__ jobs, int n_jobs, const int32_t* __restrict__ lists,
                                                                                    int32_t* __restrict__ counters, const uint32_t* __restrict__ dense,
                                                                                    const int32_t* __restrict__ level_counts, uint32_t* __restrict__ picked,
                                                                                    int32_t* __restrict__ picked_count, int32_t* __restrict__ status) {
    extern __shared__ __align__(16) uint8_t s_quad[];
    __shared__ int s_job;
    const int count = counters[CLASS];
    for (;;) {
        __syncthreads();  // the job before has left the LDS block and s_job
        if (threadIdx.x == 0) s_job = atomicAdd(&counters[kQuadClasses + 1 + CLASS], 1);
        __syncthreads();
        const int k = s_job;
        if (k >= count) break;  // (the same for every lane)
        const QuadJob J = jobs[lists[(size_t)CLASS * n_jobs + k]];
        quad_sorted_job<CLASS>(J, dense, level_counts, picked, picked_count, status, -1, s_quad);
    }
}

// The keypoints of an image: its per-level lists behind each other (level order, :1093-1137), in level pixel coordinates with the
// border added (:856-857).  One workgroup per image; the lists go to a device array (the descriptor kernel reads it) and to its pinned
// host mirror (the assembly on the host reads it), the counts likewise.
__global__ __launch_bounds__(256) void k_quadtree_gather(const QuadJob* __restrict__ jobs, const uint32_t* __restrict__ picked, const int32_t* __restrict__ picked_count,
                                                        const int32_t* __restrict__ level_counts, int first_image, int nlevels, int kp_stride, int min_border,
                                                        DevKeypoint* __restrict__ kps, DevKeypoint* __restrict__ kps_host, int32_t* __restrict__ n_kp,
                                                        int32_t* __restrict__ n_kp_host, int32_t* __restrict__ level_counts_host, int32_t* __restrict__ status) {
    const int img = first_image + (int)blockIdx.x, tid = threadIdx.x;
    int off = 0;
    for (int l = 0; l < nlevels; ++l) {
        const int j = img * nlevels + l, n = picked_count[j];
        const uint32_t* in = picked + jobs[j].out_off;
        for (int k = tid; k < n; k += 256) {
            if (off + k >= kp_stride) break;
            const uint32_t cc = in[k];
            const uint32_t x = ((cc >> 8) & 0xfff) + (uint32_t)min_border, y = (cc >> 20) + (uint32_t)min_border;
            const DevKeypoint kp{(y << 20) | (x << 8) | (cc & 0xff), ((uint32_t)img << 8) | (uint32_t)l};
            kps[(size_t)img * kp_stride + off + k] = kp;
            kps_host[(size_t)img * kp_stride + off + k] = kp;
        }
        if (tid == 0) level_counts_host[j] = level_counts[j];
        off += n;
    }
    if (tid == 0) {
        if (off > kp_stride) { atomicMax(status, 5); off = kp_stride; }
        n_kp[img] = off;
        n_kp_host[img] = off;
    }
}

// threads == 0: the sorted form in three LDS classes (4 / 2 / 1 workgroups per CU; a job takes the first class it fits), then the
// global-memory form for what is left; threads = -1 - c: the same starting with class c; threads > 0: the global-memory form alone
// with that many lanes per job.
void launch_quadtree(const QuadJob* jobs, int first_job, int n_jobs, const uint32_t* dense, const int32_t* level_counts, uint8_t* scratch, uint32_t* picked,
                     int32_t* picked_count, int32_t* status, int threads, int nlevels, hipStream_t st, int32_t* class_work) {
    if (n_jobs <= 0) return;
    const QuadJob* j0 = jobs + first_job;
    int after_sorted = 0;
    // the two wider classes need more dynamic LDS than a kernel gets by default; should the runtime refuse, every job takes the global-memory form
    if (threads <= 0 && !(ensure_dynamic_lds(reinterpret_cast<const void*>(k_quadtree_sorted<1>), (int)quad_class_lds(1)) &&
                          ensure_dynamic_lds(reinterpret_cast<const void*>(k_quadtree_sorted<2>), (int)quad_class_lds(2)))) threads = 256;
    const char* lists_env = getenv("TC2LI_QUADTREE_LISTS");  // (read per call: the tests switch it) 0: one workgroup per job and class launch
    const bool kNoLists = lists_env && atoi(lists_env) == 0;
    if (threads == 0 && class_work && n_jobs > 256 && !kNoLists) {
        // a batch: per-class job lists, every class a resident set of workgroups (k_quadtree_sorted_list)
        int32_t* const lists = class_work + 16;
        int32_t* const counters = class_work;
        (void)hipMemsetAsync(counters, 0, 16 * sizeof(int32_t), st);
        TC2LI_LAUNCH(k_quadtree_classify, dim3((n_jobs + 255) / 256), dim3(256), 0, st, j0, n_jobs, level_counts, picked_count, status, lists, counters);
        const int cus = 256;
        TC2LI_LAUNCH(k_quadtree_sorted_list<0>, dim3(std::min(n_jobs, 4 * cus)), dim3(quad_class_threads(0)), quad_class_lds(0), st, j0, n_jobs, lists, counters, dense,
                     level_counts, picked, picked_count, status);
        TC2LI_LAUNCH(k_quadtree_sorted_list<1>, dim3(std::min(n_jobs, 2 * cus)), dim3(quad_class_threads(1)), quad_class_lds(1), st, j0, n_jobs, lists, counters, dense,
                     level_counts, picked, picked_count, status);
        TC2LI_LAUNCH(k_quadtree_sorted_list<2>, dim3(std::min(n_jobs, cus)), dim3(quad_class_threads(2)), quad_class_lds(2), st, j0, n_jobs, lists, counters, dense,
                     level_counts, picked, picked_count, status);
        after_sorted = 1;
        threads = 256;
    } else if (threads <= 0) {
        // fewer jobs than CUs (a stereo pair is 16): every job in the widest class, one launch instead of three in a row
        const int first_class = threads < 0 ? min(-threads - 1, kQuadClasses - 1) : (n_jobs <= 256 ? 2 : 0);  // threads < 0: the tests choose
        if (first_class <= 0)
            TC2LI_LAUNCH(k_quadtree_sorted<0>, dim3(n_jobs), dim3(quad_class_threads(0)), quad_class_lds(0), st, j0, dense, level_counts, picked, picked_count, status, nlevels, first_class);
        if (first_class <= 1)
            TC2LI_LAUNCH(k_quadtree_sorted<1>, dim3(n_jobs), dim3(quad_class_threads(1)), quad_class_lds(1), st, j0, dense, level_counts, picked, picked_count, status, nlevels, first_class);
        TC2LI_LAUNCH(k_quadtree_sorted<2>, dim3(n_jobs), dim3(quad_class_threads(2)), quad_class_lds(2), st, j0, dense, level_counts, picked, picked_count, status, nlevels, first_class);
        after_sorted = 1 + first_class;
        threads = 256;
    }
    if (threads >= 1024) TC2LI_LAUNCH(k_quadtree<1024>, dim3(n_jobs), dim3(1024), 0, st, j0, dense, level_counts, scratch, picked, picked_count, status, after_sorted, nlevels);
    else if (threads >= 512) TC2LI_LAUNCH(k_quadtree<512>, dim3(n_jobs), dim3(512), 0, st, j0, dense, level_counts, scratch, picked, picked_count, status, after_sorted, nlevels);
    else if (threads >= 256) TC2LI_LAUNCH(k_quadtree<256>, dim3(n_jobs), dim3(256), 0, st, j0, dense, level_counts, scratch, picked, picked_count, status, after_sorted, nlevels);
    else TC2LI_LAUNCH(k_quadtree<128>, dim3(n_jobs), dim3(128), 0, st, j0, dense, level_counts, scratch, picked, picked_count, status, after_sorted, nlevels);
}
void launch_quadtree_gather(const QuadJob* jobs, const uint32_t* picked, const int32_t* picked_count, const int32_t* level_counts, int first_image, int n_images,
                            int nlevels, int kp_stride, DevKeypoint* kps, DevKeypoint* kps_host, int32_t* n_kp, int32_t* n_kp_host, int32_t* level_counts_host,
                            int32_t* status, hipStream_t st) {
    if (n_images > 0)
        TC2LI_LAUNCH(k_quadtree_gather, dim3(n_images), dim3(256), 0, st, jobs, picked, picked_count, level_counts, first_image, nlevels, kp_stride, kMinBorder, kps,
                     kps_host, n_kp, n_kp_host, level_counts_host, status);
}

size_t quadtree_scratch_bytes(int max_keys, int max_nodes) { return ScratchLayout(max_keys, max_nodes).total; }
size_t quadtree_class_work_ints(int n_jobs) { return 16 + (size_t)(kQuadClasses + 1) * n_jobs; }

}  // namespace tc2li
