#include "quadtree.hpp"

#include <algorithm>
#include <cmath>

namespace tc2li {

namespace {

struct Tree {
    QuadtreeScratch& s;
    const uint32_t* cand;
    int head = -1, tail = -1, size = 0;

    static int cx(uint32_t c) { return (int)((c >> 8) & 0xfff); }
    static int cy(uint32_t c) { return (int)(c >> 20); }
    static int cr(uint32_t c) { return (int)(c & 0xff); }

    int new_node(int ulx, int uly, int brx, int bry) {
        s.nodes.push_back({ulx, uly, brx, bry, 0, 0, -1, -1, false});
        return (int)s.nodes.size() - 1;
    }
    void push_back(int id) {
        auto& n = s.nodes[id];
        n.prev = tail; n.next = -1;
        if (tail >= 0) s.nodes[tail].next = id; else head = id;
        tail = id; ++size;
    }
    void push_front(int id) {
        auto& n = s.nodes[id];
        n.next = head; n.prev = -1;
        if (head >= 0) s.nodes[head].prev = id; else tail = id;
        head = id; ++size;
    }
    int erase(int id) {  // returns the following node
        auto& n = s.nodes[id];
        const int nx = n.next;
        if (n.prev >= 0) s.nodes[n.prev].next = n.next; else head = n.next;
        if (n.next >= 0) s.nodes[n.next].prev = n.prev; else tail = n.prev;
        --size;
        return nx;
    }

    // Splits node `id` into its (up to four) non-empty children, pushed to the list front in the order
    // upper-left, upper-right, lower-left, lower-right; children holding more than one key are recorded in
    // s.expand.  Returns how many such children there are.
    int divide(int id) {
        const QuadtreeScratch::Node p = s.nodes[id];
        const int halfX = (int)std::ceil(static_cast<float>(p.brx - p.ulx) / 2);
        const int halfY = (int)std::ceil(static_cast<float>(p.bry - p.uly) / 2);
        const int mx = p.ulx + halfX, my = p.uly + halfY;
        int cnt[4] = {0, 0, 0, 0};
        for (int k = 0; k < p.count; ++k) {
            const uint32_t c = cand[s.keys[p.begin + k]];
            const int q = (cx(c) < mx ? 0 : 1) + (cy(c) < my ? 0 : 2);
            ++cnt[q];
        }
        const int base = (int)s.keys.size();
        s.keys.resize(base + p.count);
        int off[4] = {base, base + cnt[0], base + cnt[0] + cnt[1], base + cnt[0] + cnt[1] + cnt[2]};
        int fill[4] = {off[0], off[1], off[2], off[3]};
        for (int k = 0; k < p.count; ++k) {
            const int key = s.keys[p.begin + k];
            const uint32_t c = cand[key];
            const int q = (cx(c) < mx ? 0 : 1) + (cy(c) < my ? 0 : 2);
            s.keys[fill[q]++] = key;
        }
        const int box[4][4] = {{p.ulx, p.uly, mx, my}, {mx, p.uly, p.brx, my}, {p.ulx, my, mx, p.bry}, {mx, my, p.brx, p.bry}};
        int n_multi = 0;
        for (int q = 0; q < 4; ++q) {
            if (cnt[q] == 0) continue;
            const int child = new_node(box[q][0], box[q][1], box[q][2], box[q][3]);
            s.nodes[child].begin = off[q];
            s.nodes[child].count = cnt[q];
            s.nodes[child].no_more = cnt[q] == 1;
            push_front(child);
            if (cnt[q] > 1) { ++n_multi; s.expand.emplace_back(cnt[q], child); }
        }
        return n_multi;
    }
};

}  // namespace

void distribute_quadtree(const uint32_t* cand, int ncand, int min_x, int max_x, int min_y, int max_y, int n_target,
                         QuadtreeScratch& s, std::vector<int32_t>& out) {
    s.nodes.clear();
    s.keys.clear();
    s.expand.clear();
    if (ncand <= 0) return;
    Tree t{s, cand};

    const int n_ini = (int)std::round(static_cast<float>(max_x - min_x) / (max_y - min_y));
    if (n_ini <= 0) return;  // the reference would divide by zero here (ORBextractor.cc:533-535)
    const float hX = static_cast<float>(max_x - min_x) / n_ini;
    std::vector<int> ini_cnt(n_ini, 0);
    std::vector<int> slot(ncand);
    for (int i = 0; i < ncand; ++i) {
        size_t b = (size_t)((float)Tree::cx(cand[i]) / hX);
        if (b >= (size_t)n_ini) b = (size_t)n_ini - 1;  // unreachable for in-range candidates
        slot[i] = (int)b;
        ++ini_cnt[b];
    }
    s.keys.resize(ncand);
    std::vector<int> fill(n_ini, 0);
    for (int i = 0, acc = 0; i < n_ini; ++i) {
        const int id = t.new_node((int)(hX * static_cast<float>(i)), 0, (int)(hX * static_cast<float>(i + 1)), max_y - min_y);
        s.nodes[id].begin = acc;
        s.nodes[id].count = ini_cnt[i];
        fill[i] = acc;
        acc += ini_cnt[i];
        t.push_back(id);
    }
    for (int i = 0; i < ncand; ++i) s.keys[fill[slot[i]]++] = i;
    for (int id = t.head; id >= 0;) {
        auto& n = s.nodes[id];
        if (n.count == 1) { n.no_more = true; id = n.next; }
        else if (n.count == 0) id = t.erase(id);
        else id = n.next;
    }

    auto by_size_then_x = [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {
        if (a.first != b.first) return a.first < b.first;
        return s.nodes[a.second].ulx < s.nodes[b.second].ulx;
    };

    bool finish = false;
    while (!finish) {
        int prev_size = t.size;
        int n_to_expand = 0;
        s.expand.clear();
        for (int id = t.head; id >= 0;) {
            if (s.nodes[id].no_more) { id = s.nodes[id].next; continue; }
            n_to_expand += t.divide(id);
            id = t.erase(id);
        }
        if (t.size >= n_target || t.size == prev_size) {
            finish = true;
        } else if (t.size + n_to_expand * 3 > n_target) {
            while (!finish) {
                prev_size = t.size;
                s.prev_expand = s.expand;
                s.expand.clear();
                std::sort(s.prev_expand.begin(), s.prev_expand.end(), by_size_then_x);
                for (int j = (int)s.prev_expand.size() - 1; j >= 0; --j) {
                    const int id = s.prev_expand[j].second;
                    t.divide(id);
                    t.erase(id);
                    if (t.size >= n_target) break;
                }
                if (t.size >= n_target || t.size == prev_size) finish = true;
            }
        }
    }

    for (int id = t.head; id >= 0; id = s.nodes[id].next) {
        const auto& n = s.nodes[id];
        int best = s.keys[n.begin], best_r = Tree::cr(cand[best]);
        for (int k = 1; k < n.count; ++k) {
            const int key = s.keys[n.begin + k];
            const int r = Tree::cr(cand[key]);
            if (r > best_r) { best = key; best_r = r; }
        }
        out.push_back(best);
    }
}

}  // namespace tc2li
