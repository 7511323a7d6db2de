/*
 * scene_compile.h -- host side of mort_hip_upload_world(): turns a mort_world
 * (the reference's tagged object graph) into the flat DScene arrays of
 * dev_scene.h.  Replaces world::toDevice() (world.cuh:98-102), which copies
 * the object arrays verbatim into __device__/__constant__ symbols.
 */
#ifndef MORT_SCENE_COMPILE_H
#define MORT_SCENE_COMPILE_H

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_scene.h"
#include "mort_hip.h"

namespace mortc {

static const int kMaxChain = 8;
static const int kMaxDepth = 16;

struct Ref { int kind; int idx; int cf, cc; }; /* kind: 1 sphere, 2 quad, 3 medium (world indices) */

struct Compiled {
    std::vector<DItem> items, subitems;
    std::vector<DBvhNode> nodes;
    std::vector<DSphere> spheres, wspheres;
    std::vector<DQuad> quads, wquads;
    std::vector<DXform> xforms;
    std::vector<double> media;
    std::vector<DLambert> lambert, dlight, isotropic;
    std::vector<DMetal> metal;
    std::vector<DDielectric> dielectric;
    std::vector<DSolid> solid;
    std::vector<DChecker> checker;
    std::vector<DImage> image;
    std::vector<unsigned char> texels;
    std::vector<unsigned char> noise;
    std::vector<int> list_types, list_idxs;
    int list_first[MORT_NUM_HITTABLE_LIST], list_count[MORT_NUM_HITTABLE_LIST];
    int status = MORT_OK;
    bool inverted_box = false; /* some BVH node has min > max on an axis */
    /* this build's own tree over the leaf nodes of items[0] (BVH megakernel); empty when not applicable */
    std::vector<DNode2> own_nodes;
    std::vector<DBvhNode> own_leaves;
    int own_depth = 0;
    std::vector<DLeaf2> own_leafrecs; /* own_leaves[i]'s spheres by value */
    std::vector<DNode4> own_nodes4; /* own_nodes, four children wide (collapse_own_tree); empty when its pending-children bound does not hold */
    int own4_stack = 0;             /* most children a walk of own_nodes4 can have pending */
    /* this build's UNIFIED tree over every solid primitive of a world without reference BVHs (mega_gen.hip):
     * see build_unified() */
    bool g_ok = false;
    std::vector<DNodeQ> g_nodes;     /* 32-byte nodes: child boxes as 8-bit offsets from the node's corner (dev_scene.h) */
    std::vector<uint32_t> g_entries; /* GENT(kind, chain id, index in spheres / quads) */
    std::vector<int> g_chains;       /* chain id -> (first, count) in xforms; id 0 = no transform */
    std::vector<uint32_t> g_ranks;   /* scan-order rank of spheres[i] at [i], of quads[i] at [n spheres + i] (equal-t ties) */
    uint32_t g_root = 0xffffu;       /* child reference of the root (0xffff: no solid primitive at all) */
    int g_first_medium = 0;          /* items[g_first_medium ..) are the constant media, tested after the tree */
    int g_depth = 0;
    float g_lo[3] = {0, 0, 0}, g_hi[3] = {0, 0, 0}; /* bounding box of the solids */
    float g_reach = 0;               /* ray origins must stay within this distance of that box (pads are sized for it) */
    /* spheres seen from far away: a ray whose origin is further than g_mnear - g_R from g_c widens its own error band
     * by g_kmin * (|o - g_c| + g_R)^2 world units (mega_gen.hip ray setup) */
    float g_c[3] = {0, 0, 0}, g_R = 0, g_mnear = 0, g_kmin = 0;
};

static inline DSphere to_dsphere(const mort_sphere &s) {
    DSphere d;
    d.cx = s.center1.e[0]; d.cy = s.center1.e[1]; d.cz = s.center1.e[2]; d.radius = s.radius;
    d.vx = s.moves ? s.center_vec.e[0] : 0.0f;
    d.vy = s.moves ? s.center_vec.e[1] : 0.0f;
    d.vz = s.moves ? s.center_vec.e[2] : 0.0f;
    d.mat = DREF(s.mat_type, s.mat_idx) | (s.moves ? 0x80000000u : 0u);
    return d;
}
static inline DQuad to_dquad(const mort_quad &q) {
    DQuad d;
    std::memset(&d, 0, sizeof d);
    for (int k = 0; k < 3; k++) {
        d.Q[k] = q.Q.e[k]; d.u[k] = q.u.e[k]; d.v[k] = q.v.e[k]; d.n[k] = q.normal.e[k]; d.w[k] = q.w.e[k];
    }
    d.D = q.D; d.area = q.area; d.mat = DREF(q.mat_type, q.mat_idx);
    return d;
}

struct Compiler {
    const mort_world *w;
    Compiled out;

    void fail(int st) { if (out.status == MORT_OK) out.status = st; }

    int push_chain(int pf, int pc, const DXform &x) {
        if (pc + 1 > kMaxChain) { fail(MORT_ERR_CAPACITY); return 0; }
        int first = (int)out.xforms.size();
        for (int k = 0; k < pc; k++) { DXform c = out.xforms[pf + k]; out.xforms.push_back(c); }
        out.xforms.push_back(x);
        return first;
    }

    /* hitDispatch (objects.cuh:858-887) unrolled into an ordered list of leaves */
    void flatten(int type, int idx, int cf, int cc, int depth, std::vector<Ref> &refs) {
        if (depth > kMaxDepth) { fail(MORT_ERR_CAPACITY); return; }
        const mort_world_objects &o = w->objs;
        switch (type) {
        case MORT_OBJ_SPHERE:
            if (idx < 0 || idx >= o.num_spheres) { fail(MORT_ERR_INVALID); return; }
            refs.push_back({1, idx, cf, cc});
            break;
        case MORT_OBJ_QUAD:
            if (idx < 0 || idx >= o.num_quads) { fail(MORT_ERR_INVALID); return; }
            refs.push_back({2, idx, cf, cc});
            break;
        case MORT_OBJ_TRANSLATE: {
            if (idx < 0 || idx >= o.num_translates) { fail(MORT_ERR_INVALID); return; }
            const mort_translate &t = o.host_translate[idx];
            DXform x; x.kind = XF_TRANSLATE; x.a = t.offset.e[0]; x.b = t.offset.e[1]; x.c = t.offset.e[2];
            int nf = push_chain(cf, cc, x);
            flatten(t.obj_type, t.obj_idx, nf, cc + 1, depth + 1, refs);
            break;
        }
        case MORT_OBJ_ROTATE_Y: {
            if (idx < 0 || idx >= o.num_rotate_y) { fail(MORT_ERR_INVALID); return; }
            const mort_rotate_y &r = o.host_rotate_y[idx];
            DXform x; x.kind = XF_ROTATE_Y; x.a = r.sin_theta; x.b = r.cos_theta; x.c = 0;
            int nf = push_chain(cf, cc, x);
            flatten(r.obj_type, r.obj_idx, nf, cc + 1, depth + 1, refs);
            break;
        }
        case MORT_OBJ_CONSTANT_MEDIUM:
            if (idx < 0 || idx >= o.num_constant_medium) { fail(MORT_ERR_INVALID); return; }
            refs.push_back({3, idx, cf, cc});
            break;
        case MORT_OBJ_HITTABLE_LIST: {
            if (idx < 0 || idx >= o.num_hittable_list) { fail(MORT_ERR_INVALID); return; }
            const mort_hittable_list &l = o.host_hittable_list[idx];
            for (int i = 0; i < l.num_objs; i++) flatten(l.obj_types[i], l.obj_idxs[i], cf, cc, depth + 1, refs);
            break;
        }
        default: /* OBJ_BVH and unknown tags: hitDispatch returns false */
            break;
        }
    }

    void emit_items(const std::vector<Ref> &refs, std::vector<DItem> &items, bool allow_medium) {
        size_t i = 0;
        while (i < refs.size()) {
            const Ref &r = refs[i];
            DItem it;
            std::memset(&it, 0, sizeof it);
            it.chain_first = r.cf; it.chain_count = r.cc;
            if (r.kind == 3) {
                if (!allow_medium) { fail(MORT_ERR_UNSUPPORTED); return; } /* a medium bounded by a medium */
                const mort_constant_medium &m = w->objs.host_constant_medium[r.idx];
                std::vector<Ref> b;
                flatten(m.obj_type, m.obj_idx, r.cf, r.cc, 1, b);
                it.kind = ITEM_MEDIUM;
                it.first = (int)out.subitems.size();
                emit_items(b, out.subitems, false);
                it.count = (int)out.subitems.size() - it.first;
                it.mat = DREF(m.mat_type, m.mat_idx);
                it.medium = (int)out.media.size();
                out.media.push_back(m.neg_inv_density);
                items.push_back(it);
                i++;
                continue;
            }
            size_t j = i;
            while (j < refs.size() && refs[j].kind == r.kind && refs[j].cf == r.cf && refs[j].cc == r.cc) j++;
            if (r.kind == 1) {
                it.kind = ITEM_SPHERES; it.first = (int)out.spheres.size();
                for (size_t k = i; k < j; k++) out.spheres.push_back(to_dsphere(w->objs.host_sphere[refs[k].idx]));
            } else {
                it.kind = ITEM_QUADS; it.first = (int)out.quads.size();
                for (size_t k = i; k < j; k++) out.quads.push_back(to_dquad(w->objs.host_quad[refs[k].idx]));
            }
            it.count = (int)(j - i);
            items.push_back(it);
            i = j;
        }
    }

    /* one BVH leaf child -> prim code (kind << 15 | index in the compact arrays) */
    uint32_t leaf_prim(int type, int idx) {
        if (type == MORT_OBJ_SPHERE && idx >= 0 && idx < w->objs.num_spheres) {
            out.spheres.push_back(to_dsphere(w->objs.host_sphere[idx]));
            size_t at = out.spheres.size() - 1;
            if (at >= 0x8000) { fail(MORT_ERR_CAPACITY); return 0; }
            return (uint32_t)at;
        }
        if (type == MORT_OBJ_QUAD && idx >= 0 && idx < w->objs.num_quads) {
            out.quads.push_back(to_dquad(w->objs.host_quad[idx]));
            size_t at = out.quads.size() - 1;
            if (at >= 0x8000) { fail(MORT_ERR_CAPACITY); return 0; }
            return 0x8000u | (uint32_t)at;
        }
        fail(MORT_ERR_UNSUPPORTED); /* instances / media / lists as BVH leaves */
        return 0;
    }

    /* pre-order emission of reference node `n` (objects.cuh:725-735) */
    void emit_bvh_node(const mort_bvh &b, int n, int depth) {
        if (depth > 64 || n < 0 || n >= MORT_MAX_BVH_NODES) { fail(MORT_ERR_INVALID); return; }
        size_t id = out.nodes.size();
        DBvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        const mort_aabb &bb = b.bounding_boxes[n];
        nd.xmin = bb.x.imin; nd.xmax = bb.x.imax; nd.ymin = bb.y.imin; nd.ymax = bb.y.imax; nd.zmin = bb.z.imin; nd.zmax = bb.z.imax;
        if (nd.xmin > nd.xmax || nd.ymin > nd.ymax || nd.zmin > nd.zmax) out.inverted_box = true;
        out.nodes.push_back(nd);
        bool leaf = !b.is_internal_node[n];
        uint32_t prims = 0;
        if (leaf) {
            uint32_t pa = leaf_prim(b.left_children_types[n], b.left_children_idxs[n]);
            uint32_t pb = pa;
            if (!(b.left_children_types[n] == b.right_children_types[n] && b.left_children_idxs[n] == b.right_children_idxs[n]))
                pb = leaf_prim(b.right_children_types[n], b.right_children_idxs[n]);
            prims = pa | (pb << 16);
        } else {
            emit_bvh_node(b, b.left_children_idxs[n], depth + 1);
            emit_bvh_node(b, b.right_children_idxs[n], depth + 1);
        }
        out.nodes[id].skip = (uint32_t)out.nodes.size() | (leaf ? 0x80000000u : 0u);
        out.nodes[id].prims = prims;
    }

    /* ---- boxes for this build's own trees (build_own_tree over reference BVH leaves, build_unified over everything else) ---- */
    struct Box { float lo[3], hi[3]; };
    static Box box_pad(Box b) {
        for (int k = 0; k < 3; k++) {
            float ext = b.hi[k] - b.lo[k];
            float mag = std::fabs(b.lo[k]) > std::fabs(b.hi[k]) ? std::fabs(b.lo[k]) : std::fabs(b.hi[k]);
            float pad = 1e-3f * ext + 1e-4f * mag + 1e-4f;
            b.lo[k] -= pad; b.hi[k] += pad;
        }
        return b;
    }
    /* ---- this build's own tree over the LEAF NODES of a reference BVH (mega_bvh.h walks it near-child-first).
     * Leaves keep the reference leaf node's box bit for bit; inner boxes are exact float unions of them, so the
     * reference's box test of a leaf implies the same test of every box above it (rounding is monotonic).  Split by
     * the surface-area heuristic (exhaustive sweep; a few hundred leaves), depth-limited so the device stack of
     * pending far children (MORT_OWN_STACK) can never overflow. ---- */
    struct OwnBuild {
        std::vector<Box> lb;       /* leaf boxes */
        std::vector<double> ldiag; /* leaf box diagonals */
        std::vector<DNode2> *nodes;
        int max_depth_seen = 0;
    };
    static Box box_union(Box a, const Box &b) {
        for (int k = 0; k < 3; k++) { a.lo[k] = std::fmin(a.lo[k], b.lo[k]); a.hi[k] = std::fmax(a.hi[k], b.hi[k]); }
        return a;
    }
    static double box_area(const Box &b) {
        const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
        return 2.0 * (x * y + y * z + z * x);
    }
    static double box_diag(const Box &b) {
        const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
        return std::sqrt(x * x + y * y + z * z);
    }
    /* returns the child reference of the subtree over ids[lo,hi); fills box / margin of that subtree */
    uint32_t own_emit(OwnBuild &ob, std::vector<int> &ids, int lo, int hi, int depth, Box &box_out, float &e_out) {
        const int n = hi - lo;
        Box u = ob.lb[ids[lo]];
        double ml = ob.ldiag[ids[lo]];
        for (int i = lo + 1; i < hi; i++) { u = box_union(u, ob.lb[ids[i]]); ml = std::fmax(ml, ob.ldiag[ids[i]]); }
        box_out = u;
        e_out = std::nextafter((float)(ml * 1.001 + 8e-3 * box_diag(u)), INFINITY);
        if (n == 1) return 0x8000u | (uint32_t)ids[lo];
        if (depth > ob.max_depth_seen) ob.max_depth_seen = depth;
        /* levels left below this node, this one included: both halves must fit in 2^(left-1) leaves */
        const int left = MORT_OWN_MAX_DEPTH - depth;
        const long long cap = left >= 2 ? (1ll << (left - 1 < 30 ? left - 1 : 30)) : 1;
        double best = 1e300; int bax = 0, bsplit = n / 2;
        std::vector<int> tmp(n);
        std::vector<double> ra(n);
        for (int ax = 0; ax < 3; ax++) {
            std::copy(ids.begin() + lo, ids.begin() + hi, tmp.begin());
            std::stable_sort(tmp.begin(), tmp.end(), [&](int a, int b) { return ob.lb[a].lo[ax] + ob.lb[a].hi[ax] < ob.lb[b].lo[ax] + ob.lb[b].hi[ax]; });
            Box r = ob.lb[tmp[n - 1]]; ra[n - 1] = box_area(r);
            for (int i = n - 2; i >= 0; i--) { r = box_union(r, ob.lb[tmp[i]]); ra[i] = box_area(r); }
            Box l = ob.lb[tmp[0]];
            for (int i = 1; i < n; i++) {
                if (i <= cap && n - i <= cap) {
                    const double c = box_area(l) * i + ra[i] * (n - i);
                    if (c < best) { best = c; bax = ax; bsplit = i; }
                }
                l = box_union(l, ob.lb[tmp[i]]);
            }
        }
        std::stable_sort(ids.begin() + lo, ids.begin() + hi, [&](int a, int b) { return ob.lb[a].lo[bax] + ob.lb[a].hi[bax] < ob.lb[b].lo[bax] + ob.lb[b].hi[bax]; });
        const size_t me = ob.nodes->size();
        ob.nodes->push_back(DNode2{});
        Box b0, b1; float e0, e1;
        const uint32_t c0 = own_emit(ob, ids, lo, lo + bsplit, depth + 1, b0, e0);
        const uint32_t c1 = own_emit(ob, ids, lo + bsplit, hi, depth + 1, b1, e1);
        DNode2 nd;
        nd.x0min = b0.lo[0]; nd.x0max = b0.hi[0]; nd.y0min = b0.lo[1]; nd.y0max = b0.hi[1]; nd.z0min = b0.lo[2]; nd.z0max = b0.hi[2];
        nd.x1min = b1.lo[0]; nd.x1max = b1.hi[0]; nd.y1min = b1.lo[1]; nd.y1max = b1.hi[1]; nd.z1min = b1.lo[2]; nd.z1max = b1.hi[2];
        nd.child0 = c0; nd.child1 = c1; nd.e0 = e0; nd.e1 = e1;
        (*ob.nodes)[me] = nd;
        return (uint32_t)me;
    }
    void build_own_tree() {
        out.own_nodes.clear(); out.own_leaves.clear(); out.own_depth = 0; out.own_nodes4.clear(); out.own4_stack = 0; out.own_leafrecs.clear();
        if (out.items.size() != 1 || out.items[0].kind != ITEM_BVH || !out.quads.empty()) return;
        const DItem &it = out.items[0];
        OwnBuild ob;
        ob.nodes = &out.own_nodes;
        std::vector<DBvhNode> leaves;
        for (int i = it.first; i < it.first + it.count; i++) {
            const DBvhNode &nd = out.nodes[i];
            if (!(nd.skip >> 31)) continue;
            if ((nd.prims & 0x8000u) || ((nd.prims >> 16) & 0x8000u)) return; /* spheres only */
            Box b;
            b.lo[0] = nd.xmin; b.hi[0] = nd.xmax; b.lo[1] = nd.ymin; b.hi[1] = nd.ymax; b.lo[2] = nd.zmin; b.hi[2] = nd.zmax;
            for (int k = 0; k < 3; k++) if (!std::isfinite(b.lo[k]) || !std::isfinite(b.hi[k]) || b.lo[k] > b.hi[k]) return;
            ob.lb.push_back(b); ob.ldiag.push_back(box_diag(b));
            leaves.push_back(nd);
        }
        const int n = (int)leaves.size();
        if (n < 2 || n > 0x7fff) return;
        std::vector<int> ids(n);
        for (int i = 0; i < n; i++) ids[i] = i;
        Box rb; float re;
        const uint32_t root = own_emit(ob, ids, 0, n, 0, rb, re);
        if (root != 0 || out.own_nodes.size() > 0x7fff || ob.max_depth_seen + 1 > MORT_OWN_MAX_DEPTH) { out.own_nodes.clear(); return; }
        for (const DNode2 &nd : out.own_nodes) if (!std::isfinite(nd.e0) || !std::isfinite(nd.e1)) { out.own_nodes.clear(); return; }
        out.own_leaves = leaves;
        out.own_depth = ob.max_depth_seen + 1;
        for (const DBvhNode &lf : leaves) {
            DLeaf2 r;
            std::memset(&r, 0, sizeof r);
            const uint32_t pa = lf.prims & 0x7fffu, pb = (lf.prims >> 16) & 0x7fffu;
            if (pa >= out.spheres.size() || pb >= out.spheres.size()) { out.own_nodes.clear(); out.own_leaves.clear(); return; }
            r.a = out.spheres[pa]; r.b = out.spheres[pb]; r.n = pa == pb ? 1u : 2u; r.prims = pa | (pb << 16);
            r.xmin = lf.xmin; r.xmax = lf.xmax; r.ymin = lf.ymin; r.ymax = lf.ymax; r.zmin = lf.zmin; r.zmax = lf.zmax;
            out.own_leafrecs.push_back(r);
        }
        collapse_own_tree();
    }
    /* ---- own_nodes four children wide.  A four-wide node stands for a binary node n and holds a CUT of n's subtree of at most four
     * members (inner binary nodes or leaves).  Which cut: the one that minimises the expected number of box steps of a walk under the
     * surface-area measure, cost(n) = area(n) + the least total cost of a cut of at most four -- a small dynamic program over
     * (node, members allowed), bottom up.  Member order is irrelevant to the result: the walk orders the children it enters by entry
     * distance, and equal hit distances are referred to the reference's walk whatever the order (mega_bvh.h).  Boxes and margins
     * are the binary tree's, bit for bit. ---- */
    struct Slot4 { uint32_t ref; float lo[3], hi[3], e; };
    static Slot4 slot_of(const DNode2 &nd, int k) {
        Slot4 s;
        s.ref = k ? nd.child1 : nd.child0; s.e = k ? nd.e1 : nd.e0;
        s.lo[0] = k ? nd.x1min : nd.x0min; s.hi[0] = k ? nd.x1max : nd.x0max;
        s.lo[1] = k ? nd.y1min : nd.y0min; s.hi[1] = k ? nd.y1max : nd.y0max;
        s.lo[2] = k ? nd.z1min : nd.z0min; s.hi[2] = k ? nd.z1max : nd.z0max;
        return s;
    }
    struct Cut4 { double t[5]; unsigned char left[5]; }; /* t[k]: least cost of a cut of at most k members; left[k]: members given to child 0 (0 = the node itself) */
    void cut_expand(const std::vector<Cut4> &dp, const Slot4 &s, int k, std::vector<Slot4> &outv) const {
        if ((s.ref & 0x8000u) || dp[s.ref].left[k] == 0) { outv.push_back(s); return; }
        const int i = dp[s.ref].left[k];
        cut_expand(dp, slot_of(out.own_nodes[s.ref], 0), i, outv);
        cut_expand(dp, slot_of(out.own_nodes[s.ref], 1), k - i, outv);
    }
    /* returns the most children a walk below this node can have pending: a step leaves at most (members - 1) behind */
    int collapse_emit(const std::vector<Cut4> &dp, uint32_t b2, uint32_t me) {
        std::vector<Slot4> sl;
        const int i0 = dp[b2].left[0]; /* [0]: the split of this node's OWN cut of four */
        cut_expand(dp, slot_of(out.own_nodes[b2], 0), i0, sl);
        cut_expand(dp, slot_of(out.own_nodes[b2], 1), 4 - i0, sl);
        int below = 0;
        DNode4 n4;
        std::memset(&n4, 0, sizeof n4);
        for (int k = 0; k < 4; k++) n4.child[k] = 0xffffu;
        for (size_t i = 0; i < sl.size() && i < 4; i++) {
            n4.xmin[i] = sl[i].lo[0]; n4.xmax[i] = sl[i].hi[0]; n4.ymin[i] = sl[i].lo[1]; n4.ymax[i] = sl[i].hi[1];
            n4.zmin[i] = sl[i].lo[2]; n4.zmax[i] = sl[i].hi[2]; n4.e[i] = sl[i].e;
            if (sl[i].ref & 0x8000u) { n4.child[i] = 0x8000u | ((sl[i].ref & 0x7fffu) * MORT_LEAF2_PIECES); continue; }
            const uint32_t ci = (uint32_t)out.own_nodes4.size();
            out.own_nodes4.push_back(DNode4{});
            n4.child[i] = ci * MORT_NODE4_PIECES;
            const int d = collapse_emit(dp, sl[i].ref, ci);
            if (d > below) below = d;
        }
        out.own_nodes4[me] = n4;
        return (int)sl.size() - 1 + below;
    }
    void collapse_own_tree() {
        out.own_nodes4.clear(); out.own4_stack = 0;
        const size_t n2 = out.own_nodes.size();
        if (n2 == 0) return;
        /* area of every binary node's own box: kept in its parent's record (the root: the union of its children) */
        std::vector<double> area(n2, 0.0);
        {
            const Slot4 a = slot_of(out.own_nodes[0], 0), b = slot_of(out.own_nodes[0], 1);
            Box u; for (int k = 0; k < 3; k++) { u.lo[k] = std::fmin(a.lo[k], b.lo[k]); u.hi[k] = std::fmax(a.hi[k], b.hi[k]); }
            area[0] = box_area(u);
        }
        for (size_t n = 0; n < n2; n++)
            for (int k = 0; k < 2; k++) {
                const Slot4 c = slot_of(out.own_nodes[n], k);
                if (c.ref & 0x8000u) continue;
                if (c.ref <= n || c.ref >= n2) return; /* children follow their parent (own_emit) */
                Box b; for (int j = 0; j < 3; j++) { b.lo[j] = c.lo[j]; b.hi[j] = c.hi[j]; }
                area[c.ref] = box_area(b);
            }
        std::vector<Cut4> dp(n2);
        for (size_t n = n2; n-- > 0;) {
            const DNode2 &nd = out.own_nodes[n];
            auto T = [&](uint32_t ref, int k) { return (ref & 0x8000u) ? 0.0 : dp[ref].t[k]; };
            Cut4 c;
            /* cuts of this node's subtree into i + j = k members, neither side empty */
            double best_k[5]; unsigned char arg_k[5];
            for (int k = 2; k <= 4; k++) {
                best_k[k] = 1e300; arg_k[k] = 1;
                for (int i = 1; i < k; i++) {
                    const double v = T(nd.child0, i) + T(nd.child1, k - i);
                    if (v < best_k[k]) { best_k[k] = v; arg_k[k] = (unsigned char)i; }
                }
            }
            const double own = area[n] + best_k[4];
            c.left[0] = arg_k[4];
            c.t[0] = own; c.t[1] = own; c.left[1] = 0;
            for (int k = 2; k <= 4; k++) {
                c.t[k] = c.t[k - 1]; c.left[k] = c.left[k - 1];
                if (best_k[k] < c.t[k]) { c.t[k] = best_k[k]; c.left[k] = arg_k[k]; }
            }
            dp[n] = c;
        }
        out.own_nodes4.push_back(DNode4{});
        out.own4_stack = collapse_emit(dp, 0, 0);
        if (out.own4_stack > MORT_OWN4_STACK || out.own_nodes4.size() * MORT_NODE4_PIECES > 0x7fff || out.own_leaves.size() * MORT_LEAF2_PIECES > 0x7fff) out.own_nodes4.clear();
    }

    /* ---- this build's UNIFIED tree (mega_gen.hip, wave_gen.hip).  A world without reference BVHs is a linear
     * scan over all its solid primitives (world.cuh:122-168; instances and lists flattened into runs that share a
     * transform chain) followed by its constant media.  The scan's result is: the primitive whose OWN hit test
     * succeeds with the smallest t, equal t -> the one scanned last (each test accepts t <= closest_so_far, and a
     * primitive's accepted root does not depend on t_max: the near root if it lies in range, else the far root, and a
     * root above t_max is never in range).  So any traversal that sees every
     * primitive whose own t is <= the final closest gives the same answer, exact ties being resolved by the scan
     * itself (the kernels fall back to it when two accepted hits have equal t).
     *
     * Boxes.  A sphere / quad test that accepts a root t_c places the point P(t_c) of the ray (in the primitive's
     * frame) within  delta  of the primitive:  quads: the computed point lies in the parallelogram up to the
     * rounding of alpha / beta and off its plane by |t_c - t| |n.d| <= a few ulp of M = the largest coordinate
     * involved;  spheres: the accepted discriminant is >= 0 up to its rounding error <= 10 u a M^2 (u = 2^-24), so
     * P(t_c) is at most  5 u M^2 / r  outside the sphere.  With M <= reach (ray origins are the camera centre and
     * points on / in scene objects) every primitive's box is padded by more than its delta, in its own frame, then
     * carried to world space (the 8 corners through the chain, in double) and padded again for the fp32 rounding
     * of the device's ray transform.  Hence: own t_c accepted  =>  the ray is inside the world box at parameter
     * t_c  =>  (exact) entry parameter <= t_c.  The device prunes a box only if the ray misses it by more than the
     * fp32 error band of its slab evaluation, or enters it later than closest_so_far by more than that band.
     * Media are evaluated after the tree with the final closest_so_far, as the scan does (their RNG draw depends
     * on it); a world with a solid scanned AFTER a medium is not given a unified tree. ---- */
    struct GPrim { int kind, idx, cf, cc, chain_id; Box wb; };
    static void chain_to_world(const std::vector<DXform> &xf, int cf, int cc, double p[3]) {
        for (int k = cc - 1; k >= 0; k--) { /* unapply_chain (dev_trace.h), in double */
            const DXform &x = xf[cf + k];
            if (x.kind == XF_TRANSLATE) { p[0] += x.a; p[1] += x.b; p[2] += x.c; }
            else { const double st = x.a, ct = x.b, px = p[0], pz = p[2]; p[0] = ct * px + st * pz; p[2] = -st * px + ct * pz; }
        }
    }
    /* ---- padded world boxes for this build's own trees (header of build_unified, steps 3 and 4): fills pr[i].wb.
     * Spheres: a test that accepts root t_c puts P(t_c) at most 5 u M^2 / r outside the sphere, M = max(|oc|, r).  Rays
     * whose origin is within 2R + 1 of the bounding sphere (G, R) of the ORDINARY spheres' centres are covered by the
     * static pad; a ray from further away widens its own error band by kmin * (|o - G| + R)^2 (device: gen_ray_setup).
     * GIANT spheres (radius above half the extent of all centres: a ground sphere, an enclosing shell) are left out of
     * (G, R) -- their centres would blow it up -- and carry their distance D_i from it in their own static pad and in kmin. ---- */
    struct PadInfo { Box all; float c[3]; float R, mnear, kmin, reach; bool ok; };
    PadInfo pad_prims(std::vector<GPrim> &pr, bool with_boundaries) {
        PadInfo pi;
        std::memset(&pi, 0, sizeof pi);
        pi.mnear = 1e30f;
        /* pass 1: unpadded world boxes of the solids and of the media boundaries -> where ray origins can lie */
        auto raw_box = [&](const GPrim &g, double pad_obj) {
            Box b;
            if (g.kind == ITEM_SPHERES) {
                const DSphere &s = out.spheres[g.idx];
                const float c0[3] = {s.cx, s.cy, s.cz}, c1[3] = {s.cx + s.vx, s.cy + s.vy, s.cz + s.vz};
                const float r = std::fabs(s.radius);
                for (int k = 0; k < 3; k++) { b.lo[k] = std::fmin(c0[k], c1[k]) - r; b.hi[k] = std::fmax(c0[k], c1[k]) + r; }
            } else {
                const DQuad &q = out.quads[g.idx];
                for (int k = 0; k < 3; k++) {
                    const float p0 = q.Q[k], p1 = q.Q[k] + q.u[k], p2 = q.Q[k] + q.v[k], p3 = q.Q[k] + q.u[k] + q.v[k];
                    b.lo[k] = std::fmin(std::fmin(p0, p1), std::fmin(p2, p3));
                    b.hi[k] = std::fmax(std::fmax(p0, p1), std::fmax(p2, p3));
                }
            }
            double lo[3], hi[3];
            for (int k = 0; k < 3; k++) { lo[k] = (double)b.lo[k] - pad_obj; hi[k] = (double)b.hi[k] + pad_obj; }
            Box w;
            for (int k = 0; k < 3; k++) { w.lo[k] = INFINITY; w.hi[k] = -INFINITY; }
            for (int c = 0; c < 8; c++) {
                double p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
                chain_to_world(out.xforms, g.cf, g.cc, p);
                for (int k = 0; k < 3; k++) {
                    w.lo[k] = std::fmin(w.lo[k], std::nextafter((float)p[k], -INFINITY));
                    w.hi[k] = std::fmax(w.hi[k], std::nextafter((float)p[k], INFINITY));
                }
            }
            return w;
        };
        Box all;
        for (int k = 0; k < 3; k++) { all.lo[k] = 0; all.hi[k] = 0; }
        bool have_all = false;
        auto grow = [&](const Box &w) {
            for (int k = 0; k < 3; k++) if (!std::isfinite(w.lo[k]) || !std::isfinite(w.hi[k]) || std::fabs(w.lo[k]) > 1e15f || std::fabs(w.hi[k]) > 1e15f) return false;
            all = have_all ? box_union(all, w) : w;
            have_all = true;
            return true;
        };
        for (const GPrim &g : pr) if (!grow(raw_box(g, 0.0))) return pi;
        if (with_boundaries) for (const DItem &it : out.subitems) { /* a medium scatters inside its boundary */
            if (it.kind != ITEM_SPHERES && it.kind != ITEM_QUADS) continue;
            for (int i = it.first; i < it.first + it.count; i++) {
                GPrim g; g.kind = it.kind; g.idx = i; g.cf = it.chain_first; g.cc = it.chain_count; g.chain_id = 0;
                if (!grow(raw_box(g, 0.0))) return pi;
            }
        }
        const double diag = have_all ? box_diag(all) : 1.0;
        double amag = 0;
        for (int k = 0; k < 3; k++) amag = std::fmax(amag, std::fmax(std::fabs((double)all.lo[k]), std::fabs((double)all.hi[k])));
        /* origins: the camera centre (checked per render to lie within `reach` of the box) or a point in the box */
        const double reach = 2.0 * diag + 4.0 * amag + 10.0;
        const double Mq = amag + reach + diag + 1.0; /* bound on every coordinate the quad test and the ray transform see */
        const double u = 5.9604644775390625e-08;
        /* world-space centres (both ends of a moving sphere's path) and radii */
        struct SC { double c0[3], c1[3], r; bool giant; double D; };
        std::vector<SC> sc;
        std::vector<size_t> sc_of(pr.size(), (size_t)-1);
        for (size_t i = 0; i < pr.size(); i++) {
            const GPrim &g = pr[i];
            if (g.kind != ITEM_SPHERES) continue;
            const DSphere &s = out.spheres[g.idx];
            SC e;
            for (int k = 0; k < 3; k++) { const double base[3] = {s.cx, s.cy, s.cz}, vel[3] = {s.vx, s.vy, s.vz}; e.c0[k] = base[k]; e.c1[k] = base[k] + vel[k]; }
            chain_to_world(out.xforms, g.cf, g.cc, e.c0); chain_to_world(out.xforms, g.cf, g.cc, e.c1);
            e.r = std::fabs((double)s.radius); e.giant = false; e.D = 0;
            sc_of[i] = sc.size();
            sc.push_back(e);
        }
        double G[3] = {0, 0, 0}, gR = 0, mnear = 1e30, kmin = 0;
        if (!sc.empty()) {
            auto centre_box = [&](bool skip_giants, double lo[3], double hi[3]) {
                bool any = false;
                for (const SC &e : sc) {
                    if (skip_giants && e.giant) continue;
                    for (int k = 0; k < 3; k++) {
                        const double a0 = std::fmin(e.c0[k], e.c1[k]), a1 = std::fmax(e.c0[k], e.c1[k]);
                        lo[k] = any ? std::fmin(lo[k], a0) : a0; hi[k] = any ? std::fmax(hi[k], a1) : a1;
                    }
                    any = true;
                }
                return any;
            };
            double lo[3], hi[3];
            centre_box(false, lo, hi);
            const double dall = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
            size_t ngiant = 0;
            for (SC &e : sc) { e.giant = sc.size() > 1 && e.r > 0.5 * dall; ngiant += e.giant ? 1 : 0; }
            if (ngiant == sc.size()) for (SC &e : sc) e.giant = false; /* nothing ordinary to centre on: treat all alike */
            centre_box(true, lo, hi);
            double d2 = 0;
            for (int k = 0; k < 3; k++) { G[k] = 0.5 * (lo[k] + hi[k]); const double h = 0.5 * (hi[k] - lo[k]); d2 += h * h; }
            gR = std::sqrt(d2) * 1.0001 + 1e-3 * (1.0 + amag); /* + the rounding of G and of the device's distance */
            mnear = 3.0 * gR + 1.0;
            for (SC &e : sc) {
                double far2 = 0;
                for (int end = 0; end < 2; end++) { double q = 0; for (int k = 0; k < 3; k++) { const double t = (end ? e.c1[k] : e.c0[k]) - G[k]; q += t * t; } far2 = std::fmax(far2, q); }
                e.D = std::fmax(0.0, std::sqrt(far2) - gR);
                const double ki = (e.r > 0) ? 20.0 * u * (1.0 + e.D / mnear) * (1.0 + e.D / mnear) / e.r : INFINITY;
                kmin = std::fmax(kmin, ki);
            }
        }
        for (int k = 0; k < 3; k++) pi.c[k] = (float)G[k];
        pi.R = (float)gR; pi.mnear = (float)mnear; pi.kmin = std::nextafter((float)kmin, INFINITY);
        for (GPrim &g : pr) {
            double delta = 64.0 * u * Mq + 1e-4; /* quads; the rounding of ray_at and of the ray transform */
            if (g.kind == ITEM_SPHERES) { /* near rays: |o - G| <= 2R + 1, so |oc| <= mnear + D_i; far rays widen their own band (kmin) */
                const SC &e = sc[sc_of[(size_t)(&g - &pr[0])]];
                const double M = std::fmax(e.r, mnear + e.D);
                delta += (e.r > 0) ? 20.0 * u * M * M / e.r : INFINITY;
            }
            g.wb = box_pad(raw_box(g, delta));
            for (int k = 0; k < 3; k++) {
                g.wb.lo[k] -= (float)(64.0 * u * Mq); g.wb.hi[k] += (float)(64.0 * u * Mq);
                if (!std::isfinite(g.wb.lo[k]) || !std::isfinite(g.wb.hi[k])) return pi;
            }
        }
        pi.all = all;
        pi.reach = (float)reach;
        pi.ok = true;
        return pi;
    }

    uint32_t gen_emit(std::vector<GPrim> &pr, std::vector<int> &ids, int lo, int hi, int depth, Box &box_out) {
        const int n = hi - lo;
        Box u = pr[ids[lo]].wb;
        for (int i = lo + 1; i < hi; i++) u = box_union(u, pr[ids[i]].wb);
        box_out = u;
        if (depth > out.g_depth) out.g_depth = depth;
        const int left = MORT_OWN_MAX_DEPTH - depth; /* node levels left below this point */
        if (n <= MORT_GEN_LEAF_MAX || left <= 0) { /* a leaf is a run of entries: 0x8000 | (n - 1) << 13 | first entry */
            if (n > 4 || out.g_entries.size() + (size_t)n > 0x2000u) { gen_too_big = true; return 0xffffu; }
            const size_t first = out.g_entries.size();
            for (int i = lo; i < hi; i++) {
                const GPrim &g = pr[ids[i]];
                out.g_entries.push_back(GENT(g.kind == ITEM_QUADS ? 1u : 0u, (uint32_t)g.chain_id, (uint32_t)g.idx));
            }
            return 0x8000u | ((uint32_t)(n - 1) << 13) | (uint32_t)first;
        }
        /* both halves must fit below: at most LEAF_MAX * 2^(left-1) primitives each */
        const long long cap = (long long)MORT_GEN_LEAF_MAX << (left - 1 < 40 ? left - 1 : 40);
        double best = 1e300; int bax = 0, bsplit = n / 2;
        std::vector<int> tmp(n);
        std::vector<double> ra(n);
        for (int ax = 0; ax < 3; ax++) {
            std::copy(ids.begin() + lo, ids.begin() + hi, tmp.begin());
            std::stable_sort(tmp.begin(), tmp.end(), [&](int a, int b) { return pr[a].wb.lo[ax] + pr[a].wb.hi[ax] < pr[b].wb.lo[ax] + pr[b].wb.hi[ax]; });
            Box r = pr[tmp[n - 1]].wb; ra[n - 1] = box_area(r);
            for (int i = n - 2; i >= 0; i--) { r = box_union(r, pr[tmp[i]].wb); ra[i] = box_area(r); }
            Box l = pr[tmp[0]].wb;
            for (int i = 1; i < n; i++) {
                if (i <= cap && n - i <= cap) {
                    const double c = box_area(l) * i + ra[i] * (n - i);
                    if (c < best) { best = c; bax = ax; bsplit = i; }
                }
                l = box_union(l, pr[tmp[i]].wb);
            }
        }
        std::stable_sort(ids.begin() + lo, ids.begin() + hi, [&](int a, int b) { return pr[a].wb.lo[bax] + pr[a].wb.hi[bax] < pr[b].wb.lo[bax] + pr[b].wb.hi[bax]; });
        const size_t me = out.g_nodes.size();
        out.g_nodes.push_back(DNodeQ{});
        Box b0, b1;
        const uint32_t c0 = gen_emit(pr, ids, lo, lo + bsplit, depth + 1, b0);
        const uint32_t c1 = gen_emit(pr, ids, lo + bsplit, hi, depth + 1, b1);
        if (gen_too_big) return 0xffffu;
        out.g_nodes[me] = quantize_node(b0, b1, c0, c1);
        return (uint32_t)me;
    }
    bool gen_too_big = false; /* more entries than a leaf reference can address: no unified tree for this world */
    /* the two child boxes as 8-bit offsets from the node's corner (dev_scene.h DNodeQ): every plane rounded outward and checked with the
     * decoder's own arithmetic, fmaf((float)q, step, origin), so the device sees boxes that contain b0 and b1 */
    static DNodeQ quantize_node(const Box &b0, const Box &b1, uint32_t c0, uint32_t c1) {
        DNodeQ nd;
        std::memset(&nd, 0, sizeof nd);
        float org[3]; unsigned ex[3]; unsigned q[2][3][2];
        for (int a = 0; a < 3; a++) {
            org[a] = std::fmin(b0.lo[a], b1.lo[a]);
            const double ext = (double)std::fmax(b0.hi[a], b1.hi[a]) - (double)org[a];
            int k = ext > 0 ? (int)std::ceil(std::log2(ext / 255.0)) : -126;
            if (k < -126) k = -126;
            if (k > 127) k = 127;
            for (;;) { /* the step is 2^k; widen it until both boxes fit in 0..255 steps */
                const float step = std::ldexp(1.0f, k);
                bool ok = true;
                for (int c = 0; c < 2 && ok; c++) {
                    const Box &b = c ? b1 : b0;
                    long long ql = (long long)std::floor(((double)b.lo[a] - (double)org[a]) / (double)step);
                    if (ql < 0) ql = 0;
                    if (ql > 255) ql = 255;
                    while (ql > 0 && std::fmaf((float)ql, step, org[a]) > b.lo[a]) ql--;
                    long long qh = (long long)std::ceil(((double)b.hi[a] - (double)org[a]) / (double)step);
                    if (qh < 0) qh = 0;
                    while (qh <= 255 && std::fmaf((float)qh, step, org[a]) < b.hi[a]) qh++;
                    if (qh > 255 || std::fmaf((float)ql, step, org[a]) > b.lo[a]) { ok = false; break; }
                    q[c][a][0] = (unsigned)ql; q[c][a][1] = (unsigned)qh;
                }
                if (ok || k >= 127) { ex[a] = (unsigned)(k + 127); break; }
                k++;
            }
        }
        nd.ox = org[0]; nd.oy = org[1]; nd.oz = org[2];
        nd.exps = ex[0] | (ex[1] << 8) | (ex[2] << 16);
        nd.q0 = q[0][0][0] | (q[0][0][1] << 8) | (q[0][1][0] << 16) | (q[0][1][1] << 24);
        nd.q1 = q[0][2][0] | (q[0][2][1] << 8) | (q[1][0][0] << 16) | (q[1][0][1] << 24);
        nd.q2 = q[1][1][0] | (q[1][1][1] << 8) | (q[1][2][0] << 16) | (q[1][2][1] << 24);
        nd.children = (c0 & 0xffffu) | (c1 << 16);
        return nd;
    }
    void build_unified() {
        out.g_ok = false;
        out.g_nodes.clear(); out.g_entries.clear(); out.g_chains.clear(); out.g_ranks.clear();
        gen_too_big = false;
        out.g_root = 0xffffu; out.g_depth = 0;
        out.g_chains.push_back(0); out.g_chains.push_back(0); /* id 0: no transform */
        std::vector<GPrim> pr;
        bool seen_medium = false;
        out.g_first_medium = (int)out.items.size();
        for (size_t ii = 0; ii < out.items.size(); ii++) {
            const DItem &it = out.items[ii];
            if (it.kind == ITEM_BVH) return;
            if (it.kind == ITEM_MEDIUM) { if (!seen_medium) out.g_first_medium = (int)ii; seen_medium = true; continue; }
            if (seen_medium) return; /* a solid scanned after a medium: order matters, no unified tree */
            int chain_id = 0;
            if (it.chain_count > 0) {
                for (size_t k = 1; k < out.g_chains.size() / 2; k++)
                    if (out.g_chains[2 * k] == it.chain_first && out.g_chains[2 * k + 1] == it.chain_count) chain_id = (int)k;
                if (!chain_id) { chain_id = (int)(out.g_chains.size() / 2); out.g_chains.push_back(it.chain_first); out.g_chains.push_back(it.chain_count); }
                if (chain_id > 127) return;
            }
            for (int i = it.first; i < it.first + it.count; i++) {
                GPrim g; g.kind = it.kind; g.idx = i; g.cf = it.chain_first; g.cc = it.chain_count; g.chain_id = chain_id;
                pr.push_back(g);
            }
        }
        if (pr.size() > 0xffffffu) return;
        out.g_ranks.assign(out.spheres.size() + out.quads.size(), 0u);
        for (size_t i = 0; i < pr.size(); i++) /* pr is in scan order */
            out.g_ranks[(pr[i].kind == ITEM_QUADS ? out.spheres.size() : 0) + (size_t)pr[i].idx] = (uint32_t)i + 1u;
        const PadInfo pi = pad_prims(pr, true);
        if (!pi.ok) return;
        for (int k = 0; k < 3; k++) { out.g_lo[k] = pi.all.lo[k]; out.g_hi[k] = pi.all.hi[k]; out.g_c[k] = pi.c[k]; }
        out.g_R = pi.R; out.g_mnear = pi.mnear; out.g_kmin = pi.kmin;
        out.g_reach = pi.reach;
        if (!pr.empty()) {
            std::vector<int> ids(pr.size());
            for (size_t i = 0; i < pr.size(); i++) ids[i] = (int)i;
            Box rb;
            out.g_root = gen_emit(pr, ids, 0, (int)pr.size(), 0, rb);
            if (out.status != MORT_OK) return;
            if (gen_too_big || out.g_nodes.size() > 0x7fff) return;
        }
        out.g_ok = true;
    }

    static uint32_t tex_ref(int type, int idx) { return DREF(type & 0x7fff, idx & 0xffff); }

    DLambert tex_material(int tex_type, int tex_idx) {
        DLambert d;
        d.r = d.g = d.b = 0; d.tex = tex_ref(tex_type, tex_idx);
        if (tex_type == MORT_TEXTURE_SOLID && tex_idx >= 0 && tex_idx < w->texs.num_solid_colors) {
            const mort_vec3 &c = w->texs.host_solid_color[tex_idx].color_value;
            d.r = c.e[0]; d.g = c.e[1]; d.b = c.e[2]; d.tex = 0; /* colour inlined */
        }
        return d;
    }

    void run() {
        const mort_world_objects &o = w->objs;
        /* world::hit order (world.cuh:110-168) */
        for (int i = 0; i < o.num_bvh; i++) {
            if (o.host_bvh[i].skip) continue;
            DItem it;
            std::memset(&it, 0, sizeof it);
            it.kind = ITEM_BVH; it.first = (int)out.nodes.size();
            emit_bvh_node(o.host_bvh[i], 0, 0);
            it.count = (int)out.nodes.size() - it.first;
            out.items.push_back(it);
        }
        if (!w->bvh_mode) {
            std::vector<Ref> refs;
            for (int i = 0; i < o.num_spheres; i++) if (!o.host_sphere[i].skip) refs.push_back({1, i, 0, 0});
            for (int i = 0; i < o.num_quads; i++) if (!o.host_quad[i].skip) refs.push_back({2, i, 0, 0});
            for (int i = 0; i < o.num_translates; i++) if (!o.host_translate[i].skip) flatten(MORT_OBJ_TRANSLATE, i, 0, 0, 0, refs);
            for (int i = 0; i < o.num_rotate_y; i++) if (!o.host_rotate_y[i].skip) flatten(MORT_OBJ_ROTATE_Y, i, 0, 0, 0, refs);
            for (int i = 0; i < o.num_constant_medium; i++) if (!o.host_constant_medium[i].skip) refs.push_back({3, i, 0, 0});
            for (int i = 0; i < o.num_hittable_list; i++) if (!o.host_hittable_list[i].skip) flatten(MORT_OBJ_HITTABLE_LIST, i, 0, 0, 0, refs);
            emit_items(refs, out.items, true);
        }
        build_own_tree();
        build_unified();
        /* world-order copies for light sampling */
        for (int i = 0; i < o.num_spheres; i++) out.wspheres.push_back(to_dsphere(o.host_sphere[i]));
        for (int i = 0; i < o.num_quads; i++) out.wquads.push_back(to_dquad(o.host_quad[i]));
        for (int i = 0; i < MORT_NUM_HITTABLE_LIST; i++) { out.list_first[i] = 0; out.list_count[i] = 0; }
        for (int i = 0; i < o.num_hittable_list && i < MORT_NUM_HITTABLE_LIST; i++) {
            const mort_hittable_list &l = o.host_hittable_list[i];
            out.list_first[i] = (int)out.list_types.size();
            out.list_count[i] = l.num_objs;
            for (int k = 0; k < l.num_objs; k++) { out.list_types.push_back(l.obj_types[k]); out.list_idxs.push_back(l.obj_idxs[k]); }
        }
        /* materials */
        const mort_world_materials &m = w->mats;
        for (int i = 0; i < m.num_lambertians; i++) out.lambert.push_back(tex_material(m.host_lambertian[i].texType, m.host_lambertian[i].texIdx));
        for (int i = 0; i < m.num_diffuse_lights; i++) out.dlight.push_back(tex_material(m.host_diffuse_light[i].texType, m.host_diffuse_light[i].texIdx));
        for (int i = 0; i < m.num_isotropics; i++) out.isotropic.push_back(tex_material(m.host_isotropic[i].texType, m.host_isotropic[i].texIdx));
        for (int i = 0; i < m.num_metals; i++) {
            DMetal d; d.r = m.host_metal[i].albedo.e[0]; d.g = m.host_metal[i].albedo.e[1]; d.b = m.host_metal[i].albedo.e[2]; d.fuzz = m.host_metal[i].fuzz;
            out.metal.push_back(d);
        }
        for (int i = 0; i < m.num_dielectrics; i++) { DDielectric d; d.ior = m.host_dielectric[i].ior; d.inv_ior = m.host_dielectric[i].inv_ior; out.dielectric.push_back(d); }
        /* textures */
        const mort_world_textures &t = w->texs;
        for (int i = 0; i < t.num_solid_colors; i++) {
            DSolid d; d.r = t.host_solid_color[i].color_value.e[0]; d.g = t.host_solid_color[i].color_value.e[1]; d.b = t.host_solid_color[i].color_value.e[2]; d.pad = 0;
            out.solid.push_back(d);
        }
        for (int i = 0; i < t.num_checker_textures; i++) {
            const mort_checker_texture &c = t.host_checker_texture[i];
            DChecker d; d.inv_scale = c.inv_scale; d.even = tex_ref(c.evenTextureType, c.evenTextureIdx); d.odd = tex_ref(c.oddTextureType, c.oddTextureIdx); d.pad = 0;
            out.checker.push_back(d);
        }
        for (int i = 0; i < t.num_image_textures; i++) {
            const mort_image_texture &im = t.host_image_texture[i];
            DImage d; d.offset = (uint32_t)out.texels.size(); d.width = im.texels ? im.width : 0; d.height = im.texels ? im.height : 0; d.pad = 0;
            if (im.texels && im.width > 0 && im.height > 0)
                out.texels.insert(out.texels.end(), im.texels, im.texels + (size_t)im.width * im.height * 3);
            out.image.push_back(d);
        }
        for (int i = 0; i < t.num_noise_textures; i++) {
            const unsigned char *p = (const unsigned char *)&t.host_noise_texture[i];
            out.noise.insert(out.noise.end(), p, p + sizeof(mort_noise_texture));
        }
    }
};

} // namespace mortc

#endif
