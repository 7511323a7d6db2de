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

    /* ---- this build's own acceleration of large linear runs (world.cuh:122-136 and hittable_list::hit scan every
     * primitive).  The closest hit of a run does not depend on visiting order once ties are resolved the way the
     * scan resolves them (a later primitive replaces an equal t), so a BVH over the run gives the same answer as
     * long as its boxes never hide a primitive the scan would accept: boxes are padded and the device prunes with
     * a relative margin well above the evaluation error of sphere::hit / quad::hit (dev_trace.h, run_accel). ---- */
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
    Box prim_box(int kind, int idx) const {
        Box b;
        if (kind == ITEM_SPHERES) {
            const DSphere &s = out.spheres[idx];
            const float c0[3] = {s.cx, s.cy, s.cz}, c1[3] = {s.cx + s.vx, s.cy + s.vy, s.cz + s.vz};
            for (int k = 0; k < 3; k++) {
                b.lo[k] = std::fmin(c0[k], c1[k]) - s.radius;
                b.hi[k] = std::fmax(c0[k], c1[k]) + s.radius;
            }
        } else {
            const DQuad &q = out.quads[idx];
            for (int k = 0; k < 3; k++) {
                const float p0 = q.Q[k], p1 = q.Q[k] + q.u[k], p2 = q.Q[k] + q.v[k], p3 = q.Q[k] + q.u[k] + q.v[k];
                b.lo[k] = std::fmin(std::fmin(p0, p1), std::fmin(p2, p3));
                b.hi[k] = std::fmax(std::fmax(p0, p1), std::fmax(p2, p3));
            }
        }
        return box_pad(b);
    }
    void accel_emit(int kind, std::vector<int> &ids, const std::vector<Box> &boxes, int lo, int hi) {
        const size_t id = out.nodes.size();
        Box u = boxes[ids[lo]];
        float cmin[3], cmax[3];
        for (int k = 0; k < 3; k++) { cmin[k] = 1e30f; cmax[k] = -1e30f; }
        for (int i = lo; i < hi; i++) {
            const Box &b = boxes[ids[i]];
            for (int k = 0; k < 3; k++) {
                u.lo[k] = std::fmin(u.lo[k], b.lo[k]); u.hi[k] = std::fmax(u.hi[k], b.hi[k]);
                const float c = 0.5f * (b.lo[k] + b.hi[k]);
                cmin[k] = std::fmin(cmin[k], c); cmax[k] = std::fmax(cmax[k], c);
            }
        }
        DBvhNode nd;
        std::memset(&nd, 0, sizeof nd);
        nd.xmin = u.lo[0]; nd.xmax = u.hi[0]; nd.ymin = u.lo[1]; nd.ymax = u.hi[1]; nd.zmin = u.lo[2]; nd.zmax = u.hi[2];
        out.nodes.push_back(nd);
        const int n = hi - lo;
        bool leaf = n <= 2;
        uint32_t prims = 0;
        if (leaf) {
            const uint32_t kbit = (kind == ITEM_QUADS) ? 0x8000u : 0u;
            const uint32_t pa = kbit | (uint32_t)ids[lo], pb = kbit | (uint32_t)ids[hi - 1];
            prims = pa | (pb << 16);
        } else {
            /* surface-area heuristic, exhaustive sweep over the three axes (a few thousand primitives at most);
             * the median of the widest axis when every candidate is degenerate */
            int axis = 0;
            if (cmax[1] - cmin[1] > cmax[axis] - cmin[axis]) axis = 1;
            if (cmax[2] - cmin[2] > cmax[axis] - cmin[axis]) axis = 2;
            int mid = lo + n / 2;
            if (!std::getenv("MORT_ACCEL_MEDIAN")) {
                double best = 1e300;
                int bax = -1, bsplit = n / 2;
                std::vector<int> tmp(n);
                std::vector<double> ra(n);
                auto area = [](const Box &b) {
                    const double x = (double)b.hi[0] - b.lo[0], y = (double)b.hi[1] - b.lo[1], z = (double)b.hi[2] - b.lo[2];
                    return 2.0 * (x * y + y * z + z * x);
                };
                auto uni = [](Box a, const Box &b) { for (int k = 0; k < 3; k++) { a.lo[k] = std::fmin(a.lo[k], b.lo[k]); a.hi[k] = std::fmax(a.hi[k], b.hi[k]); } return a; };
                for (int ax = 0; ax < 3; ax++) {
                    std::copy(ids.begin() + lo, ids.begin() + hi, tmp.begin());
                    std::stable_sort(tmp.begin(), tmp.end(), [&](int a, int b) { return boxes[a].lo[ax] + boxes[a].hi[ax] < boxes[b].lo[ax] + boxes[b].hi[ax]; });
                    Box r = boxes[tmp[n - 1]]; ra[n - 1] = area(r);
                    for (int i = n - 2; i >= 0; i--) { r = uni(r, boxes[tmp[i]]); ra[i] = area(r); }
                    Box l = boxes[tmp[0]];
                    for (int i = 1; i < n; i++) {
                        /* keep the tree's depth bounded: no side smaller than an eighth */
                        if (i >= n / 8 && n - i >= n / 8) {
                            const double c = area(l) * i + ra[i] * (n - i);
                            if (c < best) { best = c; bax = ax; bsplit = i; }
                        }
                        l = uni(l, boxes[tmp[i]]);
                    }
                }
                if (bax >= 0) {
                    std::stable_sort(ids.begin() + lo, ids.begin() + hi, [&](int a, int b) { return boxes[a].lo[bax] + boxes[a].hi[bax] < boxes[b].lo[bax] + boxes[b].hi[bax]; });
                    mid = lo + bsplit;
                    axis = -1;
                }
            }
            if (axis >= 0)
            std::nth_element(ids.begin() + lo, ids.begin() + mid, ids.begin() + hi, [&](int a, int b) {
                const float ca = boxes[a].lo[axis] + boxes[a].hi[axis], cb = boxes[b].lo[axis] + boxes[b].hi[axis];
                return ca < cb || (ca == cb && a < b);
            });
            accel_emit(kind, ids, boxes, lo, mid);
            accel_emit(kind, ids, boxes, mid, hi);
        }
        out.nodes[id].skip = (uint32_t)out.nodes.size() | (leaf ? 0x80000000u : 0u);
        out.nodes[id].prims = prims;
    }
    void build_accels(std::vector<DItem> &items) {
        for (DItem &it : items) {
            it.accel_first = 0;
            if (it.kind != ITEM_SPHERES && it.kind != ITEM_QUADS) continue;
            it.medium = 0; /* accel_count */
            if (it.count < 16 || it.first + it.count > 0x7fff) continue;
            std::vector<Box> boxes(it.first + it.count);
            std::vector<int> ids;
            bool finite = true;
            for (int i = it.first; i < it.first + it.count; i++) {
                boxes[i] = prim_box(it.kind, i);
                for (int k = 0; k < 3; k++) if (!std::isfinite(boxes[i].lo[k]) || !std::isfinite(boxes[i].hi[k])) finite = false;
                ids.push_back(i);
            }
            if (!finite) continue;
            it.accel_first = (int)out.nodes.size();
            accel_emit(it.kind, ids, boxes, 0, (int)ids.size());
            it.medium = (int)out.nodes.size() - it.accel_first;
        }
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
        out.own_nodes.clear(); out.own_leaves.clear(); out.own_depth = 0;
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
        if (!std::getenv("MORT_NO_ACCEL")) { build_accels(out.items); build_accels(out.subitems); }
        else { for (DItem &it : out.items) { it.accel_first = 0; if (it.kind == ITEM_SPHERES || it.kind == ITEM_QUADS) it.medium = 0; }
               for (DItem &it : out.subitems) { it.accel_first = 0; if (it.kind == ITEM_SPHERES || it.kind == ITEM_QUADS) it.medium = 0; } }
        build_own_tree();
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
