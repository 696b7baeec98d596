/*
 * kdtree.c -- ORACLE (test infrastructure): exact nearest-neighbour search that
 * stands in for pcl::KdTreeFLANN (FLANN KDTreeSingleIndex, eps = 0), which PCL's
 * CorrespondenceEstimation::determineCorrespondences drives once per source
 * point per ICP iteration (reference call sites: wave_matching/src/icp.cpp:95,116,126
 * -> icp.align(); wave_matching/src/icp_pcl_functions.cpp:67-80).
 *
 * FLANN is a third-party dependency absent from /root/reference; what is
 * restated is its contract: the exact Euclidean 1-NN / k-NN in float.  Ties (equal
 * float d2) are resolved towards the lowest point index so that results are
 * reproducible and comparable bit-for-bit with the HIP kernels.
 */
#include "wm_oracle.h"
#include "wmo_internal.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LEAF_SIZE 12

typedef struct {
    int dim;   /* -1 = leaf */
    float split;
    int left, right;
    int lo, hi; /* leaf point range in tree order */
} kd_node;

struct wmo_kdtree {
    int n;
    float *pts; /* tree-ordered copy, n x 3 */
    int *perm;  /* tree position -> original index */
    kd_node *nodes;
    int n_nodes, cap_nodes;
    float bbmin[3], bbmax[3];
};

static int new_node(wmo_kdtree *t) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? t->cap_nodes * 2 : 1024;
        t->nodes = (kd_node *) realloc(t->nodes, sizeof(kd_node) * t->cap_nodes);
    }
    return t->n_nodes++;
}

/* quickselect on perm[lo,hi) by coordinate d of the ORIGINAL array */
static void select_nth(const float *xyz, int *perm, int lo, int hi, int nth, int d) {
    while (hi - lo > 1) {
        int i = lo, j = hi - 1;
        float pivot = xyz[3 * perm[lo + (hi - lo) / 2] + d];
        while (i <= j) {
            while (xyz[3 * perm[i] + d] < pivot) ++i;
            while (xyz[3 * perm[j] + d] > pivot) --j;
            if (i <= j) {
                int tmp = perm[i];
                perm[i] = perm[j];
                perm[j] = tmp;
                ++i;
                --j;
            }
        }
        if (nth <= j)
            hi = j + 1;
        else if (nth >= i)
            lo = i;
        else
            return;
    }
}

static int build_rec(wmo_kdtree *t, const float *xyz, int lo, int hi) {
    int id = new_node(t);
    if (hi - lo <= LEAF_SIZE) {
        t->nodes[id].dim = -1;
        t->nodes[id].lo = lo;
        t->nodes[id].hi = hi;
        t->nodes[id].left = t->nodes[id].right = -1;
        return id;
    }
    {
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        int i, d, best = 0, mid, l, r;
        for (i = lo; i < hi; ++i)
            for (d = 0; d < 3; ++d) {
                float v = xyz[3 * t->perm[i] + d];
                if (v < mn[d]) mn[d] = v;
                if (v > mx[d]) mx[d] = v;
            }
        for (d = 1; d < 3; ++d)
            if (mx[d] - mn[d] > mx[best] - mn[best]) best = d;
        if (mx[best] == mn[best]) { /* all points identical: make a (large) leaf */
            t->nodes[id].dim = -1;
            t->nodes[id].lo = lo;
            t->nodes[id].hi = hi;
            t->nodes[id].left = t->nodes[id].right = -1;
            return id;
        }
        mid = lo + (hi - lo) / 2;
        select_nth(xyz, t->perm, lo, hi, mid, best);
        t->nodes[id].dim = best;
        t->nodes[id].split = xyz[3 * t->perm[mid] + best];
        t->nodes[id].lo = lo;
        t->nodes[id].hi = hi;
        l = build_rec(t, xyz, lo, mid);
        r = build_rec(t, xyz, mid, hi);
        t->nodes[id].left = l; /* t->nodes may have been realloc'd: index, not pointer */
        t->nodes[id].right = r;
    }
    return id;
}

wmo_kdtree *wmo_kdtree_build(const float *xyz, int n) {
    wmo_kdtree *t = (wmo_kdtree *) calloc(1, sizeof(wmo_kdtree));
    int i, d;
    t->n = n;
    t->perm = (int *) malloc(sizeof(int) * (n > 0 ? n : 1));
    t->pts = (float *) malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    for (i = 0; i < n; ++i) t->perm[i] = i;
    for (d = 0; d < 3; ++d) {
        t->bbmin[d] = FLT_MAX;
        t->bbmax[d] = -FLT_MAX;
    }
    for (i = 0; i < n; ++i)
        for (d = 0; d < 3; ++d) {
            float v = xyz[3 * i + d];
            if (v < t->bbmin[d]) t->bbmin[d] = v;
            if (v > t->bbmax[d]) t->bbmax[d] = v;
        }
    if (n > 0) build_rec(t, xyz, 0, n);
    for (i = 0; i < n; ++i) memcpy(t->pts + 3 * i, xyz + 3 * t->perm[i], 3 * sizeof(float));
    return t;
}

void wmo_kdtree_free(wmo_kdtree *t) {
    if (!t) return;
    free(t->pts);
    free(t->perm);
    free(t->nodes);
    free(t);
}

/* ---- k-NN result set: ascending (d2, idx), capacity k ---- */
typedef struct {
    int k, count;
    int *idx;
    float *d2;
} knn_set;

static inline float worst_d2(const knn_set *s) {
    return s->count < s->k ? FLT_MAX : s->d2[s->k - 1];
}

static inline void knn_insert(knn_set *s, float d, int id) {
    int pos;
    if (s->count == s->k) {
        float wd = s->d2[s->k - 1];
        int wi = s->idx[s->k - 1];
        if (!(d < wd || (d == wd && id < wi))) return;
        pos = s->k - 1;
    } else {
        pos = s->count++;
    }
    while (pos > 0 && (s->d2[pos - 1] > d || (s->d2[pos - 1] == d && s->idx[pos - 1] > id))) {
        s->d2[pos] = s->d2[pos - 1];
        s->idx[pos] = s->idx[pos - 1];
        --pos;
    }
    s->d2[pos] = d;
    s->idx[pos] = id;
}

static void search_rec(const wmo_kdtree *t, int id, const float *q, double mind, double *dists,
                       knn_set *s) {
    const kd_node *nd = &t->nodes[id];
    if (nd->dim < 0) {
        int i;
        for (i = nd->lo; i < nd->hi; ++i) {
            float d = wmo_d2(q, t->pts + 3 * i);
            knn_insert(s, d, t->perm[i]);
        }
        return;
    }
    {
        int d = nd->dim;
        double diff = (double) q[d] - (double) nd->split;
        int nearc = diff < 0 ? nd->left : nd->right;
        int farc = diff < 0 ? nd->right : nd->left;
        double cut = diff * diff, old = dists[d], far_mind;
        search_rec(t, nearc, q, mind, dists, s);
        far_mind = mind - old + cut;
        /* keep ties and absorb float rounding of the canonical d2 */
        if (far_mind * (1.0 - 1e-6) <= (double) worst_d2(s)) {
            dists[d] = cut;
            search_rec(t, farc, q, far_mind, dists, s);
            dists[d] = old;
        }
    }
}

int wmo_kdtree_knn(const wmo_kdtree *t, const float q[3], int k, int *idx, float *d2) {
    knn_set s;
    double dists[3], mind = 0;
    int d;
    if (t->n == 0 || k <= 0) return 0;
    s.k = k < t->n ? k : t->n;
    s.count = 0;
    s.idx = idx;
    s.d2 = d2;
    for (d = 0; d < 3; ++d) {
        double o = 0;
        if (q[d] < t->bbmin[d]) o = (double) t->bbmin[d] - q[d];
        if (q[d] > t->bbmax[d]) o = (double) q[d] - t->bbmax[d];
        dists[d] = o * o;
        mind += dists[d];
    }
    search_rec(t, 0, q, mind, dists, &s);
    return s.count;
}

int wmo_kdtree_nn(const wmo_kdtree *t, const float q[3], float *d2) {
    int idx = -1;
    float d = FLT_MAX;
    if (wmo_kdtree_knn(t, q, 1, &idx, &d) == 0) {
        if (d2) *d2 = FLT_MAX;
        return -1;
    }
    if (d2) *d2 = d;
    return idx;
}

void wmo_nn_batch(const wmo_kdtree *t, const float *q, int nq, int *idx, float *d2) {
    int i;
    for (i = 0; i < nq; ++i) idx[i] = wmo_kdtree_nn(t, q + 3 * i, d2 + i);
}

void wmo_nn_brute(const float *tgt, int m, const float *q, int nq, int *idx, float *d2) {
    int i, j;
    for (i = 0; i < nq; ++i) {
        float best = FLT_MAX;
        int bi = -1;
        for (j = 0; j < m; ++j) {
            float d = wmo_d2(q + 3 * i, tgt + 3 * j);
            if (d < best) { /* ascending j => lowest index wins ties */
                best = d;
                bi = j;
            }
        }
        idx[i] = bi;
        d2[i] = best;
    }
}
