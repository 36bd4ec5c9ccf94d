/*
 * oracle.c -- CPU restatement of mini-gunrock's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  Nothing in mini_amd/ or include/ may call it: the
 * product path is the HIP library (libmgx.so) and it fails loudly without it.
 *
 * What is restated (all citations are relative to /root/reference/):
 *   - MTX loader + CSR builder            gunrock/src/graph.hxx:96-223
 *   - BFS CPU validation                  gunrock/src/bfs/bfs_problem.hxx:52-72
 *   - SSSP CPU validation (preds)         gunrock/src/sssp/sssp_problem.hxx:59-88
 *   - operator semantics (serial)         gunrock/src/advance.hxx:20-160,
 *                                         gunrock/src/filter.hxx:11-31,
 *                                         gunrock/src/neighborhood.hxx:12-70
 *   - functors                            gunrock/src/bfs/bfs_functor.hxx:7-53,
 *                                         gunrock/src/sssp/sssp_functor.hxx:10-36,
 *                                         gunrock/src/pr/pr_functor.hxx:10-32
 *   - enactor loops                       gunrock/src/bfs/bfs_enactor.hxx:41-117,
 *                                         gunrock/src/sssp/sssp_enactor.hxx:40-72,
 *                                         gunrock/src/pr/pr_enactor.hxx:41-79
 *   - k-core CPU validation + enactor     gunrock/src/kcore/kcore_problem.hxx:54-105,
 *                                         gunrock/src/kcore/kcore_enactor.hxx:40-86,
 *                                         gunrock/src/kcore/kcore_functor.hxx:10-36
 *
 * The scan / load-balanced-search / compaction / segmented-reduce arithmetic of the
 * reference lives in moderngpu (https://github.com/yzhwang/moderngpu.git, .gitmodules:1-3,
 * un-vendored, no pinned SHA).  Its published semantics are restated here serially:
 * exclusive plus-scan, (segment, rank) enumeration of a CSR-like segment list, stable
 * compaction with the predicate evaluated exactly once per element, and per-segment
 * reduction with an identity for empty segments.
 *
 * PINNING: the reference's CPU validation functions sit in headers that include
 * moderngpu, which is absent, so the reference cannot be built here (no oracle/_ref).
 * This restatement is pinned against the golden vectors that SURVEY.md section 8c
 * records from the reference's own cpu() routines on the reference's own test
 * fixtures (tests/golden/reference_goldens.json) -- see tests/test_oracle_golden.py.
 * The moderngpu primitives themselves have no reference-side test: at the
 * primitive boundary parity is unpinned (end-to-end labels/preds only).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <float.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* small helpers                                                              */
/* ------------------------------------------------------------------------- */

ORC_API void orc_free(void *p) { free(p); }

typedef struct { uint64_t key; uint32_t pos; } keyed_t;

/* stable bottom-up merge sort on (key, original position) */
static void sort_keyed(keyed_t *a, size_t n)
{
    if (n < 2) return;
    keyed_t *b = (keyed_t *)malloc(n * sizeof(keyed_t));
    keyed_t *src = a, *dst = b;
    for (size_t width = 1; width < n; width <<= 1) {
        for (size_t lo = 0; lo < n; lo += 2 * width) {
            size_t mid = lo + width < n ? lo + width : n;
            size_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            size_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = (src[j].key < src[i].key) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        keyed_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(keyed_t));
    free(b);
}

/* ------------------------------------------------------------------------- */
/* CSR construction -- graph.hxx:129-172                                      */
/*                                                                            */
/* A tuple (t0, t1, w) is one MTX line "t0 t1 w" after the 1-based -> 0-based  */
/* shift.  The reference sorts by (t1, t0) and indexes row_offsets by t1, so   */
/* the CSR row is the SECOND field and the neighbour the FIRST.    */
/* With undir the swapped copy of every tuple is appended first                */
/* (graph.hxx:130-137); multi-edges and self loops are kept.                   */
/* Equal (t1,t0) keys: the reference's std::sort is unstable (and its          */
/* comparator is not a strict weak order, graph.hxx:156); this restatement     */
/* keeps tuple order among equal keys, which only fixes the order of the       */
/* weights of exact duplicate edges.                                           */
/* ------------------------------------------------------------------------- */
ORC_API int orc_csr_from_tuples(int num_vertices, int64_t num_tuples,
                                const int *t0, const int *t1, const float *w,
                                int undir,
                                int64_t *out_m, int **out_off, int **out_idx,
                                float **out_w, int **out_src)
{
    int64_t m = undir ? 2 * num_tuples : num_tuples;
    if (m > INT_MAX) return -2;
    keyed_t *k = (keyed_t *)malloc((size_t)(m ? m : 1) * sizeof(keyed_t));
    int *a0 = (int *)malloc((size_t)(m ? m : 1) * sizeof(int));
    int *a1 = (int *)malloc((size_t)(m ? m : 1) * sizeof(int));
    float *aw = (float *)malloc((size_t)(m ? m : 1) * sizeof(float));
    for (int64_t e = 0; e < num_tuples; ++e) {
        a0[e] = t0[e]; a1[e] = t1[e]; aw[e] = w ? w[e] : 1.0f;
        if (undir) { a0[e + num_tuples] = t1[e]; a1[e + num_tuples] = t0[e]; aw[e + num_tuples] = aw[e]; }
    }
    for (int64_t e = 0; e < m; ++e) {
        k[e].key = ((uint64_t)(uint32_t)a1[e] << 32) | (uint32_t)a0[e];
        k[e].pos = (uint32_t)e;
    }
    sort_keyed(k, (size_t)m);

    int *off = (int *)malloc((size_t)(num_vertices + 1) * sizeof(int));
    int *idx = (int *)malloc((size_t)(m ? m : 1) * sizeof(int));
    int *src = (int *)malloc((size_t)(m ? m : 1) * sizeof(int));
    float *wt = (float *)malloc((size_t)(m ? m : 1) * sizeof(float));
    for (int v = 0; v <= num_vertices; ++v) off[v] = (int)m;   /* graph.hxx:160 */
    int cur = -1;
    for (int64_t e = 0; e < m; ++e) {
        uint32_t p = k[e].pos;
        while (cur < a1[p]) off[++cur] = (int)e;               /* graph.hxx:166-168 */
        src[e] = cur;
        idx[e] = a0[p];
        wt[e] = aw[p];
    }
    free(k); free(a0); free(a1); free(aw);
    *out_m = m; *out_off = off; *out_idx = idx; *out_w = wt; *out_src = src;
    return 0;
}

/* MTX text loader -- graph.hxx:96-137.  First non-'%' line is "rows cols nnz";  */
/* each following line "a b [w]"; missing weight -> 1.0f or rand()%64.            */
/* Returns 0, -1 (cannot open; reference returns nullptr) or -3 (parse error;     */
/* reference prints and exit(0)s).                                                 */
ORC_API int orc_load_mtx(const char *path, int undir, int random_w,
                         int *out_n, int64_t *out_m, int **out_off, int **out_idx,
                         float **out_w, int **out_src)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[100];
    int got_header = 0;
    while (fgets(line, 100, f)) { if (line[0] != '%') { got_header = 1; break; } }
    int h, wd, nnz;
    if (!got_header || sscanf(line, "%d %d %d", &h, &wd, &nnz) != 3) { fclose(f); return -3; }
    int *t0 = (int *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int));
    int *t1 = (int *)malloc((size_t)(nnz ? nnz : 1) * sizeof(int));
    float *w = (float *)malloc((size_t)(nnz ? nnz : 1) * sizeof(float));
    for (int e = 0; e < nnz; ++e) {
        int a, b; float x; int items;
        if (!fgets(line, 100, f) || (items = sscanf(line, "%d %d %f", &a, &b, &x)) < 2) {
            free(t0); free(t1); free(w); fclose(f); return -3;
        }
        if (items == 2) x = random_w ? (float)(rand() % 64) : 1.0f;   /* graph.hxx:125-127 */
        t0[e] = a - 1; t1[e] = b - 1; w[e] = x;
    }
    fclose(f);
    int rc = orc_csr_from_tuples(h, nnz, t0, t1, w, undir, out_m, out_off, out_idx, out_w, out_src);
    free(t0); free(t1); free(w);
    *out_n = h;
    return rc;
}

/* ------------------------------------------------------------------------- */
/* BFS CPU validation -- bfs_problem.hxx:52-72                                 */
/* labels must come in as -1 everywhere (test_bfs.cu:44).                      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_bfs_cpu(int n, const int *row_offsets, const int *col_indices,
                         int src, int *labels)
{
    /* FIFO that may hold a vertex more than once (re-label rule below) */
    size_t cap = (size_t)n + 16, head = 0, tail = 0;
    int *q = (int *)malloc(cap * sizeof(int));
    q[tail++] = src;
    labels[src] = 0;
    while (head < tail) {
        int s = q[head++];
        for (int e = row_offsets[s]; e < row_offsets[s + 1]; ++e) {
            int d = col_indices[e];
            if (labels[d] < 0 || labels[s] + 1 < labels[d]) {      /* bfs_problem.hxx:65 */
                labels[d] = labels[s] + 1;
                if (tail == cap) { cap *= 2; q = (int *)realloc(q, cap * sizeof(int)); }
                q[tail++] = d;
            }
        }
    }
    free(q);
}

/* ------------------------------------------------------------------------- */
/* SSSP CPU validation -- sssp_problem.hxx:59-88                               */
/* Label-correcting with a min-heap of (key, vertex) pairs.  Quirks kept:      */
/*   - distances are int, weights truncated to int (:78)                       */
/*   - the key pushed for v is dist[u], not dist[v] (:82); first key is -1     */
/*   - no "already settled" check; output is preds (last improver wins)        */
/* preds must come in as -1 everywhere (test_sssp.cu:44).                      */
/* ------------------------------------------------------------------------- */
typedef struct { int key; int v; } hp_t;
static int hp_less(hp_t a, hp_t b) { return a.key < b.key || (a.key == b.key && a.v < b.v); }

ORC_API void orc_sssp_cpu(int n, const int *row_offsets, const int *col_indices,
                          const float *col_values, int src, int *preds, int *dist_out)
{
    size_t cap = 1024, sz = 0;
    hp_t *h = (hp_t *)malloc(cap * sizeof(hp_t));
    int *dist = (int *)malloc((size_t)(n + 1) * sizeof(int));
    for (int i = 0; i <= n; ++i) dist[i] = INT_MAX;
    h[sz++] = (hp_t){-1, src};
    preds[src] = -1;
    dist[src] = 0;
    while (sz) {
        int u = h[0].v;
        /* pop min */
        hp_t last = h[--sz];
        size_t i = 0;
        for (;;) {
            size_t l = 2 * i + 1, r = l + 1, c;
            if (l >= sz) break;
            c = (r < sz && hp_less(h[r], h[l])) ? r : l;
            if (!hp_less(h[c], last)) break;
            h[i] = h[c]; i = c;
        }
        if (sz) h[i] = last;
        for (int e = row_offsets[u]; e < row_offsets[u + 1]; ++e) {
            int v = col_indices[e];
            int w = (int)col_values[e];                              /* :78 truncation */
            if (dist[v] > dist[u] + w) {
                preds[v] = u;
                dist[v] = dist[u] + w;
                if (sz == cap) { cap *= 2; h = (hp_t *)realloc(h, cap * sizeof(hp_t)); }
                hp_t x = {dist[u], v};                               /* :82 key = dist[u] */
                size_t j = sz++;
                while (j) { size_t p = (j - 1) / 2; if (!hp_less(x, h[p])) break; h[j] = h[p]; j = p; }
                h[j] = x;
            }
        }
    }
    if (dist_out) memcpy(dist_out, dist, (size_t)n * sizeof(int));
    free(dist); free(h);
}

/* ------------------------------------------------------------------------- */
/* moderngpu primitive semantics, serial                                       */
/* ------------------------------------------------------------------------- */

/* K1/K8/K10: scanned[i] = sum_{j<i} deg(in[j]); returns the total (advance.hxx:32-43) */
ORC_API int64_t orc_scan_degrees(const int *offsets, const int *in, int64_t nin, int *scanned)
{
    int64_t run = 0;
    for (int64_t i = 0; i < nin; ++i) {
        scanned[i] = (int)run;
        int v = in[i];
        run += offsets[v + 1] - offsets[v];
    }
    return run;
}

/* transform_lbs enumeration: for every work item idx in [0,total) the (seg,rank)  */
/* pair such that scanned[seg] <= idx < scanned[seg+1], rank = idx - scanned[seg]. */
/* Empty segments own no work item.                                                 */
ORC_API void orc_lbs(const int *scanned, int64_t nseg, int64_t total, int *seg_out, int *rank_out)
{
    int64_t idx = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        int64_t end = (s + 1 < nseg) ? scanned[s + 1] : total;
        for (int64_t r = 0; idx < end; ++idx, ++r) { seg_out[idx] = (int)s; rank_out[idx] = (int)r; }
    }
}

/* ------------------------------------------------------------------------- */
/* BFS operators with bfs_functor_t inlined (bfs_functor.hxx:7-53)             */
/* ------------------------------------------------------------------------- */

/* advance_forward_kernel<bfs, bfs_functor, idempotence=false, has_output=true>    */
/* (advance.hxx:20-67).  Serial order = output slot order, so the first edge that   */
/* reaches an unvisited dst wins the CAS.  Returns front = sum of degrees.          */
ORC_API int64_t orc_bfs_advance(const int *row_offsets, const int *col_indices,
                                int *labels, const int *in, int64_t nin, int iteration,
                                int *out /* capacity >= front */)
{
    int64_t idx = 0;
    for (int64_t s = 0; s < nin; ++s) {
        int v = in[s];
        for (int e = row_offsets[v]; e < row_offsets[v + 1]; ++e, ++idx) {
            int nb = col_indices[e];
            int cond = (labels[nb] == -1);                     /* cond_advance  */
            int app = 0;                                       /* apply_advance */
            if (labels[nb] == -1) { labels[nb] = iteration + 1; app = 1; }
            out[idx] = (cond && app) ? nb : -1;
        }
    }
    return idx;
}

/* filter_kernel<bfs, bfs_functor> (filter.hxx:11-31; cond_filter: idx != -1) */
ORC_API int64_t orc_bfs_filter(const int *in, int64_t nin, int *out)
{
    int64_t k = 0;
    for (int64_t i = 0; i < nin; ++i) if (in[i] != -1) out[k++] = in[i];
    return k;
}

/* gen_unvisited_kernel (advance.hxx:86-106; cond_gen_unvisited: label == -1) */
ORC_API int64_t orc_bfs_gen_unvisited(const int *labels, const int *indices, int64_t nin, int *out)
{
    int64_t k = 0;
    for (int64_t i = 0; i < nin; ++i) if (labels[indices[i]] == -1) out[k++] = indices[i];
    return k;
}

/* sparse_to_dense_kernel (advance.hxx:69-84; cond_sparse_to_dense: label == iteration) */
ORC_API void orc_bfs_sparse_to_dense(const int *labels, const int *sparse, int64_t ns,
                                     int *dense, int iteration)
{
    for (int64_t i = 0; i < ns; ++i) { int v = sparse[i]; dense[v] = (labels[v] == iteration) ? 1 : 0; }
}

/* advance_backward_kernel (advance.hxx:108-160).  Every in-edge of every unvisited  */
/* vertex is inspected (no early exit).  The claim is the functor's CAS              */
/* labels[v]: -1 -> iteration+1, so only the first frontier in-neighbour of v acts.  */
/* Returns the number of inspected in-edges (the operator's "front").                */
ORC_API int64_t orc_bfs_advance_backward(const int *col_offsets, const int *row_indices,
                                         int *labels, int *unvisited, int64_t nu,
                                         const int *bitmap, int *bitmap_out, int iteration)
{
    int64_t inspected = 0;
    for (int64_t s = 0; s < nu; ++s) {
        const int v = unvisited[s];
        for (int e = col_offsets[v]; e < col_offsets[v + 1]; ++e, ++inspected) {
            int u = row_indices[e];
            if (bitmap[u] && labels[v] == -1) {
                labels[v] = iteration + 1;
                bitmap_out[v] = 1;
                unvisited[s] = -1;
            }
        }
    }
    return inspected;
}

/* bfs_enactor_t::enact_pushpull (bfs_enactor.hxx:41-117).                           */
/* stats[0]=pushed iterations, stats[1]=total iterations (== stats[0] when the pull  */
/* phase never ran), stats[2]=edges expanded by push, stats[3]=in-edges inspected by */
/* pull.  Pass the CSR again as (col_offsets,row_indices) for the reference's        */
/* behaviour (its CSC is always a copy of the CSR).                       */
/* Returns 0, or -4 where the reference would exit on frontier overflow              */
/* (frontier.hxx:53-59: the n-int bitmap is loaded into an m-capacity buffer).       */
typedef struct { int *d; int64_t size; int64_t cap; } ofr_t;

ORC_API int orc_bfs_enact_pushpull(int n, int64_t m, const int *row_offsets, const int *col_indices,
                                   const int *col_offsets, const int *row_indices,
                                   int src, float threshold, int *labels, int64_t *stats)
{
    int64_t cap = m < 1 ? 1 : m;                       /* enactor.hxx:22-23, queue_sizing 1 */
    ofr_t buf[2], unv[2];
    for (int i = 0; i < 2; ++i) {
        buf[i].d = (int *)malloc((size_t)(cap > n ? cap : n) * sizeof(int)); buf[i].size = 0; buf[i].cap = cap;
        unv[i].d = (int *)malloc((size_t)(n ? n : 1) * sizeof(int)); unv[i].size = n; unv[i].cap = n;
        for (int v = 0; v < n; ++v) unv[i].d[v] = v;   /* enactor.hxx:29-34 */
    }
    for (int v = 0; v < n; ++v) labels[v] = -1;
    labels[src] = 0;                                   /* bfs_problem.hxx:38-40 */
    buf[0].d[0] = src; buf[0].size = 1;                /* init_frontier :34-38  */
    stats[0] = stats[1] = stats[2] = stats[3] = 0;

    int rc = 0;
    int64_t frontier_length = 1;
    int sel = 0, iteration;
    int64_t num_unvisited = (int64_t)n - 1;
    for (iteration = 0;; ++iteration) {
        frontier_length = orc_bfs_advance(row_offsets, col_indices, labels,
                                          buf[sel].d, buf[sel].size, iteration, buf[sel ^ 1].d);
        buf[sel ^ 1].size = frontier_length;
        stats[2] += frontier_length;
        sel ^= 1;
        if (!frontier_length) break;
        frontier_length = orc_bfs_filter(buf[sel].d, buf[sel].size, buf[sel ^ 1].d);
        buf[sel ^ 1].size = frontier_length;
        num_unvisited -= frontier_length;
        if ((float)num_unvisited < (float)frontier_length * threshold) break;
        if (!frontier_length) break;
        sel ^= 1;
    }
    stats[0] = stats[1] = iteration;

    if (frontier_length) {                              /* :74-112 pull phase */
        ++iteration;
        frontier_length = orc_bfs_gen_unvisited(labels, unv[sel ^ 1].d, unv[sel ^ 1].size, unv[sel].d);
        unv[sel].size = frontier_length;
        if (n > buf[sel].cap) { rc = -4; goto done; }
        memset(buf[sel].d, 0, (size_t)n * sizeof(int)); buf[sel].size = n;   /* load(bitmap_array) */
        orc_bfs_sparse_to_dense(labels, buf[sel ^ 1].d, buf[sel ^ 1].size, buf[sel].d, iteration);
        for (;; ++iteration) {
            memset(buf[sel ^ 1].d, 0, (size_t)n * sizeof(int)); buf[sel ^ 1].size = n;
            stats[3] += orc_bfs_advance_backward(col_offsets, row_indices, labels,
                                                 unv[sel].d, unv[sel].size,
                                                 buf[sel].d, buf[sel ^ 1].d, iteration);
            int64_t new_len = orc_bfs_filter(unv[sel].d, unv[sel].size, unv[sel ^ 1].d);
            unv[sel ^ 1].size = new_len;
            if (!new_len || new_len == frontier_length) break;
            frontier_length = new_len;
            sel ^= 1;
        }
        stats[1] = iteration;
    }
done:
    for (int i = 0; i < 2; ++i) { free(buf[i].d); free(unv[i].d); }
    return rc;
}

/* ------------------------------------------------------------------------- */
/* SSSP operators with sssp_functor_t inlined (sssp_functor.hxx:10-36)         */
/* ------------------------------------------------------------------------- */

/* advance_forward_kernel<sssp, sssp_functor, false, true>.                          */
/* cond_advance: nd = dist[src] + w[e]; old = atomicMin(dist+dst, nd); nd < old      */
/* apply_advance: preds[dst] = src (ALWAYS); returns true                 */
ORC_API int64_t orc_sssp_advance(const int *row_offsets, const int *col_indices, const float *weights,
                                 float *dist, int *preds, const int *in, int64_t nin, int *out)
{
    int64_t idx = 0;
    for (int64_t s = 0; s < nin; ++s) {
        int v = in[s];
        for (int e = row_offsets[v]; e < row_offsets[v + 1]; ++e, ++idx) {
            int nb = col_indices[e];
            float nd = dist[v] + weights[e];
            float old = dist[nb];
            dist[nb] = fminf(nd, old);                 /* intrinsics.hxx:12-22 */
            int cond = nd < old;
            preds[nb] = v;
            out[idx] = cond ? nb : -1;
        }
    }
    return idx;
}

/* filter_kernel<sssp, sssp_functor>: drop -1, drop if visited[v]==it, else stamp (:12-18) */
ORC_API int64_t orc_sssp_filter(const int *in, int64_t nin, int *visited, int iteration, int *out)
{
    int64_t k = 0;
    for (int64_t i = 0; i < nin; ++i) {
        int v = in[i];
        if (v == -1) continue;
        if (visited[v] == iteration) continue;
        visited[v] = iteration;
        out[k++] = v;
    }
    return k;
}

/* sssp_enactor_t::enact (sssp_enactor.hxx:40-72).  dist/preds/visited are outputs.   */
/* stats[0]=iterations run, stats[1]=edge relaxations (sum of front),                 */
/* stats[2]=sum of input frontier lengths.  Returns -4 on frontier overflow.          */
ORC_API int orc_sssp_enact(int n, int64_t m, const int *row_offsets, const int *col_indices,
                           const float *weights, int src, float queue_sizing,
                           float *dist, int *preds, int64_t *stats)
{
    int64_t cap = (int64_t)(int)((float)m * queue_sizing);      /* enactor.hxx:22 float math */
    if (cap < 1) cap = 1;
    int *buf[2] = { (int *)malloc((size_t)cap * sizeof(int)), (int *)malloc((size_t)cap * sizeof(int)) };
    int *visited = (int *)malloc((size_t)(n ? n : 1) * sizeof(int));
    for (int v = 0; v < n; ++v) { dist[v] = FLT_MAX; preds[v] = -1; visited[v] = -1; }
    dist[src] = 0.0f;                                            /* sssp_problem.hxx:44-49 */
    buf[0][0] = src;
    int64_t len = 1;
    int sel = 0, it, rc = 0;
    stats[0] = stats[1] = stats[2] = 0;
    for (it = 0;; ++it) {
        /* capacity check the reference does in frontier_t::resize (frontier.hxx:83-93) */
        int64_t need = 0;
        for (int64_t i = 0; i < len; ++i) need += row_offsets[buf[sel][i] + 1] - row_offsets[buf[sel][i]];
        if (need > cap) { rc = -4; break; }
        stats[2] += len;
        int64_t front = orc_sssp_advance(row_offsets, col_indices, weights, dist, preds, buf[sel], len, buf[sel ^ 1]);
        stats[1] += front;
        sel ^= 1;
        len = orc_sssp_filter(buf[sel], front, visited, it, buf[sel ^ 1]);
        stats[0] = it + 1;
        if (!len) break;
        sel ^= 1;
    }
    free(buf[0]); free(buf[1]); free(visited);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* neighbourhood reduce (neighborhood.hxx:12-70) + PR (pr_functor.hxx, pr_enactor.hxx) */
/* ------------------------------------------------------------------------- */

/* generic serial segmented reduce over the LBS enumeration: reduced[seg] = op over the   */
/* values of the segment's edges, identity for empty segments.  op: 0 plus(f32),           */
/* 1 min(i32), 2 max(i32).  values[] are per-VERTEX; the functor reads values[neighbor].   */
ORC_API int64_t orc_neighbor_reduce_f32_plus(const int *offsets, const int *indices,
                                             const int *in, int64_t nin,
                                             const float *vertex_value, float identity, float *reduced)
{
    int64_t nz = 0;
    for (int64_t s = 0; s < nin; ++s) {
        int v = in[s];
        float acc = identity;
        int first = 1;
        for (int e = offsets[v]; e < offsets[v + 1]; ++e, ++nz) {
            float x = vertex_value[indices[e]];
            acc = first ? x : acc + x;     /* mgpu segreduce folds the segment's values, no identity mixed in */
            first = 0;
        }
        reduced[s] = acc;
    }
    return nz;
}

ORC_API int64_t orc_neighbor_reduce_i32(const int *offsets, const int *indices,
                                        const int *in, int64_t nin,
                                        const int *vertex_value, int identity, int is_max, int *reduced)
{
    int64_t nz = 0;
    for (int64_t s = 0; s < nin; ++s) {
        int v = in[s];
        int acc = identity, first = 1;
        for (int e = offsets[v]; e < offsets[v + 1]; ++e, ++nz) {
            int x = vertex_value[indices[e]];
            if (first) acc = x; else acc = is_max ? (x > acc ? x : acc) : (x < acc ? x : acc);
            first = 0;
        }
        reduced[s] = acc;
    }
    return nz;
}

/* pr_enactor_t::enact (pr_enactor.hxx:41-79) with pr_functor_t (pr_functor.hxx:10-32).      */
/* Quirk kept: reduced_ranks is written per FRONTIER POSITION                   */
/* (neighborhood.hxx:58) but cond_filter reads it per VERTEX ID (pr_functor.hxx:13), so       */
/* only iteration 0 (frontier == iota) is a true PageRank step.  The reference returns 0      */
/* from the operator without touching reduced[] when the frontier has no edges.               */
/* ranks[n] out; frontier_len_out[max_iter] records the filter output length per iteration.   */
ORC_API int orc_pr_enact(int n, const int *offsets, const int *indices, int max_iter,
                         float *ranks, int64_t *frontier_len_out)
{
    float *reduced = (float *)calloc((size_t)(n ? n : 1), sizeof(float));
    float *degrees = (float *)malloc((size_t)(n ? n : 1) * sizeof(float));
    float *gathered = (float *)malloc((size_t)(n ? n : 1) * sizeof(float));
    int *buf[2] = { (int *)malloc((size_t)(n ? n : 1) * sizeof(int)), (int *)malloc((size_t)(n ? n : 1) * sizeof(int)) };
    for (int v = 0; v < n; ++v) {
        ranks[v] = 0.15f; degrees[v] = (float)(offsets[v + 1] - offsets[v]); buf[0][v] = v;
    }
    int64_t len = n;
    int sel = 0, it = 0;
    while (len > 0 && it < max_iter) {
        /* get_value_to_reduce: isfinite(rank) ? rank : 0 */
        for (int v = 0; v < n; ++v) gathered[v] = isfinite(ranks[v]) ? ranks[v] : 0.0f;
        int64_t total = 0;
        for (int64_t s = 0; s < len; ++s) total += offsets[buf[sel][s] + 1] - offsets[buf[sel][s]];
        if (total) orc_neighbor_reduce_f32_plus(offsets, indices, buf[sel], len, gathered, 0.0f, reduced);
        /* filter with cond_filter (side effect: writes the new rank) */
        int64_t k = 0;
        for (int64_t s = 0; s < len; ++s) {
            int v = buf[sel][s];
            float old_value = ranks[v];
            float new_value = (degrees[v] > 0) ? (0.15f + 0.85f * reduced[v] / degrees[v]) : 0.15f;
            if (!isfinite(new_value)) new_value = 0;
            ranks[v] = new_value;
            if (fabsf(new_value - old_value) > (0.001f * old_value)) buf[sel ^ 1][k++] = v;
        }
        len = k;
        if (frontier_len_out) frontier_len_out[it] = len;
        ++it;
        sel ^= 1;
    }
    free(reduced); free(degrees); free(gathered); free(buf[0]); free(buf[1]);
    return it;
}

/* ------------------------------------------------------------------------- */
/* k-core decomposition: the reference's CPU validator and a      */
/* serial restatement of its enactor loop over the operators                    */
/* ------------------------------------------------------------------------- */

/* kcore_problem_t::cpu (kcore_problem.hxx:54-105).  degrees start as the CSR row      */
/* lengths (kcore_problem.hxx:43-45 via problem.hxx:23-30: multi-edges and self-loops  */
/* count), num_cores as zeros (test_kcore.cu:36).  For k = 1, 2, ...: repeat { every    */
/* vertex with 0 < degree < k that is still marked to_remain gets core k-1, degree 0;   */
/* to_remain = degree >= k; the removed vertices' neighbours lose one degree per entry  */
/* (degrees go negative: a removed vertex keeps being decremented) } until nothing was  */
/* removed; the first k that leaves nobody with degree >= k ends the run.  Returns      */
/* largest_k_core (-1 if no k <= n ended it).  Vertices without entries keep core 0.    */
ORC_API int orc_kcore_cpu(int n, const int *row_offsets, const int *col_indices, int *num_cores)
{
    int *degrees = (int *)malloc((size_t)(n ? n : 1) * sizeof(int));
    unsigned char *to_remove = (unsigned char *)malloc((size_t)(n ? n : 1));
    unsigned char *to_remain = (unsigned char *)malloc((size_t)(n ? n : 1));
    int largest = -1;
    for (int v = 0; v < n; ++v) { degrees[v] = row_offsets[v + 1] - row_offsets[v]; num_cores[v] = 0; }
    for (int k = 1; k <= n; ++k) {
        int num_to_remain = 0;
        memset(to_remove, 0, (size_t)n);
        memset(to_remain, 1, (size_t)n);
        for (;;) {
            int num_to_remove = 0;
            for (int v = 0; v < n; ++v) {
                if (degrees[v] < k && degrees[v] > 0 && to_remain[v]) {
                    num_cores[v] = k - 1; degrees[v] = 0; to_remove[v] = 1; ++num_to_remove;
                } else to_remove[v] = 0;
            }
            num_to_remain = 0;
            for (int v = 0; v < n; ++v) {
                to_remain[v] = degrees[v] >= k;
                num_to_remain += to_remain[v];
            }
            if (!num_to_remove) break;
            for (int v = 0; v < n; ++v) {
                if (!to_remove[v]) continue;
                for (int e = row_offsets[v]; e < row_offsets[v + 1]; ++e) degrees[col_indices[e]]--;
                to_remove[v] = 0;
            }
        }
        if (num_to_remain == 0) { largest = k - 1; break; }
    }
    free(degrees); free(to_remove); free(to_remain);
    return largest;
}

/* kcore_enactor_t::enact (kcore_enactor.hxx:40-86) over the serial operators, functors */
/* inlined (kcore_functor.hxx:10-36):                                                   */
/*   filter<deg_less_than_k>  keeps v with 0 < degree < k, side effect core = k-1,      */
/*                            degree = 0                                    (:11-19)    */
/*   advance<update_deg, idempotent=false, has_output=false>: one atomicAdd(-1) on the  */
/*                            neighbour's degree per expanded entry, no output, returns */
/*                            0 (advance.hxx:66)                            (:28-35)    */
/*   filter<deg_atleast_k>    counts degree >= k                            (:22-26)    */
/* `selector` never flips and the advance writes no output, so buffers[0] is the iota   */
/* of init_frontier in every pass: each filter looks at ALL vertices.  Quirk kept:      */
/* frontier_length is only refreshed in a pass that removed something, so a k whose     */
/* first pass removes nothing is judged by the previous k's count (n before any) -- on  */
/* a graph without entries the loop runs to k = n and largest_k_core stays -1, where    */
/* cpu() answers 0.  stats[0] = k values tried, [1] = passes, [2] = entries expanded,   */
/* [3] = vertices removed.                                                              */
ORC_API int orc_kcore_enact(int n, const int *row_offsets, const int *col_indices, int *num_cores, int64_t *stats)
{
    int *degrees = (int *)malloc((size_t)(n ? n : 1) * sizeof(int));
    int *removed = (int *)malloc((size_t)(n ? n : 1) * sizeof(int));
    int largest = -1;
    int64_t frontier_length = n, st[4] = {0, 0, 0, 0};
    for (int v = 0; v < n; ++v) { degrees[v] = row_offsets[v + 1] - row_offsets[v]; num_cores[v] = 0; }
    for (int k = 1; k <= n; ++k) {
        ++st[0];
        for (;;) {
            int64_t num_to_remove = 0;
            ++st[1];
            for (int v = 0; v < n; ++v)
                if (degrees[v] < k && degrees[v] > 0) { num_cores[v] = k - 1; degrees[v] = 0; removed[num_to_remove++] = v; }
            if (!num_to_remove) break;
            st[3] += num_to_remove;
            for (int64_t i = 0; i < num_to_remove; ++i)
                for (int e = row_offsets[removed[i]]; e < row_offsets[removed[i] + 1]; ++e, ++st[2]) degrees[col_indices[e]]--;
            frontier_length = 0;
            for (int v = 0; v < n; ++v) frontier_length += degrees[v] >= k;
        }
        if (frontier_length == 0) { largest = k - 1; break; }
    }
    if (stats) memcpy(stats, st, sizeof(st));
    free(degrees); free(removed);
    return largest;
}

/* ------------------------------------------------------------------------- */
/* Synthetic R-MAT input (SURVEY 8d configs 2,3,5).  Not reference code: the    */
/* reference ships no generator (F4).  This is the SPEC the HIP generator in    */
/* mini_amd/csrc must reproduce bit-exactly.                                    */
/*                                                                              */
/*  edge e in [0, ef*2^S):  x = mix64(seed*K0 + e);                             */
/*    for level l: r = mix64(x + (l+1)*K1); u = r >> 32;                        */
/*      u <  A   -> (0,0);  u < AB -> (0,1);  u < ABC -> (1,0);  else (1,1)     */
/*      (a,b,c,d) = (0.57,0.19,0.19,0.05) as 32-bit fixed-point thresholds      */
/*    src/dst bits are appended MSB first, then both ids go through the same     */
/*    bijective scramble on [0,2^S).                                             */
/*  weight(e) = float(mix64(seed*K0 + e + K2) % 64)  (same for both directions)  */
/* ------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define RMAT_K0 0xD1B54A32D192ED03ull
#define RMAT_K1 0x8CB92BA72F3D8DD7ull
#define RMAT_K2 0xA24BAED4963EE407ull
#define RMAT_A   2448131358u   /* floor(0.57 * 2^32) */
#define RMAT_AB  3264175144u   /* floor(0.76 * 2^32) */
#define RMAT_ABC 4080218931u   /* floor(0.95 * 2^32) */

static inline uint32_t bitrev(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1u); x >>= 1; }
    return r;
}

ORC_API uint32_t orc_rmat_scramble(uint32_t v, int scale)
{
    uint32_t mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
    v = (v * 0x9E3779B1u + 0x7F4A7C15u) & mask;
    v = bitrev(v, scale);
    v = (v * 0x85EBCA6Bu + 0xC2B2AE35u) & mask;
    return v;
}

ORC_API void orc_rmat_edges(int scale, int64_t first_edge, int64_t num_edges, uint64_t seed,
                            int scramble, int *src_out, int *dst_out, float *w_out)
{
    for (int64_t i = 0; i < num_edges; ++i) {
        uint64_t e = (uint64_t)(first_edge + i);
        uint64_t x = mix64(seed * RMAT_K0 + e);
        uint32_t s = 0, d = 0;
        for (int l = 0; l < scale; ++l) {
            uint32_t u = (uint32_t)(mix64(x + (uint64_t)(l + 1) * RMAT_K1) >> 32);
            uint32_t sb = (u >= RMAT_AB), db;
            if (u < RMAT_A) db = 0; else if (u < RMAT_AB) db = 1; else if (u < RMAT_ABC) db = 0; else db = 1;
            s = (s << 1) | sb; d = (d << 1) | db;
        }
        if (scramble) { s = orc_rmat_scramble(s, scale); d = orc_rmat_scramble(d, scale); }
        src_out[i] = (int)s; dst_out[i] = (int)d;
        if (w_out) w_out[i] = (float)(mix64(seed * RMAT_K0 + e + RMAT_K2) % 64ull);
    }
}

/* ------------------------------------------------------------------------- */
/* Float-distance fixed point of the GPU SSSP semantics, computed independently */
/* (binary-heap Dijkstra on float32 sums).  Valid because w >= 0 makes           */
/* d -> fl(d + w) monotone and non-decreasing; used to cross-check               */
/* orc_sssp_enact's fixed point (SURVEY 8a "SSSP distances are deterministic").  */
/* ------------------------------------------------------------------------- */
typedef struct { float key; int v; } fhp_t;
ORC_API void orc_sssp_dijkstra_f32(int n, const int *row_offsets, const int *col_indices,
                                   const float *weights, int src, float *dist)
{
    size_t cap = 1024, sz = 0;
    fhp_t *h = (fhp_t *)malloc(cap * sizeof(fhp_t));
    for (int v = 0; v < n; ++v) dist[v] = FLT_MAX;
    dist[src] = 0.0f;
    h[sz++] = (fhp_t){0.0f, src};
    while (sz) {
        fhp_t top = h[0];
        fhp_t last = h[--sz];
        size_t i = 0;
        for (;;) {
            size_t l = 2 * i + 1, r = l + 1, c;
            if (l >= sz) break;
            c = (r < sz && h[r].key < h[l].key) ? r : l;
            if (!(h[c].key < last.key)) break;
            h[i] = h[c]; i = c;
        }
        if (sz) h[i] = last;
        if (top.key > dist[top.v]) continue;
        int u = top.v;
        for (int e = row_offsets[u]; e < row_offsets[u + 1]; ++e) {
            int v = col_indices[e];
            float nd = dist[u] + weights[e];
            if (nd < dist[v]) {
                dist[v] = nd;
                if (sz == cap) { cap *= 2; h = (fhp_t *)realloc(h, cap * sizeof(fhp_t)); }
                fhp_t x = {nd, v};
                size_t j = sz++;
                while (j) { size_t p = (j - 1) / 2; if (!(x.key < h[p].key)) break; h[j] = h[p]; j = p; }
                h[j] = x;
            }
        }
    }
    free(h);
}
