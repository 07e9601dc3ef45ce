/*
 * ecgb_oracle_fasttrain.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * An incremental (linked-list + occurrence-index + lazy max-heap) BPE trainer that returns
 * EXACTLY what the literal restatement `ecgb_oracle_bpe_train` (ecgb_oracle.c, following
 * /root/reference/ecg_byte/rust_bpe/src/lib.rs:58-125) returns -- same pairs in the same
 * order, same final ids -- under the same DEFINED tie-break (max count, then numerically
 * smallest (left,right)).  It exists because the literal loop is O(N * num_merges) and the
 * SURVEY.md §8d tokenizers (2 000 ECGs x 60 000 symbols, 4 000 merges) would take hours
 * with it.  tests/test_oracle.py checks the two against each other on seeded inputs
 * that are rich in ties and same-symbol runs.
 *
 * Invariant kept after every merge: count[(l,r)] == number of adjacent windows (l,r) in the
 * current sequence (overlapping windows included, as lib.rs:28-48 get_stats counts them).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define END 0xFFFFFFFFu
#define DEAD 0xFFFFFFFFu

typedef struct {
    uint64_t key; /* UINT64_MAX = empty */
    int64_t count;
    uint32_t *occ; /* positions of the LEFT element; may hold stale entries */
    uint32_t occ_len, occ_cap;
    uint32_t stamp; /* merge step in which the count last increased (for heap pushes) */
} pentry;

typedef struct { pentry *e; size_t cap, len; } pmap;
typedef struct { int64_t count; uint64_t key; } hent;
typedef struct { hent *a; size_t len, cap; } heap;

static uint64_t mix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

static int pmap_init(pmap *m, size_t cap)
{
    m->e = (pentry *)malloc(cap * sizeof(pentry));
    if (!m->e) return -1;
    m->cap = cap; m->len = 0;
    for (size_t i = 0; i < cap; ++i) m->e[i].key = UINT64_MAX;
    return 0;
}

static pentry *pmap_find(pmap *m, uint64_t key)
{
    size_t h = (size_t)mix64(key) & (m->cap - 1);
    while (m->e[h].key != UINT64_MAX) {
        if (m->e[h].key == key) return &m->e[h];
        h = (h + 1) & (m->cap - 1);
    }
    return NULL;
}

static pentry *pmap_get_or_add(pmap *m, uint64_t key)
{
    if ((m->len + 1) * 2 > m->cap) {
        pmap n;
        if (pmap_init(&n, m->cap * 2)) return NULL;
        for (size_t i = 0; i < m->cap; ++i) {
            if (m->e[i].key == UINT64_MAX) continue;
            size_t h = (size_t)mix64(m->e[i].key) & (n.cap - 1);
            while (n.e[h].key != UINT64_MAX) h = (h + 1) & (n.cap - 1);
            n.e[h] = m->e[i];
        }
        n.len = m->len;
        free(m->e);
        *m = n;
    }
    size_t h = (size_t)mix64(key) & (m->cap - 1);
    while (m->e[h].key != UINT64_MAX) {
        if (m->e[h].key == key) return &m->e[h];
        h = (h + 1) & (m->cap - 1);
    }
    pentry *p = &m->e[h];
    p->key = key; p->count = 0; p->occ = NULL; p->occ_len = p->occ_cap = 0; p->stamp = 0;
    m->len++;
    return p;
}

/* max-heap: larger count first, then smaller key */
static int hless(const hent *a, const hent *b)
{
    if (a->count != b->count) return a->count < b->count;
    return a->key > b->key;
}

static int heap_push(heap *h, int64_t count, uint64_t key)
{
    if (h->len == h->cap) {
        size_t nc = h->cap ? h->cap * 2 : 1024;
        hent *na = (hent *)realloc(h->a, nc * sizeof(hent));
        if (!na) return -1;
        h->a = na; h->cap = nc;
    }
    size_t i = h->len++;
    hent v = { count, key };
    while (i > 0) {
        size_t p = (i - 1) / 2;
        if (!hless(&h->a[p], &v)) break;
        h->a[i] = h->a[p];
        i = p;
    }
    h->a[i] = v;
    return 0;
}

static hent heap_pop(heap *h)
{
    hent top = h->a[0];
    hent v = h->a[--h->len];
    size_t i = 0;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= h->len) break;
        if (c + 1 < h->len && hless(&h->a[c], &h->a[c + 1])) ++c;
        if (!hless(&v, &h->a[c])) break;
        h->a[i] = h->a[c];
        i = c;
    }
    if (h->len) h->a[i] = v;
    return top;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

static int occ_push(pentry *p, uint32_t pos)
{
    if (p->occ_len == p->occ_cap) {
        uint32_t nc = p->occ_cap ? p->occ_cap * 2 : 4;
        uint32_t *no = (uint32_t *)realloc(p->occ, (size_t)nc * sizeof(uint32_t));
        if (!no) return -1;
        p->occ = no; p->occ_cap = nc;
    }
    p->occ[p->occ_len++] = pos;
    return 0;
}

typedef struct { uint64_t *a; size_t len, cap; } keyvec;

static int keyvec_push(keyvec *v, uint64_t k)
{
    if (v->len == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 256;
        uint64_t *na = (uint64_t *)realloc(v->a, nc * sizeof(uint64_t));
        if (!na) return -1;
        v->a = na; v->cap = nc;
    }
    v->a[v->len++] = k;
    return 0;
}

static void dec_pair(pmap *m, uint32_t l, uint32_t r)
{
    pentry *p = pmap_find(m, ((uint64_t)l << 32) | r);
    if (!p) return;
    if (--p->count == 0) { free(p->occ); p->occ = NULL; p->occ_len = p->occ_cap = 0; }
}

static int inc_pair(pmap *m, keyvec *touched, uint32_t step, uint32_t l, uint32_t r, uint32_t pos)
{
    uint64_t key = ((uint64_t)l << 32) | r;
    pentry *p = pmap_get_or_add(m, key);
    if (!p) return -1;
    p->count++;
    if (p->stamp != step) { p->stamp = step; if (keyvec_push(touched, key)) return -1; }
    return occ_push(p, pos);
}

/* Same contract as ecgb_oracle_bpe_train: ids_io in/out, *n_io length in/out, out_pairs gets
 * (left,right) per performed merge; returns the number of merges performed, or
 * UINT32_MAX on allocation failure. */
uint32_t ecgb_oracle_bpe_train_fast(uint32_t *ids_io, size_t *n_io, uint32_t num_merges,
                                    uint32_t *out_pairs)
{
    size_t n = *n_io;
    if (n >= END) return UINT32_MAX;
    uint32_t done = 0;
    uint32_t *prv = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *nxt = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    pmap m; heap hp = { 0, 0, 0 }; keyvec touched = { 0, 0, 0 };
    if (!prv || !nxt || pmap_init(&m, 1 << 16)) return UINT32_MAX;
    for (size_t i = 0; i < n; ++i) {
        prv[i] = i ? (uint32_t)(i - 1) : END;
        nxt[i] = (i + 1 < n) ? (uint32_t)(i + 1) : END;
    }
    /* initial histogram: count, then exact-size occurrence lists in position order */
    for (size_t i = 0; i + 1 < n; ++i) {
        pentry *p = pmap_get_or_add(&m, ((uint64_t)ids_io[i] << 32) | ids_io[i + 1]);
        if (!p) return UINT32_MAX;
        p->count++;
    }
    for (size_t i = 0; i < m.cap; ++i) {
        pentry *p = &m.e[i];
        if (p->key == UINT64_MAX) continue;
        p->occ = (uint32_t *)malloc((size_t)p->count * sizeof(uint32_t));
        if (!p->occ) return UINT32_MAX;
        p->occ_cap = (uint32_t)p->count; p->occ_len = 0;
        if (heap_push(&hp, p->count, p->key)) return UINT32_MAX;
    }
    for (size_t i = 0; i + 1 < n; ++i) {
        pentry *p = pmap_find(&m, ((uint64_t)ids_io[i] << 32) | ids_io[i + 1]);
        p->occ[p->occ_len++] = (uint32_t)i;
    }

    for (uint32_t step = 0; step < num_merges; ++step) {
        /* arg-max with lazy validation */
        uint64_t best = UINT64_MAX;
        while (hp.len) {
            hent t = heap_pop(&hp);
            pentry *p = pmap_find(&m, t.key);
            if (!p || p->count <= 0) continue;
            if (p->count != t.count) { if (heap_push(&hp, p->count, t.key)) return UINT32_MAX; continue; }
            best = t.key;
            break;
        }
        if (best == UINT64_MAX) break; /* no pairs left: lib.rs:88-90 */
        const uint32_t l = (uint32_t)(best >> 32), r = (uint32_t)best, X = 256u + step;
        pentry *bp = pmap_find(&m, best);
        uint32_t *occ = bp->occ; uint32_t occ_len = bp->occ_len;
        bp->occ = NULL; bp->occ_len = bp->occ_cap = 0; /* detach: the map may rehash below */
        if (l == r) { /* runs: greedy left-to-right needs position order */
            int sorted = 1;
            for (uint32_t k = 1; k < occ_len; ++k) if (occ[k - 1] > occ[k]) { sorted = 0; break; }
            if (!sorted) qsort(occ, occ_len, sizeof(uint32_t), cmp_u32);
        }
        touched.len = 0;
        for (uint32_t k = 0; k < occ_len; ++k) {
            uint32_t pos = occ[k];
            if (ids_io[pos] != l) continue; /* stale */
            uint32_t j = nxt[pos];
            if (j == END || ids_io[j] != r) continue;
            uint32_t p = prv[pos], q = nxt[j];
            if (p != END) dec_pair(&m, ids_io[p], l);
            dec_pair(&m, l, r);
            if (q != END) dec_pair(&m, r, ids_io[q]);
            ids_io[pos] = X; ids_io[j] = DEAD;
            nxt[pos] = q;
            if (q != END) prv[q] = pos;
            if (p != END && inc_pair(&m, &touched, step + 1, ids_io[p], X, p)) return UINT32_MAX;
            if (q != END && inc_pair(&m, &touched, step + 1, X, ids_io[q], pos)) return UINT32_MAX;
        }
        free(occ);
        for (size_t k = 0; k < touched.len; ++k) {
            pentry *p = pmap_find(&m, touched.a[k]);
            if (p && p->count > 0 && heap_push(&hp, p->count, p->key)) return UINT32_MAX;
        }
        out_pairs[2 * done] = l; out_pairs[2 * done + 1] = r;
        ++done;
    }

    size_t w = 0;
    if (n) {
        /* position 0 is never consumed as a right element, so it heads the list */
        uint32_t *tmp = (uint32_t *)malloc(n * sizeof(uint32_t));
        if (!tmp) return UINT32_MAX;
        for (uint32_t i = 0; i != END; i = nxt[i]) tmp[w++] = ids_io[i];
        memcpy(ids_io, tmp, w * sizeof(uint32_t));
        free(tmp);
    }
    *n_io = w;
    for (size_t i = 0; i < m.cap; ++i) if (m.e[i].key != UINT64_MAX) free(m.e[i].occ);
    free(m.e); free(hp.a); free(touched.a); free(prv); free(nxt);
    return done;
}
