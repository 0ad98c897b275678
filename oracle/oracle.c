/* oracle.c -- CPU restatement of the reference's encode / encode_batch path in plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (splintr_amd/csrc) never links,
 * loads or calls it.
 *
 * What follows what (paths into /root/reference):
 *   orc_bpe()            src/core/bpe.rs:67-197       index-linked node list, leftmost strict
 *                                                     minimum, byte-slice re-ranking, fallbacks
 *   orc_byte_level()     src/core/byte_level.rs:46-74, 105-107
 *   orc_encode_chunk()   src/core/tokenizer.rs:693-724 (LRU :708-721 is result-transparent; the
 *                                                     optional mutex-guarded memo below mimics
 *                                                     its cost model for the baseline timing)
 *   orc_encode()         src/core/tokenizer.rs:729-808 (non-SentencePiece branch)
 *   orc_encode_special() src/core/tokenizer.rs:842-874 (Aho-Corasick Standard, non-overlapping)
 *   orc_encode_batch()   src/core/tokenizer.rs:932-942 (Rayon par_iter -> persistent pthread pool
 *                                                     that pulls documents off a shared counter)
 *   byte-keyed hash map  FxHashMap<Vec<u8>,u32> (src/core/tokenizer.rs:302) -> open addressing
 *   vocab parse          src/core/vocab.rs:57-89 semantics (later duplicate wins), from .splv
 *
 * The regex pre-tokeniser (src/core/tokenizer.rs:244-257) is third-party arithmetic that is NOT
 * under /root/reference: regexr 0.1.0-beta.5 (default) / pcre2 0.2 -> libpcre2-8 with UTF|UCP.
 * It is restated here as a tiny backtracking matcher specialised to the constructs of the two
 * verbatim patterns (tokenizer.rs:39, :42): an ordered alternation of item sequences, each item
 * a greedy quantified character class, the caseless contraction group, or the (?!\S)
 * look-ahead, with Perl leftmost-first / greedy-with-backtracking semantics.  Character classes
 * come from splintr_amd/data/unicode_classes.bin (probed from PCRE2 10.39, Unicode 14.0.0).
 * Pinned by: the 18 reference vectors (tests/golden/reference_vectors.json) and differential
 * fuzz against libpcre2-8 and Python `regex` (tests/test_oracle.py).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- class table */
enum { C_P = 0, C_AP, C_SP, C_WS, C_NL, C_N, C_LU, C_LL, C_LT, C_LM, C_LO, C_M, C_COUNT };
#define BIT(c) (1u << (c))
#define M_L (BIT(C_LU) | BIT(C_LL) | BIT(C_LT) | BIT(C_LM) | BIT(C_LO))
#define M_S (BIT(C_SP) | BIT(C_WS) | BIT(C_NL))
#define M_ALL ((1u << C_COUNT) - 1)

typedef struct {
    uint32_t shift, nblocks;
    uint16_t *stage1;
    uint8_t *stage2;
    uint32_t nfold;
    uint32_t *fold; /* pairs: code point, ascii lower */
} uclass_t;

static int uclass_load(uclass_t *u, const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char magic[4];
    uint32_t ver;
    char uver[16];
    if (fread(magic, 1, 4, f) != 4 || memcmp(magic, "SPLU", 4)) { fclose(f); return -2; }
    if (fread(&ver, 4, 1, f) != 1 || (ver < 1 || ver > 3)) { fclose(f); return -2; }   /* (versions 2 / 3 append general categories / scripts: not used here) */
    if (fread(&u->shift, 4, 1, f) != 1 || fread(&u->nblocks, 4, 1, f) != 1 ||
        fread(uver, 1, 16, f) != 16) { fclose(f); return -2; }
    size_t n1 = 0x110000u >> u->shift, n2 = (size_t)u->nblocks << u->shift;
    u->stage1 = malloc(n1 * 2);
    u->stage2 = malloc(n2);
    if (fread(u->stage1, 2, n1, f) != n1 || fread(u->stage2, 1, n2, f) != n2) { fclose(f); return -3; }
    if (fread(&u->nfold, 4, 1, f) != 1) { fclose(f); return -3; }
    u->fold = malloc(8 * (size_t)u->nfold);
    if (fread(u->fold, 8, u->nfold, f) != u->nfold) { fclose(f); return -3; }
    fclose(f);
    return 0;
}
static inline int uclass_of(const uclass_t *u, uint32_t cp) {
    if (cp >= 0x110000u) return C_P;
    return u->stage2[((size_t)u->stage1[cp >> u->shift] << u->shift) | (cp & ((1u << u->shift) - 1))];
}

/* ---------------------------------------------------------------- UTF-8 helpers
 * The reference's core only ever sees &str (valid UTF-8).  For byte strings that are not, this
 * restatement follows the policy the C ABI states (include/splintr_hip.h), one text at a time:
 * a lead byte (0xC0..0xFF; 0xF0 and above announce four bytes) takes the continuation bytes that
 * follow, at most as many as it announces -- all present: the character with the decoded value,
 * looked up as it is; otherwise ONE character of class "other" made of the bytes it got; a
 * continuation byte that no lead reaches is a character of class "other" by itself.
 * Code point 0xFFFFFFFF stands for "class other" (uclass_of maps everything >= 0x110000 to C_P). */
static inline size_t u8_want(uint8_t b) { return b < 0xC0 ? 1 : b < 0xE0 ? 2 : b < 0xF0 ? 3 : 4; }
static inline uint32_t u8_decode(const uint8_t *s, size_t n, size_t pos, size_t *len) {
    uint8_t b = s[pos];
    if (b < 0x80) { *len = 1; return b; }
    if (b < 0xC0) { *len = 1; return 0xFFFFFFFFu; }             /* (callers only ask at character starts) */
    size_t want = u8_want(b), l = 1;
    while (l < want && pos + l < n && (s[pos + l] & 0xC0) == 0x80) l++;
    *len = l;
    if (l < want) return 0xFFFFFFFFu;
    if (want == 2) return ((b & 0x1Fu) << 6) | (s[pos + 1] & 0x3Fu);
    if (want == 3) return ((b & 0x0Fu) << 12) | ((s[pos + 1] & 0x3Fu) << 6) | (s[pos + 2] & 0x3Fu);
    return ((b & 0x07u) << 18) | ((s[pos + 1] & 0x3Fu) << 12) | ((s[pos + 2] & 0x3Fu) << 6) | (s[pos + 3] & 0x3Fu);
}
static inline int u8_is_start(const uint8_t *s, size_t pos) { /* does a character start at pos? */
    if ((s[pos] & 0xC0) != 0x80) return 1;
    for (size_t k = 1; k <= 3; k++) {
        if (pos < k) return 1;
        uint8_t b = s[pos - k];
        if (b >= 0xC0) return u8_want(b) <= k;
        if (b < 0x80) return 1;
    }
    return 1;
}
static inline size_t u8_prev(const uint8_t *s, size_t pos) { /* start of the char ending at pos */
    do { pos--; } while (pos > 0 && !u8_is_start(s, pos));
    return pos;
}

/* ---------------------------------------------------------------- mini regex */
enum { IT_SET, IT_CONTR, IT_NOT_NONSPACE_AHEAD };
typedef struct { int kind; uint32_t mask; int min, max; /* max<0: unbounded */ uint32_t lit; /* extra code point in the set, 0: none */ } item_t;
typedef struct { int nitems; item_t it[6]; } alt_t;
typedef struct { int nalts; alt_t alt[8]; } pattern_t;

typedef struct {
    const uclass_t *u;
    const uint8_t *s;
    size_t n;
} subj_t;

static int fold_eq(const uclass_t *u, uint32_t cp, char lower) {
    for (uint32_t i = 0; i < u->nfold; i++)
        if (u->fold[2 * i] == cp && u->fold[2 * i + 1] == (uint32_t)lower) return 1;
    return 0;
}

/* (?i:'s|'t|'re|'ve|'m|'ll|'d) at pos; returns end or (size_t)-1 */
static size_t match_contraction(const subj_t *S, size_t pos) {
    static const char *alts[] = {"s", "t", "re", "ve", "m", "ll", "d"};
    if (pos >= S->n || S->s[pos] != '\'') return (size_t)-1;
    for (int a = 0; a < 7; a++) {
        size_t p = pos + 1;
        int ok = 1;
        for (const char *c = alts[a]; *c; c++) {
            if (p >= S->n) { ok = 0; break; }
            size_t l;
            uint32_t cp = u8_decode(S->s, S->n, p, &l);
            if (!fold_eq(S->u, cp, *c)) { ok = 0; break; }
            p += l;
        }
        if (ok) return p;
    }
    return (size_t)-1;
}

static size_t match_items(const subj_t *S, const alt_t *A, int i, size_t pos) {
    if (i == A->nitems) return pos;
    const item_t *it = &A->it[i];
    if (it->kind == IT_NOT_NONSPACE_AHEAD) { /* (?!\S) */
        if (pos < S->n) {
            size_t l;
            int c = uclass_of(S->u, u8_decode(S->s, S->n, pos, &l));
            if (!(BIT(c) & M_S)) return (size_t)-1;
        }
        return match_items(S, A, i + 1, pos);
    }
    if (it->kind == IT_CONTR) { /* min==0: optional, greedy */
        size_t e = match_contraction(S, pos);
        if (e != (size_t)-1) {
            size_t r = match_items(S, A, i + 1, e);
            if (r != (size_t)-1) return r;
        }
        if (it->min == 0) return match_items(S, A, i + 1, pos);
        return (size_t)-1;
    }
    /* greedy quantified class: take as many as allowed, then give back one at a time */
    int cnt = 0;
    size_t p = pos;
    while ((it->max < 0 || cnt < it->max) && p < S->n) {
        size_t l;
        uint32_t cp = u8_decode(S->s, S->n, p, &l);
        int c = uclass_of(S->u, cp);
        if (!(BIT(c) & it->mask) && !(it->lit && cp == it->lit)) break;
        p += l;
        cnt++;
    }
    for (;;) {
        if (cnt < it->min) return (size_t)-1;
        size_t r = match_items(S, A, i + 1, p);
        if (r != (size_t)-1) return r;
        if (cnt == 0) return (size_t)-1;
        p = u8_prev(S->s, p);
        cnt--;
    }
}

static size_t match_at(const subj_t *S, const pattern_t *P, size_t pos) {
    for (int a = 0; a < P->nalts; a++) {
        size_t e = match_items(S, &P->alt[a], 0, pos);
        if (e != (size_t)-1 && e > pos) return e;
        /* an empty match cannot occur for these patterns (every alternative consumes >= 1) */
    }
    return (size_t)-1;
}

#define SET(m, lo, hi) { IT_SET, (m), (lo), (hi), 0 }
#define SETL(m, lo, hi, lit) { IT_SET, (m), (lo), (hi), (lit) }
static void tail_alts(pattern_t *P) { /* \p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+ */
    alt_t a3 = {1, {SET(BIT(C_N), 1, 3)}};
    alt_t a4 = {3, {SET(BIT(C_SP), 0, 1), SET(M_ALL & ~(M_S | M_L | BIT(C_N)), 1, -1), SET(BIT(C_NL), 0, -1)}};
    alt_t a5 = {2, {SET(M_S, 0, -1), SET(BIT(C_NL), 1, -1)}};
    alt_t a6 = {2, {SET(M_S, 1, -1), {IT_NOT_NONSPACE_AHEAD, 0, 0, 0, 0}}};
    alt_t a7 = {1, {SET(M_S, 1, -1)}};
    P->alt[P->nalts++] = a3;
    P->alt[P->nalts++] = a4;
    P->alt[P->nalts++] = a5;
    P->alt[P->nalts++] = a6;
    P->alt[P->nalts++] = a7;
}
static void pattern_cl100k(pattern_t *P) { /* tokenizer.rs:39 */
    P->nalts = 0;
    alt_t a1 = {1, {{IT_CONTR, 0, 1, 1, 0}}};
    alt_t a2 = {2, {SET(M_ALL & ~(BIT(C_NL) | M_L | BIT(C_N)), 0, 1), SET(M_L, 1, -1)}};
    P->alt[P->nalts++] = a1;
    P->alt[P->nalts++] = a2;
    tail_alts(P);
}
static void pattern_mistral_v3(pattern_t *P) { /* tokenizer.rs:64: no contractions, \p{N} single, [\r\n/]* */
    P->nalts = 0;
    const uint32_t X = M_ALL & ~(BIT(C_NL) | M_L | BIT(C_N));
    const uint32_t U = BIT(C_LU) | BIT(C_LT) | BIT(C_LM) | BIT(C_LO) | BIT(C_M);
    const uint32_t W = BIT(C_LL) | BIT(C_LM) | BIT(C_LO) | BIT(C_M);
    alt_t a1 = {3, {SET(X, 0, 1), SET(U, 0, -1), SET(W, 1, -1)}};
    alt_t a2 = {3, {SET(X, 0, 1), SET(U, 1, -1), SET(W, 0, -1)}};
    alt_t a3 = {1, {SET(BIT(C_N), 1, 1)}};
    alt_t a4 = {3, {SET(BIT(C_SP), 0, 1), SET(M_ALL & ~(M_S | M_L | BIT(C_N)), 1, -1), SETL(BIT(C_NL), 0, -1, '/')}};
    alt_t a5 = {2, {SET(M_S, 0, -1), SET(BIT(C_NL), 1, -1)}};
    alt_t a6 = {2, {SET(M_S, 1, -1), {IT_NOT_NONSPACE_AHEAD, 0, 0, 0, 0}}};
    alt_t a7 = {1, {SET(M_S, 1, -1)}};
    P->alt[P->nalts++] = a1; P->alt[P->nalts++] = a2; P->alt[P->nalts++] = a3; P->alt[P->nalts++] = a4;
    P->alt[P->nalts++] = a5; P->alt[P->nalts++] = a6; P->alt[P->nalts++] = a7;
}
static void pattern_o200k(pattern_t *P) { /* tokenizer.rs:42 */
    P->nalts = 0;
    uint32_t X = M_ALL & ~(BIT(C_NL) | M_L | BIT(C_N));
    uint32_t U = BIT(C_LU) | BIT(C_LT) | BIT(C_LM) | BIT(C_LO) | BIT(C_M);
    uint32_t W = BIT(C_LL) | BIT(C_LM) | BIT(C_LO) | BIT(C_M);
    alt_t a1 = {4, {SET(X, 0, 1), SET(U, 0, -1), SET(W, 1, -1), {IT_CONTR, 0, 0, 1, 0}}};
    alt_t a2 = {4, {SET(X, 0, 1), SET(U, 1, -1), SET(W, 0, -1), {IT_CONTR, 0, 0, 1, 0}}};
    P->alt[P->nalts++] = a1;
    P->alt[P->nalts++] = a2;
    tail_alts(P);
}

/* ---------------------------------------------------------------- byte-keyed hash map */
typedef struct { uint32_t off, len, rank, used; } hent_t;
typedef struct {
    hent_t *e;
    uint32_t cap; /* power of two */
    uint8_t *keys;
    size_t keys_len;
    uint32_t count, max_rank;
} bmap_t;

static inline uint64_t fx_hash(const uint8_t *p, size_t n) { /* FxHasher-style word mixing */
    const uint64_t K = 0x517cc1b727220a95ull;
    uint64_t h = 0;
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = ((h << 5 | h >> 59) ^ w) * K; p += 8; n -= 8; }
    if (n >= 4) { uint32_t w; memcpy(&w, p, 4); h = ((h << 5 | h >> 59) ^ w) * K; p += 4; n -= 4; }
    while (n--) h = ((h << 5 | h >> 59) ^ *p++) * K;
    return h;
}
static int bmap_get(const bmap_t *m, const uint8_t *k, size_t n, uint32_t *rank) {
    uint32_t i = (uint32_t)(fx_hash(k, n) >> 20) & (m->cap - 1);
    for (;;) {
        const hent_t *e = &m->e[i];
        if (!e->used) return 0;
        if (e->len == n && !memcmp(m->keys + e->off, k, n)) { *rank = e->rank; return 1; }
        i = (i + 1) & (m->cap - 1);
    }
}
static void bmap_put(bmap_t *m, uint32_t off, uint32_t len, uint32_t rank) {
    const uint8_t *k = m->keys + off;
    uint32_t i = (uint32_t)(fx_hash(k, len) >> 20) & (m->cap - 1);
    for (;;) {
        hent_t *e = &m->e[i];
        if (!e->used) { e->used = 1; e->off = off; e->len = len; e->rank = rank; m->count++; return; }
        if (e->len == len && !memcmp(m->keys + e->off, k, len)) { e->rank = rank; return; } /* last wins */
        i = (i + 1) & (m->cap - 1);
    }
}

/* ---------------------------------------------------------------- tokenizer object */
typedef struct { uint8_t *lit; uint32_t len, id; } special_t;

#define MEMO_SLOTS 4096 /* DEFAULT_CACHE_SIZE, tokenizer.rs:233 */
typedef struct { uint64_t key; uint32_t n; uint32_t *ids; } memo_t;

typedef struct orc {
    bmap_t map;
    uclass_t ucl;
    pattern_t pat;
    int byte_level;
    uint8_t bl_bytes[256][2];
    uint8_t bl_len[256];
    special_t *sp;
    uint32_t nsp;
    int use_memo; /* 0: none; 1: mutex-guarded 4096-entry direct-mapped memo keyed by 64-bit hash */
    pthread_mutex_t memo_mu;
    memo_t memo[MEMO_SLOTS];
} orc_t;

typedef struct { uint32_t *v; size_t n, cap; } ivec_t;
static void iv_push(ivec_t *v, uint32_t x) {
    if (v->n == v->cap) { v->cap = v->cap ? v->cap * 2 : 64; v->v = realloc(v->v, v->cap * 4); }
    v->v[v->n++] = x;
}

static void byte_level_init(orc_t *t) { /* byte_level.rs:46-74 */
    int direct[256] = {0};
    for (int b = 33; b <= 126; b++) direct[b] = 1;
    for (int b = 161; b <= 172; b++) direct[b] = 1;
    for (int b = 174; b <= 255; b++) direct[b] = 1;
    uint32_t next = 256;
    for (int b = 0; b < 256; b++) {
        uint32_t cp = direct[b] ? (uint32_t)b : next++;
        if (cp < 0x80) { t->bl_bytes[b][0] = (uint8_t)cp; t->bl_len[b] = 1; }
        else { t->bl_bytes[b][0] = 0xC0 | (cp >> 6); t->bl_bytes[b][1] = 0x80 | (cp & 0x3F); t->bl_len[b] = 2; }
    }
}

orc_t *orc_create(const char *splv_path, const char *uclass_path, int pattern_id /*0 cl100k, 1 o200k*/,
                  int byte_level) {
    orc_t *t = calloc(1, sizeof *t);
    if (uclass_load(&t->ucl, uclass_path)) { free(t); return NULL; }
    FILE *f = fopen(splv_path, "rb");
    if (!f) { free(t); return NULL; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *blob = malloc(sz);
    if (fread(blob, 1, sz, f) != (size_t)sz) { fclose(f); free(t); return NULL; }
    fclose(f);
    if (memcmp(blob, "SPLV", 4)) { free(blob); free(t); return NULL; }
    uint32_t n;
    memcpy(&n, blob + 8, 4);
    t->map.cap = 1;
    while (t->map.cap < n * 2u + 16) t->map.cap <<= 1;
    t->map.e = calloc(t->map.cap, sizeof(hent_t));
    t->map.keys = blob; /* keys live inside the blob */
    size_t off = 20;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t rank;
        uint16_t len;
        memcpy(&rank, blob + off, 4);
        memcpy(&len, blob + off + 4, 2);
        off += 6;
        bmap_put(&t->map, (uint32_t)off, len, rank);
        if (rank > t->map.max_rank) t->map.max_rank = rank;
        off += len;
    }
    if (pattern_id == 0) pattern_cl100k(&t->pat); else if (pattern_id == 2) pattern_mistral_v3(&t->pat); else pattern_o200k(&t->pat);
    t->byte_level = byte_level;
    byte_level_init(t);
    pthread_mutex_init(&t->memo_mu, NULL);
    return t;
}

void orc_set_memo(orc_t *t, int on) { t->use_memo = on; }

void orc_add_special(orc_t *t, const uint8_t *lit, uint32_t len, uint32_t id) {
    t->sp = realloc(t->sp, (t->nsp + 1) * sizeof(special_t));
    t->sp[t->nsp].lit = malloc(len);
    memcpy(t->sp[t->nsp].lit, lit, len);
    t->sp[t->nsp].len = len;
    t->sp[t->nsp].id = id;
    t->nsp++;
}

void orc_destroy(orc_t *t) {
    if (!t) return;
    free(t->map.e); free(t->map.keys); free(t->ucl.stage1); free(t->ucl.stage2); free(t->ucl.fold);
    for (uint32_t i = 0; i < t->nsp; i++) free(t->sp[i].lit);
    free(t->sp);
    for (int i = 0; i < MEMO_SLOTS; i++) free(t->memo[i].ids);
    free(t);
}

/* ---------------------------------------------------------------- BPE (bpe.rs:67-197) */
#define NIL ((size_t)-1)
typedef struct { size_t prev, next; uint32_t rank; size_t start, len; } node_t;

static uint32_t get_rank(const orc_t *t, const uint8_t *piece, const node_t *nd, size_t l, size_t r) {
    if (l == NIL || r == NIL) return UINT32_MAX;
    uint32_t rk;
    if (bmap_get(&t->map, piece + nd[l].start, nd[l].len + nd[r].len, &rk)) return rk;
    return UINT32_MAX;
}

static void orc_bpe(const orc_t *t, const uint8_t *piece, size_t n, ivec_t *out) {
    uint32_t rk;
    if (n == 0) return;
    if (n == 1) { if (bmap_get(&t->map, piece, 1, &rk)) iv_push(out, rk); return; }
    if (bmap_get(&t->map, piece, n, &rk)) { iv_push(out, rk); return; }
    /* (the reference heap-allocates its node vector, bpe.rs:83; small pieces use the stack here so that the
     *  timed CPU baseline is not charged for the allocator) */
    node_t stack_nd[64];
    node_t *nd = n <= 64 ? stack_nd : malloc(n * sizeof(node_t));
    for (size_t i = 0; i < n; i++) {
        nd[i].prev = i == 0 ? NIL : i - 1;
        nd[i].next = i == n - 1 ? NIL : i + 1;
        nd[i].rank = UINT32_MAX;
        nd[i].start = i;
        nd[i].len = 1;
    }
    for (size_t i = 0; i + 1 < n; i++) nd[i].rank = get_rank(t, piece, nd, i, nd[i].next);
    for (;;) {
        uint32_t min_rank = UINT32_MAX;
        size_t min_idx = NIL, curr = 0;
        while (nd[curr].prev != NIL) curr = nd[curr].prev;
        while (curr != NIL) {
            if (nd[curr].rank < min_rank) { min_rank = nd[curr].rank; min_idx = curr; }
            curr = nd[curr].next;
        }
        if (min_rank == UINT32_MAX) break;
        size_t nx = nd[min_idx].next;
        nd[min_idx].len += nd[nx].len;
        size_t nn = nd[nx].next;
        nd[min_idx].next = nn;
        if (nn != NIL) nd[nn].prev = min_idx;
        if (nd[min_idx].prev != NIL) {
            size_t pv = nd[min_idx].prev;
            nd[pv].rank = get_rank(t, piece, nd, pv, min_idx);
        }
        nd[min_idx].rank = get_rank(t, piece, nd, min_idx, nd[min_idx].next);
    }
    size_t curr = 0;
    while (nd[curr].prev != NIL) curr = nd[curr].prev;
    while (curr != NIL) {
        const uint8_t *sl = piece + nd[curr].start;
        if (bmap_get(&t->map, sl, nd[curr].len, &rk)) iv_push(out, rk);
        else
            for (size_t j = 0; j < nd[curr].len; j++)
                if (bmap_get(&t->map, sl + j, 1, &rk)) iv_push(out, rk);
        curr = nd[curr].next;
    }
    if (nd != stack_nd) free(nd);
}

/* tokenizer.rs:693-724 */
static void orc_encode_chunk(orc_t *t, const uint8_t *sl, size_t n, ivec_t *out) {
    uint8_t stackbuf[512], *buf = stackbuf;
    const uint8_t *p = sl;
    size_t pn = n;
    if (t->byte_level) {
        if (2 * n > sizeof stackbuf) buf = malloc(2 * n);
        pn = 0;
        for (size_t i = 0; i < n; i++) {
            buf[pn++] = t->bl_bytes[sl[i]][0];
            if (t->bl_len[sl[i]] == 2) buf[pn++] = t->bl_bytes[sl[i]][1];
        }
        p = buf;
    }
    uint32_t rk;
    if (bmap_get(&t->map, p, pn, &rk)) { iv_push(out, rk); goto done; }
    if (t->use_memo) {
        /* hash_slice (tokenizer.rs:643-647) is `slice.hash(&mut FxHasher)`: the LENGTH goes into the hasher before
         * the bytes.  Without it "\0" and "\0\0" (h stays 0) collide and the memo hands out the wrong ids
         * (tests/test_sanitizers.py found it: 4 configurations, 3 different token counts).  Like the reference's
         * LruCache<u64, Vec<u32>>, the memo is keyed by the 64-bit hash alone. */
        const uint64_t K = 0x517cc1b727220a95ull;
        uint64_t h = (fx_hash(p, pn) ^ (((uint64_t)pn * K) << 5 | ((uint64_t)pn * K) >> 59)) * K | 1;
        memo_t *m = &t->memo[(h >> 32) % MEMO_SLOTS];
        pthread_mutex_lock(&t->memo_mu);
        if (m->key == h) {
            for (uint32_t i = 0; i < m->n; i++) iv_push(out, m->ids[i]);
            pthread_mutex_unlock(&t->memo_mu);
            goto done;
        }
        pthread_mutex_unlock(&t->memo_mu);
        size_t before = out->n;
        orc_bpe(t, p, pn, out);
        pthread_mutex_lock(&t->memo_mu);
        m->key = h;
        m->n = (uint32_t)(out->n - before);
        m->ids = realloc(m->ids, 4 * (size_t)m->n + 4);
        memcpy(m->ids, out->v + before, 4 * (size_t)m->n);
        pthread_mutex_unlock(&t->memo_mu);
        goto done;
    }
    orc_bpe(t, p, pn, out);
done:
    if (buf != stackbuf) free(buf);
}

/* tokenizer.rs:729-808 ; optionally reports the chunk boundaries */
static void orc_encode_into(orc_t *t, const uint8_t *s, size_t n, ivec_t *out, ivec_t *bounds) {
    subj_t S = {&t->ucl, s, n};
    size_t pos = 0;
    while (pos < n) {
        size_t e = match_at(&S, &t->pat, pos);
        if (e == (size_t)-1) { /* unmatched char: find_iter would skip it (never for these patterns) */
            size_t l;
            u8_decode(s, n, pos, &l);
            pos += l;
            continue;
        }
        if (bounds) iv_push(bounds, (uint32_t)pos);
        orc_encode_chunk(t, s + pos, e - pos, out);
        pos = e;
    }
}

/* tokenizer.rs:842-874 */
static void orc_encode_special_into(orc_t *t, const uint8_t *s, size_t n, ivec_t *out) {
    if (!t->nsp) { orc_encode_into(t, s, n, out, NULL); return; }
    size_t last = 0, pos = 0;
    while (pos < n) {
        size_t best_end = NIL, best_start = 0;
        uint32_t best_id = 0;
        for (uint32_t k = 0; k < t->nsp; k++) {
            const special_t *sp = &t->sp[k];
            if (sp->len == 0 || n - pos < sp->len) continue;
            const uint8_t *hit = memmem(s + pos, n - pos, sp->lit, sp->len);
            if (!hit) continue;
            size_t st = (size_t)(hit - s), en = st + sp->len;
            if (en < best_end || (en == best_end && st < best_start)) { best_end = en; best_start = st; best_id = sp->id; }
        }
        if (best_end == NIL) break;
        if (best_start > last) orc_encode_into(t, s + last, best_start - last, out, NULL);
        iv_push(out, best_id);
        last = pos = best_end;
    }
    if (last < n) orc_encode_into(t, s + last, n - last, out, NULL);
}

/* ---------------------------------------------------------------- exported entry points */
/* Encode one text.  Returns token count; *ids is malloc'd (free with orc_free). */
size_t orc_encode(orc_t *t, const uint8_t *s, size_t n, int with_special, uint32_t **ids) {
    ivec_t v = {0};
    if (with_special) orc_encode_special_into(t, s, n, &v); else orc_encode_into(t, s, n, &v, NULL);
    *ids = v.v;
    return v.n;
}
/* Chunk start offsets of the pre-tokeniser alone. */
size_t orc_split(orc_t *t, const uint8_t *s, size_t n, uint32_t **starts) {
    ivec_t v = {0}, b = {0};
    orc_encode_into(t, s, n, &v, &b);
    free(v.v);
    *starts = b.v;
    return b.n;
}
void orc_free(void *p) { free(p); }

typedef struct {
    orc_t *t;
    const uint8_t *text;
    const uint64_t *off;
    uint64_t ndocs;
    int with_special;
    ivec_t *res;
    uint64_t next; /* shared work counter */
} batch_t;

static void *batch_worker(void *arg) {
    batch_t *b = arg;
    for (;;) {
        uint64_t d = __atomic_fetch_add(&b->next, 1, __ATOMIC_RELAXED);
        if (d >= b->ndocs) break;
        const uint8_t *s = b->text + b->off[d];
        size_t n = (size_t)(b->off[d + 1] - b->off[d]);
        if (b->with_special) orc_encode_special_into(b->t, s, n, &b->res[d]);
        else orc_encode_into(b->t, s, n, &b->res[d], NULL);
    }
    return NULL;
}

/* Persistent worker pool (Rayon keeps one too: the reference pays no thread creation per call). */
typedef struct {
    pthread_t *th;
    int n;
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    uint64_t gen;
    int pending;
    batch_t *job;
    int stop;
} pool_t;
static pool_t g_pool = {NULL, 0, PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, NULL, 0};

static void *pool_main(void *arg) {
    uint64_t seen = (uint64_t)(uintptr_t)arg;      /* generation at creation: only later jobs count */
    for (;;) {
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.gen == seen && !g_pool.stop) pthread_cond_wait(&g_pool.cv_work, &g_pool.mu);
        if (g_pool.stop) { pthread_mutex_unlock(&g_pool.mu); return NULL; }
        seen = g_pool.gen;
        batch_t *job = g_pool.job;
        pthread_mutex_unlock(&g_pool.mu);
        batch_worker(job);
        pthread_mutex_lock(&g_pool.mu);
        if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.cv_done);
        pthread_mutex_unlock(&g_pool.mu);
    }
}
/* Affinity (bench.py's cpu_baseline): worker i on the (i + 1)-th CPU this process may run on, the caller on the first.
 * Unpinned, the 1 ms calls of a 1000-document batch scattered 50x between repetitions (threads woken on busy or
 * sleeping cores); pinned they stay within a few per cent.  Off by default: tests do not care. */
static int g_pin = 0;
/* An explicit CPU list for the pinned pool (bench.py: the CPUs this process may run on MINUS the ones that are busy -- a worker pinned
 * onto a CPU that something else occupies is not scheduled for a time slice, and a 1 ms call becomes a 10 ms one). */
static int g_cpus[1024], g_ncpus = 0;
void orc_pool_set_cpus(int n, const int *cpus) {
    g_ncpus = 0;
    for (int i = 0; i < n && i < 1024; i++) g_cpus[g_ncpus++] = cpus[i];
}
static void pin_self(int k) {
    cpu_set_t all, one;
    if (g_ncpus > 0) { CPU_ZERO(&one); CPU_SET(g_cpus[k % g_ncpus], &one); pthread_setaffinity_np(pthread_self(), sizeof one, &one); return; }
    if (sched_getaffinity(0, sizeof all, &all) != 0) return;
    int n = CPU_COUNT(&all), want = n ? k % n : 0, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &all)) continue;
        if (seen++ == want) { CPU_ZERO(&one); CPU_SET(c, &one); pthread_setaffinity_np(pthread_self(), sizeof one, &one); return; }
    }
}
static cpu_set_t g_home;
static int g_home_valid = 0;
void orc_pool_pin(int on) {
    if (on && !g_home_valid) { g_home_valid = sched_getaffinity(0, sizeof g_home, &g_home) == 0; }
    if (on == g_pin) return;
    g_pin = on;
    if (on) pin_self(0);
    else if (g_home_valid) pthread_setaffinity_np(pthread_self(), sizeof g_home, &g_home);
    /* the pool is rebuilt by the next call with threads */
    if (g_pool.n) {
        pthread_mutex_lock(&g_pool.mu);
        g_pool.stop = 1;
        pthread_cond_broadcast(&g_pool.cv_work);
        pthread_mutex_unlock(&g_pool.mu);
        for (int i = 0; i < g_pool.n; i++) pthread_join(g_pool.th[i], NULL);
        free(g_pool.th);
        g_pool.th = NULL; g_pool.n = 0; g_pool.stop = 0;
    }
}
typedef struct { uint64_t gen; int index; } pool_arg_t;
static void *pool_entry(void *arg) {
    pool_arg_t a = *(pool_arg_t *)arg;
    free(arg);
    if (g_pin) {
        /* workers inherit the caller's one-CPU mask: widen to the process's home set first, then take their own CPU */
        if (g_home_valid) pthread_setaffinity_np(pthread_self(), sizeof g_home, &g_home);
        pin_self(a.index + 1);
    }
    return pool_main((void *)(uintptr_t)a.gen);
}
static void pool_resize(int n) {
    if (g_pool.n == n) return;
    if (g_pool.n) {
        pthread_mutex_lock(&g_pool.mu);
        g_pool.stop = 1;
        pthread_cond_broadcast(&g_pool.cv_work);
        pthread_mutex_unlock(&g_pool.mu);
        for (int i = 0; i < g_pool.n; i++) pthread_join(g_pool.th[i], NULL);
        free(g_pool.th);
        g_pool.stop = 0;
    }
    g_pool.n = n;
    g_pool.th = malloc(sizeof(pthread_t) * (n ? n : 1));
    for (int i = 0; i < n; i++) {
        pool_arg_t *a = malloc(sizeof *a);
        a->gen = g_pool.gen; a->index = i;
        pthread_create(&g_pool.th[i], NULL, pool_entry, a);
    }
}

/* Batch encode (tokenizer.rs:932-942).  text = concatenated UTF-8, off[ndocs+1].
 * Outputs CSR: *ids (malloc'd), out_off[ndocs+1] (caller-provided). Returns total tokens. */
uint64_t orc_encode_batch(orc_t *t, const uint8_t *text, const uint64_t *off, uint64_t ndocs,
                          int with_special, int nthreads, uint32_t **ids, uint64_t *out_off) {
    batch_t b = {t, text, off, ndocs, with_special, calloc(ndocs ? ndocs : 1, sizeof(ivec_t)), 0};
    if (nthreads < 1) nthreads = 1;
    if (nthreads == 1) batch_worker(&b);
    else {
        pool_resize(nthreads - 1);               /* the calling thread works too */
        pthread_mutex_lock(&g_pool.mu);
        g_pool.job = &b;
        g_pool.pending = g_pool.n;
        g_pool.gen++;
        pthread_cond_broadcast(&g_pool.cv_work);
        pthread_mutex_unlock(&g_pool.mu);
        batch_worker(&b);
        pthread_mutex_lock(&g_pool.mu);
        while (g_pool.pending) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
        pthread_mutex_unlock(&g_pool.mu);
    }
    uint64_t total = 0;
    for (uint64_t d = 0; d < ndocs; d++) { out_off[d] = total; total += b.res[d].n; }
    out_off[ndocs] = total;
    uint32_t *all = malloc(total ? total * 4 : 4);
    for (uint64_t d = 0; d < ndocs; d++) {
        if (b.res[d].n) memcpy(all + out_off[d], b.res[d].v, b.res[d].n * 4);
        free(b.res[d].v);
    }
    free(b.res);
    *ids = all;
    return total;
}

/* Class code of one code point (lets tests cross-check the table against PCRE2). */
int orc_class_of(orc_t *t, uint32_t cp) { return uclass_of(&t->ucl, cp); }
