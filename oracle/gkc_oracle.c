/* gkc_oracle.c — CPU restatement (ORACLE) of GATB-Core's DSK k-mer-counting hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gkc_oracle.h). Plain C, scalar, single-threaded, written for clarity:
 * it follows the reference's *algorithm* step by step (rolling k-mer, rolling minimizer with rescan,
 * super-k-mer split, wire format, decode, sort, run-length count, processor chain, Bloom) and cites
 * the reference file:line for each step. Nothing here is used by the product path.
 *
 * Parity pinning: tests/test_oracle_golden.py (known-answer vectors of the reference's unit tests).
 */
#include "gkc_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <ctype.h>

static const uint64_t RANDOM_VALUES[256] = {
#include "../include/gkc_random_values.inc"
};

/* ------------------------------------------------------------------------------------------------
 * A1  Data::ConvertASCII::get + validNucleotide[]   (tools/misc/api/Data.hpp:185, Data.cpp:3)
 * ---------------------------------------------------------------------------------------------- */
int gko_nt_code(unsigned char c) { return (c >> 1) & 3; }
int gko_nt_valid(unsigned char c)
{
    switch (c) { case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return 1; default: return 0; }
}

/* ------------------------------------------------------------------------------------------------
 * tools/math : LargeInt1.pri:137-154 (revcomp64), :157-170 (hash64), :173-184 (oahash64),
 *              :190-211 (simplehash16, 3 terms), NativeInt64.hpp:210-221 (2 terms)
 * ---------------------------------------------------------------------------------------------- */
uint64_t gko_revcomp64(uint64_t x, unsigned k)
{
    /* reverse the order of the 32 two-bit groups, complement (A<->T, C<->G is code^2), right-align */
    uint64_t r = x;
    r = ((r >> 2) & 0x3333333333333333ULL) | ((r & 0x3333333333333333ULL) << 2);
    r = ((r >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((r & 0x0F0F0F0F0F0F0F0FULL) << 4);
    r = ((r >> 8) & 0x00FF00FF00FF00FFULL) | ((r & 0x00FF00FF00FF00FFULL) << 8);
    r = ((r >> 16) & 0x0000FFFF0000FFFFULL) | ((r & 0x0000FFFF0000FFFFULL) << 16);
    r = (r >> 32) | (r << 32);
    r ^= 0xAAAAAAAAAAAAAAAAULL;
    if (k == 0) return 0;          /* reference shifts by 64 here (UB); callers never use k==0 result */
    return r >> (2 * (32 - k));
}

uint64_t gko_hash64(uint64_t key, uint64_t seed)
{
    uint64_t h = seed;
    h ^= (h << 7) ^ key * (h >> 3) ^ (~((h << 11) + (key ^ (h >> 5))));
    h = (~h) + (h << 21);
    h = h ^ (h >> 24);
    h = (h + (h << 3)) + (h << 8);
    h = h ^ (h >> 14);
    h = (h + (h << 2)) + (h << 4);
    h = h ^ (h >> 28);
    h = h + (h << 31);
    return h;
}

uint64_t gko_oahash64(uint64_t e)
{
    uint64_t c = e;
    c = c ^ (c >> 14);
    c = (~c) + (c << 18);
    c = c ^ (c >> 31);
    c = c * 21;
    c = c ^ (c >> 11);
    c = c + (c << 6);
    c = c ^ (c >> 22);
    return c;
}

uint64_t gko_simplehash16_li1(uint64_t key, int shift)
{
    uint64_t in = key >> shift;
    uint64_t r = RANDOM_VALUES[in & 255];
    in >>= 8;
    r ^= RANDOM_VALUES[in & 255];
    r ^= RANDOM_VALUES[key & 255];
    return r;
}

uint64_t gko_simplehash16_ni64(uint64_t key, int shift)
{
    uint64_t in = key >> shift;
    uint64_t r = RANDOM_VALUES[in & 255];
    in >>= 8;
    r ^= RANDOM_VALUES[in & 255];
    return r;
}

/* LargeInt2.pri:168-197 : revcomp of a 128-bit k-mer composed from two 64-bit revcomps */
static gko_u128 revcomp128(gko_u128 x, unsigned k)
{
    uint64_t hi = (uint64_t)(x >> 64), lo = (uint64_t)x;
    unsigned nb_hi = k > 32 ? k - 32 : 0;
    unsigned nb_lo = k > 32 ? 32 : k;
    uint64_t rhi = (k <= 32) ? 0 : gko_revcomp64(hi, nb_hi);
    uint64_t rlo = gko_revcomp64(lo, nb_lo);
    gko_u128 r = rlo;
    r <<= 2 * nb_hi;
    r += rhi;
    return r;
}
void gko_revcomp128(uint64_t lo, uint64_t hi, unsigned k, uint64_t* olo, uint64_t* ohi)
{
    gko_u128 r = revcomp128(((gko_u128)hi << 64) | lo, k);
    *olo = (uint64_t)r; *ohi = (uint64_t)(r >> 64);
}
/* LargeInt2.pri:200-206 */
uint64_t gko_hash1_128(uint64_t lo, uint64_t hi, uint64_t seed) { return gko_hash64(hi, seed) ^ gko_hash64(lo, seed); }

/* generic helpers on the k-mer integer; W128 selects LargeInt<2> semantics (k>31) */
static inline gko_u128 kmask(unsigned k) { return (k >= 64) ? ~(gko_u128)0 : ((((gko_u128)1) << (2 * k)) - 1); }
static inline gko_u128 revcomp_k(gko_u128 x, unsigned k) { return (k <= 31) ? (gko_u128)gko_revcomp64((uint64_t)x, k) : revcomp128(x, k); }
static inline uint64_t hash1_k(gko_u128 x, uint64_t seed, int wide)
{
    return wide ? gko_hash1_128((uint64_t)x, (uint64_t)(x >> 64), seed) : gko_hash64((uint64_t)x, seed);
}
static inline uint64_t simplehash16_k(gko_u128 x, int shift, int wide)
{
    return wide ? gko_simplehash16_ni64((uint64_t)x, shift) : gko_simplehash16_li1((uint64_t)x, shift);
}

/* ------------------------------------------------------------------------------------------------
 * A2  ModelAbstract::iterate / polynom / ModelCanonical::first,next  (Model.hpp:637-657, 726-765, 858-884)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    unsigned k; gko_u128 mask; gko_u128 f, r; int bad; /* bad = indexBadChar countdown */
} kroll;

static void kroll_first(kroll* s, const char* seq, unsigned k)
{
    s->k = k; s->mask = kmask(k); s->f = 0; s->bad = -1;
    for (unsigned i = 0; i < k; i++) {
        unsigned char c = (unsigned char)seq[i];
        s->f = (s->f << 2) + (unsigned)gko_nt_code(c);           /* polynom */
        if (!gko_nt_valid(c)) s->bad = (int)i;
    }
    s->r = revcomp_k(s->f, k);                                    /* ModelCanonical::first */
}
static void kroll_next(kroll* s, unsigned char c)
{
    unsigned code = (unsigned)gko_nt_code(c);
    if (!gko_nt_valid(c)) s->bad = (int)s->k - 1; else s->bad--;  /* Model.hpp:754-758 */
    s->f = ((s->f << 2) + code) & s->mask;
    s->r = ((s->r >> 2) + ((gko_u128)(code ^ 2) << (2 * (s->k - 1)))) & s->mask;   /* _revcompTable[c] = comp_NT[c] << shift */
}
static inline int      kroll_valid(const kroll* s) { return s->bad < 0; }
static inline gko_u128 kroll_canon(const kroll* s) { return (s->f < s->r) ? s->f : s->r; }   /* updateChoice: f<r ? f : r */
static inline int      kroll_which(const kroll* s) { return s->f < s->r; }                   /* true = forward strand */

int64_t gko_kmers(const char* seq, uint64_t len, unsigned k,
                  uint64_t* flo, uint64_t* fhi, uint64_t* clo, uint64_t* chi, uint8_t* valid)
{
    if (len < k) return 0;
    int64_t n = (int64_t)(len - k + 1);
    kroll s; kroll_first(&s, seq, k);
    for (int64_t i = 0;; i++) {
        gko_u128 c = kroll_canon(&s);
        if (flo) flo[i] = (uint64_t)s.f;  if (fhi) fhi[i] = (uint64_t)(s.f >> 64);
        if (clo) clo[i] = (uint64_t)c;    if (chi) chi[i] = (uint64_t)(c >> 64);
        if (valid) valid[i] = (uint8_t)kroll_valid(&s);
        if (i + 1 >= n) break;
        kroll_next(&s, (unsigned char)seq[k + i]);
    }
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * A3  ModelMinimizer  (Model.hpp:1012-1064 ctor/LUT, :1220-1251 is_allowed, :957-973 comparator,
 *                      :1107-1139 next, :1254-1287 computeNewMinimizerOriginal)
 * ---------------------------------------------------------------------------------------------- */
static int mmer_allowed(uint32_t mmer, unsigned m, int has_freq)
{
    if (has_freq) return 1;                                   /* every minimizer allowed in frequency order */
    uint64_t mmask_m1 = ((uint64_t)1 << ((m - 2) * 2)) - 1;  /* drops the two first letters */
    uint64_t mask_ma1 = 0x5555555555555555ULL & mmask_m1;
    uint64_t a1 = mmer;
    a1 = ~(a1 | (a1 >> 2));
    a1 = ((a1 >> 1) & a1) & mask_ma1;                        /* an "AA" anywhere but as prefix */
    return a1 == 0;
}

void gko_mmer_lut(unsigned m, int has_freq, uint32_t* lut)
{
    uint64_t n = (uint64_t)1 << (2 * m);
    uint32_t mask = (uint32_t)(n - 1);
    for (uint64_t x = 0; x < n; x++) {
        uint32_t v = (uint32_t)x;
        uint32_t rc = (uint32_t)gko_revcomp64(x, m);
        if (rc < v) v = rc;                                   /* canonical m-mer (ModelCanonical) */
        if (!mmer_allowed(v, m, has_freq)) v = mask;          /* forbidden => 4^m-1 */
        lut[x] = v;
    }
}

/* ComparatorMinimizerFrequencyOrLex::operator() : true iff a is strictly before b */
static inline int mm_less(uint32_t a, uint32_t b, const uint32_t* freq)
{
    if (freq) { if (freq[a] == freq[b]) return a < b; return freq[a] < freq[b]; }
    return a < b;
}

typedef struct {
    kroll kr; unsigned m, nb_mm; uint32_t mmask; const uint32_t* lut; const uint32_t* freq;
    uint32_t minim; int pos;
} mroll;

static void mroll_rescan(mroll* s)          /* computeNewMinimizerOriginal */
{
    uint32_t best = s->mmask;               /* _minimizerDefault = getKmerMax() of the m-mer model */
    int pos = -1;
    gko_u128 val = s->kr.f;                 /* kmer.value(0) : FORWARD strand */
    for (int idx = (int)s->nb_mm - 1; idx >= 0; idx--) {
        uint32_t cand = s->lut[(uint32_t)val & s->mmask];
        if (mm_less(cand, best, s->freq)) { best = cand; pos = idx; }
        val >>= 2;
    }
    s->minim = best; s->pos = pos;
}
static void mroll_first(mroll* s, const char* seq, unsigned k, unsigned m, const uint32_t* lut, const uint32_t* freq)
{
    kroll_first(&s->kr, seq, k);
    s->m = m; s->nb_mm = k - m + 1; s->mmask = (uint32_t)(((uint64_t)1 << (2 * m)) - 1); s->lut = lut; s->freq = freq;
    mroll_rescan(s);
}
static void mroll_next(mroll* s, unsigned char c)   /* ModelMinimizer::next */
{
    kroll_next(&s->kr, c);
    uint32_t mmer = s->lut[(uint32_t)s->kr.f & s->mmask];     /* extract(): last m-mer of the forward strand through the LUT */
    s->pos--;
    if (mm_less(mmer, s->minim, s->freq)) { s->minim = mmer; s->pos = (int)s->nb_mm - 1; }
    else if (s->pos < 0) mroll_rescan(s);
}

int64_t gko_minimizers(const char* seq, uint64_t len, unsigned k, unsigned m,
                       const uint32_t* freq_order, uint32_t* out_min, uint8_t* out_valid)
{
    if (len < k) return 0;
    uint32_t* lut = (uint32_t*)malloc(sizeof(uint32_t) << (2 * m));
    gko_mmer_lut(m, freq_order != NULL, lut);
    int64_t n = (int64_t)(len - k + 1);
    mroll s; mroll_first(&s, seq, k, m, lut, freq_order);
    for (int64_t i = 0;; i++) {
        out_min[i] = s.minim; if (out_valid) out_valid[i] = (uint8_t)kroll_valid(&s.kr);
        if (i + 1 >= n) break;
        mroll_next(&s, (unsigned char)seq[k + i]);
    }
    free(lut);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * A4  Sequence2SuperKmer::operator() + KmerFunctor  (Sequence2SuperKmer.hpp:81-159)
 * ---------------------------------------------------------------------------------------------- */
#define GKO_DEFAULT_MINIMIZER 1000000000ULL

typedef void (*sk_cb)(void* ctx, uint64_t minimizer, uint64_t first_kmer_idx, unsigned nbk);

static int default_maxs(unsigned k)
{
    int bits = (k <= 31) ? 64 : 128;          /* Type::getSize() of LargeInt<1>/<2> */
    int v = (bits - 8) / 2;
    return v < 255 ? v : 255;
}

/* returns number of super-k-mers; calls cb for each one (in sequence order) */
static uint64_t split_superkmers(const char* seq, uint64_t len, unsigned k, unsigned m,
                                 const uint32_t* lut, const uint32_t* freq, int maxs,
                                 sk_cb cb, void* ctx, uint64_t* n_valid, uint64_t* n_invalid)
{
    if (len < k) return 0;
    if (maxs <= 0) maxs = default_maxs(k);
    int64_t n = (int64_t)(len - k + 1);
    uint64_t nsk = 0;
    uint64_t sk_min = GKO_DEFAULT_MINIMIZER, sk_first = 0; unsigned sk_size = 0;
    mroll s; mroll_first(&s, seq, k, m, lut, freq);
    for (int64_t i = 0;; i++) {
        if (!kroll_valid(&s.kr)) {
            /* invalid k-mer: flush the pending super-k-mer, restart "from new" */
            if (sk_size > 0 && sk_min != GKO_DEFAULT_MINIMIZER) { cb(ctx, sk_min, sk_first, sk_size); nsk++; }
            sk_size = 0; sk_min = GKO_DEFAULT_MINIMIZER;
            if (n_invalid) (*n_invalid)++;
        } else {
            if (n_valid) (*n_valid)++;
            uint64_t h = s.minim;
            if (sk_min == GKO_DEFAULT_MINIMIZER) sk_min = h;
            if (h != sk_min || sk_size >= (unsigned)maxs) {
                if (sk_size > 0) { cb(ctx, sk_min, sk_first, sk_size); nsk++; }
                sk_size = 0;
            }
            sk_min = h;
            if (sk_size == 0) sk_first = (uint64_t)i;
            sk_size++;
        }
        if (i + 1 >= n) break;
        mroll_next(&s, (unsigned char)seq[k + i]);
    }
    if (sk_size > 0 && sk_min != GKO_DEFAULT_MINIMIZER) { cb(ctx, sk_min, sk_first, sk_size); nsk++; }   /* "output last superK" */
    return nsk;
}

typedef struct { uint32_t *mn, *st, *nb; uint64_t cap, n; } sk_collect;
static void sk_collect_cb(void* c, uint64_t mn, uint64_t first, unsigned nbk)
{
    sk_collect* s = (sk_collect*)c;
    if (s->n < s->cap) { s->mn[s->n] = (uint32_t)mn; s->st[s->n] = (uint32_t)first; s->nb[s->n] = nbk; }
    s->n++;
}
int64_t gko_superkmers(const char* seq, uint64_t len, unsigned k, unsigned m, const uint32_t* freq_order,
                       int maxs, uint32_t* sk_minimizer, uint32_t* sk_start, uint32_t* sk_nbk, uint64_t cap,
                       uint64_t* n_valid, uint64_t* n_invalid)
{
    uint32_t* lut = (uint32_t*)malloc(sizeof(uint32_t) << (2 * m));
    gko_mmer_lut(m, freq_order != NULL, lut);
    sk_collect c = { sk_minimizer, sk_start, sk_nbk, cap, 0 };
    uint64_t v = 0, iv = 0;
    split_superkmers(seq, len, k, m, lut, freq_order, maxs, sk_collect_cb, &c, &v, &iv);
    if (n_valid) *n_valid = v;  if (n_invalid) *n_invalid = iv;
    free(lut);
    return (int64_t)c.n;
}

/* ------------------------------------------------------------------------------------------------
 * A6  SuperKmer::save(CacheSuperKmerBinFiles&)  (Model.hpp:1386-1471) + insertSuperkmer (Storage.cpp:567-580)
 *     record = [u8 nbK][ first k-mer forward value, little-endian bytes, 4 nt per byte;
 *                        the partial last byte is completed, then new bytes filled, LSB first,
 *                        with the last nucleotide of each following k-mer ]
 * ---------------------------------------------------------------------------------------------- */
size_t gko_superkmer_encode(const char* seq, unsigned k, unsigned nbk, uint8_t* out)
{
    size_t o = 0;
    out[o++] = (uint8_t)nbk;
    gko_u128 base = 0;
    for (unsigned i = 0; i < k; i++) base = (base << 2) + (unsigned)gko_nt_code((unsigned char)seq[i]);
    int rem = (int)k;
    while (rem >= 4) { out[o++] = (uint8_t)(base & 255); base >>= 8; rem -= 4; }
    uint8_t nb = (uint8_t)(base & 255);
    int uid = rem;                          /* nucleotides already used in nb */
    unsigned skid = 1;
    for (;;) {
        while (uid < 4 && skid < nbk) {
            uint8_t nt = (uint8_t)gko_nt_code((unsigned char)seq[k - 1 + skid]);   /* last nt of k-mer #skid */
            nb |= (uint8_t)(nt << (uid * 2));
            uid++; skid++;
        }
        if (uid > 0) out[o++] = nb;
        if (skid >= nbk) break;
        nb = 0; uid = 0;
    }
    return o;
}

/* ------------------------------------------------------------------------------------------------
 * B1  ReadSuperKCommand::execute decode loop  (PartitionsCommand.cpp:944-1128), without the kx-mer
 *     packing (a CPU-side sort trick that does not change the emitted (k-mer,count) stream).
 * ---------------------------------------------------------------------------------------------- */
size_t gko_superkmer_decode(const uint8_t* rec, unsigned k, uint64_t* clo, uint64_t* chi, unsigned* nbk_out)
{
    const uint8_t* p = rec;
    unsigned nbk = *p++;
    gko_u128 mask = kmask(k);
    gko_u128 seed = 0; int rem = (int)k; int nbr = 0; uint8_t nb = 0;
    while (rem >= 4) { nb = *p++; seed |= ((gko_u128)nb) << (8 * nbr); rem -= 4; nbr++; }
    int uid = 4;
    if (rem > 0) { nb = *p++; seed |= ((gko_u128)nb) << (8 * nbr); uid = rem; }
    seed &= mask;
    gko_u128 t = seed, rv = revcomp_k(t, k);
    unsigned left = nbk;
    for (unsigned i = 0; i < nbk; i++, left--) {
        gko_u128 mk = (t < rv) ? t : rv;
        if (clo) clo[i] = (uint64_t)mk;  if (chi) chi[i] = (uint64_t)(mk >> 64);
        if (left < 2) break;
        if (uid >= 4) { nb = *p++; uid = 0; }
        unsigned nt = (nb >> (2 * uid)) & 3; uid++;
        t = ((t << 2) | nt) & mask;
        rv = ((rv >> 2) | ((gko_u128)(nt ^ 2) << (2 * (k - 1)))) & mask;
    }
    if (nbk_out) *nbk_out = nbk;
    return (size_t)(p - rec);
}

/* ------------------------------------------------------------------------------------------------
 * Repartitor  (PartiInfo.cpp:48-218) and minimizer frequencies (RepartitionAlgorithm.cpp:311-384)
 * ---------------------------------------------------------------------------------------------- */
void gko_count_mmers(const char* seq, uint64_t len, unsigned m, uint32_t* counts)   /* MmersFrequency (RepartitionAlgorithm.cpp:88-120) */
{
    if (len < m) return;
    kroll s; kroll_first(&s, seq, m);
    int64_t n = (int64_t)(len - m + 1);
    for (int64_t i = 0;; i++) {
        if (kroll_valid(&s)) counts[(uint32_t)kroll_canon(&s)]++;
        if (i + 1 >= n) break;
        kroll_next(&s, (unsigned char)seq[m + i]);
    }
}

typedef struct { uint32_t cnt, idx; } cpair;
static int cpair_cmp(const void* a, const void* b)
{
    const cpair* x = (const cpair*)a; const cpair* y = (const cpair*)b;
    if (x->cnt != y->cnt) return x->cnt < y->cnt ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);          /* std::sort on pair<int,int>: lexicographic */
}
static cpair* sorted_counts(unsigned m, const uint32_t* counts, uint64_t* n_out)
{
    uint64_t rg = (uint64_t)1 << (2 * m), n = 0;
    cpair* v = (cpair*)malloc(sizeof(cpair) * (rg ? rg : 1));
    for (uint64_t i = 0; i < rg; i++) if (counts[i] > 0) { v[n].cnt = counts[i]; v[n].idx = (uint32_t)i; n++; }
    qsort(v, n, sizeof(cpair), cpair_cmp);
    *n_out = n; return v;
}
void gko_freq_order_from_counts(unsigned m, const uint32_t* counts, uint32_t* freq_order)
{
    uint64_t rg = (uint64_t)1 << (2 * m), n;
    cpair* v = sorted_counts(m, counts, &n);
    for (uint64_t i = 0; i < rg; i++) freq_order[i] = (uint32_t)rg;   /* unseen => "not a minimizer" */
    for (uint64_t i = 0; i < n; i++) freq_order[v[i].idx] = (uint32_t)i;
    freq_order[rg - 1] = (uint32_t)(rg - 1);                          /* the default/largest minimizer keeps the largest rank */
    free(v);
}

typedef struct { uint64_t size, idx; } bpair;
static int bpair_cmp_desc(const void* a, const void* b)
{
    const bpair* x = (const bpair*)a; const bpair* y = (const bpair*)b;
    if (x->size != y->size) return x->size > y->size ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx);   /* NOTE: the reference's std::sort leaves ties unspecified */
}
void gko_repart_compute_distrib(unsigned m, uint32_t nb_part, const uint64_t* kx, uint16_t* table)
{
    /* largest bin into the emptiest partition (PartiInfo.cpp:48-106); ties -> lowest partition index
     * (the reference's priority_queue tie order is unspecified) */
    uint64_t rg = (uint64_t)1 << (2 * m);
    bpair* bins = (bpair*)malloc(sizeof(bpair) * rg);
    for (uint64_t i = 0; i < rg; i++) { bins[i].size = kx[i]; bins[i].idx = i; }
    qsort(bins, rg, sizeof(bpair), bpair_cmp_desc);
    uint64_t* used = (uint64_t*)calloc(nb_part, sizeof(uint64_t));
    for (uint64_t c = 0; c < rg; c++) {
        uint32_t best = 0;
        for (uint32_t j = 1; j < nb_part; j++) if (used[j] < used[best]) best = j;
        table[bins[c].idx] = (uint16_t)best;
        used[best] += bins[c].size;
    }
    free(used); free(bins);
}
void gko_repart_just_group_lexi(unsigned m, uint32_t nb_part, const uint64_t* nk, uint16_t* table)
{
    uint64_t rg = (uint64_t)1 << (2 * m), sum = 0;
    for (uint64_t i = 0; i < rg; i++) { table[i] = (uint16_t)(nb_part - 1); sum += nk[i]; }
    uint64_t mean = sum / nb_part, acc = 0, j = 0;
    for (uint64_t i = 0; i < rg; i++) {
        table[i] = (uint16_t)j;
        acc += nk[i];
        if (acc > mean) { acc = 0; if (j < nb_part) j++; }
    }
    /* NOTE (PartiInfo.cpp:206-216): the reference lets j reach nb_part (one past the last partition) when the
     * last group overflows; we clamp on the caller side in tests and never rely on that out-of-range value. */
}
void gko_repart_just_group(unsigned m, uint32_t nb_part, const uint64_t* nk, const uint32_t* counts, uint16_t* table)
{
    uint64_t rg = (uint64_t)1 << (2 * m), sum = 0, n;
    cpair* v = sorted_counts(m, counts, &n);
    for (uint64_t i = 0; i < rg; i++) { table[i] = (uint16_t)(nb_part - 1); sum += nk[i]; }
    uint64_t mean = sum / nb_part, acc = 0, j = 0;
    for (uint64_t i = 0; i < n; i++) {
        table[v[i].idx] = (uint16_t)j;
        acc += nk[v[i].idx];
        if (acc > mean) { acc = 0; if (j < nb_part) j++; }
    }
    free(v);
}

/* ------------------------------------------------------------------------------------------------
 * Whole path: SortingCountAlgorithm::execute  (SortingCountAlgorithm.cpp:636-781)
 *   fillPartitions (:1211-1344) -> per-partition byte streams in the reference wire format
 *   fillSolidKmers (:1384-1602)  -> PartitionsByVectorCommand: decode (B1), sort (B2), run-length count (B3),
 *   processor chain histogram -> solidity(sum) -> dump (B5)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t* data; uint64_t n, cap, n_kmers, n_sk; } bytebuf;
static void bb_push(bytebuf* b, const uint8_t* src, size_t len)
{
    if (b->n + len > b->cap) { b->cap = (b->cap ? b->cap * 2 : 4096); while (b->cap < b->n + len) b->cap *= 2; b->data = (uint8_t*)realloc(b->data, b->cap); }
    memcpy(b->data + b->n, src, len); b->n += len;
}

typedef struct { gko_u128* v; int32_t* a; uint64_t n; uint64_t n_kmers, n_sk; } dataset;

struct gko_dsk {
    unsigned k; uint32_t nb_parts, nb_passes; dataset* ds;
    uint64_t stats[8]; uint64_t* histo; uint32_t histo_max;
};

/* the super-k-mers of one share of the reads are first logged in arrival order ([u16 partition][record in the wire format]) and then laid out
 * partition-major (what the reference's per-partition files are): one growing buffer per share instead of one per partition */
typedef struct {
    const char* seq; unsigned k; uint32_t pass, nb_passes; const uint16_t* repart; bytebuf* log;
    uint64_t* part_bytes; uint64_t* part_nk; uint64_t* part_nsk;
    uint64_t nsk; uint64_t bytes;
    const uint8_t* keep;      /* gko_dsk_run_parts: only super-k-mers of the partitions p with keep[p] != 0 are kept (NULL: all) */
} fill_ctx;

static void fill_cb(void* c, uint64_t mn, uint64_t first, unsigned nbk)   /* FillPartitions::processSuperkmer (:1081-1097) */
{
    fill_ctx* f = (fill_ctx*)c;
    if ((mn % f->nb_passes) != f->pass) return;
    uint32_t p = f->repart[mn];
    if (f->keep && !f->keep[p]) return;                        /* (test infrastructure: a sampled-partition count of a full-size input) */
    uint8_t rec[2 + 1 + 64 + 80];
    rec[0] = (uint8_t)(p & 255); rec[1] = (uint8_t)(p >> 8);
    size_t len = gko_superkmer_encode(f->seq + first, f->k, nbk, rec + 2);
    bb_push(f->log, rec, len + 2);
    f->part_bytes[p] += len; f->part_nk[p] += nbk; f->part_nsk[p]++;
    f->nsk++; f->bytes += len;
}

static int u128_cmp(const void* a, const void* b)
{
    gko_u128 x = *(const gko_u128*)a, y = *(const gko_u128*)b;
    return x < y ? -1 : (x > y);
}
static void sort_u64_radix(uint64_t* a, uint64_t* tmp, uint64_t n, unsigned key_bits)
{
    for (unsigned sh = 0; sh < key_bits; sh += 8) {
        uint64_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        for (uint64_t i = 0; i < n; i++) cnt[((a[i] >> sh) & 255) + 1]++;
        for (int d = 0; d < 256; d++) cnt[d + 1] += cnt[d];
        for (uint64_t i = 0; i < n; i++) tmp[cnt[(a[i] >> sh) & 255]++] = a[i];
        uint64_t* t = a; a = tmp; tmp = t;
    }
    if (((key_bits + 7) / 8) & 1) memcpy(tmp, a, n * sizeof(uint64_t));   /* result back into the caller's array */
}

/* The run is organised the way the reference parallelises it: fillPartitions is data-parallel over the reads (Dispatcher::iterate,
 * SortingCountAlgorithm.cpp:1266-1275: every thread cuts super-k-mers of its share of the reads into per-partition caches), fillSolidKmers is
 * parallel over the partitions (one PartitionsByVectorCommand per partition, nb_partitions_in_parallel at a time, :1456-1587). With one
 * thread (gko_dsk_run) this is the plain sequential restatement; gko_dsk_run_mt runs the same two functions on n_threads pthreads. */
typedef struct {
    const char* bases; const uint64_t* offsets; uint64_t r0, r1;
    unsigned k, m; uint32_t nb_partitions, nb_passes, pass; const uint16_t* repart; const uint32_t* freq_order; const uint32_t* lut; int maxs;
    uint8_t* arena; uint64_t* part_off;   /* [nb_partitions+1] byte offsets of the partitions inside arena: this share's "partition files" */
    uint64_t* part_nk; uint64_t* part_nsk;
    uint64_t stats[8];
    const uint8_t* keep;
} fill_job;

static void fill_range(fill_job* J)
{
    const uint32_t P = J->nb_partitions;
    bytebuf log = { NULL, 0, 0, 0, 0 };
    uint64_t* pb = (uint64_t*)calloc(P, sizeof(uint64_t));
    J->part_nk = (uint64_t*)calloc(P, sizeof(uint64_t)); J->part_nsk = (uint64_t*)calloc(P, sizeof(uint64_t));
    fill_ctx fc = { NULL, J->k, J->pass, J->nb_passes, J->repart, &log, pb, J->part_nk, J->part_nsk, 0, 0, J->keep };
    for (uint64_t r = J->r0; r < J->r1; r++) {
        const char* seq = J->bases + J->offsets[r]; uint64_t len = J->offsets[r + 1] - J->offsets[r];
        fc.seq = seq;
        uint64_t v = 0, iv = 0;
        split_superkmers(seq, len, J->k, J->m, J->lut, J->freq_order, J->maxs, fill_cb, &fc, &v, &iv);
        if (J->pass == 0) {                                    /* bank stats merged only for pass 0 (Sequence2SuperKmer.hpp:183) */
            J->stats[0] += v; J->stats[1] += iv; J->stats[5]++;
            if (len < J->k) J->stats[7]++;
        }
    }
    J->stats[4] += fc.nsk; J->stats[6] += fc.bytes;
    /* partition-major layout */
    J->part_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)P + 1));
    uint64_t run = 0;
    for (uint32_t p = 0; p < P; p++) { J->part_off[p] = run; run += pb[p]; }
    J->part_off[P] = run;
    J->arena = (uint8_t*)malloc(run ? run : 1);
    for (uint32_t p = 0; p < P; p++) pb[p] = J->part_off[p];   /* cursors */
    for (uint64_t off = 0; off < log.n; ) {
        const uint32_t p = (uint32_t)log.data[off] | ((uint32_t)log.data[off + 1] << 8);
        const unsigned nbk = log.data[off + 2];
        const size_t len = 1 + ((size_t)J->k + nbk - 1 + 3) / 4;               /* [u8 nbK][ceil((k+nbK-1)/4) bytes] (Model.hpp:1386-1471) */
        memcpy(J->arena + pb[p], log.data + off + 2, len); pb[p] += len;
        off += 2 + len;
    }
    free(log.data); free(pb);
}

/* one "PartitionsByVectorCommand": partition p of the pass, its records spread over n_src byte buffers (one per fill thread) */
static void count_partition(dataset* D, const fill_job* src, uint32_t n_src, uint32_t p, unsigned k, int32_t amin, int32_t amax, uint32_t histo_max,
                            uint64_t* histo, uint64_t* n_distinct, uint64_t* n_solid)
{
    const int wide = k > 31;
    uint64_t nk = 0;
    for (uint32_t s = 0; s < n_src; s++) { nk += src[s].part_nk[p]; D->n_kmers += src[s].part_nk[p]; D->n_sk += src[s].part_nsk[p]; }
    gko_u128* keys = (gko_u128*)malloc(sizeof(gko_u128) * (nk ? nk : 1));
    uint64_t* lo = (uint64_t*)malloc(sizeof(uint64_t) * 256), *hi = (uint64_t*)malloc(sizeof(uint64_t) * 256);
    uint64_t w = 0;
    for (uint32_t s = 0; s < n_src; s++) {                     /* executeRead */
        const uint8_t* data = src[s].arena + src[s].part_off[p]; const uint64_t nb = src[s].part_off[p + 1] - src[s].part_off[p]; uint64_t off = 0;
        while (off < nb) {
            unsigned nbk;
            off += gko_superkmer_decode(data + off, k, lo, hi, &nbk);
            for (unsigned i = 0; i < nbk; i++) keys[w++] = ((gko_u128)hi[i] << 64) | lo[i];
        }
    }
    free(lo); free(hi);
    if (!wide) {                                               /* executeSort (ascending Type order) */
        uint64_t* a = (uint64_t*)malloc(sizeof(uint64_t) * (nk ? nk : 1)), *t = (uint64_t*)malloc(sizeof(uint64_t) * (nk ? nk : 1));
        for (uint64_t i = 0; i < nk; i++) a[i] = (uint64_t)keys[i];
        sort_u64_radix(a, t, nk, 2 * k);
        for (uint64_t i = 0; i < nk; i++) keys[i] = a[i];
        free(a); free(t);
    } else qsort(keys, nk, sizeof(gko_u128), u128_cmp);
    /* executeDump: run-length count, then CountProcessorChain::process (histogram, solidity sum, dump) */
    D->v = (gko_u128*)malloc(sizeof(gko_u128) * (nk ? nk : 1));
    D->a = (int32_t*)malloc(sizeof(int32_t) * (nk ? nk : 1));
    uint64_t i = 0;
    while (i < nk) {
        uint64_t j = i + 1; while (j < nk && keys[j] == keys[i]) j++;
        int32_t cnt = (int32_t)(j - i);                        /* CountNumber is int32 (system/api/types.hpp:49) */
        (*n_distinct)++;
        histo[(uint32_t)cnt >= histo_max ? histo_max : (uint32_t)cnt]++;       /* Histogram::inc (Histogram.hpp:92) */
        if (cnt >= amin && cnt <= amax) {                      /* CountRange::includes, closed interval */
            D->v[D->n] = keys[i]; D->a[D->n] = cnt; D->n++; (*n_solid)++;
        }
        i = j;
    }
    free(keys);
}

#include <pthread.h>
#include <time.h>
#include <stdio.h>
typedef struct {
    fill_job* fills; uint32_t n_fill; uint32_t* next_fill;    /* phase 1: shares of the reads */
    gko_dsk* R; const fill_job* src; uint32_t n_src; uint32_t pass; uint32_t* next_part; int32_t amin, amax;   /* phase 2: partitions */
    uint64_t* histo; uint64_t n_distinct, n_solid;            /* private, merged by the caller */
    int phase;
} mt_worker;
static void* mt_main(void* arg)
{
    mt_worker* W = (mt_worker*)arg;
    if (W->phase == 1) {
        for (;;) { const uint32_t i = __atomic_fetch_add(W->next_fill, 1, __ATOMIC_RELAXED); if (i >= W->n_fill) break; fill_range(&W->fills[i]); }
    } else {
        gko_dsk* R = W->R;
        for (;;) {
            const uint32_t p = __atomic_fetch_add(W->next_part, 1, __ATOMIC_RELAXED);
            if (p >= R->nb_parts) break;
            count_partition(&R->ds[p + W->pass * R->nb_parts], W->src, W->n_src, p, R->k, W->amin, W->amax, R->histo_max, W->histo, &W->n_distinct, &W->n_solid);
        }
    }
    return NULL;
}

/* The same run restricted to the partitions p with keep[p] != 0 (keep == NULL: every partition): the super-k-mers of the other partitions are dropped where
 * FillPartitions::processSuperkmer would append them (SortingCountAlgorithm.cpp:1081-1097), so a FULL-SIZE input can be counted exactly for a few sampled partitions in
 * the time of its fillPartitions step (the shape of TestDSK.cpp:254-305: the solid k-mers of a run compared with an independent count). The kept partitions' datasets,
 * n_kmers and n_sk are those of the unrestricted run; valid / invalid k-mers and sequences are still those of the whole input, distinct / solid / histogram / super-k-mer
 * totals cover the kept partitions only. */
gko_dsk* gko_dsk_run_parts(const char* bases, const uint64_t* offsets, uint64_t n_reads,
                           unsigned k, unsigned m, uint32_t nb_partitions, uint32_t nb_passes,
                           const uint16_t* repart, const uint32_t* freq_order,
                           int32_t amin, int32_t amax, uint32_t histo_max, int maxs, uint32_t n_threads, const uint8_t* keep)
{
    if (n_threads < 1) n_threads = 1;
    gko_dsk* R = (gko_dsk*)calloc(1, sizeof(gko_dsk));
    R->k = k; R->nb_parts = nb_partitions; R->nb_passes = nb_passes; R->histo_max = histo_max;
    R->ds = (dataset*)calloc((size_t)nb_partitions * nb_passes, sizeof(dataset));
    R->histo = (uint64_t*)calloc((size_t)histo_max + 1, sizeof(uint64_t));
    uint32_t* lut = (uint32_t*)malloc(sizeof(uint32_t) << (2 * m));
    gko_mmer_lut(m, freq_order != NULL, lut);
    /* shares of the reads: a few per thread, so that threads finish together */
    const uint32_t n_fill = n_threads == 1 ? 1 : (uint32_t)(n_reads < 4ull * n_threads ? (n_reads ? n_reads : 1) : 4ull * n_threads);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n_threads);
    mt_worker* W = (mt_worker*)calloc(n_threads, sizeof(mt_worker));
    for (uint32_t t = 0; t < n_threads; t++) W[t].histo = (uint64_t*)calloc((size_t)histo_max + 1, sizeof(uint64_t));

    for (uint32_t pass = 0; pass < nb_passes; pass++) {
        /* ---- fillPartitions ---- */
        fill_job* fills = (fill_job*)calloc(n_fill, sizeof(fill_job));
        for (uint32_t i = 0; i < n_fill; i++) {
            fill_job* J = &fills[i];
            J->bases = bases; J->offsets = offsets; J->r0 = n_reads * i / n_fill; J->r1 = n_reads * (i + 1) / n_fill;
            J->k = k; J->m = m; J->nb_partitions = nb_partitions; J->nb_passes = nb_passes; J->pass = pass; J->repart = repart;
            J->freq_order = freq_order; J->lut = lut; J->maxs = maxs; J->keep = keep;
        }
        uint32_t next = 0;
        for (uint32_t t = 0; t < n_threads; t++) { W[t].phase = 1; W[t].fills = fills; W[t].n_fill = n_fill; W[t].next_fill = &next; }
        if (n_threads == 1) mt_main(&W[0]);
        else { for (uint32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, mt_main, &W[t]); for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL); }
        for (uint32_t i = 0; i < n_fill; i++) for (int q = 0; q < 8; q++) R->stats[q] += fills[i].stats[q];
        if (getenv("GKO_TIMING")) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[gko] pass %u fill done at %.3f\n", pass, ts.tv_sec + ts.tv_nsec * 1e-9); }

        /* ---- fillSolidKmers: partitions dealt to the threads ---- */
        uint32_t next_p = 0;
        for (uint32_t t = 0; t < n_threads; t++) { W[t].phase = 2; W[t].R = R; W[t].src = fills; W[t].n_src = n_fill; W[t].pass = pass; W[t].next_part = &next_p; W[t].amin = amin; W[t].amax = amax; }
        if (n_threads == 1) mt_main(&W[0]);
        else { for (uint32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, mt_main, &W[t]); for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL); }
        if (getenv("GKO_TIMING")) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[gko] pass %u count done at %.3f\n", pass, ts.tv_sec + ts.tv_nsec * 1e-9); }
        for (uint32_t i = 0; i < n_fill; i++) { free(fills[i].arena); free(fills[i].part_off); free(fills[i].part_nk); free(fills[i].part_nsk); }
        free(fills);
    }
    for (uint32_t t = 0; t < n_threads; t++) {
        R->stats[2] += W[t].n_distinct; R->stats[3] += W[t].n_solid;
        for (uint32_t h = 0; h <= histo_max; h++) R->histo[h] += W[t].histo[h];
        free(W[t].histo);
    }
    free(W); free(th); free(lut);
    return R;
}

gko_dsk* gko_dsk_run_mt(const char* bases, const uint64_t* offsets, uint64_t n_reads,
                        unsigned k, unsigned m, uint32_t nb_partitions, uint32_t nb_passes,
                        const uint16_t* repart, const uint32_t* freq_order,
                        int32_t amin, int32_t amax, uint32_t histo_max, int maxs, uint32_t n_threads)
{
    return gko_dsk_run_parts(bases, offsets, n_reads, k, m, nb_partitions, nb_passes, repart, freq_order, amin, amax, histo_max, maxs, n_threads, NULL);
}

gko_dsk* gko_dsk_run(const char* bases, const uint64_t* offsets, uint64_t n_reads,
                     unsigned k, unsigned m, uint32_t nb_partitions, uint32_t nb_passes,
                     const uint16_t* repart, const uint32_t* freq_order,
                     int32_t amin, int32_t amax, uint32_t histo_max, int maxs)
{
    return gko_dsk_run_mt(bases, offsets, n_reads, k, m, nb_partitions, nb_passes, repart, freq_order, amin, amax, histo_max, maxs, 1);
}

void gko_dsk_free(gko_dsk* R)
{
    if (!R) return;
    for (uint64_t i = 0; i < (uint64_t)R->nb_parts * R->nb_passes; i++) { free(R->ds[i].v); free(R->ds[i].a); }
    free(R->ds); free(R->histo); free(R);
}
uint64_t gko_dsk_part_size(const gko_dsk* R, uint32_t d) { return R->ds[d].n; }
void gko_dsk_part_copy(const gko_dsk* R, uint32_t d, uint64_t* lo, uint64_t* hi, int32_t* ab)
{
    const dataset* D = &R->ds[d];
    for (uint64_t i = 0; i < D->n; i++) { if (lo) lo[i] = (uint64_t)D->v[i]; if (hi) hi[i] = (uint64_t)(D->v[i] >> 64); if (ab) ab[i] = D->a[i]; }
}
void gko_dsk_part_copy_records(const gko_dsk* R, uint32_t d, void* out)
{
    const dataset* D = &R->ds[d];
    if (R->k <= 31) {
        uint8_t* o = (uint8_t*)out;
        for (uint64_t i = 0; i < D->n; i++, o += 16) { uint64_t v = (uint64_t)D->v[i]; memset(o, 0, 16); memcpy(o, &v, 8); memcpy(o + 8, &D->a[i], 4); }
    } else {
        uint8_t* o = (uint8_t*)out;
        for (uint64_t i = 0; i < D->n; i++, o += 32) { memset(o, 0, 32); memcpy(o, &D->v[i], 16); memcpy(o + 16, &D->a[i], 4); }
    }
}
void gko_dsk_stats(const gko_dsk* R, uint64_t s[8]) { memcpy(s, R->stats, sizeof(R->stats)); }
void gko_dsk_histogram(const gko_dsk* R, uint64_t* h) { memcpy(h, R->histo, sizeof(uint64_t) * ((size_t)R->histo_max + 1)); }
void gko_dsk_part_stats(const gko_dsk* R, uint32_t d, uint64_t* nk, uint64_t* nsk) { *nk = R->ds[d].n_kmers; *nsk = R->ds[d].n_sk; }

/* ------------------------------------------------------------------------------------------------
 * C1-C4  Bloom filters  (tools/collections/impl/Bloom.hpp)
 *   HashFunctors (:59-98), BloomContainer (:177-263), Bloom/BloomSynchronized insert (:270-412),
 *   BloomCacheCoherent (:429-502), BloomNeighborCoherent (:514-828), bit_mask (Bloom.cpp)
 * ---------------------------------------------------------------------------------------------- */
struct gko_bloom {
    int kind; unsigned nb_hash, k; int wide;
    uint64_t tai, nchar, reduced_tai; int pow2; uint8_t* a; uint64_t seeds[10];
};

void gko_bloom_seeds(uint64_t user_seed, uint64_t s[10])
{
    static const uint64_t rbase[10] = {
        0xAAAAAAAA55555555ULL, 0x33333333CCCCCCCCULL, 0x6666666699999999ULL, 0xB5B5B5B54B4B4B4BULL,
        0xAA55AA5555335533ULL, 0x33CC33CCCC66CC66ULL, 0x6699669999B599B5ULL, 0xB54BB54B4BAA4BAAULL,
        0xAA33AA3355CC55CCULL, 0x33663366CC99CC99ULL };
    for (int i = 0; i < 10; i++) s[i] = rbase[i];
    for (int i = 0; i < 10; i++) s[i] = s[i] * s[(i + 3) % 10] + user_seed;   /* in place, sequential */
}

gko_bloom* gko_bloom_create(int kind, uint64_t tai_bits, unsigned nb_hash, unsigned k)
{
    gko_bloom* b = (gko_bloom*)calloc(1, sizeof(gko_bloom));
    b->kind = kind; b->nb_hash = nb_hash; b->k = k; b->wide = k > 31;
    uint64_t tai = tai_bits;
    if (kind != GKO_BLOOM_BASIC) tai += 2 * 4096;           /* BloomCacheCoherent: tai_bloom + 2*(1<<12) */
    b->nchar = 1 + tai / 8;
    b->a = (uint8_t*)calloc(b->nchar, 1);
    b->pow2 = (tai && !(tai & (tai - 1)));
    if (b->pow2) tai--;                                      /* a % 2^N <=> a & (2^N-1) */
    b->tai = tai;
    b->reduced_tai = (kind != GKO_BLOOM_BASIC) ? tai - 2 * 4096 : tai;
    gko_bloom_seeds(0, b->seeds);
    return b;
}
void     gko_bloom_free(gko_bloom* b) { if (b) { free(b->a); free(b); } }
uint64_t gko_bloom_nbytes(const gko_bloom* b) { return b->nchar; }
uint64_t gko_bloom_bitsize(const gko_bloom* b) { return b->kind == GKO_BLOOM_BASIC ? b->tai : b->reduced_tai; }
uint8_t* gko_bloom_array(gko_bloom* b) { return b->a; }

static inline void bset(gko_bloom* b, uint64_t h) { b->a[h >> 3] |= (uint8_t)(1u << (h & 7)); }
static inline int  bget(const gko_bloom* b, uint64_t h) { return (b->a[h >> 3] >> (h & 7)) & 1; }

static const unsigned CANO2[16] = { 0, 1, 2, 3, 4, 5, 3, 7, 8, 9, 0, 4, 9, 13, 1, 5 };

/* positions of the nb_hash bits of item x; returns count */
static unsigned bloom_positions(const gko_bloom* b, gko_u128 x, uint64_t* pos)
{
    unsigned n = 0;
    if (b->kind == GKO_BLOOM_BASIC) {
        for (unsigned i = 0; i < b->nb_hash; i++) {
            uint64_t h = hash1_k(x, b->seeds[i], b->wide);
            pos[n++] = b->pow2 ? (h & b->tai) : (h % b->tai);
        }
    } else if (b->kind == GKO_BLOOM_CACHE) {
        uint64_t h0 = hash1_k(x, b->seeds[0], b->wide) % b->reduced_tai;
        pos[n++] = h0;
        for (unsigned i = 1; i < b->nb_hash; i++) pos[n++] = h0 + (simplehash16_k(x, (int)i, b->wide) & 4095);
    } else {
        unsigned k = b->k;
        unsigned suffix = (unsigned)(x & 3);
        unsigned prefix = (unsigned)((x >> (2 * (k - 1))) & 3) << 2;        /* (item & _prefmask) >> ((k-2)*2) */
        unsigned pv = CANO2[(prefix + suffix) & 15];
        gko_u128 hp = (x >> 2) & kmask(k - 2);
        gko_u128 rv = revcomp_k(hp, k - 2);
        /* NOTE: for a 128-bit Item the (k-2)-mer revcomp uses the LargeInt<2> routine even when k-2<=32 */
        if (b->wide) rv = revcomp128(hp, k - 2);
        if (rv < hp) hp = rv;
        uint64_t h0 = hash1_k(hp, b->seeds[0], b->wide) % b->reduced_tai + pv;
        pos[n++] = h0;
        for (unsigned i = 1; i < b->nb_hash; i++) pos[n++] = h0 + (simplehash16_k(hp, (int)i, b->wide) & 4095);
    }
    return n;
}

void gko_bloom_insert(gko_bloom* b, const uint64_t* lo, const uint64_t* hi, uint64_t n)
{
    uint64_t pos[32];
    for (uint64_t i = 0; i < n; i++) {
        gko_u128 x = ((gko_u128)(hi ? hi[i] : 0) << 64) | lo[i];
        unsigned c = bloom_positions(b, x, pos);
        for (unsigned j = 0; j < c; j++) bset(b, pos[j]);
    }
}
static int bloom_contains1(const gko_bloom* b, gko_u128 x)
{
    uint64_t pos[32];
    unsigned c = bloom_positions(b, x, pos);
    for (unsigned j = 0; j < c; j++) if (!bget(b, pos[j])) return 0;
    return 1;
}
void gko_bloom_contains(const gko_bloom* b, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = (uint8_t)bloom_contains1(b, ((gko_u128)(hi ? hi[i] : 0) << 64) | lo[i]);
}
/* contains4/contains8 (Bloom.hpp:645-811): membership of the 4 right (x<<2|j) and 4 left (x>>2 | j<<2(k-1))
 * raw-orientation neighbours; they share the canonical (k-2)-mer core, hence one hash1 */
void gko_bloom_contains8(const gko_bloom* b, const uint64_t* lo, const uint64_t* hi, uint64_t n, uint8_t* out)
{
    unsigned k = b->k; gko_u128 mask = kmask(k);
    for (uint64_t i = 0; i < n; i++) {
        gko_u128 x = ((gko_u128)(hi ? hi[i] : 0) << 64) | lo[i];
        uint8_t r = 0;
        for (unsigned j = 0; j < 4; j++) {
            gko_u128 right = ((x << 2) & mask) + j;
            gko_u128 left = (x >> 2) + ((gko_u128)j << (2 * (k - 1)));
            if (bloom_contains1(b, right)) r |= (uint8_t)(1u << j);
            if (bloom_contains1(b, left)) r |= (uint8_t)(1u << (4 + j));
        }
        out[i] = r;
    }
}

/* =====================================================================================================================
 * bank: FASTA / FASTQ reader (bank/impl/BankFasta.cpp:391-571), restated over a memory buffer.
 * ===================================================================================================================== */
typedef struct { const char* p; uint64_t n, pos; } fx_in;
static int fx_getc(fx_in* f) { return f->pos < f->n ? (int)(signed char)f->p[f->pos++] : -1; }           /* buffered_getc :413 */
/* buffered_gets (:425-483) with allow_spaces: appends up to (not including) the next '\n' to dst; afterwards drops one trailing '\r'
 * if the whole accumulated string is longer than 1. Returns -1 at end of input (nothing left), else the accumulated length. */
static int64_t fx_gets_line(fx_in* f, char* dst, uint64_t* len, uint64_t cap, int* overflow)
{
    if (f->pos >= f->n) return -1;
    uint64_t i = f->pos;
    while (i < f->n && f->p[i] != '\n') i++;
    uint64_t add = i - f->pos;
    if (dst) { if (*len + add > cap) { *overflow = 1; add = cap - *len; } memcpy(dst + *len, f->p + f->pos, add); }
    *len += add;
    f->pos = (i < f->n) ? i + 1 : i;
    if (*len > 1 && dst && dst[*len - 1] == '\r') (*len)--;
    return (int64_t)*len;
}
int64_t gko_fastx_parse(const char* text, uint64_t n, char* out_data, uint64_t cap_data, uint64_t* out_offsets, uint64_t cap_seq)
{
    fx_in f = { text, n, 0 };
    int last_char = 0, c, overflow = 0;
    uint64_t n_seq = 0, total = 0;
    char* qual = (char*)malloc(n + 2);
    if (!qual) return -1;
    for (;;) {                                                                     /* one iteration = one get_next_seq_from_file call */
        if (last_char == 0) {                                                     /* :496-502 go to next header */
            while ((c = fx_getc(&f)) != -1 && c != '>' && c != '@') ;
            if (c == -1) break;                                                    /* return false */
            last_char = c;
        }
        if (f.pos >= f.n) break;                                                   /* buffered_gets(header) < 0 -> return false (:505) */
        {   /* header: first token up to a whitespace (:505), then the rest of the line (:508-525); not part of the output here */
            uint64_t i = f.pos; while (i < f.n && !isspace((unsigned char)f.p[i])) i++;
            const int dret = i < f.n ? f.p[i] : 0;
            f.pos = i < f.n ? i + 1 : i;
            if (dret != '\n') { uint64_t dummy = 0; (void)fx_gets_line(&f, NULL, &dummy, 0, &overflow); }
        }
        if (n_seq + 1 > cap_seq) { free(qual); return -1; }
        out_offsets[n_seq] = total;
        char* rd = out_data + total; uint64_t rlen = 0; const uint64_t rcap = cap_data - total;
        while ((c = fx_getc(&f)) != -1 && c != '>' && c != '+' && c != '@') {    /* :532-537 */
            if (c == '\n') continue;                                               /* empty line */
            if (rlen + 1 > rcap) { overflow = 1; break; }
            rd[rlen++] = (char)c;
            (void)fx_gets_line(&f, rd, &rlen, rcap, &overflow);
        }
        if (overflow) { free(qual); return -1; }
        if (c == '>' || c == '@') last_char = c;                                   /* :538 */
        if (c == '+') {                                                            /* :546-560 fastq */
            while ((c = fx_getc(&f)) != -1 && c != '\n') ;                         /* rest of the '+' line */
            uint64_t qlen = 0; int ov2 = 0;
            while (fx_gets_line(&f, qual, &qlen, n + 1, &ov2) >= 0 && qlen < rlen) ;   /* quality, consumed by length */
            last_char = 0;
        }
        total += rlen; n_seq++;                                                    /* return true */
    }
    free(qual);
    out_offsets[n_seq] = total;
    return (int64_t)n_seq;
}

/* =====================================================================================================================
 * Histogram::compute_threshold (tools/misc/impl/Histogram.cpp:61-190)
 * ===================================================================================================================== */
void gko_histogram_cutoff(const uint64_t* h, uint64_t L, int min_auto_threshold, uint64_t out[3])
{
    uint64_t* sm = (uint64_t*)calloc(L + 2, 8);
    uint64_t sum_allk = 0, cutoff = 0, nbsolids = 0, first_peak = 0;
    if (L >= 2) { sm[1] = (uint64_t)(0.6 * (double)h[1] + 0.4 * (double)h[2]); sum_allk += h[1]; }                 /* :66-71 */
    int64_t first_inc = -1, idx_max = -1; uint64_t max_val = 0;
    for (uint64_t i = 2; i < L; i++) {                                                                              /* :78-98 */
        sum_allk += h[i] * i;
        sm[i] = (uint64_t)(0.2 * (double)h[i - 1] + 0.6 * (double)h[i] + 0.2 * (double)h[i + 1]);
        if (first_inc == -1 && sm[i - 1] < sm[i]) first_inc = (int64_t)i - 1;
        if (first_inc > 0 && sm[i] > max_val) { max_val = sm[i]; idx_max = (int64_t)i; }
    }
    sum_allk += h[L] * L;                                                                                           /* :100 */
    if (first_inc == -1) { out[0] = (uint64_t)min_auto_threshold; out[1] = 0; out[2] = 0; free(sm); return; }       /* :102-106 */
    first_peak = (uint64_t)idx_max;
    uint64_t min_val = 10000000000ULL; int64_t idx_min = -1;
    for (int64_t i = first_inc; i <= idx_max; i++) if (sm[i] < min_val) { min_val = sm[i]; idx_min = i; }           /* :116-123 */
    if (idx_min != -1) cutoff = (uint64_t)idx_min;
    uint64_t sum_elim = 0, max_cutoff = 0;
    for (uint64_t i = 0; i < L + 1; i++) {                                                                          /* :131-143 */
        sum_elim += h[i] * i;
        if ((double)sum_elim / sum_allk >= 0.25) { max_cutoff = i + 1; break; }
    }
    if (cutoff > max_cutoff) cutoff = max_cutoff;
    if (cutoff < (uint64_t)min_auto_threshold) cutoff = (uint64_t)min_auto_threshold;
    for (uint64_t i = cutoff; i < L + 1; i++) nbsolids += h[i];                                                     /* :168-172 */
    out[0] = cutoff; out[1] = nbsolids; out[2] = first_peak;
    free(sm);
}

/* =====================================================================================================================
 * MPHF (BooPHF) — thirdparty/BooPHF/BooPHF.h, tools/collections/impl/BooPHF.hpp
 * ===================================================================================================================== */
#include <math.h>
#define MPHF_NB_LEVELS 25                                         /* BooPHF.h:1029 */
#define MPHF_SEED 18006821046139946489ULL                         /* std::mt19937_64 rng(37); rng()  (BooPHF.hpp:246-249) */
struct gko_mphf {
    double gamma; uint64_t nelem, lastbitsetrank; int wide;
    uint64_t domain[MPHF_NB_LEVELS]; uint64_t nchar[MPHF_NB_LEVELS]; uint64_t* bits[MPHF_NB_LEVELS];
    uint64_t nranks[MPHF_NB_LEVELS]; uint64_t* ranks[MPHF_NB_LEVELS];
    uint64_t nfinal; uint64_t* final_lo; uint64_t* final_hi; uint64_t* final_val;
};
static void jenkins_mix(uint64_t* a, uint64_t* b, uint64_t* c)     /* BooPHF.hpp:186-201 */
{
    *a -= *b; *a -= *c; *a ^= (*c >> 43);  *b -= *c; *b -= *a; *b ^= (*a << 9);   *c -= *a; *c -= *b; *c ^= (*b >> 8);
    *a -= *b; *a -= *c; *a ^= (*c >> 38);  *b -= *c; *b -= *a; *b ^= (*a << 23);  *c -= *a; *c -= *b; *c ^= (*b >> 5);
    *a -= *b; *a -= *c; *a ^= (*c >> 35);  *b -= *c; *b -= *a; *b ^= (*a << 49);  *c -= *a; *c -= *b; *c ^= (*b >> 11);
    *a -= *b; *a -= *c; *a ^= (*c >> 12);  *b -= *c; *b -= *a; *b ^= (*a << 18);  *c -= *a; *c -= *b; *c ^= (*b >> 22);
}
/* jenkins64_hasher::operator()(byte_range) on the 8 / 16 key bytes (BooPHF.hpp:93-146): the two hashes BooPHF asks for are get<0> and get<2> (:254-258) */
static void mphf_hash_pair(uint64_t lo, uint64_t hi, int wide, uint64_t* h0, uint64_t* h1)
{
    uint64_t a = MPHF_SEED, b = MPHF_SEED, c = 0x9e3779b97f4a7c13ULL;
    c += wide ? 16 : 8;
    if (wide) b += hi;
    a += lo;
    jenkins_mix(&a, &b, &c);
    *h0 = a; *h1 = c;
}
static uint64_t xs_next(uint64_t* s)                                /* BooPHF.h:350-358 */
{
    uint64_t s1 = s[0]; const uint64_t s0 = s[1];
    s[0] = s0; s1 ^= s1 << 23;
    return (s[1] = (s1 ^ s0 ^ (s1 >> 17) ^ (s0 >> 26))) + s0;
}
/* getLevel (BooPHF.h:1062-1095): first level < maxlevel whose bit is set at the key's slot; returns the hash of the last level looked at */
static uint64_t mphf_get_level(const gko_mphf* m, uint64_t lo, uint64_t hi, int maxlevel, int* res_level, uint64_t s[2])
{
    int level = 0; uint64_t hash_raw = 0, h0, h1;
    mphf_hash_pair(lo, hi, m->wide, &h0, &h1);
    for (int ii = 0; ii < MPHF_NB_LEVELS - 1 && ii < maxlevel; ii++) {
        if (ii == 0) { s[0] = h0; hash_raw = h0; } else if (ii == 1) { s[1] = h1; hash_raw = h1; } else hash_raw = xs_next(s);
        const uint64_t pos = hash_raw % m->domain[ii];
        if ((m->bits[ii][pos >> 6] >> (pos & 63)) & 1ULL) break;
        level++;
    }
    *res_level = level;
    return hash_raw;
}
gko_mphf* gko_mphf_build(const void* keys, uint64_t n, uint32_t stride, int wide)
{
    if (n == 0) return NULL;
    gko_mphf* m = (gko_mphf*)calloc(1, sizeof(gko_mphf));
    m->gamma = 3.0; m->nelem = n; m->wide = wide;
    const uint64_t hash_domain = (uint64_t)ceil((double)n * m->gamma);                                            /* :735 */
    const double proba = 1.0 - pow(((m->gamma * (double)n - 1) / (m->gamma * (double)n)), (double)(n - 1));      /* :1024 */
    for (int ii = 0; ii < MPHF_NB_LEVELS; ii++) {                                                                /* :1034-1046 */
        m->domain[ii] = (((uint64_t)(hash_domain * pow(proba, ii)) + 63) / 64) * 64;
        if (m->domain[ii] == 0) m->domain[ii] = 64;
    }
    const uint8_t* kp = (const uint8_t*)keys;
    uint64_t offset = 0; uint64_t cap_final = 16; m->final_lo = malloc(cap_final * 8); m->final_hi = malloc(cap_final * 8); m->final_val = malloc(cap_final * 8);
    for (int i = 0; i < MPHF_NB_LEVELS; i++) {
        m->nchar[i] = 1ULL + m->domain[i] / 64ULL;                                                               /* bitVector ctor :427-431 */
        m->bits[i] = (uint64_t*)calloc(m->nchar[i], 8);
        uint64_t* coll = (uint64_t*)calloc(m->nchar[i], 8);
        uint64_t hashidx = 0;
        for (uint64_t q = 0; q < n; q++) {                                                                       /* processLevel :849-927 */
            uint64_t lo, hi = 0; memcpy(&lo, kp + q * stride, 8); if (wide) memcpy(&hi, kp + q * stride + 8, 8);
            int level; uint64_t s[2] = {0, 0};
            (void)mphf_get_level(m, lo, hi, i, &level, s);
            if (level != i) continue;
            if (i == MPHF_NB_LEVELS - 1) {                                                                       /* exact hash for what is left */
                if (hashidx == cap_final) { cap_final *= 2; m->final_lo = realloc(m->final_lo, cap_final * 8); m->final_hi = realloc(m->final_hi, cap_final * 8); m->final_val = realloc(m->final_val, cap_final * 8); }
                m->final_lo[hashidx] = lo; m->final_hi[hashidx] = hi; m->final_val[hashidx] = hashidx; hashidx++;
            } else {
                uint64_t h0, h1, lh; mphf_hash_pair(lo, hi, wide, &h0, &h1);
                if (level == 0) lh = h0; else if (level == 1) lh = h1; else lh = xs_next(s);                     /* :905-914; s holds the state after level-1 looks */
                const uint64_t pos = lh % m->domain[i];                                                          /* insertIntoLevel :1098-1108 */
                const uint64_t bit = 1ULL << (pos & 63);
                if (m->bits[i][pos >> 6] & bit) coll[pos >> 6] |= bit; else m->bits[i][pos >> 6] |= bit;
            }
        }
        if (i == MPHF_NB_LEVELS - 1) m->nfinal = hashidx;
        for (uint64_t w = 0; w < m->domain[i] / 64; w++) m->bits[i][w] &= ~coll[w];                               /* clearCollisions :511-523 */
        free(coll);
        m->nranks[i] = 0; m->ranks[i] = (uint64_t*)malloc((m->nchar[i] / 8 + 2) * 8);                            /* build_ranks :596-609 */
        uint64_t cur = offset;
        for (uint64_t w = 0; w < m->nchar[i]; w++) {
            if (((w * 64) % 512) == 0) m->ranks[i][m->nranks[i]++] = cur;
            cur += (uint64_t)__builtin_popcountll(m->bits[i][w]);
        }
        offset = cur;
    }
    m->lastbitsetrank = offset;
    return m;
}
void gko_mphf_free(gko_mphf* m)
{
    if (!m) return;
    for (int i = 0; i < MPHF_NB_LEVELS; i++) { free(m->bits[i]); free(m->ranks[i]); }
    free(m->final_lo); free(m->final_hi); free(m->final_val); free(m);
}
uint64_t gko_mphf_lookup(const gko_mphf* m, uint64_t lo, uint64_t hi)                                            /* lookup :787-822 */
{
    int level; uint64_t s[2] = {0, 0};
    const uint64_t level_hash = mphf_get_level(m, lo, hi, 100, &level, s);
    if (level == MPHF_NB_LEVELS - 1) {
        for (uint64_t i = 0; i < m->nfinal; i++) if (m->final_lo[i] == lo && m->final_hi[i] == hi) return m->final_val[i] + m->lastbitsetrank;
        return ~0ULL;
    }
    const uint64_t pos = level_hash % m->domain[level];
    const uint64_t word = pos / 64, block = pos / 512;                                                           /* bitVector::rank :611-624 */
    uint64_t r = m->ranks[level][block];
    for (uint64_t w = block * 512 / 64; w < word; w++) r += (uint64_t)__builtin_popcountll(m->bits[level][w]);
    r += (uint64_t)__builtin_popcountll(m->bits[level][word] & ((1ULL << (pos % 64)) - 1));
    return r;
}
uint64_t gko_mphf_save(const gko_mphf* m, uint8_t* out, uint64_t cap)                                            /* mphf::save :933-958, bitVector::save :627-635 */
{
    uint64_t pos = 0;
#define PUT(ptr, nbytes) do { if (out && pos + (nbytes) <= cap) memcpy(out + pos, (ptr), (nbytes)); pos += (nbytes); } while (0)
    const int nbl = MPHF_NB_LEVELS;
    PUT(&m->gamma, 8); PUT(&nbl, 4); PUT(&m->lastbitsetrank, 8); PUT(&m->nelem, 8);
    for (int i = 0; i < MPHF_NB_LEVELS; i++) {
        PUT(&m->domain[i], 8); PUT(&m->nchar[i], 8); PUT(m->bits[i], m->nchar[i] * 8);
        PUT(&m->nranks[i], 8); PUT(m->ranks[i], m->nranks[i] * 8);
    }
    PUT(&m->nfinal, 8);
    for (uint64_t i = 0; i < m->nfinal; i++) { PUT(&m->final_lo[i], 8); if (m->wide) PUT(&m->final_hi[i], 8); PUT(&m->final_val[i], 8); }
#undef PUT
    return pos;
}
int gko_abundance_index(int abundance)                                                                           /* MapMPHF.hpp:96-145 + MPHFAlgorithm.cpp:253-266 */
{
    static int disc[257]; static int init = 0;
    if (!init) {
        int total = 0, idx = 1; disc[0] = 0;
        for (int i = 1; i <= 70; i++, idx++) { total += 1; disc[idx] = total; }
        for (int i = 1; i <= 15; i++, idx++) { total += 2; disc[idx] = total; }
        for (int i = 1; i <= 40; i++, idx++) { total += 10; disc[idx] = total; }
        for (int i = 1; i <= 25; i++, idx++) { total += 20; disc[idx] = total; }
        for (int i = 1; i <= 40; i++, idx++) { total += 100; disc[idx] = total; }
        for (int i = 1; i <= 25; i++, idx++) { total += 200; disc[idx] = total; }
        for (int i = 1; i <= 40; i++, idx++) { total += 1000; disc[idx] = total; }
        disc[256] = total; init = 1;
    }
    const int max_discrete = disc[257 - 2];
    if (abundance >= max_discrete) return 257 - 2;
    int lo = 0, hi = 257;                                        /* std::upper_bound: first cell strictly greater than abundance, then the previous one */
    while (lo < hi) { const int mid = (lo + hi) / 2; if (disc[mid] <= abundance) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
