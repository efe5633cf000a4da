/*
 * umi_oracle.c -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 * See umi_oracle.h for the usage rule and the "parity unpinned" statement.
 * Plain C, scalar, single-threaded: it follows the reference's control flow
 * (linear-scan Naive store, root loop, neighbour visit) so that timing it is a
 * fair "port" CPU baseline; it is NOT tuned.
 */
#include "umi_oracle.h"

#include <stdlib.h>
#include <string.h>

/* src/utils/read.rs:13-14 */
#define ENCODING_DIST 2
#define ENCODING_LENGTH 3
#define CHUNK_SIZE 64 /* src/utils/bitset.rs:6 */

/* src/utils/read.rs:23-31: A=0b000 T=0b101 C=0b110 G=0b011 N=0b100 */
static int encode_char(uint8_t c)
{
    switch (c) {
    case 'A': return 0x0;
    case 'T': return 0x5;
    case 'C': return 0x6;
    case 'G': return 0x3;
    case 'N': return 0x4;
    default: return -1;
    }
}

int orc_to_bitset(const uint8_t *s, int len, orc_bitset *out)
{
    /* utils/mod.rs:65-68 + bitset.rs:17-27 */
    int total_bits = len * ENCODING_LENGTH;
    int cap = total_bits / CHUNK_SIZE + ((total_bits % CHUNK_SIZE) == 0 ? 0 : 1);
    if (cap > ORC_MAXW) return -1;
    memset(out, 0, sizeof(*out));
    out->nwords = cap;
    for (int i = 0; i < len; i++) {
        int enc = encode_char(s[i]);
        if (enc < 0) return -1; /* utils/mod.rs:77-79 panic */
        for (int j = 0; j < ENCODING_LENGTH; j++) { /* char_set, utils/mod.rs:38-41 */
            int idx = i * ENCODING_LENGTH + j;
            uint64_t bit = (uint64_t)1 << (idx % CHUNK_SIZE);
            if (enc & (1 << j))
                out->bits[idx / CHUNK_SIZE] |= bit;
            else
                out->bits[idx / CHUNK_SIZE] &= ~bit;
        }
        if (s[i] == 'N') { /* char_set_n_bit, utils/mod.rs:45-50,74-76 */
            out->has_n = 1;
            for (int j = 0; j < ENCODING_LENGTH; j++) {
                int idx = i * ENCODING_LENGTH + j;
                out->nbits[idx / CHUNK_SIZE] |= (uint64_t)1 << (idx % CHUNK_SIZE);
            }
        }
    }
    return 0;
}

int32_t orc_bitset_hash(const orc_bitset *b)
{
    /* bitset.rs:133-141, i64 arithmetic wraps in release builds (Cargo.toml:16-19) */
    uint64_t h = 1234;
    for (int i = b->nwords; i > 0;) {
        i -= 1;
        h ^= b->bits[i] * (uint64_t)(i + 1);
    }
    int64_t hs = (int64_t)h;
    return (int32_t)(uint32_t)(uint64_t)(hs ^ (hs >> 32));
}

int32_t orc_bit_count_xor(const orc_bitset *a, const orc_bitset *b)
{
    /* bitset.rs:77-91 */
    int32_t res = 0;
    for (int i = 0; i < a->nwords; i++) {
        uint64_t an = a->has_n ? a->nbits[i] : 0;
        uint64_t bn = b->has_n ? b->nbits[i] : 0;
        uint64_t x = an ^ bn;
        res += (int32_t)__builtin_popcountll(x | (a->bits[i] ^ b->bits[i])) -
               (int32_t)__builtin_popcountll(x) / ENCODING_LENGTH;
    }
    return res;
}

int32_t orc_umi_dist(const orc_bitset *a, const orc_bitset *b)
{
    return orc_bit_count_xor(a, b) / ENCODING_DIST; /* utils/mod.rs:24-26 */
}

/* Rust `f32 as i32`: truncate toward zero, saturate, NaN -> 0 */
static int32_t f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

int32_t orc_threshold(float percentage, int32_t freq)
{
    /* directional.rs:38 */
    /* freq + 1 wraps in a release build of the reference (Cargo.toml:16-19) */
    volatile float prod = percentage * (float)(int32_t)((uint32_t)freq + 1u);
    return f32_as_i32(prod);
}

int32_t orc_avg_qual(const uint8_t *qual, int len)
{
    /* read.rs:57-60: f32 running sum of `b as f32`, then / seq_len as f32 */
    volatile float sum = 0.0f;
    for (int i = 0; i < len; i++) sum = sum + (float)qual[i];
    volatile float avg = sum / (float)len;
    return f32_as_i32(avg);
}

/* ---- Naive ------------------------------------------------------------- */
struct orc_naive {
    const orc_bitset *umis;
    const int32_t *freq;
    uint32_t n;
    uint8_t *present; /* membership in the HashMap<&BitSet,i32> of naive.rs:13 */
    uint64_t dist_calls;
};

orc_naive *orc_naive_new(const orc_bitset *umis, const int32_t *freq, uint32_t n)
{
    orc_naive *d = (orc_naive *)calloc(1, sizeof(*d));
    if (!d) return NULL;
    d->umis = umis;
    d->freq = freq;
    d->n = n;
    d->present = (uint8_t *)malloc(n ? n : 1);
    if (!d->present) { free(d); return NULL; }
    memset(d->present, 1, n);
    return d;
}

void orc_naive_free(orc_naive *d)
{
    if (!d) return;
    free(d->present);
    free(d);
}

uint32_t orc_naive_remove_near(orc_naive *d, uint32_t query, int32_t k, int32_t max_freq,
                               uint32_t *out)
{
    /* naive.rs:26-40: retain() visits every remaining entry */
    uint32_t cnt = 0;
    const orc_bitset *q = &d->umis[query];
    for (uint32_t o = 0; o < d->n; o++) {
        if (!d->present[o]) continue;
        int32_t dist = orc_umi_dist(q, &d->umis[o]);
        d->dist_calls++;
        if (dist <= k && (dist == 0 || d->freq[o] <= max_freq)) { /* naive.rs:31 */
            d->present[o] = 0;
            out[cnt++] = o;
        }
    }
    return cnt;
}

int orc_naive_contains(const orc_naive *d, uint32_t idx) { return d->present[idx] != 0; }
uint64_t orc_naive_dist_calls(const orc_naive *d) { return d->dist_calls; }

/* stable sort of indices by freq descending (directional.rs:72, adjacency.rs:45):
 * merge sort on (index) keeps first-appearance order among equal freq. */
static void stable_rank(const int32_t *freq, uint32_t n, uint32_t *order)
{
    uint32_t *tmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    for (uint32_t width = 1; width < n; width *= 2) {
        for (uint32_t lo = 0; lo < n; lo += 2 * width) {
            uint32_t mid = lo + width < n ? lo + width : n;
            uint32_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            uint32_t a = lo, b = mid, t = lo;
            while (a < mid && b < hi) {
                /* take from the right run only if strictly greater freq */
                if (freq[order[b]] > freq[order[a]]) tmp[t++] = order[b++];
                else tmp[t++] = order[a++];
            }
            while (a < mid) tmp[t++] = order[a++];
            while (b < hi) tmp[t++] = order[b++];
        }
        memcpy(order, tmp, n * sizeof(uint32_t));
    }
    free(tmp);
}

uint32_t orc_directional_apply(const orc_bitset *umis, const int32_t *freq, uint32_t n, int32_t k,
                               float percentage, uint32_t *out_idx, uint32_t *root_of,
                               uint64_t *dist_calls)
{
    uint32_t n_out = 0;
    uint32_t *order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *stack = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *near = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    stable_rank(freq, n, order);                    /* directional.rs:67-72 */
    orc_naive *data = orc_naive_new(umis, freq, n); /* directional.rs:64-65,74 */
    for (uint32_t r = 0; r < n; r++) {              /* directional.rs:78-88 */
        uint32_t root = order[r];
        if (!orc_naive_contains(data, root)) continue; /* :81 */
        /* visit_and_remove (:30-54) with an explicit stack instead of recursion:
         * every removed UMI is the start_umi of exactly one remove_near call. */
        uint32_t sp = 0;
        stack[sp++] = root;
        while (sp) {
            uint32_t start = stack[--sp];
            int32_t threshold = orc_threshold(percentage, freq[start]); /* :38 */
            uint32_t cnt = orc_naive_remove_near(data, start, k, threshold, near); /* :39 */
            for (uint32_t t = 0; t < cnt; t++) {
                uint32_t v = near[t];
                if (root_of) root_of[v] = root; /* tracker.add_all :42-44 */
                if (v == start) continue;       /* :48-50 */
                stack[sp++] = v;                /* :52 */
            }
        }
        out_idx[n_out++] = root; /* :86 */
    }
    if (dist_calls) *dist_calls += orc_naive_dist_calls(data);
    orc_naive_free(data);
    free(order);
    free(stack);
    free(near);
    return n_out;
}

uint32_t orc_adjacency_apply(const orc_bitset *umis, const int32_t *freq, uint32_t n, int32_t k,
                             int32_t max_freq, uint32_t *out_idx, uint32_t *root_of,
                             uint64_t *dist_calls)
{
    uint32_t n_out = 0;
    uint32_t *order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *near = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    stable_rank(freq, n, order);                    /* adjacency.rs:40-45 */
    orc_naive *data = orc_naive_new(umis, freq, n); /* adjacency.rs:47-49 */
    for (uint32_t r = 0; r < n; r++) {              /* adjacency.rs:52-60 */
        uint32_t root = order[r];
        if (!orc_naive_contains(data, root)) continue;
        uint32_t cnt = orc_naive_remove_near(data, root, k, max_freq, near); /* :56 */
        if (root_of)
            for (uint32_t t = 0; t < cnt; t++) root_of[near[t]] = root;
        out_idx[n_out++] = root;
    }
    if (dist_calls) *dist_calls += orc_naive_dist_calls(data);
    orc_naive_free(data);
    free(order);
    free(near);
    return n_out;
}

int orc_dedup_batch_wide(const uint64_t *keys, const uint64_t *nmask, int n_words, const int32_t *freq,
                         const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int32_t k,
                         float percentage, int algo, int32_t adj_max_freq, uint8_t *kept,
                         uint32_t *root, uint64_t *dist_calls)
{
    int total_bits = umi_len * ENCODING_LENGTH;
    int cap = total_bits / CHUNK_SIZE + ((total_bits % CHUNK_SIZE) == 0 ? 0 : 1); /* bitset.rs:17-18 */
    if (cap != n_words || n_words < 1 || n_words > ORC_MAXW) return -2;
    for (uint64_t b = 0; b < n_buckets; b++) { /* deduplicate_sam.rs:207 */
        uint64_t s = bucket_off[b], e = bucket_off[b + 1];
        uint32_t n = (uint32_t)(e - s);
        if (n == 0) continue;
        orc_bitset *umis = (orc_bitset *)calloc(n, sizeof(orc_bitset));
        uint32_t *out = (uint32_t *)malloc(n * sizeof(uint32_t));
        uint32_t *rof = (uint32_t *)malloc(n * sizeof(uint32_t));
        for (uint32_t i = 0; i < n; i++) {
            if (i && freq[s + i] > freq[s + i - 1]) {
                free(umis); free(out); free(rof);
                return -1;
            }
            umis[i].nwords = n_words;
            int any_n = 0;
            for (int w = 0; w < n_words; w++) {
                umis[i].bits[w] = keys[(s + i) * (uint64_t)n_words + w];
                if (nmask && nmask[(s + i) * (uint64_t)n_words + w]) any_n = 1;
            }
            if (any_n) { /* n_bits is Some(..) only for a UMI with an N (utils/mod.rs:74-76) */
                umis[i].has_n = 1;
                for (int w = 0; w < n_words; w++) umis[i].nbits[w] = nmask[(s + i) * (uint64_t)n_words + w];
            }
            rof[i] = i;
        }
        uint32_t ns;
        if (algo == 0)
            ns = orc_directional_apply(umis, freq + s, n, k, percentage, out, rof, dist_calls);
        else
            ns = orc_adjacency_apply(umis, freq + s, n, k, adj_max_freq, out, rof, dist_calls);
        for (uint32_t i = 0; i < n; i++) {
            kept[s + i] = 0;
            if (root) root[s + i] = (uint32_t)(s + rof[i]);
        }
        for (uint32_t i = 0; i < ns; i++) kept[s + out[i]] = 1;
        free(umis);
        free(out);
        free(rof);
    }
    return 0;
}

int orc_dedup_batch(const uint64_t *keys, const uint64_t *nmask, const int32_t *freq,
                    const uint64_t *bucket_off, uint64_t n_buckets, int umi_len, int32_t k,
                    float percentage, int algo, int32_t adj_max_freq, uint8_t *kept,
                    uint32_t *root, uint64_t *dist_calls)
{ /* the batched form with one word per key */
    return orc_dedup_batch_wide(keys, nmask, 1, freq, bucket_off, n_buckets, umi_len, k, percentage, algo,
                                adj_max_freq, kept, root, dist_calls);
}

/* ---- staging ----------------------------------------------------------- */
typedef struct {
    uint32_t bucket;
    uint64_t key, nmask;
    int32_t freq, score;
    uint64_t rep;
} stage_entry;

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

int orc_stage_reads(const uint32_t *bucket_id, const uint8_t *umi, const int32_t *score,
                    uint64_t n_reads, int umi_len, int merge, uint64_t *keys, uint64_t *nmask,
                    int32_t *freq, uint64_t *rep, uint64_t *bucket_off, uint64_t *n_out,
                    uint64_t *b_out)
{
    int rc = 0;
    uint64_t cap = 16;
    while (cap < 2 * n_reads + 16) cap *= 2;
    uint64_t *table = (uint64_t *)malloc(cap * sizeof(uint64_t)); /* entry index + 1 */
    stage_entry *ent = (stage_entry *)malloc((n_reads ? n_reads : 1) * sizeof(stage_entry));
    memset(table, 0, cap * sizeof(uint64_t));
    uint64_t n_ent = 0;
    uint32_t max_bucket = 0;
    for (uint64_t i = 0; i < n_reads; i++) {
        orc_bitset bs;
        if (orc_to_bitset(umi + i * (uint64_t)umi_len, umi_len, &bs) != 0 || bs.nwords != 1) {
            rc = -1;
            goto done;
        }
        uint32_t b = bucket_id[i];
        if (b > max_bucket) max_bucket = b;
        uint64_t h = mix64(bs.bits[0] ^ mix64((uint64_t)b + 0x9e3779b97f4a7c15ULL)) & (cap - 1);
        for (;;) {
            uint64_t slot = table[h];
            if (slot == 0) { /* Vacant: deduplicate_sam.rs:161-163 */
                stage_entry *e = &ent[n_ent];
                e->bucket = b; e->key = bs.bits[0]; e->nmask = bs.has_n ? bs.nbits[0] : 0;
                e->freq = 1; e->score = score ? score[i] : 0; e->rep = i;
                table[h] = ++n_ent;
                break;
            }
            stage_entry *e = &ent[slot - 1];
            if (e->bucket == b && e->key == bs.bits[0]) { /* Occupied: :164-175 */
                int32_t sc = score ? score[i] : 0;
                int keep_existing = (merge == 0) ? 1 : (e->score >= sc); /* merge/mod.rs:21,35,49 */
                e->freq += 1;
                if (!keep_existing) { e->rep = i; e->score = sc; }
                break;
            }
            h = (h + 1) & (cap - 1);
        }
    }
    {
        /* buckets in order of first appearance; inside a bucket stable freq-desc */
        uint64_t nb_ids = (uint64_t)max_bucket + 1;
        uint64_t *first = (uint64_t *)malloc(nb_ids * sizeof(uint64_t)); /* dense bucket no. + 1 */
        uint64_t *count = (uint64_t *)calloc(n_ent + 1, sizeof(uint64_t));
        memset(first, 0, nb_ids * sizeof(uint64_t));
        uint64_t nb = 0;
        for (uint64_t j = 0; j < n_ent; j++) {
            if (first[ent[j].bucket] == 0) first[ent[j].bucket] = ++nb;
            count[first[ent[j].bucket] - 1]++;
        }
        bucket_off[0] = 0;
        for (uint64_t b = 0; b < nb; b++) bucket_off[b + 1] = bucket_off[b] + count[b];
        uint64_t *fill = (uint64_t *)calloc(nb + 1, sizeof(uint64_t));
        uint64_t *slot_of = (uint64_t *)malloc((n_ent ? n_ent : 1) * sizeof(uint64_t));
        for (uint64_t j = 0; j < n_ent; j++) { /* entry order == first appearance order */
            uint64_t b = first[ent[j].bucket] - 1;
            slot_of[bucket_off[b] + fill[b]++] = j;
        }
        for (uint64_t b = 0; b < nb; b++) {
            uint64_t s = bucket_off[b];
            uint32_t n = (uint32_t)(bucket_off[b + 1] - s);
            int32_t *f = (int32_t *)malloc(n * sizeof(int32_t));
            uint32_t *order = (uint32_t *)malloc(n * sizeof(uint32_t));
            for (uint32_t t = 0; t < n; t++) f[t] = ent[slot_of[s + t]].freq;
            stable_rank(f, n, order);
            for (uint32_t t = 0; t < n; t++) {
                const stage_entry *e = &ent[slot_of[s + order[t]]];
                keys[s + t] = e->key; nmask[s + t] = e->nmask; freq[s + t] = e->freq;
                rep[s + t] = e->rep;
            }
            free(f);
            free(order);
        }
        *n_out = n_ent;
        *b_out = nb;
        free(first); free(count); free(fill); free(slot_of);
    }
done:
    free(table);
    free(ent);
    return rc;
}
