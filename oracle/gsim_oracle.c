/* gsim_oracle.c -- CPU restatement of the reference hot path.  TEST
 * INFRASTRUCTURE ONLY (see gsim_oracle.h).  Plain C11 + pthreads.
 *
 * Compile with -ffp-contract=off so that the Tversky expression is evaluated
 * with one rounding per operation, the same as the device kernel's explicit
 * __fmul_rn/__fadd_rn/__fdiv_rn chain.
 */
#include "gsim_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* synthetic data                                                            */
/* ------------------------------------------------------------------------- */

uint64_t gso_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* n-th output of the SplitMix64 stream seeded with `seed`. */
static inline uint64_t stream_at(uint64_t seed, uint64_t n)
{
    return gso_splitmix64(seed + n * 0x9E3779B97F4A7C15ULL);
}

uint32_t gso_synth_word(uint64_t seed, int kind, uint64_t row, uint32_t W, uint32_t j)
{
    const uint64_t ctr = row * (uint64_t) W + j;
    if (kind == GSO_KIND_MORGAN) { /* rows of this kind are made whole (gso_synth_row_morgan) */
        uint32_t tmp[1024];
        if (W > 1024 || j >= W) return 0;
        gso_synth_row_morgan(tmp, seed, row, W);
        return tmp[j];
    }
    if (kind == GSO_KIND_DENSE) {
        return (uint32_t) stream_at(seed, ctr);
    }
    const uint64_t h0 = stream_at(seed, 2 * ctr);
    const uint64_t h1 = stream_at(seed, 2 * ctr + 1);
    return (uint32_t) h0 & (uint32_t) (h0 >> 32) & (uint32_t) h1 & (uint32_t) (h1 >> 32);
}

/* Morgan-shaped rows (GSO_KIND_MORGAN).  What the reference's numbers are quoted on is 1024-bit
 * Morgan r=2 fingerprints (python/gpusim_utils.py:21,55-66); its fixture test/small.fsim has popcounts
 * 20..53 (mean 34.5), a dozen bits set in more than half of the rows and a long tail, mean pairwise
 * Tanimoto 0.155.  The generator reproduces that shape and what real libraries add: rows come in
 * scaffold clusters (contiguous series of 16..1024 analogs, table-wide scaffolds, singletons) and a
 * few per cent are exact duplicates.  A row is: 16 "common" bits drawn per scaffold with the
 * fixture's frequencies + 10..26 scaffold bits (minus up to two dropped per member) + 2..13 member
 * bits.  Counter-based, integer-only: row r depends on (seed, r, fp width) alone.
 * The device twin is generate_morgan_kernel (gsim_device.hip), the product-side host twin
 * gsim_synth_row (gsim_capi.cpp). */
static inline uint64_t mh2(uint64_t seed, uint64_t tag, uint64_t a, uint64_t b)
{
    return gso_splitmix64(gso_splitmix64(seed + tag * 0xD1B54A32D192ED03ULL + a * 0x9E3779B97F4A7C15ULL) +
                          b * 0x9E3779B97F4A7C15ULL);
}

static const uint8_t k_morgan_common[16] = {253, 253, 236, 220, 200, 174, 166, 161, 161, 141, 133, 131, 90, 84, 74, 74};

static inline uint32_t morgan_pos(uint64_t x, uint32_t W)
{
    /* product of two uniforms: density ~ -ln(u), then the (word, bit) transposition spreads the
     * frequent low positions over the words */
    const uint32_t u = (uint32_t) (((x & 0xFFFFu) * ((x >> 16) & 0xFFFFu)) >> 16);
    const uint32_t pos = (uint32_t) (((uint64_t) u * (W * 32u)) >> 16);
    return (pos & 31u) * W + (pos >> 5);
}

void gso_synth_row_morgan(uint32_t* out, uint64_t seed, uint64_t row, uint32_t W)
{
    const uint32_t nbits = W * 32u;
    for (uint32_t j = 0; j < W; j++) out[j] = 0;
    const uint64_t rh = mh2(seed, 1, row, 0);
    const uint64_t sh = mh2(seed, 2, row >> 10, 0);
    uint64_t sid, mspace;
    switch (rh & 3u) {
    case 0: /* table-wide scaffold */
        sid = (1ULL << 62) | ((rh >> 8) & 0xFFFFu);
        mspace = 1u << 14;
        break;
    case 1: /* singleton */
        sid = (2ULL << 62) | row;
        mspace = 1;
        break;
    default: { /* a contiguous series of S = 16, 64, 256 or 1024 rows */
        const uint32_t lg = 4u + 2u * (uint32_t) (sh & 3u);
        sid = (row >> lg) | ((uint64_t) lg << 56);
        mspace = 4ULL << lg;
    }
    }
    const uint64_t m = (rh >> 24) % mspace;
    const uint64_t kh = mh2(seed, 3, sid, 0);
    const uint64_t mh = mh2(seed, 6, sid, m);
    for (uint32_t i = 0; i < 16; i++) {
        const uint64_t c = mh2(seed, 4, sid, i >> 3);
        if (((c >> (8 * (i & 7u))) & 0xFFu) < k_morgan_common[i]) {
            const uint32_t p = (i * 67u + 5u) % nbits;
            out[p >> 5] |= 1u << (p & 31u);
        }
    }
    const uint32_t ps = 10u + (uint32_t) (kh % 17u);
    const uint32_t ndrop = (uint32_t) (mh % 3u);
    const uint32_t d0 = (uint32_t) ((mh >> 8) & 0xFFu) % ps, d1 = (uint32_t) ((mh >> 16) & 0xFFu) % ps;
    for (uint32_t j = 0; j < ps; j++) {
        if ((ndrop >= 1 && j == d0) || (ndrop >= 2 && j == d1)) continue;
        const uint32_t p = morgan_pos(mh2(seed, 5, sid, j), W);
        out[p >> 5] |= 1u << (p & 31u);
    }
    const uint32_t na = 2u + (uint32_t) ((mh >> 32) % 12u);
    for (uint32_t j = 0; j < na; j++) {
        const uint32_t p = morgan_pos(gso_splitmix64(mh + (j + 1) * 0x9E3779B97F4A7C15ULL), W);
        out[p >> 5] |= 1u << (p & 31u);
    }
}

void gso_synth_rows(uint32_t* out, uint64_t seed, int kind, uint64_t first_row,
                    uint64_t nrows, uint32_t W)
{
    if (kind == GSO_KIND_MORGAN) {
        for (uint64_t r = 0; r < nrows; r++) gso_synth_row_morgan(out + r * W, seed, first_row + r, W);
        return;
    }
    for (uint64_t r = 0; r < nrows; r++) {
        for (uint32_t j = 0; j < W; j++) {
            out[r * W + j] = gso_synth_word(seed, kind, first_row + r, W, j);
        }
    }
}

uint64_t gso_query_row(uint64_t q, uint64_t nrows)
{
    return nrows ? gso_splitmix64(0xC0FFEEULL + q) % nrows : 0;
}

/* ------------------------------------------------------------------------- */
/* scoring                                                                   */
/* ------------------------------------------------------------------------- */

static inline void row_counts(const uint32_t* q, const uint32_t* d, uint32_t W,
                              uint32_t* common, uint32_t* popc_d)
{
    uint32_t c = 0, b = 0;
    for (uint32_t i = 0; i < W; i++) {
        b += (uint32_t) __builtin_popcount(d[i]);
        c += (uint32_t) __builtin_popcount(q[i] & d[i]);
    }
    *common = c;
    *popc_d = b;
}

static inline uint32_t query_popc(const uint32_t* q, uint32_t W)
{
    uint32_t a = 0;
    for (uint32_t i = 0; i < W; i++) a += (uint32_t) __builtin_popcount(q[i]);
    return a;
}

/* calculation_functors.cpp:6-20 / fingerprintdb_cuda.cu:89-98:
 *   total = sum popc(q_i) + popc(d_i); common = sum popc(q_i & d_i);
 *   score = (float)common / (float)(total - common)                          */
float gso_score_one(int metric, float alpha, float beta, uint32_t a, uint32_t b,
                    uint32_t c)
{
    if (metric == GSO_METRIC_TVERSKY) {
        const float t1 = alpha * (float) (int32_t) (a - c);
        const float t2 = beta * (float) (int32_t) (b - c);
        const float den = (t1 + t2) + (float) c;
        return (float) c / den;
    }
    const int total = (int) (a + b);
    return (float) (int) c / (float) (total - (int) c);
}

/* the denominator of gso_score_one as its own step (twin of the device's score_den):
 * gso_score_one(...) == (float) c / gso_score_den(...) bit for bit                            */
float gso_score_den(int metric, float alpha, float beta, uint32_t a, uint32_t b, uint32_t c)
{
    if (metric == GSO_METRIC_TVERSKY) {
        const float t1 = alpha * (float) (int32_t) (a - c);
        const float t2 = beta * (float) (int32_t) (b - c);
        return (t1 + t2) + (float) c;
    }
    return (float) ((int) (a + b) - (int) c);
}

void gso_tanimoto_raw(const uint32_t* query, const uint32_t* db, uint64_t nrows,
                      uint32_t W, float* scores, uint16_t* common, uint16_t* popc)
{
    const uint32_t a = query_popc(query, W);
    for (uint64_t r = 0; r < nrows; r++) {
        uint32_t c, b;
        row_counts(query, db + r * W, W, &c, &b);
        scores[r] = gso_score_one(GSO_METRIC_TANIMOTO, 0.f, 0.f, a, b, c);
        if (common) common[r] = (uint16_t) c;
        if (popc) popc[r] = (uint16_t) b;
    }
}

/* fingerprintdb_cuda.cu:101: return score >= cutoff ? score : 0;  (NaN -> 0) */
float gso_apply_cutoff(float score, float cutoff)
{
    return score >= cutoff ? score : 0.0f;
}

/* ------------------------------------------------------------------------- */
/* canonical top-k                                                           */
/* ------------------------------------------------------------------------- */

/* "a ranks before b" in canonical order: score desc, row asc. */
static inline int hit_before(const gso_hit* a, const gso_hit* b)
{
    if (a->score > b->score) return 1;
    if (a->score < b->score) return 0;
    return a->row < b->row;
}

/* bounded heap whose root is the WORST kept hit (ranks last). */
typedef struct {
    gso_hit* h;
    uint32_t n, cap;
} hit_heap;

static void heap_sift_down(hit_heap* hp, uint32_t i)
{
    for (;;) {
        uint32_t l = 2 * i + 1, r = l + 1, w = i;
        if (l < hp->n && hit_before(&hp->h[w], &hp->h[l])) w = l;
        if (r < hp->n && hit_before(&hp->h[w], &hp->h[r])) w = r;
        if (w == i) return;
        gso_hit t = hp->h[i];
        hp->h[i] = hp->h[w];
        hp->h[w] = t;
        i = w;
    }
}

static void heap_offer(hit_heap* hp, const gso_hit* x)
{
    if (hp->cap == 0) return;
    if (hp->n < hp->cap) {
        uint32_t i = hp->n++;
        hp->h[i] = *x;
        while (i > 0) {
            uint32_t p = (i - 1) / 2;
            if (!hit_before(&hp->h[p], &hp->h[i])) break;
            gso_hit t = hp->h[i];
            hp->h[i] = hp->h[p];
            hp->h[p] = t;
            i = p;
        }
        return;
    }
    if (hit_before(x, &hp->h[0])) {
        hp->h[0] = *x;
        heap_sift_down(hp, 0);
    }
}

static int hit_cmp_qsort(const void* pa, const void* pb)
{
    const gso_hit* a = (const gso_hit*) pa;
    const gso_hit* b = (const gso_hit*) pb;
    if (hit_before(a, b)) return -1;
    if (hit_before(b, a)) return 1;
    return 0;
}

typedef struct {
    const uint32_t* query;
    const uint32_t* db;
    uint64_t r0, r1;
    uint32_t W, k, a, row_base;
    float cutoff, alpha, beta;
    int metric;
    hit_heap heap;
    uint64_t kept;
} scan_job;

static void* scan_worker(void* arg)
{
    scan_job* j = (scan_job*) arg;
    uint64_t kept = 0;
    for (uint64_t r = j->r0; r < j->r1; r++) {
        uint32_t c, b;
        row_counts(j->query, j->db + r * j->W, j->W, &c, &b);
        float s = gso_score_one(j->metric, j->alpha, j->beta, j->a, b, c);
        s = gso_apply_cutoff(s, j->cutoff);
        /* fingerprintdb_cuda.cu:263-273: compaction only when cutoff > 0 */
        if (j->cutoff > 0.0f && !(s != 0.0f)) continue;
        kept++;
        gso_hit h;
        h.row = (uint32_t) r + j->row_base;
        h.score = s;
        h.common = (uint16_t) c;
        h.popc_db = (uint16_t) b;
        heap_offer(&j->heap, &h);
    }
    j->kept = kept;
    return NULL;
}

int gso_search(const uint32_t* query, const uint32_t* db, uint64_t nrows, uint32_t W,
               uint32_t k, float cutoff, int metric, float alpha, float beta,
               uint32_t row_base, int nthreads, gso_hit* hits, uint32_t* nhits,
               uint64_t* approx)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t) nthreads > nrows) nthreads = nrows ? (int) nrows : 1;
    const uint32_t kk = (uint64_t) k < nrows ? k : (uint32_t) nrows;
    scan_job* jobs = (scan_job*) calloc((size_t) nthreads, sizeof(scan_job));
    pthread_t* th = (pthread_t*) calloc((size_t) nthreads, sizeof(pthread_t));
    if (!jobs || !th) return -1;
    const uint32_t a = query_popc(query, W);
    const uint64_t per = (nrows + (uint64_t) nthreads - 1) / (uint64_t) nthreads;
    for (int t = 0; t < nthreads; t++) {
        scan_job* j = &jobs[t];
        j->query = query;
        j->db = db;
        j->r0 = per * (uint64_t) t;
        j->r1 = j->r0 + per < nrows ? j->r0 + per : nrows;
        if (j->r0 > nrows) j->r0 = nrows;
        j->W = W;
        j->k = kk;
        j->a = a;
        j->row_base = row_base;
        j->cutoff = cutoff;
        j->alpha = alpha;
        j->beta = beta;
        j->metric = metric;
        j->heap.cap = kk;
        j->heap.n = 0;
        j->heap.h = (gso_hit*) malloc(sizeof(gso_hit) * (kk ? kk : 1));
        if (!j->heap.h) return -1;
    }
    if (nthreads == 1) {
        scan_worker(&jobs[0]);
    } else {
        for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, scan_worker, &jobs[t]);
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    }
    uint64_t kept = 0;
    size_t total = 0;
    for (int t = 0; t < nthreads; t++) {
        kept += jobs[t].kept;
        total += jobs[t].heap.n;
    }
    gso_hit* all = (gso_hit*) malloc(sizeof(gso_hit) * (total ? total : 1));
    size_t o = 0;
    for (int t = 0; t < nthreads; t++) {
        memcpy(all + o, jobs[t].heap.h, sizeof(gso_hit) * jobs[t].heap.n);
        o += jobs[t].heap.n;
        free(jobs[t].heap.h);
    }
    qsort(all, total, sizeof(gso_hit), hit_cmp_qsort);
    uint32_t n = total < kk ? (uint32_t) total : kk;
    if (hits) memcpy(hits, all, sizeof(gso_hit) * n);
    if (nhits) *nhits = n;
    /* fingerprintdb_cuda.cu:272-277: survivor count; == nrows when cutoff <= 0 */
    if (approx) *approx = kept;
    free(all);
    free(jobs);
    free(th);
    return 0;
}

void gso_merge_hits(const gso_hit* lists, const uint32_t* counts, uint32_t nlists,
                    uint32_t stride, uint32_t k, gso_hit* out, uint32_t* nout)
{
    size_t total = 0;
    for (uint32_t i = 0; i < nlists; i++) total += counts[i];
    gso_hit* all = (gso_hit*) malloc(sizeof(gso_hit) * (total ? total : 1));
    size_t o = 0;
    for (uint32_t i = 0; i < nlists; i++) {
        memcpy(all + o, lists + (size_t) i * stride, sizeof(gso_hit) * counts[i]);
        o += counts[i];
    }
    qsort(all, total, sizeof(gso_hit), hit_cmp_qsort);
    uint32_t n = total < k ? (uint32_t) total : k;
    memcpy(out, all, sizeof(gso_hit) * n);
    *nout = n;
    free(all);
}

/* ------------------------------------------------------------------------- */
/* CPU-path pieces                                                           */
/* ------------------------------------------------------------------------- */

/* fingerprintdb_cuda.cpp:92-103 */
void gso_bubble_sort(int* indices, float* scores, int count, int number_required)
{
    for (int i = 0; i < number_required; i++) {
        for (int j = count - 1; j > i; j--) {
            if (scores[j] > scores[j - 1]) {
                int ti = indices[j];
                indices[j] = indices[j - 1];
                indices[j - 1] = ti;
                float tf = scores[j];
                scores[j] = scores[j - 1];
                scores[j - 1] = tf;
            }
        }
    }
}

/* fingerprintdb_cuda.cpp:20-54 */
int gso_search_cpu(const uint32_t* query, const uint32_t* db, uint64_t nrows,
                   uint32_t W, uint32_t k, int* out_rows, float* out_scores)
{
    if (k > nrows) return -1;
    int* idx = (int*) malloc(sizeof(int) * (nrows ? nrows : 1));
    float* sc = (float*) malloc(sizeof(float) * (nrows ? nrows : 1));
    if (!idx || !sc) return -1;
    for (uint64_t r = 0; r < nrows; r++) idx[r] = (int) r;
    gso_tanimoto_raw(query, db, nrows, W, sc, NULL, NULL);
    gso_bubble_sort(idx, sc, (int) nrows, (int) k);
    for (uint32_t i = 0; i < k; i++) {
        out_rows[i] = idx[i];
        out_scores[i] = sc[i];
    }
    free(idx);
    free(sc);
    return 0;
}

/* calculation_functors.cpp:22-41: bit `pos` of the unfolded fingerprint is OR-ed
 * into bit `pos % new_size` of the folded one (word = new_pos / 32, same bit
 * position inside the word).                                                  */
void gso_fold(const int* unfolded, int unfolded_intsize, int factor, int* folded)
{
    const int folded_intsize = unfolded_intsize / factor;
    const int new_size = 32 * folded_intsize;
    const int original_size = 32 * unfolded_intsize;
    for (int pos = 0; pos < original_size; pos++) {
        const int w = pos / 32, b = pos % 32;
        const unsigned on = ((unsigned) unfolded[w] >> b) & 1u;
        const int np = pos % new_size;
        folded[np / 32] = (int) ((unsigned) folded[np / 32] | (on << b));
    }
}

/* ------------------------------------------------------------------------- */
/* folding                                                                   */
/* ------------------------------------------------------------------------- */

void gso_fold_rows(const uint32_t* rows, uint64_t nrows, uint32_t W, int factor, uint32_t* folded)
{
    const uint32_t Wf = W / (uint32_t) factor;
    memset(folded, 0, (size_t) nrows * Wf * 4);
    for (uint64_t r = 0; r < nrows; r++) {
        gso_fold((const int*) (rows + r * W), (int) W, factor, (int*) (folded + r * Wf));
    }
}

/* fingerprintdb_cuda.cu:170-173 */
int gso_effective_fold_factor(uint32_t W, int requested)
{
    int f = requested;
    while (W % (uint32_t) f != 0) f++;
    return f;
}

int gso_search_folded(const uint32_t* query, const uint32_t* db, uint64_t nrows, uint32_t W,
                      int fold_factor, uint32_t k, float cutoff, uint32_t row_base,
                      gso_hit* hits, uint32_t* nhits, uint64_t* approx)
{
    const int F = fold_factor;
    const uint32_t Wf = W / (uint32_t) F;
    uint32_t* fdb = (uint32_t*) malloc((size_t) (nrows ? nrows : 1) * Wf * 4);
    uint32_t* fq = (uint32_t*) calloc(Wf, 4);
    if (!fdb || !fq) return -1;
    gso_fold_rows(db, nrows, W, F, fdb);
    gso_fold((const int*) query, (int) W, F, (int*) fq);
    /* fingerprintdb_cuda.cu:284-287: results_to_consider = min(survivors, k * F * (int)log2(2F)) */
    int lg = 0;
    while ((1 << (lg + 1)) <= 2 * F) lg++;
    const uint64_t want = (uint64_t) k * (uint64_t) F * (uint64_t) lg;
    const uint32_t R = want > nrows ? (uint32_t) nrows : (uint32_t) want;
    gso_hit* cand = (gso_hit*) malloc(sizeof(gso_hit) * (R ? R : 1));
    uint32_t ncand = 0;
    uint64_t surv = 0;
    gso_search(fq, fdb, nrows, Wf, R, cutoff, GSO_METRIC_TANIMOTO, 0.f, 0.f, 0, 1, cand, &ncand, &surv);
    /* re-score with the full fingerprints, in candidate order */
    int* idx = (int*) malloc(sizeof(int) * (ncand ? ncand : 1));
    float* sc = (float*) malloc(sizeof(float) * (ncand ? ncand : 1));
    const uint32_t a = query_popc(query, W);
    for (uint32_t i = 0; i < ncand; i++) {
        uint32_t c, b;
        row_counts(query, db + (uint64_t) cand[i].row * W, W, &c, &b);
        idx[i] = (int) cand[i].row;
        sc[i] = gso_score_one(GSO_METRIC_TANIMOTO, 0.f, 0.f, a, b, c);
    }
    gso_bubble_sort(idx, sc, (int) ncand, (int) k); /* :315 */
    uint32_t n = k < ncand ? k : ncand;
    uint32_t out = 0;
    for (uint32_t i = 0; i < n; i++) {
        if (sc[i] < cutoff) break; /* :321-325 */
        uint32_t c, b;
        row_counts(query, db + (uint64_t) idx[i] * W, W, &c, &b);
        hits[out].row = (uint32_t) idx[i] + row_base;
        hits[out].score = sc[i];
        hits[out].common = (uint16_t) c;
        hits[out].popc_db = (uint16_t) b;
        out++;
    }
    *nhits = out;
    if (approx) *approx = surv;
    free(idx);
    free(sc);
    free(cand);
    free(fdb);
    free(fq);
    return 0;
}
