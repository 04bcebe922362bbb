/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the EAST hot path.
 *
 * A plain-C, single-threaded restatement of the reference's enhanced
 * annotated suffix array (east/asts/easa.py) used ONLY as the checker in
 * tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline leg.
 * The product path (ast-text-analysis_amd/) never links, loads or calls it.
 *
 * Parity pinning: checked against every golden vector the reference holds
 * (README.rst:149-152, tests/asts/test_base.py:13-24) and against fixtures
 * generated from the imported reference (tests/golden/, made by
 * oracle/gen_golden.py); see tests/test_oracle_golden.py.
 *
 * Each function cites the reference lines it follows.  Tables are int64 like
 * the reference's np.int arrays.  Symbols are Unicode code points (uint32),
 * the per-string terminators being 0x0A00+i (east/asts/utils.py:25-40).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* ------------------------------------------------------------------ */
/* _radixpass (easa.py:230-245): stable counting sort of a[0..n) by    */
/* r[a[i]] into b; the alphabet dict becomes a dense count array [0,K]. */
static void radixpass(const i64 *a, i64 *b, const i64 *r, i64 n, i64 K)
{
    i64 *c = (i64 *)calloc((size_t)K + 2, sizeof(i64));
    for (i64 i = 0; i < n; i++) c[r[a[i]]]++;
    i64 total = 0;
    for (i64 k = 0; k <= K; k++) { i64 f = c[k]; c[k] = total; total += f; }
    for (i64 i = 0; i < n; i++) b[c[r[a[i]]]++] = a[i];
    free(c);
}

/* _kark_sort (easa.py:155-228).  s has n symbols in [1,K] followed by three
 * pad symbols smaller than every real one (the reference pads with chr(1) at
 * level 0 and with 0 in the recursion; both are "< every symbol"). */
static void kark_sort(const i64 *s, i64 *SA, i64 n, i64 K)
{
    i64 n0 = (n + 2) / 3, n1 = (n + 1) / 3, n2 = n / 3, n02 = n0 + n2;
    i64 *SA12 = (i64 *)calloc((size_t)n02 + 3, sizeof(i64));
    i64 *SA0 = (i64 *)calloc((size_t)n0 + 1, sizeof(i64));
    i64 *s12 = (i64 *)calloc((size_t)n02 + 3, sizeof(i64));
    i64 cnt = 0;
    for (i64 i = 0; i < n + n0 - n1; i++)            /* easa.py:163 */
        if (i % 3 != 0) s12[cnt++] = i;

    radixpass(s12, SA12, s + 2, n02, K);              /* easa.py:165-167 */
    radixpass(SA12, s12, s + 1, n02, K);
    radixpass(s12, SA12, s, n02, K);

    i64 name = 0, c0 = -1, c1 = -1, c2 = -1;          /* easa.py:169-182 */
    for (i64 i = 0; i < n02; i++) {
        i64 p = SA12[i];
        if (s[p] != c0 || s[p + 1] != c1 || s[p + 2] != c2) {
            name++;
            c0 = s[p]; c1 = s[p + 1]; c2 = s[p + 2];
        }
        if (p % 3 == 1) s12[p / 3] = name;
        else            s12[p / 3 + n0] = name;
    }

    if (name < n02) {                                 /* easa.py:184-190 */
        kark_sort(s12, SA12, n02, name);
        for (i64 i = 0; i < n02; i++) s12[SA12[i]] = i + 1;
    } else {
        for (i64 i = 0; i < n02; i++) SA12[s12[i] - 1] = i;
    }

    i64 *s0 = (i64 *)calloc((size_t)n0 + 1, sizeof(i64));   /* easa.py:192-194 */
    cnt = 0;
    for (i64 i = 0; i < n02; i++)
        if (SA12[i] < n0) s0[cnt++] = SA12[i] * 3;
    radixpass(s0, SA0, s, n0, K);

    i64 p = 0, k = 0, t = n0 - n1;                    /* easa.py:196-228 */
    while (k < n) {
        i64 i = SA12[t] < n0 ? SA12[t] * 3 + 1 : (SA12[t] - n0) * 3 + 2;
        i64 j = p < n0 ? SA0[p] : 0;
        int test;
        if (SA12[t] < n0) {
            test = (s[i] == s[j]) ? (s12[SA12[t] + n0] <= s12[j / 3]) : (s[i] < s[j]);
        } else if (s[i] == s[j]) {
            test = (s[i + 1] == s[j + 1]) ? (s12[SA12[t] - n0 + 1] <= s12[j / 3 + n0])
                                          : (s[i + 1] < s[j + 1]);
        } else {
            test = s[i] < s[j];
        }
        if (test) {
            SA[k] = i;
            t++;
            if (t == n02) {
                k++;
                while (p < n0) { SA[k] = SA0[p]; p++; k++; }
            }
        } else {
            SA[k] = j;
            p++;
            if (p == n0) {
                k++;
                while (t < n02) {
                    SA[k] = SA12[t] < n0 ? SA12[t] * 3 + 1 : (SA12[t] - n0) * 3 + 2;
                    t++; k++;
                }
            }
        }
        k++;
    }
    free(SA12); free(SA0); free(s12); free(s0);
}

/* _compute_suftab (easa.py:141-153): pad with three chr(1), DC3. */
int easa_suftab(const uint32_t *sym, i64 n, i64 *suftab)
{
    if (n <= 0) return -1;
    if (n == 1) { suftab[0] = 0; return 0; }   /* the reference raises IndexError here */
    i64 *s = (i64 *)malloc(((size_t)n + 3) * sizeof(i64));
    i64 K = 1;
    for (i64 i = 0; i < n; i++) {
        s[i] = (i64)sym[i];
        if (s[i] > K) K = s[i];
        /* the reference pads with chr(1) (easa.py:149): a text symbol <= U+0001 is not smaller than the
         * pad and its DC3 goes wrong (out-of-range suffix numbers) -- outside the reference's domain */
        if (s[i] <= 1) { free(s); return -2; }
    }
    s[n] = s[n + 1] = s[n + 2] = 1;
    kark_sort(s, suftab, n, K);
    free(s);
    return 0;
}

/* _compute_lcptab (easa.py:247-266): Kasai et al.  No bounds check in the
 * reference either -- the unique last terminator stops every comparison. */
int easa_lcptab(const uint32_t *sym, i64 n, const i64 *suftab, i64 *lcptab)
{
    i64 *rank = (i64 *)malloc((size_t)n * sizeof(i64));
    for (i64 i = 0; i < n; i++) rank[suftab[i]] = i;
    memset(lcptab, 0, (size_t)n * sizeof(i64));
    i64 h = 0;
    for (i64 i = 0; i < n; i++) {
        if (rank[i] >= 1) {
            i64 j = suftab[rank[i] - 1];
            while (i + h < n && j + h < n && sym[i + h] == sym[j + h]) h++;
            lcptab[rank[i]] = h;
            if (h > 0) h--;
        }
    }
    free(rank);
    return 0;
}

/* _compute_childtab (easa.py:268-287): Abouelhoda up/down, one stack scan. */
int easa_childtab(const i64 *lcptab, i64 n, i64 *up, i64 *down)
{
    i64 *stack = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    i64 sp = 0, last_index = -1;
    memset(up, 0, (size_t)n * sizeof(i64));
    memset(down, 0, (size_t)n * sizeof(i64));
    stack[sp++] = 0;
    for (i64 i = 0; i < n; i++) {
        while (lcptab[i] < lcptab[stack[sp - 1]]) {
            last_index = stack[--sp];
            if (lcptab[i] <= lcptab[stack[sp - 1]] &&
                lcptab[stack[sp - 1]] != lcptab[last_index])
                down[stack[sp - 1]] = last_index;
        }
        if (last_index != -1) { up[i] = last_index; last_index = -1; }
        stack[sp++] = i;
    }
    free(stack);
    return 0;
}

/* _compute_childtab_next_l_index (easa.py:289-304). */
int easa_next_l_index(const i64 *lcptab, i64 n, i64 *next)
{
    i64 *stack = (i64 *)malloc(((size_t)n + 1) * sizeof(i64));
    i64 sp = 0;
    memset(next, 0, (size_t)n * sizeof(i64));
    stack[sp++] = 0;
    for (i64 i = 0; i < n; i++) {
        while (lcptab[i] < lcptab[stack[sp - 1]]) sp--;
        if (lcptab[i] == lcptab[stack[sp - 1]]) {
            i64 last_index = stack[--sp];
            next[last_index] = i;
        }
        stack[sp++] = i;
    }
    free(stack);
    return 0;
}

/* asts/utils.py:6-11 -- linear scan, no bounds check in the reference. */
static i64 util_index(const i64 *array, i64 n, i64 key, i64 start)
{
    i64 i = start;
    while (i < n && array[i] != key) i++;
    return i;
}

/* _compute_anntab (easa.py:306-331) over traverse_depth_first_post_order
 * (easa.py:57-85).  Stack frames are <l, i, j, children>; children live in a
 * pool of singly linked records kept in ascending order of i. */
typedef struct { i64 l, i, j, head, tail; } frame_t;
typedef struct { i64 l, i, j, next; } child_t;

typedef struct {
    child_t *pool; i64 used, cap;
} pool_t;

static i64 pool_add(pool_t *p, i64 l, i64 i, i64 j)
{
    if (p->used == p->cap) {
        p->cap = p->cap ? p->cap * 2 : 1024;
        p->pool = (child_t *)realloc(p->pool, (size_t)p->cap * sizeof(child_t));
    }
    child_t *c = &p->pool[p->used];
    c->l = l; c->i = i; c->j = j; c->next = -1;
    return p->used++;
}

static void frame_append(pool_t *p, frame_t *f, const frame_t *child)
{
    i64 id = pool_add(p, child->l, child->i, child->j);
    if (f->head < 0) f->head = id; else p->pool[f->tail].next = id;
    f->tail = id;
}

/* process_node (easa.py:314-324) */
static void process_node(const frame_t *node, const pool_t *p, const i64 *lcptab,
                         i64 n, i64 *anntab)
{
    i64 idx = util_index(lcptab, n, node->l, node->i);   /* _interval_index :333-338 */
    i64 i = node->i;
    for (i64 c = node->head; c >= 0; c = p->pool[c].next) {
        const child_t *ch = &p->pool[c];
        if (i < ch->i) anntab[idx] += ch->i - i;
        anntab[idx] += anntab[util_index(lcptab, n, ch->l, ch->i)];
        i = ch->j + 1;
    }
    if (i <= node->j) anntab[idx] += node->j - i + 1;
}

int easa_anntab(const i64 *lcptab, i64 n, i64 m, i64 *anntab)
{
    frame_t *stack = (frame_t *)malloc(((size_t)n + 2) * sizeof(frame_t));
    pool_t pool = {0, 0, 0};
    i64 sp = 0;
    int have_last = 0;
    frame_t last;
    memset(anntab, 0, (size_t)n * sizeof(i64));
    stack[sp++] = (frame_t){0, 0, -1, -1, -1};
    for (i64 i = 1; i < n; i++) {
        i64 lb = i - 1;
        while (lcptab[i] < stack[sp - 1].l) {
            stack[sp - 1].j = i - 1;
            last = stack[--sp]; have_last = 1;
            process_node(&last, &pool, lcptab, n, anntab);
            lb = last.i;
            if (lcptab[i] <= stack[sp - 1].l) {
                frame_append(&pool, &stack[sp - 1], &last);
                have_last = 0;
            }
        }
        if (lcptab[i] > stack[sp - 1].l) {
            frame_t f = {lcptab[i], lb, -1, -1, -1};
            if (have_last) { frame_append(&pool, &f, &last); have_last = 0; }
            stack[sp++] = f;
        }
    }
    stack[sp - 1].j = n - 1;
    process_node(&stack[sp - 1], &pool, lcptab, n, anntab);
    anntab[0] -= m;                                       /* easa.py:329 */
    free(stack); free(pool.pool);
    return 0;
}

/* EnhancedAnnotatedSuffixArray.__init__ pipeline (easa.py:16-24). */
int easa_build(const uint32_t *sym, i64 n, i64 m, i64 *suftab, i64 *lcptab,
               i64 *up, i64 *down, i64 *next, i64 *anntab)
{
    if (easa_suftab(sym, n, suftab)) return -1;
    easa_lcptab(sym, n, suftab, lcptab);
    easa_childtab(lcptab, n, up, down);
    easa_next_l_index(lcptab, n, next);
    easa_anntab(lcptab, n, m, anntab);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Score walk, faithful form (easa.py:91-139 with :340-400).           */
typedef struct {
    const uint32_t *sym; i64 n;
    const i64 *suftab, *lcptab, *up, *down, *next, *anntab;
} easa_t;

typedef struct { i64 l, i, j; int ok; } ival_t;

static i64 lcp_value(const easa_t *e, i64 i, i64 j)       /* easa.py:349-356 */
{
    i64 n = e->n;
    if ((i == 0 || i == n - 1) && j == n - 1) return 0;
    i64 u = e->up[j + 1];
    if (i < u && u <= j) return e->lcptab[u];
    return e->lcptab[e->down[i]];
}

static ival_t get_child_interval(const easa_t *e, i64 i, i64 j, uint32_t ch)  /* :379-400 */
{
    ival_t none = {0, 0, 0, 0};
    if (i == j) return none;
    i64 n = e->n, l = lcp_value(e, i, j), i1;
    if (i == 0 && j == n - 1) {
        i1 = 0;
    } else {
        if (i < e->up[j + 1]) i1 = e->up[j + 1]; else i1 = e->down[i];
        if (e->sym[e->suftab[i] + l] == ch) {
            ival_t r = {lcp_value(e, i, i1 - 1), i, i1 - 1, 1};
            return r;
        }
    }
    while (e->next[i1] != 0) {
        i64 i2 = e->next[i1];
        if (e->sym[e->suftab[i1] + l] == ch) {
            ival_t r = {lcp_value(e, i1, i2 - 1), i1, i2 - 1, 1};
            return r;
        }
        i1 = i2;
    }
    if (e->sym[e->suftab[i1] + l] == ch) {
        ival_t r = {lcp_value(e, i1, j), i1, j, 1};
        return r;
    }
    return none;
}

static i64 annotation(const easa_t *e, const ival_t *v)   /* easa.py:340-347 */
{
    if (v->i == v->j) return 1;
    return e->anntab[util_index(e->lcptab, e->n, v->l, v->i)];
}

/* _score (easa.py:91-139).  q must already have U+0020 removed (score,
 * easa.py:36).  suffix_scores (nullable) receives the qlen per-suffix results
 * in suffix order.  qlen == 0 is the reference's ZeroDivisionError: returns
 * NaN-free 0 and sets *err. */
double easa_score(const uint32_t *sym, i64 n, const i64 *suftab, const i64 *lcptab,
                  const i64 *up, const i64 *down, const i64 *next, const i64 *anntab,
                  const uint32_t *q, i64 qlen, int normalized, double *suffix_scores,
                  int *err)
{
    easa_t e = {sym, n, suftab, lcptab, up, down, next, anntab};
    if (err) *err = 0;
    if (qlen <= 0) { if (err) *err = 1; return 0.0; }
    double result = 0.0;
    for (i64 start = 0; start < qlen; start++) {
        const uint32_t *suf = q + start;
        i64 slen = qlen - start;
        double suffix_score = 0.0, suffix_result = 0.0;
        i64 matched_chars = 0, nodes_matched = 0;
        ival_t parent = {0, 0, n - 1, 1};
        ival_t child = get_child_interval(&e, parent.i, parent.j, suf[0]);
        while (child.ok) {
            nodes_matched++;
            i64 sub_start = suftab[child.i] + parent.l;
            i64 sub_end = (child.i == child.j) ? n : sub_start + child.l - parent.l;
            i64 match = 0;                                 /* match_strings, asts/utils.py:14-22 */
            i64 lim = slen < sub_end - sub_start ? slen : sub_end - sub_start;
            while (match < lim && suf[match] == sym[sub_start + match]) match++;
            suffix_score += (double)annotation(&e, &child) / (double)annotation(&e, &parent);
            matched_chars += match;
            suf += match; slen -= match;
            if (slen > 0 && match == sub_end - sub_start) {
                parent = child;
                child = get_child_interval(&e, parent.i, parent.j, suf[0]);
            } else {
                break;
            }
        }
        if (matched_chars) {
            suffix_result = (suffix_score + (double)matched_chars) - (double)nodes_matched;
            if (normalized) suffix_result /= (double)matched_chars;
            result += suffix_result;
        }
        if (suffix_scores) suffix_scores[start] = suffix_result;
    }
    return result / (double)qlen;
}

/* ------------------------------------------------------------------ */
/* Score walk, interval-narrowing form.  Needs
 * only sym + suftab; mathematically the same walk, with the child lookup
 * done by binary search on the symbol at the current depth instead of the
 * sibling chain.  Validated bit-for-bit against easa_score() and the
 * reference in tests/; used as the checker for inputs where the faithful
 * form's O(m) sibling chains are too slow. */
static i64 sym_at(const uint32_t *sym, i64 n, const i64 *suftab, i64 r, i64 d)
{
    i64 p = suftab[r] + d;
    return p < n ? (i64)sym[p] : -1;
}

double easa_score_fast(const uint32_t *sym, i64 n, i64 m, const i64 *suftab,
                       const uint32_t *q, i64 qlen, int normalized,
                       double *suffix_scores, i64 *probes, int *err)
{
    if (err) *err = 0;
    if (qlen <= 0) { if (err) *err = 1; return 0.0; }
    double total = 0.0;
    i64 nprobe = 0;
    for (i64 start = 0; start < qlen; start++) {
        i64 lo = 0, hi = n - 1, d = 0, nodes = 0;
        double acc = 0.0, r = 0.0;
        for (i64 t = start; t < qlen; t++) {
            i64 c = (i64)q[t], a, b;
            if (lo == hi) {
                nprobe++;
                if (sym_at(sym, n, suftab, lo, d) != c) break;
                a = b = lo;
            } else {
                i64 x = lo, y = hi + 1;                 /* lower bound */
                while (x < y) { i64 mid = (x + y) >> 1; nprobe++;
                    if (sym_at(sym, n, suftab, mid, d) < c) x = mid + 1; else y = mid; }
                a = x; y = hi + 1;                      /* upper bound */
                while (x < y) { i64 mid = (x + y) >> 1; nprobe++;
                    if (sym_at(sym, n, suftab, mid, d) <= c) x = mid + 1; else y = mid; }
                b = x - 1;
                if (a > b) break;
            }
            if (b - a < hi - lo) {
                i64 parent = d == 0 ? n - m : hi - lo + 1;
                acc += (double)(b - a + 1) / (double)parent;
                nodes++;
            }
            lo = a; hi = b; d++;
        }
        if (d > 0) {
            r = (acc + (double)d) - (double)nodes;
            if (normalized) r /= (double)d;
            total += r;
        }
        if (suffix_scores) suffix_scores[start] = r;
    }
    if (probes) *probes = nprobe;
    return total / (double)qlen;
}
