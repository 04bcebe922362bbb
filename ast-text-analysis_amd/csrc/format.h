// format.h -- host only: the keyphrase table as text (reference east/formatting.py:14-39, table2xml / table2csv).
//
// At BASELINE configs[2] the table has 2.56 M scores; the device fills it in a millisecond, and a Python loop that formats
// one '%.3f' per score takes seconds.  The renderings are byte for byte the reference's: names are written as given
// ('%s'), scores with printf's "%.3f" -- glibc rounds the exact binary value correctly (ties to even), as Python's '%'
// operator does.  Rows are formatted by a few host threads, each into its own stretch of the output.
#pragma once
#include <string.h>
#include <thread>

static inline char *fmt_put(char *p, const char *s)
{
    const size_t n = strlen(s);
    memcpy(p, s, n);
    return p + n;
}

static inline char *fmt_score(char *p, double v) { return p + snprintf(p, 32, "%.3f", v); }

template <class F> static void fmt_rows_parallel(int32_t n_rows, int64_t work_per_row, F fn)
{
    int n_threads = (int)std::min<int64_t>(std::max<int64_t>((int64_t)n_rows * work_per_row / 200000, 1),
                                           std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    if (n_threads <= 1) { fn(0, n_rows); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back(fn, (int32_t)((int64_t)n_rows * t / n_threads), (int32_t)((int64_t)n_rows * (t + 1) / n_threads));
    for (auto &t : th) t.join();
}

extern "C" {

// table: K x D scores, row-major.  kp_order / text_order: the output order (sorted names) as indices into the table's rows
// and columns; kp_names[k] / text_names[d]: NUL-terminated UTF-8, indexed like the table.  Writes the text into out (cap
// bytes) and returns its length, or the negative number of bytes needed if cap is too small, or EAST_HIP_ERR_INVALID.
int64_t east_hip_format_table_xml(const double *table, int32_t K, int32_t D, const int32_t *kp_order, const int32_t *text_order,
                                  const char *const *kp_names, const char *const *text_names, char *out, int64_t cap)
{
    if (!table || K < 0 || D < 0 || !kp_order || !text_order || !kp_names || !text_names || !out) return EAST_HIP_ERR_INVALID;
    // upper bound per row: the keyphrase lines + per text: '    <text name="' name '">' score '</text>\n'
    int64_t text_bytes = 0;
    for (int32_t d = 0; d < D; d++) text_bytes += (int64_t)strlen(text_names[d]) + 16 + 2 + 32 + 8;
    std::vector<int64_t> row_off((size_t)K + 1, 0);
    for (int32_t r = 0; r < K; r++)
        row_off[r + 1] = row_off[r] + 20 + (int64_t)strlen(kp_names[kp_order[r]]) + 3 + text_bytes + 15;
    const int64_t need = 8 + row_off[K] + 10;
    if (need > cap) return -need;
    std::vector<int64_t> row_len((size_t)K, 0);
    char *body = out + 8;
    fmt_rows_parallel(K, D, [&](int32_t r0, int32_t r1) {
        for (int32_t r = r0; r < r1; r++) {
            char *p = body + row_off[r];
            const double *row = table + (size_t)kp_order[r] * D;
            p = fmt_put(p, "  <keyphrase value=\"");
            p = fmt_put(p, kp_names[kp_order[r]]);
            p = fmt_put(p, "\">\n");
            for (int32_t c = 0; c < D; c++) {
                const int32_t d = text_order[c];
                p = fmt_put(p, "    <text name=\"");
                p = fmt_put(p, text_names[d]);
                p = fmt_put(p, "\">");
                p = fmt_score(p, row[d]);
                p = fmt_put(p, "</text>\n");
            }
            p = fmt_put(p, "  </keyphrase>\n");
            row_len[r] = p - (body + row_off[r]);
        }
    });
    memcpy(out, "<table>\n", 8);
    char *w = body;                                       // close the gaps between the rows (each was given its upper bound)
    for (int32_t r = 0; r < K; r++) {
        if (w != body + row_off[r]) memmove(w, body + row_off[r], (size_t)row_len[r]);
        w += row_len[r];
    }
    w = fmt_put(w, "</table>\n");
    return w - out;
}

// "," + quoted keyphrases; then one row per text: quoted name, scores.  Names come already quoted (the caller's
// _csv_quote); orders as above.
int64_t east_hip_format_table_csv(const double *table, int32_t K, int32_t D, const int32_t *kp_order, const int32_t *text_order,
                                  const char *const *kp_quoted, const char *const *text_quoted, char *out, int64_t cap)
{
    if (!table || K < 1 || D < 0 || !kp_order || !text_order || !kp_quoted || !text_quoted || !out) return EAST_HIP_ERR_INVALID;
    int64_t head = 2;
    for (int32_t k = 0; k < K; k++) head += (int64_t)strlen(kp_quoted[k]) + 1;
    std::vector<int64_t> row_off((size_t)D + 1, 0);
    for (int32_t c = 0; c < D; c++) row_off[c + 1] = row_off[c] + (int64_t)strlen(text_quoted[text_order[c]]) + (int64_t)K * 33 + 2;
    const int64_t need = head + row_off[D] + 2;
    if (need > cap) return -need;
    char *p = out;
    for (int32_t r = 0; r < K; r++) { *p++ = ','; p = fmt_put(p, kp_quoted[kp_order[r]]); }
    *p++ = '\n';
    char *body = p;
    std::vector<int64_t> row_len((size_t)D, 0);
    fmt_rows_parallel(D, K, [&](int32_t c0, int32_t c1) {
        for (int32_t c = c0; c < c1; c++) {
            const int32_t d = text_order[c];
            char *q = body + row_off[c];
            q = fmt_put(q, text_quoted[d]);
            for (int32_t r = 0; r < K; r++) { *q++ = ','; q = fmt_score(q, table[(size_t)kp_order[r] * D + d]); }
            *q++ = '\n';
            row_len[c] = q - (body + row_off[c]);
        }
    });
    char *w = body;
    for (int32_t c = 0; c < D; c++) {
        if (w != body + row_off[c]) memmove(w, body + row_off[c], (size_t)row_len[c]);
        w += row_len[c];
    }
    return w - out;
}

}  // extern "C"
