/*
 * east_hip.h -- C ABI of the MI355X (gfx950) annotated-suffix-array backend.
 *
 * This is the drop-in boundary for the one hot path of EAST
 * (mikhaildubov/AST-text-analysis): building the enhanced annotated suffix
 * array of every document and filling the keyphrase x document score table.
 * The reference has no FFI on this path -- it is a Python class plugin
 * (east/asts/base.py:10-46: subclass AST, set __algorithm__, implement
 * score()).  The entry points below are what that plugin binds through
 * ctypes; INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every function returns 0 on success or a negative EAST_HIP_ERR_* code;
 *     east_hip_last_error() returns a thread-local message for the last failure
 *   - plain pointers and sizes only; the caller owns every host buffer
 *     (C-contiguous), the library owns device memory behind the opaque handle
 *   - calls are blocking unless the name ends in _async; a handle is not
 *     thread-safe, distinct handles are.  The east_hip_debug_* setters change process-wide DEFAULTS of the test
 *     knobs; every build / score / text call copies them once when it starts and runs on that copy to its end, so a
 *     setter never reaches into a call that is already in flight -- on this handle or another;
 *     one HIP stream per handle; every call runs on
 *     the handle's device and restores the calling thread's current HIP device before it returns
 *   - indices are int32 on the device (n_total < 2^31 - 8); the Python side
 *     widens to int64 to match the reference's np.int tables
 *   - symbols are Unicode code points; a symbol >= 0x0A00 is a string
 *     terminator (east/consts.py:23-24, east/asts/utils.py:25-40) and text symbols
 *     are < 0x0A00 -- the reference's own encoding, in which text at or above U+0A00
 *     cannot be told from a terminator.  Text of any script goes through the TAGGED
 *     encoding (east_hip_set_symbol_encoding): terminator i of a document is
 *     EAST_HIP_TERMINATOR_TAG | i, every other symbol a text code point < 0x110000
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     fails with EAST_HIP_ERR_NO_DEVICE
 */
#ifndef EAST_HIP_H
#define EAST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAST_HIP_OK                 0
#define EAST_HIP_ERR_NO_DEVICE     -1  /* no HIP device / bad device ordinal            */
#define EAST_HIP_ERR_INVALID       -2  /* bad argument (null pointer, size, offsets)    */
#define EAST_HIP_ERR_OOM           -3  /* device allocation failed                      */
#define EAST_HIP_ERR_HIP           -4  /* a HIP runtime call or kernel launch failed    */
#define EAST_HIP_ERR_DOMAIN        -5  /* input outside the reference's domain          */
#define EAST_HIP_ERR_NOT_BUILT     -6  /* score/get_tables before a successful build    */
#define EAST_HIP_ERR_INTERNAL      -7  /* self-check failed (a bug)                     */

#define EAST_HIP_TERMINATOR_START 0x0A00u /* east/consts.py:23-24 */
#define EAST_HIP_TERMINATOR_TAG   0x80000000u /* tagged encoding: this bit marks a string terminator */
#define EAST_HIP_SYMBOLS_REFERENCE 0 /* terminators are the code points 0x0A00+i, text is < 0x0A00 (the default) */
#define EAST_HIP_SYMBOLS_TAGGED    1 /* terminators are EAST_HIP_TERMINATOR_TAG|i, text is any code point        */

typedef struct east_hip_index *east_hip_handle_t;

/* Library / device discovery. */
const char *east_hip_version(void);
const char *east_hip_last_error(void);
int east_hip_device_count(void);       /* >= 0, or EAST_HIP_ERR_NO_DEVICE */

/*
 * Create / destroy an index object bound to one device.  reserve_symbols > 0
 * pre-allocates the device arena for builds of up to that many symbols so
 * that later builds do not allocate (0 = allocate lazily at build time).
 */
int east_hip_create(int device, int64_t reserve_symbols, east_hip_handle_t *out);
void east_hip_destroy(east_hip_handle_t h);
/* Forget the index, the keyphrases and the prepared texts but keep the stream and the device memory:
 * the handle then behaves like a new one.  Creating and destroying a handle costs ~4 ms of driver
 * calls -- more than a whole build of a small collection -- so callers that index one collection
 * after the other recycle handles (east/hip_backend.py keeps a small pool). */
int east_hip_reset(east_hip_handle_t h);

/*
 * Batched EASA build: replaces EnhancedAnnotatedSuffixArray.__init__
 * (east/asts/easa.py:16-24: make_unique_endings + _compute_suftab /
 * _compute_lcptab / _compute_childtab* / _compute_anntab) for D documents
 * at once -- i.e. HOT LOOP A of ASTRelevanceMeasure.set_text_collection
 * (east/relevance.py:34-49).
 *
 *   symbols       concatenation of every document's EASA string: for document
 *                 d, its strings in order, string i followed by the terminator
 *                 0x0A00+i (i local to the document) -- exactly the code points
 *                 of the reference's `self.string`
 *   n_total       number of symbols (sum of n_d)
 *   doc_offsets   D+1 offsets into symbols, doc_offsets[0]=0, [D]=n_total
 *   n_strings     D values m_d (strings per document = terminators per document)
 *
 * On success the suffix array, LCP table and annotation table of every document
 * are resident on the device -- everything score() needs.  The three child tables
 * (easa.py:22-23) are only needed by callers that read them: they are built by the
 * first east_hip_get_tables request that asks for them (two more launches, about a
 * quarter of a build; bench.py reports them as child_tables_ms).  (How: a window sort over all
 * suffixes with tie refinement, data-parallel DC3 as the fallback -- DESIGN.md 4;
 * either way the tables are bit for bit what easa.py computes.)  east_hip_build_device takes a
 * DEVICE pointer for `symbols` (doc_offsets / n_strings stay host pointers);
 * the buffer is only read.  east_hip_build's copy to the device: symbols of the reference's
 * encoding fit 16 bits, so from 4 Mi symbols on -- the handle's first such call pins a 24 MiB upload
 * ring, once -- host threads narrow them to 16-bit words into the ring and a kernel
 * widens them on the device (half the bytes over the link; east_hip_build_info [25]) -- a text symbol
 * is below U+0A00, everything from there on is a terminator, whose number the build never reads --,
 * and to BYTES (a quarter of the bytes) while every text symbol lies below 0xFF, as in ASCII word text: a
 * symbol that does not fit starts the upload over with 16-bit words, for that call and the handle's later ones;
 * tagged streams and small inputs take the plain 4-byte copy.
 */
/*
 * Which encoding east_hip_build / east_hip_build_device read on this handle (default: the reference's).
 * With EAST_HIP_SYMBOLS_TAGGED the text may hold code points at or above U+0A00 (Indic scripts, Thai, CJK,
 * Hangul, precomposed Vietnamese, IPA capitals ...).  The text symbols are ordered by code point and the
 * terminators sort ABOVE all of them, as they do for every text the reference is defined on.  The reference
 * itself has no meaningful answer there: its terminators chr(0x0A00+i) fall below such text, and then
 * easa.py loses the root annotation (anntab[0] = -m, negative scores: the bottom-up traversal,
 * easa.py:57-85, is never flushed when the last suffix is not a terminator) or raises IndexError
 * (easa.py:349-356 reads childtab_up[n]), and ast_linear disagrees with ast_naive
 * (tests/golden/high_text.json records all three).  What is computed here is the method itself: the
 * scores of ast_naive, and the tables easa.py gives after an order-preserving renaming of the text
 * alphabet into code points below U+0A00.
 */
int east_hip_set_symbol_encoding(east_hip_handle_t h, int32_t encoding);
int east_hip_build(east_hip_handle_t h, const uint32_t *symbols, int64_t n_total,
                   const int64_t *doc_offsets, const int32_t *n_strings, int32_t n_docs);
int east_hip_build_device(east_hip_handle_t h, const uint32_t *d_symbols, int64_t n_total,
                          const int64_t *doc_offsets, const int32_t *n_strings, int32_t n_docs);

/*
 * The same build straight from raw texts: text preparation on the device.  Replaces, for the
 * whole collection, the chain in front of the AST construction
 *   utils.prepare_text (utf-8 decode with errors='replace' + upper)      east/utils.py:31-34
 *   utils.tokenize ([\w']+ under re.U)                                    east/utils.py:37-38
 *   utils.text_to_strings_collection (tokens with len > 2 and not isdigit,
 *       groups of 3, empty -> [" "])                                       east/utils.py:49-79
 *   asts.utils.make_unique_endings + "".join                              east/asts/utils.py:25-40
 * and then runs east_hip_build on the result.
 *
 *   bytes         the texts concatenated, EVERY text followed by one 0xFF byte
 *   text_offsets  D+1 offsets: text d is bytes[text_offsets[d] .. text_offsets[d+1]-1) + its 0xFF
 *   cp_class      0x0A00 bytes: bit 0 = the code point matches [\w'], bit 1 = str.isdigit()
 *   cp_upper      0x0A00 words: 1:1 upper-case mapping (identity where upper() is not one code point)
 *   word_hi       bitmap of the [\w'] code points in [0x0A00, 0x110000), bit (cp - 0x0A00)
 *   digit_hi      the same for str.isdigit(); hi_upper_from / hi_upper_to: the sorted 1:1 upper-case
 *                 mappings of code points >= 0x0A00
 * Texts whose kept tokens hold characters at or above U+0A00 are built in the tagged encoding (see
 * east_hip_set_symbol_encoding; east_hip_prepared_encoding says which one east_hip_get_prepared returns).
 * The Unicode tables come from the caller's interpreter, so the device agrees with the host's
 * re / str semantics by construction.  east_hip_get_prepared returns what was built:
 * n_total symbols, D+1 symbol offsets, D string counts, and the symbols (each pointer nullable).
 */
int east_hip_build_texts(east_hip_handle_t h, const uint8_t *bytes, int64_t n_bytes,
                         const int64_t *text_offsets, int32_t n_docs, const uint8_t *cp_class,
                         const uint32_t *cp_upper, const uint32_t *word_hi, const uint32_t *digit_hi,
                         const uint32_t *hi_upper_from, const uint32_t *hi_upper_to, int32_t n_hi_upper);
/* The same for texts that lie apart in host memory (text d = lengths[d] bytes at texts[d], no
 * separators): they are uploaded one by one, the caller does not have to join them (for a few large
 * texts the join costs more than the build). */
int east_hip_build_texts_v(east_hip_handle_t h, const uint8_t *const *texts, const int64_t *lengths,
                           int32_t n_docs, const uint8_t *cp_class, const uint32_t *cp_upper,
                           const uint32_t *word_hi, const uint32_t *digit_hi,
                           const uint32_t *hi_upper_from, const uint32_t *hi_upper_to, int32_t n_hi_upper);
int east_hip_get_prepared(east_hip_handle_t h, int64_t *n_total, int64_t *doc_offsets,
                          int32_t *n_strings, uint32_t *symbols);
int east_hip_prepared_encoding(east_hip_handle_t h);   /* EAST_HIP_SYMBOLS_REFERENCE or _TAGGED */
double east_hip_last_prep_ms(east_hip_handle_t h);

/*
 * Copy one document's tables to the host (each pointer nullable, n_d int32
 * values each, positions local to the document): the reference attributes
 * suftab, lcptab, anntab (easa.py:20-24) and childtab_up / childtab_down /
 * childtab_next_l_index (easa.py:268-304).
 */
int east_hip_get_tables(east_hip_handle_t h, int32_t doc, int32_t *suftab, int32_t *lcptab,
                        int32_t *anntab, int32_t *childtab_up, int32_t *childtab_down,
                        int32_t *childtab_next_l_index);

/*
 * Keyphrase x document score table: replaces HOT LOOP B
 * (east/applications.py:43-52 -> relevance.py:51-53 -> easa.py:26-36,91-139).
 *
 *   q_symbols   concatenated code points of the K prepared keyphrases with
 *               U+0020 already removed (easa.py:36)
 *   q_offsets   K+1 offsets into q_symbols; every keyphrase must be non-empty
 *               (the reference raises ZeroDivisionError, easa.py:134)
 *   normalized  non-zero = divide each suffix score by its matched length
 *               (easa.py:128-129), zero = the CLI's -d
 *   out         K x D doubles, row-major (out[k*D + d])
 *   suffix_out  nullable; D x S doubles (S = q_offsets[K]), suffix_out[d*S + s]
 *               = the per-suffix result of the suffix starting at q_symbols[s]
 *               (the values of return_suffix_scores=True, easa.py:132-137)
 */
int east_hip_score_table(east_hip_handle_t h, const uint32_t *q_symbols,
                         const int64_t *q_offsets, int32_t n_keyphrases, int normalized,
                         double *out, double *suffix_out);

/*
 * The same in two steps, for callers that score one keyphrase set repeatedly
 * or want the table to stay in HBM: east_hip_set_keyphrases uploads the
 * keyphrases (host pointers, as above) and keeps them resident;
 * east_hip_score_resident runs the score kernels on the resident index and
 * keyphrases.  d_out (nullable) is a DEVICE pointer receiving the K x D table;
 * without it the table stays in the handle's own buffer.  The _async form only
 * queues the kernels on the handle's stream (pair with east_hip_synchronize).
 */
int east_hip_set_keyphrases(east_hip_handle_t h, const uint32_t *q_symbols,
                            const int64_t *q_offsets, int32_t n_keyphrases);
int east_hip_score_resident(east_hip_handle_t h, int normalized, double *d_out);
int east_hip_score_resident_async(east_hip_handle_t h, int normalized);

/*
 * Synonym-expanded scoring (east/asts/easa.py:27-34 behind relevance.py:51-53 with a synonimizer):
 * the caller expands every keyphrase into its variants (itertools.product of the per-word
 * alternatives), the variants are scored as n_queries ordinary queries and the score of keyphrase g
 * is the maximum over its variants [group_offsets[g], group_offsets[g+1]) -- a segmented max on the
 * device.  out: n_groups x D doubles.  (The reference scores the variants with normalized=True
 * whatever its caller asked for; the Python side passes 1 to reproduce that.)
 */
int east_hip_score_table_grouped(east_hip_handle_t h, const uint32_t *q_symbols,
                                 const int64_t *q_offsets, int32_t n_queries,
                                 const int64_t *group_offsets, int32_t n_groups, int normalized,
                                 double *out);

/*
 * The lcp-intervals of one document for the traversal API (easa.py:38-85, base.py:28-34): for every
 * rank k that is the first l-index of an lcp-interval (anntab[k] > 0, k > 0) left[k] = its left
 * boundary i (the interval is lcptab[k]-[i .. i + anntab[k] - 1]), -1 for every
 * other rank.  n_d int32 values, positions local to the document.
 */
int east_hip_get_lcp_intervals(east_hip_handle_t h, int32_t doc, int32_t *left);

/* Roofline accounting for bench.py: runs the score kernels on the resident index and keyphrases once more
 * and counts what the walks read -- k-gram table entries and binary-search probes (one suffix-array entry +
 * one symbol each), the "probes" of SURVEY.md 8(d).  Not a timed path. */
int east_hip_score_probes(east_hip_handle_t h, int normalized, int64_t *probes);

/*
 * Several devices in one process ("single-process/8-device fits the one-process CLI best").
 * Every document is an independent AST (east/relevance.py:41-46) and every (keyphrase, document) score is independent
 * (east/applications.py:43-52): a GROUP shards a collection at document granularity -- contiguous blocks of documents
 * balanced by size, shard s on devices[s], one handle per shard -- and drives the shards with host threads of its own
 * (ctypes drops the GIL around the call).  No collective on the build path.  east_hip_score_table_multi lets every shard
 * score its documents and assembles the K x D_local blocks with ONE all-gather: RCCL (librccl.so loaded at run time,
 * ncclCommInitAll + a grouped ncclAllGather over xGMI, blocks padded to the widest shard) where every shard has a device
 * of its own, device-to-device copies to the first shard's device otherwise (logical shards sharing a device, no
 * librccl.so; EAST_HIP_GROUP_GATHER=copy|rccl forces either); the K x D table then goes to the host once.  No torch, no
 * process spawn.  `devices` may name a device several times (logical shards: the single-GPU tests).
 *
 *   east_hip_group_build           as east_hip_build (host symbols; `encoding` = EAST_HIP_SYMBOLS_*), sharded by symbols
 *   east_hip_group_build_texts_v   as east_hip_build_texts_v (raw texts, device text preparation), sharded by bytes
 *   east_hip_group_shards          first_doc[n_shards + 1]: shard s holds the documents [first_doc[s], first_doc[s+1]);
 *                                  returns n_shards
 *   east_hip_group_handle          shard s's handle (owned by the group) for east_hip_get_tables & co.; document d of the
 *                                  collection is document d - first_doc[s] of its shard
 *   east_hip_score_table_multi     out: K x D doubles, row-major, D = all documents in their original order
 *   east_hip_group_info            [0] build, [1] score (slowest shard), [2] all-gather + copy to the host: wall ms of the
 *                                  last calls; [3] 1 = RCCL all-gather, 2 = copies; [4] shards
 * A group is not thread-safe (its shards run on threads of its own).
 */
typedef struct east_hip_group *east_hip_group_t;
int east_hip_group_create(const int32_t *devices, int32_t n_shards, east_hip_group_t *out);
void east_hip_group_destroy(east_hip_group_t g);
int east_hip_group_build(east_hip_group_t g, const uint32_t *symbols, int64_t n_total, const int64_t *doc_offsets,
                         const int32_t *n_strings, int32_t n_docs, int32_t encoding);
int east_hip_group_build_texts_v(east_hip_group_t g, const uint8_t *const *texts, const int64_t *lengths, int32_t n_docs,
                                 const uint8_t *cp_class, const uint32_t *cp_upper, const uint32_t *word_hi,
                                 const uint32_t *digit_hi, const uint32_t *hi_upper_from, const uint32_t *hi_upper_to,
                                 int32_t n_hi_upper);
int east_hip_group_shards(east_hip_group_t g, int32_t *first_doc);
east_hip_handle_t east_hip_group_handle(east_hip_group_t g, int32_t shard);
int east_hip_score_table_multi(east_hip_group_t g, const uint32_t *q_symbols, const int64_t *q_offsets,
                               int32_t n_keyphrases, int normalized, double *out);
int east_hip_group_info(east_hip_group_t g, double *out, int32_t cap);
/* Host only (needs no device): the group's sharding rule -- contiguous blocks balanced by size; first_doc[n_shards + 1]. */
int east_hip_debug_shard_documents(const int64_t *sizes, int32_t n_docs, int32_t n_shards, int32_t *first_doc);

/*
 * Host only (needs no device): the keyphrase table as text -- east/formatting.py:14-39, table2xml / table2csv, byte for byte
 * ('%s' names, '%.3f' scores) -- for tables of millions of scores, where a Python loop over the scores takes seconds
 * (BASELINE configs[2]: 2.56 M).  table: K x D doubles, row-major; kp_order / text_order: the output order (the sorted
 * names) as indices into the table's rows / columns; names NUL-terminated UTF-8 indexed like the table (CSV: already
 * quoted).  Returns the length written, minus the bytes needed when cap is too small, or EAST_HIP_ERR_INVALID.
 */
int64_t east_hip_format_table_xml(const double *table, int32_t K, int32_t D, const int32_t *kp_order, const int32_t *text_order,
                                  const char *const *kp_names, const char *const *text_names, char *out, int64_t cap);
int64_t east_hip_format_table_csv(const double *table, int32_t K, int32_t D, const int32_t *kp_order, const int32_t *text_order,
                                  const char *const *kp_quoted, const char *const *text_quoted, char *out, int64_t cap);

/* Block until everything queued on the handle's stream has finished. */
int east_hip_synchronize(east_hip_handle_t h);
/* The handle's hipStream_t (as void*) so callers can record events on it. */
void *east_hip_stream(east_hip_handle_t h);

/*
 * Build facts for reports: fills up to `cap` int64 values and returns how many
 * exist: [0] n_total, [1] n_docs, [2] total strings, [3] text alphabet size,
 * [4] bits per symbol at level 0, [5] DC3 recursion levels, [6] arena bytes,
 * [7] arena high-water bytes, [8] radix passes executed, [9] radix elements
 * moved (sum over passes), [10] bytes of one radix element (key+value) at the
 * widest level, [11] passes and [12] elements with 32-bit keys, [13] passes and
 * [14] elements with 64-bit keys, [15] DC3 levels whose few tied names were
 * ordered directly instead of recursing, [16] suffixes merged (sum over levels), [17] rounds of
 * tie refinement by further windows, [18] 1 if the all-suffix window sort produced the suffix array
 * (no DC3 level ran; [5] is 0 then), [19] elements the refinement rounds ordered inside a workgroup's LDS,
 * [20] 1 when the last radix digit and the placement ran as one pass in LDS (the fused finish), [21] suffixes the
 * first placement left in large tie groups, [22] of how many, [23] 1 when the first-level keys held variable-length
 * (order-preserving) code words instead of fixed-width symbol fields (csrc/ht_code.h), [24] 1 when the first-level sort
 * kept every document inside its own range of ranks (the segmented sort, csrc/radix_sort.h: RsSeg), [25] 1 when
 * east_hip_build's host symbols went up as 16-bit words through the pinned ring (half the bytes over the link; reference
 * encoding, 4 Mi symbols or more), 2 when they went up as bytes (text below 0xFF),
 * [26] refinement rounds run inside one persistent launch (csrc/persist_rounds.h; they count in [17] too).
 */
int east_hip_build_info(east_hip_handle_t h, int64_t *out, int32_t cap);

/* Device-time of the last build / score call in milliseconds (HIP events on
 * the handle's stream around the whole call), or a negative error code. */
double east_hip_last_build_ms(east_hip_handle_t h);
double east_hip_last_score_ms(east_hip_handle_t h);

/*
 * Per-kernel timing for bench.py's roofline leg: when enabled, every kernel the
 * handle launches is bracketed by HIP events on the handle's stream.  The
 * report is text, one line per kernel: "name<TAB>launches<TAB>total_ms".
 * Enabling resets the accumulated sums.  Returns the report length.
 */
int east_hip_profile_enable(east_hip_handle_t h, int on);
/* Restrict the bracketing to the launches of one kernel (the name as it appears in the report; NULL or "" = every
 * kernel again).  Two events per launch cost the stream time -- about 10 % of a 2 ms build with every kernel
 * bracketed --, so a timed region brackets only the kernel it reports on. */
int east_hip_profile_only(east_hip_handle_t h, const char *kernel);
int64_t east_hip_profile_report(east_hip_handle_t h, char *buf, int64_t cap);

/*
 * Kernel-level entry points used by the parity tests (host buffers in/out).
 * They exercise exactly the kernels the build uses.
 *   radix sort: stable LSD sort of (key, value) pairs on key bits [0, bits)
 *   scan:       exclusive prefix sum of uint32
 *   suffix_array: DC3 over a dense alphabet: symbols in [1, sigma], n >= 1
 */
int east_hip_debug_radix_sort_u64(int device, uint64_t *keys, uint32_t *vals, int64_t n, int bits);
int east_hip_debug_radix_sort_u32(int device, uint32_t *keys, uint32_t *vals, int64_t n, int bits);
int east_hip_debug_exclusive_scan(int device, const uint32_t *in, uint32_t *out, int64_t n);
int east_hip_debug_suffix_array(int device, const uint32_t *symbols, int64_t n, uint32_t sigma,
                                int32_t *sa_out, int32_t *levels_out);
/* Test knob: rank arrays larger than this many bytes are filled through the bucketed scatter
 * (default 192 MiB, the Infinity Cache); 0 forces the bucketed path on every input. */
int east_hip_debug_set_rank_bucket_bytes(int64_t bytes);
/* Test knob: 0 skips the all-suffix window sort, so that every build goes through DC3 (the
 * fallback for repetitive inputs); 1 (default) restores it; 2 = window sort on, but built as if
 * the device were short of memory for the tie-refinement rounds ("lean"); 3 = window sort on with
 * 64-bit window keys even where 32 bits suffice (the code path of large inputs, on small ones);
 * 4 / 5 = as 1 / 3 with every radix pass global and the separate placement pass (the fused finish,
 * csrc/window_sort.h: lvl0_finish_kernel, switched off); 6 = as 1 with the fused finish whatever the build's
 * plan says (skewed text, whose large buckets it hands to the refinement rounds); 7 = as 1 with first-level keys of
 * variable-length code words wherever a code can be made (csrc/ht_code.h), 9 = the same without the fused finish,
 * 8 = as 1 without such keys.
 * The test knobs of this section are process-wide defaults (they exist to steer a test run through every code path); a
 * call takes the values that hold when it starts (csrc/common.h: Knobs). */
int east_hip_debug_set_window_sort(int enabled);
/* Test knob: the first-level sort of a shard of several documents keeps every document in its own range pass by pass
 * (csrc/radix_sort.h: RsSeg -- no document number in the keys) -- -1 (default): five documents or more of 32 768 symbols or more on average,
 * 0: never, 1: wherever it can be done (2 .. 65 535 documents of any size). */
int east_hip_debug_set_segmented_sort(int mode);
/* Test knob: 0 = the tie-refinement rounds sort every group with the global radix sort; 1 (default) = groups
 * that fit a workgroup's LDS are sorted there (csrc/lds_group_sort.h), the global sort takes the rest. */
int east_hip_debug_set_persist(int force_large, int max_workgroups);   /* the persistent rounds: (1, n) = the large form (several tiles per workgroup, state in global memory) on every domain, at most n workgroups (0: what the device holds); (0, 0) = default */
int east_hip_debug_set_lds_rounds(int enabled);   /* 0 / 1 (default: the in-LDS rounds also classify the next domain; a domain that fits the chip is finished by one persistent launch) / 2 (in-LDS rounds + the stand-alone classification pass, launch by launch) / 3 (as 1 without the persistent launch) */
/* Test knob: 0 = every build waits for the device's answers (alphabet size, tie groups) as a handle's
 * first build does; 1 (default) = later builds on a handle are queued without waiting, on the strength
 * of what the build before found, and checked by the one read-back at their end. */
int east_hip_debug_set_speculation(int enabled);
/* Test knob: bytes of per-suffix scratch a score call may use (default 1 GiB; 0 restores it); a table over
 * more documents than fit is scored a stretch of documents at a time.  Takes effect at the next
 * east_hip_set_keyphrases / east_hip_score_table. */
int east_hip_debug_set_score_scratch(int64_t bytes);
/* Test knob: workgroups one launch of the score walk may have when the per-keyphrase sums run inside it (default 2^22, far
 * below HIP's limit of 2^32 threads per grid dimension; 0 restores it).  A table over more documents than fit one launch is
 * scored a stretch of documents at a time. */
int east_hip_debug_set_score_grid(int64_t workgroups);
/* Test knob: how east_hip_build_texts[_v] brings the raw text to the device.  -1 (default) = inputs of 8 MiB or more are
 * uploaded in four or five chunks (cut where no token spans the cut) and every chunk is prepared while the next one is on its
 * way; 0 = one upload, one preparation; > 0 = always in chunks of about that many bytes (a few dozen: every fixture goes
 * through the chunk-to-chunk carries). */
int east_hip_debug_set_text_stream(int64_t chunk_bytes);
/* Test knob: how separate texts (east_hip_build_texts_v) reach the device when the preparation is streamed.  A copy out of
 * pageable memory costs ~45 us of set-up, which one large text hides and hundreds of small ones do not: they are copied by
 * a few host threads into a ring of pinned memory and go up slot by slot.  mode -1 (default): four or more texts of less
 * than 8 MiB on average; 0: never; 1: always.  slot_bytes: size of a ring slot (0: the default, 8 MiB). */
int east_hip_debug_set_text_ring(int mode, int64_t slot_bytes);
/* Host only (needs no device): the order-preserving variable-length code csrc/ht_code.h makes for n symbols (in their
 * order) with the given weights -- code[i] = the len[i] bits of symbol i's code word, right-aligned.  EAST_HIP_ERR_DOMAIN
 * if no code with word lengths in [3, 12] exists for them (n < 8, n > 256). */
int east_hip_debug_alphabetic_code(const uint64_t *weights, int32_t n, uint32_t *code, int32_t *len);
/* Host only (needs no device): out[i] = symbols[i] below U+0A00 ? symbols[i] : 0xFFFF -- the 16-bit words east_hip_build
 * sends over the link for host symbols of the reference encoding (every symbol from U+0A00 on is a terminator, whose
 * number the build never reads).  vector != 0: the form the upload's host threads run (AVX2 with streaming stores where
 * the CPU has it), 0: the plain loop.  Returns 1 if the vector form ran, 0 if the loop did, < 0 on bad arguments. */
int east_hip_debug_narrow_symbols(const uint32_t *symbols, int64_t n, uint16_t *out, int vector);
/* The same for the BYTES east_hip_build sends when the text's code points all lie below 0xFF (0xFF on the wire = a
 * terminator): out[i] = symbols[i] < 0xFF ? symbols[i] : 0xFF.  Returns (1 if every symbol fitted -- text below 0xFF,
 * terminators from U+0A00 on --, else 0) + (2 if the AVX2 form ran); < 0 on bad arguments. */
int east_hip_debug_narrow_symbols8(const uint32_t *symbols, int64_t n, uint8_t *out, int vector);
/* Test knob (process-wide): which form of the score path runs (easa.py:91-139).  1 (default) = pair k-gram tables marked
 * off the window keys + the per-keyphrase sums inside the walk kernel; 0 = one filled table, per-suffix results in HBM
 * and a reduction kernel (rounds 1-3); 2 = pair tables with the reduction kernel; 3 = filled table with the sums in the
 * walk; 4 = as 1 with the pair tables also for collections of fewer than 16 documents (which by default keep the filled
 * table: the 8-byte marks cost their build more than a few thousand walks get back); 5 = as 1 with every interval searched
 * down to its last suffix (without the walk's register endgame, score.h: walk_endgame).  Takes effect at the next build
 * (tables) / the next east_hip_set_keyphrases (sums) / the next score call (endgame). */
int east_hip_debug_set_score_path(int mode);
/* Host-only: bytes of device arena a build of n_total symbols / n_docs documents
 * reserves (worst case over inputs).  Needs no device. */
int64_t east_hip_plan_arena_bytes(int64_t n_total, int32_t n_docs);
/* The same for a "lean" build: what a build falls back to when the full plan does not fit the
 * free device memory (no buffers for the tie-refinement rounds). */
int64_t east_hip_plan_arena_bytes_lean(int64_t n_total, int32_t n_docs);

#ifdef __cplusplus
}
#endif
#endif /* EAST_HIP_H */
