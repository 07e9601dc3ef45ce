/*
 * ecgbyte.h -- C ABI of libecgbyte_hip.so, the MI355X (gfx950) implementation of the
 * ECG-Byte tokenizer hot path:  (B,12,L) float64 signal -> 26-symbol stream -> BPE token ids
 * -> LLM input sequences.
 *
 * Every entry point is `extern "C"`, takes plain pointers/sizes and returns an `int` status
 * (ECGB_OK or a negative ECGB_ERR_*); nothing throws across the boundary.  Pointers named
 * `*_dev` are device (HBM) pointers, all others are host pointers.  The caller owns every
 * buffer.  `stream` is a `hipStream_t` passed as `void*` (NULL = the default stream); device
 * entry points only enqueue work on it and never synchronise.  A tokenizer handle is
 * immutable after creation, so concurrent calls on different streams are safe as long as
 * each call brings its own `scratch_dev`.
 *
 * What each entry point replaces in the reference (paths relative to the reference root):
 *   ecgb_tokenizer_create      trie construction inside rust_bpe.encode_text,
 *                              ecg_byte/rust_bpe/src/lib.rs:153-161 (built per CALL there,
 *                              once per tokenizer here)
 *   ecgb_quantize_hip          normalize_all, ecg_byte/utils/tokenizer_utils.py:14-19
 *   ecgb_encode_hip            rust_bpe.encode_text, ecg_byte/rust_bpe/src/lib.rs:149-193
 *   ecgb_quantize_encode_hip   the per-sample front end of ECGTokenDataset.__getitem__,
 *                              ecg_byte/data_loader.py:74-76 (normalize_all -> join -> encode_text)
 *   ecgb_assemble_hip          ECGTokenDataset._prepare_training / _prepare_inference,
 *                              ecg_byte/data_loader.py:91-132 (+ the id->LLM-id mapping of :80)
 *   ecgb_bpe_train_hip         rust_bpe.byte_pair_encoding, ecg_byte/rust_bpe/src/lib.rs:58-125
 * INTEGRATION.md shows the binding a reference maintainer would add for each.
 */
#ifndef ECGBYTE_H
#define ECGBYTE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECGB_OK 0
#define ECGB_ERR_INVALID (-1)     /* bad argument (NULL pointer, zero size, bad range) */
#define ECGB_ERR_NOMEM (-2)       /* host or device allocation failed */
#define ECGB_ERR_UNSUPPORTED (-3) /* input outside the documented limits of this build */
#define ECGB_ERR_HIP (-4)         /* a HIP runtime call failed; see ecgb_last_error() */
#define ECGB_ERR_NODEVICE (-5)    /* no usable gfx950 device */

#define ECGB_ALPHABET 26 /* tokenizer_utils.py:12 */

typedef struct ecgb_tokenizer ecgb_tokenizer; /* opaque */

/* Human-readable description of the last error raised on the calling thread. */
const char *ecgb_last_error(void);

/* Library / ABI version (major<<16 | minor). */
uint32_t ecgb_version(void);

/* ---- tokenizer handle -------------------------------------------------------------------
 * `merges` in the reference is list[(list[int] expansion_bytes, int id)]; here flattened:
 * expansion i = flat_bytes[offsets[i] .. offsets[i+1]), token id = ids[i], i in list order
 * (later duplicates overwrite earlier ones, lib.rs:145).  The 256 single-byte tokens
 * (lib.rs:155-157) are implied.  Builds the trie on the host, lays it out for the device
 * (8 bytes per node, same-class chains numbered consecutively) and uploads it.  With no GPU present the handle is
 * host-only: introspection works, device entry points return ECGB_ERR_NODEVICE.
 * Limits of this build: at most 29 distinct byte values across all expansions plus
 * 'a'..'z'; < 65535 trie nodes; token ids < 65535.  Violations -> ECGB_ERR_UNSUPPORTED. */
int ecgb_tokenizer_create(const uint32_t *flat_bytes, const uint32_t *offsets,
                          const uint32_t *ids, size_t n_merges, ecgb_tokenizer **out);
void ecgb_tokenizer_destroy(ecgb_tokenizer *tok);
/* Introspection: number of trie nodes / deepest path (longest expansion) / symbol classes. */
int ecgb_tokenizer_info(const ecgb_tokenizer *tok, uint32_t *n_nodes, uint32_t *max_depth,
                        uint32_t *n_classes);

/* Copies up to `cap` packed trie nodes (host copy; layout in DESIGN.md) and returns the node
 * count.  Works on a host-only handle (no GPU present). */
size_t ecgb_tokenizer_copy_nodes(const ecgb_tokenizer *tok, uint64_t *out, size_t cap);
/* Copies up to `cap` words of the per-node bit tables the run step of the encoder uses (word pair k:
 * [2k] bit u = node 32k+u has a continuation child, [2k+1] bit u = node 32k+u carries a token ITSELF (a
 * node's record holds the best token of its path); padded
 * with zero pairs) and returns the word count. */
size_t ecgb_tokenizer_copy_runbits(const ecgb_tokenizer *tok, uint32_t *out, size_t cap);

/* ---- quantiser --------------------------------------------------------------------------
 * normalize_all for n float64 samples: sym_dev[i] = alphabet index 0..25 ('a'+index is the
 * reference's character).  If clipped_dev != NULL it also receives the reference's first
 * return value (clip((x-a)/d, 0, 1), float64).  Bit-exact with the reference for every
 * finite and infinite input; NaN maps to index 0 (numpy's NaN->uint8 cast is unspecified). */
int ecgb_quantize_hip(const double *signal_dev, size_t n, double percentile_1,
                      double percentile_99, uint8_t *sym_dev, double *clipped_dev,
                      void *stream);

/* The 25 exact step positions the device quantiser compares against: thr[k-1] (k = 1..25) is the
 * smallest float64 x whose reference level is >= k, found on the host by bisection over float64
 * bit patterns with the reference's own operation sequence.  Returns ECGB_ERR_UNSUPPORTED for
 * degenerate parameters (non-positive or non-finite scale), for which the device falls back to
 * the literal-division kernel.  Host-only; exposed so the staircase can be tested without a GPU. */
int ecgb_quantizer_thresholds(double percentile_1, double percentile_99, double *thr25);

/* ---- encoder ----------------------------------------------------------------------------
 * Scratch the encode entry points need for a batch of `batch` streams of `n_per_stream`
 * symbols each (device bytes). */
size_t ecgb_encode_scratch_bytes(const ecgb_tokenizer *tok, size_t batch, size_t n_per_stream);

/* Greedy longest-match encode (lib.rs:163-190) of `batch` independent byte streams.
 * text_dev: batch x n_per_stream raw bytes (what `text.as_bytes()` is in the reference).
 * ids_dev:  batch x ids_stride uint32; stream b's tokens start at ids_dev + b*ids_stride.
 *           Tokens beyond ids_stride are not written (counts still report the full length).
 * counts_dev: batch uint32, the full token count of each stream. */
int ecgb_encode_hip(const ecgb_tokenizer *tok, const uint8_t *text_dev, size_t batch,
                    size_t n_per_stream, uint32_t *ids_dev, size_t ids_stride,
                    uint32_t *counts_dev, void *scratch_dev, size_t scratch_bytes,
                    void *stream);

/* Which of the two encode kernels a call uses: 0 = automatic (wave-per-stream for batches of at
 * least 2 x CUs, workgroup-per-stream below), 1 = always workgroup-per-stream, 2 = always
 * wave-per-stream (falls back to workgroup-per-stream for tokenizers with an expansion longer than
 * 225 bytes), 3 = wave-per-stream with 8 waves per CU and segments of up to 8064 symbols, 4 = (ecgb_quantize_encode_hip, records
 * of <= 65 535 samples, trie in LDS; the others as 0) one wave per record with one lane per long chunk of run-length entries
 * (encode_long_kernel: 2.6 x fewer loop trips, measured slower end to end -- opt-in; ecgb_encode_scratch_bytes() asks for its larger
 * scratch only while this mode is set), 5 = 0.  All produce identical output; the switch exists for tests and tuning.
 * Process-wide, not thread-safe against concurrent encode calls. */
int ecgb_set_encode_plan(int mode);

/* Fused front end: quantise `batch` records of n_per_record float64 samples (a (12,L) record
 * is read in C order = lead-major, data_loader.py:75) and encode each record's symbol
 * stream.  Same output convention as ecgb_encode_hip. */
int ecgb_quantize_encode_hip(const ecgb_tokenizer *tok, const double *signal_dev, size_t batch,
                             size_t n_per_record, double percentile_1, double percentile_99,
                             uint32_t *ids_dev, size_t ids_stride, uint32_t *counts_dev,
                             void *scratch_dev, size_t scratch_bytes, void *stream);

/* ---- sequence assembly -------------------------------------------------------------------
 * ECGTokenDataset._prepare_training (inference == 0, data_loader.py:101-132) or
 * _prepare_inference (inference != 0, data_loader.py:91-99) for a whole batch.
 * ids_dev/ids_stride/counts_dev: encoder output as produced by ecgb_*encode_hip.
 * lut_dev[k] = LLM token id of tokenizer id k ("signal_{k}", data_loader.py:80).
 * q_ids/a_ids: concatenated question / answer LLM ids with batch+1 offsets each.
 * Training: row_len must be pad_to_max + 4 and every sample needs len(Q)+len(A) <= pad_to_max
 * (the reference asserts, data_loader.py:123; the caller checks before the launch).
 * Outputs are batch x row_len: input_ids (int64), attn_mask (float32, 0 where the id equals
 * pad_id), labels (int64, -100 before the answer) and position_ids (int64); inference
 * writes input_ids, attn_mask and lengths_dev only (rows padded with pad_id). */
int ecgb_assemble_hip(const uint32_t *ids_dev, size_t ids_stride, const uint32_t *counts_dev,
                      size_t batch, const int32_t *lut_dev, size_t lut_len,
                      const int32_t *q_ids_dev, const uint32_t *q_offsets_dev,
                      const int32_t *a_ids_dev, const uint32_t *a_offsets_dev,
                      int32_t pad_id, int32_t bos_id, int32_t eos_id, int32_t sig_start_id,
                      int32_t sig_end_id, uint32_t pad_to_max, int inference, uint32_t row_len,
                      int64_t *input_ids_dev, float *attn_mask_dev, int64_t *labels_dev,
                      int64_t *position_ids_dev, uint32_t *lengths_dev, void *stream);

/* ---- tokenizer training ------------------------------------------------------------------
 * rust_bpe.byte_pair_encoding (lib.rs:58-125) on the device: up to num_merges rounds of
 * { most frequent adjacent pair; replace it left to right, non-overlapping, by id 256+i }.
 * Stops early when no pair is left (lib.rs:88-90).  Tie-break among equal counts: numerically
 * smallest (left, right) -- the reference's is hash-order/schedule dependent.
 * text_dev: n bytes.  pairs_dev: 2*num_merges uint32, (left,right) of merge i at [2i],[2i+1]
 * (the caller derives vocab strings / byte expansions, lib.rs:101-110).  n_done_dev: merges
 * performed.  ids_out_dev: n uint32, the final ids; n_ids_dev: their number.  Everything is
 * enqueued on `stream`; no host synchronisation.  Positions and pair counts are 64-bit: the corpus
 * may be longer than 2^32 symbols (the reference concatenates up to 200 000 records x 30 000 symbols
 * into one string, ecg_byte/utils/tokenizer_utils.py:79-93, ecg_byte/preprocess/sample_ecg.py:15). */
size_t ecgb_bpe_train_scratch_bytes(size_t n, uint32_t num_merges);
int ecgb_bpe_train_hip(const uint8_t *text_dev, size_t n, uint32_t num_merges, uint32_t *pairs_dev,
                       uint32_t *n_done_dev, uint32_t *ids_out_dev, uint64_t *n_ids_dev,
                       void *scratch_dev, size_t scratch_bytes, void *stream);

/* Tests and tuning: the number of workgroups of the trainer's count and rewrite passes (0 = the default, 1 280: five per CU).  Each workgroup walks a contiguous
 * range of 2 048-id tiles; a small number makes ranges of many tiles out of a small corpus.  Every value gives the same merges and ids.  Process-wide; applies to
 * trainers started afterwards (ecgb_bpe_train_hip, ecgb_bpe_shard_create).  ECGB_ERR_INVALID outside 0 .. 1 280. */
int ecgb_set_bpe_train_grid(int workgroups);

/* Tests and tuning: which merge step ecgb_bpe_train_hip runs.  0 (default): every workgroup keeps its range of ids in a fixed slot of the buffer and compacts inside it --
 * ONE pass over the ids per merge, ids 16 bits wide while 256 + num_merges <= 65 536; 1: the same with 32-bit ids; 2: round 4's count pass + rewrite over a globally
 * compacted buffer (two passes; what the sharded form runs).  Every value gives the same merges and ids.  Process-wide.  ECGB_ERR_INVALID outside 0 .. 2. */
int ecgb_set_bpe_train_form(int form);
/* Tests and tuning (forms 0 / 1): 0 (default) = the arg-max's row maxima are a launch of their own between two merges (round 5); 1 = those of merge i + 1 run inside merge
 * i's launch (its first workgroups, once every workgroup of the launch has flagged that its count deltas are in the table) -- ONE launch per merge (lib.rs:85-110: pick the
 * pair, merge).  The same merges and ids.  Measured on the C2 corpus (EXPERIMENTS.md R6): 30.2 us a merge against 29.5 -- the meeting inside the launch (a flag store, a poll
 * and table reads that all have to go past the XCD's L2) costs what the launch boundary did, so it is not the default.  Taken only while the merge launch has at most one
 * workgroup a CU (its workgroups wait for each other).  Process-wide. */
int ecgb_set_bpe_train_fused(int on);

/* ---- tokenizer training on a corpus sharded over ranks (one process per GPU) ---------------------------
 * The reference trains on ONE string, the concatenation of every sampled record (tokenizer_utils.py:79-93), so pairs -- and merges --
 * straddle record joins.  Here rank r holds a contiguous slice of that string; the result (merges, and the concatenation of the
 * ranks' final ids) is the single-device trainer's.  Every rank keeps the whole pair table (the arg-max needs no exchange) and per
 * merge the host enqueues:
 *     ecgb_bpe_shard_pick    arg-max, survivor counts (one record per workgroup range), this slice's 8-word summary   -> all-gather the summaries (8 x world int64)
 *     ecgb_bpe_shard_merge   neighbours from the gathered summaries (the id before the slice, the three after it, the parity of
 *                            a run of `left` entering it), rewrite + compaction, count deltas into the 6 x V slab
 *                                                                                             -> all-reduce (SUM) the slab (int64)
 *     ecgb_bpe_shard_apply   table += slab, slab = 0
 * after ecgb_bpe_shard_begin (bytes -> ids, first summary; all-gather) and ecgb_bpe_shard_count (initial histogram; all-reduce the
 * table once).  Nothing is read back by the host inside the loop; the collectives are the caller's (RCCL through torch.distributed
 * in ecg_byte_amd/trainer.py).  scratch: ecgb_bpe_train_scratch_bytes(n_local, num_merges) bytes of device memory, alive until
 * ecgb_bpe_shard_destroy.  ecgb_bpe_shard_table / _slab return the device addresses (and word counts) of the two buffers to reduce. */
typedef struct ecgb_bpe_shard ecgb_bpe_shard;
ecgb_bpe_shard *ecgb_bpe_shard_create(size_t n_local, uint32_t num_merges, void *scratch_dev, size_t scratch_bytes);
void ecgb_bpe_shard_destroy(ecgb_bpe_shard *h);
void *ecgb_bpe_shard_table(ecgb_bpe_shard *h, size_t *n_words);
void *ecgb_bpe_shard_slab(ecgb_bpe_shard *h, size_t *n_words);
int ecgb_bpe_shard_begin(ecgb_bpe_shard *h, const uint8_t *text_dev, long long *summary_dev, void *stream);
int ecgb_bpe_shard_count(ecgb_bpe_shard *h, const long long *gathered_dev, int rank, int world, void *stream);
int ecgb_bpe_shard_pick(ecgb_bpe_shard *h, uint32_t merge_index, long long *summary_dev, void *stream);
int ecgb_bpe_shard_merge(ecgb_bpe_shard *h, uint32_t merge_index, const long long *gathered_dev, int rank, int world, void *stream);
int ecgb_bpe_shard_apply(ecgb_bpe_shard *h, void *stream);
int ecgb_bpe_shard_finish(ecgb_bpe_shard *h, uint32_t *pairs_dev, uint32_t *n_done_dev, uint32_t *ids_out_dev, uint64_t *n_ids_dev,
                          void *stream);

/* ---- offline conditioning of raw records (float64, a whole batch per call) -------------------------------
 * ecg_byte/utils/preprocess_utils.py:66-88 advanced_ecg_filter: scipy.signal.filtfilt(b, a, x, axis=0) -- method 'pad', padtype 'odd',
 * padlen 3 * max(len a, len b), initial state lfilter_zi * edge sample, direct form II transposed both ways -- for up to four filters
 * applied one after the other (the reference: notch 50 Hz, notch 60 Hz, Butterworth band-pass 0.5-100 Hz order 4, high-pass 0.05 Hz
 * order 4).  x_dev, y_dev: [records, n, leads] as wfdb.rdsamp returns them (preprocess_utils.py:126); may alias.
 * n_taps[k] = coefficients of filter k (2..9, len b == len a); b, a: [n_filters][9] host arrays, zi: [n_filters][8] =
 * scipy.signal.lfilter_zi(b, a).  Scratch: ecgb_filtfilt_scratch_bytes(records, n, leads, 3 * max taps). */
size_t ecgb_filtfilt_scratch_bytes(int records, int n, int leads, int max_edge);
int ecgb_filtfilt_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int n_filters, const int *n_taps,
                      const double *b, const double *a, const double *zi, double *scratch_dev, size_t scratch_bytes, void *stream);

/* ecg_byte/utils/preprocess_utils.py:90-101 nsample_ecg: scipy.interpolate.interp1d(t, y, kind='cubic') -- the not-a-knot cubic spline
 * through all n samples -- evaluated at m equally spaced instants over the same span (the reference: 5000 samples at 500 Hz -> 2500 at
 * 250 Hz).  x_dev [records, n, leads] -> y_dev [records, m, leads], float64; n >= 4.  Scratch: ecgb_resample_cubic_scratch_bytes (n doubles per
 * sequence, sequences rounded up to whole waves of 64). */
size_t ecgb_resample_cubic_scratch_bytes(int records, int n, int leads);
int ecgb_resample_cubic_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int m, double *scratch_dev,
                            size_t scratch_bytes, void *stream);

/* ecg_byte/utils/preprocess_utils.py:43-64 wavelet_denoise: four-level db6 decomposition (pywt.wavedec, mode 'symmetric'), threshold =
 * median(|cD4|) / 0.6745, soft threshold of every detail band (zero where |c| <= epsilon, the reference passes 1e-10), pywt.waverec,
 * NaN / inf -> 0.  PyWavelets is not installed here: restated from its published algorithm, parity UNPINNED.
 * x_dev, y_dev: [records, n, leads] float64, n even and >= 96; may alias.  Scratch: ecgb_wavelet_denoise_scratch_bytes. */
size_t ecgb_wavelet_denoise_scratch_bytes(int records, int n, int leads);
int ecgb_wavelet_denoise_f64(const double *x_dev, double *y_dev, int records, int n, int leads, double epsilon, double *scratch_dev,
                             size_t scratch_bytes, void *stream);
/* Two kernels form this stage with the same bits: one workgroup per sequence with the bands in LDS (default, n up to ~10 200) and one lane per sequence through
 * the scratch (longer sequences).  on = 0 takes the second on every shape (tests compare them). */
void ecgb_set_wavelet_workgroup_kernel(int on);

/* The three stages back to back (process_instance, preprocess_utils.py:142-151) with the intermediates SEQUENCE-MAJOR, [records * leads][n]: the layout the
 * workgroup-per-sequence wavelet kernel reads and writes with whole lines (a sequence of [records][n][leads] is 8 bytes every 8 * leads).  Same arithmetic, same bits
 * as the three calls above; only where the intermediates lie differs.
 *   ecgb_filtfilt_planar_f64        x [records, n, leads] -> y [records * leads][n]   (x and y must not alias)
 *   ecgb_wavelet_denoise_planar_f64 x, y [records * leads][n] (may alias); n even, 96 <= n <= ~10 200 (the bands of a sequence live in LDS), else ECGB_ERR_UNSUPPORTED
 *   ecgb_resample_cubic_planar_f64  x [records * leads][n] -> y [records, m, leads]; out_lead (host, may be NULL): lead l of the input becomes lead out_lead[l] of
 *                                   the output -- the MIMIC lead reorder (preprocess_utils.py:35-40) folded into the store; a permutation of 0 .. leads-1, leads <= 32
 * flags_dev (may be NULL): [records] bytes, zeroed by the caller; set to 1 where the stage wrote a value that is not finite (check_nan_inf's test,
 * preprocess_utils.py:26-33, without another pass over the data; the wavelet stage zeroes such values itself, line 62).  raw_flags_dev (may be NULL): the same for
 * the record as it came in -- process_instance's `np.isnan(signal).any()` skip (134-136; any value that is not finite raises it), read off the first filter's loads. */
int ecgb_filtfilt_planar_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int n_filters, const int *n_taps,
                             const double *b, const double *a, const double *zi, double *scratch_dev, size_t scratch_bytes,
                             unsigned char *flags_dev, unsigned char *raw_flags_dev, void *stream);
int ecgb_wavelet_denoise_planar_f64(const double *x_dev, double *y_dev, int records, int n, int leads, double epsilon, void *stream);
int ecgb_resample_cubic_planar_f64(const double *x_dev, double *y_dev, int records, int n, int leads, int m, const int *out_lead,
                                   double *scratch_dev, size_t scratch_bytes, unsigned char *flags_dev, void *stream);

/* ecg_byte/utils/preprocess_utils.py:26-33 check_nan_inf's test `np.isfinite(data).all()` and process_instance's test of the raw record (134-136), for a batch:
 * flags_dev[r] = 1 if record r (per_record consecutive doubles) holds a NaN or an infinity; flags_dev must be zero on entry.  One pass at memory speed. */
int ecgb_nonfinite_records_f64(const double *x_dev, int records, size_t per_record, unsigned char *flags_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ECGBYTE_H */
