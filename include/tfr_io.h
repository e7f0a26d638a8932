/* tfr_io.h -- C ABI of libtfr_io.so: the input side of the ranking hot path (host code, no GPU).
 *
 * Replaces, for numeric features, the TensorFlow ops the reference's input pipeline is built
 * from (paths relative to /root/reference/tensorflow_ranking/):
 *   - tf.data.TFRecordDataset              (python/data.py:914-1017 build_ranking_dataset_with_parsing_fn)
 *   - tf.io.parse_example on ExampleListWithContext protos: decode, truncate / pad to
 *     list_size, sizes and mask     (python/data.py:59-96, 133-208 _ExampleInExampleParser.parse,
 *                                    383-540 _ExampleListParser / parse_from_example_list)
 *   - the LibSVM loader of the canonical example (examples/tf_ranking_libsvm.py:137-195).
 * Output is what the scorer consumes: dense row-major fp32 [B, list_size, F] with the label as
 * one more feature whose default (-1) marks padding (python/data.py:41, utils.py:78-81).
 *
 * Conventions: all pointers are HOST pointers owned by the caller; nothing is allocated or
 * retained; thread-safe.  Return >= 0 on success, < 0 on error:
 *   TFR_IO_EINVAL -1 bad argument, TFR_IO_ECORRUPT -2 truncated / malformed record or protobuf,
 *   TFR_IO_ECRC -3 checksum mismatch, TFR_IO_ESHAPE -4 a feature is present with a length
 *   different from its spec (tf.io.parse_example raises in that case), TFR_IO_ETYPE -5 a feature
 *   holds a bytes_list (numeric path only).
 */
#ifndef TFR_IO_H_
#define TFR_IO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFR_IO_EINVAL (-1)
#define TFR_IO_ECORRUPT (-2)
#define TFR_IO_ECRC (-3)
#define TFR_IO_ESHAPE (-4)
#define TFR_IO_ETYPE (-5)
#define TFR_IO_EMISSING (-6)   /* ExampleInExample without its `serialized_context` feature */

/* Record formats of python/data.py:45-47 (the numeric FixedLenFeature subset of each). */
#define TFR_IO_FORMAT_ELWC 0   /* ExampleListWithContext                                    (data.py:59-96, 383-540)  */
#define TFR_IO_FORMAT_EIE 1    /* ExampleInExample: serialized_context / serialized_examples (data.py:133-380)         */
#define TFR_IO_FORMAT_SEQ 2    /* tf.SequenceExample: context + one feature_list per feature (data.py:572-855)         */
#define TFR_IO_FORMAT_EXAMPLE 3 /* one tf.Example = a list of ONE item carrying its context features too (data.py:1348-1395) */

/* 2 (round 5 bookkeeping of round 4's additions): the EIE / SequenceExample / single-Example formats, the bf16 batch
 * entry points and TFR_IO_EMISSING.  A binding checks the value at load time (ranking_amd/_io_lib.py does). */
#define TFR_IO_ABI_VERSION 2
int tfr_io_abi_version(void);

/* CRC-32C (Castagnoli) and TFRecord's masked form ((crc >> 15 | crc << 17) + 0xa282ead8). */
uint32_t tfr_io_crc32c(const uint8_t* data, size_t n);
uint32_t tfr_io_masked_crc32c(const uint8_t* data, size_t n);
/* tfr_io_crc32c takes the SSE4.2 crc32 instruction when the host has it; this is the table (slice-by-8) path it
 * falls back to, exported for the cross-check. */
uint32_t tfr_io_crc32c_portable(const uint8_t* data, size_t n);

/* TFRecord framing: [u64 length][u32 masked crc of length][data][u32 masked crc of data].
 * Fills offsets[i] / lengths[i] (payload position and size inside `buf`) for up to max_records
 * records (both arrays may be NULL to count only); returns the number of records in `buf`. */
int64_t tfr_io_tfrecord_index(const uint8_t* buf, size_t nbytes, int verify_crc, uint64_t* offsets,
                              uint64_t* lengths, int64_t max_records);

/* FixedLenFeature([width], float32 | int64, default): values are written as fp32. */
typedef struct tfr_io_feature_spec {
  const char* name;
  int32_t width;        /* number of values per example (>= 1)                         */
  float default_value;  /* used when the feature is absent and for padded examples     */
} tfr_io_feature_spec;

/* Largest number of `examples` among the records (list_size=None in the reference means "pad to
 * the longest list of the batch"). */
int64_t tfr_io_elwc_max_list_size(const uint8_t* const* records, const uint64_t* lengths, int32_t B);

/* Parses B serialized ExampleListWithContext protos.
 *   example_out  [B, list_size, sum(example widths)]  features in spec order
 *   context_out  [B, sum(context widths)]             (nullable when n_context == 0)
 *   sizes_out    [B]  number of examples in the record, BEFORE truncation (data.py:149-152)
 *   mask_out     [B, list_size]  1 for positions < min(size, list_size)       (nullable)
 *   num_threads  <= 1: caller's thread; otherwise records are split across threads. */
int tfr_io_parse_elwc_batch(const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                            int32_t list_size, const tfr_io_feature_spec* example_specs,
                            int32_t n_example, const tfr_io_feature_spec* context_specs,
                            int32_t n_context, float* example_out, float* context_out,
                            int32_t* sizes_out, uint8_t* mask_out, int32_t num_threads);

/* The same parse with the example features rounded to bfloat16 (round to nearest even, NaN preserved: bit for bit the
 * rounding of the scorer's input cast, tfr_tower_cast_gather_f32_bf16): example_out_bf16 [B, list_size, sum(widths)]
 * uint16.  The first hidden layer multiplies bf16 operands anyway; shipping them halves the bytes of the pinned
 * buffer and of the host link (tfr_tower_cast_gather_bf16_bf16 takes them on the device).  Context features, sizes
 * and the mask as above.  f32_columns [n_f32_columns] (nullable with 0): columns of the concatenated example
 * features (labels, real-valued targets) that are ALSO written unrounded to f32_out [B, list_size, n_f32_columns]. */
int tfr_io_parse_elwc_batch_bf16(const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                                 int32_t list_size, const tfr_io_feature_spec* example_specs,
                                 int32_t n_example, const tfr_io_feature_spec* context_specs,
                                 int32_t n_context, uint16_t* example_out_bf16, float* context_out,
                                 int32_t* sizes_out, uint8_t* mask_out, int32_t num_threads,
                                 const int32_t* f32_columns, int32_t n_f32_columns, float* f32_out);

/* The batch parser for any of the record formats above: the arguments of tfr_io_parse_elwc_batch /
 * tfr_io_parse_elwc_batch_bf16 with `format` in front; exactly one of example_out (fp32) and example_out_bf16 is
 * non-NULL (f32_columns / f32_out only with the latter).  ExampleInExample: examples and context are the serialized
 * tf.Examples inside the two bytes features of the outer tf.Example; a record without `serialized_context` is
 * TFR_IO_EMISSING, more than one context TFR_IO_ESHAPE.  SequenceExample: every named example feature is a
 * FixedLenSequenceFeature(allow_missing=True) of the reference's parser -- a missing feature_list has no frames, a
 * frame must hold exactly `width` values (an empty frame is TFR_IO_ESHAPE, also among the frames truncation drops),
 * positions past a feature's own frames take its default, sizes_out = the longest named feature_list.
 * TFR_IO_FORMAT_EXAMPLE: example AND context features are looked up in the one tf.Example; size 1, rows past the first
 * are defaults. */
int tfr_io_parse_batch(int32_t format, const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                       int32_t list_size, const tfr_io_feature_spec* example_specs, int32_t n_example,
                       const tfr_io_feature_spec* context_specs, int32_t n_context, float* example_out,
                       uint16_t* example_out_bf16, float* context_out, int32_t* sizes_out, uint8_t* mask_out,
                       int32_t num_threads, const int32_t* f32_columns, int32_t n_f32_columns, float* f32_out);

/* list_size=None for any format: the largest number of examples (ELWC, EIE) / the longest feature_list among the named
 * example features (SequenceExample; example_specs is only read for that format) over the records. */
int64_t tfr_io_max_list_size(int32_t format, const uint8_t* const* records, const uint64_t* lengths, int32_t B,
                             const tfr_io_feature_spec* example_specs, int32_t n_example);

/* fp32 -> bfloat16 of n values with that rounding (labels, LibSVM features, any staged array). */
void tfr_io_f32_to_bf16(const float* src, uint16_t* dst, size_t n);

/* Process-wide totals of tf.Example messages (examples and contexts) decoded so far by tfr_io_parse_elwc_batch: by
 * replaying the previous example's byte structure (identical unmasked bytes: defaults + payload copies) and by the
 * generic protobuf walk.  Observability only -- the two paths return the same rows.  Either pointer may be NULL. */
void tfr_io_parse_counters(uint64_t* replayed, uint64_t* walked);

/* LibSVM text ("label qid:Q fid:val ... # comment" per line, features named 1..num_features).
 * Pass 1 (features_out == NULL): returns the number of distinct qids (first-seen order).
 * Pass 2: fills features_out [Q, list_size, num_features] (zeros) and labels_out [Q, list_size]
 * (-1 padding), keeping the first list_size documents of each query; stats_out[0] = documents
 * seen, stats_out[1] = documents discarded (nullable).  Returns Q. */
int64_t tfr_io_libsvm_load(const char* text, size_t nbytes, int32_t list_size, int32_t num_features,
                           float* features_out, float* labels_out, int64_t* stats_out);

#ifdef __cplusplus
}
#endif
#endif  /* TFR_IO_H_ */
