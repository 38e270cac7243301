/* satt_io.h - host-side (CPU) C ABI of the input pipeline: TFRecord framing, CRC-32C, tf.train.Example indexing and the
 * target preparation of the training path.  Plain C, no device code, no TensorFlow.  Built into
 * self-attention-tacotron_amd/libsatt_io.so by csrc/build.py (gcc -O3); bound through ctypes by satt_amd/_io.py.
 *
 * What it replaces in the reference (all of it implicit tf.data / TensorFlow C++ there):
 *   - tf.data.TFRecordDataset record framing + CRC check   (reference datasets/ljspeech/dataset.py:100-109, train.py:46-49)
 *   - tf.parse_single_example of the record payload        (reference utils/tfrecord.py:82-104)
 *   - DatasetSource._prepare_target                        (reference datasets/ljspeech/dataset.py:127-167)
 * Conventions: caller-owned buffers, no allocation, no global state, thread-safe and GIL-free (ctypes releases the GIL
 * around every call, which is what lets the Python worker pool of datasets/ljspeech.py scale).  Return values >= 0 are
 * counts; negative values are SATT_IO_E_* codes. */
#ifndef SATT_IO_H
#define SATT_IO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SATT_IO_E_TRUNCATED_HEADER (-1)
#define SATT_IO_E_CORRUPT_LENGTH (-2)
#define SATT_IO_E_TRUNCATED_RECORD (-3)
#define SATT_IO_E_CORRUPT_PAYLOAD (-4)
#define SATT_IO_E_TOO_MANY (-5)
#define SATT_IO_E_MALFORMED (-6)
#define SATT_IO_E_BADARG (-7)
#define SATT_IO_E_IO (-8)          /* a file could not be opened or read: satt_io_last_errno() holds the errno (per thread) */

int satt_io_version(void);
int satt_io_last_errno(void);
/* 1 if the SSE4.2 crc32 instruction is used, 0 for the slicing-by-8 tables (same results) */
int satt_io_crc32c_hw(void);

/* CRC-32C (Castagnoli, reflected polynomial 0x82F63B78; RFC 3720 appendix B.4 vectors).  crc32c(data) =
 * satt_crc32c_extend(0, data, n); extend continues a running checksum. */
uint32_t satt_crc32c(const void* data, size_t n);
uint32_t satt_crc32c_extend(uint32_t crc, const void* data, size_t n);
/* the table (slicing-by-8) path regardless of the CPU: test hook, same results */
uint32_t satt_crc32c_sw(const void* data, size_t n);
/* TFRecord's masked form: rotr15(crc) + 0xa282ead8 */
uint32_t satt_masked_crc32c(const void* data, size_t n);

/* Index the records of a TFRecord file image: uint64 length | uint32 masked_crc(length) | payload | uint32 masked_crc(payload).
 * offsets[i] / lengths[i] = byte offset and size of payload i inside buf.  verify != 0 checks both checksums of every
 * record.  Returns the number of records, or SATT_IO_E_*. */
int64_t satt_tfrecord_index(const uint8_t* buf, size_t n, int verify, int64_t* offsets, int64_t* lengths, int64_t max_records);
/* The same for a FILE: open + read (at most cap bytes into buf) + index in one call, so a reader thread crosses the
 * language boundary once per file.  *nbytes = the file's size; if it exceeds cap nothing is indexed and SATT_IO_E_TOO_MANY is
 * returned (call again with a larger buffer).  SATT_IO_E_IO when the file cannot be opened or read. */
int64_t satt_tfrecord_load(const char* path, int verify, uint8_t* buf, size_t cap, int64_t* nbytes, int64_t* offsets,
                           int64_t* lengths, int64_t max_records);

/* One feature of a serialized tf.train.Example (Example{Features{map<string, Feature>}}). */
typedef struct {
  int64_t name_off, name_len; /* feature name (UTF-8) inside the payload */
  int32_t kind;               /* 1 bytes_list, 2 float_list, 3 int64_list, 0 empty feature */
  int32_t packed;             /* float / int64 lists: 1 = one packed run at val_off (the common encoding), 0 = unpacked values */
  int64_t count;              /* bytes_list: number of values; float_list: number of floats; int64_list: number of varints */
  int64_t val_off, val_len;   /* bytes_list: FIRST value; packed lists: the packed run; unpacked lists: the list message body */
  int64_t first_int;          /* int64_list: its first value (the scalar fields id / target_length / mel_width ...), else 0 */
} satt_example_feature;
/* Fills feats[0..] in wire order; returns the number of features or SATT_IO_E_MALFORMED / SATT_IO_E_TOO_MANY. */
int64_t satt_example_index(const uint8_t* payload, size_t n, satt_example_feature* feats, int64_t max_features);
/* decode `count` varints of a packed (or unpacked: field-1 varint entries) int64 list body into out; returns count or error */
int64_t satt_example_int64s(const uint8_t* body, size_t n, int packed, int64_t* out, int64_t max_out);
/* offsets / lengths of the values of a bytes_list message body (field-1 length-delimited entries) */
int64_t satt_example_bytes(const uint8_t* body, size_t n, int64_t* offsets, int64_t* lengths, int64_t max_out);

/* One utterance of the reference's on-disk layout - `<key>.source.tfrecord` + `<key>.target.tfrecord`, one record each
 * (reference datasets/ljspeech/dataset.py:52-72; VCTK adds speaker_id / age / gender: datasets/vctk/dataset.py:36-38) - read,
 * checked and decoded in ONE call: both files into `arena` (source image first, target image behind it), framing + CRC,
 * Example index, fields by name.  Offsets are byte offsets into arena.  The arrays stay where the files put them: `source`
 * is raw little-endian int64, `mel` raw little-endian float32 [target_length, mel_width] (possibly unaligned). */
typedef struct {
  int64_t src_bytes, tgt_bytes;           /* file sizes; the target image starts at arena + src_bytes */
  int64_t src_records, tgt_records;       /* records in each file (fields below describe the FIRST of each) */
  int64_t id, source_length, speaker_id, age, gender;       /* speaker_id / age / gender: -1 when the record has none */
  int64_t key_off, key_len, text_off, text_len;              /* text_len = 0 when absent */
  int64_t source_off, source_count;       /* int64 values of the `source` bytes field */
  int64_t target_id, target_length, mel_width, mel_off, mel_count;   /* mel_count floats at mel_off */
  int64_t prepared_length;                /* satt_prepared_length(target_length, r) */
} satt_utterance;
/* Returns 0, or SATT_IO_E_* (framing / checksum / protobuf damage; SATT_IO_E_IO: a file cannot be opened or read; SATT_IO_E_BADARG: a required
 * field is missing or mel_count != target_length * mel_width).  SATT_IO_E_TOO_MANY: the arena is too small - out->src_bytes
 * and out->tgt_bytes then hold the file sizes (retry with cap >= their sum). */
int64_t satt_utterance_load(const char* source_path, const char* target_path, int verify, int64_t r, uint8_t* arena, size_t cap,
                            satt_utterance* out);

/* ---- native reader: tf.contrib.data.parallel_interleave(cycle_length, sloppy=False) over one-record files ---------------
 * (reference datasets/ljspeech/dataset.py:100-109, train.py:34-36,46-49).  `workers` POSIX threads run satt_utterance_load on
 * the submitted file pairs; results are delivered strictly in SUBMISSION order.  The reader owns `slots` read buffers (the
 * only allocations of this library: made by create / grown on demand, freed by destroy).  A delivered utterance LEASES its
 * slot - the offsets of satt_utterance point into *arena - until satt_reader_release; submit refuses (returns
 * SATT_IO_E_TOO_MANY) while the slot its ticket maps to is still queued, running, done-but-undelivered or leased, so
 * slots must exceed the number of utterances the consumer holds at once (a batch) plus the look-ahead it wants.
 * Thread-safety: one submitting / consuming thread; the Python GIL is never needed by the workers. */
typedef struct satt_reader satt_reader;
satt_reader* satt_reader_create(int workers, int slots, size_t arena_bytes, int verify, int64_t r);
/* ticket (>= 0, sequential), SATT_IO_E_TOO_MANY when no slot is free, SATT_IO_E_BADARG */
int64_t satt_reader_submit(satt_reader* rd, const char* source_path, const char* target_path);
/* Blocks until the OLDEST undelivered ticket is finished.  Returns its ticket and fills *out / *arena / *status (0 or the
 * SATT_IO_E_* code of satt_utterance_load for that pair); SATT_IO_E_BADARG when nothing is outstanding. */
int64_t satt_reader_next(satt_reader* rd, satt_utterance* out, uint8_t** arena, int64_t* status);
int satt_reader_release(satt_reader* rd, int64_t ticket);
int64_t satt_reader_outstanding(const satt_reader* rd);      /* submitted and not yet delivered */
void satt_reader_destroy(satt_reader* rd);                    /* waits for running loads; leased buffers die with it */

/* DatasetSource._prepare_target + the mel row of group_by_batch's padding in one pass (reference
 * datasets/ljspeech/dataset.py:127-167,264-281): out[rows_out, width] receives r silence frames, (mel - avg) / std for the
 * T input frames, and silence up to rows_out.  avg / std hold 1 or `width` entries.  Returns the PREPARED target length
 * (T + 2r rounded up to the next multiple of r when it is not one) or SATT_IO_E_BADARG (rows_out smaller than that,
 * a std entry <= 0, table sizes other than 1 / width). */
int64_t satt_prepare_mel(const float* mel, int64_t T, int64_t width, const float* avg, int64_t navg, const float* std_,
                         int64_t nstd, int64_t r, float silence, float* out, int64_t rows_out);
/* prepared length only */
int64_t satt_prepared_length(int64_t T, int64_t r);

#ifdef __cplusplus
}
#endif
#endif
