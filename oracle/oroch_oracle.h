/*
 * oroch_oracle.h -- CPU restatement of Oroch's block integer-compression path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared
 * against; nothing that ships (the HIP library, the host headers) may call
 * into it.  It restates, in plain C and in width-generic unsigned arithmetic,
 * the algorithm of the reference headers (citations are to files below
 * /root/reference/oroch).  It is pinned against the real reference compiled
 * from its own headers (oracle/_ref, see oracle/Makefile) and against the
 * golden fixtures generated from that build (tests/golden).
 *
 * Element types are described by a dtype code:
 *     dtype = 2 * log2(sizeof T) + (unsigned ? 1 : 0)
 *     0 int8  1 uint8  2 int16  3 uint16  4 int32  5 uint32  6 int64  7 uint64
 */
#ifndef OROCH_ORACLE_H_
#define OROCH_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* encoding_t, integer_codec.h:46-54 */
enum {
	ORC_NAUGHT = 0,
	ORC_NORMAL = 1,
	ORC_VARINT = 2,
	ORC_VARFOR = 3,
	ORC_BITPCK = 4,
	ORC_BITFOR = 5,
	ORC_BITPFR = 6,
};

/*
 * Flat image of encoding_metadata<T> (integer_codec.h:95-126).  The same
 * record layout is used by the reference shim (oracle/ref_shim.cc) so that
 * the two can be compared field by field.
 */
typedef struct orc_meta {
	uint32_t encoding;   /* value_desc.encoding */
	uint32_t nbits;      /* value_desc.nbits */
	uint64_t origin;     /* value_desc.origin, raw bits zero-extended to 64 */
	uint64_t dataspace;  /* value_desc.dataspace */
	uint64_t metaspace;  /* metaspace() = 1 + value_desc.metaspace */
	uint64_t noutliers;
	uint32_t index_encoding; /* outlier_index_desc.encoding */
	uint32_t index_nbits;    /* outlier_index_desc.nbits */
	uint32_t value_encoding; /* outlier_value_desc.encoding */
	uint32_t value_nbits;    /* outlier_value_desc.nbits */
} orc_meta;

/* bitpck.h:54-71 */
size_t orc_bitpck_capacity(size_t nbits);
size_t orc_bitpck_space(size_t nvalues, size_t nbits);

/* varint.h:66-83 for an unsigned value of any width */
size_t orc_varint_space_u64(uint64_t u);

/* Whole-sequence codecs with caller-supplied parameters. */

/* bitpck.h:227-236 with the default value codec (zigzag for signed T). */
size_t orc_bitpck_encode(int dtype, const void *src, size_t n, unsigned nbits, uint8_t *dst);
/* bitpck.h:238-256 */
size_t orc_bitpck_decode(int dtype, const uint8_t *src, size_t n, unsigned nbits, void *dst);
/* bitfor.h:59-63 (origin codec, no zigzag) */
size_t orc_bitfor_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			 uint8_t *dst);
/* bitfor.h:65-69 */
size_t orc_bitfor_decode(int dtype, const uint8_t *src, size_t n, uint64_t origin, unsigned nbits,
			 void *dst);
/*
 * bitpfr.h:95-99.  indices[] receives the gap-coded exception positions
 * (size_t in the reference), values[] the high parts (u >> nbits), both need
 * room for n entries.  Returns the number of packed bytes; *nexc the number
 * of exceptions.
 */
size_t orc_bitpfr_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			 uint8_t *dst, uint64_t *indices, uint64_t *values, size_t *nexc);
/* bitpfr.h:101-124 */
size_t orc_bitpfr_decode(int dtype, const uint8_t *src, size_t n, uint64_t origin, unsigned nbits,
			 const uint64_t *indices, const uint64_t *values, size_t nexc, void *dst);

/* varint.h:133-147 with the default value codec (zigzag for signed T). */
size_t orc_varint_encode(int dtype, const void *src, size_t n, uint8_t *dst);
size_t orc_varint_decode(int dtype, const uint8_t *src, size_t n, void *dst);

/* integer_stats.h:48-53, 74-80: n, min, max, hist[W+1] of used(v - min). */
void orc_stats(int dtype, const void *src, size_t n, uint64_t *min_raw, uint64_t *max_raw,
	       uint64_t *hist /* 65 entries */);

/* integer_codec.h:234-384 on a fresh metadata object. */
void orc_select(int dtype, const void *src, size_t n, orc_meta *meta);
/* integer_codec.h:196-204; returns bytes written (== meta->metaspace). */
size_t orc_meta_encode(int dtype, const orc_meta *meta, uint8_t *dst);
/* integer_codec.h:206-214; returns bytes consumed. dataspace/metaspace are not recoverable. */
size_t orc_meta_decode(int dtype, const uint8_t *src, orc_meta *meta);
/* integer_codec.h:386-393; returns bytes written (== meta->dataspace). */
size_t orc_codec_encode(int dtype, const void *src, size_t n, const orc_meta *meta, uint8_t *dst);
/* integer_codec.h:395-402; returns bytes consumed. */
size_t orc_codec_decode(int dtype, const uint8_t *src, size_t n, const orc_meta *meta, void *dst);

/*
 * integer_group.h:44-60 with aligned = true: [meta][zero pad to 8][data].
 * The reference leaves the pad bytes uninitialised; here they are zero.
 * Returns the total byte count; dst needs orc_group_bound() bytes.
 */
size_t orc_group_encode(int dtype, const void *src, size_t n, uint8_t *dst);
/* integer_group.h:62-76; returns bytes consumed. */
size_t orc_group_decode(int dtype, const uint8_t *src, size_t n, void *dst);
/* Upper bound of orc_group_encode output for n values of this dtype. */
size_t orc_group_bound(int dtype, size_t n);

/*
 * Batch form used by the tests and the CPU baseline: the array is cut into
 * blocks of `block` values (the last one may be short), each block is encoded
 * as one integer_group image, images are concatenated with every image start
 * rounded up to `align` bytes (zero filled).  offsets[] receives nblocks + 1
 * byte offsets.  Returns the total byte count.
 */
size_t orc_blocks_encode(int dtype, const void *src, size_t n, size_t block, size_t align,
			 uint8_t *dst, uint64_t *offsets);
void orc_blocks_decode(int dtype, const uint8_t *src, const uint64_t *offsets, size_t n,
		       size_t block, void *dst);

#ifdef __cplusplus
}
#endif

#endif /* OROCH_ORACLE_H_ */
