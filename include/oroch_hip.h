/*
 * oroch_hip.h -- C ABI of the MI355X (gfx950) implementation of Oroch's block
 * integer-compression path.
 *
 * Plain pointers and sizes only.  Unless a parameter says "host", every data
 * pointer is a DEVICE pointer on the current HIP device and every call is
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the default
 * stream).  Functions return 0 on success, a positive hipError_t value when
 * the runtime fails, or a negative OROCH_E_* code for argument errors.
 *
 * The reference is a header-only C++ template library without an FFI; each
 * entry point below names the template member(s) of the reference it stands
 * in for (paths are below /root/reference/oroch).  `include/oroch/` holds the
 * C++ host templates that keep the reference's names and call into this ABI.
 *
 * Element types are selected by a dtype code:
 *     dtype = 2 * log2(sizeof T) + (unsigned ? 1 : 0)
 *     0 int8  1 uint8  2 int16  3 uint16  4 int32  5 uint32  6 int64  7 uint64
 * Origins are passed as the raw bit pattern of T zero-extended to 64 bits.
 */
#ifndef OROCH_HIP_H_
#define OROCH_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
	OROCH_I8 = 0, OROCH_U8 = 1, OROCH_I16 = 2, OROCH_U16 = 3,
	OROCH_I32 = 4, OROCH_U32 = 5, OROCH_I64 = 6, OROCH_U64 = 7,
};

/* encoding_t, integer_codec.h:46-54 */
enum {
	OROCH_NAUGHT = 0, OROCH_NORMAL = 1, OROCH_VARINT = 2, OROCH_VARFOR = 3,
	OROCH_BITPCK = 4, OROCH_BITFOR = 5, OROCH_BITPFR = 6,
};

enum {
	OROCH_E_DTYPE = -1,     /* unknown dtype code */
	OROCH_E_NBITS = -2,     /* nbits outside 1..bits(T) */
	OROCH_E_BLOCK = -3,     /* unsupported values-per-block */
	OROCH_E_CAPACITY = -4,  /* destination / workspace too small */
	OROCH_E_ARGUMENT = -5,  /* null pointer or inconsistent sizes */
	OROCH_E_NOGPU = -6,     /* no usable HIP device */
	OROCH_E_RANGE = -7,     /* position outside the array */
	OROCH_E_TOO_LARGE = -8, /* beyond the fixed maximum of a non-allocating entry point */
};

/* Values per block of the batch (group) entry points; integer_array.h:44. */
#define OROCH_BLOCK_VALUES 256

/*
 * Result of codec selection for one block: the fields of
 * encoding_metadata<T> (integer_codec.h:95-126) plus the number of bytes the
 * block occupies in a batch stream.
 */
typedef struct oroch_block_desc {
	uint64_t origin;         /* value_desc.origin, raw bits */
	uint32_t dataspace;      /* value_desc.dataspace */
	uint32_t size;           /* align16(align8(metaspace) + dataspace) */
	uint16_t metaspace;      /* metaspace() = 1 + value_desc.metaspace */
	uint16_t noutliers;
	uint16_t nvalues;        /* values in this block */
	uint8_t encoding;        /* value_desc.encoding */
	uint8_t nbits;           /* value_desc.nbits */
	uint8_t index_encoding;  /* outlier_index_desc.encoding (bitpfr only) */
	uint8_t index_nbits;
	uint8_t value_encoding;  /* outlier_value_desc.encoding (bitpfr only) */
	uint8_t value_nbits;
	uint8_t reserved[4];
} oroch_block_desc;

/*
 * Statistics of one block: integer_stats<T> (integer_stats.h:35-113) after its
 * constructor and build_histogram(): count, minimum, maximum and the W + 1 bin
 * histogram of the bit length of (v - min).
 */
typedef struct oroch_block_stats {
	uint64_t min;            /* raw bits; for an empty block max() of T */
	uint64_t max;            /* raw bits; for an empty block min() of T */
	uint16_t nvalues;
	uint16_t histogram[65];  /* bins W+1 .. 64 stay zero */
	uint16_t reserved[6];
} oroch_block_stats;

/* Library / device set-up.  Selects `device`, verifies it is usable. */
int oroch_hip_init(int device);
/* Human-readable text for a return code of this library. */
const char *oroch_hip_strerror(int code);
/* The compiler and HIP version this library was built with (no counterpart in
 * the reference; benchmarks record it beside their figures). */
const char *oroch_hip_build_info(void);

/* ---- sizes (host arithmetic) --------------------------------------------- */

/* bitpck_codec::capacity / space, bitpck.h:54-71.  space(0, b) = 0. */
size_t oroch_bitpck_capacity(unsigned nbits);
size_t oroch_bitpck_space(size_t nvalues, unsigned nbits);

/* ---- whole-sequence codecs with caller-supplied parameters ---------------- */

/*
 * bitpck_codec<T>::encode / decode, bitpck.h:227-256, default value codec
 * (zigzag for signed T, zigzag.h:85-93).  dst of encode needs
 * oroch_bitpck_space(n, nbits) bytes, 16-byte aligned.
 */
int oroch_bitpck_encode(int dtype, const void *src, size_t n, unsigned nbits, void *dst,
			void *stream);
int oroch_bitpck_decode(int dtype, const void *src, size_t n, unsigned nbits, void *dst,
			void *stream);

/*
 * bitfor_codec<T>::encode / decode, bitfor.h:59-69: bitpck with
 * origin_codec (origin.h:46-54), no zigzag.
 */
int oroch_bitfor_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			void *dst, void *stream);
int oroch_bitfor_decode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			void *dst, void *stream);

/*
 * bitpck_codec<T,V>::fetch, bitpck.h:203-225, 258-266, and bitfor_codec<T>::fetch,
 * bitfor.h:71-74, for a batch of positions: dst[i] = value number index[i] of
 * the packed sequence of n values at src (8-byte aligned).  vcodec: 0 = zigzag
 * (bitpck), 1 = origin (bitfor).  A position >= n reads as a packed zero.
 */
int oroch_bitpck_fetch(int dtype, int vcodec, const void *src, size_t n, uint64_t origin,
		       unsigned nbits, const uint64_t *index, size_t m, void *dst, void *stream);

/*
 * bitpfr_codec<T>::encode, bitpfr.h:95-99 with parameters::value_encode
 * (bitpfr.h:75-85): packs the low nbits of (v - origin) and appends, in
 * sequence order, one entry per value that does not fit: the gap-coded
 * position (offset_codec<size_t,1,false>, offset.h:55-60) into indices[] and
 * (v - origin) >> nbits into values[] (both uint64_t on the device, room for
 * `exc_capacity` entries each).  *nexc (device) receives the number of
 * exceptions found even when it exceeds exc_capacity (nothing is written past
 * the capacity).  workspace: oroch_bitpfr_workspace(n) bytes of device memory
 * (about 0.35 bytes per value: per tile of the packer the count and last
 * position of its exceptions and room for one noted exception per 32 values).
 */
size_t oroch_bitpfr_workspace(size_t n);
int oroch_bitpfr_encode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			void *dst, uint64_t *indices, uint64_t *values, size_t exc_capacity,
			uint64_t *nexc, void *workspace, size_t workspace_bytes, void *stream);
/*
 * bitpfr_codec<T>::decode, bitpfr.h:101-124: unpack (decode_basic) then patch
 * (decode_patch; running sum of gaps, offset.h:62-67).
 */
int oroch_bitpfr_decode(int dtype, const void *src, size_t n, uint64_t origin, unsigned nbits,
			const uint64_t *indices, const uint64_t *values, size_t nexc, void *dst,
			void *workspace, size_t workspace_bytes, void *stream);

/*
 * bitpfr_codec<T>::decode_patch alone, bitpfr.h:114-124, on n already unpacked
 * values in dst.
 */
int oroch_bitpfr_patch(int dtype, void *dst, size_t n, uint64_t origin, unsigned nbits,
		       const uint64_t *indices, const uint64_t *values, size_t nexc,
		       void *workspace, size_t workspace_bytes, void *stream);

/*
 * varint_codec<T>::space / encode / decode over a whole sequence, varint.h
 * :124-147, default value codec (zigzag for signed T).  workspace:
 * oroch_varint_workspace(units) bytes with units = n for space / encode and
 * units = nbytes for decode.  encode needs room for the worst case,
 * n * ceil(bits(T) / 7) bytes; space and (when total_host is not NULL) encode
 * wait for the stream and return the byte count.  decode reads n values from the
 * nbytes at src (16-byte aligned); bytes that do not form n complete values
 * leave the remaining outputs untouched.
 */
size_t oroch_varint_workspace(size_t units);
int oroch_varint_space(int dtype, const void *src, size_t n, void *workspace, size_t workspace_bytes,
		       uint64_t *space_host, void *stream);
int oroch_varint_encode(int dtype, const void *src, size_t n, void *dst, size_t dst_capacity,
			void *workspace, size_t workspace_bytes, uint64_t *total_host, void *stream);
int oroch_varint_decode(int dtype, const void *src, size_t nbytes, size_t n, void *dst,
			void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same three for values and bytes in host memory (varint.h:124-147 as the
 * README's example calls them): staged to the device and back; they allocate
 * and wait.  encode writes at most dst_capacity bytes (OROCH_E_CAPACITY if the
 * code is longer; the caller sized dst from space); decode takes the nbytes at
 * src_host, which must hold n complete values (OROCH_E_ARGUMENT otherwise),
 * and reports how many of them the n values occupy.
 */
int oroch_varint_space_host(int dtype, const void *src_host, size_t n, uint64_t *space_host);
int oroch_varint_encode_host(int dtype, const void *src_host, size_t n, void *dst_host,
			     size_t dst_capacity, uint64_t *total_host);
int oroch_varint_decode_host(int dtype, const void *src_host, size_t nbytes, size_t n, void *dst_host,
			     uint64_t *consumed_host);

/* ---- per-block statistics, selection, batch encode / decode --------------- */

/*
 * integer_stats<T>, integer_stats.h:48-53 (count, min, max) and :74-80 (bit
 * length histogram), for every block of OROCH_BLOCK_VALUES values.  stats:
 * device array of ceil(n / block) records.
 */
int oroch_group_stats(int dtype, const void *src, size_t n, oroch_block_stats *stats,
		      void *stream);

/*
 * integer_codec<T>::select, integer_codec.h:234-384 (with integer_stats.h
 * :48-53, 74-80), applied independently to every block of
 * OROCH_BLOCK_VALUES values of src[0..n) (the last block may be short), each
 * on a fresh metadata object.  desc: device array of ceil(n / block) records.
 */
int oroch_group_select(int dtype, const void *src, size_t n, oroch_block_desc *desc,
		       void *stream);

/*
 * Arrays of values (src of select / stats / encode, dst of decode) are device
 * memory aligned to 4 bytes, 8 for 64-bit elements: any allocation, and any slice
 * of one that starts at a multiple of four elements.  OROCH_E_ARGUMENT otherwise.
 */

/* Worst-case bytes of the batch stream for n values; workspace bytes needed. */
size_t oroch_group_bound(int dtype, size_t n);
size_t oroch_group_workspace(int dtype, size_t n);

/*
 * integer_group<T>::encode, integer_group.h:44-60 (aligned = true), for every
 * block: stream = concatenation of [metadata][zero pad to 8][data] images,
 * every image starting at a multiple of 16 bytes (zero filled).  offsets:
 * device array of ceil(n / block) + 1 byte offsets (the last is the stream
 * length).  dst needs oroch_group_bound() bytes, 16-byte aligned.  If
 * total_host is not NULL the call waits for the stream and stores the stream
 * length there.
 */
int oroch_group_encode(int dtype, const void *src, size_t n, void *dst, size_t dst_capacity,
		       uint64_t *offsets, void *workspace, size_t workspace_bytes,
		       uint64_t *total_host, void *stream);

/*
 * oroch_group_encode without a workspace argument and without any allocation,
 * for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * The workspace is a fixed 4 MiB device array inside the library (one copy per
 * GPU, held until the process ends), which limits a call to
 * oroch_group_encode_static_max(dtype) = 2^24 values; more gives
 * OROCH_E_TOO_LARGE.  Same stream bytes and offsets as oroch_group_encode; the
 * stream length is offsets[ceil(n / block)] on the device.  All calls on one GPU
 * share that GPU's array: calls that overlap in time on one GPU (different
 * streams or threads) race and must be serialised by the caller.  Kernel-time
 * recording (oroch_hip_profile_start) skips this entry point.
 */
size_t oroch_group_encode_static_max(int dtype);
int oroch_group_encode_static(int dtype, const void *src, size_t n, void *dst, size_t dst_capacity,
			      uint64_t *offsets, void *stream);

/*
 * integer_group<T>::decode, integer_group.h:62-76, for every block of a
 * stream produced by oroch_group_encode (or by the reference's integer_group
 * laid out the same way).  src must be readable up to the stream length,
 * offsets[ceil(n / block)], rounded up to 16 bytes.  The other offsets and the
 * stream bytes are not trusted: damage there gives wrong values, not reads
 * outside the stream or writes outside dst[0..n).
 */
int oroch_group_decode(int dtype, const void *src, const uint64_t *offsets, size_t n, void *dst,
		       void *stream);

/* ---- host-buffer conveniences (stage through pinned memory, synchronous) -- */

/*
 * Same as oroch_group_encode / oroch_group_decode with HOST pointers: the
 * array goes through a pipe (see oroch_pipe_create) that the library makes on
 * first use per device and keeps, in chunks, with copies in, kernels and
 * copies out overlapped.  dst_capacity >= oroch_group_bound(); offsets_host holds
 * ceil(n / block) + 1 entries.  Returns the stream length in *total_host.
 *
 * The pipe's chunks are sized for the request: 1 MiB of values at least (about
 * 5 MiB of device and as much pinned host memory), 64 MiB at most (about 290
 * and 270 MiB); a later, larger request replaces the pipe by a larger one.  It
 * is kept until oroch_hip_release_host_buffers(), and the *_host calls of one
 * device run one at a time.  Callers that want to own this memory, or several
 * pipes side by side, use oroch_pipe_create.
 */
int oroch_group_encode_host(int dtype, const void *src_host, size_t n, void *dst_host,
			    size_t dst_capacity, uint64_t *offsets_host, uint64_t *total_host);
int oroch_group_decode_host(int dtype, const void *src_host, const uint64_t *offsets_host,
			    size_t n, void *dst_host);
/* Frees the pipes behind the two calls above, on every device (they come back on the next call). */
void oroch_hip_release_host_buffers(void);
/* oroch_group_select / oroch_group_stats with HOST pointers. */
int oroch_group_select_host(int dtype, const void *src_host, size_t n,
			    oroch_block_desc *desc_host);
int oroch_group_stats_host(int dtype, const void *src_host, size_t n,
			   oroch_block_stats *stats_host);
/*
 * The whole-sequence codecs with HOST pointers.  vcodec: 0 = bitpck (zigzag
 * for signed T), 1 = bitfor (origin).  The bitpfr calls exchange the
 * exception lists as host arrays of uint64_t; *nexc_host receives the count
 * (encode needs room for n entries in each list).
 */
int oroch_pack_host(int dtype, int vcodec, const void *src_host, size_t n, uint64_t origin,
		    unsigned nbits, void *dst_host);
int oroch_unpack_host(int dtype, int vcodec, const void *src_host, size_t n, uint64_t origin,
		      unsigned nbits, void *dst_host);
/*
 * oroch_bitpck_fetch with HOST pointers (bitpck.h:203-225, 258-266; bitfor.h
 * :71-74): dst_host[j] = value number index_host[j] of the packed sequence at
 * src_host.  Only the 16-byte words that hold the wanted values are sent to
 * the device; the caller vouches that every position lies inside the sequence.
 */
int oroch_bitpck_fetch_host(int dtype, int vcodec, const void *src_host, uint64_t origin,
			    unsigned nbits, const uint64_t *index_host, size_t m, void *dst_host);
int oroch_bitpfr_encode_host(int dtype, const void *src_host, size_t n, uint64_t origin,
			     unsigned nbits, void *dst_host, uint64_t *indices_host,
			     uint64_t *values_host, uint64_t *nexc_host);
int oroch_bitpfr_decode_host(int dtype, const void *src_host, size_t n, uint64_t origin,
			     unsigned nbits, const uint64_t *indices_host,
			     const uint64_t *values_host, size_t nexc, void *dst_host);
int oroch_bitpfr_patch_host(int dtype, void *dst_host, size_t n, uint64_t origin, unsigned nbits,
			    const uint64_t *indices_host, const uint64_t *values_host,
			    size_t nexc);

/* ---- host buffers through an overlapped pipeline -------------------------------- */

/*
 * A pipe owns two sets of device buffers, pinned staging buffers and three
 * streams, all allocated by oroch_pipe_create and kept until
 * oroch_pipe_destroy; encode and decode through it allocate nothing.  The
 * array is cut into chunks of chunk_bytes of values (0: 64 MiB; rounded up to
 * whole blocks): while chunk k is encoded, chunk k + 1 travels to the device
 * and the stream of chunk k - 1 travels back, so the two directions of the
 * link and the kernels work at the same time (blocks are independent,
 * integer_group.h:44-76; a chunk's offsets are rebased by the length of the
 * stream before it).  Results are those of oroch_group_encode_host /
 * oroch_group_decode_host.  Caller buffers that are pinned (hipHostMalloc,
 * hipHostRegister) go over the link as they are, others through the pipe's
 * staging buffers.  One call at a time per pipe (calls are serialised inside);
 * a pipe belongs to the device that was current when it was made, and a call
 * with another device current returns OROCH_E_ARGUMENT.
 */
typedef struct oroch_pipe oroch_pipe;
int oroch_pipe_create(size_t chunk_bytes, oroch_pipe **out);
void oroch_pipe_destroy(oroch_pipe *pipe);
int oroch_pipe_encode(oroch_pipe *pipe, int dtype, const void *src_host, size_t n, void *dst_host,
		      size_t dst_capacity, uint64_t *offsets_host, uint64_t *total_host);
int oroch_pipe_decode(oroch_pipe *pipe, int dtype, const void *src_host, const uint64_t *offsets_host,
		      size_t n, void *dst_host);

/* ---- one group of any length ---------------------------------------------------- */

/*
 * Codec selection for ONE sequence of any length, the way the reference's
 * integer_codec<T>::select treats the range it is given (integer_codec.h
 * :234-384): statistics over the whole sequence, pck(n, b) for the packed part,
 * one walk over all gaps per candidate width.  The fields are those of
 * encoding_metadata<T> (integer_codec.h:95-126) at full width.
 */
typedef struct oroch_sequence_desc {
	uint64_t origin;         /* value_desc.origin, raw bits */
	uint64_t dataspace;      /* value_desc.dataspace */
	uint64_t noutliers;
	uint64_t nvalues;
	uint32_t metaspace;      /* metaspace() = 1 + value_desc.metaspace */
	uint8_t encoding;        /* value_desc.encoding */
	uint8_t nbits;           /* value_desc.nbits */
	uint8_t index_encoding;  /* outlier_index_desc.encoding (bitpfr only) */
	uint8_t index_nbits;
	uint8_t value_encoding;  /* outlier_value_desc.encoding (bitpfr only) */
	uint8_t value_nbits;
	uint8_t reserved[6];
} oroch_sequence_desc;

/*
 * integer_codec<T>::select, integer_codec.h:234-384, on src[0..n) (device) as
 * one sequence; *desc_host receives the result on a fresh metadata object.
 * Waits for the stream; allocates its working memory itself.
 */
int oroch_sequence_select(int dtype, const void *src, size_t n, oroch_sequence_desc *desc_host,
			  void *stream);
/*
 * integer_codec<T>::encode, integer_codec.h:386-393 (encode_basic :498-528,
 * bitpfr :562-581): the data part of src[0..n) coded the way *desc_host says,
 * whoever made the descriptor -- encoding, origin, nbits and, for bitpfr, the
 * encodings and widths of the two outlier lists are taken from it as they are
 * (values that do not fit a width the caller chose lose their high bits, as in
 * the reference).  For bitpfr desc_host->noutliers has to be the number of
 * values that do not fit nbits (OROCH_E_ARGUMENT otherwise).  dst: device
 * memory, 16-byte aligned, dst_capacity bytes (OROCH_E_CAPACITY when the data
 * does not fit; desc_host->dataspace bytes are enough for a descriptor that
 * select made).  *bytes_host receives the length.  Waits for the stream.
 */
int oroch_sequence_encode(int dtype, const void *src, size_t n, const oroch_sequence_desc *desc_host,
			  void *dst, size_t dst_capacity, uint64_t *bytes_host, void *stream);
/*
 * integer_codec<T>::decode, integer_codec.h:395-402 (decode_basic :530-560,
 * bitpfr :583-609): n values from the data part at src (device, 16-byte
 * aligned, src_bytes readable bytes) into dst (device).  *consumed_host (may be
 * NULL) receives the bytes the data part occupies.  Waits for the stream.
 */
int oroch_sequence_decode(int dtype, const void *src, size_t src_bytes, size_t n,
			  const oroch_sequence_desc *desc_host, void *dst, uint64_t *consumed_host,
			  void *stream);
/* The three with HOST pointers for values and data. */
int oroch_sequence_select_host(int dtype, const void *src_host, size_t n, oroch_sequence_desc *desc_host);
int oroch_sequence_encode_host(int dtype, const void *src_host, size_t n,
			       const oroch_sequence_desc *desc_host, void *dst_host, size_t dst_capacity,
			       uint64_t *bytes_host);
int oroch_sequence_decode_host(int dtype, const void *src_host, size_t src_bytes, size_t n,
			       const oroch_sequence_desc *desc_host, void *dst_host, uint64_t *consumed_host);
/*
 * encoding_metadata<T>::encode / decode, integer_codec.h:128-214 (host
 * arithmetic): the metadata bytes of a descriptor (at most 32) and back.
 * Return the number of bytes written / read, or a negative error
 * (OROCH_E_CAPACITY when fewer than the metadata needs are available).
 */
int oroch_sequence_meta_encode(int dtype, const oroch_sequence_desc *desc_host, void *dst_host,
			       size_t dst_capacity);
int oroch_sequence_meta_decode(int dtype, const void *src_host, size_t src_bytes,
			       oroch_sequence_desc *desc_host);
/* Bytes that always hold the data part of n values whatever select chooses. */
size_t oroch_sequence_bound(int dtype, size_t n);

/* ---- serialised container ------------------------------------------------------ */

/*
 * One self-contained byte string for a compressed array, for files and
 * network buffers.  The reference has no counterpart: integer_array keeps one
 * heap buffer per group (integer_array.h:258) and defines no serialised form.
 *
 *   [header, 64 bytes]
 *   [offsets table: (nblocks + 1) x uint64, relative to the first block image]
 *   [zero pad to a multiple of 16]
 *   [block images as oroch_group_encode writes them]
 *
 * Every block image is byte-identical to integer_group<T>::encode(aligned =
 * true) of its 256 values and starts at a multiple of 16 bytes, so any range of
 * blocks can be decoded on its own.
 */
typedef struct oroch_container_header {
	uint8_t magic[8];        /* "OROCHBK1" */
	uint32_t dtype;          /* dtype code */
	uint32_t block_values;   /* OROCH_BLOCK_VALUES */
	uint64_t nvalues;
	uint64_t nblocks;
	uint64_t table_offset;   /* 64 */
	uint64_t stream_offset;  /* align16(64 + 8 * (nblocks + 1)) */
	uint64_t stream_bytes;   /* length of the block images */
	uint64_t reserved;       /* 0 */
} oroch_container_header;

/* Worst-case container bytes for n values. */
size_t oroch_container_bound(int dtype, size_t n);
/*
 * Encodes src[0..n) (device) into a container at dst (device, 16-byte aligned,
 * oroch_container_bound() bytes).  workspace: oroch_group_workspace() bytes.
 * Waits for the stream; *total_host receives the container length.
 */
int oroch_container_encode(int dtype, const void *src, size_t n, void *dst, size_t dst_capacity,
			   void *workspace, size_t workspace_bytes, uint64_t *total_host,
			   void *stream);
/*
 * Checks a header (host copy of the container's first 64 bytes) against the
 * number of bytes at hand; OROCH_E_ARGUMENT if it is not a container this
 * library wrote or the bytes do not cover it.
 */
int oroch_container_check(const oroch_container_header *header_host, size_t container_bytes);
/*
 * Decodes blocks [first_block, first_block + nblocks) of the container at src
 * (device, 16-byte aligned) into dst (device): values first_block * 256 ...
 * header_host: host copy of the header, as checked by oroch_container_check.
 */
int oroch_container_decode(const void *src, const oroch_container_header *header_host,
			   size_t first_block, size_t nblocks, void *dst, void *stream);
/* The same with host buffers on both sides (staged through pinned memory). */
int oroch_container_encode_host(int dtype, const void *src_host, size_t n, void *dst_host,
				size_t dst_capacity, uint64_t *total_host);
int oroch_container_decode_host(const void *src_host, size_t container_bytes, size_t first_block,
				size_t nblocks, void *dst_host);

/* ---- searching and random access over block images ------------------------------ */

/*
 * For each of the m values (device, raw bits zero-extended to 64) the lowest
 * position in the n-value array encoded at src / offsets (as written by
 * oroch_group_encode) that holds it, or OROCH_NOT_FOUND: integer_array<T>::find,
 * integer_array.h:192-208, over all groups at once.  With m == 1 the blocks whose
 * metadata rules the value out are not decoded (the width shortcut of
 * integer_array.h:110-122).  index: m x uint64 (device).
 */
#define OROCH_NOT_FOUND (~(uint64_t) 0)
int oroch_group_find(int dtype, const void *src, const uint64_t *offsets, size_t n,
		     const uint64_t *values, size_t m, uint64_t *index, void *stream);
/*
 * dst[q] = value number index[q] of the array, for m positions (device):
 * array_integer_group<T>::operator[], integer_array.h:64-69 (decode the group,
 * pick one value), for a batch.  A position >= n yields 0.
 */
int oroch_group_at(int dtype, const void *src, const uint64_t *offsets, size_t n,
		   const uint64_t *index, size_t m, void *dst, void *stream);

/* ---- growable compressed array -------------------------------------------------- */

/*
 * integer_array<T>, integer_array.h:150-259: full 256-value groups are kept
 * encoded (here: one stream of block images in device memory plus its offsets
 * table; each image byte-identical to the group's integer_group<T>::encode),
 * the last 0..255 values raw on the host.  All calls run on the device that is
 * current, use the null stream and return when their work is done; values cross
 * the ABI as raw bits zero-extended to 64.  A handle is not thread-safe.
 */
typedef struct oroch_array oroch_array;
int oroch_array_create(int dtype, oroch_array **out);
void oroch_array_destroy(oroch_array *array);
/* integer_array.h:161-164 */
uint64_t oroch_array_size(const oroch_array *array);
/* integer_array.h:210-214; keeps the device memory for reuse */
int oroch_array_clear(oroch_array *array);
/*
 * Bulk append of n values from device / host memory: every group that fills is
 * encoded in one device pass.  (The reference grows by insert() alone.)
 */
int oroch_array_append(oroch_array *array, const void *src, size_t n);
int oroch_array_append_host(oroch_array *array, const void *src_host, size_t n);
/*
 * integer_array.h:216-245.  index <= size, else OROCH_E_RANGE.  Below the tail
 * every later group is decoded, shifted and encoded again, on the device.
 */
int oroch_array_insert(oroch_array *array, uint64_t index, uint64_t value);
/*
 * integer_array.h:192-208 for m values at once (host arrays): index_host[j] =
 * first position of values_host[j] or OROCH_NOT_FOUND; groups first, then the tail.
 */
int oroch_array_find(oroch_array *array, const uint64_t *values_host, size_t m,
		     uint64_t *index_host);
/*
 * integer_array.h:166-178 for m positions at once (host arrays); dst_host gets m
 * elements of the array's type.  OROCH_E_RANGE if a position is >= size (the
 * reference throws std::out_of_range).
 */
int oroch_array_at(oroch_array *array, const uint64_t *index_host, size_t m, void *dst_host);
/* Values [first, first + count) into dst_host; OROCH_E_RANGE beyond size. */
int oroch_array_read(oroch_array *array, uint64_t first, uint64_t count, void *dst_host);
/*
 * The array as a serialised container (above) in host memory and back: the
 * full groups are the container's blocks as they are, the tail its short last
 * block.  oroch_array_load replaces the content of an array of the same dtype
 * (OROCH_E_DTYPE otherwise; OROCH_E_ARGUMENT for bytes that are no container).
 */
size_t oroch_array_save_bound(const oroch_array *array);
int oroch_array_save(const oroch_array *array, void *dst_host, size_t capacity, uint64_t *total_host);
int oroch_array_load(oroch_array *array, const void *src_host, size_t bytes);
/*
 * Copies the image of one group (what integer_group<T>::encode(aligned = true)
 * writes for its 256 values, zero padded to a multiple of 16) to dst_host;
 * *bytes receives its length, also when capacity is too small
 * (OROCH_E_CAPACITY).  OROCH_E_RANGE if there is no such group.
 */
int oroch_array_image(const oroch_array *array, uint64_t group, void *dst_host, size_t capacity,
		      uint64_t *bytes);
/*
 * The encoded groups as oroch_group_decode / oroch_group_find take them
 * (device pointers, valid until the next modifying call) and the raw tail
 * (host).  Any output pointer may be null.
 */
int oroch_array_view(const oroch_array *array, const void **stream, const uint64_t **offsets,
		     uint64_t *nblocks, uint64_t *stream_bytes, const void **tail_host,
		     uint64_t *ntail);

/* ---- diagnostics ---------------------------------------------------------------- */

/*
 * Per-kernel device time, measured with HIP events recorded on the stream each
 * kernel is launched on.  No counterpart in the reference (its benches use
 * std::chrono around whole calls, tests/bench/group-bench.cc:27-34); used by
 * bench.py to price the dominant kernel against the HBM roofline.
 * oroch_hip_profile_start arms recording for at most max_records launches;
 * oroch_hip_profile_stop waits for the recorded events, adds up milliseconds
 * and launch counts per kernel id (arrays of OROCH_K_COUNT) and disarms.
 */
enum {
	OROCH_K_SELECT = 0,  /* statistics + selection per block */
	OROCH_K_SCAN = 1,    /* the three scan launches together */
	OROCH_K_ENCODE = 2,  /* block images -> stream */
	OROCH_K_DECODE = 3,  /* stream -> values */
	OROCH_K_PACK = 4,    /* whole-sequence pack */
	OROCH_K_UNPACK = 5,  /* whole-sequence unpack */
	OROCH_K_EXCEPT = 6,  /* whole-sequence exception list passes */
	OROCH_K_COUNT = 7,
};
int oroch_hip_profile_start(int max_records);
int oroch_hip_profile_stop(double *ms, unsigned *launches);

/*
 * Run-time switches between kernels that produce the same bytes, for tests and
 * experiments (no counterpart in the reference).  Each starts from the
 * environment variable OROCH_HIP_<NAME>, read once per process; set_option
 * returns the previous value, or OROCH_E_ARGUMENT for an unknown switch.
 */
enum {
	OROCH_OPT_ENCODE_PATH = 0,       /* kept for callers that set it: the encoder has one path
					    (select, scan, encode as separate launches) */
	OROCH_OPT_SELECT_ONE_BLOCK = 1,  /* 1: one block per wavefront for every element type */
	OROCH_OPT_ENCODE_ONE_BLOCK = 2,
	OROCH_OPT_DECODE_ONE_BLOCK = 3,
	OROCH_OPT_FIND_ONE_BLOCK = 4,
	OROCH_OPT_COUNT = 5,
};
int oroch_hip_set_option(int option, int value);
int oroch_hip_get_option(int option);

#ifdef __cplusplus
}
#endif

#endif /* OROCH_HIP_H_ */
