// integer_codec.h -- metadata, codec selection and the encode / decode
// dispatch of one block, computed on the GPU.
// Stands in for /root/reference/oroch/integer_codec.h:46-610.
//
// encoding_t, the descriptors and the metadata byte format are host code: a
// few bytes per group.  select / encode / decode hand the range they are given
// to the HIP library as ONE group, whatever its length: up to
// OROCH_BLOCK_VALUES values go through the block kernels (one wavefront),
// longer ranges through the whole-sequence kernels (oroch_sequence_*).  encode
// codes the values the way the metadata it is handed says, whoever made it.
// Throughput comes from the batch entry points (oroch_group_encode and
// friends, or integer_blocks<T> in integer_group.h) -- these per-group calls pay
// a PCIe round trip each and exist so that code written against the reference
// keeps working unchanged.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_INTEGER_CODEC_H_
#define OROCH_INTEGER_CODEC_H_

#include <algorithm>
#include <cstring>
#include <iterator>
#include <ostream>

#include "bitfor.h"
#include "bitpck.h"
#include "bitpfr.h"
#include "common.h"
#include "integer_stats.h"
#include "naught.h"
#include "normal.h"
#include "offset.h"
#include "origin.h"
#include "varint.h"
#include "zigzag.h"

namespace oroch {

enum encoding_t : byte_t {
	naught = 0,
	normal = 1,
	varint = 2,
	varfor = 3,
	bitpck = 4,
	bitfor = 5,
	bitpfr = 6,
};

namespace detail {

template <typename T>
struct encoding_descriptor
{
	using original_t = T;

	encoding_t encoding;
	size_t dataspace; // bytes of data
	size_t metaspace; // bytes of metadata apart from the encoding byte
	original_t origin;
	size_t nbits;

	encoding_descriptor()
	{
		clear();
	}
	void clear()
	{
		encoding = encoding_t::normal;
		dataspace = 0;
		metaspace = 0;
		origin = 0;
		nbits = 0;
	}
};

} // namespace detail

// integer_codec.h:95-215.
template <typename T>
struct encoding_metadata
{
	using original_t = T;
	using unsigned_t = typename integer_traits<original_t>::unsigned_t;

	detail::encoding_descriptor<original_t> value_desc;

	size_t noutliers = 0;
	detail::encoding_descriptor<size_t> outlier_index_desc;
	detail::encoding_descriptor<unsigned_t> outlier_value_desc;

	size_t dataspace() const
	{
		return value_desc.dataspace;
	}
	size_t metaspace() const
	{
		return 1 + value_desc.metaspace;
	}
	void clear()
	{
		value_desc.clear();
		noutliers = 0;
		outlier_index_desc.clear();
		outlier_value_desc.clear();
	}

	void encode(dst_bytes_t &dst) const
	{
		const encoding_t e = value_desc.encoding;
		*dst++ = e;
		if (e == naught || e == varfor || e == bitfor || e == bitpfr)
			varint_codec<original_t>::value_encode(dst, value_desc.origin);
		if (e == bitpck || e == bitfor || e == bitpfr)
			*dst++ = byte_t(value_desc.nbits);
		if (e == bitpfr) {
			varint_codec<size_t>::value_encode(dst, noutliers);
			*dst++ = outlier_index_desc.encoding == bitpck ? byte_t(outlier_index_desc.nbits) : 0;
			*dst++ = outlier_value_desc.encoding == bitpck ? byte_t(outlier_value_desc.nbits) : 0;
		}
	}

	void decode(src_bytes_t &src)
	{
		const encoding_t e = static_cast<encoding_t>(*src++);
		value_desc.encoding = e;
		if (e == naught || e == varfor || e == bitfor || e == bitpfr)
			varint_codec<original_t>::value_decode(value_desc.origin, src);
		if (e == bitpck || e == bitfor || e == bitpfr)
			value_desc.nbits = *src++;
		if (e == bitpfr) {
			noutliers = varint_codec<size_t>::value_decode(src);
			extra(src, outlier_index_desc);
			extra(src, outlier_value_desc);
		}
	}

private:
	template <typename I>
	static void extra(src_bytes_t &src, detail::encoding_descriptor<I> &desc)
	{
		const byte_t nb = *src++;
		if (nb == 0) {
			desc.encoding = varint;
		} else {
			desc.encoding = bitpck;
			desc.nbits = nb;
		}
	}
};

template <typename CharT, typename Traits, typename T>
std::basic_ostream<CharT, Traits> &
operator<<(std::basic_ostream<CharT, Traits> &os, const encoding_metadata<T> &meta)
{
	os << "encoding: " << static_cast<int>(meta.value_desc.encoding)
	   << ", origin: " << meta.value_desc.origin << ", nbits: " << meta.value_desc.nbits;
	return os;
}

namespace detail {

// Bytes a list of n values occupies after src when coded as desc says.
template <typename I>
inline size_t list_extent(const encoding_descriptor<I> &desc, size_t n, src_bytes_t src)
{
	if (desc.encoding == bitpck)
		return bitpck_codec<I>::space(n, desc.nbits);
	size_t bytes = 0;
	for (size_t k = 0; k < n; k++)
		while (src[bytes++] & 0x80)
			;
	return bytes;
}

// Bytes the data part of a block of n values occupies, from the metadata
// alone where the encoding fixes it and by skipping varints otherwise.
template <typename T>
inline size_t data_extent(const encoding_metadata<T> &meta, size_t n, src_bytes_t src)
{
	const auto &d = meta.value_desc;
	switch (d.encoding) {
	case naught:
		return 0;
	case normal:
		return n * sizeof(T);
	case varint:
	case varfor: {
		encoding_descriptor<T> v;
		v.encoding = varint;
		return list_extent(v, n, src);
	}
	case bitpck:
	case bitfor:
		return bitpck_codec<T>::space(n, d.nbits);
	case bitpfr: {
		size_t bytes = bitpck_codec<T>::space(n, d.nbits);
		bytes += list_extent(meta.outlier_index_desc, meta.noutliers, src + bytes);
		bytes += list_extent(meta.outlier_value_desc, meta.noutliers, src + bytes);
		return bytes;
	}
	}
	throw std::invalid_argument("unknown block encoding");
}

template <typename T>
inline void from_record(encoding_metadata<T> &meta, const oroch_block_desc &r)
{
	using unsigned_t = typename integer_traits<T>::unsigned_t;
	meta.clear();
	meta.value_desc.encoding = static_cast<encoding_t>(r.encoding);
	meta.value_desc.dataspace = r.dataspace;
	meta.value_desc.metaspace = size_t(r.metaspace) - 1;
	meta.value_desc.origin = static_cast<T>(static_cast<unsigned_t>(r.origin));
	meta.value_desc.nbits = r.nbits;
	if (r.encoding == bitpfr) {
		meta.noutliers = r.noutliers;
		meta.outlier_index_desc.encoding = static_cast<encoding_t>(r.index_encoding);
		meta.outlier_index_desc.nbits = r.index_nbits;
		meta.outlier_value_desc.encoding = static_cast<encoding_t>(r.value_encoding);
		meta.outlier_value_desc.nbits = r.value_nbits;
	}
}

// encoding_metadata<T> <-> the descriptor of the whole-sequence entry points.
template <typename T>
inline void from_record(encoding_metadata<T> &meta, const oroch_sequence_desc &r)
{
	using unsigned_t = typename integer_traits<T>::unsigned_t;
	meta.clear();
	meta.value_desc.encoding = static_cast<encoding_t>(r.encoding);
	meta.value_desc.dataspace = size_t(r.dataspace);
	meta.value_desc.metaspace = size_t(r.metaspace) - 1;
	meta.value_desc.origin = static_cast<T>(static_cast<unsigned_t>(r.origin));
	meta.value_desc.nbits = r.nbits;
	if (r.encoding == bitpfr) {
		meta.noutliers = size_t(r.noutliers);
		meta.outlier_index_desc.encoding = static_cast<encoding_t>(r.index_encoding);
		meta.outlier_index_desc.nbits = r.index_nbits;
		meta.outlier_value_desc.encoding = static_cast<encoding_t>(r.value_encoding);
		meta.outlier_value_desc.nbits = r.value_nbits;
	}
}

template <typename T>
inline oroch_sequence_desc to_record(const encoding_metadata<T> &meta, size_t n)
{
	using unsigned_t = typename integer_traits<T>::unsigned_t;
	oroch_sequence_desc r;
	std::memset(&r, 0, sizeof r);
	r.encoding = byte_t(meta.value_desc.encoding);
	r.origin = uint64_t(static_cast<unsigned_t>(meta.value_desc.origin));
	r.nbits = byte_t(meta.value_desc.nbits);
	r.dataspace = meta.value_desc.dataspace;
	r.metaspace = uint32_t(meta.metaspace());
	r.nvalues = n;
	r.noutliers = meta.noutliers;
	r.index_encoding = byte_t(meta.outlier_index_desc.encoding);
	r.index_nbits = byte_t(meta.outlier_index_desc.nbits);
	r.value_encoding = byte_t(meta.outlier_value_desc.encoding);
	r.value_nbits = byte_t(meta.outlier_value_desc.nbits);
	return r;
}

// Whether two metadata objects say the same about how values are coded.
template <typename T>
inline bool same_coding(const encoding_metadata<T> &a, const encoding_metadata<T> &b)
{
	const encoding_t e = a.value_desc.encoding;
	if (e != b.value_desc.encoding)
		return false;
	if ((e == naught || e == varfor || e == bitfor || e == bitpfr) && a.value_desc.origin != b.value_desc.origin)
		return false;
	if ((e == bitpck || e == bitfor || e == bitpfr) && a.value_desc.nbits != b.value_desc.nbits)
		return false;
	if (e != bitpfr)
		return true;
	const bool ia = a.outlier_index_desc.encoding == bitpck, ib = b.outlier_index_desc.encoding == bitpck;
	const bool va = a.outlier_value_desc.encoding == bitpck, vb = b.outlier_value_desc.encoding == bitpck;
	return a.noutliers == b.noutliers && ia == ib && va == vb
	       && (!ia || a.outlier_index_desc.nbits == b.outlier_index_desc.nbits)
	       && (!va || a.outlier_value_desc.nbits == b.outlier_value_desc.nbits);
}

inline size_t align8(size_t x)
{
	return (x + 7) & ~size_t(7);
}

} // namespace detail

template <typename T>
class integer_codec
{
public:
	using original_t = T;
	using unsigned_t = typename integer_traits<original_t>::unsigned_t;
	using metadata = encoding_metadata<original_t>;

	// integer_codec.h:234-384.  The result is that of the reference on a
	// freshly constructed metadata object.
	template <typename Iter>
	static void select(metadata &meta, Iter const src, Iter const end)
	{
		const std::vector<T> in = hip::gather<T>(src, end);
		if (in.empty()) {
			meta.clear();
			return;
		}
		if (in.size() <= OROCH_BLOCK_VALUES) {
			oroch_block_desc rec;
			hip::check(oroch_group_select_host(hip::dtype_of<T>(), in.data(), in.size(), &rec),
				   "integer_codec::select");
			detail::from_record(meta, rec);
			return;
		}
		oroch_sequence_desc rec;
		hip::check(oroch_sequence_select_host(hip::dtype_of<T>(), in.data(), in.size(), &rec),
			   "integer_codec::select");
		detail::from_record(meta, rec);
	}

	// integer_codec.h:386-393.  A block whose metadata is what select()
	// makes of it takes the fused block kernel (one launch selects and
	// encodes); any other metadata, and any longer range, is coded as the
	// metadata says by the whole-sequence kernels.
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end, metadata &meta)
	{
		const std::vector<T> in = hip::gather<T>(src, end);
		if (in.empty())
			return;
		if (in.size() <= OROCH_BLOCK_VALUES) {
			std::vector<byte_t> image(oroch_group_bound(hip::dtype_of<T>(), in.size()));
			uint64_t offsets[2], total = 0;
			hip::check(oroch_group_encode_host(hip::dtype_of<T>(), in.data(), in.size(), image.data(),
							   image.size(), offsets, &total),
				   "integer_codec::encode");
			metadata made;
			src_bytes_t p = image.data();
			made.decode(p);
			if (detail::same_coding(made, meta)) {
				const size_t at = detail::align8(size_t(p - image.data()));
				const size_t bytes = detail::data_extent(made, in.size(), image.data() + at);
				std::memcpy(dst, image.data() + at, bytes);
				dst += bytes;
				return;
			}
		}
		const oroch_sequence_desc rec = detail::to_record(meta, in.size());
		// room for what this metadata can produce: packed words, two lists of
		// varints at their longest, or one varint per value at its longest
		const size_t w = sizeof(T), longest = (8 * w + 6) / 7;
		size_t room = in.size() * (w > longest ? w : longest) + 16;
		if (rec.encoding == bitpfr)
			room = oroch_bitpck_space(in.size(), rec.nbits ? rec.nbits : 1) + size_t(rec.noutliers) * 20 + 48;
		std::vector<byte_t> data(room);
		uint64_t bytes = 0;
		hip::check(oroch_sequence_encode_host(hip::dtype_of<T>(), in.data(), in.size(), &rec, data.data(),
						      data.size(), &bytes),
			   "integer_codec::encode");
		std::memcpy(dst, data.data(), size_t(bytes));
		dst += bytes;
	}

	// integer_codec.h:395-402.
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src, metadata &meta)
	{
		const size_t n = size_t(std::distance(dst, end));
		if (n == 0)
			return;
		const size_t bytes = detail::data_extent(meta, n, src);
		std::vector<T> out(n);
		if (n <= OROCH_BLOCK_VALUES) {
			byte_t head[48] = {};
			dst_bytes_t h = head;
			meta.encode(h);
			const size_t at = detail::align8(size_t(h - head));
			std::vector<byte_t> image(((at + bytes + 15) & ~size_t(15)) + 16);
			std::memcpy(image.data(), head, at);
			std::memcpy(image.data() + at, src, bytes);
			const uint64_t offsets[2] = {0, (at + bytes + 15) & ~uint64_t(15)};
			hip::check(oroch_group_decode_host(hip::dtype_of<T>(), image.data(), offsets, n, out.data()),
				   "integer_codec::decode");
		} else {
			const oroch_sequence_desc rec = detail::to_record(meta, n);
			hip::check(oroch_sequence_decode_host(hip::dtype_of<T>(), src, bytes, n, &rec, out.data(),
							      nullptr),
				   "integer_codec::decode");
		}
		std::copy(out.begin(), out.end(), dst);
		src += bytes;
	}
};

} // namespace oroch

#endif /* OROCH_INTEGER_CODEC_H_ */
