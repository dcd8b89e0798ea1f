// bitpck.h -- fixed-width packing into 16-byte blocks, computed on the GPU.
// Stands in for /root/reference/oroch/bitpck.h:42-258: value k of a block sits
// at bit k * nbits of a 128-bit little-endian word, floor(128 / nbits) values
// per block.  capacity / block_number / space are host arithmetic; encode and
// decode hand the whole sequence to the HIP library (oroch_pack_host /
// oroch_unpack_host), which stages it through pinned memory.  Callers that
// keep data on the device use the C ABI in oroch_hip.h directly.
//
// The value codec is fused into the kernels, so only the two the block path
// uses are accepted: zigzag_codec<T> (the default) and origin_codec<T> or a
// class derived from it.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_BITPCK_H_
#define OROCH_BITPCK_H_

#include <algorithm>
#include <iterator>

#include "common.h"
#include "origin.h"
#include "zigzag.h"

namespace oroch {

namespace detail {

template <typename T, typename V>
struct value_codec_kind
{
	static_assert(std::is_same<V, zigzag_codec<T>>::value ||
			      std::is_base_of<origin_codec<T>, V>::value,
		      "the HIP kernels fuse zigzag_codec or origin_codec only");
	static constexpr int kind = std::is_base_of<origin_codec<T>, V>::value ? 1 : 0;
	static uint64_t origin(const V &vcodec)
	{
		if constexpr (kind == 1)
			return hip::raw64<T>(vcodec.origin());
		else
			return 0;
	}
};

} // namespace detail

template <typename T, typename V = zigzag_codec<T>>
class bitpck_codec
{
public:
	using original_t = T;
	using value_codec = V;

	static constexpr size_t block_size = 16;
	static constexpr size_t block_nbits = block_size * 8;

	static constexpr size_t capacity(size_t nbits)
	{
		return block_nbits / nbits;
	}
	static constexpr size_t block_number(size_t nvalues, size_t nbits)
	{
		return (nvalues + capacity(nbits) - 1) / capacity(nbits);
	}
	static constexpr size_t space(size_t nvalues, size_t nbits)
	{
		return block_size * block_number(nvalues, nbits);
	}

	// bitpck.h:73-128: one 16-byte block from at most capacity(nbits)
	// values; src and dst advance.
	template <typename Iter>
	static void block_encode(dst_bytes_t &dst, Iter &src, Iter const end, const size_t nbits,
				 value_codec &vcodec)
	{
		const size_t c = capacity(nbits);
		std::vector<T> in;
		for (; src != end && in.size() < c; ++src)
			in.push_back(static_cast<T>(*src));
		if (in.empty()) {
			std::fill_n(dst, block_size, byte_t(0));
		} else {
			using kind = detail::value_codec_kind<T, V>;
			hip::check(oroch_pack_host(hip::dtype_of<T>(), kind::kind, in.data(), in.size(),
						   kind::origin(vcodec), unsigned(nbits), dst),
				   "bitpck_codec::block_encode");
		}
		dst += block_size;
	}

	// bitpck.h:130-169: a block that may hold fewer values than its capacity.
	template <typename Iter>
	static void block_decode(Iter dst, Iter const end, src_bytes_t &src, const size_t nbits,
				 value_codec &vcodec)
	{
		const size_t n = std::min(size_t(std::distance(dst, end)), capacity(nbits));
		src_bytes_t s = src;
		if (n)
			decode(dst, dst + n, s, nbits, vcodec);
		src += block_size;
	}

	// bitpck.h:171-201: a full block.
	template <typename Iter>
	static void block_decode(Iter dst, src_bytes_t &src, const size_t nbits, value_codec &vcodec)
	{
		src_bytes_t s = src;
		decode(dst, dst + capacity(nbits), s, nbits, vcodec);
		src += block_size;
	}

	// bitpck.h:203-225: value `index` of one block.
	static original_t block_fetch(src_bytes_t src, const size_t index, const size_t nbits,
				      value_codec &vcodec)
	{
		using kind = detail::value_codec_kind<T, V>;
		const uint64_t at = index;
		T out = T();
		hip::check(oroch_bitpck_fetch_host(hip::dtype_of<T>(), kind::kind, src, kind::origin(vcodec),
						   unsigned(nbits), &at, 1, &out),
			   "bitpck_codec::block_fetch");
		return out;
	}

	// bitpck.h:258-266: value `index` of a packed sequence.
	static original_t fetch(src_bytes_t src, const size_t index, const size_t nbits,
				value_codec vcodec = value_codec())
	{
		const size_t c = capacity(nbits);
		return block_fetch(src + (index / c) * block_size, index % c, nbits, vcodec);
	}

	// bitpck.h:227-240.  dst advances by space(n, nbits).
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end, size_t nbits,
			   value_codec vcodec = value_codec())
	{
		const std::vector<T> in = hip::gather<T>(src, end);
		if (in.empty())
			return;
		using kind = detail::value_codec_kind<T, V>;
		hip::check(oroch_pack_host(hip::dtype_of<T>(), kind::kind, in.data(), in.size(),
					   kind::origin(vcodec), unsigned(nbits), dst),
			   "bitpck_codec::encode");
		dst += space(in.size(), nbits);
	}

	// bitpck.h:242-256.  src advances by space(n, nbits).
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src, size_t nbits,
			   value_codec vcodec = value_codec())
	{
		const size_t n = size_t(std::distance(dst, end));
		if (n == 0)
			return;
		std::vector<T> out(n);
		using kind = detail::value_codec_kind<T, V>;
		hip::check(oroch_unpack_host(hip::dtype_of<T>(), kind::kind, src, n, kind::origin(vcodec),
					     unsigned(nbits), out.data()),
			   "bitpck_codec::decode");
		std::copy(out.begin(), out.end(), dst);
		src += space(n, nbits);
	}
};

} // namespace oroch

#endif /* OROCH_BITPCK_H_ */
