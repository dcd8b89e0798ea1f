// bitpck.h -- fixed-width packing into 16-byte blocks, computed on the GPU.
// Stands in for /root/reference/oroch/bitpck.h:42-258: value k of a block sits
// at bit k * nbits of a 128-bit little-endian word, floor(128 / nbits) values
// per block.  capacity / block_number / space are host arithmetic; encode and
// decode hand the whole sequence to the HIP library (oroch_pack_host /
// oroch_unpack_host), which stages it through pinned memory.  Callers that
// keep data on the device use the C ABI in oroch_hip.h directly.
//
// The kernels fuse the two value codecs the block path uses: zigzag_codec<T>
// (the default) and origin_codec<T> or a class derived from it.  Any other
// value codec V (bitpck.h:42 takes any class with value_encode / value_decode)
// is applied here, value by value on the host, and the images are packed as
// they are.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_BITPCK_H_
#define OROCH_BITPCK_H_

#include <algorithm>
#include <iterator>

#include "common.h"
#include "integer_traits.h"
#include "origin.h"
#include "zigzag.h"

namespace oroch {

namespace detail {

// 0: zigzag in the kernel, 1: origin in the kernel, 2: applied on the host
template <typename T, typename V>
struct value_codec_kind
{
	static constexpr int kind = std::is_base_of<origin_codec<T>, V>::value ? 1
				    : std::is_same<V, zigzag_codec<T>>::value ? 0 : 2;
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
			pack(in, nbits, vcodec, dst, "bitpck_codec::block_encode");
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
		if constexpr (kind::kind == 2) {
			unsigned_t image = 0;
			hip::check(oroch_bitpck_fetch_host(hip::dtype_of<unsigned_t>(), 0, src, 0, unsigned(nbits), &at,
							   1, &image),
				   "bitpck_codec::block_fetch");
			return vcodec.value_decode(image);
		} else {
			T out = T();
			hip::check(oroch_bitpck_fetch_host(hip::dtype_of<T>(), kind::kind, src, kind::origin(vcodec),
							   unsigned(nbits), &at, 1, &out),
				   "bitpck_codec::block_fetch");
			return out;
		}
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
		pack(in, nbits, vcodec, dst, "bitpck_codec::encode");
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
		using kind = detail::value_codec_kind<T, V>;
		if constexpr (kind::kind == 2) {
			std::vector<unsigned_t> images(n);
			hip::check(oroch_unpack_host(hip::dtype_of<unsigned_t>(), 0, src, n, 0, unsigned(nbits),
						     images.data()),
				   "bitpck_codec::decode");
			for (size_t i = 0; i < n; i++)
				*dst++ = vcodec.value_decode(images[i]);
		} else {
			std::vector<T> out(n);
			hip::check(oroch_unpack_host(hip::dtype_of<T>(), kind::kind, src, n, kind::origin(vcodec),
						     unsigned(nbits), out.data()),
				   "bitpck_codec::decode");
			std::copy(out.begin(), out.end(), dst);
		}
		src += space(n, nbits);
	}

private:
	using unsigned_t = typename integer_traits<T>::unsigned_t;

	// The values of `in` packed at dst: through the kernel's own value codec, or as
	// the images a codec the kernels do not know makes of them (an unsigned
	// sequence packs as it is).
	static void pack(const std::vector<T> &in, size_t nbits, value_codec &vcodec, byte_t *dst, const char *what)
	{
		using kind = detail::value_codec_kind<T, V>;
		if constexpr (kind::kind == 2) {
			std::vector<unsigned_t> images(in.size());
			for (size_t i = 0; i < in.size(); i++)
				images[i] = vcodec.value_encode(in[i]);
			hip::check(oroch_pack_host(hip::dtype_of<unsigned_t>(), 0, images.data(), images.size(), 0,
						   unsigned(nbits), dst),
				   what);
		} else {
			hip::check(oroch_pack_host(hip::dtype_of<T>(), kind::kind, in.data(), in.size(),
						   kind::origin(vcodec), unsigned(nbits), dst),
				   what);
		}
	}
};

} // namespace oroch

#endif /* OROCH_BITPCK_H_ */
