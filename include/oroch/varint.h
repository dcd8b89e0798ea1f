// varint.h -- 7-bits-per-byte variable length integers.
// Stands in for /root/reference/oroch/varint.h:46-148.  In the block path the
// varint appears as a cost model, as the coder of metadata scalars and of
// exception lists, and as the varint / varfor block encodings; all of those
// run inside the HIP kernels.  What this header keeps on the host is the
// scalar form (a handful of bytes of metadata) and, for interface parity, the
// sequence form written as a loop over it.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_VARINT_H_
#define OROCH_VARINT_H_

#include "common.h"
#include "integer_traits.h"
#include "zigzag.h"

namespace oroch {

template <typename T, typename V = zigzag_codec<T>>
class varint_codec
{
public:
	using original_t = T;
	using unsigned_t = typename integer_traits<T>::unsigned_t;
	using value_codec = V;

	static constexpr size_t nbits_space(size_t nbits)
	{
		return (nbits + 6) / 7;
	}
	static constexpr size_t nbytemax = nbits_space(integer_traits<T>::nbits);

	static size_t value_space(original_t src, value_codec vcodec = value_codec())
	{
		const unsigned_t value = vcodec.value_encode(src);
		return value == 0 ? 1 : nbits_space(integer_traits<unsigned_t>::usedcount(value));
	}
	static void value_encode(dst_bytes_t &dst, original_t src, value_codec vcodec = value_codec())
	{
		unsigned_t value = vcodec.value_encode(src);
		while (value >= 0x80) {
			*dst++ = byte_t(value | 0x80);
			value = static_cast<unsigned_t>(value >> 7);
		}
		*dst++ = byte_t(value);
	}
	static original_t value_decode(src_bytes_t &src, value_codec vcodec = value_codec())
	{
		unsigned_t value = 0;
		for (unsigned shift = 0;; shift += 7) {
			const unsigned_t byte = *src++;
			if (shift < unsigned(integer_traits<T>::nbits))
				value = static_cast<unsigned_t>(value | (static_cast<unsigned_t>(byte & 0x7f) << shift));
			if ((byte & 0x80) == 0)
				break;
		}
		return vcodec.value_decode(value);
	}
	static void value_decode(original_t &dst, src_bytes_t &src, value_codec vcodec = value_codec())
	{
		dst = value_decode(src, vcodec);
	}
	template <typename Iter>
	static size_t space(Iter src, Iter const end, value_codec vcodec = value_codec())
	{
		size_t count = 0;
		while (src != end)
			count += value_space(*src++, vcodec);
		return count;
	}
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end, value_codec vcodec = value_codec())
	{
		while (src != end)
			value_encode(dst, *src++, vcodec);
	}
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src, value_codec vcodec = value_codec())
	{
		while (dst != end)
			value_decode(*dst++, src, vcodec);
	}
};

} // namespace oroch

#endif /* OROCH_VARINT_H_ */
