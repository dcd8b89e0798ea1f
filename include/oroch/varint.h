// varint.h -- 7-bits-per-byte variable length integers.
// Stands in for /root/reference/oroch/varint.h:46-148.  In the block path the
// varint appears as a cost model, as the coder of metadata scalars and of
// exception lists, and as the varint / varfor block encodings; all of those
// run inside the HIP kernels.  What this header keeps on the host is the
// scalar form (a handful of bytes of metadata).  The sequence form
// (varint.h:124-147, the README's example) goes to the device from
// hip::varint_device_threshold values on when the value codec is the default
// one; shorter sequences, and other value codecs, are a loop over the scalar
// form.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_VARINT_H_
#define OROCH_VARINT_H_

#include <iterator>
#include <type_traits>
#include <cstring>
#include <vector>

#include "common.h"
#include "integer_traits.h"
#include "zigzag.h"

namespace oroch {

namespace hip {

// Sequences of at least this many values are coded on the device (below it
// the trip over PCIe costs more than the loop).
constexpr size_t varint_device_threshold = size_t(1) << 16;

// How many sequence calls took the device route (tests look at it).
inline size_t &varint_device_calls()
{
	static size_t calls = 0;
	return calls;
}

// The values of a range as one array: the range itself when it already is one.
template <typename T>
inline const T *contiguous(const T *src, const T *, std::vector<T> &)
{
	return src;
}
template <typename T>
inline const T *contiguous(T *src, T *, std::vector<T> &)
{
	return src;
}
template <typename T, typename Iter>
inline const T *contiguous(Iter src, Iter const end, std::vector<T> &keep)
{
	if constexpr (std::is_same<Iter, typename std::vector<T>::iterator>::value ||
		      std::is_same<Iter, typename std::vector<T>::const_iterator>::value) {
		return src == end ? nullptr : &*src;
	} else {
		keep = gather<T>(src, end);
		return keep.data();
	}
}

// Where n values can be written as one array: the range itself when it is one
// (then *direct is set), else `keep`, to be copied to the range afterwards.
template <typename T, typename Iter>
inline T *writable(Iter dst, size_t n, std::vector<T> &keep, bool *direct)
{
	if constexpr (std::is_same<Iter, T *>::value || std::is_same<Iter, typename std::vector<T>::iterator>::value) {
		*direct = true;
		return n ? &*dst : nullptr;
	} else {
		*direct = false;
		keep.resize(n);
		return keep.data();
	}
}

// The length of the first n varints at src: they end with the n-th byte whose
// top bit is clear.  Eight bytes at a time while eight more values are wanted
// (eight values take eight bytes at least, so those bytes are the caller's).
inline size_t varint_bytes(const uint8_t *src, size_t n)
{
	size_t at = 0, left = n;
	while (left >= 8) {
		uint64_t w;
		std::memcpy(&w, src + at, 8);
		left -= size_t(__builtin_popcountll(~w & 0x8080808080808080ull));
		at += 8;
	}
	for (; left; at++)
		left -= (src[at] & 0x80) == 0;
	return at;
}

} // namespace hip

template <typename T, typename V = zigzag_codec<T>>
class varint_codec
{
	// the device kernels apply the default value codec (zigzag for signed T)
	static constexpr bool device_codec = std::is_same<V, zigzag_codec<T>>::value;

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
		const size_t n = size_t(std::distance(src, end));
		if (device_codec && n >= hip::varint_device_threshold) {
			std::vector<T> keep;
			const T *values = hip::contiguous<T>(src, end, keep);
			uint64_t bytes = 0;
			hip::check(oroch_varint_space_host(hip::dtype_of<T>(), values, n, &bytes), "varint space");
			hip::varint_device_calls()++;
			return size_t(bytes);
		}
		size_t count = 0;
		while (src != end)
			count += value_space(*src++, vcodec);
		return count;
	}
	// (as in the reference, dst has room for what space() says)
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end, value_codec vcodec = value_codec())
	{
		const size_t n = size_t(std::distance(src, end));
		if (device_codec && n >= hip::varint_device_threshold) {
			std::vector<T> keep;
			const T *values = hip::contiguous<T>(src, end, keep);
			uint64_t bytes = 0;
			hip::check(oroch_varint_encode_host(hip::dtype_of<T>(), values, n, dst, ~size_t(0), &bytes),
				   "varint encode");
			hip::varint_device_calls()++;
			dst += bytes;
			return;
		}
		while (src != end)
			value_encode(dst, *src++, vcodec);
	}
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src, value_codec vcodec = value_codec())
	{
		const size_t n = size_t(std::distance(dst, end));
		if (device_codec && n >= hip::varint_device_threshold) {
			// The bytes of n values end with the n-th byte that has its top bit
			// clear; only they are the caller's to read.
			const size_t nbytes = hip::varint_bytes(src, n);
			std::vector<T> keep;
			bool direct;
			T *values = hip::writable<T>(dst, n, keep, &direct);
			uint64_t used = 0;
			hip::check(oroch_varint_decode_host(hip::dtype_of<T>(), src, nbytes, n, values, &used),
				   "varint decode");
			hip::varint_device_calls()++;
			if (!direct)
				for (size_t i = 0; i < n; i++)
					*dst++ = values[i];
			src += used;
			return;
		}
		while (dst != end)
			value_decode(*dst++, src, vcodec);
	}
};

} // namespace oroch

#endif /* OROCH_VARINT_H_ */
