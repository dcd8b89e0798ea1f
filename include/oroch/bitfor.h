// bitfor.h -- bit packing with a frame of reference.
// Stands in for /root/reference/oroch/bitfor.h:36-71: bitpck_codec with
// origin_codec as the value codec.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_BITFOR_H_
#define OROCH_BITFOR_H_

#include "bitpck.h"
#include "origin.h"

namespace oroch {

template <typename T>
class bitfor_codec
{
public:
	using original_t = T;

	// The value codec of the packed words: subtracts the frame's origin
	// (origin.h), and carries the width along for the codec's callers.
	struct parameters : public origin_codec<original_t>
	{
		const size_t nbits;

		parameters(original_t frame, size_t width) : origin_codec<original_t>(frame), nbits(width)
		{
		}
	};

	using basic_codec = bitpck_codec<original_t, parameters>;

	static size_t space(size_t nvalues, size_t nbits)
	{
		return basic_codec::space(nvalues, nbits);
	}

	// bitfor.h:59-69: the width and the origin both come from `p`.
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end, const parameters &p)
	{
		basic_codec::encode(dst, src, end, p.nbits, p);
	}
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src, const parameters &p)
	{
		basic_codec::decode(dst, end, src, p.nbits, p);
	}

	// bitfor.h:71-74: one value, without unpacking its neighbours.
	static original_t fetch(src_bytes_t src, size_t index, const parameters &p)
	{
		return basic_codec::fetch(src, index, p.nbits, p);
	}
};

} // namespace oroch

#endif /* OROCH_BITFOR_H_ */
