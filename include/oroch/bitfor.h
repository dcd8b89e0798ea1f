// bitfor.h -- bit packing with a frame of reference.
// Stands in for /root/reference/oroch/bitfor.h:36-71: bitpck_codec with
// origin_codec as the value codec.
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

	struct parameters : public origin_codec<original_t>
	{
		parameters(original_t f, size_t n) : origin_codec<original_t>(f), nbits(n)
		{
		}
		const size_t nbits;
	};

	using basic_codec = bitpck_codec<original_t, parameters>;

	static size_t space(size_t nvalues, size_t nbits)
	{
		return basic_codec::space(nvalues, nbits);
	}
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end, const parameters &params)
	{
		basic_codec::encode(dst, src, end, params.nbits, params);
	}
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src, const parameters &params)
	{
		basic_codec::decode(dst, end, src, params.nbits, params);
	}
	// bitfor.h:71-74.
	static original_t fetch(src_bytes_t src, size_t index, const parameters &params)
	{
		return basic_codec::fetch(src, index, params.nbits, params);
	}
};

} // namespace oroch

#endif /* OROCH_BITFOR_H_ */
