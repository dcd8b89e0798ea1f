// zigzag.h -- zigzag map of signed values onto unsigned ones; identity for
// unsigned element types.  Stands in for /root/reference/oroch/zigzag.h:38-94.
// On the device the same map is fused into the pack / unpack kernels; this
// class is the value-codec tag the templates are parameterised with, plus the
// scalar form used for metadata.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_ZIGZAG_H_
#define OROCH_ZIGZAG_H_

#include <type_traits>

#include "integer_traits.h"

namespace oroch {

template <typename T>
struct zigzag_codec
{
	using original_t = T;
	using signed_t = typename integer_traits<T>::signed_t;
	using unsigned_t = typename integer_traits<T>::unsigned_t;

	static unsigned_t encode(signed_t s)
	{
		const unsigned_t u = static_cast<unsigned_t>(s);
		return static_cast<unsigned_t>((u << 1) ^ (s < 0 ? ~unsigned_t(0) : unsigned_t(0)));
	}
	static signed_t decode(unsigned_t u)
	{
		return static_cast<signed_t>((u >> 1) ^ (unsigned_t(0) - (u & 1)));
	}
	static unsigned_t encode_if_signed(original_t v)
	{
		if constexpr (std::is_signed<T>::value)
			return encode(v);
		else
			return v;
	}
	static original_t decode_if_signed(unsigned_t v)
	{
		if constexpr (std::is_signed<T>::value)
			return decode(v);
		else
			return v;
	}
	unsigned_t value_encode(original_t v) const
	{
		return encode_if_signed(v);
	}
	original_t value_decode(unsigned_t v) const
	{
		return decode_if_signed(v);
	}
};

} // namespace oroch

#endif /* OROCH_ZIGZAG_H_ */
