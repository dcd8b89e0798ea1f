// offset.h -- difference against the previous value plus a constant offset.
// Stands in for /root/reference/oroch/offset.h:44-71.  The block path uses
// only offset_codec<size_t, 1, false> (exception positions); on the device
// that is an adjacent difference (encode) and a prefix sum (decode).
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_OFFSET_H_
#define OROCH_OFFSET_H_

#include "integer_traits.h"

namespace oroch {

template <typename T, T offset, bool taken_out>
class offset_codec
{
public:
	using original_t = T;
	using unsigned_t = typename integer_traits<T>::unsigned_t;

	offset_codec(original_t origin) : origin_{static_cast<T>(origin + (taken_out ? offset : 0))}
	{
	}
	unsigned_t value_encode(original_t v)
	{
		const unsigned_t u = static_cast<unsigned_t>(v - origin_);
		origin_ = static_cast<T>(v + offset);
		return u;
	}
	original_t value_decode(unsigned_t u)
	{
		const original_t v = static_cast<original_t>(origin_ + u);
		origin_ = static_cast<T>(v + offset);
		return v;
	}

private:
	original_t origin_;
};

} // namespace oroch

#endif /* OROCH_OFFSET_H_ */
