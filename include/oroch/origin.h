// origin.h -- frame-of-reference value codec: u = unsigned(v - origin).
// Stands in for /root/reference/oroch/origin.h:36-58 (wrapping arithmetic).
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_ORIGIN_H_
#define OROCH_ORIGIN_H_

#include "integer_traits.h"

namespace oroch {

template <typename T>
class origin_codec
{
public:
	using original_t = T;
	using unsigned_t = typename integer_traits<T>::unsigned_t;

	origin_codec(original_t origin) : origin_{origin}
	{
	}
	unsigned_t value_encode(original_t v) const
	{
		return static_cast<unsigned_t>(static_cast<unsigned_t>(v) - static_cast<unsigned_t>(origin_));
	}
	original_t value_decode(unsigned_t v) const
	{
		return static_cast<original_t>(static_cast<unsigned_t>(v + static_cast<unsigned_t>(origin_)));
	}
	original_t origin() const
	{
		return origin_;
	}

private:
	const original_t origin_;
};

} // namespace oroch

#endif /* OROCH_ORIGIN_H_ */
