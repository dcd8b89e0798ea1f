// integer_traits.h -- signed/unsigned mapping and bit length.
// Stands in for /root/reference/oroch/integer_traits.h:135-170; usedcount()
// is the bit length (0 for 0) that bins the histogram and sizes varints.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_INTEGER_TRAITS_H_
#define OROCH_INTEGER_TRAITS_H_

#include <type_traits>

namespace oroch {

template <typename T>
struct integer_traits
{
	using signed_t = typename std::make_signed<T>::type;
	using unsigned_t = typename std::make_unsigned<T>::type;

	static constexpr int nbytes = sizeof(signed_t);
	static constexpr int nbits = nbytes * 8;

	static int clz(unsigned_t value)
	{
		return __builtin_clzll(static_cast<unsigned long long>(value)) - (64 - nbits);
	}
	static int ctz(unsigned_t value)
	{
		return __builtin_ctzll(static_cast<unsigned long long>(value));
	}
	static int popcount(unsigned_t value)
	{
		return __builtin_popcountll(static_cast<unsigned long long>(value));
	}
	static int usedcount(unsigned_t value)
	{
		return value ? 64 - __builtin_clzll(static_cast<unsigned long long>(value)) : 0;
	}
};

} // namespace oroch

#endif /* OROCH_INTEGER_TRAITS_H_ */
