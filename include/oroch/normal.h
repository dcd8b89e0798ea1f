// normal.h -- values stored as they are in memory (little endian).
// Stands in for /root/reference/oroch/normal.h:32-64.  The standalone form is
// a byte copy; inside a block image the device kernels do it.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_NORMAL_H_
#define OROCH_NORMAL_H_

#include <cstring>

#include "common.h"

namespace oroch {

template <typename T>
class normal_codec
{
public:
	using original_t = T;

	// normal.h:37-40
	static constexpr size_t space(size_t nvalues)
	{
		return nvalues * sizeof(original_t);
	}

	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter const end)
	{
		for (; src != end; ++src) {
			const original_t v = *src;
			std::memcpy(dst, &v, sizeof v);
			dst += sizeof v;
		}
	}
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &src)
	{
		for (; dst != end; ++dst) {
			original_t v;
			std::memcpy(&v, src, sizeof v);
			src += sizeof v;
			*dst = v;
		}
	}
};

} // namespace oroch

#endif /* OROCH_NORMAL_H_ */
