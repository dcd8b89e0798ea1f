// naught.h -- the encoding of a constant run: no data bytes at all.
// Stands in for /root/reference/oroch/naught.h:32-58.  Inside a block image
// the device encoder and decoder handle it; the standalone form moves no data.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_NAUGHT_H_
#define OROCH_NAUGHT_H_

#include "common.h"

namespace oroch {

template <typename T>
class naught_codec
{
public:
	using original_t = T;

	template <typename Iter>
	static void encode(dst_bytes_t &, Iter, Iter const)
	{
	}
	template <typename Iter>
	static void decode(Iter dst, Iter const end, src_bytes_t &, original_t value)
	{
		for (; dst != end; ++dst)
			*dst = value;
	}
};

} // namespace oroch

#endif /* OROCH_NAUGHT_H_ */
