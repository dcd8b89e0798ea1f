// bitpfr.h -- patched frame of reference: bitfor on the low nbits of every
// value plus a list of exceptions for the values that do not fit.
// Stands in for /root/reference/oroch/bitpfr.h:36-125.
//
// The reference collects exceptions as a side effect of its value codec while
// packing; here one device pass packs, a second one finds the exceptions and
// writes both lists in sequence order (gap-coded positions as
// offset_codec<size_t, 1, false> produces them, and the high parts).
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_BITPFR_H_
#define OROCH_BITPFR_H_

#include <algorithm>
#include <iterator>
#include <vector>

#include "bitpck.h"
#include "common.h"
#include "offset.h"
#include "origin.h"

namespace oroch {

template <typename T>
class bitpfr_codec
{
public:
	using original_t = T;
	using unsigned_t = typename integer_traits<original_t>::unsigned_t;

	struct exceptions
	{
		std::vector<size_t> indices;
		std::vector<unsigned_t> values;

		size_t index = 0;

		void reset()
		{
			indices.clear();
			values.clear();
			index = 0;
		}
	};

	struct parameters : public origin_codec<original_t>
	{
		using basic_value_codec = origin_codec<original_t>;

		parameters(original_t f, size_t n, exceptions &x)
			: basic_value_codec(f), nbits(n), mask((unsigned_t(1) << n) - 1), excpts(x)
		{
		}
		unsigned_t basic_value_encode(original_t v) const
		{
			return basic_value_codec::value_encode(v);
		}

		// bitpfr.h:75-85, one value at a time (the sequence form runs
		// on the device and leaves the same state behind).
		unsigned_t value_encode(original_t v)
		{
			const unsigned_t u = basic_value_encode(v);
			if ((u & ~mask) != 0) {
				excpts.indices.push_back(excpts.index - next_index);
				excpts.values.push_back(unsigned_t(u >> nbits));
				next_index = excpts.index + 1;
			}
			excpts.index++;
			return u;
		}

		// Position after the last exception seen so far: what the
		// reference keeps inside its index_codec member.
		size_t next_index = 0;
		const size_t nbits;
		const unsigned_t mask;
		exceptions &excpts;
	};

	using basic_codec = bitpck_codec<original_t, parameters>;

	// bitpfr.h:95-99.  Appends to params.excpts; several calls on one
	// parameters object continue the same index sequence.
	template <typename Iter>
	static void encode(dst_bytes_t &dst, Iter src, Iter end, parameters &params)
	{
		const std::vector<T> in = hip::gather<T>(src, end);
		const size_t n = in.size();
		if (n == 0)
			return;
		std::vector<uint64_t> idx(n), val(n);
		uint64_t nexc = 0;
		hip::check(oroch_bitpfr_encode_host(hip::dtype_of<T>(), in.data(), n,
						    hip::raw64<T>(params.origin()), unsigned(params.nbits),
						    dst, idx.data(), val.data(), &nexc),
			   "bitpfr_codec::encode");
		exceptions &x = params.excpts;
		size_t pos = 0; // position within this call, after the previous exception
		for (size_t i = 0; i < nexc; i++) {
			size_t gap = size_t(idx[i]);
			pos += gap;
			if (i == 0)
				gap = x.index + pos - params.next_index;
			x.indices.push_back(gap);
			x.values.push_back(unsigned_t(val[i]));
			pos += 1;
			params.next_index = x.index + pos;
		}
		x.index += n;
		dst += basic_codec::space(n, params.nbits);
	}

	// bitpfr.h:101-106.
	template <typename Iter>
	static void decode(Iter dst, Iter end, src_bytes_t &src, const parameters &params)
	{
		const size_t n = size_t(std::distance(dst, end));
		if (n == 0)
			return;
		const exceptions &x = params.excpts;
		std::vector<uint64_t> idx(x.indices.begin(), x.indices.end());
		std::vector<uint64_t> val(x.values.begin(), x.values.end());
		std::vector<T> out(n);
		hip::check(oroch_bitpfr_decode_host(hip::dtype_of<T>(), src, n, hip::raw64<T>(params.origin()),
						    unsigned(params.nbits), idx.data(), val.data(),
						    idx.size(), out.data()),
			   "bitpfr_codec::decode");
		std::copy(out.begin(), out.end(), dst);
		src += basic_codec::space(n, params.nbits);
	}

	// bitpfr.h:108-112.
	template <typename Iter>
	static void decode_basic(Iter dst, Iter end, src_bytes_t &src, const parameters &params)
	{
		basic_codec::decode(dst, end, src, params.nbits, params);
	}

	// bitpfr.h:114-124.  Touches dst[0 .. last exception position].
	template <typename Iter>
	static void decode_patch(Iter dst, const parameters &params)
	{
		const exceptions &x = params.excpts;
		const size_t nexc = x.indices.size();
		if (nexc == 0)
			return;
		std::vector<uint64_t> idx(x.indices.begin(), x.indices.end());
		std::vector<uint64_t> val(x.values.begin(), x.values.end());
		size_t n = 0; // one past the last patched position
		for (uint64_t g : idx)
			n += size_t(g) + 1;
		std::vector<T> buf(n);
		std::copy_n(dst, n, buf.begin());
		hip::check(oroch_bitpfr_patch_host(hip::dtype_of<T>(), buf.data(), n,
						   hip::raw64<T>(params.origin()), unsigned(params.nbits),
						   idx.data(), val.data(), nexc),
			   "bitpfr_codec::decode_patch");
		std::copy(buf.begin(), buf.end(), dst);
	}
};

} // namespace oroch

#endif /* OROCH_BITPFR_H_ */
