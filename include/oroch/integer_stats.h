// integer_stats.h -- count, minimum, maximum and bit-length histogram of a
// sequence, computed on the GPU.
// Stands in for /root/reference/oroch/integer_stats.h:35-113.
//
// The device produces one record per block of OROCH_BLOCK_VALUES values
// (oroch_group_stats); the constructor folds the records of a longer sequence
// into one.  The histogram bins the bit length of (v - min()), so every block
// handed to the device for it carries min() as one extra value: the block
// minimum then equals min() and the extra value is taken out of bin 0 again.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_INTEGER_STATS_H_
#define OROCH_INTEGER_STATS_H_

#include <array>
#include <limits>

#include "common.h"
#include "integer_traits.h"

namespace oroch {

template <typename T>
class integer_stats
{
public:
	using original_t = T;
	using unsigned_t = typename integer_traits<original_t>::unsigned_t;

	static constexpr size_t nbits = integer_traits<original_t>::nbits;

	// integer_stats.h:48-53.
	template <typename Iter>
	integer_stats(Iter src, Iter const end)
	{
		const std::vector<T> in = hip::gather<T>(src, end);
		if (in.empty())
			return;
		std::vector<oroch_block_stats> rec((in.size() + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES);
		hip::check(oroch_group_stats_host(hip::dtype_of<T>(), in.data(), in.size(), rec.data()),
			   "integer_stats");
		for (const oroch_block_stats &r : rec) {
			nvalues_ += r.nvalues;
			const original_t lo = static_cast<original_t>(static_cast<unsigned_t>(r.min));
			const original_t hi = static_cast<original_t>(static_cast<unsigned_t>(r.max));
			if (minvalue_ > lo)
				minvalue_ = lo;
			if (maxvalue_ < hi)
				maxvalue_ = hi;
		}
	}

	size_t nvalues() const
	{
		return nvalues_;
	}
	original_t min() const
	{
		return minvalue_;
	}
	original_t max() const
	{
		return maxvalue_;
	}
	size_t original_space() const
	{
		return nvalues() * sizeof(original_t);
	}

	// integer_stats.h:74-80.  Values of the range must not be below min().
	template <typename Iter>
	void build_histogram(Iter src, Iter const end)
	{
		constexpr size_t chunk = OROCH_BLOCK_VALUES - 1;
		std::vector<T> in;
		size_t k = 0;
		for (; src != end; ++src) {
			in.push_back(static_cast<T>(*src));
			if (++k == chunk) {
				in.push_back(minvalue_);
				k = 0;
			}
		}
		if (k)
			in.push_back(minvalue_);
		if (in.empty())
			return;
		std::vector<oroch_block_stats> rec((in.size() + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES);
		hip::check(oroch_group_stats_host(hip::dtype_of<T>(), in.data(), in.size(), rec.data()),
			   "integer_stats::build_histogram");
		for (const oroch_block_stats &r : rec) {
			if (static_cast<original_t>(static_cast<unsigned_t>(r.min)) != minvalue_)
				throw std::domain_error("integer_stats::build_histogram: value below min()");
			for (size_t b = 0; b <= nbits; b++)
				histogram_[b] += r.histogram[b];
			histogram_[0] -= 1;
		}
	}

	size_t histogram(std::size_t index) const
	{
		return histogram_[index];
	}

private:
	size_t nvalues_ = 0;
	original_t minvalue_ = std::numeric_limits<original_t>::max();
	original_t maxvalue_ = std::numeric_limits<original_t>::min();
	std::array<size_t, nbits + 1> histogram_ = {};
};

} // namespace oroch

#endif /* OROCH_INTEGER_STATS_H_ */
