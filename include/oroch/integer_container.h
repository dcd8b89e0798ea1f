// integer_container.h -- a whole compressed array as one byte string, for
// files and network buffers.  The reference has no counterpart (integer_array
// keeps one heap buffer per group, integer_array.h:258, and defines no
// serialised form); the layout is described in oroch_hip.h.  Every block image
// inside is byte-identical to integer_group<T>::encode(aligned = true), and any
// range of blocks can be decoded without the rest.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_INTEGER_CONTAINER_H_
#define OROCH_INTEGER_CONTAINER_H_

#include <algorithm>
#include <cstring>
#include <iterator>

#include "common.h"

namespace oroch {

template <typename T>
class integer_container
{
public:
	using original_t = T;
	static constexpr size_t block_values = OROCH_BLOCK_VALUES;

	integer_container() = default;
	// Adopts bytes read from a file or a socket; throws if they are not a
	// container of this element type.
	explicit integer_container(std::vector<byte_t> bytes) : bytes_(std::move(bytes))
	{
		hip::check(oroch_container_check(header(), bytes_.size()), "integer_container");
		if (header()->dtype != uint32_t(hip::dtype_of<T>()))
			throw std::invalid_argument("integer_container: element type does not match");
	}

	template <typename Iter>
	void encode(Iter begin, Iter const end)
	{
		const std::vector<T> in = hip::gather<T>(begin, end);
		bytes_.assign(oroch_container_bound(hip::dtype_of<T>(), in.size()), 0);
		uint64_t total = 0;
		hip::check(oroch_container_encode_host(hip::dtype_of<T>(), in.data(), in.size(), bytes_.data(),
						       bytes_.size(), &total),
			   "integer_container::encode");
		bytes_.resize(size_t(total));
	}

	size_t nvalues() const
	{
		return bytes_.empty() ? 0 : size_t(header()->nvalues);
	}
	size_t nblocks() const
	{
		return bytes_.empty() ? 0 : size_t(header()->nblocks);
	}

	// All values.
	template <typename Iter>
	void decode(Iter begin, Iter const end) const
	{
		if (size_t(std::distance(begin, end)) != nvalues())
			throw std::length_error("integer_container::decode: range does not match the encoded length");
		decode_blocks(0, nblocks(), begin);
	}

	// Blocks [first, first + count): values first * 256 ... to out.
	template <typename Iter>
	void decode_blocks(size_t first, size_t count, Iter out) const
	{
		if (count == 0)
			return;
		const size_t lo = first * block_values;
		const size_t hi = std::min((first + count) * block_values, nvalues());
		std::vector<T> tmp(hi > lo ? hi - lo : 0);
		hip::check(oroch_container_decode_host(bytes_.data(), bytes_.size(), first, count, tmp.data()),
			   "integer_container::decode_blocks");
		std::copy(tmp.begin(), tmp.end(), out);
	}

	const std::vector<byte_t> &bytes() const
	{
		return bytes_;
	}

private:
	const oroch_container_header *header() const
	{
		return reinterpret_cast<const oroch_container_header *>(bytes_.data());
	}
	std::vector<byte_t> bytes_;
};

} // namespace oroch

#endif /* OROCH_INTEGER_CONTAINER_H_ */
