// integer_group.h -- one self-describing compressed block, and the batch form
// the GPU path is built around.
// Stands in for /root/reference/oroch/integer_group.h:34-86.
//
// integer_group<T> keeps the reference's class: one block of at most
// OROCH_BLOCK_VALUES values, image = [metadata][zero pad to 8][data].
// integer_blocks<T> is the same thing for a whole array at once -- what the
// reference's callers write as a loop over a vector of integer_group
// (tests/bench/group-bench.cc:50-75) is one device pass here.
#ifndef OROCH_INTEGER_GROUP_H_
#define OROCH_INTEGER_GROUP_H_

#include <memory>

#include "common.h"
#include "integer_codec.h"

namespace oroch {

template <typename T>
class integer_group
{
public:
	using original_t = T;
	using codec = integer_codec<original_t>;

	static constexpr size_t alignment = 8;
	static constexpr size_t alignment_mask = alignment - 1;

	// integer_group.h:44-60.
	template <typename Iter>
	void encode(Iter begin, Iter const end, bool aligned = true)
	{
		const std::vector<T> in = detail::gather_block<T>("integer_group::encode", begin, end);
		if (in.empty()) {
			// integer_codec.h:248-253: normal encoding, no data.
			size_ = aligned ? alignment : 1;
			data_.reset(new byte_t[size_]());
			data_[0] = encoding_t::normal;
			return;
		}
		std::vector<byte_t> image(oroch_group_bound(hip::dtype_of<T>(), in.size()));
		uint64_t offsets[2], total = 0;
		hip::check(oroch_group_encode_host(hip::dtype_of<T>(), in.data(), in.size(), image.data(),
						   image.size(), offsets, &total),
			   "integer_group::encode");
		typename codec::metadata meta;
		src_bytes_t p = image.data();
		meta.decode(p);
		const size_t metabytes = size_t(p - image.data());
		const size_t at = detail::align8(metabytes);
		const size_t bytes = detail::data_extent(meta, in.size(), image.data() + at);
		const size_t offset = aligned ? at : metabytes;
		size_ = offset + bytes;
		data_.reset(new byte_t[size_]());
		std::memcpy(data_.get(), image.data(), metabytes);
		std::memcpy(data_.get() + offset, image.data() + at, bytes);
	}

	// integer_group.h:62-76.
	template <typename Iter>
	void decode(Iter begin, Iter const end, bool aligned = true) const
	{
		typename codec::metadata meta;
		src_bytes_t meta_bytes = data_.get();
		src_bytes_t meta_start = meta_bytes;
		meta.decode(meta_bytes);

		size_t offset = size_t(meta_bytes - meta_start);
		if (aligned)
			offset = (offset + alignment_mask) & ~alignment_mask;

		src_bytes_t data_bytes = data_.get() + offset;
		codec::decode(begin, end, data_bytes, meta);
	}

	// integer_group.h:78-82.
	void decode(typename codec::metadata &meta) const
	{
		src_bytes_t meta_bytes = data_.get();
		meta.decode(meta_bytes);
	}

	// Not in the reference: the image and its length, for tests and for
	// callers that store the block.
	const byte_t *data() const
	{
		return data_.get();
	}
	size_t size() const
	{
		return size_;
	}

protected:
	std::unique_ptr<byte_t[]> data_;
	size_t size_ = 0;
};

// A whole array as consecutive blocks of OROCH_BLOCK_VALUES values, encoded
// and decoded by one device pass each.  stream() holds the block images back
// to back, image i at offsets()[i] (a multiple of 16); each image is exactly
// what integer_group<T>::encode(aligned = true) produces for that block,
// followed by zero bytes up to the next multiple of 16.
template <typename T>
class integer_blocks
{
public:
	using original_t = T;
	static constexpr size_t block_values = OROCH_BLOCK_VALUES;

	template <typename Iter>
	void encode(Iter begin, Iter const end)
	{
		const std::vector<T> in = hip::gather<T>(begin, end);
		nvalues_ = in.size();
		const size_t nblocks = (nvalues_ + block_values - 1) / block_values;
		offsets_.assign(nblocks + 1, 0);
		stream_.assign(oroch_group_bound(hip::dtype_of<T>(), nvalues_) + 16, 0);
		uint64_t total = 0;
		if (nvalues_)
			hip::check(oroch_group_encode_host(hip::dtype_of<T>(), in.data(), nvalues_,
							   stream_.data(), stream_.size(), offsets_.data(),
							   &total),
				   "integer_blocks::encode");
		stream_.resize(size_t(total));
	}

	template <typename Iter>
	void decode(Iter begin, Iter const end) const
	{
		const size_t n = size_t(std::distance(begin, end));
		if (n != nvalues_)
			throw std::length_error("integer_blocks::decode: range does not match the encoded length");
		if (n == 0)
			return;
		std::vector<T> out(n);
		hip::check(oroch_group_decode_host(hip::dtype_of<T>(), stream_.data(), offsets_.data(), n,
						   out.data()),
			   "integer_blocks::decode");
		std::copy(out.begin(), out.end(), begin);
	}

	size_t nvalues() const
	{
		return nvalues_;
	}
	const std::vector<byte_t> &stream() const
	{
		return stream_;
	}
	const std::vector<uint64_t> &offsets() const
	{
		return offsets_;
	}

private:
	size_t nvalues_ = 0;
	std::vector<byte_t> stream_;
	std::vector<uint64_t> offsets_;
};

} // namespace oroch

#endif /* OROCH_INTEGER_GROUP_H_ */
