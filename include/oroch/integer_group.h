// integer_group.h -- one self-describing compressed block, and the batch form
// the GPU path is built around.
// Stands in for /root/reference/oroch/integer_group.h:34-86.
//
// integer_group<T> keeps the reference's class: one group of any number of
// values, image = [metadata][zero pad to 8][data]; up to OROCH_BLOCK_VALUES
// values take one launch of the block kernels, longer groups the
// whole-sequence kernels.  integer_blocks<T> is a whole array cut into blocks
// of OROCH_BLOCK_VALUES, all coded by one device pass -- what the reference's
// callers write as a loop over a vector of integer_group
// (tests/bench/group-bench.cc:50-75).
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
		const std::vector<T> in = hip::gather<T>(begin, end);
		if (in.empty()) {
			// integer_codec.h:248-253: normal encoding, no data.
			size_ = aligned ? alignment : 1;
			data_.reset(new byte_t[size_]());
			data_[0] = encoding_t::normal;
			return;
		}
		if (in.size() > OROCH_BLOCK_VALUES) {
			encode_long(in, aligned);
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

	// integer_group.h:62-76: the metadata says where the data starts.
	template <typename Iter>
	void decode(Iter begin, Iter const end, bool aligned = true) const
	{
		typename codec::metadata meta;
		src_bytes_t cursor = data_.get();
		meta.decode(cursor);
		const size_t head = size_t(cursor - data_.get());
		src_bytes_t body = data_.get() + (aligned ? detail::align8(head) : head);
		codec::decode(begin, end, body, meta);
	}

	// integer_group.h:78-82: the metadata alone.
	void decode(typename codec::metadata &meta) const
	{
		src_bytes_t cursor = data_.get();
		meta.decode(cursor);
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
	// A group of more values than a block: select, then the data part as the
	// metadata says, both over the whole sequence on the device.
	void encode_long(const std::vector<T> &in, bool aligned)
	{
		oroch_sequence_desc rec;
		hip::check(oroch_sequence_select_host(hip::dtype_of<T>(), in.data(), in.size(), &rec),
			   "integer_group::encode");
		byte_t head[40];
		const int metabytes = oroch_sequence_meta_encode(hip::dtype_of<T>(), &rec, head, sizeof head);
		hip::check(metabytes < 0 ? metabytes : 0, "integer_group::encode");
		const size_t offset = aligned ? detail::align8(size_t(metabytes)) : size_t(metabytes);
		std::vector<byte_t> body(size_t(rec.dataspace) + 16);
		uint64_t bytes = 0;
		hip::check(oroch_sequence_encode_host(hip::dtype_of<T>(), in.data(), in.size(), &rec, body.data(),
						      size_t(rec.dataspace), &bytes),
			   "integer_group::encode");
		size_ = offset + size_t(bytes);
		data_.reset(new byte_t[size_]());
		std::memcpy(data_.get(), head, size_t(metabytes));
		std::memcpy(data_.get() + offset, body.data(), size_t(bytes));
	}

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
