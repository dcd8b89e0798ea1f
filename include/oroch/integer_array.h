// integer_array.h -- a growable array of integers kept compressed in groups of
// 256 values.
// Stands in for /root/reference/oroch/integer_array.h:38-259.
//
// The reference keeps one heap buffer per group and the last 0..255 values in
// a vector; here the groups are one stream of block images in device memory
// (each image what integer_group<T>::encode writes for the group) and the tail
// stays on the host.  find(), at() and insert() keep their meaning; what takes
// a loop over the array there is one device pass here, and the bulk forms
// (append, find and at over many values, read) are added for callers that have
// more than one value at hand.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_INTEGER_ARRAY_H_
#define OROCH_INTEGER_ARRAY_H_

#include <ostream>

#include "common.h"
#include "integer_codec.h"

namespace oroch {

// integer_array.h:40
constexpr size_t not_found = size_t(-1);

namespace detail {

// integer_array.h:44
constexpr size_t group_size = OROCH_BLOCK_VALUES;

} // namespace detail

template <typename T>
class integer_array
{
public:
	using original_t = T;

	integer_array()
	{
		hip::check(oroch_array_create(hip::dtype_of<T>(), &array_), "integer_array");
	}
	~integer_array()
	{
		oroch_array_destroy(array_);
	}
	integer_array(const integer_array &) = delete;
	integer_array &operator=(const integer_array &) = delete;
	integer_array(integer_array &&other) noexcept : array_(other.array_)
	{
		other.array_ = nullptr;
	}

	// integer_array.h:156-164
	bool empty() const
	{
		return size() == 0;
	}
	size_t size() const
	{
		return size_t(oroch_array_size(array_));
	}

	// integer_array.h:166-178.  std::out_of_range for npos >= size().  (The
	// reference's check compares the position within its group with the
	// length of the tail, which turns away valid positions of full groups
	// and lets npos == size() through.)
	original_t at(size_t npos) const
	{
		const uint64_t index = npos;
		original_t value;
		range_check(oroch_array_at(array_, &index, 1, &value), "integer_array::at");
		return value;
	}
	// integer_array.h:180-190 (no range check there; positions are checked here too)
	original_t operator[](size_t npos) const
	{
		return at(npos);
	}
	// Values at many positions in one pass.
	std::vector<original_t> at(const std::vector<size_t> &positions) const
	{
		const std::vector<uint64_t> index(positions.begin(), positions.end());
		std::vector<original_t> values(index.size());
		range_check(oroch_array_at(array_, index.data(), index.size(), values.data()), "integer_array::at");
		return values;
	}

	// integer_array.h:192-208: the first position that holds the value, or not_found.
	size_t find(original_t value) const
	{
		const uint64_t raw = hip::raw64<T>(value);
		uint64_t index = 0;
		hip::check(oroch_array_find(array_, &raw, 1, &index), "integer_array::find");
		return size_t(index);
	}
	// The same for many values in one pass over the array.
	std::vector<size_t> find(const std::vector<original_t> &values) const
	{
		std::vector<uint64_t> raw(values.size()), index(values.size());
		for (size_t i = 0; i < values.size(); i++)
			raw[i] = hip::raw64<T>(values[i]);
		hip::check(oroch_array_find(array_, raw.data(), raw.size(), index.data()), "integer_array::find");
		return std::vector<size_t>(index.begin(), index.end());
	}

	// integer_array.h:210-214
	void clear()
	{
		hip::check(oroch_array_clear(array_), "integer_array::clear");
	}

	// integer_array.h:216-245.  std::out_of_range for array_index > size().
	void insert(size_t array_index, original_t value)
	{
		range_check(oroch_array_insert(array_, array_index, hip::raw64<T>(value)), "integer_array::insert");
	}

	// Appends a range; every group that fills is encoded in the same device pass.
	template <typename Iter>
	void append(Iter begin, Iter const end)
	{
		const std::vector<T> in = hip::gather<T>(begin, end);
		hip::check(oroch_array_append_host(array_, in.data(), in.size()), "integer_array::append");
	}

	// Values [first, first + count).
	std::vector<original_t> read(size_t first, size_t count) const
	{
		std::vector<original_t> values(count);
		range_check(oroch_array_read(array_, first, count, values.data()), "integer_array::read");
		return values;
	}

	// The array as one byte string (the serialised container of
	// integer_container.h: full groups as they are, the tail as a short last
	// block) and back.
	std::vector<byte_t> save() const
	{
		std::vector<byte_t> bytes(oroch_array_save_bound(array_));
		uint64_t total = 0;
		hip::check(oroch_array_save(array_, bytes.data(), bytes.size(), &total), "integer_array::save");
		bytes.resize(size_t(total));
		return bytes;
	}
	void load(const std::vector<byte_t> &bytes)
	{
		hip::check(oroch_array_load(array_, bytes.data(), bytes.size()), "integer_array::load");
	}

	// integer_array.h:247-252: one line of metadata per group.
	void group_info(std::ostream &ostream) const
	{
		std::vector<byte_t> image(sizeof(T) * detail::group_size + 64);
		for (uint64_t group = 0;; group++) {
			uint64_t bytes = 0;
			const int rc = oroch_array_image(array_, group, image.data(), image.size(), &bytes);
			if (rc == OROCH_E_RANGE)
				break;
			hip::check(rc, "integer_array::group_info");
			encoding_metadata<T> meta;
			meta.clear();
			src_bytes_t p = image.data();
			meta.decode(p);
			ostream << meta << std::endl;
		}
	}

	// The C handle, for the entry points that take one (oroch_array_view ...).
	oroch_array *handle() const
	{
		return array_;
	}

private:
	static void range_check(int rc, const char *what)
	{
		if (rc == OROCH_E_RANGE)
			throw std::out_of_range("array index out of range");
		hip::check(rc, what);
	}

	oroch_array *array_ = nullptr;
};

} // namespace oroch

#endif /* OROCH_INTEGER_ARRAY_H_ */
