// common.h -- byte and pointer types shared by every codec signature.
// Stands in for /root/reference/oroch/common.h:29-35.
// Interface after Oroch (Copyright (c) 2015 Aleksey Demakov, MIT License); see NOTICE.
#ifndef OROCH_COMMON_H_
#define OROCH_COMMON_H_

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../oroch_hip.h"

namespace oroch {

using byte_t = uint8_t;
using dst_bytes_t = byte_t *;
using src_bytes_t = const byte_t *;

namespace hip {

// Every device call of the host templates funnels through here: a non-zero
// status of the C ABI becomes an exception (the reference has no error
// channel on this path; misuse there is undefined behaviour).
inline void check(int rc, const char *what)
{
	if (rc != 0)
		throw std::runtime_error(std::string(what) + ": " + oroch_hip_strerror(rc));
}

// dtype code of the C ABI for an integer type.
template <typename T>
constexpr int dtype_of()
{
	static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4 || sizeof(T) == 8,
		      "8-, 16-, 32- and 64-bit integers only");
	return (sizeof(T) == 1 ? 0 : sizeof(T) == 2 ? 2 : sizeof(T) == 4 ? 4 : 6)
	       + (static_cast<T>(-1) > static_cast<T>(0) ? 1 : 0);
}

// Raw bit pattern of an integer, zero-extended: how origins cross the C ABI.
template <typename T>
inline uint64_t raw64(T v)
{
	return static_cast<uint64_t>(static_cast<typename std::make_unsigned<T>::type>(v));
}

// The C ABI moves contiguous arrays; an arbitrary iterator range is gathered
// into one first (and scattered back after a decode).
template <typename T, typename Iter>
inline std::vector<T> gather(Iter src, Iter const end)
{
	std::vector<T> v;
	for (; src != end; ++src)
		v.push_back(static_cast<T>(*src));
	return v;
}

} // namespace hip
} // namespace oroch

#endif /* OROCH_COMMON_H_ */
