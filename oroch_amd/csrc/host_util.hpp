// host_util.hpp -- host-side helpers shared by the files that implement the C ABI.
#pragma once
#include <functional>

#include <cstddef>
#include <cstdint>
#include <hip/hip_runtime.h>

#include "../../include/oroch_hip.h"

namespace oroch_hip {

// Device memory released at scope exit.
struct dev_buf
{
	void *p = nullptr;
	dev_buf() = default;
	dev_buf(const dev_buf &) = delete;
	dev_buf &operator=(const dev_buf &) = delete;
	~dev_buf()
	{
		release();
	}
	int alloc(size_t bytes)
	{
		release();
		return int(hipMalloc(&p, bytes ? bytes : 16));
	}
	void release()
	{
		if (p)
			(void) hipFree(p);
		p = nullptr;
	}
	void swap(dev_buf &o)
	{
		void *t = p;
		p = o.p;
		o.p = t;
	}
};

// The serialised container: magic and where the block images start after the
// header and the table of nblocks + 1 offsets.
constexpr uint8_t CONTAINER_MAGIC[8] = {'O', 'R', 'O', 'C', 'H', 'B', 'K', '1'};
inline size_t container_stream_offset(size_t nblocks)
{
	return (sizeof(oroch_container_header) + (nblocks + 1) * sizeof(uint64_t) + 15) & ~size_t(15);
}

// Copies between pageable host memory and the device, staged through pinned
// buffers; returns after the data has arrived.  (api.hip)
int staged_copy(void *dst, const void *src, size_t bytes, bool to_device, hipStream_t st);

// Runs fn with the pipe behind the *_host entry points of the current device
// (made on first use, its chunks sized for `bytes` of values, kept), under that
// device's lock.  (pipeline.hip)
int with_default_pipe(size_t bytes, const std::function<int(oroch_pipe *)> &fn);

// memcpy by a few threads (pageable <-> pinned copies are bound by one core otherwise).  (api.hip)
void parallel_memcpy(void *dst, const void *src, size_t bytes);

// Whether p is host memory the device can reach directly (hipHostMalloc,
// hipHostRegister, torch's pin_memory): copies from and to it need no staging.
inline bool is_pinned(const void *p)
{
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost)
		return true;
	(void) hipGetLastError(); // an ordinary pointer is no error
	return false;
}

} // namespace oroch_hip
