// host_util.hpp -- host-side helpers shared by the files that implement the C ABI.
#pragma once

#include <cstddef>
#include <hip/hip_runtime.h>

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

// Copies between pageable host memory and the device, staged through pinned
// buffers; returns after the data has arrived.  (api.hip)
int staged_copy(void *dst, const void *src, size_t bytes, bool to_device, hipStream_t st);

} // namespace oroch_hip
