// api.hip -- library set-up, error text and the host-buffer entry points.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

#include "common.hpp"
#include "host_util.hpp"

namespace oroch_hip {

namespace {
struct record
{
	int id;
	hipEvent_t begin, end;
};
std::mutex g_prof_lock;
thread_local int g_prof_quiet = 0;
bool g_prof_on = false;
size_t g_prof_max = 0;
std::vector<record> g_prof;
} // namespace

no_profile_scope::no_profile_scope()
{
	g_prof_quiet++;
}
no_profile_scope::~no_profile_scope()
{
	g_prof_quiet--;
}

void profile_mark(int kernel_id, hipStream_t st, bool begin)
{
	if (!g_prof_on || g_prof_quiet)
		return;
	std::lock_guard<std::mutex> lock(g_prof_lock);
	if (!g_prof_on)
		return;
	if (begin) {
		if (g_prof.size() >= g_prof_max)
			return;
		record r;
		r.id = kernel_id;
		if (hipEventCreate(&r.begin) != hipSuccess)
			return;
		if (hipEventCreate(&r.end) != hipSuccess) {
			(void) hipEventDestroy(r.begin);
			return;
		}
		(void) hipEventRecord(r.begin, st);
		g_prof.push_back(r);
	} else {
		for (size_t i = g_prof.size(); i-- > 0;) {
			if (g_prof[i].id == kernel_id) {
				(void) hipEventRecord(g_prof[i].end, st);
				break;
			}
		}
	}
}

} // namespace oroch_hip

extern "C" int oroch_hip_profile_start(int max_records)
{
	using namespace oroch_hip;
	std::lock_guard<std::mutex> lock(g_prof_lock);
	for (auto &r : g_prof) {
		(void) hipEventDestroy(r.begin);
		(void) hipEventDestroy(r.end);
	}
	g_prof.clear();
	g_prof_max = max_records > 0 ? size_t(max_records) : 0;
	g_prof_on = g_prof_max > 0;
	return 0;
}

extern "C" int oroch_hip_profile_stop(double *ms, unsigned *launches)
{
	using namespace oroch_hip;
	std::lock_guard<std::mutex> lock(g_prof_lock);
	g_prof_on = false;
	int rc = 0;
	for (int k = 0; k < OROCH_K_COUNT; k++) {
		if (ms)
			ms[k] = 0;
		if (launches)
			launches[k] = 0;
	}
	for (auto &r : g_prof) {
		float t = 0;
		hipError_t e = hipEventSynchronize(r.end);
		if (e == hipSuccess)
			e = hipEventElapsedTime(&t, r.begin, r.end);
		if (e != hipSuccess)
			rc = int(e);
		else if (r.id >= 0 && r.id < OROCH_K_COUNT) {
			if (ms)
				ms[r.id] += t;
			if (launches)
				launches[r.id] += 1;
		}
		(void) hipEventDestroy(r.begin);
		(void) hipEventDestroy(r.end);
	}
	g_prof.clear();
	return rc;
}

namespace oroch_hip {
namespace {
// The switches start from the environment, which is read once.
struct option_table
{
	std::atomic<int> v[OROCH_OPT_COUNT];
	option_table()
	{
		auto env = [](const char *name) {
			const char *e = getenv(name);
			return e && *e ? atoi(e) : 0;
		};
		v[OROCH_OPT_ENCODE_PATH] = env("OROCH_HIP_ENCODE_PATH");
		v[OROCH_OPT_SELECT_ONE_BLOCK] = env("OROCH_HIP_SELECT_ONE_BLOCK");
		v[OROCH_OPT_ENCODE_ONE_BLOCK] = env("OROCH_HIP_ENCODE_ONE_BLOCK");
		v[OROCH_OPT_DECODE_ONE_BLOCK] = env("OROCH_HIP_DECODE_ONE_BLOCK");
		v[OROCH_OPT_FIND_ONE_BLOCK] = env("OROCH_HIP_FIND_ONE_BLOCK");
	}
};
option_table &options()
{
	static option_table t;
	return t;
}
} // namespace

int option(int id)
{
	return options().v[id].load(std::memory_order_relaxed);
}
} // namespace oroch_hip

extern "C" int oroch_hip_set_option(int option, int value)
{
	if (option < 0 || option >= OROCH_OPT_COUNT)
		return OROCH_E_ARGUMENT;
	return oroch_hip::options().v[option].exchange(value);
}

extern "C" int oroch_hip_get_option(int option)
{
	if (option < 0 || option >= OROCH_OPT_COUNT)
		return OROCH_E_ARGUMENT;
	return oroch_hip::option(option);
}

extern "C" int oroch_hip_init(int device)
{
	int count = 0;
	hipError_t rc = hipGetDeviceCount(&count);
	if (rc != hipSuccess || count <= 0)
		return OROCH_E_NOGPU;
	if (device < 0 || device >= count)
		return OROCH_E_ARGUMENT;
	rc = hipSetDevice(device);
	if (rc != hipSuccess)
		return int(rc);
	hipDeviceProp_t prop;
	rc = hipGetDeviceProperties(&prop, device);
	if (rc != hipSuccess)
		return int(rc);
	// the kernels are compiled for gfx950 (64-lane wavefronts) only
	if (prop.warpSize != 64 || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
		return OROCH_E_NOGPU;
	// this device's copy of the scratch array of the non-allocating entry point
	return oroch_hip::static_scratch_setup(nullptr);
}

#define OROCH_STR2(x) #x
#define OROCH_STR(x) OROCH_STR2(x)
extern "C" const char *oroch_hip_build_info(void)
{
	return "hipcc (HIP " OROCH_STR(HIP_VERSION_MAJOR) "." OROCH_STR(HIP_VERSION_MINOR) "." OROCH_STR(HIP_VERSION_PATCH)
	       ", clang " __clang_version__ "), --offload-arch=gfx950 -O3";
}

extern "C" const char *oroch_hip_strerror(int code)
{
	switch (code) {
	case 0: return "success";
	case OROCH_E_DTYPE: return "unknown dtype code";
	case OROCH_E_NBITS: return "nbits outside the valid range for the element type";
	case OROCH_E_BLOCK: return "unsupported values-per-block";
	case OROCH_E_CAPACITY: return "destination or workspace too small";
	case OROCH_E_ARGUMENT: return "null / misaligned pointer or inconsistent sizes";
	case OROCH_E_NOGPU: return "no usable gfx950 HIP device";
	case OROCH_E_RANGE: return "position outside the array";
	case OROCH_E_TOO_LARGE: return "more values than the fixed scratch buffer of the non-allocating entry point holds";
	}
	if (code > 0)
		return hipGetErrorString(hipError_t(code));
	return "unknown error";
}

// memcpy split over a few threads: one core moves 10-12 GB/s between pageable
// and pinned memory, a fifth of what the link to the device carries.
// OROCH_HIP_COPY_THREADS overrides the thread count (default 8, at most 32).
void oroch_hip::parallel_memcpy(void *dst, const void *src, size_t bytes)
{
	static const unsigned nthreads = [] {
		const char *v = getenv("OROCH_HIP_COPY_THREADS");
		long n = v ? atol(v) : 8;
		const unsigned hw = std::thread::hardware_concurrency();
		if (hw && n > long(hw))
			n = long(hw);
		return unsigned(n < 1 ? 1 : n > 32 ? 32 : n);
	}();
	const size_t PIECE = size_t(1) << 20;
	if (nthreads <= 1 || bytes < 4 * PIECE) {
		std::memcpy(dst, src, bytes);
		return;
	}
	const size_t per = ((bytes + nthreads - 1) / nthreads + 4095) & ~size_t(4095);
	std::vector<std::thread> team;
	for (unsigned t = 1; t < nthreads; t++) {
		const size_t lo = per * t;
		if (lo >= bytes)
			break;
		const size_t len = bytes - lo < per ? bytes - lo : per;
		team.emplace_back([=] { std::memcpy(static_cast<char *>(dst) + lo, static_cast<const char *>(src) + lo, len); });
	}
	std::memcpy(dst, src, bytes < per ? bytes : per);
	for (auto &th : team)
		th.join();
}

namespace {

using oroch_hip::parallel_memcpy;

// The two pinned staging buffers are kept for the life of the process (pinning
// 64 MiB costs milliseconds); one staged copy runs at a time.
struct staging
{
	static constexpr size_t CHUNK = size_t(32) << 20;
	std::mutex lock;
	void *pin[2] = {nullptr, nullptr};
	int prepare()
	{
		for (int i = 0; i < 2; i++) {
			if (!pin[i]) {
				int rc = int(hipHostMalloc(&pin[i], CHUNK, hipHostMallocDefault));
				if (rc)
					return rc;
			}
		}
		return 0;
	}
};
staging g_staging;

} // namespace

// Copies between pageable host memory and the device in chunks staged through
// two pinned buffers, so the host memcpy of one chunk overlaps the DMA of the
// other.
int oroch_hip::staged_copy(void *dst, const void *src, size_t bytes, bool to_device, hipStream_t st)
{
	if (bytes == 0)
		return 0;
	// A caller's buffer that is pinned already (hipHostMalloc, hipHostRegister,
	// torch's pin_memory) goes over the link as it is.
	{
		const void *host = to_device ? src : dst;
		hipPointerAttribute_t attr;
		if (hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost) {
			int rc = int(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st));
			return rc ? rc : int(hipStreamSynchronize(st));
		}
		(void) hipGetLastError(); // an ordinary pointer is no error
	}
	std::lock_guard<std::mutex> guard(g_staging.lock);
	const size_t CHUNK = staging::CHUNK;
	int rc = g_staging.prepare();
	if (rc)
		return rc;
	void *const *pin = g_staging.pin;
	hipEvent_t ev[2];
	for (int i = 0; i < 2; i++) {
		if ((rc = int(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)))) {
			if (i)
				(void) hipEventDestroy(ev[0]);
			return rc;
		}
	}
	size_t done = 0;
	int k = 0;
	size_t pend_off[2] = {0, 0}, pend_len[2] = {0, 0};
	bool pending[2] = {false, false};
	rc = 0;
	while (done < bytes && rc == 0) {
		const size_t len = bytes - done < CHUNK ? bytes - done : CHUNK;
		if (pending[k]) {
			rc = int(hipEventSynchronize(ev[k]));
			if (!to_device && rc == 0)
				parallel_memcpy(static_cast<char *>(dst) + pend_off[k], pin[k], pend_len[k]);
			pending[k] = false;
		}
		if (rc)
			break;
		if (to_device) {
			parallel_memcpy(pin[k], static_cast<const char *>(src) + done, len);
			rc = int(hipMemcpyAsync(static_cast<char *>(dst) + done, pin[k], len,
						hipMemcpyHostToDevice, st));
		} else {
			rc = int(hipMemcpyAsync(pin[k], static_cast<const char *>(src) + done, len,
						hipMemcpyDeviceToHost, st));
		}
		if (rc == 0)
			rc = int(hipEventRecord(ev[k], st));
		pend_off[k] = done;
		pend_len[k] = len;
		pending[k] = true;
		done += len;
		k ^= 1;
	}
	for (int i = 0; i < 2; i++) {
		const int j = (k + i) & 1;
		if (pending[j] && rc == 0) {
			rc = int(hipEventSynchronize(ev[j]));
			if (!to_device && rc == 0)
				parallel_memcpy(static_cast<char *>(dst) + pend_off[j], pin[j], pend_len[j]);
		}
	}
	for (int i = 0; i < 2; i++)
		(void) hipEventDestroy(ev[i]);
	return rc;
}


using oroch_hip::dev_buf;
using oroch_hip::staged_copy;

extern "C" int oroch_group_encode_host(int dtype, const void *src_host, size_t n, void *dst_host,
				       size_t dst_capacity, uint64_t *offsets_host,
				       uint64_t *total_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	return oroch_hip::with_default_pipe(n << (dtype >> 1), [&](oroch_pipe *pipe) {
		return oroch_pipe_encode(pipe, dtype, src_host, n, dst_host, dst_capacity, offsets_host, total_host);
	});
}

extern "C" int oroch_group_decode_host(int dtype, const void *src_host,
				       const uint64_t *offsets_host, size_t n, void *dst_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	return oroch_hip::with_default_pipe(n << (dtype >> 1), [&](oroch_pipe *pipe) {
		return oroch_pipe_decode(pipe, dtype, src_host, offsets_host, n, dst_host);
	});
}

extern "C" int oroch_group_select_host(int dtype, const void *src_host, size_t n,
				       oroch_block_desc *desc_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	if (!src_host || !desc_host)
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t nblocks = (n + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
	dev_buf in, out;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(nblocks * sizeof(oroch_block_desc))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	if ((rc = oroch_group_select(dtype, in.p, n, static_cast<oroch_block_desc *>(out.p), nullptr)))
		return rc;
	return staged_copy(desc_host, out.p, nblocks * sizeof(oroch_block_desc), false, nullptr);
}

extern "C" int oroch_group_stats_host(int dtype, const void *src_host, size_t n,
				      oroch_block_stats *stats_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	if (!src_host || !stats_host)
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t nblocks = (n + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
	dev_buf in, out;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(nblocks * sizeof(oroch_block_stats))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	if ((rc = oroch_group_stats(dtype, in.p, n, static_cast<oroch_block_stats *>(out.p), nullptr)))
		return rc;
	return staged_copy(stats_host, out.p, nblocks * sizeof(oroch_block_stats), false, nullptr);
}

extern "C" int oroch_pack_host(int dtype, int vcodec, const void *src_host, size_t n,
			       uint64_t origin, unsigned nbits, void *dst_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	if (!src_host || !dst_host)
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t space = oroch_bitpck_space(n, nbits);
	if (!space)
		return OROCH_E_NBITS;
	dev_buf in, out;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(space)))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	rc = vcodec ? oroch_bitfor_encode(dtype, in.p, n, origin, nbits, out.p, nullptr)
		    : oroch_bitpck_encode(dtype, in.p, n, nbits, out.p, nullptr);
	if (rc)
		return rc;
	return staged_copy(dst_host, out.p, space, false, nullptr);
}

extern "C" int oroch_unpack_host(int dtype, int vcodec, const void *src_host, size_t n,
				 uint64_t origin, unsigned nbits, void *dst_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	if (!src_host || !dst_host)
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t space = oroch_bitpck_space(n, nbits);
	if (!space)
		return OROCH_E_NBITS;
	dev_buf in, out;
	int rc;
	if ((rc = in.alloc(space)) || (rc = out.alloc(bytes)))
		return rc;
	if ((rc = staged_copy(in.p, src_host, space, true, nullptr)))
		return rc;
	rc = vcodec ? oroch_bitfor_decode(dtype, in.p, n, origin, nbits, out.p, nullptr)
		    : oroch_bitpck_decode(dtype, in.p, n, nbits, out.p, nullptr);
	if (rc)
		return rc;
	return staged_copy(dst_host, out.p, bytes, false, nullptr);
}

// Only the 16-byte words that hold the wanted values travel: word number j of
// the upload is the word of query j, and the query is rebased onto it.
extern "C" int oroch_bitpck_fetch_host(int dtype, int vcodec, const void *src_host, uint64_t origin,
				       unsigned nbits, const uint64_t *index_host, size_t m, void *dst_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	const size_t cap = oroch_bitpck_capacity(nbits);
	if (cap == 0 || nbits > (8u << (dtype >> 1)))
		return OROCH_E_NBITS;
	if (m == 0)
		return 0;
	if (!src_host || !index_host || !dst_host)
		return OROCH_E_ARGUMENT;
	try {
		std::vector<uint8_t> words(m * 16);
		std::vector<uint64_t> where(m);
		const uint8_t *src = static_cast<const uint8_t *>(src_host);
		for (size_t j = 0; j < m; j++) {
			memcpy(words.data() + j * 16, src + (index_host[j] / cap) * 16, 16);
			where[j] = j * cap + index_host[j] % cap;
		}
		const size_t out_bytes = m << (dtype >> 1);
		dev_buf in, idx, out;
		int rc;
		if ((rc = in.alloc(m * 16)) || (rc = idx.alloc(m * 8)) || (rc = out.alloc(out_bytes)) ||
		    (rc = staged_copy(in.p, words.data(), m * 16, true, nullptr)) ||
		    (rc = staged_copy(idx.p, where.data(), m * 8, true, nullptr)) ||
		    (rc = oroch_bitpck_fetch(dtype, vcodec, in.p, m * cap, origin, nbits,
					     static_cast<const uint64_t *>(idx.p), m, out.p, nullptr)))
			return rc;
		return staged_copy(dst_host, out.p, out_bytes, false, nullptr);
	} catch (const std::bad_alloc &) {
		return int(hipErrorOutOfMemory);
	}
}

extern "C" int oroch_bitpfr_encode_host(int dtype, const void *src_host, size_t n, uint64_t origin,
					unsigned nbits, void *dst_host, uint64_t *indices_host,
					uint64_t *values_host, uint64_t *nexc_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!nexc_host)
		return OROCH_E_ARGUMENT;
	*nexc_host = 0;
	if (n == 0)
		return 0;
	if (!src_host || !dst_host || !indices_host || !values_host)
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t space = oroch_bitpck_space(n, nbits);
	if (!space)
		return OROCH_E_NBITS;
	dev_buf in, out, idx, val, cnt, ws;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(space)) || (rc = idx.alloc(n * 8)) ||
	    (rc = val.alloc(n * 8)) || (rc = cnt.alloc(8)) || (rc = ws.alloc(oroch_bitpfr_workspace(n))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	rc = oroch_bitpfr_encode(dtype, in.p, n, origin, nbits, out.p, static_cast<uint64_t *>(idx.p),
				 static_cast<uint64_t *>(val.p), n, static_cast<uint64_t *>(cnt.p), ws.p,
				 oroch_bitpfr_workspace(n), nullptr);
	if (rc)
		return rc;
	if ((rc = staged_copy(nexc_host, cnt.p, 8, false, nullptr)) ||
	    (rc = staged_copy(dst_host, out.p, space, false, nullptr)))
		return rc;
	if (*nexc_host) {
		if ((rc = staged_copy(indices_host, idx.p, *nexc_host * 8, false, nullptr)) ||
		    (rc = staged_copy(values_host, val.p, *nexc_host * 8, false, nullptr)))
			return rc;
	}
	return 0;
}

extern "C" int oroch_bitpfr_decode_host(int dtype, const void *src_host, size_t n, uint64_t origin,
					unsigned nbits, const uint64_t *indices_host,
					const uint64_t *values_host, size_t nexc, void *dst_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	if (!src_host || !dst_host || (nexc && (!indices_host || !values_host)))
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t space = oroch_bitpck_space(n, nbits);
	if (!space)
		return OROCH_E_NBITS;
	dev_buf in, out, idx, val, ws;
	int rc;
	if ((rc = in.alloc(space)) || (rc = out.alloc(bytes)) || (rc = idx.alloc(nexc * 8)) ||
	    (rc = val.alloc(nexc * 8)) || (rc = ws.alloc(oroch_bitpfr_workspace(n))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, space, true, nullptr)))
		return rc;
	if (nexc) {
		if ((rc = staged_copy(idx.p, indices_host, nexc * 8, true, nullptr)) ||
		    (rc = staged_copy(val.p, values_host, nexc * 8, true, nullptr)))
			return rc;
	}
	rc = oroch_bitpfr_decode(dtype, in.p, n, origin, nbits, static_cast<const uint64_t *>(idx.p),
				 static_cast<const uint64_t *>(val.p), nexc, out.p, ws.p,
				 oroch_bitpfr_workspace(n), nullptr);
	if (rc)
		return rc;
	return staged_copy(dst_host, out.p, bytes, false, nullptr);
}

extern "C" int oroch_bitpfr_patch_host(int dtype, void *dst_host, size_t n, uint64_t origin,
				       unsigned nbits, const uint64_t *indices_host,
				       const uint64_t *values_host, size_t nexc)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (nexc == 0)
		return 0;
	if (!dst_host || !indices_host || !values_host || nexc > n)
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	dev_buf out, idx, val, ws;
	int rc;
	if ((rc = out.alloc(bytes)) || (rc = idx.alloc(nexc * 8)) || (rc = val.alloc(nexc * 8)) ||
	    (rc = ws.alloc(oroch_bitpfr_workspace(n))))
		return rc;
	if ((rc = staged_copy(out.p, dst_host, bytes, true, nullptr)) ||
	    (rc = staged_copy(idx.p, indices_host, nexc * 8, true, nullptr)) ||
	    (rc = staged_copy(val.p, values_host, nexc * 8, true, nullptr)))
		return rc;
	rc = oroch_bitpfr_patch(dtype, out.p, n, origin, nbits, static_cast<const uint64_t *>(idx.p),
				static_cast<const uint64_t *>(val.p), nexc, ws.p,
				oroch_bitpfr_workspace(n), nullptr);
	if (rc)
		return rc;
	return staged_copy(dst_host, out.p, bytes, false, nullptr);
}

// ---- serialised container ------------------------------------------------------------

using oroch_hip::CONTAINER_MAGIC;
using oroch_hip::container_stream_offset;

extern "C" size_t oroch_container_bound(int dtype, size_t n)
{
	const size_t nblocks = (n + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
	return container_stream_offset(nblocks) + oroch_group_bound(dtype, n);
}

extern "C" int oroch_container_encode(int dtype, const void *src, size_t n, void *dst,
				      size_t dst_capacity, void *workspace, size_t workspace_bytes,
				      uint64_t *total_host, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!dst || !total_host || (n && (!src || !workspace)) || (reinterpret_cast<uintptr_t>(dst) & 15u))
		return OROCH_E_ARGUMENT;
	if (dst_capacity < oroch_container_bound(dtype, n))
		return OROCH_E_CAPACITY;
	const size_t nblocks = (n + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
	uint8_t *base = static_cast<uint8_t *>(dst);
	oroch_container_header h;
	memset(&h, 0, sizeof h);
	memcpy(h.magic, CONTAINER_MAGIC, sizeof h.magic);
	h.dtype = uint32_t(dtype);
	h.block_values = OROCH_BLOCK_VALUES;
	h.nvalues = n;
	h.nblocks = nblocks;
	h.table_offset = sizeof(oroch_container_header);
	h.stream_offset = container_stream_offset(nblocks);
	hipStream_t st = static_cast<hipStream_t>(stream);
	// the pad between the table and the images
	int rc = int(hipMemsetAsync(base + h.stream_offset - 16, 0, 16, st));
	if (rc)
		return rc;
	uint64_t total = 0;
	rc = oroch_group_encode(dtype, src, n, base + h.stream_offset, dst_capacity - h.stream_offset,
				reinterpret_cast<uint64_t *>(base + h.table_offset), workspace, workspace_bytes,
				&total, stream);
	if (rc)
		return rc;
	h.stream_bytes = total;
	rc = int(hipMemcpyAsync(base, &h, sizeof h, hipMemcpyHostToDevice, st));
	if (rc == 0)
		rc = int(hipStreamSynchronize(st));
	*total_host = h.stream_offset + total;
	return rc;
}

extern "C" int oroch_container_check(const oroch_container_header *h, size_t container_bytes)
{
	if (!h || container_bytes < sizeof *h || memcmp(h->magic, CONTAINER_MAGIC, sizeof h->magic) != 0)
		return OROCH_E_ARGUMENT;
	if (h->dtype > 7u || h->block_values != OROCH_BLOCK_VALUES || h->reserved != 0)
		return OROCH_E_ARGUMENT;
	// no sum that could wrap: a count near 2^64 must not come out as few blocks
	const uint64_t nblocks = h->nvalues / OROCH_BLOCK_VALUES + (h->nvalues % OROCH_BLOCK_VALUES != 0);
	// the table alone needs 8 bytes per block inside the container
	if (h->nblocks != nblocks || nblocks > container_bytes / sizeof(uint64_t) || h->table_offset != sizeof *h ||
	    h->stream_offset != container_stream_offset(size_t(nblocks)) || (h->stream_bytes & 15u))
		return OROCH_E_ARGUMENT;
	if (h->stream_bytes > container_bytes || h->stream_offset > container_bytes - h->stream_bytes)
		return OROCH_E_ARGUMENT;
	return 0;
}

bool oroch_hip::offsets_table_sound(const uint64_t *table, size_t entries, uint64_t stream_bytes)
{
	uint64_t before = 0;
	for (size_t i = 0; i < entries; i++) {
		uint64_t x;
		memcpy(&x, reinterpret_cast<const uint8_t *>(table) + i * sizeof x, sizeof x); // any alignment
		if (x < before || x > stream_bytes || (x & 15u))
			return false;
		before = x;
	}
	return true;
}

extern "C" int oroch_container_decode(const void *src, const oroch_container_header *h,
				      size_t first_block, size_t nblocks, void *dst, void *stream)
{
	if (!h || !src || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	int rc = oroch_container_check(h, size_t(h->stream_offset + h->stream_bytes));
	if (rc)
		return rc;
	if (first_block > h->nblocks || nblocks > h->nblocks - first_block)
		return OROCH_E_ARGUMENT;
	if (nblocks == 0)
		return 0;
	if (!dst)
		return OROCH_E_ARGUMENT;
	const uint8_t *base = static_cast<const uint8_t *>(src);
	const uint64_t lo = uint64_t(first_block) * OROCH_BLOCK_VALUES;
	uint64_t hi = uint64_t(first_block + nblocks) * OROCH_BLOCK_VALUES;
	hi = hi < h->nvalues ? hi : h->nvalues;
	// the table's offsets are relative to the first image, which is what the
	// decoder adds them to; the table is device memory nobody has looked at,
	// so the decoder keeps every block inside the stream's length
	return oroch_hip::group_decode_within(int(h->dtype), base + h->stream_offset,
					      reinterpret_cast<const uint64_t *>(base + h->table_offset) + first_block,
					      size_t(hi - lo), dst, h->stream_bytes, stream);
}

extern "C" int oroch_container_encode_host(int dtype, const void *src_host, size_t n, void *dst_host,
					   size_t dst_capacity, uint64_t *total_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!dst_host || !total_host || (n && !src_host))
		return OROCH_E_ARGUMENT;
	const size_t bytes = n << (dtype >> 1);
	const size_t bound = oroch_container_bound(dtype, n);
	dev_buf in, out, ws;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(bound)) || (rc = ws.alloc(oroch_group_workspace(dtype, n))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	uint64_t total = 0;
	rc = oroch_container_encode(dtype, in.p, n, out.p, bound, ws.p, oroch_group_workspace(dtype, n), &total,
				    nullptr);
	if (rc)
		return rc;
	*total_host = total;
	if (total > dst_capacity)
		return OROCH_E_CAPACITY;
	return staged_copy(dst_host, out.p, size_t(total), false, nullptr);
}

extern "C" int oroch_container_decode_host(const void *src_host, size_t container_bytes,
					   size_t first_block, size_t nblocks, void *dst_host)
{
	if (!src_host || container_bytes < sizeof(oroch_container_header))
		return OROCH_E_ARGUMENT;
	oroch_container_header h;
	memcpy(&h, src_host, sizeof h);
	int rc = oroch_container_check(&h, container_bytes);
	if (rc)
		return rc;
	if (first_block > h.nblocks || nblocks > h.nblocks - first_block)
		return OROCH_E_ARGUMENT;
	if (nblocks == 0)
		return 0;
	if (!dst_host)
		return OROCH_E_ARGUMENT;
	// the table is at hand here: a damaged one is an error, not wrong values
	const uint8_t *table = static_cast<const uint8_t *>(src_host) + h.table_offset;
	uint64_t table0;
	memcpy(&table0, table, sizeof table0);
	if (table0 != 0 || !oroch_hip::offsets_table_sound(reinterpret_cast<const uint64_t *>(table) + first_block,
							   nblocks + 1, h.stream_bytes))
		return OROCH_E_ARGUMENT;
	const uint64_t lo = uint64_t(first_block) * OROCH_BLOCK_VALUES;
	uint64_t hi = uint64_t(first_block + nblocks) * OROCH_BLOCK_VALUES;
	hi = hi < h.nvalues ? hi : h.nvalues;
	const size_t out_bytes = size_t(hi - lo) << (h.dtype >> 1);
	const size_t total = size_t(h.stream_offset + h.stream_bytes);
	dev_buf in, out;
	if ((rc = in.alloc(total + 16)) || (rc = out.alloc(out_bytes)))
		return rc;
	if ((rc = staged_copy(in.p, src_host, total, true, nullptr)))
		return rc;
	rc = oroch_container_decode(in.p, &h, first_block, nblocks, out.p, nullptr);
	if (rc)
		return rc;
	return staged_copy(dst_host, out.p, out_bytes, false, nullptr);
}
