// pipeline.hip -- host buffer in, host buffer out, with the link and the
// kernels working at the same time.
//
// An array that starts and ends in host memory (a file, a network buffer)
// is cut into chunks of whole blocks.  Blocks are independent
// (integer_group.h:44-76 sees only its own values), so chunk k + 1 can be on
// its way to the device while chunk k is encoded and the stream of chunk k - 1
// travels back: three streams (in, run, out), two slots of device buffers, and
// events between them.  All device and pinned memory is allocated when the pipe
// is made and kept; a call allocates nothing.
//
// What a chunk's kernels need from the chunk before it is one number, the
// length of the stream so far, and that only on the host side: it places the
// chunk's bytes in the caller's buffer and is added to the chunk's block
// offsets while they are copied out.
#include <cstring>
#include <mutex>
#include <new>

#include "common.hpp"
#include "host_util.hpp"

namespace oroch_hip {

namespace {

struct pipe_slot
{
	void *d_vals = nullptr;    // values of a chunk
	void *d_stream = nullptr;  // block images of a chunk
	uint64_t *d_offs = nullptr;
	void *d_ws = nullptr;
	void *h_vals = nullptr;    // pinned staging for callers whose buffers are not
	void *h_stream = nullptr;
	uint64_t *h_offs = nullptr;
	uint64_t *h_total = nullptr;
	hipEvent_t in_done = nullptr, run_done = nullptr, out_done = nullptr;
	// what the slot's output stage still owes the caller
	bool owing = false;
	void *to = nullptr;
	const void *from = nullptr;
	size_t bytes = 0;
	uint64_t *offs_to = nullptr;
	size_t offs_count = 0;
	uint64_t offs_add = 0;
};

} // namespace

} // namespace oroch_hip

using namespace oroch_hip;

struct oroch_pipe
{
	std::mutex lock;
	int device = 0;
	size_t chunk_bytes = 0;   // values per chunk, in bytes
	size_t stream_bytes = 0;  // room for a chunk's images
	size_t max_blocks = 0;    // blocks per chunk of the narrowest element type
	size_t ws_bytes = 0;
	hipStream_t s_in = nullptr, s_run = nullptr, s_out = nullptr;
	pipe_slot slot[2];
};

namespace {

void pipe_free(oroch_pipe *p)
{
	for (pipe_slot &s : p->slot) {
		if (s.d_vals) (void) hipFree(s.d_vals);
		if (s.d_stream) (void) hipFree(s.d_stream);
		if (s.d_offs) (void) hipFree(s.d_offs);
		if (s.d_ws) (void) hipFree(s.d_ws);
		if (s.h_vals) (void) hipHostFree(s.h_vals);
		if (s.h_stream) (void) hipHostFree(s.h_stream);
		if (s.h_offs) (void) hipHostFree(s.h_offs);
		if (s.h_total) (void) hipHostFree(s.h_total);
		if (s.in_done) (void) hipEventDestroy(s.in_done);
		if (s.run_done) (void) hipEventDestroy(s.run_done);
		if (s.out_done) (void) hipEventDestroy(s.out_done);
	}
	if (p->s_in) (void) hipStreamDestroy(p->s_in);
	if (p->s_run) (void) hipStreamDestroy(p->s_run);
	if (p->s_out) (void) hipStreamDestroy(p->s_out);
	delete p;
}

// Hands the caller what the slot's last output copy brought (staged bytes,
// rebased offsets) once that copy has arrived.
int settle(pipe_slot &s)
{
	if (!s.owing)
		return 0;
	s.owing = false;
	int rc = int(hipEventSynchronize(s.out_done));
	if (rc)
		return rc;
	if (s.to && s.bytes)
		parallel_memcpy(s.to, s.from, s.bytes);
	for (size_t i = 0; i < s.offs_count; i++)
		s.offs_to[i] = s.h_offs[i] + s.offs_add;
	return 0;
}

} // namespace

extern "C" int oroch_pipe_create(size_t chunk_bytes, oroch_pipe **out)
{
	if (!out)
		return OROCH_E_ARGUMENT;
	*out = nullptr;
	if (chunk_bytes == 0)
		chunk_bytes = size_t(64) << 20;
	// whole blocks of every element type
	chunk_bytes = (chunk_bytes + OROCH_BLOCK_VALUES * 8 - 1) / (OROCH_BLOCK_VALUES * 8) * (OROCH_BLOCK_VALUES * 8);
	oroch_pipe *p = new (std::nothrow) oroch_pipe;
	if (!p)
		return int(hipErrorOutOfMemory);
	int rc = int(hipGetDevice(&p->device));
	p->chunk_bytes = chunk_bytes;
	for (int dtype = 0; dtype < 8 && rc == 0; dtype++) {
		const size_t n = chunk_bytes >> (dtype >> 1);
		const size_t bound = oroch_group_bound(dtype, n) + 16, ws = oroch_group_workspace(dtype, n);
		p->stream_bytes = bound > p->stream_bytes ? bound : p->stream_bytes;
		p->ws_bytes = ws > p->ws_bytes ? ws : p->ws_bytes;
	}
	p->max_blocks = chunk_bytes / OROCH_BLOCK_VALUES;
	if (rc == 0)
		rc = int(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
	if (rc == 0)
		rc = int(hipStreamCreateWithFlags(&p->s_run, hipStreamNonBlocking));
	if (rc == 0)
		rc = int(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
	for (pipe_slot &s : p->slot) {
		const size_t offs_bytes = (p->max_blocks + 1) * sizeof(uint64_t);
		if (rc == 0) rc = int(hipMalloc(&s.d_vals, chunk_bytes));
		if (rc == 0) rc = int(hipMalloc(&s.d_stream, p->stream_bytes));
		if (rc == 0) rc = int(hipMalloc(reinterpret_cast<void **>(&s.d_offs), offs_bytes));
		if (rc == 0) rc = int(hipMalloc(&s.d_ws, p->ws_bytes));
		if (rc == 0) rc = int(hipHostMalloc(&s.h_vals, chunk_bytes, hipHostMallocDefault));
		if (rc == 0) rc = int(hipHostMalloc(&s.h_stream, p->stream_bytes, hipHostMallocDefault));
		if (rc == 0) rc = int(hipHostMalloc(reinterpret_cast<void **>(&s.h_offs), offs_bytes, hipHostMallocDefault));
		if (rc == 0) rc = int(hipHostMalloc(reinterpret_cast<void **>(&s.h_total), 64, hipHostMallocDefault));
		if (rc == 0) rc = int(hipEventCreateWithFlags(&s.in_done, hipEventDisableTiming));
		if (rc == 0) rc = int(hipEventCreateWithFlags(&s.run_done, hipEventDisableTiming));
		if (rc == 0) rc = int(hipEventCreateWithFlags(&s.out_done, hipEventDisableTiming));
	}
	if (rc) {
		pipe_free(p);
		return rc;
	}
	*out = p;
	return 0;
}

extern "C" void oroch_pipe_destroy(oroch_pipe *p)
{
	if (!p)
		return;
	// whatever the pipe still has in flight (the current device may be another)
	if (p->s_in) (void) hipStreamSynchronize(p->s_in);
	if (p->s_run) (void) hipStreamSynchronize(p->s_run);
	if (p->s_out) (void) hipStreamSynchronize(p->s_out);
	pipe_free(p);
}

namespace {

// A pipe's buffers and streams live on the device it was made on; the kernels
// are launched on the current one.
bool on_pipes_device(const oroch_pipe *p)
{
	int device = -1;
	return hipGetDevice(&device) == hipSuccess && device == p->device;
}

} // namespace

extern "C" int oroch_pipe_encode(oroch_pipe *p, int dtype, const void *src_host, size_t n, void *dst_host,
				 size_t dst_capacity, uint64_t *offsets_host, uint64_t *total_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!p || !offsets_host || !total_host || (n && (!src_host || !dst_host)) || !on_pipes_device(p))
		return OROCH_E_ARGUMENT;
	std::lock_guard<std::mutex> guard(p->lock);
	no_profile_scope quiet; // events of the recorder belong to the caller's stream
	const size_t w = size_t(1) << (dtype >> 1);
	const size_t per = p->chunk_bytes / w; // values per chunk, whole blocks
	const size_t nchunks = (n + per - 1) / per;
	const bool src_pinned = n && is_pinned(src_host), dst_pinned = n && is_pinned(dst_host);
	const uint8_t *src = static_cast<const uint8_t *>(src_host);
	uint8_t *dst = static_cast<uint8_t *>(dst_host);
	*total_host = 0;
	offsets_host[0] = 0;
	uint64_t base = 0;
	bool fits = true;
	int rc = 0;
	for (pipe_slot &s : p->slot)
		s.owing = false;
	// stage k of the loop: chunk k goes in and is encoded, chunk k - 1 comes out
	for (size_t k = 0; k <= nchunks && rc == 0; k++) {
		if (k < nchunks) {
			pipe_slot &s = p->slot[k & 1];
			const size_t lo = k * per, cnt = n - lo < per ? n - lo : per;
			// the slot's buffers are free again: its kernels of two chunks ago
			// have read d_vals, its output copy has left d_stream and the staging
			if ((rc = settle(s)))
				break;
			if (k >= 2 && (rc = int(hipStreamWaitEvent(p->s_in, s.run_done, 0))))
				break;
			const void *from = src + lo * w;
			if (!src_pinned) {
				if (k >= 2 && (rc = int(hipEventSynchronize(s.in_done))))
					break;
				parallel_memcpy(s.h_vals, from, cnt * w);
				from = s.h_vals;
			}
			if ((rc = int(hipMemcpyAsync(s.d_vals, from, cnt * w, hipMemcpyHostToDevice, p->s_in))) ||
			    (rc = int(hipEventRecord(s.in_done, p->s_in))) ||
			    (rc = int(hipStreamWaitEvent(p->s_run, s.in_done, 0))) ||
			    (k >= 2 && (rc = int(hipStreamWaitEvent(p->s_run, s.out_done, 0)))))
				break;
			const size_t nb = (cnt + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
			if ((rc = oroch_group_encode(dtype, s.d_vals, cnt, s.d_stream, p->stream_bytes, s.d_offs, s.d_ws,
						     p->ws_bytes, nullptr, p->s_run)) ||
			    (rc = int(hipMemcpyAsync(s.h_total, s.d_offs + nb, sizeof(uint64_t), hipMemcpyDeviceToHost,
						     p->s_run))) ||
			    (rc = int(hipEventRecord(s.run_done, p->s_run))))
				break;
		}
		if (k >= 1) {
			pipe_slot &s = p->slot[(k - 1) & 1];
			const size_t lo = (k - 1) * per, cnt = n - lo < per ? n - lo : per;
			const size_t nb = (cnt + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
			if ((rc = int(hipEventSynchronize(s.run_done))))
				break;
			const uint64_t total = *s.h_total;
			fits = fits && base + total <= dst_capacity;
			if ((rc = int(hipStreamWaitEvent(p->s_out, s.run_done, 0))))
				break;
			s.to = nullptr;
			s.bytes = 0;
			if (fits && total) {
				void *to = dst + base;
				if (!dst_pinned) {
					s.to = to;
					s.from = s.h_stream;
					s.bytes = size_t(total);
					to = s.h_stream;
				}
				if ((rc = int(hipMemcpyAsync(to, s.d_stream, size_t(total), hipMemcpyDeviceToHost, p->s_out))))
					break;
			}
			// offsets[1 ..] of the chunk, to be rebased onto the whole stream
			if ((rc = int(hipMemcpyAsync(s.h_offs, s.d_offs + 1, nb * sizeof(uint64_t), hipMemcpyDeviceToHost,
						     p->s_out))) ||
			    (rc = int(hipEventRecord(s.out_done, p->s_out))))
				break;
			s.offs_to = offsets_host + lo / OROCH_BLOCK_VALUES + 1;
			s.offs_count = nb;
			s.offs_add = base;
			s.owing = true;
			base += total;
		}
	}
	for (pipe_slot &s : p->slot) {
		const int r = settle(s);
		rc = rc ? rc : r;
	}
	if (rc) {
		// leave nothing of this call in flight
		(void) hipStreamSynchronize(p->s_in);
		(void) hipStreamSynchronize(p->s_run);
		(void) hipStreamSynchronize(p->s_out);
		return rc;
	}
	*total_host = base;
	return fits ? 0 : OROCH_E_CAPACITY;
}

extern "C" int oroch_pipe_decode(oroch_pipe *p, int dtype, const void *src_host, const uint64_t *offsets_host,
				 size_t n, void *dst_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n == 0)
		return 0;
	if (!p || !src_host || !offsets_host || !dst_host || !on_pipes_device(p))
		return OROCH_E_ARGUMENT;
	std::lock_guard<std::mutex> guard(p->lock);
	no_profile_scope quiet;
	const size_t w = size_t(1) << (dtype >> 1);
	const size_t per = p->chunk_bytes / w;
	const size_t nchunks = (n + per - 1) / per;
	const size_t nblocks = (n + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
	const size_t blocks_per = per / OROCH_BLOCK_VALUES;
	// the table is the caller's: every chunk's images must lie inside a slot
	if (!offsets_table_sound(offsets_host, nblocks + 1, offsets_host[nblocks]))
		return OROCH_E_ARGUMENT;
	const bool src_pinned = is_pinned(src_host), dst_pinned = is_pinned(dst_host);
	const uint8_t *src = static_cast<const uint8_t *>(src_host);
	uint8_t *dst = static_cast<uint8_t *>(dst_host);
	int rc = 0;
	for (pipe_slot &s : p->slot)
		s.owing = false;
	for (size_t k = 0; k <= nchunks && rc == 0; k++) {
		if (k < nchunks) {
			pipe_slot &s = p->slot[k & 1];
			const size_t lo = k * per, cnt = n - lo < per ? n - lo : per;
			const size_t b0 = k * blocks_per, nb = (cnt + OROCH_BLOCK_VALUES - 1) / OROCH_BLOCK_VALUES;
			const uint64_t first = offsets_host[b0], bytes = offsets_host[b0 + nb] - first;
			if (bytes + 16 > p->stream_bytes) {
				rc = OROCH_E_ARGUMENT;
				break;
			}
			if ((rc = settle(s)))
				break;
			if (k >= 2 && ((rc = int(hipStreamWaitEvent(p->s_in, s.run_done, 0))) ||
				       (rc = int(hipEventSynchronize(s.in_done)))))
				break;
			// offsets relative to the chunk's first image
			for (size_t i = 0; i <= nb; i++)
				s.h_offs[i] = offsets_host[b0 + i] - first;
			const void *from = src + first;
			if (!src_pinned && bytes) {
				parallel_memcpy(s.h_stream, from, size_t(bytes));
				from = s.h_stream;
			}
			if ((bytes && (rc = int(hipMemcpyAsync(s.d_stream, from, size_t(bytes), hipMemcpyHostToDevice, p->s_in)))) ||
			    (rc = int(hipMemcpyAsync(s.d_offs, s.h_offs, (nb + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
						     p->s_in))) ||
			    (rc = int(hipEventRecord(s.in_done, p->s_in))) ||
			    (rc = int(hipStreamWaitEvent(p->s_run, s.in_done, 0))) ||
			    (k >= 2 && (rc = int(hipStreamWaitEvent(p->s_run, s.out_done, 0)))))
				break;
			if ((rc = group_decode_within(dtype, s.d_stream, s.d_offs, cnt, s.d_vals, bytes, p->s_run)) ||
			    (rc = int(hipEventRecord(s.run_done, p->s_run))))
				break;
		}
		if (k >= 1) {
			pipe_slot &s = p->slot[(k - 1) & 1];
			const size_t lo = (k - 1) * per, cnt = n - lo < per ? n - lo : per;
			if ((rc = int(hipStreamWaitEvent(p->s_out, s.run_done, 0))))
				break;
			void *to = dst + lo * w;
			s.to = nullptr;
			s.bytes = 0;
			s.offs_count = 0;
			if (!dst_pinned) {
				s.to = to;
				s.from = s.h_vals;
				s.bytes = cnt * w;
				to = s.h_vals;
			}
			if ((rc = int(hipMemcpyAsync(to, s.d_vals, cnt * w, hipMemcpyDeviceToHost, p->s_out))) ||
			    (rc = int(hipEventRecord(s.out_done, p->s_out))))
				break;
			s.owing = true;
		}
	}
	for (pipe_slot &s : p->slot) {
		const int r = settle(s);
		rc = rc ? rc : r;
	}
	if (rc) {
		(void) hipStreamSynchronize(p->s_in);
		(void) hipStreamSynchronize(p->s_run);
		(void) hipStreamSynchronize(p->s_out);
	}
	return rc;
}

// The pipes behind oroch_group_encode_host / oroch_group_decode_host: one per
// device, made on first use, sized for the request (chunks of 1 MiB at least,
// 64 MiB at most; a larger request later makes a larger pipe in its place) and
// kept until oroch_hip_release_host_buffers().  A call holds the device's lock
// from the look-up to its end, so a pipe is never replaced or freed under a
// caller.
namespace {
constexpr int MAX_DEVICES = 64;
std::mutex default_locks[MAX_DEVICES];
oroch_pipe *default_pipes[MAX_DEVICES] = {};
} // namespace

int oroch_hip::with_default_pipe(size_t bytes, const std::function<int(oroch_pipe *)> &fn)
{
	int device = 0;
	int rc = int(hipGetDevice(&device));
	if (rc)
		return rc;
	if (device < 0 || device >= MAX_DEVICES)
		return OROCH_E_ARGUMENT;
	size_t chunk = size_t(1) << 20;
	while (chunk < bytes && chunk < (size_t(64) << 20))
		chunk <<= 1;
	std::lock_guard<std::mutex> guard(default_locks[device]);
	oroch_pipe *&p = default_pipes[device];
	if (p && p->chunk_bytes < chunk) {
		oroch_pipe_destroy(p);
		p = nullptr;
	}
	if (!p && (rc = oroch_pipe_create(chunk, &p)))
		return rc;
	return fn(p);
}

extern "C" void oroch_hip_release_host_buffers(void)
{
	for (int device = 0; device < MAX_DEVICES; device++) {
		std::lock_guard<std::mutex> guard(default_locks[device]);
		if (default_pipes[device]) {
			oroch_pipe_destroy(default_pipes[device]);
			default_pipes[device] = nullptr;
		}
	}
}
