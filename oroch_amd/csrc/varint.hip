// varint.hip -- whole-sequence varints: varint_codec<T>::space / encode / decode,
// varint.h:124-147, with the default value codec (zigzag for signed T,
// zigzag.h:85-93).  7 payload bits per byte, low group first, top bit set on
// every byte but the last of a value (varint.h:74-83).
//
// Encode: per tile of 4096 values the byte count, a scan over the tiles, then
// every tile assembles its bytes in LDS and writes them out in one run.
// Decode: a value ends at each byte whose top bit is clear, so per tile of
// 4096 bytes the count of such bytes, a scan, then the thread that owns an end
// byte walks back over the (at most nine) bytes before it.
#include "common.hpp"
#include "scan.hpp"
#include "host_util.hpp"

namespace oroch_hip {

constexpr int VI_THREADS = 256;
constexpr int VI_VPT = 16;
constexpr int VI_TILE = VI_THREADS * VI_VPT; // values (encode) or bytes (decode) per tile
constexpr int VI_HALO = 16;                  // bytes before a decode tile, >= the longest value

template <typename E>
constexpr int varint_max_bytes()
{
	return (E::W + 6) / 7;
}

template <typename E>
__device__ __forceinline__ void load_tile_values(const typename E::S *src, size_t n, size_t base,
						 typename E::C (&u)[VI_VPT], uint32_t &valid)
{
	using C = typename E::C;
	if (base + VI_VPT <= n) {
		load16<E>(src + base, u);
		valid = VI_VPT;
	} else {
		valid = base < n ? uint32_t(n - base) : 0u;
#pragma unroll
		for (int s = 0; s < VI_VPT; s++)
			u[s] = uint32_t(s) < valid ? C(src[base + s]) : C(0);
	}
#pragma unroll
	for (int s = 0; s < VI_VPT; s++)
		u[s] = zz<E>(u[s]);
}

// Sum over the workgroup; the result is valid in every thread.
__device__ __forceinline__ uint32_t wg_sum(uint32_t x, uint32_t *lds)
{
	const uint32_t w = wave_sum(x);
	if (lane_id() == 0)
		lds[threadIdx.x / WAVE] = w;
	__syncthreads();
	uint32_t total = 0;
	for (int i = 0; i < VI_THREADS / WAVE; i++)
		total += lds[i];
	__syncthreads();
	return total;
}
// Exclusive prefix over the workgroup's threads; *total receives the sum.
__device__ __forceinline__ uint32_t wg_scan_excl(uint32_t x, uint32_t *lds, uint32_t *total)
{
	const int lane = lane_id(), wave = threadIdx.x / WAVE;
	const uint32_t incl = wave_scan_sum(x, lane);
	if (lane == WAVE - 1)
		lds[wave] = incl;
	__syncthreads();
	uint32_t before = 0, all = 0;
	for (int i = 0; i < VI_THREADS / WAVE; i++) {
		before += i < wave ? lds[i] : 0u;
		all += lds[i];
	}
	__syncthreads();
	*total = all;
	return before + incl - x;
}

// varint.h:124-131: bytes of every tile of 4096 values.
template <typename E>
__global__ __launch_bounds__(VI_THREADS) void k_varint_len(const typename E::S *src, size_t n,
							     uint32_t *tile_bytes)
{
	__shared__ uint32_t lds[VI_THREADS / WAVE];
	typename E::C u[VI_VPT];
	uint32_t valid;
	load_tile_values<E>(src, n, size_t(blockIdx.x) * VI_TILE + size_t(threadIdx.x) * VI_VPT, u, valid);
	uint32_t len = 0;
#pragma unroll
	for (int s = 0; s < VI_VPT; s++)
		len += uint32_t(s) < valid ? vsp(u[s]) : 0u;
	const uint32_t total = wg_sum(len, lds);
	if (threadIdx.x == 0)
		tile_bytes[blockIdx.x] = total;
}

// varint.h:133-139.
template <typename E>
__global__ __launch_bounds__(VI_THREADS) void k_varint_emit(const typename E::S *src, size_t n,
							      const uint64_t *tile_off, uint8_t *dst)
{
	__shared__ uint32_t lds[VI_THREADS / WAVE];
	__shared__ __attribute__((aligned(16))) uint8_t buf[VI_TILE * varint_max_bytes<E>() + 16];
	typename E::C u[VI_VPT];
	uint32_t valid;
	load_tile_values<E>(src, n, size_t(blockIdx.x) * VI_TILE + size_t(threadIdx.x) * VI_VPT, u, valid);
	uint32_t len = 0;
#pragma unroll
	for (int s = 0; s < VI_VPT; s++)
		len += uint32_t(s) < valid ? vsp(u[s]) : 0u;
	uint32_t total;
	uint32_t at = wg_scan_excl(len, lds, &total);
#pragma unroll
	for (int s = 0; s < VI_VPT; s++) {
		if (uint32_t(s) < valid) {
			uint64_t x = uint64_t(u[s]);
			while (x >= 0x80) {
				buf[at++] = uint8_t(x | 0x80);
				x >>= 7;
			}
			buf[at++] = uint8_t(x);
		}
	}
	__syncthreads();
	// out in one run: single bytes up to the first 4-byte boundary of the
	// destination, dwords after it, single bytes at the end
	uint8_t *out = dst + tile_off[blockIdx.x];
	const uint32_t head = uint32_t(-reinterpret_cast<uintptr_t>(out)) & 3u;
	const uint32_t lead = head < total ? head : total;
	if (threadIdx.x < lead)
		out[threadIdx.x] = buf[threadIdx.x];
	const uint32_t ndw = (total - lead) / 4u;
	for (uint32_t i = threadIdx.x; i < ndw; i += VI_THREADS) {
		const uint8_t *b = buf + lead + 4u * i;
		*reinterpret_cast<uint32_t *>(out + lead + 4u * i) =
			uint32_t(b[0]) | (uint32_t(b[1]) << 8) | (uint32_t(b[2]) << 16) | (uint32_t(b[3]) << 24);
	}
	const uint32_t done = lead + 4u * ndw;
	if (threadIdx.x < total - done)
		out[done + threadIdx.x] = buf[done + threadIdx.x];
}

// The 16 bytes of this thread's piece of a decode tile and, bit j, whether
// byte j ends a value (top bit clear, inside the stream).
__device__ __forceinline__ uint32_t end_flags(const uint8_t *src, size_t nbytes, size_t at, uint4 &w)
{
	w = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
	if (at + 16 <= nbytes) {
		w = *reinterpret_cast<const uint4 *>(src + at);
	} else if (at < nbytes) {
		uint32_t d[4] = {w.x, w.y, w.z, w.w};
		for (uint32_t j = 0; j < uint32_t(nbytes - at); j++)
			d[j >> 2] = (d[j >> 2] & ~(0xffu << (8 * (j & 3)))) | (uint32_t(src[at + j]) << (8 * (j & 3)));
		w = make_uint4(d[0], d[1], d[2], d[3]);
	}
	const uint32_t d[4] = {w.x, w.y, w.z, w.w};
	uint32_t flags = 0;
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const uint32_t t = ~d[k] & 0x80808080u;
		flags |= (((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u)) << (4 * k);
	}
	return flags;
}

template <typename E>
__global__ __launch_bounds__(VI_THREADS) void k_varint_count(const uint8_t *src, size_t nbytes,
							       uint32_t *tile_values)
{
	__shared__ uint32_t lds[VI_THREADS / WAVE];
	uint4 w;
	const uint32_t flags = end_flags(src, nbytes, size_t(blockIdx.x) * VI_TILE + size_t(threadIdx.x) * 16, w);
	const uint32_t total = wg_sum(uint32_t(__popc(flags)), lds);
	if (threadIdx.x == 0)
		tile_values[blockIdx.x] = total;
}

// varint.h:141-147.
template <typename E>
__global__ __launch_bounds__(VI_THREADS) void k_varint_take(const uint8_t *src, size_t nbytes, size_t n,
							      const uint64_t *tile_rank, typename E::S *dst,
							      uint64_t *consumed)
{
	using C = typename E::C;
	__shared__ uint32_t lds[VI_THREADS / WAVE];
	__shared__ __attribute__((aligned(16))) uint8_t buf[VI_HALO + VI_TILE];
	__shared__ C out[VI_TILE]; // at most one value per byte
	const size_t tile_at = size_t(blockIdx.x) * VI_TILE;
	uint4 w;
	uint32_t flags = end_flags(src, nbytes, tile_at + size_t(threadIdx.x) * 16, w);
	*reinterpret_cast<uint4 *>(buf + VI_HALO + threadIdx.x * 16) = w;
	if (threadIdx.x == 0) {
		// the bytes just before the tile (zero before the stream: a zero ends a value)
		uint4 h = make_uint4(0, 0, 0, 0);
		if (tile_at >= VI_HALO)
			h = *reinterpret_cast<const uint4 *>(src + tile_at - VI_HALO);
		*reinterpret_cast<uint4 *>(buf) = h;
	}
	uint32_t total;
	uint32_t r = wg_scan_excl(uint32_t(__popc(flags)), lds, &total);
	const uint64_t rank0 = tile_rank[blockIdx.x];
	// (wg_scan_excl has synchronised the workgroup: buf is complete)
	// The tile's values are put together in LDS, in order, and leave in
	// consecutive stores.
	while (flags) {
		const uint32_t p = VI_HALO + threadIdx.x * 16u + uint32_t(__builtin_ctz(flags));
		flags &= flags - 1u;
		// from the end byte back to the value's first byte, high group first
		uint64_t acc = buf[p];
		for (int k = 1; k < varint_max_bytes<E>(); k++) {
			const uint32_t b = buf[p - k];
			if (!(b & 0x80u))
				break;
			acc = (acc << 7) | (b & 0x7fu);
		}
		// where value number n - 1 ends is how many bytes the n values take
		if (consumed && rank0 + r + 1 == uint64_t(n))
			*consumed = uint64_t(tile_at) + (p - VI_HALO) + 1u;
		out[r++] = unzz<E>(C(acc) & E::MASK);
	}
	__syncthreads();
	for (uint32_t i = threadIdx.x; i < total; i += VI_THREADS) {
		if (rank0 + i < n)
			dst[rank0 + i] = typename E::S(out[i]);
	}
}

struct varint_ws
{
	uint32_t *counts;
	uint64_t *offs;      // ntiles + 1
	uint64_t *tile_sums;
	uint64_t *total;
	size_t bytes;
};

static varint_ws carve_varint(void *ws, size_t units)
{
	const size_t ntiles = (units + VI_TILE - 1) / VI_TILE + 1;
	auto up = [](size_t x) { return (x + 255) & ~size_t(255); };
	varint_ws w;
	char *p = static_cast<char *>(ws);
	size_t off = 0;
	w.counts = reinterpret_cast<uint32_t *>(p + off);
	off = up(off + (ntiles + 1) * sizeof(uint32_t));
	w.offs = reinterpret_cast<uint64_t *>(p + off);
	off = up(off + (ntiles + 2) * sizeof(uint64_t));
	w.tile_sums = reinterpret_cast<uint64_t *>(p + off);
	off = up(off + (scan_tiles(ntiles) + 2) * sizeof(uint64_t));
	w.total = reinterpret_cast<uint64_t *>(p + off);
	off = up(off + 256);
	w.bytes = off;
	return w;
}

} // namespace oroch_hip

using namespace oroch_hip;

extern "C" size_t oroch_varint_workspace(size_t units)
{
	return carve_varint(nullptr, units).bytes;
}

// Lengths and offsets of all tiles; the total ends up in w.offs[ntiles].
template <typename E>
static int varint_measure(const void *src, size_t n, const varint_ws &w, hipStream_t st)
{
	const size_t ntiles = (n + VI_TILE - 1) / VI_TILE;
	hipLaunchKernelGGL(k_varint_len<E>, dim3(unsigned(ntiles)), dim3(VI_THREADS), 0, st,
			   static_cast<const typename E::S *>(src), n, w.counts);
	return launch_scan<uint32_t, 0>(w.counts, ntiles, w.tile_sums, w.total, w.offs, st);
}

static int varint_check(int dtype, size_t n, const void *src, const void *workspace, size_t workspace_bytes,
			size_t units)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n && (!src || !workspace))
		return OROCH_E_ARGUMENT;
	if ((n + VI_TILE - 1) / VI_TILE > 0x7fffffffu)
		return OROCH_E_ARGUMENT;
	if (n && workspace_bytes < oroch_varint_workspace(units))
		return OROCH_E_CAPACITY;
	return 0;
}

extern "C" int oroch_varint_space(int dtype, const void *src, size_t n, void *workspace,
				  size_t workspace_bytes, uint64_t *space_host, void *stream)
{
	int rc = varint_check(dtype, n, src, workspace, workspace_bytes, n);
	if (rc)
		return rc;
	if (!space_host)
		return OROCH_E_ARGUMENT;
	*space_host = 0;
	if (n == 0)
		return 0;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const varint_ws w = carve_varint(workspace, n);
	const size_t ntiles = (n + VI_TILE - 1) / VI_TILE;
	rc = dispatch_dtype(dtype, [&](auto e) {
		profile_scope ps(OROCH_K_PACK, st);
		return varint_measure<decltype(e)>(src, n, w, st);
	});
	if (rc == 0)
		rc = int(hipMemcpyAsync(space_host, w.offs + ntiles, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
	if (rc == 0)
		rc = int(hipStreamSynchronize(st));
	return rc;
}

extern "C" int oroch_varint_encode(int dtype, const void *src, size_t n, void *dst, size_t dst_capacity,
				   void *workspace, size_t workspace_bytes, uint64_t *total_host,
				   void *stream)
{
	int rc = varint_check(dtype, n, src, workspace, workspace_bytes, n);
	if (rc)
		return rc;
	if (total_host)
		*total_host = 0;
	if (n == 0)
		return 0;
	const size_t worst = n * size_t(((8u << (dtype >> 1)) + 6u) / 7u);
	if (!dst || dst_capacity < worst)
		return dst ? OROCH_E_CAPACITY : OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const varint_ws w = carve_varint(workspace, n);
	const size_t ntiles = (n + VI_TILE - 1) / VI_TILE;
	rc = dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_PACK, st);
		int r = varint_measure<E>(src, n, w, st);
		if (r)
			return r;
		hipLaunchKernelGGL(k_varint_emit<E>, dim3(unsigned(ntiles)), dim3(VI_THREADS), 0, st,
				   static_cast<const typename E::S *>(src), n, w.offs, static_cast<uint8_t *>(dst));
		return int(hipGetLastError());
	});
	if (rc == 0 && total_host) {
		rc = int(hipMemcpyAsync(total_host, w.offs + ntiles, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
		if (rc == 0)
			rc = int(hipStreamSynchronize(st));
	}
	return rc;
}

// oroch_varint_decode that also reports, in *consumed (device, may be null;
// left alone when the bytes hold fewer than n values), the bytes the n values
// occupy.
int oroch_hip::varint_decode_counted(int dtype, const void *src, size_t nbytes, size_t n, void *dst,
				     void *workspace, size_t workspace_bytes, uint64_t *consumed, void *stream)
{
	int rc = varint_check(dtype, nbytes, src, workspace, workspace_bytes, nbytes);
	if (rc)
		return rc;
	if (n == 0)
		return 0;
	if (!dst || nbytes == 0 || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const varint_ws w = carve_varint(workspace, nbytes);
	const size_t ntiles = (nbytes + VI_TILE - 1) / VI_TILE;
	return dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_UNPACK, st);
		hipLaunchKernelGGL(k_varint_count<E>, dim3(unsigned(ntiles)), dim3(VI_THREADS), 0, st,
				   static_cast<const uint8_t *>(src), nbytes, w.counts);
		int r = launch_scan<uint32_t, 0>(w.counts, ntiles, w.tile_sums, w.total, w.offs, st);
		if (r)
			return r;
		hipLaunchKernelGGL(k_varint_take<E>, dim3(unsigned(ntiles)), dim3(VI_THREADS), 0, st,
				   static_cast<const uint8_t *>(src), nbytes, n, w.offs,
				   static_cast<typename E::S *>(dst), consumed);
		return int(hipGetLastError());
	});
}

extern "C" int oroch_varint_decode(int dtype, const void *src, size_t nbytes, size_t n, void *dst,
				   void *workspace, size_t workspace_bytes, void *stream)
{
	return varint_decode_counted(dtype, src, nbytes, n, dst, workspace, workspace_bytes, nullptr, stream);
}

// ---- host buffers ------------------------------------------------------------------
//
// The sequence calls of varint_codec<T> (varint.h:124-147) for values and bytes
// in host memory: staged to the device, coded there, staged back.  They allocate
// and wait; what they save is the CPU's loop over the values.

extern "C" int oroch_varint_space_host(int dtype, const void *src_host, size_t n, uint64_t *space_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!space_host || (n && !src_host))
		return OROCH_E_ARGUMENT;
	*space_host = 0;
	if (n == 0)
		return 0;
	const size_t bytes = n << (dtype >> 1);
	dev_buf in, ws;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = ws.alloc(oroch_varint_workspace(n))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	return oroch_varint_space(dtype, in.p, n, ws.p, oroch_varint_workspace(n), space_host, nullptr);
}

extern "C" int oroch_varint_encode_host(int dtype, const void *src_host, size_t n, void *dst_host,
					size_t dst_capacity, uint64_t *total_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n && (!src_host || !dst_host))
		return OROCH_E_ARGUMENT;
	if (total_host)
		*total_host = 0;
	if (n == 0)
		return 0;
	const size_t bytes = n << (dtype >> 1);
	const size_t worst = n * size_t(((8u << (dtype >> 1)) + 6u) / 7u);
	dev_buf in, out, ws;
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(worst)) || (rc = ws.alloc(oroch_varint_workspace(n))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, bytes, true, nullptr)))
		return rc;
	uint64_t total = 0;
	if ((rc = oroch_varint_encode(dtype, in.p, n, out.p, worst, ws.p, oroch_varint_workspace(n), &total, nullptr)))
		return rc;
	if (total > dst_capacity)
		return OROCH_E_CAPACITY;
	if (total_host)
		*total_host = total;
	return staged_copy(dst_host, out.p, size_t(total), false, nullptr);
}

extern "C" int oroch_varint_decode_host(int dtype, const void *src_host, size_t nbytes, size_t n, void *dst_host,
					uint64_t *consumed_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (n && (!src_host || !dst_host || !nbytes))
		return OROCH_E_ARGUMENT;
	if (consumed_host)
		*consumed_host = 0;
	if (n == 0)
		return 0;
	const size_t bytes = n << (dtype >> 1);
	dev_buf in, out, ws, used;
	int rc;
	if ((rc = in.alloc(nbytes)) || (rc = out.alloc(bytes)) || (rc = ws.alloc(oroch_varint_workspace(nbytes))) ||
	    (rc = used.alloc(sizeof(uint64_t))))
		return rc;
	if ((rc = int(hipMemsetAsync(used.p, 0xff, sizeof(uint64_t), nullptr))))
		return rc;
	if ((rc = staged_copy(in.p, src_host, nbytes, true, nullptr)))
		return rc;
	if ((rc = varint_decode_counted(dtype, in.p, nbytes, n, out.p, ws.p, oroch_varint_workspace(nbytes),
					static_cast<uint64_t *>(used.p), nullptr)))
		return rc;
	uint64_t consumed = 0;
	if ((rc = int(hipMemcpy(&consumed, used.p, sizeof consumed, hipMemcpyDeviceToHost))))
		return rc;
	// (left alone by the kernel when the bytes hold fewer than n values)
	if (consumed == ~uint64_t(0))
		return OROCH_E_ARGUMENT;
	if (consumed_host)
		*consumed_host = consumed;
	return staged_copy(dst_host, out.p, bytes, false, nullptr);
}
