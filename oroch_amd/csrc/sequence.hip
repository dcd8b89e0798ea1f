// sequence.hip -- one group of any length: selection, encode and decode of a
// whole sequence the way integer_codec<T> treats the range it is given
// (integer_codec.h:234-402), for the callers that use one integer_group for
// more values than a block holds.
//
// Selection needs three kinds of facts about the sequence.  Minimum, maximum
// and the varint length of every value come from one pass of all workgroups
// (wave reductions, one atomic per workgroup and quantity); the histogram of
// bit lengths of (v - min) and the varint length of (v - min) from a second
// (LDS histogram per workgroup, flushed by atomics).  The third, the width
// and varint length of the gaps between the values that do not fit a
// candidate width, is a walk over the whole sequence per candidate in the
// reference (integer_codec.h:338-351); here the candidate is packed by the
// whole-sequence bitpfr encoder (pack.hip), whose gap list is then reduced.
// The candidate loop itself runs on the host on 65 numbers, in the
// reference's order.
//
// Encode and decode put the whole-sequence kernels of pack.hip and varint.hip
// together as the descriptor says; list parts that start at odd addresses
// are staged through aligned scratch.
#include <cstring>

#include "common.hpp"
#include "host_util.hpp"

namespace oroch_hip {

constexpr int SQ_THREADS = 256;

struct seq_facts
{
	unsigned long long min_key;  // minimum, sign bit flipped for signed types
	unsigned long long max_key;
	unsigned long long vispace;  // sum of varint lengths of the zigzag images
	unsigned long long vfspace;  // sum of varint lengths of v - min
	unsigned long long gap_space; // sum of varint lengths of a gap list
	unsigned long long gap_bits;  // widest gap of that list
	unsigned long long consumed[2]; // bytes of a varint list (decode)
	unsigned long long hist[65];
};

__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long x)
{
	return wave_reduce<uint64_t>(x, 0, [](uint64_t a, uint64_t b) { return a + b; });
}

// integer_stats.h:48-53, 88-95, and the space of the plain varint encoding
// (integer_codec.h:466-478).
template <typename E>
__global__ __launch_bounds__(SQ_THREADS) void k_seq_range(const typename E::S *__restrict__ src, size_t n,
							   seq_facts *f)
{
	using C = typename E::C;
	uint64_t lo = ~uint64_t(0), hi = 0, space = 0;
	for (size_t i = size_t(blockIdx.x) * SQ_THREADS + threadIdx.x; i < n; i += size_t(gridDim.x) * SQ_THREADS) {
		const C v = C(src[i]) & E::MASK;
		const uint64_t key = uint64_t(v ^ E::FLIP);
		lo = key < lo ? key : lo;
		hi = key > hi ? key : hi;
		space += vsp(zz<E>(v));
	}
	lo = wave_reduce<uint64_t>(lo, ~uint64_t(0), [](uint64_t a, uint64_t b) { return a < b ? a : b; });
	hi = wave_reduce<uint64_t>(hi, 0, [](uint64_t a, uint64_t b) { return a > b ? a : b; });
	space = wave_sum64(space);
	if (lane_id() == 0) {
		atomicMin(&f->min_key, (unsigned long long) lo);
		atomicMax(&f->max_key, (unsigned long long) hi);
		atomicAdd(&f->vispace, (unsigned long long) space);
	}
}

// integer_stats.h:74-80, 97-102, and the space of varint with an origin
// (integer_codec.h:480-494).
template <typename E>
__global__ __launch_bounds__(SQ_THREADS) void k_seq_hist(const typename E::S *__restrict__ src, size_t n,
							  typename E::C mn, seq_facts *f)
{
	using C = typename E::C;
	__shared__ unsigned int h[65];
	for (int i = threadIdx.x; i < 65; i += SQ_THREADS)
		h[i] = 0;
	__syncthreads();
	uint64_t space = 0;
	// a workgroup sees fewer than 2^32 values: n < 2^40 with the grid below
	for (size_t i = size_t(blockIdx.x) * SQ_THREADS + threadIdx.x; i < n; i += size_t(gridDim.x) * SQ_THREADS) {
		const C u = (C(src[i]) - mn) & E::MASK;
		atomicAdd(&h[bitlen(u)], 1u);
		space += vsp(u);
	}
	space = wave_sum64(space);
	if (lane_id() == 0)
		atomicAdd(&f->vfspace, (unsigned long long) space);
	__syncthreads();
	for (int i = threadIdx.x; i < 65; i += SQ_THREADS)
		if (h[i])
			atomicAdd(&f->hist[i], (unsigned long long) h[i]);
}

// integer_codec.h:338-351 on the gap list the bitpfr encoder has written:
// the widest gap and the varint length of all of them.
__global__ __launch_bounds__(SQ_THREADS) void k_gap_facts(const uint64_t *__restrict__ gaps, size_t k, seq_facts *f)
{
	uint64_t space = 0, bits = 0;
	for (size_t i = size_t(blockIdx.x) * SQ_THREADS + threadIdx.x; i < k; i += size_t(gridDim.x) * SQ_THREADS) {
		const uint64_t g = gaps[i];
		const uint64_t b = bitlen(g);
		bits = b > bits ? b : bits;
		space += vsp(g);
	}
	space = wave_sum64(space);
	bits = wave_reduce<uint64_t>(bits, 0, [](uint64_t a, uint64_t b) { return a > b ? a : b; });
	if (lane_id() == 0) {
		atomicAdd(&f->gap_space, (unsigned long long) space);
		atomicMax(&f->gap_bits, (unsigned long long) bits);
	}
}

// out[i] = in[i] - origin (SUB) or in[i] + origin, modulo 2^W (origin.h:46-54).
template <typename E, bool SUB>
__global__ __launch_bounds__(SQ_THREADS) void k_seq_shift(const typename E::S *in, size_t n, typename E::C origin,
							   typename E::S *out)
{
	using C = typename E::C;
	for (size_t i = size_t(blockIdx.x) * SQ_THREADS + threadIdx.x; i < n; i += size_t(gridDim.x) * SQ_THREADS) {
		const C v = C(in[i]);
		out[i] = typename E::S((SUB ? v - origin : v + origin) & E::MASK);
	}
}

// naught.h:42-47.
template <typename E>
__global__ __launch_bounds__(SQ_THREADS) void k_seq_fill(typename E::S *out, size_t n, typename E::C value)
{
	for (size_t i = size_t(blockIdx.x) * SQ_THREADS + threadIdx.x; i < n; i += size_t(gridDim.x) * SQ_THREADS)
		out[i] = typename E::S(value);
}

// unsigned_t values held in 64 bits lose what is above the element's width
// (the reference decodes outlier values into unsigned_t).
__global__ __launch_bounds__(SQ_THREADS) void k_mask64(uint64_t *x, size_t k, uint64_t mask)
{
	for (size_t i = size_t(blockIdx.x) * SQ_THREADS + threadIdx.x; i < k; i += size_t(gridDim.x) * SQ_THREADS)
		x[i] &= mask;
}

static unsigned seq_grid(size_t n)
{
	const size_t wgs = (n + SQ_THREADS * 8 - 1) / (SQ_THREADS * 8);
	return unsigned(wgs < 1 ? 1 : wgs < 2048 ? wgs : 2048);
}

static unsigned width_of(int dtype)
{
	return 8u << (dtype >> 1);
}

static uint64_t mask_of(int dtype)
{
	const unsigned W = width_of(dtype);
	return W == 64 ? ~uint64_t(0) : (uint64_t(1) << W) - 1;
}

static unsigned used_bits(uint64_t u)
{
	return u ? 64u - unsigned(__builtin_clzll(u)) : 0u;
}

// zigzag.h:47-50 on a raw pattern; identity for unsigned types.
static uint64_t zigzag_of(int dtype, uint64_t raw)
{
	if (dtype & 1)
		return raw;
	const unsigned W = width_of(dtype);
	const uint64_t mask = mask_of(dtype);
	return ((raw << 1) ^ (((raw >> (W - 1)) & 1) ? mask : 0)) & mask;
}

static uint64_t unzigzag_of(int dtype, uint64_t u)
{
	if (dtype & 1)
		return u;
	return ((u >> 1) ^ (uint64_t(0) - (u & 1))) & mask_of(dtype);
}

// varint.h:56-59, 66-72.
static uint64_t varint_bytes(uint64_t u)
{
	return u ? (used_bits(u) + 6) / 7 : 1;
}

static uint64_t pck_bytes(uint64_t n, unsigned nbits)
{
	return uint64_t(oroch_bitpck_space(size_t(n), nbits));
}

static int fetch_facts(seq_facts *host, const seq_facts *dev, hipStream_t st)
{
	int rc = int(hipMemcpyAsync(host, dev, sizeof *host, hipMemcpyDeviceToHost, st));
	return rc ? rc : int(hipStreamSynchronize(st));
}

// The candidate of integer_codec.h:405-420: strictly smaller replaces.
struct choice
{
	uint8_t encoding;
	uint64_t dataspace, metaspace; // metaspace without the encoding byte
	uint64_t origin;
	unsigned nbits;
	void offer(uint8_t e, uint64_t meta, uint64_t data, uint64_t org, unsigned nb)
	{
		if (data + meta < dataspace + metaspace) {
			encoding = e;
			dataspace = data;
			metaspace = meta;
			origin = org;
			nbits = nb;
		}
	}
};

} // namespace oroch_hip

using namespace oroch_hip;

extern "C" size_t oroch_sequence_bound(int dtype, size_t n)
{
	if (dtype < 0 || dtype > 7)
		return 0;
	// the selected encoding never takes more than the raw copy
	return ((n << (dtype >> 1)) + 15) & ~size_t(15);
}

extern "C" int oroch_sequence_select(int dtype, const void *src, size_t n, oroch_sequence_desc *d, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || (n && !src) || n >= (size_t(1) << 40))
		return OROCH_E_ARGUMENT;
	std::memset(d, 0, sizeof *d);
	// fresh descriptors, integer_codec.h:83-90
	d->encoding = OROCH_NORMAL;
	d->index_encoding = OROCH_NORMAL;
	d->value_encoding = OROCH_NORMAL;
	d->metaspace = 1;
	d->nvalues = n;
	if (n == 0)
		return 0; // integer_codec.h:247-253
	hipStream_t st = static_cast<hipStream_t>(stream);
	const unsigned W = width_of(dtype);
	const uint64_t mask = mask_of(dtype);
	const uint64_t flip = (dtype & 1) ? 0 : uint64_t(1) << (W - 1);
	const size_t w = size_t(1) << (dtype >> 1);

	dev_buf facts_buf;
	int rc = facts_buf.alloc(sizeof(seq_facts));
	if (rc)
		return rc;
	seq_facts *facts = static_cast<seq_facts *>(facts_buf.p), h;
	std::memset(&h, 0, sizeof h);
	h.min_key = ~0ull;
	if ((rc = int(hipMemcpyAsync(facts, &h, sizeof h, hipMemcpyHostToDevice, st))))
		return rc;
	rc = dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_SELECT, st);
		hipLaunchKernelGGL(k_seq_range<E>, dim3(seq_grid(n)), dim3(SQ_THREADS), 0, st,
				   static_cast<const typename E::S *>(src), n, facts);
		return int(hipGetLastError());
	});
	if (rc || (rc = fetch_facts(&h, facts, st)))
		return rc;
	const uint64_t mn = (h.min_key ^ flip) & mask, mx = (h.max_key ^ flip) & mask;
	const uint64_t zmn = zigzag_of(dtype, mn);
	// integer_codec.h:255-264
	if (mn == mx) {
		d->encoding = OROCH_NAUGHT;
		d->origin = mn;
		d->metaspace = uint32_t(1 + varint_bytes(zmn));
		return 0;
	}
	rc = dispatch_dtype(dtype, [&](auto e) {
		using E = decltype(e);
		profile_scope ps(OROCH_K_SELECT, st);
		hipLaunchKernelGGL(k_seq_hist<E>, dim3(seq_grid(n)), dim3(SQ_THREADS), 0, st,
				   static_cast<const typename E::S *>(src), n, typename E::C(mn), facts);
		return int(hipGetLastError());
	});
	if (rc || (rc = fetch_facts(&h, facts, st)))
		return rc;

	// select_basic, integer_codec.h:422-496
	choice c = {OROCH_NORMAL, uint64_t(n) * w, 0, 0, 0};
	{
		uint64_t umax = mx;
		if (!(dtype & 1)) {
			const uint64_t a = zmn, b = zigzag_of(dtype, mx);
			umax = a > b ? a : b;
		}
		unsigned nb = used_bits(umax);
		c.offer(OROCH_BITPCK, 1, pck_bytes(n, nb), 0, nb);
		nb = used_bits((mx - mn) & mask);
		c.offer(OROCH_BITFOR, 1 + varint_bytes(zmn), pck_bytes(n, nb), mn, nb);
		c.offer(OROCH_VARINT, 0, h.vispace, 0, 0);
		c.offer(OROCH_VARFOR, varint_bytes(zmn), h.vfspace, mn, 0);
	}

	// integer_codec.h:271-383
	if (n >= 5) {
		const uint64_t basic_meta = 1 + varint_bytes(zmn);
		const unsigned nbits_max = used_bits((mx - mn) & mask);
		uint64_t noutliers = uint64_t(n) - h.hist[0];
		dev_buf packed, idx, val, ws;
		size_t packed_cap = 0, list_cap = 0;
		const size_t wsb = oroch_bitpfr_workspace(n);
		for (unsigned nbits = 1; nbits < nbits_max; nbits++) {
			if (h.hist[nbits] == 0)
				continue;
			noutliers -= h.hist[nbits];
			const uint64_t basic_data = pck_bytes(n, nbits);
			const uint64_t extra_meta = 2 + varint_bytes(noutliers);
			const uint64_t valpck = pck_bytes(noutliers, nbits_max - nbits);
			uint64_t valvar = 0;
			for (unsigned nb = nbits + 1; nb <= nbits_max; nb++)
				valvar += uint64_t((nb - nbits + 6) / 7) * h.hist[nb];
			const bool value_packed = valpck < valvar;
			const uint64_t value_data = value_packed ? valpck : valvar;
			// the estimate gate, integer_codec.h:327-335
			uint64_t indmin = pck_bytes(noutliers, 1);
			indmin = indmin > n ? n : indmin;
			const uint64_t selected = c.dataspace + c.metaspace;
			if (basic_meta + extra_meta + basic_data + value_data + indmin >= selected)
				continue;
			// the gaps of this width's outliers: pack the candidate, reduce its gap list
			if (!ws.p && (rc = ws.alloc(wsb)))
				return rc;
			if (basic_data > packed_cap) {
				if ((rc = packed.alloc(size_t(basic_data))))
					return rc;
				packed_cap = size_t(basic_data);
			}
			if (noutliers > list_cap) {
				if ((rc = idx.alloc(size_t(noutliers) * 8)) || (rc = val.alloc(size_t(noutliers) * 8)))
					return rc;
				list_cap = size_t(noutliers);
			}
			if ((rc = int(hipMemsetAsync(&facts->gap_space, 0, 2 * sizeof(unsigned long long), st))) ||
			    (rc = oroch_bitpfr_encode(dtype, src, n, mn, nbits, packed.p, static_cast<uint64_t *>(idx.p),
						      static_cast<uint64_t *>(val.p), size_t(noutliers),
						      reinterpret_cast<uint64_t *>(&facts->consumed[0]), ws.p, wsb, stream)))
				return rc;
			{
				profile_scope ps(OROCH_K_SELECT, st);
				hipLaunchKernelGGL(k_gap_facts, dim3(seq_grid(size_t(noutliers))), dim3(SQ_THREADS), 0, st,
						   static_cast<const uint64_t *>(idx.p), size_t(noutliers), facts);
			}
			if ((rc = int(hipGetLastError())) || (rc = fetch_facts(&h, facts, st)))
				return rc;
			const unsigned indnbits = h.gap_bits > 1 ? unsigned(h.gap_bits) : 1u;
			const uint64_t indpck = pck_bytes(noutliers, indnbits), indvar = h.gap_space;
			const bool index_packed = indpck < indvar;
			const uint64_t index_data = index_packed ? indpck : indvar;
			if (basic_meta + extra_meta + basic_data + value_data + index_data < selected) {
				c.encoding = OROCH_BITPFR;
				c.origin = mn;
				c.nbits = nbits;
				c.metaspace = basic_meta + extra_meta;
				c.dataspace = basic_data + value_data + index_data;
				d->noutliers = noutliers;
				d->value_encoding = value_packed ? OROCH_BITPCK : OROCH_VARINT;
				d->value_nbits = uint8_t(nbits_max - nbits);
				d->index_encoding = index_packed ? OROCH_BITPCK : OROCH_VARINT;
				d->index_nbits = uint8_t(indnbits);
			}
		}
	}
	d->encoding = c.encoding;
	d->nbits = uint8_t(c.nbits);
	d->origin = c.origin;
	d->dataspace = c.dataspace;
	d->metaspace = uint32_t(1 + c.metaspace);
	return 0;
}

namespace {

// A list of k unsigned 64-bit values (outlier positions or values) after the
// packed part: bitpck or varint (integer_codec.h:573-580).  `at` is the write
// position in dst, advanced by the list's bytes.
int put_list(unsigned encoding, unsigned nbits, const uint64_t *list, size_t k, uint8_t *dst, size_t cap,
	     uint64_t &at, hipStream_t st)
{
	if (k == 0)
		return 0;
	int rc;
	if (encoding == OROCH_BITPCK) {
		if (nbits < 1 || nbits > 64)
			return OROCH_E_NBITS;
		const size_t bytes = oroch_bitpck_space(k, nbits);
		if (at + bytes > cap)
			return OROCH_E_CAPACITY;
		if (((reinterpret_cast<uintptr_t>(dst) + at) & 15u) == 0) {
			rc = oroch_bitpck_encode(OROCH_U64, list, k, nbits, dst + at, st);
		} else {
			dev_buf tmp;
			if ((rc = tmp.alloc(bytes)) || (rc = oroch_bitpck_encode(OROCH_U64, list, k, nbits, tmp.p, st)) ||
			    (rc = int(hipMemcpyAsync(dst + at, tmp.p, bytes, hipMemcpyDeviceToDevice, st))) ||
			    (rc = int(hipStreamSynchronize(st))))
				return rc;
		}
		at += bytes;
		return rc;
	}
	// A list left at 'normal' (a descriptor nobody filled in) would be raw
	// size_t / unsigned values in the reference (integer_codec.h:573-580 through
	// encode_basic), a form no header can announce (:196-204): refused.
	if (encoding != OROCH_VARINT)
		return OROCH_E_ARGUMENT;
	dev_buf tmp, ws;
	const size_t wsb = oroch_varint_workspace(k);
	uint64_t bytes = 0;
	if ((rc = tmp.alloc(k * 10)) || (rc = ws.alloc(wsb)) ||
	    (rc = oroch_varint_encode(OROCH_U64, list, k, tmp.p, k * 10, ws.p, wsb, &bytes, st)))
		return rc;
	if (at + bytes > cap)
		return OROCH_E_CAPACITY;
	if ((rc = int(hipMemcpyAsync(dst + at, tmp.p, size_t(bytes), hipMemcpyDeviceToDevice, st))) ||
	    (rc = int(hipStreamSynchronize(st))))
		return rc;
	at += bytes;
	return 0;
}

// The reverse: k values from src + at into list (device), `at` advanced.
int get_list(unsigned encoding, unsigned nbits, const uint8_t *src, size_t src_bytes, uint64_t &at, size_t k,
	     uint64_t *list, seq_facts *facts, hipStream_t st)
{
	if (k == 0)
		return 0;
	int rc;
	if (encoding == OROCH_BITPCK) {
		if (nbits < 1 || nbits > 64)
			return OROCH_E_NBITS;
		const size_t bytes = oroch_bitpck_space(k, nbits);
		if (at + bytes > src_bytes)
			return OROCH_E_CAPACITY;
		if (((reinterpret_cast<uintptr_t>(src) + at) & 15u) == 0) {
			rc = oroch_bitpck_decode(OROCH_U64, src + at, k, nbits, list, st);
		} else {
			dev_buf tmp;
			if ((rc = tmp.alloc(bytes)) ||
			    (rc = int(hipMemcpyAsync(tmp.p, src + at, bytes, hipMemcpyDeviceToDevice, st))) ||
			    (rc = oroch_bitpck_decode(OROCH_U64, tmp.p, k, nbits, list, st)) ||
			    (rc = int(hipStreamSynchronize(st))))
				return rc;
		}
		at += bytes;
		return rc;
	}
	if (encoding != OROCH_VARINT)
		return OROCH_E_ARGUMENT; // see put_list
	// varints: at most ten bytes each, and no more than what is there
	size_t avail = src_bytes - size_t(at);
	avail = avail < k * 10 ? avail : k * 10;
	if (avail < k)
		return OROCH_E_CAPACITY;
	dev_buf tmp, ws;
	const size_t wsb = oroch_varint_workspace(avail);
	unsigned long long consumed = 0;
	if ((rc = tmp.alloc(avail + 16)) || (rc = ws.alloc(wsb)) ||
	    (rc = int(hipMemcpyAsync(tmp.p, src + at, avail, hipMemcpyDeviceToDevice, st))) ||
	    (rc = int(hipMemsetAsync(&facts->consumed[0], 0, sizeof(unsigned long long), st))) ||
	    (rc = varint_decode_counted(OROCH_U64, tmp.p, avail, k, list, ws.p, wsb,
					reinterpret_cast<uint64_t *>(&facts->consumed[0]), st)) ||
	    (rc = int(hipMemcpyAsync(&consumed, &facts->consumed[0], sizeof consumed, hipMemcpyDeviceToHost, st))) ||
	    (rc = int(hipStreamSynchronize(st))))
		return rc;
	if (consumed == 0)
		return OROCH_E_CAPACITY; // fewer than k values in the bytes that are there
	at += consumed;
	return 0;
}

} // namespace

extern "C" int oroch_sequence_encode(int dtype, const void *src, size_t n, const oroch_sequence_desc *d, void *dst,
				     size_t cap, uint64_t *bytes_host, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || !bytes_host || (n && !src) || (reinterpret_cast<uintptr_t>(dst) & 15u))
		return OROCH_E_ARGUMENT;
	*bytes_host = 0;
	if (n == 0)
		return 0;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const unsigned W = width_of(dtype);
	const size_t w = W / 8;
	const int unsigned_dtype = dtype | 1;
	int rc = 0;
	switch (d->encoding) {
	case OROCH_NAUGHT: // naught.h:37-40
		return 0;
	case OROCH_NORMAL: // normal.h:42-49
		if (n * w > cap || !dst)
			return OROCH_E_CAPACITY;
		if ((rc = int(hipMemcpyAsync(dst, src, n * w, hipMemcpyDeviceToDevice, st))) ||
		    (rc = int(hipStreamSynchronize(st))))
			return rc;
		*bytes_host = n * w;
		return 0;
	case OROCH_VARINT:  // varint.h:133-139 with zigzag
	case OROCH_VARFOR: { // the same with origin_codec
		dev_buf shifted, tmp, ws;
		const size_t worst = n * ((W + 6) / 7), wsb = oroch_varint_workspace(n);
		const void *from = src;
		int as = dtype;
		if (d->encoding == OROCH_VARFOR) {
			if ((rc = shifted.alloc(n * w)))
				return rc;
			rc = dispatch_dtype(dtype, [&](auto e) {
				using E = decltype(e);
				hipLaunchKernelGGL((k_seq_shift<E, true>), dim3(seq_grid(n)), dim3(SQ_THREADS), 0, st,
						   static_cast<const typename E::S *>(src), n,
						   typename E::C(d->origin & uint64_t(E::MASK)),
						   static_cast<typename E::S *>(shifted.p));
				return int(hipGetLastError());
			});
			if (rc)
				return rc;
			from = shifted.p;
			as = unsigned_dtype;
		}
		uint64_t bytes = 0;
		if ((rc = tmp.alloc(worst)) || (rc = ws.alloc(wsb)) ||
		    (rc = oroch_varint_encode(as, from, n, tmp.p, worst, ws.p, wsb, &bytes, st)))
			return rc;
		if (bytes > cap || !dst)
			return OROCH_E_CAPACITY;
		if ((rc = int(hipMemcpyAsync(dst, tmp.p, size_t(bytes), hipMemcpyDeviceToDevice, st))) ||
		    (rc = int(hipStreamSynchronize(st))))
			return rc;
		*bytes_host = bytes;
		return 0;
	}
	case OROCH_BITPCK:
	case OROCH_BITFOR: {
		if (d->nbits < 1 || d->nbits > W)
			return OROCH_E_NBITS;
		const size_t bytes = oroch_bitpck_space(n, d->nbits);
		if (bytes > cap || !dst)
			return OROCH_E_CAPACITY;
		rc = d->encoding == OROCH_BITPCK ? oroch_bitpck_encode(dtype, src, n, d->nbits, dst, st)
						 : oroch_bitfor_encode(dtype, src, n, d->origin, d->nbits, dst, st);
		if (rc || (rc = int(hipStreamSynchronize(st))))
			return rc;
		*bytes_host = bytes;
		return 0;
	}
	case OROCH_BITPFR: { // integer_codec.h:562-581
		if (d->nbits < 1 || d->nbits >= W)
			return OROCH_E_NBITS;
		if (d->noutliers > n)
			return OROCH_E_ARGUMENT;
		const size_t k = size_t(d->noutliers);
		uint64_t at = oroch_bitpck_space(n, d->nbits);
		if (at > cap || !dst)
			return OROCH_E_CAPACITY;
		dev_buf idx, val, ws, count;
		const size_t wsb = oroch_bitpfr_workspace(n);
		uint64_t found = 0;
		if ((rc = idx.alloc(k * 8)) || (rc = val.alloc(k * 8)) || (rc = ws.alloc(wsb)) || (rc = count.alloc(8)) ||
		    (rc = oroch_bitpfr_encode(dtype, src, n, d->origin, d->nbits, dst, static_cast<uint64_t *>(idx.p),
					      static_cast<uint64_t *>(val.p), k, static_cast<uint64_t *>(count.p), ws.p,
					      wsb, st)) ||
		    (rc = int(hipMemcpyAsync(&found, count.p, 8, hipMemcpyDeviceToHost, st))) ||
		    (rc = int(hipStreamSynchronize(st))))
			return rc;
		if (found != d->noutliers)
			return OROCH_E_ARGUMENT;
		uint8_t *out = static_cast<uint8_t *>(dst);
		if ((rc = put_list(d->index_encoding, d->index_nbits, static_cast<const uint64_t *>(idx.p), k, out, cap,
				   at, st)) ||
		    (rc = put_list(d->value_encoding, d->value_nbits, static_cast<const uint64_t *>(val.p), k, out, cap,
				   at, st)) ||
		    (rc = int(hipStreamSynchronize(st))))
			return rc;
		*bytes_host = at;
		return 0;
	}
	}
	return OROCH_E_ARGUMENT;
}

extern "C" int oroch_sequence_decode(int dtype, const void *src, size_t src_bytes, size_t n,
				     const oroch_sequence_desc *d, void *dst, uint64_t *consumed_host, void *stream)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || (n && !dst) || (reinterpret_cast<uintptr_t>(src) & 15u))
		return OROCH_E_ARGUMENT;
	if (consumed_host)
		*consumed_host = 0;
	if (n == 0)
		return 0;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const unsigned W = width_of(dtype);
	const size_t w = W / 8;
	int rc = 0;
	uint64_t used = 0;
	switch (d->encoding) {
	case OROCH_NAUGHT: // naught.h:42-47
		rc = dispatch_dtype(dtype, [&](auto e) {
			using E = decltype(e);
			hipLaunchKernelGGL(k_seq_fill<E>, dim3(seq_grid(n)), dim3(SQ_THREADS), 0, st,
					   static_cast<typename E::S *>(dst), n, typename E::C(d->origin & uint64_t(E::MASK)));
			return int(hipGetLastError());
		});
		break;
	case OROCH_NORMAL: // normal.h:51-58
		if (n * w > src_bytes || !src)
			return OROCH_E_CAPACITY;
		rc = int(hipMemcpyAsync(dst, src, n * w, hipMemcpyDeviceToDevice, st));
		used = n * w;
		break;
	case OROCH_VARINT:
	case OROCH_VARFOR: {
		if (!src || src_bytes < n)
			return OROCH_E_CAPACITY;
		size_t avail = n * ((W + 6) / 7);
		avail = avail < src_bytes ? avail : src_bytes;
		dev_buf ws, count;
		const size_t wsb = oroch_varint_workspace(avail);
		const int as = d->encoding == OROCH_VARFOR ? (dtype | 1) : dtype;
		if ((rc = ws.alloc(wsb)) || (rc = count.alloc(8)) || (rc = int(hipMemsetAsync(count.p, 0, 8, st))) ||
		    (rc = varint_decode_counted(as, src, avail, n, dst, ws.p, wsb, static_cast<uint64_t *>(count.p), st)))
			return rc;
		if (d->encoding == OROCH_VARFOR) {
			rc = dispatch_dtype(dtype, [&](auto e) {
				using E = decltype(e);
				hipLaunchKernelGGL((k_seq_shift<E, false>), dim3(seq_grid(n)), dim3(SQ_THREADS), 0, st,
						   static_cast<const typename E::S *>(dst), n,
						   typename E::C(d->origin & uint64_t(E::MASK)),
						   static_cast<typename E::S *>(dst));
				return int(hipGetLastError());
			});
			if (rc)
				return rc;
		}
		if ((rc = int(hipMemcpyAsync(&used, count.p, 8, hipMemcpyDeviceToHost, st))) ||
		    (rc = int(hipStreamSynchronize(st))))
			return rc;
		if (used == 0)
			return OROCH_E_CAPACITY; // the bytes hold fewer than n values
		break;
	}
	case OROCH_BITPCK:
	case OROCH_BITFOR:
		if (d->nbits < 1 || d->nbits > W)
			return OROCH_E_NBITS;
		used = oroch_bitpck_space(n, d->nbits);
		if (used > src_bytes || !src)
			return OROCH_E_CAPACITY;
		rc = d->encoding == OROCH_BITPCK ? oroch_bitpck_decode(dtype, src, n, d->nbits, dst, st)
						 : oroch_bitfor_decode(dtype, src, n, d->origin, d->nbits, dst, st);
		break;
	case OROCH_BITPFR: { // integer_codec.h:583-609
		if (d->nbits < 1 || d->nbits >= W)
			return OROCH_E_NBITS;
		if (d->noutliers > n)
			return OROCH_E_ARGUMENT;
		const size_t k = size_t(d->noutliers);
		used = oroch_bitpck_space(n, d->nbits);
		if (used > src_bytes || !src)
			return OROCH_E_CAPACITY;
		dev_buf idx, val, ws, facts;
		const size_t wsb = oroch_bitpfr_workspace(n);
		const uint8_t *in = static_cast<const uint8_t *>(src);
		if ((rc = idx.alloc(k * 8)) || (rc = val.alloc(k * 8)) || (rc = ws.alloc(wsb)) ||
		    (rc = facts.alloc(sizeof(seq_facts))) ||
		    (rc = get_list(d->index_encoding, d->index_nbits, in, src_bytes, used, k, static_cast<uint64_t *>(idx.p),
				   static_cast<seq_facts *>(facts.p), st)) ||
		    (rc = get_list(d->value_encoding, d->value_nbits, in, src_bytes, used, k, static_cast<uint64_t *>(val.p),
				   static_cast<seq_facts *>(facts.p), st)))
			return rc;
		if (k && W < 64) {
			hipLaunchKernelGGL(k_mask64, dim3(seq_grid(k)), dim3(SQ_THREADS), 0, st, static_cast<uint64_t *>(val.p),
					   k, mask_of(dtype));
			if ((rc = int(hipGetLastError())))
				return rc;
		}
		if ((rc = oroch_bitpfr_decode(dtype, src, n, d->origin, d->nbits, static_cast<const uint64_t *>(idx.p),
					      static_cast<const uint64_t *>(val.p), k, dst, ws.p, wsb, st)) ||
		    (rc = int(hipStreamSynchronize(st))))
			return rc;
		break;
	}
	default:
		return OROCH_E_ARGUMENT;
	}
	if (rc == 0)
		rc = int(hipStreamSynchronize(st));
	if (rc == 0 && consumed_host)
		*consumed_host = used;
	return rc;
}

// ---- host pointers ---------------------------------------------------------------------

extern "C" int oroch_sequence_select_host(int dtype, const void *src_host, size_t n, oroch_sequence_desc *d)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || (n && !src_host))
		return OROCH_E_ARGUMENT;
	dev_buf in;
	const size_t bytes = n << (dtype >> 1);
	int rc;
	if (n && ((rc = in.alloc(bytes)) || (rc = staged_copy(in.p, src_host, bytes, true, nullptr))))
		return rc;
	return oroch_sequence_select(dtype, in.p, n, d, nullptr);
}

extern "C" int oroch_sequence_encode_host(int dtype, const void *src_host, size_t n, const oroch_sequence_desc *d,
					  void *dst_host, size_t cap, uint64_t *bytes_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || !bytes_host || (n && !src_host))
		return OROCH_E_ARGUMENT;
	*bytes_host = 0;
	if (n == 0)
		return 0;
	dev_buf in, out;
	const size_t bytes = n << (dtype >> 1);
	int rc;
	if ((rc = in.alloc(bytes)) || (rc = out.alloc(cap + 16)) || (rc = staged_copy(in.p, src_host, bytes, true, nullptr)) ||
	    (rc = oroch_sequence_encode(dtype, in.p, n, d, out.p, cap, bytes_host, nullptr)))
		return rc;
	if (*bytes_host && !dst_host)
		return OROCH_E_ARGUMENT;
	return *bytes_host ? staged_copy(dst_host, out.p, size_t(*bytes_host), false, nullptr) : 0;
}

extern "C" int oroch_sequence_decode_host(int dtype, const void *src_host, size_t src_bytes, size_t n,
					  const oroch_sequence_desc *d, void *dst_host, uint64_t *consumed_host)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || (n && !dst_host) || (src_bytes && !src_host))
		return OROCH_E_ARGUMENT;
	if (consumed_host)
		*consumed_host = 0;
	if (n == 0)
		return 0;
	dev_buf in, out;
	const size_t bytes = n << (dtype >> 1);
	int rc;
	if ((rc = in.alloc(src_bytes + 16)) || (rc = out.alloc(bytes)) ||
	    (src_bytes && (rc = staged_copy(in.p, src_host, src_bytes, true, nullptr))) ||
	    (rc = oroch_sequence_decode(dtype, in.p, src_bytes, n, d, out.p, consumed_host, nullptr)))
		return rc;
	return staged_copy(dst_host, out.p, bytes, false, nullptr);
}

// ---- metadata bytes (host arithmetic) -------------------------------------------------

extern "C" int oroch_sequence_meta_encode(int dtype, const oroch_sequence_desc *d, void *dst_host, size_t cap)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || !dst_host)
		return OROCH_E_ARGUMENT;
	uint8_t tmp[40], *p = tmp;
	auto put = [&](uint64_t u) { // varint.h:74-83
		while (u >= 0x80) {
			*p++ = uint8_t(u | 0x80);
			u >>= 7;
		}
		*p++ = uint8_t(u);
	};
	const unsigned e = d->encoding;
	if (e > OROCH_BITPFR)
		return OROCH_E_ARGUMENT;
	*p++ = uint8_t(e);
	if (e == OROCH_NAUGHT || e == OROCH_VARFOR || e == OROCH_BITFOR || e == OROCH_BITPFR)
		put(zigzag_of(dtype, d->origin & mask_of(dtype)));
	if (e == OROCH_BITPCK || e == OROCH_BITFOR || e == OROCH_BITPFR)
		*p++ = d->nbits;
	if (e == OROCH_BITPFR) {
		put(d->noutliers);
		*p++ = d->index_encoding == OROCH_BITPCK ? d->index_nbits : 0;
		*p++ = d->value_encoding == OROCH_BITPCK ? d->value_nbits : 0;
	}
	const size_t bytes = size_t(p - tmp);
	if (bytes > cap)
		return OROCH_E_CAPACITY;
	std::memcpy(dst_host, tmp, bytes);
	return int(bytes);
}

extern "C" int oroch_sequence_meta_decode(int dtype, const void *src_host, size_t src_bytes, oroch_sequence_desc *d)
{
	if (dtype < 0 || dtype > 7)
		return OROCH_E_DTYPE;
	if (!d || !src_host)
		return OROCH_E_ARGUMENT;
	const uint8_t *s = static_cast<const uint8_t *>(src_host);
	size_t at = 0;
	bool short_read = false;
	auto byte = [&]() -> unsigned {
		if (at >= src_bytes) {
			short_read = true;
			return 0;
		}
		return s[at++];
	};
	auto get = [&](uint64_t mask) { // varint.h:85-115
		uint64_t value = 0;
		for (unsigned shift = 0;; shift += 7) {
			const uint64_t b = byte();
			if (shift < 64)
				value |= (b & 0x7f) << shift;
			if (!(b & 0x80))
				break;
		}
		return value & mask;
	};
	std::memset(d, 0, sizeof *d);
	d->index_encoding = OROCH_NORMAL;
	d->value_encoding = OROCH_NORMAL;
	const unsigned e = byte();
	if (e > OROCH_BITPFR)
		return OROCH_E_ARGUMENT;
	d->encoding = uint8_t(e);
	if (e == OROCH_NAUGHT || e == OROCH_VARFOR || e == OROCH_BITFOR || e == OROCH_BITPFR)
		d->origin = unzigzag_of(dtype, get(mask_of(dtype)));
	if (e == OROCH_BITPCK || e == OROCH_BITFOR || e == OROCH_BITPFR)
		d->nbits = uint8_t(byte());
	if (e == OROCH_BITPFR) {
		d->noutliers = get(~uint64_t(0));
		unsigned nb = byte();
		d->index_encoding = nb ? OROCH_BITPCK : OROCH_VARINT;
		d->index_nbits = uint8_t(nb);
		nb = byte();
		d->value_encoding = nb ? OROCH_BITPCK : OROCH_VARINT;
		d->value_nbits = uint8_t(nb);
	}
	if (short_read)
		return OROCH_E_CAPACITY;
	d->metaspace = uint32_t(at);
	return int(at);
}
